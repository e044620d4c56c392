// step_shape.hip - design probe for k_step (round 6; diagnostic, not part of the library): a kernel with k_step's memory shape, a DUMMY
// dependent arithmetic chain of `CH` fp32 FMAs between its loads and stores (the integrator is ~100-150 dependent vector instructions), and
// the two things VERDICT r05 item 1 asks about as template parameters:
//   TILES  envs per lane, ALL tiles' loads issued before the first tile's arithmetic (tile t of wavefront w = 64-env group w + t * waves):
//          1 = k_step as it is; 2 / 4 = the grid shrinks so that 1 M (4 M) envs are ONE resident round, with the next tile's loads in flight
//          while this tile computes and stores
//   BK     bookkeeping of the 1.3 % of lanes whose episode ends (pseudo-random by (env, launch)):
//          0 none (the 101 algorithmic bytes)          1 running return in / out only (every lane; what tracking cannot avoid)
//          2 = 1 + EAGER sbd / reset count / episode start dword loads in every lane, five scattered dword stores in ending lanes (k_step < 786 432 envs)
//          3 = 1 + the same three dwords loaded LAZILY in ending lanes, five scattered dword stores (k_step beyond)
//          4 = 1 + lazy ONE 32-byte record per env {sbd, reset count, start, last return, last length, pad}: two b128 loads, two b128 stores (a full sector)
//          5 = 1 + lazy ONE 16-byte record {sbd, reset count, start, last length}: one b128 load, one b128 store + one dword store (last return)
//          6 = 1 + an EAGER 4-byte "hot" word per env (reset count + sbd-is-None: all the reward and the reset draw need) and the 16-byte record loaded
//              lazily OFF the critical path: nothing but the record's own store waits for it
// Reports us per launch and the fraction of 8 TB/s on the 101 algorithmic bytes, like bench.py's `legs.step*`.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/_build/step_shape tools/micro/step_shape.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t rsrc(const void *p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, -1, 0x00020000); }
__device__ __forceinline__ float ld(rsrc_t r, uint32_t v, uint32_t s) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, v, s, 0)); }
__device__ __forceinline__ void st(rsrc_t r, uint32_t v, uint32_t s, float x, int aux) {
    if (aux) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, x), r, v, s, 2);   // non-temporal
    else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, x), r, v, s, 0);
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct Args {
    float *state; const float *act; float *rew; uint8_t *done;
    float *ep_ret; float *soa;   // soa: 5 arrays [n] (sbd, rc, es, last_ret, last_len)
    u32x4 *rec;                  // 32-byte (2 x u32x4) or 16-byte records
    uint32_t n, launch, waves, nt;
};

template <int TILES, int BK, int CH, int DEP = 1>
__global__ __launch_bounds__(256) void k_shape(const Args a) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u, w = tid >> 6;
    const uint32_t col = a.n * 4u;
    const rsrc_t rs = rsrc(a.state), ra = rsrc(a.act), re = rsrc(a.ep_ret), rx = rsrc(a.soa);
    float s[TILES][10], ac[TILES][4], er[TILES], x[TILES][3];
    uint32_t idx[TILES];
    bool ok[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
        const uint32_t i = (w + (uint32_t)t * a.waves) * 64u + lane;
        ok[t] = i < a.n;
        idx[t] = ok[t] ? i : a.n - 1u;
        const uint32_t off = idx[t] * 4u;
#pragma unroll
        for (int c = 0; c < 10; ++c) s[t][c] = ld(rs, off, c * col);
#pragma unroll
        for (int c = 0; c < 4; ++c) ac[t][c] = ld(ra, off, c * col);
        er[t] = 0.f;
        if (BK >= 1) er[t] = ld(re, off, 0);
        x[t][0] = x[t][1] = x[t][2] = 0.f;
        if (BK == 2) {
#pragma unroll
            for (int c = 0; c < 3; ++c) x[t][c] = ld(rx, off, c * col);
        }
        if (BK == 6) x[t][0] = ld(rx, off, 0);
    }
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
        const uint32_t i = idx[t], off = i * 4u;
        // the "integrator": a dependent chain
        // DEP: what the chain (and so every store) depends on - 1: every state load (the chain reads them round robin); 0: the action loads only
        // (state store c then depends on state load c and the actions); 2: actions only, then the sum of all state loads is added (every store
        // behind every load, whatever CH); 3: the actions and state components 3..6 (the attitude quaternion: what the real integrator's acceleration needs)
        float r = ac[t][0];
#pragma unroll
        for (int k = 0; k < CH; ++k) r = __builtin_fmaf(r, (k & 1) ? ac[t][1] : ac[t][2], DEP == 1 ? s[t][k % 10] : DEP == 3 ? s[t][3 + (k & 3)] : ac[t][3]);
        if (DEP == 2) {
            float sum = 0.f;
#pragma unroll
            for (int c = 0; c < 10; ++c) sum += s[t][c];
            r = __builtin_fmaf(sum, 1e-30f, r);
        }
        r = __builtin_fmaf(r, 1e-30f, ac[t][3]);
        // 1.3 % of the lanes end an episode
        uint32_t h = (i * 2654435761u) ^ (a.launch * 40503u);
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const bool fin = ((h & 0xffffu) < 852u) != (r > 1e30f);   // (known only once the chain has run, like `done`)
        float l0 = 0.f, l1 = 0.f;
        if (BK == 3 && fin) {
#pragma unroll
            for (int c = 0; c < 3; ++c) x[t][c] = ld(rx, off, c * col);
        }
        u32x4 q0 = {0, 0, 0, 0}, q1 = {0, 0, 0, 0};
        if (BK == 4 && fin) {
            q0 = a.rec[2 * (size_t)i];
            q1 = a.rec[2 * (size_t)i + 1];
            x[t][0] = __builtin_bit_cast(float, q0.x); x[t][1] = __builtin_bit_cast(float, q0.y); x[t][2] = __builtin_bit_cast(float, q0.z);
        }
        if (BK == 6 && fin) q0 = a.rec[(size_t)i];   // lazily; only the record's own store below uses it
        if (BK == 5 && fin) {
            q0 = a.rec[(size_t)i];
            x[t][0] = __builtin_bit_cast(float, q0.x); x[t][1] = __builtin_bit_cast(float, q0.y); x[t][2] = __builtin_bit_cast(float, q0.z);
        }
        if (BK >= 2 && fin) { r += x[t][0]; l0 = er[t] + r; l1 = x[t][2] + 1.f; }
        if (ok[t]) {
            st(rsrc(a.rew), off, 0, r, a.nt);
            if (a.nt) __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(fin ? 1 : 0), rsrc(a.done), i, 0, 2);
            else __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(fin ? 1 : 0), rsrc(a.done), i, 0, 0);
            if (BK >= 1) st(re, off, 0, fin ? 0.f : er[t] + r, a.nt);
            if (fin) {
                if (BK == 2 || BK == 3) {
                    st(rx, off, 0 * col, x[t][0] + 1.f, 0); st(rx, off, 1 * col, x[t][1] + 1.f, 0); st(rx, off, 2 * col, x[t][2] + 1.f, 0);
                    st(rx, off, 3 * col, l0, 0); st(rx, off, 4 * col, l1, 0);
                }
                if (BK == 4) {
                    q0.x += 1; q0.y += 1; q0.z += 1; q0.w = __builtin_bit_cast(uint32_t, l0); q1.x = __builtin_bit_cast(uint32_t, l1);
                    a.rec[2 * (size_t)i] = q0; a.rec[2 * (size_t)i + 1] = q1;
                }
                if (BK == 6) {
                    st(rx, off, 0, x[t][0] + 1.f, 0);   // the hot word
                    st(rx, off, 3 * col, l0, 0);
                }
                if (BK == 5) {
                    q0.x += 1; q0.y += 1; q0.z += 1; q0.w = __builtin_bit_cast(uint32_t, l1);
                    a.rec[(size_t)i] = q0;
                    st(rx, off, 3 * col, l0, 0);
                }
            }
#pragma unroll
            for (int c = 0; c < 10; ++c) st(rs, off, c * col, s[t][c] + r * 1e-30f, a.nt);
            if (BK == 6 && fin) {
                q0.x += 1; q0.y += 1; q0.z += 1; q0.w = __builtin_bit_cast(uint32_t, l1 + __builtin_bit_cast(float, q0.z));
                a.rec[(size_t)i] = q0;
            }
        }
    }
}

template <int TILES, int BK, int CH, int DEP = 1> void run(uint32_t n, int block, int nt, const char *tag) {
    const int RING = (int)(n >= 1048576u ? 16 : 64);
    Args a; memset(&a, 0, sizeof(a));
    float *act;
    CK(hipMalloc(&a.state, 40ull * n)); CK(hipMalloc(&act, 16ull * n * RING)); CK(hipMalloc(&a.rew, 4ull * n * RING)); CK(hipMalloc(&a.done, (size_t)n * RING));
    CK(hipMalloc(&a.ep_ret, 4ull * n)); CK(hipMalloc(&a.soa, 20ull * n)); CK(hipMalloc(&a.rec, 32ull * n));
    CK(hipMemset(a.state, 0, 40ull * n)); CK(hipMemset(act, 0, 16ull * n * RING)); CK(hipMemset(a.ep_ret, 0, 4ull * n)); CK(hipMemset(a.soa, 0, 20ull * n)); CK(hipMemset(a.rec, 0, 32ull * n));
    a.n = n; a.nt = nt;
    const uint32_t groups = (n + 63) / 64, waves = (groups + TILES - 1) / TILES, wpb = block / 64;
    a.waves = ((waves + wpb - 1) / wpb) * wpb;   // whole workgroups
    const int reps = (int)(3000ull * 65536 / n) + 200;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float *rew0 = a.rew; uint8_t *done0 = a.done;
    auto go = [&](int k) {
        a.act = act + (size_t)(k % RING) * 4 * n; a.rew = rew0 + (size_t)(k % RING) * n; a.done = done0 + (size_t)(k % RING) * n; a.launch = k;
        hipLaunchKernelGGL((k_shape<TILES, BK, CH, DEP>), dim3(a.waves / wpb), dim3(block), 0, 0, a);
    };
    for (int k = 0; k < 100; ++k) go(k);
    CK(hipDeviceSynchronize());
    float best = 1e30f, sum = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        for (int k = 0; k < reps; ++k) go(k);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best; sum += ms;
    }
    const double us = best * 1e3 / reps;
    printf("| %u | %d | %s | %d | %d | %d | %d | %s | %.2f | %.3f |\n", n, block, tag, TILES, BK, CH, DEP, nt ? "nt" : "wb", us, 101.0 * n / us / 1e6 / 8.0);
    fflush(stdout);
    CK(hipFree(a.state)); CK(hipFree(act)); CK(hipFree(rew0)); CK(hipFree(done0)); CK(hipFree(a.ep_ret)); CK(hipFree(a.soa)); CK(hipFree(a.rec));
}

template <int TILES, int CH> void all_bk(uint32_t n, int block, int nt) {
    run<TILES, 0, CH>(n, block, nt, "none");
    run<TILES, 1, CH>(n, block, nt, "ret");
    run<TILES, 2, CH>(n, block, nt, "eager soa");
    run<TILES, 3, CH>(n, block, nt, "lazy soa");
    run<TILES, 4, CH>(n, block, nt, "lazy rec32");
    run<TILES, 5, CH>(n, block, nt, "lazy rec16");
    run<TILES, 6, CH>(n, block, nt, "hot word + late rec16");
}
int main(int argc, char **argv) {
    const bool full = argc > 1 && !strcmp(argv[1], "full");
    printf("| envs | threads | bookkeeping | tiles per lane | BK | chain | DEP | stores | us per launch | frac of 8 TB/s on 101 B |\n|---|---|---|---|---|---|---|---|---|---|\n");
    for (uint32_t n : {65536u, 262144u, 1048576u, 4194304u}) {
        const int nt = (n >= 196608u && n < 786432u) ? 1 : 0;
        for (int block : {128, 256}) {
            all_bk<1, 120>(n, block, nt);
            if (full) all_bk<2, 120>(n, block, nt);
            if (full && n >= 1048576u) all_bk<4, 120>(n, block, nt);
        }
        // what the stores wait for (DEP), and how long the chain is
        run<1, 1, 0, 0>(n, 128, nt, "ret"); run<1, 1, 0, 2>(n, 128, nt, "ret");
        run<1, 1, 120, 0>(n, 128, nt, "ret"); run<1, 1, 120, 2>(n, 128, nt, "ret"); run<1, 1, 120, 3>(n, 128, nt, "ret"); run<1, 1, 120, 1>(n, 128, nt, "ret");
        run<1, 1, 240, 0>(n, 128, nt, "ret"); run<1, 1, 240, 1>(n, 128, nt, "ret");
        run<1, 5, 120, 0>(n, 128, nt, "lazy rec16"); run<1, 5, 120, 3>(n, 128, nt, "lazy rec16");
        run<2, 1, 120, 0>(n, 128, nt, "ret"); run<2, 1, 120, 3>(n, 128, nt, "ret");
    }
    return 0;
}
