// store_patterns.hip - which store pattern does the memory system of an MI355X absorb fastest once a fused rollout's
// trajectory no longer fits the 256 MiB Infinity Cache?  (diagnostic, not part of the library)
//
// Workload shape of k_rollout's trajectory: per env-step 16 dword "columns" (4 action + 10 obs + 1 reward + 1 column
// standing in for the done bytes), time-major SoA: element (t, c, env) at  arena + ((t*16 + c) * pitch + env) * 4.
// Variants:
//   width 1 : every lane stores one dword per column (what round 1's kernels do): 16 store instructions per env-step,
//             each 256 contiguous bytes per wavefront
//   width 4 : lane (cg = lane>>4, eq = lane&15) stores 16 bytes = envs 4eq..4eq+3 of column 4j+cg: 4 instructions per
//             env-step, each 4 x 256 contiguous bytes per wavefront (what an LDS hand-over tile read back with
//             ds_read_b128 gives)
//   tiled   : [wave][t][c][64] - every wavefront owns one contiguous T*4 KiB region (best-case DRAM page locality)
//   pitch   : n + pad floats between columns (power-of-two column strides are a channel/bank aliasing suspect)
//   policy  : default / sc0 sc1 (write-through) / nt / sc1
// Every launch writes a fresh arena out of a ring > 1.5 GiB, so nothing is absorbed by rewriting cache-resident lines.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/_build/sp tools/micro/store_patterns.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t rsrc(const void *p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, -1, 0x00020000); }

constexpr int C = 16;
typedef float v4f __attribute__((ext_vector_type(4)));

template <int AUX> __device__ __forceinline__ void st4(float *p, v4f v);
template <> __device__ __forceinline__ void st4<0>(float *p, v4f v) { asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
template <> __device__ __forceinline__ void st4<17>(float *p, v4f v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
template <> __device__ __forceinline__ void st4<2>(float *p, v4f v) { asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory"); }
template <> __device__ __forceinline__ void st4<16>(float *p, v4f v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }

// WIDTH 1 | 4, TILED 0 | 1
template <int AUX, int WIDTH, int TILED>
__global__ __launch_bounds__(256) void k_store(float *arena, uint32_t n, uint32_t pitch, int T) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t lane = threadIdx.x & 63u, wave_first = i - lane;
    float v = (float)i;
    if constexpr (WIDTH == 1) {
        if constexpr (TILED) {
            float *base = arena + (size_t)(wave_first >> 6) * T * C * 64;
            for (int t = 0; t < T; ++t) {
                v += 1.0f;
                const rsrc_t r = rsrc(base);
#pragma unroll
                for (int c = 0; c < C; ++c) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), r, lane * 4u, c * 256u, AUX);
                base += C * 64;
            }
        } else {
            const uint32_t col = pitch * 4u;
            float *base = arena;
            for (int t = 0; t < T; ++t) {
                v += 1.0f;
                const rsrc_t r = rsrc(base);
#pragma unroll
                for (int c = 0; c < C; ++c) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), r, i * 4u, c * col, AUX);
                base += (size_t)C * pitch;
            }
        }
    } else {
        const uint32_t cg = lane >> 4, eq = lane & 15u;
        if constexpr (TILED) {
            float *p = arena + (size_t)(wave_first >> 6) * T * C * 64 + lane * 4u;
            for (int t = 0; t < T; ++t) {
                v += 1.0f;
#pragma unroll
                for (int j = 0; j < C / 4; ++j) st4<AUX>(p + j * 256, v4f{v, v, v, v});
                p += C * 64;
            }
        } else {
            float *p = arena + (size_t)cg * pitch + wave_first + 4u * eq;
            for (int t = 0; t < T; ++t) {
                v += 1.0f;
#pragma unroll
                for (int j = 0; j < C / 4; ++j) st4<AUX>(p + (size_t)(4 * j) * pitch, v4f{v, v, v, v});
                p += (size_t)C * pitch;
            }
        }
    }
}

// "sliced": the grid covers only `slice` envs; every lane walks the batch slice by slice (all T steps of one slice,
// then the next), so the chip's concurrent write window is slice * 64 B per time step instead of n * 64 B
template <int AUX, int WIDTH>
__global__ __launch_bounds__(1024) void k_store_sliced(float *arena, uint32_t n, uint32_t pitch, int T, uint32_t slice) {
    const uint32_t i0 = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t i = i0; i < n; i += slice) {
        const uint32_t wave_first = i - lane;
        float v = (float)i;
        if constexpr (WIDTH == 1) {
            const uint32_t col = pitch * 4u;
            float *base = arena;
            for (int t = 0; t < T; ++t) {
                v += 1.0f;
                const rsrc_t r = rsrc(base);
#pragma unroll
                for (int c = 0; c < C; ++c) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), r, i * 4u, c * col, AUX);
                base += (size_t)C * pitch;
            }
        } else {
            const uint32_t cg = lane >> 4, eq = lane & 15u;
            float *p = arena + (size_t)cg * pitch + wave_first + 4u * eq;
            for (int t = 0; t < T; ++t) {
                v += 1.0f;
#pragma unroll
                for (int j = 0; j < C / 4; ++j) st4<AUX>(p + (size_t)(4 * j) * pitch, v4f{v, v, v, v});
                p += (size_t)C * pitch;
            }
        }
    }
}

// "multi": the grid covers n / M lanes and every lane stores the columns of M envs (i, i + n/M, ...) in each time step:
// the same concurrent write window as the plain kernel, from 1/M of the wavefronts
template <int AUX, int M>
__global__ __launch_bounds__(256) void k_store_multi(float *arena, uint32_t n, uint32_t pitch, int T) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t part = n / M;
    if (i >= part) return;
    float v = (float)i;
    const uint32_t col = pitch * 4u;
    float *base = arena;
    for (int t = 0; t < T; ++t) {
        v += 1.0f;
        const rsrc_t r = rsrc(base);
#pragma unroll
        for (int m = 0; m < M; ++m)
#pragma unroll
            for (int c = 0; c < C; ++c)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), r, (i + m * part) * 4u, c * col, AUX);
        base += (size_t)C * pitch;
    }
}

// "sim": a model of a fused rollout kernel with M envs per lane: per time step and env a DEPENDENT chain of `chain` FMAs
// (the integrator's latency-bound arithmetic; the M chains of a lane are independent of each other), then the env's 16
// column stores.  CONTIG: the wavefront's M sub-batches are adjacent (envs wave_first*M + m*64 + lane) instead of
// n/M apart.  The grid covers `lanes` lanes and walks the batch in slices of lanes*M envs.
template <int AUX, int M, bool CONTIG>
__global__ __launch_bounds__(256) void k_sim(float *arena, uint32_t n, uint32_t pitch, int T, int chain, uint32_t lanes) {
    const uint32_t i0 = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t col = pitch * 4u;
    for (uint32_t base = 0; base < n; base += lanes * M) {
        uint32_t env[M];
        float v[M];
#pragma unroll
        for (int m = 0; m < M; ++m) {
            env[m] = CONTIG ? base + (i0 - lane) * M + m * 64 + lane : base + i0 + m * lanes;
            v[m] = (float)env[m] * 1e-6f;
        }
        float *b = arena;
        for (int t = 0; t < T; ++t) {
            for (int k = 0; k < chain; ++k) {
#pragma unroll
                for (int m = 0; m < M; ++m) v[m] = __builtin_fmaf(v[m], 0.999f, 0.25f);
            }
            const rsrc_t r = rsrc(b);
#pragma unroll
            for (int m = 0; m < M; ++m)
#pragma unroll
                for (int c = 0; c < C; ++c)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v[m]), r, env[m] * 4u, c * col, AUX);
            b += (size_t)C * pitch;
        }
    }
}

struct Ring { std::vector<float *> a; };

template <int AUX, int M, bool CONTIG>
float run_sim(hipStream_t st, const Ring &ring, uint32_t n, int T, int chain, uint32_t lanes, int block) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    dim3 g(lanes / block), b(block);
    const size_t R = ring.a.size();
    const size_t bytes = (size_t)T * C * n * 4;
    int reps = (int)(20e9 / (double)bytes);
    if (reps < 8) reps = 8;
    for (int i = 0; i < reps / 4 + 2; ++i) k_sim<AUX, M, CONTIG><<<g, b, 0, st>>>(ring.a[i % R], n, n, T, chain, lanes);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) k_sim<AUX, M, CONTIG><<<g, b, 0, st>>>(ring.a[i % R], n, n, T, chain, lanes);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms * 1e3f / reps;
}
template <int M, bool CONTIG>
void row_sim(hipStream_t st, const Ring &ring, uint32_t n, int T, int chain, uint32_t lanes, int block) {
    if ((uint64_t)lanes * M > n || n % (lanes * M) != 0 || lanes % block != 0) return;
    const double bytes = 64.0 * n * T;
    const float d = run_sim<0, M, CONTIG>(st, ring, n, T, chain, lanes, block), w = run_sim<17, M, CONTIG>(st, ring, n, T, chain, lanes, block),
                s = run_sim<2, M, CONTIG>(st, ring, n, T, chain, lanes, block);
    printf("n=%8u T=%3d sim M=%d %s lanes=%6u (%4u WGs of %3d) chain=%3d | default %8.1f us %5.2f TB/s | sc0sc1 %8.1f us %5.2f | nt %8.1f us %5.2f\n",
           n, T, M, CONTIG ? "contig" : "apart ", lanes, lanes / block, block, chain, d, bytes / d / 1e6, w, bytes / w / 1e6, s, bytes / s / 1e6);
    fflush(stdout);
}

template <int AUX, int M>
float run_multi(hipStream_t st, const Ring &ring, uint32_t n, int T, int block) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    dim3 g((n / M + block - 1) / block), b(block);
    const size_t R = ring.a.size();
    const size_t bytes = (size_t)T * C * n * 4;
    int reps = (int)(30e9 / (double)bytes);
    if (reps < 8) reps = 8;
    for (int i = 0; i < reps / 4 + 2; ++i) k_store_multi<AUX, M><<<g, b, 0, st>>>(ring.a[i % R], n, n, T);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) k_store_multi<AUX, M><<<g, b, 0, st>>>(ring.a[i % R], n, n, T);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms * 1e3f / reps;
}
template <int M>
void row_multi(hipStream_t st, const Ring &ring, uint32_t n, int T, int block) {
    const double bytes = 64.0 * n * T;
    const float d = run_multi<0, M>(st, ring, n, T, block), w = run_multi<17, M>(st, ring, n, T, block), s = run_multi<2, M>(st, ring, n, T, block);
    printf("n=%8u T=%3d ring=%2zu multi=%d block=%3d width=1 | default %8.1f us %5.2f TB/s | sc0sc1 %8.1f us %5.2f | nt %8.1f us %5.2f\n",
           n, T, ring.a.size(), M, block, d, bytes / d / 1e6, w, bytes / w / 1e6, s, bytes / s / 1e6);
    fflush(stdout);
}

template <int AUX, int WIDTH>
float run_sliced(hipStream_t st, const Ring &ring, uint32_t n, uint32_t pitch, int T, uint32_t slice, int block = 256) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    dim3 g((slice + block - 1) / block), b(block);
    const size_t R = ring.a.size();
    const size_t bytes = (size_t)T * C * pitch * 4;
    int reps = (int)(30e9 / (double)bytes);
    if (reps < 8) reps = 8;
    for (int i = 0; i < reps / 4 + 2; ++i) k_store_sliced<AUX, WIDTH><<<g, b, 0, st>>>(ring.a[i % R], n, pitch, T, slice);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) k_store_sliced<AUX, WIDTH><<<g, b, 0, st>>>(ring.a[i % R], n, pitch, T, slice);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms * 1e3f / reps;
}

template <int WIDTH>
void row_sliced(hipStream_t st, const Ring &ring, uint32_t n, int T, uint32_t slice, int block = 256) {
    const double bytes = 64.0 * n * T;
    const float d = run_sliced<0, WIDTH>(st, ring, n, n, T, slice, block), w = run_sliced<17, WIDTH>(st, ring, n, n, T, slice, block),
                s = run_sliced<2, WIDTH>(st, ring, n, n, T, slice, block);
    printf("n=%8u T=%3d ring=%2zu sliced=%7u block=%3d width=%d | default %8.1f us %5.2f TB/s | sc0sc1 %8.1f us %5.2f | nt %8.1f us %5.2f\n",
           n, T, ring.a.size(), slice, block, WIDTH, d, bytes / d / 1e6, w, bytes / w / 1e6, s, bytes / s / 1e6);
    fflush(stdout);
}

template <int AUX, int WIDTH, int TILED>
float run(hipStream_t st, const Ring &ring, uint32_t n, uint32_t pitch, int T, int block) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    dim3 g((n + block - 1) / block), b(block);
    const size_t R = ring.a.size();
    const size_t bytes = (size_t)T * C * pitch * 4;
    int reps = (int)(30e9 / (double)bytes);   // ~30 GB of stores per measurement (~5 ms+)
    if (reps < 8) reps = 8;
    for (int i = 0; i < reps / 4 + 2; ++i) k_store<AUX, WIDTH, TILED><<<g, b, 0, st>>>(ring.a[i % R], n, pitch, T);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) k_store<AUX, WIDTH, TILED><<<g, b, 0, st>>>(ring.a[i % R], n, pitch, T);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms * 1e3f / reps;
}

template <int WIDTH, int TILED>
void row(hipStream_t st, const Ring &ring, uint32_t n, uint32_t pad, int T, int block) {
    const uint32_t pitch = n + pad;
    const double bytes = 64.0 * n * T;
    const float d = run<0, WIDTH, TILED>(st, ring, n, pitch, T, block), w = run<17, WIDTH, TILED>(st, ring, n, pitch, T, block),
                s = run<2, WIDTH, TILED>(st, ring, n, pitch, T, block), c1 = run<16, WIDTH, TILED>(st, ring, n, pitch, T, block);
    printf("n=%8u T=%3d ring=%2zu block=%3d width=%d %s pad=%6u | default %8.1f us %5.2f TB/s | sc0sc1 %8.1f us %5.2f | nt %8.1f us %5.2f | sc1 %8.1f us %5.2f\n",
           n, T, ring.a.size(), block, WIDTH, TILED ? "tiled " : "tmajor", pad, d, bytes / d / 1e6, w, bytes / w / 1e6, s, bytes / s / 1e6, c1, bytes / c1 / 1e6);
    fflush(stdout);
}

int main(int argc, char **argv) {
    const bool sliced_only = getenv("SLICED_ONLY") != nullptr;
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int T = 64;
    const uint32_t maxpad = 8192 + 64;
    std::vector<uint32_t> sizes = {65536, 131072, 262144, 524288, 1048576};
    if (argc > 1) { sizes.clear(); for (int i = 1; i < argc; ++i) sizes.push_back((uint32_t)atoi(argv[i])); }
    for (uint32_t n : sizes) {
        const size_t bytes = (size_t)T * C * (n + maxpad) * 4;
        size_t R = (size_t)(1.6e9 / (double)bytes) + 1;
        if (R < 2) R = 2;
        Ring ring;
        for (size_t r = 0; r < R; ++r) { float *p; CK(hipMalloc(&p, bytes)); ring.a.push_back(p); }
        Ring one; one.a.push_back(ring.a[0]);   // in-place rewrite of one arena (round 1's bench)
        if (n <= 131072 && !sliced_only) { row<1, 0>(st, one, n, 0, T, 256); row<4, 0>(st, one, n, 0, T, 256); }
        for (uint32_t pad : {0u, 32u, 64u, 256u, 1024u + 64u, 8192u + 64u}) {
            if (sliced_only && pad) continue;
            row<1, 0>(st, ring, n, pad, T, 256);
            row<4, 0>(st, ring, n, pad, T, 256);
        }
        if (!sliced_only) {
        row<1, 0>(st, ring, n, 0, T, 64);
        row<4, 0>(st, ring, n, 0, T, 64);
        row<1, 1>(st, ring, n, 0, T, 256);
        row<4, 1>(st, ring, n, 0, T, 256);
        }
        if (getenv("V4")) {   // grid-size scan of the sliced write-through kernel: which workgroup counts are fast?
            for (uint32_t wgs : {64u, 96u, 112u, 120u, 128u, 136u, 144u, 160u, 192u, 224u, 256u, 384u, 512u})
                row_sliced<1>(st, ring, n, T, wgs * 256u, 256);
            for (uint32_t wgs : {256u, 384u, 512u, 768u, 1024u})
                row_sliced<1>(st, ring, n, T, wgs * 64u, 64);
            for (uint32_t wgs : {128u, 192u, 256u, 512u})
                row_sliced<1>(st, ring, n, T, wgs * 128u, 128);
            for (uint32_t wgs : {32u, 48u, 64u, 96u, 128u})
                row_sliced<1>(st, ring, n, T, wgs * 512u, 512);
        } else if (getenv("SIM")) {
            for (int chain : {0, 230, 330}) {
                for (int block : {64, 256}) {
                    row_sim<1, true>(st, ring, n, T, chain, n, block);          // one env per lane, full grid (today's shape)
                    for (uint32_t lanes : {16384u, 32768u, 65536u}) {
                        row_sim<2, true>(st, ring, n, T, chain, lanes, block);
                        row_sim<2, false>(st, ring, n, T, chain, lanes, block);
                        row_sim<4, true>(st, ring, n, T, chain, lanes, block);
                        row_sim<4, false>(st, ring, n, T, chain, lanes, block);
                    }
                    row_sim<1, true>(st, ring, n, T, chain, 32768u, block);
                    row_sim<1, true>(st, ring, n, T, chain, 65536u, block);
                }
            }
        } else if (getenv("V3")) {
            for (uint32_t slice : {24576u, 32768u, 40960u, 49152u})
                for (int block : {64, 256, 1024})
                    if (slice <= n) row_sliced<1>(st, ring, n, T, slice, block);
            for (int block : {64, 256}) {
                row_multi<2>(st, ring, n, T, block);
                row_multi<4>(st, ring, n, T, block);
            }
        } else
        for (uint32_t slice : {16384u, 32768u, 65536u, 131072u, 262144u})
            if (slice <= n) { row_sliced<1>(st, ring, n, T, slice); row_sliced<4>(st, ring, n, T, slice); }
        // the runtime's fill, for scale
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const size_t fb = (size_t)T * C * n * 4;
        for (size_t r = 0; r < R; ++r) CK(hipMemsetAsync(ring.a[r], 0, fb, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        const int reps = 16;
        for (int i = 0; i < reps; ++i) CK(hipMemsetAsync(ring.a[i % R], 0, fb, st));
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("n=%8u hipMemsetAsync ring of %zu x %.1f MB: %.1f us (%.2f TB/s)\n", n, R, fb / 1e6, ms * 1e3 / reps, fb / (ms / reps * 1e-3) / 1e12);
        for (float *p : ring.a) CK(hipFree(p));
    }
    return 0;
}
