// pk_hazard.hip - stand-alone reproducer of the stale read behind packed fp32 arithmetic on a SIMD that also executes MFMAs
// (gfx950, ROCm 7.2; profiles/r05/packed_f32_hazard.md).  Diagnostic, not part of the library.  Hand-written instructions only.
//
// "Victim" wavefronts execute one of the sequences below on fresh pseudo-random operands every iteration and compare the 64-bit
// result, bit for bit and per lane, with the same arithmetic done by scalar v_mul / v_fma (built with -fno-slp-vectorize):
//   V0  the triple the SLP vectoriser made of quat_body_z() in the policy rollout kernels:
//         v_pk_mul_f32 D, Q, D op_sel_hi:[0,1] ; v_pk_fma_f32 D, P, Q, D op_sel:[0,1,0] ; [s_nop K] ; v_pk_add_f32 D, D, D
//   V1  the same triple without any op_sel           V2  v_pk_fma_f32 D, P, Q, D op_sel:[0,1,0] alone
//   V3  v_pk_mul_f32 D, Q, D op_sel_hi:[0,1] alone   V4  v_pk_fma_f32 D, P, Q, D alone (no op_sel)
//   V5  two scalar v_fma_f32 (control)               V6  v_pk_mov_b32 D, P, Q op_sel:[1,0]
//   V7  v_pk_mul_f32 D, P, Q op_sel:[0,1]            V8  v_pk_add_f32 D, P, Q op_sel:[0,1]
//   V9  v_pk_fma_f32 D, P, Q, D op_sel:[1,0,0]       V10 v_pk_fma_f32 D, P, Q, D op_sel:[0,0,1]
//   V11 v_pk_fma_f32 D, P, Q, D op_sel_hi:[1,0,1]    (hi result from the LOW half of src1: the mirror image of V2)
// while other wavefronts of the same SIMDs run an interferer: nothing, back-to-back v_mfma_f32_32x32x16_f16, v_exp / v_rcp, or
// packed fp32.  MIXED = every wavefront does a burst of MFMAs and then the sequence, like the real kernels.
// Output per run: wrong results by quarter-wavefront, and one sample (operands, got, expected) of the first mismatch seen.
//
// build + run:  hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o pk_hazard tools/micro/pk_hazard.hip && ./pk_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned long long u64;   // 64-bit register pairs travel as integers (f32x2 asm operands: hipcc compared the wrong half)
enum { I_NONE = 0, I_MFMA = 1, I_TRANS = 2, I_PK = 3, I_MIXED = 4 };

__device__ __forceinline__ float u2f(unsigned x) { return (float)(x >> 8) * (1.0f / 8388608.0f) - 1.0f; }   // [-1, 1)
__device__ __forceinline__ unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; return x ^ (x >> 16); }
__device__ __forceinline__ u64 pack(float lo, float hi) { return (u64)__builtin_bit_cast(unsigned, lo) | ((u64)__builtin_bit_cast(unsigned, hi) << 32); }

template <int V, int K> __device__ __forceinline__ u64 victim_seq(u64 d, u64 p, u64 q) {
#define NOP "s_nop %3\n\t"
    if constexpr (V == 0) asm volatile("v_pk_mul_f32 %0, %2, %0 op_sel_hi:[0,1]\n\tv_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]\n\t" NOP "v_pk_add_f32 %0, %0, %0" : "+v"(d) : "v"(p), "v"(q), "n"(K));
    if constexpr (V == 1) asm volatile("v_pk_mul_f32 %0, %2, %0\n\tv_pk_fma_f32 %0, %1, %2, %0\n\t" NOP "v_pk_add_f32 %0, %0, %0" : "+v"(d) : "v"(p), "v"(q), "n"(K));
    if constexpr (V == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]\n\t" NOP : "+v"(d) : "v"(p), "v"(q), "n"(K));
    if constexpr (V == 3) asm volatile("v_pk_mul_f32 %0, %2, %0 op_sel_hi:[0,1]\n\t" NOP : "+v"(d) : "v"(p), "v"(q), "n"(K));
    if constexpr (V == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %0\n\t" NOP : "+v"(d) : "v"(p), "v"(q), "n"(K));
    if constexpr (V == 6) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]\n\t" NOP : "+v"(d) : "v"(p), "v"(q), "n"(K));
    if constexpr (V == 7) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]\n\t" NOP : "+v"(d) : "v"(p), "v"(q), "n"(K));
    if constexpr (V == 8) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]\n\t" NOP : "+v"(d) : "v"(p), "v"(q), "n"(K));
    if constexpr (V == 9) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]\n\t" NOP : "+v"(d) : "v"(p), "v"(q), "n"(K));
    if constexpr (V == 10) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,1]\n\t" NOP : "+v"(d) : "v"(p), "v"(q), "n"(K));
    if constexpr (V == 11) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]\n\t" NOP : "+v"(d) : "v"(p), "v"(q), "n"(K));
#undef NOP
    return d;
}
template <int V> __device__ __forceinline__ u64 expected(const float *d, const float *p, const float *q) {
    if (V == 0) return pack(2.0f * __builtin_fmaf(p[0], q[1], q[0] * d[0]), 2.0f * __builtin_fmaf(p[1], q[1], q[0] * d[1]));
    if (V == 1) return pack(2.0f * __builtin_fmaf(p[0], q[0], q[0] * d[0]), 2.0f * __builtin_fmaf(p[1], q[1], q[1] * d[1]));
    if (V == 2) return pack(__builtin_fmaf(p[0], q[1], d[0]), __builtin_fmaf(p[1], q[1], d[1]));
    if (V == 3) return pack(q[0] * d[0], q[0] * d[1]);
    if (V == 6) return pack(p[1], q[0]);
    if (V == 7) return pack(p[0] * q[1], p[1] * q[1]);
    if (V == 8) return pack(p[0] + q[1], p[1] + q[1]);
    if (V == 9) return pack(__builtin_fmaf(p[1], q[0], d[0]), __builtin_fmaf(p[1], q[1], d[1]));
    if (V == 10) return pack(__builtin_fmaf(p[0], q[0], d[1]), __builtin_fmaf(p[1], q[1], d[1]));
    if (V == 11) return pack(__builtin_fmaf(p[0], q[0], d[0]), __builtin_fmaf(p[1], q[0], d[1]));
    return pack(__builtin_fmaf(p[0], q[0], d[0]), __builtin_fmaf(p[1], q[1], d[1]));   // V4, V5
}
// interferer: `n & 255` back-to-back MFMAs of kind n >> 8: 0 = f32_32x32x16_f16, 1 = f32_32x32x16_bf16, 2 = f32_16x16x32_f16 (4 passes), 3 = f32_32x32x2_f32
__device__ __forceinline__ void mfma_burst(f32x16 &acc, f16x8 a, f16x8 b, int n) {
    const int kind = n >> 8;
    for (int i = 0; i < (n & 255); ++i) {
        if (kind == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
        if (kind == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
        if (kind == 2) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(*(f32x4 *)&acc) : "v"(a), "v"(b));
        if (kind == 3) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(((float *)&a)[0]), "v"(((float *)&b)[0]));
    }
}

// 512 threads = 8 wavefronts per workgroup, two per SIMD.  Separate roles: wavefronts 0..3 are victims, 4..7 interferers.
template <int V, int K> __global__ __launch_bounds__(512) void k_probe(int mode, int iters, int burst, unsigned *bad_lane, u64 *checked,
                                                                       unsigned *sample, float *sink) {
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned h = mix(blockIdx.x * 512u + threadIdx.x + 1u), nbad = 0;
    f32x16 acc = {};
    f16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)u2f(mix(h + i)); fb[i] = (_Float16)u2f(mix(h + 8 + i)); }
    const bool victim = mode == I_MIXED || wave < 4 || mode == I_NONE;
    float tr = 1.0f + u2f(h) * 0.1f;
    u64 pk = pack(tr, -tr);
    for (int it = 0; it < iters; ++it) {
        if (victim) {
            if (mode == I_MIXED) mfma_burst(acc, fa, fb, burst);
            h = mix(h);
            const float d[2] = {u2f(h), u2f(mix(h + 1))}, p[2] = {u2f(mix(h + 2)), u2f(mix(h + 3))}, q[2] = {u2f(mix(h + 4)), u2f(mix(h + 5))};
            u64 got;
            if constexpr (V == 5) got = pack(__builtin_fmaf(p[0], q[0], d[0]), __builtin_fmaf(p[1], q[1], d[1]));
            else got = victim_seq<V, K>(pack(d[0], d[1]), pack(p[0], p[1]), pack(q[0], q[1]));
            const u64 want = expected<V>(d, p, q);
            if (got != want) {
                ++nbad;
                if (atomicCAS(&sample[0], 0u, 1u) == 0u) {   // first mismatch of the launch: keep the evidence
                    const u64 v[5] = {pack(d[0], d[1]), pack(p[0], p[1]), pack(q[0], q[1]), got, want};
                    for (int i = 0; i < 5; ++i) { sample[2 + 2 * i] = (unsigned)v[i]; sample[3 + 2 * i] = (unsigned)(v[i] >> 32); }
                    sample[1] = lane;
                }
            }
        } else if (mode == I_MFMA) {
            mfma_burst(acc, fa, fb, burst);
        } else if (mode == I_TRANS) {
            for (int i = 0; i < burst; ++i) asm volatile("v_exp_f32 %0, %0\n\tv_rcp_f32 %0, %0" : "+v"(tr));
        } else if (mode == I_PK) {
            for (int i = 0; i < burst; ++i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0 op_sel:[0,1,0]" : "+v"(pk));
        }
    }
    if (nbad) atomicAdd(&bad_lane[lane], nbad);
    if (victim && lane == 0) atomicAdd(checked, (u64)iters * 64ull);
    if (acc[0] + acc[5] + tr + (float)pk == 123.456f) sink[threadIdx.x] = acc[3];   // keep the interferers' results alive
}

template <int V, int K> void run(const char *name, int mode, int burst, int blocks, int iters) {
    unsigned *bad, *smp; u64 *chk; float *sink;
    CK(hipMalloc(&bad, 64 * 4)); CK(hipMalloc(&chk, 8)); CK(hipMalloc(&sink, 512 * 4)); CK(hipMalloc(&smp, 12 * 4));
    CK(hipMemset(bad, 0, 64 * 4)); CK(hipMemset(chk, 0, 8)); CK(hipMemset(smp, 0, 12 * 4));
    hipLaunchKernelGGL((k_probe<V, K>), dim3(blocks), dim3(512), 0, 0, mode, iters, burst, bad, chk, smp, sink);
    CK(hipDeviceSynchronize());
    unsigned hb[64], hs[12]; u64 hc;
    CK(hipMemcpy(hb, bad, sizeof(hb), hipMemcpyDeviceToHost)); CK(hipMemcpy(&hc, chk, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hs, smp, sizeof(hs), hipMemcpyDeviceToHost));
    u64 tot = 0, q[4] = {0, 0, 0, 0};
    for (int l = 0; l < 64; ++l) { tot += hb[l]; q[l / 16] += hb[l]; }
    printf("V%-2d %-6s mfma-kind %d burst %2d s_nop %d blocks %4d checked %.2e wrong %10llu (quarters %llu %llu %llu %llu)", V, name, burst >> 8, burst & 255, K, blocks, (double)hc, tot, q[0], q[1], q[2], q[3]);
    if (hs[0]) {
        auto f = [&](int i) { return (double)__builtin_bit_cast(float, hs[i]); };
        printf("  e.g. lane %u d=(%.6g, %.6g) p=(%.6g, %.6g) q=(%.6g, %.6g) got=(%.6g, %.6g) want=(%.6g, %.6g)", hs[1], f(2), f(3), f(4), f(5), f(6), f(7), f(8), f(9), f(10), f(11));
    }
    printf("\n");
    CK(hipFree(bad)); CK(hipFree(chk)); CK(hipFree(sink)); CK(hipFree(smp));
}
template <int V> void variant(int iters) {
    const char *names[] = {"none", "mfma", "trans", "pk", "mixed"};
    for (int mode : {I_NONE, I_PK, I_MFMA, I_MIXED})
        for (int burst : {4, 16}) {
            if (mode == I_NONE && burst != 4) continue;
            run<V, 0>(names[mode], mode, burst, 1024, iters);
            if (V <= 1 && mode >= I_MFMA) run<V, 7>(names[mode], mode, burst, 1024, iters);
        }
    if (V == 2)   // which matrix instructions on the SIMD do it
        for (int kind : {1, 2, 3}) run<V, 0>("mfma", I_MFMA, (kind << 8) | 16, 1024, iters);
}
int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 100000;
    variant<0>(iters); variant<1>(iters); variant<2>(iters); variant<3>(iters); variant<4>(iters); variant<5>(iters); variant<6>(iters);
    variant<7>(iters); variant<8>(iters); variant<9>(iters); variant<10>(iters); variant<11>(iters);
    return 0;
}
