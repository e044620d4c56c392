// issue_rate.hip - what does one instruction cost a lone wavefront on a gfx950 SIMD?  (diagnostic, not part of the library)
//
// The in-kernel-policy rollout (k_rollout<*, ACT_POLICY_BF16>) runs ONE wavefront per SIMD at BASELINE's 65 536 envs and is
// bound by its own instruction stream.  This measures the issue cost of the instruction classes that stream is made of -
// plain / packed / transcendental VALU, bf16 conversion, packed f16, MFMA, AGPR moves, LDS reads - for 1 and 2 wavefronts per
// SIMD, independent and dependent, and of the activation sequences under discussion (tanh via exp + rcp, packed forms, f16
// polynomial), so that the kernel's "VALU-issue roofline" in bench.py is a measured number and not 4 cycles by assumption.
//
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/ir tools/micro/issue_rate.hip && /tmp/ir
// Output: ns per instruction and cycles at the shader clock measured by the calibration row (s_memtime is 100 MHz: the clock
// is derived from a dependent v_add_f32 chain assumed to issue every 4 cycles? NO - it is read from SMI by the wrapper script;
// this program prints ns and the ratio to an independent v_fma_f32 stream).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define R2(x) x x
#define R4(x) R2(x) R2(x)
#define R8(x) R4(x) R4(x)

// eight independent float registers
#define DECL8 float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
              float b = 1.0001f, c = 0.5f; asm volatile("" : "+v"(b), "+v"(c));
#define OUT8 "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
#define SINK8 if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 123.456f) out[threadIdx.x] = a0;
// eight independent 64-bit register pairs
#define DECL8P f32x2 p0 = {(float)threadIdx.x, 1.f}, p1 = p0 + 1.f, p2 = p0 + 2.f, p3 = p0 + 3.f, p4 = p0 + 4.f, p5 = p0 + 5.f, p6 = p0 + 6.f, p7 = p0 + 7.f; \
               f32x2 pb = {1.0001f, 0.9999f}, pc = {0.5f, 0.25f}; asm volatile("" : "+v"(pb), "+v"(pc));
#define OUT8P "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
#define SINK8P if (p0[0] + p1[1] + p2[0] + p3[1] + p4[0] + p5[1] + p6[0] + p7[1] == 123.456f) out[threadIdx.x] = p0[0];

// every kernel: `iters` trips of a body that holds INSTR instructions of the measured kind
#define KERNEL(name, DECLS, SINK, ...)                                        \
    __global__ __launch_bounds__(512) void name(float *out, int iters) {      \
        DECLS                                                                 \
        for (int i = 0; i < iters; ++i) { __VA_ARGS__ }                       \
        SINK                                                                  \
    }

// ---- plain VALU ---------------------------------------------------------------------------------------------------------
KERNEL(k_fma_ind, DECL8, SINK8, R8(asm volatile(
    "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
    "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
    : OUT8 : "v"(b), "v"(c));))
KERNEL(k_fma_dep, DECL8, SINK8, R8(asm volatile(
    "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n"
    "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n"
    : OUT8 : "v"(b), "v"(c));))
KERNEL(k_add_ind, DECL8, SINK8, R8(asm volatile(
    "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
    "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
    : OUT8 : "v"(b), "v"(c));))
// ---- transcendental -----------------------------------------------------------------------------------------------------
KERNEL(k_exp_ind, DECL8, SINK8, R8(asm volatile(
    "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
    "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
    : OUT8 : "v"(b), "v"(c));))
KERNEL(k_exp_dep, DECL8, SINK8, R8(asm volatile(
    "v_exp_f32 %0, %0\n v_exp_f32 %0, %0\n v_exp_f32 %0, %0\n v_exp_f32 %0, %0\n"
    "v_exp_f32 %0, %0\n v_exp_f32 %0, %0\n v_exp_f32 %0, %0\n v_exp_f32 %0, %0\n"
    : OUT8 : "v"(b), "v"(c));))
KERNEL(k_rcp_ind, DECL8, SINK8, R8(asm volatile(
    "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
    "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
    : OUT8 : "v"(b), "v"(c));))
// exp, fma alternating (does a plain VALU hide behind a transcendental?)
KERNEL(k_exp_fma_mix, DECL8, SINK8, R8(asm volatile(
    "v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %8, %9\n v_exp_f32 %2, %2\n v_fma_f32 %3, %3, %8, %9\n"
    "v_exp_f32 %4, %4\n v_fma_f32 %5, %5, %8, %9\n v_exp_f32 %6, %6\n v_fma_f32 %7, %7, %8, %9\n"
    : OUT8 : "v"(b), "v"(c));))
KERNEL(k_exp_f16_ind, DECL8, SINK8, R8(asm volatile(
    "v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3\n"
    "v_exp_f16 %4, %4\n v_exp_f16 %5, %5\n v_exp_f16 %6, %6\n v_exp_f16 %7, %7\n"
    : OUT8 : "v"(b), "v"(c));))
// ---- packed f32 ---------------------------------------------------------------------------------------------------------
KERNEL(k_pkfma_ind, DECL8P, SINK8P, R8(asm volatile(
    "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
    "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
    : OUT8P : "v"(pb), "v"(pc));))
KERNEL(k_pkfma_dep, DECL8P, SINK8P, R8(asm volatile(
    "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %0, %0, %8, %9\n"
    "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %0, %0, %8, %9\n"
    : OUT8P : "v"(pb), "v"(pc));))
KERNEL(k_pkadd_ind, DECL8P, SINK8P, R8(asm volatile(
    "v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
    "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
    : OUT8P : "v"(pb), "v"(pc));))
KERNEL(k_pkmul_ind, DECL8P, SINK8P, R8(asm volatile(
    "v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
    "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
    : OUT8P : "v"(pb), "v"(pc));))
// ---- conversions / packed f16 -------------------------------------------------------------------------------------------
KERNEL(k_cvtbf16_ind, DECL8, SINK8, R8(asm volatile(
    "v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %4\n"
    "v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %5, %5, %6\n v_cvt_pk_bf16_f32 %6, %6, %7\n v_cvt_pk_bf16_f32 %7, %7, %0\n"
    : OUT8 : "v"(b), "v"(c));))
KERNEL(k_cvtf16_ind, DECL8, SINK8, R8(asm volatile(
    "v_cvt_pkrtz_f16_f32 %0, %0, %1\n v_cvt_pkrtz_f16_f32 %1, %1, %2\n v_cvt_pkrtz_f16_f32 %2, %2, %3\n v_cvt_pkrtz_f16_f32 %3, %3, %4\n"
    "v_cvt_pkrtz_f16_f32 %4, %4, %5\n v_cvt_pkrtz_f16_f32 %5, %5, %6\n v_cvt_pkrtz_f16_f32 %6, %6, %7\n v_cvt_pkrtz_f16_f32 %7, %7, %0\n"
    : OUT8 : "v"(b), "v"(c));))
KERNEL(k_pkfma16_ind, DECL8, SINK8, R8(asm volatile(
    "v_pk_fma_f16 %0, %0, %8, %9\n v_pk_fma_f16 %1, %1, %8, %9\n v_pk_fma_f16 %2, %2, %8, %9\n v_pk_fma_f16 %3, %3, %8, %9\n"
    "v_pk_fma_f16 %4, %4, %8, %9\n v_pk_fma_f16 %5, %5, %8, %9\n v_pk_fma_f16 %6, %6, %8, %9\n v_pk_fma_f16 %7, %7, %8, %9\n"
    : OUT8 : "v"(b), "v"(c));))
KERNEL(k_pkfma16_dep, DECL8, SINK8, R8(asm volatile(
    "v_pk_fma_f16 %0, %0, %8, %9\n v_pk_fma_f16 %0, %0, %8, %9\n v_pk_fma_f16 %0, %0, %8, %9\n v_pk_fma_f16 %0, %0, %8, %9\n"
    "v_pk_fma_f16 %0, %0, %8, %9\n v_pk_fma_f16 %0, %0, %8, %9\n v_pk_fma_f16 %0, %0, %8, %9\n v_pk_fma_f16 %0, %0, %8, %9\n"
    : OUT8 : "v"(b), "v"(c));))
KERNEL(k_pkmin16_ind, DECL8, SINK8, R8(asm volatile(
    "v_pk_min_f16 %0, %0, %8\n v_pk_max_f16 %1, %1, %8\n v_pk_min_f16 %2, %2, %8\n v_pk_max_f16 %3, %3, %8\n"
    "v_pk_min_f16 %4, %4, %8\n v_pk_max_f16 %5, %5, %8\n v_pk_min_f16 %6, %6, %8\n v_pk_max_f16 %7, %7, %8\n"
    : OUT8 : "v"(b), "v"(c));))
// ---- AGPR moves, LDS ----------------------------------------------------------------------------------------------------
KERNEL(k_accmov, DECL8, SINK8, R8(asm volatile(
    "v_accvgpr_write_b32 a0, %0\n v_accvgpr_read_b32 %1, a1\n v_accvgpr_write_b32 a2, %2\n v_accvgpr_read_b32 %3, a3\n"
    "v_accvgpr_write_b32 a4, %4\n v_accvgpr_read_b32 %5, a5\n v_accvgpr_write_b32 a6, %6\n v_accvgpr_read_b32 %7, a7\n"
    : OUT8 : "v"(b), "v"(c) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");))
KERNEL(k_mov_ind, DECL8, SINK8, R8(asm volatile(
    "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
    "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %8\n"
    : OUT8 : "v"(b), "v"(c));))

__global__ __launch_bounds__(512) void k_dsread128(float *out, int iters) {
    __shared__ f32x4 buf[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) buf[i] = (f32x4){1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    uint32_t addr = (threadIdx.x & 63u) * 16u;
    f32x4 v0, v1, v2, v3, v4, v5, v6, v7;
    for (int i = 0; i < iters; ++i) {
        R8(asm volatile(
            "ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:1024\n ds_read_b128 %2, %8 offset:2048\n ds_read_b128 %3, %8 offset:3072\n"
            "ds_read_b128 %4, %8 offset:4096\n ds_read_b128 %5, %8 offset:5120\n ds_read_b128 %6, %8 offset:6144\n ds_read_b128 %7, %8 offset:7168\n"
            "s_waitcnt lgkmcnt(0)\n"
            : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(addr));)
    }
    if (v0[0] + v1[1] + v2[2] + v3[3] + v4[0] + v5[1] + v6[2] + v7[3] == 123.456f) out[threadIdx.x] = v0[0];
}

// ---- MFMA ---------------------------------------------------------------------------------------------------------------
#define DECLM f32x16 m0 = {}, m1 = {}, m2 = {}, m3 = {}; f32x4 fa = {1.f, 2.f, 3.f, 4.f}, fb = {.5f, .25f, .125f, 1.f}; \
              m0[0] = threadIdx.x; asm volatile("" : "+v"(fa), "+v"(fb), "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3)); DECL8
#define SINKM if (m0[0] + m1[1] + m2[2] + m3[3] == 123.456f) out[threadIdx.x] = m0[5]; SINK8
#define OUTM "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3)
KERNEL(k_mfma_ind, DECLM, SINKM, R8(asm volatile(
    "v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n v_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n"
    "v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n v_mfma_f32_32x32x16_bf16 %3, %4, %5, %3\n"
    : OUTM : "v"(fa), "v"(fb));))
KERNEL(k_mfma_dep, DECLM, SINKM, R8(asm volatile(
    "v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n"
    "v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n"
    : OUTM : "v"(fa), "v"(fb));))
// one MFMA + NV independent plain VALU: how many VALU hide in the MFMA's shadow for a lone wavefront?
#define MFMA_PLUS(name, VALU)                                                                      \
    KERNEL(name, DECLM, SINKM, R8(asm volatile(                                                    \
        "v_mfma_f32_32x32x16_bf16 %0, %12, %13, %0\n" VALU                                         \
        "v_mfma_f32_32x32x16_bf16 %1, %12, %13, %1\n" VALU                                         \
        : OUTM, OUT8 : "v"(fa), "v"(fb), "v"(b), "v"(c));))
#define V4 "v_fma_f32 %4, %4, %14, %15\n v_fma_f32 %5, %5, %14, %15\n v_fma_f32 %6, %6, %14, %15\n v_fma_f32 %7, %7, %14, %15\n"
#define V8 V4 "v_fma_f32 %8, %8, %14, %15\n v_fma_f32 %9, %9, %14, %15\n v_fma_f32 %10, %10, %14, %15\n v_fma_f32 %11, %11, %14, %15\n"
#define E4 "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
#define E8 E4 "v_exp_f32 %8, %8\n v_exp_f32 %9, %9\n v_exp_f32 %10, %10\n v_exp_f32 %11, %11\n"
MFMA_PLUS(k_mfma_v4, V4)
MFMA_PLUS(k_mfma_v8, V8)
MFMA_PLUS(k_mfma_v16, V8 V8)
MFMA_PLUS(k_mfma_e4, E4)
MFMA_PLUS(k_mfma_e8, E8)

// ---- activation sequences on 16 independent values per trip (C++ with builtins, as the product code writes them; check the ISA
// with  hipcc -S --cuda-device-only  that the intended instructions came out) ------------------------------------------
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 h2;
// FORM 0: the shipped form: exp, exp, pk_add, rcp, rcp, pk_fma, cvt_pk per value pair (3.5 instructions per value)
// FORM 1: no packed f32: exp, add, rcp, fma per value + cvt_pk per pair (4.5 per value)
// FORM 2: packed-f16 odd polynomial: cvt_pkrtz, pk_min, pk_max, pk_mul, 5 pk_fma, pk_mul per pair (5 per value, no transcendental)
// FORM 3: f16 transcendental: cvt_pkrtz, 2 v_exp_f16 (sdwa), pk_add, 2 v_rcp_f16, pk_fma per pair
template <int FORM>
__global__ __launch_bounds__(512) void k_tanh(float *out, int iters) {
    f32x2 p[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) p[j] = (f32x2){0.01f * threadIdx.x + j, -0.02f * threadIdx.x - j};
    float A = -2.0f, B = 1.0f;
    asm volatile("" : "+v"(A), "+v"(B));
    h2 lim = {(_Float16)4.0f, (_Float16)4.0f}, c5 = {(_Float16)0.01f, (_Float16)0.01f}, c4 = c5, c3 = c5, c2 = c5, c1 = c5, c0 = {(_Float16)1.0f, (_Float16)1.0f};
    asm volatile("" : "+v"(lim), "+v"(c5), "+v"(c4), "+v"(c3), "+v"(c2), "+v"(c1), "+v"(c0));
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if constexpr (FORM == 0) {
                f32x2 e;
                e[0] = __builtin_amdgcn_exp2f(p[j][0]);
                e[1] = __builtin_amdgcn_exp2f(p[j][1]);
                e = e + 1.0f;
                f32x2 rc;
                rc[0] = __builtin_amdgcn_rcpf(e[0]);
                rc[1] = __builtin_amdgcn_rcpf(e[1]);
                const f32x2 v = __builtin_elementwise_fma(rc, (f32x2){A, A}, (f32x2){B, B});
                const uint32_t pk = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
                asm volatile("" ::"v"(pk));
                p[j] = v;
            } else if constexpr (FORM == 1) {
                float e0 = __builtin_amdgcn_exp2f(p[j][0]), e1 = __builtin_amdgcn_exp2f(p[j][1]);
                e0 += 1.0f;
                asm volatile("" : "+v"(e0));
                e1 += 1.0f;
                asm volatile("" : "+v"(e1));
                float v0 = __builtin_fmaf(__builtin_amdgcn_rcpf(e0), A, B);
                asm volatile("" : "+v"(v0));
                float v1 = __builtin_fmaf(__builtin_amdgcn_rcpf(e1), A, B);
                asm volatile("" : "+v"(v1));
                const f32x2 v = {v0, v1};
                const uint32_t pk = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
                asm volatile("" ::"v"(pk));
                p[j] = v;
            } else if constexpr (FORM == 2) {
                h2 x = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(p[j][0], p[j][1]));
                x = __builtin_elementwise_min(x, lim);
                x = __builtin_elementwise_max(x, -lim);
                const h2 x2 = x * x;
                h2 q = __builtin_elementwise_fma(x2, c5, c4);
                q = __builtin_elementwise_fma(x2, q, c3);
                q = __builtin_elementwise_fma(x2, q, c2);
                q = __builtin_elementwise_fma(x2, q, c1);
                q = __builtin_elementwise_fma(x2, q, c0);
                const h2 y = q * x;
                const uint32_t pk = __builtin_bit_cast(uint32_t, y);
                asm volatile("" ::"v"(pk));
                p[j][0] += __builtin_bit_cast(float, pk);   // feedback costs one plain VALU per pair (subtract 0.5 per value)
            } else {
                h2 x = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(p[j][0], p[j][1]));
                h2 e;
                e[0] = __builtin_exp2f16(x[0]);
                e[1] = __builtin_exp2f16(x[1]);
                e = e + c0;
                h2 rc;
                rc[0] = __builtin_amdgcn_rcph(e[0]);
                rc[1] = __builtin_amdgcn_rcph(e[1]);
                const h2 y = __builtin_elementwise_fma(rc, c1, c0);
                const uint32_t pk = __builtin_bit_cast(uint32_t, y);
                asm volatile("" ::"v"(pk));
                p[j][0] += __builtin_bit_cast(float, pk);
            }
        }
    }
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += p[j][0] + p[j][1];
    if (acc == 123.456f) out[threadIdx.x] = acc;
}

struct Row { const char *name; void (*fn)(float *, int); int per_trip; const char *note; };

static float run(void (*fn)(float *, int), float *out, int blocks, int threads, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(threads), 0, 0, out, iters / 8);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(threads), 0, 0, out, iters);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main(int argc, char **argv) {
    const double ghz = argc > 1 ? atof(argv[1]) : 2.4;   // shader clock to convert ns into cycles (argument: measured by the wrapper)
    float *out; CK(hipMalloc(&out, 4096));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clockRate %d kHz; cycles below assume %.2f GHz\n", prop.name, cus, prop.clockRate, ghz);
    const Row rows[] = {
        {"v_fma_f32 independent", k_fma_ind, 64, ""},
        {"v_fma_f32 dependent chain", k_fma_dep, 64, ""},
        {"v_add_f32 independent", k_add_ind, 64, ""},
        {"v_mov_b32 independent", k_mov_ind, 64, ""},
        {"v_exp_f32 independent", k_exp_ind, 64, ""},
        {"v_exp_f32 dependent chain", k_exp_dep, 64, ""},
        {"v_rcp_f32 independent", k_rcp_ind, 64, ""},
        {"v_exp_f32 / v_fma_f32 alternating", k_exp_fma_mix, 64, ""},
        {"v_exp_f16 independent", k_exp_f16_ind, 64, ""},
        {"v_pk_fma_f32 independent", k_pkfma_ind, 64, "2 values per instruction"},
        {"v_pk_fma_f32 dependent chain", k_pkfma_dep, 64, ""},
        {"v_pk_add_f32 independent", k_pkadd_ind, 64, ""},
        {"v_pk_mul_f32 independent", k_pkmul_ind, 64, ""},
        {"v_cvt_pk_bf16_f32", k_cvtbf16_ind, 64, ""},
        {"v_cvt_pkrtz_f16_f32", k_cvtf16_ind, 64, ""},
        {"v_pk_fma_f16 independent", k_pkfma16_ind, 64, "2 values per instruction"},
        {"v_pk_fma_f16 dependent chain", k_pkfma16_dep, 64, ""},
        {"v_pk_min/max_f16", k_pkmin16_ind, 64, ""},
        {"v_accvgpr_write/read", k_accmov, 64, ""},
        {"ds_read_b128 (8 in flight, then wait)", k_dsread128, 64, ""},
        {"v_mfma_f32_32x32x16_bf16 independent", k_mfma_ind, 32, ""},
        {"v_mfma_f32_32x32x16_bf16 dependent", k_mfma_dep, 32, ""},
        {"1 MFMA + 4 v_fma (per MFMA group)", k_mfma_v4, 16, "cost of the whole group"},
        {"1 MFMA + 8 v_fma (per group)", k_mfma_v8, 16, ""},
        {"1 MFMA + 16 v_fma (per group)", k_mfma_v16, 16, ""},
        {"1 MFMA + 4 v_exp (per group)", k_mfma_e4, 16, ""},
        {"1 MFMA + 8 v_exp (per group)", k_mfma_e8, 16, ""},
        {"tanh: exp,pk_add,rcp,pk_fma,cvt (per VALUE)", k_tanh<0>, 16, "3.5 instructions per value"},
        {"tanh: exp,add,rcp,fma,cvt unpacked (per VALUE)", k_tanh<1>, 16, "4.5 per value"},
        {"tanh: f16 packed polynomial (per VALUE)", k_tanh<2>, 16, "5 per value + 0.5 feedback"},
        {"tanh: f16 exp/rcp (per VALUE)", k_tanh<3>, 16, ""},
    };
    printf("%-50s %12s %12s %12s %12s\n", "instruction stream", "1w/SIMD ns", "cycles", "2w/SIMD ns", "cycles(pair)");
    for (const Row &r : rows) {
        const int iters = 4000;
        const double n = (double)iters * r.per_trip;
        const float t1 = run(r.fn, out, cus, 256, iters), t2 = run(r.fn, out, cus, 512, iters);
        const double ns1 = t1 * 1e6 / n, ns2 = t2 * 1e6 / n;
        printf("%-50s %12.3f %12.2f %12.3f %12.2f   %s\n", r.name, ns1, ns1 * ghz, ns2, ns2 * ghz, r.note);
    }
    return 0;
}
