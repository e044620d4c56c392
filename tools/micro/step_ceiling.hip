// step_ceiling.hip - the practical floor of a one-env-step-per-launch kernel on this GPU: a kernel with k_step's memory shape and NO arithmetic
// (diagnostic, not part of the library).  Per env: reads 10 state + 4 action dwords (+ `extra_rd` bookkeeping dwords), writes 10 state dwords,
// 1 reward dword, 1 done byte (+ `extra_wr` dwords); feature-major arrays, one env per lane, state rewritten in place, actions from a ring.
// Reports us per launch and the fraction of 8 TB/s on the 101 algorithmic bytes, like bench.py's `legs.step*`.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/_build/step_ceiling tools/micro/step_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t rsrc(const void *p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, -1, 0x00020000); }

template <int RD, int WR>
__global__ __launch_bounds__(256) void k_copy(float *state, const float *act, float *extra, float *rew, uint8_t *done, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t off = i * 4u, col = n * 4u;
    const rsrc_t rs = rsrc(state), ra = rsrc(act), rx = rsrc(extra);
    float s[10], a[4], x[RD > 0 ? RD : 1];
#pragma unroll
    for (int c = 0; c < 10; ++c) s[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, c * col, 0));
#pragma unroll
    for (int c = 0; c < 4; ++c) a[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, off, c * col, 0));
#pragma unroll
    for (int c = 0; c < RD; ++c) x[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off, c * col, 0));
    float r = a[0] + a[1] + a[2] + a[3];
#pragma unroll
    for (int c = 0; c < RD; ++c) r += x[c];
#pragma unroll
    for (int c = 0; c < 10; ++c) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, s[c] + r), rs, off, c * col, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, r), rsrc(rew), off, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(r > 1e30f), rsrc(done), i, 0, 0);
#pragma unroll
    for (int c = 0; c < WR; ++c) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, r + c), rx, off, c * col, 0);
}

template <int RD, int WR> void run(uint32_t n, int block) {
    const int RING = (int)(n >= 1048576u ? 16 : 64);
    float *state, *act, *extra, *rew; uint8_t *done;
    CK(hipMalloc(&state, 40ull * n)); CK(hipMalloc(&act, 16ull * n * RING)); CK(hipMalloc(&extra, 16ull * n + 64)); CK(hipMalloc(&rew, 4ull * n * RING)); CK(hipMalloc(&done, (size_t)n * RING));
    CK(hipMemset(state, 0, 40ull * n)); CK(hipMemset(act, 0, 16ull * n * RING)); CK(hipMemset(extra, 0, 16ull * n));
    const int reps = (int)(4000ull * 65536 / n) + 300;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto go = [&](int k) { hipLaunchKernelGGL((k_copy<RD, WR>), dim3((n + block - 1) / block), dim3(block), 0, 0, state, act + (size_t)(k % RING) * 4 * n, extra, rew + (size_t)(k % RING) * n, done + (size_t)(k % RING) * n, n); };
    for (int k = 0; k < 200; ++k) go(k);
    CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
    for (int k = 0; k < reps; ++k) go(k);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, phys = (double)n * (101 + 4 * (RD + WR));
    printf("| %u | %d | +%d rd / +%d wr dwords | %.2f | %.3f | %.2f |\n", n, block, RD, WR, us, 101.0 * n / us / 1e6 / 8.0, phys / us / 1e6);
    fflush(stdout);
    CK(hipFree(state)); CK(hipFree(act)); CK(hipFree(extra)); CK(hipFree(rew)); CK(hipFree(done));
}
int main() {
    printf("| envs | threads per workgroup | bookkeeping | us per launch | frac of 8 TB/s on 101 B | physical TB/s |\n|---|---|---|---|---|---|\n");
    for (uint32_t n : {65536u, 262144u, 1048576u, 4194304u})
        for (int block : {128, 256}) {
            run<0, 0>(n, block);     // exactly the 101 algorithmic bytes
            run<1, 1>(n, block);     // + running return in / out (what k_step's default moves beyond 786 432 envs)
            run<4, 1>(n, block);     // + steps_beyond_done, reset counter, episode start read eagerly (k_step below 786 432 envs)
        }
    return 0;
}
