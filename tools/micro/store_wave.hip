// store_wave.hip - would ONE store wavefront per workgroup (draining the hand-over tiles of all G pairs with 1 KiB-wide stores) give the
// fused rollout's trajectory stream a better rate than one memory wavefront per 64 envs?  (diagnostic, not part of the library)
//
// Trajectory shape of k_rollout: per env-step 16 dword rows (4 action + 10 obs + reward + one standing in for the done bytes),
// time-major [T][16][N], T = 64, every launch into a fresh arena of a ring > 1.5 GB, write-through stores (sc0 sc1) like the kernel.
//   pairs : today's writers - one wavefront per 64 envs, 16 dword stores (256 B per wavefront) per step
//   cu<G> : one store wavefront per G x 64 envs - per step and row G / 4 dwordx4 stores of 1 KiB contiguous each
// `work` emulates the time the other wavefronts need per step (0 = pure store rate): a chain of dependent v_fma per step.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/_build/store_wave tools/micro/store_wave.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t rsrc(const void *p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, -1, 0x00020000); }
constexpr int C = 16, AUX = 17;   // sc0 sc1
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float spin(float v, int work) {
    for (int i = 0; i < work; ++i) v = __builtin_fmaf(v, 1.000001f, 0.5f);
    return v;
}
// one wavefront per 64 envs
__global__ __launch_bounds__(256) void k_pairs(float *arena, uint32_t n, int T, int work) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = (float)i;
    const uint32_t col = n * 4u;
    float *base = arena;
    for (int t = 0; t < T; ++t) {
        v = spin(v, work);
        const rsrc_t r = rsrc(base);
#pragma unroll
        for (int c = 0; c < C; ++c) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), r, i * 4u, c * col, AUX);
        base += (size_t)C * n;
    }
}
// one store wavefront per G x 64 envs (G a multiple of 4): lane l stores envs [first + 4 l + 256 j, + 4) of every row, j < G / 4
template <int G> __global__ __launch_bounds__(64) void k_cu(float *arena, uint32_t n, int T, int work) {
    const uint32_t first = blockIdx.x * (64u * G), lane = threadIdx.x;
    if (first >= n) return;
    float v = (float)(first + lane);
    const uint32_t col = n * 4u;
    float *base = arena;
    for (int t = 0; t < T; ++t) {
        v = spin(v, work);
        const rsrc_t r = rsrc(base);
        const uint32_t b = __builtin_bit_cast(uint32_t, v);
        const u32x4 q = {b, b + 1u, b + 2u, b + 3u};
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int j = 0; j < G / 4; ++j) __builtin_amdgcn_raw_buffer_store_b128(q, r, (first + 256u * j + 4u * lane) * 4u, c * col, AUX);
        base += (size_t)C * n;
    }
}

// one wavefront per 64 envs, but the G = 4 wavefronts of a 256-thread workgroup share the rows: wavefront m stores rows m, m + 4, ... for all
// 256 envs of the workgroup with 1 KiB-wide dwordx4 stores (what a cooperative drain of the four pairs' hand-over tiles would issue)
__global__ __launch_bounds__(256) void k_coop(float *arena, uint32_t n, int T, int work) {
    const uint32_t first = blockIdx.x * 256u, lane = threadIdx.x & 63u, m = threadIdx.x >> 6;
    if (first >= n) return;
    float v = (float)(first + threadIdx.x);
    const uint32_t col = n * 4u;
    float *base = arena;
    for (int t = 0; t < T; ++t) {
        v = spin(v, work);
        const rsrc_t r = rsrc(base);
        const uint32_t b = __builtin_bit_cast(uint32_t, v);
        const u32x4 q = {b, b + 1u, b + 2u, b + 3u};
#pragma unroll
        for (int c = 0; c < C / 4; ++c) __builtin_amdgcn_raw_buffer_store_b128(q, r, (first + 4u * lane) * 4u, (4u * c + m) * col, AUX);
        base += (size_t)C * n;
    }
}

template <typename F> double time_us(F launch, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < reps / 4 + 2; ++i) launch(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch(reps + i);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / reps;
}

int main() {
    const int T = 64;
    printf("| envs | work per step | writers | us per launch | TB/s |\n|---|---|---|---|---|\n");
    for (int rep = 0; rep < 2; ++rep) for (uint32_t n : {65536u, 131072u, 262144u}) {
        const size_t per = (size_t)T * C * n * 4;
        const int R = (int)((size_t(1600) << 20) / per) + 2;
        std::vector<float *> ring(R);
        for (auto &p : ring) { CK(hipMalloc(&p, per)); CK(hipMemset(p, 0, per)); }
        const int reps = (int)(65536ull * 300 / n) + 20;
        for (int work : {0}) {
            auto row = [&](const char *name, double us) { printf("| %u | %d | %s | %.1f | %.2f |\n", n, work, name, us, (double)per / us / 1e6); fflush(stdout); };
            row("pairs (1 per 64 envs)", time_us([&](int i) { hipLaunchKernelGGL(k_pairs, dim3((n + 255) / 256), dim3(256), 0, 0, ring[i % R], n, T, work); }, reps));
            row("4 waves per 256 envs, rows shared, 1 KiB stores", time_us([&](int i) { hipLaunchKernelGGL(k_coop, dim3((n + 255) / 256), dim3(256), 0, 0, ring[i % R], n, T, work); }, reps));
            row("1 store wave per 256 envs", time_us([&](int i) { hipLaunchKernelGGL(k_cu<4>, dim3((n + 255) / 256), dim3(64), 0, 0, ring[i % R], n, T, work); }, reps));
            row("1 store wave per 512 envs", time_us([&](int i) { hipLaunchKernelGGL(k_cu<8>, dim3((n + 511) / 512), dim3(64), 0, 0, ring[i % R], n, T, work); }, reps));
            row("1 store wave per 1024 envs", time_us([&](int i) { hipLaunchKernelGGL(k_cu<16>, dim3((n + 1023) / 1024), dim3(64), 0, 0, ring[i % R], n, T, work); }, reps));
        }
        for (auto p : ring) CK(hipFree(p));
    }
    return 0;
}
