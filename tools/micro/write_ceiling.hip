// write_ceiling.hip - how fast can this GPU absorb the fused rollout's store pattern with no arithmetic at all?
// (diagnostic, not part of the library).  One wavefront lane per env, T time steps, per step 15 dword columns
// (4 action + 10 obs + 1 reward) + 1 byte column, time-major SoA exactly like k_rollout's trajectory.
// build here: hipcc --offload-arch=gfx950 -O3 -o tools/micro/_build/wc tools/micro/write_ceiling.hip ; run on the GPU box: tools/micro/_build/wc
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t rsrc(const void *p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, -1, 0x00020000); }

template <int AUX>
__global__ __launch_bounds__(256) void k_store(float *act, float *obs, float *rew, uint8_t *done, int n, int T) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint32_t)n) return;
    const uint32_t off = i * 4u, col = (uint32_t)n * 4u;
    float v = (float)i;
    for (int t = 0; t < T; ++t) {
        v += 1.0f;
        const rsrc_t ra = rsrc(act), ro = rsrc(obs);
#pragma unroll
        for (int c = 0; c < 4; ++c) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), ra, off, c * col, AUX);
#pragma unroll
        for (int c = 0; c < 10; ++c) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), ro, off, c * col, AUX);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rsrc(rew), off, 0, AUX);
        __builtin_amdgcn_raw_buffer_store_b8((uint8_t)t, rsrc(done), i, 0, 0);
        act += (size_t)4 * n; obs += (size_t)10 * n; rew += n; done += n;
    }
}

template <int AUX> float run(hipStream_t st, float *act, float *obs, float *rew, uint8_t *done, int n, int T, int block) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    dim3 g((n + block - 1) / block), b(block);
    for (int i = 0; i < 10; ++i) k_store<AUX><<<g, b, 0, st>>>(act, obs, rew, done, n, T);
    (void)hipStreamSynchronize(st);
    const int reps = 100;
    (void)hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i) k_store<AUX><<<g, b, 0, st>>>(act, obs, rew, done, n, T);
    (void)hipEventRecord(e1, st);
    (void)hipStreamSynchronize(st);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int n : {65536, 262144, 1048576}) {
        const int T = 64;
        float *act, *obs, *rew; uint8_t *done;
        CK(hipMalloc(&act, sizeof(float) * 4 * (size_t)n * T)); CK(hipMalloc(&obs, sizeof(float) * 10 * (size_t)n * T));
        CK(hipMalloc(&rew, sizeof(float) * (size_t)n * T)); CK(hipMalloc(&done, (size_t)n * T));
        const double bytes = 61.0 * n * T;
        for (int block : {64, 256}) {
            const float d = run<0>(st, act, obs, rew, done, n, T, block), w = run<17>(st, act, obs, rew, done, n, T, block),
                        s = run<2>(st, act, obs, rew, done, n, T, block);
            printf("n=%8d T=%d block=%3d  %.1f MB/launch:  default %7.1f us (%.2f TB/s)   sc0 sc1 %7.1f us (%.2f TB/s)   nt %7.1f us (%.2f TB/s)\n",
                   n, T, block, bytes / 1e6, d, bytes / d / 1e6, w, bytes / w / 1e6, s, bytes / s / 1e6);
        }
        // the runtime's own fill of the same number of bytes, for scale
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipMemsetAsync(obs, 0, sizeof(float) * 10 * (size_t)n * T, st); (void)hipStreamSynchronize(st);
        (void)hipEventRecord(e0, st);
        for (int i = 0; i < 20; ++i) (void)hipMemsetAsync(obs, 0, sizeof(float) * 10 * (size_t)n * T, st);
        (void)hipEventRecord(e1, st); (void)hipStreamSynchronize(st);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("n=%8d hipMemsetAsync of the obs trajectory (%.1f MB): %.1f us (%.2f TB/s)\n", n, 40.0 * n * T / 1e6, ms * 1e3 / 20, 40.0 * n * T / (ms / 20 * 1e-3) / 1e12);
        (void)hipFree(act); (void)hipFree(obs); (void)hipFree(rew); (void)hipFree(done);
    }
    return 0;
}
