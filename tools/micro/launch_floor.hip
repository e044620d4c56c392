// launch_floor.hip - what does one dependent launch cost on this GPU?  (diagnostic, not part of the library)
// build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/lf tools/micro/launch_floor.hip && /tmp/lf
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty(float *p) {}
template <int NL, int NS>
__global__ void k_copy(const float *__restrict__ in, float *__restrict__ out, int n, int spin) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v[NL];
#pragma unroll
    for (int c = 0; c < NL; ++c) v[c] = in[(size_t)c * n + i];
    float acc = 0.f;
    for (int s = 0; s < spin; ++s)
#pragma unroll
        for (int c = 0; c < NL; ++c) acc = __builtin_fmaf(v[c], 1.0001f, acc);
#pragma unroll
    for (int c = 0; c < NS; ++c) out[(size_t)c * n + i] = v[c % NL] + acc;
}

template <typename F> float time_chain(hipStream_t st, int reps, F launch) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 200; ++i) launch(i);
    (void)hipStreamSynchronize(st);
    (void)hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i) launch(i);
    (void)hipEventRecord(e1, st);
    (void)hipStreamSynchronize(st);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

int main() {
    const int n = 65536;
    float *a, *b;
    CK(hipMalloc(&a, sizeof(float) * n * 32)); CK(hipMalloc(&b, sizeof(float) * n * 32));
    CK(hipMemset(a, 0, sizeof(float) * n * 32)); CK(hipMemset(b, 0, sizeof(float) * n * 32));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int reps = 4000;
    for (int block : {64, 256}) {
        dim3 g(n / block), bl(block);
        printf("block %3d  empty kernel chain         : %6.2f us/launch\n", block, time_chain(st, reps, [&](int) { k_empty<<<g, bl, 0, st>>>(a); }));
        printf("block %3d  14 loads -> 11 stores       : %6.2f us/launch\n", block, time_chain(st, reps, [&](int i) { k_copy<14, 11><<<g, bl, 0, st>>>((i & 1) ? a : b, (i & 1) ? b : a, n, 0); }));
        printf("block %3d  18 loads -> 15 stores       : %6.2f us/launch\n", block, time_chain(st, reps, [&](int i) { k_copy<18, 15><<<g, bl, 0, st>>>((i & 1) ? a : b, (i & 1) ? b : a, n, 0); }));
        printf("block %3d  14 -> 11 + ~300 VALU        : %6.2f us/launch\n", block, time_chain(st, reps, [&](int i) { k_copy<14, 11><<<g, bl, 0, st>>>((i & 1) ? a : b, (i & 1) ? b : a, n, 20); }));
        printf("block %3d  in-place 14 -> 11 (same buf): %6.2f us/launch\n", block, time_chain(st, reps, [&](int) { k_copy<14, 11><<<g, bl, 0, st>>>(a, a, n, 0); }));
    }
    return 0;
}
