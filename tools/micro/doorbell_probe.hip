// doorbell_probe.hip - feasibility numbers for a resident "step server" (diagnostic, not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/db tools/micro/doorbell_probe.hip && timeout 60 /tmp/db
// A persistent kernel: one dispatcher wavefront polls a command ring in pinned host memory and republishes new
// commands into a device-memory ring (agent scope); W worker wavefronts consume them in order, do a token amount of
// memory work per command, and count completions; the last one stores the finished sequence number into pinned host
// memory.  Measured: (a) host round trip of ONE command (doorbell write -> completion visible), (b) commands per second
// with the ring kept full, for W = 1 and W = 1024.  Every spin loop has a wall-clock bound (the kernel cannot outlive
// `life_ticks` of the 100 MHz clock), and the host gives up after 2 s.
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int R = 256;   // ring slots
struct Ctl {             // pinned host memory
    volatile uint64_t cmd[R];      // (seq << 16) | payload16 ; 0 = empty
    volatile uint64_t host_head;   // last sequence number the host has published
    volatile uint64_t tail;        // last sequence number every worker has finished (written by the GPU)
    volatile uint64_t exited;      // 1 + reason once the kernel has left
};
struct Dev {             // device memory
    uint64_t cmd[R];
    uint64_t head;                 // last sequence number published to the workers
    uint64_t progress[2048];       // per worker: last sequence number it has finished (single writer each)
};

__device__ __forceinline__ uint64_t ld_sys(const volatile uint64_t *p) {
    return __hip_atomic_load(const_cast<const uint64_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ uint64_t ld_agent(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Completion without a shared counter (1024 same-address atomics per command would cost ~12 us): every worker publishes its
// own progress word; the dispatcher - idle between commands anyway - takes the minimum and is the ONLY writer of `tail`.
__global__ __launch_bounds__(64) void k_server(Ctl *ctl, Dev *dev, float *work, int n_workers, uint64_t idle_ticks, uint64_t life_ticks) {
    const uint64_t t_start = wall_clock64();
    const int lane = threadIdx.x;
    if ((int)blockIdx.x == n_workers) {   // dispatcher
        uint64_t seq = 0, tail = 0, t_last = wall_clock64();
        for (;;) {
            const uint64_t hh = ld_sys(&ctl->host_head);
            if (hh > seq) {
                for (uint64_t s = seq + 1 + lane; s <= hh; s += 64) {
                    uint64_t c;
                    do { c = ld_sys(&ctl->cmd[s % R]); } while ((c >> 16) != s && wall_clock64() - t_start < life_ticks);
                    __hip_atomic_store(&dev->cmd[s % R], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the agent-scope command stores are acknowledged
                if (lane == 0) __hip_atomic_store(&dev->head, hh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                seq = hh;
                t_last = wall_clock64();
            }
            if (tail < seq) {   // something is in flight: minimum of the workers' progress words
                uint64_t m = ~0ull;
                for (int w = lane; w < n_workers; w += 64) {
                    const uint64_t p = ld_agent(&dev->progress[w]);
                    m = p < m ? p : m;
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const uint64_t o = __shfl_xor(m, off, 64);
                    m = o < m ? o : m;
                }
                if (m > tail) {
                    tail = m;
                    if (lane == 0) __hip_atomic_store(const_cast<uint64_t *>(&ctl->tail), tail, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                    t_last = wall_clock64();
                }
            } else if (hh <= seq) {
                const uint64_t now = wall_clock64();
                if (now - t_last > idle_ticks || now - t_start > life_ticks) {   // publish STOP (payload 0xFFFF)
                    if (lane == 0) {
                        __hip_atomic_store(&dev->cmd[(seq + 1) % R], ((seq + 1) << 16) | 0xFFFFull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __hip_atomic_store(&dev->head, seq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    return;
                }
                __builtin_amdgcn_s_sleep(4);
            }
            if (wall_clock64() - t_start > life_ticks + 50000000ull) return;
        }
    }
    // worker
    float acc = work[blockIdx.x * 64 + lane];
    uint64_t seq = 0;
    for (;;) {
        for (;;) {
            if (ld_agent(&dev->head) > seq) break;
            if (wall_clock64() - t_start > life_ticks + 100000000ull) {   // dispatcher gone: bail out (1 s past its life)
                if (lane == 0 && blockIdx.x == 0) __hip_atomic_store(const_cast<uint64_t *>(&ctl->exited), 3ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                return;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        // no fences (an agent-scope fence writes back / invalidates the XCD's whole L2: 15 us per command with 1024 wavefronts
        // doing it): everything that crosses wavefronts moves with agent-scope (sc1) loads and stores, ordered by vmcnt
        const uint64_t s = seq + 1;
        const uint64_t c = ld_agent(&dev->cmd[s % R]);
        if ((c & 0xFFFFull) == 0xFFFFull) break;
        acc = acc * 1.0001f + (float)(c & 0xFFFFull);      // token work
        __hip_atomic_store(&work[blockIdx.x * 64 + lane], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(&dev->progress[blockIdx.x], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        seq = s;
    }
    work[blockIdx.x * 64 + lane] = acc;
    if (lane == 0 && blockIdx.x == 0) __hip_atomic_store(const_cast<uint64_t *>(&ctl->exited), 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    Ctl *ctl; Dev *dev; float *work;
    CK(hipHostMalloc((void **)&ctl, sizeof(Ctl), hipHostMallocMapped | hipHostMallocCoherent));
    Ctl *ctl_dev; CK(hipHostGetDevicePointer((void **)&ctl_dev, ctl, 0));
    CK(hipMalloc((void **)&dev, sizeof(Dev)));
    CK(hipMalloc((void **)&work, sizeof(float) * 64 * 2048));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    int large_bar = 0; (void)hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, 0);
    setvbuf(stdout, nullptr, _IONBF, 0);
    printf("large BAR: %d\n", large_bar);
    for (int W : {1, 64, 1024}) {
        memset((void *)ctl, 0, sizeof(Ctl));
        CK(hipMemsetAsync(dev, 0, sizeof(Dev), st)); CK(hipMemsetAsync(work, 0, sizeof(float) * 64 * 2048, st));
        CK(hipStreamSynchronize(st));
        hipLaunchKernelGGL(k_server, dim3(W + 1), dim3(64), 0, st, ctl_dev, dev, work, W, (uint64_t)2000000ull /* 20 ms idle */, (uint64_t)1500000000ull /* 15 s life */);
        CK(hipGetLastError());
        uint64_t seq = 0;
        auto post = [&](uint16_t payload) {
            ++seq;
            ctl->cmd[seq % R] = (seq << 16) | payload;
            std::atomic_thread_fence(std::memory_order_release);
            ctl->host_head = seq;
        };
        auto wait = [&](uint64_t s) { const double t0 = now_s(); while (ctl->tail < s) { if (now_s() - t0 > 2.0) return false; } return true; };
        // (a) round trips, one command in flight
        for (int i = 0; i < 200; ++i) {
            post(1);
            if (!wait(seq)) {
                printf("W=%d: timeout in warm-up at %d: seq %llu tail %llu exited %llu\n", W, i, (unsigned long long)seq,
                       (unsigned long long)ctl->tail, (unsigned long long)ctl->exited);
                return 1;
            }
        }
        printf("W=%d: warm-up ok\n", W);
        const int NRT = 2000;
        double t0 = now_s();
        for (int i = 0; i < NRT; ++i) { post(1); if (!wait(seq)) { printf("W=%d: timeout\n", W); return 1; } }
        const double rt = (now_s() - t0) / NRT;
        // (b) throughput: keep up to R/2 commands in flight
        const int NTP = 100000;
        t0 = now_s();
        bool stuck = false;
        for (int i = 0; i < NTP && !stuck; ++i) {
            const double ts = now_s();
            while (seq - ctl->tail >= R / 2) {
                if (now_s() - ts > 2.0) { stuck = true; break; }
            }
            post(2);
        }
        if (stuck) {
            printf("W=%d: throughput run stuck: seq %llu tail %llu exited %llu\n", W, (unsigned long long)seq, (unsigned long long)ctl->tail,
                   (unsigned long long)ctl->exited);
            fflush(stdout);
            return 1;
        }
        if (!wait(seq)) { printf("W=%d: timeout at the end of the throughput run\n", W); return 1; }
        const double tp = (now_s() - t0) / NTP;
        // idle exit
        t0 = now_s();
        while (!ctl->exited && now_s() - t0 < 2.0) {}
        const double idle = now_s() - t0;
        CK(hipStreamSynchronize(st));
        printf("W=%4d workers: round trip %.2f us, pipelined %.3f us/command (%.2f M/s), idle exit after %.1f ms (exited=%llu)\n", W, rt * 1e6,
               tp * 1e6, 1e-6 / tp, idle * 1e3, (unsigned long long)ctl->exited);
    }
    return 0;
}
