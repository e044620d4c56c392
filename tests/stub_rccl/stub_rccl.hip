// stub_rccl.hip - TEST INFRASTRUCTURE, not part of the product: the five RCCL entry points librmav.so resolves
// (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllGather, ncclGetErrorString) implemented for ranks that are
// PROCESSES SHARING ONE GPU - real RCCL refuses two ranks on one device, and the test boxes have one.  It lets the native
// statistics exchange of librmav.so (rmav_allgather_stats_arm / _post / _result / _wait, k_wait_arrivals, the 8-deep back
// pressure, the time-out path) run with world = 2 on the 1-GPU box: tests/test_gpu_stub_rccl.py.  Handed to the library with
// rmav_comm_use_library(path).
//
// Mechanism: a POSIX shared-memory segment named by the unique id holds a ring of SLOTS payload slots per rank plus two
// counters per rank; every call is enqueued ENTIRELY on the caller's stream, like the real collective:
//   wait until every peer has read the call that last used this ring slot | copy my payload device -> segment |
//   publish write_done[me] = call + 1 | wait until every peer has published this call | copy every rank's payload
//   segment -> my receive buffer | publish read_done[me] = call + 1
// The waits are one-thread kernels polling the (pinned, device-mapped) segment with system-scope loads, bounded by
// RMAV_STUB_WAIT_S seconds (default 20) so that a test that kills a rank never leaves a kernel spinning on the GPU.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

namespace {
constexpr int MAXR = 8, SLOTS = 4;   // up to 8 rank processes on the one GPU (the shape of an 8-GPU node's launch)
constexpr size_t MAXBYTES = 5u << 18;   // 1.25 MiB per rank and call (BASELINE C3's payload is 1 MiB)
struct Shm {
    std::atomic<uint32_t> joined, left;
    volatile uint32_t write_done[MAXR][16];   // [rank][0]: calls whose payload the rank has written (one cache line per rank)
    volatile uint32_t read_done[MAXR][16];    // [rank][0]: calls the rank has finished reading
    char data[SLOTS][MAXR][MAXBYTES];
};
struct Comm {
    Shm *shm, *shm_dev;
    int rank, world;
    uint32_t calls;
    unsigned long long wait_ticks;
    char name[64];
};
const char *g_last = "ok";

__global__ void k_publish(volatile uint32_t *word, uint32_t v) {
    __threadfence_system();
    __hip_atomic_store(const_cast<uint32_t *>(word), v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// wait until words[r][0] >= v for every rank r < world (bounded)
__global__ void k_wait_all(volatile uint32_t (*words)[16], int world, uint32_t v, unsigned long long max_ticks) {
    const unsigned long long t0 = wall_clock64();
    for (int r = 0; r < world; ++r) {
        while ((int32_t)(__hip_atomic_load(const_cast<uint32_t *>(&words[r][0]), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - v) < 0) {
            if (wall_clock64() - t0 > max_ticks) return;   // give up: the caller's data will be stale, its own time-out reports it
            __builtin_amdgcn_s_sleep(64);
        }
    }
}
size_t dtype_size(ncclDataType_t t) {
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
    }
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    memset(id, 0, sizeof(*id));
    timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    snprintf(id->internal, sizeof(id->internal), "/rmav_stub_%d_%ld", (int)getpid(), (long)(ts.tv_nsec ^ ts.tv_sec));
    const int fd = shm_open(id->internal, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(Shm)) != 0) { g_last = "shm_open / ftruncate failed"; return ncclSystemError; }
    close(fd);   // a fresh segment is zero-filled: counters start at 0
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank) {
    if (nranks < 1 || nranks > MAXR || rank < 0 || rank >= nranks) { g_last = "bad rank / world (stub: <= 4 ranks)"; return ncclInvalidArgument; }
    int fd = -1;
    for (int tries = 0; tries < 3000 && fd < 0; ++tries) {
        fd = shm_open(id.internal, O_RDWR, 0600);
        if (fd < 0) usleep(10000);
    }
    if (fd < 0) { g_last = "the unique id's shared-memory segment does not exist"; return ncclSystemError; }
    void *p = mmap(nullptr, sizeof(Shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { g_last = "mmap failed"; return ncclSystemError; }
    Comm *c = new Comm();
    c->shm = (Shm *)p;
    c->rank = rank;
    c->world = nranks;
    c->calls = 0;
    snprintf(c->name, sizeof(c->name), "%s", id.internal);
    const char *w = getenv("RMAV_STUB_WAIT_S");
    c->wait_ticks = (unsigned long long)((w ? atof(w) : 20.0) * 1e8);   // wall_clock64 ticks at 100 MHz
    if (hipHostRegister(p, sizeof(Shm), hipHostRegisterMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **)&c->shm_dev, p, 0) != hipSuccess) {
        g_last = "hipHostRegister of the segment failed";
        return ncclUnhandledCudaError;
    }
    c->shm->joined.fetch_add(1);
    for (int tries = 0; c->shm->joined.load() < (uint32_t)nranks; ++tries) {   // the rendezvous every rank blocks in, like the real one
        if (tries > 6000) { g_last = "rendezvous timed out (60 s)"; return ncclSystemError; }
        usleep(10000);
    }
    if (rank == 0) shm_unlink(c->name);   // every rank has it mapped: nothing is left behind whatever happens next
    *out = (ncclComm_t)c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm *c = (Comm *)comm;
    if (!c) return ncclInvalidArgument;
    (void)hipHostUnregister(c->shm);
    munmap(c->shm, sizeof(Shm));
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclComm_t comm, hipStream_t stream) {
    Comm *c = (Comm *)comm;
    const size_t bytes = count * dtype_size(dt);
    if (!c || bytes > MAXBYTES) { g_last = "payload larger than the stub's ring slot"; return ncclInvalidArgument; }
    const uint32_t call = c->calls++;
    const int slot = (int)(call % SLOTS);
    Shm *d = c->shm_dev;
    if (call >= (uint32_t)SLOTS)   // the ring slot is free once every peer has read the call that used it last
        hipLaunchKernelGGL(k_wait_all, dim3(1), dim3(1), 0, stream, d->read_done, c->world, call - SLOTS + 1, c->wait_ticks);
    if (hipMemcpyAsync(c->shm->data[slot][c->rank], send, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) return ncclUnhandledCudaError;
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(1), 0, stream, &d->write_done[c->rank][0], call + 1);
    hipLaunchKernelGGL(k_wait_all, dim3(1), dim3(1), 0, stream, d->write_done, c->world, call + 1, c->wait_ticks);
    for (int r = 0; r < c->world; ++r)
        if (hipMemcpyAsync((char *)recv + (size_t)r * bytes, c->shm->data[slot][r], bytes, hipMemcpyHostToDevice, stream) != hipSuccess)
            return ncclUnhandledCudaError;
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(1), 0, stream, &d->read_done[c->rank][0], call + 1);
    return hipGetLastError() == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int *count) {
    if (!comm || !count) return ncclInvalidArgument;
    *count = ((Comm *)comm)->world;
    return ncclSuccess;
}
ncclResult_t ncclCommUserRank(const ncclComm_t comm, int *rank) {
    if (!comm || !rank) return ncclInvalidArgument;
    *rank = ((Comm *)comm)->rank;
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : g_last; }

}  // extern "C"
