"""Stand-alone check of the stub all-gather: WORLD processes on GPU 0, a few calls, results compared (python selftest.py)."""
import ctypes as C, os, subprocess, sys, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) == 1:
    tmp = f"/tmp/stub_selftest_{os.getpid()}"; os.makedirs(tmp, exist_ok=True)
    ps = [subprocess.Popen([sys.executable, __file__, str(r), "2", tmp]) for r in range(2)]
    sys.exit(max(p.wait() for p in ps))
rank, world, tmp = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
hip = C.CDLL("libamdhip64.so"); L = C.CDLL(os.path.join(HERE, "_build", "librccl_stub.so"))
uid = (C.c_char * 128)()
idf = os.path.join(tmp, "uid")
if rank == 0:
    assert L.ncclGetUniqueId(uid) == 0; open(idf + ".t", "wb").write(uid.raw); os.rename(idf + ".t", idf)
while not os.path.exists(idf): time.sleep(0.01)
class UID(C.Structure): _fields_ = [("b", C.c_char * 128)]
u = UID(); C.memmove(C.byref(u), open(idf, "rb").read(), 128)
comm = C.c_void_p()
L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UID, C.c_int]
t0 = time.time(); rc = L.ncclCommInitRank(C.byref(comm), world, u, rank); print(rank, "init rc", rc, round(time.time() - t0, 2), flush=True)
n = 50000
send, recv = C.c_void_p(), C.c_void_p()
hip.hipMalloc(C.byref(send), C.c_size_t(4 * n)); hip.hipMalloc(C.byref(recv), C.c_size_t(4 * n * world))
st = C.c_void_p(); hip.hipStreamCreate(C.byref(st))
L.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
for it in range(10):
    x = (np.arange(n, dtype=np.int32) * (rank + 1) + it).copy()
    hip.hipMemcpy(send, x.ctypes.data_as(C.c_void_p), C.c_size_t(4 * n), 1)
    t0 = time.time()
    rc = L.ncclAllGather(send, recv, C.c_size_t(n), 2, comm, st)   # ncclInt32 = 2
    t1 = time.time()
    hip.hipStreamSynchronize(st)
    t2 = time.time()
    y = np.empty(n * world, np.int32); hip.hipMemcpy(y.ctypes.data_as(C.c_void_p), recv, C.c_size_t(4 * n * world), 2)
    exp = np.concatenate([np.arange(n, dtype=np.int32) * (r + 1) + it for r in range(world)])
    print(rank, "call", it, "rc", rc, "enqueue", round(t1 - t0, 4), "sync", round(t2 - t1, 4), "ok" if np.array_equal(y, exp) else "MISMATCH", flush=True)
L.ncclCommDestroy(comm)
