#!/usr/bin/env python3
"""Where does the per-launch episode-stats exchange cost its time? (single rank, RCCL through librmav)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'reinmav-gym_amd'))
import torch
import gym_reinmav_amd as g
A = g._abi; L = A.lib()
n, T, K = int(os.environ.get("N", "131072")), 64, 500
dev = torch.device("cuda", 0)
if os.environ.get("PROBE_PG") == "1":   # with torch's own NCCL process group alive in the process, like bench.py under torchrun
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    dist.barrier()
st = torch.cuda.Stream(device=dev)
with torch.cuda.stream(st):
    env = g.BatchedQuadrotor("quad3d", n, seed=0)
    bufs = [env.rollout(T, mode="random", want=("actions", "obs", "rew", "done"), device_out=True) for _ in range(4)]
    comms = {}
    for name, ev in (("signal", "0"), ("events", "1")):
        os.environ["RMAV_EXCHANGE_EVENTS"] = ev
        uid = (C.c_char * A.COMM_ID_BYTES)(); A.check(L.rmav_comm_unique_id(uid))
        comms[name] = C.c_void_p(); A.check(L.rmav_comm_create(C.byref(comms[name]), uid, 0, 1, 0))
    comm = comms["signal"]
    send = torch.zeros(2 * n, dtype=torch.int32, device=dev)

    def loop(kind):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(K):
            env.rollout(T, mode="random", want=("actions", "obs", "rew", "done"), device_out=True, out=bufs[i % 4])
            if kind.startswith("post"):
                A.check(L.rmav_allgather_stats_post(env._h, comms[kind[5:]], n))
            elif kind == "pack":
                env.pack_stats(send)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        return (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6
    for kind in ("none", "pack", "post_signal", "post_events") * 4:
        loop(kind)
        h, w = loop(kind)
        print(f"{kind:12s}: host enqueue {h:7.1f} us/iter, wall {w:7.1f} us/iter", flush=True)
    # host cost of the post alone
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(200): A.check(L.rmav_allgather_stats_post(env._h, comm, n))
    t1 = time.perf_counter(); torch.cuda.synchronize()
    print(f"post alone: host {(t1 - t0) / 200 * 1e6:.1f} us", flush=True)
