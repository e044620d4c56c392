#!/usr/bin/env python3
"""How do the rollout kernels tolerate a co-resident communication kernel?  One GPU: after every rollout launch the
exchange posts a stand-in for a multi-GPU ring all-gather (RMAV_DBG_EXCHANGE=3: W workgroups x 256 threads x L bytes of
LDS holding their CU slots for U microseconds) on the communicator's stream, overlapping the next rollout."""
import ctypes as C, os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.join(ROOT, "reinmav-gym_amd"))
    import torch
    import gym_reinmav_amd as g
    A = g._abi; L = A.lib()
    n, T, K = int(os.environ.get("N", "131072")), 64, 400
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        env = g.BatchedQuadrotor("quad3d", n, seed=0)
        bufs = [env.rollout(T, mode="random", want=("actions", "obs", "rew", "done"), device_out=True) for _ in range(4)]
        uid = (C.c_char * A.COMM_ID_BYTES)(); A.check(L.rmav_comm_unique_id(uid))
        comm = C.c_void_p(); A.check(L.rmav_comm_create(C.byref(comm), uid, 0, 1, 0))
        def loop(post):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(K):
                env.rollout(T, mode="random", want=("actions", "obs", "rew", "done"), device_out=True, out=bufs[i % 4])
                if post:
                    A.check(L.rmav_allgather_stats_post(env._h, comm, n))
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / K * 1e6
        loop(True); loop(False)
        print(f"{loop(False):.1f} {loop(True):.1f}")
    sys.exit(0)
print("| envs | rollout kernel | stand-in collective | alone us | with the collective us | cost |")
print("|---|---|---|---|---|---|")
for n in [int(x) for x in os.environ.get("PROBE_N", "65536,131072").split(",")]:
    for split in os.environ.get("PROBE_SPLIT", "1,0").split(","):
        for wgs, lds, us in ((8, 16384, 40), (16, 32768, 80), (32, 65536, 80), (16, 32768, 60)):
            env = dict(os.environ, N=str(n), RMAV_SPLIT=split, RMAV_DBG_EXCHANGE="3", RMAV_DBG_OCC_WGS=str(wgs), RMAV_DBG_OCC_LDS=str(lds), RMAV_DBG_OCC_US=str(us))
            r = subprocess.run([sys.executable, __file__, "child"], capture_output=True, text=True, env=env)
            try:
                a, b = (float(x) for x in re.search(r"^([0-9.]+) ([0-9.]+)$", r.stdout, re.M).groups())
                print(f"| {n} | {'two wavefronts, one workgroup per CU' if split == '1' else 'one wavefront'} | {wgs} x 256 threads, {lds // 1024} KiB LDS, {us} us | {a:.1f} | {b:.1f} | {b - a:+.1f} us |", flush=True)
            except Exception:
                print("ERR", r.stdout[-300:], r.stderr[-300:])
