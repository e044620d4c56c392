import sys, ctypes as C
sys.path.insert(0, "/root/repo/reinmav-gym_amd")
import torch
import gym_reinmav_amd as g
A = g._abi; L = A.lib()
import os
WITH_COMM = os.environ.get("WITH_COMM", "1") == "1"
def free():
    torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0]
e = g.BatchedQuadrotor("quad3d", 65536); e.rollout(8, mode="random", want=()); e.close()
f0 = free()
for i in range(400):
    e = g.BatchedQuadrotor(["quad3d", "quad3d_sl", "quad2d", "reinmav"][i % 4], 65536 + 17 * (i % 5), seed=i)
    e.rollout(8, mode="random" if i % 4 != 3 else "controller", want=())
    e.step(torch.zeros((e.num_envs, e.nA), device="cuda")) if i % 4 != 3 else None
    if i % 7 == 0 and WITH_COMM:
        uid = (C.c_char * A.COMM_ID_BYTES)(); A.check(L.rmav_comm_unique_id(uid)); comm = C.c_void_p()
        A.check(L.rmav_comm_create(C.byref(comm), uid, 0, 1, 0))
        for _ in range(12):
            A.check(L.rmav_allgather_stats_arm(e._h, comm, e.num_envs)); e.rollout(8, mode="random", want=()); A.check(L.rmav_allgather_stats_post(e._h, comm, e.num_envs))
        A.check(L.rmav_comm_destroy(comm))
    e.close()
f1 = free()
print(f"free device memory before {f0 / 1e6:.1f} MB, after 400 create / use / destroy cycles {f1 / 1e6:.1f} MB, delta {(f0 - f1) / 1e6:.2f} MB")
