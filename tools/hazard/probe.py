"""Determinism probe of the matrix-core policy rollouts for ONE build of librmav.so (RMAV_LIB_PATH): every config is run REPS
times from the same seed; a wavefront is 'bad' when any of its 64 envs differs in any output bit from the first run.  Prints
one line per config: bad wavefronts per repeat, the lanes seen, and the kernel time (so a variant's cost is visible too)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "reinmav-gym_amd"))
import torch
import gym_reinmav_amd as g
from gym_reinmav_amd.ppo import FusedPolicyCollector, MlpPolicy

kind, T, seed = "quad3d", 32, 17
torch.manual_seed(4)
pol = None
lib = os.environ.get("RMAV_LIB_PATH", "default").split("/")[-2] if os.environ.get("RMAV_LIB_PATH") else "product"
cfgs = [("f16", 65536, {"pair_group": 1}), ("f16", 65536, {"pair_group": 2}), ("bf16", 65536, {"pair_group": 2}), ("f16", 131072, {"pair_group": 2}),
        ("bf16_1w", 262144, {"policy_pair": 0})]
reps = int(os.environ.get("REPS", "5"))
for actor, n, tune in cfgs:
    ref, nbad, lanes, ms = None, [], set(), []
    for rep in range(reps):
        env = g.BatchedQuadrotor(kind, n, seed=seed)
        env.set_tuning(**tune)
        if pol is None:
            pol = MlpPolicy(env.nS, env.nA, init_logstd=0.5).cuda()
            with torch.no_grad():
                pol.pi[2].weight.mul_(30.0); pol.pi[2].bias.uniform_(0.5, 4.0)
        ro = FusedPolicyCollector(env, pol, T, bf16_mfma=actor.startswith("bf16"), f16_mfma=(actor == "f16"))
        ro.collect(); torch.cuda.synchronize()
        cur = torch.cat([ro.obs.reshape(-1, n), ro.act.reshape(-1, n), ro.val, ro.logp])
        if rep == reps - 1:   # time the kernel on the last repeat
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): ro.collect()
            e1.record(); torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1) / 10)
        env.close()
        if ref is None: ref = cur; continue
        envs = (cur != ref).any(0).nonzero()[:, 0]
        nbad.append((envs // 64).unique().numel())
        lanes |= set((envs % 64).tolist())
    print(f"{lib:18s} {actor:8s} n={n:7d} {str(tune):22s} bad waves/rep {nbad}  lanes {sorted(lanes)[:4]}..{sorted(lanes)[-2:] if lanes else ''}  {ms[0]*1e3:7.1f} us/rollout", flush=True)
