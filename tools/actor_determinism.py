import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinmav-gym_amd"))
import numpy as np, torch
import gym_reinmav_amd as g
from gym_reinmav_amd.ppo import FusedPolicyCollector, MlpPolicy
kind, T, seed = "quad3d", 32, 17
torch.manual_seed(4)
pol = None
lib = os.environ.get("RMAV_LIB_PATH", "default").split("/")[-1]
cfgs = [("bf16", 65536, {"pair_group": 2}), ("f16", 65536, {"pair_group": 1}), ("f16", 65536, {"pair_group": 2}), ("f16", 65536, {"pair_group": 4}),
        ("f16", 131072, {"pair_group": 2}), ("bf16", 131072, {"pair_group": 4}), ("bf16_1w", 262144, {"policy_pair": 0}), ("fp32_mfma", 131072, {})]
for actor, n, tune in cfgs:
    ref, nbad = None, []
    for rep in range(int(os.environ.get("REPS", "6"))):
        env = g.BatchedQuadrotor(kind, n, seed=seed)
        if tune: env.set_tuning(**tune)
        if pol is None:
            pol = MlpPolicy(env.nS, env.nA, init_logstd=0.5).cuda()
            with torch.no_grad():
                pol.pi[2].weight.mul_(30.0); pol.pi[2].bias.uniform_(0.5, 4.0)
        ro = FusedPolicyCollector(env, pol, T, bf16_mfma=actor.startswith("bf16"), f16_mfma=(actor == "f16"), f32_mfma=(actor == "fp32_mfma"))
        ro.collect(); torch.cuda.synchronize()
        cur = torch.cat([ro.obs.reshape(-1, n), ro.act.reshape(-1, n), ro.val, ro.logp])
        env.close()
        if ref is None: ref = cur; continue
        envs = (cur != ref).any(0).nonzero()[:, 0]
        nbad.append(((envs // 64).unique().numel(), sorted(set((envs % 64).tolist()))[:2]))
    print(lib, actor, n, tune, "bad waves per rep:", nbad, flush=True)
