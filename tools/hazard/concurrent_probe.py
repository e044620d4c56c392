"""Is an MFMA-FREE rollout kernel exposed to the packed-fp32 hazard when ANOTHER stream puts 32x32x16 MFMAs on its SIMDs?
(profiles/r05/packed_f32_hazard.md section 3.)  Fused random-action rollouts of every kind are run alone (reference bits) and then
again while the f16 (actor, critic) policy rollout of a second env loops on a second stream; prints the number of wavefronts whose
trajectory differs in any bit, and the lanes.  RMAV_LIB_PATH selects the build (the SLP-vectorised main translation unit vs the
product's)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "reinmav-gym_amd"))
import torch
import gym_reinmav_amd as g
from gym_reinmav_amd.ppo import FusedPolicyCollector, MlpPolicy

lib = os.environ.get("RMAV_LIB_PATH", "").split("/")[-1] or "product"
dev = torch.device("cuda", 0)
T, reps = 64, int(os.environ.get("REPS", "6"))
want = ("actions", "obs", "rew", "done")
s_env, s_mfma = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(s_mfma):
    penv = g.BatchedQuadrotor("quad3d", 65536, seed=3)
    torch.manual_seed(1)
    pol = MlpPolicy(penv.nS, penv.nA).cuda()
    ro = FusedPolicyCollector(penv, pol, 32, f16_mfma=True)
    ro.collect()
torch.cuda.synchronize()
for kind, n, mode in (("quad3d", 65536, "random"), ("quad3d", 131072, "random"), ("quad3d", 65536, "controller"), ("quad3d_sl", 65536, "random"),
                      ("quad2d", 65536, "random"), ("quad3d", 262144, "random")):
    def run(beside):
        with torch.cuda.stream(s_env):
            env = g.BatchedQuadrotor(kind, n, seed=7)
            if beside:
                with torch.cuda.stream(s_mfma):
                    for _ in range(40): ro.collect()      # ~6 ms of matrix-core work queued on the other stream
            tr = env.rollout(T, mode=mode, layout="soa", want=want, device_out=True)
            tr = env.rollout(T, mode=mode, layout="soa", want=want, device_out=True, out=tr)
            s_env.synchronize()
            out = torch.cat([tr["obs"].reshape(-1, n), tr["actions"].reshape(-1, n), tr["rew"], tr["done"].float()]).clone()
            env.close()
        torch.cuda.synchronize()
        return out
    ref = run(False)
    alone = [(run(False) != ref).any(0).sum().item() for _ in range(2)]
    bad, lanes = [], set()
    for _ in range(reps):
        cur = run(True)
        envs = (cur != ref).any(0).nonzero()[:, 0]
        bad.append((envs // 64).unique().numel())
        lanes |= set((envs % 64).tolist())
    print(f"{lib:26s} {kind:10s} n={n:7d} {mode:10s} differing envs alone {alone}; beside the f16 policy rollout: bad waves/rep {bad} lanes {sorted(lanes)[:3]}..{sorted(lanes)[-2:] if lanes else ''}", flush=True)
