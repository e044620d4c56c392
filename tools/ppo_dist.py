#!/usr/bin/env python3
"""BASELINE config 5 shape: quadrotor3d-v0, 65 536 envs per GPU, PPO2-style loop (fused in-kernel rollout +
torch learner), data-parallel over the GPUs of one node.  Launch:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/ppo_dist.py

Each rank owns a contiguous shard of the global env ids; parameters are broadcast from rank 0; gradients are
averaged with one flat all-reduce per minibatch; episode statistics are all-gathered once per iteration.
Rank 0 prints one JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinmav-gym_amd"))
import torch
import torch.distributed as dist
import gym_reinmav_amd as g
from gym_reinmav_amd.distributed import all_gather_episode_stats, all_reduce_totals
from gym_reinmav_amd.ppo import PPO, FusedPolicyCollector, MlpPolicy, sync_parameters

world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); lr = int(os.environ.get("LOCAL_RANK", "0"))
N = int(os.environ.get("N", 65536)); T = int(os.environ.get("T", 32)); iters = int(os.environ.get("ITERS", 5))
bf16 = os.environ.get("BF16", "1") == "1"
torch.cuda.set_device(lr)
dev = torch.device("cuda", lr)
if "RANK" in os.environ:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
torch.manual_seed(0)
env = g.BatchedQuadrotor("quad3d", N, device=lr, seed=0, env_id_base=rank * N)
pol = MlpPolicy(env.nS, env.nA).to(dev)
sync_parameters(pol)
ro = FusedPolicyCollector(env, pol, T, bf16_mfma=bf16)
ppo = PPO(pol)
ro.collect(); ppo.update(ro); ro.roll_over()          # warm-up
torch.cuda.synchronize()
if dist.is_initialized():
    dist.barrier()
t0 = time.perf_counter()
t_roll = 0.0
for it in range(iters):
    torch.cuda.synchronize(); a = time.perf_counter()
    ro.collect()
    torch.cuda.synchronize(); t_roll += time.perf_counter() - a
    stats = ppo.update(ro)
    ro.roll_over()
    if dist.is_initialized():
        eb = env.episode_buffers(device_out=True)
        rets, lens = all_gather_episode_stats(eb["last_return"], eb["last_length"], N * world)
torch.cuda.synchronize()
if dist.is_initialized():
    dist.barrier()
dt = time.perf_counter() - t0
tot = env.episode_totals()
if dist.is_initialized():
    tot = all_reduce_totals(tot, device=dev)
    w = torch.tensor([dt], dtype=torch.float64, device=dev); dist.all_reduce(w, op=dist.ReduceOp.MAX); dt = float(w)
if rank == 0:
    print(json.dumps({"what": "PPO2-style loop, fused in-kernel rollout + torch learner", "n_gpus": world, "envs_per_gpu": N,
                      "nsteps": T, "iterations": iters, "actor": "bf16 MFMA" if bf16 else "fp32",
                      "env_steps_per_s": N * world * T * iters / dt, "rollout_only_env_steps_per_s_per_gpu": N * T * iters / t_roll,
                      "ms_per_iteration": 1e3 * dt / iters, "mean_episode_return": tot["return_sum"] / max(1, tot["episodes"]),
                      "stats": stats}), flush=True)
env.close()
if dist.is_initialized():
    dist.destroy_process_group()
