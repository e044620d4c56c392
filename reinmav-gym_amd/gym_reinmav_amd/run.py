"""``python -m gym_reinmav_amd.run --alg=ppo2 --env=quadrotor3d-v0 --network=mlp --num_env=65536 ...``

The reference's training entry point (``gym_reinmav/run.py``, itself a copy of ``baselines/run.py``) builds
``SubprocVecEnv``/``DummyVecEnv`` of Monitor-wrapped envs with ``make_vec_env`` (:89), calls
``baselines.ppo2.learn(env, seed, total_timesteps, network='mlp', **kwargs)`` (:63-68), optionally saves the
model (:186-188) and runs a play loop that prints ``episode_rew=...`` (:190-211).  TensorFlow 1 / baselines are
third party and absent; this module keeps the *command line and the flow* and puts the MI355X path underneath:

* the env batch is one ``BatchedQuadrotor`` (``--num_env`` envs on this GPU; under ``torch.distributed.run`` every
  rank owns a shard of global env ids and gradients are averaged - the counterpart of baselines' MPI mode, :177-182);
* the rollout is the fused in-kernel one (``FusedPolicyCollector``; ``--actor=bf16`` selects the MFMA actor,
  ``--actor=torch`` the plain torch policy + ``rmav_step`` captured in a hipGraph);
* the learner is ``gym_reinmav_amd.ppo.PPO`` with baselines' ppo2 defaults; extra ``--key=value`` arguments are
  passed to it like baselines passes them to ``learn`` (:151-163; parsed with ``ast.literal_eval``, not ``eval``).

Same flags as baselines' ``common_arg_parser`` where they make sense here; the rest are accepted and ignored
(``--env_type``, ``--gamestate``, ``--save_video_*``) so existing command lines keep working.
"""
from __future__ import annotations

import argparse
import ast
import json
import os
import sys
import time


def arg_parser():
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--env", default="quadrotor3d-v0")
    p.add_argument("--env_type", default=None, help="ignored (baselines derives it from the entry point, run.py:97-119)")
    p.add_argument("--seed", type=int, default=None)
    p.add_argument("--alg", default="ppo2")
    p.add_argument("--num_timesteps", type=float, default=1e7)
    p.add_argument("--network", default="mlp")
    p.add_argument("--gamestate", default=None, help="ignored")
    p.add_argument("--num_env", type=int, default=None, help="envs on this GPU (default 4096)")
    p.add_argument("--reward_scale", type=float, default=1.0)
    p.add_argument("--save_path", default=None)
    p.add_argument("--save_video_interval", type=int, default=0, help="ignored")
    p.add_argument("--save_video_length", type=int, default=200, help="ignored")
    p.add_argument("--play", default=False, action="store_true")
    p.add_argument("--play_episodes", type=int, default=10, help="episodes of env 0 to print in --play (the reference loops forever)")
    p.add_argument("--extra_import", default=None)
    p.add_argument("--actor", default="fp32", choices=["fp32", "bf16", "torch"], help="rollout policy evaluation")
    p.add_argument("--device", type=int, default=None)
    return p


def parse_unknown(unknown):
    """'--k=v' / '--k v' leftovers -> learn kwargs (baselines parse_cmdline_kwargs, run.py:151-163)."""
    out, key = {}, None
    for tok in unknown:
        if tok.startswith("--"):
            if "=" in tok:
                k, v = tok[2:].split("=", 1)
                out[k] = v
                key = None
            else:
                key = tok[2:]
        elif key is not None:
            out[key] = tok
            key = None

    def parse(v):
        try:
            return ast.literal_eval(v)
        except (ValueError, SyntaxError):
            return v

    return {k: parse(v) for k, v in out.items()}


def learn(env_id, num_env, seed, total_timesteps, reward_scale=1.0, actor="fp32", device=0, rank=0, world=1, log=print,
          nsteps=64, lr=3e-4, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5, gamma=0.99, lam=0.95, log_interval=10,
          nminibatches=4, noptepochs=4, cliprange=0.2, load_path=None, init_logstd=0.0, **ignored):
    """PPO2 on the batched env.  Returns (policy, env)."""
    import torch

    from . import BatchedQuadrotor
    from .distributed import all_reduce_totals
    from .ppo import PPO, FusedPolicyCollector, MlpPolicy, RolloutCollector, sync_parameters
    from .vec_env import ENV_IDS

    if ignored:
        log(f"ignoring unsupported learn arguments: {sorted(ignored)}")
    kind = ENV_IDS.get(env_id, env_id)
    torch.manual_seed(seed or 0)
    env = BatchedQuadrotor(kind, num_env, device=device, seed=seed or 0, env_id_base=rank * num_env)
    pol = MlpPolicy(env.nS, env.nA, init_logstd=init_logstd).to(f"cuda:{device}")
    if load_path:
        pol.load_state_dict(torch.load(load_path, map_location=f"cuda:{device}"))
    sync_parameters(pol)
    torch.manual_seed((seed or 0) + rank)   # the torch actor's exploration noise must differ between env shards
    if actor == "torch":
        ro = RolloutCollector(env, pol, nsteps, graph=True)
    else:
        ro = FusedPolicyCollector(env, pol, nsteps, bf16_mfma=(actor == "bf16"))
    ppo = PPO(pol, lr=lr, clip=cliprange, epochs=noptepochs, minibatches=nminibatches, vf_coef=vf_coef, ent_coef=ent_coef,
              max_grad_norm=max_grad_norm, gamma=gamma, lam=lam, reward_scale=reward_scale)
    nbatch = nsteps * num_env * world
    nupdates = max(1, int(total_timesteps) // nbatch)
    t_first = time.perf_counter()
    for update in range(1, nupdates + 1):
        t0 = time.perf_counter()
        env.episode_totals(clear=True)
        ro.collect()
        stats = ppo.update(ro)
        ro.roll_over()
        if update % log_interval == 0 or update == 1 or update == nupdates:
            tot = env.episode_totals()
            if world > 1:
                tot = all_reduce_totals(tot, device=f"cuda:{device}")
            torch.cuda.synchronize()
            fps = int(nbatch / (time.perf_counter() - t0))
            if rank == 0:   # the keys baselines' ppo2 logs
                log(json.dumps({"nupdates": update, "total_timesteps": update * nbatch, "fps": fps,
                                "eprewmean": tot["return_sum"] / max(1, tot["episodes"]),
                                "eplenmean": tot["length_sum"] / max(1, tot["episodes"]),
                                "explained_variance": stats["explained_variance"], "policy_loss": stats["pg_loss"],
                                "value_loss": stats["vf_loss"], "time_elapsed": time.perf_counter() - t_first}))
    return pol, env


def play(pol, env, episodes, log=print):
    """run.py:190-211 without render(): act with the trained policy, print episode_rew of env 0."""
    import torch

    obs = env.get_state(layout="soa", device_out=True)
    episode_rew, done_eps = 0.0, 0
    out = (torch.empty_like(obs), torch.empty(env.num_envs, device=obs.device),
           torch.empty(env.num_envs, dtype=torch.uint8, device=obs.device))
    with torch.no_grad():
        while done_eps < episodes:
            mean, _ = pol(obs)
            act = mean + torch.exp(pol.logstd)[:, None] * torch.randn_like(mean)
            obs, rew, done = env.step(act, layout="soa", out=out)
            episode_rew += float(rew[0])
            if bool(done[0]):
                log(f"episode_rew={episode_rew}")
                episode_rew = 0.0
                done_eps += 1


def main(argv):
    args, unknown = arg_parser().parse_known_args(argv[1:])
    extra = parse_unknown(unknown)
    if args.extra_import:
        __import__(args.extra_import)
    if args.alg != "ppo2":
        raise SystemExit(f"--alg={args.alg}: only ppo2 is implemented on this path")
    if args.network != "mlp":
        raise SystemExit(f"--network={args.network}: the in-kernel policy is baselines' 2 x 64 tanh 'mlp'")
    import torch
    import torch.distributed as dist

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    device = args.device if args.device is not None else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(device)
    if "RANK" in os.environ and world > 1:   # the counterpart of baselines' MPI mode (run.py:177-182)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
    log = print if rank == 0 else (lambda *a, **k: None)
    num_env = args.num_env or 4096
    log(f"Training {args.alg} on {args.env} with {num_env} envs/GPU x {world} GPU(s), arguments {extra}")
    pol, env = learn(args.env, num_env, args.seed, args.num_timesteps, reward_scale=args.reward_scale, actor=args.actor,
                     device=device, rank=rank, world=world, log=log, **extra)
    if args.save_path is not None and rank == 0:
        path = os.path.expanduser(args.save_path)
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        torch.save(pol.state_dict(), path)
        log(f"saved model to {path}")
    if args.play and rank == 0:
        log("Running trained model")
        play(pol, env, args.play_episodes, log=log)
    env.close()
    if dist.is_initialized():
        dist.destroy_process_group()
    return pol


if __name__ == "__main__":
    main(sys.argv)
