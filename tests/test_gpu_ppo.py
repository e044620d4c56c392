"""The PPO2-style caller on the GPU: rollout buffers agree with the oracle step by step (eager and
hipGraph-captured), and the learner runs."""
import numpy as np
import pytest

import oracle as O
from util import TOL, near_threshold, scaled_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G(built):
    import torch

    assert torch.cuda.is_available()
    import gym_reinmav_amd as g

    return g


def _check_rollout(kind, seed, ro, rc0, base=0):
    import torch

    torch.cuda.synchronize()
    obs, act = ro.obs.cpu().numpy(), ro.act.cpu().numpy()
    rew, done = ro.rew.cpu().numpy(), ro.done.cpu().numpy().astype(bool)
    T, N = rew.shape
    rc = rc0.copy()
    for t in range(T):
        o2, r, d, _ = O.batch_step(kind, obs[t].T.astype(np.float64), act[t].T.astype(np.float64))
        ok = near_threshold(kind, o2)
        assert np.array_equal(done[t] | ok, d | ok)
        alive = ~done[t] & ~d
        assert scaled_err(obs[t + 1].T[alive], o2[alive]).max() <= TOL
        assert scaled_err(rew[t][alive], r[alive]).max() <= TOL
        if done[t].any():  # auto-reset: next obs is the env's next reset state
            assert np.array_equal(obs[t + 1].T[done[t]], O.reset_states(kind, seed, base + np.nonzero(done[t])[0], rc[done[t]]))
        rc += done[t].astype(np.uint32)
    return rc


@pytest.mark.parametrize("graph", [False, True])
def test_rollout_collector_matches_oracle(G, graph):
    import torch
    from gym_reinmav_amd.ppo import MlpPolicy, RolloutCollector

    torch.manual_seed(1)
    kind, N, T, seed = "quad3d", 2048, 24, 3
    env = G.BatchedQuadrotor(kind, N, seed=seed)
    pol = MlpPolicy(env.nS, env.nA, init_logstd=1.0).cuda()   # wide exploration: episodes do terminate
    ro = RolloutCollector(env, pol, T, graph=graph)
    rc = env.get_reset_counts()
    for it in range(6):
        ro.collect()
        rc = _check_rollout(kind, seed, ro, rc)
        assert np.array_equal(env.get_reset_counts(), rc)
        assert np.array_equal(env.get_state(layout="soa"), ro.obs[T].cpu().numpy())
        ro.roll_over()
    assert ro.done.sum() > 0 or rc.max() > 1
    v = ro.val.cpu().numpy()
    assert np.isfinite(v).all() and np.isfinite(ro.logp.cpu().numpy()).all()
    env.close()


def test_learner_linear_gradients_equal_autograd(G):
    """_LinearFM (chunked long-K weight gradient, matrix-vector bias gradient) vs plain addmm under autograd."""
    import torch
    from gym_reinmav_amd.ppo import MlpPolicy, _LinearFM

    torch.manual_seed(1)
    for n_in, n_out, B in ((10, 64, 4 * _LinearFM.CHUNK), (64, 64, 2 * _LinearFM.CHUNK), (64, 4, 12345)):
        x = torch.randn(n_in, B, device="cuda", requires_grad=True)
        w = torch.randn(n_out, n_in, device="cuda", requires_grad=True)
        b = torch.randn(n_out, device="cuda", requires_grad=True)
        g = torch.randn(n_out, B, device="cuda")
        _LinearFM.apply(x, w, b).backward(g)
        got = [t.grad.clone() for t in (x, w, b)]
        for t in (x, w, b):
            t.grad = None
        torch.addmm(b[:, None], w, x).backward(g)
        for a_, r_ in zip(got, (x.grad, w.grad, b.grad)):
            assert torch.allclose(a_, r_, rtol=2e-4, atol=2e-4 * float(r_.abs().max())), (n_in, n_out, B)
    # and through the whole policy: same loss gradients with and without it
    pol = MlpPolicy(10, 4).cuda()
    obs = torch.randn(10, 2 * _LinearFM.CHUNK, device="cuda")
    mean, v = pol(obs)
    (mean.square().mean() + v.square().mean()).backward()
    g1 = [p.grad.clone() for p in pol.parameters() if p.grad is not None]
    pol.zero_grad()
    orig = _LinearFM.apply
    try:
        _LinearFM.apply = staticmethod(lambda x, w, b: torch.addmm(b[:, None], w, x))
        mean, v = pol(obs)
        (mean.square().mean() + v.square().mean()).backward()
    finally:
        _LinearFM.apply = orig
    g2 = [p.grad for p in pol.parameters() if p.grad is not None]
    assert len(g1) == len(g2) > 0
    for a_, r_ in zip(g1, g2):
        assert torch.allclose(a_, r_, rtol=2e-4, atol=2e-5 * float(r_.abs().max()) + 1e-9)


def test_ppo_learner_runs_and_fits_values(G):
    import torch
    from gym_reinmav_amd.ppo import PPO, MlpPolicy, RolloutCollector

    torch.manual_seed(0)
    env = G.BatchedQuadrotor("quad3d", 4096, seed=0)
    pol = MlpPolicy(env.nS, env.nA).cuda()
    ro = RolloutCollector(env, pol, 32, graph=True)
    ppo = PPO(pol)
    hist = []
    for it in range(12):
        ro.collect()
        hist.append(ppo.update(ro))
        ro.roll_over()
    assert all(np.isfinite(list(h.values())).all() for h in hist)
    # the value net starts uncorrelated with the returns and must pick them up
    assert hist[-1]["explained_variance"] > max(0.02, hist[0]["explained_variance"])
    assert all(h["ratio_max"] < 5.0 for h in hist)
    env.close()


def _predicted_noise(seed, env_ids, t):
    """Box-Muller over the Philox stream tag 3 (csrc/rmav_policy.hpp), in float32 like the kernel."""
    z = np.zeros((len(env_ids), 4), np.float32)
    for i, e in enumerate(env_ids):
        r = O.philox([e & 0xFFFFFFFF, e >> 32, t & 0xFFFFFFFF, (3 << 24) | (((t >> 32) & 0xFFFF) << 8)],
                     [seed & 0xFFFFFFFF, seed >> 32])
        for p in range(2):
            u1 = np.float32((int(r[2 * p]) >> 8) + 1) * np.float32(1.0 / 16777216.0)
            u2 = np.float32(int(r[2 * p + 1]) >> 8) * np.float32(1.0 / 16777216.0)
            rad = np.sqrt(np.float32(-2.0) * np.log(u1, dtype=np.float32), dtype=np.float32)
            z[i, 2 * p] = rad * np.cos(2 * np.pi * float(u2))
            z[i, 2 * p + 1] = rad * np.sin(2 * np.pi * float(u2))
    return z


@pytest.mark.parametrize("f32_mfma", [False, True])
@pytest.mark.parametrize("kind", ["quad3d", "quad3d_sl", "quad2d", "quad2d_sl"])
def test_fused_policy_rollout_matches_torch_policy_and_oracle(G, kind, f32_mfma):
    """rmav_rollout_policy (fp32 on the vector ALU, and fp32 on v_mfma_f32_32x32x2_f32 - partial wavefronts included):
    in-kernel MLP == torch MlpPolicy (fp32, tol 2e-5 on means / values), the sampled
    action is mean + std * (spec'd Philox/Box-Muller normal), logp is consistent, and every env step agrees
    with the oracle."""
    import torch
    from gym_reinmav_amd.ppo import FusedPolicyCollector, MlpPolicy

    torch.manual_seed(2)
    N, T, seed, base = 512, 12, 21, 1000
    env = G.BatchedQuadrotor(kind, N, seed=seed, env_id_base=base)
    pol = MlpPolicy(env.nS, env.nA, init_logstd=0.7).cuda()
    with torch.no_grad():  # make the heads non-trivial (default init has gain 0.01 on the action head)
        for net in (pol.pi, pol.vf):
            net[2].weight.mul_(30.0 if net is pol.pi else 1.0)
            net[2].bias.uniform_(-0.5, 0.5)
        pol.logstd.copy_(torch.linspace(-0.5, 0.7, env.nA))
    ro = FusedPolicyCollector(env, pol, T, f32_mfma=f32_mfma)   # both fp32 actors: VALU, and fp32-input MFMA
    rc = env.get_reset_counts()
    t0 = env.step_count
    for it in range(3):
        ro.collect()
        rc = _check_rollout(kind, seed, ro, rc, base)   # env side vs oracle, incl. auto-reset states
        with torch.no_grad():
            obs = ro.obs[:T].permute(1, 0, 2).reshape(env.nS, -1)
            mean, val = pol(obs)
            mean, val = mean.reshape(env.nA, T, N), val.reshape(T, N)
            v_last = pol(ro.obs[T])[1]
        assert (ro.val[:T] - val).abs().max() < 2e-5 * max(1.0, float(val.abs().max()))
        assert (ro.val[T] - v_last).abs().max() < 2e-5 * max(1.0, float(v_last.abs().max()))
        std = torch.exp(pol.logstd)[:, None, None]
        z = ((ro.act.permute(1, 0, 2) - mean) / std)            # implied noise [nA, T, N]
        logp_ref = -0.5 * (z * z).sum(0) - pol.logstd.sum() - 0.5 * env.nA * np.log(2 * np.pi)
        assert (ro.logp - logp_ref).abs().max() < 2e-3
        zc = z.detach().cpu().numpy()
        assert abs(zc.mean()) < 0.08 and abs(zc.var() - 1.0) < 0.12 and np.abs(zc).max() < 6.5
        for t in (0, T - 1):                                    # exact noise spec on a subset of envs
            ids = np.arange(0, N, 37)
            zp = _predicted_noise(seed, base + ids, t0 + it * T + t)[:, :env.nA]
            assert np.abs(zc[:, t, ids].T - zp).max() < 2e-4 * 30
        ro.roll_over()
    assert env.step_count == t0 + 3 * T
    env.close()


def test_fused_policy_learner_loop(G):
    import torch
    from gym_reinmav_amd.ppo import PPO, FusedPolicyCollector, MlpPolicy

    torch.manual_seed(0)
    env = G.BatchedQuadrotor("quad3d", 8192, seed=0)
    pol = MlpPolicy(env.nS, env.nA).cuda()
    ro = FusedPolicyCollector(env, pol, 32)
    ppo = PPO(pol)
    hist = []
    for it in range(10):
        ro.collect()
        hist.append(ppo.update(ro))
        ro.roll_over()
    assert all(np.isfinite(list(h.values())).all() for h in hist)
    assert hist[-1]["explained_variance"] > max(0.02, hist[0]["explained_variance"])
    assert all(h["ratio_max"] < 5.0 for h in hist)   # rollout logp (kernel) and learner logp (torch) agree
    env.close()


@pytest.mark.parametrize("kind", ["quad2d", "quad2d_sl", "quad3d", "quad3d_sl", "reinmav"])
def test_native_weight_pack_equals_the_torch_pack(G, kind):
    """rmav_pack_policy (one gather launch) writes bit for bit what the torch chain of _PolicyPacker.pack writes, for the three
    weight layouts (fp32 VALU, fp32-input MFMA, bf16 MFMA fragments incl. the round-to-nearest-even bf16 conversion) and
    rmav_pack_policy_f16 for the f16 fragments (scaled by -2k / -2 in fp32, then rounded once), and follows in-place
    parameter updates."""
    import torch
    from gym_reinmav_amd.ppo import MlpPolicy, _PolicyPacker

    torch.manual_seed(11)
    env = G.BatchedQuadrotor(kind, 64, seed=1)
    pol = MlpPolicy(env.nS, env.nA, init_logstd=-0.7).cuda()
    with torch.no_grad():
        for prm in pol.parameters():
            prm.add_(torch.randn_like(prm) * 0.3)
    for bf16, f32m, f16 in ((False, False, False), (False, True, False), (True, False, False), (False, False, True)):
        pk = _PolicyPacker(pol, env.nS, bf16, f32_mfma=f32m, f16_mfma=f16)
        for rnd in range(2):
            ref = pk.pack()
            out = torch.full_like(ref, float("nan"))
            pk.pack_native(env, out)
            env.sync()
            torch.cuda.synchronize()
            assert torch.equal(out.view(torch.int32), ref.view(torch.int32)), (bf16, f32m, f16, rnd)
            with torch.no_grad():                       # an optimiser step updates in place: the next pack must see it
                for prm in pol.parameters():
                    prm.mul_(1.01).add_(0.003)
    spol = MlpPolicy(env.nS, env.nA, init_logstd=-0.7, value_network="shared").cuda()   # one trunk, two heads: ONE net of fragments
    with torch.no_grad():
        for prm in spol.parameters():
            prm.add_(torch.randn_like(prm) * 0.3)
    pk = _PolicyPacker(spol, env.nS, False, f16_mfma=True)
    ref = pk.pack()
    out = torch.full_like(ref, float("nan"))
    pk.pack_native(env, out)
    env.sync()
    torch.cuda.synchronize()
    assert ref.numel() == G._abi.lib().rmav_policy_weight_count_shared() and torch.equal(out.view(torch.int32), ref.view(torch.int32))
    env.close()


# tolerance of the matrix-core actors against the fp32 torch policy, relative to max(1, |y|): bf16 operands (8-bit mantissa)
# 3e-2; f16 operands (11 bits) with tanh folded into the next layer 4e-3
ACTOR_TOL = {"bf16": 3e-2, "bf16_1w": 3e-2, "f16": 4e-3, "f16_shared": 4e-3}


def _policy_for(actor, n_obs, n_act, **kw):
    """f16_shared: ONE trunk with a mean head and a value head (baselines' value_network='shared'); the others: two nets."""
    from gym_reinmav_amd.ppo import MlpPolicy

    return MlpPolicy(n_obs, n_act, value_network=("shared" if actor == "f16_shared" else "copy"), **kw)


def _mfma_collector(G, env, pol, T, actor):
    from gym_reinmav_amd.ppo import FusedPolicyCollector

    if actor == "bf16_1w":   # round 3's one-wavefront-per-64-envs kernel (kept selectable)
        env.set_tuning(policy_pair=0)
    return FusedPolicyCollector(env, pol, T, bf16_mfma=actor.startswith("bf16"), f16_mfma=actor.startswith("f16"))


@pytest.mark.parametrize("actor", ["bf16", "bf16_1w", "f16", "f16_shared"])
@pytest.mark.parametrize("kind,n", [("quad3d", 512), ("quad3d_sl", 300), ("quad2d", 131), ("quad2d_sl", 65), ("reinmav", 64), ("quad3d", 1)])
def test_bf16_mfma_actor_matches_fp32_policy(G, kind, n, actor):
    """RMAV_POLICY_BF16_MFMA / RMAV_POLICY_F16_MFMA: the same two nets on the matrix cores (bf16 / f16 operands, fp32
    accumulate; as an (actor, critic) wavefront pair, and bf16 also as one wavefront).  Means and
    values must agree with the fp32 torch policy to the operand accuracy for EVERY env of full, partial and
    single wavefronts (the inter-lane exchange and the fragment packing are what this checks); the env side
    and the noise spec are the same code as the fp32 mode."""
    import torch
    from gym_reinmav_amd.ppo import FusedPolicyCollector, MlpPolicy

    torch.manual_seed(5)
    T, seed = 6, 33
    env = G.BatchedQuadrotor(kind, n, seed=seed)
    if kind == "reinmav":   # spread the envs out (they all start from the same init state)
        s0 = env.get_state() + np.random.RandomState(0).normal(scale=0.1, size=(n, 13)).astype(np.float32)
        env.set_state(s0)
    pol = _policy_for(actor, env.nS, env.nA, init_logstd=-1.0).cuda()
    with torch.no_grad():
        for net in (pol.pi, pol.vf):
            net[-1].weight.mul_(20.0 if net is pol.pi else 1.0)
            for lin in net:
                lin.bias.uniform_(-0.3, 0.3)
        pol.logstd.copy_(torch.linspace(-1.2, -0.4, env.nA))
    ro = _mfma_collector(G, env, pol, T, actor)
    tol = ACTOR_TOL[actor]
    t0 = env.step_count
    rc0 = env.get_reset_counts()
    ro.collect()
    torch.cuda.synchronize()
    if kind != "reinmav":
        _check_rollout(kind, seed, ro, rc0)       # env side vs the oracle from the recorded (obs, action)
    with torch.no_grad():
        obs = ro.obs[:T].permute(1, 0, 2).reshape(env.nS, -1)
        mean, val = pol(obs)
        mean, val = mean.reshape(env.nA, T, n), val.reshape(T, n)
        v_last = pol(ro.obs[T])[1]
        std = torch.exp(pol.logstd)[:, None, None]
    scale_m = max(1.0, float(mean.abs().max()))
    scale_v = max(1.0, float(val.abs().max()))
    assert (ro.val[:T] - val).abs().max() < tol * scale_v
    assert (ro.val[T] - v_last).abs().max() < tol * scale_v
    # implied noise must match the Philox/Box-Muller spec up to the bf16 error of the mean
    z = ((ro.act.permute(1, 0, 2) - mean) / std).cpu().numpy()
    for t in (0, T - 1):
        zp = _predicted_noise(seed, np.arange(n), t0 + t)[:, :env.nA]
        assert np.abs(z[:, t, :].T - zp).max() < tol * scale_m / float(std.min())
    logp_ref = -0.5 * torch.from_numpy(_predicted_noise(seed, np.arange(n), t0)[:, :env.nA] ** 2).sum(1) \
        - float(pol.logstd.detach().sum()) - 0.5 * env.nA * np.log(2 * np.pi)
    assert (ro.logp[0].cpu() - logp_ref).abs().max() < 1e-3
    assert np.array_equal(env.get_state(layout="soa"), ro.obs[T].cpu().numpy())
    env.close()


@pytest.mark.parametrize("kind,n", [("quad3d", 4096 + 77), ("quad2d_sl", 300)])
def test_bf16_pair_kernel_equals_the_one_wavefront_kernel(G, kind, n):
    """The (actor, critic) wavefront pair computes each net with the one-wavefront kernel's instruction sequence and draws the
    same Philox counters: every output of a rollout is bit-identical, incl. episode statistics and the state left behind."""
    import torch
    from gym_reinmav_amd.ppo import MlpPolicy

    torch.manual_seed(9)
    T, seed = 40, 8
    outs = []
    pol = None
    for actor in ("bf16", "bf16_1w"):
        env = G.BatchedQuadrotor(kind, n, seed=seed, track_episodes=True)
        if pol is None:
            pol = MlpPolicy(env.nS, env.nA, init_logstd=0.8).cuda()
            with torch.no_grad():
                pol.pi[2].weight.mul_(30.0)
        ro = _mfma_collector(G, env, pol, T, actor)
        for it in range(2):
            ro.collect()
            ro.roll_over()
        torch.cuda.synchronize()
        eb = env.episode_buffers()
        tot = env.episode_totals()
        outs.append([t.clone() for t in (ro.obs, ro.act, ro.rew, ro.done, ro.logp, ro.val)] +
                    [torch.from_numpy(np.asarray(x).astype(np.float64)) for x in (env.get_state(), env.get_sbd(), env.get_reset_counts(), eb["last_return"],
                                                                                   eb["last_length"], eb["cur_return"], eb["cur_length"])] + [tot])
        env.close()
    assert int(outs[0][3].sum()) > 0
    for x, y in zip(outs[0][:-1], outs[1][:-1]):
        assert torch.equal(x.cpu(), y.cpu())
    assert outs[0][-1] == outs[1][-1]


@pytest.mark.parametrize("actor,n,tune", [("bf16", 65536, {"pair_group": 1}), ("bf16", 65536, {"pair_group": 2}), ("bf16", 65536, {"pair_group": 4}),
                                           ("f16", 65536, {"pair_group": 2}), ("f16", 131072, {"pair_group": 2}), ("f16", 131072, {"pair_group": 4}),
                                           ("f16_shared", 65536, {"pair_group": 1}), ("f16_shared", 65536, {"pair_group": 4}), ("f16_shared", 131072, {"pair_group": 2}),
                                           ("bf16_1w", 131072, {}), ("bf16_1w", 262144, {}), ("fp32_mfma", 65536, {}), ("fp32_mfma", 131072, {})])
def test_matrix_core_actors_are_deterministic(G, actor, n, tune):
    """Every matrix-core actor, at 2 - 4 wavefronts per SIMD, five 32-step rollouts from the same state: bit-identical outputs.
    (Round 4 found ~1 - 25 % of the wavefronts of such launches with wrong physics in lanes 48..63; round 5's root cause: a
    compiler-made packed-fp32 instruction with op_sel on src1 reads zero there while a 32x32x16 MFMA executes on the SIMD -
    profiles/r05/packed_f32_hazard.md.  One wavefront per SIMD never showed it, so the parity tests at 65 536 envs could not.)"""
    import torch
    from gym_reinmav_amd.ppo import FusedPolicyCollector, MlpPolicy

    torch.manual_seed(4)
    kind, T, seed = "quad3d", 32, 17
    pol, ref = None, None
    for rep in range(5):
        env = G.BatchedQuadrotor(kind, n, seed=seed, track_episodes=True)
        if tune:
            env.set_tuning(**tune)
        if pol is None:
            pol = _policy_for(actor, env.nS, env.nA, init_logstd=0.5).cuda()
            with torch.no_grad():
                pol.pi[2].weight.mul_(30.0)
                pol.pi[2].bias.uniform_(0.5, 4.0)
        ro = (FusedPolicyCollector(env, pol, T, f32_mfma=True) if actor == "fp32_mfma" else _mfma_collector(G, env, pol, T, actor))
        ro.collect()
        torch.cuda.synchronize()
        cur = [getattr(ro, k).clone() for k in ("obs", "act", "rew", "done", "logp", "val")] + [env.get_state(layout="soa", device_out=True).clone()]
        tot = env.episode_totals()
        env.close()
        if ref is None:
            ref, tot0 = cur, tot
            continue
        for x, y in zip(cur, ref):
            bad = (x != y)
            assert not bool(bad.any()), (rep, int(bad.sum()), sorted(set((bad.nonzero()[:, -1] % 64).tolist()))[:4])
        assert tot == tot0


@pytest.mark.parametrize("kind,n,mode", [("quad3d", 65536, "random"), ("quad3d", 262144, "random"), ("quad3d", 131072, "controller"),
                                         ("quad3d_sl", 65536, "random"), ("quad2d", 65536, "random")])
def test_mfma_free_rollouts_are_bit_stable_beside_matrix_core_work(G, kind, n, mode):
    """The hazard of profiles/r05/packed_f32_hazard.md needs a 32x32x16 MFMA on the SIMD - from ANY wavefront, also another stream's:
    with the SLP-vectorised build of rounds 1 - 4 the MFMA-free quadrotor3d rollout (the headline kernel) lost its x-axis thrust term
    in 12 - 46 wavefronts per launch at 65 536 envs and 500 - 900 at 262 144 while the f16 policy rollout of another env ran on a
    second stream (hazard_concurrent_streams.txt).  Fused rollouts alone give the reference bits; the same rollouts with ~6 ms of
    matrix-core work queued on the other stream must give exactly those."""
    import torch
    from gym_reinmav_amd.ppo import FusedPolicyCollector, MlpPolicy

    T, want = 64, ("actions", "obs", "rew", "done")
    s_env, s_mfma = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(s_mfma):
        penv = G.BatchedQuadrotor("quad3d", 65536, seed=3)
        torch.manual_seed(1)
        ro = FusedPolicyCollector(penv, MlpPolicy(penv.nS, penv.nA).cuda(), 32, f16_mfma=True)
        ro.collect()
    torch.cuda.synchronize()

    def run(beside):
        with torch.cuda.stream(s_env):
            env = G.BatchedQuadrotor(kind, n, seed=7)
            if beside:
                with torch.cuda.stream(s_mfma):
                    for _ in range(40):
                        ro.collect()
            tr = env.rollout(T, mode=mode, layout="soa", want=want, device_out=True)
            tr = env.rollout(T, mode=mode, layout="soa", want=want, device_out=True, out=tr)
            s_env.synchronize()
            out = [tr[k].clone() for k in want] + [env.get_state(layout="soa", device_out=True).clone()]
            env.close()
        torch.cuda.synchronize()
        return out

    ref = run(False)
    for rep in range(4):
        for x, y in zip(run(True), ref):
            bad = x != y
            assert not bool(bad.any()), (rep, int(bad.sum()), sorted(set((bad.nonzero()[:, -1] % 64).tolist()))[:4])
    with torch.cuda.stream(s_mfma):
        penv.close()


@pytest.mark.parametrize("actor", ["fp32", "fp32_mfma", "bf16", "f16", "f16_shared"])
def test_c5_size_policy_rollout(G, actor):
    """BASELINE configs[4] (C5)'s per-GPU shard at full size: quadrotor3d-v0, 65 536 envs x 32-step rollouts with the
    policy inside the kernel (fp32 and bf16-MFMA actors).  Every env step of a 4 096-env sample is checked against
    the oracle from the recorded (obs, action) - including the auto-reset states - and the values / log-probs of the
    whole batch against the torch policy; the GAE pass over the [32][65 536] result is checked against the torch loop."""
    import torch
    from gym_reinmav_amd.ppo import FusedPolicyCollector, MlpPolicy, gae

    torch.manual_seed(4)
    kind, N, T, seed = "quad3d", 65536, 32, 17
    env = G.BatchedQuadrotor(kind, N, seed=seed)
    pol = _policy_for(actor, env.nS, env.nA, init_logstd=0.5).cuda()
    with torch.no_grad():
        pol.pi[2].weight.mul_(30.0)
        pol.pi[2].bias.uniform_(0.5, 4.0)          # thrust around hover, so episodes last a while and still end
        pol.vf[-1].bias.uniform_(-0.5, 0.5)
    bf16 = actor in ("bf16", "f16", "f16_shared")   # the reduced-precision actors
    ro = FusedPolicyCollector(env, pol, T, bf16_mfma=(actor == "bf16"), f32_mfma=(actor == "fp32_mfma"), f16_mfma=actor.startswith("f16"))
    assert ro.f32_mfma == (actor == "fp32_mfma") and ro.bf16_mfma == (actor == "bf16") and ro.f16_mfma == actor.startswith("f16")
    sample = np.arange(0, N, 16)                    # 4 096 envs, every wavefront represented
    rc = env.get_reset_counts()
    for it in range(2):
        ro.collect()
        torch.cuda.synchronize()
        obs, act = ro.obs[:, :, sample].cpu().numpy(), ro.act[:, :, sample].cpu().numpy()
        rew, done = ro.rew[:, sample].cpu().numpy(), ro.done[:, sample].cpu().numpy().astype(bool)
        rcs = rc[sample].copy()
        for t in range(T):
            o2, r, d, _ = O.batch_step(kind, obs[t].T.astype(np.float64), act[t].T.astype(np.float64))
            ok = near_threshold(kind, o2)
            assert np.array_equal(done[t] | ok, d | ok)
            alive = ~done[t] & ~d
            assert scaled_err(obs[t + 1].T[alive], o2[alive]).max() <= TOL
            assert scaled_err(rew[t][alive], r[alive]).max() <= TOL
            if done[t].any():
                assert np.array_equal(obs[t + 1].T[done[t]], O.reset_states(kind, seed, sample[done[t]], rcs[done[t]]))
            rcs += done[t].astype(np.uint32)
        rc = rc + ro.done.sum(0).cpu().numpy().astype(np.uint32)
        assert np.array_equal(env.get_reset_counts(), rc)
        with torch.no_grad():
            mean, val = pol(ro.obs[:T].permute(1, 0, 2).reshape(env.nS, -1))
            mean, val = mean.reshape(env.nA, T, N), val.reshape(T, N)
            v_last = pol(ro.obs[T])[1]
            std = torch.exp(pol.logstd)[:, None, None]
            tol = (ACTOR_TOL[actor] if bf16 else 2e-5) * max(1.0, float(val.abs().max()))
            assert (ro.val[:T] - val).abs().max() < tol and (ro.val[T] - v_last).abs().max() < tol
            if not bf16:
                z = (ro.act.permute(1, 0, 2) - mean) / std
                logp_ref = -0.5 * (z * z).sum(0) - pol.logstd.sum() - 0.5 * env.nA * np.log(2 * np.pi)
                assert (ro.logp - logp_ref).abs().max() < 2e-3
            assert bool(torch.isfinite(ro.logp).all())
            # GAE kernel over the full-size trajectory vs the torch fp32 loop
            adv, ret = env.gae(ro.rew, ro.done, ro.val, 0.99, 0.95)
            adv_t, ret_t = gae(ro.rew, ro.val, ro.done, 0.99, 0.95)
            scale = max(1.0, float(adv_t.abs().max()))
            assert (adv - adv_t).abs().max() < 1e-5 * scale and (ret - ret_t).abs().max() < 1e-5 * scale
        ro.roll_over()
    assert int(ro.done.sum()) > 0
    env.close()


@pytest.mark.parametrize("kind,n", [("quad3d", 512), ("quad3d_sl", 300), ("quad2d", 131), ("quad2d_sl", 65), ("reinmav", 64), ("quad3d", 1)])
def test_f32_mfma_actor_equals_valu_actor(G, kind, n):
    """RMAV_POLICY_FP32_MFMA vs RMAV_POLICY_FP32 from the same state and weights: same noise stream, means / values /
    log-probs equal to fp32 summation-order accuracy for EVERY env of full, partial and single-lane wavefronts (the
    inter-lane exchange, the fragment packing and the clone lanes are what this checks)."""
    import torch
    from gym_reinmav_amd.ppo import FusedPolicyCollector, MlpPolicy

    torch.manual_seed(7)
    T, seed = 5, 12
    outs = []
    pol = None
    for f32m in (False, True):
        env = G.BatchedQuadrotor(kind, n, seed=seed)
        if kind == "reinmav":
            env.set_state(env.get_state() + np.random.RandomState(0).normal(scale=0.1, size=(n, 13)).astype(np.float32))
        if pol is None:
            pol = MlpPolicy(env.nS, env.nA, init_logstd=-0.5).cuda()
            with torch.no_grad():
                for net in (pol.pi, pol.vf):
                    net[2].weight.mul_(20.0 if net is pol.pi else 1.0)
                    for lin in net:
                        lin.bias.uniform_(-0.3, 0.3)
        ro = FusedPolicyCollector(env, pol, T, f32_mfma=f32m)
        ro.collect()
        torch.cuda.synchronize()
        outs.append((ro.val.clone(), ro.logp.clone(), ro.act.clone(), ro.obs.clone()))
        env.close()
    (v0, l0, a0, o0), (v1, l1, a1, o1) = outs
    sv = max(1.0, float(v0.abs().max()))
    # step 0 starts from identical states: the two actors must agree to fp32 round-off there; later steps diverge only
    # through that round-off acting on the dynamics
    assert (v0[0] - v1[0]).abs().max() < 2e-5 * sv
    assert (a0[0] - a1[0]).abs().max() < 2e-5 * max(1.0, float(a0[0].abs().max()))
    assert (l0[0] - l1[0]).abs().max() < 1e-4
    assert (v0 - v1).abs().max() < 1e-3 * sv and (o0 - o1).abs().max() < 1e-3 * max(1.0, float(o0.abs().max()))
