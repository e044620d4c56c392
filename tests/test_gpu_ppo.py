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


def _check_rollout(kind, seed, ro, rc0):
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
            assert np.array_equal(obs[t + 1].T[done[t]], O.reset_states(kind, seed, np.nonzero(done[t])[0], rc[done[t]]))
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
