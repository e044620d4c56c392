"""Host-checkable parts of the PPO2-style caller: GAE and the feature-major policy."""
import math

import numpy as np
import torch


def test_gae_matches_reference_loop():
    from gym_reinmav_amd.ppo import gae

    rng = np.random.RandomState(0)
    T, N, gamma, lam = 17, 33, 0.99, 0.95
    rew = rng.normal(size=(T, N)).astype(np.float32)
    val = rng.normal(size=(T + 1, N)).astype(np.float32)
    done = (rng.uniform(size=(T, N)) < 0.1).astype(np.uint8)
    adv, ret = gae(torch.from_numpy(rew), torch.from_numpy(val), torch.from_numpy(done), gamma, lam)
    exp = np.zeros((T, N))
    for e in range(N):  # plain per-env recursion (baselines Runner.run, with done[t] = "episode ended at t")
        last = 0.0
        for t in reversed(range(T)):
            nt = 1.0 - done[t, e]
            delta = rew[t, e] + gamma * val[t + 1, e] * nt - val[t, e]
            last = delta + gamma * lam * nt * last
            exp[t, e] = last
    assert np.abs(adv.numpy() - exp).max() < 1e-4
    assert np.abs(ret.numpy() - (exp + val[:T])).max() < 1e-4


def test_feature_major_policy_equals_batch_major():
    from gym_reinmav_amd.ppo import MlpPolicy

    torch.manual_seed(0)
    pol = MlpPolicy(10, 4)
    x = torch.randn(257, 10)
    mean, v = pol(x.t().contiguous())
    h = x
    for i, lin in enumerate(pol.pi):
        h = lin(h)
        h = torch.tanh(h) if i < 2 else h
    assert torch.allclose(mean.t(), h, atol=1e-5)
    act = mean + 0.3 * torch.randn_like(mean)
    lp = pol.log_prob(mean, act)
    ref = torch.distributions.Normal(mean.t(), torch.exp(pol.logstd)).log_prob(act.t()).sum(-1)
    assert torch.allclose(lp, ref, atol=1e-4)
    assert abs(float(pol.entropy()) - 4 * (0.5 * math.log(2 * math.pi * math.e))) < 1e-5
    assert v.shape == (257,)
