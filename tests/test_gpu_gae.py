"""rmav_gae / rmav_normalize (the advantage pass of the PPO2 caller) against a float64 per-env
recursion and the plain torch fp32 loop.  Floating point: tolerance 1e-5 * max(1, |A|max) (fp32 FMAs, T <= 2048)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G(built):
    import torch

    assert torch.cuda.is_available()
    import gym_reinmav_amd as g

    return g


def _ref_f64(rew, val, done, gamma, lam, scale):
    T, N = rew.shape
    adv = np.zeros((T, N))
    last = np.zeros(N)
    for t in reversed(range(T)):
        nt = 1.0 - done[t].astype(np.float64)
        delta = scale * rew[t].astype(np.float64) + gamma * val[t + 1].astype(np.float64) * nt - val[t]
        last = delta + gamma * lam * nt * last
        adv[t] = last
    return adv, adv + val[:T]


@pytest.mark.parametrize("N,T", [(1, 1), (63, 7), (64, 8), (1000, 17), (20037, 33), (65536, 32), (4096, 257)])
def test_gae_matches_float64_recursion_and_torch(G, N, T):
    import torch
    from gym_reinmav_amd.ppo import gae

    rng = np.random.RandomState(N + T)
    gamma, lam, scale = 0.99, 0.95, 0.5
    rew = rng.normal(size=(T, N)).astype(np.float32)
    val = rng.normal(size=(T + 1, N)).astype(np.float32)
    done = (rng.uniform(size=(T, N)) < 0.07).astype(np.uint8)
    env = G.BatchedQuadrotor("quad3d", N, track_episodes=False)
    r, v, d = torch.from_numpy(rew).cuda(), torch.from_numpy(val).cuda(), torch.from_numpy(done).cuda()
    sums = torch.zeros(2, dtype=torch.float64, device="cuda")
    adv, ret = env.gae(r, d, v, gamma, lam, reward_scale=scale, sums=sums)
    exp_a, exp_r = _ref_f64(rew, val, done, gamma, lam, scale)
    tol = 1e-5 * max(1.0, np.abs(exp_a).max())
    assert np.abs(adv.cpu().numpy() - exp_a).max() < tol
    assert np.abs(ret.cpu().numpy() - exp_r).max() < tol
    adv_t, ret_t = gae(r * scale, v, d, gamma, lam)       # plain torch fp32 reference of the same op
    assert (adv - adv_t).abs().max() < tol and (ret - ret_t).abs().max() < tol
    s = sums.cpu().numpy()
    assert abs(s[0] - exp_a.sum()) < 1e-4 * max(1.0, np.abs(exp_a).sum())
    assert abs(s[1] - (exp_a ** 2).sum()) < 1e-4 * max(1.0, (exp_a ** 2).sum())
    # advantage normalisation in place
    mean = s[0] / (T * N)
    var = max(s[1] / (T * N) - mean * mean, 0.0)
    rstd = 1.0 / (np.sqrt(var) + 1e-8)
    a2 = adv.clone()
    env.normalize_(a2, mean, rstd)
    assert (a2 - (adv - float(mean)) * float(rstd)).abs().max() < 1e-5 * max(1.0, float(a2.abs().max()))
    if T * N > 1000:
        assert abs(float(a2.mean())) < 1e-3 and abs(float(a2.std()) - 1.0) < 1e-2
    env.close()


def test_gae_without_sums_and_episode_boundaries(G):
    """done_t = 1 cuts both the bootstrap and the recursion: the advantages before a boundary do not depend on
    anything after it."""
    import torch

    N, T = 257, 12
    rng = np.random.RandomState(0)
    rew = torch.from_numpy(rng.normal(size=(T, N)).astype(np.float32)).cuda()
    val = torch.from_numpy(rng.normal(size=(T + 1, N)).astype(np.float32)).cuda()
    done = torch.zeros((T, N), dtype=torch.uint8, device="cuda")
    done[5] = 1
    env = G.BatchedQuadrotor("quad2d", N, track_episodes=False)
    a1, _ = env.gae(rew, done, val)
    rew2, val2 = rew.clone(), val.clone()
    rew2[6:] += 3.0
    val2[6:] -= 2.0
    a2, _ = env.gae(rew2, done, val2)
    assert torch.equal(a1[:6], a2[:6]) and not torch.equal(a1[6:], a2[6:])
    env.close()
