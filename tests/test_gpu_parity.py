"""Parity of the HIP path (through the C ABI) against the CPU oracle and the reference's golden vectors.

Bar (BASELINE.json north_star): |d| <= 1e-6 * max(1, |y_ref|) on state and reward for identical
fp32-representable (state, action); `done` exact except when a terminating norm is within 1e-5 of its
limit; RNG-derived values (reset states, random actions) bit-exact.
"""
import os

import numpy as np
import pytest

import oracle as O
from util import BOX, CTRL_TOL, KINDS, NA, NS, TOL, near_threshold, random_cases, scaled_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def G(built):
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import gym_reinmav_amd as g

    return g


def _check_step(kind, s_in, a_in, obs, rew, done, sbd_in=None, limits=None, params=None):
    """Compare one batched device step (no auto-reset) with the oracle from the same inputs."""
    o2, r, d, sbd = O.batch_step(kind, np.asarray(s_in, np.float64), np.asarray(a_in, np.float64), sbd_in, params=params)
    done = np.asarray(done).astype(bool)
    assert scaled_err(obs, o2).max() <= TOL
    ok = near_threshold(kind, o2, limits=limits)
    assert np.array_equal(done | ok, d | ok)
    same = done == d
    assert scaled_err(np.asarray(rew)[same], r[same]).max() <= TOL
    return o2, r, d, sbd


@pytest.mark.parametrize("kind", KINDS)
def test_step_vs_reference_golden(G, kind, golden):
    g = golden[kind]
    n = len(g["step_s"])
    env = G.BatchedQuadrotor(kind, n, auto_reset=False, track_episodes=False)
    env.set_state(g["step_s"].astype(np.float32))
    obs, rew, done = env.step(g["step_a"].astype(np.float32))
    assert scaled_err(obs, g["step_s2"]).max() <= TOL
    ok = near_threshold(kind, g["step_s2"])
    assert np.array_equal(done | ok, g["step_d"] | ok)
    same = done == g["step_d"]
    assert scaled_err(rew[same], g["step_r"][same]).max() <= TOL
    assert (rew[done] == 1.0).all()  # first termination of each env's lifetime
    assert np.array_equal(env.get_sbd(), np.where(done, 0, -1))
    assert np.array_equal(env.get_state(), obs)
    env.close()


def test_quad2d_reading_A_vs_golden(G, golden):
    g = golden["quad2d"]
    env = G.BatchedQuadrotor("quad2d", len(g["step_s"]), auto_reset=False, track_episodes=False, reading_2d="A")
    env.set_state(g["step_s"].astype(np.float32))
    obs, rew, done = env.step(g["step_a"].astype(np.float32))
    ok = near_threshold("quad2d", g["step_s2"], limits=(3.0, 10.0))
    assert np.array_equal(done | ok, g["step_d_A"] | ok)
    same = done == g["step_d_A"]
    assert scaled_err(rew[same], g["step_r_A"][same]).max() <= TOL
    env.close()


@pytest.mark.parametrize("layout", ["aos", "soa"])
@pytest.mark.parametrize("kind", KINDS)
def test_step_vs_oracle_random_device_buffers(G, kind, layout):
    import torch

    n = 1000000 + 37  # ragged: not a multiple of the 256-thread block or the 64-lane wave
    s, a = random_cases(kind, n, seed=101)
    env = G.BatchedQuadrotor(kind, n, auto_reset=False, track_episodes=False)
    env.set_state(s)
    a_dev = torch.from_numpy(a if layout == "aos" else np.ascontiguousarray(a.T)).cuda()
    obs, rew, done = env.step(a_dev, layout=layout)
    torch.cuda.synchronize()
    obs = obs.cpu().numpy()
    obs = obs if layout == "aos" else obs.T
    _check_step(kind, s, a, obs, rew.cpu().numpy(), done.cpu().numpy())
    # host-pointer path gives the same bits
    env.set_state(s)
    env.set_sbd(np.full(n, -1, np.int32))
    obs_h, rew_h, done_h = env.step(a)
    assert np.array_equal(obs_h, obs) and np.array_equal(rew_h, rew.cpu().numpy())
    assert np.array_equal(done_h, done.cpu().numpy().astype(bool))
    env.close()


@pytest.mark.parametrize("kind", KINDS)
def test_tilted_gravity_matches_the_oracle(G, kind):
    """The reference's self.g is a public VECTOR read by every step() (quadrotor3d.py:47,96-99; quadrotor2d.py:46,88) and by the
    3-D control() (:162); the 2-D control() keeps its literal (0, 9.8) (quadrotor2d.py:130).  rmav_params.g_vec carries it:
    single step, control(), and the fused controller-driven rollout against the oracle run with the same vector."""
    # first against the REFERENCE's own outputs with a tilted vector assigned to its env object (tests/golden/gravity.npz)
    gr = np.load(os.path.join(ROOT, "tests", "golden", "gravity.npz"))
    env = G.BatchedQuadrotor(kind, len(gr[kind + "_s"]), auto_reset=False, track_episodes=False)
    p = env.params
    for i, v in enumerate(gr[kind + "_g"]):
        p.g_vec[i] = float(v)
    env.params = p
    env.set_state(gr[kind + "_s"].astype(np.float32))
    assert scaled_err(env.control(), gr[kind + "_ctrl"]).max() <= CTRL_TOL
    obs, rew, done = env.step(gr[kind + "_a"].astype(np.float32))
    assert scaled_err(obs, gr[kind + "_s2"]).max() <= TOL
    okd = near_threshold(kind, gr[kind + "_s2"])
    assert np.array_equal(done | okd, gr[kind + "_d"] | okd)
    same = done == gr[kind + "_d"]
    assert scaled_err(rew[same], gr[kind + "_r"][same]).max() <= TOL
    env.close()
    # then a big batch and the fused rollout against the oracle
    n = 20000 + 11
    s, a = random_cases(kind, n, seed=77)
    dim = 2 if kind.startswith("quad2d") else 3
    gv = [1.25, -0.75, -8.5][:dim] if dim == 3 else [1.25, -8.5]
    env = G.BatchedQuadrotor(kind, n, auto_reset=False, track_episodes=False)
    p, q = env.params, O.default_params(kind)
    for i in range(dim):
        p.g_vec[i] = q.g_vec[i] = gv[i]
    env.params = p
    assert list(env.params.g_vec)[:dim] == gv
    env.set_state(s)
    ctrl = env.control()
    assert scaled_err(ctrl[::9], O.batch_control(kind, s[::9].astype(np.float64), params=q)).max() <= CTRL_TOL
    obs, rew, done = env.step(a)
    o2, *_ = _check_step(kind, s, a, obs, rew, done, params=q)
    # the default vector gives something else (the test would not notice a g_vec the kernels ignore)
    o_def, *_ = O.batch_step(kind, s.astype(np.float64), a.astype(np.float64))
    assert scaled_err(o2, o_def).max() > 1e-4
    # fused controller-driven rollout = the same single steps, teacher-forced per step against the oracle
    env.set_state(s)
    env.set_sbd(np.full(n, -1, np.int32))
    tr = env.rollout(6, mode="controller", layout="aos", want=("actions", "obs", "rew", "done"))
    prev = s.astype(np.float64)
    for k in range(6):
        act = tr["actions"][k]
        assert scaled_err(act[::9], O.batch_control(kind, prev[::9], params=q)).max() <= CTRL_TOL
        o_k, r_k, d_k, _ = O.batch_step(kind, prev, act.astype(np.float64), params=q)
        fin = np.isfinite(o_k).all(axis=1)
        if kind.endswith("_sl"):   # tether edge: the branch is decided by the last bit of a norm (see oracle_step_branch)
            fin &= np.abs(O.tether_slack(kind, prev, q)) > 1e-5
        assert scaled_err(tr["obs"][k][fin], o_k[fin]).max() <= 2e-6
        prev = tr["obs"][k].astype(np.float64)
    env.close()


@pytest.mark.parametrize("kind", KINDS)
def test_control_vs_reference_golden(G, kind, golden):
    g = golden[kind]
    env = G.BatchedQuadrotor(kind, len(g["ctrl_s"]), auto_reset=False, track_episodes=False)
    env.set_state(g["ctrl_s"].astype(np.float32))
    a = env.control()
    assert scaled_err(a, g["ctrl_a"]).max() <= CTRL_TOL
    a_soa = env.control(layout="soa")
    assert np.array_equal(a_soa.T, a)
    env.close()


@pytest.mark.parametrize("kind", KINDS)
def test_reference_closed_loop_trajectories_replayed_on_the_gpu(G, kind, golden):
    """The reference's own test loop (test/test_quadrotor3d.py:16-22 and siblings: control() -> step() -> reset() on done,
    400 steps x 4 seeds, recorded by tests/golden/make_golden.py from the imported reference) replayed on the GPU for EVERY
    kind: all 1 600 recorded pre-step states as one batch, control() against the recorded actions, step() of the recorded
    actions against the recorded next states / rewards / dones - HIP output vs the reference's output, no oracle in between.
    steps_beyond_done is set to what the reference's env object held at that point of its life (the terminal reward is 1.0 only
    for an env object's first termination: quirk Q1).  Slung-load kinds: a step that starts with |tether| - L below 2e-6 - every
    step after a taut one, the projection puts the load exactly at L - hangs on the last bit of a norm the fp32 state cannot
    carry; there the HIP step must equal the oracle's taut or slack step from the same fp32 state."""
    g = golden[kind]
    for sfx in ([""] + (["_A"] if kind == "quad2d" else [])):
        S, A_, S2, R, D = (g["traj_" + k + sfx] for k in ("s", "a", "s2", "r", "d"))
        E, T = S.shape[:2]
        n = E * T
        D = D.astype(bool)
        env = G.BatchedQuadrotor(kind, n, seed=1, auto_reset=False, track_episodes=False, reading_2d=("A" if sfx else None))
        s32 = S.reshape(n, -1).astype(np.float32)
        earlier = (np.cumsum(D, axis=1) - D).reshape(n)              # terminations of this env object before this step
        env.set_state(s32)
        env.set_sbd(np.where(earlier == 0, -1, earlier - 1).astype(np.int32))
        a = env.control()
        # the recorded state is rounded to fp32 on entry; the controllers amplify that by their gains (kp/tau ~ 50)
        assert scaled_err(a, A_.reshape(n, -1)).max() <= 1e-5, (kind, sfx)
        a32 = A_.reshape(n, -1).astype(np.float32)
        obs, rew, done = env.step(a32)
        done = np.asarray(done).astype(bool)
        s2, r, d = S2.reshape(n, -1), R.reshape(n), D.reshape(n)
        p = O.default_params(kind, "A" if sfx else "B")
        slack = O.tether_slack(kind, s32.astype(np.float64), p)
        edge = np.abs(slack) < 2e-6 if kind in ("quad2d_sl", "quad3d_sl") else np.zeros(n, bool)
        ok = ~edge
        err = scaled_err(obs, s2)
        # (fp32 inputs vs the reference's fp64 ones: 2e-6 instead of the 1e-6 of identical inputs)
        assert err[ok].max() <= 2e-6, (kind, sfx, float(err[ok].max()))
        near = near_threshold(kind, s2, limits=(p.pos_limit, p.vel_limit))
        assert np.array_equal((done | near)[ok], (d | near)[ok])
        same = ok & (done == d)
        assert scaled_err(rew[same], r[same]).max() <= 2e-6
        if kind == "quad2d":   # (its controller's thrust is scaled x10 by step(): the loop diverges and restarts - quirk Q8)
            assert int(d.sum()) >= 1 and int((earlier > 0).sum()) >= 1     # the fixture does contain terminations and second lives
        if edge.any():   # either branch of the oracle, from the very state and action the device saw
            idx = np.nonzero(edge)[0]
            best = np.full(len(idx), np.inf)
            for ft in (0, 1):
                o2 = np.stack([O.step(kind, s32[i].astype(np.float64), a32[i].astype(np.float64), None, params=p, force_taut=ft)[0] for i in idx])
                best = np.minimum(best, scaled_err(obs[idx], o2).max(axis=1))
            assert best.max() <= TOL, (kind, float(best.max()))
            assert ok.sum() > n // 4                                    # and a good part of the loop is compared with the reference directly
        env.close()


@pytest.mark.parametrize("kind", KINDS)
def test_reset_streams_bit_exact_and_shard_invariant(G, kind):
    n, seed = 3000, 77
    env = G.BatchedQuadrotor(kind, n, seed=seed)
    ids = np.arange(n)
    assert np.array_equal(env.get_state(), O.reset_states(kind, seed, ids, 0))  # constructor reset
    obs = env.reset()
    assert np.array_equal(obs, O.reset_states(kind, seed, ids, 1))
    assert np.array_equal(env.get_reset_counts(), np.full(n, 2, np.uint32))
    # two shards with global ids == one handle
    lo = G.BatchedQuadrotor(kind, 1000, seed=seed, env_id_base=0)
    hi = G.BatchedQuadrotor(kind, 2000, seed=seed, env_id_base=1000)
    full0 = O.reset_states(kind, seed, ids, 0)
    assert np.array_equal(np.concatenate([lo.get_state(), hi.get_state()]), full0)
    # 64-bit global ids
    big = G.BatchedQuadrotor(kind, 8, seed=seed, env_id_base=2**40)
    assert np.array_equal(big.get_state(), O.reset_states(kind, seed, 2**40 + np.arange(8), 0))
    # seed() rewinds the reset counters
    env.seed(5)
    assert np.array_equal(env.reset(), O.reset_states(kind, 5, ids, 0))
    for e in (env, lo, hi, big):
        e.close()


@pytest.mark.parametrize("frac", [1.0, 0.3, 0.02])
@pytest.mark.parametrize("n", [1, 3, 64, 200, 4099])
@pytest.mark.parametrize("kind", KINDS)
def test_single_step_auto_reset_any_number_of_terminations(G, kind, n, frac):
    """The single-step kernel draws reset states cooperatively (four lanes per terminating env, 16 terminations per
    Philox pass): every count of terminating lanes per wavefront - none, one, more than 16, all 64, ragged last
    wavefront, a one-env batch - must give the spec'd reset state bit for bit, twice in a row."""
    seed = 31
    rng = np.random.RandomState(n)
    env = G.BatchedQuadrotor(kind, n, seed=seed, env_id_base=5000, auto_reset=True, track_episodes=True)
    s = env.get_state() * 0.2
    kill = rng.uniform(size=n) < frac
    s[kill] += 40.0                              # every component far outside the limits: these envs terminate
    lo, hi = BOX[kind]
    for rep in range(2):
        env.set_state(s)
        rc = env.get_reset_counts()
        a = rng.uniform(lo, hi, (n, NA[kind])).astype(np.float32) * 0.1
        obs, rew, done = env.step(a)
        o2, r, d, _ = O.batch_step(kind, s.astype(np.float64), a.astype(np.float64))
        assert np.array_equal(d, kill), "test set-up: exactly the chosen envs terminate"
        assert np.array_equal(done, d)
        if (~kill).any():
            assert scaled_err(obs[~kill], o2[~kill]).max() <= TOL
        if kill.any():
            assert np.array_equal(obs[kill], O.reset_states(kind, seed, 5000 + np.nonzero(kill)[0], rc[kill]))
        assert np.array_equal(env.get_reset_counts(), rc + kill.astype(np.uint32))
        assert np.array_equal(env.get_state(), obs)
    tot = env.episode_totals()
    assert tot["episodes"] == 2 * int(kill.sum()) and tot["length_sum"] == 2 * int(kill.sum())
    env.close()


@pytest.mark.parametrize("kind", KINDS)
def test_lifetime_terminal_reward_once(G, kind, golden):
    """Q1 through the device path: rewards at the three terminations are 1.0, 0.0, 0.0 and
    steps_beyond_done follows the reference (never cleared by reset)."""
    g = golden[kind]
    env = G.BatchedQuadrotor(kind, 1, auto_reset=False, track_episodes=False)
    for s, a, s2, r, d, sb in zip(g["life_s"], g["life_a"], g["life_s2"], g["life_r"], g["life_d"], g["life_sbd"]):
        env.set_state(s.astype(np.float32)[None])
        obs, rew, done = env.step(a.astype(np.float32)[None])
        assert scaled_err(obs[0], s2).max() <= TOL and bool(done[0]) == bool(d)
        assert abs(rew[0] - r) <= TOL * max(1.0, abs(r))
        assert int(env.get_sbd()[0]) == sb
    env.close()


@pytest.mark.parametrize("T", [24, 2, 5, 7])
@pytest.mark.parametrize("mode", ["random", "controller", "buffer"])
@pytest.mark.parametrize("kind", KINDS)
def test_fused_rollout_equals_single_steps(G, kind, mode, T):
    """The fused kernel (state in registers for T steps) and T launches of the step kernel give the
    same bits, including auto-resets and episode statistics - short launches (two-wavefront kernel from 2 steps) too."""
    n, seed = 5000, 3
    acts = None
    if mode == "buffer":
        rng = np.random.RandomState(4)
        lo, hi = BOX[kind]
        acts = rng.uniform(lo, hi, (T, NA[kind], n)).astype(np.float32)
    res = []
    for fused in (True, False):
        env = G.BatchedQuadrotor(kind, n, seed=seed, auto_reset=True, track_episodes=True)
        tr = env.rollout(T, mode=mode, actions=acts, layout="soa", fused=fused, want=("actions", "obs", "rew", "done"))
        res.append((tr, env.get_state(), env.get_sbd(), env.get_reset_counts(), env.episode_totals(),
                    env.episode_buffers(), env.step_count))
        env.close()
    (t0, s0, b0, c0, e0, eb0, k0), (t1, s1, b1, c1, e1, eb1, k1) = res
    for key in ("actions", "obs", "rew", "done"):
        assert np.array_equal(t0[key], t1[key]), key
    assert np.array_equal(s0, s1) and np.array_equal(b0, b1) and np.array_equal(c0, c1)
    assert e0["episodes"] == e1["episodes"] == int(t0["done"].sum()) and e0["length_sum"] == e1["length_sum"]
    assert abs(e0["return_sum"] - e1["return_sum"]) < 1e-3
    for key in eb0:
        assert np.array_equal(eb0[key], eb1[key]), key
    assert k0 == k1 == T


@pytest.mark.parametrize("kind,n", [("quad3d_sl", 235931), ("quad2d_sl", 262144), ("quad3d", 131072 + 77), ("quad2d", 65536 + 13)])
def test_kernel_selection_variants_give_the_same_bits(G, kind, n):
    """One launch of the one-wavefront kernel, the two-wavefront kernel forced, and two rounds of it over balanced halves
    (the default for random-action slung-load batches of 1.75-2 x the capacity) write the same trajectory, state, reset
    counters and episode statistics - ragged sizes included.  The variant is an explicit per-handle override
    (rmav_set_tuning), so all of them run in this process.  The launches are 23, 64 and 24 steps long: the second and third
    start at an odd and an even step (the memory wavefront of the 2-action kinds draws one Philox block per PAIR of steps and
    has to find its parity), and quadrotor2d's episodes under random actions are short enough for second and third
    terminations inside the 64-step launch (the batched refill of the spare reset states)."""
    import hashlib

    digests = {}
    for name, tune in (("default", {}), ("one wavefront", {"split": 0, "slice": 0}),
                       ("two wavefronts, one launch", {"split": 1, "slice": 0}),
                       ("two wavefronts, sliced", {"slice": 1}),
                       ("two wavefronts, 3 pairs per workgroup", {"split": 1, "slice": 0, "split_group": 3}),
                       ("one wavefront, 64-thread workgroups, write-back stores", {"split": 0, "block": 64, "store_policy": 0}),
                       ("two wavefronts, one pair per workgroup, non-temporal stores", {"split": 1, "slice": 0, "split_group": 1, "store_policy": 2})):
        env = G.BatchedQuadrotor(kind, n, seed=5, auto_reset=True, track_episodes=True)
        env.set_tuning(**tune)
        h = hashlib.sha256()
        for T in (23, 64, 24):
            tr = env.rollout(T, mode="random", layout="soa", want=("actions", "obs", "rew", "done"))
            for k in ("actions", "obs", "rew", "done"):
                h.update(np.ascontiguousarray(tr[k]).tobytes())
        for a in (env.get_state(), env.get_sbd(), env.get_reset_counts()):
            h.update(np.ascontiguousarray(a).tobytes())
        eb = env.episode_buffers()
        for k in sorted(eb):
            h.update(np.ascontiguousarray(eb[k]).tobytes())
        t = env.episode_totals()
        digests[name] = (h.hexdigest(), t["episodes"], t["length_sum"])
        env.close()
    assert len(set(digests.values())) == 1, digests


@pytest.mark.parametrize("mode", ["random", "controller"])
@pytest.mark.parametrize("kind", KINDS)
def test_fused_rollouts_repeat_bit_for_bit_at_four_wavefronts_per_simd(G, kind, mode):
    """131 072 envs on the two-wavefront kernel = four wavefronts on every SIMD, and 262 144 on the one-wavefront kernel: four
    64-step rollouts from the same state must leave identical bits.  (Round 4 found the MFMA kernels of the policy rollouts
    reading stale registers in lanes 48..63 when wavefronts share a SIMD - csrc/rmav_policy_abi.hip; these kernels have no
    matrix instructions and have never shown it, and this test keeps watching.)"""
    import torch

    for n in (131072, 262144):
        ref = None
        for rep in range(4):
            env = G.BatchedQuadrotor(kind, n, seed=11, auto_reset=True, track_episodes=True)
            tr = env.rollout(64, mode=mode, layout="soa", want=("actions", "obs", "rew", "done"), device_out=True)
            cur = [tr[k].clone() for k in ("actions", "obs", "rew", "done")] + [env.get_state(layout="soa", device_out=True).clone()]
            tot = env.episode_totals()
            env.close()
            if ref is None:
                ref, tot0 = cur, tot
                continue
            for x, y in zip(cur, ref):
                assert torch.equal(x, y), (n, rep)
            assert tot == tot0


@pytest.mark.parametrize("kind", ["quad3d", "quad2d_sl"])
def test_episode_lengths_across_launch_kinds_and_counter_moves(G, kind):
    """Episode lengths are kept as 'step counter at episode start' (csrc/rmav_kernels.hpp: ep_clock0): single steps (eager and lazy
    bookkeeping loads), fused rollouts, device-side and host-side reads, rmav_seed / rmav_set_step_count in between, an explicit
    reset() - the running and the finished lengths always equal a host-side count driven by the returned `done` flags."""
    import torch

    n, lo, hi = 3000, *BOX[kind]
    env = G.BatchedQuadrotor(kind, n, seed=4, auto_reset=True, track_episodes=True)
    ln = np.zeros(n, np.int64)
    last = np.zeros(n, np.int64)
    fin_len = 0
    rng = np.random.RandomState(2)

    def account(done_TN):
        nonlocal fin_len
        for dk in np.asarray(done_TN).astype(bool):
            ln[:] += 1
            last[dk] = ln[dk]
            fin_len += int(ln[dk].sum())
            ln[dk] = 0

    def check():
        eb = env.episode_buffers()
        assert np.array_equal(eb["cur_length"], ln), (eb["cur_length"][:8], ln[:8])
        assert np.array_equal(eb["last_length"], last)
        ed = env.episode_buffers(device_out=True)
        assert np.array_equal(ed["cur_length"].cpu().numpy(), ln) and ed["cur_length"].dtype == torch.int32
        assert env.episode_totals()["length_sum"] == fin_len

    for phase in range(4):
        env.set_tuning(step_lazy=phase % 2)
        for _ in range(40):   # single steps: k_step
            a = rng.uniform(lo, hi, (n, NA[kind])).astype(np.float32)
            _, _, d = env.step(a)
            account(d[None])
        check()
        tr = env.rollout(50, mode="random", layout="soa", fused=True, want=("done",))
        account(tr["done"])
        check()
        tr = env.rollout(7, mode="random", layout="soa", fused=False, want=("done",))
        account(tr["done"])
        check()
        if phase == 0:
            env.step_count = 2 ** 32 - 20      # the 32-bit clock wraps inside the next phase
        elif phase == 1:
            env.seed(11)                       # rewinds the step counter to 0
            assert env.step_count == 0
        elif phase == 2:
            env.step_count = 123456789012      # beyond 32 bits
        check()
    env.reset()
    ln[:] = 0                                  # reset() starts a fresh episode (return and length), finished statistics stay
    check()
    _, _, d = env.step(rng.uniform(lo, hi, (n, NA[kind])).astype(np.float32))
    account(d[None])
    check()
    env.close()


@pytest.mark.parametrize("kind", KINDS)
def test_single_step_variants_give_the_same_bits(G, kind):
    """rmav_step through k_step (batch- and feature-major), its write-through / non-temporal store variants and k_step with
    the termination record loaded only in finishing lanes (RMAV_TUNE_STEP_LAZY): same outputs, state, counters and
    episode statistics, bit for bit, over 60 steps with ~1 % of the lanes terminating per step (ragged batch: the last
    wavefront has clones)."""
    import hashlib

    n = 4099
    lo, hi = BOX[kind]
    acts = np.random.RandomState(3).uniform(lo, hi, (60, n, NA[kind])).astype(np.float32)
    digests = {}
    for name, tune, layout in (("k_step", {}, "aos"), ("feature-major", {}, "soa"), ("write-through", {"step_store": 1}, "aos"),
                               ("non-temporal", {"step_store": 2}, "soa"), ("lazy", {"step_lazy": 1}, "aos"),
                               ("lazy, 128-thread workgroups", {"step_lazy": 1, "block": 128}, "soa")):
        env = G.BatchedQuadrotor(kind, n, seed=8, auto_reset=True, track_episodes=True)
        env.set_tuning(**tune)
        h = hashlib.sha256()
        for k in range(60):
            out = env.step(acts[k] if layout == "aos" else np.ascontiguousarray(acts[k].T), layout=layout)
            for x in (out[0] if layout == "aos" else out[0].T, out[1], out[2]):
                h.update(np.ascontiguousarray(x).tobytes())
        for x in (env.get_state(), env.get_sbd(), env.get_reset_counts()):
            h.update(np.ascontiguousarray(x).tobytes())
        eb = env.episode_buffers()
        for key in sorted(eb):
            h.update(np.ascontiguousarray(eb[key]).tobytes())
        t = env.episode_totals()
        digests[name] = (h.hexdigest(), t["episodes"], t["length_sum"])
        assert t["episodes"] > 0
        env.close()
    assert len(set(digests.values())) == 1, digests


@pytest.mark.parametrize("kind", KINDS)
def test_rollout_random_vs_oracle_teacher_forced(G, kind):
    """Random-action rollout with auto-reset: every step is checked against the oracle from the
    device's own previous state; reset states and actions are bit-exact with the RNG specification.  64 steps: quadrotor2d's
    envs terminate up to three or more times inside the launch (the spare reset state, then the batched refills of it)."""
    n, T, seed, base = 4096, 64, 11, 123456
    lo, hi = BOX[kind]
    env = G.BatchedQuadrotor(kind, n, seed=seed, env_id_base=base, auto_reset=True, track_episodes=True)
    prev = env.get_state()
    sbd = env.get_sbd()
    rc = env.get_reset_counts().copy()
    rc_start = rc.copy()
    tr = env.rollout(T, mode="random", layout="aos", fused=True, want=("actions", "obs", "rew", "done"))
    ids = base + np.arange(n)
    ret = np.zeros(n)
    ln = np.zeros(n, np.int64)
    fin_ret, fin_len, fin_n = 0.0, 0, 0
    for k in range(T):
        assert np.array_equal(tr["actions"][k], O.random_actions(kind, seed, ids, k, lo, hi)) if k < 3 else True
        sbd_prev = sbd.copy()
        o2, r, d, sbd = O.batch_step(kind, prev.astype(np.float64), tr["actions"][k].astype(np.float64), sbd)
        dk = tr["done"][k].astype(bool)
        ok = near_threshold(kind, o2)
        assert np.array_equal(dk | ok, d | ok)
        # envs whose norm sits within 1e-5 of a limit may legitimately disagree: follow the device there
        sbd = np.where(dk == d, sbd, np.where(dk, np.where(sbd_prev < 0, 0, sbd_prev + 1), sbd_prev))
        alive = ~dk & ~d
        assert scaled_err(tr["obs"][k][alive], o2[alive]).max() <= TOL
        same = dk == d
        assert scaled_err(tr["rew"][k][same], r[same]).max() <= TOL
        if dk.any():
            assert np.array_equal(tr["obs"][k][dk], O.reset_states(kind, seed, ids[dk], rc[dk]))
        rc = rc + dk.astype(np.uint32)
        ret += tr["rew"][k]
        ln += 1
        fin_ret += ret[dk].sum()
        fin_len += ln[dk].sum()
        fin_n += int(dk.sum())
        ret[dk] = 0
        ln[dk] = 0
        prev = tr["obs"][k]
    assert np.array_equal(env.get_reset_counts(), rc)
    if kind == "quad2d":
        assert int((rc - rc_start).max()) >= 3, "the refill of a used-up spare reset state was not exercised"
    tot = env.episode_totals()
    assert tot["episodes"] == fin_n and tot["length_sum"] == fin_len
    assert abs(tot["return_sum"] - fin_ret) <= 1e-4 * max(1.0, abs(fin_ret))
    eb = env.episode_buffers()
    assert np.array_equal(eb["cur_length"], ln) and np.abs(eb["cur_return"] - ret).max() < 1e-3
    env.close()


@pytest.mark.parametrize("kind", KINDS)
def test_rollout_controller_vs_oracle(G, kind):
    """control -> step fused in-kernel (the reference's test loop at scale): actions match the oracle's
    controller on the device's previous state, the step matches the oracle's step."""
    n, T = 2048, 30
    env = G.BatchedQuadrotor(kind, n, seed=2, auto_reset=True, track_episodes=False)
    prev = env.get_state()
    tr = env.rollout(T, mode="controller", layout="aos", want=("actions", "obs", "rew", "done"))
    sbd = None
    for k in range(T):
        a = O.batch_control(kind, prev.astype(np.float64))
        assert scaled_err(tr["actions"][k], a).max() <= CTRL_TOL
        o2, r, d, sbd = O.batch_step(kind, prev.astype(np.float64), tr["actions"][k].astype(np.float64), sbd)
        dk = tr["done"][k].astype(bool)
        ok = near_threshold(kind, o2)
        assert np.array_equal(dk | ok, d | ok)
        alive = ~dk & ~d
        assert scaled_err(tr["obs"][k][alive], o2[alive]).max() <= TOL
        prev = tr["obs"][k]
    env.close()


# ---- BASELINE.json full sizes: size-independent properties + sampled oracle checks ------------------------
FULL = [("quad3d", 65536), ("quad3d", 1048576), ("quad3d_sl", 262144)]


@pytest.mark.parametrize("kind,n", FULL)
def test_full_size_properties(G, kind, n):
    T, seed = 16, 0
    lo, hi = BOX[kind]
    env = G.BatchedQuadrotor(kind, n, seed=seed, auto_reset=True, track_episodes=True)
    s0 = env.get_state(layout="soa")
    tr = env.rollout(T, mode="random", layout="soa", want=("actions", "obs", "rew", "done"))
    sT = env.get_state(layout="soa")
    assert np.array_equal(sT, tr["obs"][-1])
    # (a) determinism + shard invariance: two half-size shards with global ids reproduce the same bits
    half = n // 2
    parts = []
    for base in (0, half):
        sh = G.BatchedQuadrotor(kind, half, seed=seed, env_id_base=base, auto_reset=True, track_episodes=True)
        parts.append(sh.rollout(T, mode="random", layout="soa", want=("obs", "rew", "done")))
        sh.close()
    for key in ("obs", "rew", "done"):
        assert np.array_equal(np.concatenate([parts[0][key], parts[1][key]], axis=-1), tr[key]), key
    # (b) actions inside the Box, RNG uniform
    assert tr["actions"].min() >= lo and tr["actions"].max() < hi
    assert abs(tr["actions"].mean() - 0.5 * (lo + hi)) < 0.01 * (hi - lo)
    # (c) bookkeeping identities
    done = tr["done"].astype(bool)
    tot = env.episode_totals()
    assert tot["episodes"] == int(done.sum())
    eb = env.episode_buffers()
    total_reward = float(tr["rew"].astype(np.float64).sum())
    assert abs((tot["return_sum"] + float(eb["cur_return"].astype(np.float64).sum())) - total_reward) <= 2e-5 * abs(total_reward) + 1.0
    assert tot["length_sum"] + int(eb["cur_length"].sum()) == n * T
    assert np.array_equal(env.get_reset_counts(), 1 + done.sum(axis=0).astype(np.uint32))
    # rewards: alive steps are -|pos| <= 0, terminal rewards are 1 (first in lifetime) or 0
    assert (tr["rew"][~done] <= 0).all() and set(np.unique(tr["rew"][done])) <= {0.0, 1.0}
    # (d) sampled per-step oracle check (teacher forced)
    rng = np.random.RandomState(1)
    idx = np.sort(rng.choice(n, 4096, replace=False))
    prev = s0[:, idx].T
    sbd = None
    for k in range(T):
        act = tr["actions"][k][:, idx].T
        o2, r, d, sbd = O.batch_step(kind, prev.astype(np.float64), act.astype(np.float64), sbd)
        dk = done[k][idx]
        ok = near_threshold(kind, o2)
        assert np.array_equal(dk | ok, d | ok)
        alive = ~dk & ~d
        assert scaled_err(tr["obs"][k][:, idx].T[alive], o2[alive]).max() <= TOL
        prev = tr["obs"][k][:, idx].T
    env.close()


def test_fused_trajectory_larger_than_4_GiB(G):
    """One fused launch whose obs trajectory is 4.5 GiB (1 M slung-load envs x 72 steps): the per-step
    output pointers advance in 64-bit arithmetic (only the offset inside one step is 32-bit).  The tail
    of the big launch must equal a second handle that reaches the same steps through a small launch."""
    import torch

    kind, n, T, T1 = "quad3d_sl", 1 << 20, 72, 64
    big = G.BatchedQuadrotor(kind, n, seed=5, auto_reset=True, track_episodes=False)
    tr = big.rollout(T, mode="random", layout="soa", want=("obs", "rew", "done"), device_out=True)
    assert tr["obs"].numel() * 4 > (1 << 32)
    ref = G.BatchedQuadrotor(kind, n, seed=5, auto_reset=True, track_episodes=False)
    ref.rollout(T1, mode="random", want=())
    tail = ref.rollout(T - T1, mode="random", layout="soa", want=("obs", "rew", "done"), device_out=True)
    for key in ("obs", "rew", "done"):
        assert torch.equal(tr[key][T1:], tail[key]), key
    assert torch.equal(tr["obs"][-1], big.get_state(layout="soa", device_out=True))
    assert bool(torch.isfinite(tr["rew"]).all())
    big.close()
    ref.close()


@pytest.mark.parametrize("n,T", [((1 << 19) + 37, 64), (20037, 24), (63, 9)])
@pytest.mark.parametrize("kind", KINDS)
def test_batch_major_trajectory_of_a_big_launch(G, kind, n, T):
    """layout='aos' ([T][N][nS]) must be the [T][nS][N] trajectory transposed, bit for bit, on every store path:
    a launch that writes > 448 MB takes the LDS-transposed obs stores (full wavefronts) plus the direct path for
    the ragged last wavefront; small batches run the two-wavefront kernel, whose memory wavefront drains the
    hand-over tile in output order (ragged last wavefront masked)."""
    import torch

    out = {}
    for layout in ("soa", "aos"):
        env = G.BatchedQuadrotor(kind, n, seed=9, auto_reset=True, track_episodes=True)
        out[layout] = env.rollout(T, mode="random", layout=layout, want=("actions", "obs", "rew", "done"), device_out=True)
        out[layout]["state"] = env.get_state(layout="soa", device_out=True)
        env.close()
    soa, aos = out["soa"], out["aos"]
    assert aos["obs"].shape == (T, n, NS[kind])
    assert torch.equal(aos["obs"], soa["obs"].transpose(1, 2))
    assert torch.equal(aos["actions"], soa["actions"].transpose(1, 2))
    assert torch.equal(aos["rew"], soa["rew"]) and torch.equal(aos["done"], soa["done"])
    assert torch.equal(aos["state"], soa["state"])


@pytest.mark.parametrize("mode", ["random", "controller"])
@pytest.mark.parametrize("n,T", [(63, 9), (64 + 29, 16), (20037, 24)])
@pytest.mark.parametrize("kind", KINDS)
def test_ragged_last_wavefront_writes_nothing_out_of_bounds(G, kind, n, T, mode):
    """Every trajectory buffer of a fused launch sits in front of a guard region of sentinels; with n % 64 != 0 the
    clone lanes of the last wavefront (two-wavefront kernel: batch-major obs drained through a range-checked buffer
    descriptor) must leave the guards untouched, and interior rows must equal the SoA launch's (a stray clone row of
    step k would land in rows 0.. of step k+1)."""
    import torch

    guard = 4096
    sentinel = -12345.0
    res = {}
    for layout in ("soa", "aos"):
        env = G.BatchedQuadrotor(kind, n, seed=11, auto_reset=True, track_episodes=True)
        sizes = {"actions": T * n * NA[kind], "obs": T * n * NS[kind], "rew": T * n}
        raw = {k: torch.full((v + guard,), sentinel, dtype=torch.float32, device="cuda") for k, v in sizes.items()}
        raw["done"] = torch.full((T * n + guard,), 77, dtype=torch.uint8, device="cuda")
        shp = (lambda d: (T, d, n)) if layout == "soa" else (lambda d: (T, n, d))
        out = {"actions": raw["actions"][:sizes["actions"]].view(shp(NA[kind])), "obs": raw["obs"][:sizes["obs"]].view(shp(NS[kind])),
               "rew": raw["rew"][:T * n].view(T, n), "done": raw["done"][:T * n].view(T, n)}
        env.rollout(T, mode=mode, layout=layout, want=("actions", "obs", "rew", "done"), device_out=True, out=out)
        torch.cuda.synchronize()
        for k, v in sizes.items():
            assert bool((raw[k][v:] == sentinel).all()), (layout, k)
        assert bool((raw["done"][T * n:] == 77).all()), layout
        res[layout] = out
        env.close()
    assert torch.equal(res["aos"]["obs"], res["soa"]["obs"].transpose(1, 2))
    assert torch.equal(res["aos"]["actions"], res["soa"]["actions"].transpose(1, 2))
    assert torch.equal(res["aos"]["rew"], res["soa"]["rew"]) and torch.equal(res["aos"]["done"], res["soa"]["done"])


@pytest.mark.parametrize("kind", KINDS)
def test_non_finite_inputs_follow_the_reference(G, kind):
    """The reference has no NaN / inf guard: NaN propagates, `NaN > limit` is False so the env is
    not done and the reward is -NaN; an infinite thrust terminates by |pos| = inf.  Same on the device."""
    n = 256
    s, a = random_cases(kind, n, seed=7, wide=False)
    s *= 0.3
    a_nan, a_inf = a.copy(), a.copy()
    a_nan[::2, 0] = np.nan
    a_inf[::2, 0] = np.inf
    for act in (a_nan, a_inf):
        env = G.BatchedQuadrotor(kind, n, auto_reset=False, track_episodes=False)
        env.set_state(s)
        obs, rew, done = env.step(act)
        o2, r, d, _ = O.batch_step(kind, s.astype(np.float64), act.astype(np.float64))
        assert np.array_equal(np.isnan(obs), np.isnan(o2)) and np.array_equal(np.isinf(obs), np.isinf(o2))
        assert np.array_equal(done, d)
        assert np.array_equal(np.isnan(rew), np.isnan(r))
        fin = np.isfinite(o2).all(axis=1)
        assert scaled_err(obs[fin], o2[fin]).max() <= TOL
        env.close()


@pytest.mark.parametrize("kind,n", [("quad3d", 65536), ("quad3d", 1048576)])
def test_full_size_yaw_equivariance(G, kind, n):
    """A size-independent property of the dynamics at BASELINE's full sizes: gravity is along z, so rotating the
    world about z commutes with step():  step(Rz s, a) == Rz step(s, a)  (positions / velocities rotate, the
    attitude quaternion is left-multiplied by the yaw quaternion, body rates and thrust are unchanged); reward and
    done are invariant.  (Not a property of the slung-load envs: their tether force subtracts a scalar from a
    vector, quadrotor3d_slungload.py:110, which singles out the coordinate axes - a quirk we reproduce.)"""
    rng = np.random.RandomState(12)
    env = G.BatchedQuadrotor(kind, n, seed=1, auto_reset=False, track_episodes=False)
    s = env.get_state()                                   # U[-1,1) reset states
    lo, hi = BOX[kind]
    a = rng.uniform(lo, hi, (n, 4)).astype(np.float32)
    psi = rng.uniform(-np.pi, np.pi, n)
    c, sn = np.cos(psi), np.sin(psi)

    def rot_vec(v):                                        # [n,3]
        return np.stack([c * v[:, 0] - sn * v[:, 1], sn * v[:, 0] + c * v[:, 1], v[:, 2]], axis=1)

    def rot_quat(q):                                       # q_yaw (x) q, q_yaw = (cos psi/2, 0, 0, sin psi/2)
        cw, cz = np.cos(psi / 2), np.sin(psi / 2)
        w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        return np.stack([cw * w - cz * z, cw * x - cz * y, cw * y + cz * x, cw * z + cz * w], axis=1)

    def rot_state(st):
        st = st.astype(np.float64)
        out = st.copy()
        out[:, 0:3], out[:, 3:7], out[:, 7:10] = rot_vec(st[:, 0:3]), rot_quat(st[:, 3:7]), rot_vec(st[:, 7:10])
        if kind == "quad3d_sl":
            out[:, 10:13], out[:, 13:16] = rot_vec(st[:, 10:13]), rot_vec(st[:, 13:16])
        return out

    obs, rew, done = env.step(a)
    s_rot = rot_state(s).astype(np.float32)
    env.set_state(s_rot)
    env.set_sbd(np.full(n, -1, np.int32))
    obs_r, rew_r, done_r = env.step(a)
    expect = rot_state(obs)
    near = near_threshold(kind, obs.astype(np.float64), eps=1e-4)
    assert np.array_equal(done | near, done_r | near)
    # the rotated input is rounded to fp32 once more, so allow a few ulps of the state magnitude
    assert scaled_err(obs_r, expect).max() < 2e-6
    ok = ~near & ~done
    assert np.abs(rew_r[ok] - rew[ok]).max() < 2e-6 * max(1.0, float(np.abs(rew[ok]).max()))
    env.close()


@pytest.mark.parametrize("mode", ["random", "controller", "buffer"])
@pytest.mark.parametrize("n,T,fused", [(63, 9, True), (4099, 24, True), (65599, 16, True), (131071, 12, True), (300007, 10, True), (4099, 5, False)])
@pytest.mark.parametrize("kind", KINDS)
def test_pitched_rollout_equals_the_plain_layout(G, kind, n, T, fused, mode):
    """rmav_rollout_pitched (feature columns `pitch` elements apart, so that a batch size that is not a multiple of 16 keeps the
    aligned store path) writes the same values as rmav_rollout, leaves elements [N, pitch) of every column alone, and leaves
    the envs in the same state - every kernel family (two-wavefront, one-wavefront, single-step loop), caller actions included."""
    import ctypes as C
    import torch
    from gym_reinmav_amd import _abi as A

    nS, nA = NS[kind], NA[kind]
    lo, hi = BOX[kind]
    acts = None
    if mode == "buffer":
        acts = torch.empty((T, nA, n), device="cuda").uniform_(lo, hi, generator=torch.Generator(device="cuda").manual_seed(3))
    ref_env = G.BatchedQuadrotor(kind, n, seed=9, auto_reset=True, track_episodes=True)
    ref = ref_env.rollout(T, mode=mode, actions=acts, layout="soa", fused=fused, want=("actions", "obs", "rew", "done"),
                          device_out=True, pitched=False)
    env = G.BatchedQuadrotor(kind, n, seed=9, auto_reset=True, track_episodes=True)
    P = int(env._lib.rmav_trajectory_pitch(env._h))
    assert P % 64 == 0 and n <= P < n + 64
    SENT = -12345.0
    a_out = torch.full((T, nA, P), SENT, device="cuda")
    obs = torch.full((T, nS, P), SENT, device="cuda")
    rew = torch.full((T, P), SENT, device="cuda")
    done = torch.full((T, P), 77, dtype=torch.uint8, device="cuda")
    a_in = None
    if mode == "buffer":
        a_in = torch.full((T, nA, P), SENT, device="cuda")
        a_in[..., :n] = acts
    ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    A.check(env._lib.rmav_rollout_pitched(env._h, T, {"buffer": A.ACT_BUFFER, "random": A.ACT_RANDOM, "controller": A.ACT_CONTROLLER}[mode],
                                          ptr(a_in), None if mode == "buffer" else ptr(a_out), ptr(obs), ptr(rew), ptr(done), P, 1 if fused else 0))
    torch.cuda.synchronize()
    if mode != "buffer":
        assert torch.equal(a_out[..., :n], ref["actions"]) and bool((a_out[..., n:] == SENT).all())
    assert torch.equal(obs[..., :n], ref["obs"]) and bool((obs[..., n:] == SENT).all())
    assert torch.equal(rew[..., :n], ref["rew"]) and bool((rew[..., n:] == SENT).all())
    assert torch.equal(done[..., :n], ref["done"]) and bool((done[..., n:] == 77).all())
    assert np.array_equal(env.get_state(), ref_env.get_state()) and np.array_equal(env.get_reset_counts(), ref_env.get_reset_counts())
    assert env.episode_totals() == ref_env.episode_totals()
    if mode != "buffer" and fused:
        # the Python wrapper: plain contiguous tensors unless asked (round 4 pitched by itself and broke `out=` reuse and .view());
        # pitched=True returns [..., :N] views, and handing those back as out= goes through the pitched entry point again
        env2 = G.BatchedQuadrotor(kind, n, seed=9, auto_reset=True, track_episodes=True)
        want = ("actions", "obs", "rew", "done")
        plain = env2.rollout(T, mode=mode, layout="soa", want=want, device_out=True)
        assert all(plain[k].is_contiguous() and torch.equal(plain[k], ref[k]) for k in want)
        again = env2.rollout(T, mode=mode, layout="soa", want=want, device_out=True, out=plain)     # the allocate-then-reuse idiom
        assert all(again[k].data_ptr() == plain[k].data_ptr() for k in want)
        env2.close()
        env3 = G.BatchedQuadrotor(kind, n, seed=9, auto_reset=True, track_episodes=True)
        tr = env3.rollout(T, mode=mode, layout="soa", want=want, device_out=True, pitched=True)
        assert tr["obs"].stride(-2) == P and tr["rew"].stride(0) == P and all(torch.equal(tr[k], ref[k]) for k in want)
        first = {k: v.clone() for k, v in tr.items()}
        tr2 = env3.rollout(T, mode=mode, layout="soa", want=want, device_out=True, out=tr)           # pitched views handed back
        assert all(tr2[k].data_ptr() == tr[k].data_ptr() for k in want)
        ref2 = ref_env.rollout(T, mode=mode, layout="soa", want=want, device_out=True)
        assert all(torch.equal(tr2[k], ref2[k]) for k in want) and not torch.equal(tr2["obs"], first["obs"])
        with pytest.raises(ValueError):    # a mix of pitches is refused, not silently mis-addressed
            env3.rollout(T, mode=mode, layout="soa", want=want, device_out=True, out=dict(tr, rew=torch.empty((T, n), device="cuda")))
        env3.close()
    env.close()
    ref_env.close()


@pytest.mark.parametrize("mode", ["random", "controller", "buffer"])
@pytest.mark.parametrize("kind,n,chunk,T", [("quad3d", 131072, 65536, 12), ("quad3d", 150016, 65536, 9), ("quad3d_sl", 70000, 32768, 10),
                                            ("quad2d", 20000, 8192, 16), ("quad2d_sl", 4099, 1024, 7), ("quad3d", 5000, 8192, 6)])
def test_chunked_rollout_equals_the_plain_layout(G, kind, n, chunk, T, mode):
    """rmav_rollout_chunked (chunk-major trajectory arrays [C][T][dim][chunk], one two-wavefront launch per chunk) writes the values
    rmav_rollout writes - permuted, nothing else - leaves columns past N in the last chunk alone, and leaves the envs in the same
    state with the same counters and episode statistics.  Ragged last chunks, a chunk larger than the batch (= the plain layout
    with a pitch), caller actions in the chunked layout."""
    import torch

    nS, nA = NS[kind], NA[kind]
    lo, hi = BOX[kind]
    want = ("actions", "obs", "rew", "done")
    acts = None
    if mode == "buffer":
        acts = torch.empty((T, nA, n), device="cuda").uniform_(lo, hi, generator=torch.Generator(device="cuda").manual_seed(5))
    ref_env = G.BatchedQuadrotor(kind, n, seed=21, auto_reset=True, track_episodes=True)
    ref = ref_env.rollout(T, mode=mode, actions=acts, layout="soa", want=want, device_out=True)
    env = G.BatchedQuadrotor(kind, n, seed=21, auto_reset=True, track_episodes=True)
    ch = min(chunk, (n + 63) // 64 * 64)
    nc = -(-n // ch)
    SENT = -4321.0
    out = {"obs": torch.full((nc, T, nS, ch), SENT, device="cuda"), "rew": torch.full((nc, T, ch), SENT, device="cuda"),
           "done": torch.full((nc, T, ch), 55, dtype=torch.uint8, device="cuda")}
    a_chunked = None
    if mode == "buffer":
        pad = torch.zeros((T, nA, nc * ch), device="cuda")
        pad[..., :n] = acts
        a_chunked = pad.reshape(T, nA, nc, ch).permute(2, 0, 1, 3).contiguous()
    else:
        out["actions"] = torch.full((nc, T, nA, ch), SENT, device="cuda")
    tr = env.rollout_chunked(T, mode=mode, actions=a_chunked, chunk=chunk, want=want, out=out)
    torch.cuda.synchronize()
    for k in want:
        assert torch.equal(env.unchunk(tr[k]), ref[k]), k
    tail = nc * ch - n                                   # columns of the last chunk that belong to no env: untouched
    if tail:
        assert bool((out["obs"][-1, :, :, ch - tail:] == SENT).all()) and bool((out["rew"][-1, :, ch - tail:] == SENT).all())
        assert bool((out["done"][-1, :, ch - tail:] == 55).all())
    assert np.array_equal(env.get_state(), ref_env.get_state()) and np.array_equal(env.get_reset_counts(), ref_env.get_reset_counts())
    assert np.array_equal(env.get_sbd(), ref_env.get_sbd())
    ea, eb = env.episode_buffers(), ref_env.episode_buffers()
    assert all(np.array_equal(ea[k], eb[k]) for k in ea)
    ta, tb = env.episode_totals(), ref_env.episode_totals()
    assert ta["episodes"] == tb["episodes"] and ta["length_sum"] == tb["length_sum"]
    # a second call continues from there (step counter, RNG blocks) exactly like the plain one
    tr2 = env.rollout_chunked(T, mode=("random" if mode == "buffer" else mode), chunk=chunk, want=("obs", "done"))
    ref2 = ref_env.rollout(T, mode=("random" if mode == "buffer" else mode), layout="soa", want=("obs", "done"), device_out=True)
    assert torch.equal(env.unchunk(tr2["obs"]), ref2["obs"]) and torch.equal(env.unchunk(tr2["done"]), ref2["done"])
    env.close()
    ref_env.close()


@pytest.mark.parametrize("kind,n", [("quad2d", 4099), ("quad3d", 5000), ("quad3d_sl", 63), ("quad3d", 65536 + 77)])
def test_chunked_rollout_default_chunk_for_any_batch_size(G, kind, n):
    """rmav_chunk_envs() is always a value rmav_rollout_chunked accepts (it returned N itself, which the multiple-of-64
    check refused whenever N % 64 != 0): the default call works for every batch size and equals the plain rollout."""
    import torch

    env = G.BatchedQuadrotor(kind, n, seed=3, auto_reset=True, track_episodes=True)
    ref_env = G.BatchedQuadrotor(kind, n, seed=3, auto_reset=True, track_episodes=True)
    ch = int(env._lib.rmav_chunk_envs(env._h))
    assert ch % 64 == 0 and (ch == 65536 if (kind == "quad3d" and n > 65536) else ch == (n + 63) // 64 * 64)
    tr = env.rollout_chunked(5, mode="random", want=("actions", "obs", "rew", "done"))          # chunk = None: the recommendation
    ref = ref_env.rollout(5, mode="random", layout="soa", want=("actions", "obs", "rew", "done"), device_out=True)
    torch.cuda.synchronize()
    for k in ref:
        assert torch.equal(env.unchunk(tr[k]), ref[k]), k
    # the same through rollout(layout="chunked") - what bench.py and a learner call - reusing the arrays as `out`
    tr2 = env.rollout(5, mode="random", layout="chunked", want=("actions", "obs", "rew", "done"), out=tr)
    ref2 = ref_env.rollout(5, mode="random", layout="soa", want=("actions", "obs", "rew", "done"), device_out=True)
    torch.cuda.synchronize()
    assert all(tr2[k] is tr[k] for k in tr)
    for k in ref2:
        assert torch.equal(env.unchunk(tr2[k]), ref2[k]), k
    # a learner's flattened samples: every (step, env) row of the plain layout appears, once, in the chunk-major flattening
    nc, T, nS_, chw = tr2["obs"].shape
    flat = tr2["obs"].permute(0, 1, 3, 2).reshape(nc, T, chw, nS_)
    keep = (torch.arange(nc * chw, device="cuda").reshape(nc, 1, chw) < n).expand(nc, T, chw)
    assert torch.equal(flat[keep].reshape(nc, T, -1, nS_)[0] if nc == 1 else flat[keep],
                       (ref2["obs"].permute(0, 2, 1).reshape(T, n, nS_) if nc == 1 else
                        torch.cat([ref2["obs"][:, :, c * chw:min(n, (c + 1) * chw)].permute(0, 2, 1).reshape(-1, nS_) for c in range(nc)])))
    # direct C callers: a caller-action echo would be sized for the plain layout - refused in chunk-major mode
    from gym_reinmav_amd import _abi as A
    if n > 128:
        a = torch.zeros((-(-n // 64), 5, NA[kind], 64), device="cuda")
        with pytest.raises(A.RmavError):
            A.check(env._lib.rmav_rollout_chunked(env._h, 5, A.ACT_BUFFER, a.data_ptr(), a.data_ptr() + 4, None, None, None, 64))
    env.close()
    ref_env.close()


def test_chunked_rollout_rejects_bad_arguments(G):
    from gym_reinmav_amd import _abi as A

    env = G.BatchedQuadrotor("quad3d", 200000, seed=1)
    assert int(env._lib.rmav_chunk_envs(env._h)) == 65536
    with pytest.raises(A.RmavError):
        env.rollout_chunked(8, chunk=1000)            # not a multiple of 64
    with pytest.raises(A.RmavError):
        env.rollout_chunked(1, chunk=65536)           # n_steps >= 2
    with pytest.raises(A.RmavError):
        env.rollout_chunked(8, chunk=196608)          # beyond the two-wavefront kernel's capacity
    env.close()
    small = G.BatchedQuadrotor("quad2d", 200000, seed=1)
    assert int(small._lib.rmav_chunk_envs(small._h)) == 200000   # only quadrotor3d's launches are store-bound: one chunk = the plain layout (N is a multiple of 64 here)
    small.close()


def test_pitched_rollout_rejects_bad_arguments(G):
    import torch
    from gym_reinmav_amd import _abi as A

    env = G.BatchedQuadrotor("quad3d", 1000, seed=1)
    buf = torch.empty((4, 10, 1024), device="cuda")
    with pytest.raises(A.RmavError):
        A.check(env._lib.rmav_rollout_pitched(env._h, 4, A.ACT_RANDOM, None, None, buf.data_ptr(), None, None, 999, 1))
    with pytest.raises(A.RmavError):
        A.check(env._lib.rmav_rollout_pitched(env._h, 4, A.ACT_RANDOM, None, None, buf.data_ptr(), None, None, 0, 1))
    with pytest.raises(ValueError):
        env.rollout(4, mode="random", layout="aos", device_out=True, pitched=True)
    env.close()
