"""The drop-in boundary on the GPU: gym.Env-shaped single envs, the VecEnv contract, error behaviour."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle as O
from util import BOX, CTRL_TOL, KINDS, NA, NS, TOL, near_threshold, scaled_err

pytestmark = pytest.mark.gpu

ENV_IDS = {"quad2d": "quadrotor2d-v0", "quad2d_sl": "quadrotor2d-slungload-v0", "quad3d": "quadrotor3d-v0",
           "quad3d_sl": "quadrotor3d-slungload-v0"}


@pytest.fixture(scope="module")
def G(built):
    import torch

    assert torch.cuda.is_available()
    import gym_reinmav_amd as g

    return g


@pytest.mark.parametrize("kind", KINDS)
def test_gym_env_closed_loop_like_reference_test(G, kind):
    """The loop of test/test_quadrotor3d.py:13-22 (reset; 400 x control -> step; reset on done), no render,
    through the gym-shaped class; each step is checked against the oracle from the env's own state."""
    env = G.make(ENV_IDS[kind], seed=0)
    obs = env.reset()
    assert obs.shape == (NS[kind],) and obs.dtype == np.float64
    assert np.array_equal(obs.astype(np.float32), O.reset_state(kind, 0, 0, 1))  # ctor drew reset #0
    sbd = None
    assert env.steps_beyond_done is None
    n_done = 0
    for i in range(400):
        s = env.state
        a = env.control()
        assert a.shape == (NA[kind],)
        assert scaled_err(a, O.control(kind, s)).max() <= CTRL_TOL
        obs, reward, done, info = env.step(a)
        assert isinstance(reward, float) and isinstance(done, bool) and info == {}
        o2, r, d, sbd = O.step(kind, s, a.astype(np.float32).astype(np.float64), sbd)
        assert scaled_err(obs, o2).max() <= TOL
        if not near_threshold(kind, o2[None])[0]:
            assert done == d and abs(reward - r) <= TOL * max(1.0, abs(r))
        else:
            sbd = env.steps_beyond_done
        assert env.steps_beyond_done == sbd
        if done:
            n_done += 1
            env.reset()
    env.close()


def test_quadrotor2d_config1_controller_loop_matches_golden(G, golden):
    """BASELINE config 1 (quadrotor2d-v0, batch=1, geometric controller) against the reference's own
    recorded 400-step loop, teacher-forced from the recorded pre-step states."""
    g = golden["quad2d"]
    env = G.make("quadrotor2d-v0", seed=0)
    for e in range(2):
        env.steps_beyond_done = None
        seen_done = 0
        for k in range(400):
            env.state = g["traj_s"][e, k]
            a = env.control()
            assert scaled_err(a, g["traj_a"][e, k]).max() <= 1e-5  # recorded state is rounded to fp32 on entry
            obs, reward, done, _ = env.step(g["traj_a"][e, k])
            assert scaled_err(obs, g["traj_s2"][e, k]).max() <= 2e-6
            if not near_threshold("quad2d", g["traj_s2"][e, k][None])[0]:
                assert done == bool(g["traj_d"][e, k])
                exp = g["traj_r"][e, k] if not done else (1.0 if seen_done == 0 else 0.0)
                assert abs(reward - exp) <= 2e-6
            seen_done += int(done)
    env.close()


def test_gym_env_attributes(G):
    env = G.make("quadrotor3d-slungload-v0", seed=1)
    assert env.mass == 1.0 and env.dt == 0.01 and env.load_mass == 0.1 and env.tether_length == 1.5
    assert list(env.g) == [0.0, 0.0, -9.8] and list(env.ref_pos) == [0.0, 0.0, 1.0]
    assert env.pos_threshold == 3.0 and env.vel_threshold == 10.0
    assert env.action_space.shape == (4,) and env.observation_space.shape == (16,)
    assert env.seed(42) == [42]
    assert np.array_equal(env.reset().astype(np.float32), O.reset_state("quad3d_sl", 42, 0, 0))
    env.state = np.arange(16) / 16.0
    assert np.array_equal(env.state, (np.arange(16) / 16.0).astype(np.float32).astype(np.float64))
    env.steps_beyond_done = 3
    assert env.steps_beyond_done == 3
    with pytest.raises(NotImplementedError):
        env.render()
    env.close()
    e2 = G.make("quadrotor2d-v0", reading="A")
    assert e2.vel_threshold == 10.0
    e2.close()


def test_gym_env_public_attributes_write_through(G):
    """The reference reads self.mass / dt / g / ref_pos / ref_vel / thresholds (and load_mass / tether_length) on every call
    (quadrotor3d.py:86,96-102,148,162; quadrotor3d_slungload.py:101-128), so assigning them - or an element of the array-valued
    ones - changes the next control() / step().  Same here: the attributes are views of the handle's rmav_params."""
    from util import CTRL_TOL, TOL, scaled_err

    def assert_close(x, ref, tol):
        assert scaled_err(x, ref).max() <= tol, scaled_err(x, ref).max()

    def ctrl_close(a, ref):
        assert scaled_err(a, ref).max() <= CTRL_TOL, scaled_err(a, ref).max()

    STATE_TOL = TOL

    env = G.make("quadrotor3d-v0", seed=3)
    s0 = env.reset()
    a_default = env.control()
    ctrl_close(a_default, O.control("quad3d", s0))
    # whole-array assignment, then element assignment
    env.ref_pos = (1.0, 0.0, 1.0)
    env.mass = 1.3
    p = O.default_params("quad3d")
    p.mass = 1.3
    p.ref_pos[0], p.ref_pos[1], p.ref_pos[2] = 1.0, 0.0, 1.0
    assert list(env.ref_pos) == [1.0, 0.0, 1.0] and env.mass == 1.3
    a = env.control()
    exp = O.control("quad3d", s0, p)
    assert np.abs(a - a_default).max() > 1e-3           # the set-point moved the command
    ctrl_close(a, exp)
    obs, r, d, _ = env.step(a)                          # the step uses the new mass
    s1, r1, d1, _ = O.step("quad3d", s0, a.astype(np.float32).astype(np.float64), None, p)
    assert_close(obs, s1, STATE_TOL)
    assert abs(r - r1) <= 1e-6 * max(1.0, abs(r1)) and d == d1
    # after step() the class holds a cached control() action: an attribute write must drop it
    cached = env.control()
    env.ref_pos[2] = 2.5                                # element write on the handed-out array (the reference's self.ref_pos IS the array)
    p.ref_pos[2] = 2.5
    assert env.ref_pos[2] == 2.5
    a2 = env.control()
    exp2 = O.control("quad3d", obs, p)
    assert np.abs(a2 - cached).max() > 1e-3
    ctrl_close(a2, exp2)
    old = env.ref_pos                                   # a snapshot kept across a re-assignment ...
    env.ref_pos = (1.0, 0.5, 2.5)
    p.ref_pos[1] = 0.5
    old[0] = 1.25                                       # ... writes ONE element of the CURRENT value, not its stale other elements
    p.ref_pos[0] = 1.25
    assert list(env.ref_pos) == [1.25, 0.5, 2.5] and list(old) == [1.25, 0.5, 2.5]
    env.ref_vel = [0.1, 0.0, -0.1]
    env.dt = 0.02
    env.g = np.array([0.0, 0.0, -3.7])
    p.ref_vel[0], p.ref_vel[2], p.dt, p.g_vec[2] = 0.1, -0.1, 0.02, -3.7
    assert list(env.g) == [0.0, 0.0, -3.7] and env.dt == 0.02
    a3 = env.control()
    ctrl_close(a3, O.control("quad3d", obs, p))
    o2, r2, d2, _ = env.step(a3)
    s2, rr2, dd2, _ = O.step("quad3d", obs, a3.astype(np.float32).astype(np.float64), None, p)
    assert_close(o2, s2, STATE_TOL)
    env.pos_threshold = 0.01                            # every state is now out of bounds: the next step terminates
    assert env.pos_threshold == 0.01
    _, _, d3, _ = env.step(a3)
    assert d3 is True
    env.pos_threshold = 3.0
    env.g = (1.0, -0.5, -9.0)                           # the reference's self.g is a free 3-vector (quadrotor3d.py:47,96-99,162)
    env.g[1] = 0.25                                     # ... and so are its elements
    p.pos_limit, p.g_vec[0], p.g_vec[1], p.g_vec[2] = 3.0, 1.0, 0.25, -9.0
    assert list(env.g) == [1.0, 0.25, -9.0]
    s_t = np.asarray(env.state, dtype=np.float64)
    a4 = env.control()
    ctrl_close(a4, O.control("quad3d", s_t, p))
    o4, _, _, _ = env.step(a4)
    s4, _, _, _ = O.step("quad3d", s_t, a4.astype(np.float32).astype(np.float64), None, p)
    assert_close(o4, s4, STATE_TOL)
    with pytest.raises(ValueError):
        env.g = (0.0, -9.8)                             # wrong length
    with pytest.raises(G._abi.RmavError):
        env.g = (0.0, float("nan"), -9.8)               # rmav_set_params validates
    with pytest.raises(ValueError):
        env.ref_pos = (1.0, 2.0)
    with pytest.raises(G._abi.RmavError):
        env.mass = -1.0                                 # rmav_set_params validates
    assert env.mass == 1.3
    with pytest.raises(AttributeError):
        env.load_mass                                   # noqa: B018 - Quadrotor3D has no load
    env.close()
    sl = G.make("quadrotor3d-slungload-v0", seed=4)
    s0 = sl.reset()
    sl.load_mass, sl.tether_length = 0.25, 1.0
    q = O.default_params("quad3d_sl")
    q.load_mass, q.tether_length = 0.25, 1.0
    assert sl.load_mass == 0.25 and sl.tether_length == 1.0
    a = sl.control()
    ctrl_close(a, O.control("quad3d_sl", s0, q))
    obs, r, d, _ = sl.step(a)
    s1, r1, d1, _ = O.step("quad3d_sl", s0, a.astype(np.float32).astype(np.float64), None, q)
    if abs(float(O.tether_slack("quad3d_sl", s0, q))) > 1e-5:
        assert_close(obs, s1, STATE_TOL)
    sl.close()
    e2 = G.make("quadrotor2d-v0", seed=5)
    e2.ref_pos = (0.5, -0.25)
    e2.g = (0.0, -5.0)
    p2 = O.default_params("quad2d")
    p2.ref_pos[0], p2.ref_pos[1], p2.g_vec[1] = 0.5, -0.25, -5.0   # 2-D control() keeps the literal (0, 9.8) (quadrotor2d.py:130)
    s0 = e2.reset()
    a = e2.control()
    ctrl_close(a, O.control("quad2d", s0, p2))
    e2.g = (0.75, -5.0)
    p2.g_vec[0] = 0.75
    o2d, _, _, _ = e2.step(a)
    s2d, _, _, _ = O.step("quad2d", s0, a.astype(np.float32).astype(np.float64), None, p2)
    assert_close(o2d, s2d, STATE_TOL)
    e2.close()


@pytest.mark.parametrize("numpy_io,lazy", [(False, False), (True, False), (False, True), (True, True)])
def test_vec_env_contract(G, numpy_io, lazy):
    """lazy: the infos of a big batch (> 4 096 envs by default; forced here) are a list-like that is materialised on first use -
    the same `len(infos) == num_envs` / `infos[i]['episode']` contract a baselines Runner iterates."""
    import torch

    n, seed = 512, 9
    venv = G.QuadrotorVecEnv("quadrotor3d-v0", n, seed=seed, numpy_io=numpy_io, dict_infos=not lazy)
    assert G.QuadrotorVecEnv.__init__.__defaults__ is not None and venv.dict_infos == (not lazy)
    assert venv.num_envs == n and venv.observation_space.shape == (10,) and venv.action_space.shape == (4,)
    obs = venv.reset()
    to_np = (lambda x: x) if numpy_io else (lambda x: x.cpu().numpy())
    prev = to_np(obs)
    assert prev.shape == (n, 10) and np.array_equal(prev, O.reset_states("quad3d", seed, np.arange(n), 1))
    rc = np.full(n, 2, np.uint32)
    rng = np.random.RandomState(0)
    sbd = None
    ep_ret = np.zeros(n)
    ep_len = np.zeros(n, int)
    saw_episode = False
    for k in range(150):
        a = rng.uniform(0, 10, (n, 4)).astype(np.float32)
        venv.step_async(a if numpy_io else torch.from_numpy(a).cuda())
        obs, rew, done, infos = venv.step_wait()
        obs, rew, done = to_np(obs), to_np(rew), to_np(done)
        assert obs.dtype == np.float32 and rew.dtype == np.float32 and done.dtype == bool and len(infos) == n
        o2, r, d, sbd = O.batch_step("quad3d", prev.astype(np.float64), a.astype(np.float64), sbd)
        ok = near_threshold("quad3d", o2)
        assert np.array_equal(done | ok, d | ok)
        alive = ~done & ~d
        assert scaled_err(obs[alive], o2[alive]).max() <= TOL
        ep_ret += rew
        ep_len += 1
        if done.any():  # auto-reset: the returned obs of a finished env is its post-reset obs
            assert np.array_equal(obs[done], O.reset_states("quad3d", seed, np.nonzero(done)[0], rc[done]))
            for i in np.nonzero(done)[0]:
                assert infos[i]["episode"]["l"] == ep_len[i] and set(infos[i]["episode"]) == {"r", "l", "t"}   # Monitor's keys
                assert 0.0 <= infos[i]["episode"]["t"] < 600.0
                assert abs(infos[i]["episode"]["r"] - ep_ret[i]) < 1e-3
                saw_episode = True
            ep_ret[done] = 0
            ep_len[done] = 0
        assert all("episode" not in infos[i] for i in np.nonzero(~done)[0][:16])
        if k % 50 == 0:   # what ppo2's Runner does with them: `for info in infos: maybeepinfo = info.get('episode')`
            assert sum(1 for info in infos if info.get("episode")) == int(done.sum())
            assert isinstance(infos[-1], dict) and len(infos[:3]) == 3
        rc += done.astype(np.uint32)
        prev = obs
    assert saw_episode
    if lazy:   # infos not read before the next step cannot report that step's statistics any more: an error, not stale numbers
        a = rng.uniform(0, 10, (n, 4)).astype(np.float32)
        old = venv.step(a if numpy_io else torch.from_numpy(a).cuda())[3]
        venv.step(a if numpy_io else torch.from_numpy(a).cuda())
        with pytest.raises(RuntimeError):
            old[0]
        venv.step(a if numpy_io else torch.from_numpy(a).cuda())      # ... and stays an error two steps later (round 4 recycled two
        with pytest.raises(RuntimeError):                             # objects, so an old reference answered for step s + 2)
            old[0]
        cur = venv.step(a if numpy_io else torch.from_numpy(a).cuda())[3]
        i0 = int(np.nonzero(~np.asarray(cur._done if numpy_io else cur._done.cpu().numpy()))[0][0])
        cur[i0]["mine"] = 1                                           # a wrapper writing into one env's info pollutes nothing
        assert "mine" not in cur[i0 + 1 if i0 + 1 < n else i0 - 1] and all("mine" not in x for x in cur[:8])
    venv.close()
    big = G.QuadrotorVecEnv("quadrotor3d-v0", 8192, seed=seed)       # the default beyond 4 096 envs
    big.reset()
    infos = big.step(torch.zeros((8192, 4), device="cuda"))[3]
    assert len(infos) == 8192 and isinstance(infos[5], dict)
    big.close()


def test_error_behaviour(G):
    A = G._abi
    L = A.lib()
    h = C.c_void_p()
    assert L.rmav_create(C.byref(h), 7, 16, 0, 0, 0, 0, None, None) == A.ERR_INVALID
    assert L.rmav_create(C.byref(h), A.QUAD3D, 0, 0, 0, 0, 0, None, None) == A.ERR_INVALID
    assert L.rmav_create(C.byref(h), A.QUAD3D, 16, 99, 0, 0, 0, None, None) == A.ERR_INVALID
    assert L.rmav_create(C.byref(h), A.QUAD3D, 16, 0, 0, 0, 64, None, None) == A.ERR_INVALID
    assert b"flag" in L.rmav_last_error()
    bad = A.default_params(A.QUAD3D)
    bad.dt = 0.0
    assert L.rmav_create(C.byref(h), A.QUAD3D, 16, 0, 0, 0, 0, C.byref(bad), None) == A.ERR_INVALID
    assert L.rmav_create(C.byref(h), A.QUAD3D, 16, 0, 0, 0, 0, None, None) == A.OK
    assert L.rmav_num_envs(h) == 16
    assert L.rmav_step(h, None, None, None, None, A.HOST, A.AOS) == A.ERR_INVALID
    buf = np.zeros(64, np.float32)
    assert L.rmav_step(h, buf.ctypes.data, None, None, None, 5, A.AOS) == A.ERR_INVALID
    assert L.rmav_rollout(h, 0, A.ACT_RANDOM, None, None, None, None, None, A.HOST, A.SOA, 1) == A.ERR_INVALID
    assert L.rmav_rollout(h, 4, A.ACT_BUFFER, None, None, None, None, None, A.HOST, A.SOA, 1) == A.ERR_INVALID
    assert L.rmav_rollout(h, 4, 9, None, None, None, None, None, A.HOST, A.SOA, 1) == A.ERR_INVALID
    tot = A.EpTotals()
    assert L.rmav_episode_totals(h, C.byref(tot), 0) == A.ERR_INVALID  # created without TRACK_EPISODES
    # a step with all-NULL outputs is legal (state advances on the device only)
    assert L.rmav_step(h, buf.ctypes.data, None, None, None, A.HOST, A.AOS) == A.OK
    t = C.c_uint64()
    assert L.rmav_get_step_count(h, C.byref(t)) == A.OK and t.value == 1
    assert L.rmav_destroy(h) == A.OK
    with pytest.raises(ValueError):
        env = G.BatchedQuadrotor("quad3d", 8)
        try:
            env.step(np.zeros((8, 3), np.float32))
        finally:
            env.close()


def test_params_override_changes_dynamics(G):
    """rmav_params is live: heavier vehicle / longer dt must match the oracle with the same params."""
    A = G._abi
    p = A.default_params(A.QUAD3D_SL)
    p.mass, p.load_mass, p.dt, p.tether_length = 1.7, 0.3, 0.02, 1.2
    q = O.default_params("quad3d_sl")
    q.mass, q.load_mass, q.dt, q.tether_length = 1.7, 0.3, 0.02, 1.2
    rng = np.random.RandomState(5)
    s = rng.uniform(-1, 1, (4096, 16)).astype(np.float32)
    a = rng.uniform(-10, 10, (4096, 4)).astype(np.float32)
    env = G.BatchedQuadrotor("quad3d_sl", 4096, auto_reset=False, track_episodes=False, params=p)
    env.set_state(s)
    obs, rew, done = env.step(a)
    o2, r, d, _ = O.batch_step("quad3d_sl", s.astype(np.float64), a.astype(np.float64), params=q)
    assert scaled_err(obs, o2).max() <= TOL
    ok = near_threshold("quad3d_sl", o2)
    assert np.array_equal(done | ok, d | ok)
    env.close()


def test_multi_shard_all_gather_single_process(G):
    """Two shards on one GPU (virtual ranks): concatenated per-env episode stats equal the unsharded run."""
    from gym_reinmav_amd.distributed import shard_range

    n, T = 10001, 64
    full = G.BatchedQuadrotor("quad3d", n, seed=4)
    full.rollout(T, mode="random", want=())
    fb = full.episode_buffers()
    parts = []
    for r in range(2):
        st, cnt = shard_range(n, r, 2)
        sh = G.BatchedQuadrotor("quad3d", cnt, seed=4, env_id_base=st)
        sh.rollout(T, mode="random", want=())
        parts.append(sh.episode_buffers())
        sh.close()
    for key in fb:
        assert np.array_equal(np.concatenate([parts[0][key], parts[1][key]]), fb[key])
    full.close()


@pytest.mark.parametrize("kind", ["quad3d", "quad3d_sl", "quad2d_sl"])
def test_per_env_domain_randomised_constants(G, kind):
    """rmav_set_env_param: per-env mass / load mass / tether length; each env must match the oracle run with
    that env's own constants (step and controller)."""
    rng = np.random.RandomState(8)
    n = 1500
    s = rng.uniform(-1, 1, (n, NS[kind])).astype(np.float32)
    a = rng.uniform(-10, 10, (n, NA[kind])).astype(np.float32)
    mass = rng.uniform(0.5, 2.0, n).astype(np.float32)
    lmass = rng.uniform(0.05, 0.4, n).astype(np.float32)
    tether = rng.uniform(0.4, 1.8, n).astype(np.float32)
    env = G.BatchedQuadrotor(kind, n, auto_reset=False, track_episodes=False)
    env.set_env_param("mass", mass)
    if kind.endswith("_sl"):
        env.set_env_param("load_mass", lmass)
        env.set_env_param("tether_length", tether)
    env.set_state(s)
    ctrl = env.control()
    obs, rew, done = env.step(a)
    q = O.default_params(kind)
    for i in range(0, n, 7):
        q.mass = float(mass[i])
        if kind.endswith("_sl"):
            q.load_mass, q.tether_length = float(lmass[i]), float(tether[i])
        o2, r, d, _ = O.step(kind, s[i].astype(np.float64), a[i].astype(np.float64), params=q)
        assert scaled_err(obs[i], o2).max() <= TOL
        if not near_threshold(kind, o2[None])[0]:
            assert bool(done[i]) == d
        assert scaled_err(ctrl[i], O.control(kind, s[i].astype(np.float64), params=q)).max() <= CTRL_TOL
    # back to the shared value
    env.set_env_param("mass", None)
    env.set_env_param("load_mass", None)
    env.set_env_param("tether_length", None)
    env.set_state(s)
    env.set_sbd(np.full(n, -1, np.int32))
    obs, _, _ = env.step(a)
    o2, _, _, _ = O.batch_step(kind, s.astype(np.float64), a.astype(np.float64))
    assert scaled_err(obs, o2).max() <= TOL
    env.close()


def test_rccl_all_gather_single_rank(G, tmp_path):
    """The per-rollout collective on the real backend: nccl (= RCCL) process group with one rank on the GPU box."""
    import socket
    import subprocess
    import sys

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    code = f"""
import os, sys
sys.path.insert(0, {os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'reinmav-gym_amd')!r})
import torch, torch.distributed as dist
import gym_reinmav_amd as g
from gym_reinmav_amd.distributed import all_gather_episode_stats, all_reduce_totals, make_sharded
os.environ['MASTER_ADDR']='127.0.0.1'; os.environ['MASTER_PORT']='{port}'
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
env = make_sharded('quad3d', 4099, 0, 1, device=0, seed=1)
env.rollout(96, mode='random', want=())
eb = env.episode_buffers(device_out=True)
r, l = all_gather_episode_stats(eb['last_return'], eb['last_length'], 4099)
tot = all_reduce_totals(env.episode_totals(), device=torch.device('cuda', 0))
torch.cuda.synchronize()
assert torch.equal(r, eb['last_return']) and torch.equal(l, eb['last_length'])
assert tot == env.episode_totals() and tot['episodes'] > 0
dist.destroy_process_group()
print('rccl ok')
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl ok" in r.stdout, r.stdout + r.stderr


def test_create_destroy_does_not_leak(G):
    """200 handles of 262 144 envs (all optional arrays allocated) created, stepped and destroyed: free device
    memory returns to where it started."""
    import torch

    def cycle(i):
        env = G.BatchedQuadrotor("quad3d_sl", 262144, seed=i, use_torch_stream=bool(i % 2))
        env.set_env_param("mass", np.full(262144, 1.1, np.float32))
        env.rollout(2, mode="controller", want=())
        if i % 50 < 2:
            env.rollout(2, mode="random", layout="aos")          # host-pointer path allocates scratch
        env.close()

    for i in range(2):          # first use loads code objects / creates runtime pools: not part of the measurement
        cycle(i)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for i in range(200):        # ~70 MB of device memory per handle: a per-handle leak would show as gigabytes
        cycle(i)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert abs(free0 - free1) < 64 << 20, (free0, free1)


def test_largest_supported_batch_and_limit(G):
    """N = 2^25 envs is the documented per-handle limit (32-bit buffer offsets); one more is refused."""
    A = G._abi
    h = C.c_void_p()
    assert A.lib().rmav_create(C.byref(h), A.QUAD3D, (1 << 25) + 1, 0, 0, 0, 0, None, None) == A.ERR_INVALID
    env = G.BatchedQuadrotor("quad2d", 1 << 25, seed=0, track_episodes=False)   # 2^25 x 5 floats = 640 MB of state
    env.rollout(3, mode="random", want=())
    s = env.get_state(layout="soa")
    assert np.isfinite(s).all() and s.shape == (5, 1 << 25)
    # the last env is addressed correctly: compare it with a 1-env handle carrying the same global id
    one = G.BatchedQuadrotor("quad2d", 1, seed=0, env_id_base=(1 << 25) - 1, track_episodes=False)
    one.rollout(3, mode="random", want=())
    assert np.array_equal(one.get_state(layout="soa")[:, 0], s[:, -1])
    env.close()
    one.close()


@pytest.mark.parametrize("kind", KINDS)
def test_fused_rollout_with_host_action_arrays_small_batches(G, kind):
    """Caller-provided actions as NumPy arrays: batches small enough for the pinned zero-copy path (the two-wavefront kernel
    then reads them over PCIe, four hand-overs ahead) and bigger ones (staged): same bits as single steps, both layouts."""
    rng = np.random.RandomState(2)
    for n in (1, 7, 64, 100, 3000):
        for T in (2, 3, 9, 24):
            for layout in ("soa", "aos"):
                acts = rng.uniform(*BOX[kind], (T, NA[kind], n) if layout == "soa" else (T, n, NA[kind])).astype(np.float32)
                res = []
                for fused in (True, False):
                    env = G.BatchedQuadrotor(kind, n, seed=3)
                    tr = env.rollout(T, mode="buffer", actions=acts, layout=layout, fused=fused, want=("obs", "rew", "done"))
                    res.append((tr, env.get_state(), env.episode_totals(), env.get_reset_counts()))
                    env.close()
                (a, sa, ta, ca), (b, sb, tb, cb) = res
                for k in a:
                    assert np.array_equal(a[k], b[k]), (kind, n, T, layout, k)
                assert np.array_equal(sa, sb) and np.array_equal(ca, cb) and ta["episodes"] == tb["episodes"]

