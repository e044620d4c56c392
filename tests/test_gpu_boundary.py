"""The boundaries the reference's callers use, through the C ABI: the single-launch control()+step() entry points
(rmav_control_step / rmav_step_control), the zero-copy pinned path of host-pointer calls, the gym-shaped class's
cached control(), the VecEnv's buffer reuse, and the RCCL all-gather behind the C ABI (rmav_comm_* /
rmav_allgather_stats) with one rank."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle as O
from util import CTRL_TOL, KINDS, NA, NS, TOL, near_threshold, random_cases, scaled_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def G(built):
    import torch

    assert torch.cuda.is_available()
    import gym_reinmav_amd as g

    return g


@pytest.mark.parametrize("n", [1, 77, 5000, 40000])      # 1..5000: pinned zero-copy block; 40000: bulk staging
@pytest.mark.parametrize("kind", KINDS)
def test_control_step_equals_control_then_step(G, kind, n):
    s, _ = random_cases(kind, n, seed=3, wide=False)
    s *= 0.5
    a_env = G.BatchedQuadrotor(kind, n, auto_reset=False, track_episodes=False)
    b_env = G.BatchedQuadrotor(kind, n, auto_reset=False, track_episodes=False)
    a_env.set_state(s)
    b_env.set_state(s)
    act = a_env.control()
    obs, rew, done = a_env.step(act)
    act2, obs2, rew2, done2 = b_env.control_step()
    assert np.array_equal(act, act2) and np.array_equal(obs, obs2) and np.array_equal(rew, rew2) and np.array_equal(done, done2)
    # and against the oracle's controller + step
    ca = O.batch_control(kind, s.astype(np.float64))
    assert scaled_err(act2, ca).max() <= CTRL_TOL
    a_env.close()
    b_env.close()


@pytest.mark.parametrize("layout", ["aos", "soa"])
@pytest.mark.parametrize("n", [1, 300, 70000])
@pytest.mark.parametrize("kind", KINDS)
def test_step_control_equals_step_then_control(G, kind, n, layout):
    import torch

    s, a = random_cases(kind, n, seed=5, wide=False)
    s *= 0.5
    envs = [G.BatchedQuadrotor(kind, n, auto_reset=True, track_episodes=True, seed=2) for _ in range(2)]
    for e in envs:
        e.set_state(s)
    to = (lambda x: x) if layout == "aos" else (lambda x: np.ascontiguousarray(x.T))
    obs, rew, done = envs[0].step(to(a), layout=layout)
    nxt = envs[0].control(layout=layout)
    obs2, rew2, done2, nxt2 = envs[1].step_control(to(a), layout=layout)
    assert np.array_equal(obs, obs2) and np.array_equal(rew, rew2) and np.array_equal(done, done2)
    assert np.array_equal(nxt, nxt2)
    # device pointers: same bits
    envs[0].set_state(s)
    d_obs, d_rew, d_done, d_nxt = envs[0].step_control(torch.from_numpy(to(a)).cuda(), layout=layout)
    envs[1].set_state(s)
    h = envs[1].step_control(to(a), layout=layout)
    assert np.array_equal(d_nxt.cpu().numpy(), h[3]) and np.array_equal(d_rew.cpu().numpy(), h[1])
    for e in envs:
        e.close()


@pytest.mark.parametrize("kind", KINDS)
def test_gym_env_cached_control_is_the_real_control(G, kind):
    """The gym-shaped class answers control() from what the last step()'s launch computed; it must equal a fresh
    evaluation at every point of a closed loop, and be dropped when the state changes behind its back."""
    env_id = {"quad2d": "quadrotor2d-v0", "quad2d_sl": "quadrotor2d-slungload-v0", "quad3d": "quadrotor3d-v0",
              "quad3d_sl": "quadrotor3d-slungload-v0"}[kind]
    env = G.make(env_id, seed=4)
    env.reset()
    for i in range(60):
        a = env.control()
        fresh = env._batch.control()[0].astype(np.float64)
        assert np.array_equal(a, fresh), i
        o, r, d, info = env.step(a)
        assert o.dtype == np.float64 and isinstance(r, float) and isinstance(d, bool) and info == {}
        assert np.array_equal(o, env.state)
        if d:
            env.reset()
            assert not env._ctrl_valid
    st = env.state * 0.5
    env.state = st
    assert not env._ctrl_valid
    assert scaled_err(env.control(), O.batch_control(kind, st[None].astype(np.float32).astype(np.float64))[0]).max() <= CTRL_TOL
    with pytest.raises(ValueError):
        env.step(np.zeros(NA[kind] + 1))
    env.close()


@pytest.mark.parametrize("n", [1, 37, 64])
@pytest.mark.parametrize("kind", KINDS)
def test_host_steps_with_the_pinned_completion_word(G, kind, n):
    """Host-pointer steps of a batch that fits one wavefront end with the kernel publishing a sequence number in a pinned
    word the host spins on (instead of hipStreamSynchronize).  Same results as the same steps through DEVICE pointers (no
    completion word, the caller synchronises), over 200 steps with auto-reset and tracking, for rmav_step and
    rmav_step_control; device state agrees."""
    import torch

    lo, hi = (0.0, 10.0) if kind == "quad3d" else (-10.0, 10.0)
    acts = np.random.RandomState(1).uniform(lo, hi, (200, n, NA[kind])).astype(np.float32)
    out = []
    for host in (True, False):
        env = G.BatchedQuadrotor(kind, n, seed=4, auto_reset=True, track_episodes=True)
        rec = []
        for k in range(200):
            a = acts[k] if host else torch.from_numpy(acts[k]).cuda()
            r = env.step(a) if k % 2 else env.step_control(a)
            if not host:
                torch.cuda.synchronize()
            rec.append([np.array(x if host else x.cpu().numpy(), copy=True).astype(np.float32) for x in r])
        rec.append([env.get_state(), env.get_sbd().astype(np.float32), env.get_reset_counts().astype(np.float32)])
        out.append(rec)
        env.close()
    for ra, rb in zip(*out):
        for x, y in zip(ra, rb):
            assert np.array_equal(x, y)


def test_vec_env_reuse_buffers(G):
    import torch

    n = 4096
    a = G.QuadrotorVecEnv("quadrotor3d-v0", n, seed=6)
    b = G.QuadrotorVecEnv("quadrotor3d-v0", n, seed=6, reuse_buffers=True)
    oa, ob = a.reset(), b.reset()
    assert torch.equal(oa, ob)
    act = torch.empty((n, 4), device="cuda").uniform_(0, 10)
    prev = None
    for t in range(40):
        ra, rb = a.step(act), b.step(act)
        assert torch.equal(ra[0], rb[0]) and torch.equal(ra[1], rb[1]) and torch.equal(ra[2], rb[2])
        assert rb[2].dtype == torch.bool and ra[2].dtype == torch.bool
        if prev is not None:   # the previous step's tensors are still intact (two buffer sets alternate)
            assert torch.equal(prev[0], prev_copy[0]) and torch.equal(prev[1], prev_copy[1])
        prev, prev_copy = rb, (rb[0].clone(), rb[1].clone())
    a.close()
    b.close()


def test_rccl_allgather_behind_the_c_abi_single_rank(G):
    """rmav_comm_* + rmav_allgather_stats with one rank (RCCL refuses two ranks on one device): the gathered arrays
    equal the handle's own per-env statistics; shard mismatches and missing tracking are rejected."""
    code = f"""
import ctypes as C, sys
sys.path.insert(0, {os.path.join(ROOT, 'reinmav-gym_amd')!r})
import numpy as np, torch
import gym_reinmav_amd as g
A = g._abi
L = A.lib()
n = 4099
env = g.BatchedQuadrotor('quad3d', n, seed=1)
env.rollout(96, mode='random', want=())
uid = (C.c_char * A.COMM_ID_BYTES)()
A.check(L.rmav_comm_unique_id(uid))
comm = C.c_void_p()
A.check(L.rmav_comm_create(C.byref(comm), uid, 0, 1, 0))
ret = torch.empty(n, dtype=torch.float32, device='cuda'); ln = torch.empty(n, dtype=torch.int32, device='cuda')
for _ in range(3):
    A.check(L.rmav_allgather_stats(env._h, comm, n, C.c_void_p(ret.data_ptr()), C.c_void_p(ln.data_ptr())))
env.sync()
eb = env.episode_buffers()
assert np.array_equal(ret.cpu().numpy(), eb['last_return']) and np.array_equal(ln.cpu().numpy(), eb['last_length'])
assert (eb['last_length'] > 0).any()
assert L.rmav_allgather_stats(env._h, comm, n + 1, C.c_void_p(ret.data_ptr()), C.c_void_p(ln.data_ptr())) == A.ERR_INVALID
e2 = g.BatchedQuadrotor('quad3d', n, track_episodes=False)
assert L.rmav_allgather_stats(e2._h, comm, n, C.c_void_p(ret.data_ptr()), C.c_void_p(ln.data_ptr())) == A.ERR_INVALID
assert L.rmav_comm_destroy(comm) == 0 and L.rmav_comm_destroy(None) == A.ERR_INVALID
print('rccl abi ok')
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl abi ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_armed_exchange_equals_the_packed_one_single_rank(G):
    """rmav_allgather_stats_arm: the rollout launch writes the exchange's snapshot itself (every kernel family: one and
    two wavefronts, controller, caller actions, the three in-kernel actors; ragged batch sizes), _post adds nothing to the
    handle's stream, and the gathered arrays equal the handle's statistics after THAT launch - also when the next
    rollout is already queued behind it.  Launches that cannot carry it (single steps) fall back to the pack."""
    code = f"""
import ctypes as C, sys
sys.path.insert(0, {os.path.join(ROOT, 'reinmav-gym_amd')!r})
import numpy as np, torch
import gym_reinmav_amd as g
from gym_reinmav_amd.ppo import MlpPolicy, FusedPolicyCollector
A = g._abi
L = A.lib()
uid = (C.c_char * A.COMM_ID_BYTES)()
def gathered(env, comm, n):
    ret = torch.empty(n, dtype=torch.float32, device='cuda'); ln = torch.empty(n, dtype=torch.int32, device='cuda')
    A.check(L.rmav_allgather_stats_result(env._h, comm, n, C.c_void_p(ret.data_ptr()), C.c_void_p(ln.data_ptr())))
    env.sync()
    return ret.cpu().numpy(), ln.cpu().numpy()
checked = 0
for kind, n in (('quad3d', 4099), ('quad3d', 200000), ('quad2d_sl', 777), ('quad3d_sl', 8192)):
    env = g.BatchedQuadrotor(kind, n, seed=2)
    A.check(L.rmav_comm_unique_id(uid)); comm = C.c_void_p(); A.check(L.rmav_comm_create(C.byref(comm), uid, 0, 1, 0))
    acts = torch.empty((16, env.nA, n), device='cuda').uniform_(0, 10)
    pol = MlpPolicy(env.nS, env.nA).cuda()
    launches = [lambda: env.rollout(64, mode='random', want=(), device_out=True),
                lambda: env.rollout(16, mode='controller', want=(), device_out=True),
                lambda: env.rollout(16, mode='buffer', actions=acts, want=(), device_out=True),
                lambda: env.rollout(4, mode='random', want=(), device_out=True)]       # n_steps < 8: the one-wavefront kernel
    if n <= 8192:
        for kw in (dict(f32_mfma=False), dict(bf16_mfma=True), dict(f32_mfma=True), dict(f16_mfma=True)):
            col = FusedPolicyCollector(env, pol, 8, **kw)
            launches.append(col.collect)
        launches.append(FusedPolicyCollector(env, MlpPolicy(env.nS, env.nA, value_network='shared').cuda(), 8).collect)   # shared trunk
    for fn in launches:
        A.check(L.rmav_allgather_stats_arm(env._h, comm, n))
        assert L.rmav_allgather_stats_arm(env._h, comm, n) == A.ERR_INVALID          # one at a time
        fn()
        eb = env.episode_buffers()                                                     # statistics after THIS launch
        A.check(L.rmav_allgather_stats_post(env._h, comm, n))
        env.rollout(8, mode='random', want=(), device_out=True)                        # the next rollout, queued behind it
        r, l = gathered(env, comm, n)
        assert np.array_equal(r, eb['last_return']) and np.array_equal(l, eb['last_length']), (kind, n)
        checked += 1
    # a single-step launch cannot carry it: _post packs as before
    A.check(L.rmav_allgather_stats_arm(env._h, comm, n))
    env.step(np.zeros((n, env.nA), np.float32))
    eb = env.episode_buffers()
    A.check(L.rmav_allgather_stats_post(env._h, comm, n))
    r, l = gathered(env, comm, n)
    assert np.array_equal(r, eb['last_return']) and np.array_equal(l, eb['last_length'])
    assert (eb['last_length'] > 0).any()
    # fused = 0 with T > 1 (T single-step launches of the rollout kernel): none of them may take the snapshot - the first one
    # would freeze the statistics T - 1 steps early (ADVICE r02) - so _post packs after the last step
    for mode in ('random', 'controller'):
        A.check(L.rmav_allgather_stats_arm(env._h, comm, n))
        env.rollout(12, mode=mode, fused=False, want=(), device_out=True)
        eb = env.episode_buffers()
        A.check(L.rmav_allgather_stats_post(env._h, comm, n))
        assert L.rmav_allgather_stats_wait(comm, 30.0) == 0
        r, l = gathered(env, comm, n)
        assert np.array_equal(r, eb['last_return']) and np.array_equal(l, eb['last_length']), ('fused=0', mode)
    # a second fused rollout (or a step) between _arm and _post: the first launch's snapshot is stale, _post packs afresh
    for second in (lambda: env.rollout(16, mode='random', want=(), device_out=True),
                   lambda: env.step(np.full((n, env.nA), 9.0, np.float32))):
        A.check(L.rmav_allgather_stats_arm(env._h, comm, n))
        env.rollout(32, mode='random', want=(), device_out=True)
        second()
        eb = env.episode_buffers()
        A.check(L.rmav_allgather_stats_post(env._h, comm, n))
        r, l = gathered(env, comm, n)
        assert np.array_equal(r, eb['last_return']) and np.array_equal(l, eb['last_length']), 'stale snapshot'
    # more posts than buffer pairs, armed and plain mixed
    for i in range(20):
        if i % 3: A.check(L.rmav_allgather_stats_arm(env._h, comm, n))
        env.rollout(8, mode='random', want=(), device_out=True)
        A.check(L.rmav_allgather_stats_post(env._h, comm, n))
    eb = env.episode_buffers()
    r, l = gathered(env, comm, n)
    assert np.array_equal(r, eb['last_return']) and np.array_equal(l, eb['last_length'])
    # destroying the communicator while the handle is armed disarms the handle (no dangling pointer in the next rollout)
    A.check(L.rmav_allgather_stats_arm(env._h, comm, n))
    A.check(L.rmav_comm_destroy(comm))
    env.rollout(16, mode='random', want=(), device_out=True)
    env.sync()
    A.check(L.rmav_comm_unique_id(uid)); comm = C.c_void_p(); A.check(L.rmav_comm_create(C.byref(comm), uid, 0, 1, 0))
    A.check(L.rmav_allgather_stats_arm(env._h, comm, n))                             # armable again
    env.rollout(16, mode='random', want=(), device_out=True)
    eb = env.episode_buffers()
    A.check(L.rmav_allgather_stats_post(env._h, comm, n))
    r, l = gathered(env, comm, n)
    assert np.array_equal(r, eb['last_return']) and np.array_equal(l, eb['last_length'])
    # one armed exchange per communicator: a second handle cannot arm it (its bookkeeping names ONE handle)
    env2 = g.BatchedQuadrotor(kind, n, seed=3)
    A.check(L.rmav_allgather_stats_arm(env._h, comm, n))
    assert L.rmav_allgather_stats_arm(env2._h, comm, n) == A.ERR_INVALID
    env.rollout(16, mode='random', want=(), device_out=True)
    A.check(L.rmav_allgather_stats_post(env._h, comm, n))
    A.check(L.rmav_allgather_stats_arm(env2._h, comm, n))                            # free again after the post
    env2.close()                                                                     # destroying the armed handle disarms the communicator
    A.check(L.rmav_allgather_stats_arm(env._h, comm, n))
    env.rollout(16, mode='random', want=(), device_out=True)
    A.check(L.rmav_allgather_stats_post(env._h, comm, n))
    assert L.rmav_allgather_stats_wait(comm, 30.0) == 0
    A.check(L.rmav_comm_destroy(comm)); env.close()
# An armed rollout that sits behind 3 s of other work on its stream (a PPO update, another job) must not time out: the 2 s bound
# of the communicator stream's wait counts from the moment the armed launch BEGINS (ADVICE r03: it used to count from the post).
env = g.BatchedQuadrotor('quad3d', 4099, seed=2)
A.check(L.rmav_comm_unique_id(uid)); comm = C.c_void_p(); A.check(L.rmav_comm_create(C.byref(comm), uid, 0, 1, 0))
import time
st = torch.cuda.Stream(); env.use_stream(st)
with torch.cuda.stream(st):
    t0 = time.time(); torch.cuda._sleep(20_000_000); st.synchronize(); per = (time.time() - t0) / 20_000_000
    A.check(L.rmav_allgather_stats_arm(env._h, comm, 4099))
    t0 = time.time()
    torch.cuda._sleep(int(3.0 / per))                                                # ~3 s of queued work ahead of the armed launch
    env.rollout(16, mode='random', want=(), device_out=True)
    A.check(L.rmav_allgather_stats_post(env._h, comm, 4099))                         # the waiter starts NOW, the launch in ~3 s
    assert L.rmav_allgather_stats_wait(comm, 30.0) == 0, 'spurious time-out of a late armed launch'
    waited = time.time() - t0
    eb = env.episode_buffers()
    r, l = gathered(env, comm, 4099)
    assert np.array_equal(r, eb['last_return']) and np.array_equal(l, eb['last_length']) and waited > 2.2, waited
A.check(L.rmav_comm_destroy(comm)); env.close()
print('armed exchange ok', checked)
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "armed exchange ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_pack_stats_is_the_allgather_payload(G):
    import torch

    n, cmax = 1000, 1024
    env = G.BatchedQuadrotor("quad3d", n, seed=3)
    env.rollout(80, mode="random", want=())
    send = torch.full((2 * cmax,), -7, dtype=torch.int32, device="cuda")
    env.pack_stats(send)
    eb = env.episode_buffers()
    s = send.cpu().numpy()
    assert np.array_equal(s[:n].view(np.float32), eb["last_return"]) and np.array_equal(s[cmax:cmax + n], eb["last_length"])
    assert (s[n:cmax] == 0).all() and (s[cmax + n:] == 0).all()
    env.close()


def test_error_behaviour_of_the_new_entry_points(G):
    """Same conventions as the rest of the ABI: status codes + rmav_last_error, nothing throws or aborts."""
    import torch

    A = G._abi
    L = A.lib()
    env = G.BatchedQuadrotor("quad3d", 256, track_episodes=False)
    z = C.c_void_p(0)
    rew = torch.zeros((4, 256), device="cuda")
    done = torch.zeros((4, 256), dtype=torch.uint8, device="cuda")
    val = torch.zeros((5, 256), device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    assert L.rmav_gae(env._h, 0, p(rew), p(done), p(val), 0.99, 0.95, 1.0, p(rew), p(rew), z) == A.ERR_INVALID
    assert L.rmav_gae(env._h, 4, z, p(done), p(val), 0.99, 0.95, 1.0, p(rew), p(rew), z) == A.ERR_INVALID
    assert b"required" in L.rmav_last_error()
    assert L.rmav_gae(None, 4, p(rew), p(done), p(val), 0.99, 0.95, 1.0, p(rew), p(rew), z) == A.ERR_INVALID
    assert L.rmav_normalize(env._h, C.c_void_p(rew.data_ptr() + 4), 8, 0.0, 1.0) == A.ERR_INVALID     # not 16-byte aligned
    assert L.rmav_normalize(env._h, p(rew), 0, 0.0, 1.0) == A.OK
    a = np.zeros((256, 4), np.float32)
    o = np.zeros((256, 10), np.float32)
    assert L.rmav_step_control(env._h, a.ctypes.data, o.ctypes.data, None, None, None, A.HOST, A.AOS) == A.ERR_INVALID
    assert L.rmav_step_control(env._h, None, o.ctypes.data, None, None, a.ctypes.data, A.HOST, A.AOS) == A.ERR_INVALID
    assert L.rmav_pack_stats(env._h, 256, p(val)) == A.ERR_INVALID          # created without episode tracking
    tr = G.BatchedQuadrotor("quad3d", 256)
    assert L.rmav_pack_stats(tr._h, 128, p(val)) == A.ERR_INVALID           # cmax < num_envs
    assert L.rmav_comm_create(None, None, 0, 1, 0) == A.ERR_INVALID
    c = C.c_void_p()
    assert L.rmav_comm_create(C.byref(c), None, 0, 1, 0) == A.ERR_INVALID and not c.value
    assert L.rmav_allgather_stats_post(tr._h, None, 256) == A.ERR_INVALID
    rm = G.BatchedQuadrotor("reinmav", 8)
    s13 = np.zeros((8, 13), np.float32)
    assert L.rmav_step_control(rm._h, a[:8].ctypes.data, s13.ctypes.data, None, None, a[:8].ctypes.data, A.HOST, A.AOS) == A.ERR_INVALID
    assert b"ReinmavEnv" in L.rmav_last_error()
    for e in (env, tr, rm):
        e.close()
