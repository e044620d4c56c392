"""The NATIVE statistics exchange of librmav.so with world = 2 on ONE GPU (reference analogue: SubprocVecEnv workers
via gym_reinmav/run.py:89 and the MPI rank handling of run.py:18-21,177-182).  Real RCCL refuses two ranks on one device and
the test boxes have one, so the five RCCL entry points are served by tests/stub_rccl (shared-memory all-gather enqueued on the
caller's stream) through rmav_comm_use_library; the product code - rmav_allgather_stats_arm / _post / _result / _wait,
k_wait_arrivals, k_unpack_stats with rem != 0 padding, the 8-deep back pressure - is the shipped one.  The rank processes are
pure ctypes (no torch: torch would bring the real librccl into the process)."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "stub_rccl", "_build", "librccl_stub.so")

RANK_CODE = textwrap.dedent(r'''
    import ctypes as C, os, sys, time
    sys.modules["torch"] = None                      # keep torch (and with it the real librccl) out of this process
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "reinmav-gym_amd"))
    from gym_reinmav_amd import _abi as A
    L = A.lib()
    hip = C.CDLL("libamdhip64.so")
    def dmalloc(nbytes):
        p = C.c_void_p(); assert hip.hipMalloc(C.byref(p), C.c_size_t(nbytes)) == 0; return p
    def d2h(arr, p):
        assert hip.hipMemcpy(arr.ctypes.data_as(C.c_void_p), p, C.c_size_t(arr.nbytes), 2) == 0
    A.check(L.rmav_comm_use_library(STUB.encode()))
    idf = os.path.join(TMP, "uid")
    if RANK == 0:
        buf = (C.c_char * A.COMM_ID_BYTES)(); A.check(L.rmav_comm_unique_id(buf))
        open(idf + ".tmp", "wb").write(buf.raw); os.rename(idf + ".tmp", idf)
    t0 = time.time()
    while not os.path.exists(idf):
        assert time.time() - t0 < 60; time.sleep(0.01)
    raw = open(idf, "rb").read()
    base, rem = N_TOTAL // WORLD, N_TOTAL % WORLD
    count, start = base + (1 if RANK < rem else 0), RANK * base + min(RANK, rem)
    h = C.c_void_p()
    A.check(L.rmav_create(C.byref(h), A.QUAD3D, count, 0, 7, start, A.F_AUTO_RESET | A.F_TRACK_EPISODES, None, None))
    comm = C.c_void_p()
    A.check(L.rmav_comm_create(C.byref(comm), raw, RANK, WORLD, 0))
    A.check(L.rmav_comm_warmup(comm, 30.0))
    ret_d, len_d = dmalloc(4 * N_TOTAL), dmalloc(4 * N_TOTAL)
''')

GATHER_CODE = RANK_CODE + textwrap.dedent(r'''
    out = {}
    for it in range(ITERS):
        armed = it % 3 != 2                       # two armed posts, then a packed one, and so on
        if armed:
            A.check(L.rmav_allgather_stats_arm(h, comm, N_TOTAL))
        A.check(L.rmav_rollout(h, 16, A.ACT_RANDOM, None, None, None, None, None, A.DEVICE, A.SOA, 1))
        A.check(L.rmav_allgather_stats_post(h, comm, N_TOTAL))
        if it % 4 == 3:                           # let the host run ahead now and then: the back pressure of the buffer ring
            A.check(L.rmav_allgather_stats_wait(comm, 30.0))
        A.check(L.rmav_allgather_stats_result(h, comm, N_TOTAL, ret_d, len_d))
        A.check(L.rmav_sync(h))
        r, l = np.empty(N_TOTAL, np.float32), np.empty(N_TOTAL, np.int32)
        d2h(r, ret_d); d2h(l, len_d)
        mine_r, mine_l = np.empty(count, np.float32), np.empty(count, np.int32)
        A.check(L.rmav_episode_buffers(h, mine_r.ctypes.data_as(C.c_void_p), mine_l.ctypes.data_as(C.c_void_p), None, None, A.HOST))
        out[f"g_r{it}"], out[f"g_l{it}"], out[f"m_r{it}"], out[f"m_l{it}"] = r, l, mine_r, mine_l
        print("rank", RANK, "iteration", it, "armed" if armed else "packed", round(time.time() - t0, 3), "s", flush=True)
    # a burst of posts without results in between: more than the 8 buffer pairs of the ring
    for it in range(12):
        A.check(L.rmav_allgather_stats_arm(h, comm, N_TOTAL))
        A.check(L.rmav_rollout(h, 4, A.ACT_RANDOM, None, None, None, None, None, A.DEVICE, A.SOA, 1))
        A.check(L.rmav_allgather_stats_post(h, comm, N_TOTAL))
    A.check(L.rmav_allgather_stats_result(h, comm, N_TOTAL, ret_d, len_d))
    A.check(L.rmav_sync(h))
    r, l = np.empty(N_TOTAL, np.float32), np.empty(N_TOTAL, np.int32)
    d2h(r, ret_d); d2h(l, len_d)
    mine_r, mine_l = np.empty(count, np.float32), np.empty(count, np.int32)
    A.check(L.rmav_episode_buffers(h, mine_r.ctypes.data_as(C.c_void_p), mine_l.ctypes.data_as(C.c_void_p), None, None, A.HOST))
    out["g_rB"], out["g_lB"], out["m_rB"], out["m_lB"] = r, l, mine_r, mine_l
    np.savez(os.path.join(TMP, f"rank{RANK}.npz"), **out)
    A.check(L.rmav_comm_destroy(comm)); A.check(L.rmav_destroy(h))
    print("rank", RANK, "ok", flush=True)
''')

KILL_CODE = RANK_CODE + textwrap.dedent(r'''
    for it in range(3):
        A.check(L.rmav_allgather_stats_arm(h, comm, N_TOTAL))
        A.check(L.rmav_rollout(h, 8, A.ACT_RANDOM, None, None, None, None, None, A.DEVICE, A.SOA, 1))
        A.check(L.rmav_allgather_stats_post(h, comm, N_TOTAL))
        assert L.rmav_allgather_stats_wait(comm, 30.0) == A.OK
    if RANK == 1:
        os._exit(0)                               # the peer dies between two exchanges
    A.check(L.rmav_allgather_stats_arm(h, comm, N_TOTAL))
    A.check(L.rmav_rollout(h, 8, A.ACT_RANDOM, None, None, None, None, None, A.DEVICE, A.SOA, 1))
    t0 = time.time()
    A.check(L.rmav_allgather_stats_post(h, comm, N_TOTAL))          # enqueues, must not block
    rc = L.rmav_allgather_stats_wait(comm, 1.5)                     # HOST-side bounded wait: the gather can never complete
    dt = time.time() - t0
    print("rank 0 wait rc", rc, "after", round(dt, 2), "s", flush=True)
    assert rc == A.ERR_TIMEOUT and dt < 4.0, (rc, dt)
    # the env itself is untouched by the stuck collective: its stream keeps stepping
    A.check(L.rmav_rollout(h, 8, A.ACT_RANDOM, None, None, None, None, None, A.DEVICE, A.SOA, 1))
    A.check(L.rmav_sync(h))
    print("rank 0 ok", flush=True)
    os._exit(0)                                   # (the abandoned gather's wait kernel gives up by itself: RMAV_STUB_WAIT_S)
''')


@pytest.fixture(scope="module")
def stub(built):
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "stub_rccl")], check=True)
    assert os.path.exists(STUB)
    return STUB


def _run_ranks(code, tmp_path, n_total, iters=0, env_extra=None, world=2):
    procs = []
    for rank in range(world):
        head = f"ROOT={ROOT!r}; STUB={STUB!r}; TMP={str(tmp_path)!r}; RANK={rank}; WORLD={world}; N_TOTAL={n_total}; ITERS={iters}\n"
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
        procs.append(subprocess.Popen([sys.executable, "-c", head + code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o, e))
    return outs


@pytest.mark.timeout(600)
@pytest.mark.parametrize("n_total", [2 * 4096 + 1, 2 * 20000])
def test_two_ranks_on_one_gpu_native_exchange_equals_a_plain_gather(stub, tmp_path, n_total):
    """2 ranks x odd n_total (rem != 0: rank 0 owns one env more, the padded send slot) x 24 posts (armed and packed, with and
    without host waits) + a 12-post burst through the 8-deep ring: on EVERY rank the gathered statistics equal the
    concatenation of the ranks' own per-env statistics, bit for bit."""
    iters = 24
    outs = _run_ranks(GATHER_CODE, tmp_path, n_total, iters)
    for rc, o, e in outs:
        assert rc == 0 and "ok" in o, o[-1500:] + e[-3000:]
    z = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(2)]
    finished = 0
    for it in list(range(iters)) + ["B"]:
        exp_r = np.concatenate([z[0][f"m_r{it}"], z[1][f"m_r{it}"]])
        exp_l = np.concatenate([z[0][f"m_l{it}"], z[1][f"m_l{it}"]])
        assert exp_r.shape == (n_total,)
        for r in range(2):
            assert np.array_equal(z[r][f"g_r{it}"].view(np.int32), exp_r.view(np.int32)), (it, r)
            assert np.array_equal(z[r][f"g_l{it}"], exp_l), (it, r)
        finished = int((exp_l > 0).sum())
    assert finished > n_total // 2                  # most envs have finished an episode by the end: the payload is not all zeros


@pytest.mark.timeout(300)
def test_killed_rank_gives_a_timeout_not_a_hang(stub, tmp_path):
    """Rank 1 exits between two exchanges: rank 0's post returns at once, its HOST-side bounded wait reports RMAV_ERR_TIMEOUT
    within the bound, and its env keeps stepping (nothing of the collective ever entered the env's stream)."""
    outs = _run_ranks(KILL_CODE, tmp_path, 2 * 4096 + 1, env_extra={"RMAV_STUB_WAIT_S": "4"})
    rc, o, e = outs[0]
    assert rc == 0 and "rank 0 ok" in o, o[-1500:] + e[-3000:]


@pytest.mark.timeout(900)
def test_bench_two_ranks_one_gpu_native_exchange(stub):
    """bench.py --gpus 2 on ONE GPU through the NATIVE exchange (armed rollout launches, arrival words, k_wait_arrivals, the
    all-gather behind the C ABI - served by the stub): the gathered statistics equal a plain torch.distributed all-gather, and
    the sharded run finishes exactly the episodes of the unsharded one (RNG keyed by global env id)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    common = ["--steps", "30", "--warmup", "6", "--prewarm-ms", "0", "--cpu-seconds", "0", "--no-secondary", "--chunk", "32"]
    env = dict(os.environ, RMAV_BENCH_BACKEND="gloo", RMAV_BENCH_RCCL_LIB=STUB, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                         "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2",
                         "--envs-per-gpu", "16384"] + common, capture_output=True, text=True, timeout=850, cwd=ROOT, env=env)
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-3000:]
    j2 = json.loads([l for l in r2.stdout.splitlines() if l.startswith("{")][-1])
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--envs-per-gpu", "32768"] + common,
                        capture_output=True, text=True, timeout=850, cwd=ROOT)
    assert r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-3000:]
    j1 = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][-1])
    assert j2["n_gpus"] == 2 and j2["config"]["envs_total"] == 32768 == j1["config"]["envs_total"]
    assert "rmav_allgather_stats_post" in j2["config"]["parallelism"], j2["config"]["parallelism"]
    assert j2["config"]["rccl_ranks"] == 2          # rmav_comm_info -> the collective library's own ncclCommCount
    assert j2["config"]["exchange_equals_plain_all_gather"] is True
    assert j2["config"]["finished_episodes"] == j1["config"]["finished_episodes"] > 0
    assert j2["config"]["gathered_envs_with_a_finished_episode"] == j1["config"]["gathered_envs_with_a_finished_episode"] > 0


@pytest.mark.timeout(1200)
def test_bench_eight_ranks_one_gpu_is_the_eight_gpu_launch(stub):
    """The driver's 8-GPU command line - `torch.distributed.run --nproc-per-node 8 bench.py --gpus 8` - with eight rank PROCESSES on
    the one GPU (stub collective library, gloo process group): shards by global env id, native exchange with world = 8
    (rmav_comm_info reports 8 ranks), gathered statistics equal a plain all-gather, and the sharded run finishes exactly the episodes
    of the unsharded one.  What a real 8-GPU node adds is RCCL's transports and seven more devices - no code path of ours."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    common = ["--steps", "16", "--warmup", "4", "--prewarm-ms", "0", "--cpu-seconds", "0", "--no-secondary", "--chunk", "32"]
    env = dict(os.environ, RMAV_BENCH_BACKEND="gloo", RMAV_BENCH_RCCL_LIB=STUB, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r8 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr",
                         "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8",
                         "--envs-per-gpu", "8192"] + common, capture_output=True, text=True, timeout=1100, cwd=ROOT, env=env)
    assert r8.returncode == 0, r8.stdout[-2000:] + r8.stderr[-3000:]
    last = [l for l in r8.stdout.splitlines() if l.strip()][-1]
    assert last.startswith("{") and len(last) < 4096
    j8 = json.loads(last)
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--envs-per-gpu", "65536"] + common,
                        capture_output=True, text=True, timeout=850, cwd=ROOT)
    assert r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-3000:]
    j1 = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][-1])
    c = j8["config"]
    assert j8["n_gpus"] == 8 and c["envs_total"] == 65536 == j1["config"]["envs_total"] and j8["scaling"] == "weak"
    assert "rmav_allgather_stats_post" in c["parallelism"] and c["rccl_ranks"] == 8
    assert c["exchange_equals_plain_all_gather"] is True
    assert c["finished_episodes"] == j1["config"]["finished_episodes"] > 0
    assert c["gathered_envs_with_a_finished_episode"] == j1["config"]["gathered_envs_with_a_finished_episode"] > 0


@pytest.mark.timeout(900)
def test_bench_multi_gpu_line_carries_the_c3_leg(stub):
    """Without --no-secondary a multi-rank bench line also measures BASELINE configs[2]'s shape: 131 072 envs on every rank,
    barrier-bracketed, slowest rank counts (two ranks sharing the one GPU here, so only the plumbing is checked, not the rate)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RMAV_BENCH_BACKEND="gloo", RMAV_BENCH_RCCL_LIB=STUB, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "30",
                        "--warmup", "6", "--prewarm-ms", "0", "--cpu-seconds", "0"], capture_output=True, text=True, timeout=850, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["config"]["envs_per_gpu"] == 65536 and j["scaling"] == "weak"
    last = [l for l in r.stdout.splitlines() if l.strip()][-1]
    assert last.startswith("{") and len(last) < 4096                       # the compact line, also for N > 1
    assert j["config"]["config_id"] == "C2x2"
    row = j["legs"]["c3"]                                                  # short row in the line ...
    assert row["envs_total"] == 2 * 131072 and row["value"] > 0 and 0.0 < row["frac"] <= 1.0
    c3 = json.load(open(os.path.join(ROOT, j["detail"])))["other_modes"]["c3"]   # ... the full leg in the detail file
    assert c3["envs_total"] == 2 * 131072 and c3["value"] > 0 and 0.0 < c3["roofline_frac_slowest_rank"] <= 1.0
    assert c3["is_baseline_config_2"] is False
    assert c3["chunk_major"]["value"] > 0 and 0.0 < row["chunk_major"]["frac"] <= 1.0      # ... and with chunk-major trajectory arrays
