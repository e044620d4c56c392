"""bench.py end to end on the GPU box: the single-process line (roofline fraction is a real HBM fraction, the other
modes and the CPU baseline are present) and the multi-rank path under torch.distributed.run with two ranks on the
one GPU (statistics carried by gloo, because RCCL refuses two ranks on one device): n_gpus, the gathered length
and the shard-invariance of the episode statistics."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(out):
    """The contract: the LAST stdout line is one JSON object of less than 4 KB (the driver keeps an 8 KB tail; round 4's
    22 KB line was lost to it), and it is the only JSON line."""
    lines = [l for l in out.splitlines() if l.strip()]
    assert lines and lines[-1].startswith("{"), out[-3000:]
    assert len([l for l in lines if l.startswith("{")]) == 1, out[-3000:]
    assert len(lines[-1]) < 4096, len(lines[-1])
    return json.loads(lines[-1])


def _detail(j):
    path = os.path.join(ROOT, j["detail"])
    assert os.path.exists(path), j["detail"]
    return json.load(open(path))


@pytest.mark.timeout(600)
def test_bench_driver_command_line_is_compact(built, tmp_path):
    """The driver's own command (python bench.py --gpus 1 --steps 20 --warmup 5, default legs): last stdout line < 4 KB with the
    contract keys, roofline.frac and cpu_baseline.value; the per-step kernel has a roofline row at 65 536, 262 144 and 1 048 576 envs."""
    import time

    d = str(tmp_path / "detail.json")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--detail", d],
                       capture_output=True, text=True, timeout=550, cwd=ROOT)
    el = time.time() - t0
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    j = _line(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "prewarm_ms", "prewarm_launches", "detail"):
        assert k in j, k
    assert j["steps"] == 20 and j["warmup"] == 5 and j["n_gpus"] == 1 and j["config"]["config_id"] == "C2"
    rf = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_needed", "bytes_per_launch", "launch_ms_hip_events"):
        assert k in rf, k
    assert 0.0 < rf["frac"] <= 1.0 and rf["peak"] == 8000.0 and rf["bytes_per_launch"] == 65536 * (64 * 61 + 104)
    cb = j["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] == 1 and cb["kind"] == "port" and cb["unit"] == "env-steps/s"
    # timed region = steps x ms_per_step fits in the process's run time by a wide margin
    assert j["steps"] * j["ms_per_step"] * 1e-3 < el
    legs = j["legs"]
    for k in ("step", "step_262144", "step_1048576", "sustained", "c3_shard", "c3_shard_chunked", "c4"):   # configs[2]'s shard and configs[3] ride along
        assert 0.0 < legs[k]["frac"] <= 1.0, (k, legs[k])
    assert legs["policy_rollout"]["f16_shared"]["value"] > legs["policy_rollout"]["f16_mfma"]["value"] > 0     # configs[4]'s per-GPU shard rides along too
    assert j["value_sustained"] == pytest.approx(legs["sustained"]["value"], rel=1e-3)
    assert len(json.dumps(j, separators=(",", ":"))) < 4000 and el < 60.0      # (~30 s: 10 s + 3 s of CPU baselines, 11 s sustained)
    mt = j["cpu_mt"]
    assert mt["cores"] <= mt["cores_available"] and mt["value"] > 0
    full = json.load(open(d))
    assert full["other_modes"]["step_1048576"]["roofline"]["bytes_per_launch"] == 1048576 * 101
    assert full["other_modes"]["sustained"]["seconds"] >= 10.0              # long enough for a 5 s busy sampler to see twice
    assert full["host_cpus"]["usable"] >= 1 and "cgroup_source" in full["host_cpus"]
    assert len(r.stderr) < 2000, r.stderr[-2000:]     # stderr stays quiet: the driver's tail appends it to stdout's


@pytest.mark.timeout(900)
def test_bench_single_process_line(built):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "300", "--warmup", "50", "--cpu-seconds", "1", "--secondary", "all"],
                       capture_output=True, text=True, timeout=850, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    short = _line(r.stdout)
    assert 0.0 < short["roofline"]["frac"] <= 1.0 and short["legs"]["c4"]["frac"] > 0 and short["legs"]["policy_rollout"]["bf16_mfma"]["value"] > 0
    j = _detail(short)            # the full record: every leg in full
    assert j["value"] == short["value"] and j["roofline"]["frac"] == short["roofline"]["frac"]
    assert j["n_gpus"] == 1 and j["steps"] == 300 and j["unit"] == "env-steps/s" and j["scaling"] == "weak"
    assert "configs[1]" in j["config"]["workload"] and j["config"]["envs_per_gpu"] == 65536
    assert j["config"]["trajectory_ring"] >= 5
    rf = j["roofline"]
    assert 0.0 < rf["frac"] <= 1.0, rf
    assert rf["bytes_per_launch"] == 65536 * (64 * 61 + 104)
    assert abs(rf["achieved"] - rf["bytes_per_launch"] / (rf["launch_ms_hip_events"] * 1e-3) / 1e9) < 1e-6 * rf["achieved"]
    assert j["value"] >= 50e6                                    # BASELINE target: >= 50 M env-steps/s on one GPU
    om = j["other_modes"]
    assert "error" not in om, om
    assert 0.0 < om["rollout_in_place"]["roofline_frac"] <= 1.0
    st = om["step"]["roofline"]                                  # the per-step path carries its own roofline object
    assert 0.0 < st["frac"] <= 1.0 and st["bytes_per_launch"] == 65536 * 101 and st["bound"] == "hbm"
    for leg, kind_n in (("c3_shard", 131072 * (64 * 61 + 104)), ("c3_shard_chunked", 131072 * (64 * 61 + 104)),
                        ("c4", 262144 * (64 * 85 + 152))):   # the other single-GPU configs
        lr = om[leg]["roofline"]
        assert lr["bytes_per_launch"] == kind_n and 0.0 < lr["frac"] <= 1.0, (leg, lr)
        assert om[leg]["finished_episodes"] > 0
    for actor in ("fp32_valu", "fp32_mfma", "bf16_mfma"):
        pr = om["policy_rollout"][actor]["roofline"]
        assert 0.0 < pr["hbm"]["frac"] <= 1.0 and all(0.0 < v["frac"] <= 1.0 for k, v in pr.items())
    assert j["prewarm_launches"] > 0 and j["prewarm_ms"] == 40.0
    for path in [j["roofline"]["traffic_source"], st["traffic_source"]]:   # every path the line names exists in the repo
        assert path is None or os.path.exists(os.path.join(ROOT, path.split(" ")[0])), path
    assert om["gym1"]["us_per_iteration_control_plus_step"] > 0 and om["vecenv"]["fresh_tensors_per_step"]["us_per_step"] > 0
    assert om["policy_rollout"]["fp32_mfma"]["env_steps_per_s"] > om["policy_rollout"]["fp32_valu"]["env_steps_per_s"] > 0
    assert om["policy_rollout"]["bf16_mfma"]["env_steps_per_s"] > 0
    assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["cores"] == 1
    ds = j["device_state"]                                       # socket power / cap / shader clock while the headline ran
    assert ds["samples"] == 0 or (0 < ds["power_w_mean"] <= ds["power_w_max"] and ds["sclk_mhz_mean"] > 0)


@pytest.mark.timeout(900)
def test_bench_two_ranks_on_one_gpu(built):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    common = ["--steps", "12", "--warmup", "3", "--cpu-seconds", "0", "--no-secondary", "--chunk", "32"]
    env = dict(os.environ, RMAV_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                         "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2",
                         "--envs-per-gpu", "8192"] + common, capture_output=True, text=True, timeout=850, cwd=ROOT, env=env)
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-3000:]
    j2 = _line(r2.stdout)
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--envs-per-gpu", "16384"] + common,
                        capture_output=True, text=True, timeout=850, cwd=ROOT)
    assert r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-3000:]
    j1 = _line(r1.stdout)
    assert j2["n_gpus"] == 2 and j1["n_gpus"] == 1
    assert j2["config"]["envs_total"] == 16384 == j1["config"]["envs_total"]
    assert "all-gather" in j2["config"]["parallelism"]
    # keyed by global env id: the sharded run finishes exactly the episodes of the unsharded one
    assert j2["config"]["finished_episodes"] == j1["config"]["finished_episodes"] > 0
    assert j2["config"]["gathered_envs_with_a_finished_episode"] == j1["config"]["gathered_envs_with_a_finished_episode"] > 0
    assert j2["config"]["exchange_equals_plain_all_gather"] is True
    assert 0.0 < j2["roofline"]["frac"] <= 1.0


@pytest.mark.timeout(900)
def test_bench_under_torchrun_single_rank_rccl(built):
    """One rank under torch.distributed.run: the real backend (nccl = RCCL), the overlapped per-launch all-gather."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "40",
                        "--warmup", "10", "--cpu-seconds", "0"], capture_output=True, text=True, timeout=850, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    j = _line(r.stdout)
    assert j["n_gpus"] == 1 and j["config"]["gathered_envs_with_a_finished_episode"] > 0
    assert j["config"]["exchange_equals_plain_all_gather"] is True
    assert 0.0 < j["roofline"]["frac"] <= 1.0
    assert "rmav_allgather_stats_post" in j["config"]["parallelism"], j["config"]["parallelism"]
    # the torch.distributed exchange (the fallback) gives the same statistics
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                         "127.0.0.1", "--master-port", str(port + 1), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "40",
                         "--warmup", "10", "--cpu-seconds", "0"], capture_output=True, text=True, timeout=850, cwd=ROOT,
                        env=dict(os.environ, RMAV_BENCH_EXCHANGE="torch"))
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-3000:]
    j2 = _line(r2.stdout)
    assert "all_gather_into_tensor" in j2["config"]["parallelism"]
    assert j2["config"]["gathered_envs_with_a_finished_episode"] == j["config"]["gathered_envs_with_a_finished_episode"]
    assert j2["config"]["finished_episodes"] == j["config"]["finished_episodes"]


@pytest.mark.timeout(1200)
def test_bench_two_gpus_native_exchange(built):
    """Needs >= 2 GPUs (self-skips on the 1-GPU test box): bench.py --gpus 2 under torch.distributed.run with the NATIVE
    exchange (RCCL from librmav.so: armed rollout launches, arrival words, k_wait_arrivals, ncclAllGather over xGMI) -
    the gathered statistics equal a plain torch.distributed all-gather, and the sharded run finishes exactly the
    episodes of the unsharded one (RNG keyed by global env id)."""
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs at least 2 GPUs")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    common = ["--steps", "40", "--warmup", "10", "--prewarm-ms", "0", "--cpu-seconds", "0", "--no-secondary"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                         "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2",
                         "--envs-per-gpu", "65536"] + common, capture_output=True, text=True, timeout=1100, cwd=ROOT, env=env)
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-3000:]
    j2 = _line(r2.stdout)
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--envs-per-gpu", "131072"] + common,
                        capture_output=True, text=True, timeout=850, cwd=ROOT)
    assert r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-3000:]
    j1 = _line(r1.stdout)
    assert j2["n_gpus"] == 2 and j2["config"]["envs_total"] == 131072 == j1["config"]["envs_total"]
    assert "rmav_allgather_stats_post" in j2["config"]["parallelism"], j2["config"]["parallelism"]
    assert j2["config"]["rccl_ranks"] == 2
    assert j2["config"]["exchange_equals_plain_all_gather"] is True
    assert j2["config"]["finished_episodes"] == j1["config"]["finished_episodes"] > 0
    assert j2["config"]["gathered_envs_with_a_finished_episode"] == j1["config"]["gathered_envs_with_a_finished_episode"] > 0
    assert 0.0 < j2["roofline"]["frac"] <= 1.0


@pytest.mark.timeout(1800)
def test_bench_eight_gpus_is_baseline_c3(built):
    """Needs 8 GPUs (self-skips elsewhere): `bench.py --gpus 8 --envs-per-gpu 131072` IS BASELINE configs[2] (C3: 1 048 576 envs as 8
    contiguous shards) - native exchange, RCCL reports 8 ranks, the gathered statistics equal a plain all-gather; and the driver's
    own weak-scaling command (`--gpus 8`, 65 536 envs per GPU) carries the C3 leg.  No rate is asserted: the first real scaling
    curve must need zero edits, nothing more."""
    import torch

    if torch.cuda.device_count() < 8:
        pytest.skip("needs 8 GPUs")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for extra, cid in ((["--envs-per-gpu", "131072"], "C3"), ([], "C2x8")):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                            "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "200", "--warmup", "50",
                            "--cpu-seconds", "0"] + extra, capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        j = _line(r.stdout)
        c = j["config"]
        assert j["n_gpus"] == 8 and c["config_id"] == cid and c["envs_total"] == 8 * c["envs_per_gpu"]
        assert "rmav_allgather_stats_post" in c["parallelism"], c["parallelism"]
        assert c["rccl_ranks"] == 8 and c["exchange_equals_plain_all_gather"] is True
        assert c["gathered_envs_with_a_finished_episode"] > 0 and 0.0 < j["roofline"]["frac"] <= 1.0
        if cid == "C2x8":
            assert j["legs"]["c3"]["envs_total"] == 1048576 and _detail(j)["other_modes"]["c3"]["is_baseline_config_2"] is True
