#!/bin/bash
# final evidence of the round: suite, smoke, bench lines, rocprofv3 + PMC
TAG=${1:-r02_final}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -4 > $OUT/device.txt
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket" >> $OUT/device.txt
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as e; e.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
echo "== pytest -m gpu"; timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu.log
echo "== bench (defaults)"; timeout 900 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "rc=$?"
echo "== bench (driver's K=20 W=5)"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1_k20.json 2> $OUT/bench_n1_k20.err; echo "rc=$?"
echo "== bench C3 shard, C4"; timeout 600 python bench.py --envs-per-gpu 131072 --cpu-seconds 0 --no-secondary > $OUT/bench_c3shard.json 2>/dev/null; timeout 600 python bench.py --kind quad3d_sl --envs-per-gpu 262144 --steps 500 --warmup 100 --cpu-seconds 0 --no-secondary > $OUT/bench_c4.json 2>/dev/null
echo "== bench under torchrun, 1 rank, C3 shard"; timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --envs-per-gpu 131072 > $OUT/bench_torchrun1_c3shard.json 2> $OUT/bench_dist1.err; echo "rc=$?"
python - <<PY
import json
for f in ("bench_n1", "bench_n1_k20", "bench_c3shard", "bench_c4", "bench_torchrun1_c3shard"):
    try:
        j = json.loads([l for l in open("$OUT/" + f + ".json") if l.startswith("{")][0])
        r = j["roofline"]
        print(f, round(j["value"] / 1e9, 2), "G/s", round(r["launch_ms_hip_events"] * 1e3, 2), "us kernel", "frac", round(r["frac"], 3))
        for k, v in j.get("other_modes", {}).items():
            print("   ", k, json.dumps(v)[:260])
    except Exception as e:
        print(f, "ERR", e)
PY
bash tools/profile_r02.sh r02c 2>&1 | grep -A12 "^## HBM-side"
