#!/bin/bash
# round 2, GPU session J: full suite + smoke + the bench lines the driver will run
TAG=${1:-r02_j}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as e; e.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
echo "== pytest -m gpu"; timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
echo "== bench (defaults)"; timeout 900 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "rc=$?"
echo "== bench (driver's K=20 W=5)"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1_k20.json 2> $OUT/bench_n1_k20.err; echo "rc=$?"
echo "== bench under torchrun, 1 rank, C3 shard"; timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --envs-per-gpu 131072 --steps 1000 --warmup 100 > $OUT/bench_dist1_c3shard.json 2> $OUT/bench_dist1.err; echo "rc=$?"
python - <<PY
import json
for f in ("bench_n1", "bench_n1_k20", "bench_dist1_c3shard"):
    try:
        j = json.loads([l for l in open("$OUT/" + f + ".json") if l.startswith("{")][0])
        r = j["roofline"]
        print(f, round(j["value"] / 1e9, 2), "G/s", round(j["ms_per_step"] * 1e3, 2), "us/step wall", round(r["launch_ms_hip_events"] * 1e3, 2), "us kernel", "frac", round(r["frac"], 3), "traffic_frac", r["traffic_frac"])
        for k, v in j.get("other_modes", {}).items():
            print("   ", k, json.dumps(v)[:300])
    except Exception as e:
        print(f, "ERR", e)
PY
