mkdir -p gpurun_out/r04h
python bench.py --steps 20 --warmup 5 > gpurun_out/r04h/bench_n1_k20.json 2> gpurun_out/r04h/bench_n1_k20.err; echo "bench rc=$?"; python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r04h/bench_n1_k20.json") if l.startswith("{")][-1])
print("value", j["value"] / 1e9, "frac", j["roofline"]["frac"])
for k, v in j.get("other_modes", {}).items():
    print(k, json.dumps(v)[:600])
print("cpu", json.dumps(j.get("cpu_baseline"))[:200], json.dumps(j.get("calibration"))[:200])
PY
tail -3 gpurun_out/r04h/bench_n1_k20.err
bash tools/throttle_probe.sh r04h 2>&1 | tail -60
