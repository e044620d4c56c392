OUT=gpurun_out/r05_m; mkdir -p $OUT
echo "| envs | steps per launch | MB per launch | us per launch | frac |" > $OUT/tsweep.md; echo "|---|---|---|---|---|" >> $OUT/tsweep.md
for rep in 1 2; do
for N in 65536 131072 262144; do for T in 16 32 64 128; do
  S=$(( 65536 * 64 * 500 / (N * T) + 40 ))
  timeout 300 python bench.py --envs-per-gpu $N --chunk $T --steps $S --warmup $((S/4)) --cpu-seconds 0 --no-secondary --detail - 2>/dev/null | grep '^{' | \
    python -c "import json,sys; j=json.loads(sys.stdin.readline()); r=j['roofline']; print('| $N | $T | %.0f | %.2f | %.3f |' % (r['bytes_per_launch']/1e6, r['launch_ms_hip_events']*1e3, r['frac']))" >> $OUT/tsweep.md
done; done; done
cat $OUT/tsweep.md
