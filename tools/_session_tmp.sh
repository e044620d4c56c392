bash tools/profile_actors.sh r04_actors2 2>&1 | tail -12
python bench.py --steps 100 --warmup 20 --cpu-seconds 0 --secondary policy_rollout > gpurun_out/r04_actors2/bench_policy.json 2> gpurun_out/r04_actors2/bench_policy.err; tail -c 3000 gpurun_out/r04_actors2/bench_policy.json
