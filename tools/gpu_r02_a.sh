#!/bin/bash
# round 2, first GPU session: the new tests, then the default bench line
TAG=${1:-r02_a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== new tests"
timeout 1700 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_gae.py tests/test_gpu_bench.py "tests/test_gpu_parity.py::test_ragged_last_wavefront_writes_nothing_out_of_bounds" "tests/test_gpu_ppo.py::test_c5_size_policy_rollout" -x -q > $OUT/pytest_new.log 2>&1
echo "pytest rc=$?"; tail -30 $OUT/pytest_new.log
echo "== bench"
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 3000 $OUT/bench.json; tail -5 $OUT/bench.err
