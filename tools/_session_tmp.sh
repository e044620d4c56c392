mkdir -p gpurun_out/r04f
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_boundary.py -x -q -m gpu > gpurun_out/r04f/pytest1.log 2>&1; echo "pytest api/boundary rc=$?"; tail -4 gpurun_out/r04f/pytest1.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "closed_loop or repeat_bit" > gpurun_out/r04f/pytest2.log 2>&1; echo "pytest parity rc=$?"; tail -12 gpurun_out/r04f/pytest2.log
python tools/vecenv_ab.py step_store=1 step_store=2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04f/vecenv_ab.txt
for T in "" "--tune step_store=1" "--tune step_store=2"; do python bench.py --mode step --steps 4000 --warmup 500 --cpu-seconds 0 --no-secondary $T 2>/dev/null | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('step mode $T', round(j['roofline']['launch_ms_hip_events']*1e3,3), 'us')"; done | tee gpurun_out/r04f/step_store.txt
