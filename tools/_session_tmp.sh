mkdir -p gpurun_out/r04g
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -x -q -m gpu > gpurun_out/r04g/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r04g/pytest.log
python tools/vecenv_ab.py step_fast=0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04g/vecenv_ab.txt
for T in "" "--tune step_fast=0" "" "--tune step_fast=0"; do python bench.py --mode step --steps 4000 --warmup 500 --cpu-seconds 0 --no-secondary $T 2>/dev/null | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('step mode $T', round(j['roofline']['launch_ms_hip_events']*1e3,3), 'us')"; done | tee gpurun_out/r04g/step_fast.txt
