mkdir -p gpurun_out/r04e
export RMAV_STUB_WAIT_S=5
timeout 900 python -m pytest tests/test_gpu_stub_rccl.py tests/test_gpu_boundary.py tests/test_gpu_multiprocess.py -x -q -m gpu > gpurun_out/r04e/pytest.log 2>&1; echo "pytest rc=$?"; tail -30 gpurun_out/r04e/pytest.log
