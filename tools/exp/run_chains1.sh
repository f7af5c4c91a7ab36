mkdir -p gpurun_out/r06s
timeout 900 python -m pytest tests/test_gpu_bsdiff.py tests/test_gpu_faults.py -x -q -m gpu > gpurun_out/r06s/pytest_chains.log 2>&1
tail -8 gpurun_out/r06s/pytest_chains.log
