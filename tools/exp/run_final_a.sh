mkdir -p gpurun_out/r06y
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r06y/pytest_gpu.log 2>&1
tail -3 gpurun_out/r06y/pytest_gpu.log
timeout 400 python tests/manual/stress.py 150 606 > gpurun_out/r06y/stress_606.log 2>&1
tail -1 gpurun_out/r06y/stress_606.log
timeout 400 python tests/manual/stress_bsdiff.py 150 626 > gpurun_out/r06y/stress_bsdiff_626.log 2>&1
tail -1 gpurun_out/r06y/stress_bsdiff_626.log
