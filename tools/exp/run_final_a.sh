mkdir -p gpurun_out/r06x
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r06x/pytest_gpu.log 2>&1
tail -3 gpurun_out/r06x/pytest_gpu.log
timeout 400 python tests/manual/stress.py 120 607 > gpurun_out/r06x/stress_607.log 2>&1
tail -1 gpurun_out/r06x/stress_607.log
timeout 400 python tests/manual/stress_bsdiff.py 150 628 > gpurun_out/r06x/stress_bsdiff_628.log 2>&1
tail -1 gpurun_out/r06x/stress_bsdiff_628.log
