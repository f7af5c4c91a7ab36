mkdir -p gpurun_out/r06u
timeout 700 python tests/manual/stress.py 560 610 > gpurun_out/r06u/stress_610.log 2>&1
tail -1 gpurun_out/r06u/stress_610.log
timeout 700 python tests/manual/stress_bsdiff.py 560 633 > gpurun_out/r06u/stress_bsdiff_633.log 2>&1
tail -1 gpurun_out/r06u/stress_bsdiff_633.log
