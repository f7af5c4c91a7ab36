mkdir -p gpurun_out/r06s
timeout 400 python tests/manual/stress_bsdiff.py 300 621 > gpurun_out/r06s/stress_bsdiff_621.log 2>&1
tail -5 gpurun_out/r06s/stress_bsdiff_621.log
