mkdir -p gpurun_out/r06v
timeout 400 python tests/manual/stress.py 300 609 > gpurun_out/r06v/stress_609.log 2>&1
tail -1 gpurun_out/r06v/stress_609.log
timeout 400 python tests/manual/stress_bsdiff.py 300 631 > gpurun_out/r06v/stress_bsdiff_631.log 2>&1
tail -1 gpurun_out/r06v/stress_bsdiff_631.log
