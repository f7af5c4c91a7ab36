mkdir -p gpurun_out/r06v
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r06v/pytest_gpu.log 2>&1
tail -3 gpurun_out/r06v/pytest_gpu.log
timeout 400 python tests/manual/stress_bsdiff.py 240 630 > gpurun_out/r06v/stress_bsdiff_630.log 2>&1
tail -1 gpurun_out/r06v/stress_bsdiff_630.log
timeout 400 python tests/manual/t_bsdiff_variants.py "" > gpurun_out/r06v/diff_create_pairs.log 2>&1
grep "create " gpurun_out/r06v/diff_create_pairs.log | cut -c1-130
timeout 400 python tests/manual/t_bsdiff_small.py "" "DQ_SCAN_MIN_SEG=1048576" > gpurun_out/r06v/diff_create_small_pairs.log 2>&1
grep -c create gpurun_out/r06v/diff_create_small_pairs.log
