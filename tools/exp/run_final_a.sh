mkdir -p gpurun_out/r06u
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r06u/pytest_gpu.log 2>&1
tail -3 gpurun_out/r06u/pytest_gpu.log
timeout 400 python tests/manual/stress_bsdiff.py 150 632 > gpurun_out/r06u/stress_bsdiff_632.log 2>&1
tail -1 gpurun_out/r06u/stress_bsdiff_632.log
timeout 400 python tests/manual/t_bsdiff_variants.py "" > gpurun_out/r06u/diff_create_pairs.log 2>&1
grep "create " gpurun_out/r06u/diff_create_pairs.log | cut -c1-130
