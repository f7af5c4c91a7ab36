mkdir -p gpurun_out/r06w
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r06w/pytest_gpu.log 2>&1
tail -3 gpurun_out/r06w/pytest_gpu.log
timeout 400 python tests/manual/stress.py 100 608 > gpurun_out/r06w/stress_608.log 2>&1
tail -1 gpurun_out/r06w/stress_608.log
timeout 400 python tests/manual/stress_bsdiff.py 200 629 > gpurun_out/r06w/stress_bsdiff_629.log 2>&1
tail -1 gpurun_out/r06w/stress_bsdiff_629.log
timeout 400 python tests/manual/t_bsdiff_variants.py "" > gpurun_out/r06w/diff_create_pairs.log 2>&1
grep "create " gpurun_out/r06w/diff_create_pairs.log | cut -c1-130
timeout 300 python tests/manual/t_bsdiff_big.py 128 4000 2>&1 | grep "MiB, 4000\|raw streams" | cut -c1-200 > gpurun_out/r06w/diff_create_128MiB.log
cat gpurun_out/r06w/diff_create_128MiB.log
