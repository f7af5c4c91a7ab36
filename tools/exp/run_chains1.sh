mkdir -p gpurun_out/r06s
timeout 900 python -m pytest tests/test_gpu_bsdiff.py tests/test_gpu_match_search.py -x -q -m gpu > gpurun_out/r06s/pytest_chains.log 2>&1
tail -3 gpurun_out/r06s/pytest_chains.log
timeout 400 python tests/manual/t_bsdiff_variants.py "" > gpurun_out/r06s/chains_lane.log 2>&1
grep "create " gpurun_out/r06s/chains_lane.log | cut -c1-130
timeout 100 python tests/manual/t_index_many.py 2>&1 | tail -2
