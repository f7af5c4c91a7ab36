mkdir -p gpurun_out/r06s
timeout 900 python -m pytest tests/test_gpu_bsdiff.py -x -q -m gpu > gpurun_out/r06s/pytest_chains.log 2>&1
tail -3 gpurun_out/r06s/pytest_chains.log
DQ_TRACE=1 timeout 400 python tests/manual/t_bsdiff_variants.py "" > gpurun_out/r06s/trace_chains.log 2>&1
grep "create \|framed\|scan (device)\|bzip2 block of" gpurun_out/r06s/trace_chains.log | cut -c1-200 | sed -n '20,34p'
grep "create " gpurun_out/r06s/trace_chains.log | cut -c1-130
