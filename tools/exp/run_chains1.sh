mkdir -p gpurun_out/r06s
timeout 900 python -m pytest tests/test_gpu_bsdiff.py tests/test_gpu_faults.py -x -q -m gpu > gpurun_out/r06s/pytest_chains.log 2>&1
tail -3 gpurun_out/r06s/pytest_chains.log
DQ_TRACE=1 timeout 400 python tests/manual/t_bsdiff_variants.py "" > gpurun_out/r06s/trace_chains.log 2>&1
grep "create \|framed\|scan (device)" gpurun_out/r06s/trace_chains.log | cut -c1-200 | sed -n '9,16p'
grep "create " gpurun_out/r06s/trace_chains.log | cut -c1-130
timeout 300 python tests/manual/stress_bsdiff.py 150 624 > gpurun_out/r06s/stress_bsdiff_624.log 2>&1
tail -2 gpurun_out/r06s/stress_bsdiff_624.log
