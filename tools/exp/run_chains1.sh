mkdir -p gpurun_out/r06s
timeout 900 python -m pytest tests/test_gpu_bsdiff.py tests/test_gpu_faults.py -x -q -m gpu > gpurun_out/r06s/pytest_chains.log 2>&1
tail -5 gpurun_out/r06s/pytest_chains.log
DQ_TRACE=1 timeout 400 python tests/manual/t_bsdiff_variants.py "" > gpurun_out/r06s/trace_chains.log 2>&1
grep "create \|anchor scan:\|scan (device)" gpurun_out/r06s/trace_chains.log | cut -c1-260 | sed -n '12,24p'
grep "create " gpurun_out/r06s/trace_chains.log | cut -c1-120
timeout 200 python tests/manual/stress_bsdiff.py 120 622 > gpurun_out/r06s/stress_bsdiff_622.log 2>&1
tail -3 gpurun_out/r06s/stress_bsdiff_622.log
