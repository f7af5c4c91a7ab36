mkdir -p gpurun_out/r06s
DQ_TRACE=1 timeout 400 python tests/manual/t_bsdiff_variants.py "" > gpurun_out/r06s/trace_chains8.log 2>&1
grep "create \|anchor scan:\|scan (device)\|framed\|new on device" gpurun_out/r06s/trace_chains8.log | cut -c1-200 | sed -n '20,40p'
grep "create " gpurun_out/r06s/trace_chains8.log | cut -c1-120
