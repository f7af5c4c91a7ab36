mkdir -p gpurun_out/r06s
DQ_TRACE=1 timeout 400 python tests/manual/t_bsdiff_variants.py "" > gpurun_out/r06s/trace_chains8.log 2>&1
grep "create \|emitter" gpurun_out/r06s/trace_chains8.log | cut -c1-220 
cat /sys/kernel/mm/transparent_hugepage/enabled
