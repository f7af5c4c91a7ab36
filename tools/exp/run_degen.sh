mkdir -p gpurun_out/r06s
timeout 900 python tests/manual/t_degenerate.py 64 > gpurun_out/r06s/degenerate64.log 2>&1
cat gpurun_out/r06s/degenerate64.log | cut -c1-250
