mkdir -p gpurun_out/r06s
timeout 600 python tests/manual/t_bsdiff_big.py 128 4000 > gpurun_out/r06s/bsdiff_big.log 2>&1
grep "MiB, 4000\|anchor scan: first\|joined at\|scan (device)\|new on device\|end of file\|framed\|raw streams" gpurun_out/r06s/bsdiff_big.log | cut -c1-150 | awk 'NR<=3 || /chain 15|chain 1 |end of|scan \(dev|framed|raw streams/'
timeout 400 python tests/manual/t_bsdiff_variants.py "" > gpurun_out/r06s/chains_lane.log 2>&1
grep "create " gpurun_out/r06s/chains_lane.log | cut -c1-130
