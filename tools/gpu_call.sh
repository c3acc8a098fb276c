#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r4e
timeout 1500 python -m pytest tests/test_gpu_routes.py -x -q -k "mid_size or every_leaf or clustered" 2>&1 | tail -5 > gpurun_out/r4e/routes.txt
cat gpurun_out/r4e/routes.txt
./tools/radix_bench --device 0 --verify > gpurun_out/r4e/radix_bench.txt 2>&1
grep -E "radix_sort/" gpurun_out/r4e/radix_bench.txt | head -20
python tools/size_sweep.py > gpurun_out/r4e/size_sweep.txt 2>&1
tail -24 gpurun_out/r4e/size_sweep.txt
