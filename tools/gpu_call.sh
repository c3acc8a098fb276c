#!/bin/bash
# scratch: the commands of the current gpurun call
cd /root/repo
mkdir -p gpurun_out/r4a
timeout 900 python -m pytest tests/test_gpu_hybrid.py -x -q 2>&1 | tail -15 > gpurun_out/r4a/hybrid.txt
cat gpurun_out/r4a/hybrid.txt
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r4a/bench.txt 2>&1
tail -1 gpurun_out/r4a/bench.txt | cut -c1-1500
