#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r4k
timeout 1500 python -m pytest tests/test_gpu_hybrid.py -x -q 2>&1 | tail -12 > gpurun_out/r4k/hybrid.txt
cat gpurun_out/r4k/hybrid.txt
