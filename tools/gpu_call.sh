#!/bin/bash
# scratch: the commands of the current gpurun call
set -x
cd /root/repo
mkdir -p gpurun_out/r02n
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r02n/pytest_gpu.txt 2>&1
tail -4 gpurun_out/r02n/pytest_gpu.txt
timeout 600 python tools/bench_types.py > gpurun_out/r02n/bench_types.txt 2>&1
cat gpurun_out/r02n/bench_types.txt
RSX_BENCH_TYPES="i16" RSX_NO_FILL_RUNS=1 timeout 600 python tools/bench_types.py
