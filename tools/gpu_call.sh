#!/bin/bash
# scratch: the commands of the current gpurun call
set -x
cd /root/repo
mkdir -p gpurun_out/r02n
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_byte or sweep or contract or kat or golden" > gpurun_out/r02n/pytest_u8.txt 2>&1
tail -5 gpurun_out/r02n/pytest_u8.txt
timeout 600 python tools/bench_types.py > gpurun_out/r02n/bench_types.txt 2>&1
cat gpurun_out/r02n/bench_types.txt
RSX_NO_FILL_RUNS=1 timeout 600 python tools/bench_types.py 2>&1 | head -3
