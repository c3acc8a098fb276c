#!/bin/bash
# scratch: the commands of the current gpurun call
set -x
cd /root/repo
mkdir -p gpurun_out/r02m
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "host" > gpurun_out/r02m/pytest_host.txt 2>&1
tail -5 gpurun_out/r02m/pytest_host.txt
timeout 600 tools/radix_bench --verify > gpurun_out/r02m/radix_bench.txt 2>&1
head -24 gpurun_out/r02m/radix_bench.txt
RSX_NO_HOST_SMALL=1 timeout 600 tools/radix_bench > gpurun_out/r02m/radix_bench_staged.txt 2>&1
head -12 gpurun_out/r02m/radix_bench_staged.txt
timeout 900 python -m pytest tests/test_gpu_cpp.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r02m/pytest_cpp_fuzz.txt 2>&1
tail -3 gpurun_out/r02m/pytest_cpp_fuzz.txt
