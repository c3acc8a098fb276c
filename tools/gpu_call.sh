#!/bin/bash
# scratch: the commands of the current gpurun call
set -x
cd /root/repo
mkdir -p gpurun_out/r02o
RSX_FUZZ_CHUNKS=40 RSX_FUZZ_SEED=777001 RSX_FUZZ_MAXN=6000000 timeout 3000 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r02o/fuzz_big.txt 2>&1
tail -3 gpurun_out/r02o/fuzz_big.txt
