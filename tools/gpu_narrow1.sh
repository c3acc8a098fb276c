#!/bin/bash
# Runs on the GPU box (via gpurun): the tests of 8-byte keys on the routes without a histogram, then cfg 3's timings -> gpurun_out/<tag>/
TAG=${1:-narrow1}
cd "$(dirname "$0")/.."
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
timeout 1800 python -m pytest tests/test_gpu_routes.py tests/test_gpu_async_routes.py tests/test_gpu_hybrid.py tests/test_gpu_fullsize.py -x -q \
  -k "u64 or cfg3 or constant_columns or blind" 2>&1 | tail -30 > $OUT/tests.txt
cat $OUT/tests.txt
timeout 600 python tools/bench_configs.py --only cfg3 --out $OUT/bench_configs.json > $OUT/bench_configs.txt 2>&1
tail -12 $OUT/bench_configs.txt | cut -c1-330
AB_MASK=0xFFFFFFFFFF timeout 600 python tools/ab_sizes.py RSX_NO_NARROW_LEVEL1 1 0 u64 24 48 64 96 128 256 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_sizes.txt
