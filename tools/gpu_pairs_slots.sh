#!/bin/bash
# Runs on the GPU box (via gpurun): the rank / key + payload tests of the routes, the footprint probe, cfg 4's timings -> gpurun_out/<tag>/
TAG=${1:-pslots}
cd "$(dirname "$0")/.."
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_routes.py tests/test_gpu_async_routes.py tests/test_gpu_compact.py tests/test_gpu_hybrid.py -x -q \
  -k "rank or pairs or packed or payload" 2>&1 | tail -30 > $OUT/tests.txt
cat $OUT/tests.txt
timeout 900 python tools/footprint_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/footprint_probe.txt
cat $OUT/footprint_probe.txt
timeout 600 python tools/bench_configs.py --only cfg4 --out $OUT/bench_configs.json > $OUT/bench_configs.txt 2>&1
tail -12 $OUT/bench_configs.txt | cut -c1-300
