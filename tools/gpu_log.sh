#!/bin/bash
# Runs on the GPU box (via gpurun): the log-route tests, then cfg 3's timings -> gpurun_out/<tag>/
TAG=${1:-log}
cd "$(dirname "$0")/.."
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_logroute.py -x -q 2>&1 | tail -40 > $OUT/log_tests.txt
cat $OUT/log_tests.txt
timeout 600 python tools/bench_configs.py --only Zipf --out $OUT/bench_configs.json > $OUT/bench_configs.txt 2>&1
tail -30 $OUT/bench_configs.txt | cut -c1-400
