#!/bin/bash
# Runs on the GPU box: the round's measured lines at HEAD -> gpurun_out/<tag>/ (copied to profiles/rNN/ afterwards).
#   bench.py (the driver's line), tools/bench_configs.py (cfg 3 / cfg 4), tools/size_sweep.py (2^8 .. 2^31 u32 keys),
#   tools/radix_bench --device (the reference's bench, radix_bench.cpp), tools/footprint_probe.py, tools/big_sort_check.py
# usage: tools/gpu_evidence.sh [tag]
TAG=${1:-evidence}
cd "$(dirname "$0")/.."
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
python bench.py > $OUT/bench_line.json 2> $OUT/bench_err.txt
python tools/bench_configs.py --out $OUT/bench_configs.json > $OUT/bench_configs.txt 2>&1
python tools/size_sweep.py 2>&1 | grep -v amdgpu.ids > $OUT/size_sweep.txt
(cd /tmp && $REPO/tools/radix_bench --device 0 --verify --min-time 0.2 2>&1 | grep -v amdgpu.ids > $OUT/radix_bench.txt)
python tools/footprint_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/footprint_probe.txt
(for a in "28 50000000" "28 130000000" "29 12345" "29 200000000" "30 7" "30 500000000" "31 4097"; do python tools/big_sort_check.py $a 3; done) 2>&1 | grep -v amdgpu.ids > $OUT/big_sizes.txt
python tools/show_bench_line.py $OUT/bench_line.json
cut -c1-200 $OUT/bench_configs.txt | grep config
tail -16 $OUT/size_sweep.txt
cat $OUT/footprint_probe.txt $OUT/big_sizes.txt
