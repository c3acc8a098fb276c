#!/bin/bash
# Runs on the GPU box (via gpurun): the parity, fuzz and soak tests once per switch that takes a kernel family out of the routes --
# what runs instead must sort as well.  -> gpurun_out/<tag>/switch_matrix.txt
TAG=${1:-switches}
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
: > $OUT/switch_matrix.txt
for sw in NONE RSX_NO_AUX_SLOTS RSX_NO_NARROW_LEVEL1 RSX_NO_NARROW_SLOTS RSX_NO_PASS64A RSX_NO_PASS32A RSX_NO_PASS16A RSX_NO_LEAF16 RSX_NO_BLIND RSX_NO_LOG RSX_NO_HYBRID; do
  if [ $sw = NONE ]; then E=""; else E="$sw=1"; fi
  R=$(env $E timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_soak.py -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -1)
  echo "$sw=1: $R" | tee -a $OUT/switch_matrix.txt
done
