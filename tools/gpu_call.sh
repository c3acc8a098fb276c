#!/bin/bash
# the round-end sequence: GPU tests, smoke, bench (what the driver runs).  Full logs go to gpurun_out/<tag>/; the script
# fails if any of the three fails.
# usage: tools/gpu_call.sh [tag]     (default tag: final)
set -o pipefail
TAG=${1:-final}
cd "$(dirname "$0")/.."
OUT=gpurun_out/$TAG
mkdir -p $OUT
rc=0
timeout 2400 python -m pytest tests -x -q -m gpu --durations=25 > $OUT/pytest_gpu.txt 2>&1 || rc=1
tail -40 $OUT/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1 || rc=1
tail -3 $OUT/smoke.txt
timeout 600 python bench.py > $OUT/bench.txt 2>&1 || rc=1
tail -1 $OUT/bench.txt
exit $rc
