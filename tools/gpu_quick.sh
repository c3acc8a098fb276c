#!/bin/bash
# Runs on the GPU box (via gpurun): the hybrid tests, the size sweep, bench.py and a kernel trace of it -> gpurun_out/<tag>/
TAG=${1:-quick}
cd "$(dirname "$0")/.."
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_hybrid.py -x -q 2>&1 | tail -30 > $OUT/hybrid_tests.txt
python tools/size_sweep.py > $OUT/size_sweep.txt 2>&1
python bench.py --no-cpu-baseline > $OUT/bench.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o trace -- python3 $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1
find $OUT -name "*.db" -delete
rm -f $OUT/trace/trace_kernel_trace.csv $OUT/trace/trace_agent_info.csv
cat $OUT/hybrid_tests.txt
tail -1 $OUT/bench.txt | cut -c1-2000
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$OUT/trace/trace_kernel_stats.csv')))
for r in rows[:16]:
    print(r['Name'][:100].replace('unsigned int','u32').replace('unsigned long long','u64'), r['Calls'], r['AverageNs'])
PY
