#!/bin/bash
# Runs on the GPU box: kernel traces of mid-size sorts (fused and unfused histogram) -> gpurun_out/<tag>/
TAG=${1:-mid}
cd "$(dirname "$0")/.."
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for n in 16 20; do
  python3 $REPO/tools/mid_trace.py $n 100 > $OUT/wall_$n.txt 2>&1
  RSX_NO_FUSED_HIST=1 python3 $REPO/tools/mid_trace.py $n 100 > $OUT/wall_nofuse_$n.txt 2>&1
  rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/t$n -o t -- python3 $REPO/tools/mid_trace.py $n 100 > $OUT/t$n.log 2>&1
  RSX_NO_FUSED_HIST=1 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/u$n -o t -- python3 $REPO/tools/mid_trace.py $n 100 > $OUT/u$n.log 2>&1
done
find $OUT -name "*.db" -delete; find $OUT -name "*_agent_info.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete
cat $OUT/wall_*.txt
python3 - <<PY
import csv, glob
for f in sorted(glob.glob('$OUT/*/t_kernel_stats.csv')):
    print(f.split('/')[-2])
    for r in list(csv.DictReader(open(f)))[:8]:
        print('   ', r['Name'][:90].replace('unsigned int','u32').replace('unsigned long long','u64'), r['Calls'], r['AverageNs'])
PY
