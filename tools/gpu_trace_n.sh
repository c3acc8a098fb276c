#!/bin/bash
# Runs on the GPU box: wall time and rocprofv3 kernel stats of tools/mid_trace.py at the given sizes -> gpurun_out/<tag>/
# usage: tools/gpu_trace_n.sh <tag> <n> [<n> ...]
TAG=$1; shift
cd "$(dirname "$0")/.."
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for n in "$@"; do
	python3 $REPO/tools/mid_trace.py $n 60 2>&1 | grep -v amdgpu.ids | tee $OUT/wall_$n.txt
	rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/t$n -o t -- python3 $REPO/tools/mid_trace.py $n 60 > $OUT/t$n.log 2>&1
	python3 - $OUT/t$n <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/t_kernel_stats.csv', recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if 'fill_splitmix' not in r['Name']]
    for r in rows[:12]:
        name = r['Name'][:100].replace('unsigned int', 'u32').replace('unsigned long long', 'u64').replace('unsigned short', 'u16')
        print('   %-100s calls %5s  avg %9.1f us' % (name, r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
find $OUT -name "*.db" -delete; find $OUT -name "*_agent_info.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete
