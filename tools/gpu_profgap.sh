#!/bin/bash
# One box, one call: the headline's three kernels timed (a) by the library's HIP events in a plain run, (b) by the same HIP events while
# rocprofv3 --kernel-trace is attached, (c) by the trace itself, (d) by HIP events / the trace's dispatch durations under a --pmc pass.
TAG=${1:-profgap}
cd "$(dirname "$0")/.."
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
show() { python3 - "$1" "$2" <<'PY'
import json,sys
name,f=sys.argv[1],sys.argv[2]
try:
    j=json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
    pk=j["roofline"]["per_kernel"]
    print("%-34s HIP events: step %.4f ms | "%(name,j["ms_per_step"])+"  ".join("%s %.4f"%(k,v["avg_launch_ms"]) for k,v in pk.items()))
except Exception as e:
    print(name,"FAILED",e)
PY
}
python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/plain.txt 2>&1; show "plain" $OUT/plain.txt
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o trace -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/traced.txt 2>&1; show "under --kernel-trace" $OUT/traced.txt
rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc -o pmc -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/pmc.txt 2>&1; show "under --pmc (2 SQ counters)" $OUT/pmc.txt
python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/plain2.txt 2>&1; show "plain again" $OUT/plain2.txt
python3 - $OUT <<'PY'
import csv,sys,glob
out=sys.argv[1]
for f in glob.glob(out+"/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Name"]
        if any(k in n for k in ("pass32a","pass16a","leaf16_kernel","precheck","seg_tiles","slack_plan")):
            print("trace: %-70s calls %s avg %.1f us"%(n[:70],r["Calls"],float(r["AverageNs"])/1e3))
PY
find $OUT -name "*.db" -delete
