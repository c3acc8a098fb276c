#!/bin/bash
# A/B of bench.py on ONE box: tools/gpu_ab.sh <tag> "ENV1=.. ENV2=.." "ENV=.." ...   ("-" = no switch)
# Prints value, ms per step and the per-kernel HIP-event times of every variant (boxes differ by 6 %: only runs of one call compare).
TAG=$1; shift
cd "$(dirname "$0")/.."
OUT=gpurun_out/$TAG
mkdir -p $OUT
i=0
for v in "$@"; do
	i=$((i+1))
	if [ "$v" = "-" ]; then envs=""; else envs="$v"; fi
	env $envs python bench.py --no-cpu-baseline --steps 20 --warmup 3 > $OUT/bench_$i.txt 2>&1
	python3 - "$v" $OUT/bench_$i.txt <<'PY'
import json,sys
v,f=sys.argv[1],sys.argv[2]
try:
    j=json.loads(open(f).read().strip().splitlines()[-1])
    pk=j.get("roofline",{}).get("per_kernel",{})
    print("%-40s %.1f Gkeys/s  %.4f ms  "%(v,j["value"],j["ms_per_step"])+"  ".join("%s %.4f"%(k,x.get("avg_launch_ms",0)) for k,x in pk.items()))
except Exception as e:
    print(v,"FAILED",e); print(open(f).read()[-2000:])
PY
done
