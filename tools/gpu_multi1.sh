#!/bin/bash
# One GPU, the N > 1 code path (bench.py --force-exchange): the step with 1, 2 and 4 sub-ranges, phases and host marks -> gpurun_out/<tag>/
TAG=${1:-multi1}
cd "$(dirname "$0")/.."
OUT=gpurun_out/$TAG
mkdir -p $OUT
for c in 1 2 4 1 2 4; do
	RSX_MULTI_CHUNKS=$c python bench.py --gpus 1 --force-exchange --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/chunks_$c.json
	python3 - $c $OUT/chunks_$c.json <<'PY'
import json,sys
c,f=sys.argv[1],sys.argv[2]
j=json.loads(open(f).read())
print("chunks=%s: %.2f ms/step  %.1f Gkeys/s  phases %s  host %s" % (c, j["ms_per_step"], j["value"], j.get("multi",{}).get("phases_last_step_max_over_ranks_ms"), j.get("host")))
PY
done
