#!/bin/bash
# scratch: the commands of the current gpurun call
set -x
cd /root/repo
mkdir -p gpurun_out/r02p
RSX_NO_HOT=1 timeout 600 python tools/bench_skew.py 2>/dev/null | tail -2 | cut -c1-200
timeout 600 python tools/bench_configs.py --out gpurun_out/r02p/bench_configs.json > gpurun_out/r02p/bench_configs.txt 2>&1
python3 -c "
import json
for r in json.load(open('gpurun_out/r02p/bench_configs.json')):
    print({k:(round(v,3) if isinstance(v,float) else v) for k,v in r.items() if k in ('config','name','ms_per_sort','ms','Gkeys_per_s','kept_columns','scatter_ms_per_launch','hist_ms')})
"
RSX_NO_HOT=1 timeout 600 python tools/bench_configs.py --only ipf --out gpurun_out/r02p/bench_configs_nohot.json 2>&1 | tail -3 | cut -c1-300
