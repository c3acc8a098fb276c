#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r4j
timeout 1500 python -m pytest tests/test_gpu_routes.py tests/test_gpu_hybrid.py -x -q -k "u64 or blind" 2>&1 | tail -5 > gpurun_out/r4j/routes.txt
cat gpurun_out/r4j/routes.txt
python tools/bench_configs.py --only "cfg3 u64 " --out gpurun_out/r4j/bench_configs.json > /dev/null 2>&1
python3 -c "
import json
for r in json.load(open('gpurun_out/r4j/bench_configs.json')):
    print(r['config'], 'ms', round(r['ms_per_sort'],3), 'leaf', round(r['leaf_ms'],3), 'scatter/launch', round(r['scatter_ms_per_launch'],3))
"
