#!/bin/bash
cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_hybrid.py -x -q -k "rank or pairs" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_gpu_routes.py tests/test_gpu_fullsize.py -x -q -k "ranks or pairs or cfg4" 2>&1 | tail -3
python tools/bench_configs.py --only "cfg4 f32 random" --out /tmp/bc.json 2>&1 | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['config'], 'ms', round(r['ms_per_sort'],3), 'leaf', round(r['leaf_ms'],3), 'route', r['route'])"
RSX_NO_LEAF16=1 python tools/bench_configs.py --only "cfg4 f32 random" --out /tmp/bc.json 2>&1 | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('old:', r['config'], 'ms', round(r['ms_per_sort'],3), 'leaf', round(r['leaf_ms'],3), 'route', r['route'])"
