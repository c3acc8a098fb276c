#!/bin/bash
cd /root/repo
timeout 400 python bench.py --force-exchange --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('workload','')[:120])"
