#!/bin/bash
# scratch: the commands of the current gpurun call
set -x
cd /root/repo
for i in 1 2; do
timeout 600 python tools/bench_configs.py --only cfg4 --steps 6 --out /tmp/a.json 2>/dev/null | grep "^{" | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('HOT   ', d['config'][:44].ljust(44), round(d['ms_per_sort'],3))"
RSX_NO_HOT=1 timeout 600 python tools/bench_configs.py --only cfg4 --steps 6 --out /tmp/b.json 2>/dev/null | grep "^{" | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('no HOT', d['config'][:44].ljust(44), round(d['ms_per_sort'],3))"
done
