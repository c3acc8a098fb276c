#!/bin/bash
# scratch: the commands of the current gpurun call
set -x
cd /root/repo
cp radix_sorting_amd/librsx.so /tmp/librsx_new.so
for round in 1 2 3; do
for v in new alt; do
  if [ $v = new ]; then cp /tmp/librsx_new.so radix_sorting_amd/librsx.so; else cp radix_sorting_amd/librsx_alt.so radix_sorting_amd/librsx.so; fi
  echo "== $v (round $round)"
  timeout 600 python tools/bench_configs.py --only cfg4 --steps 8 --out /tmp/bc.json 2>/dev/null | grep "^{" | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('  ', d['config'][:44].ljust(44), round(d['ms_per_sort'],3))"
done
done
cp /tmp/librsx_new.so radix_sorting_amd/librsx.so
