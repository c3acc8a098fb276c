#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r4n
python tools/bench_configs.py --out gpurun_out/r4n/bench_configs.json > gpurun_out/r4n/bench_configs.txt 2>&1
python3 -c "
import json
for r in json.load(open('gpurun_out/r4n/bench_configs.json')):
    print(r['config'], '| ms', round(r['ms_per_sort'],3), '| Gkeys/s', round(r['Gkeys_per_s'],1), '| route', r['route'], '| B/key', r['algorithmic_bytes_per_key'], '| frac', round(r['frac_of_8TBps'],3), '| leaf', round(r['leaf_ms'],3), 'scat', round(r['scatter_ms_per_launch'],3), 'hist', round(r['hist_ms'],3))
"
python tools/size_sweep.py > gpurun_out/r4n/size_sweep.txt 2>&1
./tools/radix_bench --device 0 --verify > gpurun_out/r4n/radix_bench.txt 2>&1
grep "radix_sort_device" gpurun_out/r4n/radix_bench.txt | grep -v verified
python tools/mid_route_probe.py > gpurun_out/r4n/mid.txt 2>&1
timeout 300 python bench.py > gpurun_out/r4n/bench.txt 2>&1
tail -1 gpurun_out/r4n/bench.txt | cut -c1-200
