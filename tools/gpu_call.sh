#!/bin/bash
# scratch: the commands of the current gpurun call
set -x
cd /root/repo
mkdir -p gpurun_out/r02u
timeout 600 python tools/bench_configs.py --out gpurun_out/r02u/bench_configs.json 2>&1 | grep "^{" | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('  ', d['config'][:44].ljust(44), round(d['ms_per_sort'],3), round(d['Gkeys_per_s'],1))"
timeout 600 python bench.py > gpurun_out/r02u/bench.txt 2>&1
tail -1 gpurun_out/r02u/bench.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'], d['kernels']['histogram_ms_per_step'])"
timeout 1800 bash tools/profile_bench.sh r02 all > gpurun_out/r02u/profile.log 2>&1
tail -12 gpurun_out/r02u/profile.log
