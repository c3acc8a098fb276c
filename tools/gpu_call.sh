#!/bin/bash
# scratch: the commands of the current gpurun call
set -x
cd /root/repo
mkdir -p gpurun_out/r02r
timeout 600 python bench.py > gpurun_out/r02r/bench2.txt 2>&1
tail -1 gpurun_out/r02r/bench2.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'], d['kernels'])"
timeout 600 python tools/bench_skew.py 2>/dev/null | grep "^{" > gpurun_out/r02r/bench_skew2.txt; cut -c1-200 gpurun_out/r02r/bench_skew2.txt
timeout 600 python tools/bench_configs.py --out gpurun_out/r02r/bench_configs2.json 2>&1 | grep "^{" | cut -c1-120
