#!/bin/bash
cd /root/repo
./tools/ubench/leaf16_probe.bin 28 0 | grep -v "skip mask"
timeout 1500 python -m pytest tests/test_gpu_hybrid.py -x -q -k blind 2>&1 | tail -2
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:round(v['ms_per_step'],4) for k,v in d['roofline']['per_kernel'].items()})"
done
