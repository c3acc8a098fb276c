#!/bin/bash
cd /root/repo
for i in 1 2; do
timeout 300 python bench.py 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('aux slots ', round(d['value'],1), d['ms_per_step'], {k:round(v['ms_per_step'],4) for k,v in d['roofline']['per_kernel'].items()})"
RSX_NO_AUX_SLOTS=1 timeout 300 python bench.py 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('no aux    ', round(d['value'],1), d['ms_per_step'], {k:round(v['ms_per_step'],4) for k,v in d['roofline']['per_kernel'].items()})"
done
timeout 600 python tools/footprint_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/footprint_probe.txt
