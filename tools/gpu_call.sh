#!/bin/bash
cd /root/repo
timeout 1800 python -m pytest tests/test_gpu_routes.py -x -q -k "u64" 2>&1 | tail -8
timeout 900 python tools/u64_threshold_probe.py 2>&1 | grep -v amdgpu.ids | grep "0xFF" | tee gpurun_out/u64_narrow_probe.txt
