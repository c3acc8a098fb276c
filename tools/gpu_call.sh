#!/bin/bash
cd /root/repo
timeout 900 python tools/u64_threshold_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/u64_threshold_probe.txt
timeout 1500 python -m pytest tests/test_gpu_routes.py tests/test_gpu_hybrid.py -x -q -k "u64 or rank or pair" 2>&1 | tail -5
