#!/bin/bash
cd /root/repo
timeout 2400 python -m pytest tests/test_gpu_routes.py tests/test_gpu_hybrid.py tests/test_gpu_async_routes.py tests/test_gpu_fullsize.py -x -q -k "not rank and not pairs" 2>&1 | tail -4
timeout 600 python tools/footprint_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/footprint_probe.txt
