#!/bin/bash
cd /root/repo
timeout 600 python tools/rank_threshold_probe.py 2>&1 | grep -v amdgpu.ids | grep default | tee gpurun_out/rank_probe3.txt
timeout 2400 python -m pytest tests/test_gpu_routes.py tests/test_gpu_async_routes.py tests/test_gpu_hybrid.py tests/test_gpu_fullsize.py tests/test_gpu_soak.py -x -q -k "rank or pairs or soak" 2>&1 | tail -5
