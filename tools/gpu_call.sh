#!/bin/bash
cd /root/repo
timeout 900 python tools/rank_threshold_probe.py 2>&1 | tee gpurun_out/rank_threshold_probe2.txt
timeout 1200 python -m pytest tests/test_gpu_routes.py tests/test_gpu_hybrid.py -x -q -k "rank or pair" 2>&1 | tail -5
