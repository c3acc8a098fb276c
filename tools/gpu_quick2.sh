#!/bin/bash
cd /root/repo
timeout 2400 python -m pytest tests/test_gpu_routes.py tests/test_gpu_hybrid.py tests/test_gpu_async_routes.py -x -q -k "u64 or rank or pairs" 2>&1 | tail -5
