#!/bin/bash
cd /root/repo
timeout 2400 python -m pytest tests/test_gpu_async_routes.py tests/test_gpu_routes.py -x -q -k "ranks or rank_routes or level1 or pairs" 2>&1 | tail -8
