#!/bin/bash
cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_routes.py -x -q -k "u64 or ranks" 2>&1 | tail -4
timeout 300 python tools/rank_mid_probe.py 2>&1 | grep -v amdgpu.ids | tail -6
