#!/bin/bash
cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_soak.py -x -q 2>&1 | tail -12
