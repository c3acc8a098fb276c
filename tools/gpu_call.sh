#!/bin/bash
cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_routes.py -x -q -k "stable_through" 2>&1 | tail -6
