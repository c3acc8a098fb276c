#!/bin/bash
cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_hybrid.py tests/test_gpu_parity.py tests/test_gpu_async_routes.py tests/test_gpu_cpp.py -q -k "not rank and not pairs and not report_script" 2>&1 | tail -12
