#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r4g
timeout 1500 python -m pytest tests/test_gpu_async_routes.py -x -q 2>&1 | tail -15 > gpurun_out/r4g/async.txt
cat gpurun_out/r4g/async.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py tests/test_gpu_soak.py -x -q -k "async or graph or multi or soak or verif" 2>&1 | tail -5 > gpurun_out/r4g/others.txt
cat gpurun_out/r4g/others.txt
