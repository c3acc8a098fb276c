#!/bin/bash
# scratch: the commands of the current gpurun call
cd /root/repo
mkdir -p gpurun_out/r4b
timeout 1500 python -m pytest tests/test_gpu_routes.py tests/test_gpu_fullsize.py "tests/test_gpu_parity.py::test_full_size_2p28_u32_properties" -x -q --durations=8 2>&1 | tail -25 > gpurun_out/r4b/routes.txt
cat gpurun_out/r4b/routes.txt
timeout 300 python bench.py > gpurun_out/r4b/bench.txt 2>&1
tail -1 gpurun_out/r4b/bench.txt | cut -c1-3000
