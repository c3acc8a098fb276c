#!/bin/bash
# the round-end sequence: GPU tests, smoke, bench (what the driver runs)
cd /root/repo
mkdir -p gpurun_out/final_r4g
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/final_r4g/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee gpurun_out/final_r4g/smoke.txt
timeout 600 python bench.py 2>&1 | tail -1 | tee gpurun_out/final_r4g/bench.txt
