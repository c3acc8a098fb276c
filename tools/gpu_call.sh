#!/bin/bash
# scratch: the commands of the current gpurun call (here: what the driver runs at the end of a round)
set -x
cd /root/repo
mkdir -p gpurun_out/final
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/final/pytest_gpu.txt 2>&1
tail -3 gpurun_out/final/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.txt 2>&1
tail -1 gpurun_out/final/smoke.txt
timeout 600 python bench.py > gpurun_out/final/bench.txt 2>&1
tail -1 gpurun_out/final/bench.txt | cut -c1-400
