#!/bin/bash
# scratch: the commands of the current gpurun call (here: what the driver runs at the end of a round)
cd /root/repo
mkdir -p gpurun_out/full1
timeout 2400 python -m pytest tests -x -q -m gpu --durations=12 > gpurun_out/full1/pytest_gpu.txt 2>&1
tail -22 gpurun_out/full1/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/full1/smoke.txt 2>&1
tail -1 gpurun_out/full1/smoke.txt
