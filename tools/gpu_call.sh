#!/bin/bash
# the round-end sequence (GPU tests, smoke, bench) + the profiles of round 4 at this commit
cd /root/repo
mkdir -p gpurun_out/final_r4c
timeout 2400 python -m pytest tests -x -q -m gpu --durations=15 2>&1 | tail -30 | tee gpurun_out/final_r4c/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee gpurun_out/final_r4c/smoke.txt
timeout 600 python bench.py 2>&1 | tail -1 | tee gpurun_out/final_r4c/bench.txt
timeout 900 python tools/bench_configs.py --out gpurun_out/final_r4c/bench_configs.json 2>&1 | tail -12 | tee gpurun_out/final_r4c/bench_configs.txt
timeout 2400 bash tools/profile_bench.sh r04c all > gpurun_out/final_r4c/profile.log 2>&1
tail -5 gpurun_out/final_r4c/profile.log
