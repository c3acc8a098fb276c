#!/bin/bash
# the round-end sequence (GPU tests, smoke, bench) + the profiles of round 4 at this commit
cd /root/repo
mkdir -p gpurun_out/final_r4d
timeout 2400 python -m pytest tests -x -q -m gpu --durations=15 2>&1 | tail -30 | tee gpurun_out/final_r4d/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee gpurun_out/final_r4d/smoke.txt
timeout 600 python bench.py 2>&1 | tail -1 | tee gpurun_out/final_r4d/bench.txt
timeout 900 python tools/bench_configs.py --out gpurun_out/final_r4d/bench_configs.json 2>&1 | tail -12 | tee gpurun_out/final_r4d/bench_configs.txt
timeout 2400 bash tools/profile_bench.sh r04d all > gpurun_out/final_r4d/profile.log 2>&1
tail -5 gpurun_out/final_r4d/profile.log
