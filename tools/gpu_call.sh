#!/bin/bash
# scratch: the commands of the current gpurun call
set -x
cd /root/repo
mkdir -p gpurun_out/r02q
timeout 1800 bash tools/profile_bench.sh r02 all > gpurun_out/r02q/profile.log 2>&1
tail -40 gpurun_out/r02q/profile.log
timeout 600 python bench.py > gpurun_out/r02q/bench.txt 2>&1
tail -1 gpurun_out/r02q/bench.txt | cut -c1-1800
