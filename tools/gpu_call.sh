#!/bin/bash
# scratch: the commands of the current gpurun call
set -x
cd /root/repo
mkdir -p gpurun_out/r02s
timeout 1800 bash tools/profile_bench.sh r02 all > gpurun_out/r02s/profile.log 2>&1
tail -14 gpurun_out/r02s/profile.log
