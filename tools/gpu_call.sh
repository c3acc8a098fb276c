#!/bin/bash
# scratch: the commands of the current gpurun call
set -x
cd /root/repo
mkdir -p gpurun_out/r02k
RSX_PROBE_ONE_ATOMIC=1 timeout 120 tools/ubench/scatter_probe.bin 28 > gpurun_out/r02k/scatter_probe_v10.txt 2>&1
grep -E "^v2 default|^v9|^v10|per tile: load \+ rank|fused|DIFFERS|without" gpurun_out/r02k/scatter_probe_v10.txt
