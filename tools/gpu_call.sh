#!/bin/bash
# scratch: the commands of the current gpurun call
set -x
cd /root/repo
mkdir -p gpurun_out/r02k
RSX_PROBE_LB=1 timeout 120 tools/ubench/scatter_probe.bin 28 > gpurun_out/r02k/scatter_probe_lb.txt 2>&1
grep -E "^v2|per super" gpurun_out/r02k/scatter_probe_lb.txt
