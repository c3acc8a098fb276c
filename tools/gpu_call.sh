#!/bin/bash
# scratch: the commands of the current gpurun call
set -x
cd /root/repo
mkdir -p gpurun_out/r02k
timeout 120 tools/ubench/scatter_probe.bin 28 > gpurun_out/r02k/scatter_probe_v9.txt 2>&1
grep -E "^v2|^v8|^v9|column|records|identical|DIFFERS|per super|per tile|without" gpurun_out/r02k/scatter_probe_v9.txt
