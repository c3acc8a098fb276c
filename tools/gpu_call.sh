#!/bin/bash
# scratch: the commands of the current gpurun call
set -x
cd /root/repo
mkdir -p gpurun_out/r02k
timeout 600 tools/ubench/scatter_probe.bin 28 > gpurun_out/r02k/scatter_probe_read_ahead.txt 2>&1
grep -E "v2 default|read-ahead|identical|DIFFERS" gpurun_out/r02k/scatter_probe_read_ahead.txt
