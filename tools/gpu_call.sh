#!/bin/bash
# scratch: the commands of the current gpurun call
set -x
cd /root/repo
mkdir -p gpurun_out/r02m
timeout 900 python tools/size_sweep.py > gpurun_out/r02m/size_sweep.txt 2>&1
cat gpurun_out/r02m/size_sweep.txt
