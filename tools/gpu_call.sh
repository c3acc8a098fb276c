#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r4p
RSX_VERIFY=2 timeout 600 python tools/soak_r4.py 150 > gpurun_out/r4p/soak_r4.txt 2>&1
tail -3 gpurun_out/r4p/soak_r4.txt
RSX_VERIFY=2 timeout 400 python tools/soak.py 90 > gpurun_out/r4p/soak_verify2.txt 2>&1
tail -2 gpurun_out/r4p/soak_verify2.txt
