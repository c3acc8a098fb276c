#!/bin/bash
cd /root/repo
RSX_VERIFY=2 timeout 900 python tools/soak_r4.py 600 2>&1 | grep -v amdgpu.ids | tail -5 | tee gpurun_out/soak_r4_final.txt
