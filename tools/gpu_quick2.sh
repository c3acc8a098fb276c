#!/bin/bash
cd /root/repo
RSX_VERIFY=2 timeout 600 python tools/soak_r4.py 360 2>&1 | grep -v amdgpu.ids | tail -3 | tee gpurun_out/soak_r4_final2.txt
