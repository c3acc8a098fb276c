#!/bin/bash
cd /root/repo
RSX_VERIFY=2 timeout 1300 python tools/soak_r4.py 1080 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-700 | tee gpurun_out/soak_r4_final3.txt
