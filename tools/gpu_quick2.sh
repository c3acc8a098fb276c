#!/bin/bash
cd /root/repo
timeout 400 python tools/mid_route_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/mid_route_probe_final.txt | tail -12
timeout 200 python tools/u64_small_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/u64_small_probe_final.txt | tail -12
