#!/bin/bash
cd /root/repo
timeout 900 python tools/size_sweep.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/size_sweep_final.txt | tail -14
./tools/radix_bench --device 0 --verify 2>&1 | grep -v verified | grep "radix_sort_device" | tee gpurun_out/radix_bench_final.txt | tail -9
