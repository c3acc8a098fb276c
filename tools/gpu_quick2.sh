#!/bin/bash
cd /root/repo
cp tools/ubench/librsx_base.so radix_sorting_amd/librsx.so
sed -i 's/for n in (11800000, 12582912, 13107200, 13369344, 25600000, 26738688, 40000000):/for n in (12582912, 13107200, 13369344, 26738688):/' tools/overflow_rate_probe.py
timeout 900 python tools/overflow_rate_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/overflow_rate_probe_before.txt
