#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_routes.py -x -q -k "1e7" 2>&1 | tail -3
./tools/radix_bench --device 0 --verify 2>&1 | grep "radix_sort_device/[14]0000000 " 
RSX_NO_LEAF16Q=1 ./tools/radix_bench --device 0 --verify 2>&1 | grep "radix_sort_device/[14]0000000 "
timeout 600 python tools/mid_route_probe.py 2>&1 | grep -v amdgpu.ids | tail -12
