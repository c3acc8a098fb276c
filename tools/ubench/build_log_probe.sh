#!/bin/bash
cd "$(dirname "$0")/../.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I radix_sorting_amd/csrc tools/ubench/log_probe.hip -o tools/ubench/log_probe.bin 2>&1 | grep -E "error|spill" | head
