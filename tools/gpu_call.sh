set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02h
(timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -12) > gpurun_out/r02h/pytest_all.txt
tail -6 gpurun_out/r02h/pytest_all.txt
(RSX_PROBE_ALL=1 timeout 900 tools/ubench/scatter_probe.bin 28 > gpurun_out/r02h/scatter_probe_all.txt 2>&1)
grep -E "^v[0-9]|without" gpurun_out/r02h/scatter_probe_all.txt | cut -c1-150
(RSX_ELEM_LOADS=1 timeout 600 python bench.py --no-cpu-baseline) > gpurun_out/r02h/bench_elem_loads.txt 2>&1
tail -1 gpurun_out/r02h/bench_elem_loads.txt | cut -c1-300
(timeout 600 python bench.py --no-cpu-baseline) > gpurun_out/r02h/bench_default.txt 2>&1
tail -1 gpurun_out/r02h/bench_default.txt | cut -c1-300
