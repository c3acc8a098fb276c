set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02c
(timeout 300 tools/ubench/hist_probe.bin 28 4; timeout 300 tools/ubench/hist_probe.bin 28 8; timeout 200 tools/ubench/hist_probe.bin 28 8 ffffffffff; timeout 200 tools/ubench/hist_probe.bin 28 8 ffffffff) > gpurun_out/r02c/hist_probe.txt 2>&1
(timeout 900 python -m pytest tests/test_gpu_hist.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -15) > gpurun_out/r02c/pytest.txt
(timeout 600 tools/ubench/scatter_probe.bin 28 > gpurun_out/r02c/scatter_probe.txt 2>&1)
(timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline) > gpurun_out/r02c/bench.txt 2>&1
tail -5 gpurun_out/r02c/pytest.txt
