set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
(timeout 600 tools/ubench/scatter_probe.bin 28 > gpurun_out/r02b/scatter_probe.txt 2>&1)
(timeout 300 tools/ubench/scatter_probe.bin 28 2 > gpurun_out/r02b/scatter_probe_mode2.txt 2>&1)
tail -40 gpurun_out/r02b/scatter_probe.txt
