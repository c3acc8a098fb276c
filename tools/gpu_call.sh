set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02d
(timeout 600 tools/ubench/scatter_probe.bin 28 > gpurun_out/r02d/scatter_probe.txt 2>&1)
head -12 gpurun_out/r02d/scatter_probe.txt | cut -c1-400
