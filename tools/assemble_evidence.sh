#!/bin/bash
# After the round's last GPU sequence (tools/gpu_call.sh F; tools/profile_bench.sh P all; tools/gpu_evidence.sh E; tools/gpu_profgap.sh G):
# copies what is judged from gpurun_out/ into profiles/<round>/ and stamps it with the last commit that changes code.
#   tools/assemble_evidence.sh r06 F P E G
R=$1; F=$2; P=$3; E=$4; G=$5
cd "$(dirname "$0")/.."
HEADC=$(git log -1 --format=%h -- radix_sorting_amd/csrc include bench.py radix_sorting_amd/__init__.py radix_sorting_amd/multi.py)
for w in bench configs; do
  cp gpurun_out/prof_$P/$w/trace/trace_kernel_stats.csv profiles/$R/$w/kernel_stats.csv
  python tools/summarize_prof.py gpurun_out/prof_$P/$w > profiles/$R/$w/rocprofv3_summary.txt 2>&1
  cp gpurun_out/prof_$P/$w/roofline_table.json profiles/$R/$w/roofline_table.json
done
cp gpurun_out/prof_$P/configs/bench_configs.json profiles/$R/bench_configs_profiled_run.json
cp gpurun_out/$F/pytest_gpu.txt gpurun_out/$F/smoke.txt gpurun_out/$F/bench.txt profiles/$R/final/
echo "library, bench.py and the driver at commit $HEADC (git log -1 -- radix_sorting_amd/csrc include bench.py radix_sorting_amd/*.py: the last commit that changes code); the commits behind it add profiles, documentation and tests only" > profiles/$R/final/head.txt
cp gpurun_out/$E/bench_line.json profiles/$R/bench_line.json
cp gpurun_out/$E/bench_configs.json profiles/$R/bench_configs.json
for f in size_sweep.txt radix_bench.txt footprint_probe.txt big_sizes.txt; do cp gpurun_out/$E/$f profiles/$R/$f; done
cp gpurun_out/${G}_log.txt profiles/$R/profiler_gap.txt
python tools/make_pmc_json.py profiles/$R/bench/roofline_table.json ${R#r0} $HEADC > /dev/null
grep measured_at profiles/pmc_kernels.json
cat profiles/$R/final/head.txt
