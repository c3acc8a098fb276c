#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + stats and SEPARATE PMC passes (never combined with a trace) of
#   bench.py                 (cfg 2: 2^28 u32 keys)            -> gpurun_out/prof_<tag>/bench/...
#   tools/bench_configs.py   (cfg 3: u64 uniform / P=5 / P=4 / Zipf; cfg 4: f32 -> ranks, three inputs; pairs)
#                                                              -> gpurun_out/prof_<tag>/configs/...
# Usage: tools/profile_bench.sh <tag> [bench|configs|all]
TAG=${1:-r02}
WHAT=${2:-all}
cd "$(dirname "$0")/.."
REPO=$PWD
export TMPDIR=/tmp
cd /tmp
profile() {   # <subdir> <program and arguments...>
	local OUT=$REPO/gpurun_out/prof_$TAG/$1
	shift
	mkdir -p $OUT
	rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o trace -- "$@" > $OUT/trace.log 2>&1
	rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/pmc_sq1 -o pmc -- "$@" > $OUT/pmc_sq1.log 2>&1
	rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM -d $OUT/pmc_sq2 -o pmc -- "$@" > $OUT/pmc_sq2.log 2>&1
	rocprofv3 --output-format csv --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $OUT/pmc_fetch -o pmc -- "$@" > $OUT/pmc_fetch.log 2>&1
	rocprofv3 --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_write -o pmc -- "$@" > $OUT/pmc_write.log 2>&1
	python3 $REPO/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
	find $OUT -name "*.db" -delete          # keep the merge small
	find $OUT -name "*_agent_info.csv" -delete
}
if [ "$WHAT" = bench ] || [ "$WHAT" = all ]; then
	profile bench python3 $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline   # (the driver's own K and W: a kernel's first dispatches after an idle second run 5-10 % faster than its steady state)
fi
if [ "$WHAT" = configs ] || [ "$WHAT" = all ]; then
	profile configs python3 $REPO/tools/bench_configs.py --steps 2 --warmup 1 --out $REPO/gpurun_out/prof_$TAG/configs/bench_configs.json
fi
tail -n 60 $REPO/gpurun_out/prof_$TAG/*/summary.txt
