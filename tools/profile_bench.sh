#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + stats and separate PMC passes of bench.py.
# Usage: tools/profile_bench.sh <tag>      -> gpurun_out/prof_<tag>/...
TAG=${1:-r01}
cd "$(dirname "$0")/.."
REPO=$PWD
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python3 $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline"
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/pmc_sq1 -o pmc -- $CMD > $OUT/pmc_sq1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM -d $OUT/pmc_sq2 -o pmc -- $CMD > $OUT/pmc_sq2.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $OUT/pmc_fetch -o pmc -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_write -o pmc -- $CMD > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -50
python3 $REPO/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep the merge small
find $OUT -name "*.db" -delete
