#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- bash profiles/collect_scan_mm.sh <tag>): kernel-trace statistics and PMC passes (one
# counter group per run) of K1m, the exact scan on the fp32 matrix instruction, at 8 queries per search over the bench's
# database.  Raw output: gpurun_out/prof_<tag>_mm_*; profiles/summarize.py <tag> writes the tracked summaries.
set -e
TAG=${1:-r03}
export TMPDIR=/tmp
OUT=gpurun_out
BENCH="python3 bench.py --batch 8 --steps 30 --warmup 3 --no-cpu-baseline --no-extra"
timeout -s KILL 300 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/prof_${TAG}_mm_stats -o bench -- $BENCH > $OUT/prof_${TAG}_mm_stats.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --output-format csv --pmc $C -d $OUT/prof_${TAG}_mm_pmc_$C -o bench -- $BENCH > $OUT/prof_${TAG}_mm_pmc_$C.log 2>&1
done
timeout -s KILL 300 rocprofv3 --output-format csv --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/prof_${TAG}_mm_pmc_lds -o bench -- $BENCH > $OUT/prof_${TAG}_mm_pmc_lds.log 2>&1
timeout -s KILL 300 rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d $OUT/prof_${TAG}_mm_pmc_valu -o bench -- $BENCH > $OUT/prof_${TAG}_mm_pmc_valu.log 2>&1
echo "== done"
