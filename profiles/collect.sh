#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- bash profiles/collect.sh <tag>): rocprofv3 kernel-trace statistics and
# PMC passes (each counter group in a run of its own, never combined with a trace domain) for the
# three dominant kernels.  Raw output goes to gpurun_out/prof_<tag>_*; profiles/summarize.py
# turns it into the small tracked summaries under profiles/.
set -e
TAG=${1:-r02}
export TMPDIR=/tmp
OUT=gpurun_out
# only the bench line itself: the side measurements launch the same kernels on other shapes (the exact-scan recall legs,
# the IVF coarse quantizer) and would dilute the per-kernel averages.  The process prints its own ms_per_step and
# roofline; summarize.py stores them next to the profiler's kernel average.
BENCH="python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extra"
BENCH4K="python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra --storage f16 --batch 4096"
MFMA="python3 tools/prof_mfma.py 1000000 4096 4096 10"
IVF="python3 tools/prof_ivf.py 2000000 4096 1024 16"

echo "== kernel-trace stats: bench"
timeout -s KILL 300 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/prof_${TAG}_stats -o bench -- $BENCH > $OUT/prof_${TAG}_stats.log 2>&1
echo "== kernel-trace stats: bench, batch 4096 on f16 rows"
timeout -s KILL 300 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/prof_${TAG}_b4096_stats -o bench -- $BENCH4K > $OUT/prof_${TAG}_b4096_stats.log 2>&1
echo "== kernel-trace stats: ivf"
timeout -s KILL 300 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/prof_${TAG}_ivf_stats -o ivf -- $IVF > $OUT/prof_${TAG}_ivf_stats.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $C: bench"
  timeout -s KILL 300 rocprofv3 --output-format csv --pmc $C -d $OUT/prof_${TAG}_pmc_$C -o bench -- $BENCH > $OUT/prof_${TAG}_pmc_$C.log 2>&1
  echo "== pmc $C: ivf"
  timeout -s KILL 300 rocprofv3 --output-format csv --pmc $C -d $OUT/prof_${TAG}_ivf_pmc_$C -o ivf -- $IVF > $OUT/prof_${TAG}_ivf_pmc_$C.log 2>&1
done
echo "== pmc MFMA busy / clock: mfma"
timeout -s KILL 300 rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/prof_${TAG}_mfma_pmc_sq -o mfma -- $MFMA > $OUT/prof_${TAG}_mfma_pmc_sq.log 2>&1
echo "== pmc LDS: mfma"
timeout -s KILL 300 rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 -d $OUT/prof_${TAG}_mfma_pmc_lds -o mfma -- $MFMA > $OUT/prof_${TAG}_mfma_pmc_lds.log 2>&1
echo "== pmc FETCH_SIZE: mfma"
timeout -s KILL 300 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/prof_${TAG}_mfma_pmc_FETCH_SIZE -o mfma -- $MFMA > $OUT/prof_${TAG}_mfma_pmc_FETCH_SIZE.log 2>&1
echo "== done"
