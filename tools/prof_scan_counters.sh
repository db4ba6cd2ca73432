#!/bin/bash
# Runs ON THE GPU BOX: what bounds K1 at the batch sizes below its best (5-8 queries on fp32 rows, 3-4 on fp16 rows)?
# LDS, VALU and memory counters per launch, one rocprofv3 pass per counter group and shape.
export TMPDIR=/tmp
IFS=";" read -ra LIST <<< "${SHAPES:-f32 1;f32 4;f32 8;f16 2;f16 4}"   # SHAPES="f32 8;f16 4" picks; SSMI_SCAN_MM=0 in the environment keeps K1m out
for shape in "${LIST[@]}"; do
  set -- $shape
  for grp in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM"; do
    d=gpurun_out/scanpmc_$1_$2_$(echo $grp | cut -c1-12 | tr ' ' _)
    timeout -s KILL 120 rocprofv3 --output-format csv --pmc $grp -d $d -o c -- python3 tools/prof_scan.py 1000000 4096 $1 $2 10 > $d.log 2>&1 || { echo "$shape: rocprofv3 failed"; tail -3 $d.log; exit 1; }
    python3 - "$d" "$1 x$2" <<'P'
import sys, csv, glob, collections
d, name = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "flat_scan_" in r["Kernel_Name"]:   # K1 (flat_scan_topk_kernel) or K1m (flat_scan_mm_kernel)
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(name, {k: round(sum(v) / len(v)) for k, v in acc.items()})
P
  done
done
