#!/bin/bash
# Runs ON THE GPU BOX: counters of the batch-4096 Flat kernel (1M x 4096 f16 rows) for the product build or a timing-only
# build under environment switches, one rocprofv3 pass per spec - to tell stall cycles from clock and to see the
# fabric-side bytes.     bash tools/prof_clock.sh <tag> <counters> spec...      spec = name:lib|default[:VAR=val[,VAR=val]]
#   counters e.g. "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" or "FETCH_SIZE"
export TMPDIR=/tmp
TAG=$1; CNT=$2; shift 2
K=${PROF_K:-10}
for spec in "$@"; do
  IFS=: read -r name lib envs <<< "$spec"
  unset SSMI_LIB SSMI_MFMA_STAGED SSMI_MFMA_PATCH SSMI_MFMA_SYNC
  if [ "$lib" != "default" ]; then export SSMI_LIB=$PWD/$lib; fi
  for kv in ${envs//,/ }; do export "$kv"; done
  d=gpurun_out/clk_${TAG}_$name
  timeout -s KILL 120 rocprofv3 --output-format csv --pmc $CNT -d $d -o c -- python3 tools/prof_mfma.py 1000000 4096 4096 $K > $d.log 2>&1 || { echo "$name: rocprofv3 failed"; tail -3 $d.log; exit 1; }
  python3 - "$d" "$name" <<'P'
import sys, csv, glob, collections
d, name = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "flat_mfma_p" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(name, {k: round(sum(v) / len(v)) for k, v in acc.items()}, "launches", len(next(iter(acc.values()))))
P
done
