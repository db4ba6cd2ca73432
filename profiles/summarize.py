#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of profiles/collect.sh (gpurun_out/prof_<tag>_*) into the small
summaries tracked under profiles/:  <tag>_<what>_kernel_stats.csv (top kernels of the
--kernel-trace --stats run) and <tag>_<what>_pmc.json (per-kernel mean counter values per launch).

HBM traffic follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are collected in
separate passes, are in KiB, and on gfx950 FETCH_SIZE reports half of a wide coalesced read
stream, so bytes = FETCH_SIZE*1024*2 + WRITE_SIZE*1024.
usage: python profiles/summarize.py <tag>
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
GO = os.path.join(ROOT, "gpurun_out")


def short(name):
    return name.split("(")[0].strip()


def bench_line(logname):
    """The JSON line the profiled bench.py process printed (its own ms_per_step / roofline under the profiler)."""
    try:
        for line in open(os.path.join(GO, logname)):
            if line.startswith("{") and '"metric"' in line:
                return json.loads(line)
    except Exception:
        pass
    return None


def stats(what, dirname, logname=None, kernel=None, algo=None):
    files = glob.glob(os.path.join(GO, dirname, "**", "*kernel_stats.csv"), recursive=True)
    if not files:
        return
    rows = list(csv.reader(open(files[0])))
    out = os.path.join(ROOT, "profiles", f"{tag}_{what}_kernel_stats.csv")
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        for r in rows[:13]:
            w.writerow([c[:160] for c in r])
    print("wrote", out)
    line = bench_line(logname) if logname else None
    if line and kernel:
        # the profiled process's own numbers next to the profiler's average for the same kernel
        hdr = rows[0]
        avg_ns = None
        for r in rows[1:]:
            if kernel in r[hdr.index("Name")]:
                avg_ns = float(r[hdr.index("AverageNs")])
                break
        roof = line["roofline"]
        rec = {"command": "rocprofv3 --kernel-trace --stats -- python3 bench.py ... (profiles/collect.sh)",
               "workload": line["config"]["workload"], "kernel": kernel,
               "profiled_process_ms_per_step": line["ms_per_step"],
               "profiled_process_hipevent_kernel_avg_ms": roof["kernel_avg_ms"],
               "profiled_process_roofline_frac": roof["frac"],
               "rocprof_kernel_avg_ms": avg_ns / 1e6 if avg_ns else None}
        if avg_ns:
            work = roof.get("algorithmic_flops_per_launch") if roof["bound"] == "mfma" else roof["algorithmic_bytes_per_launch"]
            scale = 1e12 if roof["bound"] == "mfma" else 1e9
            rec["rocprof_roofline_frac"] = work / (avg_ns * 1e-9) / scale / roof["peak"]
            rec["rocprof_vs_hipevent"] = avg_ns / 1e6 / roof["kernel_avg_ms"]
        o2 = os.path.join(ROOT, "profiles", f"{tag}_{what}_vs_bench.json")
        json.dump(rec, open(o2, "w"), indent=1)
        print("wrote", o2)


def counters(dirname):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(GO, dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def pmc(what, dirs, command, workload, keep):
    kernels = {}
    for d in dirs:
        for k, cs in counters(d).items():
            if not any(s in k for s in keep):
                continue
            e = kernels.setdefault(k, {})
            for c, v in cs.items():
                e[c + "_avg_per_launch"] = sum(v) / len(v)
                e["launches_" + c] = len(v)
    for k, e in kernels.items():
        if "FETCH_SIZE_avg_per_launch" in e:
            e["hbm_bytes_per_launch_corrected"] = (e["FETCH_SIZE_avg_per_launch"] * 1024 * 2 +
                                                   e.get("WRITE_SIZE_avg_per_launch", 0.0) * 1024)
        if "SQ_VALU_MFMA_BUSY_CYCLES_avg_per_launch" in e and "GRBM_GUI_ACTIVE_avg_per_launch" in e:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy cycles over the 1024 SIMDs
            cyc = e["GRBM_GUI_ACTIVE_avg_per_launch"] / 8.0
            e["kernel_cycles"] = cyc
            e["mfma_util"] = e["SQ_VALU_MFMA_BUSY_CYCLES_avg_per_launch"] / (1024.0 * cyc)
    if not kernels:
        return
    out = os.path.join(ROOT, "profiles", f"{tag}_{what}_pmc.json")
    json.dump({"command": command, "workload": workload,
               "note": "FETCH_SIZE/WRITE_SIZE in KiB; gfx950 FETCH_SIZE counts half of a wide coalesced read "
                       "(MI355X_MICROARCH.md, HBM): corrected bytes = FETCH_SIZE*1024*2 + WRITE_SIZE*1024; "
                       "each counter group collected in its own rocprofv3 --pmc run",
               "kernels": kernels}, open(out, "w"), indent=1)
    print("wrote", out)


stats("bench", f"prof_{tag}_stats", f"prof_{tag}_stats.log", "flat_scan_topk_kernel")
stats("bench_b4096", f"prof_{tag}_b4096_stats", f"prof_{tag}_b4096_stats.log", "flat_mfma_pp_kernel")
stats("ivf", f"prof_{tag}_ivf_stats")
pmc("flat_f32_1M_b1", [f"prof_{tag}_pmc_FETCH_SIZE", f"prof_{tag}_pmc_WRITE_SIZE"],
    "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate runs) -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extra",
    "Flat IP 1000000x4096 f32: batch 1 k=10 (flat_scan_topk_kernel)",
    ["flat_scan_topk_kernel", "flat_mfma_pp_kernel", "merge_keys_kernel"])
pmc("mfma_f16_1M_b4096", [f"prof_{tag}_mfma_pmc_sq", f"prof_{tag}_mfma_pmc_lds", f"prof_{tag}_mfma_pmc_FETCH_SIZE"],
    "rocprofv3 --pmc <group> (separate runs) -- python3 tools/prof_mfma.py 1000000 4096 4096 10",
    "Flat IP 1000000x4096 f16, batch 4096, k=10", ["flat_mfma_pp_kernel"])
pmc("ivf_f16_2M_b1", [f"prof_{tag}_ivf_pmc_FETCH_SIZE", f"prof_{tag}_ivf_pmc_WRITE_SIZE"],
    "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate runs) -- python3 tools/prof_ivf.py 2000000 4096 1024 16",
    "IVF1024 nprobe=16 over 2000000x4096 f16, batch 1, k=10", ["ivf_scan_qreg_kernel", "ivf_scan_topk_kernel", "flat_scan_topk_kernel", "ivf_plan_kernel"])
# K1m (round 3): profiles/collect_scan_mm.sh
stats("scan_mm_b8", f"prof_{tag}_mm_stats", f"prof_{tag}_mm_stats.log", "flat_scan_mm_kernel")
pmc("scan_mm_f32_1M_b8", [f"prof_{tag}_mm_pmc_FETCH_SIZE", f"prof_{tag}_mm_pmc_WRITE_SIZE", f"prof_{tag}_mm_pmc_lds", f"prof_{tag}_mm_pmc_valu"],
    "rocprofv3 --pmc <group> (separate runs) -- python3 bench.py --batch 8 --steps 30 --warmup 3 --no-cpu-baseline --no-extra",
    "Flat IP 1000000x4096 f32: 8 queries per search, k=10 (flat_scan_mm_kernel)", ["flat_scan_mm_kernel"])
