#!/usr/bin/env python3
"""bench.py - Stage-1 retrieval throughput on MI355X (contract: see the task statement).

Workload (BASELINE.json configs[1]): Flat inner-product search over 1M x 4096
fp32 rows, batch-1 queries, k=10.  A "step" is one search of one query batch over
the whole database.  With --gpus N (one rank per GPU under torch.distributed.run)
the SAME database is row-sharded over the ranks (strong scaling): every rank
scans N_total/N rows, the per-rank top-k lists are all-gathered over RCCL and
merged.  Data is synthetic: N(0,1) components, L2-normalised rows, generated on
the device; queries likewise.  Inputs are resident in HBM before timing starts.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16/bf16


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n", type=int, default=1_000_000, help="total database rows (sharded over --gpus)")
    ap.add_argument("--d", type=int, default=4096)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--storage", default="f32", choices=["f32", "f16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the batch-1024/4096 MFMA side measurements")
    ap.add_argument("--no-ivf-extra", action="store_true", help="skip the IVF4096 / 10M-row side measurement (configs[3])")
    ap.add_argument("--cpu-sample-rows", type=int, default=100_000)
    ap.add_argument("--workload", default="flat", choices=["flat", "ivf"],
                    help="flat = BASELINE configs[1] (the default bench line); ivf = configs[3] (IVF4096, nprobe 32; 1 GPU)")
    ap.add_argument("--nlist", type=int, default=4096)
    ap.add_argument("--nprobe", type=int, default=32)
    ap.add_argument("--ivf-sorted", action="store_true", help="IVF workload: clustered rows arrive cluster by cluster (list pages contiguous)")
    ap.add_argument("--ivf-data", default="clustered", choices=["clustered", "iid"],
                    help="clustered: unit rows around 16384 random centres (sigma 0.6); iid: unit gaussian rows")
    return ap.parse_args()


def run_ivf(args, emit=True):
    """BASELINE configs[3]: IVF<nlist> nprobe=<nprobe> over n x d f16 rows on one GPU: coarse-quantizer
    training, add, batch-1 and batch-1000 search, recall@10 against the Flat index over the same rows."""
    import semantic_search_amd as ssa
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    n, d, nlist, nprobe, k = args.n, args.d, args.nlist, args.nprobe, args.k
    g = torch.Generator(device=dev).manual_seed(1234)
    ncl, sigma = 16384, 0.6
    centres = torch.randn((ncl, d), generator=g, device=dev)
    centres /= centres.norm(dim=1, keepdim=True)

    def rows(m, first=None):
        x = torch.randn((m, d), generator=g, device=dev) * (sigma / d ** 0.5 if args.ivf_data == "clustered" else 1.0)
        if args.ivf_data == "clustered":
            if first is None or not args.ivf_sorted:
                x += centres[torch.randint(0, ncl, (m,), generator=g, device=dev)]
            else:  # rows arrive cluster by cluster: the pages of a list end up next to each other
                x += centres[(torch.arange(first, first + m, device=dev) * ncl) // n]
        return x / x.norm(dim=1, keepdim=True)

    ivf = ssa.IndexIVFFlat(ssa.IndexFlatIP(d, storage="f16"), d, nlist, ssa.METRIC_INNER_PRODUCT, storage="f16")
    flat = ssa.IndexFlatIP(d, storage="f16")
    ntrain = min(n, 256 * nlist)
    xt = rows(ntrain)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ivf.train(xt)
    torch.cuda.synchronize(); t_train = time.perf_counter() - t0
    del xt
    t_add = 0.0
    for s0 in range(0, n, 65536):
        x = rows(min(65536, n - s0), s0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ivf.add(x)
        t_add += time.perf_counter() - t0
        flat.add(x)
    del x
    ivf.nprobe = nprobe
    ls = ivf.list_sizes()
    Q = rows(1000)
    # the timed loop runs as a serving loop would: no hipEvent records between the kernels (each costs the
    # stream ~6 us); the kernel time for the roofline comes from the 32 timed sample launches below
    ivf.set_timing(False)
    for i in range(args.warmup):
        ivf.search(Q[i:i + 1], k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(args.steps):
        ivf.search(Q[i % 1000:i % 1000 + 1], k)
    torch.cuda.synchronize(); elapsed = time.perf_counter() - t0
    ivf.set_timing(True)
    # algorithmic bytes = rows of the probed lists x d x 2 B, measured per query: the bench
    # reads back the prefix-sum total of each of 32 sample queries together with that launch's hipEvent time
    rows_s, ms_s = [], []
    for i in range(32):
        ivf.search(Q[i:i + 1], k)
        t = ivf.last_timing()
        rows_s.append(t["last_scan_bytes"] / (d * 2)); ms_s.append(t["last_scan_ms"])
    probed_rows = float(np.mean(rows_s))
    algo_bytes = probed_rows * d * 2
    scan_ms_sample = float(np.mean(ms_s))
    ivf.search(Q, k)  # warm the batched path
    torch.cuda.synchronize(); t0 = time.perf_counter()
    Di, Ii = ivf.search(Q, k)
    torch.cuda.synchronize(); t_b = time.perf_counter() - t0
    Q4 = rows(4096)
    ivf.search(Q4, k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ivf.search(Q4, k)
    torch.cuda.synchronize(); t_b4 = time.perf_counter() - t0
    del Q4
    Df, If = flat.search(Q, k)
    torch.cuda.synchronize()
    Ii, If = Ii.cpu().numpy(), If.cpu().numpy()
    recall = float(np.mean([len(set(a) & set(b)) / k for a, b in zip(Ii, If)]))
    line = {
        "metric": "queries/sec at k=%d, %d-d (IVF%d nprobe=%d, batch 1)" % (k, d, nlist, nprobe),
        "value": args.steps / elapsed, "unit": "queries/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f16 storage, f32 accumulate", "data": "synthetic (%s)" % args.ivf_data,
        "config": {"workload": "IVF%d nprobe=%d over %dx%d f16 (BASELINE.json configs[3]), batch 1, k=%d" % (nlist, nprobe, n, d, k),
                   "rows_total": n, "dim": d, "nlist": nlist, "nprobe": nprobe, "k": k, "train_rows": ntrain},
        "roofline": {"bound": "hbm", "achieved": algo_bytes / (scan_ms_sample * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": algo_bytes / (scan_ms_sample * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_launch": algo_bytes,
                     "algorithmic_bytes_note": "mean rows of the probed lists over 32 sample queries (%d; nprobe*N/nlist = %d) x d x 2 B, "
                                               "over the mean hipEvent time of those launches" % (int(probed_rows), int(nprobe * n / nlist)),
                     "kernel": "ivf_scan_qreg_kernel", "kernel_avg_ms": scan_ms_sample},
        "extra": {"train_s": t_train, "add_s": t_add, "add_vectors_per_s": n / t_add,
                  "list_sizes": {"min": int(ls.min()), "median": float(np.median(ls)), "max": int(ls.max())},
                  "batch_1000_qps": 1000 / t_b, "batch_4096_qps": 4096 / t_b4, "recall_at_%d_vs_flat" % k: recall,
                  "recall_note": "1000 queries; ground truth = IndexFlatIP (f16 rows) over the same vectors"},
    }
    ivf.close(); flat.close()
    if emit:
        print(json.dumps(line), flush=True)
    return line


def cpu_baseline(index, d, k, sample_rows, queries):
    """C/OpenMP restatement of the oracle on a bounded sample of the same rows."""
    from oracle import c_oracle
    n = min(sample_rows, index.ntotal)
    X = index.reconstruct_n(0, n)
    Q = queries[:4].cpu().numpy()
    try:  # the reference's own engine, if this machine has it; the build image does not
        import faiss
    except Exception:
        faiss = None
    if faiss is not None:
        ref = faiss.IndexFlatIP(d)
        ref.add(X)
        ref.search(Q[:1], k)
        reps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 10.0 and reps < 64:
            for i in range(Q.shape[0]):
                ref.search(Q[i:i + 1], k)  # batch 1, as core/index_manager.py:371 calls it
            reps += 1
        dt = time.perf_counter() - t0
        return {"value": reps * Q.shape[0] / dt, "unit": "queries/s", "cores": faiss.omp_get_max_threads(), "kind": "reference",
                "sample": f"faiss {faiss.__version__} IndexFlatIP on the first {n} rows x {d} fp32 of the same database, "
                          f"{reps * Q.shape[0]} batch-1 queries, k={k}; linear extrapolation to the full database = value * {n}/N"}
    c_oracle.search(X[:1000], Q[:1], k)  # load + warm
    reps, t0 = 0, time.perf_counter()
    while True:
        c_oracle.search(X, Q, k)
        reps += 1
        if time.perf_counter() - t0 > 10.0 or reps >= 8:
            break
    dt = time.perf_counter() - t0
    qps = reps * Q.shape[0] / dt
    return {
        "value": qps, "unit": "queries/s", "cores": c_oracle.num_threads(), "kind": "port",
        "sample": f"oracle/flat_ip.c (OpenMP) on the first {n} rows x {d} fp32 of the same database, "
                  f"{reps * Q.shape[0]} batch-1 queries, k={k}; linear extrapolation to the full database "
                  f"= value * {n}/N",
    }


def main():
    args = parse()
    if args.workload == "ivf":
        return run_ivf(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (MI355X)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # SSMI_BENCH_FORCE_DIST=1: rehearse the multi-rank code path (process group, barriers, max over ranks, sharded
    # search with its RCCL all-gather) on a single rank
    dist_on = world > 1 or os.environ.get("SSMI_BENCH_FORCE_DIST") == "1"
    if dist_on:
        dist.init_process_group("nccl", device_id=dev)

    from semantic_search_amd.sharded import ShardedFlatIP, shard_bounds

    bounds = shard_bounds(args.n, world)
    n_local = bounds[rank + 1] - bounds[rank]
    index = ShardedFlatIP(args.d, storage=args.storage, device=local_rank)
    index.add_local_random_unit(n_local, seed=1234 + rank)
    index.finalize(device=dev)
    if dist_on and world == 1:
        index._force_pipeline = True  # rehearsal: take the all-gather + merge path although there is one rank

    g = torch.Generator(device="cpu").manual_seed(4321)
    nsets = 8  # rotate over a few query batches so no step repeats the previous one
    Q = torch.randn((nsets, args.batch, args.d), generator=g, dtype=torch.float32)
    Q = (Q / Q.norm(dim=-1, keepdim=True)).to(dev)

    def step(i):
        return index.search(Q[i % nsets], args.k)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(i)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # dominant kernel: per-launch hipEvent durations recorded by the library on the launch stream
    scan_ms = index.local.scan_times_ms(min(args.steps, 256))
    scan_avg_ms = float(np.mean(scan_ms)) if len(scan_ms) else float("nan")
    elsize = 4 if args.storage == "f32" else 2
    timing = index.local.last_timing()
    passes = max(1, timing["last_scan_bytes"] // max(1, n_local * args.d * elsize))
    algo_bytes = n_local * args.d * elsize  # N*d*s per batch, per GPU N/G
    algo_flops = 2.0 * args.batch * n_local * args.d
    intensity = algo_flops / algo_bytes
    if timing["last_path"] == 1:
        roof = {"bound": "mfma", "achieved": algo_flops / (scan_avg_ms * 1e-3) / 1e12, "peak": MFMA_F16_PEAK_TFLOPS,
                "unit": "TFLOP/s"}
    else:
        roof = {"bound": "hbm", "achieved": algo_bytes / (scan_avg_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                "unit": "GB/s"}
    roof["frac"] = roof["achieved"] / roof["peak"]
    roof["traffic"] = None
    try:  # HBM bytes per launch from the committed rocprofv3 PMC passes of this exact workload
        prof = json.load(open(os.path.join(ROOT, "profiles", "r01_flat_f32_1M_b1_pmc.json")))
        if (args.n, args.d, args.batch, args.storage, world) == (1_000_000, 4096, 1, "f32", 1):
            for name, v in prof["kernels"].items():
                if "flat_scan_topk_kernel" in name:
                    roof["traffic"] = v["hbm_bytes_per_launch_corrected"]
                    roof["traffic_source"] = ("profiles/r01_flat_f32_1M_b1_pmc.json (rocprofv3 --pmc FETCH_SIZE x2 gfx950 "
                                              "correction + WRITE_SIZE, separate passes; profiles/collect.sh)")
    except Exception:
        pass
    roof["algorithmic_bytes_per_launch"] = algo_bytes
    roof["kernel"] = {0: "flat_scan_topk_kernel", 1: "flat_mfma_topk_kernel", 2: "ivf_scan_topk_kernel"}[timing["last_path"]]
    roof["kernel_avg_ms"] = scan_avg_ms
    roof["launches_timed"] = int(len(scan_ms))
    roof["db_passes_per_launch"] = int(passes)
    roof["flop_per_byte"] = intensity

    # side measurements (not `value`): large query batches over the same sharded database run the
    # MFMA contraction (on the fp16 shadow copy when storage is f32); few steps, same timing rules
    extra = {}
    if not args.no_extra:
        for B, steps in ((1024, 5), (4096, 3)):
            Qb = torch.randn((2, B, args.d), generator=g, dtype=torch.float32)
            Qb = (Qb / Qb.norm(dim=-1, keepdim=True)).to(dev)
            index.search(Qb[0], args.k)
            torch.cuda.synchronize()
            if dist_on:
                dist.barrier()
            t0 = time.perf_counter()
            for i in range(steps):
                index.search(Qb[i % 2], args.k)
            torch.cuda.synchronize()
            if dist_on:
                dist.barrier()
            el = time.perf_counter() - t0
            if dist_on:
                t = torch.tensor([el], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t.item())
            kms = float(np.mean(index.local.scan_times_ms(steps)))
            tl = index.local.last_timing()
            flops = 2.0 * B * n_local * args.d
            extra["batch_%d" % B] = {
                "qps": B * steps / el, "ms_per_step": el / steps * 1e3, "kernel_avg_ms": kms,
                "kernel": "flat_mfma_pp_kernel" if tl["last_path"] == 1 else "flat_scan_topk_kernel",
                "achieved_tflops_per_gpu": flops / (kms * 1e-3) / 1e12,
                "mfma_frac_of_2.5PF": flops / (kms * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS,
                "operand_dtype": "f16 (shadow copy of the f32 rows), f32 accumulate" if args.storage == "f32" else "f16, f32 accumulate",
            }
        try:  # counter-based MFMA utilisation of the batch-4096 contraction (committed PMC pass)
            prof = json.load(open(os.path.join(ROOT, "profiles", "r01_mfma_f16_1M_b4096_pmc.json")))
            v = next(iter(prof["kernels"].values()))
            extra["mfma_util_pmc"] = {
                "value": v["mfma_util"], "kernel_cycles": v["kernel_cycles"],
                "definition": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE/8), rocprofv3 --pmc, "
                              "Flat 1M x 4096 f16, batch 4096 (profiles/r01_mfma_f16_1M_b4096_pmc.json)"}
        except Exception:
            pass

    if rank == 0:
        qps = args.batch * args.steps / elapsed
        line = {
            "metric": "queries/sec at k=%d, %d-d" % (args.k, args.d),
            "value": qps,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32" if args.storage == "f32" else "f16 storage, f32 accumulate",
            "data": "synthetic",
            "config": {
                "workload": "Flat IP %dx%d %s, batch %d, k=%d (BASELINE.json configs[1]); database row-sharded over n_gpus"
                            % (args.n, args.d, args.storage, args.batch, args.k),
                "rows_total": args.n, "rows_per_gpu": n_local, "dim": args.d, "k": args.k, "batch": args.batch,
                "storage": args.storage, "parallelism": "rowshard%d" % world,
            },
            "roofline": roof,
            "extra": extra,
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(index.local, args.d, args.k, args.cpu_sample_rows, Q[0].expand(4, -1) if args.batch == 1 else Q[0])
            line["cpu_baseline"]["extrapolated_full_db_qps"] = line["cpu_baseline"]["value"] * min(args.cpu_sample_rows, n_local) / args.n
        if world == 1 and not args.no_extra and not args.no_ivf_extra:
            # side measurement (not `value`): BASELINE configs[3], IVF4096 nprobe 32 over 10M x 4096 f16 rows, the same
            # procedure as `bench.py --workload ivf`; needs ~170 GB of HBM next to the flat index
            try:
                index.local.close()
                free_b, _ = torch.cuda.mem_get_info()
                if free_b > 200 * (1 << 30):
                    import copy
                    a2 = copy.copy(args)
                    a2.n, a2.steps, a2.warmup, a2.ivf_data, a2.ivf_sorted = 10_000_000, 200, 10, "clustered", False
                    iv = run_ivf(a2, emit=False)
                    extra["ivf_c4"] = {"workload": iv["config"]["workload"], "batch_1_qps": iv["value"],
                                       "scan_kernel_hbm_frac": iv["roofline"]["frac"], "scan_kernel_ms": iv["roofline"]["kernel_avg_ms"],
                                       "probed_bytes_per_query": iv["roofline"]["algorithmic_bytes_per_launch"], **iv["extra"]}
                else:
                    extra["ivf_c4"] = {"skipped": "less than 200 GiB of HBM free"}
            except Exception as e:  # the side measurement must never take the bench line down
                extra["ivf_c4"] = {"error": repr(e)}
            # side measurement: BASELINE configs[2], Flat 10M x 4096 f16, batch 4096, k=100
            try:
                free_b, _ = torch.cuda.mem_get_info()
                if free_b > 120 * (1 << 30):
                    import semantic_search_amd as ssa
                    c3 = ssa.IndexFlatIP(args.d, storage="f16")
                    c3.add_random_unit(10_000_000, 1234)
                    Q3 = torch.randn((2, 4096, args.d), generator=g, dtype=torch.float32)
                    Q3 = (Q3 / Q3.norm(dim=-1, keepdim=True)).to(dev)
                    c3.search(Q3[0], 100)
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    for i in range(3):
                        c3.search(Q3[i % 2], 100)
                    torch.cuda.synchronize(); el = time.perf_counter() - t0
                    kms = float(np.mean(c3.scan_times_ms(3)))
                    fl = 2.0 * 4096 * 10_000_000 * args.d
                    extra["flat_c3"] = {"workload": "Flat IP 10000000x%d f16, batch 4096, k=100 (BASELINE.json configs[2])" % args.d,
                                        "qps": 4096 * 3 / el, "ms_per_step": el / 3 * 1e3, "kernel_avg_ms": kms,
                                        "achieved_tflops": fl / (kms * 1e-3) / 1e12,
                                        "mfma_frac_of_2.5PF": fl / (kms * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS}
                    c3.close()
                else:
                    extra["flat_c3"] = {"skipped": "less than 120 GiB of HBM free"}
            except Exception as e:
                extra["flat_c3"] = {"error": repr(e)}
        print(json.dumps(line), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
