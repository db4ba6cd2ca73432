#!/usr/bin/env python3
"""bench.py - Stage-1 retrieval throughput on MI355X (contract: see the task statement).

Default workload (BASELINE.json configs[1]): Flat inner-product search over 1M x 4096 fp32 rows, batch-1
queries, k=10.  A "step" is one search of one query batch over the whole database.  With --gpus N (one rank
per GPU under torch.distributed.run) the SAME database is row-sharded over the ranks (strong scaling): every
rank scans N_total/N rows, the per-rank top-k lists are all-gathered over RCCL and merged; the north-star
shape (configs[4]: 12.5M fp16 rows PER GPU, batch 1 and 4096, Flat and list-sharded IVF) is then measured as
`extra.c5_weak`.  Data is synthetic: N(0,1) components, L2-normalised rows, generated on the device; queries
likewise.  Inputs are resident in HBM before timing starts.

Prints ONE JSON line on rank 0.  Everything besides `value` (side measurements under `extra`, the CPU
baseline) is labelled with the workload it ran on.
"""
import argparse
import copy
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
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
    ap.add_argument("--storage", default="f32", choices=["f32", "f16", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="only the bench line itself: no side measurements")
    ap.add_argument("--no-live-traffic", action="store_true", help="keep the committed HBM-traffic figure instead of two rocprofv3 --pmc passes of this command")
    ap.add_argument("--no-ivf-extra", action="store_true", help="skip the IVF4096 / 10M-row side measurement (configs[3])")
    ap.add_argument("--no-c5", action="store_true", help="--gpus N>1: skip the configs[4] per-GPU-shape side measurement")
    ap.add_argument("--extra-timeout", type=int, default=420,
                    help="--gpus N>1: seconds the side measurements may take before the line is printed without the rest (0: no limit)")
    ap.add_argument("--c5-rows", type=int, default=12_500_000, help="rows per GPU of the configs[4] side measurement")
    ap.add_argument("--cpu-sample-rows", type=int, default=1_000_000,
                    help="rows of the database the CPU baseline scans (default: all 1M of configs[1] = 16.4 GB of host RAM; "
                         "cut to 100k, labelled and extrapolated, when the host has less than 3x that free)")
    ap.add_argument("--workload", default="flat", choices=["flat", "ivf"],
                    help="flat = BASELINE configs[1] (the default bench line); ivf = configs[3] (IVF4096, nprobe 32; "
                         "with --gpus N the lists are sharded over the ranks and --n is the TOTAL row count)")
    ap.add_argument("--nlist", type=int, default=4096)
    ap.add_argument("--nprobe", type=int, default=32)
    ap.add_argument("--ivf-sorted", action="store_true", help="IVF workload: clustered rows arrive cluster by cluster (list pages contiguous)")
    ap.add_argument("--ivf-data", default="clustered", choices=["clustered", "iid"],
                    help="clustered: unit rows around 16384 random centres (sigma 0.6); iid: unit gaussian rows")
    return ap.parse_args()


# --------------------------------------------------------------------------------------------- launch
def launch_command(gpus, argv, port):
    """The command the contract names for N > 1: one rank per GPU under torch.distributed.run, same arguments."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def world_mismatch(gpus, env):
    """A launcher started W ranks but the command line says --gpus N != W: refuse rather than report n_gpus = W
    under a command that asked for N.  The one-rank rehearsal (SSMI_BENCH_FORCE_DIST=1, --gpus 1) passes."""
    ws = env.get("WORLD_SIZE")
    if ws is None or int(ws) == gpus:
        return None
    return "bench.py: --gpus %d but WORLD_SIZE=%s: start it as `python bench.py --gpus %d` or under torch.distributed.run with --nproc-per-node %d" % (
        gpus, ws, gpus, gpus)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def relay_ranks(gpus, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD process (this parent has made no
    GPU call, and it never replaces itself), pass the child's JSON line and return code on."""
    cmd = launch_command(gpus, argv, free_port())
    print("[bench] --gpus %d without WORLD_SIZE: starting %s" % (gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    p = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout.splitlines():
        try:
            if "metric" in json.loads(ln):
                line = ln
        except Exception:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    elif p.returncode == 0:
        print("[bench] the ranks printed no JSON line", file=sys.stderr)
        return 1
    return p.returncode


# --------------------------------------------------------------------------------------------- helpers
class Dist:
    """Rank bookkeeping + the timing rule of the contract (barrier and synchronize on both sides, max over ranks)."""

    def __init__(self):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        # SSMI_BENCH_FORCE_DIST=1: rehearse the multi-rank code path (process group, barriers, max over ranks,
        # sharded search with its RCCL all-gather, routed IVF add) on a single rank
        self.on = self.world > 1 or os.environ.get("SSMI_BENCH_FORCE_DIST") == "1"
        self.forced = self.on and self.world == 1
        # SSMI_BENCH_SHARE_GPU=1: rehearsal of a multi-rank run on a ONE-GPU box - every rank uses GPU 0 and the ranks
        # talk over gloo (RCCL refuses two ranks on one device); the numbers of such a run mean nothing, the code path does
        self.share = os.environ.get("SSMI_BENCH_SHARE_GPU") == "1"
        if self.share:
            self.local_rank = 0
        torch.cuda.set_device(self.local_rank)
        self.dev = torch.device("cuda", self.local_rank)
        if self.on:
            # RCCL prints a version banner on stdout when it initialises; stdout is reserved for the ONE JSON line,
            # so the banner goes to stderr: fd 1 points at fd 2 until the communicator is up
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            try:
                if self.share:
                    dist.init_process_group("gloo")
                else:
                    dist.init_process_group("nccl", device_id=self.dev)
                dist.barrier()
                torch.cuda.synchronize()
            finally:
                sys.stdout.flush()
                os.dup2(saved, 1)
                os.close(saved)

    def fence(self):
        torch.cuda.synchronize()
        if self.on:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(self, fn, steps, warmup):
        for i in range(warmup):
            fn(i)
        self.fence()
        t0 = time.perf_counter()
        for i in range(steps):
            fn(i)
        self.fence()
        el = time.perf_counter() - t0
        if self.on:
            t = torch.tensor([el], dtype=torch.float64, device=self.dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    def max_over_ranks(self, v):
        if not self.on:
            return float(v)
        t = torch.tensor([float(v)], dtype=torch.float64, device=self.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def devices(self):
        """What each rank runs on, gathered over the process group: the bench line names the GPUs it used."""
        p = torch.cuda.get_device_properties(self.dev)
        mine = {"rank": self.rank, "device": "cuda:%d" % self.local_rank, "name": p.name,
                "pci_bus_id": getattr(p, "pci_bus_id", None), "hbm_gib": round(p.total_memory / 2**30, 1)}
        if not self.on:
            return [mine]
        out = [None] * self.world
        dist.all_gather_object(out, mine)
        return out

    def close(self):
        if self.on:
            dist.barrier()
            dist.destroy_process_group()


def unit_queries(g, shape, dev):
    q = torch.randn(shape, generator=g, dtype=torch.float32)
    return (q / q.norm(dim=-1, keepdim=True)).to(dev)


def recall_at(I, Iref, k):
    I, Iref = np.asarray(I)[:, :k], np.asarray(Iref)[:, :k]
    return float(np.mean([len(set(a.tolist()) & set(b.tolist())) / k for a, b in zip(I, Iref)]))


def flat_workload_label(n, d, storage, batch, k, world):
    base = "Flat IP %dx%d %s, batch %d, k=%d" % (n, d, storage, batch, k)
    if world > 1:
        base += "; database row-sharded over %d GPUs" % world
    if (n, d, storage, k) == (1_000_000, 4096, "f32", 10) and batch in (1, 1024):
        return base + " (BASELINE.json configs[1])"
    if (n, d, storage, batch, k) == (10_000_000, 4096, "f16", 4096, 100) and world == 1:
        return base + " (BASELINE.json configs[2])"
    if (n, d, storage, k) == (100_000_000, 4096, "f16", 10) and world == 8:
        return base + " (BASELINE.json configs[4])"
    return base + " (custom shape, not a BASELINE.json config)"


def flat_roofline(local, n_local, d, storage, batch, launches):
    """Roofline of the dominant kernel of the last `launches` searches on `local`: algorithmic work of one launch
    (N*d*s bytes, 2*B*N*d flop, per GPU N/G) over the mean hipEvent duration of those launches,
    recorded by the library on the launch stream."""
    ms = local.scan_times_ms(min(launches, 256))
    avg = float(np.mean(ms)) if len(ms) else float("nan")
    elsize = 4 if storage == "f32" else 2
    t = local.last_timing()
    algo_bytes = n_local * d * elsize
    algo_flops = 2.0 * batch * n_local * d
    passes = max(1, t["last_scan_bytes"] // max(1, algo_bytes))
    if t["last_path"] == 1:
        roof = {"bound": "mfma", "achieved": algo_flops / (avg * 1e-3) / 1e12, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s"}
    else:
        roof = {"bound": "hbm", "achieved": algo_bytes / (avg * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s"}
    roof["frac"] = roof["achieved"] / roof["peak"]
    roof["traffic"] = None
    roof["algorithmic_bytes_per_launch"] = algo_bytes
    roof["algorithmic_flops_per_launch"] = algo_flops
    roof["kernel"] = local.last_kernel()
    roof["kernel_avg_ms"] = avg
    roof["launches_timed"] = int(len(ms))
    roof["db_passes_per_launch"] = int(passes)
    roof["flop_per_byte"] = algo_flops / algo_bytes
    return roof


def attach_committed_traffic(roof, args, world):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this exact workload (profiles/collect.sh);
    counters cannot be read from inside the timed process: this is the fallback for live_traffic() below, which measures
    the same figure with two child processes of this command under the profiler."""
    if (args.n, args.d, args.batch, args.storage, world) != (1_000_000, 4096, 1, "f32", 1):
        return
    for tag in ("r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", "%s_flat_f32_1M_b1_pmc.json" % tag)
        try:
            prof = json.load(open(path))
            for name, v in prof["kernels"].items():
                if "flat_scan_topk_kernel" in name and "hbm_bytes_per_launch_corrected" in v:
                    roof["traffic"] = v["hbm_bytes_per_launch_corrected"]
                    roof["traffic_source"] = ("profiles/%s_flat_f32_1M_b1_pmc.json (rocprofv3 --pmc FETCH_SIZE x2 gfx950 correction + "
                                              "WRITE_SIZE, separate passes; profiles/collect.sh) - committed, not measured by this run" % tag)
                    return
        except Exception:
            continue


def live_traffic(args, kernel):
    """HBM bytes per launch of `kernel`, measured by THIS run: two child processes of the same command (this file with
    --no-extra, fewer steps) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` - one counter per pass, no trace
    domain beside it, the program itself behind `--`, as MI355X_MICROARCH.md prescribes - and the corrected sum
    FETCH_SIZE*1024*2 + WRITE_SIZE*1024 of the kernel's launches.  Each pass runs in a process group of its own under a time
    limit; whatever goes wrong returns None and the line keeps the committed figure."""
    import csv, glob, shutil, signal, tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None  # this process is itself being profiled: no profiler inside a profiler
    tmp = tempfile.mkdtemp(prefix="ssmi_pmc_", dir="/tmp")
    vals = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [exe, "--output-format", "csv", "--pmc", ctr, "-d", os.path.join(tmp, ctr), "-o", "b", "--",
                   sys.executable, os.path.abspath(__file__), "--steps", "10", "--warmup", "2", "--no-extra", "--no-cpu-baseline",
                   "--n", str(args.n), "--d", str(args.d), "--k", str(args.k), "--batch", str(args.batch), "--storage", args.storage]
            env = dict(os.environ, TMPDIR="/tmp")
            for v in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
                env.pop(v, None)
            p = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env, start_new_session=True)
            try:
                rc = p.wait(timeout=150)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, signal.SIGKILL)   # the group this call started, nothing else
                except Exception:
                    pass
                return None
            if rc != 0:
                return None
            got = []
            for f in glob.glob(os.path.join(tmp, ctr, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if kernel in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                        got.append(float(r["Counter_Value"]))
            if len(got) < 5:
                return None
            vals[ctr] = (sum(got) / len(got), len(got))
        return {"bytes": vals["FETCH_SIZE"][0] * 1024 * 2 + vals["WRITE_SIZE"][0] * 1024, "launches": vals["FETCH_SIZE"][1]}
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


# --------------------------------------------------------------------------------------------- CPU baselines
def cpu_baseline_flat(index, d, k, sample_rows, n_total, dev):
    """The oracle timed on this box's host cores on a bounded sample of the same database: batch 1 (what the
    reference issues, core/index_manager.py:371) with the C/OpenMP port (or FAISS itself where it is installed),
    plus a batch-100 numpy/BLAS leg (Q @ X.T + argpartition: configs[0]'s shape of work)."""
    from oracle import c_oracle
    n = min(sample_rows, index.ntotal)
    try:  # the whole database when the host can hold it, else a labelled sample
        avail = os.sysconf("SC_AVPHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
    except (ValueError, OSError):
        avail = 0
    if avail < 3 * n * d * 4:
        n = min(n, 100_000)
    X = index.reconstruct_n(0, n)
    g = torch.Generator(device="cpu").manual_seed(97)
    Q = unit_queries(g, (100, d), "cpu").numpy()
    out = {"unit": "queries/s", "cpu_model": cpu_model(), "host_cpus_visible": os.cpu_count()}
    try:  # the reference's own engine, if this machine has it; the build image does not
        import faiss
    except Exception:
        faiss = None
    if faiss is not None:
        ref = faiss.IndexFlatIP(d)
        ref.add(X)
        ref.search(Q[:1], k)
        nqd, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 10.0 and nqd < 256:
            ref.search(Q[nqd % 100:nqd % 100 + 1], k)  # batch 1, distinct queries
            nqd += 1
        dt = time.perf_counter() - t0
        t0 = time.perf_counter(); ref.search(Q, k); dt100 = time.perf_counter() - t0
        out.update({"value": nqd / dt, "cores": faiss.omp_get_max_threads(), "kind": "reference",
                    "sample": f"faiss {faiss.__version__} IndexFlatIP on the first {n} rows x {d} fp32 of the same database, "
                              f"{nqd} distinct batch-1 queries, k={k}",
                    "batch_100_qps_on_sample": 100 / dt100})
    else:
        c_oracle.search(X[:1000], Q[:1], k)  # load + warm
        nqd, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 10.0 and nqd < 64:
            c_oracle.search(X, Q[nqd:nqd + 1], k)  # batch 1, a different query every time
            nqd += 1
        dt = time.perf_counter() - t0
        out.update({"value": nqd / dt, "cores": c_oracle.num_threads(), "kind": "port",
                    "sample": f"oracle/flat_ip.c (C/OpenMP) on the first {n} rows x {d} fp32 of the same database, "
                              f"{nqd} distinct batch-1 queries, k={k}"})
        # batch 100 through numpy/BLAS (sgemm + argpartition + sort of the k survivors)
        torch_threads = torch.get_num_threads()
        reps, t0 = 0, time.perf_counter()
        while True:
            S = Q @ X.T
            part = np.argpartition(-S, k, axis=1)[:, :k]
            sc = np.take_along_axis(S, part, axis=1)
            order = np.argsort(-sc, axis=1)
            _ = np.take_along_axis(part, order, axis=1)
            reps += 1
            if time.perf_counter() - t0 > 5.0 or reps >= 8:
                break
        dt100 = (time.perf_counter() - t0) / reps
        out["batch_100_numpy_blas"] = {"qps_on_sample": 100 / dt100, "extrapolated_full_db_qps": 100 / dt100 * n / n_total, "rows_scanned": n,
                                       "what": f"numpy {np.__version__} Q(100,{d}) @ X({n},{d}).T + argpartition + sort, fp32; "
                                               f"BLAS threads decided by the numpy build (torch reports {torch_threads})"}
    if n < n_total:
        out["extrapolated_full_db_qps"] = out["value"] * n / n_total
        out["extrapolation"] = f"linear in rows: value * {n}/{n_total} (a scan is DRAM-bound on the host)"
    else:
        out["rows_scanned"] = n   # the whole database of the bench line: nothing is extrapolated
    return out


def cpu_baseline_ivf(ivf, flat, d, k, nprobe, n_total, sample_rows=400_000):
    """IVF leg of the CPU baseline: oracle/ivf_flat.c (C/OpenMP) over the first `sample_rows` rows of the same
    database with the index's own centroids and list assignment, batch 1.  The coarse step costs what it costs at
    full size (same nlist); the list scan sees sample/N of the rows, so the full-size figure is extrapolated from
    the two parts separately."""
    from oracle import c_oracle
    n = min(sample_rows, flat.ntotal)
    X = flat.reconstruct_n(0, n)
    C = ivf.get_centroids()
    labels = ivf.assignments()[:n]
    g = c_oracle.GroupedIVF(X, C, labels)
    gq = torch.Generator(device="cpu").manual_seed(98)
    Q = X[np.random.default_rng(3).choice(n, 64, replace=False)] + 0.05 * unit_queries(gq, (64, d), "cpu").numpy()
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    g.search(Q[:1], k, nprobe)
    nqd, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 8.0 and nqd < 64:
        g.search(Q[nqd:nqd + 1], k, nprobe)
        nqd += 1
    t_total = (time.perf_counter() - t0) / nqd
    ncq, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 4.0 and ncq < 64:
        c_oracle.search(C, Q[ncq:ncq + 1], nprobe)
        ncq += 1
    t_coarse = (time.perf_counter() - t0) / ncq
    t_scan = max(t_total - t_coarse, 1e-9)
    return {"value": 1.0 / t_total, "unit": "queries/s", "cores": c_oracle.num_threads(), "kind": "port", "cpu_model": cpu_model(),
            "sample": f"oracle/ivf_flat.c (C/OpenMP) IVF{C.shape[0]} nprobe={nprobe} over the first {n} rows x {d} fp32 of the same database "
                      f"(the index's own centroids and assignment), {nqd} distinct batch-1 queries, k={k}",
            "coarse_ms": t_coarse * 1e3, "list_scan_ms_on_sample": t_scan * 1e3,
            "extrapolated_full_db_qps": 1.0 / (t_coarse + t_scan * n_total / n),
            "extrapolation": f"coarse step as measured + list scan x {n_total}/{n}"}


# --------------------------------------------------------------------------------------------- IVF workload
def ivf_rows_factory(args, dev, n_total, seed=1234):
    """Synthetic rows for the IVF workloads, generated on the device in chunks (same stream of rows whatever
    the chunking on one rank; every rank draws its own rows)."""
    d = args.d
    g = torch.Generator(device=dev).manual_seed(seed)
    gc = torch.Generator(device=dev).manual_seed(999)   # the cluster centres are common to all ranks
    ncl, sigma = 16384, 0.6
    centres = None
    if args.ivf_data == "clustered":
        centres = torch.randn((ncl, d), generator=gc, device=dev)
        centres /= centres.norm(dim=1, keepdim=True)

    def rows(m, first=None):
        x = torch.randn((m, d), generator=g, device=dev) * (sigma / d ** 0.5 if centres is not None else 1.0)
        if centres is not None:
            if first is None or not args.ivf_sorted:
                x += centres[torch.randint(0, ncl, (m,), generator=g, device=dev)]
            else:  # rows arrive cluster by cluster: the pages of a list end up next to each other
                x += centres[(torch.arange(first, first + m, device=dev) * ncl) // n_total]
        return x / x.norm(dim=1, keepdim=True)
    return rows


def run_ivf(args, D, emit=True, with_cpu=True):
    """BASELINE configs[3]: IVF<nlist> nprobe=<nprobe> over n x d f16 rows: coarse-quantizer training, add,
    batch-1 / batch-1000 / batch-4096 search, recall@k against the Flat index over the same rows.  With more
    than one rank the lists are sharded over the ranks (ShardedIVFFlat: routed add, one all-gather per search)
    and the Flat twin is row-sharded; n is the total row count."""
    import semantic_search_amd as ssa
    from semantic_search_amd.sharded import ShardedFlatIP, ShardedIVFFlat, shard_bounds
    dev, world, rank = D.dev, D.world, D.rank
    n, d, nlist, nprobe, k = args.n, args.d, args.nlist, args.nprobe, args.k
    b = shard_bounds(n, world)
    n_local, row0 = b[rank + 1] - b[rank], b[rank]
    rows = ivf_rows_factory(args, dev, n, seed=1234 + rank)
    sharded = D.on
    if sharded:
        ivf_s = ShardedIVFFlat(d, nlist, storage="f16", device=D.local_rank)
        flat_s = ShardedFlatIP(d, storage="f16", device=D.local_rank)
        ivf_s._force_pipeline = flat_s._force_pipeline = D.forced
        ivf, flat = ivf_s.local, flat_s.local
    else:
        ivf = ssa.IndexIVFFlat(ssa.IndexFlatIP(d, storage="f16"), d, nlist, ssa.METRIC_INNER_PRODUCT, storage="f16")
        flat = ssa.IndexFlatIP(d, storage="f16")
    ntrain = min(n, 256 * nlist)
    xt = ivf_rows_factory(args, dev, n, seed=99)(ntrain) if rank == 0 else None
    D.fence(); t0 = time.perf_counter()
    (ivf_s if sharded else ivf).train(xt)
    D.fence(); t_train = time.perf_counter() - t0
    del xt
    t_add = 0.0
    for s0 in range(0, n_local, 65536):
        m = min(65536, n_local - s0)
        x = rows(m, row0 + s0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if sharded:
            ivf_s.add_local(x, ids=np.arange(row0 + s0, row0 + s0 + m, dtype=np.int64))
        else:
            ivf.add(x)
        torch.cuda.synchronize(); t_add += time.perf_counter() - t0
        if os.environ.get("SSMI_BENCH_DEBUG") and (s0 // 65536) % 20 == 19:
            print("[ivf add] chunks to %d: %.3f s so far, mem free %.1f GiB" % (s0 // 65536 + 1, t_add, torch.cuda.mem_get_info()[0] / 2**30), file=sys.stderr, flush=True)
    # the Flat twin (ground truth of the recall figure) takes the same rows in a second pass of the same generator:
    # built beside the IVF index it doubles the memory in use while add() is being timed, and hipMalloc of the 1 GiB
    # storage segments then takes 10-30 ms instead of under one (measured: tools/time_ivf_add.py, WITH_FLAT=1)
    rows = ivf_rows_factory(args, dev, n, seed=1234 + rank)
    for s0 in range(0, n_local, 65536):
        flat.add(rows(min(65536, n_local - s0), row0 + s0))
    x = None
    t_add = D.max_over_ranks(t_add)
    if sharded:
        flat_s.finalize(device=dev)
        ivf_s.nprobe = nprobe
    else:
        ivf.nprobe = nprobe
    ls = ivf.list_sizes()
    qrows = ivf_rows_factory(args, dev, n, seed=4321)   # the same queries on every rank
    Q = qrows(1000)
    search = (lambda q: ivf_s.search(q, k)) if sharded else (lambda q: ivf.search(q, k))
    # the timed loop runs as a serving loop would: no hipEvent records between the kernels (each costs the
    # stream ~6 us); the kernel time for the roofline comes from the 32 timed sample launches below
    ivf.set_timing(False)
    elapsed = D.timed(lambda i: search(Q[i % 1000:i % 1000 + 1]), args.steps, args.warmup)
    ivf.set_timing(True)
    # algorithmic bytes = rows of the probed lists (this rank's) x d x 2 B, measured per query: the
    # bench reads back the prefix-sum total of each of 32 sample queries together with that launch's hipEvent time
    rows_s, ms_s = [], []
    for i in range(32):
        ivf.search(Q[i:i + 1], k)
        t = ivf.last_timing()
        rows_s.append(t["last_scan_bytes"] / (d * 2)); ms_s.append(t["last_scan_ms"])
    scan_kernel = ivf.last_kernel()
    probed_rows = float(np.mean(rows_s))
    algo_bytes = probed_rows * d * 2
    scan_ms_sample = float(np.mean(ms_s))
    small = {}
    for B in (8, 32):   # small batches still take the per-query scan: nq x (512/nq) blocks in one launch
        el = D.timed(lambda i: search(Q[(i * B) % 960:(i * B) % 960 + B]), 50, 3)
        small["batch_%d_qps" % B] = B * 50 / el
    search(Q)  # warm the batched path
    t_b = D.timed(lambda i: search(Q), 1, 0)
    batched_kernel = ivf.last_kernel()
    Q4 = qrows(4096)
    search(Q4)
    t_b4 = D.timed(lambda i: search(Q4), 1, 0)
    # batched roofline ("the roofline fraction uses the union" of the probed lists, what must minimally be
    # read): the lists a batch probes, recomputed here from the index's centroids, times this rank's list sizes
    Cdev = torch.from_numpy(ivf.get_centroids()).to(dev)
    ls_dev = torch.from_numpy(ls.astype(np.int64)).to(dev)

    def union_bytes(Qb):
        probed = torch.zeros((nlist,), dtype=torch.bool, device=dev)
        for s0 in range(0, Qb.shape[0], 1024):
            probed[torch.topk(Qb[s0:s0 + 1024] @ Cdev.T, min(nprobe, nlist), dim=1).indices.reshape(-1)] = True
        return float((ls_dev * probed).sum().item()) * d * 2
    ub1000, ub4096 = union_bytes(Q), union_bytes(Q4)
    del Q4, Cdev
    Di, Ii = search(Q)
    Df, If = (flat_s.search(Q, k) if sharded else flat.search(Q, k))
    torch.cuda.synchronize()
    recall = recall_at(Ii.cpu().numpy(), If.cpu().numpy(), k)
    ranks = {"world": dist.get_world_size() if D.on else 1, "backend": dist.get_backend() if D.on else None, "devices": D.devices()}
    frac = algo_bytes / (scan_ms_sample * 1e-3) / 1e9 / HBM_PEAK_GBS
    line = {
        "metric": "queries/sec at k=%d, %d-d (IVF%d nprobe=%d, batch 1)" % (k, d, nlist, nprobe),
        "value": args.steps / elapsed, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f16 storage, f32 accumulate", "data": "synthetic (%s)" % args.ivf_data,
        "config": {"workload": "IVF%d nprobe=%d over %dx%d f16%s, batch 1, k=%d%s" % (
                       nlist, nprobe, n, d, " (BASELINE.json configs[3])" if (nlist, nprobe, n, d, world) == (4096, 32, 10_000_000, 4096, 1) else "",
                       k, "; lists sharded over %d GPUs (list l on rank l %% %d)" % (world, world) if world > 1 else ""),
                   "rows_total": n, "dim": d, "nlist": nlist, "nprobe": nprobe, "k": k, "train_rows": ntrain,
                   "parallelism": "listshard%d" % world},
        "roofline": {"bound": "hbm", "achieved": frac * HBM_PEAK_GBS, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frac, "traffic": None,
                     "algorithmic_bytes_per_launch": algo_bytes,
                     "algorithmic_bytes_note": "mean rows of the probed lists held by rank 0 over 32 sample queries (%d; nprobe*N/nlist/world = %d) x d x 2 B, "
                                               "over the mean hipEvent time of those launches" % (int(probed_rows), int(nprobe * n / nlist / world)),
                     "kernel": scan_kernel, "kernel_avg_ms": scan_ms_sample},
        "extra": {"ranks": ranks, "train_s": t_train, "add_s": t_add, "add_vectors_per_s": n / t_add,
                  "list_sizes_rank0": {"min": int(ls.min()), "median": float(np.median(ls)), "max": int(ls.max())},
                  **small, "batch_1000_qps": 1000 / t_b, "batch_4096_qps": 4096 / t_b4, "batched_kernel": batched_kernel,
                  "batch_1000_hbm_frac_end_to_end": ub1000 / t_b / 1e9 / HBM_PEAK_GBS,
                  "batch_4096_hbm_frac_end_to_end": ub4096 / t_b4 / 1e9 / HBM_PEAK_GBS,
                  "batched_roofline_note": "bytes = rows of the UNION of the lists the batch probes (rank 0's share) x d x 2 B, "
                                           "over the whole search call (coarse step, probe-table inversion, list scan, merge)",
                  "recall_at_%d_vs_flat" % k: recall,
                  "recall_note": "1000 queries; ground truth = IndexFlatIP (f16 rows) over the same vectors"},
    }
    if with_cpu and world == 1 and not args.no_cpu_baseline:
        try:
            line["cpu_baseline"] = cpu_baseline_ivf(ivf, flat, d, k, nprobe, n)
        except Exception as e:
            line["cpu_baseline"] = {"error": repr(e)}
    if with_cpu and not sharded:
        # k-means of the coarse quantizer, both modes on the same rows: spherical (default; what FAISS's IndexIVF does for
        # the inner product [FAISS-lib]) against plain means (SSMI_IVF_SPHERICAL=0, rounds 1-2): list sizes and recall@k
        modes = {"spherical": {"list_sizes": line["extra"]["list_sizes_rank0"], "recall_at_%d_vs_flat" % k: recall,
                               "batch_1_qps": line["value"]}}
        ivf.close()
        os.environ["SSMI_IVF_SPHERICAL"] = "0"
        try:
            ivf = ssa.IndexIVFFlat(ssa.IndexFlatIP(d, storage="f16"), d, nlist, ssa.METRIC_INNER_PRODUCT, storage="f16")
        finally:
            del os.environ["SSMI_IVF_SPHERICAL"]
        ivf.train(ivf_rows_factory(args, dev, n, seed=99)(ntrain))
        rows = ivf_rows_factory(args, dev, n, seed=1234 + rank)
        for s0 in range(0, n_local, 65536):
            ivf.add(rows(min(65536, n_local - s0), row0 + s0))
        ivf.nprobe = nprobe
        ivf.set_timing(False)
        el = D.timed(lambda i: ivf.search(Q[i % 1000:i % 1000 + 1], k), args.steps, args.warmup)
        _, Ip = ivf.search(Q, k)
        torch.cuda.synchronize()
        ls2 = ivf.list_sizes()
        modes["plain_means"] = {"list_sizes": {"min": int(ls2.min()), "median": float(np.median(ls2)), "max": int(ls2.max())},
                                "recall_at_%d_vs_flat" % k: recall_at(Ip.cpu().numpy(), If.cpu().numpy(), k),
                                "batch_1_qps": args.steps / el}
        line["extra"]["kmeans_modes"] = modes
    ivf.close(); flat.close()
    if emit and rank == 0:
        print(json.dumps(line), flush=True)
    return line


# --------------------------------------------------------------------------------------------- side measurements
def kernel_clock(index, tflops):
    """Shader clock under the last 256-query MFMA launch of `index` (workgroup 0's own s_memtime / s_memrealtime stamps,
    ssmi_index_get_kernel_clock).  The kernel is power-limited: the same binary holds 1.6-1.9 GHz depending on the box and
    on its operand traffic, so the fraction of the 2.5 PF peak (quoted at 2.4 GHz) moves with the box; the fraction of the
    peak AT THE CLOCK THE CHIP HELD is the share of matrix-pipe issue slots the kernel fills, whatever the box."""
    try:
        c = index.last_kernel_clock()
    except RuntimeError:
        return {}
    out = {"shader_clock_mhz": c["mhz"], "kernel_cycles_workgroup0": c["cycles"]}
    if tflops is not None:
        out["mfma_frac_at_that_clock"] = tflops / (MFMA_F16_PEAK_TFLOPS * c["mhz"] / 2400.0)
    return out


def side_batches(index, local, n_local, args, D, g):
    """Batch 1024 / 4096 over the same (sharded) database: the MFMA contraction (on the fp16 shadow copy when the
    storage is f32), with the recall@k of that path against the exact scan of the stored rows."""
    out = {}
    for B, steps in ((1024, 5), (4096, 3)):
        Qb = unit_queries(g, (2, B, args.d), D.dev)
        el = D.timed(lambda i: index.search(Qb[i % 2], args.k), steps, 1)
        roof = flat_roofline(local, n_local, args.d, "f16", B, steps)
        e = {"workload": flat_workload_label(args.n, args.d, args.storage, B, args.k, D.world),
             "qps": B * steps / el, "ms_per_step": el / steps * 1e3, "kernel_avg_ms": roof["kernel_avg_ms"],
             "kernel": roof["kernel"], "achieved_tflops_per_gpu": roof["algorithmic_flops_per_launch"] / (roof["kernel_avg_ms"] * 1e-3) / 1e12,
             "operand_dtype": ("%s (shadow copy of the f32 rows), f32 accumulate" % (local.shadow_dtype or "f16")) if args.storage == "f32"
                              else "%s, f32 accumulate" % args.storage}
        e["mfma_frac_of_2.5PF"] = e["achieved_tflops_per_gpu"] / MFMA_F16_PEAK_TFLOPS
        e.update(kernel_clock(local, e["achieved_tflops_per_gpu"]))
        if D.world == 1:
            # recall of the batched path against the exact streaming scan (fp32 queries, stored rows as they are)
            _, Ib = index.search(Qb[0], args.k)
            sel = torch.arange(0, B, B // 64, device=D.dev)[:64]
            os.environ["SSMI_NO_SHADOW"] = "1"; os.environ["SSMI_MFMA_MIN_B"] = "1000000"
            try:
                _, Ie = local.search(Qb[0][sel].contiguous(), args.k)
            finally:
                del os.environ["SSMI_NO_SHADOW"]; del os.environ["SSMI_MFMA_MIN_B"]
            torch.cuda.synchronize()
            e["recall_at_%d_vs_exact_scan" % args.k] = recall_at(Ib[sel].cpu().numpy(), Ie.cpu().numpy(), args.k)
            e["recall_note"] = ("64 of the queries; exact scan = K1 with fp32 queries over the %s rows; the batched path rounds queries%s to %s"
                                % (args.storage, " and rows (shadow)" if args.storage == "f32" else "",
                                   (local.shadow_dtype or "f16") if args.storage == "f32" else args.storage))
        out["batch_%d" % B] = e
    try:  # counter-based MFMA utilisation of the batch-4096 contraction (committed PMC pass, labelled as such)
        for tag in ("r03", "r02", "r01"):
            path = os.path.join(ROOT, "profiles", "%s_mfma_f16_1M_b4096_pmc.json" % tag)
            if os.path.exists(path):
                v = next(iter(json.load(open(path))["kernels"].values()))
                out["mfma_util_pmc"] = {"value": v["mfma_util"], "kernel_cycles": v["kernel_cycles"],
                                        "definition": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE/8), rocprofv3 --pmc, Flat 1M x 4096 f16, "
                                                      "batch 4096 (profiles/%s_mfma_f16_1M_b4096_pmc.json) - committed, not measured by this run" % tag}
                break
    except Exception:
        pass
    return out


def side_small_batches(index, local, n_local, args, D, g):
    """2 / 4 / 8 queries per search over the bench's database: the exact streaming scan (K1 below five queries on
    fp32 rows of 8 KiB and more, K1m - the same scan with its multiply-adds on the fp32 matrix instruction - for 5-8),
    the batch sizes the reference's multi-index and exclusion flows issue (nodes/search_nodes.py:869-878)."""
    out = {}
    for B in (2, 4, 8):
        Qb = unit_queries(g, (2, B, args.d), D.dev)
        el = D.timed(lambda i: index.search(Qb[i % 2], args.k), 20, 2)
        roof = flat_roofline(local, n_local, args.d, args.storage, B, 20)
        out["batch_%d" % B] = {"qps": B * 20 / el, "ms_per_step": el / 20 * 1e3, "kernel": roof["kernel"], "kernel_avg_ms": roof["kernel_avg_ms"],
                               "hbm_frac": roof["frac"] if roof["bound"] == "hbm" else None, "bound": roof["bound"]}
    return out


def add_same_unit_rows(indexes, n, d, dev, seed, bf16_valued=False):
    """The same device-generated unit rows (N(0,1) components, L2-normalised) appended to every index of the
    list, through the product's own add() (16-bit storage rounds to nearest even there).  bf16_valued: the rows are
    rounded to bf16 and widened again first - what the reference's embedder hands over (bf16 model outputs cast to
    fp32, core/model_wrapper.py:201-203)."""
    gd = torch.Generator(device=dev).manual_seed(seed)
    for s0 in range(0, n, 65536):
        x = torch.randn((min(65536, n - s0), d), generator=gd, device=dev)
        x /= x.norm(dim=1, keepdim=True)
        if bf16_valued:
            x = x.bfloat16().float()
        for ix in indexes:
            ix.add(x)


def side_storage_recall(args, D, g):
    """north_star: 'the fp16-storage recall@10 loss stated separately' - and the same for bf16 storage, side by side.
    The same 1M device-generated rows in an f32, an f16 and a bf16 index, 256 queries: recall@10 of each 16-bit
    storage against the fp32 result, on the exact scan (fp32 queries) and on the MFMA path (queries narrowed to the
    rows' type).  Twice: on generic fp32 rows, and on bf16-VALUED rows and queries (the reference's producer)."""
    import semantic_search_amd as ssa
    n, d, k, nq = 1_000_000, args.d, 10, 256
    out = {"workload": "Flat IP %dx%d, the same rows stored as f32, f16 and bf16 (rounded once at add), %d queries, k=%d" % (n, d, nq, k)}
    for label, valued in (("generic_fp32_rows", False), ("bf16_valued_rows_and_queries", True)):
        a, b, c = ssa.IndexFlatIP(d, storage="f32"), ssa.IndexFlatIP(d, storage="f16"), ssa.IndexFlatIP(d, storage="bf16")
        add_same_unit_rows([a, b, c], n, d, D.dev, 1234, bf16_valued=valued)
        Q = unit_queries(g, (nq, d), D.dev)
        if valued:
            Q = Q.bfloat16().float()
        os.environ["SSMI_NO_SHADOW"] = "1"; os.environ["SSMI_MFMA_MIN_B"] = "1000000"
        try:
            D32, I32 = a.search(Q, k)
            D16, I16 = b.search(Q, k)
            Db, Ib = c.search(Q, k)
        finally:
            del os.environ["SSMI_NO_SHADOW"]; del os.environ["SSMI_MFMA_MIN_B"]
        D16m, I16m = b.search(Q, k)
        kern = b.last_kernel()
        Dbm, Ibm = c.search(Q, k)
        Dsh, Ish = a.search(Q, k)   # fp32 rows through their 16-bit shadow (auto: bf16 while lossless, else fp16)
        torch.cuda.synchronize()
        ref = I32.cpu().numpy()
        e = {"recall_at_10_f16_storage_exact_scan_vs_f32": recall_at(I16.cpu().numpy(), ref, k),
             "recall_at_10_f16_storage_mfma_vs_f32": recall_at(I16m.cpu().numpy(), ref, k),
             "recall_at_10_bf16_storage_exact_scan_vs_f32": recall_at(Ib.cpu().numpy(), ref, k),
             "recall_at_10_bf16_storage_mfma_vs_f32": recall_at(Ibm.cpu().numpy(), ref, k),
             "recall_at_10_f32_storage_via_shadow_vs_f32_exact": recall_at(Ish.cpu().numpy(), ref, k),
             "shadow_dtype_chosen": a.shadow_dtype, "mfma_kernel": kern,
             "ids_identical_bf16_storage_exact_scan": bool(torch.equal(Ib, I32)),
             "ids_identical_bf16_storage_mfma": bool(torch.equal(Ibm, I32)),
             "ids_identical_f32_via_shadow": bool(torch.equal(Ish, I32)),
             "max_rel_score_diff_f16": float((torch.abs(D16 - D32) / torch.abs(D32)).max().item()),
             "max_rel_score_diff_bf16": float((torch.abs(Db - D32) / torch.abs(D32)).max().item())}
        out[label] = e
        a.close(); b.close(); c.close()
    out["decision"] = ("the shadow of fp32 rows is bf16 while every row is bf16-valued (lossless: ids identical to the exact scan) "
                       "and fp16 otherwise (11 significant bits against 8: the smaller recall loss on generic rows)")
    return out


def side_flat_c3(args, D, g):
    """BASELINE configs[2]: Flat 10M x 4096 f16, batch 4096, k=100 on one GPU."""
    import semantic_search_amd as ssa
    n, B, k = 10_000_000, 4096, 100
    c3 = ssa.IndexFlatIP(args.d, storage="f16")
    torch.cuda.empty_cache()
    free_b, _ = torch.cuda.mem_get_info()
    twin = free_b > 245 * (1 << 30)   # the fp32 twin of the same rows (164 GB = 153 GiB) fits next to the fp16 index (82 GB = 76 GiB)
    c32 = ssa.IndexFlatIP(args.d, storage="f32") if twin else None
    add_same_unit_rows([c3, c32] if twin else [c3], n, args.d, D.dev, 1234)
    Q3 = unit_queries(g, (2, B, args.d), D.dev)
    c3.search(Q3[0], k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(3):
        c3.search(Q3[i % 2], k)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    clk = kernel_clock(c3, None)
    roof = flat_roofline(c3, n, args.d, "f16", B, 3)
    out = {"workload": flat_workload_label(n, args.d, "f16", B, k, 1), "qps": B * 3 / el, "ms_per_step": el / 3 * 1e3,
           "kernel": roof["kernel"], "kernel_avg_ms": roof["kernel_avg_ms"], "achieved_tflops": roof["achieved"],
           "mfma_frac_of_2.5PF": roof["frac"]}
    if clk:
        clk["mfma_frac_at_that_clock"] = roof["achieved"] / (MFMA_F16_PEAK_TFLOPS * clk["shader_clock_mhz"] / 2400.0)
        out.update(clk)
    # recall@100 of the MFMA path (fp16 queries) against the exact scan of the same fp16 rows (fp32 queries)
    _, Ib = c3.search(Q3[0], k)
    sel = torch.arange(0, B, B // 32, device=D.dev)[:32]
    os.environ["SSMI_MFMA_MIN_B"] = "1000000"
    try:
        _, Ie = c3.search(Q3[0][sel].contiguous(), k)
    finally:
        del os.environ["SSMI_MFMA_MIN_B"]
    torch.cuda.synchronize()
    out["recall_at_100_mfma_vs_exact_scan_of_the_f16_rows"] = recall_at(Ib[sel].cpu().numpy(), Ie.cpu().numpy(), k)
    # fp16 storage against fp32 storage of the same 10M rows, when both fit (164 GB + 82 GB)
    if twin:
        os.environ["SSMI_NO_SHADOW"] = "1"; os.environ["SSMI_MFMA_MIN_B"] = "1000000"
        try:
            _, I32 = c32.search(Q3[0][sel].contiguous(), k)
        finally:
            del os.environ["SSMI_NO_SHADOW"]; del os.environ["SSMI_MFMA_MIN_B"]
        torch.cuda.synchronize()
        I32 = I32.cpu().numpy()
        out["recall_at_100_f16_storage_mfma_vs_f32_storage_exact"] = recall_at(Ib[sel].cpu().numpy(), I32, k)
        out["recall_at_10_f16_storage_mfma_vs_f32_storage_exact"] = recall_at(Ib[sel].cpu().numpy(), I32, 10)
        out["recall_at_100_f16_storage_exact_vs_f32_storage_exact"] = recall_at(Ie.cpu().numpy(), I32, k)
        out["recall_note"] = "32 of the 4096 queries; f32 twin = the same 10M device-generated rows kept in fp32 (164 GB), exact fp32 scan"
        c32.close()
    else:
        out["recall_note"] = "f32 twin of the 10M rows skipped: less than 245 GiB of HBM free"
    c3.close()
    return out


def side_c5_weak(args, D, g):
    """The north-star shape (BASELINE configs[4]): `c5_rows` fp16 rows PER GPU (12.5M = 102.4 GB; 100M rows on 8
    GPUs), row-sharded Flat, batch 1 and batch 4096, k=10, through the all-gather + merge; per-rank and aggregate
    roofline fractions.  Weak scaling: per-GPU work is fixed as the rank count grows."""
    from semantic_search_amd.sharded import ShardedFlatIP
    n_local, d, k, world = args.c5_rows, args.d, 10, D.world
    idx = ShardedFlatIP(d, storage="f16", device=D.local_rank)
    idx.add_local_random_unit(n_local, seed=5000 + D.rank)
    idx.finalize(device=D.dev)
    idx._force_pipeline = D.forced
    out = {"workload": "Flat IP %dx%d f16 row-sharded over %d GPUs (%d rows per GPU), k=%d%s" % (
               n_local * world, d, world, n_local, k, " (BASELINE.json configs[4])" if (n_local * world, world) == (100_000_000, 8) else ""),
           "scaling": "weak", "rows_per_gpu": n_local, "rows_total": n_local * world}
    Q1 = unit_queries(g, (8, 1, d), D.dev)
    el = D.timed(lambda i: idx.search(Q1[i % 8], k), 30, 3)
    r1 = flat_roofline(idx.local, n_local, d, "f16", 1, 30)
    out["batch_1"] = {"qps": 30 / el, "ms_per_step": el / 30 * 1e3, "kernel": r1["kernel"], "kernel_avg_ms_rank0": r1["kernel_avg_ms"],
                      "hbm_frac_rank0_kernel": r1["frac"],
                      "aggregate_hbm_frac_end_to_end": n_local * world * d * 2 / (el / 30) / 1e9 / (HBM_PEAK_GBS * world),
                      "target": ">= 0.70 of aggregate HBM roofline"}
    Qb = unit_queries(g, (2, 4096, d), D.dev)
    el = D.timed(lambda i: idx.search(Qb[i % 2], k), 3, 1)
    r2 = flat_roofline(idx.local, n_local, d, "f16", 4096, 3)
    out["batch_4096"] = {"qps": 4096 * 3 / el, "ms_per_step": el / 3 * 1e3, "kernel": r2["kernel"], "kernel_avg_ms_rank0": r2["kernel_avg_ms"],
                         "mfma_frac_rank0_kernel": r2["frac"],
                         "aggregate_mfma_frac_end_to_end": 2.0 * 4096 * n_local * world * d / (el / 3) / 1e12 / (MFMA_F16_PEAK_TFLOPS * world),
                         "target": ">= 0.50 of aggregate dense fp16 MFMA peak"}
    idx.local.close()
    return out


# --------------------------------------------------------------------------------------------- main
def main():
    args = parse()
    msg = world_mismatch(args.gpus, os.environ)
    if msg:
        raise SystemExit(msg)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(relay_ranks(args.gpus, sys.argv[1:]))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (MI355X)")
    D = Dist()
    if args.workload == "ivf":
        run_ivf(args, D)
        D.close()
        return
    world, rank, dev = D.world, D.rank, D.dev

    from semantic_search_amd.sharded import ShardedFlatIP, shard_bounds

    bounds = shard_bounds(args.n, world)
    n_local = bounds[rank + 1] - bounds[rank]
    index = ShardedFlatIP(args.d, storage=args.storage, device=D.local_rank)
    index.add_local_random_unit(n_local, seed=1234 + rank)
    index.finalize(device=dev)
    index._force_pipeline = D.forced  # rehearsal: take the all-gather + merge path although there is one rank

    g = torch.Generator(device="cpu").manual_seed(4321)
    nsets = 8  # rotate over a few query batches so no step repeats the previous one
    Q = unit_queries(g, (nsets, args.batch, args.d), dev)

    elapsed = D.timed(lambda i: index.search(Q[i % nsets], args.k), args.steps, args.warmup)
    roof = flat_roofline(index.local, n_local, args.d, args.storage if index.local.last_timing()["last_path"] == 0 else "f16",
                         args.batch, args.steps)
    if index.local.last_timing()["last_path"] == 0:
        roof.pop("algorithmic_flops_per_launch", None)
    attach_committed_traffic(roof, args, world)

    extra = {"ranks": {"world": dist.get_world_size() if D.on else 1,
                       "backend": dist.get_backend() if D.on else None,   # "nccl" is RCCL on ROCm
                       "devices": D.devices()}}
    extra["rccl_world"] = extra["ranks"]["world"] if extra["ranks"]["backend"] == "nccl" else None
    line = None
    if rank == 0:
        line = {
            "metric": "queries/sec at k=%d, %d-d" % (args.k, args.d),
            "value": args.batch * args.steps / elapsed,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": ("f32" if args.storage == "f32" and roof["bound"] == "hbm" else
                      "%s operands, f32 accumulate" % ((index.local.shadow_dtype or "f16") if args.storage == "f32" else args.storage)),
            "data": "synthetic",
            "config": {
                "workload": flat_workload_label(args.n, args.d, args.storage, args.batch, args.k, world),
                "rows_total": args.n, "rows_per_gpu": n_local, "dim": args.d, "k": args.k, "batch": args.batch,
                "storage": args.storage, "parallelism": "rowshard%d" % world,
            },
            "roofline": roof,
            "extra": extra,
        }

    def guarded(name, fn):
        # a side measurement must never take the bench line down; every rank takes the same path
        try:
            extra[name] = fn()
        except Exception as e:
            extra[name] = {"error": repr(e)}

    # ... nor may one that hangs (a rank that raises inside a collective leaves the others waiting in it): with several
    # ranks the side measurements run under a watchdog.  When it fires, rank 0 prints the line as it stands - the timed
    # region is long finished - and every rank leaves without taking part in another collective.
    emitted = threading.Lock()

    def emit_line():
        if rank == 0 and emitted.acquire(blocking=False):
            print(json.dumps(line), flush=True)

    def watchdog_fired():
        extra["watchdog"] = "side measurements did not finish within %d s: cut off, the line is complete without them" % args.extra_timeout
        emit_line()
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)

    watchdog = None
    if D.on and not args.no_extra and args.extra_timeout > 0:
        watchdog = threading.Timer(args.extra_timeout, watchdog_fired)
        watchdog.daemon = True
        watchdog.start()

    if not args.no_extra and not args.no_live_traffic and world == 1 and rank == 0:
        # roofline.traffic from a PMC pass of this very command instead of the committed figure (which stays as the fallback)
        try:
            lt = live_traffic(args, roof["kernel"])
        except Exception:
            lt = None
        if lt:
            if roof.get("traffic") is not None:
                roof["traffic_committed"] = roof["traffic"]
            roof["traffic"] = lt["bytes"]
            roof["traffic_source"] = ("measured by this run: child processes of the same command under rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE "
                                      "(separate passes), mean over %d launches of %s, FETCH_SIZE x 1024 x 2 (gfx950 correction) + WRITE_SIZE x 1024"
                                      % (lt["launches"], roof["kernel"]))
    if not args.no_extra:
        guarded("batches", lambda: side_batches(index, index.local, n_local, args, D, g))
        guarded("small_batches", lambda: side_small_batches(index, index.local, n_local, args, D, g))
        if "error" not in extra["batches"]:
            extra.update(extra.pop("batches"))
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        line["cpu_baseline"] = cpu_baseline_flat(index.local, args.d, args.k, args.cpu_sample_rows, args.n, dev)
    index.local.close()

    if not args.no_extra and world == 1 and not D.forced:
        guarded("storage_recall", lambda: side_storage_recall(args, D, g))
        if not args.no_ivf_extra:
            # BASELINE configs[3], IVF4096 nprobe 32 over 10M x 4096 f16 rows, the same procedure as
            # `bench.py --workload ivf`; clustered rows (IVF's use case) and iid rows (SURVEY.md 8d's generator)
            def ivf_side(data):
                free_b, _ = torch.cuda.mem_get_info()
                if free_b < 200 * (1 << 30):
                    return {"skipped": "less than 200 GiB of HBM free"}
                a2 = copy.copy(args)
                a2.n, a2.steps, a2.warmup, a2.ivf_data, a2.ivf_sorted = 10_000_000, 200, 10, data, False
                iv = run_ivf(a2, D, emit=False, with_cpu=(data == "clustered"))
                out = {"workload": iv["config"]["workload"], "data": iv["data"], "batch_1_qps": iv["value"],
                       "scan_kernel": iv["roofline"]["kernel"], "scan_kernel_hbm_frac": iv["roofline"]["frac"],
                       "scan_kernel_ms": iv["roofline"]["kernel_avg_ms"],
                       "probed_bytes_per_query": iv["roofline"]["algorithmic_bytes_per_launch"], **iv["extra"]}
                if "cpu_baseline" in iv:
                    out["cpu_baseline"] = iv["cpu_baseline"]
                return out
            guarded("ivf_c4", lambda: ivf_side("clustered"))
            guarded("ivf_c4_iid", lambda: ivf_side("iid"))
        def c3():
            free_b, _ = torch.cuda.mem_get_info()
            if free_b < 120 * (1 << 30):
                return {"skipped": "less than 120 GiB of HBM free"}
            return side_flat_c3(args, D, g)
        guarded("flat_c3", c3)
    if not args.no_extra and D.on and not args.no_c5:
        guarded("c5_weak", lambda: side_c5_weak(args, D, g))
        def c5_ivf():
            a2 = copy.copy(args)
            a2.n, a2.steps, a2.warmup, a2.ivf_data, a2.ivf_sorted = args.c5_rows * world, 100, 5, "clustered", False
            iv = run_ivf(a2, D, emit=False, with_cpu=False)
            return {"workload": iv["config"]["workload"], "data": iv["data"], "scaling": "weak", "batch_1_qps": iv["value"],
                    "ms_per_step": iv["ms_per_step"], "scan_kernel": iv["roofline"]["kernel"],
                    "scan_kernel_hbm_frac_rank0": iv["roofline"]["frac"], "scan_kernel_ms_rank0": iv["roofline"]["kernel_avg_ms"],
                    "probed_bytes_per_query_rank0": iv["roofline"]["algorithmic_bytes_per_launch"], **iv["extra"]}
        guarded("c5_weak_ivf", c5_ivf)

    if watchdog is not None:
        watchdog.cancel()
    emit_line()
    D.close()


if __name__ == "__main__":
    main()
