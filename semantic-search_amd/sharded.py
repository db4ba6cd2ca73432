"""Row-sharded Flat search and list-sharded IVF search across the GPUs of one node: one process per GPU.

Each rank owns one contiguous block of database rows in its own IndexFlatIP
(global id = local row + the rank's row offset, which keeps the reference's
implicit sequential ids, core/index_manager.py:319-325).  A search is
    local top-k (HIP kernels)  ->  all_gather of the (B,k) scores and ids over
    RCCL/xGMI  ->  merge of the `world` lists (ssmi_topk_merge)
The collective moves 12*B*k bytes per rank (120 B at B=1, k=10), so it is
latency-bound; there is no other data-path communication.

The reference has no distributed code; this is new.
"""
import os
from typing import Callable, List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(ntotal: int, world: int) -> List[int]:
    """Contiguous, near-equal row blocks: bounds[r] .. bounds[r+1] is rank r's block."""
    base, rem = divmod(int(ntotal), int(world))
    bounds = [0]
    for r in range(world):
        bounds.append(bounds[-1] + base + (1 if r < rem else 0))
    return bounds


def exchange_offsets(n_local: int, group=None, device="cpu") -> Tuple[int, int]:
    """(row offset of this rank, global ntotal) from every rank's local row count."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    mine = torch.tensor([int(n_local)], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(counts, mine, group=group)
    counts = [int(c.item()) for c in counts]
    return sum(counts[:rank]), sum(counts)


def packed_result_buffers(B: int, k: int, device) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """One int64 buffer holding a rank's (B,k) int64 ids followed by its (B,k) fp32 scores, so that a
    sharded search needs ONE all-gather (the collective is latency-bound: 12*B*k bytes per rank)."""
    n = B * k
    words = n + (n + 1) // 2
    buf = torch.empty((words,), dtype=torch.int64, device=device)
    I = buf[:n].view(B, k)
    D = buf[n:].view(torch.float32)[:n].view(B, k)
    return buf, D, I


def gather_packed(buf: torch.Tensor, B: int, k: int, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """all_gather the packed per-rank buffers; returns strided (world, B, k) views of scores and ids."""
    world = dist.get_world_size(group)
    words = buf.numel()
    if buf.is_cuda and dist.get_backend(group) != "nccl":
        # a host-only backend (gloo: ranks that share one GPU, tools/two_rank_check.py): the collective runs on host copies
        gh = torch.empty((world * words,), dtype=torch.int64)
        dist.all_gather_into_tensor(gh, buf.cpu(), group=group)
        g = gh.to(buf.device)
    else:
        g = torch.empty((world * words,), dtype=torch.int64, device=buf.device)
        dist.all_gather_into_tensor(g, buf, group=group)
    g = g.view(world, words)
    n = B * k
    Ig = g[:, :n].unflatten(1, (B, k))
    Dg = g[:, n:].view(torch.float32)[:, :n].unflatten(1, (B, k))
    return Dg, Ig


class _Pipeline:
    """Three-stream pipeline of a sharded search on one GPU:

        scan stream   local top-k kernels of step i+1, i+2, ...   (never waits for a merge)
        RCCL stream   all-gather of step i       (torch's process-group stream)
        merge stream  merge of step i-1

    so that back-to-back searches are bounded by the slowest stage, not by their sum (the
    collective is latency-bound: 12*B*k bytes per rank).  OPT-IN (SSMI_SHARD_PIPELINE=1): measured on this
    ROCm 7.2 stack (tools/debug_pipeline.py, 125k x 4096 fp32 shard, batch 1) the torch-level
    three-stream form costs the host ~290 us per step (raw HIP launches and event waits across streams cost 2-14 us
    under the same load, so the cost is in the torch/RCCL stream plumbing, not in HIP), which makes it slower (397 us/step) than the plain one-stream sequence scan -> all-gather -> merge
    (337 us/step; the scan alone is 305 us), so the one-stream sequence is the default.  The caller's current stream only waits
    for the merge of its own step, so results behave like any other torch op's outputs.
    Per (B,k) shape a small ring of packed result buffers is reused; a slot is rewritten only
    after the merge that read it has finished."""

    RING = 4

    def __init__(self, device: torch.device):
        self.device = device
        self.scan = torch.cuda.Stream(device)
        self.merge = torch.cuda.Stream(device)
        self.rings = {}

    def slot(self, B: int, k: int, world: int):
        ring = self.rings.get((B, k))
        if ring is None:
            ring = {"next": 0, "slots": []}
            for _ in range(self.RING):
                buf, D, I = packed_result_buffers(B, k, self.device)
                g = torch.empty((world * buf.numel(),), dtype=torch.int64, device=self.device)
                # events are created once per slot and re-recorded: destroying a still-pending event blocks the host
                ring["slots"].append({"buf": buf, "D": D, "I": I, "g": g, "used": False,
                                      "ready": torch.cuda.Event(), "done": torch.cuda.Event()})
            self.rings[(B, k)] = ring
        s = ring["slots"][ring["next"]]
        ring["next"] = (ring["next"] + 1) % self.RING
        return s

    def run(self, local_search: Callable, merge: Callable, q: torch.Tensor, k: int, group) -> Tuple[torch.Tensor, torch.Tensor]:
        world = dist.get_world_size(group)
        B = q.shape[0] if q.dim() == 2 else 1
        cur = torch.cuda.current_stream(self.device)
        s = self.slot(B, k, world)
        if s["used"]:
            self.scan.wait_event(s["done"])             # the merge that last read this slot
        ready = s["ready"]
        ready.record(cur)                               # q is valid, earlier users of recycled memory are done
        self.scan.wait_event(ready)
        with torch.cuda.stream(self.scan):
            local_search(q, k, (s["D"], s["I"]))
            work = dist.all_gather_into_tensor(s["g"], s["buf"], group=group, async_op=True)
        q.record_stream(self.scan)
        self.merge.wait_event(ready)
        with torch.cuda.stream(self.merge):
            work.wait()                                 # merge stream waits for the collective
            n = B * k
            g = s["g"].view(world, -1)
            Ig = g[:, :n].unflatten(1, (B, k))
            Dg = g[:, n:].view(torch.float32)[:, :n].unflatten(1, (B, k))
            D, I = merge(Dg, Ig)
            s["done"].record(self.merge)
            s["used"] = True
        D.record_stream(cur)
        I.record_stream(cur)
        cur.wait_event(s["done"])
        return D, I


MAX_GLOBAL_IDS = 0xFFFFFFFF  # merge keys carry 32-bit ids (csrc/common.h): global ids must stay below 2^32-1


def _comm_device(group=None):
    """Device the collectives of this process group move tensors on (RCCL: the current GPU; gloo: CPU)."""
    backend = dist.get_backend(group) if dist.is_initialized() else "gloo"
    return torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")


class _ShardedSearch:
    """The search half both sharded classes share: local top-k -> ONE all-gather -> merge."""

    def _init_search(self, group, local, merge, real: bool) -> None:
        self.group = group
        self.local = local
        self._merge = merge
        self._real = real  # the HIP index (not a test double): queries must be device tensors on its GPU
        self._pipe = None
        self._force_pipeline = False  # tests: run the gather/merge path even with one rank
        self._pipelined = os.environ.get("SSMI_SHARD_PIPELINE", "0") == "1"

    def _check_query(self, q) -> None:
        if not self._real:
            return
        if not (isinstance(q, torch.Tensor) and q.is_cuda and q.device.index == self.local.device):
            raise ValueError("sharded search: q must be a torch tensor on this rank's GPU (cuda:%d): the per-rank "
                             "results are all-gathered over RCCL from device buffers" % self.local.device)

    def _search(self, q, k: int, id_offset: int):
        if self.world == 1 and not self._force_pipeline:
            return self.local.search(q, k, id_offset=id_offset)
        self._check_query(q)
        if q.is_cuda and self._pipelined:
            if self._pipe is None:
                self._pipe = _Pipeline(q.device)
            return self._pipe.run(lambda qq, kk, out: self.local.search(qq, kk, id_offset=id_offset, out=out),
                                  self._merge, q, k, self.group)
        B = q.shape[0] if q.dim() == 2 else 1
        buf, D, I = packed_result_buffers(B, k, q.device)
        self.local.search(q, k, id_offset=id_offset, out=(D, I))
        Dg, Ig = gather_packed(buf, B, k, self.group)
        return self._merge(Dg, Ig)


class ShardedFlatIP(_ShardedSearch):
    """Exact inner-product search over a database row-sharded across ranks."""

    def __init__(self, d: int, storage: str = "f32", group=None, device: Optional[int] = None,
                 _local_index=None, _merge: Optional[Callable] = None):
        self.d = d
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        real = _local_index is None
        if real:
            from .faiss_compat import IndexFlatIP, topk_merge
            _local_index = IndexFlatIP(d, storage=storage, device=device)
            _merge = topk_merge
        self._init_search(group, _local_index, _merge, real)
        self.row_offset = 0
        self.ntotal = 0

    def add_local(self, x) -> None:
        self.local.add(x)

    def add_local_random_unit(self, n: int, seed: int) -> None:
        self.local.add_random_unit(n, seed)

    def finalize(self, device="cpu") -> None:
        """Agree on row offsets after the ranks have added their blocks."""
        if self.world > 1:
            self.row_offset, self.ntotal = exchange_offsets(self.local.ntotal, self.group, device)
        else:
            self.row_offset, self.ntotal = 0, self.local.ntotal
        if self.ntotal >= MAX_GLOBAL_IDS:
            raise RuntimeError("sharded index holds %d rows: global ids must stay below 2^32-1" % self.ntotal)

    def search(self, q, k: int):
        """q: (B,d) tensor, identical on every rank.  Returns the global (D, I) on every rank."""
        return self._search(q, k, self.row_offset)


class ShardedIVFFlat(_ShardedSearch):
    """IVF-Flat sharded BY LIST across ranks: every rank holds the same centroids, list l lives
    wholly on rank l % world under global ids.

    add_local(x): every rank passes ITS OWN vectors (a slice of the incoming stream, or rows it generated):
    the rank assigns only those to lists, the ranks exchange (vector, list, id) triples with one all-to-all
    so that each list's members reach the owning rank, and the owner appends them.  Each vector is assigned
    once, on one rank, and need not exist anywhere else.  add(x) is the reference-shaped form (one add stream,
    core/index_manager.py:319): every rank passes the same x and takes a contiguous 1/world slice of it.
    A search probes the same nprobe lists everywhere, each rank scans the probed lists it owns (the others
    are empty locally), and the per-rank top-k are all-gathered and merged like the Flat case."""

    def __init__(self, d: int, nlist: int, storage: str = "f32", group=None, device: Optional[int] = None,
                 rank: Optional[int] = None, world: Optional[int] = None, _local_index=None,
                 _merge: Optional[Callable] = None):
        self.d, self.nlist = d, nlist
        self.world = world if world is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.rank = rank if rank is not None else (dist.get_rank(group) if dist.is_initialized() else 0)
        real = _local_index is None
        if real:
            from .faiss_compat import IndexFlatIP, IndexIVFFlat, METRIC_INNER_PRODUCT, topk_merge
            _local_index = IndexIVFFlat(IndexFlatIP(d, storage=storage, device=device), d, nlist, METRIC_INNER_PRODUCT,
                                        storage=storage, device=device)
            _merge = topk_merge
        self._init_search(group, _local_index, _merge, real)
        self.ntotal = 0

    @property
    def nprobe(self) -> int:
        return self.local.nprobe

    @nprobe.setter
    def nprobe(self, v: int) -> None:
        self.local.nprobe = v

    @property
    def is_trained(self) -> bool:
        return self.local.is_trained

    def _collective(self) -> bool:
        # _force_pipeline: a one-rank group rehearses the collectives of the routed add as well (tests, bench)
        return dist.is_initialized() and (self.world > 1 or self._force_pipeline)

    def train(self, x) -> None:
        """Rank 0 runs k-means; the centroids are broadcast so that every rank quantizes identically."""
        if not self._collective():
            self.local.train(x)
            return
        dev = _comm_device(self.group)
        c = torch.empty((self.nlist, self.d), dtype=torch.float32, device=dev)
        if self.rank == 0:
            self.local.train(x)
            c.copy_(torch.from_numpy(self.local.get_centroids()))
        dist.broadcast(c, src=0, group=self.group)
        if self.rank != 0:
            self.local.set_centroids(c.cpu().numpy())

    def add(self, x) -> None:
        """Every rank passes the same x (ids ntotal .. ntotal+n-1 in x's order)."""
        n = int(x.shape[0])
        if not self._collective():
            # one process standing for one rank of a logical sharding (tests): keep the own lists' members
            self.local.add_shard(x, self.ntotal, self.rank, self.world)
            self.ntotal += n
            return
        b = shard_bounds(n, self.world)
        self.add_local(x[b[self.rank]:b[self.rank + 1]])

    def add_local(self, x, ids=None) -> None:
        """x: (n_r, d) vectors of THIS rank (numpy, CPU tensor or device tensor).  Global ids continue from
        ntotal in rank order (rank r's vectors follow those of ranks < r) unless the caller names them
        (`ids`, int64 (n_r,), unique across ranks: e.g. row offset + local row of a row-sharded twin)."""
        if ids is not None:
            ids = torch.as_tensor(np.asarray(ids.cpu() if isinstance(ids, torch.Tensor) else ids), dtype=torch.int64).reshape(-1)
            if ids.shape[0] != int(x.shape[0]):
                raise ValueError("add_local: one id per vector")
        if not self._collective():
            n = int(x.shape[0])
            own = np.arange(self.ntotal, self.ntotal + n, dtype=np.int64) if ids is None else ids.numpy()
            self.local.add_preassigned(x, self.local.assign(x), own)
            self.ntotal += n
            return
        world, rank, group = self.world, self.rank, self.group
        dev = _comm_device(group)
        if not isinstance(x, torch.Tensor):
            x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        n = int(x.shape[0])
        labels = torch.from_numpy(self.local.assign(x).astype(np.int64)) if n else torch.zeros((0,), dtype=torch.int64)
        # id base of this rank and the new global total
        counts = torch.zeros((world,), dtype=torch.int64, device=dev)
        mine = torch.tensor([n], dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(counts, mine, group=group)
        counts = counts.cpu()
        id0 = self.ntotal + int(counts[:rank].sum())
        total = int(counts.sum())
        if self.ntotal + total >= MAX_GLOBAL_IDS:
            raise RuntimeError("sharded index would hold %d rows: global ids must stay below 2^32-1" % (self.ntotal + total))
        # route: sort this rank's vectors by owner (list % world), exchange the per-owner counts, then the payload
        owner = labels % world
        order = torch.argsort(owner, stable=True)
        send_counts = torch.bincount(owner, minlength=world)
        recv_counts = torch.empty_like(send_counts)
        sc, rc = send_counts.to(dev), recv_counts.to(dev)
        dist.all_to_all_single(rc, sc, group=group)
        send_counts, recv_counts = send_counts.tolist(), rc.cpu().tolist()
        own = torch.arange(id0, id0 + n, dtype=torch.int64) if ids is None else ids
        meta = torch.stack([labels, own], dim=1)[order].contiguous().to(dev)
        xs = x.to(dev)
        if xs.dtype not in (torch.float32, torch.float16, torch.bfloat16):
            xs = xs.float()
        storage = getattr(self.local, "storage", "f32")
        if storage == "f16":
            xs = xs.half()  # what the owner stores anyway (same round-to-nearest-even): halves the bytes on the links
        elif storage == "bf16":
            xs = xs.bfloat16()
        xs = xs[order.to(xs.device)].contiguous()
        nrecv = int(sum(recv_counts))
        meta_r = torch.empty((nrecv, 2), dtype=torch.int64, device=dev)
        x_r = torch.empty((nrecv, self.d), dtype=xs.dtype, device=dev)
        dist.all_to_all_single(meta_r, meta, recv_counts, send_counts, group=group)
        dist.all_to_all_single(x_r, xs, recv_counts, send_counts, group=group)
        if nrecv:
            meta_r = meta_r.cpu().numpy()
            assert (meta_r[:, 0] % world == rank).all(), "routed add: received a member of a list this rank does not own"
            self.local.add_preassigned(x_r, meta_r[:, 0].astype(np.int32), meta_r[:, 1])
        self.ntotal += total

    def search(self, q, k: int):
        if not self._collective() and not self._force_pipeline:
            return self.local.search(q, k)
        return self._search(q, k, 0)
