"""Row-sharded Flat search across the GPUs of one node: one process per GPU.

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


class ShardedFlatIP:
    """Exact inner-product search over a database row-sharded across ranks."""

    def __init__(self, d: int, storage: str = "f32", group=None, device: Optional[int] = None,
                 _local_index=None, _merge: Optional[Callable] = None):
        self.d = d
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if _local_index is None:
            from .faiss_compat import IndexFlatIP, topk_merge
            _local_index = IndexFlatIP(d, storage=storage, device=device)
            _merge = topk_merge
        self.local = _local_index
        self._merge = _merge
        self.row_offset = 0
        self.ntotal = 0
        self._pipe = None
        self._force_pipeline = False  # tests: run the gather/merge path even with one rank
        self._pipelined = os.environ.get("SSMI_SHARD_PIPELINE", "0") == "1"

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

    def search(self, q, k: int):
        """q: (B,d) tensor, identical on every rank.  Returns the global (D, I) on every rank."""
        if self.world == 1 and not self._force_pipeline:
            return self.local.search(q, k, id_offset=self.row_offset)
        if q.is_cuda and self._pipelined:
            if self._pipe is None:
                self._pipe = _Pipeline(q.device)
            return self._pipe.run(lambda qq, kk, out: self.local.search(qq, kk, id_offset=self.row_offset, out=out),
                                  self._merge, q, k, self.group)
        B = q.shape[0] if q.dim() == 2 else 1
        buf, D, I = packed_result_buffers(B, k, q.device)
        self.local.search(q, k, id_offset=self.row_offset, out=(D, I))
        Dg, Ig = gather_packed(buf, B, k, self.group)
        return self._merge(Dg, Ig)


class ShardedIVFFlat:
    """IVF-Flat sharded BY LIST across ranks: every rank holds the same centroids, list l lives
    wholly on rank l % world.  Every rank passes the same vectors to add() (as the reference's
    single add stream would) and keeps its own lists' members under global ids; a search probes
    the same nprobe lists everywhere, each rank scans the probed lists it owns (the others are
    empty locally), and the per-rank top-k are all-gathered and merged like the Flat case."""

    def __init__(self, d: int, nlist: int, storage: str = "f32", group=None, device: Optional[int] = None,
                 rank: Optional[int] = None, world: Optional[int] = None):
        from .faiss_compat import IndexFlatIP, IndexIVFFlat, METRIC_INNER_PRODUCT, topk_merge
        self.d, self.nlist, self.group = d, nlist, group
        self.world = world if world is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.rank = rank if rank is not None else (dist.get_rank(group) if dist.is_initialized() else 0)
        self.local = IndexIVFFlat(IndexFlatIP(d, storage=storage, device=device), d, nlist, METRIC_INNER_PRODUCT,
                                  storage=storage, device=device)
        self._merge = topk_merge
        self.ntotal = 0
        self._pipe = None

    @property
    def nprobe(self) -> int:
        return self.local.nprobe

    @nprobe.setter
    def nprobe(self, v: int) -> None:
        self.local.nprobe = v

    @property
    def is_trained(self) -> bool:
        return self.local.is_trained

    def train(self, x) -> None:
        """Rank 0 runs k-means; the centroids are broadcast so that every rank quantizes identically."""
        if self.world == 1 or not dist.is_initialized():
            self.local.train(x)
            return
        dev = torch.device("cuda", self.local.device)
        c = torch.empty((self.nlist, self.d), dtype=torch.float32, device=dev)
        if self.rank == 0:
            self.local.train(x)
            c.copy_(torch.from_numpy(self.local.get_centroids()))
        dist.broadcast(c, src=0, group=self.group)
        if self.rank != 0:
            self.local.set_centroids(c.cpu().numpy())

    def add(self, x) -> None:
        n = int(x.shape[0])
        self.local.add_shard(x, self.ntotal, self.rank, self.world)
        self.ntotal += n

    def search(self, q, k: int):
        if self.world == 1 or not dist.is_initialized():
            return self.local.search(q, k)
        if self._pipe is None:
            self._pipe = _Pipeline(q.device)
        return self._pipe.run(lambda qq, kk, out: self.local.search(qq, kk, out=out), self._merge, q, k, self.group)
