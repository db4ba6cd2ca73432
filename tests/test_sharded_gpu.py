"""The three-stream gather/merge pipeline of the sharded search (semantic_search_amd/sharded.py)
on one GPU: a one-rank RCCL group exercises the same streams, events and buffer ring that the
multi-GPU bench uses; results must equal the direct local search, step by step, when many searches
are enqueued back to back without synchronising."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

from conftest import unit_rows

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def one_rank_group():
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield
    dist.destroy_process_group()


@pytest.mark.parametrize("pipelined", [False, True])
@pytest.mark.parametrize("B,k", [(1, 10), (5, 100), (64, 10)])
def test_gathered_search_equals_direct(ssa, one_rank_group, B, k, pipelined):
    from semantic_search_amd.sharded import ShardedFlatIP
    n, d, steps = 20000, 512, 24
    X = unit_rows(n, d, 5)
    idx = ShardedFlatIP(d, storage="f32", device=0)
    idx.add_local(X)
    idx.finalize(device=torch.device("cuda", 0))
    idx.row_offset = 1000  # ids must carry the shard offset through gather and merge
    idx._force_pipeline = True
    idx._pipelined = pipelined   # default one-stream sequence and the opt-in three-stream pipeline
    Q = torch.from_numpy(unit_rows(steps * B, d, 6)).cuda().view(steps, B, d)
    outs = [idx.search(Q[i], k) for i in range(steps)]  # enqueued back to back, ring of 4 slots reused
    torch.cuda.synchronize()
    for i in range(steps):
        Dr, Ir = idx.local.search(Q[i], k, id_offset=1000)
        torch.cuda.synchronize()
        assert torch.equal(outs[i][1], Ir), f"step {i}: ids differ"
        assert torch.equal(outs[i][0], Dr), f"step {i}: scores differ"
