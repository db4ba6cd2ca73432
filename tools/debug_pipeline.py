"""Host/GPU cost of the sharded three-stream pipeline on ONE rank (RCCL group of size 1): a 125k x 4096 fp32
shard = what each of 8 GPUs scans in bench.py --gpus 8."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from semantic_search_amd.sharded import ShardedFlatIP
n = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
idx = ShardedFlatIP(4096, storage="f32", device=0)
idx.add_local_random_unit(n, seed=1)
idx.finalize(device=torch.device("cuda", 0))
Q = torch.randn((8, 1, 4096), device="cuda"); Q /= Q.norm(dim=-1, keepdim=True)
for mode in ("direct", "pipeline", "serial"):
    idx._force_pipeline = mode != "direct"
    idx._pipelined = mode == "pipeline"
    for i in range(20): idx.search(Q[i % 8], 10)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(500): idx.search(Q[i % 8], 10)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{mode}: {500 / dt:.0f} QPS, {dt / 500 * 1e6:.0f} us/step (host enqueue {t_host / 500 * 1e6:.0f} us/step), "
          f"scan kernel {idx.local.last_timing()['last_scan_ms'] * 1e3:.0f} us")
dist.destroy_process_group()
