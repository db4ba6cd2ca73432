import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import semantic_search_amd as ssa
n, d, nq, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
storage = sys.argv[5] if len(sys.argv) > 5 else "f16"
ix = ssa.IndexFlatIP(d, storage=storage); ix.add_random_unit(n, 1234)
g = torch.Generator().manual_seed(1); Q = torch.randn((nq, d), generator=g); Q = (Q / Q.norm(dim=1, keepdim=True)).cuda()
for it in range(3):
    D, I = ix.search(Q, k); torch.cuda.synchronize()
    t = ix.last_timing()
    el = 4 if storage == "f32" else 2
    print(f"path={t['last_path']} grid={t['last_grid']} scan={t['last_scan_ms']:.3f} ms merge={t['last_merge_ms']:.3f} ms  "
          f"{2*nq*n*d/t['last_scan_ms']/1e9:.1f} TFLOP/s  {n*d*el/t['last_scan_ms']/1e6:.0f} GB/s  qps={nq/(t['last_search_ms']*1e-3):.0f}")
