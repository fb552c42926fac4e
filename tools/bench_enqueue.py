#!/usr/bin/env python
"""How launch-bound is a denoise forward?  Host time to ENQUEUE one joint CFG forward (the call returns when everything is
queued) against its GPU time, at the 1.3B-480p shape (900 launches in ~0.23 s, the launch-densest BASELINE configuration).
If the host stays ahead of the GPU by a wide margin a HIP graph has nothing to remove."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from wan2gp_amd.model import WanModelHIP  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "1.3B-480p"
cfg, (F, H, W), desc = bench.WORKLOADS[name]
m = bench.random_weights(WanModelHIP(**cfg), cfg, seed=1)
g = torch.Generator(device="cuda").manual_seed(0)
lat = torch.randn(1, 16, F, H, W, device="cuda", generator=g)
ctx = torch.randn(1, 512, 4096, device="cuda", generator=g).to(torch.bfloat16)
t = torch.tensor([500])
res = []
for i in range(6):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    h0 = time.perf_counter()
    e0.record()
    m([lat, lat], t=t, context=[ctx, ctx])
    e1.record()
    h1 = time.perf_counter()
    torch.cuda.synchronize()
    if i:
        res.append((1e3 * (h1 - h0), e0.elapsed_time(e1)))
host = sorted(r[0] for r in res)[len(res) // 2]
gpu = sorted(r[1] for r in res)[len(res) // 2]
print(json.dumps({"workload": desc, "host_enqueue_ms": round(host, 2), "gpu_ms": round(gpu, 2),
                  "host_over_gpu": round(host / gpu, 3), "all": [[round(a, 2), round(b, 2)] for a, b in res]}))
