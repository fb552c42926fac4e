"""Diagnostics: wan_rmsnorm_rope launched again and again in place on fresh copies of fixed rows, beside another process: what differs?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if "--lib" in sys.argv:
    from wan2gp_amd import lib as _lib
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), sys.argv[sys.argv.index("--lib") + 1])
from wan2gp_amd import ops
from oracle import wan_oracle as O
tag, iters = sys.argv[1], int(sys.argv[2])
BF = torch.bfloat16
g = torch.Generator().manual_seed(6)
d, grid = 1536, (9, 30, 52)
Lt = grid[0] * grid[1] * grid[2]
q0 = torch.randn(2, Lt, d, generator=g).to(BF).cuda(); k0 = torch.randn(2, Lt, d, generator=g).to(BF).cuda()
wq = (1 + 0.1 * torch.randn(d, generator=g)).to(BF).cuda(); wk = (1 + 0.1 * torch.randn(d, generator=g)).to(BF).cuda()
cos, sin = [t.cuda() for t in O.rope_tables(grid)]
def run(rope, with_k, sync):
    q, k = q0.clone(), (k0.clone() if with_k else None)
    if sync: torch.cuda.synchronize()
    ops.rmsnorm_rope_(q, k if with_k else None, wq, wk, freqs=(cos, sin) if rope else None, L=Lt, q_scale=ops.attention_qscale())
    return q if not with_k else torch.cat([q.flatten(), k.flatten()])
for rope, with_k, sync in ((True, True, False), (True, True, True), (False, True, False), (True, False, False)):
    ref = run(rope, with_k, sync).clone()
    bad, shown = 0, 0
    for it in range(iters):
        o = run(rope, with_k, sync)
        ne = o != ref
        n = int(ne.sum())
        if n:
            bad += 1
            if shown < 3:
                shown += 1
                idx = torch.nonzero(ne.flatten()).flatten()
                rows = sorted(set((idx // d).tolist()))
                print(tag, "rope", rope, "k", with_k, "sync", sync, "iteration", it, "differing elements", n, "in rows", rows[:6], "... maxdiff",
                      (o.float() - ref.float()).abs().max().item(), "first cols", (idx[:8] % d).tolist(), flush=True)
    print(tag, "rope", rope, "with k", with_k, "sync before launch", sync, ": %d of %d launches differ" % (bad, iters), flush=True)
