"""Diagnostics: when wan_rmsnorm_rope (in place) leaves its first call's bits beside a neighbour, are the wrong rows what the kernel makes of
rows it has ALREADY processed (a row normalised and rotated twice)?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wan2gp_amd import ops
from oracle import wan_oracle as O
tag, iters = sys.argv[1], int(sys.argv[2])
BF = torch.bfloat16
g = torch.Generator().manual_seed(6)
d, grid = 1536, (9, 30, 52)
Lt = grid[0] * grid[1] * grid[2]
q0 = torch.randn(2, Lt, d, generator=g).to(BF).cuda()
wq = (1 + 0.1 * torch.randn(d, generator=g)).to(BF).cuda()
cos, sin = [t.cuda() for t in O.rope_tables(grid)]
def f(src):
    q = src.clone()
    ops.rmsnorm_rope_(q, None, wq, wq, freqs=(cos, sin), L=Lt, q_scale=ops.attention_qscale())
    return q
ref1 = f(q0).clone()
ref2 = f(ref1).clone()                    # every row processed twice
import time
bad = 0
t_end = time.time() + iters          # (iters = seconds to keep launching)
it = -1
while time.time() < t_end:
    it += 1
    o = f(q0)
    if torch.equal(o, ref1):
        continue
    bad += 1
    if bad <= 6:
        o2, r1, r2 = o.view(-1, d), ref1.view(-1, d), ref2.view(-1, d)
        rows = torch.nonzero((o2 != r1).any(dim=1)).flatten()
        whole2 = int((o2[rows] == r2[rows]).all(dim=1).sum())
        ch = (o2[rows].view(len(rows), d // 8, 8) != r1[rows].view(len(rows), d // 8, 8)).any(dim=2)           # wrong 16-byte chunks per row
        ch2 = (o2[rows].view(len(rows), d // 8, 8) == r2[rows].view(len(rows), d // 8, 8)).all(dim=2)          # chunks equal to the twice-processed row's
        print(tag, "iteration", it, ": wrong rows", len(rows), "of them equal to the row processed TWICE:", whole2, "; wrong chunks", int(ch.sum()),
              "of them equal to the twice-processed row's chunk:", int((ch & ch2).sum()), "; rows mod 4:", sorted(set((rows % 4).tolist())),
              "first rows", rows[:8].tolist(), flush=True)
print(tag, ": %d of %d launches differ" % (bad, it + 1), flush=True)
