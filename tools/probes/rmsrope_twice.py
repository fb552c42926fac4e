"""Diagnostics: when wan_rmsnorm_rope (in place) leaves its first call's bits beside a neighbour, are the wrong rows what the kernel makes of
rows it has ALREADY processed (a row normalised and rotated twice)?"""
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
        # what ARE the wrong chunks?  the source's (never processed / overwritten by the copy), another position's rotation (pair norms kept), or neither;
        # and do they come as a lane's three chunks (c = lane + 64 i: a lane-level cause) or singly
        s2 = q0.view(-1, d)
        raw_eq = (o2[rows].view(len(rows), d // 8, 8) == s2[rows].view(len(rows), d // 8, 8)).all(dim=2)
        def pairnorm(t): return (t.float().view(len(rows), d // 8, 4, 2) ** 2).sum(dim=3)
        pn_o, pn_r = pairnorm(o2[rows]), pairnorm(r1[rows])
        norm_kept = ((pn_o - pn_r).abs() <= 0.02 * pn_r + 1e-3).all(dim=2)
        lanes = ch.view(len(rows), d // 512, 64)                      # [row][i][lane]
        per_lane = lanes.sum(dim=1)                                   # wrong chunks per (row, lane)
        hist = [int((per_lane == n).sum()) for n in range(1, d // 512 + 1)]
        isel = [int(lanes[:, i].sum()) for i in range(d // 512)]
        dmax = (o2[rows].float() - r1[rows].float()).abs().max().item()
        print(tag, "   wrong chunks equal to the SOURCE chunk:", int((ch & raw_eq).sum()), "; with the reference's pair norms (another rotation):", int((ch & norm_kept).sum()),
              "; (row, lane) with 1 / 2 / 3 wrong chunks:", hist, "; wrong chunks by chunk index i:", isel, "; lanes hit:", sorted(set(torch.nonzero(per_lane)[:, 1].tolist()))[:24],
              "; max |diff|", dmax, flush=True)
        # is a wrong chunk some OTHER chunk of the first launch's result, of the twice-processed result or of the source (a store or a load that went elsewhere)?
        r1c, r2c, s2c = r1.view(-1, d // 8, 8), r2.view(-1, d // 8, 8), s2.view(-1, d // 8, 8)
        wr = torch.nonzero(ch)[:48]
        found = {"ref1 other row, same chunk": 0, "ref1 same row, other chunk": 0, "ref2 other row": 0, "source other row": 0, "all zero": 0}
        for ri, c in wr.tolist():
            v = o2[rows[ri]].view(d // 8, 8)[c]
            found["ref1 other row, same chunk"] += int((r1c[:, c] == v).all(dim=1).any())
            found["ref1 same row, other chunk"] += int((r1c[rows[ri]] == v).all(dim=1).any())
            found["ref2 other row"] += int((r2c[:, c] == v).all(dim=1).any())
            found["source other row"] += int((s2c[:, c] == v).all(dim=1).any())
            found["all zero"] += int((v == 0).all())
        print(tag, "   of the first", len(wr), "wrong chunks:", found, flush=True)
        if bad <= 3 and "--dump" in sys.argv:
            torch.save({"rows": rows.cpu(), "out": o2[rows].cpu(), "ref1": r1[rows].cpu(), "ref2": r2[rows].cpu(), "src": s2[rows].cpu(), "wq": wq.cpu(), "L": Lt,
                        "cos": cos.cpu(), "sin": sin.cpu(), "q_scale": ops.attention_qscale()}, sys.argv[sys.argv.index("--dump") + 1] + "_%d.pt" % bad)
        print(tag, "iteration", it, ": wrong rows", len(rows), "of them equal to the row processed TWICE:", whole2, "; wrong chunks", int(ch.sum()),
              "of them equal to the twice-processed row's chunk:", int((ch & ch2).sum()), "; rows mod 4:", sorted(set((rows % 4).tolist())),
              "first rows", rows[:8].tolist(), flush=True)
print(tag, ": %d of %d launches differ" % (bad, it + 1), flush=True)
