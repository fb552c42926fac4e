"""Diagnostics: WHERE the forward beside another process first leaves the first call's bits -- the residual stream snapshotted at every block
boundary (WanModelHIP.debug_token_stream inside the per-block callback)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if "--lib" in sys.argv:
    from wan2gp_amd import lib as _lib
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), sys.argv[sys.argv.index("--lib") + 1])
from oracle import wan_oracle as O
from wan2gp_amd.model import WanModelHIP
tag, iters = sys.argv[1], int(sys.argv[2])
cfg = O.make_config("t2v_1.3B")
cfg.num_layers = 3
fhw = (9, 60, 104)
W = O.synth_weights(cfg, seed=7)
m = WanModelHIP(model_type=cfg.model_type, dim=cfg.dim, ffn_dim=cfg.ffn_dim, num_heads=cfg.num_heads, num_layers=cfg.num_layers,
                in_dim=cfg.in_dim, out_dim=cfg.out_dim).load_state_dict(W)
lat, ctx, ctx_null, y = O.synth_inputs(cfg, *fhw, seed=3)
t = torch.tensor([500])
x, c0, c1 = lat.cuda(), ctx.cuda(), ctx_null.cuda()
Ltok = fhw[0] * fhw[1] * fhw[2] // 4
def run():
    snaps = []
    def cb(*a):
        snaps.append(m.debug_token_stream(2, Ltok).clone())
    outs = m([x, x], t=t, context=[c0, c1], callback=cb)
    return snaps, [o.clone() for o in outs]
ref_s, ref_o = run()
print(tag, "boundaries per forward:", len(ref_s), flush=True)
bad = 0
for it in range(iters):
    s, o = run()
    first = next((i for i, (a, b) in enumerate(zip(s, ref_s)) if not torch.equal(a, b)), None)
    out_diff = not all(torch.equal(a, b) for a, b in zip(o, ref_o))
    if first is not None or out_diff:
        bad += 1
        if bad <= 8:
            if first is None:
                print(tag, "iteration", it, ": every boundary equal, outputs differ (behind the last boundary: last block / head)", flush=True)
                continue
            d = (s[first].float() - ref_s[first].float()).abs()                    # [2, L, dim]
            rows = torch.nonzero(d.amax(dim=2) > 0)                                # (stream, token)
            cols = torch.nonzero(d.amax(dim=(0, 1)) > 0).flatten()
            rl = rows[:, 1]
            print(tag, "iteration", it, "first differing boundary", first, "(0 = in front of block 0): rows", len(rows), "streams", sorted(set(rows[:, 0].tolist())),
                  "token range", int(rl.min()), int(rl.max()), "distinct 256-row blocks", len(set((rl // 256).tolist())), "cols", len(cols), "range", int(cols.min()), int(cols.max()),
                  "distinct heads", len(set((cols // 128).tolist())), "maxdiff", round(d.max().item(), 4), flush=True)
print(tag, ": %d of %d forwards differ" % (bad, iters), flush=True)
