"""Diagnostics: the DiT forward launched again and again on fixed inputs -- bit-reproducible beside another process on the same GPU?
(round 6: the VAE halo convolution was not -- a ring stage refilled while a fragment read of it was still pending.)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if "--lib" in sys.argv:      # another build of the library (file name under wan2gp_amd/)
    from wan2gp_amd import lib as _lib
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), sys.argv[sys.argv.index("--lib") + 1])
from oracle import wan_oracle as O
from wan2gp_amd.model import WanModelHIP
tag, iters = sys.argv[1], int(sys.argv[2])
mode = sys.argv[3] if len(sys.argv) > 3 else "product"
from wan2gp_amd import lib as L
if mode == "walk": L.load().wan_attention_debug_no_persist(2)          # text cross-attention on round 4's persistent walk instead of attention_xkv.hip
if mode == "oneblock": L.load().wan_attention_debug_no_persist(1)      # short KV as ordinary one-block launches
if mode == "rows256": L.load().wan_gemm_debug_force_tile_rows(-1)       # round 5's GEMM dispatch (256-row tiles / gemm32)
if mode == "no16s": L.load().wan_gemm_debug_force16s(-1)
if mode.startswith("force"): L.load().wan_gemm_debug_force_tile_rows(int(mode[5:8]))   # gemm256m at that tile height wherever it fits
if mode.startswith("L1"): os.environ["_L1"] = "1"
shapes = (("small", (5, 32, 48)),) if mode != "product" else (("small", (5, 32, 48)), ("t2v_1.3B", (3, 24, 40)))
if mode.startswith("bigL"): os.environ["_BL"] = mode[4:5]; shapes = (("t2v_1.3B", (9, 60, 104)),)
if mode == "big": shapes = (("t2v_1.3B", (9, 60, 104)),)
if mode == "big14": shapes = (("t2v_14B", (9, 60, 104)),); os.environ["_BL"] = "2"        # the 14B widths (d = 5120: the persistent row kernels), two layers          # 14,040 tokens per stream: every Linear a many-tile problem (gemm256m at 256 rows)
for name, fhw in shapes:
    cfg = O.make_config(name)
    if name == "t2v_1.3B":
        cfg.num_layers = 2
    if os.environ.get("_L1"):
        cfg.num_layers = 1
    if os.environ.get("_BL"):
        cfg.num_layers = int(os.environ["_BL"])
    W = O.synth_weights(cfg, seed=7)
    m = WanModelHIP(model_type=cfg.model_type, dim=cfg.dim, ffn_dim=cfg.ffn_dim, num_heads=cfg.num_heads, num_layers=cfg.num_layers,
                    in_dim=cfg.in_dim, out_dim=cfg.out_dim).load_state_dict(W)
    if "notc" in mode: m.text_cache = False
    lat, ctx, ctx_null, y = O.synth_inputs(cfg, *fhw, seed=3)
    t = torch.tensor([500])
    x, c0, c1 = lat.cuda(), ctx.cuda(), ctx_null.cuda()
    ref = [o.clone() for o in m([x, x], t=t, context=[c0, c1])]
    bad = 0
    for it in range(iters):
        outs = m([x, x], t=t, context=[c0, c1])
        if not all(torch.equal(a, b) for a, b in zip(outs, ref)):
            bad += 1
            if bad <= 4:
                info = []
                for si, (a, b) in enumerate(zip(outs, ref)):
                    d = (a - b).abs()
                    nz = torch.nonzero(d[0].amax(dim=0) > 0)            # [frames, h, w] positions that differ in any channel
                    if len(nz): info.append((si, int(len(nz)), nz.min(dim=0).values.tolist(), nz.max(dim=0).values.tolist(), round(d.max().item(), 5)))
                print(tag, name, "iteration", it, "stream / differing positions / min (f,h,w) / max (f,h,w) / maxdiff:", info, flush=True)
    print(mode, tag, name, fhw, "L =", fhw[0] * fhw[1] * fhw[2] // 4, ": %d of %d forwards differ" % (bad, iters), flush=True)
