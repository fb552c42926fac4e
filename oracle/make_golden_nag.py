"""TEST INFRASTRUCTURE ONLY -- tests/golden/nag.npz from the REFERENCE's own WanModel, executed unmodified on CPU through
oracle/ref_shim.py with normalized attention guidance switched on the way WanAny2V.generate does it
(models/wan/any2video.py:607-608: offload.shared_state["_nag_scale" | "_nag_tau" | "_nag_alpha"], context = cat([context,
context_null])).  The branch under test is text_cross_attention's (models/wan/modules/model.py:245-293).

    cross_*   one WanT2VCrossAttention module of the reference on a seeded hidden state (dim 256, 2 heads): the NAG result,
              with the share of rows the norm clip (scale > tau) rescaled stored beside it -- both branches must be exercised
    fwd_*     full forwards: `small` (dim 512, 4 heads, 3 layers, L = 105) as a CFG pair -- NAG context [2,512,4096] on the cond
              stream, a plain [1,512,4096] on the uncond stream -- and `tiny_i2v21` (CLIP tokens beside a batch-2 text context)

Run in the build container:   python oracle/make_golden_nag.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim, wan_oracle as O  # noqa: E402
from oracle.make_golden import build_ref_model, ref_forward, f32  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "nag.npz")
NAG = (11.0, 2.5, 0.25)          # NAG_scale, NAG_tau, NAG_alpha: all exact in bf16
NAG_MILD = (1.5, 3.5, 0.5)       # a scale for which (almost) no row is clipped


def cross_inputs(cfg, L=96, seed=31):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(1, L, cfg.dim, generator=g).to(torch.bfloat16)
    ctx = (0.5 * torch.randn(2, cfg.text_len, cfg.dim, generator=g)).to(torch.bfloat16)
    ctx[0, 77:] = 0
    ctx[1, 8:] = 0
    return x, ctx


def main():
    ns = ref_shim.load()
    out = {}
    cfg = O.make_config("tiny")
    W = O.synth_weights(cfg)
    m = build_ref_model(ns, cfg, W, torch.bfloat16)
    x, ctx = cross_inputs(cfg)
    for tag, nag in (("strong", NAG), ("mild", NAG_MILD)):
        ns.offload.shared_state.update({"_nag_scale": nag[0], "_nag_tau": nag[1], "_nag_alpha": nag[2]})
        with torch.no_grad():
            o = m.blocks[1].cross_attn([x.clone()], ctx.clone(), None)
        out[f"cross_{tag}"] = f32(o)
        out[f"cross_{tag}_nag"] = np.array(nag)
        # how many rows the clip touched, recomputed from the oracle's own intermediate (diagnostic for the fixture only)
        p = "blocks.1.cross_attn."
        b, n, d = 1, cfg.num_heads, cfg.head_dim
        q = O.rms_norm(O._linear(x, W, p + "q"), W[p + "norm_q.weight"], cfg.eps).view(b, -1, n, d)
        k = O.rms_norm(O._linear(ctx, W, p + "k"), W[p + "norm_k.weight"], cfg.eps).view(2, -1, n, d)
        v = O._linear(ctx, W, p + "v").view(2, -1, n, d)
        xp, xn = O.attention(q, k[:1], v[:1]).flatten(2, 3), O.attention(q, k[1:], v[1:]).flatten(2, 3)
        gd = xn * (1 - nag[0]) + nag[0] * xp
        ratio = gd.float().abs().sum(-1) / xp.float().abs().sum(-1)
        out[f"cross_{tag}_clipped_rows"] = np.array([int((ratio > nag[1]).sum()), ratio.numel()])
        print(f"cross_{tag}: rows clipped {int((ratio > nag[1]).sum())} of {ratio.numel()}, ratio {ratio.min():.2f}..{ratio.max():.2f}")

    ns.offload.shared_state.update({"_nag_scale": NAG[0], "_nag_tau": NAG[1], "_nag_alpha": NAG[2]})
    for name, (f, h, w), tval in (("small", (3, 10, 14), 412), ("tiny_i2v21", (2, 8, 8), 731)):
        cfg = O.make_config(name)
        W = O.synth_weights(cfg)
        m = build_ref_model(ns, cfg, W, torch.bfloat16)
        lat, c, cn, y = O.synth_inputs(cfg, f, h, w)
        clip = O.synth_clip_fea() if cfg.model_type == "i2v" else None
        t = torch.tensor([tval], dtype=torch.int64)
        c2 = torch.cat([c, cn], dim=0)                                   # any2video.py:608
        r = ref_forward(ns, m, [lat, lat], t, [c2, cn], y=y, clip_fea=clip)
        out[f"fwd_{name}_cond"], out[f"fwd_{name}_uncond"] = f32(r[0]), f32(r[1])
        out[f"fwd_{name}_shape"], out[f"fwd_{name}_t"] = np.array([f, h, w]), np.array([tval])
    # skip-layer guidance (any2video.py:1502, model.py:2025-2028): blocks listed in perturbation_layers run for the first stream only
    ns.offload.shared_state.update({"_nag_scale": 0})
    cfg = O.make_config("small")
    W = O.synth_weights(cfg)
    m = build_ref_model(ns, cfg, W, torch.bfloat16)
    lat, c, cn, _ = O.synth_inputs(cfg, 3, 10, 14)
    freqs = ns.P.get_rotary_pos_embed((3, 10, 14))
    import types
    with torch.no_grad():
        r = m([lat.clone(), lat.clone()], t=torch.tensor([412]), context=[c.clone(), cn.clone()], freqs=freqs,
              pipeline=types.SimpleNamespace(_interrupt=False), perturbation_layers=[1])
    out["slg_small_cond"], out["slg_small_uncond"] = f32(r[0]), f32(r[1])
    out["fwd_nag"] = np.array(NAG)
    ns.offload.shared_state.update({"_nag_scale": 0})
    np.savez_compressed(OUT, **out)
    print("nag.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
