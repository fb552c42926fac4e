"""TEST INFRASTRUCTURE ONLY -- tests/golden/skipcache_tiny.npz from the REFERENCE's own WanModel with `.cache` set
(TeaCache / MagCache: models/wan/modules/model.py:1373-1482 thresholds, :1914-2064 skip logic).
Run in the build container:   python oracle/make_golden_skipcache.py
Tiny t2v config, bf16 plan, 8 sampler-like steps with drifting latents; three scenarios: MagCache joint pass, MagCache
two single passes (x_id 0 / 1), TeaCache joint pass.  Recorded per step: should-calc decisions (from the accumulator
state), outputs, and the chosen thresholds.
`python oracle/make_golden_skipcache.py mixed` writes skipcache_tiny_mixed.npz: the same three scenarios with the reference's
`mixed_precision_transformer` locks (model.py:1330-1371: fp32 time MLP / projection / norm3 -> fp32 residual stream, fp32 `e` for
TeaCache's distance and fp32 previous_residual)."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402
from oracle import wan_oracle as O  # noqa: E402
from oracle.make_golden import build_ref_model  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "skipcache_tiny.npz")
STEPS = 8
MAG_RATIOS = [0.998, 0.997, 0.995, 0.996, 0.97, 0.975, 0.996, 0.995, 0.993, 0.994, 0.96, 0.955, 0.99, 0.991]   # (STEPS-1) cond/uncond pairs
TEA_COEF = [0.04, 0.001]


class Bag:                                   # stands in for wgp.DynamicClass (attribute bag + update)
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def update(self, d):
        self.__dict__.update(d)


def inputs(cfg):
    f, h, w = 2, 8, 8
    lat, ctx, ctx_null, _ = O.synth_inputs(cfg, f, h, w, seed=5)
    g = torch.Generator().manual_seed(17)
    drift = torch.randn(lat.shape, generator=g)
    lats = [lat + 0.15 * i * drift for i in range(STEPS)]
    ts = [torch.tensor(v, dtype=torch.float32) for v in (999.0, 950.0, 880.0, 790.0, 660.0, 500.0, 320.0, 130.0)]
    return lats, ts, ctx, ctx_null


def new_cache(kind):
    c = Bag(cache_type=kind, multiplier=2.0, start_step=1, num_steps=STEPS, skipped_steps=0, previous_residual=None, previous_modulated_input=None)
    if kind == "mag":
        c.update({"magcache_thresh": 0, "magcache_K": 2, "def_mag_ratios": list(MAG_RATIOS)})
    else:
        c.update({"coefficients": list(TEA_COEF), "rel_l1_thresh": 0, "accumulated_rel_l1_distance": 0})
    return c


def main(mixed=False):
    ns = ref_shim.load()
    cfg = O.make_config("tiny")
    W = O.synth_weights(cfg, seed=4321, dtype=torch.bfloat16, mixed=True) if mixed else O.synth_weights(cfg, seed=4321)
    m = build_ref_model(ns, cfg, W, torch.bfloat16, mixed=mixed)
    lats, ts, ctx, ctx_null = inputs(cfg)
    freqs = ns.P.get_rotary_pos_embed(lats[0].shape[2:])
    pipe = types.SimpleNamespace(_interrupt=False)
    out = {}

    def fwd(xs, t, ctxs, step, x_id=0):
        with torch.no_grad():
            return m([x.clone() for x in xs], t=torch.stack([t]), context=[c.clone() for c in ctxs], freqs=freqs, pipeline=pipe,
                     real_step_no=step, current_step_no=step, x_id=x_id)

    # --- MagCache, joint pass ---
    c = m.cache = new_cache("mag")
    c.previous_residual = [None] * 2
    out["mag_thresh"] = np.array([m.compute_magcache_threshold(c.start_step, ts, c.multiplier)])
    out["mag_ratios"] = np.array(c.mag_ratios)
    c.accumulated_err, c.accumulated_steps, c.accumulated_ratio, c.one_for_all = [0.0] * 2, [0] * 2, [1.0] * 2, False
    flags = []
    for i in range(STEPS):
        r = fwd([lats[i], lats[i]], ts[i], [ctx, ctx_null], i)
        flags.append([int(c.accumulated_steps[k] == 0) for k in range(2)])       # reset to 0 <=> computed at this step
        out[f"magj_{i}_0"], out[f"magj_{i}_1"] = r[0].float().numpy(), r[1].float().numpy()
    out["magj_flags"] = np.array(flags); out["magj_skipped"] = np.array([c.skipped_steps])
    # --- MagCache, two single passes per step ---
    c = m.cache = new_cache("mag")
    c.previous_residual = [None] * 2
    m.compute_magcache_threshold(c.start_step, ts, c.multiplier)
    c.accumulated_err, c.accumulated_steps, c.accumulated_ratio, c.one_for_all = [0.0] * 2, [0] * 2, [1.0] * 2, False
    flags = []
    for i in range(STEPS):
        row = []
        for x_id, cc in enumerate((ctx, ctx_null)):
            r = fwd([lats[i]], ts[i], [cc], i, x_id)
            row.append(int(c.accumulated_steps[x_id] == 0))
            out[f"mags_{i}_{x_id}"] = r[0].float().numpy()
        flags.append(row)
    out["mags_flags"] = np.array(flags)
    # --- TeaCache, joint pass ---
    c = m.cache = new_cache("tea")
    c.previous_residual = [None] * 2
    out["tea_thresh"] = np.array([m.compute_teacache_threshold(c.start_step, ts, c.multiplier)])
    flags = []
    for i in range(STEPS):
        r = fwd([lats[i], lats[i]], ts[i], [ctx, ctx_null], i)
        flags.append(int(c.should_calc))
        out[f"teaj_{i}_0"], out[f"teaj_{i}_1"] = r[0].float().numpy(), r[1].float().numpy()
    out["teaj_flags"] = np.array(flags); out["teaj_skipped"] = np.array([c.skipped_steps])
    m.cache = None
    dst = OUT.replace(".npz", "_mixed.npz") if mixed else OUT
    np.savez_compressed(dst, **out)
    print("wrote", dst, "mag thresh", out["mag_thresh"], "joint flags", out["magj_flags"].tolist(), "single", out["mags_flags"].tolist(),
          "tea thresh", out["tea_thresh"], "tea flags", out["teaj_flags"].tolist())


if __name__ == "__main__":
    main(mixed="mixed" in sys.argv[1:])
