"""TEST INFRASTRUCTURE ONLY -- tests/golden/vace_extra.npz from the REFERENCE's own WanModel (needs /root/reference).
Run in the build container:   python oracle/make_golden_vace_extra.py

Two VACE corners the single-context forward fixture (forward_tiny_vace.npz) does not reach:
  * several contexts in one call (model.py:1905-1912 one hint list per context, :617-629 each through the context block,
    :713-719 added in order with its own scale; a scale of 0 switches a context off): scales (1.0, 0.6) and (0.0, 0.7);
  * VACE together with a step-skipping cache (MagCache joint pass, model.py:1914-2064): a skipped stream skips its context blocks
    with its main blocks, the stored residual of a computed stream includes the hints;
and one per-frame-timestep corner:
  * ti2v timestep injection (t = [0, t]: the injected source frame keeps timestep 0, any2video.py:1496-1499) together with
    MagCache on the 48-channel model (wan_handler enables mag_cache, not tea_cache, for the 5B model).
"""
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
from oracle.make_golden_skipcache import MAG_RATIOS, STEPS, new_cache  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "vace_extra.npz")
SEED_W, F_, H_, W_ = 1234, 2, 8, 8
SCALES = ((1.0, 0.6), (0.0, 0.7))


def inputs(cfg):
    lat, ctx, ctx_null, _ = O.synth_inputs(cfg, F_, H_, W_, seed=6)
    v0, v1 = O.synth_vace_context(cfg, F_, H_, W_, seed=13), O.synth_vace_context(cfg, F_, H_, W_, seed=14)
    g = torch.Generator().manual_seed(18)
    drift = torch.randn(lat.shape, generator=g)
    lats = [lat + 0.15 * i * drift for i in range(STEPS)]
    ts = [torch.tensor(v, dtype=torch.float32) for v in (999.0, 950.0, 880.0, 790.0, 660.0, 500.0, 320.0, 130.0)]
    return lat, lats, ts, ctx, ctx_null, v0, v1


def inputs_ti2v(cfg):
    lat, ctx, ctx_null, _ = O.synth_inputs(cfg, F_, H_, W_, seed=8)
    g = torch.Generator().manual_seed(19)
    drift = torch.randn(lat.shape, generator=g)
    lats = [lat + 0.15 * i * drift for i in range(STEPS)]
    ts = [torch.tensor(v, dtype=torch.float32) for v in (999.0, 950.0, 880.0, 790.0, 660.0, 500.0, 320.0, 130.0)]
    return lats, ts, ctx, ctx_null


def main():
    ns = ref_shim.load()
    cfg = O.make_config("tiny_vace")
    W = O.synth_weights(cfg, seed=SEED_W)
    m = build_ref_model(ns, cfg, W, torch.bfloat16)
    lat, lats, ts, ctx, ctx_null, v0, v1 = inputs(cfg)
    freqs = ns.P.get_rotary_pos_embed(lat.shape[2:])
    pipe = types.SimpleNamespace(_interrupt=False)
    out = {}
    for n, sc in enumerate(SCALES):
        with torch.no_grad():
            r = m([lat.clone(), lat.clone()], t=torch.tensor([588]), context=[ctx.clone(), ctx_null.clone()], freqs=freqs, pipeline=pipe,
                  vace_context=[v0.clone(), v1.clone()], vace_context_scale=list(sc))
        out[f"mc{n}_0"], out[f"mc{n}_1"] = r[0].float().numpy(), r[1].float().numpy()
    c = m.cache = new_cache("mag")
    c.previous_residual = [None] * 2
    out["mag_thresh"] = np.array([m.compute_magcache_threshold(c.start_step, ts, c.multiplier)])
    c.accumulated_err, c.accumulated_steps, c.accumulated_ratio, c.one_for_all = [0.0] * 2, [0] * 2, [1.0] * 2, False
    flags = []
    for i in range(STEPS):
        with torch.no_grad():
            r = m([lats[i].clone(), lats[i].clone()], t=torch.stack([ts[i]]), context=[ctx.clone(), ctx_null.clone()], freqs=freqs,
                  pipeline=pipe, real_step_no=i, current_step_no=i, vace_context=[v0.clone()], vace_context_scale=[1.0])
        flags.append([int(c.accumulated_steps[k] == 0) for k in range(2)])
        out[f"vmag_{i}_0"], out[f"vmag_{i}_1"] = r[0].float().numpy(), r[1].float().numpy()
    out["vmag_flags"] = np.array(flags)
    m.cache = None
    # --- per-frame timesteps + MagCache (tiny ti2v, 48 channels) ---
    cfg2 = O.make_config("tiny_ti2v")
    m2 = build_ref_model(ns, cfg2, O.synth_weights(cfg2, seed=SEED_W), torch.bfloat16)
    lats2, ts2, ctx2, ctx2n = inputs_ti2v(cfg2)
    freqs2 = ns.P.get_rotary_pos_embed(lats2[0].shape[2:])
    c = m2.cache = new_cache("mag")
    c.previous_residual = [None] * 2
    m2.compute_magcache_threshold(c.start_step, ts2, c.multiplier)
    c.accumulated_err, c.accumulated_steps, c.accumulated_ratio, c.one_for_all = [0.0] * 2, [0] * 2, [1.0] * 2, False
    flags = []
    for i in range(STEPS):
        tf = torch.stack([torch.zeros(()), ts2[i]])                       # source frame at t = 0
        with torch.no_grad():
            r = m2([lats2[i].clone(), lats2[i].clone()], t=tf, context=[ctx2.clone(), ctx2n.clone()], freqs=freqs2, pipeline=pipe,
                   real_step_no=i, current_step_no=i)
        flags.append([int(c.accumulated_steps[k] == 0) for k in range(2)])
        out[f"tfmag_{i}_0"], out[f"tfmag_{i}_1"] = r[0].float().numpy(), r[1].float().numpy()
    out["tfmag_flags"] = np.array(flags)
    m2.cache = None
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, "flags", out["vmag_flags"].tolist(), "thresh", out["mag_thresh"])


if __name__ == "__main__":
    main()
