"""CPU: the oracle against tests/golden/vace_extra.npz (the reference's own WanModel): several VACE contexts in one call, and
VACE together with MagCache.  Bit-exact in the bf16 plan, like every other oracle pin."""
import os

import numpy as np
import torch

from oracle import skipcache_oracle as SO
from oracle import wan_oracle as O
from oracle.make_golden_skipcache import STEPS, new_cache
from oracle.make_golden_vace_extra import SCALES, SEED_W, inputs

G = os.path.join(os.path.dirname(__file__), "golden", "vace_extra.npz")


def test_multi_context_forward_is_bit_exact():
    g = dict(np.load(G))
    cfg = O.make_config("tiny_vace")
    W = O.synth_weights(cfg, seed=SEED_W)
    lat, _, _, ctx, ctx_null, v0, v1 = inputs(cfg)
    for n, sc in enumerate(SCALES):
        r = O.dit_forward([lat, lat], torch.tensor([588]), [ctx, ctx_null], W, cfg, vace_context=[v0, v1], vace_scale=list(sc))
        assert torch.equal(r[0], torch.from_numpy(g[f"mc{n}_0"])) and torch.equal(r[1], torch.from_numpy(g[f"mc{n}_1"])), sc
    # a context with scale 0 is really off: (0, 0.7) over (v0, v1) == 0.7 over v1 alone
    solo = O.dit_forward([lat, lat], torch.tensor([588]), [ctx, ctx_null], W, cfg, vace_context=[v1], vace_scale=[0.7])
    assert torch.equal(solo[0], torch.from_numpy(g["mc1_0"]))


def test_vace_with_magcache_is_bit_exact():
    g = dict(np.load(G))
    cfg = O.make_config("tiny_vace")
    W = O.synth_weights(cfg, seed=SEED_W)
    _, lats, ts, ctx, ctx_null, v0, _ = inputs(cfg)
    c = SO.Cache(**new_cache("mag").__dict__)
    c.previous_residual = [None] * 2
    assert SO.magcache_threshold(c, c.start_step, ts, c.multiplier) == float(g["mag_thresh"][0])
    c.accumulated_err, c.accumulated_steps, c.accumulated_ratio, c.one_for_all = [0.0] * 2, [0] * 2, [1.0] * 2, False
    for i in range(STEPS):
        outs, flags = SO.dit_forward_cached([lats[i], lats[i]], torch.stack([ts[i]]), [ctx, ctx_null], W, cfg, c, real_step_no=i,
                                            vace_context=[v0], vace_scale=[1.0])
        assert [int(f) for f in flags] == g["vmag_flags"][i].tolist(), i
        assert torch.equal(outs[0], torch.from_numpy(g[f"vmag_{i}_0"])) and torch.equal(outs[1], torch.from_numpy(g[f"vmag_{i}_1"])), i


def test_per_frame_timesteps_with_magcache_is_bit_exact():
    """ti2v timestep injection (t = [0, t]) together with MagCache on the 48-channel model."""
    from oracle.make_golden_vace_extra import inputs_ti2v
    g = dict(np.load(G))
    cfg = O.make_config("tiny_ti2v")
    W = O.synth_weights(cfg, seed=SEED_W)
    lats, ts, ctx, ctx_null = inputs_ti2v(cfg)
    c = SO.Cache(**new_cache("mag").__dict__)
    c.previous_residual = [None] * 2
    SO.magcache_threshold(c, c.start_step, ts, c.multiplier)
    c.accumulated_err, c.accumulated_steps, c.accumulated_ratio, c.one_for_all = [0.0] * 2, [0] * 2, [1.0] * 2, False
    for i in range(STEPS):
        tf = torch.stack([torch.zeros(()), ts[i]])
        outs, flags = SO.dit_forward_cached([lats[i], lats[i]], tf, [ctx, ctx_null], W, cfg, c, real_step_no=i)
        assert [int(f) for f in flags] == g["tfmag_flags"][i].tolist(), i
        assert torch.equal(outs[0], torch.from_numpy(g[f"tfmag_{i}_0"])) and torch.equal(outs[1], torch.from_numpy(g[f"tfmag_{i}_1"])), i
