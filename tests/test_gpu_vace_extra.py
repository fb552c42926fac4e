"""-m gpu: two VACE corners against tests/golden/vace_extra.npz (the reference's own WanModel, oracle/make_golden_vace_extra.py):
several contexts mixed in one call (model.py:1905-1912, :617-629, :713-719) and VACE together with MagCache (:1914-2064).
Tolerance: relative L2 <= 2.5e-2 against the reference's bf16 result, the forward bar of tests/test_gpu_model.py; cache decisions
must equal the reference's exactly."""
import os

import numpy as np
import pytest
import torch

from oracle import wan_oracle as O
from oracle.make_golden_skipcache import MAG_RATIOS, STEPS
from oracle.make_golden_vace_extra import SCALES, SEED_W, inputs

pytestmark = pytest.mark.gpu
G = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "vace_extra.npz")))
CFG = O.make_config("tiny_vace")


def rel(a, b):
    a, b = a.float().cpu(), torch.as_tensor(b).float()
    return ((a - b).norm() / b.norm()).item()


@pytest.fixture(scope="module")
def model():
    from wan2gp_amd.model import WanModelHIP
    m = WanModelHIP(model_type=CFG.model_type, dim=CFG.dim, ffn_dim=CFG.ffn_dim, num_heads=CFG.num_heads, num_layers=CFG.num_layers,
                    in_dim=CFG.in_dim, vace_layers=list(CFG.vace_layers), vace_in_dim=CFG.vace_in_dim)
    return m.load_state_dict(O.synth_weights(CFG, seed=SEED_W))


def test_two_contexts_with_their_own_scales(model):
    lat, _, _, ctx, ctx_null, v0, v1 = inputs(CFG)
    run = lambda vs, sc: model([lat.cuda(), lat.cuda()], t=torch.tensor([588]), context=[ctx.cuda(), ctx_null.cuda()],
                               vace_context=[v.cuda() for v in vs], vace_context_scale=list(sc))
    for n, sc in enumerate(SCALES):
        o = run((v0, v1), sc)
        e = max(rel(o[0], G[f"mc{n}_0"]), rel(o[1], G[f"mc{n}_1"]))
        print(f"[VACE x2 contexts, scales {sc}] rel err vs reference {e:.3e}")
        assert e <= 2.5e-2
    # scale 0 switches a context off entirely: (0, 0.7) over (v0, v1) is bit-equal to 0.7 over v1 alone
    a, b = run((v0, v1), SCALES[1]), run((v1,), (0.7,))
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    # the second context matters
    assert rel(run((v0, v1), SCALES[0])[0], run((v0,), (1.0,))[0].cpu()) > 1e-3
    from wan2gp_amd.lib import WanHipError
    with pytest.raises(WanHipError, match="scales"):
        run((v0, v1), (1.0,))


def test_vace_with_magcache_vs_reference(model):
    from wan2gp_amd.skipcache import SkipStepsCache, reset_for_generation
    _, lats, ts, ctx, ctx_null, v0, _ = inputs(CFG)
    c = SkipStepsCache(cache_type="mag", multiplier=2.0, start_step=1, num_steps=STEPS, skipped_steps=0, previous_residual=None,
                       previous_modulated_input=None)
    c.update({"magcache_thresh": 0, "magcache_K": 2, "def_mag_ratios": list(MAG_RATIOS)})
    reset_for_generation(c, 2)
    model.cache = c
    try:
        assert model.compute_magcache_threshold(c.start_step, ts, c.multiplier) == pytest.approx(float(G["mag_thresh"][0]), abs=1e-12)
        worst = 0.0
        for i in range(STEPS):
            outs = model([lats[i].cuda(), lats[i].cuda()], t=torch.stack([ts[i]]), context=[ctx.cuda(), ctx_null.cuda()], real_step_no=i,
                         current_step_no=i, vace_context=[v0.cuda()], vace_context_scale=[1.0])
            assert [int(s == 0) for s in c.accumulated_steps] == G["vmag_flags"][i].tolist(), i
            for k in range(2):
                worst = max(worst, rel(outs[k], G[f"vmag_{i}_{k}"]))
        print(f"VACE + MagCache: skipped {c.skipped_steps}/{STEPS}, worst rel err {worst:.4f}")
        assert c.skipped_steps == 4 and worst <= 2.5e-2
    finally:
        model.cache = None


def test_per_frame_timesteps_with_magcache_vs_reference():
    """ti2v timestep injection (t = [0, t]) with MagCache on the 48-channel model: decisions equal the reference's, outputs of
    computed and skipped steps within the forward bar."""
    from oracle.make_golden_vace_extra import inputs_ti2v
    from wan2gp_amd.model import WanModelHIP
    from wan2gp_amd.skipcache import SkipStepsCache, reset_for_generation
    cfg = O.make_config("tiny_ti2v")
    m = WanModelHIP(model_type=cfg.model_type, dim=cfg.dim, ffn_dim=cfg.ffn_dim, num_heads=cfg.num_heads, num_layers=cfg.num_layers,
                    in_dim=cfg.in_dim, out_dim=cfg.out_dim).load_state_dict(O.synth_weights(cfg, seed=SEED_W))
    lats, ts, ctx, ctx_null = inputs_ti2v(cfg)
    c = SkipStepsCache(cache_type="mag", multiplier=2.0, start_step=1, num_steps=STEPS, skipped_steps=0, previous_residual=None,
                       previous_modulated_input=None)
    c.update({"magcache_thresh": 0, "magcache_K": 2, "def_mag_ratios": list(MAG_RATIOS)})
    reset_for_generation(c, 2)
    m.cache = c
    m.compute_magcache_threshold(c.start_step, ts, c.multiplier)
    worst = 0.0
    for i in range(STEPS):
        tf = torch.stack([torch.zeros(()), ts[i]])
        outs = m([lats[i].cuda(), lats[i].cuda()], t=tf, context=[ctx.cuda(), ctx_null.cuda()], real_step_no=i, current_step_no=i)
        assert [int(s == 0) for s in c.accumulated_steps] == G["tfmag_flags"][i].tolist(), i
        for k in range(2):
            worst = max(worst, rel(outs[k], G[f"tfmag_{i}_{k}"]))
    print(f"per-frame t + MagCache: skipped {c.skipped_steps}/{STEPS}, worst rel err {worst:.4f}")
    assert c.skipped_steps == 4 and worst <= 2.5e-2
