"""-m gpu: normalized attention guidance (text_cross_attention's NAG branch, models/wan/modules/model.py:245-293; switched on
by generate(NAG_scale > 1), models/wan/any2video.py:607-608) through the C ABI: `wan_nag_combine` against the oracle's
statement-by-statement restatement, and WanModelHIP forwards with a (positive ; negative) context against
tests/golden/nag.npz = the reference's own WanModel run with offload.shared_state["_nag_*"] set (oracle/make_golden_nag.py).

Tolerances.  wan_nag_combine follows the reference's bf16 rounding points; what can differ is the fp32 summation order of the
two L1 norms (a last-bit difference moves the bf16 norm by one ulp in ~2^-16 of the rows, and with it every element of a
clipped row's guidance term by <= 1 bf16 ulp): >= 99 % of the elements must be bit-identical, every element within 2 bf16 ulp
of the magnitude its last rounding step worked on.  Forwards: the
criterion of tests/test_gpu_model.py (no further from the fp32 anchor than the reference's bf16 run is, x 1.5 + 2e-3).
"""
import os

import numpy as np
import pytest
import torch

from oracle import wan_oracle as O

pytestmark = pytest.mark.gpu
G = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "nag.npz")))
BF = torch.bfloat16


def rel(a, b):
    return ((a - b).norm() / b.norm()).item()


def check_rows(got, ref, mag, what):
    """mag: magnitude the last rounding step worked on, |ref| + (1 - alpha) |x_pos| >= |alpha g'| -- the result is a sum of two
    rounded terms and may be far smaller than either, so a one-ulp difference of a term is many ulps of a small result."""
    got, ref = got.float().cpu(), ref.float()
    assert torch.isfinite(got).all(), what
    same = (got == ref).float().mean().item()
    ulp = mag.float().clamp_min(1e-30) * 2.0 ** -7               # one (coarse) bf16 ulp at that magnitude
    worst = ((got - ref).abs() / ulp).max().item()
    print(f"[nag_combine {what}] identical {same:.5f}, worst {worst:.2f} ulp")
    assert same >= 0.99 and worst <= 2.0, (what, same, worst)


@pytest.mark.parametrize("d,rows", [(256, 37), (1536, 515), (5120, 1030)])
@pytest.mark.parametrize("nag", [(11.0, 2.5, 0.25), (3.3, 3.3, 0.3), (1.5, 3.5, 0.5)])
def test_nag_combine_vs_oracle(d, rows, nag):
    from wan2gp_amd import ops
    g = torch.Generator().manual_seed(d + rows)
    xp = torch.randn(1, rows, d, generator=g).to(BF)
    xn = (0.6 * xp.float() + 0.8 * torch.randn(1, rows, d, generator=g)).to(BF)
    xn[0, 1] = xp[0, 1]                                         # guidance == positive: ratio 1, never clipped
    xp[0, 2] = 0                                                # |x_pos|_1 = 0: ratio inf -> largest bf16 (nan_to_num), factor 0
    xp[0, 3] = 0; xn[0, 3] = 0                                  # 0 / 0 = nan -> 10
    ref = O.nag_combine(xp, xn, *nag)
    ratio = (xn.float() * (1 - nag[0]) + nag[0] * xp.float()).abs().sum(-1) / xp.float().abs().sum(-1)
    clipped = int((ratio > nag[1]).sum())
    print(f"rows clipped: {clipped} of {rows}")
    if nag[0] > 3:
        assert 0 < clipped < rows                               # both arms of the norm clip are exercised
    got = ops.nag_combine(xp.cuda(), xn.cuda(), *nag)
    check_rows(got, ref, ref.float().abs() + (1 - nag[2]) * xp.float().abs(), f"d={d} nag={nag}")
    a, b = xp.cuda(), xn.cuda()
    assert torch.equal(ops.nag_combine(a, b, *nag, out=a), got)          # in place over x_pos ...
    a = xp.cuda()
    assert torch.equal(ops.nag_combine(a, b, *nag, out=b), got)          # ... and over x_neg


def build(cfg, W):
    from wan2gp_amd.model import WanModelHIP
    m = WanModelHIP(model_type=cfg.model_type, dim=cfg.dim, ffn_dim=cfg.ffn_dim, num_heads=cfg.num_heads, num_layers=cfg.num_layers,
                    in_dim=cfg.in_dim, out_dim=cfg.out_dim,
                    **({} if cfg.vace_layers is None else {"vace_layers": list(cfg.vace_layers), "vace_in_dim": cfg.vace_in_dim}))
    return m.load_state_dict(W)


@pytest.mark.parametrize("name", ["small", "tiny_i2v21"])
def test_forward_with_nag_vs_reference_golden(name):
    """CFG pair: the cond stream carries (positive ; negative) = [2,512,4096], the uncond stream a plain context -- the two
    shapes any2video.py:1551 hands to the model when NAG and CFG are both on; tiny_i2v21 adds the CLIP-token branch."""
    cfg = O.make_config(name)
    W = O.synth_weights(cfg)
    f, h, w = (int(v) for v in G[f"fwd_{name}_shape"])
    lat, c, cn, y = O.synth_inputs(cfg, f, h, w)
    clip = O.synth_clip_fea() if cfg.model_type == "i2v" else None
    t = torch.tensor([int(G[f"fwd_{name}_t"][0])], dtype=torch.int64)
    nag = tuple(float(v) for v in G["fwd_nag"])
    m = build(cfg, W)
    c2 = torch.cat([c, cn])
    kw = {} if clip is None else {"clip_fea": clip.cuda()}
    with pytest.raises(Exception):                               # a batch-2 context without NAG parameters is refused
        m([lat.cuda()], t=t, context=[c2.cuda()], y=None if y is None else y.cuda(), **kw)
    m.nag = nag
    outs = m([lat.cuda(), lat.cuda()], t=t, context=[c2.cuda(), cn.cuda()], y=None if y is None else y.cuda(), **kw)
    W32 = O.synth_weights(cfg, dtype=torch.float32)
    anchor = O.dit_forward([lat, lat], t, [c2.float(), cn.float()], W32, cfg, y=y, dtype=torch.float32, exact=True,
                           clip_fea=None if clip is None else clip.float(), nag=nag)
    for o, key, a in zip(outs, ("cond", "uncond"), anchor):
        ref = torch.from_numpy(G[f"fwd_{name}_{key}"])
        err_ref, err_hip = rel(ref, a), rel(o.cpu(), a)
        print(f"nag {name}/{key}: err_ref={err_ref:.4e} err_hip={err_hip:.4e} hip-vs-ref={rel(o.cpu(), ref):.4e}")
        assert err_hip <= 1.5 * err_ref + 2e-3, (err_hip, err_ref)
        assert rel(o.cpu(), ref) <= 2.5e-2
    # the guidance is really in the result: the same stream without it sits several times further from the golden
    m.nag = None
    plain = m([lat.cuda()], t=t, context=[c.cuda()], y=None if y is None else y.cuda(), **kw)[0].cpu()
    gold = torch.from_numpy(G[f"fwd_{name}_cond"])
    assert rel(plain, gold) > 3 * rel(outs[0].cpu(), gold)
    # a single NAG stream (guidance scale 1, the distilled-LoRA use of NAG) equals the cond stream of the pair
    m.nag = nag
    solo = m([lat.cuda()], t=t, context=[c2.cuda()], y=None if y is None else y.cuda(), **kw)[0]
    assert rel(solo.cpu(), outs[0].cpu()) <= 1e-6


def test_forward_with_nag_inside_vace_context_blocks():
    """NAG inside the VACE context blocks (their cross-attention sees the same batch-2 context, model.py:816-828) against the
    oracle."""
    cfg = O.make_config("tiny_vace")
    W = O.synth_weights(cfg)
    f, h, w = 2, 8, 8
    lat, c, cn, _ = O.synth_inputs(cfg, f, h, w)
    vace = O.synth_vace_context(cfg, f, h, w)
    t = torch.tensor([588])
    nag = (11.0, 2.5, 0.25)
    c2 = torch.cat([c, cn])
    m = build(cfg, W)
    m.nag = nag
    got = m([lat.cuda(), lat.cuda()], t=t, context=[c2.cuda(), cn.cuda()], vace_context=[vace.cuda()], vace_context_scale=[1.0])
    ref = O.dit_forward([lat, lat], t, [c2, cn], W, cfg, vace_context=vace, nag=nag)
    W32 = O.synth_weights(cfg, dtype=torch.float32)
    anchor = O.dit_forward([lat, lat], t, [c2.float(), cn.float()], W32, cfg, dtype=torch.float32, exact=True, vace_context=vace, nag=nag)
    for o, r, a in zip(got, ref, anchor):
        assert rel(o.cpu(), a) <= 1.5 * rel(r, a) + 2e-3 and rel(o.cpu(), r) <= 2.5e-2


def test_forward_with_nag_fp8_checkpoint_vs_oracle():
    """Scaled-fp8 block Linears: a (positive ; negative) context is ONE tensor for the activation quantisation of the text K / V
    projections (the reference quantises per tensor, shared/qtypes/scaled_fp8.py:150-169) -- the oracle's fp8 plan does the same."""
    cfg = O.make_config("small")
    Wb = O.synth_weights(cfg)
    W8 = O.quantize_checkpoint_fp8(Wb, per_row=True)
    lat, c, cn, _ = O.synth_inputs(cfg, 3, 10, 14)
    t = torch.tensor([412])
    nag = (11.0, 2.5, 0.25)
    c2 = torch.cat([c, cn])
    m = build(cfg, W8)
    m.nag = nag
    got = [o.cpu() for o in m([lat.cuda(), lat.cuda()], t=t, context=[c2.cuda(), cn.cuda()])]
    ref = O.dit_forward([lat, lat], t, [c2, cn], W8, cfg, nag=nag)
    mb = build(cfg, Wb)
    mb.nag = nag
    got_bf16 = [o.cpu() for o in mb([lat.cuda(), lat.cuda()], t=t, context=[c2.cuda(), cn.cuda()])]
    for o, ob, r in zip(got, got_bf16, ref):
        d, dq = rel(o, r), rel(ob, r)
        print(f"[nag fp8] hip-fp8 vs oracle-fp8 {d:.3e}; bf16 checkpoint vs oracle-fp8 {dq:.3e}")
        assert torch.isfinite(o).all() and d <= 2.5e-2, (d, dq)      # the scale of the fp8 plan's own noise (tests/test_gpu_fp8.py)


def test_generate_with_nag_runs_the_loop():
    """generate(NAG_scale=...) end to end on the tiny model: 3 steps at guidance 1 (one NAG stream per step) follow the oracle's
    loop with the same stacked context."""
    from wan2gp_amd.pipeline import WanAny2VHIP
    cfg = O.make_config("tiny")
    W = O.synth_weights(cfg)
    f, h, w = 2, 8, 8
    lat, c, cn, _ = O.synth_inputs(cfg, f, h, w, seed=5)
    m = build(cfg, W)
    pipe = WanAny2VHIP(m, device="cuda")
    out = pipe.generate(context=c.cuda(), context_null=cn.cuda(), width=w * 8, height=h * 8, frame_num=(f - 1) * 4 + 1, shift=5.0,
                        sampling_steps=3, guide_scale=1.0, latents=lat, return_latents=True, NAG_scale=11, NAG_tau=2.5, NAG_alpha=0.25)
    assert m.nag == (11.0, 2.5, 0.25)
    sch = O.UniPCOracle()
    ts = sch.set_timesteps(3, 5.0)
    x = lat.clone()
    c2 = torch.cat([c, cn])
    for tt in ts:
        v = O.dit_forward([x], torch.stack([tt]), [c2], W, cfg, nag=(11.0, 2.5, 0.25))[0]
        x = sch.step(v, x)
    r = rel(out["latents"].cpu(), x)
    print(f"[nag generate] 3-step latents vs oracle loop: {r:.3e}")
    assert r <= 4e-2


def test_nag_combine_at_the_14b_720p_shape_scaling_and_row_properties():
    """BASELINE configs[2] size (75,600 token rows x 5120 channels, beyond what the oracle checks in seconds): properties that do
    not depend on the size.  (a) powers of two commute with every bf16 / fp32 rounding, and the guidance is homogeneous of degree
    one in (x_pos, x_neg): scaling both by 4 scales the result by 4 BIT-EXACTLY; (b) rows are independent: reversing the row order
    reverses the result; (c) a sample of rows equals the oracle."""
    from wan2gp_amd import ops
    rows, d, nag = 75600, 5120, (11.0, 2.5, 0.25)
    g = torch.Generator(device="cuda").manual_seed(11)
    xp = torch.randn(rows, d, device="cuda", generator=g).to(BF)
    xn = (0.6 * xp.float() + 0.8 * torch.randn(rows, d, device="cuda", generator=g)).to(BF)
    out = ops.nag_combine(xp, xn, *nag)
    assert torch.isfinite(out.float()).all()
    assert torch.equal(ops.nag_combine(xp * 4, xn * 4, *nag), out * 4)
    assert torch.equal(ops.nag_combine(xp.flip(0).contiguous(), xn.flip(0).contiguous(), *nag).flip(0), out)
    idx = torch.arange(0, rows, 997, device="cuda")
    ref = O.nag_combine(xp[idx].cpu().unsqueeze(0), xn[idx].cpu().unsqueeze(0), *nag)[0]
    check_rows(out[idx], ref, ref.float().abs() + (1 - nag[2]) * xp[idx].float().cpu().abs(), "14B-720p rows, sampled")


def test_nag_together_with_magcache():
    """NAG through the step-skipping entry (wan_dit_forward_ex carries should_calc / residual beside nag_* / context_batches): on
    the steps MagCache computes the result is the uncached NAG forward's, bit for bit; on the steps it skips the stored residual
    is replayed (finite, near the computed result)."""
    from oracle.make_golden_skipcache import MAG_RATIOS, STEPS, inputs
    from wan2gp_amd.skipcache import SkipStepsCache, reset_for_generation
    cfg = O.make_config("tiny")
    m = build(cfg, O.synth_weights(cfg, seed=4321))
    lats, ts, c, cn = inputs(cfg)
    c2 = torch.cat([c, cn]).cuda()
    m.nag = (11.0, 2.5, 0.25)
    plain = [m([lats[i].cuda(), lats[i].cuda()], t=torch.stack([ts[i]]), context=[c2, cn.cuda()]) for i in range(STEPS)]
    cache = m.cache = SkipStepsCache(cache_type="mag", multiplier=2.0, start_step=1, num_steps=STEPS, skipped_steps=0,
                                     previous_residual=None, previous_modulated_input=None)
    cache.update({"magcache_thresh": 0, "magcache_K": 2, "def_mag_ratios": list(MAG_RATIOS)})
    reset_for_generation(cache, 2)
    try:
        m.compute_magcache_threshold(cache.start_step, ts, cache.multiplier)
        n_skipped = 0
        for i in range(STEPS):
            outs = m([lats[i].cuda(), lats[i].cuda()], t=torch.stack([ts[i]]), context=[c2, cn.cuda()], real_step_no=i, current_step_no=i)
            for k in range(2):
                if cache.accumulated_steps[k] == 0:                      # computed this step
                    assert torch.equal(outs[k], plain[i][k]), (i, k)
                else:
                    n_skipped += 1
                    assert torch.isfinite(outs[k]).all() and rel(outs[k], plain[i][k]) < 0.5, (i, k)
        assert n_skipped > 0 and cache.skipped_steps > 0
    finally:
        m.cache = None
