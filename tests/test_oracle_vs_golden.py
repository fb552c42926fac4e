"""Pins oracle/wan_oracle.py (the CPU restatement) to fixtures produced by the
REFERENCE's own modules (oracle/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import wan_oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return dict(np.load(os.path.join(G, name)))


def t(a):
    return torch.from_numpy(np.asarray(a))


def test_rope_tables_bit_exact():
    g = load("ops.npz")
    cos, sin = O.rope_tables((3, 4, 6))
    assert torch.equal(cos, t(g["rope_cos_3x4x6"]))
    assert torch.equal(sin, t(g["rope_sin_3x4x6"]))


def _ops_inputs():
    g = torch.Generator().manual_seed(7)
    L, H, D = 72, 2, 128
    x = torch.randn(1, L, H * D, generator=g).to(torch.bfloat16)
    w = (1 + 0.02 * torch.randn(H * D, generator=g)).to(torch.bfloat16)
    b3 = (0.01 * torch.randn(H * D, generator=g)).to(torch.bfloat16)
    v = torch.randn(1, L, H, D, generator=g).to(torch.bfloat16)
    return x, w, b3, v


def test_riflex_rope_tables_bit_exact_oracle_and_product():
    """RIFLEx (long videos, posemb_layers.py:35-85, :417-419): the oracle and the product's host function both reproduce the
    reference's tables exactly; the plain tables are untouched by the flag's code path."""
    from wan2gp_amd.rope import get_rotary_pos_embed
    g = load("ops.npz")
    cos, sin = O.rope_tables((33, 2, 3), riflex=True)
    assert torch.equal(cos, t(g["riflex_cos_33x2x3"])) and torch.equal(sin, t(g["riflex_sin_33x2x3"]))
    pc, ps = get_rotary_pos_embed((33, 4, 6), enable_RIFLEx=True)
    assert torch.equal(pc, t(g["riflex_cos_33x2x3"])) and torch.equal(ps, t(g["riflex_sin_33x2x3"]))
    pc, ps = get_rotary_pos_embed((3, 8, 12))
    assert torch.equal(pc, t(g["rope_cos_3x4x6"])) and torch.equal(ps, t(g["rope_sin_3x4x6"]))
    assert not torch.equal(O.rope_tables((33, 2, 3))[0], cos)


def test_rmsnorm_rope_ln_sdpa_bit_exact():
    g = load("ops.npz")
    x, w, b3, v = _ops_inputs()
    cos, sin = O.rope_tables((3, 4, 6))
    q = O.rms_norm(x, w, 1e-6)
    assert torch.equal(q.float(), t(g["rms_bf16"]))
    k = O.rms_norm(torch.flip(x, dims=[1]), w, 1e-6)
    qq = O.rope_apply(q.view(1, 72, 2, 128), cos, sin)
    kk = O.rope_apply(k.view(1, 72, 2, 128), cos, sin)
    assert torch.equal(qq.float(), t(g["rope_q_bf16"]))
    assert torch.equal(kk.float(), t(g["rope_k_bf16"]))
    assert torch.equal(O.layer_norm(x, 1e-6).float(), t(g["ln_bf16"]))
    assert torch.equal(b3.float(), t(g["ln3_bias"]))
    assert torch.equal(O.layer_norm(x, 1e-6, w, b3).float(), t(g["ln3_bf16"]))
    assert torch.equal(O.attention(qq, kk, v).float(), t(g["sdpa_bf16"]))
    # the backend-independent fp32 attention agrees with sdpa to bf16 rounding
    ex = O.attention(qq, kk, v, exact=True).float()
    assert (ex - t(g["sdpa_bf16"])).abs().max() < 2e-2


@pytest.mark.parametrize("name", ["tiny", "tiny_i2v", "tiny_ti2v", "tiny_i2v21", "tiny_flf2v", "tiny_vace", "small"])
@pytest.mark.parametrize("tag,dtype", [("bf16", torch.bfloat16), ("fp32", torch.float32)])
def test_forward_matches_reference(name, tag, dtype):
    g = load(f"forward_{name}.npz")
    f, h, w = [int(v) for v in g["shape"]]
    cfg = O.make_config(name)
    W = O.synth_weights(cfg, dtype=dtype)
    lat, ctx, ctx_null, y = O.synth_inputs(cfg, f, h, w)
    tt = torch.tensor([int(g["t"][0])], dtype=torch.int64)
    clip = O.synth_clip_fea(images=2 if cfg.flf else 1) if cfg.model_type == "i2v" else None   # Wan2.1 i2v / flf2v: CLIP tokens (model.py:1858-1869)
    vace = O.synth_vace_context(cfg, f, h, w) if cfg.vace_layers is not None else None     # VACE context blocks (model.py:790-828)
    cond, uncond = O.dit_forward([lat, lat], tt, [ctx.to(dtype), ctx_null.to(dtype)], W, cfg, y=y, dtype=dtype, clip_fea=clip,
                                 vace_context=vace)
    if vace is not None and dtype == torch.bfloat16:                 # fractional context scale: x.add_(hint, alpha=0.6)
        c6, u6 = O.dit_forward([lat, lat], tt, [ctx.to(dtype), ctx_null.to(dtype)], W, cfg, dtype=dtype, vace_context=vace, vace_scale=0.6)
        assert torch.equal(c6, t(g["cond_s06_bf16"])) and torch.equal(u6, t(g["uncond_s06_bf16"]))
        assert not torch.equal(c6, cond)
    if name == "tiny_ti2v" and dtype == torch.bfloat16:              # per-frame timesteps (first latent frame at t = 0)
        tf = torch.full((f,), int(g["t"][0]), dtype=torch.int64)
        tf[:1] = 0
        c2, u2 = O.dit_forward([lat, lat], tf, [ctx.to(dtype), ctx_null.to(dtype)], W, cfg, dtype=dtype)
        assert torch.equal(c2, t(g["cond_tframe_bf16"])) and torch.equal(u2, t(g["uncond_tframe_bf16"])) and not torch.equal(c2, cond)
    if dtype == torch.bfloat16:
        assert torch.equal(cond, t(g["cond_bf16"])) and torch.equal(uncond, t(g["uncond_bf16"]))
    else:
        # the fp32 anchor is the same graph without the bf16 roundings: it must sit within
        # bf16-plan noise of the reference's bf16 result (the reference cannot itself be run
        # "fp32 everywhere": WanRMSNorm aliases/squares fp32 inputs, model.py:165-166)
        for a, b in ((cond, g["cond_bf16"]), (uncond, g["uncond_bf16"])):
            rel = (a - t(b)).norm() / t(b).norm()
            assert rel < 3e-2, rel
    # isolated block
    gen = torch.Generator().manual_seed(11)
    L = f * (h // 2) * (w // 2)
    hid = torch.randn(1, L, cfg.dim, generator=gen).to(dtype)
    e0 = (0.5 * torch.randn(1, 6, cfg.dim, generator=gen)).to(dtype)
    cemb = (0.5 * torch.randn(1, 512 + (O.CLIP_TOKENS * (2 if cfg.flf else 1) if cfg.model_type == "i2v" else 0), cfg.dim, generator=gen)).to(dtype)
    cos, sin = O.rope_tables((f, h // 2, w // 2))
    bo = O.block_forward(hid, e0, cemb, cos, sin, W, 0, cfg)
    if dtype == torch.bfloat16:
        assert torch.equal(bo.float(), t(g["block0_bf16"]))
    else:
        rel = (bo - t(g["block0_bf16"])).norm() / t(g["block0_bf16"]).norm()
        assert rel < 1e-2, rel


@pytest.mark.parametrize("name", ["tiny", "tiny_i2v", "tiny_ti2v", "small", "tiny_i2v21", "tiny_flf2v"])
def test_forward_mixed_precision_plan_matches_reference(name):
    """`mixed_precision_transformer` (wgp.py:4039 -> any2video.py:190 -> lock_layers_dtypes(torch.float32), model.py:1330-1371): the time
    MLP, the time projection and every norm3 hold their bf16-valued weights in fp32, and by type promotion the residual stream, e / e0 and
    every modulate / gated residual run in fp32 between bf16 Linears.  The goldens are the reference's own forward with those locks
    (oracle/make_golden.py mixed); the restatement (block_forward's `adt` casts, the modulation dtype read off the weights) must equal
    them bit for bit, and the plan must differ from the bf16 plan's result by what a precision plan is worth (1e-4 .. 3e-2)."""
    g = load(f"forward_{name}_mixed.npz")
    f, h, w = [int(v) for v in g["shape"]]
    cfg = O.make_config(name)
    dtype = torch.bfloat16
    W = O.synth_weights(cfg, dtype=dtype, mixed=True)
    Wb = O.synth_weights(cfg, dtype=dtype)
    assert W["time_projection.1.weight"].dtype == torch.float32 and W["blocks.0.norm3.weight"].dtype == torch.float32
    assert W["blocks.0.self_attn.q.weight"].dtype == dtype and W["blocks.0.modulation"].dtype == dtype
    assert all(torch.equal(W[k].float(), Wb[k].float()) for k in W)          # the same values: the upcast of a bf16 checkpoint is exact
    lat, ctx, ctx_null, y = O.synth_inputs(cfg, f, h, w)
    tt = torch.tensor([int(g["t"][0])], dtype=torch.int64)
    # (round 6: the Wan2.1 i2v / flf2v CLIP branch under the plan -- no lock names img_emb or k_img / v_img: they stay bf16)
    clip = O.synth_clip_fea(images=2 if cfg.flf else 1) if cfg.model_type == "i2v" else None
    cond, uncond = O.dit_forward([lat, lat], tt, [ctx.to(dtype), ctx_null.to(dtype)], W, cfg, y=y, dtype=dtype, clip_fea=clip)
    assert torch.equal(cond, t(g["cond_mixed"])) and torch.equal(uncond, t(g["uncond_mixed"]))
    gb = load(f"forward_{name}.npz")
    rel = ((cond - t(gb["cond_bf16"])).norm() / t(gb["cond_bf16"]).norm()).item()
    assert 1e-4 < rel < 3e-2, rel
    if name == "tiny_ti2v":
        tf = torch.full((f,), int(g["t"][0]), dtype=torch.int64)
        tf[:1] = 0
        c2, u2 = O.dit_forward([lat, lat], tf, [ctx.to(dtype), ctx_null.to(dtype)], W, cfg, dtype=dtype)
        assert torch.equal(c2, t(g["cond_tframe_mixed"])) and torch.equal(u2, t(g["uncond_tframe_mixed"]))
    gen = torch.Generator().manual_seed(11)
    L = f * (h // 2) * (w // 2)
    hid = torch.randn(1, L, cfg.dim, generator=gen)
    e0 = 0.5 * torch.randn(1, 6, cfg.dim, generator=gen)
    cemb = (0.5 * torch.randn(1, 512, cfg.dim, generator=gen)).to(dtype)
    cos, sin = O.rope_tables((f, h // 2, w // 2))
    bo = O.block_forward(hid, e0, cemb, cos, sin, W, 0, cfg, adt=dtype)
    assert bo.dtype == torch.float32 and torch.equal(bo, t(g["block0_mixed"]))


def test_unipc_known_answer_timesteps():
    # BASELINE.md §2 known-answer from the reference run during the survey
    s = O.UniPCOracle()
    ts = s.set_timesteps(10, 5.0)
    assert ts.tolist() == [999, 978, 952, 920, 882, 833, 768, 681, 555, 356]


@pytest.mark.parametrize("steps,shift", [(10, 5.0), (30, 12.0), (4, 3.0)])
def test_unipc_matches_reference(steps, shift):
    g = load("sched.npz")
    s = O.UniPCOracle()
    ts = s.set_timesteps(steps, shift)
    assert np.array_equal(ts.numpy(), g[f"unipc_ts_{steps}_{shift}"])
    assert np.array_equal(s.sigmas.numpy(), g[f"unipc_sig_{steps}_{shift}"])
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(1, 16, 2, 4, 4, generator=gen)
    ref = g[f"unipc_trace_{steps}_{shift}"]
    for i in range(steps):
        v = torch.randn(x.shape, generator=gen) * 0.7 + 0.1 * x
        x = s.step(v, x)
        assert torch.allclose(x, t(ref[i]), atol=1e-6, rtol=1e-6), i


@pytest.mark.parametrize("steps,shift", [(10, 5.0), (4, 3.0)])
def test_euler_matches_reference(steps, shift):
    g = load("sched.npz")
    s = O.EulerOracle()
    ts = s.set_timesteps(steps, shift)
    assert np.array_equal(ts.numpy(), g[f"euler_ts_{steps}_{shift}"])
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(1, 16, 2, 4, 4, generator=gen)
    ref = g[f"euler_trace_{steps}_{shift}"]
    for i, tv in enumerate(ts):
        v = torch.randn(x.shape, generator=gen) * 0.7 + 0.1 * x
        x = s.step(v, tv, x)
        assert torch.equal(x, t(ref[i])), i


@pytest.mark.parametrize("tag,dtype", [("bf16", torch.bfloat16), ("fp32", torch.float32)])
def test_sampler_loop_matches_reference(tag, dtype):
    g = load("loop_tiny.npz")
    f, h, w = [int(v) for v in g["shape"]]
    cfg = O.make_config("tiny")
    lat, ctx, ctx_null, _ = O.synth_inputs(cfg, f, h, w, seed=5)
    W_hi = O.synth_weights(cfg, seed=1234, dtype=dtype)
    W_lo = O.synth_weights(cfg, seed=4321, dtype=dtype)
    _, trace = O.sample_loop(W_hi, cfg, lat, ctx.to(dtype), ctx_null.to(dtype), steps=3, shift=5.0, guide_scale=4.0,
                             W_lo=W_lo, switch_threshold=875, guide2_scale=3.0, dtype=dtype)
    ref = g["trace_bf16"]
    for i in range(3):
        if dtype == torch.bfloat16:
            assert torch.allclose(trace[i], t(ref[i]), atol=1e-6, rtol=1e-6), i
        else:
            rel = (trace[i] - t(ref[i])).norm() / t(ref[i]).norm()
            assert rel < 3e-2, (i, rel)


@pytest.mark.parametrize("steps,shift", [(10, 5.0), (4, 3.0), (20, 12.0)])
def test_dpmpp_matches_reference(steps, shift):
    g = load("sched2.npz")
    s = O.DpmppOracle()
    ts = s.set_timesteps(steps, shift)
    assert np.array_equal(ts.numpy(), g[f"dpm_{steps}_{shift}_ts"])
    assert np.array_equal(s.sigmas.numpy(), g[f"dpm_{steps}_{shift}_sig"])
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(1, 16, 2, 4, 4, generator=gen)
    ref = g[f"dpm_{steps}_{shift}_trace"]
    for i in range(steps):
        v = torch.randn(x.shape, generator=gen) * 0.7 + 0.1 * x
        x = s.step(v, x)
        assert torch.allclose(x, t(ref[i]), atol=1e-6, rtol=1e-6), i


@pytest.mark.parametrize("steps,shift", [(9, 7.0), (4, 5.0)])
def test_causvid_matches_reference(steps, shift):
    g = load("sched2.npz")
    s = O.FlowMatchOracle(steps, shift)
    s.timesteps = torch.tensor([1000, 934, 862, 756, 603, 410, 250, 140, 74])[:steps]
    s.sigmas = torch.cat([s.timesteps / 1000, torch.tensor([0.])])
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(1, 16, 2, 4, 4, generator=gen)
    ref = g[f"causvid_{steps}_{shift}_trace"]
    for i, tv in enumerate(s.timesteps):
        v = torch.randn(x.shape, generator=gen) * 0.7 + 0.1 * x
        x = s.step(v, tv, x)
        assert torch.equal(x, t(ref[i])), i


@pytest.mark.parametrize("steps,shift", [(4, 5.0), (8, 3.0)])
def test_lcm_matches_reference(steps, shift):
    g = load("sched2.npz")
    s = O.LcmOracle()
    ts = s.set_timesteps(steps, shift)
    assert np.array_equal(ts.numpy(), g[f"lcm_{steps}_{shift}_ts"])
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(1, 16, 2, 4, 4, generator=gen)
    ref = g[f"lcm_{steps}_{shift}_trace"]
    for i in range(len(ts)):
        v = torch.randn(x.shape, generator=gen) * 0.7 + 0.1 * x
        x = s.step(v, x)
        assert torch.equal(x, t(ref[i])), i
