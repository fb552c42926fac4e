"""-m gpu: WanModelHIP.forward, schedulers and the sampler loop against the golden fixtures
(outputs of the reference's own modules) and the CPU oracle.

Tolerances: the HIP path follows the reference's bf16 rounding points, so what remains is
bf16 accumulation noise.  We measure it against the fp32 anchor (oracle dtype=float32):
   err_hip = |hip - fp32| / |fp32|   must be <= 1.5 * err_ref + 2e-3,  err_ref = |ref_bf16 - fp32| / |fp32|
i.e. the HIP path is no further from the exact graph than the reference's own bf16 run is
(fixtures: tests/golden/forward_*.npz, produced by oracle/make_golden.py from the reference).
"""
import os
import types

import numpy as np
import pytest
import torch

from oracle import wan_oracle as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
BF = torch.bfloat16


def load(name):
    return dict(np.load(os.path.join(G, name)))


def build(cfg, seed=1234):
    from wan2gp_amd.model import WanModelHIP
    W = O.synth_weights(cfg, seed=seed)
    m = WanModelHIP(model_type=cfg.model_type, dim=cfg.dim, ffn_dim=cfg.ffn_dim, num_heads=cfg.num_heads,
                    num_layers=cfg.num_layers, in_dim=cfg.in_dim, out_dim=cfg.out_dim,
                    **({} if cfg.vace_layers is None else {"vace_layers": list(cfg.vace_layers), "vace_in_dim": cfg.vace_in_dim}))
    m.load_state_dict(W)
    return m, W


def rel(a, b):
    return ((a - b).norm() / b.norm()).item()


@pytest.mark.parametrize("name", ["tiny", "tiny_i2v", "tiny_ti2v", "tiny_i2v21", "tiny_flf2v", "tiny_vace"])
def test_forward_vs_reference_golden(name):
    g = load(f"forward_{name}.npz")
    f, h, w = [int(v) for v in g["shape"]]
    cfg = O.make_config(name)
    m, W = build(cfg)
    lat, ctx, ctx_null, y = O.synth_inputs(cfg, f, h, w)
    t = torch.tensor([int(g["t"][0])], dtype=torch.int64)
    xs = [lat.cuda(), lat.cuda()]
    # Wan2.1 i2v: CLIP tokens + k_img / v_img branch; flf2v: two images, position embedding, the second image in the text branch
    clip = O.synth_clip_fea(images=2 if cfg.flf else 1) if cfg.model_type == "i2v" else None
    kw = {} if clip is None else {"clip_fea": clip.cuda()}
    vace = O.synth_vace_context(cfg, f, h, w) if cfg.vace_layers is not None else None     # VACE context blocks
    if vace is not None:
        kw.update({"vace_context": [vace.cuda()], "vace_context_scale": [1.0]})
    outs = m(xs, t=t, context=[ctx.cuda(), ctx_null.cuda()], y=None if y is None else y.cuda(), **kw)
    assert xs == []                                                    # list consumed (model.py:1558-1559)
    W32 = O.synth_weights(cfg, dtype=torch.float32)
    anchor = O.dit_forward([lat, lat], t, [ctx.float(), ctx_null.float()], W32, cfg, y=y, dtype=torch.float32, exact=True,
                           clip_fea=None if clip is None else clip.float(), vace_context=vace)
    for o, key, a in zip(outs, ("cond_bf16", "uncond_bf16"), anchor):
        assert o.dtype == torch.float32 and tuple(o.shape) == (1, cfg.out_dim, f, h, w)
        ref = torch.from_numpy(g[key])
        err_ref, err_hip = rel(ref, a), rel(o.cpu(), a)
        print(f"{name}/{key}: err_ref={err_ref:.4e} err_hip={err_hip:.4e} hip-vs-ref={rel(o.cpu(), ref):.4e}")
        assert err_hip <= 1.5 * err_ref + 2e-3, (err_hip, err_ref)
        assert rel(o.cpu(), ref) <= 2.5e-2


def test_forward_small_config_vs_oracle():
    """4 heads / 3 layers / ragged token count (L = 3*5*7 = 105, not a multiple of any tile)."""
    cfg = O.make_config("small")
    m, W = build(cfg, seed=77)
    lat, ctx, ctx_null, _ = O.synth_inputs(cfg, 3, 10, 14, seed=9)
    t = torch.tensor([412])
    got = m([lat.cuda()], t=t, context=[ctx.cuda()])[0].cpu()
    ref = O.dit_forward([lat], t, [ctx], W, cfg, dtype=BF)[0]
    W32 = O.synth_weights(cfg, seed=77, dtype=torch.float32)
    anchor = O.dit_forward([lat], t, [ctx.float()], W32, cfg, dtype=torch.float32, exact=True)[0]
    err_ref, err_hip = rel(ref, anchor), rel(got, anchor)
    print(f"small: err_ref={err_ref:.4e} err_hip={err_hip:.4e}")
    assert err_hip <= 1.5 * err_ref + 2e-3


def test_interrupt_and_callback_contract():
    cfg = O.make_config("tiny")
    m, W = build(cfg)
    lat, ctx, ctx_null, _ = O.synth_inputs(cfg, 2, 8, 8)
    calls = []
    pipe = types.SimpleNamespace(_interrupt=False)
    out = m([lat.cuda()], t=torch.tensor([500]), context=[ctx.cuda()], pipeline=pipe,
            callback=lambda *a: calls.append(a))
    assert out[0] is not None and len(calls) == cfg.num_layers and calls[0] == (-1, None, False, True)
    pipe._interrupt = True
    out = m([lat.cuda(), lat.cuda()], t=torch.tensor([500]), context=[ctx.cuda(), ctx_null.cuda()], pipeline=pipe)
    assert out == [None, None]                                          # model.py:1997-1998
    with pytest.raises(NotImplementedError):
        m([lat.cuda()], t=torch.tensor([500]), context=[ctx.cuda()], vace_context=[lat.cuda()])


@pytest.mark.parametrize("steps,shift", [(10, 5.0), (30, 12.0), (4, 3.0)])
def test_unipc_vs_reference_golden(steps, shift):
    from wan2gp_amd.schedulers import FlowUniPCMultistepScheduler
    g = load("sched.npz")
    s = FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
    s.set_timesteps(steps, device="cuda", shift=shift)
    assert np.array_equal(s.timesteps.cpu().numpy(), g[f"unipc_ts_{steps}_{shift}"])
    assert np.array_equal(s.sigmas.numpy(), g[f"unipc_sig_{steps}_{shift}"])
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(1, 16, 2, 4, 4, generator=gen)
    xd = x.cuda()
    ref = g[f"unipc_trace_{steps}_{shift}"]
    for i, t in enumerate(s.timesteps):
        v = torch.randn(x.shape, generator=gen) * 0.7 + 0.1 * x      # same synthetic model output as the fixture
        xd = s.step(v.cuda(), t, xd, return_dict=False)[0]
        x = torch.from_numpy(ref[i])                                   # follow the reference trajectory
        assert torch.allclose(xd.cpu(), x, atol=2e-5, rtol=2e-5), (i, (xd.cpu() - x).abs().max())
        xd = x.cuda()


@pytest.mark.parametrize("steps,shift", [(10, 5.0), (4, 3.0)])
def test_euler_vs_reference_golden(steps, shift):
    from wan2gp_amd.schedulers import EulerScheduler
    g = load("sched.npz")
    s = EulerScheduler(num_train_timesteps=1000, use_timestep_transform=True)
    ts = s.set_timesteps(steps, device="cuda", shift=shift)
    assert np.array_equal(ts.numpy(), g[f"euler_ts_{steps}_{shift}"])
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(1, 16, 2, 4, 4, generator=gen)
    ref = g[f"euler_trace_{steps}_{shift}"]
    for i, t in enumerate(ts):
        v = torch.randn(x.shape, generator=gen) * 0.7 + 0.1 * x
        x = s.step(v.cuda(), t, x.cuda(), return_dict=False)[0].cpu()
        assert torch.allclose(x, torch.from_numpy(ref[i]), atol=1e-6, rtol=1e-6), i
        x = torch.from_numpy(ref[i])


@pytest.mark.parametrize("steps,shift", [(10, 5.0), (30, 12.0), (4, 3.0)])
def test_native_unipc_vs_reference_golden(steps, shift):
    """wan_sched_* (kind UniPC) through HipScheduler: timesteps / sigmas exactly the reference's, every step of the reference
    trajectory within 2e-5 -- the bar of the Python mirror above."""
    from wan2gp_amd.schedulers import HipScheduler
    g = load("sched.npz")
    s = HipScheduler("unipc", num_train_timesteps=1000)
    s.set_timesteps(steps, device="cuda", shift=shift)
    assert np.array_equal(s.timesteps.cpu().numpy(), g[f"unipc_ts_{steps}_{shift}"])
    assert np.array_equal(s.sigmas.numpy(), g[f"unipc_sig_{steps}_{shift}"])
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(1, 16, 2, 4, 4, generator=gen)
    xd = x.cuda()
    ref = g[f"unipc_trace_{steps}_{shift}"]
    for i, t in enumerate(s.timesteps):
        v = torch.randn(x.shape, generator=gen) * 0.7 + 0.1 * x
        xd = s.step(v.cuda(), t, xd, return_dict=False)[0]
        x = torch.from_numpy(ref[i])
        assert torch.allclose(xd.cpu(), x, atol=2e-5, rtol=2e-5), (i, (xd.cpu() - x).abs().max())
        xd = x.cuda()
    with pytest.raises(Exception, match="past the last"):
        s.step(v.cuda(), t, xd)
    s.set_timesteps(steps, device="cuda", shift=shift)                 # reset: the same object serves the next generate()
    assert s.step(v.cuda(), s.timesteps[0], xd)[0].shape == xd.shape


@pytest.mark.parametrize("steps,shift", [(10, 5.0), (4, 3.0)])
def test_native_euler_vs_reference_golden(steps, shift):
    from wan2gp_amd.schedulers import HipScheduler
    g = load("sched.npz")
    s = HipScheduler("euler", num_train_timesteps=1000)
    ts = s.set_timesteps(steps, device="cuda", shift=shift)
    assert np.array_equal(ts.numpy(), g[f"euler_ts_{steps}_{shift}"])
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(1, 16, 2, 4, 4, generator=gen)
    ref = g[f"euler_trace_{steps}_{shift}"]
    for i, t in enumerate(ts):
        v = torch.randn(x.shape, generator=gen) * 0.7 + 0.1 * x
        x = s.step(v.cuda(), t, x.cuda(), return_dict=False)[0].cpu()
        assert torch.allclose(x, torch.from_numpy(ref[i]), atol=1e-6, rtol=1e-6), i
        x = torch.from_numpy(ref[i])


def test_native_scheduler_rejects_bad_arguments():
    from wan2gp_amd.lib import WanHipError
    from wan2gp_amd.schedulers import HipScheduler
    s = HipScheduler("unipc")
    x = torch.zeros(1, 16, 1, 4, 4, device="cuda")
    with pytest.raises(ValueError):
        s.step(x, 999, x)                                              # set_timesteps not called
    s.set_timesteps(4, device="cuda", shift=3.0)
    with pytest.raises(WanHipError, match="not one of"):
        s.step(x, 123456, x)
    with pytest.raises(WanHipError):
        s.step(x.cpu(), int(s.timesteps[0]), x)                        # no CPU path
    with pytest.raises(NotImplementedError):
        HipScheduler("dpm++")


def _follow(ref, stepfn, timesteps, atol):
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(1, 16, 2, 4, 4, generator=gen)
    for i, tv in enumerate(timesteps):
        v = torch.randn(x.shape, generator=gen) * 0.7 + 0.1 * x
        got = stepfn(v.cuda(), tv, x.cuda()).cpu()
        x = torch.from_numpy(ref[i])                                   # follow the reference trajectory
        assert torch.allclose(got, x, atol=atol, rtol=atol), (i, (got - x).abs().max())


@pytest.mark.parametrize("steps,shift", [(10, 5.0), (4, 3.0), (20, 12.0)])
def test_dpmpp_vs_reference_golden(steps, shift):
    from wan2gp_amd.schedulers import FlowDPMSolverMultistepScheduler, get_sampling_sigmas, retrieve_timesteps
    g = load("sched2.npz")
    s = FlowDPMSolverMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
    ts, _ = retrieve_timesteps(s, device="cuda", sigmas=get_sampling_sigmas(steps, shift))
    assert np.array_equal(ts.cpu().numpy(), g[f"dpm_{steps}_{shift}_ts"])
    assert np.array_equal(s.sigmas.numpy(), g[f"dpm_{steps}_{shift}_sig"])
    _follow(g[f"dpm_{steps}_{shift}_trace"], lambda v, t, x: s.step(v, t, x, return_dict=False)[0], ts, 2e-5)


@pytest.mark.parametrize("steps,shift", [(9, 7.0), (4, 5.0)])
def test_causvid_vs_reference_golden(steps, shift):
    from wan2gp_amd.schedulers import FlowMatchScheduler
    g = load("sched2.npz")
    s = FlowMatchScheduler(num_inference_steps=steps, shift=shift, sigma_min=0, extra_one_step=True)
    s.timesteps = torch.tensor([1000, 934, 862, 756, 603, 410, 250, 140, 74])[:steps].cuda()
    s.sigmas = torch.cat([s.timesteps / 1000, torch.tensor([0.], device="cuda")])
    _follow(g[f"causvid_{steps}_{shift}_trace"], lambda v, t, x: s.step(v, t, x)[0], s.timesteps, 1e-6)


@pytest.mark.parametrize("steps,shift", [(4, 5.0), (8, 3.0)])
def test_lcm_vs_reference_golden(steps, shift):
    from wan2gp_amd.schedulers import LCMScheduler
    g = load("sched2.npz")
    s = LCMScheduler(num_train_timesteps=1000, num_inference_steps=min(steps, 8), shift=shift)
    s.set_timesteps(steps, device="cuda", shift=shift)
    assert np.array_equal(s.timesteps.cpu().numpy(), g[f"lcm_{steps}_{shift}_ts"])
    _follow(g[f"lcm_{steps}_{shift}_trace"], lambda v, t, x: s.step(v, t, x)[0], s.timesteps, 1e-6)


def test_sampler_loop_vs_reference_golden():
    """3 UniPC steps, CFG 4 -> 3, expert switch at t <= 875 (any2video.py:1437-1443)."""
    from wan2gp_amd.pipeline import WanAny2VHIP
    g = load("loop_tiny.npz")
    f, h, w = [int(v) for v in g["shape"]]
    cfg = O.make_config("tiny")
    m_hi, _ = build(cfg, 1234)
    m_lo, _ = build(cfg, 4321)
    lat, ctx, ctx_null, _ = O.synth_inputs(cfg, f, h, w, seed=5)
    trace = []
    pipe = WanAny2VHIP(m_hi, m_lo, device="cuda")
    out = pipe.generate(context=ctx.cuda(), context_null=ctx_null.cuda(), width=w * 8, height=h * 8,
                        frame_num=(f - 1) * 4 + 1, shift=5.0, sample_solver="unipc", sampling_steps=3, guide_scale=4.0,
                        guide2_scale=3.0, switch_threshold=875, guide_phases=2, model_switch_phase=1, latents=lat,
                        callback=lambda i, l, *a, **k: trace.append(l.detach().float().cpu().clone()) if i >= 0 else None,
                        return_latents=True)
    ref = g["trace_bf16"]
    assert len(trace) == 3
    for i in range(3):
        r = torch.from_numpy(ref[i])[0]
        e = rel(trace[i], r)
        print(f"loop step {i}: rel err vs reference {e:.4e}")
        assert e <= 4e-2, (i, e)
    assert torch.isfinite(out["latents"]).all()


def test_i2v21_requires_clip_features_and_changes_with_them():
    """model.py:1547 asserts clip_fea and y for model_type 'i2v'; the CLIP branch must actually contribute."""
    from wan2gp_amd.lib import WanHipError
    cfg = O.make_config("tiny_i2v21")
    m, W = build(cfg)
    lat, ctx, _, y = O.synth_inputs(cfg, 2, 8, 8)
    t = torch.tensor([500])
    with pytest.raises(WanHipError):
        m([lat.cuda()], t=t, context=[ctx.cuda()], y=y.cuda())
    a = m([lat.cuda()], t=t, context=[ctx.cuda()], y=y.cuda(), clip_fea=O.synth_clip_fea(1).cuda())[0]
    b = m([lat.cuda()], t=t, context=[ctx.cuda()], y=y.cuda(), clip_fea=O.synth_clip_fea(2).cuda())[0]
    assert torch.isfinite(a).all() and rel(a.cpu(), b.cpu()) > 1e-4
    cfg2 = O.make_config("tiny")
    m2, _ = build(cfg2)
    lat2, ctx2, _, _ = O.synth_inputs(cfg2, 2, 8, 8)
    with pytest.raises(NotImplementedError):
        m2([lat2.cuda()], t=t, context=[ctx2.cuda()], clip_fea=O.synth_clip_fea(1).cuda())      # t2v model: variant kwarg


def test_vace_scale_and_plain_paths():
    """vace_context_scale 0.6 (x.add_(hint, alpha)) against the reference golden; scale 0 and no context equal the plain
    forward of the same weights; a model without VACE blocks rejects the keyword."""
    g = load("forward_tiny_vace.npz")
    f, h, w = [int(v) for v in g["shape"]]
    cfg = O.make_config("tiny_vace")
    m, W = build(cfg)
    lat, ctx, ctx_null, _ = O.synth_inputs(cfg, f, h, w)
    t = torch.tensor([int(g["t"][0])], dtype=torch.int64)
    vace = O.synth_vace_context(cfg, f, h, w)
    run = lambda **kw: m([lat.cuda(), lat.cuda()], t=t, context=[ctx.cuda(), ctx_null.cuda()], **kw)
    o6 = run(vace_context=[vace.cuda()], vace_context_scale=[0.6])
    assert rel(o6[0].cpu(), torch.from_numpy(g["cond_s06_bf16"])) <= 2.5e-2 and rel(o6[1].cpu(), torch.from_numpy(g["uncond_s06_bf16"])) <= 2.5e-2
    o1 = run(vace_context=[vace.cuda()])
    assert rel(o1[0].cpu(), torch.from_numpy(g["cond_bf16"])) <= 2.5e-2 and rel(o6[0].cpu(), o1[0].cpu()) > 1e-3
    plain, zero = run(), run(vace_context=[vace.cuda()], vace_context_scale=[0.0])
    assert torch.equal(plain[0], zero[0]) and torch.equal(plain[1], zero[1]) and rel(plain[0].cpu(), o1[0].cpu()) > 1e-3
    m2, _ = build(O.make_config("tiny"))
    with pytest.raises(NotImplementedError):
        m2([lat.cuda()], t=t, context=[ctx.cuda()], vace_context=[vace.cuda()])


def test_per_frame_timesteps_vs_reference_golden():
    """ti2v image conditioning / diffusion forcing: t is a vector with one entry per latent frame (model.py:1812-1818; the
    reference zeroes the timestep of the injected source frames, any2video.py:1496-1499); tokens of frame f are modulated with
    e0[f].  Golden: the reference's own WanModel(model_type='ti2v2_2') forward with t = [0, 455] on a 2-frame latent."""
    g = load("forward_tiny_ti2v.npz")
    f, h, w = [int(v) for v in g["shape"]]
    cfg = O.make_config("tiny_ti2v")
    m, W = build(cfg)
    lat, ctx, ctx_null, _ = O.synth_inputs(cfg, f, h, w)
    tf = torch.full((f,), int(g["t"][0]), dtype=torch.int64)
    tf[:1] = 0
    outs = m([lat.cuda(), lat.cuda()], t=tf, context=[ctx.cuda(), ctx_null.cuda()])
    plain = m([lat.cuda(), lat.cuda()], t=torch.tensor([int(g["t"][0])]), context=[ctx.cuda(), ctx_null.cuda()])
    W32 = O.synth_weights(cfg, dtype=torch.float32)
    anchor = O.dit_forward([lat, lat], tf, [ctx.float(), ctx_null.float()], W32, cfg, dtype=torch.float32, exact=True)
    for o, p, key, a in zip(outs, plain, ("cond_tframe_bf16", "uncond_tframe_bf16"), anchor):
        ref = torch.from_numpy(g[key])
        err_ref, err_hip = rel(ref, a), rel(o.cpu(), a)
        print(f"per-frame t / {key}: err_ref={err_ref:.4e} err_hip={err_hip:.4e} hip-vs-ref={rel(o.cpu(), ref):.4e}")
        assert err_hip <= 1.5 * err_ref + 2e-3 and rel(o.cpu(), ref) <= 2.5e-2
        assert rel(o.cpu(), p.cpu()) > 1e-2                                  # frame 0 really saw t = 0
    # a [1, F] tensor (diffusion forcing) is the same call; a wrong length is an error
    o2 = m([lat.cuda()], t=tf.view(1, -1), context=[ctx.cuda()])[0]
    assert torch.equal(o2.cpu(), outs[0].cpu())
    from wan2gp_amd.lib import WanHipError
    with pytest.raises(WanHipError):
        m([lat.cuda()], t=torch.tensor([1, 2, 3]), context=[ctx.cuda()])


@pytest.mark.parametrize("name", ["tiny", "tiny_i2v", "small"])
def test_forward_as_replayed_launch_list_is_bit_identical(name):
    """wan_dit_forward_graph (round 5; WanModelHIP.graph): first sight eager, second sight captured, then replayed -- at every stage and
    for every timestep bit-identical to the eager forward (graph = "off") of the same inputs; the joint pass and a single stream are keys
    of their own; a callback keeps the per-block contract (eager)."""
    cfg = O.make_config(name)
    m, W = build(cfg)
    f, h, w = (3, 8, 12) if name != "small" else (3, 10, 14)
    lat, ctx, ctx_null, y = O.synth_inputs(cfg, f, h, w)
    cc, cn = ctx.cuda(), ctx_null.cuda()
    yy = None if y is None else y.cuda()
    g = torch.Generator().manual_seed(3)
    hows = []
    for i, tv in enumerate((900, 637, 412, 55, 999)):
        x = (lat + 0.1 * i * torch.randn(lat.shape, generator=g)).cuda()
        t = torch.tensor([tv])
        m.graph = "off"
        ref = m([x.clone(), x.clone()], t=t, context=[cc, cn], y=yy)
        assert m.last_graph_how == 0
        m.graph = "on"
        got = m([x.clone(), x.clone()], t=t, context=[cc, cn], y=yy)
        hows.append(m.last_graph_how)
        for a, b in zip(got, ref):
            assert torch.equal(a, b), (name, i, m.last_graph_how, (a - b).abs().max().item())
    assert hows == [1, 2, 3, 3, 3], hows
    one = m([lat.cuda()], t=torch.tensor([500]), context=[cc], y=yy)
    assert m.last_graph_how == 1                                               # S = 1: its own key
    m.graph = "off"
    assert torch.equal(m([lat.cuda()], t=torch.tensor([500]), context=[cc], y=yy)[0], one[0])
    m.graph = "auto"                                                           # tiny shapes are below graph_max_tokens ...
    calls = []
    m([lat.cuda()], t=torch.tensor([500]), context=[cc], y=yy, callback=lambda *a: calls.append(a))
    assert m.last_graph_how == 0 and len(calls) == cfg.num_layers              # ... but a per-block callback keeps its contract
    m.graph_max_tokens = 1
    m([lat.cuda()], t=torch.tensor([500]), context=[cc], y=yy)
    assert m.last_graph_how == 0


def test_generate_with_replayed_forwards_equals_eager_generate():
    """The sampler loop over replayed forwards (what bench.py's configs[0] block times): same seed, same latents, bit for bit."""
    from wan2gp_amd.pipeline import WanAny2VHIP
    cfg = O.make_config("tiny")
    m, W = build(cfg)
    _, ctx, ctx_null, _ = O.synth_inputs(cfg, 2, 8, 8)
    pipe = WanAny2VHIP(m)
    run = lambda: pipe.generate(context=ctx.cuda(), context_null=ctx_null.cuda(), width=64, height=64, frame_num=5, sampling_steps=6,
                                guide_scale=3.0, seed=11, return_latents=True)["latents"].cpu()
    m.graph = "off"
    base = run()
    m.graph = "on"
    fast = run()
    assert m.last_graph_how == 3 and torch.equal(fast, base)


@pytest.mark.parametrize("name", ["tiny", "tiny_i2v", "tiny_i2v21", "tiny_flf2v", "tiny_vace", "small"])
def test_text_cache_is_bit_identical_and_follows_the_context(name):
    """The text cache (round 6; wan_dit_args.context_key, WanModelHIP.text_cache): cross-attention K / V^T and the text embedding of an
    unchanged prompt are kept across forwards.  Same kernels on the same inputs -> every forward equals the text_cache = False forward
    bit for bit: the filling call, the calls that read the cache (other latents, other timesteps), a call after the context was
    written IN PLACE (torch's version counter: recomputed), after another context object arrived, and with the two streams swapped."""
    cfg = O.make_config(name)
    m, W = build(cfg)
    f, h, w = (3, 8, 12) if name != "small" else (3, 10, 14)
    lat, ctx, ctx_null, y = O.synth_inputs(cfg, f, h, w)
    yy = None if y is None else y.cuda()
    clip = O.synth_clip_fea(images=2 if cfg.flf else 1) if cfg.model_type == "i2v" else None
    kw = {} if clip is None else {"clip_fea": clip.cuda()}
    vace = O.synth_vace_context(cfg, f, h, w) if cfg.vace_layers is not None else None
    if vace is not None:
        kw.update({"vace_context": [vace.cuda()], "vace_context_scale": [0.7]})
    m.graph = "off"
    g = torch.Generator().manual_seed(5)
    cc, cn = ctx.cuda().clone(), ctx_null.cuda().clone()

    def both(x, t, contexts):
        m.text_cache = False
        ref = m([x.clone() for _ in contexts], t=t, context=list(contexts), y=yy, **kw)
        m.text_cache = True
        got = m([x.clone() for _ in contexts], t=t, context=list(contexts), y=yy, **kw)
        for a, b in zip(got, ref):
            assert torch.equal(a, b), (name, (a - b).abs().max().item())
    for i, tv in enumerate((900, 637, 55)):                       # fill, then two hits
        both((lat + 0.1 * i * torch.randn(lat.shape, generator=g)).cuda(), torch.tensor([tv]), (cc, cn))
    k1 = m._tc_keys[0][2]
    cc.mul_(0.5)                                                   # the prompt changes in place: a new key, recomputed
    both(lat.cuda(), torch.tensor([500]), (cc, cn))
    assert m._tc_keys[0][2] != k1
    both(lat.cuda(), torch.tensor([400]), (cn, cc))                # the streams swapped: another key
    both(lat.cuda(), torch.tensor([300]), (cc,))                   # one stream
    both(lat.cuda(), torch.tensor([300]), (cc,))
    c3 = (ctx.cuda() * 0.25).contiguous()                          # a third context object: evicts the least recently used key
    both(lat.cuda(), torch.tensor([200]), (c3, cn))
    both(lat.cuda(), torch.tensor([100]), (c3, cn))
