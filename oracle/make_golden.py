"""TEST INFRASTRUCTURE ONLY -- generate tests/golden/*.npz from the REFERENCE's own code.

Run in the build container (needs /root/reference):   python oracle/make_golden.py
Every fixture is produced by executing the reference's unmodified modules on CPU through
``oracle/ref_shim.py`` with seeded synthetic weights/inputs (oracle.wan_oracle.synth_*),
so the files under tests/golden/ are outputs *of the reference*, not of our restatement.
tests/test_oracle_vs_golden.py then pins oracle/wan_oracle.py to them; the -m gpu tests
compare the HIP path with the oracle and, for the stored cases, with these files.

Inputs are not stored -- they are re-derived from the seeds recorded in each file.
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim, wan_oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def f32(t):
    return t.detach().to(torch.float32).cpu().numpy()


def build_ref_model(ns, cfg: O.WanConfig, W, dtype, mixed=False):
    """Construct the reference WanModel and load the synthetic checkpoint with the
    reference's dtype locks (model.py:1330-1371): patch_embedding + head fp32."""
    m = ns.M.WanModel(model_type=cfg.model_type, dim=cfg.dim, ffn_dim=cfg.ffn_dim, num_heads=cfg.num_heads,
                      num_layers=cfg.num_layers, in_dim=cfg.in_dim, out_dim=cfg.out_dim, text_dim=cfg.text_dim,
                      freq_dim=cfg.freq_dim, eps=cfg.eps, **({"flf": True} if cfg.flf else {}),
                      **({} if cfg.vace_layers is None else {"vace_layers": list(cfg.vace_layers), "vace_in_dim": cfg.vace_in_dim}))
    sd = {k: v.clone() for k, v in W.items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("modulation" not in k for k in missing), missing
    m.apply_post_init_changes()          # modulation Parameter -> sub-module .weight (model.py:1291-1328)
    m.eval()
    if dtype != torch.float32:
        for name, p in m.named_parameters():
            if O.is_fp32_locked(name, mixed):      # mixed: + layer_list2 of lock_layers_dtypes (model.py:1338-1345), what mmgp keeps in fp32
                continue                            # for `mixed_precision_transformer` (any2video.py:190)
            p.data = p.data.to(dtype)
    return m


def ref_forward(ns, m, x_list, t, ctx_list, y=None, clip_fea=None, vace=None, vace_scale=1.0):
    grid = x_list[0].shape[2:]
    freqs = ns.P.get_rotary_pos_embed(grid)
    kw = {} if clip_fea is None else {"clip_fea": clip_fea.clone()}
    if vace is not None:
        kw.update({"vace_context": [vace.clone()], "vace_context_scale": [vace_scale]})
    with torch.no_grad():
        return m([x.clone() for x in x_list], t=t, context=[c.clone() for c in ctx_list], y=y, freqs=freqs,
                 pipeline=types.SimpleNamespace(_interrupt=False), **kw)


def gen_ops(ns):
    """Per-op goldens: RoPE tables, RMSNorm, RoPE apply, LayerNorm(+affine), sdpa."""
    out = {}
    cos, sin = ns.P.get_rotary_pos_embed((3, 8, 12))          # latent f,h,w -> grid (3,4,6)
    out["rope_cos_3x4x6"], out["rope_sin_3x4x6"] = f32(cos), f32(sin)
    rc, rs = ns.P.get_rotary_pos_embed((33, 4, 6), enable_RIFLEx=True)        # long video: RIFLEx on the time axis
    out["riflex_cos_33x2x3"], out["riflex_sin_33x2x3"] = f32(rc), f32(rs)
    g = torch.Generator().manual_seed(7)
    L, H, D = 72, 2, 128
    x = torch.randn(1, L, H * D, generator=g).to(torch.bfloat16)
    w = (1 + 0.02 * torch.randn(H * D, generator=g)).to(torch.bfloat16)
    n = ns.M.WanRMSNorm(H * D, eps=1e-6)
    n.weight.data = w.clone()
    q = n(x.clone())
    out["rms_bf16"] = f32(q)
    k = n(torch.flip(x, dims=[1]).clone())
    qq, kk = ns.P.apply_rotary_emb([q.view(1, L, H, D).clone(), k.view(1, L, H, D).clone()], (cos, sin))
    out["rope_q_bf16"], out["rope_k_bf16"] = f32(qq), f32(kk)
    ln = ns.M.WanLayerNorm(H * D, 1e-6)
    out["ln_bf16"] = f32(ln(x))
    ln3 = ns.M.WanLayerNorm(H * D, 1e-6, elementwise_affine=True)
    ln3.weight.data = w.clone(); ln3.bias.data = (0.01 * torch.randn(H * D, generator=g)).to(torch.bfloat16)
    out["ln3_bias"] = f32(ln3.bias.data)
    out["ln3_bf16"] = f32(ln3(x))
    v = torch.randn(1, L, H, D, generator=g).to(torch.bfloat16)
    out["sdpa_bf16"] = f32(ns.A.pay_attention([qq.clone(), kk.clone(), v.clone()]))
    out["seed"] = np.array([7])
    np.savez_compressed(os.path.join(OUT, "ops.npz"), **out)
    print("ops.npz", {k: v.shape for k, v in out.items()})


def gen_forward(ns, name, f, h, w, tval):
    cfg = O.make_config(name)
    out = {"shape": np.array([f, h, w]), "t": np.array([tval])}
    lat, ctx, ctx_null, y = O.synth_inputs(cfg, f, h, w)
    clip = O.synth_clip_fea(images=2 if cfg.flf else 1) if cfg.model_type == "i2v" else None
    vace = O.synth_vace_context(cfg, f, h, w) if cfg.vace_layers is not None else None
    t = torch.tensor([tval], dtype=torch.int64)
    # NOTE: only the reference's real bf16 plan is a valid golden.  Run "fp32 everywhere" the
    # reference's WanRMSNorm aliases its input (`y = x.float()` is x itself for fp32, then
    # `y.pow_(2)` squares x in place, model.py:165-166), a path the real pipeline never takes
    # (q/k are always bf16: attention_dtype = self_attn.q.weight.dtype, model.py:614).
    for tag, dtype in (("bf16", torch.bfloat16),):
        W = O.synth_weights(cfg, dtype=dtype)
        m = build_ref_model(ns, cfg, W, dtype)
        cdt = dtype
        r = ref_forward(ns, m, [lat, lat], t, [ctx.to(cdt), ctx_null.to(cdt)], y=y, clip_fea=clip, vace=vace)
        out[f"cond_{tag}"], out[f"uncond_{tag}"] = f32(r[0]), f32(r[1])
        if name == "tiny_ti2v":                                 # per-frame timesteps: ti2v timestep injection (any2video.py:1496-1499)
            tf = torch.full((f,), tval, dtype=torch.int64)
            tf[:1] = 0
            r = ref_forward(ns, m, [lat, lat], tf, [ctx.to(cdt), ctx_null.to(cdt)])
            out[f"cond_tframe_{tag}"], out[f"uncond_tframe_{tag}"] = f32(r[0]), f32(r[1])
        if vace is not None:                                    # a second run with a fractional context scale (x.add_(hint, alpha))
            r = ref_forward(ns, m, [lat, lat], t, [ctx.to(cdt), ctx_null.to(cdt)], y=y, vace=vace, vace_scale=0.6)
            out[f"cond_s06_{tag}"], out[f"uncond_s06_{tag}"] = f32(r[0]), f32(r[1])
        # one block in isolation (block 0) on a seeded hidden state
        g = torch.Generator().manual_seed(11)
        L = f * (h // 2) * (w // 2)
        hid = torch.randn(1, L, cfg.dim, generator=g).to(dtype)
        e0 = (0.5 * torch.randn(1, 6, cfg.dim, generator=g)).to(dtype)
        cemb = (0.5 * torch.randn(1, 512 + (O.CLIP_TOKENS * (2 if cfg.flf else 1) if cfg.model_type == "i2v" else 0), cfg.dim, generator=g)).to(dtype)
        freqs = ns.P.get_rotary_pos_embed((f, h, w))
        with torch.no_grad():
            bo = m.blocks[0](hid.clone(), e=e0, grid_sizes=(f, h // 2, w // 2), freqs=freqs, context=cemb)
        out[f"block0_{tag}"] = f32(bo)
    np.savez_compressed(os.path.join(OUT, f"forward_{name}.npz"), **out)
    print(f"forward_{name}.npz", {k: v.shape for k, v in out.items()})


def gen_forward_mixed(ns, name, f, h, w, tval):
    """The reference's forward under `mixed_precision_transformer` (wgp.py:4039 -> any2video.py:190 -> model.py:1330-1371): the time MLP,
    the time projection and every block's norm3 hold their (bf16-valued) weights in fp32; by type promotion the residual stream, e / e0
    and every modulate / gated residual then run in fp32 between bf16 Linears.  Same seeds and weights as forward_<name>.npz."""
    cfg = O.make_config(name)
    out = {"shape": np.array([f, h, w]), "t": np.array([tval])}
    lat, ctx, ctx_null, y = O.synth_inputs(cfg, f, h, w)
    t = torch.tensor([tval], dtype=torch.int64)
    dtype = torch.bfloat16
    W = O.synth_weights(cfg, dtype=dtype, mixed=True)
    m = build_ref_model(ns, cfg, W, dtype, mixed=True)
    assert m.time_projection[1].weight.dtype == torch.float32 and m.blocks[0].norm3.weight.dtype == torch.float32
    assert m.blocks[0].self_attn.q.weight.dtype == dtype and m.text_embedding[0].weight.dtype == dtype
    # Wan2.1 i2v / flf2v: the CLIP tokens (img_emb, k_img / v_img stay in the checkpoint's dtype: no lock names them, model.py:1330-1371)
    clip = O.synth_clip_fea(images=2 if cfg.flf else 1) if cfg.model_type == "i2v" else None
    r = ref_forward(ns, m, [lat, lat], t, [ctx.to(dtype), ctx_null.to(dtype)], y=y, clip_fea=clip)
    out["cond_mixed"], out["uncond_mixed"] = f32(r[0]), f32(r[1])
    if clip is not None:
        assert m.img_emb.proj[1].weight.dtype == dtype and m.blocks[0].cross_attn.k_img.weight.dtype == dtype
    if name == "tiny_ti2v":                                     # per-frame timesteps
        tf = torch.full((f,), tval, dtype=torch.int64)
        tf[:1] = 0
        r = ref_forward(ns, m, [lat, lat], tf, [ctx.to(dtype), ctx_null.to(dtype)])
        out["cond_tframe_mixed"], out["uncond_tframe_mixed"] = f32(r[0]), f32(r[1])
    g = torch.Generator().manual_seed(11)
    L = f * (h // 2) * (w // 2)
    hid = torch.randn(1, L, cfg.dim, generator=g)                # the residual stream is fp32 in this plan
    e0 = 0.5 * torch.randn(1, 6, cfg.dim, generator=g)
    cemb = (0.5 * torch.randn(1, 512, cfg.dim, generator=g)).to(dtype)
    freqs = ns.P.get_rotary_pos_embed((f, h, w))
    with torch.no_grad():
        bo = m.blocks[0](hid.clone(), e=e0, grid_sizes=(f, h // 2, w // 2), freqs=freqs, context=cemb)
    assert bo.dtype == torch.float32
    out["block0_mixed"] = f32(bo)
    np.savez_compressed(os.path.join(OUT, f"forward_{name}_mixed.npz"), **out)
    print(f"forward_{name}_mixed.npz", {k: v.shape for k, v in out.items()})


def gen_sched(ns):
    out = {}
    s = ns.U.FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
    for steps, shift in ((10, 5.0), (30, 12.0), (4, 3.0)):
        s.set_timesteps(steps, device="cpu", shift=shift)
        out[f"unipc_ts_{steps}_{shift}"] = s.timesteps.numpy().copy()
        out[f"unipc_sig_{steps}_{shift}"] = s.sigmas.numpy().copy()
        g = torch.Generator().manual_seed(3)
        x = torch.randn(1, 16, 2, 4, 4, generator=g)
        trace = []
        for i, t in enumerate(s.timesteps):
            v = torch.randn(x.shape, generator=g) * 0.7 + 0.1 * x      # synthetic "model output"
            x = s.step(v, t, x, return_dict=False)[0]
            trace.append(f32(x))
        out[f"unipc_trace_{steps}_{shift}"] = np.stack(trace)
    e = ns.E.EulerScheduler(num_train_timesteps=1000, use_timestep_transform=True)
    for steps, shift in ((10, 5.0), (4, 3.0)):
        ts = e.set_timesteps(steps, device="cpu", shift=shift)
        out[f"euler_ts_{steps}_{shift}"] = ts.numpy().copy()
        g = torch.Generator().manual_seed(3)
        x = torch.randn(1, 16, 2, 4, 4, generator=g)
        trace = []
        for t in ts:
            v = torch.randn(x.shape, generator=g) * 0.7 + 0.1 * x
            x = e.step(v, t, x, return_dict=False)[0]
            trace.append(f32(x))
        out[f"euler_trace_{steps}_{shift}"] = np.stack(trace)
    np.savez_compressed(os.path.join(OUT, "sched.npz"), **out)
    print("sched.npz", {k: v.shape for k, v in out.items()})


def gen_sched2(ns):
    """dpm++ / causvid / lcm schedulers exactly as WanAny2V.generate builds them (any2video.py:513-545)."""
    out = {}

    def run(name, sched, timesteps, stepfn):
        g = torch.Generator().manual_seed(3)
        x = torch.randn(1, 16, 2, 4, 4, generator=g)
        trace = []
        for t in timesteps:
            v = torch.randn(x.shape, generator=g) * 0.7 + 0.1 * x
            x = stepfn(v, t, x)
            trace.append(f32(x))
        out[name + "_trace"] = np.stack(trace)
        out[name + "_ts"] = np.asarray(timesteps.numpy() if torch.is_tensor(timesteps) else timesteps).copy()

    for steps, shift in ((10, 5.0), (4, 3.0), (20, 12.0)):
        s = ns.D.FlowDPMSolverMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
        ts, _ = ns.D.retrieve_timesteps(s, device="cpu", sigmas=ns.D.get_sampling_sigmas(steps, shift))
        run(f"dpm_{steps}_{shift}", s, ts, lambda v, t, x: s.step(v, t, x, return_dict=False)[0])
        out[f"dpm_{steps}_{shift}_sig"] = s.sigmas.numpy().copy()
    for steps, shift in ((9, 7.0), (4, 5.0)):
        s = ns.FM.FlowMatchScheduler(num_inference_steps=steps, shift=shift, sigma_min=0, extra_one_step=True)
        ts = torch.tensor([1000, 934, 862, 756, 603, 410, 250, 140, 74])[:steps]
        s.timesteps = ts
        s.sigmas = torch.cat([s.timesteps / 1000, torch.tensor([0.])])
        run(f"causvid_{steps}_{shift}", s, ts, lambda v, t, x: s.step(v, t, x)[0])
    for steps, shift in ((4, 5.0), (8, 3.0)):
        s = ns.LCM.LCMScheduler(num_train_timesteps=1000, num_inference_steps=min(steps, 8), shift=shift)
        s.set_timesteps(steps, device="cpu", shift=shift)
        run(f"lcm_{steps}_{shift}", s, s.timesteps, lambda v, t, x: s.step(v, t, x)[0])
    np.savez_compressed(os.path.join(OUT, "sched2.npz"), **out)
    print("sched2.npz", {k: v.shape for k, v in out.items()})


def gen_loop(ns):
    """3-step t2v sampler loop, two experts, CFG, UniPC -- the loop body of
    WanAny2V.generate (any2video.py:1470,1490-1501,1626-1634,1702-1722,1733) driven on the
    reference's own WanModel + scheduler objects."""
    cfg = O.make_config("tiny")
    f, h, w = 2, 8, 8
    lat, ctx, ctx_null, _ = O.synth_inputs(cfg, f, h, w, seed=5)
    out = {"shape": np.array([f, h, w])}
    for tag, dtype in (("bf16", torch.bfloat16),):
        m_hi = build_ref_model(ns, cfg, O.synth_weights(cfg, seed=1234, dtype=dtype), dtype)
        m_lo = build_ref_model(ns, cfg, O.synth_weights(cfg, seed=4321, dtype=dtype), dtype)
        s = ns.U.FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
        s.set_timesteps(3, device="cpu", shift=5.0)
        latents = lat.clone()
        trans, g, switched = m_hi, 4.0, False
        trace = []
        for i, t in enumerate(s.timesteps):
            if not switched and t <= 875:
                trans, g, switched = m_lo, 3.0, True
            cond, uncond = ref_forward(ns, trans, [latents, latents], torch.stack([t]),
                                       [ctx.to(dtype), ctx_null.to(dtype)])
            noise = uncond + g * (cond - uncond)
            latents = s.step(noise, t, latents, return_dict=False)[0]
            trace.append(f32(latents))
        out[f"trace_{tag}"] = np.stack(trace)
        out["timesteps"] = s.timesteps.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "loop_tiny.npz"), **out)
    print("loop_tiny.npz", {k: v.shape for k, v in out.items()})


def main():
    torch.manual_seed(0)
    os.makedirs(OUT, exist_ok=True)
    ns = ref_shim.load()
    which = sys.argv[1:] or ["ops", "forward", "sched", "sched2", "loop", "vae"]
    if "ops" in which:
        gen_ops(ns)
    if "forward" in which:
        gen_forward(ns, "tiny", 3, 8, 12, 637)
        gen_forward(ns, "tiny_i2v", 2, 8, 8, 912)
        gen_forward(ns, "tiny_ti2v", 2, 6, 10, 455)
        gen_forward(ns, "tiny_i2v21", 2, 8, 8, 731)
        gen_forward(ns, "tiny_vace", 2, 8, 8, 588)
        gen_forward(ns, "small", 3, 10, 14, 412)            # 4 heads, 3 layers, ragged token count L = 105
    if "forward" in which or "flf2v" in which:
        gen_forward(ns, "tiny_flf2v", 2, 8, 8, 644)          # Wan2.1 flf2v: two images' CLIP tokens + position embedding
    if "mixed" in which:                                     # (its own keyword: the files above are not regenerated by it)
        gen_forward_mixed(ns, "tiny", 3, 8, 12, 637)
        gen_forward_mixed(ns, "tiny_i2v", 2, 8, 8, 912)      # i2v2_2: y (mask + latents) concatenated in front of the patch embedding
        gen_forward_mixed(ns, "tiny_ti2v", 2, 6, 10, 455)    # + per-frame timesteps: e0 [frames, 6, dim] in fp32
        gen_forward_mixed(ns, "small", 3, 10, 14, 412)
    if "mixed_clip" in which:                                # round 6: the mixed plan with the Wan2.1 i2v CLIP branch (one image, and flf2v's two)
        gen_forward_mixed(ns, "tiny_i2v21", 2, 8, 8, 731)
        gen_forward_mixed(ns, "tiny_flf2v", 2, 8, 8, 644)
    if "sched" in which:
        gen_sched(ns)
    if "sched2" in which:
        gen_sched2(ns)
    if "loop" in which:
        gen_loop(ns)
    if "vae_endframe" in which:
        from oracle import make_golden_vae
        make_golden_vae.main_endframe(ns, OUT)
    if "vae" in which:
        try:
            from oracle import make_golden_vae
            make_golden_vae.main(ns, OUT)
        except ImportError:
            print("vae goldens: generator not present yet")


if __name__ == "__main__":
    main()
