"""-m gpu: the whole path end to end on tiny models -- sampler loop + VAE decode to uint8 (t2v), and the
i2v2_2 path (VAE encode of the start image -> y conditioning -> 36-channel DiT -> decode), checked against
the oracle loop driven with the same conditioning."""
import pytest
import torch

from oracle import vae_oracle as VO
from oracle import wan_oracle as O

pytestmark = pytest.mark.gpu


def build(cfg, seed=1234):
    from wan2gp_amd.model import WanModelHIP
    W = O.synth_weights(cfg, seed=seed)
    m = WanModelHIP(model_type=cfg.model_type, dim=cfg.dim, ffn_dim=cfg.ffn_dim, num_heads=cfg.num_heads,
                    num_layers=cfg.num_layers, in_dim=cfg.in_dim)
    return m.load_state_dict(W), W


def test_t2v_generate_to_uint8():
    from wan2gp_amd.pipeline import WanAny2VHIP
    from wan2gp_amd.vae import WanVAEHIP
    cfg = O.make_config("tiny")
    m, W = build(cfg)
    Wv = VO.synth_vae_weights()
    vae = WanVAEHIP(state_dict=Wv)
    f, h, w = 2, 8, 8
    lat, ctx, ctx_null, _ = O.synth_inputs(cfg, f, h, w, seed=5)
    pipe = WanAny2VHIP(m, vae=vae)
    out = pipe.generate(context=ctx.cuda(), context_null=ctx_null.cuda(), width=w * 8, height=h * 8, frame_num=(f - 1) * 4 + 1,
                        shift=5.0, sampling_steps=2, guide_scale=4.0, latents=lat, sample_solver="euler")
    vid = out["x"]
    assert vid.dtype == torch.uint8 and tuple(vid.shape) == (3, 5, 64, 64) and vid.device.type == "cpu"
    # oracle: same loop (bf16 plan) then the fp32 VAE
    latents, _ = O.sample_loop(W, cfg, lat, ctx, ctx_null, steps=2, shift=5.0, guide_scale=4.0, solver="euler")
    rel = ((out["latents"].cpu() - latents).norm() / latents.norm()).item()
    assert rel < 3e-2, rel
    ref = VO.float_to_uint8(VO.vae_decode(out["latents"].cpu(), Wv, VO.default_scale()))[0]
    d = (vid.int() - ref.int()).abs()
    assert int(d.max()) <= 2 and (d == 0).float().mean().item() > 0.9


def test_i2v_generate_conditioning():
    from wan2gp_amd.pipeline import WanAny2VHIP
    from wan2gp_amd.vae import WanVAEHIP
    cfg = O.make_config("tiny_i2v")
    m, W = build(cfg)
    Wv = VO.synth_vae_weights()
    vae = WanVAEHIP(state_dict=Wv)
    f, h, w = 2, 8, 8
    H, Wd, frames = h * 8, w * 8, (f - 1) * 4 + 1
    g = torch.Generator().manual_seed(3)
    img = torch.rand(3, H, Wd, generator=g) * 2 - 1
    pipe = WanAny2VHIP(m, vae=vae)
    y, ext = pipe.build_i2v_conditioning(img, frames, H, Wd)
    assert tuple(y.shape) == (20, f, h, w) and tuple(ext.shape) == (1, 16, 1, h, w)
    # reference construction (any2video.py:739-774) on the oracle VAE
    enc = torch.cat([img.unsqueeze(1), torch.zeros(3, frames - 1, H, Wd)], dim=1)
    lat_y = VO.vae_encode(enc.unsqueeze(0), Wv, VO.default_scale())[0]
    msk = torch.ones(1, frames, h, w); msk[:, 1:] = 0
    msk = torch.cat([torch.repeat_interleave(msk[:, 0:1], repeats=4, dim=1), msk[:, 1:]], dim=1)
    msk = msk.view(1, msk.shape[1] // 4, 4, h, w).transpose(1, 2)[0]
    y_ref = torch.cat([msk, lat_y])
    assert torch.equal(y[:4].cpu(), y_ref[:4])
    assert (y[4:].cpu() - y_ref[4:]).abs().max().item() <= 1e-2 * y_ref[4:].abs().max().item() + 1e-3
    lat, ctx, ctx_null, _ = O.synth_inputs(cfg, f, h, w, seed=5)
    out = pipe.generate(context=ctx.cuda(), context_null=ctx_null.cuda(), width=Wd, height=H, frame_num=frames, shift=5.0,
                        sampling_steps=2, guide_scale=4.0, latents=lat, image_start=img)
    assert out["x"].dtype == torch.uint8 and tuple(out["x"].shape) == (3, frames, H, Wd)
    # the known first latent frame is restored at the end (any2video.py:1755-1756)
    assert torch.equal(out["latents"][:, :, :1], ext)


def test_family_handler_load_model_builds_a_working_pipeline():
    """`family_handler.load_model` (the reference's plugin entry, models/wan/wan_handler.py:1116-1158) with in-memory state
    dicts standing in for the checkpoint files: a 1.3B DiT + the Wan2.1 VAE, then generate() -> uint8 frames on the host."""
    from oracle import wan_oracle as O
    from wan2gp_amd.vae import random_vae_state_dict
    from wan2gp_amd.wan_handler import family_handler as H
    cfg = O.make_config("t2v_1.3B")
    pipe, extra = H.load_model(["unused.safetensors"], "t2v_1.3B_hip", "t2v_1.3B_hip", {}, state_dicts=[O.synth_weights(cfg)],
                               vae_state_dict=random_vae_state_dict())
    assert extra == {"pipe": {}} and pipe.model is not None and pipe.model2 is None and pipe.vae is not None
    lat, ctx, ctx_null, _ = O.synth_inputs(cfg, 2, 8, 8)
    out = pipe.generate(context=ctx.cuda(), context_null=ctx_null.cuda(), width=64, height=64, frame_num=5, sampling_steps=2,
                        guide_scale=5.0, seed=1)
    assert out["x"].dtype == torch.uint8 and tuple(out["x"].shape) == (3, 5, 64, 64) and not out["x"].is_cuda
