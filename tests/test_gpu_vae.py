"""-m gpu: the HIP causal 3D VAE against the CPU oracle (oracle/vae_oracle.py, pinned bit-exactly to the
reference) and the reference-generated fixture tests/golden/vae_small.npz.

Tolerances: the HIP VAE stores activations in fp16 (the reference's default VAE dtype, wgp.py:4038) and
accumulates in fp32; the oracle runs the same graph in fp32.  Per-op: |err| <= 2e-3 * max|ref| (fp16
rounding of inputs/outputs).  End to end the check is on the INTEGER pixel output (BASELINE north star):
max |delta| <= 1 LSB, >= 90% of the bytes identical, mean |delta| <= 0.1 LSB.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import vae_oracle as VO

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
F16 = torch.float16


def cl(x):
    """[1,C,T,H,W] fp32 -> channels-last fp16 [T,H,W,C] on the GPU."""
    return x[0].permute(1, 2, 3, 0).contiguous().to(F16).cuda()


def uncl(y):
    return y.float().cpu().permute(3, 0, 1, 2).unsqueeze(0)


def h(x):
    return x.to(F16).float()


@pytest.fixture(scope="module")
def net():
    from wan2gp_amd.vae import _VaeNet
    g = torch.Generator().manual_seed(4)
    sd = {}

    def mk(name, cout, cin, k):
        fan = cin * k[0] * k[1] * k[2]
        sd[name + ".weight"] = h(torch.randn(cout, cin, *k, generator=g) / fan ** 0.5)
        sd[name + ".bias"] = h(0.1 * torch.randn(cout, generator=g))
    mk("c333", 96, 64, (3, 3, 3)); mk("c333b", 64, 96, (3, 3, 3)); mk("c111", 192, 96, (1, 1, 1))
    mk("tconv", 128, 64, (3, 1, 1)); mk("dtconv", 64, 64, (3, 1, 1)); mk("head", 3, 96, (3, 3, 3))
    sd["c2d.weight"] = h(torch.randn(32, 64, 3, 3, generator=g) / 24); sd["c2d.bias"] = h(0.1 * torch.randn(32, generator=g))
    sd["g.gamma"] = h(1 + 0.1 * torch.randn(96, 1, 1, 1, generator=g))
    return _VaeNet(sd, torch.device("cuda")), sd


def close(got, ref, what, tol=2e-3):
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= tol * scale + 1e-6, f"{what}: err {err} vs scale {scale}"


def test_conv_causal_with_and_without_cache(net):
    n, sd = net
    g = torch.Generator().manual_seed(1)
    x = h(torch.randn(1, 64, 3, 10, 14, generator=g))
    ref = VO.causal_conv3d(x, sd["c333.weight"], sd["c333.bias"])
    close(uncl(n.conv(cl(x), "c333")), ref, "3x3x3 no cache")
    cache = h(torch.randn(1, 64, 2, 10, 14, generator=g))
    ref = VO.causal_conv3d(x, sd["c333.weight"], sd["c333.bias"], cache)
    close(uncl(n.conv(cl(x), "c333", cache=cl(cache))), ref, "3x3x3 cache")
    # single-frame chunk, residual add, ragged pixel count (not a multiple of 128)
    x1 = h(torch.randn(1, 96, 1, 9, 7, generator=g)); c1 = h(torch.randn(1, 96, 2, 9, 7, generator=g))
    res = h(torch.randn(1, 64, 1, 9, 7, generator=g))
    ref = h(VO.causal_conv3d(x1, sd["c333b.weight"], sd["c333b.bias"], c1)) + res
    close(uncl(n.conv(cl(x1), "c333b", cache=cl(c1), res=cl(res))), ref, "residual")
    ref = VO.causal_conv3d(x1, sd["c111.weight"], sd["c111.bias"])
    close(uncl(n.conv(cl(x1), "c111")), ref, "1x1x1")
    ref = VO.causal_conv3d(x1, sd["head.weight"], sd["head.bias"], c1)
    got = n.conv(cl(x1), "head", cache=cl(c1), out_f32=True)
    close(uncl(got), ref, "3-channel fp32 head")


def test_conv_resample_variants(net):
    n, sd = net
    g = torch.Generator().manual_seed(2)
    x = h(torch.randn(1, 64, 2, 6, 10, generator=g))
    # upsample2d: nearest-exact 2x then Conv2d 3x3 pad 1 (vae.py:126-128)
    y = x[0].permute(1, 0, 2, 3)
    ref = F.conv2d(F.interpolate(y, scale_factor=(2., 2.), mode="nearest-exact"), sd["c2d.weight"], sd["c2d.bias"], padding=1)
    ref = ref.permute(1, 0, 2, 3).unsqueeze(0)
    close(uncl(n.conv(cl(x), "c2d", ups=True)), ref, "upsample conv2d")
    # downsample2d: ZeroPad2d((0,1,0,1)) + stride 2 (vae.py:137-139), odd size
    x2 = h(torch.randn(1, 64, 2, 7, 9, generator=g))
    y = x2[0].permute(1, 0, 2, 3)
    ref = F.conv2d(F.pad(y, (0, 1, 0, 1)), sd["c2d.weight"], sd["c2d.bias"], stride=2).permute(1, 0, 2, 3).unsqueeze(0)
    close(uncl(n.conv(cl(x2), "c2d", st_s=2, pad_s=0)), ref, "downsample conv2d")
    # upsample3d time_conv + channel->time interleave (vae.py:183-189)
    cache = h(torch.randn(1, 64, 2, 6, 10, generator=g))
    yt = VO.causal_conv3d(x, sd["tconv.weight"], sd["tconv.bias"], cache, pad=(1, 0, 0))
    b, c2, t, hh, ww = yt.shape
    c = c2 // 2
    yt = yt.reshape(b, 2, c, t, hh, ww)
    ref = torch.stack((yt[:, 0], yt[:, 1]), 3).reshape(b, c, t * 2, hh, ww)
    close(uncl(n.conv(cl(x), "tconv", cache=cl(cache), interleave=True, pad_s=0)), ref, "time interleave")
    # downsample3d time_conv: stride 2 over [last cached frame ; x] (vae.py:205-206)
    x4 = h(torch.randn(1, 64, 4, 5, 6, generator=g)); last = h(torch.randn(1, 64, 1, 5, 6, generator=g))
    ref = VO.causal_conv3d(torch.cat([last, x4], 2), sd["dtconv.weight"], sd["dtconv.bias"], stride=(2, 1, 1), pad=(0, 0, 0))
    prev2 = torch.cat([torch.zeros_like(last), last], 2)
    close(uncl(n.conv(cl(x4), "dtconv", cache=cl(prev2), st_t=2, front=1, pad_s=0)), ref, "stride-2 time conv")


def test_rmsnorm_silu(net):
    n, sd = net
    g = torch.Generator().manual_seed(3)
    x = h(torch.randn(1, 96, 2, 5, 7, generator=g) * 3)
    ref = F.silu(VO.rms_norm(x, sd["g.gamma"]))
    close(uncl(n.norm(cl(x), "g.gamma")), ref, "rmsnorm+silu", tol=3e-3)
    ref = VO.rms_norm(x, sd["g.gamma"])
    close(uncl(n.norm(cl(x), "g.gamma", silu=False)), ref, "rmsnorm", tol=3e-3)


@pytest.mark.parametrize("C,npix", [(32, 5), (96, 70), (192, 33), (384, 129), (512, 7), (640, 50), (1024, 3)])
def test_rmsnorm_kernel_every_channel_width(C, npix):
    """The lane-group layouts of wan_vae_rmsnorm_silu: 16 / 32 / 64 lanes per pixel, two chunks per lane above 512 channels (the
    Wan2.2 VAE's 640), pixel counts that do not fill the last wave; against F.normalize * sqrt(C) * gamma (vae.py:94-107)."""
    from wan2gp_amd import lib as L
    g = torch.Generator().manual_seed(C + npix)
    x = (torch.randn(npix, C, generator=g) * 2).to(F16)
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).to(F16)
    for silu in (0, 1):
        xc, gc = x.cuda(), gamma.cuda()
        out = torch.empty_like(xc)
        L.check(L.load().wan_vae_rmsnorm_silu(L.ptr(xc), L.ptr(out), L.ptr(gc), npix, C, silu, L.stream_ptr()), "rmsnorm")
        ref = F.normalize(x.float(), dim=1) * (C ** 0.5) * gamma.float()
        if silu:
            ref = F.silu(ref.half().float())
        err = (out.float().cpu() - ref).abs().max().item()
        assert err <= 3e-3 * max(1.0, ref.abs().max().item()), (C, npix, silu, err)


@pytest.fixture(scope="module")
def vae():
    from wan2gp_amd.vae import WanVAEHIP
    return WanVAEHIP(state_dict=VO.synth_vae_weights(), device="cuda")


def test_attention_block_vs_oracle(vae):
    W = VO.synth_vae_weights()
    g = torch.Generator().manual_seed(6)
    x = h(torch.randn(1, 384, 2, 8, 8, generator=g))
    ref = VO.attention_block(x, W, "decoder.middle.1.")
    got = vae.net.attention_block(cl(x), "decoder.middle.1.")
    close(uncl(got), ref, "attention block", tol=4e-3)


def test_decode_to_uint8_vs_reference_golden(vae):
    gold = dict(np.load(os.path.join(G, "vae_small.npz")))
    gen = torch.Generator().manual_seed(21)
    z = torch.randn(1, 16, 3, 8, 8, generator=gen)
    u8 = vae.decode_to_cpu_uint8([z[0]], 0)[0]
    ref = torch.from_numpy(gold["dec_u8"])[0]
    assert u8.dtype == torch.uint8 and u8.shape == ref.shape and u8.device.type == "cpu"
    d = (u8.int() - ref.int()).abs()
    frac_same = (d == 0).float().mean().item()
    print(f"VAE uint8: identical {frac_same * 100:.2f}%  max delta {int(d.max())}  mean delta {d.float().mean().item():.4f}")
    assert int(d.max()) <= 1 and frac_same >= 0.90 and d.float().mean().item() <= 0.1
    dec = vae.decode([z[0]], 0)[0].cpu()
    refd = torch.from_numpy(gold["dec"])[0].clamp(-1, 1)
    assert (dec - refd).abs().max().item() <= 1.5e-2


def test_encode_vs_reference_golden(vae):
    gold = dict(np.load(os.path.join(G, "vae_small.npz")))
    gen = torch.Generator().manual_seed(21)
    _ = torch.randn(1, 16, 3, 8, 8, generator=gen)
    vid = (torch.rand(1, 3, 9, 64, 64, generator=gen) * 2 - 1)
    vid[:, :, 1:] *= 0.5
    mu = vae.encode([vid[0]])[0].cpu()
    ref = torch.from_numpy(gold["enc"])[0]
    assert mu.shape == ref.shape and mu.dtype == torch.float32
    err = (mu - ref).abs().max().item()
    print(f"VAE encode: max abs err {err:.4e} (|ref| max {ref.abs().max().item():.3f})")
    assert err <= 1e-2 * ref.abs().max().item() + 1e-3


def test_spatial_tiling_vs_reference_golden(vae):
    """tile_size > 0 (vae.py:676-717, :769-839, :841-881): 3 x 3 overlapping 64-px tiles of a 128 x 128 clip against the
    reference's own tiled decode / uint8 decode / encode (tests/golden/vae_tiled.npz); same bars as the untiled paths."""
    gold = dict(np.load(os.path.join(G, "vae_tiled.npz")))
    gen = torch.Generator().manual_seed(22)
    z = torch.randn(1, 16, 2, 16, 16, generator=gen)
    vid = (torch.rand(1, 3, 5, 128, 128, generator=gen) * 2 - 1)
    vid[:, :, 1:] *= 0.5
    dec = vae.decode([z[0]], 64)[0].cpu()
    refd = torch.from_numpy(gold["dec"])[0].clamp(-1, 1)
    assert dec.shape == refd.shape and (dec - refd).abs().max().item() <= 1.5e-2
    u8 = vae.decode_to_cpu_uint8([z[0]], 64)[0]
    ref = torch.from_numpy(gold["dec_u8"])[0]
    d = (u8.int() - ref.int()).abs()
    frac_same = (d == 0).float().mean().item()
    print(f"VAE tiled uint8: identical {frac_same * 100:.2f}%  max delta {int(d.max())}  mean delta {d.float().mean().item():.4f}")
    assert u8.shape == ref.shape and int(d.max()) <= 1 and frac_same >= 0.90 and d.float().mean().item() <= 0.1
    crop = vae.decode_to_cpu_uint8([z[0]], 64, target_frames=3, target_height=100, target_width=120, frame_start=1)[0]
    assert torch.equal(crop, u8[:, 1:4, :100, :120]) and crop.shape == torch.from_numpy(gold["dec_u8_crop"])[0].shape
    untiled = vae.decode_to_cpu_uint8([z[0]], 0)[0]
    assert (untiled.int() - u8.int()).abs().float().mean().item() > 0.5          # tiling really changes the picture (no cross-tile context)
    mu = vae.encode([vid[0]], 64)[0].cpu()
    refe = torch.from_numpy(gold["enc"])[0]
    err = (mu - refe).abs().max().item()
    print(f"VAE tiled encode: max abs err {err:.4e} (|ref| max {refe.abs().max().item():.3f})")
    assert mu.shape == refe.shape and err <= 1e-2 * refe.abs().max().item() + 1e-3


def test_native_graph_equals_the_host_graph_bit_for_bit(vae):
    """wan_vae_decode / wan_vae_encode (csrc/vae_graph.hip) run the same kernels in the same order as the host graph of
    wan2gp_amd/vae.py: every output byte must be equal; the planned workspace is what the run needs."""
    assert vae.native is not None
    gen = torch.Generator().manual_seed(33)
    z = torch.randn(16, 4, 8, 12, generator=gen)
    vid = torch.rand(3, 9, 64, 96, generator=gen) * 2 - 1
    nat_u8, nat_f32 = vae._decode_frames(z, True, True)
    nat_mu = vae.encode([vid])[0]
    keep, vae.native = vae.native, None
    try:
        host_u8, host_f32 = vae._decode_frames(z, True, True)
        host_mu = vae.encode([vid])[0]
    finally:
        vae.native = keep
    assert torch.equal(nat_u8, host_u8) and torch.equal(nat_f32, host_f32) and torch.equal(nat_mu, host_mu)
    need = vae.native.lib.wan_vae_workspace_bytes(vae.native._h, 1, 4, 8, 12)
    assert 0 < need <= vae.native._ws.numel()
    from wan2gp_amd.lib import WanHipError, check, ptr, stream_ptr
    small = torch.empty(need // 2, dtype=torch.uint8, device="cuda")
    out = torch.empty(3, 13, 64, 96, dtype=torch.uint8, device="cuda")
    with pytest.raises(WanHipError, match="workspace too small"):
        check(vae.native.lib.wan_vae_decode(vae.native._h, ptr(z.cuda()), 4, 8, 12, ptr(out), None, ptr(small), small.numel(), stream_ptr()), "decode")
    with pytest.raises(WanHipError, match="4k \\+ 1"):
        check(vae.native.lib.wan_vae_encode(vae.native._h, ptr(vid.cuda()), 8, 64, 96, ptr(out), ptr(small), small.numel(), stream_ptr()), "encode")


def test_encode_decode_roundtrip_shapes(vae):
    """size-independent property at a larger size: chunked causal encode (1+4+4 frames) and frame-by-frame
    decode agree on shapes ( (T-1)/4+1 latents, (t-1)*4+1 frames ) and stay finite."""
    g = torch.Generator().manual_seed(7)
    vid = torch.rand(3, 13, 96, 160, generator=g) * 2 - 1
    mu = vae.encode([vid])[0]
    assert tuple(mu.shape) == (16, 4, 12, 20) and torch.isfinite(mu).all()
    u8 = vae.decode_to_cpu_uint8([mu], 0)[0]
    assert tuple(u8.shape) == (3, 13, 96, 160)


def test_any_end_frame_vs_reference_golden(vae):
    """Start + end image clips (vae.py:590-606, :646-650; any2video.py:743 / :1784 for the Wan2.1 i2v model): the end frame bypasses
    the causal feature cache.  tests/golden/vae_endframe.npz = the reference's own encode / decode / decode_to_cpu_uint8 with
    any_end_frame=True; same bars as the plain paths; the tiled form takes the same route per tile."""
    gold = dict(np.load(os.path.join(G, "vae_endframe.npz")))
    gen = torch.Generator().manual_seed(23)
    z = torch.randn(1, 16, 4, 8, 8, generator=gen)
    vid = (torch.rand(1, 3, 10, 64, 64, generator=gen) * 2 - 1)
    vid[:, :, 1:-1] *= 0.5
    u8 = vae.decode_to_cpu_uint8([z[0]], 0, any_end_frame=True)[0]
    ref = torch.from_numpy(gold["dec_u8"])[0]
    assert u8.dtype == torch.uint8 and tuple(u8.shape) == (3, 10, 64, 64) == tuple(ref.shape)
    d = (u8.int() - ref.int()).abs()
    frac_same = (d == 0).float().mean().item()
    print(f"VAE end-frame uint8: identical {frac_same * 100:.2f}%  max delta {int(d.max())}  mean delta {d.float().mean().item():.4f}")
    assert int(d.max()) <= 1 and frac_same >= 0.90 and d.float().mean().item() <= 0.1
    dec = vae.decode([z[0]], 0, any_end_frame=True)[0].cpu()
    assert (dec - torch.from_numpy(gold["dec"])[0].clamp(-1, 1)).abs().max().item() <= 1.5e-2
    mu = vae.encode([vid[0]], any_end_frame=True)[0].cpu()
    refe = torch.from_numpy(gold["enc"])[0]
    assert tuple(mu.shape) == (16, 4, 8, 8)
    err = (mu - refe).abs().max().item()
    print(f"VAE end-frame encode: max abs err {err:.4e} (|ref| max {refe.abs().max().item():.3f})")
    assert err <= 1e-2 * refe.abs().max().item() + 1e-3
    body = vae.encode([vid[0][:, :9]])[0].cpu()                        # the first nine frames are the ordinary causal chain
    assert torch.equal(mu[:, :3], body)


# ---- a convolution's result must not depend on what else the GPU is doing ----------------------------------------------------------------
def _conv_repeat_worker(tag, iters, q):
    try:
        from wan2gp_amd.vae import WanVAEHIP, random_vae_state_dict
        n = WanVAEHIP(state_dict=random_vae_state_dict()).net
        g = torch.Generator().manual_seed(3)
        bad = {}
        for name, f32 in (("decoder.head.2", True), ("decoder.middle.0.residual.2", False)):
            c = n.convs[name]
            x = (torch.randn(4, 64, 64, c.cin, generator=g) * 0.5).to(torch.float16).cuda()
            ref = n.conv(x, name, out_f32=f32).clone()
            bad[name] = sum(int(not torch.equal(n.conv(x, name, out_f32=f32), ref)) for _ in range(iters))
        q.put((tag, bad))
    except Exception:
        import traceback
        q.put((tag, traceback.format_exc()))


def test_halo_convolution_is_reproducible_beside_another_process():
    """Round 6 (runs 42-47): the halo convolution's weight ring is three stages deep and a stage is refilled one barrier after it was read.  The
    reads were only ISSUED in front of that barrier -- hipcc parks the consuming MFMA, and with it the lgkmcnt wait, behind it -- so with
    the LDS queue backed up by another process on the same GPU the refill could land first: a wave's tile of garbage in 1 of 400 launches of
    the 96-wide tiles, 1 in 6 of the decoder head's 16-channel tile.  Two processes launch the same two convolutions 300 times side by side;
    every launch must return the first launch's bits."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_conv_repeat_worker, args=(t, 300, q)) for t in ("A", "B")]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for tag, bad in res:
        assert isinstance(bad, dict), f"{tag}: {bad}"
        assert all(v == 0 for v in bad.values()), (tag, bad)
