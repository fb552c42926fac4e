"""-m gpu: the causal 3D VAE at the BASELINE size (720 x 1280) and the integer output conversion, through the C ABI.

  * wan_vae_to_video against _vae_float_to_cpu_uint8 (vae.py:18-20) -- integer work, bar = BIT-EXACT (torch.equal) on a
    crafted fp32 tensor: every tie (x + 1) * 127.5 = k + 0.5 with its fp32 neighbours, every exact level k / 127.5 - 1,
    the +-1 clamp edges, values beyond them, +-inf, zeros / denormals, uniform and normal randoms.
  * decode and encode at 720 x 1280 for 1 + 4 + 4 frames (latent t = 3) against oracle/vae_oracle.py.  The restatement is
    plain torch code; at this size the box's 128 cores need minutes per pass (65 s for the 17 x 320 x 512 decode of the
    bench's CPU baseline; two passes at 720p did not finish in 15 minutes), so its arithmetic is executed by PyTorch ON THE
    GPU in fp32 -- every convolution as the sum of its taps' matmuls (vae_oracle._conv_taps: no vendor convolution or
    attention kernel) -- and that execution is pinned to the CPU execution at the golden size in this file (identical
    uint8 frames, float frames to 1e-4).  (a) the fp32 plan = the reference-pinned restatement, bar on the uint8 frames: max |delta| <= 1 LSB,
    >= 90 % of the bytes identical (the HIP library stores activations in fp16, the reference's default VAE dtype on a
    GPU, wgp.py:4038; the golden is the reference's fp32 CPU run); (b) the fp16 storage plan of the same restatement
    (`with VO.fp16_plan()`): rounding to fp16 at the points where the library stores fp16.  Measured (run 63): an fp16
    plan is NOT reproducible across summation orders -- the restatement's own two executions (CPU / GPU) of it agree on
    92.3 % of the bytes only (its fp32 plan: 99.995 %), because a flipped fp16 rounding propagates through 60 layers.
    So (b) cannot be "the library equals the fp16-plan oracle"; what it pins is the SIZE of the effect: the library's
    distance to the fp32 frames (mean |delta| in LSB) must not exceed 1.25 x the distance of the restatement's own fp16
    plan to them, and its distance to that fp16-plan run must stay within 1.5 x of it -- an independent implementation
    of the storage plan lands as far from the fp32 golden as the library does, max 1 LSB everywhere.  That attributes
    the differing bytes of (a) to the storage plan and not to a kernel.
  * conv3d_f16_kernel<BIG> (64-bit gather offsets): forced onto an ordinary input and held bit-for-bit against the 32-bit
    instantiation (plain, cached, up-sampled, stride-2), and reached for real by a chunk of more than 2^31 elements,
    checked on sampled output pixels against an fp64 dot product of the same taps.

The tables go to gpurun_out/parity/ (committed under profiles/).
"""
import json
import os
import time

import pytest
import torch

from oracle import vae_oracle as VO

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F16 = torch.float16


def _report(name, obj):
    d = os.path.join(ROOT, "gpurun_out", "parity")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, name + ".json"), "w") as f:
            json.dump(obj, f, indent=1)
    except OSError:
        pass


# ---- a21: float -> uint8 --------------------------------------------------------------------------------------------------------
def _crafted_floats():
    k = torch.arange(0, 256, dtype=torch.float64)
    ties = ((k + 0.5) / 127.5 - 1.0).float()                  # (x + 1) * 127.5 lands on or next to k + 0.5
    levels = (k / 127.5 - 1.0).float()
    base = torch.cat([ties, levels, torch.tensor([-1.0, 1.0, 0.0, -0.0, 1e-40, -1e-40, 1e-30, -1e-30])])
    near = [base]
    for step in range(1, 4):                                   # the three fp32 neighbours on each side
        up, dn = base.clone(), base.clone()
        for _ in range(step):
            up = torch.nextafter(up, torch.full_like(up, 10.0))
            dn = torch.nextafter(dn, torch.full_like(dn, -10.0))
        near += [up, dn]
    g = torch.Generator().manual_seed(5)
    rnd = [torch.rand(200000, generator=g) * 2.4 - 1.2, torch.randn(200000, generator=g),
           torch.randn(1000, generator=g) * 1e3, torch.tensor([float("inf"), float("-inf"), 3.0e38, -3.0e38, 2.0, -2.0])]
    # ties reached in fp32 arithmetic itself: x such that fl(fl(x + 1) * 127.5) is exactly k + 0.5
    exact = []
    for kk in range(0, 255):
        x = torch.tensor([(kk + 0.5) / 127.5 - 1.0], dtype=torch.float32)
        for _ in range(8):
            x = torch.cat([x, torch.nextafter(x[-1:], torch.tensor([10.0])), torch.nextafter(x[:1], torch.tensor([-10.0]))])
        exact.append(x[((x + 1.0) * 127.5) == kk + 0.5])
    return torch.cat(near + rnd + exact)


def test_float_to_uint8_is_bit_exact():
    from wan2gp_amd import lib as L
    vals = _crafted_floats()
    n_ties = int((((vals.clamp(-1, 1) + 1.0) * 127.5) % 1.0 == 0.5).sum())
    assert n_ties >= 200, n_ties                                # the crafted set really contains round-half-even cases
    T, HW = 3, (vals.numel() + 8) // 9 + 1
    x = torch.zeros(T * HW * 3)
    x[: vals.numel()] = vals
    x = x[torch.randperm(x.numel(), generator=torch.Generator().manual_seed(1))].reshape(T, HW, 3)   # channels-last [T, HW, 3]
    ref = VO.float_to_uint8(x.permute(2, 0, 1).contiguous())    # [3, T, HW] -- the reference's own statement sequence
    lib = L.load()
    xc = x.cuda()
    Ttot, t0 = T + 3, 2
    u8 = torch.full((3, Ttot, HW), 77, dtype=torch.uint8, device="cuda")
    f32 = torch.full((3, Ttot, HW), -5.0, dtype=torch.float32, device="cuda")
    L.check(lib.wan_vae_to_video(L.ptr(xc), L.ptr(u8), L.ptr(f32), T, HW, Ttot, t0, L.stream_ptr()), "wan_vae_to_video")
    torch.cuda.synchronize()
    got = u8.cpu()
    assert torch.equal(got[:, t0:t0 + T], ref), "float -> uint8 differs from _vae_float_to_cpu_uint8"
    assert (got[:, :t0] == 77).all() and (got[:, t0 + T:] == 77).all()          # frames outside [t0, t0 + T) untouched
    assert torch.equal(f32.cpu()[:, t0:t0 + T], x.permute(2, 0, 1))             # the fp32 output is the unclamped copy (decode())
    # the Wan2.1 decode path uses the same conversion at frame granularity: u8-only and f32-only calls
    u8b = torch.empty(3, T, HW, dtype=torch.uint8, device="cuda")
    L.check(lib.wan_vae_to_video(L.ptr(xc), L.ptr(u8b), None, T, HW, T, 0, L.stream_ptr()), "wan_vae_to_video")
    assert torch.equal(u8b.cpu(), ref)
    print(f"\n[a21] {vals.numel()} crafted values ({n_ties} exact ties): bit-exact")


# ---- a18 / a20 at 720 x 1280 ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def vae():
    from wan2gp_amd.vae import WanVAEHIP
    return WanVAEHIP(state_dict=VO.synth_vae_weights(), device="cuda")


def _u8_stats(a, b):
    d = (a.int() - b.int()).abs()
    return {"identical": (d == 0).float().mean().item(), "max_lsb": int(d.max()), "mean_lsb": d.float().mean().item()}


def _on_gpu(W, scale):
    return {k: v.cuda() for k, v in W.items()}, [s.cuda() for s in scale]


def test_oracle_executed_on_the_gpu_equals_its_cpu_execution():
    """What the 720p tests below compare with: the same restatement, run by torch on the GPU.  At the golden size both
    executions are affordable: uint8 frames identical (>= 99.99 %: a tie can flip), float frames / latents within 1e-4, in
    both plans."""
    W, scale = VO.synth_vae_weights(), VO.default_scale()
    Wg, sg = _on_gpu(W, scale)
    g = torch.Generator().manual_seed(21)
    z = torch.randn(1, 16, 3, 8, 8, generator=g)
    vid = torch.rand(1, 3, 9, 64, 64, generator=g) * 2 - 1
    with torch.no_grad():
        for plan in (False, True):
            ctx = VO.fp16_plan() if plan else __import__("contextlib").nullcontext()
            with ctx:
                dc, dg = VO.vae_decode(z, W, scale), VO.vae_decode(z.cuda(), Wg, sg).cpu()
                ec, eg = VO.vae_encode(vid, W, scale), VO.vae_encode(vid.cuda(), Wg, sg).cpu()
            st = _u8_stats(VO.float_to_uint8(dg), VO.float_to_uint8(dc))
            print(f"\n[oracle gpu vs cpu, fp16 plan={plan}] decode {st}, float {float((dg - dc).abs().max()):.2e}, encode {float((eg - ec).abs().max()):.2e}")
            if plan:
                # the fp16 plan amplifies summation order: a flipped fp16 rounding propagates (measured 92.3 % identical bytes
                # between the two executions) -- this is the noise floor the 720p test measures the library against
                assert st["max_lsb"] <= 1 and st["identical"] >= 0.85
                assert (dg - dc).abs().max() <= 2e-2 and (eg - ec).abs().max() <= 2e-2 * max(1.0, float(ec.abs().max()))
            else:
                assert st["identical"] >= 0.9999 and st["max_lsb"] <= 1
                assert (dg - dc).abs().max() <= 1e-4 * max(1.0, float(dc.abs().max())) and (eg - ec).abs().max() <= 1e-4 * max(1.0, float(ec.abs().max()))


def test_decode_720p_vs_oracle_fp32_and_fp16_plan(vae):
    W, scale = _on_gpu(VO.synth_vae_weights(), VO.default_scale())
    g = torch.Generator().manual_seed(720)
    z = torch.randn(16, 3, 90, 160, generator=g)
    t0 = time.time()
    u8 = vae.decode_to_cpu_uint8([z], 0)[0]
    dec = vae.decode([z], 0)[0].cpu()
    t_hip = time.time() - t0
    assert u8.dtype == torch.uint8 and tuple(u8.shape) == (3, 9, 720, 1280)
    torch.cuda.empty_cache()
    with torch.no_grad():
        t0 = time.time()
        ref32 = VO.vae_decode(z[None].cuda(), W, scale)[0].cpu()
        t32 = time.time() - t0
        with VO.fp16_plan():
            ref16 = VO.vae_decode(z[None].cuda(), W, scale)[0].cpu()
        t16 = time.time() - t0 - t32
    torch.cuda.empty_cache()
    s32 = _u8_stats(u8, VO.float_to_uint8(ref32))
    s16 = _u8_stats(u8, VO.float_to_uint8(ref16))
    plan = _u8_stats(VO.float_to_uint8(ref16), VO.float_to_uint8(ref32))
    f32_err = (dec - ref32.clamp(-1, 1)).abs().max().item()
    f16_err = (dec - ref16.clamp(-1, 1)).abs().max().item()
    sat = ((ref32 <= -1) | (ref32 >= 1)).float().mean().item()
    res = {"shape": list(u8.shape), "hip_vs_oracle_fp32": s32, "hip_vs_oracle_fp16_plan": s16, "fp16_plan_vs_fp32_oracle": plan,
           "max_abs_err_float_frames": {"vs_fp32": f32_err, "vs_fp16_plan": f16_err}, "saturated_fraction": sat,
           "seconds": {"hip_two_decodes": t_hip, "oracle_fp32_on_gpu": t32, "oracle_fp16_plan_on_gpu": t16}}
    print("\n[VAE decode 720x1280x9f] " + json.dumps(res))
    _report("vae_decode_720p_t3", res)
    assert sat < 0.5, "the synthetic decode saturates: the byte comparison would be vacuous"
    assert s32["max_lsb"] <= 1 and s32["identical"] >= 0.90 and s32["mean_lsb"] <= 0.1, s32
    assert plan["max_lsb"] <= 1 and s16["max_lsb"] <= 1
    # the size of the effect is the storage plan's: an independent fp16-plan execution is as far from fp32 as the library
    assert s32["mean_lsb"] <= 1.25 * plan["mean_lsb"] + 5e-3, (s32, plan)
    assert s16["mean_lsb"] <= 1.5 * plan["mean_lsb"] + 5e-3, (s16, plan)
    assert f32_err <= 1.5e-2


def test_encode_720p_vs_oracle_fp32_and_fp16_plan(vae):
    W, scale = _on_gpu(VO.synth_vae_weights(), VO.default_scale())
    g = torch.Generator().manual_seed(721)
    vid = torch.rand(3, 9, 720, 1280, generator=g) * 2 - 1
    vid[:, 1:] *= 0.5
    t0 = time.time()
    mu = vae.encode([vid])[0].cpu()
    t_hip = time.time() - t0
    assert tuple(mu.shape) == (16, 3, 90, 160) and mu.dtype == torch.float32
    torch.cuda.empty_cache()
    with torch.no_grad():
        t0 = time.time()
        ref32 = VO.vae_encode(vid[None].cuda(), W, scale)[0].cpu()
        t32 = time.time() - t0
        with VO.fp16_plan():
            ref16 = VO.vae_encode(vid[None].cuda(), W, scale)[0].cpu()
        t16 = time.time() - t0 - t32
    torch.cuda.empty_cache()
    sc = ref32.abs().max().item()
    e32, e16 = (mu - ref32).abs().max().item(), (mu - ref16).abs().max().item()
    r32 = ((mu - ref32).norm() / ref32.norm()).item()
    r16 = ((mu - ref16).norm() / ref16.norm()).item()
    res = {"shape": list(mu.shape), "max_abs_ref": sc, "max_abs_err": {"vs_fp32": e32, "vs_fp16_plan": e16},
           "rel_l2_err": {"vs_fp32": r32, "vs_fp16_plan": r16},
           "seconds": {"hip": t_hip, "oracle_fp32_on_gpu": t32, "oracle_fp16_plan_on_gpu": t16}}
    print("\n[VAE encode 9f 720x1280] " + json.dumps(res))
    _report("vae_encode_720p_9f", res)
    rplan = ((ref16 - ref32).norm() / ref32.norm()).item()
    print(f"[VAE encode 9f 720x1280] fp16-plan oracle vs fp32 oracle: rel l2 {rplan:.3e}")
    assert e32 <= 1e-2 * sc + 1e-3, res                                     # the bar of the small-size golden test
    assert r32 <= 1.5 * rplan + 1e-4 and r16 <= 2.0 * rplan + 1e-4, (res, rplan)   # no further from fp32 than the storage plan itself puts an independent run


# ---- conv3d_f16_kernel<BIG> -------------------------------------------------------------------------------------------------
def _cl(x):
    return x[0].permute(1, 2, 3, 0).contiguous().to(F16).cuda()


def test_conv_big_offsets_forced_equal_the_32bit_kernel_bit_for_bit():
    from wan2gp_amd import lib as L
    from wan2gp_amd.vae import _VaeNet
    g = torch.Generator().manual_seed(14)
    sd = {}

    def mk(name, cout, cin, k):
        fan = cin * k[0] * k[1] * k[2]
        sd[name + ".weight"] = (torch.randn(cout, cin, *k, generator=g) / fan ** 0.5).half().float()
        sd[name + ".bias"] = (0.1 * torch.randn(cout, generator=g)).half().float()
    mk("c333", 96, 64, (3, 3, 3)); mk("tconv", 128, 64, (3, 1, 1)); mk("dtconv", 64, 64, (3, 1, 1))
    sd["c2d.weight"] = (torch.randn(32, 64, 3, 3, generator=g) / 24).half().float()
    sd["c2d.bias"] = (0.1 * torch.randn(32, generator=g)).half().float()
    n = _VaeNet(sd, torch.device("cuda"))
    x = _cl(torch.randn(1, 64, 3, 22, 30, generator=g)); cache = _cl(torch.randn(1, 64, 2, 22, 30, generator=g))
    res = _cl(torch.randn(1, 96, 3, 22, 30, generator=g))
    last = _cl(torch.randn(1, 64, 1, 22, 30, generator=g))
    prev2 = torch.cat([torch.zeros_like(last), last], 0)
    x4 = _cl(torch.randn(1, 64, 4, 22, 30, generator=g))
    cases = {
        "3x3x3 no cache": lambda: n.conv(x, "c333"),
        "3x3x3 cache + residual": lambda: n.conv(x, "c333", cache=cache, res=res),
        "upsampled conv2d": lambda: n.conv(x, "c2d", ups=True),
        "stride-2 conv2d": lambda: n.conv(x, "c2d", st_s=2, pad_s=0),
        "time interleave": lambda: n.conv(x, "tconv", cache=cache, interleave=True, pad_s=0),
        "stride-2 time conv": lambda: n.conv(x4, "dtconv", cache=prev2, st_t=2, front=1, pad_s=0),
    }
    lib = L.load()
    no_halo = lib.wan_vae_debug_no_halo(1)     # the 3 x 3 x 3 cases on the gather kernel both times (the halo-patch kernel sums in another order)
    try:
        small = {k: f().clone() for k, f in cases.items()}
        old = lib.wan_vae_debug_force_big(1)
        try:
            assert old == 0
            big = {k: f().clone() for k, f in cases.items()}
        finally:
            lib.wan_vae_debug_force_big(0)
    finally:
        lib.wan_vae_debug_no_halo(no_halo)
    torch.cuda.synchronize()
    for k in cases:
        assert torch.isfinite(small[k].float()).all() and small[k].float().abs().max() > 0.1, k
        assert torch.equal(small[k], big[k]), f"BIG instantiation differs from the 32-bit one: {k}"


@pytest.mark.parametrize("cin,cout,T,H,W,cached,resid,f32out", [
    (96, 96, 3, 48, 80, True, True, False),      # a residual block conv of the 96-channel level: whole tiles, cache, residual
    (192, 192, 2, 37, 53, True, False, False),   # ragged tiles on both edges, two cout tiles (the second half empty above 192)
    (32, 96, 1, 16, 16, False, False, False),    # one tile, one channel block, no cache (causal front = zeros)
    (384, 160, 2, 9, 23, True, True, False),     # image smaller than a tile in height, 12 channel blocks, ragged cout tile
    (96, 32, 4, 30, 40, False, False, True),     # the decoder's head: fp32 output, a quarter cout tile
    (64, 64, 5, 33, 17, True, False, False),
    (192, 384, 1, 20, 21, True, True, False),    # four 96-wide cout tiles
    (96, 96, 2, 18, 35, False, False, True),     # the 96-wide tile with fp32 output
], ids=["96_level", "192_ragged", "one_tile", "384_short", "head_f32", "64ch", "cout_384", "96_f32"])
def test_conv_halo_patch_kernel_against_fp64_and_the_gather_kernel(cin, cout, T, H, W, cached, resid, f32out):
    """Round 4: the 3 x 3 x 3 stride-1 convolutions run on the halo-patch kernel (csrc/vae_conv_halo.hip: the 18 x 18 input patch of a 16 x 16
    tile staged once per frame tap and channel block, nine taps read shifted windows).  Against an fp64 convolution of the same fp16
    operands on every output element (the error of an fp32-accumulating kernel: rounding of the fp16 store), and against the gather
    kernel (wan_vae_debug_no_halo) -- the two sum in different orders and may differ by an fp16 ulp here and there, never more."""
    from wan2gp_amd import lib as L
    from wan2gp_amd.vae import _VaeNet
    g = torch.Generator().manual_seed(cin + cout + H)
    wt = (torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5).half().float()
    bs = (0.1 * torch.randn(cout, generator=g)).half().float()
    n = _VaeNet({"c.weight": wt, "c.bias": bs}, torch.device("cuda"))
    x5 = torch.randn(1, cin, T, H, W, generator=g).half().float()
    c5 = torch.randn(1, cin, 2, H, W, generator=g).half().float() if cached else None
    r5 = torch.randn(1, cout, T, H, W, generator=g).half().float() if resid else None
    x, cache, res = _cl(x5), (_cl(c5) if cached else None), (_cl(r5) if resid else None)
    lib = L.load()
    got = n.conv(x, "c", cache=cache, res=res, out_f32=f32out).float().cpu()
    old = lib.wan_vae_debug_no_halo(1)
    try:
        gat = n.conv(x, "c", cache=cache, res=res, out_f32=f32out).float().cpu()
    finally:
        lib.wan_vae_debug_no_halo(old)
    # fp64 reference: causal front = the cache's two frames or zeros, "same" zero padding in space
    front = c5 if cached else torch.zeros(1, cin, 2, H, W)
    xin = torch.cat([front, x5], 2).double().cuda()
    ref = torch.nn.functional.conv3d(torch.nn.functional.pad(xin, (1, 1, 1, 1, 0, 0)), wt.double().cuda(), bs.double().cuda())
    if not f32out:
        ref = ref.half().double() if resid else ref
    if resid:
        ref = ref + r5.double().cuda()
    ref = ref[0].permute(1, 2, 3, 0).float().cpu()
    assert got.shape == ref.shape == gat.shape
    tol = 2e-5 if f32out else 2.0 ** -10                       # fp32 accumulation order / one fp16 rounding of O(1) values (+ the residual's)
    for name, o in (("halo", got), ("gather", gat)):
        err = (o - ref).abs()
        assert torch.isfinite(o).all() and (err <= tol * (1.0 + ref.abs()) * (3 if resid and not f32out else 1.5)).all(), (name, err.max().item())
    assert ((got - gat).abs() <= 2 * tol * (1.0 + ref.abs())).all()
    if not f32out:
        assert (got != gat).float().mean().item() < 0.2              # (mostly the same fp16 value)


@pytest.mark.parametrize("cin,cout,T,H,W,ups", [(192, 96, 3, 24, 40, True), (384, 192, 2, 11, 13, True), (96, 96, 2, 19, 33, False), (32, 32, 1, 8, 8, True)],
                         ids=["resample_192_96", "resample_384_ragged", "conv2d_plain", "one_tile_ups"])
def test_conv2d_3x3_on_the_halo_patch_kernel(cin, cout, T, H, W, ups):
    """Resample's Conv2d 3 x 3 behind the nearest-exact 2x up-sampling (vae.py:105-111, :124-141) on the halo-patch kernel (KT = 1; patch
    pixel (hi, wi) of the up-sampled frame = input pixel (hi >> 1, wi >> 1)): every output element against an fp64 convolution of the
    up-sampled input, and against the gather kernel."""
    from wan2gp_amd import lib as L
    from wan2gp_amd.vae import _VaeNet
    g = torch.Generator().manual_seed(cin + cout + W)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5).half().float()
    bs = (0.1 * torch.randn(cout, generator=g)).half().float()
    n = _VaeNet({"c.weight": wt, "c.bias": bs}, torch.device("cuda"))
    x5 = torch.randn(1, cin, T, H, W, generator=g).half().float()
    x = _cl(x5)
    lib = L.load()
    got = n.conv(x, "c", ups=ups).float().cpu()
    old = lib.wan_vae_debug_no_halo(1)
    try:
        gat = n.conv(x, "c", ups=ups).float().cpu()
    finally:
        lib.wan_vae_debug_no_halo(old)
    xin = x5[0].permute(1, 0, 2, 3).double().cuda()                      # [T, C, H, W]
    if ups:
        xin = torch.nn.functional.interpolate(xin, scale_factor=2.0, mode="nearest-exact")
    ref = torch.nn.functional.conv2d(xin, wt.double().cuda(), bs.double().cuda(), padding=1).permute(0, 2, 3, 1).float().cpu()
    assert got.shape == ref.shape == gat.shape
    tol = 2.0 ** -10
    for name, o in (("halo", got), ("gather", gat)):
        err = (o - ref).abs()
        assert torch.isfinite(o).all() and (err <= 1.5 * tol * (1.0 + ref.abs())).all(), (name, err.max().item())
    assert ((got - gat).abs() <= 2 * tol * (1.0 + ref.abs())).all() and (got != gat).float().mean().item() < 0.2


def test_conv_on_a_chunk_beyond_2_31_elements():
    """(Tin + 2) * H * W * C >= 2^31: the launcher takes the 64-bit instantiation by itself.  384 channels at 720 x 1280,
    5 frames + the 2-frame cache; 32 output channels keep it cheap.  Sampled output pixels (the far end of the tensor
    included, where a 32-bit offset would have wrapped) against an fp64 dot product of the same 27 x 384 taps."""
    from wan2gp_amd.vae import _VaeNet
    T, H, Wd, C, Co = 5, 720, 1280, 384, 32
    assert (T + 2) * H * Wd * C >= 2 ** 31
    g = torch.Generator(device="cuda").manual_seed(3)
    gc = torch.Generator().manual_seed(3)
    w = (torch.randn(Co, C, 3, 3, 3, generator=gc) / (27 * C) ** 0.5).half().float()
    b = (0.1 * torch.randn(Co, generator=gc)).half().float()
    n = _VaeNet({"big.weight": w, "big.bias": b}, torch.device("cuda"))
    x = torch.randn(T, H, Wd, C, device="cuda", generator=g, dtype=torch.float32).to(F16)
    cache = torch.randn(2, H, Wd, C, device="cuda", generator=g, dtype=torch.float32).to(F16)
    out = n.conv(x, "big", cache=cache)
    torch.cuda.synchronize()
    assert tuple(out.shape) == (T, H, Wd, Co)
    full = torch.cat([cache, x], 0)                              # frame t of x = frame t + 2 here; causal taps reach back 2 frames
    wd = w.double().cuda()                                       # [Co, C, 3, 3, 3]
    pts = [(0, 0, 0), (T - 1, H - 1, Wd - 1), (T - 1, H - 1, 0), (T - 1, 0, Wd - 1), (2, 359, 640), (T - 2, 719, 1279 - 1)]
    gp = torch.Generator().manual_seed(8)
    pts += [(int(torch.randint(0, T, (1,), generator=gp)), int(torch.randint(0, H, (1,), generator=gp)),
             int(torch.randint(0, Wd, (1,), generator=gp))) for _ in range(250)]
    worst = 0.0
    for (t, y, xx) in pts:
        acc = b.double().cuda().clone()
        for kt in range(3):
            for kh in range(3):
                yy = y + kh - 1
                if yy < 0 or yy >= H:
                    continue
                for kw in range(3):
                    xw = xx + kw - 1
                    if xw < 0 or xw >= Wd:
                        continue
                    acc += wd[:, :, kt, kh, kw] @ full[t + kt, yy, xw].double()
        err = (out[t, y, xx].double() - acc).abs().max().item()
        worst = max(worst, err / max(1.0, acc.abs().max().item()))
    print(f"\n[conv BIG] {(T + 2) * H * Wd * C / 2 ** 31:.2f} x 2^31 input elements, {len(pts)} sampled pixels: worst rel err {worst:.3e}")
    _report("vae_conv_big_chunk", {"elements_over_2_31": (T + 2) * H * Wd * C / 2 ** 31, "sampled_pixels": len(pts), "worst_rel_err": worst})
    assert worst <= 2e-3


# ---- the fp32 plan (`vae_precision` "32"): north_star's "VAE bit-pattern check on integer pixel output" in the only form that can be near-exact ----
def test_fp32_plan_at_the_golden_size_equals_the_references_own_cpu_run():
    """WanVAEHIP(dtype=torch.float32) (csrc/vae_f32.hip: fp32 weights, activations and accumulation) against tests/golden/vae_small.npz =
    the reference's own WanVAE_ executed in fp32 on the CPU: decoded float frames to 1e-4, uint8 frames >= 99.9 % identical with at most
    1 LSB anywhere (only the summation order differs: a value within 1e-6 of a rounding tie can flip), encoded latents to 1e-4."""
    import numpy as np
    from wan2gp_amd.vae import WanVAEHIP
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "vae_small.npz")))
    vae32 = WanVAEHIP(state_dict=VO.synth_vae_weights(), device="cuda", dtype=torch.float32)
    gen = torch.Generator().manual_seed(21)
    z = torch.randn(1, 16, 3, 8, 8, generator=gen)
    vid = (torch.rand(1, 3, 9, 64, 64, generator=gen) * 2 - 1)
    vid[:, :, 1:] *= 0.5
    ref = torch.from_numpy(g["dec"])[0]
    dec = vae32.decode([z[0]], 0)[0].cpu()
    u8 = vae32.decode_to_cpu_uint8([z[0]], 0)[0]
    st = _u8_stats(u8, torch.from_numpy(g["dec_u8"])[0] if g["dec_u8"].ndim == 5 else torch.from_numpy(g["dec_u8"]))
    e_dec = (dec - ref.clamp(-1, 1)).abs().max().item()
    enc = vae32.encode([vid[0]])[0].cpu()
    ref_e = torch.from_numpy(g["enc"])[0]
    e_enc = (enc - ref_e).abs().max().item()
    print(f"\n[VAE fp32 plan, golden size] decode float err {e_dec:.2e}, uint8 {st}, encode err {e_enc:.2e} (max |mu| {ref_e.abs().max().item():.2f})")
    assert e_dec <= 1e-4 and st["identical"] >= 0.999 and st["max_lsb"] <= 1, (e_dec, st)
    assert e_enc <= 1e-4 * max(1.0, ref_e.abs().max().item()), e_enc
    # tiled decode and the end-frame path run on the same graph: shapes only (their arithmetic is the fp16 plan's, checked there)
    assert tuple(vae32.decode_to_cpu_uint8([z[0]], 64)[0].shape) == (3, 9, 64, 64)


def test_fp32_plan_decode_and_encode_at_720p_vs_the_fp32_oracle():
    """The same at the BASELINE size, 720 x 1280 x 9 frames (latent t = 3), against the reference-pinned restatement in fp32 (executed by
    torch on the GPU, pinned to its CPU execution above): >= 99.9 % of the 24.9 M output bytes identical, max 1 LSB -- the round-3
    library (fp16 storage plan, the reference's default VAE dtype on a GPU) reaches 92.8 %; this shows, in the product, that what
    differs there is the storage plan's rounding and nothing else."""
    from wan2gp_amd.vae import WanVAEHIP
    W, scale = _on_gpu(VO.synth_vae_weights(), VO.default_scale())
    vae32 = WanVAEHIP(state_dict=VO.synth_vae_weights(), device="cuda", dtype=torch.float32)
    g = torch.Generator().manual_seed(720)
    z = torch.randn(16, 3, 90, 160, generator=g)
    torch.cuda.synchronize()
    t0 = time.time()
    u8 = vae32.decode_to_cpu_uint8([z], 0)[0]
    torch.cuda.synchronize()
    t_dec = time.time() - t0
    with torch.no_grad():
        ref32 = VO.vae_decode(z[None].cuda(), W, scale)[0].cpu()
    torch.cuda.empty_cache()
    s32 = _u8_stats(u8, VO.float_to_uint8(ref32))
    g2 = torch.Generator().manual_seed(721)
    vid = torch.rand(3, 9, 720, 1280, generator=g2) * 2 - 1
    vid[:, 1:] *= 0.5
    t0 = time.time()
    mu = vae32.encode([vid])[0].cpu()
    t_enc = time.time() - t0
    with torch.no_grad():
        refe = VO.vae_encode(vid[None].cuda(), W, scale)[0].cpu()
    r_enc = ((mu - refe).norm() / refe.norm()).item()
    res = {"shape": list(u8.shape), "hip_fp32_plan_vs_oracle_fp32": s32, "encode_rel_l2_err": r_enc, "encode_max_abs_err": (mu - refe).abs().max().item(),
           "seconds": {"hip_fp32_decode_9f": t_dec, "hip_fp32_encode_9f": t_enc}}
    print("\n[VAE fp32 plan 720x1280x9f] " + json.dumps(res))
    _report("vae_fp32_plan_720p_t3", res)
    assert tuple(u8.shape) == (3, 9, 720, 1280) and s32["max_lsb"] <= 1 and s32["identical"] >= 0.999, s32
    assert r_enc <= 1e-5, r_enc
