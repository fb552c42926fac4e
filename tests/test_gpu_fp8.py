"""-m gpu: the scaled-fp8 Linear (wan_fp8_quantize + wan_gemm_fp8, csrc/gemm_fp8.hip) against the reference's own results
(tests/golden/fp8_linear.npz: shared/qtypes/scaled_fp8.py executed on CPU) and the bit-exact-pinned oracle/fp8_oracle.py.

Tolerances: activation quantisation is integer work on bf16 inputs -- fp8 bytes and the scale must be EQUAL.  The product is
a sum of exact fp8 x fp8 terms in fp32: only the summation order differs from torch._scaled_mm, so the bf16 output may differ
by one ulp on a few elements (<= 2 bf16 ulp of max(|ref|, magnitude floor), <= 5 % of the elements unequal)."""
import os

import numpy as np
import pytest
import torch

from oracle import fp8_oracle as F

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
G = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "fp8_linear.npz")))
SHAPES = {"per_tensor_bias": (70, 256), "per_row_bias": (70, 256), "per_row_col_nobias": (33, 128), "batched_3d_per_row": (2, 40, 192),
          "zero_input": (16, 64)}


@pytest.fixture(scope="module")
def ops():
    from wan2gp_amd import ops as _ops, lib
    lib.load()
    return _ops


def bf(a):
    return torch.from_numpy(a.copy()).view(BF)


def close(got, ref, floor, ulps=2, frac=0.05, what=""):
    got, ref = got.float().cpu(), ref.float().cpu()
    assert got.shape == ref.shape and torch.isfinite(got).all(), what
    diff = (got - ref).abs()
    tol = torch.maximum(ref.abs(), torch.as_tensor(floor, dtype=torch.float32)) * 2.0 ** -7 * ulps
    assert not (diff > tol).any(), f"{what}: {int((diff > tol).sum())} beyond {ulps} ulp, worst {(diff / tol).max().item():.2f}x"
    assert (got != ref).float().mean().item() <= frac, f"{what}: {(got != ref).float().mean().item() * 100:.1f}% elements differ"


@pytest.mark.parametrize("name", sorted(SHAPES))
def test_quantize_activation_is_bit_exact(ops, name):
    x = bf(G[name + "/x"]).reshape(SHAPES[name]).cuda()
    if x.numel() % 8:
        pytest.skip("n % 8")
    q, ws = ops.fp8_quantize(x)
    assert float(ws[0].cpu()) == float(G[name + "/scale_a"])
    assert np.array_equal(q.view(torch.uint8).cpu().numpy().reshape(-1), G[name + "/x_fp8"].reshape(-1))


@pytest.mark.parametrize("name", ["per_tensor_bias", "per_row_bias", "per_row_col_nobias", "batched_3d_per_row", "zero_input"])
def test_linear_scaled_vs_reference_golden(ops, name):
    x = bf(G[name + "/x"]).reshape(SHAPES[name])
    K = x.shape[-1]
    if K % 128:
        pytest.skip("K % 128: the reference falls back to the dequantised plan below the MFMA's k-tile as well (K % 16)")
    w = torch.from_numpy(G[name + "/w"].copy()).view(torch.float8_e4m3fn)
    scale = torch.from_numpy(G[name + "/scale"].copy())
    bias = bf(G[name + "/bias"]) if name + "/bias" in G else None
    got = ops.linear_fp8(x.cuda(), w.cuda(), scale.reshape(-1).cuda(), None if bias is None else bias.cuda())
    ref = bf(G[name + "/out_scaled"]).reshape(*x.shape[:-1], w.shape[0])
    close(got, ref, floor=0.25, what=name)


@pytest.mark.parametrize("per_row", [True, False])
def test_fp8_gemm_every_epilogue_at_tile_scale(ops, per_row):
    """17 x 16 = 272 tiles of 256x256, ragged M (the last y tile has 38 rows), K = 7 k-tiles of 128 (ring wrap-around),
    per-row and per-tensor weight scales, all four epilogues; oracle = oracle/fp8_oracle.py (bit-exact to the reference)."""
    g = torch.Generator().manual_seed(17 + per_row)
    M, N, K, B = 16 * 256 + 38, 4096, 896, 2
    x = (torch.randn(M, K, generator=g) * 1.3).to(BF)
    wq, ws = F.quantize_weight(torch.randn(N, K, generator=g) / K ** 0.5, per_row=per_row)
    b = (0.1 * torch.randn(N, generator=g)).to(BF)
    y = F.linear_scaled(x, wq, ws, b)
    xc, wc, sc, bc = x.cuda(), wq.cuda(), ws.reshape(-1).cuda(), b.cuda()
    xq = ops.fp8_quantize(xc)                                                   # one quantisation shared by every Linear of this input
    # per-row scale: the output goes through three bf16 roundings (product, * scale, + bias): a last-bit difference of the fp32
    # sum can move the FIRST by one ulp, which is then carried through -- the magnitude floor is that of the pre-bias value
    fl = (y.float() - b.float()).abs() + b.float().abs() + 0.25
    close(ops.linear_fp8(xc, wc, sc, bc, x_fp8=xq), y, fl, what="none")
    close(ops.linear_fp8(xc, wc, sc, bc, epilogue=1, x_fp8=xq), torch.nn.functional.gelu(y, approximate="tanh"), fl, ulps=3, what="gelu")
    r = torch.randn(M, N, generator=g).to(BF)
    mod = (torch.randn(1, 6, N, generator=g) / N ** 0.5).to(BF); e0 = (0.5 * torch.randn(B, 6, N, generator=g)).to(BF)
    rpb = M // B
    ref = torch.cat([torch.addcmul(r[i * rpb:(i + 1) * rpb], y[i * rpb:(i + 1) * rpb], (mod + e0[i:i + 1]).chunk(6, dim=1)[5][0])
                     for i in range(B)])
    got = ops.linear_fp8(xc, wc, sc, bc, epilogue=2, residual=r.cuda(), mod=mod.cuda(), e=e0.cuda(), gate_idx=5, x_fp8=xq)
    close(got, ref, r.float().abs() + fl, what="gate residual")
    vt = ops.linear_fp8(xc, wc, sc, bc, epilogue=3, x_fp8=xq)
    assert vt.shape[1] % 64 == 0 and (vt[:, M:] == 0).all()
    close(vt[:, :M], y.t(), fl.t(), what="V^T")
    # small M (the time-embedding / text Linears of an fp8 checkpoint): one ragged tile row
    close(ops.linear_fp8(xc[:5].contiguous(), wc, sc, bc), F.linear_scaled(x[:5], wq, ws, b), 0.25, what="M=5")


def test_fp8_plan_accuracy_vs_dequantised_weights(ops):
    """Context for whole-model comparisons: the fp8 x fp8 plan sits ~2.7e-2 from the dequantised-weight bf16 Linear."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(512, 1024, generator=g).to(BF)
    wq, ws = F.quantize_weight(torch.randn(768, 1024, generator=g) / 32)
    got = ops.linear_fp8(x.cuda(), wq.cuda(), ws.cuda()).float().cpu()
    fb = F.linear_fallback(x, wq, ws).float()
    r = ((got - fb).norm() / fb.norm()).item()
    assert 1e-2 < r < 5e-2, r


@pytest.mark.parametrize("tag,per_row", [("row", True), ("tensor", False)])
def test_fp8_checkpoint_forward_vs_reference_golden(tag, per_row):
    """WanModelHIP loaded with a scaled-fp8 checkpoint (fp8 block Linears + scale_weight, everything else bf16 / fp32) against
    tests/golden/forward_tiny_fp8.npz = the reference's WanModel running the reference's `_linear_scaled` (bit-exactly
    reproduced by the oracle on CPU).  The HIP path differs from it by fp32 summation order only, but a last-bit difference
    in front of an e4m3 quantisation (3 mantissa bits) is amplified to that format's step: the distance is expected at the
    scale of the fp8 plan's own noise (~1e-2), far below its distance to the bf16 checkpoint's output."""
    from oracle import wan_oracle as O
    from wan2gp_amd.model import WanModelHIP
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "forward_tiny_fp8.npz")))
    f, h, w = [int(v) for v in g["shape"]]
    cfg = O.make_config("tiny")
    Wb = O.synth_weights(cfg)
    W8 = O.quantize_checkpoint_fp8(Wb, per_row=per_row)
    lat, ctx, ctx_null, _ = O.synth_inputs(cfg, f, h, w)
    t = torch.tensor([int(g["t"][0])])
    m = WanModelHIP(dim=cfg.dim, ffn_dim=cfg.ffn_dim, num_heads=cfg.num_heads, num_layers=cfg.num_layers).load_state_dict(W8)
    outs = [o.cpu() for o in m([lat.cuda(), lat.cuda()], t=t, context=[ctx.cuda(), ctx_null.cuda()])]
    mb = WanModelHIP(dim=cfg.dim, ffn_dim=cfg.ffn_dim, num_heads=cfg.num_heads, num_layers=cfg.num_layers).load_state_dict(Wb)
    outs_bf16 = [o.cpu() for o in mb([lat.cuda(), lat.cuda()], t=t, context=[ctx.cuda(), ctx_null.cuda()])]
    for o, ob, key in zip(outs, outs_bf16, (f"cond_{tag}", f"uncond_{tag}")):
        ref = torch.from_numpy(g[key])
        d = ((o - ref).norm() / ref.norm()).item()
        dq = ((ob - ref).norm() / ref.norm()).item()
        print(f"[fp8 forward {key}] hip-fp8 vs reference-fp8 {d:.3e}; bf16 checkpoint vs reference-fp8 {dq:.3e}")
        assert torch.isfinite(o).all() and d <= 2.5e-2 and d < dq, (d, dq)


@pytest.mark.parametrize("per_row", [True, False], ids=["row", "tensor"])
def test_e5m2_checkpoint_runs_the_reference_fallback(per_row):
    """`scaled_float8_e5m2` checkpoints (shared/qtypes/scaled_fp8.py:17,34-49; refused until round 6): torch._scaled_mm multiplies no two
    e5m2 matrices, so the reference's probe (:197-221) sends every such Linear through `_linear_fallback` (:306-322) -- weights.to(bf16) *=
    scale.to(bf16), a bf16 matmul.  WanModelHIP forms that product once at load (model.dequantize_scaled_fp8: the reference's values, bit
    for bit) and runs bf16 Linears: the forward equals the same model loaded with the dequantised checkpoint exactly, and sits at the bf16
    plan's distance from the oracle's CPU forward of the e5m2 checkpoint."""
    from oracle import wan_oracle as O, fp8_oracle as F8
    from wan2gp_amd.model import WanModelHIP, dequantize_scaled_fp8
    cfg = O.make_config("tiny")
    Wb = O.synth_weights(cfg)
    W5 = O.quantize_checkpoint_fp8(Wb, per_row=per_row, fp8_dtype=torch.float8_e5m2)
    assert sum(v.dtype == torch.float8_e5m2 for v in W5.values()) == 10 * cfg.num_layers
    Wd = {}
    for k, v in W5.items():
        if v.dtype == torch.float8_e5m2:
            Wd[k] = F8.dequantize(v, W5[k[:-7] + ".scale_weight"])
            assert torch.equal(Wd[k], dequantize_scaled_fp8(v, W5[k[:-7] + ".scale_weight"]))
        elif not k.endswith(".scale_weight"):
            Wd[k] = v
    f, h, w = 3, 8, 8
    lat, ctx, ctx_null, _ = O.synth_inputs(cfg, f, h, w)
    t = torch.tensor([637])
    kw = dict(dim=cfg.dim, ffn_dim=cfg.ffn_dim, num_heads=cfg.num_heads, num_layers=cfg.num_layers)
    m5 = WanModelHIP(**kw).load_state_dict({("model.diffusion_model." + k): v for k, v in reversed(list(W5.items()))})   # scales in front of their weights, prefixed keys
    md = WanModelHIP(**kw).load_state_dict(Wd)
    o5 = [o.cpu() for o in m5([lat.cuda(), lat.cuda()], t=t, context=[ctx.cuda(), ctx_null.cuda()])]
    od = [o.cpu() for o in md([lat.cuda(), lat.cuda()], t=t, context=[ctx.cuda(), ctx_null.cuda()])]
    ref = O.dit_forward([lat, lat], t, [ctx, ctx_null], W5, cfg)
    refb = O.dit_forward([lat, lat], t, [ctx, ctx_null], Wd, cfg)
    for a, b, r, rb in zip(o5, od, ref, refb):
        assert torch.equal(a, b)
        d = ((a - r.float()).norm() / r.float().norm()).item()
        db = ((b - rb.float()).norm() / rb.float().norm()).item()
        print(f"[e5m2 forward] hip vs oracle (fallback Linears) {d:.3e}; dequantised bf16 checkpoint hip vs oracle {db:.3e}")
        assert torch.isfinite(a).all() and d <= 2e-2 and d <= 2.0 * db + 1e-3, (d, db)


@pytest.mark.parametrize("d,rows_per_slot,rows", [(5120, 300, 600), (1536, 37, 111), (256, 5, 60)], ids=["14B_two_streams", "1.3B_three_slots", "row_form_few_rows"])
def test_absmax_folded_into_the_layernorms_gives_the_same_quantisation(ops, d, rows_per_slot, rows):
    """Round 5: wan_ln_modulate_amax / wan_ln_affine_amax leave max |out| per stream in the quantisation slots, wan_fp8_quantize_pre reads
    it -- the rows, the scale and EVERY fp8 byte equal what wan_ln_modulate / wan_ln_affine followed by the two-pass wan_fp8_quantize give
    (both modulation forms: per-batch table and per-row; three widths; slots shorter than a workgroup's four rows)."""
    g = torch.Generator(device="cuda").manual_seed(d + rows)
    x = (torch.randn(rows, d, device="cuda", generator=g) * 2 + 0.3).to(BF)
    mod = (torch.randn(1, 6, d, device="cuda", generator=g) / d ** 0.5).to(BF)
    e = (0.5 * torch.randn(1, 6, d, device="cuda", generator=g)).to(BF)
    w = (1 + 0.05 * torch.randn(d, device="cuda", generator=g)).to(BF)
    b = (0.02 * torch.randn(d, device="cuda", generator=g)).to(BF)
    nslot = (rows + rows_per_slot - 1) // rows_per_slot
    for name, plain, folded in (("modulate", lambda: ops.ln_modulate(x, mod, e, 3, 4), lambda ws: ops.ln_modulate_amax(x, mod, e, 3, 4, ws, rows_per_slot)),
                                ("affine", lambda: ops.ln_affine(x, w, b), lambda ws: ops.ln_affine_amax(x, w, b, ws, rows_per_slot))):
        ref = plain()
        ws = torch.zeros(nslot * 64, dtype=torch.float32, device="cuda")
        got = folded(ws)
        assert torch.equal(got, ref), name
        for sl in range(nslot):
            part = ref[sl * rows_per_slot:(sl + 1) * rows_per_slot]
            assert ws[sl * 64 + 1].item() == part.float().abs().max().item(), (name, sl)
            if part.numel() % 8:
                continue
            q_ref, ws_ref = ops.fp8_quantize(part.contiguous())
            q_pre = ops.fp8_quantize_pre(part.contiguous(), ws[sl * 64:(sl + 1) * 64], 1)
            assert torch.equal(q_pre.view(torch.uint8), q_ref.view(torch.uint8)) and ws[sl * 64].item() == ws_ref[0].item(), (name, sl)


@pytest.mark.parametrize("M,N,K,per_row", [(16384, 4096, 1024, True), (18900, 13824, 5120, False), (512, 256, 256, True)],
                         ids=["tile_kernel_per_row_scale", "ffn0_14B_ragged_per_tensor", "older_kernel_then_absmax_pass"])
def test_absmax_folded_into_the_gelu_epilogue(ops, M, N, K, per_row):
    """wan_gemm_fp8_amax: ffn.0 + GELU on the fp8 MFMA that also leaves max |out| for ffn.2's quantisation -- the same output bits as
    wan_gemm_fp8, the exact maximum (tile kernel: in the epilogue, ragged last tile included; small shapes: the older kernel + one abs-max
    pass), and the same fp8 bytes through wan_fp8_quantize_pre as through the two-pass quantiser."""
    g = torch.Generator(device="cuda").manual_seed(M + N)
    x = torch.randn(M, K, device="cuda", generator=g).to(BF)
    wf = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    sc = (wf.abs().amax(dim=1, keepdim=True) / 448.0).clamp_min(1e-12) if per_row else (wf.abs().max() / 448.0).reshape(1, 1)
    w8 = (wf / sc).clamp(-448, 448).to(torch.float8_e4m3fn)
    scale = sc.reshape(-1).float().contiguous()
    bias = (0.1 * torch.randn(N, device="cuda", generator=g)).to(BF)
    xq = ops.fp8_quantize(x)
    ref = ops.linear_fp8(x, w8, scale, bias, epilogue=1, x_fp8=xq)
    slot = torch.zeros(64, dtype=torch.float32, device="cuda")
    got = ops.linear_fp8_gelu_amax(xq, w8, scale, bias, slot[2:3])
    torch.cuda.synchronize()
    assert torch.equal(got, ref)
    assert slot[2].item() == ref.float().abs().max().item()
    q_ref, ws_ref = ops.fp8_quantize(ref)
    q_pre = ops.fp8_quantize_pre(ref, slot, 2)
    assert torch.equal(q_pre.view(torch.uint8), q_ref.view(torch.uint8)) and slot[0].item() == ws_ref[0].item()
