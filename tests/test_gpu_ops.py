"""-m gpu: every libwanhip op (through the C ABI) against the CPU oracle / golden fixtures.

Tolerances (stated, per BASELINE north star "within a stated fp16 tolerance"):
  * element-wise fused ops reproduce the reference's bf16 rounding points; the only freedom is
    the fp32 reduction order, so results must equal the oracle up to 1 bf16 ulp (2^-8 relative,
    we allow 2^-7) on at most 1% of the elements, everything else bit-equal.
  * GEMM: fp32 accumulation in a different order -> <= 1 bf16 ulp of the exact result.
  * attention: P is rounded to bf16 before PV (as torch's CPU flash kernel does): abs err
    <= 1.5e-2 on O(1) outputs vs the fp32-exact softmax; mean abs err <= 2e-3.
"""
import os

import numpy as np
import pytest
import torch

from oracle import wan_oracle as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from wan2gp_amd import ops as _ops, lib
    lib.load()
    return _ops


def cu(t):
    return t.cuda().contiguous()


def assert_bf16_close(got, ref, frac=0.01, ulps=2, what="", floor=None):
    """got/ref: tensors holding bf16-representable values.  An element may differ from the oracle
    by `ulps` bf16 ulps (1 ulp is 2^-8..2^-7 relative, we use 2^-7) of max(|ref|, floor): `floor`
    is the magnitude of the terms that were summed to produce it (cancellation makes a 1-ulp
    difference of an intermediate bf16 value look large relative to a small result); default
    1e-3 of the largest |ref|."""
    got, ref = got.float().cpu(), ref.float().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{what}: non-finite output"
    diff = (got - ref).abs()
    if floor is None:
        floor = 1e-3 * ref.abs().max().item()
    tol = torch.maximum(ref.abs(), torch.as_tensor(floor, dtype=torch.float32)) * (2.0 ** -7) * ulps
    bad = diff > tol
    assert not bad.any(), f"{what}: {int(bad.sum())} elements beyond {ulps} bf16 ulp, worst {diff.max().item()} at {np.unravel_index(int(diff.argmax()), diff.shape)}"
    neq = (got != ref).float().mean().item()
    assert neq <= frac, f"{what}: {neq * 100:.2f}% elements differ from the oracle (> {frac * 100}%)"


# ---------------------------------------------------------------------------------------------
def _ops_inputs():
    g = torch.Generator().manual_seed(7)
    L, H, D = 72, 2, 128
    x = torch.randn(1, L, H * D, generator=g).to(BF)
    w = (1 + 0.02 * torch.randn(H * D, generator=g)).to(BF)
    b3 = (0.01 * torch.randn(H * D, generator=g)).to(BF)
    v = torch.randn(1, L, H, D, generator=g).to(BF)
    return x, w, b3, v


def test_rmsnorm_rope_vs_reference_golden(ops):
    gold = dict(np.load(os.path.join(G, "ops.npz")))
    x, w, b3, v = _ops_inputs()
    cos, sin = O.rope_tables((3, 4, 6))
    q = cu(x.clone()); k = cu(torch.flip(x, dims=[1]).clone())
    ops.rmsnorm_rope_(q, k, cu(w), cu(w), (cu(cos), cu(sin)))
    assert_bf16_close(q.view(1, 72, 2, 128), torch.from_numpy(gold["rope_q_bf16"]), what="q")
    assert_bf16_close(k.view(1, 72, 2, 128), torch.from_numpy(gold["rope_k_bf16"]), what="k")
    # norm only (cross-attention q path)
    q2 = cu(x.clone())
    ops.rmsnorm_rope_(q2, None, cu(w), None, None)
    assert_bf16_close(q2, torch.from_numpy(gold["rms_bf16"]), what="rms only")


@pytest.mark.parametrize("d,H,world,chunks,rope", [(5120, 40, 8, 2, True), (5120, 40, 4, 2, True), (5120, 40, 8, 1, False), (1536, 12, 4, 3, True),
                                                  (1536, 12, 2, 1, True), (512, 4, 2, 2, True), (5120, 40, 2, 8, True)])
def test_rmsnorm_rope_pack_equals_norm_then_repack(ops, d, H, world, chunks, rope):
    """wan_rmsnorm_rope_pack (round 6): the norm kernel writes the Ulysses send layout itself -- bit for bit what the in-place kernel
    followed by the per-chunk re-packs (wan_permute16_ex, the round-5 path) leaves: [rows][world][Hn 128] -> [chunk][world][rows][Wc].
    Ragged row counts (not a multiple of the 4 rows per workgroup), both register forms (persistent at d = 5120, one row per wave
    below), the softmax scale folded into q, positions offset by the shard's first token; the source rows stay untouched."""
    g = torch.Generator().manual_seed(d + world + chunks)
    S, Ll = 2, 37 if d == 5120 else 50
    rows, Hn = S * Ll, H // world
    x = (torch.randn(S, Ll, d, generator=g) * 1.7).to(BF)
    w = (1 + 0.05 * torch.randn(d, generator=g)).to(BF)
    pos0 = 3 * Ll
    cos, sin = O.rope_tables((8, 4, 7))                       # 224 positions >= pos0 + Ll
    freqs = (cu(cos), cu(sin)) if rope else None
    scale = ops.attention_qscale()
    ref = cu(x.clone())
    ops.rmsnorm_rope_(ref, None, cu(w), None, freqs, L=Ll, pos0=pos0, q_scale=scale)
    want = torch.empty(rows * d, dtype=BF, device="cuda")
    C = min(chunks, Hn)
    Wd = Hn * 128
    for j in range(C):
        h0, h1 = j * Hn // C, (j + 1) * Hn // C
        wc, o0 = (h1 - h0) * 128, h0 * 128
        ops.permute16_ex(ref.view(-1)[o0:], want[o0 * rows * world:], rows, world, wc * 2, d * 2, Wd * 2, wc * 2, rows * wc * 2)
    src = cu(x.clone())
    keep = src.clone()
    got = ops.rmsnorm_rope_pack(src, cu(w), world, Hn, C, freqs, L=Ll, pos0=pos0, scale=scale)
    assert torch.equal(got, want), (got.float() - want.float()).abs().max().item()
    assert torch.equal(src, keep)


@pytest.mark.parametrize("d,H", [(1536, 12), (5120, 40), (256, 2)])
def test_rmsnorm_rope_model_widths(ops, d, H):
    g = torch.Generator().manual_seed(d)
    B, f, hh, ww = 2, 2, 3, 5
    L = f * hh * ww
    cos, sin = O.rope_tables((f, hh, ww))
    q = (torch.randn(B, L, d, generator=g) * 1.7).to(BF); k = torch.randn(B, L, d, generator=g).to(BF)
    wq = (1 + 0.05 * torch.randn(d, generator=g)).to(BF); wk = (1 + 0.05 * torch.randn(d, generator=g)).to(BF)
    rq = O.rope_apply(O.rms_norm(q, wq, 1e-6).view(B, L, H, 128), cos, sin)
    rk = O.rope_apply(O.rms_norm(k, wk, 1e-6).view(B, L, H, 128), cos, sin)
    gq, gk = cu(q.clone()), cu(k.clone())
    ops.rmsnorm_rope_(gq, gk, cu(wq), cu(wk), (cu(cos), cu(sin)))
    assert_bf16_close(gq.view(B, L, H, 128), rq, what="q")
    assert_bf16_close(gk.view(B, L, H, 128), rk, what="k")
    # sequence-parallel shard: rows [L/2, L) with pos0 = L/2 must equal the slice of the full result
    half = L // 2
    sq = cu(q[:1, half:].clone()); sk = cu(k[:1, half:].clone())
    ops.rmsnorm_rope_(sq, sk, cu(wq), cu(wk), (cu(cos), cu(sin)), L=L - half, pos0=half)
    assert torch.equal(sq.cpu(), gq[:1, half:].cpu()) and torch.equal(sk.cpu(), gk[:1, half:].cpu())


@pytest.mark.parametrize("rows", [2025, 2026, 2027, 5, 1])
def test_rmsnorm_rope_persist_ragged_rows(ops, rows):
    """d = 5120, k = None (what csrc/dit.hip always passes), rows % 4 != 0 and rows below the persistent grid's resident set: the last
    block has waves without a row, which must still stage their share of the norm weights into LDS (round-4 advisor finding: the
    trailing rows were normalised with uninitialised LDS).  Checked against the oracle AND bit-identical to the same rows inside a
    launch whose row count is a multiple of 4 (every wave alive)."""
    d, H = 5120, 40
    g = torch.Generator().manual_seed(rows)
    pad = (rows + 3) // 4 * 4
    f, hh, ww = 1, 45, 46                                   # 2070 positions >= pad
    cos, sin = O.rope_tables((f, hh, ww))
    q = (torch.randn(1, pad, d, generator=g) * 1.3).to(BF)
    wq = (1 + 0.05 * torch.randn(d, generator=g)).to(BF)
    for rope in (True, False):
        fr = (cu(cos), cu(sin)) if rope else None
        ref = O.rms_norm(q[:, :rows], wq, 1e-6)
        if rope:
            ref = O.rope_apply(ref.view(1, rows, H, 128), cos[:rows], sin[:rows]).reshape(1, rows, d)
        got = cu(q[:, :rows].clone())
        ops.rmsnorm_rope_(got, None, cu(wq), None, fr, L=f * hh * ww)
        full = cu(q.clone())
        ops.rmsnorm_rope_(full, None, cu(wq), None, fr, L=f * hh * ww)
        assert torch.equal(got.cpu(), full[:, :rows].cpu()), f"rows={rows} rope={rope}: ragged launch differs from the padded one"
        # (the rotation x0 cos - x1 sin cancels: a 1-ulp difference of a normalised bf16 value, |x| ~ 1.3, shows against a small result --
        # the floor is the magnitude of the summed terms, as assert_bf16_close's docstring says)
        assert_bf16_close(got, ref, what=f"rows={rows} rope={rope}", floor=1.0 if rope else None)


@pytest.mark.parametrize("d", [256, 1536, 5120])
def test_layernorm_family(ops, d):
    g = torch.Generator().manual_seed(d + 1)
    B, L = 2, 37
    x = (torch.randn(B, L, d, generator=g) * 2 + 0.3).to(BF)
    mod = (torch.randn(1, 6, d, generator=g) / d ** 0.5).to(BF)
    e0 = (0.5 * torch.randn(B, 6, d, generator=g)).to(BF)
    w = (1 + 0.05 * torch.randn(d, generator=g)).to(BF); b = (0.02 * torch.randn(d, generator=g)).to(BF)
    for sh, sc in ((0, 1), (3, 4)):
        ref = []
        for bi in range(B):
            e = (mod + e0[bi:bi + 1]).chunk(6, dim=1)
            y = O.layer_norm(x[bi:bi + 1], 1e-6)
            y = y * (1 + e[sc]); y = y + e[sh]
            ref.append(y)
        got = ops.ln_modulate(cu(x), cu(mod), cu(e0), sh, sc)
        assert_bf16_close(got, torch.cat(ref), what=f"ln_modulate {sh},{sc}")
    got = ops.ln_affine(cu(x), cu(w), cu(b))
    assert_bf16_close(got, O.layer_norm(x, 1e-6, w, b), what="ln_affine")
    y = torch.randn(B, L, d, generator=g).to(BF)
    ref = torch.cat([torch.addcmul(x[bi:bi + 1], y[bi:bi + 1], (mod + e0[bi:bi + 1]).chunk(6, dim=1)[2]) for bi in range(B)])
    xx = cu(x.clone())
    ops.gated_residual_(xx, cu(y), cu(mod), cu(e0), 2)
    assert_bf16_close(xx, ref, what="gated residual")
    xx = cu(x.clone())
    ops.gated_residual_(xx, cu(y))
    assert_bf16_close(xx, x + y, what="plain residual")


@pytest.mark.parametrize("d", [1536, 5120])
def test_ln_modulate_table_form_is_bit_identical_to_the_per_row_form(ops, d):
    """wan_ln_modulate derives the two modulation vectors once per batch (mod_table_kernel + layernorm_kernel<3>) when a batch has
    >= 64 rows, and per row (layernorm_kernel<0>) below that: the same bf16 operations, so the same bits.  37 rows per batch take
    the per-row form, the same rows followed by 91 more take the table form; also against the oracle, with per-frame batches."""
    g = torch.Generator().manual_seed(d + 9)
    B, Ls, Lb = 3, 37, 128
    xb = (torch.randn(B, Lb, d, generator=g) * 2 + 0.3).to(BF)
    mod = (torch.randn(1, 6, d, generator=g) / d ** 0.5).to(BF)
    e0 = (0.5 * torch.randn(B, 6, d, generator=g)).to(BF)
    for sh, sc in ((0, 1), (3, 4)):
        small = ops.ln_modulate(cu(xb[:, :Ls].contiguous()), cu(mod), cu(e0), sh, sc)
        big = ops.ln_modulate(cu(xb), cu(mod), cu(e0), sh, sc)
        assert torch.equal(small.cpu(), big.cpu()[:, :Ls]), f"table form differs from the per-row form ({sh},{sc})"
        ref, mag = [], []
        for bi in range(B):
            e = (mod + e0[bi:bi + 1]).chunk(6, dim=1)
            y = O.layer_norm(xb[bi:bi + 1], 1e-6)
            ref.append(y * (1 + e[sc]) + e[sh])
            mag.append((y * (1 + e[sc])).float().abs() + e[sh].float().abs())       # the two terms cancel in places: ulps of the terms
        assert_bf16_close(big, torch.cat(ref), what=f"ln_modulate table form {sh},{sc}", floor=torch.cat(mag))


def test_layernorm_golden(ops):
    gold = dict(np.load(os.path.join(G, "ops.npz")))
    x, w, b3, v = _ops_inputs()
    assert_bf16_close(ops.ln_affine(cu(x), cu(w), cu(b3)), torch.from_numpy(gold["ln3_bf16"]), what="ln3 golden")


# ---------------------------------------------------------------------------------------------
def _gemm_ref(x, w, b):
    return (x.float() @ w.float().t() + b.float()).to(BF)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 256, 128), (72, 512, 256), (1, 256, 256), (517, 1536, 1536),
                                   (130, 8960, 64), (257, 128, 8960)])
def test_gemm_plain_and_gelu(ops, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N)
    x = torch.randn(M, K, generator=g).to(BF); w = (torch.randn(N, K, generator=g) / K ** 0.5).to(BF)
    b = (0.1 * torch.randn(N, generator=g)).to(BF)
    ref = _gemm_ref(x, w, b)
    fl = 0.25          # outputs are N(0,1)-scaled dot products: terms of magnitude ~1 cancel
    got = ops.linear(cu(x), cu(w), cu(b))
    assert_bf16_close(got, ref, frac=0.05, what="gemm none", floor=fl)
    got = ops.linear(cu(x), cu(w), cu(b), epilogue=1)
    assert_bf16_close(got, torch.nn.functional.gelu(ref, approximate="tanh"), frac=0.05, ulps=3, what="gemm gelu", floor=fl)
    got = ops.linear(cu(x), cu(w), None)
    assert_bf16_close(got, _gemm_ref(x, w, torch.zeros(N)), frac=0.05, what="gemm no bias", floor=fl)


@pytest.mark.parametrize("M,N,K,B", [(200, 256, 128, 2), (96, 1536, 256, 1)])
def test_gemm_gate_residual(ops, M, N, K, B):
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(BF); w = (torch.randn(N, K, generator=g) / K ** 0.5).to(BF)
    b = (0.1 * torch.randn(N, generator=g)).to(BF)
    r = torch.randn(M, N, generator=g).to(BF)
    mod = (torch.randn(1, 6, N, generator=g) / N ** 0.5).to(BF); e0 = (0.5 * torch.randn(B, 6, N, generator=g)).to(BF)
    y = _gemm_ref(x, w, b)
    rpb = M // B
    ref = torch.cat([torch.addcmul(r[i * rpb:(i + 1) * rpb], y[i * rpb:(i + 1) * rpb], (mod + e0[i:i + 1]).chunk(6, dim=1)[5][0])
                     for i in range(B)])
    rr = cu(r.clone())
    got = ops.linear(cu(x), cu(w), cu(b), epilogue=2, residual=rr, mod=cu(mod), e=cu(e0), gate_idx=5, out=rr)
    assert_bf16_close(got, ref, frac=0.05, what="gemm gate residual (in place)", floor=(r.float().abs() + y.float().abs()))
    got = ops.linear(cu(x), cu(w), cu(b), epilogue=2, residual=cu(r), gate_idx=-1)
    assert_bf16_close(got, r + y, frac=0.05, what="gemm plain residual", floor=(r.float().abs() + y.float().abs()))


@pytest.mark.parametrize("M,N,K", [(72, 256, 256), (200, 256, 128), (4095, 1536, 256), (512, 256, 64)])
def test_gemm_transposed_vt(ops, M, N, K):
    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, K, generator=g).to(BF); w = (torch.randn(N, K, generator=g) / K ** 0.5).to(BF)
    b = (0.1 * torch.randn(N, generator=g)).to(BF)
    vt = ops.linear(cu(x), cu(w), cu(b), epilogue=3)
    ldv = vt.shape[1]
    assert ldv % 64 == 0 and ldv >= M
    assert_bf16_close(vt[:, :M], _gemm_ref(x, w, b).t(), frac=0.05, what="V^T", floor=0.25)
    assert (vt[:, M:] == 0).all(), "padding columns must stay zero"


@pytest.mark.parametrize("K", [128, 64, 448, 704])
def test_gemm_256x256_kernel_default_path(ops, K):
    """>= 256 tiles of 256x256 select gemm256k.hip (one wave per SIMD, accumulators in the accumulator file, k-tiles of 64
    in a ring of five 32-KB units) without any environment override: ragged M (the last y tile has 37 rows), every
    epilogue, two batches for the gate, and the transposed V^T form with a ragged token count; K = 1, 2, 7 and 11 k-tiles
    (prologue-only, clamped streams, ring wrap-around).  References as in the small-shape tests."""
    g = torch.Generator().manual_seed(256 + K)
    M, N, B = 16 * 256 + 37 + 1, 4096, 2                # 17 x 16 = 272 tiles; M even so that rows split into 2 batches
    x = torch.randn(M, K, generator=g).to(BF); w = (torch.randn(N, K, generator=g) / K ** 0.5).to(BF)
    b = (0.1 * torch.randn(N, generator=g)).to(BF)
    y = _gemm_ref(x, w, b)
    assert_bf16_close(ops.linear(cu(x), cu(w), cu(b)), y, frac=0.05, what="gemm256 none", floor=0.25)
    assert_bf16_close(ops.linear(cu(x), cu(w), cu(b), epilogue=1), torch.nn.functional.gelu(y, approximate="tanh"), frac=0.05,
                      ulps=3, what="gemm256 gelu", floor=0.25)
    r = torch.randn(M, N, generator=g).to(BF)
    mod = (torch.randn(1, 6, N, generator=g) / N ** 0.5).to(BF); e0 = (0.5 * torch.randn(B, 6, N, generator=g)).to(BF)
    rpb = M // B
    ref = torch.cat([torch.addcmul(r[i * rpb:(i + 1) * rpb], y[i * rpb:(i + 1) * rpb], (mod + e0[i:i + 1]).chunk(6, dim=1)[5][0])
                     for i in range(B)])
    rr = cu(r.clone())
    got = ops.linear(cu(x), cu(w), cu(b), epilogue=2, residual=rr, mod=cu(mod), e=cu(e0), gate_idx=5, out=rr)
    assert_bf16_close(got, ref, frac=0.05, what="gemm256 gate residual (in place)", floor=(r.float().abs() + y.float().abs()))
    # transposed: weights are the y operand (4096 rows = 16 y tiles), tokens the x operand (ragged: 4134 = 16 x tiles + 38)
    vt = ops.linear(cu(x), cu(w), cu(b), epilogue=3)
    assert_bf16_close(vt[:, :M], y.t(), frac=0.05, what="gemm256 V^T", floor=0.25)
    assert (vt[:, M:] == 0).all(), "padding columns must stay zero"


@pytest.mark.parametrize("tile", [128, 256])
@pytest.mark.parametrize("M,N,K", [(6400, 1536, 1536), (517, 1536, 128), (1024, 5120, 5120), (300, 272, 128), (3200, 1536, 8960), (33, 64, 192)])
def test_gemm16s_small_tile_kernel(ops, tile, M, N, K):
    """csrc/gemm16s.hip (round 6: 128 x 128 / 256 x 128 x 32 tiles on the 16x16x32 MFMA, two or three workgroups per CU, the kernel of
    every problem below one wave of 256 x 256 tiles) forced onto both tile heights: BASELINE configs[0]'s projections (M = 6,400,
    d = 1,536, ffn 8,960), the 14B text K / V Linears, ragged M and N (N = 272: the last x tile holds 16 columns), K = 4 k-tiles (the
    prologue + one) ... 280; every epilogue, two batches for the gate, the transposed V^T form.  References: the fp32 matmul rounded
    once (<= 2 bf16 ulp), as for the other generations; and the automatic dispatch must agree with the forced tile to the same bar."""
    from wan2gp_amd import lib as L_
    lib = L_.load()
    g = torch.Generator().manual_seed(M * 3 + N + K + tile)
    x = torch.randn(M, K, generator=g).to(BF); w = (torch.randn(N, K, generator=g) / K ** 0.5).to(BF)
    b = (0.1 * torch.randn(N, generator=g)).to(BF)
    y = _gemm_ref(x, w, b)
    old = lib.wan_gemm_debug_force16s(tile)
    try:
        assert_bf16_close(ops.linear(cu(x), cu(w), cu(b)), y, frac=0.05, what="gemm16s none", floor=0.25)
        assert_bf16_close(ops.linear(cu(x), cu(w), None), _gemm_ref(x, w, torch.zeros(N)), frac=0.05, what="gemm16s no bias", floor=0.25)
        assert_bf16_close(ops.linear(cu(x), cu(w), cu(b), epilogue=1), torch.nn.functional.gelu(y, approximate="tanh"), frac=0.05,
                          ulps=3, what="gemm16s gelu", floor=0.25)
        r = torch.randn(M, N, generator=g).to(BF)
        B = 2 if (M % 2 == 0 and M // 2 >= 256) else 1
        mod = (torch.randn(1, 6, N, generator=g) / N ** 0.5).to(BF); e0 = (0.5 * torch.randn(B, 6, N, generator=g)).to(BF)
        rpb = M // B
        ref = torch.cat([torch.addcmul(r[i * rpb:(i + 1) * rpb], y[i * rpb:(i + 1) * rpb], (mod + e0[i:i + 1]).chunk(6, dim=1)[5][0])
                         for i in range(B)])
        rr = cu(r.clone())
        got = ops.linear(cu(x), cu(w), cu(b), epilogue=2, residual=rr, mod=cu(mod), e=cu(e0), gate_idx=5, out=rr)
        assert_bf16_close(got, ref, frac=0.05, what="gemm16s gate residual (in place)", floor=(r.float().abs() + y.float().abs()))
        got = ops.linear(cu(x), cu(w), cu(b), epilogue=2, residual=cu(r), gate_idx=-1)
        assert_bf16_close(got, r + y, frac=0.05, what="gemm16s plain residual", floor=(r.float().abs() + y.float().abs()))
        vt = ops.linear(cu(x), cu(w), cu(b), epilogue=3)
        assert_bf16_close(vt[:, :M], y.t(), frac=0.05, what="gemm16s V^T", floor=0.25)
        assert (vt[:, M:] == 0).all(), "padding columns must stay zero"
    finally:
        lib.wan_gemm_debug_force16s(old)
    assert_bf16_close(ops.linear(cu(x), cu(w), cu(b)), y, frac=0.05, what="automatic dispatch", floor=0.25)


@pytest.mark.parametrize("rows", [160, 192, 224])
@pytest.mark.parametrize("M,N,K", [(6400, 1536, 1536), (517, 1536, 128), (3200, 1536, 8960), (1000, 512, 448)])
def test_gemm256m_lower_tile_heights(ops, rows, M, N, K):
    """csrc/gemm256m.hip with 160 / 192 / 224-row tiles (round 6: template argument TY = 5, 6, 7; the height wan_gemm256m_tile_rows picks
    when it leaves fewer CUs idle than 256 rows -- BASELINE configs[0]) forced onto configs[0]'s projections and ragged shapes (M = 517:
    the last y tile holds 37 / 133 / 69 rows; K = 2 ... 140 k-tiles): every epilogue, two batches for the gate, the V^T form; against the
    fp32 matmul rounded once AND bit for bit against the 256-row tile (same k order per output element)."""
    from wan2gp_amd import lib as L_
    lib = L_.load()
    g = torch.Generator().manual_seed(M * 5 + N + K + rows)
    x = torch.randn(M, K, generator=g).to(BF); w = (torch.randn(N, K, generator=g) / K ** 0.5).to(BF)
    b = (0.1 * torch.randn(N, generator=g)).to(BF)
    y = _gemm_ref(x, w, b)
    r = torch.randn(M, N, generator=g).to(BF)
    B = 2 if (M % 2 == 0 and M // 2 >= 256) else 1
    mod = (torch.randn(1, 6, N, generator=g) / N ** 0.5).to(BF); e0 = (0.5 * torch.randn(B, 6, N, generator=g)).to(BF)
    rpb = M // B
    ref_g = torch.cat([torch.addcmul(r[i * rpb:(i + 1) * rpb], y[i * rpb:(i + 1) * rpb], (mod + e0[i:i + 1]).chunk(6, dim=1)[5][0]) for i in range(B)])

    def run_all():
        out = {"none": ops.linear(cu(x), cu(w), cu(b)).cpu(), "gelu": ops.linear(cu(x), cu(w), cu(b), epilogue=1).cpu()}
        rr = cu(r.clone())
        out["gate"] = ops.linear(cu(x), cu(w), cu(b), epilogue=2, residual=rr, mod=cu(mod), e=cu(e0), gate_idx=5, out=rr).cpu()
        out["vt"] = ops.linear(cu(x), cu(w), cu(b), epilogue=3).cpu()
        return out
    old16 = lib.wan_gemm_debug_force16s(-1)
    old = lib.wan_gemm_debug_force_tile_rows(rows)
    try:
        got = run_all()
        lib.wan_gemm_debug_force_tile_rows(256)
        full = run_all()
    finally:
        lib.wan_gemm_debug_force_tile_rows(old)
        lib.wan_gemm_debug_force16s(old16)
    assert_bf16_close(got["none"], y, frac=0.05, what="none", floor=0.25)
    assert_bf16_close(got["gelu"], torch.nn.functional.gelu(y, approximate="tanh"), frac=0.05, ulps=3, what="gelu", floor=0.25)
    assert_bf16_close(got["gate"], ref_g, frac=0.05, what="gate residual", floor=(r.float().abs() + y.float().abs()))
    assert_bf16_close(got["vt"][:, :M], y.t(), frac=0.05, what="V^T", floor=0.25)
    assert (got["vt"][:, M:] == 0).all()
    for k in got:
        assert torch.equal(got[k], full[k]), f"{k}: the {rows}-row tile differs from the 256-row tile"


def test_gemm_rejects_bad_k(ops):
    from wan2gp_amd.lib import WanHipError
    x = torch.zeros(8, 96, dtype=BF).cuda(); w = torch.zeros(128, 96, dtype=BF).cuda()
    with pytest.raises(WanHipError):
        ops.linear(x, w, None)


# ---------------------------------------------------------------------------------------------
def attn_ok(got, ref, atol=1.5e-2):
    """The attention tolerance of this file: |err| <= 1.5e-2, or one coarse bf16 ulp (2^-7 |ref|) where |ref| >= 2."""
    got, ref = got.float().cpu(), ref.float().cpu()
    return bool(torch.isfinite(got).all() and ((got - ref).abs() <= torch.clamp(ref.abs() * 2.0 ** -7, min=atol)).all())


def _attn_check(ops, q, k, v, what, atol=1.5e-2):
    """|err| <= 1.5e-2 for outputs below 2 in magnitude; one coarse bf16 ulp (2^-7 |ref|) above: every entry point now runs the
    4x64 kernel, whose q carries the softmax scale in its bf16 rounding (q * c rounded instead of q), which can move an output
    by one bf16 ulp -- 0.0156 for |o| in [2, 4)."""
    ref = O.attention(q, k, v, exact=True).float()
    vt = ops.transpose_v(cu(v))
    got = ops.attention(cu(q), cu(k), vt).float().cpu()
    assert torch.isfinite(got).all(), what
    err = (got - ref).abs()
    assert (err <= torch.clamp(ref.abs() * 2.0 ** -7, min=atol)).all(), f"{what}: max abs err {err.max().item()}"
    assert err.mean().item() <= 2e-3, f"{what}: mean abs err {err.mean().item()}"
    return got


def test_attention_golden_sdpa(ops):
    gold = dict(np.load(os.path.join(G, "ops.npz")))
    x, w, b3, v = _ops_inputs()
    cos, sin = O.rope_tables((3, 4, 6))
    q = O.rope_apply(O.rms_norm(x, w, 1e-6).view(1, 72, 2, 128), cos, sin)
    k = O.rope_apply(O.rms_norm(torch.flip(x, dims=[1]), w, 1e-6).view(1, 72, 2, 128), cos, sin)
    got = _attn_check(ops, q, k, v, "golden")
    ref = torch.from_numpy(gold["sdpa_bf16"])        # the reference's own sdpa output (bf16)
    assert (got - ref).abs().max().item() <= 3e-2


@pytest.mark.parametrize("B,Lq,Lk,H,Bk", [(1, 128, 64, 1, 1), (2, 200, 333, 3, 2), (2, 131, 512, 2, 1), (1, 64, 1, 2, 1),
                                          (1, 1000, 1000, 2, 1), (1, 5, 7, 1, 1)])
def test_attention_shapes(ops, B, Lq, Lk, H, Bk):
    g = torch.Generator().manual_seed(Lq * 3 + Lk)
    q = torch.randn(B, Lq, H, 128, generator=g).to(BF); k = torch.randn(Bk, Lk, H, 128, generator=g).to(BF)
    v = torch.randn(Bk, Lk, H, 128, generator=g).to(BF)
    _attn_check(ops, q, k, v, f"B{B} Lq{Lq} Lk{Lk} H{H} Bk{Bk}")


def test_attention_forced_rescale(ops):
    """cdna guide §5.4 rule 26: spike one key against the queries so the running max jumps at a
    late tile; every accumulator must be rescaled exactly once."""
    g = torch.Generator().manual_seed(99)
    q = torch.randn(1, 64, 1, 128, generator=g).to(BF); k = torch.randn(1, 640, 1, 128, generator=g).to(BF)
    v = torch.randn(1, 640, 1, 128, generator=g).to(BF)
    k[0, 400] = (q[0, 3, 0].float() * 3.0).to(BF)      # huge score for q row 3 at kv 400 (tile 6)
    k[0, 130] = (q[0, 17, 0].float() * 2.0).to(BF)
    _attn_check(ops, q, k, v, "spiked")


def test_attention_row_normalisation_and_linearity(ops):
    """size-independent properties at a large size: V = 1 gives O = 1; O is linear in V."""
    g = torch.Generator().manual_seed(5)
    B, L, H = 1, 8192, 2
    q = cu(torch.randn(B, L, H, 128, generator=g).to(BF)); k = cu(torch.randn(B, L, H, 128, generator=g).to(BF))
    ones = torch.ones(B, L, H, 128, dtype=BF).cuda()
    o = ops.attention(q, k, ops.transpose_v(ones)).float()
    assert (o - 1).abs().max().item() <= 1e-2
    v1 = cu(torch.randn(B, L, H, 128, generator=g).to(BF)); v2 = cu(torch.randn(B, L, H, 128, generator=g).to(BF))
    o1 = ops.attention(q, k, ops.transpose_v(v1)).float(); o2 = ops.attention(q, k, ops.transpose_v(v2)).float()
    o12 = ops.attention(q, k, ops.transpose_v((v1.float() + v2.float()).to(BF))).float()
    assert (o12 - (o1 + o2)).abs().max().item() <= 3e-2


def test_attention_segments_match_contiguous(ops):
    """K/V split into 2 equal gathered segments (sequence-parallel layout) == contiguous K/V."""
    g = torch.Generator().manual_seed(8)
    S, Lq, Ll, H = 2, 96, 150, 2
    q = torch.randn(S, Lq, H, 128, generator=g).to(BF)
    k = torch.randn(2, S, Ll, H, 128, generator=g).to(BF)       # [seg][S][Ll]
    v = torch.randn(2, S, Ll, H, 128, generator=g).to(BF)
    kfull = torch.cat([k[0], k[1]], dim=1); vfull = torch.cat([v[0], v[1]], dim=1)
    ref = O.attention(q, kfull, vfull, exact=True).float()
    vt = torch.stack([ops.transpose_v(cu(v[0])), ops.transpose_v(cu(v[1]))])      # [seg][S][H*128][ldv]
    ldv = vt.shape[-1]
    got = ops.attention(cu(q), cu(k), vt.contiguous(), Lk=Ll, nseg=2, k_seg_stride=S * Ll * H * 128,
                        vt_seg_stride=S * H * 128 * ldv, Bk=S).float().cpu()
    assert attn_ok(got, ref), (got - ref).abs().max().item()


# ---- pre-scaled attention path (wan_rmsnorm_rope_scaled -> wan_attention_prescaled, csrc/attention_w64q.hip) --------
def _prescaled_ref(qs, k, v):
    """Exact attention for a q that already holds q * scale * log2(e): P = 2^(qs k^T - rowmax) (fp32)."""
    B, Lq, H, _ = qs.shape
    kk = k.float().expand(B, -1, -1, -1) if k.shape[0] != B else k.float()
    vv = v.float().expand(B, -1, -1, -1) if v.shape[0] != B else v.float()
    s = torch.einsum("bqhd,bkhd->bhqk", qs.float().double(), kk.double())
    p = torch.exp2(s - s.amax(dim=-1, keepdim=True))
    o = torch.einsum("bhqk,bkhd->bqhd", p / p.sum(dim=-1, keepdim=True), vv.double())
    return o.float()


@pytest.mark.parametrize("B,Lq,Lk,H,Bk", [(1, 128, 64, 1, 1), (2, 200, 333, 3, 2), (2, 131, 512, 2, 1), (1, 64, 1, 2, 1),
                                          (1, 1000, 1000, 2, 1), (1, 5, 7, 1, 1), (1, 300, 4160, 1, 1)])
def test_attention_prescaled_shapes(ops, B, Lq, Lk, H, Bk):
    """Same rounding points as the kernel (q*scale rounded to bf16 once): tolerance as for the exact-scale kernels.
    Against the true oracle (scale applied to fp32 scores) the extra q rounding may move an output by one bf16 ulp:
    tolerance 2 bf16 ulps of max(|ref|, 1)."""
    g = torch.Generator().manual_seed(Lq * 3 + Lk)
    q = torch.randn(B, Lq, H, 128, generator=g).to(BF); k = torch.randn(Bk, Lk, H, 128, generator=g).to(BF)
    v = torch.randn(Bk, Lk, H, 128, generator=g).to(BF)
    qs = (q.float() * ops.attention_qscale()).to(BF)
    got = ops.attention(cu(qs), cu(k), ops.transpose_v(cu(v)), q_prescaled=True).float().cpu()
    assert torch.isfinite(got).all()
    ref = _prescaled_ref(qs, k, v)
    err = (got - ref).abs()
    assert (err <= torch.clamp(ref.abs() * 2.0 ** -7, min=1.5e-2)).all() and err.mean().item() <= 2e-3, (err.max().item(), err.mean().item())
    assert_bf16_close(got, O.attention(q, k, v, exact=True), frac=1.0, ulps=2, what="vs oracle", floor=1.0)


def test_attention_prescaled_forced_rescale_and_segments(ops):
    """Lazy reference max: a late spike far above the threshold must rescale O and l exactly once; a row whose scores are
    all very negative must not underflow; 2 gathered kv segments == contiguous."""
    g = torch.Generator().manual_seed(99)
    q = torch.randn(1, 64, 1, 128, generator=g).to(BF); k = torch.randn(1, 640, 1, 128, generator=g).to(BF)
    v = torch.randn(1, 640, 1, 128, generator=g).to(BF)
    k[0, 400] = (q[0, 3, 0].float() * 3.0).to(BF)
    k[0, 130] = (q[0, 17, 0].float() * 2.0).to(BF)
    q[0, 9] = (-4.0 * k[0, :, 0].float().mean(dim=0)).to(BF) - q[0, 9]        # a row with a distinctly shifted score range
    qs = (q.float() * ops.attention_qscale()).to(BF)
    vt = ops.transpose_v(cu(v))
    got = ops.attention(cu(qs), cu(k), vt, q_prescaled=True).float().cpu()
    err = (got - _prescaled_ref(qs, k, v)).abs()
    assert attn_ok(got, _prescaled_ref(qs, k, v)), err.max().item()
    S, Ll, H = 1, 320, 1
    k2 = k.view(2, S, Ll, H, 128); v2 = v.view(2, S, Ll, H, 128)
    vt2 = torch.stack([ops.transpose_v(cu(v2[0])), ops.transpose_v(cu(v2[1]))]).contiguous()
    got2 = ops.attention(cu(qs), cu(k2.contiguous()), vt2, Lk=Ll, nseg=2, k_seg_stride=S * Ll * H * 128,
                         vt_seg_stride=S * H * 128 * vt2.shape[-1], Bk=S, q_prescaled=True).float().cpu()
    assert attn_ok(got2, _prescaled_ref(qs, k, v))


def test_rmsnorm_rope_scaled(ops):
    """q_scale is applied in fp32 in front of q's single bf16 rounding; k is untouched."""
    g = torch.Generator().manual_seed(5)
    B, f, hh, ww, d, H = 1, 2, 3, 4, 256, 2
    L = f * hh * ww
    cos, sin = O.rope_tables((f, hh, ww))
    q = (torch.randn(B, L, d, generator=g) * 1.3).to(BF); k = torch.randn(B, L, d, generator=g).to(BF)
    wq = (1 + 0.05 * torch.randn(d, generator=g)).to(BF); wk = (1 + 0.05 * torch.randn(d, generator=g)).to(BF)
    c = ops.attention_qscale()
    assert abs(c - 128 ** -0.5 * 1.4426950408889634) < 1e-7
    g0, k0 = cu(q.clone()), cu(k.clone())
    ops.rmsnorm_rope_(g0, k0, cu(wq), cu(wk), (cu(cos), cu(sin)))
    g1, k1 = cu(q.clone()), cu(k.clone())
    ops.rmsnorm_rope_(g1, k1, cu(wq), cu(wk), (cu(cos), cu(sin)), q_scale=c)
    assert torch.equal(k0.cpu(), k1.cpu())
    # unrounded reference: fp32 RoPE of the bf16-rounded norm output (the kernel's rounding points), times c
    xn = O.rms_norm(q, wq, 1e-6).view(B, L, H, 128).float()
    cs, sn = cos.view(1, L, 1, 128), sin.view(1, L, 1, 128)
    rot = torch.stack([-xn[..., 1::2], xn[..., ::2]], dim=-1).flatten(-2)
    exact = (xn * cs + rot * sn) * c
    assert_bf16_close(g1.view(B, L, H, 128), exact.to(BF), frac=0.02, ulps=1, what="scaled q")


def test_pay_attention_dropin_contract(ops):
    g = torch.Generator().manual_seed(3)
    q = cu(torch.randn(2, 70, 2, 128, generator=g).to(BF)); k = cu(torch.randn(1, 512, 2, 128, generator=g).to(BF))
    v = cu(torch.randn(1, 512, 2, 128, generator=g).to(BF))
    ref = O.attention(q.cpu(), k.cpu(), v.cpu(), exact=True).float()
    lst = [q, k, v]
    out = ops.pay_attention(lst, recycle_q=True)
    assert lst == [] and out.dtype == BF and out.shape == q.shape          # list consumed (attention.py:403)
    assert out.data_ptr() == q.data_ptr()                                  # recycle_q reuses q's storage
    assert attn_ok(out, ref)
    from wan2gp_amd.lib import WanHipError
    with pytest.raises(WanHipError):
        ops.pay_attention([q, k, v], causal=True)


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["tiny", "tiny_i2v", "tiny_ti2v"])
def test_patch_embed_and_head(ops, name):
    cfg = O.make_config(name)
    W = O.synth_weights(cfg)
    f, h, w = 3, 8, 12
    lat, ctx, ctx_null, y = O.synth_inputs(cfg, f, h, w)
    xin = lat if y is None else torch.cat([lat, y.unsqueeze(0)], dim=1)
    ref, grid = O.patch_embed(xin, W, cfg, BF)
    got = ops.patch_embed(cu(lat), cu(W["patch_embedding.weight"]), cu(W["patch_embedding.bias"]), None if y is None else cu(y))
    assert_bf16_close(got, ref, frac=0.02, what="patch_embed", floor=0.05)
    g = torch.Generator().manual_seed(2)
    L = ref.shape[1]
    hid = torch.randn(1, L, cfg.dim, generator=g).to(BF); e = (0.3 * torch.randn(1, cfg.dim, generator=g)).to(BF)
    refh = O.unpatchify(O.head_forward(hid, e, W, cfg), grid, cfg).float()
    goth = ops.head(cu(hid), cu(W["head.modulation"].reshape(2, -1)), cu(e), cu(W["head.head.weight"]), cu(W["head.head.bias"]), grid)
    assert torch.allclose(goth.cpu(), refh, atol=2e-3, rtol=2e-3), (goth.cpu() - refh).abs().max()


def test_lincomb_and_cfg(ops):
    g = torch.Generator().manual_seed(1)
    ts = [torch.randn(1, 16, 3, 10, 7, generator=g) for _ in range(4)]
    cs = [0.3, -1.7, 2.0, 0.01]
    ref = sum(c * t for c, t in zip(cs, ts))
    got = ops.lincomb([cu(t) for t in ts], cs).cpu()
    assert torch.allclose(got, ref, atol=1e-5, rtol=1e-5)
    ref = ts[1] + 4.0 * (ts[0] - ts[1])
    assert torch.equal(ops.cfg_combine(cu(ts[0]), cu(ts[1]), 4.0).cpu(), ref)


# ---- the two tile loops of attn_w64q_kernel: bounded softmax (K pre-pass proves |s| <= 96 log2 units) / lazy-max tracking ----
@pytest.mark.parametrize("B,Lq,Lk,H,Bk", [(1, 300, 4160, 2, 1), (2, 520, 2200, 3, 2), (1, 64, 2049, 1, 1)])
def test_attention_bounded_and_tracking_loops_agree_with_fp64(ops, B, Lq, Lk, H, Bk):
    """The same inputs through (a) the bounded loop (caller scratch), (b) the library-scratch entry, (c) the tracking loop
    (no pre-pass): each within the attention tolerance of the fp64 softmax, and the scratch holds max |k_h|^2 afterwards."""
    g = torch.Generator().manual_seed(Lq + Lk)
    q = torch.randn(B, Lq, H, 128, generator=g).to(BF); k = torch.randn(Bk, Lk, H, 128, generator=g).to(BF)
    v = torch.randn(Bk, Lk, H, 128, generator=g).to(BF)
    qs = (q.float() * ops.attention_qscale()).to(BF)
    ref = _prescaled_ref(qs, k, v)
    vt = ops.transpose_v(cu(v))
    scratch = torch.full((ops.attention_scratch_words(B, Bk, Lq, H),), -1.0, device="cuda")
    assert scratch.numel() == 2 * Bk * H + (Lq + 255) // 256 * H * B          # maxima, flags, (sequence parallelism) the previous maxima
    outs = {"bounded": ops.attention(cu(qs), cu(k), vt, q_prescaled=True, kmax_scratch=scratch),
            "library scratch": ops.attention(cu(qs), cu(k), vt, q_prescaled=True),
            "tracking": ops.attention(cu(qs), cu(k), vt, q_prescaled=True, kmax_scratch=False)}
    kn = (k.float() ** 2).sum(-1).amax(dim=1).reshape(-1)                       # [Bk*H]
    assert torch.allclose(scratch[:Bk * H].cpu(), kn, rtol=1e-5), (scratch.cpu(), kn)
    assert (scratch[Bk * H:-Bk * H].view(torch.int32) == 0).all()              # no workgroup had to fall back to the tracking loop
    for name, o in outs.items():
        err = (o.float().cpu() - ref).abs()
        assert attn_ok(o, ref) and err.mean().item() <= 2e-3, (name, err.max().item())
    # unscaled q through the generic entry (in-kernel pre-scaling) takes the same kernel for long KV
    o = ops.attention(cu(q), cu(k), vt).float().cpu()
    assert attn_ok(o, O.attention(q, k, v, exact=True))


def test_attention_bound_exceeded_falls_back_per_workgroup(ops):
    """Rows whose scores cannot be exponentiated against one per-row reference must take the tracking loop -- per 256-row workgroup:
    q rows 256..511 scaled 80x (scores ~ N(0, 115^2) log2 units: the spread between the maximum of the first 64 keys and the maximum
    of all 2,300 exceeds the 176 units one reference covers, see attention_w16n.hip SHIFT), the other workgroups stay plain.  Then K
    scaled 70x on top: every workgroup is out of reach.  Both must match the fp64 softmax."""
    g = torch.Generator().manual_seed(77)
    B, Lq, Lk, H = 1, 700, 2300, 2
    q = torch.randn(B, Lq, H, 128, generator=g); k = torch.randn(B, Lk, H, 128, generator=g).to(BF)
    v = torch.randn(B, Lk, H, 128, generator=g).to(BF)
    q[:, 256:512] *= 80.0
    qs = (q * ops.attention_qscale()).to(BF)
    scratch = torch.zeros(ops.attention_scratch_words(B, B, Lq, H), device="cuda")
    got = ops.attention(cu(qs), cu(k), ops.transpose_v(cu(v)), q_prescaled=True, kmax_scratch=scratch).float().cpu()
    assert scratch[B * H:-B * H].view(torch.int32).view(B * H, -1).cpu().tolist() == [[0, 1, 0]] * (B * H)   # q-block 1 of every head fell back
    ref = _prescaled_ref(qs, k, v)
    assert torch.isfinite(got).all()
    err = (got - ref).abs()
    assert attn_ok(got, ref), err.max().item()
    # and a K so large that every workgroup is out of reach (q-block 1 beyond SHIFT_LIMIT: declined before the loop)
    k2 = (k.float() * 70).to(BF)
    got2 = ops.attention(cu(qs), cu(k2), ops.transpose_v(cu(v)), q_prescaled=True, kmax_scratch=scratch).float().cpu()
    assert (scratch[B * H:-B * H].view(torch.int32) == 1).all()
    assert attn_ok(got2, _prescaled_ref(qs, k2, v))


@pytest.mark.parametrize("layout,gain", [("one_segment", 1.0), ("one_segment", 8.0), ("ulysses_segments", 1.0), ("one_segment", 12.0), ("ulysses_segments", 12.0)])
def test_attention_split_tail_agrees_with_the_one_launch_form_and_fp64(ops, layout, gain):
    """The split tail of a bounded launch (round 6; csrc/attention_w64q.hip split_tail): 270 workgroups on 256 CUs = one full round + 14
    q blocks that would run alone for a whole round -- they are attended as 8 key-range parts each (unnormalised sums to the library's
    scratch) and finished by a third launch.  The bounded softmax's partial sums add exactly up to the order of the fp32 additions:
    against the fp64 softmax on the tail's rows and a sample of the others, and against the one-launch form (wan_attention_debug_split_tail
    0) within one bf16 ulp of the output scale.  gains 8 / 12: the rows carry a reference shift -- such a q block does not split (its parts could
    not use the sample of tile 0 and would hand diffuse rows to the tracking loop: the closing run of round 6 caught exactly that at gain 12):
    part 0 attends it whole, flag 3, which the finishing launch clears -- no flag survives, nothing is redone.  The segmented form is
    the Ulysses rank's launch (2 x 1 query batches against one K / V^T batch in two segments with ragged tails)."""
    from wan2gp_amd import lib as L_
    lib = L_.load()
    cus = lib.wan_device_cus()
    if cus != 256:
        pytest.skip(f"the shape is built for 256 CUs (device has {cus})")
    g = torch.Generator().manual_seed(int(gain * 10) + len(layout))
    H = 3
    if layout == "one_segment":
        B, Bk, Lq, Lk, nseg = 1, 1, 90 * 256, 8256, 1                      # 90 q blocks x 3 heads = 270 workgroups; 129 tiles, the last one ragged... (8256 = 129 x 64)
        Lk -= 19                                                           # ... now it is
        k = (torch.randn(Bk, Lk, H, 128, generator=g) * gain).to(BF); v = torch.randn(Bk, Lk, H, 128, generator=g).to(BF)
        kk, vv = k, v
    else:
        B, Bk, Lq, Lk, nseg = 2, 1, 45 * 256, 4141, 2                      # (source rank, stream) batches; two segments of 65 tiles, ragged tails
        kseg = (torch.randn(nseg, Bk, Lk, H, 128, generator=g) * gain).to(BF); vseg = torch.randn(nseg, Bk, Lk, H, 128, generator=g).to(BF)
        kk, vv = torch.cat(list(kseg), dim=1), torch.cat(list(vseg), dim=1)     # [Bk, nseg * Lk, H, 128]: what the rows attend
    q = torch.randn(B, Lq, H, 128, generator=g)
    qs = (q * ops.attention_qscale()).to(BF)

    def run():
        scratch = torch.zeros(ops.attention_scratch_words(B, Bk, Lq, H), device="cuda")
        if layout == "one_segment":
            out = ops.attention(cu(qs), cu(k), ops.transpose_v(cu(v)), q_prescaled=True, kmax_scratch=scratch)
        else:
            vt = torch.stack([ops.transpose_v(cu(vseg[i])) for i in range(nseg)]).contiguous()
            out = ops.attention(cu(qs), cu(kseg), vt, Lk=Lk, nseg=nseg, k_seg_stride=Bk * Lk * H * 128, vt_seg_stride=Bk * H * 128 * vt.shape[-1], Bk=Bk,
                                q_prescaled=True, kmax_scratch=scratch)
        flags = scratch[Bk * H:Bk * H + (Lq // 256) * H * B].view(torch.int32).cpu()
        return out.float().cpu(), flags
    old = lib.wan_attention_debug_split_tail(1)
    try:
        split, fl1 = run()
        lib.wan_attention_debug_split_tail(0)
        whole, fl0 = run()
    finally:
        lib.wan_attention_debug_split_tail(old)
    assert (fl1 == 0).all() and (fl0 == 0).all()                           # nothing handed to the tracking loop in either form
    assert torch.isfinite(split).all()
    # the tail: workgroups 256..269 = the last 14 q blocks of the last (batch, head) pair; plus two q blocks of the full round
    nqb = Lq // 256
    for (b, h, qb0, qb1) in ((B - 1, H - 1, nqb - 14, nqb), (0, 0, 3, 5)):
        rows = slice(qb0 * 256, qb1 * 256)
        ref = _prescaled_ref(qs[b:b + 1, rows, h:h + 1], kk[:, :, h:h + 1], vv[:, :, h:h + 1])
        for name, got in (("split", split), ("whole", whole)):
            part = got[b:b + 1, rows, h:h + 1]
            err = (part - ref).abs()
            assert attn_ok(part, ref) and err.mean().item() <= 2e-3, (layout, gain, name, qb0, err.max().item())
    d = (split - whole).abs()
    assert d.max().item() <= 2.0 ** -7 * max(1.0, whole.abs().max().item()), d.max().item()
    assert torch.equal(split[:, :(nqb - 14) * 256] if B == 1 else split[0], whole[:, :(nqb - 14) * 256] if B == 1 else whole[0])   # the full rounds' rows: the same launch form


@pytest.mark.parametrize("gain", [6.0, 8.0, 12.0, 30.0])
def test_attention_shifted_bounded_loop_agrees_with_fp64(ops, gain):
    """Round 4: rows beyond the bound U = |q~| max|k| <= 96 run the bounded loop with a per-row reference shift m = U - 96
    (attention_w16n.hip, SHIFT) instead of the tracking loop -- m = U - 96 while the row's maximum over the first 64 keys lies within
    168 of U (gains 6, 8), that sample maximum + 72 beyond (gains 12, 30).  K scaled by `gain` puts every row at U ~ 120 .. 600
    (round 3: all declined): the result must be the fp64 softmax's, no workgroup may reach the tracking loop, and the workgroup flags
    end at 0 (2 = "wants the shifted loop" never survives a call).  One head stays at gain 1: plain and shifted workgroups in one launch."""
    g = torch.Generator().manual_seed(int(gain) * 7)
    B, Lq, Lk, H = 2, 520, 4200, 3
    q = torch.randn(B, Lq, H, 128, generator=g).to(BF); k = torch.randn(B, Lk, H, 128, generator=g)
    v = torch.randn(B, Lk, H, 128, generator=g).to(BF)
    k[:, :, :2] *= gain
    k = k.to(BF)
    qs = (q.float() * ops.attention_qscale()).to(BF)
    u = (qs.float().norm(dim=-1).amax(dim=1) * k.float().norm(dim=-1).amax(dim=1))           # [B, H]: the largest bound of a head
    assert (u[:, :2] > 100).all() and (u[:, 2] < 96).all()
    ref = _prescaled_ref(qs, k, v)
    scratch = torch.zeros(ops.attention_scratch_words(B, B, Lq, H), device="cuda")
    got = ops.attention(cu(qs), cu(k), ops.transpose_v(cu(v)), q_prescaled=True, kmax_scratch=scratch)
    declined = int((scratch[B * H:-B * H].view(torch.int32) != 0).sum())
    # gain 30 is the edge for diffuse rows at this Lk (score spread 43 log2 units: a row whose full maximum lies ~3.5 spreads above its
    # first tile's leaves the window -- a few per thousand rows): a stray workgroup may be redone, never a wrong number
    assert declined == 0 if gain <= 12.0 else declined <= 2, declined
    err = (got.float().cpu() - ref).abs()
    assert attn_ok(got, ref) and err.mean().item() <= 2e-3, (err.max().item(), err.mean().item())
    # shift invariance: the tracking loop on the same tensors (two roundings to bf16 apart: twice the tolerance)
    trk = ops.attention(cu(qs), cu(k), ops.transpose_v(cu(v)), q_prescaled=True, kmax_scratch=False).float().cpu()
    assert ((got.float().cpu() - trk).abs() <= torch.clamp(trk.abs() * 2.0 ** -6, min=3e-2)).all()


def test_attention_shifted_rows_that_leave_the_exponent_range_are_redone_by_the_tracking_loop(ops):
    """The shifted loop does not watch the exponent range inside the loop; the row sum does.  Head 0: every query carries a component
    along e_0 and ONE key (not among the first 64) is 400 e_0 -- its score ~ 300 sits far above the reference the sample of the first
    tile suggests (m ~ 85), P overflows to inf, the row sums say so, the workgroups flag themselves AFTER the loop and the tracking
    launch redoes them.  Head 1: one key scaled x40 in a random direction (U ~ 650, but its scores ~ N(0, 58) stay inside the window
    of all but a row or two): handled by the shifted loop.  Both match the fp64 softmax."""
    g = torch.Generator().manual_seed(123)
    B, Lq, Lk, H = 1, 700, 4100, 2
    q = torch.randn(B, Lq, H, 128, generator=g); k = torch.randn(B, Lk, H, 128, generator=g)
    v = torch.randn(B, Lk, H, 128, generator=g).to(BF)
    q[:, :, 0, 0] += 6.0
    k[0, 1234, 0] = 0.0
    k[0, 1234, 0, 0] = 400.0
    k[0, 2345, 1] *= 40.0
    k = k.to(BF)
    qs = (q * ops.attention_qscale()).to(BF)
    scratch = torch.zeros(ops.attention_scratch_words(B, B, Lq, H), device="cuda")
    got = ops.attention(cu(qs), cu(k), ops.transpose_v(cu(v)), q_prescaled=True, kmax_scratch=scratch).float().cpu()
    flags = scratch[B * H:-B * H].view(torch.int32).view(B * H, -1).cpu()
    assert flags[0].tolist() == [1, 1, 1], flags.tolist()           # (head 1: a row whose score with the x40 key exceeds ~180 is redone too: 0-2 workgroups)
    assert attn_ok(got, _prescaled_ref(qs, k, v))


def _no_persist(on):
    """wan_attention_debug_no_persist: 0 the product dispatch, 1 / True one q block per workgroup, 2 the persistent walk also where the
    K / V^T-stationary kernel (attention_xkv.hip) would serve the call; returns the old value"""
    from wan2gp_amd import lib as L
    return L.load().wan_attention_debug_no_persist(int(on))


@pytest.mark.parametrize("B,Lq,Lk,H,Bk,gain", [(2, 3405, 512, 12, 2, 1.0), (2, 3405, 512, 12, 2, 8.0), (1, 700, 512, 3, 1, 1.0), (3, 1100, 500, 5, 1, 6.0),
                                                (1, 2100, 449, 7, 1, 1.0), (2, 900, 2048, 9, 2, 12.0), (2, 1300, 1100, 8, 2, 1.0)],
                         ids=["text512_two_blocks_per_cu", "text512_gain8", "fewer_blocks_than_cus", "ragged_500_shared_kv", "449_keys", "2048_keys_gain12", "1100_keys"])
def test_cross_attention_persistent_workgroups_agree_with_fp64_and_with_one_block_launches(ops, B, Lq, Lk, H, Bk, gain):
    """Round 4: short KV with a scratch (text cross-attention: 512 keys) runs the bounded loop as ONE persistent workgroup per CU that walks a
    run of q blocks and pulls the next block's Q rows into LDS while it works (attention_w16n.hip PERSIST).  Against the fp64 softmax, and
    BIT FOR BIT against the same loop launched one block per workgroup (wan_attention_debug_no_persist): a block's arithmetic does not
    know which workgroup ran it.  Shapes: more blocks than CUs (a workgroup walks several, across (batch, head) pairs), fewer, a ragged
    last q block, a ragged last KV tile, shared K / V^T, 8 and 32 tiles, rows that carry a reference shift (gain)."""
    g = torch.Generator().manual_seed(Lq + Lk)
    q = torch.randn(B, Lq, H, 128, generator=g).to(BF); k = torch.randn(Bk, Lk, H, 128, generator=g)
    v = torch.randn(Bk, Lk, H, 128, generator=g).to(BF)
    k[:, :, : max(1, H - 1)] *= gain      # (the last head stays at gain 1: plain and shifted blocks in one walk)
    k = k.to(BF)
    qs = (q.float() * ops.attention_qscale()).to(BF)
    ref = _prescaled_ref(qs, k.expand(B, -1, -1, -1) if Bk == 1 else k, v.expand(B, -1, -1, -1) if Bk == 1 else v)
    scratch = torch.zeros(ops.attention_scratch_words(B, Bk, Lq, H), device="cuda")
    vt = ops.transpose_v(cu(v))
    old = _no_persist(2)      # (round 6: at 512 keys the product dispatch is attention_xkv.hip -- its own test below; 2 = this walk everywhere)
    try:
        got = ops.attention(cu(qs), cu(k), vt, q_prescaled=True, kmax_scratch=scratch, Bk=Bk)
    finally:
        _no_persist(old)
    nflag = ((Lq + 255) // 256) * H * B
    flags = scratch[Bk * H:Bk * H + nflag].view(torch.int32)
    assert int((flags != 0).sum()) == 0, flags.cpu().tolist()
    err = (got.float().cpu() - ref).abs()
    assert attn_ok(got, ref) and err.mean().item() <= 2e-3, (err.max().item(), err.mean().item())
    old = _no_persist(1)
    try:
        one = ops.attention(cu(qs), cu(k), vt, q_prescaled=True, kmax_scratch=scratch, Bk=Bk)
    finally:
        _no_persist(old)
    assert torch.equal(got, one)
    # not pre-scaled (the other instantiation): the same rows up to the rounding of q * scale
    got2 = ops.attention(cu(q), cu(k), vt, q_prescaled=False, kmax_scratch=scratch, Bk=Bk)
    assert attn_ok(got2, ref)


def test_cross_attention_persistent_handover_keeps_q_intact_in_place(ops):
    """wan_dit_forward attends in place (o = q).  A block whose rows leave the exponent range hands itself over to the tracking launch AFTER
    its loop -- it must not have stored anything, or that launch would read O where it expects Q.  Head 0 carries a key far above what the
    first tile suggests (row sums overflow: every block of the head is redone), head 1 is ordinary, q blocks 2.. of head 2 are beyond
    SHIFT_LIMIT (declined before the loop: the walk must still fetch the next block's rows).  In place == out of place == fp64."""
    g = torch.Generator().manual_seed(5)
    B, Lq, Lk, H = 2, 1500, 512, 3
    q = torch.randn(B, Lq, H, 128, generator=g); k = torch.randn(B, Lk, H, 128, generator=g)
    v = torch.randn(B, Lk, H, 128, generator=g).to(BF)
    q[:, :, 0, 0] += 6.0
    k[:, 300, 0] = 0.0
    k[:, 300, 0, 0] = 400.0
    q[:, 512:, 2] *= 400.0
    k = k.to(BF)
    qs = (q * ops.attention_qscale()).to(BF)
    ref = _prescaled_ref(qs, k, v)
    scratch = torch.zeros(ops.attention_scratch_words(B, B, Lq, H), device="cuda")
    vt = ops.transpose_v(cu(v))
    old = _no_persist(2)      # (the persistent walk: out of place the product dispatch is attention_xkv.hip at 512 keys)
    try:
        out = ops.attention(cu(qs), cu(k), vt, q_prescaled=True, kmax_scratch=scratch)
    finally:
        _no_persist(old)
    nqb = (Lq + 255) // 256
    flags = scratch[B * H:B * H + nqb * H * B].view(torch.int32).view(B * H, nqb).cpu()
    assert flags[0].tolist() == [1] * nqb and flags[3].tolist() == [1] * nqb          # head 0: handed over after the loop
    assert flags[1].tolist() == [0] * nqb
    assert flags[2].tolist() == [0, 0] + [1] * (nqb - 2)                                # head 2: declined before the loop from block 2 on
    assert attn_ok(out, ref)
    qio = cu(qs).clone()
    same = ops.attention(qio, cu(k), vt, q_prescaled=True, kmax_scratch=scratch, out=qio)
    assert same.data_ptr() == qio.data_ptr() and torch.equal(same, out)


@pytest.mark.parametrize("B,Lq,H,Bk", [(2, 3405, 12, 2), (1, 700, 3, 1), (3, 1100, 5, 1), (1, 16, 1, 1), (2, 4096 + 17, 40, 2), (1, 75600, 3, 1)],
                         ids=["two_runs_per_cu", "fewer_blocks_than_cus", "shared_kv_ragged_tile", "one_tile", "40_heads_ragged_row", "full_length_rows"])
def test_cross_attention_kv_stationary_agrees_with_fp64_and_with_the_persistent_walk(ops, B, Lq, H, Bk):
    """Round 6: text cross-attention (512 keys, q pre-scaled, out of place) keeps a head's K / V^T in the registers of one workgroup and streams
    the Q rows past them (attention_xkv.hip): a wave attends 128 keys, the four partial O^T and row sums are added through LDS (the bounded
    softmax's partial sums add exactly).  Against the fp64 softmax at the file's attention tolerance, and against the persistent walk it
    replaces (debug hook 2) within two bf16 ulp: the same P = 2^s, summed in another order.  Shapes: runs that cross (batch, head) pairs,
    fewer blocks than CUs, shared K / V^T with a ragged last tile (1100 = 68 x 16 + 12), a single tile, the model's 40 heads with a
    one-row last tile, the BASELINE row count."""
    g = torch.Generator().manual_seed(Lq + H)
    q = torch.randn(B, Lq, H, 128, generator=g).to(BF); k = torch.randn(Bk, 512, H, 128, generator=g).to(BF)
    v = torch.randn(Bk, 512, H, 128, generator=g).to(BF)
    qs = (q.float() * ops.attention_qscale()).to(BF)
    scratch = torch.zeros(ops.attention_scratch_words(B, Bk, Lq, H), device="cuda")
    vt = ops.transpose_v(cu(v))
    got = ops.attention(cu(qs), cu(k), vt, q_prescaled=True, kmax_scratch=scratch, Bk=Bk)
    nflag = ((Lq + 255) // 256) * H * B
    flags = scratch[Bk * H:Bk * H + nflag].view(torch.int32)
    assert int((flags != 0).sum()) == 0, flags.cpu().tolist()
    old = _no_persist(2)
    try:
        walk = ops.attention(cu(qs), cu(k), vt, q_prescaled=True, kmax_scratch=scratch, Bk=Bk)
    finally:
        _no_persist(old)
    d = (got.float() - walk.float()).abs()
    assert bool((d <= torch.clamp(walk.float().abs() * 2.0 ** -7, min=2e-3)).all()), d.max().item()
    assert not torch.equal(got, walk) or Lq <= 16       # (it IS another kernel: identical outputs would mean the hook did not switch)
    if Lq <= 5000:
        ref = _prescaled_ref(qs, k, v)
        err = (got.float().cpu() - ref).abs()
        assert attn_ok(got, ref) and err.mean().item() <= 2e-3, (err.max().item(), err.mean().item())


def test_cross_attention_kv_stationary_hands_unsound_rows_over(ops):
    """attention_xkv.hip has no running maximum and no reference shift: a row whose sum leaves [2^-80, 2^100] (a score beyond ~100 in log2
    units, or every score far below) flags its 256-row block and the tracking launch behind every bounded launch redoes the block.  Head 0 of
    batch 0 carries a key that overflows rows of every block; rows 300..399 of head 1 have all their scores near -200 (underflow: block 1 only);
    head 2 is ordinary.  Flags exactly there, result == fp64 everywhere; an in-place call (o = q) never takes this kernel and agrees."""
    g = torch.Generator().manual_seed(9)
    B, Lq, H = 2, 1500, 3
    q = torch.randn(B, Lq, H, 128, generator=g); k = torch.randn(B, 512, H, 128, generator=g)
    v = torch.randn(B, 512, H, 128, generator=g).to(BF)
    q[0, :, 0, 0] += 6.0
    k[0, 300, 0] = 0.0
    k[0, 300, 0, 0] = 400.0
    q[:, 300:400, 1] = 0.0
    q[:, 300:400, 1, 5] = -40.0
    k[:, :, 1, 5] = k[:, :, 1, 5].abs() + 40.0
    k = k.to(BF)
    qs = (q * ops.attention_qscale()).to(BF)
    ref = _prescaled_ref(qs, k, v)
    scratch = torch.zeros(ops.attention_scratch_words(B, B, Lq, H), device="cuda")
    vt = ops.transpose_v(cu(v))
    out = ops.attention(cu(qs), cu(k), vt, q_prescaled=True, kmax_scratch=scratch)
    nqb = (Lq + 255) // 256
    flags = scratch[B * H:B * H + nqb * H * B].view(torch.int32).view(B * H, nqb).cpu()
    assert flags[0].tolist() == [1] * nqb, flags.tolist()
    assert flags[1].tolist() == [0, 1] + [0] * (nqb - 2) and flags[4].tolist() == flags[1].tolist(), flags.tolist()
    assert flags[2].tolist() == [0] * nqb and flags[3].tolist() == [0] * nqb and flags[5].tolist() == [0] * nqb, flags.tolist()
    assert attn_ok(out, ref)
    qio = cu(qs).clone()
    same = ops.attention(qio, cu(k), vt, q_prescaled=True, kmax_scratch=scratch, out=qio)
    assert same.data_ptr() == qio.data_ptr() and attn_ok(same, ref)


@pytest.mark.parametrize("gains", [(1.0, 8.0), (6.0, 8.0), (8.0, 2.0)], ids=["plain_then_shifted", "shifted_then_larger_shift", "shifted_then_same"])
def test_attention_sp_partial_sums_carry_their_shift(ops, gains):
    """Sequence parallelism: phase 0 leaves partial sums shifted by m(local max|k|), phase 1 knows the maxima of ALL segments and
    rescales what it carries by 2^(m_local - m_global) (the local maxima travel behind the flags in the scratch).  Local / remote K
    gains such that the carried sums are unshifted -> shifted, shifted -> shifted further, and shifted -> unchanged.  (Partial launches
    take m = U - 96 only -- they cannot see each other's first tile -- so their window closes near gain 10 on diffuse random rows.)"""
    g_local, g_remote = gains
    g = torch.Generator().manual_seed(61)
    B, Lq, Lk, H, nseg, own = 2, 300, 2130, 2, 3, 1
    q = torch.randn(B, Lq, H, 128, generator=g).to(BF)
    k = torch.randn(nseg, B, Lk, H, 128, generator=g); v = torch.randn(nseg, B, Lk, H, 128, generator=g).to(BF)
    k *= g_remote
    k[own] *= g_local / g_remote
    k = k.to(BF)
    qs = (q.float() * ops.attention_qscale()).to(BF)
    ref = _prescaled_ref(qs, torch.cat(list(k), dim=1), torch.cat(list(v), dim=1))
    vt = torch.stack([ops.transpose_v(cu(v[s])) for s in range(nseg)]).contiguous()
    got, scratch = ops.attention_sp(cu(qs), cu(k[own]), vt[own].contiguous(), cu(k), vt, own)
    assert (scratch[B * H:-B * H].view(torch.int32) == 0).all()                  # nothing needed the tracking loop
    kn_all = (k.float() ** 2).sum(-1).amax(dim=(0, 2)).reshape(-1)
    kn_own = (k[own].float() ** 2).sum(-1).amax(dim=1).reshape(-1)
    assert torch.allclose(scratch[:B * H].cpu(), kn_all, rtol=1e-5) and torch.allclose(scratch[-B * H:].cpu(), kn_own, rtol=1e-5)
    assert attn_ok(got, ref), (got.float().cpu() - ref).abs().max().item()


def test_permute16_is_a_block_transpose(ops):
    """wan_permute16 (the Ulysses re-packs): [A][B][blk] -> [B][A][blk] for the three shapes the forward uses -- rows x world pieces of
    one head group (1,280 B at 14B / 8 ranks), streams x world V^T blocks (megabytes), and the way back -- bit for bit."""
    from wan2gp_amd.lib import WanHipError
    g = torch.Generator().manual_seed(4)
    for A, B, n in ((2 * 9450, 8, 640), (2, 8, 640 * 128), (8, 301, 128), (1, 5, 8), (3, 1, 24)):
        src = torch.randn(A, B, n, generator=g).to(BF).cuda()
        got = ops.permute16(src, A, B)
        assert torch.equal(got.view(B, A, n), src.transpose(0, 1).contiguous())
    with pytest.raises(WanHipError):
        ops.permute16(torch.zeros(3, 5, 4, dtype=BF, device="cuda"), 3, 5)          # 8-byte blocks


def test_permute16_ex_pitched_column_ranges(ops):
    """wan_permute16_ex (round 5, the chunked Ulysses re-packs): a column range of [rows][world][W] rows into a [world][rows][Wc] block of
    its own and back, a V^T row range [S][world][W][Lp] -> [world][chunk][S][Wc][Lp] -- against torch.as_strided, bit for bit."""
    g = torch.Generator().manual_seed(5)
    rows, world, W = 301, 4, 5 * 128
    d = world * W
    src = torch.randn(rows * d, generator=g).to(BF).cuda()
    for o0, Wc in ((0, 2 * 128), (2 * 128, 3 * 128)):
        dst = torch.zeros(world * rows * Wc, dtype=BF, device="cuda")
        ops.permute16_ex(src[o0:], dst, rows, world, Wc * 2, d * 2, W * 2, Wc * 2, rows * Wc * 2)
        want = torch.as_strided(src, (world, rows, Wc), (W, d, 1), o0).contiguous()
        assert torch.equal(dst.view(world, rows, Wc), want)
        back = torch.zeros(rows * d, dtype=BF, device="cuda")
        ops.permute16_ex(dst, back[o0:], world, rows, Wc * 2, rows * Wc * 2, Wc * 2, W * 2, d * 2)
        assert torch.equal(torch.as_strided(back, (world, rows, Wc), (W, d, 1), o0), want)
        assert int((back != 0).sum()) <= world * rows * Wc                      # nothing outside the column range was written
    S, Lp = 2, 128
    vt = torch.randn(S * d * Lp, generator=g).to(BF).cuda()
    vs = torch.zeros_like(vt)
    for o0, Wc in ((0, 2 * 128), (2 * 128, 3 * 128)):
        ops.permute16_ex(vt[o0 * Lp:], vs[o0 * Lp * S:], S, world, Wc * Lp * 2, d * Lp * 2, W * Lp * 2, Wc * Lp * 2, S * W * Lp * 2)
    for o0, Wc in ((0, 2 * 128), (2 * 128, 3 * 128)):
        got = torch.as_strided(vs, (world, S, Wc * Lp), (S * W * Lp, Wc * Lp, 1), o0 * Lp * S)
        want = torch.as_strided(vt, (world, S, Wc * Lp), (W * Lp, d * Lp, 1), o0 * Lp)
        assert torch.equal(got, want)
    from wan2gp_amd.lib import WanHipError
    with pytest.raises(WanHipError):
        ops.permute16_ex(src, vs, 2, 2, 24, 48, 96, 24, 48)                       # 24-byte pieces


@pytest.mark.parametrize("Lq,Lk,nseg", [(300, 2130, 2), (96, 48, 4)], ids=["long_kv_bounded", "short_kv_tracking"])
def test_attention_query_batches_share_kv_batches_modulo(ops, Lq, Lk, nseg):
    """The Ulysses launch shape: B = nseg x S query batches (source rank, stream) against Bk = S K / V^T batches held in `nseg`
    segments -- q batch b attends batch b mod Bk.  Must equal the per-(rank, stream) attention over the concatenated segments."""
    g = torch.Generator().manual_seed(Lq)
    S, H = 2, 2
    B = nseg * S
    q = torch.randn(B, Lq, H, 128, generator=g).to(BF)
    k = torch.randn(nseg, S, Lk, H, 128, generator=g).to(BF); v = torch.randn(nseg, S, Lk, H, 128, generator=g).to(BF)
    qs = (q.float() * ops.attention_qscale()).to(BF)
    vt = torch.stack([ops.transpose_v(cu(v[i])) for i in range(nseg)]).contiguous()                 # [nseg][S][H*128][ldv]
    ldv = vt.shape[-1]
    scratch = torch.zeros(ops.attention_scratch_words(B, S, Lq, H), device="cuda")
    got = ops.attention(cu(qs), cu(k), vt, Lk=Lk, nseg=nseg, k_seg_stride=S * Lk * H * 128, vt_seg_stride=S * H * 128 * ldv, Bk=S,
                        q_prescaled=True, kmax_scratch=scratch)
    kc, vc = torch.cat(list(k), dim=1), torch.cat(list(v), dim=1)                                   # [S, nseg*Lk, H, 128]
    for b in range(B):
        ref = _prescaled_ref(qs[b:b + 1], kc[b % S:b % S + 1], vc[b % S:b % S + 1])
        assert attn_ok(got[b:b + 1], ref), (b, (got[b:b + 1].float().cpu() - ref).abs().max().item())


@pytest.mark.parametrize("own", [0, 2, 3])
def test_attention_sp_local_first_equals_contiguous(ops, own):
    """wan_attention_sp_local (own segment, partial sums) + wan_attention_sp_remote (the other segments on top, own skipped) ==
    attention over the concatenated segments, for the first / a middle / the last rank; ragged segment length (2,130 = 33 tiles +
    18 rows) and ragged q (300 rows); the own segment inside k_all is poisoned to prove it is never read."""
    g = torch.Generator().manual_seed(40 + own)
    B, Lq, Lk, H, nseg = 2, 300, 2130, 2, 4
    q = torch.randn(B, Lq, H, 128, generator=g).to(BF)
    k = torch.randn(nseg, B, Lk, H, 128, generator=g).to(BF); v = torch.randn(nseg, B, Lk, H, 128, generator=g).to(BF)
    qs = (q.float() * ops.attention_qscale()).to(BF)
    ref = _prescaled_ref(qs, torch.cat(list(k), dim=1), torch.cat(list(v), dim=1))
    vt = torch.stack([ops.transpose_v(cu(v[s])) for s in range(nseg)]).contiguous()
    k_all = cu(k).clone(); vt_all = vt.clone()
    k_all[own] = float("nan"); vt_all[own] = float("nan")                      # a rank's own slot of the gather buffer is not needed
    got, scratch = ops.attention_sp(cu(qs), cu(k[own]), vt[own].contiguous(), k_all, vt_all, own)
    assert attn_ok(got, ref), (got.float().cpu() - ref).abs().max().item()
    assert (scratch[B * H:-B * H].view(torch.int32) == 0).all()                        # every workgroup stayed on the bounded path


def test_attention_sp_bound_exceeded_is_recomputed_by_the_tracking_loop(ops):
    """A q-block whose rows break the bound only against a REMOTE segment (a huge key there): phase 0 accepts it (local bound),
    phase 1 flags it, and the tracking loop recomputes it over all four segments."""
    g = torch.Generator().manual_seed(50)
    B, Lq, Lk, H, nseg, own = 1, 600, 2112, 1, 4, 1
    q = torch.randn(B, Lq, H, 128, generator=g).to(BF)
    k = torch.randn(nseg, B, Lk, H, 128, generator=g).to(BF); v = torch.randn(nseg, B, Lk, H, 128, generator=g).to(BF)
    k[3, 0, 77, 0] *= 90.0                                                       # |k| ~ 1000: bound ~ 1400 log2 units for every q row
    qs = (q.float() * ops.attention_qscale()).to(BF)
    ref = _prescaled_ref(qs, torch.cat(list(k), dim=1), torch.cat(list(v), dim=1))
    vt = torch.stack([ops.transpose_v(cu(v[s])) for s in range(nseg)]).contiguous()
    got, scratch = ops.attention_sp(cu(qs), cu(k[own]), vt[own].contiguous(), cu(k), vt, own)
    assert (scratch[B * H:-B * H].view(torch.int32) == 1).all()
    assert attn_ok(got, ref), (got.float().cpu() - ref).abs().max().item()


# ---- the narrow in-place RMSNorm + RoPE kernel beside processes that come and go on the same GPU (round 6, runs 65-77) ---------------------
def _rr_victim(seconds, q):
    try:
        import time
        from wan2gp_amd import ops as OPS
        g = torch.Generator().manual_seed(6)
        d, grid = 1536, (9, 30, 52)
        Lt = grid[0] * grid[1] * grid[2]
        q0 = torch.randn(2, Lt, d, generator=g).to(BF).cuda()
        wq = (1 + 0.1 * torch.randn(d, generator=g)).to(BF).cuda()
        cos, sin = [t.cuda() for t in O.rope_tables(grid)]

        def f():
            x = q0.clone()
            OPS.rmsnorm_rope_(x, None, wq, wq, freqs=(cos, sin), L=Lt, q_scale=OPS.attention_qscale())
            return x
        ref = f().clone()
        q.put(("ready", 0, 0))
        bad = n = 0
        t_end = time.time() + seconds
        while time.time() < t_end:
            bad += int(not torch.equal(f(), ref))
            n += 1
        q.put(("done", bad, n))
    except Exception:
        import traceback
        q.put(("error", traceback.format_exc(), 0))


def _rr_neighbour(q):
    try:                                                                          # what the reproducer's neighbour did (tools/probes/dit_determinism.py):
        from wan2gp_amd.model import WanModelHIP                                  # a 2-layer model at the 1.3B widths built, loaded, run, and gone again
        cfg = O.make_config("t2v_1.3B")
        cfg.num_layers = 2
        m = WanModelHIP(model_type=cfg.model_type, dim=cfg.dim, ffn_dim=cfg.ffn_dim, num_heads=cfg.num_heads, num_layers=cfg.num_layers,
                        in_dim=cfg.in_dim, out_dim=cfg.out_dim).load_state_dict(O.synth_weights(cfg, seed=7))
        lat, ctx, ctx_null, _ = O.synth_inputs(cfg, 9, 60, 104, seed=3)
        x, c0, c1 = lat.cuda(), ctx.cuda(), ctx_null.cuda()
        for _ in range(12):
            m([x, x], t=torch.tensor([500]), context=[c0, c1])
        torch.cuda.synchronize()
        q.put("gone")
    except Exception:
        import traceback
        q.put(traceback.format_exc())


def test_rmsnorm_rope_narrow_form_is_reproducible_while_processes_come_and_go():
    """The narrow in-place form of rmsnorm_rope_kernel (one row per wave: d < 4096) was the one kernel of the library that did not return its bits
    beside another process: in ~10-ms windows of a neighbour's start-up or exit every tenth launch left ~1 % of its rows with a few wrong
    16-byte chunks -- the 1.3B forward differed in 10-25 % of its calls when two processes shared the GPU.  It had no LDS
    and no barrier; with one LDS word per wave and one barrier it returned its bits in 333,066 of 333,066 launches (run 77).  Runs 80-84 found what was
    wrong -- the low result of `v_pk_mul_f32 ... op_sel:[0,1] op_sel_hi:[1,0]` (RoPE's crosswise products) zero in lanes 48-63 -- and the rotation
    now works on aligned register pairs (tests/test_isa_invariants.py holds the instruction out of the row kernels).  Here: one process
    launches it in place on fresh copies of fixed rows while two others, one after the other, build a 2-layer model at the 1.3B widths, run it and exit."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q, qn = ctx.Queue(), ctx.Queue()
    victim = ctx.Process(target=_rr_victim, args=(75.0, q))
    victim.start()
    msg = q.get(timeout=300)
    assert msg[0] == "ready", msg
    for _ in range(2):
        nb = ctx.Process(target=_rr_neighbour, args=(qn,))
        nb.start()
        r = qn.get(timeout=300)
        assert r == "gone", r
        nb.join(timeout=60)
    msg = q.get(timeout=300)
    victim.join(timeout=60)
    assert msg[0] == "done", msg
    print(f"rmsnorm_rope (narrow, in place): {msg[1]} of {msg[2]} launches differ")
    assert msg[1] == 0 and msg[2] > 1000, msg
