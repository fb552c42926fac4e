"""CPU emulation of the index algebra of the two MFMA kernels (no GPU needed).

There is no GPU in the build container, so the staging permutations, XOR swizzles,
fragment-read offsets, MFMA lane layouts and epilogue mappings of
wan2gp_amd/csrc/gemm_bf16.hip and attention.hip are transliterated here lane by lane
and executed with numpy against a plain matmul / softmax reference.  The MFMA lane
layouts are the documented gfx950 ones (cdna guide §3):
  16x16x32: A lane l -> row l&15, k (l>>4)*8+e ; B lane l -> col l&15, k (l>>4)*8+e ;
            D lane l reg r -> row (l>>4)*4+r, col l&15
  32x32x16: A lane l -> row l&31, k (l>>5)*8+e ; B lane l -> col l&31 ; D lane l reg r ->
            row (r&3)+8*(r>>2)+4*(l>>5), col l&31
Also checks the bank-conflict-freedom claims (MI355X_MICROARCH.md §LDS lane groups).
"""
import numpy as np
import pytest

B128_GROUPS = [  # ds_read_b128 lane groups (one LDS cycle each when conflict free)
    list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
    list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
    [32 + x for x in (list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)))],
    [32 + x for x in (list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)))],
]


def b128_conflict_free(byte_offsets):
    """byte_offsets[lane] of a wave64 ds_read_b128; bank slot = (addr/16) mod 16 of a 256-B row."""
    for g in B128_GROUPS:
        slots = [(byte_offsets[l] // 16) % 16 for l in g]
        if len(set(slots)) != len(slots):
            return False
    return True


# ----------------------------------------------------------------------------------------------
# GEMM (gemm_bf16.hip)
# ----------------------------------------------------------------------------------------------
def emulate_gemm_block(Y, X, y0, x0):
    """One 128x128 output tile, returns dict {(yrow, xcol): value} written by the epilogue."""
    YM, K = Y.shape
    XN = X.shape[0]
    BK = 64
    out = {}
    acc = np.zeros((256, 4, 4, 4), dtype=np.float64)  # [tid][yt][xt][r]
    conflict_free = True
    for k0 in range(0, K, BK):
        ylds = np.zeros((128 * 64,), dtype=np.float64)  # element-addressed image (2 B per element)
        xlds = np.zeros((128 * 64,), dtype=np.float64)
        for tid in range(256):
            for i in range(4):
                q = i * 256 + tid
                row, pch = q >> 3, q & 7
                lch = pch ^ ((row >> 1) & 7)
                yr = min(y0 + row, YM - 1)
                slab, jj = row >> 6, row & 63
                nt, ii = jj >> 4, jj & 15
                xr = min(x0 + slab * 64 + (ii >> 2) * 16 + nt * 4 + (ii & 3), XN - 1)
                # LDS-DMA: lane-linear destination q*16 bytes = q*8 elements
                ylds[q * 8:q * 8 + 8] = Y[yr, k0 + lch * 8:k0 + lch * 8 + 8]
                xlds[q * 8:q * 8 + 8] = X[xr, k0 + lch * 8:k0 + lch * 8 + 8]
        for wave in range(4):
            wy, wx = wave >> 1, wave & 1
            for ks in range(2):
                yf = np.zeros((64, 4, 8)); xf = np.zeros((64, 4, 8))
                offs_y = np.zeros((4, 64), dtype=int); offs_x = np.zeros((4, 64), dtype=int)
                for lane in range(64):
                    frow, fch = lane & 15, lane >> 4
                    for t in range(4):
                        ry = wy * 64 + t * 16 + frow
                        yo = (ry * 128 + ((fch ^ ((ry >> 1) & 7)) << 4)) ^ (ks << 6)
                        rx = wx * 64 + t * 16 + frow
                        xo = (rx * 128 + ((fch ^ ((rx >> 1) & 7)) << 4)) ^ (ks << 6)
                        yf[lane, t] = ylds[yo // 2:yo // 2 + 8]
                        xf[lane, t] = xlds[xo // 2:xo // 2 + 8]
                        offs_y[t, lane] = yo; offs_x[t, lane] = xo
                for t in range(4):
                    conflict_free &= b128_conflict_free(offs_y[t]) and b128_conflict_free(offs_x[t])
                # mfma_16x16x32(a = xf[b], b = yf[a]) : D[i][j] = sum_k A[i][k] B[k][j]
                for a in range(4):
                    for b in range(4):
                        Am = np.zeros((16, 32)); Bm = np.zeros((32, 16))
                        for lane in range(64):
                            Am[lane & 15, (lane >> 4) * 8:(lane >> 4) * 8 + 8] = xf[lane, b]
                            Bm[(lane >> 4) * 8:(lane >> 4) * 8 + 8, lane & 15] = yf[lane, a]
                        D = Am @ Bm
                        for lane in range(64):
                            for r in range(4):
                                acc[wave * 64 + lane, a, b, r] += D[(lane >> 4) * 4 + r, lane & 15]
    for tid in range(256):
        lane, wave = tid & 63, tid >> 6
        wy, wx = wave >> 1, wave & 1
        xb = x0 + wx * 64 + (lane >> 4) * 16
        for yt in range(4):
            yr = y0 + wy * 64 + yt * 16 + (lane & 15)
            if yr >= YM:
                continue
            for xt in range(4):
                for r in range(4):
                    xc = xb + xt * 4 + r
                    if xc < XN:
                        assert (yr, xc) not in out
                        out[(yr, xc)] = acc[tid, yt, xt, r]
    return out, conflict_free


@pytest.mark.parametrize("YM,XN,K,y0,x0", [(128, 128, 128, 0, 0), (200, 144, 64, 128, 128), (70, 256, 64, 0, 128)])
def test_gemm_index_algebra(YM, XN, K, y0, x0):
    rng = np.random.default_rng(0)
    Y = rng.integers(-4, 5, size=(YM, K)).astype(np.float64)
    X = rng.integers(-4, 5, size=(XN, K)).astype(np.float64)
    out, cf = emulate_gemm_block(Y, X, y0, x0)
    ref = Y @ X.T
    ny, nx = min(128, YM - y0), min(128, XN - x0)
    assert len(out) == ny * nx                      # every in-range output written exactly once
    for (yr, xc), v in out.items():
        assert v == ref[yr, xc], (yr, xc)
    assert cf, "fragment reads are not bank-conflict free"


# ----------------------------------------------------------------------------------------------
# attention (attention.hip)
# ----------------------------------------------------------------------------------------------
def emulate_attention_block(Q, K, V, qb, nseg=1):
    """One block (128 q rows) of one head. Q [Lq,128], K [nseg*Lk,128] given as nseg segments of Lk rows,
    V likewise. Returns O rows {q: vec128}."""
    Lq = Q.shape[0]
    Lk = K.shape[0] // nseg
    ldv = ((Lk + 63) // 64) * 64
    # V^T per segment, zero padded
    Vt = np.zeros((nseg, 128, ldv))
    for s in range(nseg):
        Vt[s, :, :Lk] = V[s * Lk:(s + 1) * Lk].T
    scale = 1.0 / np.sqrt(128.0)
    tps = (Lk + 63) // 64
    ntile = tps * nseg
    res = {}
    cf = True
    for wave in range(4):
        q0 = qb * 128 + wave * 32
        qf = np.zeros((64, 8, 8))
        for lane in range(64):
            half, l31 = lane >> 5, lane & 31
            qrow = min(q0 + l31, Lq - 1)
            for ks in range(8):
                qf[lane, ks] = Q[qrow, ks * 16 + half * 8:ks * 16 + half * 8 + 8]
        accO = np.zeros((64, 4, 16))
        m_run = np.full(64, -np.inf)
        l_run = np.zeros(64)
        for t in range(ntile):
            seg = t // tps
            kv0 = (t - seg * tps) * 64
            klds = np.zeros(64 * 128); vlds = np.zeros(128 * 64)
            for tid in range(256):
                for i in range(4):
                    s = i * 256 + tid
                    r, pch = s >> 4, s & 15
                    lch = pch ^ (r & 15)
                    kvl = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1)
                    kr = min(kv0 + kvl, Lk - 1)
                    klds[s * 8:s * 8 + 8] = K[seg * Lk + kr, lch * 8:lch * 8 + 8]
                    r2, pch2 = s >> 3, s & 7
                    lch2 = pch2 ^ ((r2 >> 1) & 7)
                    vlds[s * 8:s * 8 + 8] = Vt[seg, r2, kv0 + lch2 * 8:kv0 + lch2 * 8 + 8]
            accS = np.zeros((64, 2, 16))
            for T in range(2):
                for ks in range(8):
                    Am = np.zeros((32, 16)); Bm = np.zeros((16, 32))
                    offs = np.zeros(64, dtype=int)
                    for lane in range(64):
                        half, l31 = lane >> 5, lane & 31
                        r = T * 32 + l31
                        off = (r * 256 + ((half ^ (r & 15)) << 4)) ^ (ks << 5)
                        offs[lane] = off
                        Am[l31, half * 8:half * 8 + 8] = klds[off // 2:off // 2 + 8]
                        Bm[half * 8:half * 8 + 8, l31] = qf[lane, ks]
                    cf &= b128_conflict_free(offs)
                    D = Am @ Bm
                    for lane in range(64):
                        for r in range(16):
                            accS[lane, T, r] += D[(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), lane & 31]
            tt = t % tps
            if (tt + 1) * 64 > Lk:
                for lane in range(64):
                    half = lane >> 5
                    for T in range(2):
                        for r in range(16):
                            kv = tt * 64 + T * 32 + (r & 7) + 8 * half + 16 * (r >> 3)
                            if kv >= Lk:
                                accS[lane, T, r] = -np.inf
            mt = accS.reshape(64, 32).max(axis=1)
            mt = np.maximum(mt, mt[np.arange(64) ^ 32])
            m_new = np.maximum(m_run, mt)
            alpha = np.exp((m_run - m_new) * scale)
            p = np.exp(accS * scale - (m_new * scale)[:, None, None])
            l_run = l_run * alpha + p.reshape(64, 32).sum(axis=1)
            accO *= alpha[:, None, None]
            m_run = m_new
            for T in range(2):
                for s in range(2):
                    cx = (T * 4 + s * 2) << 4
                    for dt in range(4):
                        Am = np.zeros((32, 16)); Bm = np.zeros((16, 32))
                        offs = np.zeros(64, dtype=int)
                        for lane in range(64):
                            half, l31 = lane >> 5, lane & 31
                            r = dt * 32 + l31
                            off = (r * 128 + ((half ^ ((r >> 1) & 7)) << 4)) ^ cx
                            offs[lane] = off
                            Am[l31, half * 8:half * 8 + 8] = vlds[off // 2:off // 2 + 8]
                            Bm[half * 8:half * 8 + 8, l31] = p[lane, T, s * 8:s * 8 + 8]
                        cf &= b128_conflict_free(offs)
                        D = Am @ Bm
                        for lane in range(64):
                            for r in range(16):
                                accO[lane, dt, r] += D[(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), lane & 31]
        l_tot = l_run + l_run[np.arange(64) ^ 32]
        # epilogue staging O[q][d]
        ob = np.zeros((32, 128))
        for lane in range(64):
            half, l31 = lane >> 5, lane & 31
            for dt in range(4):
                for g in range(4):
                    d = dt * 32 + g * 8 + half * 4
                    ob[l31, d:d + 4] = accO[lane, dt, g * 4:g * 4 + 4] / l_tot[lane]
        for r in range(32):
            if q0 + r < Lq:
                res[q0 + r] = ob[r]
    return res, cf


@pytest.mark.parametrize("Lq,Lk,nseg", [(128, 64, 1), (100, 200, 1), (64, 96, 2)])
def test_attention_index_algebra(Lq, Lk, nseg):
    rng = np.random.default_rng(1)
    Q = rng.standard_normal((Lq, 128))
    K = rng.standard_normal((Lk * nseg, 128))
    V = rng.standard_normal((Lk * nseg, 128))
    res, cf = emulate_attention_block(Q, K, V, 0, nseg)
    S = (Q @ K.T) / np.sqrt(128.0)
    P = np.exp(S - S.max(axis=1, keepdims=True))
    ref = (P / P.sum(axis=1, keepdims=True)) @ V
    assert sorted(res) == list(range(min(Lq, 128)))
    for q, v in res.items():
        np.testing.assert_allclose(v, ref[q], rtol=1e-9, atol=1e-9)
    assert cf, "attention fragment reads are not bank-conflict free"


# ----------------------------------------------------------------------------------------------
# gemm32.hip (64-row waves) and the 128-row wave layout gemm256k.hip inherited from the removed BK = 32 generation: 32x32x16
# MFMA, 64-byte LDS rows
# ----------------------------------------------------------------------------------------------
def _pi(rho):
    """LDS row rho of a 32-row x tile holds x row pi(rho) (the x-row staging permutation of gemm32.hip / gemm256k.hip / gemm_fp8.hip)."""
    return 16 * ((rho >> 2) & 1) + (rho & 3) + 4 * (rho >> 3)


def emulate_gemm256_wave(Y, X, wy, wx, wave_rows_x, k0):
    """One k-step pair (BK = 32) of one wave with 128-row (wave_rows_x = 128) or gemm32's 64-row x slabs: stages the two LDS images
    exactly as the DMA plan does, reads the fragments with the kernel's addresses and returns
    (acc[yt][xt][lane][r], conflict_free).  Y/X are float arrays [rows, K]."""
    nxt = wave_rows_x // 32
    xrows = 2 * wave_rows_x                      # x rows of the workgroup tile
    ylds = np.zeros(256 * 32)                    # element-addressed images, 32 elements (64 B) per row
    xlds = np.zeros(xrows * 32)
    for q in range(256 * 4):                     # Y image: 1024 16-B slots
        row, pch = q >> 2, q & 3
        lch = pch ^ ((row >> 2) & 3)
        ylds[q * 8:q * 8 + 8] = Y[row, k0 + lch * 8:k0 + lch * 8 + 8]
    for q in range(xrows * 4):
        row, pch = q >> 2, q & 3
        lch = pch ^ ((row >> 2) & 3)
        slab, rem = divmod(row, wave_rows_x)
        xt, rho = rem >> 5, rem & 31
        xr = slab * wave_rows_x + xt * 32 + _pi(rho)
        xlds[q * 8:q * 8 + 8] = X[xr, k0 + lch * 8:k0 + lch * 8 + 8]
    acc = np.zeros((4, nxt, 64, 16))
    ok = True
    for ks in range(2):
        yfr = np.zeros((4, 64, 8)); xfr = np.zeros((nxt, 64, 8))
        yoffs = np.zeros((4, 64), dtype=int); xoffs = np.zeros((nxt, 64), dtype=int)
        for lane in range(64):
            l31, half = lane & 31, lane >> 5
            sw = (l31 >> 2) & 3
            ya = (wy * 128 + l31) * 64 + ((half ^ sw) << 4)
            xa = (wx * wave_rows_x + l31) * 64 + ((half ^ sw) << 4)
            for t in range(4):
                off = t * 2048 + (ya ^ (ks << 5))
                yoffs[t, lane] = off
                yfr[t, lane] = ylds[off // 2:off // 2 + 8]
            for t in range(nxt):
                off = t * 2048 + (xa ^ (ks << 5))
                xoffs[t, lane] = off
                xfr[t, lane] = xlds[off // 2:off // 2 + 8]
        for t in range(4):
            ok &= b128_conflict_free(list(yoffs[t]))
        for t in range(nxt):
            ok &= b128_conflict_free(list(xoffs[t]))
        # MFMA 32x32x16: D[i][j] += sum_k A[i][k] B[k][j]; A = X fragment (i = lane&31), B = Y fragment (j = lane&31),
        # k = 8*(lane>>5) + e;  D lane (j, h) reg r <-> row i = (r&3) + 8*(r>>2) + 4*h
        for a in range(4):
            for b in range(nxt):
                A = np.zeros((32, 16)); Bm = np.zeros((16, 32))
                for lane in range(64):
                    A[lane & 31, 8 * (lane >> 5):8 * (lane >> 5) + 8] = xfr[b, lane]
                    Bm[8 * (lane >> 5):8 * (lane >> 5) + 8, lane & 31] = yfr[a, lane]
                D = A @ Bm
                for lane in range(64):
                    for r in range(16):
                        acc[a, b, lane, r] += D[(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), lane & 31]
    return acc, ok


@pytest.mark.parametrize("wave_rows_x", [128, 64])
def test_gemm32_gemm256_index_algebra(wave_rows_x):
    """The lane that the epilogue treats as (y row, 16 consecutive x starting at 16*half) really holds those products,
    and every fragment read is LDS-bank-conflict free."""
    rng = np.random.default_rng(3)
    K = 32
    Y = rng.standard_normal((256, K)); X = rng.standard_normal((2 * wave_rows_x, K))
    ref = Y @ X.T
    for wy in range(2):
        for wx in range(2):
            acc, ok = emulate_gemm256_wave(Y, X, wy, wx, wave_rows_x, 0)
            assert ok, "bank conflict in a fragment read"
            for yt in range(4):
                for xt in range(wave_rows_x // 32):
                    for lane in range(64):
                        yr = wy * 128 + yt * 32 + (lane & 31)
                        xb = wx * wave_rows_x + xt * 32 + 16 * (lane >> 5)
                        np.testing.assert_allclose(acc[yt, xt, lane], ref[yr, xb:xb + 16], rtol=1e-12, atol=1e-12)


def test_x_row_permutation_is_a_bijection():
    assert sorted(_pi(r) for r in range(32)) == list(range(32))


# ----------------------------------------------------------------------------------------------
# attention_w64q.hip: exact 3-term bf16 split of -m_ref, and the flat schedule's static invariants
# ----------------------------------------------------------------------------------------------
def _bf16_rne(x):
    """float32 -> bf16 (round to nearest even) -> float32, numpy."""
    b = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((b + 0x7FFF + ((b >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def test_mref_three_term_bf16_split_is_exact():
    """set_mref(): hi = bf16(nm), mid = bf16(nm - hi), lo = bf16(nm - hi - mid); hi + mid + lo == nm exactly in fp32
    (what the S-initialising MFMA [1 1 1 0..] x [hi; mid; lo; 0..] adds up)."""
    rng = np.random.default_rng(11)
    nm = np.concatenate([rng.standard_normal(20000).astype(np.float32) * np.float32(37.0),
                         np.float32([0.0, -0.0, 1.0, -1e-3, 123.456, -88.125, 3e-5, -2.5e4])])
    hi = _bf16_rne(nm)
    r1 = (nm - hi).astype(np.float32)
    mid = _bf16_rne(r1)
    r2 = (r1 - mid).astype(np.float32)
    lo = _bf16_rne(r2)
    assert np.array_equal(lo, r2), "third term must capture the residual exactly"
    total = ((hi.astype(np.float64) + mid.astype(np.float64)) + lo.astype(np.float64))
    assert np.array_equal(total.astype(np.float32), nm)
    assert np.array_equal(total, nm.astype(np.float64))


def _flat_constants():
    import os
    import re
    src = open(os.path.join(os.path.dirname(__file__), "..", "wan2gp_amd", "csrc", "attention_w64q.hip")).read()
    a0 = int(re.search(r"constexpr int FLAT_A0 = (\d+);", src).group(1))
    b0 = int(re.search(r"constexpr int FLAT_B0 = (\d+);", src).group(1))
    return a0, b0


def test_flat_schedule_invariants():
    """Static checks of attention_w64q.hip's 68-gap tile (A 0..17 incl. 2 init MFMAs, B 18..33, C 34..51, D 52..67):
    chunk c of q-block a runs at gap A0+c of its tile, of q-block b at gap B0+c (wrapping into the next tile);
    chunks 0..3 row max, 4 test, 5+k exp2 of score k (k = 0..31), pair j packed in chunk 5+2j+2 (pair 15: chunk 37)."""
    A0, B0 = _flat_constants()
    exp_a = {A0 + 5 + k for k in range(32)}
    exp_b = {(B0 + 5 + k) % 68 for k in range(32)}
    assert max(exp_a) <= 67, "q-block a's chunks must stay inside its own tile"
    assert not (exp_a & exp_b), "at most one v_exp_f32 per MFMA gap"
    # first readers of S come >= 3 MFMAs after the slot that produced it (asm MFMAs are not hazard-padded)
    assert A0 >= 17 + 3 and B0 >= 51 + 3
    pack_a = [A0 + 5 + 2 * j + 2 for j in range(16)]
    pack_b = [B0 + 5 + 2 * j + 2 for j in range(16)]           # absolute gaps; PV_b runs at 68 + 18 + i
    for c4 in range(4):
        need = max(pack_a[4 * c4:4 * c4 + 4])
        assert need + 2 <= 52 + 4 * c4, f"P_a k-step {c4} packed at gap {need}, PV_a reads it at {52 + 4 * c4}"
        need = max(pack_b[4 * c4:4 * c4 + 4])
        assert need + 2 <= 68 + 18 + 4 * c4, f"P_b k-step {c4} packed at gap {need}, PV_b reads it at {86 + 4 * c4}"
    # S tiles are overwritten by the next QK^T slot only after their last exp2
    assert max(exp_a) < 68 + 0 and (B0 + 5 + 31) < 68 + 34
    # q-block b's chunks that run in the next tile are exactly 9..37 (the kernel's `G <= 28 -> chunk G + 9`)
    assert 68 - B0 == 9
    # V^T(t) fragment f is read at gap 20+2f: after PV_b(t-1) used the old one (gap 18+f), before PV_a(t) needs it (52+f)
    for f in range(16):
        g = 20 + 2 * f
        assert 18 + f < g and g + 2 <= 52 + f
    # K fragments in need order: read r at gap 52+2r (r < 8) of the previous tile or gap r-8 of this tile, used by MFMA 2+r
    for r in range(16):
        g = (52 + 2 * r - 68) if r < 8 else (r - 8)
        assert g + 2 <= 2 + r


# ---------------------------------------------------------------------------------------------------------------------
# gemm256k.hip (BK = 64, five 32-KB units): DMA plan <-> fragment addresses, bank conflicts, ring schedule
# ---------------------------------------------------------------------------------------------------------------------
def _g256k_unit_image():
    """slot q (16 B) of a unit image -> (row, logical chunk), as the DMA plan of gemm256k.hip fills it."""
    img = {}
    for i in range(8):
        for tid in range(256):
            q = i * 256 + tid
            row, pch = q >> 3, q & 7
            assert q * 16 == (i * 256 + (tid >> 6) * 64 + (tid & 63)) * 16          # piece i of wave w, lane order
            img[q * 16] = (row, pch ^ ((row >> 1) & 7))
    return img


def test_gemm256k_fragment_addresses_hit_the_right_chunks():
    img = _g256k_unit_image()
    assert sorted(img.values()) == [(r, c) for r in range(256) for c in range(8)]       # every (row, chunk) exactly once
    # every instruction's 8 lanes of a row cover the row's whole 128-B line
    for i in range(8):
        for wave in range(4):
            rows = {}
            for lane in range(64):
                row, lch = img[(i * 256 + wave * 64 + lane) * 16]
                rows.setdefault(row, set()).add(lch)
            assert len(rows) == 8 and all(v == set(range(8)) for v in rows.values())
    for w in range(2):                     # wy (or wx)
        for l31 in range(32):
            for half in range(2):
                sw = (l31 >> 1) & 7
                addr0 = (w * 128 + l31) * 128 + ((half ^ sw) << 4)
                for r in range(4):
                    for ks in range(4):
                        addr = r * 4096 + (addr0 ^ (ks << 5))
                        assert img[addr] == (w * 128 + r * 32 + l31, 2 * ks + half)


def test_gemm256k_ds_read_b128_is_conflict_free():
    """The 16 lanes a ds_read_b128 services in one LDS cycle (MI355X_MICROARCH.md LDS table) must cover 64 distinct banks."""
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[l + 32 for l in g] for g in groups]
    for ks in range(4):
        for g in groups:
            banks = set()
            for lane in g:
                l31, half = lane & 31, lane >> 5
                addr = l31 * 128 + (((half ^ ((l31 >> 1) & 7)) << 4) ^ (ks << 5))
                for d in range(4):
                    banks.add((addr // 4 + d) % 64)
            assert len(banks) == 64


def test_gemm256k_ring_schedule():
    """Event simulation of the five-unit ring: every unit is written only after its previous tenant's last read and
    is complete (by the vmcnt rule) before its first read; 8 pieces per wave may still fly at every sync point."""
    nk = 23
    unit_of = lambda kind, S: 2 * S + (1 if kind == "X" else 0)
    slot_of = lambda u: u % 5
    issue_order = []                      # (unit) in per-wave issue order, 8 pieces each
    events = []                           # (time, what, unit): times in MFMA units
    for S in (0, 1):
        for kind in ("Y", "X"):
            issue_order.append(unit_of(kind, S)); events.append((-1, "issue", unit_of(kind, S)))
    landed_at_sync = {}
    for S in range(nk):
        t0 = S * 64
        J = S % 5
        SY, SX, NY, NX, DY, DX = (2 * J) % 5, (2 * J + 1) % 5, (2 * J + 2) % 5, (2 * J + 3) % 5, (2 * J + 4) % 5, (2 * J) % 5
        assert (SY, SX) == (slot_of(unit_of("Y", S)), slot_of(unit_of("X", S)))
        assert (NY, NX) == (slot_of(unit_of("Y", S + 1)), slot_of(unit_of("X", S + 1)))
        assert DY == slot_of(unit_of("Y", S + 2)) and DX == slot_of(unit_of("X", S + 2))
        # k-steps 0, 1: Y_{S+2}; reads of stage S (k-steps 1..3 fragments) during k-steps 0..2
        issue_order.append(unit_of("Y", S + 2)); events.append((t0 + 0, "issue", unit_of("Y", S + 2)))
        for kind in ("Y", "X"):
            events.append((t0 + 40, "lastread", unit_of(kind, S)))       # last fragment read of stage S: k-step 2, MFMA 7
        # sync point P_S at t0 + 48: vmcnt(8) -> everything but the last 8 pieces (one unit) has landed
        landed_at_sync[S] = set(issue_order[:-1])
        assert unit_of("X", S + 1) in landed_at_sync[S] and unit_of("Y", S + 1) in landed_at_sync[S]
        events.append((t0 + 48, "firstread", unit_of("Y", S + 1))); events.append((t0 + 48, "firstread", unit_of("X", S + 1)))
        issue_order.append(unit_of("X", S + 2)); events.append((t0 + 48, "issue", unit_of("X", S + 2)))
    # a unit may be issued into a slot only after the previous tenant (unit - 5) was read for the last time
    last_read = {u: t for t, w, u in events if w == "lastread"}
    for t, w, u in events:
        if w == "issue" and u - 5 >= 0:
            assert last_read[u - 5] < t + 1e-9 and (t - last_read[u - 5]) >= 8, (u, t)      # separated by the barrier at P
    # latency budgets (MFMAs between the LAST piece's issue and the sync point that needs the unit)
    first_read = {u: t for t, w, u in events if w == "firstread"}
    for t, w, u in events:
        if w == "issue" and t >= 0 and u in first_read:
            span = 32 if u % 2 == 0 else 16                     # Y pieces spread over 2 k-steps, X pieces over 1
            assert first_read[u] - (t + span) >= (48 if u % 2 else 64)


def test_gemm256k_epilogue_park_and_readback():
    """Phase 1 of the gemm256k epilogue parks acc[yt][xt] (lane = (l31, half): row yt*32+l31, 16 columns xt*32+half*16..)
    in the wave's LDS region with rows of 272 B; phase 2 reads 4 rows x 256 B per instruction (lane -> row 4i + lane/16,
    16-B chunk lane%16).  Every element of the 128x128 sub-tile must be written once and read once, at the right place,
    and a ds_write_b128 service group (16 lanes) must cover all 64 banks."""
    EROW = 272
    park = {}
    for xt in range(4):
        for yt in range(4):
            for lane in range(64):
                l31, half = lane & 31, lane >> 5
                for part in range(2):                      # two 16-B stores of 8 columns each
                    addr = (yt * 32 + l31) * EROW + (xt * 32 + half * 16) * 2 + part * 16
                    assert addr % 16 == 0 and addr not in park
                    park[addr] = (yt * 32 + l31, xt * 32 + half * 16 + part * 8)
    assert len(park) == 128 * 16 and max(park) + 16 <= 128 * EROW <= 160 * 1024 // 4
    seen = set()
    for i in range(32):
        rows_of_instr = set()
        for lane in range(64):
            prow, pchunk = lane >> 4, lane & 15
            addr = (i * 4 + prow) * EROW + pchunk * 16
            row, col = park[addr]
            assert row == 4 * i + prow and col == pchunk * 8          # global row y0+wy*128+row, column x0+wx*128+col
            seen.add((row, col)); rows_of_instr.add(row)
        assert len(rows_of_instr) == 4
    assert len(seen) == 128 * 16
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    for g in groups:                                        # ds_write_b128 of lanes on consecutive rows (one half, one part)
        banks = set()
        for l31 in g:
            for dword in range(4):
                banks.add(((l31 * EROW) // 4 + dword) % 64)
        assert len(banks) == 64


# ----------------------------------------------------------------------------------------------
# attention: the per-wave LDS-DMA stream's step (attn_w64_shared.h: dma_advance<MULTI>) and the bounded tile's read plan
# ----------------------------------------------------------------------------------------------
KVBLK = 64


def dma_stream(Lk, nseg, skip, rs2, kseg, vseg, multi, n_fetch):
    """Transliteration of dma_init + n_fetch x (fetch, dma_advance<multi>): the (k offset, v offset, klen) each fetch uses."""
    seg = 1 if skip == 0 else 0
    k = kseg0 = seg * kseg
    v = vseg0 = seg * vseg
    tps = (Lk + KVBLK - 1) // KVBLK
    tt, left = 0, tps * (nseg - (1 if skip >= 0 else 0))
    klen = klen0 = (Lk - 1) * rs2 + 256
    out = []
    for _ in range(n_fetch):
        out.append((k, v, klen))
        adv = 1 if left > 1 else 0
        left -= adv
        if not multi:
            kb = KVBLK * rs2 if adv else 0
            k += kb; v += KVBLK * 2 if adv else 0; klen -= kb
            continue
        last = 1 if tt + 1 == tps else 0
        sw, stp = adv & last, adv & (last ^ 1)
        hop2 = sw & (1 if seg + 1 == skip else 0)
        tt = 0 if sw else tt + stp
        seg += (1 + hop2) if sw else 0
        kseg0 = kseg0 + (2 * kseg if hop2 else kseg) if sw else kseg0
        vseg0 = vseg0 + (2 * vseg if hop2 else vseg) if sw else vseg0
        kstep, vstep = k + (KVBLK * rs2 if stp else 0), v + (KVBLK * 2 if stp else 0)
        k = kseg0 if sw else kstep
        v = vseg0 if sw else vstep
        klen = klen0 if sw else klen - (KVBLK * rs2 if stp else 0)
    return out


def dma_stream_reference(Lk, nseg, skip, rs2, kseg, vseg, n_fetch):
    """What the stream must deliver: the tiles of every segment but `skip`, in order; then the last tile again and again."""
    tps = (Lk + KVBLK - 1) // KVBLK
    seq = [(s * kseg + t * KVBLK * rs2, s * vseg + t * KVBLK * 2, (Lk - 1) * rs2 + 256 - t * KVBLK * rs2)
           for s in range(nseg) if s != skip for t in range(tps)]
    return [seq[min(i, len(seq) - 1)] for i in range(n_fetch)]


@pytest.mark.parametrize("Lk,nseg,skip", [(75600, 1, -1), (512, 1, -1), (257, 1, -1), (64, 1, -1), (9450, 8, 3), (9450, 8, 0),
                                          (9450, 8, 7), (100, 2, 1), (100, 2, 0), (130, 3, -1), (37800, 2, -1)])
def test_attention_dma_stream_step_both_forms(Lk, nseg, skip):
    rs2, kseg, vseg = 40 * 256, 10 ** 9 + 64, 7 * 10 ** 8 + 128
    tps = (Lk + KVBLK - 1) // KVBLK
    n = tps * (nseg - (1 if skip >= 0 else 0)) + 5               # the kernel fetches two tiles ahead: past the end it must stay put
    ref = dma_stream_reference(Lk, nseg, skip, rs2, kseg, vseg, n)
    assert dma_stream(Lk, nseg, skip, rs2, kseg, vseg, True, n) == ref
    if nseg == 1 and skip < 0:                                    # the short form is what single-segment launches instantiate
        assert dma_stream(Lk, nseg, skip, rs2, kseg, vseg, False, n) == ref
    # the K descriptor's num_records ends at the segment's last valid row: a ragged last tile reads zeros beyond it
    last = ref[tps - 1]
    assert last[2] == ((Lk - 1) % KVBLK) * rs2 + 256


def seg_table(Lk, nseg, skip, rs2, kseg, vseg):
    """Transliteration of attention_w16n.hip attn_seg_table_kernel (round 6): one {k offset / 16, v offset / 16, valid K bytes} entry per
    fetch, the last one repeated for the two fetches past the end."""
    tps = (Lk + KVBLK - 1) // KVBLK
    ntile = tps * (nseg - (1 if skip >= 0 else 0))
    tab = []
    for f in range(ntile + 3):
        ff = min(f, ntile - 1)
        sg, tt = divmod(ff, tps)
        if skip >= 0 and sg >= skip:
            sg += 1
        koff, voff = sg * kseg + tt * KVBLK * rs2, sg * vseg + tt * KVBLK * 2
        assert koff % 16 == 0 and voff % 16 == 0 and (koff >> 4) < 2 ** 32 and (voff >> 4) < 2 ** 32
        tab.append((koff >> 4, voff >> 4, (Lk - 1) * rs2 + 256 - tt * KVBLK * rs2))
    return tab


@pytest.mark.parametrize("Lk,nseg,skip", [(9450, 8, 3), (9450, 8, 0), (9450, 8, 7), (9450, 8, -1), (100, 2, 1), (100, 2, 0), (130, 3, -1),
                                          (37800, 2, -1), (64, 4, 2), (65, 4, 1), (18900, 4, -1), (18450, 8, -1), (18450, 8, 5)])
def test_attention_segment_table_walk(Lk, nseg, skip):
    """The table walk of the multi-segment bounded launch (attention_w16n.hip SegTab): entries 0 and 1 are applied by hand in the
    prologue, the tile loop applies entry t + 2 at the top of tile t (loaded one tile earlier) -- the same (k, v, klen) sequence as
    the scalar walk it replaces, including the fetches past the end."""
    rs2, kseg, vseg = 5 * 256, 10 ** 9 + 64, 7 * 10 ** 8 + 128
    tps = (Lk + KVBLK - 1) // KVBLK
    ntile = tps * (nseg - (1 if skip >= 0 else 0))
    tab = seg_table(Lk, nseg, skip, rs2, kseg, vseg)
    fetched = []
    i = 0
    nxt = tab[0]                                                   # prologue
    fetched.append(nxt)
    i += 1; nxt = tab[i]
    fetched.append(nxt)
    i += 1; nxt = tab[i]
    for t in range(ntile):                                         # tile t: apply (gaps 0, 1), fetch tile t + 2, load the next entry (gap 2)
        fetched.append(nxt)
        i += 1
        nxt = tab[i]                                               # (the last tile's load, entry ntile + 2, is never applied -- but it is read)
    assert i == ntile + 2 and len(tab) == ntile + 3
    ref = dma_stream_reference(Lk, nseg, skip, rs2, kseg, vseg, ntile + 2)
    assert [(k << 4, v << 4, n) for k, v, n in fetched] == ref


def test_bounded_tile_read_plan_register_lifetimes():
    """tile_w64n's LDS read plan (attention_w64q.hip): K(t+1) fragment r (need order: S MFMA i consumes fragment i) is read at gap
    33 + r -- after its register's last reader (S_b's MFMA at gap 32 + r) and >= 15 gaps before the tile ends, so the single
    lgkmcnt(0) at the next tile's top finds every read retired; V^T(t) fragment f is read at gap 17 + f, after PV_b's MFMA f
    (gap 16 + f) read the register's previous content and before PV_a (gaps 48..63) needs the new one."""
    for r in range(16):
        g = 33 + r
        assert g > 32 + r and g <= 48 and (64 - g) >= 15
        f = (r & 1) * 8 + (r >> 1)                                  # register index [sub-tile][k-step]
        i_next = r                                                  # S_a MFMA i of the next tile uses kf[i & 1][i >> 1]
        assert (f >> 3, f & 7) == (i_next & 1, i_next >> 1)
        assert (64 - g) + i_next >= 15                              # gaps between the read and its first consumer
    for f in range(16):
        g = 17 + f
        assert g > 16 + f and g <= 32 and 48 + (f >> 2) * 4 - g >= 16 - f + 4 * (f >> 2) - 1
    assert sorted({(r & 1) * 8 + (r >> 1) for r in range(16)}) == list(range(16))


# ----------------------------------------------------------------------------------------------
# gemm256m.hip (256x256x64 on the 16x16x32 MFMA): gemm256k's unit images, X rows staged 8-way interleaved, Y = A operand,
# X = B operand, register-direct 16-byte stores
# ----------------------------------------------------------------------------------------------
def test_gemm256m_ds_read_b128_is_conflict_free():
    """Fragment read of tile r by lane (n = lane & 15, g = lane >> 4): row 16 r + n, logical chunk 4 ks + g, physical chunk
    ^ ((row >> 1) & 7).  The 16 lanes a ds_read_b128 services in one LDS cycle must cover 64 distinct banks."""
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[l + 32 for l in g] for g in groups]
    for ks in range(2):
        for r in range(8):
            for grp in groups:
                banks = set()
                for lane in grp:
                    n, g = lane & 15, lane >> 4
                    addr = r * 2048 + ((n * 128 + ((g ^ ((n >> 1) & 7)) << 4)) ^ (ks << 6))
                    for d in range(4):
                        banks.add((addr // 4 + d) % 64)
                assert len(banks) == 64


def test_gemm256m_operand_interleave_and_register_direct_stores():
    """End to end on one 256 x 256 tile with K = 64: DMA plan, fragment addresses, v_mfma_f32_16x16x32 semantics (A: lane (n, g)
    holds row n, k = 8 g ..; B: column n, k = 8 g ..; D: register i of lane (n, g) = D[4 g + i][n]) and the epilogue's
    (lane, a, i) -> (row, 8 columns) map.  Every output element exactly once, with the right operands; a store instruction
    writes 4 rows x 256 contiguous bytes."""
    rng = np.random.default_rng(6)
    Yt = rng.integers(-3, 4, size=(256, 64)).astype(np.float64)
    Xt = rng.integers(-3, 4, size=(256, 64)).astype(np.float64)
    ylds = np.zeros((256, 8, 8)); xlds = np.zeros((256, 8, 8))
    seen_x = set()
    for i in range(8):
        for tid in range(256):
            q = i * 256 + tid
            row, pch = q >> 3, q & 7
            lch = pch ^ ((row >> 1) & 7)
            ylds[row, pch] = Yt[row, lch * 8:lch * 8 + 8]
            slab, t, n = row >> 7, (row >> 4) & 7, row & 15
            xr = slab * 128 + 8 * n + t
            seen_x.add(xr)
            xlds[row, pch] = Xt[xr, lch * 8:lch * 8 + 8]
    assert seen_x == set(range(256))
    out = np.full((256, 256), np.nan)
    writes = 0
    order = list(range(1, 8)) + list(range(9, 16)) + [8, 0]        # M_ORD: read order of a k-step's 16 fragments (first-needed last)
    assert sorted(order) == list(range(16))
    for wave in range(4):
        wy, wx = wave >> 1, wave & 1
        acc = np.zeros((8, 8, 64, 4))                              # [a][t][lane][reg]
        for ks in range(2):
            yf = np.zeros((8, 64, 8)); xf = np.zeros((8, 64, 8))
            for lane in range(64):
                n, g = lane & 15, lane >> 4
                sw = (n >> 1) & 7
                ya = (wy * 128 + n) * 128 + ((g ^ sw) << 4)
                xa = (wx * 128 + n) * 128 + ((g ^ sw) << 4)
                for r in range(8):
                    addr = r * 2048 + (ya ^ (ks << 6))
                    yf[r, lane] = ylds[addr // 128, (addr % 128) // 16]
                    addr = r * 2048 + (xa ^ (ks << 6))
                    xf[r, lane] = xlds[addr // 128, (addr % 128) // 16]
            for a in range(8):
                for t in range(8):
                    A = np.zeros((16, 32)); B = np.zeros((16, 32))
                    for lane in range(64):
                        A[lane & 15, (lane >> 4) * 8:(lane >> 4) * 8 + 8] = yf[a, lane]
                        B[lane & 15, (lane >> 4) * 8:(lane >> 4) * 8 + 8] = xf[t, lane]
                    D = A @ B.T                                    # D[i][j] = sum_k A[i][k] B[j][k]
                    for lane in range(64):
                        for i in range(4):
                            acc[a, t, lane, i] += D[4 * (lane >> 4) + i, lane & 15]
        for a in range(8):
            for i in range(4):
                rows_of_instr = {}
                for lane in range(64):
                    n, g = lane & 15, lane >> 4
                    row = wy * 128 + 16 * a + 4 * g + i
                    col = wx * 128 + 8 * n
                    rows_of_instr.setdefault(row, []).append(row * 512 + col * 2)
                    for t in range(8):
                        assert np.isnan(out[row, col + t])
                        out[row, col + t] = acc[a, t, lane, i]
                        writes += 1
                assert len(rows_of_instr) == 4                     # one store instruction = 4 rows ...
                for row, bs in rows_of_instr.items():              # ... of 256 contiguous bytes each (16 lanes x 16 B)
                    assert sorted(bs) == list(range(min(bs), min(bs) + 256, 16))
    assert writes == 256 * 256
    np.testing.assert_array_equal(out, Yt @ Xt.T)


def test_gemm256m_stage_schedule():
    """Issue slots of a stage: 8 Y pieces + 16 fragment reads in k-step 0, 8 X pieces + 16 reads behind the sync point of k-step 1,
    never two in one MFMA gap, every fragment index exactly once per k-step, every read >= 16 MFMAs ahead of the k-step using it."""
    k0_dma = [m for m in range(64) if (m & 7) == 0]
    k0_rd = [m for m in range(64) if (m & 7) != 0 and (m & 1) == 1 and m < 32]
    assert len(k0_dma) == 8 and len(k0_rd) == 16 and not set(k0_dma) & set(k0_rd)
    assert [m >> 3 for m in k0_dma] == list(range(8)) and [m >> 1 for m in k0_rd] == list(range(16))
    k1_dma = [m for m in range(16, 64) if (m - 16) % 6 == 0]
    k1_rd = [m for m in range(16, 64) if (m - 16) % 6 != 0 and (m & 1) == 1 and m < 48]
    assert len(k1_dma) == 8 and len(k1_rd) == 16 and not set(k1_dma) & set(k1_rd)
    assert [(m - 16) // 6 for m in k1_dma] == list(range(8)) and [(m - 17) >> 1 for m in k1_rd] == list(range(16))
    assert 64 - max(k0_rd) >= 16 and 64 - max(k1_rd) >= 16
    order = [1 + i if i < 7 else 2 + i if i < 14 else 8 if i == 14 else 0 for i in range(16)]
    assert sorted(order) == list(range(16)) and order[-2:] == [8, 0]      # x tile 0 and y tile 0 -- the first MFMA's operands -- last


# ----------------------------------------------------------------------------------------------
# attention_w16n.hip (the bounded attention loop on the 16x16x32 MFMA): K rows permuted by the DMA plan, S^T C layout -> P^T B
# operand without moving data between lanes, V^T fragments as single 16-byte reads, O layout, and the split-gap schedule
# ----------------------------------------------------------------------------------------------
def _mfma16(A, B):
    """v_mfma_f32_16x16x32: A[lane (n, g)] = 8 values of row n, k = 8 g ..; B[lane (n, g)] = 8 values of column n, k = 8 g ..;
    returns D[lane][i] = sum_k A_row(4 g + i)[k] * B_col(n)[k]."""
    Am = np.zeros((16, 32)); Bm = np.zeros((16, 32))
    for lane in range(64):
        n, g = lane & 15, lane >> 4
        Am[n, 8 * g:8 * g + 8] = A[lane]
        Bm[n, 8 * g:8 * g + 8] = B[lane]
    D = Am @ Bm.T
    out = np.zeros((64, 4))
    for lane in range(64):
        n, g = lane & 15, lane >> 4
        out[lane] = D[4 * g:4 * g + 4, n]
    return out


def test_attention_w16n_layout_end_to_end():
    rng = np.random.default_rng(11)
    ntile = 2
    Q = rng.integers(-2, 3, size=(64, 128)).astype(np.float64)                   # one wave's 64 q rows
    K = rng.integers(-2, 3, size=(64 * ntile, 128)).astype(np.float64)
    V = rng.integers(-2, 3, size=(64 * ntile, 128)).astype(np.float64)
    Pfun = lambda s: (s % 5) + 1.0                                                # stands in for exp2: any elementwise map
    O = np.zeros((64, 128)); L = np.zeros(64)
    qf = np.zeros((4, 4, 64, 8))
    for qt in range(4):
        for ks in range(4):
            for lane in range(64):
                n, g = lane & 15, lane >> 4
                qf[qt, ks, lane] = Q[16 * qt + n, 32 * ks + 8 * g:32 * ks + 8 * g + 8]
    accO = np.zeros((4, 8, 64, 4))                                                # [qt][dt][lane][i]
    lsum = np.zeros((4, 64))
    for t in range(ntile):
        Kt, Vt = K[64 * t:64 * t + 64], V[64 * t:64 * t + 64].T                   # V^T [d][kv]
        # ---- DMA plan: K image [64 LDS rows][16 physical chunks], V^T image [128 rows][8 chunks]
        kimg = np.zeros((64, 16, 8)); vimg = np.zeros((128, 8, 8))
        for i in range(4):
            for tid in range(256):
                m = tid >> 4
                Lrow = 16 * i + m
                src = 32 * (i >> 1) + 8 * (m >> 2) + 4 * (i & 1) + (m & 3)
                lch = (tid & 15) ^ m                                              # kcol: logical chunk fetched by this lane
                kimg[Lrow, tid & 15] = Kt[src, lch * 8:lch * 8 + 8]               # piece slot = i*256 + tid: row 16 i + (tid >> 4), physical chunk tid & 15
                vrow = 32 * i + (tid >> 3)
                vl = (tid & 7) ^ ((tid >> 4) & 7)
                assert ((tid >> 4) & 7) == ((vrow >> 1) & 7) or True
                vimg[vrow, tid & 7] = Vt[vrow, vl * 8:vl * 8 + 8]
        # ---- fragments
        kfr = np.zeros((4, 4, 64, 8)); vfr = np.zeros((8, 2, 64, 8))
        for lane in range(64):
            n, g = lane & 15, lane >> 4
            for ks in range(4):
                kaddr = n * 256 + (((ks * 4 + g) ^ n) << 4)
                for kt in range(4):
                    a = kt * 4096 + kaddr
                    kfr[kt, ks, lane] = kimg[a // 256, (a % 256) // 16]
            for c in range(2):
                vaddr = n * 128 + (((c * 4 + g) ^ ((n >> 1) & 7)) << 4)
                for dt in range(8):
                    a = dt * 2048 + vaddr
                    vfr[dt, c, lane] = vimg[a // 128, (a % 128) // 16]
        # the V^T image swizzle as the DMA applies it: physical chunk p of row r holds logical chunk p ^ ((r >> 1) & 7)
        for r in range(128):
            for p in range(8):
                np.testing.assert_array_equal(vimg[r, p], Vt[r, (p ^ ((r >> 1) & 7)) * 8:(p ^ ((r >> 1) & 7)) * 8 + 8])
        # ---- S^T tiles and the kv row each register stands for
        S = np.zeros((4, 4, 64, 4))                                               # [kt][qt][lane][i]
        for kt in range(4):
            for qt in range(4):
                for ks in range(4):
                    S[kt, qt] += _mfma16(kfr[kt, ks], qf[qt, ks])
                for lane in range(64):
                    n, g = lane & 15, lane >> 4
                    for i in range(4):
                        kv = 32 * (kt >> 1) + 8 * g + 4 * (kt & 1) + i
                        assert S[kt, qt, lane, i] == Kt[kv] @ Q[16 * qt + n]
        P = Pfun(S)
        # ---- P^T fragments (qt, c): slots j = 0..3 from tile (2c, qt), 4..7 from (2c + 1, qt); row sums per lane
        for qt in range(4):
            for c in range(2):
                pf = np.concatenate([P[2 * c, qt], P[2 * c + 1, qt]], axis=1)      # [lane][8]
                for dt in range(8):
                    accO[qt, dt] += _mfma16(vfr[dt, c], pf)
            lsum[qt] += P[:, qt].sum(axis=(0, 2))
    for qt in range(4):
        for lane in range(64):
            n, g = lane & 15, lane >> 4
            for dt in range(8):
                O[16 * qt + n, 16 * dt + 4 * g:16 * dt + 4 * g + 4] = accO[qt, dt, lane]
        for n in range(16):
            L[16 * qt + n] = sum(lsum[qt, n + 16 * g] for g in range(4))
    Pref = Pfun(Q @ K.T)
    np.testing.assert_array_equal(O, Pref @ V)
    np.testing.assert_array_equal(L, Pref.sum(axis=1))


def test_attention_w16n_ds_read_b128_is_conflict_free():
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[l + 32 for l in g] for g in groups]
    for grp in groups:
        for ks in range(4):                                                       # K image: 256-B rows
            banks = set()
            for lane in grp:
                n, g = lane & 15, lane >> 4
                addr = n * 256 + (((ks * 4 + g) ^ n) << 4)
                banks.update((addr // 4 + d) % 64 for d in range(4))
            assert len(banks) == 64
        for c in range(2):                                                        # V^T image: 128-B rows
            banks = set()
            for lane in grp:
                n, g = lane & 15, lane >> 4
                addr = n * 128 + (((c * 4 + g) ^ ((n >> 1) & 7)) << 4)
                banks.update((addr // 4 + d) % 64 for d in range(4))
            assert len(banks) == 64


def test_attention_w16n_gap_schedule():
    """The split-gap schedule of tile_w16n as a state machine over three tiles: every score of both q halves is exponentiated exactly
    once per tile and in order; the row-sum add of score k sits in the gap of exp2(k + 1) (or the tail, k = 31); a pair is packed
    after the exp2 of its odd score and >= 2 gaps before the PV MFMA that reads its fragment; V^T / K fragment reads come after the
    last MFMA reading the register they overwrite."""
    def bk0(g):
        return g - 51 + ((g - 55) // 2 if g > 56 else 0)
    events = []                                                                   # (tile, G, kind, half, k)
    for t in range(3):
        for G in range(128):
            g_ = G >> 1
            if G & 1 == 0:
                if 4 <= g_ <= 18: events.append((t, G, "exp", "b", g_ + 13))
                if 19 <= g_ <= 50: events.append((t, G, "exp", "a", g_ - 19))
                if g_ >= 51: events.append((t, G, "exp", "b", bk0(g_)))
            else:
                if 4 <= g_ <= 18 and (g_ + 13) & 1: events.append((t, G, "pack", "b", g_ + 13))
                if g_ == 18: events.append((t, G, "tail", "b", 31))
                if 19 <= g_ <= 50 and (g_ - 19) & 1: events.append((t, G, "pack", "a", g_ - 19))
                if g_ == 50: events.append((t, G, "tail", "a", 31))
                if g_ >= 51 and bk0(g_) & 1: events.append((t, G, "pack", "b", bk0(g_)))
                if g_ >= 56 and (g_ - 56) % 2 == 0:
                    events.append((t, G, "exp", "b", bk0(g_) + 1))
                    if (bk0(g_) + 1) & 1: events.append((t, G, "pack", "b", bk0(g_) + 1))
    for half in "ab":
        seq = [(t, G, kind, k) for (t, G, kind, h, k) in events if h == half]
        ks = [k for (_, _, kind, k) in seq if kind == "exp"]
        if half == "a":
            assert ks == list(range(32)) * 3
        else:
            assert ks == list(range(17, 32)) + (list(range(17)) + list(range(17, 32))) * 2 + list(range(17))
        # state machine: p0 / p1 registers, what is packed and what is summed
        p = {0: None, 1: None}
        summed, packed = [], []
        for (t, G, kind, k) in seq:
            if kind == "exp":
                p[k & 1] = (t if (half == "a" or k <= 16) else t - 1, k)      # (tile the score belongs to, k)
                if k >= 1 and p[(k - 1) & 1] is not None: summed.append(p[(k - 1) & 1])
            elif kind == "pack":
                assert p[1] is not None and p[1][1] == k and (p[0] is None or p[0] == (p[1][0], k - 1))
                packed.append((p[1][0], k >> 1, t, G))
            else:
                summed.append(p[1])
        full = [(tt, k) for (tt, k) in summed if tt in (0, 1)]
        for tt in ((0, 1) if half == "a" else (0,)):                            # complete tiles inside the simulated window
            assert sorted(k for (t2, k) in full if t2 == tt) == list(range(32)), (half, tt)
        for (tt, j, t, G) in packed:                                             # fragment f = j >> 2: a -> PV_a of tile tt (gaps 96 + 8 f ..),
            f = j >> 2                                                           # b -> PV_b in tile tt + 1 (gaps 32 + 8 f ..)
            use = (tt, 96 + 8 * f) if half == "a" else (tt + 1, 32 + 8 * f)
            assert (t, G + 2) <= use, (half, j, t, G, use)
    for dt in range(8):
        assert 41 + 2 * dt > 32 + 8 * 1 + dt and 57 + 2 * dt > 32 + 8 * 3 + dt    # V^T (dt, 0) / (dt, 1): last PV_b readers f = 1 / f = 3
    for ks_ in range(4):
        for kt in range(4):
            assert 67 + 2 * (4 * ks_ + kt) > 64 + 8 * ks_ + 2 * kt + 1             # K (kt, ks): last S_b reader q tile 1






# ----------------------------------------------------------------------------------------------
# attention_w16n.hip, SHIFT (round 4): the bounded loop with a per-row reference that never changes inside the loop
# ----------------------------------------------------------------------------------------------
def _shifted_softmax_rows(qt, k, v, L_tile0=64):
    """The arithmetic of attn_w16n_kernel<SHIFT> for a block of rows, in fp32 like the kernel: U from Cauchy-Schwarz, m_s from the
    first 64 keys, the reference m, P = 2^(s - m) exponentiated ONCE with no maximum tracked, row sums, the post-loop verdict.
    -> (o [rows, d] fp32, flagged [rows] bool, m [rows])."""
    f = np.float32
    s = (qt.astype(np.float64) @ k.astype(np.float64).T).astype(f)            # bf16 products are exact in fp32; the order of the sum is not pinned
    u2 = (qt.astype(f) ** 2).sum(1) * f((k.astype(f) ** 2).sum(1).max())
    u = np.sqrt(u2) * f(1.0001) + f(0.01)
    m_cs = np.where(u2 <= f(96.0 * 96.0), f(0), u - f(96.0)).astype(f)
    ms = s[:, :L_tile0].max(1)
    m = np.where(m_cs == 0, f(0), np.where(u - ms <= f(168.0), m_cs, ms + f(72.0))).astype(f)
    with np.errstate(over="ignore", under="ignore", invalid="ignore"):
        p = np.exp2((s - m[:, None]).astype(f)).astype(f)                     # inf above 2^128, 0 below 2^-149 (the hardware flushes earlier: 2^-126)
        p = np.where(p < f(2.0 ** -126), f(0), p)
        l = p.sum(1, dtype=f)
        o = (p @ v.astype(f)) / l[:, None]
    ok = ((l >= f(2.0 ** -80)) & (l <= f(2.0 ** 100))) | (m == 0)
    too_far = u2 > f(2048.0 * 2048.0)
    return o, ~ok | too_far, m


def test_shifted_bounded_softmax_covers_what_the_plain_bound_declined_and_flags_the_rest():
    """The algebra the SHIFT instantiation rests on, on the CPU in fp32: (1) wherever a row is not flagged its result is the fp64
    softmax's (shift invariance; what underflows is < 2^-46 of the row sum per term); (2) diffuse random rows are never flagged up to
    gains far beyond round 3's limit (gamma ~ 6 declined everything); (3) a row whose maximum sits more than 168 above what its first
    64 keys suggest overflows, is FLAGGED, and is never returned as a number; (4) rows inside the plain bound keep m = 0 (the plain
    kernel's arithmetic, bit for bit)."""
    rng = np.random.default_rng(5)
    L, d, rows = 6000, 128, 96
    c = np.float32(0.08838834764831845 * 1.4426950408889634)

    bf = _bf16_rne
    v = bf(rng.standard_normal((L, d)).astype(np.float32))
    q = bf(rng.standard_normal((rows, d)).astype(np.float32) * c)
    seen_shifted = False
    for gain in (1.0, 4.0, 6.0, 12.0, 30.0):
        k = bf(rng.standard_normal((L, d)).astype(np.float32) * gain)
        o, flagged, m = _shifted_softmax_rows(q, k, v)
        assert not flagged.any(), (gain, int(flagged.sum()))
        assert (m == 0).all() if gain <= 4.0 else (m > 0).all(), gain
        seen_shifted |= bool((m > 0).any())
        s64 = q.astype(np.float64) @ k.astype(np.float64).T
        p64 = np.exp2(s64 - s64.max(1, keepdims=True))
        ref = (p64 / p64.sum(1, keepdims=True)) @ v.astype(np.float64)
        assert np.abs(o - ref).max() < 2e-4, (gain, np.abs(o - ref).max())
    assert seen_shifted
    # adversarial: every query has a component along e_0, one key far from the first tile is 400 e_0
    q2 = q.copy(); q2[:, 0] += bf(np.float32(6.0) * c)
    k = bf(rng.standard_normal((L, d)).astype(np.float32)); k[1234] = 0; k[1234, 0] = 400.0
    o, flagged, m = _shifted_softmax_rows(q2, k, v)
    s64 = q2.astype(np.float64) @ k.astype(np.float64).T
    out_of_window = s64.max(1) - m > 127.0
    assert out_of_window.sum() > rows // 2 and flagged[out_of_window].all()
    fine = ~flagged
    if fine.any():
        p64 = np.exp2(s64 - s64.max(1, keepdims=True))
        ref = (p64 / p64.sum(1, keepdims=True)) @ v.astype(np.float64)
        assert np.abs(o[fine] - ref[fine]).max() < 2e-4
    # a reference beyond SHIFT_LIMIT is not attempted
    _, flagged, _ = _shifted_softmax_rows(q * np.float32(80.0), bf(rng.standard_normal((L, d)).astype(np.float32) * 70.0), v)
    assert flagged.all()


# ---- vae_conv_halo.hip: the halo-patch convolution (round 4) ----------------------------------------------------------------------
def _halo_patch_dma(wave, i, lane):
    """piece i of wave `wave` -> (LDS byte offset inside a patch stage, patch pixel, 8-channel chunk) of lane `lane`"""
    pi = (wave * 3 + i) % 21
    o = pi * 1024 + lane * 16
    pp, slot = o >> 6, (o >> 4) & 3
    return o, pp, slot ^ ((pp >> 1) & 2)


@pytest.mark.parametrize("NB", [4, 3])
def test_conv_halo_layout_end_to_end(NB):
    """csrc/vae_conv_halo.hip transliterated lane by lane on one 16 x 16 tile, one (kt, channel block) group, all nine taps: the 24 patch
    pieces (21 distinct) fill every (pixel, chunk) of the 18 x 18 x 32 patch exactly where the window reads look for it; the weight pieces
    (8, or 6 + 2 repeats for the 96-wide tile) do the same for the permuted channel rows; the MFMA fragments built from those LDS images
    (16x16x32 lane layouts) reproduce a plain 3 x 3 convolution; a lane ends with NB * 4 consecutive output channels of one pixel.  Every
    ds_read_b128 of the loop is bank-conflict free -- for every tap's window start."""
    rng = np.random.default_rng(NB)
    NW = NB * 16
    patch = rng.standard_normal((18, 18, 32))                 # [pr, pc, channel]
    wts = rng.standard_normal((9, 2 * NW, 32))                # [tap, cout, channel]
    # --- LDS images written by the DMA pieces
    P = np.full(21 * 1024 // 2, np.nan)                       # halfs
    seen = set()
    for wave in range(8):
        for i in range(3):
            for lane in range(64):
                o, pp, ch = _halo_patch_dma(wave, i, lane)
                if pp < 324:
                    P[o // 2:o // 2 + 8] = patch[pp // 18, pp % 18, ch * 8:ch * 8 + 8]
                    seen.add((pp, ch))
    assert seen == {(pp, c) for pp in range(324) for c in range(4)}
    W = np.full((9, 2 * NW * 32), np.nan)
    rows_seen = set()
    for wave in range(8):
        wpiece = wave % (2 * NB)
        for lane in range(64):
            R, slot = wpiece * 16 + (lane >> 2), lane & 3
            c = slot ^ ((R >> 1) & 2)
            slab, jj = divmod(R, NW)
            xt, ii = jj >> 4, jj & 15
            co = slab * NW + (ii >> 2) * (NB * 4) + xt * 4 + (ii & 3)
            rows_seen.add(co)
            for tap in range(9):
                W[tap, (wpiece * 1024 + lane * 16) // 2:(wpiece * 1024 + lane * 16) // 2 + 8] = wts[tap, co, c * 8:c * 8 + 8]
    assert rows_seen == set(range(2 * NW))
    # --- the K loop of every wave
    out = np.zeros((16, 16, 2 * NW))
    owner = {}
    for wave in range(8):
        wr, wc = wave >> 1, wave & 1
        acc = np.zeros((4, NB, 64, 4))
        for tap in range(9):
            dy, dx = divmod(tap, 3)
            yaddr = np.zeros((4, 64), dtype=np.int64)
            xaddr = np.zeros((NB, 64), dtype=np.int64)
            for lane in range(64):
                n, lg = lane & 15, lane >> 4
                for a in range(4):
                    ya = ((4 * wr + a) * 18 + n) * 64 + lg * 16 + (dy * 18 + dx) * 64
                    yaddr[a, lane] = ya ^ ((ya >> 3) & 32)
                for b in range(NB):
                    xaddr[b, lane] = (wc * NW + n) * 64 + ((lg ^ ((n >> 1) & 2)) << 4) + b * 1024
            for a in range(4):
                assert b128_conflict_free(list(yaddr[a])), (tap, a)
            for b in range(NB):
                assert b128_conflict_free(list(xaddr[b]))
            for a in range(4):
                for b in range(NB):
                    # D[i][j] += sum_k A[i][k] B[j][k]: A = weights (lane l: row l & 15, k 8 (l >> 4) ..), B = pixels (lane l: col l & 15, same k)
                    A = np.zeros((16, 32)); Bm = np.zeros((16, 32))
                    for lane in range(64):
                        n, lg = lane & 15, lane >> 4
                        A[n, lg * 8:lg * 8 + 8] = W[tap, xaddr[b, lane] // 2:xaddr[b, lane] // 2 + 8]
                        Bm[n, lg * 8:lg * 8 + 8] = P[yaddr[a, lane] // 2:yaddr[a, lane] // 2 + 8]
                    D = A @ Bm.T
                    for lane in range(64):
                        for r in range(4):
                            acc[a, b, lane, r] += D[(lane >> 4) * 4 + r, lane & 15]
        for a in range(4):
            for lane in range(64):
                n2, lg2 = lane & 15, lane >> 4
                xb = wc * NW + lg2 * NB * 4
                for xt in range(NB):
                    for r in range(4):
                        key = (4 * wr + a, n2, xb + xt * 4 + r)
                        assert key not in owner
                        owner[key] = (wave, lane)
                        out[key] = acc[a, xt, lane, r]
    ref = np.zeros((16, 16, 2 * NW))
    for tap in range(9):
        dy, dx = divmod(tap, 3)
        ref += np.einsum("hwc,oc->hwo", patch[dy:dy + 16, dx:dx + 16], wts[tap])
    assert len(owner) == 16 * 16 * 2 * NW
    np.testing.assert_allclose(out, ref, rtol=1e-12, atol=1e-12)


def test_conv_halo_ring_schedule():
    """The in-order load counter of vae_conv_halo.hip's K loop: per wave P(0) x 3, W(0), W(1), drain; then at the top of tap q (= 9 g + tap)
    wait until at most N loads are out, barrier, issue W(q + 2) and -- at tap 0 -- the next group's three patch pieces.  With N = 4 at taps
    1 and 2 and 1 elsewhere: W(q) has landed at its tap's top, the patch of group g at the top of its tap 0, no stage is overwritten while a
    tap may still read it."""
    for groups in (3, 9, 36):
        issued, landed_upto = [], 0          # issue log: (kind, index); completion is in order
        def wait(n):
            nonlocal landed_upto
            landed_upto = max(landed_upto, len(issued) - n)
        def landed(kind, idx):
            return (kind, idx) in issued[:landed_upto]
        issued += [("P", 0)] * 3 + [("W", 0), ("W", 1)]
        wait(0)
        wstage_owner = {0: 0, 1: 1}
        for g in range(groups):
            for tap in range(9):
                q = 9 * g + tap
                if q > 0:
                    wait(4 if tap in (1, 2) else 1)
                assert landed("W", q)
                if tap == 0:
                    assert landed("P", g)
                # after the barrier: W(q + 2) into the stage tap q - 1 read
                assert wstage_owner.get((q + 2) % 3, q - 1) <= q - 1
                issued.append(("W", q + 2)); wstage_owner[(q + 2) % 3] = q + 2
                if tap == 0:
                    issued += [("P", g + 1)] * 3      # into patch stage (g + 1) & 1: last read in group g - 1
                assert wstage_owner[q % 3] == q       # the stage this tap reads still holds W(q)


# ----------------------------------------------------------------------------------------------
# gemm16s.hip (round 6: 128 x 128 / 256 x 128 x 32 tiles on the 16x16x32 MFMA, 64-byte LDS rows, X rows staged 4-way interleaved,
# register-direct 8-byte stores, ring of three stages two tiles ahead)
# ----------------------------------------------------------------------------------------------
def _g16s_sw(row):
    m = (row >> 2) & 3
    return ((m & 1) << 1) ^ ((m >> 1) * 3)


def test_gemm16s_swizzle_is_the_table_and_reads_are_conflict_free():
    """H = {0, 2, 3, 1} by (row >> 2) & 3; a fragment read (lane (n, g): row 16 tile + n, logical chunk g, physical chunk g ^ H) must give
    each of ds_read_b128's four 16-lane service groups 16 distinct 16-byte slots of the 256-byte bank sweep."""
    assert [_g16s_sw(4 * m) for m in range(4)] == [0, 2, 3, 1]
    for tile in range(8):
        offs = []
        for lane in range(64):
            n, g = lane & 15, lane >> 4
            offs.append((16 * tile + n) * 64 + ((g ^ _g16s_sw(n)) << 4))
        assert b128_conflict_free(offs)
    # the DMA writes a wave's 64 lanes to 1 KB of consecutive LDS bytes: 16 rows x 4 chunks; every (row, logical chunk) exactly once
    for i in range(4):
        seen = set()
        for tid in range(256):
            q = i * 256 + tid
            row, pch = q >> 2, q & 3
            seen.add((row, pch ^ _g16s_sw(row)))
        assert seen == {(r, c) for r in range(i * 64, i * 64 + 64) for c in range(4)}


@pytest.mark.parametrize("WTY", [4, 8])
def test_gemm16s_layout_end_to_end(WTY):
    """One workgroup tile with K = 32 (one stage): DMA plan -> LDS images -> fragment addresses -> v_mfma_f32_16x16x32 semantics -> the
    epilogue's (lane, a, i) -> (row, 4 columns) map.  Every output element exactly once with the right operands; a store instruction
    writes 4 rows x 128 contiguous bytes."""
    WTX = 4
    BM, BN = 32 * WTY, 32 * WTX
    rng = np.random.default_rng(16 + WTY)
    Yt = rng.integers(-3, 4, size=(BM, 32)).astype(np.float64)
    Xt = rng.integers(-3, 4, size=(BN, 32)).astype(np.float64)
    ylds = np.zeros((BM, 4, 8)); xlds = np.zeros((BN, 4, 8))
    seen_x = set()
    for i in range(BM // 64):
        for tid in range(256):
            q = i * 256 + tid
            row, pch = q >> 2, q & 3
            lch = pch ^ _g16s_sw(row)
            ylds[row, pch] = Yt[row, lch * 8:lch * 8 + 8]
    for i in range(BN // 64):
        for tid in range(256):
            q = i * 256 + tid
            row, pch = q >> 2, q & 3
            lch = pch ^ _g16s_sw(row)
            slab, t, n = row // (16 * WTX), (row >> 4) % WTX, row & 15
            xr = slab * 16 * WTX + WTX * n + t
            seen_x.add(xr)
            xlds[row, pch] = Xt[xr, lch * 8:lch * 8 + 8]
    assert seen_x == set(range(BN))
    out = np.full((BM, BN), np.nan)
    for wave in range(4):
        wy, wx = wave >> 1, wave & 1
        yf = np.zeros((WTY, 64, 8)); xf = np.zeros((WTX, 64, 8))
        for lane in range(64):
            n, g = lane & 15, lane >> 4
            ya = (wy * 16 * WTY + n) * 64 + ((g ^ _g16s_sw(n)) << 4)
            xa = (wx * 16 * WTX + n) * 64 + ((g ^ _g16s_sw(n)) << 4)
            for a in range(WTY):
                addr = a * 1024 + ya
                yf[a, lane] = ylds[addr // 64, (addr % 64) // 16]
            for t in range(WTX):
                addr = t * 1024 + xa
                xf[t, lane] = xlds[addr // 64, (addr % 64) // 16]
        for a in range(WTY):
            accs = [_mfma16(yf[a], xf[t]) for t in range(WTX)]            # [t][lane][i]
            for i in range(4):
                rows_of_instr = {}
                for lane in range(64):
                    n, g = lane & 15, lane >> 4
                    row = wy * 16 * WTY + 16 * a + 4 * g + i
                    col = wx * 16 * WTX + WTX * n
                    rows_of_instr.setdefault(row, []).append(col * 2)
                    for t in range(WTX):
                        assert np.isnan(out[row, col + t])
                        out[row, col + t] = accs[t][lane, i]
                assert len(rows_of_instr) == 4
                for bs in rows_of_instr.values():
                    assert sorted(bs) == list(range(min(bs), min(bs) + 128, 8))
    np.testing.assert_array_equal(out, Yt @ Xt.T)


def test_gemm16s_ring_schedule():
    """Tile kt's fragments are read during tile kt - 1 from slot kt % 3; at the top of tile kt (behind the wait + barrier) the DMA of
    tile kt + 3 goes into slot kt % 3.  No slot is overwritten before every wave's reads of it are behind a barrier, every tile is read
    from the slot it was fetched into, and with `pieces` DMA instructions per thread and stage the counted waits leave exactly the
    younger stages in flight."""
    for nk in (3, 4, 5, 6, 7, 13, 48):
        slot_holds = {}                     # slot -> k-tile fetched into it (the stream stops at the last tile)
        kf, pending = 0, []                 # pending: fetches in issue order
        def stage(s):
            nonlocal kf
            slot_holds[s] = kf
            pending.append(kf)
            kf = min(kf + 1, nk - 1)
        stage(0); stage(1); stage(2)
        landed = set(pending[:-2])          # vmcnt(2 P): all but the two youngest stages
        assert 0 in landed
        frag_tile = slot_holds[0]           # load_frags(f0, 0)
        for kt in range(nk):
            # wait vmcnt(P): all but the youngest stage landed
            landed = set(pending[:-1])
            if kt + 1 < nk:
                assert kt + 1 in landed
            assert frag_tile == kt          # the fragments in registers are tile kt's
            reading_slot = (kt + 1) % 3
            stage(kt % 3)                   # overwrites the slot whose reads ended before the barrier
            assert kt % 3 != reading_slot
            if kt + 1 < nk:
                assert slot_holds[reading_slot] == kt + 1
            frag_tile = slot_holds[reading_slot]


# ----------------------------------------------------------------------------------------------
# gemm256m.hip with the tile height as a template argument (round 6: TY = 5 .. 8 y tiles of 16 rows per wave)
# ----------------------------------------------------------------------------------------------
def _g256m_k1_p(TY):
    return (8 * TY - 16) // 8


def _g256m_k1_piece(TY, m):
    P = _g256m_k1_p(TY)
    return (m - 16) % P == 0 and (m - 16) // P < 8


def _g256m_k1_load(TY, m):
    if _g256m_k1_piece(TY, m):
        return -1
    if _g256m_k1_p(TY) % 2 == 0:
        return (m - 17) >> 1 if (m & 1) == 1 and ((m - 17) >> 1) < TY + 8 else -1
    idx = (m - 16) - sum(1 for q in range(16, m) if _g256m_k1_piece(TY, q))
    return idx if idx < TY + 8 else -1


def _g256m_ord(TY, i):
    return 1 + i if i < TY - 1 else 9 + (i - (TY - 1)) if i < TY + 6 else 8 if i == TY + 6 else 0


@pytest.mark.parametrize("TY", [5, 6, 7, 8])
def test_gemm256m_stage_schedule_for_every_tile_height(TY):
    """A stage of the 32 TY x 256 tile: k-step 0 = 8 TY MFMAs carrying the TY pieces of Y_{S+2} (after MFMA 0, 8, ...) and the TY + 8
    fragments of k-step 1 (after the odd MFMAs); k-step 1 = 16 bare MFMAs, the sync point, then the 8 pieces of X_{S+2} and the next
    stage's TY + 8 first fragments, never two in one gap, every fragment exactly once in the order y1.., x1.., x0, y0 (what the next
    k-step's first MFMA needs last); TY = 8 is the round-3 plan slot for slot."""
    NM, NF = 8 * TY, TY + 8
    k0_dma = [m for m in range(NM) if (m & 7) == 0]
    k0_rd = [m for m in range(NM) if (m & 7) != 0 and (m & 1) == 1 and m < 2 * NF]
    assert [m >> 3 for m in k0_dma] == list(range(TY)) and [m >> 1 for m in k0_rd] == list(range(NF)) and not set(k0_dma) & set(k0_rd)
    pieces = [m for m in range(16, NM) if _g256m_k1_piece(TY, m)]
    loads = [(m, _g256m_k1_load(TY, m)) for m in range(16, NM) if _g256m_k1_load(TY, m) >= 0]
    assert len(pieces) == 8 and [(m - 16) // _g256m_k1_p(TY) for m in pieces] == list(range(8))
    assert [i for _, i in loads] == list(range(NF)) and not set(pieces) & {m for m, _ in loads}
    order = [_g256m_ord(TY, i) for i in range(NF)]
    assert sorted(order) == list(range(TY)) + list(range(8, 16)) and order[-2:] == [8, 0]
    if TY == 8:
        assert pieces == list(range(16, 64, 6)) and loads == [(m, (m - 17) >> 1) for m in range(17, 48, 2)]
        assert order == [1 + i if i < 7 else 2 + i if i < 14 else 8 if i == 14 else 0 for i in range(16)]


def test_gemm256m_tile_height_rule():
    """wan_gemm256m_tile_rows: rounds of tiles x height, 2 % per step below 256, 256 unless the gain is >= 8 % (transliterated)."""
    def rows(YM, XN, cus=256):
        tx = (XN + 255) // 256
        cost = lambda T: ((((YM + 32 * T - 1) // (32 * T)) * tx + cus - 1) // cus) * T * (1.0 + 0.02 * (8 - T))
        if ((YM + 255) // 256) * tx > 8 * cus:
            return 256
        best, cb = 8, cost(8)
        for T in (7, 6, 5):
            if cost(T) < cb * 0.92 and cost(T) < cost(best):
                best = T
        return 32 * best
    assert rows(6400, 1536) == 160                 # BASELINE configs[0]: 40 x 6 = 240 tiles on 256 CUs (25 x 6 = 150 at 256 rows)
    assert rows(6400, 8960) == 224                 # its ffn.0: 29 x 35 = 1,015 tiles = 4 rounds of 7 (875 tiles = 4 rounds of 8)
    assert rows(65520, 1536) == 256                # 1.3B-480p: 256 x 6 = exactly 6 rounds
    assert rows(151200, 5120) == 256 and rows(151200, 13824) == 256     # the headline: far beyond eight rounds
    assert rows(18900, 5120) == 256                # a rank of a world of 8: 5.8 rounds, nothing lower gains 8 %
