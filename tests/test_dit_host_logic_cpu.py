"""CPU: the HOST LOGIC of the DiT forward driver (wan2gp_amd/csrc/dit.hip) on a recording stand-in for the kernels.

The forward is a composition of op-level launches (GEMMs, norms, attention, ...), each checked on the GPU against the oracle.
What composes them -- workspace carving, per-layer launch order, which weights go where, which streams a step-skipping call
touches, the pointer arithmetic of the NAG branch, the call order under sequence parallelism -- is host C++ and needs no GPU:
tests/mock/mock_ops.cpp defines every function dit.o calls (the library's op-level C entries with the prototypes of
include/wanhip.h, and the HIP runtime calls) as recorders, the REAL dit.hip is compiled and linked against it, and
`wan_dit_forward*` is driven through the same ctypes structures the product uses, with fake (never dereferenced) device
addresses.  Needs hipcc (to compile dit.hip's host code) and g++."""
import ctypes
import os
import shutil
import subprocess
from ctypes import POINTER, c_char_p, c_double, c_int, c_int64, c_uint64, c_void_p

import pytest

from oracle import wan_oracle as O
from wan2gp_amd.lib import DitArgs, DitConfig, GATHER_FN, GATHER_WAIT_FN, POLL_FN, SpInfo, WAN_ABORTED

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
pytestmark = pytest.mark.skipif(not (os.path.exists(HIPCC) and shutil.which("g++")), reason="needs hipcc and g++")

WS = 0x7000_0000_0000                                  # fake workspace base (256-aligned)
TL, TD = 512, 4096


class Call(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 32), ("p", c_uint64 * 8), ("i", c_int64 * 12), ("f", c_double * 4)]


@pytest.fixture(scope="module")
def mock(tmp_path_factory):
    d = tmp_path_factory.mktemp("mock")
    inc = os.path.join(ROOT, "include")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O1", "-std=c++17", "-fPIC", "-Wno-unused-function", "-I" + inc, "-c",
                    os.path.join(ROOT, "wan2gp_amd", "csrc", "dit.hip"), "-o", str(d / "dit.o")], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.run(["g++", "-std=c++17", "-O1", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-c",
                    os.path.join(ROOT, "tests", "mock", "mock_ops.cpp"), "-o", str(d / "mock.o")], check=True)
    subprocess.run(["g++", "-shared", "-fPIC", "-o", str(d / "libwanhip_mock.so"), str(d / "dit.o"), str(d / "mock.o")], check=True)
    L = ctypes.CDLL(str(d / "libwanhip_mock.so"))
    L.mock_get.restype = POINTER(Call)
    L.wan_last_error.restype = c_char_p
    L.wan_dit_workspace_bytes.restype = c_int64
    L.wan_dit_workspace_bytes.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_int]
    L.wan_dit_set_weight.argtypes = [c_void_p, c_char_p, c_void_p, c_int, c_int64]
    L.wan_dit_forward_ex.argtypes = [c_void_p, POINTER(DitArgs), c_void_p]
    L.wan_dit_forward_graph.argtypes = [c_void_p, POINTER(DitArgs), c_void_p, POINTER(c_int)]
    return L


class Model:
    """A context with every weight of the `tiny` config registered at a distinct fake address."""

    def __init__(self, L, name="tiny", fp8=False, mixed=False):
        self.L, self.cfg = L, O.make_config(name)
        c = self.cfg
        dc = DitConfig(c.dim, c.ffn_dim, c.num_heads, c.num_layers, c.in_dim, c.out_dim, c.text_dim, c.freq_dim, c.text_len, c.eps)
        self.ctx = c_void_p()
        assert L.wan_dit_create(ctypes.byref(dc), ctypes.byref(self.ctx)) == 0
        self.addr = {}
        for n, (k, shape) in enumerate(O.param_shapes(c).items()):
            numel = 1
            for s in shape:
                numel *= s
            a = 0x1000_0000_0000 + n * 0x10_0000_0000
            self.addr[k] = a
            dt = 1 if (k.startswith(("patch_embedding.", "head.")) or (mixed and O.is_fp32_locked(k, True))) else 0
            lin = k.endswith(".weight") and ".norm" not in k and "norm3" not in k and k.startswith("blocks.") and ("attn." in k or "ffn." in k)
            if fp8 and lin:
                dt = 2
                s_addr = a + 0x8_0000_0000
                assert L.wan_dit_set_weight(self.ctx, (k[:-len("weight")] + "scale_weight").encode(), c_void_p(s_addr), 1, shape[0]) == 0
            assert L.wan_dit_set_weight(self.ctx, k.encode(), c_void_p(a), dt, numel) == 0, L.wan_last_error()

    def forward(self, S=2, fhw=(2, 8, 8), should_calc=None, residual=None, nag=None, ctx_batches=None, sp=None, t_frames=None, poll=None,
                perturb=None, x_id=0, graph=False, t=637.0, x0=0x6000_0000_0000, stream=None, context_key=0):
        L, (F, H, W) = self.L, fhw
        shards = 1 if sp is None else sp.world
        nbytes = L.wan_dit_workspace_bytes(self.ctx, S, F, H, W, shards)
        assert nbytes > 0
        X = (c_void_p * S)(*[x0 + s * 0x1_0000_0000 for s in range(S)])
        C = (c_void_p * S)(*[0x6100_0000_0000 + s * 0x1_0000_0000 for s in range(S)])
        OUT = (c_void_p * S)(*[0x6200_0000_0000 + s * 0x1_0000_0000 for s in range(S)])
        FL = None if should_calc is None else (c_int * S)(*should_calc)
        RP = None if residual is None else (c_void_p * S)(*residual)
        TF = None if t_frames is None else (ctypes.c_float * F)(*t_frames)
        a = DitArgs(S, X, t, C, None, 0x6300_0000_0000, 0x6310_0000_0000, OUT, F, H, W, WS, nbytes,
                    None if sp is None else ctypes.cast(ctypes.byref(sp), c_void_p), None if poll is None else ctypes.cast(poll, c_void_p), None, FL, RP,
                    None, 1.0, TF, F if t_frames is not None else 0, 0, None, None,
                    *((0.0, 0.0, 0.0, None) if nag is None else (*nag, (c_int * S)(*ctx_batches))),
                    None if not perturb else (c_int * len(perturb))(*perturb), len(perturb or ()), x_id, context_key)
        L.mock_reset()
        if graph:
            how = c_int(-1)
            rc = L.wan_dit_forward_graph(self.ctx, ctypes.byref(a), stream, ctypes.byref(how))
            self.how = how.value
        else:
            rc = L.wan_dit_forward_ex(self.ctx, ctypes.byref(a), stream)
        calls = [L.mock_get(i).contents for i in range(L.mock_count())]
        return rc, [(c.name.decode(), list(c.p), list(c.i), list(c.f)) for c in calls], nbytes


def extents(call):
    """(pointer, bytes) pairs an op touches, from its recorded arguments."""
    name, p, i, _ = call
    if name in ("gemm", "gemm_fp8"):
        M, N, K, lda, ldc, epi = i[:6]
        a_el = 1 if name == "gemm_fp8" else 2
        out = [(p[0], M * lda * a_el), (p[3], (N if epi == 3 else M) * ldc * 2)]
        if p[4]:
            out.append((p[4], M * ldc * 2))
        return out
    if name in ("ln_modulate", "ln_affine", "ln_modulate_amax", "ln_affine_amax"):
        return [(p[0], i[0] * i[1] * 2), (p[1], i[0] * i[1] * 2)]
    if name == "rmsnorm_rope":
        return [(q, i[0] * i[3] * 2) for q in p[:2] if q]
    if name == "rmsnorm_rope_pack":
        return [(p[0], i[0] * i[3] * 2), (p[1], i[0] * i[3] * 2)]
    if name == "attention":
        B, Bk, Lq, Lk, ldv, H = i[:6]
        return [(p[0], B * Lq * H * 256), (p[1], Bk * Lk * H * 256), (p[2], Bk * H * 128 * ldv * 2), (p[3], B * Lq * H * 256)]
    if name == "nag_combine":
        return [(q, i[0] * i[1] * 2) for q in p[:3]]
    if name in ("add", "sub", "axpy"):
        return [(q, i[0] * 2) for q in p[:3]]
    if name == "act":
        return [(p[0], i[0] * 2), (p[1], i[0] * 2)]
    if name == "memset":
        return [(p[0], i[0])]
    if name == "memcpy":
        return [(p[0], i[0]), (p[1], i[0])]
    if name in ("fp8_quantize", "fp8_quantize_pre"):
        return [(p[0], i[0] * 2), (p[1], i[0])]
    if name in ("mx_ln_modulate", "mx_ln_affine"):                       # fp32 rows in, bf16 rows out
        return [(p[0], i[0] * i[1] * 4), (p[1], i[0] * i[1] * 2)]
    if name == "mx_gated_residual":
        return [(p[0], i[0] * i[1] * 4), (p[1], i[0] * i[1] * 2)]
    if name == "mx_patch_embed":
        return [(p[4], i[7] * i[5] * 4)]
    if name == "mx_linear_f32":
        return [(p[0], i[0] * i[2] * 4), (p[3], i[0] * i[1] * 4)]
    if name == "mx_sinusoid":
        return [(p[0], i[0] * 4)]
    if name == "mx_head":
        return [(p[0], i[0] * i[1] * 4), (p[5], i[0] * i[1] * 4), (p[6], i[0] * i[3] * 4)]
    return []


def in_ws(ptr, nbytes):
    return WS <= ptr < WS + nbytes


def test_plain_forward_launch_order_weights_and_workspace_bounds(mock):
    m = Model(mock)
    c, (F, H, W) = m.cfg, (2, 8, 8)
    Ll = F * (H // 2) * (W // 2)
    rc, calls, nbytes = m.forward(S=2)
    assert rc == 0, mock.wan_last_error()
    # (a) nothing the forward reads or writes inside the workspace leaves it
    touched = 0
    for cl in calls:
        for ptr, n in extents(cl):
            if in_ws(ptr, nbytes):
                assert ptr + n <= WS + nbytes, (cl[0], hex(ptr - WS), n, nbytes)
                touched += 1
    assert touched > 60
    # (b) per layer: the launch order of WanAttentionBlock.forward (model.py:631-711)
    names = [cl[0] for cl in calls]
    first_block = names.index("ln_modulate")
    per_layer = ["ln_modulate", "gemm", "gemm", "gemm", "gemm", "rmsnorm_rope", "attention", "gemm",            # v x2 (V^T per stream), q, k, norms+RoPE, o
                 "ln_affine", "gemm", "rmsnorm_rope", "gemm", "rmsnorm_rope", "gemm", "gemm", "attention", "gemm",   # cross: q, ck, cv x2, o
                 "ln_modulate", "gemm", "gemm"]                                                                # ffn
    body = names[first_block:first_block + len(per_layer) * c.num_layers]
    assert body == per_layer * c.num_layers
    assert names[first_block + len(per_layer) * c.num_layers:] == ["head", "head"]
    # (c) every Linear of block 1 multiplies by the weight registered under its name, with its bias
    blk = calls[first_block + len(per_layer):first_block + 2 * len(per_layer)]
    gemms = [cl for cl in blk if cl[0] == "gemm"]
    want = ["self_attn.v", "self_attn.v", "self_attn.q", "self_attn.k", "self_attn.o", "cross_attn.q", "cross_attn.k", "cross_attn.v",
            "cross_attn.v", "cross_attn.o", "ffn.0", "ffn.2"]
    assert [g[1][1] for g in gemms] == [m.addr[f"blocks.1.{w}.weight"] for w in want]
    assert [g[1][2] for g in gemms] == [m.addr[f"blocks.1.{w}.bias"] for w in want]
    # shapes: token GEMMs run on both streams' rows at once, the V^T form per stream, text K on 2 x 512 rows
    assert [(g[2][0], g[2][1], g[2][2], g[2][5]) for g in gemms] == [
        (Ll, c.dim, c.dim, 3), (Ll, c.dim, c.dim, 3), (2 * Ll, c.dim, c.dim, 0), (2 * Ll, c.dim, c.dim, 0), (2 * Ll, c.dim, c.dim, 2),
        (2 * Ll, c.dim, c.dim, 0), (2 * TL, c.dim, c.dim, 0), (TL, c.dim, c.dim, 3), (TL, c.dim, c.dim, 3), (2 * Ll, c.dim, c.dim, 2),
        (2 * Ll, c.ffn_dim, c.dim, 1), (2 * Ll, c.dim, c.ffn_dim, 2)]
    att = [cl for cl in blk if cl[0] == "attention"]
    assert att[0][2][:6] == [2, 2, Ll, Ll, (Ll + 63) // 64 * 64, c.num_heads] and att[1][2][:6] == [2, 2, Ll, TL, TL, c.num_heads]
    assert att[0][1][4] != 0 and att[1][1][4] == att[0][1][4]           # the K pre-pass scratch: self-attention, and since round 4 cross-attention too (the persistent bounded walk over 512 keys)


def test_mixed_precision_plan_launch_order_pointers_and_refusals(mock):
    """A context whose time_projection.1.weight is registered as fp32 runs the reference's mixed-precision plan (model.py:1330-1371, :1545):
    the fp32 kernels of csrc/mixed_ops.hip in WanAttentionBlock.forward's order, every Linear that ended in a fused residual epilogue writing
    bf16 to xm with a separate fp32 pass behind it, the residual stream as fp32 rows (stream s at x + s * L * d * 4 bytes), e0 per frame and
    stream under per-frame timesteps, a token-major head + unpatchify; nothing of the bf16 plan's row kernels; step skipping refused."""
    m = Model(mock, mixed=True)
    c, (F, H, W) = m.cfg, (2, 8, 8)
    Ll = F * (H // 2) * (W // 2)
    plain_bytes = mock.wan_dit_workspace_bytes(Model(mock).ctx, 2, F, H, W, 1)
    rc, calls, nbytes = m.forward(S=2)
    assert rc == 0, mock.wan_last_error()
    assert nbytes > plain_bytes + 2 * Ll * c.dim * 2                        # the fp32 stream and the plan's fp32 scratch
    touched = 0
    for cl in calls:
        for ptr, n in extents(cl):
            if in_ws(ptr, nbytes):
                assert ptr + n <= WS + nbytes, (cl[0], hex(ptr - WS), n, nbytes)
                touched += 1
    assert touched > 80
    names = [cl[0] for cl in calls]
    assert not {"ln_modulate", "ln_affine", "head", "patch_embed", "gemv", "sinusoid", "act"} & set(names)
    assert names[:names.index("mx_ln_modulate")].count("mx_patch_embed") == 2
    i0 = names.index("mx_sinusoid")
    assert names[i0:i0 + 4] == ["mx_sinusoid", "mx_linear_f32", "mx_linear_f32", "mx_linear_f32"]
    lins = calls[i0 + 1:i0 + 4]
    assert [(l[2][0], l[2][1], l[2][2], l[2][3]) for l in lins] == [(1, c.dim, c.freq_dim, 0), (1, c.dim, c.dim, 1), (1, 6 * c.dim, c.dim, 1)]
    assert [l[1][1] for l in lins] == [m.addr["time_embedding.0.weight"], m.addr["time_embedding.2.weight"], m.addr["time_projection.1.weight"]]
    assert lins[0][1][3] == lins[1][1][0] and lins[1][1][3] == lins[2][1][0]          # sinusoid -> hidden -> e -> e0, chained
    first_block = names.index("mx_ln_modulate")
    # (round 5: the three Linears that feed the fp32 stream carry the update in their epilogue -- wan_gemm_bf16_res32 -- instead of a
    # Linear into xm followed by a separate wan_mx_gated_residual pass; xm stays the buffer of the shape fall-back)
    per_layer = ["mx_ln_modulate", "gemm", "gemm", "gemm", "gemm", "rmsnorm_rope", "attention", "gemm_res32",
                 "mx_ln_affine", "gemm", "rmsnorm_rope", "gemm", "rmsnorm_rope", "gemm", "gemm", "attention", "gemm_res32",
                 "mx_ln_modulate", "gemm", "gemm_res32"]
    body = names[first_block:first_block + len(per_layer) * c.num_layers]
    assert body == per_layer * c.num_layers
    assert names[first_block + len(per_layer) * c.num_layers:] == ["mx_head", "unpatchify", "mx_head", "unpatchify"]
    blk = calls[first_block + len(per_layer):first_block + 2 * len(per_layer)]
    x32 = blk[0][1][0]
    xm = blk[0][1][1]
    e0 = blk[0][1][3]
    assert x32 == WS                                                        # the residual stream is the first region of the workspace
    gemms = [cl for cl in blk if cl[0] == "gemm"]
    assert all(g[2][5] in (0, 1, 3) for g in gemms)                         # no bf16 fused residual epilogue (2) in this plan
    res = [cl for cl in blk if cl[0] == "gemm_res32"]
    assert [r[2][5] for r in res] == [2, -1, 5] and all(r[1][3] == x32 and r[1][4] == xm for r in res)     # gates; x updated in place, xm = fall-back buffer
    assert [(r[2][0], r[2][1], r[2][2]) for r in res] == [(2 * Ll, c.dim, c.dim), (2 * Ll, c.dim, c.dim), (2 * Ll, c.dim, c.ffn_dim)]
    assert [r[1][1] for r in res] == [m.addr["blocks.1.self_attn.o.weight"], m.addr["blocks.1.cross_attn.o.weight"], m.addr["blocks.1.ffn.2.weight"]]
    assert res[0][1][5] == m.addr["blocks.1.modulation"] and res[0][1][6] == e0 and res[1][1][5] == 0 and res[2][1][6] == e0
    assert not [cl for cl in calls if cl[0] == "mx_gated_residual"]
    aff = [cl for cl in blk if cl[0] == "mx_ln_affine"][0]
    assert aff[1][0] == x32 and aff[1][2] == m.addr["blocks.1.norm3.weight"] and aff[1][3] == m.addr["blocks.1.norm3.bias"]
    mods = [cl for cl in blk if cl[0] == "mx_ln_modulate"]
    assert [(q[2][3], q[2][4]) for q in mods] == [(0, 1), (3, 4)] and all(q[2][0] == 2 * Ll and q[2][5] == 2 * Ll for q in mods)
    heads = [cl for cl in calls if cl[0] == "mx_head"]
    assert [h[1][0] for h in heads] == [x32, x32 + Ll * c.dim * 4] and all(h[2][0] == Ll and h[2][2] == Ll and h[2][3] == 4 * c.out_dim for h in heads)
    unp = [cl for cl in calls if cl[0] == "unpatchify"]
    assert [u[1][1] for u in unp] == [0x6200_0000_0000, 0x6200_0000_0000 + 0x1_0000_0000] and all(u[1][0] == heads[0][1][6] for u in unp)
    # per-frame timesteps: one e0 row set per frame, replicated per stream; rows_per_batch = tokens per frame
    rc, calls, _ = m.forward(S=2, t_frames=[0.0, 637.0])
    assert rc == 0, mock.wan_last_error()
    names = [cl[0] for cl in calls]
    assert names.count("mx_sinusoid") == 2 and [cl for cl in calls if cl[0] == "mx_linear_f32"][0][2][0] == 2
    assert any(cl[0] == "memcpy" and cl[2][0] == 2 * 6 * c.dim * 4 for cl in calls)
    assert all(cl[2][5] == Ll // F for cl in calls if cl[0] == "mx_ln_modulate") and all(cl[2][2] == Ll // F for cl in calls if cl[0] == "mx_head")
    # one stream (a CFG-parallel half), sequence parallelism: the head hands over token-major rows, nothing unpatchifies
    sp = SpInfo(1, 2, Ll // 2, Ll // 2, GATHER_FN(lambda *a: 0), GATHER_WAIT_FN(lambda *a: 0), None, 0, GATHER_FN(lambda *a: 0), GATHER_WAIT_FN(lambda *a: 0))
    rc, calls, _ = m.forward(S=1, sp=sp)
    assert rc == 0, mock.wan_last_error()
    names = [cl[0] for cl in calls]
    assert "unpatchify" not in names and names[-1] == "mx_head" and calls[-1][1][6] == 0x6200_0000_0000
    pe = [cl for cl in calls if cl[0] == "mx_patch_embed"][0]
    assert pe[2][6:8] == [Ll // 2, Ll // 2]                                  # this rank's token range
    # step-skipping caches in the mixed plan (round 5): the residual buffers hold FP32 rows like the stream (model.py:2044-2062 subtracts
    # two fp32 tensors) -- parked with an fp32-sized copy, re-applied / formed by fp32 linear combinations, never the bf16 add / sub
    res = [0x6400_0000_0000, 0x6410_0000_0000]
    sn = Ll * c.dim
    rc, calls, _ = m.forward(S=2, should_calc=[1, 0], residual=res)
    assert rc == 0, mock.wan_last_error()
    names = [cl[0] for cl in calls]
    assert "add" not in names and "sub" not in names
    park = [cl for cl in calls if cl[0] == "memcpy" and cl[1][0] == res[0]]
    assert len(park) == 1 and park[0][1][1] == x32 and park[0][2][0] == sn * 4                 # computing stream: x_before parked as fp32
    lc = [cl for cl in calls if cl[0] == "lincomb"]
    assert len(lc) == 2 and all(cl[2][:2] == [sn, 2] for cl in lc)
    x1 = x32 + sn * 4
    assert lc[0][1] [:3] == [x1, x1, res[1]] and lc[0][3][:2] == [1.0, 1.0]                     # skipped stream: x += stored residual
    assert lc[1][1][:3] == [res[0], x32, res[0]] and lc[1][3][:2] == [1.0, -1.0]               # computing stream: residual = x - x_before
    assert names.index("lincomb") < names.index("mx_ln_modulate") < len(names) - 1 - names[::-1].index("lincomb")
    # an inconsistent registration (fp32 projection, bf16 norm3) is an error, not a silent mix of plans
    bad = Model(mock, mixed=True)
    assert mock.wan_dit_set_weight(bad.ctx, b"blocks.0.norm3.weight", c_void_p(bad.addr["blocks.0.norm3.weight"]), 0, c.dim) == 0
    rc, _, _ = bad.forward(S=1)
    assert rc != 0 and b"norm3.weight" in mock.wan_last_error()


def test_step_skipping_touches_only_the_computing_stream(mock):
    m = Model(mock)
    c, (F, H, W) = m.cfg, (2, 8, 8)
    Ll = F * 16
    res = [0x6400_0000_0000, 0x6400_1000_0000]
    rc, calls, nbytes = m.forward(S=2, should_calc=[1, 0], residual=res)
    assert rc == 0, mock.wan_last_error()
    x1 = WS + Ll * c.dim * 2                                             # stream 1's token stream inside the workspace
    adds = [cl for cl in calls if cl[0] == "add"]
    assert len(adds) == 1 and adds[0][1][:3] == [x1, res[1], x1]         # skipped stream: x += stored residual (model.py:1977-1990)
    cp = [cl for cl in calls if cl[0] == "memcpy" and cl[1][0] == res[0]]
    assert len(cp) == 1 and cp[0][1][1] == WS                            # computing stream: x parked in its residual buffer ...
    subs = [cl for cl in calls if cl[0] == "sub"]
    assert len(subs) == 1 and subs[0][1][:3] == [WS, res[0], res[0]]     # ... and replaced by x_after - x_before at the end
    for cl in calls:                                                     # no block op reads or writes stream 1's rows
        if cl[0] in ("ln_modulate", "ln_affine"):
            assert cl[2][0] == Ll and cl[1][0] == WS
    assert sum(1 for cl in calls if cl[0] == "attention") == 2 * c.num_layers
    assert [cl[2][0] for cl in calls if cl[0] == "attention"] == [1] * (2 * c.num_layers)
    assert [cl[0] for cl in calls][-2:] == ["head", "head"]             # both streams still go through the head
    # a skipped stream without a residual buffer is refused
    rc, _, _ = m.forward(S=2, should_calc=[1, 0], residual=[res[0], None])
    assert rc == 1 and b"residual" in mock.wan_last_error()


def test_nag_branch_pointer_arithmetic(mock):
    """Stream 0 carries (positive ; negative) prompts, stream 1 a plain context (any2video.py:608, :1551)."""
    m = Model(mock)
    c, (F, H, W) = m.cfg, (2, 8, 8)
    Ll, d = F * 16, m.cfg.dim
    rc, calls, nbytes = m.forward(S=2, nag=(11.0, 2.5, 0.25), ctx_batches=[2, 1])
    assert rc == 0, mock.wan_last_error()
    for cl in calls:
        for ptr, n in extents(cl):
            if in_ws(ptr, nbytes):
                assert ptr + n <= WS + nbytes, (cl[0], hex(ptr - WS), n)
    names = [cl[0] for cl in calls]
    assert names.count("nag_combine") == c.num_layers
    # text embedding: one tensor per stream, 1024 rows for the stacked context
    te = [cl for cl in calls if cl[0] == "gemm" and cl[1][1] == m.addr["text_embedding.0.weight"]]
    assert [t_[2][0] for t_ in te] == [2 * TL, TL]
    i0 = names.index("nag_combine")
    seg = calls[i0 - 8:i0 + 5]
    seq = [s[0] for s in seg]
    print(seq)
    assert seq == ["gemm", "rmsnorm_rope",                                                                  # (cross q projection + norm, all streams)
                   "gemm", "rmsnorm_rope", "gemm", "gemm", "attention", "attention", "nag_combine",          # stream 0: K on 1024 rows, 2 x V^T, 2 x attention
                   "gemm", "rmsnorm_rope", "gemm", "attention"]                                              # stream 1: K on 512 rows, V^T, attention
    k0 = next(s for s in seg if s[0] == "gemm" and s[2][0] == 2 * TL)
    a_pos, a_neg = [s for s in seg if s[0] == "attention"][:2]
    nc = calls[i0]
    q0 = a_pos[1][0]
    assert a_neg[1][0] == q0                                             # the same q rows against both prompts
    assert a_pos[1][1] == k0[1][3] and a_neg[1][1] == k0[1][3] + TL * d * 2          # K of the negative prompt: 512 rows further
    assert a_neg[1][2] == a_pos[1][2] + TL * d * 2                       # its V^T image: d x 512 elements further
    assert a_pos[2][:4] == [1, 1, Ll, TL] and a_neg[2][:4] == [1, 1, Ll, TL]
    assert nc[1][:3] == [a_pos[1][3], a_neg[1][3], q0] and nc[2][:2] == [Ll, d] and nc[3][:3] == [11.0, 2.5, 0.25]
    assert len({a_pos[1][3], a_neg[1][3], q0}) == 3                      # three different buffers: xm, the idle FFN buffer, q
    # stream 1 (plain context): one attention on its own q rows / its own K behind stream 0's 1024 context rows
    a1 = [s for s in calls[i0:] if s[0] == "attention"][0]
    assert a1[1][0] == q0 + Ll * d * 2 and a1[1][1] == k0[1][3] + 2 * TL * d * 2 and a1[1][3] == a1[1][0]
    # a batch-2 context without NAG parameters is refused (model.py:260)
    rc, _, _ = m.forward(S=2, nag=(1.0, 2.5, 0.25), ctx_batches=[2, 1])
    assert rc == 1 and b"context_batches" in mock.wan_last_error()


@pytest.mark.parametrize("S", [2, 1])
def test_sequence_parallel_call_order(mock, S):
    """K projection -> K gather begins -> V^T -> its gather -> Q -> attention on the own segment -> both waits -> the other
    segments, per layer; shards see their token range.  S = 2: both CFG streams on every rank (pure sequence
    parallelism); S = 1: the single-stream forward of a CFG-parallel half (sp.CfgParallel) -- the gathers carry ONE stream."""
    m = Model(mock)
    c, (F, H, W) = m.cfg, (2, 8, 8)
    L_ = F * 16
    events = []

    def begin(user, which, send, recv, nbytes, stream):
        events.append(("begin", which, send, recv, nbytes, mock.mock_count()))
        return 0

    def wait(user, which, stream):
        events.append(("wait", which, mock.mock_count()))
        return 0
    cb, cw = GATHER_FN(begin), GATHER_WAIT_FN(wait)
    sp = SpInfo(1, 2, L_ // 2, L_ // 2, cb, cw, None)
    rc, calls, nbytes = m.forward(S=S, sp=sp)
    assert rc == 0, mock.wan_last_error()
    Ll, d = L_ // 2, c.dim
    assert len(events) == 4 * c.num_layers
    for layer in range(c.num_layers):
        b0, b1, w0, w1 = events[4 * layer:4 * layer + 4]
        assert (b0[0], b0[1], b1[0], b1[1], w0[:2], w1[:2]) == ("begin", 0, "begin", 1, ("wait", 0), ("wait", 1))
        assert b0[4] == S * Ll * d * 2 and b1[4] == S * d * ((Ll + 63) // 64 * 64) * 2        # K rows / V^T images of the rank's S streams
        between = [cl[0] for cl in calls[b0[5]:b1[5]]]
        assert between == ["gemm"] * S                                    # the V^T projections (one per stream) run under the K gather
        local = [cl[0] for cl in calls[b1[5]:w0[2]]]
        assert local == ["gemm", "rmsnorm_rope", "attention_sp_local"]   # Q projection, norm + RoPE, own segment -- gathers in flight
        assert calls[w1[2]][0] == "attention_sp_remote" and calls[w1[2]][2][5:9] == [2, S * Ll * d, S * d * ((Ll + 63) // 64 * 64), 1]
    pe = [cl for cl in calls if cl[0] == "patch_embed"]
    assert all(cl[2][7:9] == [Ll, Ll] for cl in pe)                      # rank 1 of 2 embeds tokens [Ll, 2 Ll)
    rr = [cl for cl in calls if cl[0] == "rmsnorm_rope" and cl[1][4] != 0]
    assert all(cl[2][2] == Ll for cl in rr)                              # RoPE positions start at the shard's first token
    hd = [cl for cl in calls if cl[0] == "head"]
    assert all(cl[2][7] == 1 for cl in hd)                               # token-major output for the gather


@pytest.mark.parametrize("S", [2, 1])
def test_ulysses_call_order_layouts_and_buffer_lifetimes(mock, S):
    """WAN_SP_ULYSSES (round 4): per layer K projection -> norm + RoPE -> head-group-major re-pack -> k all-to-all begins; V^T
    projections (+ the [S][world] -> [world][S] block swap when S = 2) -> v^T all-to-all; Q projection -> norm -> re-pack -> q
    all-to-all; the three waits; ONE attention launch over world x S query batches of L / world rows against S K / V^T batches in
    `world` segments, H / world heads; o all-to-all, its wait, the re-pack back to token-major, the output projection.  Chunk sizes
    are what one peer receives; every buffer lies inside the workspace; a buffer is never rewritten while its exchange is in flight."""
    from wan2gp_amd.lib import SP_ULYSSES
    m = Model(mock)
    c, (F, H, W) = m.cfg, (2, 8, 8)
    L_, world = F * 16, 2
    Ll, d, nh = L_ // world, c.dim, c.num_heads
    Hn, Wd, Lp, rows = nh // world, nh // world * 128, (Ll + 63) // 64 * 64, S * Ll
    events = []

    def begin(user, which, send, recv, nbytes, stream):
        events.append(("begin", which, send, recv, nbytes, mock.mock_count()))
        return 0

    def wait(user, which, stream):
        events.append(("wait", which, mock.mock_count()))
        return 0
    cb, cw = GATHER_FN(begin), GATHER_WAIT_FN(wait)
    never = GATHER_FN(lambda *a: pytest.fail("the all-gather hooks must not be used in Ulysses mode"))
    sp = SpInfo(1, world, Ll, Ll, never, GATHER_WAIT_FN(lambda *a: 1), None, SP_ULYSSES, cb, cw)
    rc, calls, nbytes = m.forward(S=S, sp=sp)
    assert rc == 0, mock.wan_last_error()
    assert len(events) == 8 * c.num_layers
    names = [cl[0] for cl in calls]
    assert "attention_sp_local" not in names and "attention_sp_remote" not in names
    for layer in range(c.num_layers):
        ev = events[8 * layer:8 * layer + 8]
        assert [(e[0], e[1]) for e in ev] == [("begin", 0), ("begin", 1), ("begin", 2), ("wait", 0), ("wait", 1), ("wait", 2), ("begin", 3), ("wait", 3)]
        bk, bv, bq, w0, w1, w2, bo, wo = ev
        assert bk[4] == bq[4] == bo[4] == rows * Wd * 2 and bv[4] == S * Wd * Lp * 2              # bytes per PEER
        # in front of the k exchange: K projection, norm + RoPE at the shard's positions, the re-pack into the send buffer
        pre = calls[bk[5] - 2:bk[5]]
        assert [cl[0] for cl in pre] == ["gemm", "rmsnorm_rope_pack"] and pre[1][2][2] == Ll       # (round 6: the norm kernel's stores carry the re-pack)
        assert pre[1][2][:7] == [rows, Ll, Ll, d, world, Hn, 1] and pre[1][1][0] == pre[0][1][3] and pre[1][1][1] == bk[2]   # [rows][world][W] -> [world][rows][W] = what is sent
        # under the k exchange: the V^T projections (+ the block swap for S = 2); the v^T exchange sends what they left
        under_k = [cl[0] for cl in calls[bk[5]:bv[5]]]
        assert under_k == ["gemm"] * S + (["permute16"] if S > 1 else [])
        if S > 1:
            sw = calls[bv[5] - 1]
            assert sw[2][:3] == [S, world, Wd * Lp * 2] and sw[1][1] == bv[2]
        else:
            assert calls[bv[5] - 1][1][3] == bv[2]                                                    # S = 1: sent straight from the epilogue's image
        under_v = [cl[0] for cl in calls[bv[5]:bq[5]]]
        assert under_v == ["gemm", "rmsnorm_rope_pack"]                                               # Q projection + norm-with-re-pack under the v^T exchange
        assert calls[bq[5] - 1][1][1] == bq[2] and calls[bq[5] - 1][2][4:7] == [world, Hn, 1]
        assert w0[2] == w1[2] == w2[2] == bq[5]                                                       # nothing between the q exchange and the waits
        att = calls[w2[2]]
        assert att[0] == "attention" and att[2][:10] == [world * S, S, Ll, Ll, Lp, Hn, world, rows * Wd, S * Wd * Lp, 1]
        assert (att[1][0], att[1][1], att[1][2]) == (bq[3], bk[3], bv[3])                            # q / k / v^T as RECEIVED
        assert att[1][3] == bo[2] == bk[2]                                                           # o written over the dead k send buffer = the o exchange's send
        assert bo[3] == bq[2]                                                                        # ... and received over the dead q send buffer
        back = calls[wo[2]]
        assert back[0] == "permute16" and back[2][:3] == [world, rows, Wd * 2] and back[1][0] == bo[3]
        proj = calls[wo[2] + 1]
        assert proj[0] == "gemm" and proj[1][0] == back[1][1] and proj[2][5] == 2                     # the output projection (gated residual) reads the re-packed rows
        # the six exchange buffers are distinct and inside the workspace
        bufs = [(bk[2], rows * d * 2), (bk[3], rows * d * 2), (bq[2], rows * d * 2), (bq[3], rows * d * 2), (bv[3], S * d * Lp * 2)]
        if S > 1:
            bufs.append((bv[2], S * d * Lp * 2))
        for a, na in bufs:
            assert in_ws(a, nbytes) and in_ws(a + na - 1, nbytes)
        for i, (a, na) in enumerate(bufs):
            for b, nb in bufs[i + 1:]:
                assert a + na <= b or b + nb <= a, (hex(a), na, hex(b), nb)
    # heads the world does not divide: refused before anything is enqueued
    sp3 = SpInfo(0, 4, 0, L_ // 4, never, GATHER_WAIT_FN(lambda *a: 1), None, SP_ULYSSES, cb, cw)
    rc, calls, _ = m.forward(S=S, sp=sp3)
    assert rc == 1 and b"head count" in mock.wan_last_error() and not calls
    # a failing exchange stops the forward at that block
    fail = GATHER_FN(lambda user, which, *a: 1 if which == 2 else 0)
    rc, calls, _ = m.forward(S=S, sp=SpInfo(1, world, Ll, Ll, never, GATHER_WAIT_FN(lambda *a: 1), None, SP_ULYSSES, fail, cw))
    assert rc == 3 and b"all-to-all 2" in mock.wan_last_error() and "attention" not in [cl[0] for cl in calls if cl[0] == "attention" and cl[2][6] == world]


@pytest.mark.parametrize("S,world,chunks", [(2, 2, 2), (1, 2, 2), (2, 1, 3)], ids=["S2_heads_1+1", "S1_heads_1+1", "S2_one_rank_group_is_plain"])
def test_ulysses_chunked_exchange_order_offsets_and_overlap(mock, S, world, chunks):
    """wan_sp_info.a2a_chunks = C > 1 (round 5): every tensor is packed chunk-major and travels per head chunk -- slots k_c = c, v_c = C + c,
    q_c = 2 C + c, o_c = 3 C + c.  Chunk 0's k leaves behind the K pack (under the V projections), its v^T behind the V pack (under the Q
    projection), its q behind the Q pack, THEN the other chunks' k / v^T / q; chunk c's launch sits between the waits for ITS three
    tensors and the begin of ITS o chunk -- the later chunks' exchanges and the earlier o chunks in flight meanwhile -- and sees the
    round-4 layout with H = the chunk's heads; every o chunk is waited for and re-packed before the output projection.  A group of
    one rank ignores the mode altogether."""
    from wan2gp_amd.lib import SP_ULYSSES
    m = Model(mock, name="small")                                            # 4 heads
    c, (F, H, W) = m.cfg, (2, 8, 8)
    L_ = F * 16
    Ll, d, nh = L_ // world, c.dim, c.num_heads
    Hn, Wd, Lp, rows = nh // world, nh // world * 128, (Ll + 63) // 64 * 64, S * Ll
    events = []

    def begin(user, which, send, recv, nbytes, stream):
        events.append(("begin", which, send, recv, nbytes, mock.mock_count()))
        return 0

    def wait(user, which, stream):
        events.append(("wait", which, mock.mock_count()))
        return 0
    cb, cw = GATHER_FN(begin), GATHER_WAIT_FN(wait)
    sp = SpInfo(world - 1, world, (world - 1) * Ll, Ll, cb, cw, None, SP_ULYSSES, cb, cw, chunks)
    rc, calls, nbytes = m.forward(S=S, sp=sp)
    assert rc == 0, mock.wan_last_error()
    if world == 1:
        assert not events and "permute16_ex" not in [cl[0] for cl in calls]
        return
    C = min(chunks, Hn)
    h0 = [cch * Hn // C for cch in range(C + 1)]
    per_layer = 2 * 4 * C
    assert len(events) == per_layer * c.num_layers
    for layer in range(c.num_layers):
        ev = events[per_layer * layer:per_layer * (layer + 1)]
        want = [("begin", 0), ("begin", C), ("begin", 2 * C)]
        for k in range(1, C):
            want += [("begin", k), ("begin", C + k), ("begin", 2 * C + k)]
        for k in range(C):
            want += [("wait", k), ("wait", C + k), ("wait", 2 * C + k), ("begin", 3 * C + k)]
        want += [("wait", 3 * C + k) for k in range(C)]
        assert [(e[0], e[1]) for e in ev] == want
        E = {(e[0], e[1]): e for e in ev}
        bk0, bv0, bq0 = E[("begin", 0)], E[("begin", C)], E[("begin", 2 * C)]
        # what sits between the first three begins: the V projections + C v^T packs under k_0; the Q projection, its norm and C q packs under v_0
        # (round 6: the K and Q norm kernels write the chunk-major send layout themselves -- one rmsnorm_rope_pack each, no re-pack passes)
        assert [cl[0] for cl in calls[bk0[5] - 2:bk0[5]]] == ["gemm", "rmsnorm_rope_pack"]
        assert [cl[0] for cl in calls[bk0[5]:bv0[5]]] == ["gemm"] * S + ["permute16_ex"] * C
        assert [cl[0] for cl in calls[bv0[5]:bq0[5]]] == ["gemm", "rmsnorm_rope_pack"]
        pack_k, pack_q = calls[bk0[5] - 1], calls[bq0[5] - 1]
        assert pack_k[2][:7] == [rows, Ll, (world - 1) * Ll, d, world, Hn, C] and pack_k[1][1] == bk0[2] and pack_k[3][0] == 1.0
        assert pack_q[2][:7] == [rows, Ll, (world - 1) * Ll, d, world, Hn, C] and pack_q[1][1] == bq0[2] and pack_q[3][0] != 1.0     # the softmax scale folded into q
        packs_v = calls[bv0[5] - C:bv0[5]]
        att_i = [i for i in range(bq0[5], len(calls)) if calls[i][0] == "attention"][:C]
        for k in range(C):
            Hc = h0[k + 1] - h0[k]
            Wc, o0 = Hc * 128, h0[k] * 128
            bk, bv, bq, bo = E[("begin", k)], E[("begin", C + k)], E[("begin", 2 * C + k)], E[("begin", 3 * C + k)]
            # the chunk's [world][rows][Wc] block sits at the chunk's offset of the packed tensor (what the norm kernel's stores address)
            assert bk[2] == bk0[2] + o0 * rows * world * 2 and bq[2] == bq0[2] + o0 * rows * world * 2
            assert packs_v[k][2][:7] == [S, world, Wc * Lp * 2, d * Lp * 2, Wd * Lp * 2, Wc * Lp * 2, S * Wc * Lp * 2]
            assert packs_v[k][1][1] == bv[2] == bv0[2] + o0 * Lp * S * world * 2
            assert bk[4] == bq[4] == bo[4] == rows * Wc * 2 and bv[4] == S * Wc * Lp * 2                       # bytes per PEER: the chunk's share
            assert bk[3] - bk0[3] == bq[3] - bq0[3] == o0 * rows * world * 2 and bv[3] - bv0[3] == o0 * Lp * S * world * 2
            att = calls[att_i[k]]
            assert max(E[("wait", k)][2], E[("wait", C + k)][2], E[("wait", 2 * C + k)][2]) <= att_i[k] < bo[5]   # between ITS waits and ITS o begin
            assert att[2][:10] == [world * S, S, Ll, Ll, Lp, Hc, world, rows * Wc, S * Wc * Lp, 1]               # the chunk's segment strides
            assert (att[1][0], att[1][1], att[1][2]) == (bq[3], bk[3], bv[3])                                  # q / k / v^T chunks as RECEIVED
            assert att[1][3] == bo[2] == bk[2] and bo[3] == bq[2]                                              # o over the dead k send chunk, back over the dead q send chunk
            if k + 1 < C:
                assert E[("begin", 2 * C + k + 1)][5] <= att_i[k] and E[("wait", 2 * C + k + 1)][2] > att_i[k]   # the next chunk's tensors are in flight
                assert bo[5] <= att_i[k + 1]                                                                   # ... and this o chunk during the next launch
        last_wait = E[("wait", 4 * C - 1)]
        back, proj = calls[last_wait[2]], calls[last_wait[2] + 1]
        assert back[0] == "permute16_ex" and proj[0] == "gemm" and proj[2][5] == 2
        ups = [cl for cl in calls[E[("wait", 3 * C)][2]:last_wait[2] + 1] if cl[0] == "permute16_ex"]
        assert len(ups) == C and all(u[1][1] - ups[0][1][1] == h0[k] * 256 for k, u in enumerate(ups)) and ups[0][1][1] == proj[1][0]
        for k, u in enumerate(ups):
            Wc = (h0[k + 1] - h0[k]) * 128
            assert u[2][:7] == [world, rows, Wc * 2, rows * Wc * 2, Wc * 2, Wd * 2, d * 2] and u[1][0] == E[("begin", 2 * C + k)][2]
        for e in ev:
            if e[0] == "begin":
                assert in_ws(e[2], nbytes) and in_ws(e[3] + e[4] * world - 1, nbytes)


def test_sequence_parallel_failing_gather_hook_stops_the_forward_with_an_error(mock):
    """Failure modes of the collective hooks (RCCL error, a Python exception inside the torch.distributed callback -> non-zero
    return): the forward stops at that block with rc 3 and a message that names the gather; nothing after it is enqueued, and a
    later forward on the same context works again (no state is left behind)."""
    m = Model(mock)
    c, (F, H, W) = m.cfg, (2, 8, 8)
    L_ = F * 16
    for fail_at, what in ((("begin", 0, 1), b"K all-gather"), (("begin", 1, 0), b"V^T all-gather"), (("wait", 0, 0), b"all-gather")):
        seen = {"begin": [0, 0], "wait": [0, 0]}

        def begin(user, which, send, recv, nbytes, stream):
            seen["begin"][which] += 1
            return 1 if fail_at[0] == "begin" and fail_at[1] == which and seen["begin"][which] - 1 == fail_at[2] else 0

        def wait(user, which, stream):
            seen["wait"][which] += 1
            return 1 if fail_at[0] == "wait" and fail_at[1] == which and seen["wait"][which] - 1 == fail_at[2] else 0
        cb, cw = GATHER_FN(begin), GATHER_WAIT_FN(wait)
        rc, calls, _ = m.forward(S=2, sp=SpInfo(0, 2, 0, L_ // 2, cb, cw, None))
        assert rc == 3 and what in mock.wan_last_error(), (rc, mock.wan_last_error())
        names = [cl[0] for cl in calls]
        assert "head" not in names and names.count("attention_sp_remote") == (1 if fail_at == ("begin", 0, 1) else 0)
    cb, cw = GATHER_FN(lambda *a: 0), GATHER_WAIT_FN(lambda *a: 0)
    rc, calls, _ = m.forward(S=2, sp=SpInfo(0, 2, 0, L_ // 2, cb, cw, None))
    assert rc == 0 and [cl[0] for cl in calls].count("attention_sp_remote") == c.num_layers


def test_fp8_checkpoint_quantises_once_per_shared_input(mock):
    """q / k / v share one activation quantisation per stream (the reference quantises per tensor = per stream), the Linears go
    through the fp8 GEMM with their scales.  Round 5: four of a block's six quantisations run WITHOUT their abs-max pass -- the
    LayerNorms in front of q / k / v, cross q and ffn.0 accumulate the abs-max of what they write into word 1 of the streams' slots
    (zeroed right in front of them), ffn.0's GELU epilogue into word 2 for ffn.2 -- `fp8_quantize_pre`; the attention outputs in front
    of the two o projections and the text context keep the two-pass form."""
    m = Model(mock, fp8=True)
    c = m.cfg
    rc, calls, nbytes = m.forward(S=2)
    assert rc == 0, mock.wan_last_error()
    names = [cl[0] for cl in calls]
    assert "ln_modulate" not in names and "ln_affine" not in names
    i0 = names.index("ln_modulate_amax")
    i1 = names.index("rmsnorm_rope", i0)
    head = names[i0:i1]
    assert head.count("fp8_quantize_pre") == 2 and head.count("fp8_quantize") == 0 and head.count("gemm_fp8") == 2 + 2 + 2       # 2 streams: V^T, q, k
    Ll = 2 * 16
    per_layer = names[i0:names.index("ln_modulate_amax", names.index("ln_modulate_amax", i0 + 1) + 1)] if c.num_layers > 1 else names[i0:]
    assert per_layer.count("ln_modulate_amax") == 2 and per_layer.count("ln_affine_amax") == 1
    assert per_layer.count("fp8_quantize_pre") == 2 + 2 + 2 + 2          # per stream: xm after norm1, norm3, norm2 and the GELU output
    assert per_layer.count("fp8_quantize") == 2 + 2 + 2                  # per stream: the two attention outputs; the text context (cross k / v)
    lay = calls[i0:i0 + len(per_layer)]
    lns = [cl for cl in lay if cl[0] in ("ln_modulate_amax", "ln_affine_amax")]
    slots = lns[0][1][4]
    assert all(cl[1][4] == slots and cl[2][6 if cl[0] == "ln_modulate_amax" else 2] == Ll for cl in lns) and in_ws(slots, nbytes)   # rows_per_slot = tokens of a stream
    for k_, cl in enumerate(lay):
        if cl[0] in ("ln_modulate_amax", "ln_affine_amax"):
            z = lay[k_ - 1]
            assert z[0] == "memset" and z[1][0] == slots and z[2][:2] == [2 * 64 * 4, 0]          # the slots' words zeroed right in front of the producer
    pre = [cl for cl in lay if cl[0] == "fp8_quantize_pre"]
    assert [cl[2][1] for cl in pre] == [1, 1, 1, 1, 1, 1, 2, 2]                                   # word 1 x 3 producers x 2 streams, then word 2 (GELU output)
    assert [cl[1][2] - slots for cl in pre] == [0, 256] * 4                                         # stream s reads slot s
    gelu = [cl for cl in lay if cl[0] == "gemm_fp8" and cl[2][5] == 1]
    assert len(gelu) == 2 and [g[1][6] - slots for g in gelu] == [8, 256 + 8]                        # ffn.0 leaves max |h| in word 2 of its stream's slot
    for cl in calls:
        for ptr, n in extents(cl):
            if in_ws(ptr, nbytes):
                assert ptr + n <= WS + nbytes, (cl[0], hex(ptr - WS), n)
    assert "gemm" in names                                               # time / text embeddings stay bf16
    # the mixed-precision plan with an fp8 checkpoint keeps the two-pass quantisation (its LayerNorms are the fp32 kernels)
    mm = Model(mock, fp8=True, mixed=True)
    rc, calls, _ = mm.forward(S=2)
    assert rc == 0, mock.wan_last_error()
    assert "fp8_quantize_pre" not in [cl[0] for cl in calls] and "ln_modulate_amax" not in [cl[0] for cl in calls]


def test_interrupt_poll_aborts_between_blocks(mock):
    m = Model(mock)
    seen = []

    def poll(user, i):
        seen.append(i)
        return 1 if i == 1 else 0
    rc, calls, _ = m.forward(S=1, poll=POLL_FN(poll))
    assert rc == 100 and seen == [0, 1]                                  # WAN_ABORTED before block 1 (model.py:1995-1998)
    assert "head" not in [cl[0] for cl in calls]


def test_skip_layer_guidance_runs_listed_blocks_for_the_first_stream_only(mock):
    """perturbation_layers (any2video.py:1502, model.py:2025-2028): a listed block runs on stream 0 of the call that carries the
    conditional stream (x_id 0) and nowhere else; unlisted blocks are untouched.  The listed block on stream 0 is exactly the
    single-stream run step skipping already uses (GPU-tested); the oracle's restatement is pinned to the reference's own forward
    (tests/test_nag_oracle_vs_golden.py)."""
    m = Model(mock)
    c = m.cfg
    Ll, d = 2 * 16, m.cfg.dim
    rc, plain, _ = m.forward(S=2)
    rc, calls, nbytes = m.forward(S=2, perturb=[1])
    assert rc == 0, mock.wan_last_error()
    names = [cl[0] for cl in calls]
    b0, per = names.index("ln_modulate"), 20
    assert [cl[0] for cl in plain][:b0 + per] == names[:b0 + per]                      # block 0 (unlisted): as in the plain forward
    assert [(cl[1], cl[2]) for cl in plain[:b0 + per]] == [(cl[1], cl[2]) for cl in calls[:b0 + per]]
    blk1 = calls[b0 + per:-2]
    # block 1 on stream 0 alone: one V^T GEMM, every token op on Ll rows at stream 0's addresses, attention with batch 1
    assert [cl[0] for cl in blk1] == ["ln_modulate", "gemm", "gemm", "gemm", "rmsnorm_rope", "attention", "gemm", "ln_affine", "gemm", "rmsnorm_rope",
                                      "gemm", "rmsnorm_rope", "gemm", "attention", "gemm", "ln_modulate", "gemm", "gemm"]
    assert all(cl[2][0] == Ll and cl[1][0] == WS for cl in blk1 if cl[0] in ("ln_modulate", "ln_affine"))
    assert [cl[2][0] for cl in blk1 if cl[0] == "attention"] == [1, 1]
    x1 = WS + Ll * d * 2
    for cl in blk1:                                                                    # nothing of block 1 writes stream 1's token rows
        if cl[0] == "gemm" and cl[2][5] == 2:
            assert cl[1][3] == WS and cl[2][0] == Ll
        assert x1 not in (cl[1][0], cl[1][3]) or cl[0] == "head"
    assert names[-2:] == ["head", "head"]
    # the unconditional call of a non-joint pass (x_id 1): the listed block is skipped altogether
    rc, solo, _ = m.forward(S=1, perturb=[1], x_id=1)
    assert rc == 0 and [cl[0] for cl in solo].count("ln_modulate") == 2 * (c.num_layers - 1)
    # ... and so it is when stream 0 is skipped by the step-skipping cache while stream 1 computes
    rc, part, _ = m.forward(S=2, perturb=[1], should_calc=[0, 1], residual=[0x6400_0000_0000, 0x6400_1000_0000])
    assert rc == 0 and [cl[0] for cl in part].count("ln_modulate") == 2 * (c.num_layers - 1)
    # every block listed: only stream 0 moves
    rc, allp, _ = m.forward(S=2, perturb=[0, 1])
    assert rc == 0 and all(cl[2][0] == 1 for cl in allp if cl[0] == "attention")


def _vace_model(L):
    m = Model.__new__(Model)
    m.L, m.cfg = L, O.make_config("tiny_vace")
    c = m.cfg
    dc = DitConfig(c.dim, c.ffn_dim, c.num_heads, c.num_layers, c.in_dim, c.out_dim, c.text_dim, c.freq_dim, c.text_len, c.eps)
    m.ctx = c_void_p()
    assert L.wan_dit_create(ctypes.byref(dc), ctypes.byref(m.ctx)) == 0
    assert L.wan_dit_set_vace_layers(m.ctx, (c_int * len(c.vace_layers))(*c.vace_layers), len(c.vace_layers)) == 0
    assert L.wan_dit_set_vace_contexts(m.ctx, 2) == 0
    m.addr = {}
    for n, (k, shape) in enumerate(O.param_shapes(c).items()):
        numel = 1
        for s in shape:
            numel *= s
        m.addr[k] = 0x1000_0000_0000 + n * 0x10_0000_0000
        dt = 1 if k.startswith(("patch_embedding.", "head.", "vace_patch_embedding.")) else 0
        assert L.wan_dit_set_weight(m.ctx, k.encode(), c_void_p(m.addr[k]), dt, numel) == 0, (k, L.wan_last_error())
    return m


def _forward_vace(L, m, contexts, scales, S=2, fhw=(2, 8, 8)):
    F, H, W = fhw
    nb = L.wan_dit_workspace_bytes(m.ctx, S, F, H, W, 1)
    X = (c_void_p * S)(*[0x6000_0000_0000 + s * 0x1_0000_0000 for s in range(S)])
    C = (c_void_p * S)(*[0x6100_0000_0000 + s * 0x1_0000_0000 for s in range(S)])
    OUT = (c_void_p * S)(*[0x6200_0000_0000 + s * 0x1_0000_0000 for s in range(S)])
    nv = len(contexts)
    a = DitArgs(S, X, 588.0, C, None, 0x6300_0000_0000, 0x6310_0000_0000, OUT, F, H, W, WS, nb, None, None, None, None, None,
                None, 1.0, None, 0, nv, (c_void_p * nv)(*contexts), (ctypes.c_float * nv)(*scales), 0.0, 0.0, 0.0, None, None, 0, 0)
    L.mock_reset()
    rc = L.wan_dit_forward_ex(m.ctx, ctypes.byref(a), None)
    return rc, [(c.name.decode(), list(c.p), list(c.i), list(c.f)) for c in (L.mock_get(i).contents for i in range(L.mock_count()))], nb


def test_vace_context_blocks_launch_order_and_scales(mock):
    """VaceWanAttentionBlock around the main blocks it is attached to (model.py:617-629, :713-719, :816-828): per active context a
    patch embedding of the context, before_proj (+ x) in front of context block 0, the same layer code on the hint stream,
    after_proj, and behind the main block x += scale * hint in context order; a context with scale 0 is switched off."""
    m = _vace_model(mock)
    c = m.cfg
    rc, calls, nbytes = _forward_vace(mock, m, [0x6500_0000_0000, 0x6510_0000_0000], [1.0, 0.5])
    assert rc == 0, mock.wan_last_error()
    for cl in calls:
        for ptr, n in extents(cl):
            if in_ws(ptr, nbytes):
                assert ptr + n <= WS + nbytes, (cl[0], hex(ptr - WS), n)
    pe = [cl for cl in calls if cl[0] == "patch_embed"]
    assert [cl[1][0] for cl in pe[2:]] == [0x6500_0000_0000, 0x6510_0000_0000] and all(cl[1][2] == m.addr["vace_patch_embedding.weight"] for cl in pe[2:])
    ax = [cl for cl in calls if cl[0] == "axpy"]
    assert len(ax) == 2 * len(c.vace_layers) and [round(cl[3][0], 6) for cl in ax] == [1.0, 0.5] * len(c.vace_layers)
    assert all(cl[1][0] == WS and cl[1][2] == WS for cl in ax)                                   # x += scale * hint, in place on the main streams
    before = [cl for cl in calls if cl[0] == "gemm" and cl[1][1] == m.addr["vace_blocks.0.before_proj.weight"]]
    assert len(before) == 2 and all(cl[2][5] == 2 and cl[1][4] == WS for cl in before)           # c = before_proj(c) + x: residual = the main streams
    after = [cl for cl in calls if cl[0] == "gemm" and cl[1][1] in (m.addr["vace_blocks.0.after_proj.weight"], m.addr["vace_blocks.1.after_proj.weight"])]
    assert len(after) == 4
    n_ln = [cl[0] for cl in calls].count("ln_modulate")
    assert n_ln == 2 * (c.num_layers + 2 * len(c.vace_layers))                                   # main blocks + a context block per context per attachment
    rc, off, _ = _forward_vace(mock, m, [0x6500_0000_0000, 0x6510_0000_0000], [0.0, 0.5])
    assert rc == 0 and [cl[0] for cl in off].count("axpy") == len(c.vace_layers) and [cl[0] for cl in off].count("patch_embed") == 3
    rc, _, _ = _forward_vace(mock, m, [0x6500_0000_0000], [1.0], S=2)
    assert rc == 0


def test_per_frame_timesteps_need_whole_frame_shards(mock):
    """t as a vector (model.py:1812-1818): the time MLP runs on F rows and every modulation lookup takes rows_per_batch = tokens per
    frame; under sequence parallelism a rank owns whole frames or the call is refused."""
    m = Model(mock)
    rc, calls, _ = m.forward(S=2, t_frames=[0.0, 637.0])
    assert rc == 0
    assert [cl[0] for cl in calls].count("sinusoid") == 2
    assert all(cl[2][5] == 16 for cl in calls if cl[0] == "ln_modulate")                        # rows_per_batch = 4 x 4 tokens per frame
    cb, cw = GATHER_FN(lambda *a: 0), GATHER_WAIT_FN(lambda *a: 0)
    rc, _, _ = m.forward(S=2, t_frames=[0.0, 637.0, 300.0], fhw=(3, 8, 8), sp=SpInfo(0, 2, 0, 24, cb, cw, None))
    assert rc == 1 and b"whole frames" in mock.wan_last_error()
    rc, calls, _ = m.forward(S=2, t_frames=[0.0, 637.0], sp=SpInfo(1, 2, 16, 16, cb, cw, None))
    assert rc == 0 and [round(cl[3][0]) for cl in calls if cl[0] == "sinusoid"] == [637]        # rank 1 embeds its own frame's timestep only


def _clip_forward(L, m, S, nag=None, ctx_batches=None):
    """forward of a Wan2.1 i2v / flf2v model (y given) on the recording mock; wan_dit_set_clip first."""
    F, H, W = 2, 8, 8
    nbytes = L.wan_dit_workspace_bytes(m.ctx, S, F, H, W, 1)
    X = (c_void_p * S)(*[0x6000_0000_0000 + s * 0x1_0000_0000 for s in range(S)])
    C = (c_void_p * S)(*[0x6100_0000_0000 + s * 0x1_0000_0000 for s in range(S)])
    OUT = (c_void_p * S)(*[0x6200_0000_0000 + s * 0x1_0000_0000 for s in range(S)])
    a = DitArgs(S, X, 637.0, C, 0x6600_0000_0000, 0x6300_0000_0000, 0x6310_0000_0000, OUT, F, H, W, WS, nbytes, None, None, None, None, None,
                None, 1.0, None, 0, 0, None, None, *((0.0, 0.0, 0.0, None) if nag is None else (*nag, (c_int * S)(*ctx_batches))), None, 0, 0)
    L.mock_reset()
    rc = L.wan_dit_forward_ex(m.ctx, ctypes.byref(a), None)
    return rc, [(c.name.decode(), list(c.p), list(c.i), list(c.f)) for c in (L.mock_get(i).contents for i in range(L.mock_count()))]


def test_mixed_precision_plan_serves_the_clip_branch(mock):
    """Round 6: a Wan2.1 i2v model under the mixed-precision plan (no lock of model.py:1330-1371 names img_emb or k_img / v_img).  The CLIP
    branch is the bf16 plan's own -- text attention into xm, the 257 CLIP tokens attended in place, the two results added -- between the fp32
    norm3 in front of the q Linear and the fp32-stream o projection behind it; nothing of the bf16 plan's row kernels runs."""
    L = mock
    L.wan_dit_set_clip.argtypes = [c_void_p, c_void_p, c_void_p]
    m = Model(L, "tiny_i2v21", mixed=True)
    assert L.wan_dit_set_clip(m.ctx, c_void_p(0x6700_0000_0000), None) == 0, L.wan_last_error()
    rc, calls = _clip_forward(L, m, 2)
    assert rc == 0, L.wan_last_error()
    names = [cl[0] for cl in calls]
    assert not {"ln_modulate", "ln_affine", "head", "patch_embed"} & set(names)
    i0 = names.index("mx_ln_affine")                                        # block 0's norm3
    j0 = i0 + 1 + names[i0 + 1:].index("gemm_res32")                         # ... its cross-attention o projection into the fp32 stream
    seg = names[i0:j0 + 1]
    assert seg.count("attention") == 2 and seg.count("add") == 1 and seg.index("add") > max(k for k, n in enumerate(seg) if n == "attention")
    att = [cl for cl in calls[i0:j0] if cl[0] == "attention"]
    add = [cl for cl in calls[i0:j0] if cl[0] == "add"][0]
    assert att[0][1][0] == att[1][1][0] == att[1][1][3] and att[0][1][3] != att[0][1][0]        # same q; text result elsewhere (xm), CLIP result in place
    assert add[1][:3] == [att[0][1][3], att[1][1][3], att[1][1][3]]                               # q <- text + image
    assert calls[j0][1][0] == att[1][1][3]                                                       # the o projection reads the sum


def test_flf2v_clip_context_of_two_images(mock):
    """flf2v_720p (model.py:878-887, :472-473): img_emb adds its position embedding to the 2 x 257 CLIP tokens and projects 514; the
    blocks' image branch takes the first 257 of them, the text branch [the other 257 ; the 512 text tokens] -- assembled once per
    forward, K / V^T projected from 769 rows, V^T row pitch 832 with zeroed pad columns.  The plain i2v model keeps 257 / 512."""
    L = mock
    L.wan_dit_set_clip.argtypes = [c_void_p, c_void_p, c_void_p]
    m = Model(L, "tiny_flf2v")
    d, S = m.cfg.dim, 2
    L.mock_reset()
    assert L.wan_dit_set_clip(m.ctx, c_void_p(0x6700_0000_0000), None) == 0, L.wan_last_error()
    calls = [(c.name.decode(), list(c.p), list(c.i)) for c in (L.mock_get(i).contents for i in range(L.mock_count()))]
    assert [c[0] for c in calls] == ["add", "ln_affine", "gemm", "act", "gemm", "ln_affine"]
    assert calls[0][1][0] == 0x6700_0000_0000 and calls[0][1][1] == m.addr["img_emb.emb_pos"] and calls[0][2][0] == 514 * 1280
    assert calls[1][1][0] == calls[0][1][2] and calls[1][2][:2] == [514, 1280]            # LayerNorm of the sum, 514 rows
    assert calls[2][2][:3] == [514, 1280, 1280] and calls[4][2][:3] == [514, d, 1280] and calls[5][2][:2] == [514, d]
    clip_ctx = calls[5][1][1]
    rc, fw = _clip_forward(L, m, S)
    assert rc == 0, L.wan_last_error()
    cp = [c for c in fw if c[0] == "memcpy" and c[2][0] in (257 * d * 2, 512 * d * 2)]
    assert len(cp) == 2 * S
    for s in range(S):                                                                     # [second image's tokens ; text tokens] per stream
        a, b = cp[2 * s], cp[2 * s + 1]
        assert a[1][1] == clip_ctx + 257 * d * 2 and a[2][0] == 257 * d * 2
        assert b[1][0] == a[1][0] + 257 * d * 2 and b[2][0] == 512 * d * 2
        if s:
            assert a[1][0] == cp[0][1][0] + s * 769 * d * 2
    ctx_x = cp[0][1][0]
    ms = [c for c in fw if c[0] == "memset" and c[2][0] == S * d * 832 * 2]
    assert len(ms) == 1
    att = [c for c in fw if c[0] == "attention"]
    text = [c for c in att if c[2][3] == 769]
    img = [c for c in att if c[2][3] == 257]
    assert len(text) == m.cfg.num_layers and len(img) == m.cfg.num_layers and all(c[2][4] == 832 for c in text) and all(c[2][4] == 320 for c in img)
    assert all(c[1][2] == ms[0][1][0] for c in text)                                       # V^T images = the zero-padded buffer
    kg = [c for c in fw if c[0] == "gemm" and c[1][0] == ctx_x and c[2][0] == S * 769]     # K projection of both streams' 769 rows
    vg = [c for c in fw if c[0] == "gemm" and c[2][0] == 769 and c[2][5] == 3]             # V^T projections, one per stream
    assert len(kg) == m.cfg.num_layers and len(vg) == S * m.cfg.num_layers and all(c[2][4] == 832 for c in vg)
    assert {c[1][0] for c in vg} == {ctx_x, ctx_x + 769 * d * 2}
    ki = [c for c in fw if c[0] == "gemm" and c[1][0] == clip_ctx]                         # k_img / v_img: the FIRST 257 tokens
    assert len(ki) == 2 * m.cfg.num_layers and all(c[2][0] == 257 for c in ki)
    # together with normalized attention guidance: refused, with a message
    rc, _ = _clip_forward(L, m, 1, nag=(3.0, 2.5, 0.25), ctx_batches=[2])
    assert rc != 0 and b"flf2v" in L.wan_last_error()
    # the plain Wan2.1 i2v model: 257 tokens, text branch of 512, no assembly
    m1 = Model(L, "tiny_i2v21")
    L.mock_reset()
    assert L.wan_dit_set_clip(m1.ctx, c_void_p(0x6700_0000_0000), None) == 0
    c1 = [(c.name.decode(), list(c.i)) for c in (L.mock_get(i).contents for i in range(L.mock_count()))]
    assert [c[0] for c in c1] == ["ln_affine", "gemm", "act", "gemm", "ln_affine"] and c1[0][1][:2] == [257, 1280]
    rc, fw1 = _clip_forward(L, m1, S)
    assert rc == 0
    assert not [c for c in fw1 if c[0] == "memcpy" and c[2][0] == 257 * d * 2]
    assert {c[2][3] for c in fw1 if c[0] == "attention"} == {16 * 2, 512, 257} or {c[2][3] for c in fw1 if c[0] == "attention"} >= {512, 257}


def test_forward_as_a_replayed_launch_list(mock):
    """wan_dit_forward_graph: a key's first call runs eagerly (timestep read from device memory), the second is captured on the
    context's own stream -- the SAME launch list as the eager forward, launch for launch -- and launched once on the caller's stream,
    every later call is one graph launch behind the one-float timestep kernel; other pointers = another key; calls that cannot be
    replayed fall through to the eager path; the poll hook is called once in front; registering a weight drops the captured lists."""
    m = Model(mock)
    c = m.cfg
    USER = 0x7777
    rc, eager, _ = m.forward(S=2)
    assert rc == 0
    def strip(calls):       # the launch list without what differs by construction: the timestep's source
        return [(n, p, i) for n, p, i, f in calls if n not in ("sinusoid", "sinusoid_dev", "set_f32")]
    # first sight: eager, t through device memory
    rc, c1, _ = m.forward(S=2, graph=True, t=900.0, stream=USER)
    assert rc == 0 and m.how == 1, mock.wan_last_error()
    assert c1[0][0] == "set_f32" and c1[0][3][0] == 900.0 and c1[0][1][1] == USER
    sd = [cl for cl in c1 if cl[0] == "sinusoid_dev"]
    assert len(sd) == 1 and sd[0][1][0] == c1[0][1][0] and not [cl for cl in c1 if cl[0] == "sinusoid"]
    assert strip(c1[1:]) == strip(eager)
    # second sight: captured (on a stream of the context's own) and launched on the caller's
    rc, c2, _ = m.forward(S=2, graph=True, t=800.0, stream=USER)
    assert rc == 0 and m.how == 2
    names = [cl[0] for cl in c2]
    assert names[0] == "set_f32" and names[1] == "begin_capture" and names[-2] == "end_capture" and names[-1] == "graph_launch"
    assert c2[1][1][0] != USER and c2[-1][1][0] == USER
    assert strip(c2[2:-2]) == strip(eager)                                         # what was captured = the eager launch list
    span = c2[-1][2][:2]
    # from then on: the timestep kernel + ONE launch
    for t in (700.0, 10.0):
        rc, c3, _ = m.forward(S=2, graph=True, t=t, stream=USER)
        assert rc == 0 and m.how == 3 and [cl[0] for cl in c3] == ["set_f32", "graph_launch"] and c3[0][3][0] == t
    # other latents' addresses: another key -> eager again, then its own capture; the first key still replays
    rc, c4, _ = m.forward(S=2, graph=True, x0=0x6800_0000_0000, stream=USER)
    assert rc == 0 and m.how == 1
    rc, _, _ = m.forward(S=2, graph=True, stream=USER)
    assert rc == 0 and m.how == 3
    rc, _, _ = m.forward(S=1, graph=True, stream=USER)
    assert rc == 0 and m.how == 1                                                  # one stream: another key
    # not replayable: per-frame timesteps, step-skipping, sequence parallelism -> the eager forward, by-value timestep
    rc, c5, _ = m.forward(S=2, graph=True, t_frames=[0.0, 637.0])
    assert rc == 0 and m.how == 0 and "set_f32" not in [cl[0] for cl in c5] and "sinusoid" in [cl[0] for cl in c5]
    rc, _, _ = m.forward(S=2, graph=True, should_calc=[1, 1], residual=[0x6400_0000_0000, 0x6410_0000_0000])
    assert rc == 0 and m.how == 0
    # the poll hook: once, in front; a stop request aborts before anything is enqueued
    seen = []
    stop = POLL_FN(lambda user, i: seen.append(i) or 1)
    rc, c6, _ = m.forward(S=2, graph=True, poll=stop, stream=USER)
    assert rc == WAN_ABORTED and seen == [0] and not c6
    go = POLL_FN(lambda user, i: seen.append(i) or 0)
    rc, c7, _ = m.forward(S=2, graph=True, poll=go, stream=USER)
    assert rc == 0 and m.how == 3 and seen == [0, 0]
    # malformed arguments never reach the key (which reads through the arrays): the eager entry reports them
    bad = DitArgs()
    bad.S = 2
    how = c_int(-1)
    assert mock.wan_dit_forward_graph(m.ctx, ctypes.byref(bad), None, ctypes.byref(how)) != 0 and how.value == 0 and b"null" in mock.wan_last_error()
    # a re-registered weight: the captured lists hold the old pointer and are dropped
    k = "blocks.0.ffn.0.bias"
    assert mock.wan_dit_set_weight(m.ctx, k.encode(), c_void_p(m.addr[k] + 0x100), 0, c.ffn_dim) == 0
    rc, _, _ = m.forward(S=2, graph=True, stream=USER)
    assert rc == 0 and m.how == 1


def test_text_cache_keeps_cross_attention_k_v_across_forwards(mock):
    """wan_dit_args.context_key (round 6): a forward with a non-zero key computes the text embedding and every block's cross-attention
    K / V^T like the plain forward -- but into buffers of the context's own -- and the next forward with that key skips their launches
    (4 per block + 3 per forward at S = 2: the K Linear, its RMSNorm, one V^T Linear per stream; the text embedding: one first Linear per stream + the second) and reads the same
    buffers; everything else is launch for launch the plain forward.  Another key, another stream count, a re-registered weight, a
    stream that skips, NAG and skip-layer guidance recompute."""
    m = Model(mock)
    c = m.cfg
    rc, plain, nbytes = m.forward(S=2)
    assert rc == 0
    names = lambda calls: [cl[0] for cl in calls]
    rc, miss, _ = m.forward(S=2, context_key=11)
    assert rc == 0 and names(miss) == names(plain), mock.wan_last_error()
    # the miss: the cross K / V^T of every block land OUTSIDE the workspace, in one buffer pair per block; the cross-attention reads them
    ck = [cl for cl in miss if cl[0] == "gemm" and cl[2][0] == 2 * TL and cl[2][5] == 0 and cl[2][1] == c.dim and cl[2][2] == c.dim]
    ck = [cl for cl in ck if not in_ws(cl[1][3], nbytes)]
    cv = [cl for cl in miss if cl[0] == "gemm" and cl[2][0] == TL and cl[2][5] == 3 and not in_ws(cl[1][3], nbytes)]
    assert len(ck) == c.num_layers and len(cv) == 2 * c.num_layers
    assert len({cl[1][3] for cl in ck}) == c.num_layers
    xatt = [cl for cl in miss if cl[0] == "attention" and cl[2][3] == TL]
    assert [cl[1][1] for cl in xatt] == [cl[1][3] for cl in ck]
    assert [cl[1][2] for cl in xatt] == [cv[2 * i][1][3] for i in range(c.num_layers)]
    # the hit: no text embedding, no cross K / V work; every other launch as before, the cross-attention on the same buffers
    rc, hit, _ = m.forward(S=2, context_key=11, t=500.0)
    assert rc == 0
    gone = [cl for cl in miss if cl not in hit and cl[0] != "sinusoid"]
    assert len(gone) == 3 + 4 * c.num_layers                     # te0 x 2 streams + te2; per block: ck GEMM + its norm + 2 V^T GEMMs
    assert sorted(set(names(gone))) == ["gemm", "rmsnorm_rope"]
    xatt_hit = [cl for cl in hit if cl[0] == "attention" and cl[2][3] == TL]
    assert [(cl[1][1], cl[1][2]) for cl in xatt_hit] == [(cl[1][1], cl[1][2]) for cl in xatt]
    strip = lambda calls: [(n, p, i) for n, p, i, f in calls if n != "sinusoid"]
    assert [x for x in strip(miss) if x not in strip(gone)] == strip(hit)
    # a second key gets the second slot; the first one still hits; a third evicts the least recently used one
    rc, m2, _ = m.forward(S=2, context_key=12)
    assert rc == 0 and names(m2) == names(plain)
    rc, h1, _ = m.forward(S=2, context_key=11)
    assert rc == 0 and names(h1) == names(hit)
    rc, m3, _ = m.forward(S=2, context_key=13)
    assert rc == 0 and names(m3) == names(plain)
    rc, h1b, _ = m.forward(S=2, context_key=11)
    assert rc == 0 and names(h1b) == names(hit)                  # 12 was the older one
    rc, m2b, _ = m.forward(S=2, context_key=12)
    assert rc == 0 and names(m2b) == names(plain)
    # another stream layout under the same key, a re-registered weight: recomputed
    rc, s1, _ = m.forward(S=1, context_key=11)
    rc1, p1, _ = m.forward(S=1)
    assert rc == 0 and rc1 == 0 and names(s1) == names(p1)
    rc, s1h, _ = m.forward(S=1, context_key=11)
    assert rc == 0 and len(s1h) == len(p1) - (2 + 3 * c.num_layers)
    k = "blocks.0.cross_attn.k.bias"
    assert mock.wan_dit_set_weight(m.ctx, k.encode(), c_void_p(m.addr[k] + 0x100), 0, c.dim) == 0
    rc, after, _ = m.forward(S=1, context_key=11)
    assert rc == 0 and names(after) == names(p1)
    # not served: a skipped stream, skip-layer guidance, NAG -- the plain launch lists, in the workspace
    rc, sk, _ = m.forward(S=2, context_key=12, should_calc=[1, 0], residual=[0x6400_0000_0000, 0x6410_0000_0000])
    rc0, sk0, _ = m.forward(S=2, should_calc=[1, 0], residual=[0x6400_0000_0000, 0x6410_0000_0000])
    assert rc == 0 and rc0 == 0 and strip(sk) == strip(sk0)
    rc, pl, _ = m.forward(S=2, context_key=12, perturb=[1])
    rc0, pl0, _ = m.forward(S=2, perturb=[1])
    assert rc == 0 and rc0 == 0 and strip(pl) == strip(pl0)
    rc, ng, _ = m.forward(S=2, context_key=12, nag=(2.0, 2.5, 0.25), ctx_batches=[2, 1])
    rc0, ng0, _ = m.forward(S=2, nag=(2.0, 2.5, 0.25), ctx_batches=[2, 1])
    assert rc == 0 and rc0 == 0 and strip(ng) == strip(ng0)
    # the replayed launch list is captured in its hit form only, keyed by the slot
    m = Model(mock)
    rc, g0, _ = m.forward(S=2, graph=True, context_key=21, stream=0x7777)
    assert rc == 0 and m.how == 1 and "set_f32" not in names(g0)                   # a miss: plain eager forward (fills the slot)
    rc, g1, _ = m.forward(S=2, graph=True, context_key=21, stream=0x7777)
    assert rc == 0 and m.how == 1 and names(g1)[0] == "set_f32"                    # first sight of the hit form
    rc, g2, _ = m.forward(S=2, graph=True, context_key=21, stream=0x7777)
    assert rc == 0 and m.how == 2
    rc, g3, _ = m.forward(S=2, graph=True, context_key=21, stream=0x7777)
    assert rc == 0 and m.how == 3 and names(g3) == ["set_f32", "graph_launch"]
    cap = [cl for cl in g2 if cl[0] == "attention" and cl[2][3] == TL]
    assert cap and all(not in_ws(cl[1][1], nbytes) for cl in cap)
