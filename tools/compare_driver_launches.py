#!/usr/bin/env python
"""Does a change of the forward driver (wan2gp_amd/csrc/dit.hip) alter what the library launches?  (no GPU needed)

Builds dit.hip of a given git revision and of the working tree against the recording mock of tests/mock/mock_ops.cpp and
compares the COMPLETE launch lists -- op, every pointer, every integer argument -- of `wan_dit_forward_ex` over a grid of
scenarios: 1-3 streams, every step-skipping pattern, NAG, per-frame timesteps, sequence parallelism at world 2 / 4, scaled-fp8
checkpoints, VACE with one / two / switched-off contexts (with step skipping), i2v.  Identical lists = the same kernels on the same
data in the same order: the GPU validation of the old revision carries over.

    python tools/compare_driver_launches.py <git-rev>          # e.g. the last revision whose GPU suite ran
"""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(rev, d):
    os.makedirs(os.path.join(d, "include"), exist_ok=True)
    os.makedirs(os.path.join(d, "csrc"), exist_ok=True)
    for rel, dst in (("include/wanhip.h", "include/wanhip.h"), ("wan2gp_amd/csrc/dit.hip", "csrc/dit.hip"), ("wan2gp_amd/csrc/common.h", "csrc/common.h")):
        text = open(os.path.join(ROOT, rel)).read() if rev is None else subprocess.check_output(["git", "-C", ROOT, "show", f"{rev}:{rel}"], text=True)
        open(os.path.join(d, dst), "w").write(text.replace("../../include/wanhip.h", "../include/wanhip.h"))
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-std=c++17", "-fPIC", "-Wno-unused-function", "-c", os.path.join(d, "csrc", "dit.hip"),
                    "-o", os.path.join(d, "dit.o")], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.run(["g++", "-std=c++17", "-O1", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-c", os.path.join(ROOT, "tests", "mock", "mock_ops.cpp"),
                    "-o", os.path.join(d, "mock.o")], check=True)
    so = os.path.join(d, "libdriver.so")
    subprocess.run(["g++", "-shared", "-fPIC", "-o", so, os.path.join(d, "dit.o"), os.path.join(d, "mock.o")], check=True)
    return so


import sys  # noqa: E402
REV = sys.argv[1] if len(sys.argv) > 1 else "HEAD"
_tmp = tempfile.mkdtemp(prefix="driver_")
OLD_SO, NEW_SO = build(REV, os.path.join(_tmp, "old")), build(None, os.path.join(_tmp, "new"))
print(f"comparing the forward driver of {REV} with the working tree")
import ctypes  # noqa: E402
sys.path.insert(0, ROOT)
from ctypes import POINTER, c_char_p, c_int, c_int64, c_void_p
import importlib
T = importlib.import_module("tests.test_dit_host_logic_cpu")
from wan2gp_amd.lib import SpInfo, GATHER_FN, GATHER_WAIT_FN, POLL_FN
from oracle import wan_oracle as O

def load(path):
    L = ctypes.CDLL(path)
    L.mock_get.restype = POINTER(T.Call); L.wan_last_error.restype = c_char_p
    L.wan_dit_workspace_bytes.restype = c_int64
    L.wan_dit_workspace_bytes.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_int]
    L.wan_dit_set_weight.argtypes = [c_void_p, c_char_p, c_void_p, c_int, c_int64]
    L.wan_dit_forward_ex.argtypes = [c_void_p, POINTER(T.DitArgs), c_void_p]
    return L
old, new = load(OLD_SO), load(NEW_SO)
cb, cw = GATHER_FN(lambda *a: 0), GATHER_WAIT_FN(lambda *a: 0)
def scenarios(L, name, fp8):
    m = T.Model(L, name, fp8=fp8)
    res = [0x6400_0000_0000, 0x6400_1000_0000, 0x6400_2000_0000]
    yield "S1", m.forward(S=1)
    yield "S2", m.forward(S=2)
    yield "S3", m.forward(S=3)
    for sc in ([1, 0], [0, 1], [1, 1], [0, 0]):
        yield f"skip{sc}", m.forward(S=2, should_calc=sc, residual=res[:2])
    yield "skip101", m.forward(S=3, should_calc=[1, 0, 1], residual=res)
    yield "nag21", m.forward(S=2, nag=(11.0, 2.5, 0.25), ctx_batches=[2, 1])
    yield "nag2", m.forward(S=1, nag=(3.0, 2.5, 0.25), ctx_batches=[2])
    yield "nag_skip", m.forward(S=2, nag=(11.0, 2.5, 0.25), ctx_batches=[2, 1], should_calc=[0, 1], residual=res[:2])
    yield "tframes", m.forward(S=2, t_frames=[0.0, 637.0])
    for r in (0, 1):
        sp = SpInfo(r, 2, r * 16, 16, cb, cw, None)
        yield f"sp{r}", m.forward(S=2, sp=sp)
    sp = SpInfo(3, 4, 24, 8, cb, cw, None)
    yield "sp4", m.forward(S=2, sp=sp)
    yield "big", m.forward(S=2, fhw=(3, 10, 14))
bad = 0
for name, fp8 in (("tiny", False), ("tiny", True), ("small", False), ("tiny_i2v", False)):
    if name == "tiny_i2v":
        continue   # needs y: covered below
    for (k, a), (_, b) in zip(scenarios(old, name, fp8), scenarios(new, name, fp8)):
        same = a[0] == b[0] and a[1] == b[1] and a[2] == b[2]
        print(f"{name:6s} fp8={fp8!s:5s} {k:12s} rc={a[0]} launches={len(a[1]):4d} identical={same}")
        bad += not same
print("MISMATCHES", bad)

# ---- VACE (one / two contexts, with and without step skipping) and i2v (y), per driver -----------------------------------------
def run_raw(L, m, S, fhw, y=None, vace=None, scales=None, should_calc=None, residual=None):
    F, H, W = fhw
    nb = L.wan_dit_workspace_bytes(m.ctx, S, F, H, W, 1)
    X = (c_void_p * S)(*[0x6000_0000_0000 + s * 0x1_0000_0000 for s in range(S)])
    C = (c_void_p * S)(*[0x6100_0000_0000 + s * 0x1_0000_0000 for s in range(S)])
    OUT = (c_void_p * S)(*[0x6200_0000_0000 + s * 0x1_0000_0000 for s in range(S)])
    FL = None if should_calc is None else (c_int * S)(*should_calc)
    RP = None if residual is None else (c_void_p * S)(*residual)
    nv = 0 if not vace or len(vace) == 1 else len(vace)
    VP = (c_void_p * nv)(*vace) if nv else None
    VS = (ctypes.c_float * nv)(*scales) if nv else None
    a = T.DitArgs(S, X, 588.0, C, y, 0x6300_0000_0000, 0x6310_0000_0000, OUT, F, H, W, T.WS, nb, None, None, None, FL, RP,
                  (vace[0] if vace and len(vace) == 1 else None), (scales[0] if vace and len(vace) == 1 else 1.0), None, 0, nv, VP, VS,
                  0.0, 0.0, 0.0, None, None, 0, 0)
    L.mock_reset()
    rc = L.wan_dit_forward_ex(m.ctx, ctypes.byref(a), None)
    return rc, [(c.name, list(c.p), list(c.i), list(c.f)) for c in (L.mock_get(i).contents for i in range(L.mock_count()))], nb

def vace_model(L):
    m = T.Model.__new__(T.Model)
    m.L, m.cfg = L, O.make_config("tiny_vace")
    c = m.cfg
    dc = T.DitConfig(c.dim, c.ffn_dim, c.num_heads, c.num_layers, c.in_dim, c.out_dim, c.text_dim, c.freq_dim, c.text_len, c.eps)
    m.ctx = c_void_p()
    assert L.wan_dit_create(ctypes.byref(dc), ctypes.byref(m.ctx)) == 0
    arr = (c_int * len(c.vace_layers))(*c.vace_layers)
    assert L.wan_dit_set_vace_layers(m.ctx, arr, len(c.vace_layers)) == 0
    assert L.wan_dit_set_vace_contexts(m.ctx, 2) == 0
    for n, (k, shape) in enumerate(O.param_shapes(c).items()):
        numel = 1
        for s_ in shape: numel *= s_
        dt = 1 if k.startswith(("patch_embedding.", "head.", "vace_patch_embedding.")) else 0
        assert L.wan_dit_set_weight(m.ctx, k.encode(), c_void_p(0x1000_0000_0000 + n * 0x10_0000_0000), dt, numel) == 0, (k, L.wan_last_error())
    return m

res = [0x6400_0000_0000, 0x6400_1000_0000]
for tag, kw in (("vace1", dict(vace=[0x6500_0000_0000], scales=[1.0])), ("vace1_s06", dict(vace=[0x6500_0000_0000], scales=[0.6])),
                ("vace2", dict(vace=[0x6500_0000_0000, 0x6510_0000_0000], scales=[1.0, 0.5])),
                ("vace2_off", dict(vace=[0x6500_0000_0000, 0x6510_0000_0000], scales=[0.0, 0.5])),
                ("vace_skip", dict(vace=[0x6500_0000_0000], scales=[1.0], should_calc=[1, 0], residual=res)),
                ("vace_none", dict())):
    a = run_raw(old, vace_model(old), 2, (2, 8, 8), **kw)
    b = run_raw(new, vace_model(new), 2, (2, 8, 8), **kw)
    same = a == b
    print(f"tiny_vace {tag:10s} rc={a[0]} launches={len(a[1]):4d} identical={same} {old.wan_last_error() if a[0] else ''}")
    bad += not same
for S in (1, 2):
    a = run_raw(old, T.Model(old, "tiny_i2v"), S, (2, 8, 8), y=0x6600_0000_0000)
    b = run_raw(new, T.Model(new, "tiny_i2v"), S, (2, 8, 8), y=0x6600_0000_0000)
    print(f"tiny_i2v S={S} rc={a[0]} launches={len(a[1])} identical={a == b}")
    bad += a != b
print("MISMATCHES (all)", bad)

sys.exit(1 if bad else 0)
