#!/usr/bin/env python
"""Record the launch lists of wan_dit_forward* on the mock kernels (tests/mock/mock_ops.cpp) for TWO versions of csrc/dit.hip and compare
them call for call, argument for argument -- what held the round-6 split of dit_forward_impl into stages to "nothing changes".
usage: compare_forward_launch_lists.py <old dit.hip> [<new dit.hip> = the tree's]   (needs hipcc and g++; no GPU)"""
import ctypes
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_dit_host_logic_cpu as T  # noqa: E402
from wan2gp_amd.lib import GATHER_FN, GATHER_WAIT_FN, SP_ULYSSES, SpInfo  # noqa: E402


def build(dit, tag):
    d = tempfile.mkdtemp(prefix="mock_" + tag)
    inc = os.path.join(ROOT, "include")
    src = os.path.join(d, "dit.hip")
    open(src, "w").write(open(dit).read())
    subprocess.run([T.HIPCC, "--offload-arch=gfx950", "-O1", "-std=c++17", "-fPIC", "-Wno-unused-function", "-I" + inc, "-I" + os.path.join(ROOT, "wan2gp_amd", "csrc"),
                    "-c", src, "-o", os.path.join(d, "dit.o")], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.run(["g++", "-std=c++17", "-O1", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-c", os.path.join(ROOT, "tests", "mock", "mock_ops.cpp"),
                    "-o", os.path.join(d, "mock.o")], check=True)
    so = os.path.join(d, "libwanhip_mock.so")
    subprocess.run(["g++", "-shared", "-fPIC", "-o", so, os.path.join(d, "dit.o"), os.path.join(d, "mock.o")], check=True)
    L = ctypes.CDLL(so)
    L.mock_get.restype = ctypes.POINTER(T.Call)
    L.wan_last_error.restype = ctypes.c_char_p
    L.wan_dit_workspace_bytes.restype = ctypes.c_int64
    L.wan_dit_workspace_bytes.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.wan_dit_set_weight.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64]
    L.wan_dit_forward_ex.argtypes = [ctypes.c_void_p, ctypes.POINTER(T.DitArgs), ctypes.c_void_p]
    L.wan_dit_forward_graph.argtypes = [ctypes.c_void_p, ctypes.POINTER(T.DitArgs), ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    return L


def scenarios(L):
    """(name, rc, calls) of every scenario; hook events are part of the record (order relative to the launches)."""
    out = []

    def sp_info(world, rank, Ll, mode, chunks, events):
        def begin(user, which, send, recv, nbytes, stream):
            events.append(("begin", which, send, recv, nbytes, L.mock_count()))
            return 0

        def wait(user, which, stream):
            events.append(("wait", which, L.mock_count()))
            return 0
        cb, cw = GATHER_FN(begin), GATHER_WAIT_FN(wait)
        keep.append((cb, cw))
        return SpInfo(rank, world, rank * Ll, Ll, cb, cw, None, mode, cb, cw, chunks)
    keep = []
    for name in ("tiny", "small"):
        for fp8 in (False, True):
            for mixed in (False, True):
                if fp8 and mixed and name == "small":
                    continue
                m = T.Model(L, name=name, fp8=fp8, mixed=mixed)
                tag = "%s%s%s" % (name, "_fp8" if fp8 else "", "_mixed" if mixed else "")
                for S in (1, 2):
                    out.append((tag + "_S%d" % S,) + m.forward(S=S)[:2])
                    out.append((tag + "_S%d_key" % S,) + m.forward(S=S, context_key=5)[:2])
                    out.append((tag + "_S%d_key_hit" % S,) + m.forward(S=S, context_key=5, t=100.0)[:2])
                out.append((tag + "_tframes",) + m.forward(S=2, t_frames=[10.0, 637.0])[:2])
                out.append((tag + "_skip",) + m.forward(S=2, should_calc=[1, 0], residual=[0x6400_0000_0000, 0x6410_0000_0000])[:2])
                out.append((tag + "_slg",) + m.forward(S=2, perturb=[1])[:2])
                if not mixed:
                    out.append((tag + "_nag",) + m.forward(S=2, nag=(2.0, 2.5, 0.25), ctx_batches=[2, 1])[:2])
                heads = m.cfg.num_heads
                for world in (2, 4):
                    Ll = 2 * 16 // world
                    for mode in (0, SP_ULYSSES):
                        if mode == SP_ULYSSES and heads % world:
                            continue
                        for chunks in ((1, 2) if mode == SP_ULYSSES else (1,)):
                            ev = []
                            rc, calls, _ = m.forward(S=2, sp=sp_info(world, world - 1, Ll, mode, chunks, ev))
                            out.append(("%s_sp%d_mode%d_c%d" % (tag, world, mode, chunks), rc, calls + [("hook",) + e for e in ev]))
                out.append((tag + "_graph1",) + m.forward(S=2, graph=True, stream=0x7777)[:2])
                out.append((tag + "_graph2",) + m.forward(S=2, graph=True, stream=0x7777)[:2])
                out.append((tag + "_graph3",) + m.forward(S=2, graph=True, stream=0x7777)[:2])
    return out


def main():
    old = sys.argv[1]
    new = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "wan2gp_amd", "csrc", "dit.hip")
    a, b = scenarios(build(old, "old")), scenarios(build(new, "new"))
    assert [x[0] for x in a] == [x[0] for x in b]
    bad = 0
    for (n, rca, ca), (_, rcb, cb) in zip(a, b):
        # heap addresses (the text cache's buffers, hipMalloc in the mock) differ between two processes' runs: compare them by order of first appearance
        def norm(calls):
            seen = {}
            outl = []
            for cl in calls:
                if cl[0] == "hook":
                    outl.append(cl)
                    continue
                name, p, i, f = cl
                q = []
                for v in p:
                    if v and not (0x1000_0000_0000 <= v < 0x5000_0000_0000 or 0x6000_0000_0000 <= v < 0x7800_0000_0000):   # (the tests' fake addresses; anything else is the mock's heap)
                        v = seen.setdefault(v, 0xAAAA_0000 + len(seen))
                    q.append(v)
                outl.append((name, q, i, f))
            return outl
        if rca != rcb or norm(ca) != norm(cb):
            bad += 1
            print("DIFFERS:", n, rca, rcb, len(ca), len(cb))
    print("%d scenarios, %d launches recorded, %d differ" % (len(a), sum(len(x[2]) for x in a), bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
