"""TEST INFRASTRUCTURE ONLY -- tests/golden/guidance.npz from the REFERENCE's own guidance helpers, executed unmodified:
`optimized_scale` (models/wan/any2video.py:67-79) and `MomentumBuffer` / `project` / `adaptive_projected_guidance`
(models/wan/multitalk/multitalk_utils.py:339-381).  The two files import half of the application, so the definitions are
lifted out of their source with `ast` (bodies untouched) and executed in a namespace that only holds torch.
Run in the build container:   python oracle/make_golden_guidance.py"""
import ast
import io
import os
import contextlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("WAN_REFERENCE_ROOT", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden", "guidance.npz")


def lift(path, names):
    src = open(os.path.join(REF, path)).read()
    tree = ast.parse(src)
    picked = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names]
    assert {n.name for n in picked} == set(names), (path, names)
    ns = {"torch": torch}
    exec(compile(ast.Module(body=picked, type_ignores=[]), path, "exec"), ns)
    return ns


def main():
    a = lift("models/wan/any2video.py", ["optimized_scale"])
    m = lift("models/wan/multitalk/multitalk_utils.py", ["MomentumBuffer", "project", "adaptive_projected_guidance"])
    g = torch.Generator().manual_seed(23)
    out = {}
    cond = torch.randn(1, 16, 3, 6, 8, generator=g)
    steps = [(cond + 0.3 * torch.randn(cond.shape, generator=g), torch.randn(cond.shape, generator=g) * (1.0 + 40.0 * (i == 2)))
             for i in range(4)]
    out["alpha"] = a["optimized_scale"](steps[0][0].view(1, -1), steps[0][1].view(1, -1)).numpy()
    buf = m["MomentumBuffer"](-0.75)
    with contextlib.redirect_stdout(io.StringIO()):                     # the reference prints diff_norm on every call
        for i, (c, u) in enumerate(steps):
            out[f"cond_{i}"], out[f"uncond_{i}"] = c.numpy(), u.numpy()
            out[f"apg_{i}"] = m["adaptive_projected_guidance"](c - u, c, momentum_buffer=buf, norm_threshold=55).numpy()
        out["apg_nomom_eta"] = m["adaptive_projected_guidance"](steps[1][0] - steps[1][1], steps[1][0], eta=0.3, norm_threshold=0).numpy()
    par, orth = m["project"](steps[0][1], steps[0][0])
    out["proj_par"], out["proj_orth"] = par.numpy(), orth.numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items() if not k.startswith(("cond_", "uncond_"))})


if __name__ == "__main__":
    main()
