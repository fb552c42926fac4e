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


def lift_cfg_branch(ns):
    """The plain two-stream branch of denoise_with_cfg_fn (any2video.py:1702-1722: `noise_pred_cond, noise_pred_uncond =
    ret_values` ... `noise_pred = noise_pred_uncond + guide_scale * (noise_pred_text - noise_pred_uncond)`), lifted by its
    first / last statement, dedented and wrapped in a function of the names it reads.  Statements untouched."""
    import textwrap
    lines = open(os.path.join(REF, "models/wan/any2video.py")).read().split("\n")
    i0 = next(i for i, l in enumerate(lines) if l.strip() == "noise_pred_cond, noise_pred_uncond = ret_values")
    i1 = next(i for i in range(i0, len(lines)) if lines[i].strip().startswith("noise_pred = noise_pred_uncond + guide_scale * (noise_pred_text - noise_pred_uncond)"))
    body = textwrap.dedent("\n".join(lines[i0:i1 + 1]))
    src = ("def cfg_branch(ret_values, guide_scale, i, apg_switch, cfg_star_switch, cfg_zero_step, batch_size, text_momentumbuffer, apg_norm_threshold):\n"
           + textwrap.indent(body, "    ") + "\n    return noise_pred\n")
    exec(compile(src, "any2video.py[%d:%d]" % (i0 + 1, i1 + 1), "exec"), ns)
    return ns["cfg_branch"]


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
    # CFG-Zero*: the branch itself, executed (steps <= cfg_zero_step and later ones); it scales its uncond argument in place
    ns = dict(a); ns.update(m)
    branch = lift_cfg_branch(ns)
    c0, u0 = steps[0]
    out["cfgzero_early"] = branch((c0.clone(), u0.clone()), 4.0, 2, 0, 1, 5, 1, None, 55).numpy()
    out["cfgzero_late"] = branch((c0.clone(), u0.clone()), 4.0, 9, 0, 1, 5, 1, None, 55).numpy()
    out["cfg_plain"] = branch((c0.clone(), u0.clone()), 4.0, 9, 0, 0, 5, 1, None, 55).numpy()
    par, orth = m["project"](steps[0][1], steps[0][0])
    out["proj_par"], out["proj_orth"] = par.numpy(), orth.numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items() if not k.startswith(("cond_", "uncond_"))})


if __name__ == "__main__":
    main()
