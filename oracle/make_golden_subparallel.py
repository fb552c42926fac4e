"""TEST INFRASTRUCTURE ONLY -- tests/golden/subparallel.npz from the REFERENCE's own sub-parallel window code
(models/wan/any2video.py:1199-1387).  That code is a set of closures nested inside `WanAny2V.generate`; their definitions (and
the six statements that turn the pixel-frame window size into latent counts, :1215-1220) are lifted *verbatim* with `ast` /
source line ranges into a synthetic outer function that supplies the enclosing variables of the plain t2v / i2v / VACE case
(no prefix, no history, every variant flag off), and executed.  Run in the build container:
    python oracle/make_golden_subparallel.py"""
import ast
import os
import textwrap

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("WAN_REFERENCE_ROOT", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden", "subparallel.npz")
NESTED = ["_build_sub_parallel_windows", "_sub_parallel_select", "_sub_parallel_narrow", "_sub_parallel_output_indices",
          "_sub_parallel_model_indices", "_sub_parallel_slice_time", "_sub_parallel_token_indices", "_sub_parallel_scail2_freqs",
          "_sub_parallel_slice_freqs", "_sub_parallel_kwargs", "_sub_parallel_weight", "_sub_parallel_denoise"]


def build_outer():
    src = open(os.path.join(REF, "models/wan/any2video.py")).read()
    lines = src.split("\n")
    tree = ast.parse(src)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "WanAny2V")
    gen = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "generate")
    defs = {n.name: n for n in ast.walk(gen) if isinstance(n, ast.FunctionDef) and n.name in NESTED}
    assert set(defs) == set(NESTED), sorted(set(NESTED) - set(defs))
    first = defs["_build_sub_parallel_windows"]
    stmts = [n for n in gen.body if first.end_lineno < n.lineno < defs["_sub_parallel_select"].lineno]      # :1215-1223
    body = []
    for n in [first] + stmts + [defs[k] for k in NESTED[1:]]:
        body.append(textwrap.dedent("\n".join(lines[n.lineno - 1:n.end_lineno])))
    params = ("self, sub_parallel_window_size, sub_parallel_window_overlap, lat_frames, target_shape, kwargs, ref_images_before, "
              "ref_images_count, scail2, extended_overlapped_latents")
    pre = ("extended_latents = y_cond = y_uncond = conditions = conditions_null = None\n"
           "extended_input_dim = 0\nwanmove = steadydancer = scail = vista4d = False\nps_h = ps_w = 2\n")
    post = "return dict(windows=sub_parallel_windows, overlap=sub_parallel_overlap_latents, window=sub_parallel_window_latents, denoise=_sub_parallel_denoise, build=_build_sub_parallel_windows)\n"
    code = "def outer(" + params + "):\n" + textwrap.indent(pre + "\n".join(body) + "\n" + post, "    ")
    ns = {"torch": torch, "F": F}
    exec(compile(code, "any2video_subparallel_lifted.py", "exec"), ns)
    return ns["outer"]


def fake_denoise_factory(kwargs):
    """A deterministic stand-in for denoise_with_cfg_fn: depends on the window latents AND on the sliced keywords it sees."""
    def fn(lat):
        cos, sin = kwargs["freqs"]
        f = lat.shape[2]
        tok = cos.shape[0] // f
        per_frame = cos.view(f, tok, -1).mean(dim=(1, 2)) + 0.5 * sin.view(f, tok, -1).mean(dim=(1, 2))
        out = 0.3 * lat + per_frame.view(1, 1, f, 1, 1)
        if kwargs.get("y") is not None:
            out = out + 0.1 * kwargs["y"].mean(dim=0).view(1, 1, f, *lat.shape[3:])
        if kwargs.get("vace_context") is not None:
            out = out + 0.05 * kwargs["vace_context"][0][:16].unsqueeze(0)
        if torch.is_tensor(kwargs.get("t")) and kwargs["t"].numel() > 1:       # per-frame timesteps (ti2v): must arrive sliced to the window
            assert kwargs["t"].numel() == f, (kwargs["t"].numel(), f)
            out = out + 0.001 * kwargs["t"].to(out.dtype).view(1, 1, f, 1, 1)
        return out
    return fn


def cases():
    return [dict(name="w17o5_f21", size=17, overlap=5, lat=21), dict(name="w33o9_f21", size=33, overlap=9, lat=21),
            dict(name="w9o0_f11", size=9, overlap=0, lat=11), dict(name="w81o16_f21", size=81, overlap=16, lat=21),
            dict(name="w13o20_f9", size=13, overlap=20, lat=9),
            # reference-image prefix (VACE, ref_images_before): `prefix` extra latent frames in front of every window; per-frame t (ti2v)
            dict(name="w17o5_f21_p2", size=17, overlap=5, lat=21, prefix=2), dict(name="w9o3_f11_p1_t", size=9, overlap=3, lat=11, prefix=1, t=True),
            dict(name="w17o5_f21_t", size=17, overlap=5, lat=21, t=True)]


def make_inputs(lat, prefix=0, per_frame_t=False):
    g = torch.Generator().manual_seed(100 + lat + 1000 * prefix)
    h, w = 4, 6
    lat = lat + prefix                      # the prefix frames lead every time axis (target_shape[1] = lat_frames + prefix)
    latents = torch.randn(1, 16, lat, h, w, generator=g)
    tok = (h // 2) * (w // 2)
    cos = torch.randn(lat * tok, 128, generator=g); sin = torch.randn(lat * tok, 128, generator=g)
    y = torch.randn(20, lat, h, w, generator=g)
    vace = [torch.randn(96, lat, h, w, generator=g)]
    if per_frame_t:
        t = torch.full((lat,), 700, dtype=torch.int64)
        t[:prefix + 2] = 0
        return latents, (cos, sin), y, vace, t
    return latents, (cos, sin), y, vace


def main():
    import types
    outer = build_outer()
    me = types.SimpleNamespace(vae_stride=(4, 8, 8))
    out = {}
    for c in cases():
        P = c.get("prefix", 0)
        latents, freqs, y, vace, *tt = make_inputs(c["lat"], P, c.get("t", False))
        kwargs = {"freqs": freqs, "y": y, "vace_context": vace, "other": 3}
        if tt:
            kwargs["t"] = tt[0]
        r = outer(me, c["size"], c["overlap"], c["lat"], (16, c["lat"] + P, 4, 6), kwargs, P > 0, P, False, None)
        out[c["name"] + "_windows"] = np.array(r["windows"] if r["windows"] is not None else [], dtype=np.int64).reshape(-1, 2)
        out[c["name"] + "_counts"] = np.array([r["window"], r["overlap"]])
        if r["windows"] is not None:
            pred = r["denoise"](latents.clone(), fake_denoise_factory(kwargs))
            assert kwargs["freqs"] is freqs and kwargs["y"] is y and kwargs["other"] == 3          # restored
            out[c["name"] + "_pred"] = pred.numpy()
    for total, size, ov in ((21, 5, 2), (21, 5, 4), (7, 3, 0), (10, 10, 3), (10, 4, 9), (3, 2, 1)):
        w = r["build"](total, size, ov)
        out[f"build_{total}_{size}_{ov}"] = np.array(w if w is not None else [], dtype=np.int64).reshape(-1, 2)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: (v.tolist() if v.size <= 12 else v.shape) for k, v in out.items()})


if __name__ == "__main__":
    main()
