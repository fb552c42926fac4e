"""TEST INFRASTRUCTURE ONLY -- tests/golden/vace_context.npz from the REFERENCE's own `WanAny2V.vace_encode_frames` /
`vace_encode_masks` (models/wan/any2video.py:270-331), executed unmodified: the two method definitions are lifted out of the
class with `ast` (any2video.py imports half of the application) and called with a stand-in `self` that carries a
deterministic fake VAE (average-pool "encoder": the methods only concatenate / rearrange what it returns) and the Wan2.1
stride (4, 8, 8).  Run in the build container:   python oracle/make_golden_vace_context.py"""
import ast
import os
import types

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("WAN_REFERENCE_ROOT", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden", "vace_context.npz")


class FakeVAE:
    """[3,T,H,W] -> [16,(T-1)/4+1,H/8,W/8]: channel mix of an average-pooled video; deterministic, cheap, shape-faithful."""

    def encode(self, videos, tile_size=0):
        out = []
        for v in videos:
            t = (v.shape[1] - 1) // 4 + 1
            p = F.adaptive_avg_pool3d(v.unsqueeze(0), (t, v.shape[2] // 8, v.shape[3] // 8))[0]
            mix = torch.linspace(-1, 1, 48).view(16, 3)
            out.append(torch.einsum("oc,cthw->othw", mix, p))
        return out


def lift_methods(names):
    src = open(os.path.join(REF, "models/wan/any2video.py")).read()
    tree = ast.parse(src)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "WanAny2V")
    picked = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert {n.name for n in picked} == set(names)
    ns = {"torch": torch, "F": F}
    exec(compile(ast.Module(body=picked, type_ignores=[]), "any2video.py", "exec"), ns)
    return ns


def inputs():
    g = torch.Generator().manual_seed(77)
    frames = torch.rand(3, 9, 32, 48, generator=g) * 2 - 1
    mask = (torch.rand(1, 9, 32, 48, generator=g) > 0.6).float()
    refs = [torch.rand(3, 1, 32, 48, generator=g) * 2 - 1 for _ in range(2)]
    return frames, mask, refs


def main():
    ns = lift_methods(["vace_encode_frames", "vace_encode_masks"])
    me = types.SimpleNamespace(vae=FakeVAE(), vae_stride=(4, 8, 8))
    frames, mask, refs = inputs()
    out = {}
    z = ns["vace_encode_frames"](me, [frames], None, masks=[mask])
    m = ns["vace_encode_masks"](me, [mask], None)
    out["z_noref"], out["m_noref"] = z[0].numpy(), m[0].numpy()
    z = ns["vace_encode_frames"](me, [frames], refs, masks=[mask])
    m = ns["vace_encode_masks"](me, [mask], refs)
    out["z_ref"], out["m_ref"] = z[0].numpy(), m[0].numpy()
    # the background-mask variant of the first reference image: the composition of any2video.py:1138-1145 around the same two functions
    bgm = (torch.rand(1, 1, 32, 48, generator=torch.Generator().manual_seed(78)) > 0.5).float()
    ref_masks = [bgm, None]
    z0 = ns["vace_encode_frames"](me, [frames], refs, masks=[mask])
    m0 = ns["vace_encode_masks"](me, [mask], refs)
    zbg = ns["vace_encode_frames"](me, refs[:1] * 1, None, masks=ref_masks[0])
    mbg = ns["vace_encode_masks"](me, ref_masks[:1] * 1, None)
    for zz0, mm0, zzbg, mmbg in zip(z0, m0, zbg, mbg):
        zz0[:, 0:1] = zzbg
        mm0[:, 0:1] = mmbg
    out["bg_mask"], out["z_bg"], out["m_bg"] = bgm.numpy(), z0[0].numpy(), m0[0].numpy()
    out["z_nomask"] = ns["vace_encode_frames"](me, [frames], None, masks=None)[0].numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
