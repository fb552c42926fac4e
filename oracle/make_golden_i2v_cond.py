"""TEST INFRASTRUCTURE ONLY -- tests/golden/i2v_cond.npz from the REFERENCE's own i2v conditioning statements
(models/wan/any2video.py: from `remaining_frames = frame_num - control_pre_frames_count` to
`extended_overlapped_latents = lat_y[...]`, ~:699-783).  The statements sit inside `WanAny2V.generate`; the source lines are
taken verbatim, dedented, and executed inside a synthetic function that supplies the enclosing variables of the plain
i2v2_2 case (no end frame, no svi / infinitetalk modes) and a deterministic stand-in VAE.
Run in the build container:   python oracle/make_golden_i2v_cond.py"""
import os
import sys
import textwrap
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.make_golden_vace_context import FakeVAE as _PoolVAE  # noqa: E402

REF = os.environ.get("WAN_REFERENCE_ROOT", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden", "i2v_cond.npz")


class FakeVAE(_PoolVAE):
    def encode(self, videos, tile_size=0, any_end_frame=False):
        if not any_end_frame:
            return super().encode(videos, tile_size)
        # the stand-in of WanVAE_.encode(any_end_frame=True) (vae.py:590-606): first frame, groups of four, the last frame alone
        out = []
        for v in videos:
            body, last = super().encode([v[:, :1 + 4 * ((v.shape[1] - 2) // 4)], v[:, -1:]], tile_size)
            out.append(torch.cat([body, last], dim=1))
        return out


def build_block():
    lines = open(os.path.join(REF, "models/wan/any2video.py")).read().split("\n")
    a = next(i for i, l in enumerate(lines) if l.strip() == "remaining_frames = frame_num - control_pre_frames_count")
    b = next(i for i, l in enumerate(lines) if i > a and l.strip().startswith("extended_overlapped_latents = lat_y[:, :overlapped_latents_frames_num]"))
    block = textwrap.dedent("\n".join(lines[a:b + 1]))
    params = "self, control_video, frame_num, height, width, lat_h, lat_w, control_pre_frames_count, preframes_count, motion_amplitude, VAE_tile_size"
    pre = ("any_end_frame = add_frames_for_end_image = svi_pro = svi_mode = infinitetalk = False\nsvi_ref_pad_num = 0\n"
           "ref_images_count = 0\nuse_extended_overlapped_latents = True\nextended_overlapped_latents = None\nkwargs = {}\n"
           "lat_frames = (frame_num - 1) // 4 + 1\n")
    post = "return y, extended_overlapped_latents\n"
    code = "def block(" + params + "):\n" + textwrap.indent(pre + block + "\n" + post, "    ")
    ns = {"torch": torch}
    exec(compile(code, "any2video_i2v_cond_lifted.py", "exec"), ns)
    return ns["block"], (a + 1, b + 1)


def build_block_end():
    """The same statements with an end image (any2video.py:684-692 supply `any_end_frame`, `add_frames_for_end_image` -- the Wan2.1
    i2v model gets one extra frame / latent frame -- and `img_end_frame`; the lifted block takes its `if any_end_frame` arms)."""
    lines = open(os.path.join(REF, "models/wan/any2video.py")).read().split("\n")
    a = next(i for i, l in enumerate(lines) if l.strip() == "remaining_frames = frame_num - control_pre_frames_count")
    b = next(i for i, l in enumerate(lines) if i > a and l.strip().startswith("extended_overlapped_latents = lat_y[:, :overlapped_latents_frames_num]"))
    block = textwrap.dedent("\n".join(lines[a:b + 1]))
    params = ("self, control_video, frame_num, height, width, lat_h, lat_w, control_pre_frames_count, preframes_count, motion_amplitude, "
              "VAE_tile_size, img_end_frame, add_frames_for_end_image")
    pre = ("any_end_frame = True\nsvi_pro = svi_mode = infinitetalk = False\nsvi_ref_pad_num = 0\n"
           "ref_images_count = 0\nuse_extended_overlapped_latents = True\nextended_overlapped_latents = None\nkwargs = {}\n"
           "lat_frames = (frame_num - 1) // 4 + 1\n"
           "if add_frames_for_end_image:\n    frame_num += 1\n    lat_frames = int((frame_num - 2) // 4 + 2)\n")     # :688-691
    post = "return y, extended_overlapped_latents\n"
    code = "def block(" + params + "):\n" + textwrap.indent(pre + block + "\n" + post, "    ")
    ns = {"torch": torch}
    exec(compile(code, "any2video_i2v_cond_end_lifted.py", "exec"), ns)
    return ns["block"], (a + 1, b + 1)


def end_cases():
    return [dict(name="end22", P=1, frames=17, amp=1.0, add=False), dict(name="end21", P=1, frames=17, amp=1.0, add=True),
            dict(name="end22_amp", P=1, frames=13, amp=1.3, add=False), dict(name="end21_video5_amp", P=5, frames=21, amp=1.2, add=True)]


def cases():
    return [dict(name="image", P=1, frames=17, amp=1.0), dict(name="video5", P=5, frames=17, amp=1.0),
            dict(name="image_amp", P=1, frames=13, amp=1.4), dict(name="video9_amp", P=9, frames=21, amp=1.2)]


def make_video(P, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(3, P, 32, 48, generator=g) * 2 - 1


def main():
    block, rng = build_block()
    me = types.SimpleNamespace(vae=FakeVAE(), device="cpu", VAE_dtype=torch.float32)
    out = {"line_range": np.array(rng)}
    for i, c in enumerate(cases()):
        v = make_video(c["P"], 40 + i)
        y, ext = block(me, v, c["frames"], 32, 48, 4, 6, c["P"], c["P"], c["amp"], 0)
        out[c["name"] + "_y"], out[c["name"] + "_ext"] = y.numpy(), ext.numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items()}, "lines", rng)
    block, rng = build_block_end()
    out = {"line_range": np.array(rng)}
    for i, c in enumerate(end_cases()):
        v, e = make_video(c["P"], 60 + i), make_video(1, 80 + i)
        y, ext = block(me, v, c["frames"], 32, 48, 4, 6, c["P"], c["P"], c["amp"], 0, e, c["add"])
        out[c["name"] + "_y"], out[c["name"] + "_ext"] = y.numpy(), ext.numpy()
    np.savez_compressed(OUT.replace("i2v_cond.npz", "i2v_cond_end.npz"), **out)
    print("wrote i2v_cond_end.npz", {k: v.shape for k, v in out.items()}, "lines", rng)


if __name__ == "__main__":
    main()
