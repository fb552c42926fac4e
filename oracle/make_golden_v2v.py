"""TEST INFRASTRUCTURE ONLY -- tests/golden/v2v.npz from the REFERENCE's own video-to-video statements of `WanAny2V.generate`
(models/wan/any2video.py): the set-up block behind `source_latents = self.vae.encode([input_frames])` (`injection_denoising_step
= 0` ... `masked_steps = math.ceil(...)`, ~:1007-1042), the per-step re-injection of the noised source (`if denoising_strength < 1
and i <= injection_denoising_step:` ..., ~:1504-1515) and the masked merge behind the scheduler step (`if image_mask_latents is not
None and i< masked_steps:` ..., ~:1737-1740).  The statements sit inside `generate`; the source lines are taken verbatim, dedented
and executed inside synthetic functions that supply the enclosing variables.
Run in the build container:   python oracle/make_golden_v2v.py"""
import math
import os
import sys
import textwrap
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get("WAN_REFERENCE_ROOT", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden", "v2v.npz")


def _lines():
    return open(os.path.join(REF, "models/wan/any2video.py")).read().split("\n")


def _block(first, last_startswith, after=0):
    lines = _lines()
    a = next(i for i, l in enumerate(lines) if i >= after and l.strip() == first)
    b = next(i for i, l in enumerate(lines) if i >= a and l.strip().startswith(last_startswith))
    return textwrap.dedent("\n".join(lines[a:b + 1])), (a + 1, b + 1)


def build():
    ns = {"torch": torch, "math": math}
    setup, r1 = _block("injection_denoising_step = 0", "masked_steps = math.ceil(sampling_steps * masking_strength)")
    code = ("def setup(self, input_frames, input_masks, source_latents, lat_frames, sampling_steps, denoising_strength, masking_strength, "
            "keep_frames_parsed, prefix_frames_count, timesteps, sample_scheduler, video_prompt_type, image_outputs):\n"
            "    start_step_no = 0\n    latent_keep_frames = []\n    image_mask_latents = None\n    masked_steps = 0\n"
            + textwrap.indent(setup, "    ") +
            "\n    return dict(injection_denoising_step=injection_denoising_step, inject_from_start=inject_from_start, "
            "latent_keep_frames=latent_keep_frames, timesteps=timesteps, start_step_no=start_step_no, "
            "image_mask_latents=image_mask_latents, masked_steps=masked_steps)\n")
    exec(compile(code, "any2video_v2v_setup_lifted.py", "exec"), ns)
    inject, r2 = _block("if denoising_strength < 1 and i <= injection_denoising_step:", "latents[...] = randn * sigma + (1 - sigma) * source_latents")
    code = ("def inject(latents, randn, source_latents, t, i, denoising_strength, injection_denoising_step, inject_from_start, latent_keep_frames):\n"
            + textwrap.indent(inject, "    ") + "\n    return latents\n")
    exec(compile(code, "any2video_v2v_inject_lifted.py", "exec"), ns)
    merge, r3 = _block("if image_mask_latents is not None and i< masked_steps:", "latents[:, :, :source_latents.shape[2]] = noisy_image * (1-image_mask_latents)")
    code = ("def merge(latents, randn, source_latents, image_mask_latents, timesteps, i, masked_steps):\n"
            + textwrap.indent(merge, "    ") + "\n    return latents\n")
    exec(compile(code, "any2video_v2v_merge_lifted.py", "exec"), ns)
    return ns, (r1, r2, r3)


def cases():
    """name, source latent frames, clip latent frames, steps, denoising strength, masking strength, keep list, prefix frames, mask?"""
    return [dict(name="full_clip", src=5, lat=5, steps=10, ds=0.6, ms=0.5, keep=[], prefix=0, mask=False),
            dict(name="short_source", src=3, lat=5, steps=8, ds=0.75, ms=1.0, keep=[], prefix=0, mask=True),
            dict(name="keep_list", src=5, lat=5, steps=10, ds=0.5, ms=0.7, keep=[True] * 9 + [False] * 8, prefix=0, mask=True),
            dict(name="keep_with_prefix", src=5, lat=5, steps=6, ds=0.34, ms=0.3, keep=[True, False, True, True] * 3, prefix=5, mask=False),
            dict(name="one_frame_mask", src=4, lat=4, steps=5, ds=0.8, ms=1.0, keep=[], prefix=0, mask="one")]


def inputs(c, seed):
    g = torch.Generator().manual_seed(seed)
    T = (c["src"] - 1) * 4 + 1
    frames = torch.rand(3, T, 32, 48, generator=g) * 2 - 1
    src = torch.randn(1, 16, c["src"], 4, 6, generator=g)
    randn = torch.randn(1, 16, c["lat"], 4, 6, generator=g)
    lat = torch.randn(1, 16, c["lat"], 4, 6, generator=g)
    if c["mask"] == "one":
        masks = (torch.rand(1, 1, 32, 48, generator=g) > 0.5).float()
    elif c["mask"]:
        masks = (torch.rand(1, T, 32, 48, generator=g) > 0.5).float()
    else:
        masks = None
    ts = torch.linspace(999, 40, c["steps"]).round()
    return frames, masks, src, randn, lat, ts


def main():
    ns, ranges = build()
    me = types.SimpleNamespace(device="cpu")
    out = {"line_ranges": np.array(ranges)}
    for n, c in enumerate(cases()):
        frames, masks, src, randn, lat, ts = inputs(c, 90 + n)
        sched = types.SimpleNamespace(timesteps=ts.clone(), sigmas=torch.cat([ts / 1000, torch.zeros(1)]))
        st = ns["setup"](me, frames, masks, src, c["lat"], c["steps"], c["ds"], c["ms"], list(c["keep"]), c["prefix"], ts.clone(), sched,
                         "G", False)
        p = c["name"] + "_"
        out[p + "ints"] = np.array([st["injection_denoising_step"], int(st["inject_from_start"]), st["start_step_no"], st["masked_steps"]])
        out[p + "keep"] = np.array([int(v) for v in st["latent_keep_frames"]], dtype=np.int64)
        out[p + "timesteps"] = st["timesteps"].numpy()
        out[p + "sched_timesteps"], out[p + "sched_sigmas"] = sched.timesteps.numpy(), sched.sigmas.numpy()
        if st["image_mask_latents"] is not None:
            out[p + "mask_latents"] = st["image_mask_latents"].numpy()
        x = lat.clone()
        for i, t in enumerate(st["timesteps"]):
            x = ns["inject"](x, randn, src, t, i, c["ds"], st["injection_denoising_step"], st["inject_from_start"], st["latent_keep_frames"])
            out[p + f"inj_{i}"] = x.numpy().copy()
            x = x + 0.1 * torch.roll(x, 1, dims=-1)                       # stands in for the model + scheduler step
            x = ns["merge"](x, randn, src, st["image_mask_latents"], st["timesteps"], i, st["masked_steps"])
            out[p + f"mrg_{i}"] = x.numpy().copy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, len(out), "arrays; lines", ranges)


if __name__ == "__main__":
    main()
