"""Video-to-video inside the sampler loop: denoising strength, kept frames and masked regeneration.

Mirrors the three fragments of `WanAny2V.generate` (models/wan/any2video.py) that implement the "G" video prompt type -- start
from the VAE latents of a source video instead of pure noise:

  * `plan`   (any2video.py:1007-1042)  how many of the first sampler steps are replaced by the noised source
             (`sampling_steps * (1 - denoising_strength)`); when the source covers the whole clip and every frame is regenerated the
             schedule is simply cut short, otherwise the noised source is re-injected in front of each of those steps, except for
             the latent frames whose four video frames are not all marked "keep"; the pixel mask reduced to the latent grid and the
             number of steps (`sampling_steps * masking_strength`) during which unmasked regions are pinned to the source.
  * `inject` (:1504-1515)              latents <- randn * sigma + (1 - sigma) * source in front of a step, sigma = t / 1000.
  * `merge`  (:1737-1740)              behind the scheduler step: outside the mask the latents are the source noised to the NEXT
             step's sigma.

Latent-sized torch arithmetic on the device the latents live on (a few elementwise passes over 19 MB per step); every function is
pinned to the reference's own statements by tests/test_v2v_vs_golden.py (oracle/make_golden_v2v.py lifts them verbatim).
"""
import math
from typing import List, Optional

import torch


class V2VPlan:
    __slots__ = ("injection_denoising_step", "inject_from_start", "latent_keep_frames", "timesteps", "start_step_no",
                 "image_mask_latents", "masked_steps")


def plan(input_frames: torch.Tensor, input_masks: Optional[torch.Tensor], source_latents: torch.Tensor, lat_frames: int,
         sampling_steps: int, denoising_strength: float, masking_strength: float, keep_frames_parsed: List[bool],
         prefix_frames_count: int, timesteps: torch.Tensor, sample_scheduler, image_outputs: bool = False, device=None,
         video_prompt_type: str = "G") -> V2VPlan:
    """any2video.py:1007-1042.  `source_latents` [1,C,f,h,w] = vae.encode([input_frames]); `input_masks` [1,T,H,W] (1 = regenerate)
    or None.  May cut `timesteps` and the scheduler's `timesteps` / `sigmas` short (the schedule then starts at the injection
    step: `start_step_no`)."""
    p = V2VPlan()
    p.injection_denoising_step, p.inject_from_start, p.latent_keep_frames = 0, False, []
    p.timesteps, p.start_step_no, p.image_mask_latents, p.masked_steps = timesteps, 0, None, 0
    keep = list(keep_frames_parsed)
    if denoising_strength < 1:
        overlapped = prefix_frames_count if prefix_frames_count > 0 else 0
        if len(keep) == 0 or image_outputs or (overlapped + len(keep)) == input_frames.shape[1] and all(keep):
            keep = []                                                                   # :1018 (`and` binds tighter than `or`)
        p.injection_denoising_step = int(round(sampling_steps * (1. - denoising_strength), 4))
        if source_latents.shape[2] < lat_frames or len(keep) > 0:
            p.inject_from_start = True
            if len(keep) > 0:
                if overlapped > 0:
                    keep = [True] * overlapped + keep
                p.latent_keep_frames = [keep[0]] + [all(keep[i:i + 4]) for i in range(1, len(keep), 4)]
        else:                                                                           # the schedule is simply cut short (:1029-1033)
            p.timesteps = timesteps[p.injection_denoising_step:]
            p.start_step_no = p.injection_denoising_step
            if hasattr(sample_scheduler, "timesteps"):
                sample_scheduler.timesteps = p.timesteps
            if hasattr(sample_scheduler, "sigmas"):
                sample_scheduler.sigmas = sample_scheduler.sigmas[p.injection_denoising_step:]
            p.injection_denoising_step = 0
    if input_masks is not None and "U" not in video_prompt_type:                        # :1035-1042
        m = torch.nn.functional.interpolate(input_masks, size=source_latents.shape[-2:], mode="nearest").unsqueeze(0)
        if m.shape[2] != 1:
            m = torch.cat([m[:, :, :1], torch.nn.functional.interpolate(m, size=(source_latents.shape[-3] - 1, *source_latents.shape[-2:]),
                                                                       mode="nearest")], dim=2)
        p.image_mask_latents = torch.where(m >= 0.5, 1., 0.)[:1].to(device if device is not None else source_latents.device)
        p.masked_steps = math.ceil(sampling_steps * masking_strength)
    return p


def inject(latents: torch.Tensor, randn: torch.Tensor, source_latents: torch.Tensor, t, i: int, denoising_strength: float,
           p: V2VPlan) -> torch.Tensor:
    """any2video.py:1504-1515, in front of step i (timestep t)."""
    if not (denoising_strength < 1 and i <= p.injection_denoising_step):
        return latents
    sigma = t / 1000
    n = source_latents.shape[2]
    if p.inject_from_start:
        noisy = latents.clone()
        noisy[:, :, :n] = randn[:, :, :n] * sigma + (1 - sigma) * source_latents
        for k, keep in enumerate(p.latent_keep_frames):
            if not keep:
                noisy[:, :, k:k + 1] = latents[:, :, k:k + 1]
        return noisy
    latents[...] = randn * sigma + (1 - sigma) * source_latents
    return latents


def merge(latents: torch.Tensor, randn: torch.Tensor, source_latents: torch.Tensor, timesteps: torch.Tensor, i: int,
          p: V2VPlan) -> torch.Tensor:
    """any2video.py:1737-1740, behind the scheduler step of step i: where the mask is 0 the result is the source at the next
    step's noise level."""
    if p.image_mask_latents is None or i >= p.masked_steps:
        return latents
    n = source_latents.shape[2]
    sigma = 0 if i == len(timesteps) - 1 else timesteps[i + 1] / 1000
    noisy = randn[:, :, :n] * sigma + (1 - sigma) * source_latents
    latents[:, :, :n] = noisy * (1 - p.image_mask_latents) + p.image_mask_latents * latents[:, :, :n]
    return latents
