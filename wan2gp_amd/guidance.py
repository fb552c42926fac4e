"""Guidance variants of the t2v branch of the sampler loop (models/wan/any2video.py:1703-1722): CFG-Zero* and adaptive
projected guidance (APG).  fp32 latent-sized tensors between two forwards -- a few reductions and element-wise updates
(9.7 M elements at 720p x 81 frames), written on torch so that they run wherever the noise predictions live; plain CFG stays
on `wan_cfg_combine`.

  optimized_scale              any2video.py:67-79           alpha = <cond, uncond> / (|uncond|^2 + 1e-8) per sample
  MomentumBuffer               multitalk_utils.py:339-347   running = value + momentum * running
  project                      multitalk_utils.py:351-360   parallel / orthogonal parts of v0 w.r.t. v1, in float64
  adaptive_projected_guidance  multitalk_utils.py:362-381   momentum, norm clipping at norm_threshold, projection, eta mix
"""
import torch


def optimized_scale(positive_flat, negative_flat):
    dot = torch.sum(positive_flat * negative_flat, dim=1, keepdim=True)
    sq = torch.sum(negative_flat ** 2, dim=1, keepdim=True) + 1e-8
    return dot / sq


class MomentumBuffer:
    def __init__(self, momentum: float):
        self.momentum = momentum
        self.running_average = 0

    def update(self, update_value: torch.Tensor):
        self.running_average = update_value + self.momentum * self.running_average


def project(v0: torch.Tensor, v1: torch.Tensor):
    dtype = v0.dtype
    v0, v1 = v0.double(), v1.double()
    v1 = torch.nn.functional.normalize(v1, dim=[-1, -2, -3, -4])
    par = (v0 * v1).sum(dim=[-1, -2, -3, -4], keepdim=True) * v1
    return par.to(dtype), (v0 - par).to(dtype)


def adaptive_projected_guidance(diff, pred_cond, momentum_buffer: MomentumBuffer = None, eta: float = 0.0, norm_threshold: float = 55):
    if momentum_buffer is not None:
        momentum_buffer.update(diff)
        diff = momentum_buffer.running_average
    if norm_threshold > 0:
        norm = diff.norm(p=2, dim=[-1, -2, -3, -4], keepdim=True)
        diff = diff * torch.minimum(torch.ones_like(diff), norm_threshold / norm)
    par, orth = project(diff, pred_cond)
    return orth + eta * par


def combine(cond, uncond, guide_scale, step_no, apg_switch=0, cfg_star_switch=0, cfg_zero_step=5, momentum_buffer=None,
            apg_norm_threshold=55):
    """The plain two-stream branch of denoise_with_cfg_fn (any2video.py:1703-1722)."""
    if apg_switch != 0:
        return cond + (guide_scale - 1) * adaptive_projected_guidance(cond - uncond, cond, momentum_buffer=momentum_buffer,
                                                                      norm_threshold=apg_norm_threshold)
    if cfg_star_switch:
        b = cond.shape[0]
        alpha = optimized_scale(cond.view(b, -1), uncond.view(b, -1)).view(b, 1, 1, 1)
        # any2video.py:1717-1722: for step_no <= cfg_zero_step the reference computes `noise_pred = text * 0.` and then
        # OVERWRITES it with the plain CFG line below (which is not nested under the `if`), with an unscaled uncond; later
        # steps scale uncond by alpha in place and fall through to the same line.
        if step_no > cfg_zero_step:
            uncond = uncond * alpha
    return uncond + guide_scale * (cond - uncond)
