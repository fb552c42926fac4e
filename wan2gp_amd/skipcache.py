"""Step-skipping caches of the reference (TeaCache / MagCache) for the resident HIP model -- SURVEY.md section 8(f) rank 4.

Reference: the cache object hangs on `WanModel.cache` (an attribute bag built by `models/wan/wan_handler.py:160-215` with
the per-model calibration data -- `coefficients` for TeaCache, `def_mag_ratios` for MagCache -- and `start_step`,
`multiplier`, `num_steps`, `cache_type`); `WanAny2V.generate` resets it and picks the threshold that meets the requested
speed-up (`any2video.py:1398-1408`; `WanModel.compute_teacache_threshold` / `compute_magcache_threshold`,
`models/wan/modules/model.py:1373-1482`); `WanModel.forward` decides per call whether the block chain runs or the previous
residual is re-applied (`model.py:1914-2064`).

Here the decisions are the same host arithmetic (numpy / CPU torch on the [1, dim] time embedding -- the reference itself
does `.cpu().item()` there), and the residual bookkeeping runs in `wan_dit_forward_skip`: a computing stream leaves
`x_after_blocks - x_after_patch_embed` (bf16) in its residual buffer, a skipped stream gets `patch_embed(x) + residual` and
goes straight to the head.  The residual buffers stay in HBM (0.77 GB per stream at 14B-720p).
"""
from typing import List, Optional

import numpy as np
import torch


class SkipStepsCache:
    """Attribute bag with `.update(dict)`, like the reference's `DynamicClass` (wgp.py:6193-6215)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def update(self, d):
        self.__dict__.update(d)
        return self


def _nearest_interp(src, n):
    """model.py:1375-1380."""
    if n == 1:
        return np.array([src[-1]])
    scale = (len(src) - 1) / (n - 1)
    return src[np.round(np.arange(n) * scale).astype(int)]


def compute_magcache_threshold(cache, start_step, timesteps, speed_factor):
    """`WanModel.compute_magcache_threshold` (model.py:1373-1430): resample the calibrated magnitude ratios to the step
    count, then scan thresholds 0.01, 0.02, ... for the one whose simulated number of computed steps is closest to
    len(timesteps) / speed_factor.  Sets cache.mag_ratios and cache.magcache_thresh; returns the threshold."""
    n = len(timesteps)
    ratios = np.array([1.0] * 2 + list(cache.def_mag_ratios))
    if len(ratios) != n * 2:
        con, ucon = _nearest_interp(ratios[0::2], n), _nearest_interp(ratios[1::2], n)
        ratios = np.concatenate([con.reshape(-1, 1), ucon.reshape(-1, 1)], axis=1).reshape(-1)
    cache.mag_ratios = ratios
    target = int(n / speed_factor)
    best_threshold, best_diff, best_signed = 0.01, 1000, 1000
    threshold = 0.01
    while threshold <= 0.6:
        nb, diff, signed = 0, 1000, None
        err, steps, ratio = 0, 0, 1.0
        skip = False
        for i in range(n):
            if i <= start_step:
                skip = False
            else:
                ratio *= ratios[i * 2]
                steps += 1
                err += np.abs(1 - ratio)
                if err < threshold and steps <= cache.magcache_K:
                    skip = True
                else:
                    skip = False
                    err, steps, ratio = 0, 0, 1.0
            if not skip:
                nb += 1
                signed = target - nb
                diff = abs(signed)
        if diff < best_diff:
            best_threshold, best_diff, best_signed = threshold, diff, signed
        elif diff > best_diff:
            break
        threshold += 0.01
    cache.magcache_thresh = best_threshold
    return best_threshold


def _rel_l1(e, prev):
    """((e - prev).abs().mean() / prev.abs().mean()).cpu().item() on the model-dtype tensors (model.py:1458, :1954)."""
    e, prev = e.detach().cpu(), prev.detach().cpu()
    return ((e - prev).abs().mean() / prev.abs().mean()).item()


def compute_teacache_threshold(cache, start_step, e_list: List[torch.Tensor], speed_factor):
    """`WanModel.compute_teacache_threshold` (model.py:1432-1482) given the time embeddings e_i of every timestep."""
    rescale = np.poly1d(cache.coefficients)
    n = len(e_list)
    deltas = [None] + [abs(rescale(_rel_l1(e_list[i], e_list[i - 1]))) for i in range(1, n)]
    target = int(n / speed_factor)
    best_threshold, best_diff, best_signed = 0.01, 1000, 1000
    threshold = 0.01
    while threshold <= 0.6:
        acc, nb, diff, signed = 0, 0, 1000, None
        for i in range(n):
            skip = False
            if not (i <= start_step or i == n - 1):
                acc += deltas[i]
                if acc < threshold:
                    skip = True
                else:
                    acc = 0
            if not skip:
                nb += 1
                signed = target - nb
                diff = abs(signed)
        if diff < best_diff:
            best_threshold, best_diff, best_signed = threshold, diff, signed
        elif diff > best_diff:
            break
        threshold += 0.01
    cache.rel_l1_thresh = best_threshold
    return best_threshold


def reset_for_generation(cache, x_count=2):
    """any2video.py:1398-1408, the part that does not need the model."""
    cache.previous_residual = [None] * x_count
    if cache.cache_type != "tea":
        cache.accumulated_err, cache.accumulated_steps, cache.accumulated_ratio = [0.0] * x_count, [0] * x_count, [1.0] * x_count
        cache.one_for_all = x_count > 2
    cache.skipped_steps = 0                                   # wgp.py:7726-7732 resets it per generation window


def decide(cache, n_streams: int, x_id: int, real_step_no: int, e: Optional[torch.Tensor] = None) -> List[bool]:
    """The should_calc decision of `WanModel.forward` (model.py:1914-1963) for one call: n_streams > 1 is the joint pass.
    Updates the cache's accumulators exactly as the reference does and returns x_should_calc (one flag per stream)."""
    joint = n_streams > 1
    should_calc, x_should_calc = True, None
    if cache.cache_type == "mag":
        if real_step_no <= cache.start_step:
            should_calc = True
        elif cache.one_for_all and x_id != 0:
            assert n_streams == 1
            should_calc = cache.should_calc
        else:
            x_should_calc = []
            for i in range(1 if cache.one_for_all else n_streams):
                cur = i if joint else x_id
                cache.accumulated_ratio[cur] *= cache.mag_ratios[real_step_no * 2 + cur]
                cache.accumulated_steps[cur] += 1
                cache.accumulated_err[cur] += np.abs(1 - cache.accumulated_ratio[cur])
                if cache.accumulated_err[cur] < cache.magcache_thresh and cache.accumulated_steps[cur] <= cache.magcache_K:
                    skip = True
                    if i == 0 and x_id == 0:
                        cache.skipped_steps += 1
                else:
                    skip = False
                    cache.accumulated_err[cur], cache.accumulated_steps[cur], cache.accumulated_ratio[cur] = 0, 0, 1.0
                x_should_calc.append(not skip)
            if cache.one_for_all:
                should_calc = cache.should_calc = x_should_calc[0]
                x_should_calc = None
    else:
        if x_id != 0:
            should_calc = cache.should_calc
        else:
            if real_step_no <= cache.start_step or real_step_no == cache.num_steps - 1 or cache.previous_modulated_input is None:
                should_calc = True
                cache.accumulated_rel_l1_distance = 0
            else:
                delta = abs(np.poly1d(cache.coefficients)(_rel_l1(e, cache.previous_modulated_input)))
                cache.accumulated_rel_l1_distance += delta
                if cache.accumulated_rel_l1_distance < cache.rel_l1_thresh:
                    should_calc = False
                    cache.skipped_steps += 1
                else:
                    should_calc = True
                    cache.accumulated_rel_l1_distance = 0
            cache.previous_modulated_input = e
            cache.should_calc = should_calc
    if x_should_calc is None:
        x_should_calc = [should_calc] * n_streams
    return x_should_calc
