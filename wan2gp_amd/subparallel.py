"""Sub-parallel temporal windows of the sampler loop (models/wan/any2video.py:1199-1387): for long videos the reference can run
every denoising step on overlapping windows of latent frames instead of the whole clip -- attention cost grows with the
window, not the clip -- and blend the per-window predictions with linear ramps over the overlap.  Plain t2v / i2v / VACE
case plus VACE's reference-image prefix (extra latent frames that lead every window) and per-frame timesteps (ti2v); no history
latents, none of the other variants' tensors.

Each window [start, end) is run with one extra *anchor* latent frame in front when start > 0 (context only, its prediction is
dropped); the model keywords that carry a latent-frame axis are sliced to the window: RoPE tables (token rows of the window's
frames), `y` (dim 1), `vace_context` (dim 1 of each item).  Step-skipping caches are switched off while windows are active
(:1392-1397), as in the reference.
"""
from typing import Callable, Dict, List, Optional, Tuple

import torch


def window_latent_counts(window_size: int, window_overlap: int, lat_frames: int, vae_stride_t: int = 4) -> Tuple[int, int]:
    """Pixel-frame window size / overlap -> latent-frame counts (any2video.py:1215-1220)."""
    window_size, window_overlap = max(0, int(window_size or 0)), max(0, int(window_overlap or 0))
    win = min(lat_frames, max(1, int((window_size - 1) // vae_stride_t) + 1)) if window_size > 0 else 0
    ov = max(0, int((window_overlap - 1) // vae_stride_t) + 1) if window_overlap > 0 else 0
    if win > 0:
        ov = min(ov, win - 1)
    return win, ov


def build_windows(total: int, size: int, overlap: int) -> Optional[List[Tuple[int, int]]]:
    """`_build_sub_parallel_windows` (any2video.py:1199-1213): None when one window would cover everything."""
    if size <= 0 or size >= total:
        return None
    overlap = min(max(0, overlap), size - 1)
    out, start = [], 0
    while True:
        end = start + size
        if end >= total:
            start = max(0, total - size)
            if not out or out[-1][0] != start:
                out.append((start, total))
            return out
        out.append((start, end))
        start += size - overlap


def window_weight(start, end, overlap, lat_frames, dtype, device):
    """`_sub_parallel_weight` (:1319-1326): 1 inside, a 1e-6..1 ramp over the overlap at every interior edge."""
    w = torch.ones(end - start, dtype=dtype, device=device)
    ramp = min(overlap, end - start)
    if ramp > 0 and start > 0:
        w[:ramp] = torch.linspace(1e-6, 1, ramp, dtype=dtype, device=device)
    if ramp > 0 and end < lat_frames:
        w[-ramp:] = torch.linspace(1, 1e-6, ramp, dtype=dtype, device=device)
    return w.view(1, 1, -1, 1, 1)


def _slice_time(u, dim: int, start: int, end: int, lat_frames: int, prefix: int, include_prefix: bool):
    """`_sub_parallel_slice_time` (:1240-1255, no history latents): a tensor whose `dim` spans prefix + clip keeps the prefix frames and
    the window's; one that spans the clip alone is narrowed; anything else passes through."""
    if not torch.is_tensor(u) or u.ndim <= dim:
        return u
    if include_prefix and u.shape[dim] == lat_frames + prefix:
        win = u.narrow(dim, prefix + start, end - start)
        return torch.cat([u.narrow(dim, 0, prefix), win], dim=dim) if prefix > 0 else win
    if u.shape[dim] == lat_frames:
        return u.narrow(dim, start, end - start)
    return u


def slice_kwargs(kwargs: Dict, start: int, end: int, lat_frames: int, tokens_per_frame: int, prefix: int = 0) -> Dict:
    """`_sub_parallel_kwargs` (:1301-1323) for freqs / t / y / vace_context; `prefix` = reference-image latent frames that lead the
    clip (`sub_parallel_prefix_latents`, :1222) and stay in front of every window."""
    out = {}
    freqs = kwargs.get("freqs")
    if isinstance(freqs, tuple):
        dev = freqs[0].device
        frames = torch.arange(prefix + start, prefix + end, device=dev)                               # _sub_parallel_model_indices
        if prefix > 0:
            frames = torch.cat([torch.arange(prefix, device=dev), frames])
        idx = [(frames[:, None] * tokens_per_frame + torch.arange(tokens_per_frame, device=dev)).reshape(-1)]
        main_count = (lat_frames + prefix) * tokens_per_frame
        if freqs[0].shape[0] > main_count:                                                            # rows behind the clip's stay (:1296-1297)
            idx.append(torch.arange(main_count, freqs[0].shape[0], device=dev))
        idx = torch.cat(idx)
        out["freqs"] = (freqs[0].index_select(0, idx), freqs[1].index_select(0, idx))
    elif "freqs" in kwargs:
        out["freqs"] = freqs
    if "t" in kwargs:                                         # a per-frame timestep vector (ti2v) follows the window; [1] passes through
        out["t"] = _slice_time(kwargs["t"], 0, start, end, lat_frames, prefix, True)
    if kwargs.get("y") is not None:
        out["y"] = _slice_time(kwargs["y"], 1, start, end, lat_frames, prefix, True)
    if "vace_context" in kwargs and kwargs["vace_context"] is not None:
        out["vace_context"] = [_slice_time(u, 1, start, end, lat_frames, prefix, True) for u in kwargs["vace_context"]]
    return out


def denoise(latents: torch.Tensor, denoise_fn: Callable, windows, overlap: int, kwargs: Dict, tokens_per_frame: int, prefix: int = 0):
    """`_sub_parallel_denoise` (:1334-1393): denoise_fn(latent_window) reads `kwargs` (updated in place for the duration of
    the call, restored afterwards).  With `prefix` reference-image frames in front, every window carries them and their
    predictions are averaged over the windows.  Returns the blended prediction, or None if a window was interrupted."""
    lat_frames = latents.shape[2] - prefix
    pred_sum = torch.zeros_like(latents)
    w_sum = torch.zeros(1, 1, latents.shape[2], 1, 1, dtype=latents.dtype, device=latents.device)
    for start, end in windows:
        anchor = 1 if start > 0 else 0
        c0 = start - anchor
        window = latents[:, :, prefix + c0:prefix + end]
        if prefix > 0:
            window = torch.cat([latents[:, :, :prefix], window], dim=2)
        saved = dict(kwargs)
        try:
            kwargs.update(slice_kwargs(kwargs, c0, end, lat_frames, tokens_per_frame, prefix))
            pred = denoise_fn(window)
        finally:
            kwargs.clear()
            kwargs.update(saved)
        if pred is None:
            return None
        p0 = prefix if prefix > 0 and pred.shape[2] == prefix + end - c0 else 0
        if p0 > 0:
            pred_sum[:, :, :prefix] += pred[:, :, :prefix]
            w_sum[:, :, :prefix] += 1
        w = window_weight(start, end, overlap, lat_frames, pred.dtype, pred.device)
        part = pred[:, :, p0 + anchor:p0 + anchor + end - start]
        part.mul_(w)
        pred_sum[:, :, prefix + start:prefix + end] += part
        w_sum[:, :, prefix + start:prefix + end] += w
    pred_sum.div_(w_sum.clamp_min_(1e-6))
    return pred_sum
