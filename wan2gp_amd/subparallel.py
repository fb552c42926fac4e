"""Sub-parallel temporal windows of the sampler loop (models/wan/any2video.py:1199-1387): for long videos the reference can run
every denoising step on overlapping windows of latent frames instead of the whole clip -- attention cost grows with the
window, not the clip -- and blend the per-window predictions with linear ramps over the overlap.  Plain t2v / i2v / VACE
case: no reference-image prefix, no history latents, none of the variant tensors.

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


def slice_kwargs(kwargs: Dict, start: int, end: int, lat_frames: int, tokens_per_frame: int) -> Dict:
    """`_sub_parallel_kwargs` (:1296-1317) for freqs / y / vace_context."""
    out = {}
    freqs = kwargs.get("freqs")
    if isinstance(freqs, tuple):
        idx = (torch.arange(start, end, device=freqs[0].device)[:, None] * tokens_per_frame
               + torch.arange(tokens_per_frame, device=freqs[0].device)).reshape(-1)
        out["freqs"] = (freqs[0].index_select(0, idx), freqs[1].index_select(0, idx))
    elif "freqs" in kwargs:
        out["freqs"] = freqs
    y = kwargs.get("y")
    if torch.is_tensor(y) and y.ndim > 1 and y.shape[1] == lat_frames:
        out["y"] = y.narrow(1, start, end - start)
    if "vace_context" in kwargs and kwargs["vace_context"] is not None:
        out["vace_context"] = [u.narrow(1, start, end - start) if torch.is_tensor(u) and u.ndim > 1 and u.shape[1] == lat_frames else u
                               for u in kwargs["vace_context"]]
    return out


def denoise(latents: torch.Tensor, denoise_fn: Callable, windows, overlap: int, kwargs: Dict, tokens_per_frame: int):
    """`_sub_parallel_denoise` (:1328-1387): denoise_fn(latent_window) reads `kwargs` (updated in place for the duration of
    the call, restored afterwards).  Returns the blended prediction, or None if a window was interrupted."""
    lat_frames = latents.shape[2]
    pred_sum = torch.zeros_like(latents)
    w_sum = torch.zeros(1, 1, lat_frames, 1, 1, dtype=latents.dtype, device=latents.device)
    for start, end in windows:
        anchor = 1 if start > 0 else 0
        c0 = start - anchor
        saved = dict(kwargs)
        try:
            kwargs.update(slice_kwargs(kwargs, c0, end, lat_frames, tokens_per_frame))
            pred = denoise_fn(latents[:, :, c0:end])
        finally:
            kwargs.clear()
            kwargs.update(saved)
        if pred is None:
            return None
        w = window_weight(start, end, overlap, lat_frames, pred.dtype, pred.device)
        part = pred[:, :, anchor:anchor + end - start]
        part.mul_(w)
        pred_sum[:, :, start:end] += part
        w_sum[:, :, start:end] += w
    pred_sum.div_(w_sum.clamp_min_(1e-6))
    return pred_sum
