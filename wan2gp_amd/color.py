"""Colour matching of a decoded window to a reference frame -- the post-processing step `WanAny2V.generate` applies to every sliding
window after the first (any2video.py:1783, :1799-1808: `match_and_blend_colors`, models/wan/multitalk/multitalk_utils.py:382-477).

Per frame and per CIE-Lab channel the window's mean / standard deviation are moved onto the reference frame's, the result converted
back and blended with the original by `strength`.  The reference converts with scikit-image (`skimage.color.rgb2lab / lab2rgb`,
sRGB, D65 / 2 degree observer); that package is a dependency of the reference, not part of its tree, so the two conversions are
restated here from its published definition (the CIE formulas with scikit-image's constants: the 0.412453 ... sRGB matrix, white point
(0.95047, 1, 1.08883), thresholds 0.008856 / 0.2068966, slope 7.787) -- PARITY UNPINNED against scikit-image itself (not installable
here); pinned instead by CIE reference colours, the round trip and the statistics the function must produce (tests/test_color_cpu.py).
Host-side numpy on the decoded uint8 frames, like the reference's: a few milliseconds per window, not on the hot path."""
import numpy as np
import torch

_XYZ_FROM_RGB = np.array([[0.412453, 0.357580, 0.180423],
                          [0.212671, 0.715160, 0.072169],
                          [0.019334, 0.119193, 0.950227]], dtype=np.float64)
_RGB_FROM_XYZ = np.linalg.inv(_XYZ_FROM_RGB)
_WHITE = np.array([0.95047, 1.0, 1.08883], dtype=np.float64)          # D65, 2 degree observer


def _float(a):
    """scikit-image keeps a float32 image in float32 (`_supported_float_type`) and casts its constants to it; anything else -> float64.
    The reference hands it float32 frames (numpy views of float32 tensors), so that is the arithmetic type of the whole transfer."""
    a = np.asarray(a)
    return a if a.dtype == np.float32 else a.astype(np.float64)


def rgb2lab(rgb):
    """[..., 3] sRGB in [0, 1] -> CIE-Lab (L in [0, 100])."""
    a = _float(rgb)
    lin = np.where(a > 0.04045, np.power((a + 0.055) / 1.055, 2.4), a / 12.92).astype(a.dtype)
    xyz = lin @ _XYZ_FROM_RGB.T.astype(a.dtype) / _WHITE.astype(a.dtype)
    f = np.where(xyz > 0.008856, np.cbrt(xyz), 7.787 * xyz + 16.0 / 116.0).astype(a.dtype)
    fx, fy, fz = f[..., 0], f[..., 1], f[..., 2]
    return np.stack([116.0 * fy - 16.0, 500.0 * (fx - fy), 200.0 * (fy - fz)], axis=-1).astype(a.dtype)


def lab2rgb(lab):
    """CIE-Lab -> [..., 3] sRGB, clipped to [0, 1] (a negative z is clamped to 0 first, as scikit-image does)."""
    a = _float(lab)
    fy = (a[..., 0] + 16.0) / 116.0
    fx = a[..., 1] / 500.0 + fy
    fz = np.maximum(fy - a[..., 2] / 200.0, 0.0)
    f = np.stack([fx, fy, fz], axis=-1).astype(a.dtype)
    xyz = (np.where(f > 0.2068966, np.power(f, 3.0), (f - 16.0 / 116.0) / 7.787) * _WHITE).astype(a.dtype)
    lin = xyz @ _RGB_FROM_XYZ.T.astype(a.dtype)
    rgb = np.where(lin > 0.0031308, 1.055 * np.power(np.maximum(lin, 0.0), 1.0 / 2.4) - 0.055, 12.92 * lin)
    return np.clip(rgb, 0.0, 1.0).astype(a.dtype)


def match_and_blend_colors(source_chunk: torch.Tensor, reference_image: torch.Tensor, strength: float) -> torch.Tensor:
    """multitalk_utils.py:382-477.  source_chunk [1, 3, T, H, W] and reference_image [1, 3, 1, H, W] in [-1, 1]; every frame's Lab
    channels get the reference's mean / std (a channel without variation is set to the reference mean), converted back, clipped,
    blended: (1 - strength) * frame + strength * corrected."""
    if strength == 0.0:
        return source_chunk
    if not 0.0 <= strength <= 1.0:
        raise ValueError(f"Strength must be between 0.0 and 1.0, got {strength}")
    src = np.clip((source_chunk.squeeze(0).permute(1, 2, 3, 0).cpu().numpy() + 1.0) / 2.0, 0.0, 1.0)       # [T, H, W, 3]
    ref = np.clip((reference_image.squeeze(0).squeeze(1).permute(1, 2, 0).cpu().numpy() + 1.0) / 2.0, 0.0, 1.0)
    ref_lab = rgb2lab(ref)
    out = []
    for frame in src:
        lab = rgb2lab(frame)
        cor = lab.copy()
        for j in range(3):
            mean_s, std_s = lab[:, :, j].mean(), lab[:, :, j].std()
            mean_r, std_r = ref_lab[:, :, j].mean(), ref_lab[:, :, j].std()
            cor[:, :, j] = mean_r if std_s == 0 else (cor[:, :, j] - mean_s) * (std_r / std_s) + mean_r
        out.append((1 - strength) * frame + strength * lab2rgb(cor))
    res = torch.from_numpy(np.stack(out, axis=0) * 2.0 - 1.0).permute(3, 0, 1, 2).unsqueeze(0).contiguous()
    return res.to(device=source_chunk.device, dtype=source_chunk.dtype)


def correct_window(videos_u8: torch.Tensor, color_reference_frame: torch.Tensor, strength: float) -> torch.Tensor:
    """any2video.py:1799-1808 on one decoded window: uint8 [3, T, H, W] -> float in [-1, 1] -> match_and_blend_colors against
    color_reference_frame [3, 1, H, W] (in [-1, 1]) -> uint8 with the reference's rounding."""
    v = videos_u8.float().div_(127.5).sub_(1.0) if videos_u8.dtype == torch.uint8 else videos_u8
    v = match_and_blend_colors(v.unsqueeze(0), color_reference_frame.to(v.device, torch.float32).unsqueeze(0), strength).squeeze(0)
    return v.clamp_(-1, 1).add_(1.0).mul_(127.5).round_().clamp_(0, 255).to(torch.uint8)
