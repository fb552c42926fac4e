"""3-axis RoPE tables for Wan (host-side, computed once per generate()).

Mirrors models/wan/modules/posemb_layers.py: get_rotary_pos_embed (:492-525) ->
get_nd_rotary_pos_embed (:346-431) -> get_1d_rotary_pos_embed (:434-476), fp32
(USE_FP32_ROPE_FREQS=True, :6), head_dim 128 split [44,42,42] over (t,h,w) (:356),
theta 10000, cos/sin repeat-interleaved x2; RIFLEx (:35-85, :417-419) on the time axis when enable_RIFLEx.
"""
from typing import Sequence, Tuple

import torch

ROPE_DIM_LIST = (44, 42, 42)
PATCH_SIZE = (1, 2, 2)


RIFLEX_K = 6


def get_rotary_pos_embed(latents_size: Sequence[int], enable_RIFLEx: bool = False, device=None
                         ) -> Tuple[torch.Tensor, torch.Tensor]:
    """latents_size = (f, h, w) of the latent video -> (cos, sin) each [f*(h/2)*(w/2), 128] fp32."""
    assert all(s % p == 0 for s, p in zip(latents_size, PATCH_SIZE)), \
        f"latent size {tuple(latents_size)} not divisible by patch size {PATCH_SIZE}"
    sizes = [int(s) // p for s, p in zip(latents_size, PATCH_SIZE)]
    axes = torch.meshgrid(*[torch.arange(n, dtype=torch.float32) for n in sizes], indexing="ij")
    cos_parts, sin_parts = [], []
    for axis, (dim_axis, pos) in enumerate(zip(ROPE_DIM_LIST, axes)):
        freqs = 1.0 / (10000.0 ** (torch.arange(0, dim_axis, 2, dtype=torch.float32)[: dim_axis // 2] / dim_axis))
        if axis == 0 and enable_RIFLEx:
            # RIFLEx (posemb_layers.py:70-76, :417-419): the intrinsic temporal frequency (k = 6, posemb_layers.py:353) is
            # lowered so that the L_test = latent-frame count stays within 90 % of one period
            freqs[RIFLEX_K - 1] = 0.9 * 2 * torch.pi / int(latents_size[0])
        ang = torch.outer(pos.reshape(-1), freqs)
        cos_parts.append(ang.cos().repeat_interleave(2, dim=1))
        sin_parts.append(ang.sin().repeat_interleave(2, dim=1))
    cos, sin = torch.cat(cos_parts, dim=1).contiguous(), torch.cat(sin_parts, dim=1).contiguous()
    if device is not None:
        cos, sin = cos.to(device), sin.to(device)
    return cos, sin
