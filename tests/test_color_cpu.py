"""wan2gp_amd/color.py: the Lab colour matching of sliding windows (multitalk_utils.py:382-477 over scikit-image's rgb2lab / lab2rgb,
which is not in the reference tree: PARITY UNPINNED against scikit-image, pinned by CIE reference colours, the round trip and the
statistics the function must produce)."""
import numpy as np
import torch

from wan2gp_amd import color as C


def test_lab_of_reference_colours():
    """sRGB primaries / white / mid grey under D65 (the values every colour-science table carries; scikit-image prints the same)."""
    want = {(1, 1, 1): (100.0, 0.0, 0.0), (0, 0, 0): (0.0, 0.0, 0.0), (1, 0, 0): (53.24, 80.09, 67.20), (0, 1, 0): (87.73, -86.18, 83.18),
            (0, 0, 1): (32.30, 79.19, -107.86), (0.5, 0.5, 0.5): (53.39, 0.0, 0.0)}
    for rgb, lab in want.items():
        got = C.rgb2lab(np.array(rgb, dtype=np.float64))
        assert np.allclose(got, lab, atol=0.03), (rgb, got)


def test_round_trip_and_gamut_clip():
    rng = np.random.default_rng(3)
    x = rng.random((17, 19, 3))
    assert np.abs(C.lab2rgb(C.rgb2lab(x)) - x).max() < 1e-6
    x32 = x.astype(np.float32)                                     # float32 frames stay float32 (scikit-image's rule; the reference's case)
    lab32 = C.rgb2lab(x32)
    assert lab32.dtype == np.float32 and C.lab2rgb(lab32).dtype == np.float32
    assert np.abs(C.lab2rgb(lab32) - x32).max() < 2e-4 and np.abs(lab32 - C.rgb2lab(x)).max() < 2e-3
    out = C.lab2rgb(np.array([[60.0, 120.0, -120.0], [100.0, 0.0, 300.0], [-5.0, 0.0, 0.0]]))
    assert out.min() >= 0.0 and out.max() <= 1.0


def test_match_and_blend_moves_the_lab_statistics_onto_the_reference():
    g = torch.Generator().manual_seed(1)
    src = (torch.rand(1, 3, 3, 24, 32, generator=g) * 0.5 + 0.1) * 2 - 1              # dim, low contrast, no channel clips
    ref = (torch.rand(1, 3, 1, 24, 32, generator=g) * 0.4 + 0.4) * 2 - 1
    assert C.match_and_blend_colors(src, ref, 0.0) is src
    full = C.match_and_blend_colors(src, ref, 1.0)
    assert full.shape == src.shape and full.dtype == src.dtype
    ref_lab = C.rgb2lab(((ref[0, :, 0] + 1) / 2).permute(1, 2, 0).numpy())
    for t in range(3):
        lab = C.rgb2lab(((full[0, :, t] + 1) / 2).permute(1, 2, 0).double().numpy())
        for j in range(3):                                                            # in gamut: mean and std are the reference's
            assert abs(lab[:, :, j].mean() - ref_lab[:, :, j].mean()) < 0.35 and abs(lab[:, :, j].std() - ref_lab[:, :, j].std()) < 0.35
    half = C.match_and_blend_colors(src, ref, 0.5)
    assert torch.allclose(half, 0.5 * src + 0.5 * full, atol=1e-6)                    # the blend is linear in RGB
    flat = torch.full((1, 3, 1, 8, 8), 0.2)
    out = C.match_and_blend_colors(flat, ref[:, :, :, :8, :8], 1.0)                   # a channel without variation takes the reference mean
    lab = C.rgb2lab(((out[0, :, 0] + 1) / 2).permute(1, 2, 0).double().numpy())
    assert lab[:, :, 0].std() < 1e-6
    try:
        C.match_and_blend_colors(src, ref, 1.5)
        raise AssertionError("strength > 1 accepted")
    except ValueError:
        pass


def test_correct_window_keeps_uint8_and_is_identity_on_its_own_reference_statistics():
    g = torch.Generator().manual_seed(2)
    vid = torch.randint(30, 220, (3, 2, 16, 16), generator=g, dtype=torch.uint8)
    ref = vid[:, :1].float() / 127.5 - 1.0
    out = C.correct_window(vid, ref, 1.0)
    assert out.dtype == torch.uint8 and out.shape == vid.shape
    assert (out[:, 0].int() - vid[:, 0].int()).abs().max() <= 1                       # frame 0 already has the reference's statistics
