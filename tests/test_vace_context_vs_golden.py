"""VACE context construction (`WanAny2VHIP.vace_encode_frames / vace_encode_masks / vace_context`) against
tests/golden/vace_context.npz, recorded from the reference's own `WanAny2V` methods (oracle/make_golden_vace_context.py) with
the same deterministic stand-in VAE: the methods only split by the mask, call the VAE, fold the 8x8 pixel mask into 64
channels and concatenate, so equality is exact on CPU."""
import os

import numpy as np
import torch

from oracle.make_golden_vace_context import FakeVAE, inputs
from wan2gp_amd.pipeline import WanAny2VHIP

G = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "vace_context.npz")))
t = lambda k: torch.from_numpy(G[k])


def test_vace_context_reproduces_reference():
    pipe = WanAny2VHIP(model=None, vae=FakeVAE(), device="cpu")
    frames, mask, refs = inputs()
    assert torch.equal(pipe.vace_encode_frames([frames], None, masks=[mask])[0], t("z_noref"))
    assert torch.equal(pipe.vace_encode_masks([mask], None)[0], t("m_noref"))
    assert torch.equal(pipe.vace_encode_frames([frames], refs, masks=[mask])[0], t("z_ref"))
    assert torch.equal(pipe.vace_encode_masks([mask], refs)[0], t("m_ref"))
    assert torch.equal(pipe.vace_encode_frames([frames], None, masks=None)[0], t("z_nomask"))
    z = pipe.vace_context([frames], [mask])[0]
    assert tuple(z.shape) == (96, 3, 4, 6) and torch.equal(z, torch.cat([t("z_noref"), t("m_noref")], dim=0))
    assert tuple(t("z_ref").shape) == (32, 5, 4, 6) and tuple(t("m_ref").shape) == (64, 5, 4, 6)      # two reference frames in front


def test_background_mask_of_the_first_reference_image_replaces_its_latent_and_mask_frame():
    """any2video.py:1138-1145 (recorded from the reference's two methods composed as generate() composes them): with
    input_ref_masks[0] the first reference frame of the context is the masked (inactive | reactive) encoding and its folded mask."""
    pipe = WanAny2VHIP(model=None, vae=FakeVAE(), device="cpu")
    frames, mask, refs = inputs()
    z = pipe.vace_context([frames], [mask], refs, 0, [t("bg_mask"), None])[0]
    assert torch.equal(z, torch.cat([t("z_bg"), t("m_bg")], dim=0))
    plain = pipe.vace_context([frames], [mask], refs)[0]
    assert torch.equal(z[:, 1:], plain[:, 1:]) and not torch.equal(z[:, :1], plain[:, :1])
    assert torch.equal(pipe.vace_context([frames], [mask], refs, 0, [None, None])[0], plain)
