"""CPU: the host half of `wan_vae_*` (csrc/vae_graph.hip).  Registration packs weights in host memory and the workspace planner
runs the layer graph without launching anything, so both work without a GPU: the planner must visit every registered layer
(missing ones are named), respect the causal chunking (one latent frame / 1+4k video frames at a time: the peak does not grow
with clip length beyond the chunk), and stay within what the host graph needed on the GPU (11.1 GB at 720p x 81f)."""
import pytest
import torch

from oracle import vae_oracle as VO
from wan2gp_amd import lib as L
from wan2gp_amd.vae import _NativeGraph


@pytest.fixture(scope="module")
def graph():
    return _NativeGraph(VO.synth_vae_weights(), "cpu")


def plan(g, decode, t, h, w):
    return g.lib.wan_vae_workspace_bytes(g._h, 1 if decode else 0, t, h, w)


def test_decode_plan_is_bounded_by_the_chunk_not_the_clip(graph):
    a, b, c = plan(graph, True, 2, 8, 8), plan(graph, True, 21, 8, 8), plan(graph, True, 41, 8, 8)
    assert 0 < a <= b <= c
    per_frame = (c - b) / 20                       # what an extra latent frame costs: its 32-channel latent rows (a few KB), not
    assert per_frame <= 32 * 1024                  # a decoder chunk's activations (3 MB at this size)
    big = plan(graph, True, 21, 90, 160)           # 720p x 81f
    assert 6e9 < big < 13e9, big


def test_encode_plan(graph):
    a, b = plan(graph, False, 9, 64, 64), plan(graph, False, 17, 64, 64)
    assert 0 < a <= b and (b - a) <= 3 * 8 * 64 * 64 * 32 * 2                          # grows with the packed input frames only (first-fit slack included)
    # (round 6: the causal caches point at the previous chunk's input tensors instead of copying their last two frames -- a 4-frame
    # encoder chunk stays alive where a 2-frame copy did: 14.6 GB against 11-12 GB, of 288)
    assert 2e9 < plan(graph, False, 81, 720, 1280) < 16e9


def test_missing_layers_are_named_and_duplicates_rejected():
    sd = VO.synth_vae_weights()
    sd.pop("decoder.upsamples.3.resample.1.weight")
    g = _NativeGraph(sd, "cpu")
    assert plan(g, True, 2, 8, 8) == -1
    assert b"decoder.upsamples.3.resample.1" in g.lib.wan_last_error()
    assert plan(g, False, 5, 64, 64) > 0           # the encoder is complete
    w = torch.zeros(4, 4, 1, 1, 1)
    assert g.lib.wan_vae_set_conv(g._h, b"conv1", L.ptr(w), 4, 4, 1, 1, 1, None, 0) != 0 and b"twice" in g.lib.wan_last_error()
    assert plan(g, True, 0, 8, 8) == -1


def test_vae_tile_size_choice_equals_the_references():
    """WanVAE.get_VAE_tile_size (vae.py:969-1001) lifted from the reference with `ast`: automatic choice by memory / resolution,
    user presets kept."""
    import ast
    import os
    import pytest
    src = os.path.join(os.environ.get("WAN_REFERENCE_ROOT", "/root/reference"), "models", "wan", "modules", "vae.py")
    if not os.path.isfile(src):
        pytest.skip("reference tree not present")
    tree = ast.parse(open(src).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "WanVAE")
    fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "get_VAE_tile_size")
    fn.decorator_list = []
    ns = {}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), src, "exec"), ns)
    from wan2gp_amd.vae import WanVAEHIP
    for cfg in (0, 1, 2, 3):
        for mem in (4000, 8000, 12000, 16000, 23999, 24000, 48000, 294000):
            for mixed in (False, True):
                for hw in ((None, None), (720, 1280), (1088, 1920), (1440, 2560)):
                    assert WanVAEHIP.get_VAE_tile_size(cfg, mem, mixed, *hw) == ns["get_VAE_tile_size"](cfg, mem, mixed, *hw), (cfg, mem, mixed, hw)


def test_fp32_plan_is_refused_by_a_subclass_that_does_not_declare_it():
    """`vae_precision` "32" (dtype=torch.float32): a subclass gets the fp32 host graph only if it says SUPPORTS_F32 -- the Wan2.2 subclass
    used to get the Wan2.1 fp32 graph silently (round-4 advisor), refused it in rounds 4-5, and runs its own graph in fp32 since round 6
    (vae22.py `_op22`, csrc/vae22_ops.hip `_f32` forms; tests/test_gpu_vae22.py).  A subclass that declares nothing stays refused."""
    import pytest
    import torch
    from wan2gp_amd.vae import WanVAEHIP
    from wan2gp_amd.vae22 import Wan22VAEHIP
    assert WanVAEHIP.SUPPORTS_F32 and Wan22VAEHIP.SUPPORTS_F32

    class Other(WanVAEHIP):
        SUPPORTS_F32 = False
    v = object.__new__(Other)
    v.dtype, v.device = torch.float32, "cpu"
    with pytest.raises(NotImplementedError, match="fp32 plan"):
        v.load_state_dict({})
