"""Sub-parallel temporal windows (wan2gp_amd/subparallel.py) against tests/golden/subparallel.npz: the reference's own nested
closures of `WanAny2V.generate` (any2video.py:1199-1387), lifted verbatim and executed by oracle/make_golden_subparallel.py
for the plain case, with a deterministic stand-in for the CFG denoise function that depends on the window's latents AND on
the sliced keywords (RoPE rows, t, y, vace_context), incl. a reference-image prefix in front of every window and per-frame t.  Exact equality on CPU: windows, latent counts, blended predictions."""
import os

import numpy as np
import pytest
import torch

from oracle.make_golden_subparallel import cases, fake_denoise_factory, make_inputs
from wan2gp_amd import subparallel as SP

G = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "subparallel.npz")))


def as_list(a):
    return None if a.size == 0 else [tuple(int(v) for v in r) for r in a]


@pytest.mark.parametrize("c", cases(), ids=lambda c: c["name"])
def test_windows_counts_and_blended_prediction(c):
    win, ov = SP.window_latent_counts(c["size"], c["overlap"], c["lat"], 4)
    assert [win, ov] == G[c["name"] + "_counts"].tolist()
    windows = SP.build_windows(c["lat"], win, ov)
    assert windows == as_list(G[c["name"] + "_windows"])
    if windows is None:
        return
    P = c.get("prefix", 0)
    latents, freqs, y, vace, *tt = make_inputs(c["lat"], P, c.get("t", False))
    kwargs = {"freqs": freqs, "y": y, "vace_context": vace, "other": 3}
    if tt:
        kwargs["t"] = tt[0]                                     # per-frame timesteps: the stand-in asserts they arrive sliced
    pred = SP.denoise(latents.clone(), fake_denoise_factory(kwargs), windows, ov, kwargs, tokens_per_frame=2 * 3, prefix=P)
    assert kwargs["freqs"] is freqs and kwargs["y"] is y and kwargs["other"] == 3                 # restored after every window
    assert torch.equal(pred, torch.from_numpy(G[c["name"] + "_pred"]))


def test_build_windows_table():
    for key in [k for k in G if k.startswith("build_")]:
        total, size, ov = (int(v) for v in key.split("_")[1:])
        assert SP.build_windows(total, size, ov) == as_list(G[key]), key


def test_interrupted_window_returns_none():
    latents, freqs, y, vace = make_inputs(21)
    kwargs = {"freqs": freqs}
    calls = []
    assert SP.denoise(latents, lambda lat: (calls.append(lat.shape[2]), None)[1], [(0, 5), (3, 8)], 2, kwargs, 6) is None
    assert calls == [5] and kwargs["freqs"] is freqs
