"""-m gpu: the HIP LoRA merge held to the reference's own adapter-file producer.  Written after round 3's
GPU budget was spent -- its body ran on the CPU with torch stand-ins for the three HIP ops -- so it is named to run behind the suites
that have run on hardware.  The CPU half (oracle + key handling on the same file): tests/test_lora_extract_vs_golden.py."""
import types

import numpy as np
import pytest
import torch

from oracle import loader_oracle as LO

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _fake_model(W):
    return types.SimpleNamespace(_weights={k: v.cuda().contiguous() for k, v in W.items()}, device=torch.device("cuda"))


def test_merge_of_the_references_extracted_file_gives_the_finetuned_checkpoint_back():
    """The adapter algebra pinned to reference-held code: tests/golden/lora_extract.npz is an (original, finetuned) pair and the
    file the reference's own `shared/extract_lora.py` wrote for it (oracle/make_golden_lora_extract.py; its meaning --
    finetuned = original + lora_up @ lora_down, + diff_b, no alpha -- is stated there, :13-30, :254-256).  Merged at multiplier 1
    into the bf16 original, every Linear weight / bias must be a correct bf16 rounding of bf16(original) + (finetuned - original);
    unloading gives the original back.  (The CPU half -- oracle and key handling on the same file -- is
    tests/test_lora_extract_vs_golden.py.)"""
    import os
    from wan2gp_amd.lora import MergedLoras
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "lora_extract.npz"))
    orig = {k[5:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("orig/")}
    fine = {k[5:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("fine/")}
    file = {str(k): torch.from_numpy(gold["file/" + str(k)]) for k in gold["file_keys"]}
    # the Linear layers (the 1-D RMSNorm gain's `.diff` stays with the CPU test: the resident model keeps those gains in fp32)
    lin = sorted(k[:-7] for k, v in orig.items() if k.endswith(".weight") and v.dim() == 2)
    W = {}
    for m in lin:
        W[m + ".weight"] = orig[m + ".weight"].to(BF)
        if m + ".bias" in orig:
            W[m + ".bias"] = orig[m + ".bias"].to(BF)
    model = _fake_model(W)
    ml = MergedLoras(model)
    ml.add({k: v for k, v in file.items() if not k.endswith("norm_q.diff")})          # through normalize_lora_keys, as wgp.py's loader does
    assert ml.errors == []
    ml.set_multipliers([1.0])
    moved = 0
    for m in lin:
        w0 = W[m + ".weight"]
        dw = fine[m + ".weight"].double() - orig[m + ".weight"].double()
        exact = w0.double() + dw
        mag = float(w0.abs().max()) + float(dw.abs().max())
        # 2e-6 of the terms: the reference extractor's fp32 SVD does not reproduce the difference better than that
        assert LO.bf16_round_ok(model._weights[m + ".weight"].cpu(), exact, mag, slack=4e-6).all(), m
        moved += int(not torch.equal(model._weights[m + ".weight"].cpu(), w0))
        if m + ".bias" in W:
            b0 = W[m + ".bias"]
            exb = b0.double() + (fine[m + ".bias"].double() - orig[m + ".bias"].double())
            assert LO.bf16_round_ok(model._weights[m + ".bias"].cpu(), exb, float(b0.abs().max()) + 0.1).all(), m
    assert moved == 5                                                   # every fine-tuned Linear moved, the untouched one did not
    assert torch.equal(model._weights["blocks.1.self_attn.v.weight"].cpu(), W["blocks.1.self_attn.v.weight"])
    ml.unload()
    assert all(torch.equal(model._weights[k].cpu(), W[k]) for k in W)
