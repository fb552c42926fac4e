"""The adapter algebra of the checkpoint / LoRA row pinned to reference-held code.

tests/golden/lora_extract.npz holds an (original, finetuned) checkpoint pair and the adapter file the REFERENCE's own
`shared/extract_lora.py` `LoRAExtractor` wrote for it (oracle/make_golden_lora_extract.py executes that class unmodified).  The
extractor states what an adapter file means -- finetuned = original + lora_up @ lora_down (no `.alpha` => scale 1), + diff_b on
biases, + diff on weights that are not 2-D -- so merging its file at multiplier 1 must give the finetuned checkpoint back:

  * oracle/loader_oracle.py (the float64 restatement every GPU merge test is checked against) does,
  * the product's host-side handling of the file (key normalisation, grouping, scale) reads it without loss,
  * with the reference tree present, re-running the extractor reproduces the fixture.
The HIP merge itself is held to the same round trip in tests/test_gpu_zzz_lora_extract.py.  NOT covered by this pin: `alpha / rank` for
files that carry `.alpha`, and per-step multipliers other than 1 (applied by mmgp, which the reference tree does not hold)."""
import os

import numpy as np
import pytest
import torch

from oracle import loader_oracle as LO
from oracle import ref_shim
from wan2gp_amd import lora as LR

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "lora_extract.npz"))
ORIG = {k[5:]: torch.from_numpy(GOLD[k]) for k in GOLD.files if k.startswith("orig/")}
FINE = {k[5:]: torch.from_numpy(GOLD[k]) for k in GOLD.files if k.startswith("fine/")}
FILE = {str(k): torch.from_numpy(GOLD["file/" + str(k)]) for k in GOLD["file_keys"]}


def test_file_is_what_the_reference_extractor_writes():
    """Key names / shapes as stated at extract_lora.py:13-30: one (down, up) pair per changed Linear weight, diff_b per changed bias,
    diff for the 1-D weight, nothing for the untouched tensor, no alpha."""
    lin = [k[:-7] for k, v in ORIG.items() if k.endswith(".weight") and v.dim() == 2 and not torch.equal(v, FINE[k])]
    assert len(lin) == 5 and "blocks.1.self_attn.v" not in lin
    want = set()
    for m in lin:
        want |= {f"diffusion_model.{m}.lora_down.weight", f"diffusion_model.{m}.lora_up.weight", f"diffusion_model.{m}.diff_b"}
    want.add("diffusion_model.blocks.0.self_attn.norm_q.diff")
    assert set(FILE) == want and not any(k.endswith(".alpha") for k in FILE)
    for m in lin:
        n, k = ORIG[m + ".weight"].shape
        assert tuple(FILE[f"diffusion_model.{m}.lora_down.weight"].shape) == (8, k)
        assert tuple(FILE[f"diffusion_model.{m}.lora_up.weight"].shape) == (n, 8)


def _grouped():
    errors = []
    mods = LR.group_adapter(LR.normalize_lora_keys(dict(FILE), "t2v", False, None), errors)
    assert errors == []
    return mods


def test_host_side_reads_the_file_without_loss():
    mods = _grouped()
    assert set(mods) == {k[len("diffusion_model."):].rsplit(".lora_", 1)[0].removesuffix(".diff_b").removesuffix(".diff") for k in FILE}
    for name, m in mods.items():
        assert LR.adapter_scale(m) == 1.0                                  # no alpha in the reference's file => up @ down as is
        if "A" in m:
            assert torch.equal(m["A"], FILE[f"diffusion_model.{name}.lora_down.weight"])
            assert torch.equal(m["B"], FILE[f"diffusion_model.{name}.lora_up.weight"])
            assert torch.equal(m["diff_b"], FILE[f"diffusion_model.{name}.diff_b"])
        else:
            assert torch.equal(m["diff"], FILE[f"diffusion_model.{name}.diff"])


def test_oracle_merge_of_the_extractors_file_gives_the_finetuned_checkpoint_back():
    """extract (reference) -> merge (oracle/loader_oracle.py) = identity: rank-8 differences at rank 8 leave only the fp32 SVD's
    rounding (1e-6 of the largest weight), biases and the 1-D weight come back exactly (diff = fine - orig in fp32)."""
    mods = _grouped()
    for name, m in mods.items():
        w0, w1 = ORIG[name + ".weight"], FINE[name + ".weight"]
        got = LO.merged_weight_exact(w0, [m], [1.0])
        if "A" in m:
            assert (got - w1.double()).abs().max().item() <= 2e-6 * w1.abs().max().item(), name
            assert (w1 - w0).abs().max().item() > 1e-2                      # the adapter is not a no-op
            b = LO.merged_bias_exact(ORIG[name + ".bias"], [m], [1.0])
            assert torch.equal(b.float(), (ORIG[name + ".bias"] + m["diff_b"]))
            assert (b - FINE[name + ".bias"].double()).abs().max().item() <= 2.0 ** -24
        else:
            assert (got.reshape(-1) - w1.double()).abs().max().item() <= 2.0 ** -23, name
    # half the multiplier = half the way (the extractor's file is linear in the difference)
    name = "blocks.0.ffn.0"
    half = LO.merged_weight_exact(ORIG[name + ".weight"], [mods[name]], [0.5])
    mid = 0.5 * (ORIG[name + ".weight"].double() + FINE[name + ".weight"].double())
    assert (half - mid).abs().max().item() <= 2e-6 * mid.abs().max().item()


@pytest.mark.skipif(not ref_shim.available(), reason="needs the reference tree (build container)")
def test_fixture_regenerates_from_the_reference(tmp_path):
    from oracle import make_golden_lora_extract as G
    rec = G.generate(str(tmp_path / "again.npz"))
    assert [str(k) for k in rec["file_keys"]] == [str(k) for k in GOLD["file_keys"]]
    for k in GOLD.files:
        if k.startswith(("orig/", "fine/")) or k.endswith((".diff", ".diff_b")):
            assert np.array_equal(rec[k], GOLD[k]), k
    # the SVD's factors are unique up to sign / LAPACK build: compare what the file means, lora_up @ lora_down
    for k in GOLD["file_keys"]:
        k = str(k)
        if k.endswith(".lora_down.weight"):
            up = k.replace("lora_down", "lora_up")
            a, b = rec["file/" + up] @ rec["file/" + k], GOLD["file/" + up] @ GOLD["file/" + k]
            assert np.abs(a - b).max() <= 2e-6 * np.abs(b).max(), k
