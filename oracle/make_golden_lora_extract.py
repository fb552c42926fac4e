"""TEST INFRASTRUCTURE ONLY -- tests/golden/lora_extract.npz from the REFERENCE's own LoRA extractor.

Run in the build container (needs /root/reference):   python oracle/make_golden_lora_extract.py

The adapter arithmetic of the checkpoint / LoRA row is applied by mmgp, which is not in the
reference tree.  What the tree DOES hold is the producer of the adapter files that arithmetic consumes:
`shared/extract_lora.py` (`LoRAExtractor`, :13-30, :175-214, :216-284).  It states the file format and its algebra --

    finetuned.weight = original.weight + lora_up @ lora_down      (2-D weights; no `.alpha` key is written  => scale 1)
    finetuned.bias   = original.bias   + diff_b
    finetuned.weight = original.weight + diff                      (weights that are not 2-D)

under the key names `diffusion_model.<module>.lora_down.weight` / `.lora_up.weight` / `.diff_b` / `.diff`.  This script executes
that class, unmodified, on a synthetic (original, finetuned) pair over Wan module names whose weight differences have exact rank 8,
and records the pair and the extractor's file.  The tests then require that merging the file at multiplier 1 gives the finetuned
checkpoint back (tests/test_lora_extract_vs_golden.py on the CPU for oracle/loader_oracle.py and the host-side key handling,
tests/test_gpu_zzz_lora_extract.py for the HIP merge): the round trip extract -> merge = identity pins the alpha-less case of the algebra to
reference-held code.  Not pinned by it: the `alpha / rank` factor of files that carry `.alpha` and the multiplier (mmgp only).
"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "lora_extract.npz")
RANK = 8
# module -> (out, in): the Linear layers of a Wan block at toy width, plus a 1-D weight (RMSNorm gain) for the `.diff` form
LINEARS = {"blocks.0.self_attn.q": (64, 64), "blocks.0.self_attn.o": (64, 64), "blocks.0.cross_attn.k": (64, 64),
           "blocks.0.ffn.0": (160, 64), "blocks.1.ffn.2": (64, 160)}
NORMS = {"blocks.0.self_attn.norm_q": 64}


def load_extractor():
    spec = importlib.util.spec_from_file_location("ref_extract_lora", os.path.join(ref_shim.REF_ROOT, "shared", "extract_lora.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.LoRAExtractor


def make_pair(seed=11):
    g = torch.Generator().manual_seed(seed)
    orig, fine = {}, {}
    for mod, (n, k) in LINEARS.items():
        w = torch.randn(n, k, generator=g) * k ** -0.5
        b = torch.randn(n, generator=g) * 0.1
        dw = (torch.randn(n, RANK, generator=g) * 0.2) @ (torch.randn(RANK, k, generator=g) * 0.2)      # exact rank 8
        orig[mod + ".weight"], orig[mod + ".bias"] = w, b
        fine[mod + ".weight"], fine[mod + ".bias"] = w + dw, b + torch.randn(n, generator=g) * 0.02
    for mod, n in NORMS.items():
        w = 1 + torch.randn(n, generator=g) * 0.05
        orig[mod + ".weight"], fine[mod + ".weight"] = w, w + torch.randn(n, generator=g) * 0.03
    # a tensor the fine-tune left alone: the extractor must write nothing for it
    orig["blocks.1.self_attn.v.weight"] = fine["blocks.1.self_attn.v.weight"] = torch.randn(64, 64, generator=g) * 64 ** -0.5
    return orig, fine


def generate(out=OUT):
    Extractor = load_extractor()
    orig, fine = make_pair()
    file = Extractor(rank=RANK).extract_lora_from_state_dicts(orig, fine, device="cpu", show_progress=False)
    rec = {"orig/" + k: v.numpy() for k, v in orig.items()}
    rec.update({"fine/" + k: v.numpy() for k, v in fine.items()})
    rec.update({"file/" + k: v.numpy() for k, v in file.items()})
    rec["file_keys"] = np.array(list(file.keys()))                      # in the order the extractor wrote them
    np.savez_compressed(out, **rec)
    return rec


if __name__ == "__main__":
    r = generate()
    print(f"wrote {OUT}: {len(r['file_keys'])} adapter tensors, {os.path.getsize(OUT)} bytes")
