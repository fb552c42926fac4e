"""TEST INFRASTRUCTURE ONLY -- tests/golden/t5_small.npz from the REFERENCE's own T5Encoder.

Run in the build container (needs /root/reference):   python oracle/make_golden_t5.py
Loads models/wan/modules/t5.py unmodified (its two non-arithmetic imports -- the gguf key mapper and the
HuggingFace tokenizer wrapper -- are stubbed), builds T5Encoder at the SMALL config, loads the seeded synthetic
state dict of oracle/t5_oracle.synth_t5_weights and runs it on the seeded ids / mask on CPU in bf16 and fp32.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import t5_oracle as T  # noqa: E402

REF_ROOT = os.environ.get("WAN_REFERENCE_ROOT", "/root/reference")


def load_ref_t5():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__path__ = []
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    for n in ("shared", "shared.utils", "models", "models.wan", "models.wan.modules"):
        if n not in sys.modules:
            mod(n)
    mod("shared.utils.gguf_mapping", has_standard_gguf_tensor_names=lambda *a, **k: False, remap_state_dict_triplet=lambda *a, **k: a)
    mod("models.wan.modules.tokenizers", HuggingfaceTokenizer=object)
    torch.cuda.current_device = lambda: "cpu"   # t5.py:675 evaluates it as a default argument at import; no GPU here
    spec = importlib.util.spec_from_file_location("models.wan.modules.t5", os.path.join(REF_ROOT, "models/wan/modules/t5.py"))
    m = importlib.util.module_from_spec(spec)
    sys.modules["models.wan.modules.t5"] = m
    spec.loader.exec_module(m)
    return m


def main():
    ref = load_ref_t5()
    cfg = T.SMALL
    out = {}
    for tag, dtype in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
        W = T.synth_t5_weights(cfg, dtype=dtype)
        enc = ref.T5Encoder(cfg.vocab_size, cfg.dim, cfg.dim_attn, cfg.dim_ffn, cfg.num_heads, cfg.num_layers, cfg.num_buckets,
                            shared_pos=False, dropout=0.0).eval()
        missing, unexpected = enc.load_state_dict({k: v.clone() for k, v in W.items()}, strict=True)
        enc = enc.to(dtype)
        ids, mask = T.synth_t5_inputs(cfg)
        with torch.no_grad():
            y = enc(ids, mask)
        out[f"out_{tag}"] = y.float().numpy()
        # per-op pins
        rel = ref.T5RelativeEmbedding(cfg.num_buckets, cfg.num_heads, bidirectional=True)
        rel.embedding.weight.data = W["blocks.0.pos_embedding.embedding.weight"].clone()
        with torch.no_grad():
            out[f"posbias_{tag}"] = rel(ids.shape[1], ids.shape[1]).float().numpy()
            ln = ref.T5LayerNorm(cfg.dim)
            ln.weight.data = W["blocks.0.norm1.weight"].clone()
            x0 = W["token_embedding.weight"][ids]
            out[f"ln_{tag}"] = ln(x0).float().numpy()
            out[f"gelu_{tag}"] = ref.GELU()(x0).float().numpy()
    out["shape"] = np.array(list(T.synth_t5_inputs(cfg)[0].shape))
    path = os.path.join(ROOT, "tests", "golden", "t5_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
