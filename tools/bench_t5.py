#!/usr/bin/env python
"""UMT5-XXL text-encoder timing (tuning tool, not the judged bench): full umt5_xxl geometry (t5.py:460-472), random
weights generated on the device, one prompt of 512 tokens as any2video.py:587-593 encodes it."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wan2gp_amd.t5 import T5EncoderHIP  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--L", type=int, default=512)
    ap.add_argument("--B", type=int, default=1)
    ap.add_argument("--layers", type=int, default=24)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    enc = T5EncoderHIP(num_layers=a.layers)
    g = torch.Generator(device="cuda").manual_seed(0)
    rn = lambda *s, std: (torch.randn(*s, device="cuda", generator=g) * std).to(torch.bfloat16)
    sd = {"token_embedding.weight": rn(enc.vocab_size, enc.dim, std=1.0), "norm.weight": rn(enc.dim, std=0.05) + 1}
    for i in range(a.layers):
        b = f"blocks.{i}."
        sd[b + "norm1.weight"] = rn(enc.dim, std=0.05) + 1; sd[b + "norm2.weight"] = rn(enc.dim, std=0.05) + 1
        sd[b + "attn.q.weight"] = rn(enc.dim_attn, enc.dim, std=enc.dim ** -0.5 / 8)
        for n in ("k", "v", "o"):
            sd[b + f"attn.{n}.weight"] = rn(enc.dim_attn, enc.dim, std=enc.dim ** -0.5)
        sd[b + "ffn.gate.0.weight"] = rn(enc.dim_ffn, enc.dim, std=enc.dim ** -0.5)
        sd[b + "ffn.fc1.weight"] = rn(enc.dim_ffn, enc.dim, std=enc.dim ** -0.5)
        sd[b + "ffn.fc2.weight"] = rn(enc.dim, enc.dim_ffn, std=enc.dim_ffn ** -0.5)
        sd[b + "pos_embedding.embedding.weight"] = rn(enc.num_buckets, enc.num_heads, std=0.5)
    enc.load_state_dict(sd)
    ids = torch.randint(1, enc.vocab_size, (a.B, a.L), device="cuda", generator=g)
    mask = torch.ones(a.B, a.L, dtype=torch.long, device="cuda"); mask[:, a.L * 3 // 4:] = 0
    out = enc(ids, mask)
    torch.cuda.synchronize()
    ts = []
    for _ in range(a.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = enc(ids, mask); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    wbytes = sum(v.numel() * 2 for k, v in enc.w.items() if k != "token_embedding.weight")
    flops = 2.0 * a.B * a.L * (wbytes / 2)
    t = min(ts)
    print(json.dumps({"tokens": a.B * a.L, "layers": a.layers, "ms_best": t, "ms_all": ts, "weight_GB": wbytes / 1e9,
                      "weight_stream_GBps": wbytes / t / 1e6, "gemm_TFLOPs": flops / t / 1e9, "finite": bool(torch.isfinite(out).all())}))


if __name__ == "__main__":
    main()
