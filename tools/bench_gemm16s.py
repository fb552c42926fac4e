#!/usr/bin/env python
"""In-process A/B of the bf16 GEMM dispatch on the shapes below one wave of 256 x 256 tiles and on the short-K shapes (round 6):
  off   wan_gemm_debug_force16s(-1): the round-5 dispatch (gemm256m / gemm32 / first generation)
  t128  gemm16s.hip, 128 x 128 tiles (three workgroups per CU)
  t256  gemm16s.hip, 256 x 128 tiles (two per CU)
  auto  the shipped rule
The variants alternate inside a round on the same tensors (same box, same clock state).  usage: bench_gemm16s.py [--set configs0|1.3B|text|all]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wan2gp_amd import lib as L_, ops  # noqa: E402

SETS = {
    # BASELINE configs[0]: 1.3B, L = 3,200, S = 2 -> M = 6,400
    "configs0": [("c0 qkvo", 6400, 1536, 1536, 0), ("c0 o+gate", 6400, 1536, 1536, 2), ("c0 ffn1+gelu", 6400, 8960, 1536, 1),
                 ("c0 ffn2+gate", 6400, 1536, 8960, 2), ("c0 vT", 3200, 1536, 1536, 3)],
    # BASELINE configs[1]: 1.3B-480p, M = 65,520
    "1.3B": [("1.3B qkvo", 65520, 1536, 1536, 0), ("1.3B o+gate", 65520, 1536, 1536, 2), ("1.3B ffn1+gelu", 65520, 8960, 1536, 1),
             ("1.3B ffn2+gate", 65520, 1536, 8960, 2)],
    # the text K / V Linears of the 14B and 1.3B blocks (S x 512 context rows), the text embedding, UMT5's projections
    "text": [("14B text k", 1024, 5120, 5120, 0), ("14B text vT", 512, 5120, 5120, 3), ("14B text emb", 1024, 5120, 4096, 1),
             ("1.3B text k", 1024, 1536, 1536, 0), ("umt5 qkv", 512, 4096, 4096, 0), ("umt5 ffn", 512, 10240, 4096, 1)],
    # a rank of a world of 8 at 14B-720p: M = 2 x 9,450
    "rank8": [("r8 qkvo", 18900, 5120, 5120, 0), ("r8 ffn1+gelu", 18900, 13824, 5120, 1), ("r8 ffn2+gate", 18900, 5120, 13824, 2)],
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--set", default="all")
    ap.add_argument("--rounds", type=int, default=5)
    a = ap.parse_args()
    lib = L_.load()
    names = list(SETS) if a.set == "all" else a.set.split(",")
    g = torch.Generator(device="cuda").manual_seed(0)
    res = {}
    # (16s code, tile-rows code): off = the round-5 dispatch (gemm256m at 256 rows from 256 tiles up, else gemm32 / first generation); h160 .. h256 = gemm256m at that height on every problem
    variants = (("off", (-1, -1)), ("h256", (-1, 256)), ("t128", (128, 0)), ("t256", (256, 0)), ("h160", (-1, 160)), ("h192", (-1, 192)), ("h224", (-1, 224)), ("auto", (0, 0)))
    for sn in names:
        for name, M, N, K, epi in SETS[sn]:
            x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
            w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
            b = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
            r = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16) if epi == 2 else None
            mod = torch.randn(1, 6, N, device="cuda", generator=g).to(torch.bfloat16) if epi == 2 else None
            e = torch.randn(1, 6, N, device="cuda", generator=g).to(torch.bfloat16) if epi == 2 else None
            out = torch.zeros(N, (M + 63) // 64 * 64, device="cuda", dtype=torch.bfloat16) if epi == 3 else torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            ts = {vn: [] for vn, _ in variants}
            outs = {}
            reps = 8 if M * N * K < 1e11 else 3
            for i in range(a.rounds + 1):
                for vn, code in variants:
                    lib.wan_gemm_debug_force16s(code[0])
                    lib.wan_gemm_debug_force_tile_rows(code[1])
                    ops.linear(x, w, b, epilogue=epi, residual=r, mod=mod, e=e, gate_idx=5 if epi == 2 else -1, out=out)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(reps):
                        ops.linear(x, w, b, epilogue=epi, residual=r, mod=mod, e=e, gate_idx=5 if epi == 2 else -1, out=out)
                    e1.record()
                    torch.cuda.synchronize()
                    if i:
                        ts[vn].append(e0.elapsed_time(e1) / reps)
                    elif epi != 2:
                        outs[vn] = out.float().clone()
            lib.wan_gemm_debug_force16s(0)
            lib.wan_gemm_debug_force_tile_rows(0)
            row = {}
            for vn, _ in variants:
                t = sorted(ts[vn])
                row[vn] = {"us": round(t[len(t) // 2] * 1e3, 1), "TF": round(2.0 * M * N * K / t[len(t) // 2] / 1e9, 1)}
                if vn in outs and "off" in outs:
                    row[vn]["maxdiff_vs_off"] = float((outs[vn] - outs["off"]).abs().max())
            res[name] = {"M": M, "N": N, "K": K, "epi": epi, **row}
            print(name, json.dumps(res[name]), flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
