#!/usr/bin/env python
"""bench.py -- denoise-steps/s of the Wan DiT hot path on MI355X (BASELINE.json metric).

One "step" = one iteration of the sampler loop at any2video.py:1490: a joint CFG pass
(2 x WanModel.forward: cond + uncond), the CFG combine and one UniPC scheduler step, on
synthetic random latents / context / random-init weights of the named architecture
(no checkpoints or datasets are reachable here).  Inputs are resident in HBM before the timed
region.  Default workload = BASELINE.json configs[2]: Wan2.2 t2v 14B (both experts resident),
720p x 81 frames (latent 16x21x90x160, L = 75,600 tokens), bf16.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 14B-720p|1.3B-480p|tiny]

N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): the token axis is sharded
across ranks (temporal sequence parallelism) -- total work fixed -> "strong".
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (config, latent f,h,w, description)
    "14B-720p": (dict(dim=5120, ffn_dim=13824, num_heads=40, num_layers=40), (21, 90, 160),
                 "Wan2.2 t2v 14B 720x1280x81f (L=75600), CFG joint pass, UniPC, both experts resident"),
    "1.3B-480p": (dict(dim=1536, ffn_dim=8960, num_heads=12, num_layers=30), (21, 60, 104),
                  "Wan2.1 t2v 1.3B 480x832x81f (L=32760), CFG joint pass, UniPC"),
    "tiny": (dict(dim=256, ffn_dim=512, num_heads=2, num_layers=2), (3, 16, 16), "plumbing check"),
}
PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)


def forward_flops(cfg, L, text_len=512):
    d, f, n = cfg["dim"], cfg["ffn_dim"], cfg["num_layers"]
    per_block = 12 * L * d * d + 4 * L * d * f + 4 * L * L * d + 4 * L * text_len * d + 4 * text_len * d * d
    return per_block * n


def random_weights(model, cfg, seed):
    """Random-init weights of the named architecture, generated directly in HBM."""
    import math
    g = torch.Generator(device="cuda").manual_seed(seed)
    d, f = cfg["dim"], cfg["ffn_dim"]

    def rn(*shape, std=0.02, dtype=torch.bfloat16, mean=0.0):
        return (torch.randn(*shape, generator=g, device="cuda", dtype=torch.float32) * std + mean).to(dtype)

    sd = {"patch_embedding.weight": rn(d, 16, 1, 2, 2, dtype=torch.float32), "patch_embedding.bias": rn(d, std=0.01, dtype=torch.float32),
          "text_embedding.0.weight": rn(d, 4096), "text_embedding.0.bias": rn(d, std=0.01),
          "text_embedding.2.weight": rn(d, d), "text_embedding.2.bias": rn(d, std=0.01),
          "time_embedding.0.weight": rn(d, 256), "time_embedding.0.bias": rn(d, std=0.01),
          "time_embedding.2.weight": rn(d, d), "time_embedding.2.bias": rn(d, std=0.01),
          "time_projection.1.weight": rn(6 * d, d), "time_projection.1.bias": rn(6 * d, std=0.01),
          "head.modulation": rn(1, 2, d, std=1 / math.sqrt(d), dtype=torch.float32),
          "head.head.weight": rn(64, d, dtype=torch.float32), "head.head.bias": rn(64, std=0.01, dtype=torch.float32)}
    for i in range(cfg["num_layers"]):
        b = f"blocks.{i}."
        sd[b + "modulation"] = rn(1, 6, d, std=1 / math.sqrt(d))
        for a in ("self_attn", "cross_attn"):
            for l in "qkvo":
                sd[b + f"{a}.{l}.weight"] = rn(d, d); sd[b + f"{a}.{l}.bias"] = rn(d, std=0.01)
            sd[b + f"{a}.norm_q.weight"] = rn(d, mean=1.0); sd[b + f"{a}.norm_k.weight"] = rn(d, mean=1.0)
        sd[b + "norm3.weight"] = rn(d, mean=1.0); sd[b + "norm3.bias"] = rn(d, std=0.01)
        sd[b + "ffn.0.weight"] = rn(f, d); sd[b + "ffn.0.bias"] = rn(f, std=0.01)
        sd[b + "ffn.2.weight"] = rn(d, f); sd[b + "ffn.2.bias"] = rn(d, std=0.01)
    model.load_state_dict(sd)
    return model


def cpu_baseline(cfg, L_full, n_layers):
    """The oracle (CPU restatement of the reference, kind 'port') timed on the host cores on a
    bounded sample: ONE 14B-config transformer block, CFG pair, at L=2048 tokens, scaled to a
    full denoise step by the FLOP ratio.  Baseline only."""
    from oracle import wan_oracle as O
    ocfg = O.WanConfig(dim=cfg["dim"], ffn_dim=cfg["ffn_dim"], num_heads=cfg["num_heads"], num_layers=1)
    W = O.synth_weights(ocfg)
    f, hh, ww = 4, 16, 32
    Ls = f * hh * ww
    g = torch.Generator().manual_seed(0)
    hid = [torch.randn(1, Ls, ocfg.dim, generator=g).to(torch.bfloat16) for _ in range(2)]
    e0 = (0.5 * torch.randn(1, 6, ocfg.dim, generator=g)).to(torch.bfloat16)
    ctx = (0.5 * torch.randn(1, 512, ocfg.dim, generator=g)).to(torch.bfloat16)
    cos, sin = O.rope_tables((f, hh, ww))
    cores = torch.get_num_threads()
    with torch.no_grad():
        O.block_forward(hid[0][:, :256], e0, ctx, cos[:256], sin[:256], W, 0, ocfg)     # warm-up
        t0 = time.perf_counter()
        for s in range(2):
            O.block_forward(hid[s], e0, ctx, cos, sin, W, 0, ocfg)
        dt = time.perf_counter() - t0
    fl_sample = 2 * forward_flops(dict(cfg, num_layers=1), Ls)
    fl_step = 2 * forward_flops(cfg, L_full)
    est = dt * fl_step / fl_sample
    return {"value": 1.0 / est, "unit": "denoise-steps/s", "cores": cores, "kind": "port",
            "sample": f"1 of {n_layers} blocks, CFG pair, L={Ls} of {L_full} tokens: {dt:.2f} s measured "
                      f"({fl_sample / dt / 1e12:.3f} TFLOP/s), scaled by FLOP ratio {fl_step / fl_sample:.0f}x"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("WAN_BENCH_WORKLOAD", "14B-720p"), choices=list(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from wan2gp_amd import lib as L_
    from wan2gp_amd.model import WanModelHIP
    from wan2gp_amd.pipeline import WanAny2VHIP
    from wan2gp_amd.rope import get_rotary_pos_embed
    from wan2gp_amd.schedulers import FlowUniPCMultistepScheduler, cfg_combine

    cfg, (f, h, w), desc = WORKLOADS[args.workload]
    L = f * (h // 2) * (w // 2)
    two_experts = args.workload == "14B-720p"
    model = random_weights(WanModelHIP(**cfg), cfg, 1234)
    model2 = random_weights(WanModelHIP(**cfg), cfg, 4321) if two_experts else None
    if world > 1:
        from wan2gp_amd.sp import SequenceParallel
        sp = SequenceParallel(rank, world)
        model.sp = sp
        if model2 is not None:
            model2.sp = sp

    g = torch.Generator(device="cuda").manual_seed(42)
    latents = torch.randn(1, 16, f, h, w, device="cuda", generator=g)
    ctx = (torch.randn(1, 512, 4096, device="cuda", generator=g) * 0.5).to(torch.bfloat16); ctx[:, 77:] = 0
    ctx_null = (torch.randn(1, 512, 4096, device="cuda", generator=g) * 0.5).to(torch.bfloat16); ctx_null[:, 8:] = 0
    freqs = get_rotary_pos_embed((f, h, w), device="cuda")
    sched = FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
    total_steps = args.steps + args.warmup
    sched.set_timesteps(max(30, total_steps), device="cuda", shift=12.0)
    guide, switch_threshold = 4.0, 875
    lib = L_.load()

    def one_step(i, lat):
        t = sched.timesteps[i]
        trans = model2 if (model2 is not None and int(t) <= switch_threshold) else model
        cond, uncond = trans([lat, lat], t=torch.stack([t]), context=[ctx, ctx_null], freqs=freqs)
        noise = cfg_combine(cond, uncond, guide if trans is model else 3.0)
        return sched.step(noise, t, lat)[0]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    lat = latents
    for i in range(args.warmup):
        lat = one_step(i, lat)
    barrier()
    lib.wan_prof_enable(1)
    t0 = time.perf_counter()
    for i in range(args.warmup, total_steps):
        lat = one_step(i, lat)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    assert torch.isfinite(lat).all(), "non-finite latents"

    import ctypes
    prof = {}
    for cls, name in ((0, "self_attn"), (1, "cross_attn"), (2, "ffn_gemm_pair"), (3, "rmsnorm_rope")):
        ms, n = ctypes.c_double(), ctypes.c_int()
        L_.check(lib.wan_prof_collect(cls, ctypes.byref(ms), ctypes.byref(n)), "wan_prof_collect")
        prof[name] = (ms.value, n.value)
    lib.wan_prof_enable(0)

    if rank == 0:
        S = 2
        d, ffn = cfg["dim"], cfg["ffn_dim"]
        Ll = L // world
        ms, n = prof["self_attn"]
        attn_flops = 4.0 * Ll * L * d * S                      # algorithmic FLOP of one launch (S streams)
        achieved = attn_flops / (ms / n * 1e-3) / 1e12 if n else 0.0
        kern = {}
        if prof["ffn_gemm_pair"][1]:
            m2, n2 = prof["ffn_gemm_pair"]
            kern["ffn_gemm_pair_TFLOPs"] = 4.0 * S * Ll * d * ffn / (m2 / n2 * 1e-3) / 1e12
        if prof["rmsnorm_rope"][1]:
            m3, n3 = prof["rmsnorm_rope"]
            kern["rmsnorm_rope_GBs"] = 4.0 * S * Ll * d * 2 / (m3 / n3 * 1e-3) / 1e9
        if prof["cross_attn"][1]:
            m4, n4 = prof["cross_attn"]
            kern["cross_attn_TFLOPs"] = 4.0 * S * Ll * 512 * d / (m4 / n4 * 1e-3) / 1e12
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "attn_pmc_traffic.json")
        if os.path.isfile(pmc):
            try:
                traffic = json.load(open(pmc)).get(args.workload, {}).get("traffic_bytes")
            except Exception:
                traffic = None
        out = {
            "metric": "denoise-steps/s", "value": args.steps / dt, "unit": "steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": desc, "latent": [16, f, h, w], "tokens": L, "streams": 2, "guide_scale": guide,
                       "solver": "unipc", "parallelism": "sp%d" % world if world > 1 else "single",
                       "forward_TFLOP": forward_flops(cfg, L) / 1e12},
            "roofline": {"kernel": "attn_w64q_kernel (self-attention)", "bound": "mfma", "achieved": achieved,
                         "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_TFLOPS,
                         "traffic": traffic, "launches": n, "avg_ms": ms / n if n else None,
                         "flop_per_launch": attn_flops, "other_kernels": kern},
            "step_TFLOPs": 2 * forward_flops(cfg, L) / (dt / args.steps) / 1e12,
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, L, cfg["num_layers"])
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
