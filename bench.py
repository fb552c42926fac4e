#!/usr/bin/env python
"""bench.py -- denoise-steps/s (+ end-to-end s/video) of the Wan DiT hot path on MI355X (BASELINE.json metric).

One "step" = one iteration of the sampler loop at any2video.py:1490: a joint CFG pass
(2 x WanModel.forward: cond + uncond), the CFG combine and one UniPC scheduler step, on
synthetic random latents / context / random-init weights of the named architecture
(no checkpoints or datasets are reachable here).  Inputs are resident in HBM before the timed
region.  Default workload = BASELINE.json configs[2]: Wan2.2 t2v 14B (both experts resident),
720p x 81 frames (latent 16x21x90x160, L = 75,600 tokens), bf16.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 14B-720p|14B-720p-161f|1.3B-480p|i2v-14B-720p|tiny] [--fp8]

N > 1: one rank per GPU over RCCL; the token axis is sharded across ranks (temporal sequence parallelism,
SURVEY.md section 8e) -- total work fixed -> "strong".  An even N first splits the two CFG streams over the two halves of the
world (no per-block exchange between the halves, one swap of the predictions per step) and shards tokens inside a half
(--parallelism cfg-sp, the default there; --parallelism sp = tokens over all N ranks, both streams on every rank).  Launched by `torch.distributed.run` (the driver's way) the ranks
are already there; launched as plain `python bench.py --gpus N` this process re-executes itself under
torch.distributed.run with N ranks on 127.0.0.1.  Prints ONE JSON line on rank 0.

Besides the contract fields the line carries
  roofline      the dominant kernel (self-attention): algorithmic FLOP per launch / its average duration measured with HIP
                events on the launch stream inside the timed region
  e2e           end-to-end seconds per video: noise -> steps -> causal 3D VAE decode -> uint8 on the host, (a) measured
                over the W+K steps this run executed and (b) composed for the 30-step default from the measured step time
                plus the VAE decode / host copy measured in this run (text encoding excluded: the context is synthetic)
  secondary     (N = 1) BASELINE configs[1], Wan2.1 t2v 1.3B 480x832x81f: a full generate() -- 30 steps + VAE decode
  cpu_baseline  the oracle (CPU restatement of the reference, kind "port") on the host cores: BASELINE configs[0], one real
                CFG step of the 1.3B model at L = 3,200 -- baseline only
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (config, latent f,h,w, description)
    "14B-720p": (dict(dim=5120, ffn_dim=13824, num_heads=40, num_layers=40), (21, 90, 160),
                 "Wan2.2 t2v 14B 720x1280x81f (L=75600), CFG joint pass, UniPC, both experts resident"),
    # BASELINE configs[3]: 161 frames -> (161 - 1) // 4 + 1 = 41 latent frames (any2video.py:647,1166), L = 41 x 45 x 80 = 147,600
    "14B-720p-161f": (dict(dim=5120, ffn_dim=13824, num_heads=40, num_layers=40), (41, 90, 160),
                      "Wan2.2 t2v 14B 720x1280x161f (L=147600), CFG joint pass, UniPC, both experts resident"),
    "i2v-14B-720p": (dict(dim=5120, ffn_dim=13824, num_heads=40, num_layers=40, in_dim=36, model_type="i2v2_2"), (21, 90, 160),
                     "Wan2.2 i2v 14B 720x1280x81f (L=75600, in_dim 36), CFG joint pass, UniPC, both experts resident, "
                     "VAE encode of the conditioning video + decode in the e2e figure"),
    "1.3B-480p": (dict(dim=1536, ffn_dim=8960, num_heads=12, num_layers=30), (21, 60, 104),
                  "Wan2.1 t2v 1.3B 480x832x81f (L=32760), CFG joint pass, UniPC"),
    # BASELINE configs[0] (defaults/t2v_1.3B.json at 320 x 512 x 17 frames: the CPU-runnable case the cpu_baseline leg is quoted on)
    "1.3B-320x512x17f": (dict(dim=1536, ffn_dim=8960, num_heads=12, num_layers=30), (5, 40, 64),
                         "Wan2.1 t2v 1.3B 320x512x17f (L=3200), CFG joint pass, UniPC"),
    "tiny": (dict(dim=256, ffn_dim=512, num_heads=2, num_layers=2), (3, 16, 16), "plumbing check"),
}
TWO_EXPERT_WORKLOADS = ("14B-720p", "14B-720p-161f", "i2v-14B-720p")
PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
VIDEO_STEPS = 30            # UI default sampling steps (defaults/t2v_2_2.json)
T_PROCESS0 = time.perf_counter()


def forward_flops(cfg, L, text_len=512):
    d, f, n = cfg["dim"], cfg["ffn_dim"], cfg["num_layers"]
    per_block = 12 * L * d * d + 4 * L * d * f + 4 * L * L * d + 4 * L * text_len * d + 4 * text_len * d * d
    return per_block * n


def random_weights(model, cfg, seed, fp8=False):
    """Random-init weights of the named architecture, generated directly in HBM.  fp8: the ten Linears of every block become a
    scaled-fp8 checkpoint (float8_e4m3fn weight + fp32 per-row `scale_weight`, shared/qtypes/scaled_fp8.py:563-637)."""
    import math
    import torch
    g = torch.Generator(device="cuda").manual_seed(seed)
    d, f, cin = cfg["dim"], cfg["ffn_dim"], cfg.get("in_dim", 16)

    def rn(*shape, std=0.02, dtype=torch.bfloat16, mean=0.0):
        return (torch.randn(*shape, generator=g, device="cuda", dtype=torch.float32) * std + mean).to(dtype)

    sd = {"patch_embedding.weight": rn(d, cin, 1, 2, 2, dtype=torch.float32), "patch_embedding.bias": rn(d, std=0.01, dtype=torch.float32),
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
        if fp8:
            for name in [f"{a}.{l}" for a in ("self_attn", "cross_attn") for l in "qkvo"] + ["ffn.0", "ffn.2"]:
                w = sd[b + name + ".weight"].float()
                sc = (w.abs().amax(dim=1, keepdim=True) / 448.0).clamp_min(1e-12)
                sd[b + name + ".weight"] = (w / sc).clamp(-448, 448).to(torch.float8_e4m3fn)
                sd[b + name + ".scale_weight"] = sc.reshape(-1)
    model.load_state_dict(sd)
    return model


def _cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(flops_step_main, sweep=(8, 16, 32, 64, 128), _cfg_name="t2v_1.3B", _fhw=(5, 40, 64)):
    """BASELINE.md section 3 on the host cores of this box, through the oracle: kind "port" = the CPU restatement of the
    reference (oracle/wan_oracle.py, oracle/vae_oracle.py), bit-exact to the reference's own modules on tests/golden/cfg1_*.npz
    and vae_small.npz.  The reference tree itself does not travel to the GPU box (/root/reference exists only in the build
    container), so "reference" is not available here.
    Workload = BASELINE configs[0]: Wan2.1 t2v 1.3B, 320x512x17f (latent 16x5x40x64, L = 3,200), 10 steps, unipc, shift 5,
    guidance 5 -> 20 forwards + 10 scheduler steps + 1 VAE decode to uint8 [3,17,320,512].  BOUNDED sample of it (the bench must
    finish in minutes): ONE real CFG step (joint cond + uncond forward of all 30 layers, CFG combine, UniPC step) and the ONE
    VAE decode are timed; the 10-step end-to-end figure is composed as 10 x step + decode and labelled as composed.  The
    14B-720p figure next to it is a FLOP-ratio extrapolation and labelled as such.  Baseline only.
    Thread count: L = 3,200 problems do not feed 128 cores (round 3's line ran 3.9x slower on 128 threads of the GPU box than on 8
    cores of the build container), so both legs are first timed on a short sample -- the first 3 layers of the joint forward at the
    full L; the first latent frame of the decode -- at every count of `sweep` the machine has, and the full sample runs at the best
    one; the sweep is reported.  Runs alone: nothing is on the GPU while it runs."""
    import dataclasses
    import torch
    from oracle import vae_oracle as VO
    from oracle import wan_oracle as O
    cfg = O.make_config(_cfg_name)                   # (the underscore arguments exist for the CPU test: a tiny stand-in, seconds)
    f, h, w = _fhw
    t0 = time.perf_counter()
    W = O.synth_weights(cfg)
    t_w = time.perf_counter() - t0
    lat, ctx, ctx_null, _ = O.synth_inputs(cfg, f, h, w)
    sch = O.UniPCOracle()
    ts = sch.set_timesteps(10, 5.0)
    freqs = O.rope_tables((f, h // 2, w // 2))
    default_threads = torch.get_num_threads()
    ncpu = os.cpu_count() or default_threads
    counts = sorted({n for n in sweep if n <= ncpu} | {default_threads})
    cfg3 = dataclasses.replace(cfg, num_layers=min(3, cfg.num_layers))
    WV = VO.synth_vae_weights()
    z1 = torch.randn(1, 16, 1, h // 2, w, generator=torch.Generator().manual_seed(3))       # half the rows: the sweep stays a few seconds per count
    sweep_dit, sweep_vae = {}, {}
    with torch.no_grad():
        for n in counts:
            torch.set_num_threads(n)
            O.dit_forward([lat[:, :, :1, :8, :8]], torch.stack([ts[0]]), [ctx], W, cfg3)             # warm-up: thread pool, kernels
            t0 = time.perf_counter()
            O.dit_forward([lat, lat], torch.stack([ts[0]]), [ctx, ctx_null], W, cfg3, freqs=freqs)
            sweep_dit[n] = time.perf_counter() - t0
            t0 = time.perf_counter()
            VO.vae_decode(z1, WV, VO.default_scale())
            sweep_vae[n] = time.perf_counter() - t0
        n_dit = min(sweep_dit, key=sweep_dit.get)
        n_vae = min(sweep_vae, key=sweep_vae.get)
        torch.set_num_threads(n_dit)
        t0 = time.perf_counter()
        cond, uncond = O.dit_forward([lat, lat], torch.stack([ts[0]]), [ctx, ctx_null], W, cfg, freqs=freqs)
        nxt = sch.step(O.cfg_combine(cond, uncond, 5.0), lat)
        dt = time.perf_counter() - t0
        del W
        z = (nxt[0] if isinstance(nxt, (tuple, list)) else nxt).float()
        torch.set_num_threads(n_vae)
        t0 = time.perf_counter()
        frames = VO.float_to_uint8(VO.vae_decode(z.reshape(1, 16, f, h, w), WV, VO.default_scale())[0])
        dt_vae = time.perf_counter() - t0
        torch.set_num_threads(default_threads)
    assert tuple(frames.shape) == (3, (f - 1) * 4 + 1, h * 8, w * 8)
    L = f * (h // 2) * (w // 2)
    fl = 2 * forward_flops(dict(dim=cfg.dim, ffn_dim=cfg.ffn_dim, num_layers=cfg.num_layers), L)
    return {"value": 1.0 / dt, "unit": "denoise-steps/s", "cores": n_dit, "cores_vae_decode": n_vae, "cpu_model": _cpu_model_name(),
            "os_cpu_count": os.cpu_count(), "torch_default_threads": default_threads, "torch": torch.__version__,
            "thread_sweep": {"sample": "joint CFG forward, first 3 of 30 layers, L = %d; VAE decode of the first latent frame, upper half of the picture" % L,
                             "dit_3_layers_s": {str(k): round(v, 3) for k, v in sweep_dit.items()},
                             "vae_first_frame_s": {str(k): round(v, 3) for k, v in sweep_vae.items()},
                             "note": "the full sample below ran at the best count of each leg ('cores', 'cores_vae_decode'), alone on the box"},
            "kind": "port",
            "kind_note": "CPU restatement of the reference (oracle/), bit-exact to the reference's own modules on tests/golden/cfg1_*.npz "
                         "and vae_small.npz; the reference tree does not exist on the GPU box.  Calibration on a machine that has both (8-core build "
                         "container, profiles/r03_cpu_reference_e2e_configs0.json vs r03_cpu_port_same_machine_configs0.json; NOT measured in this "
                         "run): the reference's own WanModel / UniPC / WanVAE_ take 8.90 s per CFG step (median of steps 2..10), 51.3 s for the "
                         "decode and 140.3 s for the 10-step video end to end; this port 9.15 s, 53.7 s and 145.1 s composed -- within 3-5 %",
            "sample": f"BASELINE configs[0]: Wan2.1 t2v 1.3B 320x512x17f (L={L}): one full CFG step (2 forwards x 30 layers + "
                      f"combine + UniPC) in the reference's bf16 plan on {n_dit} threads: {dt:.2f} s measured ({fl / dt / 1e12:.3f} TFLOP/s); one VAE "
                      f"decode (fp32) to uint8 [3,17,320,512] on {n_vae} threads: {dt_vae:.2f} s measured; synthetic checkpoint built in {t_w:.0f} s (untimed)",
            "step_s": dt, "vae_decode_s": dt_vae,
            "e2e_s_per_video_composed": 10 * dt + dt_vae,
            "e2e_note": "configs[0] end to end = 10 steps + VAE decode, COMPOSED from the one timed step and the one timed decode "
                        "(BASELINE.md section 3 asks for all 10 steps; the bench keeps the CPU leg to a bounded sample)",
            "extrapolated_main_workload_steps_per_s": fl / dt / flops_step_main,
            "extrapolation": f"FLOP ratio {flops_step_main / fl:.0f}x to the main workload (not measured; BASELINE.md section 3)"}


def choose_layout(world, parallelism, heads):
    """-> (cfg_sp, sp_degree, sp_mode): how `world` ranks share a guided step.  cfg_sp: the conditional stream on ranks [0, world/2), the
    unconditional one on the other half, one swap of the predictions per step; sp_degree: ranks that share one stream's token axis;
    sp_mode: the per-block exchange inside such a group.  `auto` follows the link-modelled one-GPU tables (runs 02 / 05,
    25 / 50 / 100 GB/s per peer):
      * world 2: cfg2 x sp1 -- no per-block exchange at all (0.98);
      * world 4: cfg2 x sp2 with the K / V^T all-gathers (0.925: one peer, and the gathers hide under a local attention segment half a
        block long; the Ulysses forms 0.83-0.88);
      * world >= 8 where the heads divide by it (14B: 40 heads, world 8): the token axis over ALL ranks with the head-chunked Ulysses
        all-to-alls -- every exchange spreads over 7 links instead of 3, 0.864 / 0.872 / 0.897 at 25 / 50 / 100 GB/s against 0.817 / 0.868 /
        0.895 for cfg2 x sp4 (ulysses) and 0.59-0.84 for the all-gather forms;
      * otherwise an even world splits the streams (cfg2 x sp world/2), with the Ulysses exchange from a degree of 4 up where the heads
        divide by it (1.3B: 12 heads -> cfg2 x sp4 at world 8), else the all-gathers; an odd world is plain sequence parallelism."""
    if world <= 1:
        return False, 1, "allgather"
    if parallelism == "auto":
        if world >= 8 and heads % world == 0:
            return False, world, "ulysses"
        cfg_sp = world % 2 == 0
        deg = world // 2 if cfg_sp else world
        return cfg_sp, deg, ("ulysses" if deg >= 4 and heads % deg == 0 else "allgather")
    cfg_sp = parallelism in ("cfg-sp", "cfg-ulysses") and world % 2 == 0
    return cfg_sp, (world // 2 if cfg_sp else world), ("ulysses" if parallelism in ("ulysses", "cfg-ulysses") else "allgather")


# ---- first-run hardening of `--gpus N` (round 6) ---------------------------------------------------------------
# No multi-GPU node was ever available to the builder: the first real run is the driver's.  Three rules make sure it produces a line:
#   * every layout self-test runs under a wall-clock guard (a collective whose peer never arrives blocks its caller for the backend's
#     timeout: the bench decides after GUARD_S instead) and the ranks agree on the outcome over a side channel of their own (a gloo group:
#     a rank that skipped a data collective does not shift its sequence numbers).  Outcome 1 (an error somewhere) demotes the layout on
#     every rank; outcome 2 (a HANG somewhere) also gives whatever follows fresh process groups -- the old ones hold an unmatched collective;
#   * if the agreement itself cannot complete, or the basic all-gather fails too, or the whole multi-GPU part is not through its timed
#     region WORLD_DEADLINE_S after the process started, the world is given up: ranks > 0 exit quietly (status 0: torch.distributed.run
#     then lets rank 0 finish), rank 0 re-executes itself as a short single-GPU run whose line says what happened (`layout_fell_back_to`);
#   * init_process_group has a timeout of its own (180 s) instead of the backend's 10 minutes.
GUARD_S = float(os.environ.get("WAN_BENCH_GUARD_S", 45))
AGREE_S = float(os.environ.get("WAN_BENCH_AGREE_S", 90))
WORLD_DEADLINE_S = float(os.environ.get("WAN_BENCH_WORLD_DEADLINE_S", 420))
_AGREE = {"group": None}
_TORCHRUN_VARS = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE", "ROLE_NAME",
                  "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS", "TORCHELASTIC_RUN_ID",
                  "TORCHELASTIC_USE_AGENT_STORE", "TORCH_NCCL_ASYNC_ERROR_HANDLING", "TORCHELASTIC_ERROR_FILE")


class WorldLost(RuntimeError):
    """The ranks can no longer decide anything together: the multi-GPU run is given up (see _leave_world)."""


def _guarded(fn, seconds, device=None):
    """fn() on a thread of its own -> (finished in time, result or the exception it raised).  A thread that does not finish is left
    behind (daemon): it sits in a collective nobody will complete.  `device`: the HIP device is per thread."""
    box = {}

    def run():
        try:
            if device is not None and str(device) != "cpu":
                import torch
                torch.cuda.set_device(device)
            box["r"] = fn()
        except BaseException as ex:  # noqa: BLE001 -- handed to the caller
            box["e"] = ex
    th = threading.Thread(target=run, daemon=True)
    th.start()
    th.join(seconds)
    if th.is_alive():
        return False, TimeoutError("no answer after %.0f s" % seconds)
    return True, box.get("e", box.get("r"))


def make_agreement_group():
    """The side channel the ranks decide on (collective over the world, once, right after init_process_group)."""
    import datetime
    import torch.distributed as dist
    try:
        _AGREE["group"] = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=AGREE_S + 30))
    except Exception as ex:                                         # noqa: BLE001 -- the default group then carries the decisions
        log("no gloo side channel (%r): decisions travel on the default group" % (ex,))
        _AGREE["group"] = None
    return _AGREE["group"]


def _agree(code, device):
    """max over all ranks of `code` (0 = fine, 1 = failed, 2 = hung), bounded by AGREE_S; WorldLost if that cannot be had."""
    import torch
    import torch.distributed as dist
    g = _AGREE["group"]
    on_cpu = g is not None or dist.get_backend() == "gloo"
    t = torch.tensor([float(code)], device="cpu" if on_cpu else device)

    def run():
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=g)
        return float(t.item())
    done, r = _guarded(run, AGREE_S, None if on_cpu else (torch.cuda.current_device() if torch.cuda.is_available() else None))
    if not done or isinstance(r, BaseException):
        raise WorldLost("the ranks could not agree on a layout: %r" % (r,))
    return int(r)


def _outcome(done, r):
    """(finished, result) of a guarded self-test -> (code, message)"""
    if not done:
        return 2, "hung: %s" % r
    if isinstance(r, BaseException):
        return 1, repr(r)
    return 0, ""


def _fresh_groups(world, degree):
    """New process groups for the world's world / degree runs of `degree` consecutive ranks (collective over ALL ranks, same order
    everywhere) -> the list; degree == world: one group of everybody."""
    import torch.distributed as dist
    return [dist.new_group(list(range(i * degree, (i + 1) * degree))) for i in range(world // degree)]


def _leave_world(rank, world, argv_tail, reason):
    """Give the multi-GPU run up.  Ranks > 0 leave with status 0 (a non-zero exit makes torch.distributed.run kill rank 0 too); rank 0
    becomes a short single-GPU bench whose line carries `layout_fell_back_to`."""
    print("[bench rank %d] giving the %d-GPU run up: %s" % (rank, world, reason), file=sys.stderr, flush=True)
    if rank != 0:
        os._exit(0)
    env = {k: v for k, v in os.environ.items() if k not in _TORCHRUN_VARS}
    env["WAN_BENCH_FELL_BACK"] = json.dumps({"requested_gpus": world, "reason": reason})
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--no-e2e", "--no-secondary", "--no-robustness", "--no-configs3", "--no-config5",
           "--no-cpu-baseline", "--simulate-world", ""] + argv_tail
    sys.stderr.flush()
    os.execve(sys.executable, cmd, env)


class _Deadline(threading.Thread):
    """One clock for the whole multi-GPU part: not disarmed WORLD_DEADLINE_S after the process started -> _leave_world."""

    def __init__(self, seconds, fire):
        super().__init__(daemon=True)
        self.seconds, self.fire, self.done = seconds, fire, threading.Event()

    def run(self):
        left = self.seconds - (time.perf_counter() - T_PROCESS0)
        if not self.done.wait(max(left, 1.0)):
            self.fire("not through the timed region %.0f s after the process started (WAN_BENCH_WORLD_DEADLINE_S)" % self.seconds)

    def disarm(self):
        self.done.set()


def log(msg):
    """Progress on stderr (stdout carries exactly one JSON line)."""
    if int(os.environ.get("RANK", 0)) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("WAN_BENCH_WORKLOAD", "14B-720p"), choices=list(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the VAE decode / end-to-end block")
    ap.add_argument("--no-secondary", action="store_true", help="skip the 1.3B-480p generate() line")
    ap.add_argument("--fp8", action="store_true", help="scaled-fp8 checkpoint: block Linears on the fp8 MFMA (BASELINE configs[4])")
    ap.add_argument("--mixed-precision", action="store_true", help="the reference's mixed_precision_transformer plan (wgp.py:4039): time MLP, time projection "
                    "and norm3 in fp32 -> fp32 residual stream and modulation between bf16 Linears (csrc/mixed_ops.hip, DESIGN.md section 4.6); an option of "
                    "the reference, not the headline configuration")
    ap.add_argument("--no-robustness", action="store_true", help="skip the self-attention launches on gain-12 and adversarial inputs (roofline.robustness)")
    ap.add_argument("--no-configs3", action="store_true", help="skip the BASELINE configs[3] block (14B 720p x 161 frames, L = 147,600: 3 steps + a simulated rank of 8)")
    ap.add_argument("--e2e-full", action="store_true", help="after the timed region: ONE fully measured video of this workload -- tokenizer + UMT5-XXL "
                    "text encoding (random weights, the committed tokenizer fixture) of a positive and a negative prompt -> noise -> 30 guided steps -> VAE decode -> "
                    "uint8 frames on the host, wall clock (e2e_full; ~4.5 minutes at 14B-720p: the builder's run, not the driver's default)")
    ap.add_argument("--no-s1", action="store_true", help="skip the guidance-1 block (one stream per forward, 4 steps: the regime of the reference's Wan2.2 "
                    "lightning profiles, profiles/wan_2_2/*.json)")
    ap.add_argument("--no-config5", action="store_true", help="skip the BASELINE configs[4] block (i2v 14B, scaled-fp8 weights, VAE encode + decode)")
    ap.add_argument("--parallelism", default="auto", choices=["auto", "sp", "cfg-sp", "ulysses", "cfg-ulysses"],
                    help="N > 1: 'sp' = the token axis over all N ranks, both CFG streams on every rank; 'cfg-sp' = the conditional stream on "
                         "ranks [0, N/2), the unconditional one on [N/2, N), the token axis over the N/2 ranks of a half, one 2-rank swap of "
                         "the predictions per step (wan2gp_amd/sp.py CfgParallel); 'ulysses' / 'cfg-ulysses' = the same two layouts with the "
                         "per-block exchange as four all-to-alls of q, k, v^T, o (heads sharded inside the attention, WAN_SP_ULYSSES; the head "
                         "count must divide by the sequence-parallel degree) instead of the K / V^T all-gathers; 'auto' = what the link-modelled "
                         "one-GPU tables favour (choose_layout: N = 2 cfg-sp, N = 4 cfg-sp with all-gathers, N >= 8 ulysses over all ranks where the "
                         "heads divide, else cfg-ulysses / cfg-sp; DESIGN.md section 7)")
    ap.add_argument("--extras-budget-s", type=float, default=float(os.environ.get("WAN_BENCH_EXTRAS_BUDGET_S", 900)),
                    help="an OPTIONAL block behind the timed region (secondary workload, simulated ranks, config 5) is skipped -- and says so "
                         "in its place -- when the process is already older than this; the headline measurement, roofline and cpu_baseline never are")
    ap.add_argument("--simulate-layout", default="all", choices=["sp", "cfg-sp", "both", "ulysses", "cfg-ulysses", "all"],
                    help="which rank the simulated-ranks block runs: 'sp' = both streams at L / N rows (--parallelism sp), 'cfg-sp' = one stream "
                         "at L / (N/2) rows + the per-step swap as a device-to-device copy (--parallelism cfg-sp); 'both' = a row for each; 'ulysses' / "
                         "'cfg-ulysses' = the same shards with the all-to-all exchange; 'all' = the four of them")
    ap.add_argument("--simulate-world", default="2,4,8", help="comma-separated world sizes (e.g. 2,4,8): after the timed region, run ONE "
                    "rank's shard of a sequence-parallel world of that size on this GPU, the K / V^T all-gathers replaced by "
                    "device-to-device copies of the bytes that rank would receive -> compute-side upper bound of the scaling curve; '' = skip")
    ap.add_argument("--simulate-link-GBs", type=float, default=50.0,
                    help="the link model of the simulated-ranks block: every exchange a simulated rank issues is followed, on its side stream, by "
                         "a delay of (bytes from ONE peer) / this rate -- xGMI is a full mesh of point-to-point links, every peer's share moves over "
                         "a link of its own (MI355X: 7 x ~77 GB/s per direction; 50 = a conservative achieved RCCL rate).  Every row then also "
                         "carries the link-modelled step time / efficiency and the exposed exchange time per block; 0 = compute side only")
    ap.add_argument("--sp-chunks", type=int, default=0, help="N > 1, Ulysses exchange: head chunks of the q / o all-to-alls (wan_sp_info.a2a_chunks); "
                    "0 = the library default (2 when a rank holds >= 4 heads)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher of N ranks (one per GPU, RCCL, rendezvous on 127.0.0.1)
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))

    import torch
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if args.gpus != world:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if torch.cuda.device_count() < world and world > 1:
        sys.exit(f"bench.py: {world} ranks need {world} GPUs on this node, found {torch.cuda.device_count()}")
    _cfg, (_f, _h, _w), _ = WORKLOADS[args.workload]
    _L = _f * (_h // 2) * (_w // 2)
    if args.parallelism in ("cfg-sp", "cfg-ulysses") and world > 1 and world % 2:
        sys.exit(f"bench.py: --parallelism {args.parallelism} splits the ranks in two halves: {world} is odd")
    cfg_sp, sp_degree, sp_mode = choose_layout(world, args.parallelism, _cfg["num_heads"])
    if _L % sp_degree:
        sys.exit(f"bench.py: the {_L} tokens of workload {args.workload} do not shard over {sp_degree} sequence-parallel ranks "
                 f"(divisors: 2, 4, 8 ...)")
    if sp_mode == "ulysses" and _cfg["num_heads"] % sp_degree:
        sys.exit(f"bench.py: --parallelism {args.parallelism} shards the {_cfg['num_heads']} heads of workload {args.workload} over "
                 f"{sp_degree} sequence-parallel ranks: not divisible")
    torch.cuda.set_device(local)
    deadline = None
    # what a single-GPU re-run of this command keeps of its arguments (_leave_world)
    argv_tail = ["--workload", args.workload, "--steps", str(min(args.steps, 5)), "--warmup", str(min(args.warmup, 1))] + \
                (["--fp8"] if args.fp8 else []) + (["--mixed-precision"] if args.mixed_precision else [])
    if world > 1:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
        deadline = _Deadline(WORLD_DEADLINE_S + 10.0 * max(args.steps + args.warmup - 25, 0), lambda why: _leave_world(rank, world, argv_tail, why))
        deadline.start()
        try:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=datetime.timedelta(seconds=180))
            assert dist.get_world_size() == args.gpus and dist.get_backend() == "nccl"          # RCCL sees every rank
            make_agreement_group()

            def first_collective():
                chk = torch.ones(1, device="cuda")
                dist.all_reduce(chk)
                if int(chk.item()) != world:
                    raise RuntimeError("RCCL all-reduce saw %d of %d ranks" % (int(chk.item()), world))
            done, r = _guarded(first_collective, GUARD_S, local)
            if _agree(_outcome(done, r)[0], "cuda") != 0:
                raise WorldLost("the first RCCL all-reduce %s" % ("hung" if not done else "failed: %r" % (r,)))
        except WorldLost as ex:
            _leave_world(rank, world, argv_tail, str(ex))
        except Exception as ex:                                  # noqa: BLE001 -- init_process_group itself
            _leave_world(rank, world, argv_tail, "init_process_group / first collective: %r" % (ex,))

    from wan2gp_amd import lib as L_
    from wan2gp_amd.model import WanModelHIP
    from wan2gp_amd.rope import get_rotary_pos_embed
    from wan2gp_amd.schedulers import HipScheduler, cfg_combine

    cfg, (f, h, w), desc = WORKLOADS[args.workload]
    mcfg = {k: v for k, v in cfg.items()}
    L = f * (h // 2) * (w // 2)
    two_experts = args.workload in TWO_EXPERT_WORKLOADS
    i2v = cfg.get("in_dim", 16) == 36
    log(f"workload {args.workload}: building random-init weights")
    if args.mixed_precision:
        mcfg["mixed_precision"] = True            # load_state_dict registers the locked tensors in fp32: the library picks the plan from them
    model = random_weights(WanModelHIP(**mcfg), cfg, 1234, args.fp8)
    model2 = random_weights(WanModelHIP(**mcfg), cfg, 4321, args.fp8) if two_experts else None
    cfgp, layout_note = None, None
    if world > 1:
        try:
            cfgp, cfg_sp, sp_degree, layout_note = setup_parallel(rank, world, cfg_sp, L, (model, model2), args.parallelism in ("cfg-sp", "cfg-ulysses"),
                                                                  sp_mode=sp_mode, sp_mode_demanded=args.parallelism in ("ulysses", "cfg-ulysses"),
                                                                  chunks=args.sp_chunks or None)
        except WorldLost as ex:
            _leave_world(rank, world, argv_tail, str(ex))
        eff = cfgp.sp if cfgp is not None else model.sp        # the exchange the run ended up with (the self-test may have fallen back)
        sp_mode = eff.mode if eff is not None else "allgather"

    vae = None
    want_e2e = not args.no_e2e and rank == 0 and args.workload != "tiny"
    if want_e2e or i2v:
        from wan2gp_amd.vae import WanVAEHIP, random_vae_state_dict
        vae = WanVAEHIP(state_dict=random_vae_state_dict())

    g = torch.Generator(device="cuda").manual_seed(42)
    ctx = (torch.randn(1, 512, 4096, device="cuda", generator=g) * 0.5).to(torch.bfloat16); ctx[:, 77:] = 0
    ctx_null = (torch.randn(1, 512, 4096, device="cuda", generator=g) * 0.5).to(torch.bfloat16); ctx_null[:, 8:] = 0
    freqs = get_rotary_pos_embed((f, h, w), device="cuda")
    sched = HipScheduler("unipc", num_train_timesteps=1000)          # wan_sched_*: what generate() steps with on a GPU
    total_steps = args.steps + args.warmup
    sched.set_timesteps(max(VIDEO_STEPS, total_steps), device="cuda", shift=12.0)
    guide, switch_threshold = 4.0, 875
    lib = L_.load()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    # ---- the video starts here: (i2v: VAE-encode the conditioning clip,) noise ---------------------------------------------
    sync()
    t_video0 = time.perf_counter()
    y, enc_s = None, None
    if i2v:
        # any2video.py:739-774: start image + zero frames -> VAE encode -> y = cat(mask[4], latents[16]); every rank encodes
        img = torch.rand(3, 1, h * 8, w * 8, device="cuda", generator=g) * 2 - 1
        clip = torch.cat([img, torch.zeros(3, (f - 1) * 4, h * 8, w * 8, device="cuda")], dim=1)
        vae.encode([clip])                                   # (first call: allocator warm-up, not the encode -- see config5_block)
        torch.cuda.synchronize()
        te = time.perf_counter()
        lat_y = vae.encode([clip])[0]
        torch.cuda.synchronize()
        enc_s = time.perf_counter() - te
        msk = torch.zeros(4, f, h, w, device="cuda"); msk[:, 0] = 1
        y = torch.cat([msk, lat_y])
        del clip
    latents = torch.randn(1, 16, f, h, w, device="cuda", generator=g)

    par = {"cfgp": cfgp}           # (the simulated-ranks block swaps a stand-in in)

    def new_sched(n_steps=VIDEO_STEPS):
        """A scheduler of its own for every block behind the timed region: a block never depends on how many timesteps the
        timed region (or the block before it) consumed -- the native scheduler refuses to step past its last timestep."""
        sc = HipScheduler("unipc", num_train_timesteps=1000)
        sc.set_timesteps(n_steps, device="cuda", shift=12.0)
        return sc

    def one_step(i, lat, sc=None, fr=None):
        sc = sched if sc is None else sc
        t = sc.timesteps[i]
        trans = model2 if (model2 is not None and int(t) <= switch_threshold) else model
        fr = freqs if fr is None else fr
        if par["cfgp"] is not None:       # this rank's stream, then the 2-rank swap: every rank holds (cond, uncond) bit-identically
            cond, uncond = par["cfgp"].guided_pair(trans, lat, ctx, ctx_null, t=torch.stack([t]), freqs=fr, y=y)
        else:
            cond, uncond = trans([lat, lat], t=torch.stack([t]), context=[ctx, ctx_null], freqs=fr, y=y)
        noise = cfg_combine(cond, uncond, guide if trans is model else 3.0)
        return sc.step(noise, t, lat)[0]

    lat = latents
    log(f"{args.warmup} warm-up + {args.steps} timed steps")
    for i in range(args.warmup):
        lat = one_step(i, lat)
    sync()
    replayed = world == 1 and 2 * L <= model.graph_max_tokens and model.graph != "off"
    if not replayed:                  # (the event brackets around single launches keep a forward on the eager path: launch-bound shapes go without)
        lib.wan_prof_enable(1)
    t0 = time.perf_counter()
    for i in range(args.warmup, total_steps):
        lat = one_step(i, lat)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    assert torch.isfinite(lat).all(), "non-finite latents"
    if deadline is not None:
        deadline.disarm()

    # ---- ... and ends here: causal 3D VAE decode -> uint8 on the host (rank 0; latents are replicated) --------------------------
    log(f"timed region done: {dt / args.steps * 1e3:.1f} ms/step")
    e2e = None
    if want_e2e:
        td = time.perf_counter()
        video = vae.decode_to_cpu_uint8([lat[0]], 0)[0]
        torch.cuda.synchronize()
        t_end = time.perf_counter()
        dec_s = t_end - td
        assert video.dtype == torch.uint8 and tuple(video.shape) == (3, (f - 1) * 4 + 1, h * 8, w * 8) and not video.is_cuda
        step_s = dt / args.steps
        e2e = {"unit": "s/video", "video": [3, (f - 1) * 4 + 1, h * 8, w * 8],
               "measured_s": t_end - t_video0, "measured_sampling_steps": total_steps,
               "measured_note": "noise -> W+K sampler steps (the first includes one-time workspace allocation) -> VAE decode -> uint8 "
                                "on the host, wall clock of this run; text encoding excluded (synthetic context)",
               "vae_decode_to_host_s": dec_s, "vae_encode_s": enc_s,
               # ~6.4e14 FLOP per 720 x 1280 x 81-frame decode (convolutions), scaled by the pixel count; the time includes
               # RMS_norm / SiLU passes, float -> uint8 and the device -> host copy (round 4: the 3x3x3 convolutions on a halo patch)
               "vae_decode_TFLOPs": 6.4e14 * ((f - 1) * 4 + 1) * h * 8 * w * 8 / (81.0 * 720 * 1280) / dec_s / 1e12,
               "composed_s_at_%d_steps" % VIDEO_STEPS: VIDEO_STEPS * step_s + dec_s + (enc_s or 0.0),
               "composed_note": "%d x the step time of the timed region + the VAE times measured in this run" % VIDEO_STEPS}
    if world > 1:
        torch.distributed.barrier()

    import ctypes
    prof = collect_prof(lib)
    if rank == 0:
        S = 1 if cfg_sp else 2                                  # streams one launch of this rank carries
        d, ffn = cfg["dim"], cfg["ffn_dim"]
        Ll = L // sp_degree
        ms, n = prof["self_attn"]
        eff_sp = (cfgp.sp if cfgp is not None else model.sp) if world > 1 else None
        attn_chunks = eff_sp.resolved_chunks(cfg["num_heads"]) if eff_sp is not None else 1     # Ulysses: C launches per block, each 1 / C of the heads
        attn_flops = 4.0 * Ll * L * d * S / attn_chunks        # algorithmic FLOP of one launch (S streams; a head chunk's share)
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
        # HBM traffic of one launch: PMC counters cannot be read from inside this process; the figure is the committed
        # FETCH_SIZE + WRITE_SIZE pass (rocprofv3 --pmc, profiles/) of this kernel at this shape, NOT a measurement of this run
        traffic, traffic_source = None, None
        pmc = os.path.join(ROOT, "profiles", "attn_pmc_traffic.json")
        if os.path.isfile(pmc) and world == 1:
            try:
                ent = json.load(open(pmc)).get(args.workload.replace("i2v-", ""), {})
                traffic = ent.get("traffic_bytes")
                traffic_source = ent.get("source", "profiles/attn_pmc_traffic.json") + " -- separate rocprofv3 --pmc pass, not measured in this run"
            except Exception:
                traffic = None
        # the ceiling THIS box reaches in THIS state: a bare MFMA loop on random bf16, no memory traffic (csrc/probe.hip) -- the
        # data sheet's 2.5 PFLOP/s assumes 2.4 GHz; under random data the chip settles at its power limit well below it
        sustained = None
        try:
            fl = ctypes.c_double()
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            L_.check(lib.wan_mfma_sustained_probe(20000, ctypes.byref(fl), L_.stream_ptr()), "probe")      # warm-up / clock settles
            for a, b in ((0, 1), (2, 3)):
                evs[a].record()
                L_.check(lib.wan_mfma_sustained_probe(40000, ctypes.byref(fl), L_.stream_ptr()), "probe")
                evs[b].record()
            torch.cuda.synchronize()
            probe_ms = min(evs[0].elapsed_time(evs[1]), evs[2].elapsed_time(evs[3]))
            sustained = {"TFLOPs": fl.value / (probe_ms * 1e-3) / 1e12, "ms": probe_ms,
                         "what": "v_mfma_f32_32x32x16_bf16 back to back, one wave per SIMD on every CU, random bf16 operands in registers, no "
                                 "memory traffic: the power-limited rate of this box (wan_mfma_sustained_probe)"}
        except Exception as ex:            # never lose the bench line to the probe
            sustained = {"error": str(ex)}
        declined, launched = ctypes.c_int64(), ctypes.c_int64()
        L_.check(lib.wan_prof_attention_declined(ctypes.byref(declined), ctypes.byref(launched)), "wan_prof_attention_declined")
        out = {
            "metric": "denoise-steps/s", "value": args.steps / dt, "unit": "steps/s", "n_gpus": world, "world": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": ("fp8-e4m3 block Linears (weights + dynamically quantised activations), bf16 elsewhere" if args.fp8 else "bf16") +
                     (" -- mixed_precision_transformer: fp32 residual stream / modulation between the Linears" if args.mixed_precision else ""), "data": "synthetic",
            "config": {"workload": desc, "latent": [16, f, h, w], "tokens": L, "streams": 2, "guide_scale": guide,
                       "solver": "unipc", "parallelism": (("cfg2 x sp%d" % sp_degree if cfg_sp else "sp%d" % world) + ((" (ulysses, %d head chunks)" % attn_chunks if attn_chunks > 1 else " (ulysses)") if sp_mode == "ulysses" else "")) if world > 1 else "single",
                       **({"parallelism_note": layout_note} if layout_note else {}),
                       "forward_TFLOP": forward_flops(cfg, L) / 1e12},
            "roofline": {"kernel": "attn_w16n_kernel (self-attention: the bounded loop on the 16x16x32 MFMA)", "bound": "mfma", "achieved": achieved,
                         "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_TFLOPS,
                         "traffic": traffic, "traffic_source": traffic_source, "launches": n, "avg_ms": ms / n if n else None,
                         "flop_per_launch": attn_flops,
                         # data-dependent loop choice: workgroups of the timed launches whose rows failed the score bound
                         # |q~_row| max|k_h| <= 96 and ran the tracking loop instead of the bounded one (read after the timed region)
                         "declined_workgroups": declined.value, "total_workgroups": launched.value,
                         "declined_frac": declined.value / launched.value if launched.value else None,
                         "sustained_mfma": sustained,
                         "frac_of_sustained_mfma": (achieved / sustained["TFLOPs"]) if sustained and sustained.get("TFLOPs") else None,
                         "other_kernels": kern},
            "step_TFLOPs": 2 * forward_flops(cfg, L) / (dt / args.steps) / 1e12,
            "forwards_per_s": 2 * args.steps / dt,          # a CFG step is two forwards (SURVEY.md section 8d reports both)
        }
        # first-run hardening: null = the layout asked for (or chosen by `auto`) ran; else what the run ended up on, and why
        out["layout_fell_back_to"] = out["config"]["parallelism"] if layout_note else None
        if os.environ.get("WAN_BENCH_FELL_BACK"):
            fb = json.loads(os.environ["WAN_BENCH_FELL_BACK"])
            out["layout_fell_back_to"] = "single GPU (rank 0 of the %d-GPU launch, alone)" % fb["requested_gpus"]
            out["requested_gpus"] = fb["requested_gpus"]
            out["fell_back_reason"] = fb["reason"]
        if replayed:
            out["forwards"] = {"mode": "replayed launch lists (wan_dit_forward_graph)" if model.last_graph_how == 3 else "eager",
                               "last_forward_how": int(model.last_graph_how),
                               "note": "launch-bound shape: no per-launch event brackets (they would keep the forward eager), roofline.achieved is not measured here"}
        lib.wan_prof_enable(0)
        if e2e is not None:
            out["e2e"] = e2e
        if world == 1 and not args.no_robustness and args.workload in TWO_EXPERT_WORKLOADS:
            # what the headline's roofline.frac depends on: nothing a normalised head produces (the shifted loop), and the price of the
            # worst case (every workgroup redone by the tracking loop)
            log("attention on gain-1 / gain-12 / adversarial inputs")
            rb = _extra_block(attention_robustness, cfg, L, budget_s=args.extras_budget_s)
            out["roofline"]["robustness"] = rb
            if isinstance(rb, dict) and "adversarial_key_all_redone_by_tracking_loop" in rb:
                out["roofline"]["frac_gain_12"] = rb["gain_12_shifted_loop"]["frac"]
                out["roofline"]["frac_all_declined"] = rb["adversarial_key_all_redone_by_tracking_loop"]["frac"]
        if world == 1 and not args.no_s1 and args.workload in TWO_EXPERT_WORKLOADS:
            log("s1: guidance 1, one stream per forward, 1 warm-up + 4 timed steps")
            out["s1"] = _extra_block(s1_block, model, model2, latents, ctx, freqs, y, new_sched, e2e, budget_s=args.extras_budget_s)
        if world == 1 and args.e2e_full and vae is not None:
            log("e2e_full: tokenizer + UMT5-XXL encode -> %d guided steps -> VAE decode -> uint8 on the host, measured" % VIDEO_STEPS)
            out["e2e_full"] = _extra_block(e2e_full_block, model, model2, vae, freqs, y, (f, h, w), new_sched, guide, switch_threshold)
        if world == 1 and not args.no_secondary and args.workload in TWO_EXPERT_WORKLOADS:
            log("secondary: 1.3B-480p generate(), 30 steps + VAE decode")
            out["secondary"] = _extra_block(secondary_1p3b, vae, budget_s=args.extras_budget_s)
        if world == 1 and args.simulate_world and args.workload in TWO_EXPERT_WORKLOADS:
            log("simulated sequence-parallel ranks: " + args.simulate_world)
            out["simulated_scaling"] = _extra_block(simulate_world, [int(v) for v in args.simulate_world.split(",") if v], model, model2, one_step,
                                                    latents, new_sched, dt / args.steps, cfg, L, par, args.simulate_layout, None,
                                                    1 if args.simulate_layout == "all" else 2, args.simulate_link_GBs, args.sp_chunks or None,
                                                    budget_s=args.extras_budget_s)
        if world == 1 and not args.no_configs3 and args.workload == "14B-720p":
            log("configs3: 14B 720p x 161 frames (L = 147,600), 1 warm-up + 2 timed steps, simulated rank of a world of 8")
            out["configs3"] = _extra_block(configs3_block, model, model2, one_step, new_sched, par, lib, args.simulate_link_GBs, budget_s=args.extras_budget_s)
        if world == 1 and not args.no_config5 and args.workload == "14B-720p" and not args.fp8:
            log("config5: i2v 14B, scaled-fp8 weights, VAE encode + 3 steps + decode")
            model = model2 = None                       # the bf16 experts of the main workload are done
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            out["config5"] = _extra_block(config5_block, vae, budget_s=args.extras_budget_s)
        if not args.no_cpu_baseline and world == 1:
            # LAST and alone: nothing runs on the GPU beside it (round 3 ran it next to the config-5 block: its 128 host threads cost
            # launch-dense GPU blocks 5-8 %, and the GPU's launching thread cost the CPU leg cores)
            log("cpu_baseline: thread sweep, then the config-1 oracle step + VAE decode on the host cores")
            same = None
            if args.workload in TWO_EXPERT_WORKLOADS:
                log("configs[0] on the GPU (the configuration the CPU leg is quoted on): generate() 10 steps + decode, replayed and eager forwards")
                model = model2 = None
                import gc
                gc.collect()
                torch.cuda.empty_cache()
                same = _extra_block(configs0_gpu, vae, budget_s=args.extras_budget_s + 120)
            try:
                out["cpu_baseline"] = cpu_baseline(2 * forward_flops(cfg, L))
            except Exception as ex:                      # noqa: BLE001 -- reported in place, never costs the line
                out["cpu_baseline"] = {"error": repr(ex)}
            if same is not None:
                out["cpu_baseline"]["gpu_same_config"] = same
                if isinstance(same, dict) and "step_s" in same and "step_s" in out["cpu_baseline"]:
                    out["cpu_baseline"]["gpu_same_config_step_s"] = same["step_s"]
                    out["cpu_baseline"]["gpu_same_config_e2e_s"] = same["e2e_s"]
                    out["cpu_baseline"]["gpu_over_cpu_same_config"] = {"step": out["cpu_baseline"]["step_s"] / same["step_s"],
                                                                       "e2e_composed_cpu_over_measured_gpu": out["cpu_baseline"]["e2e_s_per_video_composed"] / same["e2e_s"]}
        log("done")
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def attention_robustness(cfg, L, S=2):
    """The self-attention launch of the headline shape on three kinds of input, standalone (HIP events, 1 warm-up + 2 timed launches
    each): RMS-normalised-like rows at gain 1 (every workgroup takes the plain bounded loop -- the state the timed region's synthetic
    weights produce), the same at K gain 12 (round 3: every workgroup declined to the tracking loop; round 4: the SAME loop with a
    per-row reference shift, attention_w16n.hip SHIFT), and an adversarial key in every head (a score ~ 300 log2 units where the
    first tile suggests ~ 85: P overflows, the row sums say so, every workgroup is redone by the tracking loop after a wasted pass)
    -- the worst case of the protocol.  -> TFLOP/s, fraction of 2.5 PFLOP/s and the share of workgroups that reached the tracking loop."""
    import ctypes
    import torch
    from wan2gp_amd import lib as L_, ops
    lib = L_.load()
    H = cfg["num_heads"]
    g = torch.Generator(device="cuda").manual_seed(7)
    q = (torch.randn(S, L, H, 128, device="cuda", generator=g) * ops.attention_qscale()).to(torch.bfloat16)
    k = torch.randn(S, L, H, 128, device="cuda", generator=g).to(torch.bfloat16)
    ldv = (L + 63) // 64 * 64
    vt = torch.zeros(S, H * 128, ldv, device="cuda", dtype=torch.bfloat16)
    vt[..., :L] = torch.randn(S, H * 128, L, device="cuda", generator=g).to(torch.bfloat16)
    scratch = torch.zeros(ops.attention_scratch_words(S, S, L, H), dtype=torch.float32, device="cuda")
    acc = torch.zeros(2, dtype=torch.int64, device="cuda")
    out = torch.empty_like(q)
    flops = 4.0 * S * L * L * H * 128
    res = {}

    def run(name, qq, kk):
        ops.attention(qq, kk, vt, q_prescaled=True, kmax_scratch=scratch, out=out)
        acc.zero_()
        L_.check(lib.wan_attention_count_declined(L_.ptr(scratch), S, S, L, H, L_.ptr(acc), L_.stream_ptr()), "count")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(2):
            ops.attention(qq, kk, vt, q_prescaled=True, kmax_scratch=scratch, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 2
        tf = flops / (ms * 1e-3) / 1e12
        res[name] = {"ms": ms, "TFLOPs": tf, "frac": tf / PEAK_BF16_TFLOPS, "reached_tracking_loop_frac": int(acc[0]) / max(int(acc[1]), 1),
                     "finite": bool(torch.isfinite(out.float()).all())}
    run("gain_1_plain_loop", q, k)
    k12 = (k.float() * 12.0).to(torch.bfloat16)
    run("gain_12_shifted_loop", q, k12)
    del k12
    qa = q.clone(); qa[..., 0] += 6.0 * ops.attention_qscale()
    ka = k.clone(); ka[:, 12345] = 0; ka[:, 12345, :, 0] = 400.0
    run("adversarial_key_all_redone_by_tracking_loop", qa, ka)
    return res


def collect_prof(lib):
    """{class: (total ms, launches)} of the launches bracketed with HIP events on the launch stream since wan_prof_enable(1)."""
    import ctypes
    from wan2gp_amd import lib as L_
    prof = {}
    for cls, name in ((0, "self_attn"), (1, "cross_attn"), (2, "ffn_gemm_pair"), (3, "rmsnorm_rope")):
        ms, n = ctypes.c_double(), ctypes.c_int()
        L_.check(lib.wan_prof_collect(cls, ctypes.byref(ms), ctypes.byref(n)), "wan_prof_collect")
        prof[name] = (ms.value, n.value)
    return prof


def configs3_block(model, model2, one_step, new_sched, par, lib, link_GBs=50.0):
    """BASELINE configs[3] on one GPU: Wan2.2 t2v 14B, 720 x 1280 x 161 frames -> 41 latent frames (any2video.py:647,1166),
    L = 147,600 tokens, on the resident experts of the main workload.  1 warm-up + 2 timed CFG steps with a scheduler of their own;
    self-attention's roofline at this L from HIP events like the headline's; then rank 0 of a world of 8 (both layouts) on this GPU
    -- the compute side of the configuration BASELINE quotes at 8 GPUs."""
    import torch
    from wan2gp_amd.rope import get_rotary_pos_embed
    cfg, (f, h, w), desc = WORKLOADS["14B-720p-161f"]
    L = f * (h // 2) * (w // 2)
    freqs = get_rotary_pos_embed((f, h, w), device="cuda")
    g = torch.Generator(device="cuda").manual_seed(44)
    latents = torch.randn(1, 16, f, h, w, device="cuda", generator=g)
    sc = new_sched()
    lat = one_step(0, latents, sc, freqs)                                    # warm-up: the workspace grows to this L
    torch.cuda.synchronize()
    lib.wan_prof_enable(1)
    t0 = time.perf_counter()
    k = 2
    for i in range(k):
        lat = one_step(1 + i, lat, sc, freqs)
    torch.cuda.synchronize()
    step_s = (time.perf_counter() - t0) / k
    assert torch.isfinite(lat).all()
    ms, n = collect_prof(lib)["self_attn"]
    lib.wan_prof_enable(0)
    attn_flops = 4.0 * L * L * cfg["dim"] * 2
    achieved = attn_flops / (ms / n * 1e-3) / 1e12 if n else 0.0
    out = {"workload": desc, "latent": [16, f, h, w], "tokens": L, "metric": "denoise-steps/s", "value": 1.0 / step_s,
           "ms_per_step": step_s * 1e3, "steps": k, "warmup": 1, "step_TFLOPs": 2 * forward_flops(cfg, L) / step_s / 1e12,
           "composed_s_at_%d_steps_without_vae" % VIDEO_STEPS: VIDEO_STEPS * step_s,
           "roofline": {"kernel": "self-attention", "bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                        "frac": achieved / PEAK_BF16_TFLOPS, "launches": n, "avg_ms": ms / n if n else None, "flop_per_launch": attn_flops}}
    del lat
    # (the two layouts a world of 8 would choose between: the halves' exchange as all-gathers or as chunked all-to-alls)
    out["simulated_scaling"] = simulate_world([8], model, model2, one_step, latents, new_sched, step_s, cfg, L, par, "cfg-both", fr=freqs, k=1, link_GBs=link_GBs)
    return out


def setup_parallel(rank, world, cfg_sp, L, models, cfg_sp_demanded=False, device="cuda", sp_mode="allgather", sp_mode_demanded=False, chunks=None):
    """The multi-GPU layout of this run on the resident experts -> (CfgParallel | None, cfg_sp, sp_degree, note).

    cfg-sp (the default for an even world) is tried first: groups, then a self-test of the 2-rank swap on a tiny tensor.  Every self-test
    runs under a wall-clock guard (_guarded) and the ranks agree on its outcome over the side channel (_agree): if ANY rank failed or
    hung, EVERY rank takes plain sequence parallelism over the whole world instead (when the token count shards that way) and the JSON
    line carries the reason -- a scaling run is worth more than the preferred layout.  With `--parallelism cfg-sp` given explicitly the
    failure is fatal instead.  The same agreement in front of the Ulysses exchange (sp_mode "ulysses": _ulysses_self_test), whose
    fall-back is the all-gather exchange, and last in front of the all-gather itself (_allgather_self_test) -- if that fails there is no
    multi-GPU layout left: WorldLost (main: rank 0 alone, one GPU).  After a HANG the layout that follows runs on fresh process groups."""
    import torch
    from wan2gp_amd.sp import CfgParallel, SequenceParallel
    note = None
    poisoned = False
    dev_id = torch.cuda.current_device() if (str(device) != "cpu" and torch.cuda.is_available()) else None
    if cfg_sp:
        def st():
            c = CfgParallel(rank, world, mode=sp_mode, chunks=chunks)
            mine = torch.full((8,), float(c.stream), device=device)
            a, b = c.exchange(mine)
            if not (bool((a == 0).all()) and bool((b == 1).all())):
                raise RuntimeError("the 2-rank swap returned (%r, %r), not (conditional, unconditional)" % (a.tolist(), b.tolist()))
            return c
        done, r = _guarded(st, GUARD_S, dev_id)
        code, err = _outcome(done, r)
        verdict = _agree(code, device)
        if verdict == 0:
            cfgp = r
            note = _ulysses_self_test(cfgp.sp, device, sp_mode_demanded, None, world)
            note = _allgather_self_test(cfgp.sp, device, note)
            cfgp.attach(*models)                                # this rank's stream; the half's sequence-parallel group on the experts
            return cfgp, True, world // 2, note
        poisoned = verdict == 2
        note = "cfg-sp setup %s on %s: %s -- fell back to sequence parallelism over all %d ranks" % (
            "hung" if verdict == 2 else "failed", "this rank" if err else "another rank", err or "(see that rank's log)", world)
        log(note)
        if cfg_sp_demanded or L % world:
            sys.exit("bench.py: " + note + (" refused: --parallelism cfg-sp was asked for" if cfg_sp_demanded else
                                            " impossible: %d tokens do not shard over %d ranks" % (L, world)))
    sp = SequenceParallel(rank, world, group=_fresh_groups(world, world)[0] if poisoned else None, mode=sp_mode, chunks=chunks)
    note = _ulysses_self_test(sp, device, sp_mode_demanded, note, world)
    note = _allgather_self_test(sp, device, note)
    for m in models:
        if m is not None:
            m.sp = sp
    return None, False, world, note


def _sp_group_index(sp, world):
    """which of the world's world / sp.world runs of consecutive ranks `sp` shares its token axis with"""
    import torch.distributed as dist
    return 0 if sp.world == world else dist.get_rank() // sp.world


def _ulysses_self_test(sp, device, demanded, note, world=None):
    """The Ulysses exchange before its first use on this node: one tiny all-to-all on the sequence-parallel group (chunk j must arrive
    from rank j) under the wall-clock guard, every rank reports over the side channel; if ANY rank failed or hung EVERY rank keeps the
    all-gather exchange (sp.mode) and the line says so -- unless the mode was asked for explicitly, then the failure is fatal.  After a
    hang the group gets replaced (its all-to-all will never be matched)."""
    import torch
    import torch.distributed as dist
    if sp is None or sp.mode != "ulysses" or sp.world < 2:
        return note
    world = world or dist.get_world_size()
    dev_id = torch.cuda.current_device() if (str(device) != "cpu" and torch.cuda.is_available()) else None

    def st():
        if int(os.environ.get("WAN_BENCH_INJECT_A2A_HANG_RANK", -1)) == dist.get_rank():     # tests: this rank never enters the exchange
            time.sleep(float(os.environ.get("WAN_BENCH_INJECT_A2A_HANG_S", 1e9)))
            raise RuntimeError("injected: woke up after the exchange was given up")
        send = (torch.arange(sp.world * 4, device=device, dtype=torch.float32) // 4) * 0 + float(sp.rank)
        recv = torch.full_like(send, -1.0)
        if dist.get_backend(sp.group) == "gloo":
            hs, hr = send.cpu(), recv.cpu()
            dist.all_to_all_single(hr, hs, group=sp.group)
            recv = hr.to(device)
        else:
            dist.all_to_all_single(recv, send, group=sp.group)
        want = torch.arange(sp.world, device=device, dtype=torch.float32).repeat_interleave(4)
        if not torch.equal(recv, want):
            raise RuntimeError("all-to-all returned %r" % recv.tolist())
    done, r = _guarded(st, GUARD_S, dev_id)
    code, err = _outcome(done, r)
    verdict = _agree(code, device)                              # over the WORLD: both halves of a cfg layout decide together
    if verdict != 0:
        msg = "the Ulysses all-to-all self-test %s on %s: %s -- the per-block exchange stays the K / V^T all-gather" % (
            "hung" if verdict == 2 else "failed", "this rank" if err else "another rank", err or "(see that rank's log)")
        if demanded:
            sys.exit("bench.py: " + msg + " refused: the mode was asked for")
        log(msg)
        sp.mode = "allgather"
        if verdict == 2:
            sp.group = _fresh_groups(world, sp.world)[_sp_group_index(sp, world)]
        note = (note + "; " if note else "") + msg
    return note


def _allgather_self_test(sp, device, note):
    """What every layout rests on -- one tiny all-gather over the sequence-parallel group, guarded and agreed like the others.  A failure
    here leaves no multi-GPU layout to fall back to: WorldLost."""
    import torch
    import torch.distributed as dist
    if sp is None or sp.world < 2:
        return note
    dev_id = torch.cuda.current_device() if (str(device) != "cpu" and torch.cuda.is_available()) else None

    def st():
        got = sp.all_gather(torch.full((1, 4), float(sp.rank), device=device))
        want = torch.arange(sp.world, device=device, dtype=torch.float32).repeat_interleave(4).view(sp.world, 4)
        if not torch.equal(got, want):
            raise RuntimeError("all-gather returned %r" % got.tolist())
    done, r = _guarded(st, GUARD_S, dev_id)
    code, err = _outcome(done, r)
    verdict = _agree(code, device)
    if verdict != 0:
        raise WorldLost("the all-gather self-test %s on %s: %s" % ("hung" if verdict == 2 else "failed", "this rank" if err else "another rank",
                                                                   err or "(see that rank's log)"))
    return note


def _extra_block(fn, *a, budget_s=None):
    """The blocks behind the headline measurement (secondary workload, simulated ranks, config 5) must never cost the bench its JSON
    line: a failure is recorded in the block's place, and a block that would START after `budget_s` seconds of process time (a slow
    box, a long --steps) is skipped with a note instead of pushing the run past whatever limit its caller has."""
    age = time.perf_counter() - T_PROCESS0
    if budget_s is not None and age > budget_s:
        log("block %s skipped: process is %.0f s old (--extras-budget-s %.0f)" % (getattr(fn, "__name__", "?"), age, budget_s))
        return {"skipped": "process was %.0f s old when this optional block would have started (--extras-budget-s %.0f)" % (age, budget_s)}
    try:
        return fn(*a)
    except Exception as ex:  # noqa: BLE001 -- reported, not swallowed
        import traceback
        log("block %s failed: %r" % (getattr(fn, "__name__", "?"), ex))
        return {"error": repr(ex), "traceback": traceback.format_exc()[-1500:]}


def config5_block(vae):
    """BASELINE configs[4] on one GPU: Wan2.2 i2v 14B 720x1280x81f with a scaled-fp8 checkpoint (block Linears on the fp8 MFMA,
    shared/qtypes/scaled_fp8.py), both experts resident; the conditioning clip goes through the VAE encoder, the result through
    the decoder.  1 warm-up + 2 timed CFG steps; the 30-step video is composed from them like the main block's."""
    import torch
    from wan2gp_amd.model import WanModelHIP
    from wan2gp_amd.rope import get_rotary_pos_embed
    from wan2gp_amd.schedulers import HipScheduler, cfg_combine
    cfg, (f, h, w), desc = WORKLOADS["i2v-14B-720p"]
    m1 = random_weights(WanModelHIP(**cfg), cfg, 1234, True)
    m2 = random_weights(WanModelHIP(**cfg), cfg, 4321, True)
    if vae is None:
        from wan2gp_amd.vae import WanVAEHIP, random_vae_state_dict
        vae = WanVAEHIP(state_dict=random_vae_state_dict())
    g = torch.Generator(device="cuda").manual_seed(43)
    ctx = (torch.randn(1, 512, 4096, device="cuda", generator=g) * 0.5).to(torch.bfloat16); ctx[:, 77:] = 0
    ctx_null = (torch.randn(1, 512, 4096, device="cuda", generator=g) * 0.5).to(torch.bfloat16); ctx_null[:, 8:] = 0
    freqs = get_rotary_pos_embed((f, h, w), device="cuda")
    sched = HipScheduler("unipc", num_train_timesteps=1000)
    sched.set_timesteps(VIDEO_STEPS, device="cuda", shift=12.0)
    # the conditioning clip is BUILT first, then the encode is timed alone -- twice: the first call after the empty_cache() in front of this
    # block pays the first-touch allocation of its ~10 GB of activations (round 4's driver line: 4.01 s against 0.45 s in every builder
    # run, where something had warmed the allocator), the second is the encode (what a server pays per video)
    img = torch.rand(3, 1, h * 8, w * 8, device="cuda", generator=g) * 2 - 1
    clip = torch.cat([img, torch.zeros(3, (f - 1) * 4, h * 8, w * 8, device="cuda")], dim=1)
    torch.cuda.synchronize()
    te = time.perf_counter()
    lat_y = vae.encode([clip])[0]
    torch.cuda.synchronize()
    enc_first_s = time.perf_counter() - te
    te = time.perf_counter()
    lat_y = vae.encode([clip])[0]
    torch.cuda.synchronize()
    enc_s = time.perf_counter() - te
    msk = torch.zeros(4, f, h, w, device="cuda"); msk[:, 0] = 1
    y = torch.cat([msk, lat_y])
    del clip
    lat = torch.randn(1, 16, f, h, w, device="cuda", generator=g)

    def step(i, lat):
        t = sched.timesteps[i]
        trans = m2 if int(t) <= 875 else m1
        cond, uncond = trans([lat, lat], t=torch.stack([t]), context=[ctx, ctx_null], freqs=freqs, y=y)
        return sched.step(cfg_combine(cond, uncond, 4.0 if trans is m1 else 3.0), t, lat)[0]
    lat = step(0, lat)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k = 2
    for i in range(1, 1 + k):
        lat = step(i, lat)
    torch.cuda.synchronize()
    step_s = (time.perf_counter() - t0) / k
    assert torch.isfinite(lat).all()
    td = time.perf_counter()
    video = vae.decode_to_cpu_uint8([lat[0]], 0)[0]
    torch.cuda.synchronize()
    dec_s = time.perf_counter() - td
    assert video.dtype == torch.uint8 and tuple(video.shape) == (3, (f - 1) * 4 + 1, h * 8, w * 8)
    L = f * (h // 2) * (w // 2)
    return {"workload": desc, "dtype": "fp8-e4m3 block Linears (weights + dynamically quantised activations), bf16 elsewhere",
            "metric": "denoise-steps/s", "value": 1.0 / step_s, "ms_per_step": step_s * 1e3, "steps": k, "warmup": 1,
            "step_TFLOPs": 2 * forward_flops(cfg, L) / step_s / 1e12, "vae_encode_s": enc_s, "vae_encode_first_call_s": enc_first_s,
            "vae_encode_note": "the encode of the 81-frame conditioning clip alone (inputs built before the clock starts), second call; "
                               "first_call includes the allocator's first touch of the encoder's activations after empty_cache()",
            "vae_decode_to_host_s": dec_s,
            "composed_s_at_%d_steps" % VIDEO_STEPS: VIDEO_STEPS * step_s + enc_s + dec_s}


def _link_delay(nbytes, link_GBs):
    """The link model's transfer time on the CURRENT stream: nbytes / (link_GBs GB/s), spent by one spinning lane (wan_debug_delay)."""
    from wan2gp_amd import lib as L_
    L_.check(L_.load().wan_debug_delay(nbytes / (link_GBs * 1e3), L_.stream_ptr()), "wan_debug_delay")     # bytes / (GB/s) -> microseconds


def simulate_world(worlds, model, model2, one_step, latents, new_sched, step_s_1gpu, cfg, L, par=None, layout="sp", fr=None, k=2, link_GBs=50.0, chunks=None):
    """ONE rank (rank 0) of a sequence-parallel world of N on this GPU: its token shard (L / N query rows against N gathered K / V^T
    segments, every token-local kernel at M = S L / N rows), the exchanges replaced by device-to-device copies of what the rank
    would receive, on a side stream like the RCCL path.  `rank_step_ms` / `compute_side_efficiency` are the COMPUTE side of the
    scaling curve (tile quantisation at L / N rows, GEMM and attention efficiency at the shard's shapes, the local / remote attention
    split).  The LINK MODEL (round 5, --simulate-link-GBs R > 0) puts the transfer time behind every copy: on the side stream, a delay
    of (bytes one peer sends this rank) / R -- xGMI is a full mesh of point-to-point links, every peer's share moves over its own link
    concurrently -- so an exchange that the schedule does not hide stalls the compute stream exactly as it would on the node;
    `rank_step_ms_link`, `link_modelled_efficiency`, `exposed_ms_per_block` = (t_link - t_compute) / layers.  Ulysses rows run the
    exchange in head chunks (the library default) AND as one exchange per tensor (`one_exchange`), both with the model on: what the
    chunking buys.  Efficiency = t(1) / (N t(N)).  The outputs of a simulated rank are not a forward's (the copies deliver the rank's
    own data); what IS checked: finite, and the chunked and one-exchange runs of a row are bit-identical (same inputs, same fake
    exchange -- the property tests/test_gpu_sp.py holds on real exchanges).  Parity of these layouts at this size:
    tests/test_gpu_baseline_configs.py::test_ulysses_world_rank_dryruns_at_baseline_size, tests/test_gpu_sp.py::test_ulysses_ranks_at_baseline_size_*."""
    import torch
    from wan2gp_amd.sp import SequenceParallel

    class SimulatedRank(SequenceParallel):
        def __init__(self, world, mode="allgather"):
            super().__init__(0, world, mode=mode)
            self.side = torch.cuda.Stream()
            self.bytes = 0
            self.link_GBs = 0.0          # 0: copies only

        def _transfer(self, copy, per_peer_bytes, key):
            """The stand-in of one exchange on the side stream, ordered behind the compute stream: the copy, then the link time."""
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.side):
                self.side.wait_event(ev)
                copy()
                if self.link_GBs > 0:
                    _link_delay(per_peer_bytes, self.link_GBs)
                done = torch.cuda.Event()
                done.record(self.side)
            self._pending[key] = done
            self.bytes += per_peer_bytes * (self.world - 1)

        def _a2a_begin_cb(self, user, which, send, recv, nbytes, stream):
            """Ulysses: what the all-to-all leaves in recv = `world` chunks of nbytes (here: this rank's own chunks, copied)."""
            try:
                base = self._ws.data_ptr()
                sv = self._ws[send - base:send - base + nbytes * self.world]
                rv = self._ws[recv - base:recv - base + nbytes * self.world]
                self._transfer(lambda: rv.copy_(sv), nbytes, ("a2a", which))
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1

        def _a2a_wait_cb(self, user, which, stream):
            ev = self._pending.pop(("a2a", which), None)
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)
            return 0

        def _gather_begin_cb(self, user, which, send, recv, nbytes, stream):
            try:
                base = self._ws.data_ptr()
                sv = self._ws[send - base:send - base + nbytes]
                rv = self._ws[recv - base:recv - base + nbytes * self.world].view(self.world, nbytes)
                self._transfer(lambda: rv.copy_(sv.unsqueeze(0).expand(self.world, nbytes)), nbytes, which)   # world x nbytes written: what the gather leaves in recv
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1

        def _gather_wait_cb(self, user, which, stream):
            ev = self._pending.pop(which, None)
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)
            return 0

        def all_gather(self, send):
            return send.repeat(self.world, *([1] * (send.dim() - 1)))

    class SimulatedCfgRank:
        """Rank 0 of a cfg2 x sp(N/2) world: the conditional stream alone through the model (S = 1, the half's simulated sequence-parallel
        group on it), the partner's prediction = a device-to-device copy of this rank's (the bytes the 2-rank swap delivers; with the
        link model on, the swap's transfer time on the compute stream -- nothing runs beside it)."""

        def __init__(self, sp):
            self.sp, self.stream, self.link_GBs = sp, 0, 0.0

        def guided_pair(self, model_, lat, context, context_null, **kw):
            r = model_(x=[lat], context=[context], x_id=0, **kw)[0]
            other = r.clone()
            if self.link_GBs > 0:
                _link_delay(r.numel() * r.element_size(), self.link_GBs)
            return r, other

    rows = []
    lays = {"both": ("sp", "cfg-sp"), "all": ("sp", "cfg-sp", "ulysses", "cfg-ulysses"), "cfg-both": ("cfg-sp", "cfg-ulysses")}.get(layout, (layout,))
    plans = [(n, lay) for n in worlds for lay in lays]
    layers = cfg["num_layers"]
    for n, lay in plans:
        cfg_half = lay.startswith("cfg-")                      # the two CFG streams on the two halves of the world
        uly = lay.endswith("ulysses")                          # the per-block exchange: four all-to-alls instead of two all-gathers
        deg = n // 2 if cfg_half else n                        # ranks sharing one stream's token axis
        name = ("cfg2 x sp%d" % deg if cfg_half else "sp%d" % n) + (" (ulysses)" if uly else "")
        if cfg_half and n % 2:
            rows.append({"world": n, "layout": lay, "skipped": f"world {n} is odd"})
            continue
        if L % deg:
            rows.append({"world": n, "layout": lay, "skipped": f"{L} tokens do not divide by {deg}"})
            continue
        if uly and cfg["num_heads"] % deg:
            rows.append({"world": n, "layout": lay, "skipped": f"{cfg['num_heads']} heads do not divide by {deg}"})
            continue
        if uly and deg == 1:
            continue                                           # cfg2 x sp1: no exchange inside a half -- the cfg-sp row already is this layout
        sp = SimulatedRank(deg, "ulysses" if uly else "allgather") if deg > 1 else None
        cfgr = SimulatedCfgRank(sp) if cfg_half else None
        try:
            if cfg_half and par is not None:
                par["cfgp"] = cfgr
            model.sp = sp
            if model2 is not None:
                model2.sp = sp
            one_step(0, latents, new_sched(), fr)                             # warm-up: workspace of this sharding (a scheduler of its own:
            torch.cuda.synchronize()                                          # never the timed region's, which may be at its last timestep)

            def timed(link, chunks):
                """k steps on a fresh scheduler from the SAME latents (a run's result can be compared with another's) -> (s / step, latents)"""
                if sp is not None:
                    sp.link_GBs, sp.chunks, sp.bytes = link, chunks, 0
                if cfgr is not None:
                    cfgr.link_GBs = link
                sck = new_sched()
                lat = latents
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(k):
                    lat = one_step(i, lat, sck, fr)
                torch.cuda.synchronize()
                dt_ = (time.perf_counter() - t0) / k
                assert torch.isfinite(lat).all()
                return dt_, lat

            if sp is not None:
                sp.chunks = chunks                                            # None: the library default (--sp-chunks overrides)
            C = sp.resolved_chunks(cfg.get("num_heads")) if sp is not None else 1
            dt, lat_c = timed(0.0, chunks)
            row = {"world": n, "layout": name,
                   "rank_step_ms": dt * 1e3, "compute_side_efficiency": step_s_1gpu / (n * dt),
                   # all-gather form: K + V^T of the other ranks of the group; Ulysses: the (deg - 1) / deg of q, k, v^T, o this rank
                   # sends away (= receives) -- per block, for the streams this rank runs
                   "gathered_bytes_per_block_and_rank": (sp.bytes / (k * layers)) if sp is not None else 0.0,
                   "exchange": ("all-to-all: q, k, v^T, o each in %d head chunks" % C if C > 1 else "all-to-all x 4 (q, k, v^T, o)") if uly
                               else ("all-gather x 2 (K, V^T)" if sp is not None else "none"),
                   "tokens_per_rank": L // deg, "streams_per_rank": 1 if cfg_half else 2}
            if link_GBs > 0:
                dl, _ = timed(link_GBs, chunks)
                row.update({"rank_step_ms_link": dl * 1e3, "link_modelled_efficiency": step_s_1gpu / (n * dl),
                            "exposed_ms_per_block": (dl - dt) * 1e3 / layers})
                if uly and C > 1:
                    d1, lat_1 = timed(0.0, 1)
                    d1l, _ = timed(link_GBs, 1)
                    # bit-identical while every launch is one launch; with the split tail (round 6) the q blocks of a launch's last partial
                    # round are summed as key-range parts, and WHICH blocks those are depends on the launch's size: a few fp32 additions
                    # in another order on those rows (tests/test_gpu_baseline_configs.py pins both statements)
                    same = bool(torch.equal(lat_1, lat_c))
                    dmax = float((lat_1.float() - lat_c.float()).abs().max())
                    scale = max(1.0, float(lat_c.float().abs().max()))
                    row["one_exchange"] = {"rank_step_ms": d1 * 1e3, "compute_side_efficiency": step_s_1gpu / (n * d1), "rank_step_ms_link": d1l * 1e3,
                                           "link_modelled_efficiency": step_s_1gpu / (n * d1l), "exposed_ms_per_block": (d1l - d1) * 1e3 / layers,
                                           "latents_bit_identical_to_chunked": same, "latents_max_abs_diff_to_chunked": dmax}
                    row["chunking_gain_points"] = 100.0 * (row["link_modelled_efficiency"] - row["one_exchange"]["link_modelled_efficiency"])
                    assert dmax <= 2.0 ** -6 * scale, "the chunked and the one-exchange Ulysses runs of the simulated rank differ by %g (scale %g)" % (dmax, scale)
            rows.append(row)
        finally:                                                              # whatever happened: the models leave as they came
            model.sp = None
            if model2 is not None:
                model2.sp = None
            if par is not None:
                par["cfgp"] = None
    return {"note": "one rank's shard on one GPU, all-gathers / all-to-alls = device-to-device copies on a side stream; rank_step_ms / "
                    "compute_side_efficiency: copies only (compute-side upper bound); *_link: every copy followed by (bytes from one peer) / "
                    "link rate on the side stream (the link model); timed steps per figure: %d" % k,
            "link_model_GBs_per_peer": link_GBs,
            "one_gpu_step_ms": step_s_1gpu * 1e3, "ranks": rows}


def s1_block(model, model2, latents, ctx, freqs, y, new_sched, e2e):
    """Guidance scale 1: ONE stream per forward, no unconditional pass, no combine (any2video.py:1626-1643 with guide_scale 1) -- the
    regime the reference's Wan2.2 profiles ship (profiles/wan_2_2/*.json: lightning LoRAs, 4 steps, guidance 1, the high-noise expert for
    the first half).  1 warm-up + 4 timed steps at S = 1 on the resident experts; a 4-step video composed with the VAE decode of this run."""
    import torch
    sc = new_sched(4)
    lat = latents

    def step(i, lat):
        t = sc.timesteps[i]
        trans = model2 if (model2 is not None and i >= 2) else model        # two steps per expert
        pred = trans([lat], t=torch.stack([t]), context=[ctx], freqs=freqs, y=y)[0]
        return sc.step(pred, t, lat)[0]
    step(0, lat)                                                            # warm-up: the S = 1 workspace
    sc = new_sched(4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(4):
        lat = step(i, lat)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert torch.isfinite(lat).all()
    dec = e2e["vae_decode_to_host_s"] if isinstance(e2e, dict) and "vae_decode_to_host_s" in e2e else None
    return {"what": "guidance 1: one stream per forward, 4 UniPC steps (two per expert), S = 1", "ms_per_forward": dt / 4 * 1e3, "steps": 4, "warmup": 1,
            "forwards_per_s": 4 / dt, "sampler_s_per_4_step_video": dt,
            "composed_s_per_4_step_video_with_vae_decode": (dt + dec) if dec is not None else None}


def e2e_full_block(model, model2, vae, freqs, y, fhw, new_sched, guide, switch_threshold):
    """One video, everything measured: prompt strings -> HuggingfaceTokenizer (the committed fixture tests/golden/tiny_tokenizer; its vocabulary
    sizes the embedding table) -> UMT5-XXL encoder at full width (24 blocks, d 4096, ffn 10240, 64 heads; random weights) for the positive and
    the negative prompt (any2video.py:587-593) -> noise -> VIDEO_STEPS guided UniPC steps on the resident experts -> causal 3D VAE decode ->
    uint8 frames on the host.  Wall clock from the first string to the last frame; the encoder's weights are built before the clock starts
    (a server holds them resident like the experts)."""
    import torch
    from wan2gp_amd.schedulers import cfg_combine
    from wan2gp_amd.t5 import T5EncoderModelHIP
    from wan2gp_amd.tokenizers import HuggingfaceTokenizer
    f, h, w = fhw
    tok = HuggingfaceTokenizer(os.path.join(ROOT, "tests", "golden", "tiny_tokenizer"), seq_len=512, clean="whitespace")
    g = torch.Generator(device="cuda").manual_seed(5)
    vocab = int(max(tok.vocab_size, len(tok.tokenizer)))
    kw = dict(vocab_size=vocab, dim=4096, dim_attn=4096, dim_ffn=10240, num_heads=64, num_layers=24, num_buckets=32)

    def rn(*shape, std=0.02):
        return (torch.randn(*shape, device="cuda", generator=g) * std).to(torch.bfloat16)
    sd = {"token_embedding.weight": rn(vocab, 4096, std=1.0), "norm.weight": torch.ones(4096, device="cuda", dtype=torch.bfloat16)}
    for i in range(24):
        b = "blocks.%d." % i
        sd.update({b + "norm1.weight": torch.ones(4096, device="cuda", dtype=torch.bfloat16), b + "norm2.weight": torch.ones(4096, device="cuda", dtype=torch.bfloat16),
                   b + "attn.q.weight": rn(4096, 4096), b + "attn.k.weight": rn(4096, 4096), b + "attn.v.weight": rn(4096, 4096), b + "attn.o.weight": rn(4096, 4096),
                   b + "ffn.gate.0.weight": rn(10240, 4096), b + "ffn.fc1.weight": rn(10240, 4096), b + "ffn.fc2.weight": rn(4096, 10240),
                   b + "pos_embedding.embedding.weight": rn(32, 64, std=0.5)})
    enc = T5EncoderModelHIP(512, tok, state_dict=sd)
    del sd
    prompt = "a red fox runs across a snowy field at dawn, cinematic, shallow depth of field"
    negative = "blurry, low quality, static, watermark"
    enc([prompt])                                                           # warm-up: relative-position tables, allocator
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctxs = enc([prompt, negative])
    pad = [torch.cat([u, u.new_zeros(512 - u.shape[0], u.shape[1])]).unsqueeze(0) for u in ctxs]        # model.py: context padded to text_len
    torch.cuda.synchronize()
    t_enc = time.perf_counter()
    sc = new_sched()
    lat = torch.randn(1, 16, f, h, w, device="cuda", generator=g)
    for i in range(VIDEO_STEPS):
        t = sc.timesteps[i]
        trans = model2 if (model2 is not None and int(t) <= switch_threshold) else model
        cond, uncond = trans([lat, lat], t=torch.stack([t]), context=[pad[0], pad[1]], freqs=freqs, y=y)
        lat = sc.step(cfg_combine(cond, uncond, guide if trans is model else 3.0), t, lat)[0]
    torch.cuda.synchronize()
    t_steps = time.perf_counter()
    video = vae.decode_to_cpu_uint8([lat[0]], 0)[0]
    torch.cuda.synchronize()
    t_end = time.perf_counter()
    assert video.dtype == torch.uint8 and tuple(video.shape) == (3, (f - 1) * 4 + 1, h * 8, w * 8) and not video.is_cuda
    return {"unit": "s/video", "measured_s": t_end - t0, "text_encode_s": t_enc - t0, "sampling_s": t_steps - t_enc, "sampling_steps": VIDEO_STEPS,
            "ms_per_step": (t_steps - t_enc) / VIDEO_STEPS * 1e3, "vae_decode_to_host_s": t_end - t_steps, "video": [3, (f - 1) * 4 + 1, h * 8, w * 8],
            "prompt_tokens": [int(u.shape[0]) for u in ctxs],
            "note": "measured, not composed: tokenizer (fixture vocabulary) + UMT5-XXL at full width with random weights, two prompts -> %d guided steps "
                    "-> VAE decode -> uint8 on the host" % VIDEO_STEPS}


def configs0_gpu(vae):
    """BASELINE configs[0] on the GPU -- Wan2.1 t2v 1.3B, 320 x 512 x 17 frames (L = 3,200), 10 UniPC steps, shift 5, guidance 5
    (defaults/t2v_1.3B.json): the SAME configuration the cpu_baseline leg times on the host cores, through WanAny2VHIP.generate():
    noise -> 10 CFG steps -> VAE decode -> uint8 frames on the host.  At this size a step is ~900 launches of microseconds each; the
    forwards are replayed launch lists (wan_dit_forward_graph, WanModelHIP.graph = "auto" engages below 16,384 tokens per joint pass)
    and the same video is timed once more with the eager forwards beside it; both produce the same frames (asserted)."""
    import torch
    from wan2gp_amd.model import WanModelHIP
    from wan2gp_amd.pipeline import WanAny2VHIP
    cfg, (f, h, w), desc = WORKLOADS["1.3B-320x512x17f"]
    m = random_weights(WanModelHIP(**cfg), cfg, 99)
    if vae is None:
        from wan2gp_amd.vae import WanVAEHIP, random_vae_state_dict
        vae = WanVAEHIP(state_dict=random_vae_state_dict())
    g = torch.Generator(device="cuda").manual_seed(7)
    ctx = (torch.randn(1, 512, 4096, device="cuda", generator=g) * 0.5).to(torch.bfloat16); ctx[:, 77:] = 0
    ctx_null = torch.zeros_like(ctx)
    pipe = WanAny2VHIP(m, vae=vae, device="cuda")
    kw = dict(context=ctx, context_null=ctx_null, width=w * 8, height=h * 8, frame_num=(f - 1) * 4 + 1, shift=5.0,
              sample_solver="unipc", guide_scale=5.0, seed=3, sampling_steps=10)
    res = {}
    frames = {}
    for mode in ("auto", "off"):
        m.graph = mode
        pipe.generate(**kw)                                                    # warm-up: workspace, VAE buffers, (auto) the captures
        torch.cuda.synchronize()
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            out = pipe.generate(**kw)
            torch.cuda.synchronize()
            dt_ = time.perf_counter() - t0
            best = dt_ if best is None else min(best, dt_)
        assert out["x"].dtype == torch.uint8 and tuple(out["x"].shape) == (3, (f - 1) * 4 + 1, h * 8, w * 8)
        lat = pipe.generate(return_latents=True, **kw)["latents"]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        vae.decode_to_cpu_uint8([lat[0] if lat.dim() == 5 else lat], 0)
        torch.cuda.synchronize()
        dec = time.perf_counter() - t0
        frames[mode] = out["x"]
        res[mode] = {"e2e_s": best, "vae_decode_to_host_s": dec, "step_s": (best - dec) / 10.0, "last_forward": int(m.last_graph_how)}
    L = f * (h // 2) * (w // 2)
    a, e = res["auto"], res["off"]
    return {"workload": desc + " -- BASELINE configs[0], 10 steps, shift 5, guidance 5", "e2e_s": a["e2e_s"], "step_s": a["step_s"],
            "vae_decode_to_host_s": a["vae_decode_to_host_s"], "forwards": "replayed launch lists (hipGraph)" if a["last_forward"] == 3 else
            "eager (capture refused: last_forward = %d)" % a["last_forward"],
            "eager_forwards": {"e2e_s": e["e2e_s"], "step_s": e["step_s"]}, "replay_over_eager_step": e["step_s"] / a["step_s"],
            "frames_identical_to_eager": bool(torch.equal(frames["auto"], frames["off"])),
            "step_TFLOPs": 2 * forward_flops(cfg, L) / a["step_s"] / 1e12,
            "note": "best of 3 full generate() calls (noise -> 10 CFG steps -> VAE decode -> uint8 on the host); step_s = (e2e - decode) / 10"}


def secondary_1p3b(vae):
    """BASELINE configs[1] on the same code: Wan2.1 t2v 1.3B 480x832x81f through WanAny2VHIP.generate() -- noise, 30 CFG
    UniPC steps, VAE decode, uint8 frames on the host -- timed as a whole and per step."""
    import torch
    from wan2gp_amd.model import WanModelHIP
    from wan2gp_amd.pipeline import WanAny2VHIP
    cfg, (f, h, w), desc = WORKLOADS["1.3B-480p"]
    m = random_weights(WanModelHIP(**cfg), cfg, 99)
    if vae is None:
        from wan2gp_amd.vae import WanVAEHIP, random_vae_state_dict
        vae = WanVAEHIP(state_dict=random_vae_state_dict())
    g = torch.Generator(device="cuda").manual_seed(7)
    ctx = (torch.randn(1, 512, 4096, device="cuda", generator=g) * 0.5).to(torch.bfloat16); ctx[:, 77:] = 0
    ctx_null = torch.zeros_like(ctx)
    pipe = WanAny2VHIP(m, vae=vae, device="cuda")
    stamps = []

    def cb(i, *a):
        if i >= 0:
            torch.cuda.synchronize()
            stamps.append(time.perf_counter())
    kw = dict(context=ctx, context_null=ctx_null, width=w * 8, height=h * 8, frame_num=(f - 1) * 4 + 1, shift=5.0,
              sample_solver="unipc", guide_scale=5.0, seed=3)
    pipe.generate(sampling_steps=2, **kw)                                   # warm-up: workspace, VAE buffers
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = pipe.generate(sampling_steps=VIDEO_STEPS, callback=cb, **kw)
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    assert out["x"].dtype == torch.uint8 and tuple(out["x"].shape) == (3, 81, h * 8, w * 8)
    step_ms = (stamps[-1] - stamps[4]) / (len(stamps) - 5) * 1e3
    L = f * (h // 2) * (w // 2)
    return {"workload": desc, "metric": "denoise-steps/s", "value": 1e3 / step_ms, "ms_per_step": step_ms,
            "sampling_steps": VIDEO_STEPS, "e2e_s_per_video": total, "vae_decode_to_host_s": total - (stamps[-1] - t0),
            "step_TFLOPs": 2 * forward_flops(cfg, L) / (step_ms * 1e-3) / 1e12,
            "note": "full generate(): noise -> 30 CFG steps -> VAE decode -> uint8 on host; step time = mean of steps 5..29"}


if __name__ == "__main__":
    main()
