"""Temporal sequence parallelism over RCCL / xGMI (new capability: the reference has no
multi-GPU code on the Wan path).

The token axis (f-major, so contiguous token ranges are temporal slabs) is split into
`world` equal contiguous shards, one per GPU/process.  Everything in a DiT block except
self-attention is token-local; per block each rank all-gathers K and V^T of the other shards
(one `all_gather_into_tensor` each -- all-gather uses all 7 xGMI links of a rank concurrently,
there is no all-reduce anywhere) and attends its local Q rows against the `world` gathered
segments (wan_attention_seg).  Weights are replicated, latents are replicated (19 MB), every
rank runs the identical scheduler arithmetic, so there is no broadcast either; the only other
exchange is the all-gather of the head's token-major output once per forward.

Host logic here is device-agnostic (`gloo` on CPU in tests, `nccl` == RCCL on GPUs).

Two ways to run the collectives: through `torch.distributed` (default: host callbacks `gather_begin` / `gather_wait`, torch owns
the communicator and its side stream), or `native=True`: the library's own RCCL communicator (`wan_sp_init`, csrc/sp_rccl.hip) --
the gather hooks are then C functions and the forward never re-enters Python; torch.distributed is only used once, to hand
rank 0's 128-byte communicator id to the other ranks.
"""
import ctypes
from ctypes import c_void_p

import torch
import torch.distributed as dist

from .lib import GATHER_FN, GATHER_WAIT_FN, SP_ALLGATHER, SP_ULYSSES, SpInfo


def shard_range(L: int, rank: int, world: int):
    """Equal contiguous token shards; L must divide evenly (75,600 / 147,600 / 32,760 all do for 2,4,8)."""
    if L % world != 0:
        raise ValueError(f"token count {L} is not divisible by {world} sequence shards")
    n = L // world
    return rank * n, n


class SequenceParallel:
    """mode "allgather" (default): every rank gathers the other shards' K / V^T per block and attends its own query rows.
    mode "ulysses": per block four all-to-alls re-shard q, k, v^T from "my tokens, all heads" to "all tokens, my heads" and o back
    (wan_dit_forward with WAN_SP_ULYSSES, csrc/dit.hip): one attention launch at full L for H / world heads; 4 (world - 1) / world
    shard-sized transfers per block and rank instead of 2 (world - 1); needs H % world == 0 (14B: 40 heads -> 2, 4, 8; 1.3B: 12
    heads -> 2, 4).  Everything outside self-attention is identical in both modes.
    `chunks` (ulysses): head chunks of the exchanges (wan_sp_info.a2a_chunks, csrc/dit.hip): q, k, v^T and o travel per chunk, chunk 0's
    first; chunk c's attention launch runs while chunk c + 1's tensors arrive and chunk c - 1's o returns; None = 2 when a rank holds >= 4 heads (14B at 2 / 4 / 8 ranks: 20 / 10 / 5
    heads -> launches of 10 / 5 / 2-3 heads still fill the chip: DESIGN.md section 6), else 1; results are bit-identical for every value."""

    def __init__(self, rank: int, world: int, group=None, native: bool = False, mode: str = "allgather", chunks=None):
        if mode not in ("allgather", "ulysses"):
            raise ValueError(f"SequenceParallel: mode {mode!r} is not 'allgather' or 'ulysses'")
        self.rank, self.world, self.group, self.mode = rank, world, group, mode
        self.chunks = chunks
        self.a2a_bytes = 0          # bytes this rank SENT to other ranks through the Ulysses exchanges (bench: per block and rank)
        self._ws = None
        self._cb = None
        self._cbw = None
        self._info = None
        self._pending = {}          # which -> async Work handle of the in-flight all-gather
        self._native = None         # wan_sp* (library-owned RCCL communicator)
        if native:
            self._init_native()

    def _init_native(self):
        from . import lib as _L
        lib = _L.load()
        dev = "cuda" if (not dist.is_initialized() or dist.get_backend(self.group) != "gloo") else "cpu"
        ident = torch.zeros(128, dtype=torch.uint8)
        if self.rank == 0:
            _L.check(lib.wan_sp_unique_id(c_void_p(ident.data_ptr())), "wan_sp_unique_id")
        if self.world > 1:
            t = ident.to(dev)
            # `src` is a GLOBAL rank: the group's first member, which need not be global rank 0 (CFG-parallel x SP layouts)
            dist.broadcast(t, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
            ident = t.cpu()
        h = c_void_p()
        _L.check(lib.wan_sp_init(ctypes.byref(h), self.rank, self.world, c_void_p(ident.data_ptr())), "wan_sp_init")
        self._native, self._lib = h, lib

    def __del__(self):
        h, self._native = getattr(self, "_native", None), None
        if h:
            self._lib.wan_sp_destroy(h)

    # ---- collectives (torch.distributed; backend nccl == RCCL on ROCm) ---------------------------
    def _all_gather_into(self, out: torch.Tensor, send: torch.Tensor):
        """out[world*n] <- concat over ranks of send[n] (flat views).  RCCL path: one
        all_gather_into_tensor on device memory.  `gloo` (CPU tests, and the single-GPU functional
        test where both ranks share one device) stages through host memory."""
        if dist.get_backend(self.group) == "gloo":
            src = send.detach().cpu().contiguous()
            parts = [torch.empty_like(src) for _ in range(self.world)]
            dist.all_gather(parts, src, group=self.group)
            out.copy_(torch.cat([p.reshape(-1) for p in parts]).view_as(out))
        else:
            dist.all_gather_into_tensor(out, send.contiguous(), group=self.group)

    def all_gather(self, send: torch.Tensor) -> torch.Tensor:
        """[n, ...] per rank -> [world*n, ...] in rank order."""
        out = torch.empty((self.world * send.shape[0],) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        if self._native is not None:
            from . import lib as _L
            send = send.contiguous()
            _L.check(self._lib.wan_sp_all_gather(self._native, c_void_p(send.data_ptr()), c_void_p(out.data_ptr()),
                                                 send.numel() * send.element_size(), _L.stream_ptr()), "wan_sp_all_gather")
            return out
        self._all_gather_into(out, send)
        return out

    def bind_workspace(self, ws: torch.Tensor):
        """The C++ forward hands raw pointers into the torch-owned workspace back to the gather
        callback; they are mapped to views of `ws` so torch.distributed orders the collective
        against the compute stream."""
        self._ws = ws

    def _gather_begin_cb(self, user, which, send, recv, nbytes, stream):
        """Start the all-gather of a workspace region.  RCCL: async on the communicator's stream, ordered behind
        the work already enqueued on the compute stream; the compute stream keeps going (overlap)."""
        try:
            base = self._ws.data_ptr()
            s_off, r_off = send - base, recv - base
            sv = self._ws[s_off:s_off + nbytes]
            rv = self._ws[r_off:r_off + nbytes * self.world]
            if dist.get_backend(self.group) == "gloo":
                self._all_gather_into(rv, sv)          # host-staged, synchronous
                self._pending[which] = None
            else:
                self._pending[which] = dist.all_gather_into_tensor(rv, sv, group=self.group, async_op=True)
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return 1

    def _gather_wait_cb(self, user, which, stream):
        try:
            w = self._pending.pop(which, None)
            if w is not None:
                w.wait()                               # current stream waits for the collective's completion event
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return 1

    def _a2a_begin_cb(self, user, which, send, recv, nbytes, stream):
        """Start an all-to-all of workspace regions: chunk j of send[world][nbytes] -> rank j, chunk i of recv <- rank i.  RCCL: async
        on the communicator's stream behind the work already enqueued on the compute stream."""
        try:
            base = self._ws.data_ptr()
            s_off, r_off = send - base, recv - base
            sv = self._ws[s_off:s_off + nbytes * self.world]
            rv = self._ws[r_off:r_off + nbytes * self.world]
            if dist.get_backend(self.group) == "gloo":       # host-staged, synchronous (CPU tests; all ranks of a test on one GPU)
                src = sv.detach().cpu().contiguous()
                dst = torch.empty_like(src)
                dist.all_to_all_single(dst, src, group=self.group)
                rv.copy_(dst)
                self._pending[("a2a", which)] = None
            else:
                self._pending[("a2a", which)] = dist.all_to_all_single(rv, sv, group=self.group, async_op=True)
            self.a2a_bytes += nbytes * (self.world - 1)
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return 1

    def _a2a_wait_cb(self, user, which, stream):
        try:
            w = self._pending.pop(("a2a", which), None)
            if w is not None:
                w.wait()
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return 1

    def resolved_chunks(self, heads=None) -> int:
        """Head chunks a Ulysses block of a model with `heads` heads runs with (the library clamps to [1, min(heads / world, 8)])."""
        if self.mode != "ulysses":
            return 1
        hn = (heads // self.world) if heads else None
        c = self.chunks if self.chunks is not None else (2 if (hn is None or hn >= 4) else 1)
        c = max(1, min(int(c), 8))
        return c if hn is None else max(1, min(c, hn))

    def make_info(self, L: int, heads=None) -> SpInfo:
        tok0, n = shard_range(L, self.rank, self.world)
        mode = SP_ULYSSES if self.mode == "ulysses" else SP_ALLGATHER
        chunks = self.resolved_chunks(heads)
        if self._native is not None:                          # the library's own hooks: no Python in the block loop
            if self._cb is None:
                self._cb = ctypes.cast(self._lib.wan_sp_gather_begin, GATHER_FN)
                self._cbw = ctypes.cast(self._lib.wan_sp_gather_wait, GATHER_WAIT_FN)
                self._cba = ctypes.cast(self._lib.wan_sp_a2a_begin, GATHER_FN)
            self._info = SpInfo(self.rank, self.world, tok0, n, self._cb, self._cbw, self._native, mode, self._cba, self._cbw, chunks)
            return self._info
        if self._cb is None:
            self._cb = GATHER_FN(self._gather_begin_cb)      # keep the ctypes thunks alive
            self._cbw = GATHER_WAIT_FN(self._gather_wait_cb)
            self._cba = GATHER_FN(self._a2a_begin_cb)
            self._cbaw = GATHER_WAIT_FN(self._a2a_wait_cb)
        self._info = SpInfo(self.rank, self.world, tok0, n, self._cb, self._cbw, None, mode, self._cba, self._cbaw, chunks)
        return self._info

    def gather_output(self, tok_major: torch.Tensor, grid):
        """[1, L/world, 4*out_dim] fp32 per rank -> full [1,out_dim,F,H,W] on every rank."""
        from . import ops
        full = self.all_gather(tok_major[0]).unsqueeze(0)
        return ops.unpatchify(full.contiguous(), grid)


class CfgParallel:
    """The two streams of a classifier-free-guidance step on two halves of the world, sequence parallelism inside each half
    (new capability like the class above; the reference runs both streams on one device, any2video.py:1626-1643).

    A guided step is two independent forwards -- conditional and unconditional -- that meet only in the combine
    `uncond + g (cond - uncond)` (any2video.py:1722).  With `world` = 2 k ranks, ranks [0, k) run the conditional stream and ranks
    [k, 2k) the unconditional one, each half sharding its stream's token axis k ways (`SequenceParallel` over the half's subgroup;
    k = 1: no sequence parallelism at all).  Per STEP, rank i and rank i + k swap their halves' finished noise predictions (one
    2-rank all-gather of a latent-sized fp32 tensor, 19 MB at 720p x 81 frames); every rank then holds (cond, uncond) bit-identically
    and runs the same combine and scheduler arithmetic -- latents stay replicated, nothing is broadcast.

    Why on xGMI (point-to-point links, 7 per GPU): against pure sequence parallelism over all 2 k ranks the per-block K / V^T
    gathers carry ONE stream over k ranks instead of two over 2 k -- at 8 GPUs 1.16 GB per rank and block instead of 2.71 GB, the
    same bytes per link (3 peers instead of 7) but against a local attention segment of 1/4 instead of 1/8 of the block's attention to
    hide under, and a shard of L / 4 instead of L / 8 query rows (tile quantisation); at 2 GPUs there is no per-block exchange at all.

    `new_group` is collective over ALL ranks and must run in the same order everywhere: every rank creates every group here."""

    def __init__(self, rank: int, world: int, native: bool = False, mode: str = "allgather", chunks=None):
        if world < 2 or world % 2:
            raise ValueError(f"CFG parallelism splits the world in two halves: world size {world} is not a positive even number")
        self.rank, self.world, self.half = rank, world, world // 2
        self.stream = rank // self.half                      # 0: conditional stream, 1: unconditional stream
        self.sp_rank = rank % self.half
        halves = [dist.new_group(list(range(s * self.half, (s + 1) * self.half))) if self.half > 1 else None for s in (0, 1)]
        pairs = [dist.new_group([i, i + self.half]) for i in range(self.half)]
        self.pair = pairs[self.sp_rank]                      # {i, i + k}: group rank 0 = conditional, 1 = unconditional
        self.sp = SequenceParallel(self.sp_rank, self.half, group=halves[self.stream], native=native, mode=mode, chunks=chunks) if self.half > 1 else None

    def attach(self, *models):
        """The half's sequence-parallel group on every resident expert (None for a world of 2)."""
        for m in models:
            if m is not None:
                m.sp = self.sp
                m.cfg_parallel_stream = self.stream          # (a step-skipping cache on stream 1 makes stream 0's decision itself: model.py forward)
        return self

    def exchange(self, mine: torch.Tensor):
        """This rank's finished prediction -> (cond, uncond), identical on both ranks of the pair."""
        mine = mine.contiguous()
        if dist.get_backend(self.pair) == "gloo":
            parts = [torch.empty_like(mine, device="cpu") for _ in range(2)]
            dist.all_gather(parts, mine.detach().cpu(), group=self.pair)
            return parts[0].to(mine.device), parts[1].to(mine.device)
        out = torch.empty((2,) + tuple(mine.shape), dtype=mine.dtype, device=mine.device)
        dist.all_gather_into_tensor(out, mine, group=self.pair)
        return out[0], out[1]

    def guided_pair(self, model, lat, context, context_null, **kwargs):
        """The joint pass of any2video.py:1626-1634 across the two halves: this rank's stream through `model`, then the swap.
        `x_id` is the stream's identity as in the reference's single passes (:1638-1643).  Returns (cond, uncond), or None when the
        forward was interrupted -- an interrupt has to reach every rank (the host application sets `_interrupt` on all of them), a
        rank that alone leaves the step leaves its partner waiting in the swap."""
        r = model(x=[lat], context=[context if self.stream == 0 else context_null], x_id=self.stream, **kwargs)[0]
        if r is None:
            return None
        return self.exchange(r)
