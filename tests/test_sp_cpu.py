"""CPU, world_size 2 and 4, gloo: the host logic of temporal sequence parallelism (wan2gp_amd/sp.py) --
token sharding, gather ordering, RoPE position offsets, segmented K/V attention, token-major output
gather + unpatchify -- driven on the CPU oracle's arithmetic.  Every rank must reproduce the
single-process oracle block/forward result for its shard."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import wan_oracle as O


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from wan2gp_amd.sp import SequenceParallel, shard_range
        torch.set_num_threads(2)
        sp = SequenceParallel(rank, world)
        cfg = O.make_config("tiny")
        W = O.synth_weights(cfg, dtype=torch.float32)
        f, h, w = 2, 8, 8
        lat, ctx, _, _ = O.synth_inputs(cfg, f, h, w)
        grid = (f, h // 2, w // 2)
        L = f * (h // 2) * (w // 2)
        tok0, n = shard_range(L, rank, world)
        assert (tok0, n) == (rank * L // world, L // world)
        cos, sin = O.rope_tables(grid)
        t = torch.tensor([500])
        dt = torch.float32
        # reference: single-process oracle
        full_h, _ = O.patch_embed(lat, W, cfg, dt)
        e, e0 = O.time_embed(t, W, cfg, dt)
        cemb = O.text_embed(ctx.float(), W)
        ref = full_h
        for i in range(cfg.num_layers):
            ref = O.block_forward(ref, e0, cemb, cos, sin, W, i, cfg, exact=True)
        ref_out = O.unpatchify(O.head_forward(ref, e, W, cfg), grid, cfg)
        # sharded: everything token-local except self-attention, which sees gathered K / V segments
        x = full_h[:, tok0:tok0 + n]
        for i in range(cfg.num_layers):
            p = f"blocks.{i}."
            ee = (W[p + "modulation"] + e0).chunk(6, dim=1)
            xm = O.layer_norm(x, cfg.eps) * (1 + ee[1]) + ee[0]
            sa = p + "self_attn."
            qq = O.rms_norm(O._linear(xm, W, sa + "q"), W[sa + "norm_q.weight"], cfg.eps).view(1, n, cfg.num_heads, 128)
            kk = O.rms_norm(O._linear(xm, W, sa + "k"), W[sa + "norm_k.weight"], cfg.eps).view(1, n, cfg.num_heads, 128)
            vv = O._linear(xm, W, sa + "v").view(1, n, cfg.num_heads, 128)
            qq = O.rope_apply(qq, cos[tok0:tok0 + n], sin[tok0:tok0 + n])       # pos0 = tok0
            kk = O.rope_apply(kk, cos[tok0:tok0 + n], sin[tok0:tok0 + n])
            kf = sp.all_gather(kk[0]).unsqueeze(0)                               # [world*n, H, 128] in rank order
            vf = sp.all_gather(vv[0]).unsqueeze(0)
            y = O._linear(O.attention(qq, kf, vf, exact=True).flatten(2), W, sa + "o")
            x = torch.addcmul(x, y, ee[2])
            y = O.layer_norm(x, cfg.eps, W[p + "norm3.weight"], W[p + "norm3.bias"])
            x = x + O.cross_attention(y, cemb, W, p + "cross_attn.", cfg, True)
            y = O.layer_norm(x, cfg.eps) * (1 + ee[4]) + ee[3]
            y = O._linear(torch.nn.functional.gelu(O._linear(y, W, p + "ffn.0"), approximate="tanh"), W, p + "ffn.2")
            x = torch.addcmul(x, y, ee[5])
        assert torch.allclose(x, ref[:, tok0:tok0 + n], atol=1e-4, rtol=1e-4)
        tok_major = O.head_forward(x, e, W, cfg)                                 # [1, n, 64]
        full_tok = sp.all_gather(tok_major[0]).unsqueeze(0)                      # [1, L, 64]
        out = O.unpatchify(full_tok, grid, cfg)
        assert torch.allclose(out, ref_out, atol=1e-4, rtol=1e-4)
        q.put((rank, "ok"))
    except Exception as ex:  # noqa
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sequence_parallel_host_logic(world):
    """world 2 (the driver's smallest multi-GPU point) and 4 (L = 32 tokens -> 8 per rank): rank-ordered gathers, RoPE
    offsets and the token-major output gather do not depend on the shard count."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def test_shard_range_rejects_ragged():
    from wan2gp_amd.sp import shard_range
    assert shard_range(75600, 3, 8) == (3 * 9450, 9450)
    assert shard_range(147600, 7, 8) == (7 * 18450, 18450)
    with pytest.raises(ValueError):
        shard_range(75601, 0, 8)


# ---- the gather callbacks the C++ forward calls back into (wan_sp_info.gather_begin / gather_wait) ------------------------------
def _cb_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from wan2gp_amd.sp import SequenceParallel
        sp = SequenceParallel(rank, world)
        # a stand-in for the forward workspace: [K send | V^T send | K recv (world slots) | V^T recv (world slots)]
        nk, nv = 96, 160
        ws = torch.zeros(nk + nv + world * (nk + nv), dtype=torch.uint8)
        ws[:nk] = torch.arange(nk, dtype=torch.uint8) + rank * 3
        ws[nk:nk + nv] = torch.arange(nv, dtype=torch.uint8) + 100 + rank * 5
        sp.bind_workspace(ws)
        info = sp.make_info(L=64 * world)
        assert (info.rank, info.world, info.tok0, info.tok_local) == (rank, world, 64 * rank, 64)
        base = ws.data_ptr()
        # the order wan_dit_forward uses: K gather, V^T gather, (compute), wait K, wait V^T
        assert sp._gather_begin_cb(None, 0, base, base + nk + nv, nk, None) == 0
        assert sp._gather_begin_cb(None, 1, base + nk, base + nk + nv + world * nk, nv, None) == 0
        assert sp._gather_wait_cb(None, 0, None) == 0 and sp._gather_wait_cb(None, 1, None) == 0
        kr = ws[nk + nv: nk + nv + world * nk].view(world, nk)
        vr = ws[nk + nv + world * nk:].view(world, nv)
        for r in range(world):
            assert torch.equal(kr[r], torch.arange(nk, dtype=torch.uint8) + r * 3)
            assert torch.equal(vr[r], torch.arange(nv, dtype=torch.uint8) + 100 + r * 5)
        assert sp._gather_wait_cb(None, 0, None) == 0                     # a second wait is a no-op
        q.put((rank, "ok"))
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_gather_callbacks_fill_every_ranks_slot_in_rank_order(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cb_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


# ---- Ulysses (WAN_SP_ULYSSES): the four all-to-alls of a block with the layouts of csrc/dit.hip, on oracle arithmetic ---------------
def _ulysses_worker(rank, world, port, q, cfg_name):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from wan2gp_amd.lib import SP_ULYSSES
        from wan2gp_amd.sp import SequenceParallel, shard_range
        torch.set_num_threads(2)
        sp = SequenceParallel(rank, world, mode="ulysses")
        cfg = O.make_config(cfg_name)
        W = O.synth_weights(cfg, dtype=torch.float32)
        f, h, w = 2, 8, 8
        S = 2
        lats = [O.synth_inputs(cfg, f, h, w, seed=42 + s)[0] for s in range(S)]
        ctx = O.synth_inputs(cfg, f, h, w)[1]
        grid = (f, h // 2, w // 2)
        L = f * (h // 2) * (w // 2)
        tok0, Ll = shard_range(L, rank, world)
        H, d = cfg.num_heads, cfg.dim
        assert H % world == 0
        Hn, Wd, Lp, rows = H // world, H // world * 128, (Ll + 63) // 64 * 64, S * Ll
        cos, sin = O.rope_tables(grid)
        t = torch.tensor([500])
        dt = torch.float32
        e, e0 = O.time_embed(t, W, cfg, dt)
        cemb = O.text_embed(ctx.float(), W)
        info = sp.make_info(L)
        assert (info.mode, info.rank, info.world, info.tok0, info.tok_local) == (SP_ULYSSES, rank, world, tok0, Ll)
        # the forward's exchange buffers as ONE flat fp32 "workspace" (the callbacks address it by byte offsets)
        blkq, blkv = rows * d, S * d * Lp
        ws = torch.zeros(4 * blkq + 2 * blkv, dtype=torch.float32)
        sp.bind_workspace(ws.view(torch.uint8))
        base, el = ws.data_ptr(), 4
        ks, kr, qs, qr = (ws[i * blkq:(i + 1) * blkq] for i in range(4))
        vs, vr = ws[4 * blkq:4 * blkq + blkv], ws[4 * blkq + blkv:]

        def permute16(src, A, B, n):                              # wan_permute16: [A][B][n] -> [B][A][n]
            return src.reshape(A, B, n).transpose(0, 1).contiguous().reshape(-1)

        def a2a(which, send, recv, n_per_peer):
            assert sp._a2a_begin_cb(None, which, send.data_ptr(), recv.data_ptr(), n_per_peer * el, None) == 0

        refs, xs = [], []
        for s in range(S):
            full_h, _ = O.patch_embed(lats[s], W, cfg, dt)
            refs.append(O.block_forward(full_h, e0, cemb, cos, sin, W, 0, cfg, exact=True))
            xs.append(full_h[:, tok0:tok0 + Ll])
        p = "blocks.0."
        sa = p + "self_attn."
        ee = (W[p + "modulation"] + e0).chunk(6, dim=1)
        xm = torch.cat([O.layer_norm(x, cfg.eps) * (1 + ee[1]) + ee[0] for x in xs], dim=1)[0]          # [S Ll, d]: the streams stacked as rows
        pos = slice(tok0, tok0 + Ll)

        def normed(name, nw):
            y = O.rms_norm(O._linear(xm, W, sa + name), W[sa + nw], cfg.eps).view(S, Ll, H, 128)
            return O.rope_apply(y, cos[pos], sin[pos]).reshape(rows, d)
        # k: projection, norm + RoPE, re-pack head-group-major, exchange
        ks.copy_(permute16(normed("k", "norm_k.weight"), rows, world, Wd)); a2a(0, ks, kr, rows * Wd)
        # v^T: the transposed epilogue's [S][d][Lp] image (zero padded), blocks swapped [S][world] -> [world][S], exchange
        vt = torch.zeros(S, d, Lp)
        vt[:, :, :Ll] = O._linear(xm, W, sa + "v").view(S, Ll, d).transpose(1, 2)
        vs.copy_(permute16(vt.reshape(-1), S, world, Wd * Lp)); a2a(1, vs, vr, S * Wd * Lp)
        qs.copy_(permute16(normed("q", "norm_q.weight"), rows, world, Wd)); a2a(2, qs, qr, rows * Wd)
        for wch in range(3):
            assert sp._a2a_wait_cb(None, wch, None) == 0
        # ONE attention over world x S query batches of Ll rows, S K / V^T batches in `world` segments, Hn heads (q batch b -> b mod S)
        Q = qr.view(world, S, Ll, Hn, 128)
        K = kr.view(world, S, Ll, Hn, 128)
        V = vr.view(world, S, Hn, 128, Lp)
        out = torch.empty(world, S, Ll, Hn, 128)
        for s in range(S):
            kf = K[:, s].reshape(1, world * Ll, Hn, 128)                                               # the segments in rank order = token order
            vf = V[:, s, :, :, :Ll].permute(0, 3, 1, 2).reshape(1, world * Ll, Hn, 128)
            for i in range(world):
                out[i, s] = O.attention(Q[i, s].unsqueeze(0), kf, vf, exact=True)[0]
        # o: written in the send layout of the way back (over the dead k send buffer), received over the dead q send buffer
        ks.copy_(out.reshape(-1)); a2a(3, ks, qs, rows * Wd)
        assert sp._a2a_wait_cb(None, 3, None) == 0
        o_rows = permute16(qs, world, rows, Wd).view(S, Ll, d)
        assert sp.a2a_bytes == (3 * rows * Wd + S * Wd * Lp) * el * (world - 1)
        for s in range(S):
            x = torch.addcmul(xs[s], O._linear(o_rows[s:s + 1], W, sa + "o"), ee[2])
            y = O.layer_norm(x, cfg.eps, W[p + "norm3.weight"], W[p + "norm3.bias"])
            x = x + O.cross_attention(y, cemb, W, p + "cross_attn.", cfg, True)
            y = O.layer_norm(x, cfg.eps) * (1 + ee[4]) + ee[3]
            y = O._linear(torch.nn.functional.gelu(O._linear(y, W, p + "ffn.0"), approximate="tanh"), W, p + "ffn.2")
            x = torch.addcmul(x, y, ee[5])
            assert torch.allclose(x, refs[s][:, tok0:tok0 + Ll], atol=1e-4, rtol=1e-4), (s, (x - refs[s][:, tok0:tok0 + Ll]).abs().max().item())
        q.put((rank, "ok"))
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,cfg_name", [(2, "tiny"), (4, "small")])
def test_ulysses_exchanges_with_the_forwards_layouts(world, cfg_name):
    """The four all-to-alls of a WAN_SP_ULYSSES block through sp.py's callbacks, with the buffer layouts csrc/dit.hip uses (re-pack
    [rows][world][W] -> [world][rows][W]; v^T blocks [S][world] -> [world][S]; received = world x S query batches and `world` K / V^T
    segments; o back over the dead send buffers): every rank's block output equals the single-process oracle block on its token
    shard, for both CFG streams.  world 2: 2 heads -> 1 per rank; world 4: 4 heads, 8 tokens per rank (one padded V^T tile)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ulysses_worker, args=(r, world, port, q, cfg_name)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


# ---- the chunked form (round 5, wan_sp_info.a2a_chunks): csrc/dit.hip's per-chunk layouts and offsets restated on oracle arithmetic ---
def _ulysses_chunked_worker(rank, world, port, q, heads, chunks):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from wan2gp_amd.sp import SequenceParallel, shard_range
        torch.set_num_threads(2)
        sp = SequenceParallel(rank, world, mode="ulysses", chunks=chunks)
        cfg = O.WanConfig(dim=128 * heads, ffn_dim=256, num_heads=heads, num_layers=1)
        W = O.synth_weights(cfg, dtype=torch.float32)
        f, h, w = 2, 8, 8
        S = 2
        lats = [O.synth_inputs(cfg, f, h, w, seed=42 + s)[0] for s in range(S)]
        ctx = O.synth_inputs(cfg, f, h, w)[1]
        grid = (f, h // 2, w // 2)
        L = f * (h // 2) * (w // 2)
        tok0, Ll = shard_range(L, rank, world)
        H, d = cfg.num_heads, cfg.dim
        Hn, Wd, Lp, rows = H // world, H // world * 128, (Ll + 63) // 64 * 64, S * Ll
        info = sp.make_info(L, heads=H)
        C = info.a2a_chunks
        assert C == min(chunks, Hn) and sp.resolved_chunks(H) == C
        h0 = [c * Hn // C for c in range(C + 1)]                                # the library's chunk boundaries (dit.hip)
        cos, sin = O.rope_tables(grid)
        t = torch.tensor([500])
        dt = torch.float32
        e, e0 = O.time_embed(t, W, cfg, dt)
        cemb = O.text_embed(ctx.float(), W)
        blkq, blkv = rows * d, S * d * Lp
        ws = torch.zeros(4 * blkq + 2 * blkv, dtype=torch.float32)
        sp.bind_workspace(ws.view(torch.uint8))
        el = 4
        KS, KR, QS, QR = 0, blkq, 2 * blkq, 3 * blkq                            # element offsets of ks, kr, qs, qr, vs, vr in the workspace
        VS, VR = 4 * blkq, 4 * blkq + blkv

        def permute_ex(src, src_off, dst_off, A, B, n, sa, sb, da, db):        # wan_permute16_ex, pitches in elements
            sv = torch.as_strided(src, (A, B, n), (sa, sb, 1), src_off)
            torch.as_strided(ws, (A, B, n), (da, db, 1), dst_off).copy_(sv)

        def a2a(which, send_off, recv_off, n_per_peer):
            base = ws.data_ptr()
            assert sp._a2a_begin_cb(None, which, base + send_off * el, base + recv_off * el, n_per_peer * el, None) == 0

        refs, xs = [], []
        for s in range(S):
            full_h, _ = O.patch_embed(lats[s], W, cfg, dt)
            refs.append(O.block_forward(full_h, e0, cemb, cos, sin, W, 0, cfg, exact=True))
            xs.append(full_h[:, tok0:tok0 + Ll])
        p = "blocks.0."
        sa = p + "self_attn."
        ee = (W[p + "modulation"] + e0).chunk(6, dim=1)
        xm = torch.cat([O.layer_norm(x, cfg.eps) * (1 + ee[1]) + ee[0] for x in xs], dim=1)[0]
        pos = slice(tok0, tok0 + Ll)

        def normed(name, nw):
            y = O.rms_norm(O._linear(xm, W, sa + name), W[sa + nw], cfg.eps).view(S, Ll, H, 128)
            return O.rope_apply(y, cos[pos], sin[pos]).reshape(-1).contiguous()
        geo = [((h0[c + 1] - h0[c]) * 128, h0[c] * 128) for c in range(C)]        # (Wc, o0) per chunk
        kk = normed("k", "norm_k.weight")
        for Wc, o0 in geo:                                                      # [rows][world][Hn 128] -> [chunk][world][rows][Wc]
            permute_ex(kk, o0, KS + o0 * rows * world, rows, world, Wc, d, Wd, Wc, rows * Wc)
        send_k = lambda c: a2a(c, KS + geo[c][1] * rows * world, KR + geo[c][1] * rows * world, rows * geo[c][0])
        send_k(0)                                                               # chunk 0 first: under the V projection
        vt = torch.zeros(S, d, Lp)
        vt[:, :, :Ll] = O._linear(xm, W, sa + "v").view(S, Ll, d).transpose(1, 2)
        vt = vt.reshape(-1).contiguous()
        for Wc, o0 in geo:                                                      # [S][world][Hn 128][Lp] -> [chunk][world][S][Wc][Lp]
            permute_ex(vt, o0 * Lp, VS + o0 * Lp * S * world, S, world, Wc * Lp, d * Lp, Wd * Lp, Wc * Lp, S * Wc * Lp)
        send_v = lambda c: a2a(C + c, VS + geo[c][1] * Lp * S * world, VR + geo[c][1] * Lp * S * world, S * geo[c][0] * Lp)
        send_v(0)
        qq = normed("q", "norm_q.weight")
        for Wc, o0 in geo:                                                      # [rows][world][Hn 128] -> [chunk][world][rows][Wc]
            permute_ex(qq, o0, QS + o0 * rows * world, rows, world, Wc, d, Wd, Wc, rows * Wc)
        send_q = lambda c: a2a(2 * C + c, QS + geo[c][1] * rows * world, QR + geo[c][1] * rows * world, rows * geo[c][0])
        send_q(0)
        for c in range(1, C):                                                   # the later chunks, in the order their launches need them
            send_k(c); send_v(c); send_q(c)
        for c in range(C):
            Hc = h0[c + 1] - h0[c]
            Wc, o0 = geo[c]
            for which in (c, C + c, 2 * C + c):
                assert sp._a2a_wait_cb(None, which, None) == 0
            # the launch of chunk c: the round-4 layout with Hc heads, the CHUNK's segment strides
            Q = ws[QR + o0 * rows * world:QR + (o0 + Wc) * rows * world].view(world, S, Ll, Hc, 128)
            K = ws[KR + o0 * rows * world:KR + (o0 + Wc) * rows * world].view(world, S, Ll, Hc, 128)
            V = ws[VR + o0 * Lp * S * world:VR + (o0 + Wc) * Lp * S * world].view(world, S, Hc, 128, Lp)
            out = torch.empty(world, S, Ll, Hc, 128)
            for s in range(S):
                kf = K[:, s].reshape(1, world * Ll, Hc, 128)
                vf = V[:, s, :, :, :Ll].permute(0, 3, 1, 2).reshape(1, world * Ll, Hc, 128)
                for i in range(world):
                    out[i, s] = O.attention(Q[i, s].unsqueeze(0), kf, vf, exact=True)[0]
            ws[KS + o0 * rows * world:KS + (o0 + Wc) * rows * world] = out.reshape(-1)        # o over the dead k send chunk ...
            a2a(3 * C + c, KS + o0 * rows * world, QS + o0 * rows * world, rows * Wc)         # ... received over the dead q send chunk
        o_rows = torch.zeros(rows * d)
        for c in range(C):                                                      # [chunk][world][rows][Wc] -> [rows][world][Hn 128]
            Wc, o0 = (h0[c + 1] - h0[c]) * 128, h0[c] * 128
            assert sp._a2a_wait_cb(None, 3 * C + c, None) == 0
            sv = torch.as_strided(ws, (world, rows, Wc), (rows * Wc, Wc, 1), QS + o0 * rows * world)
            torch.as_strided(o_rows, (world, rows, Wc), (Wd, d, 1), o0).copy_(sv)
        assert sp.a2a_bytes == (3 * rows * Wd + S * Wd * Lp) * el * (world - 1)      # the same bytes as the unchunked form
        o_rows = o_rows.view(S, Ll, d)
        for s in range(S):
            x = torch.addcmul(xs[s], O._linear(o_rows[s:s + 1], W, sa + "o"), ee[2])
            y = O.layer_norm(x, cfg.eps, W[p + "norm3.weight"], W[p + "norm3.bias"])
            x = x + O.cross_attention(y, cemb, W, p + "cross_attn.", cfg, True)
            y = O.layer_norm(x, cfg.eps) * (1 + ee[4]) + ee[3]
            y = O._linear(torch.nn.functional.gelu(O._linear(y, W, p + "ffn.0"), approximate="tanh"), W, p + "ffn.2")
            x = torch.addcmul(x, y, ee[5])
            assert torch.allclose(x, refs[s][:, tok0:tok0 + Ll], atol=1e-4, rtol=1e-4), (s, (x - refs[s][:, tok0:tok0 + Ll]).abs().max().item())
        q.put((rank, "ok"))
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,heads,chunks", [(2, 4, 2), (2, 6, 2), (4, 12, 3), (2, 10, 5)],
                         ids=["w2_heads_1+1", "w2_heads_1+2", "w4_heads_1+1+1", "w2_heads_5x1"])
def test_ulysses_chunked_exchanges_with_the_forwards_layouts(world, heads, chunks):
    """The chunked Ulysses block (round 5): k / v^T / q / o packed chunk-major and exchanged per head chunk (chunk 0's first), every chunk's
    attention on the round-4 layout at the chunk's offsets and segment strides, the per-chunk un-pack -- with exactly the offsets and pitches csrc/dit.hip
    passes to wan_permute16_ex / wan_attention_bounded / a2a_begin, through sp.py's callbacks over gloo.  Every rank's block output
    equals the single-process oracle block on its token shard for both CFG streams; unequal chunks (3 heads as 1 + 2) included."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ulysses_chunked_worker, args=(r, world, port, q, heads, chunks)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def test_sequence_parallel_mode_is_validated():
    from wan2gp_amd.sp import SequenceParallel
    with pytest.raises(ValueError):
        SequenceParallel(0, 2, mode="ring")
    assert SequenceParallel(0, 2, mode="ulysses").mode == "ulysses" and SequenceParallel(0, 2).mode == "allgather"
