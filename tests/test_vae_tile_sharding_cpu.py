"""CPU, world_size 2 and 3, gloo: the multi-GPU split of the tiled VAE decode / encode (wan2gp_amd/vae.py:_sharded_tiles).  Spatial
tiles are independent until the blend (models/wan/modules/vae.py:676-717, :769-839, :841-881): rank k % world computes tile k,
every rank receives every tile and blends in the reference's order.  The per-tile HIP decode / encode is replaced IN THE TEST by a
torch stand-in (a fixed nonlinear function of the latent tile), so what is checked is the host logic: tile enumeration, ownership,
shapes a non-owner allocates for ragged edge tiles, exchange order -- every rank must reproduce the single-process tiled result
bit for bit, for the fp32 decode, the streaming uint8 decode and the tiled encode."""
import os
import socket
import types

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def make_vae(sp, calls):
    from wan2gp_amd.vae import WanVAEHIP
    v = object.__new__(WanVAEHIP)                       # no HIP library: only the tiling host logic is exercised
    v.device, v.upsampler_factor, v.z_dim, v.sp = torch.device("cpu"), 1, 16, sp

    def decode_frames(z, want_u8, want_f32):            # [16,t,h,w] -> fp32 [3,T,8h,8w]; depends on every latent value of the tile
        calls.append(("dec", tuple(z.shape)))
        T = (z.shape[1] - 1) * 4 + 1
        base = torch.nn.functional.interpolate(z[:3].unsqueeze(0), size=(T, z.shape[2] * 8, z.shape[3] * 8), mode="nearest")[0]
        return None, torch.tanh(base * 0.7 + 0.1 * z.mean())

    def encode(videos, tile_size=0, any_end_frame=False):
        if int(tile_size or 0) > 0:
            return WanVAEHIP.encode(v, videos, tile_size)
        outs = []
        for x in videos:
            calls.append(("enc", tuple(x.shape)))
            t = (x.shape[1] - 1) // 4 + 1
            y = torch.nn.functional.adaptive_avg_pool3d(x.unsqueeze(0), (t, x.shape[2] // 8, x.shape[3] // 8))[0]
            outs.append(torch.cat([y] * 5 + [y[:1]], dim=0) * (1 + 0.01 * x.std()))
        return outs
    v._decode_frames = decode_frames
    v.encode = encode
    return v


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        g = torch.Generator().manual_seed(3)
        z = torch.randn(16, 2, 19, 27, generator=g)                              # ragged: edge tiles are narrower and shorter
        vid = torch.rand(3, 5, 72, 104, generator=g) * 2 - 1
        ref_calls, calls = [], []
        ref = make_vae(None, ref_calls)
        sh = make_vae(types.SimpleNamespace(rank=rank, world=world, group=None), calls)
        res = {}
        for ts in (64, 48):
            a, b = ref._tiled_decode_f32(z.clone(), ts), sh._tiled_decode_f32(z.clone(), ts)
            res[f"f32_{ts}"] = torch.equal(a, b)
            a, b = ref._tiled_decode_u8(z.clone(), ts), sh._tiled_decode_u8(z.clone(), ts)
            res[f"u8_{ts}"] = torch.equal(a, b) and a.dtype == torch.uint8
        a, b = ref.encode([vid.clone()], 64)[0], sh.encode([vid.clone()], 64)[0]
        res["enc"] = torch.equal(a, b) and tuple(a.shape) == (16, 2, 9, 13)
        # each rank computed its share only: tiles k with k % world == rank
        n_ref, n_own = len(ref_calls), len(calls)
        res["share"] = (n_ref, n_own)
        q.put((rank, res, None))
    except Exception as e:                                                      # pragma: no cover
        import traceback
        q.put((rank, None, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_tiled_vae_sharded_over_ranks_equals_single_process(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    shares = []
    for rank, res, err in sorted(got):
        assert err is None, err
        n_ref, n_own = res.pop("share")
        shares.append(n_own)
        assert all(res.values()), (rank, res)
        assert n_own < n_ref                                                   # nobody did all the work ...
    assert sum(shares) == n_ref and max(shares) - min(shares) <= 5             # ... together exactly the single-process tile count
