"""Wan22VAEHIP -- drop-in for the reference `Wan2_2_VAE` (models/wan/modules/vae2_2.py:1144-1340), the 5B ti2v VAE:
z_dim 48, encoder dim 160, decoder dim 256, stride (4, 16, 16).

Same surface as `WanVAEHIP` (encode / decode / decode_to_cpu_uint8 / scale / model.z_dim / get_VAE_tile_size).  The graph is
the Wan2.1 one plus (vae2_2.py): 2x2 patchify in front of the encoder and unpatchify behind the decoder (:299-332, :806,
:879), `Down_ResidualBlock` / `Up_ResidualBlock` with parameter-free AvgDown3D / DupUp3D shortcuts around each
resolution level (:335-516), an upsample Conv2d that keeps its channel count (:108-117).  Convolutions, RMS_norm+SiLU and
the attention block are the vae_ops.hip kernels through `_VaeNet`; the four new pieces are vae22_ops.hip.  tile_size is
accepted and ignored (288 GB: the tile_size == 0 path, :971-975).  dtype=torch.float32 (`vae_precision` "32", wgp.py:4038) runs the same
graph in fp32 throughout (`_VaeNetF32` + the `_f32` forms of the four pieces), like the Wan2.1 VAE's fp32 plan.
"""
import torch

from .lib import check, ptr, stream_ptr
from .vae import F16, WanVAEHIP, _cache_update

MEAN22 = [-0.2289, -0.0052, -0.1323, -0.2339, -0.2799, 0.0174, 0.1838, 0.1557, -0.1382, 0.0542, 0.2813, 0.0891, 0.1570, -0.0098,
          0.0375, -0.1825, -0.2246, -0.1207, -0.0698, 0.5109, 0.2665, -0.2108, -0.2158, 0.2502, -0.2055, -0.0322, 0.1109, 0.1567,
          -0.0729, 0.0899, -0.2799, -0.1230, -0.0313, -0.1649, 0.0117, 0.0723, -0.2839, -0.2083, -0.0520, 0.3748, 0.0152, 0.1957,
          0.1433, -0.2944, 0.3573, -0.0548, -0.1681, -0.0667]
STD22 = [0.4765, 1.0364, 0.4514, 1.1677, 0.5313, 0.4990, 0.4818, 0.5013, 0.8158, 1.0344, 0.5894, 1.0901, 0.6885, 0.6165, 0.8454,
         0.4978, 0.5759, 0.3523, 0.7135, 0.6804, 0.5833, 1.4146, 0.8986, 0.5659, 0.7069, 0.5338, 0.4889, 0.4917, 0.4069, 0.4999,
         0.6866, 0.4093, 0.5709, 0.6065, 0.6415, 0.4944, 0.5726, 1.2042, 0.5458, 1.6887, 0.3971, 1.0600, 0.3943, 0.5537, 0.5444,
         0.4089, 0.7468, 0.7744]


def _pad32(c):
    return (c + 31) // 32 * 32


class Wan22VAEHIP(WanVAEHIP):
    NATIVE_GRAPH = False          # this graph (patchify, AvgDown / DupUp shortcuts, 48 latent channels) stays on the host
    SUPPORTS_F32 = True           # round 6: `vae_precision` "32" -- this graph on _VaeNetF32 (csrc/vae_f32.hip) and the wan_vae22_*_f32 pieces
    CFG = dict(dim=160, dec_dim=256, z_dim=48, dim_mult=[1, 2, 4, 4], num_res_blocks=2, temperal_downsample=[False, True, True])

    @staticmethod
    def get_VAE_tile_size(vae_config, device_mem_capacity, mixed_precision, output_height=None, output_width=None):
        """vae2_2.py:1286-1307 picks 0 / 256 / 128; this class decodes and encodes untiled whatever it is given (see the module
        docstring), so it reports the size it honours."""
        return 0

    def __init__(self, z_dim=48, c_dim=160, vae_pth=None, dim_mult=(1, 2, 4, 4), temperal_downsample=(False, True, True),
                 dtype=torch.float16, upsampler_factor=1, device="cuda", state_dict=None, dec_dim=256, **unused):
        assert upsampler_factor == 1                                     # vae2_2.py:1160
        self.CFG = dict(dim=c_dim, dec_dim=dec_dim, z_dim=z_dim, dim_mult=list(dim_mult), num_res_blocks=2,
                        temperal_downsample=list(temperal_downsample))
        super().__init__(z_dim=z_dim, vae_pth=vae_pth, dtype=dtype, device=device, state_dict=state_dict)
        self.mean = torch.tensor(MEAN22[:z_dim], dtype=torch.float32, device=self.device)
        self.std = torch.tensor(STD22[:z_dim], dtype=torch.float32, device=self.device)
        self.scale = [self.mean, 1.0 / self.std]

    def _op22(self, name):
        """wan_vae22_<name> of the plan in use."""
        return getattr(self.net.lib, f"wan_vae22_{name}_f32" if self.dtype == torch.float32 else f"wan_vae22_{name}")

    # ---- Down_ResidualBlock / Up_ResidualBlock (vae2_2.py:434-516) --------------------------------------------------------
    def _down_block(self, x, pre, cout, t_down, down, cache, idx):
        n, nres = self.net, self.CFG["num_res_blocks"]
        x0 = x
        for j in range(nres):
            x = self._res(x, f"{pre}{j}.", cache, idx)
        if down:
            p = f"{pre}{nres}."
            x = n.conv(x, p + "resample.1", st_s=2, pad_s=0)             # ZeroPad2d((0,1,0,1)) + stride 2 (:122-124)
            if t_down:                                                   # downsample3d (:178-189)
                j = idx[0]
                if cache[j] is None:
                    cache[j] = x[-1:].clone()
                else:
                    cx = x[-1:].clone()
                    prev2 = torch.cat([torch.zeros_like(cache[j]), cache[j]], 0)
                    x = n.conv(x, p + "time_conv", cache=prev2, st_t=2, front=1, pad_s=0)
                    cache[j] = cx
                idx[0] += 1
        T, H, W, C = x0.shape
        check(self._op22("avgdown_add")(ptr(x0), ptr(x), T, H, W, C, cout, 2 if t_down else 1, 2 if down else 1, stream_ptr()),
              "wan_vae22_avgdown_add")
        return x

    def _up_block(self, x, pre, cout, t_up, up, cache, idx, first_chunk):
        n, nres = self.net, self.CFG["num_res_blocks"]
        xm = x
        for j in range(nres + 1):
            xm = self._res(xm, f"{pre}{j}.", cache, idx)
        if not up:
            return xm
        p = f"{pre}{nres + 1}."
        if t_up:                                                         # upsample3d (:131-170)
            j = idx[0]
            if cache[j] is None:
                cache[j] = "Rep"
            else:
                prev = None if isinstance(cache[j], str) else cache[j]
                cx = _cache_update(xm, prev)
                xm = n.conv(xm, p + "time_conv", cache=prev, interleave=True, pad_s=0)
                cache[j] = cx
            idx[0] += 1
        xm = n.conv(xm, p + "resample.1", ups=True)                      # nearest-exact 2x + Conv2d dim -> dim (:108-111)
        T, H, W, C = x.shape
        check(self._op22("dupup_add")(ptr(x), ptr(xm), T, H, W, C, cout, 2 if t_up else 1, 2, 1 if first_chunk else 0,
                                      stream_ptr()), "wan_vae22_dupup_add")
        return xm

    def _decoder(self, x, cache, idx, first_chunk=False):
        """Decoder3d.forward (vae2_2.py:691-742) on one latent frame [1,h,w,pad32(z)] -> fp32 [Ti, 8h, 8w, 12]."""
        n, cfg = self.net, self.CFG
        mult = cfg["dim_mult"]
        tus = cfg["temperal_downsample"][::-1]
        dd = [cfg["dec_dim"] * u for u in [mult[-1]] + mult[::-1]]
        x = self._cached_conv(x, "decoder.conv1", cache, idx)
        x = self._res(x, "decoder.middle.0.", cache, idx)
        x = n.attention_block(x, "decoder.middle.1.")
        x = self._res(x, "decoder.middle.2.", cache, idx)
        for i, cout in enumerate(dd[1:]):
            x = self._up_block(x, f"decoder.upsamples.{i}.upsamples.", cout, tus[i] if i < len(tus) else False, i != len(mult) - 1,
                               cache, idx, first_chunk)
        x = n.norm(x, "decoder.head.0.gamma")
        return self._cached_conv(x, "decoder.head.2", cache, idx, out_f32=True)

    def _encoder(self, x, cache, idx):
        """Encoder3d.forward (vae2_2.py:578-632) on a patchified chunk [t, H/2, W/2, 32]."""
        n, cfg = self.net, self.CFG
        mult, tds = cfg["dim_mult"], cfg["temperal_downsample"]
        dims = [cfg["dim"] * u for u in [1] + mult]
        x = self._cached_conv(x, "encoder.conv1", cache, idx)
        for i, cout in enumerate(dims[1:]):
            x = self._down_block(x, f"encoder.downsamples.{i}.downsamples.", cout, tds[i] if i < len(tds) else False,
                                 i != len(mult) - 1, cache, idx)
        x = self._res(x, "encoder.middle.0.", cache, idx)
        x = n.attention_block(x, "encoder.middle.1.")
        x = self._res(x, "encoder.middle.2.", cache, idx)
        x = n.norm(x, "encoder.head.0.gamma")
        return self._cached_conv(x, "encoder.head.2", cache, idx)

    # ---- WanVAE_.decode (vae2_2.py:845-880) -----------------------------------------------------------------------------
    def _decode_frames(self, z, want_u8, want_f32):
        lib = self.net.lib
        z = z.to(device=self.device, dtype=torch.float32).contiguous()          # [z_dim, t, h, w]
        C, t, h, w = z.shape
        Cp = _pad32(C)
        zp = torch.empty(t, h, w, Cp, dtype=self._adt, device=self.device)
        inv_std = (1.0 / self.scale[1]).contiguous()                             # z / scale[1] + scale[0]
        self._pack(z, zp, inv_std, self.scale[0].contiguous(), C, Cp, t * h * w)
        x = self.net.conv(zp, "conv2")                                           # 1x1x1, z -> z (padded to a multiple of 32)
        T_out = (t - 1) * 4 + 1
        H, W = h * 16, w * 16
        u8 = torch.empty(3, T_out, H, W, dtype=torch.uint8, device=self.device) if want_u8 else None
        f32 = torch.empty(3, T_out, H, W, dtype=torch.float32, device=self.device) if want_f32 else None
        cache = [None] * self._n_cached("decoder.")
        t0 = 0
        for i in range(t):
            y = self._decoder(x[i:i + 1], cache, [0], first_chunk=(i == 0))      # fp32 [Ti, 8h, 8w, 12]
            Ti = y.shape[0]
            check(lib.wan_vae22_to_video(ptr(y), ptr(u8), ptr(f32), Ti, h * 8, w * 8, T_out, t0, stream_ptr()), "wan_vae22_to_video")
            t0 += Ti
        assert t0 == T_out, (t0, T_out)
        return u8, f32

    # ---- WanVAE_.encode (vae2_2.py:802-842) -----------------------------------------------------------------------------
    def encode(self, videos, tile_size=0, any_end_frame=False):
        if any_end_frame:
            raise NotImplementedError("any_end_frame encode is outside the hot path")
        lib = self.net.lib
        zd = self.z_dim
        outs = []
        for v in videos:
            v = v.to(device=self.device, dtype=torch.float32).contiguous()       # [3, T, H, W]
            C, T, H, W = v.shape
            vp = torch.empty(T, H // 2, W // 2, 32, dtype=self._adt, device=self.device)
            check(self._op22("patchify")(ptr(v), ptr(vp), T, H, W, 32, stream_ptr()), "wan_vae22_patchify")
            cache = [None] * self._n_cached("encoder.")
            chunks = []
            for i in range(1 + (T - 1) // 4):
                xc = vp[:1] if i == 0 else vp[1 + 4 * (i - 1):1 + 4 * i]
                chunks.append(self._encoder(xc, cache, [0]))
            enc = torch.cat(chunks, 0)                                           # [t, h, w, 2z]
            mu = self.net.conv(enc, "conv1")                                     # 1x1x1 2z -> 2z; mu = first z channels
            t, h, w, Cs = mu.shape
            out = torch.empty(zd, t, h, w, dtype=torch.float32, device=self.device)
            self._unpack(mu, out, self.scale[0].contiguous(), self.scale[1].contiguous(), zd, Cs, t * h * w)
            outs.append(out)
        return outs
