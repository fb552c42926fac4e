"""TEST INFRASTRUCTURE ONLY -- tests/golden/vae22_small.npz from the REFERENCE's own Wan2.2 WanVAE_
(models/wan/modules/vae2_2.py), fp32 on CPU.   Run in the build container:   python oracle/make_golden_vae22.py
Small config (dim 32, dec_dim 32, z 16; same graph as the 5B VAE: dim_mult [1,2,4,4], temporal downsample [F,T,T]), the
seeded weights of oracle/vae22_oracle.synth_vae22_weights, seeded inputs.  Also records the reference's patchify /
AvgDown3D / DupUp3D modules on their own."""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402
from oracle import vae22_oracle as V2  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "vae22_small.npz")


def load_ref():
    ns = ref_shim.load()
    sys.modules["models.wan.modules.vae"] = ns.V                       # vae2_2.py:9 imports three helpers from .vae
    spec = importlib.util.spec_from_file_location("models.wan.modules.vae2_2", os.path.join(ref_shim.REF_ROOT, "models/wan/modules/vae2_2.py"))
    m = importlib.util.module_from_spec(spec)
    sys.modules["models.wan.modules.vae2_2"] = m
    spec.loader.exec_module(m)
    return m


def main():
    R = load_ref()
    cfg = V2.SMALL
    W = V2.synth_vae22_weights(cfg=cfg)
    vae = R.WanVAE_(dim=cfg["dim"], dec_dim=cfg["dec_dim"], z_dim=cfg["z_dim"], dim_mult=cfg["dim_mult"], num_res_blocks=cfg["num_res_blocks"],
                    attn_scales=[], temperal_downsample=cfg["temperal_downsample"]).eval()
    vae.load_state_dict({k: v.clone() for k, v in W.items()}, strict=True)
    scale = V2.default_scale(z_dim=cfg["z_dim"])
    g = torch.Generator().manual_seed(31)
    z = torch.randn(1, cfg["z_dim"], 3, 4, 4, generator=g)
    vid = torch.rand(1, 3, 9, 64, 64, generator=g) * 2 - 1
    vid[:, :, 1:] *= 0.5
    with torch.no_grad():
        dec = vae.decode(z, scale)
        u8 = vae.decode_to_cpu_uint8(z, scale, 0)
        enc = vae.encode(vid, scale)
        x = torch.randn(1, 8, 4, 6, 6, generator=g)
        out = {"dec": dec.numpy(), "dec_u8": u8.numpy(), "enc": enc.numpy(),
               "patch_in": vid[:, :, :2, :8, :8].numpy(), "patch_out": R.patchify(vid[:, :, :2, :8, :8], 2).numpy(),
               "unpatch_out": R.unpatchify(R.patchify(vid[:, :, :2, :8, :8], 2), 2).numpy(),
               "avg_in": x.numpy(), "avg_t2s2": R.AvgDown3D(8, 16, 2, 2)(x).numpy(), "avg_t1s2": R.AvgDown3D(8, 16, 1, 2)(x).numpy(),
               "avg_t2s2_odd": R.AvgDown3D(8, 16, 2, 2)(x[:, :, :1]).numpy(), "avg_t1s1": R.AvgDown3D(8, 4, 1, 1)(x).numpy(),
               "dup_t2s2": R.DupUp3D(8, 4, 2, 2)(x).numpy(), "dup_t2s2_first": R.DupUp3D(8, 4, 2, 2)(x, True).numpy(),
               "dup_t1s2": R.DupUp3D(8, 4, 1, 2)(x).numpy(), "seed": np.array([31])}
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items()}, "dec range", float(dec.min()), float(dec.max()))


if __name__ == "__main__":
    main()
