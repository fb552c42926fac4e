"""Diagnostics: one VAE convolution launched again and again on fixed input -- bit-reproducible beside another process on the same GPU?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if "--lib" in sys.argv:      # another build of the library (file name under wan2gp_amd/)
    from wan2gp_amd import lib as _lib
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), sys.argv[sys.argv.index("--lib") + 1])
from wan2gp_amd.vae import WanVAEHIP, random_vae_state_dict
vae = WanVAEHIP(state_dict=random_vae_state_dict())
n = vae.net
tag, iters = sys.argv[1], int(sys.argv[2])
g = torch.Generator().manual_seed(3)
names = [k for k in n.convs if k.startswith("decoder.") and n.convs[k].k == (3, 3, 3)]
pick = {"head": "decoder.head.2"}
for k in names:
    c = n.convs[k]
    if c.cin == 96 and c.cout == 96 and "mid96" not in pick: pick["mid96"] = k
    if c.cin == 384 and c.cout == 384 and "mid384" not in pick: pick["mid384"] = k
for label, name in pick.items():
    c = n.convs[name]
    x = (torch.randn(4, 64, 64, c.cin, generator=g) * 0.5).to(torch.float16).cuda()
    f32 = label == "head"
    ref = n.conv(x, name, out_f32=f32).clone()
    bad = 0
    for it in range(iters):
        o = n.conv(x, name, out_f32=f32)
        ne = int((o != ref).sum())
        if ne:
            bad += 1
            if bad <= 3: print(tag, label, name, "iteration", it, "differing elements", ne, "maxdiff", (o.float() - ref.float()).abs().max().item(), flush=True)
    print(tag, label, name, "cin", c.cin, "cout", c.cout, ": %d of %d launches differ" % (bad, iters), flush=True)
