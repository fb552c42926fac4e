"""Diagnostics (CPU): what ARE the wrong elements the LDS-less narrow in-place RMSNorm + RoPE kernel of rounds 3-5 left beside a neighbour process?
Reads an extract of the rows dumped by tools/probes/rmsrope_twice.py --dump (run 82; profiles/r06_rmsrope_wrong_rows_run82_extract.npz: source rows,
first launch's rows, the wrong launch's rows, their cos / sin rows) and restates the kernel's arithmetic in torch: every wrong element is an EVEN element
of a rotation pair and equals x0 cos0 WITHOUT its - x1 sin0 -- the low result of `v_pk_mul_f32 ... op_sel:[0,1] op_sel_hi:[1,0]` came out as zero.
  python tools/probes/rmsrope_wrong_rows_analysis.py [extract.npz]          (make the extract from a dump:  ... --extract dump.pt out.npz)"""
import sys
import numpy as np
import torch
BF = torch.bfloat16
if len(sys.argv) > 1 and sys.argv[1] == "--extract":
    D = torch.load(sys.argv[2]); n = 48; L = D["L"]; pos = (D["rows"][:n] % L)
    np.savez_compressed(sys.argv[3], rows=D["rows"][:n].numpy(), out=D["out"][:n].view(torch.int16).numpy(), ref1=D["ref1"][:n].view(torch.int16).numpy(),
                        src=D["src"][:n].view(torch.int16).numpy(), wq=D["wq"].view(torch.int16).numpy(), cos=D["cos"].view(-1, 128)[pos].numpy(),
                        sin=D["sin"].view(-1, 128)[pos].numpy(), q_scale=np.float32(D["q_scale"]))
    sys.exit(0)
Z = np.load(sys.argv[1] if len(sys.argv) > 1 else "profiles/r06_rmsrope_wrong_rows_run82_extract.npz")
bf = lambda a: torch.from_numpy(a.copy()).view(BF).float()
out, ref1, src, wq = bf(Z["out"]), bf(Z["ref1"]), bf(Z["src"]), bf(Z["wq"])
cos, sin, qs = torch.from_numpy(Z["cos"]), torch.from_numpy(Z["sin"]), float(Z["q_scale"])
d = out.shape[1]
r = torch.rsqrt((src * src).sum(1, keepdim=True) / d + 1e-6)
y = ((src * r).to(BF).float() * wq).to(BF).float()          # x * rsqrt -> bf16 -> * weight -> bf16   (model.py:160-175)
col = torch.arange(d) % 128
c, s = cos[:, col], sin[:, col]
y0, y1, c0, c1, s0, s1 = y[:, 0::2], y[:, 1::2], c[:, 0::2], c[:, 1::2], s[:, 0::2], s[:, 1::2]
a, b = y0 * c0, y1 * s0
even_ref, even_out, odd_ref, odd_out = ref1[:, 0::2], out[:, 0::2], ref1[:, 1::2], out[:, 1::2]
rb = lambda t: (t * qs).to(BF).float()
print("the restatement reproduces the first launch: even elements %.4f, odd elements %.4f of all" % (float((rb(a - b) == even_ref).float().mean()), float((rb(y1 * c1 + y0 * s1) == odd_ref).float().mean())))
wrong = even_out != even_ref
print("wrong even elements:", int(wrong.sum()), "; wrong odd elements:", int((odd_out != odd_ref).sum()))
for name, v in (("x0 cos0 - x1 sin0 (right)", a - b), ("x0 cos0 + x1 sin0", a + b), ("x0 cos0 ALONE", a), ("- x1 sin0 alone", -b)):
    print("  wrong elements equal to %-28s %d" % (name + ":", int((rb(v)[wrong] == even_out[wrong]).sum())))
lanes = sorted(set(((torch.nonzero(wrong)[:, 1] // 4) % 64).tolist()))
print("lanes (16-byte chunk index mod 64) holding wrong elements:", lanes)
