#!/usr/bin/env python
"""Per-MFMA-gap view of a kernel's hot loop in the gfx950 ISA hipcc emits (tuning tool; needs hipcc, no GPU).

With one wave per SIMD the sequencer issues at most one instruction of that wave every 4 cycles, of ANY kind, so what sits between
two consecutive MFMAs -- VALU, LDS, LDS-DMA, scalar bookkeeping, s_waitcnt, s_nop, branches -- is the budget of that 32-cycle gap.
This prints, for one barrier-to-barrier segment of a kernel, the instructions of every gap (scalar ALU folded into a count) and the
segment's totals.  It is how the second session of round 2 found the 60-instruction scalar lump and the loop-header lgkmcnt(0) in
the attention tile.

    python tools/isa_tile_view.py attention_w64q attn_w64q_kernelILi6E            # bounded self-attention, single segment
    python tools/isa_tile_view.py gemm256k gemm256k_kernelILi0ELb0ELb0E --segment 2
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHORT = {"v_exp_f32_e32": "EXP", "v_add_f32": "ADD", "v_cvt_pk_bf16_f32": "CVT", "ds_read_b128": "DSR", "buffer_load_dwordx4": "DMA"}


def asm_of(unit, defines):
    out = os.path.join(tempfile.mkdtemp(prefix="isa_"), unit + ".s")
    src = os.path.join(ROOT, "wan2gp_amd", "csrc", unit + ".hip")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-ffp-contract=on",
                    "-I" + os.path.join(ROOT, "include"), *defines, "-S", "--cuda-device-only", src, "-o", out], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return open(out).read()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("unit", help="translation unit under wan2gp_amd/csrc (without .hip)")
    ap.add_argument("kernel", help="substring of the kernel's mangled name")
    ap.add_argument("--segment", type=int, default=None, help="index of the barrier-to-barrier segment (default: the first with >= 64 MFMAs)")
    ap.add_argument("-D", action="append", default=[], help="extra -D defines")
    a = ap.parse_args()
    asm = asm_of(a.unit, ["-D" + d for d in a.D])
    m = next((m for m in re.finditer(r"^(_Z\S+):", asm, re.M) if a.kernel in m.group(1)), None)
    if m is None:
        sys.exit(f"no kernel matching {a.kernel!r} in {a.unit}.hip")
    end = asm.index(".Lfunc_end", m.end())
    meta = dict(re.findall(r"; (NumVgprs|NumAgprs|ScratchSize|LDSByteSize): (\d+)", asm[end:end + 6000]))
    body = [l.split(";")[0].strip() for l in asm[m.end():end].split("\n")]
    body = [l for l in body if l and not l.startswith(".") and not l.endswith(":")]
    bars = [i for i, l in enumerate(body) if l.startswith("s_barrier")]
    segs = list(zip(bars, bars[1:] + [len(body)]))          # the last segment runs to the end of the kernel (single-barrier loops)
    if a.segment is None:
        a.segment = next((i for i, (x, y) in enumerate(segs) if sum(1 for l in body[x:y] if l.startswith("v_mfma")) >= 64), 0)
    x, y = segs[a.segment]
    print(m.group(1)[:100], meta, f"segment {a.segment} of {len(segs)}: instructions {x}..{y}")
    gap, row = -1, []

    def flush():
        salu = sum(1 for r in row if r.startswith("s_") and not r.startswith(("s_waitcnt", "s_barrier", "s_cbranch", "s_nop")))
        rest = [r for r in row if not (r.startswith("s_") and not r.startswith(("s_waitcnt", "s_barrier", "s_cbranch", "s_nop")))]
        print(f"{gap:3d}  " + " | ".join(rest) + (f"   [scalar ALU x{salu}]" if salu else ""))
    for l in body[x:y]:
        op = l.split()[0]
        if op.startswith("v_mfma"):
            flush()
            gap, row = gap + 1, []
            continue
        row.append(SHORT.get(op, l if op.startswith(("s_waitcnt", "s_cbranch", "v_")) else op))
    flush()
    c = collections.Counter(l.split()[0] for l in body[x:y])
    salu = sum(v for k, v in c.items() if k.startswith("s_") and not k.startswith(("s_waitcnt", "s_barrier", "s_cbranch", "s_nop")))
    print(f"total {y - x} instructions: MFMA {sum(v for k, v in c.items() if k.startswith('v_mfma'))}, other VALU "
          f"{sum(v for k, v in c.items() if k.startswith('v_') and not k.startswith('v_mfma'))}, ds_read {sum(v for k, v in c.items() if k.startswith('ds_read'))}, "
          f"LDS-DMA {c['buffer_load_dwordx4']}, scalar ALU {salu}, s_waitcnt {c['s_waitcnt']}, s_nop {c['s_nop']}, branches "
          f"{sum(v for k, v in c.items() if k.startswith('s_cbranch'))}")


if __name__ == "__main__":
    main()
