"""Build-time guard for the two hot kernels: the properties their performance rests on are visible in the gfx950 ISA hipcc
emits, so they are asserted here (CPU, cross-compile only, ~30 s):
  * gemm256k: 256 accumulators in the accumulator file, no scratch in any instantiation (a spill there costs the epilogue its
    prefetch), 160 KB of LDS, and the main loop's instruction mix per stage -- 64 MFMAs, 32
    ds_read_b128, 16 LDS-DMA pieces, one counted wait + barrier;
  * attention w64q: no scratch, accumulators in the accumulator file, one v_exp_f32 per score element and packed bf16
    conversions (no software rounding) in the steady-state segments."""
import collections
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")


_ASM = {}


def asm_of(name, tmp_path_factory, defines=()):
    key = (name, tuple(defines))
    if key not in _ASM:
        _ASM[key] = _asm_of(name, tmp_path_factory, defines)
    return _ASM[key]


def _asm_of(name, tmp_path_factory, defines=()):
    out = tmp_path_factory.mktemp("isa") / (name + ".s")
    src = os.path.join(ROOT, "wan2gp_amd", "csrc", name + ".hip")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-ffp-contract=on", *defines,
                    "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", src, "-o", str(out)], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return open(out).read()


def kernels(asm, stem):
    """name -> (body ops, metadata dict) for every kernel whose mangled name contains `stem`."""
    res = {}
    for m in re.finditer(r"^(_Z\S*" + stem + r"\S*):", asm, re.M):
        name = m.group(1)
        end = asm.index(".Lfunc_end", m.end())                      # a kernel may hold several s_endpgm (early returns)
        ops = [l.split()[0] for l in (x.strip() for x in asm[m.end():end].split("\n")) if l and not l.startswith((";", ".")) and not l.endswith(":")]
        tail = asm[end:end + 6000]
        meta = {k: int(v) for k, v in re.findall(r"; (NumVgprs|NumAgprs|ScratchSize|LDSByteSize): (\d+)", tail)}
        res[name] = (ops, meta)
    return res


@pytest.fixture(scope="module")
def gemm(tmp_path_factory):
    return kernels(asm_of("gemm256k", tmp_path_factory), "gemm256k_kernel")


@pytest.fixture(scope="module")
def attn(tmp_path_factory):
    return kernels(asm_of("attention_w64q", tmp_path_factory), "attn_w64q_kernel")


def test_gemm256k_registers_lds_and_no_scratch(gemm):
    assert len(gemm) == 6                                            # NONE / GELU / GATE_RES, bias rows, fp16 variants
    for name, (ops, meta) in gemm.items():
        assert meta["ScratchSize"] == 0, name
        assert meta["NumAgprs"] >= 252 and meta["LDSByteSize"] == 160 * 1024, (name, meta)


def test_gemm256k_main_loop_instruction_mix(gemm):
    name = next(n for n in gemm if "ILi0ELb0ELb0E" in n)            # EPI NONE, column bias, bf16
    ops = gemm[name][0]
    mf = [i for i, o in enumerate(ops) if o.startswith("v_mfma_f32_32x32x16")]
    assert len(mf) == 5 * 64                                         # loop unrolled over the five ring positions
    br = [i for i, o in enumerate(ops) if o.startswith(("s_cbranch", "s_branch"))]
    stages = []
    prev = 0
    for b in br + [len(ops)]:
        if sum(1 for i in mf if prev <= i < b) == 64:
            stages.append(collections.Counter(ops[prev:b]))
        prev = b
    assert len(stages) == 5
    for j, c in enumerate(stages):
        if j == 0:                                                   # the first segment also holds the tail of the prologue
            assert c["ds_read_b128"] >= 32 and c["buffer_load_dwordx4"] >= 16
            continue
        assert c["ds_read_b128"] == 32 and c["buffer_load_dwordx4"] == 16 and c["s_barrier"] == 1
        valu = sum(v for k, v in c.items() if k.startswith("v_") and not k.startswith("v_mfma"))
        assert valu <= 24, valu                                      # address adds only: nothing competes with the MFMAs for issue slots
        assert not any(k.startswith("scratch_") for k in c)


def test_attention_w64q_no_scratch_and_hardware_conversions(attn):
    # {bounded, tracking} x {q pre-scaled, pre-scaling pass} x {one kv segment, several (FLAGS bit 6: the long form of the DMA
    # stream's step)} + the two partial-sum forms of sequence parallelism (RAW_OUT on one segment, CARRY_IN over the others)
    assert len(attn) == 10, sorted(attn)
    for name, (ops, meta) in attn.items():
        assert meta["ScratchSize"] == 0 and meta["NumAgprs"] == 256 and meta["LDSByteSize"] <= 96 * 1024 + 256, (name, meta)
    # tracking loop (FLAGS 2): 68 MFMAs per KV tile (4 carry -m_ref), one v_exp_f32 per score, hardware bf16 packing
    c = collections.Counter(attn[next(n for n in attn if "ILi2E" in n)][0])
    assert c["v_exp_f32_e32"] >= 3 * 64 and c["v_cvt_pk_bf16_f32"] >= 3 * 32 and c["v_max3_f32"] >= 3 * 32
    assert sum(v for k, v in c.items() if k.startswith("v_mfma")) >= 3 * 68


def test_attention_bounded_loop_instruction_mix(attn):
    """The DiT's self-attention (FLAGS 6 = pre-scaled q, bounded softmax): per 64-kv tile of a wave exactly 64 MFMAs, 64
    v_exp_f32, 32 packed conversions, 64 plain v_add_f32 row-sum updates (NOT v_pk_add_f32: packed f32 VALU beside MFMAs is an
    anti-lever), 32 ds_read_b128, 8 LDS-DMA pieces, one barrier and ONE vmcnt wait (the tile top: an LDS-DMA issued through the
    builtin makes hipcc serialise the ring with vmcnt(0) in front of ds_reads) -- and no row-max / compare / rescale instruction."""
    ops, meta = attn[next(n for n in attn if "ILi6E" in n)]
    assert meta["NumVgprs"] <= 216, meta
    bars = [i for i, o in enumerate(ops) if o == "s_barrier"]
    tiles = []
    for a, b in zip(bars, bars[1:]):
        c = collections.Counter(ops[a:b])
        if sum(v for k, v in c.items() if k.startswith("v_mfma")) == 64:
            tiles.append(c)
    assert len(tiles) >= 2, [sum(v for k, v in collections.Counter(ops[a:b]).items() if k.startswith("v_mfma")) for a, b in zip(bars, bars[1:])]
    for c in tiles:
        assert c["v_exp_f32_e32"] == 64 and c["v_cvt_pk_bf16_f32"] == 32 and c["v_add_f32"] == 64 and c["v_pk_add_f32"] == 0, c
        assert c["ds_read_b128"] == 32 and c["buffer_load_dwordx4"] == 8, c
        assert not any(k.startswith(("v_max", "v_cmp", "v_permlane", "scratch_")) for k in c), c
        valu = sum(v for k, v in c.items() if k.startswith("v_") and not k.startswith("v_mfma"))
        assert valu <= 168, valu                                     # 160 + a few address / mask ops


def test_attention_bounded_tile_scalar_and_wait_instructions(tmp_path_factory):
    """One wave per SIMD issues at most one instruction per 4 cycles of ANY kind, so scalar bookkeeping and waits compete with the
    exps for the MFMA gaps.  Properties of the single-segment bounded kernel (FLAGS 6) that cost matrix-pipe time when lost:
    (a) the DMA stream's step is the short form: a tile issues < 50 scalar ALU instructions and no v_cndmask / v_readfirstlane
        (the segment walk is ~60 scalar instructions + lane-mask round trips, sunk by LLVM into ONE MFMA gap);
    (b) at most 4 s_waitcnt per tile: every K(t+1) fragment is read >= 15 MFMAs before the tile ends, so the single lgkmcnt(0)
        at the next tile's top never stalls and replaces the per-MFMA counted waits (19 before);
    (c) <= 325 instructions per tile in all (335 before)."""
    asm = asm_of("attention_w64q", tmp_path_factory)
    m = re.search(r"^(_Z\S*attn_w64q_kernelILi6E\S*):", asm, re.M)
    body = [l.strip() for l in asm[m.end():asm.index(".Lfunc_end", m.end())].split("\n")]
    body = [l.split(";")[0].strip() for l in body if l and not l.startswith((";", ".")) and not l.endswith(":")]
    bars = [i for i, l in enumerate(body) if l.startswith("s_barrier")]
    tiles = [(a, b) for a, b in zip(bars, bars[1:]) if sum(1 for l in body[a:b] if l.startswith("v_mfma")) == 64]
    assert len(tiles) >= 2
    for a, b in tiles:
        ops = [l.split()[0] for l in body[a:b]]
        salu = [o for o in ops if o.startswith("s_") and not o.startswith(("s_waitcnt", "s_barrier", "s_cbranch", "s_nop"))]
        assert len(salu) < 50, len(salu)
        assert not any(o.startswith(("v_cndmask", "v_readfirstlane")) for o in ops)
        assert ops.count("s_waitcnt") <= 4, [l for l in body[a:b] if l.startswith("s_waitcnt")]
        assert len(ops) <= 325, len(ops)
        mf = [i for i, o in enumerate(ops) if o.startswith("v_mfma")]
        assert "ds_read_b128" not in ops[mf[49]:], "an LDS read in the last 15 MFMA gaps of the tile"


def test_attention_w16n_tile_instruction_mix(tmp_path_factory):
    """attention_w16n.hip (the shipped bounded self-attention loop since round 3, 16x16x32 MFMA): every instantiation without scratch
    and with the whole accumulator file; between two barriers of the single-segment pre-scaled kernel (FLAGS 6) exactly one tile: 128
    MFMAs, 64 v_exp_f32, 64 plain v_add_f32 (never packed), 32 v_cvt_pk_bf16_f32, 32 ds_read_b128, 8 LDS-DMA pieces, no max / compare /
    permute, <= 6 s_waitcnt, <= 12 s_nop beside the DMA pieces' own, and no MFMA gap with more than 6 instructions (the scalar step of
    the DMA stream is spread over the gaps behind the barrier)."""
    asm = asm_of("attention_w16n", tmp_path_factory)
    ks = kernels(asm, "attn_w16n_kernel")
    assert len(ks) == 16                                             # six plain + their six shifted twins (FLAGS | 128, round 4) + the two persistent short-KV forms (| 256) + the split tail's two (round 6: segmented parts, one-segment finish)
    for name, (ops, meta) in ks.items():
        persist = any(t in name for t in ("ILi388E", "ILi390E"))
        assert meta["ScratchSize"] == 0 and meta["NumAgprs"] == 256, (name, meta)
        if persist:   # the ring + 64 KB for the next block's Q rows: all of the LDS (its votes go through the ring, not through __syncthreads_or's own 256 bytes)
            assert meta["LDSByteSize"] == 160 * 1024 and meta["NumVgprs"] <= 256, (name, meta)
        else:
            assert 96 * 1024 <= meta["LDSByteSize"] <= 97 * 1024, (name, meta)  # the ring + the workgroup vote
            assert meta["NumVgprs"] <= 216, (name, meta)             # the shifted twins carry 16 more (the C tuples): 448 / 464 of 512 with the accumulators
    tiles = []
    for tag in ("ILi6E", "ILi134E", "ILi390E"):                      # the single-segment pre-scaled kernel, plain, shifted and persistent: the SAME tile
        ops = ks[[n for n in ks if tag in n][0]][0]
        bars = [i for i, o in enumerate(ops) if o == "s_barrier"]
        mine = [ops[a:b] for a, b in zip(bars, bars[1:]) if sum(1 for o in ops[a:b] if o.startswith("v_mfma")) == 128]
        assert len(mine) >= 2                                        # the ring of three: two barrier-to-barrier spans inside the loop
        tiles += mine
    for t in tiles:
        c = collections.Counter(t)
        assert c["v_mfma_f32_16x16x32_bf16"] == 128 and c["v_exp_f32_e32"] == 64 and c["v_cvt_pk_bf16_f32"] == 32, c
        assert c["v_add_f32_e32"] + c["v_add_f32"] == 64 and c["v_pk_add_f32"] == 0, c
        assert c["ds_read_b128"] == 32 and c["buffer_load_dwordx4"] in (8, 10), c   # (10: the persistent form's two Q pieces at the tile's top)
        assert not any(k.startswith(("v_max", "v_permlane", "scratch_", "v_readfirstlane", "v_cndmask")) for k in c), c
        assert c["buffer_load_dwordx4"] == 10 or not any(k.startswith("v_cmp") for k in c), c
        assert c["s_waitcnt"] <= 6 and c["s_nop"] <= 8 + 12 + 8 + (2 if c["buffer_load_dwordx4"] == 10 else 0), c     # (the shifted twin: 7 more wait-state nops in exp2-only gaps)
        mf = [i for i, o in enumerate(t) if o.startswith("v_mfma")]
        gaps = [b - a - 1 for a, b in zip(mf, mf[1:])]
        assert max(gaps) <= 6, gaps
        assert len(t) <= (440 if c["buffer_load_dwordx4"] == 10 else 400), len(t)


def test_gemm256m_registers_and_stage_instruction_mix(tmp_path_factory):
    """gemm256m.hip (the product GEMM since round 3, 16x16x32 MFMA): no scratch, the whole LDS, and between two barriers of the main loop
    exactly one stage.  The 256-row tile (TY = 8): 256 accumulators in the accumulator file, 128 MFMAs, 32 ds_read_b128, 16 LDS-DMA pieces
    and NO vector-ALU instruction (a 16-cycle MFMA gap hides two issue slots; an address computation there is a stall).  Round 6: the tile
    height is a template argument -- TY = 5, 6, 7 (160 / 192 / 224 rows): 32 TY accumulators, 16 TY MFMAs, 2 (TY + 8) fragment reads and
    TY + 8 pieces per stage, likewise without vector-ALU work."""
    ks = kernels(asm_of("gemm256m", tmp_path_factory), "gemm256m_kernel")
    assert len(ks) == 17                                             # {NONE, GELU, GATE_RES, the row-bias (V^T) form} x TY 5..8, and the fp32-stream residual (TY 8)
    seen = collections.Counter()
    for name, (ops, meta) in ks.items():
        ty = int(re.search(r"Li(\d)EEEv", name).group(1))
        seen[ty] += 1
        assert meta["ScratchSize"] == 0 and meta["NumAgprs"] == 32 * ty and meta["LDSByteSize"] == 160 * 1024, (name, meta)
        bars = [i for i, o in enumerate(ops) if o == "s_barrier"]
        stages = [collections.Counter(ops[a:b]) for a, b in zip(bars, bars[1:])]
        stages = [c for c in stages if c["v_mfma_f32_16x16x32_bf16"] == 16 * ty]
        assert len(stages) >= 4, name                                # the loop is unrolled by 5: four barrier-to-barrier spans inside it
        for c in stages:
            assert c["ds_read_b128"] == 2 * (ty + 8) and c["buffer_load_dwordx4"] == ty + 8, c
            valu = sum(v for k, v in c.items() if k.startswith("v_") and not k.startswith("v_mfma"))
            assert valu == 0, c
            assert c["s_waitcnt"] <= 20, c
    assert seen == {8: 5, 7: 4, 6: 4, 5: 4}, seen


def test_attention_xkv_stream_has_no_valu_write_in_front_of_an_mfma_that_reads_it(tmp_path_factory):
    """attention_xkv.hip's MFMAs are inline asm: hipcc pads no hazard around them.  The one it can create by itself is a register copy (v_mov /
    v_accvgpr) placed straight in front of an MFMA that reads the copy -- it did, when an accumulator's old value was kept alive across the
    next tile's first MFMA (registers 2, 3 of every O^T tile came out stale on hardware).  Asserted on the ISA: one kernel, 64 MFMAs per
    iteration, K / V^T in all 256 accumulator registers, no scratch, 32 v_exp_f32, and no MFMA whose previous instruction writes one of its
    source registers."""
    asm = asm_of("attention_xkv", tmp_path_factory)
    ks = kernels(asm, "attn_xkv_kernel")
    assert len(ks) == 1
    (name, (ops, meta)), = ks.items()
    assert meta["ScratchSize"] == 0 and meta["NumAgprs"] == 256, meta
    assert ops.count("v_mfma_f32_16x16x32_bf16") == 64 and ops.count("v_exp_f32_e32") == 32 and ops.count("s_barrier") == 3, collections.Counter(ops).most_common(12)
    assert not [o for o in ops if o.startswith(("v_accvgpr", "scratch_", "flat_load", "flat_store"))]
    body = asm[asm.index(name + ":"):]
    lines = [l.strip() for l in body[:body.index(".Lfunc_end")].split("\n") if l.strip() and not l.strip().startswith((";", "."))]

    def regs(tok):
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
        if m:
            return set(range(int(m.group(1)), int(m.group(2)) + 1))
        m = re.fullmatch(r"v(\d+)", tok)
        return {int(m.group(1))} if m else set()

    bad = []
    for prev, cur in zip(lines, lines[1:]):
        if not cur.startswith("v_mfma") or not prev.startswith("v_") or prev.startswith("v_mfma"):
            continue
        written = regs(prev.split(None, 1)[1].split(",")[0].strip())
        srcs = set()
        for tok in cur.split(None, 1)[1].split(",")[1:]:
            srcs |= regs(tok.strip())
        if written & srcs:
            bad.append((prev, cur))
    assert not bad, bad[:4]


def test_no_kernel_of_the_library_spills_to_scratch(tmp_path_factory):
    """Every kernel of every translation unit: ScratchSize 0 (a spill inside a tile loop is a performance cliff, and the hot kernels
    run at the edge of the 512-register file).  No exceptions: round 2's one (the scaled-fp8 GEMM with GELU and a per-row weight
    scale parked an accumulator tile in scratch at the start of its epilogue) is gone since the epilogue pins each accumulator tile
    to the accumulator file until the statement that converts it."""
    import concurrent.futures as cf
    import glob
    srcs = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(ROOT, "wan2gp_amd", "csrc", "*.hip")))
    with cf.ThreadPoolExecutor(max_workers=8) as ex:
        asms = dict(zip(srcs, ex.map(lambda n: asm_of(n, tmp_path_factory), srcs)))
    seen, offenders = 0, []
    for unit, asm in asms.items():
        for m in re.finditer(r"^(_Z\S+):", asm, re.M):
            end = asm.find(".Lfunc_end", m.end())
            if end < 0:
                continue
            meta = re.search(r"; ScratchSize: (\d+)", asm[end:end + 6000])
            if meta is None:
                continue
            seen += 1
            if int(meta.group(1)) > 0:
                offenders.append((unit, m.group(1), int(meta.group(1))))
    assert seen >= 100 and not offenders, offenders
    assert "gemm256m" in asms and asms["gemm256m"].count("gemm256m_kernel") >= 4


def test_row_kernels_hold_no_packed_instruction_that_reads_a_register_pair_crosswise(tmp_path_factory):
    """Round 6, runs 80-86: `v_pk_mul_f32 d, a, b op_sel:[0,1] op_sel_hi:[1,0]` -- what hipcc's SLP vectoriser made of RoPE's
    (-x1 sin0, x0 sin1) -- returned a LOW result of zero in lanes 48-63 of a few waves while another process on the same GPU started or exited:
    every wrong element of the narrow in-place RMSNorm + RoPE launch was x0 cos0 without its - x1 sin0.  One LDS word per wave shielded it (not
    understood); the rotation now works on aligned pairs (E = x0 of two pairs, O = x1 of two pairs) and the modulation sums are plain adds.  Held here:
    (1) no kernel of elementwise.hip / mixed_ops.hip holds a packed instruction whose LOW lane selects a HIGH source register (an op_sel bit), except
    the three timestep-sinusoid kernels (the device library's sin / cos), which hold an LDS word; (2) in the whole library every kernel that holds
    such an instruction is an LDS-holding workgroup; (3) the control: the build of rounds 3-5 (ROPE_FORM 0) does compile to the instruction."""
    import concurrent.futures as cf
    import glob
    cross = re.compile(r"^\s*(v_pk_\w+) .*op_sel:\[[01,]*1[01,]*\]", re.M)

    def per_kernel(asm):
        found = {}
        for m in re.finditer(r"^(_Z\S+):", asm, re.M):
            end = asm.find(".Lfunc_end", m.end())
            if end > 0:
                n = len(cross.findall(asm[m.end():end]))
                if n:
                    found[m.group(1)] = (n, int(re.search(r"; LDSByteSize: (\d+)", asm[end:end + 6000]).group(1)))
        return found

    srcs = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(ROOT, "wan2gp_amd", "csrc", "*.hip")))
    with cf.ThreadPoolExecutor(max_workers=8) as ex:
        asms = dict(zip(srcs, ex.map(lambda n: asm_of(n, tmp_path_factory), srcs)))
    found = {unit: per_kernel(asm) for unit, asm in asms.items()}
    rows = {k: v for unit in ("elementwise", "mixed_ops") for k, v in found[unit].items()}
    assert sorted(k for k in rows if "sinusoid" not in k) == [], rows                                   # (1)
    assert len(rows) == 3 and all(lds > 0 for _, lds in rows.values()), rows
    without_lds = {(unit, k): v for unit, ks in found.items() for k, v in ks.items() if v[1] == 0}
    assert not without_lds, without_lds                                                                 # (2)
    assert sum(len(ks) for ks in found.values()) >= 10                                                  # (the pattern still finds the GEMM / VAE ones)
    asm = asms["elementwise"]
    rope = [m.group(1) for m in re.finditer(r"^(_Z19rmsnorm_rope_kernel\S+):", asm, re.M)]
    assert len(rope) >= 20
    body = asm[asm.index(rope[0] + ":"):]
    assert body[:body.index(".Lfunc_end")].count("v_pk_mul_f32") >= 8          # the packed design is still there, on aligned pairs
    old = per_kernel(asm_of("elementwise", tmp_path_factory, defines=("-DROPE_FORM=0",)))
    assert sum(n for k, (n, _) in old.items() if "rmsnorm_rope_kernel" in k) >= 500                      # (3)
