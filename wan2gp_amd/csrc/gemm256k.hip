// bf16 / fp16 NT GEMM for gfx950, fourth generation: a one-wave-per-SIMD 256x256 tile (4 waves, each 128x128 = 4x4 MFMA 32x32x16
// tiles, 256 accumulators in the accumulator file) with FULL-LINE operand fetches.  Out[y][x] = epilogue( sum_k Y[y][k] * X[x][k] + bias ), same contract / epilogues as the other generations.
//
// Why: tools/probes/line_probe.hip streams the operand panels of a 16x16 grid of 256x256 tiles through LDS-DMA with no
// MFMA work at all and 64 KB in flight per CU.  With 64 B per row per piece (the k-tile of 32 that gemm32 and the
// removed third generation fetch) a CU ingests 24.2 B/clk -- the L2 -> L1 path moves whole 128-B lines, half of each is thrown away and
// re-fetched one k-tile later -- with 128 B per row it ingests 38.0 B/clk (13.5 vs 20.8 TB/s chip-wide).  A 256x256 tile
// needs 32 B/clk/CU at full MFMA rate: the BK=32 kernels are capped at 76 % before any other loss.
//
// So: k-tile of 64 (128 B = one line per row per fetch, 8 lanes of a DMA piece cover a row), 64 KB per stage.  Two
// stages do not leave enough in flight (the last piece issued would have ~1000 cycles to land), three do not fit, so
// the 160 KB of LDS are a ring of FIVE 32-KB units, a unit being one operand of one stage: Y0 X0 Y1 X1 Y2 | X2 Y3 ...
//   * sync point P_S sits in front of the LAST k-step of stage S (vmcnt(8) + barrier): stage S+1 is complete and visible,
//     and nobody reads stage S any more (its last fragments are already in registers) -> its two units are free;
//   * k-step 3 of S issues X_{S+2} (8 pieces per wave, every other MFMA gap) into Y_S's unit: 48 MFMAs = 1536 cycles
//     before P_{S+1} needs it; k-steps 0 and 1 of S+1 issue Y_{S+3} (4 pieces each) into X_S's unit: > 3000 cycles ahead;
//   * per MFMA: 0.5 ds_read_b128, 0.25 DMA pieces, no VALU; loop unrolled by 5 (unit index = 2S % 5).
// LDS rows are 128 B; physical 16-B chunk p of row r holds logical chunk p ^ ((r >> 1) & 7): the 16 lanes a ds_read_b128
// services together (rows {0-3,12-15,20-27} or {4-11,16-19,28-31} of a 32-row tile, same logical chunk) then cover all
// 64 banks exactly once.  The X operand is the first MFMA operand and its rows are staged in a permuted order (see the DMA plan
// below), so that a lane's 16 accumulators of one output row are 16 consecutive columns (gemm_bf16.hip's header).
#include <stdlib.h>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 g256k_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 g256k_f16x8;
typedef __attribute__((address_space(3))) const char g256k_lds_cchar;
typedef uint32_t g256k_u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const g256k_u4 g256k_lds_u4;

constexpr int G_BM = 256, G_BN = 256, G_BK = 64;
constexpr int G_UNIT = 256 * G_BK * 2;  // 32 KiB: one operand of one stage (256 rows x 128 B)
constexpr int G_NU = 5;                  // ring of five units: unit u (Y_S = 2S, X_S = 2S+1) lives in slot u % 5

__device__ __forceinline__ float g256k_gelu_tanh(float x) {
  const float c = -2.0f * 0.7978845608028654f * 1.4426950408889634f;  // -2 sqrt(2/pi) log2(e)
  const float t = __builtin_fmaf(x * x, 0.044715f, 1.0f);
  const float ex = __builtin_amdgcn_exp2f(x * t * c);
  return x * __builtin_amdgcn_rcpf(1.0f + ex);
}

struct G256kFrags {
  g256k_u4 y[4], x[4];
};
// accumulators pinned to the accumulator file; operands straight from ds_read_b128 (arch VGPRs)
template <bool F16>
__device__ __forceinline__ void mfma256k(f32x16& acc, const g256k_u4& a, const g256k_u4& b) {
  if (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

// LDS-DMA piece as inline asm: hipcc puts an s_waitcnt vmcnt(0) in front of every ds_read that follows a builtin
// LDS-DMA to the same array (it cannot tell the ring slots apart), which would serialise the prefetch; the asm form is
// invisible to its waitcnt pass -- completion is counted by hand (vmcnt(0) + barrier at the top of each k-tile).
__device__ __forceinline__ void g256k_dma16(uint32_t voff, const g256k_u4& rsrc, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rsrc), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ g256k_u4 g256k_rsrc(const void* base) {
  const uint64_t b = (uint64_t)base;
  g256k_u4 r;
  r[0] = (uint32_t)b;
  r[1] = (uint32_t)(b >> 32) & 0xffffu;  // stride 0
  r[2] = 0xffffffffu;                    // num_records: no range check needed (rows are clamped)
  r[3] = 0x00020000u;
  return r;
}

template <int EPI, bool BIAS_ROWS, bool F16>
__global__ __launch_bounds__(256) void gemm256k_kernel(const bf16_t* __restrict__ Y, int64_t ldy, int64_t YM,
                                                       const bf16_t* __restrict__ X, int64_t ldx, int64_t XN, int K,
                                                       bf16_t* __restrict__ Out, int64_t ldo, const bf16_t* __restrict__ bias,
                                                       const bf16_t* __restrict__ R, const bf16_t* __restrict__ mod,
                                                       const bf16_t* __restrict__ e, int n_mod, int gate_idx,
                                                       int64_t rows_per_batch, int tiles_y, int tiles_x, float out_scale, int group) {
  __shared__ __attribute__((aligned(16))) char smem[G_NU * G_UNIT];  // 160 KiB
  g256k_lds_cchar* lds = (g256k_lds_cchar*)smem;
#ifdef G256K_TIMING
  const uint64_t t_entry = __builtin_amdgcn_s_memtime();
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wy = wave >> 1, wx = wave & 1;
  const int l31 = lane & 31, half = lane >> 5;

  // ---- tile assignment: XCD-contiguous ids, then grouped (8 y-tiles per group) ordering -----------------------
  const int nwg = tiles_y * tiles_x;
  const int wg = xcd_remap(blockIdx.x, nwg);
  const int GROUP = group;  // y tiles per group: the 32 workgroups resident on an XCD cover GROUP y x 32/GROUP x tiles
  const int per_group = GROUP * tiles_x;
  const int gidx = wg / per_group;
  const int first_y = gidx * GROUP;
  const int gsz = min(tiles_y - first_y, GROUP);
  const int in_g = wg - gidx * per_group;
  const int ty = first_y + (in_g % gsz);
  const int tx = in_g / gsz;
  const int64_t y0 = (int64_t)ty * G_BM;
  const int64_t x0 = (int64_t)tx * G_BN;

  // ---- DMA plan: loop-invariant per-lane byte offsets relative to the tile's first row ---------------------------
  // A unit image is 256 rows x 8 chunks of 16 B; piece i (0..7) of wave w fills 16-B slots q = i*256 + w*64 + lane, i.e.
  // rows i*32 + w*8 .. +8, eight lanes per row = the row's whole 128-B line in one instruction.
  // descriptor bases: Y + y0*ldy (+ k), X + x0*ldx (+ k); offsets fit 32 bits (checked by the launcher)
  uint32_t yofs[8], xofs[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int q = i * 256 + tid;
    const int row = q >> 3, pch = q & 7;
    const int lch = pch ^ ((row >> 1) & 7);
    int64_t yr = y0 + row;
    if (yr > YM - 1) yr = YM - 1;  // ragged tile: re-read the last row (its results are never stored)
    yofs[i] = (uint32_t)((yr - y0) * ldy * 2 + lch * 16);
    const int slab = row >> 7, xt = (row >> 5) & 3, rho = row & 31;
    int64_t xr = x0 + slab * 128 + xt * 32 + 16 * ((rho >> 2) & 1) + (rho & 3) + 4 * (rho >> 3);
    if (xr > XN - 1) xr = XN - 1;
    xofs[i] = (uint32_t)((xr - x0) * ldx * 2 + lch * 16);
  }
  const char* ybase = reinterpret_cast<const char*>(Y + y0 * ldy);  // next Y unit to fetch
  const char* xbase = reinterpret_cast<const char*>(X + x0 * ldx);  // next X unit to fetch
  const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;  // LDS byte address of the ring
  const int nk = K / G_BK;
  int ky = 0, kx = 0;  // stages the two streams point at; they stop at the last one (later fetches re-read it into a dead unit)
  auto y_piece = [&](int slot, int p) { g256k_dma16(yofs[p], g256k_rsrc(ybase), smem_lds + slot * G_UNIT + (p * 256 + wave * 64) * 16); };
  auto x_piece = [&](int slot, int p) { g256k_dma16(xofs[p], g256k_rsrc(xbase), smem_lds + slot * G_UNIT + (p * 256 + wave * 64) * 16); };
  auto y_advance = [&]() { const bool ok = ky + 1 < nk; ybase += ok ? G_BK * 2 : 0; ky += ok ? 1 : 0; };
  auto x_advance = [&]() { const bool ok = kx + 1 < nk; xbase += ok ? G_BK * 2 : 0; kx += ok ? 1 : 0; };

  f32x16 acc[4][4];  // [y tile][x tile], accumulator file
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
      asm volatile("" : "+a"(acc[a][b]));
    }

  // ---- fragment addresses: k-step ks (0..3) reads logical chunk 2ks + half; (row >> 1) & 7 == (l31 >> 1) & 7 for every tile ----
  const int sw = (l31 >> 1) & 7;
  const int yaddr0 = (wy * 128 + l31) * 128 + ((half ^ sw) << 4);  // k-step 0; k-step ks = ^ (ks << 5)
  const int xaddr0 = (wx * 128 + l31) * 128 + ((half ^ sw) << 4);
  // fragment r = 0..7 (0..3: Y tiles, 4..7: X tiles) of k-step ks of the stage whose Y unit sits in slot sy (X unit in sx)
  auto load_frag = [&](G256kFrags& f, int sy, int sx, int ks, int r) {
    if (r < 4) f.y[r] = *(g256k_lds_u4*)(lds + (sy * G_UNIT + r * 4096) + (yaddr0 ^ (ks << 5)));
    else f.x[r - 4] = *(g256k_lds_u4*)(lds + (sx * G_UNIT + (r - 4) * 4096) + (xaddr0 ^ (ks << 5)));
  };

  // prologue: stages 0 and 1 (units 0..3); Y_2 is issued by stage 0's first k-steps like every later Y unit
#pragma unroll
  for (int p = 0; p < 8; ++p) y_piece(0, p);
  y_advance();
#pragma unroll
  for (int p = 0; p < 8; ++p) x_piece(1, p);
  x_advance();
#pragma unroll
  for (int p = 0; p < 8; ++p) y_piece(2, p);
  y_advance();
#pragma unroll
  for (int p = 0; p < 8; ++p) x_piece(3, p);
  x_advance();
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // stage 0 landed, stage 1 may be in flight
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#ifdef G256K_TIMING
  const uint64_t t_prologue = __builtin_amdgcn_s_memtime();
#endif
  G256kFrags f0, f1;
#pragma unroll
  for (int r = 0; r < 8; ++r) load_frag(f0, 0, 1, 0, r);
  // Stage S = 5i + J: Y in slot 2J % 5, X in (2J+1) % 5; 4 k-steps of 16 MFMAs, MFMA m multiplies (y tile m>>2, x tile m&3).
  //   k-step 0 (f0): Y_{S+2} pieces 0..3 -> slot (2J+4) % 5 after MFMA 0,4,8,12; fragments of k-step 1 -> f1 after the others
  //   k-step 1 (f1): Y_{S+2} pieces 4..7;                                          fragments of k-step 2 -> f0
  //   k-step 2 (f0): no DMA;                                            fragments of k-step 3 -> f1 after MFMAs 0..7
  //   P_S: vmcnt(8) (X_{S+1} and everything older landed; Y_{S+2} may fly), lgkmcnt(0) (stage S fully read) + barrier
  //   k-step 3 (f1): X_{S+2} pieces 0..7 -> slot 2J % 5 (= Y_S, dead now) after the even MFMAs; stage S+1's k-step-0
  //                  fragments -> f0 after the odd ones
#ifdef G256K_TIMING
  uint64_t stamp[7] = {};
#define G256K_STAMP(I) if (kt + J_ == 30) stamp[I] = __builtin_amdgcn_s_memtime();
#else
#define G256K_STAMP(I)
#endif
#define G256K_SB() __builtin_amdgcn_sched_barrier(0)
#define G256K_KSTEP_A(FU, FL, SLOTD, PBASE, SY, SX, KS)  /* 4 Y pieces + 8 fragment reads */                   \
  _Pragma("unroll") for (int m = 0; m < 16; ++m) {                                                              \
    mfma256k<F16>(acc[m >> 2][m & 3], FU.x[m & 3], FU.y[m >> 2]); G256K_SB();                                    \
    if ((m & 3) == 0) y_piece(SLOTD, (PBASE) + (m >> 2));                                                        \
    else if (m - (m >> 2) - 1 < 8) load_frag(FL, SY, SX, KS, m - (m >> 2) - 1);                                  \
    G256K_SB();                                                                                                 \
  }
#define G256K_STEP(J)                                                                                           \
  if (__builtin_expect(kt + (J) < nk, 1)) {                                                                     \
    constexpr int J_ = (J);                                                                                     \
    constexpr int SY = (2 * J_) % 5, SX = (2 * J_ + 1) % 5, NY = (2 * J_ + 2) % 5, NX = (2 * J_ + 3) % 5;        \
    constexpr int DY = (2 * J_ + 4) % 5, DX = (2 * J_) % 5;                                                      \
    G256K_STAMP(0)                                                                                              \
    G256K_KSTEP_A(f0, f1, DY, 0, SY, SX, 1)                                                                     \
    G256K_STAMP(1)                                                                                              \
    G256K_KSTEP_A(f1, f0, DY, 4, SY, SX, 2)                                                                     \
    y_advance();                                                                                                \
    G256K_STAMP(2)                                                                                              \
    _Pragma("unroll") for (int m = 0; m < 16; ++m) {                                                            \
      mfma256k<F16>(acc[m >> 2][m & 3], f0.x[m & 3], f0.y[m >> 2]); G256K_SB();                                  \
      if (m < 8) load_frag(f1, SY, SX, 3, m);  /* early: every LDS read of stage S has returned at P_S */        \
      G256K_SB();                                                                                               \
    }                                                                                                           \
    G256K_STAMP(3)                                                                                              \
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");                                                 \
    G256K_STAMP(4)                                                                                              \
    __builtin_amdgcn_s_barrier();                                                                               \
    asm volatile("" ::: "memory");                                                                              \
    G256K_STAMP(5)                                                                                              \
    _Pragma("unroll") for (int m = 0; m < 16; ++m) {                                                            \
      mfma256k<F16>(acc[m >> 2][m & 3], f1.x[m & 3], f1.y[m >> 2]); G256K_SB();                                  \
      if ((m & 1) == 0) x_piece(DX, m >> 1);                                                                     \
      else load_frag(f0, NY, NX, 0, m >> 1);                                                                     \
      G256K_SB();                                                                                               \
    }                                                                                                           \
    x_advance();                                                                                                \
    G256K_STAMP(6)                                                                                              \
  }
  for (int kt = 0; kt < nk; kt += 5) {
    G256K_STEP(0)
    G256K_STEP(1)
    G256K_STEP(2)
    G256K_STEP(3)
    G256K_STEP(4)
  }
#undef G256K_STEP
#undef G256K_KSTEP_A
#undef G256K_SB
#ifdef G256K_TIMING
  if (blockIdx.x == 0 && tid == 0) {  // tuning aid: s_memtime stamps of k-tile 60 -> row 0 of tile (0,0), whose epilogue is skipped
    uint64_t* dbg = reinterpret_cast<uint64_t*>(Out);
    for (int i = 0; i < 7; ++i) dbg[i] = stamp[i];
  }
  if (blockIdx.x == 0) return;
  const uint64_t t_loop = __builtin_amdgcn_s_memtime();
#endif
  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");  // trailing DMA; last asm MFMAs -> accumulator reads of the epilogue

  // ---- epilogue, through LDS --------------------------------------------------------------------------------------------
  // Measured with row-per-lane stores (each lane writing 32 B of its own row, a store instruction
  // touching 32 rows): 30.5k cycles per tile = 13.5 % of a K = 5120 tile, three times what 128 KB cost at the rate a CU
  // can store.  The ring is dead now, so each wave parks its 128 x 128 result in its own 34-KB LDS region (bias / GELU
  // applied in the accumulator layout, rows of 256 B + 16 B pad) and reads it back row-major: a store instruction then
  // writes 4 rows x 256 contiguous bytes, and the residual / gate operands of the gated epilogue are fetched in the same
  // coalesced shape, two 4-row groups ahead.
  __builtin_amdgcn_s_barrier();  // every wave is past its last fragment read: the ring may be overwritten
  asm volatile("" ::: "memory");
  constexpr int EROW = 272;      // bytes per parked row: 256 + 16 (ds_write_b128 of 16 lanes on consecutive rows: 64 banks once)
  char* const park = smem + wave * (128 * EROW);
  // read-back geometry: instruction i covers rows 4i .. 4i+3 of the wave's sub-tile, lane -> (row 4i + lane/16, 16-B chunk lane%16)
  const int prow = lane >> 4, pchunk = lane & 15;
  const int64_t xc = x0 + wx * 128 + pchunk * 8;          // first of the lane's 8 columns: the same for every row group
  const bool col_full = xc + 8 <= XN;
  const int64_t yrow0 = y0 + wy * 128 + prow;
  // Gated epilogue: the residual (and the per-batch gate row) of a 4-row group is one 16-byte load per lane from HBM.  GDEPTH
  // groups are kept in flight during the read-back (the accumulators' registers are free by then; issued before the parking
  // phase they cost an accumulator spill).  With two groups in flight (the first version) this epilogue was latency-bound:
  // 65 % matrix-pipe utilisation on the o-projection against 82 % for the plain epilogue (PMC).
  constexpr int GDEPTH = 8;
  uint4 rq[GDEPTH] = {}, eq[GDEPTH] = {};
  // (values returned, not written through references: arrays handed to a lambda by reference were left in scratch memory)
  auto fetch_r = [&](int i) -> uint4 {
    uint4 z = {};
    if (EPI != WAN_EPI_GATE_RES || !col_full) return z;
    int64_t yr = yrow0 + i * 4;
    if (yr > YM - 1) yr = YM - 1;
    return *reinterpret_cast<const uint4*>(R + yr * ldo + xc);
  };
  auto fetch_e = [&](int i) -> uint4 {
    uint4 z = {};
    if (EPI != WAN_EPI_GATE_RES || !col_full || gate_idx < 0) return z;
    int64_t yr = yrow0 + i * 4;
    if (yr > YM - 1) yr = YM - 1;
    const uint32_t bidx = (uint32_t)yr / (uint32_t)rows_per_batch;
    return *reinterpret_cast<const uint4*>(e + ((int64_t)bidx * n_mod + gate_idx) * XN + xc);
  };
#pragma unroll
  for (int xt = 0; xt < 4; ++xt) {
    const int64_t xb = x0 + wx * 128 + xt * 32 + half * 16;
    float bcol[16];
    if (!BIAS_ROWS) {
#pragma unroll
      for (int j = 0; j < 16; ++j) bcol[j] = 0.f;
      if (bias != nullptr) {
        if (xb + 16 <= XN) {
          unpack8t<F16>(*reinterpret_cast<const uint4*>(bias + xb), bcol);
          unpack8t<F16>(*reinterpret_cast<const uint4*>(bias + xb + 8), bcol + 8);
        } else {  // ragged x edge
#pragma unroll
          for (int j = 0; j < 16; ++j)
            if (xb + j < XN) bcol[j] = ld16<F16>(bias[xb + j]);
        }
      }
    }
#pragma unroll
    for (int yt = 0; yt < 4; ++yt) {
      int64_t yr = y0 + wy * 128 + yt * 32 + l31;
      if (yr > YM - 1) yr = YM - 1;
      float v[16];
      const float brow = (BIAS_ROWS && bias != nullptr) ? ld16<F16>(bias[yr]) : 0.f;
      const f32x16 av = acc[yt][xt];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        v[r] = av[r] * out_scale + (BIAS_ROWS ? brow : bcol[r]);
        // nn.Linear output is a 16-bit tensor: GELU sees the rounded value; otherwise the pack below is that rounding
        if (EPI == WAN_EPI_GELU_TANH) v[r] = g256k_gelu_tanh(rnd16<F16>(v[r]));
      }
      uint4* dst = reinterpret_cast<uint4*>(park + (yt * 32 + l31) * EROW + (xt * 32 + half * 16) * 2);
      dst[0] = pack8t<F16>(v);
      dst[1] = pack8t<F16>(v + 8);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the region is private to the wave: its own writes are all it waits for
  uint4 mchunk = {};
  if (EPI == WAN_EPI_GATE_RES && gate_idx >= 0 && col_full) mchunk = *reinterpret_cast<const uint4*>(mod + (int64_t)gate_idx * XN + xc);
#pragma unroll
  for (int i = 0; i < GDEPTH; ++i) { rq[i] = fetch_r(i); eq[i] = fetch_e(i); }
#ifdef G256K_NT_STORE  // experiment (round 3, run 62): non-temporal stores for the 128 KB a tile writes once
#define G256K_ST(PTR, VAL) __builtin_nontemporal_store((VAL), (PTR))
#else
#define G256K_ST(PTR, VAL) (*(PTR) = (VAL))
#endif
  auto emit = [&](int i, const uint4 rq_, const uint4 eq_) {
    const int64_t yr = yrow0 + i * 4;
    const uint4 raw = *reinterpret_cast<const uint4*>(park + (i * 4 + prow) * EROW + pchunk * 16);
    if (yr >= YM) return;
    bf16_t* optr = Out + yr * ldo + xc;
    if (col_full) {
      if (EPI == WAN_EPI_GATE_RES) {
        float v[8], rv[8];
        unpack8t<F16>(raw, v);
        unpack8t<F16>(rq_, rv);
        if (gate_idx >= 0) {
          float mv[8], ev[8];
          unpack8t<F16>(mchunk, mv);
          unpack8t<F16>(eq_, ev);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = rv[j] + v[j] * rnd16<F16>(mv[j] + ev[j]);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = rv[j] + v[j];
        }
        G256K_ST(reinterpret_cast<g256k_u4*>(optr), __builtin_bit_cast(g256k_u4, pack8t<F16>(v)));
      } else {
        G256K_ST(reinterpret_cast<g256k_u4*>(optr), __builtin_bit_cast(g256k_u4, raw));
      }
    } else if (EPI == WAN_EPI_NONE) {
      // ragged x edge: only the transposed / V^T form (x = tokens, EPI NONE) can hit it -- the launcher requires
      // N % 16 == 0 for every other epilogue
      const uint32_t w4[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (xc + j < XN) optr[j] = (bf16_t)(w4[j >> 1] >> ((j & 1) * 16));
    }
  };
#pragma unroll
  for (int i = 0; i < 32; ++i) {  // fully unrolled: the slot index i % GDEPTH is a constant, the arrays stay in registers
    emit(i, rq[i % GDEPTH], eq[i % GDEPTH]);
    if (i + GDEPTH < 32) { rq[i % GDEPTH] = fetch_r(i + GDEPTH); eq[i % GDEPTH] = fetch_e(i + GDEPTH); }
  }
#ifdef G256K_TIMING
  if (blockIdx.x == 40 && tid == 0) {  // whole-tile timeline of one ordinary workgroup -> row 1 of tile (0,0) (whose epilogue is skipped)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint64_t* dbg = reinterpret_cast<uint64_t*>(Out + ldo);
    dbg[0] = t_entry; dbg[1] = t_prologue; dbg[2] = t_loop; dbg[3] = __builtin_amdgcn_s_memtime();
  }
#endif
}

}  // namespace

// Returns -1 when the problem does not fit this kernel (the caller falls back to gemm32.hip / gemm_bf16.hip), else the
// launch status.
template <int EPI, bool BIAS_ROWS, bool F16>
int wan_gemm256k_try(const bf16_t* Y, int64_t ldy, int64_t YM, const bf16_t* X, int64_t ldx, int64_t XN, int K, bf16_t* Out,
                    int64_t ldo, const bf16_t* bias, const bf16_t* R, const bf16_t* mod, const bf16_t* e, int n_mod,
                    int gate_idx, int64_t rows_per_batch, hipStream_t st, float out_scale) {
  if (K % G_BK != 0) return -1;
  // 32-bit DMA offsets: a tile's 256 rows times the row pitch in bytes, plus the row itself
  if (256 * ldy * 2 + (int64_t)K * 2 >= ((int64_t)1 << 32) || 256 * ldx * 2 + (int64_t)K * 2 >= ((int64_t)1 << 32)) return -1;
  const int64_t ty = (YM + G_BM - 1) / G_BM, tx = (XN + G_BN - 1) / G_BN;
  if (ty * tx >= ((int64_t)1 << 31)) return -1;
  // y tiles per group of the tile order (measured: 4 beats 8 by 2-7 % on the y = tokens shapes; 16 / 32 lose 10-25 %)
  const int group = BIAS_ROWS ? 8 : 4;
  hipLaunchKernelGGL((gemm256k_kernel<EPI, BIAS_ROWS, F16>), dim3((unsigned)(ty * tx)), dim3(256), 0, st, Y, ldy, YM, X, ldx,
                     XN, K, Out, ldo, bias, R, mod, e, n_mod, gate_idx, rows_per_batch, (int)ty, (int)tx, out_scale, group);
  WAN_LAUNCH_CHECK();
  return 0;
}

#define G256K_INST(EPI, BR, F)                                                                                              \
  template int wan_gemm256k_try<EPI, BR, F>(const bf16_t*, int64_t, int64_t, const bf16_t*, int64_t, int64_t, int, bf16_t*, \
                                           int64_t, const bf16_t*, const bf16_t*, const bf16_t*, const bf16_t*, int, int,  \
                                           int64_t, hipStream_t, float);
G256K_INST(WAN_EPI_NONE, false, false)
G256K_INST(WAN_EPI_GELU_TANH, false, false)
G256K_INST(WAN_EPI_GATE_RES, false, false)
G256K_INST(WAN_EPI_NONE, true, false)
G256K_INST(WAN_EPI_NONE, false, true)
G256K_INST(WAN_EPI_NONE, true, true)
#undef G256K_INST
