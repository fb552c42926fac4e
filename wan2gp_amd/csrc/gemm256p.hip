// bf16 NT GEMM for gfx950, fifth generation: gemm256k.hip's 256x256x64 one-wave-per-SIMD tile loop made PERSISTENT.
// Out[y][x] = epilogue( sum_k Y[y][k] * X[x][k] + bias[x] ), same contract / epilogues as the other generations (bf16, bias per
// column; the transposed V^T form and the fp16 VAE GEMMs stay on gemm256k.hip).
//
// Why: the stamps of gemm256k (profiles/r02, r03) put 6.3k cycles of prologue and 10.5k of epilogue beside the ~184k-cycle loop
// of a K = 5120 tile -- 8 % of every such tile with nothing else resident on the CU to hide it (one workgroup owns the whole
// register file and LDS).  The vendor library's hand-tuned kernel for these shapes is a persistent stream-K loop whose epilogue
// stores straight from registers.  Here:
//   * one workgroup per CU walks tiles bidv = it * gridDim + blockIdx of the same XCD-contiguous grouped order; the LDS-DMA
//     stream is CONTINUOUS across tiles -- it runs two stages ahead of the MFMAs, so while the last two stages of a tile are
//     being multiplied the first two stages of the next tile are already landing in the units that freed up: no prologue;
//   * the MFMA operands swap roles (Y rows are the A operand, X rows the B operand) and the X rows are staged 4-way interleaved
//     (LDS row xt*32 + j of a wave's slab holds global column 4 j + xt), so that after the loop a lane holds, for each of its 64
//     (row-pair) slots, FOUR CONSECUTIVE output columns: the epilogue packs them to 8 bytes and stores them straight from
//     registers -- the 32 lanes of a half-wave write 256 contiguous bytes of one row, a store instruction 2 rows x 256 B (whole
//     128-B lines).  No LDS parking: the ring keeps the next tile's stages during the epilogue, and the stores drain under the
//     next tile's first MFMAs (the stage's DMA wait becomes vmcnt(63) once per tile: the 64 stores are younger than the pieces
//     it waits for and the counter has 6 bits);
//   * rows / columns beyond the matrix are fetched through the buffer descriptor's range check (zeros), so the per-lane DMA
//     offsets do not depend on the tile.
// Everything inside a stage -- ring of five 32-KB units, sync point in front of the last k-step, issue slots of the DMA pieces
// and fragment reads -- is gemm256k.hip's (see its header); only the operand order of the MFMAs differs.
#include <stdlib.h>

#include "common.h"

namespace {

typedef __attribute__((address_space(3))) const char g256p_lds_cchar;
typedef uint32_t g256p_u4 __attribute__((ext_vector_type(4)));
typedef uint32_t g256p_u2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const g256p_u4 g256p_lds_u4;

constexpr int P_BM = 256, P_BN = 256, P_BK = 64;
constexpr int P_UNIT = 256 * P_BK * 2;  // 32 KiB: one operand of one stage (256 rows x 128 B)
constexpr int P_NU = 5;                  // ring of five units

__device__ __forceinline__ float g256p_gelu_tanh(float x) {
  const float c = -2.0f * 0.7978845608028654f * 1.4426950408889634f;  // -2 sqrt(2/pi) log2(e)
  const float t = __builtin_fmaf(x * x, 0.044715f, 1.0f);
  const float ex = __builtin_amdgcn_exp2f(x * t * c);
  return x * __builtin_amdgcn_rcpf(1.0f + ex);
}

struct G256pFrags {
  g256p_u4 y[4], x[4];
};
// D[i][j] += sum_k A[i][k] B[j][k]: A = the Y fragment (i = output row), B = the X fragment (j = output column): lane (j, half)
// holds column j of rows 8 (r >> 2) + (r & 3) + 4 half in register r.  Accumulators pinned to the accumulator file.
__device__ __forceinline__ void mfma256p(f32x16& acc, const g256p_u4& ya, const g256p_u4& xb) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(ya), "v"(xb));
}
// LDS-DMA piece as inline asm (invisible to hipcc's waitcnt pass, see gemm256k.hip); completion is counted by hand
// The LDS address of a piece = the wave's base + a compile-time offset.  The base is laundered through an empty asm at every
// use: left visible, hipcc hoists all 16 pieces x 5 slots of addresses into SGPRs for the whole kernel and the scalar file
// (102) overflows into VGPR lanes and scratch; this way each address is one s_add right in front of its piece.
__device__ __forceinline__ void g256p_dma16(uint32_t voff, const g256p_u4& rsrc, uint32_t lds_wave_base, uint32_t off) {
  asm volatile("" : "+s"(lds_wave_base));
  const uint32_t lds_addr = lds_wave_base + off;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rsrc), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ g256p_u4 g256p_rsrc(const char* base, uint32_t num_records) {
  const uint64_t b = (uint64_t)base;
  g256p_u4 r;
  r[0] = (uint32_t)b;
  r[1] = (uint32_t)(b >> 32) & 0xffffu;  // stride 0
  r[2] = num_records;                    // bytes from base that may be read; beyond: zeros (range check) = rows past the matrix
  r[3] = 0x00020000u;
  return r;
}
__device__ __forceinline__ int p_uni(int x) { return __builtin_amdgcn_readfirstlane(x); }

#ifdef G256P_TIMING
__device__ uint64_t g256p_stamps[16];  // tuning aid: s_memtime stamps of workgroup 40's third tile (tools/gemm_stamp.py --persistent)
#define P_STAMP(I) do { if (blockIdx.x == 40 && it == 2 && threadIdx.x == 0) g256p_stamps[I] = __builtin_amdgcn_s_memtime(); } while (0)
#define P_STAMP_NEXT(I) do { if (blockIdx.x == 40 && it == 3 && threadIdx.x == 0) g256p_stamps[I] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define P_STAMP(I)
#define P_STAMP_NEXT(I)
#endif

struct TileXY {
  int ty, tx;
};
// the tile order of gemm256k.hip: XCD-contiguous ids, then grouped (GROUP y-tiles per group)
__device__ __forceinline__ TileXY tile_of(int bidv, int nwg, int tiles_y, int tiles_x, int GROUP) {
  const int wg = xcd_remap(bidv, nwg);
  const int per_group = GROUP * tiles_x;
  const int gidx = wg / per_group;
  const int first_y = gidx * GROUP;
  const int gsz = min(tiles_y - first_y, GROUP);
  const int in_g = wg - gidx * per_group;
  TileXY t;
  t.ty = p_uni(first_y + (in_g % gsz));
  t.tx = p_uni(in_g / gsz);
  return t;
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm256p_kernel(const bf16_t* __restrict__ Y, int64_t ldy, int64_t YM,
                                                       const bf16_t* __restrict__ X, int64_t ldx, int64_t XN, int K,
                                                       bf16_t* __restrict__ Out, int64_t ldo, const bf16_t* __restrict__ bias,
                                                       const bf16_t* __restrict__ R, const bf16_t* __restrict__ mod,
                                                       const bf16_t* __restrict__ e, int n_mod, int gate_idx,
                                                       int64_t rows_per_batch, int tiles_y, int tiles_x, float out_scale, int group) {
  __shared__ __attribute__((aligned(16))) char smem[P_NU * P_UNIT];  // 160 KiB
  g256p_lds_cchar* lds = (g256p_lds_cchar*)smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wy = wave >> 1, wx = wave & 1;
  const int l31 = lane & 31, half = lane >> 5;
  const int nwg = tiles_y * tiles_x;
  const int nk = K / P_BK;
  const int G = (int)gridDim.x;

  // ---- DMA plan: tile-independent per-lane byte offsets relative to the tile's first row of each operand -------------------
  // A unit image is 256 rows x 8 chunks of 16 B; piece i (0..7) of wave w fills 16-B slots q = i*256 + w*64 + lane, i.e. rows
  // i*32 + w*8 .. +8, eight lanes per row = the row's whole 128-B line in one instruction; physical chunk p of row r holds
  // logical chunk p ^ ((r >> 1) & 7).  Y rows in place; X row (slab, xt, rho) = column slab*128 + 4 rho + xt of the tile.
  uint32_t yofs[8], xofs[8];
  const uint32_t ldy2 = (uint32_t)(ldy * 2), ldx2 = (uint32_t)(ldx * 2);   // 32-bit on purpose (the launcher checks 256 rows fit): 64-bit
#pragma unroll                                                              // products were kept as register PAIRS for the whole kernel
  for (int i = 0; i < 8; ++i) {
    const int q = i * 256 + tid;
    const int row = q >> 3, pch = q & 7;
    const int lch = pch ^ ((row >> 1) & 7);
    uint32_t ty_ = (uint32_t)row * ldy2;
    asm volatile("" : "+v"(ty_));  // product and sum apart: fused, hipcc emits v_mad_u64_u32 and keeps the 64-bit PAIR allocated
    yofs[i] = ty_ + (uint32_t)(lch * 16);
    const int slab = row >> 7, xt = (row >> 5) & 3, rho = row & 31;
    uint32_t tx_ = (uint32_t)(slab * 128 + 4 * rho + xt) * ldx2;
    asm volatile("" : "+v"(tx_));
    xofs[i] = tx_ + (uint32_t)(lch * 16);
  }
  const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  // ---- the two DMA streams (wave-uniform state): position = (tile, stage); they run two stages ahead of the MFMAs --------------
  auto panel_bytes = [&](int64_t rows_total, int t, int64_t ld) -> uint32_t {  // valid bytes of a 256-row panel from its first row, per stage
    int64_t v = rows_total - (int64_t)t * 256;
    if (v > 256) v = 256;
    return (uint32_t)((v - 1) * ld * 2 + P_BK * 2);
  };
  int it = 0;                                   // tile iteration the MFMAs are at
  TileXY cur = tile_of(blockIdx.x, nwg, tiles_y, tiles_x, group);
  const char* ybase = reinterpret_cast<const char*>(Y + (int64_t)cur.ty * P_BM * ldy);
  const char* xbase = reinterpret_cast<const char*>(X + (int64_t)cur.tx * P_BN * ldx);
  uint32_t ynum = panel_bytes(YM, cur.ty, ldy), xnum = panel_bytes(XN, cur.tx, ldx);
  int ky = 0, kx = 0;
  // where the streams go when they run off the current tile: the next tile of this workgroup (or the same tile again after the
  // last one: re-read into dead units, never multiplied)
  const char* ynext = ybase;
  const char* xnext = xbase;
  uint32_t ynext_num = ynum, xnext_num = xnum;
  TileXY nxt = cur;
  bool has_next = false;
  auto plan_next = [&]() {
    const int bidn = (it + 1) * G + (int)blockIdx.x;
    has_next = bidn < nwg;
    if (has_next) {
      nxt = tile_of(bidn, nwg, tiles_y, tiles_x, group);
      ynext = reinterpret_cast<const char*>(Y + (int64_t)nxt.ty * P_BM * ldy);
      xnext = reinterpret_cast<const char*>(X + (int64_t)nxt.tx * P_BN * ldx);
      ynext_num = panel_bytes(YM, nxt.ty, ldy);
      xnext_num = panel_bytes(XN, nxt.tx, ldx);
    } else {  // stay on this tile's panels (stage 0 again)
      ynext = reinterpret_cast<const char*>(Y + (int64_t)cur.ty * P_BM * ldy);
      xnext = reinterpret_cast<const char*>(X + (int64_t)cur.tx * P_BN * ldx);
      ynext_num = panel_bytes(YM, cur.ty, ldy);
      xnext_num = panel_bytes(XN, cur.tx, ldx);
    }
  };
  plan_next();
  const uint32_t lds_wave = smem_lds + (uint32_t)wave * 1024u;   // + slot * P_UNIT + piece * 4096 per DMA piece
#define y_piece(SLOT, P) g256p_dma16(yofs[P], g256p_rsrc(ybase, ynum), lds_wave, (uint32_t)((SLOT) * P_UNIT + (P) * 4096))
#define x_piece(SLOT, P) g256p_dma16(xofs[P], g256p_rsrc(xbase, xnum), lds_wave, (uint32_t)((SLOT) * P_UNIT + (P) * 4096))
  auto y_advance = [&]() {
    const bool wrap = ky + 1 == nk;
    ybase = wrap ? ynext : ybase + P_BK * 2;
    ynum = wrap ? ynext_num : ynum;
    ky = wrap ? 0 : ky + 1;
  };
  auto x_advance = [&]() {
    const bool wrap = kx + 1 == nk;
    xbase = wrap ? xnext : xbase + P_BK * 2;
    xnum = wrap ? xnext_num : xnum;
    kx = wrap ? 0 : kx + 1;
  };

  f32x16 acc[4][4];  // [y tile][x tile], accumulator file
  // Zeroed on the matrix pipe: D = 0 * 0 + 0 writes the 16 accumulator registers of a tile without staging 16 zeros in arch
  // VGPRs first (the allocator materialised all 256 at once and spilled long-lived values around it); 16 MFMAs per tile.
  auto zero_acc = [&]() {
    // the zero fragment is PRODUCED by a volatile asm, per tile: a loop-invariant value would be kept across the MFMA loop, i.e. in
    // scratch, and its reload at the top of every tile drains the DMA stream (s_waitcnt vmcnt(0) in front of the first use)
    g256p_u4 zfrag;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zfrag[0]));
    asm volatile("v_mov_b32 %0, 0" : "=v"(zfrag[1]));
    asm volatile("v_mov_b32 %0, 0" : "=v"(zfrag[2]));
    asm volatile("v_mov_b32 %0, 0" : "=v"(zfrag[3]));
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %1, 0" : "=a"(acc[a][b]) : "v"(zfrag));
  };

  // ---- fragment addresses (as gemm256k.hip): k-step ks reads logical chunk 2ks + half ------------------------------------------
  const int sw = (l31 >> 1) & 7;
  const int yaddr0 = (wy * 128 + l31) * 128 + ((half ^ sw) << 4);
  const int xaddr0 = (wx * 128 + l31) * 128 + ((half ^ sw) << 4);
  auto load_frag = [&](G256pFrags& f, int sy, int sx, int ks, int r) {
    if (r < 4) f.y[r] = *(g256p_lds_u4*)(lds + (sy * P_UNIT + r * 4096) + (yaddr0 ^ (ks << 5)));
    else f.x[r - 4] = *(g256p_lds_u4*)(lds + (sx * P_UNIT + (r - 4) * 4096) + (xaddr0 ^ (ks << 5)));
  };

  // prologue (once per workgroup): stages 0 and 1 of the first tile (units 0..3)
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
  G256pFrags f0, f1;
#pragma unroll
  for (int r = 0; r < 8; ++r) load_frag(f0, 0, 1, 0, r);

  // Stage with global index g (counted over all tiles of this workgroup), J = g % 5: Y in slot 2J % 5, X in (2J+1) % 5;
  // the schedule of a stage is gemm256k.hip's.  `after_epi`: the first stage behind an epilogue -- 64 stores are younger than
  // the X pieces its sync point waits for.
#define P_SB() __builtin_amdgcn_sched_barrier(0)
#define P_KSTEP_A(FU, FL, SLOTD, PBASE, SY, SX, KS)  /* 4 Y pieces + 8 fragment reads */                         \
  _Pragma("unroll") for (int m = 0; m < 16; ++m) {                                                              \
    mfma256p(acc[m >> 2][m & 3], FU.y[m >> 2], FU.x[m & 3]); P_SB();                                             \
    if ((m & 3) == 0) y_piece(SLOTD, (PBASE) + (m >> 2));                                                        \
    else if (m - (m >> 2) - 1 < 8) load_frag(FL, SY, SX, KS, m - (m >> 2) - 1);                                  \
    P_SB();                                                                                                     \
  }
#define P_STEP(J)                                                                                               \
  if (gm == (J) && left > 0) {                                                                                  \
    constexpr int J_ = (J);                                                                                     \
    constexpr int SY = (2 * J_) % 5, SX = (2 * J_ + 1) % 5, NY = (2 * J_ + 2) % 5, NX = (2 * J_ + 3) % 5;        \
    constexpr int DY = (2 * J_ + 4) % 5, DX = (2 * J_) % 5;                                                      \
    P_KSTEP_A(f0, f1, DY, 0, SY, SX, 1)                                                                         \
    P_KSTEP_A(f1, f0, DY, 4, SY, SX, 2)                                                                         \
    y_advance();                                                                                                \
    _Pragma("unroll") for (int m = 0; m < 16; ++m) {                                                            \
      mfma256p(acc[m >> 2][m & 3], f0.y[m >> 2], f0.x[m & 3]); P_SB();                                           \
      if (m < 8) load_frag(f1, SY, SX, 3, m);                                                                    \
      P_SB();                                                                                                   \
    }                                                                                                           \
    if (after_epi) asm volatile("s_waitcnt vmcnt(63) lgkmcnt(0)" ::: "memory");                                 \
    else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");                                            \
    after_epi = false;                                                                                          \
    __builtin_amdgcn_s_barrier();                                                                               \
    asm volatile("" ::: "memory");                                                                              \
    _Pragma("unroll") for (int m = 0; m < 16; ++m) {                                                            \
      mfma256p(acc[m >> 2][m & 3], f1.y[m >> 2], f1.x[m & 3]); P_SB();                                           \
      if ((m & 1) == 0) x_piece(DX, m >> 1);                                                                     \
      else load_frag(f0, NY, NX, 0, m >> 1);                                                                     \
      P_SB();                                                                                                   \
    }                                                                                                           \
    x_advance();                                                                                                \
    if (!first_done) { P_STAMP_NEXT(5); P_STAMP(1); first_done = true; }                                       \
    --left;                                                                                                     \
    gm = (J) == 4 ? 0 : (J) + 1;                                                                                \
  }

  int gm = 0;            // global stage index mod 5
  bool after_epi = false;
  // epilogue geometry: lane (l31, half) owns columns 4 l31 .. 4 l31 + 3 of the wave's 128, and of the 32-row block yt the rows
  // 8 (r >> 2) + (r & 3) + 4 half
  for (;;) {
    P_STAMP(0);
    zero_acc();
    int left = nk;
    bool first_done = false;
    do {
      P_STEP(0)
      P_STEP(1)
      P_STEP(2)
      P_STEP(3)
      P_STEP(4)
    } while (left > 0);
    // ---- epilogue of tile `cur`: straight from the registers ---------------------------------------------------------------
    // Stores (and the residual loads of the gated form) go through a buffer descriptor over the tile's rows of Out: rows past
    // the matrix fall outside num_records and are dropped / read as zero by the hardware -- no per-row predicate, and every wave
    // issues exactly 64 stores per tile (what the vmcnt(63) of the next stage counts on).  Everything lane-dependent is derived
    // from an opaque lane id INSIDE this block: derived from threadIdx it is loop-invariant, hoisted in front of the tile loop
    // and kept in VGPRs across the MFMA loop (measured on the ISA: reloads from scratch in every stage).
    P_STAMP(2);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // last asm MFMAs -> accumulator reads
    {
      uint32_t lane_e;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
      const uint32_t l31e = lane_e & 31u, halfe = lane_e >> 5;
      const int64_t y0 = (int64_t)cur.ty * P_BM, x0 = (int64_t)cur.tx * P_BN;
      const uint32_t colb = (uint32_t)(wx * 128) * 2u + l31e * 8u;                      // byte offset of the lane's 4 columns in the tile row
      int64_t rows_valid = YM - y0;
      if (rows_valid > P_BM) rows_valid = P_BM;
      const uint32_t onum = (uint32_t)((rows_valid - 1) * ldo * 2 + P_BN * 2);          // rows >= rows_valid: out of range
      const uint32_t ldo2 = (uint32_t)(ldo * 2);
      const __amdgpu_buffer_rsrc_t odesc = __builtin_amdgcn_make_buffer_rsrc((void*)(Out + y0 * ldo + x0), 0, (int)onum, 0x00020000);
      const __amdgpu_buffer_rsrc_t rdesc =
          __builtin_amdgcn_make_buffer_rsrc((void*)((EPI == WAN_EPI_GATE_RES ? R : Out) + y0 * ldo + x0), 0, (int)onum, 0x00020000);
      const uint32_t lane_off = ((uint32_t)(wy * 128) + 4u * halfe) * ldo2 + colb;      // row wy*128 + 4 half of the tile
      float bcol[4] = {0.f, 0.f, 0.f, 0.f};
      if (bias != nullptr) {
        const g256p_u2 bw = *reinterpret_cast<const g256p_u2*>(reinterpret_cast<const char*>(bias + x0) + colb);
        bcol[0] = __uint_as_float(bw[0] << 16); bcol[1] = __uint_as_float(bw[0] & 0xffff0000u);
        bcol[2] = __uint_as_float(bw[1] << 16); bcol[3] = __uint_as_float(bw[1] & 0xffff0000u);
      }
      // gated residual: gate row = rnd16(mod[gate] + e[batch(row)][gate]) (model.py:658-660).  A 256-row tile touches at most
      // two batches (the launcher requires rows_per_batch >= 256: tokens per stream / per frame): both gate rows are fetched once.
      float gA[4] = {1.f, 1.f, 1.f, 1.f}, gB[4] = {1.f, 1.f, 1.f, 1.f};
      uint32_t rb = 0xffffffffu;  // first row (in the tile) of the tile's second batch
      const bool gated = EPI == WAN_EPI_GATE_RES && gate_idx >= 0;
      auto gate_row = [&](int64_t bidx, float* g) {
        const g256p_u2 mw = *reinterpret_cast<const g256p_u2*>(reinterpret_cast<const char*>(mod + (int64_t)gate_idx * XN + x0) + colb);
        const g256p_u2 ew = *reinterpret_cast<const g256p_u2*>(reinterpret_cast<const char*>(e + (bidx * n_mod + gate_idx) * XN + x0) + colb);
        g[0] = rbf(__uint_as_float(mw[0] << 16) + __uint_as_float(ew[0] << 16));
        g[1] = rbf(__uint_as_float(mw[0] & 0xffff0000u) + __uint_as_float(ew[0] & 0xffff0000u));
        g[2] = rbf(__uint_as_float(mw[1] << 16) + __uint_as_float(ew[1] << 16));
        g[3] = rbf(__uint_as_float(mw[1] & 0xffff0000u) + __uint_as_float(ew[1] & 0xffff0000u));
      };
      if (gated) {
        const int64_t b0 = y0 / rows_per_batch;
        const int64_t yb = (b0 + 1) * rows_per_batch;
        gate_row(b0, gA);
        if (yb < y0 + rows_valid) {
          rb = (uint32_t)(yb - y0);
          gate_row(b0 + 1, gB);
        }
      }
      const uint32_t row_lane = (uint32_t)(wy * 128) + 4u * halfe;
      // 16 chunks of 4 row pairs (one accumulator quarter of each of the block's four x tiles): the residual rows of chunk c + 1
      // are requested before chunk c is converted (latency under 16 accumulator reads + the pack), and no more than two chunks'
      // worth of registers is ever live beside the next tile's first fragments
      auto rload = [&](int c, int k) -> g256p_u2 {
        const int yt = c >> 2, r = (c & 3) * 4 + k;
        return __builtin_bit_cast(g256p_u2, __builtin_amdgcn_raw_buffer_load_b64(
                   rdesc, (int)(lane_off + (uint32_t)(yt * 32 + 8 * (r >> 2) + (r & 3)) * ldo2), 0, 0));
      };
      P_STAMP(3);
      g256p_u2 rq[4] = {}, rn[4] = {};
      if (EPI == WAN_EPI_GATE_RES) {
#pragma unroll
        for (int k = 0; k < 4; ++k) rq[k] = rload(0, k);
      }
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const int yt = c >> 2;
        if (EPI == WAN_EPI_GATE_RES && c + 1 < 16) {
#pragma unroll
          for (int k = 0; k < 4; ++k) rn[k] = rload(c + 1, k);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int r = (c & 3) * 4 + k;
          const uint32_t rit = (uint32_t)(yt * 32 + 8 * (r >> 2) + (r & 3));              // row in the wave's 128, before the half term
          float v[4] = {acc[yt][0][r] * out_scale + bcol[0], acc[yt][1][r] * out_scale + bcol[1], acc[yt][2][r] * out_scale + bcol[2],
                        acc[yt][3][r] * out_scale + bcol[3]};
          // nn.Linear output is a 16-bit tensor: GELU sees the rounded value; otherwise the pack below is that rounding
          if (EPI == WAN_EPI_GELU_TANH) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = g256p_gelu_tanh(rbf(v[j]));
          }
          if (EPI == WAN_EPI_GATE_RES) {
            const float rv[4] = {__uint_as_float(rq[k][0] << 16), __uint_as_float(rq[k][0] & 0xffff0000u),
                                 __uint_as_float(rq[k][1] << 16), __uint_as_float(rq[k][1] & 0xffff0000u)};
            if (gated) {
              const bool second = row_lane + rit >= rb;
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] = rv[j] + rbf(v[j]) * (second ? gB[j] : gA[j]);
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] = rv[j] + rbf(v[j]);
            }
          }
          g256p_u2 w;
          w[0] = pack2bf(v[0], v[1]);
          w[1] = pack2bf(v[2], v[3]);
          typedef unsigned int g256p_st2 __attribute__((__vector_size__(8)));
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(g256p_st2, w), odesc, (int)(lane_off + rit * ldo2), 0, 0);
          if (EPI == WAN_EPI_GELU_TANH) __builtin_amdgcn_sched_barrier(0);  // 4 GELUs' temporaries at a time
        }
        if (EPI == WAN_EPI_GATE_RES) {
#pragma unroll
          for (int k = 0; k < 4; ++k) rq[k] = rn[k];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    P_STAMP(4);
    if (!has_next) break;
    after_epi = true;  // every wave issued exactly 64 stores (range-checked, never skipped)
    cur = nxt;
    ++it;
    plan_next();
  }
#undef P_STEP
#undef P_KSTEP_A
#undef P_SB
#undef y_piece
#undef x_piece
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // trailing DMA of the dead stages must land before the LDS is released
}

}  // namespace

// Returns -1 when the problem does not fit this kernel (the caller falls back to gemm256k.hip and the generations before it),
// else the launch status.
static int g256p_cus() {  // CUs of the current device (one workgroup per CU: the kernel owns the whole LDS and register file)
  static int cached = 0;
  if (cached == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = -1;
    cached = n;
  }
  return cached;
}
template <int EPI>
int wan_gemm256p_try(const bf16_t* Y, int64_t ldy, int64_t YM, const bf16_t* X, int64_t ldx, int64_t XN, int K, bf16_t* Out,
                     int64_t ldo, const bf16_t* bias, const bf16_t* R, const bf16_t* mod, const bf16_t* e, int n_mod,
                     int gate_idx, int64_t rows_per_batch, hipStream_t st, float out_scale) {
  if (K % P_BK != 0 || K / P_BK < 3) return -1;
  if (XN % P_BN != 0 || ldo % 4 != 0) return -1;               // whole x tiles (every Wan width is a multiple of 256); 8-byte stores
  if ((((uintptr_t)Out | (uintptr_t)bias | (uintptr_t)R | (uintptr_t)mod | (uintptr_t)e) & 7) != 0) return -1;
  if (EPI == WAN_EPI_GATE_RES && gate_idx >= 0 && rows_per_batch < P_BM) return -1;   // at most two batches per tile
  if (255 * ldo * 2 + 512 >= ((int64_t)1 << 32)) return -1;    // 32-bit store offsets inside a tile
  // 32-bit DMA offsets: a tile's 256 rows times the row pitch in bytes, plus the row itself
  if (256 * ldy * 2 + (int64_t)K * 2 >= ((int64_t)1 << 32) || 256 * ldx * 2 + (int64_t)K * 2 >= ((int64_t)1 << 32)) return -1;
  const int64_t ty = (YM + P_BM - 1) / P_BM, tx = (XN + P_BN - 1) / P_BN;
  if (ty * tx >= ((int64_t)1 << 31)) return -1;
  const int group = 4;
  const int cus = g256p_cus();
  if (cus <= 0 || cus % 8 != 0 || ty * tx < cus) return -1;    // the XCD-contiguous order needs a grid that is a multiple of 8
  const int64_t grid = cus;
  hipLaunchKernelGGL((gemm256p_kernel<EPI>), dim3((unsigned)grid), dim3(256), 0, st, Y, ldy, YM, X, ldx, XN, K, Out, ldo, bias,
                     R, mod, e, n_mod, gate_idx, rows_per_batch, (int)ty, (int)tx, out_scale, group);
  WAN_LAUNCH_CHECK();
  return 0;
}

#ifdef G256P_TIMING
extern "C" int wan_gemm256p_stamps(uint64_t* out16) {
  WAN_CHECK_HIP(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g256p_stamps), sizeof(uint64_t) * 16));
  return 0;
}
#endif

#define G256P_INST(EPI)                                                                                                    \
  template int wan_gemm256p_try<EPI>(const bf16_t*, int64_t, int64_t, const bf16_t*, int64_t, int64_t, int, bf16_t*, int64_t, \
                                     const bf16_t*, const bf16_t*, const bf16_t*, const bf16_t*, int, int, int64_t, hipStream_t, float);
G256P_INST(WAN_EPI_NONE)
G256P_INST(WAN_EPI_GELU_TANH)
G256P_INST(WAN_EPI_GATE_RES)
#undef G256P_INST
