// EXPERIMENT FOR THE NEXT ROUND -- NOT VALIDATED ON HARDWARE, NOT IN THE DISPATCH, NOT IN libwanhip.so.
// gemm256m.hip's stage (256x256x64 on the 16x16x32 MFMA, register-direct 16-byte stores) inside gemm256p.hip's persistent tile walk
// (one workgroup per CU, the LDS-DMA stream continuous across tiles: no prologue, the stores of a tile drain under the next tile's
// first stage).  Out[y][x] = epilogue( sum_k Y[y][k] * X[x][k] + bias[x] ), bf16, bias per column, EPI NONE / GELU / GATE_RES.
//
// Why it exists: a K = 5120 tile of gemm256m spends 6k cycles in its prologue and 5-17k in its epilogue beside a ~176k-cycle loop
// (profiles/r03_gemm256m_stamps_run73.log); at K = 1536 (Wan 1.3B: 24 k-tiles per tile) the same 11-23k cycles stand beside ~53k.
// On the 32x32x16 kernel persistence bought 8 % of the cycles and no time (the chip was on the flat end of its voltage curve,
// DESIGN.md section 3.0); on the 16x16x32 kernel the clock has headroom again.  What this file settles WITHOUT a GPU: the merge
// fits the register file -- 240 / 256 / 256 VGPRs + 256 accumulators, NO scratch in any of the three instantiations -- and a
// steady-state stage is 128 MFMAs, 32 ds_read_b128, 16 LDS-DMA pieces, 4 s_waitcnt and no vector-ALU instruction
// (tests/test_isa_invariants.py); its layout is gemm256m's and its stream logic gemm256p's, both emulated on the CPU
// (tests/test_kernel_index_emulation.py) and both validated on hardware in their own kernels.  What is left for the first GPU
// call of the next round: `make -C wan2gp_amd/csrc persistent16` (libwanhip_mp.so), the GEMM suites through
// tools/_gpu/pytest_with_lib.py, tools/bench_gemm.py --lib libwanhip_mp.so at the 14B and 1.3B shapes.
// The stage behind an epilogue waits vmcnt(40): its 32 stores and the 8 Y pieces of its own first k-step are younger than the X
// pieces it needs.
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
  g256p_u4 y[8], x[8];
};
// D[i][j] += sum_k A[i][k] B[j][k]: A = the Y fragment (i = output row), B = the X fragment (j = output column): lane (j, half)
// holds column j of rows 8 (r >> 2) + (r & 3) + 4 half in register r.  Accumulators pinned to the accumulator file.
__device__ __forceinline__ void mfma256p(f32x4& acc, const g256p_u4& ya, const g256p_u4& xb) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(ya), "v"(xb));
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
__device__ uint64_t g256mp_stamps[16];  // tuning aid: s_memtime stamps of workgroup 40's third tile (tools/gemm_stamp.py --persistent)
#define P_STAMP(I) do { if (blockIdx.x == 40 && it == 2 && threadIdx.x == 0) g256mp_stamps[I] = __builtin_amdgcn_s_memtime(); } while (0)
#define P_STAMP_NEXT(I) do { if (blockIdx.x == 40 && it == 3 && threadIdx.x == 0) g256mp_stamps[I] = __builtin_amdgcn_s_memtime(); } while (0)
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
__global__ __launch_bounds__(256) void gemm256mp_kernel(const bf16_t* __restrict__ Y, int64_t ldy, int64_t YM,
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
  const int l15 = lane & 15, lg = lane >> 4;
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
    const int slab = row >> 7, t_ = (row >> 4) & 7, n_ = row & 15;
    uint32_t tx_ = (uint32_t)(slab * 128 + 8 * n_ + t_) * ldx2;
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

  f32x4 acc[8][8];  // [y tile][x tile], accumulator file
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
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 8; ++b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %1, 0" : "=a"(acc[a][b]) : "v"(zfrag));
  };

  const int sw = (l15 >> 1) & 7;
  uint32_t ybw[2][3], xbw[2][3];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int w = 0; w < 3; ++w) {
      ybw[ks][w] = (uint32_t)((((wy * 128 + l15) * 128 + ((lg ^ sw) << 4)) ^ (ks << 6)) + w * 65536);
      xbw[ks][w] = (uint32_t)((((wx * 128 + l15) * 128 + ((lg ^ sw) << 4)) ^ (ks << 6)) + w * 65536);
      asm volatile("" : "+v"(ybw[ks][w]), "+v"(xbw[ks][w]));
    }
  auto load_frag = [&](G256pFrags& f, int sy, int sx, int ks, int r) {
    if (r < 8) {
      const int u = sy * P_UNIT + r * 2048;
      f.y[r] = *(g256p_lds_u4*)(lds + ybw[ks][u >> 16] + (u & 0xffff));
    } else {
      const int u = sx * P_UNIT + (r - 8) * 2048;
      f.x[r - 8] = *(g256p_lds_u4*)(lds + xbw[ks][u >> 16] + (u & 0xffff));
    }
  };
#define M_ORD(I) ((I) < 7 ? 1 + (I) : (I) < 14 ? 2 + (I) : (I) == 14 ? 8 : 0)

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
  for (int r = 0; r < 16; ++r) load_frag(f0, 0, 1, 0, M_ORD(r));

  // Stage with global index g (counted over all tiles of this workgroup), J = g % 5: Y in slot 2J % 5, X in (2J+1) % 5;
  // the schedule of a stage is gemm256k.hip's.  `after_epi`: the first stage behind an epilogue -- 32 stores are younger than
  // the X pieces its sync point waits for.
#define P_SB() __builtin_amdgcn_sched_barrier(0)
#define P_STEP(J)                                                                                               \
  if (gm == (J) && left > 0) {                                                                                  \
    constexpr int J_ = (J);                                                                                     \
    constexpr int SY = (2 * J_) % 5, SX = (2 * J_ + 1) % 5, NY = (2 * J_ + 2) % 5, NX = (2 * J_ + 3) % 5;        \
    constexpr int DY = (2 * J_ + 4) % 5, DX = (2 * J_) % 5;                                                      \
    _Pragma("unroll") for (int m = 0; m < 64; ++m) {                                                            \
      mfma256p(acc[m >> 3][m & 7], f0.y[m >> 3], f0.x[m & 7]); P_SB();                                           \
      if ((m & 7) == 0) y_piece(DY, m >> 3);                                                                     \
      else if ((m & 1) == 1 && m < 32) load_frag(f1, SY, SX, 1, M_ORD(m >> 1));                                  \
      P_SB();                                                                                                   \
    }                                                                                                           \
    y_advance();                                                                                                \
    _Pragma("unroll") for (int m = 0; m < 16; ++m) {                                                            \
      mfma256p(acc[m >> 3][m & 7], f1.y[m >> 3], f1.x[m & 7]); P_SB();                                           \
    }                                                                                                           \
    if (after_epi) asm volatile("s_waitcnt vmcnt(40) lgkmcnt(0)" ::: "memory");                                 \
    else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");                                            \
    after_epi = false;                                                                                          \
    __builtin_amdgcn_s_barrier();                                                                               \
    asm volatile("" ::: "memory");                                                                              \
    _Pragma("unroll") for (int m = 16; m < 64; ++m) {                                                           \
      mfma256p(acc[m >> 3][m & 7], f1.y[m >> 3], f1.x[m & 7]); P_SB();                                           \
      if ((m - 16) % 6 == 0) x_piece(DX, (m - 16) / 6);                                                          \
      else if ((m & 1) == 1 && m < 48) load_frag(f0, NY, NX, 0, M_ORD((m - 17) >> 1));                           \
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
    // issues exactly 32 16-byte stores per tile (what the vmcnt(40) of the next stage counts on: 32 stores + its own 8 Y pieces; the
    // counter retires loads and stores in issue order on gfx9-class parts, as gemm256p.hip's validated runs rely on).  Everything lane-dependent is derived
    // from an opaque lane id INSIDE this block: derived from threadIdx it is loop-invariant, hoisted in front of the tile loop
    // and kept in VGPRs across the MFMA loop (measured on the ISA: reloads from scratch in every stage).
    P_STAMP(2);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // last asm MFMAs -> accumulator reads
    {
      uint32_t lane_e;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
      const uint32_t ne = lane_e & 15u, ge = lane_e >> 4;
      const int64_t y0 = (int64_t)cur.ty * P_BM, x0 = (int64_t)cur.tx * P_BN;
      const uint32_t colb = (uint32_t)(wx * 128) * 2u + ne * 16u;
      int64_t rows_valid = YM - y0;
      if (rows_valid > P_BM) rows_valid = P_BM;
      const uint32_t onum = (uint32_t)((rows_valid - 1) * ldo * 2 + P_BN * 2);
      const uint32_t ldo2 = (uint32_t)(ldo * 2);
      const __amdgpu_buffer_rsrc_t odesc = __builtin_amdgcn_make_buffer_rsrc((void*)(Out + y0 * ldo + x0), 0, (int)onum, 0x00020000);
      const __amdgpu_buffer_rsrc_t rdesc =
          __builtin_amdgcn_make_buffer_rsrc((void*)((EPI == WAN_EPI_GATE_RES ? R : Out) + y0 * ldo + x0), 0, (int)onum, 0x00020000);
      const uint32_t row_lane = (uint32_t)(wy * 128) + 4u * ge;
      const uint32_t lane_off = row_lane * ldo2 + colb;
      float bcol[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (bias != nullptr) unpack8(*reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(bias + x0) + colb), bcol);
      float gA[8], gB[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) gA[j] = gB[j] = 1.f;
      uint32_t rb = 0xffffffffu;
      const bool gated = EPI == WAN_EPI_GATE_RES && gate_idx >= 0;
      auto gate_row = [&](int64_t bidx, float* gq) {
        float mv[8], ev[8];
        unpack8(*reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(mod + (int64_t)gate_idx * XN + x0) + colb), mv);
        unpack8(*reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(e + (bidx * n_mod + gate_idx) * XN + x0) + colb), ev);
#pragma unroll
        for (int j = 0; j < 8; ++j) gq[j] = rbf(mv[j] + ev[j]);
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
      typedef unsigned int g256m_st4 __attribute__((__vector_size__(16)));
      auto rload = [&](int a, int i) -> uint4 {
        return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rdesc, (int)(lane_off + (uint32_t)(a * 16 + i) * ldo2), 0, 0));
      };
      P_STAMP(3);
      uint4 rq[2][4] = {};
      if (EPI == WAN_EPI_GATE_RES) {
#pragma unroll
        for (int i = 0; i < 4; ++i) rq[0][i] = rload(0, i);
      }
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        if (EPI == WAN_EPI_GATE_RES && a + 1 < 8) {
#pragma unroll
          for (int i = 0; i < 4; ++i) rq[(a + 1) & 1][i] = rload(a + 1, i);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint32_t rit = (uint32_t)(a * 16 + i);
          float v[8];
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            v[t] = acc[a][t][i] * out_scale + bcol[t];
            if (EPI == WAN_EPI_GELU_TANH) v[t] = g256p_gelu_tanh(rbf(v[t]));
          }
          if (EPI == WAN_EPI_GATE_RES) {
            float rv[8];
            unpack8(rq[a & 1][i], rv);
            if (gated) {
              const bool second = row_lane + rit >= rb;
#pragma unroll
              for (int t = 0; t < 8; ++t) v[t] = rv[t] + rbf(v[t]) * (second ? gB[t] : gA[t]);
            } else {
#pragma unroll
              for (int t = 0; t < 8; ++t) v[t] = rv[t] + rbf(v[t]);
            }
          }
          const uint4 w = pack8(v);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(g256m_st4, w), odesc, (int)(lane_off + rit * ldo2), 0, 0);
          if (EPI == WAN_EPI_GELU_TANH) __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    P_STAMP(4);
    if (!has_next) break;
    after_epi = true;  // every wave issued exactly 32 stores (range-checked, never skipped)
    cur = nxt;
    ++it;
    plan_next();
  }
#undef P_STEP
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
int wan_gemm256mp_try(const bf16_t* Y, int64_t ldy, int64_t YM, const bf16_t* X, int64_t ldx, int64_t XN, int K, bf16_t* Out,
                     int64_t ldo, const bf16_t* bias, const bf16_t* R, const bf16_t* mod, const bf16_t* e, int n_mod,
                     int gate_idx, int64_t rows_per_batch, hipStream_t st, float out_scale) {
  if (K % P_BK != 0 || K / P_BK < 3) return -1;
  if (XN % P_BN != 0 || ldo % 8 != 0) return -1;               // whole x tiles (every Wan width is a multiple of 256); 8-byte stores
  if ((((uintptr_t)Out | (uintptr_t)bias | (uintptr_t)R | (uintptr_t)mod | (uintptr_t)e) & 15) != 0) return -1;
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
  hipLaunchKernelGGL((gemm256mp_kernel<EPI>), dim3((unsigned)grid), dim3(256), 0, st, Y, ldy, YM, X, ldx, XN, K, Out, ldo, bias,
                     R, mod, e, n_mod, gate_idx, rows_per_batch, (int)ty, (int)tx, out_scale, group);
  WAN_LAUNCH_CHECK();
  return 0;
}

#ifdef G256P_TIMING
extern "C" int wan_gemm256mp_stamps(uint64_t* out16) {
  WAN_CHECK_HIP(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g256mp_stamps), sizeof(uint64_t) * 16));
  return 0;
}
#endif

#define G256P_INST(EPI)                                                                                                    \
  template int wan_gemm256mp_try<EPI>(const bf16_t*, int64_t, int64_t, const bf16_t*, int64_t, int64_t, int, bf16_t*, int64_t, \
                                     const bf16_t*, const bf16_t*, const bf16_t*, const bf16_t*, int, int, int64_t, hipStream_t, float);
G256P_INST(WAN_EPI_NONE)
G256P_INST(WAN_EPI_GELU_TANH)
G256P_INST(WAN_EPI_GATE_RES)
#undef G256P_INST
