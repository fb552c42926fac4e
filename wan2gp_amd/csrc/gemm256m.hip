// bf16 NT GEMM for gfx950, fifth generation: gemm256k.hip's 256x256x64 tile, ring and sync structure on the 16x16x32 MFMA.
// Out[y][x] = epilogue( sum_k Y[y][k] * X[x][k] + bias[x] ), same contract / epilogues as the other generations (bf16, bias per
// column, or per row with EPI NONE = the transposed V^T form; the fp16 forms stay with gemm256k.hip).
//
// Why (round 3, runs 68-71): these GEMMs run at the chip's power limit, where what a kernel is paid in is
// energy per FLOP, not cycles.  rocprofv3 counters on tools/probes/mfma_power_probe.hip: a loop of v_mfma_f32_32x32x16_bf16 fed by
// LDS fragment reads keeps the matrix pipe 92 % busy at 1.73 GHz; the same FLOPs as v_mfma_f32_16x16x32_bf16 keep it 90 % busy at
// 1.99 GHz -- the K = 32 form reads and writes each accumulator half as often per MAC (4 registers per 8,192 MACs against 16 per
// 16,384) and the chip clocks 15 % higher under it.  The vendor library's kernel for these shapes (MT256x256x64, MI16x16x1) is
// built on it and runs 7-13 % ahead of gemm256k in one process on the same tensors although it needs MORE cycles per tile
// (profiles/r03_gemm_vs_vendor_same_process_run68.json, r03_gemm_vendor_vs_gemm256k_pmc_*.json: 77 % busy at 1.72 GHz against
// 81 % at 1.60).
//
// What changes against gemm256k.hip (its header explains the tile, the k-tile of 64, the ring of five 32-KB units, the XOR swizzle):
//   * a wave's 128 x 128 is 8 x 8 tiles of 16 x 16, four accumulators per lane each (256 in the accumulator file, as before);
//     a k-tile of 64 is TWO k-steps of 32, 64 MFMAs of 16 cycles each; both k-steps' 16 fragments are register-resident
//     (2 x 64 VGPRs): k-step 1's are read during k-step 0, the next stage's k-step 0's after the sync point.
//   * fragment of tile r, lane (n = lane & 15, g = lane >> 4): row 16 r + n, logical 16-B chunk 4 ks + g.  The unit images, the
//     DMA plan and the swizzle p = chunk ^ ((row >> 1) & 7) are gemm256k's: the four 16-lane groups a ds_read_b128 is serviced in
//     still cover the 64 banks exactly once (tests/test_kernel_index_emulation.py).
//   * Y is the A operand, X the B operand: D[row = 4 g + i][col = n].  X rows are staged 8-way interleaved -- image row
//     (slab, t, n) holds column slab * 128 + 8 n + t of the tile -- so register i of a lane's eight x tiles t = 0..7 are EIGHT
//     CONSECUTIVE output columns of row 16 a + 4 g + i: the epilogue packs them to 16 bytes and stores straight from the
//     registers, a store instruction writing 4 rows x 256 contiguous bytes (32 per wave).  No LDS parking; rows past the matrix
//     are dropped by the buffer descriptor's range check.
//   * sync point P_S sits in front of MFMA 16 of k-step 1 (every fragment of stage S is in registers by the end of k-step 0);
//     the 48 MFMAs behind it issue X_{S+2} (8 pieces, every sixth gap) and the next stage's first fragments (the odd gaps up to
//     MFMA 47).  Per stage: 128 MFMAs, 32 ds_read_b128, 16 LDS-DMA pieces, three s_waitcnt, no vector-ALU instruction.
#include <stdlib.h>

#include "common.h"

namespace {

typedef __attribute__((address_space(3))) const char g256m_lds_cchar;
typedef uint32_t g256m_u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const g256m_u4 g256m_lds_u4;

constexpr int M_BN = 256, M_BK = 64;   // (the tile is 32 TY rows high: 256 for TY = 8)
// internal epilogue id (not in the public enum: reached through wan_gemm_bf16_res32): the mixed-precision plan's gated residual --
// Out is the FP32 residual stream, updated in place: x += bf16(acc + bias) * (mod[gate] + e0[batch][gate]) with an fp32 gate from a
// bf16 modulation row and an fp32 e0 row (mixed_ops.hip mx_gated_residual_kernel's arithmetic on the accumulators: one pass less
// over 6 bytes per element, three times per block)
constexpr int M_EPI_RES32 = 4;
constexpr int M_UNIT = 256 * M_BK * 2;  // 32 KiB: one operand of one stage (256 rows x 128 B)
constexpr int M_NU = 5;                  // ring of five units: unit u (Y_S = 2S, X_S = 2S+1) lives in slot u % 5

__device__ __forceinline__ float g256m_gelu_tanh(float x) {
  const float c = -2.0f * 0.7978845608028654f * 1.4426950408889634f;  // -2 sqrt(2/pi) log2(e)
  const float t = __builtin_fmaf(x * x, 0.044715f, 1.0f);
  const float ex = __builtin_amdgcn_exp2f(x * t * c);
  return x * __builtin_amdgcn_rcpf(1.0f + ex);
}

struct G256mFrags {
  g256m_u4 y[8], x[8];
};
// D[i][j] += sum_k A[i][k] B[k][j]: A = the Y fragment (lane (n, g): row n, k = 8 g ..), B = the X fragment (lane (n, g): column n,
// k = 8 g ..); lane (n, g) holds D[4 g + i][n] in register i.  Accumulators pinned to the accumulator file.
__device__ __forceinline__ void mfma256m(f32x4& acc, const g256m_u4& ya, const g256m_u4& xb) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(ya), "v"(xb));
}
// LDS-DMA piece as inline asm (invisible to hipcc's waitcnt pass, see gemm256k.hip); completion is counted by hand
__device__ __forceinline__ void g256m_dma16(uint32_t voff, const g256m_u4& rsrc, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rsrc), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ g256m_u4 g256m_rsrc(const void* base) {
  const uint64_t b = (uint64_t)base;
  g256m_u4 r;
  r[0] = (uint32_t)b;
  r[1] = (uint32_t)(b >> 32) & 0xffffu;  // stride 0
  r[2] = 0xffffffffu;                    // num_records: no range check needed (rows are clamped)
  r[3] = 0x00020000u;
  return r;
}

#ifdef G256M_TIMING
__device__ uint64_t g256m_stamps[16];  // tuning aid: s_memtime stamps of workgroup 40 (tools/gemm_stamp_m.py)
#define M_STAMP(I) do { if (blockIdx.x == 40 && threadIdx.x == 0) g256m_stamps[I] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define M_STAMP(I)
#endif

// the slots of k-step 1 behind the sync point (MFMAs 16 .. 8 TY - 1): the 8 X pieces every P-th gap, the next stage's TY + 8 first fragments in
// the gaps between them -- for TY = 8 (and 6: P even) the round-3 plan, pieces in even gaps and fragments in the odd ones
template <int TY> __device__ __forceinline__ constexpr int g256m_k1_p() { return (8 * TY - 16) / 8; }
template <int TY> __device__ __forceinline__ constexpr bool g256m_k1_piece(int m) { return (m - 16) % g256m_k1_p<TY>() == 0 && (m - 16) / g256m_k1_p<TY>() < 8; }
template <int TY> __device__ __forceinline__ constexpr int g256m_k1_load(int m) {   // index of the fragment read behind MFMA m, or -1
  if (g256m_k1_piece<TY>(m)) return -1;
  if (g256m_k1_p<TY>() % 2 == 0) return ((m & 1) == 1 && ((m - 17) >> 1) < TY + 8) ? ((m - 17) >> 1) : -1;
  int pieces = 0;
  for (int q = 16; q < m; ++q) pieces += g256m_k1_piece<TY>(q) ? 1 : 0;
  const int idx = (m - 16) - pieces;
  return idx < TY + 8 ? idx : -1;
}
// the order a k-step's TY + 8 fragments are read in: y 1 .. TY - 1, x 1 .. 7 (fragment indices 9 .. 15), then x 0 (8) and y 0 (0)
template <int TY> __device__ __forceinline__ constexpr int g256m_ord(int i) { return i < TY - 1 ? 1 + i : i < TY + 6 ? 9 + (i - (TY - 1)) : i == TY + 6 ? 8 : 0; }

// TY (round 6): y tiles of 16 rows per wave -- the tile is 32 TY rows high (8: the 256 x 256 tile of every many-tile problem; 5, 6, 7: 160,
// 192, 224 rows for problems of a few hundred tiles, where the height that leaves the fewest CUs idle beats the one with the most reuse:
// BASELINE configs[0]'s M = 6,400 rows are 25 x 6 tiles of 256 rows on 150 of 256 CUs, but 40 x 6 tiles of 160 rows on 240).  Units, ring,
// swizzle and sync structure do not change: a Y unit simply holds 32 TY of its 256 rows, a k-step is 8 TY MFMAs.
template <int EPI, bool BIAS_ROWS, int TY>
__global__ __launch_bounds__(256) void gemm256m_kernel(const bf16_t* __restrict__ Y, int64_t ldy, int64_t YM,
                                                       const bf16_t* __restrict__ X, int64_t ldx, int64_t XN, int K,
                                                       bf16_t* __restrict__ Out, int64_t ldo, const bf16_t* __restrict__ bias,
                                                       const bf16_t* __restrict__ R, const bf16_t* __restrict__ mod,
                                                       const bf16_t* __restrict__ e, int n_mod, int gate_idx,
                                                       int64_t rows_per_batch, int tiles_y, int tiles_x, float out_scale, int group) {
  __shared__ __attribute__((aligned(16))) char smem[M_NU * M_UNIT];  // 160 KiB
  g256m_lds_cchar* lds = (g256m_lds_cchar*)smem;
  M_STAMP(0);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wy = wave >> 1, wx = wave & 1;
  const int l15 = lane & 15, lg = lane >> 4;

  // ---- tile assignment: XCD-contiguous ids, then grouped ordering (gemm256k.hip) --------------------------------------------
  const int nwg = tiles_y * tiles_x;
  const int wg = xcd_remap(blockIdx.x, nwg);
  const int per_group = group * tiles_x;
  const int gidx = wg / per_group;
  const int first_y = gidx * group;
  const int gsz = min(tiles_y - first_y, group);
  const int in_g = wg - gidx * per_group;
  const int ty = first_y + (in_g % gsz);
  const int tx = in_g / gsz;
  constexpr int BMY = 32 * TY;   // rows of the tile (256 for TY = 8)
  constexpr int NM = 8 * TY;     // MFMAs of a k-step
  constexpr int NF = TY + 8;     // fragments of a k-step
  const int64_t y0 = (int64_t)ty * BMY;
  const int64_t x0 = (int64_t)tx * M_BN;

  // ---- DMA plan: loop-invariant per-lane byte offsets relative to the tile's first row ---------------------------------------
  // A unit image is 256 rows x 8 chunks of 16 B; piece i (0..7) of wave w fills 16-B slots q = i*256 + w*64 + lane, i.e. rows
  // i*32 + w*8 .. +8, eight lanes per row = the row's whole 128-B line in one instruction; physical chunk p of row r holds
  // logical chunk p ^ ((r >> 1) & 7).  Y rows in place; X image row (slab, t, n) = row slab*128 + 8 n + t of the tile's X panel.
  uint32_t yofs[TY], xofs[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int q = i * 256 + tid;
    const int row = q >> 3, pch = q & 7;
    const int lch = pch ^ ((row >> 1) & 7);
    if (i < TY) {
      int64_t yr = y0 + row;
      if (yr > YM - 1) yr = YM - 1;  // ragged tile: re-read the last row (its results are never stored)
      yofs[i] = (uint32_t)((yr - y0) * ldy * 2 + lch * 16);
    }
    const int slab = row >> 7, t = (row >> 4) & 7, n = row & 15;
    int64_t xr = x0 + slab * 128 + 8 * n + t;
    if (xr > XN - 1) xr = XN - 1;  // ragged x edge (the row-bias / V^T form: x = tokens): re-read the last row, its columns are never stored
    xofs[i] = (uint32_t)((xr - x0) * ldx * 2 + lch * 16);
  }
  const char* ybase = reinterpret_cast<const char*>(Y + y0 * ldy);  // next Y unit to fetch
  const char* xbase = reinterpret_cast<const char*>(X + x0 * ldx);  // next X unit to fetch
  const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;  // LDS byte address of the ring
  const int nk = K / M_BK;
  int ky = 0, kx = 0;  // stages the two streams point at; they stop at the last one (later fetches re-read it into a dead unit)
  auto y_piece = [&](int slot, int p) { g256m_dma16(yofs[p], g256m_rsrc(ybase), smem_lds + slot * M_UNIT + (p * 256 + wave * 64) * 16); };
  auto x_piece = [&](int slot, int p) { g256m_dma16(xofs[p], g256m_rsrc(xbase), smem_lds + slot * M_UNIT + (p * 256 + wave * 64) * 16); };
  auto y_advance = [&]() { const bool ok = ky + 1 < nk; ybase += ok ? M_BK * 2 : 0; ky += ok ? 1 : 0; };
  auto x_advance = [&]() { const bool ok = kx + 1 < nk; xbase += ok ? M_BK * 2 : 0; kx += ok ? 1 : 0; };

  // ---- fragment addresses: k-step ks (0, 1) reads logical chunk 4 ks + g; (row >> 1) & 7 == (n >> 1) & 7 for every tile ------
  // A ds_read carries a 16-bit immediate; the ring is 160 KB.  One base register per (operand, k-step, 64-KB window), opaque to
  // the compiler, and every fragment read is base + immediate: no address arithmetic inside the stages.
  const int sw = (l15 >> 1) & 7;
  uint32_t ybw[2][3], xbw[2][3];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int w = 0; w < 3; ++w) {
      ybw[ks][w] = (uint32_t)((((wy * (16 * TY) + l15) * 128 + ((lg ^ sw) << 4)) ^ (ks << 6)) + w * 65536);
      xbw[ks][w] = (uint32_t)((((wx * 128 + l15) * 128 + ((lg ^ sw) << 4)) ^ (ks << 6)) + w * 65536);
      asm volatile("" : "+v"(ybw[ks][w]), "+v"(xbw[ks][w]));
    }
  // fragment r = 0..15 (0..7: Y tiles, 8..15: X tiles) of k-step ks of the stage whose Y unit sits in slot sy (X unit in sx)
  auto load_frag = [&](G256mFrags& f, int sy, int sx, int ks, int r) {
    if (r < 8) {
      const int u = sy * M_UNIT + r * 2048;
      f.y[r] = *(g256m_lds_u4*)(lds + ybw[ks][u >> 16] + (u & 0xffff));
    } else {
      const int u = sx * M_UNIT + (r - 8) * 2048;
      f.x[r - 8] = *(g256m_lds_u4*)(lds + xbw[ks][u >> 16] + (u & 0xffff));
    }
  };

  // prologue: stages 0 and 1 (units 0..3); Y_2 is issued by stage 0's first k-step like every later Y unit
#pragma unroll
  for (int p = 0; p < TY; ++p) y_piece(0, p);
  y_advance();
#pragma unroll
  for (int p = 0; p < 8; ++p) x_piece(1, p);
  x_advance();
#pragma unroll
  for (int p = 0; p < TY; ++p) y_piece(2, p);
  y_advance();
#pragma unroll
  for (int p = 0; p < 8; ++p) x_piece(3, p);
  x_advance();
  f32x4 acc[TY][8];  // [y tile][x tile], accumulator file; zeroed while the first stages are in flight (256 writes: ~1k cycles)
#pragma unroll
  for (int a = 0; a < TY; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) {
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;
      asm volatile("" : "+a"(acc[a][b]));
    }

  // stage 0 landed, stage 1 (its TY + 8 pieces) may be in flight
  if (TY == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if (TY == 7) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
  else if (TY == 6) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  M_STAMP(1);
  G256mFrags f0, f1;
  // the order the 16 fragments of a k-step are read in: what its FIRST MFMA needs (y tile 0, x tile 0) LAST -- hipcc then puts one
  // lgkmcnt(0) in front of that MFMA and none behind it (in any other order it counts the reads down with a wait per fragment: 31
  // s_waitcnt per stage, each an issue slot of a 16-cycle gap); every read is issued >= 16 MFMAs before the k-step that uses it
#define M_ORD(I) g256m_ord<TY>(I)
#pragma unroll
  for (int r = 0; r < NF; ++r) load_frag(f0, 0, 1, 0, M_ORD(r));
  // Stage S = 5i + J: Y in slot 2J % 5, X in (2J+1) % 5; 2 k-steps of 64 MFMAs, MFMA m multiplies (y tile m>>3, x tile m&7).
  //   k-step 0 (f0): Y_{S+2} pieces 0..7 -> slot (2J+4) % 5 after MFMA 0,8,..,56; fragments of k-step 1 -> f1 after MFMA 1,3,..,31
  //   k-step 1 (f1): MFMAs 0..15 bare;
  //   P_S: vmcnt(8) (X_{S+1} and everything older landed; Y_{S+2} may fly), lgkmcnt(0) (stage S fully read) + barrier
  //                  MFMAs 16..63: X_{S+2} pieces 0..7 -> slot 2J % 5 (= Y_S, dead now) after MFMA 16,22,..,58; stage S+1's
  //                  k-step-0 fragments -> f0 after MFMA 17,19,..,47
#define M_SB() __builtin_amdgcn_sched_barrier(0)
#define M_STEP(J)                                                                                               \
  if (__builtin_expect(kt + (J) < nk, 1)) {                                                                     \
    constexpr int J_ = (J);                                                                                     \
    constexpr int SY = (2 * J_) % 5, SX = (2 * J_ + 1) % 5, NY = (2 * J_ + 2) % 5, NX = (2 * J_ + 3) % 5;        \
    constexpr int DY = (2 * J_ + 4) % 5, DX = (2 * J_) % 5;                                                      \
    _Pragma("unroll") for (int m = 0; m < NM; ++m) {                                                            \
      mfma256m(acc[m >> 3][m & 7], f0.y[m >> 3], f0.x[m & 7]); M_SB();                                           \
      if ((m & 7) == 0) y_piece(DY, m >> 3);                                                                     \
      else if ((m & 1) == 1 && m < 2 * NF) load_frag(f1, SY, SX, 1, M_ORD(m >> 1));                              \
      M_SB();                                                                                                   \
    }                                                                                                           \
    y_advance();                                                                                                \
    _Pragma("unroll") for (int m = 0; m < 16; ++m) {                                                            \
      mfma256m(acc[m >> 3][m & 7], f1.y[m >> 3], f1.x[m & 7]); M_SB();                                           \
    }                                                                                                           \
    if (TY == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");                                    \
    else if (TY == 7) asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory");                               \
    else if (TY == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");                               \
    else asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");                                            \
    __builtin_amdgcn_s_barrier();                                                                               \
    asm volatile("" ::: "memory");                                                                              \
    _Pragma("unroll") for (int m = 16; m < NM; ++m) {                                                           \
      mfma256m(acc[m >> 3][m & 7], f1.y[m >> 3], f1.x[m & 7]); M_SB();                                           \
      if (g256m_k1_piece<TY>(m)) x_piece(DX, (m - 16) / g256m_k1_p<TY>());                                       \
      else if (g256m_k1_load<TY>(m) >= 0) load_frag(f0, NY, NX, 0, M_ORD(g256m_k1_load<TY>(m)));                 \
      M_SB();                                                                                                   \
    }                                                                                                           \
    x_advance();                                                                                                \
  }
  for (int kt = 0; kt < nk; kt += 5) {
    M_STEP(0)
    M_STEP(1)
    M_STEP(2)
    M_STEP(3)
    M_STEP(4)
  }
#undef M_STEP
#undef M_SB
#undef M_ORD
  M_STAMP(2);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");  // trailing DMA (the LDS is released at the end); last asm MFMAs -> accumulator reads

  // ---- epilogue: straight from the registers ------------------------------------------------------------------------------------
  // Lane (n, g), y tile a, register i: row wy*128 + 16 a + 4 g + i, columns wx*128 + 8 n .. + 7 (one register of each of the eight
  // x tiles).  Stores (and the residual loads of the gated form) go through a buffer descriptor over the tile's rows of Out: rows
  // past the matrix fall outside num_records and are dropped / read as zero by the hardware -- no per-row predicate.
  if constexpr (EPI == M_EPI_RES32) {
    // the fp32 residual stream: a lane's 8 columns are 32 bytes of a row (two 16-byte accesses), read, updated and written back by
    // the same lane; the next chunk's rows are in flight while this one is converted
    float* X32 = reinterpret_cast<float*>(Out);
    const float* e32 = reinterpret_cast<const float*>(e);
    uint32_t lane_e;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    const uint32_t ne = lane_e & 15u, ge = lane_e >> 4;
    const uint32_t colb = (uint32_t)(wx * 128) * 2u + ne * 16u;   // bf16 vectors (bias, modulation)
    const uint32_t colf = (uint32_t)(wx * 128) * 4u + ne * 32u;   // fp32 rows (x, e0)
    int64_t rows_valid = YM - y0;
    if (rows_valid > BMY) rows_valid = BMY;
    const uint32_t onum = (uint32_t)((rows_valid - 1) * ldo * 4 + M_BN * 4);
    const uint32_t ldo4 = (uint32_t)(ldo * 4);
    const __amdgpu_buffer_rsrc_t xdesc = __builtin_amdgcn_make_buffer_rsrc((void*)(X32 + y0 * ldo + x0), 0, (int)onum, 0x00020000);
    const uint32_t row_lane = (uint32_t)(wy * (16 * TY)) + 4u * ge;
    const bool col_in = x0 + wx * 128 + 8 * (int64_t)ne + 8 <= XN;
    const uint32_t lane_off = col_in ? row_lane * ldo4 + colf : 0x80000000u;
    float bcol[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (bias != nullptr) unpack8(*reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(bias + x0) + colb), bcol);
    float gA[8], gB[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) gA[j] = gB[j] = 1.f;
    uint32_t rb = 0xffffffffu;
    const bool gated = gate_idx >= 0;
    auto gate_row = [&](int64_t bidx, float* gq) {
      float mv[8];
      unpack8(*reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(mod + (int64_t)gate_idx * XN + x0) + colb), mv);
      const float4* ep = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(e32 + (bidx * n_mod + gate_idx) * XN + x0) + colf);
      const float4 e0v = ep[0], e1v = ep[1];
      const float ev[8] = {e0v.x, e0v.y, e0v.z, e0v.w, e1v.x, e1v.y, e1v.z, e1v.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) gq[j] = mv[j] + ev[j];    // fp32 gate: bf16 modulation + fp32 e0, no rounding (model.py:658-660 in the mixed plan)
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
    auto xload = [&](int a, int i, int half) -> uint4 {
      return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xdesc, (int)(lane_off + (uint32_t)(a * 16 + i) * ldo4 + (uint32_t)half * 16u), 0, 0));
    };
    uint4 xq[2][4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) { xq[0][i][0] = xload(0, i, 0); xq[0][i][1] = xload(0, i, 1); }
#pragma unroll
    for (int a = 0; a < TY; ++a) {
      if (a + 1 < TY) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { xq[(a + 1) & 1][i][0] = xload(a + 1, i, 0); xq[(a + 1) & 1][i][1] = xload(a + 1, i, 1); }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t rit = (uint32_t)(a * 16 + i);
        const uint4 lo = xq[a & 1][i][0], hi = xq[a & 1][i][1];
        float xv[8] = {__uint_as_float(lo.x), __uint_as_float(lo.y), __uint_as_float(lo.z), __uint_as_float(lo.w),
                       __uint_as_float(hi.x), __uint_as_float(hi.y), __uint_as_float(hi.z), __uint_as_float(hi.w)};
        const bool second = row_lane + rit >= rb;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const float y = rbf(acc[a][t][i] * out_scale + bcol[t]);                       // the Linear's bf16 output
          xv[t] = gated ? __fadd_rn(xv[t], __fmul_rn(y, second ? gB[t] : gA[t])) : xv[t] + y;   // addcmul_: the product rounded first
        }
        uint4 wlo, whi;
        wlo.x = __float_as_uint(xv[0]); wlo.y = __float_as_uint(xv[1]); wlo.z = __float_as_uint(xv[2]); wlo.w = __float_as_uint(xv[3]);
        whi.x = __float_as_uint(xv[4]); whi.y = __float_as_uint(xv[5]); whi.z = __float_as_uint(xv[6]); whi.w = __float_as_uint(xv[7]);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(g256m_st4, wlo), xdesc, (int)(lane_off + rit * ldo4), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(g256m_st4, whi), xdesc, (int)(lane_off + rit * ldo4 + 16u), 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    uint32_t lane_e;  // opaque lane id: derived from threadIdx the epilogue's offsets are hoisted in front of the MFMA loop and spilled
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    const uint32_t ne = lane_e & 15u, ge = lane_e >> 4;
    const uint32_t colb = (uint32_t)(wx * 128) * 2u + ne * 16u;  // byte offset of the lane's 8 columns in the tile row
    int64_t rows_valid = YM - y0;
    if (rows_valid > BMY) rows_valid = BMY;
    const uint32_t onum = (uint32_t)((rows_valid - 1) * ldo * 2 + M_BN * 2);  // rows >= rows_valid: out of range
    const uint32_t ldo2 = (uint32_t)(ldo * 2);
    const __amdgpu_buffer_rsrc_t odesc = __builtin_amdgcn_make_buffer_rsrc((void*)(Out + y0 * ldo + x0), 0, (int)onum, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdesc =
        __builtin_amdgcn_make_buffer_rsrc((void*)((EPI == WAN_EPI_GATE_RES ? R : Out) + y0 * ldo + x0), 0, (int)onum, 0x00020000);
    const uint32_t row_lane = (uint32_t)(wy * (16 * TY)) + 4u * ge;
    // a lane's 8 columns are inside the matrix or outside as a whole (the launcher requires XN % 8 == 0); outside: an offset past
    // num_records, the stores are dropped like the rows past the matrix
    const bool col_in = x0 + wx * 128 + 8 * (int64_t)ne + 8 <= XN;
    const uint32_t lane_off = col_in ? row_lane * ldo2 + colb : 0x80000000u;
    float bcol[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (!BIAS_ROWS && bias != nullptr) unpack8(*reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(bias + x0) + colb), bcol);
    // gated residual: gate row = rnd16(mod[gate] + e[batch(row)][gate]) (model.py:658-660).  A 256-row tile touches at most two
    // batches (the launcher requires rows_per_batch >= 256: tokens per stream / per frame): both gate rows are fetched once.
    float gA[8], gB[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) gA[j] = gB[j] = 1.f;
    uint32_t rb = 0xffffffffu;  // first row (in the tile) of the tile's second batch
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
    // 8 chunks (y tiles) of 4 rows
    auto rload = [&](int a, int i) -> uint4 {
      return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rdesc, (int)(lane_off + (uint32_t)(a * 16 + i) * ldo2), 0, 0));
    };
    // residual rows of chunks a + 1 and a + 2 are in flight while chunk a is converted (the fragment registers are dead by now:
    // three chunks = 48 VGPRs); with one chunk ahead the gated epilogue was latency-bound (15-17k cycles against 5k plain)
    uint4 rq[3][4] = {};
    if (EPI == WAN_EPI_GATE_RES) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { rq[0][i] = rload(0, i); rq[1][i] = rload(1, i); }
    }
#pragma unroll
    for (int a = 0; a < TY; ++a) {
      if (EPI == WAN_EPI_GATE_RES && a + 2 < TY) {
#pragma unroll
        for (int i = 0; i < 4; ++i) rq[(a + 2) % 3][i] = rload(a + 2, i);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t rit = (uint32_t)(a * 16 + i);  // row in the wave's 128, before the lane-group term
        float v[8];
        float brow = 0.f;
        if (BIAS_ROWS && bias != nullptr) {
          int64_t yr = y0 + row_lane + rit;
          if (yr > YM - 1) yr = YM - 1;
          brow = bf2f(bias[yr]);
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          v[t] = acc[a][t][i] * out_scale + (BIAS_ROWS ? brow : bcol[t]);
          // nn.Linear output is a 16-bit tensor: GELU sees the rounded value; otherwise the pack below is that rounding
          if (EPI == WAN_EPI_GELU_TANH) v[t] = g256m_gelu_tanh(rbf(v[t]));
        }
        if (EPI == WAN_EPI_GATE_RES) {
          float rv[8];
          unpack8(rq[a % 3][i], rv);
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
        if (EPI == WAN_EPI_GELU_TANH) __builtin_amdgcn_sched_barrier(0);  // 8 GELUs' temporaries at a time
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  M_STAMP(3);
}

}  // namespace

// Returns -1 when the problem does not fit this kernel (the caller falls back to gemm256k.hip and the generations before it),
// else the launch status.  tile_rows: 256 (every many-tile problem), or 224 / 192 / 160 (round 6: problems of a few hundred tiles, the
// height chosen by wan_gemm256m_tile_rows below; not for the fp32-stream epilogue).
template <int EPI, bool BIAS_ROWS, int TY>
static int gemm256m_launch(const bf16_t* Y, int64_t ldy, int64_t YM, const bf16_t* X, int64_t ldx, int64_t XN, int K, bf16_t* Out,
                           int64_t ldo, const bf16_t* bias, const bf16_t* R, const bf16_t* mod, const bf16_t* e, int n_mod,
                           int gate_idx, int64_t rows_per_batch, hipStream_t st, float out_scale) {
  constexpr int BMY = 32 * TY;
  if (EPI == M_EPI_RES32 || EPI == WAN_EPI_GATE_RES) {
    if (gate_idx >= 0 && rows_per_batch < BMY) return -1;      // a tile touches at most two batches
  }
  const int64_t ty = (YM + BMY - 1) / BMY, tx = (XN + M_BN - 1) / M_BN;
  if (ty * tx >= ((int64_t)1 << 31)) return -1;
  // y tiles per group of the tile order (gemm256k.hip); WAN_GEMM_GROUP overrides the column-bias forms' 4 for A/B runs (round 5, run 07)
  static const int group_env = [] { const char* e = getenv("WAN_GEMM_GROUP"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 0; }();
  const int group = BIAS_ROWS ? 8 : (group_env ? group_env : 4);
  hipLaunchKernelGGL((gemm256m_kernel<EPI, BIAS_ROWS, TY>), dim3((unsigned)(ty * tx)), dim3(256), 0, st, Y, ldy, YM, X, ldx, XN, K, Out, ldo, bias, R,
                     mod, e, n_mod, gate_idx, rows_per_batch, (int)ty, (int)tx, out_scale, group);
  WAN_LAUNCH_CHECK();
  return 0;
}

template <int EPI, bool BIAS_ROWS>
int wan_gemm256m_try_h(const bf16_t* Y, int64_t ldy, int64_t YM, const bf16_t* X, int64_t ldx, int64_t XN, int K, bf16_t* Out,
                       int64_t ldo, const bf16_t* bias, const bf16_t* R, const bf16_t* mod, const bf16_t* e, int n_mod,
                       int gate_idx, int64_t rows_per_batch, hipStream_t st, float out_scale, int tile_rows) {
  if (K % M_BK != 0 || XN % 8 != 0) return -1;             // a lane stores 8 columns or none
  if (!BIAS_ROWS && XN % M_BN != 0) return -1;             // column bias / gate rows are fetched 16 bytes per lane without an edge form
  if (ldo % 8 != 0 || ((uintptr_t)Out & 15) != 0 || (!BIAS_ROWS && bias != nullptr && ((uintptr_t)bias & 15) != 0)) return -1;  // 16-byte stores / bias loads
  // 32-bit DMA offsets: a tile's 256 rows times the row pitch in bytes, plus the row itself; 31-bit store offsets
  if (256 * ldy * 2 + (int64_t)K * 2 >= ((int64_t)1 << 32) || 256 * ldx * 2 + (int64_t)K * 2 >= ((int64_t)1 << 32)) return -1;
  if (256 * ldo * (EPI == M_EPI_RES32 ? 4 : 2) + 1024 >= ((int64_t)1 << 31)) return -1;
  if (EPI == M_EPI_RES32) {  // Out = the fp32 stream, e = the fp32 e0 rows (both behind bf16-typed parameters), mod = bf16 rows
    if (ldo % 4 != 0 || XN % M_BN != 0) return -1;
    if (gate_idx >= 0 && (((uintptr_t)mod & 15) != 0 || ((uintptr_t)e & 15) != 0)) return -1;
  }
  if (EPI == WAN_EPI_GATE_RES) {
    if (((uintptr_t)R & 15) != 0) return -1;
    if (gate_idx >= 0 && (((uintptr_t)mod & 15) != 0 || ((uintptr_t)e & 15) != 0)) return -1;
  }
#define G256M_GO(T) return gemm256m_launch<EPI, BIAS_ROWS, T>(Y, ldy, YM, X, ldx, XN, K, Out, ldo, bias, R, mod, e, n_mod, gate_idx, rows_per_batch, st, out_scale)
  if constexpr (EPI != M_EPI_RES32) {
    if (tile_rows == 224) G256M_GO(7);
    if (tile_rows == 192) G256M_GO(6);
    if (tile_rows == 160) G256M_GO(5);
  }
  G256M_GO(8);
#undef G256M_GO
}
template <int EPI, bool BIAS_ROWS>
int wan_gemm256m_try(const bf16_t* Y, int64_t ldy, int64_t YM, const bf16_t* X, int64_t ldx, int64_t XN, int K, bf16_t* Out,
                    int64_t ldo, const bf16_t* bias, const bf16_t* R, const bf16_t* mod, const bf16_t* e, int n_mod,
                    int gate_idx, int64_t rows_per_batch, hipStream_t st, float out_scale) {
  return wan_gemm256m_try_h<EPI, BIAS_ROWS>(Y, ldy, YM, X, ldx, XN, K, Out, ldo, bias, R, mod, e, n_mod, gate_idx, rows_per_batch, st, out_scale, 256);
}

// The tile height for a problem of YM x XN outputs on `cus` compute units: the one of {256, 224, 192, 160} whose tiles need the least
// time -- rounds of tiles x height, with 2 % per step below 256 for the operand reuse a lower tile gives up -- and 256 unless the gain
// is at least 8 % (and always 256 beyond eight rounds, where the last round's quantisation is noise).
int wan_gemm256m_tile_rows(int64_t YM, int64_t XN, int cus) {
  if (cus < 1) cus = 256;
  const int64_t tx = (XN + M_BN - 1) / M_BN;
  auto cost = [&](int T) {
    const int64_t tiles = ((YM + 32 * T - 1) / (32 * T)) * tx;
    return (double)((tiles + cus - 1) / cus) * T * (1.0 + 0.02 * (8 - T));
  };
  const int64_t t8 = ((YM + 255) / 256) * tx;
  if (t8 > 8 * (int64_t)cus) return 256;
  int best = 8;
  double cb = cost(8);
  for (int T = 7; T >= 5; --T)
    if (cost(T) < cb * 0.92 && cost(T) < cost(best)) best = T;
  return 32 * best;
}

#define G256M_INST(EPI, BR)                                                                                                        \
  template int wan_gemm256m_try<EPI, BR>(const bf16_t*, int64_t, int64_t, const bf16_t*, int64_t, int64_t, int, bf16_t*, int64_t,  \
                                         const bf16_t*, const bf16_t*, const bf16_t*, const bf16_t*, int, int, int64_t, hipStream_t, float); \
  template int wan_gemm256m_try_h<EPI, BR>(const bf16_t*, int64_t, int64_t, const bf16_t*, int64_t, int64_t, int, bf16_t*, int64_t,  \
                                           const bf16_t*, const bf16_t*, const bf16_t*, const bf16_t*, int, int, int64_t, hipStream_t, float, int);
G256M_INST(WAN_EPI_NONE, false)
G256M_INST(WAN_EPI_GELU_TANH, false)
G256M_INST(WAN_EPI_GATE_RES, false)
G256M_INST(WAN_EPI_NONE, true)   // the transposed / V^T form: bias per output row, x = tokens (ragged)
G256M_INST(M_EPI_RES32, false)   // the mixed-precision plan's gated residual on the fp32 stream (wan_gemm_bf16_res32)
#undef G256M_INST

#ifdef G256M_TIMING
extern "C" int wan_gemm256m_stamps(uint64_t* out16) {
  WAN_CHECK_HIP(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g256m_stamps), sizeof(uint64_t) * 16));
  return 0;
}
#endif
