// Scaled-fp8 Linear for gfx950, second generation (round 4): gemm_fp8.hip's contract on gemm256m.hip's stage discipline and on the
// K = 128 form of the fp8 MFMA, v_mfma_f32_16x16x128_f8f6f4.
//   Out = epilogue( (A_fp8 W_fp8^T) * scale_a * scale_w + bias ), OCP e4m3fn operands, fp32 accumulate -- the arithmetic and the
//   rounding points of gemm_fp8.hip's epilogue (ScaledFP8WeightTensor._linear_scaled, shared/qtypes/scaled_fp8.py:324-380), which
//   stays the fallback for the shapes this kernel declines.
//
// Why: counters on gemm_fp8.hip (profiles/r03_gemm_fp8_pmc_sq_run88.json) -- 63.8 % matrix-pipe busy at 1.75 GHz where the bare fp8
// loop runs 98 % at 2.13: stalled (gemm256k's structure: LDS-parked epilogue, a counted wait per fragment) AND slow-clocked (the
// 32x32x64 form rewrites 16 accumulator registers per instruction).  One byte per element makes a k-tile of 128 elements exactly
// gemm256m's 128-byte rows: unit images, DMA plan, swizzle, ring of five 32-KB units and the register-direct epilogue carry over
// byte for byte.  What changes:
//   * ONE k-step per stage: 64 MFMAs of 16x16x128 (32 cycles each = the bf16 stage's 2,048 matrix cycles, twice the FLOP).  A lane's
//     operand is 32 bytes of its row: the two 16-byte LDS chunks gemm256m reads for k-steps 0 and 1 (logical chunks g and 4 + g)
//     side by side in one 8-register tuple.  Which k a byte means to the hardware is irrelevant as long as both operands are
//     loaded the same way (a sum over k).
//   * fragment residency without a second register set: the 64 MFMAs run as four 4 x 4 quadrants, (y 0-3, x 4-7), (y 0-3, x 0-3),
//     (y 4-7, x 4-7), (y 4-7, x 0-3).  Set P = {y 0-3, x 4-7} is all the first quadrant needs and is dead after the third; set
//     Q = {y 4-7, x 0-3} is first needed by the second.  Stage S's Q is read during its first quadrant, stage S + 1's P behind the
//     sync point (y 0-3 during quadrant 3, x 4-7 during quadrant 4): 128 fragment registers, 256 accumulators.
//   * sync point P_S after MFMA 31: vmcnt(8) (X_{S+1} and everything older landed; Y_{S+2} may fly), lgkmcnt(0) (stage S fully
//     read) + barrier.  Y_{S+2} pieces behind MFMA 0, 4, .., 28; X_{S+2} pieces (into Y_S's slot, dead now) behind 32, 36, .., 60.
#include <stdlib.h>

#include "common.h"

namespace {

typedef __attribute__((address_space(3))) const char f8m_lds_cchar;
typedef uint32_t f8m_u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const f8m_u4 f8m_lds_u4;

constexpr int M_BM = 256, M_BN = 256, M_BK = 128;
constexpr int M_UNIT = 256 * M_BK;  // 32 KiB: one operand of one stage (256 rows x 128 B, one byte per element)
constexpr int M_NU = 5;                  // ring of five units: unit u (Y_S = 2S, X_S = 2S+1) lives in slot u % 5

__device__ __forceinline__ float f8m_gelu_tanh(float x) {
  const float c = -2.0f * 0.7978845608028654f * 1.4426950408889634f;  // -2 sqrt(2/pi) log2(e)
  const float t = __builtin_fmaf(x * x, 0.044715f, 1.0f);
  const float ex = __builtin_amdgcn_exp2f(x * t * c);
  return x * __builtin_amdgcn_rcpf(1.0f + ex);
}

typedef uint32_t f8m_u8 __attribute__((ext_vector_type(8)));
struct F8mFrags {
  f8m_u8 y[8], x[8];   // [tile]: dwords 0..3 = logical chunk g of the row, 4..7 = chunk 4 + g -- ONE operand of the K = 128 MFMA
};
// D[i][j] += sum_k A[i][k] B[k][j] over 128 fp8 (e4m3, the instruction's default formats): A = the Y fragment (lane (n, g): row n), B = the X
// fragment (lane (n, g): column n); lane (n, g) holds D[4 g + i][n] in register i.  Accumulators pinned to the accumulator file.
__device__ __forceinline__ void mfma_f8m(f32x4& acc, const f8m_u8& ya, const f8m_u8& xb) {
  asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0" : "+a"(acc) : "v"(ya), "v"(xb));
}
// LDS-DMA piece as inline asm (invisible to hipcc's waitcnt pass, see gemm256k.hip); completion is counted by hand
__device__ __forceinline__ void f8m_dma16(uint32_t voff, const f8m_u4& rsrc, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rsrc), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ f8m_u4 f8m_rsrc(const void* base) {
  const uint64_t b = (uint64_t)base;
  f8m_u4 r;
  r[0] = (uint32_t)b;
  r[1] = (uint32_t)(b >> 32) & 0xffffu;  // stride 0
  r[2] = 0xffffffffu;                    // num_records: no range check needed (rows are clamped)
  r[3] = 0x00020000u;
  return r;
}

#ifdef F8M_TIMING
__device__ uint64_t f8m_stamps[16];  // tuning aid: s_memtime stamps of workgroup 40 (tools/gemm_stamp_m.py)
#define M_STAMP(I) do { if (blockIdx.x == 40 && threadIdx.x == 0) f8m_stamps[I] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define M_STAMP(I)
#endif

template <int EPI, bool BIAS_ROWS, bool SCALE_VEC>
__global__ __launch_bounds__(256) void gemm_fp8m_kernel(const uint8_t* __restrict__ Y, int64_t ldy, int64_t YM,
                                                       const uint8_t* __restrict__ X, int64_t ldx, int64_t XN, int K,
                                                       bf16_t* __restrict__ Out, int64_t ldo, const bf16_t* __restrict__ bias,
                                                       const float* __restrict__ scale_a, const float* __restrict__ scale_w,
                                                       const bf16_t* __restrict__ R, const bf16_t* __restrict__ mod,
                                                       const bf16_t* __restrict__ e, int n_mod, int gate_idx,
                                                       int64_t rows_per_batch, int tiles_y, int tiles_x, int group,
                                                       unsigned int* __restrict__ amax_out) {
  __shared__ __attribute__((aligned(16))) char smem[M_NU * M_UNIT];  // 160 KiB
  f8m_lds_cchar* lds = (f8m_lds_cchar*)smem;
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
  const int64_t y0 = (int64_t)ty * M_BM;
  const int64_t x0 = (int64_t)tx * M_BN;

  // ---- DMA plan: loop-invariant per-lane byte offsets relative to the tile's first row ---------------------------------------
  // A unit image is 256 rows x 8 chunks of 16 B; piece i (0..7) of wave w fills 16-B slots q = i*256 + w*64 + lane, i.e. rows
  // i*32 + w*8 .. +8, eight lanes per row = the row's whole 128-B line in one instruction; physical chunk p of row r holds
  // logical chunk p ^ ((r >> 1) & 7).  Y rows in place; X image row (slab, t, n) = row slab*128 + 8 n + t of the tile's X panel.
  uint32_t yofs[8], xofs[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int q = i * 256 + tid;
    const int row = q >> 3, pch = q & 7;
    const int lch = pch ^ ((row >> 1) & 7);
    int64_t yr = y0 + row;
    if (yr > YM - 1) yr = YM - 1;  // ragged tile: re-read the last row (its results are never stored)
    yofs[i] = (uint32_t)((yr - y0) * ldy + lch * 16);
    const int slab = row >> 7, t = (row >> 4) & 7, n = row & 15;
    int64_t xr = x0 + slab * 128 + 8 * n + t;
    if (xr > XN - 1) xr = XN - 1;  // ragged x edge (the row-bias / V^T form: x = tokens): re-read the last row, its columns are never stored
    xofs[i] = (uint32_t)((xr - x0) * ldx + lch * 16);
  }
  const char* ybase = reinterpret_cast<const char*>(Y + y0 * ldy);  // next Y unit to fetch
  const char* xbase = reinterpret_cast<const char*>(X + x0 * ldx);  // next X unit to fetch
  const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;  // LDS byte address of the ring
  const int nk = K / M_BK;
  int ky = 0, kx = 0;  // stages the two streams point at; they stop at the last one (later fetches re-read it into a dead unit)
  auto y_piece = [&](int slot, int p) { f8m_dma16(yofs[p], f8m_rsrc(ybase), smem_lds + slot * M_UNIT + (p * 256 + wave * 64) * 16); };
  auto x_piece = [&](int slot, int p) { f8m_dma16(xofs[p], f8m_rsrc(xbase), smem_lds + slot * M_UNIT + (p * 256 + wave * 64) * 16); };
  auto y_advance = [&]() { const bool ok = ky + 1 < nk; ybase += ok ? M_BK : 0; ky += ok ? 1 : 0; };
  auto x_advance = [&]() { const bool ok = kx + 1 < nk; xbase += ok ? M_BK : 0; kx += ok ? 1 : 0; };

  // ---- fragment addresses: k-step ks (0, 1) reads logical chunk 4 ks + g; (row >> 1) & 7 == (n >> 1) & 7 for every tile ------
  // A ds_read carries a 16-bit immediate; the ring is 160 KB.  One base register per (operand, k-step, 64-KB window), opaque to
  // the compiler, and every fragment read is base + immediate: no address arithmetic inside the stages.
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
  // half h (0, 1: logical chunk g / 4 + g of the row = gemm256m's k-steps) of fragment r = 0..15 (0..7: Y tiles, 8..15: X tiles) of the
  // stage whose Y unit sits in slot sy (X unit in sx): one ds_read_b128 into the lower / upper four dwords of the operand tuple
  auto load_half = [&](F8mFrags& f, int sy, int sx, int h, int r) {
    const bool isy = r < 8;
    const int u = (isy ? sy : sx) * M_UNIT + (r & 7) * 2048;
    const f8m_u4 v = *(f8m_lds_u4*)(lds + (isy ? ybw : xbw)[h][u >> 16] + (u & 0xffff));
    f8m_u8& d = isy ? f.y[r] : f.x[r - 8];
    d[4 * h + 0] = v[0]; d[4 * h + 1] = v[1]; d[4 * h + 2] = v[2]; d[4 * h + 3] = v[3];
  };
  auto load_frag = [&](F8mFrags& f, int sy, int sx, int r) { load_half(f, sy, sx, 0, r); load_half(f, sy, sx, 1, r); };

  // prologue: stages 0 and 1 (units 0..3); Y_2 is issued by stage 0's first k-step like every later Y unit
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
  f32x4 acc[8][8];  // [y tile][x tile], accumulator file; zeroed while the first stages are in flight (256 writes: ~1k cycles)
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) {
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;
      asm volatile("" : "+a"(acc[a][b]));
    }

  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // stage 0 landed, stage 1 may be in flight
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  M_STAMP(1);
  F8mFrags f;
  // set P of stage 0 (y tiles 0..3, x tiles 4..7): everything the first quadrant needs
#pragma unroll
  for (int r = 0; r < 4; ++r) { load_frag(f, 0, 1, r); load_frag(f, 0, 1, 12 + r); }
  // Stage S = 5i + J: Y in slot 2J % 5, X in (2J+1) % 5; 64 MFMAs in four quadrants of 16:
  //   m  0..15  (y m>>2, x 4 + (m&3))      set P only.  Y_{S+2} pieces 0..3 -> slot (2J+4) % 5 behind MFMA 0, 4, 8, 12; set Q of THIS stage
  //                                         (x 0..3 first, then y 4..7: 16 fragments halves = 16 reads, two per gap) behind 1, 2, 3, 5, 6, 7, 9, 10
  //   m 16..31  (y (m-16)>>2, x m&3)       Y_{S+2} pieces 4..7 behind 16, 20, 24, 28
  //   P_S: vmcnt(8) lgkmcnt(0) + barrier
  //   m 32..47  (y 4 + ((m-32)>>2), x 4 + (m&3))   X_{S+2} pieces 0..3 -> slot 2J % 5 behind 32, 36, 40, 44; stage S+1's y 0..3 behind 33, 34, 35, 37 (two halves each)
  //   m 48..63  (y 4 + ((m-48)>>2), x m&3)          X_{S+2} pieces 4..7 behind 48, 52, 56, 60; stage S+1's x 4..7 behind 49, 50, 51, 53
#define M_SB() __builtin_amdgcn_sched_barrier(0)
#define M_YT(m) ((((m) >> 5) << 2) + (((m) & 15) >> 2))
#define M_XT(m) (((((m) >> 4) & 1) ^ 1) * 4 + ((m) & 3))
#define M_STEP(J)                                                                                               \
  if (__builtin_expect(kt + (J) < nk, 1)) {                                                                     \
    constexpr int J_ = (J);                                                                                     \
    constexpr int SY = (2 * J_) % 5, SX = (2 * J_ + 1) % 5, NY = (2 * J_ + 2) % 5, NX = (2 * J_ + 3) % 5;        \
    constexpr int DY = (2 * J_ + 4) % 5, DX = (2 * J_) % 5;                                                      \
    _Pragma("unroll") for (int m = 0; m < 32; ++m) {                                                            \
      mfma_f8m(acc[M_YT(m)][M_XT(m)], f.y[M_YT(m)], f.x[M_XT(m)]); M_SB();                                       \
      if ((m & 3) == 0) y_piece(DY, m >> 2);                                                                     \
      else if (m < 11) {                                                                                        \
        const int q_ = (m - 1 - (m >> 2)) * 2;           /* 0, 2, .., 14: two half-fragment reads per gap */      \
        const int r0_ = q_ < 8 ? 8 + (q_ >> 1) : 4 + ((q_ - 8) >> 1);   /* x 0..3 (needed at MFMA 16), then y 4..7 (at 32) */ \
        load_frag(f, SY, SX, r0_);                                                                              \
      }                                                                                                         \
      M_SB();                                                                                                   \
    }                                                                                                           \
    y_advance();                                                                                                \
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");                                                 \
    __builtin_amdgcn_s_barrier();                                                                               \
    asm volatile("" ::: "memory");                                                                              \
    _Pragma("unroll") for (int m = 32; m < 64; ++m) {                                                           \
      mfma_f8m(acc[M_YT(m)][M_XT(m)], f.y[M_YT(m)], f.x[M_XT(m)]); M_SB();                                       \
      if ((m & 3) == 0) x_piece(DX, (m - 32) >> 2);                                                              \
      else if (m >= 33 && m <= 37) load_frag(f, NY, NX, m - 33 - ((m - 32) >> 2));          /* y 0..3: dead since MFMA 31 */ \
      else if (m >= 49 && m <= 53) load_frag(f, NY, NX, 12 + (m - 49 - ((m - 48) >> 2)));   /* x 4..7: dead since MFMA 47 */ \
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
#undef M_YT
#undef M_XT
  M_STAMP(2);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");  // trailing DMA (the LDS is released at the end); last asm MFMAs -> accumulator reads

  // ---- epilogue: straight from the registers ------------------------------------------------------------------------------------
  // Lane (n, g), y tile a, register i: row wy*128 + 16 a + 4 g + i, columns wx*128 + 8 n .. + 7 (one register of each of the eight
  // x tiles).  Stores (and the residual loads of the gated form) go through a buffer descriptor over the tile's rows of Out: rows
  // past the matrix fall outside num_records and are dropped / read as zero by the hardware -- no per-row predicate.
  {
    uint32_t lane_e;  // opaque lane id: derived from threadIdx the epilogue's offsets are hoisted in front of the MFMA loop and spilled
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    const uint32_t ne = lane_e & 15u, ge = lane_e >> 4;
    const uint32_t colb = (uint32_t)(wx * 128) * 2u + ne * 16u;  // byte offset of the lane's 8 columns in the tile row
    int64_t rows_valid = YM - y0;
    if (rows_valid > M_BM) rows_valid = M_BM;
    const uint32_t onum = (uint32_t)((rows_valid - 1) * ldo * 2 + M_BN * 2);  // rows >= rows_valid: out of range
    const uint32_t ldo2 = (uint32_t)(ldo * 2);
    const __amdgpu_buffer_rsrc_t odesc = __builtin_amdgcn_make_buffer_rsrc((void*)(Out + y0 * ldo + x0), 0, (int)onum, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdesc =
        __builtin_amdgcn_make_buffer_rsrc((void*)((EPI == WAN_EPI_GATE_RES ? R : Out) + y0 * ldo + x0), 0, (int)onum, 0x00020000);
    const uint32_t row_lane = (uint32_t)(wy * 128) + 4u * ge;
    // a lane's 8 columns are inside the matrix or outside as a whole (the launcher requires XN % 8 == 0); outside: an offset past
    // num_records, the stores are dropped like the rows past the matrix
    const bool col_in = x0 + wx * 128 + 8 * (int64_t)ne + 8 <= XN;
    const uint32_t lane_off = col_in ? row_lane * ldo2 + colb : 0x80000000u;
    float bcol[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (!BIAS_ROWS && bias != nullptr) unpack8(*reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(bias + x0) + colb), bcol);
    // the scales (gemm_fp8.hip's epilogue, scaled_fp8.py:324-380): per-tensor weight scale -> (acc * scale_a) * scale_b + bias, one
    // rounding; per-output-row weight scale -> bf16(acc * scale_a), then *= bf16(scale_w[n]), then += bias, one rounding each
    const float sa = scale_a[0];
    const float sb = SCALE_VEC ? 1.0f : scale_w[0];
    float scol[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
    if (SCALE_VEC && !BIAS_ROWS && col_in) {
      const float4 s0 = *reinterpret_cast<const float4*>(scale_w + x0 + wx * 128 + 8 * (int64_t)ne);
      const float4 s1 = *reinterpret_cast<const float4*>(scale_w + x0 + wx * 128 + 8 * (int64_t)ne + 4);
      scol[0] = rbf(s0.x); scol[1] = rbf(s0.y); scol[2] = rbf(s0.z); scol[3] = rbf(s0.w);
      scol[4] = rbf(s1.x); scol[5] = rbf(s1.y); scol[6] = rbf(s1.z); scol[7] = rbf(s1.w);
    }
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
    typedef unsigned int f8m_st4 __attribute__((__vector_size__(16)));
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
    // round 5 (GELU form, amax_out != NULL): |max| of what this tile stores, as float bits of the bf16 values -- the abs-max pass of the
    // NEXT Linear's activation quantisation (ffn.2 reads this tensor) folded into its producer; rows of a ragged last tile are re-reads of
    // the last valid row, so they cannot raise the maximum
    uint32_t amax = 0;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      if (EPI == WAN_EPI_GATE_RES && a + 2 < 8) {
#pragma unroll
        for (int i = 0; i < 4; ++i) rq[(a + 2) % 3][i] = rload(a + 2, i);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t rit = (uint32_t)(a * 16 + i);  // row in the wave's 128, before the lane-group term
        float v[8];
        float brow = 0.f, srow = 1.f;
        if (BIAS_ROWS) {
          int64_t yr = y0 + row_lane + rit;
          if (yr > YM - 1) yr = YM - 1;
          if (bias != nullptr) brow = bf2f(bias[yr]);
          if (SCALE_VEC) srow = rbf(scale_w[yr]);
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const float bj = BIAS_ROWS ? brow : bcol[t];
          float o = acc[a][t][i] * sa * sb;
          if (SCALE_VEC) {
            o = rbf(o);                                          // _scaled_mm output (scale_b = 1, no bias), bf16
            o = rbf(o * (BIAS_ROWS ? srow : scol[t]));           // out *= output_scale
            if (bias != nullptr) o = rbf(o + bj);                // out += bias
          } else {
            o = o + bj;                                          // bias inside _scaled_mm, one rounding (the pack below / rbf)
          }
          // nn.Linear output is a 16-bit tensor: GELU sees the rounded value; otherwise the pack below is that rounding
          if (EPI == WAN_EPI_GELU_TANH) o = f8m_gelu_tanh(rbf(o));
          v[t] = o;
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
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(f8m_st4, w), odesc, (int)(lane_off + rit * ldo2), 0, 0);
        if (EPI == WAN_EPI_GELU_TANH) {
          const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) amax = max(amax, max((ww[j] << 16) & 0x7fffffffu, ww[j] & 0x7fff0000u));
          __builtin_amdgcn_sched_barrier(0);  // 8 GELUs' temporaries at a time
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (EPI == WAN_EPI_GELU_TANH && amax_out != nullptr) {
      if (!col_in) amax = 0;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) amax = max(amax, (uint32_t)__shfl_xor((int)amax, o, 64));
      if (lane_e == 0 && amax != 0 && amax > __atomic_load_n(amax_out, __ATOMIC_RELAXED)) atomicMax(amax_out, amax);   // (read first: few tiles raise the running maximum)
    }
  }
  M_STAMP(3);
}

}  // namespace

// Returns -1 when the problem does not fit this kernel (the caller keeps gemm_fp8.hip's kernel), else the launch status.
template <int EPI, bool BIAS_ROWS>
int wan_gemm_fp8m_try(const uint8_t* Y, int64_t ldy, int64_t YM, const uint8_t* X, int64_t ldx, int64_t XN, int K, bf16_t* Out,
                      int64_t ldo, const bf16_t* bias, const float* scale_a, const float* scale_w, bool scale_vec, const bf16_t* R,
                      const bf16_t* mod, const bf16_t* e, int n_mod, int gate_idx, int64_t rows_per_batch, hipStream_t st, unsigned int* amax_out) {
  if (K % M_BK != 0 || XN % 8 != 0) return -1;             // a lane stores 8 columns or none
  if (!BIAS_ROWS && XN % M_BN != 0) return -1;             // column bias / scale / gate rows are fetched per lane without an edge form
  if (ldo % 8 != 0 || ((uintptr_t)Out & 15) != 0 || (!BIAS_ROWS && bias != nullptr && ((uintptr_t)bias & 15) != 0)) return -1;
  if (!BIAS_ROWS && scale_vec && ((uintptr_t)scale_w & 15) != 0) return -1;   // 16-byte scale loads
  if (ldy % 16 != 0 || ldx % 16 != 0) return -1;
  if (256 * ldy + (int64_t)K >= ((int64_t)1 << 32) || 256 * ldx + (int64_t)K >= ((int64_t)1 << 32)) return -1;  // 32-bit DMA offsets
  if (256 * ldo * 2 + 512 >= ((int64_t)1 << 31)) return -1;
  if (EPI == WAN_EPI_GATE_RES) {
    if (((uintptr_t)R & 15) != 0) return -1;
    if (gate_idx >= 0 && (rows_per_batch < M_BM || ((uintptr_t)mod & 15) != 0 || ((uintptr_t)e & 15) != 0)) return -1;
  }
  const int64_t ty = (YM + M_BM - 1) / M_BM, tx = (XN + M_BN - 1) / M_BN;
  if (ty * tx < 256 || ty * tx >= ((int64_t)1 << 31)) return -1;   // fewer tiles than CUs: the older kernel's shapes
  const int group = BIAS_ROWS ? 8 : 4;  // y tiles per group of the tile order (gemm256k.hip)
  if (scale_vec)
    hipLaunchKernelGGL((gemm_fp8m_kernel<EPI, BIAS_ROWS, true>), dim3((unsigned)(ty * tx)), dim3(256), 0, st, Y, ldy, YM, X, ldx, XN, K, Out, ldo,
                       bias, scale_a, scale_w, R, mod, e, n_mod, gate_idx, rows_per_batch, (int)ty, (int)tx, group, amax_out);
  else
    hipLaunchKernelGGL((gemm_fp8m_kernel<EPI, BIAS_ROWS, false>), dim3((unsigned)(ty * tx)), dim3(256), 0, st, Y, ldy, YM, X, ldx, XN, K, Out, ldo,
                       bias, scale_a, scale_w, R, mod, e, n_mod, gate_idx, rows_per_batch, (int)ty, (int)tx, group, amax_out);
  WAN_LAUNCH_CHECK();
  return 0;
}

#define F8M_INST(EPI, BR)                                                                                                            \
  template int wan_gemm_fp8m_try<EPI, BR>(const uint8_t*, int64_t, int64_t, const uint8_t*, int64_t, int64_t, int, bf16_t*, int64_t, \
                                          const bf16_t*, const float*, const float*, bool, const bf16_t*, const bf16_t*, const bf16_t*, \
                                          int, int, int64_t, hipStream_t, unsigned int*);
F8M_INST(WAN_EPI_NONE, false)
F8M_INST(WAN_EPI_GELU_TANH, false)
F8M_INST(WAN_EPI_GATE_RES, false)
F8M_INST(WAN_EPI_NONE, true)   // the transposed / V^T form: bias and the per-row scale run along output rows, x = tokens (ragged)
#undef F8M_INST
