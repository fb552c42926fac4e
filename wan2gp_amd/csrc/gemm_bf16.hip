// bf16 NT GEMM on MFMA for gfx950:  Out[y][x] = epilogue( sum_k Y[y][k] * X[x][k] + bias )
//
// Both operands are K-contiguous ("NT"), so every MFMA fragment is one 16-byte LDS read.
//   normal     : Y = activations A [M,K], X = weights W [N,K]      -> C[M,N]
//   transposed : Y = weights W [N,K],     X = activations A [M,K]  -> Ct[N,M]  (emits V^T)
//
// This file: the small-problem kernel (128(y) x 128(x) x 64(k) tiles, 4 waves in 2x2, each wave 64x64 = 4x4 MFMA
// 16x16x32 tiles, two workgroups per CU, 2-deep LDS ring) and the launcher that picks a kernel by problem size:
//   >= 256 tiles of 256x256  -> gemm256k.hip (every Wan projection at 480p and above)
//   M >= 512 and N >= 128    -> gemm32.hip   (256x128x32 tiles)
//   anything else            -> this kernel  (time / text MLPs, toy shapes)
// (A 256-row, 8-wave variant of this kernel -- one workgroup per CU -- measured 20 % slower than two 4-wave workgroups, and the
// BK = 32 256x256 kernel gemm256.hip 13 % slower than gemm256k; both were removed in round 2.)
// Global->LDS staging uses the LDS-DMA path (global_load_lds_dwordx4, 16 B per lane, 1 KiB per wave-instruction).
//
// LDS image: rows of 64 bf16 (128 B = 8 chunks of 16 B).  LDS-DMA writes lane-linearly, so the
// bank-conflict swizzle is applied to the per-lane SOURCE address (cdna guide rule 21):
//   physical chunk p of row r holds logical chunk p ^ ((r>>1)&7); readers apply the same XOR.
// With 128-B rows two rows share one 256-B bank row, and ((r>>1)&7) makes the 16 rows of a
// ds_read_b128 lane group land on 16 distinct 16-B slots (conflict-free, derivation in DESIGN.md).
//
// The X operand is the FIRST MFMA operand, so a lane's accumulator registers run along x.  X rows
// are staged in a permuted order (LDS row nt*16+i <- x row (i>>2)*16 + nt*4 + (i&3) of the wave's
// 64) so that the 16 accumulators a lane holds for one y row are 16 CONSECUTIVE x: the epilogue
// reads bias/residual/gate and writes the output with 2 x 16-byte accesses per lane per row
// (128 B contiguous per y row across the 4 lane groups) with no LDS transpose.
#include <stdlib.h>
#include <string.h>

#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 mfma_f16x8;

template <bool F16>
__device__ __forceinline__ f32x4 mfma16(const uint4& a, const uint4& b, f32x4 c) {
  if (F16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(mfma_f16x8, a), __builtin_bit_cast(mfma_f16x8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, a), __builtin_bit_cast(mfma_bf16x8, b), c, 0, 0, 0);
}

#define BN 128
#define BK 64
#define XSTAGE_BYTES (BN * BK * 2)  // 16 KiB: X operand per stage

__device__ __forceinline__ void glds16(const void* gsrc, void* ldst) { wan_lds_dma16(gsrc, ldst); }

// torch GELU(approximate='tanh'): 0.5 x (1 + tanh(u)), u = sqrt(2/pi) (x + 0.044715 x^3)  ==  x / (1 + exp(-2u))
__device__ __forceinline__ float gelu_tanh_f(float x) {
  const float c = -2.0f * 0.7978845608028654f * 1.4426950408889634f;  // -2 sqrt(2/pi) log2(e)
  const float t = __builtin_fmaf(x * x, 0.044715f, 1.0f);
  const float ex = __builtin_amdgcn_exp2f(x * t * c);
  return x * __builtin_amdgcn_rcpf(1.0f + ex);
}

// F16 = false: bf16 storage (DiT);  F16 = true: fp16 storage (VAE) -- same tiles, MFMA f16 variant.
// out_scale multiplies the fp32 accumulator before the bias (used for QK^T / sqrt(C) in the VAE).
template <int EPI, bool BIAS_ROWS, bool F16>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(
    const bf16_t* __restrict__ Y, int64_t ldy, int64_t YM, const bf16_t* __restrict__ X, int64_t ldx,
    int64_t XN, int K, bf16_t* __restrict__ Out, int64_t ldo, const bf16_t* __restrict__ bias,
    const bf16_t* __restrict__ R, const bf16_t* __restrict__ mod, const bf16_t* __restrict__ e, int n_mod,
    int gate_idx, int64_t rows_per_batch, int tiles_y, int tiles_x, float out_scale) {
  constexpr int BM = 128;                       // y rows per workgroup tile (4 waves in 2 x 2, 64 x 64 each)
  constexpr int NTHR = 256;
  constexpr int NSTG = 2;                       // LDS ring depth
  constexpr int YSTAGE_BYTES = BM * BK * 2;
  constexpr int STAGE_ALL = YSTAGE_BYTES + XSTAGE_BYTES;
  constexpr int XS = 1024 / NTHR;               // X staging slots per thread (Y: always 4)
  __shared__ __attribute__((aligned(16))) char smem[NSTG * STAGE_ALL];  // [stage][Y|X]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wy = wave >> 1, wx = wave & 1;

  // ---- tile assignment: XCD-contiguous ids, then grouped (8 y-tiles per group) ordering -------
  const int nwg = tiles_y * tiles_x;
  const int wg = xcd_remap(blockIdx.x, nwg);
  const int GROUP = 8;
  const int per_group = GROUP * tiles_x;
  const int gidx = wg / per_group;
  const int first_y = gidx * GROUP;
  const int gsz = min(tiles_y - first_y, GROUP);
  const int in_g = wg - gidx * per_group;
  const int ty = first_y + (in_g % gsz);
  const int tx = in_g / gsz;
  const int64_t y0 = (int64_t)ty * BM;
  const int64_t x0 = (int64_t)tx * BN;

  // ---- per-thread staging addresses (4 x 16 B per operand per stage) ---------------------------
  const bf16_t* ysrc[4];
  const bf16_t* xsrc[XS];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = i * NTHR + tid;  // linear 16-B slot in the Y image
    const int row = q >> 3, pch = q & 7;
    const int lch = pch ^ ((row >> 1) & 7);
    int64_t yr = y0 + row;
    if (yr > YM - 1) yr = YM - 1;
    ysrc[i] = Y + yr * ldy + lch * 8;
  }
#pragma unroll
  for (int i = 0; i < XS; ++i) {
    const int q = i * NTHR + tid;  // linear 16-B slot in the 16 KiB X image
    const int row = q >> 3, pch = q & 7;
    const int lch = pch ^ ((row >> 1) & 7);
    // X image row -> permuted x row inside the owning wave's 64-row slab
    const int slab = row >> 6, jj = row & 63;
    const int nt = jj >> 4, ii = jj & 15;
    int64_t xr = x0 + slab * 64 + (ii >> 2) * 16 + nt * 4 + (ii & 3);
    if (xr > XN - 1) xr = XN - 1;
    xsrc[i] = X + xr * ldx + lch * 8;
  }

  auto stage = [&](int s, int k0) {
    char* ybase = smem + s * STAGE_ALL;
    char* xbase = ybase + YSTAGE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(ysrc[i] + k0, ybase + (i * NTHR + wave * 64) * 16);  // wave-uniform LDS base; lanes land at +lane*16
#pragma unroll
    for (int i = 0; i < XS; ++i) glds16(xsrc[i] + k0, xbase + (i * NTHR + wave * 64) * 16);
  };

  f32x4 acc[4][4];  // [yt][xt]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment read offsets (bytes) inside a stage image, for ksub = 0; ksub = 1 flips chunk bit 2
  const int frow = lane & 15, fch = lane >> 4;
  int yoff[4], xoff[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int ry = wy * 64 + t * 16 + frow;
    yoff[t] = ry * 128 + ((fch ^ ((ry >> 1) & 7)) << 4);
    const int rx = wx * 64 + t * 16 + frow;
    xoff[t] = rx * 128 + ((fch ^ ((rx >> 1) & 7)) << 4);
  }

  const int nk = K / BK;
  auto compute = [&](const char* ybase, const char* xbase) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint4 yf[4], xf[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        yf[t] = *reinterpret_cast<const uint4*>(ybase + (yoff[t] ^ (ks << 6)));
        xf[t] = *reinterpret_cast<const uint4*>(xbase + (xoff[t] ^ (ks << 6)));
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = mfma16<F16>(xf[b], yf[a], acc[a][b]);
    }
  };
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) stage((kt + 1) & 1, (kt + 1) * BK);
    const char* ybase = smem + (kt & 1) * STAGE_ALL;
    compute(ybase, ybase + YSTAGE_BYTES);
  }

  // ---- epilogue: lane holds, for y row (yt, lane&15), 16 consecutive x starting at xb ---------
  const int64_t xb = x0 + wx * 64 + (lane >> 4) * 16;
  float bcol[16];
  if (!BIAS_ROWS) {
#pragma unroll
    for (int j = 0; j < 16; ++j) bcol[j] = 0.f;
    if (bias != nullptr && xb + 16 <= XN) {
      unpack8t<F16>(*reinterpret_cast<const uint4*>(bias + xb), bcol);
      unpack8t<F16>(*reinterpret_cast<const uint4*>(bias + xb + 8), bcol + 8);
    }
  }
#pragma unroll
  for (int yt = 0; yt < 4; ++yt) {
    const int64_t yr = y0 + wy * 64 + yt * 16 + (lane & 15);
    if (yr >= YM) continue;
    float v[16];
    const float brow = (BIAS_ROWS && bias != nullptr) ? ld16<F16>(bias[yr]) : 0.f;
#pragma unroll
    for (int xt = 0; xt < 4; ++xt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float b = BIAS_ROWS ? brow : bcol[xt * 4 + r];
        v[xt * 4 + r] = rnd16<F16>(acc[yt][xt][r] * out_scale + b);  // nn.Linear output is a 16-bit tensor
      }
    bf16_t* optr = Out + yr * ldo + xb;
    if (xb + 16 <= XN) {
      if (EPI == WAN_EPI_GELU_TANH) {
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = gelu_tanh_f(v[j]);
      } else if (EPI == WAN_EPI_GATE_RES) {
        float rv[16];
        const bf16_t* rptr = R + yr * ldo + xb;
        unpack8t<F16>(*reinterpret_cast<const uint4*>(rptr), rv);
        unpack8t<F16>(*reinterpret_cast<const uint4*>(rptr + 8), rv + 8);
        if (gate_idx >= 0) {
          const int64_t bidx = yr / rows_per_batch;
          float mv[16], ev[16];
          const bf16_t* mp = mod + (int64_t)gate_idx * XN + xb;
          const bf16_t* ep = e + (bidx * n_mod + gate_idx) * XN + xb;
          unpack8t<F16>(*reinterpret_cast<const uint4*>(mp), mv);
          unpack8t<F16>(*reinterpret_cast<const uint4*>(mp + 8), mv + 8);
          unpack8t<F16>(*reinterpret_cast<const uint4*>(ep), ev);
          unpack8t<F16>(*reinterpret_cast<const uint4*>(ep + 8), ev + 8);
#pragma unroll
          for (int j = 0; j < 16; ++j) v[j] = rv[j] + v[j] * rnd16<F16>(mv[j] + ev[j]);
        } else {
#pragma unroll
          for (int j = 0; j < 16; ++j) v[j] = rv[j] + v[j];
        }
      }
      *reinterpret_cast<uint4*>(optr) = pack8t<F16>(v);
      *reinterpret_cast<uint4*>(optr + 8) = pack8t<F16>(v + 8);
    } else {
      // ragged x edge (only the transposed/V^T form can hit this: x = tokens)
      for (int j = 0; j < 16; ++j) {
        if (xb + j < XN) {
          float o = v[j];
          if (!BIAS_ROWS && bias != nullptr) o = rnd16<F16>(acc[yt][j >> 2][j & 3] * out_scale + ld16<F16>(bias[xb + j]));
          if (EPI == WAN_EPI_GELU_TANH) o = gelu_tanh_f(o);
          if (EPI == WAN_EPI_GATE_RES) {
            float g = 1.f;
            if (gate_idx >= 0)
              g = rnd16<F16>(ld16<F16>(mod[(int64_t)gate_idx * XN + xb + j]) +
                             ld16<F16>(e[((yr / rows_per_batch) * n_mod + gate_idx) * XN + xb + j]));
            o = ld16<F16>(R[yr * ldo + xb + j]) + o * g;
          }
          optr[j] = st16<F16>(o);
        }
      }
    }
  }
}

// second-generation kernel (gemm32.hip); returns -1 when the problem does not fit it
template <int EPI, bool BIAS_ROWS, bool F16>
int wan_gemm32_try(const bf16_t* Y, int64_t ldy, int64_t YM, const bf16_t* X, int64_t ldx, int64_t XN, int K, bf16_t* Out,
                   int64_t ldo, const bf16_t* bias, const bf16_t* R, const bf16_t* mod, const bf16_t* e, int n_mod,
                   int gate_idx, int64_t rows_per_batch, hipStream_t st, float out_scale);

// fourth-generation kernel (gemm256k.hip): 256x256x64 tiles, one wave per SIMD, accumulators in the accumulator file
template <int EPI, bool BIAS_ROWS, bool F16>
int wan_gemm256k_try(const bf16_t* Y, int64_t ldy, int64_t YM, const bf16_t* X, int64_t ldx, int64_t XN, int K, bf16_t* Out,
                    int64_t ldo, const bf16_t* bias, const bf16_t* R, const bf16_t* mod, const bf16_t* e, int n_mod,
                    int gate_idx, int64_t rows_per_batch, hipStream_t st, float out_scale);

// fifth-generation kernel (gemm256m.hip): gemm256k's tile on the 16x16x32 MFMA, register-direct epilogue; bf16, bias per column
template <int EPI, bool BIAS_ROWS>
int wan_gemm256m_try(const bf16_t* Y, int64_t ldy, int64_t YM, const bf16_t* X, int64_t ldx, int64_t XN, int K, bf16_t* Out,
                     int64_t ldo, const bf16_t* bias, const bf16_t* R, const bf16_t* mod, const bf16_t* e, int n_mod,
                     int gate_idx, int64_t rows_per_batch, hipStream_t st, float out_scale);

// ... with the tile height as an argument (round 6: 256 / 224 / 192 / 160 rows) and the rule that picks it
template <int EPI, bool BIAS_ROWS>
int wan_gemm256m_try_h(const bf16_t* Y, int64_t ldy, int64_t YM, const bf16_t* X, int64_t ldx, int64_t XN, int K, bf16_t* Out,
                       int64_t ldo, const bf16_t* bias, const bf16_t* R, const bf16_t* mod, const bf16_t* e, int n_mod,
                       int gate_idx, int64_t rows_per_batch, hipStream_t st, float out_scale, int tile_rows);
int wan_gemm256m_tile_rows(int64_t YM, int64_t XN, int cus);

// sixth-generation kernel (gemm16s.hip): 128 x 128 / 256 x 128 tiles on the 16x16x32 MFMA, three / two workgroups per CU; bf16
template <int EPI, bool BIAS_ROWS>
int wan_gemm16s_try(const bf16_t* Y, int64_t ldy, int64_t YM, const bf16_t* X, int64_t ldx, int64_t XN, int K, bf16_t* Out,
                    int64_t ldo, const bf16_t* bias, const bf16_t* R, const bf16_t* mod, const bf16_t* e, int n_mod,
                    int gate_idx, int64_t rows_per_batch, hipStream_t st, float out_scale, int tile_rows);

static int g_force_rows = 0;   // test / A-B hook: 0 = wan_gemm256m_tile_rows decides, 256 / 224 / 192 / 160 = that height wherever gemm256m runs, -1 = the round-5 rule (256 rows, many-tile problems only)
extern "C" int wan_gemm_debug_force_tile_rows(int v) {
  const int old = g_force_rows;
  g_force_rows = (v == 256 || v == 224 || v == 192 || v == 160 || v == -1) ? v : 0;
  return old;
}
static int g_force16s = [] { const char* e = getenv("WAN_GEMM16S"); return e ? atoi(e) : 0; }();
extern "C" int wan_gemm_debug_force16s(int v) {
  const int old = g_force16s;
  g_force16s = (v == 128 || v == 256 || v == -1) ? v : 0;
  return old;
}

#ifndef WAN_GEMM_MIN_TILES_DEFAULT
#define WAN_GEMM_MIN_TILES_DEFAULT 256
#endif
// how many 256 x 256 tiles make a "many-tile" problem (WAN_GEMM_MIN_TILES overrides for A/B runs): ONE value for launch_gemm and for the
// fp32-stream form below (round-5 advisor: wan_gemm_bf16_res32 hard-coded 256 while launch_gemm read the variable)
static int64_t gemm_min_tiles() {
  static const int64_t v = [] { const char* e = getenv("WAN_GEMM_MIN_TILES"); const long n = e ? atol(e) : 0; return (int64_t)(n > 0 ? n : WAN_GEMM_MIN_TILES_DEFAULT); }();
  return v;
}
template <int EPI, bool BIAS_ROWS, bool F16 = false>
static int launch_gemm(const bf16_t* Y, int64_t ldy, int64_t YM, const bf16_t* X, int64_t ldx, int64_t XN, int K,
                       bf16_t* Out, int64_t ldo, const bf16_t* bias, const bf16_t* R, const bf16_t* mod,
                       const bf16_t* e, int n_mod, int gate_idx, int64_t rows_per_batch, hipStream_t st,
                       float out_scale = 1.0f) {
  const int64_t tx = (XN + BN - 1) / BN;
  // how many 256 x 256 tiles a problem must have for the one-wave-per-SIMD kernels (a tile per CU: below one wave of tiles the chip is
  // partly idle for a whole tile time, and the 128-wide kernels balance better).  WAN_GEMM_MIN_TILES overrides for A/B runs.
  const int64_t min_tiles = gemm_min_tiles();
  const bool many_tiles = ((YM + 255) / 256) * ((XN + 255) / 256) >= min_tiles;
  // Round 6 dispatch of the bf16 Linears (run 09 / 10, profiles/r06_ab_gemm_tile_heights_run09.log, r06_ab_gemm_text_shapes_run10.log; one
  // process, alternating, bit-identical outputs):
  //   1. gemm256m.hip at the tile HEIGHT that leaves the fewest CUs idle (wan_gemm256m_tile_rows: 256, 224, 192 or 160 rows x 256 columns)
  //      whenever that gives at least 128 tiles: every many-tile problem as before (256 rows; 224 where a round is saved), and what gemm32
  //      served until round 5 -- BASELINE configs[0]'s M = 6,400 rows as 40 x 6 tiles of 160 rows on 240 CUs instead of 25 x 6 of 256 on 150:
  //      q / k / o 46.9 -> 28.0 us, ffn.2 + gate 240.8 -> 128.2 us; the 14B text K Linear (M = 1,024) 93.0 -> 58.9 us;
  //   2. below that, gemm16s.hip's co-resident small tiles (256 x 128 from 128 such tiles up, else 128 x 128): UMT5's projections at
  //      M = 512 (70.3 -> 35.0 us), the 1.3B text K / V Linears (28.4 -> 14.6 us), the 14B text V^T (92.3 -> 53.4 us).
  //      (A 128 x 128 x 32 stage fetches a byte per 64 FLOP, a 256 x 128 one per 85, the 256 x 256 tile per 128: 64 / 48 / 32 B/clk/CU of
  //      LDS-DMA at the matrix pipe's peak.  Measured at many tiles, same shapes, one process: 0.30 / 0.40 / 0.58 of peak -- the fetch path,
  //      not the matrix pipe, bounds the small tiles; they lose by 1.4-1.8 x there, which is why they stop here.)
  // Hooks (tests, in-process A/B): g_force_rows (wan_gemm_debug_force_tile_rows) and g_force16s (env WAN_GEMM16S at load,
  // wan_gemm_debug_force16s): -1 = the round-5 rule, 0 = this dispatch, a tile size = that tile on every problem it fits.
#ifndef WAN_GEMM_NO_MI16  // (defined only for the A/B library libwanhip_k.so: gemm256k on every shape)
  if constexpr (!F16 && (!BIAS_ROWS || EPI == WAN_EPI_NONE)) {
    static const int cus = [] { int dev = 0, n = 0; return (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256; }();
    if (g_force16s > 0) {   // (forced small tiles come first, many-tile problems included: the A/B tool's t128 / t256 columns)
      const int rc = wan_gemm16s_try<EPI, BIAS_ROWS>(Y, ldy, YM, X, ldx, XN, K, Out, ldo, bias, R, mod, e, n_mod, gate_idx, rows_per_batch, st, out_scale, g_force16s);
      if (rc >= 0) return rc;
    }
    const int rows = g_force_rows == -1 ? 256 : g_force_rows ? g_force_rows : wan_gemm256m_tile_rows(YM, XN, cus);
    const int64_t th = ((YM + rows - 1) / rows) * ((XN + 255) / 256);
    const bool tall = g_force_rows == -1 ? many_tiles : (g_force_rows > 0 || many_tiles || (th >= 128 && min_tiles == WAN_GEMM_MIN_TILES_DEFAULT && g_force16s <= 0));
    if (tall) {
      const int rc = wan_gemm256m_try_h<EPI, BIAS_ROWS>(Y, ldy, YM, X, ldx, XN, K, Out, ldo, bias, R, mod, e, n_mod, gate_idx, rows_per_batch, st, out_scale, rows);
      if (rc >= 0) return rc;
      if (rows != 256 && many_tiles) {   // (a batch shorter than the chosen height, ...: the full-height tile)
        const int rc2 = wan_gemm256m_try<EPI, BIAS_ROWS>(Y, ldy, YM, X, ldx, XN, K, Out, ldo, bias, R, mod, e, n_mod, gate_idx, rows_per_batch, st, out_scale);
        if (rc2 >= 0) return rc2;
      }
    }
    const int f16s = g_force16s;
    if (f16s > 0 || (f16s == 0 && !many_tiles && YM >= 32 && XN >= 32)) {
      const int64_t t256 = ((YM + 255) / 256) * ((XN + 127) / 128);
      const int rc = wan_gemm16s_try<EPI, BIAS_ROWS>(Y, ldy, YM, X, ldx, XN, K, Out, ldo, bias, R, mod, e, n_mod, gate_idx, rows_per_batch, st, out_scale,
                                                     f16s > 0 ? f16s : (t256 >= 128 ? 256 : 128));
      if (rc >= 0) return rc;
    }
  }
#endif
  if (many_tiles) {
    const int rc = wan_gemm256k_try<EPI, BIAS_ROWS, F16>(Y, ldy, YM, X, ldx, XN, K, Out, ldo, bias, R, mod, e, n_mod, gate_idx,
                                                         rows_per_batch, st, out_scale);
    if (rc >= 0) return rc;
  }
  if (YM >= 512 && XN >= 128) {
    const int rc = wan_gemm32_try<EPI, BIAS_ROWS, F16>(Y, ldy, YM, X, ldx, XN, K, Out, ldo, bias, R, mod, e, n_mod, gate_idx,
                                                       rows_per_batch, st, out_scale);
    if (rc >= 0) return rc;
  }
  const int64_t ty = (YM + 127) / 128;
  WAN_REQUIRE(ty * tx < (int64_t)1 << 31, "wan_gemm: too many tiles");
  hipLaunchKernelGGL((gemm_bf16_kernel<EPI, BIAS_ROWS, F16>), dim3((unsigned)(ty * tx)), dim3(256), 0, st, Y, ldy, YM, X, ldx,
                     XN, K, Out, ldo, bias, R, mod, e, n_mod, gate_idx, rows_per_batch, (int)ty, (int)tx, out_scale);
  WAN_LAUNCH_CHECK();
  return 0;
}

// The mixed-precision plan's Linear + gated residual in one launch (csrc/gemm256m.hip, epilogue M_EPI_RES32 = 4):
//   x32[M, N] (fp32, in place) += bf16(A W^T + bias) * (mod[gate_idx] + e0[row / rows_per_batch][gate_idx])        (gate_idx < 0: gate = 1)
// mod bf16 [n_mod, N], e0 fp32 [batches, n_mod, N] -- wan_gemm_bf16(WAN_EPI_NONE) into `tmp` followed by wan_mx_gated_residual, which
// is also what runs when the shape does not fit the 256 x 256 tile kernel (few tiles, N % 256 != 0, a batch shorter than a tile).
// Same operations in the same order on the same values: bit-identical to the two-launch form.
extern "C" int wan_gemm_bf16_res32(const wan_bf16* A, int64_t lda, const wan_bf16* W, const wan_bf16* bias, float* x32, wan_bf16* tmp,
                                   int64_t M, int N, int K, const wan_bf16* mod, const float* e0, int n_mod, int gate_idx,
                                   int64_t rows_per_batch, void* stream) {
  WAN_REQUIRE(A && W && x32 && tmp, "wan_gemm_bf16_res32: null operand");
  WAN_REQUIRE(K > 0 && K % BK == 0 && N % 16 == 0, "wan_gemm_bf16_res32: K=%d must be a positive multiple of %d, N=%d of 16", K, BK, N);
  WAN_REQUIRE(lda % 8 == 0 && (((uintptr_t)A | (uintptr_t)W | (uintptr_t)x32 | (uintptr_t)tmp) & 15) == 0,
              "wan_gemm_bf16_res32: lda must be a multiple of 8, pointers 16-byte aligned");
  WAN_REQUIRE(gate_idx < 0 || (mod && e0 && gate_idx < n_mod && rows_per_batch > 0), "wan_gemm_bf16_res32: bad gate arguments");
  if (M == 0) return 0;
  hipStream_t st = as_stream(stream);
#ifndef WAN_GEMM_NO_MI16
  if (((M + 255) / 256) * (((int64_t)N + 255) / 256) >= gemm_min_tiles()) {
    const int rc = wan_gemm256m_try<4, false>(A, lda, M, W, K, N, K, reinterpret_cast<bf16_t*>(x32), N, bias, nullptr, mod,
                                              reinterpret_cast<const bf16_t*>(e0), n_mod, gate_idx, rows_per_batch > 0 ? rows_per_batch : 1, st, 1.0f);
    if (rc >= 0) return rc;
  }
#endif
  if (int rc = launch_gemm<WAN_EPI_NONE, false>(A, lda, M, W, K, N, K, tmp, N, bias, nullptr, nullptr, nullptr, 0, -1, 1, st)) return rc;
  return wan_mx_gated_residual(x32, tmp, mod, e0, n_mod, gate_idx, M, rows_per_batch > 0 ? rows_per_batch : M, N, stream);
}

// fp16 GEMM for the VAE attention block: C[M,N] (ldc) = scale * A[M,K](lda) @ W[N,K](ldw)^T + bias, or its
// transpose Ct[N, ldc] (transposed != 0).  K % 64 == 0; N % 16 == 0 unless transposed.
extern "C" int wan_gemm_f16(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, const uint16_t* bias,
                            uint16_t* C, int64_t ldc, int64_t M, int64_t N, int K, float scale, int transposed,
                            void* stream) {
  WAN_REQUIRE(A && W && C, "wan_gemm_f16: null operand");
  WAN_REQUIRE(K > 0 && K % BK == 0, "wan_gemm_f16: K=%d must be a positive multiple of %d", K, BK);
  WAN_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && ldc % 8 == 0, "wan_gemm_f16: leading dimensions must be multiples of 8");
  WAN_REQUIRE((((uintptr_t)A | (uintptr_t)W | (uintptr_t)C) & 15) == 0, "wan_gemm_f16: pointers must be 16-B aligned");
  if (M == 0 || N == 0) return 0;
  hipStream_t st = as_stream(stream);
  if (transposed)
    return launch_gemm<WAN_EPI_NONE, true, true>(W, ldw, N, A, lda, M, K, C, ldc, bias, nullptr, nullptr, nullptr, 0, -1,
                                                 1, st, scale);
  WAN_REQUIRE(N % 16 == 0, "wan_gemm_f16: N=%lld must be a multiple of 16", (long long)N);
  return launch_gemm<WAN_EPI_NONE, false, true>(A, lda, M, W, ldw, N, K, C, ldc, bias, nullptr, nullptr, nullptr, 0, -1, 1,
                                                st, scale);
}

extern "C" int wan_gemm_bf16(const wan_bf16* A, int64_t lda, const wan_bf16* W, const wan_bf16* bias, wan_bf16* C,
                             int64_t ldc, int64_t M, int N, int K, int epilogue, const wan_bf16* R,
                             const wan_bf16* mod, const wan_bf16* e, int n_mod, int gate_idx, int64_t rows_per_batch,
                             void* stream) {
  WAN_REQUIRE(A && W && C, "wan_gemm_bf16: null operand");
  WAN_REQUIRE(K > 0 && K % BK == 0, "wan_gemm_bf16: K=%d must be a positive multiple of %d", K, BK);
  WAN_REQUIRE(lda % 8 == 0 && ldc % 8 == 0, "wan_gemm_bf16: lda/ldc must be multiples of 8 elements (16 B)");
  WAN_REQUIRE((((uintptr_t)A | (uintptr_t)W | (uintptr_t)C) & 15) == 0, "wan_gemm_bf16: pointers must be 16-B aligned");
  if (M == 0 || N == 0) return 0;
  hipStream_t st = as_stream(stream);
  switch (epilogue) {
    case WAN_EPI_NONE:
      WAN_REQUIRE(N % 16 == 0, "wan_gemm_bf16: N=%d must be a multiple of 16", N);
      return launch_gemm<WAN_EPI_NONE, false>(A, lda, M, W, K, N, K, C, ldc, bias, nullptr, nullptr, nullptr, 0, -1, 1, st);
    case WAN_EPI_GELU_TANH:
      WAN_REQUIRE(N % 16 == 0, "wan_gemm_bf16: N=%d must be a multiple of 16", N);
      return launch_gemm<WAN_EPI_GELU_TANH, false>(A, lda, M, W, K, N, K, C, ldc, bias, nullptr, nullptr, nullptr, 0, -1, 1, st);
    case WAN_EPI_GATE_RES:
      WAN_REQUIRE(N % 16 == 0, "wan_gemm_bf16: N=%d must be a multiple of 16", N);
      WAN_REQUIRE(R != nullptr, "wan_gemm_bf16: GATE_RES needs the residual R");
      WAN_REQUIRE(gate_idx < 0 || (mod && e && gate_idx < n_mod && rows_per_batch > 0), "wan_gemm_bf16: bad gate args");
      return launch_gemm<WAN_EPI_GATE_RES, false>(A, lda, M, W, K, N, K, C, ldc, bias, R, mod, e, n_mod, gate_idx,
                                                  rows_per_batch > 0 ? rows_per_batch : 1, st);
    case WAN_EPI_TRANSPOSED:
      // Ct[N, ldc]: roles swapped, bias runs along output rows
      return launch_gemm<WAN_EPI_NONE, true>(W, K, N, A, lda, M, K, C, ldc, bias, nullptr, nullptr, nullptr, 0, -1, 1, st);
    default:
      WAN_REQUIRE(false, "wan_gemm_bf16: unknown epilogue %d", epilogue);
  }
  return 0;
}
