// Scaled-fp8 Linear for gfx950: Out = epilogue( (A_fp8 W_fp8^T) * scale_a * scale_w + bias ), OCP e4m3fn operands, fp32 accumulate
// on v_mfma_f32_32x32x64_f8f6f4 (K = 64 per instruction, twice the bf16 MFMA rate).
//
// Replaces the reference's fp8 checkpoint path, ScaledFP8WeightTensor._linear_scaled (shared/qtypes/scaled_fp8.py:324-380):
//   * the ACTIVATION is quantised dynamically, per tensor (:162-169 _quantize_activation): scale_a = absmax / 448 (fp32),
//     q = fp8( clamp( bf16( x / bf16(scale_a) ), -448, 448 ) ) -- wan_fp8_quantize below (two streaming kernels);
//   * torch._scaled_mm(x_fp8, W^T, scale_a, scale_b, bias, out_dtype=bf16): fp32 accumulation, (acc * scale_a) * scale_b (+ bias),
//     one bf16 rounding -- per-tensor weight scale;
//   * per-output-row weight scale (:150-159, :368-378): scale_b = 1 and no bias inside the product; then, on the bf16
//     result, `out *= bf16(scale_w[n])` and `out += bias`, one rounding each.
// The same fused epilogues as the bf16 GEMMs follow (GELU-tanh, gated residual, transposed V^T output).
//
// Kernel = gemm256k.hip's structure with one byte per element: 256x256 output tile, 4 waves = one per SIMD, 256 accumulators
// in the accumulator file, k-tile of 128 elements = 128 B = one cache line per row per fetch, ring of five 32-KB LDS units
// (one operand of one stage each: Y0 X0 Y1 X1 Y2 | X2 ...), LDS-DMA with loop-invariant per-lane offsets, chunk swizzle
// p ^ ((row >> 1) & 7).  A stage is TWO k-steps of 16 MFMAs (64 cycles each): the same 2048 MFMA cycles and the same 64 KB
// per stage as the bf16 kernel -- twice the FLOP.
//   k-step 0 (fragments f0): Y_{S+2} pieces 0..7 after the even MFMAs (into X_{S-1}'s slot, free since P_{S-1}); the 16
//            ds_read_b128 of k-step 1's fragments -> f1, one per MFMA gap
//   P_S: s_waitcnt vmcnt(8) lgkmcnt(0) + barrier: X_{S+1} and everything older landed (Y_{S+2} may fly), stage S fully read
//   k-step 1 (f1): X_{S+2} pieces 0..7 after the even MFMAs (into Y_S's slot, dead now); stage S+1's k-step-0 fragments -> f0
// A lane's MFMA operand is 32 consecutive k bytes of its row (k = 64 ks + 32 half ..): two 16-B LDS chunks.  Which k each
// byte means to the hardware is irrelevant as long as both operands are loaded the same way (a sum over k).
#include <stdlib.h>

#include "common.h"

namespace {

typedef uint32_t f8_u4 __attribute__((ext_vector_type(4)));
typedef uint32_t f8_u8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) const char f8_lds_cchar;
typedef __attribute__((address_space(3))) const f8_u4 f8_lds_u4;

constexpr int F_BM = 256, F_BN = 256, F_BK = 128;
constexpr int F_UNIT = 256 * F_BK;  // 32 KiB: one operand of one stage (256 rows x 128 B)
constexpr int F_NU = 5;

__device__ __forceinline__ float f8_gelu_tanh(float x) {
  const float c = -2.0f * 0.7978845608028654f * 1.4426950408889634f;  // -2 sqrt(2/pi) log2(e)
  const float t = __builtin_fmaf(x * x, 0.044715f, 1.0f);
  const float ex = __builtin_amdgcn_exp2f(x * t * c);
  return x * __builtin_amdgcn_rcpf(1.0f + ex);
}

struct F8Frags {
  f8_u8 y[4], x[4];
};
__device__ __forceinline__ void mfma_f8(f32x16& acc, const f8_u8& a, const f8_u8& b) {
  asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void f8_dma16(uint32_t voff, const f8_u4& rsrc, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rsrc), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ f8_u4 f8_rsrc(const void* base) {
  const uint64_t b = (uint64_t)base;
  f8_u4 r;
  r[0] = (uint32_t)b;
  r[1] = (uint32_t)(b >> 32) & 0xffffu;
  r[2] = 0xffffffffu;
  r[3] = 0x00020000u;
  return r;
}

// SCALE_ROWS: the weight scale is a vector along the weights' rows (0 = one scalar).  BIAS_ROWS: the weights are the y
// operand (transposed V^T output: out[n][token]); otherwise the x operand.
template <int EPI, bool BIAS_ROWS, bool SCALE_VEC>
__global__ __launch_bounds__(256) void gemm_fp8_kernel(const uint8_t* __restrict__ Y, int64_t ldy, int64_t YM,
                                                      const uint8_t* __restrict__ X, int64_t ldx, int64_t XN, int K,
                                                      bf16_t* __restrict__ Out, int64_t ldo, const bf16_t* __restrict__ bias,
                                                      const float* __restrict__ scale_a, const float* __restrict__ scale_w,
                                                      const bf16_t* __restrict__ R, const bf16_t* __restrict__ mod,
                                                      const bf16_t* __restrict__ e, int n_mod, int gate_idx,
                                                      int64_t rows_per_batch, int tiles_y, int tiles_x, int group) {
  __shared__ __attribute__((aligned(16))) char smem[F_NU * F_UNIT];  // 160 KiB
  f8_lds_cchar* lds = (f8_lds_cchar*)smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wy = wave >> 1, wx = wave & 1;
  const int l31m = lane & 31, halfm = lane >> 5;  // main-loop copies (fragment addresses)

  // ---- tile assignment: XCD-contiguous ids, then grouped ordering (gemm256k.hip) ------------------------------------------
  const int nwg = tiles_y * tiles_x;
  const int wg = xcd_remap(blockIdx.x, nwg);
  const int per_group = group * tiles_x;
  const int gidx = wg / per_group;
  const int first_y = gidx * group;
  const int gsz = min(tiles_y - first_y, group);
  const int in_g = wg - gidx * per_group;
  const int ty = first_y + (in_g % gsz);
  const int tx = in_g / gsz;
  const int64_t y0 = (int64_t)ty * F_BM;
  const int64_t x0 = (int64_t)tx * F_BN;

  // ---- DMA plan: piece i (0..7) of wave w fills 16-B slots q = i*256 + w*64 + lane = rows i*32 + w*8 .. +8, 8 lanes per row ----
  uint32_t yofs[8], xofs[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int q = i * 256 + tid;
    const int row = q >> 3, pch = q & 7;
    const int lch = pch ^ ((row >> 1) & 7);
    int64_t yr = y0 + row;
    if (yr > YM - 1) yr = YM - 1;  // ragged tile: re-read the last row (its results are never stored)
    yofs[i] = (uint32_t)((yr - y0) * ldy + lch * 16);
    const int slab = row >> 7, xt = (row >> 5) & 3, rho = row & 31;
    int64_t xr = x0 + slab * 128 + xt * 32 + 16 * ((rho >> 2) & 1) + (rho & 3) + 4 * (rho >> 3);  // x rows permuted: a lane's 16 accumulators = 16 consecutive columns
    if (xr > XN - 1) xr = XN - 1;
    xofs[i] = (uint32_t)((xr - x0) * ldx + lch * 16);
  }
  const char* ybase = reinterpret_cast<const char*>(Y + y0 * ldy);
  const char* xbase = reinterpret_cast<const char*>(X + x0 * ldx);
  const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int nk = K / F_BK;
  int ky = 0, kx = 0;
  auto y_piece = [&](int slot, int p) { f8_dma16(yofs[p], f8_rsrc(ybase), smem_lds + slot * F_UNIT + (p * 256 + wave * 64) * 16); };
  auto x_piece = [&](int slot, int p) { f8_dma16(xofs[p], f8_rsrc(xbase), smem_lds + slot * F_UNIT + (p * 256 + wave * 64) * 16); };
  auto y_advance = [&]() { const bool ok = ky + 1 < nk; ybase += ok ? F_BK : 0; ky += ok ? 1 : 0; };
  auto x_advance = [&]() { const bool ok = kx + 1 < nk; xbase += ok ? F_BK : 0; kx += ok ? 1 : 0; };

  f32x16 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
      asm volatile("" : "+a"(acc[a][b]));
    }

  // ---- fragment addresses: k-step ks (0, 1) reads logical chunks 4 ks + 2 half and + 1 of the lane's row --------------------
  const int sw = (l31m >> 1) & 7;
  const int yaddr0 = (wy * 128 + l31m) * 128 + (((2 * halfm) ^ sw) << 4);  // k-step 0, low 16 B; high 16 B: ^ 16; k-step 1: ^ 64
  const int xaddr0 = (wx * 128 + l31m) * 128 + (((2 * halfm) ^ sw) << 4);
  // read h = 0..15 of a fragment set: operand tile (h >> 1) & 3 of y (h < 8) or x (h >= 8), 16-B part h & 1
  auto load_part = [&](F8Frags& f, int sy, int sx, int ks, int h) {
    const int r = (h >> 1) & 3, part = h & 1;
    const f8_u4 v = (h < 8) ? *(f8_lds_u4*)(lds + (sy * F_UNIT + r * 4096) + ((yaddr0 ^ (ks << 6)) ^ (part << 4)))
                            : *(f8_lds_u4*)(lds + (sx * F_UNIT + r * 4096) + ((xaddr0 ^ (ks << 6)) ^ (part << 4)));
    f8_u8& d = (h < 8) ? f.y[r] : f.x[r];
    d[part * 4 + 0] = v[0]; d[part * 4 + 1] = v[1]; d[part * 4 + 2] = v[2]; d[part * 4 + 3] = v[3];
  };

  // prologue: stages 0 and 1 (units 0..3)
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
  F8Frags f0, f1;
#pragma unroll
  for (int h = 0; h < 16; ++h) load_part(f0, 0, 1, 0, h);

#define F8_SB() __builtin_amdgcn_sched_barrier(0)
#define F8_STEP(J)                                                                                              \
  if (__builtin_expect(kt + (J) < nk, 1)) {                                                                     \
    constexpr int J_ = (J);                                                                                     \
    constexpr int SY = (2 * J_) % 5, SX = (2 * J_ + 1) % 5, NY = (2 * J_ + 2) % 5, NX = (2 * J_ + 3) % 5;        \
    constexpr int DY = (2 * J_ + 4) % 5, DX = (2 * J_) % 5;                                                      \
    _Pragma("unroll") for (int m = 0; m < 16; ++m) {                                                            \
      mfma_f8(acc[m >> 2][m & 3], f0.x[m & 3], f0.y[m >> 2]); F8_SB();                                           \
      if ((m & 1) == 0) y_piece(DY, m >> 1);                                                                     \
      load_part(f1, SY, SX, 1, m);                                                                               \
      F8_SB();                                                                                                  \
    }                                                                                                           \
    y_advance();                                                                                                \
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");                                                 \
    __builtin_amdgcn_s_barrier();                                                                               \
    asm volatile("" ::: "memory");                                                                              \
    _Pragma("unroll") for (int m = 0; m < 16; ++m) {                                                            \
      mfma_f8(acc[m >> 2][m & 3], f1.x[m & 3], f1.y[m >> 2]); F8_SB();                                           \
      if ((m & 1) == 0) x_piece(DX, m >> 1);                                                                     \
      load_part(f0, NY, NX, 0, m);                                                                               \
      F8_SB();                                                                                                  \
    }                                                                                                           \
    x_advance();                                                                                                \
  }
  for (int kt = 0; kt < nk; kt += 5) {
    F8_STEP(0)
    F8_STEP(1)
    F8_STEP(2)
    F8_STEP(3)
    F8_STEP(4)
  }
#undef F8_STEP
#undef F8_SB
  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");  // trailing DMA; last asm MFMAs -> accumulator reads

  // ---- epilogue, through LDS (gemm256k.hip): scales / bias / GELU in the accumulator layout, parked per wave, read back row-major
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  // every lane-derived index of the epilogue is re-derived from ONE value that crosses the main loop (the allocator otherwise
  // keeps four of them alive through it -- and spills one)
  int tid_e = tid;
  asm volatile("" : "+v"(tid_e));
  const int lane_e = tid_e & 63;
  const int l31 = lane_e & 31, half = lane_e >> 5;
  const float sa = scale_a[0];
  const float sb = SCALE_VEC ? 1.0f : scale_w[0];
  constexpr int EROW = 272;
  char* const park = smem + wave * (128 * EROW);
#pragma unroll
  for (int xt = 0; xt < 4; ++xt) {
    const int64_t xb = x0 + wx * 128 + xt * 32 + half * 16;
    float bcol[16], scol[16];
    if (!BIAS_ROWS) {
#pragma unroll
      for (int j = 0; j < 16; ++j) { bcol[j] = 0.f; scol[j] = 1.f; }
      const bool full = xb + 16 <= XN;
      if (bias != nullptr) {
        if (full) {
          unpack8(*reinterpret_cast<const uint4*>(bias + xb), bcol);
          unpack8(*reinterpret_cast<const uint4*>(bias + xb + 8), bcol + 8);
        } else {
#pragma unroll
          for (int j = 0; j < 16; ++j)
            if (xb + j < XN) bcol[j] = bf2f(bias[xb + j]);
        }
      }
      if (SCALE_VEC) {
#pragma unroll
        for (int j = 0; j < 16; ++j)
          if (xb + j < XN) scol[j] = rbf(scale_w[xb + j]);  // output_scale.to(out.dtype)
      }
    }
#pragma unroll
    for (int yt = 0; yt < 4; ++yt) {
      int64_t yr = y0 + wy * 128 + yt * 32 + l31;
      if (yr > YM - 1) yr = YM - 1;
      float v[16];
      const float brow = (BIAS_ROWS && bias != nullptr) ? bf2f(bias[yr]) : 0.f;
      const float srow = (BIAS_ROWS && SCALE_VEC) ? rbf(scale_w[yr]) : 1.f;
      asm volatile("" : "+a"(acc[yt][xt]));  // the tile stays in the accumulator file up to HERE (its copy-out cannot be hoisted above)
      const f32x16 av = acc[yt][xt];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float bj = BIAS_ROWS ? brow : bcol[r];
        float o = av[r] * sa * sb;
        if (SCALE_VEC) {
          o = rbf(o);                                          // _scaled_mm output (scale_b = 1, no bias), bf16
          o = rbf(o * (BIAS_ROWS ? srow : scol[r]));           // out *= output_scale
          if (bias != nullptr) o = rbf(o + bj);                // out += bias
        } else {
          o = o + bj;                                          // bias inside _scaled_mm, one rounding (the pack below / rbf)
        }
        if (EPI == WAN_EPI_GELU_TANH) o = f8_gelu_tanh(rbf(o));
        v[r] = o;
      }
      uint4* dst = reinterpret_cast<uint4*>(park + (yt * 32 + l31) * EROW + (xt * 32 + half * 16) * 2);
      dst[0] = pack8(v);
      dst[1] = pack8(v + 8);
      // one accumulator tile at a time: left free, the scheduler hoists the next tiles' accumulator reads over this tile's GELU and,
      // in the GELU + per-row-scale instantiation (32 live scale / bias values), ran out of registers (68 B of scratch per lane)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int prow = lane_e >> 4, pchunk = lane_e & 15;
  const int64_t xc = x0 + wx * 128 + pchunk * 8;
  const bool col_full = xc + 8 <= XN;
  const int64_t yrow0 = y0 + wy * 128 + prow;
  uint4 mchunk = {};
  if (EPI == WAN_EPI_GATE_RES && gate_idx >= 0 && col_full) mchunk = *reinterpret_cast<const uint4*>(mod + (int64_t)gate_idx * XN + xc);
  // residual / gate rows of GDEPTH 4-row groups in flight during the read-back (gemm256k.hip: with two the gated epilogue was
  // latency-bound); values are returned from the lambdas, arrays passed by reference end up in scratch memory
  constexpr int GDEPTH = 8;
  uint4 rqa[GDEPTH] = {}, eqa[GDEPTH] = {};
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
  auto emit = [&](int i, const uint4 rq, const uint4 eq) {
    const int64_t yr = yrow0 + i * 4;
    const uint4 raw = *reinterpret_cast<const uint4*>(park + (i * 4 + prow) * EROW + pchunk * 16);
    if (yr >= YM) return;
    bf16_t* optr = Out + yr * ldo + xc;
    if (col_full) {
      if (EPI == WAN_EPI_GATE_RES) {
        float v[8], rv[8];
        unpack8(raw, v);
        unpack8(rq, rv);
        if (gate_idx >= 0) {
          float mv[8], ev[8];
          unpack8(mchunk, mv);
          unpack8(eq, ev);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = rv[j] + v[j] * rbf(mv[j] + ev[j]);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = rv[j] + v[j];
        }
        *reinterpret_cast<uint4*>(optr) = pack8(v);
      } else {
        *reinterpret_cast<uint4*>(optr) = raw;
      }
    } else if (EPI == WAN_EPI_NONE) {  // ragged x edge: only the transposed / V^T form (x = tokens) can hit it
      const uint32_t w4[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (xc + j < XN) optr[j] = (bf16_t)(w4[j >> 1] >> ((j & 1) * 16));
    }
  };
#pragma unroll
  for (int i = 0; i < GDEPTH; ++i) { rqa[i] = fetch_r(i); eqa[i] = fetch_e(i); }
#pragma unroll
  for (int i = 0; i < 32; ++i) {  // fully unrolled: the slot index is a constant, the arrays stay in registers
    emit(i, rqa[i % GDEPTH], eqa[i % GDEPTH]);
    if (i + GDEPTH < 32) { rqa[i % GDEPTH] = fetch_r(i + GDEPTH); eqa[i % GDEPTH] = fetch_e(i + GDEPTH); }
  }
}

// ---- activation quantisation (scaled_fp8.py:162-169) --------------------------------------------------------------------------
// pass 1: absmax over the tensor (|x| as float bits: non-negative floats order like unsigned integers; NaN bits order above
// +inf, so a NaN survives -- as it does through torch's .max()).  ws[1] must be zero on entry.
__global__ __launch_bounds__(256) void fp8_absmax_kernel(const bf16_t* __restrict__ x, int64_t n8, unsigned int* __restrict__ amax) {
  uint32_t m = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const uint4 v = reinterpret_cast<const uint4*>(x)[i];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      m = max(m, (w[j] << 16) & 0x7fffffffu);
      m = max(m, w[j] & 0x7fff0000u);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
  if ((threadIdx.x & 63) == 0 && m != 0 && m > __atomic_load_n(amax, __ATOMIC_RELAXED)) atomicMax(amax, m);   // (read first: 16k atomics on one address otherwise)
}
// pass 2: scale = absmax / 448 (1 if absmax == 0) -> ws[0]; q = fp8(clamp(bf16(x / bf16(scale)), +-448)).
// v_cvt_pk_fp8_f32 converts with round-to-nearest-even into OCP e4m3fn on gfx950; the operands are bf16 values, exact in fp32.
__global__ __launch_bounds__(256) void fp8_quant_kernel(const bf16_t* __restrict__ x, uint8_t* __restrict__ out, int64_t n8,
                                                       float* __restrict__ ws, const float* __restrict__ amax_src) {
  const float absmax = *amax_src;
  const float scale = absmax > 0.f ? absmax / 448.0f : 1.0f;
  if (blockIdx.x == 0 && threadIdx.x == 0) ws[0] = scale;
  const float sdiv = rbf(scale);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    float f[8];
    unpack8(reinterpret_cast<const uint4*>(x)[i], f);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = fminf(fmaxf(rbf(f[j] / sdiv), -448.0f), 448.0f);
    uint2 o;
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w, true);
    o.x = (uint32_t)w;
    w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], w, true);
    o.y = (uint32_t)w;
    reinterpret_cast<uint2*>(out)[i] = o;
  }
}

}  // namespace
// second-generation kernel (gemm_fp8m.hip): gemm256m's stage discipline on the K = 128 fp8 MFMA; tried first on the shapes it accepts
// (-DWAN_FP8_NO_M: the A/B library libwanhip_f8k.so keeps this file's kernel on every shape)
template <int EPI, bool BIAS_ROWS>
int wan_gemm_fp8m_try(const uint8_t* Y, int64_t ldy, int64_t YM, const uint8_t* X, int64_t ldx, int64_t XN, int K, bf16_t* Out,
                      int64_t ldo, const bf16_t* bias, const float* scale_a, const float* scale_w, bool scale_vec, const bf16_t* R,
                      const bf16_t* mod, const bf16_t* e, int n_mod, int gate_idx, int64_t rows_per_batch, hipStream_t st, unsigned int* amax_out);
namespace {

template <int EPI, bool BIAS_ROWS>
int launch_fp8(const uint8_t* Y, int64_t ldy, int64_t YM, const uint8_t* X, int64_t ldx, int64_t XN, int K, bf16_t* Out,
               int64_t ldo, const bf16_t* bias, const float* scale_a, const float* scale_w, bool scale_vec, const bf16_t* R,
               const bf16_t* mod, const bf16_t* e, int n_mod, int gate_idx, int64_t rows_per_batch, hipStream_t st,
               unsigned int* amax_out = nullptr) {
  // amax_out (GELU form only, wan_gemm_fp8_amax): the abs-max of the stored tensor -- inside the tile kernel's epilogue, or by the
  // streaming abs-max pass behind the older kernel (the caller gets its maximum either way)
#ifndef WAN_FP8_NO_M
  {
    const int rc = wan_gemm_fp8m_try<EPI, BIAS_ROWS>(Y, ldy, YM, X, ldx, XN, K, Out, ldo, bias, scale_a, scale_w, scale_vec, R, mod, e, n_mod,
                                                     gate_idx, rows_per_batch, st, EPI == WAN_EPI_GELU_TANH ? amax_out : nullptr);
    if (rc >= 0) return rc;
  }
#endif
  WAN_REQUIRE(256 * ldy + (int64_t)K < ((int64_t)1 << 32) && 256 * ldx + (int64_t)K < ((int64_t)1 << 32),
              "wan_gemm_fp8: row pitch exceeds the 32-bit DMA offsets of a tile");
  const int64_t ty = (YM + F_BM - 1) / F_BM, tx = (XN + F_BN - 1) / F_BN;
  WAN_REQUIRE(ty * tx < ((int64_t)1 << 31), "wan_gemm_fp8: too many tiles");
  const int group = BIAS_ROWS ? 8 : 4;
  if (scale_vec)
    hipLaunchKernelGGL((gemm_fp8_kernel<EPI, BIAS_ROWS, true>), dim3((unsigned)(ty * tx)), dim3(256), 0, st, Y, ldy, YM, X, ldx, XN, K,
                       Out, ldo, bias, scale_a, scale_w, R, mod, e, n_mod, gate_idx, rows_per_batch, (int)ty, (int)tx, group);
  else
    hipLaunchKernelGGL((gemm_fp8_kernel<EPI, BIAS_ROWS, false>), dim3((unsigned)(ty * tx)), dim3(256), 0, st, Y, ldy, YM, X, ldx, XN, K,
                       Out, ldo, bias, scale_a, scale_w, R, mod, e, n_mod, gate_idx, rows_per_batch, (int)ty, (int)tx, group);
  WAN_LAUNCH_CHECK();
  if (amax_out != nullptr && EPI == WAN_EPI_GELU_TANH) {   // this kernel has no abs-max epilogue: one streaming pass over its (contiguous) output
    WAN_REQUIRE(ldo == XN && (YM * XN) % 8 == 0, "wan_gemm_fp8_amax: the output must be contiguous rows of a multiple of 8 elements");
    const int64_t n8 = YM * XN / 8;
    hipLaunchKernelGGL(fp8_absmax_kernel, dim3((unsigned)min((int64_t)4096, (n8 + 255) / 256)), dim3(256), 0, st, (const bf16_t*)Out, n8, amax_out);
    WAN_LAUNCH_CHECK();
  }
  return 0;
}

}  // namespace

extern "C" int wan_fp8_quantize(const wan_bf16* x, uint8_t* out, float* ws, int64_t n, void* stream) {
  WAN_REQUIRE(x && out && ws, "wan_fp8_quantize: null argument");
  WAN_REQUIRE(n > 0 && n % 8 == 0, "wan_fp8_quantize: n=%lld must be a positive multiple of 8", (long long)n);
  WAN_REQUIRE((((uintptr_t)x) & 15) == 0 && (((uintptr_t)out) & 7) == 0, "wan_fp8_quantize: x must be 16-byte, out 8-byte aligned");
  hipStream_t st = as_stream(stream);
  WAN_CHECK_HIP(hipMemsetAsync(ws + 1, 0, sizeof(float), st));
  const int64_t n8 = n / 8;
  const int blocks = (int)min((int64_t)4096, (n8 + 255) / 256);
  hipLaunchKernelGGL(fp8_absmax_kernel, dim3(blocks), dim3(256), 0, st, x, n8, reinterpret_cast<unsigned int*>(ws + 1));
  WAN_LAUNCH_CHECK();
  hipLaunchKernelGGL(fp8_quant_kernel, dim3(blocks), dim3(256), 0, st, x, out, n8, ws, (const float*)(ws + 1));
  WAN_LAUNCH_CHECK();
  return 0;
}

// The second half alone (round 5): ws[amax_word] already holds the tensor's abs-max as float bits -- accumulated with atomicMax by the
// kernel that PRODUCED x (wan_ln_modulate_amax / wan_ln_affine_amax: word 1 of the stream's slot; wan_gemm_fp8_amax's GELU epilogue: word
// 2) into a word the caller zeroed before that kernel ran.  The same maximum, hence the same scale and the same bytes as
// wan_fp8_quantize; 3 instead of 5 bytes per element move.
extern "C" int wan_fp8_quantize_pre(const wan_bf16* x, uint8_t* out, float* ws, int64_t n, int amax_word, void* stream) {
  WAN_REQUIRE(x && out && ws, "wan_fp8_quantize_pre: null argument");
  WAN_REQUIRE(n > 0 && n % 8 == 0 && amax_word >= 1 && amax_word < 64, "wan_fp8_quantize_pre: n=%lld must be a positive multiple of 8, amax_word=%d in [1, 64)",
              (long long)n, amax_word);
  WAN_REQUIRE((((uintptr_t)x) & 15) == 0 && (((uintptr_t)out) & 7) == 0, "wan_fp8_quantize_pre: x must be 16-byte, out 8-byte aligned");
  const int64_t n8 = n / 8;
  const int blocks = (int)min((int64_t)4096, (n8 + 255) / 256);
  hipLaunchKernelGGL(fp8_quant_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), x, out, n8, ws, (const float*)(ws + amax_word));
  WAN_LAUNCH_CHECK();
  return 0;
}

static int gemm_fp8_entry(const uint8_t* A, int64_t lda, const float* scale_a, const uint8_t* W, const float* w_scale,
                            int w_scale_n, const wan_bf16* bias, wan_bf16* C, int64_t ldc, int64_t M, int N, int K,
                            int epilogue, const wan_bf16* R, const wan_bf16* mod, const wan_bf16* e, int n_mod, int gate_idx,
                            int64_t rows_per_batch, void* stream, unsigned int* amax_out) {
  WAN_REQUIRE(A && W && C && scale_a && w_scale, "wan_gemm_fp8: null operand");
  WAN_REQUIRE(K > 0 && K % F_BK == 0, "wan_gemm_fp8: K=%d must be a positive multiple of %d", K, F_BK);
  WAN_REQUIRE(w_scale_n == 1 || w_scale_n == N, "wan_gemm_fp8: weight scale must be a scalar or one value per output row (%d for N=%d)",
              w_scale_n, N);
  WAN_REQUIRE(lda % 16 == 0 && ldc % 8 == 0, "wan_gemm_fp8: lda must be a multiple of 16 bytes, ldc of 8 elements");
  WAN_REQUIRE((((uintptr_t)A | (uintptr_t)W | (uintptr_t)C) & 15) == 0, "wan_gemm_fp8: pointers must be 16-B aligned");
  if (M == 0 || N == 0) return 0;
  hipStream_t st = as_stream(stream);
  const bool sv = w_scale_n != 1;
  switch (epilogue) {
    case WAN_EPI_NONE:
      WAN_REQUIRE(N % 16 == 0, "wan_gemm_fp8: N=%d must be a multiple of 16", N);
      return launch_fp8<WAN_EPI_NONE, false>(A, lda, M, W, K, N, K, C, ldc, bias, scale_a, w_scale, sv, nullptr, nullptr, nullptr, 0, -1, 1, st);
    case WAN_EPI_GELU_TANH:
      WAN_REQUIRE(N % 16 == 0, "wan_gemm_fp8: N=%d must be a multiple of 16", N);
      return launch_fp8<WAN_EPI_GELU_TANH, false>(A, lda, M, W, K, N, K, C, ldc, bias, scale_a, w_scale, sv, nullptr, nullptr, nullptr, 0, -1, 1, st,
                                                  amax_out);
    case WAN_EPI_GATE_RES:
      WAN_REQUIRE(N % 16 == 0, "wan_gemm_fp8: N=%d must be a multiple of 16", N);
      WAN_REQUIRE(R != nullptr, "wan_gemm_fp8: GATE_RES needs the residual R");
      WAN_REQUIRE(gate_idx < 0 || (mod && e && gate_idx < n_mod && rows_per_batch > 0), "wan_gemm_fp8: bad gate args");
      return launch_fp8<WAN_EPI_GATE_RES, false>(A, lda, M, W, K, N, K, C, ldc, bias, scale_a, w_scale, sv, R, mod, e, n_mod, gate_idx,
                                                 rows_per_batch > 0 ? rows_per_batch : 1, st);
    case WAN_EPI_TRANSPOSED:  // Ct[N, ldc]: weights are the y operand; bias and the per-row scale run along output rows
      return launch_fp8<WAN_EPI_NONE, true>(W, K, N, A, lda, M, K, C, ldc, bias, scale_a, w_scale, sv, nullptr, nullptr, nullptr, 0, -1, 1, st);
    default:
      WAN_REQUIRE(false, "wan_gemm_fp8: unknown epilogue %d", epilogue);
  }
  return 0;
}

extern "C" int wan_gemm_fp8(const uint8_t* A, int64_t lda, const float* scale_a, const uint8_t* W, const float* w_scale,
                            int w_scale_n, const wan_bf16* bias, wan_bf16* C, int64_t ldc, int64_t M, int N, int K,
                            int epilogue, const wan_bf16* R, const wan_bf16* mod, const wan_bf16* e, int n_mod, int gate_idx,
                            int64_t rows_per_batch, void* stream) {
  return gemm_fp8_entry(A, lda, scale_a, W, w_scale, w_scale_n, bias, C, ldc, M, N, K, epilogue, R, mod, e, n_mod, gate_idx, rows_per_batch, stream, nullptr);
}

// wan_gemm_fp8 with WAN_EPI_GELU_TANH that also leaves max|C| (float bits of the bf16 values, atomicMax) in *amax -- a word the caller
// zeroed; what the NEXT Linear's wan_fp8_quantize_pre reads (ffn.0 -> ffn.2: the largest tensor of a block is never re-read for its maximum)
extern "C" int wan_gemm_fp8_amax(const uint8_t* A, int64_t lda, const float* scale_a, const uint8_t* W, const float* w_scale, int w_scale_n,
                                 const wan_bf16* bias, wan_bf16* C, int64_t M, int N, int K, float* amax, void* stream) {
  WAN_REQUIRE(amax != nullptr, "wan_gemm_fp8_amax: null amax word");
  return gemm_fp8_entry(A, lda, scale_a, W, w_scale, w_scale_n, bias, C, N, M, N, K, WAN_EPI_GELU_TANH, nullptr, nullptr, nullptr, 0, -1, 1, stream,
                        reinterpret_cast<unsigned int*>(amax));
}
