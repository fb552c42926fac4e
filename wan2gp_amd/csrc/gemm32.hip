// bf16 / fp16 NT GEMM for gfx950, second generation:  Out[y][x] = epilogue( sum_k Y[y][k] * X[x][k] + bias )
// (same contract and epilogues as gemm_bf16.hip; that kernel remains the path for small problems).
//
// Built on what the attention work measured about the CU: the SIMD's issue port is the scarce
// resource (~4 non-MFMA instructions ride free under one 32-cycle v_mfma_f32_32x32x16, an LDS-DMA instruction costs
// 25-60 cycles of issue, a taken branch ~100), two independent 4-wave workgroups per CU cover each other's barrier
// bubbles.  Hence:
//   * MFMA 32x32x16 (half the MFMA instructions of 16x16x32 for the same flops) on a 128(y) x 64(x) wave tile:
//     0.75 fragment reads and 0.375 DMA pieces per MFMA (gemm_bf16.hip: 0.5 and 0.25 per half-sized MFMA);
//   * workgroup tile 256(y) x 128(x) x 32(k), 4 waves in 2x2, 24 KB per stage, 3-deep ring = 72 KB: two workgroups
//     per CU; ONE s_waitcnt vmcnt(0) + s_barrier per k-tile, the DMA of tile t+2 issued right after it, so tile t+1 is
//     already visible and its first fragments are read under the last MFMAs of tile t (no exposed LDS latency);
//   * DMA as buffer_load_dwordx4 .. offen lds: loop-invariant per-lane offsets (ragged rows clamped once), the k
//     position advances in the descriptor base (2 SALU per operand per tile);
//   * the k loop is unrolled by 3 (static ring stages): one taken branch per 3 k-tiles.
//
// LDS images: rows of 32 k = 64 B = 4 chunks of 16 B; the DMA writes lane-linearly, so the bank swizzle is applied
// to the per-lane SOURCE chunk: physical chunk p of row r holds logical chunk p ^ ((r>>2)&3); a ds_read_b128 lane
// group (16 lanes = 16 rows, e.g. rows 0-3,12-15,20-27) then covers all 16 16-B slots of a 256-B bank sweep
// (rows with equal r&3 differ in (r>>2)&3 inside every lane group; tests/test_kernel_index_emulation.py).
//
// X is the FIRST MFMA operand (MFMA rows = x).  MFMA C layout: lane (j = lane&31, h = lane>>5) holds column j (= y)
// and rows i = (r&3) + 8(r>>2) + 4h, r = 0..15.  X rows are staged permuted -- LDS row rho of a 32-row x tile holds
// x row 16*((rho>>2)&1) + (rho&3) + 4*(rho>>3) -- so that a lane's 16 accumulators are the 16 CONSECUTIVE x
// 16h .. 16h+15: the fused epilogues read bias / residual / gate and write 2 x 16 B per lane with no LDS transpose.
#include <stdlib.h>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 g32_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 g32_f16x8;
typedef __attribute__((address_space(3))) const char g32_lds_cchar;
typedef uint32_t g32_u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const g32_u4 g32_lds_u4;

constexpr int G_BM = 256, G_BN = 128, G_BK = 32;
constexpr int G_YST = G_BM * G_BK * 2;  // 16 KiB: Y image per stage
constexpr int G_XST = G_BN * G_BK * 2;  // 8 KiB:  X image per stage
constexpr int G_NST = 3;
constexpr int G_XBASE = G_NST * G_YST;  // LDS: [Y st0][Y st1][Y st2][X st0][X st1][X st2]

template <bool F16>
__device__ __forceinline__ f32x16 mfma32(const g32_u4& a, const g32_u4& b, f32x16 c) {
  if (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(g32_f16x8, a), __builtin_bit_cast(g32_f16x8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(g32_bf16x8, a), __builtin_bit_cast(g32_bf16x8, b), c, 0, 0, 0);
}

// torch GELU(approximate='tanh'): 0.5 x (1 + tanh(u)), u = sqrt(2/pi) (x + 0.044715 x^3)  ==  x / (1 + exp(-2u)):
// one v_exp_f32 + one v_rcp_f32 + 5 plain VALU instead of tanhf's ~25 instructions (the result is rounded to bf16)
__device__ __forceinline__ float g32_gelu_tanh(float x) {
  const float c = -2.0f * 0.7978845608028654f * 1.4426950408889634f;  // -2 sqrt(2/pi) log2(e)
  const float t = __builtin_fmaf(x * x, 0.044715f, 1.0f);
  const float ex = __builtin_amdgcn_exp2f(x * t * c);
  return x * __builtin_amdgcn_rcpf(1.0f + ex);
}

struct G32Frags {
  g32_u4 y[4], x[2];
};

// LDS-DMA piece as inline asm: hipcc puts an s_waitcnt vmcnt(0) in front of every ds_read that follows a builtin
// LDS-DMA to the same array (it cannot tell the ring slots apart), which would serialise the prefetch; the asm form is
// invisible to its waitcnt pass -- completion is counted by hand (vmcnt(0) + barrier at the top of each k-tile).
__device__ __forceinline__ void g32_dma16(uint32_t voff, const g32_u4& rsrc, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rsrc), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ g32_u4 g32_rsrc(const void* base) {
  const uint64_t b = (uint64_t)base;
  g32_u4 r;
  r[0] = (uint32_t)b;
  r[1] = (uint32_t)(b >> 32) & 0xffffu;  // stride 0
  r[2] = 0xffffffffu;                    // num_records: no range check needed (rows are clamped)
  r[3] = 0x00020000u;
  return r;
}

// DIST = 1: tile t+1 has landed when tile t starts (its first fragments are prefetched under tile t's last MFMAs), the
//           DMA has one k-tile of flight time;  DIST = 2: two k-tiles of flight time (counted vmcnt), fragments of a
//           tile are read after its own barrier.
template <int EPI, bool BIAS_ROWS, bool F16, int DIST>
__global__ __launch_bounds__(256, 2) void gemm32_kernel(const bf16_t* __restrict__ Y, int64_t ldy, int64_t YM,
                                                       const bf16_t* __restrict__ X, int64_t ldx, int64_t XN, int K,
                                                       bf16_t* __restrict__ Out, int64_t ldo, const bf16_t* __restrict__ bias,
                                                       const bf16_t* __restrict__ R, const bf16_t* __restrict__ mod,
                                                       const bf16_t* __restrict__ e, int n_mod, int gate_idx,
                                                       int64_t rows_per_batch, int tiles_y, int tiles_x, float out_scale) {
  __shared__ __attribute__((aligned(16))) char smem[G_NST * (G_YST + G_XST)];  // 72 KiB
  g32_lds_cchar* lds = (g32_lds_cchar*)smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wy = wave >> 1, wx = wave & 1;
  const int l31 = lane & 31, half = lane >> 5;

  // ---- tile assignment: XCD-contiguous ids, then grouped (8 y-tiles per group) ordering -----------------------
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
  const int64_t y0 = (int64_t)ty * G_BM;
  const int64_t x0 = (int64_t)tx * G_BN;

  // ---- DMA plan: loop-invariant per-lane byte offsets relative to the tile's first row ---------------------------
  // descriptor bases: Y + y0*ldy (+ k), X + x0*ldx (+ k); offsets fit 32 bits (checked by the launcher)
  uint32_t yofs[4], xofs[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = i * 256 + tid;  // 16-B slot of the Y image
    const int row = q >> 2, pch = q & 3;
    const int lch = pch ^ ((row >> 2) & 3);
    int64_t yr = y0 + row;
    if (yr > YM - 1) yr = YM - 1;  // ragged tile: re-read the last row (its results are never stored)
    yofs[i] = (uint32_t)((yr - y0) * ldy * 2 + lch * 16);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int q = i * 256 + tid;  // 16-B slot of the X image
    const int row = q >> 2, pch = q & 3;
    const int lch = pch ^ ((row >> 2) & 3);
    const int slab = row >> 6, xt = (row >> 5) & 1, rho = row & 31;
    int64_t xr = x0 + slab * 64 + xt * 32 + 16 * ((rho >> 2) & 1) + (rho & 3) + 4 * (rho >> 3);
    if (xr > XN - 1) xr = XN - 1;
    xofs[i] = (uint32_t)((xr - x0) * ldx * 2 + lch * 16);
  }
  const char* ybase = reinterpret_cast<const char*>(Y + y0 * ldy);
  const char* xbase = reinterpret_cast<const char*>(X + x0 * ldx);
  const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;  // LDS byte address of the ring
  auto stage = [&](int s) {  // fetch the k-tile at ybase / xbase into ring slot s, then advance the stream
    const g32_u4 ry = g32_rsrc(ybase), rx = g32_rsrc(xbase);
#pragma unroll
    for (int i = 0; i < 4; ++i) g32_dma16(yofs[i], ry, smem_lds + s * G_YST + (i * 256 + wave * 64) * 16);
#pragma unroll
    for (int i = 0; i < 2; ++i) g32_dma16(xofs[i], rx, smem_lds + G_XBASE + s * G_XST + (i * 256 + wave * 64) * 16);
    ybase += G_BK * 2;
    xbase += G_BK * 2;
  };

  f32x16 acc[4][2];  // [y tile][x tile]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // ---- fragment addresses: k-step s reads logical chunk 2s + half; (row>>2)&3 == (l31>>2)&3 for every tile ----------
  const int sw = (l31 >> 2) & 3;
  const int yaddr0 = (wy * 128 + l31) * 64 + ((half ^ sw) << 4);           // k-step 0; k-step 1 = ^ 32
  const int xaddr0 = G_XBASE + (wx * 64 + l31) * 64 + ((half ^ sw) << 4);
  auto load_frags = [&](G32Frags& f, int s, int ks) {  // ring slot s (compile-time after unrolling), k-step ks
#pragma unroll
    for (int t = 0; t < 4; ++t) f.y[t] = *(g32_lds_u4*)(lds + (s * G_YST + t * 2048) + (yaddr0 ^ (ks << 5)));
#pragma unroll
    for (int t = 0; t < 2; ++t) f.x[t] = *(g32_lds_u4*)(lds + (s * G_XST + t * 2048) + (xaddr0 ^ (ks << 5)));
  };
  auto mma = [&](const G32Frags& f) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = mfma32<F16>(f.x[b], f.y[a], acc[a][b]);
  };

  const int nk = K / G_BK;
#ifdef G32_TIMING
  uint64_t stamp[6] = {};
#endif
  stage(0);
  if (nk > 1) stage(1);
  G32Frags f0, f1;
  if (DIST == 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (nk > 2) stage(2);
    load_frags(f0, 0, 0);
    // k-tile kt in ring slot J: [k-step 0 MFMAs || k-step 1 fragments] [k-step 1 MFMAs || next tile's k-step 0 fragments]
#ifdef G32_TIMING
#define G32_STAMP(I) if (kt + (J0) == 60) stamp[I] = __builtin_amdgcn_s_memtime();
#else
#define G32_STAMP(I)
#endif
#define G32_STEP(J)                                                                          \
  if (__builtin_expect(kt + (J) < nk, 1)) {                                                  \
    constexpr int J0 = (J); (void)J0;                                                        \
    G32_STAMP(0)                                                                             \
    if (kt + (J) > 0) {                                                                      \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); /* tile kt+J+1 landed (issued a tile ago); this wave's fragment reads returned (round 6: vae_conv_halo.hip's race) */ \
      G32_STAMP(1)                                                                           \
      __builtin_amdgcn_s_barrier();                                                          \
      asm volatile("" ::: "memory");                                                         \
      G32_STAMP(2)                                                                           \
      if (kt + (J) + 2 < nk) stage(((J) + 2) % 3);                                           \
    }                                                                                        \
    G32_STAMP(3)                                                                             \
    load_frags(f1, (J), 1);                                                                  \
    __builtin_amdgcn_sched_barrier(0); /* reads first: they land under the 8 MFMAs below */  \
    mma(f0);                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    G32_STAMP(4)                                                                             \
    load_frags(f0, ((J) + 1) % 3, 0); /* stale but harmless data after the last tile */      \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    mma(f1);                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    G32_STAMP(5)                                                                             \
  }
    for (int kt = 0; kt < nk; kt += 3) {
      G32_STEP(0)
      G32_STEP(1)
      G32_STEP(2)
    }
#undef G32_STEP
  } else {
    // tile kt has landed once at most the younger tile's 6 pieces per lane are still in flight
#define G32_STEP2(J)                                                                         \
  if (__builtin_expect(kt + (J) < nk, 1)) {                                                  \
    if (kt + (J) + 1 < nk) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");       \
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                         \
    __builtin_amdgcn_s_barrier();                                                            \
    asm volatile("" ::: "memory");                                                           \
    if (kt + (J) + 2 < nk) stage(((J) + 2) % 3); /* slot of tile kt+J-1: every wave is past it */ \
    load_frags(f0, (J), 0);                                                                  \
    load_frags(f1, (J), 1);                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    mma(f0);                                                                                 \
    mma(f1);                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                       \
  }
    for (int kt = 0; kt < nk; kt += 3) {
      G32_STEP2(0)
      G32_STEP2(1)
      G32_STEP2(2)
    }
#undef G32_STEP2
  }

#ifdef G32_TIMING
  if (blockIdx.x == 0 && tid == 0) {  // tuning aid: s_memtime stamps of k-tile 60 -> row 0 of tile (0,0), whose epilogue is skipped
    uint64_t* dbg = reinterpret_cast<uint64_t*>(Out);
    for (int i = 0; i < 6; ++i) dbg[i] = stamp[i];
  }
  if (blockIdx.x == 0) return;
#endif
  // ---- epilogue: lane holds, for y row (yt, l31), the 16 consecutive x  xb .. xb+15 of x tile xt ----------------------
#pragma unroll
  for (int xt = 0; xt < 2; ++xt) {
    const int64_t xb = x0 + wx * 64 + xt * 32 + half * 16;
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
      const int64_t yr = y0 + wy * 128 + yt * 32 + l31;
      if (yr >= YM) continue;
      float v[16];
      const float brow = (BIAS_ROWS && bias != nullptr) ? ld16<F16>(bias[yr]) : 0.f;
      const f32x16 av = acc[yt][xt];
#pragma unroll
      for (int r = 0; r < 16; ++r)
        v[r] = rnd16<F16>(av[r] * out_scale + (BIAS_ROWS ? brow : bcol[r]));  // nn.Linear output is a 16-bit tensor
      bf16_t* optr = Out + yr * ldo + xb;
      if (xb + 16 <= XN) {
        if (EPI == WAN_EPI_GELU_TANH) {
#pragma unroll
          for (int j = 0; j < 16; ++j) v[j] = g32_gelu_tanh(v[j]);
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
      } else if (EPI == WAN_EPI_NONE) {
        // ragged x edge: only the transposed / V^T form (x = tokens, EPI NONE) can hit it -- the launcher requires
        // N % 16 == 0 for every other epilogue.  Fully unrolled: no runtime index into v[].
#pragma unroll
        for (int j = 0; j < 16; ++j)
          if (xb + j < XN) optr[j] = st16<F16>(v[j]);
      }
    }
  }
}

}  // namespace

// Returns -1 when the problem does not fit this kernel (the caller falls back to gemm_bf16.hip), else the launch status.
template <int EPI, bool BIAS_ROWS, bool F16>
int wan_gemm32_try(const bf16_t* Y, int64_t ldy, int64_t YM, const bf16_t* X, int64_t ldx, int64_t XN, int K, bf16_t* Out,
                   int64_t ldo, const bf16_t* bias, const bf16_t* R, const bf16_t* mod, const bf16_t* e, int n_mod,
                   int gate_idx, int64_t rows_per_batch, hipStream_t st, float out_scale) {
  if (K % G_BK != 0) return -1;
  // 32-bit DMA offsets: a tile's rows (256 / 128) times the row pitch in bytes, plus the row itself
  if (256 * ldy * 2 + (int64_t)K * 2 >= ((int64_t)1 << 32) || 128 * ldx * 2 + (int64_t)K * 2 >= ((int64_t)1 << 32)) return -1;
  const int64_t ty = (YM + G_BM - 1) / G_BM, tx = (XN + G_BN - 1) / G_BN;
  if (ty * tx >= ((int64_t)1 << 31)) return -1;
  static const int dist = [] { const char* e = getenv("WAN_GEMM32_DIST"); return e ? atoi(e) : 1; }();
  if (dist == 2)
    hipLaunchKernelGGL((gemm32_kernel<EPI, BIAS_ROWS, F16, 2>), dim3((unsigned)(ty * tx)), dim3(256), 0, st, Y, ldy, YM, X, ldx,
                       XN, K, Out, ldo, bias, R, mod, e, n_mod, gate_idx, rows_per_batch, (int)ty, (int)tx, out_scale);
  else
    hipLaunchKernelGGL((gemm32_kernel<EPI, BIAS_ROWS, F16, 1>), dim3((unsigned)(ty * tx)), dim3(256), 0, st, Y, ldy, YM, X, ldx,
                       XN, K, Out, ldo, bias, R, mod, e, n_mod, gate_idx, rows_per_batch, (int)ty, (int)tx, out_scale);
  WAN_LAUNCH_CHECK();
  return 0;
}

#define G32_INST(EPI, BR, F)                                                                                              \
  template int wan_gemm32_try<EPI, BR, F>(const bf16_t*, int64_t, int64_t, const bf16_t*, int64_t, int64_t, int, bf16_t*, \
                                          int64_t, const bf16_t*, const bf16_t*, const bf16_t*, const bf16_t*, int, int,  \
                                          int64_t, hipStream_t, float);
G32_INST(WAN_EPI_NONE, false, false)
G32_INST(WAN_EPI_GELU_TANH, false, false)
G32_INST(WAN_EPI_GATE_RES, false, false)
G32_INST(WAN_EPI_NONE, true, false)
G32_INST(WAN_EPI_NONE, false, true)
G32_INST(WAN_EPI_NONE, true, true)
#undef G32_INST
