// bf16 / fp16 NT GEMM for gfx950, third generation:  Out[y][x] = epilogue( sum_k Y[y][k] * X[x][k] + bias )
// (same contract and epilogues as gemm_bf16.hip / gemm256.hip; the launcher falls back to those for small problems).
//
// gemm32.hip's s_memtime stamps put 720 of the 1500 cycles of a k-tile into the LDS-DMA (issue 70 cycles per 1-KB piece,
// then waiting for it): the 256x128 tile moves 24 KB per 2.1 MFLOP through the texture path.  This kernel is the
// structure of the attention kernel (attention_w64q.hip) applied to GEMM:
//   * workgroup tile 256(y) x 256(x) x 32(k): 32 KB per 4.2 MFLOP (-33 % bytes per flop);
//   * 4 waves = ONE wave per SIMD, wave tile 128 x 128 = 4 x 4 MFMA 32x32x16 tiles: the 256 accumulators live in the
//     accumulator file (inline-asm MFMAs with "+a" operands), fragments / addresses in arch VGPRs;
//   * 0.5 fragment reads and 0.25 DMA pieces per MFMA, placed BETWEEN the MFMAs (one DMA piece per 4 MFMAs, one
//     ds_read_b128 per 2) with the order pinned by sched_barrier(0): there is no VALU work in the main loop at all, so the
//     ~4 free issue slots under each 32-cycle MFMA carry them;
//   * 4-deep LDS ring (128 KB), one counted s_waitcnt vmcnt(8) + s_barrier per k-tile (32 MFMAs = 1024 cycles per wave):
//     tile t+1 is visible when tile t starts (its first fragments are read under tile t's last MFMAs) while tiles t+2
//     and t+3 are in flight -- with ONE tile (32 KB) in flight the kernel ran at one k-tile per ~2070 cycles = the
//     memory latency (tools/probes/dma_probe: the DMA instruction itself costs ~4 cycles when the memory system keeps
//     up; the 70 cycles per piece seen in the GEMMs are queue back-pressure); k loop unrolled by 4.
// LDS images, swizzle and the X-row permutation that makes a lane's 16 accumulators 16 consecutive x are those of
// gemm32.hip (rows of 32 k = 64 B, physical chunk p of row r holds logical chunk p ^ ((r>>2)&3)).
#include <stdlib.h>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 g256_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 g256_f16x8;
typedef __attribute__((address_space(3))) const char g256_lds_cchar;
typedef uint32_t g256_u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const g256_u4 g256_lds_u4;

constexpr int G_BM = 256, G_BN = 256, G_BK = 32;
constexpr int G_YST = G_BM * G_BK * 2;  // 16 KiB: Y image per stage
constexpr int G_XST = G_BN * G_BK * 2;  // 16 KiB: X image per stage
#ifndef G256_NST
#define G256_NST 4
#endif
constexpr int G_NST = G256_NST;  // LDS ring depth: tiles t+2 .. t+NST-1 in flight
constexpr int G_XBASE = G_NST * G_YST;  // LDS: [Y st0][Y st1][Y st2][X st0][X st1][X st2]

__device__ __forceinline__ float g256_gelu_tanh(float x) {
  const float c = -2.0f * 0.7978845608028654f * 1.4426950408889634f;  // -2 sqrt(2/pi) log2(e)
  const float t = __builtin_fmaf(x * x, 0.044715f, 1.0f);
  const float ex = __builtin_amdgcn_exp2f(x * t * c);
  return x * __builtin_amdgcn_rcpf(1.0f + ex);
}

struct G256Frags {
  g256_u4 y[4], x[4];
};
// accumulators pinned to the accumulator file; operands straight from ds_read_b128 (arch VGPRs)
template <bool F16>
__device__ __forceinline__ void mfma256(f32x16& acc, const g256_u4& a, const g256_u4& b) {
  if (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

// LDS-DMA piece as inline asm: hipcc puts an s_waitcnt vmcnt(0) in front of every ds_read that follows a builtin
// LDS-DMA to the same array (it cannot tell the ring slots apart), which would serialise the prefetch; the asm form is
// invisible to its waitcnt pass -- completion is counted by hand (vmcnt(0) + barrier at the top of each k-tile).
__device__ __forceinline__ void g256_dma16(uint32_t voff, const g256_u4& rsrc, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rsrc), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ g256_u4 g256_rsrc(const void* base) {
  const uint64_t b = (uint64_t)base;
  g256_u4 r;
  r[0] = (uint32_t)b;
  r[1] = (uint32_t)(b >> 32) & 0xffffu;  // stride 0
  r[2] = 0xffffffffu;                    // num_records: no range check needed (rows are clamped)
  r[3] = 0x00020000u;
  return r;
}

template <int EPI, bool BIAS_ROWS, bool F16>
__global__ __launch_bounds__(256) void gemm256_kernel(const bf16_t* __restrict__ Y, int64_t ldy, int64_t YM,
                                                       const bf16_t* __restrict__ X, int64_t ldx, int64_t XN, int K,
                                                       bf16_t* __restrict__ Out, int64_t ldo, const bf16_t* __restrict__ bias,
                                                       const bf16_t* __restrict__ R, const bf16_t* __restrict__ mod,
                                                       const bf16_t* __restrict__ e, int n_mod, int gate_idx,
                                                       int64_t rows_per_batch, int tiles_y, int tiles_x, float out_scale, int group) {
  __shared__ __attribute__((aligned(16))) char smem[G_NST * (G_YST + G_XST)];  // 128 KiB
  g256_lds_cchar* lds = (g256_lds_cchar*)smem;
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
  // descriptor bases: Y + y0*ldy (+ k), X + x0*ldx (+ k); offsets fit 32 bits (checked by the launcher)
  uint32_t yofs[4], xofs[4];
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
  for (int i = 0; i < 4; ++i) {
    const int q = i * 256 + tid;  // 16-B slot of the X image
    const int row = q >> 2, pch = q & 3;
    const int lch = pch ^ ((row >> 2) & 3);
    const int slab = row >> 7, xt = (row >> 5) & 3, rho = row & 31;
    int64_t xr = x0 + slab * 128 + xt * 32 + 16 * ((rho >> 2) & 1) + (rho & 3) + 4 * (rho >> 3);
    if (xr > XN - 1) xr = XN - 1;
    xofs[i] = (uint32_t)((xr - x0) * ldx * 2 + lch * 16);
  }
  const char* ybase = reinterpret_cast<const char*>(Y + y0 * ldy);
  const char* xbase = reinterpret_cast<const char*>(X + x0 * ldx);
  const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;  // LDS byte address of the ring
  // one DMA piece (p = 0..3: Y, 4..7: X) of the k-tile at ybase / xbase into ring slot s; advance() after the 8th
  auto dma_piece = [&](int s, int p) {
    if (p < 4) g256_dma16(yofs[p], g256_rsrc(ybase), smem_lds + s * G_YST + (p * 256 + wave * 64) * 16);
    else g256_dma16(xofs[p - 4], g256_rsrc(xbase), smem_lds + G_XBASE + s * G_XST + ((p - 4) * 256 + wave * 64) * 16);
  };
  int kpos = 0;  // k-tile the stream points at; it stops at the last one (later fetches re-read it into a dead slot)
  auto advance = [&]() {
    const bool ok = kpos + 1 < K / G_BK;
    ybase += ok ? G_BK * 2 : 0;
    xbase += ok ? G_BK * 2 : 0;
    kpos += ok ? 1 : 0;
  };
  auto stage = [&](int s) {
#pragma unroll
    for (int p = 0; p < 8; ++p) dma_piece(s, p);
    advance();
  };

  f32x16 acc[4][4];  // [y tile][x tile], accumulator file
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
      asm volatile("" : "+a"(acc[a][b]));
    }

  // ---- fragment addresses: k-step s reads logical chunk 2s + half; (row>>2)&3 == (l31>>2)&3 for every tile ----------
  const int sw = (l31 >> 2) & 3;
  const int yaddr0 = (wy * 128 + l31) * 64 + ((half ^ sw) << 4);           // k-step 0; k-step 1 = ^ 32
  const int xaddr0 = G_XBASE + (wx * 128 + l31) * 64 + ((half ^ sw) << 4);
  // fragment r = 0..7 (0..3: Y tiles, 4..7: X tiles) of k-step ks from ring slot s (all compile-time after unrolling)
  auto load_frag = [&](G256Frags& f, int s, int ks, int r) {
    if (r < 4) f.y[r] = *(g256_lds_u4*)(lds + (s * G_YST + r * 2048) + (yaddr0 ^ (ks << 5)));
    else f.x[r - 4] = *(g256_lds_u4*)(lds + (s * G_XST + (r - 4) * 2048) + (xaddr0 ^ (ks << 5)));
  };

  // ---- gated-residual epilogue operands: loaded one x tile ahead ---------------------------------------------------------
  // One wave per SIMD: nothing covers the latency of these loads if they are issued where they are used (measured:
  // the o / ffn2 projections lost 10-30 % to it, depending on the box's memory latency).
  uint4 rraw[2][4][2], eraw[2][4][2], mraw[2][2];  // [x tile parity][y tile][half]; requested one x tile ahead
  uint32_t brow_idx[4];                            // batch (stream) of each of the lane's four y rows, for the gate lookup
  if (EPI == WAN_EPI_GATE_RES) {
#pragma unroll
    for (int yt = 0; yt < 4; ++yt) {
      const int64_t yr = y0 + wy * 128 + yt * 32 + l31;
      brow_idx[yt] = (uint32_t)(yr < YM ? yr : YM - 1) / (uint32_t)rows_per_batch;
    }
  }
  auto epi_loads = [&](int xt) {  // xt is a compile-time constant at every call site
    if (EPI != WAN_EPI_GATE_RES) return;
    const int64_t xb = x0 + wx * 128 + xt * 32 + half * 16;
    if (xb + 16 > XN) return;
    const int pb = xt & 1;
    if (gate_idx >= 0) {
      const bf16_t* mp = mod + (int64_t)gate_idx * XN + xb;
      mraw[pb][0] = *reinterpret_cast<const uint4*>(mp);
      mraw[pb][1] = *reinterpret_cast<const uint4*>(mp + 8);
    }
#pragma unroll
    for (int yt = 0; yt < 4; ++yt) {
      int64_t yr = y0 + wy * 128 + yt * 32 + l31;
      if (yr > YM - 1) yr = YM - 1;
      const bf16_t* rptr = R + yr * ldo + xb;
      rraw[pb][yt][0] = *reinterpret_cast<const uint4*>(rptr);
      rraw[pb][yt][1] = *reinterpret_cast<const uint4*>(rptr + 8);
      if (gate_idx >= 0) {
        const bf16_t* ep = e + ((int64_t)brow_idx[yt] * n_mod + gate_idx) * XN + xb;
        eraw[pb][yt][0] = *reinterpret_cast<const uint4*>(ep);
        eraw[pb][yt][1] = *reinterpret_cast<const uint4*>(ep + 8);
      }
    }
  };

  const int nk = K / G_BK;
  stage(0);
  stage(1);
  stage(2);
  if (G_NST == 5) stage(3);
  if (G_NST == 5) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // tiles 0 and 1 landed, the younger ones may be in flight
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  G256Frags f0, f1;
#pragma unroll
  for (int r = 0; r < 8; ++r) load_frag(f0, 0, 0, r);
  // k-tile kt in ring slot J, 32 MFMAs.  MFMA m of a k-step multiplies (y tile m>>2, x tile m&3); after MFMA 0,4,8,12 one
  // DMA piece of tile kt+3 (ring slot (J+3)%4, free since this tile's barrier), after the other of the first 12 one
  // fragment read: k-step 0 reads k-step 1's fragments, k-step 1 reads the next tile's k-step 0 fragments (slot (J+1)%4).
#ifdef G256_TIMING
  uint64_t stamp[5] = {};
#define G256_STAMP(I, J) if (kt + (J) == 60) stamp[I] = __builtin_amdgcn_s_memtime();
#else
#define G256_STAMP(I, J)
#endif
#define G256_SB() __builtin_amdgcn_sched_barrier(0)
#define G256_STEP(J)                                                                                   \
  if (__builtin_expect(kt + (J) < nk, 1)) {                                                            \
    G256_STAMP(0, J)                                                                                   \
    if (kt + (J) > 0) {                                                                                \
      if (G_NST == 5) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                                \
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); /* tile kt+J+1 landed; younger pieces may fly */ \
      G256_STAMP(1, J)                                                                                 \
      __builtin_amdgcn_s_barrier();                                                                    \
      asm volatile("" ::: "memory");                                                                   \
    }                                                                                                  \
    G256_STAMP(2, J)                                                                                   \
    _Pragma("unroll") for (int m = 0; m < 16; ++m) {                                                   \
      mfma256<F16>(acc[m >> 2][m & 3], f0.x[m & 3], f0.y[m >> 2]); G256_SB();                          \
      if ((m & 3) == 0) dma_piece(((J) + G_NST - 1) % G_NST, m >> 2);                                              \
      else if (m - (m >> 2) - 1 < 8) load_frag(f1, (J), 1, m - (m >> 2) - 1);                          \
      G256_SB();                                                                                       \
    }                                                                                                  \
    G256_STAMP(3, J)                                                                                   \
    _Pragma("unroll") for (int m = 0; m < 16; ++m) {                                                   \
      mfma256<F16>(acc[m >> 2][m & 3], f1.x[m & 3], f1.y[m >> 2]); G256_SB();                          \
      if ((m & 3) == 0) dma_piece(((J) + G_NST - 1) % G_NST, 4 + (m >> 2));                                        \
      else if (m - (m >> 2) - 1 < 8) load_frag(f0, ((J) + 1) % G_NST, 0, m - (m >> 2) - 1);                \
      G256_SB();                                                                                       \
    }                                                                                                  \
    advance();                                                                                         \
    G256_STAMP(4, J)                                                                                   \
  }
  for (int kt = 0; kt < nk; kt += G_NST) {
    G256_STEP(0)
    G256_STEP(1)
    G256_STEP(2)
    G256_STEP(3)
    if (G_NST == 5) { G256_STEP(4) }
  }
#undef G256_STEP
#undef G256_SB
  epi_loads(0);  // x tile 0's operands: requested before the accumulators are drained (kept out of the main loop: live
                 // across it they were spilled to scratch, which exposes their latency at the start instead)
#ifdef G256_TIMING
  if (blockIdx.x == 0 && tid == 0) {  // tuning aid: s_memtime stamps of k-tile 60 -> row 0 of tile (0,0), whose epilogue is skipped
    uint64_t* dbg = reinterpret_cast<uint64_t*>(Out);
    for (int i = 0; i < 5; ++i) dbg[i] = stamp[i];
  }
  if (blockIdx.x == 0) return;
#endif
  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");  // trailing DMA; last asm MFMAs -> accumulator reads of the epilogue

  // ---- epilogue: lane holds, for y row (yt, l31), the 16 consecutive x  xb .. xb+15 of x tile xt ----------------------
  // The residual / gate operands of x tile xt were requested one tile ahead (epi_loads); the batch index is a 32-bit
  // division per row (4 per lane), not a 64-bit one per group.
#pragma unroll
  for (int xt = 0; xt < 4; ++xt) {
    const int64_t xb = x0 + wx * 128 + xt * 32 + half * 16;
    const bool full = xb + 16 <= XN;
    float bcol[16];
    if (!BIAS_ROWS) {
#pragma unroll
      for (int j = 0; j < 16; ++j) bcol[j] = 0.f;
      if (bias != nullptr) {
        if (full) {
          unpack8t<F16>(*reinterpret_cast<const uint4*>(bias + xb), bcol);
          unpack8t<F16>(*reinterpret_cast<const uint4*>(bias + xb + 8), bcol + 8);
        } else {  // ragged x edge
#pragma unroll
          for (int j = 0; j < 16; ++j)
            if (xb + j < XN) bcol[j] = ld16<F16>(bias[xb + j]);
        }
      }
    }
    const int pb = xt & 1;
    if (xt + 1 < 4) epi_loads(xt + 1);  // next x tile's operands fly while this one is computed
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
      if (full) {
        if (EPI == WAN_EPI_GELU_TANH) {
#pragma unroll
          for (int j = 0; j < 16; ++j) v[j] = g256_gelu_tanh(v[j]);
        } else if (EPI == WAN_EPI_GATE_RES) {
          float rv[16];
          unpack8t<F16>(rraw[pb][yt][0], rv);
          unpack8t<F16>(rraw[pb][yt][1], rv + 8);
          if (gate_idx >= 0) {
            float mv[16], ev[16];
            unpack8t<F16>(mraw[pb][0], mv);
            unpack8t<F16>(mraw[pb][1], mv + 8);
            unpack8t<F16>(eraw[pb][yt][0], ev);
            unpack8t<F16>(eraw[pb][yt][1], ev + 8);
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

// Returns -1 when the problem does not fit this kernel (the caller falls back to gemm32.hip / gemm_bf16.hip), else the
// launch status.
template <int EPI, bool BIAS_ROWS, bool F16>
int wan_gemm256_try(const bf16_t* Y, int64_t ldy, int64_t YM, const bf16_t* X, int64_t ldx, int64_t XN, int K, bf16_t* Out,
                    int64_t ldo, const bf16_t* bias, const bf16_t* R, const bf16_t* mod, const bf16_t* e, int n_mod,
                    int gate_idx, int64_t rows_per_batch, hipStream_t st, float out_scale) {
  if (K % G_BK != 0) return -1;
  // 32-bit DMA offsets: a tile's 256 rows times the row pitch in bytes, plus the row itself
  if (256 * ldy * 2 + (int64_t)K * 2 >= ((int64_t)1 << 32) || 256 * ldx * 2 + (int64_t)K * 2 >= ((int64_t)1 << 32)) return -1;
  const int64_t ty = (YM + G_BM - 1) / G_BM, tx = (XN + G_BN - 1) / G_BN;
  if (ty * tx >= ((int64_t)1 << 31)) return -1;
  // y tiles per group of the tile order (measured: 4 beats 8 by 2-7 % on the y = tokens shapes; 16 / 32 lose 10-25 %)
  static const int group_env = [] { const char* e = getenv("WAN_GEMM_GROUP"); return e ? atoi(e) : 0; }();
  const int group = group_env > 0 ? group_env : (BIAS_ROWS ? 8 : 4);
  hipLaunchKernelGGL((gemm256_kernel<EPI, BIAS_ROWS, F16>), dim3((unsigned)(ty * tx)), dim3(256), 0, st, Y, ldy, YM, X, ldx,
                     XN, K, Out, ldo, bias, R, mod, e, n_mod, gate_idx, rows_per_batch, (int)ty, (int)tx, out_scale, group);
  WAN_LAUNCH_CHECK();
  return 0;
}

#define G256_INST(EPI, BR, F)                                                                                              \
  template int wan_gemm256_try<EPI, BR, F>(const bf16_t*, int64_t, int64_t, const bf16_t*, int64_t, int64_t, int, bf16_t*, \
                                           int64_t, const bf16_t*, const bf16_t*, const bf16_t*, const bf16_t*, int, int,  \
                                           int64_t, hipStream_t, float);
G256_INST(WAN_EPI_NONE, false, false)
G256_INST(WAN_EPI_GELU_TANH, false, false)
G256_INST(WAN_EPI_GATE_RES, false, false)
G256_INST(WAN_EPI_NONE, true, false)
G256_INST(WAN_EPI_NONE, false, true)
G256_INST(WAN_EPI_NONE, true, true)
#undef G256_INST
