// Flash attention forward for gfx950, "4 x 64, issue-balanced" kernel: the structure of attention_w64.hip (4 waves =
// one wave per SIMD, 64 q rows per wave as two 32-row q-blocks a and b, 3-deep LDS-DMA ring, asm MFMAs with pinned
// register files) rebuilt around the SIMD's ISSUE budget.
//
// Measured on MI355X (tools/probes/mfma_valu_probe.hip, s_memtime stamps of attention_w64.hip): a wave hides ~4 plain
// VALU instructions under each 32-cycle v_mfma_f32_32x32x16_bf16; every further VALU costs ~4.6 cycles, v_exp_f32 ~8.5.
// The classic online softmax needs ~4.7 VALU issue slots per score (fma, exp = 2 slots, max, add, 1/2 cvt_pk): 300+
// slots per 64-MFMA tile against a budget of 256, and they cluster in half of the tile.  Hence:
//   1. Q is pre-multiplied by scale*log2(e) once (bf16), and the running reference max m_ref enters through the
//      matrix pipe: each S sub-tile starts with one extra MFMA  S := [1 1 1 0..](kv x 16) * [hi; mid; lo; 0..](16 x q)
//      where hi + mid + lo = -m_ref exactly (three bf16 terms carry an fp32; a constant A fragment and a 4-register B
//      fragment that is rewritten only when m_ref moves), so that
//      S' = K Q~^T - m_ref  comes out ready for exp2 -- no per-score fma / subtract in the softmax stream, at the
//      price of 4 MFMAs per 64.
//   2. lazy max: m_ref is the row max of the first tile and afterwards only moves when a tile's max exceeds it by
//      more than 2^THR (P <= 2^THR, exact in bf16/fp32); the O / l rescale is a rare wave-uniform branch.
//   3. q-block b runs half a tile behind q-block a, so each of the tile's four 16-MFMA slots carries one half of one
//      block's softmax, in chunks of <= 5 instructions per MFMA gap, staggered (exp of pair j next to add/pack of pair
//      j-1) so that no instruction waits on the one before it.
//   4. V^T fragments live in the accumulator file ("a" operands, filled by ds_read_b128 directly); K fragments, S, P
//      and the -m_ref tuples in arch VGPRs.
//
//   A(t): S_a(t)  = K(t) Q_a^T - m_a   || softmax b(t-1) chunks 10..21 || LDS-DMA of tile t+2
//   B(t): O_b    += V^T(t-1) P_b^T     || softmax a(t)   chunks 0..9
//   C(t): S_b(t)  = K(t) Q_b^T - m_b   || softmax a(t)   chunks 10..21 || V^T(t) fragment reads
//   D(t): O_a    += V^T(t) P_a^T       || softmax b(t)   chunks 0..9   || K(t+1) fragment reads
//   chunks: 0-3 row max (quarters), 4 cross-half max + threshold test (+ rare rescale), 5 exp(pair 0),
//           6..20 exp(pair j) + sum/pack(pair j-1), 21 sum/pack(pair 15).
// Math, HBM layouts and LDS images are those of attention.hip (S^T = K Q^T, O^T = V^T P^T, V transposed in HBM, K rows
// bit-2/3 swapped, XOR-swizzled lane-linear LDS-DMA images).
#include <stdlib.h>
#include <string.h>

// Ring depth 3: tile t+1 visible when tile t starts, tile t+2 in flight.  Depth 4 (two tiles in flight; build with
// -DW64_NST=4) was measured neutral here (1246 vs 1250 TFLOP/s): at ~11 B/clk/CU the K/V stream is not latency-bound,
// unlike the GEMMs (gemm256.hip gained 13 % from the same change).
#ifndef W64_NST
#define W64_NST 3
#endif
#include "attn_w64_shared.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr float LAZY_THR = 8.0f;  // log2 units: P <= 256

struct QB {          // one 32-row q-block of the wave
  f32x16 accO[4];    // O^T tiles ("a")
  f32x16 s[2];       // S'^T of the current tile: [kv sub-tile]
  mfma_bf16x8 mfrag; // B fragment of the S-initialising MFMA: -m_ref as hi + mid + lo in k slots 0..2 (lanes < 32)
  float nm;          // -m_ref
  u32x4 pk[4];       // P^T of the current tile as packed bf16: [16-kv k-step]
  float l_run;       // this lane's share of the row sum (relative to m_ref)
  float mt;          // row max being reduced
  float pe0, pe1;    // exp2 of the pair whose sum / pack is still pending
  float cur0, ps;    // flat schedule: first exp2 of the pair in progress; sum of the pending pair
  float tmax;        // flat schedule: row max of S' found by chunk 4 ...
  bool need;         // ... and its wave-uniform verdict, consumed by chunk 5 one MFMA later (a branch on a fresh VALU compare stalls)
};

// O^T += V^T P^T with the V^T fragment in the accumulator file
#ifndef W64Q_VF_AGPR_KSTEPS
#define W64Q_VF_AGPR_KSTEPS 4  // V^T fragments of k-steps < this live in the accumulator file, the rest in arch VGPRs
#endif
__device__ __forceinline__ void pv_mfma(f32x16& acc, const mfma_bf16x8& v, const mfma_bf16x8& p, int c4) {
  if (c4 < W64Q_VF_AGPR_KSTEPS) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(v), "v"(p));
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(v), "v"(p));
}
// i = 0, 1: S sub-tile i := -m_ref (ones x mfrag);  i = 2..17: k-step (i-2)>>1 of sub-tile i&1
__device__ __forceinline__ void qk_stepq(QB& x, const mfma_bf16x8 (&kf)[2][8], const mfma_bf16x8 (&qf)[8],
                                         const mfma_bf16x8& kones, int i) {
  if (i < 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(x.s[i]) : "v"(kones), "v"(x.mfrag));
#ifdef W64Q_KF_AGPR
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(x.s[i & 1]) : "a"(kf[i & 1][(i - 2) >> 1]), "a"(qf[(i - 2) >> 1]));
#else
  else mfma_qk(x.s[i & 1], kf[i & 1][(i - 2) >> 1], qf[(i - 2) >> 1]);
#endif
}
__device__ __forceinline__ void set_mref(QB& q, float nm, int half) {
  q.nm = nm;
  // exact 3-term bf16 split (round to nearest at each step; every residual is exactly representable in fp32)
  const uint32_t hi = cvt_pk(nm, 0.f) & 0xffffu;
  const float r1 = nm - __uint_as_float(hi << 16);
  const uint32_t mid = cvt_pk(r1, 0.f) & 0xffffu;
  const float r2 = r1 - __uint_as_float(mid << 16);
  const uint32_t lo = cvt_pk(r2, 0.f) & 0xffffu;
  uint4 w;
  w.x = half ? 0u : (hi | (mid << 16));
  w.y = half ? 0u : lo;
  w.z = w.w = 0u;
  q.mfrag = __builtin_bit_cast(mfma_bf16x8, w);
}
// v_max3_f32 as asm: fmaxf() on values hipcc cannot see through (asm MFMA outputs) gets a canonicalising v_max_f32 per
// input -- one extra VALU per score
__device__ __forceinline__ float vmax3(float a, float b, float c) {
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// max of 8 scores and the running value m (4 instructions)
__device__ __forceinline__ float max8(const f32x16& s, int r0, float m, bool with_m) {
  const float a = vmax3(s[r0], s[r0 + 1], s[r0 + 2]);
  const float b = vmax3(s[r0 + 3], s[r0 + 4], s[r0 + 5]);
  const float c = vmax3(s[r0 + 6], s[r0 + 7], with_m ? m : s[r0 + 7]);
  return vmax3(a, b, c);
}
// reg r of s[T] <-> kv = kv0 + T*32 + (r&7) + 8*half + 16*(r>>3); kv_rem = valid kv rows from the tile's first row
__device__ __forceinline__ void mask_tail(QB& q, int kv_rem, int half) {
  if (__builtin_expect(kv_rem < KVBLK, 0)) {
    asm volatile("" ::: "memory");  // keep this rare path a real (wave-uniform) branch, out of line
    int lim = kv_rem - 8 * half;
    asm volatile("" : "+v"(lim));  // opaque inside the branch: the 32 compares below must not be hoisted into the hot path
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        q.s[T][r] = (T * 32 + (r & 7) + 16 * (r >> 3) >= lim) ? -INFINITY : q.s[T][r];
        asm volatile("" : "+v"(q.s[T][r]));
      }
  }
}

// the rare path of the lazy reference max: move m_ref to absorb tp = max(t, 0) (t on the first tile), rescale O and l
__device__ __forceinline__ void move_mref(QB& q, float t, int half, bool first) {
  asm volatile("" ::: "memory");
  const float tp = first ? t : fmaxf(t, 0.f);
  const float nm_new = q.nm - tp;
  const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(-tp);  // O and l are still zero on the first tile
#pragma unroll
  for (int T = 0; T < 2; ++T)
#pragma unroll
    for (int r = 0; r < 16; ++r) q.s[T][r] -= tp;
  set_mref(q, nm_new, half);
  q.l_run *= alpha;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {  // one O tile at a time: 16 temporaries, and the accumulator <-> VGPR copies stay in here
    asm volatile("" : "+a"(q.accO[dt]));
#pragma unroll
    for (int r = 0; r < 16; ++r) q.accO[dt][r] *= alpha;
    asm volatile("" : "+a"(q.accO[dt]));
  }
  asm volatile("" : "+v"(q.nm), "+v"(q.mfrag), "+v"(q.s[0]), "+v"(q.s[1]));
}
__device__ __forceinline__ float row_tmax(const QB& q) {  // max of S' = s - m_ref over the row's 64 kv of this tile
  float t;
  const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(q.mt), __float_as_uint(q.mt), false, false);
  asm("v_max_f32 %0, %1, %2" : "=v"(t) : "v"(__uint_as_float(sw[0])), "v"(__uint_as_float(sw[1])));
  return t;
}

// softmax chunk idx of q-block q (see the header).  Every chunk ends in an opaque asm use of what it produced: that
// pins it between the two asm MFMAs around it (LLVM would otherwise sink it into the block of its consumer).
__device__ __forceinline__ void chunk(QB& q, int idx, int kv_rem, int half, bool first) {
  if (idx < 4) {
    if (idx == 0) mask_tail(q, kv_rem, half);
    q.mt = max8(q.s[idx >> 1], (idx & 1) * 8, q.mt, idx != 0);
    asm volatile("" : "+v"(q.mt));
  } else if (idx == 4) {
    const float t = row_tmax(q);
    if (__builtin_expect(first || __any(t > LAZY_THR), 0)) move_mref(q, t, half, first);  // cold: out of line
  } else {
    const int j = idx - 5;  // pair whose exp2 is issued here (0..15); pair j-1 is summed and packed
    float n0 = 0.f, n1 = 0.f;
    if (j < 16) {
      n0 = __builtin_amdgcn_exp2f(q.s[j >> 3][(j & 7) * 2]);
      n1 = __builtin_amdgcn_exp2f(q.s[j >> 3][(j & 7) * 2 + 1]);
    }
    if (j > 0) {
      const int jp = j - 1;
      q.l_run += q.pe0 + q.pe1;
      q.pk[jp >> 2][jp & 3] = cvt_pk(q.pe0, q.pe1);
      asm volatile("" : "+v"(q.pk[jp >> 2]), "+v"(q.l_run));
    }
    if (j < 16) {
      q.pe0 = n0;
      q.pe1 = n1;
      asm volatile("" : "+v"(q.pe0), "+v"(q.pe1));
    }
  }
}

template <int ST, bool TIMING>
__device__ __forceinline__ void tile_w64q(lds_cchar* smem, const int (&kaddr)[8], const int (&vaddr)[4],
                                          const mfma_bf16x8 (&qfa)[8], const mfma_bf16x8 (&qfb)[8], mfma_bf16x8 (&kf)[2][8],
                                          mfma_bf16x8 (&vf)[4][4], const mfma_bf16x8& kones, QB& a, QB& b, int kv_rem_prev,
                                          int kv_rem, int half, bool first, bool first_prev, char* smem_rw, Dma& dma,
                                          uint64_t* stamp, bool rec) {
  constexpr int VB = ST * IMG, KN = ((ST + 1) % NST) * IMG, DST = (ST + NST - 1) % NST;
#define STAMP(K) do { if (TIMING && rec) stamp[K] = __builtin_amdgcn_s_memtime(); } while (0)
  // ---- A: S_a = -m_a + K Q_a^T (18 MFMAs)  ||  softmax b(t-1) chunks 10..21  ||  DMA of tile t+2
#pragma unroll
  for (int i = 0; i < 18; ++i) {
    qk_stepq(a, kf, qfa, kones, i); SB();
    if (i < 12) chunk(b, 10 + i, kv_rem_prev, half, first_prev);
    if (i >= 10) dma_piece_i<DST>(smem_rw, dma, ((i - 10) & 1) * 4 + ((i - 10) >> 1));  // K0 V0 K1 V1 ...
    SB();
  }
  dma_advance(dma);
  SB();
  STAMP(2);
  // ---- B: O_b += V^T(t-1) P_b(t-1)^T  ||  softmax a(t) chunks 0..9
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    pv_mfma(b.accO[i & 3], vf[i >> 2][i & 3], __builtin_bit_cast(mfma_bf16x8, b.pk[i >> 2]), i >> 2); SB();
    if (i >= 2 && i <= 11) chunk(a, i - 2, kv_rem, half, first);
    SB();
  }
  STAMP(3);
  // ---- C: S_b = -m_b + K Q_b^T (18 MFMAs)  ||  softmax a(t) chunks 10..21  ||  V^T(t) fragments (two per gap)
#pragma unroll
  for (int i = 0; i < 18; ++i) {
    qk_stepq(b, kf, qfb, kones, i); SB();
    if (i < 12) chunk(a, 10 + i, kv_rem, half, first);
    if (i >= 2 && i < 10) {
      const int x0 = (i - 2) * 2, x1 = x0 + 1;
      vf[x0 >> 2][x0 & 3] = *(lds_frag*)(smem + (VB + (x0 & 3) * 4096) + vaddr[x0 >> 2]);
      vf[x1 >> 2][x1 & 3] = *(lds_frag*)(smem + (VB + (x1 & 3) * 4096) + vaddr[x1 >> 2]);
    }
    SB();
  }
  STAMP(4);
  // ---- D: O_a += V^T(t) P_a(t)^T  ||  softmax b(t) chunks 0..9  ||  K(t+1) fragments
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    pv_mfma(a.accO[i & 3], vf[i >> 2][i & 3], __builtin_bit_cast(mfma_bf16x8, a.pk[i >> 2]), i >> 2); SB();
    if (i < 8) {
      const int x0 = i * 2, x1 = x0 + 1;
      kf[x0 >> 3][x0 & 7] = *(lds_frag*)(smem + (KN + (x0 >> 3) * 8192) + kaddr[x0 & 7]);
      kf[x1 >> 3][x1 & 7] = *(lds_frag*)(smem + (KN + (x1 >> 3) * 8192) + kaddr[x1 & 7]);
    }
    if (i >= 2 && i <= 11) chunk(b, i - 2, kv_rem, half, first);
    SB();
  }
  STAMP(5);
#undef STAMP
}

// ---- flat schedule ("w64f") -------------------------------------------------------------------------------------
// Same data flow, finer placement: ONE v_exp_f32 per MFMA gap (2 exps + 3 other VALU in one gap cost ~47 cycles against
// the MFMA's 32; 1 exp + 1-2 others fit).  The tile is 68 gaps (A 0..17, B 18..33, C 34..51, D 52..67); each q-block's
// softmax is a run of 38 single-gap chunks:
//   0-3 row max (quarters)   4 cross-half max + threshold test (+ rare rescale)
//   5+k (k = 0..31) exp2 of score k, plus for the PREVIOUS pair j = k/2 - 1:  k even: ps = p0 + p1, pack;  k odd: l += ps
//   37  sum / pack / l += of pair 15
// q-block a runs its chunks at gaps 23..60 of its own tile, q-block b at gaps 59..67 and 0..28 of the next tile: the two
// exp streams (a: gaps 28..59, b: 64..67 + 0..27) never overlap.  P pairs are packed >= 2 gaps before the PV MFMA that
// reads them (a: pair j at gap 30+2j, PV k-step c at 52+4c; b: pair j at 66+2j (mod 68), PV k-step c at 18+4c).
// LDS reads are spread (1 KB x 4 waves per read: bursts saturate the LDS pipe): V^T(t) fragment f at gap 20+2f, the K
// fragments in the order the QK^T MFMAs need them, 8 at the even gaps 52..66 and 8 at gaps 0..7 of the next tile.
constexpr int FLAT_A0 = 23;  // first chunk gap of q-block a
constexpr int FLAT_B0 = 59;  // first chunk gap of q-block b
__device__ __forceinline__ void chunkf(QB& q, int c, int kv_rem, int half, bool first) {
  if (c < 4) {
    chunk(q, c, kv_rem, half, first);
  } else if (c == 4) {
    q.tmax = row_tmax(q);
    q.need = first || __any(q.tmax > LAZY_THR);
    asm volatile("" : "+v"(q.tmax));
  } else if (c < 37) {
    const int k = c - 5;
    if (c == 5 && __builtin_expect(q.need, 0)) move_mref(q, q.tmax, half, first);  // cold: out of line
    const float e = __builtin_amdgcn_exp2f(q.s[k >> 4][k & 15]);
    if ((k & 1) == 0) {
      if (k >= 2) {
        const int j = (k >> 1) - 1;
        q.ps = q.pe0 + q.pe1;
        q.pk[j >> 2][j & 3] = cvt_pk(q.pe0, q.pe1);
        asm volatile("" : "+v"(q.pk[j >> 2]), "+v"(q.ps));
      }
      q.cur0 = e;
      asm volatile("" : "+v"(q.cur0));
    } else {
      if (k >= 3) {
        q.l_run += q.ps;
        asm volatile("" : "+v"(q.l_run));
      }
      q.pe0 = q.cur0;
      q.pe1 = e;
      asm volatile("" : "+v"(q.pe0), "+v"(q.pe1));
    }
  } else {
    q.ps = q.pe0 + q.pe1;
    q.pk[3][3] = cvt_pk(q.pe0, q.pe1);
    q.l_run += q.ps;
    asm volatile("" : "+v"(q.pk[3]), "+v"(q.l_run));
  }
}

template <int ST, bool TIMING>
__device__ __forceinline__ void tile_w64f(lds_cchar* smem, const int (&kaddr)[8], const int (&vaddr)[4],
                                          const mfma_bf16x8 (&qfa)[8], const mfma_bf16x8 (&qfb)[8], mfma_bf16x8 (&kf)[2][8],
                                          mfma_bf16x8 (&vf)[4][4], const mfma_bf16x8& kones, QB& a, QB& b, int kv_rem,
                                          int half, bool first, char* smem_rw, Dma& dma, uint64_t* stamp, bool rec) {
  constexpr int VB = ST * IMG, KN = ((ST + 1) % NST) * IMG, DST = (ST + NST - 1) % NST;
#define STAMP(K) do { if (TIMING && rec) stamp[K] = __builtin_amdgcn_s_memtime(); } while (0)
  // the non-MFMA work of global gap G
#define FLAT_GAP(G)                                                                                              \
  do {                                                                                                           \
    if ((G) <= 28) chunkf(b, (G) + 9, kv_rem, half, false);           /* q-block b, tile t-1: chunks 9..37 */     \
    if ((G) >= FLAT_A0 && (G) <= FLAT_A0 + 37) chunkf(a, (G) - FLAT_A0, kv_rem, half, first);                     \
    if ((G) >= FLAT_B0) chunkf(b, (G) - FLAT_B0, kv_rem, half, first);  /* q-block b, tile t: chunks 0..8 */      \
    if ((G) >= 20 && (G) <= 50 && (((G) - 20) & 1) == 0) {              /* V^T(t) fragment f at gap 20 + 2f */     \
      const int f = ((G) - 20) >> 1;                                                                              \
      vf[f >> 2][f & 3] = *(lds_frag*)(smem + (VB + (f & 3) * 4096) + vaddr[f >> 2]);                             \
    }                                                                                                            \
    if ((G) >= 52 && (((G) - 52) & 1) == 0) {                           /* K(t+1), need order, first 8 */         \
      const int r = ((G) - 52) >> 1, f = (r & 1) * 8 + (r >> 1);                                                  \
      kf[f >> 3][f & 7] = *(lds_frag*)(smem + (KN + (f >> 3) * 8192) + kaddr[f & 7]);                             \
    }                                                                                                            \
    if ((G) <= 7) {                                                     /* K(t) read 8..15: this tile's stage */  \
      const int r = 8 + (G), f = (r & 1) * 8 + (r >> 1);                                                          \
      kf[f >> 3][f & 7] = *(lds_frag*)(smem + (ST * IMG + (f >> 3) * 8192) + kaddr[f & 7]);                       \
    }                                                                                                            \
    if ((G) >= 3 && (G) <= 17 && (((G) - 3) & 1) == 0) {                                                          \
      const int pc = ((G) - 3) >> 1;                                    /* DMA pieces K0 V0 K1 V1 ... */          \
      dma_piece_i<DST>(smem_rw, dma, (pc & 1) * 4 + (pc >> 1));                                                   \
    }                                                                                                            \
    if (TIMING && rec && ((G) & 3) == 3) stamp[3 + ((G) >> 2)] = __builtin_amdgcn_s_memtime();                    \
  } while (0)
  if (TIMING && rec) stamp[2] = __builtin_amdgcn_s_memtime();
  // ---- A: S_a = -m_a + K Q_a^T (18 MFMAs)
#pragma unroll
  for (int i = 0; i < 18; ++i) {
    qk_stepq(a, kf, qfa, kones, i); SB();
    FLAT_GAP(i);
    SB();
  }
  dma_advance(dma);
  SB();
  // ---- B: O_b += V^T(t-1) P_b(t-1)^T
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    pv_mfma(b.accO[i & 3], vf[i >> 2][i & 3], __builtin_bit_cast(mfma_bf16x8, b.pk[i >> 2]), i >> 2); SB();
    FLAT_GAP(18 + i);
    SB();
  }
  // ---- C: S_b = -m_b + K Q_b^T (18 MFMAs)
#pragma unroll
  for (int i = 0; i < 18; ++i) {
    qk_stepq(b, kf, qfb, kones, i); SB();
    FLAT_GAP(34 + i);
    SB();
  }
  // ---- D: O_a += V^T(t) P_a(t)^T
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    pv_mfma(a.accO[i & 3], vf[i >> 2][i & 3], __builtin_bit_cast(mfma_bf16x8, a.pk[i >> 2]), i >> 2); SB();
    FLAT_GAP(52 + i);
    SB();
  }
#undef FLAT_GAP
#undef STAMP
}

// 8 bf16 -> * c -> 8 bf16 (round to nearest even)
__device__ __forceinline__ mfma_bf16x8 prescale8(const uint4 raw, float c) {
  uint4 o;
  const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
  uint32_t r[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    r[i] = cvt_pk(__uint_as_float(w[i] << 16) * c, __uint_as_float(w[i] & 0xffff0000u) * c);
  o.x = r[0]; o.y = r[1]; o.z = r[2]; o.w = r[3];
  return __builtin_bit_cast(mfma_bf16x8, o);
}

// FLAGS bit0: s_memtime stamps of tile 300 of workgroup 0 -> first 48 B of O (tuning aid)
//       bit1: q already holds q * scale * log2(e) (wan_rmsnorm_rope_scaled): skip the pre-scaling pass
//       bit2: flat one-exp-per-gap schedule (tile_w64f) instead of the half-slot chunks (tile_w64q)
template <int FLAGS>
__global__ __launch_bounds__(256) void attn_w64q_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ Kg,
                                                       const bf16_t* __restrict__ Vt, bf16_t* __restrict__ O, int B, int Bk,
                                                       int64_t Lq, int64_t Lk, int64_t ldv, int H, int nqb, float scale_log2e,
                                                       int nseg, int64_t k_seg_stride, int64_t vt_seg_stride) {
  constexpr bool TIMING = (FLAGS & 1) != 0;
  constexpr bool PRESCALED = (FLAGS & 2) != 0;
  constexpr bool FLAT = (FLAGS & 4) != 0;  // one-exp-per-gap schedule (tile_w64f)
  uint64_t stamp[20] = {};
  __shared__ __attribute__((aligned(16))) char smem[2 * NST * IMG];  // [K stages][V^T stages] = 96 KB
  lds_cchar* lds = (lds_cchar*)smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int l31 = lane & 31;

  const int total = nqb * H * B;
  const int v = xcd_remap(blockIdx.x, total);
  const int pair = v / nqb;
  const int qb = v - pair * nqb;
  const int b = pair / H, h = pair - b * H;
  const int bk = (Bk == 1) ? 0 : b;
  const int64_t rs = (int64_t)H * 128;

  const bf16_t* qbase = Q + ((int64_t)b * Lq) * rs + (int64_t)h * 128;
  const bf16_t* kbase = Kg + ((int64_t)bk * Lk) * rs + (int64_t)h * 128;
  const bf16_t* vbase = Vt + ((int64_t)bk * H * 128 + (int64_t)h * 128) * ldv;
  bf16_t* obase = O + ((int64_t)b * Lq) * rs + (int64_t)h * 128;

  const int64_t q0 = (int64_t)qb * 256 + wave * 64;
  mfma_bf16x8 qfa[8], qfb[8];  // Q~ = bf16(q * scale * log2 e)
  {
    int64_t ra = q0 + l31, rb = q0 + 32 + l31;
    if (ra > Lq - 1) ra = Lq - 1;
    if (rb > Lq - 1) rb = Lq - 1;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const uint4 wa = *reinterpret_cast<const uint4*>(qbase + ra * rs + ks * 16 + half * 8);
      const uint4 wb = *reinterpret_cast<const uint4*>(qbase + rb * rs + ks * 16 + half * 8);
      qfa[ks] = PRESCALED ? __builtin_bit_cast(mfma_bf16x8, wa) : prescale8(wa, scale_log2e);
      qfb[ks] = PRESCALED ? __builtin_bit_cast(mfma_bf16x8, wb) : prescale8(wb, scale_log2e);
      // make each fragment ONE accumulator-file tuple from here on (otherwise the allocator keeps scattered master copies
      // and assembles the operand tuple with v_accvgpr_mov before every MFMA)
      asm volatile("" : "+a"(qfa[ks]));
      asm volatile("" : "+a"(qfb[ks]));
    }
  }

  // ---- DMA stream ---------------------------------------------------------------------------------------
  const int Lk32 = (int)Lk;
  const int tps = (Lk32 + KVBLK - 1) / KVBLK;
  const int ntile = tps * nseg;
  Dma dma;
  dma_init(dma, kbase, vbase, k_seg_stride * 2, vt_seg_stride * 2, Lk32, nseg, (uint32_t)(rs * 2), (uint32_t)(ldv * 2), tid, wave);
  int cur_tt = 0;
  auto next_kv_rem = [&]() {  // valid kv rows from the start of the tile being consumed to the end of its segment
    const int rem = Lk32 - cur_tt * KVBLK;
    if (++cur_tt == tps) cur_tt = 0;
    return rem;
  };

  // ---- LDS fragment addresses: per-lane VGPR + compile-time immediates ------------------------------------
  int kaddr[8], vaddr[4];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) kaddr[ks] = l31 * 256 + (((ks * 2 + half) ^ (l31 & 15)) << 4);
#pragma unroll
  for (int c4 = 0; c4 < 4; ++c4)  // + V^T region base: keeps every ds_read offset inside the 16-bit immediate
    vaddr[c4] = NST * IMG + l31 * 128 + (((c4 * 2 + half) ^ ((l31 >> 1) & 7)) << 4);

  QB qa, qbk;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { qa.accO[dt][r] = 0.f; qbk.accO[dt][r] = 0.f; }
  set_mref(qa, 0.f, half);
  set_mref(qbk, 0.f, half);
  mfma_bf16x8 kones;  // A fragment of the S-initialising MFMA: 1.0 in k slots 0..2 (lanes < 32), zeros elsewhere
  {
    uint4 w;
    w.x = half ? 0u : 0x3f803f80u;
    w.y = half ? 0u : 0x3f80u;
    w.z = w.w = 0u;
    kones = __builtin_bit_cast(mfma_bf16x8, w);
    asm volatile("" : "+v"(kones));
  }
  qa.l_run = qbk.l_run = 0.f;
  qa.pe0 = qa.pe1 = qbk.pe0 = qbk.pe1 = 0.f;
  qa.cur0 = qa.ps = qbk.cur0 = qbk.ps = 0.f;
  qa.tmax = qbk.tmax = 0.f;
  qa.need = qbk.need = false;
  qa.mt = qbk.mt = 0.f;
  // q-block b starts half a tile behind: its first "chunks 10..21" / PV_b run on an all-masked dummy tile (P = 0)
  {
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) { qa.pk[c4] = z; qbk.pk[c4] = z; }
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
      for (int r = 0; r < 16; ++r) { qbk.s[T][r] = -INFINITY; qa.s[T][r] = -INFINITY; }
  }
  mfma_bf16x8 vf[4][4];  // V^T fragments, carried from C(t) to B(t+1)
  {
    uint4 z; z.x = z.y = z.z = z.w = 0u;
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) vf[c4][dt] = __builtin_bit_cast(mfma_bf16x8, z);
  }

  dma_tile<0>(smem, dma);
  dma_tile<1>(smem, dma);
  if (NST == 4) {
    dma_tile<2>(smem, dma);
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");  // tiles 0, 1 landed; tile 2's 8 pieces may be in flight
  } else {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  mfma_bf16x8 kf[2][8];
#pragma unroll
  for (int T = 0; T < 2; ++T)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) kf[T][ks] = *(lds_frag*)(lds + T * 8192 + kaddr[ks]);

  int kv_rem_prev = KVBLK;
#define W64Q_STEP(J)                                                                                         \
  if (__builtin_expect(t + (J) < ntile, 1)) {                                                                \
    const bool rec = TIMING && (t + (J) == 300);                                                             \
    if (TIMING && rec) stamp[0] = __builtin_amdgcn_s_memtime();                                              \
    if (t + (J) > 0) {                                                                                       \
      if (NST == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); /* tile t+J+1 landed; the next one's 8 pieces may fly */ \
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                  \
      __builtin_amdgcn_s_barrier();                                                                          \
      asm volatile("" ::: "memory");                                                                         \
    }                                                                                                        \
    if (TIMING && rec) stamp[1] = __builtin_amdgcn_s_memtime();                                              \
    const int kv_rem = next_kv_rem();                                                                        \
    if (FLAT) tile_w64f<J, TIMING>(lds, kaddr, vaddr, qfa, qfb, kf, vf, kones, qa, qbk, kv_rem, half, t + (J) == 0, smem, \
                                   dma, stamp, rec);                                                         \
    else tile_w64q<J, TIMING>(lds, kaddr, vaddr, qfa, qfb, kf, vf, kones, qa, qbk, kv_rem_prev, kv_rem, half,  \
                              t + (J) == 0, t + (J) == 1, smem, dma, stamp, rec);                            \
    kv_rem_prev = kv_rem;                                                                                    \
  }
  for (int t = 0; t < ntile; t += NST) {
    W64Q_STEP(0)
    W64Q_STEP(1)
    W64Q_STEP(2)
    if (NST == 4) { W64Q_STEP(3) }
  }
#undef W64Q_STEP
  // drain: q-block b's last tile
  if (FLAT) {
#pragma unroll
    for (int c = 9; c < 38; ++c) chunkf(qbk, c, kv_rem_prev, half, false);
  } else {
#pragma unroll
    for (int idx = 10; idx < 22; ++idx) chunk(qbk, idx, kv_rem_prev, half, ntile == 1);
  }
  asm volatile("s_nop 1" ::: "memory");
#pragma unroll
  for (int i = 0; i < 16; ++i) pv_mfma(qbk.accO[i & 3], vf[i >> 2][i & 3], __builtin_bit_cast(mfma_bf16x8, qbk.pk[i >> 2]), i >> 2);

  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");  // trailing DMA lands before O staging reuses LDS; last PV MFMAs -> accumulator reads
  // ---- epilogue: normalise, stage the wave's 64 x 128 O tile through LDS, store whole rows -------------------
  const float inva = 1.0f / (qa.l_run + __shfl_xor(qa.l_run, 32, 64));
  const float invb = 1.0f / (qbk.l_run + __shfl_xor(qbk.l_run, 32, 64));
  __syncthreads();
  char* ob = smem + wave * (64 * 256);
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    const QB& x = blk ? qbk : qa;
    const float inv = blk ? invb : inva;
    const int row = blk * 32 + l31;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 w;
        w.x = cvt_pk(x.accO[dt][g * 4 + 0] * inv, x.accO[dt][g * 4 + 1] * inv);
        w.y = cvt_pk(x.accO[dt][g * 4 + 2] * inv, x.accO[dt][g * 4 + 3] * inv);
        const int ch = (dt * 4 + g) ^ (l31 & 15);
        *reinterpret_cast<uint2*>(ob + row * 256 + ch * 16 + half * 8) = w;
      }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = i * 4 + (lane >> 4), c = lane & 15;
    const int64_t qr = q0 + r;
    if (qr < Lq) {
      const uint4 val = *reinterpret_cast<const uint4*>(ob + r * 256 + ((c ^ (r & 15)) << 4));
      *reinterpret_cast<uint4*>(obase + qr * rs + c * 8) = val;
    }
  }
  if (TIMING && blockIdx.x == 0 && tid == 0) {
    __builtin_amdgcn_s_waitcnt(0);
#pragma unroll
    for (int k6 = 0; k6 < 20; ++k6) reinterpret_cast<uint64_t*>(O)[k6] = stamp[k6];
  }
}

}  // namespace

// called from attention.hip's dispatcher.  flags bit0: s_memtime stamps (tuning aid); bit1: q is pre-scaled; bit2: flat schedule
int wan_attention_w64q_launch(int flags, const bf16_t* q, const bf16_t* k, const bf16_t* vt, bf16_t* o, int B, int Bk,
                              int64_t Lq, int64_t Lk, int64_t ldv, int H, int nseg, int64_t k_seg_stride,
                              int64_t vt_seg_stride, float scale_log2e, hipStream_t stream) {
  WAN_REQUIRE(Lk * (int64_t)H * 256 < ((int64_t)1 << 32) && ldv * 256 < ((int64_t)1 << 32),
              "wan_attention: K/V^T extent exceeds the 32-bit DMA offsets of this kernel");
  const int64_t nqb = (Lq + 255) / 256;
  const int64_t total = nqb * H * B;
  WAN_REQUIRE(total < ((int64_t)1 << 31), "wan_attention: grid too large");
#define W64Q_LAUNCH(FL)                                                                                              \
  hipLaunchKernelGGL((attn_w64q_kernel<FL>), dim3((unsigned)total), dim3(256), 0, stream, q, k, vt, o, B, Bk, Lq, Lk, \
                     ldv, H, (int)nqb, scale_log2e, nseg, k_seg_stride, vt_seg_stride)
  switch (flags & 7) {
    case 0: W64Q_LAUNCH(0); break;
    case 1: W64Q_LAUNCH(1); break;
    case 2: W64Q_LAUNCH(2); break;
    case 3: W64Q_LAUNCH(3); break;
    case 4: W64Q_LAUNCH(4); break;
    case 5: W64Q_LAUNCH(5); break;
    case 6: W64Q_LAUNCH(6); break;
    default: W64Q_LAUNCH(7); break;
  }
#undef W64Q_LAUNCH
  WAN_LAUNCH_CHECK();
  return 0;
}
