// Flash attention forward for gfx950, the BOUNDED tile loop of attention_w64q.hip on the 16x16x32 MFMA ("w16n").
// Same contract, same launch protocol (flags, workgroup flags, raw partial sums), same 4 waves x 64 q rows, same 3-deep LDS-DMA
// ring and V^T / K images in LDS -- the matrix instruction is v_mfma_f32_16x16x32_bf16 instead of v_mfma_f32_32x32x16_bf16.
//
// Why (round 3): self-attention runs at the chip's power limit.  Counters on tools/probes/mfma_power_probe.hip:
// the K = 32 form keeps the matrix pipe as busy at a 15 % higher clock (it reads and writes each accumulator half as often per
// MAC); gemm256m.hip gained 9-17 % from nothing but this change of instruction.
//
// Layout (lane: n = lane & 15, g = lane >> 4).  A wave's 64 q rows are four 16-row q tiles; halves a = q tiles 0, 1 and b = 2, 3
// play the roles of attention_w64q.hip's q-blocks (b runs half a tile behind a).
//   S^T = K Q^T    A = K fragment (kv tile kt, k-step ks): lane (n, g) holds LDS row 16 kt + n, d = 32 ks + 8 g .. + 7;
//                  B = Q fragment (q tile qt, ks): q row 16 qt + n, the same d; lane (n, g) then holds in register i of tile
//                  (kt, qt) the score of q = 16 qt + n against LDS row 16 kt + 4 g + i.
//   O^T = V^T P^T  A = V^T fragment (d tile dt, kv step c): d = 16 dt + n, k slots 8 g .. + 7 of the 32 kv of step c;
//                  B = P^T fragment (qt, c): k slots 8 g + j, j = 0..3 from S tile (2c, qt) registers 0..3, j = 4..7 from tile
//                  (2c + 1, qt).  P never moves between lanes.
//   For the P^T slots to be the 8 CONSECUTIVE kv 32 c + 8 g + j (so that a V^T fragment is one ds_read_b128 of the untouched V^T
//   image), LDS row 16 kt + m of the K image holds kv row 32 (kt >> 1) + 8 (m >> 2) + 4 (kt & 1) + (m & 3) of the tile -- a
//   permutation of the DMA source rows (attn_w64_shared.h, dma_init).  Both images keep their XOR swizzles (K: chunk ^ (row & 15),
//   V^T: chunk ^ ((row >> 1) & 7)): a ds_read_b128's four 16-lane service groups still cover the 64 banks once.
//   O tile (dt, qt) register i = O[q = 16 qt + n][d = 16 dt + 4 g + i].
// Registers: O 128 + V^T 64 + Q 64 in the accumulator file, K 64 + S 64 + P 32 in arch VGPRs -- as attention_w64q.hip.
//
// Schedule: attention_w64q.hip's bounded schedule with every 32-cycle MFMA gap split into two 16-cycle gaps (128 per tile:
// A 0..31 S_a, B 32..63 PV_b of tile t-1, C 64..95 S_b, D 96..127 PV_a).  Old gap g -> new gaps 2g (the v_exp_f32 of g) and 2g + 1
// (LDS fragment reads, the row-sum add of that score, the pack of a finished pair); five of the eight DMA pieces and the DMA
// stream's scalar step sit in the gaps behind the barrier, which carry nothing else.  Counters: 82 % matrix-pipe utilisation at
// 1.72-1.81 GHz (the 32x32x16 loop: 80 % at 1.63); a bare loop with this tile's instruction mix reaches 81.6 %.
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "attn_w64_shared.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr float BOUND16_LOG2 = 96.0f;

struct QH {           // one half of the wave's q rows: two 16-row q tiles
  f32x4 accO[8][2];   // O^T tiles [d tile][q tile] ("a")
  f32x4 s[4][2];      // S^T of the current tile [kv tile][q tile]
  u32x4 pk[4];        // P^T fragments, f = 2 c + q tile
  float p0, p1;       // the exp2 pair whose pack / sums are pending
  float l[2][2];      // this lane's share of the row sums [q tile][even / odd score] (scalars + asm v_add_f32, see attention_w64q.hip)
};

__device__ __forceinline__ void qk16_0(f32x4& d, const mfma_bf16x8& k, const mfma_bf16x8& q) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(k), "a"(q));
}
// the first k-step of an S tile when the q tile carries a reference shift: C = {-m, -m, -m, -m} of the lane's q column (a lane's four
// registers of a tile belong to ONE q row), so the tile comes out as s - m at no instruction's cost
__device__ __forceinline__ void qk16_c(f32x4& d, const mfma_bf16x8& k, const mfma_bf16x8& q, const f32x4& c) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(k), "a"(q), "v"(c));
}
__device__ __forceinline__ void qk16(f32x4& d, const mfma_bf16x8& k, const mfma_bf16x8& q) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(k), "a"(q));
}
__device__ __forceinline__ void pv16(f32x4& acc, const mfma_bf16x8& v, const mfma_bf16x8& p) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(v), "v"(p));
}
// A row-sum add: a plain add fenced by an empty asm on its result (two of them can then not be SLP-packed into a v_pk_add_f32, an
// anti-lever beside MFMAs).  NOT an asm instruction: hipcc's hazard recognizer does not count inline asm as wait states, and the pack
// that follows the add in its gap reads a transcendental's result -- behind an asm add it gets an s_nop (15 per tile), behind this none.
__device__ __forceinline__ void vadd16(float& acc, float x) {
  acc = acc + x;
  asm volatile("" : "+v"(acc));
}

// score k = 0..31 of a half, in the order its P^T fragments are needed: f = k >> 3 (= 2 c + q tile), then kv tile parity, then register
__device__ __forceinline__ float score(const QH& q, int k) {
  const int f = k >> 3;
  return q.s[2 * (f >> 1) + ((k >> 2) & 1)][f & 1][k & 3];
}
// register (kt, i) of lane group g <-> kv row 32 (kt >> 1) + 8 g + 4 (kt & 1) + i of the tile; kv_rem = valid kv rows from the tile's first
__device__ __forceinline__ void mask_tail16(QH& q, int kv_rem, int g) {
  if (__builtin_expect(kv_rem < KVBLK, 0)) {
    asm volatile("" ::: "memory");  // keep this rare path a real (wave-uniform) branch, out of line
    int lim = kv_rem - 8 * g;
    asm volatile("" : "+v"(lim));
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          q.s[kt][qt][i] = (32 * (kt >> 1) + 4 * (kt & 1) + i >= lim) ? -INFINITY : q.s[kt][qt][i];
          asm volatile("" : "+v"(q.s[kt][qt][i]));
        }
  }
}
// exp2 of score k.  The row-sum add of score k - 1 (sum16; its register is the other one of the pair) sits in the NEXT (odd) gap, in
// front of the pack: MFMA + transcendental fill an even gap's 16 cycles by themselves (measured both ways: -DW16N_ADD_EVEN puts the
// add behind the exp2).  Score 31's add is tail16.
__device__ __forceinline__ void sum16(QH& q, int k) {
  if (k < 1) return;
  if ((k & 1) == 0) vadd16(q.l[((k - 1) >> 3) & 1][1], q.p1);
  else vadd16(q.l[((k - 1) >> 3) & 1][0], q.p0);
}
__device__ __forceinline__ void exp16(QH& q, int k, int kv_rem, int g) {
  if (k == 0) mask_tail16(q, kv_rem, g);  // cold: only a segment's ragged last tile (zero-filled K rows -> s = 0 -> -inf)
  if ((k & 1) == 0) {
    q.p0 = __builtin_amdgcn_exp2f(score(q, k));
    asm volatile("" : "+v"(q.p0));
  } else {
    q.p1 = __builtin_amdgcn_exp2f(score(q, k));
    asm volatile("" : "+v"(q.p1));
  }
#ifdef W16N_ADD_EVEN
  sum16(q, k);
#endif
}
#ifdef W16N_ADD_EVEN
#define SUM16_ODD(Q, K)
#else
#define SUM16_ODD(Q, K) sum16(Q, K)
#endif
// behind exp2 of the ODD score k (at least one VALU later): pair k >> 1 -> its P^T register
__device__ __forceinline__ void pack16(QH& q, int k) {
  const int j = k >> 1;
  q.pk[j >> 2][j & 3] = cvt_pk(q.p0, q.p1);
  asm volatile("" : "+v"(q.pk[j >> 2]));
}
__device__ __forceinline__ void tail16(QH& q) { vadd16(q.l[1][1], q.p1); }  // score 31's share of its row sum (q tile 1)
// MFMA i = 0..31 of an S phase: k-step i >> 3, kv tile (i >> 1) & 3, q tile i & 1 -- an accumulator is revisited every 8 MFMAs
template <bool SHIFT>
__device__ __forceinline__ void qk_step16(QH& x, const mfma_bf16x8 (&kf)[4][4], const mfma_bf16x8 (&qf)[4][4], const f32x4 (&negm)[4], int qt0, int i) {
  const int ks = i >> 3, kt = (i >> 1) & 3, qt = i & 1;
  if (ks == 0) {
    if (SHIFT) qk16_c(x.s[kt][qt], kf[kt][0], qf[qt0 + qt][0], negm[qt0 + qt]);
    else qk16_0(x.s[kt][qt], kf[kt][0], qf[qt0 + qt][0]);
  } else {
    qk16(x.s[kt][qt], kf[kt][ks], qf[qt0 + qt][ks]);
  }
}
// MFMA i = 0..31 of a PV phase: P^T fragment f = i >> 3 (c = f >> 1, q tile f & 1), d tile i & 7
__device__ __forceinline__ void pv_step16(QH& x, const mfma_bf16x8 (&vf)[8][2], int i) {
  const int f = i >> 3, dt = i & 7;
  pv16(x.accO[dt][f & 1], vf[dt][f >> 1], __builtin_bit_cast(mfma_bf16x8, x.pk[f]));
}

// old-gap position of q-half b's exp2 of score k0 in the part of its stream that sits in ITS OWN tile (old gaps 51..63: scores 0..16,
// two in the even gaps 56..62) -- see attention_w64q.hip
__device__ __forceinline__ constexpr int bk0(int g) { return g - 51 + (g > 56 ? (g - 55) / 2 : 0); }

// The segment-walking stream (sequence parallelism: the all-gather form's segments, the Ulysses rank's `world` source ranks) as a TABLE
// walk (round 6).  dma_advance<true> (attn_w64_shared.h) tracks tile counter, segment, segment bases and the left-out segment in scalar
// registers: ~40 scalar instructions per tile more than the single-segment stream, spread over the three empty gaps behind the barrier --
// and still 2-4 % of a segmented launch against the same work on contiguous K / V^T (profiles/r06_attn_launch_by_layout_run02.log; the
// rank-of-8 trace r06_rank_world8_kernel_trace_run01.json).  The walk does not depend on the workgroup: the launcher leaves one 16-byte
// entry per fetch -- {k offset / 16, v offset / 16, valid K bytes} relative to the (batch, head) bases, the last entry repeated for the
// fetches past the end -- in device memory once per (shape, stream) (seg_table below), and a step is one s_load_dwordx4 issued a
// whole tile before its result is applied, plus two 64-bit adds: the single-segment stream's cost.
struct SegTab {
  const uint4* tab;     // entry f: the f-th tile the stream fetches
  const char* kb;       // K / V^T of this workgroup's (batch, head), segment 0
  const char* vb;
  uint32_t i;           // index of `nxt`
  uint4 nxt;            // the entry the NEXT step applies (loaded one tile ago)
};
__device__ __forceinline__ void segtab_apply_k(Dma& d, const SegTab& g) {
  d.k = g.kb + ((uint64_t)g.nxt.x << 4);
  d.klen = g.nxt.z;
}
__device__ __forceinline__ void segtab_apply_v(Dma& d, const SegTab& g) { d.v = g.vb + ((uint64_t)g.nxt.y << 4); }
__device__ __forceinline__ void segtab_load(SegTab& g) {
  g.i += 1u;
  g.nxt = g.tab[g.i];
}

template <int ST, bool MULTI, bool TIMING, bool SHIFT>
__device__ __forceinline__ void tile_w16n(lds_cchar* smem, const int (&kaddr)[4], const int (&vaddr)[2], const mfma_bf16x8 (&qf)[4][4],
                                          const f32x4 (&negm)[4], mfma_bf16x8 (&kf)[4][4], mfma_bf16x8 (&vf)[8][2], QH& a, QH& b, int kv_rem, int lg,
                                          char* smem_rw, Dma& dma, SegTab& seg, int& cur_tt, int tps, uint64_t* stamp, bool rec) {
  constexpr int VB = ST * IMG, KN = ((ST + 1) % NST) * IMG, DST = (ST + NST - 1) % NST;
  int adv_ = 0;
  uint32_t kb_ = 0;
  // the non-MFMA work of new gap G: g = G >> 1 is attention_w64q.hip's gap; the even sub-gap carries its exp2 (MFMA + transcendental
  // fill its 16 cycles), the odd one the fragment reads, the DMA pieces, the row-sum add of the score exponentiated one gap earlier
  // and the pack of a finished pair
#define N16_GAP(G)                                                                                                  \
  do {                                                                                                              \
    const int g_ = (G) >> 1;                                                                                        \
    if (((G) & 1) == 0) {                                                /* exp2 + the previous score's row-sum add */  \
      if (g_ >= 4 && g_ <= 18) exp16(b, g_ + 13, kv_rem, lg);            /* q-half b, tile t-1: scores 17..31 */       \
      if (g_ >= 19 && g_ <= 50) exp16(a, g_ - 19, kv_rem, lg);           /* q-half a, tile t */                        \
      if (g_ >= 51) exp16(b, bk0(g_), kv_rem, lg);                       /* q-half b, tile t: scores 0..16 */          \
    }                                                                                                               \
    if ((G) >= 41 && (G) <= 55 && (((G) - 41) & 1) == 0) {               /* V^T(t) fragment (dt, 0): last read by PV_b's MFMA 8 + dt */ \
      const int dt = ((G) - 41) >> 1;                                                                                \
      vf[dt][0] = *(lds_frag*)(smem + (VB + dt * 2048) + vaddr[0]);                                                  \
    }                                                                                                               \
    if ((G) >= 57 && (G) <= 71 && (((G) - 57) & 1) == 0) {               /* V^T(t) fragment (dt, 1): last read by MFMA 24 + dt */ \
      const int dt = ((G) - 57) >> 1;                                                                                \
      vf[dt][1] = *(lds_frag*)(smem + (VB + dt * 2048) + vaddr[1]);                                                  \
    }                                                                                                               \
    if ((G) >= 67 && (G) <= 97 && (((G) - 67) & 1) == 0) {               /* K(t+1) fragment (kt, ks): last read by S_b's MFMA 8 ks + 2 kt + 1 */ \
      const int r = ((G) - 67) >> 1, ks = r >> 2, kt = r & 3;                                                        \
      kf[kt][ks] = *(lds_frag*)(smem + (KN + kt * 4096) + kaddr[ks]);                                                \
    }                                                                                                               \
    if (((G) >= 3 && (G) <= 7) || (G) == 11 || (G) == 19 || (G) == 27) {  /* DMA pieces K0 V0 K1 V1 ... of tile t+2: five in the gaps behind */ \
      const int pc = (G) <= 7 ? (G) - 3 : 5 + (((G) - 11) >> 3);         /* the barrier that carry nothing else, three beside row-sum adds */ \
      dma_piece_i<DST>(smem_rw, dma, (pc & 1) * 4 + (pc >> 1));                                                      \
    }                                                                                                               \
    if (((G) & 1) == 1) {                                                /* the pack of a finished pair (behind the gap's reads: no wait-state nop) */ \
      if (g_ >= 4 && g_ <= 18) { SUM16_ODD(b, g_ + 13); if ((g_ + 13) & 1) pack16(b, g_ + 13); }                    \
      if (g_ == 18) tail16(b);                                                                                      \
      if (g_ >= 19 && g_ <= 50) { SUM16_ODD(a, g_ - 19); if ((g_ - 19) & 1) pack16(a, g_ - 19); }                   \
      if (g_ == 50) tail16(a);                                                                                      \
      if (g_ >= 51) { SUM16_ODD(b, bk0(g_)); if (bk0(g_) & 1) pack16(b, bk0(g_)); }                                 \
      if (g_ >= 56 && ((g_ - 56) & 1) == 0) {                            /* the second exp2 of an old even gap 56..62 */ \
        exp16(b, bk0(g_) + 1, kv_rem, lg);                                                                          \
        SUM16_ODD(b, bk0(g_) + 1);                                                                                  \
        if ((bk0(g_) + 1) & 1) pack16(b, bk0(g_) + 1);                                                              \
      }                                                                                                             \
    }                                                                                                               \
    /* the DMA stream's step to tile t+2, in the three empty gaps behind the barrier (in one piece hipcc sinks it into a single gap: */ \
    /* ~12 scalar instructions = 50 cycles in front of one MFMA); the pieces below (gaps 3..31) fetch from the stepped stream */       \
    if (!MULTI && (G) == 0) {                                            /* plain integer arithmetic: a bool select travels through a lane mask */ \
      const int lo_ = dma.left;                                                                                                    \
      dma.left = lo_ > 2 ? lo_ - 1 : 1;                                  /* = lo - (lo > 1) for lo >= 1 */                           \
      adv_ = lo_ - dma.left;                                                                                                       \
      kb_ = (uint32_t)adv_ * ((uint32_t)KVBLK * dma.rs2);                                                                          \
    }                                                                                                                              \
    if (!MULTI && (G) == 1) { dma.k += kb_; dma.klen -= kb_; }                                                                     \
    if (!MULTI && (G) == 2) { dma.v += adv_ * (KVBLK * 2); }                                                                        \
    if (MULTI && (G) == 4) { const int n_ = cur_tt + 1; cur_tt = n_ == tps ? 0 : n_; }  /* the segment's tile counter (ragged-tail test) */ \
    /* the segment walk (sequence parallelism): the table entry loaded a tile ago, then the load of the next one */                   \
    if (MULTI && (G) == 0) segtab_apply_k(dma, seg);                                                                                \
    if (MULTI && (G) == 1) segtab_apply_v(dma, seg);                                                                                \
    if (MULTI && (G) == 2) segtab_load(seg);                                                                                        \
    if (TIMING && rec && ((G) & 7) == 7) stamp[3 + ((G) >> 3)] = __builtin_amdgcn_s_memtime();  /* tuning build: one stamp per 8 gaps */ \
  } while (0)
  if (TIMING && rec) stamp[2] = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int i = 0; i < 32; ++i) {  // ---- A: S_a = K Q_a^T
    qk_step16<SHIFT>(a, kf, qf, negm, 0, i); SB();
    N16_GAP(i);
    SB();
  }
#pragma unroll
  for (int i = 0; i < 32; ++i) {  // ---- B: O_b += V^T(t-1) P_b(t-1)^T
    pv_step16(b, vf, i); SB();
    N16_GAP(32 + i);
    SB();
  }
#pragma unroll
  for (int i = 0; i < 32; ++i) {  // ---- C: S_b = K Q_b^T
    qk_step16<SHIFT>(b, kf, qf, negm, 2, i); SB();
    N16_GAP(64 + i);
    SB();
  }
#pragma unroll
  for (int i = 0; i < 32; ++i) {  // ---- D: O_a += V^T(t) P_a(t)^T
    pv_step16(a, vf, i); SB();
    N16_GAP(96 + i);
    SB();
  }
#undef N16_GAP
}

__device__ __forceinline__ float sumsq8_16(const mfma_bf16x8& f) {
  const uint4 w = __builtin_bit_cast(uint4, f);
  const uint32_t u[4] = {w.x, w.y, w.z, w.w};
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float lo = __uint_as_float(u[i] << 16), hi = __uint_as_float(u[i] & 0xffff0000u);
    s = __builtin_fmaf(lo, lo, s);
    s = __builtin_fmaf(hi, hi, s);
  }
  return s;
}
__device__ __forceinline__ mfma_bf16x8 prescale8_16(const uint4 raw, float c) {
  uint4 o;
  const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
  uint32_t r[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = cvt_pk(__uint_as_float(w[i] << 16) * c, __uint_as_float(w[i] & 0xffff0000u) * c);
  o.x = r[0]; o.y = r[1]; o.z = r[2]; o.w = r[3];
  return __builtin_bit_cast(mfma_bf16x8, o);
}

// FLAGS: attention_w64q.hip's bits -- bit1 q pre-scaled, bit2 (always set here: the bounded loop), bit4 RAW_OUT, bit5 CARRY_IN,
// bit6 MULTI (several kv segments / a left-out one), bit7 SHIFT.
//
// Since the second half of round 4 a call launches ONLY the shifted instantiation for its bounded pass: a row inside the plain bound
// carries m = 0, i.e. C = +0 -- the plain kernel's arithmetic bit for bit -- and one launch has no second tail and no hand-over (the
// two-launch protocol below cost 4 % when a model's heads were split between the two loops; it remains as the A/B build
// libwanhip_a2l.so, -DWAN_ATTN_TWO_LAUNCH, and the plain instantiations remain for it).  The protocol as first built:
// three launches back to back share one flag word per workgroup (wg_flags: 0 = done, 2 = wants the shifted loop, 1 = wants the
// tracking loop):
//   plain (SHIFT = 0)  every row of the workgroup obeys U = |Q~_row| max|k_h| <= 96: P = 2^s unshifted, as since round 2.  Else the
//                      workgroup sets its flag -- 2 if every U is finite and <= SHIFT_LIMIT, 1 otherwise -- and returns before touching LDS.
//   shifted (SHIFT = 1, round 4)  runs the workgroups flagged 2: the SAME loop, instruction for instruction, with P = 2^(s - m) for a
//                      per-row constant m that enters through the C operand of every S tile's first MFMA.  The row's true maximum
//                      lies in [m_s, U]: U by Cauchy-Schwarz, m_s = its maximum over the 64 keys of tile 0 (one S phase in the
//                      prologue).  U - m_s <= 168: m = U - 96, nothing can overflow or lose its maximum -- guaranteed.  Wider: m =
//                      m_s + 72 and the interval [m_s, m_s + 168] for the true maximum.  Nothing in the loop watches the exponent range
//                      -- the row sum does: a row that left it ends with l outside [2^-80, 2^100] (or NaN), the workgroup then flags
//                      itself 1 AFTER the loop and the tracking launch redoes it.  (Partial launches -- RAW_OUT / CARRY_IN -- must agree
//                      on m without seeing each other's tiles: m = U - 96 only.)  Softmax is shift invariant, so wherever l >= 2^-80 the result is
//                      the max-subtracting kernel's: the terms that underflowed are < 2^-46 of the row sum each.  What this buys: the
//                      fast loop no longer depends on RMSNorm gains staying near 1 -- diffuse random heads pass while the maximum of L
//                      scores stays within 168 of the maximum of 64 (gamma_q gamma_k ~ 40; plain: ~ 6), peaky rows (true maximum near
//                      U) at any gain up to SHIFT_LIMIT.
//   tracking           attention_w64q.hip's instantiation: the workgroups flagged 1.
// Partial sums (sequence parallelism): RAW_OUT leaves them shifted by m(local max|k|); the CARRY_IN launch recomputes that m from the
// previous maxima (kept behind the flags in the scratch) and rescales by 2^(m_old - m_new) before it continues.
constexpr float SHIFT_LIMIT = 2048.0f;      // |m| beyond this costs the fp32 scores visible bits: tracking loop
constexpr float SHIFT_MIN_ROWSUM = 8.271806125530277e-25f;  // 2^-80
constexpr float SHIFT_MAX_ROWSUM = 1.2676506002282294e30f;  // 2^100: |sum P V| <= l max|v| stays finite for |v| < 2^27
// a row maximum may lie 72 below its reference m (its term is then 2^-72: the row sum clears 2^-80 by itself -- with 80 the first
// hardware run flagged the rows whose largest key sits in tile 0, l = 2^-80 (1 + tiny) rounding below the bar) and 96 above it
constexpr float SHIFT_UNDER = 72.0f, SHIFT_WINDOW = 168.0f;
__device__ __forceinline__ float ref_shift16(float u2) {     // u2 = U^2 of a row; the reference m its scores are shifted by
  return u2 <= BOUND16_LOG2 * BOUND16_LOG2 ? 0.f : __builtin_amdgcn_sqrtf(u2) * 1.0001f + 0.01f - BOUND16_LOG2;  // 1-ulp sqrt + margin
}
template <int FLAGS>
__global__ __launch_bounds__(256) void attn_w16n_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ Kg,
                                                       const bf16_t* __restrict__ Vt, bf16_t* __restrict__ O, int B, int Bk,
                                                       int64_t Lq, int64_t Lk, int64_t ldv, int H, int nqb, float scale_log2e,
                                                       int nseg, int64_t k_seg_stride, int64_t vt_seg_stride,
                                                       const float* __restrict__ kmax2, int* __restrict__ wg_flags,
                                                       float* __restrict__ raw, int skip_seg, const uint32_t* __restrict__ seg_tab,
                                                       int v_base, int kv_split, int carry_n, int part_from) {
  // v_base / kv_split / carry_n (round 6, the split tail of a launch; 0 / 1 / 1 = a whole launch as before): the launch covers the
  // workgroups (q blocks) from v_base on.  kv_split = k > 1 (RAW_OUT): k workgroups per q block, part j attends tiles
  // [j n / k, (j + 1) n / k) of the block's n and leaves its unnormalised sums in raw slot (v - v_base) k + j.  carry_n = k > 1 (CARRY_IN):
  // the finishing launch -- no tiles of its own, the k parts' sums added (the bounded softmax's partial sums add exactly: one shift m per
  // row for all parts), normalised, judged and stored like any workgroup's.  part_from (a launch that is neither RAW_OUT nor CARRY_IN):
  // workgroups [0, part_from) of the grid are whole q blocks 0 .. part_from - 1, the workgroups behind them the kv_split parts of the q
  // blocks from part_from on -- ONE launch, so that the parts fill the CUs as the last whole round drains (as a launch of their own they
  // waited for its last workgroup: run 11 measured a third of the predicted gain).
  constexpr bool TIMING = (FLAGS & 1) != 0;  // s_memtime stamps of tile 300 of workgroup 0 -> first 160 B of O (tuning build only, -DW64Q_TIMING)
  constexpr bool PRESCALED = (FLAGS & 2) != 0;
  constexpr bool RAW_OUT = (FLAGS & 16) != 0, CARRY_IN = (FLAGS & 32) != 0;
  constexpr bool MULTI = (FLAGS & 64) != 0;
  constexpr bool SHIFT = (FLAGS & 128) != 0;
  constexpr bool SAMPLE = SHIFT && !RAW_OUT && !CARRY_IN;  // the reference also uses a lower bound of the row maximum (partial launches must agree on m: Cauchy-Schwarz only)
  // PERSIST (bit 8; short KV -- cross-attention, 8 to 32 tiles): one workgroup per CU walks a contiguous run of q blocks.  With a handful
  // of tiles per block everything around the tile loop used to be exposed -- the workgroup's launch, the Q rows' trip from HBM, the first
  // two tiles' DMA, the O rows' way out, one workgroup per CU and nobody to run meanwhile (cross-attention at 0.30 of peak for four
  // rounds).  Here the NEXT block's 64 KB of Q rows arrive by LDS-DMA into the 64 KB of LDS the ring leaves free, two pieces per wave at
  // the top of each of the block's first eight tiles, and a block switch costs the O staging, 16 DMA pieces and 16 ds_reads.
  constexpr bool PERSIST = (FLAGS & 256) != 0;
  static_assert(!PERSIST || (SHIFT && !RAW_OUT && !CARRY_IN && !MULTI && !TIMING), "PERSIST: the one-launch bounded pass of a single-segment call");
  constexpr int QAREA = 2 * NST * IMG;  // PERSIST: the next block's Q rows, 16 KB per wave (row r of the wave at r * 256, 16-B chunk c at slot c ^ (r & 15))
  constexpr int WAVE_RAW = 2 * (64 * 64 + 128), WG_RAW = 4 * WAVE_RAW;  // floats: per half 64 accumulators x 64 lanes + 2 x 64 row-sum shares
  __shared__ __attribute__((aligned(16))) char smem[2 * NST * IMG + (PERSIST ? 4 * 16384 : 0)];    // [K stages][V^T stages] = 96 KB (+ 64 KB)
  lds_cchar* lds = (lds_cchar*)smem;
  uint64_t stamp[20] = {};
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lg = lane >> 4;

  const int total = nqb * H * B;
  const int64_t rs = (int64_t)H * 128;
  // PERSIST: the wave's 64 Q rows of block vn -> its 16 KB of the Q area, piece p = rows 4 p .. 4 p + 3 (a lane: row 4 p + lg, slot l15).  Rows
  // past Lq read as zeros (the descriptor ends at the last valid row); a wave wholly past Lq re-reads row Lq - 1 (never stored).
  const char* qpf_base = nullptr;
  uint32_t qpf_len = 0;
  // PERSIST: the block's coordinates are carried along (an integer division is a ~40-instruction VALU sequence: three per block cost 2,000 cycles)
  int p_pair = 0, p_qb = 0, p_b = 0, p_h = 0;
  auto p_adv = [&]() {
    if (!PERSIST) return;
    if (++p_qb == nqb) { p_qb = 0; ++p_pair; p_b = p_pair / H; p_h = p_pair - p_b * H; }
  };
  auto qpf_set = [&](int pr, int qbn, int bn, int hn) {
    int64_t r0 = (int64_t)qbn * 256 + wave * 64;
    if (r0 > Lq - 1) r0 = Lq - 1;
    int64_t nv = Lq - r0;
    if (nv > 64) nv = 64;
    qpf_base = uni(reinterpret_cast<const char*>(Q + ((int64_t)bn * Lq + r0) * rs + (int64_t)hn * 128));
    qpf_len = uni((uint32_t)(nv - 1) * (uint32_t)(rs * 2) + 256u);
  };
  auto qpf_piece = [&](int pc) {
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t voff = (uint32_t)(4 * pc + lg) * (uint32_t)(rs * 2) + (uint32_t)((l15 ^ ((4 * pc + lg) & 15)) << 4);
    dma_issue(voff, qpf_base, qpf_len, lds0 + QAREA + wave * 16384 + pc * 1024);
  };
  // -DW16N_PSTAMPS (make pstamp; diagnostics): s_memtime at the stations of workgroup 0's fourth block -> raw (the launcher prints them)
#ifdef W16N_PSTAMPS
#define PST(I) do { if (PERSIST && pst_on && tid == 0) reinterpret_cast<uint64_t*>(raw)[I] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PST(I) do { } while (0)
#endif
  int v_first, v_end, kv_part = 0;
  bool is_part = false;
  if (PERSIST) {
    const int per = (total + (int)gridDim.x - 1) / (int)gridDim.x;
    v_first = (int)blockIdx.x * per;
    v_end = v_first + per < total ? v_first + per : total;
    if (v_first >= v_end) return;
    p_pair = v_first / nqb; p_qb = v_first - p_pair * nqb; p_b = p_pair / H; p_h = p_pair - p_b * H;
    qpf_set(p_pair, p_qb, p_b, p_h);
    for (int pc = 0; pc < 16; ++pc) qpf_piece(pc);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  } else {
    if (!RAW_OUT && !CARRY_IN && kv_split > 1 && (int)blockIdx.x >= part_from) {   // a part behind the whole q blocks of its launch
      const int u = xcd_remap((int)blockIdx.x - part_from, (int)gridDim.x - part_from);
      is_part = true;
      v_base = part_from;
      v_first = part_from + u / kv_split;
      kv_part = u - (u / kv_split) * kv_split;
    } else if (!RAW_OUT && !CARRY_IN && kv_split > 1) {
      v_first = xcd_remap(blockIdx.x, part_from);
    } else {
      const int u = xcd_remap(blockIdx.x, (int)gridDim.x);   // (a whole launch: gridDim.x == total)
      v_first = v_base + u / kv_split;
      kv_part = u - (u / kv_split) * kv_split;
    }
    v_end = v_first + 1;
  }
  auto qpf_set_next = [&]() {   // the block after the current one
    if (p_qb + 1 < nqb) { qpf_set(p_pair, p_qb + 1, p_b, p_h); return; }
    const int pr = p_pair + 1, bn = pr / H;
    qpf_set(pr, 0, bn, pr - bn * H);
  };
  for (int v = v_first; v < v_end; ++v, p_adv()) {
#ifdef W16N_PSTAMPS
  const bool pst_on = PERSIST && raw != nullptr && blockIdx.x == 0 && v == v_first + 3;
#endif
  PST(0);
  if (PERSIST && v != v_first) __syncthreads();   // every wave is done with the O staging area: the ring may be written again
  PST(1);
#ifdef WAN_ATTN_TWO_LAUNCH   // (the A/B library libwanhip_a2l.so: plain launch first, the shifted twin for what it hands over)
  if (SHIFT) { if (wg_flags[v] != 2) return; }  // only what the plain launch handed over
  else if (CARRY_IN && wg_flags[v] != 0) return;  // an earlier partial launch already gave this workgroup up
#else
  if (CARRY_IN) {                                 // an earlier partial launch already gave this workgroup up (1), or finished it whole (3: a split tail's shifted block)
    const int fv = wg_flags[v];
    if (fv != 0) {
      if (fv == 3) {
        __syncthreads();                          // (every thread has read the 3)
        if (tid == 0) wg_flags[v] = 0;
      }
      return;
    }
  }
#endif
  const int pair = PERSIST ? p_pair : v / nqb;
  const int qb = PERSIST ? p_qb : v - pair * nqb;
  const int b = PERSIST ? p_b : pair / H, h = PERSIST ? p_h : pair - b * H;
  const int bk = b % Bk;  // Bk == B: its own K / V^T; Bk == 1: shared; Bk | B (Ulysses: q batches = (source rank, stream), K / V^T batches = stream): b mod Bk

  const bf16_t* qbase = Q + ((int64_t)b * Lq) * rs + (int64_t)h * 128;
  const bf16_t* kbase = Kg + ((int64_t)bk * Lk) * rs + (int64_t)h * 128;
  const bf16_t* vbase = Vt + ((int64_t)bk * H * 128 + (int64_t)h * 128) * ldv;   // (both move t_lo tiles in for a part of a split tail)
  bf16_t* obase = O + ((int64_t)b * Lq) * rs + (int64_t)h * 128;

  const int64_t q0 = (int64_t)qb * 256 + wave * 64;
  int Lk32 = (int)Lk;
  Dma dma;
  if (PERSIST) {  // tiles 0 and 1 first: they travel while the Q fragments are read and the rows are judged
    dma_init(dma, kbase, vbase, 0, 0, Lk32, 1, (uint32_t)(rs * 2), (uint32_t)(ldv * 2), tid, wave, -1, /*k_rows_16x16=*/true);
    dma_tile<0, false>(smem, dma);
#pragma unroll
    for (int I = 0; I < 8; ++I) dma_piece_i<1>(smem, dma, I);
  }
  mfma_bf16x8 qf[4][4];  // Q~ = bf16(q * scale * log2 e), [q tile][k-step]
  float ss[4] = {0.f, 0.f, 0.f, 0.f};  // |Q~_row|^2, this lane's 32 of the row's 128 channels
  {
    uint4 w[4][4];  // all 16 loads in flight before the first use
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      int64_t r = q0 + 16 * qt + l15;
      if (r > Lq - 1) r = Lq - 1;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (PERSIST) w[qt][ks] = *reinterpret_cast<const uint4*>(smem + QAREA + wave * 16384 + (16 * qt + l15) * 256 + (((ks * 4 + lg) ^ l15) << 4));
        else w[qt][ks] = *reinterpret_cast<const uint4*>(qbase + r * rs + ks * 32 + lg * 8);
      }
    }
#pragma unroll
    for (int qt = 0; qt < 4; ++qt)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        qf[qt][ks] = PRESCALED ? __builtin_bit_cast(mfma_bf16x8, w[qt][ks]) : prescale8_16(w[qt][ks], scale_log2e);
        ss[qt] += sumsq8_16(qf[qt][ks]);
        asm volatile("" : "+a"(qf[qt][ks]));  // one accumulator-file tuple from here on
      }
  }
  PST(2);   // Q fragments in registers
  // ---- the bound: every row of the workgroup must satisfy |Q~_row| * max|k_h| <= 96 (workgroup-uniform) -- or carry a reference shift
  f32x4 negm[4];      // SHIFT: {-m} x 4 per q tile, the C operand of the tile's first MFMA
  float mref[4] = {0.f, 0.f, 0.f, 0.f};
  bool wg_any_shift = true;  // does any row of the workgroup carry a shift (else the sample of tile 0 is skipped)
  {
    const float km = kmax2[bk * H + h];
    bool ok = true, shiftable = true;
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      float s = ss[qt];
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      ss[qt] = s;
      const float u2 = s * km;
      ok = ok && (u2 <= BOUND16_LOG2 * BOUND16_LOG2);          // false for NaN
      shiftable = shiftable && (u2 <= SHIFT_LIMIT * SHIFT_LIMIT);
      if (SHIFT) {
        mref[qt] = ref_shift16(u2);
        const float n = -mref[qt];
        negm[qt] = f32x4{n, n, n, n};
      }
    }
    if (!SHIFT) {
      if (__syncthreads_and(ok ? 1 : 0) == 0) {
        const int can = __syncthreads_and(shiftable ? 1 : 0);
        if (tid == 0) wg_flags[v] = can ? 2 : 1;
        return;
      }
    }
#ifndef WAN_ATTN_TWO_LAUNCH
    else if (PERSIST) {
      // (all 160 KB of LDS are taken: the workgroup-wide votes go through one word per wave instead of __syncthreads_and / _or's own
      // 256 bytes.  Here the word is the first of the wave's Q row 63 -- its Q fragments are in registers, and the piece of the next block's
      // rows that lands there is issued at the top of tile 7, seven barriers from here.)
      const int mine = (__builtin_amdgcn_ballot_w64(!shiftable) != 0 ? 1 : 0) | (__builtin_amdgcn_ballot_w64(!ok) != 0 ? 2 : 0);
      if (lane == 0) *reinterpret_cast<int*>(smem + QAREA + wave * 16384 + 63 * 256) = mine;
      __syncthreads();
      int any = 0;
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) any |= *reinterpret_cast<const int*>(smem + QAREA + w4 * 16384 + 63 * 256);
      if (any & 1) {  // a reference beyond SHIFT_LIMIT (or NaN): tracking loop
        if (tid == 0) wg_flags[v] = 1;
        if (v + 1 < v_end) {  // the next block's Q rows, all at once (this block has no tile loop to spread them over)
          __syncthreads();    // (every wave has read the votes)
          qpf_set_next();
          for (int pc = 0; pc < 16; ++pc) qpf_piece(pc);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (tiles 0 and 1 of this block were on their way)
        continue;
      }
      wg_any_shift = (any & 2) != 0;
    }
    else {  // the ONE bounded launch of a call: a row inside the plain bound has m = 0 (C = +0: the plain kernel's arithmetic, bit for bit)
      if (__syncthreads_and(shiftable ? 1 : 0) == 0) {  // a reference beyond SHIFT_LIMIT (or NaN): tracking loop, before LDS is touched
        if (tid == 0) wg_flags[v] = 1;
        return;
      }
      wg_any_shift = __syncthreads_or(ok ? 0 : 1) != 0;
    }
#endif
  }
  const bool qpf_on = PERSIST && v + 1 < v_end;
  if (qpf_on) qpf_set_next();
  PST(3);   // votes done

  // A part of a split tail whose q block carries a reference shift (gains beyond the plain bound) does not split: the parts would have to agree
  // on m without the sample of tile 0 (the Cauchy-Schwarz reference alone underflows diffuse rows, and the tracking launch would redo the
  // block).  Part 0 attends the whole block like any workgroup and leaves flag 3 -- "done, nothing to finish" -- which the finishing launch
  // turns back into 0; the other parts leave.  (wg_any_shift is a function of the block's Q rows and the head's max |k|: the same in every part.)
  bool whole_by_part0 = false;
  if (SHIFT && !RAW_OUT && !CARRY_IN && is_part && wg_any_shift) {
    if (kv_part != 0) return;
    is_part = false;
    whole_by_part0 = true;
  }
  const bool raw_out = RAW_OUT || is_part;   // this workgroup leaves unnormalised sums
  // the split tail: part kv_part of kv_split attends the tiles [t_lo, t_hi) of the block's walk
  const int tps_all = ((int)Lk + KVBLK - 1) / KVBLK;
  const int ntile_all = tps_all * (nseg - (skip_seg >= 0 ? 1 : 0));
  const int t_lo = (raw_out && kv_split > 1) ? (int)((int64_t)kv_part * ntile_all / kv_split) : 0;
  const int t_hi = (raw_out && kv_split > 1) ? (int)((int64_t)(kv_part + 1) * ntile_all / kv_split) : ntile_all;
  if (!MULTI && raw_out && kv_split > 1) {   // one segment: the part is a K / V^T of its own, t_lo tiles in
    kbase += (int64_t)t_lo * KVBLK * rs;
    vbase += (int64_t)t_lo * KVBLK;
    const int left = (int)Lk - t_lo * KVBLK;
    const int mine = (t_hi - t_lo) * KVBLK;
    Lk32 = left < mine ? left : mine;
  }
  // ---- DMA stream ---------------------------------------------------------------------------------------------------------------
  const int tps = MULTI ? tps_all : (Lk32 + KVBLK - 1) / KVBLK;
  // (CARRY_IN with carry_n > 1: the finishing launch of a split tail has no tiles of its own)
  const int ntile = (CARRY_IN && carry_n > 1) ? 0 : (MULTI ? t_hi - t_lo : tps);
  if (!PERSIST) dma_init(dma, kbase, vbase, k_seg_stride * 2, vt_seg_stride * 2, Lk32, nseg, (uint32_t)(rs * 2), (uint32_t)(ldv * 2), tid, wave, skip_seg, /*k_rows_16x16=*/true);
  int cur_tt = MULTI ? t_lo % tps_all : 0;   // (a part of a split tail starts t_lo tiles into the walk)
  // valid kv rows from the start of the tile being consumed to the end of its segment: Lk - cur_tt * 64 (cur_tt steps inside the tile)

  // ---- LDS fragment addresses: per-lane VGPR + compile-time immediates ------------------------------------------------------------
  int kaddr[4], vaddr[2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) kaddr[ks] = l15 * 256 + (((ks * 4 + lg) ^ l15) << 4);
#pragma unroll
  for (int c = 0; c < 2; ++c)  // + V^T region base: keeps every ds_read offset inside the 16-bit immediate
    vaddr[c] = NST * IMG + l15 * 128 + (((c * 4 + lg) ^ ((l15 >> 1) & 7)) << 4);

  QH qa, qb2;
#pragma unroll
  for (int dt = 0; dt < 8; ++dt)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { qa.accO[dt][qt][r] = 0.f; qb2.accO[dt][qt][r] = 0.f; }
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) { qa.l[qt][0] = qa.l[qt][1] = 0.f; qb2.l[qt][0] = qb2.l[qt][1] = 0.f; }
  // raw slot of this workgroup: a whole launch: v; a split tail: (v - v_base) x parts + part
  const size_t raw_slot = (size_t)(v - v_base) * (size_t)((raw_out || CARRY_IN) ? kv_split * carry_n : 1) + (size_t)kv_part;
  if (CARRY_IN) {  // partial sums of an earlier launch over other kv segments (same grid, same workgroup -> same slots); carry_n > 1: of the k parts
    for (int j = 0; j < carry_n; ++j) {
      const float4* src = reinterpret_cast<const float4*>(raw + (raw_slot + (size_t)j) * WG_RAW + (size_t)wave * WAVE_RAW);
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int dt = 0; dt < 8; ++dt)
#pragma unroll
          for (int qt = 0; qt < 2; ++qt) {
            f32x4& acc = hf ? qb2.accO[dt][qt] : qa.accO[dt][qt];
            const float4 w4 = src[((hf * 8 + dt) * 2 + qt) * 64 + lane];
            acc[0] += w4.x; acc[1] += w4.y; acc[2] += w4.z; acc[3] += w4.w;    // (the accumulators were zeroed above: the first part's sums exactly)
            asm volatile("" : "+a"(acc));
          }
      const float* ls = raw + (raw_slot + (size_t)j) * WG_RAW + (size_t)wave * WAVE_RAW + 2 * 64 * 64;
      qa.l[0][0] += ls[lane]; qa.l[1][0] += ls[64 + lane];
      qb2.l[0][0] += ls[128 + lane]; qb2.l[1][0] += ls[192 + lane];
    }
    if (SHIFT) {  // the carried sums were shifted by m(previous maxima) <= m: bring them to this launch's reference
      const float kmp = kmax2[(size_t)Bk * H + (size_t)total + bk * H + h];
      float f[4];
#pragma unroll
      for (int qt = 0; qt < 4; ++qt) f[qt] = __builtin_amdgcn_exp2f(ref_shift16(ss[qt] * kmp) - mref[qt]);
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int dt = 0; dt < 8; ++dt)
#pragma unroll
          for (int qt = 0; qt < 2; ++qt) {
            f32x4& acc = hf ? qb2.accO[dt][qt] : qa.accO[dt][qt];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] *= f[hf * 2 + qt];
            asm volatile("" : "+a"(acc));
          }
      qa.l[0][0] *= f[0]; qa.l[1][0] *= f[1]; qb2.l[0][0] *= f[2]; qb2.l[1][0] *= f[3];
    }
  }
  qa.p0 = qa.p1 = qb2.p0 = qb2.p1 = 0.f;
  // q-half b starts half a tile behind: its first PV_b runs on an all-masked dummy tile (P = 0)
  {
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int f = 0; f < 4; ++f) { qa.pk[f] = z; qb2.pk[f] = z; }
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { qb2.s[kt][qt][r] = -INFINITY; qa.s[kt][qt][r] = -INFINITY; }
  }
  mfma_bf16x8 vf[8][2];  // V^T fragments, carried from tile t (PV_a) to B of tile t+1 (PV_b)
  {
    uint4 z; z.x = z.y = z.z = z.w = 0u;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt)
#pragma unroll
      for (int c = 0; c < 2; ++c) vf[dt][c] = __builtin_bit_cast(mfma_bf16x8, z);
  }

  PST(4);   // accumulators cleared, stream set up
  SegTab seg = {};
  if (MULTI) {   // entries 0 and 1 by hand, entry 2 on its way for the first tile's step
    seg.tab = reinterpret_cast<const uint4*>(seg_tab) + t_lo;
    seg.kb = uni(reinterpret_cast<const char*>(kbase));
    seg.vb = uni(reinterpret_cast<const char*>(vbase));
    seg.i = 0u;
    seg.nxt = seg.tab[0];
    segtab_apply_k(dma, seg);
    segtab_apply_v(dma, seg);
#pragma unroll
    for (int I = 0; I < 8; ++I) dma_piece_i<0>(smem, dma, I);
    segtab_load(seg);
    segtab_apply_k(dma, seg);
    segtab_apply_v(dma, seg);
#pragma unroll
    for (int I = 0; I < 8; ++I) dma_piece_i<1>(smem, dma, I);
    segtab_load(seg);
  } else if (!PERSIST) {
    dma_tile<0, false>(smem, dma);
#pragma unroll
    for (int I = 0; I < 8; ++I) dma_piece_i<1>(smem, dma, I);  // tile 1; the stream steps at the top of every tile of the loop, in front of its pieces
  }
  PST(5);   // 16 pieces issued
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  PST(6);   // tiles 0, 1 landed
  mfma_bf16x8 kf[4][4];
  // in the order the tile loop re-reads them (k-step major, then kv tile)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    kf[r & 3][r >> 2] = *(lds_frag*)(lds + (r & 3) * 4096 + kaddr[r >> 2]);
    SB();
  }

  if (SAMPLE && !is_part && Lk32 >= KVBLK && wg_any_shift) {   // (parts must agree on m without seeing each other's tiles: the Cauchy-Schwarz reference only)
    // ---- a LOWER bound of every row's maximum: its scores against the 64 keys of tile 0 (the MFMAs of one S phase, once per workgroup).
    // The row's true maximum lies in [m_s, U].  Where that interval is at most 176 wide, m = U - 96 covers it (nothing can overflow, the
    // maximum cannot underflow below 2^-80): guaranteed.  Where it is wider -- large gains on diffuse rows: U grows like 16 gamma, the
    // maximum of L random scores like 6 gamma -- m = m_s + 80 keeps the guarantee on the low side and leaves [m_s, m_s + 176] for the true
    // maximum; beyond that P overflows to inf, the row sum says so, and the workgroup goes to the tracking loop like an underflowed one.
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
          f32x4& d = (qt < 2) ? qa.s[kt][qt] : qb2.s[kt][qt - 2];
          if (ks == 0) qk16_0(d, kf[kt][0], qf[qt][0]);
          else qk16(d, kf[kt][ks], qf[qt][ks]);
        }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // inline-asm MFMAs: the hazard recognizer does not see their latency
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        f32x4& d = (qt < 2) ? qa.s[kt][qt] : qb2.s[kt][qt - 2];
        asm volatile("" : "+v"(d));
        mx = fmaxf(fmaxf(mx, fmaxf(d[0], d[1])), fmaxf(d[2], d[3]));
        d = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};   // back to the dummy tile the first PV_b expects
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      if (mref[qt] != 0.f) {
        const float u = mref[qt] + BOUND16_LOG2;                // the row's U (with its margin)
        const float m = (u - mx <= SHIFT_WINDOW) ? mref[qt] : mx + SHIFT_UNDER;   // NaN scores: the comparison fails, m = NaN, the row sum flags it
        mref[qt] = m;
        const float n = -m;
        negm[qt] = f32x4{n, n, n, n};
      }
    }
  }

  PST(7);   // K fragments read, sample done
  int kv_rem_prev = KVBLK;
#define W16N_STEP(J)                                                                                         \
  if (__builtin_expect(t + (J) < ntile, 1)) {                                                                \
    const bool rec = TIMING && (t + (J) == 300);                                                             \
    if (TIMING && rec) stamp[0] = __builtin_amdgcn_s_memtime();                                              \
    if (t + (J) > 0) {                                                                                       \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* tile t+J+1 landed */                               \
      __builtin_amdgcn_s_barrier();                                                                          \
      asm volatile("" ::: "memory");                                                                         \
    }                                                                                                        \
    __builtin_amdgcn_s_waitcnt(0xc07f); /* lgkmcnt(0): every K fragment of this tile was read >= 30 gaps ago */ \
    if (PERSIST && __builtin_expect(qpf_on && t + (J) < 8, 1)) { qpf_piece(2 * (t + (J))); qpf_piece(2 * (t + (J)) + 1); } /* the next block's Q rows */ \
    if (TIMING && rec) stamp[1] = __builtin_amdgcn_s_memtime();                                              \
    const int kv_rem = Lk32 - (MULTI ? cur_tt : t + (J)) * KVBLK;                                            \
    tile_w16n<J, MULTI, TIMING, SHIFT>(lds, kaddr, vaddr, qf, negm, kf, vf, qa, qb2, kv_rem, lg, smem, dma, seg, cur_tt, tps, stamp, rec); \
    kv_rem_prev = kv_rem;                                                                                    \
  }
  for (int t = 0; t < ntile; t += NST) {
    W16N_STEP(0)
    W16N_STEP(1)
    W16N_STEP(2)
  }
#undef W16N_STEP
  PST(8);   // tile loop done
  // drain: q-half b's last tile
#pragma unroll
  for (int k = 17; k < 32; ++k) {
    exp16(qb2, k, kv_rem_prev, lg);
    SUM16_ODD(qb2, k);
    if (k & 1) pack16(qb2, k);
  }
  tail16(qb2);
  asm volatile("s_nop 1" ::: "memory");
#pragma unroll
  for (int i = 0; i < 32; ++i) pv_step16(qb2, vf, i);

  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");  // trailing DMA lands before O staging reuses LDS; last PV MFMAs -> accumulator reads
  PST(9);   // drained
  float lsum[4] = {qa.l[0][0] + qa.l[0][1], qa.l[1][0] + qa.l[1][1], qb2.l[0][0] + qb2.l[0][1], qb2.l[1][0] + qb2.l[1][1]};
  if (raw_out) {  // partial result: accumulators and row-sum shares as they are, lane-major (1-KB stores)
    float4* dst = reinterpret_cast<float4*>(raw + raw_slot * WG_RAW + (size_t)wave * WAVE_RAW);
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int dt = 0; dt < 8; ++dt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          const f32x4 acc = hf ? qb2.accO[dt][qt] : qa.accO[dt][qt];
          dst[((hf * 8 + dt) * 2 + qt) * 64 + lane] = float4{acc[0], acc[1], acc[2], acc[3]};
        }
    float* ls = raw + raw_slot * WG_RAW + (size_t)wave * WAVE_RAW + 2 * 64 * 64;
#pragma unroll
    for (int c = 0; c < 4; ++c) ls[c * 64 + lane] = lsum[c];
    if (SHIFT && tid == 0) wg_flags[v] = 0;  // partial sums left; underflow is judged on the final row sums (CARRY_IN launch)
    return;
  }
  // ---- epilogue: normalise, stage the wave's 64 x 128 O tile through LDS, store whole rows ------------------------------------------
  // (PERSIST: lane-derived addresses of the epilogue are recomputed per block from opaque copies -- hoisted out of the block loop they
  // would sit in vector registers through the tile loop, which has none to spare)
  int l15e = l15, lge = lg, lanee = lane;
  if (PERSIST) asm volatile("" : "+v"(l15e), "+v"(lge), "+v"(lanee));
  float inv[4];
  bool bad = false;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float s = lsum[c];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    inv[c] = 1.0f / s;
    if (SHIFT) bad = bad || !((s >= SHIFT_MIN_ROWSUM && s <= SHIFT_MAX_ROWSUM) || mref[c] == 0.f);  // under- / overflowed against its shift (or NaN): the tracking loop redoes the workgroup
  }
  if (SHIFT) {
    int any_bad;
    if (PERSIST) {  // the vote through V^T stage 2 (the O staging below covers the K stages and V^T stage 0; the ring is idle behind the first barrier)
      __syncthreads();
      if (lanee == 0) *reinterpret_cast<int*>(smem + (NST + 2) * IMG + wave * 4) = __builtin_amdgcn_ballot_w64(bad) != 0 ? 1 : 0;
      __syncthreads();
      const int4 vt4 = *reinterpret_cast<const int4*>(smem + (NST + 2) * IMG);
      any_bad = vt4.x | vt4.y | vt4.z | vt4.w;
    } else {
      any_bad = __syncthreads_or(bad ? 1 : 0);
    }
    if (tid == 0) wg_flags[v] = any_bad ? 1 : (whole_by_part0 ? 3 : 0);
    // a handed-over workgroup stores nothing: wan_dit_forward attends IN PLACE (o = q), and the tracking launch that redoes the workgroup
    // reads its Q rows again
    if (any_bad) { if (PERSIST) continue; return; }
  } else {
    __syncthreads();
  }
  PST(10);  // verdict
  char* ob = smem + wave * (64 * 256);
#pragma unroll
  for (int hf = 0; hf < 2; ++hf)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const float iv = inv[hf * 2 + qt];
      const int row = (hf * 2 + qt) * 16 + l15e;
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) {
        const f32x4 acc = hf ? qb2.accO[dt][qt] : qa.accO[dt][qt];
        uint2 w;
        w.x = cvt_pk(acc[0] * iv, acc[1] * iv);
        w.y = cvt_pk(acc[2] * iv, acc[3] * iv);
        const int ch = (dt * 2 + (lge >> 1)) ^ (l15e & 15);  // d = 16 dt + 4 g .. + 3: 16-B chunk 2 dt + (g >> 1), its half g & 1
        *reinterpret_cast<uint2*>(ob + row * 256 + ch * 16 + (lge & 1) * 8) = w;
      }
    }
  __syncthreads();
  if (PERSIST) {  // all sixteen staged rows first: one read per conditional store serialises sixteen LDS round trips (2,400 cycles of a 4,500-cycle epilogue)
    uint4 vals[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = i * 4 + (lanee >> 4), c = lanee & 15;
      vals[i] = *reinterpret_cast<const uint4*>(ob + r * 256 + ((c ^ (r & 15)) << 4));
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = i * 4 + (lanee >> 4), c = lanee & 15;
      const int64_t qr = q0 + r;
      if (qr < Lq) *reinterpret_cast<uint4*>(obase + qr * rs + c * 8) = vals[i];
    }
  } else {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = i * 4 + (lanee >> 4), c = lanee & 15;
    const int64_t qr = q0 + r;
    if (qr < Lq) {
      const uint4 val = *reinterpret_cast<const uint4*>(ob + r * 256 + ((c ^ (r & 15)) << 4));
      *reinterpret_cast<uint4*>(obase + qr * rs + c * 8) = val;
    }
  }
  }
  PST(11);  // O rows on their way
  if (TIMING && blockIdx.x == 0 && tid == 0) {
    __builtin_amdgcn_s_waitcnt(0);
#pragma unroll
    for (int k6 = 0; k6 < 20; ++k6) reinterpret_cast<uint64_t*>(O)[k6] = stamp[k6];
  }
  }  // q blocks of the workgroup (one unless PERSIST)
}

}  // namespace

// ---- the segment table of a multi-segment launch (see SegTab) ----------------------------------------------------------------------
namespace {
__global__ void attn_seg_table_kernel(uint4* __restrict__ tab, int n_entries, int ntile, int tps, int skip, int64_t kseg_bytes, int64_t vseg_bytes,
                                      uint32_t rs2, int Lk) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n_entries) return;
  const int ff = f < ntile ? f : ntile - 1;                 // past the end the stream stays put (later fetches land in a dead stage)
  int sg = ff / tps;
  const int tt = ff - sg * tps;
  if (skip >= 0 && sg >= skip) ++sg;                        // the left-out segment (the rank's own: attended by the local launch)
  const uint64_t koff = (uint64_t)sg * (uint64_t)kseg_bytes + (uint64_t)tt * KVBLK * rs2;
  const uint64_t voff = (uint64_t)sg * (uint64_t)vseg_bytes + (uint64_t)tt * (KVBLK * 2);
  uint4 e;
  e.x = (uint32_t)(koff >> 4);
  e.y = (uint32_t)(voff >> 4);
  e.z = (uint32_t)(Lk - 1) * rs2 + 256u - (uint32_t)tt * KVBLK * rs2;   // valid K bytes from the tile's first row to the segment's last
  e.w = 0u;
  tab[f] = e;
}
struct SegTabKey {
  int device;
  hipStream_t stream;
  int nseg, skip, Lk;
  int64_t kseg, vseg;
  uint32_t rs2;
  bool operator==(const SegTabKey& o) const {
    return device == o.device && stream == o.stream && nseg == o.nseg && skip == o.skip && Lk == o.Lk && kseg == o.kseg && vseg == o.vseg && rs2 == o.rs2;
  }
};
struct SegTabEntry {
  SegTabKey key;
  uint4* ptr;
  size_t bytes;
  uint64_t last_use;
};
std::mutex g_segtab_mutex;
SegTabEntry g_segtabs[32];
int g_nsegtabs = 0;
uint64_t g_segtab_clock = 0;
}  // namespace

// -> device pointer of the table for this walk on this (device, stream), building it (one tiny launch, ordered on `stream` in front of
// the attention launch that reads it) at the walk's first appearance there; a forward's 40 blocks and a video's steps reuse it.
// 32 live walks; the least recently used one makes room (hipFree synchronises the device: nothing in flight reads it).
static int seg_table(const uint32_t** out, int nseg, int skip, int Lk, int64_t kseg_bytes, int64_t vseg_bytes, uint32_t rs2, hipStream_t stream) {
  WAN_REQUIRE((kseg_bytes & 15) == 0 && (vseg_bytes & 15) == 0 && kseg_bytes >= 0 && vseg_bytes >= 0,
              "wan_attention: segment strides must be multiples of 16 bytes");
  const int tps = (Lk + KVBLK - 1) / KVBLK;
  const int ntile = tps * (nseg - (skip >= 0 ? 1 : 0));
  WAN_REQUIRE(ntile >= 1, "wan_attention: a segmented launch without a segment to attend");
  WAN_REQUIRE(((uint64_t)nseg * (uint64_t)kseg_bytes >> 4) < ((uint64_t)1 << 32) && ((uint64_t)nseg * (uint64_t)vseg_bytes >> 4) < ((uint64_t)1 << 32),
              "wan_attention: K / V^T segments exceed the segment table's 36-bit offsets");
  const int n_entries = ntile + 3;                        // the stream fetches two tiles past the end, and the last tile still loads the entry behind those
  SegTabKey key;
  WAN_CHECK_HIP(hipGetDevice(&key.device));
  key.stream = stream; key.nseg = nseg; key.skip = skip; key.Lk = Lk; key.kseg = kseg_bytes; key.vseg = vseg_bytes; key.rs2 = rs2;
  std::lock_guard<std::mutex> lock(g_segtab_mutex);
  for (int i = 0; i < g_nsegtabs; ++i)
    if (g_segtabs[i].key == key) {
      g_segtabs[i].last_use = ++g_segtab_clock;
      *out = reinterpret_cast<const uint32_t*>(g_segtabs[i].ptr);
      return 0;
    }
  SegTabEntry* e = nullptr;
  if (g_nsegtabs < 32) {
    e = &g_segtabs[g_nsegtabs++];
    e->ptr = nullptr;
    e->bytes = 0;
  } else {
    e = &g_segtabs[0];
    for (int i = 1; i < 32; ++i)
      if (g_segtabs[i].last_use < e->last_use) e = &g_segtabs[i];
  }
  const size_t need = (size_t)n_entries * sizeof(uint4);
  if (e->bytes < need) {
    if (e->ptr) (void)hipFree(e->ptr);
    e->ptr = nullptr;
    e->bytes = 0;
    e->key = SegTabKey{};
    e->key.device = -1;
    WAN_CHECK_HIP(hipMalloc((void**)&e->ptr, need));
    e->bytes = need;
  } else if (e->key.device >= 0) {
    WAN_CHECK_HIP(hipDeviceSynchronize());                   // a recycled buffer: its old walk may still be read
  }
  hipLaunchKernelGGL(attn_seg_table_kernel, dim3((unsigned)((n_entries + 255) / 256)), dim3(256), 0, stream, e->ptr, n_entries, ntile, tps, skip,
                     kseg_bytes, vseg_bytes, rs2, Lk);
  WAN_LAUNCH_CHECK();
  e->key = key;
  e->last_use = ++g_segtab_clock;
  *out = reinterpret_cast<const uint32_t*>(e->ptr);
  return 0;
}

// The bounded launch of attention_w64q.hip's protocol on this kernel.  fl = that file's FLAGS value (bit 2 set); every other argument as
// attn_w64q_kernel's.  Returns -1 for a combination this file does not instantiate.
// `total` = the launch's grid (a whole call: q blocks x heads x batches; a piece of a split call: see v_base / kv_split / carry_n at the kernel).
int wan_attention_w16n_launch(int fl, unsigned total, hipStream_t stream, const bf16_t* q, const bf16_t* k, const bf16_t* vt, bf16_t* o, int B,
                              int Bk, int64_t Lq, int64_t Lk, int64_t ldv, int H, int nqb, float scale_log2e, int nseg, int64_t k_seg_stride,
                              int64_t vt_seg_stride, const float* kmax2, int* wg_flags, float* raw, int skip_seg, int v_base, int kv_split,
                              int carry_n, int part_from) {
  if (part_from < 0) part_from = (int)total;
  const uint32_t* seg_tab = nullptr;
  if (fl & 64) {   // MULTI: the walk over the segments as a table (its offsets are relative to each workgroup's (batch, head) bases)
    switch (fl) { case 4 | 64: case 6 | 64: case 2 | 4 | 32 | 64: case 4 | 64 | 128: case 6 | 64 | 128: case 2 | 4 | 32 | 64 | 128: case 2 | 4 | 16 | 64 | 128: break; default: return -1; }
    if (int rc = seg_table(&seg_tab, nseg, skip_seg, (int)Lk, k_seg_stride * 2, vt_seg_stride * 2, (uint32_t)(H * 256), stream)) return rc;
  }
#define W16N_CASE(FL)                                                                                                              \
  case FL:                                                                                                                         \
    hipLaunchKernelGGL((attn_w16n_kernel<FL>), dim3(total), dim3(256), 0, stream, q, k, vt, o, B, Bk, Lq, Lk, ldv, H, nqb, scale_log2e, \
                       nseg, k_seg_stride, vt_seg_stride, kmax2, wg_flags, raw, skip_seg, seg_tab, v_base, kv_split, carry_n, part_from); \
    break;
  switch (fl) {
    W16N_CASE(4)
    W16N_CASE(6)
    W16N_CASE(4 | 64)
    W16N_CASE(6 | 64)
    W16N_CASE(2 | 4 | 16)
    W16N_CASE(2 | 4 | 32 | 64)
    W16N_CASE(4 | 128)
    W16N_CASE(6 | 128)
    W16N_CASE(4 | 64 | 128)
    W16N_CASE(6 | 64 | 128)
    W16N_CASE(2 | 4 | 16 | 128)
    W16N_CASE(2 | 4 | 32 | 64 | 128)
    W16N_CASE(2 | 4 | 16 | 64 | 128)   /* the parts of a split tail over segmented K / V^T (round 6) */
    W16N_CASE(2 | 4 | 32 | 128)        /* ... and the finishing launch of a split tail over one segment */
    W16N_CASE(4 | 128 | 256)
    W16N_CASE(6 | 128 | 256)
#ifdef W64Q_TIMING
    W16N_CASE(7)
#endif
    default: return -1;
  }
#undef W16N_CASE
  WAN_LAUNCH_CHECK();
  return 0;
}
