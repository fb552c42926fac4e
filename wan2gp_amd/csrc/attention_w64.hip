// Flash attention forward for gfx950, "4 x 64" structure: a workgroup is 4 waves = ONE wave per SIMD, each wave
// owns 64 q rows (two 32-row q-blocks a and b) and the whole 512-entry register file.
//
// Why: with 32 q rows per wave every K / V^T fragment byte read from LDS feeds one 32x32x16 MFMA, i.e. 32 KB of
// ds_read_b128 per 32 MFMAs per wave -- the LDS pipe is ~50 % busy at full MFMA rate and the measured ceiling of
// the 32-row kernels with softmax and DMA removed is only 74 % of the (DVFS) MFMA peak (profiles/
// r01_attn_variants.md, ablation 56).  With 64 rows per wave each fragment feeds TWO MFMAs: LDS traffic per
// flop is halved.  One wave per SIMD has no second wave to fill its MFMA shadows, so the wave interleaves its
// own streams -- the two q-blocks are independent online-softmax problems:
//
//   per 64-kv tile t (LDS stage ST of a 3-deep ring; K(t) fragments already in registers):
//     A:  S_a = K Q_a^T        16 MFMA   || 16 ds_read_b128 of V^T(t)
//     B:  S_b = K Q_b^T        16 MFMA   || softmax(a): max, exp2, pack          (VALU chunks between MFMAs)
//         -- rare uniform branch: rescale O_a, l_a if the running max moved --
//     C:  O_a += V^T P_a^T     16 MFMA   || softmax(b)
//         -- rare uniform branch: rescale O_b, l_b --
//     D:  O_b += V^T P_b^T     16 MFMA   || 16 ds_read_b128 of K(t+1)           (next tile's stage: landed and
//                                                                                  visible since this tile's barrier)
//   one s_waitcnt vmcnt(0) + s_barrier per tile; the LDS-DMA of tile t+2 is issued right after it.
//
// Math, HBM layouts and LDS images are those of attention.hip / attention_pp.hip (S^T = K Q^T, O^T = V^T P^T,
// V transposed in HBM, K rows bit-2/3 swapped, XOR-swizzled lane-linear LDS-DMA images).
// Registers (per lane): O 128, Q 64, K fragments 64, V^T fragments 64, S 64, P 32  (+ ~40 scalars/addresses).
#include <stdlib.h>
#include <string.h>

#include "attn_w64_shared.h"

namespace {

struct QBlock {      // one 32-row q-block of the wave
  f32x16 accO[4];    // O^T tiles: [d tile][..]
  f32x16 accL;       // ROWSUM: ones x P^T accumulator (every row = l[q]); else unused
  f32x16 s[2];       // S^T of the current tile: [kv sub-tile]
  uint32_t pk[2][8]; // P^T of the current tile as packed bf16: [kv sub-tile][..]
  float m_run, m_prev, l_run, mt, mb, lsum;
};

// ---- softmax chunks (each is placed between two MFMAs of the OTHER q-block) --------------------------------
// reg r of s[T] <-> kv = kv0 + T*32 + (r&7) + 8*half + 16*(r>>3); the lane's q row is lane&31.
// kv_rem = Lk - kv0 (valid kv rows left in the segment, counted from this tile's first row)
__device__ __forceinline__ void smx_mask_tail(QBlock& q, int kv_rem, int half) {
  if (__builtin_expect(kv_rem < KVBLK, 0)) {
    asm volatile("" ::: "memory");  // keep this rare path a real (wave-uniform) branch
    const int lim = kv_rem - 8 * half;
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        q.s[T][r] = (T * 32 + (r & 7) + 16 * (r >> 3) >= lim) ? -INFINITY : q.s[T][r];
        asm volatile("" : "+v"(q.s[T][r]));  // one compare + select at a time: no 32 live SGPR-pair masks
      }
  }
}
template <int T>
__device__ __forceinline__ void smx_max(QBlock& q) {
  float m = fmaxf(q.s[T][0], q.s[T][1]);
#pragma unroll
  for (int r = 2; r < 16; ++r) m = fmaxf(m, q.s[T][r]);
  q.mt = (T == 0) ? m : fmaxf(q.mt, m);
  asm volatile("" : "+v"(q.mt));  // pin the chunk here: without an opaque use LLVM sinks it into the block of its consumer
}
__device__ __forceinline__ void smx_xhalf(QBlock& q, float c) {
  const float m_new = fmaxf(q.m_run, xhalf_max(q.mt));
  q.m_prev = q.m_run;
  q.m_run = m_new;
  q.mb = m_new * c;
  q.lsum = 0.f;
  asm volatile("" : "+v"(q.mb), "+v"(q.m_run));
}
template <int T, int R0, bool ROWSUM>
__device__ __forceinline__ void smx_exp4(QBlock& q, float c) {
  const float p0 = __builtin_amdgcn_exp2f(q.s[T][R0] * c - q.mb), p1 = __builtin_amdgcn_exp2f(q.s[T][R0 + 1] * c - q.mb);
  const float p2 = __builtin_amdgcn_exp2f(q.s[T][R0 + 2] * c - q.mb), p3 = __builtin_amdgcn_exp2f(q.s[T][R0 + 3] * c - q.mb);
  if (!ROWSUM) q.lsum += (p0 + p1) + (p2 + p3);
  q.pk[T][R0 >> 1] = cvt_pk(p0, p1);
  q.pk[T][(R0 >> 1) + 1] = cvt_pk(p2, p3);
  if (ROWSUM) asm volatile("" : "+v"(q.pk[T][R0 >> 1]), "+v"(q.pk[T][(R0 >> 1) + 1]));
  else asm volatile("" : "+v"(q.pk[T][R0 >> 1]), "+v"(q.pk[T][(R0 >> 1) + 1]), "+v"(q.lsum));
}
template <bool ROWSUM>
__device__ __forceinline__ void rescale_if_moved(QBlock& q, float c) {
  if (__builtin_expect(!__all(q.m_run == q.m_prev), 0)) {  // running max moved: rescale O and l once, before P(t) enters O
    asm volatile("" ::: "memory");
    // opaque AGPR re-definitions on both sides keep the accumulator <-> VGPR copies INSIDE this rare branch (without
    // them the allocator hoists 128 v_accvgpr_read to the top of every tile)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) asm volatile("" : "+a"(q.accO[dt]));
    const float alpha = __builtin_amdgcn_exp2f((q.m_prev - q.m_run) * c);
    if (ROWSUM) q.accL[0] *= alpha; else q.l_run *= alpha;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) q.accO[dt][r] *= alpha;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) asm volatile("" : "+a"(q.accO[dt]));
  }
  if (!ROWSUM) q.l_run += q.lsum;
}
__device__ __forceinline__ mfma_bf16x8 pfrag(const QBlock& q, int c4) {
  uint4 pr;
  const int T = c4 >> 1, s = c4 & 1;
  pr.x = q.pk[T][s * 4 + 0]; pr.y = q.pk[T][s * 4 + 1]; pr.z = q.pk[T][s * 4 + 2]; pr.w = q.pk[T][s * 4 + 3];
  return __builtin_bit_cast(mfma_bf16x8, pr);
}

// the 11 softmax chunks of q-block X, issued one per MFMA gap: I = 0..10
template <int I, bool ROWSUM>
__device__ __forceinline__ void smx_chunk(QBlock& x, float c, int kv_rem, int half) {
  if (I == 0) { smx_mask_tail(x, kv_rem, half); smx_max<0>(x); }
  else if (I == 1) smx_max<1>(x);
  else if (I == 2) smx_xhalf(x, c);
  else if (I == 3) smx_exp4<0, 0, ROWSUM>(x, c);
  else if (I == 4) smx_exp4<0, 4, ROWSUM>(x, c);
  else if (I == 5) smx_exp4<0, 8, ROWSUM>(x, c);
  else if (I == 6) smx_exp4<0, 12, ROWSUM>(x, c);
  else if (I == 7) smx_exp4<1, 0, ROWSUM>(x, c);
  else if (I == 8) smx_exp4<1, 4, ROWSUM>(x, c);
  else if (I == 9) smx_exp4<1, 8, ROWSUM>(x, c);
  else if (I == 10) smx_exp4<1, 12, ROWSUM>(x, c);
}

// One 64-kv tile for both q-blocks.  kf = K(t) fragments (in registers on entry); on exit kf = the fragments of
// the NEXT ring stage (K(t+1); stale but harmless data after the last tile).  PIN: pin the hand-placed order with
// sched_barrier(0).  Register liveness is part of the placement: V^T fragments are read into the registers the K
// fragments vacate (k-steps 0,1 once S_b's first sub-tile is done, k-steps 2,3 at the start of C), K(t+1) into the
// registers S vacates, so the peak stays ~380 of the 512 registers.
template <int ST, bool ROWSUM, bool TIMING = false>
__device__ __forceinline__ void tile_w64(lds_cchar* smem, const int (&kaddr)[8], const int (&vaddr)[4],
                                         const mfma_bf16x8 (&qfa)[8], const mfma_bf16x8 (&qfb)[8], mfma_bf16x8 (&kf)[2][8],
                                         QBlock& a, QBlock& b, int kv_rem, int half, float c, char* smem_rw, Dma& dma,
                                         uint64_t* stamp = nullptr, bool rec = false) {
#define STAMP(K) do { if (TIMING && rec) stamp[K] = __builtin_amdgcn_s_memtime(); } while (0)
  uint4 oraw;
  oraw.x = oraw.y = oraw.z = oraw.w = 0x3f803f80u;  // bf16 1.0 x 8
  const mfma_bf16x8 ones = __builtin_bit_cast(mfma_bf16x8, oraw);
  constexpr int VB = ST * IMG, KN = ((ST + 1) % NST) * IMG;  // vaddr already includes the V^T region base
  mfma_bf16x8 vf[4][4];  // V^T(t) fragments: [16-kv k-step c4][d tile]
#define PINB() SB()
#define RDV(C4, DT) vf[C4][DT] = *(lds_frag*)(smem + (VB + (DT) * 4096) + vaddr[C4])
#define RDK(T, KS) kf[T][KS] = *(lds_frag*)(smem + (KN + (T) * 8192) + kaddr[KS])
#define SMX(X, I) smx_chunk<I, ROWSUM>(X, c, kv_rem, half)

  // ---- A: S_a = K Q_a^T (the two kv sub-tile chains alternate: dependent MFMAs are 2 apart)
#define QK(X, QF, I) do { if ((I) < 2) mfma_qk0(X.s[(I) & 1], kf[(I) & 1][(I) >> 1], QF[(I) >> 1]); \
                          else mfma_qk(X.s[(I) & 1], kf[(I) & 1][(I) >> 1], QF[(I) >> 1]); } while (0)
  // the LDS-DMA of tile t+2 (ring stage (ST+2)%3, free since this tile's barrier) is issued in the MFMA shadows
  constexpr int DST = (ST + 2) % NST;
  QK(a, qfa, 0); PINB(); dma_piece<0, DST>(smem_rw, dma); PINB();
  QK(a, qfa, 1); PINB(); dma_piece<4, DST>(smem_rw, dma); PINB();
  QK(a, qfa, 2); PINB(); dma_piece<1, DST>(smem_rw, dma); PINB();
  QK(a, qfa, 3); PINB(); dma_piece<5, DST>(smem_rw, dma); PINB();
  QK(a, qfa, 4); PINB(); dma_piece<2, DST>(smem_rw, dma); PINB();
  QK(a, qfa, 5); PINB(); dma_piece<6, DST>(smem_rw, dma); PINB();
  QK(a, qfa, 6); PINB(); dma_piece<3, DST>(smem_rw, dma); PINB();
  QK(a, qfa, 7); PINB(); dma_piece<7, DST>(smem_rw, dma); PINB();
  QK(a, qfa, 8); PINB(); dma_advance(dma); PINB();
  QK(a, qfa, 9); QK(a, qfa, 10); QK(a, qfa, 11); QK(a, qfa, 12); QK(a, qfa, 13); QK(a, qfa, 14); QK(a, qfa, 15);
  SB();
  STAMP(2);

  // ---- B: S_b = K Q_b^T  ||  softmax(a)  ||  V^T k-steps 0,1 into the registers vacated by K k-steps 0..3
  QK(b, qfb, 0); PINB();
  QK(b, qfb, 1); PINB();
  QK(b, qfb, 2); PINB();
  SMX(a, 0); PINB();
  QK(b, qfb, 3); PINB();
  SMX(a, 1); PINB();
  QK(b, qfb, 4); PINB();
  SMX(a, 2); PINB();
  QK(b, qfb, 5); PINB();
  SMX(a, 3); PINB();
  QK(b, qfb, 6); PINB();
  SMX(a, 4); PINB();
  QK(b, qfb, 7); PINB();
  SMX(a, 5); PINB();
  QK(b, qfb, 8); PINB();
  RDV(0, 0); RDV(0, 1); SMX(a, 6); PINB();
  QK(b, qfb, 9); PINB();
  RDV(0, 2); RDV(0, 3); SMX(a, 7); PINB();
  QK(b, qfb, 10); PINB();
  RDV(1, 0); RDV(1, 1); SMX(a, 8); PINB();
  QK(b, qfb, 11); PINB();
  RDV(1, 2); RDV(1, 3); SMX(a, 9); PINB();
  QK(b, qfb, 12); PINB();
  SMX(a, 10); PINB();
  QK(b, qfb, 13);
  QK(b, qfb, 14);
  QK(b, qfb, 15);
#undef QK
  SB();
  rescale_if_moved<ROWSUM>(a, c);
  STAMP(3);

  // ---- C: O_a += V^T P_a^T  ||  softmax(b)  ||  V^T k-steps 2,3 into the vacated K sub-tile 1 registers
  {
    const mfma_bf16x8 p0 = pfrag(a, 0), p1 = pfrag(a, 1), p2 = pfrag(a, 2), p3 = pfrag(a, 3);
    MF(a.accO[0], vf[0][0], p0); PINB();
    RDV(2, 0); RDV(2, 1); PINB();
    MF(a.accO[1], vf[0][1], p0); PINB();
    RDV(2, 2); RDV(2, 3); PINB();
    MF(a.accO[2], vf[0][2], p0); PINB();
    RDV(3, 0); RDV(3, 1); SMX(b, 0); PINB();
    MF(a.accO[3], vf[0][3], p0); PINB();
    RDV(3, 2); RDV(3, 3); SMX(b, 1); PINB();
    MF(a.accO[0], vf[1][0], p1); PINB();
    SMX(b, 2); PINB();
    MF(a.accO[1], vf[1][1], p1); PINB();
    SMX(b, 3); PINB();
    MF(a.accO[2], vf[1][2], p1); PINB();
    SMX(b, 4); PINB();
    MF(a.accO[3], vf[1][3], p1); PINB();
    SMX(b, 5); PINB();
    MF(a.accO[0], vf[2][0], p2); PINB();
    SMX(b, 6); PINB();
    MF(a.accO[1], vf[2][1], p2); PINB();
    SMX(b, 7); PINB();
    MF(a.accO[2], vf[2][2], p2); PINB();
    SMX(b, 8); PINB();
    MF(a.accO[3], vf[2][3], p2); PINB();
    SMX(b, 9); PINB();
    MF(a.accO[0], vf[3][0], p3); PINB();
    SMX(b, 10); PINB();
    MF(a.accO[1], vf[3][1], p3);
    MF(a.accO[2], vf[3][2], p3);
    MF(a.accO[3], vf[3][3], p3);
    if (ROWSUM) { MF(a.accL, ones, p0); MF(a.accL, ones, p1); MF(a.accL, ones, p2); MF(a.accL, ones, p3); }
  }
  SB();
  rescale_if_moved<ROWSUM>(b, c);
  STAMP(4);

  // ---- D: O_b += V^T P_b^T  ||  K fragments of the next ring stage (landed and visible since this tile's barrier)
  {
    const mfma_bf16x8 p0 = pfrag(b, 0), p1 = pfrag(b, 1), p2 = pfrag(b, 2), p3 = pfrag(b, 3);
    MF(b.accO[0], vf[0][0], p0); PINB(); RDK(0, 0); PINB();
    MF(b.accO[1], vf[0][1], p0); PINB(); RDK(0, 1); PINB();
    MF(b.accO[2], vf[0][2], p0); PINB(); RDK(0, 2); PINB();
    MF(b.accO[3], vf[0][3], p0); PINB(); RDK(0, 3); PINB();
    MF(b.accO[0], vf[1][0], p1); PINB(); RDK(0, 4); PINB();
    MF(b.accO[1], vf[1][1], p1); PINB(); RDK(0, 5); PINB();
    MF(b.accO[2], vf[1][2], p1); PINB(); RDK(0, 6); PINB();
    MF(b.accO[3], vf[1][3], p1); PINB(); RDK(0, 7); PINB();
    MF(b.accO[0], vf[2][0], p2); PINB(); RDK(1, 0); PINB();
    MF(b.accO[1], vf[2][1], p2); PINB(); RDK(1, 1); PINB();
    MF(b.accO[2], vf[2][2], p2); PINB(); RDK(1, 2); PINB();
    MF(b.accO[3], vf[2][3], p2); PINB(); RDK(1, 3); PINB();
    MF(b.accO[0], vf[3][0], p3); PINB(); RDK(1, 4); PINB();
    MF(b.accO[1], vf[3][1], p3); PINB(); RDK(1, 5); PINB();
    MF(b.accO[2], vf[3][2], p3); PINB(); RDK(1, 6); PINB();
    MF(b.accO[3], vf[3][3], p3); PINB(); RDK(1, 7); PINB();
    if (ROWSUM) { MF(b.accL, ones, p0); MF(b.accL, ones, p1); MF(b.accL, ones, p2); MF(b.accL, ones, p3); }
  }
  SB();
  STAMP(5);
#undef STAMP
#undef PINB
#undef RDV
#undef RDK
#undef SMX
}

// FLAGS bit0: row sums on the matrix pipe (ones x P^T); bit2: s_memtime stamps (tuning aid)
template <int FLAGS>
__global__ __launch_bounds__(256) void attn_w64_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ Kg,
                                                      const bf16_t* __restrict__ Vt, bf16_t* __restrict__ O, int B, int Bk,
                                                      int64_t Lq, int64_t Lk, int64_t ldv, int H, int nqb, float scale_log2e,
                                                      int nseg, int64_t k_seg_stride, int64_t vt_seg_stride) {
  constexpr bool ROWSUM = (FLAGS & 1) != 0;
  constexpr bool TIMING = (FLAGS & 4) != 0;  // tuning aid: s_memtime stamps of tile 300 of workgroup 0 -> first 48 B of O
  uint64_t stamp[6] = {0, 0, 0, 0, 0, 0};
  __shared__ __attribute__((aligned(16))) char smem[2 * NST * IMG];  // [K stage 0..2][V^T stage 0..2] = 96 KB
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
  mfma_bf16x8 qfa[8], qfb[8];
  {
    int64_t ra = q0 + l31, rb = q0 + 32 + l31;
    if (ra > Lq - 1) ra = Lq - 1;
    if (rb > Lq - 1) rb = Lq - 1;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      qfa[ks] = *reinterpret_cast<const mfma_bf16x8*>(qbase + ra * rs + ks * 16 + half * 8);
      qfb[ks] = *reinterpret_cast<const mfma_bf16x8*>(qbase + rb * rs + ks * 16 + half * 8);
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

  QBlock qa, qbk;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { qa.accO[dt][r] = 0.f; qbk.accO[dt][r] = 0.f; }
#pragma unroll
  for (int r = 0; r < 16; ++r) { qa.accL[r] = 0.f; qbk.accL[r] = 0.f; }
  qa.m_run = qbk.m_run = -INFINITY;
  qa.l_run = qbk.l_run = 0.f;

  dma_tile<0>(smem, dma);
  dma_tile<1>(smem, dma);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  mfma_bf16x8 kf[2][8];
#pragma unroll
  for (int T = 0; T < 2; ++T)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) kf[T][ks] = *(lds_frag*)(lds + T * 8192 + kaddr[ks]);

#define W64_STEP(J)                                                                                          \
  if (t + (J) < ntile) {                                                                                     \
    const bool rec = TIMING && (t + (J) == 300);                                                             \
    if (TIMING && rec) stamp[0] = __builtin_amdgcn_s_memtime();                                              \
    if (t + (J) > 0) {                                                                                       \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                       \
      __builtin_amdgcn_s_barrier();                                                                          \
      asm volatile("" ::: "memory");                                                                         \
    }                                                                                                        \
    if (TIMING && rec) stamp[1] = __builtin_amdgcn_s_memtime();                                              \
    const int kv_rem = next_kv_rem();                                                                        \
    tile_w64<J, ROWSUM, TIMING>(lds, kaddr, vaddr, qfa, qfb, kf, qa, qbk, kv_rem, half, scale_log2e, smem, dma, stamp, rec); \
  }
  for (int t = 0; t < ntile; t += 3) {
    W64_STEP(0)
    W64_STEP(1)
    W64_STEP(2)
  }
#undef W64_STEP
  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");  // trailing DMA lands before O staging reuses LDS; last PV MFMAs -> accumulator reads below (asm MFMAs are not padded)
  // ---- epilogue: normalise, stage the wave's 64 x 128 O tile through LDS, store whole rows -------------------
  const float inva = 1.0f / (ROWSUM ? qa.accL[0] : qa.l_run + __shfl_xor(qa.l_run, 32, 64));
  const float invb = 1.0f / (ROWSUM ? qbk.accL[0] : qbk.l_run + __shfl_xor(qbk.l_run, 32, 64));
  __syncthreads();
  char* ob = smem + wave * (64 * 256);
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    const QBlock& x = blk ? qbk : qa;
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
    for (int k6 = 0; k6 < 6; ++k6) reinterpret_cast<uint64_t*>(O)[k6] = stamp[k6];
  }
}

}  // namespace

// called from attention.hip's dispatcher.  flags bit0: row sums on the matrix pipe; bit2: s_memtime stamps
int wan_attention_w64_launch(int flags, const bf16_t* q, const bf16_t* k, const bf16_t* vt, bf16_t* o, int B, int Bk,
                             int64_t Lq, int64_t Lk, int64_t ldv, int H, int nseg, int64_t k_seg_stride,
                             int64_t vt_seg_stride, float scale_log2e, hipStream_t stream) {
  WAN_REQUIRE(Lk * (int64_t)H * 256 < ((int64_t)1 << 32) && ldv * 256 < ((int64_t)1 << 32),
              "wan_attention: K/V^T extent exceeds the 32-bit DMA offsets of this kernel");
  const int64_t nqb = (Lq + 255) / 256;
  const int64_t total = nqb * H * B;
  WAN_REQUIRE(total < ((int64_t)1 << 31), "wan_attention: grid too large");
#define W64_LAUNCH(FL)                                                                                              \
  hipLaunchKernelGGL((attn_w64_kernel<FL>), dim3((unsigned)total), dim3(256), 0, stream, q, k, vt, o, B, Bk, Lq, Lk, \
                     ldv, H, (int)nqb, scale_log2e, nseg, k_seg_stride, vt_seg_stride)
  if (flags & 4) W64_LAUNCH(4); else if (flags & 1) W64_LAUNCH(1); else W64_LAUNCH(0);
#undef W64_LAUNCH
  WAN_LAUNCH_CHECK();
  return 0;
}
