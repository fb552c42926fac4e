// "Ping-pong" flash attention forward for gfx950: 8 waves (256 q rows) per workgroup, the two waves
// that share a SIMD run HALF A TILE OUT OF PHASE so that one is always in an MFMA-only phase
// while the other does its softmax VALU work (MI355X_MICROARCH.md "Two waves per SIMD": the matrix
// pipe is per SIMD, VALU and MFMA of different waves overlap).
//
// Same math, layouts and LDS images as attention.hip (S^T = K Q^T, O^T = V^T P^T, V transposed in
// HBM, K rows bit-2/3 swapped, XOR-swizzled LDS-DMA images); what changes is the schedule:
//
//   group G0 = waves 0-3, group G1 = waves 4-7 (wave i and i+4 share a SIMD)
//   iteration t:   --bar1--  P1:  G0: QK^T(t) + softmax(t)      G1: PV(t-1)
//                  --bar2--  P2:  G0: PV(t)                     G1: QK^T(t) + softmax(t)
//   K(t+1) is DMA'd at the start of P1(t), V^T(t+1) at the start of P2(t); the waits are COUNTED
//   (vmcnt(2): the newer image's two 16-B pieces per lane stay in flight across the barrier).
//   One extra iteration drains G1's last PV.  The tile loop is unrolled by 2 so every LDS address
//   is (precomputed per-lane VGPR) + (compile-time immediate): 12 address VGPRs, no per-tile
//   address arithmetic; DMA source addresses are uniform 64-bit bases + 32-bit per-lane offsets.
#include <stdlib.h>
#include <string.h>

#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 hw_bf16x2;
typedef __attribute__((ext_vector_type(2))) float hw_f32x2;

#define KVBLK 64
#define IMG 16384  // bytes per K or V^T image

__device__ __forceinline__ uint32_t cvt_pk_bf16_pp(float lo, float hi) {
  hw_f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hw_bf16x2));
}
__device__ __forceinline__ void glds16p(const void* gsrc, void* ldst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)ldst, 16, 0, 0);
}

typedef __attribute__((address_space(3))) const char lds_cchar;
typedef __attribute__((address_space(3))) const mfma_bf16x8 lds_frag;

struct PPState {
  f32x16 accO[4];
  f32x16 accL;  // row sums of P on the matrix pipe: accL = ones(32 x kv) * P^T, every row = l[q] (only [0] is read)
  uint32_t pk[2][8];
  float m_run;
  float l_run;  // per-lane partial row sum (used when the row sums are NOT taken on the matrix pipe)
};

// S^T = K Q^T for one 64-kv tile held in LDS stage ST, then the online-softmax update of `st`
// LDS map (NST stages): K images at [ST*IMG), V^T images at [NST*IMG + ST*IMG): every fragment address is
// a per-lane VGPR + an immediate < 64 KiB for NST <= 3.
//
// Schedule inside a tile (one wave).  hipcc left to itself issues each ds_read one MFMA ahead of its use
// and waits lgkmcnt(0) before every MFMA (LDS latency > MFMA issue interval => ~3x the MFMA time per tile,
// SQ_WAIT_ANY 31-44%).  The fragment reads are therefore batched explicitly and pinned with
// sched_barrier(0) fences:
//   [16 K reads] | [16 QK^T MFMAs] [8 V^T reads (T=0) issued under them] | [softmax VALU] |
//   [8 V^T reads (T=1)] [8 PV MFMAs (T=0)] [8 PV MFMAs (T=1)]
struct VFrags {
  mfma_bf16x8 f[8];  // V^T fragments of kv sub-tile T=0: [s][dt]
};

template <int ST, bool PRIO, int NST, bool ROWSUM_MFMA = true, bool ABL_NOSM = false>
__device__ __forceinline__ void qk_softmax(lds_cchar* smem, const int (&kaddr)[8], const int (&vaddr)[4],
                                           const mfma_bf16x8 (&qf)[8], PPState& st, VFrags& vf, int64_t kv0,
                                           int64_t Lk, int half, float scale_log2e) {
  f32x16 accS[2];
  mfma_bf16x8 kf[2][8];
#pragma unroll
  for (int T = 0; T < 2; ++T)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
      kf[T][ks] = *(lds_frag*)(smem + (ST * IMG + T * 8192) + kaddr[ks]);
  __builtin_amdgcn_sched_barrier(0);
  if (PRIO) __builtin_amdgcn_s_setprio(1);
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int T = 0; T < 2; ++T) {
    accS[T] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[T][0], qf[0], zero16, 0, 0, 0);
#pragma unroll
    for (int ks = 1; ks < 8; ++ks) accS[T] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[T][ks], qf[ks], accS[T], 0, 0, 0);
  }
  if (PRIO) __builtin_amdgcn_s_setprio(0);
  __builtin_amdgcn_sched_barrier(0);
  if (ABL_NOSM) {  // timing ablation only (wrong results): no max / exp, just the bf16 packing
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
      for (int r = 0; r < 16; r += 2) st.pk[T][r >> 1] = cvt_pk_bf16_pp(accS[T][r], accS[T][r + 1]);
    return;
  }
  if (kv0 + KVBLK > Lk) {  // ragged tail of a segment: reg r <-> kv = kv0 + T*32 + (r&7) + 8*half + 16*(r>>3)
    asm volatile("" ::: "memory");  // keep this rare path a real (wave-uniform) branch: no if-conversion
    const int lim = (int)(Lk - kv0) - 8 * half;
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (T * 32 + (r & 7) + 16 * (r >> 3) >= lim) accS[T][r] = -INFINITY;
  }
  float mt = accS[0][0];
#pragma unroll
  for (int T = 0; T < 2; ++T)
#pragma unroll
    for (int r = 0; r < 16; ++r) mt = fmaxf(mt, accS[T][r]);
  mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
  const float m_new = fmaxf(st.m_run, mt);
  const float mb = m_new * scale_log2e;
  float psum = 0.f;
#pragma unroll
  for (int T = 0; T < 2; ++T)
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const float p0 = __builtin_amdgcn_exp2f(accS[T][r] * scale_log2e - mb);
      const float p1 = __builtin_amdgcn_exp2f(accS[T][r + 1] * scale_log2e - mb);
      if (!ROWSUM_MFMA) psum += p0 + p1;
      st.pk[T][r >> 1] = cvt_pk_bf16_pp(p0, p1);
    }
  if (!__all(m_new == st.m_run)) {  // running max moved: rescale O and l once, before P(t) enters O
    const float alpha = __builtin_amdgcn_exp2f((st.m_run - m_new) * scale_log2e);
    if (ROWSUM_MFMA) st.accL[0] *= alpha; else st.l_run *= alpha;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) st.accO[dt][r] *= alpha;
  }
  if (!ROWSUM_MFMA) st.l_run += psum;
  st.m_run = m_new;
}

// O^T += V^T P^T with the V^T tile in LDS stage ST, plus the row sums l += ones * P^T on the matrix pipe (the
// softmax VALU stream is the co-bottleneck: 4 extra MFMAs per tile replace 32 v_add_f32 per lane).
template <int ST, int NST, bool PRELOADED, bool ROWSUM_MFMA = true>
__device__ __forceinline__ void pv(lds_cchar* smem, const int (&vaddr)[4], PPState& st, VFrags& vf) {
  mfma_bf16x8 vg[8];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      vf.f[s * 4 + dt] = *(lds_frag*)(smem + (NST * IMG + ST * IMG + dt * 4096) + vaddr[s]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      vg[s * 4 + dt] = *(lds_frag*)(smem + (NST * IMG + ST * IMG + dt * 4096) + vaddr[2 + s]);
  uint4 oraw;
  oraw.x = oraw.y = oraw.z = oraw.w = 0x3f803f80u;  // bf16 1.0 x 8
  const mfma_bf16x8 ones = __builtin_bit_cast(mfma_bf16x8, oraw);
#pragma unroll
  for (int T = 0; T < 2; ++T)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      uint4 praw;
      praw.x = st.pk[T][s * 4 + 0]; praw.y = st.pk[T][s * 4 + 1];
      praw.z = st.pk[T][s * 4 + 2]; praw.w = st.pk[T][s * 4 + 3];
      const mfma_bf16x8 pf = __builtin_bit_cast(mfma_bf16x8, praw);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        st.accO[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(T == 0 ? vf.f[s * 4 + dt] : vg[s * 4 + dt], pf, st.accO[dt], 0, 0, 0);
      if (ROWSUM_MFMA) st.accL = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pf, st.accL, 0, 0, 0);
    }
  __builtin_amdgcn_sched_barrier(0);
}

// ---- software-pipelined tile (non-tail tiles of MODE 0/2) -------------------------------------------------
// The 64-kv tile is processed as two 32-kv online-softmax sub-steps whose VALU work is placed UNDER the
// other sub-step's MFMAs inside one basic block (an in-order wave overlaps MFMA and VALU only when they are
// interleaved in its instruction stream):
//   B1: QK^T(T0)                        8 MFMA
//   B2: QK^T(T1)  ||  softmax(T0)       8 MFMA + ~60 VALU     (+ V^T(T0) fragment reads)
//   -- rare uniform branch: rescale O,l if the running max moved --
//   B3: PV(T0)    ||  softmax(T1)       8(+2) MFMA + ~60 VALU (+ V^T(T1) fragment reads)
//   -- rare uniform branch --
//   B4: PV(T1)                          8(+2) MFMA
// The cross-half row-max exchange uses v_permlane32_swap (VALU) instead of ds_bpermute so no LDS round trip
// sits in the softmax chain.  sched_group_barrier pins the 1 MFMA : N VALU interleave.
__device__ __forceinline__ float xhalf_max(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// max / exp / pack of one 32-kv sub-tile; returns alpha-needed flag via m_old/m_new in st
template <bool ROWSUM_MFMA>
__device__ __forceinline__ void softmax_half(const f32x16& accS, uint32_t (&pk)[8], float& m_run, float& m_prev,
                                             float& lsum, float scale_log2e) {
  float mt = fmaxf(accS[0], accS[1]);
#pragma unroll
  for (int r = 2; r < 16; ++r) mt = fmaxf(mt, accS[r]);
  mt = xhalf_max(mt);
  const float m_new = fmaxf(m_run, mt);
  const float mb = m_new * scale_log2e;
  float ps = 0.f;
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    const float p0 = __builtin_amdgcn_exp2f(accS[r] * scale_log2e - mb);
    const float p1 = __builtin_amdgcn_exp2f(accS[r + 1] * scale_log2e - mb);
    if (!ROWSUM_MFMA) ps += p0 + p1;
    pk[r >> 1] = cvt_pk_bf16_pp(p0, p1);
  }
  if (!ROWSUM_MFMA) lsum = ps;
  m_prev = m_run;
  m_run = m_new;
}

template <bool ROWSUM_MFMA>
__device__ __forceinline__ void rescale_if_moved(PPState& st, float m_prev, float scale_log2e) {
  if (!__all(st.m_run == m_prev)) {
    asm volatile("" ::: "memory");
    const float alpha = __builtin_amdgcn_exp2f((m_prev - st.m_run) * scale_log2e);
    if (ROWSUM_MFMA) st.accL[0] *= alpha; else st.l_run *= alpha;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) st.accO[dt][r] *= alpha;
  }
}

template <int ST, int NST, bool ROWSUM_MFMA>
__device__ __forceinline__ void tile_fast(lds_cchar* smem, const int (&kaddr)[8], const int (&vaddr)[4],
                                          const mfma_bf16x8 (&qf)[8], PPState& st, float scale_log2e) {
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  uint4 oraw;
  oraw.x = oraw.y = oraw.z = oraw.w = 0x3f803f80u;  // bf16 1.0 x 8
  const mfma_bf16x8 ones = __builtin_bit_cast(mfma_bf16x8, oraw);
  mfma_bf16x8 kf0[8], kf1[8], va0[4], va1[4], vb0[4], vb1[4];  // V^T fragments: [kv sub-tile a/b][k-step 0/1][dt]
  f32x16 accS0, accS1;
  float m_prev, ls0 = 0.f, ls1 = 0.f;
  constexpr int KB = ST * IMG, VB = NST * IMG + ST * IMG;
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) kf0[ks] = *(lds_frag*)(smem + KB + kaddr[ks]);
  __builtin_amdgcn_sched_barrier(0);
  // ---- B1: QK^T(T0), K(T1) fragments in flight
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) kf1[ks] = *(lds_frag*)(smem + (KB + 8192) + kaddr[ks]);
  accS0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf0[0], qf[0], zero16, 0, 0, 0);
#pragma unroll
  for (int ks = 1; ks < 8; ++ks) accS0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf0[ks], qf[ks], accS0, 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  // ---- B2: QK^T(T1) || softmax(T0)
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) va0[dt] = *(lds_frag*)(smem + (VB + dt * 4096) + vaddr[0]);
  accS1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf1[0], qf[0], zero16, 0, 0, 0);
#pragma unroll
  for (int ks = 1; ks < 8; ++ks) accS1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf1[ks], qf[ks], accS1, 0, 0, 0);
  softmax_half<ROWSUM_MFMA>(accS0, st.pk[0], st.m_run, m_prev, ls0, scale_log2e);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
    __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);  // 8 VALU
  }
  __builtin_amdgcn_sched_barrier(0);
  rescale_if_moved<ROWSUM_MFMA>(st, m_prev, scale_log2e);
  if (!ROWSUM_MFMA) st.l_run += ls0;
  // ---- B3: PV(T0) || softmax(T1)
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) va1[dt] = *(lds_frag*)(smem + (VB + dt * 4096) + vaddr[1]);
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) vb0[dt] = *(lds_frag*)(smem + (VB + dt * 4096) + vaddr[2]);
  {
    uint4 praw;
    praw.x = st.pk[0][0]; praw.y = st.pk[0][1]; praw.z = st.pk[0][2]; praw.w = st.pk[0][3];
    const mfma_bf16x8 pf = __builtin_bit_cast(mfma_bf16x8, praw);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) st.accO[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va0[dt], pf, st.accO[dt], 0, 0, 0);
    if (ROWSUM_MFMA) st.accL = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pf, st.accL, 0, 0, 0);
    praw.x = st.pk[0][4]; praw.y = st.pk[0][5]; praw.z = st.pk[0][6]; praw.w = st.pk[0][7];
    const mfma_bf16x8 pg = __builtin_bit_cast(mfma_bf16x8, praw);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) st.accO[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va1[dt], pg, st.accO[dt], 0, 0, 0);
    if (ROWSUM_MFMA) st.accL = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pg, st.accL, 0, 0, 0);
  }
  softmax_half<ROWSUM_MFMA>(accS1, st.pk[1], st.m_run, m_prev, ls1, scale_log2e);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  rescale_if_moved<ROWSUM_MFMA>(st, m_prev, scale_log2e);
  if (!ROWSUM_MFMA) st.l_run += ls1;
  // ---- B4: PV(T1)
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) vb1[dt] = *(lds_frag*)(smem + (VB + dt * 4096) + vaddr[3]);
  {
    uint4 praw;
    praw.x = st.pk[1][0]; praw.y = st.pk[1][1]; praw.z = st.pk[1][2]; praw.w = st.pk[1][3];
    const mfma_bf16x8 pf = __builtin_bit_cast(mfma_bf16x8, praw);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) st.accO[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb0[dt], pf, st.accO[dt], 0, 0, 0);
    if (ROWSUM_MFMA) st.accL = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pf, st.accL, 0, 0, 0);
    praw.x = st.pk[1][4]; praw.y = st.pk[1][5]; praw.z = st.pk[1][6]; praw.w = st.pk[1][7];
    const mfma_bf16x8 pg = __builtin_bit_cast(mfma_bf16x8, praw);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) st.accO[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb1[dt], pg, st.accO[dt], 0, 0, 0);
    if (ROWSUM_MFMA) st.accL = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pg, st.accL, 0, 0, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
}

// ---- tile_fast2: the same sub-tile pipeline with a HAND-PLACED interleave ---------------------------------
// hipcc's scheduler did not honour the sched_group_barrier pattern of tile_fast (it clustered the 8 MFMAs in
// front of the softmax VALU), so here every MFMA is followed by one small softmax chunk and the order is
// pinned with sched_barrier(0) after each item:  M max M xhalf M exp M exp M exp M exp M M
#define SB() __builtin_amdgcn_sched_barrier(0)
struct SmxState {
  float mt, mb, m_prev, lsum;
};
__device__ __forceinline__ void smx_max(const f32x16& a, SmxState& x) {
  float m = fmaxf(a[0], a[1]);
#pragma unroll
  for (int r = 2; r < 16; ++r) m = fmaxf(m, a[r]);
  x.mt = m;
}
__device__ __forceinline__ void smx_xhalf(SmxState& x, float& m_run, float c) {
  const float m_new = fmaxf(m_run, xhalf_max(x.mt));
  x.m_prev = m_run;
  m_run = m_new;
  x.mb = m_new * c;
  x.lsum = 0.f;
}
template <int R0, bool ROWSUM_MFMA>
__device__ __forceinline__ void smx_exp4(const f32x16& a, uint32_t (&pk)[8], SmxState& x, float c) {
  const float p0 = __builtin_amdgcn_exp2f(a[R0] * c - x.mb), p1 = __builtin_amdgcn_exp2f(a[R0 + 1] * c - x.mb);
  const float p2 = __builtin_amdgcn_exp2f(a[R0 + 2] * c - x.mb), p3 = __builtin_amdgcn_exp2f(a[R0 + 3] * c - x.mb);
  if (!ROWSUM_MFMA) x.lsum += (p0 + p1) + (p2 + p3);
  pk[R0 >> 1] = cvt_pk_bf16_pp(p0, p1);
  pk[(R0 >> 1) + 1] = cvt_pk_bf16_pp(p2, p3);
}

template <int ST, int NST, bool ROWSUM_MFMA>
__device__ __forceinline__ void tile_fast2(lds_cchar* smem, const int (&kaddr)[8], const int (&vaddr)[4],
                                           const mfma_bf16x8 (&qf)[8], PPState& st, float c) {
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  uint4 oraw;
  oraw.x = oraw.y = oraw.z = oraw.w = 0x3f803f80u;
  const mfma_bf16x8 ones = __builtin_bit_cast(mfma_bf16x8, oraw);
  mfma_bf16x8 kf0[8], kf1[8], va0[4], va1[4], vb0[4], vb1[4];
  f32x16 s0, s1;
  SmxState x0, x1;
  constexpr int KB = ST * IMG, VB = NST * IMG + ST * IMG;
#define MF(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0)
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) kf0[ks] = *(lds_frag*)(smem + KB + kaddr[ks]);
  SB();
  // B1: QK^T(T0); K(T1) fragment reads issued underneath
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) kf1[ks] = *(lds_frag*)(smem + (KB + 8192) + kaddr[ks]);
  s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf0[0], qf[0], zero16, 0, 0, 0);
#pragma unroll
  for (int ks = 1; ks < 8; ++ks) MF(s0, kf0[ks], qf[ks]);
  SB();
  // B2: QK^T(T1) with softmax(T0) chunks between the MFMAs
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) va0[dt] = *(lds_frag*)(smem + (VB + dt * 4096) + vaddr[0]);
  s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf1[0], qf[0], zero16, 0, 0, 0); SB();
  smx_max(s0, x0); SB();
  MF(s1, kf1[1], qf[1]); SB();
  smx_xhalf(x0, st.m_run, c); SB();
  MF(s1, kf1[2], qf[2]); SB();
  smx_exp4<0, ROWSUM_MFMA>(s0, st.pk[0], x0, c); SB();
  MF(s1, kf1[3], qf[3]); SB();
  smx_exp4<4, ROWSUM_MFMA>(s0, st.pk[0], x0, c); SB();
  MF(s1, kf1[4], qf[4]); SB();
  smx_exp4<8, ROWSUM_MFMA>(s0, st.pk[0], x0, c); SB();
  MF(s1, kf1[5], qf[5]); SB();
  smx_exp4<12, ROWSUM_MFMA>(s0, st.pk[0], x0, c); SB();
  MF(s1, kf1[6], qf[6]);
  MF(s1, kf1[7], qf[7]); SB();
  rescale_if_moved<ROWSUM_MFMA>(st, x0.m_prev, c);
  if (!ROWSUM_MFMA) st.l_run += x0.lsum;
  // B3: PV(T0) with softmax(T1) chunks between the MFMAs
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) va1[dt] = *(lds_frag*)(smem + (VB + dt * 4096) + vaddr[1]);
  {
    uint4 pr;
    pr.x = st.pk[0][0]; pr.y = st.pk[0][1]; pr.z = st.pk[0][2]; pr.w = st.pk[0][3];
    const mfma_bf16x8 pf = __builtin_bit_cast(mfma_bf16x8, pr);
    pr.x = st.pk[0][4]; pr.y = st.pk[0][5]; pr.z = st.pk[0][6]; pr.w = st.pk[0][7];
    const mfma_bf16x8 pg = __builtin_bit_cast(mfma_bf16x8, pr);
    MF(st.accO[0], va0[0], pf); SB();
    smx_max(s1, x1); SB();
    MF(st.accO[1], va0[1], pf); SB();
    smx_xhalf(x1, st.m_run, c); SB();
    MF(st.accO[2], va0[2], pf); SB();
    smx_exp4<0, ROWSUM_MFMA>(s1, st.pk[1], x1, c); SB();
    MF(st.accO[3], va0[3], pf); SB();
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) vb0[dt] = *(lds_frag*)(smem + (VB + dt * 4096) + vaddr[2]);  // va0 is dead now
    smx_exp4<4, ROWSUM_MFMA>(s1, st.pk[1], x1, c); SB();
    MF(st.accO[0], va1[0], pg); SB();
    smx_exp4<8, ROWSUM_MFMA>(s1, st.pk[1], x1, c); SB();
    MF(st.accO[1], va1[1], pg); SB();
    smx_exp4<12, ROWSUM_MFMA>(s1, st.pk[1], x1, c); SB();
    MF(st.accO[2], va1[2], pg);
    MF(st.accO[3], va1[3], pg);
    if (ROWSUM_MFMA) { MF(st.accL, ones, pf); MF(st.accL, ones, pg); }
    SB();
  }
  // NOTE: PV(T0) used the max as of sub-tile T0; the T1 rescale below multiplies O (which now includes P0 V0,
  // exponentiated against m after T0) by exp2((m_T0 - m_T1) c): exactly once, before P1 V1 is added.
  rescale_if_moved<ROWSUM_MFMA>(st, x1.m_prev, c);
  if (!ROWSUM_MFMA) st.l_run += x1.lsum;
  // B4: PV(T1)
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) vb1[dt] = *(lds_frag*)(smem + (VB + dt * 4096) + vaddr[3]);
  {
    uint4 pr;
    pr.x = st.pk[1][0]; pr.y = st.pk[1][1]; pr.z = st.pk[1][2]; pr.w = st.pk[1][3];
    const mfma_bf16x8 pf = __builtin_bit_cast(mfma_bf16x8, pr);
    pr.x = st.pk[1][4]; pr.y = st.pk[1][5]; pr.z = st.pk[1][6]; pr.w = st.pk[1][7];
    const mfma_bf16x8 pg = __builtin_bit_cast(mfma_bf16x8, pr);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) MF(st.accO[dt], vb0[dt], pf);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) MF(st.accO[dt], vb1[dt], pg);
    if (ROWSUM_MFMA) { MF(st.accL, ones, pf); MF(st.accL, ones, pg); }
  }
  SB();
#undef MF
}
#undef SB

// MODE 0: single-phase schedule (all waves: QK^T -> softmax -> PV per tile, one barrier per tile) with
//         the static-stage / precomputed-address / 32-bit-DMA-offset loop body ("v2" of attention.hip)
// MODE 1: two-phase ping-pong schedule described above (NW must be 8)
template <int FLAGS, int MODE, int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn_pp_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ Kg,
                                                         const bf16_t* __restrict__ Vt, bf16_t* __restrict__ O, int B,
                                                         int Bk, int64_t Lq, int64_t Lk, int64_t ldv, int H, int nqb,
                                                         float scale_log2e, int nseg, int64_t k_seg_stride,
                                                         int64_t vt_seg_stride) {
  constexpr bool PRIO = (FLAGS & 1) != 0;
  constexpr bool ROWSUM = (FLAGS & 2) != 0;     // row sums l on the matrix pipe (ones x P^T) instead of v_add
  constexpr bool PIPELINED = (FLAGS & 4) != 0;  // software-pipelined tile_fast for full tiles (MODE 0/2)
  constexpr bool HANDSCHED = (FLAGS & 64) != 0;  // tile_fast2 (hand-placed MFMA/softmax interleave)
  constexpr bool ABL_NOSM = (FLAGS & 8) != 0, ABL_NODMA = (FLAGS & 16) != 0, ABL_NOBAR = (FLAGS & 32) != 0;  // timing ablations
  constexpr int NST = (MODE == 2) ? 3 : 2;  // LDS ring depth
  __shared__ __attribute__((aligned(16))) char smem[2 * NST * IMG];  // [K stage 0..NST-1][V^T stage 0..NST-1]
  lds_cchar* lds = (lds_cchar*)smem;  // LDS-typed view: fragment reads become ds_read_b128 v, offset:imm
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
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

  constexpr int NSLOT = 1024 / (NW * 64);  // 16-B DMA pieces per lane per image
  const int64_t q0 = (int64_t)qb * (NW * 32) + wave * 32;
  int64_t qrow = q0 + l31;
  if (qrow > Lq - 1) qrow = Lq - 1;
  mfma_bf16x8 qf[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks)
    qf[ks] = *reinterpret_cast<const mfma_bf16x8*>(qbase + qrow * rs + ks * 16 + half * 8);

  // ---- DMA plan: 2 x 16-B slots per lane per image ----------------------------------------------
  const uint32_t rs2 = (uint32_t)(rs * 2);  // K row pitch in bytes
  uint32_t krow[NSLOT], kcol[NSLOT], vofs[NSLOT];
#pragma unroll
  for (int i = 0; i < NSLOT; ++i) {
    const int s = i * (NW * 64) + tid;
    {
      const int r = s >> 4, pch = s & 15;
      krow[i] = (uint32_t)((r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1));  // bits 2<->3 swapped
      kcol[i] = (uint32_t)((pch ^ (r & 15)) << 4);
    }
    {
      const int r = s >> 3, pch = s & 7;
      vofs[i] = (uint32_t)r * (uint32_t)(ldv * 2) + (uint32_t)((pch ^ ((r >> 1) & 7)) << 4);
    }
  }
  const int tps = (int)((Lk + KVBLK - 1) / KVBLK);
  const int ntile = tps * nseg;

  auto issueK = [&](int ST, int t) {
    const int seg = (nseg == 1) ? 0 : t / tps;
    const int64_t kv0 = (int64_t)(t - seg * tps) * KVBLK;
    const char* base = reinterpret_cast<const char*>(kbase + (int64_t)seg * k_seg_stride + kv0 * rs);
    int64_t lim64 = Lk - 1 - kv0;
    const uint32_t lim = (uint32_t)(lim64 > 63 ? 63 : lim64);
    char* dst = smem + ST * IMG;
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
      const uint32_t r = krow[i] < lim ? krow[i] : lim;  // clamp rows past the end of the segment
      glds16p(base + (r * rs2 + kcol[i]), dst + (i * (NW * 64) + wave * 64) * 16);
    }
  };
  auto issueV = [&](int ST, int t) {
    const int seg = (nseg == 1) ? 0 : t / tps;
    const int64_t kv0 = (int64_t)(t - seg * tps) * KVBLK;
    const char* base = reinterpret_cast<const char*>(vbase + (int64_t)seg * vt_seg_stride + kv0);
    char* dst = smem + NST * IMG + ST * IMG;
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) glds16p(base + vofs[i], dst + (i * (NW * 64) + wave * 64) * 16);
  };

  // MODE 0/2: K and V^T of a tile are issued together, walking running (uniform) source pointers: no
  // per-tile multiplies or divisions.
  int dma_tt = 0, dma_seg = 0;
  const char* dma_k = reinterpret_cast<const char*>(kbase);
  const char* dma_v = reinterpret_cast<const char*>(vbase);
  auto issue_next = [&](int ST) {
    uint32_t lim = 63;
    if (dma_tt == tps - 1) lim = (uint32_t)(Lk - 1 - (int64_t)dma_tt * KVBLK);  // ragged tail: clamp rows
    char* kd = smem + ST * IMG;
    char* vd = smem + NST * IMG + ST * IMG;
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
      const uint32_t r = krow[i] < lim ? krow[i] : lim;
      glds16p(dma_k + (r * rs2 + kcol[i]), kd + (i * (NW * 64) + wave * 64) * 16);
      glds16p(dma_v + vofs[i], vd + (i * (NW * 64) + wave * 64) * 16);
    }
    if (++dma_tt == tps) {
      dma_tt = 0;
      ++dma_seg;
      dma_k = reinterpret_cast<const char*>(kbase + (int64_t)dma_seg * k_seg_stride);
      dma_v = reinterpret_cast<const char*>(vbase + (int64_t)dma_seg * vt_seg_stride);
    } else {
      dma_k += (int64_t)KVBLK * rs2;
      dma_v += KVBLK * 2;
    }
  };
  int cur_tt = 0;  // tile index inside the current segment of the tile being consumed
  auto next_kv0 = [&]() {
    const int64_t kv0 = (int64_t)cur_tt * KVBLK;
    if (++cur_tt == tps) cur_tt = 0;
    return kv0;
  };

  // ---- LDS fragment addresses: per-lane VGPR + compile-time immediates -----------------------------
  int kaddr[8], vaddr[4];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) kaddr[ks] = l31 * 256 + (((ks * 2 + half) ^ (l31 & 15)) << 4);
#pragma unroll
  for (int c4 = 0; c4 < 4; ++c4) vaddr[c4] = l31 * 128 + (((c4 * 2 + half) ^ ((l31 >> 1) & 7)) << 4);

  PPState st;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) st.accO[dt][r] = 0.f;
#pragma unroll
  for (int T = 0; T < 2; ++T)
#pragma unroll
    for (int r = 0; r < 8; ++r) st.pk[T][r] = 0u;
#pragma unroll
  for (int r = 0; r < 16; ++r) st.accL[r] = 0.f;
  st.m_run = -INFINITY;
  st.l_run = 0.f;
  VFrags vf, vscratch;  // vscratch: the ping-pong mode reads (and discards) V^T early; see pv<..., false>

  auto kv0_of = [&](int t) { return (int64_t)((nseg == 1) ? t : t % tps) * KVBLK; };

  if (MODE == 1) {
    issueK(0, 0);
    issueV(0, 0);
  } else {
    issue_next(0);
  }

  // The two groups run separate loops (identical barrier counts; s_barrier only counts arrivals), so
  // each loop body is a straight-line [phase | barrier | phase | barrier] stream for the allocator.
#define PP_BAR1(t)                                                                 \
  if ((t) < ntile) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");     \
  else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                          \
  __builtin_amdgcn_s_barrier();                                                    \
  asm volatile("" ::: "memory");                                                   \
  if ((t) + 1 < ntile) issueK(((t) + 1) & 1, (t) + 1);
#define PP_BAR2(t)                                                                 \
  if ((t) + 1 < ntile) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); \
  else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                 \
  __builtin_amdgcn_s_barrier();                                                    \
  asm volatile("" ::: "memory");                                                   \
  if ((t) + 1 < ntile) issueV(((t) + 1) & 1, (t) + 1);

  // one tile from LDS stage STG: pipelined fast path for full tiles, sequential path for a segment's ragged tail
#define COMPUTE_TILE(STG)                                                                                   \
  {                                                                                                         \
    const int64_t kv0_ = next_kv0();                                                                        \
    if (HANDSCHED && kv0_ + KVBLK <= Lk) {                                                                  \
      tile_fast2<STG, NST, ROWSUM>(lds, kaddr, vaddr, qf, st, scale_log2e);                                 \
    } else if (PIPELINED && kv0_ + KVBLK <= Lk) {                                                           \
      tile_fast<STG, NST, ROWSUM>(lds, kaddr, vaddr, qf, st, scale_log2e);                                  \
    } else {                                                                                                \
      qk_softmax<STG, PRIO, NST, ROWSUM, ABL_NOSM>(lds, kaddr, vaddr, qf, st, vf, kv0_, Lk, half, scale_log2e); \
      pv<STG, NST, true, ROWSUM>(lds, vaddr, st, vf);                                                       \
    }                                                                                                       \
  }
  if (MODE == 0) {
    // single phase: tile t+1 (K and V^T) is in flight while tile t is consumed
    for (int t = 0; t < ntile; t += 2) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      if (!ABL_NOBAR) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (t + 1 < ntile && !(ABL_NODMA && t > 0)) issue_next(1);
      COMPUTE_TILE(0);
      if (t + 1 < ntile) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (!ABL_NOBAR) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + 2 < ntile && !ABL_NODMA) issue_next(0);
        COMPUTE_TILE(1);
      }
    }
  } else if (MODE == 2) {
    // single phase, 3-deep ring: tiles t+1 and t+2 are in flight while tile t is consumed; the wait before
    // the barrier is COUNTED (the younger tile's 2*NSLOT pieces per lane stay in flight across it)
    if (1 < ntile) issue_next(1);
#define V3_STEP(J)                                                                                     \
  if (t + (J) < ntile) {                                                                               \
    const int tt = t + (J);                                                                            \
    if (tt + 1 < ntile) {                                                                              \
      if (NSLOT == 2) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");                      \
      else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");                                 \
    } else {                                                                                           \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                      \
    }                                                                                                  \
    __builtin_amdgcn_s_barrier();                                                                      \
    asm volatile("" ::: "memory");                                                                     \
    if (tt + 2 < ntile) issue_next(((J) + 2) % 3);                                                     \
    COMPUTE_TILE(J)                                                                                    \
  }
    for (int t = 0; t < ntile; t += 3) {
      V3_STEP(0)
      V3_STEP(1)
      V3_STEP(2)
    }
#undef V3_STEP
  } else if (grp == 0) {
    // G0: P1 = QK^T(t)+softmax(t), P2 = PV(t)
    for (int t = 0; t <= ntile; t += 2) {
      PP_BAR1(t);
      if (t < ntile) qk_softmax<0, PRIO, NST, ROWSUM>(lds, kaddr, vaddr, qf, st, vscratch, kv0_of(t), Lk, half, scale_log2e);
      PP_BAR2(t);
      if (t < ntile) pv<0, NST, false, ROWSUM>(lds, vaddr, st, vf);
      if (t + 1 <= ntile) {
        PP_BAR1(t + 1);
        if (t + 1 < ntile) qk_softmax<1, PRIO, NST, ROWSUM>(lds, kaddr, vaddr, qf, st, vscratch, kv0_of(t + 1), Lk, half, scale_log2e);
        PP_BAR2(t + 1);
        if (t + 1 < ntile) pv<1, NST, false, ROWSUM>(lds, vaddr, st, vf);
      }
    }
  } else {
    // G1: P1 = PV(t-1), P2 = QK^T(t)+softmax(t)
    for (int t = 0; t <= ntile; t += 2) {
      PP_BAR1(t);
      if (t >= 1) pv<1, NST, false, ROWSUM>(lds, vaddr, st, vf);
      PP_BAR2(t);
      if (t < ntile) qk_softmax<0, PRIO, NST, ROWSUM>(lds, kaddr, vaddr, qf, st, vscratch, kv0_of(t), Lk, half, scale_log2e);
      if (t + 1 <= ntile) {
        PP_BAR1(t + 1);
        pv<0, NST, false, ROWSUM>(lds, vaddr, st, vf);
        PP_BAR2(t + 1);
        if (t + 1 < ntile) qk_softmax<1, PRIO, NST, ROWSUM>(lds, kaddr, vaddr, qf, st, vscratch, kv0_of(t + 1), Lk, half, scale_log2e);
      }
    }
  }
#undef PP_BAR1
#undef PP_BAR2

  // ---- epilogue ----------------------------------------------------------------------------------------
  // ROWSUM: accL[0] = sum over ALL kv of P[q][kv] (the MFMA contracts both halves' kv), q = lane&31;
  // otherwise the two halves hold partial sums
  const float inv = 1.0f / (ROWSUM ? st.accL[0] : st.l_run + __shfl_xor(st.l_run, 32, 64));
  __syncthreads();
  char* ob = smem + wave * (32 * 256);
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      uint2 w;
      w.x = cvt_pk_bf16_pp(st.accO[dt][g * 4 + 0] * inv, st.accO[dt][g * 4 + 1] * inv);
      w.y = cvt_pk_bf16_pp(st.accO[dt][g * 4 + 2] * inv, st.accO[dt][g * 4 + 3] * inv);
      const int ch = (dt * 4 + g) ^ (l31 & 15);
      *reinterpret_cast<uint2*>(ob + l31 * 256 + ch * 16 + half * 8) = w;
    }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = i * 4 + (lane >> 4), c = lane & 15;
    const int64_t qr = q0 + r;
    if (qr < Lq) {
      const uint4 val = *reinterpret_cast<const uint4*>(ob + r * 256 + ((c ^ (r & 15)) << 4));
      *reinterpret_cast<uint4*>(obase + qr * rs + c * 8) = val;
    }
  }
}

// called from attention.hip's dispatcher.  flags bit0: setprio around QK^T; mode 0/1; nw 4/8
int wan_attention_pp_launch(int flags, int mode, int nw, const bf16_t* q, const bf16_t* k, const bf16_t* vt, bf16_t* o,
                            int B, int Bk, int64_t Lq, int64_t Lk, int64_t ldv, int H, int nseg, int64_t k_seg_stride,
                            int64_t vt_seg_stride, float scale_log2e, hipStream_t stream) {
  WAN_REQUIRE(Lk * (int64_t)H * 256 < ((int64_t)1 << 32) && ldv * 256 < ((int64_t)1 << 32),
              "wan_attention: K/V^T extent exceeds the 32-bit DMA offsets of this kernel");
  WAN_REQUIRE((mode == 0 && (nw == 4 || nw == 8)) || ((mode == 1 || mode == 2) && nw == 8), "wan_attention: bad kernel variant");
  const int64_t nqb = (Lq + nw * 32 - 1) / (nw * 32);
  const int64_t total = nqb * H * B;
  WAN_REQUIRE(total < ((int64_t)1 << 31), "wan_attention: grid too large");
#define PP_LAUNCH(FL, MD, NWV)                                                                                      \
  hipLaunchKernelGGL((attn_pp_kernel<FL, MD, NWV>), dim3((unsigned)total), dim3(NWV * 64), 0, stream, q, k, vt, o, B, \
                     Bk, Lq, Lk, ldv, H, (int)nqb, scale_log2e, nseg, k_seg_stride, vt_seg_stride)
  // flags: bit0 setprio, bit1 row sums on the matrix pipe, bit2 software-pipelined tiles
  if (mode == 1) {
    PP_LAUNCH(2, 1, 8);
  } else if (mode == 2) {
    if (flags & 4) PP_LAUNCH(6, 2, 8); else PP_LAUNCH(2, 2, 8);
  } else if (flags & 64) {
    if (nw == 8) { if (flags & 2) PP_LAUNCH(66, 0, 8); else PP_LAUNCH(64, 0, 8); }
    else { if (flags & 2) PP_LAUNCH(66, 0, 4); else PP_LAUNCH(64, 0, 4); }
  } else if (nw == 8 && (flags & 56)) {  // timing ablations (tools/bench_attn.py only)
    switch (flags & 56) {
      case 8: PP_LAUNCH(8, 0, 8); break;
      case 16: PP_LAUNCH(16, 0, 8); break;
      case 32: PP_LAUNCH(32, 0, 8); break;
      case 24: PP_LAUNCH(24, 0, 8); break;
      case 48: PP_LAUNCH(48, 0, 8); break;
      default: PP_LAUNCH(56, 0, 8); break;
    }
  } else if (nw == 8) {
    switch (flags & 6) {
      case 6: PP_LAUNCH(6, 0, 8); break;
      case 4: PP_LAUNCH(4, 0, 8); break;
      case 2: PP_LAUNCH(2, 0, 8); break;
      default: PP_LAUNCH(0, 0, 8); break;
    }
  } else {
    switch (flags & 6) {
      case 6: PP_LAUNCH(6, 0, 4); break;
      case 4: PP_LAUNCH(4, 0, 4); break;
      case 2: PP_LAUNCH(2, 0, 4); break;
      default: PP_LAUNCH(0, 0, 4); break;
    }
  }
#undef PP_LAUNCH
  WAN_LAUNCH_CHECK();
  return 0;
}
