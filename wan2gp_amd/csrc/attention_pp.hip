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

struct PPState {
  f32x16 accO[4];
  uint32_t pk[2][8];
  float m_run, l_run;
};

// S^T = K Q^T for one 64-kv tile held in LDS stage ST, then the online-softmax update of `st`
template <int ST, bool PRIO>
__device__ __forceinline__ void qk_softmax(const char* smem, const int (&kaddr)[8], const mfma_bf16x8 (&qf)[8],
                                           PPState& st, int64_t kv0, int64_t Lk, int half, float scale_log2e) {
  f32x16 accS[2];
  if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int T = 0; T < 2; ++T) {
#pragma unroll
    for (int r = 0; r < 16; ++r) accS[T][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const mfma_bf16x8 kf = *reinterpret_cast<const mfma_bf16x8*>(smem + ST * 2 * IMG + T * 8192 + kaddr[ks]);
      accS[T] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], accS[T], 0, 0, 0);
    }
  }
  if (PRIO) __builtin_amdgcn_s_setprio(0);
  if (kv0 + KVBLK > Lk) {  // ragged tail of a segment: reg r <-> kv = kv0 + T*32 + (r&7) + 8*half + 16*(r>>3)
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kv0 + T * 32 + (r & 7) + 8 * half + 16 * (r >> 3) >= Lk) accS[T][r] = -INFINITY;
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
      psum += p0 + p1;
      st.pk[T][r >> 1] = cvt_pk_bf16_pp(p0, p1);
    }
  if (!__all(m_new == st.m_run)) {  // running max moved: rescale O and l once, before P(t) enters O
    const float alpha = __builtin_amdgcn_exp2f((st.m_run - m_new) * scale_log2e);
    st.l_run *= alpha;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) st.accO[dt][r] *= alpha;
  }
  st.l_run += psum;
  st.m_run = m_new;
}

// O^T += V^T P^T with V^T tile in LDS stage ST
template <int ST>
__device__ __forceinline__ void pv(const char* smem, const int (&vaddr)[4], PPState& st) {
#pragma unroll
  for (int T = 0; T < 2; ++T)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      uint4 praw;
      praw.x = st.pk[T][s * 4 + 0]; praw.y = st.pk[T][s * 4 + 1];
      praw.z = st.pk[T][s * 4 + 2]; praw.w = st.pk[T][s * 4 + 3];
      const mfma_bf16x8 pf = __builtin_bit_cast(mfma_bf16x8, praw);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const mfma_bf16x8 vf =
            *reinterpret_cast<const mfma_bf16x8*>(smem + ST * 2 * IMG + IMG + dt * 4096 + vaddr[T * 2 + s]);
        st.accO[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, st.accO[dt], 0, 0, 0);
      }
    }
}

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
  __shared__ __attribute__((aligned(16))) char smem[4 * IMG];  // [stage0: K | V^T][stage1: K | V^T]
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
    char* dst = smem + ST * 2 * IMG;
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
    char* dst = smem + ST * 2 * IMG + IMG;
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) glds16p(base + vofs[i], dst + (i * (NW * 64) + wave * 64) * 16);
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
  st.m_run = -INFINITY;
  st.l_run = 0.f;

  auto kv0_of = [&](int t) { return (int64_t)((nseg == 1) ? t : t % tps) * KVBLK; };

  issueK(0, 0);
  issueV(0, 0);

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

  if (MODE == 0) {
    // single phase: tile t+1 (K and V^T) is in flight while tile t is consumed
    for (int t = 0; t < ntile; t += 2) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (t + 1 < ntile) { issueK(1, t + 1); issueV(1, t + 1); }
      qk_softmax<0, PRIO>(smem, kaddr, qf, st, kv0_of(t), Lk, half, scale_log2e);
      pv<0>(smem, vaddr, st);
      if (t + 1 < ntile) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + 2 < ntile) { issueK(0, t + 2); issueV(0, t + 2); }
        qk_softmax<1, PRIO>(smem, kaddr, qf, st, kv0_of(t + 1), Lk, half, scale_log2e);
        pv<1>(smem, vaddr, st);
      }
    }
  } else if (grp == 0) {
    // G0: P1 = QK^T(t)+softmax(t), P2 = PV(t)
    for (int t = 0; t <= ntile; t += 2) {
      PP_BAR1(t);
      if (t < ntile) qk_softmax<0, PRIO>(smem, kaddr, qf, st, kv0_of(t), Lk, half, scale_log2e);
      PP_BAR2(t);
      if (t < ntile) pv<0>(smem, vaddr, st);
      if (t + 1 <= ntile) {
        PP_BAR1(t + 1);
        if (t + 1 < ntile) qk_softmax<1, PRIO>(smem, kaddr, qf, st, kv0_of(t + 1), Lk, half, scale_log2e);
        PP_BAR2(t + 1);
        if (t + 1 < ntile) pv<1>(smem, vaddr, st);
      }
    }
  } else {
    // G1: P1 = PV(t-1), P2 = QK^T(t)+softmax(t)
    for (int t = 0; t <= ntile; t += 2) {
      PP_BAR1(t);
      if (t >= 1) pv<1>(smem, vaddr, st);
      PP_BAR2(t);
      if (t < ntile) qk_softmax<0, PRIO>(smem, kaddr, qf, st, kv0_of(t), Lk, half, scale_log2e);
      if (t + 1 <= ntile) {
        PP_BAR1(t + 1);
        pv<0>(smem, vaddr, st);
        PP_BAR2(t + 1);
        if (t + 1 < ntile) qk_softmax<1, PRIO>(smem, kaddr, qf, st, kv0_of(t + 1), Lk, half, scale_log2e);
      }
    }
  }
#undef PP_BAR1
#undef PP_BAR2

  // ---- epilogue ----------------------------------------------------------------------------------------
  const float l_tot = st.l_run + __shfl_xor(st.l_run, 32, 64);
  const float inv = 1.0f / l_tot;
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
  WAN_REQUIRE((mode == 0 && (nw == 4 || nw == 8)) || (mode == 1 && nw == 8), "wan_attention: bad kernel variant");
  const int64_t nqb = (Lq + nw * 32 - 1) / (nw * 32);
  const int64_t total = nqb * H * B;
  WAN_REQUIRE(total < ((int64_t)1 << 31), "wan_attention: grid too large");
#define PP_LAUNCH(FL, MD, NWV)                                                                                      \
  hipLaunchKernelGGL((attn_pp_kernel<FL, MD, NWV>), dim3((unsigned)total), dim3(NWV * 64), 0, stream, q, k, vt, o, B, \
                     Bk, Lq, Lk, ldv, H, (int)nqb, scale_log2e, nseg, k_seg_stride, vt_seg_stride)
  const int pr = flags & 1;
  if (mode == 1) {
    if (pr) PP_LAUNCH(1, 1, 8); else PP_LAUNCH(0, 1, 8);
  } else if (nw == 8) {
    if (pr) PP_LAUNCH(1, 0, 8); else PP_LAUNCH(0, 0, 8);
  } else {
    if (pr) PP_LAUNCH(1, 0, 4); else PP_LAUNCH(0, 0, 4);
  }
#undef PP_LAUNCH
  WAN_LAUNCH_CHECK();
  return 0;
}
