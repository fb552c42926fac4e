// Exact (unmasked, non-causal) flash attention forward for gfx950, bf16 in/out, head_dim 128.
// Replaces pay_attention -> sdpa_wrapper (shared/attention.py:360-373, :208-225).
//
// Formulation (all on v_mfma_f32_32x32x16_bf16, fp32 accumulate):
//   S^T = K Q^T      first operand K (kv rows), second operand Q^T  -> a lane owns ONE q column
//                    (q = lane&31) and 16 kv rows per 32x32 tile: row max / row sum are
//                    in-register reductions + one cross-half exchange per 64-kv tile.
//   O^T = V^T P^T    first operand V^T (d rows, kv contiguous), second operand P^T.  Because a
//                    lane's S^T registers are exactly the P^T fragment it must supply (its own
//                    q, 8 kv per k-step), P never moves between lanes and never touches LDS.
// Two layout tricks make that work with 16-byte LDS reads only:
//   * V arrives TRANSPOSED in HBM ([H*128, ldv], produced by the V-projection GEMM epilogue,
//     gemm_bf16.hip WAN_EPI_TRANSPOSED), so the V^T fragment (8 consecutive kv of one d row) is
//     one ds_read_b128 -- no transpose reads, no ds_permute.
//   * K rows are staged into LDS with bits 2<->3 of the in-tile row index swapped, which turns
//     the MFMA C layout (row = (reg&3) + 8*(reg>>2) + 4*half) into "regs 0..7 = 8 consecutive
//     kv, regs 8..15 = the next-but-one 8", i.e. directly the B-operand k-order of the PV MFMA.
//
// Block = 4 waves x 32 q rows = 128 q rows of one (batch, head); KV tile = 64 rows; K and V^T
// tiles are DMA'd global->LDS (global_load_lds_dwordx4) into a 2-deep ring, the next tile in
// flight while the current one is consumed.  LDS images are XOR-swizzled on the DMA *source*
// address (K: 256-B rows, chunk ^= row&15;  V^T: 128-B rows, chunk ^= (row>>1)&7) so every
// ds_read_b128 lane group hits 16 distinct 16-B slots.
// Workgroup ids are remapped so that each XCD owns whole (batch, head) pairs: the 64 blocks
// resident on an XCD stream the same K/V through that XCD's private L2.
#include <stdlib.h>
#include <string.h>

#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;

typedef __attribute__((ext_vector_type(2))) __bf16 hw_bf16x2;
typedef __attribute__((ext_vector_type(2))) float hw_f32x2;
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {  // one v_cvt_pk_bf16_f32 (RNE)
  hw_f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hw_bf16x2));
}
#define KVBLK 64
#define K_STAGE (KVBLK * 256)        // 16 KiB: 64 rows x 128 d x 2 B
#define V_STAGE (128 * KVBLK * 2)    // 16 KiB: 128 d rows x 64 kv x 2 B

__device__ __forceinline__ void glds16a(const void* gsrc, void* ldst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)ldst, 16, 0, 0);
}

// FLAGS bit0: lean softmax (hardware v_cvt_pk_bf16_f32 packing, rescale O only when the running max
//             moved), bit1: s_setprio(1) around the MFMA clusters.  NW = waves per block (4 or 8).
template <int FLAGS, int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ Kg,
                                                          const bf16_t* __restrict__ Vt, bf16_t* __restrict__ O,
                                                          int B, int Bk, int64_t Lq, int64_t Lk, int64_t ldv, int H,
                                                          int nqb, float scale_log2e, int nseg, int64_t k_seg_stride,
                                                          int64_t vt_seg_stride) {
  __shared__ __attribute__((aligned(16))) char smem[2 * (K_STAGE + V_STAGE)];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int half = lane >> 5;
  const int l31 = lane & 31;

  const int total = nqb * H * B;
  const int v = xcd_remap(blockIdx.x, total);
  const int pair = v / nqb;
  const int qb = v - pair * nqb;
  const int b = pair / H, h = pair - b * H;
  const int bk = (Bk == 1) ? 0 : b;
  const int64_t rs = (int64_t)H * 128;  // token row stride of q/k/o (elements)

  const bf16_t* qbase = Q + ((int64_t)b * Lq) * rs + (int64_t)h * 128;
  // K/V may arrive as `nseg` equal segments of Lk rows each (one per sequence-parallel rank, laid
  // out [seg][Bk][...]); Lk is the per-segment length and every segment's tail tile is masked.
  const bf16_t* kbase = Kg + ((int64_t)bk * Lk) * rs + (int64_t)h * 128;
  const bf16_t* vbase = Vt + ((int64_t)bk * H * 128 + (int64_t)h * 128) * ldv;
  bf16_t* obase = O + ((int64_t)b * Lq) * rs + (int64_t)h * 128;

  // ---- Q fragments: B operand of S^T = K Q^T : lane holds Q[q=l31][ks*16 + half*8 .. +8] -------
  constexpr int QBLK = NW * 32;  // q rows per block
  constexpr bool LEAN = (FLAGS & 1) != 0;
  constexpr bool PRIO = (FLAGS & 2) != 0;
  constexpr int NLD = 1024 / (NW * 64);  // 16-B DMA slots per thread per 16 KiB image
  const int64_t q0 = (int64_t)qb * QBLK + wave * 32;
  int64_t qrow = q0 + l31;
  if (qrow > Lq - 1) qrow = Lq - 1;
  mfma_bf16x8 qf[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks)
    qf[ks] = *reinterpret_cast<const mfma_bf16x8*>(qbase + qrow * rs + ks * 16 + half * 8);

  // ---- staging addresses -----------------------------------------------------------------------
  // K image: LDS row r (0..63) <- kv_local = (r & ~12) | ((r&4)<<1) | ((r&8)>>1)  (swap bits 2,3)
  int kk_row[NLD], kk_col[NLD];  // kv_local, element column
  int vv_row[NLD], vv_col[NLD];  // d row, kv element column
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int s = i * (NW * 64) + tid;
    {
      const int r = s >> 4, pch = s & 15;
      const int lch = pch ^ (r & 15);
      kk_row[i] = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);
      kk_col[i] = lch * 8;
    }
    {
      const int r = s >> 3, pch = s & 7;
      const int lch = pch ^ ((r >> 1) & 7);
      vv_row[i] = r;
      vv_col[i] = lch * 8;
    }
  }
  const int tps = (int)((Lk + KVBLK - 1) / KVBLK);  // tiles per segment
  auto stage = [&](int st, int seg, int tt) {
    const int64_t kv0 = (int64_t)tt * KVBLK;
    const bf16_t* kseg = kbase + (int64_t)seg * k_seg_stride;
    const bf16_t* vseg = vbase + (int64_t)seg * vt_seg_stride;
    char* kb = smem + st * (K_STAGE + V_STAGE);
    char* vb = kb + K_STAGE;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int woff = (i * (NW * 64) + wave * 64) * 16;
      int64_t kr = kv0 + kk_row[i];
      if (kr > Lk - 1) kr = Lk - 1;
      glds16a(kseg + kr * rs + kk_col[i], kb + woff);
      glds16a(vseg + (int64_t)vv_row[i] * ldv + kv0 + vv_col[i], vb + woff);
    }
  };

  // ---- fragment read offsets ---------------------------------------------------------------------
  // K frag (A operand, tile T, step ks): row T*32+l31, logical chunk ks*2+half
  int koff[2];
#pragma unroll
  for (int T = 0; T < 2; ++T) {
    const int r = T * 32 + l31;
    koff[T] = r * 256 + (((half) ^ (r & 15)) << 4);  // ks folded in below by XOR with (ks*2)<<4
  }
  // V^T frag (A operand, d-tile dt, chunk c = T*4+s*2+half): row dt*32+l31
  int voff[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    const int r = dt * 32 + l31;
    voff[dt] = r * 128 + (((half) ^ ((r >> 1) & 7)) << 4);  // (T*4+s*2) folded in by XOR
  }

  f32x16 accO[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) accO[dt][r] = 0.f;
  float m_run = -INFINITY;  // running max of raw scores (same value in both halves)
  float l_run = 0.f;        // this lane's partial sum of exp

  const int ntile = tps * nseg;
  stage(0, 0, 0);
  int tt = 0;                    // tile index inside the current segment (no per-tile division)
  int pf_seg = 0, pf_tt = 0;     // (segment, tile) of the prefetched tile
  for (int t = 0; t < ntile; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 1 < ntile) {
      if (++pf_tt == tps) { pf_tt = 0; ++pf_seg; }
      stage((t + 1) & 1, pf_seg, pf_tt);
    }
    const char* kb = smem + (t & 1) * (K_STAGE + V_STAGE);
    const char* vb = kb + K_STAGE;

    // ---- S^T = K Q^T ---------------------------------------------------------------------------
    f32x16 accS[2];
    if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int T = 0; T < 2; ++T) {
#pragma unroll
      for (int r = 0; r < 16; ++r) accS[T][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const mfma_bf16x8 kf = *reinterpret_cast<const mfma_bf16x8*>(kb + (koff[T] ^ (ks << 5)));
        accS[T] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], accS[T], 0, 0, 0);
      }
    }
    if (PRIO) __builtin_amdgcn_s_setprio(0);
    // reg r of tile T, this half  <->  kv = t*64 + T*32 + (r&7) + 8*half + 16*(r>>3)
    if ((int64_t)(tt + 1) * KVBLK > Lk) {
      const int64_t kv0 = (int64_t)tt * KVBLK;
#pragma unroll
      for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t kv = kv0 + T * 32 + (r & 7) + 8 * half + 16 * (r >> 3);
          if (kv >= Lk) accS[T][r] = -INFINITY;
        }
    }
    // ---- online softmax -------------------------------------------------------------------------
    float mt = accS[0][0];
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
      for (int r = 0; r < 16; ++r) mt = fmaxf(mt, accS[T][r]);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m_run, mt);
    const float mb = m_new * scale_log2e;
    float psum = 0.f;
    uint32_t pk[2][8];
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float p0 = __builtin_amdgcn_exp2f(accS[T][r] * scale_log2e - mb);
        const float p1 = __builtin_amdgcn_exp2f(accS[T][r + 1] * scale_log2e - mb);
        psum += p0 + p1;
        pk[T][r >> 1] = LEAN ? cvt_pk_bf16(p0, p1) : pack2bf(p0, p1);
      }
    if (!LEAN || !__all(m_new == m_run)) {
      // the running max moved for at least one q row of this wave: rescale O and l (exactly once,
      // before this tile's P enters O); otherwise alpha == 1 for every lane and the pass is skipped
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2e);
      l_run *= alpha;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) accO[dt][r] *= alpha;
    }
    l_run += psum;
    m_run = m_new;
    if (PRIO) __builtin_amdgcn_s_setprio(1);

    // ---- O^T += V^T P^T ---------------------------------------------------------------------------
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        uint4 praw;
        praw.x = pk[T][s * 4 + 0]; praw.y = pk[T][s * 4 + 1];
        praw.z = pk[T][s * 4 + 2]; praw.w = pk[T][s * 4 + 3];
        const mfma_bf16x8 pf = __builtin_bit_cast(mfma_bf16x8, praw);
        const int cx = (T * 4 + s * 2) << 4;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const mfma_bf16x8 vf = *reinterpret_cast<const mfma_bf16x8*>(vb + (voff[dt] ^ cx));
          accO[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, accO[dt], 0, 0, 0);
        }
      }
    if (PRIO) __builtin_amdgcn_s_setprio(0);
    if (++tt == tps) tt = 0;
  }

  // ---- epilogue: normalise, stage O[q][d] through LDS, store whole 256-B rows -----------------------
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  __syncthreads();  // all waves done with the K/V ring
  // per-wave [32 q][128 d] bf16 image (8 KiB), 16-B chunks XOR-swizzled by the row (chunk ^= q&15)
  char* ob = smem + wave * (32 * 256);
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      uint2 w;
      w.x = cvt_pk_bf16(accO[dt][g * 4 + 0] * inv, accO[dt][g * 4 + 1] * inv);
      w.y = cvt_pk_bf16(accO[dt][g * 4 + 2] * inv, accO[dt][g * 4 + 3] * inv);
      const int ch = (dt * 4 + g) ^ (l31 & 15);  // d = dt*32 + g*8 + half*4 .. +4
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

extern "C" int wan_attention_seg(const wan_bf16* q, const wan_bf16* k, const wan_bf16* vt, wan_bf16* o, int B, int Bk,
                                 int64_t Lq, int64_t Lk, int64_t ldv, int H, int nseg, int64_t k_seg_stride,
                                 int64_t vt_seg_stride, void* stream);
int wan_attention_pp_launch(int flags, int mode, int nw, const bf16_t* q, const bf16_t* k, const bf16_t* vt, bf16_t* o, int B, int Bk,
                            int64_t Lq, int64_t Lk, int64_t ldv, int H, int nseg, int64_t k_seg_stride,
                            int64_t vt_seg_stride, float scale_log2e, hipStream_t stream);
int wan_attention_w64_launch(int flags, const bf16_t* q, const bf16_t* k, const bf16_t* vt, bf16_t* o, int B, int Bk,
                             int64_t Lq, int64_t Lk, int64_t ldv, int H, int nseg, int64_t k_seg_stride,
                             int64_t vt_seg_stride, float scale_log2e, hipStream_t stream);
int wan_attention_w64q_launch(int flags, const bf16_t* q, const bf16_t* k, const bf16_t* vt, bf16_t* o, int B, int Bk,
                             int64_t Lq, int64_t Lk, int64_t ldv, int H, int nseg, int64_t k_seg_stride,
                             int64_t vt_seg_stride, float scale_log2e, hipStream_t stream);

extern "C" int wan_attention(const wan_bf16* q, const wan_bf16* k, const wan_bf16* vt, wan_bf16* o, int B, int Bk,
                             int64_t Lq, int64_t Lk, int64_t ldv, int H, void* stream) {
  return wan_attention_seg(q, k, vt, o, B, Bk, Lq, Lk, ldv, H, 1, 0, 0, stream);
}

static int attention_dispatch(const wan_bf16* q, const wan_bf16* k, const wan_bf16* vt, wan_bf16* o, int B, int Bk,
                              int64_t Lq, int64_t Lk, int64_t ldv, int H, int nseg, int64_t k_seg_stride,
                              int64_t vt_seg_stride, bool q_prescaled, void* stream);

extern "C" int wan_attention_seg(const wan_bf16* q, const wan_bf16* k, const wan_bf16* vt, wan_bf16* o, int B, int Bk,
                                 int64_t Lq, int64_t Lk, int64_t ldv, int H, int nseg, int64_t k_seg_stride,
                                 int64_t vt_seg_stride, void* stream) {
  return attention_dispatch(q, k, vt, o, B, Bk, Lq, Lk, ldv, H, nseg, k_seg_stride, vt_seg_stride, false, stream);
}

extern "C" int wan_attention_prescaled(const wan_bf16* q, const wan_bf16* k, const wan_bf16* vt, wan_bf16* o, int B,
                                       int Bk, int64_t Lq, int64_t Lk, int64_t ldv, int H, int nseg,
                                       int64_t k_seg_stride, int64_t vt_seg_stride, void* stream) {
  return attention_dispatch(q, k, vt, o, B, Bk, Lq, Lk, ldv, H, nseg, k_seg_stride, vt_seg_stride, true, stream);
}

extern "C" float wan_attention_qscale(void) { return 0.08838834764831845f * 1.4426950408889634f; }

static int attention_dispatch(const wan_bf16* q, const wan_bf16* k, const wan_bf16* vt, wan_bf16* o, int B, int Bk,
                              int64_t Lq, int64_t Lk, int64_t ldv, int H, int nseg, int64_t k_seg_stride,
                              int64_t vt_seg_stride, bool q_prescaled, void* stream) {
  WAN_REQUIRE(q && k && vt && o, "wan_attention: null pointer");
  WAN_REQUIRE(nseg >= 1, "wan_attention: nseg must be >= 1");
  WAN_REQUIRE(B >= 1 && (Bk == B || Bk == 1), "wan_attention: Bk must be B or 1 (B=%d Bk=%d)", B, Bk);
  WAN_REQUIRE(Lq >= 1 && Lk >= 1 && H >= 1, "wan_attention: empty problem (Lq=%lld Lk=%lld H=%d)", (long long)Lq,
              (long long)Lk, H);
  WAN_REQUIRE(ldv % KVBLK == 0 && ldv >= Lk, "wan_attention: ldv=%lld must be a multiple of 64 and >= Lk",
              (long long)ldv);
  WAN_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)vt | (uintptr_t)o) & 15) == 0,
              "wan_attention: pointers must be 16-byte aligned");
  const float scale_log2e = 0.08838834764831845f * 1.4426950408889634f;  // 1/sqrt(128) * log2(e)
  // kernel variant: default = lean softmax, 4 waves.  WAN_ATTN_VARIANT (tuning/A-B only):
  //   "base" = first kernel, "lean", "lean_prio", "lean8" (8 waves / 256 q rows), "lean8_prio",
  //   "pp"/"pp_prio" = the two-phase ping-pong schedule of attention_pp.hip (correct, slower: see DESIGN.md)
  // measured on MI355X (profiles/r01_attn_variants.md): 8 waves sharing each K/V tile win for long
  // KV (self-attention), 4-wave blocks win for short KV (cross-attention, Lk = 512)
  // (v2 = static-stage loop body of attention_pp.hip MODE 0)
  int variant = (Lk * (int64_t)nseg > 2048) ? 20 : 7;  // w64 (4 waves x 64 q rows, exact softmax scaling) : v2_4
  if (Lk * (int64_t)H * 256 >= ((int64_t)1 << 32) || ldv * 256 >= ((int64_t)1 << 32)) variant = 3;  // 64-bit addressing kernel
  {
    const char* ev = getenv("WAN_ATTN_VARIANT");
    if (ev) {
      if (!strcmp(ev, "base")) variant = 0;
      else if (!strcmp(ev, "lean")) variant = 1;
      else if (!strcmp(ev, "lean_prio")) variant = 2;
      else if (!strcmp(ev, "lean8")) variant = 3;
      else if (!strcmp(ev, "lean8_prio")) variant = 4;
      else if (!strcmp(ev, "pp")) variant = 5;
      else if (!strcmp(ev, "pp_prio")) variant = 6;
      else if (!strcmp(ev, "v2_4")) variant = 7;
      else if (!strcmp(ev, "v2_8")) variant = 8;
      else if (!strcmp(ev, "v3_8")) variant = 9;
      else if (!strcmp(ev, "v2r_8")) variant = 10;   // v2 + row sums on the matrix pipe
      else if (!strcmp(ev, "v4_8")) variant = 11;    // software-pipelined tiles, VALU row sums
      else if (!strcmp(ev, "v4r_8")) variant = 12;   // software-pipelined tiles, MFMA row sums
      else if (!strcmp(ev, "v4_4")) variant = 13;
      else if (!strcmp(ev, "v4r_4")) variant = 14;
      else if (!strcmp(ev, "v4_8s3")) variant = 15;  // pipelined + 3-stage ring
      else if (!strcmp(ev, "v5_8")) variant = 16;    // hand-placed interleave, VALU row sums
      else if (!strcmp(ev, "v5r_8")) variant = 17;   // hand-placed interleave, MFMA row sums
      else if (!strcmp(ev, "v5_4")) variant = 18;
      else if (!strcmp(ev, "w64")) variant = 20;     // attention_w64.hip: 4 waves x 64 q rows, VALU row sums
      else if (!strcmp(ev, "w64r")) variant = 21;    // ... MFMA row sums
      else if (!strcmp(ev, "w64t")) variant = 24;    // ... w64 + s_memtime stamps (tools/bench_attn.py --stamps)
      else if (!strcmp(ev, "w64q")) variant = 28;    // attention_w64q.hip: issue-balanced 4 x 64 kernel (lazy max in the MFMA C operand)
      else if (!strcmp(ev, "w64qt")) variant = 29;   // ... + s_memtime stamps
      else if (!strcmp(ev, "w64f")) variant = 32;    // ... flat one-exp-per-gap schedule
      else if (!strcmp(ev, "w64ft")) variant = 33;
      else if (!strncmp(ev, "abl", 3)) variant = 100 + atoi(ev + 3);  // timing ablations: abl8 / abl16 / abl32 / abl24 / abl48 / abl56
    }
  }
  if (q_prescaled) {  // only the w64q kernel takes a pre-scaled q; flat schedule unless WAN_ATTN_VARIANT=w64q / w64qt
    WAN_REQUIRE(Lk * (int64_t)H * 256 < ((int64_t)1 << 32) && ldv * 256 < ((int64_t)1 << 32),
                "wan_attention_prescaled: K/V^T extent exceeds the 32-bit DMA offsets of the w64q kernel");
    const int fl = 2 | ((variant == 28 || variant == 29) ? 0 : 4) | ((variant == 29 || variant == 33) ? 1 : 0);
    return wan_attention_w64q_launch(fl, q, k, vt, o, B, Bk, Lq, Lk, ldv, H, nseg, k_seg_stride, vt_seg_stride, scale_log2e,
                                     as_stream(stream));
  }
  if (variant >= 32 && variant <= 33)
    return wan_attention_w64q_launch((variant - 32) | 4, q, k, vt, o, B, Bk, Lq, Lk, ldv, H, nseg, k_seg_stride, vt_seg_stride,
                                     scale_log2e, as_stream(stream));
  if (variant >= 28 && variant <= 29)
    return wan_attention_w64q_launch(variant - 28, q, k, vt, o, B, Bk, Lq, Lk, ldv, H, nseg, k_seg_stride, vt_seg_stride,
                                     scale_log2e, as_stream(stream));
  if (variant >= 20 && variant <= 24)
    return wan_attention_w64_launch(variant - 20, q, k, vt, o, B, Bk, Lq, Lk, ldv, H, nseg, k_seg_stride, vt_seg_stride,
                                    scale_log2e, as_stream(stream));
  if (variant >= 5) {
    int fl = 0, md = 0, nwv = 8;
    switch (variant) {
      case 5: case 6: md = 1; fl = 2; break;
      case 7: nwv = 4; break;
      case 8: break;
      case 9: md = 2; fl = 2; break;
      case 10: fl = 2; break;
      case 11: fl = 4; break;
      case 12: fl = 6; break;
      case 13: fl = 4; nwv = 4; break;
      case 14: fl = 6; nwv = 4; break;
      case 15: fl = 4 | 2; md = 2; break;
      case 16: fl = 64; break;
      case 17: fl = 66; break;
      case 18: fl = 64; nwv = 4; break;
      default: if (variant >= 100) fl = variant - 100; break;
    }
    return wan_attention_pp_launch(fl, md, nwv, q, k, vt, o, B, Bk, Lq, Lk, ldv, H, nseg, k_seg_stride, vt_seg_stride,
                                   scale_log2e, as_stream(stream));
  }
  const int nw = (variant >= 3) ? 8 : 4;
  const int64_t nqb = (Lq + nw * 32 - 1) / (nw * 32);
  const int64_t total = nqb * H * B;
  WAN_REQUIRE(total < ((int64_t)1 << 31), "wan_attention: grid too large");
#define LAUNCH_ATTN(FL, NWV)                                                                                        \
  hipLaunchKernelGGL((attn_fwd_kernel<FL, NWV>), dim3((unsigned)total), dim3(NWV * 64), 0, as_stream(stream), q, k, \
                     vt, o, B, Bk, Lq, Lk, ldv, H, (int)nqb, scale_log2e, nseg, k_seg_stride, vt_seg_stride)
  switch (variant) {
    case 0: LAUNCH_ATTN(0, 4); break;
    case 2: LAUNCH_ATTN(3, 4); break;
    case 3: LAUNCH_ATTN(1, 8); break;
    case 4: LAUNCH_ATTN(3, 8); break;
    default: LAUNCH_ATTN(1, 4); break;
  }
#undef LAUNCH_ATTN
  WAN_LAUNCH_CHECK();
  return 0;
}
