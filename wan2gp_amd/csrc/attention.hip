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
// The kernel (attention_w64q.hip, attn_w64q_kernel): 4 waves = one per SIMD, 64 q rows per wave, 3-deep LDS-DMA ring, for
// every call.  Two tile loops: the bounded softmax (no running max; long KV only, taken when a pre-pass over K proves
// |s| <= 96 in log2 units for the whole workgroup) and the lazy-max tracking loop (any input; short KV -- cross-attention,
// Lk = 512 -- always: measured equal to the former 4 x 32-row short-KV kernel there, 775 vs 772 TFLOP/s).
// LDS images are XOR-swizzled on the DMA *source* address (K: 256-B rows, chunk ^= row&15; V^T: 128-B rows,
// chunk ^= (row>>1)&7) so every ds_read_b128 lane group hits 16 distinct 16-B slots.  Workgroup ids are remapped so that
// each XCD owns whole (batch, head) pairs: the blocks resident on an XCD stream the same K/V through its private L2.
#include <stdlib.h>
#include <string.h>

#include "common.h"

int wan_attention_w64q_launch(int flags, const bf16_t* q, const bf16_t* k, const bf16_t* vt, bf16_t* o, int B, int Bk,
                             int64_t Lq, int64_t Lk, int64_t ldv, int H, int nseg, int64_t k_seg_stride,
                             int64_t vt_seg_stride, float scale_log2e, float* kmax_scratch, hipStream_t stream);

int wan_attention_w64q_sp(int phase, const bf16_t* q, const bf16_t* k, const bf16_t* vt, bf16_t* o, int B, int64_t Lq, int64_t Lk,
                          int64_t ldv, int H, int nseg, int64_t k_seg_stride, int64_t vt_seg_stride, int own_seg,
                          float scale_log2e, float* kmax_scratch, float* raw, hipStream_t stream);

constexpr int KVBLK = 64;
constexpr float SCALE_LOG2E = 0.08838834764831845f * 1.4426950408889634f;  // 1/sqrt(128) * log2(e)

extern "C" int64_t wan_attention_scratch_words(int B, int Bk, int64_t Lq, int H) {
  // max |k_h|^2 per (batch, head) + one flag per 256-row workgroup + (sequence parallelism) the maxima of the previous partial launch
  return (int64_t)Bk * H + ((Lq + 255) / 256) * H * B + (int64_t)Bk * H;
}

// Library-owned scratch for the K pre-pass of callers that bring none (wan_attention / _seg / _prescaled): rings of 16 slots x
// 64 Ki words per (device, stream) (common.h wan_scratch_ring_slot).  wan_dit_forward passes a slice of its own workspace instead
// (wan_attention_bounded).
constexpr int64_t KMAX_SLOT = 65536;
constexpr int KMAX_NSLOT = 16;
static float* kmax_ring_slot(int64_t need, hipStream_t stream) {
  return reinterpret_cast<float*>(wan_scratch_ring_slot(/*tag=*/2, (size_t)KMAX_SLOT * sizeof(float), KMAX_NSLOT, (size_t)need * sizeof(float), stream));
}

enum { SCRATCH_NONE = 0, SCRATCH_RING = 1, SCRATCH_CALLER = 2 };
static int attention_dispatch(const wan_bf16* q, const wan_bf16* k, const wan_bf16* vt, wan_bf16* o, int B, int Bk,
                              int64_t Lq, int64_t Lk, int64_t ldv, int H, int nseg, int64_t k_seg_stride,
                              int64_t vt_seg_stride, bool q_prescaled, int scratch_kind, float* scratch, void* stream) {
  WAN_REQUIRE(q && k && vt && o, "wan_attention: null pointer");
  WAN_REQUIRE(nseg >= 1, "wan_attention: nseg must be >= 1");
  WAN_REQUIRE(B >= 1 && Bk >= 1 && B % Bk == 0, "wan_attention: Bk must divide B (q batch b attends K / V^T batch b mod Bk; B=%d Bk=%d)", B, Bk);
  WAN_REQUIRE(Lq >= 1 && Lk >= 1 && H >= 1, "wan_attention: empty problem (Lq=%lld Lk=%lld H=%d)", (long long)Lq,
              (long long)Lk, H);
  WAN_REQUIRE(ldv % KVBLK == 0 && ldv >= Lk, "wan_attention: ldv=%lld must be a multiple of 64 and >= Lk",
              (long long)ldv);
  WAN_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)vt | (uintptr_t)o) & 15) == 0,
              "wan_attention: pointers must be 16-byte aligned");
  // both kernels address a K / V^T segment with 32-bit DMA offsets: 419,430 kv rows per segment at 40 heads, 2.9x the
  // largest BASELINE configuration (720p x 161 frames = 147,600 tokens)
  WAN_REQUIRE(Lk * (int64_t)H * 256 < ((int64_t)1 << 32) && ldv * 256 < ((int64_t)1 << 32),
              "wan_attention: a K / V^T segment of %lld rows x %d heads exceeds the 32-bit DMA offsets of the kernels",
              (long long)Lk, H);
  // long KV: the bounded loop.  Short KV of at least 8 tiles in one segment (cross-attention: 512 text tokens) takes it too when the caller
  // brings a scratch -- as ONE persistent workgroup per CU (attention_w16n.hip PERSIST).  Below that (CLIP's 257 tokens, toy shapes): tracking loop.
  const bool long_kv = Lk * (int64_t)nseg > 2048 || (nseg == 1 && Lk > 448 && scratch_kind == SCRATCH_CALLER);
  {
    // the bounded loop sums UNROUNDED P into l while P enters the PV product rounded to bf16: negligible over thousands of
    // keys, a visible 2^-9 for a handful (Lk = 1: O = bf16(2^s) v / 2^s instead of v) -- short KV always takes the tracking
    // loop, whose dominant term is exactly 1
    float* km = nullptr;
    if (long_kv) km = scratch_kind == SCRATCH_CALLER ? scratch : (scratch_kind == SCRATCH_RING ? kmax_ring_slot(wan_attention_scratch_words(B, Bk, Lq, H), as_stream(stream)) : nullptr);
    return wan_attention_w64q_launch(q_prescaled ? 2 : 0, q, k, vt, o, B, Bk, Lq, Lk, ldv, H, nseg, k_seg_stride,
                                     vt_seg_stride, SCALE_LOG2E, km, as_stream(stream));
  }
}

extern "C" int wan_attention(const wan_bf16* q, const wan_bf16* k, const wan_bf16* vt, wan_bf16* o, int B, int Bk,
                             int64_t Lq, int64_t Lk, int64_t ldv, int H, void* stream) {
  return attention_dispatch(q, k, vt, o, B, Bk, Lq, Lk, ldv, H, 1, 0, 0, false, SCRATCH_RING, nullptr, stream);
}

extern "C" int wan_attention_seg(const wan_bf16* q, const wan_bf16* k, const wan_bf16* vt, wan_bf16* o, int B, int Bk,
                                 int64_t Lq, int64_t Lk, int64_t ldv, int H, int nseg, int64_t k_seg_stride,
                                 int64_t vt_seg_stride, void* stream) {
  return attention_dispatch(q, k, vt, o, B, Bk, Lq, Lk, ldv, H, nseg, k_seg_stride, vt_seg_stride, false, SCRATCH_RING, nullptr,
                            stream);
}

extern "C" int wan_attention_prescaled(const wan_bf16* q, const wan_bf16* k, const wan_bf16* vt, wan_bf16* o, int B,
                                       int Bk, int64_t Lq, int64_t Lk, int64_t ldv, int H, int nseg,
                                       int64_t k_seg_stride, int64_t vt_seg_stride, void* stream) {
  return attention_dispatch(q, k, vt, o, B, Bk, Lq, Lk, ldv, H, nseg, k_seg_stride, vt_seg_stride, true, SCRATCH_RING, nullptr,
                            stream);
}

extern "C" int wan_attention_bounded(const wan_bf16* q, const wan_bf16* k, const wan_bf16* vt, wan_bf16* o, int B, int Bk,
                                     int64_t Lq, int64_t Lk, int64_t ldv, int H, int nseg, int64_t k_seg_stride,
                                     int64_t vt_seg_stride, int q_prescaled, float* kmax_scratch, void* stream) {
  return attention_dispatch(q, k, vt, o, B, Bk, Lq, Lk, ldv, H, nseg, k_seg_stride, vt_seg_stride, q_prescaled != 0,
                            kmax_scratch ? SCRATCH_CALLER : SCRATCH_NONE, kmax_scratch, stream);
}

extern "C" int64_t wan_attention_raw_words(int B, int64_t Lq, int H) {
  return ((Lq + 255) / 256) * H * B * (int64_t)(4 * 2 * (64 * 64 + 128));  // per 256-row workgroup: 4 waves x 2 q halves x (accumulators + row-sum shares of two q tiles)
}

// Sequence-parallel self-attention, local segment first (q pre-scaled): see wan_attention_w64q_sp.
extern "C" int wan_attention_sp_local(const wan_bf16* q, const wan_bf16* k_local, const wan_bf16* vt_local, int B, int64_t Lq,
                                      int64_t Lk, int64_t ldv, int H, float* scratch, float* raw, void* stream) {
  WAN_REQUIRE(q && k_local && vt_local && scratch && raw, "wan_attention_sp_local: null pointer");
  WAN_REQUIRE(ldv % KVBLK == 0 && ldv >= Lk && Lq >= 1 && Lk >= 1, "wan_attention_sp_local: bad shape");
  return wan_attention_w64q_sp(0, q, k_local, vt_local, nullptr, B, Lq, Lk, ldv, H, 2, 0, 0, 0, SCALE_LOG2E, scratch, raw, as_stream(stream));
}
extern "C" int wan_attention_sp_remote(const wan_bf16* q, const wan_bf16* k_all, const wan_bf16* vt_all, wan_bf16* o, int B, int64_t Lq,
                                       int64_t Lk, int64_t ldv, int H, int nseg, int64_t k_seg_stride, int64_t vt_seg_stride,
                                       int own_seg, float* scratch, float* raw, void* stream) {
  WAN_REQUIRE(q && k_all && vt_all && o && scratch && raw, "wan_attention_sp_remote: null pointer");
  WAN_REQUIRE(ldv % KVBLK == 0 && ldv >= Lk && Lq >= 1 && Lk >= 1, "wan_attention_sp_remote: bad shape");
  return wan_attention_w64q_sp(1, q, k_all, vt_all, o, B, Lq, Lk, ldv, H, nseg, k_seg_stride, vt_seg_stride, own_seg, SCALE_LOG2E,
                               scratch, raw, as_stream(stream));
}

extern "C" float wan_attention_qscale(void) { return SCALE_LOG2E; }

// Measurement hook (bench.py `roofline.declined_workgroups`): after a wan_attention_bounded launch with a caller scratch, the
// scratch holds one flag per 256-row workgroup (1 = the workgroup's rows failed |q~| max|k| <= 96 and the tracking loop ran it).
// Adds (flagged, total) to acc[0..1] (device, 64-bit) on `stream`.
__global__ void attn_count_flags_kernel(const int* __restrict__ flags, int n, unsigned long long* __restrict__ acc) {
  unsigned long long c = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) c += flags[i] != 0;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  __shared__ unsigned long long part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(acc, part[0] + part[1] + part[2] + part[3]);
    atomicAdd(acc + 1, (unsigned long long)n);
  }
}
extern "C" int wan_attention_count_declined(const float* scratch, int B, int Bk, int64_t Lq, int H, uint64_t* acc, void* stream) {
  WAN_REQUIRE(scratch && acc && B >= 1 && Bk >= 1 && H >= 1 && Lq >= 1, "wan_attention_count_declined: bad args");
  const int64_t n = ((Lq + 255) / 256) * H * B;
  WAN_REQUIRE(n < ((int64_t)1 << 31), "wan_attention_count_declined: grid too large");
  hipLaunchKernelGGL(attn_count_flags_kernel, dim3(1), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const int*>(scratch + (size_t)Bk * H), (int)n, reinterpret_cast<unsigned long long*>(acc));
  WAN_LAUNCH_CHECK();
  return 0;
}
