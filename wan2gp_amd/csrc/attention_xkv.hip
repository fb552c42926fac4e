// Text cross-attention for gfx950 with K and V^T STATIONARY in registers ("xkv", round 6).
//   WanT2VCrossAttention / text_cross_attention (models/wan/modules/model.py:245-307, :410-445) -> pay_attention (shared/attention.py:360-373)
//   at Lk = 512 text tokens, head_dim 128, q pre-scaled by scale * log2(e) (what csrc/dit.hip passes).
//
// Why another kernel.  attention_w16n.hip's persistent walk (round 4) streams the head's 8 K / V^T tiles through its LDS ring for EVERY
// 256-row q block: 16.4k matrix cycles per block inside ~36k -- 43.5 % matrix-pipe busy at 1.89 GHz, 0.34 of peak -- because one workgroup per
// CU has nobody to run while it judges rows, stores O and restarts its ring (docs/history section 3.1).  With 512 keys the
// roles can be swapped: a head's K and V^T are 256 KB = exactly the 4 x 64 lanes x 256 accumulator-file registers of a workgroup.  Here
//   * wave w keeps K of keys 128 w .. 128 w + 127 (8 x 4 A-fragments: S^T = K Q^T over its keys) and V^T of channels d = 32 w .. 32 w + 31 over
//     ALL 512 keys (2 x 16 A-fragments: O^T = V^T P^T for its channels), loaded ONCE per (batch, head) run of the workgroup's walk and pinned
//     to the accumulator file;
//   * Q rows stream through a 16-deep LDS-DMA ring in tiles of 16 rows (4 KB; one 1-KB piece per wave per tile, 15 tiles ahead);
//   * per tile a wave issues 32 MFMAs of S^T, 32 exp2, packs its four P^T fragments (bf16, 16 q x 128 keys = 4 KB) into LDS, and -- behind
//     one barrier -- multiplies ALL sixteen P^T fragments of the tile (its own and the other waves': plain lane-linear ds_read_b128) with its
//     V^T: 32 MFMAs that end in the finished O[16 rows][32 channels] of the wave, normalised by the row sums (exchanged the same way) and
//     stored.  What crosses LDS is P in bf16 -- half the bytes of the first form of this kernel (run 21), which exchanged fp32 partial O^T
//     and was bound by the LDS store path -- and nothing is added outside the matrix pipe;
//   * everything is ONE software-pipelined stream: S^T of tile t, three quarters of O^T of tile t - 1, the last quarter and the store of tile
//     t - 2 in one iteration of 64 MFMA slots -- a q block has no prologue, no verdict stage and no epilogue of its own; the only serial work
//     left is the K / V^T load of a run (once or twice per workgroup) and two pipeline-fill iterations per run.
// The bounded softmax has no running maximum (P = 2^s, as attention_w16n.hip's plain loop).  A row is sound when its row sum lies in
// [2^-80, 2^100] (no score overflowed, the sum did not underflow; attention_w16n.hip's verdict with m = 0); a tile with an unsound row flags its
// 256-row block 1 in wg_flags and the tracking launch that follows every bounded launch redoes that block -- which re-reads its Q rows, so
// this kernel serves OUT-OF-PLACE calls only (o != q; the launcher sends in-place calls to the persistent walk).
//
// Fragment layout (v_mfma_f32_16x16x32_bf16; lane: n = lane & 15, g = lane >> 4): attention_w16n.hip's.
//   S^T tile kt (16 key slots x 16 q): A = K fragment: lane (n, g) holds key slot 16 kt + n, d = 32 ks + 8 g .. + 7; B = Q fragment: q row n, the
//   same d.  Register i of lane (n, g) = score of q row n against slot 16 kt + 4 g + i.  Slot (kt, m) holds key 32 (kt >> 1) + 8 (m >> 2) +
//   4 (kt & 1) + (m & 3) of the wave's 128, so that registers 0..3 of tiles 2 p, 2 p + 1 are the 8 CONSECUTIVE keys 32 p + 8 g .. + 7 of q row n:
//   the lane's 16 bytes of the P^T B-fragment of global k-step C = 4 wave + p -- LDS slot [C][lane], read back by every wave at [C][lane].
//   O^T tile j of wave w: register i of lane (n, g) = O[q row n][d = 32 w + 16 j + 4 g + i]; A = V^T fragment (j, C): channel 32 w + 16 j + n,
//   keys 32 C + 8 g .. + 7, straight from memory.
//
// One iteration I = 64 slots of one MFMA + its fillers, four blocks of 16.  MFMAs of block p: even slots: S^T of tile I, key tiles 2 p, 2 p + 1
// (k-step major); odd slots: O^T, four k-steps x two channel tiles -- blocks 2, 3: tile I - 1, k-steps 0-3, 4-7; blocks 0, 1: tile I - 2,
// k-steps 8-11, 12-15.  Two MFMAs on one accumulator are four slots apart.  Fillers -- the transcendental in EVERY even slot (MFMA + v_exp_f32
// fill a slot's 16 matrix cycles by themselves, attention_w16n.hip), the rest in odd slots:
//   exp2 of score r = 0..7 of pair p in slot 16 p + 16 + 2 r (>= 3 MFMAs behind the tile's last MFMA: asm MFMAs are not padded by hipcc), in
//   place; its row-sum add THREE slots behind (one behind, hipcc pads the transcendental's result with an s_nop: an issue slot -- and this
//   stream is bound by its instruction count: 4 cycles per instruction of one wave per SIMD, run 30's counters); pair 3 ends in slots 0 .. 17
//   of the next iteration;
//   the pack of word w of pair p in slot 16 p + 21 + 4 w, its 16 bytes -> LDS in slot 16 p + 35 (pairs 2, 3: slots 3, 19 of the next iteration);
//   slot 21: the row-sum shares of tile I - 1 -> LDS; slot 23: the Q piece of tile I + 15; slot 27: vmcnt / lgkmcnt + the iteration's barrier;
//   P^T fragment reads behind the last use of their register (1, 5 .. 17 | 29 .. 35 | 37 .. 49 | 53 .. 61);
//   slots 32, 34: tile I - 2 normalised and packed in front of the next tile's first O^T MFMAs (33, 35); slot 36: one 16-byte store per lane;
//   slots 39 .. 59: the four waves' row-sum shares of tile I - 1, their sum over waves and lane groups, 1 / l and the verdict;
//   slots 51 .. 63: the next tile's Q fragments.
#include <stdlib.h>

#include <utility>

#include "attn_w64_shared.h"

namespace {

constexpr int XR = 16;                          // Q ring depth (tiles)
constexpr int XQT = 4096;                       // bytes per Q tile (16 rows x 256 B)
constexpr int XQ_BYTES = XR * XQT;              // 64 KB
constexpr int XP_P = 16 * 1024;                 // P^T of one tile: [k-step 0..15][lane] x 16 B
constexpr int XP_BUF = XP_P + 4 * 256;          // + row-sum shares [wave][lane] x 4 B
constexpr float X_MIN_ROWSUM = 8.271806125530277e-25f;   // 2^-80
constexpr float X_MAX_ROWSUM = 1.2676506002282294e30f;   // 2^100

typedef uint32_t xkv_u4 __attribute__((ext_vector_type(4)));
typedef unsigned int xkv_st4 __attribute__((__vector_size__(16)));

__device__ __forceinline__ void xs0(f32x4& d, const mfma_bf16x8& k, const mfma_bf16x8& q) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(d) : "a"(k), "v"(q));
}
__device__ __forceinline__ void xs1(f32x4& d, const mfma_bf16x8& k, const mfma_bf16x8& q) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d) : "a"(k), "v"(q));
}

// -DXKV_ABL=bits (make xabl; diagnostics, outputs are garbage): 1 = every MFMA without its accumulate dependency, 2 = no exp2 / row sums / packs,
// 4 = no P^T / row-sum exchange through LDS, 8 = no barrier and no counted waits, 16 = no stores
#ifndef XKV_ABL
#define XKV_ABL 0
#endif
struct XState {
  mfma_bf16x8 qf[4];   // Q fragments of the tile whose S^T runs in this iteration
  f32x4 s[8];          // S^T tiles, then P in place
  xkv_u4 pk;           // the pair being packed
  mfma_bf16x8 pf[4];   // P^T fragments of the O^T group in flight
  f32x4 o[2];          // O^T of this wave's 32 channels
  float lacc, lh;      // this lane's share of the row sums: being summed / complete (tile I - 1)
  float inv;           // 1 / l of tile I - 2
};

// MFMA of slot i.  Block p = i >> 4: S^T of pair p in its even slots, O^T k-steps in its odd slots -- two MFMAs on the SAME accumulator are four
// slots apart (with fillers between them a dependent MFMA two slots behind its producer waits for the write-back: measured 31 cycles per
// slot against 20 in this form, run 26 / 27).
__device__ __forceinline__ void xkv_mfma(XState& x, const mfma_bf16x8 (&kf)[8][4], const mfma_bf16x8 (&vf)[2][16], int i) {
  const int p = i >> 4, m = (i & 15) >> 1;
  if ((i & 1) == 0) {
    const int kt = 2 * p + (m & 1), ks = m >> 1;
    if (ks == 0 || (XKV_ABL & 1)) xs0(x.s[kt], kf[kt][ks], x.qf[ks]);
    else xs1(x.s[kt], kf[kt][ks], x.qf[ks]);
  } else {
    const int j = m >> 1, d = m & 1;
    const int C = ((p + 2) & 3) * 4 + j;
    if (C == 0 || (XKV_ABL & 1)) xs0(x.o[d], vf[d][C], x.pf[j]);
    else xs1(x.o[d], vf[d][C], x.pf[j]);
  }
}
// The exp2 stream of a tile: score r = 0..7 of pair p in the EVEN slot 16 p + 16 + 2 r (pair 3: 0 .. 14 of the next iteration), its row-sum add
// in the odd slot behind it, in place.
__device__ __forceinline__ constexpr int xkv_exp_pair(int i) { return i >= 16 ? (i - 16) >> 4 : 3; }
__device__ __forceinline__ constexpr int xkv_exp_reg(int i) { return (i & 15) >> 1; }
__device__ __forceinline__ void xkv_exp(XState& x, int i) {   // i: an even slot
  if (XKV_ABL & 2) return;
  const int p = xkv_exp_pair(i), r = xkv_exp_reg(i);
  float v = __builtin_amdgcn_exp2f(x.s[2 * p + (r >> 2)][r & 3]);
  asm volatile("" : "+v"(v));
  x.s[2 * p + (r >> 2)][r & 3] = v;
}
__device__ __forceinline__ void xkv_sum(XState& x, int i) {   // i: the odd slot THREE behind an exp2 (one behind, hipcc pads the transcendental's
  if (XKV_ABL & 2) return;                                    // result with an s_nop -- an issue slot; this stream is bound by its instruction count)
  const int e = (i + 61) & 63;
  const int p = xkv_exp_pair(e), r = xkv_exp_reg(e);
  x.lacc += x.s[2 * p + (r >> 2)][r & 3];
  asm volatile("" : "+v"(x.lacc));
}
// pack word w of pair p: odd slot 16 p + 21 + 4 w, three behind its second exp2 (pair 3: 5, 9, 13, 17 of the next iteration)
__device__ __forceinline__ void xkv_pack(XState& x, int i) {
  if (XKV_ABL & 2) return;
  const int k = i >= 21 ? i - 21 : i + 43;
  const int p = k >> 4, w = (k & 15) >> 2;
  if ((k & 3) != 0) return;
  const f32x4& t = x.s[2 * p + (w >> 1)];
  x.pk[w] = cvt_pk(t[2 * (w & 1)], t[2 * (w & 1) + 1]);
  asm volatile("" : "+v"(x.pk));
}

#ifdef XKV_STAMPS
__device__ uint64_t xkv_stamps[4][80];   // tuning build (make xstamp): s_memtime per slot of iteration 200 of workgroup 0, per wave
#define XST(I) do { if (c.rec && (((I) & 7) == 0 || (I) >= 64)) c.st[I] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define XST(I) do { } while (0)
#endif
struct XQ {            // the Q stream of a run (wave-uniform)
  const char* base;    // first row of the next tile to fetch
  int left, next;      // rows from that tile's first to the run's last; its index
  uint32_t voff, rs2, lds;
};
__device__ __forceinline__ void xkv_q_issue(XQ& q) {
  const uint32_t len = q.left > 0 ? (uint32_t)(q.left - 1) * q.rs2 + 256u : 0u;   // rows past the run read as zeros
  dma_issue(q.voff, q.base, len, q.lds + (uint32_t)((q.next & (XR - 1)) * XQT));
  q.base += 16 * (int64_t)q.rs2;
  q.left -= 16;
  q.next += 1;
}
struct XLane {         // loop-invariant per-lane addresses
  char* smem;
  lds_cchar* lds;
  uint32_t q[4], pw, pr, lw, lr, st;
};
struct XIter {         // one iteration's values
  uint32_t pb1, pb2, qoff;   // LDS offsets: the P buffer of tile I - 1 (= of tile I + 1), of tile I - 2 (= of tile I); the next tile's Q ring slot
  __amdgpu_buffer_rsrc_t odesc;
  float lsrc[4], lt, inv;
  uint32_t ow[4];
  int bad;
#ifdef XKV_STAMPS
  bool rec;
  uint64_t st[66];
#endif
};
// slot I of an iteration: its MFMA, then its fillers
template <int I>
__device__ __forceinline__ void xkv_slot(XState& x, const mfma_bf16x8 (&kf)[8][4], const mfma_bf16x8 (&vf)[2][16], XIter& c, const XLane& L, XQ& q) {
  constexpr int i = I;
  XST(i);
  xkv_mfma(x, kf, vf, i);
  SB();
  if ((i & 1) == 0) {
    xkv_exp(x, i);
    // tile I - 2's finished O^T tile d (last MFMA: slot 29 + 2 d) normalised, packed, stored -- in front of the next tile's first MFMA on that
    // accumulator (slot 33 + 2 d): a value kept across it lives in other registers, and hipcc then returns the loop-carried accumulator to
    // its own with v_movs straight in front of an asm MFMA that reads it -- two wait states short, silently (run 27: registers 2, 3 stale)
    if (i == 32 || i == 34) {
      constexpr int d = ((i - 32) >> 1) & 1;
      const f32x4 t = x.o[d];
      c.ow[2 * d] = cvt_pk(t[0] * x.inv, t[1] * x.inv);
      c.ow[2 * d + 1] = cvt_pk(t[2] * x.inv, t[3] * x.inv);
    }
    // ... as ONE 16-byte store per lane: lane group g holds channels 4 g .. + 3 of tile 0 (ow 0, 1) and of tile 1 (ow 2, 3); v_permlane16_swap
    // (vdst = tile 0's word, vsrc = tile 1's: vdst's odd rows <-> vsrc's even rows) leaves an even group with tile 0's channels 4 g .. 4 g + 7
    // and the odd group behind it with tile 1's 4 (g - 1) .. + 7 -- 8 consecutive channels.  (Two 8-byte stores per lane cost the stream
    // 0.2 ms of a 1.8-ms call: run 28's ablation -- a wave's store instruction is priced by the rows it touches, not by its bytes.)
    if (i == 36) {
      const auto r0 = __builtin_amdgcn_permlane16_swap(c.ow[0], c.ow[2], false, false);
      const auto r1 = __builtin_amdgcn_permlane16_swap(c.ow[1], c.ow[3], false, false);
      xkv_u4 w = {r0[0], r1[0], r0[1], r1[1]};
      if (!(XKV_ABL & 16)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(xkv_st4, w), c.odesc, (int)L.st, 0, 0);
      else asm volatile("" :: "v"(w));
    }
  } else {
    if (i == 27 && !(XKV_ABL & 8)) {   // the iteration's barrier, in front of the slot's other fillers
      asm volatile("s_waitcnt vmcnt(14) lgkmcnt(0)" ::: "memory");   // this wave's piece of tile I + 1 landed; its P^T and row sums of tile I - 1 are written
      XST(64);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      XST(65);
    }
    xkv_sum(x, i);
    if (i == 17) { x.lh = (XKV_ABL & 2) ? 1.f : x.lacc; x.lacc = 0.f; asm volatile("" : "+v"(x.lh), "+v"(x.lacc)); }   // (behind the add of the tail's last exp2, slot 14; slot 19 adds the new tile's first)
    xkv_pack(x, i);
    // the packed pair -> LDS slot [4 wave + pair][lane]: pairs 0, 1 of tile I (its buffer = tile I - 2's), pairs 2, 3 of tile I - 1
    if (!(XKV_ABL & 4)) if (i == 35) *reinterpret_cast<xkv_u4*>(L.smem + c.pb2 + L.pw + 0 * 1024) = x.pk;
    if (!(XKV_ABL & 4)) if (i == 51) *reinterpret_cast<xkv_u4*>(L.smem + c.pb2 + L.pw + 1 * 1024) = x.pk;
    if (!(XKV_ABL & 4)) if (i == 3) *reinterpret_cast<xkv_u4*>(L.smem + c.pb1 + L.pw + 2 * 1024) = x.pk;
    if (!(XKV_ABL & 4)) if (i == 19) *reinterpret_cast<xkv_u4*>(L.smem + c.pb1 + L.pw + 3 * 1024) = x.pk;
    if (!(XKV_ABL & 4)) if (i == 21) *reinterpret_cast<float*>(L.smem + c.pb1 + L.lw) = x.lh;
    if (i == 23) xkv_q_issue(q);
    // P^T fragment reads: register j behind its last use (slot 16 p + 4 j + 3 of the block in flight), 6 to 12 slots ahead of its next
    if (!(XKV_ABL & 4) && (i == 5 || i == 9 || i == 13 || i == 17)) { constexpr int j = ((i - 5) >> 2) & 3; x.pf[j] = *reinterpret_cast<const mfma_bf16x8*>(L.smem + c.pb2 + L.pr + (12 + j) * 1024); }    // tile I - 2, k-steps 12..15
    if (!(XKV_ABL & 4) && (i == 29 || i == 31 || i == 33 || i == 35)) { constexpr int j = ((i - 29) >> 1) & 3; x.pf[j] = *reinterpret_cast<const mfma_bf16x8*>(L.smem + c.pb1 + L.pr + (0 + j) * 1024); }  // tile I - 1, k-steps 0..3 (behind the barrier)
    if (!(XKV_ABL & 4) && (i == 37 || i == 41 || i == 45 || i == 49)) { constexpr int j = ((i - 37) >> 2) & 3; x.pf[j] = *reinterpret_cast<const mfma_bf16x8*>(L.smem + c.pb1 + L.pr + (4 + j) * 1024); }  // k-steps 4..7
    if (!(XKV_ABL & 4) && (i == 53 || i == 57 || i == 61)) { constexpr int j = ((i - 53) >> 2) & 3; x.pf[j] = *reinterpret_cast<const mfma_bf16x8*>(L.smem + c.pb1 + L.pr + (8 + j) * 1024); }             // k-steps 8..10 (the next iteration's block 0)
    if (!(XKV_ABL & 4) && i == 1) x.pf[3] = *reinterpret_cast<const mfma_bf16x8*>(L.smem + c.pb2 + L.pr + 11 * 1024);                                                                                  // tile I - 2, k-step 11
    // the row sums of tile I - 1: four waves' shares, the four lane groups (v_permlane16_swap / v_permlane32_swap: VALU, no LDS round trip)
    if (!(XKV_ABL & 4) && i == 39) { c.lsrc[0] = *reinterpret_cast<const float*>(L.smem + c.pb1 + L.lr); c.lsrc[1] = *reinterpret_cast<const float*>(L.smem + c.pb1 + L.lr + 256); }
    if (!(XKV_ABL & 4) && i == 43) { c.lsrc[2] = *reinterpret_cast<const float*>(L.smem + c.pb1 + L.lr + 512); c.lsrc[3] = *reinterpret_cast<const float*>(L.smem + c.pb1 + L.lr + 768); }
    if (i == 47) c.lt = (c.lsrc[0] + c.lsrc[1]) + (c.lsrc[2] + c.lsrc[3]);
    if (i == 51) {
      const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(c.lt), __float_as_uint(c.lt), false, false);
      c.lt = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    if (i == 55) {
      const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(c.lt), __float_as_uint(c.lt), false, false);
      c.lt = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    if (i == 59) {
      c.inv = __builtin_amdgcn_rcpf(c.lt);
      asm volatile("" : "+v"(c.inv));
      // (no branch inside the stream: a block boundary lets the code sinker pull values down to their first use)
      c.bad = !(c.lt >= X_MIN_ROWSUM && c.lt <= X_MAX_ROWSUM) ? 1 : 0;
    }
    if (i == 51 || i == 55 || i == 59 || i == 63) {   // the next tile's Q fragment ks (last read by slot 50 + 4 ks)
      constexpr int ks = ((i - 51) >> 2) & 3;
      x.qf[ks] = *(lds_frag*)(L.lds + c.qoff + L.q[ks]);
    }
  }
  SB();
}
template <int... I>
__device__ __forceinline__ void xkv_iter(XState& x, const mfma_bf16x8 (&kf)[8][4], const mfma_bf16x8 (&vf)[2][16], XIter& c, const XLane& L, XQ& q,
                                         std::integer_sequence<int, I...>) {
  (xkv_slot<I>(x, kf, vf, c, L, q), ...);
}

__global__ __launch_bounds__(256) void attn_xkv_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ Kg, const bf16_t* __restrict__ Vt,
                                                      bf16_t* __restrict__ O, int B, int Bk, int64_t Lq, int64_t ldv, int H, int nqb,
                                                      int* __restrict__ wg_flags) {
  constexpr int Lk = 512;
  __shared__ __attribute__((aligned(16))) char smem[XQ_BYTES + 2 * XP_BUF];
  lds_cchar* lds = (lds_cchar*)smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  const int total = nqb * H * B;
  const int per = (total + (int)gridDim.x - 1) / (int)gridDim.x;
  int v = uni((int)blockIdx.x * per);
  const int v_end = uni(v + per < total ? v + per : total);
  if (v >= v_end) return;
  const int64_t rs = (int64_t)H * 128;
  const uint32_t rs2 = (uint32_t)(rs * 2);

  // ---- loop-invariant per-lane addresses ------------------------------------------------------------------------------------------
  // Q DMA piece of this wave: rows 4 wave + g of a tile, 16-byte slot n <- chunk n ^ row (attention_w16n.hip's Q area swizzle)
  const uint32_t qrow = (uint32_t)(4 * wave + g);
  const uint32_t qvoff = qrow * rs2 + (((uint32_t)n ^ qrow) << 4);
  uint32_t qlane[4];   // Q fragment ks inside a ring slot: row n, chunk 4 ks + g
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qlane[ks] = (uint32_t)(n * 256 + (((ks * 4 + g) ^ n) << 4));
  const uint32_t pw_lane = (uint32_t)(XQ_BYTES + (4 * wave * 64 + lane) * 16);      // P^T write: [4 wave + pair][lane]
  const uint32_t pr_lane = (uint32_t)(XQ_BYTES + lane * 16);                        // P^T read: [k-step][lane]
  const uint32_t lw_lane = (uint32_t)(XQ_BYTES + XP_P + (wave * 64 + lane) * 4);
  const uint32_t lr_lane = (uint32_t)(XQ_BYTES + XP_P + lane * 4);
  const uint32_t st_lane = (uint32_t)n * rs2 + (uint32_t)(wave * 64 + (g & 1) * 32 + (g >> 1) * 16);   // O store: row n, channels 32 wave + 16 (g & 1) + 8 (g >> 1) .. + 7

  XLane L;
  L.smem = smem; L.lds = lds; L.pw = pw_lane; L.pr = pr_lane; L.lw = lw_lane; L.lr = lr_lane; L.st = st_lane;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) L.q[ks] = qlane[ks];
#ifdef XKV_STAMPS
  const uint64_t t_start = __builtin_amdgcn_s_memtime();
  int n_iter = 0;
#endif
  XState x;
#pragma unroll
  for (int i = 0; i < 8; ++i) x.s[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  x.o[0] = x.o[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  x.pk = xkv_u4{0u, 0u, 0u, 0u};
#pragma unroll
  for (int i = 0; i < 4; ++i) x.pf[i] = __builtin_bit_cast(mfma_bf16x8, xkv_u4{0u, 0u, 0u, 0u});
  x.lacc = x.lh = x.inv = 0.f;

  while (v < v_end) {
    // ---- a run: the blocks of ONE (batch, head) pair ------------------------------------------------------------------------------
    const int pair = v / nqb, qb0 = v - pair * nqb;
    const int b = pair / H, h = pair - b * H, bk = b % Bk;
    int nblk = nqb - qb0;
    if (nblk > v_end - v) nblk = v_end - v;
    const int64_t row0 = (int64_t)qb0 * 256;
    int64_t rows64 = (int64_t)(qb0 + nblk) * 256;
    if (rows64 > Lq) rows64 = Lq;
    const int run_rows = uni((int)(rows64 - row0));
    const int n_tiles = uni((run_rows + 15) >> 4);

    // K / V^T of this wave's 128 keys -> the accumulator file
    mfma_bf16x8 kf[8][4], vf[2][16];
    {
      const bf16_t* kb = Kg + ((int64_t)bk * Lk + wave * 128) * rs + (int64_t)h * 128 + g * 8;
      const bf16_t* vb = Vt + ((int64_t)bk * H * 128 + (int64_t)h * 128 + 32 * wave + n) * ldv + g * 8;
#pragma unroll
      for (int kt = 0; kt < 8; ++kt) {
        const int row = 32 * (kt >> 1) + 8 * (n >> 2) + 4 * (kt & 1) + (n & 3);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kf[kt][ks] = *reinterpret_cast<const mfma_bf16x8*>(kb + (int64_t)row * rs + ks * 32);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int C = 0; C < 16; ++C) vf[j][C] = *reinterpret_cast<const mfma_bf16x8*>(vb + (int64_t)(16 * j) * ldv + 32 * C);
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) asm volatile("" : "+a"(kf[a][c]));
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int C = 0; C < 16; ++C) asm volatile("" : "+a"(vf[j][C]));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- the Q stream of the run -------------------------------------------------------------------------------------------------
    const char* qbase = uni(reinterpret_cast<const char*>(Q + ((int64_t)b * Lq + row0) * rs + (int64_t)h * 128));
    char* obase = const_cast<char*>(uni(reinterpret_cast<const char*>(O + ((int64_t)b * Lq + row0) * rs + (int64_t)h * 128)));
    XQ q;
    q.base = qbase; q.left = run_rows; q.next = 0; q.voff = qvoff; q.rs2 = rs2; q.lds = lds0 + (uint32_t)(wave * 1024);
#pragma unroll 1
    for (int t = 0; t < XR - 1; ++t) xkv_q_issue(q);
    asm volatile("s_waitcnt vmcnt(14)" ::: "memory");   // tile 0 (14 younger pieces may fly)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) x.qf[ks] = *(lds_frag*)(lds + qlane[ks]);

    // ---- iterations: S^T of tile it, O^T of tile it - 1 (k-steps 0..11), O^T k-steps 12..15 and the store of tile it - 2 --------------------
    int o_left = run_rows + 32;   // rows from tile it - 2's first to the run's last
    for (int it = 0; it < n_tiles + 2; ++it) {
      XIter c;
      c.pb1 = (uint32_t)(((it + 1) & 1) * XP_BUF);
      c.pb2 = (uint32_t)((it & 1) * XP_BUF);
      c.qoff = (uint32_t)(((it + 1) & (XR - 1)) * XQT);
      int valid = o_left > 16 ? 16 : o_left;
      if (it < 2) valid = 0;                                          // (it - 2 < n_tiles by the loop bound)
      const uint32_t onum = valid > 0 ? (uint32_t)(valid - 1) * rs2 + 256u : 0u;
      c.odesc = __builtin_amdgcn_make_buffer_rsrc((void*)(obase + (int64_t)(it - 2) * 16 * (int64_t)rs2), 0, (int)onum, 0x00020000);
      c.lt = 0.f; c.inv = 0.f; c.bad = 0;
      c.lsrc[0] = c.lsrc[1] = c.lsrc[2] = c.lsrc[3] = 1.f;
#ifdef XKV_STAMPS
      c.rec = blockIdx.x == 0 && it == 200;
#endif
      xkv_iter(x, kf, vf, c, L, q, std::make_integer_sequence<int, 64>{});
#ifdef XKV_STAMPS
      if (c.rec && lane == 0) {
        const uint64_t t1 = __builtin_amdgcn_s_memtime();
        for (int k6 = 0; k6 < 66; ++k6) xkv_stamps[wave][k6] = c.st[k6];
        xkv_stamps[wave][66] = t1;
      }
#endif
#ifdef XKV_STAMPS
      n_iter += 1;
#endif
      x.inv = c.inv;
      // the verdict of tile it - 1 (every wave holds the same sums: thread 0 speaks)
      if (__builtin_expect(it >= 1 && it <= n_tiles && __builtin_amdgcn_ballot_w64(c.bad != 0) != 0, 0)) {
        if (tid == 0) wg_flags[v + ((it - 1) >> 4)] = 1;
      }
      o_left -= 16;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // every piece landed, every wave is done with the ring before the next run refills it
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    v += nblk;
  }
#ifdef XKV_STAMPS
  if (blockIdx.x == 0 && lane == 0) { xkv_stamps[wave][67] = __builtin_amdgcn_s_memtime() - t_start; xkv_stamps[wave][68] = (uint64_t)n_iter; }
#endif
}

}  // namespace

// The launch (attention_w64q.hip's short-KV branch): one workgroup per CU walks total = nqb x H x B blocks.  wg_flags: one word per block, zeroed
// by the caller; the tracking launch behind this one redoes what is flagged.
int wan_attention_xkv_launch(unsigned grid, hipStream_t stream, const bf16_t* q, const bf16_t* k, const bf16_t* vt, bf16_t* o, int B, int Bk, int64_t Lq,
                             int64_t ldv, int H, int nqb, int* wg_flags) {
  hipLaunchKernelGGL(attn_xkv_kernel, dim3(grid), dim3(256), 0, stream, q, k, vt, o, B, Bk, Lq, ldv, H, nqb, wg_flags);
  WAN_LAUNCH_CHECK();
#ifdef XKV_STAMPS
  {
    static int printed = 0;
    uint64_t h[4][80];
    WAN_CHECK_HIP(hipStreamSynchronize(stream));
    WAN_CHECK_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(xkv_stamps), sizeof(h)));
    if (printed++ < 2)
      for (int w = 0; w < 4; ++w) {
        fprintf(stderr, "xkv stamps wave %d: iteration %lld; wait+barrier at 24: %lld + %lld; per 8 slots:", w, (long long)(h[w][66] - h[w][0]),
                (long long)(h[w][64] - h[w][24]), (long long)(h[w][65] - h[w][64]));
        for (int i = 0; i < 64; i += 8) fprintf(stderr, " %lld", (long long)((i < 56 ? h[w][i + 8] : h[w][66]) - h[w][i]));
        fprintf(stderr, "; kernel %lld ticks, %lld iterations\n", (long long)h[w][67], (long long)h[w][68]);
      }
  }
#endif
  return 0;
}
