// Text cross-attention for gfx950 with K and V^T STATIONARY in registers ("xkv", round 6).
//   WanT2VCrossAttention / text_cross_attention (models/wan/modules/model.py:245-307, :410-445) -> pay_attention (shared/attention.py:360-373)
//   at Lk = 512 text tokens, head_dim 128, q pre-scaled by scale * log2(e) (what csrc/dit.hip passes).
//
// Why another kernel.  attention_w16n.hip's persistent walk (round 4) streams the head's 8 K / V^T tiles through its LDS ring for EVERY
// 256-row q block: 16.4k matrix cycles per block inside ~36k -- 43.5 % matrix-pipe busy at 1.89 GHz, 0.34 of peak -- because one workgroup per
// CU has nobody to run while it judges rows, stores O and restarts its ring (docs/history section 3.1).  With 512 keys the
// roles can be swapped: a head's K and V^T are 256 KB = exactly the 4 x 64 lanes x 256 accumulator-file registers of a workgroup.  Here
//   * wave w keeps keys 128 w .. 128 w + 127 of the head: K as 8 x 4 A-fragments (S^T = K Q^T), V^T as 8 x 4 A-fragments (O^T = V^T P^T), loaded
//     ONCE per (batch, head) run of the workgroup's walk, pinned to the accumulator file;
//   * Q rows stream through a 16-deep LDS-DMA ring in tiles of 16 rows (4 KB; one 1-KB piece per wave per tile, 15 tiles ahead);
//   * per tile a wave issues 32 MFMAs of S^T (its 128 keys), 32 exp2, 32 MFMAs of O^T (partial over its keys), then the four partial O^T
//     (and row sums) are added through LDS: wave w ends up with d = 32 w .. 32 w + 31 of the tile's 16 rows, normalises and stores them;
//   * everything is ONE software-pipelined stream -- S of tile t, O^T of tile t - 1, the sum / store of tile t - 2 in one iteration of 64 MFMA
//     slots -- so a q block has no prologue, no verdict stage and no epilogue of its own; the only serial work left is the K / V^T load of a run
//     (once or twice per workgroup) and two pipeline-fill iterations per run.
// The bounded softmax has no running maximum (P = 2^s, as attention_w16n.hip's plain loop): its partial sums over disjoint keys add exactly.
// A row is sound when its row sum lies in [2^-80, 2^100] (no score overflowed, the sum did not underflow; attention_w16n.hip's verdict with
// m = 0); a tile with an unsound row flags its 256-row block 1 in wg_flags and the tracking launch that follows every bounded launch redoes
// that block -- which re-reads its Q rows, so this kernel serves OUT-OF-PLACE calls only (o != q; the launcher sends in-place calls to the
// persistent walk).
//
// Fragment layout (v_mfma_f32_16x16x32_bf16; lane: n = lane & 15, g = lane >> 4): attention_w16n.hip's.
//   S^T tile kt (16 key slots x 16 q): A = K fragment: lane (n, g) holds key slot 16 kt + n, d = 32 ks + 8 g .. + 7; B = Q fragment: q row n, the
//   same d.  Register i of lane (n, g) = score of q row n against slot 16 kt + 4 g + i.  Slot (kt, m) holds key 32 (kt >> 1) + 8 (m >> 2) +
//   4 (kt & 1) + (m & 3) of the wave's 128, so that registers 0..3 of tiles 2 c, 2 c + 1 are the 8 CONSECUTIVE keys 32 c + 8 g .. + 7: the P^T
//   B-fragment of k-step c without moving anything between lanes, against a V^T A-fragment read straight from memory.
//   O^T tile dt: register i of lane (n, g) = O[q row n][d = 16 dt + 4 g + i].
//
// One iteration = 64 slots of one MFMA + its fillers, four groups of [8 x S^T (tile pair p = keys 32 p ..) | 8 x O^T (k-step c = p)]:
//   exp2 of pair p's 8 scores in slots 16 p + 10 .. 17 (>= 3 MFMAs behind the tile's last MFMA: asm MFMAs are not padded by hipcc), in place;
//   the pack of P^T fragment p in slots 16 p + 18 .. 21 (its previous value was last read by slot 16 p + 15);
//   pair 3's tail (2 exp2, 4 packs) in slots 0 .. 5 of the next iteration;
//   slots 3 .. 11: the finished partial O^T of the previous tile (last written by slot 56 + dt) -> LDS, slot 24: barrier, slots 26 .. 37: the
//   other waves' partials of this wave's quarter, slots 38 .. 55: sums, 1 / l, bf16, two 8-byte stores per lane;
//   slots 50 .. 56: the next tile's Q fragments (Q fragment ks is last read by slot 49 + 2 ks).
#include <stdlib.h>

#include <utility>

#include "attn_w64_shared.h"

namespace {

constexpr int XR = 16;                          // Q ring depth (tiles)
constexpr int XQT = 4096;                       // bytes per Q tile (16 rows x 256 B)
constexpr int XQ_BYTES = XR * XQT;              // 64 KB
constexpr int XP_O = 4 * 8 * 1024;              // partial O^T of one tile: [wave][d tile][lane] x 16 B
constexpr int XP_BUF = XP_O + 4 * 256;          // + row-sum shares [wave][lane] x 4 B
constexpr float X_MIN_ROWSUM = 8.271806125530277e-25f;   // 2^-80
constexpr float X_MAX_ROWSUM = 1.2676506002282294e30f;   // 2^100

typedef uint32_t xkv_u4 __attribute__((ext_vector_type(4)));
typedef unsigned int xkv_st2 __attribute__((__vector_size__(8)));

__device__ __forceinline__ void xs0(f32x4& d, const mfma_bf16x8& k, const mfma_bf16x8& q) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(d) : "a"(k), "v"(q));
}
__device__ __forceinline__ void xs1(f32x4& d, const mfma_bf16x8& k, const mfma_bf16x8& q) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d) : "a"(k), "v"(q));
}

struct XState {
  mfma_bf16x8 qf[4];   // Q fragments of the tile whose S^T runs in this iteration
  f32x4 s[8];          // S^T tiles, then P in place
  xkv_u4 pk[4];        // P^T fragments of the tile whose O^T runs in this iteration
  f32x4 o[8];          // partial O^T (this wave's 128 keys)
  float lacc, lh, lw;  // this lane's share of the row sums: being summed / of the tile in O^T / of the tile being added up
};

// MFMA of slot i
__device__ __forceinline__ void xkv_mfma(XState& x, const mfma_bf16x8 (&kf)[8][4], const mfma_bf16x8 (&vf)[8][4], int i) {
  const int p = i >> 4, w = i & 15;
  if (w < 8) {
    const int kt = 2 * p + (w & 1), ks = w >> 1;
    if (ks == 0) xs0(x.s[kt], kf[kt][0], x.qf[0]);
    else xs1(x.s[kt], kf[kt][ks], x.qf[ks]);
  } else {
    const int dt = w - 8;
    if (p == 0) xs0(x.o[dt], vf[dt][0], __builtin_bit_cast(mfma_bf16x8, x.pk[0]));
    else xs1(x.o[dt], vf[dt][p], __builtin_bit_cast(mfma_bf16x8, x.pk[p]));
  }
}
__device__ __forceinline__ void xkv_exp(XState& x, int e) {   // e = 0..63: position in the exp2 stream of a tile (pair e >> 4, its scores 0..7 at e & 15 < 8)
  const int p = e >> 4, r = e & 15;
  if (r >= 8) return;
  float v = __builtin_amdgcn_exp2f(x.s[2 * p + (r >> 2)][r & 3]);
  asm volatile("" : "+v"(v));
  x.s[2 * p + (r >> 2)][r & 3] = v;
  x.lacc += v;
  asm volatile("" : "+v"(x.lacc));
}
__device__ __forceinline__ void xkv_pack(XState& x, int k) {  // k = 0..63: pair k >> 4, word k & 15 < 4 of its fragment
  const int p = k >> 4, r = k & 15;
  if (r >= 4) return;
  const f32x4& t = x.s[2 * p + (r >> 1)];
  x.pk[p][r] = cvt_pk(t[2 * (r & 1)], t[2 * (r & 1) + 1]);
  asm volatile("" : "+v"(x.pk[p]));
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
  int tid;
};
struct XIter {         // one iteration's values
  uint32_t pbuf, qoff;
  bool live;
  __amdgpu_buffer_rsrc_t odesc;
  int* flag;
  f32x4 rd[4][2], acc0, acc1;
  float lsrc[4], lt, inv;
  int bad;
#ifdef XKV_STAMPS
  bool rec;
  uint64_t st[66];
#endif
};
// slot I of an iteration: its MFMA, then its fillers
template <int I>
__device__ __forceinline__ void xkv_slot(XState& x, const mfma_bf16x8 (&kf)[8][4], const mfma_bf16x8 (&vf)[8][4], XIter& c, const XLane& L, XQ& q) {
  constexpr int i = I;
  XST(i);
  xkv_mfma(x, kf, vf, i);
  SB();
  xkv_exp(x, i >= 10 ? i - 10 : i + 54);
  if (i == 2) { x.lw = x.lh; x.lh = x.lacc; x.lacc = 0.f; asm volatile("" : "+v"(x.lw), "+v"(x.lh), "+v"(x.lacc)); }
  xkv_pack(x, i >= 18 ? i - 18 : i + 46);
  if (i >= 3 && i <= 10) {   // the finished partial O^T tile dt = i - 3 (last written by slot 56 + dt of the previous iteration; rewritten by slot 8 + dt)
    constexpr int dt = (i - 3) & 7;
    *reinterpret_cast<f32x4*>(L.smem + c.pbuf + L.pw + dt * 1024) = x.o[dt];
  }
  if (i == 11) *reinterpret_cast<float*>(L.smem + c.pbuf + L.lw) = x.lw;
  if (i == 22) xkv_q_issue(q);
  if (i == 24) {
    asm volatile("s_waitcnt vmcnt(14) lgkmcnt(0)" ::: "memory");   // this wave's piece of tile it + 1 landed; its partials are written
    XST(64);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    XST(65);
  }
  if (i >= 26 && i <= 33) {   // the four waves' partials of this wave's d tiles 2 wave, 2 wave + 1
    constexpr int r = (i - 26) & 7, src = r >> 1, j = r & 1;
    c.rd[src][j] = *reinterpret_cast<const f32x4*>(L.smem + c.pbuf + L.pr + src * 8192 + j * 1024);
  }
  if (i >= 34 && i <= 37) c.lsrc[(i - 34) & 3] = *reinterpret_cast<const float*>(L.smem + c.pbuf + L.lr + ((i - 34) & 3) * 256);
  if (i == 38) { c.acc0 = c.rd[0][0] + c.rd[1][0]; c.acc1 = c.rd[0][1] + c.rd[1][1]; asm volatile("" : "+v"(c.acc0), "+v"(c.acc1)); }
  if (i == 39) { c.acc0 += c.rd[2][0]; c.acc1 += c.rd[2][1]; asm volatile("" : "+v"(c.acc0), "+v"(c.acc1)); }
  if (i == 40) { c.acc0 += c.rd[3][0]; c.acc1 += c.rd[3][1]; asm volatile("" : "+v"(c.acc0), "+v"(c.acc1)); }
  if (i == 41) c.lt = (c.lsrc[0] + c.lsrc[1]) + (c.lsrc[2] + c.lsrc[3]);
  // the row's sum over the four lane groups: v_permlane16_swap / v_permlane32_swap (VALU; a ds_bpermute's round trip would sit in the stream twice)
  if (i == 42) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(c.lt), __float_as_uint(c.lt), false, false);
    c.lt = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  if (i == 43) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(c.lt), __float_as_uint(c.lt), false, false);
    c.lt = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  if (i == 44) {
    c.inv = __builtin_amdgcn_rcpf(c.lt);
    asm volatile("" : "+v"(c.inv));
    // (no branch inside the stream: a block boundary lets the code sinker pull the sums above down to their first use)
    c.bad = !(c.lt >= X_MIN_ROWSUM && c.lt <= X_MAX_ROWSUM) ? 1 : 0;
  }
  if (i == 48) {
    xkv_st2 w0;
    w0[0] = cvt_pk(c.acc0[0] * c.inv, c.acc0[1] * c.inv);
    w0[1] = cvt_pk(c.acc0[2] * c.inv, c.acc0[3] * c.inv);
    __builtin_amdgcn_raw_buffer_store_b64(w0, c.odesc, (int)L.st, 0, 0);
  }
  if (i == 49) {
    xkv_st2 w1;
    w1[0] = cvt_pk(c.acc1[0] * c.inv, c.acc1[1] * c.inv);
    w1[1] = cvt_pk(c.acc1[2] * c.inv, c.acc1[3] * c.inv);
    __builtin_amdgcn_raw_buffer_store_b64(w1, c.odesc, (int)(L.st + 32u), 0, 0);
  }
  if (i == 50 || i == 52 || i == 54 || i == 56) {   // the next tile's Q fragment ks (last read by slot 49 + 2 ks)
    constexpr int ks = ((i - 50) >> 1) & 3;
    x.qf[ks] = *(lds_frag*)(L.lds + c.qoff + L.q[ks]);
  }
  SB();
}
template <int... I>
__device__ __forceinline__ void xkv_iter(XState& x, const mfma_bf16x8 (&kf)[8][4], const mfma_bf16x8 (&vf)[8][4], XIter& c, const XLane& L, XQ& q,
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
  const uint32_t pw_lane = (uint32_t)(XQ_BYTES + (wave * 512 + lane) * 16);        // partial write: [wave][dt][lane]
  const uint32_t pr_lane = (uint32_t)(XQ_BYTES + (2 * wave * 64 + lane) * 16);      // partial read: [src][2 wave + j][lane]
  const uint32_t lw_lane = (uint32_t)(XQ_BYTES + XP_O + (wave * 64 + lane) * 4);
  const uint32_t lr_lane = (uint32_t)(XQ_BYTES + XP_O + lane * 4);
  const uint32_t st_lane = (uint32_t)n * rs2 + (uint32_t)(wave * 64 + g * 8);       // O store: row n, d = 32 wave + 4 g (+ 16)

  XLane L;
  L.smem = smem; L.lds = lds; L.pw = pw_lane; L.pr = pr_lane; L.lw = lw_lane; L.lr = lr_lane; L.st = st_lane; L.tid = tid;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) L.q[ks] = qlane[ks];
  XState x;
#pragma unroll
  for (int i = 0; i < 8; ++i) { x.s[i] = f32x4{0.f, 0.f, 0.f, 0.f}; x.o[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int i = 0; i < 4; ++i) x.pk[i] = xkv_u4{0u, 0u, 0u, 0u};
  x.lacc = x.lh = x.lw = 0.f;

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
    mfma_bf16x8 kf[8][4], vf[8][4];
    {
      const bf16_t* kb = Kg + ((int64_t)bk * Lk + wave * 128) * rs + (int64_t)h * 128 + g * 8;
      const bf16_t* vb = Vt + ((int64_t)bk * H * 128 + (int64_t)h * 128 + n) * ldv + wave * 128 + g * 8;
#pragma unroll
      for (int kt = 0; kt < 8; ++kt) {
        const int row = 32 * (kt >> 1) + 8 * (n >> 2) + 4 * (kt & 1) + (n & 3);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kf[kt][ks] = *reinterpret_cast<const mfma_bf16x8*>(kb + (int64_t)row * rs + ks * 32);
      }
#pragma unroll
      for (int dt = 0; dt < 8; ++dt)
#pragma unroll
        for (int c = 0; c < 4; ++c) vf[dt][c] = *reinterpret_cast<const mfma_bf16x8*>(vb + (int64_t)(16 * dt) * ldv + 32 * c);
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) { asm volatile("" : "+a"(kf[a][c])); asm volatile("" : "+a"(vf[a][c])); }
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

    // ---- iterations: S^T of tile it, O^T of tile it - 1, sum / store of tile it - 2 ------------------------------------------------------
    int o_left = run_rows + 32;   // rows from tile it - 2's first to the run's last (<= 0 or > run_rows: no such tile)
    for (int it = 0; it < n_tiles + 2; ++it) {
      const uint32_t pbuf = (uint32_t)((it & 1) * XP_BUF);
      const uint32_t qoff = (uint32_t)(((it + 1) & (XR - 1)) * XQT);
      const bool live = it >= 2;                                     // (it - 2 < n_tiles by the loop bound)
      int valid = o_left > 16 ? 16 : o_left;
      if (!live) valid = 0;
      const uint32_t onum = valid > 0 ? (uint32_t)(valid - 1) * rs2 + 256u : 0u;
      const __amdgpu_buffer_rsrc_t odesc = __builtin_amdgcn_make_buffer_rsrc((void*)(obase + (int64_t)(it - 2) * 16 * (int64_t)rs2), 0, (int)onum, 0x00020000);
      XIter c;
      c.pbuf = pbuf; c.qoff = qoff; c.live = live; c.odesc = odesc; c.flag = wg_flags + v + ((it - 2) >> 4);
      c.lt = 0.f; c.inv = 0.f; c.bad = 0;
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
      if (__builtin_expect(live && __builtin_amdgcn_ballot_w64(c.bad != 0) != 0, 0)) {   // (every wave holds the same sums: thread 0 speaks)
        if (tid == 0) *c.flag = 1;
      }
      o_left -= 16;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // every piece landed, every wave is done with the ring before the next run refills it
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    v += nblk;
  }
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
        fprintf(stderr, "\n");
      }
  }
#endif
  return 0;
}
