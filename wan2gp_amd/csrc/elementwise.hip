// Memory-bound fused row kernels of the Wan DiT block (gfx950, wave64).
//
// One 64-lane wave owns one token row; a row of d bf16 is held in registers as NCH chunks of
// 8 bf16 per lane (16-byte coalesced loads: lane i reads chunk i, i+64, ...), statistics are
// reduced with DPP/bpermute shuffles only (no LDS, no barrier), and the row is written back once.
// HBM traffic = the algorithmic bytes (read + write of the row); cos/sin and the [d] vectors
// are L2-resident.
//
// Rounding points follow the reference's eager bf16 graph (see include/wanhip.h).
#include <string.h>

#include <algorithm>
#include <mutex>

#include "common.h"

#define ROWS_PER_BLOCK 4  // 4 waves of 64 lanes

#ifndef RMSROPE_WG
#define RMSROPE_WG 3
#endif
#ifndef ROPE_FORM
#define ROPE_FORM 1
#endif
// ------------------------------------------------------------------------------------------------
// RMSNorm(q,k) + RoPE  -- model.py:160-175, posemb_layers.py:251-269
// ------------------------------------------------------------------------------------------------
// PERSIST = true (wide rows, d >= 4096): the grid is the resident set, a wave walks rows row0, row0 + stride, ... and issues the
// 16-byte loads of its NEXT row before it touches the current one, so every resident wave keeps one row (d * 2 bytes) in flight
// for the whole time it computes -- the bytes in flight no longer depend on how many waves the register budget admits (one row
// per wave holds 3 waves per SIMD at d = 5120).  Measured, RMSNorm+RoPE at the 14B-720p shape: 4.30-4.33 TB/s against 3.56.
// PERSIST = false (one row per wave, grid = rows / 4): narrow rows, where 7 waves per SIMD already cover the latency and the
// persistent form loses a third (d = 1536: 3.5 against 5.2 TB/s), and the LayerNorm family, which is VALU-bound either way
// (~30 VALU operations per element at the reference's rounding points; 3.3 TB/s both ways at d = 5120).
// Rows stay PACKED in registers (4 VGPRs per chunk), unpacked again in each pass.
template <int NCH, bool FULL = false>
__device__ __forceinline__ void load_row(uint4 (&r)[NCH], const bf16_t* x, int lane, int nchunk) {
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    if (FULL || c < nchunk) r[i] = *reinterpret_cast<const uint4*>(x + c * 8);
  }
}

// x0' = x0*cos0 - x1*sin0 ; x1' = x1*cos1 + x0*sin1 -- fp32, the four products rounded SEPARATELY (posemb_layers.py:251-267: no fused
// multiply-add), one bf16 rounding by the caller.  The pair-wise form of rounds 3-5: only diagnostics builds (ROPE_FORM 0 / 2) call it.
__device__ __forceinline__ wan_f32x2 rope_pair(wan_f32x2 y, float c0, float c1, float s0, float s1) {
#pragma clang fp contract(off)
  const wan_f32x2 cc = {c0, c1};
  const wan_f32x2 ss = {-s0, s1};
  const wan_f32x2 p = y * cc;
#if ROPE_FORM == 2     // diagnostics: the packed crosswise multiply, but never onto its own source pair
  wan_f32x2 q;
  asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(q) : "v"(ss), "v"(y));
#else                  // hipcc fuses this into v_pk_mul_f32 ... op_sel:[0,1] op_sel_hi:[1,0]
  const wan_f32x2 ys = {y.y, y.x};
  const wan_f32x2 q = ys * ss;      // (-x1*sin0, x0*sin1): the sign flip of a product is exact
#endif
  return p + q;
}

// SCATTER (round 6; the Ulysses exchange's send layout, csrc/dit.hip): the rows are written to `pack` head-group-major instead of in
// place -- 16-byte chunk c of row r (head c >> 4 = dest rank w, head hl of its `hn`, head chunk j of `nchunk_h` with heads [h0_j, h0_j+1))
// lands at pack + 16 (A[c] + r B[c]) bytes, A[c] = (128 h0_j rows world + w rows Wc_j + 128 (hl - h0_j)) / 8 + (c & 15), B[c] = Wc_j / 8,
// Wc_j = 128 (h0_j+1 - h0_j): exactly what wan_permute16_ex leaves from the in-place result ([rows][world][hn 128] -> [chunk][world][rows][Wc]),
// without the pass over the tensor (4 of a block's 8 re-packs: q and k).  The (A, B) pairs sit in LDS beside the norm weights.
template <int NCH, bool PERSIST, bool ROPE, bool FULL, bool SCATTER = false>
__global__ __launch_bounds__(256, (PERSIST && FULL && NCH <= 10 ? (SCATTER ? 2 : 3) : 1)) void rmsnorm_rope_kernel(   // (SCATTER: two address registers more than three waves per SIMD leave room for)
    bf16_t* __restrict__ q, bf16_t* __restrict__ k, const bf16_t* __restrict__ wq,
    const bf16_t* __restrict__ wk, const float* __restrict__ cosT, const float* __restrict__ sinT,
    int64_t rows, int64_t L, int64_t pos0, int d, float eps, float q_scale, bf16_t* __restrict__ pack = nullptr, int world = 1, int hn = 1,
    int nchunk_h = 1) {
  // FULL: d == NCH * 512 exactly (every Wan width): no per-chunk bounds test -- each test is an exec-masked branch that also keeps
  // the chunk's 64-bit address in a VGPR pair.  The wave index is made scalar so that the row pointers live in SGPRs and every
  // access is `base(SGPR) + lane * 16 (one VGPR) + immediate`: round 3's kernel spent ~40 VGPRs on addresses at d = 5120.
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + wave;
  // The narrow in-place form (one row per wave, no weights in LDS, no address table) had NO LDS and NO barrier -- and was the one kernel of the
  // library that did not return its bits beside another process on the same GPU: in ~10-ms windows of the neighbour's start-up / exit every
  // tenth launch left ~1 % of its rows with a few wrong 16-byte chunks (runs 65-85; the 1.3B forward differed in 10-25 % of its calls, the 14B
  // widths -- the persistent form below -- never).  What was wrong (run 82): in lanes 48-63 the EVEN element of a rotation pair came out as
  // x0 cos0 without its - x1 sin0 -- the low result of the packed crosswise multiply (common.h), which the rotation no longer contains (run 84:
  // 0 of 316,187 launches without the LDS word, 58 of 310,685 with the old rotation).  Found first, and kept: one LDS word per wave (the word
  // cures the old rotation by itself, the barrier alone does not: run 80) -- why a workgroup's LDS allocation shields that instruction is not
  // understood; it costs nothing measurable.
  // (RMSROPE_WG, diagnostics only -- `make rrwg`: 1 = the LDS word without the barrier, 2 = the barrier without LDS, 0 = neither, the form of rounds 3-5)
  if (!PERSIST && !SCATTER && RMSROPE_WG) {
    __shared__ int wg_word[(RMSROPE_WG & 1) ? ROWS_PER_BLOCK : 1];
    if ((RMSROPE_WG & 1) && lane == 0) wg_word[wave] = wave;
    if (RMSROPE_WG & 2) __syncthreads();
    if (RMSROPE_WG & 1) asm volatile("" :: "v"(wg_word[wave ^ 1]));
  }
  if (!PERSIST && !SCATTER && row >= rows) return;
  const int64_t stride = (int64_t)gridDim.x * ROWS_PER_BLOCK;
  const float oscale = (blockIdx.y == 0) ? q_scale : 1.0f;  // q only: fp32 scale folded in front of the ONE bf16 rounding
  bf16_t* base = (blockIdx.y == 0 ? q : k);
  const bf16_t* w = (blockIdx.y == 0 ? wq : wk);
  const int nchunk = d >> 3;

  // PERSIST: the norm weights (one 16-byte chunk per lane and chunk index, the same for every row) live in LDS instead of 4 NCH
  // registers the compiler would otherwise keep across the row loop: 194 -> <= 168 VGPRs at d = 5120 = a third wave per SIMD for a
  // kernel that is bound by the bytes it keeps in flight (round 4; the outputs do not change: same operations on the same values).
  // ALL 256 threads stage the weights and reach the barrier, then a wave without a row leaves: the last block of a grid with
  // rows % 4 != 0 (k == nullptr and rows below the resident set, e.g. a 2,025-token single-frame run) has waves past the end whose
  // share of wlds the live waves read (round 4 returned before the staging: the advisor's finding; tests/test_gpu_ops.py
  // test_rmsnorm_rope_persist_ragged_rows)
  __shared__ uint4 wlds[PERSIST ? NCH * 64 : 1];
  __shared__ uint2 ptab[SCATTER ? NCH * 64 : 1];
  if (PERSIST) {
    for (int c = threadIdx.x; c < nchunk; c += 256) wlds[c] = *reinterpret_cast<const uint4*>(w + c * 8);
  }
  if (SCATTER) {
    for (int c = threadIdx.x; c < nchunk; c += 256) {
      const int head = c >> 4, wr = head / hn, hl = head - wr * hn;
      int j = 0;
      while (j + 1 < nchunk_h && (int)((int64_t)(j + 1) * hn / nchunk_h) <= hl) ++j;
      const int h0 = (int)((int64_t)j * hn / nchunk_h), h1 = (int)((int64_t)(j + 1) * hn / nchunk_h);
      const int64_t wc = (int64_t)(h1 - h0) * 128;
      const int64_t A = (int64_t)h0 * 128 * rows * world + (int64_t)wr * rows * wc + (int64_t)(hl - h0) * 128 + (c & 15) * 8;
      ptab[c] = make_uint2((uint32_t)(A >> 3), (uint32_t)(wc >> 3));
    }
  }
  uint4 raw[NCH], nxt[NCH];
  const bool alive = (!PERSIST && !SCATTER) || row < rows;   // wave-uniform
  if (alive) load_row<NCH, FULL>(raw, base + row * (int64_t)d, lane, nchunk);
  if (PERSIST || SCATTER) {
    __syncthreads();
    if (!alive) return;
  }
  for (;;) {
    const int64_t nrow = row + stride;
    const bool more = PERSIST && nrow < rows;  // wave-uniform
    if (more) load_row<NCH, FULL>(nxt, base + nrow * (int64_t)d, lane, nchunk);
    bf16_t* x = base + row * (int64_t)d;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      if (FULL || c < nchunk) {
        float v[8];
        unpack8(raw[i], v);
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += v[j] * v[j];
      }
      if (PERSIST) __builtin_amdgcn_sched_barrier(0);   // one chunk unpacked at a time (the scheduler otherwise unpacks the whole row: 8 NCH registers)
    }
    ss = wave_sum(ss);
    const float r = rsqrtf(ss / (float)d + eps);
    const int64_t pos = (int64_t)((uint32_t)row % (uint32_t)L) + pos0;  // rows < 2^31 (launcher): 32-bit modulo, a 64-bit one costs ~100 VALU per lane
    // the lane's column inside the 128-wide head is the same for all of its chunks (c = lane + 64 i, 64*8 = 0 mod 128):
    // ONE cos / sin fetch per row instead of one per chunk (the kernel was VMEM-issue bound: 6 loads per 16 B of data)
    float cs[8], sn[8];
    if (ROPE) {
      const int hc = (lane * 8) & 127;
      const float4* cp = reinterpret_cast<const float4*>(cosT + pos * 128 + hc);
      const float4* sp = reinterpret_cast<const float4*>(sinT + pos * 128 + hc);
      const float4 c0 = cp[0], c1 = cp[1], s0 = sp[0], s1 = sp[1];
      cs[0] = c0.x; cs[1] = c0.y; cs[2] = c0.z; cs[3] = c0.w; cs[4] = c1.x; cs[5] = c1.y; cs[6] = c1.z; cs[7] = c1.w;
      sn[0] = s0.x; sn[1] = s0.y; sn[2] = s0.z; sn[3] = s0.w; sn[4] = s1.x; sn[5] = s1.y; sn[6] = s1.z; sn[7] = s1.w;
    }
    uint4 wnext;
    if (PERSIST) wnext = wlds[lane];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      uint4 wcur;
      if (PERSIST) {  // the chunk's weights one chunk ahead, and a scheduling fence per chunk: nothing of chunk i + 2 is live while chunk i computes
        wcur = wnext;
        if (i + 1 < NCH && (FULL || c + 64 < nchunk)) wnext = wlds[c + 64];
      }
      if (FULL || c < nchunk) {
        // Pairs of elements all the way (round 3): the two bf16 of a 32-bit word stay together through both roundings, the rotation and
        // the pack, so that every multiply / add is ONE packed-f32 instruction per pair and every rounding one v_cvt_pk_bf16_f32 per
        // pair (the scalar form rounded with cvt_pk(f, 0): 144 VALU instructions per 16-byte chunk, this form ~90).  Same operations
        // per element in the same order: bit-identical results (tests/test_gpu_ops.py).
        const uint4 wraw = PERSIST ? wcur : *reinterpret_cast<const uint4*>(w + c * 8);
        const uint32_t ww[4] = {wraw.x, wraw.y, wraw.z, wraw.w};
        const uint32_t vw[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
        uint32_t ow[4];
#if ROPE_FORM == 1
        // Two words (= two rotation pairs A, B) at a time, as E = (x0 of A, x0 of B) and O = (x1 of A, x1 of B): every multiply and add of the
        // row is then a packed-f32 instruction on ALIGNED register pairs -- x0' = E C0 + O (-S0), x1' = O C1 + E S1 -- where the pair-wise form
        // (rounds 3-5, below) needed (-x1 sin0, x0 sin1): v_pk_mul_f32 ... op_sel:[0,1] op_sel_hi:[1,0], the instruction that lost its low product
        // beside a neighbour process (common.h).  Same operations per element, same roundings: identical bytes (tools/rows_hash.py).
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma clang fp contract(off)
          const uint32_t va = vw[2 * h], vb = vw[2 * h + 1], wa = ww[2 * h], wb = ww[2 * h + 1];
          wan_f32x2 E = {__uint_as_float(va << 16), __uint_as_float(vb << 16)};
          wan_f32x2 O = {__uint_as_float(va & 0xffff0000u), __uint_as_float(vb & 0xffff0000u)};
          const wan_f32x2 WE = {__uint_as_float(wa << 16), __uint_as_float(wb << 16)};
          const wan_f32x2 WO = {__uint_as_float(wa & 0xffff0000u), __uint_as_float(wb & 0xffff0000u)};
          E = E * r; O = O * r;                                        // x *= rsqrt
          uint32_t ua = pack2bf(E.x, O.x), ub = pack2bf(E.y, O.y);     // bf16 rounding
          E = wan_f32x2{__uint_as_float(ua << 16), __uint_as_float(ub << 16)} * WE;   // x *= weight
          O = wan_f32x2{__uint_as_float(ua & 0xffff0000u), __uint_as_float(ub & 0xffff0000u)} * WO;
          ua = pack2bf(E.x, O.x); ub = pack2bf(E.y, O.y);
          E = wan_f32x2{__uint_as_float(ua << 16), __uint_as_float(ub << 16)};
          O = wan_f32x2{__uint_as_float(ua & 0xffff0000u), __uint_as_float(ub & 0xffff0000u)};
          if (ROPE) {
            const wan_f32x2 C0 = {cs[4 * h], cs[4 * h + 2]}, C1 = {cs[4 * h + 1], cs[4 * h + 3]};
            const wan_f32x2 NS0 = {-sn[4 * h], -sn[4 * h + 2]}, S1 = {sn[4 * h + 1], sn[4 * h + 3]};   // the sign flip of a product is exact
            const wan_f32x2 pe = E * C0, qe = O * NS0, po = O * C1, qo = E * S1;   // four products rounded separately (posemb_layers.py:251-267)
            E = pe + qe; O = po + qo;
          }
          E = E * oscale; O = O * oscale;                              // (x * 1.0f is x: no select on a launch constant inside the row)
          ow[2 * h] = pack2bf(E.x, O.x); ow[2 * h + 1] = pack2bf(E.y, O.y);
        }
#else   // ROPE_FORM 0 / 2: the pair-wise form of rounds 3-5 (diagnostics, `make rrwg`)
#pragma unroll
        for (int pq = 0; pq < 4; ++pq) {
          wan_f32x2 v2 = {__uint_as_float(vw[pq] << 16), __uint_as_float(vw[pq] & 0xffff0000u)};
          const wan_f32x2 w2 = {__uint_as_float(ww[pq] << 16), __uint_as_float(ww[pq] & 0xffff0000u)};
          v2 = v2 * r;                                                 // x *= rsqrt
          uint32_t u = pack2bf(v2.x, v2.y);                            // bf16 rounding
          v2 = wan_f32x2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)} * w2;  // x *= weight
          u = pack2bf(v2.x, v2.y);
          wan_f32x2 y2 = {__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
          if (ROPE) y2 = rope_pair(y2, cs[2 * pq], cs[2 * pq + 1], sn[2 * pq], sn[2 * pq + 1]);
          y2 = y2 * oscale;                                            // (x * 1.0f is x: no select on a launch constant inside the row)
          ow[pq] = pack2bf(y2.x, y2.y);
        }
#endif
        uint4 o;
        o.x = ow[0]; o.y = ow[1]; o.z = ow[2]; o.w = ow[3];
        if (SCATTER) {
          const uint2 ab = ptab[c];
          *reinterpret_cast<uint4*>(pack + ((size_t)(ab.x + (uint32_t)row * ab.y) << 3)) = o;
        } else {
          *reinterpret_cast<uint4*>(x + c * 8) = o;
        }
      }
      if (PERSIST) __builtin_amdgcn_sched_barrier(0);
    }
    if (!more) break;
    for (int i = 0; i < NCH; ++i) raw[i] = nxt[i];
    row = nrow;
  }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm family.  MODE 0: LN + modulate (norm1/norm2), MODE 1: LN affine (norm3),
// MODE 2: LN + modulate with fp32 modulation table (head, model.py:856-862)
// ------------------------------------------------------------------------------------------------
// AMAX (round 5, scaled-fp8 checkpoints): the kernel also leaves max |out| of every `rows_per_amax` rows (= a stream of the joint pass) in
// word 1 of that stream's 64-word quantisation slot, amax[(row / rows_per_amax) * 64 + 1], as float bits of the bf16 values by atomicMax
// (non-negative floats order like unsigned integers; the caller zeroed the words) -- the abs-max pass of the NEXT Linear's activation
// quantisation (scaled_fp8.py:162-169) folded into the producer of its input: the tensor is not read a second time for its maximum.
template <int NCH, int MODE, bool PERSIST = false, bool AMAX = false>
__global__ __launch_bounds__(256) void layernorm_kernel(
    const bf16_t* __restrict__ xin, bf16_t* __restrict__ out, const void* __restrict__ p0,
    const bf16_t* __restrict__ p1, int n_mod, int shift_idx, int scale_idx, int64_t rows,
    int64_t rows_per_batch, int d, float eps, unsigned int* __restrict__ amax = nullptr, int64_t rows_per_amax = 1) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + wave;
  if (row >= rows) return;
  const int64_t stride = (int64_t)gridDim.x * ROWS_PER_BLOCK;
  const int nchunk = d >> 3;

  uint4 raw[NCH], nxt[NCH];  // rows packed in registers; the PERSIST form exists but is not launched (see above)
  uint32_t am = 0;
  load_row<NCH>(raw, xin + row * (int64_t)d, lane, nchunk);
  for (;;) {
    const int64_t nrow = row + stride;
    const bool more = PERSIST && nrow < rows;  // wave-uniform
    if (more) load_row<NCH>(nxt, xin + nrow * (int64_t)d, lane, nchunk);
    bf16_t* o = out + row * (int64_t)d;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
        float v[8];
        unpack8(raw[i], v);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
      }
    }
    const float mean = wave_sum(s) / (float)d;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
        float v[8];
        unpack8(raw[i], v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float t = v[j] - mean;
          ss += t * t;
        }
      }
    }
    const float rstd = rsqrtf(wave_sum(ss) / (float)d + eps);
    const int64_t b = (int64_t)((uint32_t)row / (uint32_t)rows_per_batch);  // 32-bit: a 64-bit division costs ~100 VALU per lane

#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
        float y[8], v[8];
        unpack8(raw[i], v);
        if (MODE == 1) {
          const bf16_t* w = reinterpret_cast<const bf16_t*>(p0);
          float wf[8], bfv[8];
          unpack8(*reinterpret_cast<const uint4*>(w + c * 8), wf);
          unpack8(*reinterpret_cast<const uint4*>(p1 + c * 8), bfv);
#pragma unroll
          for (int j = 0; j < 8; ++j) y[j] = (v[j] - mean) * rstd * wf[j] + bfv[j];
        } else if (MODE == 0) {
          const bf16_t* mod = reinterpret_cast<const bf16_t*>(p0);
          float msh[8], msc[8], esh[8], esc[8];
          unpack8(*reinterpret_cast<const uint4*>(mod + (int64_t)shift_idx * d + c * 8), msh);
          unpack8(*reinterpret_cast<const uint4*>(mod + (int64_t)scale_idx * d + c * 8), msc);
          const bf16_t* eb = p1 + b * (int64_t)n_mod * d;
          unpack8(*reinterpret_cast<const uint4*>(eb + (int64_t)shift_idx * d + c * 8), esh);
          unpack8(*reinterpret_cast<const uint4*>(eb + (int64_t)scale_idx * d + c * 8), esc);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float ln = rbf((v[j] - mean) * rstd);           // F.layer_norm -> bf16
            const float sc = rbf(1.0f + rbf(wan_add_f32(msc[j], esc[j])));    // 1 + e[scale]    (bf16 ops; plain adds: common.h)
            const float sh = rbf(wan_add_f32(msh[j], esh[j]));                // e[shift]
            y[j] = rbf(ln * sc) + sh;                             // x *= 1+scale ; x += shift
          }
        } else if (MODE == 3) {
          // LN + modulate against a PRECOMPUTED table [batch][2][d] (mod_table_kernel below): row 0 = rbf(1 + rbf(mod + e)[scale]),
          // row 1 = rbf((mod + e)[shift]) -- the two vectors MODE 0 re-derives for every token row (13 of its 30 VALU operations per
          // element); the same bf16 values, so the output is bit-identical
          const bf16_t* tb = p1 + b * (int64_t)2 * d;
          float sc[8], sh[8];
          unpack8(*reinterpret_cast<const uint4*>(tb + c * 8), sc);
          unpack8(*reinterpret_cast<const uint4*>(tb + d + c * 8), sh);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float ln = rbf((v[j] - mean) * rstd);           // F.layer_norm -> bf16
            y[j] = rbf(ln * sc[j]) + sh[j];                       // x *= 1+scale ; x += shift
          }
        } else {
          // head: modulation fp32 [2,d] + e bf16 [B,d] -> fp32; x bf16 updated in place twice
          const float* hm = reinterpret_cast<const float*>(p0);
          float ev[8];
          unpack8(*reinterpret_cast<const uint4*>(p1 + b * (int64_t)d + c * 8), ev);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float ln = rbf((v[j] - mean) * rstd);
            const float sc = 1.0f + (hm[(int64_t)scale_idx * d + c * 8 + j] + ev[j]);
            const float sh = hm[(int64_t)shift_idx * d + c * 8 + j] + ev[j];
            y[j] = rbf(ln * sc) + sh;
          }
        }
        const uint4 pk = pack8(y);
        *reinterpret_cast<uint4*>(o + c * 8) = pk;
        if (AMAX) {
          const uint32_t ww[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) am = max(am, max((ww[j] << 16) & 0x7fffffffu, ww[j] & 0x7fff0000u));
        }
      }
    }
    if (AMAX) {
#pragma unroll
      for (int o2 = 32; o2 > 0; o2 >>= 1) am = max(am, (uint32_t)__shfl_xor((int)am, o2, 64));
      // one atomic per ROW on one address would serialise in the L2 (151,200 of them per call at 14B-720p: run 08 measured the fold's
      // gain eaten); the running maximum is read first -- after the first few rows almost no row raises it
      if (lane == 0 && am != 0) {
        unsigned int* dst = amax + (int64_t)((uint32_t)row / (uint32_t)rows_per_amax) * 64 + 1;
        if (am > __atomic_load_n(dst, __ATOMIC_RELAXED)) atomicMax(dst, am);
      }
      am = 0;
    }
    if (!more) break;
    for (int i = 0; i < NCH; ++i) raw[i] = nxt[i];
    row = nrow;
  }
}

// ------------------------------------------------------------------------------------------------
// gated residual  x = bf16(x + y*gate)   (addcmul_: fp32 internally, one rounding)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gated_residual_kernel(
    bf16_t* __restrict__ x, const bf16_t* __restrict__ y, const bf16_t* __restrict__ mod,
    const bf16_t* __restrict__ e, int n_mod, int gate_idx, int64_t rows, int64_t rows_per_batch,
    int d) {
  const int nchunk = d >> 3;
  const int64_t total = rows * nchunk;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = idx / nchunk;
    const int c = (int)(idx - row * nchunk);
    float xv[8], yv[8], g[8];
    unpack8(*reinterpret_cast<const uint4*>(x + row * d + c * 8), xv);
    unpack8(*reinterpret_cast<const uint4*>(y + row * d + c * 8), yv);
    if (gate_idx >= 0) {
      const int64_t b = (int64_t)((uint32_t)row / (uint32_t)rows_per_batch);  // 32-bit: a 64-bit division costs ~100 VALU per lane
      float m[8], ev[8];
      unpack8(*reinterpret_cast<const uint4*>(mod + (int64_t)gate_idx * d + c * 8), m);
      unpack8(*reinterpret_cast<const uint4*>(e + (b * n_mod + gate_idx) * (int64_t)d + c * 8), ev);
#pragma unroll
      for (int j = 0; j < 8; ++j) g[j] = rbf(wan_add_f32(m[j], ev[j]));   // (a plain add: common.h)
#pragma unroll
      for (int j = 0; j < 8; ++j) xv[j] = xv[j] + yv[j] * g[j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) xv[j] = xv[j] + yv[j];
    }
    *reinterpret_cast<uint4*>(x + row * d + c * 8) = pack8(xv);
  }
}

// ------------------------------------------------------------------------------------------------
// normalized attention guidance on the two cross-attention results -- text_cross_attention, model.py:276-293
// ------------------------------------------------------------------------------------------------
// One wave per token row, two passes over the row pair (the second read comes out of L2: 4 * d bytes per wave):
//   pass 1  g = bf16(bf16(x_neg * (1 - s)) + s * x_pos)   (mul_ then add_(alpha): two roundings, :278-279), L1 norms of x_pos and g
//   row     n+ , ng rounded to bf16 (torch.norm returns the input dtype), ratio = bf16(ng / n+) with nan -> 10 and inf -> the
//           largest bf16 (nan_to_num, :285), factor = bf16(bf16(bf16(1 / bf16(ng + 1e-7)) * n+) * tau)   (:286, a rounding per op)
//   pass 2  g' = ratio > tau ? bf16(g * factor) : g;  out = bf16(bf16(g' * alpha) + bf16(x_pos * (1 - alpha)))   (:287-291)
// The scalars arrive the way torch's CPU kernels see them (probed on the torch of this image): mul_ by a Python number keeps it in
// fp32, add_(alpha=) and the comparison round it to the tensor's dtype first -- hence s_add / tau_cmp beside tau_mul.
// out may alias x_pos or x_neg (a lane reads its chunk of both rows before it writes it; pass 1 ends in a wave reduction).
__global__ __launch_bounds__(256) void nag_combine_kernel(const bf16_t* x_pos, const bf16_t* x_neg, bf16_t* out, int64_t rows,
                                                         int d, float one_minus_s, float s_add, float tau_cmp, float tau_mul,
                                                         float one_minus_alpha, float alpha) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nchunk = d >> 3;
  const bf16_t* xp = x_pos + row * d;
  const bf16_t* xn = x_neg + row * d;
  bf16_t* o = out + row * d;
  float sp = 0.f, sg = 0.f;
  for (int c = lane; c < nchunk; c += 64) {
    float p[8], n[8];
    unpack8(*reinterpret_cast<const uint4*>(xp + c * 8), p);
    unpack8(*reinterpret_cast<const uint4*>(xn + c * 8), n);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float g = rbf(__builtin_fmaf(s_add, p[j], rbf(n[j] * one_minus_s)));
      sp += __builtin_fabsf(p[j]);
      sg += __builtin_fabsf(g);
    }
  }
  const float np = rbf(wave_sum(sp)), ng = rbf(wave_sum(sg));
  float ratio = rbf(ng / np);
  if (ratio != ratio) ratio = 10.f;
  if (ratio > 3.3895313892515355e38f) ratio = 3.3895313892515355e38f;
  const float factor = rbf(rbf(rbf(1.0f / rbf(ng + 1e-7f)) * np) * tau_mul);
  const bool clip = ratio > tau_cmp;
  for (int c = lane; c < nchunk; c += 64) {
    float p[8], n[8];
    unpack8(*reinterpret_cast<const uint4*>(xp + c * 8), p);
    unpack8(*reinterpret_cast<const uint4*>(xn + c * 8), n);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float g = rbf(__builtin_fmaf(s_add, p[j], rbf(n[j] * one_minus_s)));
      if (clip) g = rbf(g * factor);
      n[j] = rbf(g * alpha) + rbf(p[j] * one_minus_alpha);
    }
    *reinterpret_cast<uint4*>(o + c * 8) = pack8(n);
  }
}

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
__global__ void sinusoid_kernel(const float* __restrict__ t, bf16_t* __restrict__ out, int n, int dim) {
  // cat([cos(t * 10000^(-i/half)), sin(...)])   model.py:32-42
  wan_hold_lds_word();   // (common.h: cosf / sinf hold packed instructions that read a register pair crosswise)
  const int half = dim / 2;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * half) return;
  const int r = idx / half, i = idx - r * half;
  const float freq = powf(10000.0f, -((float)i / (float)half));
  const float a = t[r] * freq;
  out[(int64_t)r * dim + i] = f2bf(cosf(a));
  out[(int64_t)r * dim + half + i] = f2bf(sinf(a));
}

__global__ void sinusoid_val_kernel(float tval, bf16_t* __restrict__ out, int dim) {
  wan_hold_lds_word();
  const int half = dim / 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= half) return;
  const float freq = powf(10000.0f, -((float)i / (float)half));
  const float a = tval * freq;
  out[i] = f2bf(cosf(a));
  out[half + i] = f2bf(sinf(a));
}

__global__ void act_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int64_t n, int act) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = bf2f(x[i]);
    float r = v;
    if (act == 1) r = v / (1.0f + expf(-v));  // SiLU
    else if (act == 2) r = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));  // GELU (erf), torch.nn.GELU() of MLPProj (model.py:875)
    y[i] = f2bf(r);
  }
}

// one wave per output column n (M <= 16 rows); W rows are K-contiguous -> 16-byte coalesced loads
__global__ __launch_bounds__(256) void gemv_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                   const bf16_t* __restrict__ bias, bf16_t* __restrict__ C,
                                                   int M, int N, int K) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float acc[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) acc[m] = 0.f;
  for (int c = lane; c < (K >> 3); c += 64) {
    float wv[8];
    unpack8(*reinterpret_cast<const uint4*>(W + (int64_t)n * K + c * 8), wv);
    for (int m = 0; m < M; ++m) {
      float av[8];
      unpack8(*reinterpret_cast<const uint4*>(A + (int64_t)m * K + c * 8), av);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[m] += av[j] * wv[j];
    }
  }
  for (int m = 0; m < M; ++m) {
    const float s = wave_sum(acc[m]);
    if (lane == 0) C[(int64_t)m * N + n] = f2bf(s + (bias ? bf2f(bias[n]) : 0.f));
  }
}

// out = sum_i coef[i]*in[i], fp32, float4-vectorised grid-stride
struct LinCombArgs {
  const float* in[6];
  float coef[6];
  int n_in;
};
__global__ __launch_bounds__(256) void lincomb_kernel(float* __restrict__ out, LinCombArgs a, int64_t n) {
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < a.n_in; ++j) {
      const float4 v = reinterpret_cast<const float4*>(a.in[j])[i];
      const float c = a.coef[j];
      if (j == 0) {
        acc.x = c * v.x; acc.y = c * v.y; acc.z = c * v.z; acc.w = c * v.w;
      } else {
        acc.x += c * v.x; acc.y += c * v.y; acc.z += c * v.z; acc.w += c * v.w;
      }
    }
    reinterpret_cast<float4*>(out)[i] = acc;
  }
  // tail
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    float acc = 0.f;
    for (int j = 0; j < a.n_in; ++j) acc = (j == 0) ? a.coef[j] * a.in[j][i] : acc + a.coef[j] * a.in[j][i];
    out[i] = acc;
  }
}

// noise_pred = uncond + g * (cond - uncond), same operation order as any2video.py:1722
__global__ __launch_bounds__(256) void cfg_combine_kernel(float* __restrict__ out, const float* __restrict__ cond,
                                                          const float* __restrict__ uncond, float g, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float u = uncond[i];
    out[i] = u + __fmul_rn(g, cond[i] - u);
  }
}

// vt[b, c, l] = v[b, l, c]; 64x64 tiles through LDS; zero-fill l in [L, ldv)
__global__ __launch_bounds__(256) void transpose_v_kernel(const bf16_t* __restrict__ v, bf16_t* __restrict__ vt,
                                                          int64_t L, int64_t ldv, int C) {
  __shared__ bf16_t tile[64][66];
  const int b = blockIdx.z;
  const int64_t l0 = (int64_t)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  const bf16_t* src = v + (int64_t)b * L * C;
  bf16_t* dst = vt + (int64_t)b * C * ldv;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, cc = i & 63;  // r: l offset, cc: c offset
    const int64_t l = l0 + r;
    const int c = c0 + cc;
    tile[r][cc] = (l < L && c < C) ? src[l * C + c] : (bf16_t)0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int cc = i >> 6, r = i & 63;
    const int64_t l = l0 + r;
    const int c = c0 + cc;
    if (c < C && l < ldv) dst[(int64_t)c * ldv + l] = tile[r][cc];
  }
}

// ------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------
static int pick_nch(int d) {
  const int need = ((d >> 3) + 63) / 64;
  const int opts[] = {1, 2, 3, 4, 6, 8, 10, 12, 16};
  for (int o : opts)
    if (o >= need) return o;
  return -1;
}

// grid of a persistent row kernel: every workgroup resident at once (occupancy x CUs, queried once per instantiation), never
// more workgroups than there are 4-row groups
template <typename K>
static unsigned persistent_blocks(K kernel, int64_t rows) {
  static const int resident = [&] {
    int per_cu = 0, dev = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    return per_cu * cus;
  }();
  const int64_t groups = (rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
  return (unsigned)(groups < resident ? groups : resident);
}

#define DISPATCH_NCH(nch, ...)                          \
  switch (nch) {                                        \
    case 1: { constexpr int NCH = 1; __VA_ARGS__; } break;   \
    case 2: { constexpr int NCH = 2; __VA_ARGS__; } break;   \
    case 3: { constexpr int NCH = 3; __VA_ARGS__; } break;   \
    case 4: { constexpr int NCH = 4; __VA_ARGS__; } break;   \
    case 6: { constexpr int NCH = 6; __VA_ARGS__; } break;   \
    case 8: { constexpr int NCH = 8; __VA_ARGS__; } break;   \
    case 10: { constexpr int NCH = 10; __VA_ARGS__; } break; \
    case 12: { constexpr int NCH = 12; __VA_ARGS__; } break; \
    default: { constexpr int NCH = 16; __VA_ARGS__; } break; \
  }

extern "C" int wan_rmsnorm_rope(wan_bf16* q, wan_bf16* k, const wan_bf16* wq, const wan_bf16* wk,
                                const float* cos, const float* sin, int64_t rows, int64_t L, int64_t pos0,
                                int d, float eps, void* stream) {
  return wan_rmsnorm_rope_scaled(q, k, wq, wk, cos, sin, rows, L, pos0, d, eps, 1.0f, stream);
}

extern "C" int wan_rmsnorm_rope_scaled(wan_bf16* q, wan_bf16* k, const wan_bf16* wq, const wan_bf16* wk,
                                       const float* cos, const float* sin, int64_t rows, int64_t L, int64_t pos0,
                                       int d, float eps, float q_scale, void* stream) {
  WAN_REQUIRE(q && wq, "wan_rmsnorm_rope: q/wq null");
  WAN_REQUIRE(d % 8 == 0 && d <= 8192, "wan_rmsnorm_rope: d=%d must be a multiple of 8 and <= 8192", d);
  WAN_REQUIRE((cos == nullptr) == (sin == nullptr), "wan_rmsnorm_rope: cos/sin must both be set or both null");
  WAN_REQUIRE(cos == nullptr || d % 128 == 0, "wan_rmsnorm_rope: RoPE needs d %% 128 == 0 (head_dim 128)");
  WAN_REQUIRE(k == nullptr || wk != nullptr, "wan_rmsnorm_rope: wk null");
  WAN_REQUIRE(rows < ((int64_t)1 << 31) && L > 0 && L < ((int64_t)1 << 31), "wan_rmsnorm_rope: rows / L must fit 31 bits");
  if (rows == 0) return 0;
  const int nch = pick_nch(d);
#define RMSROPE_LAUNCH(ROPE_, FULL_)                                                                                                       \
  do {                                                                                                                                    \
    unsigned gx = (unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);                                                               \
    if (P) { /* q and k rows share the resident slots (grid.y = 2): half the persistent width each */                                     \
      gx = persistent_blocks(rmsnorm_rope_kernel<NCH, P, ROPE_, FULL_>, rows);                                                            \
      if (k && gx > 1) gx = (gx + 1) / 2;                                                                                                 \
    }                                                                                                                                     \
    hipLaunchKernelGGL((rmsnorm_rope_kernel<NCH, P, ROPE_, FULL_>), dim3(gx, k ? 2 : 1), dim3(256), 0, as_stream(stream), q, k, wq, wk, cos, \
                       sin, rows, L, pos0, d, eps, q_scale);                                                                              \
  } while (0)
  DISPATCH_NCH(nch, {
    constexpr bool P = NCH >= 8;
    const bool full = d == NCH * 512;
    if (cos) { if (full) RMSROPE_LAUNCH(true, true); else RMSROPE_LAUNCH(true, false); }
    else { if (full) RMSROPE_LAUNCH(false, true); else RMSROPE_LAUNCH(false, false); }
  });
#undef RMSROPE_LAUNCH
  WAN_LAUNCH_CHECK();
  return 0;
}

// RMSNorm (+ RoPE) of ONE tensor written straight into the Ulysses exchange's send layout (see SCATTER at the kernel): x [rows, d] is read,
// pack receives what wan_rmsnorm_rope_scaled(x) followed by the head-chunk re-packs (wan_permute16 / wan_permute16_ex, csrc/dit.hip) would
// leave in it; x itself is not written.  d = world x heads_per_rank x 128; head_chunks <= heads_per_rank.
extern "C" int wan_rmsnorm_rope_pack(const wan_bf16* x, wan_bf16* pack, const wan_bf16* w, const float* cos, const float* sin, int64_t rows,
                                     int64_t L, int64_t pos0, int d, float eps, float scale, int world, int heads_per_rank, int head_chunks,
                                     void* stream) {
  WAN_REQUIRE(x && pack && w, "wan_rmsnorm_rope_pack: null pointer");
  WAN_REQUIRE(d % 128 == 0 && d <= 8192 && world >= 1 && heads_per_rank >= 1 && (int64_t)world * heads_per_rank * 128 == d,
              "wan_rmsnorm_rope_pack: d = %d must be world (%d) x heads per rank (%d) x 128", d, world, heads_per_rank);
  WAN_REQUIRE(head_chunks >= 1 && head_chunks <= heads_per_rank, "wan_rmsnorm_rope_pack: %d head chunks for %d heads", head_chunks, heads_per_rank);
  WAN_REQUIRE((cos == nullptr) == (sin == nullptr), "wan_rmsnorm_rope_pack: cos/sin must both be set or both null");
  WAN_REQUIRE(rows < ((int64_t)1 << 31) && L > 0 && L < ((int64_t)1 << 31) && rows * (int64_t)d / 8 < ((int64_t)1 << 32),
              "wan_rmsnorm_rope_pack: rows / L must fit 31 bits, rows x d / 8 32 bits");
  WAN_REQUIRE(x != pack, "wan_rmsnorm_rope_pack: the packed rows cannot overwrite their source");
  if (rows == 0) return 0;
  const int nch = pick_nch(d);
  bf16_t* xin = const_cast<bf16_t*>(x);
#define RMSPACK_LAUNCH(ROPE_, FULL_)                                                                                                        \
  do {                                                                                                                                      \
    unsigned gx = (unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);                                                                 \
    if (P) gx = persistent_blocks(rmsnorm_rope_kernel<NCH, P, ROPE_, FULL_, true>, rows);                                                    \
    hipLaunchKernelGGL((rmsnorm_rope_kernel<NCH, P, ROPE_, FULL_, true>), dim3(gx, 1), dim3(256), 0, as_stream(stream), xin, (bf16_t*)nullptr, w, \
                       (const bf16_t*)nullptr, cos, sin, rows, L, pos0, d, eps, scale, pack, world, heads_per_rank, head_chunks);            \
  } while (0)
  DISPATCH_NCH(nch, {
    constexpr bool P = NCH >= 8;
    const bool full = d == NCH * 512;
    if (cos) { if (full) RMSPACK_LAUNCH(true, true); else RMSPACK_LAUNCH(true, false); }
    else { if (full) RMSPACK_LAUNCH(false, true); else RMSPACK_LAUNCH(false, false); }
  });
#undef RMSPACK_LAUNCH
  WAN_LAUNCH_CHECK();
  return 0;
}

// table[b][0][c] = rbf(1 + rbf(mod[scale][c] + e[b][scale][c])), table[b][1][c] = rbf(mod[shift][c] + e[b][shift][c]): the bf16
// operations of model.py:632-638 on the [6, d] modulation vectors, once per (layer, batch) instead of once per token row
__global__ __launch_bounds__(256) void mod_table_kernel(const bf16_t* __restrict__ mod, const bf16_t* __restrict__ e, bf16_t* __restrict__ tab,
                                                        int n_mod, int shift_idx, int scale_idx, int d, int nb) {
  const int i = blockIdx.x * 256 + threadIdx.x;  // 8-element chunk
  const int nchunk = d >> 3;
  if (i >= nb * nchunk) return;
  const int b = i / nchunk, c = i - b * nchunk;
  float msh[8], msc[8], esh[8], esc[8], sc[8], sh[8];
  unpack8(*reinterpret_cast<const uint4*>(mod + (int64_t)shift_idx * d + c * 8), msh);
  unpack8(*reinterpret_cast<const uint4*>(mod + (int64_t)scale_idx * d + c * 8), msc);
  const bf16_t* eb = e + (int64_t)b * n_mod * d;
  unpack8(*reinterpret_cast<const uint4*>(eb + (int64_t)shift_idx * d + c * 8), esh);
  unpack8(*reinterpret_cast<const uint4*>(eb + (int64_t)scale_idx * d + c * 8), esc);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = rbf(1.0f + rbf(wan_add_f32(msc[j], esc[j])));   // (plain adds: common.h)
    sh[j] = rbf(wan_add_f32(msh[j], esh[j]));
  }
  *reinterpret_cast<uint4*>(tab + (int64_t)b * 2 * d + c * 8) = pack8(sc);
  *reinterpret_cast<uint4*>(tab + (int64_t)b * 2 * d + d + c * 8) = pack8(sh);
}
// Library-owned scratch for those tables: rings of 16 slots x 1 MiB per (device, stream) (21 frames x 2 streams x 2 x 5120 x 2 B =
// 860 KB is the largest Wan case: per-frame timesteps at 14B); a call whose table does not fit keeps the per-row form.  Calls on one
// stream are serialised and each table is dead when its LN kernel has run, 16 calls earlier at the latest.
constexpr size_t MODTAB_SLOT = (size_t)1 << 20;
constexpr int MODTAB_NSLOT = 16;
static bf16_t* modtab_slot(size_t need_bytes, hipStream_t stream) {
  return reinterpret_cast<bf16_t*>(wan_scratch_ring_slot(/*tag=*/1, MODTAB_SLOT, MODTAB_NSLOT, need_bytes, stream));
}

namespace {
struct ScratchRing {
  int tag, device;
  hipStream_t stream;
  char* base;
  size_t slot_bytes;
  int nslot;
  unsigned next;
};
std::mutex g_ring_mutex;
ScratchRing g_rings[64];
int g_nrings = 0;
}  // namespace
void* wan_scratch_ring_slot(int tag, size_t slot_bytes, int nslot, size_t need_bytes, hipStream_t stream) {
  if (need_bytes > slot_bytes) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  std::lock_guard<std::mutex> lock(g_ring_mutex);
  ScratchRing* r = nullptr;
  for (int i = 0; i < g_nrings; ++i)
    if (g_rings[i].tag == tag && g_rings[i].device == dev && g_rings[i].stream == stream) r = &g_rings[i];
  if (r == nullptr) {
    if (g_nrings == 64) return nullptr;
    char* base = nullptr;
    if (hipMalloc((void**)&base, slot_bytes * (size_t)nslot) != hipSuccess) {  // on the CURRENT device: the one the caller launches on
      (void)hipGetLastError();
      return nullptr;
    }
    r = &g_rings[g_nrings++];
    *r = ScratchRing{tag, dev, stream, base, slot_bytes, nslot, 0u};
  }
  return r->base + (size_t)(r->next++ % (unsigned)r->nslot) * r->slot_bytes;
}

static int ln_modulate_impl(const wan_bf16* x, wan_bf16* out, const wan_bf16* mod, const wan_bf16* e, int n_mod,
                            int shift_idx, int scale_idx, int64_t rows, int64_t rows_per_batch, int d, float eps,
                            unsigned int* amax, int64_t rows_per_amax, void* stream) {
  WAN_REQUIRE(x && out && mod && e, "wan_ln_modulate: null pointer");
  WAN_REQUIRE(d % 8 == 0 && d <= 8192, "wan_ln_modulate: d=%d must be a multiple of 8 and <= 8192", d);
  WAN_REQUIRE(shift_idx >= 0 && shift_idx < n_mod && scale_idx >= 0 && scale_idx < n_mod, "wan_ln_modulate: bad idx");
  WAN_REQUIRE(rows < ((int64_t)1 << 31) && rows_per_batch > 0 && rows_per_batch < ((int64_t)1 << 31),
              "wan_ln_modulate: rows / rows_per_batch must fit 31 bits");
  WAN_REQUIRE(amax == nullptr || (rows_per_amax > 0 && rows_per_amax < ((int64_t)1 << 31)), "wan_ln_modulate_amax: rows_per_slot must fit 31 bits");
  if (rows == 0) return 0;
  const int nch = pick_nch(d);
  const int64_t nb = (rows + rows_per_batch - 1) / rows_per_batch;
  const dim3 grid((unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK));
  // many rows per batch: derive the two modulation vectors once per batch (a ~2 us kernel) and let the row kernel read them
  bf16_t* tab = (rows >= 64 * nb) ? modtab_slot((size_t)nb * 2 * d * 2, as_stream(stream)) : nullptr;
  if (tab != nullptr) {
    const int chunks = (int)nb * (d >> 3);
    hipLaunchKernelGGL(mod_table_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, as_stream(stream), mod, e, tab, n_mod, shift_idx,
                       scale_idx, d, (int)nb);
    WAN_LAUNCH_CHECK();
    if (amax != nullptr) {
      DISPATCH_NCH(nch, hipLaunchKernelGGL((layernorm_kernel<NCH, 3, false, true>), grid, dim3(256), 0, as_stream(stream), x, out, (const void*)nullptr,
                                           (const bf16_t*)tab, n_mod, shift_idx, scale_idx, rows, rows_per_batch, d, eps, amax, rows_per_amax));
    } else {
      DISPATCH_NCH(nch, hipLaunchKernelGGL((layernorm_kernel<NCH, 3>), grid, dim3(256), 0, as_stream(stream), x, out, (const void*)nullptr,
                                           (const bf16_t*)tab, n_mod, shift_idx, scale_idx, rows, rows_per_batch, d, eps, (unsigned int*)nullptr, (int64_t)1));
    }
    WAN_LAUNCH_CHECK();
    return 0;
  }
  if (amax != nullptr) {
    DISPATCH_NCH(nch, hipLaunchKernelGGL((layernorm_kernel<NCH, 0, false, true>), grid, dim3(256), 0, as_stream(stream), x, out, (const void*)mod, e, n_mod,
                                         shift_idx, scale_idx, rows, rows_per_batch, d, eps, amax, rows_per_amax));
  } else {
    DISPATCH_NCH(nch, hipLaunchKernelGGL((layernorm_kernel<NCH, 0>), grid, dim3(256), 0, as_stream(stream), x, out, (const void*)mod, e, n_mod,
                                         shift_idx, scale_idx, rows, rows_per_batch, d, eps, (unsigned int*)nullptr, (int64_t)1));
  }
  WAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int wan_ln_modulate(const wan_bf16* x, wan_bf16* out, const wan_bf16* mod, const wan_bf16* e, int n_mod,
                               int shift_idx, int scale_idx, int64_t rows, int64_t rows_per_batch, int d, float eps,
                               void* stream) {
  return ln_modulate_impl(x, out, mod, e, n_mod, shift_idx, scale_idx, rows, rows_per_batch, d, eps, nullptr, 1, stream);
}

// wan_ln_modulate that also accumulates max |out| per `rows_per_slot` rows into word 1 of consecutive 64-word quantisation slots at
// `amax_ws` (float bits, atomicMax; the caller zeroed the words): what wan_fp8_quantize_pre(..., amax_word 1) reads (round 5)
extern "C" int wan_ln_modulate_amax(const wan_bf16* x, wan_bf16* out, const wan_bf16* mod, const wan_bf16* e, int n_mod,
                                    int shift_idx, int scale_idx, int64_t rows, int64_t rows_per_batch, int d, float eps,
                                    float* amax_ws, int64_t rows_per_slot, void* stream) {
  WAN_REQUIRE(amax_ws != nullptr, "wan_ln_modulate_amax: null amax words");
  return ln_modulate_impl(x, out, mod, e, n_mod, shift_idx, scale_idx, rows, rows_per_batch, d, eps, reinterpret_cast<unsigned int*>(amax_ws),
                          rows_per_slot, stream);
}

static int ln_affine_impl(const wan_bf16* x, wan_bf16* out, const wan_bf16* w, const wan_bf16* b, int64_t rows, int d, float eps,
                          unsigned int* amax, int64_t rows_per_amax, void* stream) {
  WAN_REQUIRE(x && out && w && b, "wan_ln_affine: null pointer");
  WAN_REQUIRE(d % 8 == 0 && d <= 8192, "wan_ln_affine: d=%d must be a multiple of 8 and <= 8192", d);
  WAN_REQUIRE(amax == nullptr || (rows < ((int64_t)1 << 31) && rows_per_amax > 0 && rows_per_amax < ((int64_t)1 << 31)),
              "wan_ln_affine_amax: rows / rows_per_slot must fit 31 bits");
  if (rows == 0) return 0;
  const int nch = pick_nch(d);
  const dim3 grid((unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK));
  if (amax != nullptr) {
    DISPATCH_NCH(nch, hipLaunchKernelGGL((layernorm_kernel<NCH, 1, false, true>), grid, dim3(256), 0, as_stream(stream), x, out, (const void*)w, b, 0, 0, 0,
                                         rows, rows, d, eps, amax, rows_per_amax));
  } else {
    DISPATCH_NCH(nch, hipLaunchKernelGGL((layernorm_kernel<NCH, 1>), grid, dim3(256), 0, as_stream(stream), x, out, (const void*)w, b, 0, 0, 0,
                                         rows, rows, d, eps, (unsigned int*)nullptr, (int64_t)1));
  }
  WAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int wan_ln_affine(const wan_bf16* x, wan_bf16* out, const wan_bf16* w, const wan_bf16* b, int64_t rows,
                             int d, float eps, void* stream) {
  return ln_affine_impl(x, out, w, b, rows, d, eps, nullptr, 1, stream);
}

extern "C" int wan_ln_affine_amax(const wan_bf16* x, wan_bf16* out, const wan_bf16* w, const wan_bf16* b, int64_t rows, int d, float eps,
                                  float* amax_ws, int64_t rows_per_slot, void* stream) {
  WAN_REQUIRE(amax_ws != nullptr, "wan_ln_affine_amax: null amax words");
  return ln_affine_impl(x, out, w, b, rows, d, eps, reinterpret_cast<unsigned int*>(amax_ws), rows_per_slot, stream);
}

// head LN+modulate (fp32 modulation): exposed to head.hip
int wan_ln_modulate_head(const bf16_t* x, bf16_t* out, const float* hmod, const bf16_t* e, int64_t rows,
                         int64_t rows_per_batch, int d, float eps, void* stream) {
  if (rows == 0) return 0;
  const int nch = pick_nch(d);
  DISPATCH_NCH(nch, hipLaunchKernelGGL((layernorm_kernel<NCH, 2>), dim3((unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)),
                                       dim3(256), 0, as_stream(stream), x, out, (const void*)hmod, e, 2, 0, 1, rows, rows_per_batch, d, eps,
                                       (unsigned int*)nullptr, (int64_t)1));
  WAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int wan_gated_residual(wan_bf16* x, const wan_bf16* y, const wan_bf16* mod, const wan_bf16* e, int n_mod,
                                  int gate_idx, int64_t rows, int64_t rows_per_batch, int d, void* stream) {
  WAN_REQUIRE(x && y, "wan_gated_residual: null pointer");
  WAN_REQUIRE(d % 8 == 0, "wan_gated_residual: d %% 8 != 0");
  WAN_REQUIRE(gate_idx < 0 || (mod && e && gate_idx < n_mod), "wan_gated_residual: gate needs mod/e");
  if (rows == 0) return 0;
  const int64_t total = rows * (d >> 3);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(gated_residual_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), x, y, mod, e, n_mod,
                     gate_idx, rows, rows_per_batch, d);
  WAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int wan_nag_combine(const wan_bf16* x_pos, const wan_bf16* x_neg, wan_bf16* out, int64_t rows, int d, float nag_scale,
                               float nag_tau, float nag_alpha, void* stream) {
  WAN_REQUIRE(x_pos && x_neg && out, "wan_nag_combine: null pointer");
  WAN_REQUIRE(d % 8 == 0 && d > 0, "wan_nag_combine: d %% 8 != 0");
  if (rows == 0) return 0;
  auto to_bf16 = [](float f) {  // round-to-nearest-even, as torch casts a Python scalar to a bf16 tensor's dtype
    uint32_t u;
    memcpy(&u, &f, 4);
    u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
    memcpy(&f, &u, 4);
    return f;
  };
  const unsigned blocks = (unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
  hipLaunchKernelGGL(nag_combine_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), x_pos, x_neg, out, rows, d,
                     (float)(1.0 - (double)nag_scale), to_bf16(nag_scale), to_bf16(nag_tau), nag_tau,
                     (float)(1.0 - (double)nag_alpha), nag_alpha);
  WAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int wan_sinusoid(const float* t, wan_bf16* out, int n, int dim, void* stream) {
  WAN_REQUIRE(t && out && dim % 2 == 0, "wan_sinusoid: bad args");
  const int total = n * dim / 2;
  hipLaunchKernelGGL(sinusoid_kernel, dim3((total + 255) / 256), dim3(256), 0, as_stream(stream), t, out, n, dim);
  WAN_LAUNCH_CHECK();
  return 0;
}

__global__ void set_f32_kernel(float* p, float v) { *p = v; }
int wan_set_f32(float* p, float v, void* stream) {   // (internal: the timestep of a replayed forward, csrc/dit.hip wan_dit_forward_graph)
  hipLaunchKernelGGL(set_f32_kernel, dim3(1), dim3(1), 0, as_stream(stream), p, v);
  WAN_LAUNCH_CHECK();
  return 0;
}
int wan_sinusoid_val(float t, bf16_t* out, int dim, void* stream) {
  hipLaunchKernelGGL(sinusoid_val_kernel, dim3((dim / 2 + 255) / 256), dim3(256), 0, as_stream(stream), t, out, dim);
  WAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int wan_act_bf16(const wan_bf16* x, wan_bf16* y, int64_t n, int act, void* stream) {
  WAN_REQUIRE(x && y, "wan_act_bf16: null pointer");
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (blocks == 0) return 0;
  hipLaunchKernelGGL(act_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), x, y, n, act);
  WAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int wan_gemv_bf16(const wan_bf16* A, const wan_bf16* W, const wan_bf16* bias, wan_bf16* C, int M, int N,
                             int K, void* stream) {
  WAN_REQUIRE(A && W && C, "wan_gemv_bf16: null pointer");
  WAN_REQUIRE(M >= 1 && M <= 16 && K % 8 == 0, "wan_gemv_bf16: need 1<=M<=16 and K %% 8 == 0 (M=%d K=%d)", M, K);
  hipLaunchKernelGGL(gemv_kernel, dim3((N + 3) / 4), dim3(256), 0, as_stream(stream), A, W, bias, C, M, N, K);
  WAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int wan_lincomb(float* out, int n_in, const float* const* in, const float* coef, int64_t n, void* stream) {
  WAN_REQUIRE(out && in && coef && n_in >= 1 && n_in <= 6, "wan_lincomb: bad args (n_in=%d)", n_in);
  WAN_REQUIRE((((uintptr_t)out) & 15) == 0, "wan_lincomb: out must be 16-byte aligned");
  LinCombArgs a;
  a.n_in = n_in;
  for (int i = 0; i < n_in; ++i) {
    WAN_REQUIRE(in[i] && (((uintptr_t)in[i]) & 15) == 0, "wan_lincomb: in[%d] null or not 16-byte aligned", i);
    a.in[i] = in[i];
    a.coef[i] = coef[i];
  }
  if (n == 0) return 0;
  int blocks = (int)(((n >> 2) + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(lincomb_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), out, a, n);
  WAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int wan_cfg_combine(float* out, const float* cond, const float* uncond, float guide_scale, int64_t n,
                               void* stream) {
  WAN_REQUIRE(out && cond && uncond, "wan_cfg_combine: null pointer");
  if (n == 0) return 0;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(cfg_combine_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), out, cond, uncond, guide_scale, n);
  WAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int wan_transpose_v(const wan_bf16* v, wan_bf16* vt, int B, int64_t L, int64_t ldv, int C, void* stream) {
  WAN_REQUIRE(v && vt && ldv >= L, "wan_transpose_v: bad args");
  dim3 grid((unsigned)((ldv + 63) / 64), (unsigned)((C + 63) / 64), (unsigned)B);
  hipLaunchKernelGGL(transpose_v_kernel, grid, dim3(256), 0, as_stream(stream), v, vt, L, ldv, C);
  WAN_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// src [A][B][n16 x 16 B] -> dst [B][A][n16 x 16 B]: the head-group-major re-packs around the Ulysses all-to-alls (dit.hip)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void permute16_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t A, int64_t B,
                                                        int64_t n16) {
  const int64_t total = A * B * n16, stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {  // i walks dst: coalesced stores, piece-wise coalesced loads
    const int64_t c = i % n16, ba = i / n16;
    const int64_t a = ba % A, b = ba / A;
    dst[i] = src[(a * B + b) * n16 + c];
  }
}
extern "C" int wan_permute16(const void* src, void* dst, int64_t A, int64_t B, int64_t bytes, void* stream) {
  WAN_REQUIRE(src && dst && src != dst, "wan_permute16: null or aliased pointers");
  WAN_REQUIRE(A >= 0 && B >= 0 && bytes >= 0 && bytes % 16 == 0, "wan_permute16: bytes=%lld must be a multiple of 16", (long long)bytes);
  WAN_REQUIRE((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "wan_permute16: pointers must be 16-byte aligned");
  const int64_t total = A * B * (bytes / 16);
  if (total == 0) return 0;
  const int64_t blocks = std::min<int64_t>((total + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(permute16_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), (const uint4*)src, (uint4*)dst, A, B, bytes / 16);
  WAN_LAUNCH_CHECK();
  return 0;
}

__global__ __launch_bounds__(256) void permute16_ex_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t A, int64_t B, int64_t n16,
                                                           int64_t sa, int64_t sb, int64_t da, int64_t db) {
  const int64_t total = A * B * n16, stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {  // (b, a, piece) with the piece fastest: 16-byte accesses, coalesced within a piece
    const int64_t c = i % n16, ba = i / n16;
    const int64_t a = ba % A, b = ba / A;
    dst[b * db + a * da + c] = src[a * sa + b * sb + c];
  }
}
extern "C" int wan_permute16_ex(const void* src, void* dst, int64_t A, int64_t B, int64_t bytes, int64_t src_a_pitch, int64_t src_b_pitch,
                                int64_t dst_a_pitch, int64_t dst_b_pitch, void* stream) {
  WAN_REQUIRE(src && dst && src != dst, "wan_permute16_ex: null or aliased pointers");
  WAN_REQUIRE(A >= 0 && B >= 0 && bytes >= 0 && ((bytes | src_a_pitch | src_b_pitch | dst_a_pitch | dst_b_pitch) & 15) == 0,
              "wan_permute16_ex: bytes=%lld and the pitches must be multiples of 16", (long long)bytes);
  WAN_REQUIRE((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "wan_permute16_ex: pointers must be 16-byte aligned");
  const int64_t total = A * B * (bytes / 16);
  if (total == 0) return 0;
  const int64_t blocks = std::min<int64_t>((total + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(permute16_ex_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), (const uint4*)src, (uint4*)dst, A, B, bytes / 16,
                     src_a_pitch / 16, src_b_pitch / 16, dst_a_pitch / 16, dst_b_pitch / 16);
  WAN_LAUNCH_CHECK();
  return 0;
}
