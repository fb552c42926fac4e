// Shared pieces of the "4 waves x 64 q rows" attention kernel (attention_w64q.hip): fragment
// types, the inline-asm MFMAs that pin operands to the VGPR / accumulator files, and the per-wave LDS-DMA stream.
#pragma once
#include "common.h"

namespace {


typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 hw_bf16x2;
typedef __attribute__((ext_vector_type(2))) float hw_f32x2;
typedef __attribute__((address_space(3))) const char lds_cchar;
typedef __attribute__((address_space(3))) const mfma_bf16x8 lds_frag;

constexpr int KVBLK = 64;
constexpr int IMG = 16384;  // bytes per K or V^T image
constexpr int NST = 3;  // LDS ring depth

__device__ __forceinline__ uint32_t cvt_pk(float lo, float hi) {
  hw_f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hw_bf16x2));
}
__device__ __forceinline__ float xhalf_max(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

#define SB() __builtin_amdgcn_sched_barrier(0)
// O^T += V^T P^T: accumulators pinned to the accumulator file ("+a"); see mfma_qk below for why these are asm
#define MF(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))

// S^T = K Q^T MFMAs as inline asm: once a kernel needs the accumulator file hipcc gives EVERY builtin MFMA an AGPR
// destination, so S (consumed by the softmax VALU) would be copied out with 64 v_accvgpr_read per tile and Q (parked in
// AGPRs by the allocator) copied in before every use.  The constraints pin the files: S in arch VGPRs, Q in AGPRs, K
// fragments in VGPRs (straight from ds_read_b128).  hipcc does not pad hazards of an asm statement: the first VALU
// reader of an S tile is placed >= 3 independent MFMAs (> 96 cycles) after the tile's last MFMA by tile_w64's pinned
// order (the XDL write -> VALU read hazard of a 16-pass MFMA is 18 wait states); P is packed a whole slot before the
// PV MFMAs read it; O is read by VALU only in the rare rescale branch (a whole slot after its last MFMA) and in the
// epilogue (behind explicit s_nops).  The order is ALWAYS pinned with sched_barrier(0): hipcc must not move a VALU
// reader next to an asm MFMA it knows nothing about.
__device__ __forceinline__ void mfma_qk0(f32x16& d, const mfma_bf16x8& k, const mfma_bf16x8& q) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(k), "a"(q));
}
__device__ __forceinline__ void mfma_qk(f32x16& d, const mfma_bf16x8& k, const mfma_bf16x8& q) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(k), "a"(q));
}

// LDS-DMA stream of one wave: walks the K / V^T tiles of all kv segments in order, one 16-B piece per lane per
// call.  piece s = i*256 + tid (i = 0..3) of an image.  K: row 16 i + (tid>>4) (bits 2<->3 swapped inside the low
// nibble), 16-B chunk (tid&15) ^ (row&15); V^T: row 32 i + (tid>>3), chunk (tid&7) ^ ((row>>1)&7).
// Issued as `buffer_load_dwordx4 v_off, s[desc], 0 offen lds`: the per-lane offsets of the 8 pieces are loop-invariant
// VGPRs, the tile position lives in the descriptor base (3 SALU per tile), and the K descriptor's num_records ends at
// the segment's last valid row, so rows of a ragged tail tile read as zeros (hardware range check) -- no per-piece
// address arithmetic, no clamping, no branch.
struct Dma {
  const char* k;   // K rows of the next tile to fetch (uniform)
  const char* v;   // V^T columns of the next tile
  const char* kseg0;  // base of the current segment, and the segment strides (bytes)
  const char* vseg0;
  int64_t kseg, vseg;
  int tt, tps, left;  // tile inside the segment, tiles per segment, tiles not yet fetched
  int seg, skip;      // current segment; segment to leave out (-1: none) -- sequence parallelism attends the rank's own
                      // segment in a first launch while the others are still arriving
  uint32_t klen, klen0;    // valid K bytes from d.k to the end of the segment ((rows-1)*pitch + 256); at a segment start
  uint32_t rs2, ldv2;      // K / V^T row pitch in bytes
  uint32_t kofs[4], vofs[4];
  int wave;
};
// The stream's state is wave-uniform and must LIVE in scalar registers: the bases arrive through integer divisions of the
// workgroup id (VALU sequences), and a loop-carried value that starts in a VGPR stays there -- every step then runs on the vector
// ALU beside the MFMAs and every DMA descriptor is rebuilt with v_readfirstlane.  These put the initial values into SGPRs.
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ int64_t uni(int64_t x) {
  const uint64_t u = (uint64_t)x;
  return (int64_t)(((uint64_t)uni((uint32_t)(u >> 32)) << 32) | uni((uint32_t)u));
}
__device__ __forceinline__ const char* uni(const char* p) { return (const char*)uni((int64_t)(uintptr_t)p); }
__device__ __forceinline__ void dma_init(Dma& d, const void* kbase, const void* vbase, int64_t kseg_bytes, int64_t vseg_bytes,
                                         int Lk, int nseg, uint32_t rs2, uint32_t ldv2, int tid, int wave, int skip = -1, bool k_rows_16x16 = false) {
  d.seg = uni((skip == 0) ? 1 : 0);
  d.skip = uni(skip);
  d.k = d.kseg0 = uni(reinterpret_cast<const char*>(kbase) + (int64_t)d.seg * kseg_bytes);
  d.v = d.vseg0 = uni(reinterpret_cast<const char*>(vbase) + (int64_t)d.seg * vseg_bytes);
  d.kseg = uni(kseg_bytes);
  d.vseg = uni(vseg_bytes);
  d.tps = uni((Lk + KVBLK - 1) / KVBLK);
  d.tt = uni(0); d.left = uni(d.tps * (nseg - (skip >= 0 ? 1 : 0)));
  d.klen = d.klen0 = uni((uint32_t)(Lk - 1) * rs2 + 256u);
  d.rs2 = uni(rs2);
  d.ldv2 = uni(ldv2);
  const uint32_t kr0 = (uint32_t)(tid >> 4);
  const uint32_t krow0 = (kr0 & 3u) | ((kr0 & 4u) << 1) | ((kr0 & 8u) >> 1);
  const uint32_t kcol = (uint32_t)(((tid & 15) ^ (int)kr0) << 4);
  const uint32_t vofs0 = (uint32_t)(tid >> 3) * ldv2 + (uint32_t)(((tid & 7) ^ ((tid >> 4) & 7)) << 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // LDS row 16 i + m (m = tid >> 4) of the K image <- kv row of the tile: the 32x32x16 kernel swaps bits 2 <-> 3 of m; the
    // 16x16x32 kernel (attention_w16n.hip) wants 32 (i >> 1) + 8 (m >> 2) + 4 (i & 1) + (m & 3)
    const uint32_t src = k_rows_16x16 ? 32u * (i >> 1) + 8u * (kr0 >> 2) + 4u * (i & 1) + (kr0 & 3u) : krow0 + 16u * i;
    d.kofs[i] = src * rs2 + kcol;
    d.vofs[i] = vofs0 + 32u * i * ldv2;
  }
  d.wave = wave;
}
// piece I = 0..3: K, 4..7: V^T, into ring stage ST (I is a template argument or an unrolled loop variable).
// Issued as INLINE ASM: for the builtin form hipcc's waitcnt pass puts `s_waitcnt vmcnt(0)` in front of every ds_read it
// cannot prove disjoint from a pending LDS-DMA (it cannot tell the ring stages apart once a per-lane address is involved) --
// measured: four such waits per tile in the bounded loop, ~2,500 cycles each, the whole DMA latency serialised.  The asm form
// is invisible to that pass; completion is counted by hand (one vmcnt(0) + barrier at the top of each tile).
typedef uint32_t dma_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma_issue(uint32_t voff, const char* base, uint32_t num_records, uint32_t lds_addr) {
  const uint64_t b = (uint64_t)base;
  dma_u4 r;
  r[0] = (uint32_t)b;
  r[1] = (uint32_t)(b >> 32) & 0xffffu;  // stride 0
  r[2] = num_records;                    // bytes from base that may be read; beyond: zeros (range check)
  r[3] = 0x00020000u;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(r), "s"(lds_addr) : "memory");
}
template <int ST>
__device__ __forceinline__ void dma_piece_i(char* smem, const Dma& d, int I) {
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  if (I < 4) dma_issue(d.kofs[I], d.k, d.klen, lds0 + ST * IMG + d.wave * 1024 + I * 4096);
  else dma_issue(d.vofs[I - 4], d.v, 0xffffffffu, lds0 + NST * IMG + ST * IMG + d.wave * 1024 + (I - 4) * 4096);
}
template <int I, int ST>
__device__ __forceinline__ void dma_piece(char* smem, const Dma& d) {
  dma_piece_i<ST>(smem, d, I);
}
// step to the next tile; after the last tile the stream stays put (later fetches re-read it into a dead stage).
// Branch-free (scalar selects): a taken branch in the tile loop costs ~100 cycles of instruction fetch.  MULTI = false is the
// single-segment stream (every self- / cross-attention launch outside sequence parallelism): ~10 scalar instructions per tile
// instead of the ~60 the segment walk needs -- the compiler sinks the whole step behind the tile's last branch, i.e. into ONE MFMA
// gap, where the long form held the matrix pipe up for 60-90 cycles per tile (the "PV_a head" of the stamps).  Flags are 0 / 1
// integers, not bools: hipcc carried bool selects through a lane mask and a v_cndmask + v_readfirstlane round trip.
template <bool MULTI>
__device__ __forceinline__ void dma_advance(Dma& d) {
  const int adv = d.left > 1 ? 1 : 0;
  d.left -= adv;
  if (!MULTI) {
    const uint32_t kb = adv ? (uint32_t)KVBLK * d.rs2 : 0u;
    d.k += kb;
    d.v += adv ? KVBLK * 2 : 0;
    d.klen -= kb;
    return;
  }
  const int last = (d.tt + 1 == d.tps) ? 1 : 0;
  const int sw = adv & last;                                     // move to the next kv segment
  const int stp = adv & (last ^ 1);                              // next tile of the same segment
  const int hop2 = sw & ((d.seg + 1 == d.skip) ? 1 : 0);         // jump over the left-out segment
  d.tt = sw ? 0 : d.tt + stp;
  d.seg += sw ? 1 + hop2 : 0;
  d.kseg0 = sw ? d.kseg0 + (hop2 ? 2 * d.kseg : d.kseg) : d.kseg0;
  d.vseg0 = sw ? d.vseg0 + (hop2 ? 2 * d.vseg : d.vseg) : d.vseg0;
  const char* kstep = d.k + (stp ? (int64_t)KVBLK * d.rs2 : (int64_t)0);
  const char* vstep = d.v + (stp ? KVBLK * 2 : 0);
  d.k = sw ? d.kseg0 : kstep;
  d.v = sw ? d.vseg0 : vstep;
  d.klen = sw ? d.klen0 : d.klen - (stp ? (uint32_t)KVBLK * d.rs2 : 0u);
}
template <int ST, bool MULTI>
__device__ __forceinline__ void dma_tile(char* smem, Dma& d) {
#pragma unroll
  for (int I = 0; I < 8; ++I) dma_piece_i<ST>(smem, d, I);
  dma_advance<MULTI>(d);
}

}  // namespace
