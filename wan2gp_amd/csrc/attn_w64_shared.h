// Shared pieces of the "4 waves x 64 q rows" attention kernels (attention_w64.hip, attention_w64q.hip): fragment
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
constexpr int NST = 3;      // LDS ring depth

__device__ __forceinline__ uint32_t cvt_pk(float lo, float hi) {
  hw_f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hw_bf16x2));
}
__device__ __forceinline__ void glds16(const void* gsrc, void* ldst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)ldst, 16, 0, 0);
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
// nibble), 16-B chunk (tid&15) ^ (row&15); V^T: row 32 i + (tid>>3), chunk (tid&7) ^ ((row>>1)&7).  The swizzles do
// not depend on i, so the per-lane plan is three VGPRs (krow0, kcol, vofs0) and everything else is uniform.
struct Dma {
  const char* k;   // K rows of the next tile to fetch (uniform)
  const char* v;   // V^T columns of the next tile
  const char* k0;  // segment 0 bases and segment strides (bytes)
  const char* v0;
  int64_t kseg, vseg;
  int tt, seg, tps, left;  // tile inside the segment, segment, tiles per segment, tiles not yet fetched
  int tail_lim;            // last valid row of a segment's last tile (63 when Lk % 64 == 0)
  uint32_t rs2, ldv2;      // K / V^T row pitch in bytes
  uint32_t krow0, kcol, vofs0;
  int wave;
};
// piece I = 0..3: K, 4..7: V^T, into ring stage ST
template <int I, int ST>
__device__ __forceinline__ void dma_piece(char* smem, const Dma& d) {
  if (I < 4) {
    const uint32_t lim = (d.tt == d.tps - 1) ? (uint32_t)d.tail_lim : 63u;  // ragged tail: clamp rows to the last valid one
    const uint32_t r = d.krow0 + 16u * I;
    glds16(d.k + ((r < lim ? r : lim) * d.rs2 + d.kcol), smem + ST * IMG + d.wave * 1024 + I * 4096);
  } else {
    glds16(d.v + (size_t)(I - 4) * 32u * d.ldv2 + d.vofs0, smem + NST * IMG + ST * IMG + d.wave * 1024 + (I - 4) * 4096);
  }
}
// step to the next tile; after the last tile the stream stays put (later fetches re-read it into a dead stage)
__device__ __forceinline__ void dma_advance(Dma& d) {
  if (d.left > 1) {
    --d.left;
    if (++d.tt == d.tps) {
      d.tt = 0;
      ++d.seg;
      d.k = d.k0 + d.seg * d.kseg;
      d.v = d.v0 + d.seg * d.vseg;
    } else {
      d.k += (int64_t)KVBLK * d.rs2;
      d.v += KVBLK * 2;
    }
  }
}
// same with the piece index as an (unrolled, hence constant) loop variable
template <int ST>
__device__ __forceinline__ void dma_piece_i(char* smem, const Dma& d, int I) {
  if (I < 4) {
    const uint32_t lim = (d.tt == d.tps - 1) ? (uint32_t)d.tail_lim : 63u;
    const uint32_t r = d.krow0 + 16u * I;
    glds16(d.k + ((r < lim ? r : lim) * d.rs2 + d.kcol), smem + ST * IMG + d.wave * 1024 + I * 4096);
  } else {
    glds16(d.v + (size_t)(I - 4) * 32u * d.ldv2 + d.vofs0, smem + NST * IMG + ST * IMG + d.wave * 1024 + (I - 4) * 4096);
  }
}
template <int ST>
__device__ __forceinline__ void dma_tile(char* smem, Dma& d) {
  dma_piece<0, ST>(smem, d); dma_piece<1, ST>(smem, d); dma_piece<2, ST>(smem, d); dma_piece<3, ST>(smem, d);
  dma_piece<4, ST>(smem, d); dma_piece<5, ST>(smem, d); dma_piece<6, ST>(smem, d); dma_piece<7, ST>(smem, d);
  dma_advance(d);
}

}  // namespace
