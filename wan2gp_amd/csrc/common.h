// Shared device/host helpers for libwanhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/wanhip.h"

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // 8 bf16 = one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// ---- error plumbing -------------------------------------------------------------------------
void wan_set_error(const char* fmt, ...);
#define WAN_CHECK_HIP(expr)                                                              \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      wan_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return 2;                                                                          \
    }                                                                                    \
  } while (0)
#define WAN_REQUIRE(cond, ...)      \
  do {                              \
    if (!(cond)) {                  \
      wan_set_error(__VA_ARGS__);   \
      return 1;                     \
    }                               \
  } while (0)
#define WAN_LAUNCH_CHECK() WAN_CHECK_HIP(hipGetLastError())

// ---- bf16 <-> f32 -----------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even through the hardware converter (v_cvt_pk_bf16_f32: one instruction per PAIR; the integer
// add-and-shift form it replaces cost ~6 VALU operations per element and made the GEMM epilogues VALU-bound: 33k cycles to
// drain a 256x256 tile).  Bit-identical to torch's float->bfloat16 for every non-NaN input; NaNs come out quiet.
typedef float wan_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 wan_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  wan_f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, wan_bf16x2));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack2bf(f, 0.f) & 0xffffu); }
// round a float through bf16 (value of the bf16 the reference would have stored)
__device__ __forceinline__ float rbf(float f) { return __uint_as_float(pack2bf(f, 0.f) << 16); }

// ---- fp32 products / sums hipcc must NOT fuse into a packed instruction that reads a register pair crosswise ------------------------
// Round 6, runs 80-83: `v_pk_mul_f32 d, a, b op_sel:[0,1] op_sel_hi:[1,0]` (low result = a.lo * b.HI, high result = a.hi * b.LO -- what the
// SLP vectoriser makes of RoPE's (-x1 sin0, x0 sin1)) returned a low result of ZERO in lanes 48-63 of a few waves while another process on
// the same GPU started or exited: every wrong element of the narrow in-place RMSNorm + RoPE launch was exactly x0 cos0 without its
// - x1 sin0 (2,533 of 2,533 in a dump; odd elements, other lanes: never).  With the two products as plain v_mul_f32 the same LDS-less kernel
// returned its bits in 385,549 of 385,549 launches; onto another register pair the packed form still failed.  So the library contains no
// packed-f32 instruction whose LOW lane selects a HIGH source register or the reverse (tests/test_isa_invariants.py); same IEEE operations, same bits.
__device__ __forceinline__ float wan_mul_f32(float a, float b) { float r; asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float wan_add_f32(float a, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// Where such an instruction cannot be written away (the device library's sin / cos in the timestep-sinusoid kernels) the kernel holds one LDS word:
// the allocation alone shielded the old rotation (run 80: 0 of 379,910 launches with the word and no barrier, 119 of 383,085 without).  Call it
// first, in front of any return; every other kernel of the library that holds one is an LDS-holding workgroup already (tests/test_isa_invariants.py).
__device__ __forceinline__ void wan_hold_lds_word() {
  __shared__ int wan_lds_word;
  if (threadIdx.x == 0) wan_lds_word = 0;
  asm volatile("" :: "v"(wan_lds_word));
}

// ---- fp16 <-> f32 (VAE path) ---------------------------------------------------------------------
__device__ __forceinline__ float h2f(uint16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ uint16_t f2h(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }

// element-type dispatch for kernels templated on the 16-bit storage type (bf16 DiT / fp16 VAE)
template <bool F16> __device__ __forceinline__ float ld16(uint16_t v) { return F16 ? h2f(v) : bf2f(v); }
template <bool F16> __device__ __forceinline__ uint16_t st16(float f) { return F16 ? f2h(f) : f2bf(f); }
template <bool F16> __device__ __forceinline__ float rnd16(float f) { return ld16<F16>(st16<F16>(f)); }
template <bool F16> __device__ __forceinline__ void unpack8t(const uint4& v, float* f);
template <bool F16> __device__ __forceinline__ uint4 pack8t(const float* f);

struct alignas(16) U4 {
  uint32_t x, y, z, w;
};

__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  uint4 v;
  v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]);
  v.z = pack2bf(f[4], f[5]); v.w = pack2bf(f[6], f[7]);
  return v;
}

template <> __device__ __forceinline__ void unpack8t<false>(const uint4& v, float* f) { unpack8(v, f); }
template <> __device__ __forceinline__ uint4 pack8t<false>(const float* f) { return pack8(f); }
template <> __device__ __forceinline__ void unpack8t<true>(const uint4& v, float* f) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = h2f((uint16_t)(w[i] & 0xffffu));
    f[2 * i + 1] = h2f((uint16_t)(w[i] >> 16));
  }
}
template <> __device__ __forceinline__ uint4 pack8t<true>(const float* f) {
  uint4 v;
  v.x = (uint32_t)f2h(f[0]) | ((uint32_t)f2h(f[1]) << 16);
  v.y = (uint32_t)f2h(f[2]) | ((uint32_t)f2h(f[3]) << 16);
  v.z = (uint32_t)f2h(f[4]) | ((uint32_t)f2h(f[5]) << 16);
  v.w = (uint32_t)f2h(f[6]) | ((uint32_t)f2h(f[7]) << 16);
  return v;
}

// ---- wave (64-lane) reductions ------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- XCD-aware block remap (cdna guide T1, bijective form) --------------------------------------
// Hardware places block b on XCD b % 8; give each XCD a contiguous range of logical ids so that
// neighbouring tiles (sharing operand panels) hit the same private L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int NX = 8;
  int xcd = bid % NX, idx = bid / NX;
  int q = nwg / NX, r = nwg % NX;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// ---- LDS-DMA (global -> LDS without registers), 16 B per lane, 1 KiB per wave-instruction -----------------------------------
// `ldst` is the wave-uniform LDS address the wave's 64 lanes fill contiguously (lane l lands at +16 l); `gsrc` is per lane.
// INLINE ASM on purpose: for the builtin (__builtin_amdgcn_global_load_lds) hipcc's waitcnt pass inserts `s_waitcnt vmcnt(0)` in
// front of every later ds_read it cannot prove disjoint from the pending DMA -- in a ring of LDS stages that serialises the
// prefetch with the compute of the stage being read (measured in the attention kernel: 2,500 cycles per wait).  The asm form is
// invisible to the pass: every kernel that uses it counts completion by hand (s_waitcnt vmcnt(N) + barrier before the reads).
__device__ __forceinline__ void wan_lds_dma16(const void* gsrc, void* ldst) {
  const uint32_t lds_addr = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ldst);  // wave-uniform by contract
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_addr) : "memory");
}

static inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }

// ---- library-owned scratch rings for ops called without a caller workspace (elementwise.hip) --------------------------------------
// One ring of `nslot` slots x `slot_bytes` per (purpose tag, device, stream): calls on one stream are serialised, so a slot is dead
// `nslot` calls later; another stream or another device gets a ring of its own (round-3 advisor finding: one process-global ring was
// allocated on whichever device was current first and handed the same slots to every stream).  Thread-safe.  NULL when the request
// does not fit a slot, the table of rings is full (64) or the allocation fails -- callers then take their scratch-free path.
void* wan_scratch_ring_slot(int tag, size_t slot_bytes, int nslot, size_t need_bytes, hipStream_t stream);
