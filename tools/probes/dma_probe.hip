// Probe: issue cost of LDS-DMA (buffer_load_dwordx4 .. offen lds) between MFMAs, one workgroup of 4 waves (one per SIMD).
// modes: 0 no DMA; 1 every wave one DMA per 4 MFMAs (same gaps); 2 only wave (it & 3) issues, one DMA per MFMA;
//        3 every wave issues one DMA per MFMA but EXEC is zero except in wave (it & 3); 4 like 1 but wave w uses gap w.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE>
__global__ __launch_bounds__(256) void probe(const char* src, uint64_t* out, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[65536];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)1.0f; b[i] = (__bf16)0.5f; }
  const uint64_t base = (uint64_t)src;
  u4 rs;
  rs[0] = (uint32_t)base; rs[1] = (uint32_t)(base >> 32) & 0xffffu; rs[2] = 0xffffffffu; rs[3] = 0x00020000u;
  const uint32_t voff = lane * 16 + wave * 4096;
  const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds + wave * 16384;
  __syncthreads();
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(a), "v"(b));
      const uint32_t la = lbase + u * 1024;
      if (MODE == 1) {
        if (u == 0) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rs), "s"(la) : "memory");
      } else if (MODE == 4) {
        if (u == wave) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rs), "s"(la) : "memory");
      } else if (MODE == 2) {
        if ((it & 3) == wave) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rs), "s"(la) : "memory");
      } else if (MODE == 3) {
        const uint32_t mask = __builtin_amdgcn_readfirstlane(((it & 3) == wave) ? 0xffffffffu : 0u);
        asm volatile("s_mov_b32 exec_lo, %3\n\ts_mov_b32 exec_hi, %3\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds\n\ts_mov_b64 exec, -1" ::"v"(voff), "s"(rs), "s"(la), "s"(mask) : "memory");
      }
    }
    if ((it & 7) == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0];
  if (lane == 0) { out[wave] = t1 - t0; out[8 + wave] = (uint64_t)s + lds[lane]; }
}

template <int MODE>
void run(const char* what, const char* src, uint64_t* dbuf) {
  const int iters = 4000;
  uint64_t h[16];
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(256), 0, 0, src, dbuf, iters); hipDeviceSynchronize(); }
  hipMemcpy(h, dbuf, sizeof(h), hipMemcpyDeviceToHost);
  printf("dma_probe mode %d (%s): cycles per MFMA, waves 0..3: %.1f %.1f %.1f %.1f\n", MODE, what, h[0] / (iters * 4.0), h[1] / (iters * 4.0),
         h[2] / (iters * 4.0), h[3] / (iters * 4.0));
}
int main() {
  char* src; uint64_t* dbuf;
  hipMalloc(&src, 1 << 20); hipMemset(src, 0, 1 << 20); hipMalloc(&dbuf, 16 * 8);
  run<0>("no DMA", src, dbuf);
  run<1>("all waves, 1 DMA per 4 MFMAs, same gap", src, dbuf);
  run<4>("all waves, 1 DMA per 4 MFMAs, wave w in gap w", src, dbuf);
  run<2>("one wave at a time, 1 DMA per MFMA (branch)", src, dbuf);
  run<3>("all waves issue, EXEC=0 except one wave, 1 per MFMA", src, dbuf);
  return 0;
}
