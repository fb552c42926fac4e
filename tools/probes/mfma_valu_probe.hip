// Probe: does a wave's own VALU work issue under its own MFMAs on gfx950?
// One workgroup; each wave runs ITER x { 1 MFMA 32x32x16 bf16 (4 rotating accumulators) + N fillers } and reports
// s_memtime cycles per MFMA.  KIND 0: v_fma_f32 (independent registers), 1: v_exp_f32, 2: ds_read_b128, 3: v_cvt_pk_bf16_f32.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_valu_probe.hip -o gpurun_out/mfma_valu_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int N, int KIND, bool AGPR>
__global__ void probe(uint64_t* out, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[16384];
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)1.0f; b[i] = (__bf16)0.5f; }
  float x[8];
  for (int j = 0; j < 8; ++j) x[j] = threadIdx.x * 0.001f + j;
  uint4 d[8];
  const int laddr = (threadIdx.x & 63) * 16;
  __syncthreads();
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[u]) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(a), "v"(b));
#pragma unroll
      for (int j = 0; j < N; ++j) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[j & 7]) : "v"(x[(j + 1) & 7]));
        else if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x[j & 7]));
        else if (KIND == 2) asm volatile("ds_read_b128 %0, %1" : "=v"(d[j & 7]) : "v"(laddr));
        else asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x[j & 7]) : "v"(x[(j + 1) & 7]));
      }
    }
    if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_nop 15\n\ts_nop 15");
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) { if (AGPR) asm volatile("" : "+a"(acc[i])); s += acc[i][0]; }
  for (int j = 0; j < 8; ++j) s += x[j];
  if (KIND == 2) for (int j = 0; j < 8; ++j) s += (float)d[j].x;
  if ((threadIdx.x & 63) == 0) {
    out[threadIdx.x >> 6] = t1 - t0;
    out[16 + (threadIdx.x >> 6)] = (uint64_t)s;
  }
}

template <int N, int KIND, bool AGPR>
void run(const char* name, uint64_t* dbuf, int threads) {
  const int iters = 2000;
  uint64_t h[32];
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((probe<N, KIND, AGPR>), dim3(1), dim3(threads), 0, 0, dbuf, iters);
    hipDeviceSynchronize();
  }
  hipMemcpy(h, dbuf, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-10s N=%2d agpr=%d waves/SIMD=%d : %.1f cycles per MFMA (wave0), %.1f (last wave)\n", name, N, (int)AGPR, threads / 256,
         (double)h[0] / (iters * 4.0), (double)h[threads / 64 - 1] / (iters * 4.0));
}


// probe2: MFMA operand files selectable.  CLS bit0: acc in AGPR, bit1: A operand in AGPR, bit2: B operand in AGPR.
// fillers per MFMA: NE v_exp_f32 + NF v_fma_f32 + ND ds_read_b128 (DCLS 0: VGPR destination, 1: AGPR destination)
template <int CLS, int NE, int NF, int ND, int DCLS>
__global__ void probe2(uint64_t* out, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[16384];
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)1.0f; b[i] = (__bf16)0.5f; }
  if (CLS & 2) asm volatile("" : "+a"(a));
  if (CLS & 4) asm volatile("" : "+a"(b));
  float x[8];
  for (int j = 0; j < 8; ++j) x[j] = threadIdx.x * 0.001f + j;
  uint4 d[4];
  const int laddr = (threadIdx.x & 63) * 16;
  __syncthreads();
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (CLS == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(a), "v"(b));
      else if (CLS == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[u]) : "v"(a), "v"(b));
      else if (CLS == 3) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[u]) : "a"(a), "v"(b));
      else if (CLS == 4) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(a), "a"(b));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[u]) : "a"(a), "a"(b));
#pragma unroll
      for (int j = 0; j < NE; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(x[j & 7]));
#pragma unroll
      for (int j = 0; j < NF; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[(j + 4) & 7]) : "v"(x[(j + 5) & 7]));
#pragma unroll
      for (int j = 0; j < ND; ++j) {
        if (DCLS) asm volatile("ds_read_b128 %0, %1" : "=a"(d[j & 3]) : "v"(laddr));
        else asm volatile("ds_read_b128 %0, %1" : "=v"(d[j & 3]) : "v"(laddr));
      }
    }
    if (ND) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_nop 15\n\ts_nop 15");
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) { if (CLS & 1) asm volatile("" : "+a"(acc[i])); s += acc[i][0]; }
  for (int j = 0; j < 8; ++j) s += x[j];
  if ((threadIdx.x & 63) == 0) {
    out[threadIdx.x >> 6] = t1 - t0;
    out[16 + (threadIdx.x >> 6)] = (uint64_t)s;
  }
}
template <int CLS, int NE, int NF, int ND, int DCLS>
void run2(uint64_t* dbuf) {
  const int iters = 2000;
  uint64_t h[32];
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((probe2<CLS, NE, NF, ND, DCLS>), dim3(1), dim3(256), 0, 0, dbuf, iters);
    hipDeviceSynchronize();
  }
  hipMemcpy(h, dbuf, sizeof(h), hipMemcpyDeviceToHost);
  printf("probe2 cls=%d (acc %c, A %c, B %c)  exp=%d fma=%d ds_read=%d(dst %c) : %.1f cycles per MFMA\n", CLS, (CLS & 1) ? 'a' : 'v',
         (CLS & 2) ? 'a' : 'v', (CLS & 4) ? 'a' : 'v', NE, NF, ND, DCLS ? 'a' : 'v', (double)h[0] / (iters * 4.0));
}

int main() {
  uint64_t* dbuf;
  hipMalloc(&dbuf, 32 * sizeof(uint64_t));
  run2<0, 1, 2, 0, 0>(dbuf); run2<1, 1, 2, 0, 0>(dbuf); run2<3, 1, 2, 0, 0>(dbuf); run2<4, 1, 2, 0, 0>(dbuf); run2<7, 1, 2, 0, 0>(dbuf);
  run2<0, 1, 1, 0, 0>(dbuf); run2<3, 1, 1, 0, 0>(dbuf); run2<4, 1, 1, 0, 0>(dbuf);
  run2<4, 1, 2, 1, 0>(dbuf); run2<4, 1, 2, 1, 1>(dbuf); run2<3, 1, 2, 1, 0>(dbuf); run2<3, 1, 2, 1, 1>(dbuf);
  run2<4, 0, 0, 1, 0>(dbuf); run2<4, 0, 0, 1, 1>(dbuf); run2<3, 0, 0, 1, 0>(dbuf); run2<3, 0, 0, 1, 1>(dbuf);
  run2<4, 2, 3, 0, 0>(dbuf); run2<3, 2, 3, 0, 0>(dbuf);
  if (getenv("PROBE2_ONLY")) return 0;
  for (int threads : {256, 512}) {
    run<0, 0, false>("none", dbuf, threads);
    run<2, 0, false>("fma", dbuf, threads);
    run<4, 0, false>("fma", dbuf, threads);
    run<6, 0, false>("fma", dbuf, threads);
    run<8, 0, false>("fma", dbuf, threads);
    run<12, 0, false>("fma", dbuf, threads);
    run<16, 0, false>("fma", dbuf, threads);
    run<4, 0, true>("fma", dbuf, threads);
    run<8, 0, true>("fma", dbuf, threads);
    run<2, 1, false>("exp", dbuf, threads);
    run<4, 1, false>("exp", dbuf, threads);
    run<8, 1, false>("exp", dbuf, threads);
    run<4, 2, false>("ds_read", dbuf, threads);
    run<8, 2, false>("ds_read", dbuf, threads);
    run<4, 3, false>("cvt_pk", dbuf, threads);
    run<8, 3, false>("cvt_pk", dbuf, threads);
  }
  return 0;
}
