// Probe: sustained bf16 MFMA throughput of the whole chip on RANDOM data, 32x32x16 against 16x16x32, with and without the GEMM's
// LDS fragment traffic.  Question (round 3, run 64): gemm256p saves 8 % of gemm256k's cycles per tile and gains nothing in wall
// time -- the chip runs these kernels at its power limit (effective clock 1.42 vs 1.56 GHz), so what matters is energy per FLOP.
// The vendor's hand-tuned 256x256x64 kernel uses 16x16x32 (half the accumulator read/write traffic per MAC: K = 32 per
// instruction instead of 16).  One workgroup of 4 waves per CU, 256 accumulator registers per lane, ~100 ms per kernel.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_power_probe mfma_power_probe.hip && ./mfma_power_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(4))) float f4v;
typedef __attribute__((ext_vector_type(4))) uint32_t u4;

template <bool LDS>
__global__ __launch_bounds__(256) void k32(const u4* __restrict__ src, float* __restrict__ out, int iters) {
  __shared__ __attribute__((aligned(16))) u4 lds[4096];  // 64 KB
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = src[i];
  __syncthreads();
  const int lane = threadIdx.x;
  bf8 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = __builtin_bit_cast(bf8, src[lane + 256 * i]);
    b[i] = __builtin_bit_cast(bf8, src[lane + 256 * (4 + i)]);
  }
  f16v acc[4][4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  int off = lane;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (LDS) {  // the GEMM's rate: 8 fragment reads per 16 MFMAs
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          a[i] = __builtin_bit_cast(bf8, lds[(off + 256 * i) & 4095]);
          b[i] = __builtin_bit_cast(bf8, lds[(off + 256 * (4 + i) + 64) & 4095]);
        }
        off += 1024;
      }
#pragma unroll
      for (int m = 0; m < 16; ++m) acc[m >> 2][m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m >> 2], b[m & 3], acc[m >> 2][m & 3], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <bool LDS>
__global__ __launch_bounds__(256) void k16(const u4* __restrict__ src, float* __restrict__ out, int iters) {
  __shared__ __attribute__((aligned(16))) u4 lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = src[i];
  __syncthreads();
  const int lane = threadIdx.x;
  bf8 a[8], b[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = __builtin_bit_cast(bf8, src[lane + 256 * (i & 3)]);
    b[i] = __builtin_bit_cast(bf8, src[lane + 256 * (4 + (i & 3)) + 64 * (i >> 2)]);
  }
  f4v acc[8][8];
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 8; ++j)
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
  int off = lane;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {  // k = 32 per step: the same 64 k per iteration as k32
      if (LDS) {                      // 16 fragment reads per 64 MFMAs: the same bytes per FLOP
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          a[i] = __builtin_bit_cast(bf8, lds[(off + 256 * i) & 4095]);
          b[i] = __builtin_bit_cast(bf8, lds[(off + 256 * i + 2048 + 64) & 4095]);
        }
        off += 1024;
      }
#pragma unroll
      for (int m = 0; m < 64; ++m) acc[m >> 3][m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m >> 3], b[m & 7], acc[m >> 3][m & 7], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 8; ++j)
      for (int r = 0; r < 4; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// fp8 (e4m3fn) operands, v_mfma_f32_32x32x64_f8f6f4: 8 VGPRs per operand (32 bytes per lane), K = 64 per instruction
typedef __attribute__((ext_vector_type(8))) uint32_t u8v;
__global__ __launch_bounds__(256) void k32_fp8(const u4* __restrict__ src, float* __restrict__ out, int iters) {
  const int lane = threadIdx.x;
  u8v a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    const u4 lo = src[lane + 256 * i], hi = src[lane + 256 * i + 1024];
    const u4 lo2 = src[lane + 256 * (4 + i)], hi2 = src[lane + 256 * (4 + i) + 1024];
    a[i] = u8v{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    b[i] = u8v{lo2[0], lo2[1], lo2[2], lo2[3], hi2[0], hi2[1], hi2[2], hi2[3]};
    // keep the bytes inside the finite e4m3 range (0x7f / 0xff are NaN): clear bit 6 of every byte -> |x| < 2^1, all mantissas
    for (int k = 0; k < 8; ++k) { a[i][k] &= 0xbfbfbfbfu; b[i][k] &= 0xbfbfbfbfu; }
  }
  f16v acc[4][4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int m = 0; m < 16; ++m) asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0" : "+a"(acc[m >> 2][m & 3]) : "v"(a[m >> 2]), "v"(b[m & 3]));
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}


// The GEMM's operand stream under the MFMA loop: per k-tile of 64 (64 MFMAs 32x32x16, or 128 MFMAs 16x16x32, of each wave) the
// workgroup brings 64 KB into LDS with 64 LDS-DMA pieces (buffer_load_dwordx4 ... lds, 16 per wave), 128 B per row per fetch, in
// gemm256k's pattern: CU b works on "tile" (b / 16, b % 16) of a [8192 x KW] bf16 matrix -- Y rows (b/16)*256.., X rows 4096 + (b%16)*256..
// SRC 0: no stream; 1: every CU streams the same 2 MB (L2-resident); 2: the GEMM pattern over the 128-MB matrix (L2 + MALL/HBM mix).
// MODE bit 0: 16x16x32 instead of 32x32x16; bit 1: a workgroup barrier per k-tile (the GEMM's sync point); bit 2: odd waves (SIMDs 1, 3)
// issue their fragment reads and DMA pieces half a k-step later than even waves (the vendor kernel's SIMD-parity pair of loop bodies);
// bit 3: no MFMAs (the stream alone).  Fragments are double-buffered (read for k-step q + 1 while k-step q multiplies), as in the kernels.
// Run under rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES: clock = GRBM_GUI_ACTIVE / 8 / time, utilisation = busy / (1024 x cycles).
__device__ __forceinline__ void dma16(uint32_t voff, const u4& rsrc, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rsrc), "s"(lds_addr) : "memory");
}
template <int SRC, int MODE>
__global__ __launch_bounds__(256) void ks(const u4* __restrict__ src, const char* __restrict__ big, float* __restrict__ out, int iters, int KW) {
  constexpr bool MI16 = MODE & 1, BAR = MODE & 2, STAG = MODE & 4, MFMA = !(MODE & 8);
  constexpr int NF = MI16 ? 8 : 4;                          // fragments per operand per k-step
  constexpr int NQ = MI16 ? 2 : 4;                          // k-steps per k-tile of 64
  constexpr int NM = NF * NF;                               // MFMAs per k-step
  extern __shared__ __attribute__((aligned(16))) u4 lds[];  // 2 x 64 KB
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = src[i & 4095];
  __syncthreads();
  const int lane = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool late = STAG && (wave & 1);
  bf8 a[2][NF], b[2][NF];
  for (int i = 0; i < NF; ++i)
    for (int h = 0; h < 2; ++h) {
      a[h][i] = __builtin_bit_cast(bf8, src[(lane + 256 * i + 64 * h) & 4095]);
      b[h][i] = __builtin_bit_cast(bf8, src[(lane + 256 * (NF + i) + 64 * h) & 4095]);
    }
  f16v acc32[MI16 ? 1 : 4][MI16 ? 1 : 4];
  f4v acc16[MI16 ? 8 : 1][MI16 ? 8 : 1];
  for (auto& r : acc32) for (auto& c : r) for (int k = 0; k < 16; ++k) c[k] = 0.f;
  for (auto& r : acc16) for (auto& c : r) for (int k = 0; k < 4; ++k) c[k] = 0.f;
  const uint64_t bb = (uint64_t)big;
  u4 rsrc;
  rsrc[0] = __builtin_amdgcn_readfirstlane((uint32_t)bb);
  rsrc[1] = __builtin_amdgcn_readfirstlane((uint32_t)(bb >> 32) & 0xffffu);
  rsrc[2] = __builtin_amdgcn_readfirstlane(0xffffffffu);
  rsrc[3] = __builtin_amdgcn_readfirstlane(0x00020000u);
  const uint32_t ldsb = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
  // piece p (0..15) of this wave: operand p >> 3, rows ((p & 7) * 4 + wave) * 8 + lane / 8 of its 256, 16 B at (lane & 7) * 16
  uint32_t rowofs[16];
  const uint32_t rowbytes = (uint32_t)KW * 2u;
  for (int p = 0; p < 16; ++p) {
    const uint32_t r = (((p & 7) * 4 + wave) * 8 + ((lane & 63) >> 3));
    const uint32_t panel = (SRC == 2) ? ((p >> 3) ? 4096u + (blockIdx.x & 15) * 256u : (blockIdx.x >> 4) * 256u) : ((p >> 3) * 256u);
    rowofs[p] = (panel + r) * ((SRC == 2) ? rowbytes : 4096u) + (lane & 7) * 16u;   // SRC 1: a 4096-B pitch: 512 rows x 4 KB = 2 MB
  }
  const int nk = (SRC == 2) ? KW / 64 : 32;
  int off = lane, ks_ = 0, slot = 0;
  auto reads = [&](int h) {
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      a[h][i] = __builtin_bit_cast(bf8, lds[(off + 256 * i) & 4095]);
      b[h][i] = __builtin_bit_cast(bf8, lds[(off + 256 * (NF + i) + 64) & 4095]);
    }
    off += 256 * 2 * NF;
  };
  for (int it = 0; it < iters; ++it) {
    const uint32_t kofs = (uint32_t)ks_ * 128u;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if (MFMA && !late) reads((q + 1) & 1);
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        if (MFMA && late && m == NM / 2) reads((q + 1) & 1);
        if (MFMA) {
          if (MI16) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc16[MI16 ? m / NF : 0][MI16 ? m % NF : 0]) : "v"(a[q & 1][m / NF]), "v"(b[q & 1][m % NF]));
          else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc32[MI16 ? 0 : m / NF][MI16 ? 0 : m % NF]) : "v"(a[q & 1][m / NF]), "v"(b[q & 1][m % NF]));
        }
        constexpr int every = NM * NQ / 16;                 // MFMAs per DMA piece
        if (SRC && (m % every) == (late ? every / 2 : 0)) {
          const int p = (q * NM + m) / every;
          dma16(rowofs[p] + kofs, rsrc, __builtin_amdgcn_readfirstlane(ldsb + slot * 65536 + (p * 256 + wave * 64) * 16));
        }
      }
    }
    if (SRC) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // one k-tile in flight behind the one being issued
    if (BAR) __builtin_amdgcn_s_barrier();
    ks_ = ks_ + 1 == nk ? 0 : ks_ + 1;
    slot ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float sum = 0.f;
  for (auto& r : acc32) for (auto& c : r) for (int k = 0; k < 16; ++k) sum += c[k];
  for (auto& r : acc16) for (auto& c : r) for (int k = 0; k < 4; ++k) sum += c[k];
  out[blockIdx.x * 256 + threadIdx.x] = sum;
}

// The attention tile's energy budget (round 3, run 89): 16x16x32 MFMAs with the kernel's LDS fragment traffic (32 ds_read_b128 per 128
// MFMAs = twice the GEMM's bytes per FLOP) and, per MFMA pair, the softmax's vector work (one v_exp_f32, one v_add_f32, half a
// v_cvt_pk_bf16_f32).  MODE bit 0: the fragment reads; bit 1: the vector work.
template <int MODE>
__global__ __launch_bounds__(256) void ka(const u4* __restrict__ src, float* __restrict__ out, int iters) {
  __shared__ __attribute__((aligned(16))) u4 lds[4096 + 512];  // 72 KB
  for (int i = threadIdx.x; i < 4096 + 512; i += 256) lds[i] = src[i & 4095];
  __syncthreads();
  const int lane = threadIdx.x;
  bf8 a[2][8], b[8];
  for (int i = 0; i < 8; ++i) {
    a[0][i] = __builtin_bit_cast(bf8, src[(lane + 256 * i) & 4095]);
    a[1][i] = __builtin_bit_cast(bf8, src[(lane + 256 * i + 64) & 4095]);
    b[i] = __builtin_bit_cast(bf8, src[(lane + 256 * (8 + i)) & 4095]);
  }
  f4v acc[8][8];
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 8; ++j)
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
  float e[4] = {0.3f + lane * 1e-3f, -0.7f, 1.1f, -0.2f}, l = 0.f;
  uint32_t pk = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {   // 64 MFMAs per q: 16 fragment reads (immediate offsets from one base register), 32 exps
      uint32_t boff = lane + 64 * ((it + q) & 3);
      asm volatile("" : "+v"(boff));
#pragma unroll
      for (int m = 0; m < 64; ++m) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[m >> 3][m & 7]) : "v"(a[q][m >> 3]), "v"(b[m & 7]));
        __builtin_amdgcn_sched_barrier(0);
        if ((MODE & 1) && (m & 3) == 1) {   // 16 reads per 64 MFMAs: 8 into the other A buffer, 8 into B
          if (((m >> 2) & 1) == 0) a[q ^ 1][(m >> 3) & 7] = __builtin_bit_cast(bf8, lds[boff + 256 * (m >> 3)]);
          else b[(m >> 3) & 7] = __builtin_bit_cast(bf8, lds[boff + 256 * (8 + (m >> 3)) + 32]);
        }
        if (MODE & 2) {
          if ((m & 1) == 0) {
            e[(m >> 1) & 3] = __builtin_amdgcn_exp2f(-e[((m >> 1) + 1) & 3]);   // stays in (0.5, 1): 2^-x of x in (0, 1)
            asm volatile("" : "+v"(e[(m >> 1) & 3]));
          } else {
            l += e[((m >> 1) + 2) & 3];
            asm volatile("" : "+v"(l));
            if ((m & 3) == 3) {
              typedef float f2v __attribute__((ext_vector_type(2)));
              typedef __bf16 b2v __attribute__((ext_vector_type(2)));
              f2v t = {e[0], e[1]};
              pk ^= __builtin_bit_cast(uint32_t, __builtin_convertvector(t, b2v));
              asm volatile("" : "+v"(pk));
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float s = l + (float)pk;
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 8; ++j)
      for (int r = 0; r < 4; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

static uint16_t bf16_of(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1)) >> 16);
}

int main() {
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  std::vector<uint16_t> h(4096 * 8);
  srand(7);
  for (auto& v : h) {  // ~N(0,1) by summing uniforms: full-mantissa random bf16 like activations / weights
    float s = 0;
    for (int k = 0; k < 12; ++k) s += rand() / (float)RAND_MAX;
    v = bf16_of(s - 6.0f);
  }
  u4* src;
  float* out;
  hipMalloc(&src, 4096 * 16);
  hipMalloc(&out, cus * 256 * 4);
  hipMemcpy(src, h.data(), 4096 * 16, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 60000;  // x 64 MFMA-equivalents of 32 cycles = 123 M cycles ~ 60-80 ms
  const double flop = (double)cus * 4 * iters * 4.0 * 16 * 32768.0;
  for (int rep = 0; rep < 2; ++rep) {
    for (int v = 0; v < 4; ++v) {
      hipEventRecord(e0);
      if (v == 0) hipLaunchKernelGGL(k32<false>, dim3(cus), dim3(256), 0, 0, src, out, iters);
      if (v == 1) hipLaunchKernelGGL(k16<false>, dim3(cus), dim3(256), 0, 0, src, out, iters);
      if (v == 2) hipLaunchKernelGGL(k32<true>, dim3(cus), dim3(256), 0, 0, src, out, iters);
      if (v == 3) hipLaunchKernelGGL(k16<true>, dim3(cus), dim3(256), 0, 0, src, out, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      const char* names[4] = {"32x32x16 registers only", "16x16x32 registers only", "32x32x16 + LDS fragment reads", "16x16x32 + LDS fragment reads"};
      printf("%-32s %8.2f ms  %7.1f TFLOP/s  (implied clock %.3f GHz at one MFMA pipe pass per cycle)\n", names[v], ms, flop / ms / 1e9,
             (double)iters * 64 * 32 / (ms * 1e-3) / 1e9);
    }
  }
  {  // fp8: 4 x the FLOP per instruction of the bf16 32x32x16 at twice its duration
    const int it8 = 30000;
    const double flop8 = (double)cus * 4 * it8 * 4.0 * 16 * 32.0 * 32 * 64 * 2;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k32_fp8, dim3(cus), dim3(256), 0, 0, src, out, it8);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      printf("%-32s %8.2f ms  %7.1f TFLOP/s  (of the 5 PFLOP/s fp8 peak: %.3f)\n", "32x32x64 fp8 registers only", ms, flop8 / ms / 1e9, flop8 / ms / 1e9 / 5000.0);
    }
  }
  {  // the operand stream: what it costs, and what it is alone
    const int KW = 8192;
    const size_t big_bytes = (size_t)8192 * KW * 2;   // 128 MB
    char* big;
    hipMalloc(&big, big_bytes);
    for (size_t o = 0; o < big_bytes; o += 4096 * 16) hipMemcpy(big + o, src, 4096 * 16, hipMemcpyDeviceToDevice);
    typedef void (*kfn)(const u4*, const char*, float*, int, int);
    struct V { kfn f; const char* name; bool mfma; bool stream; };
    const V vs[] = {
        {ks<0, 0>, "A  32x32x16 + LDS reads", true, false},
        {ks<1, 0>, "B  A + stream, 2 MB window (L2 hits)", true, true},
        {ks<2, 0>, "C  A + stream, GEMM pattern over 128 MB", true, true},
        {ks<2, 2>, "D  C + barrier per k-tile", true, true},
        {ks<2, 6>, "E  D + SIMD-parity stagger", true, true},
        {ks<2, 4>, "F  C + SIMD-parity stagger, no barrier", true, true},
        {ks<0, 1>, "G  16x16x32 + LDS reads", true, false},
        {ks<2, 1>, "H  G + stream, GEMM pattern", true, true},
        {ks<2, 3>, "I  H + barrier per k-tile", true, true},
        {ks<2, 7>, "J  I + SIMD-parity stagger", true, true},
        {ks<1, 8>, "K  stream alone, 2 MB window", false, true},
        {ks<2, 8>, "L  stream alone, GEMM pattern", false, true},
    };
    const int its = 50000;
    const double fl = (double)cus * 4 * its * 4.0 * 16 * 32768.0, bytes = (double)cus * its * 65536.0;
    for (const V& v : vs) hipFuncSetAttribute((const void*)v.f, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    for (int rep = 0; rep < 2; ++rep)
      for (const V& v : vs) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(v.f, dim3(cus), dim3(256), 131072, 0, src, big, out, its, KW);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (hipGetLastError() != hipSuccess) printf("launch error\n");
        if (v.mfma) printf("%-44s %8.2f ms  %7.1f TFLOP/s  stream %5.2f TB/s into the CUs\n", v.name, ms, fl / ms / 1e9, v.stream ? bytes / ms / 1e9 : 0.0);
        else printf("%-44s %8.2f ms  stream %5.2f TB/s into the CUs\n", v.name, ms, bytes / ms / 1e9);
      }
  }
  {  // the attention tile's energy budget
    typedef void (*kfn)(const u4*, float*, int);
    const kfn fs[4] = {ka<0>, ka<1>, ka<2>, ka<3>};
    const char* nm[4] = {"AT0 16x16x32, registers only (asm, AGPR acc)", "AT1 + the attention tile's fragment reads", "AT2 + its softmax vector work (exp, add, cvt/2 per MFMA pair)", "AT3 + both"};
    const int its = 120000;  // ~130 ms per kernel: the power-limited state (a 27-ms kernel still runs at 2.04 GHz)
    const double fl = (double)cus * 4 * its * 128.0 * 16384.0;
    for (int rep = 0; rep < 2; ++rep)
      for (int v = 0; v < 4; ++v) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(fs[v], dim3(cus), dim3(256), 0, 0, src, out, its);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-66s %8.2f ms  %7.1f TFLOP/s\n", nm[v], ms, fl / ms / 1e9);
      }
  }
  return 0;
}
