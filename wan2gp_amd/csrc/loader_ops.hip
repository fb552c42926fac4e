// Checkpoint-load-time kernels: LoRA merge and int8 dequantisation.
//
// The reference applies adapters at run time inside mmgp's patched Linear.forward (wgp.py:6922-6931,
// shared/utils/loras_mutipliers.py:143-148): y += m * (alpha / r) * (x A^T) B^T on every call.  Here the weights are
// resident in HBM, so the adapters are merged once:  W <- bf16(W + sum_i m_i (alpha_i / r_i) B_i A_i + sum_i m_i diff_i),
// the sum held in an fp32 scratch so that W sees a single rounding no matter how many adapters are stacked.
//
//   wan_lora_accumulate   acc[N,K] (fp32) += scale * B[N,r] A[r,K]        (fp32 FMA; r <= a few hundred)
//   wan_axpy_f32          acc += alpha * x                                 (`.diff` / `.diff_b` tensors)
//   wan_add_f32_into_bf16 W = bf16(float(W) + acc)                         (the one rounding)
//   wan_dequant_i8        W[n,k] = bf16(float(data[n,k]) * scale[n])       (optimum-quanto qint8 weights)
//
// All four are streaming kernels bound by HBM (acc 8 B, W 4 B per element); run once per LoRA / phase change, never
// inside a denoise step.  Plain FMA on purpose: at r = 64 the rank-r update is 16 FLOP/B, below the fp32 VALU ridge.
#include "common.h"

#define LT 64   // output tile edge
#define LR 16   // rank chunk staged through LDS

__global__ __launch_bounds__(256) void lora_accumulate_kernel(float* __restrict__ acc, const float* __restrict__ Bm,
                                                              const float* __restrict__ Am, float scale, int N, int K, int r) {
  __shared__ float sB[LR][LT + 1];   // [j][n]  (+1: the transposing store would otherwise hit one bank)
  __shared__ float sA[LR][LT];       // [j][k]
  const int n0 = blockIdx.y * LT, k0 = blockIdx.x * LT;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;   // thread owns rows ty*4..+3, cols tx*4..+3
  float sum[4][4] = {};
  for (int j0 = 0; j0 < r; j0 += LR) {
    for (int i = threadIdx.x; i < LT * LR; i += 256) {
      const int n = i / LR, j = i % LR;                      // B is [N, r]: j fastest in memory
      sB[j][n] = (n0 + n < N && j0 + j < r) ? Bm[(int64_t)(n0 + n) * r + j0 + j] : 0.f;
      const int jj = i / LT, k = i % LT;                     // A is [r, K]: k fastest
      sA[jj][k] = (j0 + jj < r && k0 + k < K) ? Am[(int64_t)(j0 + jj) * K + k0 + k] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < LR; ++j) {
      float b[4], a[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { b[u] = sB[j][ty * 4 + u]; a[u] = sA[j][tx * 4 + u]; }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int w = 0; w < 4; ++w) sum[u][w] = fmaf(b[u], a[w], sum[u][w]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int n = n0 + ty * 4 + u;
    if (n >= N) continue;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int k = k0 + tx * 4 + w;
      if (k < K) acc[(int64_t)n * K + k] = fmaf(scale, sum[u][w], acc[(int64_t)n * K + k]);
    }
  }
}

__global__ void axpy_f32_kernel(float* __restrict__ acc, const float* __restrict__ x, float alpha, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    acc[i] = fmaf(alpha, x[i], acc[i]);
}

__global__ void add_f32_into_bf16_kernel(bf16_t* __restrict__ w, const float* __restrict__ acc, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    w[i] = f2bf(bf2f(w[i]) + acc[i]);
}

__global__ void dequant_i8_kernel(const int8_t* __restrict__ data, const float* __restrict__ scale, bf16_t* __restrict__ out,
                                  int64_t N, int64_t K) {
  const int64_t total = N * K;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = f2bf((float)data[i] * scale[i / K]);
}

static inline int stream_blocks(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b > 256 * 32 ? 256 * 32 : (b < 1 ? 1 : b));
}

extern "C" int wan_lora_accumulate(float* acc, const float* lora_B, const float* lora_A, float scale, int N, int K, int r,
                                   void* stream) {
  WAN_REQUIRE(acc && lora_B && lora_A, "wan_lora_accumulate: null pointer");
  WAN_REQUIRE(N >= 1 && K >= 1 && r >= 1, "wan_lora_accumulate: bad shape N=%d K=%d r=%d", N, K, r);
  dim3 grid((K + LT - 1) / LT, (N + LT - 1) / LT);
  WAN_REQUIRE(grid.y <= 65535, "wan_lora_accumulate: N=%d too large", N);
  hipLaunchKernelGGL(lora_accumulate_kernel, grid, dim3(256), 0, as_stream(stream), acc, lora_B, lora_A, scale, N, K, r);
  WAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int wan_axpy_f32(float* acc, const float* x, float alpha, int64_t n, void* stream) {
  WAN_REQUIRE(acc && x && n >= 0, "wan_axpy_f32: null pointer");
  if (n == 0) return 0;
  hipLaunchKernelGGL(axpy_f32_kernel, dim3(stream_blocks(n)), dim3(256), 0, as_stream(stream), acc, x, alpha, n);
  WAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int wan_add_f32_into_bf16(wan_bf16* w, const float* acc, int64_t n, void* stream) {
  WAN_REQUIRE(w && acc && n >= 0, "wan_add_f32_into_bf16: null pointer");
  if (n == 0) return 0;
  hipLaunchKernelGGL(add_f32_into_bf16_kernel, dim3(stream_blocks(n)), dim3(256), 0, as_stream(stream), w, acc, n);
  WAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int wan_dequant_i8(const int8_t* data, const float* scale, wan_bf16* out, int64_t N, int64_t K, void* stream) {
  WAN_REQUIRE(data && scale && out, "wan_dequant_i8: null pointer");
  WAN_REQUIRE(N >= 1 && K >= 1, "wan_dequant_i8: bad shape");
  hipLaunchKernelGGL(dequant_i8_kernel, dim3(stream_blocks(N * K)), dim3(256), 0, as_stream(stream), data, scale, out, N, K);
  WAN_LAUNCH_CHECK();
  return 0;
}
