// UMT5 text-encoder pieces that the DiT kernels do not already provide (SURVEY.md section 8(f) rank 1:
// models/wan/modules/t5.py).  The encoder runs once per prompt (24 layers x 512 tokens, ~5 TFLOP): its Linear layers
// reuse wan_gemm_bf16, T5LayerNorm reuses the RMSNorm kernel; what is new is the attention -- head_dim 64, NO
// 1/sqrt(d) scaling, an additive relative-position bias and a padding mask (t5.py:92-131) -- and the gated-GELU product.
//
// wan_t5_attention follows the reference's rounding points: scores are a bf16 tensor (einsum output), the bias is added
// in bf16, masked keys are set to finfo(bf16).min, the softmax runs in fp32 and is cast back to bf16 before P V.
// One workgroup per (16 query rows, head, batch); the whole key range (L <= 1024) is held in LDS as fp32 scores.
// Not a hot kernel (0.2 % of one denoise step, once per video): plain FMA, no MFMA.
#include "common.h"

#define T5_QROWS 16
#define T5_HD 64
#define T5_LMAX 1024

__global__ __launch_bounds__(256) void t5_attention_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                          const bf16_t* __restrict__ V, const bf16_t* __restrict__ relbias,
                                                          const int32_t* __restrict__ mask, bf16_t* __restrict__ O, int L, int H) {
  __shared__ float qs[T5_QROWS][T5_HD];
  __shared__ float ps[T5_QROWS][T5_LMAX];
  const int tid = threadIdx.x;
  const int q0 = blockIdx.x * T5_QROWS, h = blockIdx.y, b = blockIdx.z;
  const int64_t rs = (int64_t)H * T5_HD;  // row stride of q / k / v / o
  const bf16_t* qb = Q + ((int64_t)b * L) * rs + (int64_t)h * T5_HD;
  const bf16_t* kb = K + ((int64_t)b * L) * rs + (int64_t)h * T5_HD;
  const bf16_t* vb = V + ((int64_t)b * L) * rs + (int64_t)h * T5_HD;
  bf16_t* ob = O + ((int64_t)b * L) * rs + (int64_t)h * T5_HD;
  const bf16_t* bias = relbias + (int64_t)h * (2 * L - 1);
  const float NEG = -3.3895313892515355e38f;  // torch.finfo(torch.bfloat16).min

  for (int e = tid; e < T5_QROWS * T5_HD; e += 256) {
    const int i = e / T5_HD, c = e % T5_HD;
    qs[i][c] = (q0 + i < L) ? bf2f(qb[(int64_t)(q0 + i) * rs + c]) : 0.f;
  }
  __syncthreads();
  // ---- scores: key j = tid, tid + 256, ...  (k row in registers, q rows broadcast from LDS) ---------------------------
  for (int j = tid; j < L; j += 256) {
    float kr[T5_HD];
#pragma unroll
    for (int c8 = 0; c8 < T5_HD / 8; ++c8) unpack8(*reinterpret_cast<const uint4*>(kb + (int64_t)j * rs + c8 * 8), kr + c8 * 8);
    const bool keep = (mask == nullptr) || (mask[(int64_t)b * L + j] != 0);
#pragma unroll 4
    for (int i = 0; i < T5_QROWS; ++i) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < T5_HD; ++c) s += qs[i][c] * kr[c];
      // einsum output is a bf16 tensor; + attn_bias (bf16) rounds again; masked_fill_ AFTER the position bias was added
      const int qi = q0 + i;
      if (qi >= L) { ps[i][j] = 0.f; continue; }  // padding rows of the last q block: no bias entry exists for them (index < 0)
      float a = rbf(rbf(s) + (keep ? bf2f(bias[j - qi + L - 1]) : 0.f));
      if (!keep) a = rbf(rbf(s) + NEG);  // attn_bias = min, then scores + attn_bias in bf16 (saturates at min / -inf)
      ps[i][j] = a;
    }
  }
  __syncthreads();
  // ---- fp32 softmax per row (4 rows per wave), cast to bf16 (".type_as(attn)") -------------------------------------------
  const int lane = tid & 63, wave = tid >> 6;
  for (int i = wave; i < T5_QROWS; i += 4) {
    float m = -INFINITY;
    for (int j = lane; j < L; j += 64) m = fmaxf(m, ps[i][j]);
    m = wave_max(m);
    float sum = 0.f;
    for (int j = lane; j < L; j += 64) {
      const float p = __expf(ps[i][j] - m);
      ps[i][j] = p;
      sum += p;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int j = lane; j < L; j += 64) ps[i][j] = rbf(ps[i][j] * inv);
  }
  __syncthreads();
  // ---- O = P V: thread -> (row i, 4 consecutive channels) -------------------------------------------------------------------
  const int i = tid >> 4, c4 = (tid & 15) * 4;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int j = 0; j < L; ++j) {
    const float p = ps[i][j];
    const uint2 raw = *reinterpret_cast<const uint2*>(vb + (int64_t)j * rs + c4);
    acc[0] += p * __uint_as_float(raw.x << 16);
    acc[1] += p * __uint_as_float(raw.x & 0xffff0000u);
    acc[2] += p * __uint_as_float(raw.y << 16);
    acc[3] += p * __uint_as_float(raw.y & 0xffff0000u);
  }
  if (q0 + i < L) {
    uint2 w;
    w.x = (uint32_t)f2bf(acc[0]) | ((uint32_t)f2bf(acc[1]) << 16);
    w.y = (uint32_t)f2bf(acc[2]) | ((uint32_t)f2bf(acc[3]) << 16);
    *reinterpret_cast<uint2*>(ob + (int64_t)(q0 + i) * rs + c4) = w;
  }
}

// OP: 0 = a * b, 1 = a + b, 2 = a - b; fp32 arithmetic on the bf16 values, one rounding (what torch does for bf16 tensors).
// a, b and o may alias element for element (no __restrict__).
template <int OP>
__global__ void binary_bf16_kernel(const bf16_t* a, const bf16_t* b, bf16_t* o, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    float x[8], y[8];
    unpack8(*reinterpret_cast<const uint4*>(a + i * 8), x);
    unpack8(*reinterpret_cast<const uint4*>(b + i * 8), y);
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = OP == 0 ? x[j] * y[j] : OP == 1 ? x[j] + y[j] : x[j] - y[j];
    *reinterpret_cast<uint4*>(o + i * 8) = pack8(x);
  }
}

extern "C" int wan_t5_attention(const wan_bf16* q, const wan_bf16* k, const wan_bf16* v, const wan_bf16* relbias,
                                const int32_t* mask, wan_bf16* out, int B, int L, int H, void* stream) {
  WAN_REQUIRE(q && k && v && relbias && out, "wan_t5_attention: null pointer");
  WAN_REQUIRE(B >= 1 && H >= 1 && L >= 1 && L <= T5_LMAX, "wan_t5_attention: L=%d must be in [1, %d]", L, T5_LMAX);
  WAN_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) == 0, "wan_t5_attention: pointers must be 16-byte aligned");
  dim3 grid((L + T5_QROWS - 1) / T5_QROWS, H, B);
  hipLaunchKernelGGL(t5_attention_kernel, grid, dim3(256), 0, as_stream(stream), q, k, v, relbias, mask, out, L, H);
  WAN_LAUNCH_CHECK();
  return 0;
}

// out = bf16(x + alpha * y): torch's `x.add_(y, alpha=alpha)` on bf16 tensors (fp32 product, fp32 sum, one rounding;
// __fmul_rn / __fadd_rn keep the compiler from contracting the two into an FMA).  x and out may alias.
__global__ void axpy_bf16_kernel(const bf16_t* x, const bf16_t* __restrict__ y, float alpha, bf16_t* o, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    float a[8], b[8];
    unpack8(*reinterpret_cast<const uint4*>(x + i * 8), a);
    unpack8(*reinterpret_cast<const uint4*>(y + i * 8), b);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = __fadd_rn(a[j], __fmul_rn(alpha, b[j]));
    *reinterpret_cast<uint4*>(o + i * 8) = pack8(a);
  }
}
extern "C" int wan_axpy_bf16(const wan_bf16* x, const wan_bf16* y, float alpha, wan_bf16* out, int64_t n, void* stream) {
  WAN_REQUIRE(x && y && out, "wan_axpy_bf16: null pointer");
  WAN_REQUIRE(n % 8 == 0 && ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)out) & 15) == 0), "wan_axpy_bf16: n %% 8 and 16-byte alignment required");
  if (n == 0) return 0;
  const int64_t n8 = n / 8;
  int blocks = (int)((n8 + 255) / 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(axpy_bf16_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), x, y, alpha, out, n8);
  WAN_LAUNCH_CHECK();
  return 0;
}

template <int OP>
static int binary_bf16(const wan_bf16* a, const wan_bf16* b, wan_bf16* out, int64_t n, void* stream, const char* what) {
  WAN_REQUIRE(a && b && out, "%s: null pointer", what);
  WAN_REQUIRE(n % 8 == 0 && ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) == 0), "%s: n %% 8 and 16-byte alignment required", what);
  if (n == 0) return 0;
  const int64_t n8 = n / 8;
  int blocks = (int)((n8 + 255) / 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(binary_bf16_kernel<OP>, dim3(blocks), dim3(256), 0, as_stream(stream), a, b, out, n8);
  WAN_LAUNCH_CHECK();
  return 0;
}
extern "C" int wan_mul_bf16(const wan_bf16* a, const wan_bf16* b, wan_bf16* out, int64_t n, void* stream) {
  return binary_bf16<0>(a, b, out, n, stream, "wan_mul_bf16");
}
// residual bookkeeping of the step-skipping caches (model.py:1967-1971, :2044-2062): x += residual, residual = x - ori
extern "C" int wan_add_bf16(const wan_bf16* a, const wan_bf16* b, wan_bf16* out, int64_t n, void* stream) {
  return binary_bf16<1>(a, b, out, n, stream, "wan_add_bf16");
}
extern "C" int wan_sub_bf16(const wan_bf16* a, const wan_bf16* b, wan_bf16* out, int64_t n, void* stream) {
  return binary_bf16<2>(a, b, out, n, stream, "wan_sub_bf16");
}
