// fp32 edge ops of the DiT: patch embedding (Conv3d k=s=(1,2,2) == a [L x Cin*4] x [Cin*4 x d]
// GEMM, K = 64 or 144) and the output head (Linear d->64 in fp32 + unpatchify scatter).
// Both are ~5e10 FLOP at 720p x 81f (0.001 of a forward) and run as plain LDS-tiled fp32 FMA
// kernels; the reference keeps them in fp32 (model.py:1330-1371) so no MFMA bf16 path applies.
#include "common.h"

int wan_ln_modulate_head(const bf16_t* x, bf16_t* out, const float* hmod, const bf16_t* e, int64_t rows,
                         int64_t rows_per_batch, int d, float eps, void* stream);

#define PE_TOK 32
// grid.x = ceil(ntok/32), block 256; token t (local) = tok0 + blockIdx.x*32 + i
// OutT = bf16_t: the bf16 plan (one rounding of the fp32 result); float: the mixed-precision plan's fp32 stream (model.py:1620-1631, no rounding)
template <typename OutT>
__global__ __launch_bounds__(256) void patch_embed_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                          const float* __restrict__ w, const float* __restrict__ bias,
                                                          OutT* __restrict__ out, int Cin, int Cy, int F, int H, int W,
                                                          int d, int64_t tok0, int64_t ntok, int64_t out_batch_stride,
                                                          int64_t x_batch_stride) {
  extern __shared__ float patch[];  // [PE_TOK][Kd]
  const int Ct = Cin + Cy;
  const int Kd = Ct * 4;
  const int Hg = H / 2, Wg = W / 2;
  const int b = blockIdx.y;
  const int64_t t0 = (int64_t)blockIdx.x * PE_TOK;
  const float* xb = x + (int64_t)b * x_batch_stride;
  for (int i = threadIdx.x; i < PE_TOK * Kd; i += 256) {
    const int tk = i / Kd, kk = i - tk * Kd;
    const int64_t tl = t0 + tk;
    float val = 0.f;
    if (tl < ntok) {
      const int64_t tg = tok0 + tl;
      const int ww = (int)(tg % Wg);
      const int hh = (int)((tg / Wg) % Hg);
      const int f = (int)(tg / ((int64_t)Wg * Hg));
      const int c = kk >> 2, qq = (kk >> 1) & 1, rr = kk & 1;
      const int64_t sp = ((int64_t)f * H + (2 * hh + qq)) * W + (2 * ww + rr);
      val = (c < Cin) ? xb[(int64_t)c * F * H * W + sp] : y[(int64_t)(c - Cin) * F * H * W + sp];
    }
    patch[i] = val;
  }
  __syncthreads();
  for (int n = threadIdx.x; n < d; n += 256) {
    float acc[PE_TOK];
    const float bn = bias[n];
#pragma unroll
    for (int tk = 0; tk < PE_TOK; ++tk) acc[tk] = bn;
    const float* wr = w + (int64_t)n * Kd;
    for (int kk = 0; kk < Kd; ++kk) {
      const float wv = wr[kk];
#pragma unroll
      for (int tk = 0; tk < PE_TOK; ++tk) acc[tk] += patch[tk * Kd + kk] * wv;
    }
#pragma unroll
    for (int tk = 0; tk < PE_TOK; ++tk) {
      const int64_t tl = t0 + tk;
      if (tl < ntok) {
        if constexpr (sizeof(OutT) == 4) out[(int64_t)b * out_batch_stride + tl * d + n] = acc[tk];
        else out[(int64_t)b * out_batch_stride + tl * d + n] = f2bf(acc[tk]);
      }
    }
  }
}

template <typename OutT>
static int patch_embed_launch(const float* x, const float* y, const float* w, const float* bias, OutT* out, int B, int Cin,
                              int Cy, int F, int H, int W, int d, int64_t tok0, int64_t ntok, void* stream) {
  WAN_REQUIRE(x && w && bias && out, "wan_patch_embed: null pointer");
  WAN_REQUIRE(H % 2 == 0 && W % 2 == 0, "wan_patch_embed: H, W must be even (patch 1x2x2)");
  WAN_REQUIRE(Cy == 0 || y != nullptr, "wan_patch_embed: y missing");
  const int Kd = (Cin + Cy) * 4;
  const size_t shm = (size_t)PE_TOK * Kd * sizeof(float);
  WAN_REQUIRE(shm <= 64 * 1024, "wan_patch_embed: too many input channels");
  if (ntok == 0) return 0;
  dim3 grid((unsigned)((ntok + PE_TOK - 1) / PE_TOK), (unsigned)B);
  hipLaunchKernelGGL(patch_embed_kernel<OutT>, grid, dim3(256), shm, as_stream(stream), x, y, w, bias, out, Cin, Cy, F, H, W,
                     d, tok0, ntok, ntok * (int64_t)d, (int64_t)Cin * F * H * W);
  WAN_LAUNCH_CHECK();
  return 0;
}
int wan_patch_embed_range(const float* x, const float* y, const float* w, const float* bias, bf16_t* out, int B, int Cin,
                          int Cy, int F, int H, int W, int d, int64_t tok0, int64_t ntok, void* stream) {
  return patch_embed_launch<bf16_t>(x, y, w, bias, out, B, Cin, Cy, F, H, W, d, tok0, ntok, stream);
}
// the mixed-precision plan's form (csrc/mixed_ops.hip wan_mx_patch_embed): the same tile kernel, fp32 rows out
int wan_patch_embed_f32_range(const float* x, const float* y, const float* w, const float* bias, float* out, int Cin, int Cy, int F, int H, int W,
                              int d, int64_t tok0, int64_t ntok, void* stream) {
  return patch_embed_launch<float>(x, y, w, bias, out, 1, Cin, Cy, F, H, W, d, tok0, ntok, stream);
}

extern "C" int wan_patch_embed(const float* x, const float* y, const float* w, const float* bias, wan_bf16* out, int B,
                               int Cin, int Cy, int F, int H, int W, int d, void* stream) {
  return wan_patch_embed_range(x, y, w, bias, out, B, Cin, Cy, F, H, W, d, 0, (int64_t)F * (H / 2) * (W / 2), stream);
}

// ---- head GEMM: out[tok][j] = bias[j] + sum_k xm[tok][k] * w[j][k]; 64 tokens x 64 outputs per block, grid.z covers
// nout = 4 * out_dim outputs in chunks of 64 (64 for the 16-channel latents, 192 for the 48-channel ti2v 5B model) ----
#define HD_KC 64
// InT = bf16_t: the bf16 plan's modulated head input; float: the mixed-precision plan's (unrounded)
template <typename InT>
__global__ __launch_bounds__(256) void head_gemm_kernel(const InT* __restrict__ xm, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ out, int d,
                                                        int F, int Hg, int Wg, int nout, int64_t tok0, int64_t ntok,
                                                        int64_t rows_per_batch_local, int token_major_out) {
  __shared__ float xs[64][HD_KC + 1];
  __shared__ float ws[HD_KC][64 + 4];  // [k][j]
  const int tid = threadIdx.x;
  const int tk = tid >> 2, jq = tid & 3;
  const int b = blockIdx.y;
  const int j0 = blockIdx.z * 64;
  const int64_t t0 = (int64_t)blockIdx.x * 64;
  const InT* xb = xm + (int64_t)b * rows_per_batch_local * d;
  float acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  for (int k0 = 0; k0 < d; k0 += HD_KC) {
    for (int i = tid; i < 64 * HD_KC; i += 256) {
      const int r = i / HD_KC, c = i - r * HD_KC;
      const int64_t tl = t0 + r;
      float xv0 = 0.f;
      if (tl < ntok && k0 + c < d) {
        if constexpr (sizeof(InT) == 4) xv0 = xb[tl * d + k0 + c];
        else xv0 = bf2f(xb[tl * d + k0 + c]);
      }
      xs[r][c] = xv0;
      // w: r -> output j, c -> k
      ws[c][r] = (j0 + r < nout && k0 + c < d) ? w[(int64_t)(j0 + r) * d + k0 + c] : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int c = 0; c < HD_KC; ++c) {
      const float xv = xs[tk][c];
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] += xv * ws[c][jq * 16 + j];
    }
    __syncthreads();
  }
  const int64_t tl = t0 + tk;
  if (tl >= ntok) return;
  if (token_major_out) {
    // [B][ntok][nout] (sequence-parallel: gathered by the host, unpatchified afterwards)
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (j0 + jq * 16 + j < nout) out[((int64_t)b * ntok + tl) * nout + j0 + jq * 16 + j] = acc[j] + bias[j0 + jq * 16 + j];
    return;
  }
  // unpatchify 'fhwpqrc->cfphqwr' (model.py:2119-2121): j = (q*2+r)*C + c, C = nout/4
  const int C = nout / 4;
  const int64_t tg = tok0 + tl;
  const int ww = (int)(tg % Wg);
  const int hh = (int)((tg / Wg) % Hg);
  const int f = (int)(tg / ((int64_t)Wg * Hg));
  const int H = Hg * 2, W = Wg * 2;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int jj = j0 + jq * 16 + j;
    if (jj < nout) {
      const int c = jj % C, qr = jj / C;
      const int qq = qr >> 1, rr = qr & 1;
      out[(((int64_t)b * C + c) * F + f) * H * W + (int64_t)(2 * hh + qq) * W + (2 * ww + rr)] = acc[j] + bias[jj];
    }
  }
}

// unpatchify from a token-major [B][L][64] fp32 buffer
__global__ void unpatchify_kernel(const float* __restrict__ in, float* __restrict__ out, int F, int Hg, int Wg, int nout,
                                  int64_t L) {
  const int C = nout / 4;
  const int64_t total = L * nout;
  const int b = blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t tg = i / nout;
    const int jj = (int)(i - tg * nout);
    const int ww = (int)(tg % Wg);
    const int hh = (int)((tg / Wg) % Hg);
    const int f = (int)(tg / ((int64_t)Wg * Hg));
    const int c = jj % C, qr = jj / C;
    const int qq = qr >> 1, rr = qr & 1;
    const int H = Hg * 2, W = Wg * 2;
    out[(((int64_t)b * C + c) * F + f) * H * W + (int64_t)(2 * hh + qq) * W + (2 * ww + rr)] = in[(int64_t)b * total + i];
  }
}

int wan_head_range(const bf16_t* x, const float* hmod, const bf16_t* e, const float* w, const float* bias, bf16_t* tmp,
                   float* out, int B, int F, int Hg, int Wg, int d, float eps, int64_t tok0, int64_t ntok,
                   int token_major_out, int64_t e_rows_per_batch, int nout, void* stream) {
  WAN_REQUIRE(x && hmod && e && w && bias && tmp && out, "wan_head: null pointer");
  WAN_REQUIRE(nout >= 4 && nout % 4 == 0 && nout <= 1024, "wan_head: nout=%d must be 4 * out_dim", nout);
  const int64_t rows = (int64_t)B * ntok;
  // e_rows_per_batch: 0 = one e [1,d] for every row (the streams of a joint CFG pass share t); else e [n,d] with row r using
  // e[r / e_rows_per_batch] (per-frame timesteps: tokens per frame; or one e per batch element: ntok)
  int rc = wan_ln_modulate_head(x, tmp, hmod, e, rows, e_rows_per_batch > 0 ? e_rows_per_batch : (rows > 0 ? rows : 1), d, eps, stream);
  if (rc) return rc;
  if (ntok == 0) return 0;
  dim3 grid((unsigned)((ntok + 63) / 64), (unsigned)B, (unsigned)((nout + 63) / 64));
  hipLaunchKernelGGL(head_gemm_kernel<bf16_t>, grid, dim3(256), 0, as_stream(stream), (const bf16_t*)tmp, w, bias, out, d, F, Hg, Wg, nout, tok0,
                     ntok, ntok, token_major_out);
  WAN_LAUNCH_CHECK();
  return 0;
}
// the mixed-precision plan's head Linear (csrc/mixed_ops.hip wan_mx_head): fp32 rows in, token-major fp32 [ntok, nout] out
int wan_head_gemm_f32(const float* xm, const float* w, const float* bias, float* out, int64_t ntok, int d, int nout, void* stream) {
  if (ntok == 0) return 0;
  dim3 grid((unsigned)((ntok + 63) / 64), 1u, (unsigned)((nout + 63) / 64));
  hipLaunchKernelGGL(head_gemm_kernel<float>, grid, dim3(256), 0, as_stream(stream), xm, w, bias, out, d, 1, 1, 1, nout, (int64_t)0, ntok, ntok, 1);
  WAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int wan_head(const wan_bf16* x, const float* hmod, const wan_bf16* e, const float* w, const float* bias,
                        wan_bf16* tmp, float* out, int B, int F, int Hg, int Wg, int d, float eps, void* stream) {
  return wan_head_range(x, hmod, e, w, bias, tmp, out, B, F, Hg, Wg, d, eps, 0, (int64_t)F * Hg * Wg, 0, (int64_t)F * Hg * Wg, 64, stream);
}

extern "C" int wan_head_n(const wan_bf16* x, const float* hmod, const wan_bf16* e, const float* w, const float* bias,
                          wan_bf16* tmp, float* out, int B, int F, int Hg, int Wg, int d, float eps, int nout, void* stream) {
  return wan_head_range(x, hmod, e, w, bias, tmp, out, B, F, Hg, Wg, d, eps, 0, (int64_t)F * Hg * Wg, 0, (int64_t)F * Hg * Wg, nout, stream);
}

static int unpatchify_n(const float* in, float* out, int B, int F, int Hg, int Wg, int nout, void* stream) {
  WAN_REQUIRE(in && out, "wan_unpatchify: null pointer");
  WAN_REQUIRE(nout >= 4 && nout % 4 == 0, "wan_unpatchify: nout=%d must be 4 * out_dim", nout);
  const int64_t L = (int64_t)F * Hg * Wg;
  int blocks = (int)((L * nout + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) return 0;
  hipLaunchKernelGGL(unpatchify_kernel, dim3(blocks, B), dim3(256), 0, as_stream(stream), in, out, F, Hg, Wg, nout, L);
  WAN_LAUNCH_CHECK();
  return 0;
}
extern "C" int wan_unpatchify(const float* in, float* out, int B, int F, int Hg, int Wg, void* stream) {
  return unpatchify_n(in, out, B, F, Hg, Wg, 64, stream);
}
extern "C" int wan_unpatchify_n(const float* in, float* out, int B, int F, int Hg, int Wg, int nout, void* stream) {
  return unpatchify_n(in, out, B, F, Hg, Wg, nout, stream);
}
