// The Wan2.1 VAE in an fp32 plan (round 4): what the reference runs with `vae_precision` "32" (wgp.py:4038 -> WanVAE(dtype =
// torch.float32), models/wan/modules/vae.py) -- fp32 weights, fp32 activations, fp32 accumulation, no 16-bit rounding point
// anywhere.  Served as an OPTION (WanVAEHIP(dtype=torch.float32)); the fp16 plan of vae_ops.hip / vae_graph.hip stays the default,
// as it is the reference's default on a GPU.  Why it exists at all: the fp16 plan is chaotic across summation orders (a 720p decode
// of the same latents by two correct fp16 implementations agrees on ~93 % of the output bytes), so "the VAE's
// integer pixel output matches the reference" can only be shown near-exactly in fp32.
//
// Plain fp32 FMA kernels, no matrix cores: the fp32 MFMA runs at the vector rate anyway (MI355X_MICROARCH.md: 1/16 of bf16) and a
// 6.4e14-FLOP decode at 720p x 81 frames is a minute-class job in this plan on any path -- these kernels are written to be
// obviously the reference's arithmetic (same operations, fp32 throughout, K summed in tap-major / channel-minor order), not fast:
//   wan_vae_conv3d_f32        CausalConv3d / Resample convs (vae.py:43-82, :124-141, :186-189): implicit GEMM, 64 pixels x 64
//                             output channels per workgroup, K in steps of 16 through LDS, 4 x 4 outputs per thread; the same
//                             gather semantics as conv3d_f16_kernel (2-frame causal cache, zero padding, stride 2 with
//                             ZeroPad2d((0,1,0,1)), fused nearest-exact 2x upsample, time_conv channel -> frame interleave)
//   wan_vae_rmsnorm_silu_f32  RMS_norm (+ SiLU) (vae.py:97-103, :246)
//   wan_gemm_f32              C = scale * A B^T (+ bias) or A B: the to_qkv / q k^T / p v products of AttentionBlock (vae.py:294-315)
//   wan_vae_softmax_f32       its row softmax
//   wan_vae_pack_f32 / wan_vae_unpack_f32   [C,T,H,W] <-> channels-last [T,H,W,Cp], with the latent (de)normalisation
#include "common.h"

namespace {

struct ConvF {
  const float *x, *cache, *w, *bias, *res;
  float* out;
  int Tin, Hin, Win, Cin, Tout, Hout, Wout, Cout, KT, KH, KW, st_t, st_s, front, pad_s, ups, ncache, interleave;
  int64_t M;
  int K, ldw;  // K = taps * Cin; weight rows are ldw floats apart
};

constexpr int FT = 64, FK = 16;  // tile: 64 pixels x 64 output channels, K step 16

__global__ __launch_bounds__(256) void conv3d_f32_kernel(ConvF p) {
  __shared__ float As[FK][FT + 4], Bs[FK][FT + 4];
  const int tid = threadIdx.x;
  const int tiles_x = (p.Cout + FT - 1) / FT;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int64_t y0 = (int64_t)ty * FT;
  const int x0 = tx * FT;
  const int HWo = p.Hout * p.Wout;
  const int Heff = p.ups ? 2 * p.Hin : p.Hin, Weff = p.ups ? 2 * p.Win : p.Win;
  const int64_t frame_in = (int64_t)p.Hin * p.Win * p.Cin;

  // this thread's slot of the A tile: pixel `arow`, four consecutive k (= four consecutive input channels of one tap: Cin % 16 == 0)
  const int arow = tid >> 2, ak = (tid & 3) * 4;
  int64_t pp = y0 + arow;
  const bool pvalid = pp < p.M;
  if (!pvalid) pp = p.M - 1;
  const int to = (int)(pp / HWo);
  const int rem = (int)(pp - (int64_t)to * HWo);
  const int ho = rem / p.Wout, wo = rem - ho * p.Wout;
  const int t_in0 = to * p.st_t - p.front, h_in0 = ho * p.st_s - p.pad_s, w_in0 = wo * p.st_s - p.pad_s;
  // ... and of the B tile: output channel `brow`, the same four k
  int bco = x0 + arow;
  if (bco > p.Cout - 1) bco = p.Cout - 1;
  const float* wrow = p.w + (int64_t)bco * p.ldw + ak;

  float acc[4][4] = {};
  const int py = (tid >> 4) * 4, px = (tid & 15) * 4;  // this thread's 4 pixels x 4 channels of the tile
  for (int k0 = 0; k0 < p.K; k0 += FK) {
    const int tap = k0 / p.Cin, c0 = k0 - tap * p.Cin;
    const int kw = tap % p.KW, t2 = tap / p.KW;
    const int kh = t2 % p.KH, kt = t2 / p.KH;
    const int ti = t_in0 + kt, hi = h_in0 + kh, wi = w_in0 + kw;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((unsigned)hi < (unsigned)Heff && (unsigned)wi < (unsigned)Weff && ti >= -p.ncache && ti < p.Tin) {
      const int hs = p.ups ? hi >> 1 : hi, ws = p.ups ? wi >> 1 : wi;  // nearest-exact 2x: source pixel floor(i / 2) (vae.py:105-111)
      const float* src = (ti >= 0 ? p.x + (int64_t)ti * frame_in : p.cache + (int64_t)(ti + 2) * frame_in) + ((int64_t)hs * p.Win + ws) * p.Cin + c0 + ak;
      a = *reinterpret_cast<const float4*>(src);
    }
    const float4 b = *reinterpret_cast<const float4*>(wrow + k0);
    __syncthreads();
    As[ak + 0][arow] = a.x; As[ak + 1][arow] = a.y; As[ak + 2][arow] = a.z; As[ak + 3][arow] = a.w;
    Bs[ak + 0][arow] = b.x; Bs[ak + 1][arow] = b.y; Bs[ak + 2][arow] = b.z; Bs[ak + 3][arow] = b.w;
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < FK; ++kk) {
      const float4 av = *reinterpret_cast<const float4*>(&As[kk][py]);
      const float4 bv = *reinterpret_cast<const float4*>(&Bs[kk][px]);
      const float ar[4] = {av.x, av.y, av.z, av.w}, br[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_fmaf(ar[i], br[j], acc[i][j]);
    }
  }
  const int C2 = p.Cout >> 1;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t q = y0 + py + i;
    if (q >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int co = x0 + px + j;
      if (co >= p.Cout) continue;
      int64_t oidx;
      if (p.interleave) {  // time_conv output [T,H,W,2*C2] -> [2T,H,W,C2]: channel half s goes to frame 2t+s (vae.py:186-189)
        const int tq = (int)(q / HWo);
        const int rq = (int)(q - (int64_t)tq * HWo);
        const int s = co >= C2;
        oidx = ((int64_t)(2 * tq + s) * HWo + rq) * C2 + (co - s * C2);
      } else {
        oidx = q * p.Cout + co;
      }
      float o = acc[i][j] + (p.bias ? p.bias[co] : 0.f);
      if (p.res) o = o + p.res[oidx];
      p.out[oidx] = o;
    }
  }
}

// one wave per pixel: C <= 1024 channels, lane l holds channels l, l + 64, ...
__global__ __launch_bounds__(256) void vae_rmsnorm_f32_kernel(const float* __restrict__ x, float* __restrict__ out, const float* __restrict__ gamma,
                                                              int64_t npix, int C, int silu) {
  const int lane = threadIdx.x & 63;
  const int64_t pix = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pix >= npix) return;
  const float* row = x + pix * C;
  float v[16];
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int c = lane + 64 * k;
    v[k] = c < C ? row[c] : 0.f;
    ss += v[k] * v[k];
  }
  ss = wave_sum(ss);
  const float inv = sqrtf((float)C) / fmaxf(sqrtf(ss), 1e-12f);  // F.normalize(x, dim = channels) * sqrt(C) (vae.py:97-103)
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int c = lane + 64 * k;
    if (c < C) {
      float y = v[k] * inv * gamma[c];
      if (silu) y = y / (1.0f + expf(-y));
      out[pix * C + c] = y;
    }
  }
}

// C[m][n] = scale * sum_k A[m][k] * (BT ? B[n][k] : B[k][n]) (+ bias[n]); 64 x 64 tiles, K step 16
template <bool BT>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
                                                       const float* __restrict__ bias, float* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                       float scale) {
  __shared__ float As[FK][FT + 4], Bs[FK][FT + 4];
  const int tid = threadIdx.x;
  const int tiles_x = (N + FT - 1) / FT;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int y0 = ty * FT, x0 = tx * FT;
  float acc[4][4] = {};
  const int py = (tid >> 4) * 4, px = (tid & 15) * 4;
  for (int k0 = 0; k0 < K; k0 += FK) {
    __syncthreads();
    {  // A tile: row tid >> 2, k (tid & 3) * 4 .. + 3
      const int r = tid >> 2, kk = (tid & 3) * 4;
      const int m = min(y0 + r, M - 1);
#pragma unroll
      for (int j = 0; j < 4; ++j) As[kk + j][r] = (k0 + kk + j < K) ? A[(int64_t)m * lda + k0 + kk + j] : 0.f;
    }
    if (BT) {
      const int r = tid >> 2, kk = (tid & 3) * 4;
      const int n = min(x0 + r, N - 1);
#pragma unroll
      for (int j = 0; j < 4; ++j) Bs[kk + j][r] = (k0 + kk + j < K) ? B[(int64_t)n * ldb + k0 + kk + j] : 0.f;
    } else {  // B[k][n]: k = tid >> 4, n = (tid & 15) * 4 .. + 3
      const int kk = tid >> 4, c = (tid & 15) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = min(x0 + c + j, N - 1);
        Bs[kk][c + j] = (k0 + kk < K) ? B[(int64_t)(k0 + kk) * ldb + n] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < FK; ++kk) {
      const float4 av = *reinterpret_cast<const float4*>(&As[kk][py]);
      const float4 bv = *reinterpret_cast<const float4*>(&Bs[kk][px]);
      const float ar[4] = {av.x, av.y, av.z, av.w}, br[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_fmaf(ar[i], br[j], acc[i][j]);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = y0 + py + i, n = x0 + px + j;
      if (m < M && n < N) C[(int64_t)m * ldc + n] = acc[i][j] * scale + (bias ? bias[n] : 0.f);
    }
}

__global__ __launch_bounds__(256) void vae_softmax_f32_kernel(float* __restrict__ S, int L, int64_t ld) {
  __shared__ float red[8];
  float* s = S + (int64_t)blockIdx.x * ld;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float m = -INFINITY;
  for (int i = threadIdx.x; i < L; i += 256) m = fmaxf(m, s[i]);
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int i = threadIdx.x; i < L; i += 256) {
    const float e = expf(s[i] - m);
    s[i] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  for (int i = threadIdx.x; i < L; i += 256) s[i] *= inv;
}

__global__ void vae_pack_f32_kernel(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ mul, const float* __restrict__ add,
                                    int C, int Cp, int64_t thw) {
  const int64_t total = thw * Cp;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pix = i / Cp;
    const int c = (int)(i - pix * Cp);
    float v = 0.f;
    if (c < C) {
      v = in[(int64_t)c * thw + pix];
      if (mul) v = v * mul[c] + add[c];
    }
    out[i] = v;
  }
}
__global__ void vae_unpack_f32_kernel(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ sub, const float* __restrict__ mul,
                                      int C, int Cs, int64_t thw) {
  const int64_t total = thw * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i / thw);
    const int64_t pix = i - (int64_t)c * thw;
    float v = in[pix * Cs + c];
    if (sub) v = (v - sub[c]) * mul[c];
    out[i] = v;
  }
}

}  // namespace

extern "C" int wan_vae_conv3d_f32(const float* x, const float* cache, const float* w, int64_t ldw, const float* bias, const float* res, float* out,
                                  int Tin, int Hin, int Win, int Cin, int Tout, int Hout, int Wout, int Cout, int KT, int KH, int KW, int st_t,
                                  int st_s, int front, int pad_s, int ups, int interleave, void* stream) {
  WAN_REQUIRE(x && w && out, "wan_vae_conv3d_f32: null pointer");
  WAN_REQUIRE(Cin % 16 == 0 && ldw % 4 == 0 && ldw >= (int64_t)KT * KH * KW * Cin, "wan_vae_conv3d_f32: Cin=%d must be a multiple of 16 (pad), ldw a multiple of 4 covering K", Cin);
  WAN_REQUIRE(!interleave || (Cout % 2 == 0 && res == nullptr), "wan_vae_conv3d_f32: bad interleave use");
  WAN_REQUIRE(front >= 0 && front <= 2, "wan_vae_conv3d_f32: front must be 0..2");
  WAN_REQUIRE((((uintptr_t)x | (uintptr_t)w | (uintptr_t)(cache ? cache : x)) & 15) == 0, "wan_vae_conv3d_f32: pointers must be 16-byte aligned");
  ConvF p;
  p.x = x; p.cache = cache ? cache : x; p.w = w; p.bias = bias; p.res = res; p.out = out;
  p.Tin = Tin; p.Hin = Hin; p.Win = Win; p.Cin = Cin; p.Tout = Tout; p.Hout = Hout; p.Wout = Wout; p.Cout = Cout;
  p.KT = KT; p.KH = KH; p.KW = KW; p.st_t = st_t; p.st_s = st_s; p.front = front; p.pad_s = pad_s; p.ups = ups;
  p.ncache = cache ? 2 : 0;
  p.interleave = interleave;
  p.M = (int64_t)Tout * Hout * Wout;
  p.K = KT * KH * KW * Cin;
  p.ldw = (int)ldw;
  if (p.M == 0) return 0;
  const int64_t tiles = ((p.M + FT - 1) / FT) * ((Cout + FT - 1) / FT);
  WAN_REQUIRE(tiles < ((int64_t)1 << 31), "wan_vae_conv3d_f32: too many tiles");
  hipLaunchKernelGGL(conv3d_f32_kernel, dim3((unsigned)tiles), dim3(256), 0, as_stream(stream), p);
  WAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int wan_vae_rmsnorm_silu_f32(const float* x, float* out, const float* gamma, int64_t npix, int C, int silu, void* stream) {
  WAN_REQUIRE(x && out && gamma && C >= 1 && C <= 1024, "wan_vae_rmsnorm_silu_f32: bad args (C=%d <= 1024)", C);
  if (npix == 0) return 0;
  hipLaunchKernelGGL(vae_rmsnorm_f32_kernel, dim3((unsigned)((npix + 3) / 4)), dim3(256), 0, as_stream(stream), x, out, gamma, npix, C, silu);
  WAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int wan_gemm_f32(const float* A, int64_t lda, const float* B, int64_t ldb, int b_transposed, const float* bias, float* C, int64_t ldc,
                            int M, int N, int K, float scale, void* stream) {
  WAN_REQUIRE(A && B && C && M >= 0 && N >= 0 && K >= 1, "wan_gemm_f32: bad args");
  if (M == 0 || N == 0) return 0;
  const int64_t tiles = (int64_t)((M + FT - 1) / FT) * ((N + FT - 1) / FT);
  WAN_REQUIRE(tiles < ((int64_t)1 << 31), "wan_gemm_f32: too many tiles");
  if (b_transposed) hipLaunchKernelGGL((gemm_f32_kernel<true>), dim3((unsigned)tiles), dim3(256), 0, as_stream(stream), A, lda, B, ldb, bias, C, ldc, M, N, K, scale);
  else hipLaunchKernelGGL((gemm_f32_kernel<false>), dim3((unsigned)tiles), dim3(256), 0, as_stream(stream), A, lda, B, ldb, bias, C, ldc, M, N, K, scale);
  WAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int wan_vae_softmax_f32(float* S, int64_t rows, int L, int64_t ld, void* stream) {
  WAN_REQUIRE(S && ld >= L && L >= 1, "wan_vae_softmax_f32: bad args");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(vae_softmax_f32_kernel, dim3((unsigned)rows), dim3(256), 0, as_stream(stream), S, L, ld);
  WAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int wan_vae_pack_f32(const float* in, float* out, const float* mul, const float* add, int C, int Cp, int64_t thw, void* stream) {
  WAN_REQUIRE(in && out && Cp >= C && (mul == nullptr) == (add == nullptr), "wan_vae_pack_f32: bad args");
  int blocks = (int)((thw * Cp + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  if (blocks == 0) return 0;
  hipLaunchKernelGGL(vae_pack_f32_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), in, out, mul, add, C, Cp, thw);
  WAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int wan_vae_unpack_f32(const float* in, float* out, const float* sub, const float* mul, int C, int Cs, int64_t thw, void* stream) {
  WAN_REQUIRE(in && out && Cs >= C, "wan_vae_unpack_f32: bad args");
  int blocks = (int)((thw * C + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  if (blocks == 0) return 0;
  hipLaunchKernelGGL(vae_unpack_f32_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), in, out, sub, mul, C, Cs, thw);
  WAN_LAUNCH_CHECK();
  return 0;
}
