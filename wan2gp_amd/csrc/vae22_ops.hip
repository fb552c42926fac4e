// Wan2.2 VAE (5B ti2v: z = 48, stride (4,16,16)) -- the four data-movement pieces that models/wan/modules/vae2_2.py adds to
// the Wan2.1 VAE graph.  Convolutions, RMS_norm+SiLU and the attention block are the
// vae_ops.hip kernels; the layer graph is host code (wan2gp_amd/vae22.py).  Activations are fp16 channels-last [T,H,W,C].
//
//   wan_vae22_patchify     patchify(x, 2) (vae2_2.py:299-315) fused with the fp32 -> fp16 channels-last packing:
//                          out[t,h,w,(c*2+r)*2+q] = video[c,t,2h+q,2w+r], channels 12..Cp-1 zero.
//   wan_vae22_to_video     unpatchify (:318-332) fused with _vae_float_to_cpu_uint8 (vae.py:18-20) / the fp32 video layout.
//   wan_vae22_avgdown_add  io += AvgDown3D(x) (:354-386; Down_ResidualBlock.forward :466-471): zero frames in FRONT up to a
//                          multiple of ft, the (ft,fs,fs) block folded into the channels channel-major, groups of
//                          C*ft*fs*fs/Co consecutive folded channels averaged.  mean in fp32, rounded to fp16, added, rounded.
//   wan_vae22_dupup_add    io += DupUp3D(x, first_chunk) (:409-431; Up_ResidualBlock.forward :508-516).
// All four are pure gathers bound by HBM; one thread per 8 output channels (16-byte stores).
// Round 6: every piece also in fp32 (`vae_precision` "32", wgp.py:4038 -> Wan2_2_VAE(dtype=torch.float32)): the same gathers on fp32
// channels-last activations, no 16-bit rounding point (wan_vae22_*_f32; the layer graph runs on csrc/vae_f32.hip's kernels).
#include "common.h"

template <bool F32> struct V22El { using type = uint16_t; };
template <> struct V22El<true> { using type = float; };
template <bool F32>
__device__ __forceinline__ void v22_load8(const typename V22El<F32>::type* p, float (&f)[8]) {
  if constexpr (F32) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
  } else {
    unpack8t<true>(*reinterpret_cast<const uint4*>(p), f);
  }
}
template <bool F32>
__device__ __forceinline__ void v22_store8(typename V22El<F32>::type* p, const float (&f)[8]) {
  if constexpr (F32) {
    *reinterpret_cast<float4*>(p) = float4{f[0], f[1], f[2], f[3]};
    *reinterpret_cast<float4*>(p + 4) = float4{f[4], f[5], f[6], f[7]};
  } else {
    *reinterpret_cast<uint4*>(p) = pack8t<true>(f);
  }
}
template <bool F32>
__device__ __forceinline__ float v22_ld(const typename V22El<F32>::type* p) {
  if constexpr (F32) return *p;
  else return h2f(*p);
}

template <bool F32>
__global__ void vae22_patchify_kernel(const float* __restrict__ v, typename V22El<F32>::type* __restrict__ out, int T, int H, int W, int Cp) {
  const int h2 = H >> 1, w2 = W >> 1;
  const int64_t npix = (int64_t)T * h2 * w2;
  const int cpp = Cp >> 3;  // 8-channel groups per pixel
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix * cpp; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pix = i / cpp;
    const int cg = (int)(i - pix * cpp);
    const int t = (int)(pix / ((int64_t)h2 * w2));
    const int rem = (int)(pix - (int64_t)t * h2 * w2);
    const int hh = rem / w2, ww = rem - hh * w2;
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ch = cg * 8 + j;
      float val = 0.f;
      if (ch < 12) {
        const int c = ch >> 2, r = (ch >> 1) & 1, q = ch & 1;
        val = v[(((int64_t)c * T + t) * H + (2 * hh + q)) * W + (2 * ww + r)];
      }
      f[j] = val;
    }
    v22_store8<F32>(out + pix * Cp + cg * 8, f);
  }
}

__global__ void vae22_to_video_kernel(const float* __restrict__ y, uint8_t* __restrict__ u8, float* __restrict__ f32, int Ti,
                                      int h, int w, int Ttot, int t0) {
  const int H = 2 * h, W = 2 * w;
  const int64_t total = (int64_t)3 * Ti * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int X = (int)(i % W);
    const int Y = (int)((i / W) % H);
    const int t = (int)((i / ((int64_t)W * H)) % Ti);
    const int c = (int)(i / ((int64_t)W * H * Ti));
    const int ch = (c * 2 + (X & 1)) * 2 + (Y & 1);
    const float val = y[(((int64_t)t * h + (Y >> 1)) * w + (X >> 1)) * 12 + ch];
    const int64_t o = (((int64_t)c * Ttot + t0 + t) * H + Y) * W + X;
    if (f32) f32[o] = val;
    if (u8) {
      float q = fminf(fmaxf(val, -1.0f), 1.0f);
      q = rintf((q + 1.0f) * 127.5f);  // round half to even, like torch.round_
      u8[o] = (uint8_t)fminf(fmaxf(q, 0.0f), 255.0f);
    }
  }
}

template <bool F32>
__global__ void vae22_avgdown_add_kernel(const typename V22El<F32>::type* __restrict__ x, typename V22El<F32>::type* __restrict__ io, int T, int H, int W, int C,
                                         int Co, int ft, int fs) {
  const int pad_t = (ft - T % ft) % ft;
  const int To = (T + pad_t) / ft, Ho = H / fs, Wo = W / fs;
  const int factor = ft * fs * fs;
  const int g = C * factor / Co;  // folded channels per output channel
  const int cgp = Co >> 3;
  const int64_t total = (int64_t)To * Ho * Wo * cgp;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pix = i / cgp;
    const int cg = (int)(i - pix * cgp);
    const int to = (int)(pix / ((int64_t)Ho * Wo));
    const int rem = (int)(pix - (int64_t)to * Ho * Wo);
    const int ho = rem / Wo, wo = rem - ho * Wo;
    float cur[8];
    typename V22El<F32>::type* dst = io + pix * Co + cg * 8;
    v22_load8<F32>(dst, cur);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int o = cg * 8 + j;
      float sum = 0.f;
      for (int k = 0; k < g; ++k) {
        const int cf = o * g + k;  // folded channel = c*factor + it*fs*fs + ih*fs + iw
        const int c = cf / factor, blk = cf - c * factor;
        const int it = blk / (fs * fs), ih = (blk / fs) % fs, iw = blk % fs;
        const int t = to * ft + it - pad_t;
        if (t >= 0) sum += v22_ld<F32>(x + (((int64_t)t * H + ho * fs + ih) * W + wo * fs + iw) * C + c);
      }
      cur[j] = cur[j] + (F32 ? sum / (float)g : rnd16<true>(sum / (float)g));
    }
    v22_store8<F32>(dst, cur);
  }
}

template <bool F32>
__global__ void vae22_dupup_add_kernel(const typename V22El<F32>::type* __restrict__ x, typename V22El<F32>::type* __restrict__ io, int T, int H, int W, int C,
                                       int Co, int ft, int fs, int drop) {
  const int To = T * ft - drop, Ho = H * fs, Wo = W * fs;
  const int factor = ft * fs * fs;
  const int rep = Co * factor / C;
  const int cgp = Co >> 3;
  const int64_t total = (int64_t)To * Ho * Wo * cgp;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pix = i / cgp;
    const int cg = (int)(i - pix * cgp);
    const int to = (int)(pix / ((int64_t)Ho * Wo));
    const int rem = (int)(pix - (int64_t)to * Ho * Wo);
    const int ho = rem / Wo, wo = rem - ho * Wo;
    const int tf = to + drop;  // frame index before the first-chunk trim
    const int t = tf / ft, it = tf - t * ft;
    const int hh = ho / fs, ih = ho - hh * fs, ww = wo / fs, iw = wo - ww * fs;
    const typename V22El<F32>::type* src = x + (((int64_t)t * H + hh) * W + ww) * C;
    float cur[8];
    typename V22El<F32>::type* dst = io + pix * Co + cg * 8;
    v22_load8<F32>(dst, cur);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int o = cg * 8 + j;
      cur[j] += v22_ld<F32>(src + (o * factor + it * fs * fs + ih * fs + iw) / rep);
    }
    v22_store8<F32>(dst, cur);
  }
}

static inline int v22_blocks(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

template <bool F32>
static int v22_patchify(const float* video, typename V22El<F32>::type* out, int T, int H, int W, int Cp, void* stream) {
  WAN_REQUIRE(video && out, "wan_vae22_patchify: null pointer");
  WAN_REQUIRE(T >= 1 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0 && Cp >= 16 && Cp % 8 == 0,
              "wan_vae22_patchify: bad shape T=%d H=%d W=%d Cp=%d (even H, W; Cp >= 16, multiple of 8)", T, H, W, Cp);
  const int64_t n = (int64_t)T * (H / 2) * (W / 2) * (Cp / 8);
  hipLaunchKernelGGL(vae22_patchify_kernel<F32>, dim3(v22_blocks(n)), dim3(256), 0, as_stream(stream), video, out, T, H, W, Cp);
  WAN_LAUNCH_CHECK();
  return 0;
}
extern "C" int wan_vae22_patchify(const float* video, uint16_t* out, int T, int H, int W, int Cp, void* stream) {
  return v22_patchify<false>(video, out, T, H, W, Cp, stream);
}
extern "C" int wan_vae22_patchify_f32(const float* video, float* out, int T, int H, int W, int Cp, void* stream) {
  return v22_patchify<true>(video, out, T, H, W, Cp, stream);
}

extern "C" int wan_vae22_to_video(const float* y, uint8_t* u8, float* f32, int Ti, int h, int w, int Ttot, int t0, void* stream) {
  WAN_REQUIRE(y && (u8 || f32), "wan_vae22_to_video: null pointer");
  WAN_REQUIRE(Ti >= 0 && h >= 1 && w >= 1 && t0 >= 0 && t0 + Ti <= Ttot, "wan_vae22_to_video: frames [%d, %d) outside 0..%d", t0, t0 + Ti, Ttot);
  if (Ti == 0) return 0;
  const int64_t n = (int64_t)3 * Ti * 4 * h * w;
  hipLaunchKernelGGL(vae22_to_video_kernel, dim3(v22_blocks(n)), dim3(256), 0, as_stream(stream), y, u8, f32, Ti, h, w, Ttot, t0);
  WAN_LAUNCH_CHECK();
  return 0;
}

template <bool F32>
static int v22_avgdown_add(const typename V22El<F32>::type* x, typename V22El<F32>::type* io, int T, int H, int W, int C, int Co, int ft, int fs, void* stream) {
  WAN_REQUIRE(x && io, "wan_vae22_avgdown_add: null pointer");
  WAN_REQUIRE(T >= 1 && (ft == 1 || ft == 2) && (fs == 1 || fs == 2) && H % fs == 0 && W % fs == 0 && Co % 8 == 0 &&
                  (C * ft * fs * fs) % Co == 0,
              "wan_vae22_avgdown_add: bad shape T=%d H=%d W=%d C=%d Co=%d ft=%d fs=%d", T, H, W, C, Co, ft, fs);
  const int To = (T + (ft - T % ft) % ft) / ft;
  const int64_t n = (int64_t)To * (H / fs) * (W / fs) * (Co / 8);
  hipLaunchKernelGGL(vae22_avgdown_add_kernel<F32>, dim3(v22_blocks(n)), dim3(256), 0, as_stream(stream), x, io, T, H, W, C, Co, ft, fs);
  WAN_LAUNCH_CHECK();
  return 0;
}
extern "C" int wan_vae22_avgdown_add(const uint16_t* x, uint16_t* io, int T, int H, int W, int C, int Co, int ft, int fs, void* stream) {
  return v22_avgdown_add<false>(x, io, T, H, W, C, Co, ft, fs, stream);
}
extern "C" int wan_vae22_avgdown_add_f32(const float* x, float* io, int T, int H, int W, int C, int Co, int ft, int fs, void* stream) {
  return v22_avgdown_add<true>(x, io, T, H, W, C, Co, ft, fs, stream);
}

template <bool F32>
static int v22_dupup_add(const typename V22El<F32>::type* x, typename V22El<F32>::type* io, int T, int H, int W, int C, int Co, int ft, int fs, int first_chunk,
                         void* stream) {
  WAN_REQUIRE(x && io, "wan_vae22_dupup_add: null pointer");
  WAN_REQUIRE(T >= 1 && (ft == 1 || ft == 2) && (fs == 1 || fs == 2) && Co % 8 == 0 && (Co * ft * fs * fs) % C == 0,
              "wan_vae22_dupup_add: bad shape T=%d C=%d Co=%d ft=%d fs=%d", T, C, Co, ft, fs);
  const int drop = first_chunk ? ft - 1 : 0;
  const int64_t n = (int64_t)(T * ft - drop) * (H * fs) * (W * fs) * (Co / 8);
  hipLaunchKernelGGL(vae22_dupup_add_kernel<F32>, dim3(v22_blocks(n)), dim3(256), 0, as_stream(stream), x, io, T, H, W, C, Co, ft, fs, drop);
  WAN_LAUNCH_CHECK();
  return 0;
}
extern "C" int wan_vae22_dupup_add(const uint16_t* x, uint16_t* io, int T, int H, int W, int C, int Co, int ft, int fs, int first_chunk, void* stream) {
  return v22_dupup_add<false>(x, io, T, H, W, C, Co, ft, fs, first_chunk, stream);
}
extern "C" int wan_vae22_dupup_add_f32(const float* x, float* io, int T, int H, int W, int C, int Co, int ft, int fs, int first_chunk, void* stream) {
  return v22_dupup_add<true>(x, io, T, H, W, C, Co, ft, fs, first_chunk, stream);
}
