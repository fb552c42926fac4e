// The row / edge kernels of the reference's MIXED-PRECISION transformer plan (`mixed_precision_transformer`: wgp.py:4039 server
// setting "mixed_precision" -> any2video.py:190 -> lock_layers_dtypes(torch.float32), models/wan/modules/model.py:1330-1371).
// With the time MLP, the time projection and every block's norm3 held in fp32, the modulation dtype (model.py:1545) is fp32 and, by type
// promotion, the residual stream x, e / e0, every AdaLN modulate and every gated residual run in fp32; each Linear / attention still sees
// bf16 (ONE rounding, `.to(attention_dtype)`, model.py:650,665,692) and returns bf16 that is widened back (:654,668,708).
// An option, not the default plan: these kernels are plain (one wave per token row, rows re-read from L2 for the second and third pass;
// GEMM epilogues are separate passes) -- the bf16 plan's fused kernels (elementwise.hip, gemm256m.hip) are untouched.
//   wan_mx_ln_modulate    norm1 / norm2 + modulate   model.py:634-638, :686-692   x fp32 -> bf16
//   wan_mx_ln_affine      norm3 (fp32 weight, bias)  model.py:664-665             x fp32 -> bf16
//   wan_mx_gated_residual x.addcmul_(y, e[k]) / x += y   model.py:658, :668, :708   x fp32 in place, y bf16
//   wan_mx_patch_embed    patch_embedding(x).to(fp32)    model.py:1620-1631      fp32 Conv3d k = s = (1, 2, 2), no rounding
//   wan_mx_sinusoid, wan_mx_linear_f32                   model.py:1815-1818      the time MLP and projection in fp32
//   wan_mx_head           Head.forward on fp32 x         model.py:847-865        token-major fp32 [ntok, nout]
#include "common.h"

// the edge GEMMs are the bf16 plan's LDS-tiled fp32 kernels instantiated for fp32 rows (csrc/edge_ops.hip; round 5: the plain per-element
// forms they replace cost 37 ms per forward at 14B-720p -- 0.4 % of a step -- against 5 ms)
int wan_patch_embed_f32_range(const float* x, const float* y, const float* w, const float* bias, float* out, int Cin, int Cy, int F, int H, int W,
                              int d, int64_t tok0, int64_t ntok, void* stream);
int wan_head_gemm_f32(const float* xm, const float* w, const float* bias, float* out, int64_t ntok, int d, int nout, void* stream);

namespace {

// 0: modulate -> bf16 (p0 = modulation bf16 [n_mod, d], p1 = e0 fp32 [batches, n_mod, d]);  1: affine -> bf16 (p0 = weight fp32, p1 = bias fp32)
// 2: modulate -> fp32 (p0 = head.modulation fp32 [2, d], p1 = e fp32 [batches, d]; shift row 0, scale row 1)
// NV > 0 (round 5): d == NV * 256 -- the Wan widths -- and the row lives in NV float4 registers per lane: ONE read of the row instead of
// three (the second and third came out of L2: 8 of the pass's 14 bytes per element), the modulation / e0 vectors in 8- and 16-byte loads.
// The same operations on the same values in the same order as the generic form (NV = 0): bit-identical results (tests/test_gpu_mixed.py).
// Wide rows (NV >= 12): three workgroups per CU (168 registers: the row is 80 at d = 5120), each wave with its whole row in flight.
template <int MODE, int NV = 0>
__global__ __launch_bounds__(256, (NV >= 12 ? 3 : 4)) void mx_ln_kernel(const float* __restrict__ x, void* __restrict__ out, const void* __restrict__ p0,
                                                    const float* __restrict__ p1, int n_mod, int shift_idx, int scale_idx, int64_t rows,
                                                    int64_t rpb, int d, float eps) {
  const int lane = threadIdx.x & 63;
  // the wave index as a scalar: row, batch and every row pointer then live in SGPRs and an access is base (SGPR) + lane offset (one VGPR)
  // + immediate -- per-lane 64-bit addresses for 4 vectors x NV chunks were what spilled at d = 5120
  const int64_t row = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (row >= rows) return;
  const float* xr = x + row * d;
  if constexpr (NV > 0) {
    float4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = *reinterpret_cast<const float4*>(xr + lane * 4 + i * 256);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += wan_add_f32(wan_add_f32(v[i].x, v[i].y), wan_add_f32(v[i].z, v[i].w));   // (plain adds: common.h)
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, dd = v[i].w - mean;
      q += (a * a + b * b) + (cc * cc + dd * dd);
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + eps);
    const int64_t batch = (int64_t)((uint32_t)row / (uint32_t)rpb);   // rows < 2^31 (launcher)
    // the per-column vectors of chunk i + 1 are fetched while chunk i is computed; a compiler barrier per chunk keeps the scheduler from
    // hoisting all NV chunks' loads (and their 64-bit addresses) to the top -- that cost 120 spilled registers at d = 5120
    struct Vec { uint2 a, b; float4 c, d; };
    auto fetch = [&](int i) -> Vec {
      const int c = lane * 4 + i * 256;
      Vec r = {};
      if (MODE == 3) {   // p1 = table fp32 [batches][2][d]: row 0 = 1 + (mod + e0)[scale], row 1 = (mod + e0)[shift] (mx_modtab_kernel)
        r.c = *reinterpret_cast<const float4*>(p1 + batch * (int64_t)2 * d + c);
        r.d = *reinterpret_cast<const float4*>(p1 + batch * (int64_t)2 * d + d + c);
      } else if (MODE == 1) {
        r.c = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p0) + c);
        r.d = *reinterpret_cast<const float4*>(p1 + c);
      } else {
        const float* hm = reinterpret_cast<const float*>(p0);
        r.c = *reinterpret_cast<const float4*>(hm + c);
        r.d = *reinterpret_cast<const float4*>(hm + d + c);
        const float4 ee = *reinterpret_cast<const float4*>(p1 + batch * (int64_t)d + c);
        r.a.x = __float_as_uint(ee.x); r.a.y = __float_as_uint(ee.y); r.b.x = __float_as_uint(ee.z); r.b.y = __float_as_uint(ee.w);
      }
      return r;
    };
    Vec cur = fetch(0);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane * 4 + i * 256;
      Vec nxt = cur;
      if (i + 1 < NV) nxt = fetch(i + 1);
      asm volatile("" ::: "memory");
      const float y[4] = {(v[i].x - mean) * rstd, (v[i].y - mean) * rstd, (v[i].z - mean) * rstd, (v[i].w - mean) * rstd};
      float r[4];
      if (MODE == 3) {
        const float scf[4] = {cur.c.x, cur.c.y, cur.c.z, cur.c.w}, shf[4] = {cur.d.x, cur.d.y, cur.d.z, cur.d.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = __fadd_rn(__fmul_rn(y[j], scf[j]), shf[j]);      // `x_mod *= 1 + e[1]; x_mod += e[0]`: two fp32 roundings
      } else if (MODE == 1) {
        r[0] = y[0] * cur.c.x + cur.d.x; r[1] = y[1] * cur.c.y + cur.d.y; r[2] = y[2] * cur.c.z + cur.d.z; r[3] = y[3] * cur.c.w + cur.d.w;
      } else {
        const float h0f[4] = {cur.c.x, cur.c.y, cur.c.z, cur.c.w}, h1f[4] = {cur.d.x, cur.d.y, cur.d.z, cur.d.w};
        const float ef[4] = {__uint_as_float(cur.a.x), __uint_as_float(cur.a.y), __uint_as_float(cur.b.x), __uint_as_float(cur.b.y)};
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = __fadd_rn(__fmul_rn(y[j], 1.0f + (h1f[j] + ef[j])), h0f[j] + ef[j]);
      }
      static_assert(NV == 0 || MODE != 0, "the register-resident modulate reads the table form (MODE 3)");
      if (MODE == 2) {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + row * d + c) = float4{r[0], r[1], r[2], r[3]};
      } else {
        uint2 w2;
        w2.x = pack2bf(r[0], r[1]);
        w2.y = pack2bf(r[2], r[3]);
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(out) + row * d + c) = w2;
      }
      cur = nxt;
    }
    return;
  }
  float s = 0.f;
  for (int c = lane * 4; c < d; c += 256) {
    const float4 v = *reinterpret_cast<const float4*>(xr + c);
    s += wan_add_f32(wan_add_f32(v.x, v.y), wan_add_f32(v.z, v.w));
  }
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
  for (int c = lane * 4; c < d; c += 256) {
    const float4 v = *reinterpret_cast<const float4*>(xr + c);
    const float a = v.x - mean, b = v.y - mean, cc = v.z - mean, dd = v.w - mean;
    q += (a * a + b * b) + (cc * cc + dd * dd);
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)d + eps);
  const int64_t batch = row / rpb;
  for (int c = lane * 4; c < d; c += 256) {
    const float4 v = *reinterpret_cast<const float4*>(xr + c);
    float y[4] = {(v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd};
    float r[4];
    if (MODE == 0) {
      const bf16_t* mod = reinterpret_cast<const bf16_t*>(p0);
      const float* e0 = p1 + batch * (int64_t)n_mod * d;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float sc = bf2f(mod[(int64_t)scale_idx * d + c + j]) + e0[(int64_t)scale_idx * d + c + j];   // bf16 parameter + fp32 e0 -> fp32 (:632)
        const float sh = bf2f(mod[(int64_t)shift_idx * d + c + j]) + e0[(int64_t)shift_idx * d + c + j];
        r[j] = __fadd_rn(__fmul_rn(y[j], 1.0f + sc), sh);                                                 // `x_mod *= 1 + e[1]; x_mod += e[0]`: two fp32 roundings
      }
    } else if (MODE == 1) {
      const float* w = reinterpret_cast<const float*>(p0);
#pragma unroll
      for (int j = 0; j < 4; ++j) r[j] = y[j] * w[c + j] + p1[c + j];
    } else {
      const float* hm = reinterpret_cast<const float*>(p0);
      const float* e = p1 + batch * (int64_t)d;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float sh = hm[c + j] + e[c + j], sc = hm[d + c + j] + e[c + j];                             // (head.modulation + e.unsqueeze(1)).chunk(2): shift, scale (:857)
        r[j] = __fadd_rn(__fmul_rn(y[j], 1.0f + sc), sh);
      }
    }
    if (MODE == 2) {
      *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + row * d + c) = float4{r[0], r[1], r[2], r[3]};
    } else {
      uint2 w2;
      w2.x = pack2bf(r[0], r[1]);
      w2.y = pack2bf(r[2], r[3]);
      *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(out) + row * d + c) = w2;
    }
  }
}

// x[row][c] += y[row][c] * gate[batch][c]   (gate = modulation[gate_idx] + e0[batch][gate_idx], fp32)   or   x += y  (mod == nullptr)
__global__ __launch_bounds__(256) void mx_gated_residual_kernel(float* __restrict__ x, const bf16_t* __restrict__ y, const bf16_t* __restrict__ mod,
                                                                const float* __restrict__ e0, int n_mod, int gate_idx, int64_t rows,
                                                                int64_t rpb, int d) {
  const int64_t n4 = rows * (int64_t)d / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t el = i * 4;
    const int64_t row = el / d;
    const int c = (int)(el - row * d);
    float4 xv = *reinterpret_cast<const float4*>(x + el);
    const uint2 yv = *reinterpret_cast<const uint2*>(y + el);
    const float yy[4] = {__uint_as_float(yv.x << 16), __uint_as_float(yv.x & 0xffff0000u), __uint_as_float(yv.y << 16), __uint_as_float(yv.y & 0xffff0000u)};
    float xx[4] = {xv.x, xv.y, xv.z, xv.w};
    if (mod != nullptr) {
      const float* e = e0 + ((row / rpb) * n_mod + gate_idx) * (int64_t)d;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float g = bf2f(mod[(int64_t)gate_idx * d + c + j]) + e[c + j];
        xx[j] = __fadd_rn(xx[j], __fmul_rn(yy[j], g));      // addcmul_: self + (t1 * t2), the product rounded first
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) xx[j] = xx[j] + yy[j];
    }
    *reinterpret_cast<float4*>(x + el) = float4{xx[0], xx[1], xx[2], xx[3]};
  }
}

__global__ void mx_sinusoid_kernel(float tval, float* __restrict__ out, int dim) {
  wan_hold_lds_word();   // (common.h)
  const int half = dim / 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= half) return;
  const float freq = powf(10000.0f, -((float)i / (float)half));
  const float a = tval * freq;
  out[i] = cosf(a);
  out[half + i] = sinf(a);
}

// C[m][n] = bias[n] + sum_k act(A[m][k]) * W[n][k]   (act 0 none, 1 SiLU on the INPUT);  one wave per (m, n)
__global__ __launch_bounds__(256) void mx_linear_f32_kernel(const float* __restrict__ A, const float* __restrict__ Wt, const float* __restrict__ bias,
                                                            float* __restrict__ C, int M, int N, int K, int act) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int m = blockIdx.y;
  if (n >= N) return;
  const float* a = A + (int64_t)m * K;
  const float* w = Wt + (int64_t)n * K;
  float s = 0.f;
  for (int k = lane; k < K; k += 64) {
    float v = a[k];
    if (act == 1) v = v / (1.0f + expf(-v));
    s += v * w[k];
  }
  s = wave_sum(s);
  if (lane == 0) C[(int64_t)m * N + n] = s + (bias ? bias[n] : 0.f);
}

inline hipStream_t mx_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
// tab[b][0][c] = 1 + (mod[scale][c] + e0[b][scale][c]), tab[b][1][c] = mod[shift][c] + e0[b][shift][c]: the two fp32 vectors the modulate derives
// for every token row, once per (layer, batch) -- the same fp32 operations, so the modulate's results do not change
__global__ __launch_bounds__(256) void mx_modtab_kernel(const bf16_t* __restrict__ mod, const float* __restrict__ e0, float* __restrict__ tab, int n_mod,
                                                        int shift_idx, int scale_idx, int d, int nb) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nb * d) return;
  const int b = i / d, c = i - b * d;
  const float* e = e0 + (int64_t)b * n_mod * d;
  tab[(int64_t)b * 2 * d + c] = 1.0f + (bf2f(mod[(int64_t)scale_idx * d + c]) + e[(int64_t)scale_idx * d + c]);
  tab[(int64_t)b * 2 * d + d + c] = bf2f(mod[(int64_t)shift_idx * d + c]) + e[(int64_t)shift_idx * d + c];
}
constexpr size_t MXTAB_SLOT = (size_t)2 << 20;   // 42 batches (21 frames x 2 streams) x 2 x 5120 x 4 B = 1.7 MB is the largest Wan case
constexpr int MXTAB_NSLOT = 16;

bool g_mx_generic = false;   // test hook (wan_mx_debug_generic_rows): the re-reading form on every width
// register-resident form for the widths d == NV * 256 it is instantiated for (every Wan width and the test configs'), else the generic one
#define MX_LN_LAUNCH(MODE_, ...)                                                                                                          \
  do {                                                                                                                                   \
    const dim3 grid_((unsigned)((rows + 3) / 4));                                                                                        \
    const bool fits_ = !g_mx_generic && d % 256 == 0 && rows < ((int64_t)1 << 31) && rows_per_batch_ < ((int64_t)1 << 31);                                \
    switch (fits_ ? d / 256 : 0) {                                                                                                       \
      case 1: hipLaunchKernelGGL((mx_ln_kernel<MODE_, 1>), grid_, dim3(256), 0, mx_stream(stream), __VA_ARGS__); break;                  \
      case 2: hipLaunchKernelGGL((mx_ln_kernel<MODE_, 2>), grid_, dim3(256), 0, mx_stream(stream), __VA_ARGS__); break;                  \
      case 6: hipLaunchKernelGGL((mx_ln_kernel<MODE_, 6>), grid_, dim3(256), 0, mx_stream(stream), __VA_ARGS__); break;                  \
      case 12: hipLaunchKernelGGL((mx_ln_kernel<MODE_, 12>), grid_, dim3(256), 0, mx_stream(stream), __VA_ARGS__); break;                \
      case 20: hipLaunchKernelGGL((mx_ln_kernel<MODE_, 20>), grid_, dim3(256), 0, mx_stream(stream), __VA_ARGS__); break;                \
      default: hipLaunchKernelGGL((mx_ln_kernel<(MODE_ == 3 ? 1 : MODE_), 0>), grid_, dim3(256), 0, mx_stream(stream), __VA_ARGS__); break; /* (3: never reached, `wide` is checked by the caller) */ \
    }                                                                                                                                    \
  } while (0)

inline int mx_blocks(int64_t work, int per_block) {
  int64_t b = (work + per_block - 1) / per_block;
  if (b > 65536) b = 65536;
  return (int)(b < 1 ? 1 : b);
}

}  // namespace

extern "C" void wan_mx_debug_generic_rows(int on) { g_mx_generic = on != 0; }

extern "C" int wan_mx_ln_modulate(const float* x, wan_bf16* out, const wan_bf16* mod, const float* e0, int n_mod, int shift_idx, int scale_idx,
                                  int64_t rows, int64_t rows_per_batch, int d, float eps, void* stream) {
  WAN_REQUIRE(x && out && mod && e0, "wan_mx_ln_modulate: null pointer");
  WAN_REQUIRE(d % 4 == 0 && rows >= 0 && rows_per_batch >= 1 && n_mod >= 1 && shift_idx >= 0 && shift_idx < n_mod && scale_idx >= 0 && scale_idx < n_mod,
              "wan_mx_ln_modulate: bad arguments (d=%d n_mod=%d shift=%d scale=%d)", d, n_mod, shift_idx, scale_idx);
  if (rows == 0) return 0;
  const int64_t rows_per_batch_ = rows_per_batch;
  // the Wan widths: the modulation vectors from a per-(layer, batch) table in the library's scratch ring (16 slots per stream: a table is
  // dead when its row kernel has run), the row register-resident; other widths, or a table that does not fit: the generic form
  const int64_t nb = (rows + rows_per_batch - 1) / rows_per_batch;
  const bool wide = !g_mx_generic && (d == 256 || d == 512 || d == 1536 || d == 3072 || d == 5120) && rows < ((int64_t)1 << 31) &&
                    rows_per_batch < ((int64_t)1 << 31) && nb * d < ((int64_t)1 << 31);
  float* tab = wide ? reinterpret_cast<float*>(wan_scratch_ring_slot(/*tag=*/3, MXTAB_SLOT, MXTAB_NSLOT, (size_t)nb * 2 * d * 4, mx_stream(stream))) : nullptr;
  if (tab != nullptr) {
    hipLaunchKernelGGL(mx_modtab_kernel, dim3((unsigned)((nb * d + 255) / 256)), dim3(256), 0, mx_stream(stream), mod, e0, tab, n_mod, shift_idx, scale_idx, d, (int)nb);
    WAN_LAUNCH_CHECK();
    MX_LN_LAUNCH(3, x, (void*)out, (const void*)nullptr, (const float*)tab, n_mod, shift_idx, scale_idx, rows, rows_per_batch, d, eps);
  } else {
    hipLaunchKernelGGL((mx_ln_kernel<0, 0>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, mx_stream(stream), x, (void*)out, (const void*)mod, e0, n_mod,
                       shift_idx, scale_idx, rows, rows_per_batch, d, eps);
  }
  WAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int wan_mx_ln_affine(const float* x, wan_bf16* out, const float* w, const float* b, int64_t rows, int d, float eps, void* stream) {
  WAN_REQUIRE(x && out && w && b, "wan_mx_ln_affine: null pointer");
  WAN_REQUIRE(d % 4 == 0 && rows >= 0, "wan_mx_ln_affine: bad arguments (d=%d)", d);
  if (rows == 0) return 0;
  const int64_t rows_per_batch_ = rows;
  MX_LN_LAUNCH(1, x, (void*)out, (const void*)w, b, 1, 0, 0, rows, rows, d, eps);
  WAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int wan_mx_gated_residual(float* x, const wan_bf16* y, const wan_bf16* mod, const float* e0, int n_mod, int gate_idx, int64_t rows,
                                     int64_t rows_per_batch, int d, void* stream) {
  WAN_REQUIRE(x && y, "wan_mx_gated_residual: null pointer");
  const bool gated = gate_idx >= 0;
  WAN_REQUIRE(d % 4 == 0 && rows >= 0 && (!gated || (mod && e0 && gate_idx < n_mod && rows_per_batch >= 1)),
              "wan_mx_gated_residual: bad arguments (d=%d gate=%d n_mod=%d)", d, gate_idx, n_mod);
  if (rows == 0) return 0;
  hipLaunchKernelGGL(mx_gated_residual_kernel, dim3(mx_blocks(rows * (int64_t)d / 4, 256)), dim3(256), 0, mx_stream(stream), x, y,
                     gated ? mod : (const wan_bf16*)nullptr, e0, n_mod, gated ? gate_idx : 0, rows, gated ? rows_per_batch : rows, d);
  WAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int wan_mx_patch_embed(const float* x, const float* y, const float* w, const float* bias, float* out, int Cin, int Cy, int F, int H, int W,
                                  int d, int64_t tok0, int64_t ntok, void* stream) {
  WAN_REQUIRE(x && w && bias && out && (Cy == 0 || y), "wan_mx_patch_embed: null pointer");
  WAN_REQUIRE(H % 2 == 0 && W % 2 == 0 && tok0 >= 0 && ntok >= 0 && tok0 + ntok <= (int64_t)F * (H / 2) * (W / 2),
              "wan_mx_patch_embed: H, W must be even and the token range inside the grid");
  if (ntok == 0) return 0;
  return wan_patch_embed_f32_range(x, y, w, bias, out, Cin, Cy, F, H, W, d, tok0, ntok, stream);
}

extern "C" int wan_mx_sinusoid(float t, float* out, int dim, void* stream) {
  WAN_REQUIRE(out && dim >= 2 && dim % 2 == 0, "wan_mx_sinusoid: bad arguments");
  hipLaunchKernelGGL(mx_sinusoid_kernel, dim3((dim / 2 + 255) / 256), dim3(256), 0, mx_stream(stream), t, out, dim);
  WAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int wan_mx_linear_f32(const float* A, const float* W, const float* bias, float* C, int M, int N, int K, int act, void* stream) {
  WAN_REQUIRE(A && W && C && M >= 1 && M <= 65535 && N >= 1 && K >= 1 && (act == 0 || act == 1), "wan_mx_linear_f32: bad arguments");
  hipLaunchKernelGGL(mx_linear_f32_kernel, dim3((unsigned)((N + 3) / 4), (unsigned)M), dim3(256), 0, mx_stream(stream), A, W, bias, C, M, N, K, act);
  WAN_LAUNCH_CHECK();
  return 0;
}

// Head.forward on the fp32 token stream: LayerNorm, modulate with (head.modulation + e) in fp32 (into `tmp`, fp32 [ntok, d]), the fp32 head
// Linear -> out [ntok, nout] token-major (the caller unpatchifies, or gathers the shards first under sequence parallelism).
// e: [batches, d] fp32, one row per `e_rows_per_batch` tokens (per-frame timesteps: tokens of a frame).
extern "C" int wan_mx_head(const float* x, const float* hmod, const float* e, const float* w, const float* bias, float* tmp, float* out, int64_t ntok,
                           int d, float eps, int64_t e_rows_per_batch, int nout, void* stream) {
  WAN_REQUIRE(x && hmod && e && w && bias && tmp && out, "wan_mx_head: null pointer");
  WAN_REQUIRE(d % 4 == 0 && ntok >= 0 && e_rows_per_batch >= 1 && nout >= 1, "wan_mx_head: bad arguments");
  if (ntok == 0) return 0;
  hipLaunchKernelGGL(mx_ln_kernel<2>, dim3((unsigned)((ntok + 3) / 4)), dim3(256), 0, mx_stream(stream), x, (void*)tmp, (const void*)hmod, e, 2, 0, 1,
                     ntok, e_rows_per_batch, d, eps);
  WAN_LAUNCH_CHECK();
  return wan_head_gemm_f32(tmp, w, bias, out, ntok, d, nout, stream);
}
