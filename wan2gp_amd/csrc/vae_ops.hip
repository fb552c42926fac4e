// Wan2.1 causal 3D VAE kernels for gfx950 (fp16 storage, fp32 accumulate, channels-last).
// Replaces the torch ops of models/wan/modules/vae.py: CausalConv3d (:43-82) incl. the 2-frame
// causal cache, Conv2d of Resample (:124-141) with the nearest-exact 2x upsample fused into the
// gather (:105-111) or stride 2 + ZeroPad2d((0,1,0,1)), the time_conv channel->time interleave
// (:186-189), RMS_norm + SiLU (:97-103, :246), the softmax of AttentionBlock (:303-308) and
// _vae_float_to_cpu_uint8 (:18-20).
//
// Convolution = implicit GEMM on v_mfma_f32_16x16x32_f16 with the tile/LDS/epilogue structure of
// gemm_bf16.hip: Out[pixel][cout] = sum_k A[pixel][k] * Wp[cout][k], K enumerated in UNITS of 32
// input channels, unit u = tap * (Cin/32) + channel_block, tap = (kt*KH + kh)*KW + kw; a K-step
// (64) is two units.  The A operand is never materialised: every 16-byte LDS-DMA piece (8
// channels of one input pixel of one tap) is fetched straight from the channels-last activation
// tensor [T,H,W,C]; out-of-range taps (spatial zero padding, the causal front before the first
// frame) fetch from a 16-byte zero page, frames t < 0 come from the layer's 2-frame cache.
// Activations are [T, H, W, C] fp16 with C % 32 == 0 (3- and 16-channel tensors are zero padded
// to 32 channels by the layout kernels / weight packer).
#include <type_traits>

#include "common.h"

typedef __attribute__((ext_vector_type(8))) _Float16 mfma_f16x8v;

#define CBM 128
#define CBN 128
#define CSTAGE (CBM * 64 * 2)

__device__ __forceinline__ void glds16v(const void* gsrc, void* ldst) { wan_lds_dma16(gsrc, ldst); }

struct ConvP {
  const uint16_t* x;
  const uint16_t* cache0;   // input frame -2 (round 6: the two cached frames are separate pointers -- they may live in two earlier chunks' tensors)
  const uint16_t* cache1;   // input frame -1
  const uint16_t* zero16;
  const uint16_t* w;
  const uint16_t* bias;
  const uint16_t* res;
  uint16_t* out;
  float* out_f32;
  int Tin, Hin, Win, Cin;
  int Tout, Hout, Wout, Cout;
  int KT, KH, KW;
  int st_t, st_s, front, pad_s, ups;
  int CB, U, nk, ncache, interleave;
  int64_t M;
  int tiles_y, tiles_x;
};

// LDS-DMA with a wave-uniform 64-bit base in SGPRs and a 32-bit per-lane byte offset: the weight rows of a tile sit at fixed
// offsets from a base that advances by one k-tile per stage -- no per-lane pointer arithmetic in the K loop
// (LDS destinations as 32-bit LDS addresses: a generic pointer costs a null-check select per cast, four scalar instructions per piece)
__device__ __forceinline__ void glds16s(uint32_t voff, const void* sbase, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void glds16a(const void* gsrc, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_addr) : "memory");
}

constexpr int CONV_UMAX = 27 * 32;  // K units of the largest layer served: 3x3x3 taps x 1024 input channels (Wan2.2 VAE)

template <bool BIG, bool UPS>  // BIG: input chunks of 2^31 elements and more (64-bit gather offsets); UPS: 2x upsampled input
__global__ __launch_bounds__(256, 2) void conv3d_f16_kernel(ConvP p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * CSTAGE];
  // Gather plan of the K loop, built once per workgroup (round 3; the K-step used to decode its two units with ~140 scalar
  // instructions -- divisions by the channel-block / tap counts -- and re-derive every slot's three range checks: 146 VALU +
  // 139 SALU beside 32 MFMAs, i.e. issue-bound at a quarter of the matrix rate):
  //   utab[u]   unit u = tap * CB + channel block: {element offset of the tap + block (lo, hi), tap | kt << 8 | valid << 16 |
  //             kh << 20 | kw << 24, 32 * block}; a thread reads ITS unit of a K-step with one ds_read_b128, a step ahead;
  //   vmask[i]  per gather slot: bit tap = "the tap's input pixel exists" (spatial zero padding, causal front), bit 27 + kt =
  //             "frame t + kt lies in the 2-frame cache" -- the K loop tests bits instead of comparing coordinates.
  __shared__ __attribute__((aligned(16))) uint4 utab[CONV_UMAX + 2];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wy = wave >> 1, wx = wave & 1;

  const int nwg = p.tiles_y * p.tiles_x;
  const int wg = xcd_remap(blockIdx.x, nwg);
  // x tiles (cout) fastest: the few cout tiles of one pixel tile run together and share the gathered pixels in L2
  const int ty = wg / p.tiles_x, tx = wg - ty * p.tiles_x;
  const int64_t y0 = (int64_t)ty * CBM;
  const int x0 = tx * CBN;

  const int HWo = p.Hout * p.Wout;
  const int64_t frame_in = (int64_t)p.Hin * p.Win * p.Cin;
  const int Heff = UPS ? p.Hin * 2 : p.Hin, Weff = UPS ? p.Win * 2 : p.Win;
  const int Kp = p.nk * 64;
  using off_t = typename std::conditional<BIG, int64_t, int>::type;
  const off_t fin = (off_t)frame_in;

  for (int u = tid; u < 2 * p.nk; u += 256) {  // (the last K-step of an odd unit count has one invalid unit)
    const bool uok = u < p.U;
    const int uu = uok ? u : 0;
    const int tap = uu / p.CB, cb = uu - tap * p.CB;
    const int kw = tap % p.KW, t2 = tap / p.KW;
    const int kh = t2 % p.KH, kt = t2 / p.KH;
    // UPS: the unit's whole element offset (the up-sampled gather recomputes its address per slot anyway); else the SPATIAL part only
    // -- (kh Win + kw) Cin + 32 cb, 32 bits -- the frame part kt * frame sits in the slot's per-kt base pointer (round 4)
    const int64_t toff = (UPS ? (int64_t)kt * frame_in : (int64_t)0) + ((int64_t)kh * p.Win + kw) * p.Cin + cb * 32;
    uint4 e;
    e.x = (uint32_t)toff;
    e.y = (uint32_t)((uint64_t)toff >> 32);
    e.z = (uint32_t)tap | ((uint32_t)kt << 8) | ((uok ? 1u : 0u) << 16) | ((uint32_t)kh << 20) | ((uint32_t)kw << 24);
    e.w = (uint32_t)(cb * 32);
    utab[u] = e;
  }

  // per-slot constants: the output pixel's position in INPUT coordinates before the tap offset is added (time / row / column),
  // which of the K-step's two units the slot's chunk belongs to, and its channel offset inside the unit
  int s_to[4], s_ho[4], s_wo[4], s_usel[4], s_coff[4];
  uint32_t vmask[4], xoff32[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = i * 256 + tid;
    const int row = q >> 3, pch = q & 7;
    const int lch = pch ^ ((row >> 1) & 7);
    int64_t pp = y0 + row;
    if (pp > p.M - 1) pp = p.M - 1;
    const int to = (int)(pp / HWo);
    const int rem = (int)(pp - (int64_t)to * HWo);
    s_ho[i] = rem / p.Wout;
    s_wo[i] = rem - s_ho[i] * p.Wout;
    s_to[i] = to * p.st_t - p.front;
    s_ho[i] = s_ho[i] * p.st_s - p.pad_s;
    s_wo[i] = s_wo[i] * p.st_s - p.pad_s;
    s_usel[i] = lch >> 2;
    s_coff[i] = (lch & 3) * 8;
    const int slab = row >> 6, jj = row & 63;
    const int nt = jj >> 4, ii = jj & 15;
    int xr = x0 + slab * 64 + (ii >> 2) * 16 + nt * 4 + (ii & 3);
    if (xr > p.Cout - 1) xr = p.Cout - 1;
    xoff32[i] = (uint32_t)(((int64_t)xr * Kp + lch * 8) * 2);  // bytes from the k-tile's first weight (Cout x Kp x 2 B < 2^32)
    // unsigned compares fold the two-sided range checks: 0 <= hi < Heff, 0 <= wi < Weff, -ncache <= ti < Tin
    uint32_t m = 0;
    for (int kt = 0; kt < p.KT; ++kt) {
      const int ti = s_to[i] + kt;
      const bool okt = (unsigned)(ti + p.ncache) < (unsigned)(p.Tin + p.ncache);
      if (ti < 0) m |= 1u << (27 + kt);
      for (int kh = 0; kh < p.KH; ++kh)
        for (int kw = 0; kw < p.KW; ++kw)
          if (okt && (unsigned)(s_ho[i] + kh) < (unsigned)Heff && (unsigned)(s_wo[i] + kw) < (unsigned)Weff)
            m |= 1u << ((kt * p.KH + kh) * p.KW + kw);
    }
    vmask[i] = m;
  }

  off_t s_base[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) s_base[i] = (off_t)s_to[i] * fin + ((off_t)s_ho[i] * p.Win + s_wo[i]) * p.Cin + s_coff[i];
  // all four slots of a thread sit in the same unit of a K-step: chunk bit 2 after the swizzle is (tid >> 2 ^ tid >> 6) & 1
  const int my_unit = s_usel[0] != 0 ? 1 : 0;
  // frames t >= 0 live in x, frames -2 / -1 in the layer's cache: with each cached frame's base shifted by its own index both are
  // base + t * frame (32-bit element offsets unless the chunk has 2^31 elements or more)
  const uint16_t* const cbase0 = p.cache0 + 2 * frame_in;
  const uint16_t* const cbase1 = p.cache1 + frame_in;
  const uint16_t* wbase = p.w;  // first weight of the next k-tile to fetch (wave-uniform)
  __syncthreads();              // utab complete
  uint4 ent = utab[my_unit];    // this thread's unit of K-step 0; the next one is read a step ahead
  // Round 4 (counters, profiles/r04_vae_conv_pmc_sq_run15.json: 3.9 vector instructions per MFMA, the matrix pipe 44 % busy -- the K
  // loop was bound by the gather's address arithmetic: per slot a 64-bit tap offset, the cache / input pointer select, the zero-page
  // select).  The frame index kt of a unit changes every KH KW CB units only: each slot keeps the POINTER of its output pixel at the
  // current kt -- cache or input, frame offset folded in -- and a K-step adds the unit's 32-bit spatial offset to it.
  const uint16_t* cur[4];
  int cur_kt = -1;
  const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + (uint32_t)wave * 1024u;
  auto stage = [&](int s, int ks) {
    const uint32_t ybase = smem_lds + (uint32_t)(s * 2 * CSTAGE);
    const uint32_t xbase = ybase + CSTAGE;
    const uint4 E = ent;
    ent = utab[2 * (ks + 1) + my_unit];  // (one entry past the last K-step is inside the array, never used)
    const uint32_t tap = E.z & 0xffu, kt = (E.z >> 8) & 0xffu, uok = (E.z >> 16) & 1u;
    if (!UPS && (int)kt != cur_kt) {  // (lanes of a wave may sit in two consecutive units: a lane-masked update at a kt boundary)
      cur_kt = (int)kt;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        cur[i] = ((((vmask[i] >> (27 + kt)) & 1u) != 0) ? (s_to[i] + (int)kt == -2 ? cbase0 : cbase1) : p.x) + (s_base[i] + (off_t)kt * fin);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t woff = (uint32_t)(i * 256 * 16);
      const bool ok = (uok & (vmask[i] >> tap) & 1u) != 0;
      const uint16_t* src;
      if (UPS) {  // nearest-exact 2x upsample on the fly: source pixel (hi >> 1, wi >> 1)
        const bool in_cache = ((vmask[i] >> (27 + kt)) & 1u) != 0;
        const int hi = (s_ho[i] + (int)((E.z >> 20) & 0xfu)) >> 1, wi = (s_wo[i] + (int)((E.z >> 24) & 0xfu)) >> 1;
        const off_t off = (off_t)(s_to[i] + (int)kt) * fin + ((off_t)hi * p.Win + wi) * p.Cin + (int)E.w + s_coff[i];
        src = (in_cache ? (s_to[i] + (int)kt == -2 ? cbase0 : cbase1) : p.x) + off;
      } else {      // the slot's pointer at this kt + the unit's spatial offset
        src = cur[i] + E.x;
      }
      glds16a(ok ? src : p.zero16, ybase + woff);
      glds16s(xoff32[i], wbase, xbase + woff);
    }
    wbase += 64;  // stages are issued for consecutive K-steps: the weight rows advance by one 64-wide k-tile
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fch = lane >> 4;
  int yoff[4], xoff[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int ry = wy * 64 + t * 16 + frow;
    yoff[t] = ry * 128 + ((fch ^ ((ry >> 1) & 7)) << 4);
    const int rx = wx * 64 + t * 16 + frow;
    xoff[t] = rx * 128 + ((fch ^ ((rx >> 1) & 7)) << 4);
  }

  // WAN_CONV_ABL (make cabl; diagnostics builds only, outputs are garbage): 1 = no gather in the K loop, 2 = no LDS reads / MFMAs,
  // 3 = MFMAs on fragments read once -- what each of the loop's three streams costs alone (profiles/r04_vae_conv_ablation_*.log)
#ifndef WAN_CONV_ABL
#define WAN_CONV_ABL 0
#endif
  stage(0, 0);
#if WAN_CONV_ABL == 3
  mfma_f16x8v yf0[4], xf0[4];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = 0; t < 4; ++t) {
    yf0[t] = *reinterpret_cast<const mfma_f16x8v*>(smem + yoff[t]);
    xf0[t] = *reinterpret_cast<const mfma_f16x8v*>(smem + CSTAGE + xoff[t]);
  }
#endif
  for (int kt = 0; kt < p.nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < p.nk && WAN_CONV_ABL != 1) stage((kt + 1) & 1, kt + 1);
    const char* ybase = smem + (kt & 1) * 2 * CSTAGE;
    const char* xbase = ybase + CSTAGE;
#pragma unroll
    for (int ks = 0; ks < (WAN_CONV_ABL == 2 ? 0 : 2); ++ks) {
      mfma_f16x8v yf[4], xf[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#if WAN_CONV_ABL == 3
        yf[t] = yf0[t]; xf[t] = xf0[t];
        asm volatile("" : "+v"(yf[t]), "+v"(xf[t]));
#else
        yf[t] = *reinterpret_cast<const mfma_f16x8v*>(ybase + (yoff[t] ^ (ks << 6)));
        xf[t] = *reinterpret_cast<const mfma_f16x8v*>(xbase + (xoff[t] ^ (ks << 6)));
#endif
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf[b], yf[a], acc[a][b], 0, 0, 0);
    }
  }

  // ---- epilogue -------------------------------------------------------------------------------------
  const int xb = x0 + wx * 64 + (lane >> 4) * 16;
  float bcol[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) bcol[j] = 0.f;
  const bool full = xb + 16 <= p.Cout;
  if (p.bias != nullptr) {
    if (full) {
      unpack8t<true>(*reinterpret_cast<const uint4*>(p.bias + xb), bcol);
      unpack8t<true>(*reinterpret_cast<const uint4*>(p.bias + xb + 8), bcol + 8);
    } else {
      for (int j = 0; j < 16; ++j)
        if (xb + j < p.Cout) bcol[j] = h2f(p.bias[xb + j]);
    }
  }
  const int C2 = p.Cout >> 1;
#pragma unroll
  for (int yt = 0; yt < 4; ++yt) {
    const int64_t pp = y0 + wy * 64 + yt * 16 + (lane & 15);
    if (pp >= p.M) continue;
    float v[16];
#pragma unroll
    for (int xt = 0; xt < 4; ++xt)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[xt * 4 + r] = acc[yt][xt][r] + bcol[xt * 4 + r];
    if (full) {
      int64_t oidx;
      if (p.interleave) {
        // time_conv output [T,H,W,2*C2] -> [2T,H,W,C2]: channel half s goes to frame 2t+s (vae.py:186-189)
        const int to = (int)(pp / HWo);
        const int rem = (int)(pp - (int64_t)to * HWo);
        const int s = xb >= C2;
        oidx = ((int64_t)(2 * to + s) * HWo + rem) * C2 + (xb - s * C2);
      } else {
        oidx = pp * p.Cout + xb;
      }
      if (p.res != nullptr) {
        float rv[16];
        unpack8t<true>(*reinterpret_cast<const uint4*>(p.res + oidx), rv);
        unpack8t<true>(*reinterpret_cast<const uint4*>(p.res + oidx + 8), rv + 8);
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = rnd16<true>(v[j]) + rv[j];  // conv output is fp16, then x + h
      }
      if (p.out_f32 != nullptr) {
#pragma unroll
        for (int j = 0; j < 16; ++j) p.out_f32[oidx + j] = v[j];
      } else {
        *reinterpret_cast<uint4*>(p.out + oidx) = pack8t<true>(v);
        *reinterpret_cast<uint4*>(p.out + oidx + 8) = pack8t<true>(v + 8);
      }
    } else {
      for (int j = 0; j < 16; ++j) {
        if (xb + j < p.Cout) {
          const int64_t oidx = pp * p.Cout + xb + j;
          float o = v[j];
          if (p.res != nullptr) o = rnd16<true>(o) + h2f(p.res[oidx]);
          if (p.out_f32 != nullptr) p.out_f32[oidx] = o;
          else p.out[oidx] = f2h(o);
        }
      }
    }
  }
}

static uint16_t* g_zero_page = nullptr;
static int ensure_zero_page(hipStream_t st) {
  if (g_zero_page) return 0;
  WAN_CHECK_HIP(hipMalloc((void**)&g_zero_page, 256));
  WAN_CHECK_HIP(hipMemsetAsync(g_zero_page, 0, 256, st));
  return 0;
}

// Test hook: take the 64-bit-offset instantiations (BIG) regardless of the input size, so that a parity test can hold them
// bit-for-bit against the 32-bit ones on the same input (they are otherwise reached only by chunks of 2^31 elements and more).
// Test / A-B hook: keep the 3 x 3 x 3 stride-1 layers on the gather kernel above instead of the halo-patch kernel (vae_conv_halo.hip)
static int g_no_halo = 0;
extern "C" int wan_vae_debug_no_halo(int on) {
  const int old = g_no_halo;
  g_no_halo = on ? 1 : 0;
  return old;
}
int wan_vae_conv3d_halo_launch(const uint16_t* x, const uint16_t* cache0, const uint16_t* cache1, const uint16_t* zero16, const uint16_t* w,
                               const uint16_t* bias, const uint16_t* res, uint16_t* out, float* out_f32, int Tin, int H, int W, int Cin, int Tout,
                               int Cout, int front, int Kp, int KT, int ups, hipStream_t stream);

static int g_force_big = 0;
extern "C" int wan_vae_debug_force_big(int on) {
  const int old = g_force_big;
  g_force_big = on ? 1 : 0;
  return old;
}

extern "C" int wan_vae_conv3d_ex(const uint16_t* x, const uint16_t* cache0, const uint16_t* cache1, const uint16_t* w, const uint16_t* bias,
                                 const uint16_t* res, uint16_t* out, float* out_f32, int Tin, int Hin, int Win, int Cin,
                                 int Tout, int Hout, int Wout, int Cout, int KT, int KH, int KW, int st_t, int st_s,
                                 int front, int pad_s, int ups, int interleave, void* stream);
extern "C" int wan_vae_conv3d(const uint16_t* x, const uint16_t* cache, const uint16_t* w, const uint16_t* bias,
                              const uint16_t* res, uint16_t* out, float* out_f32, int Tin, int Hin, int Win, int Cin,
                              int Tout, int Hout, int Wout, int Cout, int KT, int KH, int KW, int st_t, int st_s,
                              int front, int pad_s, int ups, int interleave, void* stream) {
  return wan_vae_conv3d_ex(x, cache, cache ? cache + (int64_t)Hin * Win * Cin : nullptr, w, bias, res, out, out_f32, Tin, Hin, Win, Cin, Tout, Hout, Wout,
                           Cout, KT, KH, KW, st_t, st_s, front, pad_s, ups, interleave, stream);
}
// The causal cache as two frame pointers: cache1 = input frame -1, cache0 = input frame -2 (NULL with a non-NULL cache1: frame -2 reads
// as zeros, the first chunk of a stream whose chunks hold one frame); both NULL: no cache.  The frames may belong to the input tensors
// of two earlier chunks -- the layer graph (vae_graph.hip) keeps those alive instead of copying their last frames (vae.py:254-273's clone()s).
extern "C" int wan_vae_conv3d_ex(const uint16_t* x, const uint16_t* cache0, const uint16_t* cache1, const uint16_t* w, const uint16_t* bias,
                                 const uint16_t* res, uint16_t* out, float* out_f32, int Tin, int Hin, int Win, int Cin,
                                 int Tout, int Hout, int Wout, int Cout, int KT, int KH, int KW, int st_t, int st_s,
                                 int front, int pad_s, int ups, int interleave, void* stream) {
  WAN_REQUIRE(x && w && (out || out_f32), "wan_vae_conv3d: null pointer");
  WAN_REQUIRE(cache0 == nullptr || cache1 != nullptr, "wan_vae_conv3d: cache frame -2 without frame -1");
  WAN_REQUIRE((((uintptr_t)(cache0 ? cache0 : x) | (uintptr_t)(cache1 ? cache1 : x)) & 15) == 0, "wan_vae_conv3d: cache frames must be 16-byte aligned");
  WAN_REQUIRE(Cin % 32 == 0, "wan_vae_conv3d: Cin=%d must be a multiple of 32 (pad the activation/weights)", Cin);
  WAN_REQUIRE(!interleave || (Cout % 32 == 0 && res == nullptr && out_f32 == nullptr), "wan_vae_conv3d: bad interleave use");
  WAN_REQUIRE(front <= 2 && front >= 0, "wan_vae_conv3d: front must be 0..2");
  hipStream_t st = as_stream(stream);
  if (int rc = ensure_zero_page(st)) return rc;
  ConvP p;
  p.x = x; p.cache1 = cache1 ? cache1 : x; p.cache0 = cache0 ? cache0 : p.cache1; p.zero16 = g_zero_page; p.w = w; p.bias = bias; p.res = res; p.out = out;
  p.out_f32 = out_f32;
  p.Tin = Tin; p.Hin = Hin; p.Win = Win; p.Cin = Cin; p.Tout = Tout; p.Hout = Hout; p.Wout = Wout; p.Cout = Cout;
  p.KT = KT; p.KH = KH; p.KW = KW; p.st_t = st_t; p.st_s = st_s; p.front = front; p.pad_s = pad_s; p.ups = ups;
  p.CB = Cin / 32;
  p.U = KT * KH * KW * p.CB;
  p.nk = (p.U + 1) / 2;
  p.ncache = cache1 ? (cache0 ? 2 : 1) : 0;
  p.interleave = interleave;
  p.M = (int64_t)Tout * Hout * Wout;
  if (p.M == 0) return 0;
  const bool big = g_force_big || (int64_t)(Tin + 2) * Hin * Win * Cin >= ((int64_t)1 << 31);
  // the residual blocks' convolutions (3 x 3 x 3, stride 1, "same" in space): the halo-patch kernel -- each input pixel staged once per
  // frame tap and channel block instead of gathered once per tap (round 4)
  // ... and Resample's Conv2d 3 x 3 behind the nearest-exact 2x up-sampling (KT = 1: the patch pixel (hi, wi) is input pixel (hi >> 1, wi >> 1))
  if (!big && !g_no_halo && (KT == 3 || KT == 1) && KH == 3 && KW == 3 && st_t == 1 && st_s == 1 && pad_s == 1 && !interleave &&
      Hout == (ups ? 2 * Hin : Hin) && Wout == (ups ? 2 * Win : Win) && (KT == 3 ? !ups : (front == 0 && cache1 == nullptr)) &&
      (int64_t)Tout * Hout * Wout * Cout < ((int64_t)1 << 31))
    return wan_vae_conv3d_halo_launch(x, cache0, cache1, g_zero_page, w, bias, res, out, out_f32, Tin, Hout, Wout, Cin, Tout, Cout, front, p.nk * 64, KT,
                                      ups, st);
  p.tiles_y = (int)((p.M + CBM - 1) / CBM);
  p.tiles_x = (Cout + CBN - 1) / CBN;
  const dim3 grid((unsigned)(p.tiles_y * p.tiles_x));
  if (big && ups) hipLaunchKernelGGL((conv3d_f16_kernel<true, true>), grid, dim3(256), 0, st, p);
  else if (big) hipLaunchKernelGGL((conv3d_f16_kernel<true, false>), grid, dim3(256), 0, st, p);
  else if (ups) hipLaunchKernelGGL((conv3d_f16_kernel<false, true>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((conv3d_f16_kernel<false, false>), grid, dim3(256), 0, st, p);
  WAN_LAUNCH_CHECK();
  return 0;
}

// ---- RMS_norm (+ optional SiLU), channels-last ----------------------------------------------------------------------------
// G lanes per pixel (the power of two >= C/8, lanes beyond the pixel's C/8 chunks idle), 64/G pixels per wave: the wave's
// 16-byte loads cover one contiguous run of pixels (coalesced), each element is read once and written once, the sum of
// squares is reduced with xor-shuffles inside the lane group.  (The first version ran one THREAD per pixel: every lane of a
// load on its own cache line, two passes -- 23 % of a 720p decode.)
template <int G, int CPL>  // CPL: 16-byte chunks per lane (2 for the 640-channel levels of the Wan2.2 VAE)
__global__ __launch_bounds__(256) void vae_rmsnorm_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ out,
                                                          const uint16_t* __restrict__ gamma, int64_t npix, int C,
                                                          int silu) {
  constexpr int PPW = 64 / G;  // pixels per wave and iteration
  const int lane = threadIdx.x & 63;
  const int sub = lane & (G - 1);
  const int nchunk = C >> 3;
  float g[CPL][8];
#pragma unroll
  for (int k = 0; k < CPL; ++k)
    if (sub + k * G < nchunk) unpack8t<true>(*reinterpret_cast<const uint4*>(gamma + (sub + k * G) * 8), g[k]);
  const float sqrtC = sqrtf((float)C);
  const int64_t wave_id = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  // U pixel groups per wave and iteration, their loads issued together (round 6: with one 16-byte load per lane in flight the kernel ran at
  // 0.47 of the HBM roof on the 96-channel level -- 8 % of a 720p decode; the arithmetic of a pixel is untouched)
  constexpr int U = CPL == 1 ? 4 : 2;
  for (int64_t p0 = wave_id * (PPW * U); p0 < npix; p0 += nwaves * (PPW * U)) {
    float v[U][CPL][8];
    bool row[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t pix = p0 + u * PPW + lane / G;
      row[u] = pix < npix;
#pragma unroll
      for (int k = 0; k < CPL; ++k)
        if (row[u] && sub + k * G < nchunk) unpack8t<true>(*reinterpret_cast<const uint4*>(x + pix * C + (sub + k * G) * 8), v[u][k]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t pix = p0 + u * PPW + lane / G;
      float ss = 0.f;
#pragma unroll
      for (int k = 0; k < CPL; ++k)
        if (row[u] && sub + k * G < nchunk) {
#pragma unroll
          for (int j = 0; j < 8; ++j) ss += v[u][k][j] * v[u][k][j];
        }
#pragma unroll
      for (int m = G >> 1; m >= 1; m >>= 1) ss += __shfl_xor(ss, m, 64);
      // F.normalize: x / max(||x||, eps), eps = 1e-12; then * sqrt(C) * gamma
      const float inv = sqrtC / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
      for (int k = 0; k < CPL; ++k)
        if (row[u] && sub + k * G < nchunk) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float y = v[u][k][j] * inv * g[k][j];
            if (silu) {
              y = rnd16<true>(y);  // RMS_norm returns an fp16 tensor, SiLU then acts on it
              y = y / (1.0f + __expf(-y));
            }
            v[u][k][j] = y;
          }
          *reinterpret_cast<uint4*>(out + pix * C + (sub + k * G) * 8) = pack8t<true>(v[u][k]);
        }
    }
  }
}

extern "C" int wan_vae_rmsnorm_silu(const uint16_t* x, uint16_t* out, const uint16_t* gamma, int64_t npix, int C,
                                    int silu, void* stream) {
  WAN_REQUIRE(x && out && gamma && C % 8 == 0 && C >= 8 && C <= 1024, "wan_vae_rmsnorm_silu: bad args (C=%d: a multiple of 8, <= 1024)", C);
  if (npix == 0) return 0;
  const int nchunk = C >> 3;
  const int G = nchunk <= 16 ? 16 : nchunk <= 32 ? 32 : 64;
  const int U = (nchunk <= 64) ? 4 : 2;      // (pixel groups per wave and iteration: vae_rmsnorm_kernel's U)
  const int64_t waves = (npix + (64 / G) * U - 1) / ((64 / G) * U);
  int64_t blocks = (waves + 3) / 4;
  if (blocks > 256 * 32) blocks = 256 * 32;  // grid-stride beyond 32 workgroups per CU
  const dim3 grid((unsigned)blocks);
#define RMS_LAUNCH(GG, CC) hipLaunchKernelGGL((vae_rmsnorm_kernel<GG, CC>), grid, dim3(256), 0, as_stream(stream), x, out, gamma, npix, C, silu)
  if (G == 16) RMS_LAUNCH(16, 1);
  else if (G == 32) RMS_LAUNCH(32, 1);
  else if (nchunk <= 64) RMS_LAUNCH(64, 1);
  else RMS_LAUNCH(64, 2);
#undef RMS_LAUNCH
  WAN_LAUNCH_CHECK();
  return 0;
}

// ---- row softmax for the VAE attention block: P[r, :L] = softmax(S[r, :L]), P[r, L:ld] = 0 -------------
__global__ __launch_bounds__(256) void vae_softmax_kernel(const uint16_t* __restrict__ S, uint16_t* __restrict__ P,
                                                          int L, int64_t ld) {
  __shared__ float red[8];
  const uint16_t* s = S + (int64_t)blockIdx.x * ld;
  uint16_t* o = P + (int64_t)blockIdx.x * ld;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float m = -INFINITY;
  for (int i = threadIdx.x; i < L; i += 256) m = fmaxf(m, h2f(s[i]));
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int i = threadIdx.x; i < L; i += 256) sum += __expf(h2f(s[i]) - m);
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  for (int i = threadIdx.x; i < ld; i += 256) o[i] = (i < L) ? f2h(__expf(h2f(s[i]) - m) * inv) : (uint16_t)0;
}

extern "C" int wan_vae_softmax(const uint16_t* S, uint16_t* P, int64_t rows, int L, int64_t ld, void* stream) {
  WAN_REQUIRE(S && P && ld >= L, "wan_vae_softmax: bad args");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(vae_softmax_kernel, dim3((unsigned)rows), dim3(256), 0, as_stream(stream), S, P, L, ld);
  WAN_LAUNCH_CHECK();
  return 0;
}

// ---- layout / dtype edges ------------------------------------------------------------------------------
// fp32 [C, T, H, W] -> fp16 channels-last [T, H, W, Cp] (channels >= C zero); optional per-channel affine
// v = v * mul[c] + add[c]  (decode: z / scale[1] + scale[0], vae.py:631-635)
__global__ void vae_pack_kernel(const float* __restrict__ in, uint16_t* __restrict__ out, const float* __restrict__ mul,
                                const float* __restrict__ add, int C, int Cp, int64_t thw) {
  const int64_t total = thw * Cp;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pix = i / Cp;
    const int c = (int)(i - pix * Cp);
    float v = 0.f;
    if (c < C) {
      v = in[(int64_t)c * thw + pix];
      if (mul) v = v * mul[c] + add[c];
    }
    out[i] = f2h(v);
  }
}
extern "C" int wan_vae_pack(const float* in, uint16_t* out, const float* mul, const float* add, int C, int Cp,
                            int64_t thw, void* stream) {
  WAN_REQUIRE(in && out && Cp >= C && (mul == nullptr) == (add == nullptr), "wan_vae_pack: bad args");
  int blocks = (int)((thw * Cp + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  if (blocks == 0) return 0;
  hipLaunchKernelGGL(vae_pack_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), in, out, mul, add, C, Cp, thw);
  WAN_LAUNCH_CHECK();
  return 0;
}

// fp16 channels-last [thw, Cs] (first C channels) -> fp32 [C, thw] with v = (v - sub[c]) * mul[c] (encode: mu
// normalisation, vae.py:618-624) when sub != null
__global__ void vae_unpack_kernel(const uint16_t* __restrict__ in, float* __restrict__ out, const float* __restrict__ sub,
                                  const float* __restrict__ mul, int C, int Cs, int64_t thw) {
  const int64_t total = thw * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i / thw);
    const int64_t pix = i - (int64_t)c * thw;
    float v = h2f(in[pix * Cs + c]);
    if (sub) v = (v - sub[c]) * mul[c];
    out[i] = v;
  }
}
extern "C" int wan_vae_unpack(const uint16_t* in, float* out, const float* sub, const float* mul, int C, int Cs,
                              int64_t thw, void* stream) {
  WAN_REQUIRE(in && out && Cs >= C, "wan_vae_unpack: bad args");
  int blocks = (int)((thw * C + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  if (blocks == 0) return 0;
  hipLaunchKernelGGL(vae_unpack_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), in, out, sub, mul, C, Cs, thw);
  WAN_LAUNCH_CHECK();
  return 0;
}

// decoder head output fp32 channels-last [T*H*W, 3] -> uint8 [3, Ttot, H, W] at frame offset t0
// (clamp(-1,1) -> +1 -> *127.5 -> round-half-even -> clamp(0,255), vae.py:18-20), and/or fp32 [3,Ttot,H,W]
__global__ void vae_to_video_kernel(const float* __restrict__ in, uint8_t* __restrict__ u8, float* __restrict__ f32,
                                    int T, int64_t hw, int Ttot, int t0) {
  const int64_t total = (int64_t)T * hw * 3;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i / ((int64_t)T * hw));
    const int64_t r = i - (int64_t)c * T * hw;
    const int t = (int)(r / hw);
    const int64_t px = r - (int64_t)t * hw;
    const float v = in[((int64_t)t * hw + px) * 3 + c];
    const int64_t o = ((int64_t)c * Ttot + t0 + t) * hw + px;
    if (f32) f32[o] = v;
    if (u8) {
      float q = fminf(fmaxf(v, -1.0f), 1.0f);
      q = (q + 1.0f) * 127.5f;
      q = rintf(q);  // round half to even, like torch.round_
      u8[o] = (uint8_t)fminf(fmaxf(q, 0.0f), 255.0f);
    }
  }
}
extern "C" int wan_vae_to_video(const float* in, uint8_t* u8, float* f32, int T, int64_t hw, int Ttot, int t0,
                                void* stream) {
  WAN_REQUIRE(in && (u8 || f32), "wan_vae_to_video: bad args");
  int blocks = (int)(((int64_t)T * hw * 3 + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  if (blocks == 0) return 0;
  hipLaunchKernelGGL(vae_to_video_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), in, u8, f32, T, hw, Ttot, t0);
  WAN_LAUNCH_CHECK();
  return 0;
}
