// The 3 x 3 x 3 stride-1 causal convolutions of the Wan VAE (vae.py:43-82: every ResidualBlock conv, conv1 / conv_in / head of the
// decoder and encoder -- most of a decode's FLOPs) as an implicit GEMM whose activation operand is a HALO PATCH in LDS.
//
// Why (round 4): the gather kernel of vae_ops.hip fetches every K-step's 128 x 64 im2col slab from L2 -- each input
// pixel once per tap, 27 times -- and the ablation (make cabl, profiles/r04_vae_conv_ablation_run18.log) showed that stream alone costs 80 %
// of the loop's time: 64 B per matrix-peak clock and CU are needed, ~23 are delivered.  Here a workgroup owns a 16 x 16 pixel tile of one
// output frame and 128 output channels; for each (frame tap kt, 32-channel block cb) it stages the 18 x 18 x 32-channel input patch ONCE
// (20.7 KB) and the nine spatial taps read shifted windows of it: the activation half of the stream drops ninefold, the weight half is
// shared by 256 pixels instead of 128.  Per (kt, cb): 21 KB of patch + 9 x 8 KB of weights for 9 x 16 MFMAs per wave.
//
// K order: (kt, cb, kh, kw) -- not vae_ops.hip's (kt, kh, kw, cb): the fp32 sums are taken in another order, so the two kernels agree to
// rounding, not bit for bit (tests compare both with the oracle; the bit-for-bit BIG / non-BIG comparison runs with this kernel switched off).
//
// Workgroup: 512 threads = 8 waves, wave (wr = wave >> 1, wc = wave & 1) owns pixel rows 4 wr .. + 3 of the tile (four 16-pixel MFMA tiles,
// one per row) x output channels 64 wc .. + 63 (four 16-channel tiles); two workgroups per CU (66 KB of LDS each, <= 128 registers).
// LDS: patch stage s (2): pixel pp = pr * 18 + pc of the patch at s * 21,504 + pp * 64, its 8-channel chunk c in 16-byte slot
//      c ^ ((pp >> 1) & 2); weights ring (3): row R (= permuted output channel, as vae_ops.hip: MFMA tile xt row 4 g + r <-> channel
//      16 g + 4 xt + r, so a lane ends with 16 consecutive channels) at R * 64, chunk c in slot c ^ ((R >> 1) & 2).  That swizzle is the one
//      (found by enumeration over ds_read_b128's four lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...) that keeps 16 consecutive rows x
//      4 chunks conflict-free from ANY starting row -- the patch windows start anywhere.  The first version (80-byte pixels, no swizzle)
//      spent half of its LDS cycles in bank conflicts (profiles/r04_vae_conv_pmc_lds_run24.json).
// MFMA: acc[a][b] += W frag(b) x P frag(a), v_mfma_f32_16x16x32_f16, one k-step per (tap, cb).
#include <stdlib.h>

#include "common.h"

typedef __attribute__((ext_vector_type(8))) _Float16 halo_f16x8;

namespace {

constexpr int HT = 16;                       // tile edge (pixels)
constexpr int HP = HT + 2;                   // patch edge
constexpr int HPITCH = 64;                   // bytes per patch pixel: 32 channels, no padding
constexpr int HPIECES = 21;                  // 1-KB DMA pieces per patch stage (324 x 64 = 20,736 B)
constexpr int HPATCH = HPIECES * 1024;       // bytes per patch stage
constexpr int HWST = 128 * 64;               // bytes per weight stage (96 x 64 used by the 96-channel tile)
constexpr int HW0 = 2 * HPATCH;              // weights ring offset

struct HaloP {
  const uint16_t* x;
  const uint16_t* cache0;    // input frame -2 (or null: zeros) and
  const uint16_t* cache1;    // input frame -1 (or null: no cache) -- separate pointers (round 6: the frames may live in two earlier chunks' tensors)
  const uint16_t* zero16;
  const uint16_t* w;         // [Cout][Kp], K in units of 32 channels, unit = tap * CB + cb (vae_ops.hip's packing)
  const uint16_t* bias;
  const uint16_t* res;
  uint16_t* out;
  float* out_f32;
  int Tin, H, W, Cin, Tout, Cout, CB, Kp, front, ncache;
  int tiles_h, tiles_w, tiles_x;
};

__device__ __forceinline__ void hglds_s(uint32_t voff, const void* sbase, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void hglds_a(const void* gsrc, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ const char* huni(const char* p) {
  const uint64_t u = (uint64_t)(uintptr_t)p;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
  return (const char*)(uintptr_t)(((uint64_t)hi << 32) | lo);
}

template <int KT, bool UPS, int NB, int WCX = 2>  // WCX: waves across the output channels -- 2: wave (wr = wave >> 1, wc = wave & 1) owns 4 pixel rows x NB channel tiles
                                     // (the forms above); 1 (round 6, NB = 1): a convolution of at most 16 output channels -- the decoder's head, 96 -> 3 --
                                     // gives every wave 2 pixel rows x ONE channel tile: the 96-wide tile spent 32 MFMAs on 3 channels (5 % of a
                                     // 720p decode's matrix time); same K order per output element, so the same bits.
                                     // NB: 16-channel tiles per wave -- 4: 128 output channels per workgroup; 3: 96 (Cout a multiple of 96: the 96- / 192- /
                                     // 384-channel levels -- a 128-wide tile would spend a quarter of its MFMAs on channels that do not exist).  KT = 3: the causal 3 x 3 x 3 convolution; KT = 1: Resample's Conv2d 3 x 3 -- UPS: on the nearest-exact 2x up-sampled input (vae.py:105-111,
                             // :124-141), i.e. patch pixel (hi, wi) of the up-sampled frame is input pixel (hi >> 1, wi >> 1); p.H / p.W are the OUTPUT sizes
__global__ __launch_bounds__(512, 4) void conv3d_halo_kernel(HaloP p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * HPATCH + 3 * HWST];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int RT = 16 / (8 / WCX);        // pixel rows (MFMA tiles) per wave: 4 with two waves across the channels, 2 with one
  const int wr = WCX == 2 ? wave >> 1 : wave, wc = WCX == 2 ? (wave & 1) : 0;
  const int n = lane & 15, lg = lane >> 4;

  const int nwg = p.Tout * p.tiles_h * p.tiles_w * p.tiles_x;
  int wg = xcd_remap(blockIdx.x, nwg);
  const int tx = wg % p.tiles_x; wg /= p.tiles_x;       // cout tiles fastest: they share the patch in L2
  const int tw = wg % p.tiles_w; wg /= p.tiles_w;
  const int th = wg % p.tiles_h;
  const int to = wg / p.tiles_h;
  constexpr int NW = NB * 16;            // output channels per wave
  const int h0 = th * HT, w0 = tw * HT, x0 = tx * (WCX * NW);
  const int Hs = UPS ? p.H >> 1 : p.H, Ws = UPS ? p.W >> 1 : p.W;   // the stored input frame
  const int64_t frame = (int64_t)Hs * Ws * p.Cin;

  // ---- patch pieces: this wave's three 1-KB pieces of a stage (24 issued for 21: the last three repeat pieces 0..2 -- same bytes to the
  // same place -- so that every wave has the same number of loads in flight and one s_waitcnt serves all)
  constexpr int NPP = 3;
  int so[NPP];          // element offset of the lane's 16 bytes inside a frame (channel block 0), -1: zeros
  uint32_t pdst[NPP];   // LDS offset of the piece inside a stage
#pragma unroll
  for (int i = 0; i < NPP; ++i) {
    const int pi = (wave * NPP + i) % HPIECES;
    const int o = pi * 1024 + lane * 16;
    const int pp = o >> 6, slot = (o >> 4) & 3;
    const int ch = slot ^ ((pp >> 1) & 2);               // the 8-channel chunk that lives in this slot (see the LDS note above)
    const int pr = pp / HP, pc = pp - pr * HP;
    const int hi = h0 - 1 + pr, wi = w0 - 1 + pc;
    const bool ok = pp < HP * HP && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
    so[i] = ok ? ((UPS ? hi >> 1 : hi) * Ws + (UPS ? wi >> 1 : wi)) * p.Cin + ch * 8 : -1;
    pdst[i] = (uint32_t)__builtin_amdgcn_readfirstlane(pi * 1024);   // (wave-uniform: a scalar register)
  }
  // ---- weight piece: 16 rows x 64 B of a stage per wave (NB = 3: six pieces, waves 6 and 7 repeat pieces 0 and 1)
  uint32_t wvoff;
  const int wpiece = __builtin_amdgcn_readfirstlane(wave % (WCX * NB));
  {
    const int R = wpiece * 16 + (lane >> 2), slot = lane & 3;
    const int c = slot ^ ((R >> 1) & 2);
    const int slab = R / NW, jj = R - slab * NW, xt = jj >> 4, ii = jj & 15;
    int co = x0 + slab * NW + (ii >> 2) * (NB * 4) + xt * 4 + (ii & 3);   // MFMA tile xt row 4 g + r <-> channel (NB * 4) g + 4 xt + r of the wave's NW
    if (co > p.Cout - 1) co = p.Cout - 1;
    wvoff = (uint32_t)(((int64_t)co * p.Kp + c * 8) * 2);
  }
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  // frame tap kt of this output frame: input frame ti = to - front + kt; < 0: the cache (two frames in front of x), beyond: zeros
  auto frame_ptr = [&](int kt) -> const uint16_t* {
    const int ti = to - p.front + kt;
    if (ti >= 0) return ti < p.Tin ? p.x + (int64_t)ti * frame : nullptr;
    return (p.ncache > 0 && ti >= -p.ncache) ? (ti == -2 ? p.cache0 : p.cache1) : nullptr;
  };
  const int G = KT * p.CB;  // groups (kt, cb), walked with carried coordinates (a scalar division by CB per tap is a VALU sequence)
  auto issue_patch = [&](int stage, int kt, int cb) {
    const uint16_t* fp = frame_ptr(kt);
    const uint32_t st = lds0 + (uint32_t)(stage * HPATCH);
#pragma unroll
    for (int i = 0; i < NPP; ++i) {
      const uint16_t* src = (fp != nullptr && so[i] >= 0) ? fp + so[i] + cb * 32 : p.zero16;
      hglds_a(src, st + pdst[i]);
    }
  };
  auto issue_w = [&](int u, int stage) {   // K unit u = (kt * 9 + tap) * CB + cb: 64 bytes of every weight row
    hglds_s(wvoff, huni(reinterpret_cast<const char*>(p.w) + (int64_t)u * 64), lds0 + HW0 + stage * HWST + wpiece * 1024);
  };

  f32x4 acc[RT][NB];
#pragma unroll
  for (int a = 0; a < RT; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int ybase[RT];
#pragma unroll
  for (int a = 0; a < RT; ++a) ybase[a] = ((RT * wr + a) * HP + n) * HPITCH + lg * 16;   // before the swizzle (it depends on the tap: bit 2 of the patch pixel)
  const int xbase = HW0 + (wc * NW + n) * 64 + ((lg ^ ((n >> 1) & 2)) << 4);   // (NW is a multiple of 16: bit 2 of the row is bit 2 of n)

  int kt = 0, cb = 0;                                   // group g
  int kt1 = (G > 1 && p.CB == 1) ? 1 : 0, cb1 = (G > 1 && p.CB > 1) ? 1 : 0;   // group g + 1 (the last group repeats itself: its re-loads land in idle stages)
  issue_patch(0, 0, 0);
  issue_w(0, 0);
  issue_w(p.CB, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int g = 0; g < G; ++g) {
    int poff = (g & 1) * HPATCH;
    asm volatile("" : "+s"(poff));   // (opaque: the swizzled window addresses are computed per tap, not kept for all nine across the loop)
    const int ub = kt * 9 * p.CB + cb, ub1 = kt1 * 9 * p.CB + cb1;   // K unit of tap 0 of the group: ((kt * 3 + kh) * 3 + kw) * CB + cb
#define HALO_TAP(TAP)                                                                                                              \
  {                                                                                                                                \
    if ((TAP) > 0 || g > 0) {                                                                                                      \
      /* lgkmcnt(0): THIS wave's fragment reads of the previous tap have RETURNED before the barrier lets any wave overwrite their ring stage.    */ \
      /* The MFMAs are builtins: hipcc software-pipelines the loop and parks the consuming MFMA (with its lgkmcnt wait) BELOW the next tap's     */ \
      /* barrier and DMA issue -- the read was then merely issued when the stage's next weights were on their way.  Beside another process on    */ \
      /* the same GPU (LDS queue backed up, L2-warm weights back in ~300 cycles) the DMA won: a wave's tile of garbage in 1 of 400 launches of   */ \
      /* the 96-wide tiles, 1 in 6 of the 16-channel head tile (runs 42-46: tools/probes/vae_conv_determinism.py).                                */ \
      if ((TAP) == 1 || (TAP) == 2) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); /* W(q) landed; W(q+1) and the next patch's three pieces may be out */ \
      else asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");                                                             \
      __builtin_amdgcn_s_barrier();                                                                                                \
      asm volatile("" ::: "memory");                                                                                               \
    }                                                                                                                              \
    issue_w((TAP) + 2 < 9 ? ub + ((TAP) + 2) * p.CB : ub1 + ((TAP) + 2 - 9) * p.CB, ((TAP) + 2) % 3);                               \
    if ((TAP) == 0) issue_patch((g + 1) & 1, kt1, cb1);                                                                            \
    constexpr int dy = (TAP) / 3, dx = (TAP) % 3;                                                                                  \
    halo_f16x8 yf[RT], xf[NB];                                                                                                     \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                                                \
      if (t < RT) {                                                                                                                \
        const int ya = ybase[t < RT ? t : 0] + (poff + (dy * HP + dx) * HPITCH);                                                   \
        yf[t < RT ? t : 0] = *reinterpret_cast<const halo_f16x8*>(smem + (ya ^ ((ya >> 3) & 32)));   /* chunk slot ^= 2 where bit 2 of the pixel is set (poff is a multiple of 512) */ \
      }                                                                                                                            \
      if (t < NB) xf[t < NB ? t : 0] = *reinterpret_cast<const halo_f16x8*>(smem + xbase + ((TAP) % 3) * HWST + t * 1024);         \
    }                                                                                                                              \
    _Pragma("unroll") for (int a = 0; a < RT; ++a)                                                                                 \
      _Pragma("unroll") for (int b = 0; b < NB; ++b)                                                                               \
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf[b], yf[a], acc[a][b], 0, 0, 0);                                      \
  }
    HALO_TAP(0) HALO_TAP(1) HALO_TAP(2) HALO_TAP(3) HALO_TAP(4) HALO_TAP(5) HALO_TAP(6) HALO_TAP(7) HALO_TAP(8)
#undef HALO_TAP
    kt = kt1; cb = cb1;
    if (g + 2 < G) { if (++cb1 == p.CB) { cb1 = 0; ++kt1; } }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing (clamped) loads

  // ---- epilogue (vae_ops.hip's: bias, fp16 rounding then + residual, fp16 or fp32 store; a lane owns 16 consecutive channels of a pixel)
  // (lane-derived addresses of the epilogue come from an opaque copy: hoisted above the K loop they cost it registers it does not have)
  int lane2 = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));   // the lane id again, from the exec mask (nothing kept across the loop)
  asm volatile("" : "+v"(lane2));
  const int n2 = lane2 & 15, lg2 = lane2 >> 4;
  constexpr int NV = NB * 4;             // consecutive output channels a lane owns
  const int xb = x0 + wc * NW + lg2 * NV;
  float bcol[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) bcol[j] = 0.f;
  const bool full = xb + NV <= p.Cout;
  if (p.bias != nullptr) {
    for (int j = 0; j < NV; ++j)
      if (xb + j < p.Cout) bcol[j] = h2f(p.bias[xb + j]);
  }
#pragma unroll
  for (int a = 0; a < RT; ++a) {
    const int h = h0 + RT * wr + a, w = w0 + n2;
    if (h >= p.H || w >= p.W) continue;
    const int64_t pp = ((int64_t)to * p.H + h) * p.W + w;
    float v[NV];
#pragma unroll
    for (int xt = 0; xt < NB; ++xt)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[xt * 4 + r] = acc[a][xt][r] + bcol[xt * 4 + r];
    if (full) {   // 8-byte pieces (4 channels): a 96-channel row is 8- but not 16-byte aligned at every lane
      const int64_t oidx = pp * p.Cout + xb;
      if (p.res != nullptr) {
#pragma unroll
        for (int q = 0; q < NB; ++q) {
          const uint2 rw = *reinterpret_cast<const uint2*>(p.res + oidx + 4 * q);
          const float rv[4] = {h2f((uint16_t)(rw.x & 0xffffu)), h2f((uint16_t)(rw.x >> 16)), h2f((uint16_t)(rw.y & 0xffffu)), h2f((uint16_t)(rw.y >> 16))};
#pragma unroll
          for (int j = 0; j < 4; ++j) v[4 * q + j] = rnd16<true>(v[4 * q + j]) + rv[j];  // conv output is fp16, then x + h
        }
      }
      if (p.out_f32 != nullptr) {
#pragma unroll
        for (int j = 0; j < NV; ++j) p.out_f32[oidx + j] = v[j];
      } else {
#pragma unroll
        for (int q = 0; q < NB; ++q) {
          uint2 ow;
          ow.x = (uint32_t)f2h(v[4 * q]) | ((uint32_t)f2h(v[4 * q + 1]) << 16);
          ow.y = (uint32_t)f2h(v[4 * q + 2]) | ((uint32_t)f2h(v[4 * q + 3]) << 16);
          *reinterpret_cast<uint2*>(p.out + oidx + 4 * q) = ow;
        }
      }
    } else {
      for (int j = 0; j < NV; ++j) {
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

}  // namespace

// Launch for wan_vae_conv3d (vae_ops.hip): the caller has checked KT x 3 x 3 (KT = 3 or 1), stride 1, pad 1, no interleave, 32-bit offsets.
// H, W: the OUTPUT frame (= the input's, or twice it with ups).
int wan_vae_conv3d_halo_launch(const uint16_t* x, const uint16_t* cache0, const uint16_t* cache1, const uint16_t* zero16, const uint16_t* w,
                               const uint16_t* bias, const uint16_t* res, uint16_t* out, float* out_f32, int Tin, int H, int W, int Cin, int Tout,
                               int Cout, int front, int Kp, int KT, int ups, hipStream_t stream) {
  HaloP p;
  p.x = x; p.cache0 = cache0; p.cache1 = cache1; p.zero16 = zero16; p.w = w; p.bias = bias; p.res = res; p.out = out; p.out_f32 = out_f32;
  p.Tin = Tin; p.H = H; p.W = W; p.Cin = Cin; p.Tout = Tout; p.Cout = Cout; p.CB = Cin / 32; p.Kp = Kp; p.front = front;
  p.ncache = cache1 ? (cache0 ? 2 : 1) : 0;
  // the tile width that pads Cout least: 96 for the 96 / 192 / 384-channel levels (exact), for 160 (192 against 256), for the 32-channel
  // heads and latents; 128 for 128, 256, 640, 1024 ...
  const bool n96 = (Cout + 95) / 96 * 96 < (Cout + 127) / 128 * 128;
  static const bool no16 = [] { const char* e = getenv("WAN_VAE_NO_HEAD16"); return e && e[0] == '1'; }();   // (A/B and bisecting: the 96-wide tile for the head)
  const bool n16 = Cout <= 16 && KT == 3 && !ups && !no16;   // the decoder's head (96 -> 3): one 16-channel tile, every wave on pixels
  p.tiles_h = (H + HT - 1) / HT; p.tiles_w = (W + HT - 1) / HT; p.tiles_x = n16 ? 1 : (n96 ? (Cout + 95) / 96 : (Cout + 127) / 128);
  const int64_t nwg = (int64_t)Tout * p.tiles_h * p.tiles_w * p.tiles_x;
  if (nwg == 0) return 0;
  WAN_REQUIRE(nwg < ((int64_t)1 << 31), "wan_vae_conv3d: grid too large");
#define HALO_GO(K, U, N) hipLaunchKernelGGL((conv3d_halo_kernel<K, U, N>), dim3((unsigned)nwg), dim3(512), 0, stream, p)
  if (n16) hipLaunchKernelGGL((conv3d_halo_kernel<3, false, 1, 1>), dim3((unsigned)nwg), dim3(512), 0, stream, p);
  else if (KT == 3) { if (n96) HALO_GO(3, false, 3); else HALO_GO(3, false, 4); }
  else if (ups) { if (n96) HALO_GO(1, true, 3); else HALO_GO(1, true, 4); }
  else { if (n96) HALO_GO(1, false, 3); else HALO_GO(1, false, 4); }
#undef HALO_GO
  WAN_LAUNCH_CHECK();
  return 0;
}
