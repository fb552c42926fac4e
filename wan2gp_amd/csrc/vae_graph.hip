// wan_vae_*: the whole Wan2.1 VAE encode / decode behind the C ABI (SURVEY.md section 8b `wan_vae_encode`, `wan_vae_decode_u8`).
//
// The layer graph and the causal feature-cache bookkeeping of Encoder3d / Decoder3d / WanVAE_.encode / .decode
// (models/wan/modules/vae.py:318-662) as host C++; every tensor operation is one of the VAE entry points of this library
// (wan_vae_conv3d, wan_vae_rmsnorm_silu, wan_gemm_f16, wan_vae_softmax, wan_vae_pack / _unpack, wan_vae_to_video) on fp16
// channels-last activations [T,H,W,C].  Weights are packed once at registration (wan_vae_set_conv: [Cout_p][Kp] fp16 with
// K = ((kt*KH+kh)*KW+kw)*Cin_p + c).  Activations, caches and scratch live in ONE caller-provided workspace, handed out by a
// first-fit region allocator; wan_vae_workspace_bytes runs the same graph in planning mode (no launches) and returns the peak.
// Architecture: WanVAE_(dim=96, z_dim=16, dim_mult=[1,2,4,4], num_res_blocks=2, temperal_downsample=[F,T,T]) (vae.py:906-918).
#include <map>
#include <string>
#include <vector>

#include "common.h"

extern "C" {
int wan_vae_conv3d_ex(const uint16_t* x, const uint16_t* cache0, const uint16_t* cache1, const uint16_t* w, const uint16_t* bias, const uint16_t* res,
                      uint16_t* out, float* out_f32, int Tin, int Hin, int Win, int Cin, int Tout, int Hout, int Wout, int Cout, int KT,
                      int KH, int KW, int st_t, int st_s, int front, int pad_s, int ups, int interleave, void* stream);
int wan_vae_rmsnorm_silu(const uint16_t* x, uint16_t* out, const uint16_t* gamma, int64_t npix, int C, int silu, void* stream);
int wan_gemm_f16(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, const uint16_t* bias, uint16_t* C, int64_t ldc,
                 int64_t M, int64_t N, int K, float scale, int transposed, void* stream);
int wan_vae_softmax(const uint16_t* S, uint16_t* P, int64_t rows, int L, int64_t ld, void* stream);
int wan_vae_pack(const float* in, uint16_t* out, const float* mul, const float* add, int C, int Cp, int64_t thw, void* stream);
int wan_vae_unpack(const uint16_t* in, float* out, const float* sub, const float* mul, int C, int Cs, int64_t thw, void* stream);
int wan_vae_to_video(const float* in, uint8_t* u8, float* f32, int T, int64_t hw, int Ttot, int t0, void* stream);
}

namespace {

constexpr int Z = 16;
const float VAE_MEAN[Z] = {-0.7571f, -0.7089f, -0.9113f, 0.1075f, -0.1745f, 0.9653f, -0.1517f, 1.5508f,
                           0.4134f, -0.0715f, 0.5517f, -0.3632f, -0.1922f, -0.9497f, 0.2503f, -0.2921f};
const float VAE_STD[Z] = {2.8184f, 1.4541f, 2.3275f, 2.6558f, 1.2196f, 1.7708f, 2.6052f, 2.0743f,
                          3.2687f, 2.1526f, 2.8652f, 1.5579f, 1.6382f, 1.1253f, 2.8251f, 1.9160f};

inline int pad32(int c) { return (c + 31) / 32 * 32; }

// Packed parameters wait in host memory until the first encode / decode uploads them (ensure_uploaded): registration and the
// workspace planner need no device.
struct VConv {
  uint16_t* w = nullptr;
  uint16_t* b = nullptr;
  std::vector<uint16_t> hw, hb;
  int cin = 0, cout = 0, kt = 1, kh = 1, kw = 1;
};
struct VAttn {
  int C = 0;
  uint16_t* wqkv = nullptr;
  uint16_t* bqkv = nullptr;
  std::vector<uint16_t> hw, hb;
};

// first-fit region allocator over the caller's workspace (offsets, 256-byte granules); planning mode only tracks the peak
struct Arena {
  char* base = nullptr;
  int64_t cap = 0, peak = 0;
  std::map<int64_t, int64_t> used;  // offset -> size
  std::map<int64_t, int> refs;      // offset -> owners: the graph's own handle + the causal caches that point at the tensor's frames (round 6)
  int64_t alloc(int64_t bytes) {
    bytes = (bytes + 255) / 256 * 256;
    int64_t pos = 0;
    for (auto& kv : used) {
      if (kv.first - pos >= bytes) break;
      pos = kv.first + kv.second;
    }
    used[pos] = bytes;
    refs[pos] = 1;
    if (pos + bytes > peak) peak = pos + bytes;
    return pos;
  }
  void retain(int64_t off) { ++refs[off]; }
  bool owned(int64_t off) const { return refs.count(off) != 0; }
  void free(int64_t off) {          // one owner less; the region returns to the allocator with the last
    auto it = refs.find(off);
    if (it != refs.end() && --it->second > 0) return;
    if (it != refs.end()) refs.erase(it);
    used.erase(off);
  }
};

// a tensor in the arena: fp16 [T,H,W,C] (or fp32 when f32)
struct Ten {
  int64_t off = -1;
  int T = 0, H = 0, W = 0, C = 0;
  bool f32 = false;
  bool valid() const { return off >= 0; }
  int64_t numel() const { return (int64_t)T * H * W * C; }
  int64_t frame() const { return (int64_t)H * W * C; }
};

}  // namespace

struct wan_vae {
  std::map<std::string, VConv> convs;
  std::map<std::string, uint16_t*> gamma;
  std::map<std::string, std::vector<uint16_t>> gamma_h;
  bool dirty = true;  // something registered since the last upload
  std::map<std::string, VAttn> attn;
  std::vector<void*> owned;
  float* mean_d = nullptr;  // scale[0]
  float* std_d = nullptr;   // 1 / scale[1]
  float* istd_d = nullptr;  // scale[1]
};

namespace {

// The causal cache of a convolution (vae.py:254-273: `cache_x = x[:, :, -CACHE_T:].clone()`, with the previous chunk's last frame in
// front when the chunk has one frame) as POINTERS (round 6): frame -1 and frame -2 are frames of the input
// tensors of this and the previous chunk, kept alive by the slot (Arena::retain) instead of copied -- 2,838 device-to-device copies and
// 5.2 % of a 720p x 81-frame decode in round 4's trace.  f1 = input frame -1, f0 = frame -2 (invalid: zeros).  `t` holds a copied cache
// for inputs the arena does not own (views of the latent / video tensor).
struct FrameRef {
  Ten owner;          // a tensor of the arena (owner.off is its allocation)
  int frame = 0;
  bool valid() const { return owner.valid(); }
};
struct CacheSlot {
  enum Kind { NONE, REP, TENSOR, REFS } kind = NONE;
  Ten t;
  FrameRef f0, f1;
};

struct ConvOpt {
  const Ten* cache = nullptr;
  const CacheSlot* slot = nullptr;   // kind REFS: the two frame pointers; kind TENSOR: its two contiguous frames
  const Ten* res = nullptr;
  bool out_f32 = false, ups = false, interleave = false;
  int st_t = 1, st_s = 1, front = -1, pad_s = -1;
  Ten* into = nullptr;  // write into this preallocated tensor instead of a new one
};

struct Graph {
  wan_vae* v;
  Arena ar;
  bool plan;
  void* stream;
  int rc = 0;

  char* p(const Ten& t, int64_t elem_off = 0) const { return ar.base + t.off + elem_off * (t.f32 ? 4 : 2); }
  uint16_t* h(const Ten& t, int64_t elem_off = 0) const { return reinterpret_cast<uint16_t*>(p(t, elem_off)); }

  Ten make(int T, int H, int W, int C, bool f32 = false) {
    Ten t;
    t.T = T; t.H = H; t.W = W; t.C = C; t.f32 = f32;
    t.off = ar.alloc(t.numel() * (f32 ? 4 : 2));
    if (!plan && t.off + t.numel() * (f32 ? 4 : 2) > ar.cap && rc == 0) {
      wan_set_error("wan_vae: workspace too small (need at least %lld bytes, have %lld)", (long long)ar.peak, (long long)ar.cap);
      rc = 1;
    }
    return t;
  }
  void drop(Ten& t) {
    if (t.valid()) ar.free(t.off);
    t.off = -1;
  }
  // a view of frames [t0, t0 + n) of a tensor (no ownership)
  static Ten frames(const Ten& t, int t0, int n) {
    Ten r = t;
    r.off = t.off + (int64_t)t0 * t.frame() * (t.f32 ? 4 : 2);
    r.T = n;
    return r;
  }
  void copy(const Ten& dst, int64_t dst_elem, const Ten& src, int64_t src_elem, int64_t n) {
    if (plan || rc) return;
    if (hipMemcpyAsync(p(dst, dst_elem), p(src, src_elem), (size_t)n * (src.f32 ? 4 : 2), hipMemcpyDeviceToDevice, as_stream(stream)) != hipSuccess) {
      wan_set_error("wan_vae: hipMemcpyAsync failed");
      rc = 2;
    }
  }
  void zero(const Ten& dst, int64_t elem, int64_t n) {
    if (plan || rc) return;
    if (hipMemsetAsync(p(dst, elem), 0, (size_t)n * (dst.f32 ? 4 : 2), as_stream(stream)) != hipSuccess) {
      wan_set_error("wan_vae: hipMemsetAsync failed");
      rc = 2;
    }
  }
  const VConv* conv_of(const std::string& name) {
    auto it = v->convs.find(name);
    if (it == v->convs.end()) {
      if (rc == 0) { wan_set_error("wan_vae: convolution '%s' was not registered (wan_vae_set_conv)", name.c_str()); rc = 1; }
      return nullptr;
    }
    return &it->second;
  }

  // _VaeNet.conv (wan2gp_amd/vae.py): output geometry of CausalConv3d / Resample convs, then wan_vae_conv3d
  Ten conv(const Ten& x, const std::string& name, ConvOpt o = ConvOpt()) {
    const VConv* c = conv_of(name);
    Ten out;
    if (!c) return out;
    if (x.C != c->cin && rc == 0) { wan_set_error("wan_vae: '%s' expects %d input channels, got %d", name.c_str(), c->cin, x.C); rc = 1; }
    const int front = o.front < 0 ? c->kt - 1 : o.front;      // causal: 2*padding[0] frames in front (vae.py:49-51)
    const int pad_s = o.pad_s < 0 ? c->kh / 2 : o.pad_s;
    const int He = o.ups ? 2 * x.H : x.H, We = o.ups ? 2 * x.W : x.W;
    int Ho = He, Wo = We;
    if (o.st_s == 2) {  // ZeroPad2d((0,1,0,1)) + stride 2 (vae.py:137-139)
      Ho = (He + 1 - c->kh) / 2 + 1;
      Wo = (We + 1 - c->kw) / 2 + 1;
    }
    const int To = (x.T + front - c->kt) / o.st_t + 1;
    if (o.into) out = *o.into;
    else if (o.interleave) out = make(2 * To, Ho, Wo, c->cout / 2);
    else out = make(To, Ho, Wo, c->cout, o.out_f32);
    if (!plan && rc == 0) {
      const uint16_t *c0 = nullptr, *c1 = nullptr;
      if (o.slot != nullptr && o.slot->kind == CacheSlot::REFS) {
        c1 = h(o.slot->f1.owner, (int64_t)o.slot->f1.frame * o.slot->f1.owner.frame());
        if (o.slot->f0.valid()) c0 = h(o.slot->f0.owner, (int64_t)o.slot->f0.frame * o.slot->f0.owner.frame());
      } else if (o.slot != nullptr && o.slot->kind == CacheSlot::TENSOR) {
        c0 = h(o.slot->t);
        c1 = h(o.slot->t, o.slot->t.frame());
      } else if (o.cache != nullptr) {
        c0 = h(*o.cache);
        c1 = h(*o.cache, o.cache->frame());
      }
      const int r = wan_vae_conv3d_ex(h(x), c0, c1, c->w, c->b, o.res ? h(*o.res) : nullptr,
                                      o.out_f32 ? nullptr : h(out), o.out_f32 ? reinterpret_cast<float*>(p(out)) : nullptr, x.T, x.H, x.W,
                                      x.C, To, Ho, Wo, c->cout, c->kt, c->kh, c->kw, o.st_t, o.st_s, front, pad_s, o.ups ? 1 : 0,
                                      o.interleave ? 1 : 0, stream);
      if (r) rc = r;
    }
    return out;
  }
  Ten norm(const Ten& x, const std::string& gname, bool silu = true) {
    Ten out = make(x.T, x.H, x.W, x.C);
    if (v->gamma_h.find(gname) == v->gamma_h.end()) {
      if (rc == 0) { wan_set_error("wan_vae: gamma '%s' was not registered (wan_vae_set_gamma)", gname.c_str()); rc = 1; }
      return out;
    }
    if (!plan && rc == 0) {
      const int r = wan_vae_rmsnorm_silu(h(x), h(out), v->gamma[gname], x.numel() / x.C, x.C, silu ? 1 : 0, stream);
      if (r) rc = r;
    }
    return out;
  }
  // cache_x bookkeeping (vae.py:256-263): the last 2 frames of [old ; x], as a NEW slot value.  x is a tensor of the arena: the slot
  // points at its frames and keeps it alive (no copy); x is a view of something else (the latent / video tensor): a copied cache, as before.
  CacheSlot next_cache(const Ten& x, const CacheSlot& old) {
    CacheSlot n;
    if (!ar.owned(x.off)) {
      n.kind = CacheSlot::TENSOR;
      n.t = make(2, x.H, x.W, x.C);
      const int64_t fr = x.frame();
      if (x.T >= 2) {
        copy(n.t, 0, x, (int64_t)(x.T - 2) * fr, 2 * fr);
      } else if (old.kind == CacheSlot::TENSOR) {
        copy(n.t, 0, old.t, fr, fr);
        copy(n.t, fr, x, (int64_t)(x.T - 1) * fr, fr);
      } else if (old.kind == CacheSlot::REFS) {
        copy(n.t, 0, old.f1.owner, (int64_t)old.f1.frame * fr, fr);
        copy(n.t, fr, x, (int64_t)(x.T - 1) * fr, fr);
      } else {
        zero(n.t, 0, fr);
        copy(n.t, fr, x, (int64_t)(x.T - 1) * fr, fr);
      }
      return n;
    }
    n.kind = CacheSlot::REFS;
    n.f1.owner = x; n.f1.frame = x.T - 1;
    ar.retain(x.off);
    if (x.T >= 2) {
      n.f0.owner = x; n.f0.frame = x.T - 2;
      ar.retain(x.off);
    } else if (old.kind == CacheSlot::REFS) {
      n.f0 = old.f1;
      ar.retain(old.f1.owner.off);
    } else if (old.kind == CacheSlot::TENSOR) {      // (a slot that changed from copies to pointers mid-stream: frame 1 of the copied pair)
      n.f0.owner = old.t; n.f0.frame = 1;
      ar.retain(old.t.off);
    }                                                 // else: the stream's first chunk -- frame -2 reads as zeros (f0 invalid)
    return n;
  }
  void release_cache(CacheSlot& s) {
    if (s.kind == CacheSlot::TENSOR) drop(s.t);
    if (s.kind == CacheSlot::REFS) {
      if (s.f0.valid()) ar.free(s.f0.owner.off);
      if (s.f1.valid()) ar.free(s.f1.owner.off);
    }
    s = CacheSlot();
  }
  void set_cache(CacheSlot& s, const CacheSlot& n) {
    release_cache(s);
    s = n;
  }
  static const CacheSlot* slot_arg(const CacheSlot& s) { return (s.kind == CacheSlot::TENSOR || s.kind == CacheSlot::REFS) ? &s : nullptr; }
  Ten cached_conv(const Ten& x, const std::string& name, std::vector<CacheSlot>& cache, int& idx, ConvOpt o = ConvOpt()) {
    CacheSlot cx = next_cache(x, cache[idx]);
    o.slot = slot_arg(cache[idx]);
    Ten y = conv(x, name, o);
    set_cache(cache[idx], cx);
    ++idx;
    return y;
  }
  // ResidualBlock (vae.py:214-268); consumes x
  Ten res_block(Ten x, const std::string& pfx, std::vector<CacheSlot>& cache, int& idx) {
    const bool has_sc = v->convs.count(pfx + "shortcut") != 0;
    Ten hsc = has_sc ? conv(x, pfx + "shortcut") : x;
    Ten y = x;
    const char* gi[2] = {"0", "3"};
    const char* ci[2] = {"2", "6"};
    for (int k = 0; k < 2; ++k) {
      Ten n = norm(y, pfx + "residual." + gi[k] + ".gamma");
      if (y.off != hsc.off && (k == 1 || has_sc)) drop(y);  // y == x is still the shortcut when there is no shortcut conv
      CacheSlot cx = next_cache(n, cache[idx]);
      ConvOpt o;
      o.slot = slot_arg(cache[idx]);
      if (k == 1) o.res = &hsc;
      y = conv(n, pfx + "residual." + ci[k], o);
      drop(n);
      set_cache(cache[idx], cx);
      ++idx;
    }
    drop(hsc);  // the shortcut conv's output, or x itself when there is none (with a shortcut conv x was released above)
    return y;
  }
  // AttentionBlock.forward (vae.py:294-315) per frame: tokens = h*w, one head of C channels; consumes x
  Ten attention_block(Ten x, const std::string& pfx) {
    auto it = v->attn.find(pfx);
    Ten out = make(x.T, x.H, x.W, x.C);
    if (it == v->attn.end()) {
      if (rc == 0) { wan_set_error("wan_vae: attention '%s' was not registered (wan_vae_set_attention)", pfx.c_str()); rc = 1; }
      return out;
    }
    const VAttn& a = it->second;
    const int C = a.C, L = x.H * x.W, Lp = (L + 63) / 64 * 64;
    if (L % 16 != 0 && rc == 0) { wan_set_error("wan_vae: attention needs h*w %% 16 == 0 at the lowest resolution (h*w = %d)", L); rc = 1; }
    Ten xn = norm(x, pfx + "norm.gamma", false);
    const float scale = (float)(1.0 / sqrt((double)C));  // 1 / math.sqrt(C) in double, rounded once (as the host graph passes it)
    for (int t = 0; t < x.T; ++t) {
      Ten qk = make(1, 1, L, 2 * C), vt = make(1, 1, C, Lp), S = make(1, 1, L, Lp), o = make(1, x.H, x.W, C);
      zero(vt, 0, vt.numel());
      if (!plan && rc == 0) {
        const uint16_t* xt = h(xn, (int64_t)t * xn.frame());
        int r = wan_gemm_f16(xt, C, a.wqkv, C, a.bqkv, h(qk), 2 * C, L, 2 * C, C, 1.0f, 0, stream);                    // [q | k] = x Wqk^T + b
        if (!r) r = wan_gemm_f16(xt, C, a.wqkv + (int64_t)2 * C * C, C, a.bqkv + 2 * C, h(vt), Lp, L, C, C, 1.0f, 1, stream);  // V^T
        if (!r) r = wan_gemm_f16(h(qk), 2 * C, h(qk, C), 2 * C, nullptr, h(S), Lp, L, L, C, scale, 0, stream);         // q k^T / sqrt(C)
        if (!r) r = wan_vae_softmax(h(S), h(S), L, L, Lp, stream);
        if (!r) r = wan_gemm_f16(h(S), Lp, h(vt), Lp, nullptr, h(o), C, L, C, Lp, 1.0f, 0, stream);
        if (r) rc = r;
      }
      Ten xt_res = frames(x, t, 1), dst = frames(out, t, 1);
      ConvOpt co;
      co.res = &xt_res;
      co.into = &dst;
      conv(o, pfx + "proj", co);
      drop(qk); drop(vt); drop(S); drop(o);
    }
    drop(xn);
    drop(x);
    return out;
  }
};

int n_cached(const wan_vae* v, const std::string& side) {
  int n = 0;
  for (auto& kv : v->convs)
    if (kv.first.compare(0, side.size(), side) == 0 && kv.second.kt == 3) ++n;
  return n;
}

// Decoder3d.forward (vae.py:486-538) on one latent frame [1,h,w,32]; x is a view (not consumed)
Ten decoder(Graph& g, const Ten& xin, std::vector<CacheSlot>& cache) {
  int idx = 0;
  Ten x = g.cached_conv(xin, "decoder.conv1", cache, idx);
  x = g.res_block(x, "decoder.middle.0.", cache, idx);
  x = g.attention_block(x, "decoder.middle.1.");
  x = g.res_block(x, "decoder.middle.2.", cache, idx);
  int li = 0;
  const bool tus[3] = {true, true, false};  // temperal_downsample reversed
  for (int i = 0; i < 4; ++i) {
    for (int k = 0; k < 3; ++k) {
      x = g.res_block(x, "decoder.upsamples." + std::to_string(li) + ".", cache, idx);
      ++li;
    }
    if (i != 3) {
      const std::string p = "decoder.upsamples." + std::to_string(li) + ".";
      if (tus[i]) {  // upsample3d (vae.py:151-189): the first chunk only marks the slot ('Rep'), later chunks run time_conv
        CacheSlot& s = cache[idx];
        if (s.kind == CacheSlot::NONE) {
          s.kind = CacheSlot::REP;
        } else {
          CacheSlot cx = g.next_cache(x, s);
          ConvOpt o;
          o.slot = Graph::slot_arg(s);
          o.interleave = true;
          o.pad_s = 0;
          Ten y = g.conv(x, p + "time_conv", o);
          g.drop(x);
          x = y;
          g.set_cache(s, cx);
        }
        ++idx;
      }
      ConvOpt o;
      o.ups = true;  // nearest-exact 2x + Conv2d 3x3 (vae.py:126-128)
      Ten y = g.conv(x, p + "resample.1", o);
      g.drop(x);
      x = y;
      ++li;
    }
  }
  Ten n = g.norm(x, "decoder.head.0.gamma");
  g.drop(x);
  ConvOpt o;
  o.out_f32 = true;
  Ten y = g.cached_conv(n, "decoder.head.2", cache, idx, o);
  g.drop(n);
  return y;
}

// Encoder3d.forward (vae.py:371-428) on a chunk [t,H,W,32]; x is a view (not consumed)
Ten encoder(Graph& g, const Ten& xin, std::vector<CacheSlot>& cache) {
  int idx = 0;
  Ten x = g.cached_conv(xin, "encoder.conv1", cache, idx);
  int li = 0;
  const bool tds[3] = {false, true, true};
  for (int i = 0; i < 4; ++i) {
    for (int k = 0; k < 2; ++k) {
      x = g.res_block(x, "encoder.downsamples." + std::to_string(li) + ".", cache, idx);
      ++li;
    }
    if (i != 3) {
      const std::string p = "encoder.downsamples." + std::to_string(li) + ".";
      ConvOpt o;
      o.st_s = 2;
      o.pad_s = 0;  // ZeroPad2d((0,1,0,1)) + stride 2
      Ten y = g.conv(x, p + "resample.1", o);
      g.drop(x);
      x = y;
      if (tds[i]) {  // downsample3d (vae.py:195-211): the slot keeps the chunk's last frame
        CacheSlot& s = cache[idx];
        // the slot points at the chunk's last frame (x stays alive through the pointer; the time_conv reads it as input frame -1)
        CacheSlot last;
        last.kind = CacheSlot::REFS;
        last.f1.owner = x; last.f1.frame = x.T - 1;
        g.ar.retain(x.off);
        if (s.kind == CacheSlot::NONE) {
          g.set_cache(s, last);
        } else {
          ConvOpt t;
          t.slot = &s;                           // f0 invalid: frame -2 is never read (front = 1)
          t.st_t = 2;
          t.front = 1;
          t.pad_s = 0;
          Ten z = g.conv(x, p + "time_conv", t);
          g.drop(x);
          x = z;
          g.set_cache(s, last);
        }
        ++idx;
      }
      ++li;
    }
  }
  x = g.res_block(x, "encoder.middle.0.", cache, idx);
  x = g.attention_block(x, "encoder.middle.1.");
  x = g.res_block(x, "encoder.middle.2.", cache, idx);
  Ten n = g.norm(x, "encoder.head.0.gamma");
  g.drop(x);
  Ten y = g.cached_conv(n, "encoder.head.2", cache, idx);
  g.drop(n);
  return y;
}

void drop_caches(Graph& g, std::vector<CacheSlot>& cache) {
  for (auto& s : cache) g.release_cache(s);
}

// WanVAE_.decode (vae.py:628-662): z [16,t,h,w] fp32 -> uint8 and/or fp32 [3,T,H,W]
int run_decode(wan_vae* v, const float* z, int t, int h, int w, uint8_t* u8, float* f32, void* ws, int64_t ws_bytes, void* stream,
               bool plan, int64_t* peak) {
  Graph g{v, Arena(), plan, stream};
  g.ar.base = static_cast<char*>(ws);
  g.ar.cap = ws_bytes;
  Ten zp = g.make(t, h, w, 32);
  if (!plan && g.rc == 0) {
    const int r = wan_vae_pack(z, g.h(zp), v->std_d, v->mean_d, Z, 32, (int64_t)t * h * w, stream);  // z / scale[1] + scale[0]
    if (r) g.rc = r;
  }
  Ten x = g.conv(zp, "conv2");  // 1x1x1, 16 -> 16 (padded to 32)
  g.drop(zp);
  const int T_out = (t - 1) * 4 + 1, H = h * 8, W = w * 8;
  std::vector<CacheSlot> cache(n_cached(v, "decoder."));
  int t0 = 0;
  for (int i = 0; i < t && g.rc == 0; ++i) {
    Ten xi = Graph::frames(x, i, 1);
    Ten y = decoder(g, xi, cache);  // fp32 [T_i, H, W, 3]
    if (!plan && g.rc == 0) {
      const int r = wan_vae_to_video(reinterpret_cast<const float*>(g.p(y)), u8, f32, y.T, (int64_t)H * W, T_out, t0, stream);
      if (r) g.rc = r;
    }
    t0 += y.T;
    g.drop(y);
  }
  drop_caches(g, cache);
  g.drop(x);
  if (g.rc == 0 && t0 != T_out) { wan_set_error("wan_vae_decode: produced %d frames, expected %d", t0, T_out); g.rc = 1; }
  if (peak) *peak = g.ar.peak;
  return g.rc;
}

// WanVAE_.encode (vae.py:586-625): video [3,T,H,W] fp32 -> normalised mu [16,t,h,w] fp32
int run_encode(wan_vae* v, const float* video, int T, int H, int W, float* mu, void* ws, int64_t ws_bytes, void* stream, bool plan,
               int64_t* peak) {
  Graph g{v, Arena(), plan, stream};
  g.ar.base = static_cast<char*>(ws);
  g.ar.cap = ws_bytes;
  Ten vp = g.make(T, H, W, 32);
  if (!plan && g.rc == 0) {
    const int r = wan_vae_pack(video, g.h(vp), nullptr, nullptr, 3, 32, (int64_t)T * H * W, stream);
    if (r) g.rc = r;
  }
  const int t = 1 + (T - 1) / 4, h = H / 8, w = W / 8;
  Ten enc = g.make(t, h, w, 32);
  std::vector<CacheSlot> cache(n_cached(v, "encoder."));
  int t0 = 0;
  for (int i = 0; i < t && g.rc == 0; ++i) {
    Ten xc = i == 0 ? Graph::frames(vp, 0, 1) : Graph::frames(vp, 1 + 4 * (i - 1), 4);
    Ten y = encoder(g, xc, cache);
    if (g.rc == 0 && (y.H != h || y.W != w || y.C != 32 || t0 + y.T > t)) {
      wan_set_error("wan_vae_encode: chunk %d produced [%d,%d,%d,%d], expected [*,%d,%d,32]", i, y.T, y.H, y.W, y.C, h, w);
      g.rc = 1;
    }
    if (g.rc == 0) g.copy(enc, (int64_t)t0 * enc.frame(), y, 0, y.numel());
    t0 += y.T;
    g.drop(y);
  }
  drop_caches(g, cache);
  g.drop(vp);
  Ten m = g.conv(enc, "conv1");  // 1x1x1 32 -> 32; mu = the first 16 channels
  g.drop(enc);
  if (!plan && g.rc == 0) {
    const int r = wan_vae_unpack(g.h(m), mu, v->mean_d, v->istd_d, Z, 32, (int64_t)t * h * w, stream);
    if (r) g.rc = r;
  }
  g.drop(m);
  if (peak) *peak = g.ar.peak;
  return g.rc;
}

template <typename T>
int upload(wan_vae* v, const std::vector<T>& host, T** dev) {
  void* d = nullptr;
  WAN_CHECK_HIP(hipMalloc(&d, host.size() * sizeof(T)));
  v->owned.push_back(d);
  WAN_CHECK_HIP(hipMemcpy(d, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
  *dev = static_cast<T*>(d);
  return 0;
}
inline uint16_t to_half(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }

int ensure_uploaded(wan_vae* v) {
  if (!v->dirty) return 0;
  if (!v->mean_d) {
    std::vector<float> mean(VAE_MEAN, VAE_MEAN + Z), sd(Z), isd(Z);
    for (int i = 0; i < Z; ++i) {
      isd[i] = 1.0f / VAE_STD[i];  // scale[1] as the reference builds it (fp32 reciprocal, vae.py:956-958)
      sd[i] = 1.0f / isd[i];       // decode multiplies by 1 / scale[1]: the reciprocal of the reciprocal, not std itself
    }
    if (int rc = upload(v, mean, &v->mean_d)) return rc;
    if (int rc = upload(v, sd, &v->std_d)) return rc;
    if (int rc = upload(v, isd, &v->istd_d)) return rc;
  }
  for (auto& kv : v->convs) {
    VConv& c = kv.second;
    if (c.w) continue;
    if (int rc = upload(v, c.hw, &c.w)) return rc;
    if (int rc = upload(v, c.hb, &c.b)) return rc;
    std::vector<uint16_t>().swap(c.hw);
    std::vector<uint16_t>().swap(c.hb);
  }
  for (auto& kv : v->gamma_h)
    if (v->gamma.find(kv.first) == v->gamma.end()) {
      uint16_t* d = nullptr;
      if (int rc = upload(v, kv.second, &d)) return rc;
      v->gamma[kv.first] = d;
    }
  for (auto& kv : v->attn) {
    VAttn& a = kv.second;
    if (a.wqkv) continue;
    if (int rc = upload(v, a.hw, &a.wqkv)) return rc;
    if (int rc = upload(v, a.hb, &a.bqkv)) return rc;
    std::vector<uint16_t>().swap(a.hw);
    std::vector<uint16_t>().swap(a.hb);
  }
  v->dirty = false;
  return 0;
}

}  // namespace

extern "C" int wan_vae_create(wan_vae** out) {
  WAN_REQUIRE(out != nullptr, "wan_vae_create: null out");
  *out = new wan_vae();
  return 0;
}

extern "C" void wan_vae_destroy(wan_vae* v) {
  if (!v) return;
  for (void* p : v->owned) (void)hipFree(p);
  delete v;
}

// weight [cout, cin, kt, kh, kw] fp32 (host), bias [cout] fp32 (host) or NULL; cout_pad: 0, or the padded output width
// (the latent-side 1x1x1 conv "conv2" feeds a 32-channel-padded tensor).  Packing of wan2gp_amd/vae.py:_Conv.
extern "C" int wan_vae_set_conv(wan_vae* v, const char* name, const float* w, int cout, int cin, int kt, int kh, int kw,
                                const float* bias, int cout_pad) {
  WAN_REQUIRE(v && name && w && cout > 0 && cin > 0 && kt > 0 && kh > 0 && kw > 0, "wan_vae_set_conv: bad arguments");
  WAN_REQUIRE(cout_pad == 0 || cout_pad >= cout, "wan_vae_set_conv: cout_pad %d < cout %d", cout_pad, cout);
  const int cin_p = pad32(cin), cout_p = cout_pad ? cout_pad : cout;
  const int64_t K = (int64_t)kt * kh * kw * cin_p, Kp = (K + 63) / 64 * 64;
  std::vector<uint16_t> wp((size_t)cout_p * Kp, 0), bp((size_t)cout_p, 0);
  for (int o = 0; o < cout; ++o)
    for (int c = 0; c < cin; ++c)
      for (int a = 0; a < kt; ++a)
        for (int b = 0; b < kh; ++b)
          for (int d = 0; d < kw; ++d)
            wp[(size_t)o * Kp + (((int64_t)a * kh + b) * kw + d) * cin_p + c] =
                to_half(w[((((int64_t)o * cin + c) * kt + a) * kh + b) * kw + d]);
  if (bias)
    for (int o = 0; o < cout; ++o) bp[o] = to_half(bias[o]);
  WAN_REQUIRE(v->convs.find(name) == v->convs.end(), "wan_vae_set_conv: '%s' registered twice", name);
  VConv& c = v->convs[name];
  c.cin = cin_p; c.cout = cout_p; c.kt = kt; c.kh = kh; c.kw = kw;
  c.hw.swap(wp);
  c.hb.swap(bp);
  v->dirty = true;
  return 0;
}

extern "C" int wan_vae_set_gamma(wan_vae* v, const char* name, const float* g, int C) {
  WAN_REQUIRE(v && name && g && C > 0, "wan_vae_set_gamma: bad arguments");
  std::vector<uint16_t> hp((size_t)C);
  for (int i = 0; i < C; ++i) hp[i] = to_half(g[i]);
  WAN_REQUIRE(v->gamma_h.find(name) == v->gamma_h.end(), "wan_vae_set_gamma: '%s' registered twice", name);
  v->gamma_h[name].swap(hp);
  v->dirty = true;
  return 0;
}

// to_qkv of an AttentionBlock: weight [3C, C] and bias [3C] fp32 (host); prefix e.g. "decoder.middle.1."
extern "C" int wan_vae_set_attention(wan_vae* v, const char* prefix, const float* wqkv, const float* bqkv, int C) {
  WAN_REQUIRE(v && prefix && wqkv && bqkv && C > 0, "wan_vae_set_attention: bad arguments");
  std::vector<uint16_t> wp((size_t)3 * C * C), bp((size_t)3 * C);
  for (size_t i = 0; i < wp.size(); ++i) wp[i] = to_half(wqkv[i]);
  for (size_t i = 0; i < bp.size(); ++i) bp[i] = to_half(bqkv[i]);
  WAN_REQUIRE(v->attn.find(prefix) == v->attn.end(), "wan_vae_set_attention: '%s' registered twice", prefix);
  VAttn& a = v->attn[prefix];
  a.C = C;
  a.hw.swap(wp);
  a.hb.swap(bp);
  v->dirty = true;
  return 0;
}

extern "C" int64_t wan_vae_workspace_bytes(wan_vae* v, int decode, int t, int h, int w) {
  if (!v || t < 1 || h < 1 || w < 1) return -1;
  int64_t peak = 0;
  const int rc = decode ? run_decode(v, nullptr, t, h, w, nullptr, nullptr, nullptr, 0, nullptr, true, &peak)
                        : run_encode(v, nullptr, t, h, w, nullptr, nullptr, 0, nullptr, true, &peak);
  return rc ? -1 : peak;
}

extern "C" int wan_vae_decode(wan_vae* v, const float* z, int t, int h, int w, uint8_t* u8, float* f32, void* workspace,
                              int64_t workspace_bytes, void* stream) {
  WAN_REQUIRE(v && z && (u8 || f32) && workspace, "wan_vae_decode: null argument");
  WAN_REQUIRE(t >= 1 && h >= 1 && w >= 1, "wan_vae_decode: bad latent shape [16,%d,%d,%d]", t, h, w);
  if (int rc = ensure_uploaded(v)) return rc;
  return run_decode(v, z, t, h, w, u8, f32, workspace, workspace_bytes, stream, false, nullptr);
}

extern "C" int wan_vae_encode(wan_vae* v, const float* video, int T, int H, int W, float* mu, void* workspace,
                              int64_t workspace_bytes, void* stream) {
  WAN_REQUIRE(v && video && mu && workspace, "wan_vae_encode: null argument");
  WAN_REQUIRE(T >= 1 && (T - 1) % 4 == 0 && H % 8 == 0 && W % 8 == 0 && H >= 8 && W >= 8,
              "wan_vae_encode: video [3,%d,%d,%d] needs T = 4k + 1 and H, W multiples of 8", T, H, W);
  if (int rc = ensure_uploaded(v)) return rc;
  return run_encode(v, video, T, H, W, mu, workspace, workspace_bytes, stream, false, nullptr);
}
