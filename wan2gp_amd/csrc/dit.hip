// WanModel.forward (t2v / i2v2_2 path) as a resident-weights C++ driver over the HIP kernels.
// Replaces models/wan/modules/model.py:1485-2098 (forward), :575-724 (block), with mmgp's weight
// streaming deleted: every weight of both experts stays in HBM.  One call enqueues the whole
// forward on the caller's stream; the only host interaction is the between-blocks poll
// (model.py:1994-1998) and, under sequence parallelism, the K / V^T all-gather callback.
#include <stdarg.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include <stdlib.h>

#include "common.h"

// ---- error state ---------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";
void wan_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* wan_last_error(void) { return g_err; }
extern "C" int wan_version(void) { return 8; }
extern "C" int wan_device_cus(void) {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
  return n;
}

// internal (non-ABI) entry points from the other translation units
int wan_patch_embed_range(const float* x, const float* y, const float* w, const float* bias, bf16_t* out, int B, int Cin,
                          int Cy, int F, int H, int W, int d, int64_t tok0, int64_t ntok, void* stream);
int wan_head_range(const bf16_t* x, const float* hmod, const bf16_t* e, const float* w, const float* bias, bf16_t* tmp,
                   float* out, int B, int F, int Hg, int Wg, int d, float eps, int64_t tok0, int64_t ntok,
                   int token_major_out, int64_t e_rows_per_batch, int nout, void* stream);
int wan_sinusoid_val(float t, bf16_t* out, int dim, void* stream);
int wan_set_f32(float* p, float v, void* stream);
// wan_dit_forward_graph: the timestep of the forward being enqueued lives in device memory (one float) instead of a launch argument,
// so that the captured launch list can be replayed for another t (sinusoid_kernel and sinusoid_val_kernel do the same arithmetic).  The
// pointer travels as dit_forward_impl's `t_dev` argument (round 6: it was a process-wide static, read by whatever forward overlapped).

// ---- optional per-kernel-class timing with HIP events on the launch stream (bench.py roofline) ----
// Off by default.  When enabled, wan_dit_forward brackets the launches of each class with a
// hipEvent pair recorded on the caller's stream; wan_prof_collect() synchronises and sums them.
enum { PROF_SELF_ATTN = 0, PROF_CROSS_ATTN = 1, PROF_GEMM = 2, PROF_ROWOPS = 3, PROF_NCLASS = 4 };
struct ProfPair {
  hipEvent_t a, b;
  int cls;
};
static bool g_prof_on = false;
static std::vector<ProfPair> g_prof;
static uint64_t* g_prof_declined = nullptr;  // device [2]: self-attention workgroups the bounded loop declined / launched, while profiling
extern "C" int wan_prof_enable(int on) {
  g_prof_on = on != 0;
  if (g_prof_on) {
    if (g_prof_declined == nullptr) WAN_CHECK_HIP(hipMalloc((void**)&g_prof_declined, 16));
    WAN_CHECK_HIP(hipMemset(g_prof_declined, 0, 16));
  }
  for (auto& p : g_prof) {
    (void)hipEventDestroy(p.a);
    (void)hipEventDestroy(p.b);
  }
  g_prof.clear();
  return 0;
}
extern "C" int wan_prof_collect(int cls, double* total_ms, int* count) {
  WAN_REQUIRE(total_ms && count && cls >= 0 && cls < PROF_NCLASS, "wan_prof_collect: bad args");
  double t = 0;
  int n = 0;
  for (auto& p : g_prof) {
    if (p.cls != cls) continue;
    WAN_CHECK_HIP(hipEventSynchronize(p.b));
    float ms = 0;
    WAN_CHECK_HIP(hipEventElapsedTime(&ms, p.a, p.b));
    t += ms;
    ++n;
  }
  *total_ms = t;
  *count = n;
  return 0;
}
extern "C" int wan_prof_attention_declined(int64_t* declined, int64_t* total) {
  WAN_REQUIRE(declined && total, "wan_prof_attention_declined: bad args");
  uint64_t h[2] = {0, 0};
  if (g_prof_declined != nullptr) WAN_CHECK_HIP(hipMemcpy(h, g_prof_declined, 16, hipMemcpyDeviceToHost));  // synchronises with the device
  *declined = (int64_t)h[0];
  *total = (int64_t)h[1];
  return 0;
}
struct ProfScope {
  bool on;
  hipEvent_t b;
  hipStream_t st;
  ProfScope(int cls, hipStream_t s) : on(g_prof_on), st(s) {
    if (!on) return;
    ProfPair p;
    p.cls = cls;
    if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) {
      on = false;
      return;
    }
    (void)hipEventRecord(p.a, st);
    b = p.b;
    g_prof.push_back(p);
  }
  ~ProfScope() {
    if (on) (void)hipEventRecord(b, st);
  }
};

constexpr int CLIP_TOK = 257, CLIP_DIM = 1280, CLIP_LDV = 320;

struct Tensor {
  const void* ptr;
  int dtype;  // 0 bf16, 1 fp32, 2 fp8 e4m3fn (Linear weights of scaled-fp8 checkpoints)
  int64_t numel;
};

struct Lin {
  const bf16_t* w = nullptr;   // bf16 weight [N, K] ...
  const bf16_t* b = nullptr;
  const uint8_t* w8 = nullptr; // ... or fp8 e4m3fn weight + `<name>.scale_weight` (QLinearScaledFP8, shared/qtypes/scaled_fp8.py:563-637)
  const float* ws = nullptr;
  int ns = 0;                  // 1 (per tensor) or N (per output row)
};
struct Attn {
  Lin q, k, v, o;
  const bf16_t* nq = nullptr;
  const bf16_t* nk = nullptr;
};
struct Layer {
  const bf16_t* mod = nullptr;
  Attn self, cross;
  Lin kimg, vimg;                 // Wan2.1 i2v: WanI2VCrossAttention.k_img / v_img (model.py:460-463)
  const bf16_t* nkimg = nullptr;  //             norm_k_img
  Lin before, after;              // VACE context block: before_proj (block 0 only) / after_proj (model.py:805-812)
  const bf16_t* n3w = nullptr;
  const bf16_t* n3b = nullptr;
  const float* n3w32 = nullptr;   // mixed-precision plan: norm3 under the fp32 lock (model.py:1342-1346)
  const float* n3b32 = nullptr;
  Lin f0, f2;
};

struct wan_ctx {
  wan_dit_config cfg;
  std::map<std::string, Tensor> weights;
  bool resolved = false;
  bool any_fp8 = false;
  std::vector<Layer> layers;
  const float* pe_w = nullptr;
  const float* pe_b = nullptr;
  Lin te0, te2, tm0, tm2, tp1;
  // `mixed_precision_transformer` (any2video.py:190 -> lock_layers_dtypes(torch.float32), model.py:1330-1371): the context runs that plan
  // when 'time_projection.1.weight' was registered as fp32 -- the reference's own rule (modulation_dtype = time_projection[1].weight.dtype,
  // model.py:1545).  Then the time MLP, the projection and every norm3 are fp32 tensors, the residual stream / e / e0 live in fp32 and the
  // row kernels of csrc/mixed_ops.hip replace the bf16 plan's (fused GEMM epilogues become separate passes).
  bool mixed = false;
  const float *mx_tm0w = nullptr, *mx_tm0b = nullptr, *mx_tm2w = nullptr, *mx_tm2b = nullptr, *mx_tp1w = nullptr, *mx_tp1b = nullptr;
  const float* head_mod = nullptr;
  const float* head_w = nullptr;
  const float* head_b = nullptr;
  // Wan2.1 i2v (model_type 'i2v'): img_emb = MLPProj(1280, dim) (model.py:868-889) and the projected CLIP tokens
  bool has_img = false;
  bool has_flf = false;          // flf2v: img_emb carries emb_pos, clip_fea holds TWO images (model.py:878-887)
  const bf16_t* ie_pos = nullptr;  // img_emb.emb_pos [514, 1280]
  const bf16_t *ie_ln0w = nullptr, *ie_ln0b = nullptr, *ie_ln4w = nullptr, *ie_ln4b = nullptr;
  Lin ie1, ie3;
  // VACE (model.py:790-828, :1178-1206): context blocks attached to the main blocks listed in vace_layers
  std::vector<int> vace_layers;   // main-block index of context block n
  std::vector<int> vace_at;       // main-block index -> n or -1
  std::vector<Layer> vlayers;
  const float* vpe_w = nullptr;   // vace_patch_embedding as fp32 copies of the bf16 parameters
  const float* vpe_b = nullptr;
  int vace_in_dim = 0;
  int vace_max_ctx = 1;           // hint-stream sets the workspace holds (wan_dit_set_vace_contexts): contexts mixed in one call
  bf16_t* clip_ctx = nullptr;   // [257, dim], owned; filled by wan_dit_set_clip
  bf16_t* clip_tmp = nullptr;   // 2 x [257, 1280] scratch, owned
  bool clip_set = false;
  // wan_dit_forward_graph: captured launch lists (hipGraphExec) keyed by everything a forward's launches depend on except t
  struct GraphEntry {
    std::vector<uint8_t> key;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;   // nullptr: the key has been seen (and run eagerly) once; the next call captures
    bool bad = false;                // capture or instantiation failed once: this key stays on the eager path
    uint64_t last_use = 0;
  };
  std::vector<GraphEntry> graphs;
  uint64_t graph_clock = 0;
  float* t_dev = nullptr;           // owned: the timestep of a replayed forward
  hipStream_t cap_stream = nullptr; // owned: the stream a launch list is captured on (capture is not allowed on the legacy default stream)
  // Text-side tensors kept across forwards (round 6; wan_dit_args.context_key).  The text context does not change during a generation,
  // and neither do the weights: the text embedding and every block's cross-attention K (Linear + RMSNorm) and V^T are the SAME tensors in
  // every step -- 3 launches per block and 3 per forward recomputed 2 x 30 times per video by the reference (model.py:1856, :421-433), and
  // by every rank of a sequence-parallel world (12 ms of a 1,194-ms rank step at 14B-720p, profiles/r06_rank_world8_kernel_trace_run01.json).
  // With a non-zero context_key the forward leaves them in buffers of the context's own (2 x S x text_len x dim x 2 B per block and tensor:
  // 1.7 GB per 14B expert, of 288) and a later forward with the same key, stream layout and weights reads them back: same kernels on the
  // same inputs, so the same bits.  Two keys are kept (the conditional and the unconditional stream of sequential single passes).
  struct TextCache {
    uint64_t key = 0;
    int S = 0, TLx = 0, LDVx = 0;
    bool valid = false;
    uint64_t last_use = 0;
    std::vector<bf16_t*> ck, cvt;    // per block (main blocks, then VACE context blocks): owned
    size_t elems = 0;                // elements of one ck / cvt buffer
  };
  TextCache tcache[2];
  uint64_t tcache_clock = 0;
};
static void text_cache_free(wan_ctx::TextCache& t) {
  for (auto* p : t.ck) if (p) (void)hipFree(p);
  for (auto* p : t.cvt) if (p) (void)hipFree(p);
  t = wan_ctx::TextCache();
}
extern "C" int wan_dit_create(const wan_dit_config* cfg, wan_ctx** out) {
  WAN_REQUIRE(cfg && out, "wan_dit_create: null argument");
  WAN_REQUIRE(cfg->dim % cfg->num_heads == 0 && cfg->dim / cfg->num_heads == 128,
              "wan_dit_create: head_dim must be 128 (dim=%d heads=%d)", cfg->dim, cfg->num_heads);
  WAN_REQUIRE(cfg->dim % 64 == 0 && cfg->ffn_dim % 64 == 0 && cfg->text_dim % 64 == 0 && cfg->freq_dim % 8 == 0,
              "wan_dit_create: dims must be multiples of 64");
  // latents have out_dim channels (16; 48 for the ti2v 5B model, configs/ti2v_2_2.json); in_dim - out_dim more come from y
  WAN_REQUIRE(cfg->out_dim >= 4 && cfg->out_dim % 4 == 0 && cfg->out_dim <= 256 && cfg->in_dim >= cfg->out_dim,
              "wan_dit_create: out_dim=%d must be a multiple of 4 in [4, 256] and in_dim=%d >= out_dim", cfg->out_dim, cfg->in_dim);
  wan_ctx* c = new wan_ctx();
  c->cfg = *cfg;
  *out = c;
  return 0;
}
// captured launch lists hold the pointers and the launch order of the state they were captured in: every setter that changes either drops them
static void drop_graphs(wan_ctx* ctx) {
  for (auto& e : ctx->graphs) {
    if (e.exec) (void)hipGraphExecDestroy(e.exec);
    if (e.graph) (void)hipGraphDestroy(e.graph);
  }
  ctx->graphs.clear();
}
extern "C" void wan_dit_destroy(wan_ctx* ctx) {
  if (ctx) {
    if (ctx->clip_ctx) (void)hipFree(ctx->clip_ctx);
    if (ctx->clip_tmp) (void)hipFree(ctx->clip_tmp);
    drop_graphs(ctx);
    for (auto& t : ctx->tcache) text_cache_free(t);
    if (ctx->t_dev) (void)hipFree(ctx->t_dev);
    if (ctx->cap_stream) (void)hipStreamDestroy(ctx->cap_stream);
  }
  delete ctx;
}

extern "C" int wan_dit_set_weight(wan_ctx* ctx, const char* name, const void* ptr, int dtype, int64_t numel) {
  WAN_REQUIRE(ctx && name && ptr, "wan_dit_set_weight: null argument");
  WAN_REQUIRE(dtype == 0 || dtype == 1 || dtype == 2, "wan_dit_set_weight: dtype must be 0 (bf16), 1 (fp32) or 2 (fp8 e4m3fn)");
  WAN_REQUIRE((((uintptr_t)ptr) & 15) == 0, "wan_dit_set_weight: %s is not 16-byte aligned", name);
  ctx->weights[name] = Tensor{ptr, dtype, numel};
  ctx->resolved = false;
  drop_graphs(ctx);
  for (auto& t : ctx->tcache) t.valid = false;    // (computed with the old weights)
  return 0;
}

static int get_w(wan_ctx* c, const std::string& name, int dtype, int64_t numel, const void** out) {
  auto it = c->weights.find(name);
  WAN_REQUIRE(it != c->weights.end(), "missing weight '%s'", name.c_str());
  WAN_REQUIRE(it->second.dtype == dtype, "weight '%s' has dtype %d, expected %d (0=bf16,1=fp32; model.py:1330-1371)",
              name.c_str(), it->second.dtype, dtype);
  WAN_REQUIRE(it->second.numel == numel, "weight '%s' has %lld elements, expected %lld", name.c_str(),
              (long long)it->second.numel, (long long)numel);
  *out = it->second.ptr;
  return 0;
}
#define GETB(field, name, n)                                        \
  do {                                                              \
    const void* _p;                                                 \
    if (int _rc = get_w(c, name, 0, n, &_p)) return _rc;            \
    field = (const bf16_t*)_p;                                      \
  } while (0)
#define GETF(field, name, n)                                        \
  do {                                                              \
    const void* _p;                                                 \
    if (int _rc = get_w(c, name, 1, n, &_p)) return _rc;            \
    field = (const float*)_p;                                       \
  } while (0)

static int get_lin(wan_ctx* c, Lin& l, const std::string& prefix, int64_t out_f, int64_t in_f) {
  l = Lin();
  auto it = c->weights.find(prefix + ".weight");
  if (it != c->weights.end() && it->second.dtype == 2) {
    // scaled-fp8 Linear: fp8 weight + fp32 scale (scalar, [N] or [N, 1]) registered as `<prefix>.scale_weight`
    WAN_REQUIRE(it->second.numel == out_f * in_f, "weight '%s.weight' has %lld elements, expected %lld", prefix.c_str(),
                (long long)it->second.numel, (long long)(out_f * in_f));
    WAN_REQUIRE(in_f % 128 == 0 && out_f % 16 == 0, "fp8 Linear '%s': in_features %lld must be a multiple of 128, out_features %lld of 16",
                prefix.c_str(), (long long)in_f, (long long)out_f);
    auto is = c->weights.find(prefix + ".scale_weight");
    WAN_REQUIRE(is != c->weights.end() && is->second.dtype == 1 && (is->second.numel == 1 || is->second.numel == out_f),
                "fp8 Linear '%s' needs '%s.scale_weight': fp32, 1 or %lld elements", prefix.c_str(), prefix.c_str(), (long long)out_f);
    l.w8 = (const uint8_t*)it->second.ptr;
    l.ws = (const float*)is->second.ptr;
    l.ns = (int)is->second.numel;
    c->any_fp8 = true;
  } else {
    GETB(l.w, prefix + ".weight", out_f * in_f);
  }
  GETB(l.b, prefix + ".bias", out_f);
  return 0;
}

// the mixed-precision plan is chosen by the registered weights, like the reference chooses its modulation dtype (model.py:1545)
static bool ctx_is_mixed(const wan_ctx* c) {
  auto it = c->weights.find("time_projection.1.weight");
  return it != c->weights.end() && it->second.dtype == 1;
}

static int load_layer(wan_ctx* c, Layer& L, const std::string& p) {
  const int64_t d = c->cfg.dim, f = c->cfg.ffn_dim;
  GETB(L.mod, p + "modulation", 6 * d);
  for (int a = 0; a < 2; ++a) {
    Attn& A = a == 0 ? L.self : L.cross;
    const std::string ap = p + (a == 0 ? "self_attn." : "cross_attn.");
    if (int rc = get_lin(c, A.q, ap + "q", d, d)) return rc;
    if (int rc = get_lin(c, A.k, ap + "k", d, d)) return rc;
    if (int rc = get_lin(c, A.v, ap + "v", d, d)) return rc;
    if (int rc = get_lin(c, A.o, ap + "o", d, d)) return rc;
    GETB(A.nq, ap + "norm_q.weight", d);
    GETB(A.nk, ap + "norm_k.weight", d);
  }
  if (c->mixed) {
    GETF(L.n3w32, p + "norm3.weight", d);
    GETF(L.n3b32, p + "norm3.bias", d);
  } else {
    GETB(L.n3w, p + "norm3.weight", d);
    GETB(L.n3b, p + "norm3.bias", d);
  }
  if (int rc = get_lin(c, L.f0, p + "ffn.0", f, d)) return rc;
  if (int rc = get_lin(c, L.f2, p + "ffn.2", d, f)) return rc;
  return 0;
}

static int resolve(wan_ctx* c) {
  if (c->resolved) return 0;
  c->any_fp8 = false;
  const wan_dit_config& g = c->cfg;
  const int64_t d = g.dim;
  GETF(c->pe_w, "patch_embedding.weight", d * g.in_dim * 4);
  GETF(c->pe_b, "patch_embedding.bias", d);
  if (int rc = get_lin(c, c->te0, "text_embedding.0", d, g.text_dim)) return rc;
  if (int rc = get_lin(c, c->te2, "text_embedding.2", d, d)) return rc;
  c->mixed = ctx_is_mixed(c);
  if (c->mixed) {
    GETF(c->mx_tm0w, "time_embedding.0.weight", d * g.freq_dim);
    GETF(c->mx_tm0b, "time_embedding.0.bias", d);
    GETF(c->mx_tm2w, "time_embedding.2.weight", d * d);
    GETF(c->mx_tm2b, "time_embedding.2.bias", d);
    GETF(c->mx_tp1w, "time_projection.1.weight", 6 * d * d);
    GETF(c->mx_tp1b, "time_projection.1.bias", 6 * d);
  } else {
    if (int rc = get_lin(c, c->tm0, "time_embedding.0", d, g.freq_dim)) return rc;
    if (int rc = get_lin(c, c->tm2, "time_embedding.2", d, d)) return rc;
    if (int rc = get_lin(c, c->tp1, "time_projection.1", 6 * d, d)) return rc;
  }
  GETF(c->head_mod, "head.modulation", 2 * d);
  GETF(c->head_w, "head.head.weight", (int64_t)4 * g.out_dim * d);
  GETF(c->head_b, "head.head.bias", 4 * g.out_dim);
  c->layers.assign(g.num_layers, Layer());
  for (int i = 0; i < g.num_layers; ++i)
    if (int rc = load_layer(c, c->layers[i], "blocks." + std::to_string(i) + ".")) return rc;
  // VACE context blocks: the same block structure under vace_blocks.N. plus the two projections
  c->vlayers.assign(c->vace_layers.size(), Layer());
  c->vace_at.assign(g.num_layers, -1);
  for (size_t n = 0; n < c->vace_layers.size(); ++n) {
    const std::string p = "vace_blocks." + std::to_string(n) + ".";
    if (int rc = load_layer(c, c->vlayers[n], p)) return rc;
    if (n == 0)
      if (int rc = get_lin(c, c->vlayers[n].before, p + "before_proj", d, d)) return rc;
    if (int rc = get_lin(c, c->vlayers[n].after, p + "after_proj", d, d)) return rc;
    c->vace_at[c->vace_layers[n]] = (int)n;
  }
  if (!c->vace_layers.empty()) {
    auto it = c->weights.find("vace_patch_embedding.weight");
    WAN_REQUIRE(it != c->weights.end() && it->second.dtype == 1 && it->second.numel % (d * 4) == 0,
                "VACE: 'vace_patch_embedding.weight' must be registered as fp32 [dim, vace_in_dim, 1, 2, 2]");
    c->vace_in_dim = (int)(it->second.numel / (d * 4));
    c->vpe_w = (const float*)it->second.ptr;
    GETF(c->vpe_b, "vace_patch_embedding.bias", d);
  }
  // Wan2.1 i2v checkpoints carry the CLIP branch (img_emb + per-block k_img / v_img / norm_k_img)
  c->has_img = c->weights.count("img_emb.proj.1.weight") != 0;
  if (c->has_img) {
    GETB(c->ie_ln0w, "img_emb.proj.0.weight", CLIP_DIM);
    GETB(c->ie_ln0b, "img_emb.proj.0.bias", CLIP_DIM);
    if (int rc = get_lin(c, c->ie1, "img_emb.proj.1", CLIP_DIM, CLIP_DIM)) return rc;
    if (int rc = get_lin(c, c->ie3, "img_emb.proj.3", d, CLIP_DIM)) return rc;
    GETB(c->ie_ln4w, "img_emb.proj.4.weight", d);
    GETB(c->ie_ln4b, "img_emb.proj.4.bias", d);
    c->has_flf = c->weights.count("img_emb.emb_pos") != 0;   // MLPProj(flf_pos_emb=True): the flf2v_720p checkpoint
    if (c->has_flf) GETB(c->ie_pos, "img_emb.emb_pos", 2 * CLIP_TOK * CLIP_DIM);
    for (int i = 0; i < g.num_layers; ++i) {
      Layer& L = c->layers[i];
      const std::string ap = "blocks." + std::to_string(i) + ".cross_attn.";
      if (int rc = get_lin(c, L.kimg, ap + "k_img", d, d)) return rc;
      if (int rc = get_lin(c, L.vimg, ap + "v_img", d, d)) return rc;
      GETB(L.nkimg, ap + "norm_k_img.weight", d);
    }
  }
  c->resolved = true;
  return 0;
}

// ---- workspace carving -----------------------------------------------------------------------------
struct Carve {
  char* base;
  int64_t off = 0;
  explicit Carve(void* b) : base((char*)b) {}
  template <typename T>
  T* take(int64_t n) {
    T* p = base ? (T*)(base + off) : nullptr;
    off += ((n * (int64_t)sizeof(T) + 255) / 256) * 256;
    return p;
  }
};

struct Bufs {
  bf16_t *x, *xm, *q, *k, *vt, *h, *ctx_h, *ctx_e, *ck, *cvt, *sinus, *e_h, *e, *e_s, *e0, *kfull, *vtfull, *ckimg, *cvtimg, *vc, *vskip;
  uint8_t* xq;  // fp8 checkpoints: quantised activations, one slot per stream (q8_slot bytes each), and their scale pairs
  float* qws;
  int64_t q8_slot;
  float* raw;   // sequence parallelism: partial attention sums of the local segment (wan_attention_raw_words)
  float* kmax;  // scratch of the self-attention K pre-pass (wan_attention_bounded): wan_attention_scratch_words(S, S, Ll, heads)
  int64_t Lp;
  // mixed-precision plan only (x then holds fp32 rows): sinusoid / time MLP / e / e0 in fp32, the head's modulated rows, its token-major result
  float *mx_sin = nullptr, *mx_eh = nullptr, *mx_e = nullptr, *mx_e0 = nullptr, *mx_tmp = nullptr, *mx_tok = nullptr;
};

static int64_t carve_all(const wan_dit_config& g, int S, int64_t Ll, int world, void* ws, Bufs* b, int nvace, bool fp8, int F, bool mixed = false) {
  Carve c(ws);
  const int64_t d = g.dim, rows = (int64_t)S * Ll;
  const int64_t Lp = ((Ll + 63) / 64) * 64;
  Bufs t;
  t.Lp = Lp;
  t.x = c.take<bf16_t>(rows * d * (mixed ? 2 : 1));   // mixed-precision plan: the residual stream in fp32
  t.xm = c.take<bf16_t>(rows * d);
  t.q = c.take<bf16_t>(rows * d);
  t.k = c.take<bf16_t>(rows * d);
  t.vt = c.take<bf16_t>((int64_t)S * d * Lp);
  t.h = c.take<bf16_t>(rows * g.ffn_dim);
  // text context: up to two prompts per stream (normalized attention guidance: positive ; negative, any2video.py:608)
  t.ctx_h = c.take<bf16_t>((int64_t)2 * S * g.text_len * d);
  t.ctx_e = c.take<bf16_t>((int64_t)2 * S * g.text_len * d);
  t.ck = c.take<bf16_t>((int64_t)2 * S * g.text_len * d);
  t.cvt = c.take<bf16_t>((int64_t)2 * S * d * g.text_len);
  // time embedding / projection: one row per timestep -- 1 normally, up to F with per-frame timesteps (model.py:1812-1818;
  // ti2v image conditioning any2video.py:1496-1499, diffusion forcing); e0 is replicated per stream so that a kernel's
  // row / rows_per_batch lookup works on the stacked streams
  t.sinus = c.take<bf16_t>((int64_t)F * g.freq_dim);
  t.e_h = c.take<bf16_t>((int64_t)F * d);
  t.e = c.take<bf16_t>((int64_t)F * d);
  t.e_s = c.take<bf16_t>((int64_t)F * d);
  t.e0 = c.take<bf16_t>((int64_t)S * F * 6 * d);
  t.kmax = c.take<float>(wan_attention_scratch_words(S, S, Ll, g.num_heads));
  // scaled-fp8 Linears quantise their input per tensor = per stream (the reference runs the streams of a joint pass one
  // after the other through each block, model.py:1993-2036): the widest Linear input of one stream, S slots
  const int64_t tmax = std::max<int64_t>(std::max<int64_t>(Ll, 2 * (int64_t)g.text_len), CLIP_TOK);
  const int64_t kmax_in = std::max(std::max(g.dim, g.ffn_dim), std::max(g.text_dim, CLIP_DIM));
  t.q8_slot = fp8 ? ((tmax * kmax_in + 255) / 256) * 256 : 0;
  t.xq = fp8 ? c.take<uint8_t>(t.q8_slot * S) : nullptr;
  t.qws = fp8 ? c.take<float>(64 * S) : nullptr;
  t.ckimg = c.take<bf16_t>((int64_t)CLIP_TOK * d);      // i2v CLIP branch: K_img [257, d] and V_img^T [d, 320] (3.3 + 3.3 MB at 14B)
  t.cvtimg = c.take<bf16_t>((int64_t)d * CLIP_LDV);
  t.vc = nvace ? c.take<bf16_t>((int64_t)nvace * rows * d) : nullptr;      // VACE: per context, the hint token streams and the projected hint of one block
  t.vskip = nvace ? c.take<bf16_t>((int64_t)nvace * rows * d) : nullptr;
  if (world > 1) {
    // all-gather form: the gathered K / V^T of every rank.  Ulysses form (wan_sp_info.mode): the same two regions hold its six
    // exchange buffers -- k / q send and receive (4 x rows x d) in kfull, v^T send and receive (2 x S d Lp) in vtfull
    t.kfull = c.take<bf16_t>((int64_t)std::max(world, 4) * rows * d);
    t.vtfull = c.take<bf16_t>((int64_t)world * S * d * Lp);
    t.raw = c.take<float>(wan_attention_raw_words(S, Ll, g.num_heads));
  } else {
    t.kfull = nullptr;
    t.vtfull = nullptr;
    t.raw = nullptr;
  }
  if (mixed) {  // (behind everything else: the bf16 plan's layout does not move)
    t.mx_sin = c.take<float>((int64_t)F * g.freq_dim);
    t.mx_eh = c.take<float>((int64_t)F * d);
    t.mx_e = c.take<float>((int64_t)F * d);
    t.mx_e0 = c.take<float>((int64_t)S * F * 6 * d);
    t.mx_tmp = c.take<float>(Ll * d);
    t.mx_tok = c.take<float>(Ll * 4 * (int64_t)g.out_dim);
  }
  if (b) *b = t;
  return c.off;
}

static bool ctx_has_fp8(const wan_ctx* c) {
  for (const auto& kv : c->weights)
    if (kv.second.dtype == 2) return true;
  return false;
}

extern "C" int64_t wan_dit_workspace_bytes(const wan_ctx* ctx, int S, int F, int H, int W, int seq_shards) {
  if (!ctx || S < 1 || seq_shards < 1) return -1;
  const int64_t L = (int64_t)F * (H / 2) * (W / 2);
  if (L % seq_shards != 0) return -1;
  return carve_all(ctx->cfg, S, L / seq_shards, seq_shards, nullptr, nullptr, ctx->vace_layers.empty() ? 0 : ctx->vace_max_ctx, ctx_has_fp8(ctx), F,
                   ctx_is_mixed(ctx));
}

#define RC(expr)             \
  do {                       \
    if (int _rc = (expr)) return _rc; \
  } while (0)

// One nn.Linear of the forward.  bf16 weights: one GEMM over all M rows.  Scaled-fp8 weights (Lin::w8): the M rows are `nt`
// equal tensors (the streams of the joint pass), each quantised on its own (_quantize_activation is per tensor) into its
// fp8 slot and multiplied by its own launch; `reuse` = the slots already hold this input (q / k / v share one quantisation).
// ldc != 0 with WAN_EPI_TRANSPOSED: C is [N, ldc] (V^T); callers pass one stream at a time (nt = 1, slot = stream).
struct Q8 {
  uint8_t* xq = nullptr;
  float* ws = nullptr;
  int64_t slot_bytes = 0;
};
// amax_word (round 5): 0 = quantise with the abs-max pass (wan_fp8_quantize); 1 / 2 = the abs-max of every stream's rows already sits in
// that word of its slot -- left there by the kernel that produced A (the AMAX LayerNorm forms: word 1; ffn.0's GELU epilogue: word 2) -- and
// only the quantising half runs.  out_amax_word: for a GELU Linear, the word of each stream's slot that receives max |C| for the next one.
static int linear(const bf16_t* A, const Lin& l, bf16_t* C, int64_t M, int N, int K, int epi, void* st,
                  const bf16_t* R = nullptr, const bf16_t* mod = nullptr, const bf16_t* e = nullptr, int gate = -1,
                  int64_t rpb = 1, int64_t ldc = 0, const Q8* q8 = nullptr, int nt = 1, int slot0 = 0, bool reuse = false, int amax_word = 0,
                  int out_amax_word = 0) {
  if (l.w8 == nullptr) return wan_gemm_bf16(A, K, l.w, l.b, C, ldc ? ldc : N, M, N, K, epi, R, mod, e, 6, gate, rpb, st);
  WAN_REQUIRE(q8 && q8->xq && M % nt == 0, "fp8 Linear without quantisation scratch (internal)");
  const int64_t Mt = M / nt;
  WAN_REQUIRE(Mt * K <= q8->slot_bytes, "fp8 Linear input %lld x %d exceeds the quantisation slot (internal)", (long long)Mt, K);
  for (int t = 0; t < nt; ++t) {
    uint8_t* xq = q8->xq + (int64_t)(slot0 + t) * q8->slot_bytes;
    float* ws = q8->ws + (int64_t)(slot0 + t) * 64;
    if (!reuse) {
      if (amax_word > 0) RC(wan_fp8_quantize_pre(A + (int64_t)t * Mt * K, xq, ws, Mt * K, amax_word, st));
      else RC(wan_fp8_quantize(A + (int64_t)t * Mt * K, xq, ws, Mt * K, st));
    }
    const int64_t co = (epi == WAN_EPI_TRANSPOSED) ? 0 : (int64_t)t * Mt * N;
    if (out_amax_word > 0 && epi == WAN_EPI_GELU_TANH && ldc == 0)
      RC(wan_gemm_fp8_amax(xq, K, ws, l.w8, l.ws, l.ns, l.b, C + co, Mt, N, K, ws + out_amax_word, st));
    else
      RC(wan_gemm_fp8(xq, K, ws, l.w8, l.ws, l.ns, l.b, C + co, ldc ? ldc : N, Mt, N, K, epi, R ? R + co : nullptr, mod, e, 6, gate,
                      rpb < Mt ? rpb : Mt, st));
  }
  return 0;
}

extern "C" int wan_add_bf16(const wan_bf16* a, const wan_bf16* b, wan_bf16* out, int64_t n, void* stream);
extern "C" int wan_sub_bf16(const wan_bf16* a, const wan_bf16* b, wan_bf16* out, int64_t n, void* stream);
extern "C" int wan_axpy_bf16(const wan_bf16* x, const wan_bf16* y, float alpha, wan_bf16* out, int64_t n, void* stream);

// context_clip = img_emb(clip_fea) (model.py:1858-1859; MLPProj :868-889): LayerNorm(1280) -> Linear -> GELU(erf) -> Linear ->
// LayerNorm(dim), eps 1e-5 (torch.nn.LayerNorm).  The CLIP features do not change during a generation, so this runs once
// per call of wan_dit_set_clip instead of once per forward; the result stays in a buffer owned by the context.
extern "C" int wan_dit_set_clip(wan_ctx* c, const wan_bf16* clip_fea, void* stream) {
  WAN_REQUIRE(c && clip_fea, "wan_dit_set_clip: null argument");
  RC(resolve(c));
  WAN_REQUIRE(c->has_img, "wan_dit_set_clip: the loaded checkpoint has no img_emb / k_img / v_img weights (not a Wan2.1 i2v model)");
  const int d = c->cfg.dim;
  // flf2v (model.py:884-887): clip_fea holds the start AND the end image, [2, 257, 1280]; MLPProj views them as one sequence of
  // 514 tokens and adds its position embedding (one bf16 rounding) before the MLP.  The block's cross-attention then takes the
  // FIRST 257 projected tokens as its image context and the other 257 in front of the text tokens (it splits at 257, :472-473).
  const int64_t CT = (int64_t)CLIP_TOK * (c->has_flf ? 2 : 1);
  if (!c->clip_ctx) WAN_CHECK_HIP(hipMalloc((void**)&c->clip_ctx, (size_t)CT * d * 2));
  if (!c->clip_tmp) WAN_CHECK_HIP(hipMalloc((void**)&c->clip_tmp, (size_t)2 * CT * CLIP_DIM * 2 + (size_t)CT * d * 2));
  bf16_t* t1 = c->clip_tmp;
  bf16_t* t2 = t1 + CT * CLIP_DIM;
  bf16_t* t3 = t2 + CT * CLIP_DIM;
  WAN_REQUIRE(!c->ie1.w8 && !c->ie3.w8, "wan_dit_set_clip: img_emb Linears must be bf16 (dequantise them at load)");
  const bf16_t* src = clip_fea;
  if (c->has_flf) {
    RC(wan_add_bf16(clip_fea, c->ie_pos, t2, CT * CLIP_DIM, stream));
    src = t2;
  }
  RC(wan_ln_affine(src, t1, c->ie_ln0w, c->ie_ln0b, CT, CLIP_DIM, 1e-5f, stream));
  RC(linear(t1, c->ie1, t2, CT, CLIP_DIM, CLIP_DIM, WAN_EPI_NONE, stream));
  RC(wan_act_bf16(t2, t2, CT * CLIP_DIM, 2, stream));
  RC(linear(t2, c->ie3, t3, CT, d, CLIP_DIM, WAN_EPI_NONE, stream));
  RC(wan_ln_affine(t3, c->clip_ctx, c->ie_ln4w, c->ie_ln4b, CT, d, 1e-5f, stream));
  c->clip_set = true;
  return 0;
}

// The text cache slot of (key, stream layout): -> index of a valid slot holding it, or -1
static int text_cache_find(const wan_ctx* c, uint64_t key, int S, int TLx, int LDVx) {
  if (key == 0) return -1;
  for (int i = 0; i < 2; ++i) {
    const auto& t = c->tcache[i];
    if (t.valid && t.key == key && t.S == S && t.TLx == TLx && t.LDVx == LDVx) return i;
  }
  return -1;
}
// a slot to fill for `key`: an invalid one, else the least recently used; buffers for `nblk` blocks of `elems` elements each (kept when
// they fit).  -1: no memory -- the forward then runs without the cache.  Replayed launch lists hold these pointers: a reallocation drops them.
static void drop_graphs(wan_ctx* ctx);
static int text_cache_claim(wan_ctx* c, uint64_t key, int S, int TLx, int LDVx, size_t nblk, size_t elems) {
  int pick = 0;
  if (!c->tcache[0].valid) pick = 0;
  else if (!c->tcache[1].valid) pick = 1;
  else pick = c->tcache[0].last_use <= c->tcache[1].last_use ? 0 : 1;
  auto& t = c->tcache[pick];
  t.valid = false;
  if (t.ck.size() != nblk || t.elems < elems) {
    text_cache_free(t);
    drop_graphs(c);
    t.ck.assign(nblk, nullptr);
    t.cvt.assign(nblk, nullptr);
    for (size_t i = 0; i < nblk; ++i) {
      if (hipMalloc((void**)&t.ck[i], elems * 2) != hipSuccess || hipMalloc((void**)&t.cvt[i], elems * 2) != hipSuccess) {
        (void)hipGetLastError();
        text_cache_free(t);
        return -1;
      }
    }
    t.elems = elems;
  }
  t.key = key; t.S = S; t.TLx = TLx; t.LDVx = LDVx;
  return pick;
}

// should_calc / residual: the step-skipping caches of the reference (TeaCache / MagCache, model.py:1914-2064).  Stream s
// with residual[s] != NULL either runs the block chain and leaves residual[s] = x_after_blocks - x_after_patch_embed
// (should_calc[s] != 0), or skips the chain and adds the stored residual to its freshly embedded tokens.  The decision is
// host logic (wan2gp_amd/skipcache.py); both arrays NULL = the plain forward.
// ---- WanModel.forward as stages (round 6: the review's item 8) -----------------------------------------------------------------------
// Until round 5 this was ONE 590-line function whose nested lambdas carried every plan (bf16 / mixed / scaled fp8) x every self-attention
// form (plain, all-gather, Ulysses one-exchange / chunked) x VACE x NAG x skip-layer guidance x the step-skipping caches.  The same
// statements, in the same order, now sit in one method per stage of a `Forward` object: prepare (validation, geometry, the text cache,
// the workspace), embed (patch / time / text embeddings), skip_prologue / skip_epilogue (TeaCache / MagCache residual bookkeeping),
// vace_embed, run_blocks -> run_layer -> { norm1, self_attention_{plain, allgather, ulysses_chunked, ulysses_one}, o_projection,
// cross_attention, ffn }, block_chain (skip-layer guidance cuts), head.  A Linear goes through linear() / linear_res32(): the plan is the
// weight's (bf16 or fp8 bytes) and the context's (fp32 stream or not), not the stage's.  What held the split to 'nothing changes':
// tools/compare_forward_launch_lists.py replays 122 scenarios (plans x stream counts x text cache x per-frame timesteps x step skipping x
// skip-layer guidance x NAG x all-gather / Ulysses worlds of 2 and 4 x replayed launch lists) through the function before and after on
// the recording mock: 10,502 launches and hook calls, argument for argument, 0 differ; tests/test_dit_host_logic_cpu.py asserts the order.
namespace {
struct B2 {   // the buffers a run of streams works on: the workspace's scratch from its base, token streams and text context offset to the run
  bf16_t *x, *xm, *q, *k, *vt, *h, *ck, *cvt, *ctx_e, *e0, *kfull, *vtfull, *ckimg, *cvtimg;
  float* kmax;
  float* raw;
};
struct Forward {
  // ---- the call -----------------------------------------------------------------------------------------------------------------
  wan_ctx* c; int S; const float* const* x; float t; const float* t_frames; const wan_bf16* const* context; const float* y; const float* cos;
  const float* sin; float* const* outs; int F, H, W; void* workspace; int64_t workspace_bytes; const wan_sp_info* sp; wan_poll_fn poll;
  void* poll_user; const int* should_calc; wan_bf16* const* residual; int n_vace; const float* const* vace_contexts; const float* vace_scales;
  const float* nag; const int* context_batches; const int* perturb_layers; int n_perturb; int x_id; void* stream; const float* t_dev;
  uint64_t context_key;
  // ---- geometry, plan, workspace (prepare) -----------------------------------------------------------------------------------------
  const wan_dit_config& g;
  int d = 0, ffn = 0, nh = 0, TL = 0, Hg = 0, Wg = 0, world = 1, XT = 0, TLx = 0, LDVx = 0;
  int64_t L = 0, Ll = 0, tok0 = 0, Lp = 0, sn = 0;
  bool ulysses = false, any_nag = false, tc_hit = false, mx = false;
  int cb[8], crow[9];
  wan_ctx::TextCache* tc = nullptr;
  Bufs b;
  float* x32 = nullptr;
  hipStream_t st = nullptr;
  Q8 q8v;
  const Q8* q8 = nullptr;
  // ---- embeddings -------------------------------------------------------------------------------------------------------------------
  int64_t tpf = 0;  // tokens per frame
  int nt = 1, frame0 = 0;
  // ---- VACE ---------------------------------------------------------------------------------------------------------------------------
  int S_all = 0, n_on = 0, on_k[8];
  bool vace = false;
  // ---- the run of streams the block chain is on (run_blocks) and the block being enqueued (run_layer) -------------------------------------
  int Sn = 0, s0 = 0;
  int64_t rows = 0, rpb = 0;
  B2 b2;
  bf16_t* x_main = nullptr;
  const float* e0f = nullptr;
  bf16_t *vc[8], *vskip[8];
  float* xf = nullptr;
  bool fold = false;
  int a1 = 0;

  explicit Forward(wan_ctx* c_) : c(c_), g(c_->cfg) {}
  bool calc(int s) const { return should_calc == nullptr || should_calc[s] != 0; }

  int prepare() {
    WAN_REQUIRE(c && x && context && cos && sin && outs && workspace, "wan_dit_forward: null argument");
    WAN_REQUIRE(S >= 1 && S <= 8, "wan_dit_forward: S=%d streams unsupported", S);
    WAN_REQUIRE(H % 2 == 0 && W % 2 == 0 && F >= 1, "wan_dit_forward: latent H,W must be even");
    RC(resolve(c));
    d = g.dim; ffn = g.ffn_dim; nh = g.num_heads; TL = g.text_len;
    Hg = H / 2; Wg = W / 2;
    L = (int64_t)F * Hg * Wg;
    world = sp ? sp->world : 1;
    ulysses = sp && world > 1 && sp->mode == WAN_SP_ULYSSES;
    WAN_REQUIRE(world >= 1 && L % world == 0, "wan_dit_forward: L=%lld not divisible by %d sequence shards",
                (long long)L, world);
    Ll = L / world;
    tok0 = sp ? sp->tok0 : 0;
    WAN_REQUIRE(!ulysses || (sp->a2a_begin && sp->a2a_wait && g.num_heads % world == 0),
                "wan_dit_forward: the Ulysses exchange needs its all-to-all hooks and a head count the world divides (%d heads, world %d)",
                g.num_heads, world);
    WAN_REQUIRE(!sp || (sp->tok_local == Ll && sp->tok0 == (int64_t)sp->rank * Ll &&
                        (world == 1 || ulysses || (sp->gather_begin && sp->gather_wait))),
                "wan_dit_forward: inconsistent sequence-parallel info");
    WAN_REQUIRE((g.in_dim > g.out_dim) == (y != nullptr), "wan_dit_forward: y must be given iff in_dim > out_dim (model.py:1597)");
    WAN_REQUIRE(!c->has_img || c->clip_set, "wan_dit_forward: this is a Wan2.1 i2v model -- call wan_dit_set_clip first (model.py:1547)");
    // Normalized attention guidance (any2video.py:607-608): a stream whose context holds two prompts (positive ; negative) runs its
    // text cross-attention against both and combines the results (text_cross_attention, model.py:260-292).  crow[s] = first row of
    // stream s's context in the stacked context buffers, cb[s] = its prompts.
    any_nag = false;
    crow[0] = 0;
    for (int s = 0; s < S; ++s) {
      cb[s] = context_batches ? context_batches[s] : 1;
      WAN_REQUIRE(cb[s] == 1 || (cb[s] == 2 && nag && nag[0] > 1.f),
                  "wan_dit_forward: context_batches[%d] = %d (1, or 2 together with nag_scale > 1: model.py:260)", s, cb[s]);
      any_nag = any_nag || cb[s] == 2;
      crow[s + 1] = crow[s] + cb[s] * TL;
    }
    WAN_REQUIRE(!any_nag || ffn >= d, "wan_dit_forward: normalized attention guidance parks a result in the FFN buffer (ffn_dim >= dim)");
    // (flf2v: the text branch's context is [the second image's 257 CLIP tokens ; the text tokens], see below)
    XT = c->has_flf ? CLIP_TOK : 0;
    TLx = TL + XT; LDVx = XT ? ((TLx + 63) / 64) * 64 : TL;
    // ---- the text cache (wan_ctx::TextCache): served when every stream of the call runs the whole block chain on one prompt each ----
    tc = nullptr;
    tc_hit = false;
    {
      bool all_calc = true;
      for (int s = 0; should_calc != nullptr && s < S; ++s) all_calc = all_calc && should_calc[s] != 0;
      if (context_key != 0 && !any_nag && n_perturb == 0 && all_calc) {
        int slot = text_cache_find(c, context_key, S, TLx, LDVx);
        tc_hit = slot >= 0;
        if (slot < 0) slot = text_cache_claim(c, context_key, S, TLx, LDVx, (size_t)g.num_layers + c->vlayers.size(), (size_t)2 * S * TL * d);
        if (slot >= 0) {
          tc = &c->tcache[slot];
          tc->last_use = ++c->tcache_clock;
        }
      }
    }
    mx = c->mixed;
    const int64_t need = carve_all(g, S, Ll, world, workspace, &b, c->vace_layers.empty() ? 0 : c->vace_max_ctx, c->any_fp8, F, mx);
    // the mixed-precision plan serves the block chain of the t2v / i2v2_2 / ti2v / i2v (CLIP) models (also under sequence parallelism, NAG,
    // skip-layer guidance, fp8 Linears); what keeps bf16 state of its own is refused rather than silently run in the other plan
    bool any_residual = false;
    for (int s = 0; residual != nullptr && s < S; ++s) any_residual = any_residual || residual[s] != nullptr;
    (void)any_residual;   // (round 5: the step-skipping caches run in the mixed plan too -- their residual buffers then hold fp32 rows)
    // (round 6: the Wan2.1 i2v / flf2v CLIP branch runs in the plan -- no lock of model.py:1330-1371 names img_emb or k_img / v_img, so the
    // branch is the bf16 plan's own between the fp32 norm3 in front of the q Linear and the fp32-stream o projection behind it)
    WAN_REQUIRE(!mx || (n_vace == 0 && c->vace_layers.empty()),
                "wan_dit_forward: the mixed-precision plan (fp32 time_projection / norm3 weights) does not serve VACE context blocks");
    x32 = reinterpret_cast<float*>(b.x);
    WAN_REQUIRE(workspace_bytes >= need, "wan_dit_forward: workspace %lld < required %lld bytes",
                (long long)workspace_bytes, (long long)need);
    WAN_REQUIRE((((uintptr_t)workspace) & 255) == 0, "wan_dit_forward: workspace must be 256-byte aligned");
    st = as_stream(stream);
    Lp = b.Lp;
    q8v.xq = b.xq; q8v.ws = b.qws; q8v.slot_bytes = b.q8_slot;
    q8 = c->any_fp8 ? &q8v : nullptr;
    sn = Ll * (int64_t)d;
    return 0;
  }

  // the time MLP: GEMV for one row of bf16 weights, the tile GEMM otherwise (several rows, or fp8 weights: one tensor)
  int tlin(const bf16_t* A, const Lin& l, bf16_t* C, int N, int K) {
      if (l.w8 || nt > 1) return linear(A, l, C, nt, N, K, WAN_EPI_NONE, stream, nullptr, nullptr, nullptr, -1, 1, 0, q8);
      return wan_gemv_bf16(A, l.w, l.b, C, 1, N, K, stream);
  }

  int embed() {
    // V^T padding columns must be finite for the PV MFMA (P = 0 there)
    WAN_CHECK_HIP(hipMemsetAsync(b.vt, 0, (size_t)S * d * Lp * 2, st));
    if (c->has_img) WAN_CHECK_HIP(hipMemsetAsync(b.cvtimg, 0, (size_t)d * CLIP_LDV * 2, st));

    // ---- embeddings (model.py:1631,1731 ; :1815-1818 ; :1856) -----------------------------------------
    for (int s = 0; s < S; ++s) {
      if (mx) RC(wan_mx_patch_embed(x[s], y, c->pe_w, c->pe_b, x32 + (int64_t)s * Ll * d, g.out_dim, g.in_dim - g.out_dim, F, H, W, d, tok0, Ll, stream));
      else RC(wan_patch_embed_range(x[s], y, c->pe_w, c->pe_b, b.x + (int64_t)s * Ll * d, 1, g.out_dim, g.in_dim - g.out_dim, F, H, W, d,
                                    tok0, Ll, stream));
    }
    // Timesteps: one scalar, or one per latent frame (t_frames[F], HOST pointer): tokens of frame f are modulated by
    // e0[f] (model.py:631-638, :856-862).  Under sequence parallelism a rank needs the rows of its own frames only, which
    // requires shards made of whole frames.
    tpf = (int64_t)Hg * Wg;  // tokens per frame
    nt = 1; frame0 = 0;
    if (t_frames != nullptr) {
      WAN_REQUIRE(Ll % tpf == 0 && tok0 % tpf == 0, "wan_dit_forward: per-frame timesteps need sequence shards made of whole frames "
                  "(%lld tokens per shard, %lld per frame)", (long long)Ll, (long long)tpf);
      nt = (int)(Ll / tpf);
      frame0 = (int)(tok0 / tpf);
    }
    if (mx) {
      // the time MLP and the projection under the fp32 lock (model.py:1815-1818 with an fp32 modulation dtype): e, e0 stay fp32
      for (int f = 0; f < nt; ++f)
        RC(wan_mx_sinusoid(t_frames ? t_frames[frame0 + f] : t, b.mx_sin + (int64_t)f * g.freq_dim, g.freq_dim, stream));
      RC(wan_mx_linear_f32(b.mx_sin, c->mx_tm0w, c->mx_tm0b, b.mx_eh, nt, d, g.freq_dim, 0, stream));
      RC(wan_mx_linear_f32(b.mx_eh, c->mx_tm2w, c->mx_tm2b, b.mx_e, nt, d, d, 1, stream));          // Linear(SiLU(.))
      RC(wan_mx_linear_f32(b.mx_e, c->mx_tp1w, c->mx_tp1b, b.mx_e0, nt, 6 * d, d, 1, stream));       // time_projection = SiLU -> Linear
      for (int s = 1; s < S && nt > 1; ++s)
        WAN_CHECK_HIP(hipMemcpyAsync(b.mx_e0 + (int64_t)s * nt * 6 * d, b.mx_e0, (size_t)nt * 6 * d * 4, hipMemcpyDeviceToDevice, st));
    }
    for (int f = 0; f < nt && !mx; ++f) {
      if (t_dev != nullptr && t_frames == nullptr) RC(wan_sinusoid(t_dev, b.sinus, 1, g.freq_dim, stream));
      else RC(wan_sinusoid_val(t_frames ? t_frames[frame0 + f] : t, b.sinus + (int64_t)f * g.freq_dim, g.freq_dim, stream));
    }
    // the time MLP: GEMV for one row of bf16 weights, the tile GEMM otherwise (several rows, or fp8 weights: one tensor)
    if (!mx) {
      RC(tlin(b.sinus, c->tm0, b.e_h, d, g.freq_dim));
      RC(wan_act_bf16(b.e_h, b.e_h, (int64_t)nt * d, 1, stream));
      RC(tlin(b.e_h, c->tm2, b.e, d, d));
      RC(wan_act_bf16(b.e, b.e_s, (int64_t)nt * d, 1, stream));
      RC(tlin(b.e_s, c->tp1, b.e0, 6 * d, d));
      for (int s = 1; s < S && nt > 1; ++s)
        WAN_CHECK_HIP(hipMemcpyAsync(b.e0 + (int64_t)s * nt * 6 * d, b.e0, (size_t)nt * 6 * d * 2, hipMemcpyDeviceToDevice, st));
    }
    for (int s = 0; s < S && !tc_hit; ++s) {  // one tensor per stream (fp8: one quantisation per tensor), [cb[s] * TL, text_dim]
      RC(linear(context[s], c->te0, b.ctx_h + (int64_t)crow[s] * d, (int64_t)cb[s] * TL, d, g.text_dim, WAN_EPI_GELU_TANH, stream, nullptr,
                nullptr, nullptr, -1, 1, 0, q8, 1, s));
    }
    if (tc_hit) {
      // (the text embedding feeds the blocks' cross-attention K / V Linears only: with those cached it is not needed)
    } else if (!any_nag) {
      RC(linear(b.ctx_h, c->te2, b.ctx_e, (int64_t)S * TL, d, d, WAN_EPI_NONE, stream, nullptr, nullptr, nullptr, -1, 1, 0, q8, S));
    } else {
      for (int s = 0; s < S; ++s)
        RC(linear(b.ctx_h + (int64_t)crow[s] * d, c->te2, b.ctx_e + (int64_t)crow[s] * d, (int64_t)cb[s] * TL, d, d, WAN_EPI_NONE, stream,
                  nullptr, nullptr, nullptr, -1, 1, 0, q8, 1, s));
    }

    // flf2v: the text branch of the cross-attention sees [the second image's 257 CLIP tokens ; the text tokens] (model.py:472-473 splits
    // the context at 257, img_emb produced 514).  Assembled once per forward in ctx_h (dead after the text embedding above): per
    // stream TLx = 257 + text_len rows; the V^T images get a row pitch of TLx rounded up to 64, their pad columns zeroed here (the
    // projection GEMMs of the blocks never write them).
    if (XT) {
      WAN_REQUIRE(!any_nag, "wan_dit_forward: normalized attention guidance is not served together with the flf2v CLIP context");
      for (int s = 0; s < S && !tc_hit; ++s) {
        bf16_t* dst = b.ctx_h + (int64_t)s * TLx * d;
        WAN_CHECK_HIP(hipMemcpyAsync(dst, c->clip_ctx + (int64_t)CLIP_TOK * d, (size_t)CLIP_TOK * d * 2, hipMemcpyDeviceToDevice, st));
        WAN_CHECK_HIP(hipMemcpyAsync(dst + (int64_t)XT * d, b.ctx_e + (int64_t)s * TL * d, (size_t)TL * d * 2, hipMemcpyDeviceToDevice, st));
      }
      WAN_CHECK_HIP(hipMemsetAsync(b.cvt, 0, (size_t)S * d * LDVx * 2, st));
    }

    // every stream of the joint pass shares t, hence e0: one "batch" for the modulation lookups (rpb = rows in run_blocks)

    // Self-attention: softmax scale * log2(e) is folded into q inside the fused RMSNorm+RoPE kernel (in front of q's single
    // bf16 rounding); the attention kernel's score tiles then come out of the matrix pipe ready for exp2.

    // ---- step-skipping: park x_before in the residual buffer, or add the stored residual and skip (model.py:1967-1990) ----
    return 0;
  }

  int skip_prologue() {
    // ---- step-skipping: park x_before in the residual buffer, or add the stored residual and skip (model.py:1967-1990) ----
    for (int s = 0; s < S; ++s) {
      if (residual == nullptr || residual[s] == nullptr) {
        WAN_REQUIRE(calc(s), "wan_dit_forward: stream %d is skipped but has no residual buffer", s);
        continue;
      }
      if (mx) {
        // the mixed-precision plan: x is the fp32 stream, so is the reference's previous_residual (torch.sub of two fp32 tensors,
        // model.py:2044-2062); the caller's buffer holds sn floats.  1 * x + 1 * r is one fma: the fp32 sum, rounded once.
        float* xs = x32 + s * sn;
        float* rs = reinterpret_cast<float*>(residual[s]);
        if (calc(s)) WAN_CHECK_HIP(hipMemcpyAsync(rs, xs, (size_t)sn * 4, hipMemcpyDeviceToDevice, st));
        else {
          const float* in2[2] = {xs, rs};
          const float one2[2] = {1.f, 1.f};
          RC(wan_lincomb(xs, 2, in2, one2, sn, stream));
        }
        continue;
      }
      if (calc(s)) WAN_CHECK_HIP(hipMemcpyAsync(residual[s], b.x + s * sn, (size_t)sn * 2, hipMemcpyDeviceToDevice, st));
      else RC(wan_add_bf16(b.x + s * sn, residual[s], b.x + s * sn, sn, stream));
    }
    return 0;
  }

  int vace_embed() {
    // ---- VACE: c_k = vace_patch_embedding(vace_context[k]) for every context and stream (model.py:1905-1912) -------------------------
    // Context k with scale 0 is switched off (model.py:622-626): no hint stream, no context block, no add.
    S_all = S;
    n_on = 0;
    if (n_vace > 0) {
      WAN_REQUIRE(!c->vace_layers.empty(), "wan_dit_forward: vace_context given but the model has no VACE blocks (wan_dit_set_vace_layers)");
      WAN_REQUIRE(vace_contexts && vace_scales && n_vace <= c->vace_max_ctx && n_vace <= 8,
                  "wan_dit_forward: %d VACE contexts, the workspace holds %d (wan_dit_set_vace_contexts)", n_vace, c->vace_max_ctx);
      for (int k = 0; k < n_vace; ++k) {
        if (vace_scales[k] == 0.f) continue;
        WAN_REQUIRE(vace_contexts[k] != nullptr, "wan_dit_forward: vace_contexts[%d] is null", k);
        bf16_t* vck = b.vc + (int64_t)k * S * sn;
        RC(wan_patch_embed_range(vace_contexts[k], nullptr, c->vpe_w, c->vpe_b, vck, 1, c->vace_in_dim, 0, F, H, W, d, tok0, Ll, stream));
        for (int s = 1; s < S; ++s)
          WAN_CHECK_HIP(hipMemcpyAsync(vck + s * sn, vck, (size_t)sn * 2, hipMemcpyDeviceToDevice, st));
        on_k[n_on++] = k;
      }
    }
    vace = n_on > 0;
    return 0;
  }

    // one WanAttentionBlock (model.py:575-724) on the token streams at b.x, with the weights Lw
    // mixed-precision plan: a Linear whose bf16 result is added to the fp32 stream -- x += y * gate.  bf16 weights: ONE launch, the update
    // in the GEMM's epilogue (wan_gemm_bf16_res32; round 5); scaled-fp8 weights: the Linear into xm, then the separate pass
  int linear_res32(const bf16_t* A, const Lin& l, float* xf_, bf16_t* tmp, int64_t M, int N, int K, const bf16_t* mod, const float* e0g, int gate,
                   int64_t rpb_, int nt_) {
      if (l.w8 == nullptr) return wan_gemm_bf16_res32(A, K, l.w, l.b, xf_, tmp, M, N, K, mod, e0g, 6, gate, rpb_, stream);
      RC(linear(A, l, tmp, M, N, K, WAN_EPI_NONE, stream, nullptr, nullptr, nullptr, -1, 1, 0, q8, nt_));
      return wan_mx_gated_residual(xf_, tmp, gate >= 0 ? mod : nullptr, gate >= 0 ? e0g : nullptr, 6, gate, M, rpb_, N, stream);
  }

  int zero_slots() {   // (the streams' quantisation slots, in front of the producer that leaves an abs-max there)
    WAN_CHECK_HIP(hipMemsetAsync(q8->ws, 0, (size_t)Sn * 64 * sizeof(float), st));
    return 0;
  }

  int norm1(const Layer& Lw) {
    auto& b = b2;
    // -- self attention (model.py:632-660) --
    // mixed-precision plan (mx): b.x holds fp32 rows; modulate / norm3 / the gated residuals are the fp32 kernels of mixed_ops.hip, each
    // Linear that ended in a fused residual epilogue writes its bf16 result to xm (dead at those three points) and a separate pass adds it
    xf = reinterpret_cast<float*>(b.x);
    // scaled-fp8 checkpoints (round 5): the LayerNorms in front of fp8 Linears (norm1 -> q / k / v, norm3 -> cross q, norm2 -> ffn.0) and
    // ffn.0's GELU epilogue (-> ffn.2) leave the abs-max of what they write in the streams' quantisation slots, so those four of a block's six
    // activation quantisations run without their abs-max pass (slot word 1: a LayerNorm's output, word 2: the GELU output; zeroed here, in
    // front of the producer -- every earlier user of the slots has been enqueued).  `a1` is handed to every consumer: the one that quantises
    // (reuse == false) takes the short path, the others share its slots as before.
    static const bool no_fold = [] { const char* e = getenv("WAN_FP8_NO_FOLD"); return e && e[0] == '1'; }();   // A/B runs: the two-pass quantisation everywhere
    fold = q8 != nullptr && !mx && !no_fold;
    a1 = fold ? 1 : 0;
    if (mx) RC(wan_mx_ln_modulate(xf, b.xm, Lw.mod, e0f, 6, 0, 1, rows, rpb, d, g.eps, stream));
    else if (fold) {
      RC(zero_slots());
      RC(wan_ln_modulate_amax(b.x, b.xm, Lw.mod, b.e0, 6, 0, 1, rows, rpb, d, g.eps, q8->ws, Ll, stream));
    } else RC(wan_ln_modulate(b.x, b.xm, Lw.mod, b.e0, 6, 0, 1, rows, rpb, d, g.eps, stream));
    return 0;
  }

  int self_attention_ulysses(const Layer& Lw) {
    const int S = Sn;
    auto& b = b2;
    // Ulysses (round 4): re-shard q, k, v from "my tokens, all heads" to "all tokens, my heads" by all-to-all, attend the whole
    // sequence for nh / world heads in ONE launch, bring o back the same way.  Order K, V, Q as below: the k exchange runs under
    // the V projection, the v exchange under the Q projection + norm; q and o are exposed.  Layouts (W = 128 nh / world):
    //   send  [world][S][Ll][W]      the projection's [S Ll][d] rows re-packed head-group-major (wan_permute16: one pass)
    //   recv  [world][S][Ll][W]      = world x S query "batches" of Ll rows for the attention kernel (q batch b attends K / V^T
    //                                batch b mod S) and world K segments (seg stride S Ll W) -- the all-gather form's layout
    //   v^T   [world][S][W][Lp]      the transposed epilogue's [S][d][Lp] rows ARE head-group-major per stream: blocks swapped
    //                                [S][world] -> [world][S] (nothing to do for S = 1); received = world V^T segments
    //   o     the attention writes [world][S][Ll][W] = the send layout of the way back; received and un-permuted to [S Ll][d]
    const int Hn = nh / world;
    const int64_t Wd = (int64_t)Hn * 128, blkq = rows * d, blkv = (int64_t)S * d * Lp;
    bf16_t *ks = b.kfull, *kr = b.kfull + blkq, *qs = b.kfull + 2 * blkq, *qr = b.kfull + 3 * blkq;
    bf16_t *vs = b.vtfull, *vr = b.vtfull + blkv;
    // Round 5: the rank's Hn heads in C chunks (wan_sp_info.a2a_chunks; 1 = the round-4 form below).  EVERY tensor travels per chunk,
    // chunk-major --
    //   k, q recv / o send  [chunk][world][S][Ll][Wc]        v^T recv  [chunk][world][S][Wc][Lp]        (chunk c: heads [h0_c, h1_c), Wc = 128 (h1_c - h0_c))
    // -- so a chunk's launch sees exactly the round-4 layout with H = h1_c - h0_c heads (segment strides rows Wc / S Wc Lp): the same
    // kernel on the same rows of the same heads, results bit-identical to C = 1 (tests/test_gpu_sp.py).  What the schedule buys: chunk
    // 0's k, v^T and q leave FIRST (k_0 under the V projection, v_0 under the Q projection, then q_0), so the first launch waits for
    // one chunk of q, not for three whole tensors queued on the same links; the other chunks' k / v^T / q flow under chunk 0's launch,
    // chunk c's o returns under chunk c + 1's launch: exposed are q_0 and the last o chunk.  Slots: k_c = c, v_c = C + c, q_c = 2 C + c,
    // o_c = 3 C + c (C = 1: the round-4 numbering 0..3).
    int C = sp->a2a_chunks < 1 ? 1 : sp->a2a_chunks;
    if (C > Hn) C = Hn;
    if (C > WAN_SP_MAX_CHUNKS) C = WAN_SP_MAX_CHUNKS;
    if (C > 1) return ulysses_chunked(Lw, C, Hn, Wd, ks, kr, qs, qr, vs, vr);
    return ulysses_one(Lw, Hn, Wd, ks, kr, qs, qr, vs, vr);
  }

  int a2a(int which, const bf16_t* send, bf16_t* recv, int64_t bytes) {
    if (sp->a2a_begin(sp->user, which, send, recv, bytes, stream)) {
      wan_set_error("wan_dit_forward: all-to-all %d (C head chunks: k 0.., v^T C.., q 2C.., o 3C..) failed", which);
      return 3;
    }
    return 0;
  }
  int a2a_wait(int which) {
    if (sp->a2a_wait(sp->user, which, stream)) {
      wan_set_error("wan_dit_forward: all-to-all %d failed", which);
      return 3;
    }
    return 0;
  }

  int ulysses_chunked(const Layer& Lw, int C, int Hn, int64_t Wd, bf16_t* ks, bf16_t* kr, bf16_t* qs, bf16_t* qr, bf16_t* vs, bf16_t* vr) {
    const int S = Sn;
    auto& b = b2;
    int h0[WAN_SP_MAX_CHUNKS + 1];
    for (int cch = 0; cch <= C; ++cch) h0[cch] = (int)((int64_t)cch * Hn / C);
    auto wc = [&](int cch) { return (int64_t)(h0[cch + 1] - h0[cch]) * 128; };
    auto o0 = [&](int cch) { return (int64_t)h0[cch] * 128; };
    auto send_k = [&](int cch) { return a2a(cch, ks + o0(cch) * rows * world, kr + o0(cch) * rows * world, rows * wc(cch) * 2); };
    auto send_v = [&](int cch) { return a2a(C + cch, vs + o0(cch) * Lp * S * world, vr + o0(cch) * Lp * S * world, (int64_t)S * wc(cch) * Lp * 2); };
    auto send_q = [&](int cch) { return a2a(2 * C + cch, qs + o0(cch) * rows * world, qr + o0(cch) * rows * world, rows * wc(cch) * 2); };
    RC(linear(b.xm, Lw.self.k, b.k, rows, d, d, WAN_EPI_NONE, stream, nullptr, nullptr, nullptr, -1, 1, 0, q8, S, 0, false, a1));
    // RMSNorm + RoPE written straight into the send layout [rows][world][Hn 128] -> [chunk][world][rows][Wc] (round 6: the norm kernel's
    // stores carry the re-pack; until then C wan_permute16_ex passes over the tensor followed it)
    RC(wan_rmsnorm_rope_pack(b.k, ks, Lw.self.nk, cos, sin, rows, Ll, tok0, d, g.eps, 1.0f, world, Hn, C, stream));
    RC(send_k(0));
    for (int s = 0; s < S; ++s)
      RC(linear(b.xm + (int64_t)s * Ll * d, Lw.self.v, b.vt + (int64_t)s * d * Lp, Ll, d, d, WAN_EPI_TRANSPOSED, stream, nullptr,
                nullptr, nullptr, -1, 1, Lp, q8, 1, s, Lw.self.k.w8 != nullptr, a1));
    for (int cch = 0; cch < C; ++cch)   // [S][world][Hn 128][Lp] -> [chunk][world][S][Wc][Lp]
      RC(wan_permute16_ex(b.vt + o0(cch) * Lp, vs + o0(cch) * Lp * S * world, S, world, wc(cch) * Lp * 2, (int64_t)d * Lp * 2, Wd * Lp * 2,
                          wc(cch) * Lp * 2, (int64_t)S * wc(cch) * Lp * 2, stream));
    RC(send_v(0));
    RC(linear(b.xm, Lw.self.q, b.q, rows, d, d, WAN_EPI_NONE, stream, nullptr, nullptr, nullptr, -1, 1, 0, q8, S, 0,
              Lw.self.k.w8 != nullptr || Lw.self.v.w8 != nullptr, a1));
    RC(wan_rmsnorm_rope_pack(b.q, qs, Lw.self.nq, cos, sin, rows, Ll, tok0, d, g.eps, wan_attention_qscale(), world, Hn, C, stream));
    RC(send_q(0));
    for (int cch = 1; cch < C; ++cch) {   // the later chunks, in the order their launches need them
      RC(send_k(cch));
      RC(send_v(cch));
      RC(send_q(cch));
    }
    for (int cch = 0; cch < C; ++cch) {
      const int Hc = h0[cch + 1] - h0[cch];
      RC(a2a_wait(cch));
      RC(a2a_wait(C + cch));
      RC(a2a_wait(2 * C + cch));
      {
        ProfScope ps(PROF_SELF_ATTN, st);
        RC(wan_attention_bounded(qr + o0(cch) * rows * world, kr + o0(cch) * rows * world, vr + o0(cch) * Lp * S * world, ks + o0(cch) * rows * world,
                                 world * S, S, Ll, Ll, Lp, Hc, world, rows * wc(cch), (int64_t)S * wc(cch) * Lp, 1, b.kmax, stream));
      }
      if (g_prof_on && g_prof_declined != nullptr && Ll * (int64_t)world > 2048)
        RC(wan_attention_count_declined(b.kmax, world * S, S, Ll, Hc, g_prof_declined, stream));
      // o chunk c over the dead k send chunk (its exchange was waited for above), back over the dead q send chunk, while the next launch runs
      RC(a2a(3 * C + cch, ks + o0(cch) * rows * world, qs + o0(cch) * rows * world, rows * wc(cch) * 2));
    }
    for (int cch = 0; cch < C; ++cch) {   // [chunk][world][rows][Wc] -> [rows][world][Hn 128]
      RC(a2a_wait(3 * C + cch));
      RC(wan_permute16_ex(qs + o0(cch) * rows * world, b.q + o0(cch), world, rows, wc(cch) * 2, rows * wc(cch) * 2, wc(cch) * 2, Wd * 2, (int64_t)d * 2, stream));
    }
    return 0;
  }

  int ulysses_one(const Layer& Lw, int Hn, int64_t Wd, bf16_t* ks, bf16_t* kr, bf16_t* qs, bf16_t* qr, bf16_t* vs, bf16_t* vr) {
    const int S = Sn;
    auto& b = b2;
    RC(linear(b.xm, Lw.self.k, b.k, rows, d, d, WAN_EPI_NONE, stream, nullptr, nullptr, nullptr, -1, 1, 0, q8, S, 0, false, a1));
    RC(wan_rmsnorm_rope_pack(b.k, ks, Lw.self.nk, cos, sin, rows, Ll, tok0, d, g.eps, 1.0f, world, Hn, 1, stream));   // norm + RoPE + re-pack in one pass
    RC(a2a(0, ks, kr, rows * Wd * 2));
    for (int s = 0; s < S; ++s)
      RC(linear(b.xm + (int64_t)s * Ll * d, Lw.self.v, b.vt + (int64_t)s * d * Lp, Ll, d, d, WAN_EPI_TRANSPOSED, stream, nullptr,
                nullptr, nullptr, -1, 1, Lp, q8, 1, s, Lw.self.k.w8 != nullptr, a1));
    const bf16_t* vsend = b.vt;
    if (S > 1) {
      RC(wan_permute16(b.vt, vs, S, world, Wd * Lp * 2, stream));
      vsend = vs;
    }
    RC(a2a(1, vsend, vr, (int64_t)S * Wd * Lp * 2));
    RC(linear(b.xm, Lw.self.q, b.q, rows, d, d, WAN_EPI_NONE, stream, nullptr, nullptr, nullptr, -1, 1, 0, q8, S, 0,
              Lw.self.k.w8 != nullptr || Lw.self.v.w8 != nullptr, a1));
    RC(wan_rmsnorm_rope_pack(b.q, qs, Lw.self.nq, cos, sin, rows, Ll, tok0, d, g.eps, wan_attention_qscale(), world, Hn, 1, stream));
    RC(a2a(2, qs, qr, rows * Wd * 2));
    for (int w3 = 0; w3 < 3; ++w3) RC(a2a_wait(w3));
    {
      ProfScope ps(PROF_SELF_ATTN, st);
      RC(wan_attention_bounded(qr, kr, vr, ks, world * S, S, Ll, Ll, Lp, Hn, world, rows * Wd, (int64_t)S * Wd * Lp, 1, b.kmax, stream));
    }
    if (g_prof_on && g_prof_declined != nullptr && Ll * (int64_t)world > 2048)
      RC(wan_attention_count_declined(b.kmax, world * S, S, Ll, Hn, g_prof_declined, stream));
    RC(a2a(3, ks, qs, rows * Wd * 2));
    RC(a2a_wait(3));
    RC(wan_permute16(qs, b.q, world, rows, Wd * 2, stream));
    return 0;
  }

  int self_attention_allgather(const Layer& Lw) {
    const int S = Sn;
    auto& b = b2;
    // Sequence parallelism: K first (projection, RMSNorm + RoPE), its all-gather started at once; then V^T and its gather; the
    // Q projection / norm and the attention over the rank's OWN K / V^T segment run while both collectives are in flight;
    // only then the waits, and the other ranks' segments on top of the partial sums (wan_attention_sp_local / _remote).
    RC(linear(b.xm, Lw.self.k, b.k, rows, d, d, WAN_EPI_NONE, stream, nullptr, nullptr, nullptr, -1, 1, 0, q8, S, 0, false, a1));
    RC(wan_rmsnorm_rope_scaled(b.k, nullptr, Lw.self.nk, nullptr, cos, sin, rows, Ll, tok0, d, g.eps, 1.0f, stream));
    if (sp->gather_begin(sp->user, 0, b.k, b.kfull, rows * (int64_t)d * 2, stream)) {
      wan_set_error("wan_dit_forward: K all-gather failed");
      return 3;
    }
    for (int s = 0; s < S; ++s)
      RC(linear(b.xm + (int64_t)s * Ll * d, Lw.self.v, b.vt + (int64_t)s * d * Lp, Ll, d, d, WAN_EPI_TRANSPOSED, stream, nullptr,
                nullptr, nullptr, -1, 1, Lp, q8, 1, s, Lw.self.k.w8 != nullptr, a1));
    if (sp->gather_begin(sp->user, 1, b.vt, b.vtfull, (int64_t)S * d * Lp * 2, stream)) {
      wan_set_error("wan_dit_forward: V^T all-gather failed");
      return 3;
    }
    RC(linear(b.xm, Lw.self.q, b.q, rows, d, d, WAN_EPI_NONE, stream, nullptr, nullptr, nullptr, -1, 1, 0, q8, S, 0,
              Lw.self.k.w8 != nullptr || Lw.self.v.w8 != nullptr, a1));
    RC(wan_rmsnorm_rope_scaled(b.q, nullptr, Lw.self.nq, nullptr, cos, sin, rows, Ll, tok0, d, g.eps, wan_attention_qscale(), stream));
    ProfScope ps(PROF_SELF_ATTN, st);
    RC(wan_attention_sp_local(b.q, b.k, b.vt, S, Ll, Ll, Lp, nh, b.kmax, b.raw, stream));
    if (sp->gather_wait(sp->user, 0, stream) || sp->gather_wait(sp->user, 1, stream)) {
      wan_set_error("wan_dit_forward: K / V^T all-gather failed");
      return 3;
    }
    RC(wan_attention_sp_remote(b.q, b.kfull, b.vtfull, b.q, S, Ll, Ll, Lp, nh, world, rows * (int64_t)d, (int64_t)S * d * Lp, sp->rank,
                               b.kmax, b.raw, stream));
    return 0;
  }

  int self_attention_plain(const Layer& Lw) {
    const int S = Sn;
    auto& b = b2;
    for (int s = 0; s < S; ++s)  // (fp8: this quantises stream s of xm into slot s; q and k below reuse the slots)
      RC(linear(b.xm + (int64_t)s * Ll * d, Lw.self.v, b.vt + (int64_t)s * d * Lp, Ll, d, d, WAN_EPI_TRANSPOSED, stream, nullptr,
                nullptr, nullptr, -1, 1, Lp, q8, 1, s, false, a1));
    const bool vq = Lw.self.v.w8 != nullptr;  // slots hold xm already
    RC(linear(b.xm, Lw.self.q, b.q, rows, d, d, WAN_EPI_NONE, stream, nullptr, nullptr, nullptr, -1, 1, 0, q8, S, 0, vq, a1));
    RC(linear(b.xm, Lw.self.k, b.k, rows, d, d, WAN_EPI_NONE, stream, nullptr, nullptr, nullptr, -1, 1, 0, q8, S, 0, vq || Lw.self.q.w8, a1));
    {
      ProfScope ps(PROF_ROWOPS, st);  // fused RMSNorm(q,k)+RoPE: 4*rows*d*2 B
      RC(wan_rmsnorm_rope_scaled(b.q, b.k, Lw.self.nq, Lw.self.nk, cos, sin, rows, Ll, tok0, d, g.eps, wan_attention_qscale(),
                                 stream));
    }
    ProfScope ps(PROF_SELF_ATTN, st);
    RC(wan_attention_bounded(b.q, b.k, b.vt, b.q, S, S, Ll, Ll, Lp, nh, 1, 0, 0, 1, b.kmax, stream));
    return 0;
  }

  int o_projection(const Layer& Lw) {
    const int S = Sn;
    auto& b = b2;
    // outside the timed bracket: which share of the launch's workgroups failed the score bound and ran the tracking loop
    // (attention.hip hands the scratch to the kernels only for long KV -- Lk > 2048: for shorter sequences the flags are never written)
    if (!ulysses && g_prof_on && g_prof_declined != nullptr && b.kmax != nullptr && Ll > 2048) RC(wan_attention_count_declined(b.kmax, S, S, Ll, nh, g_prof_declined, stream));
    if (mx) {
      RC(linear_res32(b.q, Lw.self.o, xf, b.xm, rows, d, d, Lw.mod, e0f, 2, rpb, S));
    } else {
      RC(linear(b.q, Lw.self.o, b.x, rows, d, d, WAN_EPI_GATE_RES, stream, b.x, Lw.mod, b.e0, 2, rpb, 0, q8, S));
    }
    return 0;
  }

  int cross_attention(const Layer& Lw, const int li) {
    const int S = Sn;
    auto& b = b2;
    // -- cross attention (model.py:663-668, :245-265) --
    if (mx) RC(wan_mx_ln_affine(xf, b.xm, Lw.n3w32, Lw.n3b32, rows, d, g.eps, stream));
    else if (fold) {
      RC(zero_slots());
      RC(wan_ln_affine_amax(b.x, b.xm, Lw.n3w, Lw.n3b, rows, d, g.eps, q8->ws, Ll, stream));
    } else RC(wan_ln_affine(b.x, b.xm, Lw.n3w, Lw.n3b, rows, d, g.eps, stream));
    RC(linear(b.xm, Lw.cross.q, b.q, rows, d, d, WAN_EPI_NONE, stream, nullptr, nullptr, nullptr, -1, 1, 0, q8, S, 0, false, a1));
    // norm only; the softmax scale * log2(e) folded into q as for self-attention
    RC(wan_rmsnorm_rope_scaled(b.q, nullptr, Lw.cross.nq, nullptr, nullptr, nullptr, rows, Ll, 0, d, g.eps, wan_attention_qscale(), stream));
    if (any_nag) {
      // text_cross_attention with a (positive ; negative) context on some streams (model.py:260-292).  K / V per stream: its context
      // is ONE tensor of cb * TL rows (fp8: one quantisation), V^T one [d, TL] image per prompt.  Per stream: attention against the
      // positive prompt -> xm, against the negative one -> h (the FFN buffer is idle here), wan_nag_combine -> q (xm under the
      // CLIP branch, which adds the image result to the text result below).
      const int64_t crow0 = crow[s0];
      for (int s = 0; s < S; ++s) {
        const int64_t r0 = crow[s0 + s] - crow0, nr = (int64_t)cb[s0 + s] * TL;
        bf16_t* cs = b.ctx_e + r0 * d;
        bf16_t* ks = b.ck + r0 * d;
        bf16_t* vs = b.cvt + r0 * d;
        RC(linear(cs, Lw.cross.k, ks, nr, d, d, WAN_EPI_NONE, stream, nullptr, nullptr, nullptr, -1, 1, 0, q8, 1, s));
        RC(wan_rmsnorm_rope(ks, nullptr, Lw.cross.nk, nullptr, nullptr, nullptr, nr, TL, 0, d, g.eps, stream));
        if (Lw.cross.v.w8 == nullptr) {
          for (int j = 0; j < cb[s0 + s]; ++j)
            RC(wan_gemm_bf16(cs + (int64_t)j * TL * d, d, Lw.cross.v.w, Lw.cross.v.b, vs + (int64_t)j * TL * d, TL, TL, d, d, WAN_EPI_TRANSPOSED,
                             nullptr, nullptr, nullptr, 6, -1, 1, stream));
        } else {
          uint8_t* xq = q8->xq + (int64_t)s * q8->slot_bytes;
          float* ws = q8->ws + (int64_t)s * 64;
          if (Lw.cross.k.w8 == nullptr) RC(wan_fp8_quantize(cs, xq, ws, nr * d, stream));  // else: slot s holds this tensor already
          for (int j = 0; j < cb[s0 + s]; ++j)
            RC(wan_gemm_fp8(xq + (int64_t)j * TL * d, d, ws, Lw.cross.v.w8, Lw.cross.v.ws, Lw.cross.v.ns, Lw.cross.v.b, vs + (int64_t)j * TL * d, TL,
                            TL, d, d, WAN_EPI_TRANSPOSED, nullptr, nullptr, nullptr, 6, -1, 1, stream));
        }
        bf16_t* qs = b.q + (int64_t)s * Ll * d;
        bf16_t* xs = b.xm + (int64_t)s * Ll * d;
        bf16_t* dst = c->has_img ? xs : qs;
        ProfScope ps(PROF_CROSS_ATTN, st);
        if (cb[s0 + s] == 2) {
          bf16_t* hs = b.h + (int64_t)s * Ll * d;
          RC(wan_attention_bounded(qs, ks, vs, xs, 1, 1, Ll, TL, TL, nh, 1, 0, 0, 1, b.kmax, stream));
          RC(wan_attention_bounded(qs, ks + (int64_t)TL * d, vs + (int64_t)TL * d, hs, 1, 1, Ll, TL, TL, nh, 1, 0, 0, 1, b.kmax, stream));
          RC(wan_nag_combine(xs, hs, dst, Ll, d, nag[0], nag[1], nag[2], stream));
        } else {
          RC(wan_attention_bounded(qs, ks, vs, dst, 1, 1, Ll, TL, TL, nh, 1, 0, 0, 1, b.kmax, stream));
        }
      }
    } else if (!tc_hit) {
    // (TLx = text_len, or 257 + text_len under flf2v: b.ctx_e is then the assembled [CLIP tail ; text] context)
    // with a text cache: into the block's own buffers, where the next forward with this context finds them
    if (tc != nullptr && XT) WAN_CHECK_HIP(hipMemsetAsync(tc->cvt[li], 0, (size_t)S * d * LDVx * 2, st));
    bf16_t* const ckw = tc ? tc->ck[li] : b.ck;
    bf16_t* const cvtw = tc ? tc->cvt[li] : b.cvt;
    RC(linear(b.ctx_e, Lw.cross.k, ckw, (int64_t)S * TLx, d, d, WAN_EPI_NONE, stream, nullptr, nullptr, nullptr, -1, 1, 0, q8, S));
    RC(wan_rmsnorm_rope(ckw, nullptr, Lw.cross.nk, nullptr, nullptr, nullptr, (int64_t)S * TLx, TLx, 0, d, g.eps, stream));
    for (int s = 0; s < S; ++s)
      RC(linear(b.ctx_e + (int64_t)s * TLx * d, Lw.cross.v, cvtw + (int64_t)s * d * LDVx, TLx, d, d, WAN_EPI_TRANSPOSED, stream, nullptr,
                nullptr, nullptr, -1, 1, LDVx, q8, 1, s, Lw.cross.k.w8 != nullptr));
    }
    const bf16_t* const ck_l = tc ? tc->ck[li] : b.ck;
    const bf16_t* const cvt_l = tc ? tc->cvt[li] : b.cvt;
    // the attention's result: where the o projection reads its input.  The plain text branch attends OUT OF PLACE into the self-attention K
    // buffer (dead since the block's self-attention): 512 keys then take the K / V^T-stationary kernel (attention_xkv.hip, round 6), which
    // serves out-of-place calls only.  Every other branch ends in b.q as before.
    const bf16_t* co = b.q;
    if (!c->has_img) {
      if (!any_nag) {
        ProfScope ps(PROF_CROSS_ATTN, st);
        RC(wan_attention_bounded(b.q, ck_l, cvt_l, b.k, S, S, Ll, TLx, LDVx, nh, 1, 0, 0, 1, b.kmax, stream));
        co = b.k;
      }
    } else {
      // WanI2VCrossAttention (model.py:466-499): the same q attends the text tokens and the 257 CLIP tokens (K_img / V_img
      // shared by every stream), the two bf16 results are added, then o.  xm is free here: it takes the text result.
      RC(linear(c->clip_ctx, Lw.kimg, b.ckimg, CLIP_TOK, d, d, WAN_EPI_NONE, stream, nullptr, nullptr, nullptr, -1, 1, 0, q8));
      RC(wan_rmsnorm_rope(b.ckimg, nullptr, Lw.nkimg, nullptr, nullptr, nullptr, CLIP_TOK, CLIP_TOK, 0, d, g.eps, stream));
      RC(linear(c->clip_ctx, Lw.vimg, b.cvtimg, CLIP_TOK, d, d, WAN_EPI_TRANSPOSED, stream, nullptr, nullptr, nullptr, -1, 1, CLIP_LDV,
                q8, 1, 0, Lw.kimg.w8 != nullptr));
      ProfScope ps(PROF_CROSS_ATTN, st);
      if (!any_nag) RC(wan_attention_bounded(b.q, ck_l, cvt_l, b.xm, S, S, Ll, TLx, LDVx, nh, 1, 0, 0, 1, b.kmax, stream));
      RC(wan_attention_bounded(b.q, b.ckimg, b.cvtimg, b.q, S, 1, Ll, CLIP_TOK, CLIP_LDV, nh, 1, 0, 0, 1, nullptr, stream));
      RC(wan_add_bf16(b.xm, b.q, b.q, rows * (int64_t)d, stream));
    }
    if (mx) {
      RC(linear_res32(co, Lw.cross.o, xf, b.xm, rows, d, d, nullptr, nullptr, -1, rpb, S));
    } else {
      RC(linear(co, Lw.cross.o, b.x, rows, d, d, WAN_EPI_GATE_RES, stream, b.x, nullptr, nullptr, -1, rpb, 0, q8, S));
    }
    return 0;
  }

  int ffn_block(const Layer& Lw) {
    const int S = Sn;
    auto& b = b2;
    // -- FFN (model.py:686-711) --
    if (mx) RC(wan_mx_ln_modulate(xf, b.xm, Lw.mod, e0f, 6, 3, 4, rows, rpb, d, g.eps, stream));
    else if (fold) {
      RC(zero_slots());
      RC(wan_ln_modulate_amax(b.x, b.xm, Lw.mod, b.e0, 6, 3, 4, rows, rpb, d, g.eps, q8->ws, Ll, stream));
    } else RC(wan_ln_modulate(b.x, b.xm, Lw.mod, b.e0, 6, 3, 4, rows, rpb, d, g.eps, stream));
    {
      ProfScope ps(PROF_GEMM, st);  // the two FFN GEMMs: 4*rows*d*ffn FLOP
      // (ffn.2 reads h's abs-max from word 2 only if ffn.0 IS an fp8 Linear: a bf16 ffn.0 has no such epilogue)
      const int a2 = (fold && Lw.f0.w8 != nullptr) ? 2 : 0;
      RC(linear(b.xm, Lw.f0, b.h, rows, ffn, d, WAN_EPI_GELU_TANH, stream, nullptr, nullptr, nullptr, -1, 1, 0, q8, S, 0, false, a1, a2));
      if (mx) RC(linear_res32(b.h, Lw.f2, xf, b.xm, rows, d, ffn, Lw.mod, e0f, 5, rpb, S));
      else RC(linear(b.h, Lw.f2, b.x, rows, d, ffn, WAN_EPI_GATE_RES, stream, b.x, Lw.mod, b.e0, 5, rpb, 0, q8, S, 0, false, a2));
    }
    return 0;
  }

  // one WanAttentionBlock (model.py:575-724) on the token streams at b2.x, with the weights Lw; li: the block's index in the text cache
  // (main blocks, then VACE context blocks)
  int run_layer(const Layer& Lw, const int li) {
    RC(norm1(Lw));
    if (ulysses) RC(self_attention_ulysses(Lw));
    else if (world > 1) RC(self_attention_allgather(Lw));
    else RC(self_attention_plain(Lw));
    RC(o_projection(Lw));
    RC(cross_attention(Lw, li));
    return ffn_block(Lw);
  }

    // the block chain over streams [s0, s0 + Sn): every scratch buffer is used from its base, only the token stream and the
    // text context are offset (maximal runs of computing streams; all of them in the plain forward)
    // layers [l0, l1) of the block chain over streams [s0, s0 + Sn)
  int run_blocks(const int s0_, const int Sn_, const int l0, const int l1) {
    s0 = s0_;
    Sn = Sn_;
    const int S = Sn;
    rows = (int64_t)Sn * Ll; rpb = nt > 1 ? tpf : rows;
    b2 = B2{
        mx ? reinterpret_cast<bf16_t*>(x32 + s0 * sn) : b.x + s0 * sn, b.xm, b.q, b.k, b.vt, b.h, b.ck, b.cvt, XT ? b.ctx_h + (int64_t)s0 * TLx * d : b.ctx_e + (int64_t)crow[s0] * d, b.e0, b.kfull, b.vtfull, b.ckimg, b.cvtimg, b.kmax, b.raw};
    x_main = mx ? reinterpret_cast<bf16_t*>(x32 + s0 * sn) : b.x + s0 * sn;
    e0f = b.mx_e0;   // (the outer Bufs: captured before `b` is shadowed below)
    // hint streams of this run, per active context; vskip[k] doubles as the swap buffer of before_proj
    for (int j = 0; j < n_on; ++j) {
      vc[j] = b.vc + ((int64_t)on_k[j] * S_all + s0) * sn;
      vskip[j] = b.vskip + ((int64_t)on_k[j] * S_all + s0) * sn;
    }
    auto& b = b2;
    for (int i = l0; i < l1; ++i) {
      if (poll && poll(poll_user, i)) return WAN_ABORTED;  // model.py:1995-1998
      const int n = vace ? c->vace_at[i] : -1;
      if (n >= 0) {
        // VaceWanAttentionBlock.forward (model.py:816-828), called at the top of main block i (:617-629) once per active context:
        // the context block runs the same layer code on that context's hint streams, with the main streams' e0 / text context / RoPE.
        const Layer& Vw = c->vlayers[n];
        for (int j = 0; j < n_on; ++j) {
          if (n == 0) {  // c = before_proj(c) + x
            RC(linear(vc[j], Vw.before, vskip[j], rows, d, d, WAN_EPI_GATE_RES, stream, x_main, nullptr, nullptr, -1, rpb, 0, q8, S));
            bf16_t* t2 = vc[j]; vc[j] = vskip[j]; vskip[j] = t2;
          }
          b.x = vc[j];
          RC(run_layer(Vw, g.num_layers + n));
          b.x = x_main;
          RC(linear(vc[j], Vw.after, vskip[j], rows, d, d, WAN_EPI_NONE, stream, nullptr, nullptr, nullptr, -1, 1, 0, q8, S));  // c_skip = after_proj(c)
        }
      }
      RC(run_layer(c->layers[i], i));
      if (n >= 0)  // x.add_(hint[, alpha=scale]) per context, in context order (:713-719)
        for (int j = 0; j < n_on; ++j) RC(wan_axpy_bf16(b.x, vskip[j], vace_scales[on_k[j]], b.x, rows * (int64_t)d, stream));
    }
    return 0;
  }

  // maximal runs of computing streams through layers [l0, l1)
  int run_streams(const int l0, const int l1) {
    for (int s0 = 0; s0 < S;) {
      if (!calc(s0)) { ++s0; continue; }
      int Sn = 1;
      while (s0 + Sn < S && calc(s0 + Sn)) ++Sn;
      if (int rc = run_blocks(s0, Sn, l0, l1)) return rc;
      s0 += Sn;
    }
    return 0;
  }

  int block_chain() {
    if (n_perturb == 0) {
      RC(run_streams(0, g.num_layers));
    } else {
      // Skip-layer guidance (any2video.py:1502; model.py:2025-2028): a block listed in perturbation_layers runs for the FIRST stream
      // of the call only -- and only in the call that carries the conditional stream (x_id 0) -- every other stream passes through
      // it unchanged.  The chain is cut at those blocks: [unlisted blocks: every computing stream] [listed block: stream 0] ...
      WAN_REQUIRE(perturb_layers != nullptr && !vace, "wan_dit_forward: perturbation_layers %s",
                  vace ? "together with VACE context blocks is not implemented (the hint streams' state spans the chain)" : "is null");
      auto listed = [&](int i) {
        for (int k = 0; k < n_perturb; ++k)
          if (perturb_layers[k] == i) return true;
        return false;
      };
      for (int l = 0; l < g.num_layers;) {
        if (listed(l)) {
          if (x_id == 0 && calc(0)) RC(run_blocks(0, 1, l, l + 1));
          ++l;
          continue;
        }
        int e = l;
        while (e < g.num_layers && !listed(e)) ++e;
        RC(run_streams(l, e));
        l = e;
      }
    }
    return 0;
  }

  int skip_epilogue() {
    for (int s = 0; s < S; ++s)
      if (residual != nullptr && residual[s] != nullptr && calc(s)) {
        if (mx) {
          float* rs = reinterpret_cast<float*>(residual[s]);
          const float* in2[2] = {x32 + s * sn, rs};
          const float pm[2] = {1.f, -1.f};
          RC(wan_lincomb(rs, 2, in2, pm, sn, stream));                          // fp32 x - ori, one rounding
        } else {
          RC(wan_sub_bf16(b.x + s * sn, residual[s], residual[s], sn, stream));  // previous_residual = x - ori (model.py:2044-2062)
        }
      }
    return 0;
  }

  int head() {
    // ---- head + unpatchify (model.py:2068-2097) -------------------------------------------------------
    for (int s = 0; s < S; ++s) {
      if (mx) {
        // Head.forward on the fp32 stream (model.py:847-865): token-major result; one rank unpatchifies here, a sequence-parallel rank
        // hands its shard's rows to the host's gather like the bf16 plan does
        RC(wan_mx_head(x32 + (int64_t)s * Ll * d, c->head_mod, b.mx_e, c->head_w, c->head_b, b.mx_tmp, world > 1 ? outs[s] : b.mx_tok, Ll, d, g.eps,
                       nt > 1 ? tpf : Ll, 4 * g.out_dim, stream));
        if (world == 1) RC(wan_unpatchify_n(b.mx_tok, outs[s], 1, F, Hg, Wg, 4 * g.out_dim, stream));
      } else {
        RC(wan_head_range(b.x + (int64_t)s * Ll * d, c->head_mod, b.e, c->head_w, c->head_b, b.xm, outs[s], 1, F, Hg, Wg, d,
                          g.eps, tok0, Ll, world > 1 ? 1 : 0, nt > 1 ? tpf : 0, 4 * g.out_dim, stream));
      }
    }
    return 0;
  }

  int run() {
    RC(prepare());
    RC(embed());
    RC(skip_prologue());
    RC(vace_embed());
    RC(block_chain());
    RC(skip_epilogue());
    RC(head());
    if (tc != nullptr && !tc_hit) tc->valid = true;   // every block's K / V^T is enqueued (an aborted forward returned above: the slot stays invalid)
    return 0;
  }
};
}  // namespace

static int dit_forward_impl(wan_ctx* c, int S, const float* const* x, float t, const float* t_frames, const wan_bf16* const* context,
                            const float* y, const float* cos, const float* sin, float* const* outs, int F, int H,
                            int W, void* workspace, int64_t workspace_bytes, const wan_sp_info* sp, wan_poll_fn poll,
                            void* poll_user, const int* should_calc, wan_bf16* const* residual, int n_vace,
                            const float* const* vace_contexts, const float* vace_scales, const float* nag, const int* context_batches,
                            const int* perturb_layers, int n_perturb, int x_id, void* stream, const float* t_dev = nullptr,
                            uint64_t context_key = 0) {
  WAN_REQUIRE(c != nullptr, "wan_dit_forward: null argument");
  Forward f(c);
  f.S = S; f.x = x; f.t = t; f.t_frames = t_frames; f.context = context; f.y = y; f.cos = cos; f.sin = sin; f.outs = outs; f.F = F; f.H = H; f.W = W;
  f.workspace = workspace; f.workspace_bytes = workspace_bytes; f.sp = sp; f.poll = poll; f.poll_user = poll_user; f.should_calc = should_calc;
  f.residual = residual; f.n_vace = n_vace; f.vace_contexts = vace_contexts; f.vace_scales = vace_scales; f.nag = nag;
  f.context_batches = context_batches; f.perturb_layers = perturb_layers; f.n_perturb = n_perturb; f.x_id = x_id; f.stream = stream;
  f.t_dev = t_dev; f.context_key = context_key;
  return f.run();
}

extern "C" int wan_dit_forward(wan_ctx* c, int S, const float* const* x, float t, const wan_bf16* const* context,
                               const float* y, const float* cos, const float* sin, float* const* outs, int F, int H,
                               int W, void* workspace, int64_t workspace_bytes, const wan_sp_info* sp, wan_poll_fn poll,
                               void* poll_user, void* stream) {
  return dit_forward_impl(c, S, x, t, nullptr, context, y, cos, sin, outs, F, H, W, workspace, workspace_bytes, sp, poll, poll_user,
                          nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, stream);
}

extern "C" int wan_dit_forward_skip(wan_ctx* c, int S, const float* const* x, float t, const wan_bf16* const* context,
                                    const float* y, const float* cos, const float* sin, float* const* outs, int F, int H,
                                    int W, void* workspace, int64_t workspace_bytes, const wan_sp_info* sp, wan_poll_fn poll,
                                    void* poll_user, const int* should_calc, wan_bf16* const* residual, void* stream) {
  return dit_forward_impl(c, S, x, t, nullptr, context, y, cos, sin, outs, F, H, W, workspace, workspace_bytes, sp, poll, poll_user,
                          should_calc, residual, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, stream);
}

static int forward_ex_impl(wan_ctx* c, const wan_dit_args* a, void* stream, const float* t_dev) {
  WAN_REQUIRE(c && a, "wan_dit_forward_ex: null argument");
  WAN_REQUIRE(a->n_t_frames == 0 || (a->t_frames != nullptr && a->n_t_frames == a->F),
              "wan_dit_forward_ex: t_frames must hold one timestep per latent frame (%d given, F = %d)", a->n_t_frames, a->F);
  // one context through the original pair of fields, or n_vace of them through the arrays
  const float* one_ctx[1] = {a->vace_context};
  const float one_scale[1] = {a->vace_scale};
  const bool many = a->n_vace > 0;
  WAN_REQUIRE(!many || (a->vace_contexts && a->vace_scales), "wan_dit_forward_ex: n_vace = %d but the arrays are null", a->n_vace);
  const float nag[3] = {a->nag_scale, a->nag_tau, a->nag_alpha};
  WAN_REQUIRE(a->n_perturbation_layers <= 0 || a->perturbation_layers, "wan_dit_forward_ex: n_perturbation_layers = %d but the array is null",
              a->n_perturbation_layers);
  return dit_forward_impl(c, a->S, a->x, a->t, a->n_t_frames ? a->t_frames : nullptr, a->context, a->y, a->cos, a->sin, a->outs, a->F, a->H, a->W, a->workspace,
                          a->workspace_bytes, a->sp, a->poll, a->poll_user, a->should_calc, a->residual,
                          many ? a->n_vace : (a->vace_context ? 1 : 0), many ? a->vace_contexts : one_ctx, many ? a->vace_scales : one_scale,
                          nag, a->context_batches, a->n_perturbation_layers > 0 ? a->perturbation_layers : nullptr,
                          a->n_perturbation_layers > 0 ? a->n_perturbation_layers : 0, a->x_id, stream, t_dev, a->context_key);
}
extern "C" int wan_dit_forward_ex(wan_ctx* c, const wan_dit_args* a, void* stream) { return forward_ex_impl(c, a, stream, nullptr); }

// ---- the forward as a replayed launch list ("HIP-graph capture per (shape, expert)") -------------------
// At small L a forward is launch-bound: ~900 launches of a few microseconds each, enqueued one by one from the host (configs[0], the
// 1.3B model at L = 3,200: the host needs longer to enqueue a step than the GPU to run it).  wan_dit_forward_graph keys a call by
// everything its launches depend on EXCEPT the timestep -- shapes, every pointer (streams, contexts, outputs, rope tables, workspace),
// guidance parameters -- runs an unknown key eagerly once (lazy one-time allocations happen there), captures the SAME host code path
// into a hipGraph at the key's second appearance, and replays it from then on: one hipGraphLaunch per forward.  The timestep is read
// from device memory (set by a one-float kernel in front of the launch).  The kernels, their arguments and their order are those of
// the eager path: outputs are bit-identical (tests/test_gpu_model.py).  Not replayed (the call falls through to the eager forward,
// *how = 0): sequence parallelism (host hooks between launches), per-frame timesteps, step-skipping caches (flags change per step), the
// mixed-precision plan, profiling on.  The between-blocks poll of the eager path becomes ONE poll in front of the launch -- a forward
// this small is shorter than a pause request's latency.  *how: 0 eager (not eligible), 1 eager (first sight of the key), 2 captured
// and launched, 3 replayed.
template <typename T>
static void key_put(std::vector<uint8_t>& k, const T& v) {
  const uint8_t* p = reinterpret_cast<const uint8_t*>(&v);
  k.insert(k.end(), p, p + sizeof(T));
}

extern "C" int wan_dit_forward_graph(wan_ctx* c, const wan_dit_args* a, void* stream, int* how) {
  WAN_REQUIRE(c && a, "wan_dit_forward_graph: null argument");
  if (how) *how = 0;
  RC(resolve(c));
  const bool world1 = a->sp == nullptr || a->sp->world <= 1;
  // anything the eager entry would refuse (null arrays, stream counts ...) is ITS error to report: the key below reads through these pointers
  const bool well_formed = a->x && a->context && a->outs && a->S >= 1 && a->S <= 8 && a->n_vace >= 0 && a->n_vace <= 8 &&
                           (a->n_vace == 0 || (a->vace_contexts && a->vace_scales)) && a->n_perturbation_layers <= 64 &&
                           (a->n_perturbation_layers <= 0 || a->perturbation_layers);
  if (!well_formed || !world1 || a->n_t_frames != 0 || a->should_calc != nullptr || a->residual != nullptr || c->mixed || g_prof_on)
    return wan_dit_forward_ex(c, a, stream);
  std::vector<uint8_t> key;
  key_put(key, a->S); key_put(key, a->F); key_put(key, a->H); key_put(key, a->W);
  for (int s = 0; s < a->S; ++s) {
    key_put(key, a->x[s]); key_put(key, a->context[s]); key_put(key, a->outs[s]);
    key_put(key, a->context_batches ? a->context_batches[s] : 1);
  }
  key_put(key, a->y); key_put(key, a->cos); key_put(key, a->sin); key_put(key, a->workspace); key_put(key, a->workspace_bytes);
  key_put(key, a->vace_context); key_put(key, a->vace_scale); key_put(key, a->n_vace);
  for (int k = 0; k < a->n_vace; ++k) { key_put(key, a->vace_contexts[k]); key_put(key, a->vace_scales[k]); }
  key_put(key, a->nag_scale); key_put(key, a->nag_tau); key_put(key, a->nag_alpha);
  key_put(key, a->n_perturbation_layers);
  for (int k = 0; k < a->n_perturbation_layers; ++k) key_put(key, a->perturbation_layers[k]);
  key_put(key, a->x_id); key_put(key, c->clip_set);
  // the text cache (context_key): a launch list is captured in its HIT form only -- a call that misses runs eagerly and fills the slot --
  // and the slot's index is part of the key (the captured launches read that slot's buffers)
  if (a->context_key != 0) {
    bool nagb = false;
    for (int s = 0; a->context_batches != nullptr && s < a->S; ++s) nagb = nagb || a->context_batches[s] == 2;
    if (!nagb && a->n_perturbation_layers <= 0) {
      const int XTg = c->has_flf ? CLIP_TOK : 0;
      const int TLg = c->cfg.text_len + XTg;
      const int slot = text_cache_find(c, a->context_key, a->S, TLg, XTg ? ((TLg + 63) / 64) * 64 : c->cfg.text_len);
      if (slot < 0) {
        const int rc0 = wan_dit_forward_ex(c, a, stream);
        if (how) *how = 1;
        return rc0;
      }
      c->tcache[slot].last_use = ++c->tcache_clock;
      key_put(key, a->context_key); key_put(key, slot);
    }
  }
  if (a->poll && a->poll(a->poll_user, 0)) return WAN_ABORTED;
  hipStream_t st = as_stream(stream);
  if (c->t_dev == nullptr) WAN_CHECK_HIP(hipMalloc((void**)&c->t_dev, 256));
  RC(wan_set_f32(c->t_dev, a->t, stream));
  wan_dit_args b = *a;
  b.poll = nullptr;
  wan_ctx::GraphEntry* ent = nullptr;
  for (auto& e : c->graphs)
    if (e.key == key) ent = &e;
  if (ent != nullptr && ent->exec != nullptr) {                 // known: one launch
    ent->last_use = ++c->graph_clock;
    WAN_CHECK_HIP(hipGraphLaunch(ent->exec, st));
    if (how) *how = 3;
    return 0;
  }
  if (ent == nullptr) {                                         // first sight: eagerly, through the same t-from-memory path
    if (c->graphs.size() >= 8) {                                // a handful of live keys (experts x streams layouts); the oldest goes
      size_t old = 0;
      for (size_t i = 1; i < c->graphs.size(); ++i)
        if (c->graphs[i].last_use < c->graphs[old].last_use) old = i;
      if (c->graphs[old].exec) (void)hipGraphExecDestroy(c->graphs[old].exec);
      if (c->graphs[old].graph) (void)hipGraphDestroy(c->graphs[old].graph);
      c->graphs.erase(c->graphs.begin() + (long)old);
    }
    wan_ctx::GraphEntry e;
    e.key = key;
    e.last_use = ++c->graph_clock;
    c->graphs.push_back(e);
    const int rc = forward_ex_impl(c, &b, stream, c->t_dev);
    if (how) *how = 1;
    return rc;
  }
  // second sight: capture the launch list on the context's own stream, instantiate, launch on the caller's.  A runtime that refuses
  // (capture or instantiation error) costs nothing but the replay: nothing captured has run, the call is enqueued eagerly instead,
  // the key is marked and *how says 1; wan_last_error() keeps the reason.
  auto eager = [&]() -> int {
    const int rc2 = forward_ex_impl(c, &b, stream, c->t_dev);
    if (how) *how = 1;
    return rc2;
  };
  if (ent->bad) return eager();
  if (c->cap_stream == nullptr && hipStreamCreateWithFlags(&c->cap_stream, hipStreamNonBlocking) != hipSuccess) {
    (void)hipGetLastError();
    c->cap_stream = nullptr;
    ent->bad = true;
    return eager();
  }
  if (hipStreamBeginCapture(c->cap_stream, hipStreamCaptureModeRelaxed) != hipSuccess) {
    (void)hipGetLastError();
    ent->bad = true;
    wan_set_error("wan_dit_forward_graph: hipStreamBeginCapture failed; the forward stays on the eager path");
    return eager();
  }
  const int rc = forward_ex_impl(c, &b, c->cap_stream, c->t_dev);
  hipGraph_t graph = nullptr;
  const hipError_t ec = hipStreamEndCapture(c->cap_stream, &graph);
  if (rc != 0) {                                                 // the forward itself refused its arguments: that is the caller's error
    if (graph) (void)hipGraphDestroy(graph);
    (void)hipGetLastError();
    return rc;
  }
  hipGraphExec_t exec = nullptr;
  hipError_t ei = ec;
  if (ec == hipSuccess && graph != nullptr) ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  if (ec != hipSuccess || graph == nullptr || ei != hipSuccess || exec == nullptr) {
    if (graph) (void)hipGraphDestroy(graph);
    (void)hipGetLastError();
    ent->bad = true;
    const int rc2 = eager();
    wan_set_error("wan_dit_forward_graph: capture / instantiation failed (%s); the forward stays on the eager path",
                  hipGetErrorString(ec != hipSuccess ? ec : ei));
    return rc2;
  }
  ent->graph = graph;
  ent->exec = exec;
  ent->last_use = ++c->graph_clock;
  WAN_CHECK_HIP(hipGraphLaunch(exec, st));
  if (how) *how = 2;
  return 0;
}

extern "C" int wan_dit_set_vace_contexts(wan_ctx* c, int n) {
  WAN_REQUIRE(c && n >= 1 && n <= 8, "wan_dit_set_vace_contexts: 1..8 contexts, got %d", n);
  c->vace_max_ctx = n;
  drop_graphs(c);     // (the workspace layout moves)
  return 0;
}

extern "C" int wan_dit_set_vace_layers(wan_ctx* c, const int* layers, int n) {
  WAN_REQUIRE(c && (n == 0 || layers), "wan_dit_set_vace_layers: null argument");
  WAN_REQUIRE(n >= 0 && n <= c->cfg.num_layers, "wan_dit_set_vace_layers: %d context blocks for %d layers", n, c->cfg.num_layers);
  for (int i = 0; i < n; ++i)
    WAN_REQUIRE(layers[i] >= 0 && layers[i] < c->cfg.num_layers && (i == 0 ? layers[0] == 0 : layers[i] > layers[i - 1]),
                "wan_dit_set_vace_layers: layers must start at 0 and increase (model.py:1182)");
  c->vace_layers.assign(layers, layers + n);
  c->resolved = false;
  drop_graphs(c);
  return 0;
}
