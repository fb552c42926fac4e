// TEST INFRASTRUCTURE ONLY.  A recording stand-in for everything wan2gp_amd/csrc/dit.hip calls: the op-level C entry points of
// the library (same prototypes, from include/wanhip.h) and the dozen HIP runtime functions the forward driver uses.  Linked with
// the REAL dit.o it gives libwanhip_mock.so: wan_dit_forward* then runs on a host without a GPU and leaves the list of launches it
// would have enqueued -- which op, on which pointers, with which sizes.  tests/test_dit_host_logic_cpu.py checks the forward's HOST
// LOGIC on that list (workspace carving inside bounds, per-layer op sequence, stream subsets of step skipping, the NAG branch's
// pointer arithmetic, the sequence-parallel call order); the kernels themselves are checked on the GPU, op by op.
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/wanhip.h"

typedef uint16_t bf16_t;
struct Call {
  char name[32];
  uint64_t p[8];
  int64_t i[12];
  double f[4];
};
static std::vector<Call> g_calls;
static int rec(const char* name, std::initializer_list<const void*> ps, std::initializer_list<int64_t> is, std::initializer_list<double> fs = {}) {
  Call c;
  memset(&c, 0, sizeof(c));
  strncpy(c.name, name, sizeof(c.name) - 1);
  int k = 0;
  for (auto p : ps) c.p[k++] = (uint64_t)(uintptr_t)p;
  k = 0;
  for (auto v : is) c.i[k++] = v;
  k = 0;
  for (auto v : fs) c.f[k++] = v;
  g_calls.push_back(c);
  return 0;
}
extern "C" int mock_count(void) { return (int)g_calls.size(); }
extern "C" const Call* mock_get(int i) { return &g_calls[i]; }
extern "C" void mock_reset(void) { g_calls.clear(); }

// ---- HIP runtime ----------------------------------------------------------------------------------------------------------
extern "C" {
hipError_t hipMemsetAsync(void* dst, int value, size_t n, hipStream_t) { rec("memset", {dst}, {(int64_t)n, value}); return hipSuccess; }
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind, hipStream_t) { rec("memcpy", {dst, src}, {(int64_t)n}); return hipSuccess; }
hipError_t hipMalloc(void** p, size_t n) { *p = aligned_alloc(256, (n + 255) / 256 * 256); return hipSuccess; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemset(void* dst, int value, size_t n) { memset(dst, value, n); return hipSuccess; }
hipError_t hipMemcpy(void* dst, const void* src, size_t n, hipMemcpyKind) { memcpy(dst, src, n); return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "mock"; }
hipError_t hipGetLastError(void) { return hipSuccess; }
// stream capture / graphs (wan_dit_forward_graph): the "graph" is the span of recorded calls between begin and end; launching it
// records one call that names the span
static int g_capture_from = -1;
struct MockGraph { int from, to; };
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)(uintptr_t)0xCA97; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode) { rec("begin_capture", {(void*)s}, {}); g_capture_from = (int)g_calls.size(); return hipSuccess; }
hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t* g) {
  MockGraph* m = new MockGraph{g_capture_from, (int)g_calls.size()};
  *g = (hipGraph_t)m;
  rec("end_capture", {(void*)s}, {m->from, m->to});
  return hipSuccess;
}
hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t g, hipGraphNode_t*, char*, size_t) { *e = (hipGraphExec_t)g; return hipSuccess; }
hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t s) { MockGraph* m = (MockGraph*)e; rec("graph_launch", {(void*)s}, {m->from, m->to}); return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t g) { delete (MockGraph*)g; return hipSuccess; }
}

// ---- the library's op-level entries (prototypes from wanhip.h) -------------------------------------------------------------
extern "C" {
int wan_rmsnorm_rope(wan_bf16* q, wan_bf16* k, const wan_bf16* wq, const wan_bf16* wk, const float* cos, const float* sin, int64_t rows,
                     int64_t L, int64_t pos0, int d, float eps, void*) {
  return rec("rmsnorm_rope", {q, k, wq, wk, cos, sin}, {rows, L, pos0, d});
}
int wan_rmsnorm_rope_scaled(wan_bf16* q, wan_bf16* k, const wan_bf16* wq, const wan_bf16* wk, const float* cos, const float* sin,
                            int64_t rows, int64_t L, int64_t pos0, int d, float eps, float q_scale, void*) {
  return rec("rmsnorm_rope", {q, k, wq, wk, cos, sin}, {rows, L, pos0, d}, {q_scale});
}
int wan_rmsnorm_rope_pack(const wan_bf16* x, wan_bf16* pack, const wan_bf16* w, const float* cos, const float* sin, int64_t rows, int64_t L,
                          int64_t pos0, int d, float, float scale, int world, int heads_per_rank, int head_chunks, void*) {
  return rec("rmsnorm_rope_pack", {x, pack, w, cos, sin}, {rows, L, pos0, d, world, heads_per_rank, head_chunks}, {scale});
}
int wan_ln_modulate(const wan_bf16* x, wan_bf16* out, const wan_bf16* mod, const wan_bf16* e, int n_mod, int shift_idx, int scale_idx,
                    int64_t rows, int64_t rows_per_batch, int d, float, void*) {
  return rec("ln_modulate", {x, out, mod, e}, {rows, d, n_mod, shift_idx, scale_idx, rows_per_batch});
}
int wan_ln_affine(const wan_bf16* x, wan_bf16* out, const wan_bf16* w, const wan_bf16* b, int64_t rows, int d, float, void*) {
  return rec("ln_affine", {x, out, w, b}, {rows, d});
}
int wan_nag_combine(const wan_bf16* xp, const wan_bf16* xn, wan_bf16* out, int64_t rows, int d, float s, float tau, float alpha, void*) {
  return rec("nag_combine", {xp, xn, out}, {rows, d}, {s, tau, alpha});
}
int wan_gemm_bf16(const wan_bf16* A, int64_t lda, const wan_bf16* W, const wan_bf16* bias, wan_bf16* C, int64_t ldc, int64_t M, int N, int K,
                  int epi, const wan_bf16* R, const wan_bf16* mod, const wan_bf16* e, int n_mod, int gate_idx, int64_t rpb, void*) {
  return rec("gemm", {A, W, bias, C, R, mod, e}, {M, N, K, lda, ldc, epi, gate_idx, rpb});
}
int wan_fp8_quantize(const wan_bf16* x, uint8_t* out, float* ws, int64_t n, void*) { return rec("fp8_quantize", {x, out, ws}, {n}); }
int wan_fp8_quantize_pre(const wan_bf16* x, uint8_t* out, float* ws, int64_t n, int amax_word, void*) { return rec("fp8_quantize_pre", {x, out, ws}, {n, amax_word}); }
int wan_ln_modulate_amax(const wan_bf16* x, wan_bf16* out, const wan_bf16* mod, const wan_bf16* e, int n_mod, int shift_idx, int scale_idx,
                         int64_t rows, int64_t rows_per_batch, int d, float, float* amax_ws, int64_t rows_per_slot, void*) {
  return rec("ln_modulate_amax", {x, out, mod, e, amax_ws}, {rows, d, n_mod, shift_idx, scale_idx, rows_per_batch, rows_per_slot});
}
int wan_ln_affine_amax(const wan_bf16* x, wan_bf16* out, const wan_bf16* w, const wan_bf16* b, int64_t rows, int d, float, float* amax_ws,
                       int64_t rows_per_slot, void*) {
  return rec("ln_affine_amax", {x, out, w, b, amax_ws}, {rows, d, rows_per_slot});
}
int wan_gemm_fp8_amax(const uint8_t* A, int64_t lda, const float* sa, const uint8_t* W, const float* wsc, int wn, const wan_bf16* bias, wan_bf16* C,
                      int64_t M, int N, int K, float* amax, void*) {
  return rec("gemm_fp8", {A, W, bias, C, nullptr, nullptr, amax, sa}, {M, N, K, lda, N, 1, -1, 1, wn});
}
int wan_gemm_fp8(const uint8_t* A, int64_t lda, const float* sa, const uint8_t* W, const float* wsc, int wn, const wan_bf16* bias, wan_bf16* C,
                 int64_t ldc, int64_t M, int N, int K, int epi, const wan_bf16* R, const wan_bf16* mod, const wan_bf16* e, int n_mod, int gate_idx,
                 int64_t rpb, void*) {
  return rec("gemm_fp8", {A, W, bias, C, R, mod, e, sa}, {M, N, K, lda, ldc, epi, gate_idx, rpb, wn});
}
int64_t wan_attention_scratch_words(int B, int Bk, int64_t Lq, int H) { return (int64_t)Bk * H + (Lq + 255) / 256 * H * B + (int64_t)Bk * H; }
int64_t wan_attention_raw_words(int B, int64_t Lq, int H) { return (Lq + 255) / 256 * H * B * 4 * 2 * (64 * 64 + 64); }
float wan_attention_qscale(void) { return 0.12752041f; }
int wan_attention_bounded(const wan_bf16* q, const wan_bf16* k, const wan_bf16* vt, wan_bf16* o, int B, int Bk, int64_t Lq, int64_t Lk,
                          int64_t ldv, int H, int nseg, int64_t ks, int64_t vs, int pre, float* scratch, void*) {
  return rec("attention", {q, k, vt, o, scratch}, {B, Bk, Lq, Lk, ldv, H, nseg, ks, vs, pre});
}
int wan_attention_count_declined(const float* scratch, int B, int Bk, int64_t Lq, int H, uint64_t* acc, void*) {
  return rec("attn_count_declined", {scratch, acc}, {B, Bk, Lq, H});
}
int wan_attention_sp_local(const wan_bf16* q, const wan_bf16* k, const wan_bf16* vt, int B, int64_t Lq, int64_t Lk, int64_t ldv, int H,
                           float* scratch, float* raw, void*) {
  return rec("attention_sp_local", {q, k, vt, scratch, raw}, {B, Lq, Lk, ldv, H});
}
int wan_attention_sp_remote(const wan_bf16* q, const wan_bf16* k, const wan_bf16* vt, wan_bf16* o, int B, int64_t Lq, int64_t Lk, int64_t ldv,
                            int H, int nseg, int64_t ks, int64_t vs, int own, float* scratch, float* raw, void*) {
  return rec("attention_sp_remote", {q, k, vt, o, scratch, raw}, {B, Lq, Lk, ldv, H, nseg, ks, vs, own});
}
int wan_permute16(const void* src, void* dst, int64_t A, int64_t B, int64_t bytes, void*) { return rec("permute16", {src, dst}, {A, B, bytes}); }
int wan_permute16_ex(const void* src, void* dst, int64_t A, int64_t B, int64_t bytes, int64_t sa, int64_t sb, int64_t da, int64_t db, void*) {
  return rec("permute16_ex", {src, dst}, {A, B, bytes, sa, sb, da, db});
}
int wan_act_bf16(const wan_bf16* x, wan_bf16* y, int64_t n, int act, void*) { return rec("act", {x, y}, {n, act}); }
int wan_gemv_bf16(const wan_bf16* A, const wan_bf16* W, const wan_bf16* bias, wan_bf16* C, int M, int N, int K, void*) {
  return rec("gemv", {A, W, bias, C}, {M, N, K});
}
int wan_add_bf16(const wan_bf16* a, const wan_bf16* b, wan_bf16* out, int64_t n, void*) { return rec("add", {a, b, out}, {n}); }
int wan_sub_bf16(const wan_bf16* a, const wan_bf16* b, wan_bf16* out, int64_t n, void*) { return rec("sub", {a, b, out}, {n}); }
int wan_gemm_bf16_res32(const wan_bf16* A, int64_t lda, const wan_bf16* W, const wan_bf16* bias, float* x, wan_bf16* tmp, int64_t M, int N, int K,
                        const wan_bf16* mod, const float* e0, int n_mod, int gate_idx, int64_t rpb, void*) {
  return rec("gemm_res32", {A, W, bias, x, tmp, mod, e0}, {M, N, K, lda, n_mod, gate_idx, rpb});
}
int wan_lincomb(float* out, int n_in, const float* const* in, const float* coef, int64_t n, void*) {
  return rec("lincomb", {out, in[0], n_in > 1 ? in[1] : nullptr}, {n, n_in}, {coef[0], n_in > 1 ? coef[1] : 0.0});
}
int wan_axpy_bf16(const wan_bf16* x, const wan_bf16* y, float alpha, wan_bf16* out, int64_t n, void*) { return rec("axpy", {x, y, out}, {n}, {alpha}); }
// the mixed-precision plan's kernels (csrc/mixed_ops.hip)
int wan_mx_ln_modulate(const float* x, wan_bf16* out, const wan_bf16* mod, const float* e0, int n_mod, int shift_idx, int scale_idx, int64_t rows,
                       int64_t rows_per_batch, int d, float, void*) {
  return rec("mx_ln_modulate", {x, out, mod, e0}, {rows, d, n_mod, shift_idx, scale_idx, rows_per_batch});
}
int wan_mx_ln_affine(const float* x, wan_bf16* out, const float* w, const float* b, int64_t rows, int d, float, void*) {
  return rec("mx_ln_affine", {x, out, w, b}, {rows, d});
}
int wan_mx_gated_residual(float* x, const wan_bf16* y, const wan_bf16* mod, const float* e0, int n_mod, int gate_idx, int64_t rows, int64_t rpb, int d,
                          void*) {
  return rec("mx_gated_residual", {x, y, mod, e0}, {rows, d, n_mod, gate_idx, rpb});
}
int wan_mx_patch_embed(const float* x, const float* y, const float* w, const float* bias, float* out, int Cin, int Cy, int F, int H, int W, int d,
                       int64_t tok0, int64_t ntok, void*) {
  return rec("mx_patch_embed", {x, y, w, bias, out}, {Cin, Cy, F, H, W, d, tok0, ntok});
}
int wan_mx_sinusoid(float t, float* out, int dim, void*) { return rec("mx_sinusoid", {out}, {dim}, {t}); }
int wan_mx_linear_f32(const float* A, const float* W, const float* bias, float* C, int M, int N, int K, int act, void*) {
  return rec("mx_linear_f32", {A, W, bias, C}, {M, N, K, act});
}
int wan_mx_head(const float* x, const float* hmod, const float* e, const float* w, const float* bias, float* tmp, float* out, int64_t ntok, int d, float,
                int64_t e_rpb, int nout, void*) {
  return rec("mx_head", {x, hmod, e, w, bias, tmp, out}, {ntok, d, e_rpb, nout});
}
int wan_unpatchify_n(const float* in, float* out, int B, int F, int Hg, int Wg, int nout, void*) { return rec("unpatchify", {in, out}, {B, F, Hg, Wg, nout}); }
}
// internal (non-ABI) entry points of the other translation units, as dit.hip declares them
int wan_patch_embed_range(const float* x, const float* y, const float* w, const float* bias, bf16_t* out, int B, int Cin, int Cy, int F, int H,
                          int W, int d, int64_t tok0, int64_t ntok, void*) {
  return rec("patch_embed", {x, y, w, bias, out}, {B, Cin, Cy, F, H, W, d, tok0, ntok});
}
int wan_head_range(const bf16_t* x, const float* hmod, const bf16_t* e, const float* w, const float* bias, bf16_t* tmp, float* out, int B, int F,
                   int Hg, int Wg, int d, float, int64_t tok0, int64_t ntok, int token_major, int64_t e_rpb, int nout, void*) {
  return rec("head", {x, hmod, e, w, bias, tmp, out}, {B, F, Hg, Wg, d, tok0, ntok, token_major, e_rpb, nout});
}
int wan_sinusoid_val(float t, bf16_t* out, int dim, void*) { return rec("sinusoid", {out}, {dim}, {t}); }
int wan_set_f32(float* p, float v, void* stream) { return rec("set_f32", {p, stream}, {}, {v}); }
extern "C" int wan_sinusoid(const float* t, wan_bf16* out, int n, int dim, void*) { return rec("sinusoid_dev", {t, out}, {n, dim}); }
