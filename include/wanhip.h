/*
 * wanhip.h -- C ABI of libwanhip.so: the MI355X (gfx950) Wan 2.1/2.2 denoise hot path.
 *
 * The reference (deepbeepmeep/Wan2GP) has no FFI/operator registry on this path; its seams are
 * Python call sites.  Every entry point below names the reference interface it
 * replaces (file:line into /root/reference).  INTEGRATION.md shows the ctypes binding a
 * maintainer adds on the reference side.
 *
 * Conventions
 *   - all pointers are DEVICE pointers owned by the caller (torch); the library never frees them
 *   - bf16 tensors are raw uint16 storage, row-major, innermost dimension contiguous
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream)
 *   - every function returns 0 on success, non-zero on error (1 = rejected arguments, 2 = HIP runtime error,
 *     3 = a host callback failed, WAN_ABORTED = the interrupt poll asked to stop -- not an error: the reference
 *     returns [None]*n there); wan_last_error() returns the message for the calling thread.  Nothing here falls
 *     back to a CPU path.
 *   - kernels reproduce the reference's bf16 rounding points (RMSNorm 2 roundings, RoPE 1,
 *     LayerNorm 1 + modulate 2, Linear output 1, GELU 1, addcmul 1) so results track the
 *     reference's eager bf16 path; accumulation is fp32 everywhere.
 */
#ifndef WANHIP_H
#define WANHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint16_t wan_bf16; /* raw bfloat16 bits */
#define WAN_ABORTED 100   /* wan_dit_forward*: stopped by the between-blocks poll (model.py:1997-1998) */

/* ---- library ------------------------------------------------------------------------- */
const char* wan_last_error(void);
int wan_version(void);               /* ABI version, bumps on any signature change */
/* number of CUs of the current device (used by callers to size workspaces / report) */
int wan_device_cus(void);

/* ---- memory-bound fused ops ------------------------------------------------------------ */

/* Fused full-width RMSNorm(q) [+ RMSNorm(k)] [+ 3-axis RoPE(q,k)], in place.
 * Replaces WanRMSNorm.forward (models/wan/modules/model.py:160-175) applied to q and k
 * (model.py:343-344, :253,:256) followed by apply_rotary_emb (posemb_layers.py:288-340,
 * rotate :251-259) in WanSelfAttention.forward (model.py:345-350).
 *   q, k   : [rows, d] bf16, d = H*128; k may be NULL (cross-attn q: norm only)
 *   wq, wk : [d] bf16 RMSNorm weights
 *   cos,sin: [L, 128] fp32 tables (get_rotary_pos_embed, posemb_layers.py:492); NULL = no RoPE
 *   rows   : B*L token rows; token position = (row % L) + pos0  (pos0: first global token id
 *            of this rank's shard under sequence parallelism) */
int wan_rmsnorm_rope(wan_bf16* q, wan_bf16* k, const wan_bf16* wq, const wan_bf16* wk,
                     const float* cos, const float* sin, int64_t rows, int64_t L, int64_t pos0,
                     int d, float eps, void* stream);
/* Same, with q multiplied by q_scale in fp32 in front of its single bf16 rounding (k is untouched).
 * With q_scale = wan_attention_qscale() the result feeds wan_attention_prescaled: the softmax scale of
 * flash_attention (shared/attention.py:399 ff., softmax_scale = 1/sqrt(head_dim)) and the exp -> exp2
 * base change are folded into q, so the attention kernel's score tile comes out of the matrix pipe ready
 * for exp2.  q then carries ONE rounding, of q*scale instead of q (same relative error). */
int wan_rmsnorm_rope_scaled(wan_bf16* q, wan_bf16* k, const wan_bf16* wq, const wan_bf16* wk,
                            const float* cos, const float* sin, int64_t rows, int64_t L, int64_t pos0,
                            int d, float eps, float q_scale, void* stream);
/* wan_rmsnorm_rope_scaled on ONE tensor whose result goes to `pack` in the Ulysses exchange's send layout instead of in place (round 6):
 * x [rows, d] with d = world x heads_per_rank x 128 is read only; pack [chunk j][world][rows][128 (h0_j+1 - h0_j)], h0_j = j heads_per_rank /
 * head_chunks, receives what the in-place kernel followed by wan_permute16 (head_chunks 1) / wan_permute16_ex per chunk leaves there --
 * the same values, bit for bit, without the extra pass (wan_dit_forward, WAN_SP_ULYSSES: the q and k re-packs). */
int wan_rmsnorm_rope_pack(const wan_bf16* x, wan_bf16* pack, const wan_bf16* w, const float* cos, const float* sin, int64_t rows, int64_t L,
                          int64_t pos0, int d, float eps, float scale, int world, int heads_per_rank, int head_chunks, void* stream);

/* LayerNorm (no affine) + AdaLN modulate: out = bf16(bf16(LN(x) * bf16(1+scale)) + shift),
 * scale = bf16(mod[scale_idx] + e[b][scale_idx]), shift likewise.
 * Replaces norm1/norm2 + the two in-place ops of WanAttentionBlock.forward
 * (model.py:632-638, :686-691).  x,out [B*L, d] bf16; mod [n_mod, d] bf16 (modulation.weight);
 * e [B, n_mod, d] bf16 (e0); rows_per_batch = L. */
int wan_ln_modulate(const wan_bf16* x, wan_bf16* out, const wan_bf16* mod, const wan_bf16* e,
                    int n_mod, int shift_idx, int scale_idx, int64_t rows, int64_t rows_per_batch,
                    int d, float eps, void* stream);

/* LayerNorm with affine (norm3): out = bf16(LN(x)*w + b).  Replaces WanLayerNorm.forward with
 * elementwise_affine=True (model.py:199-212, used at :664). */
int wan_ln_affine(const wan_bf16* x, wan_bf16* out, const wan_bf16* w, const wan_bf16* b,
                  int64_t rows, int d, float eps, void* stream);

/* Gated residual x = bf16(x + y * gate), gate = bf16(mod[idx] + e[b][idx]) (idx<0: gate = 1,
 * i.e. x += y).  Replaces x.addcmul_(y, e[2]) / e[5] (model.py:658-660,:709-711) and the
 * cross-attention `x += ...` (:668).  Also available fused as a GEMM epilogue. */
int wan_gated_residual(wan_bf16* x, const wan_bf16* y, const wan_bf16* mod, const wan_bf16* e,
                       int n_mod, int gate_idx, int64_t rows, int64_t rows_per_batch, int d,
                       void* stream);

/* Normalized attention guidance on the two results of the text cross-attention (text_cross_attention, model.py:276-293; switched
 * on by WanAny2V.generate(NAG_scale > 1), any2video.py:607-608): x_pos = attention against the positive prompt, x_neg against
 * the negative one, both [rows, d] bf16.  g = bf16(bf16(x_neg (1 - s)) + s x_pos); rows with |g|_1 / |x_pos|_1 > tau are
 * scaled by bf16(tau |x_pos|_1 / |g|_1); out = bf16(bf16(alpha g) + bf16((1 - alpha) x_pos)) -- one bf16 rounding per
 * reference statement.  out may alias x_pos or x_neg. */
int wan_nag_combine(const wan_bf16* x_pos, const wan_bf16* x_neg, wan_bf16* out, int64_t rows, int d, float nag_scale,
                    float nag_tau, float nag_alpha, void* stream);

/* ---- GEMM ------------------------------------------------------------------------------ */
enum {
  WAN_EPI_NONE = 0,       /* C = bf16(A W^T + bias)                                  nn.Linear     */
  WAN_EPI_GELU_TANH = 1,  /* C = bf16(gelu_tanh(bf16(A W^T + bias)))   ffn[0]+ffn[1] model.py:552  */
  WAN_EPI_GATE_RES = 2,   /* C = bf16(R + bf16(A W^T + bias) * gate)   o-proj/ffn2 + addcmul_      */
  WAN_EPI_TRANSPOSED = 3  /* Ct[N, ldc] = bf16(A W^T + bias)^T : emits V^T for the attention kernel */
};

/* C[M,N] = epilogue(A[M,K] @ W[N,K]^T + bias[N]) on MFMA (bf16 in, fp32 accumulate).
 * Replaces every nn.Linear on the block path: self_attn.q/k/v/o (model.py:322,337,406),
 * cross_attn.q/k/v/o (:251-258,:444), ffn.0/ffn.2 (:703-705), text_embedding (:1856).
 *   lda/ldc : row strides in elements (lda of A, ldc of C / residual R)
 *   gate    : for WAN_EPI_GATE_RES: mod,e as in wan_gated_residual (gate_idx<0: gate=1)
 *   K % 64 == 0 required; M, N arbitrary. */
int wan_gemm_bf16(const wan_bf16* A, int64_t lda, const wan_bf16* W, const wan_bf16* bias,
                  wan_bf16* C, int64_t ldc, int64_t M, int N, int K, int epilogue,
                  const wan_bf16* R, const wan_bf16* mod, const wan_bf16* e, int n_mod,
                  int gate_idx, int64_t rows_per_batch, void* stream);

/* ---- scaled-fp8 Linear (fp8 checkpoints; BASELINE configs[4]) ---------------------------------------------------------
 * Replaces ScaledFP8WeightTensor._linear_scaled (shared/qtypes/scaled_fp8.py:324-380), the plan the reference runs for
 * `<name>.weight` float8_e4m3fn + `<name>.scale_weight` checkpoints (QLinearScaledFP8, :490-637) on a GPU with an fp8
 * matrix unit.  OCP e4m3fn bytes; both operands fp8, fp32 accumulation on v_mfma_f32_32x32x64_f8f6f4.
 *
 * wan_fp8_quantize: _quantize_activation (:162-169) -- per-tensor dynamic scale.  x [n] bf16 -> out [n] fp8 bytes;
 *   ws[0] <- scale_a = absmax / 448 (1 if the tensor is all zero), ws[1] = scratch (2 floats of device memory);
 *   q = fp8( clamp( bf16( x / bf16(scale_a) ), -448, 448 ) ), round-to-nearest-even.  n % 8 == 0.
 * wan_gemm_fp8: C[M,N] = epilogue( (A[M,K] W[N,K]^T) * scale_a * w_scale + bias ), A / W fp8, C bf16.
 *   scale_a: DEVICE pointer to the activation scale (ws of wan_fp8_quantize); w_scale: device fp32, w_scale_n = 1 (per
 *   tensor: scale and bias applied in fp32, one bf16 rounding -- torch._scaled_mm) or N (per output row: bf16(acc * scale_a),
 *   then `*= bf16(w_scale[n])`, then `+= bias`, one rounding each, :368-378).  epilogue / R / mod / e / gate as in
 *   wan_gemm_bf16.  K % 128 == 0, lda % 16 == 0. */
int wan_fp8_quantize(const wan_bf16* x, uint8_t* out, float* ws, int64_t n, void* stream);
int wan_gemm_fp8(const uint8_t* A, int64_t lda, const float* scale_a, const uint8_t* W, const float* w_scale, int w_scale_n,
                 const wan_bf16* bias, wan_bf16* C, int64_t ldc, int64_t M, int N, int K, int epilogue, const wan_bf16* R,
                 const wan_bf16* mod, const wan_bf16* e, int n_mod, int gate_idx, int64_t rows_per_batch, void* stream);
/* The abs-max pass of _quantize_activation folded into the kernel that PRODUCES the tensor (round 5): the maximum of |x| is the same
 * number whoever computes it, so the scale and every fp8 byte are the ones wan_fp8_quantize gives -- but the tensor is not read a second
 * time for its maximum (3 instead of 5 bytes per element move per quantisation).  Quantisation slots: 64 fp32 words per tensor (a
 * stream of the joint pass), word 0 = scale_a, word 1 / 2 = abs-max accumulators (float bits, atomicMax) the caller zeroes before the
 * producer runs.
 *   wan_ln_modulate_amax / wan_ln_affine_amax: wan_ln_modulate / wan_ln_affine, plus max |out| of every `rows_per_slot` rows into word 1
 *     of consecutive slots at amax_ws.
 *   wan_gemm_fp8_amax: wan_gemm_fp8 with WAN_EPI_GELU_TANH (ldc = N), plus max |C| into *amax (ffn.0's output is ffn.2's input).
 *   wan_fp8_quantize_pre: the quantising half of wan_fp8_quantize, reading the abs-max from ws[amax_word]; ws[0] <- scale_a. */
int wan_ln_modulate_amax(const wan_bf16* x, wan_bf16* out, const wan_bf16* mod, const wan_bf16* e, int n_mod, int shift_idx, int scale_idx,
                         int64_t rows, int64_t rows_per_batch, int d, float eps, float* amax_ws, int64_t rows_per_slot, void* stream);
int wan_ln_affine_amax(const wan_bf16* x, wan_bf16* out, const wan_bf16* w, const wan_bf16* b, int64_t rows, int d, float eps, float* amax_ws,
                       int64_t rows_per_slot, void* stream);
int wan_gemm_fp8_amax(const uint8_t* A, int64_t lda, const float* scale_a, const uint8_t* W, const float* w_scale, int w_scale_n,
                      const wan_bf16* bias, wan_bf16* C, int64_t M, int N, int K, float* amax, void* stream);
int wan_fp8_quantize_pre(const wan_bf16* x, uint8_t* out, float* ws, int64_t n, int amax_word, void* stream);

/* ---- attention ------------------------------------------------------------------------- */

/* Exact (non-causal, unmasked) flash attention, bf16 in/out, fp32 softmax/accumulate,
 * head_dim 128, scale 1/sqrt(128).  Replaces pay_attention(qkv_list) -> sdpa_wrapper
 * (shared/attention.py:360-373, :208-225) as called from model.py:264,385.
 *   q  : [B, Lq, H, 128]   (row stride H*128)
 *   k  : [Bk, Lk, H, 128]  Bk divides B: q batch b attends K / V^T batch b mod Bk (Bk == B: its own; Bk == 1: the
 *                          broadcast of attention.py:415; in between: the Ulysses layout, q batches = (source rank, stream))
 *   vt : [Bk, H*128, ldv]  V transposed (kv contiguous), columns [Lk, ldv) must be finite;
 *        ldv % 64 == 0.  Produced directly by wan_gemm_bf16(WAN_EPI_TRANSPOSED) or by
 *        wan_transpose_v for callers that hold V as [B, Lk, H, 128].
 *   o  : [B, Lq, H, 128] */
int wan_attention(const wan_bf16* q, const wan_bf16* k, const wan_bf16* vt, wan_bf16* o, int B,
                  int Bk, int64_t Lq, int64_t Lk, int64_t ldv, int H, void* stream);

/* Same, with K / V^T given as `nseg` equal segments of Lk rows (one per sequence-parallel rank,
 * as an all-gather leaves them: [seg][Bk][Lk][H*128] and [seg][Bk][H*128][ldv]); every
 * segment's tail tile is masked.  Strides in elements. */
int wan_attention_seg(const wan_bf16* q, const wan_bf16* k, const wan_bf16* vt, wan_bf16* o, int B,
                      int Bk, int64_t Lq, int64_t Lk, int64_t ldv, int H, int nseg,
                      int64_t k_seg_stride, int64_t vt_seg_stride, void* stream);
/* wan_attention_seg for a q that already holds q * wan_attention_qscale() (see wan_rmsnorm_rope_scaled):
 * runs the 4x64 kernel (csrc/attention_w64q.hip) without its in-kernel pre-scaling pass.  A K / V^T segment must stay
 * inside the kernels' 32-bit DMA offsets: Lk * H * 256 < 2^32 (419,430 rows at 40 heads). */
int wan_attention_prescaled(const wan_bf16* q, const wan_bf16* k, const wan_bf16* vt, wan_bf16* o, int B,
                            int Bk, int64_t Lq, int64_t Lk, int64_t ldv, int H, int nseg,
                            int64_t k_seg_stride, int64_t vt_seg_stride, void* stream);
/* The same kernel with caller-owned scratch for its K pre-pass: kmax_scratch = wan_attention_scratch_words(B, Bk, Lq, H)
 * 4-byte words of device memory (overwritten; NULL = no pre-pass).  The pre-pass leaves max |k_h|^2 per (batch, head) in
 * the first Bk*H floats; where |q~_row| * max|k_h| <= 96 (log2 units: no softmax term can overflow or go subnormal) for a
 * whole 256-row workgroup, that workgroup skips the running-max bookkeeping altogether (softmax is shift-invariant; bf16 /
 * fp32 carry P with relative precision); the others are flagged in the rest of the scratch and run the lazy-max loop in a
 * second launch.  wan_attention / _seg / _prescaled do the same with a library-owned scratch ring.  q_prescaled != 0:
 * q already holds q * wan_attention_qscale().
 * Short KV (one segment, 449 <= Lk <= 2048: text cross-attention, model.py:410-445) with a scratch takes the same bounded loop as ONE
 * persistent workgroup per CU that walks a run of q blocks and fetches the next block's Q rows while it works (round 4); without a
 * scratch, and below 449 keys, the lazy-max loop as before.  o may alias q: a workgroup that hands itself over to the lazy-max launch
 * stores nothing.
 * Exactly 512 keys in one segment, q pre-scaled, a scratch and o != q (round 6; what wan_dit_forward's text branch passes): the head's K and
 * V^T stay in the registers of one workgroup per CU and the Q rows stream past them (csrc/attention_xkv.hip).  No K pre-pass: a row is sound
 * when its row sum lies in [2^-80, 2^100]; a 256-row block with an unsound row is flagged in the scratch and the lazy-max launch redoes it
 * from its Q rows -- hence out of place only (o == q keeps the persistent walk).
 * wan_attention_debug_no_persist(n): test / A-B hook, returns the old value.  1: short KV with a scratch runs as ordinary one-block
 * workgroups; 2: the persistent walk also where the K / V^T-stationary kernel would serve the call; 0: the product dispatch. */
int64_t wan_attention_scratch_words(int B, int Bk, int64_t Lq, int H);
int wan_attention_bounded(const wan_bf16* q, const wan_bf16* k, const wan_bf16* vt, wan_bf16* o, int B, int Bk,
                          int64_t Lq, int64_t Lk, int64_t ldv, int H, int nseg, int64_t k_seg_stride,
                          int64_t vt_seg_stride, int q_prescaled, float* kmax_scratch, void* stream);
/* Sequence-parallel self-attention in two launches (q pre-scaled, Bk == B), so that compute starts before the K / V^T
 * all-gathers of the other ranks' segments have finished:
 *   wan_attention_sp_local : the rank's OWN segment k_local [B,Lk,H,128], vt_local [B,H*128,ldv] -> unnormalised partial sums
 *                            in raw (wan_attention_raw_words(B, Lq, H) floats);
 *   wan_attention_sp_remote: after the gathers -- every other segment of k_all / vt_all (layout of wan_attention_seg; own_seg is
 *                            skipped) on top of raw, normalised into o.
 * Partial sums add because the bounded softmax has no reference shift; workgroups whose rows exceed the bound are recomputed by
 * the lazy-max loop over all segments.  scratch: wan_attention_scratch_words(B, B, Lq, H) words, shared by the two calls. */
int64_t wan_attention_raw_words(int B, int64_t Lq, int H);
int wan_attention_sp_local(const wan_bf16* q, const wan_bf16* k_local, const wan_bf16* vt_local, int B, int64_t Lq, int64_t Lk,
                           int64_t ldv, int H, float* scratch, float* raw, void* stream);
int wan_attention_sp_remote(const wan_bf16* q, const wan_bf16* k_all, const wan_bf16* vt_all, wan_bf16* o, int B, int64_t Lq,
                            int64_t Lk, int64_t ldv, int H, int nseg, int64_t k_seg_stride, int64_t vt_seg_stride, int own_seg,
                            float* scratch, float* raw, void* stream);
/* (1/sqrt(128)) * log2(e): the factor wan_attention_prescaled expects folded into q */
float wan_attention_qscale(void);


/* vt[b, c, l] = v[b, l, c] for c < C, l < L; zero-fills l in [L, ldv). */
int wan_transpose_v(const wan_bf16* v, wan_bf16* vt, int B, int64_t L, int64_t ldv, int C,
                    void* stream);

/* ---- fp32 edge ops of the DiT ------------------------------------------------------------ */

/* patch_embedding: Conv3d k=s=(1,2,2) fp32 -> bf16 tokens [B, L, d] (model.py:1131,1631,1731).
 *   x : [B, Cin, F, H, W] fp32 (y : optional [Cy, F, H, W] fp32 concatenated on the channel
 *   axis for i2v2_2, model.py:1597-1600), w : [d, Cin+Cy, 1, 2, 2] fp32, bias [d] fp32 */
int wan_patch_embed(const float* x, const float* y, const float* w, const float* bias,
                    wan_bf16* out, int B, int Cin, int Cy, int F, int H, int W, int d,
                    void* stream);

/* Head: LN + 2-way modulate (fp32 modulation + bf16 e) + Linear(d -> 64) fp32 + unpatchify to
 * [B, 16, F, H, W] fp32 (model.py:847-865, :2100-2126, :2096).
 *   x [B, L, d] bf16; hmod [2, d] fp32; e [B, d] bf16; w [64, d] fp32; bias [64] fp32;
 *   tmp [B*L, d] bf16 scratch. */
int wan_head(const wan_bf16* x, const float* hmod, const wan_bf16* e, const float* w,
             const float* bias, wan_bf16* tmp, float* out, int B, int F, int Hg, int Wg, int d,
             float eps, void* stream);

/* token-major head output [B, L, 64] fp32 -> [B, 16, F, 2*Hg, 2*Wg]  ('fhwpqrc->cfphqwr',
 * model.py:2119-2121); used after the sequence-parallel gather of per-rank head outputs. */
int wan_unpatchify(const float* in, float* out, int B, int F, int Hg, int Wg, void* stream);
/* The same two for latents of out_dim = nout / 4 channels (48 for the ti2v 5B model, models/wan/configs/ti2v_2_2.json:
 * w [nout, d], bias [nout], out [B, nout/4, F, 2*Hg, 2*Wg], token-major input [B, L, nout]). */
int wan_head_n(const wan_bf16* x, const float* hmod, const wan_bf16* e, const float* w, const float* bias, wan_bf16* tmp,
               float* out, int B, int F, int Hg, int Wg, int d, float eps, int nout, void* stream);
int wan_unpatchify_n(const float* in, float* out, int B, int F, int Hg, int Wg, int nout, void* stream);

/* sinusoidal_embedding_1d(256, t) -> bf16 [n, dim] (model.py:32-42, :1816) */
int wan_sinusoid(const float* t, wan_bf16* out, int n, int dim, void* stream);
/* y = act(x) elementwise on bf16 (act: 1 = SiLU, used between the M=1 time MLP GEMVs; 2 = GELU(erf), img_emb) */
int wan_act_bf16(const wan_bf16* x, wan_bf16* y, int64_t n, int act, void* stream);
/* small-M Linear (GEMV): C[M,N] = bf16(A[M,K] W[N,K]^T + bias), M <= 16 (time MLP) */
int wan_gemv_bf16(const wan_bf16* A, const wan_bf16* W, const wan_bf16* bias, wan_bf16* C, int M,
                  int N, int K, void* stream);

/* ---- Wan2.2 VAE additions (models/wan/modules/vae2_2.py) -------------------------- */
/* The 5B ti2v VAE (z 48, stride (4,16,16)) reuses wan_vae_conv3d / wan_vae_rmsnorm_silu / wan_gemm_f16 / wan_vae_softmax and
 * wan_vae_pack / wan_vae_unpack; fp16 channels-last activations [T,H,W,C].  New data movement:
 *   wan_vae22_patchify:    video fp32 [3,T,H,W] -> [T,H/2,W/2,Cp] fp16, channel (c*2+r)*2+q = pixel (2h+q, 2w+r)
 *                          (patchify, vae2_2.py:299-315), channels 12..Cp-1 zero
 *   wan_vae22_to_video:    decoder head output fp32 [Ti,h,w,12] -> video [3,Ttot,2h,2w] at frame t0, as uint8
 *                          (_vae_float_to_cpu_uint8, vae.py:18-20) and/or fp32 (unpatchify, vae2_2.py:318-332)
 *   wan_vae22_avgdown_add: io[To,H/fs,W/fs,Co] += AvgDown3D(x[T,H,W,C]) (vae2_2.py:354-386, :466-471); To = ceil(T/ft)
 *   wan_vae22_dupup_add:   io[T*ft-(first_chunk?ft-1:0),H*fs,W*fs,Co] += DupUp3D(x[T,H,W,C]) (vae2_2.py:409-431, :508-516) */
int wan_vae22_patchify(const float* video, uint16_t* out, int T, int H, int W, int Cp, void* stream);
int wan_vae22_to_video(const float* y, uint8_t* u8, float* f32, int Ti, int h, int w, int Ttot, int t0, void* stream);
int wan_vae22_avgdown_add(const uint16_t* x, uint16_t* io, int T, int H, int W, int C, int Co, int ft, int fs, void* stream);
int wan_vae22_dupup_add(const uint16_t* x, uint16_t* io, int T, int H, int W, int C, int Co, int ft, int fs, int first_chunk,
                        void* stream);
/* The same four in the fp32 plan (`vae_precision` "32", wgp.py:4038 -> Wan2_2_VAE(dtype=torch.float32), vae2_2.py:1144-1160): fp32 channels-last
 * activations, no 16-bit rounding point (wan_vae22_to_video reads fp32 in both plans). */
int wan_vae22_patchify_f32(const float* video, float* out, int T, int H, int W, int Cp, void* stream);
int wan_vae22_avgdown_add_f32(const float* x, float* io, int T, int H, int W, int C, int Co, int ft, int fs, void* stream);
int wan_vae22_dupup_add_f32(const float* x, float* io, int T, int H, int W, int C, int Co, int ft, int fs, int first_chunk, void* stream);

/* ---- UMT5 text encoder (models/wan/modules/t5.py) ------------------------------ */
/* T5Attention core (t5.py:109-131) for head_dim 64: out = softmax_fp32(bf16(q k^T) + pos_bias, masked) v.
 * NO 1/sqrt(d) scaling.  q, k, v, out: [B, L, H*64] bf16 (the Linear outputs viewed per head).
 * relbias: [H, 2L-1] bf16 with relbias[h][j - i + L - 1] = T5RelativeEmbedding bias of key j for query i
 * (t5.py:232-263 evaluated once per distinct relative position).  mask: [B, L] int32, 0 = padding key
 * (masked_fill_(mask == 0, finfo.min), t5.py:119-123) or NULL.  L <= 1024.
 * The Linear layers of the encoder are wan_gemm_bf16 (bias = NULL), T5LayerNorm is wan_rmsnorm_rope(q, NULL, w, ...)
 * without RoPE (same two roundings, t5.py:66-71), residual adds are the WAN_EPI_GATE_RES epilogue with gate_idx -1. */
int wan_t5_attention(const wan_bf16* q, const wan_bf16* k, const wan_bf16* v, const wan_bf16* relbias,
                     const int32_t* mask, wan_bf16* out, int B, int L, int H, void* stream);
/* out = bf16(a * b) elementwise (T5FeedForward: fc1(x) * gate(x), t5.py:149); n % 8 == 0 */
int wan_mul_bf16(const wan_bf16* a, const wan_bf16* b, wan_bf16* out, int64_t n, void* stream);
/* out = bf16(a + b), out = bf16(a - b); operands may alias out.  The residual bookkeeping of the step-skipping caches
 * (model.py:1967-1971: x += previous_residual; :2044-2062: previous_residual = x - ori). */
int wan_add_bf16(const wan_bf16* a, const wan_bf16* b, wan_bf16* out, int64_t n, void* stream);
int wan_sub_bf16(const wan_bf16* a, const wan_bf16* b, wan_bf16* out, int64_t n, void* stream);
/* out = bf16(x + alpha * y) (torch's x.add_(y, alpha=alpha) on bf16: the scaled VACE hint, model.py:713-719); x may alias out */
int wan_axpy_bf16(const wan_bf16* x, const wan_bf16* y, float alpha, wan_bf16* out, int64_t n, void* stream);

/* ---- checkpoint load: LoRA merge + qint8 dequantisation -------------------------- */
/* The reference hands LoRA files to mmgp.offload.load_loras_into_model / activate_loras (wgp.py:6922-6931,
 * shared/utils/loras_mutipliers.py:143-148), whose patched Linear.forward adds m*(alpha/r)*(x A^T) B^T per call.  Resident
 * weights are merged instead: acc (fp32 scratch [N,K], zeroed by the caller) collects every adapter's delta, then the
 * bf16 weight takes one rounding.
 *   wan_lora_accumulate:   acc += scale * lora_B[N,r] @ lora_A[r,K]      (scale = multiplier * alpha / r)
 *   wan_axpy_f32:          acc += alpha * x                              (`.diff` / `.diff_b` full deltas)
 *   wan_add_f32_into_bf16: w = bf16(float(w) + acc)  elementwise
 *   wan_dequant_i8:        out[n,k] = bf16(float(data[n,k]) * scale[n])  (optimum-quanto qint8 `_data` / `_scale`,
 *                          the `quanto_*_int8` checkpoints of any2video.py:187-224) */
int wan_lora_accumulate(float* acc, const float* lora_B, const float* lora_A, float scale, int N, int K, int r, void* stream);
int wan_axpy_f32(float* acc, const float* x, float alpha, int64_t n, void* stream);
int wan_add_f32_into_bf16(wan_bf16* w, const float* acc, int64_t n, void* stream);
int wan_dequant_i8(const int8_t* data, const float* scale, wan_bf16* out, int64_t N, int64_t K, void* stream);

/* ---- sampler (fp32 latents) -------------------------------------------------------------- */

/* out = sum_i coef[i] * in[i]  (n_in <= 6), fp32.  The flow-matching scheduler updates
 * (fm_solvers_unipc.py:313-315 x0 = x - sigma*v, :350-480 UniP, :482-626 UniC;
 * euler_scheduler.py:79) and CFG (any2video.py:1722) are all such combinations with host
 * scalars; coefficients are computed on the host exactly as the reference does. */
/* Block transpose (the Ulysses re-packs, wan_dit_forward with WAN_SP_ULYSSES): src [A][B][bytes] -> dst [B][A][bytes], bytes a
 * multiple of 16, src != dst.  One pass at the copy rate. */
int wan_permute16(const void* src, void* dst, int64_t A, int64_t B, int64_t bytes, void* stream);
/* The same with explicit pitches (bytes, multiples of 16): dst[b * dst_b_pitch + a * dst_a_pitch + i] = src[a * src_a_pitch +
 * b * src_b_pitch + i] for a < A, b < B, i < bytes -- the per-head-chunk re-packs of the chunked Ulysses exchange (a column range
 * of [rows][d] rows into a [world][rows][Wc] block of its own, and back). */
int wan_permute16_ex(const void* src, void* dst, int64_t A, int64_t B, int64_t bytes, int64_t src_a_pitch, int64_t src_b_pitch,
                     int64_t dst_a_pitch, int64_t dst_b_pitch, void* stream);
int wan_lincomb(float* out, int n_in, const float* const* in, const float* coef, int64_t n,
                void* stream);

/* noise_pred = uncond + guide_scale * (cond - uncond), evaluated in that order
 * (classifier-free guidance combine, any2video.py:1722). */
int wan_cfg_combine(float* out, const float* cond, const float* uncond, float guide_scale, int64_t n,
                    void* stream);

/* ---- sampler step objects (SURVEY.md section 8b `wan_sched_*`) ------------------------------ */
/* The scheduler object WanAny2V.generate() builds (any2video.py:505-543) and steps once per denoise step (:1463-1467):
 *   kind 0  FlowUniPCMultistepScheduler(shift=1, use_dynamic_shifting=False)  -- shared/utils/fm_solvers_unipc.py:77-132,
 *           set_timesteps :163-239, step :640-721 (convert_model_output :279-348, UniP :350-480, UniC :482-626)
 *   kind 1  EulerScheduler(use_timestep_transform=True)                       -- shared/utils/euler_scheduler.py:26-87
 * Host scalars (sigmas, bh2 coefficients) are computed in the library with the reference's precision and order of
 * operations; each tensor update is one wan_lincomb launch on `stream`; the x0-prediction history and the corrected sample
 * (UniPC) live in library-owned fp32 device buffers sized at the first step.  Not re-entrant per object. */
typedef struct wan_sched wan_sched;
int wan_sched_create(wan_sched** out, int kind, int num_train_timesteps);
void wan_sched_destroy(wan_sched* s);
/* set_timesteps(num_inference_steps, shift=...): resets the multistep state.  timesteps_out[steps] (UniPC: the int64 values,
 * Euler: the fp32 values, both exact in a double) and sigmas_out[steps + 1] (UniPC only) may be NULL. */
int wan_sched_set_timesteps(wan_sched* s, int steps, double shift, double* timesteps_out, float* sigmas_out);
/* prev_sample = step(model_output, timestep, sample): fp32 device tensors of n elements; prev_out must not alias an input. */
int wan_sched_step(wan_sched* s, const float* model_output, double timestep, const float* sample, float* prev_out,
                   int64_t n, void* stream);

/* ---- whole-DiT context (weights resident in HBM) ----------------------------------------- */
typedef struct wan_ctx wan_ctx;

typedef struct {
  int dim, ffn_dim, num_heads, num_layers, in_dim, out_dim, text_dim, freq_dim, text_len;
  float eps;
} wan_dit_config;

/* Creates a DiT context.  Replaces the WanModel module object built in WanAny2V.__init__
 * (any2video.py:187-224) minus mmgp's offload machinery: all weights stay resident. */
int wan_dit_create(const wan_dit_config* cfg, wan_ctx** out);
void wan_dit_destroy(wan_ctx* ctx);
/* Registers a (device) weight by checkpoint key (models/wan/convert_wan.py:19-76), e.g.
 * "blocks.3.self_attn.q.weight".  dtype: 0 = bf16, 1 = fp32.  The pointer is borrowed. */
int wan_dit_set_weight(wan_ctx* ctx, const char* name, const void* ptr, int dtype, int64_t numel);
/* Bytes of scratch wan_dit_forward needs for S streams of B=1 and a (F,H,W) latent. */
int64_t wan_dit_workspace_bytes(const wan_ctx* ctx, int S, int F, int H, int W, int seq_shards);
/* interrupt/pause poll between blocks: return non-zero to abort (model.py:1995-1998) */
typedef int (*wan_poll_fn)(void* user, int block_idx);

/* Sequence-parallel hooks: the library calls back into the host runtime (torch.distributed /
 * RCCL) to all-gather the K and V^T shards; NULL = single GPU.
 *   gather_begin(user, which, send, recv, bytes_per_rank, stream): start the all-gather of `send`
 *       into recv[world][bytes]; which 0 = K, 1 = V^T.  May return before the collective has run
 *       (RCCL: enqueued on a side stream behind everything already on `stream`).
 *   gather_wait(user, which, stream): make `stream` wait for that collective.
 * wan_dit_forward issues the V^T gather right after the V projection so that it overlaps the Q/K
 * projections and the RMSNorm+RoPE kernel; both waits sit immediately before the attention launch. */
typedef int (*wan_gather_begin_fn)(void* user, int which, const void* send, void* recv, int64_t bytes,
                                   void* stream);
typedef int (*wan_gather_wait_fn)(void* user, int which, void* stream);
/* mode WAN_SP_ULYSSES (round 4): the other way to shard self-attention -- instead of gathering every rank's K / V^T, an
 * ALL-TO-ALL re-shards q, k, v from "my tokens, all heads" to "all tokens, my heads" (H % world == 0), the rank attends the whole
 * sequence for H / world heads in ONE launch, and a fourth all-to-all brings the output back to token shards.  Per block and rank
 * 4 x (world - 1) / world x S L/world d 2 bytes move (0.68 GB at 8 ranks, 14B, 720p x 81 f) instead of 2 (world - 1) of them
 * (2.71 GB); the k and v exchanges hide under the V and Q projections, the q and o exchanges do not.
 *   a2a_begin(user, which, send, recv, bytes_per_peer, stream): send = world chunks of bytes_per_peer, chunk j for rank j; recv
 *       = world chunks, chunk i from rank i; which 0 = k, 1 = v^T, 2 = q, 3 = o.  Same ordering contract as gather_begin.
 *   a2a_wait(user, which, stream).
 * Round 5: a2a_chunks = C > 1 splits the rank's H / world heads into C chunks (heads [c Hn / C, (c + 1) Hn / C)) and every tensor travels
 * PER CHUNK -- which = c: k chunk c, C + c: v^T, 2 C + c: q, 3 C + c: o (C = 1: the numbering above).  Chunk 0's k, v^T, q leave first (under
 * the V and Q projections), the other chunks flow under chunk 0's attention launch, chunk c's o returns under chunk c + 1's launch: only
 * q chunk 0 and the last o chunk are exposed.  Results are bit-identical to C = 1.  C is clamped to [1, min(H / world, WAN_SP_MAX_CHUNKS)]. */
enum { WAN_SP_ALLGATHER = 0, WAN_SP_ULYSSES = 1 };
enum { WAN_SP_MAX_CHUNKS = 8 };
typedef struct {
  int rank, world;          /* this rank, number of sequence shards */
  int64_t tok0, tok_local;  /* first global token and number of local tokens */
  wan_gather_begin_fn gather_begin;
  wan_gather_wait_fn gather_wait;
  void* user;
  int mode;                 /* WAN_SP_ALLGATHER (gather_* are used) or WAN_SP_ULYSSES (a2a_* are used) */
  wan_gather_begin_fn a2a_begin;
  wan_gather_wait_fn a2a_wait;
  int a2a_chunks;           /* WAN_SP_ULYSSES: head chunks of the q / o exchanges (0 or 1: one exchange each, the round-4 form) */
} wan_sp_info;

/* A library-owned RCCL communicator for those hooks (SURVEY.md section 8b `wan_sp_init(rank, nranks, ncclUniqueId)`): one
 * communicator per process / GPU, a side HIP stream and one completion event per gather slot.  Rank 0 makes the 128-byte id
 * (wan_sp_unique_id), the host runtime hands it to every rank, every rank calls wan_sp_init (collective; binds to the current
 * device).  wan_sp_gather_begin / wan_sp_gather_wait have the hook signatures: set wan_sp_info.gather_begin / gather_wait to
 * them and `user` to the wan_sp* and wan_dit_forward drives its all-gathers without leaving the library.  RCCL is bound at run
 * time (dlopen), sharing the instance the process already carries (PyTorch's).  wan_sp_all_gather: the same collective ordered
 * on `stream` on both sides. */
typedef struct wan_sp wan_sp;
int wan_sp_unique_id(void* id128);
int wan_sp_init(wan_sp** out, int rank, int nranks, const void* id128);
void wan_sp_destroy(wan_sp* sp);
int wan_sp_gather_begin(void* sp, int which, const void* send, void* recv, int64_t bytes, void* stream);
int wan_sp_gather_wait(void* sp, int which, void* stream);
/* the all-to-all pair of WAN_SP_ULYSSES on the same communicator and side stream (grouped ncclSend / ncclRecv: `bytes` to and from
 * every peer; the rank's own chunk is a device-to-device copy); which = 0 .. 4 WAN_SP_MAX_CHUNKS - 1; wait with wan_sp_gather_wait */
int wan_sp_a2a_begin(void* sp, int which, const void* send, void* recv, int64_t bytes, void* stream);
int wan_sp_all_gather(wan_sp* sp, const void* send, void* recv, int64_t bytes, void* stream);

/* WanModel.forward for the t2v / i2v2_2 path (model.py:1485-2098): S streams (the joint CFG
 * pass, any2video.py:1626-1634), each x_s [1, 16, F, H, W] fp32, t scalar, context_s
 * [1, 512, text_dim] bf16, y optional [in_dim-out_dim, F, H, W] fp32 (x streams are [1, out_dim, F, H, W]), cos/sin [L,128] fp32.
 * outs[s] [1, 16, F, H, W] fp32.  Returns WAN_ABORTED if stopped by poll (reference returns [None]*n). */
int wan_dit_forward(wan_ctx* ctx, int S, const float* const* x, float t, const wan_bf16* const* context,
                    const float* y, const float* cos, const float* sin, float* const* outs, int F,
                    int H, int W, void* workspace, int64_t workspace_bytes, const wan_sp_info* sp,
                    wan_poll_fn poll, void* poll_user, void* stream);
/* Every argument of a forward in one struct (the positional entry points above/below are wrappers around it):
 * wan_dit_forward's arguments, the step-skipping pair of wan_dit_forward_skip, and VACE (model.py:790-828, :1905-1912):
 * vace_context [vace_in_dim, F, H, W] fp32 holding bf16-representable values (the reference feeds the bf16 conv
 * vace_patch_embedding with u.to(weight.dtype)), vace_scale = vace_context_scale[0]; NULL = no VACE. */
typedef struct {
  int S;
  const float* const* x;
  float t;
  const wan_bf16* const* context;
  const float* y;
  const float* cos;
  const float* sin;
  float* const* outs;
  int F, H, W;
  void* workspace;
  int64_t workspace_bytes;
  const wan_sp_info* sp;
  wan_poll_fn poll;
  void* poll_user;
  const int* should_calc;
  wan_bf16* const* residual;
  const float* vace_context;
  float vace_scale;
  /* per-frame timesteps (model.py:1812-1818: t is a vector with one entry per latent frame -- ti2v image conditioning,
   * any2video.py:1496-1499, diffusion forcing): HOST array of F floats, n_t_frames = F; n_t_frames = 0: the scalar t.
   * Tokens of frame f are modulated by e0[f] (model.py:631-638) and the head by e[f] (:856-862). */
  const float* t_frames;
  int n_t_frames;
  /* several VACE contexts in one call (model.py:1905-1912 one hint list per context, :617-629 each through the context block,
   * :713-719 added to x in context order with its own scale; scale 0 = that context is off).  n_vace > 0: HOST arrays of
   * n_vace device pointers / scales replace the vace_context / vace_scale pair above; n_vace <= wan_dit_set_vace_contexts. */
  int n_vace;
  const float* const* vace_contexts;
  const float* vace_scales;
  /* normalized attention guidance (any2video.py:607-608; text_cross_attention, model.py:245-293): nag_scale > 1 and
   * context_batches[s] == 2 -> context[s] is [2, text_len, text_dim] = (positive ; negative) prompt, stream s's text
   * cross-attention runs against both and combines them with wan_nag_combine (in every block, VACE context blocks included).
   * context_batches: HOST array of S ints (1 or 2), NULL = every context has batch 1; a 2 needs nag_scale > 1. */
  float nag_scale, nag_tau, nag_alpha;
  const int* context_batches;
  /* skip-layer guidance (any2video.py:1502; WanModel.forward perturbation_layers, model.py:2025-2028): HOST array of block
   * indices that run for the FIRST stream of the call only, and only when x_id == 0 (the call that carries the conditional
   * stream); every other stream passes through them unchanged.  n_perturbation_layers = 0: off.  Not together with VACE. */
  const int* perturbation_layers;
  int n_perturbation_layers;
  int x_id;
  /* Text cache (round 6, wan_version() >= 8).  0 = off.  Non-zero: the caller's name for the CONTENTS of `context` -- while it passes the
   * same value, the same S and the same registered weights, the text embedding (model.py:1856) and every block's cross-attention K / V^T
   * (model.py:421-433) of this forward are the previous forward's: the library keeps them in buffers of the context's own (two keys; 2 x S x
   * text_len x dim x 2 bytes per block and tensor) and skips their 3 launches per block + 3 per forward.  Same kernels on the same
   * inputs: bit-identical to context_key 0.  The caller changes the value when the contents change (wan2gp_amd/model.py: identity +
   * version counter of the context tensors, which it keeps alive).  Not served (computed as with 0): normalized attention guidance,
   * skip-layer guidance, a step-skipping call that skips a stream. */
  uint64_t context_key;
} wan_dit_args;
int wan_dit_forward_ex(wan_ctx* ctx, const wan_dit_args* args, void* stream);
/* The same forward as a REPLAYED launch list: a call is keyed by everything its launches depend on except
 * the timestep (shapes, every pointer of args, guidance parameters); a key's first call runs eagerly, its second is captured into a
 * hipGraph, every later one is ONE hipGraphLaunch -- for small token counts, where enqueueing ~900 launches from the host takes longer
 * than running them (BASELINE configs[0]).  The caller keeps the pointers stable (staging buffers for the latents and outputs); the
 * timestep is read from device memory.  Same kernels, arguments and order as wan_dit_forward_ex: bit-identical outputs.  Calls that
 * cannot be replayed (sequence parallelism, per-frame timesteps, step-skipping caches, the mixed-precision plan, profiling on) fall
 * through to the eager forward.  poll is called once, in front of the launch.  *how (optional): 0 eager (not eligible), 1 eager
 * (first sight of the key), 2 captured + launched, 3 replayed.  Registering a weight drops the context's captured lists. */
int wan_dit_forward_graph(wan_ctx* ctx, const wan_dit_args* args, void* stream, int* how);
/* How many VACE contexts one forward may mix (default 1): sizes the hint-stream region of the workspace
 * (wan_dit_workspace_bytes grows by 2 x S x L x dim x 2 bytes per extra context). */
int wan_dit_set_vace_contexts(wan_ctx* ctx, int n);
/* VACE: main-block indices that carry a context block (WanModel(vace_layers=...), model.py:1178-1183; 0,5,...,35 for the
 * 14B VACE model).  Weights: vace_blocks.N.* (a block's keys + before_proj for N = 0 + after_proj), and
 * vace_patch_embedding.weight / .bias registered as fp32 copies of the bf16 parameters.  Call before the first forward /
 * wan_dit_workspace_bytes. */
int wan_dit_set_vace_layers(wan_ctx* ctx, const int* layers, int n);

/* Wan2.1 i2v (model_type 'i2v': checkpoints with img_emb.* and cross_attn.k_img / v_img / norm_k_img): projects the CLIP
 * vision features clip_fea [257, 1280] bf16 through img_emb (MLPProj, model.py:868-889, :1858-1859) and keeps the 257 image
 * tokens for the k_img / v_img branch of every block's cross-attention (WanI2VCrossAttention, model.py:448-499).  Must be
 * called before wan_dit_forward for such a model (the reference asserts clip_fea is not None, model.py:1547); the result
 * is kept in the context until the next call (the features do not depend on the step).
 * flf2v_720p (a checkpoint that also registers img_emb.emb_pos [514, 1280], MLPProj(flf_pos_emb=True), model.py:878-887): clip_fea
 * is [2 x 257, 1280] -- the start and the end image (any2video.py:949-950) --, the position embedding is added before the MLP,
 * and the blocks see what the reference's split at 257 gives them (model.py:472-473): the first image's 257 tokens in the
 * k_img / v_img branch, the second image's 257 tokens in front of the text tokens of the text branch.  Not together with
 * normalized attention guidance. */
int wan_dit_set_clip(wan_ctx* ctx, const wan_bf16* clip_fea, void* stream);

/* wan_dit_forward with the reference's step-skipping caches (TeaCache / MagCache, model.py:1373-1482 thresholds,
 * :1914-2064 skip logic; the decisions are host code): should_calc [S] (NULL = all), residual [S] bf16 buffers of
 * tokens_local * dim elements (NULL entries = stream not cached).  A computing stream with a buffer leaves
 * residual = x_after_blocks - x_after_patch_embed there; a skipped stream gets x = patch_embed(x) + residual and goes
 * straight to the head.  A context on the mixed-precision plan (fp32 time_projection / norm3 weights registered) keeps its
 * residual stream in fp32, and so its residual buffers: tokens_local * dim FLOATS behind the same pointers. */
int wan_dit_forward_skip(wan_ctx* ctx, int S, const float* const* x, float t, const wan_bf16* const* context,
                    const float* y, const float* cos, const float* sin, float* const* outs, int F,
                    int H, int W, void* workspace, int64_t workspace_bytes, const wan_sp_info* sp,
                    wan_poll_fn poll, void* poll_user, const int* should_calc, wan_bf16* const* residual,
                         void* stream);

/* ---- causal 3D VAE (fp16, channels-last [T,H,W,C], C % 32 == 0) ---------------------------------
 * Replaces the torch ops of models/wan/modules/vae.py; the layer graph / cache bookkeeping of
 * Encoder3d / Decoder3d / WanVAE_.encode / .decode stays on the host (wan2gp_amd/vae.py). */

/* CausalConv3d (vae.py:43-82) / Conv2d of Resample (:124-141) as an implicit GEMM on MFMA.
 *   x     : [Tin,Hin,Win,Cin] fp16;  cache: the layer's previous 2 input frames [2,Hin,Win,Cin] or NULL
 *           (NULL = causal zero padding, first chunk)
 *   w     : packed weights [Cout][Kp] fp16, K index = ((kt*KH+kh)*KW+kw)*Cin + c, Kp = K rounded up to 64
 *   res   : optional residual [Tout,Hout,Wout,Cout] added after rounding the conv output to fp16
 *   out   : fp16 output, or out_f32 (fp32, used for the 3-channel decoder head)
 *   front : temporal context in front of output frame 0 (2: causal k=3; 1: stride-2 time_conv that
 *           prepends the last cached frame, vae.py:205-206; 0: none);  st_t/st_s: strides
 *   pad_s : spatial low-side zero padding (1 for 3x3 "same", 0 for 1x1 and the (0,1,0,1)-padded stride-2 conv)
 *   ups   : 1 = the input is nearest-exact 2x upsampled on the fly (Upsample + Conv2d, vae.py:126-128)
 *   interleave : 1 = time_conv epilogue: output channels [s*C2,(s+1)*C2) go to frame 2t+s (vae.py:186-189) */
int wan_vae_conv3d(const uint16_t* x, const uint16_t* cache, const uint16_t* w, const uint16_t* bias,
                   const uint16_t* res, uint16_t* out, float* out_f32, int Tin, int Hin, int Win, int Cin,
                   int Tout, int Hout, int Wout, int Cout, int KT, int KH, int KW, int st_t, int st_s,
                   int front, int pad_s, int ups, int interleave, void* stream);
/* RMS_norm (F.normalize * sqrt(C) * gamma, vae.py:97-103) [+ SiLU], per pixel over C channels */
int wan_vae_rmsnorm_silu(const uint16_t* x, uint16_t* out, const uint16_t* gamma, int64_t npix, int C,
                         int silu, void* stream);
/* fp16 GEMM for AttentionBlock (vae.py:294-315): C = scale * A W^T + bias (or its transpose) */
int wan_gemm_f16(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, const uint16_t* bias,
                 uint16_t* C, int64_t ldc, int64_t M, int64_t N, int K, float scale, int transposed,
                 void* stream);
/* wan_vae_conv3d with the causal cache as TWO frame pointers (round 6): cache1 = input frame -1, cache0 = input frame -2; cache0 NULL with
 * cache1 set: frame -2 reads as zeros; both NULL: no cache.  wan_vae_conv3d(x, cache, ...) == wan_vae_conv3d_ex(x, cache, cache + one frame, ...).
 * The frames may live in the input tensors of two earlier chunks: wan_vae_decode / wan_vae_encode keep those alive and pass pointers where
 * the reference clones the frames (vae.py:254-273, :149-212). */
int wan_vae_conv3d_ex(const uint16_t* x, const uint16_t* cache0, const uint16_t* cache1, const uint16_t* w, const uint16_t* bias,
                      const uint16_t* res, uint16_t* out, float* out_f32, int Tin, int Hin, int Win, int Cin,
                      int Tout, int Hout, int Wout, int Cout, int KT, int KH, int KW, int st_t, int st_s,
                      int front, int pad_s, int ups, int interleave, void* stream);
/* Test hook (not a product entry): force wan_vae_conv3d onto its 64-bit-offset instantiations, which inputs below 2^31
 * elements never reach; returns the previous setting.  Process-wide. */
int wan_vae_debug_force_big(int on);
/* Test / A-B hook: 1 = the 3 x 3 x 3 stride-1 layers stay on the gather kernel instead of the halo-patch kernel (csrc/vae_conv_halo.hip; the two
 * sum in different orders and agree to fp32 rounding, not bit for bit); returns the old value. */
int wan_vae_debug_no_halo(int on);
int wan_attention_debug_no_persist(int on);
/* Test / A-B hook: 0 = a bounded self-attention call is ONE launch over its q blocks x heads x batches (the form until round 5); 1 (default;
 * env WAN_ATTN_SPLIT_TAIL=0 at load for the other) = a call whose workgroups do not fill their last round of CUs attends that round's q
 * blocks as k key-range parts each and finishes them in a third launch (csrc/attention_w64q.hip split_tail).  Returns the old value. */
int wan_attention_debug_split_tail(int on);

/* ---- the mixed-precision transformer plan (`mixed_precision_transformer`: wgp.py:4039 "mixed_precision" -> any2video.py:190 ->
 * WanModel.lock_layers_dtypes(torch.float32), models/wan/modules/model.py:1330-1371).  The time MLP, the time projection and every
 * block's norm3 hold their (bf16-valued) weights in fp32; the modulation dtype (model.py:1545) is then fp32 and by type promotion the
 * residual stream x, e / e0, every AdaLN modulate and every gated residual run in fp32 between bf16 Linears (one rounding in front of
 * each: `.to(attention_dtype)`, model.py:650,665,692).  Row / edge kernels of that plan (csrc/mixed_ops.hip).  A wan_dit context runs the
 * plan when 'time_projection.1.weight' was registered as fp32 (wan_dit_set_weight dtype 1) -- the reference's own rule; 'time_embedding.*',
 * 'time_projection.*' and every 'blocks.N.norm3.*' must then be fp32, wan_dit_workspace_bytes grows by the fp32 stream, and the forward
 * refuses what keeps bf16 state of its own (step-skipping residuals, VACE context blocks, the Wan2.1 i2v CLIP branch). ---- */
/* norm1 / norm2 + modulate (model.py:634-638, :686-692): out = bf16( LN(x) * (1 + scale) + shift ) with LN, scale, shift in fp32;
 * scale = float(mod[scale_idx]) + e0[b][scale_idx].  x [rows, d] fp32; mod [n_mod, d] bf16; e0 [batches, n_mod, d] fp32. */
int wan_mx_ln_modulate(const float* x, wan_bf16* out, const wan_bf16* mod, const float* e0, int n_mod, int shift_idx, int scale_idx,
                       int64_t rows, int64_t rows_per_batch, int d, float eps, void* stream);
/* norm3 with its fp32 weight and bias (model.py:664-665; WanLayerNorm.forward :199-212): out = bf16( LN(x) * w + b ). */
int wan_mx_ln_affine(const float* x, wan_bf16* out, const float* w, const float* b, int64_t rows, int d, float eps, void* stream);
/* Test hook: on != 0 keeps wan_mx_ln_modulate / wan_mx_ln_affine on the generic row form (the row re-read from L2 for the second and
 * third pass) at every width; the register-resident form they take at the Wan widths must give the same bits. */
void wan_mx_debug_generic_rows(int on);
/* x.addcmul_(y, e[gate_idx]) (model.py:658, :708) on the fp32 stream, y the bf16 result of a Linear: x += y * (float(mod[gate_idx]) +
 * e0[b][gate_idx]) with the product rounded first; gate_idx < 0: x += y (the cross-attention residual, :668; mod, e0 unused). */
int wan_mx_gated_residual(float* x, const wan_bf16* y, const wan_bf16* mod, const float* e0, int n_mod, int gate_idx, int64_t rows,
                          int64_t rows_per_batch, int d, void* stream);
/* A bf16 Linear and that update in ONE launch (round 5): x (fp32, in place) += bf16(A W^T + bias) * gate on the accumulators of the
 * 256 x 256 tile GEMM -- what wan_dit_forward runs for the o projections and ffn.2 of the mixed plan with bf16 weights.  `tmp`
 * [M, N] bf16 is used only when the shape does not fit that kernel (then: wan_gemm_bf16 into tmp + wan_mx_gated_residual).
 * Bit-identical to the two-launch form. */
int wan_gemm_bf16_res32(const wan_bf16* A, int64_t lda, const wan_bf16* W, const wan_bf16* bias, float* x, wan_bf16* tmp, int64_t M, int N,
                        int K, const wan_bf16* mod, const float* e0, int n_mod, int gate_idx, int64_t rows_per_batch, void* stream);
/* patch_embedding(x).to(fp32) (model.py:1620-1631; i2v: y concatenated behind x's channels, :1597-1600): the fp32 Conv3d with
 * kernel = stride = (1, 2, 2), result left in fp32.  x [Cin, F, H, W], y [Cy, F, H, W] or NULL, w [d, Cin + Cy, 1, 2, 2], out
 * [ntok, d] = tokens tok0 .. tok0 + ntok of the f-major grid. */
int wan_mx_patch_embed(const float* x, const float* y, const float* w, const float* bias, float* out, int Cin, int Cy, int F, int H,
                       int W, int d, int64_t tok0, int64_t ntok, void* stream);
/* sinusoidal_embedding_1d(dim, t) in fp32 (model.py:32-42), and an fp32 Linear on few rows with an optional SiLU on its INPUT:
 * C[m][n] = bias[n] + sum_k act(A[m][k]) W[n][k] -- time_embedding and time_projection under the fp32 lock (model.py:1815-1818). */
int wan_mx_sinusoid(float t, float* out, int dim, void* stream);
int wan_mx_linear_f32(const float* A, const float* W, const float* bias, float* C, int M, int N, int K, int act, void* stream);
/* Head.forward (model.py:847-865) on the fp32 stream: LayerNorm, modulate with head.modulation + e in fp32 (tmp: fp32 [ntok, d]),
 * the fp32 head Linear; out [ntok, nout] token-major (wan_unpatchify_n lays it out as [out_dim, F, H, W]).  e [batches, d] fp32,
 * one row per e_rows_per_batch tokens. */
int wan_mx_head(const float* x, const float* hmod, const float* e, const float* w, const float* bias, float* tmp, float* out,
                int64_t ntok, int d, float eps, int64_t e_rows_per_batch, int nout, void* stream);
/* P[r,:L] = softmax(S[r,:L]); P[r,L:ld] = 0 */
int wan_vae_softmax(const uint16_t* S, uint16_t* P, int64_t rows, int L, int64_t ld, void* stream);
/* fp32 [C,thw] -> fp16 [thw,Cp] (zero padded channels), optional v*mul[c]+add[c] */
int wan_vae_pack(const float* in, uint16_t* out, const float* mul, const float* add, int C, int Cp,
                 int64_t thw, void* stream);
/* fp16 [thw,Cs] -> fp32 [C,thw], optional (v-sub[c])*mul[c] */
int wan_vae_unpack(const uint16_t* in, float* out, const float* sub, const float* mul, int C, int Cs,
                   int64_t thw, void* stream);
/* decoder head fp32 [T*hw,3] -> uint8 and/or fp32 [3,Ttot,hw] at frame t0; uint8 conversion is
 * _vae_float_to_cpu_uint8 (vae.py:18-20): clamp, +1, *127.5, round-half-even, clamp */
int wan_vae_to_video(const float* in, uint8_t* u8, float* f32, int T, int64_t hw, int Ttot, int t0,
                     void* stream);

/* ---- the fp32 plan of the Wan2.1 VAE (round 4; `vae_precision` "32", wgp.py:4038 -> WanVAE(dtype = torch.float32)): the same ops
 * on fp32 channels-last activations [T,H,W,C] and fp32 weights, fp32 throughout -- plain FMA kernels, an option beside the fp16
 * default (csrc/vae_f32.hip).  wan_vae_conv3d_f32: the arguments of wan_vae_conv3d, weights [Cout][ldw] with K = ((kt*KH+kh)*KW+kw)
 * * Cin + c, Cin % 16 == 0.  wan_gemm_f32: C[M][N] = scale * A[M][K] . (b_transposed ? B[N][K]^T : B[K][N]) (+ bias[N]) -- the
 * to_qkv / q k^T / p v products of AttentionBlock (vae.py:294-315).  wan_vae_softmax_f32: rows in place. */
int wan_vae_conv3d_f32(const float* x, const float* cache, const float* w, int64_t ldw, const float* bias, const float* res, float* out,
                       int Tin, int Hin, int Win, int Cin, int Tout, int Hout, int Wout, int Cout, int KT, int KH, int KW, int st_t,
                       int st_s, int front, int pad_s, int ups, int interleave, void* stream);
int wan_vae_rmsnorm_silu_f32(const float* x, float* out, const float* gamma, int64_t npix, int C, int silu, void* stream);
int wan_gemm_f32(const float* A, int64_t lda, const float* B, int64_t ldb, int b_transposed, const float* bias, float* C, int64_t ldc,
                 int M, int N, int K, float scale, void* stream);
int wan_vae_softmax_f32(float* S, int64_t rows, int L, int64_t ld, void* stream);
int wan_vae_pack_f32(const float* in, float* out, const float* mul, const float* add, int C, int Cp, int64_t thw, void* stream);
int wan_vae_unpack_f32(const float* in, float* out, const float* sub, const float* mul, int C, int Cs, int64_t thw, void* stream);

/* ---- the whole Wan2.1 VAE (SURVEY.md section 8b `wan_vae_encode`, `wan_vae_decode_u8`) ----------------------------------
 * Encoder3d / Decoder3d / WanVAE_.encode / .decode (vae.py:318-662) as one call each: the layer graph and the causal
 * feature-cache bookkeeping run inside the library (csrc/vae_graph.hip) on the op-level entry points above.
 * Registration (host fp32 arrays, packed to fp16 on the way in; names are the checkpoint's, e.g. "decoder.conv1",
 * "decoder.middle.0.residual.2", "decoder.upsamples.3.resample.1", "conv2"; gammas by their full key):
 *   wan_vae_set_conv(v, name, weight[cout,cin,kt,kh,kw], cout, cin, kt, kh, kw, bias[cout] or NULL, cout_pad)
 *       (Conv2d: kt = 1; cout_pad = 32 for "conv2", whose output feeds a 32-channel-padded tensor, else 0)
 *   wan_vae_set_gamma(v, "decoder.head.0.gamma", g[C], C);   wan_vae_set_attention(v, "decoder.middle.1.", wqkv[3C,C], bqkv[3C], C)
 * wan_vae_workspace_bytes(v, decode, t, h, w): decode = 1: latent [16,t,h,w]; decode = 0: video [3,t,h,w] (t = T frames,
 * h, w = pixels).  Runs the graph in planning mode -- call it after the weights are registered.  -1 on error.
 * wan_vae_decode: z [16,t,h,w] fp32 (normalised latents, WanVAE.decode's input) -> u8 [3,T,H,W] (decode_to_cpu_uint8's
 * conversion, vae.py:18-20) and/or f32 [3,T,H,W] (unclamped), T = 4(t-1)+1, H = 8h, W = 8w; either may be NULL.
 * wan_vae_encode: video [3,T,H,W] fp32 in [-1,1], T = 4k+1 -> mu [16,t,h,w] fp32, normalised (WanVAE.encode's output). */
typedef struct wan_vae wan_vae;
int wan_vae_create(wan_vae** out);
void wan_vae_destroy(wan_vae* v);
int wan_vae_set_conv(wan_vae* v, const char* name, const float* w, int cout, int cin, int kt, int kh, int kw, const float* bias,
                     int cout_pad);
int wan_vae_set_gamma(wan_vae* v, const char* name, const float* g, int C);
int wan_vae_set_attention(wan_vae* v, const char* prefix, const float* wqkv, const float* bqkv, int C);
int64_t wan_vae_workspace_bytes(wan_vae* v, int decode, int t, int h, int w);
int wan_vae_decode(wan_vae* v, const float* z, int t, int h, int w, uint8_t* u8, float* f32, void* workspace,
                   int64_t workspace_bytes, void* stream);
int wan_vae_encode(wan_vae* v, const float* video, int T, int H, int W, float* mu, void* workspace, int64_t workspace_bytes,
                   void* stream);

/* ---- measurement hooks (bench.py) ----------------------------------------------------------
 * wan_prof_enable(1): wan_dit_forward brackets each kernel class with HIP events recorded on the
 * launch stream; wan_prof_collect sums the elapsed ms per class (0 self-attention, 1 cross-
 * attention, 2 FFN GEMM pair, 3 fused RMSNorm+RoPE) and returns the bracket count. */
int wan_prof_enable(int on);
int wan_prof_collect(int cls, double* total_ms, int* count);
/* The bounded softmax of the self-attention kernel is taken per 256-row workgroup when |q~_row| max|k_h| <= 96 holds for all
 * of its rows; the others are flagged and run the tracking loop.  While profiling is enabled every self-attention launch of
 * wan_dit_forward adds its flagged / launched workgroup counts; this returns the sums since wan_prof_enable(1) (synchronises).
 * wan_attention_count_declined: the same count for one wan_attention_bounded launch given its scratch (acc: device uint64[2],
 * added to on `stream`). */
int wan_prof_attention_declined(int64_t* declined, int64_t* total);
/* What the matrix pipe sustains on this device right now: enqueues iters x 64 back-to-back v_mfma_f32_32x32x16_bf16 per wave, one
 * wave per SIMD on every CU, random bf16 operands in registers, no memory traffic (iters = 40,000 is ~50 ms).  *flop_out = the FLOP
 * of the launch; the caller times it (events on `stream`).  On MI355X the answer is the power limit, not the 2.5 PFLOP/s of the
 * data sheet: 1.7-1.8 PFLOP/s on random data (csrc/probe.hip). */
int wan_mfma_sustained_probe(int iters, double* flop_out, void* stream);
/* Measurement aid: occupies `stream` for `microseconds` (one lane spinning on the constant 100 MHz clock) and nothing else of the
 * chip -- bench.py's link model puts an exchange's xGMI transfer time behind the device-to-device copy that stands in for it. */
int wan_debug_delay(double microseconds, void* stream);
int wan_attention_count_declined(const float* scratch, int B, int Bk, int64_t Lq, int H, uint64_t* acc, void* stream);
/* Test / A-B hook of the bf16 GEMM dispatch (csrc/gemm_bf16.hip launch_gemm): which problems run on the co-resident small-tile kernel
 * csrc/gemm16s.hip (round 6).  0 = automatic: problems of fewer than 128 tiles of 256 x 256 (default); 128 / 256 = that tile height on every
 * problem the kernel accepts, many-tile ones included; -1 = never (the round-5 dispatch).  Returns the old value.  Process-wide. */
int wan_gemm_debug_force16s(int v);
/* Test / A-B hook: the tile height of csrc/gemm256m.hip (round 6: 256, 224, 192 or 160 rows x 256 columns).  0 = automatic -- the height whose
 * tiles leave the fewest CUs idle (BASELINE configs[0]: 160), 256 unless the gain is at least 8 %; 256 / 224 / 192 / 160 = that height on every
 * problem the kernel accepts, problems below 128 tiles included; -1 = the round-5 rule (256 rows, problems of 256 tiles and more only).  Returns the
 * old value.  Process-wide. */
int wan_gemm_debug_force_tile_rows(int v);

#ifdef __cplusplus
}
#endif
#endif /* WANHIP_H */
