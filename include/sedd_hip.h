/* libsedd_hip — C ABI of the MI355X (gfx950) SEDD hot path.
 *
 * The reference (architvasan/protein_lig_sedd) is pure Python and has no FFI of its own; its seams are Python
 * call sites into third-party native ops (flash-attn, cuBLAS via nn.Linear, TorchScript fusions, ATen
 * elementwise).  Each entry point below replaces one such call site; the comment above it cites the reference
 * file:line whose arithmetic it reproduces (paths relative to the reference root).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller; the library never allocates, frees or synchronises
 *    (one exception: sedd_ln_modulate_bwd_noalloc works in a file-scope __device__ scratch of the library, see there), launches are asynchronous on `stream` (a hipStream_t passed as void*).
 *  - tensors are dense row-major unless strides are given; index tensors are int64 (torch.long) as in the
 *    reference; per-batch scalars (sigma, dsigma, ...) are fp32 arrays of length B; M = B*L rows, row r belongs
 *    to batch r / L.
 *  - `dtype` arguments take SEDD_F32 or SEDD_BF16 and describe the activation/logit element type; reductions
 *    and all transcendental maths are fp32.
 *  - return value: 0 on success, <0 = sedd_status; sedd_last_error() gives a thread-local message.
 *  - re-entrant; no mutable global state (exceptions, documented at their declarations: sedd_gemm_set_cu_budget and the
 *    file-scope scratch of sedd_ln_modulate_bwd_noalloc).
 */
#ifndef SEDD_HIP_H
#define SEDD_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SEDD_ABI_VERSION 6

typedef void* sedd_stream_t; /* hipStream_t */

enum sedd_status { SEDD_OK = 0, SEDD_ERR_BAD_ARG = -1, SEDD_ERR_UNSUPPORTED_SHAPE = -2, SEDD_ERR_LAUNCH = -3 };
enum sedd_dtype { SEDD_F32 = 0, SEDD_BF16 = 1 };
enum sedd_graph { SEDD_ABSORB = 0, SEDD_UNIFORM = 1 };
/* predictor step kinds: sampling/sampling.py:59-80 (Euler), :88-99 (Analytic), :102-132 (Denoiser) */
enum sedd_step { SEDD_STEP_EULER = 0, SEDD_STEP_ANALYTIC = 1, SEDD_STEP_DENOISE = 2 };
/* GEMM epilogues */
enum sedd_epilogue { SEDD_EPI_NONE = 0, SEDD_EPI_BIAS = 1, SEDD_EPI_BIAS_GELU = 2, SEDD_EPI_BIAS_GELU_GRAD = 3 };

int sedd_abi_version(void);
const char* sedd_last_error(void);

/* ------------------------------------------------------------------------------------------------ graph ops
 * processing/graph_lib.py. V = graph.dim (absorbing: tokens+1, mask id V-1). */

/* Absorbing.sample_transition (graph_lib.py:258-262) / Uniform.sample_transition (:156-160):
 * xt = (u < 1 - exp(-sigma_b)) ? (absorb: V-1 | uniform: r) : x0.  u fp32 [M]; r int64 [M] (uniform only). */
int sedd_sample_transition(int graph, const int64_t* x0, const float* sigma, const float* u, const int64_t* r,
                           int64_t* xt, int64_t M, int64_t L, int V, sedd_stream_t stream);

/* Absorbing.score_entropy (graph_lib.py:274-314) / Uniform.score_entropy (:170-197). out fp32 [M].
 * If dscore != NULL also writes d(sum_r gw[r/L]*out[r]) / dscore (same dtype as score) — the fused
 * loss-forward+backward of losses.py:27-29 where gw_b = dsigma_b / M. */
int sedd_score_entropy_fwd(int graph, const void* score, int dtype, const float* sigma, const int64_t* x,
                           const int64_t* x0, float* out, void* dscore, const float* gw, int64_t M, int64_t L,
                           int V, sedd_stream_t stream);
/* autograd backward of the above for an arbitrary upstream gradient gout fp32 [M]. */
int sedd_score_entropy_bwd(int graph, const void* score, int dtype, const float* sigma, const int64_t* x,
                           const int64_t* x0, const float* gout, void* dscore, int64_t M, int64_t L, int V,
                           sedd_stream_t stream);

/* utils/catsample.py:10-13 with the uniform noise passed in: out = argmax_j p_j / (1e-10 - log(u_j + 1e-10)),
 * first index wins ties. p has `dtype`, u fp32, both [M, V]. */
int sedd_gumbel_argmax(const void* p, int dtype, const float* u, int64_t* out, int64_t M, int V,
                       sedd_stream_t stream);

/* One fused predictor update (sampling/sampling.py:61-80, :90-99, :107-132 with graph_lib.py:79-96, :162-165,
 * :226-234, :264-269 and catsample.py:12-13): log-score -> exp -> reverse rate | staggered score x transposed
 * transition -> Gumbel arg-max.  coef fp32 [B]: Euler: dt*dsigma_b; Analytic: sigma(t)-sigma(t-dt); Denoise:
 * sigma(t).  u fp32 [M, Vu], Vu = V, except Denoise on the absorbing graph where Vu = V-1 (mask column dropped). */
int sedd_predictor_step(int graph, int step, const void* logscore, int dtype, const int64_t* x, const float* coef,
                        const float* u, int64_t* xnew, int64_t M, int64_t L, int V, sedd_stream_t stream);

/* The same update with the uniform noise generated in the kernel instead of read from memory — the reference draws it
 * with torch.rand_like inside sample_categorical (utils/catsample.py:12, called from sampling/sampling.py:79, :98,
 * :131), one more [M, Vu] fp32 tensor written and read per step.  Noise definition (independent of launch geometry):
 * u[row][j] = (w >> 8) * 2^-24 with w = word (j & 3) of Philox4x32-10(counter = {row_lo, row_hi, j >> 2, offset},
 * key = {seed_lo, seed_hi}); offset < 2^32 (the sampler passes its step index).  Results equal
 * sedd_predictor_step on that u bit for bit.  Row-per-lane path only (V <= 64, V * elem_size not a multiple of 32,
 * 16-byte aligned log-score), else SEDD_ERR_UNSUPPORTED_SHAPE.  HBM bytes M (V e + 16) instead of M (V e + 4 Vu + 16). */
int sedd_predictor_step_philox(int graph, int step, const void* logscore, int dtype, const int64_t* x, const float* coef,
                               uint64_t seed, uint64_t offset, int64_t* xnew, int64_t M, int64_t L, int V,
                               sedd_stream_t stream);

/* The uniforms sedd_predictor_step_philox draws, written out as u fp32 [M, Vu] (same definition): for shapes its
 * row-per-lane path does not take (V > 64, unaligned log-scores) a host gets the same stream by calling this and then
 * sedd_predictor_step; no counterpart in the reference (torch.rand_like, utils/catsample.py:12). */
int sedd_philox_uniform(uint64_t seed, uint64_t offset, float* u, int64_t M, int Vu, sedd_stream_t stream);

/* Stand-alone Graph methods (for Graph-API parity; the predictor kernel above fuses them).  All outputs fp32
 * [M, V].  graph_lib.py:218-221/:139-145 (transp_rate), :79-93 (reverse_rate), :264-269/:162-165
 * (staggered_score), :226-234/:147-154 (transp_transition). score fp32 [M,V]; sig fp32 [B]. */
int sedd_transp_rate(int graph, const int64_t* x, float* out, int64_t M, int V, sedd_stream_t stream);
int sedd_reverse_rate(int graph, const int64_t* x, const float* score, float* out, int64_t M, int V,
                      sedd_stream_t stream);
int sedd_staggered_score(int graph, const float* score, const float* dsigma, float* out, int64_t M, int64_t L,
                         int V, sedd_stream_t stream);
int sedd_transp_transition(int graph, const int64_t* x, const float* sigma, float* out, int64_t M, int64_t L,
                           int V, sedd_stream_t stream);

/* ------------------------------------------------------------------------------------------ model: pointwise */

/* SEDD.forward's pooled conditioners (model/transformer.py:289-301; BatchNorm1d modules :259-260):
 * out = mean([c, BN_e(e), BN_m(m)]) over the terms whose input pointer is non-NULL; c, e, m, out fp32 [B, C] (e / m
 * are the outputs of esm_proj / mol_proj, computed with sedd_gemm_nt).  training != 0: batch statistics (biased
 * variance), running statistics updated in place with `momentum` (unbiased variance), B >= 2 required as in torch;
 * training == 0: the running statistics.  *_mean / *_rstd [C] receive the statistics used (saved for the backward). */
int sedd_cond_combine_fwd(const float* c, const float* e, const float* e_w, const float* e_b, float* e_rmean,
                          float* e_rvar, float* e_mean, float* e_rstd, const float* m, const float* m_w,
                          const float* m_b, float* m_rmean, float* m_rvar, float* m_mean, float* m_rstd, float* out,
                          int training, float momentum, float eps, int B, int C, sedd_stream_t stream);
/* autograd backward of the above: dc = dout / n; de / dm = BatchNorm backward of dout / n (NULL: not needed);
 * *_dw / *_db [C] = BatchNorm weight / bias gradients (accumulate != 0: added to what the buffers hold). */
int sedd_cond_combine_bwd(const float* dout, const float* e, const float* e_w, const float* e_mean,
                          const float* e_rstd, float* de, float* e_dw, float* e_db, const float* m, const float* m_w,
                          const float* m_mean, const float* m_rstd, float* dm, float* m_dw, float* m_db, float* dc,
                          int training, int accumulate, int B, int C, sedd_stream_t stream);

/* model/transformer.py:70-89: out[b] = [cos(sigma_b f_k), sin(sigma_b f_k)], f_k = exp(-ln(1e4) k / half). */
int sedd_timestep_embedding(const float* sigma, float* out, int B, int dim, sedd_stream_t stream);
/* SiLU forward / backward (transformer.py:62-66, :285), fp32, n elements. dx = dy * silu'(x). */
int sedd_silu_fwd(const float* x, float* y, int64_t n, sedd_stream_t stream);
int sedd_silu_bwd(const float* x, const float* dy, float* dx, int64_t n, sedd_stream_t stream);

/* EmbeddingLayer (transformer.py:193-218): out[r] = table[idx[r]] (fp32 [V, D]). */
int sedd_embedding_fwd(const float* table, const int64_t* idx, float* out, int64_t M, int D, int V,
                       sedd_stream_t stream);
/* dtable[idx[r]] += dout[r] (dtable fp32 [V, D], accumulated into). */
int sedd_embedding_bwd(const float* dout, const int64_t* idx, float* dtable, float* ws, int64_t M, int D, int V,
                       sedd_stream_t stream);
/* ws: NULL, or fp32 scratch of sedd_embedding_bwd_ws_floats(M, D, V) floats: where that is non-zero (V <= 32, the protein
 * vocabulary) every 512-row block writes its partial table there and a fixed-order fold adds them to dtable
 * (bit-reproducible).  Shapes with V * M <= 2^22 (the cross model's tables) are summed by an owner-computes kernel in row
 * order, also bit-reproducible and without scratch; everything else adds with fp32 atomics. */
int64_t sedd_embedding_bwd_ws_floats(int64_t M, int D, int V);

/* LayerNorm (transformer.py:30-38, fp32, eps 1e-5, scale w, no bias) fused with modulate_fused
 * (fused_add_dropout_scale.py:33-34): y = LN(x)*w*(1+scale_b)+shift_b.  x fp32 [M, D]; shift/scale are rows of
 * the adaLN output: element (b, j) at shift[b*mod_stride + j] in `mod_dtype`; y in `out_dtype`;
 * mean/rstd fp32 [M] saved for backward.  shift == NULL (and scale == NULL) gives plain LN(x)*w (+ bias b if
 * non-NULL: the affine nn.LayerNorm of transformers_protlig_cross.py:114). */
int sedd_ln_modulate_fwd(const float* x, const float* w, const float* bias, const void* shift, const void* scale,
                         int mod_dtype, int64_t mod_stride, void* y, int out_dtype, float* mean, float* rstd,
                         int64_t M, int64_t L, int D, sedd_stream_t stream);
/* backward: dx (fp32 [M,D]) = dx_add + d(LN-modulate)/dx when dx_add != NULL (dx_add == dx accumulates in place;
 * a different buffer adds the incoming residual-stream gradient without a copy), else just the LN term.
 * Partial column sums go to caller workspace ws fp32 [3, nchunk, D] (nchunk = sedd_ln_bwd_chunks(M, L)), in the
 * order d(w), d(shift)/d(bias), d(scale); chunks never straddle a batch element so that
 * sedd_ln_modulate_bwd_reduce can fold them into dw[D] (+=), dshift[B, .] and dscale[B, .] (fp32, written with
 * stride dmod_stride). */
int64_t sedd_ln_bwd_chunks(int64_t M, int64_t L);
/* sedd_ln_modulate_bwd plus the backward of the gated residual that produced x (sedd_gate_residual_bwd on the finished
 * dx; transformer.py:185's bias_dropout_add_scale feeding :186's norm2) in one pass: da (dy's dtype) = dx * gate_b *
 * keep*inv_keep, ws_gate fp32 [nchunk, D] = partial sums of dgate (fold with sedd_gate_residual_bwd_reduce).
 * a has dy's dtype; gate has mod_dtype and row stride mod_stride; dropout by the (seed, p_drop) hash.  D <= 1024. */
int sedd_ln_modulate_bwd_gate(const void* dy, int dy_dtype, const float* x, const float* w, const void* scale,
                              int mod_dtype, int64_t mod_stride, const float* mean, const float* rstd, float* dx,
                              const float* dx_add, float* ws, const void* a, const void* gate, float inv_keep,
                              uint64_t seed, float p_drop, void* da, float* ws_gate, int64_t M, int64_t L, int D,
                              sedd_stream_t stream);
/* The same at a block BOUNDARY (transformer.py:188-189 feeding the next block's :166 norm1): the gated residual is the
 * previous block's MLP branch, so the column sums of da as stored — mlp[2]'s bias gradient, sedd_gate_residual_bwd's
 * want_colsum — ride along: ws_gate is fp32 [2, nchunk, D] (gate-gradient partials, then bias-gradient partials), folded
 * by sedd_gate_residual_bwd_reduce(ws_gate, dgate, stride, dbias, ...). */
int sedd_ln_modulate_bwd_gate_colsum(const void* dy, int dy_dtype, const float* x, const float* w, const void* scale,
                                     int mod_dtype, int64_t mod_stride, const float* mean, const float* rstd, float* dx,
                                     const float* dx_add, float* ws, const void* a, const void* gate, float inv_keep,
                                     uint64_t seed, float p_drop, void* da, float* ws_gate, int64_t M, int64_t L, int D,
                                     sedd_stream_t stream);
int sedd_ln_modulate_bwd(const void* dy, int dy_dtype, const float* x, const float* w, const void* scale,
                         int mod_dtype, int64_t mod_stride, const float* mean, const float* rstd, float* dx,
                         const float* dx_add, float* ws, int64_t M, int64_t L, int D, sedd_stream_t stream);
int sedd_ln_modulate_bwd_reduce(const float* ws, float* dw, float* dshift, float* dscale, int64_t dmod_stride,
                                int64_t M, int64_t L, int D, sedd_stream_t stream);

/* NO COUNTERPART IN THE REFERENCE — for callers that cannot allocate device memory (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc): sedd_ln_modulate_bwd followed by
 * sedd_ln_modulate_bwd_reduce in one call, with the [3, nchunk, D] partial-sum workspace in a file-scope device array
 * of the library (16 MiB per GPU, held until the process ends) instead of caller memory.  Same arguments and results as
 * that pair (dw [D] +=; dshift / dscale [B, .] written with row stride dmod_stride, either may be NULL).  Limit:
 * 3 * sedd_ln_bwd_chunks(M, L) * D <= SEDD_LN_BWD_NOALLOC_MAX_FLOATS, otherwise SEDD_ERR_UNSUPPORTED_SHAPE and nothing
 * is launched (e.g. D = 768: up to 1820 chunks = 116480 rows of 64).  Calls no hipMalloc / hipFree.  All calls on one
 * GPU share that GPU's array: calls that may run concurrently on one GPU (different streams) race — serialise them. */
#define SEDD_LN_BWD_NOALLOC_MAX_FLOATS (4 * 1024 * 1024)
int sedd_ln_modulate_bwd_noalloc(const void* dy, int dy_dtype, const float* x, const float* w, const void* scale,
                                 int mod_dtype, int64_t mod_stride, const float* mean, const float* rstd, float* dx,
                                 const float* dx_add, float* dw, float* dshift, float* dscale, int64_t dmod_stride,
                                 int64_t M, int64_t L, int D, sedd_stream_t stream);

/* bias_dropout_add_scale (fused_add_dropout_scale.py:13-48), bias == None at every reference call site:
 * out = residual + gate_b * (a * keep / (1-p)).  residual/out fp32 [M,D], a in `a_dtype`, gate rows as in
 * ln_modulate.  Dropout (p = model.dropout in training): either keep = uint8 [M,D] keep-mask supplied by the caller,
 * or keep == NULL and p_drop > 0: the mask is generated in the kernel from a counter-based hash of (seed, element
 * index), so the backward call with the same seed reproduces it and no mask is stored; keep == NULL and p_drop == 0:
 * inference.  inv_keep = 1/(1-p) in both dropout forms. 
 * A seed with bit 63 set is INDIRECT: bits 47..0 = device address of a uint64 salt, bits 62..48 = a site number; the
 * kernels then hash with salt + site * 0xD1B54A32D192ED03 — launch arguments frozen by a HIP-graph capture still give
 * fresh masks when the salt in device memory changes between replays. */
int sedd_gate_residual_fwd(const float* residual, const void* a, int a_dtype, const void* gate, int mod_dtype,
                           int64_t mod_stride, const uint8_t* keep, float inv_keep, uint64_t seed, float p_drop,
                           float* out, int64_t M, int64_t L, int D, sedd_stream_t stream);
/* backward: da = dout * gate_b * keep*inv_keep (a_dtype); per-chunk partial sums of dgate = sum_l dout*a*keep*inv
 * to ws fp32 [nchunk, D] (same chunking as sedd_ln_bwd_chunks), folded by sedd_gate_residual_bwd_reduce.
 * want_colsum != 0: ws is [2, nchunk, D] and its second plane receives the partial column sums of da (as stored),
 * which the reduce call adds into dbias fp32 [D] (+=): the bias gradient of the Linear that produced `a`
 * (torch.nn.functional.linear's backward for e.g. transformer.py:132), without a separate pass over da. */
int sedd_gate_residual_bwd(const float* dout, const void* a, int a_dtype, const void* gate, int mod_dtype,
                           int64_t mod_stride, const uint8_t* keep, float inv_keep, uint64_t seed, float p_drop,
                           void* da, float* ws, int want_colsum, int64_t M, int64_t L, int D, sedd_stream_t stream);
int sedd_gate_residual_bwd_reduce(const float* ws, float* dgate, int64_t dmod_stride, float* dbias, int64_t M,
                                  int64_t L, int D, sedd_stream_t stream);

/* sedd_gate_residual_fwd followed by the next sub-layer's sedd_ln_modulate_fwd in one pass over the row
 * (transformer.py:185-186: x = bias_dropout_add_scale(attn_out, None, gate_msa, x_skip, p); modulate(norm2(x), ...)):
 * x_out fp32 [M, D] = residual + gate_b * dropout(a);  y (dtype of a) = LN(x_out) * w * (1 + scale_b) + shift_b;
 * mean / rstd fp32 [M] saved for backward.  Dropout by the (seed, p_drop) hash only (p_drop == 0: none).  D <= 1024. */
int sedd_gate_residual_ln_modulate_fwd(const float* residual, const void* a, int a_dtype, const void* gate,
                                       const void* shift, const void* scale, int mod_dtype, int64_t mod_stride,
                                       float inv_keep, uint64_t seed, float p_drop, const float* w, float* x_out,
                                       void* y, float* mean, float* rstd, int64_t M, int64_t L, int D,
                                       sedd_stream_t stream);

/* Rotary (model/rotary.py:13-51): in-place NeoX rotation of the Q and K slots of qkv [B, L, 3, H, d]:
 * out = x*cos + rotate_half(x)*sin with rotate_half(x) = [-x2, x1]; conj != 0 applies the transposed rotation
 * (backward).  cos/sin fp32 [L, d/2] are the first halves of the reference's cached tables (rotary.py:17-22;
 * built once per L by the caller exactly like the reference builds them). d % 8 == 0. */
int sedd_rotary_qkv(void* qkv, int dtype, const float* cos, const float* sin, int B, int L, int H, int d, int conj,
                    sedd_stream_t stream);

/* GELU(tanh) forward/backward (transformer.py:138): g = gelu(u); du = dg * gelu'(u). */
int sedd_gelu_fwd(const void* u, void* g, int dtype, int64_t n, sedd_stream_t stream);
int sedd_gelu_bwd(const void* u, const void* dg, void* du, int dtype, int64_t n, sedd_stream_t stream);

/* Output epilogue of SEDD.forward (transformer.py:313-322): in place on logits [M, V]:
 * x -= log(E(sigma_b)) (cast to dtype) + log(V-1) when scale_by_sigma; x[r, idx[r]] = 0. */
int sedd_logits_epilogue(void* logits, int dtype, const float* sigma, const int64_t* idx, int scale_by_sigma,
                         int64_t M, int64_t L, int V, sedd_stream_t stream);
/* its backward: dlogits[r, idx[r]] = 0 in place. */
int sedd_logits_epilogue_bwd(void* dlogits, int dtype, const int64_t* idx, int64_t M, int V, sedd_stream_t stream);

/* Column sums: out[j] (+)= sum_r a[r, j]  (bias gradients); a [M, N] with row stride lda; out fp32 [N].  ws: NULL, or fp32
 * scratch of ceil(M / 128) * N floats — the 128-row blocks then write their partial sums there and a fixed-order fold adds
 * them to out (bit-reproducible); with NULL they add to out with fp32 atomics.  M <= 8192 with N >= 256 is one launch
 * that needs neither (a block per 8 columns sums all rows in a fixed order). */
int sedd_colsum(const void* a, int dtype, int64_t lda, float* out, float* ws, int accumulate, int64_t M, int N,
                sedd_stream_t stream);
/* out[N, M] = a[M, N]^T with optional dtype conversion (operand re-layout for the wgrad/dgrad GEMMs). */
int sedd_transpose(const void* a, int a_dtype, int64_t lda, void* out, int out_dtype, int64_t ldo, int64_t M,
                   int64_t N, sedd_stream_t stream);
/* elementwise cast fp32 <-> bf16, n elements. */
int sedd_cast(const void* in, int in_dtype, void* out, int out_dtype, int64_t n, sedd_stream_t stream);
/* x[i] *= scale[0] (scale: one fp32 on the device; product in fp32, rounded to x's dtype; x 16-byte aligned): the
 * upstream gradient of the scalar loss (losses.py:29 `.mean()` / the trainer's `loss.backward()`) applied to the
 * log-score gradient without a host read of the scalar. */
int sedd_scale_(void* x, int dtype, const float* scale, int64_t n, sedd_stream_t stream);

/* ------------------------------------------------------------------------------------------------- GEMM
 * nn.Linear (transformer.py:125-134,140,225,229; transformers_protlig_cross.py:250-268):
 * C[M,N] = A[M,K] * B[N,K]^T (+ bias[N]) (GELU-tanh), both operands K-contiguous ("NT").
 * in_dtype: SEDD_BF16 (v_mfma_f32_16x16x32_bf16) or SEDD_F32 (exact-fp32 v_mfma_f32_32x32x2_f32);
 * fp32 accumulation; C in out_dtype; accumulate != 0: C += (fp32 C only; weight-gradient accumulation).
 * bias fp32 or NULL.  For SEDD_EPI_BIAS_GELU, pre (may be NULL) receives the pre-activation (out_dtype); for
 * SEDD_EPI_BIAS_GELU_GRAD it receives gelu_tanh'(pre-activation) instead (same sigmoid as the activation: what autograd's
 * GELU backward multiplies by, handed to sedd_gemm_nn_dmul / sedd_gemm_nt_dmul).
 * Requirements: K % 8 == 0 (bf16) / K % 4 == 0 (fp32), lda/ldb multiples of the same, 16-byte aligned bases. */
int sedd_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int in_dtype,
                 int out_dtype, const float* bias, int epilogue, void* pre, int accumulate, int64_t M, int64_t N,
                 int64_t K, sedd_stream_t stream);
/* sedd_gemm_nt with a caller-owned workspace (16-byte aligned, may be NULL): accumulating fp32 outputs of small shapes
 * (one or a few output tiles, a long reduction: the adaLN conditioning gradient, the [V, D] output-layer weight
 * gradient; fp32 operands: any output of fewer than 128 tiles of 64 x 64, accumulating or not, bias allowed) split K
 * over blocks; with workspace_bytes >= 4 * slices * M * N (slices <= 64) the slices park their partial
 * tiles there and a fixed-order fold adds them to C — bit-reproducible run to run — instead of adding to C with fp32
 * atomics.  Everything else as sedd_gemm_nt. */
int sedd_gemm_nt_ws(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int in_dtype,
                    int out_dtype, const float* bias, int epilogue, void* pre, int accumulate, int64_t M, int64_t N,
                    int64_t K, void* workspace, int64_t workspace_bytes, sedd_stream_t stream);
/* Data gradient of Linear -> GELU(tanh) in one launch (autograd of the MLP nn.Sequential(Linear, GELU, Linear),
 * transformer.py:129-133): C[M,N] = (A[M,K] * B[N,K]^T) .* gelu_tanh'(pre[M,N]) with A = dY of the second Linear,
 * B = its weight transposed [in, out], pre = the saved pre-activation (row stride ldc, dtype of C); colsum (fp32 [N],
 * may be NULL) receives += the column sums of C, i.e. the bias gradient of the first Linear.  bf16 only, M >= 192,
 * N >= 96, N % 8 == 0, K % 32 == 0, 16-byte aligned C / pre rows; otherwise SEDD_ERR_UNSUPPORTED_SHAPE (callers then use
 * sedd_gemm_nt + sedd_gelu_bwd + sedd_colsum).  colsum_ws: NULL, or fp32 [ceil(M/256), N] scratch — each row tile then
 * writes its column sums to its own row and a fixed-order fold adds them to colsum (bit-reproducible run to run); with
 * NULL the row tiles add to colsum with fp32 atomics (sum order varies). */
int sedd_gemm_nt_dgelu(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int in_dtype,
                       int out_dtype, const void* pre, float* colsum, float* colsum_ws, int64_t M, int64_t N, int64_t K,
                       sedd_stream_t stream);
/* sedd_gemm_nt_dgelu with the derivative already evaluated: C = (A * B^T) .* mul[M,N] (mul = what sedd_gemm_nt's
 * SEDD_EPI_BIAS_GELU_GRAD epilogue stored in the forward pass); colsum / colsum_ws and requirements as above. */
int sedd_gemm_nt_dmul(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int in_dtype,
                      int out_dtype, const void* mul, float* colsum, float* colsum_ws, int64_t M, int64_t N, int64_t K,
                      sedd_stream_t stream);
/* The QKV projection with the rotary position embedding applied in the GEMM epilogue (transformer.py:165-171: attn_qkv,
 * rearrange 'b s (three h d) -> b s three h d', rotary.apply_rotary_pos_emb): C = packed qkv [M, 3*H*d] = A[M,K] *
 * B[N,K]^T (+ bias), then the NeoX rotation of sedd_rotary_qkv (conj = 0) on the q and k columns with position
 * row % L; cos / sin fp32 [L, d/2].  bf16 in/out, d == 64, N == 3*H*d, K % 32 == 0, M >= 192, M % L == 0, 16-byte
 * aligned C rows; otherwise SEDD_ERR_UNSUPPORTED_SHAPE (callers then use sedd_gemm_nt + sedd_rotary_qkv). */
int sedd_gemm_nt_rotary(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int in_dtype,
                        int out_dtype, const float* bias, int epilogue, const float* cos, const float* sin, int L, int H,
                        int d, int64_t M, int64_t N, int64_t K, sedd_stream_t stream);
/* Data gradient of nn.Linear against the weight AS STORED (dX = dY W with W [out_features, in_features] row-major; the
 * reference gets it from torch.nn.functional.linear's backward, e.g. for transformer.py:125-133's Linears):
 * C[M,N] = A[M,K] * B[K,N] (+ bias[N] for SEDD_EPI_BIAS) — "NN": B's reduction index is the row, so no transposed
 * weight copy is made.  bf16 in/out: M >= 192, N >= 96, K % 32 == 0, N % 8 == 0, lda/ldb multiples of 8; fp32 in/out
 * (the exact-fp32 MFMA kernel of sedd_gemm_nt reading B k-major): any M, K % 4 == 0, N % 4 == 0, lda/ldb multiples
 * of 4; 16-byte aligned bases; otherwise SEDD_ERR_UNSUPPORTED_SHAPE (callers then transpose + sedd_gemm_nt). */
int sedd_gemm_nn(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int in_dtype,
                 int out_dtype, const float* bias, int epilogue, int64_t M, int64_t N, int64_t K, sedd_stream_t stream);
/* sedd_gemm_nn with a caller-owned workspace (16-byte aligned, may be NULL), as sedd_gemm_nt_ws: an fp32 product whose
 * output has fewer than 128 tiles of 64 x 64 (the ligand track's data gradients: a few hundred rows against a long
 * reduction) splits K over blocks when workspace_bytes >= 4 * slices * M * N (slices <= 4), and a fixed-order fold adds
 * the slices (and the bias) — bit-reproducible.  bf16 products ignore the workspace. */
int sedd_gemm_nn_ws(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int in_dtype,
                    int out_dtype, const float* bias, int epilogue, int64_t M, int64_t N, int64_t K, void* workspace,
                    int64_t workspace_bytes, sedd_stream_t stream);
/* sedd_gemm_nt_dgelu with B given as [K, N] (the second MLP weight as stored, transformer.py:132):
 * C = (A[M,K] * B[K,N]) .* gelu_tanh'(pre[M,N]); colsum += column sums of C.  Same requirements as sedd_gemm_nn. */
int sedd_gemm_nn_dgelu(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int in_dtype,
                       int out_dtype, const void* pre, float* colsum, float* colsum_ws, int64_t M, int64_t N, int64_t K,
                       sedd_stream_t stream);
/* sedd_gemm_nt_dmul with B given as [K, N]: C = (A[M,K] * B[K,N]) .* mul[M,N]; colsum += column sums of C. */
int sedd_gemm_nn_dmul(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int in_dtype,
                      int out_dtype, const void* mul, float* colsum, float* colsum_ws, int64_t M, int64_t N, int64_t K,
                      sedd_stream_t stream);
/* CUs the persistent big-tile bf16 GEMM launches (sedd_gemm_nt / _nn / _tn and their fused forms) may count on: a
 * multiple of 8 in [8, 256], default 256 (or the environment's SEDD_GEMM_CUS).  Process-wide setting, not a per-call
 * argument — the one piece of mutable state in this library besides the no-alloc scratch; set it before launching, not
 * concurrently with launches.  No counterpart in the reference: a data-parallel trainer that overlaps RCCL collectives
 * with backward (run_train_ddp_protlig_transf.py:197's DDP buckets) leaves the collective's workgroups a few CUs so that
 * a one-block-per-CU GEMM launch still fits one round. */
int sedd_gemm_set_cu_budget(int cus);
int sedd_gemm_get_cu_budget(void);
/* Weight gradient of nn.Linear (autograd of the Linear layers above; the reference gets it from
 * torch.nn.functional.linear's backward): C[M,N] (+)= A[K,M]^T * B[K,N] with A = dY [tokens, out_features] and
 * B = X [tokens, in_features] row-major as the forward left them ("TN": the reduction index is the row), so no
 * transposed copies are made.  bf16 operands (fp32 operands: last paragraph), fp32 accumulation, C fp32 or bf16; accumulate != 0 adds into an
 * fp32 C (the flat gradient buffer).  An output with too few 256-wide tiles for the chip is split along K: with a
 * caller-owned `workspace` (16-byte aligned, workspace_bytes >= 256 MiB always suffices; exactly
 * slices * tiles * 256 * tile_width * 4 with slices * tiles <= 256) the slices park fp32 partial tiles there and a
 * second launch sums them into C; with workspace == NULL an accumulating call falls back to fp32 atomics.
 * Requirements: K % 32 == 0, M % 8 == 0, N % 8 == 0, lda % 8 == 0, ldb % 8 == 0, 16-byte aligned bases; otherwise
 * SEDD_ERR_UNSUPPORTED_SHAPE (callers fall back to sedd_transpose + sedd_gemm_nt).
 * fp32 operands (fp32 C): the exact-fp32 MFMA kernel reading both operands k-major as stored; any K, M % 4 == 0,
 * N % 4 == 0, lda % 4 == 0, ldb % 4 == 0; an output of fewer than 128 tiles of 64 x 64 is split along K when the
 * workspace holds slices * M * N floats (fixed-order fold, bit-reproducible), and not split otherwise. */
int sedd_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int in_dtype,
                 int out_dtype, int accumulate, int64_t M, int64_t N, int64_t K, void* workspace,
                 int64_t workspace_bytes, sedd_stream_t stream);
/* The same weight gradient with the Linear's bias gradient out of the same launch (torch.nn.functional.linear's backward
 * yields both; transformer.py:125-134, transformers_protlig_cross.py:250-268 — every nn.Linear there has a bias):
 * colsum[M] (+)= sum over the tokens k of A[k, m], fp32, accumulated with C's `accumulate` flag.  bf16 operands, fp32 C
 * only.  The sums come from one extra MFMA per 16-row fragment and K tile against a selector operand, dealt over the
 * tile columns, in a fixed order (bit-reproducible).  With a split reduction the workspace needs
 * slices * 256 * ceil(M / 256) * 4 bytes beyond sedd_gemm_tn's.  A reduction of at most 64 K tiles (2048 tokens) is not
 * split: 128-wide tiles, one block per tile adding into C directly (one launch instead of three for dW and db). */
int sedd_gemm_tn_colsum(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, void* colsum,
                        int in_dtype, int accumulate, int64_t M, int64_t N, int64_t K, void* workspace,
                        int64_t workspace_bytes, sedd_stream_t stream);

/* ------------------------------------------------------------------------------------------ attention
 * The flash-attn boundary (transformer.py:172-183: flash_attn_varlen_qkvpacked_func, non-causal, no dropout,
 * softmax scale d^-1/2, equal-length sequences) and nn.MultiheadAttention's core
 * (transformers_protlig_cross.py:135-138, :305) as ONE strided kernel family: element (b, l, h, :) of q at
 * q + b*q_bs + l*q_ls + h*d, likewise k, v (k/v share strides), o at o + b*o_bs + l*o_ls + h*d (head dim
 * contiguous).  key_mask: optional uint8 [B, Lk], 1 = attend (the inverse of key_padding_mask).
 * lse fp32 [B, H, Lq] is written for backward.  d in {32, 64, 96, 128}. */
int sedd_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const uint8_t* key_mask,
                  int dtype, int B, int H, int Lq, int Lk, int d, int64_t q_bs, int64_t q_ls, int64_t kv_bs,
                  int64_t kv_ls, int64_t o_bs, int64_t o_ls, sedd_stream_t stream);
/* backward: dq/dk/dv written with the strides of q/k/v; delta fp32 [B, H, Lq] is caller workspace.
 * rot_cos / rot_sin (both NULL or both fp32 [L, d/2], Lq == Lk == L): additionally apply the gradient of
 * apply_rotary_pos_emb (rotary.py:42-51; the conjugate rotation) to dq and dk as they are written, i.e. the result is
 * the gradient w.r.t. the PRE-rotary q and k — what sedd_rotary_qkv(conj = 1) would make of the plain output. */
int sedd_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                  const uint8_t* key_mask, void* dq, void* dk, void* dv, float* delta, int dtype, int B, int H, int Lq,
                  int Lk, int d, int64_t q_bs, int64_t q_ls, int64_t kv_bs, int64_t kv_ls, int64_t o_bs, int64_t o_ls,
                  const float* rot_cos, const float* rot_sin, sedd_stream_t stream);

/* flash_attn_varlen_qkvpacked_func in its general, variable-length form (transformer.py:173-181 with `seqlens` given;
 * spec: the vendored model/flash_attn_interface.py:1212-1275): the token axis of q / k / v / o ([total, H, d] views, token
 * strides x_ls in elements) is cut into nseq segments [cu_seqlens[s], cu_seqlens[s+1]) (int32, nseq + 1 entries on the
 * device, starting at 0) and attention runs inside each segment only (non-causal, scale d^-1/2, no dropout).
 * max_len: upper bound of the segment lengths (the reference passes the padded row length); longer segments are cut.
 * lse / delta fp32 [H, total].  Tokens outside every segment are not written.  Rotary is NOT fused here (positions are
 * row positions, not segment positions: apply sedd_rotary_qkv before / its conjugate after). */
int sedd_attn_varlen_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const int32_t* cu_seqlens,
                         int nseq, int total, int max_len, int dtype, int H, int d, int64_t q_ls, int64_t kv_ls,
                         int64_t o_ls, sedd_stream_t stream);
int sedd_attn_varlen_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                         void* dq, void* dk, void* dv, float* delta, const int32_t* cu_seqlens, int nseq, int total,
                         int max_len, int dtype, int H, int d, int64_t q_ls, int64_t kv_ls, int64_t o_ls,
                         sedd_stream_t stream);

/* ------------------------------------------------------------------------------------- optimiser / EMA
 * One pass over the flat fp32 parameter buffer: gradient clip coefficient (losses.py:65-66) -> AdamW
 * (losses.py:39-41; torch.optim.AdamW) -> EMA shadow update (model/ema.py:32-52) -> bf16 compute copy.
 * clip_coef: device scalar fp32 (min(1, max_norm/(norm+1e-6))), grad_scale multiplies g first (1/world etc.).
 * ema/p_bf16 may be NULL.  step is 1-based.  decoupled != 0: AdamW (p *= 1 - lr*weight_decay); 0: torch.optim.Adam's
 * L2 form, g += weight_decay * p before the moments (losses.py:36-38 vs :39-41). */
int sedd_adamw_ema_step(float* p, const float* g, float* m, float* v, float* ema, void* p_bf16,
                        const float* clip_coef, float grad_scale, float lr, float beta1, float beta2, float eps,
                        float weight_decay, int decoupled, int step, float ema_decay, int64_t n, sedd_stream_t stream);
/* The dual-track trainer's iteration advances the optimiser TWICE on the same clipped gradients
 * (training/run_train_protlig_cross_att.py:247 `scaler.step(optimizer)` inside optimize_fn, then :325
 * `optimizer.step()`), the EMA update (:248) falling between the two.  One pass over the buffers doing both:
 * bit-identical to sedd_adamw_ema_step(step, ema_decay) followed by sedd_adamw_ema_step(step + 1, ema_decay = 1). */
int sedd_adamw_ema_double_step(float* p, const float* g, float* m, float* v, float* ema, void* p_bf16,
                               const float* clip_coef, float grad_scale, float lr, float beta1, float beta2, float eps,
                               float weight_decay, int decoupled, int step, float ema_decay, int64_t n,
                               sedd_stream_t stream);
/* sedd_adamw_ema_step / _double_step (steps = 1 | 2) with the per-step scalars read from DEVICE memory: hyper = six floats
 * {1 - lr*wd (decoupled) or 1, lr / bias_corr1, 1 / sqrt(bias_corr2), 1 - ema_decay, and the second step's lr / bias_corr1,
 * 1 / sqrt(bias_corr2)}, as sedd_adamw_step_scalars computes them.  No counterpart in the reference: for callers that
 * capture a whole train step (losses.py:80-115) in a HIP graph, where launch arguments are frozen at capture. */
int sedd_adamw_ema_step_dev(float* p, const float* g, float* m, float* v, float* ema, void* p_bf16,
                            const float* clip_coef, const float* hyper, float grad_scale, float beta1, float beta2,
                            float eps, float weight_decay, int decoupled, int steps, int64_t n, sedd_stream_t stream);
/* writes hyper_dev[0..5] = v0..v5 and *salt_dev = salt (either may be NULL) from a one-thread launch: the values travel as
 * launch arguments, so the per-step scalars of a captured step reach device memory in stream order without a host buffer
 * that has to outlive the call.  salt_dev: the uint64 an INDIRECT dropout seed points at (sedd_gate_residual_fwd). */
int sedd_write_step_scalars(float* hyper_dev, uint64_t* salt_dev, float v0, float v1, float v2, float v3, float v4,
                            float v5, uint64_t salt, sedd_stream_t stream);
/* host arithmetic only: the six scalars above for optimiser step `step` (1-based), written to HOST memory out6_host. */
int sedd_adamw_step_scalars(float lr, float beta1, float beta2, float weight_decay, int decoupled, int step,
                            float ema_decay, float* out6_host);
/* out[0] = sum_r w[r / L] * x[r]  (losses.py:29: (dsigma[:, None] * loss).mean() with w_b = dsigma_b / M).
 * fp32, deterministic two-stage; ws fp32 [1024] caller workspace. */
int sedd_batch_weighted_sum(const float* x, const float* w, int64_t M, int64_t L, float* ws, float* out,
                            sedd_stream_t stream);
/* model/ema.py:44-52 stand-alone: shadow -= one_minus_decay * (shadow - p), fp32, n elements. */
int sedd_ema_update(float* shadow, const float* p, float one_minus_decay, int64_t n, sedd_stream_t stream);
/* out[0] = sum_i g[i]^2 (fp32, deterministic two-stage); ws fp32 [1024] caller workspace.
 * Then, when max_norm >= 0: out[1] = min(1, max_norm / (sqrt(out[0]) * gscale + 1e-6)) (max_norm == 0 zeroes the
 * gradients, as clip_grad_norm_ does); max_norm < 0: no clipping, out[1] is left untouched. */
int sedd_sumsq_clip(const float* g, int64_t n, float gscale, float max_norm, float* ws, float* out,
                    sedd_stream_t stream);
/* the last step of sedd_sumsq_clip on its own: out[1] = min(1, max_norm / (sqrt(out[0]) * gscale + 1e-6)) from a sum of
 * squares that is already in out[0] — the sharded optimiser (reduce-scatter -> optimiser on the 1/N shard -> all-gather)
 * all-reduces the per-shard sums into out[0] first; clip_grad_norm_ of losses.py:65-66 over the WHOLE model. */
int sedd_clip_from_sumsq(float* out, float gscale, float max_norm, sedd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SEDD_HIP_H */
