/* mdlm_hip.h -- C-ABI of libmdlm_hip.so: MI355X (gfx950) kernels for the MDLM
 * denoise-and-sample hot path.
 *
 * Every entry point takes raw DEVICE pointers + sizes + a hipStream_t (passed
 * as void*), launches asynchronously on that stream, allocates nothing, keeps
 * no pointer after return and returns int: 0 = OK, <0 = argument error
 * (message via mdlm_last_error_string(), thread-local), >0 = hipError_t.
 * Token ids are int64, activations fp32 or bf16 (MDLM_F32 / MDLM_BF16), all
 * matrices row-major with the innermost dimension contiguous and an explicit
 * leading dimension `ld` (in ELEMENTS) where noted.
 *
 * Each declaration cites the reference code (kuleshov-group/mdlm) it replaces.
 */
#ifndef MDLM_HIP_H
#define MDLM_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDLM_F32 0
#define MDLM_BF16 1

/* GEMM epilogues (mdlm_gemm_bf16) */
#define MDLM_EPI_STORE 0          /* C = acc (+bias)                       */
#define MDLM_EPI_BIAS_GELU 1      /* C = gelu_tanh(acc + bias)             */
#define MDLM_EPI_GATE_RESIDUAL 2  /* C = residual + gate * (acc (+bias))   */
#define MDLM_EPI_QKV_ROTARY 3     /* q,k rotary -> head-major; v -> v^T    */

int mdlm_version(void);
const char* mdlm_last_error_string(void);

/* ---------------------------------------------------------------- sampler */

/* (a) Diffusion._subs_parameterization, diffusion.py:261-277.
 * logits (rows, V) with leading dim ld_in -> log-probs (rows, V), ld_out.
 * Masked rows (xt == mask_index): logit - logsumexp with the mask column
 * shifted by -1e6; other rows: -1e6 everywhere, 0 at xt.  `logits` is not
 * modified (the reference mutates its mask column in place; callers that
 * rely on that see the same values through the returned tensor).
 * as_probs != 0 writes exp(log-prob) instead (the `.exp()` of
 * diffusion.py:602 fused in: p_x0 for _ddpm_caching_update). */
int mdlm_subs_logprobs(const void* logits, int64_t ld_in, int dtype_in,
                       const int64_t* xt, void* out, int64_t ld_out,
                       int dtype_out, int64_t rows, int V, int mask_index,
                       int as_probs, void* stream);

/* Row log-sum-exp of the SUBS-shifted logits (diffusion.py:263-268) for the
 * masked rows only; rows with xt != mask get lse = 0 and are not read.
 * sample_map/L/n_rows_dev (optional, may be NULL/any/NULL): visit only the
 * rows of the listed samples -- launch row r < *n_rows_dev handles row
 * sample_map[r / L] * L + r % L of logits, xt and lse. */
int mdlm_subs_row_lse(const void* logits, int64_t ld, int dtype,
                      const int64_t* xt, float* lse, int64_t rows, int V,
                      int mask_index, const int32_t* sample_map, int L,
                      const int32_t* n_rows_dev, void* stream);

/* (a') noise removal: argmax_v SUBS(logits), diffusion.py:690-697 / 982.
 * Unmasked rows return xt; masked rows the best non-mask column (first index
 * on ties). */
int mdlm_subs_argmax(const void* logits, int64_t ld, int dtype,
                     const int64_t* xt, int64_t* out, int64_t rows, int V,
                     int mask_index, void* stream);

/* (b) Diffusion._ddpm_caching_update with a cached p_x0, diffusion.py:592-610
 * + _sample_categorical, diffusion.py:24-28.
 *   q = p_x0 * (t - (t - dt)); q[mask] = t - dt (fp32, in that order);
 *   x_s = argmax_v q_v / (1e-10 - log(u_v + 1e-10)); carry-over of unmasked.
 * p_x0 (B*L, V) probabilities, ld elements per row.  t (B) fp32.  u: (B*L, V)
 * contiguous fp32 uniforms, or NULL to draw in-kernel (Philox4x32-7 keyed by
 * seed; element (row, v) uses counter offset + row*ceil(V/4) + v/4, lane v%4 --
 * the same stream mdlm_philox_uniform writes).  changed (B) int32, may be
 * NULL: set to 1 when any token of the sample changed, else 0 (replaces the
 * host-side torch.allclose of diffusion.py:682).
 * The DEFINING arithmetic of the race (what token ids are bit-exact to):
 *   g = 1e-10f - RN(ln(u + 1e-10f)) (correctly rounded fp32 log), key = q / g
 *   (IEEE fp32 division), argmax with torch semantics (NaN wins, first index on
 *   ties).
 * fast_math == 0 (or u supplied): every element in that arithmetic.
 * fast_math != 0 with in-kernel uniforms: SAME token ids at lower cost -- the
 *   columns are ranked by an approximate key (hardware log2 / rcp); a row whose
 *   best and runner-up approximate keys are closer than a proven bound on the
 *   error of the two arithmetics (or whose values leave the analysed range) is
 *   re-evaluated in the defining arithmetic.  n_exact_rows (optional device
 *   uint64): += number of rows that took the re-evaluation (diagnostic). */
int mdlm_ddpm_cache_update(const void* p_x0, int64_t ld, int dtype,
                           const int64_t* x, const float* t, double dt,
                           const float* u, uint64_t seed, uint64_t offset,
                           int64_t* x_next, int32_t* changed, int B, int L,
                           int V, int mask_index, int fast_math,
                           uint64_t* n_exact_rows, void* stream);

/* (a)+(b) fused from cached LOGITS: p_x0 = exp(logit - lse) is formed in
 * registers (diffusion.py:602 `.exp()` + 261-277), never written.  lse (B*L)
 * as from mdlm_subs_row_lse.  fresh (B) int32, optional: samples flagged
 * non-zero have new logits -- their masked rows' lse is recomputed in-kernel
 * (and stored to lse) before the race, so no separate lse pass is needed.
 * Same sampling semantics as above.
 * Precondition of fast_math != 0 from logits: the logits of masked rows are
 *   finite or -inf.  A NaN logit makes every lse this library computes (in-kernel
 *   `fresh` pass, mdlm_subs_row_lse, the vocabulary GEMM's epilogue) NaN, which
 *   routes the row to the defining arithmetic (torch semantics: NaN wins); with a
 *   CALLER-computed finite lse over a row that holds a NaN the approximate pass
 *   drops the NaN column instead (its prefilter and column skip take maxima).
 *   fast_math == 0 has no precondition. */
int mdlm_subs_ddpm_update(const void* logits, int64_t ld, int dtype,
                          float* lse, const int32_t* fresh,
                          const int64_t* x, const float* t,
                          double dt, const float* u, uint64_t seed,
                          uint64_t offset, int64_t* x_next, int32_t* changed,
                          int B, int L, int V, int mask_index, int fast_math,
                          uint64_t* n_exact_rows, void* stream);

/* Generic masked-row categorical race from cached LOGITS, for the other two
 * samplers of the reference.  A masked position draws
 *   argmax_v q_v / g_v,  q_v = exp(logit_v - lse) * q_scale[b] (v != mask),
 *                         q_mask = q_mask[b];
 * unmasked positions keep their token.  q_scale, q_mask: (B) fp32, computed by
 * the caller from the noise schedule:
 *   Diffusion._ddpm_update      (diffusion.py:612-637): q_scale = mc_t - mc_s,
 *                                q_mask = mc_s  (mc = 1 - exp(-sigma));
 *   Diffusion._analytic_update  (diffusion.py:773-780 with get_score 721-764,
 *     _staggered_score 766-771, _transp_transition 791-798), in closed form for
 *     SUBS outputs: q_scale = k e^{ds} (1 - e^{-ds}), q_mask = e^{ds} +
 *     (1 - e^{ds})(k + 1), k = 1/expm1(sigma); an unmasked position has a
 *     single non-zero probability (its own token);
 *   Diffusion._denoiser_update  (diffusion.py:782-789): same with ds = sigma
 *     and q_mask = 0.
 * Other arguments as mdlm_subs_ddpm_update. */
int mdlm_subs_race_update(const void* logits, int64_t ld, int dtype,
                          float* lse, const int32_t* fresh,
                          const int64_t* x, const float* q_scale,
                          const float* q_mask, const float* u, uint64_t seed,
                          uint64_t offset, int64_t* x_next, int32_t* changed,
                          int B, int L, int V, int mask_index, int fast_math,
                          uint64_t* n_exact_rows, void* stream);

/* Diagnostic for the error budget of the fast_math race (tests only; no
 * reference counterpart): for n uniform bit patterns rbits[i] (u = (rbits >> 8)
 * * 2^-24, as the kernels form it) writes the approximate pass's g~ and
 * log2(g~) and g_exact[i] = the defining g.  The exhaustive test over all 2^24
 * uniforms pins the bounds quoted in csrc/sampler.hip. */
int mdlm_race_key_probe(const uint32_t* rbits, float* g_fast, float* log2_g_fast,
                        float* g_exact, int64_t n, void* stream);

/* Discrete-time (T > 0) SUBS loss term, Diffusion._d3pm_loss,
 * diffusion.py:329-358, given neg_logp_x0 (B, L) = -log p_theta(x0 | xt) (the
 * output of mdlm_subs_nll with weight 1, no attention mask), xt (B, L), the
 * discretised t (B) of diffusion.py:849-853 and T; attn_mask (B, L) fp32 or
 * NULL multiplies the result (diffusion.py:908).  out (B, L) fp32.  The mask
 * column of a SUBS output has probability exactly 0, so no second gather is
 * needed; samples with t == 1/T yield NaN exactly as the reference does
 * (0/0 at line 350). */
int mdlm_d3pm_loss(const float* neg_logp_x0, const int64_t* xt, const float* t,
                   int T, const float* attn_mask, float* out, int B, int L,
                   int mask_index, void* stream);

/* ---- opt-in two-stage ("lazy") ddpm_cache step; no reference counterpart ----
 * The race of diffusion.py:592-610 leaves a masked position masked with
 * probability (t - dt)/t and otherwise draws its token from p_x0.  Stage 1
 * (mdlm_lazy_select) makes that Bernoulli decision with one Philox uniform per
 * position: x_next = x, rows[0..n) = positions that unmask (padded with row 0
 * up to a multiple of `pad`), n_rows[0] = padded count, n_rows[1] = n,
 * changed[b] = 1 if sample b unmasks anything.  Stage 2 (mdlm_lazy_race) draws
 * x_next[rows[i]] from softmax(logits row i) over v != mask; logits is the dense
 * (max_rows, ld) output of the vocabulary projection of those rows only.
 * Distribution-identical to the reference step, NOT bit-identical to it on a
 * shared uniform tensor (a different map from random numbers to outcomes):
 * `sampling.lazy_unmask=true` only. */
int mdlm_lazy_select(const int64_t* x, const float* t, double dt, uint64_t seed,
                     uint64_t offset, int B, int L, int mask_index, int pad,
                     int64_t* x_next, int32_t* rows, int32_t* n_rows,
                     int32_t* changed, void* stream);
int mdlm_lazy_race(const void* logits, int64_t ld, int dtype, const int32_t* rows,
                   const int32_t* n_rows, int max_rows, uint64_t seed,
                   uint64_t offset, int64_t* x_next, int V, int mask_index,
                   void* stream);
/* x_dst[i] = x_src[gather[i]] (fp32 rows) and a_dst[i] = a_src[gather[i]] (bf16
 * rows) for i < *n_rows_dev, D elements each.  Only the masked positions' logits
 * are ever read (diffusion.py:270-277, 609-610), so after the LAST block's
 * attention (dit.py:265-275) everything that remains -- out-projection, MLP,
 * final layer (dit.py:277-288, 317-321) -- is row-wise and runs on those rows
 * alone, packed densely by this gather. */
int mdlm_gather_rows(const float* x_src, float* x_dst, const void* a_src,
                     void* a_dst, const int32_t* gather, int D, int64_t n_rows,
                     const int32_t* n_rows_dev, void* stream);

/* dst[sample_map[r / L] * L + r % L] = src[r] for r < *n_rows_dev (fp32 rows of
 * D elements): writes the residual stream of the compacted active samples back
 * to a per-sample hidden-state cache (lazy sampler; SURVEY 8(f)-1 "cache the
 * (B, L, 768) final hidden state instead of (B, L, V) p_x0"). */
int mdlm_scatter_rows_f32(const float* src, float* dst, const int32_t* sample_map,
                          int L, int D, int64_t n_rows, const int32_t* n_rows_dev,
                          void* stream);

/* The uniforms the in-kernel generator uses, materialised (rows, V) fp32, so
 * that tests can feed the oracle the same draw.  Test/diagnostic helper; no
 * reference counterpart (the reference calls torch.rand_like, diffusion.py:27). */
int mdlm_philox_uniform(float* out, int64_t rows, int V, uint64_t seed,
                        uint64_t offset, void* stream);

/* (c1) Diffusion.q_xt, diffusion.py:575-586: xt = mask where u < move_chance.
 * move_chance (B) fp32; u (B, L) fp32 or NULL for in-kernel Philox (element i
 * uses counter offset + i/4, lane i%4). */
int mdlm_q_xt(const int64_t* x0, const float* move_chance, const float* u,
              uint64_t seed, uint64_t offset, int64_t* xt, int B, int L,
              int mask_index, void* stream);

/* (c2) SUBS + gather + NELBO weight, diffusion.py:884-894 and 908:
 *   nll[b,l] = -log p_theta(x0[b,l] | xt) * weight[b] * attn_mask[b,l]
 * with weight = dsigma/expm1(sigma) supplied per sample; log-probs are never
 * materialised.  attn_mask may be NULL (= ones).  row_lse (B*L, optional): the
 * SUBS row lse of the masked rows if the caller already has it (mdlm_lse_combine
 * after the vocabulary GEMM) -- the kernel then reads one logit per row instead
 * of the row. */
int mdlm_subs_nll(const void* logits, int64_t ld, int dtype, const int64_t* xt,
                  const int64_t* x0, const float* weight,
                  const float* attn_mask, const float* row_lse, float* nll,
                  int B, int L, int V, int mask_index, void* stream);

/* fp64 metric sums, diffusion.py:44-65,369-376 (MeanMetric in float64):
 * sums[0] += sum(nll), sums[1] += sum(attn_mask) (or n if NULL).  One
 * workgroup, fixed order: deterministic. */
int mdlm_nll_sums(const float* nll, const float* attn_mask, double* sums,
                  int64_t n, void* stream);

/* ------------------------------------------------------------ DiT backbone */

/* EmbeddingLayer.forward, dit.py:298-299: out[r,:] = table[idx[src(r)],:] (fp32).
 * sample_map (int32) may be NULL (src(r) = r); otherwise output row r belongs
 * to compacted sample r / L and reads idx[sample_map[r / L] * L + r % L]
 * (gathers the rows of the active samples).  n_rows_dev, if not NULL, is a
 * device int32 holding how many of n_rows are live. */
int mdlm_embed_rows(const float* table, const int64_t* idx,
                    const int32_t* sample_map, int L, float* out,
                    int64_t n_rows, int D, int vocab,
                    const int32_t* n_rows_dev, void* stream);

/* TimestepEmbedder.timestep_embedding, dit.py:162-184: (B) -> (B, dim). */
int mdlm_timestep_embedding(const float* t, float* out, int B, int dim,
                            void* stream);

/* Small fp32 linear for the conditioning path (sigma_map MLP dit.py:186-189,
 * adaLN_modulation dit.py:249-250, 318): out[b,n] = act_out(sum_k
 * act_in(in[b,k]) * W[n,k] + bias[n]); act: 0 none, 1 SiLU. */
int mdlm_linear_f32(const float* in, const float* W, const float* bias,
                    float* out, int B, int N, int K, int act_in, int act_out,
                    void* stream);

/* LayerNorm (no affine, eps 1e-5, fp32) * weight, then modulate_fused,
 * dit.py:126-134 + 74-78: out = LN(x)*w*(1+scale)+shift -> bf16.
 * shift/scale: n_mod rows of D fp32, mod_ld elements apart; row r uses
 * modulation row (r / rows_per_mod) % n_mod (n_mod == 1: batch-invariant
 * conditioning, as with time_conditioning=False).  row_gather (optional):
 * output row r is computed from input row row_gather[r] (row compaction). */
int mdlm_ln_modulate(const float* x, const float* ln_w, const float* shift,
                     const float* scale, void* out_bf16, int64_t rows, int D,
                     int64_t rows_per_mod, int n_mod, int64_t mod_ld,
                     const int32_t* row_gather, const int32_t* n_rows_dev,
                     void* stream);

/* nn.Linear on bf16 MFMA, dit.py:220-228, 256, 277, 306/320:
 * C[M,N] = epilogue(A[M,K] @ W[N,K]^T).  A bf16 (lda), W bf16 contiguous.
 * Requires K % 64 == 0, M % 128 == 0.  bias (N) fp32 or NULL.
 *   STORE          : C (ldc, dtype_c) = acc + bias.  out_sample_map (optional):
 *                    output row r is written to row out_sample_map[r / L] * L
 *                    + r % L with L = rows_per_mod (scatter of compacted samples
 *                    back into a per-sample logits cache).  out_row_map
 *                    (optional, exclusive with out_sample_map): output row r
 *                    is written to row out_row_map[r], or dropped if < 0
 *                    (logits of the masked rows only, see mdlm_masked_row_map).
 *   BIAS_GELU      : C bf16 = gelu_tanh(acc + bias)            (dit.py:225-228)
 *   GATE_RESIDUAL  : C fp32 = residual + gate[(row/rows_per_mod) % n_mod] *
 *                    (acc + bias); gate rows mod_ld elements apart
 *                    (bias_dropout_add_scale, eval mode, dit.py:20-34)
 *   QKV_ROTARY     : N = 3*H*64; q,k get the NeoX rotary (dit.py:112-115) with
 *                    rot_cos/rot_sin (L,32) fp32 and are written head-major
 *                    q[b,h,l,64], k[b,h,l,64]; v is written transposed
 *                    vt[b,h,64,l]; C = base of a buffer of 3*M*H*64 bf16
 *                    (q | k | vt).  rows_per_mod = L.
 * m_rows_dev: optional device int32 with the number of live rows (tiles
 * beyond it exit), so a fixed launch can serve a device-side active count.
 * lse_partials_out (optional; bf16 STORE with M % 256 == 0 only): SUBS row
 * statistics out of the epilogue (diffusion.py:263-268 fused
 * behind dit.py:317-321).  For every output row and every group of 64 columns
 * the pair (max, sum exp(y - max)) over the bf16-rounded outputs y, column
 * lse_skip_col (the mask token: its SUBS logit is shifted by -1e6, so it adds
 * exp(.) = 0; -1 = none) excluded; layout [row / 256][ceil(N/64)][row % 256]
 * float2, mdlm_gemm_lse_partials_workspace_bytes(M, N) bytes, indexed by the
 * GEMM's own (compact) row.  mdlm_lse_combine turns them into the row lse.
 * tile: 0 = choose (256x256 when M % 256 == 0 and N >= 256, else 128x128),
 * 1 = 128x128 always -- per call, so both configurations can be parity-tested on
 * any shape; the library holds no tuning state.  | 16: keep whole tiles in a short
 * last round (default: such tiles are cut into two half tiles of 128 rows, which
 * finishes a lone launch sooner but costs more CU time -- callers that overlap
 * launches on several streams pass 16).  | 32: the two-accumulator-set kernel
 * (128x256 tiles, the finished tile leaves from registers inside the next tile's
 * K loop) where it applies -- gate/residual fp32 with N % 256 == 0, N <= 2048, one
 * gate row, and bias + GELU with N <= 4096, K >= 512 -- else the default kernel;
 * measured not faster, kept opt-in.  Results do not depend on any of
 * these bits: every configuration runs the same K order per output element, so a
 * caller may pick the tile by launch size (mdlm_amd.dit.DIT.run gives launches of
 * a few samples the 128x128 tiles). */
int mdlm_gemm_bf16(const void* A, int64_t lda, const void* W, const float* bias,
                   void* C, int64_t ldc, int dtype_c, int64_t M, int N, int K,
                   int epilogue, const float* gate, const float* residual,
                   int64_t rows_per_mod, int n_mod, int64_t mod_ld,
                   const float* rot_cos, const float* rot_sin, int n_heads,
                   const int32_t* m_rows_dev, const int32_t* out_sample_map,
                   const int32_t* out_row_map, float* lse_partials_out,
                   int lse_skip_col, int tile, void* stream);

/* lse[out_row_map[r]] (or lse[r] when the map is NULL; entries < 0 skipped) =
 * log-sum-exp of row r from the partials mdlm_gemm_bf16 wrote, for r <
 * *n_rows_dev (NULL: r < M): the SUBS normaliser of diffusion.py:265-268
 * without a pass over the (rows, V) logits. */
int mdlm_lse_combine(const float* partials, int64_t M, int N,
                     const int32_t* out_row_map, const int32_t* n_rows_dev,
                     float* lse, void* stream);

/* Weight gradient of an nn.Linear in the training step (autograd of
 * dit.py:220-232, 256, 277, 306/320 under diffusion.py:390-397): the
 * contraction runs over the token dimension K_total and the output is small,
 * so K is cut into `splits` ranges that run as independent tiles:
 *   partials[s][M][N] = A[:, s*Kc:(s+1)*Kc] @ W[:, s*Kc:(s+1)*Kc]^T   (fp32),
 * Kc = K_total / splits, K_total % (splits*64) == 0, M % 128 == 0, N % 4 == 0;
 * A (M, K_total) ld lda, W (N, K_total) ld ldw, both bf16.
 * mdlm_splitk_reduce then writes out[i] = sum_s partials[s*split_stride + i]
 * for i < n, added in split order (deterministic). */
int mdlm_gemm_bf16_splitk(const void* A, int64_t lda, const void* W, int64_t ldw,
                          float* partials, int64_t M, int N, int64_t K_total,
                          int splits, void* stream);
int mdlm_splitk_reduce(const float* partials, float* out, int64_t n, int splits,
                       int64_t split_stride, void* stream);

/* GELU (dit.py:227, approximate='tanh') fused into the MLP GEMMs of the
 * training step; C and aux are bf16 (M, N) with leading dimension ldc:
 *   mode 0: aux = A W^T + bias (the pre-activation autograd saves),
 *           C   = gelu_tanh(A W^T + bias)               -- forward of mlp.0
 *   mode 1: C   = bf16(A W^T) * gelu_tanh'(aux)         -- d(pre) from d(act) = dY W2,
 *           the backward of the GELU applied in the epilogue of the dX GEMM of mlp.2
 * M % 128 == 0, K % 64 == 0, ldc % 8 == 0. */
int mdlm_gemm_bf16_gelu(const void* A, int64_t lda, const void* W, const float* bias,
                        void* C, void* aux, int64_t ldc, int64_t M, int N, int K,
                        int mode, void* stream);

/* The same weight gradient on TOKEN-MAJOR operands, as the forward / backward
 * kernels leave them (no transposed copies): dY (T, N) ld ldy, X (T, K) ld ldx,
 * both bf16;  partials[s][N][K] = sum over the tokens of range s of
 * dY[t, n] * X[t, k]  (fp32; ranges of ceil(T/64/splits) 64-token slabs, the
 * last may be shorter).  bias_partials (optional): [s][N] = sum_t dY[t, n], the
 * bias gradient of the same Linear.  N % 256 == 0, K % 256 == 0, T % 64 == 0.
 * Finish both with mdlm_splitk_reduce. */
int mdlm_gemm_tn_splitk(const void* dY, int64_t ldy, const void* X, int64_t ldx,
                        float* partials, float* bias_partials, int64_t T, int N,
                        int K, int splits, void* stream);

/* Bidirectional attention, dit.py:265-275 (flash_attn_varlen_qkvpacked_func,
 * equal lengths, no dropout, softmax scale 1/sqrt(64)): q,k (n_seq,H,L,64)
 * bf16, vt (n_seq,H,64,L) bf16 -> out (n_seq*L, H*64) bf16.  L % 128 == 0. */
int mdlm_attn_fwd(const void* q, const void* k, const void* vt, void* out,
                  int n_seq, int H, int L, const int32_t* n_seq_dev,
                  void* stream);

/* ------------------------------------------------------- sampler plumbing */

/* Build the active-sample list from per-sample `changed` flags (device side
 * replacement for the host decision at diffusion.py:682-685): active[0..n)
 * = ascending sample ids with changed != 0, *n_active = n, *n_rows = n*L.
 * all_or_nothing != 0 reproduces the reference policy literally: if any
 * sample changed, every sample is listed.  nfe_accum (optional int64):
 * += n (backbone evaluations, in samples).  fresh (optional, B int32):
 * fresh[b] = 1 if sample b is listed, else 0 (input of mdlm_subs_ddpm_update). */
int mdlm_compact_active(const int32_t* changed, int32_t* active,
                        int32_t* n_active, int32_t* n_rows, int B, int L,
                        int all_or_nothing, int64_t* nfe_accum, int32_t* fresh,
                        void* stream);

/* Row lists of the MASKED rows of the active samples: only those rows need
 * logits (an unmasked token is carried over, diffusion.py:609-610, and SUBS
 * ignores its logits, 274-276).  gather[k] = compacted workspace row
 * (i*L + pos for active[i]), scatter[k] = logits-cache row (active[i]*L + pos),
 * in ascending order, padded with (0, -1) to a multiple of `pad`;
 * *n_out = padded count.  No reference counterpart (the reference projects
 * every row, dit.py:320). */
int mdlm_masked_row_map(const int64_t* x, const int32_t* active,
                        const int32_t* n_active, int L, int mask_index, int pad,
                        int32_t* gather, int32_t* scatter, int32_t* n_out,
                        void* stream);

/* ------------------------------------------------------------ training step
 * SURVEY §8 (f)-3.  The reference obtains every backward from torch autograd
 * (and flash-attn's backward); these are the hand-written halves for the DiT
 * block under bf16 autocast with an fp32 residual stream.  Backward GEMMs are
 * mdlm_gemm_bf16 on operands made K-contiguous by mdlm_transpose_bf16. */

/* mdlm_attn_fwd that also writes lse (n_seq,H,L) fp32 = log2-domain
 * log-sum-exp of the scaled scores, the statistic flash-attn saves for its
 * backward (dit.py:272-273 under autograd). */
int mdlm_attn_fwd_lse(const void* q, const void* k, const void* vt, void* out,
                      float* lse, int n_seq, int H, int L, void* stream);

/* Backward of dit.py:265-275: q,k head-major, vt, o/dout (n_seq*L, H*64),
 * lse from mdlm_attn_fwd_lse, delta_ws (n_seq*H*L fp32 workspace) ->
 * dq, dk, dv head-major (n_seq,H,L,64) bf16.  L % 64 == 0.  No atomics. */
int mdlm_attn_bwd(const void* q, const void* k, const void* vt, const void* o,
                  const void* dout, const float* lse, float* delta_ws, void* dq,
                  void* dk, void* dv, int n_seq, int H, int L, void* stream);

/* d(qkv) (n_seq*L, 3*H*64) bf16 from dq, dk, dv: inverse rotary on dq / dk
 * (dit.py:112-115 transposed) + the 'three h d' packing of dit.py:257-260. */
int mdlm_qkv_grad_pack(const void* dq, const void* dk, const void* dv,
                       const float* rot_cos, const float* rot_sin, void* dqkv,
                       int n_seq, int H, int L, void* stream);

/* Backward of mdlm_ln_modulate (dit.py:126-134 + 74-78): dx_inout (+)= dLN,
 * partial[(rows+31)/32][3][D] = per-32-row column sums of (dw, dscale, dshift). */
int mdlm_ln_modulate_bwd(const void* dh_bf16, const float* x, const float* ln_w,
                         const float* scale, float* dx_inout, float* partial,
                         int64_t rows, int D, int64_t rows_per_mod, int n_mod,
                         int64_t mod_ld, int accumulate, void* stream);

/* out[o][c] = sum_{i < per_out} partial[o*per_out + i][c] (fixed order). */
int mdlm_colsum_reduce(const float* partial, float* out, int64_t C, int per_out,
                       int n_out, void* stream);

/* partial[(rows+127)/128][N] = per-128-row column sums of a bf16 matrix (bias
 * gradients: the reference's autograd of nn.Linear bias, dit.py:224-227, 306). */
int mdlm_colsum_bf16_partial(const void* x_bf16, int64_t ld, float* partial,
                             int64_t rows, int N, void* stream);

/* bias_dropout_add_scale with training=True (dit.py:20-34): out = residual +
 * gate * dropout(y, p); keep mask = Philox(seed, offset) in the element layout
 * of mdlm_philox_uniform(rows, D), regenerated by the backward. */
int mdlm_dropout_gate_residual_fwd(const void* y_bf16, const float* gate,
                                   const float* residual, float* out,
                                   int64_t rows, int D, int64_t rows_per_mod,
                                   int n_mod, int64_t mod_ld, float p,
                                   uint64_t seed, uint64_t offset, void* stream);

/* dy = dout * gate * keep/(1-p) (bf16); partial[(rows+31)/32][D] = per-32-row
 * column sums of dout * dropout(y) (the gate gradient). */
int mdlm_dropout_gate_residual_bwd(const float* dout, const void* y_bf16,
                                   const float* gate, void* dy_bf16,
                                   float* partial, int64_t rows, int D,
                                   int64_t rows_per_mod, int n_mod,
                                   int64_t mod_ld, float p, uint64_t seed,
                                   uint64_t offset, void* stream);

/* GELU(tanh) of dit.py:226 and its derivative, bf16 elementwise (n % 8 == 0). */
int mdlm_gelu_fwd(const void* pre_bf16, void* act_bf16, int64_t n, void* stream);
int mdlm_gelu_bwd(const void* dact_bf16, const void* pre_bf16, void* dpre_bf16,
                  int64_t n, void* stream);

/* out[b][c][r] = in[b][r][c], bf16 (operand layout for the backward GEMMs;
 * no reference counterpart: torch.matmul takes strided operands).
 * colsum_partial (optional; batch 1, R, C and leading dimensions multiples of
 * 8): partial[i][c] = fp32 sum of rows 64 i .. 64 i + 63 of column c, i.e. the
 * first stage of the bias gradient of the Linear whose dY is transposed here
 * (ceil(R/64) x C floats; finish with mdlm_colsum_reduce). */
int mdlm_transpose_bf16(const void* in, int64_t ld_in, int64_t batch_stride_in,
                        void* out, int64_t ld_out, int64_t batch_stride_out,
                        int R, int C, int batch, float* colsum_partial,
                        void* stream);

/* dtable[idx[r]] += dx[r]: backward of dit.py:298-299.  fp32 atomics, except
 * for rows whose index is hot_index (>= 0: the mask token, which about half of
 * a q_xt batch holds): those are summed in a fixed order through hot_partial
 * (ceil(rows/128) x D floats) and WRITTEN to dtable[hot_index].  hot_index < 0:
 * atomics only, hot_partial may be NULL. */
int mdlm_embed_grad(const float* dx, const int64_t* idx, float* dtable,
                    int64_t rows, int D, int vocab, int hot_index,
                    float* hot_partial, void* stream);

/* d(loss)/d(logits) of diffusion.py:261-277 + 884-916 for loss = sum_r
 * row_weight[r] * nll[r]: w*(softmax - onehot(x0)) on masked rows, 0 in the
 * mask column, on unmasked rows and in the padding columns [V, ldo). */
int mdlm_subs_nll_bwd(const void* logits, int64_t ld, int dtype,
                      const int64_t* xt, const int64_t* x0,
                      const float* row_weight, void* dlogits_bf16, int64_t ldo,
                      int64_t rows, int V, int mask_index, void* stream);

/* out[0] = sum g^2 (two fixed-order stages; partial_1024 = 1024 floats). */
int mdlm_grad_sumsq(const float* g, int64_t n, float* partial_1024, float* out,
                    void* stream);

/* mdlm_grad_sumsq without the workspace argument and without any allocation.
 * NO COUNTERPART IN THE REFERENCE: it exists for callers that cannot allocate
 * or carry a workspace (code under HIP graph capture, tight serving / training
 * loops, harnesses that forbid hipMalloc).  The 1024 partial sums live in a
 * file-scope device array of the library (4 KiB per GPU, held until the process
 * ends); the entry point calls no hipMalloc / hipFree / allocating routine.
 * Same kernels, same summation order, hence bit-identical to mdlm_grad_sumsq.
 * n > MDLM_SUMSQ_STATIC_MAX_N is rejected (-1, message).  The first call on a
 * device looks the array's address up (hipGetSymbolAddress): make that call
 * outside a graph capture.  All calls on one GPU share the array -- every call
 * rewrites all of it, so consecutive calls are independent, but calls that
 * OVERLAP on one GPU (two streams) race on it. */
#define MDLM_SUMSQ_STATIC_MAX_N (1ll << 34)
int mdlm_grad_sumsq_static(const float* g, int64_t n, float* out, void* stream);

/* clip_grad_norm_ (configs/config.yaml:69) + torch.optim.AdamW step
 * (diffusion.py:449-459) + EMA (models/ema.py:44-49, diffusion.py:254-259) +
 * bf16 copy of the new weights, on flat buffers.  g is scaled by grad_scale
 * (1/world after a sum all-reduce) and by min(1, max_norm/(norm+1e-6)) with
 * norm = grad_scale*sqrt(*sumsq) when sumsq != NULL and max_norm > 0.
 * ema / w_bf16 may be NULL.  step is 1-based. */
int mdlm_adamw_ema(float* p, const float* g, float* m, float* v, float* ema,
                   void* w_bf16, int64_t n, float lr, float beta1, float beta2,
                   float eps, float weight_decay, int64_t step,
                   float ema_one_minus_decay, const float* sumsq, float max_norm,
                   float grad_scale, void* stream);

/* ------------------------------------------------------- workspace sizes
 * SURVEY 8(b) "Ownership": the caller allocates every buffer and sizes the
 * workspaces with these queries (host arithmetic only; bytes, or -1 for a bad
 * shape).  No reference counterpart: torch allocates inside its operators.
 *   attn_bwd            : delta_ws
 *   grad_sumsq          : partial_1024
 *   gemm_bf16_splitk    : partials
 *   gemm_tn_splitk      : partials (return value), bias_partials (*bias_partials_bytes)
 *   embed_grad          : hot_partial
 *   ln_modulate_bwd, dropout_gate_residual_bwd, colsum_bf16_partial : partial
 *   transpose_bf16      : colsum_partial
 *   masked_row_map      : gather and scatter, each (rows = B*L) */
int64_t mdlm_attn_bwd_workspace_bytes(int n_seq, int H, int L);
int64_t mdlm_grad_sumsq_workspace_bytes(void);
int64_t mdlm_gemm_bf16_splitk_workspace_bytes(int64_t M, int N, int splits);
int64_t mdlm_gemm_tn_splitk_workspace_bytes(int N, int K, int splits,
                                            int64_t* bias_partials_bytes);
int64_t mdlm_embed_grad_workspace_bytes(int64_t rows, int D);
int64_t mdlm_ln_modulate_bwd_workspace_bytes(int64_t rows, int D);
int64_t mdlm_dropout_gate_residual_bwd_workspace_bytes(int64_t rows, int D);
int64_t mdlm_colsum_bf16_partial_workspace_bytes(int64_t rows, int N);
int64_t mdlm_transpose_bf16_workspace_bytes(int R, int C);
int64_t mdlm_masked_row_map_workspace_bytes(int64_t rows, int pad);
int64_t mdlm_gemm_lse_partials_workspace_bytes(int64_t M, int N);   /* lse_partials_out */

#ifdef __cplusplus
}
#endif
#endif
