/* dinox.h -- C ABI of libdinox_hip.so, the MI355X (gfx950) kernel library for the DINO-X hot path.
 *
 * The reference (timlawrenz/DINO-X) has no FFI or operator registry: every op on its hot path is
 * an ATen call made from Python.  The drop-in seam is therefore the Python
 * module surface (zoo.arch / zoo.hub / zoo.encode / scripts/phase5_big_run.py), and this header is
 * the boundary *underneath* it: one entry point per ATen call sequence that the reference issues
 * on the path.  Each entry cites the reference lines it replaces (paths relative to the reference
 * checkout).  INTEGRATION.md shows the ctypes binding a reference maintainer would add.
 *
 * Conventions (all entries):
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless it says "host";
 *   - the caller allocates and owns every buffer, including workspaces (sizes via *_ws_bytes);
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *     no entry synchronises, allocates or frees, so every entry is hipGraph-capturable;
 *   - return 0 on success, a negative DINOX_E* code on bad arguments, or the positive
 *     hipError_t of a failed launch; the message is kept per thread for dinox_last_error();
 *   - re-entrant from any host thread (autograd's backward thread included): no mutable globals;
 *   - dtype codes: activations/weights may be fp32 ("parity mode", exact-fp32 MFMA / VALU) or
 *     bf16 ("throughput mode": bf16 MFMA operands, fp32 accumulation); the residual stream,
 *     LayerNorm statistics, softmax, losses, gradients of parameters and optimiser state are
 *     always fp32, mirroring what torch.autocast does on the reference path.
 */
#ifndef DINOX_H
#define DINOX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DINOX_ABI_VERSION 3   /* 3: + dinox_block_forward / _backward, dinox_gemm_timer_* (additive) */

/* dtype codes */
#define DINOX_F32 0
#define DINOX_BF16 1

/* error codes (negative); positive returns are hipError_t values */
#define DINOX_OK 0
#define DINOX_EINVAL (-1)      /* bad argument (null pointer, non-positive size, ...) */
#define DINOX_EUNSUPPORTED (-2) /* combination not implemented (e.g. bf16 GEMM with transA!=transB) */
#define DINOX_EALIGN (-3)      /* pointer / leading dimension not aligned as the bf16 path needs */

int dinox_version(void);
/* Human-readable description of the last failure on the calling thread ("" if none). */
const char* dinox_last_error(void);
/* 1 if device 0 is a gfx950 part and kernels can launch, 0 otherwise (host query, no launch). */
int dinox_device_ok(void);

/* ------------------------------------------------------------------------------------------
 * GEMM with fused epilogue -- replaces nn.Linear / nn.Conv2d(k=s=patch) / torch.bmm:
 *   zoo/arch.py:46 (qkv), :53 (proj), :76 (fc1 -> GELU -> fc2), :216 (patch_embed),
 *   :253-255 (head), scripts/phase5_big_run.py:727 (Gram bmm), and their autograd backward.
 *
 *   C[b][m][n] = epilogue( alpha * sum_k A(b,m,k) * B(b,n,k) )
 *   transA = 0: A stored [M][K] (K contiguous, lda >= K);  1: A stored [K][M] (lda >= M)
 *   transB = 0: B stored [N][K] (nn.Linear weight layout, ldb >= K);  1: B stored [K][N]
 *   in_dtype applies to A and B; out_dtype to C and aux.  bf16 inputs support (0,0) "NT" and
 *   (1,1) "TN" only (pre-transposed bf16 weight copies make every hot-path product one of them).
 *
 * epilogue bits (applied in this order):
 *   BIAS      acc += bias[n]                       (fp32 [N])
 *   GELU      if aux: aux = acc (pre-activation, out_dtype, ld = ldaux); acc = gelu_erf(acc)
 *   DGELU     acc *= gelu_erf'(aux[m][n])          (aux read, out_dtype)
 *   RESIDUAL  acc += residual[m][n]                (fp32, ld = ldr)
 *   ACCUM     C += acc instead of C = acc          (fp32 C only)
 * ------------------------------------------------------------------------------------------ */
#define DINOX_EPI_BIAS 1
#define DINOX_EPI_GELU 2
#define DINOX_EPI_DGELU 4
#define DINOX_EPI_RESIDUAL 8
#define DINOX_EPI_ACCUM 16
#define DINOX_EPI_AUXGRAD 32   /* modifies GELU: aux receives gelu_erf'(pre-activation) instead of the pre-activation;
                                * modifies DGELU: acc *= aux (aux already holds the derivative).  Saves the
                                * transcendental work of the backward epilogue: forward evaluates erf/exp once for both. */

typedef struct dinox_gemm_args {
  const void* A;
  const void* B;
  void* C;
  int64_t M, N, K;
  int64_t lda, ldb, ldc;
  int64_t batch;                       /* >= 1 */
  int64_t strideA, strideB, strideC;   /* elements between batch items (0 = shared operand) */
  int32_t transA, transB;
  int32_t in_dtype, out_dtype;
  int32_t epilogue;
  float alpha;
  const float* bias;
  const float* residual;
  int64_t ldr;
  void* aux;
  int64_t ldaux;
  float* colsum;                       /* optional, transA = 1 only: colsum[m] = sum_k A(m,k) (overwritten; added to
                                        * under ACCUM, like C) -- the bias
                                        * gradient sum_rows(dY) rides along the dW = dY^T X product that already streams dY */
  void* ws;                            /* optional workspace of dinox_gemm_ws_bytes() bytes; its contents need not be initialised.  With it a
                                        * split-K product (the dW = dY^T X products, K = every token of the batch) runs as TWO launches on
                                        * the caller's stream: the split kernel stores its partial tiles there by plain stores, a reduction
                                        * kernel then sums them in split order into C -- no counters, no atomics: results are
                                        * bit-reproducible from run to run, and 33 MB of memory-side atomics per launch (1.3 TB/s on this
                                        * part) become plain stores (6 TB/s).  NULL: fp32 atomics into C (one launch, not reproducible).
                                        * One product at a time per workspace; reuse in stream order is fine.  (NT products ignore it,
                                        * except that tools/pp_stamps.py hands the ping-pong kernels a diagnostic stamp buffer here.) */
} dinox_gemm_args;

int dinox_gemm(const dinox_gemm_args* args, void* stream);
/* Bytes of workspace that make this product deterministic (0: it needs none / cannot use one). */
int64_t dinox_gemm_ws_bytes(const dinox_gemm_args* args);
/* Name of the device kernel dinox_gemm would launch for these arguments ("gemm_bf16_nt", "gemm_bf16_tn",
 * "gemm_f32"); host-only query used by bench.py to attribute per-launch timings.  Static string. */
const char* dinox_gemm_kernel_name(const dinox_gemm_args* args);

/* out[n] (+)= sum_m x[m][n]   -- bias gradients (autograd of nn.Linear bias, zoo/arch.py:40-41,71-73). */
int dinox_colsum(const void* x, float* out, int64_t M, int64_t N, int64_t ldx, int dtype, int accumulate,
                 void* stream);

/* ------------------------------------------------------------------------------------------
 * Linear + residual + LayerNorm in one launch (bf16 mode, model width N = 384) -- replaces, inside a pre-norm block
 * (zoo/arch.py:94-97), the tail of one sub-block and the head of the next:
 *     x_out = residual + a W^T + bias          (proj :53 / fc2 :76 and the residual add :95 / :96; fp32 residual stream)
 *     y     = LayerNorm(x_out; gamma, beta)    (the following norm2 / next block's norm1 / final norm :96,:95,:237)
 * a: [M,K] bf16, w: [N,K] bf16 (nn.Linear layout), bias [N] or NULL, residual [M,N] fp32 or NULL; x_out [M,N] fp32;
 * y [M,N] in y_dtype (DINOX_BF16 when it feeds the next GEMM, DINOX_F32 for the model's final norm); mean, rstd [M]
 * (biased variance, what dinox_layernorm_bwd consumes).  One workgroup owns 128 complete rows, so the residual stream is
 * not read back by a separate LayerNorm launch.  dinox_linear_residual_ln_ok tells whether the fused kernel takes a shape
 * (N == 384, K % 32 == 0); otherwise use dinox_gemm + dinox_layernorm_fwd.
 * ------------------------------------------------------------------------------------------ */
int dinox_linear_residual_ln_ok(int64_t M, int N, int K);
int dinox_linear_residual_ln(const void* a, const void* w, const float* bias, const float* residual, float* x_out,
                             const float* gamma, const float* beta, float eps, void* y, int y_dtype, float* mean,
                             float* rstd, int64_t M, int N, int K, void* stream);

/* ------------------------------------------------------------------------------------------
 * LayerNorm -- replaces nn.LayerNorm(D), eps 1e-5, affine (zoo/arch.py:89,91,126,187; calls :95,96,237).
 * x is the fp32 residual stream; y is written in out_dtype (bf16 when it only feeds a GEMM).
 * bwd: dx = (dx_add ? dx_add : 0) + LN'(dy)  -- dx_add is the gradient arriving over the skip connection of the
 *      pre-norm block (may be NULL, may alias dx);  dw, db overwritten, or added to when accumulate != 0 (gradient
 *      arena).  ws: dinox_layernorm_bwd_ws_bytes(rows, dim) bytes.
 *      dx_lowp (optional, may be NULL): bf16 copy of the final dx -- the residual-stream gradient is the
 *      dY operand of the next backward GEMMs, which autocast rounds to bf16 at that point as well.
 * ------------------------------------------------------------------------------------------ */
int dinox_layernorm_fwd(const float* x, const float* w, const float* b, void* y, float* mean, float* rstd,
                        int64_t rows, int dim, float eps, int out_dtype, void* stream);
int64_t dinox_layernorm_bwd_ws_bytes(int64_t rows, int dim);
int dinox_layernorm_bwd(const void* dy, const float* x, const float* w, const float* mean, const float* rstd,
                        float* dx, const float* dx_add, void* dx_lowp, float* dw, float* db, void* ws,
                        int64_t rows, int dim, int dy_dtype, int accumulate, void* stream);
/* The input-gradient product in front of a LayerNorm and that LayerNorm's backward as ONE launch (width N = 384, bf16 operands;
 * backward of zoo/arch.py:95-96 with :46 / :75): dy = a w^T (a [M,K], w [384,K] = W^T of the Linear, rounded to bf16 as dinox_gemm
 * would hand it over), then dinox_layernorm_bwd's contract on it -- dx equal to the two calls to the last bit.  ws as dinox_layernorm_bwd. */
int dinox_linear_ln_bwd_ok(int64_t M, int N, int K);
int dinox_linear_ln_bwd(const void* a, const void* w, const float* x, const float* gamma, const float* mean, const float* rstd,
                        float* dx, const float* dx_add, void* dx_lowp, float* dgamma, float* dbeta, void* ws, int64_t M, int N,
                        int K, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------
 * Multi-head self-attention core -- replaces the reshape/permute/unbind +
 * F.scaled_dot_product_attention + transpose/reshape of zoo/arch.py:45-52 (scale 1/sqrt(d), no mask).
 * qkv is the packed output of the qkv Linear, [B][N][3][heads][d]; o is [B][N][heads*d];
 * lse is the per-row log-sum-exp of the scaled scores, [B][heads][N] fp32 (saved for backward).
 * bwd recomputes P from lse (flash style); dqkv has the layout of qkv; ws: dinox_attention_bwd_ws_bytes() bytes.
 * ------------------------------------------------------------------------------------------ */
int dinox_attention_fwd(const void* qkv, void* o, float* lse, int B, int N, int heads, int d, int dtype,
                        void* stream);
/* Rows of a materialised fp32 score matrix [rows][n] (rows = (image, query) pairs of one head; row r's log-sum-exp lives at
 * lse[(r / rows_per_image) * lse_image_stride + r % rows_per_image], i.e. inside the [B][heads][N] tensor of the calls above):
 *   softmax_rows:     s <- softmax(s) in place, lse written;
 *   softmax_bwd_rows: s <- p = exp(s - lse), dp <- ds = p * (dp - sum_j p_j dp_j) * scale, both in place.
 * Used by the fp32 parity mode, which runs full-size attention as batched exact-fp32 products around these two. */
int dinox_softmax_rows(float* s, float* lse, int64_t rows, int n, int rows_per_image, int64_t lse_image_stride, void* stream);
int dinox_softmax_bwd_rows(float* s, float* dp, const float* lse, float scale, int64_t rows, int n, int rows_per_image,
                           int64_t lse_image_stride, void* stream);
int64_t dinox_attention_bwd_ws_bytes(int B, int N, int heads);
int dinox_attention_bwd(const void* d_o, const void* qkv, const void* o, const float* lse, void* dqkv, void* ws,
                        int B, int N, int heads, int d, int dtype, void* stream);
/* north_star's "fused QKV projection + multi-head attention" as ONE launch, for passes that keep nothing for a backward (the teacher
 * of a training step, encode()): qkv = x wqkv^T + bias (zoo/arch.py:46 `self.qkv`) and softmax(q k^T / sqrt(d)) v (:47-52) per
 * (image, head) without the packed qkv tensor ever reaching HBM.  bf16: x [B*N][D], wqkv [3*heads*d][D], bias fp32 [3*heads*d] or NULL
 * -> o [B*N][heads*d].  qkv_out ([B*N][3*heads*d]) and lse ([B*heads][N] fp32) are optional outputs (NULL: not written).
 * dinox_qkv_attention_ok: 1 inside the kernel's envelope (d = 64, 193 <= N <= 224, D % 32 == 0); outside it the entry returns
 * DINOX_EUNSUPPORTED and the caller composes dinox_gemm + dinox_attention_fwd.  Measured on MI355X it is on a par with those two
 * launches at ViT-S and slower at ViT-L: dinox_block_forward uses it only when asked to (qkv == NULL). */
int dinox_qkv_attention_ok(int B, int N, int heads, int d, int D);
int dinox_qkv_attention_fwd(const void* x, const void* wqkv, const float* bias, void* o, void* qkv_out, float* lse, int B, int N,
                            int heads, int d, int D, void* stream);

/* ------------------------------------------------------------------------------------------
 * Patch unfold + token assembly -- replaces the im2col half of nn.Conv2d(3,D,k=p,s=p) and
 * flatten/transpose/cat/+pos_embed/+scale_embed/cat of zoo/arch.py:216-229.
 * unfold: x fp32 NCHW [V][3][H][W] -> u [V*P][3*p*p] in out_dtype (the A operand of the patch GEMM;
 *         computed once per step and shared by student and teacher, which see the same batch).
 * assemble fwd: tokens[v][0] = cls + pos[0] (+scale[v]);  tokens[v][1+i] = patches[v][i] + pos[1+i] (+scale[v]);
 *               tokens[v][1+P+r] = registers[r].   tokens is the fp32 residual stream [V][N][D].
 * assemble bwd: dpatches (dtype) for the patch GEMM's dW/db; dcls, dpos, dregs reduced over the batch;
 *               dscale[v] = sum over the 1+P body tokens (NULL when not scale-aware).
 * ------------------------------------------------------------------------------------------ */
int dinox_patch_unfold(const float* x, void* u, int V, int H, int W, int patch, int out_dtype, void* stream);
/* The same with rows of ld >= 3 p^2 elements, the tail zero-filled (patch sizes whose 3 p^2 is no multiple of 8: the operand of the
 * MFMA bf16 products is padded; the weight operand gets zero columns to match). */
int dinox_patch_unfold_ld(const float* x, void* u, int V, int H, int W, int patch, int ld, int out_dtype, void* stream);
int dinox_tokens_fwd(const void* patches, const float* cls, const float* pos, const float* registers,
                     const float* scale, float* tokens, int V, int P, int R, int D, int patches_dtype,
                     void* stream);
int dinox_tokens_bwd(const float* dtokens, void* dpatches, float* dcls, float* dpos, float* dregs,
                     float* dscale, int V, int P, int R, int D, int patches_dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * ScaleEmbedding -- replaces Linear(3,h) -> GELU -> Linear(h,D) -> LayerNorm(D) (zoo/arch.py:119-140).
 * All fp32 (V rows only).  fwd saves hpre [V][h], e [V][D] (pre-LN), mean/rstd [V] for backward.
 * bwd overwrites every parameter gradient and dspacing [V][3] (may be NULL).
 * ------------------------------------------------------------------------------------------ */
int dinox_scale_embed_fwd(const float* spacing, const float* w0, const float* b0, const float* w2,
                          const float* b2, const float* lnw, const float* lnb, float* out, float* hpre,
                          float* e, float* mean, float* rstd, int V, int h, int D, float eps, void* stream);
int64_t dinox_scale_embed_bwd_ws_bytes(int V, int h, int D);
int dinox_scale_embed_bwd(const float* dout, const float* spacing, const float* w0, const float* w2,
                          const float* lnw, const float* hpre, const float* e, const float* mean,
                          const float* rstd, float* dw0, float* db0, float* dw2, float* db2, float* dlnw,
                          float* dlnb, float* dspacing, void* ws, int V, int h, int D, void* stream);

/* ------------------------------------------------------------------------------------------
 * DINO centring/sharpening cross-entropy -- replaces DINOLoss.forward/update_center
 * (scripts/phase5_big_run.py:686-720).  s, t: [2B][K] fp32 logits, rows [view1; view2].
 *   loss[0] = (1/2B) sum_i -sum_k softmax((t[pair(i)]-center)/tt)[k] * log_softmax(s[i]/ts)[k],
 *   pair(i) = (i+B) mod 2B;   ds = grad_scale * dloss/ds (NULL to skip);  row_loss: [2B] workspace.
 * colmean: out[k] = mean_i t[i][k]  (the batch centre; all-reduced across ranks under DP before
 * dinox_center_ema applies  center = center*m + mean*(1-m), phase5_big_run.py:689-690).
 * ------------------------------------------------------------------------------------------ */
int dinox_dino_ce(const float* s, const float* t, const float* center, float student_temp, float teacher_temp,
                  float grad_scale, float* loss, float* ds, float* row_loss, int rows2B, int K, void* stream);
/* Multi-crop form (an extension: the reference trains on 2 global views only).  s: [n_views][B][K] student logits, view-major,
 * the first n_global views being the ones the teacher saw; t: [n_global][B][K].  Every pair (teacher view q, student view
 * v != q) contributes mean_b of the cross-entropy above; loss[0] = their average over the n_global*(n_views-1) pairs.
 * n_global = n_views = 2 is dinox_dino_ce.  ws: (n_views + 2 n_global) * B floats. */
int dinox_dino_ce_multi(const float* s, const float* t, const float* center, float student_temp, float teacher_temp,
                        float grad_scale, float* loss, float* ds, float* ws, int B, int n_global, int n_views, int K,
                        void* stream);
int dinox_colmean(const float* t, float* out, int rows, int K, void* stream);
int dinox_center_ema(float* center, const float* batch_mean, float momentum, int K, void* stream);

/* ------------------------------------------------------------------------------------------
 * Gram anchoring -- replaces compute_gram_matrix / compute_gram_anchoring_loss
 * (scripts/phase5_big_run.py:723-739): tokens 1..N-1 (registers included), F.normalize eps 1e-12,
 * G = Xh Xh^T, mse_loss mean.  The Gram products themselves go through dinox_gemm (batched);
 * these entries are the fused pieces around it:
 *   normalize: cat[v][t][0:D] = s_hat, cat[v][t][D:2D] = t_hat; catneg = [s_hat, -t_hat] (in_dtype of
 *              the GEMM), so that diff = cat * catneg^T = Gs - Gt in ONE batched NT GEMM with K = 2D;
 *              snorm [V][T] = max(||s||, eps) and a copy of s_hat alone (shat, GEMM dtype) for backward.
 *   sqsum:     loss[0] = scale * sum(diff^2)   (scale = 1/(V*T*T));  ws: [blocks] fp32, >= 1024 floats.
 *   normalize_bwd: dfeats[v][1+t] (+)= (dxh - xh (xh.dxh)) / norm  (dxh / eps when ||x|| <= eps);
 *              dfeats[v][0] untouched (CLS does not enter the Gram loss).
 * ------------------------------------------------------------------------------------------ */
int dinox_gram_normalize(const float* sfeats, const float* tfeats, void* cat, void* catneg, void* shat,
                         float* snorm, int V, int N, int D, int out_dtype, void* stream);
int dinox_sqsum(const float* x, int64_t n, float scale, float* loss, float* ws, void* stream);
int dinox_gram_normalize_bwd(const float* dxh, const void* shat, const float* snorm, const float* sfeats,
                             float* dfeats, int V, int N, int D, int shat_dtype, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------
 * View pipeline for the 2.5D slice stacks -- replaces, after the PNG decode, PngDataset._load_hu01 and the
 * torchvision transform stack (scripts/phase5_big_run.py:493-497, 516-528, 549-555): stored u16 -> HU ->
 * window -> RandomResizedCrop (antialiased bicubic = torch's _upsample_bicubic2d_aa, align_corners = 0) ->
 * horizontal flip -> (x - mean) / std, one kernel, out = fp32 [V][3][S][S] (the batch PatchViT.forward takes).
 * The random draws stay on the host:  view_i[v] = {element offset of the (3,H,W) u16 stack in raw, H, W, top,
 * left, h, w, flip};  view_f[v] = {level - width/2, max(width, 1)}   (both tables in device memory).
 * max_crop = the largest h or w in the table (sizes the LDS footprint; a view that exceeds it is written as NaN).
 * ------------------------------------------------------------------------------------------ */
int64_t dinox_slice_views_lds_bytes(int S, int max_crop);
int dinox_slice_views(const void* raw_u16, const int64_t* view_i, const float* view_f, float* out, int V, int S, int max_crop,
                      void* stream);
/* The same pipeline writing the patch-embed operand directly: u[(v P + gy g + gx)][c p^2 + py p + px] (row stride ld >= 3 p^2,
 * g = S / patch, out_dtype DINOX_BF16 | DINOX_F32; columns 3 p^2 .. ld-1 are zeroed) -- bit for bit what dinox_patch_unfold(_ld)
 * makes of dinox_slice_views' output, without writing and re-reading the fp32 image batch (4 + 4 B per pixel). */
int dinox_slice_views_patches(const void* raw_u16, const int64_t* view_i, const float* view_f, void* u, int V, int S, int max_crop,
                              int patch, int ld, int out_dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * KoLeo regulariser -- replaces KoLeoLoss.forward (scripts/phase5_big_run.py:742-773), which the loop applies
 * to the student head output (:1764-1766): x^ = F.normalize(x); d_i = min_{j != i} ||x^_i - x^_j||;
 * loss = -mean_i log(d_i + eps).  fp32 throughout.  The all-pairs products G = X^_local X^_all^T go through
 * dinox_gemm (fp32); these entries are the pieces around it.  "local" rows are this rank's V_l rows, at
 * global positions row0 .. row0+V_l-1 of the V_g gathered rows (V_l = V_g, row0 = 0 on one GPU).
 *   normalize: xh = x / max(||x||, eps); norm[r] = ||x_r||; sq[r] = ||xh_r||^2.
 *   nn:        idx[i] = argmin_{j != row0+i} (sq_i + sq_j - 2 G[i][j]) (lowest j on ties; -1 if V_g = 1),
 *              dist[i] = ||xh_i - xh_idx|| measured on the rows themselves.
 *   bwd:       dx[r] = d(-gscale * sum_i log(dist_i + eps)) / dx_r over ALL V_g pairs (idx_all/dist_all are
 *              the gathered results of nn): the row's own pair plus every pair that chose r as neighbour,
 *              then back through the normalisation (norm_eps = the eps given to normalize).
 *              gscale = upstream gradient / V_l.  V_g <= 15359.
 * ------------------------------------------------------------------------------------------ */
int dinox_koleo_normalize(const float* x, float* xh, float* norm, float* sq, int64_t V, int D, float eps, void* stream);
int dinox_koleo_nn(const float* G, int64_t ldg, const float* sq_all, const float* xh_all, int row0, int V_l, int V_g, int D,
                   int* idx, float* dist, void* stream);
/* loss[0] = -mean_i log(dist[i] + eps) over the V local rows (fixed summation order). */
int dinox_koleo_loss(const float* dist, int V, float eps, float* loss, void* stream);
int dinox_koleo_bwd(const float* xh_all, const int* idx_all, const float* dist_all, const float* norm_loc, int row0, int V_l,
                    int V_g, int D, float gscale, float eps, float norm_eps, float* dx, void* stream);

/* ------------------------------------------------------------------------------------------
 * Optimiser tail -- replaces the per-parameter grad-norm loop (scripts/phase5_big_run.py:1784-1789),
 * torch.optim.AdamW.step (:1794; betas .9/.999, eps 1e-8, decoupled decay on EVERY parameter) and the
 * per-parameter EMA teacher update (:1799-1802) with ONE pass over flat fp32 arenas of n elements.
 *   g is multiplied by grad_scale first (1/world_size after a sum all-reduce);
 *   gnorm_sq[0] = sum((grad_scale*g)^2)  (fp32; ws: >= 4096 floats);  teacher may be NULL (no EMA).
 *   step_t = 1-based optimiser step for the bias corrections.
 * cast: bf16 copies of weights for the MFMA path (and [C][R] transposed copies for dX products).
 * ------------------------------------------------------------------------------------------ */
int dinox_adamw_ema(float* p, const float* g, float* m, float* v, float* teacher, int64_t n, float lr,
                    float weight_decay, float beta1, float beta2, float eps, int step_t, float ema,
                    float grad_scale, float* gnorm_sq, float* ws, void* stream);
/* The same pass with its per-step scalars in DEVICE memory -- hyper[3] = {lr, 1/(1-beta1^t), 1/sqrt(1-beta2^t)} -- so that the
 * launch can sit in a captured hipGraph and be replayed with a new learning rate / step count (the host writes hyper before
 * each replay).  New in this engine (the reference has no graph capture); same arithmetic as dinox_adamw_ema. */
int dinox_adamw_ema_dev(float* p, const float* g, float* m, float* v, float* teacher, int64_t n, const float* hyper,
                        float weight_decay, float beta1, float beta2, float eps, float ema, float grad_scale,
                        float* gnorm_sq, float* ws, void* stream);
int dinox_sumsq(const float* x, int64_t n, float* out, float* ws, void* stream);
int dinox_cast_bf16(const float* src, void* dst, int64_t n, void* stream);
int dinox_cast_transpose_bf16(const float* src, void* dst, int R, int C, void* stream);
/* All matrices of a parameter arena in one launch: table (device, int64 [n_mats][4]) holds per matrix {element offset (the same
 * in src_base and dst_base), R, C, index of its first 32x32 tile}; total_tiles = sum of ceil(R/32)*ceil(C/32). */
int dinox_cast_transpose_bf16_multi(const float* src_base, void* dst_base, const int64_t* table, int n_mats,
                                    int64_t total_tiles, void* stream);
/* ------------------------------------------------------------------------------------------
 * Glue between the big kernels, so that no framework elementwise kernel runs inside a training step.
 *   take_rows: dst[dst_row0 + v][0..D) = src[v * src_stride + 0..D)   (fp32 -> dst_dtype).  With src = features + 0 and
 *              src_stride = N*D this is `feats[:, 0]` (the CLS row handed to the DINO head, zoo/arch.py:260-261).
 *   put_rows:  dst[v * dst_stride + 0..D) (+)= src[src_row0 + v][0..D)   (src_dtype -> fp32): the head's input gradient written
 *              into row 0 of the feature gradient (the reference's slice backward: zero fill + copy + full-size add).
 *   axpy:      y += alpha * x (fp32).     lincomb3: out[0] = a[0] + wb*b[0] + wc*c[0] (b, c may be NULL): the total loss
 *              (scripts/phase5_big_run.py:1755-1766).     zero: asynchronous zero fill.
 * ------------------------------------------------------------------------------------------ */
int dinox_take_rows(const float* src, void* dst, int64_t V, int64_t src_stride, int D, int64_t dst_row0, int dst_dtype,
                    void* stream);
int dinox_put_rows(const void* src, float* dst, int64_t V, int64_t dst_stride, int D, int64_t src_row0, int src_dtype,
                   int accumulate, void* stream);
int dinox_axpy(float* y, const float* x, float alpha, int64_t n, void* stream);
int dinox_lincomb3(const float* a, const float* b, const float* c, float wb, float wc, float* out, void* stream);
int dinox_zero(void* p, int64_t bytes, void* stream);
/* Elementwise helpers used by the host-side modules: y = gelu_erf(x) / dx = dy * gelu_erf'(x) (fp32). */
int dinox_gelu_fwd(const float* x, float* y, int64_t n, void* stream);
int dinox_gelu_bwd(const float* dy, const float* x, float* dx, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------
 * One pre-norm transformer block per call (bf16 throughput mode) -- the launch SEQUENCE of zoo/arch.py:94-97 with Attention :43-54 and
 * Mlp :75-76 inlined, enqueued by the library instead of by ~13 (forward) / ~15 (backward) calls from the host language:
 *     x1 = x0 + proj(attention(qkv(norm1(x0))));   x2 = x1 + fc2(gelu(fc1(norm2(x1))))
 * Same kernels, same order, same results as the entries above called one by one (LayerNorm, dinox_gemm with its fused epilogues,
 * dinox_attention_*, dinox_linear_residual_ln); still no allocation, no synchronisation, hipGraph-capturable, re-entrant: every tensor,
 * the saved activations of the backward pass and all workspaces are the caller's.  What it buys is host time: one foreign call per
 * block instead of one per launch (Python: 24 ms of enqueue per ViT-S bs-256 step before), under data parallelism and
 * gradient accumulation too.  All activations are row-major [V*N, .]; bf16 tensors are operands of the next product, fp32 ones the
 * residual stream, the statistics and the gradients.
 * forward:  xn1_in (+ mean1_in, rstd1_in) non-NULL = norm1(x0) was already produced by the previous block's epilogue; then xn1 /
 *           mean1 / rstd1 are not written.  fuse_proj_ln / fuse_fc2_ln: run the product and the LayerNorm behind it as ONE launch
 *           (dinox_linear_residual_ln; needs dinox_linear_residual_ln_ok).  next_g non-NULL: also return yn = LayerNorm(x2; next_g,
 *           next_b, next_eps) in next_dtype with its statistics (the next block's norm1, or the model's final norm).
 *           pre non-NULL (training): fc1 also writes gelu'(pre-activation) there (the DINOX_EPI_AUXGRAD side tensor).
 * backward: g = d loss / d x2 (fp32), g_lowp = its bf16 copy (NULL: cast here into g_lowp_buf).  Weight operands are the TRANSPOSED
 *           bf16 images W^T [in][out] (dX = dY . W as an NT product).  Parameter gradients are ACCUMULATED into dwqkv .. dn2b (slices
 *           of a gradient arena).  Scratch: dpre [M,H], dxn2 / d_o / dxn1 [M,D], dqkv [M,3D] bf16; g1 [M,D] fp32 and g1_lowp bf16.
 *           Outputs: g0 = d loss / d x0 written IN PLACE of g1 (same buffer), g0_lowp its bf16 copy.
 *           attn_ws: dinox_attention_bwd_ws_bytes; ln_ws: dinox_layernorm_bwd_ws_bytes(M, D) (used twice, in stream order);
 *           tn_ws / tn_ws_bytes: workspace of the deterministic dW products (>= the largest dinox_gemm_ws_bytes of the four).
 * ------------------------------------------------------------------------------------------ */
typedef struct dinox_block_fwd_args {
  int64_t V, N;                        /* views, tokens per view: M = V * N rows */
  int32_t D, H, heads;                 /* width, MLP hidden width, attention heads */
  int32_t train;                       /* != 0: `pre` receives the GELU' side tensor */
  int32_t fuse_proj_ln, fuse_fc2_ln;
  float eps;                           /* of norm1 / norm2 */
  const float* x0;
  const void* xn1_in; const float* mean1_in; const float* rstd1_in;
  void* xn1; float* mean1; float* rstd1;
  void* qkv; void* o; float* lse;
  float* x1; void* xn2; float* mean2; float* rstd2;
  void* act; void* pre;
  float* x2;
  const float* next_g; const float* next_b; float next_eps; int32_t next_dtype;
  void* yn; float* meann; float* rstdn;
  const float *n1w, *n1b, *n2w, *n2b;
  const void *wqkv, *wproj, *w1, *w2;  /* bf16 [3D,D], [D,D], [H,D], [D,H] */
  const float *bqkv, *bproj, *b1, *b2; /* fp32 or NULL */
} dinox_block_fwd_args;

typedef struct dinox_block_bwd_args {
  int64_t V, N;
  int32_t D, H, heads;
  int32_t reserved;                             /* flags: bit 0 = the two dX products into the LayerNorms run dinox_linear_ln_bwd */
  const float* g; const void* g_lowp; void* g_lowp_buf;
  /* saved by the forward */
  const float* x0; const float* x1; const void* xn1; const void* xn2; const void* qkv; const void* o; const float* lse;
  const void* pre; const void* act; const float *mean1, *rstd1, *mean2, *rstd2;
  const float *n1w, *n2w;
  const void *wqkv_t, *wproj_t, *w1_t, *w2_t;   /* bf16 W^T: [D,3D], [D,D], [D,H], [H,D] */
  /* gradient arena slices (accumulated into); bias gradients may be NULL */
  float *dwqkv, *dbqkv, *dwproj, *dbproj, *dw1, *db1, *dw2, *db2, *dn1w, *dn1b, *dn2w, *dn2b;
  /* scratch and outputs */
  void* dpre; void* dxn2; void* d_o; void* dqkv; void* dxn1;
  float* g1; void* g1_lowp; void* g0_lowp;
  void* attn_ws; void* ln_ws; void* tn_ws; int64_t tn_ws_bytes;
} dinox_block_bwd_args;

int dinox_block_forward(const dinox_block_fwd_args* args, void* stream);
int dinox_block_backward(const dinox_block_bwd_args* args, void* stream);

/* ------------------------------------------------------------------------------------------
 * Per-launch timing of dinox_gemm (diagnostic; bench.py's roofline object).  Between start and stop every dinox_gemm launch -- also
 * the ones dinox_block_* issue -- is counted per (kernel, shape, epilogue), and one launch in `every` (a fixed hash of the launch
 * counter) is bracketed by a HIP event pair on its stream.  stop synchronises those events and writes one text line per shape into
 * buf:  "<kernel> M N K batch epilogue in_dtype out_dtype has_aux shared_b launches timed ms_timed\n"  and returns the bytes written
 * (-1: buffer too small).  Not for use under graph capture; one timer per process.
 * ------------------------------------------------------------------------------------------ */
int dinox_gemm_timer_start(int every);
int64_t dinox_gemm_timer_stop(char* buf, int64_t buflen);

#ifdef __cplusplus
}
#endif
#endif /* DINOX_H */
