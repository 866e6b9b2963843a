// block.hip -- dinox_block_forward / dinox_block_backward: the launch sequence of one pre-norm transformer block (bf16 throughput mode)
// behind ONE C-ABI call each.  Nothing new runs on the device: these functions call the library's own entry points (LayerNorm, dinox_gemm
// with its fused epilogues, attention, linear + residual + LayerNorm) in the order the host-side block node (dinox/ops.py BlockFn) used to
// call them one by one -- ~13 launches forward, ~15 backward.  What changes is the host: a Python caller pays one foreign call and one
// struct per block instead of ~35 us per launch (24 ms of enqueue per ViT-S bs-256 step, of a 38 ms step), and the sequencing works under
// data parallelism and gradient accumulation, where a captured hipGraph (TrainEngine(use_graph=True)) does not.
// Reference: zoo/arch.py:94-97 (TransformerBlock.forward) with Attention :43-54 and Mlp :75-76, and their autograd backward.
#include <cstring>

#include "common.h"

using namespace dinox;

namespace {

dinox_gemm_args gemm_args(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int out_dtype) {
  dinox_gemm_args g;
  memset(&g, 0, sizeof g);
  g.A = A; g.B = B; g.C = C;
  g.M = M; g.N = N; g.K = K;
  g.lda = K; g.ldb = K; g.ldc = N;
  g.batch = 1;
  g.strideC = M * N;
  g.in_dtype = DINOX_BF16;
  g.out_dtype = out_dtype;
  g.alpha = 1.0f;
  g.ldr = N;
  g.ldaux = N;
  return g;
}

// dW += dy^T x (+ db += column sums of dy): dy [M, n_out], x [M, n_in] bf16, both stored token-major = the TN layout
int weight_grad(const void* dy, const void* x, float* dw, float* db, int64_t M, int64_t n_out, int64_t n_in, void* ws, int64_t ws_bytes, void* stream) {
  dinox_gemm_args g;
  memset(&g, 0, sizeof g);
  g.A = dy; g.B = x; g.C = dw;
  g.M = n_out; g.N = n_in; g.K = M;
  g.lda = n_out; g.ldb = n_in; g.ldc = n_in;
  g.batch = 1;
  g.strideC = n_out * n_in;
  g.transA = 1; g.transB = 1;
  g.in_dtype = DINOX_BF16;
  g.out_dtype = DINOX_F32;
  g.epilogue = DINOX_EPI_ACCUM;
  g.alpha = 1.0f;
  g.ldr = n_in;
  g.ldaux = n_in;
  g.colsum = db;
  const int64_t need = dinox_gemm_ws_bytes(&g);
  if (need > 0) {
    DX_REQUIRE(ws && need <= ws_bytes, DINOX_EINVAL, "block_backward: tn_ws of %lld bytes, the dW product [%lld x %lld] over %lld rows needs %lld",
               (long long)ws_bytes, (long long)n_out, (long long)n_in, (long long)M, (long long)need);
    g.ws = ws;
  }
  return dinox_gemm(&g, stream);
}

}  // namespace

#define BLK_TRY(call)        \
  do {                       \
    const int rc_ = (call);  \
    if (rc_) return rc_;     \
  } while (0)

extern "C" int dinox_block_forward(const dinox_block_fwd_args* a, void* stream) {
  DX_REQUIRE(a, DINOX_EINVAL, "block_forward: null args");
  DX_REQUIRE(a->V > 0 && a->N > 0 && a->D > 0 && a->H > 0 && a->heads > 0 && a->D % a->heads == 0, DINOX_EINVAL,
             "block_forward: V=%lld N=%lld D=%d H=%d heads=%d", (long long)a->V, (long long)a->N, a->D, a->H, a->heads);
  DX_REQUIRE(a->x0 && a->o && a->x1 && a->xn2 && a->mean2 && a->rstd2 && a->act && a->x2, DINOX_EINVAL, "block_forward: null activation");
  // qkv == NULL (a pass that keeps nothing for a backward): the qkv projection and the attention run as ONE launch and the packed qkv
  // tensor never exists (dinox_qkv_attention_fwd; the caller has asked dinox_qkv_attention_ok)
  DX_REQUIRE(a->qkv ? a->lse != nullptr : !a->train, DINOX_EINVAL, "block_forward: qkv / lse buffers (only a no-grad pass may leave qkv out)");
  DX_REQUIRE(a->n1w && a->n1b && a->n2w && a->n2b && a->wqkv && a->wproj && a->w1 && a->w2, DINOX_EINVAL, "block_forward: null parameter");
  DX_REQUIRE(a->xn1_in ? (a->mean1_in && a->rstd1_in) : (a->xn1 && a->mean1 && a->rstd1), DINOX_EINVAL, "block_forward: norm1 buffers");
  DX_REQUIRE(!a->next_g || (a->next_b && a->yn && a->meann && a->rstdn), DINOX_EINVAL, "block_forward: next LayerNorm buffers");
  DX_REQUIRE(!a->train || a->pre, DINOX_EINVAL, "block_forward: train without the side tensor");
  const int64_t M = a->V * a->N;
  const int D = a->D, H = a->H;
  // norm1
  const void* xn1 = a->xn1_in;
  if (!xn1) {
    BLK_TRY(dinox_layernorm_fwd(a->x0, a->n1w, a->n1b, a->xn1, a->mean1, a->rstd1, M, D, a->eps, DINOX_BF16, stream));
    xn1 = a->xn1;
  }
  if (!a->qkv) {
    BLK_TRY(dinox_qkv_attention_fwd(xn1, a->wqkv, a->bqkv, a->o, nullptr, a->lse, (int)a->V, (int)a->N, a->heads, D / a->heads, D, stream));
  } else {
    // qkv = xn1 Wqkv^T + b
    dinox_gemm_args g = gemm_args(xn1, a->wqkv, a->qkv, M, 3 * (int64_t)D, D, DINOX_BF16);
    if (a->bqkv) { g.epilogue |= DINOX_EPI_BIAS; g.bias = a->bqkv; }
    BLK_TRY(dinox_gemm(&g, stream));
    BLK_TRY(dinox_attention_fwd(a->qkv, a->o, a->lse, (int)a->V, (int)a->N, a->heads, D / a->heads, DINOX_BF16, stream));
  }
  // x1 = x0 + o Wproj^T + b ; xn2 = norm2(x1)
  if (a->fuse_proj_ln) {
    BLK_TRY(dinox_linear_residual_ln(a->o, a->wproj, a->bproj, a->x0, a->x1, a->n2w, a->n2b, a->eps, a->xn2, DINOX_BF16, a->mean2, a->rstd2, M, D, D, stream));
  } else {
    dinox_gemm_args g = gemm_args(a->o, a->wproj, a->x1, M, D, D, DINOX_F32);
    g.epilogue = DINOX_EPI_RESIDUAL | (a->bproj ? DINOX_EPI_BIAS : 0);
    g.bias = a->bproj;
    g.residual = a->x0;
    BLK_TRY(dinox_gemm(&g, stream));
    BLK_TRY(dinox_layernorm_fwd(a->x1, a->n2w, a->n2b, a->xn2, a->mean2, a->rstd2, M, D, a->eps, DINOX_BF16, stream));
  }
  // act = gelu(xn2 W1^T + b1)  (+ the GELU' side tensor)
  {
    dinox_gemm_args g = gemm_args(a->xn2, a->w1, a->act, M, H, D, DINOX_BF16);
    g.epilogue = DINOX_EPI_GELU | DINOX_EPI_AUXGRAD | (a->b1 ? DINOX_EPI_BIAS : 0);
    g.bias = a->b1;
    g.aux = a->train ? a->pre : nullptr;
    BLK_TRY(dinox_gemm(&g, stream));
  }
  // x2 = x1 + act W2^T + b2 (; yn = LayerNorm(x2))
  if (a->next_g && a->fuse_fc2_ln) {
    BLK_TRY(dinox_linear_residual_ln(a->act, a->w2, a->b2, a->x1, a->x2, a->next_g, a->next_b, a->next_eps, a->yn, a->next_dtype, a->meann, a->rstdn, M, D, H, stream));
  } else {
    dinox_gemm_args g = gemm_args(a->act, a->w2, a->x2, M, D, H, DINOX_F32);
    g.epilogue = DINOX_EPI_RESIDUAL | (a->b2 ? DINOX_EPI_BIAS : 0);
    g.bias = a->b2;
    g.residual = a->x1;
    BLK_TRY(dinox_gemm(&g, stream));
    if (a->next_g) BLK_TRY(dinox_layernorm_fwd(a->x2, a->next_g, a->next_b, a->yn, a->meann, a->rstdn, M, D, a->next_eps, a->next_dtype, stream));
  }
  return 0;
}

extern "C" int dinox_block_backward(const dinox_block_bwd_args* a, void* stream) {
  DX_REQUIRE(a, DINOX_EINVAL, "block_backward: null args");
  DX_REQUIRE(a->V > 0 && a->N > 0 && a->D > 0 && a->H > 0 && a->heads > 0 && a->D % a->heads == 0, DINOX_EINVAL,
             "block_backward: V=%lld N=%lld D=%d H=%d heads=%d", (long long)a->V, (long long)a->N, a->D, a->H, a->heads);
  DX_REQUIRE(a->g && (a->g_lowp || a->g_lowp_buf), DINOX_EINVAL, "block_backward: gradient");
  DX_REQUIRE(a->x0 && a->x1 && a->xn1 && a->xn2 && a->qkv && a->o && a->lse && a->pre && a->act && a->mean1 && a->rstd1 && a->mean2 && a->rstd2, DINOX_EINVAL,
             "block_backward: null saved tensor");
  DX_REQUIRE(a->n1w && a->n2w && a->wqkv_t && a->wproj_t && a->w1_t && a->w2_t, DINOX_EINVAL, "block_backward: null parameter");
  DX_REQUIRE(a->dwqkv && a->dwproj && a->dw1 && a->dw2 && a->dn1w && a->dn1b && a->dn2w && a->dn2b, DINOX_EINVAL, "block_backward: null gradient slice");
  DX_REQUIRE(a->dpre && a->dxn2 && a->d_o && a->dqkv && a->dxn1 && a->g1 && a->g1_lowp && a->g0_lowp && a->attn_ws && a->ln_ws, DINOX_EINVAL,
             "block_backward: null scratch");
  const int64_t M = a->V * a->N;
  const int D = a->D, H = a->H;
  const void* g_op = a->g_lowp;
  if (!g_op) {
    BLK_TRY(dinox_cast_bf16(a->g, a->g_lowp_buf, M * D, stream));
    g_op = a->g_lowp_buf;
  }
  // ---- MLP: x2 = x1 + fc2(gelu(fc1(xn2)))
  {
    dinox_gemm_args g = gemm_args(g_op, a->w2_t, a->dpre, M, H, D, DINOX_BF16);            // dpre = (g W2) o gelu'
    g.epilogue = DINOX_EPI_DGELU | DINOX_EPI_AUXGRAD;
    g.aux = const_cast<void*>(a->pre);
    BLK_TRY(dinox_gemm(&g, stream));
  }
  BLK_TRY(weight_grad(g_op, a->act, a->dw2, a->db2, M, D, H, a->tn_ws, a->tn_ws_bytes, stream));
  const bool fuse_ln = (a->reserved & 1) != 0;                    // dX product + LayerNorm backward in one launch (dinox_linear_ln_bwd)
  if (!fuse_ln) {
    dinox_gemm_args g = gemm_args(a->dpre, a->w1_t, a->dxn2, M, D, H, DINOX_BF16);
    BLK_TRY(dinox_gemm(&g, stream));
  }
  BLK_TRY(weight_grad(a->dpre, a->xn2, a->dw1, a->db1, M, H, D, a->tn_ws, a->tn_ws_bytes, stream));
  // g1 = g + LN2'(dxn2)   (+ its bf16 copy)
  if (fuse_ln)
    BLK_TRY(dinox_linear_ln_bwd(a->dpre, a->w1_t, a->x1, a->n2w, a->mean2, a->rstd2, a->g1, a->g, a->g1_lowp, a->dn2w, a->dn2b, a->ln_ws, M, (int)D,
                                (int)H, 1, stream));
  else
    BLK_TRY(dinox_layernorm_bwd(a->dxn2, a->x1, a->n2w, a->mean2, a->rstd2, a->g1, a->g, a->g1_lowp, a->dn2w, a->dn2b, a->ln_ws, M, D, DINOX_BF16, 1, stream));
  // ---- attention: x1 = x0 + proj(attn(qkv(xn1)))
  {
    dinox_gemm_args g = gemm_args(a->g1_lowp, a->wproj_t, a->d_o, M, D, D, DINOX_BF16);
    BLK_TRY(dinox_gemm(&g, stream));
  }
  BLK_TRY(weight_grad(a->g1_lowp, a->o, a->dwproj, a->dbproj, M, D, D, a->tn_ws, a->tn_ws_bytes, stream));
  BLK_TRY(dinox_attention_bwd(a->d_o, a->qkv, a->o, a->lse, a->dqkv, a->attn_ws, (int)a->V, (int)a->N, a->heads, D / a->heads, DINOX_BF16, stream));
  if (!fuse_ln) {
    dinox_gemm_args g = gemm_args(a->dqkv, a->wqkv_t, a->dxn1, M, D, 3 * (int64_t)D, DINOX_BF16);
    BLK_TRY(dinox_gemm(&g, stream));
  }
  BLK_TRY(weight_grad(a->dqkv, a->xn1, a->dwqkv, a->dbqkv, M, 3 * (int64_t)D, D, a->tn_ws, a->tn_ws_bytes, stream));
  // g0 = g1 + LN1'(dxn1), in place on g1   (+ its bf16 copy)
  if (fuse_ln)
    BLK_TRY(dinox_linear_ln_bwd(a->dqkv, a->wqkv_t, a->x0, a->n1w, a->mean1, a->rstd1, a->g1, a->g1, a->g0_lowp, a->dn1w, a->dn1b, a->ln_ws, M, (int)D,
                                3 * (int)D, 1, stream));
  else
    BLK_TRY(dinox_layernorm_bwd(a->dxn1, a->x0, a->n1w, a->mean1, a->rstd1, a->g1, a->g1, a->g0_lowp, a->dn1w, a->dn1b, a->ln_ws, M, D, DINOX_BF16, 1, stream));
  return 0;
}
