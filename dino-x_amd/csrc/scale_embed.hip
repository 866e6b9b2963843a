// scale_embed.hip -- ScaleEmbedding: Linear(3,h) -> GELU -> Linear(h,D) -> LayerNorm(D), fp32.
// Replaces reference zoo/arch.py:119-140.  V rows only (one per image), so the whole MLP of a row is
// one workgroup and the layer is a single launch (fwd) / two launches (bwd) instead of 4 + 8 ATen ops.
#include "common.h"

namespace dinox {

constexpr int SE_THREADS = 256;

__global__ __launch_bounds__(SE_THREADS) void scale_embed_fwd_kernel(
    const float* __restrict__ sp, const float* __restrict__ w0, const float* __restrict__ b0, const float* __restrict__ w2,
    const float* __restrict__ b2, const float* __restrict__ lnw, const float* __restrict__ lnb, float* __restrict__ out,
    float* __restrict__ hpre, float* __restrict__ e, float* __restrict__ mean, float* __restrict__ rstd, int h, int D,
    float eps) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // a[h] | red[16]
  float* a = lds;
  float* red = lds + h;
  const int v = blockIdx.x, t = threadIdx.x;
  const float s0 = sp[v * 3 + 0], s1 = sp[v * 3 + 1], s2 = sp[v * 3 + 2];
  for (int j = t; j < h; j += SE_THREADS) {
    const float z = b0[j] + w0[j * 3 + 0] * s0 + w0[j * 3 + 1] * s1 + w0[j * 3 + 2] * s2;
    hpre[(int64_t)v * h + j] = z;
    a[j] = gelu_erf(z);
  }
  __syncthreads();
  float part = 0.f;
  for (int dd = t; dd < D; dd += SE_THREADS) {
    float z = b2[dd];
    const float* wr = w2 + (int64_t)dd * h;
    for (int j = 0; j < h; ++j) z += wr[j] * a[j];
    e[(int64_t)v * D + dd] = z;
    part += z;
  }
  const float mu = block_sum(part, red) / (float)D;
  float q = 0.f;
  for (int dd = t; dd < D; dd += SE_THREADS) {
    const float c = e[(int64_t)v * D + dd] - mu;  // own writes: same thread wrote these elements
    q += c * c;
  }
  const float rs = rsqrtf(block_sum(q, red) / (float)D + eps);
  if (t == 0) {
    mean[v] = mu;
    rstd[v] = rs;
  }
  for (int dd = t; dd < D; dd += SE_THREADS) out[(int64_t)v * D + dd] = (e[(int64_t)v * D + dd] - mu) * rs * lnw[dd] + lnb[dd];
}

// per-row: de (LN backward), dhpre, dspacing
__global__ __launch_bounds__(SE_THREADS) void scale_embed_bwd_rows(
    const float* __restrict__ dout, const float* __restrict__ w0, const float* __restrict__ w2, const float* __restrict__ lnw,
    const float* __restrict__ hpre, const float* __restrict__ e, const float* __restrict__ mean, const float* __restrict__ rstd,
    float* __restrict__ de, float* __restrict__ dhpre, float* __restrict__ hact, float* __restrict__ dspacing, int h, int D) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // de_row[D] | dh[h] | red[16]
  float* der = lds;
  float* dh = lds + D;
  float* red = dh + h;
  const int v = blockIdx.x, t = threadIdx.x;
  const float mu = mean[v], rs = rstd[v];
  float s1 = 0.f, s2 = 0.f;
  for (int dd = t; dd < D; dd += SE_THREADS) {
    const float g = dout[(int64_t)v * D + dd] * lnw[dd];
    const float xh = (e[(int64_t)v * D + dd] - mu) * rs;
    s1 += g;
    s2 += g * xh;
  }
  const float m1 = block_sum(s1, red) / (float)D;
  const float m2 = block_sum(s2, red) / (float)D;
  for (int dd = t; dd < D; dd += SE_THREADS) {
    const float g = dout[(int64_t)v * D + dd] * lnw[dd];
    const float xh = (e[(int64_t)v * D + dd] - mu) * rs;
    const float d = rs * (g - m1 - xh * m2);
    der[dd] = d;
    de[(int64_t)v * D + dd] = d;
  }
  __syncthreads();
  for (int j = t; j < h; j += SE_THREADS) {
    float s = 0.f;
    for (int dd = 0; dd < D; ++dd) s += der[dd] * w2[(int64_t)dd * h + j];
    const float hp = hpre[(int64_t)v * h + j];
    const float d = s * gelu_erf_grad(hp);
    dh[j] = d;
    dhpre[(int64_t)v * h + j] = d;
    hact[(int64_t)v * h + j] = gelu_erf(hp);            // once per element here, not V times per weight in the parameter kernel
  }
  __syncthreads();
  if (dspacing && t < 3) {
    float s = 0.f;
    for (int j = 0; j < h; ++j) s += dh[j] * w0[j * 3 + t];
    dspacing[v * 3 + t] = s;
  }
}

// parameter gradients: one thread per parameter element, loop over the V rows -- eight rows per trip with eight independent partial
// sums (fixed order: deterministic), so eight loads are in flight per thread (one dependent add per row: 178 us for 74 KFLOP at V = 512).
// index space: [0, D*h) dw2 | +D db2 | +3h dw0 | +h db0 | +D dlnw | +D dlnb
template <typename F>
__device__ __forceinline__ float se_sum_rows(int V, F term) {
  float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int v = 0;
  for (; v + 8 <= V; v += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) p[u] += term(v + u);
  }
  for (; v < V; ++v) p[0] += term(v);
  return ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
}

__global__ __launch_bounds__(256) void scale_embed_bwd_params(
    const float* __restrict__ dout, const float* __restrict__ sp, const float* __restrict__ hact, const float* __restrict__ e,
    const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ de, const float* __restrict__ dhpre,
    float* __restrict__ dw0, float* __restrict__ db0, float* __restrict__ dw2, float* __restrict__ db2, float* __restrict__ dlnw,
    float* __restrict__ dlnb, int V, int h, int D) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n_dw2 = (int64_t)D * h;
  if (idx < n_dw2) {
    const int dd = (int)(idx / h), j = (int)(idx % h);
    dw2[idx] = se_sum_rows(V, [&](int v) { return de[(int64_t)v * D + dd] * hact[(int64_t)v * h + j]; });
    return;
  }
  idx -= n_dw2;
  if (idx < D) {
    db2[idx] = se_sum_rows(V, [&](int v) { return de[(int64_t)v * D + idx]; });
    return;
  }
  idx -= D;
  if (idx < 3 * h) {
    const int j = (int)(idx / 3), c = (int)(idx % 3);
    dw0[idx] = se_sum_rows(V, [&](int v) { return dhpre[(int64_t)v * h + j] * sp[v * 3 + c]; });
    return;
  }
  idx -= 3 * h;
  if (idx < h) {
    db0[idx] = se_sum_rows(V, [&](int v) { return dhpre[(int64_t)v * h + idx]; });
    return;
  }
  idx -= h;
  if (idx < D) {
    dlnw[idx] = se_sum_rows(V, [&](int v) { return dout[(int64_t)v * D + idx] * (e[(int64_t)v * D + idx] - mean[v]) * rstd[v]; });
    return;
  }
  idx -= D;
  if (idx < D) dlnb[idx] = se_sum_rows(V, [&](int v) { return dout[(int64_t)v * D + idx]; });
}

}  // namespace dinox

using namespace dinox;

extern "C" int dinox_scale_embed_fwd(const float* spacing, const float* w0, const float* b0, const float* w2,
                                     const float* b2, const float* lnw, const float* lnb, float* out, float* hpre,
                                     float* e, float* mean, float* rstd, int V, int h, int D, float eps, void* stream) {
  DX_REQUIRE(spacing && w0 && b0 && w2 && b2 && lnw && lnb && out && hpre && e && mean && rstd, DINOX_EINVAL, "scale_embed_fwd: null pointer");
  DX_REQUIRE(V > 0 && h > 0 && D > 0 && h <= 8192, DINOX_EINVAL, "scale_embed_fwd: V=%d h=%d D=%d", V, h, D);
  hipLaunchKernelGGL(scale_embed_fwd_kernel, dim3(V), dim3(SE_THREADS), (size_t)(h + 16) * sizeof(float), as_stream(stream),
                     spacing, w0, b0, w2, b2, lnw, lnb, out, hpre, e, mean, rstd, h, D, eps);
  return check_launch("scale_embed_fwd");
}

extern "C" int64_t dinox_scale_embed_bwd_ws_bytes(int V, int h, int D) {
  if (V <= 0 || h <= 0 || D <= 0) return 0;
  return (int64_t)V * (D + 2 * h) * (int64_t)sizeof(float);       // de [V][D] | dhpre [V][h] | gelu(hpre) [V][h]
}

extern "C" int dinox_scale_embed_bwd(const float* dout, const float* spacing, const float* w0, const float* w2,
                                     const float* lnw, const float* hpre, const float* e, const float* mean,
                                     const float* rstd, float* dw0, float* db0, float* dw2, float* db2, float* dlnw,
                                     float* dlnb, float* dspacing, void* ws, int V, int h, int D, void* stream) {
  DX_REQUIRE(dout && spacing && w0 && w2 && lnw && hpre && e && mean && rstd && dw0 && db0 && dw2 && db2 && dlnw && dlnb && ws,
             DINOX_EINVAL, "scale_embed_bwd: null pointer");
  DX_REQUIRE(V > 0 && h > 0 && D > 0 && (D + h + 16) * 4 <= 64 * 1024, DINOX_EINVAL, "scale_embed_bwd: V=%d h=%d D=%d", V, h, D);
  hipStream_t st = as_stream(stream);
  float* de = (float*)ws;
  float* dhp = de + (int64_t)V * D;
  float* hact = dhp + (int64_t)V * h;
  hipLaunchKernelGGL(scale_embed_bwd_rows, dim3(V), dim3(SE_THREADS), (size_t)(D + h + 16) * sizeof(float), st, dout, w0, w2,
                     lnw, hpre, e, mean, rstd, de, dhp, hact, dspacing, h, D);
  int rc = check_launch("scale_embed_bwd_rows");
  if (rc) return rc;
  const int64_t total = (int64_t)D * h + D + 3 * h + h + D + D;
  hipLaunchKernelGGL(scale_embed_bwd_params, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st, dout, spacing, hact, e,
                     mean, rstd, de, dhp, dw0, db0, dw2, db2, dlnw, dlnb, V, h, D);
  return check_launch("scale_embed_bwd_params");
}
