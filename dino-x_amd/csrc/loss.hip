// loss.hip -- DINO centring/sharpening cross-entropy and the fused pieces of the Gram-anchoring loss.
// Replaces DINOLoss.forward/update_center and compute_gram_anchoring_loss of the reference
// (scripts/phase5_big_run.py:686-720, 723-739).  Everything here is fp32 (autocast keeps softmax,
// log_softmax, normalize and mse_loss in fp32 too); the Gram products go through dinox_gemm.
#include "common.h"

namespace dinox {

constexpr int CE_THREADS = 256;

// One workgroup per student row i.  Teacher row pair(i) = (i+B) mod 2B.  log-sum-exp form throughout
// (the reference hit NaNs with a separate softmax+log, phase5_big_run.py:1845-1846).
__global__ __launch_bounds__(CE_THREADS) void dino_ce_kernel(const float* __restrict__ s, const float* __restrict__ t,
                                                             const float* __restrict__ center, float inv_ts, float inv_tt,
                                                             float gscale, float* __restrict__ ds,
                                                             float* __restrict__ row_loss, int rows, int K) {
  __shared__ float red[16];
  const int i = blockIdx.x, B = rows / 2;
  const int pi = (i + B) % rows;
  const float* sr = s + (int64_t)i * K;
  const float* tr = t + (int64_t)pi * K;
  float ms = -INFINITY, mt = -INFINITY;
  for (int k = threadIdx.x; k < K; k += CE_THREADS) {
    ms = fmaxf(ms, sr[k] * inv_ts);
    mt = fmaxf(mt, (tr[k] - center[k]) * inv_tt);
  }
  ms = block_max(ms, red);
  mt = block_max(mt, red);
  float ss = 0.f, st = 0.f;
  for (int k = threadIdx.x; k < K; k += CE_THREADS) {
    ss += expf(sr[k] * inv_ts - ms);
    st += expf((tr[k] - center[k]) * inv_tt - mt);
  }
  ss = block_sum(ss, red);
  st = block_sum(st, red);
  const float log_ss = logf(ss), inv_ss = 1.0f / ss, inv_st = 1.0f / st;
  float acc = 0.f;
  for (int k = threadIdx.x; k < K; k += CE_THREADS) {
    const float zs = sr[k] * inv_ts - ms;
    const float tp = expf((tr[k] - center[k]) * inv_tt - mt) * inv_st;
    acc -= tp * (zs - log_ss);
    if (ds) ds[(int64_t)i * K + k] = gscale * (expf(zs) * inv_ss - tp) * inv_ts / (float)rows;
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) row_loss[i] = acc;
}

// Multi-crop form (extension, not in the reference: DINO paper Alg. 1 with local crops).  Student rows are view-major
// [(G+L) views][B samples], the first G views are the global ones the teacher also saw; teacher rows [G][B].
// One workgroup per student row (v, b):  row_loss = sum_{q < G, q != v} -sum_k p_q[k] (z[k] - lse),  z = s/ts,
// p_q = softmax((t[q][b] - c)/tt);   ds = gscale/(B*terms) * (n_q softmax(z) - sum_q p_q)/ts,  n_q = #{q != v}.
// G = 2, L = 0 is dino_ce_kernel.  The teacher's max / sum-exp per row come from dino_teacher_stats_kernel.
__global__ __launch_bounds__(CE_THREADS) void dino_teacher_stats_kernel(const float* __restrict__ t, const float* __restrict__ center,
                                                                        float inv_tt, float* __restrict__ tmax, float* __restrict__ tinv, int K) {
  __shared__ float red[16];
  const int i = blockIdx.x;
  const float* tr = t + (int64_t)i * K;
  float m = -INFINITY;
  for (int k = threadIdx.x; k < K; k += CE_THREADS) m = fmaxf(m, (tr[k] - center[k]) * inv_tt);
  m = block_max(m, red);
  float a = 0.f;
  for (int k = threadIdx.x; k < K; k += CE_THREADS) a += expf((tr[k] - center[k]) * inv_tt - m);
  a = block_sum(a, red);
  if (threadIdx.x == 0) {
    tmax[i] = m;
    tinv[i] = 1.0f / a;
  }
}

__global__ __launch_bounds__(CE_THREADS) void dino_ce_multi_kernel(const float* __restrict__ s, const float* __restrict__ t,
                                                                   const float* __restrict__ center, const float* __restrict__ tmax,
                                                                   const float* __restrict__ tinv, float inv_ts, float inv_tt, float gscale,
                                                                   float* __restrict__ ds, float* __restrict__ row_loss, int B, int G, int K) {
  __shared__ float red[16];
  const int i = blockIdx.x, v = i / B, b = i % B;
  const float* sr = s + (int64_t)i * K;
  float ms = -INFINITY;
  for (int k = threadIdx.x; k < K; k += CE_THREADS) ms = fmaxf(ms, sr[k] * inv_ts);
  ms = block_max(ms, red);
  float ss = 0.f;
  for (int k = threadIdx.x; k < K; k += CE_THREADS) ss += expf(sr[k] * inv_ts - ms);
  ss = block_sum(ss, red);
  const float log_ss = logf(ss), inv_ss = 1.0f / ss;
  const int nq = v < G ? G - 1 : G;
  float acc = 0.f;
  for (int k = threadIdx.x; k < K; k += CE_THREADS) {
    const float zs = sr[k] * inv_ts - ms;
    float tp = 0.f;
    for (int q = 0; q < G; ++q) {
      if (q == v) continue;
      const int tr = q * B + b;
      tp += expf((t[(int64_t)tr * K + k] - center[k]) * inv_tt - tmax[tr]) * tinv[tr];
    }
    acc -= tp * (zs - log_ss);
    if (ds) ds[(int64_t)i * K + k] = gscale * ((float)nq * expf(zs) * inv_ss - tp) * inv_ts;
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) row_loss[i] = acc;
}

// loss[0] = scale * sum(x[0..n))  -- single workgroup, deterministic order.
__global__ __launch_bounds__(256) void final_sum_kernel(const float* __restrict__ x, int n, float scale, float* __restrict__ out) {
  __shared__ float red[16];
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) a += x[i];
  a = block_sum(a, red);
  if (threadIdx.x == 0) out[0] = a * scale;
}

// A workgroup owns 64 float4 column groups; its four 64-thread slices sum interleaved quarters of the rows (4 independent
// float4 loads in flight per thread) and meet in LDS in a fixed order (deterministic).
__global__ __launch_bounds__(256) void colmean_kernel(const float* __restrict__ t, float* __restrict__ out, int rows, int K) {
  __shared__ float4 part[4][64];
  const int K4 = K / 4, c4 = blockIdx.x * 64 + (threadIdx.x & 63), slice = threadIdx.x >> 6;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c4 < K4) {
    const float* base = t + c4 * 4;
    int r = slice;
    for (; r + 12 < rows; r += 16) {
      const float4 a = *reinterpret_cast<const float4*>(base + (int64_t)r * K), b = *reinterpret_cast<const float4*>(base + (int64_t)(r + 4) * K);
      const float4 c = *reinterpret_cast<const float4*>(base + (int64_t)(r + 8) * K), d = *reinterpret_cast<const float4*>(base + (int64_t)(r + 12) * K);
      s.x += (a.x + b.x) + (c.x + d.x); s.y += (a.y + b.y) + (c.y + d.y); s.z += (a.z + b.z) + (c.z + d.z); s.w += (a.w + b.w) + (c.w + d.w);
    }
    for (; r < rows; r += 4) {
      const float4 a = *reinterpret_cast<const float4*>(base + (int64_t)r * K);
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
  }
  part[slice][threadIdx.x & 63] = s;
  __syncthreads();
  if (slice == 0 && c4 < K4) {
    const float inv = 1.0f / (float)rows;
    const float4 p0 = part[0][threadIdx.x], p1 = part[1][threadIdx.x], p2 = part[2][threadIdx.x], p3 = part[3][threadIdx.x];
    *reinterpret_cast<float4*>(out + c4 * 4) = make_float4(((p0.x + p1.x) + (p2.x + p3.x)) * inv, ((p0.y + p1.y) + (p2.y + p3.y)) * inv,
                                                           ((p0.z + p1.z) + (p2.z + p3.z)) * inv, ((p0.w + p1.w) + (p2.w + p3.w)) * inv);
  }
}
// any K / alignment
__global__ __launch_bounds__(256) void colmean_scalar_kernel(const float* __restrict__ t, float* __restrict__ out, int rows, int K) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  float a = 0.f;
  for (int r = 0; r < rows; ++r) a += t[(int64_t)r * K + k];
  out[k] = a / (float)rows;
}

__global__ __launch_bounds__(256) void center_ema_kernel(float* __restrict__ c, const float* __restrict__ m, float mom, int K) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < K) c[k] = c[k] * mom + m[k] * (1.0f - mom);
}

// One wave per (image v, token t>=1).  cat row = [s_hat | t_hat], catneg row = [s_hat | -t_hat].
template <int DT>
__global__ __launch_bounds__(256) void gram_normalize_kernel(const float* __restrict__ sf, const float* __restrict__ tf,
                                                             void* __restrict__ cat, void* __restrict__ catneg,
                                                             void* __restrict__ shat, float* __restrict__ snorm, int V,
                                                             int N, int D) {
  const int lane = threadIdx.x & 63;
  const int T = N - 1;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)V * T) return;
  const int64_t v = row / T;
  const int tt = (int)(row % T);
  const float* xs = sf + (v * N + 1 + tt) * D;
  const float* xt = tf + (v * N + 1 + tt) * D;
  float qs = 0.f, qt = 0.f;
  for (int d = lane; d < D; d += 64) {
    qs += xs[d] * xs[d];
    qt += xt[d] * xt[d];
  }
  const float ns = fmaxf(sqrtf(wave_sum(qs)), 1e-12f), nt = fmaxf(sqrtf(wave_sum(qt)), 1e-12f);
  if (lane == 0) snorm[row] = ns;
  const float is = 1.0f / ns, it = 1.0f / nt;
  for (int d = lane; d < D; d += 64) {
    const float a = xs[d] * is, b = xt[d] * it;
    elem<DT>::st(cat, row * 2 * D + d, a);
    elem<DT>::st(cat, row * 2 * D + D + d, b);
    elem<DT>::st(catneg, row * 2 * D + d, a);
    elem<DT>::st(catneg, row * 2 * D + D + d, -b);
    elem<DT>::st(shat, row * D + d, a);
  }
}

__global__ __launch_bounds__(256) void sqsum_partial_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ ws) {
  __shared__ float red[16];
  float a = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) a += x[i] * x[i];
  a = block_sum(a, red);
  if (threadIdx.x == 0) ws[blockIdx.x] = a;
}

template <int DT>
__global__ __launch_bounds__(256) void gram_normalize_bwd_kernel(const float* __restrict__ dxh, const void* __restrict__ shat,
                                                                 const float* __restrict__ snorm, float* __restrict__ dfeats,
                                                                 int V, int N, int D, int accumulate) {
  const int lane = threadIdx.x & 63;
  const int T = N - 1;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)V * T) return;
  const int64_t v = row / T;
  const int tt = (int)(row % T);
  const float nrm = snorm[row];
  const bool clamped = !(nrm > 1e-12f);
  float proj = 0.f;
  for (int d = lane; d < D; d += 64) proj += elem<DT>::ld(shat, row * D + d) * dxh[row * D + d];
  proj = wave_sum(proj);
  float* dst = dfeats + (v * N + 1 + tt) * D;
  const float inv = 1.0f / nrm;
  for (int d = lane; d < D; d += 64) {
    const float g = dxh[row * D + d];
    float o = clamped ? g * 1e12f : (g - elem<DT>::ld(shat, row * D + d) * proj) * inv;
    if (accumulate) o += dst[d];
    dst[d] = o;
  }
}

}  // namespace dinox

using namespace dinox;

extern "C" int dinox_dino_ce(const float* s, const float* t, const float* center, float student_temp, float teacher_temp,
                             float grad_scale, float* loss, float* ds, float* row_loss, int rows2B, int K, void* stream) {
  DX_REQUIRE(s && t && center && loss && row_loss, DINOX_EINVAL, "dino_ce: null pointer");
  DX_REQUIRE(rows2B >= 2 && rows2B % 2 == 0 && K > 0, DINOX_EINVAL, "dino_ce: rows=%d (must be even: [view1; view2]) K=%d", rows2B, K);
  DX_REQUIRE(student_temp > 0.f && teacher_temp > 0.f, DINOX_EINVAL, "dino_ce: temperatures must be > 0");
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(dino_ce_kernel, dim3(rows2B), dim3(CE_THREADS), 0, st, s, t, center, 1.0f / student_temp,
                     1.0f / teacher_temp, grad_scale, ds, row_loss, rows2B, K);
  int rc = check_launch("dino_ce");
  if (rc) return rc;
  hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, st, row_loss, rows2B, 1.0f / (float)rows2B, loss);
  return check_launch("dino_ce_sum");
}

extern "C" int dinox_dino_ce_multi(const float* s, const float* t, const float* center, float student_temp, float teacher_temp,
                                   float grad_scale, float* loss, float* ds, float* ws, int B, int n_global, int n_views, int K,
                                   void* stream) {
  DX_REQUIRE(s && t && center && loss && ws, DINOX_EINVAL, "dino_ce_multi: null pointer");
  DX_REQUIRE(B >= 1 && n_global >= 1 && n_views >= n_global && n_views >= 2 && K > 0, DINOX_EINVAL,
             "dino_ce_multi: B=%d global=%d views=%d K=%d", B, n_global, n_views, K);
  DX_REQUIRE(student_temp > 0.f && teacher_temp > 0.f, DINOX_EINVAL, "dino_ce_multi: temperatures must be > 0");
  DX_REQUIRE((int64_t)n_views * B <= 0x7fffffff, DINOX_EINVAL, "dino_ce_multi: too many rows");
  hipStream_t st = as_stream(stream);
  const int srows = n_views * B, trows = n_global * B;
  const int terms = n_global * (n_views - 1);
  float* row_loss = ws;                   // [srows]
  float* tmax = ws + srows;               // [trows]
  float* tinv = tmax + trows;             // [trows]
  hipLaunchKernelGGL(dino_teacher_stats_kernel, dim3(trows), dim3(CE_THREADS), 0, st, t, center, 1.0f / teacher_temp, tmax, tinv, K);
  int rc = check_launch("dino_teacher_stats");
  if (rc) return rc;
  const float norm = 1.0f / ((float)B * (float)terms);
  hipLaunchKernelGGL(dino_ce_multi_kernel, dim3(srows), dim3(CE_THREADS), 0, st, s, t, center, tmax, tinv, 1.0f / student_temp,
                     1.0f / teacher_temp, grad_scale * norm, ds, row_loss, B, n_global, K);
  rc = check_launch("dino_ce_multi");
  if (rc) return rc;
  hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, st, row_loss, srows, norm, loss);
  return check_launch("dino_ce_multi_sum");
}

extern "C" int dinox_colmean(const float* t, float* out, int rows, int K, void* stream) {
  DX_REQUIRE(t && out && rows > 0 && K > 0, DINOX_EINVAL, "colmean: bad arguments");
  if (K % 4 == 0 && (((uintptr_t)t | (uintptr_t)out) & 15) == 0)
    hipLaunchKernelGGL(colmean_kernel, dim3((unsigned)ceil_div(K / 4, 64)), dim3(256), 0, as_stream(stream), t, out, rows, K);
  else
    hipLaunchKernelGGL(colmean_scalar_kernel, dim3((unsigned)ceil_div(K, 256)), dim3(256), 0, as_stream(stream), t, out, rows, K);
  return check_launch("colmean");
}

extern "C" int dinox_center_ema(float* center, const float* batch_mean, float momentum, int K, void* stream) {
  DX_REQUIRE(center && batch_mean && K > 0, DINOX_EINVAL, "center_ema: bad arguments");
  hipLaunchKernelGGL(center_ema_kernel, dim3((unsigned)ceil_div(K, 256)), dim3(256), 0, as_stream(stream), center, batch_mean, momentum, K);
  return check_launch("center_ema");
}

extern "C" int dinox_gram_normalize(const float* sfeats, const float* tfeats, void* cat, void* catneg, void* shat,
                                    float* snorm, int V, int N, int D, int out_dtype, void* stream) {
  DX_REQUIRE(sfeats && tfeats && cat && catneg && shat && snorm, DINOX_EINVAL, "gram_normalize: null pointer");
  DX_REQUIRE(V > 0 && N > 1 && D > 0, DINOX_EINVAL, "gram_normalize: V=%d N=%d D=%d", V, N, D);
  DX_REQUIRE(out_dtype == DINOX_F32 || out_dtype == DINOX_BF16, DINOX_EINVAL, "gram_normalize: dtype %d", out_dtype);
  const unsigned blocks = (unsigned)ceil_div((int64_t)V * (N - 1), 4);
  if (out_dtype == DINOX_F32)
    hipLaunchKernelGGL((gram_normalize_kernel<DINOX_F32>), dim3(blocks), dim3(256), 0, as_stream(stream), sfeats, tfeats, cat, catneg, shat, snorm, V, N, D);
  else
    hipLaunchKernelGGL((gram_normalize_kernel<DINOX_BF16>), dim3(blocks), dim3(256), 0, as_stream(stream), sfeats, tfeats, cat, catneg, shat, snorm, V, N, D);
  return check_launch("gram_normalize");
}

extern "C" int dinox_sqsum(const float* x, int64_t n, float scale, float* loss, float* ws, void* stream) {
  DX_REQUIRE(x && loss && ws && n > 0, DINOX_EINVAL, "sqsum: bad arguments");
  int64_t blocks = ceil_div(n, 256 * 8);
  if (blocks > 1024) blocks = 1024;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(sqsum_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, n, ws);
  hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, st, ws, (int)blocks, scale, loss);
  return check_launch("sqsum");
}

extern "C" int dinox_gram_normalize_bwd(const float* dxh, const void* shat, const float* snorm, const float* sfeats,
                                        float* dfeats, int V, int N, int D, int shat_dtype, int accumulate, void* stream) {
  (void)sfeats;
  DX_REQUIRE(dxh && shat && snorm && dfeats, DINOX_EINVAL, "gram_normalize_bwd: null pointer");
  DX_REQUIRE(V > 0 && N > 1 && D > 0, DINOX_EINVAL, "gram_normalize_bwd: V=%d N=%d D=%d", V, N, D);
  DX_REQUIRE(shat_dtype == DINOX_F32 || shat_dtype == DINOX_BF16, DINOX_EINVAL, "gram_normalize_bwd: dtype %d", shat_dtype);
  const unsigned blocks = (unsigned)ceil_div((int64_t)V * (N - 1), 4);
  if (shat_dtype == DINOX_F32)
    hipLaunchKernelGGL((gram_normalize_bwd_kernel<DINOX_F32>), dim3(blocks), dim3(256), 0, as_stream(stream), dxh, shat, snorm, dfeats, V, N, D, accumulate);
  else
    hipLaunchKernelGGL((gram_normalize_bwd_kernel<DINOX_BF16>), dim3(blocks), dim3(256), 0, as_stream(stream), dxh, shat, snorm, dfeats, V, N, D, accumulate);
  return check_launch("gram_normalize_bwd");
}
