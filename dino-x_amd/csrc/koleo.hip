// koleo.hip -- KoLeo (Kozachenko-Leonenko) entropy regulariser on the student head output.
// Replaces KoLeoLoss.forward of the reference (scripts/phase5_big_run.py:742-773):
//     x^ = F.normalize(x);  d_ij = ||x^_i - x^_j||;  loss = -mean_i log(min_{j != i} d_ij + eps)
// Everything is fp32 (autocast keeps cdist in fp32).  The all-pairs inner products come from dinox_gemm (exact-fp32 MFMA);
// here: row normalisation, the nearest-neighbour search over a row of that product, and the backward pass.
//
// Data parallel: the nearest neighbour is searched over the GLOBAL batch.  Every rank normalises its own rows, all-gathers the
// unit rows (host side, RCCL), searches its rows against all of them and all-gathers the (index, distance) pairs; the backward
// of a local row r then needs only gathered data: its own pair, and every pair (i -> r) that chose r as neighbour.
#include "common.h"

namespace dinox {

constexpr int KL_THREADS = 256;

// One workgroup per row: xh = x / max(||x||, eps), norm = ||x||, sq = ||xh||^2 (1 unless the clamp acted).
__global__ __launch_bounds__(KL_THREADS) void koleo_normalize_kernel(const float* __restrict__ x, float* __restrict__ xh,
                                                                    float* __restrict__ norm, float* __restrict__ sq, int D, float eps) {
  __shared__ float red[16];
  const int64_t r = blockIdx.x;
  const float* xr = x + r * D;
  float a = 0.f;
  for (int d = threadIdx.x; d < D; d += KL_THREADS) a += xr[d] * xr[d];
  a = block_sum(a, red);
  const float n = sqrtf(a), inv = 1.0f / fmaxf(n, eps);
  for (int d = threadIdx.x; d < D; d += KL_THREADS) xh[r * D + d] = xr[d] * inv;
  if (threadIdx.x == 0) {
    norm[r] = n;
    sq[r] = a * inv * inv;
  }
}

// One workgroup per local row i (global index row0 + i).  G[i][j] = xh_i . xh_j over all V_g rows.  Picks
// j* = argmin_{j != self} (sq_i + sq_j - 2 G_ij) (lowest j on ties, like torch.min), then measures the distance on the rows
// themselves: sqrt(sum (xh_i - xh_j*)^2) keeps its digits for close pairs, where the expanded form cancels.
__global__ __launch_bounds__(KL_THREADS) void koleo_nn_kernel(const float* __restrict__ G, int64_t ldg, const float* __restrict__ sq_all,
                                                             const float* __restrict__ xh_all, int row0, int V_g, int D,
                                                             int* __restrict__ idx, float* __restrict__ dist) {
  __shared__ float red[16];
  __shared__ float s_val[KL_THREADS / 64];
  __shared__ int s_idx[KL_THREADS / 64];
  const int i = blockIdx.x, self = row0 + i;
  const float sqi = sq_all[self];
  float best = INFINITY;
  int bj = 0x7fffffff;
  for (int j = threadIdx.x; j < V_g; j += KL_THREADS) {
    if (j == self) continue;
    const float v = sqi + sq_all[j] - 2.0f * G[(int64_t)i * ldg + j];
    if (v < best) {                      // j ascends within a thread: strict < keeps the lowest index
      best = v;
      bj = j;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oj = __shfl_xor(bj, o, 64);
    if (ov < best || (ov == best && oj < bj)) {
      best = ov;
      bj = oj;
    }
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    s_val[w] = best;
    s_idx[w] = bj;
  }
  __syncthreads();
  best = s_val[0];
  bj = s_idx[0];
#pragma unroll
  for (int q = 1; q < KL_THREADS / 64; ++q)
    if (s_val[q] < best || (s_val[q] == best && s_idx[q] < bj)) {
      best = s_val[q];
      bj = s_idx[q];
    }
  float a = 0.f;
  if (bj < V_g) {
    const float* xi = xh_all + (int64_t)self * D;
    const float* xj = xh_all + (int64_t)bj * D;
    for (int d = threadIdx.x; d < D; d += KL_THREADS) {
      const float t = xi[d] - xj[d];
      a += t * t;
    }
  }
  a = block_sum(a, red);
  if (threadIdx.x == 0) {
    idx[i] = bj < V_g ? bj : -1;         // -1: a batch of one row has no neighbour
    dist[i] = bj < V_g ? sqrtf(a) : 1e9f;
  }
}

// One workgroup per local row r (global rg = row0 + r).  With c_i = -gscale / ((d_i + eps) d_i)  (0 where d_i = 0, as cdist's
// backward does):   g = c_rg (xh_r - xh_{j_rg}) + sum_{i : j_i = rg} c_i (xh_r - xh_i)     [own pair + pairs that chose r]
// then through the normalisation:  dx = (g - xh_r (xh_r . g)) / ||x_r||   (dx = g / eps where the clamp acted).
// The pairs that chose r are collected in ascending i (ballot compaction), so the sum order is fixed.
__global__ __launch_bounds__(KL_THREADS) void koleo_bwd_kernel(const float* __restrict__ xh_all, const int* __restrict__ idx_all,
                                                              const float* __restrict__ dist_all, const float* __restrict__ norm_loc,
                                                              int row0, int V_g, int D, float gscale, float eps, float norm_eps,
                                                              float* __restrict__ dx) {
  extern __shared__ int s_list[];        // V_g + 1 ints: rows whose gradient term touches r
  __shared__ float red[16];
  __shared__ int s_wcount[KL_THREADS / 64];
  __shared__ int s_n;
  const int r = blockIdx.x, rg = row0 + r;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  for (int base = 0; base < V_g; base += KL_THREADS) {
    const int i = base + threadIdx.x;
    const bool hit = i < V_g && idx_all[i] == rg;
    const unsigned long long m = __ballot(hit);
    if (lane == 0) s_wcount[w] = __popcll(m);
    __syncthreads();
    int off = s_n;
    for (int q = 0; q < w; ++q) off += s_wcount[q];
    if (hit) s_list[off + __popcll(m & ((1ull << lane) - 1ull))] = i;
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
      for (int q = 0; q < KL_THREADS / 64; ++q) t += s_wcount[q];
      s_n += t;
    }
    __syncthreads();
  }
  const int n_in = s_n;
  const float* xr = xh_all + (int64_t)rg * D;
  const int jr = idx_all[rg];
  const float dr = dist_all[rg];
  const float cr = (jr >= 0 && dr > 0.f) ? -gscale / ((dr + eps) * dr) : 0.f;
  float dot = 0.f;
  for (int d = threadIdx.x; d < D; d += KL_THREADS) {
    const float xv = xr[d];
    float g = 0.f;
    if (cr != 0.f) g += cr * (xv - xh_all[(int64_t)jr * D + d]);
    for (int q = 0; q < n_in; ++q) {
      const int i = s_list[q];
      const float di = dist_all[i];
      if (di > 0.f) g += (gscale / ((di + eps) * di)) * (xh_all[(int64_t)i * D + d] - xv);
    }
    dx[(int64_t)r * D + d] = g;
    dot += g * xv;
  }
  dot = block_sum(dot, red);
  const float n = norm_loc[r];
  if (n >= norm_eps) {
    const float inv = 1.0f / n;
    for (int d = threadIdx.x; d < D; d += KL_THREADS) dx[(int64_t)r * D + d] = (dx[(int64_t)r * D + d] - xr[d] * dot) * inv;
  } else {
    const float inv = 1.0f / norm_eps;
    for (int d = threadIdx.x; d < D; d += KL_THREADS) dx[(int64_t)r * D + d] *= inv;
  }
}

}  // namespace dinox

using namespace dinox;

extern "C" int dinox_koleo_normalize(const float* x, float* xh, float* norm, float* sq, int64_t V, int D, float eps, void* stream) {
  DX_REQUIRE(x && xh && norm && sq, DINOX_EINVAL, "koleo_normalize: null pointer");
  DX_REQUIRE(V > 0 && V <= 0x7fffffff && D > 0, DINOX_EINVAL, "koleo_normalize: V=%lld D=%d", (long long)V, D);
  hipLaunchKernelGGL(koleo_normalize_kernel, dim3((unsigned)V), dim3(KL_THREADS), 0, as_stream(stream), x, xh, norm, sq, D, eps);
  return check_launch("koleo_normalize");
}

extern "C" int dinox_koleo_nn(const float* G, int64_t ldg, const float* sq_all, const float* xh_all, int row0, int V_l, int V_g, int D,
                              int* idx, float* dist, void* stream) {
  DX_REQUIRE(G && sq_all && xh_all && idx && dist, DINOX_EINVAL, "koleo_nn: null pointer");
  DX_REQUIRE(V_l > 0 && V_g > 0 && row0 >= 0 && row0 + V_l <= V_g && ldg >= V_g && D > 0, DINOX_EINVAL,
             "koleo_nn: row0=%d V_l=%d V_g=%d ldg=%lld D=%d", row0, V_l, V_g, (long long)ldg, D);
  hipLaunchKernelGGL(koleo_nn_kernel, dim3((unsigned)V_l), dim3(KL_THREADS), 0, as_stream(stream), G, ldg, sq_all, xh_all, row0, V_g, D,
                     idx, dist);
  return check_launch("koleo_nn");
}

// loss = -mean_i log(dist_i + eps) over this rank's rows: one block, fixed order
__global__ __launch_bounds__(256) void koleo_loss_kernel(const float* __restrict__ dist, int V, float eps, float* __restrict__ out) {
  __shared__ float red[16];
  float a = 0.f;
  for (int i = threadIdx.x; i < V; i += 256) a += logf(dist[i] + eps);
  a = dinox::block_sum(a, red);
  if (threadIdx.x == 0) out[0] = -a / (float)V;
}

extern "C" int dinox_koleo_loss(const float* dist, int V, float eps, float* loss, void* stream) {
  DX_REQUIRE(dist && loss, DINOX_EINVAL, "koleo_loss: null pointer");
  DX_REQUIRE(V > 0, DINOX_EINVAL, "koleo_loss: V=%d", V);
  hipLaunchKernelGGL(koleo_loss_kernel, dim3(1), dim3(256), 0, as_stream(stream), dist, V, eps, loss);
  return check_launch("koleo_loss");
}

extern "C" int dinox_koleo_bwd(const float* xh_all, const int* idx_all, const float* dist_all, const float* norm_loc, int row0, int V_l,
                               int V_g, int D, float gscale, float eps, float norm_eps, float* dx, void* stream) {
  DX_REQUIRE(xh_all && idx_all && dist_all && norm_loc && dx, DINOX_EINVAL, "koleo_bwd: null pointer");
  DX_REQUIRE(V_l > 0 && V_g > 0 && row0 >= 0 && row0 + V_l <= V_g && D > 0, DINOX_EINVAL, "koleo_bwd: row0=%d V_l=%d V_g=%d D=%d", row0,
             V_l, V_g, D);
  const size_t lds = ((size_t)V_g + 1) * sizeof(int);
  DX_REQUIRE(lds <= 60 * 1024, DINOX_EUNSUPPORTED, "koleo_bwd: global batch of %d rows exceeds the neighbour list (15359 rows)", V_g);
  hipLaunchKernelGGL(koleo_bwd_kernel, dim3((unsigned)V_l), dim3(KL_THREADS), lds, as_stream(stream), xh_all, idx_all, dist_all, norm_loc,
                     row0, V_g, D, gscale, eps, norm_eps, dx);
  return check_launch("koleo_bwd");
}
