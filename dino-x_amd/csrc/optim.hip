// optim.hip -- fused optimiser tail over flat fp32 arenas, and the bf16 weight casts.
// Replaces (a) the 155-161 blocking .norm(2).item() calls of the grad-norm loop
// (scripts/phase5_big_run.py:1784-1789), (b) torch.optim.AdamW.step (:1794) and (c) the per-parameter
// EMA teacher loop (:1799-1802) with one streaming pass: 5 arenas read, 4 written, 36 B per parameter.
#include "common.h"

namespace dinox {

__global__ __launch_bounds__(256) void adamw_ema_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, float* __restrict__ teacher, int64_t n,
                                                        float lr, float wd, float b1, float b2, float eps, float inv_bc1,
                                                        float inv_sqrt_bc2, float ema, float gscale, float* __restrict__ ws,
                                                        const float* __restrict__ hyper) {
  __shared__ float red[16];
  if (hyper) {                 // per-step scalars from device memory: the launch can be replayed from a captured hipGraph
    lr = hyper[0];
    inv_bc1 = hyper[1];
    inv_sqrt_bc2 = hyper[2];
  }
  float sq = 0.f;
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    float* P = &pp.x; float* G = &gg.x; float* M = &mm.x; float* Vv = &vv.x;
    float4 tt = teacher ? reinterpret_cast<float4*>(teacher)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    float* T = &tt.x;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float gr = G[c] * gscale;
      sq += gr * gr;
      float w = P[c] * (1.0f - lr * wd);
      M[c] = b1 * M[c] + (1.0f - b1) * gr;
      Vv[c] = b2 * Vv[c] + (1.0f - b2) * gr * gr;
      const float denom = sqrtf(Vv[c]) * inv_sqrt_bc2 + eps;
      w -= lr * inv_bc1 * (M[c] / denom);
      P[c] = w;
      T[c] = ema * T[c] + (1.0f - ema) * w;
    }
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
    if (teacher) reinterpret_cast<float4*>(teacher)[i] = tt;
  }
  // tail (n % 4) handled by the first block's first threads
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    const float gr = g[i] * gscale;
    sq += gr * gr;
    float w = p[i] * (1.0f - lr * wd);
    const float mn = b1 * m[i] + (1.0f - b1) * gr, vn = b2 * v[i] + (1.0f - b2) * gr * gr;
    m[i] = mn;
    v[i] = vn;
    w -= lr * inv_bc1 * (mn / (sqrtf(vn) * inv_sqrt_bc2 + eps));
    p[i] = w;
    if (teacher) teacher[i] = ema * teacher[i] + (1.0f - ema) * w;
  }
  sq = block_sum(sq, red);
  if (threadIdx.x == 0) ws[blockIdx.x] = sq;
}

__global__ __launch_bounds__(256) void sumsq_partial(const float* __restrict__ x, int64_t n, float* __restrict__ ws) {
  __shared__ float red[16];
  float a = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) a += x[i] * x[i];
  a = block_sum(a, red);
  if (threadIdx.x == 0) ws[blockIdx.x] = a;
}

__global__ __launch_bounds__(256) void sum_parts(const float* __restrict__ ws, int n, float* __restrict__ out) {
  __shared__ float red[16];
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) a += ws[i];
  a = block_sum(a, red);
  if (threadIdx.x == 0) out[0] = a;
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = f32_to_bf16(src[i]);
}

// dst[c][r] = bf16(src[r][c]) through a 32x33 LDS tile (coalesced on both sides).
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int R, int C) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int j = ty; j < 32; j += 8) {
    const int r = r0 + j, c = c0 + tx;
    tile[j][tx] = (r < R && c < C) ? src[(int64_t)r * C + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, r = r0 + tx;
    if (c < C && r < R) dst[(int64_t)c * R + r] = f32_to_bf16(tile[tx][j]);
  }
}

// The same for every matrix of a parameter arena in ONE launch.  table[i] = {element offset in both arenas, R, C, first tile};
// a workgroup finds its matrix by bisection over the first-tile column and transposes one 32x32 tile of it.
__global__ __launch_bounds__(256) void cast_transpose_multi_kernel(const float* __restrict__ src_base, bf16_t* __restrict__ dst_base,
                                                                   const int64_t* __restrict__ table, int n_mats) {
  __shared__ float tile[32][33];
  const int64_t bid = blockIdx.x;
  int lo = 0, hi = n_mats - 1;
  while (lo < hi) {                                   // last i with table[i].first_tile <= bid
    const int mid = (lo + hi + 1) >> 1;
    if (table[4 * mid + 3] <= bid) lo = mid; else hi = mid - 1;
  }
  const int64_t off = table[4 * lo];
  const int R = (int)table[4 * lo + 1], C = (int)table[4 * lo + 2];
  const int t = (int)(bid - table[4 * lo + 3]);
  const int tiles_c = (C + 31) >> 5;
  const int c0 = (t % tiles_c) * 32, r0 = (t / tiles_c) * 32;
  const float* src = src_base + off;
  bf16_t* dst = dst_base + off;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) {
    const int r = r0 + j, c = c0 + tx;
    tile[j][tx] = (r < R && c < C) ? src[(int64_t)r * C + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, r = r0 + tx;
    if (c < C && r < R) dst[(int64_t)c * R + r] = f32_to_bf16(tile[tx][j]);
  }
}

__global__ __launch_bounds__(256) void gelu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = gelu_erf(x[i]);
}
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dx[i] = dy[i] * gelu_erf_grad(x[i]);
}

static unsigned stream_grid(int64_t n, int per_thread) {
  int64_t b = ceil_div(n, (int64_t)256 * per_thread);
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace dinox

using namespace dinox;

static int adamw_launch(float* p, const float* g, float* m, float* v, float* teacher, int64_t n, float lr, float weight_decay,
                        float beta1, float beta2, float eps, float inv_bc1, float inv_sqrt_bc2, float ema, float grad_scale,
                        float* gnorm_sq, float* ws, const float* hyper, void* stream) {
  DX_REQUIRE(p && g && m && v && gnorm_sq && ws, DINOX_EINVAL, "adamw_ema: null pointer");
  DX_REQUIRE(n > 0, DINOX_EINVAL, "adamw_ema: n=%lld", (long long)n);
  DX_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)teacher) & 15) == 0, DINOX_EALIGN,
             "adamw_ema: arenas must be 16-byte aligned");
  const unsigned blocks = stream_grid(n, 4);
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(adamw_ema_kernel, dim3(blocks), dim3(256), 0, st, p, g, m, v, teacher, n, lr, weight_decay, beta1, beta2,
                     eps, inv_bc1, inv_sqrt_bc2, ema, grad_scale, ws, hyper);
  int rc = check_launch("adamw_ema");
  if (rc) return rc;
  hipLaunchKernelGGL(sum_parts, dim3(1), dim3(256), 0, st, ws, (int)blocks, gnorm_sq);
  return check_launch("adamw_ema_norm");
}

extern "C" int dinox_adamw_ema(float* p, const float* g, float* m, float* v, float* teacher, int64_t n, float lr,
                               float weight_decay, float beta1, float beta2, float eps, int step_t, float ema,
                               float grad_scale, float* gnorm_sq, float* ws, void* stream) {
  DX_REQUIRE(step_t >= 1, DINOX_EINVAL, "adamw_ema: step_t=%d", step_t);
  const double bc1 = 1.0 - pow((double)beta1, step_t), bc2 = 1.0 - pow((double)beta2, step_t);
  return adamw_launch(p, g, m, v, teacher, n, lr, weight_decay, beta1, beta2, eps, (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), ema,
                      grad_scale, gnorm_sq, ws, nullptr, stream);
}

extern "C" int dinox_adamw_ema_dev(float* p, const float* g, float* m, float* v, float* teacher, int64_t n, const float* hyper,
                                   float weight_decay, float beta1, float beta2, float eps, float ema, float grad_scale,
                                   float* gnorm_sq, float* ws, void* stream) {
  DX_REQUIRE(hyper, DINOX_EINVAL, "adamw_ema_dev: null hyper");
  return adamw_launch(p, g, m, v, teacher, n, 0.f, weight_decay, beta1, beta2, eps, 1.f, 1.f, ema, grad_scale, gnorm_sq, ws, hyper, stream);
}

extern "C" int dinox_sumsq(const float* x, int64_t n, float* out, float* ws, void* stream) {
  DX_REQUIRE(x && out && ws && n > 0, DINOX_EINVAL, "sumsq: bad arguments");
  const unsigned blocks = stream_grid(n, 8);
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(sumsq_partial, dim3(blocks), dim3(256), 0, st, x, n, ws);
  hipLaunchKernelGGL(sum_parts, dim3(1), dim3(256), 0, st, ws, (int)blocks, out);
  return check_launch("sumsq");
}

extern "C" int dinox_cast_bf16(const float* src, void* dst, int64_t n, void* stream) {
  DX_REQUIRE(src && dst && n > 0, DINOX_EINVAL, "cast_bf16: bad arguments");
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(stream_grid(n, 4)), dim3(256), 0, as_stream(stream), src, (bf16_t*)dst, n);
  return check_launch("cast_bf16");
}

extern "C" int dinox_cast_transpose_bf16(const float* src, void* dst, int R, int C, void* stream) {
  DX_REQUIRE(src && dst && R > 0 && C > 0, DINOX_EINVAL, "cast_transpose_bf16: bad arguments");
  dim3 grid((unsigned)ceil_div(C, 32), (unsigned)ceil_div(R, 32));
  hipLaunchKernelGGL(cast_transpose_kernel, grid, dim3(256), 0, as_stream(stream), src, (bf16_t*)dst, R, C);
  return check_launch("cast_transpose_bf16");
}

extern "C" int dinox_cast_transpose_bf16_multi(const float* src_base, void* dst_base, const int64_t* table, int n_mats,
                                               int64_t total_tiles, void* stream) {
  DX_REQUIRE(src_base && dst_base && table, DINOX_EINVAL, "cast_transpose_bf16_multi: null pointer");
  DX_REQUIRE(n_mats > 0 && total_tiles > 0 && total_tiles <= 0x7fffffff, DINOX_EINVAL, "cast_transpose_bf16_multi: n_mats=%d tiles=%lld",
             n_mats, (long long)total_tiles);
  hipLaunchKernelGGL(cast_transpose_multi_kernel, dim3((unsigned)total_tiles), dim3(256), 0, as_stream(stream), src_base,
                     (bf16_t*)dst_base, table, n_mats);
  return check_launch("cast_transpose_bf16_multi");
}

extern "C" int dinox_gelu_fwd(const float* x, float* y, int64_t n, void* stream) {
  DX_REQUIRE(x && y && n > 0, DINOX_EINVAL, "gelu_fwd: bad arguments");
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3(stream_grid(n, 4)), dim3(256), 0, as_stream(stream), x, y, n);
  return check_launch("gelu_fwd");
}

extern "C" int dinox_gelu_bwd(const float* dy, const float* x, float* dx, int64_t n, void* stream) {
  DX_REQUIRE(dy && x && dx && n > 0, DINOX_EINVAL, "gelu_bwd: bad arguments");
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(stream_grid(n, 4)), dim3(256), 0, as_stream(stream), dy, x, dx, n);
  return check_launch("gelu_bwd");
}
