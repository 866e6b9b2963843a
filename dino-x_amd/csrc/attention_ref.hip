// attention_ref.hip -- straightforward fp32-math attention core (any N, d <= 128, fp32 or bf16 storage).
// Parity-mode kernel and the on-GPU reference for the MFMA flash kernels in attention_bf16.hip.
// One wave per query row (forward, dQ) or per key row (dK/dV); scores for the row live in LDS.
// Replaces reshape/permute/unbind + F.scaled_dot_product_attention + transpose/reshape of the
// reference (zoo/arch.py:45-52): softmax(Q K^T / sqrt(d)) V, no mask, no dropout.
#include "common.h"

namespace dinox {

constexpr int AR_THREADS = 256;  // 4 waves
constexpr int AR_ROWS = 16;      // rows (queries or keys) per block, 4 per wave

// element offset of (token i, which in {0:q,1:k,2:v}, head hh, dim 0) in packed qkv [B][N][3][h][d]
__device__ __forceinline__ int64_t qkv_off(int b, int N, int i, int which, int heads, int hh, int d) {
  return (((int64_t)b * N + i) * 3 + which) * ((int64_t)heads * d) + (int64_t)hh * d;
}

template <int DT>
__global__ __launch_bounds__(AR_THREADS) void attn_ref_fwd(const void* __restrict__ qkv, void* __restrict__ o,
                                                           float* __restrict__ lse, int B, int N, int heads, int d,
                                                           float sc) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int b = blockIdx.x / heads, hh = blockIdx.x % heads;
  float* sbuf = lds + (size_t)wv * (N + 128);
  float* qrow = sbuf + N;
  const int64_t C = (int64_t)heads * d;
  for (int rr = wv; rr < AR_ROWS; rr += 4) {
    const int i = blockIdx.y * AR_ROWS + rr;
    if (i >= N) break;  // wave-uniform
    const int64_t qo = qkv_off(b, N, i, 0, heads, hh, d);
    for (int dd = lane; dd < d; dd += 64) qrow[dd] = elem<DT>::ld(qkv, qo + dd);
    __builtin_amdgcn_s_waitcnt(0);  // LDS writes of this wave visible to its own later reads
    __builtin_amdgcn_wave_barrier();
    float mx = -INFINITY;
    for (int j = lane; j < N; j += 64) {
      const int64_t ko = qkv_off(b, N, j, 1, heads, hh, d);
      float s = 0.f;
      for (int dd = 0; dd < d; ++dd) s += qrow[dd] * elem<DT>::ld(qkv, ko + dd);
      s *= sc;
      sbuf[j] = s;
      mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < N; j += 64) {
      const float e = expf(sbuf[j] - mx);
      sbuf[j] = e;
      sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    if (lane == 0) lse[((int64_t)b * heads + hh) * N + i] = mx + logf(sum);
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    for (int dd = lane; dd < d; dd += 64) {
      float acc = 0.f;
      for (int j = 0; j < N; ++j) acc += sbuf[j] * elem<DT>::ld(qkv, qkv_off(b, N, j, 2, heads, hh, d) + dd);
      elem<DT>::st(o, ((int64_t)b * N + i) * C + (int64_t)hh * d + dd, acc * inv);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// dQ: one wave per query row.
template <int DT>
__global__ __launch_bounds__(AR_THREADS) void attn_ref_bwd_dq(const void* __restrict__ d_o, const void* __restrict__ qkv,
                                                              const void* __restrict__ o, const float* __restrict__ lse,
                                                              void* __restrict__ dqkv, int B, int N, int heads, int d,
                                                              float sc) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int b = blockIdx.x / heads, hh = blockIdx.x % heads;
  float* sbuf = lds + (size_t)wv * (N + 256);
  float* qrow = sbuf + N;
  float* dorow = qrow + 128;
  const int64_t C = (int64_t)heads * d;
  for (int rr = wv; rr < AR_ROWS; rr += 4) {
    const int i = blockIdx.y * AR_ROWS + rr;
    if (i >= N) break;
    const int64_t qo = qkv_off(b, N, i, 0, heads, hh, d);
    const int64_t oo = ((int64_t)b * N + i) * C + (int64_t)hh * d;
    float dl = 0.f;
    for (int dd = lane; dd < d; dd += 64) {
      qrow[dd] = elem<DT>::ld(qkv, qo + dd);
      const float g = elem<DT>::ld(d_o, oo + dd);
      dorow[dd] = g;
      dl += g * elem<DT>::ld(o, oo + dd);
    }
    const float delta = wave_sum(dl);
    const float L = lse[((int64_t)b * heads + hh) * N + i];
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    for (int j = lane; j < N; j += 64) {
      const int64_t ko = qkv_off(b, N, j, 1, heads, hh, d), vo = qkv_off(b, N, j, 2, heads, hh, d);
      float s = 0.f, dp = 0.f;
      for (int dd = 0; dd < d; ++dd) {
        s += qrow[dd] * elem<DT>::ld(qkv, ko + dd);
        dp += dorow[dd] * elem<DT>::ld(qkv, vo + dd);
      }
      const float p = expf(s * sc - L);
      sbuf[j] = p * (dp - delta);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    for (int dd = lane; dd < d; dd += 64) {
      float acc = 0.f;
      for (int j = 0; j < N; ++j) acc += sbuf[j] * elem<DT>::ld(qkv, qkv_off(b, N, j, 1, heads, hh, d) + dd);
      elem<DT>::st(dqkv, qo + dd, acc * sc);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// dK, dV: one wave per key row.
template <int DT>
__global__ __launch_bounds__(AR_THREADS) void attn_ref_bwd_dkv(const void* __restrict__ d_o, const void* __restrict__ qkv,
                                                               const void* __restrict__ o, const float* __restrict__ lse,
                                                               void* __restrict__ dqkv, int B, int N, int heads, int d,
                                                               float sc) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int b = blockIdx.x / heads, hh = blockIdx.x % heads;
  float* pbuf = lds + (size_t)wv * (2 * N + 256);
  float* dsbuf = pbuf + N;
  float* krow = dsbuf + N;
  float* vrow = krow + 128;
  const int64_t C = (int64_t)heads * d;
  for (int rr = wv; rr < AR_ROWS; rr += 4) {
    const int j = blockIdx.y * AR_ROWS + rr;
    if (j >= N) break;
    const int64_t ko = qkv_off(b, N, j, 1, heads, hh, d), vo = qkv_off(b, N, j, 2, heads, hh, d);
    for (int dd = lane; dd < d; dd += 64) {
      krow[dd] = elem<DT>::ld(qkv, ko + dd);
      vrow[dd] = elem<DT>::ld(qkv, vo + dd);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < N; i += 64) {
      const int64_t qo = qkv_off(b, N, i, 0, heads, hh, d);
      const int64_t oo = ((int64_t)b * N + i) * C + (int64_t)hh * d;
      float s = 0.f, dp = 0.f, dl = 0.f;
      for (int dd = 0; dd < d; ++dd) {
        const float g = elem<DT>::ld(d_o, oo + dd);
        s += krow[dd] * elem<DT>::ld(qkv, qo + dd);
        dp += g * vrow[dd];
        dl += g * elem<DT>::ld(o, oo + dd);
      }
      const float p = expf(s * sc - lse[((int64_t)b * heads + hh) * N + i]);
      pbuf[i] = p;
      dsbuf[i] = p * (dp - dl);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    for (int dd = lane; dd < d; dd += 64) {
      float av = 0.f, ak = 0.f;
      for (int i = 0; i < N; ++i) {
        const int64_t oo = ((int64_t)b * N + i) * C + (int64_t)hh * d;
        av += pbuf[i] * elem<DT>::ld(d_o, oo + dd);
        ak += dsbuf[i] * elem<DT>::ld(qkv, qkv_off(b, N, i, 0, heads, hh, d) + dd);
      }
      elem<DT>::st(dqkv, vo + dd, av);
      elem<DT>::st(dqkv, ko + dd, ak * sc);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------------------------ rows of a materialised score matrix
// The fp32 parity mode at full size runs attention as batched exact-fp32 products (S = scale Q K^T, O = P V, and the five products of the
// backward pass) with the softmax on the materialised [rows][n] scores in between: the per-lane kernels above take 23 ms per layer at bs 64.
// One wave per row; rows are (image, query) pairs of ONE head; lse lives in the [B][heads][N] tensor of the attention API.
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ s, float* __restrict__ lse, int64_t rows, int n,
                                                           int rows_per_image, int64_t lse_image_stride) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float* row = s + r * n;
  float mx = -INFINITY;
  for (int j = lane; j < n; j += 64) mx = fmaxf(mx, row[j]);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < n; j += 64) sum += __expf(row[j] - mx);
  sum = wave_sum(sum);
  const float l = mx + __logf(sum);
  for (int j = lane; j < n; j += 64) row[j] = __expf(row[j] - l);
  if (lane == 0) lse[(r / rows_per_image) * lse_image_stride + (r % rows_per_image)] = l;
}

// p = exp(s - lse) (in place of s), ds = p * (dp - sum_j p_j dp_j) * scale (in place of dp)
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(float* __restrict__ s, float* __restrict__ dp, const float* __restrict__ lse,
                                                               float scale, int64_t rows, int n, int rows_per_image,
                                                               int64_t lse_image_stride) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float* srow = s + r * n;
  float* drow = dp + r * n;
  const float l = lse[(r / rows_per_image) * lse_image_stride + (r % rows_per_image)];
  float dot = 0.f;
  for (int j = lane; j < n; j += 64) {
    const float p = __expf(srow[j] - l);
    srow[j] = p;
    dot += p * drow[j];
  }
  dot = wave_sum(dot);
  for (int j = lane; j < n; j += 64) drow[j] = srow[j] * (drow[j] - dot) * scale;
}

int launch_attention_ref_fwd(const void* qkv, void* o, float* lse, int B, int N, int heads, int d, int dtype,
                             hipStream_t st) {
  dim3 grid((unsigned)(B * heads), (unsigned)ceil_div(N, AR_ROWS));
  const size_t lds = (size_t)4 * (N + 128) * sizeof(float);
  if (lds > 64 * 1024) return fail(DINOX_EUNSUPPORTED, "attention_fwd: N=%d needs %zu B of LDS", N, lds);
  const float sc = 1.0f / sqrtf((float)d);
  if (dtype == DINOX_F32)
    hipLaunchKernelGGL((attn_ref_fwd<DINOX_F32>), grid, dim3(AR_THREADS), lds, st, qkv, o, lse, B, N, heads, d, sc);
  else
    hipLaunchKernelGGL((attn_ref_fwd<DINOX_BF16>), grid, dim3(AR_THREADS), lds, st, qkv, o, lse, B, N, heads, d, sc);
  return check_launch("attention_ref_fwd");
}

int launch_attention_ref_bwd(const void* d_o, const void* qkv, const void* o, const float* lse, void* dqkv, int B, int N,
                             int heads, int d, int dtype, hipStream_t st) {
  dim3 grid((unsigned)(B * heads), (unsigned)ceil_div(N, AR_ROWS));
  const size_t lds1 = (size_t)4 * (N + 256) * sizeof(float);
  const size_t lds2 = (size_t)4 * (2 * N + 256) * sizeof(float);
  if (lds2 > 64 * 1024) return fail(DINOX_EUNSUPPORTED, "attention_bwd: N=%d needs %zu B of LDS", N, lds2);
  const float sc = 1.0f / sqrtf((float)d);
  if (dtype == DINOX_F32) {
    hipLaunchKernelGGL((attn_ref_bwd_dq<DINOX_F32>), grid, dim3(AR_THREADS), lds1, st, d_o, qkv, o, lse, dqkv, B, N, heads, d, sc);
    hipLaunchKernelGGL((attn_ref_bwd_dkv<DINOX_F32>), grid, dim3(AR_THREADS), lds2, st, d_o, qkv, o, lse, dqkv, B, N, heads, d, sc);
  } else {
    hipLaunchKernelGGL((attn_ref_bwd_dq<DINOX_BF16>), grid, dim3(AR_THREADS), lds1, st, d_o, qkv, o, lse, dqkv, B, N, heads, d, sc);
    hipLaunchKernelGGL((attn_ref_bwd_dkv<DINOX_BF16>), grid, dim3(AR_THREADS), lds2, st, d_o, qkv, o, lse, dqkv, B, N, heads, d, sc);
  }
  return check_launch("attention_ref_bwd");
}

}  // namespace dinox

extern "C" int dinox_softmax_rows(float* s, float* lse, int64_t rows, int n, int rows_per_image, int64_t lse_image_stride, void* stream) {
  DX_REQUIRE(s && lse, DINOX_EINVAL, "softmax_rows: null pointer");
  DX_REQUIRE(rows > 0 && n > 0 && rows_per_image > 0, DINOX_EINVAL, "softmax_rows: rows=%lld n=%d", (long long)rows, n);
  hipLaunchKernelGGL(dinox::softmax_rows_kernel, dim3((unsigned)dinox::ceil_div(rows, (int64_t)4)), dim3(256), 0, dinox::as_stream(stream), s, lse, rows, n,
                     rows_per_image, lse_image_stride);
  return dinox::check_launch("softmax_rows");
}

extern "C" int dinox_softmax_bwd_rows(float* s, float* dp, const float* lse, float scale, int64_t rows, int n, int rows_per_image,
                                      int64_t lse_image_stride, void* stream) {
  DX_REQUIRE(s && dp && lse, DINOX_EINVAL, "softmax_bwd_rows: null pointer");
  DX_REQUIRE(rows > 0 && n > 0 && rows_per_image > 0, DINOX_EINVAL, "softmax_bwd_rows: rows=%lld n=%d", (long long)rows, n);
  hipLaunchKernelGGL(dinox::softmax_bwd_rows_kernel, dim3((unsigned)dinox::ceil_div(rows, (int64_t)4)), dim3(256), 0, dinox::as_stream(stream), s, dp, lse,
                     scale, rows, n, rows_per_image, lse_image_stride);
  return dinox::check_launch("softmax_bwd_rows");
}
