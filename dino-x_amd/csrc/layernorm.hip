// layernorm.hip -- LayerNorm forward/backward over the fp32 residual stream.
// Replaces nn.LayerNorm(D) (reference zoo/arch.py:89,91,126,187).  HBM-bound: one wave per row, the
// row is read once into registers (float4 per lane), statistics by wave shuffles, no LDS on the
// forward path.  Algorithmic bytes per row: fwd 4D read + (4|2)D write; bwd (4|2)D + 4D read, 4D
// (+4D when accumulating, +2D for the bf16 copy) written.
#include "common.h"

namespace dinox {

constexpr int LN_THREADS = 256;   // 4 waves = 4 rows in flight per block
constexpr int LN_MAXV = 8;        // float4 per lane kept in registers -> dim <= 2048 on the fast path
constexpr int LN_MAX_PARTS = 1024;

template <int OUT_DT, int NV>
__global__ __launch_bounds__(LN_THREADS) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ b, void* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd,
                                                            int64_t rows, int dim, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (LN_THREADS / 64) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (LN_THREADS / 64);
  const int nvec = dim >> 2;
  const float inv = 1.0f / (float)dim;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const float4* xr = reinterpret_cast<const float4*>(x + r * dim);
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = lane + 64 * j;
      v[j] = (c < nvec) ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
      s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    }
    const float mu = wave_sum(s) * inv;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = lane + 64 * j;
      if (c < nvec) {
        const float a = v[j].x - mu, bb = v[j].y - mu, cc = v[j].z - mu, d = v[j].w - mu;
        q += (a * a + bb * bb) + (cc * cc + d * d);
      }
    }
    const float rs = rsqrtf(wave_sum(q) * inv + eps);
    if (lane == 0) {
      mean[r] = mu;
      rstd[r] = rs;
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = lane + 64 * j;
      if (c < nvec) {
        const float4 ww = reinterpret_cast<const float4*>(w)[c];
        const float4 bb = reinterpret_cast<const float4*>(b)[c];
        float4 o;
        o.x = (v[j].x - mu) * rs * ww.x + bb.x;
        o.y = (v[j].y - mu) * rs * ww.y + bb.y;
        o.z = (v[j].z - mu) * rs * ww.z + bb.z;
        o.w = (v[j].w - mu) * rs * ww.w + bb.w;
        if (OUT_DT == DINOX_F32) {
          store_stream(reinterpret_cast<float4*>((float*)y + r * dim) + c, o);
        } else {
          ushort4 p;
          p.x = f32_to_bf16(o.x);
          p.y = f32_to_bf16(o.y);
          p.z = f32_to_bf16(o.z);
          p.w = f32_to_bf16(o.w);
          store_stream(reinterpret_cast<ushort4*>((bf16_t*)y + r * dim) + c, p);
        }
      }
    }
  }
}

// Width 384 (ViT-S): a row is 96 float4 = 1.5 wave-loads, so the one-wave-per-row kernel above leaves half the lanes idle in its second
// load and has one row (1.5 KB) in flight per wave.  Here a HALF wave owns a row (32 lanes x 3 float4, every lane busy, reductions over
// 32 lanes) and each half walks two rows per iteration, the second requested before the first is reduced: 6 KB in flight per wave.
template <int OUT_DT>
__global__ __launch_bounds__(LN_THREADS) void ln_fwd_384(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ b, void* __restrict__ y, float* __restrict__ mean,
                                                         float* __restrict__ rstd, int64_t rows, float eps) {
  constexpr int DIM = 384;
  const int lane = threadIdx.x & 63, sl = lane & 31;
  const int64_t half = ((int64_t)blockIdx.x * (LN_THREADS / 64) + (threadIdx.x >> 6)) * 2 + (lane >> 5);
  const int64_t nhalves = (int64_t)gridDim.x * (LN_THREADS / 64) * 2;
  float4 ww[3], bb[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    ww[j] = reinterpret_cast<const float4*>(w)[sl + 32 * j];
    bb[j] = reinterpret_cast<const float4*>(b)[sl + 32 * j];
  }
  auto sum32 = [](float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  };
  auto finish = [&](const float4 (&v)[3], int64_t r) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    const float mu = sum32(s) * (1.0f / DIM);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float a = v[j].x - mu, c = v[j].y - mu, d = v[j].z - mu, e = v[j].w - mu;
      q += (a * a + c * c) + (d * d + e * e);
    }
    const float rs = rsqrtf(sum32(q) * (1.0f / DIM) + eps);
    if (sl == 0) {
      mean[r] = mu;
      rstd[r] = rs;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float4 o;
      o.x = (v[j].x - mu) * rs * ww[j].x + bb[j].x;
      o.y = (v[j].y - mu) * rs * ww[j].y + bb[j].y;
      o.z = (v[j].z - mu) * rs * ww[j].z + bb[j].z;
      o.w = (v[j].w - mu) * rs * ww[j].w + bb[j].w;
      if (OUT_DT == DINOX_F32) {
        store_stream(reinterpret_cast<float4*>((float*)y + r * DIM) + (sl + 32 * j), o);
      } else {
        ushort4 p;
        p.x = f32_to_bf16(o.x);
        p.y = f32_to_bf16(o.y);
        p.z = f32_to_bf16(o.z);
        p.w = f32_to_bf16(o.w);
        store_stream(reinterpret_cast<ushort4*>((bf16_t*)y + r * DIM) + (sl + 32 * j), p);
      }
    }
  };
  for (int64_t r0 = half; r0 < rows; r0 += 2 * nhalves) {
    const int64_t r1 = r0 + nhalves;
    const bool two = r1 < rows;
    float4 v0[3], v1[3];
    const float4* x0 = reinterpret_cast<const float4*>(x + r0 * DIM);
    const float4* x1 = reinterpret_cast<const float4*>(x + (two ? r1 : r0) * DIM);
#pragma unroll
    for (int j = 0; j < 3; ++j) v0[j] = x0[sl + 32 * j];
#pragma unroll
    for (int j = 0; j < 3; ++j) v1[j] = x1[sl + 32 * j];
    finish(v0, r0);
    if (two) finish(v1, r1);
  }
}

// Generic (any dim) fallback: one wave per row, three cached passes.
template <int OUT_DT>
__global__ __launch_bounds__(LN_THREADS) void ln_fwd_generic(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ b, void* __restrict__ y,
                                                             float* __restrict__ mean, float* __restrict__ rstd,
                                                             int64_t rows, int dim, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (LN_THREADS / 64) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (LN_THREADS / 64);
  const float inv = 1.0f / (float)dim;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const float* xr = x + r * dim;
    float s = 0.f;
    for (int c = lane; c < dim; c += 64) s += xr[c];
    const float mu = wave_sum(s) * inv;
    float q = 0.f;
    for (int c = lane; c < dim; c += 64) {
      const float d = xr[c] - mu;
      q += d * d;
    }
    const float rs = rsqrtf(wave_sum(q) * inv + eps);
    if (lane == 0) {
      mean[r] = mu;
      rstd[r] = rs;
    }
    for (int c = lane; c < dim; c += 64) elem<OUT_DT>::st(y, r * dim + c, (xr[c] - mu) * rs * w[c] + b[c]);
  }
}

// Backward, stage 1.  Each lane owns fixed columns, so dw/db partial sums stay in registers across
// the rows its wave visits; the block's 4 waves are combined through LDS and one partial row per
// block is written to ws[part][2][dim].  Stage 2 sums the parts.
template <int DY_DT, int NV>
__global__ __launch_bounds__(LN_THREADS) void ln_bwd_kernel(const void* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ w, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, float* dx,
                                                            void* __restrict__ dx_lowp, float* __restrict__ ws,
                                                            int64_t rows, int dim, const float* dx_add) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t wave = (int64_t)blockIdx.x * (LN_THREADS / 64) + wv;
  const int64_t nwaves = (int64_t)gridDim.x * (LN_THREADS / 64);
  const int nvec = dim >> 2;
  const float inv = 1.0f / (float)dim;
  float4 aw[NV], ab[NV], wreg[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    aw[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    ab[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int c = lane + 64 * j;
    wreg[j] = (c < nvec) ? reinterpret_cast<const float4*>(w)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int64_t r = wave; r < rows; r += nwaves) {
    const float mu = mean[r], rs = rstd[r];
    float4 xh[NV], g[NV], add[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {      // the incoming residual gradient is requested with the row, not after the two reductions
      const int c = lane + 64 * j;
      add[j] = (dx_add && c < nvec) ? reinterpret_cast<const float4*>(dx_add + r * dim)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = lane + 64 * j;
      float4 d = make_float4(0.f, 0.f, 0.f, 0.f), xv = make_float4(mu, mu, mu, mu);
      if (c < nvec) {
        xv = reinterpret_cast<const float4*>(x + r * dim)[c];
        if (DY_DT == DINOX_F32) {
          d = reinterpret_cast<const float4*>((const float*)dy + r * dim)[c];
        } else {
          const ushort4 p = reinterpret_cast<const ushort4*>((const bf16_t*)dy + r * dim)[c];
          d = make_float4(bf16_to_f32(p.x), bf16_to_f32(p.y), bf16_to_f32(p.z), bf16_to_f32(p.w));
        }
      }
      xh[j] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
      g[j] = make_float4(d.x * wreg[j].x, d.y * wreg[j].y, d.z * wreg[j].z, d.w * wreg[j].w);
      aw[j].x += d.x * xh[j].x; aw[j].y += d.y * xh[j].y; aw[j].z += d.z * xh[j].z; aw[j].w += d.w * xh[j].w;
      ab[j].x += d.x; ab[j].y += d.y; ab[j].z += d.z; ab[j].w += d.w;
      s1 += (g[j].x + g[j].y) + (g[j].z + g[j].w);
      s2 += (g[j].x * xh[j].x + g[j].y * xh[j].y) + (g[j].z * xh[j].z + g[j].w * xh[j].w);
    }
    const float m1 = wave_sum(s1) * inv, m2 = wave_sum(s2) * inv;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = lane + 64 * j;
      if (c < nvec) {
        float4 o;
        o.x = rs * (g[j].x - m1 - xh[j].x * m2);
        o.y = rs * (g[j].y - m1 - xh[j].y * m2);
        o.z = rs * (g[j].z - m1 - xh[j].z * m2);
        o.w = rs * (g[j].w - m1 - xh[j].w * m2);
        float4* dst = reinterpret_cast<float4*>(dx + r * dim) + c;
        o.x += add[j].x; o.y += add[j].y; o.z += add[j].z; o.w += add[j].w;
        store_stream(dst, o);
        if (dx_lowp) {
          ushort4 p;
          p.x = f32_to_bf16(o.x); p.y = f32_to_bf16(o.y); p.z = f32_to_bf16(o.z); p.w = f32_to_bf16(o.w);
          store_stream(reinterpret_cast<ushort4*>((bf16_t*)dx_lowp + r * dim) + c, p);
        }
      }
    }
  }
  // combine the 4 waves: lds[wv][2][dim]
  float* mine = lds + (size_t)wv * 2 * dim;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = lane + 64 * j;
    if (c < nvec) {
      reinterpret_cast<float4*>(mine)[c] = aw[j];
      reinterpret_cast<float4*>(mine + dim)[c] = ab[j];
    }
  }
  __syncthreads();
  float* out = ws + (size_t)blockIdx.x * 2 * dim;
  for (int c = threadIdx.x; c < 2 * dim; c += LN_THREADS)
    out[c] = (lds[c] + lds[2 * dim + c]) + (lds[4 * dim + c] + lds[6 * dim + c]);
}

// Width 384, backward: the same half-wave-per-row form as ln_fwd_384 (every lane busy, two rows requested per half before the first
// is reduced, the incoming residual gradient dx_add fetched with them instead of after the reductions).  dx may alias dx_add: a row is
// read and written by the same lanes, each element read before it is written.  The per-column partial sums of a half wave meet its
// twin (lane ^ 32 holds the same columns) by one exchange, then the block's four waves through LDS as in ln_bwd_kernel.
template <int DY_DT>
__global__ __launch_bounds__(LN_THREADS) void ln_bwd_384(const void* __restrict__ dy, const float* __restrict__ x,
                                                         const float* __restrict__ w, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, float* dx, void* __restrict__ dx_lowp,
                                                         float* __restrict__ ws, int64_t rows, const float* dx_add) {
  constexpr int DIM = 384;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, sl = lane & 31, wv = threadIdx.x >> 6;
  const int64_t half = ((int64_t)blockIdx.x * (LN_THREADS / 64) + wv) * 2 + (lane >> 5);
  const int64_t nhalves = (int64_t)gridDim.x * (LN_THREADS / 64) * 2;
  float4 aw[3], ab[3], wreg[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    aw[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    ab[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    wreg[j] = reinterpret_cast<const float4*>(w)[sl + 32 * j];
  }
  auto sum32 = [](float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  };
  auto fetch = [&](int64_t r, float4 (&xv)[3], float4 (&d)[3], float4 (&a)[3]) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int c = sl + 32 * j;
      xv[j] = reinterpret_cast<const float4*>(x + r * DIM)[c];
      if (DY_DT == DINOX_F32) {
        d[j] = reinterpret_cast<const float4*>((const float*)dy + r * DIM)[c];
      } else {
        const ushort4 p = reinterpret_cast<const ushort4*>((const bf16_t*)dy + r * DIM)[c];
        d[j] = make_float4(bf16_to_f32(p.x), bf16_to_f32(p.y), bf16_to_f32(p.z), bf16_to_f32(p.w));
      }
      a[j] = dx_add ? reinterpret_cast<const float4*>(dx_add + r * DIM)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto finish = [&](int64_t r, const float4 (&xv)[3], const float4 (&d)[3], const float4 (&a)[3]) {
    const float mu = mean[r], rs = rstd[r];
    float4 xh[3], g[3];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      xh[j] = make_float4((xv[j].x - mu) * rs, (xv[j].y - mu) * rs, (xv[j].z - mu) * rs, (xv[j].w - mu) * rs);
      g[j] = make_float4(d[j].x * wreg[j].x, d[j].y * wreg[j].y, d[j].z * wreg[j].z, d[j].w * wreg[j].w);
      aw[j].x += d[j].x * xh[j].x; aw[j].y += d[j].y * xh[j].y; aw[j].z += d[j].z * xh[j].z; aw[j].w += d[j].w * xh[j].w;
      ab[j].x += d[j].x; ab[j].y += d[j].y; ab[j].z += d[j].z; ab[j].w += d[j].w;
      s1 += (g[j].x + g[j].y) + (g[j].z + g[j].w);
      s2 += (g[j].x * xh[j].x + g[j].y * xh[j].y) + (g[j].z * xh[j].z + g[j].w * xh[j].w);
    }
    const float m1 = sum32(s1) * (1.0f / DIM), m2 = sum32(s2) * (1.0f / DIM);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int c = sl + 32 * j;
      float4 o;
      o.x = rs * (g[j].x - m1 - xh[j].x * m2) + a[j].x;
      o.y = rs * (g[j].y - m1 - xh[j].y * m2) + a[j].y;
      o.z = rs * (g[j].z - m1 - xh[j].z * m2) + a[j].z;
      o.w = rs * (g[j].w - m1 - xh[j].w * m2) + a[j].w;
      store_stream(reinterpret_cast<float4*>(dx + r * DIM) + c, o);
      if (dx_lowp) {
        ushort4 p;
        p.x = f32_to_bf16(o.x); p.y = f32_to_bf16(o.y); p.z = f32_to_bf16(o.z); p.w = f32_to_bf16(o.w);
        store_stream(reinterpret_cast<ushort4*>((bf16_t*)dx_lowp + r * DIM) + c, p);
      }
    }
  };
  for (int64_t r0 = half; r0 < rows; r0 += 2 * nhalves) {
    const int64_t r1 = r0 + nhalves;
    const bool two = r1 < rows;
    float4 x0[3], d0[3], a0[3], x1[3], d1[3], a1[3];
    fetch(r0, x0, d0, a0);
    fetch(two ? r1 : r0, x1, d1, a1);
    finish(r0, x0, d0, a0);
    if (two) finish(r1, x1, d1, a1);
  }
  // the twin half (same columns), then the block's four waves: lds[wv][2][DIM]
  float* mine = lds + (size_t)wv * 2 * DIM;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float4 sw, sb;
    sw.x = aw[j].x + __shfl_xor(aw[j].x, 32, 64); sw.y = aw[j].y + __shfl_xor(aw[j].y, 32, 64);
    sw.z = aw[j].z + __shfl_xor(aw[j].z, 32, 64); sw.w = aw[j].w + __shfl_xor(aw[j].w, 32, 64);
    sb.x = ab[j].x + __shfl_xor(ab[j].x, 32, 64); sb.y = ab[j].y + __shfl_xor(ab[j].y, 32, 64);
    sb.z = ab[j].z + __shfl_xor(ab[j].z, 32, 64); sb.w = ab[j].w + __shfl_xor(ab[j].w, 32, 64);
    if (lane < 32) {
      reinterpret_cast<float4*>(mine)[sl + 32 * j] = sw;
      reinterpret_cast<float4*>(mine + DIM)[sl + 32 * j] = sb;
    }
  }
  __syncthreads();
  float* out = ws + (size_t)blockIdx.x * 2 * DIM;
  for (int c = threadIdx.x; c < 2 * DIM; c += LN_THREADS)
    out[c] = (lds[c] + lds[2 * DIM + c]) + (lds[4 * DIM + c] + lds[6 * DIM + c]);
}

template <int DY_DT>
__global__ __launch_bounds__(LN_THREADS) void ln_bwd_generic(const void* __restrict__ dy, const float* __restrict__ x,
                                                             const float* __restrict__ w, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, float* dx,
                                                             void* __restrict__ dx_lowp, float* __restrict__ ws,
                                                             int64_t rows, int dim, const float* dx_add) {
  // Generic fallback: one wave per row for dx; dw/db partials via LDS atomics per block.
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int c = threadIdx.x; c < 2 * dim; c += LN_THREADS) lds[c] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (LN_THREADS / 64) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (LN_THREADS / 64);
  const float inv = 1.0f / (float)dim;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const float mu = mean[r], rs = rstd[r];
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < dim; c += 64) {
      const float d = elem<DY_DT>::ld(dy, r * dim + c), xh = (x[r * dim + c] - mu) * rs, g = d * w[c];
      s1 += g;
      s2 += g * xh;
      atomicAdd(&lds[c], d * xh);
      atomicAdd(&lds[dim + c], d);
    }
    const float m1 = wave_sum(s1) * inv, m2 = wave_sum(s2) * inv;
    for (int c = lane; c < dim; c += 64) {
      const float d = elem<DY_DT>::ld(dy, r * dim + c), xh = (x[r * dim + c] - mu) * rs;
      float o = rs * (d * w[c] - m1 - xh * m2);
      if (dx_add) o += dx_add[r * dim + c];
      dx[r * dim + c] = o;
      if (dx_lowp) ((bf16_t*)dx_lowp)[r * dim + c] = f32_to_bf16(o);
    }
  }
  __syncthreads();
  float* out = ws + (size_t)blockIdx.x * 2 * dim;
  for (int c = threadIdx.x; c < 2 * dim; c += LN_THREADS) out[c] = lds[c];
}

// Stage 2: out[c] (+)= sum_p ws[p][c], in a FIXED order (no atomics: bit-reproducible).  A block owns 4 columns: thread (c, g), g < 64,
// sums the parts p == g (mod 64) in increasing p (16 independent loads for 1024 parts); the 64 groups meet in LDS and thread c adds
// them in order 0..63; the result is stored (or added to what the gradient arena holds).
__global__ __launch_bounds__(256) void ln_bwd_reduce(const float* __restrict__ ws, float* __restrict__ dw, float* __restrict__ db,
                                                     int parts, int dim, int accumulate) {
  __shared__ float red[64][4];
  const int cl = threadIdx.x & 3, g = threadIdx.x >> 2;
  const int c = blockIdx.x * 4 + cl;
  float s = 0.f;
  if (c < 2 * dim)
    for (int p = g; p < parts; p += 64) s += ws[(size_t)p * 2 * dim + c];
  red[g][cl] = s;
  __syncthreads();
  if (g == 0 && c < 2 * dim) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 64; ++k) v += red[k][cl];
    float* dst = c < dim ? &dw[c] : &db[c - dim];
    *dst = (accumulate ? *dst : 0.f) + v;
  }
}

static int ln_parts(int64_t rows) {
  int64_t p = ceil_div(rows, LN_THREADS / 64);
  return (int)(p < LN_MAX_PARTS ? p : LN_MAX_PARTS);
}

// gemm_bf16_pp384.hip: the input-gradient product into width 384 with LayerNorm backward as its epilogue
int pp384_lnbwd_tiles(int64_t M);
bool gemm_bf16_nt_pp384_ln_ok(int64_t M, int K);
int launch_gemm_bf16_nt_pp384_lnbwd(const void* a, const void* w, const float* x, const float* gamma, const float* mean, const float* rstd,
                                    float* dx, const float* dx_add, void* dx_lowp, float* ws, int64_t M, int K, hipStream_t st);

}  // namespace dinox

using namespace dinox;

extern "C" int dinox_layernorm_fwd(const float* x, const float* w, const float* b, void* y, float* mean, float* rstd,
                                   int64_t rows, int dim, float eps, int out_dtype, void* stream) {
  DX_REQUIRE(x && w && b && y && mean && rstd, DINOX_EINVAL, "layernorm_fwd: null pointer");
  DX_REQUIRE(rows > 0 && dim > 0, DINOX_EINVAL, "layernorm_fwd: rows=%lld dim=%d", (long long)rows, dim);
  DX_REQUIRE(out_dtype == DINOX_F32 || out_dtype == DINOX_BF16, DINOX_EINVAL, "layernorm_fwd: dtype %d", out_dtype);
  int64_t blocks = ceil_div(rows, LN_THREADS / 64);
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipStream_t st = as_stream(stream);
  const bool fast = (dim % 4 == 0) && dim <= 256 * LN_MAXV;
  static const bool no384 = getenv("DINOX_LN_NO384") != nullptr;                     // A/B knob
  if (dim == 384 && !no384 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)w | (uintptr_t)b) & 15) == 0) {
    int64_t blk = ceil_div(rows, (int64_t)(LN_THREADS / 64) * 2 * 2);             // two rows per half wave and iteration
    if (blk > 256 * 8) blk = 256 * 8;
    if (out_dtype == DINOX_F32)
      hipLaunchKernelGGL((ln_fwd_384<DINOX_F32>), dim3((unsigned)blk), dim3(LN_THREADS), 0, st, x, w, b, y, mean, rstd, rows, eps);
    else
      hipLaunchKernelGGL((ln_fwd_384<DINOX_BF16>), dim3((unsigned)blk), dim3(LN_THREADS), 0, st, x, w, b, y, mean, rstd, rows, eps);
    return check_launch("layernorm_fwd");
  }
#define LN_FWD(DT, NV) hipLaunchKernelGGL((ln_fwd_kernel<DT, NV>), dim3((unsigned)blocks), dim3(LN_THREADS), 0, st, x, w, b, y, mean, rstd, rows, dim, eps)
  if (fast) {
    const int nv = (int)ceil_div(dim / 4, 64);
    if (out_dtype == DINOX_F32) {
      if (nv <= 1) LN_FWD(DINOX_F32, 1); else if (nv <= 2) LN_FWD(DINOX_F32, 2); else if (nv <= 4) LN_FWD(DINOX_F32, 4); else LN_FWD(DINOX_F32, 8);
    } else {
      if (nv <= 1) LN_FWD(DINOX_BF16, 1); else if (nv <= 2) LN_FWD(DINOX_BF16, 2); else if (nv <= 4) LN_FWD(DINOX_BF16, 4); else LN_FWD(DINOX_BF16, 8);
    }
  } else {
    if (out_dtype == DINOX_F32)
      hipLaunchKernelGGL((ln_fwd_generic<DINOX_F32>), dim3((unsigned)blocks), dim3(LN_THREADS), 0, st, x, w, b, y, mean, rstd, rows, dim, eps);
    else
      hipLaunchKernelGGL((ln_fwd_generic<DINOX_BF16>), dim3((unsigned)blocks), dim3(LN_THREADS), 0, st, x, w, b, y, mean, rstd, rows, dim, eps);
  }
#undef LN_FWD
  return check_launch("layernorm_fwd");
}

extern "C" int64_t dinox_layernorm_bwd_ws_bytes(int64_t rows, int dim) {
  if (rows <= 0 || dim <= 0) return 0;
  return (int64_t)ln_parts(rows) * 2 * dim * (int64_t)sizeof(float);
}

extern "C" int dinox_layernorm_bwd(const void* dy, const float* x, const float* w, const float* mean, const float* rstd,
                                   float* dx, const float* dx_add, void* dx_lowp, float* dw, float* db, void* ws,
                                   int64_t rows, int dim, int dy_dtype, int accumulate, void* stream) {
  DX_REQUIRE(dy && x && w && mean && rstd && dx && dw && db && ws, DINOX_EINVAL, "layernorm_bwd: null pointer");
  DX_REQUIRE(rows > 0 && dim > 0, DINOX_EINVAL, "layernorm_bwd: rows=%lld dim=%d", (long long)rows, dim);
  DX_REQUIRE(dy_dtype == DINOX_F32 || dy_dtype == DINOX_BF16, DINOX_EINVAL, "layernorm_bwd: dtype %d", dy_dtype);
  const int parts = ln_parts(rows);
  hipStream_t st = as_stream(stream);
  const size_t lds = (size_t)(LN_THREADS / 64) * 2 * dim * sizeof(float);
  const bool fast = (dim % 4 == 0) && dim <= 256 * LN_MAXV && lds <= 64 * 1024;
  float* wsf = (float*)ws;
  static const bool no384 = getenv("DINOX_LN_NO384") != nullptr;                     // A/B knob
  const bool al16 = (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)dx_add | (uintptr_t)dx_lowp | (uintptr_t)w) & 15) == 0;
  if (dim == 384 && !no384 && al16) {
    // 167 registers = three waves per SIMD = three blocks per CU: 768 blocks are one resident round (the workspace holds `parts` >= that)
    const int64_t want = ceil_div(rows, (int64_t)(LN_THREADS / 64) * 2 * 2);
    const int nblk = (int)(want < parts ? want : (parts < 768 ? parts : 768));
    if (dy_dtype == DINOX_F32)
      hipLaunchKernelGGL((ln_bwd_384<DINOX_F32>), dim3(nblk), dim3(LN_THREADS), lds, st, dy, x, w, mean, rstd, dx, dx_lowp, wsf, rows, dx_add);
    else
      hipLaunchKernelGGL((ln_bwd_384<DINOX_BF16>), dim3(nblk), dim3(LN_THREADS), lds, st, dy, x, w, mean, rstd, dx, dx_lowp, wsf, rows, dx_add);
    if (int rc = check_launch("layernorm_bwd")) return rc;
    hipLaunchKernelGGL(ln_bwd_reduce, dim3((unsigned)ceil_div(2 * dim, 4)), dim3(256), 0, st, wsf, dw, db, nblk, dim, accumulate ? 1 : 0);
    return check_launch("layernorm_bwd_reduce");
  }
#define LN_BWD(DT, NV) hipLaunchKernelGGL((ln_bwd_kernel<DT, NV>), dim3(parts), dim3(LN_THREADS), lds, st, dy, x, w, mean, rstd, dx, dx_lowp, wsf, rows, dim, dx_add)
  if (fast) {
    const int nv = (int)ceil_div(dim / 4, 64);
    if (dy_dtype == DINOX_F32) {
      if (nv <= 1) LN_BWD(DINOX_F32, 1); else if (nv <= 2) LN_BWD(DINOX_F32, 2); else if (nv <= 4) LN_BWD(DINOX_F32, 4); else LN_BWD(DINOX_F32, 8);
    } else {
      if (nv <= 1) LN_BWD(DINOX_BF16, 1); else if (nv <= 2) LN_BWD(DINOX_BF16, 2); else if (nv <= 4) LN_BWD(DINOX_BF16, 4); else LN_BWD(DINOX_BF16, 8);
    }
  } else {
    const size_t l2 = (size_t)2 * dim * sizeof(float);
    DX_REQUIRE(l2 <= 64 * 1024, DINOX_EUNSUPPORTED, "layernorm_bwd: dim %d too large", dim);
    if (dy_dtype == DINOX_F32)
      hipLaunchKernelGGL((ln_bwd_generic<DINOX_F32>), dim3(parts), dim3(LN_THREADS), l2, st, dy, x, w, mean, rstd, dx, dx_lowp, wsf, rows, dim, dx_add);
    else
      hipLaunchKernelGGL((ln_bwd_generic<DINOX_BF16>), dim3(parts), dim3(LN_THREADS), l2, st, dy, x, w, mean, rstd, dx, dx_lowp, wsf, rows, dim, dx_add);
  }
#undef LN_BWD
  int rc = check_launch("layernorm_bwd");
  if (rc) return rc;
  hipLaunchKernelGGL(ln_bwd_reduce, dim3((unsigned)ceil_div(2 * dim, 4)), dim3(256), 0, st, wsf, dw, db, parts, dim, accumulate ? 1 : 0);
  return check_launch("layernorm_bwd_reduce");
}

// dX product + LayerNorm backward in one launch (width 384, bf16 operands):  dy = a w^T rounded to bf16 (what the two launches hand
// over), dx = (dx_add ? dx_add : 0) + LN'(dy) -- equal to dinox_gemm + dinox_layernorm_bwd to the last bit --, dgamma / dbeta through the same
// workspace and reduction (ws: dinox_layernorm_bwd_ws_bytes(M, 384) bytes).  dx may alias dx_add.
extern "C" int dinox_linear_ln_bwd_ok(int64_t M, int N, int K) {
  return (N == 384 && gemm_bf16_nt_pp384_ln_ok(M, K) && pp384_lnbwd_tiles(M) <= ln_parts(M)) ? 1 : 0;
}

extern "C" int dinox_linear_ln_bwd(const void* a, const void* w, const float* x, const float* gamma, const float* mean, const float* rstd,
                                   float* dx, const float* dx_add, void* dx_lowp, float* dgamma, float* dbeta, void* ws, int64_t M, int N,
                                   int K, int accumulate, void* stream) {
  DX_REQUIRE(a && w && x && gamma && mean && rstd && dx && dgamma && dbeta && ws, DINOX_EINVAL, "linear_ln_bwd: null pointer");
  DX_REQUIRE(dinox_linear_ln_bwd_ok(M, N, K), DINOX_EUNSUPPORTED, "linear_ln_bwd: M=%lld N=%d K=%d", (long long)M, N, K);
  DX_REQUIRE((((uintptr_t)a | (uintptr_t)w | (uintptr_t)x | (uintptr_t)dx | (uintptr_t)dx_add | (uintptr_t)dx_lowp | (uintptr_t)gamma) & 15) == 0,
             DINOX_EALIGN, "linear_ln_bwd: operands must be 16-byte aligned");
  hipStream_t st = as_stream(stream);
  if (int rc = launch_gemm_bf16_nt_pp384_lnbwd(a, w, x, gamma, mean, rstd, dx, dx_add, dx_lowp, (float*)ws, M, K, st)) return rc;
  hipLaunchKernelGGL(ln_bwd_reduce, dim3((unsigned)ceil_div(2 * 384, 4)), dim3(256), 0, st, (const float*)ws, dgamma, dbeta, pp384_lnbwd_tiles(M), 384,
                     accumulate ? 1 : 0);
  return check_launch("linear_ln_bwd_reduce");
}
