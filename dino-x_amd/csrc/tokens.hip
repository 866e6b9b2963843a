// tokens.hip -- patch unfold (im2col for stride==kernel) and token assembly, forward and backward.
// Replaces the input side of nn.Conv2d(3,D,k=p,s=p) and the flatten/transpose/cat/+pos/+scale/cat
// sequence of the reference (zoo/arch.py:216-229).  All HBM-bound, coalesced along the innermost axis.
#include "common.h"

namespace dinox {

// u[(v*P + gy*g + gx)][c*p*p + py*p + px] = x[v][c][gy*p+py][gx*p+px].
// One thread per output element pair; consecutive threads walk px (contiguous in x and in u).
template <int DT>
__global__ __launch_bounds__(256) void unfold_kernel(const float* __restrict__ x, void* __restrict__ u, int V, int H,
                                                     int W, int p) {
  const int g = W / p, gh = H / p;
  const int Kd = 3 * p * p;
  const int64_t total = (int64_t)V * gh * g * Kd;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(idx % Kd);
    const int64_t row = idx / Kd;
    const int gx = (int)(row % g);
    const int gy = (int)((row / g) % gh);
    const int64_t v = row / ((int64_t)g * gh);
    const int px = k % p, py = (k / p) % p, c = k / (p * p);
    const float val = x[((v * 3 + c) * H + (gy * p + py)) * (int64_t)W + gx * p + px];
    elem<DT>::st(u, idx, val);
  }
}

// Vector form for p % 8 == 0: one thread moves 8 consecutive px (two float4 loads, one 16-B bf16 or two float4 stores).
template <int DT>
__global__ __launch_bounds__(256) void unfold_vec8_kernel(const float* __restrict__ x, void* __restrict__ u, int V, int H, int W, int p) {
  const int g = W / p, gh = H / p, p8 = p / 8;
  const int Kd8 = 3 * p * p8;                       // 8-element groups per output row
  const int64_t total = (int64_t)V * gh * g * Kd8;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int k8 = (int)(idx % Kd8);
    const int64_t row = idx / Kd8;
    const int gx = (int)(row % g);
    const int64_t t2 = row / g;
    const int gy = (int)(t2 % gh);
    const int64_t v = t2 / gh;
    const int px8 = k8 % p8, py = (k8 / p8) % p, c = k8 / (p8 * p);
    const float* src = x + ((v * 3 + c) * H + (gy * p + py)) * (int64_t)W + gx * p + px8 * 8;
    const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
    const int64_t o = row * (int64_t)(3 * p * p) + (int64_t)(c * p + py) * p + px8 * 8;
    if (DT == DINOX_BF16) {
      s16x8 pk;
      pk[0] = (short)f32_to_bf16(a.x); pk[1] = (short)f32_to_bf16(a.y); pk[2] = (short)f32_to_bf16(a.z); pk[3] = (short)f32_to_bf16(a.w);
      pk[4] = (short)f32_to_bf16(b.x); pk[5] = (short)f32_to_bf16(b.y); pk[6] = (short)f32_to_bf16(b.z); pk[7] = (short)f32_to_bf16(b.w);
      *reinterpret_cast<s16x8*>((bf16_t*)u + o) = pk;
    } else {
      *reinterpret_cast<float4*>((float*)u + o) = a;
      *reinterpret_cast<float4*>((float*)u + o + 4) = b;
    }
  }
}

// tokens[v][n][:]:  n=0: cls+pos[0]+scale[v];  1<=n<=P: patches[v][n-1]+pos[n]+scale[v];  n>P: registers[n-1-P]
template <int DT>
__global__ __launch_bounds__(256) void tokens_fwd_kernel(const void* __restrict__ patches, const float* __restrict__ cls,
                                                         const float* __restrict__ pos, const float* __restrict__ regs,
                                                         const float* __restrict__ scale, float* __restrict__ tokens,
                                                         int V, int P, int R, int D) {
  const int N = 1 + P + R;
  const int64_t total = (int64_t)V * N * D;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int dd = (int)(idx % D);
    const int n = (int)((idx / D) % N);
    const int64_t v = idx / ((int64_t)D * N);
    float val;
    if (n > P) {
      val = regs[(int64_t)(n - 1 - P) * D + dd];
    } else {
      val = (n == 0) ? cls[dd] : elem<DT>::ld(patches, (v * P + (n - 1)) * D + dd);
      val += pos[(int64_t)n * D + dd];
      if (scale) val += scale[v * D + dd];
    }
    tokens[idx] = val;
  }
}

// Vector forms for D % 4 == 0: one thread per 4 consecutive features.
template <int DT>
__device__ __forceinline__ float4 ld4(const void* p, int64_t i) {
  if (DT == DINOX_BF16) {
    const uint2 w = *reinterpret_cast<const uint2*>((const bf16_t*)p + i);
    return make_float4(__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16), __uint_as_float(w.y & 0xffff0000u));
  }
  return *reinterpret_cast<const float4*>((const float*)p + i);
}
template <int DT>
__device__ __forceinline__ void st4(void* p, int64_t i, float4 v) {
  if (DT == DINOX_BF16) {
    uint2 w;
    w.x = (unsigned)f32_to_bf16(v.x) | ((unsigned)f32_to_bf16(v.y) << 16);
    w.y = (unsigned)f32_to_bf16(v.z) | ((unsigned)f32_to_bf16(v.w) << 16);
    *reinterpret_cast<uint2*>((bf16_t*)p + i) = w;
  } else {
    *reinterpret_cast<float4*>((float*)p + i) = v;
  }
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

template <int DT>
__global__ __launch_bounds__(256) void tokens_fwd_vec4_kernel(const void* __restrict__ patches, const float* __restrict__ cls,
                                                              const float* __restrict__ pos, const float* __restrict__ regs,
                                                              const float* __restrict__ scale, float* __restrict__ tokens,
                                                              int V, int P, int R, int D) {
  const int N = 1 + P + R, D4 = D / 4;
  const int64_t total = (int64_t)V * N * D4;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int dd = (int)(idx % D4) * 4;
    const int64_t t2 = idx / D4;
    const int n = (int)(t2 % N);
    const int64_t v = t2 / N;
    float4 val;
    if (n > P) {
      val = *reinterpret_cast<const float4*>(regs + (int64_t)(n - 1 - P) * D + dd);
    } else {
      val = (n == 0) ? *reinterpret_cast<const float4*>(cls + dd) : ld4<DT>(patches, (v * P + (n - 1)) * D + dd);
      val = add4(val, *reinterpret_cast<const float4*>(pos + (int64_t)n * D + dd));
      if (scale) val = add4(val, *reinterpret_cast<const float4*>(scale + v * D + dd));
    }
    *reinterpret_cast<float4*>(tokens + (v * N + n) * D + dd) = val;
  }
}

template <int DT>
__global__ __launch_bounds__(256) void tokens_bwd_patches_vec4(const float* __restrict__ dt, void* __restrict__ dpatches, int V, int P,
                                                               int R, int D) {
  const int N = 1 + P + R, D4 = D / 4;
  const int64_t total = (int64_t)V * P * D4;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int dd = (int)(idx % D4) * 4;
    const int64_t t2 = idx / D4;
    const int i = (int)(t2 % P);
    const int64_t v = t2 / P;
    st4<DT>(dpatches, (v * P + i) * D + dd, *reinterpret_cast<const float4*>(dt + (v * N + 1 + i) * D + dd));
  }
}

// Batch reduction, vector form: a workgroup owns one token position n and 64 float4 feature groups; its four 64-thread
// slices sum interleaved quarters of the V images (independent float4 loads, 4 in flight per thread) and meet in LDS in a
// fixed order, so the result is deterministic.
__global__ __launch_bounds__(256) void tokens_bwd_params_vec4(const float* __restrict__ dt, float* __restrict__ dcls,
                                                              float* __restrict__ dpos, float* __restrict__ dregs, int V, int P, int R,
                                                              int D) {
  __shared__ float4 part[4][64];
  const int N = 1 + P + R, D4 = D / 4;
  const int n = blockIdx.x, c4 = blockIdx.y * 64 + (threadIdx.x & 63), slice = threadIdx.x >> 6;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c4 < D4) {
    const float* base = dt + (int64_t)n * D + c4 * 4;
    const int64_t stride = (int64_t)N * D;
    int v = slice;
    for (; v + 12 < V; v += 16) {
      const float4 a = *reinterpret_cast<const float4*>(base + (int64_t)v * stride);
      const float4 b = *reinterpret_cast<const float4*>(base + (int64_t)(v + 4) * stride);
      const float4 c = *reinterpret_cast<const float4*>(base + (int64_t)(v + 8) * stride);
      const float4 d = *reinterpret_cast<const float4*>(base + (int64_t)(v + 12) * stride);
      s = add4(s, add4(add4(a, b), add4(c, d)));
    }
    for (; v < V; v += 4) s = add4(s, *reinterpret_cast<const float4*>(base + (int64_t)v * stride));
  }
  part[slice][threadIdx.x & 63] = s;
  __syncthreads();
  if (slice == 0 && c4 < D4) {
    const float4 r = add4(add4(part[0][threadIdx.x], part[1][threadIdx.x]), add4(part[2][threadIdx.x], part[3][threadIdx.x]));
    const int dd = c4 * 4;
    if (n > P) {
      if (dregs) *reinterpret_cast<float4*>(dregs + (int64_t)(n - 1 - P) * D + dd) = r;
    } else {
      *reinterpret_cast<float4*>(dpos + (int64_t)n * D + dd) = r;
      if (n == 0) *reinterpret_cast<float4*>(dcls + dd) = r;
    }
  }
}

// dpatches[v][i] = dtokens[v][1+i]  (cast to the GEMM dtype)
template <int DT>
__global__ __launch_bounds__(256) void tokens_bwd_patches(const float* __restrict__ dt, void* __restrict__ dpatches, int V,
                                                          int P, int R, int D) {
  const int N = 1 + P + R;
  const int64_t total = (int64_t)V * P * D;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int dd = (int)(idx % D);
    const int i = (int)((idx / D) % P);
    const int64_t v = idx / ((int64_t)D * P);
    elem<DT>::st(dpatches, idx, dt[(v * N + 1 + i) * D + dd]);
  }
}

// Batch reductions: one thread per (n, dd); dpos[n] = sum_v dt[v][n] (n<=P), dcls = dpos-like for n=0,
// dregs[n-1-P] = sum_v dt[v][n] (n>P).  V-loop reads are coalesced across dd.
__global__ __launch_bounds__(256) void tokens_bwd_params(const float* __restrict__ dt, float* __restrict__ dcls,
                                                         float* __restrict__ dpos, float* __restrict__ dregs, int V, int P,
                                                         int R, int D) {
  const int N = 1 + P + R;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)N * D) return;
  const int dd = (int)(idx % D);
  const int n = (int)(idx / D);
  float s = 0.f;
  for (int v = 0; v < V; ++v) s += dt[((int64_t)v * N + n) * D + dd];
  if (n > P) {
    if (dregs) dregs[(int64_t)(n - 1 - P) * D + dd] = s;
  } else {
    dpos[(int64_t)n * D + dd] = s;
    if (n == 0) dcls[dd] = s;
  }
}

// dscale[v][dd] = sum_{n<=P} dt[v][n][dd]
__global__ __launch_bounds__(256) void tokens_bwd_scale(const float* __restrict__ dt, float* __restrict__ dscale, int V,
                                                        int P, int R, int D) {
  const int N = 1 + P + R;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)V * D) return;
  const int dd = (int)(idx % D);
  const int64_t v = idx / D;
  float s = 0.f;
  for (int n = 0; n <= P; ++n) s += dt[(v * N + n) * D + dd];
  dscale[idx] = s;
}

static unsigned grid_for(int64_t total) {
  int64_t b = ceil_div(total, 256);
  return (unsigned)(b < 256 * 16 ? b : 256 * 16);
}

}  // namespace dinox

using namespace dinox;

// Rows of `ld` >= 3 p^2 elements, the tail zero-filled: 3 x 14 x 14 = 588 columns are no multiple of 8, so the MFMA bf16 products
// do not take the plain matrix; padded to 640 they do (the padded weight columns are zero as well).
template <int DT>
__global__ __launch_bounds__(256) void unfold_ld_kernel(const float* __restrict__ x, void* __restrict__ u, int V, int H, int W, int p, int ld) {
  const int g = W / p, gh = H / p;
  const int Kd = 3 * p * p;
  const int64_t total = (int64_t)V * gh * g * ld;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(idx % ld);
    const int64_t row = idx / ld;
    float val = 0.f;
    if (k < Kd) {
      const int gx = (int)(row % g);
      const int gy = (int)((row / g) % gh);
      const int64_t v = row / ((int64_t)g * gh);
      const int px = k % p, py = (k / p) % p, c = k / (p * p);
      val = x[((v * 3 + c) * H + (gy * p + py)) * (int64_t)W + gx * p + px];
    }
    elem<DT>::st(u, idx, val);
  }
}

extern "C" int dinox_patch_unfold_ld(const float* x, void* u, int V, int H, int W, int patch, int ld, int out_dtype, void* stream) {
  DX_REQUIRE(x && u, DINOX_EINVAL, "patch_unfold_ld: null pointer");
  DX_REQUIRE(V > 0 && H > 0 && W > 0 && patch > 0 && H % patch == 0 && W % patch == 0 && ld >= 3 * patch * patch, DINOX_EINVAL,
             "patch_unfold_ld: V=%d H=%d W=%d patch=%d ld=%d", V, H, W, patch, ld);
  DX_REQUIRE(out_dtype == DINOX_F32 || out_dtype == DINOX_BF16, DINOX_EINVAL, "patch_unfold_ld: dtype %d", out_dtype);
  const int64_t total = (int64_t)V * (H / patch) * (W / patch) * ld;
  hipStream_t st = as_stream(stream);
  if (out_dtype == DINOX_F32)
    hipLaunchKernelGGL((unfold_ld_kernel<DINOX_F32>), dim3(grid_for(total)), dim3(256), 0, st, x, u, V, H, W, patch, ld);
  else
    hipLaunchKernelGGL((unfold_ld_kernel<DINOX_BF16>), dim3(grid_for(total)), dim3(256), 0, st, x, u, V, H, W, patch, ld);
  return check_launch("patch_unfold_ld");
}

extern "C" int dinox_patch_unfold(const float* x, void* u, int V, int H, int W, int patch, int out_dtype, void* stream) {
  DX_REQUIRE(x && u, DINOX_EINVAL, "patch_unfold: null pointer");
  DX_REQUIRE(V > 0 && H > 0 && W > 0 && patch > 0 && H % patch == 0 && W % patch == 0, DINOX_EINVAL,
             "patch_unfold: V=%d H=%d W=%d patch=%d", V, H, W, patch);
  DX_REQUIRE(out_dtype == DINOX_F32 || out_dtype == DINOX_BF16, DINOX_EINVAL, "patch_unfold: dtype %d", out_dtype);
  const int64_t total = (int64_t)V * 3 * H * W;
  hipStream_t st = as_stream(stream);
  const bool vec = patch % 8 == 0 && (((uintptr_t)x | (uintptr_t)u) & 15) == 0;
  if (vec) {
    if (out_dtype == DINOX_F32)
      hipLaunchKernelGGL((unfold_vec8_kernel<DINOX_F32>), dim3(grid_for(total / 8)), dim3(256), 0, st, x, u, V, H, W, patch);
    else
      hipLaunchKernelGGL((unfold_vec8_kernel<DINOX_BF16>), dim3(grid_for(total / 8)), dim3(256), 0, st, x, u, V, H, W, patch);
  } else if (out_dtype == DINOX_F32)
    hipLaunchKernelGGL((unfold_kernel<DINOX_F32>), dim3(grid_for(total)), dim3(256), 0, st, x, u, V, H, W, patch);
  else
    hipLaunchKernelGGL((unfold_kernel<DINOX_BF16>), dim3(grid_for(total)), dim3(256), 0, st, x, u, V, H, W, patch);
  return check_launch("patch_unfold");
}

extern "C" int dinox_tokens_fwd(const void* patches, const float* cls, const float* pos, const float* registers,
                                const float* scale, float* tokens, int V, int P, int R, int D, int patches_dtype,
                                void* stream) {
  DX_REQUIRE(patches && cls && pos && tokens, DINOX_EINVAL, "tokens_fwd: null pointer");
  DX_REQUIRE(V > 0 && P > 0 && R >= 0 && D > 0 && (R == 0 || registers), DINOX_EINVAL, "tokens_fwd: V=%d P=%d R=%d D=%d", V, P, R, D);
  DX_REQUIRE(patches_dtype == DINOX_F32 || patches_dtype == DINOX_BF16, DINOX_EINVAL, "tokens_fwd: dtype %d", patches_dtype);
  const int64_t total = (int64_t)V * (1 + P + R) * D;
  hipStream_t st = as_stream(stream);
  const bool vec = D % 4 == 0 && (((uintptr_t)patches | (uintptr_t)cls | (uintptr_t)pos | (uintptr_t)registers | (uintptr_t)scale |
                                   (uintptr_t)tokens) & 15) == 0;
  if (vec) {
    if (patches_dtype == DINOX_F32)
      hipLaunchKernelGGL((tokens_fwd_vec4_kernel<DINOX_F32>), dim3(grid_for(total / 4)), dim3(256), 0, st, patches, cls, pos, registers, scale, tokens, V, P, R, D);
    else
      hipLaunchKernelGGL((tokens_fwd_vec4_kernel<DINOX_BF16>), dim3(grid_for(total / 4)), dim3(256), 0, st, patches, cls, pos, registers, scale, tokens, V, P, R, D);
  } else if (patches_dtype == DINOX_F32)
    hipLaunchKernelGGL((tokens_fwd_kernel<DINOX_F32>), dim3(grid_for(total)), dim3(256), 0, st, patches, cls, pos, registers, scale, tokens, V, P, R, D);
  else
    hipLaunchKernelGGL((tokens_fwd_kernel<DINOX_BF16>), dim3(grid_for(total)), dim3(256), 0, st, patches, cls, pos, registers, scale, tokens, V, P, R, D);
  return check_launch("tokens_fwd");
}

extern "C" int dinox_tokens_bwd(const float* dtokens, void* dpatches, float* dcls, float* dpos, float* dregs,
                                float* dscale, int V, int P, int R, int D, int patches_dtype, void* stream) {
  DX_REQUIRE(dtokens && dpatches && dcls && dpos, DINOX_EINVAL, "tokens_bwd: null pointer");
  DX_REQUIRE(V > 0 && P > 0 && R >= 0 && D > 0 && (R == 0 || dregs), DINOX_EINVAL, "tokens_bwd: V=%d P=%d R=%d D=%d", V, P, R, D);
  DX_REQUIRE(patches_dtype == DINOX_F32 || patches_dtype == DINOX_BF16, DINOX_EINVAL, "tokens_bwd: dtype %d", patches_dtype);
  hipStream_t st = as_stream(stream);
  const int64_t tp = (int64_t)V * P * D;
  const bool vec = D % 4 == 0 && (((uintptr_t)dtokens | (uintptr_t)dpatches | (uintptr_t)dcls | (uintptr_t)dpos | (uintptr_t)dregs) & 15) == 0;
  if (vec) {
    if (patches_dtype == DINOX_F32)
      hipLaunchKernelGGL((tokens_bwd_patches_vec4<DINOX_F32>), dim3(grid_for(tp / 4)), dim3(256), 0, st, dtokens, dpatches, V, P, R, D);
    else
      hipLaunchKernelGGL((tokens_bwd_patches_vec4<DINOX_BF16>), dim3(grid_for(tp / 4)), dim3(256), 0, st, dtokens, dpatches, V, P, R, D);
    hipLaunchKernelGGL(tokens_bwd_params_vec4, dim3((unsigned)(1 + P + R), (unsigned)ceil_div(D / 4, 64)), dim3(256), 0, st, dtokens, dcls, dpos, dregs,
                       V, P, R, D);
  } else {
    if (patches_dtype == DINOX_F32)
      hipLaunchKernelGGL((tokens_bwd_patches<DINOX_F32>), dim3(grid_for(tp)), dim3(256), 0, st, dtokens, dpatches, V, P, R, D);
    else
      hipLaunchKernelGGL((tokens_bwd_patches<DINOX_BF16>), dim3(grid_for(tp)), dim3(256), 0, st, dtokens, dpatches, V, P, R, D);
    hipLaunchKernelGGL(tokens_bwd_params, dim3((unsigned)ceil_div((int64_t)(1 + P + R) * D, 256)), dim3(256), 0, st, dtokens, dcls, dpos, dregs, V, P, R, D);
  }
  if (dscale)
    hipLaunchKernelGGL(tokens_bwd_scale, dim3((unsigned)ceil_div((int64_t)V * D, 256)), dim3(256), 0, st, dtokens, dscale, V, P, R, D);
  return check_launch("tokens_bwd");
}
