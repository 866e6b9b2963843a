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

extern "C" int dinox_patch_unfold(const float* x, void* u, int V, int H, int W, int patch, int out_dtype, void* stream) {
  DX_REQUIRE(x && u, DINOX_EINVAL, "patch_unfold: null pointer");
  DX_REQUIRE(V > 0 && H > 0 && W > 0 && patch > 0 && H % patch == 0 && W % patch == 0, DINOX_EINVAL,
             "patch_unfold: V=%d H=%d W=%d patch=%d", V, H, W, patch);
  DX_REQUIRE(out_dtype == DINOX_F32 || out_dtype == DINOX_BF16, DINOX_EINVAL, "patch_unfold: dtype %d", out_dtype);
  const int64_t total = (int64_t)V * 3 * H * W;
  hipStream_t st = as_stream(stream);
  if (out_dtype == DINOX_F32)
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
  if (patches_dtype == DINOX_F32)
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
  if (patches_dtype == DINOX_F32)
    hipLaunchKernelGGL((tokens_bwd_patches<DINOX_F32>), dim3(grid_for(tp)), dim3(256), 0, st, dtokens, dpatches, V, P, R, D);
  else
    hipLaunchKernelGGL((tokens_bwd_patches<DINOX_BF16>), dim3(grid_for(tp)), dim3(256), 0, st, dtokens, dpatches, V, P, R, D);
  hipLaunchKernelGGL(tokens_bwd_params, dim3((unsigned)ceil_div((int64_t)(1 + P + R) * D, 256)), dim3(256), 0, st, dtokens, dcls, dpos, dregs, V, P, R, D);
  if (dscale)
    hipLaunchKernelGGL(tokens_bwd_scale, dim3((unsigned)ceil_div((int64_t)V * D, 256)), dim3(256), 0, st, dtokens, dscale, V, P, R, D);
  return check_launch("tokens_bwd");
}
