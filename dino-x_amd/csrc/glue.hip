// glue.hip -- the small data movements between the big kernels of a training step, so that NO framework elementwise kernel
// (slice copies, zero fills, scalar combines, gradient accumulation adds) runs on the hot path:
//   dinox_take_rows / dinox_put_rows   the CLS row of every view: features [V][N][D] fp32 -> head input [V][D] (bf16 / fp32), and the
//                                      head's input gradient back into row 0 of the feature gradient (zoo/arch.py:260-261 `feats[:, 0]`,
//                                      whose backward in the reference is a zero fill + slice copy + full-tensor add);
//   dinox_axpy                         y += alpha * x  (the KoLeo term's gradient joining the DINO term's, phase5_big_run.py:1764-1766);
//   dinox_lincomb3                     out = a + wb * b + wc * c  (loss = dino + gram_weight * gram + koleo_weight * koleo, :1755-1766);
//   dinox_zero                         asynchronous zero fill (gradient arena, once per optimiser step).
#include "common.h"
#include "gemm_common.h"

namespace dinox {

template <int DT>
__global__ __launch_bounds__(256) void take_rows_kernel(const float* __restrict__ src, void* __restrict__ dst, int64_t V, int64_t src_stride, int D,
                                                       int64_t dst_row0) {
  const int64_t n4 = V * (D >> 2);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int64_t v = i / (D >> 2);
    const int c = (int)(i - v * (D >> 2)) * 4;
    const float4 x = *reinterpret_cast<const float4*>(src + v * src_stride + c);
    const int64_t o = (dst_row0 + v) * D + c;
    if (DT == DINOX_F32) {
      *reinterpret_cast<float4*>((float*)dst + o) = x;
    } else {
      uint2 pk;
      pk.x = (unsigned)f32_to_bf16(x.x) | ((unsigned)f32_to_bf16(x.y) << 16);
      pk.y = (unsigned)f32_to_bf16(x.z) | ((unsigned)f32_to_bf16(x.w) << 16);
      *reinterpret_cast<uint2*>((bf16_t*)dst + o) = pk;
    }
  }
}

template <int DT>
__global__ __launch_bounds__(256) void put_rows_kernel(const void* __restrict__ src, float* __restrict__ dst, int64_t V, int64_t dst_stride, int D,
                                                      int64_t src_row0, int accumulate) {
  const int64_t n = V * D;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t v = i / D;
    const int c = (int)(i - v * D);
    float x = elem<DT>::ld(src, (src_row0 + v) * D + c);
    float* d = dst + v * dst_stride + c;
    if (accumulate) x += *d;
    *d = x;
  }
}

__global__ __launch_bounds__(256) void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float alpha, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] += alpha * x[i];
}

__global__ void lincomb3_kernel(const float* a, const float* b, const float* c, float wb, float wc, float* out) {
  if (threadIdx.x == 0) out[0] = a[0] + (b ? wb * b[0] : 0.f) + (c ? wc * c[0] : 0.f);
}

static unsigned glue_grid(int64_t n) {
  int64_t b = ceil_div(n, (int64_t)256);
  return (unsigned)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace dinox

using namespace dinox;

extern "C" int dinox_take_rows(const float* src, void* dst, int64_t V, int64_t src_stride, int D, int64_t dst_row0, int dst_dtype,
                               void* stream) {
  DX_REQUIRE(src && dst && V > 0 && D > 0 && (D & 3) == 0 && (src_stride & 3) == 0 && dst_row0 >= 0, DINOX_EINVAL,
             "take_rows: V=%lld D=%d stride=%lld", (long long)V, D, (long long)src_stride);
  DX_REQUIRE((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, DINOX_EALIGN, "take_rows: pointers must be 16-byte aligned");
  const unsigned g = glue_grid(V * (D >> 2));
  if (dst_dtype == DINOX_F32)
    hipLaunchKernelGGL((take_rows_kernel<DINOX_F32>), dim3(g), dim3(256), 0, as_stream(stream), src, dst, V, src_stride, D, dst_row0);
  else if (dst_dtype == DINOX_BF16)
    hipLaunchKernelGGL((take_rows_kernel<DINOX_BF16>), dim3(g), dim3(256), 0, as_stream(stream), src, dst, V, src_stride, D, dst_row0);
  else
    return fail(DINOX_EINVAL, "take_rows: dtype %d", dst_dtype);
  return check_launch("take_rows");
}

extern "C" int dinox_put_rows(const void* src, float* dst, int64_t V, int64_t dst_stride, int D, int64_t src_row0, int src_dtype,
                              int accumulate, void* stream) {
  DX_REQUIRE(src && dst && V > 0 && D > 0 && src_row0 >= 0, DINOX_EINVAL, "put_rows: V=%lld D=%d", (long long)V, D);
  const unsigned g = glue_grid(V * (int64_t)D);
  if (src_dtype == DINOX_F32)
    hipLaunchKernelGGL((put_rows_kernel<DINOX_F32>), dim3(g), dim3(256), 0, as_stream(stream), src, dst, V, dst_stride, D, src_row0, accumulate);
  else if (src_dtype == DINOX_BF16)
    hipLaunchKernelGGL((put_rows_kernel<DINOX_BF16>), dim3(g), dim3(256), 0, as_stream(stream), src, dst, V, dst_stride, D, src_row0, accumulate);
  else
    return fail(DINOX_EINVAL, "put_rows: dtype %d", src_dtype);
  return check_launch("put_rows");
}

extern "C" int dinox_axpy(float* y, const float* x, float alpha, int64_t n, void* stream) {
  DX_REQUIRE(y && x && n > 0, DINOX_EINVAL, "axpy: bad arguments");
  hipLaunchKernelGGL(axpy_kernel, dim3(glue_grid(n)), dim3(256), 0, as_stream(stream), y, x, alpha, n);
  return check_launch("axpy");
}

extern "C" int dinox_lincomb3(const float* a, const float* b, const float* c, float wb, float wc, float* out, void* stream) {
  DX_REQUIRE(a && out, DINOX_EINVAL, "lincomb3: null pointer");
  hipLaunchKernelGGL(lincomb3_kernel, dim3(1), dim3(64), 0, as_stream(stream), a, b, c, wb, wc, out);
  return check_launch("lincomb3");
}

extern "C" int dinox_zero(void* p, int64_t bytes, void* stream) {
  DX_REQUIRE(p && bytes > 0, DINOX_EINVAL, "zero: bad arguments");
  const hipError_t e = hipMemsetAsync(p, 0, (size_t)bytes, as_stream(stream));
  if (e != hipSuccess) return fail((int)e, "zero: %s", hipGetErrorString(e));
  return 0;
}
