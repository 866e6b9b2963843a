// common.h -- shared device/host helpers for libdinox_hip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>
#include <string>

#include "../../include/dinox.h"

namespace dinox {

// ---------------------------------------------------------------- host: error plumbing
void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);
int check_launch(const char* what);  // hipGetLastError -> 0 or positive hipError_t
int reserve_lds(const void* kern, size_t bytes, const char* what);  // hipFuncSetAttribute(max dynamic LDS), once per (kernel, size)

#define DX_REQUIRE(cond, code, ...)                  \
  do {                                               \
    if (!(cond)) return ::dinox::fail((code), __VA_ARGS__); \
  } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
__host__ __device__ static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------- device: types
typedef uint16_t bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// Plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, keeps NaN a NaN) -- MI355X_MICROARCH correctness table.
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}

template <int DT>
struct elem;
template <>
struct elem<DINOX_F32> {
  using type = float;
  static __device__ __forceinline__ float ld(const void* p, int64_t i) { return ((const float*)p)[i]; }
  static __device__ __forceinline__ void st(void* p, int64_t i, float v) { ((float*)p)[i] = v; }
};
template <>
struct elem<DINOX_BF16> {
  using type = bf16_t;
  static __device__ __forceinline__ float ld(const void* p, int64_t i) { return bf16_to_f32(((const bf16_t*)p)[i]); }
  static __device__ __forceinline__ void st(void* p, int64_t i, float v) { ((bf16_t*)p)[i] = f32_to_bf16(v); }
};

// ---------------------------------------------------------------- device: streaming stores
// Big outputs that the NEXT kernel reads (or nobody reads soon) leave by non-temporal stores: they do not push the operand slices of the
// tiles still running out of L2, and the consumer finds them in the memory-side cache (measured on the LayerNorm epilogue of
// gemm_bf16_pp384.hip: the kernel reading y next 272 -> 263 us).  HIP's uint4 / float4 are structs: the builtin wants native vectors.
typedef unsigned dx_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned dx_u32x2 __attribute__((ext_vector_type(2)));
template <typename T>
__device__ __forceinline__ void store_stream(T* dst, const T& v) {
#ifdef DINOX_PLAIN_OUT_STORES
  *dst = v;
#else
  static_assert(sizeof(T) == 16 || sizeof(T) == 8, "store_stream: 8- or 16-byte values");
  if constexpr (sizeof(T) == 16) __builtin_nontemporal_store(__builtin_bit_cast(dx_u32x4, v), reinterpret_cast<dx_u32x4*>(dst));
  else __builtin_nontemporal_store(__builtin_bit_cast(dx_u32x2, v), reinterpret_cast<dx_u32x2*>(dst));
#endif
}

// ---------------------------------------------------------------- device: math
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * __expf(-0.5f * x * x) * 0.39894228040143268f;
}

// erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, far below bf16's 2^-9): 1 rcp + 1 exp + 6 fma,
// used by the bf16 MFMA epilogues where 64+ evaluations per lane would otherwise rival the MFMA time.
// Returns erf(z) and passes out e = exp(-z*z), which gelu' needs too.
__device__ __forceinline__ float erf_as(float z, float& e) {
  const float a = fabsf(z);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * a);
  e = __expf(-a * a);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  return copysignf(1.0f - poly * e, z);
}
__device__ __forceinline__ float gelu_fast(float x) {
  float e;
  return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752f, e));
}
// gelu and its derivative from ONE erf/exp evaluation
__device__ __forceinline__ void gelu_fast_both(float x, float& y, float& dy) {
  float e;
  const float h = 0.5f * (1.0f + erf_as(x * 0.70710678118654752f, e));
  y = x * h;
  dy = h + x * e * 0.39894228040143268f;
}
__device__ __forceinline__ float gelu_fast_grad(float x) {
  float e;  // e = exp(-x^2/2)
  const float er = erf_as(x * 0.70710678118654752f, e);
  return 0.5f * (1.0f + er) + x * e * 0.39894228040143268f;
}

// ---------------------------------------------------------------- device: reductions (wave64)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// Block-wide sum; every thread gets the result. `red` is >= 16 floats of LDS. blockDim.x multiple of 64.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = red[0];
  for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
  return t;
}

}  // namespace dinox
