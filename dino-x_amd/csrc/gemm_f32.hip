// gemm_f32.hip -- general strided GEMM on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).
// "Parity mode" product: every operand layout (NT/TN/NN/TT via element strides), any M/N/K, fp32 or
// bf16 inputs (converted to fp32 on load; products of bf16 values are exact in fp32, so with bf16
// inputs this kernel is also the on-GPU reference for the bf16 MFMA kernels in gemm_bf16.hip).
// The result is bit-for-bit a k-ordered fmaf chain per output (MI355X_MICROARCH.md, Matrix cores),
// 1/16 of the bf16 MFMA rate -- it is the accuracy path, not the throughput path.
#include "common.h"
#include "gemm_common.h"

namespace dinox {

constexpr int GF_BM = 64, GF_BN = 64, GF_BK = 16, GF_THREADS = 256;

template <int IN_DT>
__device__ __forceinline__ void gf_load_tile(const void* __restrict__ base, int64_t s_row, int64_t s_k, int64_t row0,
                                             int64_t k0, int64_t rows, int64_t K, float (*dst)[GF_BM + 4]) {
  // 64 rows x 16 k = 1024 elements, 4 per thread.  Map consecutive threads along the contiguous axis.
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int r, k;
    if (s_k == 1) {
      k = t & 15;
      r = (t >> 4) + 16 * i;
    } else {
      r = t & 63;
      k = (t >> 6) + 4 * i;
    }
    const int64_t gr = row0 + r, gk = k0 + k;
    float v = 0.f;
    if (gr < rows && gk < K) v = elem<IN_DT>::ld(base, gr * s_row + gk * s_k);
    dst[k][r] = v;
  }
}

template <int IN_DT, int OUT_DT>
__global__ __launch_bounds__(GF_THREADS) void gemm_f32_kernel(GemmParams p) {
  __shared__ float As[GF_BK][GF_BM + 4];
  __shared__ float Bs[GF_BK][GF_BN + 4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wr = wv >> 1, wc = wv & 1;
  const int64_t m0 = (int64_t)blockIdx.y * GF_BM, n0 = (int64_t)blockIdx.x * GF_BN;
  const int64_t bz = blockIdx.z;
  const char* A = (const char*)p.A + bz * p.strideA * (IN_DT == DINOX_F32 ? 4 : 2);
  const char* B = (const char*)p.B + bz * p.strideB * (IN_DT == DINOX_F32 ? 4 : 2);
  // element strides: A(m,k), B(n,k)
  const int64_t a_sm = p.transA ? 1 : p.lda, a_sk = p.transA ? p.lda : 1;
  const int64_t b_sn = p.transB ? 1 : p.ldb, b_sk = p.transB ? p.ldb : 1;

  f32x16 acc;
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;

  for (int64_t k0 = 0; k0 < p.K; k0 += GF_BK) {
    gf_load_tile<IN_DT>(A, a_sm, a_sk, m0, k0, p.M, p.K, As);
    gf_load_tile<IN_DT>(B, b_sn, b_sk, n0, k0, p.N, p.K, Bs);
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < GF_BK; kk += 2) {
      const float a = As[kk + (lane >> 5)][wr * 32 + (lane & 31)];
      const float b = Bs[kk + (lane >> 5)][wc * 32 + (lane & 31)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }

  const int64_t n = n0 + wc * 32 + (lane & 31);
  if (n >= p.N) return;
  const float bias = (p.epilogue & DINOX_EPI_BIAS) ? p.bias[n] : 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int64_t m = m0 + wr * 32 + (j & 3) + 8 * (j >> 2) + 4 * (lane >> 5);
    if (m >= p.M) continue;
    epilogue_store<OUT_DT>(p, bz, m, n, acc[j], bias);
  }
}

// ------------------------------------------------------------------------------------------ 128 x 128 tiles
// Same arithmetic (per output a k-ordered chain of exact-fp32 MFMA steps: results are bit-identical to the kernel above), four times
// the work per LDS fragment: each of the four waves owns 64 x 64 (2 x 2 MFMA blocks), so a K-step of two reads 2 + 2 fragments for four
// products instead of 1 + 1 for one; the next K-slab (16) is fetched into registers while this one is multiplied.  Used when both
// output extents reach 128 -- the Linear products of the fp32 parity mode (17 -> ~60 TFLOP/s); small products keep 64 x 64 tiles.
constexpr int GB_M = 128, GB_N = 128, GB_K = 16;

template <int IN_DT>
__device__ __forceinline__ void gb_fetch(const void* __restrict__ base, int64_t s_row, int64_t s_k, int64_t row0, int64_t k0, int64_t rows,
                                         int64_t K, float (&v)[8]) {
  const int t = threadIdx.x;                       // 128 rows x 16 k = 2048 elements, 8 per thread, consecutive threads along the contiguous axis
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int r, k;
    if (s_k == 1) {
      k = t & 15;
      r = (t >> 4) + 16 * i;
    } else {
      r = t & 127;
      k = (t >> 7) + 2 * i;
    }
    const int64_t gr = row0 + r, gk = k0 + k;
    v[i] = (gr < rows && gk < K) ? elem<IN_DT>::ld(base, gr * s_row + gk * s_k) : 0.f;
  }
}

__device__ __forceinline__ void gb_put(bool k_contig, const float (&v)[8], float (*dst)[GB_M + 4]) {
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (k_contig) dst[t & 15][(t >> 4) + 16 * i] = v[i];
    else dst[(t >> 7) + 2 * i][t & 127] = v[i];
  }
}

template <int IN_DT, int OUT_DT>
__global__ __launch_bounds__(GF_THREADS) void gemm_f32_big(GemmParams p) {
  __shared__ float As[GB_K][GB_M + 4];
  __shared__ float Bs[GB_K][GB_N + 4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wr = wv >> 1, wc = wv & 1;
  const int64_t m0 = (int64_t)blockIdx.y * GB_M, n0 = (int64_t)blockIdx.x * GB_N;
  const int64_t bz = blockIdx.z;
  const char* A = (const char*)p.A + bz * p.strideA * (IN_DT == DINOX_F32 ? 4 : 2);
  const char* B = (const char*)p.B + bz * p.strideB * (IN_DT == DINOX_F32 ? 4 : 2);
  const int64_t a_sm = p.transA ? 1 : p.lda, a_sk = p.transA ? p.lda : 1;
  const int64_t b_sn = p.transB ? 1 : p.ldb, b_sk = p.transB ? p.ldb : 1;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  float va[8], vb[8];
  gb_fetch<IN_DT>(A, a_sm, a_sk, m0, 0, p.M, p.K, va);
  gb_fetch<IN_DT>(B, b_sn, b_sk, n0, 0, p.N, p.K, vb);
  for (int64_t k0 = 0; k0 < p.K; k0 += GB_K) {
    gb_put(a_sk == 1, va, As);
    gb_put(b_sk == 1, vb, Bs);
    __syncthreads();
    if (k0 + GB_K < p.K) {                                   // next slab: in flight under the products below
      gb_fetch<IN_DT>(A, a_sm, a_sk, m0, k0 + GB_K, p.M, p.K, va);
      gb_fetch<IN_DT>(B, b_sn, b_sk, n0, k0 + GB_K, p.N, p.K, vb);
    }
#pragma unroll
    for (int kk = 0; kk < GB_K; kk += 2) {
      const int kr = kk + (lane >> 5), c = lane & 31;
      const float a0 = As[kr][wr * 64 + c], a1 = As[kr][wr * 64 + 32 + c];
      const float b0 = Bs[kr][wc * 64 + c], b1 = Bs[kr][wc * 64 + 32 + c];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int64_t n = n0 + wc * 64 + j * 32 + (lane & 31);
    if (n >= p.N) continue;
    const float bias = (p.epilogue & DINOX_EPI_BIAS) ? p.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t m = m0 + wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (m < p.M) epilogue_store<OUT_DT>(p, bz, m, n, acc[i][j][e], bias);
      }
  }
}

int launch_gemm_f32(const GemmParams& p, hipStream_t st) {
  // (big tiles only when there is at least one per CU)
  if (p.M >= GB_M && p.N >= GB_N && ceil_div(p.M, (int64_t)GB_M) * ceil_div(p.N, (int64_t)GB_N) * p.batch >= 256) {
    dim3 gridb((unsigned)ceil_div(p.N, (int64_t)GB_N), (unsigned)ceil_div(p.M, (int64_t)GB_M), (unsigned)p.batch);
#define GFB(IN, OUT) hipLaunchKernelGGL((gemm_f32_big<IN, OUT>), gridb, dim3(GF_THREADS), 0, st, p)
    if (p.in_dtype == DINOX_F32) {
      if (p.out_dtype == DINOX_F32) GFB(DINOX_F32, DINOX_F32); else GFB(DINOX_F32, DINOX_BF16);
    } else {
      if (p.out_dtype == DINOX_F32) GFB(DINOX_BF16, DINOX_F32); else GFB(DINOX_BF16, DINOX_BF16);
    }
#undef GFB
    return check_launch("gemm_f32_big");
  }
  dim3 grid((unsigned)ceil_div(p.N, GF_BN), (unsigned)ceil_div(p.M, GF_BM), (unsigned)p.batch);
#define GF(IN, OUT) hipLaunchKernelGGL((gemm_f32_kernel<IN, OUT>), grid, dim3(GF_THREADS), 0, st, p)
  if (p.in_dtype == DINOX_F32) {
    if (p.out_dtype == DINOX_F32) GF(DINOX_F32, DINOX_F32); else GF(DINOX_F32, DINOX_BF16);
  } else {
    if (p.out_dtype == DINOX_F32) GF(DINOX_BF16, DINOX_F32); else GF(DINOX_BF16, DINOX_BF16);
  }
#undef GF
  return check_launch("gemm_f32");
}

}  // namespace dinox
