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

int launch_gemm_f32(const GemmParams& p, hipStream_t st) {
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
