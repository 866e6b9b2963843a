// gemm_common.h -- kernel-side view of dinox_gemm_args and the shared scalar epilogue.
#pragma once
#include "common.h"

namespace dinox {

struct GemmParams {
  const void* A;
  const void* B;
  void* C;
  int64_t M, N, K;
  int64_t lda, ldb, ldc;
  int64_t batch, strideA, strideB, strideC;
  int transA, transB, in_dtype, out_dtype, epilogue;
  float alpha;
  const float* bias;
  const float* residual;
  int64_t ldr;
  void* aux;
  int64_t ldaux;
  float* colsum;
  void* ws;
};

// One output element: BIAS -> GELU(+aux write) -> DGELU(aux read) -> RESIDUAL -> ACCUM -> store.
// `bias` is the already-fetched bias[n] (0 when the bit is clear).
template <int OUT_DT>
__device__ __forceinline__ void epilogue_store(const GemmParams& p, int64_t bz, int64_t m, int64_t n, float acc,
                                               float bias) {
  float v = acc * p.alpha + bias;
  if (p.epilogue & DINOX_EPI_GELU) {
    if (p.aux) elem<OUT_DT>::st(p.aux, bz * p.M * p.ldaux + m * p.ldaux + n, (p.epilogue & DINOX_EPI_AUXGRAD) ? gelu_erf_grad(v) : v);
    v = gelu_erf(v);
  }
  if (p.epilogue & DINOX_EPI_DGELU) {
    const float a = elem<OUT_DT>::ld(p.aux, bz * p.M * p.ldaux + m * p.ldaux + n);
    v *= (p.epilogue & DINOX_EPI_AUXGRAD) ? a : gelu_erf_grad(a);
  }
  if (p.epilogue & DINOX_EPI_RESIDUAL) v += p.residual[bz * p.M * p.ldr + m * p.ldr + n];
  const int64_t ci = bz * p.strideC + m * p.ldc + n;
  if (p.epilogue & DINOX_EPI_ACCUM) v += ((const float*)p.C)[ci];
  elem<OUT_DT>::st(p.C, ci, v);
}

int launch_gemm_f32(const GemmParams& p, hipStream_t st);
const char* gemm_bf16_variant(const GemmParams& p);  // kernel the bf16 dispatcher would use, or nullptr
int launch_gemm_bf16(const GemmParams& p, hipStream_t st);  // returns DINOX_EUNSUPPORTED when it cannot take the shape
int64_t gemm_bf16_ws_bytes(const GemmParams& p);            // workspace of the deterministic split-K reduction (0: none)

}  // namespace dinox
