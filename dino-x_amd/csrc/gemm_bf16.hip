// gemm_bf16.hip -- placeholder until the MFMA bf16 kernels land (next commit).
#include "common.h"
#include "gemm_common.h"
namespace dinox {
int launch_gemm_bf16(const GemmParams&, hipStream_t) { return DINOX_EUNSUPPORTED; }
}  // namespace dinox
