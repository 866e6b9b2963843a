// attention_bf16.hip -- placeholder until the MFMA flash kernels land.
#include "common.h"
namespace dinox {
int launch_attention_bf16_fwd(const void*, void*, float*, int, int, int, int, hipStream_t) { return DINOX_EUNSUPPORTED; }
int launch_attention_bf16_bwd(const void*, const void*, const void*, const float*, void*, int, int, int, int, hipStream_t) { return DINOX_EUNSUPPORTED; }
}  // namespace dinox
