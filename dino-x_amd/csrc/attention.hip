// attention.hip -- dinox_attention_{fwd,bwd} dispatch: MFMA flash kernels for bf16 when the shape is
// inside their envelope, otherwise the fp32-math reference kernels (attention_ref.hip).
#include <cstdlib>

#include "common.h"

namespace dinox {
int launch_attention_ref_fwd(const void*, void*, float*, int, int, int, int, int, hipStream_t);
int launch_attention_ref_bwd(const void*, const void*, const void*, const float*, void*, int, int, int, int, int, hipStream_t);
int launch_attention_bf16_fwd(const void*, void*, float*, int, int, int, int, hipStream_t);   // EUNSUPPORTED if outside envelope
int launch_attention_bf16_bwd(const void*, const void*, const void*, const float*, void*, float*, int, int, int, int, hipStream_t);
int launch_attention_flash_fwd(const void*, void*, float*, int, int, int, int, hipStream_t);  // attention_flash.hip: any N, d <= 128 (d % 8 == 0)
int launch_attention_flash_bwd(const void*, const void*, const void*, const float*, void*, float*, int, int, int, int, hipStream_t);
}  // namespace dinox

using namespace dinox;

static int check_attn(const char* who, int B, int N, int heads, int d, int dtype) {
  DX_REQUIRE(B > 0 && N > 0 && heads > 0 && d > 0 && d <= 128, DINOX_EINVAL, "%s: B=%d N=%d heads=%d d=%d", who, B, N, heads, d);
  DX_REQUIRE(dtype == DINOX_F32 || dtype == DINOX_BF16, DINOX_EINVAL, "%s: dtype %d", who, dtype);
  return 0;
}

extern "C" int dinox_attention_fwd(const void* qkv, void* o, float* lse, int B, int N, int heads, int d, int dtype,
                                   void* stream) {
  DX_REQUIRE(qkv && o && lse, DINOX_EINVAL, "attention_fwd: null pointer");
  if (int rc = check_attn("attention_fwd", B, N, heads, d, dtype)) return rc;
  hipStream_t st = as_stream(stream);
  if (dtype == DINOX_BF16) {
    int rc = launch_attention_bf16_fwd(qkv, o, lse, B, N, heads, d, st);              // head size 64, whole score strips in registers
    if (rc == DINOX_EUNSUPPORTED && !getenv("DINOX_ATTN_NO_FLASH"))                    // (A/B and tests: the per-lane reference kernels instead)
      rc = launch_attention_flash_fwd(qkv, o, lse, B, N, heads, d, st);                // the tiled form: long sequences, other head sizes
    if (rc != DINOX_EUNSUPPORTED) return rc;
  }
  return launch_attention_ref_fwd(qkv, o, lse, B, N, heads, d, dtype, st);
}

extern "C" int64_t dinox_attention_bwd_ws_bytes(int B, int N, int heads) {
  if (B <= 0 || N <= 0 || heads <= 0) return 0;
  return (int64_t)B * heads * N * (int64_t)sizeof(float);          // delta[q] = rowsum(dO * O), shared by the two backward kernels
}

extern "C" int dinox_attention_bwd(const void* d_o, const void* qkv, const void* o, const float* lse, void* dqkv, void* ws,
                                   int B, int N, int heads, int d, int dtype, void* stream) {
  DX_REQUIRE(d_o && qkv && o && lse && dqkv, DINOX_EINVAL, "attention_bwd: null pointer");
  if (int rc = check_attn("attention_bwd", B, N, heads, d, dtype)) return rc;
  hipStream_t st = as_stream(stream);
  if (dtype == DINOX_BF16) {
    int rc = launch_attention_bf16_bwd(d_o, qkv, o, lse, dqkv, (float*)ws, B, N, heads, d, st);
    if (rc == DINOX_EUNSUPPORTED && !getenv("DINOX_ATTN_NO_FLASH")) rc = launch_attention_flash_bwd(d_o, qkv, o, lse, dqkv, (float*)ws, B, N, heads, d, st);
    if (rc != DINOX_EUNSUPPORTED) return rc;
  }
  return launch_attention_ref_bwd(d_o, qkv, o, lse, dqkv, B, N, heads, d, dtype, st);
}
