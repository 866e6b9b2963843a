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
bool attention_qkv_fused_ok(int B, int N, int heads, int d, int D);
int launch_attention_qkv_fused_fwd(const void*, const void*, const float*, void*, void*, float*, int, int, int, int, int, hipStream_t);
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

// qkv = x W^T + b and softmax(q k^T / sqrt(d)) v in ONE launch (attention_bf16.hip, attn_qkv_fused_fwd): bf16 operands, head size 64,
// 193..224 tokens, D % 32 == 0.  qkv_out / lse may be null (a pass that keeps nothing for a backward).
extern "C" int dinox_qkv_attention_ok(int B, int N, int heads, int d, int D) { return attention_qkv_fused_ok(B, N, heads, d, D) ? 1 : 0; }

extern "C" int dinox_qkv_attention_fwd(const void* x, const void* wqkv, const float* bias, void* o, void* qkv_out, float* lse, int B, int N,
                                       int heads, int d, int D, void* stream) {
  DX_REQUIRE(x && wqkv && o, DINOX_EINVAL, "qkv_attention_fwd: null pointer");
  DX_REQUIRE(B > 0 && N > 0 && heads > 0 && d > 0 && D > 0, DINOX_EINVAL, "qkv_attention_fwd: B=%d N=%d heads=%d d=%d D=%d", B, N, heads, d, D);
  const int rc = launch_attention_qkv_fused_fwd(x, wqkv, bias, o, qkv_out, lse, B, N, heads, d, D, as_stream(stream));
  if (rc == DINOX_EUNSUPPORTED) return fail(rc, "qkv_attention_fwd: B=%d N=%d heads=%d d=%d D=%d outside the fused kernel (head 64, 193..224 tokens, D %% 32 == 0)", B, N, heads, d, D);
  if (rc == DINOX_EALIGN) return fail(rc, "qkv_attention_fwd: operands must be 16-byte aligned");
  return rc;
}
