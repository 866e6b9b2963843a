// attention_bf16.hip -- MFMA flash-style attention core for head_dim 64 on gfx950 (bf16 in, fp32 softmax).
//
// Token counts on this path are small (201 at /16, 261 at /14), so a whole head's K and V (or Q and dO)
// fit in LDS (28 KiB each at N=201) and one wave can hold a full 32-query score strip in registers:
// the forward needs no online-softmax rescaling.  All three kernels use ONE LDS image layout that
// serves both row reads (ds_read_b128, MFMA A operands) and transposed reads (ds_read_b64_tr_b16, MFMA B
// operands whose reduction index runs down the rows) without bank conflicts -- the 8-row x 32-column
// sub-tile image of cdna_hip_programming.md T10(a), adapted to 64-column (128-B) rows.
//
// Score tiles are computed TRANSPOSED (S^T = K . Q^T: keys in the accumulator rows, queries on the lanes)
// so that softmax statistics are lane-local and the accumulator can be re-used directly as the A operand
// of the next product (P^T)^T . V  /  (dS^T)^T . K  with no lane movement (guide section 3, "An accumulator
// tile as the next MFMA's operand"); the other operand is fetched in the matching permuted k order.
//
//   fwd   : one wave per 32 queries.   S^T (all key tiles, registers) -> softmax -> O = P V ; writes lse.
//   bwd dQ: one wave per 32 queries.   streams key tiles: S^T, dP^T = V . dO^T, dS^T -> dQ += dS K.
//   bwd dKV: one wave per 32 keys.     streams query tiles: S = Q . K^T, dP = dO . V^T -> dV += P^T dO, dK += dS^T Q.
// (two backward kernels recompute S/dP once more than a single fused pass would: 7 instead of 5 tile
//  products, traded for no cross-wave reduction of dQ; attention is ~8 % of the block's FLOPs.)
#include "common.h"

namespace dinox {

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
constexpr int AT_D = 64;

// byte offset of 16-B chunk ch (0..7) of row r in a [rows][64 bf16] image
__device__ __forceinline__ int img_off(int r, int ch) {
  return 1024 * (r >> 3) + 512 * (ch >> 2) + 64 * (r & 7) + 16 * ((ch & 3) ^ ((r >> 2) & 3));
}

// Cooperative load of `rows_pad` rows x 64 bf16 (rows >= n_valid are zero) from a strided global matrix.
__device__ __forceinline__ void load_image(char* __restrict__ img, const bf16_t* __restrict__ src, int64_t row_stride,
                                           int n_valid, int rows_pad) {
  for (int idx = threadIdx.x; idx < rows_pad * 8; idx += blockDim.x) {
    const int r = idx >> 3, ch = idx & 7;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r < n_valid) v = *reinterpret_cast<const uint4*>(src + (int64_t)r * row_stride + ch * 8);
    *reinterpret_cast<uint4*>(img + img_off(r, ch)) = v;
  }
}

// A-operand fragment (standard k order): lane (row = l&31, hl = l>>5) gets img[row0 + row][16*ks + 8*hl + j], j<8.
__device__ __forceinline__ bf16x8 frag_rows(const char* __restrict__ img, int row0, int ks, int lane) {
  return *reinterpret_cast<const bf16x8*>(img + img_off(row0 + (lane & 31), 2 * ks + (lane >> 5)));
}

// B-operand fragment in the PERMUTED k order that matches an accumulator tile used as the A operand:
// lane (col = l&31, hl = l>>5), element j  <-  img[row0 + 16*s + 8*(j>>2) + 4*hl + (j&3)][d0 + col].
__device__ __forceinline__ bf16x8 frag_tr_perm(const char* __restrict__ img, int row0, int s, int d0, int lane) {
  const int i = lane & 15, g = lane >> 4;
  const int q4 = i >> 2, p = i & 3, hl = g >> 1;
  const int ch = (d0 >> 3) + 2 * (g & 1) + (p >> 1);
  const int r0 = row0 + 16 * s + 4 * hl + q4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + img_off(r0, ch) + 8 * (p & 1)));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + img_off(r0 + 8, ch) + 8 * (p & 1)));
  s16x8 v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
  v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bf16x8, v);
}

// accumulator registers [8s, 8s+8) -> bf16 A fragment of k-step s
__device__ __forceinline__ bf16x8 acc_as_a(const f32x16& x, int s) {
  bf16x8 a;
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = (__bf16)x[8 * s + j];
  return a;
}

// row index inside a 32-row accumulator tile held by register e of lane-half hl
__device__ __forceinline__ int acc_row(int e, int hl) { return (e & 3) + 8 * (e >> 2) + 4 * hl; }

// 4 fragments (ks = 0..3) of one 64-element global row: lane (row = l&31, hl) takes d = 16*ks + 8*hl + j
__device__ __forceinline__ void load_row_frags(bf16x8 (&f)[4], const bf16_t* __restrict__ rowp, int lane) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) f[ks] = *reinterpret_cast<const bf16x8*>(rowp + 16 * ks + 8 * (lane >> 5));
}

__device__ __forceinline__ void zero16(f32x16& x) {
#pragma unroll
  for (int e = 0; e < 16; ++e) x[e] = 0.f;
}

// store a 32x32 accumulator tile (rows = tokens row0.., cols = d0..) as bf16 rows of a strided matrix
__device__ __forceinline__ void store_tile(bf16_t* __restrict__ dst, int64_t row_stride, int row0, int n_valid, int d0,
                                           const f32x16& x, float scale, int lane) {
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int r = row0 + acc_row(e, lane >> 5);
    if (r < n_valid) dst[(int64_t)r * row_stride + d0 + (lane & 31)] = f32_to_bf16(x[e] * scale);
  }
}

// ------------------------------------------------------------------------------------------ forward
// Two passes over the key tiles: pass 1 keeps only the running row maximum, pass 2 recomputes each S^T tile, exponentiates
// against the exact maximum and feeds P straight into the PV MFMAs.  Holding all NKT score tiles instead (one pass) costs
// 16*NKT accumulator VGPRs (188 total at N=201): one 7-wave workgroup per CU, load phase and MFMA phase never overlap.
// This form needs ~100 VGPRs -> two workgroups per CU overlap each other's K/V staging; the extra QK^T MFMAs are cheap.
template <int NKT>
__global__ __launch_bounds__(512) void attn_fwd_bf16(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o,
                                                     float* __restrict__ lse, int N, int heads, float sc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int b = blockIdx.x / heads, hh = blockIdx.x % heads;
  const int C = heads * AT_D;
  const int64_t rs = 3 * (int64_t)C;                               // token stride of packed qkv
  const bf16_t* base = qkv + (int64_t)b * N * rs + hh * AT_D;
  const int npad = NKT * 32;
  char* kimg = smem;
  char* vimg = smem + npad * 128;
  float* inv_s = reinterpret_cast<float*>(smem + 2 * npad * 128) + wv * 32;   // per-wave 1/rowsum, query-indexed
  load_image(kimg, base + C, rs, N, npad);
  load_image(vimg, base + 2 * C, rs, N, npad);
  __syncthreads();
  const int qb = blockIdx.y * nw + wv;
  if (qb * 32 >= N) return;
  const int q0 = qb * 32, hl = lane >> 5;
  int qrow = q0 + (lane & 31);
  if (qrow >= N) qrow = N - 1;                                      // clamp: computed, never stored
  bf16x8 qf[4];
  load_row_frags(qf, base + (int64_t)qrow * rs, lane);

  float mx = -INFINITY;
#pragma unroll 1
  for (int kt = 0; kt < NKT; ++kt) {
    f32x16 s;
    zero16(s);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(kimg, kt * 32, ks, lane), qf[ks], s, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = kt * 32 + acc_row(e, hl);
      mx = fmaxf(mx, key < N ? s[e] : -INFINITY);
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * sc;                      // sc > 0: max commutes with the scaling

  f32x16 oacc[2];
  zero16(oacc[0]);
  zero16(oacc[1]);
  float sum = 0.f;
#pragma unroll 1
  for (int kt = 0; kt < NKT; ++kt) {
    f32x16 s;
    zero16(s);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(kimg, kt * 32, ks, lane), qf[ks], s, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = kt * 32 + acc_row(e, hl);
      const float p = key < N ? __expf(s[e] * sc - mx) : 0.f;
      s[e] = p;
      sum += p;
    }
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      const bf16x8 pa = acc_as_a(s, ss);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
        oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, frag_tr_perm(vimg, kt * 32, ss, dt * 32, lane), oacc[dt], 0, 0, 0);
    }
  }
  sum += __shfl_xor(sum, 32, 64);
  if (hl == 0) {
    inv_s[lane] = 1.0f / sum;                                       // lane = query within the block
    if (q0 + lane < N) lse[((int64_t)b * heads + hh) * N + q0 + lane] = mx + __logf(sum);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // O rows are queries in the accumulator layout: fetch each row's 1/sum (4 consecutive rows per register group)
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 iv = *reinterpret_cast<const float4*>(inv_s + 8 * g + 4 * hl);
    const float ivv[4] = {iv.x, iv.y, iv.z, iv.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      oacc[0][4 * g + r] *= ivv[r];
      oacc[1][4 * g + r] *= ivv[r];
    }
  }
  bf16_t* ob = o + (int64_t)b * N * C + hh * AT_D;
  store_tile(ob, C, q0, N, 0, oacc[0], 1.0f, lane);
  store_tile(ob, C, q0, N, 32, oacc[1], 1.0f, lane);
}

// ------------------------------------------------------------------------------------------ backward: dQ
__global__ __launch_bounds__(512) void attn_bwd_dq_bf16(const bf16_t* __restrict__ d_o, const bf16_t* __restrict__ qkv,
                                                        const bf16_t* __restrict__ o, const float* __restrict__ lse,
                                                        bf16_t* __restrict__ dqkv, int N, int heads, int nkt, float sc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int b = blockIdx.x / heads, hh = blockIdx.x % heads;
  const int C = heads * AT_D;
  const int64_t rs = 3 * (int64_t)C;
  const bf16_t* base = qkv + (int64_t)b * N * rs + hh * AT_D;
  const int npad = nkt * 32;
  char* kimg = smem;
  char* vimg = smem + npad * 128;
  load_image(kimg, base + C, rs, N, npad);
  load_image(vimg, base + 2 * C, rs, N, npad);
  __syncthreads();
  const int qb = blockIdx.y * nw + wv;
  if (qb * 32 >= N) return;
  const int q0 = qb * 32, hl = lane >> 5;
  int qrow = q0 + (lane & 31);
  if (qrow >= N) qrow = N - 1;
  const int64_t orow = ((int64_t)b * N + qrow) * C + hh * AT_D;
  bf16x8 qf[4], dof[4], of[4];
  load_row_frags(qf, base + (int64_t)qrow * rs, lane);
  load_row_frags(dof, d_o + orow, lane);
  load_row_frags(of, o + orow, lane);
  float delta = 0.f;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int j = 0; j < 8; ++j) delta += (float)dof[ks][j] * (float)of[ks][j];
  delta += __shfl_xor(delta, 32, 64);
  const float L = lse[((int64_t)b * heads + hh) * N + qrow];

  f32x16 dq[2];
  zero16(dq[0]);
  zero16(dq[1]);
  for (int kt = 0; kt < nkt; ++kt) {
    f32x16 st, dp;
    zero16(st);
    zero16(dp);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(kimg, kt * 32, ks, lane), qf[ks], st, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(vimg, kt * 32, ks, lane), dof[ks], dp, 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = kt * 32 + acc_row(e, hl);
      const float p = __expf(st[e] * sc - L);
      st[e] = key < N ? p * (dp[e] - delta) * sc : 0.f;             // dS^T (scaled), padded keys contribute nothing
    }
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      const bf16x8 a = acc_as_a(st, ss);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
        dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, frag_tr_perm(kimg, kt * 32, ss, dt * 32, lane), dq[dt], 0, 0, 0);
    }
  }
  bf16_t* dqb = dqkv + (int64_t)b * N * rs + hh * AT_D;
  store_tile(dqb, rs, q0, N, 0, dq[0], 1.0f, lane);
  store_tile(dqb, rs, q0, N, 32, dq[1], 1.0f, lane);
}

// ------------------------------------------------------------------------------------------ backward: dK, dV
__global__ __launch_bounds__(512) void attn_bwd_dkv_bf16(const bf16_t* __restrict__ d_o, const bf16_t* __restrict__ qkv,
                                                         const bf16_t* __restrict__ o, const float* __restrict__ lse,
                                                         bf16_t* __restrict__ dqkv, int N, int heads, int nqt, float sc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int b = blockIdx.x / heads, hh = blockIdx.x % heads;
  const int C = heads * AT_D;
  const int64_t rs = 3 * (int64_t)C;
  const bf16_t* base = qkv + (int64_t)b * N * rs + hh * AT_D;
  const bf16_t* dob = d_o + (int64_t)b * N * C + hh * AT_D;
  const bf16_t* ob = o + (int64_t)b * N * C + hh * AT_D;
  const int npad = nqt * 32;
  char* qimg = smem;
  char* doimg = smem + npad * 128;
  float* lse_s = reinterpret_cast<float*>(smem + 2 * npad * 128);
  float* del_s = lse_s + npad;
  load_image(qimg, base, rs, N, npad);
  // dO image + delta[q] = sum_d dO*O in the same pass: 8 consecutive threads own the 8 chunks of one row
  for (int idx = threadIdx.x; idx < npad * 8; idx += blockDim.x) {
    const int r = idx >> 3, ch = idx & 7;
    uint4 v = make_uint4(0, 0, 0, 0);
    float part = 0.f;
    if (r < N) {
      v = *reinterpret_cast<const uint4*>(dob + (int64_t)r * C + ch * 8);
      const uint4 w = *reinterpret_cast<const uint4*>(ob + (int64_t)r * C + ch * 8);
      const bf16_t* a = reinterpret_cast<const bf16_t*>(&v);
      const bf16_t* c = reinterpret_cast<const bf16_t*>(&w);
#pragma unroll
      for (int j = 0; j < 8; ++j) part += bf16_to_f32(a[j]) * bf16_to_f32(c[j]);
    }
    *reinterpret_cast<uint4*>(doimg + img_off(r, ch)) = v;
    part += __shfl_xor(part, 1, 64);
    part += __shfl_xor(part, 2, 64);
    part += __shfl_xor(part, 4, 64);
    if (ch == 0) {
      del_s[r] = part;
      lse_s[r] = r < N ? lse[((int64_t)b * heads + hh) * N + r] : INFINITY;   // exp(x - inf) = 0 for padded queries
    }
  }
  __syncthreads();
  const int kb = blockIdx.y * nw + wv;
  if (kb * 32 >= N) return;
  const int k0 = kb * 32, hl = lane >> 5;
  int krow = k0 + (lane & 31);
  if (krow >= N) krow = N - 1;
  bf16x8 kf[4], vf[4];
  load_row_frags(kf, base + (int64_t)krow * rs + C, lane);
  load_row_frags(vf, base + (int64_t)krow * rs + 2 * C, lane);

  f32x16 dk[2], dv[2];
  zero16(dk[0]); zero16(dk[1]); zero16(dv[0]); zero16(dv[1]);
  for (int qt = 0; qt < nqt; ++qt) {
    f32x16 st, dp;
    zero16(st);
    zero16(dp);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(qimg, qt * 32, ks, lane), kf[ks], st, 0, 0, 0);   // S[q][key]
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(doimg, qt * 32, ks, lane), vf[ks], dp, 0, 0, 0);  // dP[q][key]
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int q = qt * 32 + acc_row(e, hl);
      const float p = __expf(st[e] * sc - lse_s[q]);
      st[e] = p;                                                   // P
      dp[e] = p * (dp[e] - del_s[q]) * sc;                         // dS (scaled)
    }
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      const bf16x8 pa = acc_as_a(st, ss), da = acc_as_a(dp, ss);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, frag_tr_perm(doimg, qt * 32, ss, dt * 32, lane), dv[dt], 0, 0, 0);
        dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, frag_tr_perm(qimg, qt * 32, ss, dt * 32, lane), dk[dt], 0, 0, 0);
      }
    }
  }
  bf16_t* dkb = dqkv + (int64_t)b * N * rs + hh * AT_D + C;
  bf16_t* dvb = dkb + C;
  store_tile(dkb, rs, k0, N, 0, dk[0], 1.0f, lane);
  store_tile(dkb, rs, k0, N, 32, dk[1], 1.0f, lane);
  store_tile(dvb, rs, k0, N, 0, dv[0], 1.0f, lane);
  store_tile(dvb, rs, k0, N, 32, dv[1], 1.0f, lane);
}

// ------------------------------------------------------------------------------------------ launchers
static void geometry(int N, int& nblk, int& nwg, int& waves) {
  nblk = (N + 31) / 32;
  nwg = (nblk + 7) / 8;
  waves = (nblk + nwg - 1) / nwg;
}

template <typename K>
static int allow_lds(K kern, size_t bytes) {
  if (bytes <= 64 * 1024) return 0;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  return e == hipSuccess ? 0 : (int)e;
}

int launch_attention_bf16_fwd(const void* qkv, void* o, float* lse, int B, int N, int heads, int d, hipStream_t st) {
  if (d != AT_D || N > 288 || ((uintptr_t)qkv & 15) || ((uintptr_t)o & 15)) return DINOX_EUNSUPPORTED;
  int nblk, nwg, waves;
  geometry(N, nblk, nwg, waves);
  const float sc = 1.0f / sqrtf((float)d);
  dim3 grid((unsigned)(B * heads), (unsigned)nwg), block((unsigned)(waves * 64));
#define FWD(NKT)                                                                                                   \
  do {                                                                                                             \
    const size_t lds = (size_t)2 * NKT * 32 * 128 + 8 * 32 * sizeof(float);                                        \
    if (int rc = allow_lds(attn_fwd_bf16<NKT>, lds)) return fail(rc, "attention_fwd: cannot reserve %zu B of LDS", lds); \
    hipLaunchKernelGGL((attn_fwd_bf16<NKT>), grid, block, lds, st, (const bf16_t*)qkv, (bf16_t*)o, lse, N, heads, sc); \
  } while (0)
  if (nblk <= 2) FWD(2); else if (nblk <= 4) FWD(4); else if (nblk <= 7) FWD(7); else FWD(9);
#undef FWD
  return check_launch("attention_bf16_fwd");
}

int launch_attention_bf16_bwd(const void* d_o, const void* qkv, const void* o, const float* lse, void* dqkv, int B, int N,
                              int heads, int d, hipStream_t st) {
  if (d != AT_D || N > 544 || ((uintptr_t)qkv & 15) || ((uintptr_t)o & 15) || ((uintptr_t)d_o & 15)) return DINOX_EUNSUPPORTED;
  int nblk, nwg, waves;
  geometry(N, nblk, nwg, waves);
  const float sc = 1.0f / sqrtf((float)d);
  dim3 grid((unsigned)(B * heads), (unsigned)nwg), block((unsigned)(waves * 64));
  const size_t lds1 = (size_t)2 * nblk * 32 * 128;
  const size_t lds2 = lds1 + (size_t)2 * nblk * 32 * sizeof(float);
  if (int rc = allow_lds(attn_bwd_dq_bf16, lds1)) return fail(rc, "attention_bwd: cannot reserve %zu B of LDS", lds1);
  if (int rc = allow_lds(attn_bwd_dkv_bf16, lds2)) return fail(rc, "attention_bwd: cannot reserve %zu B of LDS", lds2);
  hipLaunchKernelGGL(attn_bwd_dq_bf16, grid, block, lds1, st, (const bf16_t*)d_o, (const bf16_t*)qkv, (const bf16_t*)o, lse,
                     (bf16_t*)dqkv, N, heads, nblk, sc);
  hipLaunchKernelGGL(attn_bwd_dkv_bf16, grid, block, lds2, st, (const bf16_t*)d_o, (const bf16_t*)qkv, (const bf16_t*)o, lse,
                     (bf16_t*)dqkv, N, heads, nblk, sc);
  return check_launch("attention_bf16_bwd");
}

}  // namespace dinox
