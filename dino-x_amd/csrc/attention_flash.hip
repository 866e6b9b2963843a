// attention_flash.hip -- tiled bf16 MFMA attention for the shapes the whole-strip kernels of attention_bf16.hip do not take: any
// sequence length (448 / 512 px inputs: 789 / 1029 tokens and beyond) and head sizes up to 128 (the reference's vit-giant preset:
// 1408 / 16 = 88 per head, /root/reference scripts/phase5_big_run.py:212-220), on the same packed qkv [B, N, 3, heads, d] layout.
// Before round 3 such shapes ran the exact-fp32 product form (1/16-rate matrix instructions, [B, N, N] fp32 score buffers in HBM).
//
// Same conventions as attention_bf16.hip (read its header first): v_mfma_f32_32x32x16_bf16; score tiles computed TRANSPOSED
// (S^T = K . Q^T: keys in the accumulator rows, a query per lane) so that softmax statistics are lane-local and the accumulator is
// the next product's A operand as it stands; ONE LDS image layout for row reads (ds_read_b128) and transposed reads
// (ds_read_b64_tr_b16).  What is new:
//   * head size DH = 64 / 96 / 128 columns per image row (88 runs as 96: the missing columns are zero in Q, K, V, dO, so they add
//     nothing to a product and their outputs are not stored); d % 8 == 0;
//   * K / V (forward, dQ) and Q / dO (dK, dV) move through LDS in CHUNKS of 64 rows, so the sequence length is unbounded;
//   * the forward keeps the running row maximum and sum of online softmax (per 32-key tile; the output block is rescaled only when
//     some row's maximum really moved -- after the first few tiles it rarely does).
// Replaces F.scaled_dot_product_attention (reference zoo/arch.py:51) and its backward.
#include <cstdlib>

#include "common.h"

namespace dinox {

typedef __attribute__((address_space(3))) s16x4 fl_lds_s16x4;
constexpr int FL_KC = 64;                                     // rows of a chunk

// byte offset of 16-B chunk ch (0 .. DH/8 - 1) of row r in a [rows][DH bf16] image: 8-row x 32-column sub-tiles of 512 B
template <int DH>
__device__ __forceinline__ int fl_off(int r, int ch) {
  return (DH * 16) * (r >> 3) + 512 * (ch >> 2) + 64 * (r & 7) + 16 * ((ch & 3) ^ ((r >> 2) & 3));
}

// rows [row0, row0 + FL_KC) of a strided global matrix -> image rows 0 .. FL_KC - 1 (rows >= n_valid and columns >= d are zero)
template <int DH>
__device__ __forceinline__ void fl_load_chunk(char* __restrict__ img, const bf16_t* __restrict__ src, int64_t row_stride, int row0, int n_valid, int d) {
  constexpr int CH = DH / 8;
  for (int idx = threadIdx.x; idx < FL_KC * CH; idx += blockDim.x) {
    const int r = idx / CH, ch = idx - r * CH;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row0 + r < n_valid && ch * 8 < d) v = *reinterpret_cast<const uint4*>(src + (int64_t)(row0 + r) * row_stride + ch * 8);
    *reinterpret_cast<uint4*>(img + fl_off<DH>(r, ch)) = v;
  }
}

// A-operand fragment (standard k order): lane (row = l & 31, hl = l >> 5) gets img[row0 + row][16 ks + 8 hl + j], j < 8
template <int DH>
__device__ __forceinline__ bf16x8 fl_rows(const char* __restrict__ img, int row0, int ks, int lane) {
  return *reinterpret_cast<const bf16x8*>(img + fl_off<DH>(row0 + (lane & 31), 2 * ks + (lane >> 5)));
}

// B-operand fragment in the PERMUTED k order of an accumulator tile used as the A operand:
// lane (col = l & 31, hl = l >> 5), element j  <-  img[row0 + 16 s + 8 (j >> 2) + 4 hl + (j & 3)][d0 + col]
template <int DH>
__device__ __forceinline__ bf16x8 fl_tr(const char* __restrict__ img, int row0, int s, int d0, int lane) {
  const int i = lane & 15, g = lane >> 4;
  const int q4 = i >> 2, p = i & 3, hl = g >> 1;
  const int ch = (d0 >> 3) + 2 * (g & 1) + (p >> 1);
  const int r0 = row0 + 16 * s + 4 * hl + q4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((fl_lds_s16x4*)(img + fl_off<DH>(r0, ch) + 8 * (p & 1)));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((fl_lds_s16x4*)(img + fl_off<DH>(r0 + 8, ch) + 8 * (p & 1)));
  s16x8 v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
  v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ bf16x8 fl_acc_as_a(const f32x16& x, int s) {
  bf16x8 a;
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = (__bf16)x[8 * s + j];
  return a;
}
__device__ __forceinline__ int fl_acc_row(int e, int hl) { return (e & 3) + 8 * (e >> 2) + 4 * hl; }
__device__ __forceinline__ void fl_zero(f32x16& x) {
#pragma unroll
  for (int e = 0; e < 16; ++e) x[e] = 0.f;
}

// DH / 16 fragments of one global row of d valid elements: lane (hl) takes columns 16 ks + 8 hl .. + 7 (zero past d)
template <int DH>
__device__ __forceinline__ void fl_row_frags(bf16x8 (&f)[DH / 16], const bf16_t* __restrict__ rowp, int d, int lane) {
#pragma unroll
  for (int ks = 0; ks < DH / 16; ++ks) {
    const int c = 16 * ks + 8 * (lane >> 5);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (c < d) v = *reinterpret_cast<const uint4*>(rowp + c);
    f[ks] = __builtin_bit_cast(bf16x8, v);
  }
}

// A 32-row x DH-column block of fp32 accumulators (DH / 32 tiles of 32 x 32: rows = tokens, a column per lane) -> bf16 rows of a
// strided matrix, through a per-wave LDS scratch of 32 rows x (2 DH + 16) bytes: re-read by rows, a lane stores 16-byte pieces and
// a row leaves as whole segments.  Columns >= d are dropped.
template <int DH>
__device__ __forceinline__ void fl_store_block(bf16_t* __restrict__ dst, int64_t row_stride, int row0, int n_valid, int d, const f32x16 (&x)[DH / 32],
                                               char* scratch, int lane) {
  constexpr int ROWB = 2 * DH + 16, CH = DH / 8;
  const int col = lane & 31, hl = lane >> 5;
#pragma unroll
  for (int t = 0; t < DH / 32; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) *reinterpret_cast<bf16_t*>(scratch + fl_acc_row(e, hl) * ROWB + (t * 32 + col) * 2) = f32_to_bf16(x[t][e]);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int idx = lane; idx < 32 * CH; idx += 64) {
    const int r = idx / CH, c = idx - r * CH;
    const uint4 v = *reinterpret_cast<const uint4*>(scratch + r * ROWB + c * 16);
    if (row0 + r < n_valid && c * 8 < d) *reinterpret_cast<uint4*>(dst + (int64_t)(row0 + r) * row_stride + c * 8) = v;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------------------------ forward
// One wave per 32 queries, four waves per workgroup; K and V chunks of 64 keys shared by the workgroup.  Statistics live in the log2
// domain (m2 = max * sc * log2 e); lse is written in natural units like the other attention kernels.
template <int DH>
__global__ __launch_bounds__(256, 2) void attn_flash_fwd(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o, float* __restrict__ lse, int N,
                                                      int heads, int d, float sc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KS = DH / 16, DT = DH / 32, IMG = FL_KC * DH * 2, SCR = 32 * (2 * DH + 16);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, hl = lane >> 5;
  const int b = blockIdx.x / heads, hh = blockIdx.x % heads;
  const int C = heads * d;
  const int64_t rs = 3 * (int64_t)C;
  const bf16_t* base = qkv + (int64_t)b * N * rs + hh * d;
  char* kimg = smem;
  char* vimg = smem + IMG;
  char* scratch = smem + 2 * IMG + wv * SCR;
  float* rowf = reinterpret_cast<float*>(smem + 2 * IMG + 4 * SCR) + wv * 32;      // per-wave per-query factors (rescale, 1 / rowsum)
  const int q0 = (blockIdx.y * 4 + wv) * 32;
  const bool active = q0 < N;                                        // wave-uniform; inactive waves still help loading
  int qrow = q0 + (lane & 31);
  if (qrow >= N) qrow = N - 1;
  bf16x8 qf[KS];
  fl_row_frags<DH>(qf, base + (int64_t)qrow * rs, d, lane);
  const float c2 = sc * 1.4426950408889634f;
  float m2 = -INFINITY, l = 0.f;
  f32x16 oacc[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t) fl_zero(oacc[t]);

  for (int kc0 = 0; kc0 < N; kc0 += FL_KC) {
    __syncthreads();                                                 // every wave is done with the previous chunk
    fl_load_chunk<DH>(kimg, base + C, rs, kc0, N, d);
    fl_load_chunk<DH>(vimg, base + 2 * C, rs, kc0, N, d);
    __syncthreads();
    if (!active) continue;
#pragma unroll
    for (int kt = 0; kt < FL_KC / 32; ++kt) {
      const int key0 = kc0 + kt * 32;
      if (key0 >= N) break;                                          // (uniform)
      f32x16 st;
      fl_zero(st);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl_rows<DH>(kimg, kt * 32, ks, lane), qf[ks], st, 0, 0, 0);
      float tmax = -INFINITY;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const bool ok = key0 + fl_acc_row(e, hl) < N;
        st[e] = ok ? st[e] * c2 : -INFINITY;
        tmax = fmaxf(tmax, st[e]);
      }
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      const float mn = fmaxf(m2, tmax);                              // finite: the tile holds at least one valid key
      const float alpha = __builtin_amdgcn_exp2f(m2 - mn);           // 0 on the first tile (m2 = -inf)
      const bool moved = mn > m2;
      m2 = mn;
      l *= alpha;
      if (__any(moved)) {                                            // rescale the output block: its rows are queries, the factors are per lane
        if (hl == 0) rowf[lane] = alpha;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const float4 a4 = *reinterpret_cast<const float4*>(rowf + 8 * g4 + 4 * hl);
          const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < DT; ++t) oacc[t][4 * g4 + r] *= av[r];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float pv = __builtin_amdgcn_exp2f(st[e] - mn);         // masked keys: exp2(-inf) = 0
        st[e] = pv;
        l += pv;
      }
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        const bf16x8 pa = fl_acc_as_a(st, ss);
#pragma unroll
        for (int t = 0; t < DT; ++t) oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, fl_tr<DH>(vimg, kt * 32, ss, t * 32, lane), oacc[t], 0, 0, 0);
      }
    }
  }
  if (!active) return;
  l += __shfl_xor(l, 32, 64);
  if (hl == 0) {
    rowf[lane] = 1.0f / l;
    if (q0 + lane < N) lse[((int64_t)b * heads + hh) * N + q0 + lane] = m2 * 0.6931471805599453f + __logf(l);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    const float4 a4 = *reinterpret_cast<const float4*>(rowf + 8 * g4 + 4 * hl);
    const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int t = 0; t < DT; ++t) oacc[t][4 * g4 + r] *= av[r];
  }
  fl_store_block<DH>(o + (int64_t)b * N * C + hh * d, C, q0, N, d, oacc, scratch, lane);
}

// ------------------------------------------------------------------------------------------ backward: dQ (and delta)
// One wave per 32 queries; K and V chunks.  delta = rowsum(dO o O) goes to the workspace for the dK / dV kernel.
template <int DH>
__global__ __launch_bounds__(256, 2) void attn_flash_bwd_dq(const bf16_t* __restrict__ d_o, const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                         const float* __restrict__ lse, bf16_t* __restrict__ dqkv, float* __restrict__ delta_ws, int N,
                                                         int heads, int d, float sc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KS = DH / 16, DT = DH / 32, IMG = FL_KC * DH * 2, SCR = 32 * (2 * DH + 16);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, hl = lane >> 5;
  const int b = blockIdx.x / heads, hh = blockIdx.x % heads;
  const int C = heads * d;
  const int64_t rs = 3 * (int64_t)C;
  const bf16_t* base = qkv + (int64_t)b * N * rs + hh * d;
  char* kimg = smem;
  char* vimg = smem + IMG;
  char* scratch = smem + 2 * IMG + wv * SCR;
  const int q0 = (blockIdx.y * 4 + wv) * 32;
  const bool active = q0 < N;
  int qrow = q0 + (lane & 31);
  if (qrow >= N) qrow = N - 1;
  const int64_t orow = ((int64_t)b * N + qrow) * C + hh * d;
  bf16x8 qf[KS], dof[KS];
  fl_row_frags<DH>(qf, base + (int64_t)qrow * rs, d, lane);
  fl_row_frags<DH>(dof, d_o + orow, d, lane);
  float delta = 0.f;
  {
    bf16x8 of[KS];
    fl_row_frags<DH>(of, o + orow, d, lane);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) delta += (float)dof[ks][j] * (float)of[ks][j];
  }
  delta += __shfl_xor(delta, 32, 64);
  if (active && hl == 0 && q0 + lane < N) delta_ws[((int64_t)b * heads + hh) * N + q0 + lane] = delta;
  const float L2 = lse[((int64_t)b * heads + hh) * N + qrow] * 1.4426950408889634f, c2 = sc * 1.4426950408889634f;
  f32x16 dq[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t) fl_zero(dq[t]);
  for (int kc0 = 0; kc0 < N; kc0 += FL_KC) {
    __syncthreads();
    fl_load_chunk<DH>(kimg, base + C, rs, kc0, N, d);
    fl_load_chunk<DH>(vimg, base + 2 * C, rs, kc0, N, d);
    __syncthreads();
    if (!active) continue;
#pragma unroll
    for (int kt = 0; kt < FL_KC / 32; ++kt) {
      const int key0 = kc0 + kt * 32;
      if (key0 >= N) break;
      f32x16 st, dp;
      fl_zero(st);
      fl_zero(dp);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl_rows<DH>(kimg, kt * 32, ks, lane), qf[ks], st, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl_rows<DH>(vimg, kt * 32, ks, lane), dof[ks], dp, 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float pv = __builtin_amdgcn_exp2f(st[e] * c2 - L2);
        st[e] = key0 + fl_acc_row(e, hl) < N ? pv * (dp[e] - delta) * sc : 0.f;           // dS^T (scaled)
      }
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        const bf16x8 a = fl_acc_as_a(st, ss);
#pragma unroll
        for (int t = 0; t < DT; ++t) dq[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, fl_tr<DH>(kimg, kt * 32, ss, t * 32, lane), dq[t], 0, 0, 0);
      }
    }
  }
  if (!active) return;
  fl_store_block<DH>(dqkv + (int64_t)b * N * rs + hh * d, rs, q0, N, d, dq, scratch, lane);
}

// ------------------------------------------------------------------------------------------ backward: dK, dV
// One wave per 32 keys; Q and dO chunks (with their lse / delta) shared by the workgroup.
template <int DH>
__global__ __launch_bounds__(256, 2) void attn_flash_bwd_dkv(const bf16_t* __restrict__ d_o, const bf16_t* __restrict__ qkv, const float* __restrict__ lse,
                                                          const float* __restrict__ delta_ws, bf16_t* __restrict__ dqkv, int N, int heads, int d, float sc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KS = DH / 16, DT = DH / 32, IMG = FL_KC * DH * 2, SCR = 32 * (2 * DH + 16);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, hl = lane >> 5;
  const int b = blockIdx.x / heads, hh = blockIdx.x % heads;
  const int C = heads * d;
  const int64_t rs = 3 * (int64_t)C;
  const bf16_t* base = qkv + (int64_t)b * N * rs + hh * d;
  const bf16_t* dob = d_o + (int64_t)b * N * C + hh * d;
  char* qimg = smem;
  char* doimg = smem + IMG;
  char* scratch = smem + 2 * IMG + wv * SCR;
  float* lse2_s = reinterpret_cast<float*>(smem + 2 * IMG + 4 * SCR);
  float* del_s = lse2_s + FL_KC;
  const int k0 = (blockIdx.y * 4 + wv) * 32;
  const bool active = k0 < N;
  int krow = k0 + (lane & 31);
  if (krow >= N) krow = N - 1;
  bf16x8 kf[KS], vf[KS];
  fl_row_frags<DH>(kf, base + (int64_t)krow * rs + C, d, lane);
  fl_row_frags<DH>(vf, base + (int64_t)krow * rs + 2 * C, d, lane);
  const float c2 = sc * 1.4426950408889634f;
  f32x16 dk[DT], dv[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t) {
    fl_zero(dk[t]);
    fl_zero(dv[t]);
  }
  for (int qc0 = 0; qc0 < N; qc0 += FL_KC) {
    __syncthreads();
    fl_load_chunk<DH>(qimg, base, rs, qc0, N, d);
    fl_load_chunk<DH>(doimg, dob, C, qc0, N, d);
    if (threadIdx.x < FL_KC) {
      const int q = qc0 + threadIdx.x;
      const int64_t si = ((int64_t)b * heads + hh) * N + q;
      lse2_s[threadIdx.x] = q < N ? lse[si] * 1.4426950408889634f : INFINITY;      // exp2(x - inf) = 0 for padded queries
      del_s[threadIdx.x] = q < N ? delta_ws[si] : 0.f;
    }
    __syncthreads();
    if (!active) continue;
#pragma unroll
    for (int qt = 0; qt < FL_KC / 32; ++qt) {
      if (qc0 + qt * 32 >= N) break;
      f32x16 st, dp;
      fl_zero(st);
      fl_zero(dp);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl_rows<DH>(qimg, qt * 32, ks, lane), kf[ks], st, 0, 0, 0);      // S[q][key]
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl_rows<DH>(doimg, qt * 32, ks, lane), vf[ks], dp, 0, 0, 0);     // dP[q][key]
      }
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int q = qt * 32 + 8 * g4 + 4 * hl;
        const float4 l4 = *reinterpret_cast<const float4*>(lse2_s + q);
        const float4 d4 = *reinterpret_cast<const float4*>(del_s + q);
        const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dvv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int e = 4 * g4 + r;
          const float pv = __builtin_amdgcn_exp2f(st[e] * c2 - lv[r]);
          st[e] = pv;                                                  // P
          dp[e] = pv * (dp[e] - dvv[r]) * sc;                          // dS (scaled)
        }
      }
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        const bf16x8 pa = fl_acc_as_a(st, ss), da = fl_acc_as_a(dp, ss);
#pragma unroll
        for (int t = 0; t < DT; ++t) {
          dv[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, fl_tr<DH>(doimg, qt * 32, ss, t * 32, lane), dv[t], 0, 0, 0);
          dk[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, fl_tr<DH>(qimg, qt * 32, ss, t * 32, lane), dk[t], 0, 0, 0);
        }
      }
    }
  }
  if (!active) return;
  bf16_t* dkb = dqkv + (int64_t)b * N * rs + hh * d + C;
  fl_store_block<DH>(dkb, rs, k0, N, d, dk, scratch, lane);
  fl_store_block<DH>(dkb + C, rs, k0, N, d, dv, scratch, lane);
}

// ------------------------------------------------------------------------------------------ launchers
static int fl_dh(int d) { return d <= 64 ? 64 : d <= 96 ? 96 : 128; }
static bool fl_ok(const void* a, const void* b, int d) { return d >= 8 && d <= 128 && (d & 7) == 0 && !((uintptr_t)a & 15) && !((uintptr_t)b & 15); }
int reserve_lds(const void* kern, size_t bytes, const char* what);

int launch_attention_flash_fwd(const void* qkv, void* o, float* lse, int B, int N, int heads, int d, hipStream_t st) {
  if (!fl_ok(qkv, o, d)) return DINOX_EUNSUPPORTED;
  const float sc = 1.0f / sqrtf((float)d);
  const dim3 grid((unsigned)(B * heads), (unsigned)((N + 127) / 128)), block(256);
#define FL_F(DH)                                                                                                                      \
  do {                                                                                                                                \
    const size_t lds = (size_t)2 * FL_KC * DH * 2 + 4 * 32 * (2 * DH + 16) + 4 * 32 * sizeof(float);                                 \
    if (int rc = reserve_lds(reinterpret_cast<const void*>(attn_flash_fwd<DH>), lds, "attention_flash_fwd")) return rc;              \
    hipLaunchKernelGGL((attn_flash_fwd<DH>), grid, block, lds, st, (const bf16_t*)qkv, (bf16_t*)o, lse, N, heads, d, sc);            \
  } while (0)
  const int dh = fl_dh(d);
  if (dh == 64) FL_F(64); else if (dh == 96) FL_F(96); else FL_F(128);
#undef FL_F
  return check_launch("attention_flash_fwd");
}

int launch_attention_flash_bwd(const void* d_o, const void* qkv, const void* o, const float* lse, void* dqkv, float* ws, int B, int N, int heads,
                               int d, hipStream_t st) {
  if (!ws || !fl_ok(qkv, o, d) || !fl_ok(d_o, dqkv, d)) return DINOX_EUNSUPPORTED;
  const float sc = 1.0f / sqrtf((float)d);
  const dim3 grid((unsigned)(B * heads), (unsigned)((N + 127) / 128)), block(256);
#define FL_B(DH)                                                                                                                      \
  do {                                                                                                                                \
    const size_t lds1 = (size_t)2 * FL_KC * DH * 2 + 4 * 32 * (2 * DH + 16);                                                         \
    const size_t lds2 = lds1 + 2 * FL_KC * sizeof(float);                                                                            \
    if (int rc = reserve_lds(reinterpret_cast<const void*>(attn_flash_bwd_dq<DH>), lds1, "attention_flash_bwd")) return rc;          \
    if (int rc = reserve_lds(reinterpret_cast<const void*>(attn_flash_bwd_dkv<DH>), lds2, "attention_flash_bwd")) return rc;         \
    hipLaunchKernelGGL((attn_flash_bwd_dq<DH>), grid, block, lds1, st, (const bf16_t*)d_o, (const bf16_t*)qkv, (const bf16_t*)o, lse,  \
                       (bf16_t*)dqkv, ws, N, heads, d, sc);                                                                          \
    hipLaunchKernelGGL((attn_flash_bwd_dkv<DH>), grid, block, lds2, st, (const bf16_t*)d_o, (const bf16_t*)qkv, lse, (const float*)ws, \
                       (bf16_t*)dqkv, N, heads, d, sc);                                                                              \
  } while (0)
  const int dh = fl_dh(d);
  if (dh == 64) FL_B(64); else if (dh == 96) FL_B(96); else FL_B(128);
#undef FL_B
  return check_launch("attention_flash_bwd");
}

}  // namespace dinox
