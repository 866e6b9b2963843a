// gemm_bf16_tnbig.hip -- the dW products  C[M,N] (+)= A[K,M]^T . B[K,N]  (K = every token of the batch, ~1e5) on BIG tiles.
//
// Why.  Round 2 measured what bounds the GEMMs of this path: not HBM and not the matrix pipe but the L2 -> CU request rate.
// Every product sits at  (bytes staged into LDS) / ~13.9 TB/s  (+ its epilogue traffic): 128 L2 channels x one 64-byte request per
// clock.  dW1 on 128 x 128 tiles stages K x 2 B x (M x N/128 + N x M/128) = 1.9 GB -> 137 us (measured 139); qkv 1.42 GB -> 102 us
// (99); the K = 1536 dX product 1.9 GB -> 137 us (140).  A fifth wave that touched every sector of a step ahead of the computing
// waves (doubling the requests) made dW1 63 % SLOWER whatever its lead, un-swizzled 256-byte row pieces changed nothing: requests
// are counted per 64 bytes, and what is left is to need fewer of them -- more reuse per byte brought into the CU, i.e. bigger tiles.
// The dW products are where that is cheapest: no epilogue to fuse, a tiny output, operands that lie in memory as [k][cols] rows.
//
// Tile 256 x 192 (or 384 x 128 for M <= 384), eight waves 4 x 2, each 64 x 96 (or 96 x 64) = 96 accumulator registers; one
// workgroup per CU.  K-step = 32 tokens: the slab of each operand is cut into [32 k][64 col] sub-images of 4 KiB (128-byte rows, the
// two 64-byte halves of a row XOR-ed with (k >> 1) & 1 so that the four k-rows of a transposing read hit disjoint banks); 7 or 8
// sub-images per step arrive by LDS-DMA (full 128-byte row pieces), FOUR steps deep (112 / 128 KiB of LDS, three steps = 84-96 KiB
// in flight per CU), retired by counted vmcnt, one barrier per step.  Fragments by ds_read_b64_tr_b16.  Bytes staged: dW1 1.1 GB
// instead of 1.9 GB, dW2 1.26 GB instead of 1.9 GB.
// K is split over the chip (one resident round of 256 workgroups); the splits ALWAYS meet in the caller's workspace by plain
// stores + the fixed-order reduction of gemm_bf16.hip (tn_reduce_kernel): bit-reproducible.  The bias gradient colsum(A) rides
// along as MFMA column sums on the N-tile-0 workgroups' ... no: on every workgroup's wc == 0 waves, k-steps dealt round-robin over
// the N tiles, exactly as in gemm_bf16_tn_dma.
#include "common.h"
#include "gemm_common.h"

namespace dinox {

typedef __attribute__((address_space(3))) void tb_lds_void;
typedef __attribute__((address_space(3))) s16x4 tb_lds_s16x4;

constexpr int TB_BK = 32, TB_STAGES = 4, TB_SUB = 4096;      // one [32 k][64 col] sub-image

__device__ __forceinline__ int tb_xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Transposed fragment for mfma_32x32x16 from a [32 k][64 col] sub-image: lane (col = l & 31, h = l >> 5) gets
// sub[k = kbase + 8h + j][c0 + col], j < 8 (c0 = 0 or 32).  Two ds_read_b64_tr_b16 (rows k0 and k0 + 4 of each 16-lane group).
__device__ __forceinline__ bf16x8 tb_frag(const char* __restrict__ sub, int kbase, int c0, int lane) {
  const int i = lane & 15, g = lane >> 4;
  const int q = i >> 2, pp = i & 3;
  const int colb = (c0 + 16 * (g & 1) + 4 * pp) * 2;          // byte offset of this lane's 4-element piece in its 128-byte k-row
  const int k0 = kbase + 8 * (g >> 1) + q, k1 = k0 + 4;
  const int o0 = k0 * 128 + (colb ^ (((k0 >> 1) & 1) << 6));
  const int o1 = k1 * 128 + (colb ^ (((k1 >> 1) & 1) << 6));
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tb_lds_s16x4*)(sub + o0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tb_lds_s16x4*)(sub + o1));
  s16x8 v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
  v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bf16x8, v);
}

// IM x JN accumulator blocks of 32 x 32 per wave (2 x 3: tile 256 x 192; 3 x 2: tile 384 x 128), waves 4 (M) x 2 (N).
template <int IM, int JN>
__global__ __launch_bounds__(512, 2) void gemm_bf16_tn_big(GemmParams p, int tiles_m, int tiles_n, int splits, int64_t k_per_split) {
  constexpr int TM = 4 * 32 * IM, TN = 2 * 32 * JN;
  constexpr int SA = TM / 64, SB = TN / 64, NSUB = SA + SB;            // sub-images per K-step: 4 + 3 or 6 + 2
  constexpr int STAGE = NSUB * TB_SUB;
  constexpr int NDMA = NSUB * 4;                                          // wave-instructions per K-step (1 KiB each): 28 or 32
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wv >> 1, wc = wv & 1;
  const int ntile = tiles_m * tiles_n;
  const int wid = tb_xcd_remap(blockIdx.x, ntile * splits);               // split-major: the tiles of a split share their panels in one L2
  const int split = wid / ntile, tile = wid % ntile;
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int64_t m0 = (int64_t)tm * TM, n0 = (int64_t)tn * TN;
  const int64_t kbeg = (int64_t)split * k_per_split;
  int64_t kend = kbeg + k_per_split;
  if (kend > p.K) kend = p.K;
  const int nk = kend > kbeg ? (int)ceil_div(kend - kbeg, (int64_t)TB_BK) : 0;

  // ---- staging.  DMA instruction d (0 .. NDMA-1) of a step fills 8 k-rows x 128 B of sub-image d >> 2: lane -> (row = 8 (d & 3) +
  // (lane >> 3), position c' = lane & 7); position c' of row k holds the logical 16-byte chunk c' ^ (4 ((k >> 1) & 1)).  Rows
  // k >= K fall outside the buffer descriptor and read as zeros; columns past the edge are clamped (never stored).
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)(p.K * p.lda * 2), 0x00020000);
  const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)(p.K * p.ldb * 2), 0x00020000);
  constexpr int PERW = (NDMA + 7) / 8;                                     // instructions per wave and step (4; waves 4..7 issue 3 of 28)
  static_assert(PERW == 4, "four (three) staging instructions per wave and step: the counted waits below are written for that");
  unsigned voff[4];      // (a fixed bound, not PERW: with a template-dependent array type as its offset operand the host pass of hipcc
                         //  7.2 silently fails to instantiate a kernel that calls raw_ptr_buffer_load_lds -- no diagnostic, no stub)
#pragma unroll
  for (int t = 0; t < PERW; ++t) {
    const int d = wv + 8 * t;
    const int sub = d >> 2, r = 8 * (d & 3) + (lane >> 3), cp = lane & 7;
    const int c = cp ^ (((r >> 1) & 1) << 2);
    if (sub < SA) {
      int64_t col = m0 + sub * 64 + c * 8;
      col = col < p.M ? col : p.M - 8;
      voff[t] = (unsigned)(((kbeg + r) * p.lda + col) * 2);
    } else {
      int64_t col = n0 + (sub - SA) * 64 + c * 8;
      col = col < p.N ? col : p.N - 8;
      voff[t] = (unsigned)(((kbeg + r) * p.ldb + col) * 2);
    }
  }
  const unsigned stepA = (unsigned)(TB_BK * p.lda * 2), stepB = (unsigned)(TB_BK * p.ldb * 2);
  const int nmine = (wv + 8 * (PERW - 1) < NDMA) ? PERW : PERW - 1;       // wave-uniform: 4 or 3
#define TB_STAGE(SLOT, KT)                                                                                               \
  do {                                                                                                                   \
    char* base_ = smem + (SLOT) * STAGE;                                                                                 \
    _Pragma("unroll") for (int t_ = 0; t_ < PERW; ++t_) {                                                                \
      const int d_ = wv + 8 * t_;                                                                                        \
      if (d_ < NDMA) {                                                                                                   \
        char* dst_ = base_ + (d_ >> 2) * TB_SUB + (d_ & 3) * 1024;                                                       \
        if ((d_ >> 2) < SA) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (tb_lds_void*)dst_, 16, voff[t_] + (KT) * stepA, 0, 0, 0); \
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (tb_lds_void*)dst_, 16, voff[t_] + (KT) * stepB, 0, 0, 0);    \
      }                                                                                                                  \
    }                                                                                                                    \
  } while (0)

  f32x16 acc[IM][JN];
#pragma unroll
  for (int i = 0; i < IM; ++i)
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // bias gradient riding along: colsum[m] = (A^T . 1)[m] on the wc == 0 waves, k-steps dealt round-robin over the N tiles
  const bool cs_wave = p.colsum != nullptr && wc == 0;
  f32x16 csacc[IM];
#pragma unroll
  for (int i = 0; i < IM; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) csacc[i][e] = 0.f;
  s16x8 ones_s;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones_s[j] = (short)0x3F80;   // bf16 1.0
  const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_s);

  // ---- K loop: four-deep ring, counted waits.  At the top of step kt the stages kt + 1 and kt + 2 may still be in flight.
  for (int s = 0; s < TB_STAGES - 1 && s < nk; ++s) TB_STAGE(s, s);
  for (int kt = 0; kt < nk; ++kt) {
    const int inflight = (nk - 1 - kt) < 2 ? (nk - 1 - kt) : 2;            // stages issued after stage kt
    if (inflight == 2) {
      if (nmine == PERW) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else if (inflight == 1) {
      if (nmine == PERW) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // my fragment reads of step kt - 1: its slot is refilled below
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (kt + TB_STAGES - 1 < nk) TB_STAGE((kt + TB_STAGES - 1) & (TB_STAGES - 1), kt + TB_STAGES - 1);
    const char* sa = smem + (kt & (TB_STAGES - 1)) * STAGE;
    const char* sb = sa + SA * TB_SUB;
    const bool cs_now = cs_wave && (kt % tiles_n) == tn;      // wave-uniform
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[IM], bfr[JN];
#pragma unroll
      for (int i = 0; i < IM; ++i) {
        const int col = wr * 32 * IM + i * 32;                  // column of the tile's M range
        af[i] = tb_frag(sa + (col >> 6) * TB_SUB, ks * 16, col & 63, lane);
      }
#pragma unroll
      for (int j = 0; j < JN; ++j) {
        const int col = wc * 32 * JN + j * 32;
        bfr[j] = tb_frag(sb + (col >> 6) * TB_SUB, ks * 16, col & 63, lane);
      }
#pragma unroll
      for (int i = 0; i < IM; ++i)
#pragma unroll
        for (int j = 0; j < JN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      if (cs_now) {
#pragma unroll
        for (int i = 0; i < IM; ++i) csacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], ones, csacc[i], 0, 0, 0);
      }
    }
  }

  // ---- partial results into the workspace: [split][M][N] and [split * tiles_n + tn][M]
  float* const part = (float*)p.ws + (int64_t)split * p.M * p.N;
  if (cs_wave && (lane & 31) == 0) {   // every column of csacc holds the same sums: lanes 0 and 32 own all 32 rows
    float* csp = (float*)p.ws + (int64_t)splits * p.M * p.N + ((int64_t)split * tiles_n + tn) * p.M;
#pragma unroll
    for (int i = 0; i < IM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t m = m0 + wr * 32 * IM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (m < p.M) csp[m] = csacc[i][e];
      }
  }
#pragma unroll
  for (int j = 0; j < JN; ++j) {
    const int64_t n = n0 + wc * 32 * JN + j * 32 + (lane & 31);
    if (n >= p.N) continue;
#pragma unroll
    for (int i = 0; i < IM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t m = m0 + wr * 32 * IM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (m < p.M) part[m * p.N + n] = acc[i][j][e];
      }
  }
}

#undef TB_STAGE

// Which big-tile form (0: none, 1: 256 x 192, 2: 384 x 128, 3: 256 x 256) and its split plan.  Needs the deterministic-reduction envelope
// (one contiguous fp32 [M][N] result, plain epilogue) and enough K to fill a four-deep ring on every workgroup.
int tn_big_plan(const GemmParams& p, int& tiles_m, int& tiles_n, int& splits, int64_t& kps) {
  if (p.transA != 1 || p.transB != 1 || p.batch != 1 || p.ldc != p.N || p.out_dtype != DINOX_F32 || p.in_dtype != DINOX_BF16) return 0;
  if ((p.epilogue & ~DINOX_EPI_ACCUM) != 0) return 0;
  if ((p.M & 7) || (p.N & 7) || p.M < 64 || p.N < 64 || (p.lda & 7) || (p.ldb & 7)) return 0;
  if ((((uintptr_t)p.A) | ((uintptr_t)p.B)) & 15) return 0;
  if (p.K * p.lda * 2 >= (int64_t)0x7fffffff || p.K * p.ldb * 2 >= (int64_t)0x7fffffff) return 0;
  if (p.K < 8192) return 0;
  static const bool off = getenv("DINOX_TN_BIG_OFF") != nullptr;
  if (off) return 0;
  // Three tile shapes (1: 256 x 192, 2: 384 x 128, 3: 256 x 256); every workgroup of the ONE resident round does kps k-rows of a whole
  // tile (padding included), so the product's time goes with  kps x TM x TN  of the shape's own split plan -- ViT-L's 4096 x 1024
  // dW1 is 96 tiles of 256 x 192 (two splits: 192 of 256 CUs busy, 11 % of the tile area padding) but 64 tiles of 256 x 256 (four
  // splits, every CU, no padding).  Ties go to the shape that stages fewer bytes per k-row (TM + TN).  DINOX_TN_FORM=1|2|3 forces one (A/B).
  const char* fenv = getenv("DINOX_TN_FORM");                 // (read per call: tools flip it between launches)
  const int forced = fenv ? atoi(fenv) : 0;
  static const int TMs[4] = {0, 256, 384, 256}, TNs[4] = {0, 192, 128, 256};
  int form = 0;
  int64_t best = 0;
  for (int f = 1; f <= 3; ++f) {
    if (forced >= 1 && forced <= 3 && f != forced) continue;
    const int tm_ = (int)ceil_div(p.M, (int64_t)TMs[f]), tn_ = (int)ceil_div(p.N, (int64_t)TNs[f]);
    const int64_t nt = (int64_t)tm_ * tn_;
    if (nt > 256) continue;
    int sp = (int)(256 / nt);                                   // one resident round: one workgroup per CU
    const int64_t max_splits = ceil_div(p.K, (int64_t)(8 * TB_BK));
    if (sp > max_splits) sp = (int)max_splits;
    if (sp < 1) sp = 1;
    const int64_t kp = ceil_div(ceil_div(p.K, (int64_t)sp), (int64_t)TB_BK) * TB_BK;
    const int64_t cost = kp * TMs[f] * TNs[f];
    if (form == 0 || cost < best || (cost == best && TMs[f] + TNs[f] < TMs[form] + TNs[form])) {
      form = f; best = cost; tiles_m = tm_; tiles_n = tn_; kps = kp;
    }
  }
  if (!form) return 0;
  splits = (int)ceil_div(p.K, kps);
  return form;
}

int64_t tn_big_ws_bytes(const GemmParams& p) {
  int tiles_m, tiles_n, splits;
  int64_t kps;
  if (!tn_big_plan(p, tiles_m, tiles_n, splits, kps)) return 0;
  return ((int64_t)splits * p.M * p.N + (p.colsum ? (int64_t)splits * tiles_n * p.M : 0)) * (int64_t)sizeof(float);
}

// Launches the product into the workspace; the caller (launch_gemm_bf16) runs the fixed-order reduction afterwards.
int launch_gemm_bf16_tn_big(const GemmParams& p, hipStream_t st, int& splits_out, int& tiles_n_out) {
  int tiles_m, tiles_n, splits;
  int64_t kps;
  const int form = tn_big_plan(p, tiles_m, tiles_n, splits, kps);
  if (!form || !p.ws) return DINOX_EUNSUPPORTED;
  splits_out = splits;
  tiles_n_out = tiles_n;
  const unsigned grid = (unsigned)(tiles_m * tiles_n * splits);
  if (form == 1) {
    constexpr size_t lds = (size_t)TB_STAGES * 7 * TB_SUB;
    if (int rc = reserve_lds(reinterpret_cast<const void*>(gemm_bf16_tn_big<2, 3>), lds, "gemm_bf16_tn_big")) return rc;
    hipLaunchKernelGGL((gemm_bf16_tn_big<2, 3>), dim3(grid), dim3(512), lds, st, p, tiles_m, tiles_n, splits, kps);
  } else if (form == 2) {
    constexpr size_t lds = (size_t)TB_STAGES * 8 * TB_SUB;
    if (int rc = reserve_lds(reinterpret_cast<const void*>(gemm_bf16_tn_big<3, 2>), lds, "gemm_bf16_tn_big")) return rc;
    hipLaunchKernelGGL((gemm_bf16_tn_big<3, 2>), dim3(grid), dim3(512), lds, st, p, tiles_m, tiles_n, splits, kps);
  } else {
    constexpr size_t lds = (size_t)TB_STAGES * 8 * TB_SUB;
    if (int rc = reserve_lds(reinterpret_cast<const void*>(gemm_bf16_tn_big<2, 4>), lds, "gemm_bf16_tn_big")) return rc;
    hipLaunchKernelGGL((gemm_bf16_tn_big<2, 4>), dim3(grid), dim3(512), lds, st, p, tiles_m, tiles_n, splits, kps);
  }
  return check_launch("gemm_bf16_tn_big");
}

}  // namespace dinox
