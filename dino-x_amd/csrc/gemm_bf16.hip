// gemm_bf16.hip -- bf16 MFMA GEMMs (v_mfma_f32_32x32x16_bf16, fp32 accumulate) for gfx950.
//
//   gemm_bf16_nt : C[M,N] = A[M,K] . B[N,K]^T   both operands K-contiguous (forward products and, with the
//                  pre-transposed bf16 weight copies, every dX product).  Fragments by ds_read_b128 from an
//                  XOR-swizzled [rows][64] bf16 LDS image (chunk ^= (row>>1)&7: conflict-free for the
//                  ds_read_b128 lane groups of gfx950, MI355X_MICROARCH.md LDS table).
//   gemm_bf16_tn : C[M,N] = A[K,M]^T . B[K,N]   both operands stored reduction-major (dW = dY^T X with the
//                  token dimension as K).  Tiles are staged as they lie in memory ([k][m], 256-B rows) and
//                  the MFMA operands are fetched with the gfx950 transposing LDS read ds_read_b64_tr_b16;
//                  64-B granules are XOR-swizzled with (k&3) so the 4 k-rows of a read hit distinct banks.
//                  K (= tokens, ~1e5) is split across workgroups; partial tiles meet through fp32 atomics.
//
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 2x2 MFMA tiles, 64 accumulator VGPRs.
// LDS: 2 stages x (16 KiB A + 16 KiB B) = 64 KiB -> 2 workgroups per CU.  Global loads are issued one
// K-step ahead into registers and written to the other LDS stage after the MFMAs (one barrier per step).
// Workgroup ids are remapped so that each XCD walks a contiguous run of tiles: the N-tiles that share an
// A row-panel hit that XCD's private L2 (cdna_hip_programming.md T1, bijective form).
#include <cstdlib>

#include <cstring>

#include "common.h"
#include "gemm_common.h"

namespace dinox {

constexpr int GB_BM = 128, GB_BN = 128, GB_BK = 64, GB_THREADS = 256;
constexpr int GB_TILE_BYTES = 128 * 64 * 2;  // 16 KiB per operand tile (NT: [128][64]; TN: [64][128])

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ uint4 ldg16(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }

// XCD-aware bijective remap of a 1-D workgroup id (8 XCDs, round-robin dispatch).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ------------------------------------------------------------------------------------------ NT
struct NtStage {
  uint4 a[4], b[4];
};

// thread t loads chunk c = t&7 (8 bf16 = 16 B along K) of rows (t>>3) + 32*i
__device__ __forceinline__ void nt_load(NtStage& s, const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, int64_t lda,
                                        int64_t ldb, int64_t m0, int64_t n0, int64_t k0, int64_t M, int64_t N, int64_t K) {
  const int c = threadIdx.x & 7, r0 = threadIdx.x >> 3;
  const int64_t k = k0 + c * 8;
  const bool kin = k < K;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + 32 * i;
    int64_t gm = m0 + r, gn = n0 + r;
    gm = gm < M ? gm : M - 1;  // clamp: rows past the edge are computed but never stored
    gn = gn < N ? gn : N - 1;
    s.a[i] = kin ? ldg16(A + gm * lda + k) : make_uint4(0, 0, 0, 0);
    s.b[i] = kin ? ldg16(B + gn * ldb + k) : make_uint4(0, 0, 0, 0);
  }
}

__device__ __forceinline__ void nt_store(const NtStage& s, char* __restrict__ sa, char* __restrict__ sb) {
  const int c = threadIdx.x & 7, r0 = threadIdx.x >> 3;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + 32 * i;
    const int off = r * 128 + ((c ^ ((r >> 1) & 7)) << 4);
    *reinterpret_cast<uint4*>(sa + off) = s.a[i];
    *reinterpret_cast<uint4*>(sb + off) = s.b[i];
  }
}

template <int OUT_DT>
__global__ __launch_bounds__(GB_THREADS, 2) void gemm_bf16_nt(GemmParams p, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wr = wv >> 1, wc = wv & 1;
  const int ntile = tiles_m * tiles_n;
  const int tile = xcd_remap(blockIdx.x, ntile);
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int64_t m0 = (int64_t)tm * GB_BM, n0 = (int64_t)tn * GB_BN;
  const int64_t bz = blockIdx.y;
  const bf16_t* A = (const bf16_t*)p.A + bz * p.strideA;
  const bf16_t* B = (const bf16_t*)p.B + bz * p.strideB;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = (int)ceil_div(p.K, GB_BK);
  NtStage st;
  nt_load(st, A, B, p.lda, p.ldb, m0, n0, 0, p.M, p.N, p.K);
  nt_store(st, smem, smem + GB_TILE_BYTES);
  __syncthreads();

  const int frow = lane & 31, fh = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    char* sa = smem + cur * 2 * GB_TILE_BYTES;
    char* sb = sa + GB_TILE_BYTES;
    if (kt + 1 < nk) nt_load(st, A, B, p.lda, p.ldb, m0, n0, (int64_t)(kt + 1) * GB_BK, p.M, p.N, p.K);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ra = wr * 64 + i * 32 + frow;
        const int rb = wc * 64 + i * 32 + frow;
        const int kc = 2 * ks + fh;
        af[i] = *reinterpret_cast<const bf16x8*>(sa + ra * 128 + ((kc ^ ((ra >> 1) & 7)) << 4));
        bfr[i] = *reinterpret_cast<const bf16x8*>(sb + rb * 128 + ((kc ^ ((rb >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) {
      char* na = smem + (cur ^ 1) * 2 * GB_TILE_BYTES;
      nt_store(st, na, na + GB_TILE_BYTES);
    }
    __syncthreads();
  }

#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int64_t n = n0 + wc * 64 + j * 32 + (lane & 31);
    if (n >= p.N) continue;
    const float bias = (p.epilogue & DINOX_EPI_BIAS) ? p.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t m = m0 + wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (m < p.M) epilogue_store<OUT_DT>(p, bz, m, n, acc[i][j][e], bias);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ TN
struct TnStage {
  uint4 a[4], b[4];
};

// tile rows are K (64 rows of 256 B = 128 bf16 along M or N); thread t: chunk c = t&15, rows (t>>4) + 16*i
__device__ __forceinline__ void tn_load(TnStage& s, const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, int64_t lda,
                                        int64_t ldb, int64_t m0, int64_t n0, int64_t k0, int64_t kend, int64_t M, int64_t N) {
  const int c = threadIdx.x & 15, r0 = threadIdx.x >> 4;
  const int64_t m = m0 + c * 8, n = n0 + c * 8;
  const bool min_ = m < M, nin = n < N;  // M, N are multiples of 8: a chunk is wholly in or out
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t k = k0 + r0 + 16 * i;
    const bool kin = k < kend;
    s.a[i] = (kin && min_) ? ldg16(A + k * lda + m) : make_uint4(0, 0, 0, 0);
    s.b[i] = (kin && nin) ? ldg16(B + k * ldb + n) : make_uint4(0, 0, 0, 0);
  }
}

// 16-B chunk c of k-row r goes to granule ((c>>2) ^ (r&3)), slot (c&3)
__device__ __forceinline__ void tn_store(const TnStage& s, char* __restrict__ sa, char* __restrict__ sb) {
  const int c = threadIdx.x & 15, r0 = threadIdx.x >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + 16 * i;
    const int off = r * 256 + ((((c >> 2) ^ (r & 3)) << 6) | ((c & 3) << 4));
    *reinterpret_cast<uint4*>(sa + off) = s.a[i];
    *reinterpret_cast<uint4*>(sb + off) = s.b[i];
  }
}

// Transposed fragment for mfma_32x32x16: lane (col = l&31, h = l>>5) gets tile[k = kbase + 8h + j][col0 + col], j<8.
__device__ __forceinline__ bf16x8 tn_frag(const char* __restrict__ tile, int kbase, int col0, int lane) {
  const int i = lane & 15, g = lane >> 4;
  const int q = i >> 2, pp = i & 3;
  const int colb = (col0 + 16 * (g & 1) + 4 * pp) * 2;  // byte offset of this lane's 4-element piece in its k-row
  const int k0 = kbase + 8 * (g >> 1) + q;
  const int k1 = k0 + 4;
  const int o0 = k0 * 256 + ((((colb >> 6) ^ (k0 & 3)) << 6) | (colb & 63));
  const int o1 = k1 * 256 + ((((colb >> 6) ^ (k1 & 3)) << 6) | (colb & 63));
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + o0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + o1));
  s16x8 v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
  v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bf16x8, v);
}

template <int OUT_DT>
__global__ __launch_bounds__(GB_THREADS, 2) void gemm_bf16_tn(GemmParams p, int tiles_m, int tiles_n, int splits,
                                                              int64_t k_per_split) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wr = wv >> 1, wc = wv & 1;
  // Flattened work id = (split-major, tile-minor), then XCD-remapped: each XCD owns a contiguous run of ids, i.e. all
  // tiles of ~1.75 K-splits, and they are co-resident (63 workgroups on 64 slots).  The tiles of one split stream the
  // same K-range, so the A panel shared along tn and the B panel shared along tm are fetched from HBM once per XCD
  // and re-read from that XCD's L2.  (Tile-major order scattered a split over all 8 private L2s: rocprof FETCH_SIZE
  // showed 2.8x the algorithmic bytes and the kernel ran at the HBM roofline of that inflated traffic.)
  const int ntile = tiles_m * tiles_n;
  const int wid = xcd_remap(blockIdx.x, ntile * splits);
  const int split = wid / ntile, tile = wid % ntile;
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int64_t m0 = (int64_t)tm * GB_BM, n0 = (int64_t)tn * GB_BN;
  const int64_t bz = blockIdx.y;
  const int64_t kbeg = (int64_t)split * k_per_split;
  int64_t kend = kbeg + k_per_split;
  if (kend > p.K) kend = p.K;
  const bf16_t* A = (const bf16_t*)p.A + bz * p.strideA;
  const bf16_t* B = (const bf16_t*)p.B + bz * p.strideB;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = kend > kbeg ? (int)ceil_div(kend - kbeg, GB_BK) : 0;
  // Bias gradient riding along: colsum[m] = sum_k A[k][m] = (A^T . 1)[m].  The wc == 0 waves multiply their A
  // fragments with an all-ones B fragment on every tiles_n-th K-step (k-steps are dealt round-robin over the
  // N-tiles of a row panel, so the extra MFMAs are spread evenly over the grid and no VALU work is added).
  const bool cs_wave = p.colsum != nullptr && wc == 0;
  f32x16 csacc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) csacc[i][e] = 0.f;
  s16x8 ones_s;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones_s[j] = (short)0x3F80;   // bf16 1.0
  const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_s);
  TnStage st;
  if (nk > 0) {
    tn_load(st, A, B, p.lda, p.ldb, m0, n0, kbeg, kend, p.M, p.N);
    tn_store(st, smem, smem + GB_TILE_BYTES);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const char* sa = smem + cur * 2 * GB_TILE_BYTES;
    const char* sb = sa + GB_TILE_BYTES;
    if (kt + 1 < nk) tn_load(st, A, B, p.lda, p.ldb, m0, n0, kbeg + (int64_t)(kt + 1) * GB_BK, kend, p.M, p.N);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = tn_frag(sa, ks * 16, wr * 64 + i * 32, lane);
        bfr[i] = tn_frag(sb, ks * 16, wc * 64 + i * 32, lane);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      if (cs_wave && (kt % tiles_n) == tn) {                 // wave-uniform
        csacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], ones, csacc[0], 0, 0, 0);
        csacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], ones, csacc[1], 0, 0, 0);
      }
    }
    if (kt + 1 < nk) {
      char* na = smem + (cur ^ 1) * 2 * GB_TILE_BYTES;
      tn_store(st, na, na + GB_TILE_BYTES);
    }
    __syncthreads();
  }

  // Split-K results meet either in the caller's workspace -- partial tiles [split][M][N] and partial column sums
  // [split * tiles_n + tn][M] by PLAIN stores, summed in split order by tn_reduce_kernel: bit-reproducible, and plain stores run at
  // 6 TB/s where memory-side fp32 atomics run at 1.3 TB/s (33 MB per dW launch) -- or, without a workspace, through atomics.
  const bool det = p.ws != nullptr && splits > 1;
  if (cs_wave && (lane & 31) == 0) {   // every column of csacc holds the same sums: lanes 0 and 32 own all 32 rows
    float* csp = det ? (float*)p.ws + (int64_t)splits * p.M * p.N + ((int64_t)split * tiles_n + tn) * p.M : nullptr;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t m = m0 + wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (m < p.M) {
          if (det) csp[m] = csacc[i][e];
          else atomicAdd(p.colsum + m, csacc[i][e]);
        }
      }
  }

#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int64_t n = n0 + wc * 64 + j * 32 + (lane & 31);
    if (n >= p.N) continue;
    const float bias = (p.epilogue & DINOX_EPI_BIAS) ? p.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t m = m0 + wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (m >= p.M) continue;
        if (det)
          ((float*)p.ws)[((int64_t)split * p.M + m) * p.N + n] = acc[i][j][e];              // alpha / ACCUM are applied by the reduction
        else if (splits > 1)
          atomicAdd((float*)p.C + bz * p.strideC + m * p.ldc + n, acc[i][j][e] * p.alpha);  // C zeroed / accumulating
        else
          epilogue_store<OUT_DT>(p, bz, m, n, acc[i][j][e], bias);
      }
    }
  }
}

typedef __attribute__((address_space(3))) void lds_void_t;

template <int OUT_DT>
__global__ __launch_bounds__(GB_THREADS, 2) void gemm_bf16_tn_dma(GemmParams p, int tiles_m, int tiles_n, int splits, int64_t k_per_split) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wr = wv >> 1, wc = wv & 1;
  // Flattened work id = (split-major, tile-minor), then XCD-remapped: each XCD owns a contiguous run of ids, i.e. all
  // tiles of ~1.75 K-splits, and they are co-resident (63 workgroups on 64 slots).  The tiles of one split stream the
  // same K-range, so the A panel shared along tn and the B panel shared along tm are fetched from HBM once per XCD
  // and re-read from that XCD's L2.  (Tile-major order scattered a split over all 8 private L2s: rocprof FETCH_SIZE
  // showed 2.8x the algorithmic bytes and the kernel ran at the HBM roofline of that inflated traffic.)
  const int ntile = tiles_m * tiles_n;
  const int wid = xcd_remap(blockIdx.x, ntile * splits);
  const int split = wid / ntile, tile = wid % ntile;
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int64_t m0 = (int64_t)tm * GB_BM, n0 = (int64_t)tn * GB_BN;
  const int64_t bz = blockIdx.y;
  const int64_t kbeg = (int64_t)split * k_per_split;
  int64_t kend = kbeg + k_per_split;
  if (kend > p.K) kend = p.K;
  const bf16_t* A = (const bf16_t*)p.A + bz * p.strideA;
  const bf16_t* B = (const bf16_t*)p.B + bz * p.strideB;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = kend > kbeg ? (int)ceil_div(kend - kbeg, GB_BK) : 0;
  // Bias gradient riding along: colsum[m] = sum_k A[k][m] = (A^T . 1)[m].  The wc == 0 waves multiply their A
  // fragments with an all-ones B fragment on every tiles_n-th K-step (k-steps are dealt round-robin over the
  // N-tiles of a row panel, so the extra MFMAs are spread evenly over the grid and no VALU work is added).
  const bool cs_wave = p.colsum != nullptr && wc == 0;
  f32x16 csacc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) csacc[i][e] = 0.f;
  s16x8 ones_s;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones_s[j] = (short)0x3F80;   // bf16 1.0
  const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_s);
  // Direct-to-LDS staging: each wave issues 4 + 4 buffer_load_dwordx4 ... lds per K-step, every one moving 4 k-rows
  // x 256 B.  LDS slot (r, chunk') receives logical chunk c = (((chunk'>>2) ^ (r&3))<<2) | (chunk'&3) (the granule
  // swizzle, applied to the SOURCE address).  Rows k >= K fall outside the buffer descriptor and read as zeros.
  const int wvu = __builtin_amdgcn_readfirstlane(wv);
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)(p.K * p.lda * 2), 0x00020000);
  const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)(p.K * p.ldb * 2), 0x00020000);
  unsigned voffA[4], voffB[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = (wvu * 4 + q) * 4 + (lane >> 4);
    const int cp = lane & 15;
    const int c = (((cp >> 2) ^ (r & 3)) << 2) | (cp & 3);
    int64_t ma = m0 + c * 8, nb = n0 + c * 8;
    ma = ma < p.M ? ma : p.M - 8;                      // columns past the edge only feed rows/cols that are never stored
    nb = nb < p.N ? nb : p.N - 8;
    voffA[q] = (unsigned)(((kbeg + r) * p.lda + ma) * 2);
    voffB[q] = (unsigned)(((kbeg + r) * p.ldb + nb) * 2);
  }
  const unsigned stepA = (unsigned)(GB_BK * p.lda * 2), stepB = (unsigned)(GB_BK * p.ldb * 2);
  auto stage = [&](int buf, int kt) {
    char* sa = smem + buf * 2 * GB_TILE_BYTES + wvu * 4096;
    char* sb = sa + GB_TILE_BYTES;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_t*)(sa + q * 1024), 16, voffA[q] + kt * stepA, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_t*)(sb + q * 1024), 16, voffB[q] + kt * stepB, 0, 0, 0);
    }
  };
  if (nk > 0) stage(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const char* sa = smem + cur * 2 * GB_TILE_BYTES;
    const char* sb = sa + GB_TILE_BYTES;
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = tn_frag(sa, ks * 16, wr * 64 + i * 32, lane);
        bfr[i] = tn_frag(sb, ks * 16, wc * 64 + i * 32, lane);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      if (cs_wave && (kt % tiles_n) == tn) {                 // wave-uniform
        csacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], ones, csacc[0], 0, 0, 0);
        csacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], ones, csacc[1], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // Split-K results meet either in the caller's workspace -- partial tiles [split][M][N] and partial column sums
  // [split * tiles_n + tn][M] by PLAIN stores, summed in split order by tn_reduce_kernel: bit-reproducible, and plain stores run at
  // 6 TB/s where memory-side fp32 atomics run at 1.3 TB/s (33 MB per dW launch) -- or, without a workspace, through atomics.
  const bool det = p.ws != nullptr && splits > 1;
  if (cs_wave && (lane & 31) == 0) {   // every column of csacc holds the same sums: lanes 0 and 32 own all 32 rows
    float* csp = det ? (float*)p.ws + (int64_t)splits * p.M * p.N + ((int64_t)split * tiles_n + tn) * p.M : nullptr;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t m = m0 + wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (m < p.M) {
          if (det) csp[m] = csacc[i][e];
          else atomicAdd(p.colsum + m, csacc[i][e]);
        }
      }
  }

#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int64_t n = n0 + wc * 64 + j * 32 + (lane & 31);
    if (n >= p.N) continue;
    const float bias = (p.epilogue & DINOX_EPI_BIAS) ? p.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t m = m0 + wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (m >= p.M) continue;
        if (det)
          ((float*)p.ws)[((int64_t)split * p.M + m) * p.N + n] = acc[i][j][e];              // alpha / ACCUM are applied by the reduction
        else if (splits > 1)
          atomicAdd((float*)p.C + bz * p.strideC + m * p.ldc + n, acc[i][j][e] * p.alpha);  // C zeroed / accumulating
        else
          epilogue_store<OUT_DT>(p, bz, m, n, acc[i][j][e], bias);
      }
    }
  }
}

// Second stage of the deterministic split-K reduction: C (+)= alpha * sum_s part[s], colsum (+)= sum_t cs_part[t], always in the same
// order.  A block of 256 threads owns 64 float4 (or 64 column sums): four groups of 64 threads each take a quarter of the splits in
// increasing order -- their loads in flight together, eight per round -- and meet in LDS as ((g0 + g1) + (g2 + g3)).  (A first
// version walked all splits per thread: 14-85 dependent round trips, 44 us per launch for 50 MB.)
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ part, int splits, int64_t MN, float alpha, float* __restrict__ C,
                                                       int accumulate, const float* __restrict__ cs_part, int cs_terms, int64_t M,
                                                       float* __restrict__ colsum, int c_blocks) {
  __shared__ float4 red[4][64];
  const int l = threadIdx.x & 63, g = threadIdx.x >> 6;
  if ((int)blockIdx.x < c_blocks) {
    const int64_t n4 = MN >> 2, i = (int64_t)blockIdx.x * 64 + l;
    const int per = (splits + 3) >> 2, s_lo = g * per, s_hi = s_lo + per < splits ? s_lo + per : splits;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4) {
      for (int s0 = s_lo; s0 < s_hi; s0 += 8) {
        float4 b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          b[u] = s0 + u < s_hi ? reinterpret_cast<const float4*>(part + (int64_t)(s0 + u) * MN)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          a.x += b[u].x; a.y += b[u].y; a.z += b[u].z; a.w += b[u].w;
        }
      }
    }
    red[g][l] = a;
    __syncthreads();
    if (g == 0 && i < n4) {
      const float4 r0 = red[0][l], r1 = red[1][l], r2 = red[2][l], r3 = red[3][l];
      float4 c = accumulate ? reinterpret_cast<const float4*>(C)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      c.x += alpha * ((r0.x + r1.x) + (r2.x + r3.x));
      c.y += alpha * ((r0.y + r1.y) + (r2.y + r3.y));
      c.z += alpha * ((r0.z + r1.z) + (r2.z + r3.z));
      c.w += alpha * ((r0.w + r1.w) + (r2.w + r3.w));
      reinterpret_cast<float4*>(C)[i] = c;
    }
    if (blockIdx.x == 0 && threadIdx.x < (MN & 3)) {                      // MN % 4 tail (never on this path: M, N are multiples of 8)
      const int64_t j = (n4 << 2) + threadIdx.x;
      float t = 0.f;
      for (int s2 = 0; s2 < splits; ++s2) t += part[(int64_t)s2 * MN + j];
      C[j] = (accumulate ? C[j] : 0.f) + alpha * t;
    }
  } else {                                                                 // column sums: 64 of them per block
    const int64_t m = (int64_t)((int)blockIdx.x - c_blocks) * 64 + l;
    const int per = (cs_terms + 3) >> 2, t_lo = g * per, t_hi = t_lo + per < cs_terms ? t_lo + per : cs_terms;
    float a = 0.f;
    if (m < M) {
      for (int t0 = t_lo; t0 < t_hi; t0 += 8) {
        float b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) b[u] = t0 + u < t_hi ? cs_part[(int64_t)(t0 + u) * M + m] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) a += b[u];
      }
    }
    red[g][l].x = a;
    __syncthreads();
    if (g == 0 && m < M) colsum[m] = (accumulate ? colsum[m] : 0.f) + ((red[0][l].x + red[1][l].x) + (red[2][l].x + red[3][l].x));
  }
}

static int launch_tn_reduce(const GemmParams& p, int splits, int tiles_n, hipStream_t st) {
  const int64_t MN = p.M * p.N;
  const int c_blocks = (int)ceil_div(MN >> 2, (int64_t)64);
  const int cs_blocks = p.colsum ? (int)ceil_div(p.M, (int64_t)64) : 0;
  const float* part = (const float*)p.ws;
  hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)(c_blocks + cs_blocks)), dim3(256), 0, st, part, splits, MN, p.alpha, (float*)p.C,
                     (p.epilogue & DINOX_EPI_ACCUM) ? 1 : 0, part + (int64_t)splits * MN, splits * tiles_n, p.M, p.colsum, c_blocks);
  return check_launch("gemm_bf16_tn_reduce");
}

// Split plan of the TN products: one resident round (2 workgroups per CU x 256 CUs = 512 slots; a grid of 513..1023 would run two
// rounds), at least 4 K-steps per workgroup.  Shared by the launcher and dinox_gemm_ws_bytes.
static void tn_split_plan(const GemmParams& p, int& splits, int64_t& kps) {
  const int tiles_m = (int)ceil_div(p.M, GB_BM), tiles_n = (int)ceil_div(p.N, GB_BN);
  const int64_t ntile = (int64_t)tiles_m * tiles_n;
  splits = 1;
  const bool plain = (p.epilogue & ~DINOX_EPI_ACCUM) == 0 && p.out_dtype == DINOX_F32;
  if (plain) {
    const int64_t slots = 512, have = ntile * p.batch;
    splits = (int)(slots / have);
    const int64_t max_splits = ceil_div(p.K, 4 * GB_BK);
    if (splits > max_splits) splits = (int)max_splits;
    if (splits < 1) splits = 1;
  }
  kps = ceil_div(ceil_div(p.K, splits), GB_BK) * GB_BK;
  splits = (int)ceil_div(p.K, kps);
}

static bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
int tn_big_plan(const GemmParams& p, int& tiles_m, int& tiles_n, int& splits, int64_t& kps);     // gemm_bf16_tnbig.hip
int64_t tn_big_ws_bytes(const GemmParams& p);
int launch_gemm_bf16_tn_big(const GemmParams& p, hipStream_t st, int& splits_out, int& tiles_n_out);
bool gemm_bf16_nt_glds_ok(const GemmParams& p);
int launch_gemm_bf16_nt_glds(const GemmParams& p, hipStream_t st);
bool gemm_bf16_nt_areg_ok(const GemmParams& p);
int launch_gemm_bf16_nt_areg(const GemmParams& p, hipStream_t st);
bool gemm_bf16_nt_pp_ok(const GemmParams& p);                                                   // gemm_bf16_pp.hip
int launch_gemm_bf16_nt_pp(const GemmParams& p, hipStream_t st);
bool gemm_bf16_nt_pp128_ok(const GemmParams& p);                                                // gemm_bf16_pp128.hip
int launch_gemm_bf16_nt_pp128(const GemmParams& p, hipStream_t st);
bool gemm_bf16_nt_pp384_ok(const GemmParams& p);                                                // gemm_bf16_pp384.hip
int launch_gemm_bf16_nt_pp384(const GemmParams& p, hipStream_t st);

// Which NT products take the persistent ping-pong kernels (gemm_bf16_pp.hip: 256 x 256 tiles, gemm_bf16_pp128.hip: 256 x 128), and which
// of the two.  DINOX_NT_PP (read per call, so one process can A/B and the tests can force small shapes onto them): 0 = never; 1 = every
// product inside the envelope, tile width by shape; 2 / 3 = every product on the 128-wide / 256-wide tiles; unset = the measured policy.
static const char* nt_pp_choice(const GemmParams& p) {
  if (p.transA || p.transB) return nullptr;
  const char* e = getenv("DINOX_NT_PP");
  const int mode = e ? atoi(e) : -1;
  if (mode == 0) return nullptr;
  const bool ok256 = gemm_bf16_nt_pp_ok(p), ok128 = gemm_bf16_nt_pp128_ok(p);
  {
    // The full-row tile for N = 384 (gemm_bf16_pp384.hip; DINOX_NT_PP384 read per call: 0 = never, 1 = every product in its envelope,
    // unset = the measured policy).  tools/pp384_check.py, M = 102 912, interleaved with the 256 x 128 kernel on one box: dX K 1536
    // 139 -> 118 us, dX K 1152 95 -> 86 (M = 25 728: 46 -> 43, 37 -> 34); K = 384 a tie (43 vs 45); with the fp32 residual epilogue a tie
    // too (fc2 175 vs 175: its two rounds of tiles run in lockstep, so the 316 MB of residual traffic are not hidden behind K loops).
    const char* e3 = getenv("DINOX_NT_PP384");
    const int m3 = e3 ? atoi(e3) : -1;
    if (m3 != 0 && mode != 2 && mode != 3 && gemm_bf16_nt_pp384_ok(p) &&
        (m3 > 0 || (p.out_dtype == DINOX_BF16 && !(p.epilogue & DINOX_EPI_RESIDUAL) && p.K >= 768 && p.M >= 8192)))
      return "gemm_bf16_nt_pp384";
  }
  if (mode == 2) return ok128 ? "gemm_bf16_nt_pp128" : nullptr;
  if (mode == 3) return ok256 ? "gemm_bf16_nt_pp" : nullptr;
  // a narrow last column tile wastes matrix work on the 256-wide form: N = 384 is 1.5 tiles (and 804 tiles = 3.14 rounds on 256 CUs)
  const int64_t rem = p.N % 256;
  const bool narrow = p.N < 1024 && rem != 0 && rem <= 128;
  const char* pick = narrow ? (ok128 ? "gemm_bf16_nt_pp128" : nullptr) : (ok256 ? "gemm_bf16_nt_pp" : nullptr);
  if (mode >= 1 || !pick) return pick;
  // Measured policy (tools/pp_check.py, MI355X, old = gemm_bf16_nt_areg / _glds; M = 102 912 / 51 456 / 25 728 tokens):
  //   256-wide: qkv 140 -> 104-112 us, plain K 384 N 1536 164 -> 125-131, GELU' product 186 -> 172-176 (x0.9 at every M); ViT-L qkv 425 -> 291,
  //             fc1 651 -> 480, fc2 525 -> 452 us.  The GELU epilogue at K = 384 is bound by its ~50 VALU cycles per element either way
  //             (fc1 206 vs 203-214 us in isolation; see below for the step);
  //   128-wide: dX K 1152 123 -> 92-96, dX K 1536 160 -> 138, fc2 196 -> 176 at every M; the K = 384 products only on a full chip
  //             (dX of proj 55 -> 45 us at M = 102 912, 28 vs 30 at 51 456) and not with the fp32 residual (proj: 83 vs 93 us).
  // Small problems (less than one round of tiles) keep the 128 x 128 kernels, whose tiles are four times as many.
  const int64_t units = ceil_div(p.M, (int64_t)256) * ceil_div(p.N, (int64_t)(narrow ? 128 : 256));
  if (narrow) {
    if (p.K >= 768) return units >= 192 ? pick : nullptr;
    return (units >= 1024 && !(p.epilogue & DINOX_EPI_RESIDUAL)) ? pick : nullptr;
  }
  // (Until the last day of round 3 the GELU epilogue at K < 768 stayed on the register-prefetch kernel -- a tie in the micro-benchmark, where
  //  the token operand sits in the memory-side cache from the previous iteration.  In the step it does not: behind the LayerNorm-fused
  //  proj the 128 x 128 kernel takes 230 us; on the 256 x 256 tiles the step is 0.25 ms shorter.  DINOX_FC1_AREG=1 restores the old choice.)
  if ((p.epilogue & DINOX_EPI_GELU) && p.K < 768) {
    const char* ea = getenv("DINOX_FC1_AREG");
    if ((ea && atoi(ea) != 0) || units < 1024) return nullptr;   // (bs 64, 606 tiles: 12.46 ms per step on the 128 x 128 kernel, 12.50 here)
  }
  return units >= 192 ? pick : nullptr;
}

const char* gemm_bf16_variant(const GemmParams& p) {
  if (p.in_dtype != DINOX_BF16) return nullptr;
  if (!aligned16(p.A) || !aligned16(p.B) || (p.lda & 7) || (p.ldb & 7) || (p.strideA & 7) || (p.strideB & 7)) return nullptr;
  if (const char* pp = nt_pp_choice(p)) return pp;
  if (gemm_bf16_nt_glds_ok(p)) {
    // Short reductions (K = 384: qkv, proj, fc1, GELU' products of ViT-S) take the form that prefetches the token operand through
    // registers (gemm_bf16_areg.hip): -7 .. -12 % per launch, where the first loads' latency is a large part of a 12-step tile.
    // Measured against this kernel at bs256: K = 768 equal, K = 1152 / 1536 +7 .. +10 % (the extra LDS writes cost more than the
    // deeper prefetch gains once the ring is in steady state), so longer reductions stay here.  DINOX_NT_AREG_MAXK moves the
    // boundary (0 switches the register form off, 1 << 30 sends every K % 192 == 0 product to it) for A/B runs and tests.
    const char* e = getenv("DINOX_NT_AREG_MAXK");
    const int64_t maxk = e ? atoll(e) : 576;
    static const bool no_areg = getenv("DINOX_NT_NO_AREG") != nullptr;
    return !no_areg && p.K <= maxk && gemm_bf16_nt_areg_ok(p) ? "gemm_bf16_nt_areg" : "gemm_bf16_nt_glds";
  }
  if (p.transA == 0 && p.transB == 0 && (p.K & 7) == 0) return "gemm_bf16_nt";
  if (p.transA == 1 && p.transB == 1 && (p.M & 7) == 0 && (p.N & 7) == 0) {
    if (p.ws) {                                     // with a workspace the long-K dW products run on big tiles (gemm_bf16_tnbig.hip)
      int a, b, c;
      int64_t d;
      if (tn_big_plan(p, a, b, c, d)) return "gemm_bf16_tn_big";
    }
    const bool small = p.K * p.lda * 2 < (int64_t)0x7fffffff && p.K * p.ldb * 2 < (int64_t)0x7fffffff && p.M >= 8 && p.N >= 8;
    return small ? "gemm_bf16_tn_dma" : "gemm_bf16_tn";
  }
  return nullptr;
}

// The deterministic reduction needs one contiguous [M][N] fp32 result of one problem, written by the DMA form of the TN kernel.
static bool tn_det_ok(const GemmParams& p, const char* variant, int splits) {
  return variant && !strcmp(variant, "gemm_bf16_tn_dma") && splits > 1 && p.batch == 1 && p.ldc == p.N && p.out_dtype == DINOX_F32 &&
         (p.epilogue & ~DINOX_EPI_ACCUM) == 0;
}

int64_t gemm_bf16_ws_bytes(const GemmParams& p) {
  if (const int64_t big = tn_big_ws_bytes(p)) return big;
  const char* v = gemm_bf16_variant(p);
  if (!v || strncmp(v, "gemm_bf16_tn", 12)) return 0;
  int splits;
  int64_t kps;
  tn_split_plan(p, splits, kps);
  if (!tn_det_ok(p, v, splits)) return 0;
  const int tiles_n = (int)ceil_div(p.N, GB_BN);
  return ((int64_t)splits * p.M * p.N + (p.colsum ? (int64_t)splits * tiles_n * p.M : 0)) * (int64_t)sizeof(float);
}

int launch_gemm_bf16(const GemmParams& p, hipStream_t st) {
  const char* v = gemm_bf16_variant(p);
  if (!v) return DINOX_EUNSUPPORTED;
  if (v[10] == 'n' && v[12] == '_') return v[13] == 'p' ? (v[15] == '1' ? launch_gemm_bf16_nt_pp128(p, st) : v[15] == '3' ? launch_gemm_bf16_nt_pp384(p, st) : launch_gemm_bf16_nt_pp(p, st)) : v[13] == 'a' ? launch_gemm_bf16_nt_areg(p, st) : launch_gemm_bf16_nt_glds(p, st);
  const int tiles_m = (int)ceil_div(p.M, GB_BM), tiles_n = (int)ceil_div(p.N, GB_BN);
  const int64_t ntile = (int64_t)tiles_m * tiles_n;
  if (ntile > 0x7fffffff || p.batch > 65535) return DINOX_EUNSUPPORTED;
  const size_t lds = 4 * GB_TILE_BYTES;
  if (v[10] == 'n') {  // "gemm_bf16_nt"
    dim3 grid((unsigned)ntile, (unsigned)p.batch);
    if (p.out_dtype == DINOX_F32)
      hipLaunchKernelGGL((gemm_bf16_nt<DINOX_F32>), grid, dim3(GB_THREADS), lds, st, p, tiles_m, tiles_n);
    else
      hipLaunchKernelGGL((gemm_bf16_nt<DINOX_BF16>), grid, dim3(GB_THREADS), lds, st, p, tiles_m, tiles_n);
    return check_launch("gemm_bf16_nt");
  }
  if (!strcmp(v, "gemm_bf16_tn_big")) {
    int sp = 1, tn_ = 1;
    if (int rc = launch_gemm_bf16_tn_big(p, st, sp, tn_)) return rc;
    return launch_tn_reduce(p, sp, tn_, st);
  }
  // TN: split K so that the grid has ~2 workgroups per CU; split results meet in the caller's workspace (deterministic two-stage
  // reduction) or, without one, through fp32 atomics.
  int splits;
  int64_t kps;
  tn_split_plan(p, splits, kps);
  if (p.batch > 65535) return DINOX_EUNSUPPORTED;
  GemmParams q = p;
  const bool det = q.ws != nullptr && tn_det_ok(p, v, splits);
  if (!det) q.ws = nullptr;
  if (!det && splits > 1 && !(p.epilogue & DINOX_EPI_ACCUM)) {
    // zero C (rows may be strided by ldc; batch by strideC): contiguous case only, else fall back to 1 split
    if (p.ldc == p.N && (p.batch == 1 || p.strideC == p.M * p.N)) {
      hipError_t e = hipMemsetAsync(p.C, 0, (size_t)p.batch * p.M * p.N * sizeof(float), st);
      if (e != hipSuccess) return fail((int)e, "gemm_bf16_tn: memset: %s", hipGetErrorString(e));
    } else {
      splits = 1;
      kps = ceil_div(p.K, GB_BK) * GB_BK;
    }
  }
  if (!det && p.colsum && !(p.epilogue & DINOX_EPI_ACCUM)) {     // with ACCUM the column sums are added to what colsum holds, like C
    hipError_t e = hipMemsetAsync(p.colsum, 0, (size_t)p.M * sizeof(float), st);
    if (e != hipSuccess) return fail((int)e, "gemm_bf16_tn: memset colsum: %s", hipGetErrorString(e));
  }
  if (ntile * splits > 0x7fffffff) return DINOX_EUNSUPPORTED;
  dim3 grid((unsigned)(ntile * splits), (unsigned)p.batch);
  const char* what = "gemm_bf16_tn";
  if (v[12] == '_') {  // "gemm_bf16_tn_dma"
    what = "gemm_bf16_tn_dma";
    if (p.out_dtype == DINOX_F32)
      hipLaunchKernelGGL((gemm_bf16_tn_dma<DINOX_F32>), grid, dim3(GB_THREADS), lds, st, q, tiles_m, tiles_n, splits, kps);
    else
      hipLaunchKernelGGL((gemm_bf16_tn_dma<DINOX_BF16>), grid, dim3(GB_THREADS), lds, st, q, tiles_m, tiles_n, splits, kps);
  } else if (p.out_dtype == DINOX_F32) {
    hipLaunchKernelGGL((gemm_bf16_tn<DINOX_F32>), grid, dim3(GB_THREADS), lds, st, q, tiles_m, tiles_n, splits, kps);
  } else {
    hipLaunchKernelGGL((gemm_bf16_tn<DINOX_BF16>), grid, dim3(GB_THREADS), lds, st, q, tiles_m, tiles_n, splits, kps);
  }
  if (int rc = check_launch(what)) return rc;
  if (det) return launch_tn_reduce(p, splits, tiles_n, st);
  return 0;
}

}  // namespace dinox
