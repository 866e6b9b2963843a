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
#include <type_traits>

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
  const int o0 = k0 * 128 + (colb ^ (((k0 >> 1) & 1) << 6) ^ (((k0 >> 3) & 1) << 5));
  const int o1 = k1 * 128 + (colb ^ (((k1 >> 1) & 1) << 6) ^ (((k1 >> 3) & 1) << 5));
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tb_lds_s16x4*)(sub + o0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tb_lds_s16x4*)(sub + o1));
  s16x8 v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
  v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bf16x8, v);
}

// Transposed fragment for mfma_16x16x32 from the same sub-image: lane (col = l & 15, g = l >> 4) gets sub[k = 8 g + j][c0 + col], j < 8:
// the whole 32-row step in one fragment.  A 32-lane pass of the read touches rows {q, 8 + q} (then {16 + q, 24 + q}), q < 4, at the same
// columns: the (k >> 3) & 1 term of the row swizzle keeps the two eight-row blocks on different banks.
__device__ __forceinline__ bf16x8 tb_frag16(const char* __restrict__ sub, int c0, int lane) {
  const int i = lane & 15, g = lane >> 4;
  const int q = i >> 2, pp = i & 3;
  const int colb = (c0 + 4 * pp) * 2;
  const int k0 = 8 * g + q, k1 = k0 + 4;
  const int o0 = k0 * 128 + (colb ^ (((k0 >> 1) & 1) << 6) ^ (((k0 >> 3) & 1) << 5));
  const int o1 = k1 * 128 + (colb ^ (((k1 >> 1) & 1) << 6) ^ (((k1 >> 3) & 1) << 5));
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tb_lds_s16x4*)(sub + o0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tb_lds_s16x4*)(sub + o1));
  s16x8 v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
  v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bf16x8, v);
}

// IM x JN accumulator blocks of 32 x 32 per wave (2 x 3: tile 256 x 192; 3 x 2: tile 384 x 128), waves 4 (M) x 2 (N).
template <int IM, int JN, int PP>
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
  // (lane >> 3), position c' = lane & 7); position c' of row k holds the logical 16-byte chunk c' ^ (4 ((k >> 1) & 1)) ^ (2 ((k >> 3) & 1)).  Rows
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
    const int c = cp ^ (((r >> 1) & 1) << 2) ^ (((r >> 3) & 1) << 1);
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

  if constexpr (PP != 0) {
    // ---- K loop, two wave groups in anti-phase (the schedule of gemm_bf16_pp.hip): waves 0-3 and 4-7 are the two waves of every SIMD;
    // in each barrier interval one group reads the fragments of a whole K-step (L slot: 4 (IM + JN) transposing reads, two of its
    // staging requests, the counted wait that retires step kt + 1) while the other issues the step's MFMAs (M slot, the rest of the
    // requests behind them): the matrix pipe of a SIMD always has one wave's MFMAs queued.  Group 1 runs one interval late.
    //   interval 2 kt: G0 L(kt) | G1 M(kt - 1);   interval 2 kt + 1: G0 M(kt) | G1 L(kt)
    // The ring is FIVE steps deep here (140 / 160 KiB): a wave retires step kt + 1 in its L(kt), half a step earlier than the in-step
    // loop would, and the requests need their ~1.5 us.  Stage kt + 4 goes into the slot of stage kt - 1, whose last reader (G1,
    // L(kt - 1)) finished before the barrier in front of interval 2 kt.  Every wave certifies ITS requests of stage kt + 1 in its own
    // L(kt) (a barrier lies between the later of them, interval 2 kt + 1, and the first read, interval 2 kt + 2).
    // The steady state is STRAIGHT-LINE code (one uniform branch, for the bias-gradient MFMAs): which descriptor a request uses, how
    // many requests a wave owns and which wait retires a step are compile-time there -- a dozen taken scalar branches per interval
    // (the first build) cost as much as the interval's MFMAs.  The last four steps run a generic copy.
    // PP == 2: v_mfma_f32_16x16x32_bf16, operands swapped (first = the N-side fragment): lane l holds C[m = l & 15][n = 4 (l >> 4) .. + 3]
    // of a 16 x 16 block; one fragment spans the step's 32 k-rows.  PP == 1: v_mfma_f32_32x32x16_bf16 as in the in-step loop.
    constexpr bool W16 = PP == 2;
    constexpr int NST = TB_STAGES + 1;
    constexpr int TA = SA / 2;                                  // requests t < TA of a wave fetch A sub-images, the rest B (d >> 2 = 2 t + (wv >> 2))
    const int grp = wv >> 2;
    f32x4 acc16[W16 ? 2 * IM : 1][W16 ? 2 * JN : 1], cs16[W16 ? 2 * IM : 1];
    if constexpr (W16) {
#pragma unroll
      for (int i = 0; i < 2 * IM; ++i) {
        cs16[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2 * JN; ++j) acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    unsigned dst[4];                                            // LDS offsets of this wave's requests inside a stage
#pragma unroll
    for (int t = 0; t < PERW; ++t) {
      const int d = wv + 8 * t;
      dst[t] = (unsigned)((d >> 2) * TB_SUB + (d & 3) * 1024);
    }
    auto wait_younger = [&](int n) {                            // wave-uniform n (prologue and the last steps only)
      switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
      }
    };
#define TB_REQ(T, SLOTBASE, KOFFA, KOFFB)                                                                                \
  do {                                                                                                                   \
    if ((T) < TA) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (tb_lds_void*)((SLOTBASE) + dst[T]), 16, voff[T] + (KOFFA), 0, 0, 0); \
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (tb_lds_void*)((SLOTBASE) + dst[T]), 16, voff[T] + (KOFFB), 0, 0, 0); \
  } while (0)
#define TB_SYNC                                                                                                          \
  {                                                                                                                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                   \
    __builtin_amdgcn_s_barrier();                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
  }
    const int npro = nk < NST - 1 ? nk : NST - 1;
    for (int s = 0; s < npro; ++s) TB_STAGE(s, s);
    if (nk > 0) {
      wait_younger((npro - 1) * nmine);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (grp == 1) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    bf16x8 fa[2 * IM], fb[2 * JN];
    auto read_frags = [&](const char* sa) {                     // L slot: the step's fragments
      const char* sb = sa + SA * TB_SUB;
      if constexpr (W16) {
#pragma unroll
        for (int i = 0; i < 2 * IM; ++i) {
          const int col = wr * 32 * IM + i * 16;
          fa[i] = tb_frag16(sa + (col >> 6) * TB_SUB, col & 63, lane);
        }
#pragma unroll
        for (int j = 0; j < 2 * JN; ++j) {
          const int col = wc * 32 * JN + j * 16;
          fb[j] = tb_frag16(sb + (col >> 6) * TB_SUB, col & 63, lane);
        }
      } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
          for (int i = 0; i < IM; ++i) {
            const int col = wr * 32 * IM + i * 32;
            fa[ks * IM + i] = tb_frag(sa + (col >> 6) * TB_SUB, ks * 16, col & 63, lane);
          }
#pragma unroll
          for (int j = 0; j < JN; ++j) {
            const int col = wc * 32 * JN + j * 32;
            fb[ks * JN + j] = tb_frag(sb + (col >> 6) * TB_SUB, ks * 16, col & 63, lane);
          }
        }
      }
    };
    auto mma = [&](bool cs_now) {                               // M slot: the step's MFMAs
      __builtin_amdgcn_s_setprio(1);
      if constexpr (W16) {
#pragma unroll
        for (int i = 0; i < 2 * IM; ++i)
#pragma unroll
          for (int j = 0; j < 2 * JN; ++j) acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc16[i][j], 0, 0, 0);
        if (cs_now) {
#pragma unroll
          for (int i = 0; i < 2 * IM; ++i) cs16[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[i], cs16[i], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
          for (int i = 0; i < IM; ++i)
#pragma unroll
            for (int j = 0; j < JN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks * IM + i], fb[ks * JN + j], acc[i][j], 0, 0, 0);
          if (cs_now) {
#pragma unroll
            for (int i = 0; i < IM; ++i) csacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks * IM + i], ones, csacc[i], 0, 0, 0);
          }
        }
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    };
    int slot = 0;                                               // ring slot of stage kt
    int cs_cnt = tn;                                            // steps until this N tile's next turn at the bias gradient ((kt % tiles_n) == tn)
    unsigned koffA = (unsigned)(NST - 1) * stepA, koffB = (unsigned)(NST - 1) * stepB;     // k offset of stage kt + 4
    const int nsteady = nk - (NST - 1);                         // steps that still request a stage
    auto steady = [&](auto nm_c) {
      constexpr int NM = decltype(nm_c)::value;
      for (int kt = 0; kt < nsteady; ++kt) {
        const char* sa = smem + slot * STAGE;
        char* const rbase = smem + (slot == 0 ? NST - 1 : slot - 1) * STAGE;   // slot of stage kt + 4 (= of stage kt - 1)
        read_frags(sa);
        TB_REQ(0, rbase, koffA, koffB);
        TB_REQ(1, rbase, koffA, koffB);
        // younger than stage kt + 1: stages kt + 2, kt + 3 and the two requests above
        if constexpr (NM == 4) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        TB_SYNC
        const bool cs_now = cs_wave && cs_cnt == 0;             // wave-uniform
        mma(cs_now);
        TB_REQ(2, rbase, koffA, koffB);
        if constexpr (NM == 4) TB_REQ(3, rbase, koffA, koffB);
        TB_SYNC
        slot = slot == NST - 1 ? 0 : slot + 1;
        cs_cnt = cs_cnt == 0 ? tiles_n - 1 : cs_cnt - 1;
        koffA += stepA;
        koffB += stepB;
      }
    };
    if (nsteady > 0) {
      if (NDMA == 8 * PERW || nmine == PERW) steady(std::integral_constant<int, 4>{});
      else steady(std::integral_constant<int, 3>{});
    }
    for (int kt = nsteady > 0 ? nsteady : 0; kt < nk; ++kt) {    // the last steps: nothing left to request
      read_frags(smem + slot * STAGE);
      if (kt + 1 < nk) wait_younger((nk - 2 - kt) * nmine);
      TB_SYNC
      mma(cs_wave && cs_cnt == 0);
      if (!(kt + 1 == nk && grp == 1)) TB_SYNC
      slot = slot == NST - 1 ? 0 : slot + 1;
      cs_cnt = cs_cnt == 0 ? tiles_n - 1 : cs_cnt - 1;
    }
#undef TB_SYNC
#undef TB_REQ
    if constexpr (W16) {
      float* const part16 = (float*)p.ws + (int64_t)split * p.M * p.N;
      if (cs_wave && lane < 16) {        // every n-row of cs16 holds the same sums: lanes 0..15 own the block's 16 columns m
        float* csp = (float*)p.ws + (int64_t)splits * p.M * p.N + ((int64_t)split * tiles_n + tn) * p.M;
#pragma unroll
        for (int i = 0; i < 2 * IM; ++i) {
          const int64_t m = m0 + wr * 32 * IM + i * 16 + lane;
          if (m < p.M) csp[m] = cs16[i][0];
        }
      }
#pragma unroll
      for (int i = 0; i < 2 * IM; ++i) {
        const int64_t m = m0 + wr * 32 * IM + i * 16 + (lane & 15);
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 2 * JN; ++j) {
          const int64_t n = n0 + wc * 32 * JN + j * 16 + 4 * (lane >> 4);
          if (n < p.N) *reinterpret_cast<f32x4*>(part16 + m * p.N + n) = acc16[i][j];      // (N is a multiple of 8)
        }
      }
      return;
    }
  } else {
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
  const char* epp = getenv("DINOX_TN_PP");                    // (read per call: tools flip it between launches)  0 = every wave in step
  const int mode = epp ? atoi(epp) : 2;
  if (mode != 0) {
    const size_t lds = (size_t)(TB_STAGES + 1) * (form == 1 ? 7 : 8) * TB_SUB;     // five stages: 140 / 160 KiB
#define TB_L(IM_, JN_, MODE_)                                                                                            \
  do {                                                                                                                   \
    auto kern = gemm_bf16_tn_big<IM_, JN_, MODE_>;                                                                       \
    if (int rc = reserve_lds(reinterpret_cast<const void*>(kern), lds, "gemm_bf16_tn_big")) return rc;                   \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, p, tiles_m, tiles_n, splits, kps);                          \
  } while (0)
    if (mode == 1) {
      if (form == 1) TB_L(2, 3, 1); else if (form == 2) TB_L(3, 2, 1); else TB_L(2, 4, 1);
    } else {
      if (form == 1) TB_L(2, 3, 2); else if (form == 2) TB_L(3, 2, 2); else TB_L(2, 4, 2);
    }
#undef TB_L
    return check_launch("gemm_bf16_tn_big");
  }
  if (form == 1) {
    constexpr size_t lds = (size_t)TB_STAGES * 7 * TB_SUB;
    if (int rc = reserve_lds(reinterpret_cast<const void*>(gemm_bf16_tn_big<2, 3, 0>), lds, "gemm_bf16_tn_big")) return rc;
    hipLaunchKernelGGL((gemm_bf16_tn_big<2, 3, 0>), dim3(grid), dim3(512), lds, st, p, tiles_m, tiles_n, splits, kps);
  } else if (form == 2) {
    constexpr size_t lds = (size_t)TB_STAGES * 8 * TB_SUB;
    if (int rc = reserve_lds(reinterpret_cast<const void*>(gemm_bf16_tn_big<3, 2, 0>), lds, "gemm_bf16_tn_big")) return rc;
    hipLaunchKernelGGL((gemm_bf16_tn_big<3, 2, 0>), dim3(grid), dim3(512), lds, st, p, tiles_m, tiles_n, splits, kps);
  } else {
    constexpr size_t lds = (size_t)TB_STAGES * 8 * TB_SUB;
    if (int rc = reserve_lds(reinterpret_cast<const void*>(gemm_bf16_tn_big<2, 4, 0>), lds, "gemm_bf16_tn_big")) return rc;
    hipLaunchKernelGGL((gemm_bf16_tn_big<2, 4, 0>), dim3(grid), dim3(512), lds, st, p, tiles_m, tiles_n, splits, kps);
  }
  return check_launch("gemm_bf16_tn_big");
}

}  // namespace dinox
