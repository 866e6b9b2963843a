// gemm_bf16_pp.hip -- round 3: persistent "ping-pong" NT bf16 MFMA GEMM on 256 x 256 x 64 tiles, ONE workgroup of eight waves per CU.
//
// Why.  The 128 x 128 kernels (gemm_bf16_glds.hip, gemm_bf16_areg.hip) keep the matrix pipe 29-32 % busy:
// a K-step is 8 MFMAs per wave between two barriers, every wave of a workgroup waits for LDS-DMA, fragment reads and the barrier at the
// same moment, and the prologue / epilogue of a 6-24-step tile is as long as its K loop.  This kernel is built the other way round:
//   * eight waves = 2 (rows) x 4 (columns), each 128 x 64 of the tile: 8 x 4 accumulators of v_mfma_f32_16x16x32_bf16 (128 VGPRs); the
//     operands are SWAPPED (D^T = W . X^T) so that a lane ends up with four consecutive output COLUMNS of one row;
//   * the two row groups (waves 0-3 / 4-7: the two waves of every SIMD) run ONE barrier interval apart: while a wave issues the 16 MFMAs of
//     a quadrant (M slot) its SIMD partner reads its next fragments from LDS and issues its share of the LDS-DMA (L slot), then they swap.
//     A K-tile is four quadrants (m-half x n-half, 64 x 32 x 64 each); the matrix pipe of a SIMD sees back-to-back MFMA clusters from
//     alternating waves and no wave ever waits for its own fragment reads with the pipe idle;
//   * operands arrive by global_load_lds_dwordx4 into two 64 KiB K-tile buffers ([row][64 k], 16-byte chunk c of row r stored at
//     c ^ (r & 7): conflict-free ds_read_b128 for the 16x16x32 operand map); a K-tile is cut into four 16 KiB pieces in the order the
//     quadrants need them (A rows of m-half 0, B rows of n-half 0, B rows of n-half 1, A rows of m-half 1), piece j of K-tile g + 1 is
//     requested in L slot j of K-tile g and retired by a counted s_waitcnt vmcnt(4) one slot before its first reader's slot, with a
//     barrier in between (the LDS-DMA visibility rule of cdna_hip_programming.md section 5);
//   * PERSISTENT: a workgroup walks its share of the tiles; the request stream runs one K-tile ahead ACROSS tile boundaries, so the
//     first K-tile of the next tile lands under the epilogue of this one.  gfx950 has one in-order counter for loads and stores, so
//     every DMA of the next K-tile is retired BEFORE the epilogue's first store (vmcnt(0): they were requested 1-4 slots earlier) and
//     the first K-tile after an epilogue skips the two waits that would otherwise stand behind the output stores;
//   * epilogue: 16-row slices of the wave's block pass through a private 4 KiB LDS tile (ds_write_b128 in accumulator layout,
//     ds_read_b128 by rows, chunks XOR row) so that memory sees whole 128-byte (bf16) / 256-byte (fp32) row segments; bias, GELU with
//     its derivative side tensor, x GELU', fp32 residual as in the other NT kernels; bf16 outputs leave by non-temporal stores.
// Envelope: K % 64 == 0, K >= 128, N % 8 == 0, no batch, 16-byte aligned operands, leading dimensions < 2^21 (32-bit byte offsets
// inside a tile).  Replaces nn.Linear forward / dX products (reference zoo/arch.py:46,53,75-76).
#include <cstdlib>

#include "common.h"
#include "gemm_common.h"
#include "gemm_pp_common.h"

namespace dinox {

constexpr int PP_BM = 256, PP_BN = 256, PP_BK = 64;
constexpr int PP_A_BYTES = PP_BM * PP_BK * 2;                 // 32 KiB
constexpr int PP_KT_BYTES = (PP_BM + PP_BN) * PP_BK * 2;      // 64 KiB per K-tile buffer
constexpr int PP_STAGE0 = 2 * PP_KT_BYTES;                    // per-wave epilogue staging tiles behind the two buffers
constexpr int PP_LDS = PP_STAGE0 + 8 * 4096;                  // 160 KiB: the whole LDS of a CU

template <int OUT_DT, int ACT, bool RES>
__global__ __launch_bounds__(512, 2) void gemm_bf16_nt_pp(GemmParams p, int tiles_n, int units, int order, int stagger) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wv >> 2, wc = wv & 3;
  const int nk = (int)(p.K / PP_BK);

  const int w = (int)blockIdx.x;
  int u_first, u_step, my, my_max;
  pp_my_tiles(order, units, u_first, u_step, my, my_max);
  if (my <= 0) return;                                         // (workgroup-uniform)
  const int total = my * nk;

  // ---- fragment reads.  Operand map of v_mfma_f32_16x16x32_bf16: lane l holds row (l & 15), k = 8 (l >> 4) .. + 7 of a 16 x 32 block.
  const int fr = lane & 15, fq = lane >> 4;
  unsigned foff[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) foff[ks] = (unsigned)(fr * 128 + (((ks * 4 + fq) ^ (lane & 7)) << 4));
  const unsigned a_rd = (unsigned)(grp * 16384), b_rd = (unsigned)(PP_A_BYTES + wc * 8192);

  // ---- the request stream: pieces a (A rows of m-half 0 of both row groups), b / c (B rows of n-half 0 / 1 of the four column strips),
  // d (A rows of m-half 1).  A piece is 128 rows = 16 instructions of 8 rows x 128 B; wave wv issues instructions 2 wv and 2 wv + 1.
  // LDS slot (row, s) holds logical chunk s ^ (row & 7); (row & 7) == lane >> 3 for every instruction.
  unsigned dst_a[2], dst_b[2], dst_c[2], dst_d[2];            // LDS byte offsets inside a K-tile buffer (wave-uniform)
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int i0 = (wv * 2 + q) * 8;
    const int ra0 = i0 < 64 ? i0 : i0 + 64, rb0 = (i0 >> 5) * 64 + (i0 & 31);
    dst_a[q] = (unsigned)(ra0 * 128);
    dst_d[q] = (unsigned)((ra0 + 64) * 128);
    dst_b[q] = (unsigned)(PP_A_BYTES + rb0 * 128);
    dst_c[q] = (unsigned)(PP_A_BYTES + (rb0 + 32) * 128);
  }
  const unsigned src_chunk = (unsigned)((((lane & 7) ^ (lane >> 3)) & 7) << 4);
  // Tile bookkeeping is kept out of the K loop (a division and 64-bit address arithmetic in one L slot held all eight waves at the next
  // barrier for ~950 cycles per tile): the scalars of the tile the stream enters NEXT are computed inside an epilogue (`nx_*`), and the
  // per-lane offsets are recomputed at a crossing only when an edge tile is involved.
  struct TileAt { const char* a; const char* b; int mlast, nlast; };
  auto tile_at = [&](int t) {
    const int u = u_first + t * u_step, tm = u / tiles_n, tn = u - tm * tiles_n;
    const int64_t m0 = (int64_t)tm * PP_BM, n0 = (int64_t)tn * PP_BN;
    TileAt r;
    r.a = (const char*)((const bf16_t*)p.A + m0 * p.lda);
    r.b = (const char*)((const bf16_t*)p.B + n0 * p.ldb);
    r.mlast = (int)(p.M - m0 < PP_BM ? p.M - m0 : PP_BM) - 1;
    r.nlast = (int)(p.N - n0 < PP_BN ? p.N - n0 : PP_BN) - 1;
    return r;
  };
  const char* ia = nullptr;                                    // uniform: A + (tile row origin) * lda + k offset, bytes
  const char* ib = nullptr;
  unsigned oa[2], ob[2], oc[2], od[2];
  bool cur_full = false;
  int iu = 0, ik = 0;                                          // tile (index into this workgroup's list) and K-tile of the next request
  auto enter_tile = [&](const TileAt& ta) {
    ia = ta.a;
    ib = ta.b;
    const bool full = ta.mlast == PP_BM - 1 && ta.nlast == PP_BN - 1;
    if (!(full && cur_full)) {
      const int lr = (int)(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) >> 3);   // lane >> 3, re-derived here: not kept live across the K loop
#pragma unroll
      for (int q = 0; q < 2; ++q) {                             // rows past the edge re-read the last valid row (never stored)
        const int i = (wv * 2 + q) * 8 + lr;                    // this lane's row of the piece (0..127)
        const int row_a = i < 64 ? i : i + 64, row_d = row_a + 64, row_b = (i >> 5) * 64 + (i & 31), row_c = row_b + 32;
        oa[q] = (unsigned)(row_a < ta.mlast ? row_a : ta.mlast) * (unsigned)p.lda * 2u + src_chunk;
        od[q] = (unsigned)(row_d < ta.mlast ? row_d : ta.mlast) * (unsigned)p.lda * 2u + src_chunk;
        ob[q] = (unsigned)(row_b < ta.nlast ? row_b : ta.nlast) * (unsigned)p.ldb * 2u + src_chunk;
        oc[q] = (unsigned)(row_c < ta.nlast ? row_c : ta.nlast) * (unsigned)p.ldb * 2u + src_chunk;
      }
    }
    cur_full = full;
  };
  TileAt nx = tile_at(my > 1 ? 1 : 0);
  auto issue = [&](const char* base, const unsigned (&off)[2], const unsigned (&dst)[2], int buf) {
    char* l = smem + buf * PP_KT_BYTES;
#pragma unroll
    for (int q = 0; q < 2; ++q)
      __builtin_amdgcn_global_load_lds((pp_gbl_void*)(base + off[q]), (pp_lds_void*)(l + dst[q]), 16, 0, 0);
  };
  auto advance = [&]() {                                        // after piece d: the stream moves to the next K-tile / tile
    ia += PP_BK * 2;
    ib += PP_BK * 2;
    if (++ik == nk) {
      ik = 0;
      if (++iu < my) enter_tile(nx);
    }
  };
  auto issue_ktile = [&](int buf) {
    issue(ia, oa, dst_a, buf);
    issue(ib, ob, dst_b, buf);
    issue(ib, oc, dst_c, buf);
    issue(ia, od, dst_d, buf);
    advance();
  };

  pp_f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = pp_f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 af[4][2], bfr[2][2];

  auto read_a = [&](int buf, int mh) {
    const char* s = smem + buf * PP_KT_BYTES + a_rd + mh * 8192;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) af[i][ks] = *reinterpret_cast<const bf16x8*>(s + i * 2048 + foff[ks]);
  };
  auto read_b = [&](int buf, int nh) {
    const char* s = smem + buf * PP_KT_BYTES + b_rd + nh * 4096;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) bfr[j][ks] = *reinterpret_cast<const bf16x8*>(s + j * 2048 + foff[ks]);
  };
#define PP_MMA(MH, NH)                                                                                                   \
  {                                                                                                                       \
    __builtin_amdgcn_s_setprio(1);                                                                                        \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                      \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                         \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                         \
      acc[(MH) * 4 + i][(NH) * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j][ks], af[i][ks], acc[(MH) * 4 + i][(NH) * 2 + j], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                                        \
  }
  // slot boundary: my LDS reads have returned (they are the next M slot's operands, and the buffer may be refilled behind the barrier)
#define PP_SYNC                                                                                                           \
  {                                                                                                                       \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                    \
    __builtin_amdgcn_s_barrier();                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
  }
  // Counted waits.  gfx950 retires loads, LDS-DMA and stores of a wave in ONE issue-ordered counter, so "piece X has landed" is
  // vmcnt(number of operations issued after X), and that number includes the output stores of the previous tile's epilogue whenever X
  // was requested before them.  `sq` says how many stores that epilogue certainly issued (a full tile: 16 or 32; an edge tile or the
  // first tile: 0 -- a LOWER bound is always safe, it only waits for more than necessary).
#define PP_VMCNT(BASE)                                                                                                    \
  {                                                                                                                       \
    if (sq == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BASE) : "memory");                                              \
    else if (sq == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((BASE) + 16) : "memory");                                  \
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((BASE) + 32) : "memory");                                               \
  }

  char* const stage = smem + PP_STAGE0 + wv * 4096;
  const bool two_out = ACT == PP_GELU && p.aux != nullptr;
  // stores one full tile's epilogue issues per wave: 16 row passes x (1 | 2 sixteen-byte stores per output) x outputs; capped at 32
  const int sq_full = (OUT_DT == DINOX_BF16 && !two_out) ? 1 : 2;
  // diagnostic cycle stamps (tools/pp_stamps.py hands a buffer over in p.ws, which NT products never use otherwise):
  // waves 0 and 4 of every workgroup write s_memtime at the marked points, 256 slots per wave
  long long* const dbg = (p.ws && (wv & 3) == 0) ? (long long*)p.ws + ((int64_t)w * 2 + grp) * 256 : nullptr;
  int dbgi = 0;
#define PP_STAMP if (dbg && dbgi < 256 && lane == 0) dbg[dbgi] = (long long)__builtin_amdgcn_s_memtime(); ++dbgi;

  // ---- de-phasing.  Every workgroup has the same work per tile, so without it all 256 run in lockstep: every CU is in its K loop (no
  // stores at all) or in its epilogue (stores only) at the same time, and the output stream meets HBM in bursts of twice its bandwidth
  // (measured: the plain bf16 epilogue took 5.4k cycles, GELU + side tensor 28k, against ~2k / ~12k of instructions).  Workgroups that own
  // one tile less than the busiest ones have a whole tile period to spare: they start late by a pseudo-random share of `stagger` cycles.
  if (stagger > 0 && (my < my_max || (order & 512))) {          // (order bit 9: diagnostic -- every workgroup starts late)
    const unsigned h = ((unsigned)w * 2654435761u) >> 16;      // 16 bits
    const int naps = (int)(((int64_t)stagger * h) >> 26);       // stagger * h / 65536 cycles, in naps of 1024
    for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(16);
  }

  // ---- prologue: K-tiles 0 and 1 of the first tile; the first one is retired before anybody reads
  enter_tile(tile_at(0));
  issue_ktile(0);
  issue_ktile(1);                                               // (nk >= 2)
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  int g = 0;                                                    // K-tiles done (over all tiles): buffer = g & 1
  int sq = 0;                                                   // stores of the previous epilogue standing in the counter: 0 / 16 / 32
  for (int t = 0; t < my; ++t) {
    PP_STAMP
    if (grp == 1) {                                             // the second row group runs one barrier interval behind the first
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    for (int kt = 0; kt < nk; ++kt, ++g) {
      const int buf = g & 1, nbuf = buf ^ 1;
      // K-tile g + 1 is requested in this K-tile's L slots -- except in the first K-tile of a tile: that request went out ahead of
      // the previous epilogue's stores (or in the prologue)
      const bool req = kt > 0 && g + 1 < total, last = kt + 1 == nk;
      PP_STAMP
      // L0: piece c of this K-tile must have landed
      read_a(buf, 0);
      read_b(buf, 0);
      if (req) issue(ia, oa, dst_a, nbuf);
      if (kt == 0) PP_VMCNT(10)                                 // younger: d, the whole next K-tile, the stores
      else if (!req) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (kt == 1) PP_VMCNT(4)                             // younger: d, the stores, a of the next
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      PP_SYNC
      PP_MMA(0, 0)
      PP_SYNC
      // L1: piece d
      read_b(buf, 1);
      if (req) issue(ib, ob, dst_b, nbuf);
      if (kt == 0) PP_VMCNT(8)
      else if (!req) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (kt == 1) PP_VMCNT(4)
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      PP_SYNC
      PP_MMA(0, 1)
      PP_SYNC
      // L2
      read_a(buf, 1);
      if (req) issue(ib, oc, dst_c, nbuf);
      PP_SYNC
      PP_MMA(1, 1)
      PP_SYNC
      // L3: pieces a, b of the next K-tile
      read_b(buf, 0);
      if (req) {
        issue(ia, od, dst_d, nbuf);
        advance();
      }
      if (kt == 0) PP_VMCNT(4)                                  // younger: c, d of the next K-tile and the stores
      else if (!req) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // (kt == 1: the stores are older than a, b -- they are waited for here)
      PP_SYNC
      PP_MMA(1, 0)
      if (!(last && grp == 1)) PP_SYNC
    }

    // ---- epilogue of tile t
    PP_STAMP
    {
      const int u = u_first + t * u_step, tm = u / tiles_n, tn = u - tm * tiles_n;
      const int64_t m0t = (int64_t)tm * PP_BM, n0t = (int64_t)tn * PP_BN, mw = m0t + grp * 128, nw = n0t + wc * 64;     // this wave's block
      sq = (p.M - mw >= 128 && nw + 64 <= p.N) ? sq_full : 0;
      pp_epilogue<OUT_DT, ACT, RES, 8>(p, acc, stage, mw, nw, m0t, n0t, lane, (order & 256) != 0, [&]() {
        // The second K-tile of the next tile goes out NOW, ahead of this tile's stores: the waits of the next tile's first K-tile then
        // count the stores instead of standing behind them (the buffer of this tile's last K-tile is free: every wave is past its reads).
        if (t + 2 < my) nx = tile_at(t + 2);                    // (the stream may cross into it below when nk == 2)
        if (g + 1 < total) issue_ktile((g + 1) & 1);
        PP_STAMP
      });
    }
    PP_STAMP
  }
#undef PP_STAMP
#undef PP_MMA
#undef PP_SYNC
#undef PP_VMCNT
}

// Shapes and epilogues this kernel takes (the caller has checked in_dtype == bf16, transA == transB == 0).
bool gemm_bf16_nt_pp_ok(const GemmParams& p) { return pp_envelope_ok(p, PP_BK); }

int launch_gemm_bf16_nt_pp(const GemmParams& p, hipStream_t st) {
  const int64_t tiles_m = ceil_div(p.M, (int64_t)PP_BM), tiles_n = ceil_div(p.N, (int64_t)PP_BN);
  const int64_t units = tiles_m * tiles_n;
  if (units > 0x3fffffff) return DINOX_EUNSUPPORTED;
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return fail(DINOX_EINVAL, "gemm_bf16_nt_pp: no device");
    ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const char* eo = getenv("DINOX_PP_ORDER");
  const int order = eo ? atoi(eo) : 1;
  // start delay of the workgroups that own one tile less than the busiest ones (up to about one tile period, cycles): pays where the
  // epilogue is long (GELU' product 184 -> 172 us, fc2 180 -> 176), costs where it is short (qkv 104 vs 112 us): off for plain / bias
  const char* es = getenv("DINOX_PP_STAGGER");
  const bool heavy = (p.epilogue & (DINOX_EPI_GELU | DINOX_EPI_DGELU | DINOX_EPI_RESIDUAL)) != 0;
  const int stagger = es ? atoi(es) : heavy ? (int)(p.K / PP_BK) * 2600 + ((p.epilogue & (DINOX_EPI_GELU | DINOX_EPI_DGELU)) ? 12000 : 4000) : 0;
  const unsigned grid = (unsigned)(units < ncu ? units : ncu);
  const int act = (p.epilogue & DINOX_EPI_GELU) ? PP_GELU : (p.epilogue & DINOX_EPI_DGELU) ? PP_DGELU : PP_PLAIN;
  const bool res = (p.epilogue & DINOX_EPI_RESIDUAL) != 0;
#define PP_L(OUT, ACT, RES)                                                                                               \
  do {                                                                                                                    \
    auto kern = gemm_bf16_nt_pp<OUT, ACT, RES>;                                                                           \
    if (int rc = reserve_lds(reinterpret_cast<const void*>(kern), PP_LDS, "gemm_bf16_nt_pp")) return rc;                  \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), PP_LDS, st, p, (int)tiles_n, (int)units, order, stagger);                    \
  } while (0)
#define PP_A(OUT)                                                                                                         \
  switch (act * 2 + (res ? 1 : 0)) {                                                                                      \
    case 0: PP_L(OUT, PP_PLAIN, false); break;                                                                            \
    case 1: PP_L(OUT, PP_PLAIN, true); break;                                                                             \
    case 2: PP_L(OUT, PP_GELU, false); break;                                                                             \
    case 4: PP_L(OUT, PP_DGELU, false); break;                                                                            \
    default: return DINOX_EUNSUPPORTED;                                                                                   \
  }
  if (p.out_dtype == DINOX_BF16) { PP_A(DINOX_BF16) } else { PP_A(DINOX_F32) }
#undef PP_A
#undef PP_L
  return check_launch("gemm_bf16_nt_pp");
}

}  // namespace dinox
