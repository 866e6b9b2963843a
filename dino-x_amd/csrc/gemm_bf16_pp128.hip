// gemm_bf16_pp128.hip -- the persistent ping-pong NT kernel (gemm_bf16_pp.hip) on 256 x 128 x 64 tiles, for the products whose N is a
// multiple of 128 but not of 256: every N = 384 product of ViT-S (proj, fc2, the three dX products: 11 ms of a 40 ms step).  On 256-wide
// tiles N = 384 is 1.5 column tiles (a quarter of the matrix work wasted) and 804 tiles = 3.14 rounds over 256 CUs; on 128-wide tiles it is
// 3 x 402 = 1206 tiles = 4.71 rounds (94 % full).
//
// Same structure as the 256 x 256 kernel, re-cut for the narrower tile:
//   * eight waves = 4 (rows) x 2 (columns), each 64 x 64: 4 x 4 accumulators of v_mfma_f32_16x16x32_bf16 (64 VGPRs), operands swapped so
//     that a lane holds four consecutive output columns;
//   * waves 0-3 (rows 0-127) and 4-7 (rows 128-255) are the two waves of every SIMD and run one barrier interval apart: one issues the 16
//     MFMAs of a k-half (M slot) while its partner reads fragments and issues LDS-DMA (L slot).  A K-tile is TWO slots per wave, cut
//     along K (k 0..31, then k 32..63 of the whole 64 x 64 block: 4 A + 4 B fragment reads each -- cut along the rows the first slot
//     carried 12 of the 16 reads and the matrix pipe waited for it: 1700 cycles per K-tile instead of ~1250);
//   * THREE 48 KiB K-tile buffers ([256 + 128 rows][64 k], chunk c of row r at c ^ (r & 7)): K-tile g + 2 is requested during K-tile g
//     (three instructions per wave and slot) and ONE counted wait per K-tile retires K-tile g + 1 before the barrier in front of its
//     first read.  The stores of the previous epilogue are counted in that wait for the first K-tile of a tile (gfx950: one in-order
//     counter for loads and stores), so it does not stand behind them;
//   * epilogue staging in the K-tile buffer the tile's last K-tile has just freed (nothing is requested into it before the next tile's
//     first L slot, and a workgroup barrier separates the two), 4 KiB per wave, then the shared fused epilogue (gemm_pp_common.h).
// Envelope as gemm_bf16_pp.hip.  Replaces nn.Linear forward / dX products (reference zoo/arch.py:53,76 and their backward).
#include <cstdlib>

#include "common.h"
#include "gemm_common.h"
#include "gemm_pp_common.h"

namespace dinox {

constexpr int PQ_BM = 256, PQ_BN = 128, PQ_BK = 64;
constexpr int PQ_A_BYTES = PQ_BM * PQ_BK * 2;                 // 32 KiB
constexpr int PQ_KT_BYTES = (PQ_BM + PQ_BN) * PQ_BK * 2;      // 48 KiB per K-tile buffer
constexpr int PQ_LDS = 3 * PQ_KT_BYTES;                       // 144 KiB

template <int OUT_DT, int ACT, bool RES>
__global__ __launch_bounds__(512, 2) void gemm_bf16_nt_pp128(GemmParams p, int tiles_n, int units, int order, int stagger) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wv >> 2, wm = wv >> 1, wn = wv & 1;          // row block wm (64 rows), column block wn (64 columns)
  const int nk = (int)(p.K / PQ_BK);

  const int w = (int)blockIdx.x;
  int u_first, u_step, my, my_max;
  pp_my_tiles(order, units, u_first, u_step, my, my_max);
  if (my <= 0) return;                                         // (workgroup-uniform)
  const int total = my * nk;

  // ---- fragment reads (operand map of v_mfma_f32_16x16x32_bf16: lane l holds row l & 15, k = 8 (l >> 4) .. + 7 of a 16 x 32 block)
  const int fr = lane & 15, fq = lane >> 4;
  unsigned foff[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) foff[ks] = (unsigned)(fr * 128 + (((ks * 4 + fq) ^ (lane & 7)) << 4));
  const unsigned a_rd = (unsigned)(wm * 8192), b_rd = (unsigned)(PQ_A_BYTES + wn * 8192);

  // ---- the request stream.  A K-tile is 48 instructions of 8 rows x 128 B; wave wv issues six: in the first L slot rows
  // {wm' * 64 + 0..31} of A for its share (piece a) and half of its B share, in the second the rest (the whole K-tile is retired at once).  Instruction j of a
  // 128-row piece covers rows 8 j .. 8 j + 7 of the piece; piece a = A rows (i / 32) * 64 + i % 32, piece d = the same + 32, piece b = B.
  unsigned dst_a[2], dst_b[2], dst_d[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int i0 = (wv * 2 + q) * 8;
    const int ra0 = (i0 >> 5) * 64 + (i0 & 31);
    dst_a[q] = (unsigned)(ra0 * 128);
    dst_d[q] = (unsigned)((ra0 + 32) * 128);
    dst_b[q] = (unsigned)(PQ_A_BYTES + i0 * 128);
  }
  const unsigned src_chunk = (unsigned)((((lane & 7) ^ (lane >> 3)) & 7) << 4);
  struct TileAt { const char* a; const char* b; int mlast, nlast; };
  auto tile_at = [&](int t) {
    const int u = u_first + t * u_step, tm = u / tiles_n, tn = u - tm * tiles_n;
    const int64_t m0 = (int64_t)tm * PQ_BM, n0 = (int64_t)tn * PQ_BN;
    TileAt r;
    r.a = (const char*)((const bf16_t*)p.A + m0 * p.lda);
    r.b = (const char*)((const bf16_t*)p.B + n0 * p.ldb);
    r.mlast = (int)(p.M - m0 < PQ_BM ? p.M - m0 : PQ_BM) - 1;
    r.nlast = (int)(p.N - n0 < PQ_BN ? p.N - n0 : PQ_BN) - 1;
    return r;
  };
  const char* ia = nullptr;
  const char* ib = nullptr;
  unsigned oa[2], ob[2], od[2];
  bool cur_full = false;
  int iu = 0, ik = 0;
  auto enter_tile = [&](const TileAt& ta) {
    ia = ta.a;
    ib = ta.b;
    const bool full = ta.mlast == PQ_BM - 1 && ta.nlast == PQ_BN - 1;
    if (!(full && cur_full)) {
      const int lr = (int)(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) >> 3);   // lane >> 3 (not kept live across the K loop)
#pragma unroll
      for (int q = 0; q < 2; ++q) {                             // rows past the edge re-read the last valid row (never stored)
        const int i = (wv * 2 + q) * 8 + lr;
        const int row_a = (i >> 5) * 64 + (i & 31), row_d = row_a + 32;
        oa[q] = (unsigned)(row_a < ta.mlast ? row_a : ta.mlast) * (unsigned)p.lda * 2u + src_chunk;
        od[q] = (unsigned)(row_d < ta.mlast ? row_d : ta.mlast) * (unsigned)p.lda * 2u + src_chunk;
        ob[q] = (unsigned)(i < ta.nlast ? i : ta.nlast) * (unsigned)p.ldb * 2u + src_chunk;
      }
    }
    cur_full = full;
  };
  TileAt nx = tile_at(my > 1 ? 1 : 0);
  auto advance = [&]() {
    ia += PQ_BK * 2;
    ib += PQ_BK * 2;
    if (++ik == nk) {
      ik = 0;
      if (++iu < my) enter_tile(nx);
    }
  };
  // A K-tile's six requests per wave.  An LDS-DMA instruction costs the issuing wave ~100 cycles inside an L slot (the K loop of this
  // tile shape is bound by that, not by the matrix pipe: 48 KiB staged per 4.2 MFLOP), but only ~50 behind the MFMAs of an M slot, whose
  // pipe keeps running: two requests go out in each L slot and one at the end of each M slot.
  auto dma = [&](const char* base, unsigned off, unsigned dst, int buf) {
    __builtin_amdgcn_global_load_lds((pp_gbl_void*)(base + off), (pp_lds_void*)(smem + buf * PQ_KT_BYTES + dst), 16, 0, 0);
  };
  pp_f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = pp_f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 af[4], bfr[4];
  auto read_ab = [&](const char* kt, int ks) {                  // the fragments of k-half ks: four row blocks of A, four column blocks of B
    const char* sa = kt + a_rd + foff[ks];
    const char* sb = kt + b_rd + foff[ks];
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sa + i * 2048);
#pragma unroll
    for (int j = 0; j < 4; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(sb + j * 2048);
  };
#define PQ_MMA                                                                                                            \
  {                                                                                                                       \
    __builtin_amdgcn_s_setprio(1);                                                                                        \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                         \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                         \
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);                             \
    __builtin_amdgcn_s_setprio(0);                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
  }
#define PQ_SYNC                                                                                                           \
  {                                                                                                                       \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                    \
    __builtin_amdgcn_s_barrier();                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
  }
  // "K-tile g + 1 has landed" = vmcnt(operations issued after it): the five requests of K-tile g + 2 issued so far (the sixth follows in
  // the M slot behind the wait) and, for the first K-tile after an epilogue, that epilogue's stores (sq = 0 / 8 / 16 of them for certain:
  // a LOWER bound is always safe).
#define PQ_VMCNT(BASE)                                                                                                    \
  {                                                                                                                       \
    if (sq == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BASE) : "memory");                                              \
    else if (sq == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((BASE) + 8) : "memory");                                   \
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((BASE) + 16) : "memory");                                               \
  }
  const bool two_out = ACT == PP_GELU && p.aux != nullptr;
  const int sq_full = (OUT_DT == DINOX_BF16 && !two_out) ? 1 : 2;     // 8 row passes x (1 | 2) sixteen-byte stores x outputs, capped at 16
  long long* const dbg = (p.ws && (wv & 3) == 0) ? (long long*)p.ws + ((int64_t)w * 2 + grp) * 256 : nullptr;
  int dbgi = 0;
#define PQ_STAMP if (dbg && dbgi < 256 && lane == 0) dbg[dbgi] = (long long)__builtin_amdgcn_s_memtime(); ++dbgi;

  // de-phasing of the workgroups that own one tile less than the busiest ones (see gemm_bf16_pp.hip)
  if (stagger > 0 && (my < my_max || (order & 512))) {
    const unsigned h = ((unsigned)w * 2654435761u) >> 16;
    const int naps = (int)(((int64_t)stagger * h) >> 26);
    for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(16);
  }

  // ---- prologue: K-tiles 0 and 1; the first one is retired before anybody reads
  enter_tile(tile_at(0));
#pragma unroll
  for (int b = 0; b < 2; ++b) {                                 // (nk >= 3)
    dma(ia, oa[0], dst_a[0], b);
    dma(ia, oa[1], dst_a[1], b);
    dma(ib, ob[0], dst_b[0], b);
    dma(ib, ob[1], dst_b[1], b);
    dma(ia, od[0], dst_d[0], b);
    dma(ia, od[1], dst_d[1], b);
    advance();
  }
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  int g = 0, buf = 0;                                           // K-tiles done (over all tiles); ring slot of K-tile g
  int sq = 0;
  for (int t = 0; t < my; ++t) {
    PQ_STAMP
    if (grp == 1) {                                             // the second row group runs one barrier interval behind the first
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    for (int kt = 0; kt < nk; ++kt, ++g) {
      const char* const cur = smem + buf * PQ_KT_BYTES;
      const int rbuf = buf == 0 ? 2 : buf - 1;                  // ring slot of K-tile g + 2 (= of K-tile g - 1: every wave is past its reads)
      const bool req = g + 2 < total, last = kt + 1 == nk;
      PQ_STAMP
      // L0
      read_ab(cur, 0);
      if (req) {
        dma(ia, oa[0], dst_a[0], rbuf);
        dma(ia, oa[1], dst_a[1], rbuf);
      }
      PQ_SYNC
      PQ_MMA
      if (req) dma(ib, ob[0], dst_b[0], rbuf);
      PQ_SYNC
      // L1: K-tile g + 1 must have landed before the barrier in front of its first read
      read_ab(cur, 1);
      if (req) {
        dma(ib, ob[1], dst_b[1], rbuf);
        dma(ia, od[0], dst_d[0], rbuf);
      }
      if (g + 1 < total) {
        if (!req) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (kt == 0) PQ_VMCNT(5)                           // younger: five requests of K-tile g + 2 and the previous epilogue's stores
        else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      }
      PQ_SYNC
      PQ_MMA
      if (req) {
        dma(ia, od[1], dst_d[1], rbuf);
        advance();
      }
      if (!(last && grp == 1)) PQ_SYNC
      buf = buf == 2 ? 0 : buf + 1;
    }

    // ---- epilogue of tile t: staging in the ring slot of the tile's last K-tile (free until the next tile's first L slot requests
    // into it, which lies behind the barrier below)
    PQ_STAMP
    {
      const int u = u_first + t * u_step, tm = u / tiles_n, tn = u - tm * tiles_n;
      const int64_t m0t = (int64_t)tm * PQ_BM, n0t = (int64_t)tn * PQ_BN, mw = m0t + wm * 64, nw = n0t + wn * 64;
      char* const stage = smem + (buf == 0 ? 2 : buf - 1) * PQ_KT_BYTES + wv * 4096;
      sq = (p.M - mw >= 64 && nw + 64 <= p.N) ? sq_full : 0;
      pp_epilogue<OUT_DT, ACT, RES, 4>(p, acc, stage, mw, nw, m0t, n0t, lane, (order & 256) != 0, [&]() {
        if (t + 2 < my) nx = tile_at(t + 2);                    // (the stream crosses into it during the next tile)
        PQ_STAMP
      });
    }
    PQ_STAMP
    // every wave is done with its staging tile before the next request may land there
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
#undef PQ_STAMP
#undef PQ_MMA
#undef PQ_SYNC
#undef PQ_VMCNT
}

bool gemm_bf16_nt_pp128_ok(const GemmParams& p) { return pp_envelope_ok(p, PQ_BK) && p.K >= 3 * PQ_BK; }

int launch_gemm_bf16_nt_pp128(const GemmParams& p, hipStream_t st) {
  const int64_t tiles_m = ceil_div(p.M, (int64_t)PQ_BM), tiles_n = ceil_div(p.N, (int64_t)PQ_BN);
  const int64_t units = tiles_m * tiles_n;
  if (units > 0x3fffffff) return DINOX_EUNSUPPORTED;
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return fail(DINOX_EINVAL, "gemm_bf16_nt_pp128: no device");
    ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const char* eo = getenv("DINOX_PP_ORDER");
  const int order = eo ? atoi(eo) : 1;
  // start delay of the workgroups that own one tile less than the busiest ones (up to about one tile period, cycles): pays where the
  // epilogue is long (GELU' product 184 -> 172 us, fc2 180 -> 176), costs where it is short (qkv 104 vs 112 us): off for plain / bias
  const char* es = getenv("DINOX_PP_STAGGER");
  const bool heavy = (p.epilogue & (DINOX_EPI_GELU | DINOX_EPI_DGELU | DINOX_EPI_RESIDUAL)) != 0;
  const int stagger = es ? atoi(es) : heavy ? (int)(p.K / PQ_BK) * 1300 + 3000 : 0;
  const unsigned grid = (unsigned)(units < ncu ? units : ncu);
  const int act = (p.epilogue & DINOX_EPI_GELU) ? PP_GELU : (p.epilogue & DINOX_EPI_DGELU) ? PP_DGELU : PP_PLAIN;
  const bool res = (p.epilogue & DINOX_EPI_RESIDUAL) != 0;
#define PQ_L(OUT, ACT, RES)                                                                                               \
  do {                                                                                                                    \
    auto kern = gemm_bf16_nt_pp128<OUT, ACT, RES>;                                                                        \
    if (int rc = reserve_lds(reinterpret_cast<const void*>(kern), PQ_LDS, "gemm_bf16_nt_pp128")) return rc;               \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), PQ_LDS, st, p, (int)tiles_n, (int)units, order, stagger);             \
  } while (0)
#define PQ_A(OUT)                                                                                                         \
  switch (act * 2 + (res ? 1 : 0)) {                                                                                      \
    case 0: PQ_L(OUT, PP_PLAIN, false); break;                                                                            \
    case 1: PQ_L(OUT, PP_PLAIN, true); break;                                                                             \
    case 2: PQ_L(OUT, PP_GELU, false); break;                                                                             \
    case 4: PQ_L(OUT, PP_DGELU, false); break;                                                                            \
    default: return DINOX_EUNSUPPORTED;                                                                                   \
  }
  if (p.out_dtype == DINOX_BF16) { PQ_A(DINOX_BF16) } else { PQ_A(DINOX_F32) }
#undef PQ_A
#undef PQ_L
  return check_launch("gemm_bf16_nt_pp128");
}

}  // namespace dinox
