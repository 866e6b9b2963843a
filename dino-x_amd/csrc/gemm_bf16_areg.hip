// gemm_bf16_areg.hip -- NT bf16 MFMA GEMM with the TOKEN operand prefetched through the register file, for every reduction length
// that is a multiple of 192 (K = 384, 768, 1152, 1536, 2304, 3072: all products of ViT-S and ViT-B, most of ViT-L's) and every
// epilogue of the LDS-DMA kernel (bias, GELU with its side tensor, GELU' from the side tensor, fp32 residual).
//
// Why ("What bounds the LDS-DMA feed"): a CU sustains about bytes-in-flight / latency from memory; the 3-slot
// LDS ring of gemm_bf16_glds.hip keeps two K-steps (2 x 16 KiB per workgroup) in flight, and the A operand (activations, streamed
// from HBM at 2-2.5 us under load) is what the K-loop waits for.  Here a workgroup keeps SIX K-steps of A in flight in six register
// sets (48 VGPRs per thread, 48 KiB per workgroup): A(k) is requested five steps before it is needed, written to its LDS slot
// (ds_write_b128, the same swizzled image the MFMA fragment reads expect) two steps before it is used, and its register set is
// re-requested for A(k + 6) in the following step.  The weight operand B (L2-resident, short latency) keeps the LDS-DMA ring.
//
// All global loads of A are issued by inline asm and retired by counted s_waitcnt vmcnt placed from the static issue order: hipcc,
// left to track them itself, drains the DMA ring (vmcnt(0)) at every use of a loaded register.  The K loop is unrolled by six
// steps (lcm of 3 ring slots and 6 register sets, so slot and set of every step are compile-time); the first and the last block of
// six are separate copies because their issue order differs.  Per thread / wave, step j = 6 b + r issues, in this order,
//     B(j+2)  [2 DMA]   then   A(j+7) -> set (r+1) % 6  [2 loads]      (step 0: A6 and A7; nothing that does not exist)
// and then writes A(j+2) from set (r+2) % 6 to slot (r+2) % 3 and multiplies slot r % 3.  The wait in front of step j's barrier must
// retire B(j) and A(j+2); memory reads return in order, so it is "vmcnt(number of loads issued after B(j))":
//     first block   -  -  8  6  6  6          (B0 B1 A0..A5 are requested before the loop and retired with vmcnt(0))
//     middle block  6  6  6  6  6  6
//     last block    4  2  2  2  3  P          (step NK-3 also issues the bias slice, 1 DMA, BEFORE B(NK-1); step NK-2 issues the P
//                                              loads of the epilogue's first operand rows: residual 8, GELU' input 4 or 8, else 0)
// A(j+2) is always older than an operation one of these waits retires (A(j+7) is issued right after B(j+2), i.e. before B(j+3),
// which the wait of step j+3 retires; it is written at step j+5).
//
// Lessons kept from the experiments behind this file ("NT kernel log"):
//   * the barrier in front of the accumulator parking needs an explicit s_waitcnt lgkmcnt(0): the compiler leaves the last step's
//     LDS reads in flight across it (their MFMAs follow the barrier) and another wave's park writes then overtake them -- one tile
//     in ~1e5 multiplied by parked fp32 words;
//   * vmcnt counts an LDS-DMA done a moment before its bytes are readable: every wait is followed by a barrier before the read;
//   * per-lane addresses are an SGPR base + a 32-bit VGPR offset (half the address registers; at 168 VGPRs a spilled register
//     set is stored before its load has landed);
//   * a persistent form (workgroups walking tiles, next tile's first loads issued under the current epilogue) was built and measured
//     5 % SLOWER: gfx950 has one in-order counter for loads and stores, so a tile's first counted wait also waits for the previous
//     tile's stores to be acknowledged.
#include "common.h"
#include "gemm_common.h"
#include <type_traits>

namespace dinox {

typedef __attribute__((address_space(3))) void ar_lds_void;
typedef __attribute__((address_space(1))) const void ar_gbl_void;
typedef unsigned ar_u32x4 __attribute__((ext_vector_type(4)));      // (a native vector: HIP's uint4 is a struct, which inline asm cannot tie)

typedef float ar_f32x4 __attribute__((ext_vector_type(4)));

// Output stores are NON-TEMPORAL (global_store ... nt): results are 79-316 MB tensors that nobody re-reads before they have left the
// 32 MB of L2, while the operand slices of the tiles still running are re-read from L2 3 to 12 times.  With ordinary stores the
// output stream evicts them: tools/tile_probe.hip, K loop + output skeleton, qkv 156.8 -> 129.6 us, fc1 200.1 -> 153.5 us with `nt`;
// this kernel: fc1 268 -> 218 us, teacher fc1 224 -> 197, GELU' product 222 -> 200, qkv 147 -> 143.
// (a compile-time choice: written as a run-time branch the two stores are merged into one plain store and the hint is lost)
template <bool NT, typename V>
__device__ __forceinline__ void ar_store(char* dst, V v) {
  if (NT) __builtin_nontemporal_store(v, reinterpret_cast<V*>(dst));
  else *reinterpret_cast<V*>(dst) = v;
}

constexpr int AR_BK = 32, AR_BM = 128, AR_BN = 128;
constexpr int AR_SETS = 6;                                             // register sets of the A prefetch (K-step mod 6)
constexpr int AR_ATILE = AR_BM * AR_BK * 2, AR_BTILE = AR_BN * AR_BK * 2, AR_SLOT = AR_ATILE + AR_BTILE;
constexpr int AR_KBLOCK = 6 * AR_BK * 2;                               // bytes of K one block of six steps advances a row by

enum { AR_PLAIN = 0, AR_GELU = 1, AR_DGELU = 2 };

__device__ __forceinline__ int ar_xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int OUT_DT, int ACT, bool RES, bool NT>
__global__ __launch_bounds__(256, 3) void gemm_bf16_nt_areg(GemmParams p, int ntiles, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wv >> 1, wc = wv & 1;
  const int tile = ar_xcd_remap(blockIdx.x, ntiles);
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int64_t m0 = (int64_t)tm * AR_BM, n0 = (int64_t)tn * AR_BN;
  constexpr int ESZ = OUT_DT == DINOX_BF16 ? 2 : 4;
  const float* biasp = (p.epilogue & DINOX_EPI_BIAS) ? p.bias : (const float*)p.B;    // (always a readable address: the DMA is unconditional)
  char* const biasl = smem + 3 * AR_SLOT + wv * 256;                       // this wave's 64 bias values

  // ---- addresses: a workgroup-uniform 64-bit base (SGPRs, advanced by one K block per loop iteration) + a 32-bit per-lane byte offset
  const char* abase = (const char*)((const bf16_t*)p.A + m0 * p.lda);
  const char* bbase = (const char*)((const bf16_t*)p.B + n0 * p.ldb);
  unsigned avoff[2], bvoff[2];
  {
    const int mrem = (int)(p.M - m0 < AR_BM ? p.M - m0 : AR_BM) - 1, nrem = (int)(p.N - n0 < AR_BN ? p.N - n0 : AR_BN) - 1;   // last valid row
#pragma unroll
    for (int q = 0; q < 2; ++q) {   // B by LDS-DMA: 8 instructions per slot (16 rows x 64 B each), two per wave; slot (row, c') gets chunk c' ^ ((row>>2)&3)
      const int row = (wv * 2 + q) * 16 + (lane >> 2);
      const int c = (lane & 3) ^ ((row >> 2) & 3);
      bvoff[q] = (unsigned)(((int64_t)(row < nrem ? row : nrem) * p.ldb + c * 8) * 2);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {   // A through registers: per K-step two 16-B pieces per thread (piece id = t + 256 u: row = id / 4, chunk = id % 4)
      const int row = ((int)threadIdx.x + 256 * u) >> 2, c = threadIdx.x & 3;
      avoff[u] = (unsigned)(((int64_t)(row < mrem ? row : mrem) * p.lda + c * 8) * 2);
    }
  }
  // K-step `s` of the current block (s may run past 5: the next block's first steps) into ring slot s % 3.  The K offset rides in the
  // instruction's immediate, which the hardware adds to the global address AND to the LDS address: the LDS pointer is pre-compensated.
  auto stage_b = [&](auto sc) {
    constexpr int s = decltype(sc)::value;
    char* sb = smem + (s % 3) * AR_SLOT + AR_ATILE + wv * 2048 - s * (AR_BK * 2);
#pragma unroll
    for (int q = 0; q < 2; ++q) __builtin_amdgcn_global_load_lds((ar_gbl_void*)(bbase + bvoff[q]), (ar_lds_void*)(sb + q * 1024), 16, s * (AR_BK * 2), 0);
  };
#define AR_IC(N) std::integral_constant<int, (N)>{}
  ar_u32x4 areg[AR_SETS][2];
  // (the immediate offset must be a literal: one statement per K-step; S is relative to the current block)
#define AR_LOAD_A(S, SET)                                                                                                  \
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(areg[SET][0]) : "v"(avoff[0]), "s"(abase), "n"((S) * 64)); \
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(areg[SET][1]) : "v"(avoff[1]), "s"(abase), "n"((S) * 64));
#define AR_PIN(SET) asm volatile("" : "+v"(areg[SET][0]), "+v"(areg[SET][1]));       /* "this set has landed": a plain value from here on */
  // piece u of a thread is row (t >> 2) + 64 u: same chunk swizzle ((row >> 2) & 3 is unchanged by + 64), 4096 B further on
  const unsigned adst = (unsigned)(((int)threadIdx.x >> 2) * 64 + (((threadIdx.x & 3) ^ ((threadIdx.x >> 4) & 3)) << 4));
  auto write_a = [&](int slot, int set) {                       // register set -> ring slot (A half)
    char* sa = smem + slot * AR_SLOT + adst;
#pragma unroll
    for (int u = 0; u < 2; ++u) *reinterpret_cast<ar_u32x4*>(sa + u * 4096) = areg[set][u];
  };

  // ---- prologue: everything the first steps need is requested at once
  stage_b(AR_IC(0));
  stage_b(AR_IC(1));
  AR_LOAD_A(0, 0) AR_LOAD_A(1, 1) AR_LOAD_A(2, 2) AR_LOAD_A(3, 3) AR_LOAD_A(4, 4) AR_LOAD_A(5, 5)

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int frow = lane & 31, fh = lane >> 5;
  auto compute = [&](int slot) {
    const char* sa = smem + slot * AR_SLOT;
    const char* sb = sa + AR_ATILE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[2], bfr[2];
      const int kc = 2 * ks + fh;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ra = wr * 64 + i * 32 + frow;
        af[i] = *reinterpret_cast<const bf16x8*>(sa + ra * 64 + ((kc ^ ((ra >> 2) & 3)) << 4));
        const int rb = wc * 64 + i * 32 + frow;
        bfr[i] = *reinterpret_cast<const bf16x8*>(sb + rb * 64 + ((kc ^ ((rb >> 2) & 3)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  };

  // ---- epilogue operands (residual rows, GELU' input rows): a lane owns 8 consecutive columns of 4 rows per pass
  const int c8 = lane & 7;
  const bool n_ok = n0 + wc * 64 + c8 * 8 < p.N;                            // N % 8 == 0: a lane's 8 columns are all in or all out
  const int64_t mw = m0 + wr * 64, nw = n0 + wc * 64;                       // this wave's 64 x 64 block origin (uniform)
  const int mleft = (int)(p.M - mw < 64 ? p.M - mw : 64);                   // valid rows of the block (may be <= 0)
  char* const cblk = (char*)p.C + (mw * p.ldc + nw) * ESZ;
  char* const ablk = (char*)p.aux + (mw * p.ldaux + nw) * ESZ;
  // operand LOADS of a wave whose rows are all past M (mleft <= 0) go to the tile's first rows instead: valid addresses, never used
  const int64_t mwl = mleft > 0 ? mw : m0;
  const int mleft_l = (int)(p.M - mwl < 64 ? p.M - mwl : 64);              // >= 1
  const char* const ablk_l = (const char*)p.aux + (mwl * p.ldaux + nw) * ESZ;
  const char* const rblk = (const char*)(p.residual + (mwl * p.ldr + nw));
  constexpr bool PF_AUX = ACT == AR_DGELU;
  constexpr int AUXV = OUT_DT == DINOX_BF16 ? 1 : 2;                        // 16-B vectors of GELU' input per row (bf16: 8 values in one)
  float4 pf_aux[PF_AUX ? 4 : 1][AUXV];
  float4 pf_res[RES ? 4 : 1][2];
  auto prefetch = [&](int ps) {                                             // rows past M are loaded (clamped address) and never stored
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      int mr = ps * 32 + it * 8 + (lane >> 3);
      mr = mr < mleft_l ? mr : mleft_l - 1;
      const unsigned col = n_ok ? c8 * 8 : 0;
      if (PF_AUX) {
        const unsigned ai = (unsigned)((mr * (int)p.ldaux + col) * ESZ);
#pragma unroll
        for (int h = 0; h < AUXV; ++h) {
          if (NT) {                                              // read once: streamed past L2 like the outputs
            const ar_f32x4 xv = __builtin_nontemporal_load(reinterpret_cast<const ar_f32x4*>(ablk_l + ai + 16 * h));
            pf_aux[it][h] = make_float4(xv[0], xv[1], xv[2], xv[3]);
          } else {
            pf_aux[it][h] = *reinterpret_cast<const float4*>(ablk_l + ai + 16 * h);
          }
        }
      }
      if (RES) {
        const unsigned ri = (unsigned)((mr * (int)p.ldr + col) * 4);
        pf_res[it][0] = *reinterpret_cast<const float4*>(rblk + ri);
        pf_res[it][1] = *reinterpret_cast<const float4*>(rblk + ri + 16);
      }
    }
  };
  constexpr int PF_OPS = (RES ? 8 : 0) + (PF_AUX ? 4 * AUXV : 0);           // loads one prefetch() issues

  // retire the whole first burst (B0 B1 A0..A5), put A0 and A1 in place
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  AR_PIN(0) AR_PIN(1)
  write_a(0, 0);
  write_a(1, 1);

  // ---- K loop: blocks of six steps.  R = step within the block, WAITN = loads issued after B(j) (-1: nothing to wait for),
  // NEXT_B / NEXT_A = does B(j+2) / A(j+2) exist, PRE = issued before B(j+2), LOADS = the A requests of this step.
#define AR_STEP(R, WAITN, NEXT_B, NEXT_A, PRE, LOADS)                                                                     \
  {                                                                                                                       \
    if (WAITN >= 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN < 0 ? 0 : WAITN) : "memory");                          \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                    \
    __builtin_amdgcn_s_barrier();                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    PRE                                                                                                                   \
    if (NEXT_B) stage_b(AR_IC((R) + 2));                                                                                  \
    LOADS                                                                                                                 \
    if (NEXT_A) {                                                                                                         \
      AR_PIN(((R) + 2) % AR_SETS)                                                                                         \
      write_a(((R) + 2) % 3, ((R) + 2) % AR_SETS);                                                                        \
    }                                                                                                                     \
    compute((R) % 3);                                                                                                     \
  }
  const int nblk = (int)(p.K / (6 * AR_BK));                               // >= 2
  // first block: A6 A7 at step 0 (sets 0 1 were written out above), then one request per step
  AR_STEP(0, -1, true, true, , AR_LOAD_A(6, 0) AR_LOAD_A(7, 1))
  AR_STEP(1, -1, true, true, , AR_LOAD_A(8, 2))
  AR_STEP(2, 8, true, true, , AR_LOAD_A(9, 3))
  AR_STEP(3, 6, true, true, , AR_LOAD_A(10, 4))
  AR_STEP(4, 6, true, true, , AR_LOAD_A(11, 5))
  AR_STEP(5, 6, true, true, , if (nblk > 2) { AR_LOAD_A(12, 0) })
  abase += AR_KBLOCK;
  bbase += AR_KBLOCK;
  for (int b = 1; b + 1 < nblk; ++b) {
    AR_STEP(0, 6, true, true, , AR_LOAD_A(7, 1))
    AR_STEP(1, 6, true, true, , AR_LOAD_A(8, 2))
    AR_STEP(2, 6, true, true, , AR_LOAD_A(9, 3))
    AR_STEP(3, 6, true, true, , AR_LOAD_A(10, 4))
    AR_STEP(4, 6, true, true, , AR_LOAD_A(11, 5))
    AR_STEP(5, 6, true, true, , if (b + 2 < nblk) { AR_LOAD_A(12, 0) })
    abase += AR_KBLOCK;
    bbase += AR_KBLOCK;
  }
  // last block: nothing left to request for A; the bias slice goes out BEFORE B(NK-1), so that the wait of the last step retires
  // it too and two barriers lie between that wait and the epilogue's read
#define AR_BIAS_DMA                                                                                                       \
  {                                                                                                                       \
    int64_t bn = nw + lane;                                                                                               \
    bn = bn < p.N ? bn : p.N - 1;                                                                                         \
    __builtin_amdgcn_global_load_lds((ar_gbl_void*)(biasp + bn), (ar_lds_void*)biasl, 4, 0, 0);                           \
  }
  AR_STEP(0, 4, true, true, , )
  AR_STEP(1, 2, true, true, , )
  AR_STEP(2, 2, true, true, , )
  AR_STEP(3, 2, true, true, AR_BIAS_DMA, )
  AR_STEP(4, 3, false, false, , if (PF_OPS) prefetch(0);)
  AR_STEP(5, PF_OPS, false, false, , )
#undef AR_STEP
#undef AR_LOAD_A
#undef AR_PIN
#undef AR_IC
#undef AR_BIAS_DMA
  // The LDS reads of the last step must have RETURNED before this wave reports in (see the file header).
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();     // all waves done with the ring before it becomes the park area

  // ---- epilogue (as gemm_bf16_glds.hip): each wave parks one 32-row half of its 64x64 block at a time (row = 256 B, 16-B chunks
  // XOR (row & 15)) and re-reads it by rows: 16-B vector math and stores.
  char* const park = smem + wv * (32 * 256);
  const float alpha = p.alpha;
  float bias[8];
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * fh;
        const int nn = j * 32 + frow;
        *reinterpret_cast<float*>(park + row * 256 + ((((nn >> 2) ^ (row & 15))) << 4) + (nn & 3) * 4) = acc[ps][j][e];
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (ps == 0) {
      const bool hb = (p.epilogue & DINOX_EPI_BIAS) != 0;
      const float4 b0 = *reinterpret_cast<const float4*>(biasl + c8 * 32), b1 = *reinterpret_cast<const float4*>(biasl + c8 * 32 + 16);
      bias[0] = hb ? b0.x : 0.f; bias[1] = hb ? b0.y : 0.f; bias[2] = hb ? b0.z : 0.f; bias[3] = hb ? b0.w : 0.f;
      bias[4] = hb ? b1.x : 0.f; bias[5] = hb ? b1.y : 0.f; bias[6] = hb ? b1.z : 0.f; bias[7] = hb ? b1.w : 0.f;
    }
    float4 cur_aux[PF_AUX ? 4 : 1][AUXV];
    float4 cur_res[RES ? 4 : 1][2];
    if (PF_OPS) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        if (PF_AUX) {
#pragma unroll
          for (int h = 0; h < AUXV; ++h) cur_aux[it][h] = pf_aux[it][h];
        }
        if (RES) {
          cur_res[it][0] = pf_res[it][0];
          cur_res[it][1] = pf_res[it][1];
        }
      }
      if (ps == 0) prefetch(1);
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      __builtin_amdgcn_sched_barrier(0);                                   // one row group at a time: interleaving them costs registers
      const int row = it * 8 + (lane >> 3);
      const int mr = ps * 32 + row;
      const float4 lo = *reinterpret_cast<const float4*>(park + row * 256 + (((2 * c8) ^ (row & 15)) << 4));
      const float4 hi = *reinterpret_cast<const float4*>(park + row * 256 + (((2 * c8 + 1) ^ (row & 15)) << 4));
      if (mr >= mleft || !n_ok) continue;
      // Four columns at a time, packed as soon as they are final: eight GELUs in flight at once (the compiler's choice when left
      // alone) cost ~50 registers.
      const bool ag = (p.epilogue & DINOX_EPI_AUXGRAD) != 0;            // workgroup-uniform
      const unsigned ci = (unsigned)((mr * (int)p.ldc + c8 * 8) * ESZ), ai = (unsigned)((mr * (int)p.ldaux + c8 * 8) * ESZ);
      unsigned pv[4], pa[4];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        __builtin_amdgcn_sched_barrier(0);
        const float4 x4 = h ? hi : lo;
        float v[4] = {x4.x, x4.y, x4.z, x4.w}, a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = v[u] * alpha + bias[4 * h + u];
        if (ACT == AR_GELU) {
          if (ag) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              float y, d;
              gelu_fast_both(v[u], y, d);
              a[u] = d;
              v[u] = y;
            }
          } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              a[u] = v[u];
              v[u] = gelu_fast(v[u]);
            }
          }
          if (OUT_DT == DINOX_BF16) {
            pa[2 * h] = (unsigned)f32_to_bf16(a[0]) | ((unsigned)f32_to_bf16(a[1]) << 16);
            pa[2 * h + 1] = (unsigned)f32_to_bf16(a[2]) | ((unsigned)f32_to_bf16(a[3]) << 16);
          } else if (p.aux) {
            ar_store<NT>(ablk + ai + 16 * h, ar_f32x4{a[0], a[1], a[2], a[3]});
          }
        }
        if (ACT == AR_DGELU) {
          float x[4];
          if (OUT_DT == DINOX_BF16) {
            const float4 raw = cur_aux[it][0];
            const unsigned w0 = __float_as_uint(h ? raw.z : raw.x), w1 = __float_as_uint(h ? raw.w : raw.y);
            x[0] = __uint_as_float(w0 << 16); x[1] = __uint_as_float(w0 & 0xffff0000u);
            x[2] = __uint_as_float(w1 << 16); x[3] = __uint_as_float(w1 & 0xffff0000u);
          } else {
            const float4 xv = cur_aux[it][h ? AUXV - 1 : 0];
            x[0] = xv.x; x[1] = xv.y; x[2] = xv.z; x[3] = xv.w;
          }
          if (ag) {
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] *= x[u];
          } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] *= gelu_fast_grad(x[u]);
          }
        }
        if (RES) {
          const float4 r = cur_res[it][h];
          v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
        }
        if (OUT_DT == DINOX_BF16) {
          pv[2 * h] = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
          pv[2 * h + 1] = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
        } else {
          ar_store<NT>(cblk + ci + 16 * h, ar_f32x4{v[0], v[1], v[2], v[3]});
        }
      }
      if (OUT_DT == DINOX_BF16) {
        if (ACT == AR_GELU && p.aux) ar_store<NT>(ablk + ai, ar_u32x4{pa[0], pa[1], pa[2], pa[3]});
        ar_store<NT>(cblk + ci, ar_u32x4{pv[0], pv[1], pv[2], pv[3]});
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// Envelope inside gemm_bf16_nt_glds's (its alignment rules are checked by the caller first): K a multiple of 192 and >= 384, one
// problem (no batch), leading dimensions small enough for 32-bit byte offsets inside a tile.
bool gemm_bf16_nt_areg_ok(const GemmParams& p) {
  const int64_t ldmax = 1 << 22;
  if (p.K < 2 * 6 * AR_BK || p.K % (6 * AR_BK) || p.batch != 1 || p.M < 1) return false;
  if (p.lda >= ldmax || p.ldb >= ldmax || p.ldc >= ldmax) return false;
  if ((p.epilogue & (DINOX_EPI_GELU | DINOX_EPI_DGELU)) && p.ldaux >= ldmax) return false;
  if ((p.epilogue & DINOX_EPI_DGELU) && !p.aux) return false;
  if ((p.epilogue & DINOX_EPI_GELU) && (p.epilogue & DINOX_EPI_DGELU)) return false;
  if ((p.epilogue & DINOX_EPI_RESIDUAL) && p.ldr >= ldmax) return false;
  // (residual together with GELU / GELU' does not occur on the path and does not fit the register budget beside the operand prefetch)
  if ((p.epilogue & DINOX_EPI_RESIDUAL) && (p.epilogue & (DINOX_EPI_GELU | DINOX_EPI_DGELU))) return false;
  return true;
}

template <int OUT_DT, int ACT, bool RES>
static void ar_launch(const GemmParams& p, unsigned ntile, int tiles_n, size_t lds, hipStream_t st) {
  // non-temporal output stores (see ar_store).  Same box, back to back: fc1 + side tensor 268 -> 218 us, teacher fc1 224 -> 197,
  // GELU' product 222 -> 200, qkv 147 -> 143, proj (fp32 residual stream, 3 column tiles) 93 -> 93.  DINOX_NT_STORES=0 restores
  // ordinary stores for A/B runs.
  // Only for bf16 outputs: the fp32 outputs are the residual stream, which the following LayerNorm reads straight back (neutral on
  // proj here, +9 % on the K = 1536 fc2 product when tried in gemm_bf16_nt_glds, whose bf16 products gain 1-2 %: not adopted there).
  static const int knob = getenv("DINOX_NT_STORES") ? atoi(getenv("DINOX_NT_STORES")) : -1;      // read once per process
  int nt = knob >= 0 ? knob : (OUT_DT == DINOX_BF16 ? 1 : 0);
  // (A/B values: 2 = only outputs of 10 or more column tiles, 3 = only narrower ones.  Whole step on one box: off 44.40 ms, 3: 44.05,
  //  2: 43.34, all bf16 outputs: 43.45.  LayerNorm outputs are the opposite case: stored non-temporally the step LOSES 1.1 ms, the
  //  79 MB they write are still in the last-level cache when the next GEMM reads them.)
  if (nt == 2) nt = OUT_DT == DINOX_BF16 && tiles_n >= 10;
  if (nt == 3) nt = OUT_DT == DINOX_BF16 && tiles_n < 10;
  if (nt) hipLaunchKernelGGL((gemm_bf16_nt_areg<OUT_DT, ACT, RES, true>), dim3(ntile), dim3(256), lds, st, p, (int)ntile, tiles_n);
  else hipLaunchKernelGGL((gemm_bf16_nt_areg<OUT_DT, ACT, RES, false>), dim3(ntile), dim3(256), lds, st, p, (int)ntile, tiles_n);
}

int launch_gemm_bf16_nt_areg(const GemmParams& p, hipStream_t st) {
  const int tiles_m = (int)ceil_div(p.M, (int64_t)AR_BM), tiles_n = (int)ceil_div(p.N, (int64_t)AR_BN);
  const int64_t ntile64 = (int64_t)tiles_m * tiles_n;
  if (ntile64 > 0x7fffffff) return DINOX_EUNSUPPORTED;
  const unsigned ntile = (unsigned)ntile64;
  const size_t lds = 3 * (size_t)AR_SLOT + 4 * 256;
  const int act = (p.epilogue & DINOX_EPI_GELU) ? AR_GELU : (p.epilogue & DINOX_EPI_DGELU) ? AR_DGELU : AR_PLAIN;
  const bool res = (p.epilogue & DINOX_EPI_RESIDUAL) != 0;
#define AR_L(OUT)                                                                                                         \
  switch (act * 2 + (res ? 1 : 0)) {                                                                                      \
    case 0: ar_launch<OUT, AR_PLAIN, false>(p, ntile, tiles_n, lds, st); break;                                           \
    case 1: ar_launch<OUT, AR_PLAIN, true>(p, ntile, tiles_n, lds, st); break;                                            \
    case 2: ar_launch<OUT, AR_GELU, false>(p, ntile, tiles_n, lds, st); break;                                            \
    case 4: ar_launch<OUT, AR_DGELU, false>(p, ntile, tiles_n, lds, st); break;                                           \
    default: return DINOX_EUNSUPPORTED;                                                                                   \
  }
  if (p.out_dtype == DINOX_BF16) { AR_L(DINOX_BF16) } else { AR_L(DINOX_F32) }
#undef AR_L
  return check_launch("gemm_bf16_nt_areg");
}

}  // namespace dinox
