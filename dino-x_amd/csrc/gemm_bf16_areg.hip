// gemm_bf16_areg.hip -- NT bf16 MFMA GEMM for K = 384 (the model width of ViT-S: qkv, fc1 and every product whose reduction
// runs over D) with the TOKEN operand prefetched through the register file.
//
// Why ("What bounds the LDS-DMA feed"): a CU sustains about bytes-in-flight / latency from memory; the 3-slot
// LDS ring of gemm_bf16_glds.hip keeps two K-steps (2 x 16 KiB per workgroup) in flight, and the A operand (activations, streamed
// from HBM at 2-2.5 us under load) is what the K-loop waits for: loading A only once per tile (an experiment with wrong results)
// cut 19-31 % off these products.  Here a workgroup asks for 8 of its 12 A K-steps at once, before anything else -- 64 KiB per
// workgroup in flight from the first cycle, held in 64 VGPRs per thread -- and for the remaining 4 as soon as registers free up;
// each K-step's A slice is written to its LDS slot (ds_write_b128, same swizzled image the MFMA fragment reads expect) two steps
// before it is used.  The weight operand B (L2-resident, short latency) keeps the LDS-DMA ring.
//
// All global loads of A are issued by inline asm and retired by counted s_waitcnt vmcnt placed from the static issue order (the
// K-loop is fully unrolled: 12 steps): hipcc, left to track them itself, drains the DMA ring (vmcnt(0)) at every use of a loaded
// register.
//
// The kernel is PERSISTENT: a workgroup walks tiles t = w, w + nb, ... and asks for the NEXT tile's first burst (its bias slice, B0,
// A0..A7 at step 10, B1 at step 11) while the current tile still has two K-steps and its whole epilogue to run, so the 2-2.5 us the
// burst takes are covered by work instead of being waited for at every tile start.  Issue order of vector-memory operations per
// thread / wave (x' = next tile; "bias" = the current tile's bias slice by LDS-DMA, one instruction):
//   first tile  B0 B1 (2 DMA each) | A0 .. A5 (2 loads each)                   -> vmcnt(0)
//   tile top    A0 A1 written to LDS (after a barrier: their slots were park areas)
//   step 0    B2 | A6 A7   step 1  B3 | A8   step 2  B4 | A9   step 3  B5 | A10   step 4  B6 | A11   step s (5..8)  B(s+2)
//   step 9    bias | B11     step 10   B0' | A0' .. A3'        step 11  B1'        epilogue  stores of pass 0 | A4' A5' | stores of pass 1
//   (six register sets, K-step mod 6, and the next tile's burst split in two: eight sets, or six beside 64 live accumulators and
//   the epilogue's temporaries, do not fit 168 registers -- and a spilled set is stored before its load has landed)
//   wait before step kt's barrier = number of LOADS issued after B(kt):  kt 1: 10 (A4 A5 of this tile, B2 A6 A7), kt 2: 8
//   (retires A4 A5, issued ~2400 cycles earlier), kt 3, 4, 5: 6, kt 6: 4, kt 7..9: 2, kt 10: 3, kt 11: 10 (0 without a next tile);
//   before the epilogue's first use of bias / first store: 2 (only B1' younger than A0'..A3'; 0 without a next tile).
//   Step kt writes A(kt+2) to LDS: A2 A3 were retired in the previous epilogue, A4 A5 are older than B2 (wait of step 2), A6 A7
//   older than B3, A8 older than B4, A9 older than B5, A10 older than B6, A11 older than B7.
// Stores are never counted on: every wait is placed so that stores still in flight can only make it stricter (they are older than
// the operation it retires), never weaker.
// While the next burst lands in the A halves' registers, the epilogue parks accumulators in the LDS that is free at that point:
// slot 2 (waves 0, 1) and the A halves of slots 0 and 1 (waves 2, 3) -- the B halves of slots 0 / 1 are receiving B0' / B1'.
#include "common.h"
#include "gemm_common.h"
#include <type_traits>

namespace dinox {

typedef __attribute__((address_space(3))) void ar_lds_void;
typedef __attribute__((address_space(1))) const void ar_gbl_void;
typedef unsigned ar_u32x4 __attribute__((ext_vector_type(4)));      // (a native vector: HIP's uint4 is a struct, which inline asm cannot tie)

constexpr int AR_K = 384, AR_BK = 32, AR_NK = AR_K / AR_BK, AR_BM = 128, AR_BN = 128;
constexpr int AR_SETS = 6;                                             // register sets of the A prefetch (K-step mod 6)
constexpr int AR_ATILE = AR_BM * AR_BK * 2, AR_BTILE = AR_BN * AR_BK * 2, AR_SLOT = AR_ATILE + AR_BTILE;

enum { AR_PLAIN = 0, AR_GELU = 1 };

__device__ __forceinline__ int ar_xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int OUT_DT, int ACT>
__global__ __launch_bounds__(256, 3) void gemm_bf16_nt_areg(GemmParams p, int ntiles, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wv >> 1, wc = wv & 1;
  const int nb = gridDim.x;                                               // workers; worker ids are XCD-contiguous
  int tile = ar_xcd_remap(blockIdx.x, nb);
  const bf16_t* A = (const bf16_t*)p.A;
  const bf16_t* B = (const bf16_t*)p.B;
  const float* biasp = (p.epilogue & DINOX_EPI_BIAS) ? p.bias : (const float*)p.B;    // (always a readable address: the DMA is unconditional)
  char* const biasl = smem + 3 * AR_SLOT + wv * 256;                       // this wave's 64 bias values

  // ---- B by LDS-DMA: 8 instructions per slot (16 rows x 64 B each), two per wave; slot (row, c') gets chunk c = c' ^ ((row>>2)&3)
  // ---- A through registers: per K-step two 16-B pieces per thread (piece id = t + 256 u: row = id / 4, chunk = id % 4)
  // Addresses are a workgroup-uniform 64-bit base (SGPRs) plus a 32-bit per-lane byte offset: two VGPRs per operand, not four.
  const char* abase;
  const char* bbase;
  unsigned avoff[2], bvoff[2];
  int64_t m0, n0;                                                          // of the tile abase / bbase point into
  auto set_tile = [&](int t, int tid) {
    const int lane = tid & 63;
    const int tm = t / tiles_n, tn = t - tm * tiles_n;
    m0 = (int64_t)tm * AR_BM;
    n0 = (int64_t)tn * AR_BN;
    abase = (const char*)(A + m0 * p.lda);
    bbase = (const char*)(B + n0 * p.ldb);
    const int mrem = (int)(p.M - m0 < AR_BM ? p.M - m0 : AR_BM) - 1, nrem = (int)(p.N - n0 < AR_BN ? p.N - n0 : AR_BN) - 1;   // last valid row
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int row = (wv * 2 + q) * 16 + (lane >> 2);
      const int c = (lane & 3) ^ ((row >> 2) & 3);
      bvoff[q] = (unsigned)(((int64_t)(row < nrem ? row : nrem) * p.ldb + c * 8) * 2);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int row = (tid + 256 * u) >> 2, c = tid & 3;
      avoff[u] = (unsigned)(((int64_t)(row < mrem ? row : mrem) * p.lda + c * 8) * 2);
    }
  };
  set_tile(tile, threadIdx.x);
  // One SGPR base and one 32-bit VGPR offset per DMA instruction serve all twelve K-steps; the K offset is added to the 32-bit
  // offset (one VALU add per instruction).  NOT the instruction's immediate offset field: with it (LDS pointer pre-compensated,
  // since the hardware adds the immediate to both addresses) about one tile in 1e5 read stale B rows right after vmcnt(0) + barrier.
  auto stage_b = [&](auto sc) {
    constexpr int s = decltype(sc)::value;
    char* sb = smem + (s % 3) * AR_SLOT + AR_ATILE + wv * 2048;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const unsigned off = bvoff[q] + (unsigned)(s * (AR_BK * 2));
#ifdef AR_B_VADDR
      const char* gp = bbase + off;
      asm volatile("" : "+v"(gp));
      __builtin_amdgcn_global_load_lds((ar_gbl_void*)gp, (ar_lds_void*)(sb + q * 1024), 16, 0, 0);
#else
      __builtin_amdgcn_global_load_lds((ar_gbl_void*)(bbase + off), (ar_lds_void*)(sb + q * 1024), 16, 0, 0);
#endif
    }
  };
#define AR_IC(N) std::integral_constant<int, (N)>{}
  ar_u32x4 areg[AR_SETS][2];
  // (the immediate offset must be a literal: one statement per K-step)
#ifdef AR_A_VADDR
#define AR_LOAD_A(S, SET)                                                                                                 \
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(areg[SET][0]) : "v"(abase + avoff[0]), "n"((S) * 64));  \
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(areg[SET][1]) : "v"(abase + avoff[1]), "n"((S) * 64));
#else
#define AR_LOAD_A(S, SET)                                                                                                 \
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(areg[SET][0]) : "v"(avoff[0]), "s"(abase), "n"((S) * 64)); \
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(areg[SET][1]) : "v"(avoff[1]), "s"(abase), "n"((S) * 64));
#endif

#define AR_BURST_LO AR_LOAD_A(0, 0) AR_LOAD_A(1, 1) AR_LOAD_A(2, 2) AR_LOAD_A(3, 3)
#define AR_BURST_HI AR_LOAD_A(4, 4) AR_LOAD_A(5, 5)
#define AR_PIN(SET) asm volatile("" : "+v"(areg[SET][0]), "+v"(areg[SET][1]));       /* "this set has landed": a plain value from here on */
  // ---- prologue of the first tile: everything the first steps need is requested at once
  stage_b(AR_IC(0));
  stage_b(AR_IC(1));
  AR_BURST_LO AR_BURST_HI

  f32x16 acc[2][2];
  // retire the whole first burst (B0 B1 A0..A5); later tiles retire theirs inside the previous tile's epilogue
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const float alpha = p.alpha;

  for (;;) {
  // Every per-lane address below is derived from a thread id the compiler cannot see through, once per tile: hoisted out of the
  // tile loop they would stay live across it (LICM is blind to register pressure) and push the kernel past its 168 registers.
  int tl = threadIdx.x;
  asm volatile("" : "+v"(tl));
  const int lane = tl & 63;
  // piece u of a thread is row (tl >> 2) + 64 u: same chunk swizzle ((row >> 2) & 3 is unchanged by + 64), 4096 B further on
  const unsigned adst = (unsigned)((tl >> 2) * 64 + (((tl & 3) ^ ((tl >> 4) & 3)) << 4));
  auto write_a = [&](int s, int set) {                           // register set -> slot s % 3 (A half)
    char* sa = smem + (s % 3) * AR_SLOT + adst;
#pragma unroll
    for (int u = 0; u < 2; ++u) *reinterpret_cast<ar_u32x4*>(sa + u * 4096) = areg[set][u];
  };
  const int frow = lane & 31, fh = lane >> 5;
  auto compute = [&](int s) {
    const char* sa = smem + (s % 3) * AR_SLOT;
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

  const int64_t m0c = m0, n0c = n0;                                        // this tile's origin (set_tile moves m0 / n0 on at step 10)
  const int next = tile + nb;
  const bool has_next = next < ntiles;                                     // workgroup-uniform
  AR_PIN(0) AR_PIN(1)
  __builtin_amdgcn_s_barrier();                                            // waves 2, 3 have read their park areas back
  write_a(0, 0);
  write_a(1, 1);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // ---- K loop, fully unrolled (the vmcnt counts below are the static issue order of the file header)
#define AR_STEP(KT, WAITN, LOADS)                                                                                         \
  {                                                                                                                       \
    if (WAITN >= 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN < 0 ? 0 : WAITN) : "memory");                          \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                    \
    __builtin_amdgcn_s_barrier();                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    if (KT + 2 < AR_NK) stage_b(AR_IC(KT + 2 < AR_NK ? KT + 2 : 0));                                                                                \
    LOADS                                                                                                                 \
    if (KT + 2 < AR_NK) {                                                                                                 \
      AR_PIN((KT + 2) % AR_SETS)                                                                                          \
      write_a(KT + 2, (KT + 2) % AR_SETS);                                                                                \
    }                                                                                                                     \
    compute(KT);                                                                                                          \
  }
  // A6 .. A11 re-use the register sets of A0 .. A5, each issued in the step after its set was written out (LOADS come before the
  // step's own write_a: A6 at step 0 goes to set 0, written at the tile top).  A(kt + 2) is always older than an operation the
  // wait of step kt (or an earlier one) retires: see the file header.
  AR_STEP(0, -1, AR_LOAD_A(6, 0) AR_LOAD_A(7, 1))
  AR_STEP(1, -1, AR_LOAD_A(8, 2))
  AR_STEP(2, 8, AR_LOAD_A(9, 3))
  AR_STEP(3, 6, AR_LOAD_A(10, 4))
  AR_STEP(4, 6, AR_LOAD_A(11, 5))
  AR_STEP(5, 6, )
  AR_STEP(6, 4, )
  AR_STEP(7, 2, )
  AR_STEP(8, 2, )
  {  // step 9: this tile's bias slice goes out BEFORE B11, so that the wait of step 11 (which retires B11) retires it too and two
     // barriers lie between that wait and the epilogue's read: vmcnt counts an LDS-DMA as done a moment before its last bytes are
     // visible in LDS (seen as 16 stale bytes of bias when the read followed the wait directly)
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    {
      int64_t bn = n0c + wc * 64 + lane;
      bn = bn < p.N ? bn : p.N - 1;
      __builtin_amdgcn_global_load_lds((ar_gbl_void*)(biasp + bn), (ar_lds_void*)biasl, 4, 0, 0);
    }
    stage_b(AR_IC(11));
    AR_PIN(11 % AR_SETS)
    write_a(11, 11 % AR_SETS);
    compute(9);
  }
  {  // step 10: the next tile's B0 and the first half of its A burst (slot 0 and every register set are free)
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    int t10 = threadIdx.x;                                                 // (opaque again: nothing per-lane stays live across the K loop)
    asm volatile("" : "+v"(t10));
    if (has_next) {
      set_tile(next, t10);
      stage_b(AR_IC(0));
      AR_BURST_LO
    }
    compute(10);
  }
  {  // step 11: retire B11 and (older) the bias slice (younger: B0', the half burst); the next tile's B1 into slot 1
    if (has_next) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (has_next) stage_b(AR_IC(1));
    compute(11);
  }
  // The LDS reads of compute(11) must have RETURNED before this wave reports in: the compiler is free to leave them in flight across
  // the barrier (their MFMAs can follow it), and another wave's park writes into slot 2 then overtake them -- seen as one tile in
  // ~1e5 multiplying by parked fp32 words.  (In the K loop every barrier has the same explicit wait in front of it.)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();     // all waves done with slot 2 and the A halves before they become park areas

  {
  // ---- epilogue (as gemm_bf16_glds.hip): each wave parks one 32-row half of its 64x64 block at a time (row = 256 B, 16-B chunks
  // XOR (row & 15)) and re-reads it by rows: 16-B vector math and stores.
  // (a second opaque thread id: the epilogue's lane constants must not be live across the K loop either)
  int te = threadIdx.x;
  asm volatile("" : "+v"(te));
  const int lane = te & 63, frow = lane & 31, fh = (lane >> 5) & 1, c8 = lane & 7;
  // park areas (8 KiB per wave): slot 2's two halves, then the A halves of slots 0 and 1
  char* const park = smem + (wv < 2 ? 2 * AR_SLOT + wv * AR_ATILE : (wv - 2) * AR_SLOT);
  const bool n_ok = n0c + wc * 64 + c8 * 8 < p.N;
  // output addresses: this wave's 64 x 64 block origin (uniform, SGPRs) + a 32-bit per-lane element offset
  constexpr int ESZ = OUT_DT == DINOX_BF16 ? 2 : 4;
  const int64_t mw = m0c + wr * 64, nw = n0c + wc * 64;
  char* const cblk = (char*)p.C + (mw * p.ldc + nw) * ESZ;
  char* const ablk = (char*)p.aux + (mw * p.ldaux + nw) * ESZ;
  const int mleft = (int)(p.M - mw < 64 ? p.M - mw : 64);                  // valid rows of the block (may be <= 0)
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
    if (ps == 1 && has_next) {                                             // every accumulator is out: room for A4' A5'
      AR_BURST_HI
    }
    if (ps == 0) {
      // the bias slice and the next tile's burst have landed (only B1' may still be in flight); nothing is stored before this
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const bool hb = (p.epilogue & DINOX_EPI_BIAS) != 0;
      const float4 b0 = *reinterpret_cast<const float4*>(biasl + c8 * 32), b1 = *reinterpret_cast<const float4*>(biasl + c8 * 32 + 16);
      bias[0] = hb ? b0.x : 0.f; bias[1] = hb ? b0.y : 0.f; bias[2] = hb ? b0.z : 0.f; bias[3] = hb ? b0.w : 0.f;
      bias[4] = hb ? b1.x : 0.f; bias[5] = hb ? b1.y : 0.f; bias[6] = hb ? b1.z : 0.f; bias[7] = hb ? b1.w : 0.f;
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
      // alone) need ~50 registers the next tile's burst is sitting in.
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
            *reinterpret_cast<float4*>(ablk + ai + 16 * h) = make_float4(a[0], a[1], a[2], a[3]);
          }
        }
        if (OUT_DT == DINOX_BF16) {
          pv[2 * h] = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
          pv[2 * h + 1] = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
        } else {
          *reinterpret_cast<float4*>(cblk + ci + 16 * h) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
      if (OUT_DT == DINOX_BF16) {
        if (ACT == AR_GELU && p.aux) *reinterpret_cast<ar_u32x4*>(ablk + ai) = ar_u32x4{pa[0], pa[1], pa[2], pa[3]};
        *reinterpret_cast<ar_u32x4*>(cblk + ci) = ar_u32x4{pv[0], pv[1], pv[2], pv[3]};
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  }
  if (!has_next) break;
  tile = next;
  }   // tile loop
#undef AR_STEP
#undef AR_LOAD_A
#undef AR_BURST_LO
#undef AR_BURST_HI
#undef AR_PIN
#undef AR_IC
}

// Envelope inside gemm_bf16_nt_glds's: K = 384 exactly, one problem (no batch), no residual / GELU' epilogue.
bool gemm_bf16_nt_areg_ok(const GemmParams& p) {
  const int64_t ldmax = 1 << 22;                                           // per-lane offsets inside a tile are 32-bit byte offsets
  return p.K == AR_K && p.batch == 1 && !(p.epilogue & (DINOX_EPI_RESIDUAL | DINOX_EPI_DGELU)) && p.M >= 1 && p.lda < ldmax && p.ldb < ldmax &&
         p.ldc < ldmax && p.ldaux < ldmax;
}

int launch_gemm_bf16_nt_areg(const GemmParams& p, hipStream_t st) {
  const int tiles_m = (int)ceil_div(p.M, (int64_t)AR_BM), tiles_n = (int)ceil_div(p.N, (int64_t)AR_BN);
  const int64_t ntile = (int64_t)tiles_m * tiles_n;
  if (ntile > 0x7fffffff) return DINOX_EUNSUPPORTED;
  const size_t lds = 3 * (size_t)AR_SLOT + 4 * 256;
  const bool gelu = (p.epilogue & DINOX_EPI_GELU) != 0;
  // Default: one workgroup per tile.  DINOX_NT_AREG_WORKERS=N (e.g. 768 = 3 per CU) makes N persistent workgroups walk the tiles
  // with the cross-tile prefetch above.  Measured (qkv / fc1 / teacher fc1 at bs256): 154 / 281 / 238 us persistent against
  // 145 / 269 / 223 us one-tile-per-workgroup on the same box: gfx950 has ONE in-order counter for loads and stores, so the first
  // counted wait of a tile also waits for the previous tile's stores to be acknowledged, which costs more than the prefetch saves.
  const char* we = getenv("DINOX_NT_AREG_WORKERS");
  const int workers_env = we ? atoi(we) : 0;
  const unsigned nwork = (unsigned)(workers_env > 0 && ntile > workers_env ? workers_env : ntile);
#define AR_L(OUT, ACT) hipLaunchKernelGGL((gemm_bf16_nt_areg<OUT, ACT>), dim3(nwork), dim3(256), lds, st, p, (int)ntile, tiles_n)
  if (p.out_dtype == DINOX_BF16) {
    if (gelu) AR_L(DINOX_BF16, AR_GELU); else AR_L(DINOX_BF16, AR_PLAIN);
  } else {
    if (gelu) AR_L(DINOX_F32, AR_GELU); else AR_L(DINOX_F32, AR_PLAIN);
  }
#undef AR_L
  return check_launch("gemm_bf16_nt_areg");
}

}  // namespace dinox
