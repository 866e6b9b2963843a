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
// register.  Issue order of vector-memory operations per thread / wave:
//   prologue  B0 B1 (2 DMA each) | A0 .. A7 (2 loads each)                     -> vmcnt(0), A0 A1 written to LDS
//   step 0    B2 | A8 A9      step 1  B3 | A10      step 2  B4 | A11      step s (3..9)  B(s+2)
//   wait before step kt's barrier = number of operations issued after B(kt):  kt 2: 8, kt 3: 6, kt 4: 4, kt 5..10: 2, kt 11: 0
#include "common.h"
#include "gemm_common.h"

namespace dinox {

typedef __attribute__((address_space(3))) void ar_lds_void;
typedef __attribute__((address_space(1))) const void ar_gbl_void;
typedef unsigned ar_u32x4 __attribute__((ext_vector_type(4)));      // (a native vector: HIP's uint4 is a struct, which inline asm cannot tie)

constexpr int AR_K = 384, AR_BK = 32, AR_NK = AR_K / AR_BK, AR_BM = 128, AR_BN = 128;
constexpr int AR_ATILE = AR_BM * AR_BK * 2, AR_BTILE = AR_BN * AR_BK * 2, AR_SLOT = AR_ATILE + AR_BTILE;

enum { AR_PLAIN = 0, AR_GELU = 1 };

__device__ __forceinline__ int ar_xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int OUT_DT, int ACT>
__global__ __launch_bounds__(256, 3) void gemm_bf16_nt_areg(GemmParams p, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wv >> 1, wc = wv & 1;
  const int tile = ar_xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int64_t m0 = (int64_t)tm * AR_BM, n0 = (int64_t)tn * AR_BN;
  const bf16_t* A = (const bf16_t*)p.A;
  const bf16_t* B = (const bf16_t*)p.B;

  // ---- B by LDS-DMA: 8 instructions per slot (16 rows x 64 B each), two per wave; slot (row, c') gets chunk c = c' ^ ((row>>2)&3)
  const bf16_t* bsrc[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int row = (wv * 2 + q) * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ((row >> 2) & 3);
    int64_t gn = n0 + row;
    gn = gn < p.N ? gn : p.N - 1;
    bsrc[q] = B + gn * p.ldb + c * 8;
  }
  auto stage_b = [&](int s) {
    char* sb = smem + (s % 3) * AR_SLOT + AR_ATILE + wv * 2048;
#pragma unroll
    for (int q = 0; q < 2; ++q) __builtin_amdgcn_global_load_lds((ar_gbl_void*)(bsrc[q] + s * AR_BK), (ar_lds_void*)(sb + q * 1024), 16, 0, 0);
  };
  // ---- A through registers: per K-step two 16-B pieces per thread (piece id = t + 256 u: row = id / 4, chunk = id % 4)
  const bf16_t* asrc[2];
  unsigned adst[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int id = threadIdx.x + 256 * u;
    const int row = id >> 2, c = id & 3;
    int64_t gm = m0 + row;
    gm = gm < p.M ? gm : p.M - 1;
    asrc[u] = A + gm * p.lda + c * 8;
    adst[u] = (unsigned)(row * 64 + ((c ^ ((row >> 2) & 3)) << 4));
  }
  ar_u32x4 areg[8][2];
  // (the immediate offset must be a literal: one statement per K-step)
#define AR_LOAD_A(S, SET)                                                                                                 \
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(areg[SET][0]) : "v"(asrc[0]), "n"((S) * 64));          \
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(areg[SET][1]) : "v"(asrc[1]), "n"((S) * 64));
  auto write_a = [&](int s, int set) {                           // register set -> slot s % 3 (A half)
    char* sa = smem + (s % 3) * AR_SLOT;
#pragma unroll
    for (int u = 0; u < 2; ++u) *reinterpret_cast<ar_u32x4*>(sa + adst[u]) = areg[set][u];
  };

  // ---- prologue: everything the first steps need is requested at once
  stage_b(0);
  stage_b(1);
  AR_LOAD_A(0, 0) AR_LOAD_A(1, 1) AR_LOAD_A(2, 2) AR_LOAD_A(3, 3) AR_LOAD_A(4, 4) AR_LOAD_A(5, 5) AR_LOAD_A(6, 6) AR_LOAD_A(7, 7)

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
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

  // retire the whole first burst (B0 B1 A0..A7), put A0 and A1 in place
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int s = 0; s < 8; ++s) asm volatile("" : "+v"(areg[s][0]), "+v"(areg[s][1]));
  write_a(0, 0);
  write_a(1, 1);

  // ---- K loop, fully unrolled (the vmcnt counts below are the static issue order of the file header)
#define AR_STEP(KT, WAITN, LOADS)                                                                                         \
  {                                                                                                                       \
    if (WAITN >= 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN < 0 ? 0 : WAITN) : "memory");                          \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                    \
    __builtin_amdgcn_s_barrier();                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    if (KT + 2 < AR_NK) stage_b(KT + 2);                                                                                  \
    LOADS                                                                                                                 \
    if (KT + 2 < AR_NK) write_a(KT + 2, (KT + 2) & 7);                                                                    \
    compute(KT);                                                                                                          \
  }
  // A8 .. A11 re-use the register sets of A0 .. A3 (sets are indexed K-step mod 8), each issued after that set was written out
  AR_STEP(0, -1, AR_LOAD_A(8, 0) AR_LOAD_A(9, 1))
  AR_STEP(1, -1, AR_LOAD_A(10, 2))
  AR_STEP(2, 8, AR_LOAD_A(11, 3))
  AR_STEP(3, 6, )
  AR_STEP(4, 4, )
  AR_STEP(5, 2, )
  {  // A8..A11 have landed (the wait of step 5 left only B6 outstanding): hand them to the compiler as plain values
#pragma unroll
    for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(areg[s][0]), "+v"(areg[s][1]));
  }
  AR_STEP(6, 2, )
  AR_STEP(7, 2, )
  AR_STEP(8, 2, )
  AR_STEP(9, 2, )
  AR_STEP(10, 2, )
  AR_STEP(11, 0, )
#undef AR_STEP
#undef AR_LOAD_A
  __builtin_amdgcn_s_barrier();     // all waves done with the ring before it becomes the park area

  // ---- epilogue (as gemm_bf16_glds.hip): each wave parks one 32-row half of its 64x64 block at a time (row = 256 B, 16-B chunks
  // XOR (row & 15)) and re-reads it by rows: 16-B vector math and stores.
  char* park = smem + wv * (32 * 256);
  const int c8 = lane & 7;
  const int64_t n = n0 + wc * 64 + c8 * 8;
  const bool n_ok = n < p.N;
  float bias[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) bias[u] = 0.f;
  if (n_ok && (p.epilogue & DINOX_EPI_BIAS)) {
    const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n), b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
    bias[0] = b0.x; bias[1] = b0.y; bias[2] = b0.z; bias[3] = b0.w;
    bias[4] = b1.x; bias[5] = b1.y; bias[6] = b1.z; bias[7] = b1.w;
  }
  const float alpha = p.alpha;
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
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + (lane >> 3);
      const int64_t m = m0 + wr * 64 + ps * 32 + row;
      const float4 lo = *reinterpret_cast<const float4*>(park + row * 256 + (((2 * c8) ^ (row & 15)) << 4));
      const float4 hi = *reinterpret_cast<const float4*>(park + row * 256 + (((2 * c8 + 1) ^ (row & 15)) << 4));
      if (m >= p.M || !n_ok) continue;
      float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = v[u] * alpha + bias[u];
      if (ACT == AR_GELU) {
        const bool ag = (p.epilogue & DINOX_EPI_AUXGRAD) != 0;          // workgroup-uniform
        float a[8];
        if (ag) {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            float y, d;
            gelu_fast_both(v[u], y, d);
            a[u] = d;
            v[u] = y;
          }
        } else {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            a[u] = v[u];
            v[u] = gelu_fast(v[u]);
          }
        }
        if (p.aux) {
          const int64_t ai = m * p.ldaux + n;
          if (OUT_DT == DINOX_BF16) {
            s16x8 pk;
#pragma unroll
            for (int u = 0; u < 8; ++u) pk[u] = (short)f32_to_bf16(a[u]);
            *reinterpret_cast<s16x8*>((bf16_t*)p.aux + ai) = pk;
          } else {
            *reinterpret_cast<float4*>((float*)p.aux + ai) = make_float4(a[0], a[1], a[2], a[3]);
            *reinterpret_cast<float4*>((float*)p.aux + ai + 4) = make_float4(a[4], a[5], a[6], a[7]);
          }
        }
      }
      const int64_t ci = m * p.ldc + n;
      if (OUT_DT == DINOX_BF16) {
        s16x8 pk;
#pragma unroll
        for (int u = 0; u < 8; ++u) pk[u] = (short)f32_to_bf16(v[u]);
        *reinterpret_cast<s16x8*>((bf16_t*)p.C + ci) = pk;
      } else {
        *reinterpret_cast<float4*>((float*)p.C + ci) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>((float*)p.C + ci + 4) = make_float4(v[4], v[5], v[6], v[7]);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// Envelope inside gemm_bf16_nt_glds's: K = 384 exactly, one problem (no batch), no residual / GELU' epilogue.
bool gemm_bf16_nt_areg_ok(const GemmParams& p) {
  return p.K == AR_K && p.batch == 1 && !(p.epilogue & (DINOX_EPI_RESIDUAL | DINOX_EPI_DGELU)) && p.M >= 1;
}

int launch_gemm_bf16_nt_areg(const GemmParams& p, hipStream_t st) {
  const int tiles_m = (int)ceil_div(p.M, (int64_t)AR_BM), tiles_n = (int)ceil_div(p.N, (int64_t)AR_BN);
  const int64_t ntile = (int64_t)tiles_m * tiles_n;
  if (ntile > 0x7fffffff) return DINOX_EUNSUPPORTED;
  const size_t lds = 3 * (size_t)AR_SLOT;
  const bool gelu = (p.epilogue & DINOX_EPI_GELU) != 0;
#define AR_L(OUT, ACT) hipLaunchKernelGGL((gemm_bf16_nt_areg<OUT, ACT>), dim3((unsigned)ntile), dim3(256), lds, st, p, tiles_m, tiles_n)
  if (p.out_dtype == DINOX_BF16) {
    if (gelu) AR_L(DINOX_BF16, AR_GELU); else AR_L(DINOX_BF16, AR_PLAIN);
  } else {
    if (gelu) AR_L(DINOX_F32, AR_GELU); else AR_L(DINOX_F32, AR_PLAIN);
  }
#undef AR_L
  return check_launch("gemm_bf16_nt_areg");
}

}  // namespace dinox
