// gemm_bf16_wide.hip -- NT bf16 MFMA GEMM on 128 x 384 tiles for the WIDE products of the block (qkv: N = 3D, fc1 and the GELU'
// product: N = 4D) at short reductions (K <= 576).
//
// Why: the cost of an NT product on this machine is (bytes that come from HBM) / 6 TB/s + (bytes that enter
// LDS) / ~30 TB/s, and with 128 x 128 tiles the token operand of these products enters LDS 9 or 12 times.  tools/tile_probe.hip
// priced the K loops of seven tile shapes: 128 x 384 with eight waves (2 x 4, each 64 x 96: 96 accumulator registers), a TWO-slot
// ring and two workgroups per CU came out best for these shapes (qkv 115.6 -> 96.1 us, fc1 150.5 -> 115.2 us of K loop): the token
// operand is re-read from L2 three (four) times instead of nine (twelve), and sixteen waves per CU hide the one-step prefetch.
//
// STATUS: correct (tests/test_gpu_parity.py::test_gemm_nt_wide) and OPT-IN (DINOX_NT_WIDE=1).  The shorter K loop does not carry
// over to the whole kernel yet: with non-temporal stores on both sides qkv 153 vs 144 us, fc1 231 vs 218 us, teacher fc1 197 vs 202 us,
// GELU' product 274 vs 202 us against gemm_bf16_nt_areg.  What is left after the K loop (~50 us for qkv against ~32 us there) is the epilogue: four passes of park /
// re-read per wave with two workgroups per CU, 7-9 spilled registers in the GELU forms at the 128-VGPR budget of four waves per
// SIMD, and the GELU' side tensor read where it is used.  Kept as the starting point for that work.
//
// K loop: operands global -> LDS by LDS-DMA ([rows][32 k] images, 64 B per row, 16-B chunk c of row r at c ^ ((r >> 2) & 3), the
// swizzle applied to the SOURCE address), per step: vmcnt(0) + lgkmcnt(0) + barrier, request step kt + 1 into the other slot, 12
// MFMAs (32x32x16) per wave from 2 + 3 fragments per 16-k half.
// Epilogue: four passes of 16 rows; each wave parks 16 x 96 accumulators in LDS (rows of 384 B, 16-B chunks XOR (row & 7)) and
// re-reads them by rows: a lane owns 8 consecutive columns of 3 rows per pass -> bias, GELU (+ side tensor) or x GELU' (side tensor
// read), one 16-byte store.  The bias slice of the tile (384 floats) is staged in LDS by plain loads at kernel entry.
#include "common.h"
#include "gemm_common.h"

namespace dinox {

typedef __attribute__((address_space(3))) void gw_lds_void;
typedef __attribute__((address_space(1))) const void gw_gbl_void;
typedef unsigned gw_u32x4 __attribute__((ext_vector_type(4)));

constexpr int GW_BM = 128, GW_BN = 384, GW_BK = 32;
constexpr int GW_ATILE = GW_BM * 64, GW_BTILE = GW_BN * 64, GW_SLOT = GW_ATILE + GW_BTILE;       // 8 + 24 = 32 KiB
constexpr int GW_LDS = 2 * GW_SLOT + GW_BN * 4;                                                    // ring + bias slice

enum { GW_PLAIN = 0, GW_GELU = 1, GW_DGELU = 2 };

__device__ __forceinline__ int gw_xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int ACT>
__global__ __launch_bounds__(512, 4) void gemm_bf16_nt_wide(GemmParams p, int ntiles, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wv >> 2, wc = wv & 3;                                   // 2 x 4 waves, each 64 rows x 96 columns
  const int tile = gw_xcd_remap(blockIdx.x, ntiles);
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int64_t m0 = (int64_t)tm * GW_BM, n0 = (int64_t)tn * GW_BN;
  float* const bias_s = reinterpret_cast<float*>(smem + 2 * GW_SLOT);
  if (threadIdx.x < GW_BN) bias_s[threadIdx.x] = (p.epilogue & DINOX_EPI_BIAS) ? p.bias[n0 + threadIdx.x] : 0.f;   // N % 384 == 0

  // ---- staging: one DMA instruction per wave for A (16 rows x 64 B), three for B; SGPR base + 32-bit per-lane offset
  const char* abase = (const char*)((const bf16_t*)p.A + m0 * p.lda);
  const char* bbase = (const char*)((const bf16_t*)p.B + n0 * p.ldb);
  unsigned avoff, bvoff[3];
  {
    const int mrem = (int)(p.M - m0 < GW_BM ? p.M - m0 : GW_BM) - 1;    // last valid row of the tile
    const int row = wv * 16 + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
    avoff = (unsigned)(((int64_t)(row < mrem ? row : mrem) * p.lda + c * 8) * 2);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int rb = (wv * 3 + q) * 16 + (lane >> 2), cb = (lane & 3) ^ ((rb >> 2) & 3);
      bvoff[q] = (unsigned)(((int64_t)rb * p.ldb + cb * 8) * 2);
    }
  }
  auto stage = [&](int slot, int kt) {
    char* sa = smem + slot * GW_SLOT + wv * 1024;
    char* sb = smem + slot * GW_SLOT + GW_ATILE + wv * 3072;
    __builtin_amdgcn_global_load_lds((gw_gbl_void*)(abase + avoff + kt * 64), (gw_lds_void*)sa, 16, 0, 0);
#pragma unroll
    for (int q = 0; q < 3; ++q) __builtin_amdgcn_global_load_lds((gw_gbl_void*)(bbase + bvoff[q] + kt * 64), (gw_lds_void*)(sb + q * 1024), 16, 0, 0);
  };

  f32x16 acc[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int nk = (int)(p.K / GW_BK), frow = lane & 31, fh = lane >> 5;
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int slot = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this step's pieces (and, at kt = 0, the bias loads)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the previous step's fragment reads: its slot is refilled below
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < nk) stage(slot ^ 1, kt + 1);
    const char* sa = smem + slot * GW_SLOT;
    const char* sb = sa + GW_ATILE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[2], bfr[3];
      const int kc = 2 * ks + fh;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ra = wr * 64 + i * 32 + frow;
        af[i] = *reinterpret_cast<const bf16x8*>(sa + ra * 64 + ((kc ^ ((ra >> 2) & 3)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int rb = wc * 96 + j * 32 + frow;
        bfr[j] = *reinterpret_cast<const bf16x8*>(sb + rb * 64 + ((kc ^ ((rb >> 2) & 3)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  }
  // the LDS reads of the last step must have returned before the ring becomes the park area (gemm_bf16_areg.hip, lessons)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // ---- epilogue: four passes of 16 rows
  char* const park = smem + wv * (16 * 384);
  const bool ag = (p.epilogue & DINOX_EPI_AUXGRAD) != 0;                 // workgroup-uniform
  const float alpha = p.alpha;
  const int64_t mw = m0 + wr * 64, nw = n0 + wc * 96;
  const int mleft = (int)(p.M - mw < 64 ? p.M - mw : 64);               // valid rows of the wave's block (may be <= 0)
  char* const cblk = (char*)p.C + (mw * p.ldc + nw) * 2;
  char* const ablk = (char*)p.aux + (mw * p.ldaux + nw) * 2;
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const int i = ps >> 1, e0 = (ps & 1) * 8;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * fh, col = j * 32 + frow;          // row 0..15 of this pass
        *reinterpret_cast<float*>(park + row * 384 + (((col >> 2) ^ (row & 7)) << 4) + (col & 3) * 4) = acc[i][j][e0 + e];
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      __builtin_amdgcn_sched_barrier(0);
      const int id = lane + 64 * t, row = id / 12, g = id - row * 12;    // 16 rows x 12 groups of 8 columns
      const int mr = ps * 16 + row;
      const float4 lo = *reinterpret_cast<const float4*>(park + row * 384 + (((2 * g) ^ (row & 7)) << 4));
      const float4 hi = *reinterpret_cast<const float4*>(park + row * 384 + (((2 * g + 1) ^ (row & 7)) << 4));
      if (mr >= mleft) continue;
      const unsigned ci = (unsigned)((mr * (int)p.ldc + g * 8) * 2), ai = (unsigned)((mr * (int)p.ldaux + g * 8) * 2);
      gw_u32x4 xraw = {0u, 0u, 0u, 0u};
      if (ACT == GW_DGELU) xraw = *reinterpret_cast<const gw_u32x4*>(ablk + ai);
      unsigned pv[4], pa[4];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        __builtin_amdgcn_sched_barrier(0);
        const float4 x4 = h ? hi : lo;
        const float4 b4 = *reinterpret_cast<const float4*>(bias_s + wc * 96 + g * 8 + 4 * h);
        float v[4] = {x4.x * alpha + b4.x, x4.y * alpha + b4.y, x4.z * alpha + b4.z, x4.w * alpha + b4.w}, a[4];
        if (ACT == GW_GELU) {
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
          pa[2 * h] = (unsigned)f32_to_bf16(a[0]) | ((unsigned)f32_to_bf16(a[1]) << 16);
          pa[2 * h + 1] = (unsigned)f32_to_bf16(a[2]) | ((unsigned)f32_to_bf16(a[3]) << 16);
        }
        if (ACT == GW_DGELU) {
          const unsigned w0 = xraw[2 * h], w1 = xraw[2 * h + 1];
          const float x[4] = {__uint_as_float(w0 << 16), __uint_as_float(w0 & 0xffff0000u), __uint_as_float(w1 << 16),
                              __uint_as_float(w1 & 0xffff0000u)};
          if (ag) {
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] *= x[u];
          } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] *= gelu_fast_grad(x[u]);
          }
        }
        pv[2 * h] = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
        pv[2 * h + 1] = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
      }
      // non-temporal: the output stream must not evict the operand slices the running tiles re-read from L2 (gemm_bf16_areg.hip)
      if (ACT == GW_GELU && p.aux) __builtin_nontemporal_store(gw_u32x4{pa[0], pa[1], pa[2], pa[3]}, reinterpret_cast<gw_u32x4*>(ablk + ai));
      __builtin_nontemporal_store(gw_u32x4{pv[0], pv[1], pv[2], pv[3]}, reinterpret_cast<gw_u32x4*>(cblk + ci));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// Envelope (the alignment rules of gemm_bf16_nt_glds are checked by the caller first): N a multiple of 384, bf16 output, short K, one
// problem, no residual, 32-bit offsets inside a tile.
bool gemm_bf16_nt_wide_ok(const GemmParams& p) {
  const int64_t ldmax = 1 << 22;
  if (p.N % GW_BN || p.N < 2 * GW_BN || p.K % GW_BK || p.K < GW_BK || p.batch != 1 || p.M < 1 || p.out_dtype != DINOX_BF16) return false;
  if (p.epilogue & (DINOX_EPI_RESIDUAL | DINOX_EPI_ACCUM)) return false;
  if ((p.epilogue & DINOX_EPI_GELU) && (p.epilogue & DINOX_EPI_DGELU)) return false;
  if ((p.epilogue & DINOX_EPI_DGELU) && !p.aux) return false;
  if (p.lda >= ldmax || p.ldb >= ldmax || p.ldc >= ldmax) return false;
  if ((p.epilogue & (DINOX_EPI_GELU | DINOX_EPI_DGELU)) && p.ldaux >= ldmax) return false;
  return true;
}

int launch_gemm_bf16_nt_wide(const GemmParams& p, hipStream_t st) {
  const int tiles_m = (int)ceil_div(p.M, (int64_t)GW_BM), tiles_n = (int)(p.N / GW_BN);
  const int64_t ntile64 = (int64_t)tiles_m * tiles_n;
  if (ntile64 > 0x7fffffff) return DINOX_EUNSUPPORTED;
  const unsigned ntile = (unsigned)ntile64;
#define GW_L(ACT)                                                                                                         \
  do {                                                                                                                    \
    auto kern = gemm_bf16_nt_wide<ACT>;                                                                                   \
    if (int rc = reserve_lds(reinterpret_cast<const void*>(kern), GW_LDS, "gemm_bf16_nt_wide")) return rc;                \
    hipLaunchKernelGGL(kern, dim3(ntile), dim3(512), GW_LDS, st, p, (int)ntile, tiles_n);                                 \
  } while (0)
  if (p.epilogue & DINOX_EPI_GELU) GW_L(GW_GELU);
  else if (p.epilogue & DINOX_EPI_DGELU) GW_L(GW_DGELU);
  else GW_L(GW_PLAIN);
#undef GW_L
  return check_launch("gemm_bf16_nt_wide");
}

}  // namespace dinox
