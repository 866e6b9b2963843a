// gemm_bf16_glds.hip -- second-generation NT bf16 MFMA GEMM: direct-to-LDS operand staging and a
// row-major, vectorised fused epilogue.
//
// Why (rocprof, round 1): the products of this path have tiny K (384..1536) and huge M (~1e5 tokens), so a
// 128x128 tile runs only 6..24 K-steps and the per-tile prologue/epilogue is as long as its MFMA work.  The
// first-generation kernel (gemm_bf16.hip) staged operands through VGPRs (ds_write_b128 moves ~79 B/clk/CU,
// MI355X_MICROARCH.md LDS table: slower than the MFMAs it feeds) and stored the accumulator column-wise
// (64-B segments, scalar epilogue math with run-time flag tests).  Here:
//   * operands go global -> LDS with global_load_lds_dwordx4 (16 B/lane, no VGPRs, no ds_write); the LDS
//     image stays lane-linear and the bank swizzle (chunk ^ ((row>>1)&7)) is applied to the per-lane SOURCE
//     address and to the ds_read_b128 address (cdna_hip_programming.md rule 21);
//   * after the K loop each wave parks its 64x64 fp32 accumulator block in its own 16 KiB of the (now idle)
//     LDS stages, XOR-swizzled, and re-reads it row-major: every lane then owns 8 consecutive outputs of a
//     row, so bias / GELU / GELU' / residual use 16-B vector loads and the stores are whole 128-B (bf16) or
//     256-B (fp32) row segments;
//   * epilogue variants are compile-time (no flag tests in the inner code) and GELU uses a 1.5e-7-accurate
//     rational erf instead of erff.
// Envelope: K % 64 == 0, N % 8 == 0, 16-B aligned operands/outputs; anything else takes gemm_bf16_nt.
#include <cstdlib>

#include "common.h"
#include "gemm_common.h"

namespace dinox {

constexpr int GG_BN = 128;

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

__device__ __forceinline__ int gg_xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

enum { GG_PLAIN = 0, GG_GELU = 1, GG_DGELU = 2 };

// swizzle term of a tile row: 8 chunks/row (BK 64): (row>>1)&7 ; 4 chunks/row (BK 32): (row>>2)&3 -- each makes the
// four 16-lane groups of ds_read_b128 hit 16 distinct 16-B bank slots.
template <int CH>
__device__ __forceinline__ int gg_swz(int row) { return CH == 8 ? ((row >> 1) & 7) : ((row >> 2) & 3); }

// 4 waves (2x2).  GG_BM = 128: each wave owns 64x64 (2x2 MFMA tiles);  GG_BM = 256: each wave owns 128x64 (4x2 tiles, 128
// accumulator VGPRs).  The taller wave tile feeds 8 MFMAs from 6 fragment reads instead of 4 from 4 (LDS read traffic
// per MFMA -25 %: with 64x64 wave tiles ds_read_b128 alone keeps the LDS ~50 % busy at the measured MFMA rate) and
// re-uses every staged B tile for twice the rows (global->LDS bytes per FLOP -25 %).
template <int OUT_DT, int ACT, bool RES, int GG_BK, int STAGES, int GG_BM>
__global__ __launch_bounds__(256, (GG_BM == 256 ? 2 : (GG_BK == 64 ? 2 : 3))) void gemm_bf16_nt_glds(GemmParams p, int tiles_m, int tiles_n) {
  constexpr int WAVES = 4;
  constexpr int WM = GG_BM / 64;                  // 32-row MFMA tiles per wave along M (2 or 4)
  constexpr int WROWS = 32 * WM;                  // rows per wave
  constexpr int A_TILE = GG_BM * GG_BK * 2;       // bytes per stage
  constexpr int B_TILE = GG_BN * GG_BK * 2;
  constexpr int STAGE_BYTES = A_TILE + B_TILE;
  constexpr int CH = GG_BK / 8;                   // 16-B chunks per tile row
  constexpr int RPI = 64 / CH;                    // tile rows moved by one wave-instruction (1 KiB)
  constexpr int NQA = GG_BM / RPI / WAVES;        // staging instructions per wave, A tile
  constexpr int NQB = GG_BN / RPI / WAVES;        //                               B tile
  constexpr int ROWB = GG_BK * 2;                 // bytes per tile row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wv >> 1, wc = wv & 1;
  const int tile = gg_xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int64_t m0 = (int64_t)tm * GG_BM, n0 = (int64_t)tn * GG_BN;
  const int64_t bz = blockIdx.y;
  const bf16_t* A = (const bf16_t*)p.A + bz * p.strideA;
  const bf16_t* B = (const bf16_t*)p.B + bz * p.strideB;

  // Per-lane source pointers of this wave's 4 + 4 staging instructions (each moves 8 rows x 128 B).
  // LDS slot (row, c') of a tile receives logical chunk c = c' ^ ((row>>1)&7) of that row.
  const bf16_t* asrc[NQA];
  const bf16_t* bsrc[NQB];
#pragma unroll
  for (int q = 0; q < NQA; ++q) {
    const int row = (wv * NQA + q) * RPI + lane / CH;
    const int c = (lane % CH) ^ gg_swz<CH>(row);
    int64_t gm = m0 + row;
    gm = gm < p.M ? gm : p.M - 1;
    asrc[q] = A + gm * p.lda + c * 8;
  }
#pragma unroll
  for (int q = 0; q < NQB; ++q) {
    const int row = (wv * NQB + q) * RPI + lane / CH;
    const int c = (lane % CH) ^ gg_swz<CH>(row);
    int64_t gn = n0 + row;
    gn = gn < p.N ? gn : p.N - 1;
    bsrc[q] = B + gn * p.ldb + c * 8;
  }
  auto stage = [&](int buf, int64_t k0) {
    char* sa = smem + buf * STAGE_BYTES + wv * (NQA * 1024);
    char* sb = smem + buf * STAGE_BYTES + A_TILE + wv * (NQB * 1024);
#pragma unroll
    for (int q = 0; q < NQA; ++q) __builtin_amdgcn_global_load_lds((gbl_void*)(asrc[q] + k0), (lds_void*)(sa + q * 1024), 16, 0, 0);
#pragma unroll
    for (int q = 0; q < NQB; ++q) __builtin_amdgcn_global_load_lds((gbl_void*)(bsrc[q] + k0), (lds_void*)(sb + q * 1024), 16, 0, 0);
  };

  f32x16 acc[WM][2];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = (int)(p.K / GG_BK);
  const int frow = lane & 31, fh = lane >> 5;

  // ---- epilogue operand prefetch.  The fused epilogue reads one more tile from HBM (GELU' input for DGELU, the fp32 residual for
  // RES); fetched where it is used, that read's full latency is exposed once per pass of every tile (rocprof: +130 us on the dgelu
  // product).  The loads of pass 0 are issued here, ahead of the K loop (older than every staging DMA, so the counted vmcnt waits below
  // still cover what they must), the loads of pass 1 right before pass 0 is parked.
  constexpr int PASSES = (STAGES * STAGE_BYTES >= WAVES * WROWS * 256) ? 1 : WM;    // one pass, or one 32-row MFMA tile row per pass
  constexpr int PROWS = WROWS / PASSES;
  constexpr int NIT = PROWS / 8;                  // row-iterations of one pass (a lane owns 8 consecutive columns of one row in each)
  constexpr bool PF_AUX = ACT == GG_DGELU, PF_RES = RES;
  const int c8 = lane & 7;
  const int64_t n = n0 + wc * 64 + c8 * 8;
  const bool n_ok = n < p.N;                               // N % 8 == 0: a lane's 8 columns are all in or all out
  const int64_t n_ld = n_ok ? n : 0;
  float4 pf_aux[PF_AUX ? NIT : 1][OUT_DT == DINOX_BF16 ? 1 : 2];     // bf16 aux: 16 B per row-iteration; fp32: 32 B
  float4 pf_res[PF_RES ? NIT : 1][2];
  auto prefetch = [&](int ps) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      int64_t m = m0 + wr * WROWS + ps * PROWS + it * 8 + (lane >> 3);
      m = m < p.M ? m : p.M - 1;                           // rows past M are loaded (valid address) and never stored
      if (PF_AUX) {
        const int64_t ai = bz * p.M * p.ldaux + m * p.ldaux + n_ld;
        if (OUT_DT == DINOX_BF16) pf_aux[it][0] = *reinterpret_cast<const float4*>((const bf16_t*)p.aux + ai);
        else {
          pf_aux[it][0] = *reinterpret_cast<const float4*>((const float*)p.aux + ai);
          pf_aux[it][OUT_DT == DINOX_BF16 ? 0 : 1] = *reinterpret_cast<const float4*>((const float*)p.aux + ai + 4);
        }
      }
      if (PF_RES) {
        const float* rp = p.residual + bz * p.M * p.ldr + m * p.ldr + n_ld;
        pf_res[it][0] = *reinterpret_cast<const float4*>(rp);
        pf_res[it][1] = *reinterpret_cast<const float4*>(rp + 4);
      }
    }
  };
  if (PF_AUX || PF_RES) prefetch(0);
  auto compute = [&](int buf) {
    const char* sa = smem + buf * STAGE_BYTES;
    const char* sb = sa + A_TILE;
#pragma unroll
    for (int ks = 0; ks < GG_BK / 16; ++ks) {
      bf16x8 af[WM], bfr[2];
      const int kc = 2 * ks + fh;
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        const int ra = wr * WROWS + i * 32 + frow;
        af[i] = *reinterpret_cast<const bf16x8*>(sa + ra * ROWB + ((kc ^ gg_swz<CH>(ra)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int rb = wc * 64 + i * 32 + frow;
        bfr[i] = *reinterpret_cast<const bf16x8*>(sb + rb * ROWB + ((kc ^ gg_swz<CH>(rb)) << 4));
      }
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  };
  if (STAGES == 2) {
    stage(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < nk) stage(cur ^ 1, (int64_t)(kt + 1) * GG_BK);
      compute(cur);
      __syncthreads();
    }
  } else {
    // 3-stage ring, two K-steps of LDS-DMA in flight across the barrier: counted vmcnt + raw s_barrier (a
    // __syncthreads() would drain the DMA queue).  Per step: wait for THIS step's tile (leave the next one in
    // flight) -> barrier (every wave's pieces landed; every wave is done reading the slot refilled next) ->
    // issue step kt+2 into the slot read at step kt-1 -> MFMAs.
    stage(0, 0);
    if (nk > 1) stage(1, GG_BK);
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NQA + NQB) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (kt + 2 < nk) stage(buf == 0 ? 2 : buf - 1, (int64_t)(kt + 2) * GG_BK);
      compute(buf);
      buf = buf == 2 ? 0 : buf + 1;
    }
    // The last step's LDS reads must have returned before this wave reports in (the compiler may leave them in flight across the
    // barrier, their MFMAs after it): park writes of another wave into the same slot could overtake them.  The park area is slots
    // 0-1; the last step reads slot (nk - 1) % 3, so this bites when nk % 3 != 0 (K = 1024, 4096: ViT-L), not on ViT-S / ViT-B.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();     // all waves done with the ring before it becomes the park area
  }

  // ---- epilogue: park the wave's accumulator block in LDS (row = 256 B, 16-B chunks XOR (row&15)) and re-read it by
  // rows.  With BK = 64 the four waves park 64 rows each at once (64 KiB = both stages); with BK = 32 the stages are
  // 32 KiB, so each wave parks its two 32-row halves one after the other.
  char* park = smem + wv * (PROWS * 256);
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
  for (int ps = 0; ps < PASSES; ++ps) {
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      if (PASSES != 1 && i != ps) continue;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = (PASSES != 1 ? 0 : i * 32) + (e & 3) + 8 * (e >> 2) + 4 * fh;
          const int nn = j * 32 + frow;
          *reinterpret_cast<float*>(park + row * 256 + ((((nn >> 2) ^ (row & 15))) << 4) + (nn & 3) * 4) = acc[i][j][e];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float4 cur_aux[PF_AUX ? NIT : 1][OUT_DT == DINOX_BF16 ? 1 : 2];
    float4 cur_res[PF_RES ? NIT : 1][2];
    if (PF_AUX || PF_RES) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        if (PF_AUX) {
          cur_aux[it][0] = pf_aux[it][0];
          if (OUT_DT != DINOX_BF16) cur_aux[it][OUT_DT == DINOX_BF16 ? 0 : 1] = pf_aux[it][OUT_DT == DINOX_BF16 ? 0 : 1];
        }
        if (PF_RES) {
          cur_res[it][0] = pf_res[it][0];
          cur_res[it][1] = pf_res[it][1];
        }
      }
      if (ps + 1 < PASSES) prefetch(ps + 1);
    }
#pragma unroll
    for (int it = 0; it < PROWS / 8; ++it) {
      const int row = it * 8 + (lane >> 3);
      const int64_t m = m0 + wr * WROWS + ps * PROWS + row;
      const float4 lo = *reinterpret_cast<const float4*>(park + row * 256 + (((2 * c8) ^ (row & 15)) << 4));
      const float4 hi = *reinterpret_cast<const float4*>(park + row * 256 + (((2 * c8 + 1) ^ (row & 15)) << 4));
      if (m >= p.M || !n_ok) continue;
      float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = v[u] * alpha + bias[u];
      if (ACT == GG_GELU) {
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
          const int64_t ai = bz * p.M * p.ldaux + m * p.ldaux + n;
          if (OUT_DT == DINOX_BF16) {
            s16x8 pk;
#pragma unroll
            for (int u = 0; u < 8; ++u) pk[u] = (short)f32_to_bf16(a[u]);
            __builtin_nontemporal_store(pk, reinterpret_cast<s16x8*>((bf16_t*)p.aux + ai));   // (bf16 outputs: see gemm_bf16_areg.hip, ar_store)
          } else {
            *reinterpret_cast<float4*>((float*)p.aux + ai) = make_float4(a[0], a[1], a[2], a[3]);
            *reinterpret_cast<float4*>((float*)p.aux + ai + 4) = make_float4(a[4], a[5], a[6], a[7]);
          }
        }
      }
      if (ACT == GG_DGELU) {
        float x[8];
        if (OUT_DT == DINOX_BF16) {
          const float4 raw = cur_aux[it][0];
          const uint32_t w[4] = {__float_as_uint(raw.x), __float_as_uint(raw.y), __float_as_uint(raw.z), __float_as_uint(raw.w)};
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            x[2 * u] = __uint_as_float(w[u] << 16);
            x[2 * u + 1] = __uint_as_float(w[u] & 0xffff0000u);
          }
        } else {
          const float4 x0 = cur_aux[it][0], x1 = cur_aux[it][OUT_DT == DINOX_BF16 ? 0 : 1];
          x[0] = x0.x; x[1] = x0.y; x[2] = x0.z; x[3] = x0.w; x[4] = x1.x; x[5] = x1.y; x[6] = x1.z; x[7] = x1.w;
        }
        if (p.epilogue & DINOX_EPI_AUXGRAD) {
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] *= x[u];
        } else {
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] *= gelu_fast_grad(x[u]);
        }
      }
      if (RES) {
        const float4 r0 = cur_res[it][0], r1 = cur_res[it][1];
        v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w;
        v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
      }
      const int64_t ci = bz * p.strideC + m * p.ldc + n;
      if (OUT_DT == DINOX_BF16) {
        s16x8 pk;
#pragma unroll
        for (int u = 0; u < 8; ++u) pk[u] = (short)f32_to_bf16(v[u]);
        __builtin_nontemporal_store(pk, reinterpret_cast<s16x8*>((bf16_t*)p.C + ci));
      } else {
        *reinterpret_cast<float4*>((float*)p.C + ci) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>((float*)p.C + ci + 4) = make_float4(v[4], v[5], v[6], v[7]);
      }
    }
    if (PASSES != 1) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
}

static bool al16(const void* q) { return (((uintptr_t)q) & 15) == 0; }

bool gemm_bf16_nt_glds_ok(const GemmParams& p) {
  if (p.in_dtype != DINOX_BF16 || p.transA || p.transB) return false;
  if ((p.K % 64) || (p.N & 7) || (p.lda & 7) || (p.ldb & 7) || (p.strideA & 7) || (p.strideB & 7)) return false;
  if (!al16(p.A) || !al16(p.B) || !al16(p.C)) return false;
  const int esz = p.out_dtype == DINOX_BF16 ? 2 : 4;
  if ((p.ldc * esz) & 15 || (p.strideC * esz) & 15) return false;
  if (p.epilogue & DINOX_EPI_ACCUM) return false;
  if ((p.epilogue & DINOX_EPI_BIAS) && !al16(p.bias)) return false;
  if ((p.epilogue & DINOX_EPI_RESIDUAL) && (!al16(p.residual) || (p.ldr & 3))) return false;
  if ((p.epilogue & (DINOX_EPI_GELU | DINOX_EPI_DGELU)) && p.aux && (!al16(p.aux) || ((p.ldaux * esz) & 15))) return false;
  if (p.batch > 65535) return false;
  return true;
}

int launch_gemm_bf16_nt_glds(const GemmParams& p, hipStream_t st) {
  static const int knob = getenv("DINOX_NT_BK") ? atoi(getenv("DINOX_NT_BK")) : 0;   // tuning knobs (A/B testing)
  static const int knob_bm = getenv("DINOX_NT_BM") ? atoi(getenv("DINOX_NT_BM")) : 0;
  // measured (tools/gemm_bench.py): K <= 512 runs faster on the 3-stage BK=32 ring, longer K on the 2-stage BK=64 form
  // ... unless the tile count fills the 512 resident slots of the BK = 64 form so badly (bs 64: fc2 / dX are 603 tiles = 1.18 rounds) that the 768 slots
  // of the BK = 32 form win although it is ~10 % slower per tile (measured at bs 256: fc2 233 vs 210 us, dX 195 vs 175 us)
  int bk_auto = p.K <= 512 ? 32 : 64;
  if (bk_auto == 64) {
    const int64_t nt = ceil_div(p.M, (int64_t)128) * ceil_div(p.N, (int64_t)GG_BN) * p.batch;
    const double e512 = (double)nt / (double)(ceil_div(nt, (int64_t)512) * 512), e768 = (double)nt / (double)(ceil_div(nt, (int64_t)768) * 768);
    if (0.9 * e768 > e512) bk_auto = 32;
  }
  const int bk = knob ? knob : bk_auto;
  // the 256-row tile (128x64 per wave) measures within +-5 % of the 128-row tile on every hot-path shape (tools/gemm_bench.py),
  // so the smaller one (more workgroups per CU, finer tail) stays the default; DINOX_NT_BM=256 selects the other
  const int bm = knob_bm ? knob_bm : 128;
  const int stages = bk == 64 ? 2 : 3;

  const int tiles_m = (int)ceil_div(p.M, bm), tiles_n = (int)ceil_div(p.N, GG_BN);
  const int64_t ntile = (int64_t)tiles_m * tiles_n;
  if (ntile > 0x7fffffff) return DINOX_EUNSUPPORTED;
  dim3 grid((unsigned)ntile, (unsigned)p.batch);
  const size_t lds = (size_t)stages * (bm + GG_BN) * bk * 2;
  const int act = (p.epilogue & DINOX_EPI_GELU) ? GG_GELU : (p.epilogue & DINOX_EPI_DGELU) ? GG_DGELU : GG_PLAIN;
  const bool res = (p.epilogue & DINOX_EPI_RESIDUAL) != 0;
#define GG_L(OUT, ACT, RES, BK, ST, BM)                                                                                  \
  do {                                                                                                                    \
    auto kern = gemm_bf16_nt_glds<OUT, ACT, RES, BK, ST, BM>;                                                             \
    if (int rc = reserve_lds(reinterpret_cast<const void*>(kern), lds, "gemm_bf16_nt_glds")) return rc;                  \
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p, tiles_m, tiles_n);                                           \
  } while (0)
#define GG(OUT, ACT, RES)                                                       \
  do {                                                                          \
    if (bm == 256) {                                                            \
      if (bk == 32) GG_L(OUT, ACT, RES, 32, 3, 256); else GG_L(OUT, ACT, RES, 64, 2, 256); \
    } else {                                                                    \
      if (bk == 32) GG_L(OUT, ACT, RES, 32, 3, 128); else GG_L(OUT, ACT, RES, 64, 2, 128); \
    }                                                                           \
  } while (0)
#define GG_ACT(OUT, RES)                                                                  \
  do {                                                                                    \
    if (act == GG_GELU) GG(OUT, GG_GELU, RES); else if (act == GG_DGELU) GG(OUT, GG_DGELU, RES); else GG(OUT, GG_PLAIN, RES); \
  } while (0)
  if (p.out_dtype == DINOX_BF16) {
    if (res) GG_ACT(DINOX_BF16, true); else GG_ACT(DINOX_BF16, false);
  } else {
    if (res) GG_ACT(DINOX_F32, true); else GG_ACT(DINOX_F32, false);
  }
#undef GG_ACT
#undef GG
#undef GG_L
  return check_launch("gemm_bf16_nt_glds");
}

}  // namespace dinox
