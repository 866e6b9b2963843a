// gemm_pp_common.h -- shared pieces of the persistent ping-pong NT kernels (gemm_bf16_pp.hip: 256 x 256 tiles, gemm_bf16_pp128.hip: 256 x 128):
// the fused epilogue of one wave's accumulator block and the workgroup -> tile assignment.
#pragma once
#include "common.h"
#include "gemm_common.h"

namespace dinox {

typedef __attribute__((address_space(3))) void pp_lds_void;
typedef __attribute__((address_space(1))) const void pp_gbl_void;
typedef unsigned pp_u32x4 __attribute__((ext_vector_type(4)));
typedef float pp_f32x4 __attribute__((ext_vector_type(4)));
typedef float pp_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 pp_bf16x2 __attribute__((ext_vector_type(2)));

enum { PP_PLAIN = 0, PP_GELU = 1, PP_DGELU = 2 };

// two floats -> one dword of two bf16 (ONE v_cvt_pk_bf16_f32; the scalar casts cost a convert each plus a v_or_b32_sdwa)
__device__ __forceinline__ unsigned pp_pack2(float a, float b) {
  const pp_bf16x2 r = __builtin_convertvector(pp_f32x2{a, b}, pp_bf16x2);
  return __builtin_bit_cast(unsigned, r);
}

// This workgroup's tiles.  order 0: a contiguous run of the row-major tile list (the column tiles of a row panel follow each other in
// ONE workgroup); order 1 (default): the workgroups of an XCD interleave over the XCD's share of the list, so the column tiles of a row
// panel run at the same time on neighbouring CUs and share the panel in that XCD's L2 (measured: qkv 123 -> 104 us, plain N 1536
// 154 -> 125 us against order 0).  my_max = the largest tile count among the workgroups this one shares its start with.
__device__ __forceinline__ void pp_my_tiles(int order, int units, int& u_first, int& u_step, int& my, int& my_max) {
  const int nwg = (int)gridDim.x, w = (int)blockIdx.x;
  if ((order & 255) == 0 || (nwg & 7)) {
    const int u0 = (int)((int64_t)units * w / nwg), u1 = (int)((int64_t)units * (w + 1) / nwg);
    u_first = u0; u_step = 1; my = u1 - u0; my_max = (units + nwg - 1) / nwg;
  } else {
    const int xcd = w & 7, j = w >> 3, nxw = nwg >> 3;
    const int x0 = (int)((int64_t)units * xcd / 8), x1 = (int)((int64_t)units * (xcd + 1) / 8);
    u_first = x0 + j; u_step = nxw; my = x0 + j < x1 ? (x1 - x0 - j + nxw - 1) / nxw : 0; my_max = (x1 - x0 + nxw - 1) / nxw;
  }
}

// Epilogue of one wave's block of NS x 16 rows x 64 columns at (mw, nw); (m0t, n0t) is the tile's origin (a wave whose rows / columns
// are all past the edge reads its extra operands from the tile's first rows / columns instead: valid addresses, values never used).
// acc[s][j]: accumulators of v_mfma_f32_16x16x32_bf16 with SWAPPED operands -- lane (fr = lane & 15, fq = lane >> 4) holds row fr,
// columns 16 j + 4 fq .. + 3 of slice s.  A slice passes through the wave's private 4 KiB LDS tile `stage` ([16 rows][256 B], 16-byte
// chunk c of row r at c ^ r: conflict-free both ways) and is re-read by rows: a lane then owns 8 consecutive columns of rows r8 and
// r8 + 8, so memory sees whole 128-byte (bf16) / 256-byte (fp32) row segments.  bias -> GELU (+ GELU' side tensor) | x GELU' ->
// fp32 residual, as in the other NT kernels; bf16 outputs leave by non-temporal stores.  Every address is a wave-uniform 64-bit
// origin + a 32-bit per-lane byte offset.  `mid()` runs after the bias loads have been issued and before the first store.
// Returns nothing; the accumulators are zero afterwards.
template <int OUT_DT, int ACT, bool RES, int NS, typename Mid>
__device__ __forceinline__ void pp_epilogue(const GemmParams& p, pp_f32x4 (&acc)[NS][4], char* stage, int64_t mw, int64_t nw, int64_t m0t,
                                            int64_t n0t, int lane, bool nostore, Mid&& mid) {
  constexpr int ESZ = OUT_DT == DINOX_BF16 ? 2 : 4;
  constexpr int ROWS = NS * 16;
  const int fr = lane & 15, fq = lane >> 4, c8 = lane & 7, r8 = lane >> 3;
  const unsigned st_wr = (unsigned)(fr * 256);
  const float alpha = p.alpha;
  const bool has_bias = (p.epilogue & DINOX_EPI_BIAS) != 0;
  const int mleft = (int)(p.M - mw < ROWS ? p.M - mw : ROWS);                                     // valid rows (may be <= 0)
  const bool n_ok = nw + c8 * 8 < p.N;
  const int col = n_ok ? c8 * 8 : 0;
  const int64_t nshift = nw < p.N ? 0 : nw - n0t, mshift = mleft > 0 ? 0 : mw - m0t;
  char* const cblk = (char*)p.C + (mw * p.ldc + nw) * ESZ;
  char* const xblk = (char*)p.aux + (mw * p.ldaux + nw) * ESZ;
  const char* const xblk_l = xblk - (mshift * p.ldaux + nshift) * ESZ;
  const char* const rblk_l = (const char*)(p.residual + ((mw - mshift) * p.ldr + nw - nshift));
  // bias: loaded unconditionally (no control flow: its wait floats down to the first use, behind the LDS round trip of slice 0)
  float bias[8];
  {
    const bool use = has_bias && n_ok;
    const float* bp = (has_bias ? p.bias : (const float*)p.B) + (use ? nw + c8 * 8 : 0);
    const float4 b0 = *reinterpret_cast<const float4*>(bp), b1 = *reinterpret_cast<const float4*>(bp + 4);
    bias[0] = use ? b0.x : 0.f; bias[1] = use ? b0.y : 0.f; bias[2] = use ? b0.z : 0.f; bias[3] = use ? b0.w : 0.f;
    bias[4] = use ? b1.x : 0.f; bias[5] = use ? b1.y : 0.f; bias[6] = use ? b1.z : 0.f; bias[7] = use ? b1.w : 0.f;
  }
  mid();
  // extra operand rows of slice s (GELU' input 16 B, residual 32 B per lane and row): requested one slice ahead
  constexpr bool PF_AUX = ACT == PP_DGELU;
  constexpr int AUXV = OUT_DT == DINOX_BF16 ? 1 : 2;
  // ring of NPF slots: a block of up to four slices (the 128-wide kernel, which has the registers) requests ALL its rows at once -- one
  // slice ahead, every slice paid a full HBM round trip (measured: 12-17k cycles for the fc2 epilogue of 4 slices) -- the 8-slice block
  // of the 256-wide kernel keeps two slots (its register file is full)
  constexpr int NPF = NS <= 4 ? NS : 2;
  float4 pf_aux[NPF][PF_AUX ? 2 : 1][AUXV], pf_res[NPF][RES ? 2 : 1][2];
  // last row the extra-operand loads may touch, relative to their (possibly shifted) origin: a wave that is past M altogether reads the
  // tile's first rows, of which only M - m0t exist
  const int mclamp = mleft > 0 ? mleft - 1 : (int)(p.M - m0t < ROWS ? p.M - m0t : ROWS) - 1;
  auto prefetch = [&](int s, int slot) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int mr = s * 16 + r8 + 8 * h;
      mr = mr < mclamp ? mr : mclamp;
      if (PF_AUX) {
        const unsigned o = (unsigned)((mr * (int)p.ldaux + col) * ESZ);
#pragma unroll
        for (int v = 0; v < AUXV; ++v) pf_aux[slot][h][v] = *reinterpret_cast<const float4*>(xblk_l + o + 16 * v);
      }
      if (RES) {
        const unsigned o = (unsigned)((mr * (int)p.ldr + col) * 4);
        pf_res[slot][h][0] = *reinterpret_cast<const float4*>(rblk_l + o);
        pf_res[slot][h][1] = *reinterpret_cast<const float4*>(rblk_l + o + 16);
      }
    }
  };
  const unsigned lane_c = (unsigned)((r8 * (int)p.ldc + c8 * 8) * ESZ), lane_x = (unsigned)((r8 * (int)p.ldaux + c8 * 8) * ESZ);
  if (PF_AUX || RES) {
#pragma unroll
    for (int s = 0; s < NPF - 1; ++s) prefetch(s, s);
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    if ((PF_AUX || RES) && s + NPF - 1 < NS) prefetch(s + NPF - 1, (s + NPF - 1) % NPF);
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<pp_f32x4*>(stage + st_wr + ((((j * 4 + fq) ^ fr) & 15) << 4)) = acc[s][j];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      __builtin_amdgcn_sched_barrier(0);                        // one row pass at a time (interleaving them costs registers)
      const int row = r8 + 8 * h, mr = s * 16 + row;
      const pp_f32x4 lo = *reinterpret_cast<const pp_f32x4*>(stage + row * 256 + ((((2 * c8) ^ row) & 15) << 4));
      const pp_f32x4 hi = *reinterpret_cast<const pp_f32x4*>(stage + row * 256 + ((((2 * c8 + 1) ^ row) & 15) << 4));
      unsigned pv[4], pa[4];
      float vf[8], af32[8];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {                          // four columns at a time, packed as soon as they are final
        __builtin_amdgcn_sched_barrier(0);
        const pp_f32x4 x4 = hh ? hi : lo;
        float v[4] = {x4[0], x4[1], x4[2], x4[3]}, a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] * alpha + bias[4 * hh + e];
        if (ACT == PP_GELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float y, d;
            gelu_fast_both(v[e], y, d);
            a[e] = d;
            v[e] = y;
          }
        }
        if (ACT == PP_DGELU) {
          float x[4];
          if (OUT_DT == DINOX_BF16) {
            const float4 raw = pf_aux[s % NPF][h][0];
            const unsigned w0 = __float_as_uint(hh ? raw.z : raw.x), w1 = __float_as_uint(hh ? raw.w : raw.y);
            x[0] = __uint_as_float(w0 << 16); x[1] = __uint_as_float(w0 & 0xffff0000u);
            x[2] = __uint_as_float(w1 << 16); x[3] = __uint_as_float(w1 & 0xffff0000u);
          } else {
            const float4 xv = pf_aux[s % NPF][h][hh ? AUXV - 1 : 0];
            x[0] = xv.x; x[1] = xv.y; x[2] = xv.z; x[3] = xv.w;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= x[e];
        }
        if (RES) {
          const float4 r = pf_res[s % NPF][h][hh];
          v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
        }
        if (OUT_DT == DINOX_BF16) {
          pv[2 * hh] = pp_pack2(v[0], v[1]);
          pv[2 * hh + 1] = pp_pack2(v[2], v[3]);
          if (ACT == PP_GELU) {
            pa[2 * hh] = pp_pack2(a[0], a[1]);
            pa[2 * hh + 1] = pp_pack2(a[2], a[3]);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            vf[4 * hh + e] = v[e];
            af32[4 * hh + e] = a[e];
          }
        }
      }
      if (nostore) {                                            // (diagnostic of the timing tools: all the arithmetic, no output stores)
        if (OUT_DT == DINOX_BF16) asm volatile("" ::"v"(pv[0]), "v"(pv[1]), "v"(pv[2]), "v"(pv[3]));
        if (OUT_DT == DINOX_BF16 && ACT == PP_GELU) asm volatile("" ::"v"(pa[0]), "v"(pa[1]), "v"(pa[2]), "v"(pa[3]));
        if (OUT_DT != DINOX_BF16) asm volatile("" ::"v"(vf[0]), "v"(vf[1]), "v"(vf[2]), "v"(vf[3]), "v"(vf[4]), "v"(vf[5]), "v"(vf[6]), "v"(vf[7]));
      } else if (mr < mleft && n_ok) {
        char* const crow = cblk + (int64_t)(s * 16 + 8 * h) * p.ldc * ESZ;          // (uniform)
        char* const xrow = xblk + (int64_t)(s * 16 + 8 * h) * p.ldaux * ESZ;
        if (OUT_DT == DINOX_BF16) {
          if (ACT == PP_GELU && p.aux) __builtin_nontemporal_store(pp_u32x4{pa[0], pa[1], pa[2], pa[3]}, reinterpret_cast<pp_u32x4*>(xrow + lane_x));
          __builtin_nontemporal_store(pp_u32x4{pv[0], pv[1], pv[2], pv[3]}, reinterpret_cast<pp_u32x4*>(crow + lane_c));
        } else {
          if (ACT == PP_GELU && p.aux) {
            *reinterpret_cast<pp_f32x4*>(xrow + lane_x) = pp_f32x4{af32[0], af32[1], af32[2], af32[3]};
            *reinterpret_cast<pp_f32x4*>(xrow + lane_x + 16) = pp_f32x4{af32[4], af32[5], af32[6], af32[7]};
          }
          *reinterpret_cast<pp_f32x4*>(crow + lane_c) = pp_f32x4{vf[0], vf[1], vf[2], vf[3]};
          *reinterpret_cast<pp_f32x4*>(crow + lane_c + 16) = pp_f32x4{vf[4], vf[5], vf[6], vf[7]};
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[s][j] = pp_f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

// Epilogue / operand envelope shared by both kernels (the caller has checked in_dtype == bf16 and transA == transB == 0).
static inline bool pp_al16(const void* q) { return (((uintptr_t)q) & 15) == 0; }
static inline bool pp_envelope_ok(const GemmParams& p, int bk) {
  const int64_t ldmax = 1 << 21;                                // 32-bit byte offsets inside a tile
  if (p.batch != 1 || p.K < 2 * bk || (p.K % bk) || (p.N & 7) || p.M < 1) return false;
  if ((p.lda & 7) || (p.ldb & 7) || p.lda >= ldmax || p.ldb >= ldmax || p.ldc >= ldmax) return false;
  if (!pp_al16(p.A) || !pp_al16(p.B) || !pp_al16(p.C)) return false;
  const int esz = p.out_dtype == DINOX_BF16 ? 2 : 4;
  if ((p.ldc * esz) & 15) return false;
  const int e = p.epilogue;
  if (e & DINOX_EPI_ACCUM) return false;
  if ((e & DINOX_EPI_BIAS) && !pp_al16(p.bias)) return false;
  if ((e & DINOX_EPI_GELU) && (e & DINOX_EPI_DGELU)) return false;
  if (e & (DINOX_EPI_GELU | DINOX_EPI_DGELU)) {
    if (e & DINOX_EPI_RESIDUAL) return false;
    if ((e & DINOX_EPI_DGELU) && !p.aux) return false;
    if (p.aux && !(e & DINOX_EPI_AUXGRAD)) return false;        // the side tensor carries GELU' (the hot path's form)
    if (p.aux && (!pp_al16(p.aux) || ((p.ldaux * esz) & 15) || p.ldaux >= ldmax)) return false;
  }
  if ((e & DINOX_EPI_RESIDUAL) && (!pp_al16(p.residual) || (p.ldr & 3) || p.ldr >= ldmax)) return false;
  return true;
}

}  // namespace dinox
