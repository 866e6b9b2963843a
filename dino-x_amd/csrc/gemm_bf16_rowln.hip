// gemm_bf16_rowln.hip -- x_new = residual + a W^T + bias  AND  y = LayerNorm(x_new)  in ONE kernel, for model width N = 384.
//
// Replaces, per transformer block and network pass (reference zoo/arch.py:94-97):
//     proj / fc2 product (+ bias + residual add)      one NT GEMM writing the fp32 residual stream (158 MB at bs 256) ...
//     nn.LayerNorm of the next sub-block              ... which a LayerNorm launch read straight back (51 us, 48 launches per step)
// with one launch on 128 x 384 tiles: a workgroup owns 128 COMPLETE rows of the residual stream, so its epilogue can emit
// x_new (fp32, the residual stream the backward pass and the next residual add need), LayerNorm(x_new) in the GEMM operand dtype
// of the consumer (bf16; fp32 for the model's final norm) and the row statistics (mean, rstd) the LayerNorm backward needs.
//
// K loop (from the 128 x 384 probe of DESIGN.md section 4 / the former opt-in gemm_bf16_wide.hip, whose tests it passed): eight
// waves 2 x 4, each 64 rows x 96 columns = 96 accumulator registers; operands global -> LDS by LDS-DMA into [rows][32 k] images (64 B
// per row, 16-B chunk c of row r at c ^ ((r >> 2) & 3), swizzle applied to the SOURCE address); two-slot ring, per step:
// vmcnt(0) + lgkmcnt(0) + barrier, request step kt + 1, 12 MFMAs (32x32x16) per wave; two workgroups per CU.
//
// The MFMA operands are SWAPPED (weights as the A operand, tokens as B), so an accumulator block is [32 features][32 tokens] and a
// lane holds, of token row (lane & 31), four runs of 4 consecutive features per block -- 48 of the wave's 96 columns of that row,
// the other 48 sit in lane ^ 32.  The epilogue therefore needs NO parking in LDS and no passes: x = acc + bias + residual is
// formed in place in the 96 accumulator registers and stored as 16-byte pieces (a first version parked 16-row slices in LDS and
// re-read them by rows: four workgroup barriers per tile, ~30 spilled registers, projLN 150 us against 91 + 52 us for the two
// launches it replaces); a row's sum and M2 over the wave's 96 columns are 48 lane-local terms + one exchange with lane ^ 32
// (two passes over registers: exact, no E[x^2] - mean^2 cancellation); the four column-waves of a row meet through 8 bytes of LDS
// per wave and row and ONE workgroup barrier per tile and merge by Chan's formula (equal counts: mean = avg of means, M2 = sum M2
// + 96 sum (mean_w - mean)^2).  Global traffic of the epilogue (residual in, x out, y out) passes through a 4-KiB per-wave LDS tile so
// that memory sees whole 128-byte lines (second version; the first let every lane load / store its own 16-byte pieces: 32 rows x 32
// bytes per instruction, 49 of 164 us for the residual loads alone).
#include <cstdlib>

#include "common.h"
#include "gemm_common.h"

namespace dinox {

typedef __attribute__((address_space(3))) void rl_lds_void;
typedef __attribute__((address_space(1))) const void rl_gbl_void;
typedef unsigned rl_u32x4 __attribute__((ext_vector_type(4)));

constexpr int RL_BM = 128, RL_BN = 384, RL_BK = 32;
constexpr int RL_ATILE = RL_BM * 64, RL_BTILE = RL_BN * 64, RL_SLOT = RL_ATILE + RL_BTILE;       // 8 + 24 = 32 KiB
// LDS: ring of NS slots | bias, gamma, beta slices (3 x 384 floats) | [128 rows][4 column-waves] x (mean, M2)
constexpr int rl_vec(int ns) { return ns * RL_SLOT; }
constexpr int rl_stat(int ns) { return rl_vec(ns) + 3 * RL_BN * 4; }
constexpr int rl_lds(int ns) { return rl_stat(ns) + RL_BM * 4 * 8; }            // NS = 2: 72.5 KiB (two workgroups per CU); NS = 4: 136.5 KiB

struct RowLnParams {
  const bf16_t* A;       // [M][K]
  const bf16_t* W;       // [384][K]
  const float* bias;     // [384] or null
  const float* residual; // [M][384] fp32 or null
  float* x_out;          // [M][384] fp32
  const float* gamma;    // [384]
  const float* beta;     // [384]
  void* y;               // [M][384] bf16 or fp32
  float* mean;           // [M]
  float* rstd;           // [M]
  int64_t M;
  int K;
  float eps;
};

// NS = 2: two-slot ring, two workgroups per CU (short reductions: proj).  NS = 4: four-slot ring with counted waits, three K-steps
// (96 KiB) in flight, one workgroup per CU (long reductions: fc2) -- the two-slot loop alone takes 185 us at K = 1536.
template <int Y_DT, int NS>
__global__ __launch_bounds__(512, NS == 2 ? 4 : 2) void gemm_bf16_rowln(RowLnParams p) {
  constexpr int RL_VEC = rl_vec(NS), RL_STAT = rl_stat(NS);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wv >> 2, wc = wv & 3;                                   // 2 x 4 waves, each 64 rows x 96 columns
  const int64_t m0 = (int64_t)blockIdx.x * RL_BM;
  float* const vec_s = reinterpret_cast<float*>(smem + RL_VEC);
  if (threadIdx.x < RL_BN) {
    vec_s[threadIdx.x] = p.bias ? p.bias[threadIdx.x] : 0.f;
    vec_s[RL_BN + threadIdx.x] = p.gamma[threadIdx.x];
    vec_s[2 * RL_BN + threadIdx.x] = p.beta[threadIdx.x];
  }

  // ---- staging: one DMA instruction per wave for A (16 rows x 64 B), three for W; SGPR base + 32-bit per-lane offset
  const char* abase = (const char*)(p.A + m0 * p.K);
  const char* bbase = (const char*)p.W;
  unsigned avoff, bvoff[3];
  {
    const int mrem = (int)(p.M - m0 < RL_BM ? p.M - m0 : RL_BM) - 1;    // last valid row of the tile
    const int row = wv * 16 + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
    avoff = (unsigned)(((int64_t)(row < mrem ? row : mrem) * p.K + c * 8) * 2);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int rb = (wv * 3 + q) * 16 + (lane >> 2), cb = (lane & 3) ^ ((rb >> 2) & 3);
      bvoff[q] = (unsigned)(((int64_t)rb * p.K + cb * 8) * 2);
    }
  }
  auto stage = [&](int slot, int kt) {
    char* sa = smem + slot * RL_SLOT + wv * 1024;
    char* sb = smem + slot * RL_SLOT + RL_ATILE + wv * 3072;
    __builtin_amdgcn_global_load_lds((rl_gbl_void*)(abase + avoff + kt * 64), (rl_lds_void*)sa, 16, 0, 0);
#pragma unroll
    for (int q = 0; q < 3; ++q) __builtin_amdgcn_global_load_lds((rl_gbl_void*)(bbase + bvoff[q] + kt * 64), (rl_lds_void*)(sb + q * 1024), 16, 0, 0);
  };

  f32x16 acc[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int nk = p.K / RL_BK, frow = lane & 31, fh = lane >> 5;
  if constexpr (NS == 4) {
    // ---- long reductions: two wave groups in anti-phase (the schedule of gemm_bf16_pp.hip / gemm_bf16_tnbig.hip).  Waves 0-3 (rows
    // 0-63) and 4-7 (rows 64-127) are the two waves of every SIMD; in each barrier interval one group reads the fragments of a whole
    // K-step (L slot: 10 ds_read_b128, two of its four staging requests, the counted wait that retires step kt + 1) while the other
    // issues the step's 12 MFMAs (M slot, the other two requests behind them).  Group 1 runs one interval late:
    //   interval 2 kt: G0 L(kt) | G1 M(kt - 1);   interval 2 kt + 1: G0 M(kt) | G1 L(kt)
    // Stage kt + 3 goes into the slot of stage kt - 1, whose last reader (G1, L(kt - 1)) finished before the barrier in front of interval
    // 2 kt.  The steady state is straight-line code; the last three steps (nothing left to request) run a generic copy.
    const int grp = wr;
    auto wait_younger = [&](int n) {                            // wave-uniform n (prologue and the last steps only)
      switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      }
    };
#define RL_SYNC                                                                                                          \
  {                                                                                                                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                   \
    __builtin_amdgcn_s_barrier();                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
  }
    const int npro = nk < NS - 1 ? nk : NS - 1;
    for (int s0 = 0; s0 < npro; ++s0) stage(s0, s0);
    if (npro >= 3) wait_younger(8);
    else wait_younger((npro - 1) * 4);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (the vector slices written above)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (grp == 1) {
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    bf16x8 fa[4], fb[6];
    auto read_frags = [&](const char* sa) {
      const char* sb = sa + RL_ATILE;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int kc = 2 * ks + fh;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int ra = wr * 64 + i * 32 + frow;
          fa[ks * 2 + i] = *reinterpret_cast<const bf16x8*>(sa + ra * 64 + ((kc ^ ((ra >> 2) & 3)) << 4));
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int rb = wc * 96 + j * 32 + frow;
          fb[ks * 3 + j] = *reinterpret_cast<const bf16x8*>(sb + rb * 64 + ((kc ^ ((rb >> 2) & 3)) << 4));
        }
      }
    };
    auto mma = [&]() {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks * 3 + j], fa[ks * 2 + i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    };
    int slot = 0;                                               // ring slot of stage kt
    unsigned koff = (unsigned)(NS - 1) * 64;                    // byte offset of stage kt + 3 along K
    const int nsteady = nk - (NS - 1);
    for (int kt = 0; kt < nsteady; ++kt) {
      char* const rs = smem + ((slot + NS - 1) & (NS - 1)) * RL_SLOT;      // slot of stage kt + 3 (= of stage kt - 1)
      read_frags(smem + slot * RL_SLOT);
      __builtin_amdgcn_global_load_lds((rl_gbl_void*)(abase + avoff + koff), (rl_lds_void*)(rs + wv * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((rl_gbl_void*)(bbase + bvoff[0] + koff), (rl_lds_void*)(rs + RL_ATILE + wv * 3072), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");          // younger than stage kt + 1: stage kt + 2 and the two requests above
      RL_SYNC
      mma();
      __builtin_amdgcn_global_load_lds((rl_gbl_void*)(bbase + bvoff[1] + koff), (rl_lds_void*)(rs + RL_ATILE + wv * 3072 + 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((rl_gbl_void*)(bbase + bvoff[2] + koff), (rl_lds_void*)(rs + RL_ATILE + wv * 3072 + 2048), 16, 0, 0);
      RL_SYNC
      slot = (slot + 1) & (NS - 1);
      koff += 64;
    }
    for (int kt = nsteady > 0 ? nsteady : 0; kt < nk; ++kt) {
      read_frags(smem + slot * RL_SLOT);
      if (kt + 1 < nk) wait_younger((nk - 2 - kt) * 4);
      RL_SYNC
      mma();
      if (!(kt + 1 == nk && grp == 1)) RL_SYNC
      slot = (slot + 1) & (NS - 1);
    }
#undef RL_SYNC
  } else {
  for (int s0 = 0; s0 < NS - 1 && s0 < nk; ++s0) stage(s0, s0);
  for (int kt = 0; kt < nk; ++kt) {
    const int slot = kt & (NS - 1);
    // this step's pieces have landed (at kt = 0 also the vector loads); with NS = 4 the two younger stages (4 instructions per wave
    // each) may stay in flight
    if (NS == 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      const int younger = nk - 1 - kt;
      if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the previous step's fragment reads: its slot is refilled below
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (kt + NS - 1 < nk) stage((kt + NS - 1) & (NS - 1), kt + NS - 1);
    const char* sa = smem + slot * RL_SLOT;
    const char* sb = sa + RL_ATILE;
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
        for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);   // [features][tokens]
    }
  }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                          // every wave has left the ring: it becomes the epilogue's staging area

  // ---- epilogue.  acc[i][j][r]: token row wr*64 + i*32 + frow, feature wc*96 + j*32 + 8*(r >> 2) + 4*fh + (r & 3).
  // The accumulators hold a token ROW per lane, which is what the LayerNorm statistics want -- but memory wants whole 128-byte
  // lines: a lane reading / writing its own 16-byte pieces touches 32 rows x 32 bytes per instruction (measured: 49 of 164 us for
  // the residual loads alone).  So every [32 rows][32 columns] block passes through a 4-KiB per-wave LDS tile ([row][eight 16-byte
  // chunks], chunk ^ (row & 7)): residual in by coalesced loads (eight lanes = one 128-byte line) and read back row-per-lane;
  // x out written row-per-lane and read back coalesced; y (bf16) likewise with 64-byte rows.
  float2* const stat = reinterpret_cast<float2*>(smem + RL_STAT);
  char* const stg = smem + wv * 4096;
  const int64_t mw = m0 + wr * 64;
  const int nrows = (int)(p.M - mw < 0 ? 0 : (p.M - mw > 64 ? 64 : p.M - mw));     // valid rows of the wave's 64-row block (uniform)
  const int64_t mwl = nrows > 0 ? mw : p.M - 1;                                    // loads stay inside the tensors
  const char* const rbase = (const char*)(p.residual ? p.residual + mwl * RL_BN : p.x_out) + wc * 96 * 4;
  char* const xbase = (char*)(p.x_out + mw * RL_BN) + wc * 96 * 4;
  char* const ybase = (char*)p.y + (mw * RL_BN + wc * 96) * (Y_DT == DINOX_BF16 ? 2 : 4);
  const bool has_res = p.residual != nullptr;
  const int cr = lane >> 3, cc = lane & 7;                               // coalesced side: row cr + 8 q, chunk cc
  const int rowl = frow;                                                 // row-per-lane side: row frow, chunks 2 g + fh
  unsigned rp_off[4];                                                    // this lane's four chunk addresses in the staging tile
#pragma unroll
  for (int g = 0; g < 4; ++g) rp_off[g] = (unsigned)(rowl * 128 + (((2 * g + fh) ^ (rowl & 7)) << 4));
  const float* const bias_s = vec_s + wc * 96;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      // residual block in
      float4 rin[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = i * 32 + 8 * q + cr;
        const int rl = r < nrows ? r : (nrows > 0 ? nrows - 1 : 0);
        rin[q] = has_res ? *reinterpret_cast<const float4*>(rbase + (unsigned)((rl * RL_BN + j * 32 + cc * 4) * 4)) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rt = 8 * q + cr;
        *reinterpret_cast<float4*>(stg + rt * 128 + ((cc ^ (rt & 7)) << 4)) = rin[q];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 r4 = *reinterpret_cast<const float4*>(stg + rp_off[g]);
        const float4 b4 = *reinterpret_cast<const float4*>(bias_s + j * 32 + 8 * g + 4 * fh);
        const float x0 = acc[i][j][4 * g] + b4.x + r4.x, x1 = acc[i][j][4 * g + 1] + b4.y + r4.y;
        const float x2 = acc[i][j][4 * g + 2] + b4.z + r4.z, x3 = acc[i][j][4 * g + 3] + b4.w + r4.w;
        acc[i][j][4 * g] = x0;
        acc[i][j][4 * g + 1] = x1;
        acc[i][j][4 * g + 2] = x2;
        acc[i][j][4 * g + 3] = x3;
        s += (x0 + x1) + (x2 + x3);
        *reinterpret_cast<float4*>(stg + rp_off[g]) = make_float4(x0, x1, x2, x3);          // same chunk this lane has just read
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // x block out
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rt = 8 * q + cr, r = i * 32 + rt;
        const float4 v = *reinterpret_cast<const float4*>(stg + rt * 128 + ((cc ^ (rt & 7)) << 4));
        if (r < nrows) *reinterpret_cast<float4*>(xbase + (unsigned)((r * RL_BN + j * 32 + cc * 4) * 4)) = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();                                    // the tile is refilled by the next block
    }
    s += __shfl_xor(s, 32, 64);                                           // the other 48 columns of the wave's 96
    const float mw_ = s * (1.0f / 96.0f);
    float m2 = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float d = acc[i][j][e] - mw_;
        m2 += d * d;
      }
    m2 += __shfl_xor(m2, 32, 64);
    if (fh == 0) stat[(wr * 64 + i * 32 + frow) * 4 + wc] = make_float2(mw_, m2);
  }
  __syncthreads();                                                        // the four column-waves of every row have written
  const float* const gam_s = vec_s + RL_BN + wc * 96;
  const float* const bet_s = vec_s + 2 * RL_BN + wc * 96;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = i * 32 + frow;
    const float4 s01 = *reinterpret_cast<const float4*>(stat + (wr * 64 + r) * 4);
    const float4 s23 = *reinterpret_cast<const float4*>(stat + (wr * 64 + r) * 4 + 2);
    const float mean = 0.25f * ((s01.x + s01.z) + (s23.x + s23.z));
    const float d0 = s01.x - mean, d1 = s01.z - mean, d2 = s23.x - mean, d3 = s23.z - mean;
    const float var = ((s01.y + s01.w) + (s23.y + s23.w) + 96.0f * ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3))) * (1.0f / 384.0f);
    const float rstd = rsqrtf(var + p.eps);
    if (r < nrows && wc == 0 && fh == 0) {
      p.mean[mw + r] = mean;
      p.rstd[mw + r] = rstd;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 g4 = *reinterpret_cast<const float4*>(gam_s + j * 32 + 8 * g + 4 * fh);
        const float4 b4 = *reinterpret_cast<const float4*>(bet_s + j * 32 + 8 * g + 4 * fh);
        const float y0 = (acc[i][j][4 * g] - mean) * rstd * g4.x + b4.x, y1 = (acc[i][j][4 * g + 1] - mean) * rstd * g4.y + b4.y;
        const float y2 = (acc[i][j][4 * g + 2] - mean) * rstd * g4.z + b4.z, y3 = (acc[i][j][4 * g + 3] - mean) * rstd * g4.w + b4.w;
        if (Y_DT == DINOX_BF16) {
          uint2 pk;
          pk.x = (unsigned)f32_to_bf16(y0) | ((unsigned)f32_to_bf16(y1) << 16);
          pk.y = (unsigned)f32_to_bf16(y2) | ((unsigned)f32_to_bf16(y3) << 16);
          // bf16 tile: 64-byte rows, four 16-byte chunks; this lane's 8 bytes are half fh of chunk g
          *reinterpret_cast<uint2*>(stg + rowl * 64 + ((g ^ (rowl & 3)) << 4) + 8 * fh) = pk;
        } else {
          *reinterpret_cast<float4*>(stg + rp_off[g]) = make_float4(y0, y1, y2, y3);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (Y_DT == DINOX_BF16) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int rt = 16 * q + (lane >> 2), c4 = lane & 3, rr = i * 32 + rt;
          const rl_u32x4 v = *reinterpret_cast<const rl_u32x4*>(stg + rt * 64 + ((c4 ^ (rt & 3)) << 4));
          if (rr < nrows) *reinterpret_cast<rl_u32x4*>(ybase + (unsigned)((rr * RL_BN + j * 32 + c4 * 8) * 2)) = v;
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int rt = 8 * q + cr, rr = i * 32 + rt;
          const float4 v = *reinterpret_cast<const float4*>(stg + rt * 128 + ((cc ^ (rt & 7)) << 4));
          if (rr < nrows) *reinterpret_cast<float4*>(ybase + (unsigned)((rr * RL_BN + j * 32 + cc * 4) * 4)) = v;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
}

}  // namespace dinox

namespace dinox {
bool gemm_bf16_nt_pp384_ln_ok(int64_t M, int K);                                                   // gemm_bf16_pp384.hip
int launch_gemm_bf16_nt_pp384_ln(const void* a, const void* w, const float* bias, const float* residual, float* x_out, const float* gamma,
                                 const float* beta, float eps, void* y, float* mean, float* rstd, int64_t M, int K, hipStream_t st);
}  // namespace dinox

using namespace dinox;

extern "C" int dinox_linear_residual_ln_ok(int64_t M, int N, int K) {
  return (N == RL_BN && K >= RL_BK && K % RL_BK == 0 && M >= 1 && M * (int64_t)K * 2 < ((int64_t)1 << 40) &&
          (int64_t)RL_BM * K * 2 < ((int64_t)1 << 31)) ? 1 : 0;
}

extern "C" int dinox_linear_residual_ln(const void* a, const void* w, const float* bias, const float* residual, float* x_out,
                                        const float* gamma, const float* beta, float eps, void* y, int y_dtype, float* mean,
                                        float* rstd, int64_t M, int N, int K, void* stream) {
  DX_REQUIRE(a && w && x_out && gamma && beta && y && mean && rstd, DINOX_EINVAL, "linear_residual_ln: null pointer");
  DX_REQUIRE(dinox_linear_residual_ln_ok(M, N, K), DINOX_EUNSUPPORTED, "linear_residual_ln: M=%lld N=%d K=%d (needs N = 384, K %% 32 = 0)",
             (long long)M, N, K);
  DX_REQUIRE(y_dtype == DINOX_BF16 || y_dtype == DINOX_F32, DINOX_EINVAL, "linear_residual_ln: y_dtype=%d", y_dtype);
  DX_REQUIRE((((uintptr_t)a | (uintptr_t)w | (uintptr_t)residual | (uintptr_t)x_out | (uintptr_t)y) & 15) == 0, DINOX_EALIGN,
             "linear_residual_ln: operands must be 16-byte aligned");
  RowLnParams p{(const bf16_t*)a, (const bf16_t*)w, bias, residual, x_out, gamma, beta, y, mean, rstd, M, K, eps};
  const unsigned tiles = (unsigned)ceil_div(M, (int64_t)RL_BM);
  hipStream_t st = as_stream(stream);
  {
    // The full-row 208 x 384 kernel with the LayerNorm epilogue (gemm_bf16_pp384.hip) for bf16 y on a chip's worth of rows.
    // DINOX_ROWLN_PP (read per call): 0 = never, 1 = every shape in its envelope (tests), unset = M >= 40000 (about a round of its 208-row
    // tiles: at bs 64, M = 25 728 = 124 tiles, the step is 0.25 ms shorter on the 128 x 384 kernel's 201 tiles).
    const char* e = getenv("DINOX_ROWLN_PP");
    const int mode = e ? atoi(e) : -1;
    if (y_dtype == DINOX_BF16 && mode != 0 && gemm_bf16_nt_pp384_ln_ok(M, K) && (mode > 0 || M >= 40000))
      return launch_gemm_bf16_nt_pp384_ln(a, w, bias, residual, x_out, gamma, beta, eps, y, mean, rstd, M, K, st);
  }
#define RL_LAUNCH(YDT, NS)                                                                                                        \
  do {                                                                                                                            \
    if (int rc = reserve_lds(reinterpret_cast<const void*>(gemm_bf16_rowln<YDT, NS>), rl_lds(NS), "linear_residual_ln")) return rc; \
    hipLaunchKernelGGL((gemm_bf16_rowln<YDT, NS>), dim3(tiles), dim3(512), rl_lds(NS), st, p);                                   \
  } while (0)
  const bool deep = K > 576;
  if (y_dtype == DINOX_BF16) {
    if (deep) RL_LAUNCH(DINOX_BF16, 4); else RL_LAUNCH(DINOX_BF16, 2);
  } else {
    if (deep) RL_LAUNCH(DINOX_F32, 4); else RL_LAUNCH(DINOX_F32, 2);
  }
#undef RL_LAUNCH
  return check_launch("linear_residual_ln");
}
