// mlp_fused.hip -- fc1 -> GELU -> fc2 (+ bias, + fp32 residual) in ONE kernel for passes that save nothing for a backward
// (the EMA teacher, inference): out = residual + b2 + GELU(xn W1^T + b1) W2^T, zoo/arch.py:75-76 with :96's skip connection.
//
// Why: at ViT-S sizes the MLP products are HBM-bound and the hidden activation [tokens][4D] is two thirds of
// their traffic (written by fc1, read by fc2: 620 MB per block at 512 views).  Here it never leaves the chip.
//
// One wave owns 32 tokens and ALL D outputs of them: D/32 accumulator tiles of v_mfma_f32_32x32x16_bf16 (192 VGPRs at D = 384)
// plus its 32 x D slice of xn as MFMA B fragments in registers (96 VGPRs) -- 1 wave per SIMD, 4 waves = 128 tokens per workgroup.
// The hidden dimension is walked in chunks of 32:
//   phase 1   Hc^T[32 hidden][32 tokens] = W1c[32][D] . xn^T          D/16 MFMAs, one dependent chain
//   GELU      on the 16 accumulator values per lane, + b1, packed to bf16
//   phase 3   Y[32 tokens][D] += Hc[32 tokens][32 hidden] . W2c^T      2 K-steps x D/32 tiles = D/16 MFMAs, independent
// The Hc^T accumulator IS the A operand of phase 3 (lanes = tokens in both layouts; cdna_hip_programming.md "an accumulator tile
// as the next MFMA's operand").  A lane's 8 consecutive accumulator registers hold hidden rows {0-3, 8-11} + 4h of the tile, not 8
// consecutive ones; instead of permuting W2's K order (8-byte LDS reads, 2-way bank conflicts) the ROWS of W1 are fed permuted
// (sigma below, an address computation), so that register e of half h holds hidden 16 (e>>3) + 8h + (e&7) and W2's fragments are
// plain 16-byte reads.
// Software pipeline over chunks, iteration s: phase 1 of chunk s (split over two accumulators by K-step parity, so that a
// dependent MFMA is four issue slots behind its producer) interleaved 1:1 with phase 3 of chunk s-2, and GELU of chunk s-1 spread
// over the same instruction stream (VALU work in the MFMAs' shadow).  Weights stream L2 -> LDS by LDS-DMA in a 3-slot ring, slot t =
// {W1 chunk t, W2 chunk t-2} = 48 KiB, counted vmcnt + raw s_barrier, same tile images and swizzles as gemm_bf16_glds.hip.
//
// STATUS (round 1): correct (tests/test_gpu_parity.py::test_mlp_fused_matches_two_gemms) but NOT yet faster: 534 us at M = 102 912,
// D = 384, H = 1536 against 438 us for the two GEMM launches (gemm_bf16_nt_glds) it replaces, so dinox/ops.py keeps it opt-in
// (DINOX_FUSED_MLP=1).  Cycle stamps: 4 000 cycles per 32-hidden chunk for 1 536 cycles of MFMA; with MFMAs and GELU removed a
// chunk still takes 2 300 cycles -- barrier + counted waits, 12 LDS-DMA issues, the first fragment group's LDS latency, scalar bias
// loads: with ONE wave per SIMD (468 VGPRs) nothing overlaps them.  The prologue (xn fragments straight from HBM) and the
// three-pass epilogue add ~15 us per 128-token tile.  What would change the picture: a wider hidden chunk per iteration (needs
// more LDS than 160 KiB allows next to a 3-slot ring) or a second wave per SIMD (needs the accumulator tile split over waves).
#include "common.h"

namespace dinox {

typedef __attribute__((address_space(3))) void mf_lds_void;
typedef __attribute__((address_space(1))) const void mf_gbl_void;

constexpr int MF_BM = 128;

// hidden row (within a 32-row chunk) that A-operand row r must carry so that accumulator register e of lane-half h ends up with
// hidden 16 (e>>3) + 8h + (e&7):  rows 4-7 <-> 8-11 swapped inside each 16.
__device__ __forceinline__ int mf_sigma(int r) { return (r & 16) | (((r >> 2) & 1) << 3) | (((r >> 3) & 1) << 2) | (r & 3); }

// Wait until only the PENDING newest LDS reads are outstanding; the fragment registers are in/out operands of the statement, so no
// MFMA that uses them can be moved above it.
template <int PENDING, int G>
__device__ __forceinline__ void mf_retire(bf16x8 (&a)[G], bf16x8 (&b)[G]) {
  if constexpr (G == 4)
    asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(PENDING));
  else
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]) : "n"(PENDING));
}

template <int ND>
__global__ __launch_bounds__(256, 1) void mlp_fused_fwd_bf16(const bf16_t* __restrict__ xn, const bf16_t* __restrict__ w1,
                                                            const float* __restrict__ b1, const bf16_t* __restrict__ w2,
                                                            const float* __restrict__ b2, const float* __restrict__ res,
                                                            float* __restrict__ out, int64_t M, int H) {
  constexpr int D = 32 * ND;
  constexpr int KS = D / 16;                 // K-steps of phase 1 = MFMAs of phase 3 per chunk
  constexpr int W1_BYTES = 32 * D * 2;       // [D/64 k-tiles][32 rows][128 B]
  constexpr int W2_BYTES = D * 64;           // [D rows][64 B]
  constexpr int SLOT = W1_BYTES + W2_BYTES;
  constexpr int NQ1 = W1_BYTES / 1024 / 4;   // LDS-DMA instructions per wave and slot, W1 part
  constexpr int NQ2 = W2_BYTES / 1024 / 4;   //                                         W2 part
  extern __shared__ __attribute__((aligned(16))) char mf_smem[];         // [3 slots]; becomes the park area of the epilogue
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int frow = lane & 31, fh = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * MF_BM + wv * 32;
  const int NC = H / 32;

  // ---- this wave's 32 x D slice of xn as B fragments: lane (token frow, half fh) holds xn[token][16 ks + 8 fh .. +7]
  bf16x8 xb[KS];
  {
    int64_t tok = m0 + frow;
    tok = tok < M ? tok : M - 1;
    const bf16_t* xr = xn + tok * D + 8 * fh;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xb[ks] = *reinterpret_cast<const bf16x8*>(xr + 16 * ks);
  }

  // ---- staging: per-lane source offsets (elements) of this wave's DMA instructions
  // W1 part of a slot: instruction q (0 .. D/16-1) moves 8 rows x 128 B of k-tile q/4; LDS slot (row, c') gets chunk c = c' ^ ((row>>1)&7)
  int w1off[NQ1], w2off[NQ2];
#pragma unroll
  for (int i = 0; i < NQ1; ++i) {
    const int q = wv * NQ1 + i;
    const int kt = q >> 2, row = (q & 3) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    w1off[i] = row * D + kt * 64 + c * 8;                       // + chunk * 32 * D
  }
  // W2 part: instruction q (0 .. D/16-1) moves 16 rows x 64 B; LDS slot (row, c') gets chunk c = c' ^ ((row>>2)&3)
#pragma unroll
  for (int i = 0; i < NQ2; ++i) {
    const int q = wv * NQ2 + i;
    const int row = q * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ((row >> 2) & 3);
    w2off[i] = row * H + c * 8;                                 // + chunk * 32
  }
  // slot t = {W1 chunk t, W2 chunk t-2}, t = 0 .. NC+1; out-of-range chunks are clamped (valid, finite data) so that every
  // iteration has the same straight-line body: their products meet a zero A operand or are never used
  auto stage = [&](int t) {
    char* base = mf_smem + (t % 3) * SLOT;
    const bf16_t* s1 = w1 + (int64_t)(t < NC ? t : NC - 1) * 32 * D;
    const bf16_t* s2 = w2 + (t >= 2 ? t - 2 : 0) * 32;
#pragma unroll
    for (int i = 0; i < NQ1; ++i)
      __builtin_amdgcn_global_load_lds((mf_gbl_void*)(s1 + w1off[i]), (mf_lds_void*)(base + (wv * NQ1 + i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < NQ2; ++i)
      __builtin_amdgcn_global_load_lds((mf_gbl_void*)(s2 + w2off[i]), (mf_lds_void*)(base + W1_BYTES + (wv * NQ2 + i) * 1024), 16, 0, 0);
  };

  f32x16 yacc[ND];
#pragma unroll
  for (int nt = 0; nt < ND; ++nt)
#pragma unroll
    for (int e = 0; e < 16; ++e) yacc[nt][e] = 0.f;
  bf16x8 pa[2];                                                 // A operand of phase 3 (chunk s-2): GELU output packed, two K-steps
  s16x8 pn[2];                                                  // the next one being built (chunk s-1)
  f32x16 hprev;                                                 // fc1 accumulator of chunk s-1
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      pa[j][e] = 0;
      pn[j][e] = 0;
    }
#pragma unroll
  for (int e = 0; e < 16; ++e) hprev[e] = 0.f;

  const int srow = mf_sigma(frow);                              // W1 row this lane feeds as A-operand row frow
  // Retire the xn loads HERE and hand the fragments to the loop as plain register values: while hipcc still counts them as
  // pending loads it puts s_waitcnt vmcnt(0) in front of their first use inside the loop, every iteration, which drains the
  // DMA ring (measured: 3.3 us per chunk instead of ~1).
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(xb[ks]));
  stage(0);
  stage(1);
  float bnext[32];                                              // fc1 bias of the chunk whose GELU runs next (SGPRs)
#pragma unroll
  for (int i = 0; i < 32; ++i) bnext[i] = b1[i];
#pragma unroll 1
  for (int s = 0; s <= NC + 1; ++s) {
    // wait for slot s (leave slot s+1 in flight) -> barrier (every wave's pieces landed; every wave is done with the slot that is
    // refilled next) -> issue slot s+2
    if (s <= NC) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NQ1 + NQ2) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    f32x16 he, ho;
#pragma unroll
    for (int e = 0; e < 16; ++e) he[e] = ho[e] = 0.f;
    // b1 of chunk s-1: its 32 floats were requested by SCALAR loads at the end of the previous iteration (wave-uniform address:
    // lgkmcnt, scalar cache) and are pinned into SGPRs here; the half a lane needs is picked by fh.  Why this shape: a vector
    // global load, or an LDS read of a staged copy, makes hipcc put s_waitcnt vmcnt(0) in front of it -- the LDS-DMA ring would be
    // drained every chunk; without the SGPR pin hipcc folds the per-half select of an element into such a vector load; and a
    // scalar load issued here instead of one iteration early exposes its latency (1 wave per SIMD: nothing else hides it).
    float bsel[16];
#pragma unroll
    for (int i = 0; i < 32; ++i) asm volatile("" : "+s"(bnext[i]));
#pragma unroll
    for (int e = 0; e < 16; ++e) bsel[e] = fh ? bnext[16 * (e >> 3) + 8 + (e & 7)] : bnext[16 * (e >> 3) + (e & 7)];
#pragma unroll
    for (int e = 0; e < 16; ++e) asm volatile("" : "+v"(bsel[e]));
    // Fragments are fetched a group of G K-steps ahead of the MFMAs that use them (two register sets): left to itself hipcc
    // re-uses one register quad for every fragment and waits lgkmcnt(0) in front of each MFMA.
    constexpr int G = KS % 4 == 0 ? 4 : 2, NG = KS / G, EPG = (16 + NG - 1) / NG;
    // Fragment reads are issued by hand (inline ds_read_b128) and retired by counted s_waitcnt lgkmcnt: hipcc only ever emits
    // lgkmcnt(0) here, which also waits for the group just issued -- with one wave per SIMD that exposes the LDS latency 6 times
    // per chunk.  The wait statement takes the fragment registers as in/out operands, so no MFMA that uses them can move above it.
    bf16x8 fa[2][G], fb[2][G];
    // addresses: six per-lane bases (A: one per K-step position inside a 64-wide k-tile; B: one per K-step of the chunk), the rest
    // of every address is an instruction immediate (k-tile * 4 KiB, output tile * 2 KiB)
    const unsigned slot0 = (unsigned)(uintptr_t)(mf_lds_void*)mf_smem + (unsigned)((s % 3) * SLOT);
    unsigned abase[4], bbase[2];
#pragma unroll
    for (int q = 0; q < 4; ++q) abase[q] = slot0 + (unsigned)(srow * 128 + (((2 * q + fh) ^ ((srow >> 1) & 7)) << 4));
#pragma unroll
    for (int q = 0; q < 2; ++q) bbase[q] = slot0 + (unsigned)(W1_BYTES + frow * 64 + (((2 * q + fh) ^ ((frow >> 2) & 3)) << 4));
    auto fetch = [&](int g, int set) {
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const int i = g * G + u;                                // phase-1 MFMA i: K-step i (chunk s); phase-3 MFMA i (chunk s-2): K-step i / ND, tile i % ND
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[set][u]) : "v"(abase[i & 3]), "n"((i >> 2) * 4096));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[set][u]) : "v"(bbase[i / ND]), "n"((i % ND) * 2048));
      }
    };
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // nothing of the compiler's own may still be in flight: counted waits follow
    fetch(0, 0);                                                // first fragment group: its LDS latency runs under the DMA issue
    if (s + 2 <= NC + 1) stage(s + 2);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g + 1 < NG) {
        fetch(g + 1, (g + 1) & 1);
        mf_retire<2 * G, G>(fa[g & 1], fb[g & 1]);
      } else {
        mf_retire<0, G>(fa[g & 1], fb[g & 1]);
      }
      __builtin_amdgcn_sched_barrier(0);                        // keep the next group's reads ahead of this group's MFMAs
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const int i = g * G + u;
        if (i & 1) ho = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g & 1][u], xb[i], ho, 0, 0, 0);
        else he = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g & 1][u], xb[i], he, 0, 0, 0);
        yacc[i % ND] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[i / ND], fb[g & 1][u], yacc[i % ND], 0, 0, 0);
      }
      // a slice of GELU(chunk s-1) in the shadow of those MFMAs
#pragma unroll
      for (int e = g * EPG; e < (g + 1) * EPG && e < 16; ++e) {
        float y = gelu_fast(hprev[e] + bsel[e]);
        asm volatile("" : "+v"(y));                             // pin the value HERE: otherwise LLVM sinks the whole GELU below the MFMAs
        pn[e >> 3][e & 7] = (short)f32_to_bf16(y);
      }
    }
    // hand over: chunk s-1's packed activation becomes the phase-3 operand of the next iteration; this chunk's fc1 result waits
    // for its GELU.  (After iteration 0 there is no chunk -1: keep the operand zero.)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      pa[j] = __builtin_bit_cast(bf16x8, pn[j]);
      if (s == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) pa[j][e] = 0;
      }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) hprev[e] = he[e] + ho[e];
    {                                                           // bias of chunk s for the GELU of the next iteration
      const float* bb = b1 + (s < NC ? s : NC - 1) * 32;
#pragma unroll
      for (int i = 0; i < 32; ++i) bnext[i] = bb[i];
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // (reads still in flight across the barrier could be overtaken by park writes)
  __builtin_amdgcn_s_barrier();                                 // every wave is done with the ring: it becomes the park area

  // ---- epilogue: up to 4 output tiles (128 columns) per pass parked per wave (32 rows x 512 B, 16-B chunks XOR (row & 31)),
  // re-read by rows: + b2 + residual, fp32 stores of whole 128-B segments.
  char* park = mf_smem + wv * (32 * 512);
  constexpr int NPASS = (ND + 3) / 4;
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    const int ntiles = ND - 4 * ps < 4 ? ND - 4 * ps : 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j >= ntiles) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * fh;
        const int nn = j * 32 + frow;
        *reinterpret_cast<float*>(park + row * 512 + (((nn >> 2) ^ (row & 31)) << 4) + (nn & 3) * 4) = yacc[4 * ps + j][e];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int cols = ntiles * 32;                               // columns of this pass
#pragma unroll
    for (int it = 0; it < 16; ++it) {                           // 32 rows x 32 four-column groups = 1024 units, 64 lanes
      const int u = it * 64 + lane;
      const int row = u >> 5, c4 = u & 31;
      if (c4 * 4 >= cols) continue;
      const int64_t m = m0 + row;
      const float4 v = *reinterpret_cast<const float4*>(park + row * 512 + ((c4 ^ (row & 31)) << 4));
      if (m >= M) continue;
      const int n = ps * 128 + c4 * 4;
      const float4 bv = *reinterpret_cast<const float4*>(b2 + n);
      const float4 rv = *reinterpret_cast<const float4*>(res + m * D + n);
      *reinterpret_cast<float4*>(out + m * D + n) = make_float4(v.x + bv.x + rv.x, v.y + bv.y + rv.y, v.z + bv.z + rv.z, v.w + bv.w + rv.w);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

template <int ND>
static int launch_mlp_fused(const bf16_t* xn, const bf16_t* w1, const float* b1, const bf16_t* w2, const float* b2, const float* res,
                            float* out, int64_t M, int H, hipStream_t st) {
  constexpr int D = 32 * ND;
  const size_t ring = 3 * (size_t)(32 * D * 2 + D * 64), parkb = 4 * 32 * 512;
  const size_t lds = ring > parkb ? ring : parkb;
  auto kern = mlp_fused_fwd_bf16<ND>;
  if (int rc = reserve_lds(reinterpret_cast<const void*>(kern), lds, "mlp_fwd_fused")) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)ceil_div(M, (int64_t)MF_BM)), dim3(256), lds, st, xn, w1, b1, w2, b2, res, out, M, H);
  return check_launch("mlp_fwd_fused");
}

}  // namespace dinox

using namespace dinox;

extern "C" int dinox_mlp_fwd_fused_ok(int D, int H) {
  return (D == 64 || D == 128 || D == 192 || D == 256 || D == 384) && H >= 64 && H % 32 == 0;
}

extern "C" int dinox_mlp_fwd_fused(const void* xn, const void* w1, const float* b1, const void* w2, const float* b2,
                                   const float* residual, float* out, int64_t M, int D, int H, void* stream) {
  DX_REQUIRE(xn && w1 && b1 && w2 && b2 && residual && out, DINOX_EINVAL, "mlp_fwd_fused: null pointer");
  DX_REQUIRE(M > 0 && M <= ((int64_t)1 << 31) * MF_BM - 1, DINOX_EINVAL, "mlp_fwd_fused: M=%lld", (long long)M);
  DX_REQUIRE(dinox_mlp_fwd_fused_ok(D, H), DINOX_EUNSUPPORTED, "mlp_fwd_fused: D=%d H=%d outside the envelope (D in {64,128,192,256,384}, H %% 32 == 0)", D, H);
  DX_REQUIRE(((((uintptr_t)xn | (uintptr_t)w1 | (uintptr_t)w2 | (uintptr_t)b1 | (uintptr_t)b2 | (uintptr_t)residual | (uintptr_t)out) & 15) == 0),
             DINOX_EALIGN, "mlp_fwd_fused: operands must be 16-byte aligned");
  hipStream_t st = as_stream(stream);
  const bf16_t *x = (const bf16_t*)xn, *a = (const bf16_t*)w1, *b = (const bf16_t*)w2;
  switch (D / 32) {
    case 2: return launch_mlp_fused<2>(x, a, b1, b, b2, residual, out, M, H, st);
    case 4: return launch_mlp_fused<4>(x, a, b1, b, b2, residual, out, M, H, st);
    case 6: return launch_mlp_fused<6>(x, a, b1, b, b2, residual, out, M, H, st);
    case 8: return launch_mlp_fused<8>(x, a, b1, b, b2, residual, out, M, H, st);
    default: return launch_mlp_fused<12>(x, a, b1, b, b2, residual, out, M, H, st);
  }
}
