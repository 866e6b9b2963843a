// attention_bf16.hip -- MFMA flash-style attention core for head_dim 64 on gfx950 (bf16 in, fp32 softmax).
//
// Token counts on this path are small (201 at /16, 261 at /14), so a whole head's K and V (or Q and dO)
// fit in LDS (28 KiB each at N=201) and one wave can hold a full 32-query score strip in registers:
// the forward needs no online-softmax rescaling.  All three kernels use ONE LDS image layout that
// serves both row reads (ds_read_b128, MFMA A operands) and transposed reads (ds_read_b64_tr_b16, MFMA B
// operands whose reduction index runs down the rows) without bank conflicts -- the 8-row x 32-column
// sub-tile image of cdna_hip_programming.md T10(a), adapted to 64-column (128-B) rows.
//
// Score tiles are computed TRANSPOSED (S^T = K . Q^T: keys in the accumulator rows, queries on the lanes)
// so that softmax statistics are lane-local and the accumulator can be re-used directly as the A operand
// of the next product (P^T)^T . V  /  (dS^T)^T . K  with no lane movement (guide section 3, "An accumulator
// tile as the next MFMA's operand"); the other operand is fetched in the matching permuted k order.
//
//   fwd   : attn_fwd_bf16_persist: persistent workgroups, one (image, head) per iteration, the NEXT head's K,V image
//           streamed into the second LDS buffer by LDS-DMA (buffer_load ... lds, swizzle on the source address) while this
//           one computes; one wave per 32 queries.  Up to 7 key tiles (N <= 224): S^T of the whole strip stays in
//           registers (one pass: max, exp, sum, O = P V); more keys: two passes over the key tiles.  Writes lse.
//           attn_fwd_bf16<NKT> is the non-persistent two-pass form (DINOX_ATTN_NO_PERSIST=1, A/B only).
//   bwd dQ: one wave per 32 queries.   K,V images by LDS-DMA; streams key tiles: S^T, dP^T = V . dO^T, dS^T -> dQ += dS K;
//           writes delta = rowsum(dO o O) to the workspace for the dKV kernel.
//   bwd dKV: one wave per 32 keys.     Q,dO images by LDS-DMA, lse/delta in LDS; streams query tiles: S = Q . K^T,
//           dP = dO . V^T -> dV += P^T dO, dK += dS^T Q.
// (two backward kernels recompute S/dP once more than a single fused pass would: 7 instead of 5 tile
//  products, traded for no cross-wave reduction of dQ; attention is ~8 % of the block's FLOPs.)
#include <cstdlib>

#include "common.h"
#include <type_traits>

namespace dinox {

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
constexpr int AT_D = 64;

// byte offset of 16-B chunk ch (0..7) of row r in a [rows][64 bf16] image
__device__ __forceinline__ int img_off(int r, int ch) {
  return 1024 * (r >> 3) + 512 * (ch >> 2) + 64 * (r & 7) + 16 * ((ch & 3) ^ ((r >> 2) & 3));
}

// Cooperative load of `rows_pad` rows x 64 bf16 (rows >= n_valid are zero) from a strided global matrix.
__device__ __forceinline__ void load_image(char* __restrict__ img, const bf16_t* __restrict__ src, int64_t row_stride,
                                           int n_valid, int rows_pad) {
  for (int idx = threadIdx.x; idx < rows_pad * 8; idx += blockDim.x) {
    const int r = idx >> 3, ch = idx & 7;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r < n_valid) v = *reinterpret_cast<const uint4*>(src + (int64_t)r * row_stride + ch * 8);
    *reinterpret_cast<uint4*>(img + img_off(r, ch)) = v;
  }
}

// A-operand fragment (standard k order): lane (row = l&31, hl = l>>5) gets img[row0 + row][16*ks + 8*hl + j], j<8.
__device__ __forceinline__ bf16x8 frag_rows(const char* __restrict__ img, int row0, int ks, int lane) {
  return *reinterpret_cast<const bf16x8*>(img + img_off(row0 + (lane & 31), 2 * ks + (lane >> 5)));
}

// B-operand fragment in the PERMUTED k order that matches an accumulator tile used as the A operand:
// lane (col = l&31, hl = l>>5), element j  <-  img[row0 + 16*s + 8*(j>>2) + 4*hl + (j&3)][d0 + col].
__device__ __forceinline__ bf16x8 frag_tr_perm(const char* __restrict__ img, int row0, int s, int d0, int lane) {
  const int i = lane & 15, g = lane >> 4;
  const int q4 = i >> 2, p = i & 3, hl = g >> 1;
  const int ch = (d0 >> 3) + 2 * (g & 1) + (p >> 1);
  const int r0 = row0 + 16 * s + 4 * hl + q4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + img_off(r0, ch) + 8 * (p & 1)));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + img_off(r0 + 8, ch) + 8 * (p & 1)));
  s16x8 v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
  v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bf16x8, v);
}

// accumulator registers [8s, 8s+8) -> bf16 A fragment of k-step s
__device__ __forceinline__ bf16x8 acc_as_a(const f32x16& x, int s) {
  bf16x8 a;
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = (__bf16)x[8 * s + j];
  return a;
}

// row index inside a 32-row accumulator tile held by register e of lane-half hl
__device__ __forceinline__ int acc_row(int e, int hl) { return (e & 3) + 8 * (e >> 2) + 4 * hl; }

// 4 fragments (ks = 0..3) of one 64-element global row: lane (row = l&31, hl) takes d = 16*ks + 8*hl + j
__device__ __forceinline__ void load_row_frags(bf16x8 (&f)[4], const bf16_t* __restrict__ rowp, int lane) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) f[ks] = *reinterpret_cast<const bf16x8*>(rowp + 16 * ks + 8 * (lane >> 5));
}

__device__ __forceinline__ void zero16(f32x16& x) {
#pragma unroll
  for (int e = 0; e < 16; ++e) x[e] = 0.f;
}

// store a 32x32 accumulator tile (rows = tokens row0.., cols = d0..) as bf16 rows of a strided matrix
__device__ __forceinline__ void store_tile(bf16_t* __restrict__ dst, int64_t row_stride, int row0, int n_valid, int d0,
                                           const f32x16& x, float scale, int lane) {
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int r = row0 + acc_row(e, lane >> 5);
    if (r < n_valid) dst[(int64_t)r * row_stride + d0 + (lane & 31)] = f32_to_bf16(x[e] * scale);
  }
}

// Store a 32-row x 64-column block of fp32 accumulators (two 32x32 tiles: columns 0..31 and 32..63 of one head) as bf16 rows of a
// strided matrix, THROUGH a per-wave LDS scratch of 32 rows x ST_ROWB bytes: the accumulator layout puts one column per lane, so a
// direct store is 32 two-byte stores per lane, each behind its own row test (about 500 instructions per wave: measured 29 of
// 110 us in the forward kernel); re-read by rows, a lane stores four 16-byte pieces and a row leaves as one 128-byte segment.
// dst must be 16-byte aligned and row_stride a multiple of 8 elements.
constexpr int ST_ROWB = 144;                       // 128 B of data + 16: rows r and r + 4 (the two lane halves) land in different banks
constexpr int ST_BYTES = 32 * ST_ROWB;
__device__ __forceinline__ void store_block(bf16_t* __restrict__ dst, int64_t row_stride, int row0, int n_valid, const f32x16& x0,
                                            const f32x16& x1, char* scratch, int lane) {
  const int col = lane & 31, hl = lane >> 5;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    char* rowp = scratch + acc_row(e, hl) * ST_ROWB + col * 2;
    *reinterpret_cast<bf16_t*>(rowp) = f32_to_bf16(x0[e]);
    *reinterpret_cast<bf16_t*>(rowp + 64) = f32_to_bf16(x1[e]);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int r = (lane >> 3) + 8 * it, c = lane & 7;
    const uint4 v = *reinterpret_cast<const uint4*>(scratch + r * ST_ROWB + c * 16);
    if (row0 + r < n_valid) store_stream(reinterpret_cast<uint4*>(dst + (int64_t)(row0 + r) * row_stride + c * 8), v);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------------------------ forward
// Two passes over the key tiles: pass 1 keeps only the running row maximum, pass 2 recomputes each S^T tile, exponentiates
// against the exact maximum and feeds P straight into the PV MFMAs.  Holding all NKT score tiles instead (one pass) costs
// 16*NKT accumulator VGPRs (188 total at N=201): one 7-wave workgroup per CU, load phase and MFMA phase never overlap.
// This form needs ~100 VGPRs -> two workgroups per CU overlap each other's K/V staging; the extra QK^T MFMAs are cheap.
template <int NKT>
__global__ __launch_bounds__(512) void attn_fwd_bf16(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o,
                                                     float* __restrict__ lse, int N, int heads, float sc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int b = blockIdx.x / heads, hh = blockIdx.x % heads;
  const int C = heads * AT_D;
  const int64_t rs = 3 * (int64_t)C;                               // token stride of packed qkv
  const bf16_t* base = qkv + (int64_t)b * N * rs + hh * AT_D;
  const int npad = NKT * 32;
  char* kimg = smem;
  char* vimg = smem + npad * 128;
  float* inv_s = reinterpret_cast<float*>(smem + 2 * npad * 128) + wv * 32;   // per-wave 1/rowsum, query-indexed
  load_image(kimg, base + C, rs, N, npad);
  load_image(vimg, base + 2 * C, rs, N, npad);
  __syncthreads();
  const int qb = blockIdx.y * nw + wv;
  if (qb * 32 >= N) return;
  const int q0 = qb * 32, hl = lane >> 5;
  int qrow = q0 + (lane & 31);
  if (qrow >= N) qrow = N - 1;                                      // clamp: computed, never stored
  bf16x8 qf[4];
  load_row_frags(qf, base + (int64_t)qrow * rs, lane);

  float mx = -INFINITY;
#pragma unroll 1
  for (int kt = 0; kt < NKT; ++kt) {
    f32x16 s;
    zero16(s);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(kimg, kt * 32, ks, lane), qf[ks], s, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = kt * 32 + acc_row(e, hl);
      mx = fmaxf(mx, key < N ? s[e] : -INFINITY);
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * sc;                      // sc > 0: max commutes with the scaling

  f32x16 oacc[2];
  zero16(oacc[0]);
  zero16(oacc[1]);
  float sum = 0.f;
#pragma unroll 1
  for (int kt = 0; kt < NKT; ++kt) {
    f32x16 s;
    zero16(s);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(kimg, kt * 32, ks, lane), qf[ks], s, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = kt * 32 + acc_row(e, hl);
      const float p = key < N ? __expf(s[e] * sc - mx) : 0.f;
      s[e] = p;
      sum += p;
    }
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      const bf16x8 pa = acc_as_a(s, ss);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
        oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, frag_tr_perm(vimg, kt * 32, ss, dt * 32, lane), oacc[dt], 0, 0, 0);
    }
  }
  sum += __shfl_xor(sum, 32, 64);
  if (hl == 0) {
    inv_s[lane] = 1.0f / sum;                                       // lane = query within the block
    if (q0 + lane < N) lse[((int64_t)b * heads + hh) * N + q0 + lane] = mx + __logf(sum);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // O rows are queries in the accumulator layout: fetch each row's 1/sum (4 consecutive rows per register group)
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 iv = *reinterpret_cast<const float4*>(inv_s + 8 * g + 4 * hl);
    const float ivv[4] = {iv.x, iv.y, iv.z, iv.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      oacc[0][4 * g + r] *= ivv[r];
      oacc[1][4 * g + r] *= ivv[r];
    }
  }
  bf16_t* ob = o + (int64_t)b * N * C + hh * AT_D;
  store_tile(ob, C, q0, N, 0, oacc[0], 1.0f, lane);
  store_tile(ob, C, q0, N, 32, oacc[1], 1.0f, lane);
}

// Per-lane constant parts of the fragment addresses.  Tile t of an image starts 4096 B after tile t-1 (32 rows = four 1-KiB
// row groups; the swizzle terms depend on the row inside the tile only), so address = table entry + 4096*t.
struct RowAddr {            // A-operand row reads: entry ks
  int off[4];
  __device__ __forceinline__ void init(int lane) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) off[ks] = img_off(lane & 31, 2 * ks + (lane >> 5));
  }
};
struct TrAddr {             // permuted-k transposed reads: entry [ss][dt][lo/hi]
  int off[2][2][2];
  __device__ __forceinline__ void init(int lane) {
    const int i = lane & 15, g = lane >> 4, q4 = i >> 2, pp = i & 3;
#pragma unroll
    for (int ss = 0; ss < 2; ++ss)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const int ch = (dt * 32 >> 3) + 2 * (g & 1) + (pp >> 1);
        const int r0 = 16 * ss + 4 * (g >> 1) + q4;
        off[ss][dt][0] = img_off(r0, ch) + 8 * (pp & 1);
        off[ss][dt][1] = img_off(r0 + 8, ch) + 8 * (pp & 1);
      }
  }
};
__device__ __forceinline__ bf16x8 rd_rows(const char* img, const RowAddr& a, int ks, int t) {
  return *reinterpret_cast<const bf16x8*>(img + a.off[ks] + t * 4096);
}
__device__ __forceinline__ bf16x8 rd_tr(const char* img, const TrAddr& a, int ss, int dt, int t) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + a.off[ss][dt][0] + t * 4096));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + a.off[ss][dt][1] + t * 4096));
  s16x8 v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
  v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bf16x8, v);
}

// Two transposed 8-byte LDS reads -> one B fragment.  The __restrict__ parameters are not decoration: after inlining they give the
// two reads alias-scope metadata, and hipcc's waitcnt insertion only falls back to "wait for every LDS-DMA in flight" (vmcnt(0),
// which in the persistent forward kernel means the NEXT pair's images, requested a moment earlier) for LDS reads that carry none.
__device__ __forceinline__ bf16x8 rd_tr_pair(const char* __restrict__ p0, const char* __restrict__ p1) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p1);
  s16x8 v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
  v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bf16x8, v);
}

// ------------------------------------------------------------------------------------------ forward, persistent
typedef __attribute__((address_space(3))) void at_lds_void;
typedef __attribute__((address_space(1))) const void at_gbl_void;

// LDS-DMA of one [npad][64] image: wave-instruction `blk` fills rows 8*blk .. 8*blk+7 (1 KiB, lane-linear destination); the
// img_off() placement is produced by permuting the per-lane SOURCE address.  Rows >= n_valid re-read row n_valid-1 (finite
// data; every consumer masks those keys/queries), so no address leaves the tensor.
__device__ __forceinline__ void dma_image(char* __restrict__ img, const bf16_t* __restrict__ src, int64_t row_stride, int n_valid,
                                          int npad, int wv, int nw, int lane) {
  const int half = lane >> 5, rr = (lane >> 2) & 7, slot = lane & 3;
  for (int blk = wv; blk < (npad >> 3); blk += nw) {
    int row = 8 * blk + rr;
    const int ch = (half << 2) | (slot ^ ((row >> 2) & 3));
    row = row < n_valid ? row : n_valid - 1;
    __builtin_amdgcn_global_load_lds((at_gbl_void*)(src + (int64_t)row * row_stride + ch * 8), (at_lds_void*)(img + blk * 1024), 16, 0, 0);
  }
}

// One workgroup per CU walks (batch, head) pairs.  K and V images are double-buffered and filled by LDS-DMA for pair i+1
// while the MFMAs of pair i run, so HBM latency and the image fill never sit on the critical path (the one-pair-per-
// workgroup kernel spends about half its time filling LDS with nothing to overlap).  Wave w owns query block w (nblk <= nw).
template <int NKT>   // NKT > 0: one pass, all NKT score tiles stay in registers (16*NKT VGPRs);  NKT == 0: two passes, any nkt
__global__ __launch_bounds__(512) void attn_fwd_bf16_persist(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o,
                                                             float* __restrict__ lse, int N, int heads, int nkt, float sc,
                                                             int npairs) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const int C = heads * AT_D;
  const int64_t rs = 3 * (int64_t)C;
  const int npad = nkt * 32, img_bytes = npad * 128;
  float* inv_s = reinterpret_cast<float*>(smem + 4 * img_bytes) + wv * 32;
  char* const st_scratch = smem + 4 * img_bytes + 8 * 32 * (int)sizeof(float) + wv * ST_BYTES;     // this wave's output staging area
  const int hl = lane >> 5;
  int pair = blockIdx.x;
  if (pair >= npairs) return;

  auto base_of = [&](int pr) { return qkv + (int64_t)(pr / heads) * N * rs + (pr % heads) * AT_D; };
  auto issue = [&](int pr, int buf) {
    const bf16_t* base = base_of(pr);
    dma_image(smem + (2 * buf) * img_bytes, base + C, rs, N, npad, wv, nw, lane);
    dma_image(smem + (2 * buf + 1) * img_bytes, base + 2 * C, rs, N, npad, wv, nw, lane);
  };

  // The wave's first query block of a pair does not come by per-lane row loads (64 lanes x 16 B from 32 different rows per
  // instruction: 32 tag look-ups each; the address unit of this kernel was stalled by the cache 23 % of the time) but by LDS-DMA, as a
  // 32-row image in the wave's output staging area -- free between the deferred store at the top of an iteration and the next one
  // -- one pair ahead, and is read from there as fragments at the top of its iteration.
  auto q_dma = [&](int pr) {
    const int q0w = wv * 32;
    dma_image(st_scratch, base_of(pr) + (int64_t)q0w * rs, rs, N - q0w, 32, 0, 1, lane);
  };
  q_dma(pair);
  issue(pair, 0);
  // Output stores are DEFERRED by one pair: the tile a wave finishes in iteration i is stored at the top of iteration i + 1, after
  // that iteration's wait and barrier.  vmcnt counts stores as well as loads, so with the stores at the end of an iteration the
  // vmcnt(0) at the top of the next one also waited for them to be acknowledged: 29 of 110 us at the hot-path shape (ablation:
  // without its stores the kernel ran 29 us faster even with all of its arithmetic removed).  Issued at the top, they have a whole
  // pair's arithmetic to drain.  The registers that hold the pending tile are free at that point (no score tile is live).
  f32x16 pend[2];
  float pend_lse = 0.f;
  int pend_pair = -1, pend_q0 = 0;          // wave-uniform
  auto store_pending = [&]() {
    const int pb = pend_pair / heads, ph = pend_pair % heads;
    if (hl == 0 && pend_q0 + lane < N) lse[((int64_t)pb * heads + ph) * N + pend_q0 + lane] = pend_lse;
    store_block(o + (int64_t)pb * N * C + ph * AT_D, C, pend_q0, N, pend[0], pend[1], st_scratch, lane);
  };
  // The first pair's loads are retired here, and hipcc is SHOWN that they are (a use of the fragments, an LDS read of the image):
  // it does not see through the asm, and with loads still pending in its own bookkeeping at the loop head it puts a vmcnt(0) in
  // front of the first MFMA of every iteration -- which waits for the NEXT pair's images, requested a moment earlier.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  {
    const float seen = *reinterpret_cast<const volatile float*>(smem + lane * 4);
    asm volatile("" ::"v"(seen));
  }
  for (int it = 0; pair < npairs; pair += gridDim.x, ++it) {
    const int buf = it & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // my DMA pieces (images and Q block) for this pair have landed
    __builtin_amdgcn_s_barrier();                         // everyone's pieces landed; everyone is done with the other buffer
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(st_scratch + img_off(lane & 31, 2 * ks + hl));
    const int nxt = pair + gridDim.x;
    if (nxt < npairs) issue(nxt, buf ^ 1);
    if (it > 0) store_pending();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the staging area's reads (fragments above, the store's re-reads) have returned
    if (nxt < npairs) q_dma(nxt);
    const char* kimg = smem + (2 * buf) * img_bytes;
    const char* vimg = kimg + img_bytes;
    const int b = pair / heads, hh = pair % heads;
    // One query block of this pair.  DEFER (the wave's last block, always executed so that the pending tile is redefined in every
    // iteration and is never live across another block's arithmetic): keep the normalised tile for the next iteration's stores.
    auto do_block = [&](int qb, auto defer_c) {
      constexpr bool DEFER = decltype(defer_c)::value;
      const int q0 = qb * 32;
      // Key tile kt sits 4096 B after tile kt-1 in the image (32 rows = four 1-KiB row groups, swizzle terms unchanged), so
      // every fragment address is a per-lane constant plus kt*4096: computed once here, not per tile.
      const char* kr[4];
      const char* vt[2][2][2];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) kr[ks] = kimg + img_off(lane & 31, 2 * ks + hl);
      {
        const int i = lane & 15, g = lane >> 4, q4 = i >> 2, pp = i & 3;
#pragma unroll
        for (int ss = 0; ss < 2; ++ss)
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            const int ch = (dt * 32 >> 3) + 2 * (g & 1) + (pp >> 1);
            const int r0 = 16 * ss + 4 * (g >> 1) + q4;
            vt[ss][dt][0] = vimg + img_off(r0, ch) + 8 * (pp & 1);
            vt[ss][dt][1] = vimg + img_off(r0 + 8, ch) + 8 * (pp & 1);
          }
      }
      const float c2 = sc * 1.4426950408889634f;         // exp(x*sc - m) = exp2(x*c2 - m*log2e)
      f32x16 oacc[2];
      zero16(oacc[0]);
      zero16(oacc[1]);
      float sum = 0.f, mx;
      if constexpr (NKT > 0) {
        // one pass: every S^T tile is produced first (independent accumulators: the matrix pipe never waits), the row
        // maximum is taken over registers, and each tile's exp2 / bf16 pack feeds its PV MFMAs while later tiles are
        // still being exponentiated (straight-line code: MFMA and VALU of different tiles overlap inside the wave)
        f32x16 st[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
          zero16(st[kt]);
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
            st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(kr[ks] + kt * 4096), qf[ks], st[kt], 0, 0, 0);
        }
        mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            if (kt == NKT - 1 && !(kt * 32 + acc_row(e, hl) < N)) st[kt][e] = -INFINITY;   // padded keys live in the last tile only
            mx = fmaxf(mx, st[kt][e]);
          }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mx2 = mx * c2;
        mx *= sc;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const float pr = __builtin_amdgcn_exp2f(st[kt][e] * c2 - mx2);      // exp2(-inf) = 0 for padded keys
            st[kt][e] = pr;
            sum += pr;
          }
#pragma unroll
          for (int ss = 0; ss < 2; ++ss) {
            const bf16x8 pa = acc_as_a(st[kt], ss);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
              oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, rd_tr_pair(vt[ss][dt][0] + kt * 4096, vt[ss][dt][1] + kt * 4096), oacc[dt], 0, 0, 0);
            }
          }
        }
      } else {
      mx = -INFINITY;
#pragma unroll 1
      for (int kt = 0; kt < nkt; ++kt) {
        f32x16 st;
        zero16(st);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(kr[ks] + kt * 4096), qf[ks], st, 0, 0, 0);
        if (kt == nkt - 1) {                              // only the last tile can hold padded keys
#pragma unroll
          for (int e = 0; e < 16; ++e) mx = fmaxf(mx, kt * 32 + acc_row(e, hl) < N ? st[e] : -INFINITY);
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e) mx = fmaxf(mx, st[e]);
        }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mx2 = mx * c2;
      mx *= sc;
#pragma unroll 1
      for (int kt = 0; kt < nkt; ++kt) {
        f32x16 st;
        zero16(st);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(kr[ks] + kt * 4096), qf[ks], st, 0, 0, 0);
        if (kt == nkt - 1) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const float pr = kt * 32 + acc_row(e, hl) < N ? __builtin_amdgcn_exp2f(st[e] * c2 - mx2) : 0.f;
            st[e] = pr;
            sum += pr;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const float pr = __builtin_amdgcn_exp2f(st[e] * c2 - mx2);
            st[e] = pr;
            sum += pr;
          }
        }
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
          const bf16x8 pa = acc_as_a(st, ss);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, rd_tr_pair(vt[ss][dt][0] + kt * 4096, vt[ss][dt][1] + kt * 4096), oacc[dt], 0, 0, 0);
          }
        }
      }
      }
      sum += __shfl_xor(sum, 32, 64);
      if (hl == 0) inv_s[lane] = 1.0f / sum;
      const float lse_q = mx + __logf(sum);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 iv = *reinterpret_cast<const float4*>(inv_s + 8 * g + 4 * hl);
        const float ivv[4] = {iv.x, iv.y, iv.z, iv.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          oacc[0][4 * g + r] *= ivv[r];
          oacc[1][4 * g + r] *= ivv[r];
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (DEFER) {                                        // the wave's last (for N <= 256: only) block of this pair: stored next iteration
        pend[0] = oacc[0];
        pend[1] = oacc[1];
        pend_lse = lse_q;
        pend_pair = pair;
        pend_q0 = q0;
      } else {
        if (hl == 0 && q0 + lane < N) lse[((int64_t)b * heads + hh) * N + q0 + lane] = lse_q;
        store_block(o + (int64_t)b * N * C + hh * AT_D, C, q0, N, oacc[0], oacc[1], st_scratch, lane);
      }
    };
    // One query block per wave: the launcher guarantees nblk <= nw (its LDS budget admits at most 7 blocks).  A wave walking
    // several blocks would store through st_scratch while q_dma(nxt) streams the next pair's Q block into it.
    do_block(wv, std::true_type{});
  }
  store_pending();
}

// ------------------------------------------------------------------------------------------ backward: dQ
__global__ __launch_bounds__(512) void attn_bwd_dq_bf16(const bf16_t* __restrict__ d_o, const bf16_t* __restrict__ qkv,
                                                        const bf16_t* __restrict__ o, const float* __restrict__ lse,
                                                        bf16_t* __restrict__ dqkv, float* __restrict__ delta_ws, int N, int heads,
                                                        int nkt, float sc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x / heads, hh = blockIdx.x % heads;
  const int C = heads * AT_D;
  const int64_t rs = 3 * (int64_t)C;
  const bf16_t* base = qkv + (int64_t)b * N * rs + hh * AT_D;
  const int npad = nkt * 32;
  char* kimg = smem;
  char* vimg = smem + npad * 128;
  // K and V images by LDS-DMA (all pieces in flight at once; a register-staged fill loop serialises one HBM round trip per
  // iteration), the wave's own Q / dO / O rows by ordinary loads issued alongside
  dma_image(kimg, base + C, rs, N, npad, wv, nw, lane);
  dma_image(vimg, base + 2 * C, rs, N, npad, wv, nw, lane);
  const int qb = blockIdx.y * nw + wv;
  const bool active = qb * 32 < N;                                  // wave-uniform
  const int q0 = qb * 32, hl = lane >> 5;
  int qrow = q0 + (lane & 31);
  if (qrow >= N) qrow = N - 1;
  const int64_t orow = ((int64_t)b * N + qrow) * C + hh * AT_D;
  bf16x8 qf[4], dof[4], of[4];
  load_row_frags(qf, base + (int64_t)qrow * rs, lane);
  load_row_frags(dof, d_o + orow, lane);
  load_row_frags(of, o + orow, lane);
  const float L = lse[((int64_t)b * heads + hh) * N + qrow];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  f32x16 dq[2];
  zero16(dq[0]);
  zero16(dq[1]);
  if (active) {
  float delta = 0.f;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int j = 0; j < 8; ++j) delta += (float)dof[ks][j] * (float)of[ks][j];
  delta += __shfl_xor(delta, 32, 64);
  if (hl == 0 && q0 + lane < N) delta_ws[((int64_t)b * heads + hh) * N + q0 + lane] = delta;   // re-used by the dK/dV kernel

  RowAddr ra;
  TrAddr ta;
  ra.init(lane);
  ta.init(lane);
  const float c2 = sc * 1.4426950408889634f, L2 = L * 1.4426950408889634f;     // exp(s*sc - L) = exp2(s*c2 - L*log2e)
#pragma unroll 1
  for (int kt = 0; kt < nkt; ++kt) {
    f32x16 st, dp;
    zero16(st);
    zero16(dp);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_rows(kimg, ra, ks, kt), qf[ks], st, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_rows(vimg, ra, ks, kt), dof[ks], dp, 0, 0, 0);
    }
    if (kt == nkt - 1) {                                               // padded keys only in the last tile
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float p = __builtin_amdgcn_exp2f(st[e] * c2 - L2);
        st[e] = kt * 32 + acc_row(e, hl) < N ? p * (dp[e] - delta) * sc : 0.f;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) st[e] = __builtin_amdgcn_exp2f(st[e] * c2 - L2) * (dp[e] - delta) * sc;   // dS^T (scaled)
    }
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      const bf16x8 a = acc_as_a(st, ss);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, rd_tr(kimg, ta, ss, dt, kt), dq[dt], 0, 0, 0);
    }
  }
  // (direct stores: staging them through the images' LDS needs a workgroup barrier first, and waves that finish early then wait
  //  instead of storing under the others' arithmetic -- measured 4 % slower for the two backward kernels together)
  bf16_t* dqb = dqkv + (int64_t)b * N * rs + hh * AT_D;
  store_tile(dqb, rs, q0, N, 0, dq[0], 1.0f, lane);
  store_tile(dqb, rs, q0, N, 32, dq[1], 1.0f, lane);
  }   // active
}

// ------------------------------------------------------------------------------------------ backward: dK, dV
__global__ __launch_bounds__(512) void attn_bwd_dkv_bf16(const bf16_t* __restrict__ d_o, const bf16_t* __restrict__ qkv,
                                                         const float* __restrict__ lse, const float* __restrict__ delta_ws,
                                                         bf16_t* __restrict__ dqkv, int N, int heads, int nqt, float sc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x / heads, hh = blockIdx.x % heads;
  const int C = heads * AT_D;
  const int64_t rs = 3 * (int64_t)C;
  const bf16_t* base = qkv + (int64_t)b * N * rs + hh * AT_D;
  const bf16_t* dob = d_o + (int64_t)b * N * C + hh * AT_D;
  const int npad = nqt * 32;
  char* qimg = smem;
  char* doimg = smem + npad * 128;
  float* lse2_s = reinterpret_cast<float*>(smem + 2 * npad * 128);     // lse * log2(e), query-indexed
  float* del_s = lse2_s + npad;
  dma_image(qimg, base, rs, N, npad, wv, nw, lane);
  dma_image(doimg, dob, C, N, npad, wv, nw, lane);
  for (int r = threadIdx.x; r < npad; r += blockDim.x) {              // per-query statistics (delta comes from the dQ kernel)
    const int64_t si = ((int64_t)b * heads + hh) * N + r;
    lse2_s[r] = r < N ? lse[si] * 1.4426950408889634f : INFINITY;    // exp2(x - inf) = 0 for padded queries
    del_s[r] = r < N ? delta_ws[si] : 0.f;
  }
  const int kb = blockIdx.y * nw + wv;
  const int k0 = kb * 32, hl = lane >> 5;
  int krow = k0 + (lane & 31);
  if (krow >= N) krow = N - 1;
  bf16x8 kf[4], vf[4];
  load_row_frags(kf, base + (int64_t)krow * rs + C, lane);
  load_row_frags(vf, base + (int64_t)krow * rs + 2 * C, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                                   // DMA pieces + the LDS statistics of every wave
  const bool active = kb * 32 < N;                                   // wave-uniform
  f32x16 dk[2], dv[2];
  zero16(dk[0]); zero16(dk[1]); zero16(dv[0]); zero16(dv[1]);
  if (active) {
  RowAddr ra;
  TrAddr ta;
  ra.init(lane);
  ta.init(lane);
  const float c2 = sc * 1.4426950408889634f;
#pragma unroll 1
  for (int qt = 0; qt < nqt; ++qt) {
    f32x16 st, dp;
    zero16(st);
    zero16(dp);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_rows(qimg, ra, ks, qt), kf[ks], st, 0, 0, 0);     // S[q][key]
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_rows(doimg, ra, ks, qt), vf[ks], dp, 0, 0, 0);    // dP[q][key]
    }
    // per-row statistics of the 16 query rows this lane holds: 4 consecutive rows per register group -> one 16-B LDS read
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int q = qt * 32 + 8 * g4 + 4 * hl;
      const float4 l4 = *reinterpret_cast<const float4*>(lse2_s + q);
      const float4 d4 = *reinterpret_cast<const float4*>(del_s + q);
      const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dvv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int e = 4 * g4 + r;
        const float p = __builtin_amdgcn_exp2f(st[e] * c2 - lv[r]);      // lse2 = lse*log2e; +inf for padded queries -> p = 0
        st[e] = p;                                                   // P
        dp[e] = p * (dp[e] - dvv[r]) * sc;                           // dS (scaled)
      }
    }
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      const bf16x8 pa = acc_as_a(st, ss), da = acc_as_a(dp, ss);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, rd_tr(doimg, ta, ss, dt, qt), dv[dt], 0, 0, 0);
        dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, rd_tr(qimg, ta, ss, dt, qt), dk[dt], 0, 0, 0);
      }
    }
  }
  bf16_t* dkb = dqkv + (int64_t)b * N * rs + hh * AT_D + C;
  bf16_t* dvb = dkb + C;
  store_tile(dkb, rs, k0, N, 0, dk[0], 1.0f, lane);
  store_tile(dkb, rs, k0, N, 32, dk[1], 1.0f, lane);
  store_tile(dvb, rs, k0, N, 0, dv[0], 1.0f, lane);
  store_tile(dvb, rs, k0, N, 32, dv[1], 1.0f, lane);
  }   // active
}

// Two 16-byte LDS reads of per-query statistics.  __restrict__ for the reason given at rd_tr_pair: without alias-scope metadata
// hipcc makes these reads wait for every LDS-DMA in flight (here: the next pair's K, V images, requested a moment earlier).
__device__ __forceinline__ void ld_stats(const float* __restrict__ a, const float* __restrict__ b, float4& x, float4& y) {
  x = *reinterpret_cast<const float4*>(a);
  y = *reinterpret_cast<const float4*>(b);
}

// ------------------------------------------------------------------------------------------ backward, one persistent kernel
// dQ, dK and dV of a (batch, head) pair from ONE residency of its data: the two-kernel form above reads K, V images + q, dO, O rows
// (dQ kernel) and then Q, dO images + k, v rows (dKV kernel) -- 950 MB per layer at the hot-path shape, 632 MB if every tensor
// moves once -- and each of its workgroups spends its first third filling LDS with nothing to overlap.  Here one workgroup per CU
// (one wave per 32 tokens, N <= 224) walks the pairs with FOUR images in LDS:
//     phase 1 (dQ)   wave w = query block w.  K, V images; own q, dO, O rows in registers (requested one pair ahead);
//                    delta = rowsum(dO o O) and lse go to LDS for phase 2; then the wave's k, v rows are lifted out of the images.
//     phase 2 (dKV)  wave w = key block w.  Q, dO images + the statistics in LDS; k, v rows from registers.
// and every load runs under the OTHER phase's arithmetic: Q, dO of pair p are requested (LDS-DMA) at the top of phase 1 of p, which
// does not touch those buffers; K, V of pair p + 1 and the rows of p + 1 at the top of phase 2 of p.  Results leave one phase late
// (dQ at the top of phase 2, dK / dV at the top of the next phase 1) through a per-wave LDS staging tile as 128-byte rows: vmcnt
// counts stores too, so a store issued right before a phase boundary's wait would be waited for (attention forward log).
__global__ __launch_bounds__(448) void attn_bwd_fused_bf16(const bf16_t* __restrict__ d_o, const bf16_t* __restrict__ qkv,
                                                           const bf16_t* __restrict__ o, const float* __restrict__ lse,
                                                           bf16_t* __restrict__ dqkv, int N, int heads, int nt, float sc, int npairs) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int C = heads * AT_D, hl = lane >> 5;
  const int64_t rs = 3 * (int64_t)C;
  const int npad = nt * 32, img_bytes = npad * 128;
  char* const kimg = smem;
  char* const vimg = smem + img_bytes;
  char* const qimg = smem + 2 * img_bytes;
  char* const doimg = smem + 3 * img_bytes;
  float* const lse2_s = reinterpret_cast<float*>(smem + 4 * img_bytes);      // lse * log2(e), query-indexed (+inf for padded queries)
  float* const del_s = lse2_s + npad;
  char* const st_scratch = smem + 4 * img_bytes + 2 * npad * (int)sizeof(float) + wv * ST_BYTES;
  int pair = blockIdx.x;
  if (pair >= npairs) return;
  const int t0 = wv * 32;                                                    // this wave's token block (queries in phase 1, keys in phase 2)
  int trow = t0 + (lane & 31);
  trow = trow < N ? trow : N - 1;

  auto qkv_of = [&](int pr) { return qkv + (int64_t)(pr / heads) * N * rs + (pr % heads) * AT_D; };
  auto do_of = [&](int pr) { return (int64_t)(pr / heads) * N * C + (pr % heads) * AT_D; };
  auto issue_kv = [&](int pr) {
    dma_image(kimg, qkv_of(pr) + C, rs, N, npad, wv, nw, lane);
    dma_image(vimg, qkv_of(pr) + 2 * C, rs, N, npad, wv, nw, lane);
  };
  auto issue_qdo = [&](int pr) {
    dma_image(qimg, qkv_of(pr), rs, N, npad, wv, nw, lane);
    dma_image(doimg, d_o + do_of(pr), C, N, npad, wv, nw, lane);
  };
  bf16x8 qf[4], dof[4], of[4];
  float L = 0.f;
  auto load_rows = [&](int pr) {
    load_row_frags(qf, qkv_of(pr) + (int64_t)trow * rs, lane);
    load_row_frags(dof, d_o + do_of(pr) + (int64_t)trow * C, lane);
    load_row_frags(of, o + do_of(pr) + (int64_t)trow * C, lane);
    L = lse[(int64_t)pr * N + trow];                                          // (pair index = b * heads + h)
  };
  RowAddr ra;
  TrAddr ta;
  ra.init(lane);
  ta.init(lane);
  const float c2 = sc * 1.4426950408889634f;
  f32x16 dk[2], dv[2];
  int prev = -1;

  issue_kv(pair);
  load_rows(pair);
  for (; pair < npairs; pair += gridDim.x) {
    // ---------------- top: K, V images and the rows of this pair have landed; everyone has left phase 2 of the previous pair
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // (hipcc does not see through the asm wait: shown a use of the prefetched rows here, it retires them in ITS bookkeeping now and
    //  not with a vmcnt(0) in front of their first arithmetic use -- which would also wait for the Q, dO images requested below)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[ks]), "+v"(dof[ks]), "+v"(of[ks]));
    asm volatile("" : "+v"(L));
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    issue_qdo(pair);                                                          // lands under phase 1
    if (prev >= 0) {                                                          // dK, dV of the previous pair, one phase late
      bf16_t* dkb = dqkv + (int64_t)(prev / heads) * N * rs + (prev % heads) * AT_D + C;
      store_block(dkb, rs, t0, N, dk[0], dk[1], st_scratch, lane);
      store_block(dkb + C, rs, t0, N, dv[0], dv[1], st_scratch, lane);
    }
    // ---------------- phase 1: dQ of query block wv
    float delta = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) delta += (float)dof[ks][j] * (float)of[ks][j];
    delta += __shfl_xor(delta, 32, 64);
    const float L2 = L * 1.4426950408889634f;
    if (hl == 0) {
      const bool real = t0 + lane < N;
      lse2_s[t0 + lane] = real ? L2 : INFINITY;                               // exp2(x - inf) = 0: padded queries vanish in phase 2
      del_s[t0 + lane] = real ? delta : 0.f;
    }
    f32x16 dq[2];
    zero16(dq[0]);
    zero16(dq[1]);
#pragma unroll 1
    for (int kt = 0; kt < nt; ++kt) {
      f32x16 st, dp;
      zero16(st);
      zero16(dp);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_rows(kimg, ra, ks, kt), qf[ks], st, 0, 0, 0);      // S^T[key][q]
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_rows(vimg, ra, ks, kt), dof[ks], dp, 0, 0, 0);     // dP^T[key][q]
      }
      if (kt == nt - 1) {                                                     // padded keys only in the last tile
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float p = __builtin_amdgcn_exp2f(st[e] * c2 - L2);
          st[e] = kt * 32 + acc_row(e, hl) < N ? p * (dp[e] - delta) : 0.f;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) st[e] = __builtin_amdgcn_exp2f(st[e] * c2 - L2) * (dp[e] - delta);   // dS^T, UNSCALED: sc multiplies dQ once, below
      }
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        const bf16x8 a = acc_as_a(st, ss);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
          dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, rd_tr_pair(kimg + ta.off[ss][dt][0] + kt * 4096, kimg + ta.off[ss][dt][1] + kt * 4096),
                                                           dq[dt], 0, 0, 0);
      }
    }
    // softmax scale, once per output element instead of once per score (sc = 1/8 at head size 64: exact, the result is bit for bit the same)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      dq[0][e] *= sc;
      dq[1][e] *= sc;
    }
    // the wave's k, v rows for phase 2 come out of the images (rows >= N repeat row N - 1, like a clamped row load)
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kf[ks] = rd_rows(kimg, ra, ks, wv);
      vf[ks] = rd_rows(vimg, ra, ks, wv);
    }
    // ---------------- middle: Q, dO images landed; everyone has left phase 1 (K, V buffers free, statistics complete)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const int nxt = pair + gridDim.x;
    if (nxt < npairs) {
      issue_kv(nxt);                                                          // land under phase 2
      load_rows(nxt);
    }
    {
      bf16_t* dqb = dqkv + (int64_t)(pair / heads) * N * rs + (pair % heads) * AT_D;
      store_block(dqb, rs, t0, N, dq[0], dq[1], st_scratch, lane);            // dQ, one phase late
    }
    // ---------------- phase 2: dK, dV of key block wv
    zero16(dk[0]); zero16(dk[1]); zero16(dv[0]); zero16(dv[1]);
#pragma unroll 1
    for (int qt = 0; qt < nt; ++qt) {
      f32x16 st, dp;
      zero16(st);
      zero16(dp);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_rows(qimg, ra, ks, qt), kf[ks], st, 0, 0, 0);      // S[q][key]
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rd_rows(doimg, ra, ks, qt), vf[ks], dp, 0, 0, 0);     // dP[q][key]
      }
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int q = qt * 32 + 8 * g4 + 4 * hl;
        float4 l4, d4;
        ld_stats(lse2_s + q, del_s + q, l4, d4);
        const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dvv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int e = 4 * g4 + r;
          const float p = __builtin_amdgcn_exp2f(st[e] * c2 - lv[r]);
          st[e] = p;                                                          // P
          dp[e] = p * (dp[e] - dvv[r]);                                       // dS, unscaled (sc multiplies dK once, below)
        }
      }
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        const bf16x8 pa = acc_as_a(st, ss), da = acc_as_a(dp, ss);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, rd_tr_pair(doimg + ta.off[ss][dt][0] + qt * 4096, doimg + ta.off[ss][dt][1] + qt * 4096),
                                                           dv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, rd_tr_pair(qimg + ta.off[ss][dt][0] + qt * 4096, qimg + ta.off[ss][dt][1] + qt * 4096),
                                                           dk[dt], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      dk[0][e] *= sc;
      dk[1][e] *= sc;
    }
    prev = pair;
  }
  {
    bf16_t* dkb = dqkv + (int64_t)(prev / heads) * N * rs + (prev % heads) * AT_D + C;
    store_block(dkb, rs, t0, N, dk[0], dk[1], st_scratch, lane);
    store_block(dkb + C, rs, t0, N, dv[0], dv[1], st_scratch, lane);
  }
}

// ------------------------------------------------------------------------------------------ qkv projection + attention, one kernel
// north_star's "fused QKV projection + multi-head attention" as ONE launch: for the passes that keep nothing for a backward (the teacher
// of a training step, encode()) the packed qkv tensor -- 237 MB per layer at the hot-path shape, written by one launch and read straight
// back by the next -- never exists.  One workgroup per CU walks (image, head) pairs, one wave per 32 tokens (193..224 tokens):
//   projection   [q | k | v](32 tokens x 192) of the wave's tokens = LN(x)[32 x D] . W_h^T + b_h, as six 32 x 32 accumulator blocks computed
//                TRANSPOSED (weights as the A operand: a lane holds four consecutive features of ONE token, 8 bytes of an image row);
//                K-steps of 32: the head's weight slice [192][32] (12 KiB, shared) and the wave's own token rows [32][32] (2 KiB) arrive
//                by LDS-DMA in a two-slot ring, one barrier per step;
//   hand-over    q -> the wave's 32-row image in its staging tile, k / v -> rows 32 w .. of the K / V images, in the layout the
//                attention kernels read (img_off); optionally the packed qkv rows leave for HBM too (qkv_out != null);
//   attention    exactly attn_fwd_bf16_persist's one-pass block (S^T strip in registers, lane-local softmax, P as the next A operand).
// The pairs of one image are kept on one XCD (its six heads run side by side: the token rows come out of that XCD's L2, the weights
// -- 0.9 MB for all heads -- stay there).
// MEASURED (tools/qkv_attn_bench.py): correct, and NOT faster than the two launches it replaces -- 219-227 us against
// 208-222 us at (512, 201, 6, 384), 500 against 408 us at ViT-L's (256, 201, 16, 1024).  The projection of a pair streams
// D x (192 + 224) x 2 B through L2 -> LDS with a 224 x 192 tile's reuse (the stand-alone product runs 256 x 256 tiles in a ping-pong
// schedule), every pair starts its ring cold (~3 us: s_memtime stamps) and the hand-over costs ~2 us; a four-stage ring with the token
// rows prefetched into registers three steps ahead was built and measured no faster (227 us): the steps are bound by the ~2.5 us
// issue-to-landed latency of the staging loads divided by the ring depth.  The kernel therefore stays OPT-IN (DINOX_QKV_FUSED=1).
constexpr int QA_WSLOT = 192 * 64, QA_XSLOT = 32 * 64;          // [192 features][32 k], [32 tokens][32 k] bf16 (64-byte rows)

template <int NKT>
__global__ __launch_bounds__(NKT * 64) void attn_qkv_fused_fwd(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                              const float* __restrict__ bias, bf16_t* __restrict__ o,
                                                              bf16_t* __restrict__ qkv_out, float* __restrict__ lse, int B, int N, int heads,
                                                              int D, float sc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int npad = NKT * 32, img_bytes = npad * 128;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int C = heads * AT_D, hl = lane >> 5;
  char* const kimg = smem;
  char* const vimg = smem + img_bytes;
  char* const wring = smem + 2 * img_bytes;
  char* const xring = wring + 2 * QA_WSLOT + wv * 2 * QA_XSLOT;
  char* const st_scratch = wring + 2 * QA_WSLOT + NKT * 2 * QA_XSLOT + wv * ST_BYTES;
  float* const inv_s = reinterpret_cast<float*>(wring + 2 * QA_WSLOT + NKT * 2 * QA_XSLOT + NKT * ST_BYTES) + wv * 32;
  float* const bias_s = reinterpret_cast<float*>(wring + 2 * QA_WSLOT + NKT * 2 * QA_XSLOT + NKT * ST_BYTES) + NKT * 32;
  const int nk = D / 32;

  // pairs of this workgroup: image b lives on XCD b % 8 (consecutive workgroup ids go round the XCDs), its heads on neighbouring CUs;
  // a handful of images (encode() of one slice) are dealt out pair by pair instead
  const int nx = B >= 64 ? 8 : 1;
  const int xcd = nx == 8 ? (int)(blockIdx.x & 7) : 0, widx = nx == 8 ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int nwx = nx == 8 ? (int)((gridDim.x + 7 - xcd) >> 3) : (int)gridDim.x;                      // workgroups that share this list
  const int nimg = nx == 8 ? (B > xcd ? (B - xcd + 7) >> 3 : 0) : B;                                 // images on the list
  const int npx = nimg * heads;

  // staging addresses: instruction i of a slab covers rows 16 i .. 16 i + 15 (64 B each), chunk c of row r from source chunk c ^ ((r >> 2) & 3)
  const int srow = lane >> 2, schunk = lane & 3;
  unsigned xoff[2];
  int64_t woff[2];
  bool wmine[2];
  f32x16 pend[2];
  float pend_lse = 0.f;
  int pend_b = -1, pend_h = 0;                          // wave-uniform
  auto store_pending = [&]() {
    if (lse != nullptr && hl == 0 && wv * 32 + lane < N) lse[((int64_t)pend_b * heads + pend_h) * N + wv * 32 + lane] = pend_lse;
    store_block(o + (int64_t)pend_b * N * C + pend_h * AT_D, C, wv * 32, N, pend[0], pend[1], st_scratch, lane);
  };

  for (int j = widx; j < npx; j += nwx) {
    const int b = nx * (j / heads) + xcd, hh = j % heads;
    const bf16_t* const xb = x + (int64_t)b * N * D;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int r = 16 * q + srow;                        // token row of the wave's block
      int tok = wv * 32 + r;
      tok = tok < N ? tok : N - 1;                        // rows past N repeat the last token (finite; masked as keys, never stored as queries)
      xoff[q] = (unsigned)((tok * D + (schunk ^ ((r >> 2) & 3)) * 8) * 2);
      const int i = wv + NKT * q, f = 16 * i + srow;      // weight slab instruction i (12 of them), feature f of [q | k | v] of this head
      wmine[q] = i < 12;
      const int fc = f < 192 ? f : 0;
      woff[q] = ((int64_t)((fc >> 6) * C + hh * AT_D + (fc & 63)) * D + (schunk ^ ((fc >> 2) & 3)) * 8) * 2;
    }
    auto stage = [&](int slot, int kt) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (wmine[q])
          __builtin_amdgcn_global_load_lds((at_gbl_void*)((const char*)w + woff[q] + kt * 64), (at_lds_void*)(wring + slot * QA_WSLOT + (wv + NKT * q) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((at_gbl_void*)((const char*)xb + xoff[q] + kt * 64), (at_lds_void*)(xring + slot * QA_XSLOT + q * 1024), 16, 0, 0);
      }
    };
    if (threadIdx.x < 192) {
      const int f = threadIdx.x;
      bias_s[f] = bias ? bias[(f >> 6) * C + hh * AT_D + (f & 63)] : 0.f;     // (read after the K loop's barriers)
    }
    f32x16 acc[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) zero16(acc[i]);
    stage(0, 0);
    const int frow = lane & 31;
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
      if (kt == 1 && pend_b >= 0) store_pending();        // the previous pair's output, one pair late (its stores drain under the K loop)
      const char* ws = wring + (kt & 1) * QA_WSLOT;
      const char* xs = xring + (kt & 1) * QA_XSLOT;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int kc = 2 * ks + hl;
        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xs + frow * 64 + ((kc ^ ((frow >> 2) & 3)) << 4));
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const int rw = i * 32 + frow;
          const bf16x8 wf = *reinterpret_cast<const bf16x8*>(ws + rw * 64 + ((kc ^ ((rw >> 2) & 3)) << 4));
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, acc[i], 0, 0, 0);       // [features][tokens]
        }
      }
    }
    if (nk < 2 && pend_b >= 0) store_pending();
    // ---- hand-over: acc[i][4 g + r] = feature 32 (i & 1) + 8 g + 4 hl + r of section i >> 1, token frow of the wave's block
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      char* const img = i < 2 ? st_scratch : (i < 4 ? kimg : vimg);
      const int row = i < 2 ? frow : wv * 32 + frow;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b4 = *reinterpret_cast<const float4*>(bias_s + (i >> 1) * 64 + (i & 1) * 32 + 8 * g + 4 * hl);
        uint2 pk;
        pk.x = (unsigned)f32_to_bf16(acc[i][4 * g] + b4.x) | ((unsigned)f32_to_bf16(acc[i][4 * g + 1] + b4.y) << 16);
        pk.y = (unsigned)f32_to_bf16(acc[i][4 * g + 2] + b4.z) | ((unsigned)f32_to_bf16(acc[i][4 * g + 3] + b4.w) << 16);
        *reinterpret_cast<uint2*>(img + img_off(row, 4 * (i & 1) + g) + 8 * hl) = pk;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                         // every wave's k, v rows are in the images
    __builtin_amdgcn_sched_barrier(0);
    if (qkv_out != nullptr) {                             // the packed rows, for a caller that keeps them (whole 128-byte row pieces)
#pragma unroll
      for (int sec = 0; sec < 3; ++sec) {
        const char* img = sec == 0 ? st_scratch : (sec == 1 ? kimg : vimg);
        const int rbase = sec == 0 ? 0 : wv * 32;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int r = (lane >> 3) + 8 * t, c = lane & 7, tok = wv * 32 + r;
          const uint4 v = *reinterpret_cast<const uint4*>(img + img_off(rbase + r, c));
          if (tok < N) *reinterpret_cast<uint4*>(qkv_out + ((int64_t)b * N + tok) * 3 * C + sec * C + hh * AT_D + c * 8) = v;
        }
      }
    }
    // ---- attention of the wave's 32 queries (attn_fwd_bf16_persist, one pass)
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(st_scratch + img_off(frow, 2 * ks + hl));
    const char* kr[4];
    const char* vt[2][2][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kr[ks] = kimg + img_off(frow, 2 * ks + hl);
    {
      const int i = lane & 15, g = lane >> 4, q4 = i >> 2, pp = i & 3;
#pragma unroll
      for (int ss = 0; ss < 2; ++ss)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const int ch = (dt * 32 >> 3) + 2 * (g & 1) + (pp >> 1);
          const int r0 = 16 * ss + 4 * (g >> 1) + q4;
          vt[ss][dt][0] = vimg + img_off(r0, ch) + 8 * (pp & 1);
          vt[ss][dt][1] = vimg + img_off(r0 + 8, ch) + 8 * (pp & 1);
        }
    }
    const float c2 = sc * 1.4426950408889634f;
    f32x16 oacc[2];
    zero16(oacc[0]);
    zero16(oacc[1]);
    float sum = 0.f, mx = -INFINITY;
    {
      f32x16 st[NKT];
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        zero16(st[kt]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(kr[ks] + kt * 4096), qf[ks], st[kt], 0, 0, 0);
      }
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          if (kt == NKT - 1 && !(kt * 32 + acc_row(e, hl) < N)) st[kt][e] = -INFINITY;
          mx = fmaxf(mx, st[kt][e]);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mx2 = mx * c2;
      mx *= sc;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float pr = __builtin_amdgcn_exp2f(st[kt][e] * c2 - mx2);
          st[kt][e] = pr;
          sum += pr;
        }
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
          const bf16x8 pa = acc_as_a(st[kt], ss);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt)
            oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, rd_tr_pair(vt[ss][dt][0] + kt * 4096, vt[ss][dt][1] + kt * 4096), oacc[dt], 0, 0, 0);
        }
      }
    }
    sum += __shfl_xor(sum, 32, 64);
    if (hl == 0) inv_s[lane] = 1.0f / sum;
    pend_lse = mx + __logf(sum);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 iv = *reinterpret_cast<const float4*>(inv_s + 8 * g + 4 * hl);
      const float ivv[4] = {iv.x, iv.y, iv.z, iv.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        oacc[0][4 * g + r] *= ivv[r];
        oacc[1][4 * g + r] *= ivv[r];
      }
    }
    __builtin_amdgcn_wave_barrier();
    pend[0] = oacc[0];
    pend[1] = oacc[1];
    pend_b = b;
    pend_h = hh;
  }
  if (pend_b >= 0) store_pending();
}

// ------------------------------------------------------------------------------------------ launchers
// Workgroups of a persistent attention kernel that fit one CU: 160 KiB of LDS, and 8 waves (the kernels sit at ~250 registers: two per SIMD).
static int persist_wgs_per_cu(size_t lds_bytes, int waves) {
  const int by_lds = (int)((size_t)160 * 1024 / (lds_bytes ? lds_bytes : 1)), by_waves = 8 / (waves > 0 ? waves : 1);
  const int n = by_lds < by_waves ? by_lds : by_waves;
  return n < 1 ? 1 : n;
}
static void geometry(int N, int& nblk, int& nwg, int& waves) {
  nblk = (N + 31) / 32;
  nwg = (nblk + 7) / 8;
  waves = (nblk + nwg - 1) / nwg;
}

template <typename K>
static int allow_lds(K kern, size_t bytes) {
  return reserve_lds(reinterpret_cast<const void*>(kern), bytes, "attention");
}

int launch_attention_bf16_fwd(const void* qkv, void* o, float* lse, int B, int N, int heads, int d, hipStream_t st) {
  if (d != AT_D || N > 288 || ((uintptr_t)qkv & 15) || ((uintptr_t)o & 15)) return DINOX_EUNSUPPORTED;
  int nblk, nwg, waves;
  geometry(N, nblk, nwg, waves);
  const float sc = 1.0f / sqrtf((float)d);
  {
    const int nw = nblk < 8 ? nblk : 8;
    const size_t lds = (size_t)4 * nblk * 32 * 128 + 8 * 32 * sizeof(float) + (size_t)nw * ST_BYTES;    // images, 1/rowsum, output staging (per wave)
    static const bool off = getenv("DINOX_ATTN_NO_PERSIST") != nullptr;
    if (lds <= 160 * 1024 && !off && nblk <= nw) {          // one query block per wave (the kernel relies on it)
      const int npairs = B * heads;
      // resident workgroups: one per CU at 7 blocks (201 tokens); short sequences (the 41-token local crops: two waves, 41 KiB) get as many as
      // LDS and two waves per SIMD allow -- with one the chip ran half a wave per SIMD (113 us for 5 GFLOP at N = 41)
      const int per_cu = persist_wgs_per_cu(lds, nw);
      const int nwgp = npairs < 256 * per_cu ? npairs : 256 * per_cu;
#define PFWD(NKT)                                                                                                                  \
  do {                                                                                                                             \
    if (int rc = allow_lds(attn_fwd_bf16_persist<NKT>, lds)) return fail(rc, "attention_fwd: cannot reserve %zu B of LDS", lds);   \
    hipLaunchKernelGGL((attn_fwd_bf16_persist<NKT>), dim3(nwgp), dim3(nw * 64), lds, st, (const bf16_t*)qkv, (bf16_t*)o, lse, N,    \
                       heads, nblk, sc, npairs);                                                                                   \
  } while (0)
      if (nblk == 7) PFWD(7); else if (nblk == 1) PFWD(1); else if (nblk == 2) PFWD(2); else PFWD(0);   // 3..6 blocks: two-pass form
#undef PFWD
      return check_launch("attention_bf16_fwd_persist");
    }
  }
  dim3 grid((unsigned)(B * heads), (unsigned)nwg), block((unsigned)(waves * 64));
#define FWD(NKT)                                                                                                   \
  do {                                                                                                             \
    const size_t lds = (size_t)2 * NKT * 32 * 128 + 8 * 32 * sizeof(float);                                        \
    if (int rc = allow_lds(attn_fwd_bf16<NKT>, lds)) return fail(rc, "attention_fwd: cannot reserve %zu B of LDS", lds); \
    hipLaunchKernelGGL((attn_fwd_bf16<NKT>), grid, block, lds, st, (const bf16_t*)qkv, (bf16_t*)o, lse, N, heads, sc); \
  } while (0)
  if (nblk <= 2) FWD(2); else if (nblk <= 4) FWD(4); else if (nblk <= 7) FWD(7); else FWD(9);
#undef FWD
  return check_launch("attention_bf16_fwd");
}

// x [B N][D] bf16 (LayerNorm output), w [3 C][D] bf16 (C = heads * 64), bias [3 C] fp32 or null -> o [B N][C] bf16 (+ qkv_out [B N][3 C], lse)
bool attention_qkv_fused_ok(int B, int N, int heads, int d, int D) {
  return d == AT_D && N > 192 && N <= 224 && D >= 32 && D % 32 == 0 && B > 0 && heads > 0 && (int64_t)N * D * 2 < ((int64_t)1 << 31);
}
int launch_attention_qkv_fused_fwd(const void* x, const void* w, const float* bias, void* o, void* qkv_out, float* lse, int B, int N, int heads,
                                   int d, int D, hipStream_t st) {
  if (!attention_qkv_fused_ok(B, N, heads, d, D)) return DINOX_EUNSUPPORTED;
  if ((((uintptr_t)x | (uintptr_t)w | (uintptr_t)o | (uintptr_t)qkv_out | (uintptr_t)bias) & 15) != 0) return DINOX_EALIGN;
  constexpr int NKT = 7;
  const size_t lds = (size_t)2 * NKT * 32 * 128 + 2 * QA_WSLOT + (size_t)NKT * 2 * QA_XSLOT + (size_t)NKT * ST_BYTES + NKT * 32 * sizeof(float) + 192 * sizeof(float);
  if (int rc = allow_lds(attn_qkv_fused_fwd<NKT>, lds)) return fail(rc, "qkv_attention_fwd: cannot reserve %zu B of LDS", lds);
  const int64_t npairs = (int64_t)B * heads;
  const int nwg = npairs < 256 ? (int)npairs : 256;
  hipLaunchKernelGGL((attn_qkv_fused_fwd<NKT>), dim3(nwg), dim3(NKT * 64), lds, st, (const bf16_t*)x, (const bf16_t*)w, bias, (bf16_t*)o,
                     (bf16_t*)qkv_out, lse, B, N, heads, D, 1.0f / sqrtf((float)d));
  return check_launch("qkv_attention_fwd");
}

int launch_attention_bf16_bwd(const void* d_o, const void* qkv, const void* o, const float* lse, void* dqkv, float* ws, int B,
                              int N, int heads, int d, hipStream_t st) {
  if (!ws) return DINOX_EUNSUPPORTED;
  if (d != AT_D || N > 544 || ((uintptr_t)qkv & 15) || ((uintptr_t)o & 15) || ((uintptr_t)d_o & 15) || ((uintptr_t)dqkv & 15)) return DINOX_EUNSUPPORTED;
  int nblk, nwg, waves;
  geometry(N, nblk, nwg, waves);
  const float sc = 1.0f / sqrtf((float)d);
  {
    // one persistent kernel (N <= 224): every tensor moves once, loads run under the other phase's arithmetic
    const size_t ldsf = (size_t)4 * nblk * 32 * 128 + 2 * (size_t)nblk * 32 * sizeof(float) + (size_t)nblk * ST_BYTES;
    static const bool split = getenv("DINOX_ATTN_BWD_SPLIT") != nullptr;
    if (nblk <= 7 && ldsf <= 160 * 1024 && !split) {
      const int npairs = B * heads;
      const int per_cu = persist_wgs_per_cu(ldsf, nblk);
      const int nwgp = npairs < 256 * per_cu ? npairs : 256 * per_cu;
      if (int rc = allow_lds(attn_bwd_fused_bf16, ldsf)) return fail(rc, "attention_bwd: cannot reserve %zu B of LDS", ldsf);
      hipLaunchKernelGGL(attn_bwd_fused_bf16, dim3(nwgp), dim3(nblk * 64), ldsf, st, (const bf16_t*)d_o, (const bf16_t*)qkv, (const bf16_t*)o, lse,
                         (bf16_t*)dqkv, N, heads, nblk, sc, npairs);
      return check_launch("attention_bf16_bwd_fused");
    }
  }
  dim3 grid((unsigned)(B * heads), (unsigned)nwg), block((unsigned)(waves * 64));
  const size_t lds1 = (size_t)2 * nblk * 32 * 128;
  const size_t lds2 = lds1 + (size_t)2 * nblk * 32 * sizeof(float);
  if (int rc = allow_lds(attn_bwd_dq_bf16, lds1)) return fail(rc, "attention_bwd: cannot reserve %zu B of LDS", lds1);
  if (int rc = allow_lds(attn_bwd_dkv_bf16, lds2)) return fail(rc, "attention_bwd: cannot reserve %zu B of LDS", lds2);
  hipLaunchKernelGGL(attn_bwd_dq_bf16, grid, block, lds1, st, (const bf16_t*)d_o, (const bf16_t*)qkv, (const bf16_t*)o, lse,
                     (bf16_t*)dqkv, ws, N, heads, nblk, sc);
  hipLaunchKernelGGL(attn_bwd_dkv_bf16, grid, block, lds2, st, (const bf16_t*)d_o, (const bf16_t*)qkv, lse, (const float*)ws,
                     (bf16_t*)dqkv, N, heads, nblk, sc);
  return check_launch("attention_bf16_bwd");
}

}  // namespace dinox
