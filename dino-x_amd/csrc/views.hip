// views.hip -- device-side view pipeline for the 2.5D slice stacks: stored u16 -> HU -> random window -> RandomResizedCrop
// (antialiased bicubic) -> horizontal flip -> ImageNet normalise, in ONE kernel writing the fp32 [V][3][S][S] batch the
// patch-embed reads.  Replaces PngDataset._load_hu01 + the torchvision transform stack of the reference
// (scripts/phase5_big_run.py:493-497, 516-528, 549-555) for everything after the PNG decode; the random draws (window
// level/width, crop box, flip) stay on the host and arrive as a per-view table.
//
// The resize is PyTorch's _upsample_bicubic2d_aa (what torchvision's tensor resize calls, antialias=True, align_corners=False):
// separable; per output index i: centre = scale (i + 0.5), support = 2 max(scale, 1), taps [int(centre - support + .5),
// int(centre + support + .5)) clipped to the crop, weights = Keys cubic (a = -0.5) of (tap + .5 - centre) / max(scale, 1),
// renormalised to sum 1; horizontal pass first, then vertical, fp32 throughout.
//
// One workgroup per 16x16 output tile of one channel of one view.  The windowed source footprint of the tile (at most
// 16 scale + 2 support + 2 pixels a side) is staged in LDS once, filtered horizontally into a [rows][16] strip, then vertically.
// HBM-bound by design: every source pixel of the crop is read ~(1 + 2 support / (16 scale))^2 times (1.6x at scale 2.3), from L2
// after the first touch; the output is written once, coalesced.
#include "common.h"

namespace dinox {

constexpr int VW_T = 16;          // output tile side
constexpr int VW_THREADS = 256;

__device__ __forceinline__ float vw_cubic(float x) {
  const float a = -0.5f;
  x = fabsf(x);
  if (x < 1.0f) return ((a + 2.0f) * x - (a + 3.0f)) * x * x + 1.0f;
  if (x < 2.0f) return (((x - 5.0f) * x + 8.0f) * x - 4.0f) * a;
  return 0.0f;
}

// taps of output index i along one axis (in: crop extent, out: S)
__device__ __forceinline__ void vw_taps(int i, float scale, float support, int in_size, int& xmin, int& xsize, float& center) {
  center = scale * ((float)i + 0.5f);
  xmin = max((int)(center - support + 0.5f), 0);
  xsize = min((int)(center + support + 0.5f), in_size) - xmin;
}

// Where output pixel (c, oy, ox) of view v goes.  LAYOUT 0: the fp32 image batch [V][3][S][S].  LAYOUT 1 / 2: the patch-embed
// operand itself, u[(v P + (oy / p) g + ox / p)][c p^2 + (oy % p) p + ox % p] with g = S / p, row stride ld, in bf16 (1) or fp32 (2) --
// what dinox_patch_unfold would make of the image batch (tokens.hip), without the 4 B/pixel image round trip through HBM.  With
// p = 16 a tile is one channel of one patch: 256 consecutive operand elements per workgroup.
template <int LAYOUT>
struct vw_out {
  void* base; int S, p, g, ld;
  __device__ __forceinline__ void put(int v, int c, int oy, int ox, float val) const {
    if (LAYOUT == 0) {
      ((float*)base)[(((int64_t)v * 3 + c) * S + oy) * (int64_t)S + ox] = val;
    } else {
      const int gy = oy / p, gx = ox / p;
      const int64_t o = ((int64_t)v * g * g + gy * g + gx) * (int64_t)ld + (c * p + (oy - gy * p)) * p + (ox - gx * p);
      if (LAYOUT == 1) ((bf16_t*)base)[o] = f32_to_bf16(val);
      else ((float*)base)[o] = val;
    }
  }
};

// vi[v] = {element offset of the stack in raw, H, W, top, left, h, w, flip};  vf[v] = {wmin, wden}
template <int LAYOUT>
__global__ __launch_bounds__(VW_THREADS) void slice_views_kernel(const unsigned short* __restrict__ raw, const int64_t* __restrict__ vi,
                                                               const float* __restrict__ vf, vw_out<LAYOUT> out, int S,
                                                               int tiles, int F, int MAXT) {
  extern __shared__ float vw_smem[];
  float* src = vw_smem;                  // [F][F]   windowed footprint
  float* strip = src + F * F;            // [F][16]  after the horizontal pass
  float* wx = strip + F * VW_T;          // [16][MAXT]
  float* wy = wx + VW_T * MAXT;          // [16][MAXT]
  __shared__ int s_min[2][VW_T], s_n[2][VW_T];

  const int v = blockIdx.z, c = blockIdx.y;
  const int ty = blockIdx.x / tiles, tx = blockIdx.x % tiles;
  const int64_t* p = vi + (int64_t)v * 8;
  const int64_t off = p[0];
  const int H = (int)p[1], W = (int)p[2], top = (int)p[3], left = (int)p[4], ch = (int)p[5], cw = (int)p[6];
  const bool flip = p[7] != 0;
  const float wmin = vf[2 * v], wden = vf[2 * v + 1];
  const float mean = c == 0 ? 0.485f : (c == 1 ? 0.456f : 0.406f);
  const float stdv = c == 0 ? 0.229f : (c == 1 ? 0.224f : 0.225f);

  const float sx = (float)cw / (float)S, sy = (float)ch / (float)S;
  const float supx = sx >= 1.0f ? 2.0f * sx : 2.0f, supy = sy >= 1.0f ? 2.0f * sy : 2.0f;
  const float invx = sx >= 1.0f ? 1.0f / sx : 1.0f, invy = sy >= 1.0f ? 1.0f / sy : 1.0f;
  const int ox0 = tx * VW_T, oy0 = ty * VW_T;

  // per output column / row of the tile: tap range and normalised weights.  Column q of the tile is OUTPUT column ox0+q, which
  // shows source-space column i = flip ? S-1-(ox0+q) : ox0+q.
  const int t = threadIdx.x;
  if (t < 2 * VW_T) {
    const int axis = t / VW_T, q = t % VW_T;
    const int o = (axis == 0 ? ox0 : oy0) + q;
    int xmin = 0, xn = 0;
    float center = 0.f;
    float* w = (axis == 0 ? wx : wy) + q * MAXT;
    if (o < S) {
      const int i = (axis == 0 && flip) ? S - 1 - o : o;
      vw_taps(i, axis == 0 ? sx : sy, axis == 0 ? supx : supy, axis == 0 ? cw : ch, xmin, xn, center);
      if (xn > MAXT) xn = -1;                                  // table too small for this crop: poison the tile (host sizes it; never expected)
      float tot = 0.f;
      for (int k = 0; k < xn; ++k) {
        const float wk = vw_cubic(((float)(k + xmin) - center + 0.5f) * (axis == 0 ? invx : invy));
        w[k] = wk;
        tot += wk;
      }
      for (int k = 0; k < xn; ++k) w[k] = w[k] / tot;
    }
    s_min[axis][q] = xmin;
    s_n[axis][q] = xn;
  }
  __syncthreads();
  // footprint of the tile in crop coordinates
  int fx0 = 0x7fffffff, fx1 = 0, fy0 = 0x7fffffff, fy1 = 0;
  bool bad = false;
#pragma unroll
  for (int q = 0; q < VW_T; ++q) {
    if (ox0 + q < S) {
      bad |= s_n[0][q] < 0;
      fx0 = min(fx0, s_min[0][q]);
      fx1 = max(fx1, s_min[0][q] + s_n[0][q]);
    }
    if (oy0 + q < S) {
      bad |= s_n[1][q] < 0;
      fy0 = min(fy0, s_min[1][q]);
      fy1 = max(fy1, s_min[1][q] + s_n[1][q]);
    }
  }
  const int fw = fx1 - fx0, fh = fy1 - fy0;
  bad |= fw > F || fh > F;
  if (bad) {                                                   // never silently wrong
    const int oy = oy0 + t / VW_T, ox = ox0 + t % VW_T;
    if (oy < S && ox < S) out.put(v, c, oy, ox, __builtin_nanf(""));
    return;
  }
  // stage the windowed footprint (HU decode + window, scripts/phase5_big_run.py:519-526)
  const unsigned short* plane = raw + off + (int64_t)c * H * W;
  for (int e = t; e < fh * fw; e += VW_THREADS) {
    const int ry = e / fw, rx = e % fw;
    const float u = (float)plane[(int64_t)(top + fy0 + ry) * W + (left + fx0 + rx)];
    const float hu = (u - 32768.0f) * 0.1f;
    src[ry * F + rx] = fminf(fmaxf((hu - wmin) / wden, 0.0f), 1.0f);
  }
  __syncthreads();
  // horizontal pass: strip[ry][q] = sum_k wx[q][k] * src[ry][xmin_q - fx0 + k]
  for (int e = t; e < fh * VW_T; e += VW_THREADS) {
    const int ry = e / VW_T, q = e % VW_T;
    float a = 0.f;
    if (ox0 + q < S) {
      const float* s = src + ry * F + (s_min[0][q] - fx0);
      const float* w = wx + q * MAXT;
      const int n = s_n[0][q];
      for (int k = 0; k < n; ++k) a += w[k] * s[k];
    }
    strip[ry * VW_T + q] = a;
  }
  __syncthreads();
  // vertical pass + normalise
  {
    const int qy = t / VW_T, qx = t % VW_T;
    const int oy = oy0 + qy, ox = ox0 + qx;
    if (oy < S && ox < S) {
      const float* w = wy + qy * MAXT;
      const int n = s_n[1][qy], r0 = s_min[1][qy] - fy0;
      float a = 0.f;
      for (int k = 0; k < n; ++k) a += w[k] * strip[(r0 + k) * VW_T + qx];
      out.put(v, c, oy, ox, (a - mean) / stdv);
    }
  }
}

}  // namespace dinox

using namespace dinox;

extern "C" int64_t dinox_slice_views_lds_bytes(int S, int max_crop) {
  if (S <= 0 || max_crop <= 0) return -1;
  const double s = (double)max_crop / (double)S, sup = 2.0 * (s > 1.0 ? s : 1.0);
  const int64_t F = (int64_t)(VW_T * s + 2.0 * sup) + 4, MAXT = (int64_t)(2.0 * sup) + 3;
  return (F * F + F * VW_T + 2 * VW_T * MAXT) * (int64_t)sizeof(float);
}

template <int LAYOUT>
static int launch_slice_views(const void* raw_u16, const int64_t* view_i, const float* view_f, vw_out<LAYOUT> out, int V, int S, int max_crop,
                              void* stream, const char* what) {
  const double s = (double)max_crop / (double)S, sup = 2.0 * (s > 1.0 ? s : 1.0);
  const int F = (int)(VW_T * s + 2.0 * sup) + 4, MAXT = (int)(2.0 * sup) + 3;
  const size_t lds = (size_t)dinox_slice_views_lds_bytes(S, max_crop);
  DX_REQUIRE(lds <= 150 * 1024, DINOX_EUNSUPPORTED, "%s: a %d-pixel crop down to %d needs %zu B of LDS (limit 150 KiB)", what, max_crop, S, lds);
  if (int rc = reserve_lds(reinterpret_cast<const void*>(slice_views_kernel<LAYOUT>), lds, what)) return rc;
  const int tiles = (S + VW_T - 1) / VW_T;
  hipLaunchKernelGGL(slice_views_kernel<LAYOUT>, dim3((unsigned)(tiles * tiles), 3, (unsigned)V), dim3(VW_THREADS), lds, as_stream(stream),
                     (const unsigned short*)raw_u16, view_i, view_f, out, S, tiles, F, MAXT);
  return check_launch(what);
}

extern "C" int dinox_slice_views(const void* raw_u16, const int64_t* view_i, const float* view_f, float* out, int V, int S, int max_crop,
                                 void* stream) {
  DX_REQUIRE(raw_u16 && view_i && view_f && out, DINOX_EINVAL, "slice_views: null pointer");
  DX_REQUIRE(V > 0 && V <= 65535 && S > 0 && max_crop > 0, DINOX_EINVAL, "slice_views: V=%d S=%d max_crop=%d", V, S, max_crop);
  return launch_slice_views<0>(raw_u16, view_i, view_f, vw_out<0>{out, S, 1, S, 0}, V, S, max_crop, stream, "slice_views");
}

extern "C" int dinox_slice_views_patches(const void* raw_u16, const int64_t* view_i, const float* view_f, void* u, int V, int S, int max_crop,
                                         int patch, int ld, int out_dtype, void* stream) {
  DX_REQUIRE(raw_u16 && view_i && view_f && u, DINOX_EINVAL, "slice_views_patches: null pointer");
  DX_REQUIRE(V > 0 && V <= 65535 && S > 0 && max_crop > 0, DINOX_EINVAL, "slice_views_patches: V=%d S=%d max_crop=%d", V, S, max_crop);
  DX_REQUIRE(patch > 0 && S % patch == 0 && ld >= 3 * patch * patch, DINOX_EINVAL, "slice_views_patches: S=%d patch=%d ld=%d", S, patch, ld);
  DX_REQUIRE(out_dtype == DINOX_F32 || out_dtype == DINOX_BF16, DINOX_EINVAL, "slice_views_patches: dtype %d", out_dtype);
  const int g = S / patch;
  if (ld > 3 * patch * patch) {          // padded operand (patch 14 in bf16: 588 -> 640 columns): the tail columns are zeros
    const size_t bytes = (size_t)V * g * g * ld * (out_dtype == DINOX_BF16 ? 2 : 4);
    DX_REQUIRE(hipMemsetAsync(u, 0, bytes, as_stream(stream)) == hipSuccess, DINOX_EINVAL, "slice_views_patches: memset failed");
  }
  if (out_dtype == DINOX_BF16)
    return launch_slice_views<1>(raw_u16, view_i, view_f, vw_out<1>{u, S, patch, g, ld}, V, S, max_crop, stream, "slice_views_patches");
  return launch_slice_views<2>(raw_u16, view_i, view_f, vw_out<2>{u, S, patch, g, ld}, V, S, max_crop, stream, "slice_views_patches");
}
