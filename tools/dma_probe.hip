// dma_probe.hip -- what does one CU sustain from memory into LDS, as a function of how many bytes it keeps in flight?
//
// Every GEMM on the hot path is fed by LDS-DMA (global_load_lds_dwordx4) through a ring of LDS slots, and DESIGN.md section 4 models
// their K loops as "bytes in flight / round-trip latency".  This probe measures that model directly, without any arithmetic: each
// workgroup (256 threads) streams its share of a buffer through a ring of DEPTH slots of SLOT bytes with the same counted-vmcnt +
// barrier structure as the kernels (DEPTH - 1 slots in flight while one is "consumed"), for
//   * 1, 2 or 3 workgroups per CU (the LDS it allocates decides),
//   * a source that streams from HBM (every byte touched once) or that stays in L2 (all workgroups of an XCD re-read 2 MiB).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/dma_probe tools/dma_probe.hip     Run (GPU box): tools/dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

#define CK(x)                                                                            \
  do {                                                                                   \
    hipError_t e_ = (x);                                                                 \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
  } while (0)

// One step = SLOT bytes = SLOT / 4096 DMA instructions per wave (4 waves x 64 lanes x 16 B = 4 KiB per instruction round).
template <int DEPTH, int SLOT>
__global__ __launch_bounds__(256) void probe(const char* __restrict__ src, long bytes_per_wg, long wrap, int share, int* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int NI = SLOT / 4096;                       // instructions per wave per step
  // `share` consecutive workgroups (same XCD after the 8-way round-robin: ids b, b + 8, ...) read the SAME stream, like the tiles
  // of a GEMM that share an operand panel: one of them misses to HBM, the others hit under that miss
  const long stream = (blockIdx.x & 7) + 8 * (long)((blockIdx.x >> 3) / share);
  const long base = (stream * bytes_per_wg) % wrap;
  const int nsteps = (int)(bytes_per_wg / SLOT);
  auto stage = [&](int slot, int step) {
    const char* g = src + (base + (long)step * SLOT) % wrap + wv * (NI * 1024) + lane * 16;
    char* l = smem + slot * SLOT + wv * (NI * 1024);
#pragma unroll
    for (int q = 0; q < NI; ++q) __builtin_amdgcn_global_load_lds((gbl_void*)(g + q * 1024), (lds_void*)(l + q * 1024), 16, 0, 0);
  };
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d)
    if (d < nsteps) stage(d, d);
  int slot = 0;
  for (int k = 0; k < nsteps; ++k) {
    // leave DEPTH - 2 younger steps in flight (fewer at the tail: wait for everything there)
    if (k + DEPTH - 1 <= nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 2) * NI) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (k + DEPTH - 1 < nsteps) stage(slot == 0 ? DEPTH - 1 : slot - 1, k + DEPTH - 1);
    slot = slot == DEPTH - 1 ? 0 : slot + 1;
  }
  if (sink && threadIdx.x == 0 && smem[lane] == 123) sink[0] = 1;     // (keeps the LDS image observable)
}

template <int DEPTH, int SLOT>
static double run(const char* src, long total, long wrap, int wg_per_cu, int ncu, int share = 1) {
  const size_t lds_ring = (size_t)DEPTH * SLOT;
  // pad the allocation so that exactly wg_per_cu workgroups fit in 160 KiB
  size_t lds = 160 * 1024 / wg_per_cu;
  lds = lds / 1024 * 1024;
  if (lds < lds_ring) return -1;
  if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)probe<DEPTH, SLOT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int grid = ncu * wg_per_cu;
  long per_wg = total / grid * share / SLOT * SLOT;            // every workgroup still moves total/grid*share... see below
  per_wg = per_wg / share;                                   // (kept equal to the unshared case: same bytes into LDS per launch)
  per_wg = per_wg / SLOT * SLOT;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((probe<DEPTH, SLOT>), dim3(grid), dim3(256), lds, 0, src, per_wg, wrap, share, (int*)nullptr);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe<DEPTH, SLOT>), dim3(grid), dim3(256), lds, 0, src, per_wg, wrap, share, (int*)nullptr);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  return (double)per_wg * grid / (best * 1e-3) / 1e9;   // GB/s, whole chip
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  const long total = 2L << 30;                           // 2 GiB moved per launch
  char* buf;
  CK(hipMalloc(&buf, total));
  CK(hipMemset(buf, 1, total));
  printf("device %s, %d CUs; %ld MiB per launch, best of 5\n", prop.name, ncu, total >> 20);
  printf("%-10s %-8s %-6s %-6s %12s %14s %16s\n", "source", "wg/CU", "depth", "slot", "in flight/CU", "chip GB/s", "per-CU GB/s");
  for (int srcmode = 0; srcmode < 2; ++srcmode) {
    const long wrap = srcmode == 0 ? total : (2L << 20);  // HBM stream, or a 2 MiB window that lives in L2
    for (int wg = 1; wg <= 3; ++wg) {
#define ROW(DEPTH, SLOT)                                                                                               \
  {                                                                                                                    \
    const double gbs = run<DEPTH, SLOT>(buf, total, wrap, wg, ncu);                                                    \
    if (gbs > 0)                                                                                                       \
      printf("%-10s %-8d %-6d %-6d %9d KiB %14.0f %16.1f\n", srcmode ? "L2" : "HBM", wg, DEPTH, SLOT >> 10,            \
             wg * (DEPTH - 1) * (SLOT >> 10), gbs, gbs / ncu);                                                         \
  }
      ROW(2, 16384) ROW(3, 16384) ROW(4, 16384) ROW(6, 16384) ROW(8, 16384)
      ROW(2, 32768) ROW(3, 32768) ROW(4, 32768)
#undef ROW
    }
  }
  printf("\nshared streams (HBM source; N workgroups of an XCD read the same bytes at the same time: LDS traffic as above, HBM traffic / N)\n");
  printf("%-10s %-8s %-6s %-6s %12s %14s %16s\n", "sharers", "wg/CU", "depth", "slot", "in flight/CU", "chip GB/s", "per-CU GB/s");
  for (int share : {3, 12}) {
    for (int wg = 1; wg <= 3; ++wg) {
#define ROW(DEPTH, SLOT)                                                                                               \
  {                                                                                                                    \
    const double gbs = run<DEPTH, SLOT>(buf, total, total, wg, ncu, share);                                            \
    if (gbs > 0)                                                                                                       \
      printf("%-10d %-8d %-6d %-6d %9d KiB %14.0f %16.1f\n", share, wg, DEPTH, SLOT >> 10, wg * (DEPTH - 1) * (SLOT >> 10), gbs,  \
             gbs / ncu);                                                                                               \
  }
      ROW(2, 16384) ROW(3, 16384) ROW(4, 16384) ROW(6, 16384) ROW(8, 16384)
      ROW(2, 32768) ROW(3, 32768) ROW(4, 32768)
#undef ROW
    }
  }
  CK(hipFree(buf));
  return 0;
}
