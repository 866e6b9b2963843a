// cu_hog.hip -- a co-tenant for scheduling experiments: `n` workgroups that each take a whole CU (all of its LDS) and spin for `cycles`
// shader clocks, on the stream given.  Stands in for RCCL's channel workgroups, which hold CUs for the length of a collective while the
// step's GEMMs are launched beside them (tools/cotenant_probe.py).   hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/cu_hog.so tools/cu_hog.hip
#include <hip/hip_runtime.h>
#include <cstdint>

__global__ __launch_bounds__(256) void hog_kernel(long long cycles, int* sink) {
  extern __shared__ char smem[];
  const long long t0 = (long long)__builtin_amdgcn_s_memtime();
  int acc = 0;
  while ((long long)__builtin_amdgcn_s_memtime() - t0 < cycles) {
    smem[threadIdx.x] = (char)acc;
    acc += smem[(threadIdx.x + 1) & 255];
    __builtin_amdgcn_s_sleep(8);
  }
  if (acc == 0x7fffffff) sink[0] = acc;
}

extern "C" int cu_hog(int n, long long cycles, void* stream) {
  static bool once = false;
  if (!once) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(hog_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1;
    once = true;
  }
  hipLaunchKernelGGL(hog_kernel, dim3((unsigned)n), dim3(256), 160 * 1024, (hipStream_t)stream, cycles, (int*)nullptr);
  return (int)hipGetLastError();
}
