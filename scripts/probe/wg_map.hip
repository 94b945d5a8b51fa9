// Probe: which CU / XCD does workgroup i of a 2-WG-per-CU kernel land on, and when does it start?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256, 2) void probe(unsigned* out, int spin) {
  extern __shared__ float lds[];
  unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_REG_HW_ID
  unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float acc = 0.f;
  for (int i = 0; i < spin; ++i) { lds[threadIdx.x] = acc; __syncthreads(); acc += lds[(threadIdx.x + 1) & 255]; }
  if (threadIdx.x == 0) {
    out[blockIdx.x * 4 + 0] = hw;
    out[blockIdx.x * 4 + 1] = xcc;
    out[blockIdx.x * 4 + 2] = (unsigned)(t0 >> 6);
    out[blockIdx.x * 4 + 3] = (unsigned)acc;
  }
}
int main() {
  const int n = 1536;
  unsigned* d;
  hipMalloc(&d, n * 16);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(probe, dim3(n), dim3(256), 74 * 1024, 0, d, 20000);
    hipDeviceSynchronize();
  }
  std::vector<unsigned> h(n * 4);
  hipMemcpy(h.data(), d, n * 16, hipMemcpyDeviceToHost);
  unsigned tmin = ~0u;
  for (int i = 0; i < n; ++i) tmin = h[i * 4 + 2] < tmin ? h[i * 4 + 2] : tmin;
  for (int i = 0; i < n; ++i) {
    unsigned hw = h[i * 4];
    if (i < 80 || (i % 64) < 2 || i >= 500 && i < 540)
      printf("wg %4d xcc %u se %u sh %u cu %2u simd %u wave %u  t0 %u\n", i, h[i * 4 + 1] & 15, (hw >> 13) & 7,
             (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3, hw & 15, h[i * 4 + 2] - tmin);
  }
  return 0;
}
