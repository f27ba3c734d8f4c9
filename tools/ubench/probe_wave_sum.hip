// wave_sum (common.h: v_permlane32_swap / v_permlane16_swap / DPP row rotations) against the __shfl_xor butterfly it replaces: bit for bit,
// every lane, on random data.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -I include -I vibravox_amd/csrc tools/ubench/probe_wave_sum.hip -o /tmp/probe_wave_sum && /tmp/probe_wave_sum
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"

__global__ void probe(const float* in, float* a, float* b) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  a[i] = eben::wave_sum(in[i]);
  b[i] = eben::wave_sum_shfl(in[i]);
}

int main() {
  const int n = 64 * 4096;
  std::vector<float> h(n);
  srand(1234);
  for (int i = 0; i < n; ++i) h[i] = ((rand() % 20001) - 10000) * 1e-3f * (1.f + (rand() % 1000) * 1e-3f) * ((i & 7) == 0 ? 1e-6f : 1.f);
  float *in, *a, *b;
  hipMalloc(&in, n * 4); hipMalloc(&a, n * 4); hipMalloc(&b, n * 4);
  hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(n / 256), dim3(256), 0, 0, in, a, b);
  std::vector<float> ha(n), hb(n);
  hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; ++i) bad += std::memcmp(&ha[i], &hb[i], 4) != 0;
  printf("wave_sum vs shuffle butterfly: %d of %d lanes differ\n", bad, n);
  return bad != 0;
}
