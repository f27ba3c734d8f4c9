// clock_probe.hip -- what a "cycle" of s_memtime is on this part and what the shader clock does under MFMA load: every wave runs a
// dependency-free stream of v_mfma_f32_32x32x16_bf16 (four accumulators) and stamps s_memtime (shader-clock ticks) and s_memrealtime
// (100 MHz) around it.  Prints ticks per MFMA (the instruction's issue interval in shader cycles), shader ticks per microsecond (= the clock
// in MHz) and the bf16 rate, for one block on an idle chip and for a grid that fills every CU at 1, 2 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O2 clock_probe.hip -o clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256, 2) void probe(unsigned long long* out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (threadIdx.x + j)); b[j] = (__bf16)(0.002f * (threadIdx.x ^ j)); }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0];
  asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) {
    unsigned long long* o = out + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    o[0] = t1 - t0; o[1] = r1 - r0; o[2] = (unsigned long long)(s == 12345.f);
  }
}

static void run(const char* name, int blocks, unsigned long long* dbuf) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, dbuf, 200);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, dbuf, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h((size_t)blocks * 16);
  (void)hipMemcpy(h.data(), dbuf, h.size() * 8, hipMemcpyDeviceToHost);
  double ticks = 0, real = 0;
  for (int i = 0; i < blocks * 4; ++i) { ticks += (double)h[(size_t)i * 4]; real += (double)h[(size_t)i * 4 + 1]; }
  ticks /= blocks * 4; real /= blocks * 4;
  const double mfmas = 4.0 * iters;
  printf("%-34s %6d blocks: %8.1f us, %6.2f s_memtime ticks per MFMA of a wave, %7.1f ticks per us of s_memrealtime, %7.1f TFLOP/s\n", name, blocks, ms * 1e3,
         ticks / mfmas, ticks / (real / 100.0), mfmas * 4 * blocks * 32768.0 / (ms * 1e-3) / 1e12);
}

int main() {
  unsigned long long* dbuf;
  (void)hipMalloc(&dbuf, (size_t)1024 * 16 * 8);
  run("one block, idle chip", 1, dbuf);
  run("one wave per SIMD on every CU", 256, dbuf);
  run("two waves per SIMD on every CU", 512, dbuf);
  run("one block again", 1, dbuf);
  return 0;
}
