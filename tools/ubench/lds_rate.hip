// lds_rate.hip -- issue rate of the LDS reads the bundle-layout kernels are made of, per CU: two blocks of four waves per CU, every wave
// a dependent-free stream of reads at a fixed per-lane address pattern.  Prints LDS-pipe cycles per wave instruction (at the measured
// launch time and 2.4 GHz) for: ds_read_b64_tr_b16 at bl_dw's A-tile addresses (row stride 68 units), at its contiguous-X addresses
// (stride 4), at lane * 8 (one contiguous 512-byte run); ds_read_b64 at the A addresses; ds_read_b128 at lane * 16.
// build: hipcc --offload-arch=gfx950 -O2 lds_rate.hip -o lds_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void rate_kernel(unsigned* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned lds[];
  for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int G4 = lane >> 4, js = (lane & 15) >> 2, qs = lane & 3;
  const int koff = 8 * (G4 >> 1) + js;
  unsigned addr;
  if (MODE == 0 || MODE == 3) addr = (unsigned)(((wave * 4 + 2 * (G4 & 1) + (qs >> 1)) * 68 + koff) * 16 + 8 * (qs & 1));           // A tile
  else if (MODE == 1) addr = (unsigned)(((2 * (G4 & 1) + (qs >> 1)) + koff * 4 + wave * 301) * 16 + 8 * (qs & 1));                   // contiguous X
  else if (MODE == 2) addr = (unsigned)(lane * 8 + wave * 1024);
  else addr = (unsigned)(lane * 16 + wave * 2048);
  unsigned acc = 0;
  typedef __attribute__((address_space(3))) s16x4* lds4_t;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const unsigned a = addr + (unsigned)((u & 3) * 256);
      if (MODE <= 2) {
        s16x4 v;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(a));
        asm volatile("" :: "v"(v));
      } else if (MODE == 3) {
        u32x2 v;
        asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(a));
        asm volatile("" :: "v"(v));
      } else {
        u32x4 v;
        asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a));
        asm volatile("" :: "v"(v));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (out && acc == 12345u) out[threadIdx.x] = acc;
}

template <int MODE>
static void run(const char* name, int blocks_per_cu) {
  const int iters = 2000, cus = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto k = rate_kernel<MODE>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL(k, dim3(cus * blocks_per_cu), dim3(256), 65536, 0, nullptr, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(cus * blocks_per_cu), dim3(256), 65536, 0, nullptr, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_cu = (double)iters * 16 * 4 * blocks_per_cu;
  printf("%-44s %d block(s)/CU: %7.1f us, %5.2f cycles per wave instruction per CU (2.4 GHz)\n", name, blocks_per_cu, ms * 1e3, ms * 1e-3 * 2.4e9 / instr_per_cu);
}

int main() {
  for (int b = 1; b <= 2; ++b) {
    run<0>("ds_read_b64_tr_b16, A-tile addresses", b);
    run<1>("ds_read_b64_tr_b16, contiguous-X addresses", b);
    run<2>("ds_read_b64_tr_b16, lane * 8", b);
    run<3>("ds_read_b64, A-tile addresses", b);
    run<4>("ds_read_b128, lane * 16", b);
  }
  return 0;
}
