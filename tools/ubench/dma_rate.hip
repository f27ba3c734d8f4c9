// dma_rate.hip -- bytes per clock and CU that reach LDS from L2, for the two ways a tile can be staged: LDS-DMA
// (global_load_lds_dwordx4: 1 KB per wave instruction, no registers) and global_load_dwordx4 into registers + ds_write_b128.
// Two blocks of four waves per CU; every block streams one of eight 256 KB windows (2 MB: L2 hits, beyond the 32 KB L1) (wave w reads pieces
// w, w + 4, ...), BATCH pieces in flight per wave before it waits.
// build: hipcc --offload-arch=gfx950 -O2 dma_rate.hip -o dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int BATCH>
__global__ __launch_bounds__(256, 2) void dma_kernel(const u32x4* __restrict__ src, unsigned* out, int iters, int window_units) {
  extern __shared__ __attribute__((aligned(16))) u32x4 lds[];
  typedef __attribute__((address_space(3))) void* lds_t;
  const unsigned lds0 = (unsigned)(unsigned long long)(lds_t)lds;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const u32x4* base = src + (size_t)(blockIdx.x % 8) * window_units;   // 2 MB in all: every piece is an L2 hit, none an L1 hit
  unsigned acc = 0;
  int piece = wave;
  for (int it = 0; it < iters; ++it) {
    u32x4 v[BATCH];
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const u32x4* p = base + ((piece * 64) % window_units) + lane;
      const unsigned dst = lds0 + (unsigned)(((wave * BATCH + u) & 31) * 1024);
      if (MODE == 0) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(p), "s"(dst) : "memory");
      } else {
        v[u] = __builtin_nontemporal_load(p);
      }
      piece += 4;
    }
    if (MODE == 1) {
#pragma unroll
      for (int u = 0; u < BATCH; ++u) lds[((wave * BATCH + u) & 31) * 64 + lane] = v[u];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (out && lds[threadIdx.x][0] == 0x12345u) out[threadIdx.x] = acc;
}

template <int MODE, int BATCH>
static void run(const char* name, const u32x4* src, int blocks_per_cu) {
  const int iters = 2000, cus = 256, window_units = 16384;   // 256 KB per block
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto k = dma_kernel<MODE, BATCH>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL(k, dim3(cus * blocks_per_cu), dim3(256), 32768 + 1024, 0, src, nullptr, 20, window_units);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(cus * blocks_per_cu), dim3(256), 32768 + 1024, 0, src, nullptr, iters, window_units);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double bytes_per_cu = (double)iters * BATCH * 1024 * 4 * blocks_per_cu;
  printf("%-40s batch %d, %d block(s)/CU: %8.1f us, %6.1f bytes per clock per CU (2.4 GHz), %5.2f TB/s chip\n", name, BATCH, blocks_per_cu, ms * 1e3,
         bytes_per_cu / (ms * 1e-3 * 2.4e9), bytes_per_cu * cus / (ms * 1e-3) / 1e12);
}

int main() {
  u32x4* src;
  const size_t units = (size_t)512 * 16384;   // 128 MB: inside the 256 MB MALL, beyond the L2s
  (void)hipMalloc(&src, units * 16);
  (void)hipMemset(src, 1, units * 16);
  for (int b = 1; b <= 2; ++b) {
    run<0, 2>("global_load_lds_dwordx4", src, b);
    run<0, 8>("global_load_lds_dwordx4", src, b);
    run<1, 2>("global_load_dwordx4 + ds_write_b128", src, b);
    run<1, 8>("global_load_dwordx4 + ds_write_b128", src, b);
  }
  return 0;
}
