// Micro-benchmark: sustained rate of v_mfma_f32_32x32x2_f32 / 16x16x4 with and without the LDS reads
// of the tap-conv k-step beside them.  hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k32(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 8192; i += 256) {
    if (MODE >= 3) {   // pseudo-random operands in [-1, 1): realistic bit toggling
      unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
      smem[i] = (float)(int)(h & 0xffffff) / 8388608.f - 1.f;
    } else smem[i] = (float)(i & 7) * 0.125f;
  }
  __syncthreads();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f32x4 a = {1.f, 2.f, 3.f, 4.f};
  float b = 0.5f;
  const unsigned wa = lane * 16, xa = 16384 + lane * 4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      if (MODE == 1 || MODE == 3) {   // compiler-visible LDS reads
        a = *reinterpret_cast<const f32x4*>(smem + lane * 4 + ks * 256);
        b = smem[4096 + lane + ks * 64];
      }
      if (MODE == 2) {   // asm reads two steps ahead would need rotation; here: read + counted wait
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a) : "v"(wa), "i"(ks * 1024));
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(b) : "v"(xa), "i"(ks * 256));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b, acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + tid] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = 1.f + threadIdx.x, b = 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename K>
static void run(const char* name, K kern, int blocks, size_t lds, double flop_per_block_iter, int iters) {
  float* out;
  hipMalloc(&out, blocks * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s blocks %5d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks, ms, flop_per_block_iter * blocks * iters / ms / 1e9);
  hipFree(out);
}

int main() {
  const int iters = 400;
  const double f32 = 4.0 * 16 * 4 * 2.0 * 32 * 32 * 2;   // waves * ksteps * mfma * flop
  for (int blocks : {256, 512, 1024}) {
    run("32x32x2 regs only", k32<0>, blocks, 65536, f32, iters);
    run("32x32x2 + compiler LDS reads", k32<1>, blocks, 65536, f32, iters);
    run("32x32x2 + asm LDS reads, wait(0)", k32<2>, blocks, 65536, f32, iters);
    run("32x32x2 + LDS reads, RANDOM operands", k32<3>, blocks, 65536, f32, iters);
    run("16x16x4 x16 acc regs only", k16<16>, blocks, 0, 4.0 * 16 * 16 * 2.0 * 16 * 16 * 4, iters);
    run("16x16x4 x4 acc regs only", k16<4>, blocks, 0, 4.0 * 16 * 4 * 2.0 * 16 * 16 * 4, iters);
  }
  return 0;
}
