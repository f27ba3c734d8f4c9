// Micro-benchmark: sustained rate of v_mfma_f32_32x32x16_bf16 in the shape of the tap-conv k-step (tapconv3.hip): NA A-fragment +
// NB B-fragment ds_read_b128 per k-step feeding NM MFMAs on FM accumulators, optionally a barrier every KSC k-steps and an LDS-DMA
// weight stream of the matching size.  hipcc --offload-arch=gfx950 -O3 mfma_bf16.hip -o mfma_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// MODE bit 0: LDS reads, bit 1: barrier per KSC k-steps, bit 2: LDS-DMA stream (KSC * NPW * FM KB per chunk), bit 3: random data
template <int FM, int NPW, int NPX, int KSC, int MODE>
__global__ __launch_bounds__(256, 1) void kb(float* out, const u32x4* wsrc, int iters) {
  extern __shared__ __attribute__((aligned(16))) u32x4 smem[];
  constexpr int WCHU = KSC * NPW * FM * 64;
  u32x4* Ws = smem;               // 2 x WCHU
  u32x4* Xs = smem + 2 * WCHU;    // 1024 units
  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
  for (int i = tid; i < 2 * WCHU + 1024; i += 256) {
    unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    u32x4 v;
    for (int e = 0; e < 4; ++e) { h = h * 1664525u + 1013904223u; v[e] = (MODE & 8) ? ((h & 0x7fff7fffu) | 0x30003000u) & 0xbfffbfffu : 0x3f803f80u; }
    smem[i] = v;
  }
  __syncthreads();
  f32x16 acc[FM];
  for (int i = 0; i < FM; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  u32x4 a[KSC][NPW][FM], b[KSC][NPX];
  for (int ks = 0; ks < KSC; ++ks) {
    for (int q = 0; q < NPW; ++q) for (int i = 0; i < FM; ++i) a[ks][q][i] = smem[lane + 64 * i];
    for (int q = 0; q < NPX; ++q) b[ks][q] = Xs[lane + 64 * q];
  }
  constexpr int NPM = NPW > NPX ? NPW : NPX;
  for (int it = 0; it < iters; ++it) {
    if (MODE & 4) {
      const u32x4* src = wsrc + (size_t)(it & 63) * WCHU;
      u32x4* dst = Ws + ((it + 1) & 1) * WCHU;
#pragma unroll
      for (int u = 0; u * 256 < WCHU; ++u) {
        const int idx = u * 256 + tid;
        if (WCHU % 256 == 0 || idx < WCHU)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + idx),
                                           (__attribute__((address_space(3))) void*)(dst + (idx & ~63)), 16, 0, 0);
      }
    }
    const u32x4* wb = Ws + (it & 1) * WCHU + lane;
    const u32x4* xb = Xs + wn * 32 + (lane & 31) + (lane >> 5) * 256;
    if (MODE & 1) {
#pragma unroll
      for (int ks = 0; ks < KSC; ++ks) {
#pragma unroll
        for (int q = 0; q < NPX; ++q) b[ks][q] = xb[ks * 8 + q * 64];
#pragma unroll
        for (int q = 0; q < NPW; ++q)
#pragma unroll
          for (int i = 0; i < FM; ++i) a[ks][q][i] = wb[((ks * NPW + q) * FM + i) * 64];
      }
    }
#pragma unroll
    for (int ks = 0; ks < KSC; ++ks) {
#pragma unroll
      for (int lvl = NPM - 1; lvl >= 0; --lvl)
#pragma unroll
        for (int qw = 0; qw < NPW; ++qw) {
          const int qx = lvl - qw;
          if (qx < 0 || qx >= NPX) continue;
#pragma unroll
          for (int i = 0; i < FM; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[ks][qw][i]), __builtin_bit_cast(bf16x8, b[ks][qx]), acc[i], 0, 0, 0);
        }
    }
    if (MODE & 2) __syncthreads();
  }
  float s = 0.f;
  for (int i = 0; i < FM; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + tid] = s;
}

template <typename K>
static void run(const char* name, K kern, int blocks, size_t lds, double mfma_per_wave_iter, int iters, const u32x4* w) {
  float* out;
  hipMalloc(&out, blocks * 256 * sizeof(float));
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, out, w, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, out, w, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = mfma_per_wave_iter * 4.0 * 32768.0 * blocks * iters;
  printf("%-58s blocks %4d lds %6zu  %8.3f ms  %7.1f TFLOP/s  (%5.1f %% of 2500)\n", name, blocks, lds, ms, flop / ms / 1e9, flop / ms / 1e9 / 25.0);
  hipFree(out);
}

#define RUN(FM, NPW, NPX, KSC, MODE, BLK, LDSX, label)                                                                   \
  run(label, kb<FM, NPW, NPX, KSC, MODE>, BLK, (size_t)(2 * KSC * NPW * FM * 64 + 1024) * 16 + LDSX,                       \
      (double)KSC * FM * (NPW == 1 ? NPX : (NPW == 2 ? 3 : 6)), iters, w)

int main() {
  const int iters = 2000;
  u32x4* w;
  hipMalloc(&w, 64 * 4 * 3 * 4 * 64 * 16 * 2);
  hipMemset(w, 0x3f, 64 * 4 * 3 * 4 * 64 * 16 * 2);
  for (int blocks : {256, 512}) {
    const size_t pad = blocks == 256 ? 90 * 1024 : 0;   // 256 blocks: one per CU (LDS padding keeps a second one out)
    printf("---- %d blocks (%s)\n", blocks, blocks == 256 ? "one per CU" : "two per CU");
    RUN(4, 1, 1, 4, 8, blocks, pad, "bf16 FM4 KSC4: regs only");
    RUN(4, 1, 1, 4, 9, blocks, pad, "bf16 FM4 KSC4: + LDS reads (5 per 4 MFMA)");
    RUN(4, 1, 1, 4, 11, blocks, pad, "bf16 FM4 KSC4: + LDS reads + barrier");
    RUN(4, 1, 1, 4, 15, blocks, pad, "bf16 FM4 KSC4: + LDS reads + barrier + LDS-DMA 16 KB");
    RUN(4, 3, 3, 2, 8, blocks, pad, "x6   FM4 KSC2: regs only");
    RUN(4, 3, 3, 2, 9, blocks, pad, "x6   FM4 KSC2: + LDS reads (15 per 24 MFMA)");
    RUN(4, 3, 3, 2, 11, blocks, pad, "x6   FM4 KSC2: + LDS reads + barrier");
    RUN(4, 3, 3, 2, 15, blocks, pad, "x6   FM4 KSC2: + LDS reads + barrier + LDS-DMA 24 KB");
    RUN(2, 3, 3, 2, 15, blocks, pad, "x6   FM2 KSC2: + LDS reads + barrier + LDS-DMA 12 KB");
    RUN(4, 2, 2, 2, 15, blocks, pad, "x3   FM4 KSC2: + LDS reads + barrier + LDS-DMA 16 KB");
  }
  return 0;
}
