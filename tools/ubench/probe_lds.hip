// probe_lds.hip -- two hardware facts the bundle-layout kernels (bl_*.hip) rely on, checked on the device:
//   1. ds_read_b64_tr_b16: destination lane l, element j <- the bf16 at (address supplied by lane 16*(l>>4) + 4*j + ((l&15)>>2)) + 2*((l&15)&3)
//   2. buffer_load_dwordx4 ... lds with an out-of-range offset writes ZEROS to the lane's LDS slot (zero padding by descriptor bounds)
// build: hipcc --offload-arch=gfx950 -O2 probe_lds.hip -o probe_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void tr_kernel(const unsigned short* in, const int* addr, unsigned short* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = in[i];
  __syncthreads();
  auto p = (__attribute__((address_space(3))) s16x4*)(lds + addr[threadIdx.x]);
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}

__global__ void oob_kernel(const u32x4* in, u32x4* out, int n, int off) {
  extern __shared__ __attribute__((aligned(16))) u32x4 l4[];
  const int lane = threadIdx.x;
  l4[lane] = u32x4{0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu};
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, n * 16, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)l4, 16, (lane + off) * 16, 0, 0, 0);
  __syncthreads();
  out[lane] = l4[lane];
}

int main() {
  std::vector<unsigned short> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = (unsigned short)i;
  unsigned short *din, *dout;
  int* daddr;
  hipMalloc(&din, 8192); hipMalloc(&dout, 512); hipMalloc(&daddr, 256);
  hipMemcpy(din, h.data(), 8192, hipMemcpyHostToDevice);
  int bad = 0;
  for (int trial = 0; trial < 2; ++trial) {
    // trial 0: lane i -> elements 4i..4i+3 (the natural image); trial 1: scattered 8-byte-aligned addresses
    std::vector<int> a(64);
    for (int i = 0; i < 64; ++i) a[i] = trial == 0 ? 4 * i : 4 * ((i * 37 + 11) % 1000);
    hipMemcpy(daddr, a.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(tr_kernel, dim3(1), dim3(64), 8192, 0, din, daddr, dout);
    std::vector<unsigned short> o(256);
    hipMemcpy(o.data(), dout, 512, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 4; ++j) {
        const int src = 16 * (l >> 4) + 4 * j + ((l & 15) >> 2);
        const int want = a[src] + ((l & 15) & 3);
        if (o[l * 4 + j] != want) {
          if (bad < 8) printf("tr trial %d lane %d elem %d: got %d want %d\n", trial, l, j, o[l * 4 + j], want);
          ++bad;
        }
      }
    if (trial == 0) { printf("tr natural image, lane 0..3 elems:"); for (int i = 0; i < 16; ++i) printf(" %d", o[i]); printf("\n"); }
  }
  printf("ds_read_b64_tr_b16 mapping %s (%d mismatches)\n", bad ? "DIFFERS" : "as assumed", bad);
  u32x4 *bin, *bout;
  hipMalloc(&bin, 64 * 16); hipMalloc(&bout, 64 * 16);
  std::vector<unsigned> hb(256);
  for (int i = 0; i < 256; ++i) hb[i] = 1000 + i;
  hipMemcpy(bin, hb.data(), 1024, hipMemcpyHostToDevice);
  for (int off : {0, 40, -8}) {
    hipLaunchKernelGGL(oob_kernel, dim3(1), dim3(64), 1024, 0, bin, bout, 64, off);
    std::vector<unsigned> ob(256);
    hipMemcpy(ob.data(), bout, 1024, hipMemcpyDeviceToHost);
    int zeros = 0, kept = 0, data = 0, other = 0;
    for (int l = 0; l < 64; ++l) {
      const int s = l + off;
      const unsigned v = ob[4 * l];
      if (s >= 0 && s < 64) { if (v == 1000u + 4 * s) ++data; else ++other; }
      else if (v == 0) ++zeros; else if (v == 0xdeadbeefu) ++kept; else ++other;
    }
    printf("buffer_load lds off %d: in-range correct %d, out-of-range zero %d, out-of-range untouched %d, other %d\n", off, data, zeros, kept, other);
  }
  return 0;
}
