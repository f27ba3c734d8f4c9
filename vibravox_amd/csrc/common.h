// Shared host/device helpers for libeben_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "eben_hip.h"

namespace eben {

// ---- error plumbing ------------------------------------------------------------------
char* tls_error_buffer();
inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(tls_error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}
inline int hip_fail(hipError_t e, const char* what) {
  snprintf(tls_error_buffer(), 512, "%s: %s", what, hipGetErrorString(e));
  return (int)e;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is PER DEVICE: a process that touches a second GPU needs it there too.  One bit per
// device ordinal (mod 64) and kernel; setting it twice (two host threads racing) is harmless.
struct LdsAttrOnce { std::atomic<unsigned long long> devices{0ull}; };
inline int current_device_ordinal() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
  return dev;
}
inline hipError_t lds_attr_once(LdsAttrOnce& once, const void* kern) {
  const unsigned long long bit = 1ull << (current_device_ordinal() & 63);
  if (once.devices.load(std::memory_order_acquire) & bit) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) once.devices.fetch_or(bit, std::memory_order_release);
  return e;
}
// compute units of the CURRENT device (cached per ordinal)
inline int device_cus() {
  static std::atomic<int> cache[64];
  const int dev = current_device_ordinal() & 63;
  int n = cache[dev].load(std::memory_order_relaxed);
  if (n <= 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cache[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}
#define EBEN_CHECK_LAUNCH(what)                         \
  do {                                                  \
    hipError_t e__ = hipGetLastError();                 \
    if (e__ != hipSuccess) return hip_fail(e__, what);  \
  } while (0)
#define EBEN_REQUIRE(cond, ...) \
  do {                          \
    if (!(cond)) return fail(EBEN_EINVAL, __VA_ARGS__); \
  } while (0)

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int round_up(int a, int b) { return ceil_div(a, b) * b; }
inline size_t ceil_div_z(size_t a, size_t b) { return (a + b - 1) / b; }

// ---- device helpers ------------------------------------------------------------------
__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }
__device__ __forceinline__ float dlrelu(float ref, float slope) { return ref > 0.f ? 1.f : slope; }

// MI355X: 8 XCDs, block b is dispatched to XCD b%8 (speed-only assumption).  Remap the linear
// block id so that each XCD works on a contiguous range of tiles (they share weight panels in
// that XCD's private L2).  Bijective for any grid size.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblocks) {
  const unsigned nx = 8;
  const unsigned q = nblocks / nx, r = nblocks % nx;
  const unsigned xcd = bid % nx, idx = bid / nx;
  const unsigned start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + idx;
}

// Sum over the wave, every lane receives it: the butterfly v += v[lane ^ 32], ^ 16, ^ 8, ^ 4, ^ 2, ^ 1 -- on the cross-lane VALU paths
// instead of six ds_bpermute_b32 (__shfl_xor): v_permlane32_swap / v_permlane16_swap pair the halves / the odd and even rows, a DPP
// row rotation by 8 / 4 / 2 / 1 pairs lane ^ 8 ... lane ^ 1 once the partial sums repeat with that period.  Same pairs, same order:
// bit-identical to the shuffle form ([MI355X] the 128 reductions of a MelGAN-head weight-gradient block: 24 of the kernel's 86 us).
__device__ __forceinline__ float wave_sum(float v) {
  // (inline asm: the swaps exchange lanes BETWEEN their two operands, and handed the same value twice through the builtin hipcc
  // (ROCm 7.2) takes both results from one register; "+v" on two variables keeps two registers.  s_nop 1 = the two wait states
  // between a VALU write of an operand and the swap reading it)
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  v = a + b;
  a = v; b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  v = a + b;
#define EBEN_ROW_ROR_ADD(N)                                                                                                   \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + (N), 0xf, 0xf, false));
  EBEN_ROW_ROR_ADD(8)
  EBEN_ROW_ROR_ADD(4)
  EBEN_ROW_ROR_ADD(2)
  EBEN_ROW_ROR_ADD(1)
#undef EBEN_ROW_ROR_ADD
  return v;
}
// the shuffle form (reference of tests / scratch comparisons)
__device__ __forceinline__ float wave_sum_shfl(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum for 256-thread blocks; result valid in every thread
__device__ __forceinline__ float block_sum_256(float v, float* red /* >= 4 floats of LDS */) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Canonical strided convolution behind a Conv1d (as is) or a ConvTranspose1d (its adjoint):
// big side X (B,Cin,Lin), small side Y (B,Cout,Lout), weight (Cout, Cin/g, k).
struct Canon {
  int B, Cin, Cout, Lin, Lout, k, s, d, g, pl, pr, reflect;
  int bf16;   // EbenConv1dDesc.math != EBEN_MATH_F32: bf16 MFMA operands where tapconv3.hip / conv_dw3.hip cover the layer
  // EBEN_MATH_BF16X2: the tap-conv direction (0 gather-strided, 1 phase-scatter) whose INPUT operand is the layer's activation
  // tensor (the layer's forward) -- staged as hi + lo bf16 tiles there and in the weight gradient; -1 otherwise
  int xsplit_dir;
  int np;     // pieces per MFMA operand in the bf16 tap-conv: 1 (EBEN_MATH_BF16 / BF16X2), 2 (BF16X3), 3 (BF16X6)
  int bl;     // EBEN_LAYOUT_BL: activations / gradients at rest as bf16 bundles (include/eben_hip.h), eben_bl_* entry points
};
int canon_from_desc(const EbenConv1dDesc* d, Canon* c);

// operands / fused element-wise stages of one tap-conv launch (either kernel generation)
struct TapIO {
  const float* x; const float* xmask; int in_mode; float in_slope;
  const float* wp; const float* bias; const float* res; float res_slope;
  const float* emask; float emask_slope; float out_slope; float* y; int accumulate;
  // batched right-hand sides (eben_conv1d_bwd_dx_ex): the residual is added only for batch rows
  // b < res_rows (0 = all rows); the epilogue mask is read from batch row
  // em_map[b / em_seg] * em_seg + b % em_seg (em_seg = 0: row b)
  int res_rows; int em_seg; int em_map[4];
  // feature-matching gradient formed in the epilogue (eben_conv1d_bwd_dx_fm): `res` then points at the REFERENCE rows of the
  // embedding the mask is read from, and rows b < res_rows receive c1 sgn(mask - res) - c2 sgn(mask) with c1 = fm_gs / s2,
  // c2 = fm_gs s1 / s2^2, (s1, s2) = fm_sums[0..1] on the device (feature_loss.py:40-47); nullptr = plain residual
  const float* fm_sums; float fm_gs;
  // bundle layout (Canon.bl): input planes (xl: the lo plane of a split input), output planes (yl nullable), the saved activation
  // planes the epilogue takes its mask and feature-matching operands from (reference rows bl_ref_off batch rows behind the enhanced)
  const void* xh; const void* xl; void* yh; void* yl; const void* eh; const void* el; int bl_ref_off;
  const void* ec;   // nullable: feature-matching code plane of the embedding eh / el (bl_edge.hip: bl_fm_code), 8 bytes per unit of its enhanced rows
  // phases as rows (eben_bl_conv1d_bwd_dx_pr): the launch is the stride-1 gather form of a strided conv's input gradient whose output rows
  // are (phase, channel); they are stored depth-to-space into planes of pr_CBy bundles x pr_Ly positions per batch row (pr_cbg bundles per group)
  int pr_S, pr_cbg, pr_Ly, pr_CBy, pr_order;   // pr_order 1: primed rows (channel bundle, phase, channel in bundle) -- tap4_kernel only
};

// Tap geometry of one output phase.  mode 0 (gather-strided): every block uses (J0, off0_gs, nt_gs).
// mode 1 (phase-scatter): phase ph = output position mod OS keeps the taps k = k0 + j*kstep with
// (ph + pad - k*dil) % OS == 0; J = 0 when the phase has no tap.
struct PhaseGeom { int J, off0, minoff, nt, oo, k0; };
__host__ __device__ inline PhaseGeom phase_geom(int mode, int ph, int J0, int off0_gs, int nt_gs, int dstep, int OS,
                                                int ps_pad, int ps_k, int ps_d, int ps_kstep, int Ly) {
  PhaseGeom q;
  q.J = J0; q.off0 = off0_gs; q.nt = nt_gs; q.oo = 0; q.k0 = 0;
  if (mode == 1) {
    int k0 = -1;
    for (int c = 0; c < ps_kstep; ++c) {
      int v = (ph + ps_pad - c * ps_d) % OS;
      if (v < 0) v += OS;
      if (v == 0) { k0 = c; break; }
    }
    if (k0 >= 0 && k0 < ps_k) {
      q.J = (ps_k - 1 - k0) / ps_kstep + 1;
      q.off0 = (ph + ps_pad - k0 * ps_d) / OS;
    } else {
      q.J = 0;
      q.off0 = 0;
    }
    q.k0 = k0;
    q.nt = ph < Ly ? (Ly - ph + OS - 1) / OS : 0;
    q.oo = ph;
  }
  q.minoff = (dstep >= 0 || q.J == 0) ? q.off0 : q.off0 + (q.J - 1) * dstep;
  return q;
}

// second-generation tap-conv (tapconv2.hip): 32x32x2 MFMA tiles, LDS-DMA weight stream
int tap2_applicable(const Canon& c, int dir);
size_t tap2_packed_floats(const Canon& c, int dir);
int tap2_pack(const Canon& c, int dir, const float* w, const float* scale, float* wp, hipStream_t st);
int tap2_launch(const Canon& c, int dir, const TapIO& io, int reflect, hipStream_t st);

// bf16-operand form of the second generation (tapconv3.hip): 32x32x16 bf16 MFMA, fp32 accumulate / storage
int tap3_applicable(const Canon& c, int dir);
int tap3_is_big(const Canon& c, int dir);
size_t tap3_packed_floats(const Canon& c, int dir);
int tap3_pack(const Canon& c, int dir, const float* w, const float* scale, float* wp, hipStream_t st);
int tap3_pack_multi(const Canon* cs, const int* dirs, const float* const* ws, const float* const* scales, float* const* wps, int n, hipStream_t st);
int tap3_launch(const Canon& c, int dir, const TapIO& io, int reflect, hipStream_t st);

// the generator's strided / transposed / latent convs with the MFMA B operand loaded straight from the fp32 rows (gen_conv.hip): forward
// launches only (kernel generation 5 of eben_conv1d_kernel_generation)
int gc_applicable(const Canon& c, int dir);
size_t gc_packed_floats(const Canon& c, int dir);
int gc_pack(const Canon& c, int dir, const float* w, const float* scale, float* wp, hipStream_t st);
int gc_launch(const Canon& c, int dir, const TapIO& io, hipStream_t st);

// direct (VALU) tap-conv for layers with a handful of output channels per group (thinconv.hip)
int thin_applicable(const Canon& c, int dir);
size_t thin_packed_floats(const Canon& c, int dir);
int thin_pack(const Canon& c, int dir, const float* w, const float* scale, float* wp, hipStream_t st);
int thin_launch(const Canon& c, int dir, const TapIO& io, int reflect, hipStream_t st);
// which kernel serves (layer, direction): 1 tapconv.hip, 2 tapconv2.hip, 3 thinconv.hip, 4 tapconv3.hip (bf16)
int tap_generation(const Canon& c, int dir);

// second-generation weight-gradient kernel (conv_dw2.hip): pre-transposed A image + LDS-DMA
struct Dw2Args {
  const float* a; const float* amask; int a_mode; float a_slope;     // pack pre-pass input
  const float* x; const float* xmask; int x_mode; float x_slope;
  float* ap;        // packed A image
  float* slabs;
  int B, G, Cg, Mg, Ca, Cx, La, Lx;
  int S, d, off0, J, Ng, has_bias, row_stride, reflect;
  int nsplit, nct, nchunks, nnt, nmt, XSTR, nch_max;
  long long slab_stride;
};

int dw2_applicable(const Canon& c);
size_t dw2_workspace(const Canon& c, int* nslab, int* row_stride);
int dw2_launch(const Canon& c, Dw2Args a, float* workspace, size_t ws_bytes, hipStream_t st);


// bf16-operand weight-gradient kernel (conv_dw3.hip): k-step = one time step x 16 batch items
int dw3_applicable(const Canon& c);
size_t dw3_workspace(const Canon& c, int* nslab, int* row_stride);
int dw3_launch(const Canon& c, const float* a, const float* amask, float a_slope, const float* x, float x_slope, int has_bias, float* workspace,
               size_t ws_bytes, hipStream_t st);

}  // namespace eben
