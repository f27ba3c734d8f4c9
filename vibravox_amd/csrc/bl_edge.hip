// bl_edge.hip -- the ends of the discriminator chains in the bf16 BUNDLE LAYOUT (include/eben_hip.h, "bundle layout"), gfx950.
//
// Between the chain heads and the logits every activation / gradient of the batched discriminator engine is at rest as bf16
// [batch][channels / 8][length][8] (planes hi = bf16(v), lo = bf16(v - hi)): one 16-byte unit = 8 channels at one position = the unit
// the bf16 tap-conv stages into LDS (tapconv3.hip, template flag BL) and the weight-gradient kernel transposes on read (bl_dw.hip).
// The layers at the two ends have one side in plain fp32 (batch, channel, time) -- the band / waveform inputs, the logits and their
// seeds -- and a handful of channels there: they are streaming kernels (fp32 FMA, one thread = one position x one bundle), bound by
// the bundle side's HBM traffic:
//   head:  ReflectionPad1d(P) + Conv1d(c_in -> c_out, k, dilation, groups = c_in, zero padding) + bias + LeakyReLU
//          (eben_discriminator.py:66-76 layer 0: 4 -> 24, k 3, dilation 1 / 2 / 3;  melgan_discriminator.py:89-98 layer 0: 1 -> 16, k 15)
//          forward, input gradient (with the reflect fold, summed over the chains that share the input) and weight gradient;
//   tail:  Conv1d(C -> 1, k 3, padding 1) (eben_discriminator.py:150-157, melgan_discriminator.py:147-156: the logits)
//          forward, input gradient (+ the feature-matching term and the LeakyReLU mask of the embedding below) and weight gradient;
//   feature-matching sums over bundle planes (feature_loss.py:40-47), fp32 <-> bundle conversions (tests, tools).
#include "common.h"

#include <cstdlib>

namespace eben {

#ifdef EBEN_EDGE_NO_WAVESUM   // scratch ablation (wrong results): what the 8 (K + 1) wave reductions of a block cost
#define EDGE_WAVE_SUM(x) (x)
#else
#define EDGE_WAVE_SUM(x) wave_sum(x)
#endif


typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned bl_pk(float a, float b) {
  const f32x2v v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v));   // v_cvt_pk_bf16_f32 (RNE)
}
__device__ __forceinline__ void bl_unpack8(const u32x4 w, float (&f)[8]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    f[2 * e] = __builtin_bit_cast(float, w[e] << 16);
    f[2 * e + 1] = __builtin_bit_cast(float, w[e] & 0xffff0000u);
  }
}
__device__ __forceinline__ u32x4 bl_pack8(const float (&f)[8]) {
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = bl_pk(f[2 * e], f[2 * e + 1]);
  return o;
}
// lo plane of values whose hi plane is `hi`: bf16(v - hi), the residual exact in fp32
__device__ __forceinline__ u32x4 bl_pack8_lo(const float (&f)[8], const u32x4 hi) {
  float h[8];
  bl_unpack8(hi, h);
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = bl_pk(f[2 * e] - h[2 * e], f[2 * e + 1] - h[2 * e + 1]);
  return o;
}
// hi (+ lo) planes -> fp32 values
__device__ __forceinline__ void bl_load8(const u32x4* __restrict__ hi, const u32x4* __restrict__ lo, long long idx, float (&f)[8]) {
  bl_unpack8(hi[idx], f);
  if (lo) {
    float l[8];
    bl_unpack8(lo[idx], l);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] += l[e];
  }
}

// ---- conversions ----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bl_from_f32_kernel(const float* __restrict__ x, long long units, int CB, int L, u32x4* __restrict__ hi,
                                                          u32x4* __restrict__ lo) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < units; i += (long long)gridDim.x * 256) {
    const long long row = i / L;              // (batch, bundle)
    const int t = (int)(i - row * L);
    const float* p = x + row * 8 * L + t;     // channel 8 * bundle of that batch item
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = p[(long long)e * L];
    const u32x4 h = bl_pack8(f);
    hi[i] = h;
    if (lo) lo[i] = bl_pack8_lo(f, h);
  }
}
__global__ __launch_bounds__(256) void bl_to_f32_kernel(const u32x4* __restrict__ hi, const u32x4* __restrict__ lo, long long units, int L,
                                                        float* __restrict__ x) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < units; i += (long long)gridDim.x * 256) {
    const long long row = i / L;
    const int t = (int)(i - row * L);
    float f[8];
    bl_load8(hi, lo, i, f);
    float* p = x + row * 8 * L + t;
#pragma unroll
    for (int e = 0; e < 8; ++e) p[(long long)e * L] = f[e];
  }
}

// ---- chain heads ----------------------------------------------------------------------------------------------------------------
constexpr int BL_HEAD_JOBS = 4;
struct BlHeadJob {
  const float* x; const float* v; const float* scale; const float* bias;
  u32x4* yh; u32x4* yl;               // forward outputs / input-gradient inputs (g planes)
  int c_in, c_out, l_in, l_out, k, dil, pad, rpad;
  float out_slope;
};
struct BlHeadTable { BlHeadJob job[BL_HEAD_JOBS]; int n, batch; };

// position u of the reflect-padded (rpad) and then zero-extended signal: index into the row, or -1
__device__ __forceinline__ int bl_head_src(int p, int rpad, int l_in) {
  if (p < 0 || p >= l_in + 2 * rpad) return -1;
  int u = p - rpad;
  u = u < 0 ? -u : u;
  return u >= l_in ? 2 * (l_in - 1) - u : u;
}

typedef const __attribute__((address_space(4))) float* cfloat_t;   // uniform reads through the scalar cache (s_load)

// One thread = one output position x ALL output channels of one (job, batch item): the CIN * K input samples are loaded once (coalesced
// along time), the weights are block-uniform and arrive in SGPRs (static offsets into the job's v: s_load_dwordx8/16), the weight-norm
// scale is applied once per channel to the finished sum (the layer is linear in v).  OG = output channels per input channel.
template <int CIN, int OG, int K>
__global__ __launch_bounds__(256) void bl_head_fwd_kernel(const BlHeadTable T) {
  constexpr int COUT = CIN * OG, CB = COUT / 8;
  const BlHeadJob& J = T.job[blockIdx.z];
  const int b = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= J.l_out) return;
  const float* xb = J.x + (long long)b * CIN * J.l_in;
  float xv[CIN][K];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const int u = bl_head_src(t - J.pad + j * J.dil, J.rpad, J.l_in);
#pragma unroll
    for (int c = 0; c < CIN; ++c) xv[c][j] = u >= 0 ? xb[(long long)c * J.l_in + u] : 0.f;
  }
  cfloat_t v = (cfloat_t)J.v;
  cfloat_t sc = (cfloat_t)J.scale;
  cfloat_t bs = (cfloat_t)J.bias;
#pragma unroll
  for (int ob = 0; ob < CB; ++ob) {
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int co = 8 * ob + e, ci = co / OG;
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < K; ++j) a = fmaf(v[co * K + j], xv[ci][j], a);
      a = a * (J.scale ? sc[co] : 1.f) + (J.bias ? bs[co] : 0.f);
      acc[e] = lrelu(a, J.out_slope);
    }
    const long long idx = ((long long)b * CB + ob) * J.l_out + t;
    const u32x4 h = bl_pack8(acc);
    J.yh[idx] = h;
    if (J.yl) J.yl[idx] = bl_pack8_lo(acc, h);
  }
}

// Input gradient of the heads, summed over the jobs (the three PQMF-band chains share their input):
//   dxp[b, ci, p] = sum_{co in group ci} sum_j w[co, j] g[b, co, p + pad - j dil],   dx[u] = dxp[u + P] + reflect folds
// One block = 256 consecutive input positions of one batch item.  Per job the gradient tile those positions touch (hi + lo planes ->
// fp32, [channel][position]) is staged in LDS once -- every unit is read from memory once instead of K times -- and a thread's K * OG
// products per input channel read it with conflict-free ds_read_b32 (lanes = consecutive positions); weights through the scalar cache.
// The <= 2 P positions next to each end of a row receive their folded terms from a slow path on global memory.
template <int CIN, int OG, int K>
__global__ __launch_bounds__(256) void bl_head_dx_kernel(const BlHeadTable T, float* __restrict__ dx) {
  constexpr int COUT = CIN * OG, CB = COUT / 8;
  extern __shared__ __attribute__((aligned(16))) float gt[];   // [COUT][TL]
  const int l_in = T.job[0].l_in;
  const int b = blockIdx.y;
  const int u0 = blockIdx.x * 256;
  const int u = u0 + threadIdx.x;
  float acc[CIN];
#pragma unroll
  for (int c = 0; c < CIN; ++c) acc[c] = 0.f;
  for (int jb = 0; jb < T.n; ++jb) {
    const BlHeadJob& J = T.job[jb];
    const int P = J.rpad, span = (K - 1) * J.dil, TL = 256 + span;
    // gradient positions t = p + pad - j dil for p = u + P, u in the block: [tlo, tlo + TL)
    const int tlo = u0 + P + J.pad - span;
    __syncthreads();   // the previous job's tile has been consumed
    for (int i = threadIdx.x; i < CB * TL; i += 256) {
      const int cb = i / TL, r = i - cb * TL, t = tlo + r;
      float f[8];
      if (t >= 0 && t < J.l_out) bl_load8(J.yh, J.yl, ((long long)b * CB + cb) * J.l_out + t, f);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) gt[(8 * cb + e) * TL + r] = f[e];
    }
    __syncthreads();
    cfloat_t v = (cfloat_t)J.v;
    cfloat_t sc = (cfloat_t)J.scale;
    if (u < l_in) {
#pragma unroll
      for (int c = 0; c < CIN; ++c) {
        float a = 0.f;
#pragma unroll
        for (int m = 0; m < OG; ++m) {
          const int co = c * OG + m;
          float s = 0.f;
#pragma unroll
          for (int j = 0; j < K; ++j) s = fmaf(v[co * K + j], gt[co * TL + (int)threadIdx.x + span - j * J.dil], s);
          a = fmaf(s, J.scale ? sc[co] : 1.f, a);
        }
        acc[c] += a;
      }
      // folded terms of the reflection (nn.ReflectionPad1d's adjoint): padded positions P - u and P + 2 (l_in - 1) - u
      int pts[2], np = 0;
      if (u >= 1 && u <= P) pts[np++] = P - u;
      if (u >= l_in - 1 - P && u <= l_in - 2) pts[np++] = P + 2 * (l_in - 1) - u;
      for (int q = 0; q < np; ++q)
        for (int j = 0; j < K; ++j) {
          const int t = pts[q] + J.pad - j * J.dil;
          if (t < 0 || t >= J.l_out) continue;
          for (int cb = 0; cb < CB; ++cb) {
            float gv[8];
            bl_load8(J.yh, J.yl, ((long long)b * CB + cb) * J.l_out + t, gv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const int co = 8 * cb + e, ci = co / OG;
              const float w = J.v[co * K + j] * (J.scale ? J.scale[co] : 1.f) * gv[e];
#pragma unroll
              for (int c = 0; c < CIN; ++c) acc[c] += c == ci ? w : 0.f;
            }
          }
        }
    }
  }
  if (u < l_in) {
#pragma unroll
    for (int c = 0; c < CIN; ++c) dx[((long long)b * CIN + c) * l_in + u] = acc[c];
  }
}

// The one-channel head at dilation 1 (MelGAN: 1 -> 16, k 15), TWO consecutive positions per thread: the two outputs share a window of
// K + 1 staged samples per output channel, read as (K + 1) / 2 aligned ds_read_b64 (lane t reads floats 2 t .. of the row: contiguous over
// the wave) instead of 2 K ds_read_b32 -- the one-position form is bound by its 240 LDS reads per output.  The next channel's window is
// asked for before the current channel's products (two window register sets).  Same products in the same order per output as above.
template <int OG, int K>
__global__ __launch_bounds__(256) void bl_head_dx2_kernel(const BlHeadTable T, float* __restrict__ dx) {
  static_assert(K % 2 == 1, "the window of two positions, K + 1 samples, is a whole number of 8-byte reads");
  constexpr int COUT = OG, CB = COUT / 8, NW = (K + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) float gt2[];   // [COUT][TL], TL even
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef const __attribute__((address_space(3))) f32x2_t* lds_f2;
  const int l_in = T.job[0].l_in;
  const int b = blockIdx.y;
  const int u0 = blockIdx.x * 512;
  const int ua = u0 + 2 * (int)threadIdx.x;
  float acc[2] = {0.f, 0.f};
  for (int jb = 0; jb < T.n; ++jb) {
    const BlHeadJob& J = T.job[jb];
    const int P = J.rpad, span = K - 1, TL = 512 + span;
    const int tlo = u0 + P + J.pad - span;
    __syncthreads();
    for (int i = threadIdx.x; i < CB * TL; i += 256) {
      const int cb = i / TL, r = i - cb * TL, t = tlo + r;
      float f[8];
      if (t >= 0 && t < J.l_out) bl_load8(J.yh, J.yl, ((long long)b * CB + cb) * J.l_out + t, f);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) gt2[(8 * cb + e) * TL + r] = f[e];
    }
    __syncthreads();
    cfloat_t v = (cfloat_t)J.v;
    cfloat_t sc = (cfloat_t)J.scale;
    if (ua < l_in) {
      // window of output channel co: W[i] = g[co][tlo + 2 t + i], i = 0 .. K; output q (0, 1) takes tap j from W[K - 1 - j + q]
      const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)gt2 + (unsigned)(2 * threadIdx.x) * 4u;
      f32x2_t wa[NW], wb[NW];
#pragma unroll
      for (int i = 0; i < NW; ++i) wa[i] = ((lds_f2)(size_t)base)[i];
      float a0 = 0.f, a1 = 0.f;
      auto channel = [&](int co, const f32x2_t (&w)[NW]) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int j = 0; j < K; ++j) {
          const int i0 = K - 1 - j, i1 = K - j;
          const float wj = v[co * K + j];
          s0 = fmaf(wj, w[i0 >> 1][i0 & 1], s0);
          s1 = fmaf(wj, w[i1 >> 1][i1 & 1], s1);
        }
        const float scl = J.scale ? sc[co] : 1.f;
        a0 = fmaf(s0, scl, a0);
        a1 = fmaf(s1, scl, a1);
      };
      // channel pairs, NOT unrolled: fully unrolled hipcc hoists all sixteen windows (256 registers, one wave per SIMD)
#pragma unroll 1
      for (int co = 0; co < COUT; co += 2) {
#pragma unroll
        for (int i = 0; i < NW; ++i) wb[i] = ((lds_f2)(size_t)(base + (unsigned)((co + 1) * TL) * 4u))[i];
        channel(co, wa);
        const int nx = co + 2 < COUT ? co + 2 : co;   // the last pair re-reads its own first window (unused)
#pragma unroll
        for (int i = 0; i < NW; ++i) wa[i] = ((lds_f2)(size_t)(base + (unsigned)(nx * TL) * 4u))[i];
        channel(co + 1, wb);
      }
      acc[0] += a0;
      acc[1] += a1;
      // folded terms of the reflection, as above, for each of the two positions
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int u = ua + q;
        if (u >= l_in) continue;
        int pts[2], np = 0;
        if (u >= 1 && u <= P) pts[np++] = P - u;
        if (u >= l_in - 1 - P && u <= l_in - 2) pts[np++] = P + 2 * (l_in - 1) - u;
        for (int qq = 0; qq < np; ++qq)
          for (int j = 0; j < K; ++j) {
            const int t = pts[qq] + J.pad - j;
            if (t < 0 || t >= J.l_out) continue;
            for (int cb = 0; cb < CB; ++cb) {
              float gv[8];
              bl_load8(J.yh, J.yl, ((long long)b * CB + cb) * J.l_out + t, gv);
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const int co = 8 * cb + e;
                acc[q] += J.v[co * K + j] * (J.scale ? J.scale[co] : 1.f) * gv[e];
              }
            }
          }
      }
    }
  }
  if (ua < l_in) dx[(long long)b * l_in + ua] = acc[0];
  if (ua + 1 < l_in) dx[(long long)b * l_in + ua + 1] = acc[1];
}

// ([MI355X] tried: four consecutive positions per thread sharing a window of K + 3 samples per channel -- five ds_read_b128 for 4 K products
// instead of one ds_read_b32 per product: 109 -> 145 us for the MelGAN head; the one-position form keeps all 240 reads of a thread in flight
// (251 registers), the windowed one waits for each channel's window in turn.)
// Weight (+ bias) gradient of a head: slab[z][co][j] = sum over the (batch item, position) pairs of slice z of g[b, co, t] xp[b, ci(co), t - pad + j dil],
// column K = sum g.  One block = one output bundle x one slice; the 8 (K + 1) per-thread sums meet in a fixed-order block reduction.
template <int CIN, int OG, int K>
__global__ __launch_bounds__(256) void bl_head_dw_kernel(const u32x4* __restrict__ gh, const float* __restrict__ x, int rows, int l_in, int l_out,
                                                         int dil, int pad, int rpad, int nslab, float* __restrict__ slabs) {
  constexpr int COUT = CIN * OG, CB = COUT / 8;
  __shared__ float red[4][8 * (K + 1)];
  const int ob = blockIdx.x, z = blockIdx.y;
  const long long total = (long long)rows * l_out;
  const long long per = (total + nslab - 1) / nslab;
  const long long lo = (long long)z * per, hi = lo + per < total ? lo + per : total;
  float acc[8][K + 1];
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int j = 0; j <= K; ++j) acc[e][j] = 0.f;
  const int ci0 = (8 * ob) / OG, ci1 = (8 * ob + 7) / OG;   // OG >= 4: at most two input channels feed one bundle
  // two (batch item, position) pairs per iteration, every load of both issued before the first product: the loop is a chain of
  // dependent global round trips at two waves per SIMD (128 accumulators per thread) -- [MI355X] MelGAN head 165 us one pair at a time
  // ([MI355X] four / eight pairs per iteration for the 3-tap heads: 36.7 -> 37.0 / 39.6 us -- not what they wait for)
  constexpr int UN = 2;
  for (long long i0 = lo + threadIdx.x; i0 < hi; i0 += 256 * UN) {
    u32x4 gu[UN];
    float x0[UN][K], x1[CIN > 1 ? UN : 1][K];
#pragma unroll
    for (int un = 0; un < UN; ++un) {
      const long long i = i0 + 256 * un;
      const bool live = i < hi;
      const long long ic = live ? i : lo;
      const int b = (int)(ic / l_out), t = (int)(ic - (long long)b * l_out);
      gu[un] = gh[((long long)b * CB + ob) * l_out + t];
      if (!live) gu[un] = u32x4{0u, 0u, 0u, 0u};
      const float* xb = x + (long long)b * CIN * l_in;
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const int u = bl_head_src(t - pad + j * dil, rpad, l_in);
        x0[un][j] = u >= 0 ? xb[(long long)ci0 * l_in + u] : 0.f;
        if constexpr (CIN > 1) x1[un][j] = (u >= 0 && ci1 != ci0) ? xb[(long long)ci1 * l_in + u] : x0[un][j];
      }
    }
#pragma unroll
    for (int un = 0; un < UN; ++un) {
      float gv[8];
      bl_unpack8(gu[un], gv);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e][K] += gv[e];
#pragma unroll
      for (int j = 0; j < K; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e][j] = fmaf(gv[e], (CIN == 1 || (8 * ob + e) / OG == ci0) ? x0[un][j] : x1[CIN > 1 ? un : 0][j], acc[e][j]);
    }
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int j = 0; j <= K; ++j) {
      const float sm = EDGE_WAVE_SUM(acc[e][j]);
      if (lane == 0) red[w][e * (K + 1) + j] = sm;
    }
  __syncthreads();
  for (int i = threadIdx.x; i < 8 * (K + 1); i += 256)
    slabs[((long long)z * COUT + 8 * ob + i / (K + 1)) * (K + 1) + i % (K + 1)] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
}

// ---- chain tails (the logits layer: C -> 1, k taps, zero padding) ------------------------------------------------------------------
// forward: one block = 64 output positions of one batch item x 16 waves; wave w sums the bundles cb = w, w + 16, ... (fp32 FMA on
// hi + lo), the 16 partial sums meet in LDS; weights staged once per block as [bundle][tap][8] (two ds_read_b128 per unit)
__global__ __launch_bounds__(1024) void bl_tail_fwd_kernel(const u32x4* __restrict__ xh, const u32x4* __restrict__ xl, int CB, int L, int k, int pad,
                                                           int l_out, const float* __restrict__ v, const float* __restrict__ scale,
                                                           const float* __restrict__ bias, float out_slope, float* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) float wt[];   // [CB][k][8] scaled weights, then 16 x 64 partial sums
  float* part = wt + CB * 8 * k;
  const float sc = scale ? scale[0] : 1.f;
  for (int i = threadIdx.x; i < CB * 8 * k; i += 1024) {
    const int cb = i / (8 * k), r = i - cb * 8 * k, j = r >> 3, e = r & 7;
    wt[i] = v[(cb * 8 + e) * k + j] * sc;
  }
  __syncthreads();
  const int b = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + lane;
  float acc = 0.f;
  if (t < l_out) {
    for (int cb = w; cb < CB; cb += 16) {
      const long long row = ((long long)b * CB + cb) * L;
      for (int j = 0; j < k; ++j) {
        const int q = t - pad + j;
        const int qc = q < 0 ? 0 : (q >= L ? L - 1 : q);
        float f[8];
        bl_load8(xh, xl, row + qc, f);
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wt + (cb * k + j) * 8), w1 = *reinterpret_cast<const f32x4*>(wt + (cb * k + j) * 8 + 4);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) s = fmaf(w0[e], f[e], fmaf(w1[e], f[4 + e], s));
        acc += (q == qc) ? s : 0.f;
      }
    }
  }
  part[w * 64 + lane] = acc;
  __syncthreads();
  if (w == 0 && t < l_out) {
    float s = bias ? bias[0] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += part[i * 64 + lane];
    y[(long long)b * l_out + t] = lrelu(s, out_slope);
  }
}

// The same for k = K taps known at compile time: the loop above waits for each (bundle, tap)'s two unit loads in turn -- 18-24 dependent
// round trips per thread.  Here a bundle's 2 K loads go out together and the next bundle's are in flight while the current one is
// summed (two register sets, bundle pairs per iteration).  Same products and sums in the same order.
template <int K>
__global__ __launch_bounds__(1024) void bl_tail_fwd_k_kernel(const u32x4* __restrict__ xh, const u32x4* __restrict__ xl, int CB, int L, int pad,
                                                             int l_out, const float* __restrict__ v, const float* __restrict__ scale,
                                                             const float* __restrict__ bias, float out_slope, float* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) float wt[];   // [CB][K][8] scaled weights, then 16 x 64 partial sums
  float* part = wt + CB * 8 * K;
  const float sc = scale ? scale[0] : 1.f;
  for (int i = threadIdx.x; i < CB * 8 * K; i += 1024) {
    const int cb = i / (8 * K), r = i - cb * 8 * K, j = r >> 3, e = r & 7;
    wt[i] = v[(cb * 8 + e) * K + j] * sc;
  }
  __syncthreads();
  const int b = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + lane;
  float acc = 0.f;
  if (t < l_out) {
    u32x4 ha[K], la[K], hb[K], lb[K];
    auto issue = [&](int cb, u32x4 (&h)[K], u32x4 (&l)[K]) {
      const long long row = ((long long)b * CB + cb) * L;
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const int q = t - pad + j;
        const int qc = q < 0 ? 0 : (q >= L ? L - 1 : q);
        h[j] = xh[row + qc];
        l[j] = xl[row + qc];
      }
    };
    auto sum = [&](int cb, const u32x4 (&h)[K], const u32x4 (&l)[K]) {
#pragma unroll
      for (int j = 0; j < K; ++j) {
        float f[8], g[8];
        bl_unpack8(h[j], f);
        bl_unpack8(l[j], g);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] += g[e];
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wt + (cb * K + j) * 8), w1 = *reinterpret_cast<const f32x4*>(wt + (cb * K + j) * 8 + 4);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) s = fmaf(w0[e], f[e], fmaf(w1[e], f[4 + e], s));
        const int q = t - pad + j;
        acc += (q >= 0 && q < L) ? s : 0.f;
      }
    };
    if (w < CB) issue(w, ha, la);
#pragma unroll 1
    for (int cb = w; cb < CB; cb += 32) {
      const bool second = cb + 16 < CB;
      if (second) issue(cb + 16, hb, lb);
      sum(cb, ha, la);
      if (cb + 32 < CB) issue(cb + 32, ha, la);
      if (second) sum(cb + 16, hb, lb);
    }
  }
  part[w * 64 + lane] = acc;
  __syncthreads();
  if (w == 0 && t < l_out) {
    float s = bias ? bias[0] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += part[i * 64 + lane];
    y[(long long)b * l_out + t] = lrelu(s, out_slope);
  }
}

// input gradient of the logits layer, with the epilogue of the engine's stacked backward:
//   g[b, c, t] = ( sum_j w[c, j] seed[b, t + pad - j] + (b < fm_rows ? fm term of the embedding : 0) ) * lrelu'(act[map(b), c, t])
struct BlTailDxArgs {
  const float* seeds; const float* v; const float* scale;
  const u32x4* ah; const u32x4* al;
  u32x4* gh; u32x4* gl;
  const float* fm_sums; float fm_gs, mask_slope;
  int rows, CB, L, k, pad, l_out, fm_rows, ref_off, seg, map[4];
};
__global__ __launch_bounds__(256) void bl_tail_dx_kernel(const BlTailDxArgs P) {
  __shared__ float wsh[8 * 8];
  const int cb = blockIdx.y, b = blockIdx.z;
  const float sc = P.scale ? P.scale[0] : 1.f;
  if (threadIdx.x < 8 * P.k) wsh[threadIdx.x] = P.v[(long long)cb * 8 * P.k + threadIdx.x] * sc;   // [e][j], k <= 8
  __syncthreads();
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= P.L) return;
  float val[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) val[e] = 0.f;
  const float* sd = P.seeds + (long long)b * P.l_out;
  for (int j = 0; j < P.k; ++j) {
    const int q = t + P.pad - j;
    const float s = (q >= 0 && q < P.l_out) ? sd[q] : 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) val[e] = fmaf(wsh[e * P.k + j], s, val[e]);
  }
  const int eb = P.seg > 0 ? P.map[b / P.seg] * P.seg + b % P.seg : b;
  float a0[8];
  const long long ai = ((long long)eb * P.CB + cb) * P.L + t;
  bl_unpack8(P.ah[ai], a0);
  if (P.fm_sums != nullptr && b < P.fm_rows) {
    const float s1 = P.fm_sums[0], s2 = P.fm_sums[1];
    const float k1 = P.fm_gs / s2, k2 = P.fm_gs * s1 / (s2 * s2);
    float a1[8], r[8];
    bl_unpack8(P.al[ai], a1);
    bl_load8(P.ah, P.al, ((long long)(b + P.ref_off) * P.CB + cb) * P.L + t, r);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float av = a0[e] + a1[e], dv = av - r[e];
      val[e] += k1 * (float)((dv > 0.f) - (dv < 0.f)) - k2 * (float)((av > 0.f) - (av < 0.f));
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) val[e] *= dlrelu(a0[e], P.mask_slope);
  const long long gi = ((long long)b * P.CB + cb) * P.L + t;
  const u32x4 h = bl_pack8(val);
  P.gh[gi] = h;
  if (P.gl) P.gl[gi] = bl_pack8_lo(val, h);
}

// weight (+ bias) gradient of the logits layer, one launch for the `nbranch` hinge branches (seed rows / embedding rows br * rows ..):
//   slab[br][z][c k + j] = sum_{(b, t) in slice z of branch br} seed[b, t] x[b, c, t - pad + j];   column C k = sum seed
// One block = one input bundle x one slice x one branch: 8 k + 1 sums per thread, fixed-order block reduction.
__global__ __launch_bounds__(256) void bl_tail_dw_kernel(const float* __restrict__ seeds, const u32x4* __restrict__ xh, const u32x4* __restrict__ xl,
                                                         int rows, int CB, int L, int k, int pad, int l_out, int nslab, float* __restrict__ slabs) {
  __shared__ float red[4][8 * 8 + 1];
  const int cb = blockIdx.x, z = blockIdx.y, br = blockIdx.z;
  const long long total = (long long)rows * l_out;
  const long long per = (total + nslab - 1) / nslab;
  const long long lo = (long long)z * per, hi = lo + per < total ? lo + per : total;
  const float* sd = seeds + (long long)br * total;
  float acc[8][8], sb = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[e][j] = 0.f;
  for (long long i = lo + threadIdx.x; i < hi; i += 256) {
    const int b = (int)(i / l_out), t = (int)(i - (long long)b * l_out);
    const float s = sd[i];
    sb += s;
    const long long row = ((long long)(br * rows + b) * CB + cb) * L;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j < k) {
        const int q = t - pad + j;
        const int qc = q < 0 ? 0 : (q >= L ? L - 1 : q);
        float f[8];
        bl_load8(xh, xl, row + qc, f);
        const float sq = q == qc ? s : 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e][j] = fmaf(sq, f[e], acc[e][j]);
      }
    }
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float s = wave_sum(acc[e][j]);
      if (lane == 0) red[w][e * 8 + j] = s;
    }
  sb = wave_sum(sb);
  if (lane == 0) red[w][64] = sb;
  __syncthreads();
  const long long rs = (long long)CB * 8 * k + 1;
  float* out = slabs + ((long long)br * nslab + z) * rs;
  for (int i = threadIdx.x; i < 65; i += 256) {
    const float s = red[0][i] + red[1][i] + red[2][i] + red[3][i];
    if (i == 64) { if (cb == 0) out[rs - 1] = s; }
    else {
      const int e = i >> 3, j = i & 7;
      if (j < k) out[(long long)(cb * 8 + e) * k + j] = s;
    }
  }
}

// ---- feature-matching sums over bundle planes -------------------------------------------------------------------------------------
constexpr int BLFM_MAX_PAIRS = 32;
constexpr int BLFM_BLOCKS = 1024;   // == FM_BLOCKS of direct.hip: the partial sums are finished by the same fixed-order pass
struct BlFmTable {
  const u32x4* hi[BLFM_MAX_PAIRS]; const u32x4* lo[BLFM_MAX_PAIRS];
  long long units[BLFM_MAX_PAIRS];   // units of the enhanced rows; the reference rows follow them in the same planes
  uint2* code[BLFM_MAX_PAIRS];       // nullable: one byte per element of the enhanced rows, the signs the input gradients' feature-matching term needs
};
// code byte of one element: bits 0-1 = sgn(a - r) + 1, bits 2-3 = sgn(a) + 1 (a = hi + lo of the enhanced row, r of the reference row, both
// sums in fp32 as the input-gradient epilogues form them).  A unit of 8 channels = 8 bytes, at the index of the unit in the hi plane.
__device__ __forceinline__ unsigned bl_fm_code(float av, float rv) {
  const float dv = av - rv;
  return (unsigned)((int)(dv > 0.f) - (int)(dv < 0.f) + 1) | ((unsigned)((int)(av > 0.f) - (int)(av < 0.f) + 1) << 2);
}
__device__ __forceinline__ uint2 bl_fm_code8(const float (&av)[8], const float (&rv)[8]) {
  uint2 c;
  c.x = bl_fm_code(av[0], rv[0]) | (bl_fm_code(av[1], rv[1]) << 8) | (bl_fm_code(av[2], rv[2]) << 16) | (bl_fm_code(av[3], rv[3]) << 24);
  c.y = bl_fm_code(av[4], rv[4]) | (bl_fm_code(av[5], rv[5]) << 8) | (bl_fm_code(av[6], rv[6]) << 16) | (bl_fm_code(av[7], rv[7]) << 24);
  return c;
}
__global__ __launch_bounds__(256) void bl_fm_partial_kernel(const BlFmTable T, float* __restrict__ partial) {
  __shared__ float red[4];
  const int p = blockIdx.y;
  const u32x4* hi = T.hi[p];
  const u32x4* lo = T.lo[p];
  const long long n = T.units[p];
  uint2* code = T.code[p];
  const long long per = (n + BLFM_BLOCKS - 1) / BLFM_BLOCKS;
  const long long b0 = (long long)blockIdx.x * per;
  const long long b1 = b0 + per < n ? b0 + per : n;
  float s1 = 0.f, s2 = 0.f;
  // two units per iteration, the eight 16-byte loads issued before the first use (four per iteration reached 4.2 TB/s of the 1.6 GB)
  long long i = b0 + threadIdx.x;
  for (; i + 256 < b1; i += 512) {
    const u32x4 ah0 = hi[i], al0 = lo[i], rh0 = hi[i + n], rl0 = lo[i + n];
    const u32x4 ah1 = hi[i + 256], al1 = lo[i + 256], rh1 = hi[i + 256 + n], rl1 = lo[i + 256 + n];
    float a[8], l[8], r[8], q[8];
    bl_unpack8(ah0, a); bl_unpack8(al0, l); bl_unpack8(rh0, r); bl_unpack8(rl0, q);
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] += l[e]; r[e] += q[e]; s1 += fabsf(a[e] - r[e]); s2 += fabsf(a[e]); }
    if (code) code[i] = bl_fm_code8(a, r);
    bl_unpack8(ah1, a); bl_unpack8(al1, l); bl_unpack8(rh1, r); bl_unpack8(rl1, q);
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] += l[e]; r[e] += q[e]; s1 += fabsf(a[e] - r[e]); s2 += fabsf(a[e]); }
    if (code) code[i + 256] = bl_fm_code8(a, r);
  }
  for (; i < b1; i += 256) {
    float a[8], r[8];
    bl_load8(hi, lo, i, a);
    bl_load8(hi, lo, i + n, r);
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1 += fabsf(a[e] - r[e]); s2 += fabsf(a[e]); }
    if (code) code[i] = bl_fm_code8(a, r);
  }
  s1 = block_sum_256(s1, red);
  s2 = block_sum_256(s2, red);
  if (threadIdx.x == 0) {
    partial[(p * BLFM_BLOCKS + blockIdx.x) * 2 + 0] = s1;
    partial[(p * BLFM_BLOCKS + blockIdx.x) * 2 + 1] = s2;
  }
}
__global__ __launch_bounds__(64) void bl_fm_final_kernel(const float* __restrict__ partial, float* __restrict__ sums) {
  const int p = blockIdx.x, l = threadIdx.x;
  float s1 = 0.f, s2 = 0.f;
  for (int k = l; k < BLFM_BLOCKS; k += 64) {   // fixed order: deterministic
    s1 += partial[(p * BLFM_BLOCKS + k) * 2 + 0];
    s2 += partial[(p * BLFM_BLOCKS + k) * 2 + 1];
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if (l == 0) { sums[2 * p] = s1; sums[2 * p + 1] = s2; }
}

static unsigned bl_grid(long long n, int cap = 8192) {
  long long b = (n + 255) / 256;
  if (b > cap) b = cap;
  return (unsigned)(b < 1 ? 1 : b);
}

}  // namespace eben

using namespace eben;

extern "C" int eben_bl_from_f32(const float* x, int batch, int channels, int length, void* hi, void* lo, void* stream) {
  EBEN_REQUIRE(x && hi && batch > 0 && channels > 0 && length > 0 && channels % 8 == 0, "bl_from_f32: channels must be a positive multiple of 8");
  const long long units = (long long)batch * (channels / 8) * length;
  hipLaunchKernelGGL(bl_from_f32_kernel, dim3(bl_grid(units)), dim3(256), 0, as_stream(stream), x, units, channels / 8, length,
                     static_cast<u32x4*>(hi), static_cast<u32x4*>(lo));
  EBEN_CHECK_LAUNCH("bl_from_f32_kernel");
  return EBEN_OK;
}

extern "C" int eben_bl_to_f32(const void* hi, const void* lo, int batch, int channels, int length, float* x, void* stream) {
  EBEN_REQUIRE(x && hi && batch > 0 && channels > 0 && length > 0 && channels % 8 == 0, "bl_to_f32: channels must be a positive multiple of 8");
  const long long units = (long long)batch * (channels / 8) * length;
  hipLaunchKernelGGL(bl_to_f32_kernel, dim3(bl_grid(units)), dim3(256), 0, as_stream(stream), static_cast<const u32x4*>(hi),
                     static_cast<const u32x4*>(lo), units, length, x);
  EBEN_CHECK_LAUNCH("bl_to_f32_kernel");
  return EBEN_OK;
}

static int bl_head_table(const EbenBlHeadJob* jobs, int n, int batch, bool backward, BlHeadTable* T) {
  EBEN_REQUIRE(jobs && n >= 1 && n <= BL_HEAD_JOBS && batch > 0, "1..%d head jobs", BL_HEAD_JOBS);
  T->n = n; T->batch = batch;
  for (int i = 0; i < n; ++i) {
    const EbenBlHeadJob& s = jobs[i];
    BlHeadJob& d = T->job[i];
    EBEN_REQUIRE(s.v && s.y_hi && (backward || s.x), "null pointer in head job %d", i);
    EBEN_REQUIRE(s.dilation >= 1 && s.pad >= 0 && s.reflect_pad >= 0 && s.reflect_pad < s.l_in, "head job %d: bad geometry", i);
    EBEN_REQUIRE(s.l_out == s.l_in + 2 * s.reflect_pad + 2 * s.pad - s.dilation * (s.ksize - 1) && s.l_out > 0, "head job %d: l_out %d does not match the layer", i, s.l_out);
    EBEN_REQUIRE(s.c_in == jobs[0].c_in && s.c_out == jobs[0].c_out && s.ksize == jobs[0].ksize && s.l_in == jobs[0].l_in,
                 "the head jobs of one launch must share channels, taps and the input length");
    d.x = s.x; d.v = s.v; d.scale = s.scale; d.bias = s.bias; d.yh = static_cast<u32x4*>(s.y_hi); d.yl = static_cast<u32x4*>(s.y_lo);
    d.c_in = s.c_in; d.c_out = s.c_out; d.l_in = s.l_in; d.l_out = s.l_out; d.k = s.ksize; d.dil = s.dilation; d.pad = s.pad; d.rpad = s.reflect_pad;
    d.out_slope = s.out_slope;
  }
  return EBEN_OK;
}

// the two head shapes of DiscriminatorEBENMultiScales: 0 = PQMF-band (4 -> 24, k 3), 1 = MelGAN (1 -> 16, k 15); -1: not built
static int bl_head_shape(const EbenBlHeadJob& j) {
  if (j.c_in == 4 && j.c_out == 24 && j.ksize == 3) return 0;
  if (j.c_in == 1 && j.c_out == 16 && j.ksize == 15) return 1;
  return -1;
}

extern "C" int eben_bl_head_fwd(const EbenBlHeadJob* jobs, int njobs, int batch, void* stream) {
  BlHeadTable T;
  int rc = bl_head_table(jobs, njobs, batch, false, &T);
  if (rc) return rc;
  int lmax = 0;
  for (int i = 0; i < njobs; ++i) if (jobs[i].l_out > lmax) lmax = jobs[i].l_out;
  const dim3 grid(ceil_div(lmax, 256), batch, njobs);
  switch (bl_head_shape(jobs[0])) {
    case 0: hipLaunchKernelGGL((bl_head_fwd_kernel<4, 6, 3>), grid, dim3(256), 0, as_stream(stream), T); break;
    case 1: hipLaunchKernelGGL((bl_head_fwd_kernel<1, 16, 15>), grid, dim3(256), 0, as_stream(stream), T); break;
    default: return fail(EBEN_EUNSUPPORTED, "chain head %d -> %d, k %d: not one of the built shapes", jobs[0].c_in, jobs[0].c_out, jobs[0].ksize);
  }
  EBEN_CHECK_LAUNCH("bl_head_fwd_kernel");
  return EBEN_OK;
}

extern "C" int eben_bl_head_dx(const EbenBlHeadJob* jobs, int njobs, int rows, float* dx, void* stream) {
  BlHeadTable T;
  int rc = bl_head_table(jobs, njobs, rows, true, &T);
  if (rc) return rc;
  EBEN_REQUIRE(dx != nullptr, "null dx");
  int span = 0;
  for (int i = 0; i < njobs; ++i) if ((jobs[i].ksize - 1) * jobs[i].dilation > span) span = (jobs[i].ksize - 1) * jobs[i].dilation;
  const size_t lds = sizeof(float) * (size_t)jobs[0].c_out * (256 + span);
  EBEN_REQUIRE(lds <= 64 * 1024, "head input gradient: the gradient tile does not fit");
  const dim3 grid(ceil_div(jobs[0].l_in, 256), rows);
  static const int two_pos = getenv("EBEN_HEAD_DX2") ? atoi(getenv("EBEN_HEAD_DX2")) : 1;
  bool unit_dil = true;
  for (int i = 0; i < njobs; ++i) unit_dil = unit_dil && jobs[i].dilation == 1;
  switch (bl_head_shape(jobs[0])) {
    case 0: hipLaunchKernelGGL((bl_head_dx_kernel<4, 6, 3>), grid, dim3(256), lds, as_stream(stream), T, dx); break;
    case 1:
      if (two_pos && unit_dil) {
        hipLaunchKernelGGL((bl_head_dx2_kernel<16, 15>), dim3(ceil_div(jobs[0].l_in, 512), rows), dim3(256), sizeof(float) * 16 * (512 + 14), as_stream(stream), T, dx);
        EBEN_CHECK_LAUNCH("bl_head_dx2_kernel");
        return EBEN_OK;
      }
      hipLaunchKernelGGL((bl_head_dx_kernel<1, 16, 15>), grid, dim3(256), lds, as_stream(stream), T, dx); break;
    default: return fail(EBEN_EUNSUPPORTED, "chain head %d -> %d, k %d: not one of the built shapes", jobs[0].c_in, jobs[0].c_out, jobs[0].ksize);
  }
  EBEN_CHECK_LAUNCH("bl_head_dx_kernel");
  return EBEN_OK;
}

// split-K slices of a head's weight gradient: ~2048 (batch item, position) pairs per thread block column
static int bl_head_slabs(long long pairs) {
  // [MI355X] 24 pairs per thread put the 3-tap heads on 249 blocks (one per CU, a dozen dependent round trips each): 36.0 -> 21.1 us at 6;
  // the 15-tap head sits at the cap either way (more, shorter blocks lose to their fixed cost: 82 -> 108 us at 512 slabs)
  static const int per_thread = getenv("EBEN_HEAD_DW_PAIRS") ? atoi(getenv("EBEN_HEAD_DW_PAIRS")) : 6;
  static const int cap = getenv("EBEN_HEAD_DW_SLABS") ? atoi(getenv("EBEN_HEAD_DW_SLABS")) : 256;
  long long n = pairs / (256 * (per_thread > 0 ? per_thread : 24));
  return (int)(n < 32 ? 32 : (n > cap ? cap : n));
}
extern "C" size_t eben_bl_head_dw_workspace(const EbenBlHeadJob* job, int rows, int* nslab, int* row_stride) {
  if (!job || rows <= 0) return 0;
  const int ns = bl_head_slabs((long long)rows * job->l_out);
  if (nslab) *nslab = ns;
  if (row_stride) *row_stride = job->ksize + 1;
  return sizeof(float) * (size_t)ns * job->c_out * (job->ksize + 1);
}

// job->y_hi: the gradient at the head's output (rows x c_out x l_out, hi plane); job->x: the head's input rows it is paired with
extern "C" int eben_bl_head_dw(const EbenBlHeadJob* job, int rows, float* slabs, size_t ws_bytes, void* stream) {
  BlHeadTable T;
  int rc = bl_head_table(job, 1, rows, false, &T);
  if (rc) return rc;
  EBEN_REQUIRE(slabs != nullptr, "null slabs");
  int ns = 0;
  if (ws_bytes < eben_bl_head_dw_workspace(job, rows, &ns, nullptr)) return fail(EBEN_EWORKSPACE, "bl_head_dw workspace too small");
  const BlHeadJob& J = T.job[0];
  const dim3 grid(J.c_out / 8, ns);
  switch (bl_head_shape(*job)) {
    case 0: hipLaunchKernelGGL((bl_head_dw_kernel<4, 6, 3>), grid, dim3(256), 0, as_stream(stream), J.yh, J.x, rows, J.l_in, J.l_out, J.dil, J.pad, J.rpad, ns, slabs); break;
    case 1: hipLaunchKernelGGL((bl_head_dw_kernel<1, 16, 15>), grid, dim3(256), 0, as_stream(stream), J.yh, J.x, rows, J.l_in, J.l_out, J.dil, J.pad, J.rpad, ns, slabs); break;
    default: return fail(EBEN_EUNSUPPORTED, "chain head %d -> %d, k %d: not one of the built shapes", job->c_in, job->c_out, job->ksize);
  }
  EBEN_CHECK_LAUNCH("bl_head_dw_kernel");
  return EBEN_OK;
}

extern "C" int eben_bl_tail_fwd(const void* x_hi, const void* x_lo, int batch, int channels, int length, int ksize, int pad, const float* v,
                                const float* scale, const float* bias, float out_slope, float* y, void* stream) {
  EBEN_REQUIRE(x_hi && v && y && batch > 0 && channels % 8 == 0 && channels > 0 && length > 0 && ksize >= 1 && ksize <= 8 && pad >= 0, "bad tail forward arguments");
  const int l_out = length + 2 * pad - (ksize - 1);
  EBEN_REQUIRE(l_out > 0, "tail forward: empty output");
  const size_t lds = sizeof(float) * ((size_t)channels * ksize + 16 * 64);
  EBEN_REQUIRE(lds <= 64 * 1024, "tail forward: %d channels x %d taps exceed the weight buffer", channels, ksize);
  static const int pipelined = getenv("EBEN_TAIL_FWD_K") ? atoi(getenv("EBEN_TAIL_FWD_K")) : 1;
  if (ksize == 3 && pipelined && x_lo) {
    hipLaunchKernelGGL((bl_tail_fwd_k_kernel<3>), dim3(ceil_div(l_out, 64), batch), dim3(1024), lds, as_stream(stream), static_cast<const u32x4*>(x_hi),
                       static_cast<const u32x4*>(x_lo), channels / 8, length, pad, l_out, v, scale, bias, out_slope, y);
    EBEN_CHECK_LAUNCH("bl_tail_fwd_k_kernel");
    return EBEN_OK;
  }
  hipLaunchKernelGGL(bl_tail_fwd_kernel, dim3(ceil_div(l_out, 64), batch), dim3(1024), lds, as_stream(stream), static_cast<const u32x4*>(x_hi),
                     static_cast<const u32x4*>(x_lo), channels / 8, length, ksize, pad, l_out, v, scale, bias, out_slope, y);
  EBEN_CHECK_LAUNCH("bl_tail_fwd_kernel");
  return EBEN_OK;
}

extern "C" int eben_bl_tail_dx(const float* seeds, int rows, int channels, int length, int ksize, int pad, const float* v, const float* scale,
                               const void* act_hi, const void* act_lo, float mask_slope, int seg, const int* seg_map, int fm_rows,
                               int ref_row_offset, const float* fm_sums, float fm_gs, void* g_hi, void* g_lo, void* stream) {
  EBEN_REQUIRE(seeds && v && act_hi && g_hi && rows > 0 && channels % 8 == 0 && channels > 0 && length > 0 && ksize >= 1 && ksize <= 8, "bad tail input-gradient arguments");
  EBEN_REQUIRE(seg >= 0 && (seg == 0 || (seg_map && rows <= 4 * seg)), "bad batch segment map");
  EBEN_REQUIRE(fm_rows == 0 || (fm_sums && act_lo && fm_rows > 0), "feature-matching rows need the sums and both planes of the embedding");
  BlTailDxArgs P;
  P.seeds = seeds; P.v = v; P.scale = scale; P.ah = static_cast<const u32x4*>(act_hi); P.al = static_cast<const u32x4*>(act_lo);
  P.gh = static_cast<u32x4*>(g_hi); P.gl = static_cast<u32x4*>(g_lo);
  P.fm_sums = fm_rows > 0 ? fm_sums : nullptr; P.fm_gs = fm_gs; P.mask_slope = mask_slope;
  P.rows = rows; P.CB = channels / 8; P.L = length; P.k = ksize; P.pad = pad; P.l_out = length + 2 * pad - (ksize - 1);
  P.fm_rows = fm_rows; P.ref_off = ref_row_offset; P.seg = seg;
  for (int i = 0; i < 4; ++i) P.map[i] = (seg > 0 && seg_map) ? seg_map[i] : i;
  hipLaunchKernelGGL(bl_tail_dx_kernel, dim3(ceil_div(length, 256), channels / 8, rows), dim3(256), 0, as_stream(stream), P);
  EBEN_CHECK_LAUNCH("bl_tail_dx_kernel");
  return EBEN_OK;
}

// The same for k = K taps known at compile time (every logits layer of the reference: K = 3), UN (batch item, position) pairs per
// iteration with all 2 K UN unit loads issued before the first product: the loop above is a chain of dependent global round trips
// (~10 per thread, six loads each) -- [MI355X] kernel time 768 -> 1, L 250: 37.8 -> 20.6 us; 1024 -> 1, L 125: 43.1 -> 16.2 us.  Same sums in the same order per thread.
template <int K, int UN>
__global__ __launch_bounds__(256) void bl_tail_dw_k_kernel(const float* __restrict__ seeds, const u32x4* __restrict__ xh, const u32x4* __restrict__ xl,
                                                           int rows, int CB, int L, int pad, int l_out, int nslab, float* __restrict__ slabs) {
  __shared__ float red[4][8 * K + 1];
  const int cb = blockIdx.x, z = blockIdx.y, br = blockIdx.z;
  const long long total = (long long)rows * l_out;
  const long long per = (total + nslab - 1) / nslab;
  const long long lo = (long long)z * per, hi = lo + per < total ? lo + per : total;
  const float* sd = seeds + (long long)br * total;
  float acc[8][K], sb = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int j = 0; j < K; ++j) acc[e][j] = 0.f;
  for (long long i0 = lo + threadIdx.x; i0 < hi; i0 += 256 * UN) {
    u32x4 uh[UN][K], ul[UN][K];
    float sq[UN][K], sv[UN];
#pragma unroll
    for (int un = 0; un < UN; ++un) {
      const long long i = i0 + 256 * un;
      const bool live = i < hi;
      const long long ic = live ? i : lo;
      const int b = (int)(ic / l_out), t = (int)(ic - (long long)b * l_out);
      const float s = live ? sd[ic] : 0.f;
      sv[un] = s;
      const long long row = ((long long)(br * rows + b) * CB + cb) * L;
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const int q = t - pad + j;
        const int qc = q < 0 ? 0 : (q >= L ? L - 1 : q);
        uh[un][j] = xh[row + qc];
        ul[un][j] = xl ? xl[row + qc] : u32x4{0u, 0u, 0u, 0u};
        sq[un][j] = q == qc ? s : 0.f;
      }
    }
#pragma unroll
    for (int un = 0; un < UN; ++un) {
      sb += sv[un];
#pragma unroll
      for (int j = 0; j < K; ++j) {
        float f[8], l[8];
        bl_unpack8(uh[un][j], f);
        bl_unpack8(ul[un][j], l);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e][j] = fmaf(sq[un][j], f[e] + l[e], acc[e][j]);
      }
    }
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const float s = wave_sum(acc[e][j]);
      if (lane == 0) red[w][e * K + j] = s;
    }
  sb = wave_sum(sb);
  if (lane == 0) red[w][8 * K] = sb;
  __syncthreads();
  const long long rs = (long long)CB * 8 * K + 1;
  float* out = slabs + ((long long)br * nslab + z) * rs;
  for (int i = threadIdx.x; i < 8 * K + 1; i += 256) {
    const float s = red[0][i] + red[1][i] + red[2][i] + red[3][i];
    if (i == 8 * K) { if (cb == 0) out[rs - 1] = s; }
    else out[(long long)(cb * 8 + i / K) * K + i % K] = s;
  }
}

// split-K slices of a logits layer's weight gradient: ~8 (batch item, position) pairs per thread ([MI355X] 2 pairs per thread, four times
// the blocks and slabs: 54 -> 80 us per launch)
static int bl_tail_slabs(long long pairs) {
  static const int per_thread = getenv("EBEN_TAIL_DW_PAIRS") ? atoi(getenv("EBEN_TAIL_DW_PAIRS")) : 8;
  long long n = pairs / (256 * (per_thread > 0 ? per_thread : 8));
  return (int)(n < 1 ? 1 : (n > 32 ? 32 : n));
}
extern "C" size_t eben_bl_tail_dw_workspace(int rows, int channels, int length, int ksize, int nbranch, int* nslab, int* row_stride) {
  if (rows <= 0 || channels <= 0 || length <= 0 || ksize <= 0 || nbranch <= 0) return 0;
  const int ns = bl_tail_slabs((long long)rows * length);
  if (nslab) *nslab = ns;
  if (row_stride) *row_stride = channels * ksize + 1;
  return sizeof(float) * (size_t)nbranch * ns * ((size_t)channels * ksize + 1);
}

// `nbranch` branches in one launch: seed rows / embedding rows [br * rows, (br + 1) * rows) of the given pointers, slabs [br][nslab][row]
extern "C" int eben_bl_tail_dw(const float* seeds, const void* x_hi, const void* x_lo, int rows, int nbranch, int channels, int length, int ksize, int pad,
                               float* slabs, size_t ws_bytes, void* stream) {
  EBEN_REQUIRE(seeds && x_hi && slabs && rows > 0 && nbranch > 0 && channels % 8 == 0 && channels > 0 && length > 0 && ksize >= 1 && ksize <= 8,
               "bad tail weight-gradient arguments");
  const int l_out = length + 2 * pad - (ksize - 1);
  int ns = 0;
  if (ws_bytes < eben_bl_tail_dw_workspace(rows, channels, l_out, ksize, nbranch, &ns, nullptr)) return fail(EBEN_EWORKSPACE, "bl_tail_dw workspace too small");
  static const int un = getenv("EBEN_TAIL_DW_UN") ? atoi(getenv("EBEN_TAIL_DW_UN")) : 4;
  if (ksize == 3 && un > 1) {
    hipLaunchKernelGGL((bl_tail_dw_k_kernel<3, 4>), dim3(channels / 8, ns, nbranch), dim3(256), 0, as_stream(stream), seeds, static_cast<const u32x4*>(x_hi),
                       static_cast<const u32x4*>(x_lo), rows, channels / 8, length, pad, l_out, ns, slabs);
    EBEN_CHECK_LAUNCH("bl_tail_dw_k_kernel");
    return EBEN_OK;
  }
  hipLaunchKernelGGL(bl_tail_dw_kernel, dim3(channels / 8, ns, nbranch), dim3(256), 0, as_stream(stream), seeds, static_cast<const u32x4*>(x_hi),
                     static_cast<const u32x4*>(x_lo), rows, channels / 8, length, ksize, pad, l_out, ns, slabs);
  EBEN_CHECK_LAUNCH("bl_tail_dw_kernel");
  return EBEN_OK;
}

extern "C" size_t eben_bl_fm_sums_workspace(int npairs) { return sizeof(float) * 2 * BLFM_BLOCKS * (size_t)(npairs > 0 ? npairs : 0); }

extern "C" int eben_bl_fm_sums(const void* const* planes, const int64_t* units, int npairs, float* partial_ws, size_t ws_bytes, float* sums, void* stream) {
  return eben_bl_fm_sums_codes(planes, units, nullptr, npairs, partial_ws, ws_bytes, sums, stream);
}

extern "C" int eben_bl_fm_sums_codes(const void* const* planes, const int64_t* units, void* const* codes, int npairs, float* partial_ws, size_t ws_bytes,
                                     float* sums, void* stream) {
  EBEN_REQUIRE(planes && units && npairs > 0 && partial_ws && sums, "bad bl_fm_sums arguments");
  if (ws_bytes < eben_bl_fm_sums_workspace(npairs)) return fail(EBEN_EWORKSPACE, "bl_fm_sums workspace too small");
  for (int p0 = 0; p0 < npairs; p0 += BLFM_MAX_PAIRS) {
    const int cnt = npairs - p0 < BLFM_MAX_PAIRS ? npairs - p0 : BLFM_MAX_PAIRS;
    BlFmTable T;
    for (int i = 0; i < cnt; ++i) {
      T.hi[i] = static_cast<const u32x4*>(planes[2 * (p0 + i)]);
      T.lo[i] = static_cast<const u32x4*>(planes[2 * (p0 + i) + 1]);
      T.units[i] = units[p0 + i];
      T.code[i] = codes ? static_cast<uint2*>(codes[p0 + i]) : nullptr;
      if (!T.hi[i] || T.units[i] <= 0) return fail(EBEN_EINVAL, "feature-matching pair %d is null or empty", p0 + i);
    }
    float* part = partial_ws + (size_t)p0 * BLFM_BLOCKS * 2;
    hipLaunchKernelGGL(bl_fm_partial_kernel, dim3(BLFM_BLOCKS, cnt), dim3(256), 0, as_stream(stream), T, part);
    EBEN_CHECK_LAUNCH("bl_fm_partial_kernel");
    hipLaunchKernelGGL(bl_fm_final_kernel, dim3(cnt), dim3(64), 0, as_stream(stream), part, sums + 2 * p0);
    EBEN_CHECK_LAUNCH("bl_fm_final_kernel");
  }
  return EBEN_OK;
}
