// ru_dw.hip -- both weight gradients of the generator's ResidualUnit in ONE launch (gfx950, bf16 matrix pipe).
//
//   dW_pw [m][c]    = sum_{b,t} g_z[b,m,t] * h[b,c,t],                 g_z = g_y * lrelu'(u, out_slope)
//   dW_dil[m][c][j] = sum_{b,t} g_h[b,m,t] * xin[b,c,reflect(t + (j-1) d)],   xin = lrelu(x, in_slope)
//                                                                       (eben_generator.py:287-316, the backward of both convs)
//
// The general weight-gradient kernels (conv_dw3.hip) order the reduction (time, 16 batch items), because along time the X operand
// of a strided conv is a strided gather -- which costs them a transposing pre-pass over every gradient tensor (dw3_pack_a_kernel).
// Both convs of the unit have stride 1: along TIME a lane's eight reduction elements are eight consecutive samples of one row, i.e.
// two 16-byte loads (4-byte aligned) straight from the (batch, channel, time) tensors, for the gradient operand and for the (shifted) input operand
// alike.  So: no pack pass, no LDS, no barrier -- global -> registers -> v_mfma_f32_32x32x16_bf16.
//   * block = 4 waves = one 32-row tile of both gradients x one K slab (SEG positions of one batch item); wave 0 accumulates the
//     pointwise gradient (A = g_z, B = h), waves 1-3 the three taps (A = g_h, B = xin shifted by (j-1) d, reflected at the ends);
//     each wave holds CT = C / 32 accumulators (its 32 rows x all C columns);
//   * split-K: one private slab per K slab in the layout eben_wn_bwd sums ([slab][row][row stride]; fixed order, no atomics);
//   * operands: NP = 1 single bf16 (EBEN_MATH_BF16, the bf16 generator backward), NP = 3 three pieces per operand and six
//     products (EBEN_MATH_BF16X6: fp32-grade, the fp32 backward).
#include "common.h"

namespace eben {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte load at 4-byte alignment (the shifted taps)

__device__ __forceinline__ unsigned rd_pack_bf16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

template <int NP>
__device__ __forceinline__ void rd_split8(const float (&v)[8], u32x4 (&p)[NP]) {
  float r[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = v[e];
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = rd_pack_bf16(r[2 * e], r[2 * e + 1]);
    p[q] = o;
    if (q + 1 < NP) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        r[2 * e] -= __builtin_bit_cast(float, o[e] << 16);
        r[2 * e + 1] -= __builtin_bit_cast(float, o[e] & 0xffff0000u);
      }
    }
  }
}

struct RuDwArgs {
  const float* gy; const float* u; const float* h; const float* gh; const float* x;
  float* slab_p; float* slab_d;   // [nslab][C][C] and [nslab][C][3 C]
  int B, L, d, nseg, seg;         // K slabs per batch item, positions per slab (a multiple of 16)
  float out_slope, in_slope;
};

// eight consecutive samples row[p0 .. p0 + 7] of a row of length L; positions outside [0, L) are reflected (refl) or read as zero;
// positions at or beyond `lim` (the end of the K slab) read as zero
template <bool REFL>
__device__ __forceinline__ void rd_load8(const float* __restrict__ row, int p0, int L, int lim, float (&v)[8]) {
  if (p0 >= 0 && p0 + 8 <= L && p0 + 8 <= lim) {
    const f32x4_u a = *reinterpret_cast<const f32x4_u*>(row + p0), b = *reinterpret_cast<const f32x4_u*>(row + p0 + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int q = p0 + e;
      bool ok = q < lim;
      if (REFL) {
        q = q < 0 ? -q : q;
        q = q >= L ? 2 * (L - 1) - q : q;
        ok = ok && q >= 0 && q < L;
      } else {
        ok = ok && q >= 0 && q < L;
      }
      v[e] = ok ? row[q] : 0.f;
    }
  }
}

template <int CT, int NP, int UN>
__global__ __launch_bounds__(256) void ru_dw_kernel(const RuDwArgs P) {
  constexpr int C = 32 * CT;
  // UN k-steps (16 UN positions) per iteration: every operand load of an iteration is in flight before its first MFMA, and the two
  // 64-byte halves of each row's cache line are asked for together
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);     // 0: pointwise, 1..3: tap wv - 1
  const int i = __builtin_amdgcn_readfirstlane(blockIdx.x % CT);   // row tile
  const int ks = __builtin_amdgcn_readfirstlane(blockIdx.x / CT);  // K slab: (batch item, segment)
  const int b = ks / P.nseg, sg = ks - b * P.nseg;
  const int L = P.L;
  const int t_lo = sg * P.seg;
  const int t_hi = t_lo + P.seg < L ? t_lo + P.seg : L;
  const long long base = (long long)b * C * L;
  const int m = 32 * i + (lane & 31), kh = lane >> 5;
  const int shift = wv == 0 ? 0 : (wv - 2) * P.d;              // tap j = wv - 1 reads xin(t + (j - 1) d)

  // A rows: the gradient operand of this wave's contraction; B rows: its input operand, one per column tile
  const float* arow = (wv == 0 ? P.gy : P.gh) + base + (long long)m * L;
  const float* urow = P.u + base + (long long)m * L;
  const float* brow = (wv == 0 ? P.h : P.x) + base + (long long)(lane & 31) * L;
  const bool act_b = wv != 0 && P.in_slope != 1.f;

  f32x16 acc[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

  float av[UN][8], uv[UN][8], bv[UN][CT][8];
  auto load = [&](int t) {
#pragma unroll
    for (int un = 0; un < UN; ++un) {
      const int p = t + 16 * un + 8 * kh;
      rd_load8<false>(arow, p, L, t_hi, av[un]);
      if (wv == 0) rd_load8<false>(urow, p, L, t_hi, uv[un]);
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        if (wv == 0) rd_load8<false>(brow + (long long)c * 32 * L, p, L, L, bv[un][c]);
        else rd_load8<true>(brow + (long long)c * 32 * L, p + shift, L, 0x7fffffff, bv[un][c]);
      }
    }
  };

  if (t_lo < t_hi) load(t_lo);
  for (int t = t_lo; t < t_hi; t += 16 * UN) {
    u32x4 ap[UN][NP], bp[UN][CT][NP];
#pragma unroll
    for (int un = 0; un < UN; ++un) {
      float a[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] = wv == 0 ? av[un][e] * dlrelu(uv[un][e], P.out_slope) : av[un][e];
      rd_split8<NP>(a, ap[un]);
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        float bb[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bb[e] = act_b ? lrelu(bv[un][c][e], P.in_slope) : bv[un][c][e];
        rd_split8<NP>(bb, bp[un][c]);
      }
    }
    if (t + 16 * UN < t_hi) load(t + 16 * UN);   // the next iteration's loads fly under this one's MFMAs
#pragma unroll
    for (int un = 0; un < UN; ++un)
#pragma unroll
      for (int lvl = NP - 1; lvl >= 0; --lvl)
#pragma unroll
        for (int q = 0; q <= lvl; ++q)
#pragma unroll
          for (int c = 0; c < CT; ++c)
            acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ap[un][q]), __builtin_bit_cast(bf16x8, bp[un][c][lvl - q]),
                                                             acc[c], 0, 0, 0);
  }

  // ---- this block's rows of slab ks: D tile column = lane & 31 (input channel), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) ----
  if (wv == 0) {
    float* sp = P.slab_p + (long long)ks * C * C;
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int mm = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        sp[(long long)mm * C + c * 32 + (lane & 31)] = acc[c][r];
      }
  } else {
    float* sd = P.slab_d + (long long)ks * C * 3 * C;
    const int j = wv - 1;
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int mm = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        sd[(long long)mm * 3 * C + (c * 32 + (lane & 31)) * 3 + j] = acc[c][r];
      }
  }
}

static int rdw_seg(int length) {
  // ~500 positions per K slab (a multiple of 16): 512 / 256 / 64 slabs of 16 / 64 / 256 KB at the generator's three widths
  static const int target = getenv("EBEN_RUDW_SEG") ? atoi(getenv("EBEN_RUDW_SEG")) : 512;
  const int nseg = ceil_div(length, target > 32 ? target : 32);
  return round_up(ceil_div(length, nseg), 16);
}

}  // namespace eben

using namespace eben;

extern "C" int eben_ru_dw_slabs(int batch, int channels, int length) {
  if (batch <= 0 || length <= 0 || (channels != 32 && channels != 64 && channels != 128)) return 0;
  return batch * ceil_div(length, rdw_seg(length));
}

extern "C" int eben_ru_dw(int math, int batch, int channels, int length, int dilation, const float* gy, const float* u, float out_slope,
                          const float* h, const float* gh, const float* x, float in_slope, float* slabs_pw, float* slabs_dil, void* stream) {
  EBEN_REQUIRE(channels == 32 || channels == 64 || channels == 128, "fused ResidualUnit weight gradient: 32, 64 or 128 channels (got %d)", channels);
  EBEN_REQUIRE(math == EBEN_MATH_BF16 || math == EBEN_MATH_BF16X6, "ru_dw: EBEN_MATH_BF16 or EBEN_MATH_BF16X6 (got %d)", math);
  EBEN_REQUIRE(batch > 0 && length > 0 && dilation >= 1 && dilation < length, "bad ResidualUnit geometry");
  EBEN_REQUIRE(gy && u && h && gh && x && slabs_pw && slabs_dil, "null pointer in ru_dw");
  RuDwArgs a;
  a.gy = gy; a.u = u; a.h = h; a.gh = gh; a.x = x; a.slab_p = slabs_pw; a.slab_d = slabs_dil;
  a.B = batch; a.L = length; a.d = dilation;
  a.seg = rdw_seg(length); a.nseg = ceil_div(length, a.seg);
  a.out_slope = out_slope; a.in_slope = in_slope;
  const int CT = channels / 32;
  const long long nb = (long long)batch * a.nseg * CT;
  if (nb > 0x7fffffffLL) return fail(EBEN_EINVAL, "ru_dw grid too large");
  hipStream_t st = as_stream(stream);
#define EBEN_RUDW(CTV, NPV, UNV) hipLaunchKernelGGL((ru_dw_kernel<CTV, NPV, UNV>), dim3((unsigned)nb), dim3(256), 0, st, a)
  static const int un = getenv("EBEN_RUDW_UNROLL") ? atoi(getenv("EBEN_RUDW_UNROLL")) : 2;
  if (math == EBEN_MATH_BF16 && un >= 2) {
    switch (CT) { case 1: EBEN_RUDW(1, 1, 2); break; case 2: EBEN_RUDW(2, 1, 2); break; default: EBEN_RUDW(4, 1, 2); }
  } else if (math == EBEN_MATH_BF16) {
    switch (CT) { case 1: EBEN_RUDW(1, 1, 1); break; case 2: EBEN_RUDW(2, 1, 1); break; default: EBEN_RUDW(4, 1, 1); }
  } else {
    switch (CT) { case 1: EBEN_RUDW(1, 3, 1); break; case 2: EBEN_RUDW(2, 3, 1); break; default: EBEN_RUDW(4, 3, 1); }
  }
#undef EBEN_RUDW
  EBEN_CHECK_LAUNCH("ru_dw_kernel");
  return EBEN_OK;
}
