// ru_fused.hip -- the generator's ResidualUnit forward as ONE kernel (gfx950, exact fp32).
//
//   y = xin + lrelu( W_pw . ( W_dil (*) xin ), out_slope ),   xin = lrelu(x, in_slope)          (eben_generator.py:287-316)
//
// W_dil: k = 3, dilation d in {1, 3, 9}, "same" reflect padding; W_pw: k = 1; C in {32, 64, 128} channels in and out, no
// activation between the two convs.  Unfused this is three launches and seven tensor passes (dilated r/w, pointwise r/w,
// add 2r/1w) over (B, C, L) fp32 tensors that are far too large for any cache; fused, x is read once and y written once
// (training also writes h = W_dil (*) xin for the pointwise weight gradient and u = lrelu(z) whose sign is the activation mask
// of the backward).  Both contractions run on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate):
//   * block = 4 waves = one batch item x 128 positions x ALL C channels; wave w owns positions 32w .. 32w+31 and CT = C / 32
//     accumulator tiles per stage;
//   * stage 1 (K = 3 C): B fragments are ds_read_b32 of the staged x tile (C rows of 128 + 2 d + alignment floats; lanes 0-31 /
//     32-63 read two consecutive channels at consecutive positions: conflict free), one k-step = one tap x two channels;
//   * stage 2 (K = C) never leaves the registers: in the 32x32 accumulator layout lane l < 32 holds rows {0-3, 8-11, 16-19,
//     24-27} and lane l >= 32 rows {4-7, ...} of column l & 31 -- exactly the two k rows a 32x32x2 B fragment wants at one
//     column -- so accumulator register r of stage 1 IS the B operand of k-step r of stage 2 (k pair (rho, rho + 4)); the
//     pack kernel orders W_pw's k rows to match;
//   * the weights of both stages (weight-norm scale folded in) are ONE pre-packed LDS image [chunk][k-step][lane][CT], streamed
//     by global_load_lds_dwordx4 in 16-k-step chunks, double buffered, one barrier per chunk (as tapconv2.hip);
//   * the residual is read back from the staged tile, not from HBM.
#include "common.h"

namespace eben {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct RuArgs {
  const float* x; const float* wimg; float* y; float* h; float* u;
  int B, C, L, d, ntt, XS, vec;   // vec: x is 16-byte aligned and L % 4 == 0 (float4 staging of interior tiles)
  float in_slope, out_slope;
};

template <int CT> struct RuFrag;
template <> struct RuFrag<1> { typedef float type; };
template <> struct RuFrag<2> { typedef f32x2 type; };
template <> struct RuFrag<4> { typedef f32x4 type; };
template <int CT> __device__ __forceinline__ float ru_elem(const typename RuFrag<CT>::type& a, int i) { return a[i]; }
template <> __device__ __forceinline__ float ru_elem<1>(const float& a, int) { return a; }

template <int CT>
__global__ __launch_bounds__(256, CT == 4 ? 1 : (CT == 2 ? 2 : 4)) void ru_fwd_kernel(const RuArgs P) {
  constexpr int NT = 256, BN = 128, C = 32 * CT;
  constexpr int WCH = 16 * 64 * CT;           // floats per weight chunk (16 k-steps)
  constexpr int PIECES = WCH / 4 / NT;        // 16-byte LDS-DMA pieces per thread per chunk
  constexpr int NCH1 = 3 * CT, NCH = 4 * CT;  // chunks of stage 1 / of both stages
  typedef typename RuFrag<CT>::type afrag_t;

  extern __shared__ __attribute__((aligned(16))) float ru_smem[];
  float* Ws = ru_smem;             // 2 x WCH
  float* Xs = ru_smem + 2 * WCH;   // C rows of XS floats

  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tt = __builtin_amdgcn_readfirstlane(blockIdx.x % P.ntt);
  const int b = __builtin_amdgcn_readfirstlane(blockIdx.x / P.ntt);
  const int t0 = tt * BN, d = P.d, L = P.L, XS = P.XS;
  const int q0 = t0 - d;                      // first position the tile needs
  const int qa = q0 >= 0 ? (q0 & ~3) : q0;    // first position staged (16-byte aligned in the interior)
  const int xshift = q0 - qa;
  const float* xrow = P.x + (long long)b * C * L;

  auto issue_w = [&](int ch) {
    const float* src = P.wimg + (long long)ch * WCH;
    float* dst = Ws + (ch & 1) * WCH;
#pragma unroll
    for (int p = 0; p < PIECES; ++p)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (p * NT + tid) * 4),
                                       (__attribute__((address_space(3))) void*)(dst + (p * NT + (tid & ~63)) * 4), 16, 0, 0);
  };
  issue_w(0);

  // ---- stage the x tile: rows of XS floats holding positions qa .. qa + XS - 1 (reflected at the signal's ends) ----
  const bool interior = P.vec && q0 >= 0 && qa + XS <= L;
  if (interior) {
    const int x4 = XS >> 2, tot4 = C * x4;
    const unsigned x4_magic = (unsigned)((0x100000000ull + (unsigned)x4 - 1) / (unsigned)x4);
    for (int base = 0; base < tot4; base += 4 * NT) {
      f32x4 v[4];
      int sl[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = base + tid + e * NT;
        const int c = (int)__umulhi((unsigned)i, x4_magic);
        const int k4 = i - c * x4;
        const bool ok = i < tot4;
        v[e] = *reinterpret_cast<const f32x4*>(xrow + (ok ? (long long)c * L + qa + 4 * k4 : 0));
        sl[e] = ok ? c * XS + 4 * k4 : -1;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        f32x4 t;
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = lrelu(v[e][k], P.in_slope);
        if (sl[e] >= 0) *reinterpret_cast<f32x4*>(Xs + sl[e]) = t;
      }
    }
  } else {
    const int tot = C * XS;
    const unsigned xs_magic = (unsigned)((0x100000000ull + (unsigned)XS - 1) / (unsigned)XS);
    for (int base = 0; base < tot; base += 8 * NT) {
      float v[8];
      int sl[8], ok[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int i = base + tid + e * NT;
        const int c = (int)__umulhi((unsigned)i, xs_magic);
        const int p = i - c * XS;
        int q = qa + p;
        q = q < 0 ? -q : q;
        q = q >= L ? 2 * (L - 1) - q : q;
        ok[e] = (int)(i < tot) & (int)(q >= 0) & (int)(q < L);
        v[e] = xrow[ok[e] ? (long long)c * L + q : 0];
        sl[e] = i < tot ? c * XS + p : -1;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (sl[e] >= 0) Xs[sl[e]] = ok[e] ? lrelu(v[e], P.in_slope) : 0.f;
    }
  }
  __syncthreads();

  f32x16 acc1[CT], acc2[CT];
#pragma unroll
  for (int i = 0; i < CT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc1[i][r] = 0.f; acc2[i][r] = 0.f; }

  const int col = wn * 32 + (lane & 31);
  const float* xb = Xs + (lane >> 5) * XS + xshift + col;

  // ---- stage 1: h = W_dil (*) xin; chunk ch = tap ch / CT, input channels 32 (ch % CT) .. + 31 ----
  for (int ch = 0; ch < NCH1; ++ch) {
    issue_w(ch + 1);   // NCH1 < NCH: there is always a next chunk
    const int j = ch / CT, cb = ch - j * CT;
    const float* wb = Ws + (ch & 1) * WCH + lane * CT;
    const float* xk = xb + cb * 32 * XS + j * d;
    float bv[16];
    afrag_t a[16];
    bv[0] = xk[0]; a[0] = *reinterpret_cast<const afrag_t*>(wb);
    bv[1] = xk[2 * XS]; a[1] = *reinterpret_cast<const afrag_t*>(wb + 64 * CT);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      if (ks + 2 < 16) {
        bv[ks + 2] = xk[(2 * (ks + 2)) * XS];
        a[ks + 2] = *reinterpret_cast<const afrag_t*>(wb + (ks + 2) * 64 * CT);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < CT; ++i) acc1[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ru_elem<CT>(a[ks], i), bv[ks], acc1[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }

  const int t = t0 + col;
  const bool live = t < L;
  if (P.h != nullptr && live) {
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        P.h[((long long)b * C + m) * L + t] = acc1[i][r];
      }
  }

  // ---- stage 2: z = W_pw . h with the B operands taken from stage 1's accumulators ----
#pragma unroll
  for (int c2 = 0; c2 < CT; ++c2) {
    const int ch = NCH1 + c2;
    if (ch + 1 < NCH) issue_w(ch + 1);
    const float* wb = Ws + (ch & 1) * WCH + lane * CT;
    afrag_t a[16];
    a[0] = *reinterpret_cast<const afrag_t*>(wb);
    a[1] = *reinterpret_cast<const afrag_t*>(wb + 64 * CT);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      if (ks + 2 < 16) a[ks + 2] = *reinterpret_cast<const afrag_t*>(wb + (ks + 2) * 64 * CT);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < CT; ++i) acc2[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ru_elem<CT>(a[ks], i), acc1[c2][ks], acc2[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (c2 + 1 < CT) __syncthreads();
  }

  // ---- epilogue: y = xin + lrelu(z) (xin from the staged tile) ----
  if (!live) return;
  const float* xc = Xs + xshift + d + col;
#pragma unroll
  for (int i = 0; i < CT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const long long idx = ((long long)b * C + m) * L + t;
      const float uu = lrelu(acc2[i][r], P.out_slope);
      if (P.u != nullptr) P.u[idx] = uu;
      P.y[idx] = xc[m * XS] + uu;
    }
}

// ---- weight image: [chunk][k-step][lane][CT] ---------------------------------------------------------
__global__ __launch_bounds__(256) void ru_pack_kernel(const float* __restrict__ vd, const float* __restrict__ sd, const float* __restrict__ vp,
                                                      const float* __restrict__ sp, float* __restrict__ img, int CT) {
  const int C = 32 * CT;
  const int total = 4 * CT * 16 * 64 * CT;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int i = e % CT;
    const int lane = (e / CT) & 63;
    const int ks = (e / (CT * 64)) & 15;
    const int ch = e / (CT * 64 * 16);
    const int m = 32 * i + (lane & 31);
    float w;
    if (ch < 3 * CT) {
      const int j = ch / CT, cb = ch - j * CT;
      const int c = cb * 32 + 2 * ks + (lane >> 5);
      w = vd[((long long)m * C + c) * 3 + j] * (sd ? sd[m] : 1.f);
    } else {
      const int c2 = ch - 3 * CT;
      const int k = 32 * c2 + (ks & 3) + 8 * (ks >> 2) + 4 * (lane >> 5);
      w = vp[(long long)m * C + k] * (sp ? sp[m] : 1.f);
    }
    img[e] = w;
  }
}

static int ru_xs(int d) { return round_up(128 + 2 * d + 3, 4); }

}  // namespace eben

using namespace eben;

extern "C" size_t eben_ru_packed_floats(int channels) {
  if (channels != 32 && channels != 64 && channels != 128) return 0;
  return (size_t)4 * channels * channels;
}

extern "C" int eben_ru_pack(int channels, const float* v_dil, const float* scale_dil, const float* v_pw, const float* scale_pw, float* wimg,
                            void* stream) {
  EBEN_REQUIRE(channels == 32 || channels == 64 || channels == 128, "fused ResidualUnit: 32, 64 or 128 channels (got %d)", channels);
  EBEN_REQUIRE(v_dil && v_pw && wimg, "null pointer in ru_pack");
  const int CT = channels / 32;
  const int total = 4 * CT * 16 * 64 * CT;
  hipLaunchKernelGGL(ru_pack_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), v_dil, scale_dil, v_pw, scale_pw, wimg, CT);
  EBEN_CHECK_LAUNCH("ru_pack_kernel");
  return EBEN_OK;
}

template <int CT>
static int launch_ru(const RuArgs& a, hipStream_t st) {
  static bool attr_set = false;
  auto kern = ru_fwd_kernel<CT>;
  const size_t lds = sizeof(float) * (2 * 16 * 64 * CT + (size_t)32 * CT * a.XS);
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(ru_fwd)");
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.B * a.ntt), dim3(256), lds, st, a);
  EBEN_CHECK_LAUNCH("ru_fwd_kernel");
  return EBEN_OK;
}

extern "C" int eben_ru_fwd(int batch, int channels, int length, int dilation, const float* x, float in_slope, float out_slope,
                           const float* wimg, float* y, float* h, float* u, void* stream) {
  EBEN_REQUIRE(channels == 32 || channels == 64 || channels == 128, "fused ResidualUnit: 32, 64 or 128 channels (got %d)", channels);
  EBEN_REQUIRE(batch > 0 && length > 0 && dilation >= 1 && dilation <= 16 && dilation < length, "bad ResidualUnit geometry");
  EBEN_REQUIRE(x && wimg && y, "null pointer in ru_fwd");
  RuArgs a;
  a.x = x; a.wimg = wimg; a.y = y; a.h = h; a.u = u;
  a.B = batch; a.C = channels; a.L = length; a.d = dilation; a.ntt = ceil_div(length, 128); a.XS = ru_xs(dilation);
  a.in_slope = in_slope; a.out_slope = out_slope;
  a.vec = ((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (length & 3) == 0) ? 1 : 0;
  if ((long long)a.B * a.ntt > 0x7fffffffLL) return fail(EBEN_EINVAL, "ResidualUnit grid too large");
  switch (channels / 32) {
    case 1: return launch_ru<1>(a, as_stream(stream));
    case 2: return launch_ru<2>(a, as_stream(stream));
    default: return launch_ru<4>(a, as_stream(stream));
  }
}
