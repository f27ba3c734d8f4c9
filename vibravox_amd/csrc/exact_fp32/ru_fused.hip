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

// ---------------------------------------------------------------------------------------------------------------
// Backward of the unit's input-gradient chain as ONE kernel:
//   g_z = g_y * lrelu'(u)                       (u = lrelu(z): its sign is the mask)
//   g_h = W_pw^T g_z                                                   [stage A, K = C]
//   g_x = ( g_y + fold( sum_j W_dil[j]^T g_h(t - (j-1) d) ) ) * lrelu'(x, in_slope) + post     [stage B, K = 3 C]
// A block owns BO = 128 - 2 d output positions of one item and computes g_h on the 128-position window [t0 - d, t0 - d + 128)
// around them (pointwise: the halo costs 2d / 128 of stage A's products, nothing is exchanged between blocks).  The masked
// gradient window is staged in LDS (C rows), stage A's result overwrites it IN PLACE -- a wave reads and writes only its own 32
// columns -- and stage B reads it back at the three tap shifts.  The reflect padding of the forward folds onto the first / last d
// positions as ONE extra tap each (W_dil[0]^T g_h(d - t) for 1 <= t <= d, W_dil[2]^T g_h(2 (L-1) - t - d) for L-1-d <= t <= L-2):
// extra chunks in the first / last tile only, so no padded workspace and no fold pass.  g_h is also written out (the dilated
// conv's weight gradient needs it).  Same weight-stream structure as the forward.
struct RuBwdArgs {
  const float* gy; const float* u; const float* wimg; const float* xmask; const float* post; float* gx; float* gh;
  int B, C, L, d, ntt, BO, GS, vec;
  float out_slope, in_slope;
};

template <int CT>
__global__ __launch_bounds__(256, CT == 4 ? 1 : (CT == 2 ? 2 : 4)) void ru_bwd_kernel(const RuBwdArgs P) {
  constexpr int NT = 256, C = 32 * CT;
  constexpr int WCH = 16 * 64 * CT;
  constexpr int PIECES = WCH / 4 / NT;
  typedef typename RuFrag<CT>::type afrag_t;

  extern __shared__ __attribute__((aligned(16))) float ru_smem[];
  float* Ws = ru_smem;             // 2 x WCH
  float* Gs = ru_smem + 2 * WCH;   // C rows of GS floats: masked gradient window, then g_h

  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tt = __builtin_amdgcn_readfirstlane(blockIdx.x % P.ntt);
  const int b = __builtin_amdgcn_readfirstlane(blockIdx.x / P.ntt);
  const int d = P.d, L = P.L, GS = P.GS, BO = P.BO;
  const int t0 = tt * BO;
  const int w0 = t0 - d;                       // first window position
  const int wa = w0 >= 0 ? (w0 & ~3) : w0;     // first staged position
  const int xshift = w0 - wa;
  const long long rowbase = (long long)b * C * L;

  // chunk sequence of this tile: stage A (CT), stage B (3 CT), left fold (CT, first tile), right fold (CT, tiles that reach L-1-d .. L-2)
  const bool fold_l = t0 <= d && L > 1;                                   // contains some t in [1, d]
  const bool fold_r = t0 + BO > L - 1 - d && t0 <= L - 2;                   // contains some t in [L-1-d, L-2]
  const int NS = 4 * CT + (fold_l ? CT : 0) + (fold_r ? CT : 0);
  auto img_chunk = [&](int sq) -> int {
    if (sq < 4 * CT) return sq;
    sq -= 4 * CT;
    if (fold_l) { if (sq < CT) return CT + 2 * CT + sq; sq -= CT; }       // tap j = 0 lives at shift index jj = 2
    return CT + sq;                                                        // tap j = 2 at jj = 0
  };
  auto issue_w = [&](int sq) {
    const float* src = P.wimg + (long long)img_chunk(sq) * WCH;
    float* dst = Ws + (sq & 1) * WCH;
#pragma unroll
    for (int p = 0; p < PIECES; ++p)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (p * NT + tid) * 4),
                                       (__attribute__((address_space(3))) void*)(dst + (p * NT + (tid & ~63)) * 4), 16, 0, 0);
  };
  issue_w(0);

  // ---- stage the masked gradient window (zero outside the signal) ----
  const float* gyr = P.gy + rowbase;
  const float* ur = P.u + rowbase;
  const bool interior = P.vec && w0 >= 0 && wa + GS <= L;
  if (interior) {
    const int x4 = GS >> 2, tot4 = C * x4;
    const unsigned x4_magic = (unsigned)((0x100000000ull + (unsigned)x4 - 1) / (unsigned)x4);
    for (int base = 0; base < tot4; base += 4 * NT) {
      f32x4 v[4], m[4];
      int sl[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = base + tid + e * NT;
        const int c = (int)__umulhi((unsigned)i, x4_magic);
        const int k4 = i - c * x4;
        const bool ok = i < tot4;
        const long long o = ok ? (long long)c * L + wa + 4 * k4 : 0;
        v[e] = *reinterpret_cast<const f32x4*>(gyr + o);
        m[e] = *reinterpret_cast<const f32x4*>(ur + o);
        sl[e] = ok ? c * GS + 4 * k4 : -1;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        f32x4 t;
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = v[e][k] * dlrelu(m[e][k], P.out_slope);
        if (sl[e] >= 0) *reinterpret_cast<f32x4*>(Gs + sl[e]) = t;
      }
    }
  } else {
    const int tot = C * GS;
    const unsigned gs_magic = (unsigned)((0x100000000ull + (unsigned)GS - 1) / (unsigned)GS);
    for (int base = 0; base < tot; base += 8 * NT) {
      float v[8], m[8];
      int sl[8], ok[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int i = base + tid + e * NT;
        const int c = (int)__umulhi((unsigned)i, gs_magic);
        const int p = i - c * GS;
        const int q = wa + p;
        ok[e] = (int)(i < tot) & (int)(q >= 0) & (int)(q < L);
        const long long o = ok[e] ? (long long)c * L + q : 0;
        v[e] = gyr[o];
        m[e] = ur[o];
        sl[e] = i < tot ? c * GS + p : -1;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (sl[e] >= 0) Gs[sl[e]] = ok[e] ? v[e] * dlrelu(m[e], P.out_slope) : 0.f;
    }
  }
  __syncthreads();

  f32x16 acc1[CT], acc2[CT];
#pragma unroll
  for (int i = 0; i < CT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc1[i][r] = 0.f; acc2[i][r] = 0.f; }

  const int wc = wn * 32 + (lane & 31);        // window / output column of this lane
  const float* gbA = Gs + (lane >> 5) * GS + xshift + wc;
  // stage B shifts a column by up to 2 d: lanes beyond the BO output columns (their results are dropped) stay inside the row
  const float* gbB = Gs + (lane >> 5) * GS + xshift + (wc < BO ? wc : BO - 1);

  // one chunk of 16 k-steps: rows 32 cb .. 32 cb + 31 of the LDS matrix at column offset `off` (relative to this lane's column)
  auto chunk = [&](int sq, int cb, const float* gb, int off, f32x16 (&acc)[CT], bool sel) {
    const float* wb = Ws + (sq & 1) * WCH + lane * CT;
    const float* xk = gb + cb * 32 * GS + off;
    float bv[16];
    afrag_t a[16];
    bv[0] = xk[0]; a[0] = *reinterpret_cast<const afrag_t*>(wb);
    bv[1] = xk[2 * GS]; a[1] = *reinterpret_cast<const afrag_t*>(wb + 64 * CT);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      if (ks + 2 < 16) {
        bv[ks + 2] = xk[(2 * (ks + 2)) * GS];
        a[ks + 2] = *reinterpret_cast<const afrag_t*>(wb + (ks + 2) * 64 * CT);
      }
      __builtin_amdgcn_sched_barrier(0);
      const float bb = sel ? bv[ks] : 0.f;
#pragma unroll
      for (int i = 0; i < CT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ru_elem<CT>(a[ks], i), bb, acc[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  int sq = 0;
  // ---- stage A: g_h = W_pw^T g_z on the window ----
  for (int cb = 0; cb < CT; ++cb, ++sq) {
    issue_w(sq + 1);
    chunk(sq, cb, gbA, 0, acc1, true);
    if (cb == CT - 1) {
      // in place: this wave's 32 columns, every row; and out to HBM for the columns this tile owns
      const int q = w0 + wc;
      const bool own = wc >= d && wc < d + BO && q < L;
#pragma unroll
      for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const float v = (q >= 0 && q < L) ? acc1[i][r] : 0.f;   // nothing of g_h exists beyond the signal
          Gs[m * GS + xshift + wc] = v;
          if (own) P.gh[rowbase + (long long)m * L + q] = v;
        }
    }
    __syncthreads();
  }
  // ---- stage B: the three taps; shift index jj <-> tap j = 2 - jj reads column wc + jj d ----
  for (int jj = 0; jj < 3; ++jj)
    for (int cb = 0; cb < CT; ++cb, ++sq) {
      if (sq + 1 < NS) issue_w(sq + 1);
      chunk(sq, cb, gbB, jj * d, acc2, true);
      __syncthreads();
    }
  const int t = t0 + wc;
  if (fold_l) {   // W_dil[0]^T g_h(d - t) for 1 <= t <= d: window column (d - t) - w0
    const bool in = t >= 1 && t <= d && wc < BO;
    const int off = in ? (d - t - w0) - wc : 0;
    for (int cb = 0; cb < CT; ++cb, ++sq) {
      if (sq + 1 < NS) issue_w(sq + 1);
      chunk(sq, cb, gbA, off, acc2, in);
      __syncthreads();
    }
  }
  if (fold_r) {   // W_dil[2]^T g_h(2 (L-1) - t - d) for L-1-d <= t <= L-2
    const bool in = t >= L - 1 - d && t <= L - 2 && wc < BO;
    const int off = in ? (2 * (L - 1) - t - d - w0) - wc : 0;
    for (int cb = 0; cb < CT; ++cb, ++sq) {
      if (sq + 1 < NS) issue_w(sq + 1);
      chunk(sq, cb, gbA, off, acc2, in);
      __syncthreads();
    }
  }

  // ---- epilogue ----
  if (wc >= BO || t >= L) return;
#pragma unroll
  for (int i = 0; i < CT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const long long idx = rowbase + (long long)m * L + t;
      float v = acc2[i][r] + P.gy[idx];
      if (P.xmask) v *= dlrelu(P.xmask[idx], P.in_slope);
      if (P.post) v += P.post[idx];
      P.gx[idx] = v;
    }
}

// weight image of the backward: chunks [0, CT): W_pw^T (rows = pointwise input channel, k = its output channel);
// chunk CT + jj CT + cb: W_dil[j = 2 - jj]^T (rows = dilated input channel, k = output channels 32 cb ..)
__global__ __launch_bounds__(256) void ru_pack_bwd_kernel(const float* __restrict__ vd, const float* __restrict__ sd, const float* __restrict__ vp,
                                                          const float* __restrict__ sp, float* __restrict__ img, int CT) {
  const int C = 32 * CT;
  const int total = 4 * CT * 16 * 64 * CT;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int i = e % CT;
    const int lane = (e / CT) & 63;
    const int ks = (e / (CT * 64)) & 15;
    const int ch = e / (CT * 64 * 16);
    const int row = 32 * i + (lane & 31);           // output row of the contraction = an INPUT channel of the conv
    float w;
    if (ch < CT) {
      const int m = ch * 32 + 2 * ks + (lane >> 5);  // reduction index = pointwise OUTPUT channel
      w = vp[(long long)m * C + row] * (sp ? sp[m] : 1.f);
    } else {
      const int jj = (ch - CT) / CT, cb = (ch - CT) - jj * CT;
      const int m = cb * 32 + 2 * ks + (lane >> 5);  // dilated conv OUTPUT channel
      w = vd[((long long)m * C + row) * 3 + (2 - jj)] * (sd ? sd[m] : 1.f);
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
  static LdsAttrOnce attr_once;
  auto kern = ru_fwd_kernel<CT>;
  const size_t lds = sizeof(float) * (2 * 16 * 64 * CT + (size_t)32 * CT * a.XS);
  {
    const hipError_t e = lds_attr_once(attr_once, reinterpret_cast<const void*>(kern));
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(ru_fwd)");
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

extern "C" int eben_ru_pack_bwd(int channels, const float* v_dil, const float* scale_dil, const float* v_pw, const float* scale_pw,
                                float* wimg, void* stream) {
  EBEN_REQUIRE(channels == 32 || channels == 64 || channels == 128, "fused ResidualUnit: 32, 64 or 128 channels (got %d)", channels);
  EBEN_REQUIRE(v_dil && v_pw && wimg, "null pointer in ru_pack_bwd");
  const int CT = channels / 32;
  const int total = 4 * CT * 16 * 64 * CT;
  hipLaunchKernelGGL(ru_pack_bwd_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), v_dil, scale_dil, v_pw, scale_pw, wimg, CT);
  EBEN_CHECK_LAUNCH("ru_pack_bwd_kernel");
  return EBEN_OK;
}

template <int CT>
static int launch_ru_bwd(const RuBwdArgs& a, hipStream_t st) {
  static LdsAttrOnce attr_once;
  auto kern = ru_bwd_kernel<CT>;
  const size_t lds = sizeof(float) * (2 * 16 * 64 * CT + (size_t)32 * CT * a.GS);
  {
    const hipError_t e = lds_attr_once(attr_once, reinterpret_cast<const void*>(kern));
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(ru_bwd)");
  }
  hipLaunchKernelGGL(kern, dim3(a.B * a.ntt), dim3(256), lds, st, a);
  EBEN_CHECK_LAUNCH("ru_bwd_kernel");
  return EBEN_OK;
}

extern "C" int eben_ru_bwd(int batch, int channels, int length, int dilation, const float* gy, const float* u, float out_slope,
                           const float* x, float in_slope, const float* post, const float* wimg_bwd, float* gx, float* gh, void* stream) {
  EBEN_REQUIRE(channels == 32 || channels == 64 || channels == 128, "fused ResidualUnit: 32, 64 or 128 channels (got %d)", channels);
  EBEN_REQUIRE(batch > 0 && length > 0 && dilation >= 1 && dilation <= 16 && dilation < length, "bad ResidualUnit geometry");
  EBEN_REQUIRE(gy && u && wimg_bwd && gx && gh, "null pointer in ru_bwd");
  EBEN_REQUIRE(in_slope == 1.f || x, "x is required to differentiate the fused input activation");
  RuBwdArgs a;
  a.gy = gy; a.u = u; a.wimg = wimg_bwd; a.xmask = in_slope != 1.f ? x : nullptr; a.post = post; a.gx = gx; a.gh = gh;
  a.B = batch; a.C = channels; a.L = length; a.d = dilation;
  a.BO = 128 - 2 * dilation; a.ntt = ceil_div(length, a.BO); a.GS = 132;
  a.out_slope = out_slope; a.in_slope = in_slope;
  a.vec = (((reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(u)) & 15) == 0 && (length & 3) == 0) ? 1 : 0;
  if ((long long)a.B * a.ntt > 0x7fffffffLL) return fail(EBEN_EINVAL, "ResidualUnit grid too large");
  switch (channels / 32) {
    case 1: return launch_ru_bwd<1>(a, as_stream(stream));
    case 2: return launch_ru_bwd<2>(a, as_stream(stream));
    default: return launch_ru_bwd<4>(a, as_stream(stream));
  }
}
