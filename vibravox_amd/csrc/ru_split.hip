// ru_split.hip -- the generator's fused ResidualUnit kernels on the bf16 matrix pipe with SPLIT operands (gfx950).
//
// ru_fused.hip runs both contractions of the unit on v_mfma_f32_32x32x2_f32: exact, but at 1/16 of the bf16 MFMA rate the
// 64- and 128-channel units are bound by the matrix pipe (28 us of MFMA time against 10-21 us of HBM time per launch).  Here every
// fp32 operand is split in registers into NP bf16 pieces
//     x = x0 + x1 (+ x2),   x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)        (every residual is exact in fp32)
// and a product is the sum of the piece products a_q * b_s with q + s < NP on v_mfma_f32_32x32x16_bf16 (fp32 accumulate; a
// bf16 x bf16 product is exact in fp32):
//     NP = 3  (EBEN_MATH_BF16X6): 6 MFMAs, dropped terms <= 2^-26 |a||b| -- below the rounding of an fp32 product; three pieces
//             hold all 24 mantissa bits, so this is fp32 arithmetic at 6/16 of the fp32 MFMA's cost (the forward's mode);
//     NP = 2  (EBEN_MATH_BF16X3): 3 MFMAs, error ~2^-17 per product;
//     NP = 1  (EBEN_MATH_BF16):   plain bf16 operands (the generator-backward mode next to a bf16 discriminator).
// Layout: the x tile stays fp32 in LDS exactly as ru_fused.hip stages it ([channel][position] rows, float4 loads, the residual
// and the reflect padding served from it); a B fragment is 8 ds_read_b32 of one lane's 8 channels at one position (lanes =
// consecutive positions: conflict free for any dilation), split in registers (cvt_pk + exact subtractions: VALU beside the
// MFMAs).  The weights are split once by the pack kernel into the LDS image [k-step][piece][row tile][lane] of 16-byte units and
// streamed with global_load_lds_dwordx4, double buffered.  The pointwise stage still never leaves the registers: in the 32x32
// accumulator layout lane l holds rows (r & 3) + 8 (r >> 2) + 4 (l >> 5), r = 0..15, of column l & 31 -- registers 8 s .. 8 s + 7
// are exactly one lane's 8 reduction elements of k-step s if W_pw's reduction index is permuted to match (pack kernel).
#include "common.h"

#include <cstdlib>

namespace eben {

#ifndef EBEN_RU_DBG
#define EBEN_RU_DBG 0   // scratch builds (results wrong by construction): 1 no y stores, 2 no saved-plane stores, 4 no x loads, 8 no MFMAs
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned rs_pack_bf16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));   // v_cvt_pk_bf16_f32 (RNE)
}

// 8 fp32 values -> NP units of 8 bf16: piece q holds bf16(v - p0 - .. - p(q-1))
template <int NP>
__device__ __forceinline__ void rs_split8(const float (&v)[8], u32x4 (&p)[NP]) {
  float r[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = v[e];
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = rs_pack_bf16(r[2 * e], r[2 * e + 1]);
    p[q] = o;
    if (q + 1 < NP) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        r[2 * e] -= __builtin_bit_cast(float, o[e] << 16);              // a bf16 is the upper half of its fp32
        r[2 * e + 1] -= __builtin_bit_cast(float, o[e] & 0xffff0000u);
      }
    }
  }
}

// acc[i] += sum over q + s < NP of A_q[i] B_s, smallest terms first
template <int NP, int CT>
__device__ __forceinline__ void rs_mma(const u32x4 (&a)[NP][CT], const u32x4 (&b)[NP], f32x16 (&acc)[CT]) {
#pragma unroll
  for (int lvl = NP - 1; lvl >= 0; --lvl)
#pragma unroll
    for (int q = 0; q <= lvl; ++q)
#pragma unroll
      for (int i = 0; i < CT; ++i) {
        if (EBEN_RU_DBG & 8) acc[i][0] += __builtin_bit_cast(float, a[q][i][0] ^ b[lvl - q][i & 3]);
        else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[q][i]), __builtin_bit_cast(bf16x8, b[lvl - q]), acc[i], 0, 0, 0);
      }
}

constexpr int RS_DMAX = 9;   // largest dilation the tile strides are laid out for (EBEN: 1, 3, 9)

struct Ru3Args {
  const float* x; const u32x4* wimg; float* y; float* h; float* u;
  // BL (what the bundle-layout backward reads, ru_bl.hip): xin = lrelu(x) and h as bf16 bundles [item][C / 8][L][8], and one byte per
  // (bundle, position) with bit e = (u[8 g + e] > 0)
  u32x4* xb; u32x4* hb; unsigned char* um;
  int B, L, d, ntt, vec;
  float in_slope, out_slope;
};

// ---------------------------------------------------------------------------------------------------------------------------
// forward: y = xin + lrelu(W_pw . (W_dil (*) xin)), xin = lrelu(x).  Block = NW waves = one item x 32 NW positions x all C = 32 CT
// channels; a wave owns 32 positions and every row (so that stage 2 finds its whole reduction in the wave's own accumulators).
// ---------------------------------------------------------------------------------------------------------------------------
template <int CT, int NW, int NP, int KSC, bool BL = false>
__global__ __launch_bounds__(NW * 64, (NW == 4 && CT < 4) ? 2 : 1) void ru3_fwd_kernel(const Ru3Args P) {
  constexpr int NT = NW * 64, BN = NW * 32, C = 32 * CT;
  constexpr int XS = BN + 2 * RS_DMAX + 6;          // floats per staged row: BN + 2 d + 3 (alignment shift) fits; multiple of 4
  constexpr int KB = C / 16;                        // k-steps per tap
  constexpr int KS1 = 3 * KB, KS2 = KB;
  constexpr int U = NP * CT * 64;                   // 16-byte units per k-step
  constexpr int WCHU = KSC * U;                     // ... per streamed chunk
  constexpr int NCH1 = KS1 / KSC, NCH2 = KS2 / KSC, NCHK = NCH1 + NCH2;
  static_assert(KS1 % KSC == 0 && KS2 % KSC == 0, "a chunk never straddles the two stages");
  static_assert(XS % 4 == 0 && WCHU % 64 == 0, "tile rows are float4-aligned, chunks are whole wave pieces");

  extern __shared__ __attribute__((aligned(16))) u32x4 rs_smem[];
  u32x4* Ws = rs_smem;                                        // 2 x WCHU
  float* Xs = reinterpret_cast<float*>(rs_smem + 2 * WCHU);   // C rows of XS floats

  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tt = __builtin_amdgcn_readfirstlane(blockIdx.x % P.ntt);
  const int b = __builtin_amdgcn_readfirstlane(blockIdx.x / P.ntt);
  const int t0 = tt * BN, d = P.d, L = P.L;
  const int q0 = t0 - d;
  const int qa = q0 >= 0 ? (q0 & ~3) : q0;
  const int xshift = q0 - qa;
  const float* xrow = P.x + (long long)b * C * L;

  auto issue_w = [&](int ch) {
    const u32x4* src = P.wimg + (long long)ch * WCHU;
    u32x4* dst = Ws + (ch & 1) * WCHU;
#pragma unroll
    for (int p = 0; p * NT < WCHU; ++p) {
      const int idx = p * NT + tid;
      if (idx < WCHU)   // wave-uniform: WCHU is a multiple of 64
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + idx),
                                         (__attribute__((address_space(3))) void*)(dst + (idx & ~63)), 16, 0, 0);
    }
  };
  issue_w(0);

  // ---- stage the x tile (as ru_fused.hip): rows of XS floats holding positions qa .. (reflected at the signal's ends) ----
  const bool interior = P.vec && q0 >= 0 && qa + XS <= L;
  if (interior) {
    constexpr int x4 = XS >> 2, tot4 = C * x4;
    for (int base = 0; base < ((EBEN_RU_DBG & 4) ? 4 * NT : tot4); base += 4 * NT) {
      f32x4 v[4];
      int sl[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = base + tid + e * NT;
        const int c = i / x4;
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
    constexpr int tot = C * XS;
    for (int base = 0; base < tot; base += 8 * NT) {
      float v[8];
      int sl[8], ok[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int i = base + tid + e * NT;
        const int c = i / XS;
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
  const float* xb = Xs + (lane >> 5) * 8 * XS + xshift + col;   // this lane's 8 channels of a 16-channel k-step start here
  const int t = t0 + col;
  const bool live = t < L;
  // bundle rows of this item: unit (bundle g, position t) at ubase + g L
  const long long ubase = (long long)b * (C / 8) * L + t;

  // ---- stage 1: h = W_dil (*) xin; k-step ks = tap (ks / KB) x channels 16 (ks % KB) .. + 15 ----
#pragma nounroll
  for (int ch = 0; ch < NCH1; ++ch) {
    issue_w(ch + 1);   // NCH1 < NCHK: there is always a next chunk
    const u32x4* wb = Ws + (ch & 1) * WCHU + lane;
#pragma unroll
    for (int kk = 0; kk < KSC; ++kk) {
      const int ks = ch * KSC + kk;
      const int j = ks / KB, cb = ks - j * KB;
      const float* xk = xb + cb * 16 * XS + j * d;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = xk[e * XS];
      u32x4 a[NP][CT], bq[NP];
#pragma unroll
      for (int q = 0; q < NP; ++q)
#pragma unroll
        for (int i = 0; i < CT; ++i) a[q][i] = wb[kk * U + (q * CT + i) * 64];
      rs_split8<NP>(v, bq);
      if constexpr (BL) {
        // the centre tap's fragment IS the saved input: piece 0 = bf16(xin) of channels 16 cb + 8 (lane >> 5) .. + 7 at this column
        if (j == 1 && live && (!(EBEN_RU_DBG & 2) || bq[0][0] == 0x12345u)) P.xb[ubase + (long long)(2 * cb + (lane >> 5)) * L] = bq[0];
      }
      rs_mma<NP, CT>(a, bq, acc1);
    }
    __syncthreads();
  }

  // stores: one 64-bit base per lane (item, column), 32-bit row offsets
  const long long lbase = (long long)b * C * L + t;
  if constexpr (BL) {
    // lane holds rows 8 q + 4 (lane >> 5) + e of tile i in registers 4 q + e: one 8-byte half of the unit (bundle 4 i + q, t)
    if (live) {
      uint2* __restrict__ hu = reinterpret_cast<uint2*>(P.hb + ubase) + (lane >> 5);
#pragma unroll
      for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          uint2 v;
          v.x = rs_pack_bf16(acc1[i][4 * q], acc1[i][4 * q + 1]);
          v.y = rs_pack_bf16(acc1[i][4 * q + 2], acc1[i][4 * q + 3]);
          if (!(EBEN_RU_DBG & 2) || v.x == 0x12345u) hu[(long long)(4 * i + q) * L * 2] = v;
        }
    }
  } else if (P.h != nullptr && live) {
    float* __restrict__ hb = P.h + lbase;
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned m = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        hb[m * (unsigned)L] = acc1[i][r];
      }
  }

  // ---- stage 2: z = W_pw . h, the B operands split out of stage 1's accumulators ----
#pragma unroll
  for (int c2 = 0; c2 < NCH2; ++c2) {
    const int ch = NCH1 + c2;
    if (ch + 1 < NCHK) issue_w(ch + 1);
    const u32x4* wb = Ws + (ch & 1) * WCHU + lane;
#pragma unroll
    for (int kk = 0; kk < KSC; ++kk) {
      const int ks2 = c2 * KSC + kk;
      const int isrc = ks2 >> 1, s = ks2 & 1;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = acc1[isrc][8 * s + e];
      u32x4 a[NP][CT], bq[NP];
#pragma unroll
      for (int q = 0; q < NP; ++q)
#pragma unroll
        for (int i = 0; i < CT; ++i) a[q][i] = wb[kk * U + (q * CT + i) * 64];
      rs_split8<NP>(v, bq);
      rs_mma<NP, CT>(a, bq, acc2);
    }
    if (c2 + 1 < NCH2) __syncthreads();
  }

  if constexpr (BL) {
    // sign bits of z (= of u): nibble (i, q) of this lane = its four rows of bundle 4 i + q; the other four sit in lane ^ 32
    unsigned nlo = 0, nhi = 0;
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        unsigned nib = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) nib |= (acc2[i][4 * q + e] > 0.f ? 1u : 0u) << e;
        if (i < 2) nlo |= nib << (4 * (4 * i + q));
        else nhi |= nib << (4 * (4 * (i - 2) + q));
      }
    const unsigned olo = __shfl_xor(nlo, 32, 64), ohi = CT > 2 ? __shfl_xor(nhi, 32, 64) : 0u;
    const int hf = lane >> 5;
    const unsigned l0 = hf ? olo : nlo, l1 = hf ? nlo : olo, h0 = hf ? ohi : nhi, h1 = hf ? nhi : ohi;   // rows 0-3 / 4-7 of every bundle
    if (live) {
      unsigned char* __restrict__ ub8 = P.um + ubase;
#pragma unroll
      for (int g = 0; g < 4 * CT; g += 2) {   // each half-wave stores every other bundle row
        const int gg = g + hf;
        const unsigned w0 = gg < 8 ? l0 : h0, w1 = gg < 8 ? l1 : h1;
        const int sh = 4 * (gg & 7);
        ub8[(long long)gg * L] = (unsigned char)(((w0 >> sh) & 15u) | (((w1 >> sh) & 15u) << 4));
      }
    }
  }

  // ---- epilogue: y = xin + lrelu(z) (xin from the staged tile) ----
  if (!live) return;
  const float* xc = Xs + xshift + d + col;
  float* __restrict__ yb = P.y + lbase;
  float* __restrict__ ub = P.u + lbase;
  const bool keep_u = !BL && P.u != nullptr;
#pragma unroll
  for (int i = 0; i < CT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const unsigned m = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const unsigned off = m * (unsigned)L;
      const float uu = lrelu(acc2[i][r], P.out_slope);
      if (keep_u) ub[off] = uu;
      if (!(EBEN_RU_DBG & 1) || uu == 12345.f) yb[off] = xc[m * XS] + uu;
    }
}

// forward weight image: unit ((ks NP + q) CT + i) 64 + lane; ks < 3 KB: tap ks / KB, channels 16 (ks % KB) + 8 (lane >> 5) + e;
// then the pointwise k-steps in the accumulator order (see the header)
__device__ __forceinline__ void ru3_pack_fwd_body(const float* __restrict__ vd, const float* __restrict__ sd, const float* __restrict__ vp,
                                                  const float* __restrict__ sp, u32x4* __restrict__ img, int CT, int NP, unsigned bid, unsigned nblk) {
  const int C = 32 * CT, KB = C / 16;
  const int total = 4 * KB * CT * 64;   // (k-step, row tile, lane)
  for (int t = bid * 256 + threadIdx.x; t < total; t += nblk * 256) {
    const int lane = t & 63;
    const int i = (t >> 6) % CT;
    const int ks = t / (64 * CT);
    const int m = 32 * i + (lane & 31), kh = lane >> 5;
    float w[8];
    if (ks < 3 * KB) {
      const int j = ks / KB, cb = ks - j * KB;
#pragma unroll
      for (int e = 0; e < 8; ++e) w[e] = vd[((long long)m * C + cb * 16 + 8 * kh + e) * 3 + j] * (sd ? sd[m] : 1.f);
    } else {
      const int ks2 = ks - 3 * KB, isrc = ks2 >> 1, s = ks2 & 1;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int chn = 32 * isrc + (e & 3) + 8 * (2 * s + (e >> 2)) + 4 * kh;
        w[e] = vp[(long long)m * C + chn] * (sp ? sp[m] : 1.f);
      }
    }
    float r[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = w[e];
    for (int q = 0; q < NP; ++q) {
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = rs_pack_bf16(r[2 * e], r[2 * e + 1]);
      img[((long long)(ks * NP + q) * CT + i) * 64 + lane] = o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        r[2 * e] -= __builtin_bit_cast(float, o[e] << 16);
        r[2 * e + 1] -= __builtin_bit_cast(float, o[e] & 0xffff0000u);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// backward (the structure of ru_fused.hip's ru_bwd_kernel): a block owns BO = W - 2 d output positions of one item (W = 32 NW
// window columns), stages the masked gradient window g_y * lrelu'(u) as fp32 rows, computes g_h = W_pw^T g_z on the window
// (stage A), writes it back IN PLACE (and to HBM for the columns it owns), then g_x = (g_y + fold(sum_j W_dil[j]^T g_h(. + shift)))
// * lrelu'(x) + post (stage B + the two reflect folds as extra taps in the first / last tiles).
// Weight entries: one entry = 32 reduction channels = 2 k-steps; the sequence is stage A (CT entries), three taps (3 CT), left
// fold (CT), right fold (CT); G entries are streamed per barrier (G divides CT, so a group never straddles a stage).
// ---------------------------------------------------------------------------------------------------------------------------
struct Ru3BwdArgs {
  const float* gy; const float* u; const u32x4* wimg; const float* xmask; const float* post; float* gx; float* gh;
  int B, L, d, ntt, BO, vec;
  float out_slope, in_slope;
};

template <int CT, int NW, int NP, int G>
__global__ __launch_bounds__(NW * 64, (NW == 4 && CT < 4) ? 2 : 1) void ru3_bwd_kernel(const Ru3BwdArgs P) {
  constexpr int NT = NW * 64, WN = NW * 32, C = 32 * CT;
  constexpr int GS = WN + 4;
  constexpr int U = NP * CT * 64;      // units per k-step
  constexpr int EU = 2 * U;            // units per entry (32 channels)
  constexpr int WCHU = G * EU;
  static_assert(CT % G == 0, "an entry group never straddles a stage");

  extern __shared__ __attribute__((aligned(16))) u32x4 rs_smem[];
  u32x4* Ws = rs_smem;                                        // 2 x WCHU
  float* Gs = reinterpret_cast<float*>(rs_smem + 2 * WCHU);   // C rows of GS floats: masked gradient window, then g_h

  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tt = __builtin_amdgcn_readfirstlane(blockIdx.x % P.ntt);
  const int b = __builtin_amdgcn_readfirstlane(blockIdx.x / P.ntt);
  const int d = P.d, L = P.L, BO = P.BO;
  const int t0 = tt * BO;
  const int w0 = t0 - d;
  const int wa = w0 >= 0 ? (w0 & ~3) : w0;
  const int xshift = w0 - wa;
  const long long rowbase = (long long)b * C * L;

  const bool fold_l = t0 <= d && L > 1;
  const bool fold_r = t0 + BO > L - 1 - d && t0 <= L - 2;
  const int NS = 4 * CT + (fold_l ? CT : 0) + (fold_r ? CT : 0);   // entries of this tile
  const int NSG = NS / G;                                             // ... in groups of G
  auto img_entry = [&](int sq) -> int {
    if (sq < 4 * CT) return sq;
    sq -= 4 * CT;
    if (fold_l) { if (sq < CT) return CT + 2 * CT + sq; sq -= CT; }   // tap j = 0 lives at shift index jj = 2
    return CT + sq;                                                    // tap j = 2 at jj = 0
  };
  auto issue_w = [&](int sg) {
    u32x4* dst = Ws + (sg & 1) * WCHU;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const u32x4* src = P.wimg + (long long)img_entry(sg * G + g) * EU;
#pragma unroll
      for (int p = 0; p * NT < EU; ++p) {
        const int idx = p * NT + tid;
        if (idx < EU)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + idx),
                                           (__attribute__((address_space(3))) void*)(dst + g * EU + (idx & ~63)), 16, 0, 0);
      }
    }
  };
  issue_w(0);

  // ---- stage the masked gradient window (zero outside the signal) ----
  const float* gyr = P.gy + rowbase;
  const float* ur = P.u + rowbase;
  const bool interior = P.vec && w0 >= 0 && wa + GS <= L;
  if (interior) {
    constexpr int x4 = GS >> 2, tot4 = C * x4;
    for (int base = 0; base < tot4; base += 4 * NT) {
      f32x4 v[4], m[4];
      int sl[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = base + tid + e * NT;
        const int c = i / x4;
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
    constexpr int tot = C * GS;
    for (int base = 0; base < tot; base += 8 * NT) {
      float v[8], m[8];
      int sl[8], ok[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int i = base + tid + e * NT;
        const int c = i / GS;
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
  const float* gbA = Gs + (lane >> 5) * 8 * GS + xshift + wc;
  // stage B shifts a column by up to 2 d: lanes beyond the BO output columns (their results are dropped) stay inside the row
  const float* gbB = Gs + (lane >> 5) * 8 * GS + xshift + (wc < BO ? wc : BO - 1);

  // one entry (32 reduction channels 32 cb .. + 31 = two k-steps) from slot g of the current weight buffer
  auto entry = [&](const u32x4* wslot, int cb, const float* gb, int off, f32x16 (&acc)[CT], bool sel) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const float* xk = gb + (cb * 32 + kk * 16) * GS + off;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = sel ? xk[e * GS] : 0.f;
      u32x4 a[NP][CT], bq[NP];
#pragma unroll
      for (int q = 0; q < NP; ++q)
#pragma unroll
        for (int i = 0; i < CT; ++i) a[q][i] = wslot[kk * U + (q * CT + i) * 64];
      rs_split8<NP>(v, bq);
      rs_mma<NP, CT>(a, bq, acc);
    }
  };

  int sg = 0;
  // ---- stage A: g_h = W_pw^T g_z on the window ----
#pragma nounroll
  for (int c0 = 0; c0 < CT; c0 += G, ++sg) {
    issue_w(sg + 1);   // stage B always follows
    const u32x4* wb = Ws + (sg & 1) * WCHU + lane;
#pragma unroll
    for (int g = 0; g < G; ++g) entry(wb + g * EU, c0 + g, gbA, 0, acc1, true);
    if (c0 + G >= CT) {
      // in place: a wave reads and writes only its own 32 columns (every row), so no barrier separates the two
      const int q = w0 + wc;
      const bool own = wc >= d && wc < d + BO && q < L;
      float* __restrict__ ghb = P.gh + rowbase + (own ? q : 0);
#pragma unroll
      for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const float v = (q >= 0 && q < L) ? acc1[i][r] : 0.f;   // nothing of g_h exists beyond the signal
          Gs[m * GS + xshift + wc] = v;
          if (own) ghb[(unsigned)m * (unsigned)L] = v;
        }
    }
    __syncthreads();
  }
  // ---- stage B: the three taps; shift index jj <-> tap j = 2 - jj reads column wc + jj d ----
#pragma nounroll
  for (int jj = 0; jj < 3; ++jj)
#pragma nounroll
    for (int c0 = 0; c0 < CT; c0 += G, ++sg) {
      if (sg + 1 < NSG) issue_w(sg + 1);
      const u32x4* wb = Ws + (sg & 1) * WCHU + lane;
#pragma unroll
      for (int g = 0; g < G; ++g) entry(wb + g * EU, c0 + g, gbB, jj * d, acc2, true);
      __syncthreads();
    }
  const int t = t0 + wc;
  if (fold_l) {   // W_dil[0]^T g_h(d - t) for 1 <= t <= d: window column (d - t) - w0
    const bool in = t >= 1 && t <= d && wc < BO;
    const int off = in ? (d - t - w0) - wc : 0;
#pragma nounroll
    for (int c0 = 0; c0 < CT; c0 += G, ++sg) {
      if (sg + 1 < NSG) issue_w(sg + 1);
      const u32x4* wb = Ws + (sg & 1) * WCHU + lane;
#pragma unroll
      for (int g = 0; g < G; ++g) entry(wb + g * EU, c0 + g, gbA, off, acc2, in);
      __syncthreads();
    }
  }
  if (fold_r) {   // W_dil[2]^T g_h(2 (L-1) - t - d) for L-1-d <= t <= L-2
    const bool in = t >= L - 1 - d && t <= L - 2 && wc < BO;
    const int off = in ? (2 * (L - 1) - t - d - w0) - wc : 0;
#pragma nounroll
    for (int c0 = 0; c0 < CT; c0 += G, ++sg) {
      if (sg + 1 < NSG) issue_w(sg + 1);
      const u32x4* wb = Ws + (sg & 1) * WCHU + lane;
#pragma unroll
      for (int g = 0; g < G; ++g) entry(wb + g * EU, c0 + g, gbA, off, acc2, in);
      __syncthreads();
    }
  }

  // ---- epilogue: the three operands of the join are read for all 16 rows of a tile in one batch each (value by value the
  // compiler waits for every load in turn) ----
  if (wc >= BO || t >= L) return;
  const float* __restrict__ gyb = P.gy + rowbase + t;
  const float* __restrict__ xmb = P.xmask + rowbase + t;
  const float* __restrict__ pob = P.post + rowbase + t;
  float* __restrict__ gxb = P.gx + rowbase + t;
#pragma unroll
  for (int i = 0; i < CT; ++i) {
    float gv[16], xv[16], pv[16];
    unsigned off[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      off[r] = (unsigned)(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * (unsigned)L;
      gv[r] = gyb[off[r]];
    }
    if (P.xmask) {
#pragma unroll
      for (int r = 0; r < 16; ++r) xv[r] = xmb[off[r]];
    }
    if (P.post) {
#pragma unroll
      for (int r = 0; r < 16; ++r) pv[r] = pob[off[r]];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = acc2[i][r] + gv[r];
      if (P.xmask) v *= dlrelu(xv[r], P.in_slope);
      if (P.post) v += pv[r];
      gxb[off[r]] = v;
    }
  }
}

// backward weight image: entry en = 32 reduction channels: unit (((en 2 + kk) NP + q) CT + i) 64 + lane.
// en < CT: W_pw^T (row = pointwise input channel, reduction = its output channel); en = CT + jj CT + cb: W_dil[2 - jj]^T
__device__ __forceinline__ void ru3_pack_bwd_body(const float* __restrict__ vd, const float* __restrict__ sd, const float* __restrict__ vp,
                                                  const float* __restrict__ sp, u32x4* __restrict__ img, int CT, int NP, unsigned bid, unsigned nblk) {
  const int C = 32 * CT;
  const int total = 4 * CT * 2 * CT * 64;   // (entry, k-step of the entry, row tile, lane)
  for (int t = bid * 256 + threadIdx.x; t < total; t += nblk * 256) {
    const int lane = t & 63;
    const int i = (t >> 6) % CT;
    const int kk = (t / (64 * CT)) & 1;
    const int en = t / (128 * CT);
    const int row = 32 * i + (lane & 31), kh = lane >> 5;
    float w[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (en < CT) {
        const int m = en * 32 + kk * 16 + 8 * kh + e;
        w[e] = vp[(long long)m * C + row] * (sp ? sp[m] : 1.f);
      } else {
        const int jj = (en - CT) / CT, cb = (en - CT) - jj * CT;
        const int m = cb * 32 + kk * 16 + 8 * kh + e;
        w[e] = vd[((long long)m * C + row) * 3 + (2 - jj)] * (sd ? sd[m] : 1.f);
      }
    }
    float r[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = w[e];
    for (int q = 0; q < NP; ++q) {
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = rs_pack_bf16(r[2 * e], r[2 * e + 1]);
      img[(((long long)(en * 2 + kk) * NP + q) * CT + i) * 64 + lane] = o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        r[2 * e] -= __builtin_bit_cast(float, o[e] << 16);
        r[2 * e + 1] -= __builtin_bit_cast(float, o[e] & 0xffff0000u);
      }
    }
  }
}

__global__ __launch_bounds__(256) void ru3_pack_fwd_kernel(const float* vd, const float* sd, const float* vp, const float* sp, u32x4* img, int CT, int NP) {
  ru3_pack_fwd_body(vd, sd, vp, sp, img, CT, NP, blockIdx.x, gridDim.x);
}
__global__ __launch_bounds__(256) void ru3_pack_bwd_kernel(const float* vd, const float* sd, const float* vp, const float* sp, u32x4* img, int CT, int NP) {
  ru3_pack_bwd_body(vd, sd, vp, sp, img, CT, NP, blockIdx.x, gridDim.x);
}

// the images of many units in one launch (eben_ru_pack_multi): block -> job by the prefix sums of the jobs' block counts
constexpr int RU_PACK_MULTI = 48;
struct RuPackTable {
  int n;
  unsigned short first[RU_PACK_MULTI + 1];
  struct { const float* vd; const float* sd; const float* vp; const float* sp; u32x4* img; short CT, NP, which, pad; } job[RU_PACK_MULTI];
};
__global__ __launch_bounds__(256) void ru3_pack_multi_kernel(const RuPackTable T) {
  int j = 0;
#pragma unroll 1
  while (j + 1 < T.n && blockIdx.x >= T.first[j + 1]) ++j;
  const unsigned bid = blockIdx.x - T.first[j], nblk = T.first[j + 1] - T.first[j];
  if (T.job[j].which == 0) ru3_pack_fwd_body(T.job[j].vd, T.job[j].sd, T.job[j].vp, T.job[j].sp, T.job[j].img, T.job[j].CT, T.job[j].NP, bid, nblk);
  else ru3_pack_bwd_body(T.job[j].vd, T.job[j].sd, T.job[j].vp, T.job[j].sp, T.job[j].img, T.job[j].CT, T.job[j].NP, bid, nblk);
}

// ---- host side ----------------------------------------------------------------------------------------------------------
static int rs_pieces(int math) {
  switch (math) {
    case EBEN_MATH_BF16: return 1;
    case EBEN_MATH_BF16X3: return 2;
    case EBEN_MATH_BF16X6: return 3;
    default: return 0;
  }
}

template <int CT, int NW, int NP, int KSC, bool BL = false>
static int launch_ru3_fwd(const Ru3Args& a, hipStream_t st) {
  static LdsAttrOnce attr_once;
  auto kern = ru3_fwd_kernel<CT, NW, NP, KSC, BL>;
  constexpr int XS = NW * 32 + 2 * RS_DMAX + 6;
  const size_t lds = (size_t)2 * KSC * NP * CT * 64 * 16 + sizeof(float) * 32 * CT * XS;
  {
    const hipError_t e = lds_attr_once(attr_once, reinterpret_cast<const void*>(kern));
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(ru3_fwd)");
  }
  hipLaunchKernelGGL(kern, dim3(a.B * a.ntt), dim3(NW * 64), lds, st, a);
  EBEN_CHECK_LAUNCH("ru3_fwd_kernel");
  return EBEN_OK;
}

template <int CT, int NW, int NP, int G>
static int launch_ru3_bwd(const Ru3BwdArgs& a, hipStream_t st) {
  static LdsAttrOnce attr_once;
  auto kern = ru3_bwd_kernel<CT, NW, NP, G>;
  constexpr int GS = NW * 32 + 4;
  const size_t lds = (size_t)2 * G * 2 * NP * CT * 64 * 16 + sizeof(float) * 32 * CT * GS;
  {
    const hipError_t e = lds_attr_once(attr_once, reinterpret_cast<const void*>(kern));
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(ru3_bwd)");
  }
  hipLaunchKernelGGL(kern, dim3(a.B * a.ntt), dim3(NW * 64), lds, st, a);
  EBEN_CHECK_LAUNCH("ru3_bwd_kernel");
  return EBEN_OK;
}

// window / tile width in waves per channel count (32 positions per wave)
static int rs_waves(int channels) {
  static const int w128 = getenv("EBEN_RU3_W128") ? atoi(getenv("EBEN_RU3_W128")) : 4;
  return channels == 128 ? (w128 == 4 ? 4 : 2) : 4;
}

int ru3_supported(int channels, int dilation, int math) {
  return (channels == 32 || channels == 64 || channels == 128) && dilation >= 1 && dilation <= RS_DMAX && rs_pieces(math) > 0;
}

size_t ru3_packed_floats(int channels, int math) {
  const int np = rs_pieces(math);
  if (np == 0 || (channels != 32 && channels != 64 && channels != 128)) return 0;
  return (size_t)2 * np * channels * channels;   // 4 C^2 weights x np pieces x 2 bytes
}

int ru3_pack(int channels, int math, int which, const float* vd, const float* sd, const float* vp, const float* sp, float* wimg, hipStream_t st) {
  const int np = rs_pieces(math), CT = channels / 32;
  const int total = which == 0 ? 8 * CT * CT * 64 : 4 * CT * 2 * CT * 64;
  if (which == 0)
    hipLaunchKernelGGL(ru3_pack_fwd_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st, vd, sd, vp, sp, reinterpret_cast<u32x4*>(wimg), CT, np);
  else
    hipLaunchKernelGGL(ru3_pack_bwd_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st, vd, sd, vp, sp, reinterpret_cast<u32x4*>(wimg), CT, np);
  EBEN_CHECK_LAUNCH("ru3_pack_kernel");
  return EBEN_OK;
}

int ru3_fwd(int math, int batch, int channels, int length, int dilation, const float* x, float in_slope, float out_slope, const float* wimg,
            float* y, float* h, float* u, hipStream_t st) {
  Ru3Args a;
  a.x = x; a.wimg = reinterpret_cast<const u32x4*>(wimg); a.y = y; a.h = h; a.u = u;
  a.xb = nullptr; a.hb = nullptr; a.um = nullptr;
  a.B = batch; a.L = length; a.d = dilation;
  const int bn = rs_waves(channels) * 32;
  a.ntt = ceil_div(length, bn);
  a.in_slope = in_slope; a.out_slope = out_slope;
  a.vec = ((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (length & 3) == 0) ? 1 : 0;
  if ((long long)a.B * a.ntt > 0x7fffffffLL) return fail(EBEN_EINVAL, "ResidualUnit grid too large");
  const int np = rs_pieces(math);
#define EBEN_RU3_FWD(NPV)                                                 \
  switch (channels / 32) {                                                \
    case 1: return launch_ru3_fwd<1, 4, NPV, 2>(a, st);                   \
    case 2: return launch_ru3_fwd<2, 4, NPV, 2>(a, st);                   \
    default: return rs_waves(128) == 4 ? launch_ru3_fwd<4, 4, NPV, 2>(a, st) : launch_ru3_fwd<4, 2, NPV, 1>(a, st); \
  }
  switch (np) {
    case 1: EBEN_RU3_FWD(1)
    case 2: EBEN_RU3_FWD(2)
    default: EBEN_RU3_FWD(3)
  }
#undef EBEN_RU3_FWD
}

// forward that saves for the bundle-layout backward (ru_bl.hip): y fp32, xin / h as bf16 bundles, the sign bits of u
int ru3_fwd_bl(int math, int batch, int channels, int length, int dilation, const float* x, float in_slope, float out_slope, const float* wimg,
               float* y, void* xb, void* hb, void* um, hipStream_t st) {
  Ru3Args a;
  a.x = x; a.wimg = reinterpret_cast<const u32x4*>(wimg); a.y = y; a.h = nullptr; a.u = nullptr;
  a.xb = static_cast<u32x4*>(xb); a.hb = static_cast<u32x4*>(hb); a.um = static_cast<unsigned char*>(um);
  a.B = batch; a.L = length; a.d = dilation;
  a.ntt = ceil_div(length, 128);
  a.in_slope = in_slope; a.out_slope = out_slope;
  a.vec = ((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (length & 3) == 0) ? 1 : 0;
  if ((long long)a.B * a.ntt > 0x7fffffffLL) return fail(EBEN_EINVAL, "ResidualUnit grid too large");
  // EBEN_MATH_BF16X6 (fp32-grade: six piece products) or EBEN_MATH_BF16X3 (hi + lo operands, three products: the bf16-mixed plan --
  // 2^-17 per product, the generator's output stays orders of magnitude inside north_star's 1e-5 MSE; [MI355X] 42 -> 35 us at 64 channels)
  const int np = rs_pieces(math);
  if (np != 3 && np != 2) return fail(EBEN_EUNSUPPORTED, "eben_rubl_fwd: the forward computes in EBEN_MATH_BF16X6 or EBEN_MATH_BF16X3 (got %d)", math);
  if (np == 2) {
    switch (channels / 32) {
      case 1: return launch_ru3_fwd<1, 4, 2, 2, true>(a, st);
      case 2: return launch_ru3_fwd<2, 4, 2, 2, true>(a, st);
      default: return launch_ru3_fwd<4, 4, 2, 2, true>(a, st);
    }
  }
  switch (channels / 32) {
    case 1: return launch_ru3_fwd<1, 4, 3, 2, true>(a, st);
    case 2: return launch_ru3_fwd<2, 4, 3, 2, true>(a, st);
    default: return launch_ru3_fwd<4, 4, 3, 2, true>(a, st);
  }
}

int ru3_bwd(int math, int batch, int channels, int length, int dilation, const float* gy, const float* u, float out_slope, const float* x,
            float in_slope, const float* post, const float* wimg, float* gx, float* gh, hipStream_t st) {
  Ru3BwdArgs a;
  a.gy = gy; a.u = u; a.wimg = reinterpret_cast<const u32x4*>(wimg); a.xmask = in_slope != 1.f ? x : nullptr; a.post = post; a.gx = gx; a.gh = gh;
  a.B = batch; a.L = length; a.d = dilation;
  // (A five-wave, 160-column window for the 128-channel units -- 288 / 320 blocks at d = 3 / 9 are two rounds of one block per CU, 224 / 256
  // one -- measured 50 -> 41 us per launch in round 3 but needs 2 waves on one SIMD at 256 registers each: 536 bytes of scratch.  Gone;
  // the bundle-layout backward (ru_bl.hip) halves the LDS per block instead.)
  const int nw = rs_waves(channels);
  const int wn = nw * 32;
  a.BO = wn - 2 * dilation; a.ntt = ceil_div(length, a.BO);
  a.out_slope = out_slope; a.in_slope = in_slope;
  a.vec = (((reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(u)) & 15) == 0 && (length & 3) == 0) ? 1 : 0;
  if ((long long)a.B * a.ntt > 0x7fffffffLL) return fail(EBEN_EINVAL, "ResidualUnit grid too large");
  const int np = rs_pieces(math);
  // entries per barrier: ~8-16 KB of weights per group
  switch (np) {
    case 1:
      switch (channels / 32) {
        case 1: return launch_ru3_bwd<1, 4, 1, 1>(a, st);
        case 2: return launch_ru3_bwd<2, 4, 1, 2>(a, st);
        default: return nw == 4 ? launch_ru3_bwd<4, 4, 1, 2>(a, st) : launch_ru3_bwd<4, 2, 1, 2>(a, st);
      }
    case 2:
      switch (channels / 32) {
        case 1: return launch_ru3_bwd<1, 4, 2, 1>(a, st);
        case 2: return launch_ru3_bwd<2, 4, 2, 2>(a, st);
        default: return rs_waves(128) == 4 ? launch_ru3_bwd<4, 4, 2, 1>(a, st) : launch_ru3_bwd<4, 2, 2, 1>(a, st);
      }
    default:
      switch (channels / 32) {
        case 1: return launch_ru3_bwd<1, 4, 3, 1>(a, st);
        case 2: return launch_ru3_bwd<2, 4, 3, 1>(a, st);
        default: return rs_waves(128) == 4 ? launch_ru3_bwd<4, 4, 3, 1>(a, st) : launch_ru3_bwd<4, 2, 3, 1>(a, st);
      }
  }
}

}  // namespace eben

using namespace eben;

extern "C" size_t eben_ru_packed_floats_ex(int channels, int math) {
  if (math == EBEN_MATH_F32) return eben_ru_packed_floats(channels);
  return ru3_packed_floats(channels, math);
}

extern "C" int eben_ru_supported(int channels, int dilation, int math) {
  if (math == EBEN_MATH_F32) return (channels == 32 || channels == 64 || channels == 128) && dilation >= 1 && dilation <= 16;
  return ru3_supported(channels, dilation, math);
}

extern "C" int eben_ru_pack_ex(int channels, int math, int which, const float* v_dil, const float* scale_dil, const float* v_pw,
                               const float* scale_pw, float* wimg, void* stream) {
  EBEN_REQUIRE(which == 0 || which == 1, "ru_pack_ex: which = 0 (forward image) or 1 (backward image)");
  if (math == EBEN_MATH_F32)
    return which == 0 ? eben_ru_pack(channels, v_dil, scale_dil, v_pw, scale_pw, wimg, stream)
                      : eben_ru_pack_bwd(channels, v_dil, scale_dil, v_pw, scale_pw, wimg, stream);
  EBEN_REQUIRE(ru3_supported(channels, 1, math), "fused ResidualUnit: 32, 64 or 128 channels and a known math mode (got %d, %d)", channels, math);
  EBEN_REQUIRE(v_dil && v_pw && wimg, "null pointer in ru_pack_ex");
  return ru3_pack(channels, math, which, v_dil, scale_dil, v_pw, scale_pw, wimg, as_stream(stream));
}

extern "C" int eben_ru_pack_multi(const EbenRuPackJob* jobs, int n, void* stream) {
  EBEN_REQUIRE(n >= 0 && (n == 0 || jobs != nullptr), "bad ru pack job list");
  for (int base = 0; base < n; base += RU_PACK_MULTI) {
    RuPackTable T;
    T.n = n - base < RU_PACK_MULTI ? n - base : RU_PACK_MULTI;
    T.first[0] = 0;
    for (int j = 0; j < T.n; ++j) {
      const EbenRuPackJob& jb = jobs[base + j];
      EBEN_REQUIRE(jb.math != EBEN_MATH_F32, "ru_pack_multi: the split-operand images only (pack the fp32 MFMA images with eben_ru_pack)");
      EBEN_REQUIRE(ru3_supported(jb.channels, 1, jb.math) && (jb.which == 0 || jb.which == 1) && jb.v_dil && jb.v_pw && jb.wimg, "bad ru pack job %d", base + j);
      const int CT = jb.channels / 32;
      T.job[j].vd = jb.v_dil; T.job[j].sd = jb.scale_dil; T.job[j].vp = jb.v_pw; T.job[j].sp = jb.scale_pw;
      T.job[j].img = reinterpret_cast<u32x4*>(jb.wimg);
      T.job[j].CT = (short)CT; T.job[j].NP = (short)rs_pieces(jb.math); T.job[j].which = (short)jb.which; T.job[j].pad = 0;
      T.first[j + 1] = (unsigned short)(T.first[j] + ceil_div(8 * CT * CT * 64, 256));
    }
    hipLaunchKernelGGL(ru3_pack_multi_kernel, dim3(T.first[T.n]), dim3(256), 0, as_stream(stream), T);
    EBEN_CHECK_LAUNCH("ru3_pack_multi_kernel");
  }
  return EBEN_OK;
}

extern "C" int eben_ru_fwd_ex(int math, int batch, int channels, int length, int dilation, const float* x, float in_slope, float out_slope,
                              const float* wimg, float* y, float* h, float* u, void* stream) {
  if (math == EBEN_MATH_F32) return eben_ru_fwd(batch, channels, length, dilation, x, in_slope, out_slope, wimg, y, h, u, stream);
  EBEN_REQUIRE(ru3_supported(channels, dilation, math), "fused ResidualUnit (split bf16): 32 / 64 / 128 channels, dilation 1..%d (got %d, %d)",
               RS_DMAX, channels, dilation);
  EBEN_REQUIRE(batch > 0 && length > 0 && dilation < length, "bad ResidualUnit geometry");
  EBEN_REQUIRE(x && wimg && y, "null pointer in ru_fwd_ex");
  return ru3_fwd(math, batch, channels, length, dilation, x, in_slope, out_slope, wimg, y, h, u, as_stream(stream));
}

extern "C" int eben_ru_bwd_ex(int math, int batch, int channels, int length, int dilation, const float* gy, const float* u, float out_slope,
                              const float* x, float in_slope, const float* post, const float* wimg_bwd, float* gx, float* gh, void* stream) {
  if (math == EBEN_MATH_F32) return eben_ru_bwd(batch, channels, length, dilation, gy, u, out_slope, x, in_slope, post, wimg_bwd, gx, gh, stream);
  EBEN_REQUIRE(ru3_supported(channels, dilation, math), "fused ResidualUnit (split bf16): 32 / 64 / 128 channels, dilation 1..%d (got %d, %d)",
               RS_DMAX, channels, dilation);
  EBEN_REQUIRE(batch > 0 && length > 0 && dilation < length, "bad ResidualUnit geometry");
  EBEN_REQUIRE(gy && u && wimg_bwd && gx && gh, "null pointer in ru_bwd_ex");
  EBEN_REQUIRE(in_slope == 1.f || x, "x is required to differentiate the fused input activation");
  return ru3_bwd(math, batch, channels, length, dilation, gy, u, out_slope, x, in_slope, post, wimg_bwd, gx, gh, as_stream(stream));
}

extern "C" int eben_rubl_fwd(int math, int batch, int channels, int length, int dilation, const float* x, float in_slope, float out_slope,
                             const float* wimg, float* y, void* xb, void* hb, void* umask, void* stream) {
  EBEN_REQUIRE(ru3_supported(channels, dilation, math), "fused ResidualUnit (split bf16): 32 / 64 / 128 channels, dilation 1..%d (got %d, %d)",
               RS_DMAX, channels, dilation);
  EBEN_REQUIRE(batch > 0 && length > 0 && dilation < length, "bad ResidualUnit geometry");
  EBEN_REQUIRE(x && wimg && y && xb && hb && umask, "null pointer in rubl_fwd");
  return ru3_fwd_bl(math, batch, channels, length, dilation, x, in_slope, out_slope, wimg, y, xb, hb, umask, as_stream(stream));
}
