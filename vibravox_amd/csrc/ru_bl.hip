// ru_bl.hip -- the generator's ResidualUnit backward with what the forward saved AT REST AS bf16 BUNDLES (gfx950).
//
// The bf16 generator backward (gen_backward_math = "bf16") rounds every saved tensor to bf16 on load, so the fp32 copies the forward
// wrote (h, u, and x re-read) were twice the bytes the backward can use.  Here the forward (ru3_fwd_kernel<.., BL = true>, ru_split.hip)
// writes exactly what the two backward launches consume, once, in the layout their MFMA operands want:
//   XB   bf16(xin) = bf16(lrelu(x))   [item][C / 8][L][8]   16-byte unit = 8 channels at one position = one lane's B operand of a k-step
//   HB   bf16(h)                      same                   (h = dilated conv of xin)
//   UM   sign bits of u               [item][C / 8][L] bytes, bit e = (u[8 g + e] > 0)    (u is only ever needed for lrelu'(u))
// and this file holds the two consumers (eben_generator.py:287-316, the backward of `x + lrelu(pointwise(dilated(x)))`):
//   rubl_bwd_kernel   input-gradient chain of the unit: g_z = g_y lrelu'(u) -> g_h = W_pw^T g_z -> g_x = (g_y + fold(W_dil^T g_h)) lrelu'(x) + skip;
//                     the fp32 gradient chain (g_y in, g_x out) stays fp32 [item][C][L]; g_z and g_h -- MFMA operands of the weight gradients
//                     only -- leave as bf16 bundles GZB / GHB.  The window lives in LDS as bundles (half of ru3_bwd's fp32 rows: two to
//                     three blocks per CU at 128 channels instead of one), a B fragment is ONE ds_read_b128 instead of eight ds_read_b32 +
//                     conversions, the dilated shifts are whole units.
//   rubl_dw_kernel    both weight gradients, reduction ALONG TIME on ds_read_b64_tr_b16 fragments (the bl_dw.hip idiom): all four operand
//                     tiles (GZB, HB, GHB, XB) are plain LDS-DMA copies of units, a dilated tap is a whole-unit offset into the XB rows
//                     (reflect padding = the source address of the copy), no conversion, no register staging: 8 bytes per element and
//                     launch instead of 20.
// Per element of the unit: forward 4 (x) + 4 (y) + 2 + 2 + 1/8, backward 4 (g_y) + 1/8 + 4 (g_x) + 2 + 2, weight gradients 8 -- 32.3 bytes
// against 52-56 for the fp32-at-rest path.  Arithmetic: the same roundings of the same values as ru3_bwd<NP = 1> / ru_dw<NP = 1> (RNE to
// bf16 of xin, h, g_z, g_h; fp32 accumulation; the residual path g_y -> g_x in fp32).
#include "common.h"

#include <cstdlib>

namespace eben {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int RUBL_DMAX = 9;   // largest dilation (EBEN: 1, 3, 9)

__device__ __forceinline__ unsigned rb_pack_bf16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));   // v_cvt_pk_bf16_f32 (RNE)
}

__device__ u32x4 rubl_zero_unit = {0u, 0u, 0u, 0u};

// ---------------------------------------------------------------------------------------------------------------------------------
// Input gradients.  Block = NW waves = one item x a window of WN = 32 NW columns, of which BO = WN - 2 d are outputs (g_h is needed d
// columns to either side of an output, so it is recomputed in the halo).  Weight entries as ru3_bwd (ru_split.hip, NP = 1 image of
// ru3_pack_bwd): entry = 32 reduction channels = two k-steps; stage A (CT entries), three taps, the two reflect folds.
// ---------------------------------------------------------------------------------------------------------------------------------
struct RublBwdArgs {
  const float* gy; const unsigned char* um; const u32x4* wimg; const float* xmask; const float* post;
  float* gx; u32x4* gzb; u32x4* ghb;
  int B, L, d, ntt, BO;
  float out_slope, in_slope;
};

template <int CT, int NW, int G>
__global__ __launch_bounds__(NW * 64, CT == 4 ? 2 : CT == 2 ? 3 : 4) void rubl_bwd_kernel(const RublBwdArgs P) {
  constexpr int NT = NW * 64, WN = NW * 32, C = 32 * CT, CB = C / 8;
  constexpr int GS = WN;               // units per bundle row of the window
  constexpr int U = CT * 64;           // weight units per k-step
  constexpr int EU = 2 * U;            // ... per entry
  constexpr int WCHU = G * EU;
  static_assert(CT % G == 0, "an entry group never straddles a stage");
  static_assert(NT == 2 * WN, "staging: a thread keeps one column and every other bundle");

  extern __shared__ __attribute__((aligned(16))) u32x4 rb_smem[];
  u32x4* Ws = rb_smem;                 // 2 x WCHU
  u32x4* Gs = rb_smem + 2 * WCHU;      // CB rows of GS units: g_z, then g_h

  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tt = __builtin_amdgcn_readfirstlane(blockIdx.x % P.ntt);
  const int b = __builtin_amdgcn_readfirstlane(blockIdx.x / P.ntt);
  const int d = P.d, L = P.L, BO = P.BO;
  const int t0 = tt * BO;
  const int w0 = t0 - d;               // signal position of window column 0
  const long long rowbase = (long long)b * C * L;
  const long long ubase = (long long)b * CB * L;

  const bool fold_l = t0 <= d && L > 1;
  const bool fold_r = t0 + BO > L - 1 - d && t0 <= L - 2;
  const int NS = 4 * CT + (fold_l ? CT : 0) + (fold_r ? CT : 0);
  const int NSG = NS / G;
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

  // ---- stage the window: thread = one column, every other bundle; unit = bf16(g_y lrelu'(u)) of 8 channels, zero outside the signal;
  // the columns this block owns also go to GZB (the pointwise weight gradient's operand) ----
  {
    const int p = tid & (WN - 1), gpar = tid / WN;
    const int q = w0 + p;
    const bool ok = q >= 0 && q < L;
    const bool own = p >= d && p < d + BO && q < L;
    const float* gyc = P.gy + rowbase + (ok ? q : 0);
    const unsigned char* umc = P.um + ubase + (ok ? q : 0);
    constexpr int UB = CB >= 8 ? 4 : CB / 2;       // units in flight per thread (CB / 2 units per thread in all)
#pragma unroll
    for (int k0 = 0; k0 < CB / 2; k0 += UB) {
      float v[UB][8];
      unsigned m[UB];
#pragma unroll
      for (int k = 0; k < UB; ++k) {
        const int g = gpar + 2 * (k0 + k);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[k][e] = gyc[(long long)(8 * g + e) * L];
        m[k] = umc[(long long)g * L];
      }
#pragma unroll
      for (int k = 0; k < UB; ++k) {
        const int g = gpar + 2 * (k0 + k);
        float z[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = ok ? v[k][e] * (((m[k] >> e) & 1u) ? 1.f : P.out_slope) : 0.f;
        u32x4 un;
#pragma unroll
        for (int e = 0; e < 4; ++e) un[e] = rb_pack_bf16(z[2 * e], z[2 * e + 1]);
        Gs[g * GS + p] = un;
        if (own) P.gzb[ubase + (long long)g * L + q] = un;
      }
    }
  }
  __syncthreads();

  f32x16 acc1[CT], acc2[CT];
#pragma unroll
  for (int i = 0; i < CT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc1[i][r] = 0.f; acc2[i][r] = 0.f; }

  const int wc = wn * 32 + (lane & 31);        // window / output column of this lane
  const int hf = lane >> 5;
  const u32x4* gbA = Gs + hf * GS + wc;
  // stage B shifts a column by up to 2 d: lanes beyond the BO output columns (their results are dropped) stay inside the row
  const u32x4* gbB = Gs + hf * GS + (wc < BO ? wc : BO - 1);

  // one entry (32 reduction channels = bundles 4 cb .. 4 cb + 3 = two k-steps) from slot `wslot` of the current weight buffer
  auto entry = [&](const u32x4* wslot, int cb, const u32x4* gb, int off, f32x16 (&acc)[CT], bool sel) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      u32x4 bq = gb[(4 * cb + 2 * kk) * GS + off];
      if (!sel) bq = u32x4{0u, 0u, 0u, 0u};
      u32x4 a[CT];
#pragma unroll
      for (int i = 0; i < CT; ++i) a[i] = wslot[kk * U + i * 64];
#pragma unroll
      for (int i = 0; i < CT; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, bq), acc[i], 0, 0, 0);
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
      // in place: a wave reads and writes only its own 32 columns (every bundle row), so no barrier separates the two.  Registers
      // 4 q' .. 4 q' + 3 of tile i are rows 8 q' + 4 hf + e = one 8-byte half of the unit (bundle 4 i + q', column)
      const int q = w0 + wc;
      const bool in = q >= 0 && q < L;          // nothing of g_h exists beyond the signal
      const bool own = wc >= d && wc < d + BO && q < L;
      uint2* __restrict__ ghu = reinterpret_cast<uint2*>(P.ghb + ubase + (own ? q : 0)) + hf;
      uint2* gsu = reinterpret_cast<uint2*>(Gs + wc) + hf;
#pragma unroll
      for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          uint2 v;
          v.x = in ? rb_pack_bf16(acc1[i][4 * qq], acc1[i][4 * qq + 1]) : 0u;
          v.y = in ? rb_pack_bf16(acc1[i][4 * qq + 2], acc1[i][4 * qq + 3]) : 0u;
          gsu[(4 * i + qq) * GS * 2] = v;
          if (own) ghu[(long long)(4 * i + qq) * L * 2] = v;
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

  // ---- epilogue: g_x = (acc + g_y) lrelu'(x) + post; the operands of a tile are read in one batch each ----
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
      off[r] = (unsigned)(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf) * (unsigned)L;
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
    __builtin_amdgcn_sched_barrier(0);   // tile by tile: all four tiles' operands in flight at once is 190 registers
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Weight gradients.
//   dW_pw [m][c]    = sum_{b,t} g_z[b,m,t] h[b,c,t]
//   dW_dil[m][c][j] = sum_{b,t} g_h[b,m,t] xin[b,c,reflect(t + (j-1) d)]
// Block = 4 waves = 32 RT output rows of BOTH gradients x one K slab (SEG positions of one item); wave 0 = pointwise (A = g_z rows,
// B = h), waves 1..3 = tap j = wave - 1 (A = g_h rows, B = xin shifted by (j - 1) d units); each wave RT x CT accumulator tiles.
// K chunk = BKT positions, double buffered: rows of units moved by LDS-DMA (one 64-lane piece per row; the xin rows carry the
// +-9-position halo), fragments by ds_read_b64_tr_b16 (bl_dw.hip: lane (G4, js, qs) supplies the 8-byte half qs & 1 of bundle
// +2 (G4 & 1) + (qs >> 1) at time step 8 (G4 >> 1) + js; row strides = 4 mod 16 units keep a half-wave on distinct banks).
// Slabs in ru_dw.hip's layout ([slab][C][C] and [slab][C][3 C]), summed in fixed order by eben_wn_bwd_multi.
// ---------------------------------------------------------------------------------------------------------------------------------
struct RublDwArgs {
  const u32x4* gz; const u32x4* h; const u32x4* gh; const u32x4* xb;
  float* slab_p; float* slab_d;
  int B, L, d, nseg, seg;
};

__device__ __forceinline__ bf16x8 rb_tr_frag(unsigned addr) {
  typedef __attribute__((address_space(3))) s16x4* lds4_t;
  const s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(size_t)(addr));
  const s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(size_t)(addr + 64u));   // + 4 units: k = 4..7 of the lane's eight
  const s16x8 v = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// one LDS-DMA piece (see bl_dw.hip: inline asm so that hipcc does not drain vmcnt in front of every transposing read)
__device__ __forceinline__ void rb_dma_piece(const u32x4* src, unsigned dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(dst) : "memory");
}

template <int CT, int RT, int BKT>
__global__ __launch_bounds__(256, (CT * RT >= 4) ? 2 : 3) void rubl_dw_kernel(const RublDwArgs P) {
  constexpr int C = 32 * CT, CB = C / 8, RB = 4 * RT;             // bundles per operand; A bundles per block and operand
  constexpr int TS = BKT + 4;                                      // A / h row stride in units (= 4 mod 16)
  constexpr int XW = BKT + 2 * RUBL_DMAX;                          // xin positions per chunk
  constexpr int XP = (XW + 63) / 64;                               // ... in 64-lane pieces
  constexpr int RS = XP * 64 + 4;                                  // xin row stride
  constexpr int A_UNITS = 2 * RB * TS, H_UNITS = CB * TS, X_UNITS = CB * RS;
  constexpr int BUF = A_UNITS + H_UNITS + X_UNITS;
  constexpr int NRG = CT / RT;                                     // row groups
  static_assert(BKT % 16 == 0 && (TS & 15) == 4 && (RS & 15) == 4 && CT % RT == 0, "tile geometry");
  extern __shared__ __attribute__((aligned(16))) u32x4 rbdw_smem[];
  typedef __attribute__((address_space(3))) void* lds_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);         // 0: pointwise, 1..3: tap wv - 1

  unsigned id = xcd_remap(blockIdx.x, gridDim.x);
  const int rg = __builtin_amdgcn_readfirstlane(id % NRG); id /= NRG;
  const int sgm = __builtin_amdgcn_readfirstlane(id % P.nseg);
  const int b = __builtin_amdgcn_readfirstlane(id / P.nseg);
  const int L = P.L;
  const int t_lo = sgm * P.seg;
  const int t_hi = t_lo + P.seg < L ? t_lo + P.seg : L;
  const int nchunks = (t_hi - t_lo + BKT - 1) / BKT;

  const unsigned lds0 = (unsigned)(unsigned long long)(lds_t)rbdw_smem;
  const u32x4* zero = &rubl_zero_unit;
  const long long ub = (long long)b * CB * L;

  // ---- tile movement: row r of a buffer -> its source row; rows are dealt to the waves round robin --------------------------------
  // rows: [0, RB) g_z bundles RB rg + r | [RB, 2 RB) g_h | [2 RB, 2 RB + CB) h bundles | then CB xin rows of XP pieces
  auto issue = [&](int q, int bsel) {
    const int t0 = t_lo + q * BKT;
    const unsigned dst = lds0 + (unsigned)(bsel * BUF * 16);
    const int ta = t0 + lane;
    const bool a_ok = lane < BKT && ta < t_hi;                     // A beyond the slab (or the signal) is zero: it masks the products
    const bool h_ok = lane < BKT && ta < L;
#pragma unroll
    for (int r = wv; r < 2 * RB + CB; r += 4) {
      const u32x4* src;
      bool ok;
      if (r < RB) { src = P.gz + ub + (long long)(RB * rg + r) * L + ta; ok = a_ok; }
      else if (r < 2 * RB) { src = P.gh + ub + (long long)(RB * rg + r - RB) * L + ta; ok = a_ok; }
      else { src = P.h + ub + (long long)(r - 2 * RB) * L + ta; ok = h_ok; }
      if (BKT == 64 || lane < BKT) rb_dma_piece(ok ? src : zero, __builtin_amdgcn_readfirstlane(dst + (unsigned)(r * TS * 16)));
    }
#pragma unroll
    for (int r = wv; r < CB * XP; r += 4) {
      const int g = r / XP, piece = r - g * XP;
      const int u = piece * 64 + lane;                             // unit of the row = position t0 - DMAX + u, reflected at the ends
      int pos = t0 - RUBL_DMAX + u;
      pos = pos < 0 ? -pos : pos;
      pos = pos >= L ? 2 * (L - 1) - pos : pos;
      const bool ok = pos >= 0 && pos < L;
      if (XW % 64 == 0 || u < XW)
        rb_dma_piece(ok ? P.xb + ub + (long long)g * L + pos : zero, __builtin_amdgcn_readfirstlane(dst + (unsigned)((A_UNITS + H_UNITS + g * RS + piece * 64) * 16)));
    }
  };

  // ---- per-lane fragment addresses (bytes inside a buffer) ---------------------------------------------------------------------------
  const int G4 = lane >> 4, js = (lane & 15) >> 2, qs = lane & 3;
  const int koff = 8 * (G4 >> 1) + js;
  const int fb = 2 * (G4 & 1) + (qs >> 1);                         // bundle inside a 32-row tile
  const int shift = wv == 0 ? 0 : (wv - 2) * P.d;
  unsigned aoff[RT], boff[CT];
#pragma unroll
  for (int i = 0; i < RT; ++i) aoff[i] = (unsigned)((((wv == 0 ? 0 : RB) + 4 * i + fb) * TS + koff) * 16 + 8 * (qs & 1));
#pragma unroll
  for (int c = 0; c < CT; ++c)
    boff[c] = wv == 0 ? (unsigned)((A_UNITS + (4 * c + fb) * TS + koff) * 16 + 8 * (qs & 1))
                      : (unsigned)((A_UNITS + H_UNITS + (4 * c + fb) * RS + RUBL_DMAX + shift + koff) * 16 + 8 * (qs & 1));

  f32x16 acc[RT][CT];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][c][r] = 0.f;

  int bsel = 0;
  if (nchunks > 0) issue(0, 0);
  for (int q = 0; q < nchunks; ++q) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of chunk q have landed ...
    __syncthreads();                                   // ... and everybody's; the other buffer (read during the previous chunk) is free
    if (q + 1 < nchunks) issue(q + 1, bsel ^ 1);
    const unsigned base = lds0 + (unsigned)(bsel * BUF * 16);
#pragma unroll
    for (int ks = 0; ks < BKT / 16; ++ks) {
      bf16x8 av[RT], bv[CT];
#pragma unroll
      for (int i = 0; i < RT; ++i) av[i] = rb_tr_frag(base + aoff[i] + (unsigned)(ks * 256));
#pragma unroll
      for (int c = 0; c < CT; ++c) bv[c] = rb_tr_frag(base + boff[c] + (unsigned)(ks * 256));
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[c], acc[i][c], 0, 0, 0);
    }
    bsel ^= 1;
  }

  // ---- this block's rows of slab (b, sgm): D tile column = lane & 31 (input channel), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) ----
  const int ks = b * P.nseg + sgm;
  if (wv == 0) {
    float* sp = P.slab_p + (long long)ks * C * C;
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int mm = 32 * (RT * rg + i) + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          sp[(long long)mm * C + c * 32 + (lane & 31)] = acc[i][c][r];
        }
  } else {
    float* sd = P.slab_d + (long long)ks * C * 3 * C;
    const int j = wv - 1;
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int mm = 32 * (RT * rg + i) + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          sd[(long long)mm * 3 * C + (c * 32 + (lane & 31)) * 3 + j] = acc[i][c][r];
        }
  }
}

// ---- host side -----------------------------------------------------------------------------------------------------------------
template <int CT, int NW, int G>
static int launch_rubl_bwd(const RublBwdArgs& a, hipStream_t st) {
  static LdsAttrOnce attr_once;
  auto kern = rubl_bwd_kernel<CT, NW, G>;
  const size_t lds = (size_t)2 * G * 2 * CT * 64 * 16 + (size_t)(4 * CT) * (NW * 32) * 16;
  {
    const hipError_t e = lds_attr_once(attr_once, reinterpret_cast<const void*>(kern));
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(rubl_bwd)");
  }
  hipLaunchKernelGGL(kern, dim3(a.B * a.ntt), dim3(NW * 64), lds, st, a);
  EBEN_CHECK_LAUNCH("rubl_bwd_kernel");
  return EBEN_OK;
}

template <int CT, int RT, int BKT>
static int launch_rubl_dw(const RublDwArgs& a, hipStream_t st) {
  static LdsAttrOnce attr_once;
  auto kern = rubl_dw_kernel<CT, RT, BKT>;
  constexpr int CB = 4 * CT, RB = 4 * RT, TS = BKT + 4, XP = (BKT + 2 * RUBL_DMAX + 63) / 64, RS = XP * 64 + 4;
  const size_t lds = (size_t)2 * 16 * (2 * RB * TS + CB * TS + CB * RS);
  {
    const hipError_t e = lds_attr_once(attr_once, reinterpret_cast<const void*>(kern));
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(rubl_dw)");
  }
  const long long nb = (long long)a.B * a.nseg * (CT / RT);
  if (nb > 0x7fffffffLL) return fail(EBEN_EINVAL, "rubl_dw grid too large");
  hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(256), lds, st, a);
  EBEN_CHECK_LAUNCH("rubl_dw_kernel");
  return EBEN_OK;
}

static int rubl_env(const char* name, int dflt) {
  const char* s = getenv(name);
  return s ? atoi(s) : dflt;
}

// K slab length: a multiple of 64; ~1024 positions at 32 channels, ~512 at 64, half an item at 128 (the slabs are C x 4 C floats each: what
// the weight-gradient launch writes and the slab sum reads back.  [MI355X, same box] step 10.4 ms at 512 / 512 / 512, 10.2 at 1024 / 512 / 512
// -- the launch alone is 2 us slower, the slab sum behind the generator's last weight gradients 8 MB lighter per unit; 1024 at 64 channels
// or 4096 at 32: no further gain / slower)
static int rubl_seg(int channels, int length) {
  static const int t32 = rubl_env("EBEN_RUBL_SEG32", 1024), t64 = rubl_env("EBEN_RUBL_SEG64", 512), t128 = rubl_env("EBEN_RUBL_SEG128", 512);
  const int target = channels == 32 ? t32 : channels == 64 ? t64 : t128;
  const int nseg = ceil_div(length, target > 64 ? target : 64);
  return round_up(ceil_div(length, nseg), 64);
}

}  // namespace eben

using namespace eben;

extern "C" int eben_rubl_supported(int channels, int dilation) {
  return (channels == 32 || channels == 64 || channels == 128) && dilation >= 1 && dilation <= RUBL_DMAX;
}

extern "C" int eben_rubl_bwd(int batch, int channels, int length, int dilation, const float* gy, const void* umask, float out_slope, const float* x,
                             float in_slope, const float* post, const float* wimg_bwd, float* gx, void* gzb, void* ghb, void* stream) {
  EBEN_REQUIRE(eben_rubl_supported(channels, dilation), "bundle-layout ResidualUnit backward: 32 / 64 / 128 channels, dilation 1..%d (got %d, %d)",
               RUBL_DMAX, channels, dilation);
  EBEN_REQUIRE(batch > 0 && length > 0 && dilation < length, "bad ResidualUnit geometry");
  EBEN_REQUIRE(gy && umask && wimg_bwd && gx && gzb && ghb, "null pointer in rubl_bwd");
  EBEN_REQUIRE(in_slope == 1.f || x, "x is required to differentiate the fused input activation");
  RublBwdArgs a;
  a.gy = gy; a.um = static_cast<const unsigned char*>(umask); a.wimg = reinterpret_cast<const u32x4*>(wimg_bwd);
  a.xmask = in_slope != 1.f ? x : nullptr; a.post = post; a.gx = gx; a.gzb = static_cast<u32x4*>(gzb); a.ghb = static_cast<u32x4*>(ghb);
  a.B = batch; a.L = length; a.d = dilation;
  static const int nw32 = rubl_env("EBEN_RUBL_NW32", 4), nw64 = rubl_env("EBEN_RUBL_NW64", 4), nw128 = rubl_env("EBEN_RUBL_NW128", 4);
  static const int g128 = rubl_env("EBEN_RUBL_G128", 2);
  const int nw = channels == 32 ? nw32 : channels == 64 ? nw64 : nw128;
  const int wn = (nw == 8 ? 8 : 4) * 32;
  a.BO = wn - 2 * dilation; a.ntt = ceil_div(length, a.BO);
  a.out_slope = out_slope; a.in_slope = in_slope;
  if ((long long)a.B * a.ntt > 0x7fffffffLL) return fail(EBEN_EINVAL, "ResidualUnit grid too large");
  hipStream_t st = as_stream(stream);
  switch (channels / 32) {
    case 1: return nw == 8 ? launch_rubl_bwd<1, 8, 1>(a, st) : launch_rubl_bwd<1, 4, 1>(a, st);
    case 2: return nw == 8 ? launch_rubl_bwd<2, 8, 2>(a, st) : launch_rubl_bwd<2, 4, 2>(a, st);
    default: return g128 == 1 ? launch_rubl_bwd<4, 4, 1>(a, st) : launch_rubl_bwd<4, 4, 2>(a, st);
  }
}

extern "C" int eben_rubl_dw_slabs(int batch, int channels, int length) {
  if (batch <= 0 || length <= 0 || (channels != 32 && channels != 64 && channels != 128)) return 0;
  return batch * ceil_div(length, rubl_seg(channels, length));
}

extern "C" int eben_rubl_dw(int batch, int channels, int length, int dilation, const void* gzb, const void* hb, const void* ghb, const void* xb,
                            float* slabs_pw, float* slabs_dil, void* stream) {
  EBEN_REQUIRE(eben_rubl_supported(channels, dilation), "bundle-layout ResidualUnit weight gradient: 32 / 64 / 128 channels, dilation 1..%d (got %d, %d)",
               RUBL_DMAX, channels, dilation);
  EBEN_REQUIRE(batch > 0 && length > 0 && dilation < length, "bad ResidualUnit geometry");
  EBEN_REQUIRE(gzb && hb && ghb && xb && slabs_pw && slabs_dil, "null pointer in rubl_dw");
  RublDwArgs a;
  a.gz = static_cast<const u32x4*>(gzb); a.h = static_cast<const u32x4*>(hb); a.gh = static_cast<const u32x4*>(ghb); a.xb = static_cast<const u32x4*>(xb);
  a.slab_p = slabs_pw; a.slab_d = slabs_dil;
  a.B = batch; a.L = length; a.d = dilation;
  a.seg = rubl_seg(channels, length); a.nseg = ceil_div(length, a.seg);
  hipStream_t st = as_stream(stream);
  static const int rt64 = rubl_env("EBEN_RUBL_RT64", 1), bkt128 = rubl_env("EBEN_RUBL_BKT128", 64), rt128 = rubl_env("EBEN_RUBL_RT128", 1);
  switch (channels / 32) {
    case 1: return launch_rubl_dw<1, 1, 64>(a, st);
    case 2: return rt64 == 2 ? launch_rubl_dw<2, 2, 64>(a, st) : launch_rubl_dw<2, 1, 64>(a, st);
    default:
      if (rt128 == 2) return launch_rubl_dw<4, 2, 32>(a, st);
      return bkt128 == 64 ? launch_rubl_dw<4, 1, 64>(a, st) : launch_rubl_dw<4, 1, 32>(a, st);
  }
}
