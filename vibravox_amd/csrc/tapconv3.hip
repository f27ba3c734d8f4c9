// tapconv3.hip -- bf16-operand form of the second-generation tap convolution (gfx950), selected per
// launch by EbenConv1dDesc.math == EBEN_MATH_BF16 (BASELINE config 2 names bf16 for the train step; the
// discriminator passes -- 91 % of the step's FLOPs -- run here, the generator stays on the exact-fp32 kernels).
//
//   y[b, g*Mg+m, t*OS+oo] = epi( sum_{c<Cg} sum_{j<J} W[j,c,m] * xin(b, g*Cg+c, t*S + off0 + j*dstep) )
//
// Activations, gradients, bias, the fused element-wise stages and the accumulators stay fp32 in HBM and in
// registers; only the two MFMA operands are rounded (RNE) to bf16 -- the weights by the pack kernel, the input
// tile as it is staged into LDS.  Same machine mapping as tapconv2.hip with every "channel" replaced by a
// BUNDLE of 8 channels (one 16-byte LDS unit):
//   * v_mfma_f32_32x32x16_bf16 (32 cycles, 16x the fp32 MFMA rate): one k-step = one tap x 16 channels; lane l
//     feeds column l&31 with channels 8*(l>>5) .. +7, i.e. ONE ds_read_b128 of a staged bundle row -- conflict
//     free for any stride / dilation because consecutive positions of a tile phase are consecutive units;
//   * weights: bf16 LDS image [k-step][row tile][lane] of 16-byte units, streamed with global_load_lds_dwordx4
//     in chunks of KSC k-steps, double-buffered, one barrier per chunk;
//   * the per-layer k-step table (scalar cache), phase geometry, double / triple-buffered input tiles and the
//     epilogue are the second generation's.
#include "common.h"
#include "tap3.h"

#include <cstdlib>

namespace eben {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));   // v_cvt_pk_bf16_f32 (RNE)
}

#ifndef EBEN_T3_KSC
#define EBEN_T3_KSC 4
#endif
#ifndef EBEN_T3_DBG
#define EBEN_T3_DBG 0   // scratch-build ablations (wrong results): 1 no weight stream, 2 no tile refresh, 4 no barrier, 8 no MFMA, 16 phase-major stores, 32 no mask / feature-matching loads in the bundle epilogue, 64 no tile staging in the prologue, 128 no output stores (bundle epilogue), 256 one k-step per chunk
#endif
#if EBEN_T3_DBG & 512
// scratch build: cycle stamps (s_memtime) of wave 0 of every block at the phase boundaries, one row of 8 per block
constexpr int T3_STAMP_ROWS = 65536;
__device__ unsigned long long t3_stamp[T3_STAMP_ROWS * 16];
#define T3_STAMP(k) do { if (tid == 0 && blockIdx.x < T3_STAMP_ROWS) { t3_stamp[blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime(); \
  if ((k) == 0) { t3_stamp[blockIdx.x * 16 + 8] = __builtin_amdgcn_s_getreg(4 | (31 << 11)); t3_stamp[blockIdx.x * 16 + 9] = __builtin_amdgcn_s_getreg(20 | (31 << 11)); } } } while (0)
#else
#define T3_STAMP(k) do { } while (0)
#endif
// Weight stream: chunks of KSC k-steps go into a RING of LDS slots by LDS-DMA, DIST = RING - 1 chunks ahead of the one being multiplied.
// The end of a chunk waits only for the NEXT chunk's pieces (s_waitcnt vmcnt(N) with the younger chunks' DMAs left in flight) and
// crosses a raw s_barrier: a __syncthreads() there carries a fence, i.e. vmcnt(0), which exposes one full DMA round trip
// (~1-1.8k cycles) per chunk of 128-576 MFMA cycles -- measured with cycle stamps: the loop of a 3-chunk input-gradient block took
// 2.9k cycles for 384 cycles of MFMAs, the 11-chunk hi + lo forward of a PQMF-band layer 20k for 6k.
// [MI355X] the stamps then showed that the DMA was NOT what a chunk waited for: with 4 slots of <= 8 KB (3 of the larger chunks) the loops
// shortened by 0-7 % only (the per-chunk cost is the k-step table's scalar load + barrier + first fragment reads, see the loop), while
// the larger LDS footprint took the thin launches from 2-3 blocks per CU to 1: whole discriminator forward 1.64 -> 2.18 ms.  Two slots
// (DIST = 1: the wait is vmcnt(0) again, one chunk of cover) is the default; the depth stays a build knob.
#ifndef EBEN_T3_RING_SMALL
#define EBEN_T3_RING_SMALL 2   // slots of chunks <= 8 KB
#endif
#ifndef EBEN_T3_RING_BIG
#define EBEN_T3_RING_BIG 2     // slots of larger chunks
#endif
#ifndef EBEN_T3_KSC_BIGFM
#define EBEN_T3_KSC_BIGFM EBEN_T3_KSC   // k-steps per chunk of the 96 / 128-row tiles (single-piece weights)
#endif
// k-steps (of 16 reduction elements) per weight chunk; split weights (NPW pieces per weight, EBEN_MATH_BF16X3 / X6): a k-step carries
// NPW times the weight bytes and 3 / 6 times the MFMAs
#ifndef EBEN_T3_KSC_X6_FM1
#define EBEN_T3_KSC_X6_FM1 2   // six-product launches (the generator's forward convs), 32-row tiles
#endif
#ifndef EBEN_T3_KSC_X6_FM2
#define EBEN_T3_KSC_X6_FM2 2   // ... 64-row tiles
#endif
#ifndef EBEN_T3_SPLIT_OCC2
#define EBEN_T3_SPLIT_OCC2 0   // 1: register budget of two blocks per CU for the split-weight launches with <= 64-row tiles
#endif
__host__ __device__ constexpr int t3_ksc(int npw, int fm = 4) {
  return npw == 1 ? (fm <= 2 ? EBEN_T3_KSC : EBEN_T3_KSC_BIGFM) : npw == 3 ? (fm == 1 ? EBEN_T3_KSC_X6_FM1 : fm == 2 ? EBEN_T3_KSC_X6_FM2 : 2) : 2;
}
#ifndef EBEN_T3_RING_SPLIT
#define EBEN_T3_RING_SPLIT 2   // slots of the split-weight (fp32 tensors at rest, NPW >= 2) launches: the generator's six-product convs
#endif
__host__ __device__ constexpr int t3_ring(int npw, int fm, bool bl = false) {
  return (!bl && npw >= 2) ? EBEN_T3_RING_SPLIT : t3_ksc(npw, fm) * npw * fm <= 8 ? EBEN_T3_RING_SMALL : EBEN_T3_RING_BIG;
}
// 16 zero bytes in device memory: where the lanes of an input-tile LDS-DMA piece that fall into the zero padding read from
__device__ u32x4 t3_zero_unit = {0u, 0u, 0u, 0u};

template <int N> __device__ __forceinline__ void t3_wait_vm() {   // s_waitcnt vmcnt(N) lgkmcnt(0)
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit count");
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | 0x70);
}


// NPX / NPW: pieces per operand.  The input operand is staged as NPX bf16 tiles, piece q = bf16(x - p0 - .. - p(q-1)) (every
// residual exact in fp32), the weights arrive as NPW pieces from the pack kernel, and a k-step issues the piece products
// a_q * b_s with q + s < max(NPW, NPX):
//   (1, 1) EBEN_MATH_BF16;   (1, 2) EBEN_MATH_BF16X2: x to ~2^-17, weights single bf16;
//   (2, 2) EBEN_MATH_BF16X3: three products, ~2^-17 per product;
//   (3, 3) EBEN_MATH_BF16X6: six products, all 24 mantissa bits of both operands, dropped terms <= 2^-26: fp32-grade products at
//          6/16 of the fp32 MFMA's cost (and 0.6 LDS fragment reads per MFMA instead of 1.25: this form is MFMA-bound).
// BL: both operands at rest in the bundle layout -- the input tile is a COPY of 16-byte units (8 channels at one position: exactly
// the staged LDS unit; no conversion, an eighth of the loads), the epilogue packs its four consecutive rows per lane into 8-byte
// halves of the output units (lanes 0-31 channels +0..3, lanes 32-63 channels +4..7 of a bundle: a wave stores 512 contiguous
// bytes per row quad) and reads the mask / feature-matching operands the same way.
template <int FM, int XRB, bool IM = false, int NPW = 1, int NPX = 1, bool BL = false>
__global__ __launch_bounds__(256, NPW >= 2 ? ((EBEN_T3_SPLIT_OCC2 && FM <= 2) ? 2 : 1) : 2) void tap3_kernel(const Tap3Args P) {
  constexpr int NT = 256, BN = 128, BM = FM * 32, KSC = t3_ksc(NPW, FM);
  constexpr bool SP = NPX > 1;
  constexpr int NPM = NPW > NPX ? NPW : NPX;
  constexpr int WCHU = KSC * NPW * FM * 64;       // 16-byte units per weight chunk
  constexpr int RING = t3_ring(NPW, FM, BL), DIST = RING - 1;
  constexpr int WU = (WCHU + NT - 1) / NT;        // LDS-DMA instructions per thread and chunk
  static_assert(WCHU % 64 == 0, "weight chunk must split into whole wave pieces");
  static_assert(DIST >= 1 && DIST <= 3, "ring depth");
  static_assert(!BL || DIST == 1, "the input-tile LDS-DMA of the bundle layout is waited for by the chunk's vmcnt(0)");

  extern __shared__ __attribute__((aligned(16))) u32x4 smem3[];
  u32x4* Ws = smem3;              // RING x WCHU
  u32x4* Xs = smem3 + RING * WCHU;   // nxb input tiles of CI_B * CSTRIDE units + one spare unit (SP: the lo tiles behind them)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  T3_STAMP(0);

  unsigned id;
  {
    const unsigned bid = blockIdx.x, xcd = bid & 7u, idx = bid >> 3;
    id = (xcd < P.xr ? xcd * (P.xq + 1) : P.xr * (P.xq + 1) + (xcd - P.xr) * P.xq) + idx;   // xcd_remap with the host's quotient
  }
  // the OS phases of one output tile interleave in memory (element t*OS + phase): neighbouring block ids, i.e. the same XCD at
  // about the same time, so that their partial cache lines meet in that XCD's L2
  // (MelGAN L2 / L3 / L4 input gradients at 128 items: 0.44 / 0.57 / 0.56 -> 0.36 / 0.50 / 0.46 ms against phase-slowest)
  int ph, tt, b, mt, g;
  if (P.id_fast) {
    // magic 0 = divisor 1 (ceil(2^32 / 1) does not fit)
    unsigned qd = P.m_nph ? __umulhi(id, P.m_nph) : id; ph = (int)(id - qd * (unsigned)P.nph); id = qd;
    qd = P.m_ntt ? __umulhi(id, P.m_ntt) : id; tt = (int)(id - qd * (unsigned)P.ntt); id = qd;
    qd = P.m_B ? __umulhi(id, P.m_B) : id; b = (int)(id - qd * (unsigned)P.B); id = qd;
    qd = P.m_nmt ? __umulhi(id, P.m_nmt) : id; mt = (int)(id - qd * (unsigned)P.nmt); g = (int)qd;
  } else {
    ph = id % P.nph; id /= P.nph;
    tt = id % P.ntt; id /= P.ntt;
    b = id % P.B; id /= P.B;
    mt = id % P.nmt;
    g = id / P.nmt;
  }
  ph = __builtin_amdgcn_readfirstlane(ph); tt = __builtin_amdgcn_readfirstlane(tt); b = __builtin_amdgcn_readfirstlane(b);
  mt = __builtin_amdgcn_readfirstlane(mt); g = __builtin_amdgcn_readfirstlane(g);
  const int t0 = tt * BN, m0 = mt * BM;

  int J, nt, oo, minoff, span;
  unsigned span_magic;
  if (ph < P.pg_n) {
    J = P.pg[ph].J; nt = P.pg[ph].nt; oo = P.pg[ph].oo; minoff = P.pg[ph].minoff; span = P.pg[ph].span; span_magic = P.pg[ph].span_magic;
  } else {
    const PhaseGeom q = phase_geom(P.mode, ph, P.J0, P.off0, P.nt, P.dstep, P.OS, P.ps_pad, P.ps_k, P.ps_d, P.ps_kstep, P.Ly);
    J = q.J; nt = q.nt; oo = q.oo; minoff = q.minoff;
    const int ad = P.dstep >= 0 ? P.dstep : -P.dstep;
    span = J > 0 ? (BN - 1) * P.S + (J - 1) * ad + 1 : 0;
    span_magic = span > 0 ? (unsigned)((0x100000000ull + (unsigned)span - 1) / (unsigned)span) : 0u;
  }
  if (t0 >= nt) return;
  const int KS_CC = J * P.CP;
  const int KS = P.ncc * KS_CC;
  const int nch = (KS + KSC - 1) / KSC;
  const int q0 = t0 * P.S + minoff;
  const int xtot = P.CI_B * span;          // bundle-positions per input tile
  const int XBUF = P.CI_B * P.CSTRIDE;
  const int LO = SP ? P.nxb * XBUF + 1 : 0;   // split input: unit offset from piece q to piece q + 1 of every tile slot

  const u32x4* wsrc = P.wp + (long long)ph * P.w_phase + ((long long)g * P.nmt + mt) * P.w_tile;
  typedef const __attribute__((address_space(4))) int* ctab_t;
  ctab_t tab = (ctab_t)(P.tab + (long long)ph * P.tab_phase);

  f32x16 acc[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // ---- input tile staging: one element = one bundle (8 channels) at one position -------------------
  const long long xrow0 = ((long long)b * P.Cx + (long long)g * P.Cg) * P.Lx;
  const int refl = P.reflect;
  // tiles that lie inside the signal (block-uniform: every tile but the first / last few of a row) need no bounds arithmetic
  const bool inside = q0 >= 0 && q0 + span <= P.Lx;
  auto x_pos = [&](int r, int& ok) -> int {
    int qq = q0 + r;
    if (inside) { ok = 1; return qq; }
    const int m1 = qq < 0 ? -qq : qq;
    const int m2 = m1 >= P.Lx ? 2 * (P.Lx - 1) - m1 : m1;
    qq = refl ? m2 : qq;
    ok = (int)(qq >= 0) & (int)(qq < P.Lx);
    return ok ? qq : 0;
  };
  auto x_slot = [&](int bb, int r) -> int {
    int p = 0, d = r;
    if (P.S != 1) { d = (int)__umulhi((unsigned)r, P.s_magic); p = r - d * P.S; }
    return bb * P.CSTRIDE + p * P.PLEN + d;
  };
  // block-uniform shortcuts of the staging arithmetic (the short layers spend most of their instructions here): whole bundles
  // (no channel of a bundle lies beyond the group) and no activation on load (the engine's launches: the producer applied it)
  const bool full_c = (P.Cg & 7) == 0;
  const bool plain_in = !IM && P.in_slope == 1.f;
  auto load8 = [&](const float* base, int c0, int qq, float (&v)[8]) {
    // channels c0 .. c0+7 of this group at position qq; rows past the group's last channel re-read row c0 (zeroed later)
    if (full_c) {
      const float* p = base + (long long)(c0 < P.Cg ? c0 : 0) * P.Lx + qq;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = p[(long long)e * P.Lx];
      return;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = c0 + e < P.Cg ? c0 + e : (c0 < P.Cg ? c0 : 0);
      v[e] = base[(long long)c * P.Lx + qq];
    }
  };
  auto cvt8 = [&](const float (&v)[8], const float (&mk)[8], int c0, int ok, u32x4 (&pc)[NPX]) {
    float t[8];
    if (plain_in && full_c) {
      const bool live = ok && c0 < P.Cg;
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] = live ? v[e] : 0.f;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float w = IM ? v[e] * dlrelu(mk[e], P.in_slope) : lrelu(v[e], P.in_slope);
        t[e] = (ok && c0 + e < P.Cg) ? w : 0.f;
      }
    }
#pragma unroll
    for (int q = 0; q < NPX; ++q) {
      u32x4 o;
      o[0] = pack_bf16(t[0], t[1]); o[1] = pack_bf16(t[2], t[3]); o[2] = pack_bf16(t[4], t[5]); o[3] = pack_bf16(t[6], t[7]);
      pc[q] = o;
      if (q + 1 < NPX) {
        // residual against the rounded value (a bf16 is the upper half of its fp32): exact in fp32
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          t[2 * e] -= __builtin_bit_cast(float, o[e] << 16);
          t[2 * e + 1] -= __builtin_bit_cast(float, o[e] & 0xffff0000u);
        }
      }
    }
  };

  int xg[XRB];
#pragma unroll
  for (int u = 0; u < XRB; ++u) {
    const int i = tid + u * NT;
    const int bb = (int)__umulhi((unsigned)i, span_magic);
    xg[u] = (P.ncc > 1 && i < xtot) ? ((bb << 16) | (i - bb * span)) : -1;
  }
  // BL: unit (bundle cbn of this group, position qq) of the input planes; bundles past the group's last re-read bundle 0 (zeroed later)
  const int CgB = P.Cg >> 3;
  const long long xrowB = ((long long)b * P.CBx + (long long)g * CgB) * P.Lx;
  auto loadu = [&](int cbn, int qq, u32x4 (&v)[NPX]) {
    const long long idx = xrowB + (long long)(cbn < CgB ? cbn : 0) * P.Lx + qq;
    v[0] = P.xh[idx];
    if constexpr (NPX > 1) v[1] = P.xl[idx];
  };
  float xreg[BL ? 1 : XRB][8], mreg[IM ? XRB : 1][8];
  u32x4 xru[BL ? XRB : 1][NPX];
  unsigned okmask = 0;
  auto fetch_x = [&](int cc) {
    const float* xp = P.x + xrow0;
    const float* mp = P.xmask + xrow0;
    okmask = 0;
#pragma unroll
    for (int u = 0; u < XRB; ++u) {
      int ok;
      const int qq = x_pos(xg[u] >= 0 ? (xg[u] & 0xffff) : 0, ok);   // lanes without a unit re-read the tile's first position
      ok &= (int)(xg[u] >= 0);
      if constexpr (BL) {
        const int cbn = cc * P.CI_B + (xg[u] >= 0 ? (xg[u] >> 16) : 0);
        ok &= (int)(cbn < CgB);
        okmask |= (unsigned)ok << u;
        loadu(cbn, qq, xru[u]);
      } else {
        okmask |= (unsigned)ok << u;
        load8(xp, cc * P.CI_T + (xg[u] >= 0 ? (xg[u] >> 16) * 8 : 0), qq, xreg[u]);
        if constexpr (IM) load8(mp, cc * P.CI_T + (xg[u] >= 0 ? (xg[u] >> 16) * 8 : 0), qq, mreg[u]);
      }
    }
  };
  const int dead_slot = P.nxb * XBUF;
  auto store_x = [&](int cc) {
    const int bsel = cc % P.nxb;
    u32x4* dst = Xs + bsel * XBUF;
#pragma unroll
    for (int u = 0; u < XRB; ++u) {
      const int bb = xg[u] >> 16;
      const int sl = xg[u] >= 0 ? x_slot(bb, xg[u] & 0xffff) : dead_slot - bsel * XBUF;
      if constexpr (BL) {
        const bool live = (okmask >> u) & 1u;
#pragma unroll
        for (int q = 0; q < NPX; ++q) dst[sl + q * LO] = live ? xru[u][q] : u32x4{0u, 0u, 0u, 0u};
      } else {
        u32x4 pc[NPX];
        cvt8(xreg[u], mreg[IM ? u : 0], cc * P.CI_T + (xg[u] >= 0 ? bb * 8 : 0), (int)((okmask >> u) & 1u), pc);
#pragma unroll
        for (int q = 0; q < NPX; ++q) dst[sl + q * LO] = pc[q];
      }
    }
  };

  // Bundle layout at stride 1 (every input gradient, the stride-1 forwards): a tile row -- one bundle over `span` consecutive positions
  // -- is a run of consecutive 16-byte units both in the plane and in LDS, so the tile is MOVED by LDS-DMA in pieces of 64 units (one
  // per wave instruction; lanes past the row's end masked off, lanes in the zero padding / past the group's last bundle reading the
  // zero unit) instead of loaded, selected and written unit by unit: ~8 vector instructions per piece against ~40 per unit pair, no
  // ds_write, no staging registers.  (The thin layers are bound by vector-instruction issue, see the epilogue.)
  // [MI355X, 64 / 128 rows] single-tile launches (PQMF-band input gradients) 0.070 / 0.063 / 0.071 -> 0.067 / 0.058 / 0.068 ms; with
  // the tile refreshed inside the loop (MelGAN L3 input gradient, L5 forward) the DMA issued at the hand-over has one chunk to land
  // where the register path had its loads in flight a chunk earlier: 0.405 -> 0.429, 0.091 -> 0.104 -- those keep the register path
  const bool dma_x = BL && P.S == 1 && P.ncc == 1 && !(EBEN_T3_DBG & 1024);
  auto dma_tile = [&](int cc, u32x4* dst) {
    const int ppr = (span + 63) >> 6;   // pieces per bundle row
    int bb = 0, pr = __builtin_amdgcn_readfirstlane(tid >> 6);
    while (pr >= ppr) { pr -= ppr; ++bb; }
    while (bb < P.CI_B) {
      const int r = pr * 64 + lane;
      const int qq = q0 + r;
      const int cbn = cc * P.CI_B + bb;
      const bool in = qq >= 0 && qq < P.Lx && cbn < CgB;
      const long long idx = xrowB + (long long)(cbn < CgB ? cbn : 0) * P.Lx + (in ? qq : 0);
      u32x4* d = dst + bb * P.CSTRIDE + pr * 64;   // uniform
      if (r < span) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(in ? P.xh + idx : &t3_zero_unit),
                                         (__attribute__((address_space(3))) void*)d, 16, 0, 0);
        if constexpr (NPX > 1)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(in ? P.xl + idx : &t3_zero_unit),
                                           (__attribute__((address_space(3))) void*)(d + LO), 16, 0, 0);
      }
      pr += NT / 64;
      while (pr >= ppr) { pr -= ppr; ++bb; }
    }
  };

  auto issue_w = [&](int ch) {
    const u32x4* src = wsrc + (long long)ch * WCHU;
    u32x4* dst = Ws + (ch % RING) * WCHU;
#pragma unroll
    for (int u = 0; u < WU; ++u) {
      int idx = u * NT + tid;
      // every wave issues WU instructions per chunk (the counted waits below rely on it): a wave past the chunk's end repeats its
      // previous piece (same bytes to the same place)
      if (WCHU % NT != 0 && idx >= WCHU) idx -= NT;   // wave-uniform
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + idx),
                                       (__attribute__((address_space(3))) void*)(dst + (idx & ~63)), 16, 0, 0);
    }
  };

  int written = 0;
  T3_STAMP(1);
  // ---- bundle layout: what the epilogue needs besides the accumulators is asked for HERE, under the tile staging and the reduction ----
  // (cycle stamps: the epilogue written as load-then-use took 4-10k cycles of a 15-37k block life: the bias values one exec-masked load
  // each, the feature-matching sums a dependent round trip of their own, several kernel-argument loads in sequence)
  const int eb = P.em_seg > 0 ? P.em_map[(int)(b >= P.em_seg) + (int)(b >= 2 * P.em_seg) + (int)(b >= 3 * P.em_seg)] * P.em_seg +
                                    (b - ((int)(b >= P.em_seg) + (int)(b >= 2 * P.em_seg) + (int)(b >= 3 * P.em_seg)) * P.em_seg) : b;
  float* Bs = reinterpret_cast<float*>(Xs + NPX * (P.nxb * XBUF + 1));   // BM bias values of this block's rows (BL)
  float fk1 = 0.f, fk2 = 0.f;
  const bool fmr = BL && P.fm_sums != nullptr && P.res_rows > 0 && b < P.res_rows;
  if constexpr (BL) {
    if (tid < BM) {
      const int m = m0 + tid;
      Bs[tid] = P.bias ? P.bias[(long long)g * P.Mg + (m < P.Mg ? m : P.Mg - 1)] : 0.f;
    }
    if (fmr) {
      typedef const __attribute__((address_space(4))) float* cf_t;
      const float s1 = ((cf_t)P.fm_sums)[0], s2 = ((cf_t)P.fm_sums)[1];
      fk1 = P.fm_gs / s2; fk2 = P.fm_gs * s1 / (s2 * s2);
    }
  }
  if (nch > 0) {
    // the first DIST chunks: their LDS-DMA round trips run under the tile staging (short reductions -- the PQMF-band layers'
    // input gradients: 3 chunks -- are complete before the loop starts)
    issue_w(0);
#pragma unroll
    for (int c = 1; c < DIST; ++c)
      if (c < nch) issue_w(c);
    const float* xp = P.x + xrow0;
    const float* mp = P.xmask + xrow0;
    // two rounds of (2 units = 16 loads per thread) in flight: round r + 1 is asked for before round r is converted and written --
    // a single-tile layer (every PQMF-band layer, the k = 1 STFT contractions) runs 3-5 such rounds back to back with nothing else
    // of the block to hide them behind
    struct Round { float v[BL ? 1 : 2][8]; float mk[IM ? 2 : 1][8]; u32x4 w[BL ? 2 : 1][NPX]; int sl[2], ok[2], c0[2]; };
    auto pro_load = [&](int base, Round& R) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int i = base + tid + u * NT;
        const int bb = (int)__umulhi((unsigned)i, span_magic);
        const int r = i - bb * span;
        const int qq = x_pos(i < xtot ? r : 0, R.ok[u]);
        R.ok[u] &= (int)(i < xtot);
        if constexpr (BL) {
          R.c0[u] = i < xtot ? bb : 0;
          R.ok[u] &= (int)(R.c0[u] < CgB);
          loadu(R.c0[u], qq, R.w[u]);
        } else {
          R.c0[u] = i < xtot ? bb * 8 : 0;
          load8(xp, R.c0[u], qq, R.v[u]);
          if constexpr (IM) load8(mp, R.c0[u], qq, R.mk[u]);
        }
        R.sl[u] = i < xtot ? x_slot(bb, r) : -1;
      }
    };
    auto pro_store = [&](Round& R) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (R.sl[u] >= 0) {
          if constexpr (BL) {
#pragma unroll
            for (int q = 0; q < NPX; ++q) Xs[R.sl[u] + q * LO] = R.ok[u] ? R.w[u][q] : u32x4{0u, 0u, 0u, 0u};
          } else {
            u32x4 pc[NPX];
            cvt8(R.v[u], R.mk[IM ? u : 0], R.c0[u], R.ok[u], pc);
#pragma unroll
            for (int q = 0; q < NPX; ++q) Xs[R.sl[u] + q * LO] = pc[q];
          }
        }
    };
    Round ra, rb;
    if (dma_x) dma_tile(0, Xs);
    else if (xtot > 0) pro_load(0, ra);
    T3_STAMP(2);
    for (int base = 0; base < (((EBEN_T3_DBG & 64) || dma_x) ? 0 : xtot); base += 4 * NT) {
      const bool second = base + 2 * NT < xtot;
      if (second) pro_load(base + 2 * NT, rb);
      pro_store(ra);
      if (second) {
        if (base + 4 * NT < xtot) pro_load(base + 4 * NT, ra);
        pro_store(rb);
      }
    }
    if (P.ncc > 1 && !dma_x) fetch_x(1);
  }
  T3_STAMP(3);
  __syncthreads();
  T3_STAMP(4);
  // activation mask (hi plane of the saved embedding) of this lane's outputs: asked for before the reduction where the registers allow
  constexpr bool PREF = BL && FM <= 2;
  uint2 pah[PREF ? FM : 1][4];
  if constexpr (PREF) {
    if (P.eh != nullptr && P.pr_S == 0 && !(EBEN_T3_DBG & 32)) {
      const int tq = t0 + wn * 32 + (lane & 31);
      const unsigned loffq = (((unsigned)(tq < nt ? tq : nt - 1) * (unsigned)P.OS + (unsigned)oo) * 2u + (unsigned)(lane >> 5)) * 8u;
      const long long Lrowq = (long long)P.Ly * 16;
      const char* ehq = reinterpret_cast<const char*>(P.eh) + (long long)eb * P.CBy * Lrowq + ((long long)((g * P.Mg + m0) >> 3)) * Lrowq;
      const int quadsq = (P.Mg - m0 + 7) >> 3;
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int q = 4 * i + r4;
          pah[i][r4] = *reinterpret_cast<const uint2*>(ehq + (long long)(q < quadsq ? q : 0) * Lrowq + loffq);
        }
    }
  }

  const int lanebase = (lane >> 5) * P.CSTRIDE + wn * 32 + (lane & 31);
  int te[KSC];
#pragma unroll
  for (int ks = 0; ks < KSC; ++ks) te[ks] = nch > 0 ? tab[ks] : 0;
  int pending = -1;   // tile whose global loads are issued at the top of the next chunk (not in front of the barrier)
  // The k-steps of a chunk run in two halves.  The first half's fragments are read at the top of the chunk, and the MFMAs issued while
  // they are in flight are the SECOND half of the PREVIOUS chunk, held back in registers across the barrier: behind every barrier a wave
  // has H k-steps of matrix work that needs no LDS, so the first fragment reads of a chunk (300-400 cycles with four to eight waves
  // reading at once) are no longer what the first MFMA of every chunk waits for.  Same order of accumulation as chunk by chunk.
  // Not where the mask-on-load staging of the generator backward holds 2 x 5 x 8 tile registers beside four accumulator tiles (the
  // held-back fragments do not fit in the 256 registers of two blocks per CU: 72 bytes of scratch): those instantiations read two
  // k-steps ahead inside the chunk.
  constexpr bool HOLD = !(IM && FM == 4 && XRB == 5);
  constexpr int H = !HOLD ? 2 : KSC >= 2 ? KSC / 2 : 1, H2 = HOLD ? KSC - H : 0;
  u32x4 bvA[H][NPX], aA[H][NPW][FM], bvB[H2 > 0 ? H2 : 1][NPX], aB[H2 > 0 ? H2 : 1][NPW][FM];
  auto mma = [&](const u32x4 (&bq)[NPX], const u32x4 (&aq)[NPW][FM]) {
    // piece products, smallest first: (qw, qx) with qw + qx = lvl
#pragma unroll
    for (int lvl = NPM - 1; lvl >= 0; --lvl)
#pragma unroll
      for (int qw = 0; qw < NPW; ++qw) {
        const int qx = lvl - qw;
        if (qx < 0 || qx >= NPX) continue;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          if (EBEN_T3_DBG & 8) acc[i][0] += __builtin_bit_cast(float, aq[qw][i][0] ^ bq[qx][i & 3]);
          else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aq[qw][i]), __builtin_bit_cast(bf16x8, bq[qx]), acc[i], 0, 0, 0);
        }
      }
  };
  for (int ch = 0; ch < nch; ++ch) {
    // the next chunk's table entries are asked for FIRST: their scalar-load round trip (~300 cycles) runs under this chunk's fragment
    // reads and MFMAs -- asked for behind the MFMAs it was waited for in front of every barrier
    int tn[KSC];
#pragma unroll
    for (int ks = 0; ks < KSC; ++ks) tn[ks] = tab[(ch + 1 < nch ? ch + 1 : ch) * KSC + ks];
    if ((EBEN_T3_DBG & 1) == 0 && ch + DIST < nch) issue_w(ch + DIST);   // into the slot of chunk ch - 1 (behind the barrier that ended it)
    if ((EBEN_T3_DBG & 2) == 0 && pending >= 0 && !dma_x) { fetch_x(pending); pending = -1; }
    const u32x4* wb = Ws + (ch % RING) * WCHU + lane;
    const u32x4* xb = Xs + lanebase;
    auto rd = [&](int ks, u32x4 (&bq)[NPX], u32x4 (&aq)[NPW][FM]) {
#pragma unroll
      for (int q = 0; q < NPX; ++q) bq[q] = xb[te[ks] + q * LO];
#pragma unroll
      for (int q = 0; q < NPW; ++q)
#pragma unroll
        for (int i = 0; i < FM; ++i) aq[q][i] = wb[((ks * NPW + q) * FM + i) * 64];
    };
    if constexpr (HOLD) {
#pragma unroll
      for (int h = 0; h < H; ++h) rd(h, bvA[h], aA[h]);
      __builtin_amdgcn_sched_barrier(0);
      if (ch > 0) {
#pragma unroll
        for (int h = 0; h < H2; ++h) mma(bvB[h], aB[h]);       // the held-back half of chunk ch - 1
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int h = 0; h < H2; ++h) rd(H + h, bvB[h], aB[h]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int h = 0; h < H; ++h) mma(bvA[h], aA[h]);
      __builtin_amdgcn_sched_barrier(0);
    } else {   // fragments of k-step ks + 2 asked for before the products of k-step ks (two register slots, used alternately)
      rd(0, bvA[0], aA[0]);
      if (KSC > 1) rd(1, bvA[1], aA[1]);
#pragma unroll
      for (int ks = 0; ks < KSC; ++ks) {
        u32x4 bq[NPX], aq[NPW][FM];
#pragma unroll
        for (int q = 0; q < NPX; ++q) bq[q] = bvA[ks & 1][q];
#pragma unroll
        for (int q = 0; q < NPW; ++q)
#pragma unroll
          for (int i = 0; i < FM; ++i) aq[q][i] = aA[ks & 1][q][i];
        if (ks + 2 < KSC) rd(ks + 2, bvA[ks & 1], aA[ks & 1]);
        __builtin_amdgcn_sched_barrier(0);
        mma(bq, aq);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int ks = 0; ks < KSC; ++ks) te[ks] = tn[ks];
    if ((EBEN_T3_DBG & 2) == 0 && P.ncc > 1 && written + 1 < P.ncc && (written + 1) * KS_CC < (ch + 2) * KSC) {
      ++written;
      store_x(written);
      if (written + 1 < P.ncc) pending = written + 1;
    }
    // chunk ch + 1 has landed once at most the DMAs of the chunks issued after it (ch + 2 .. ch + DIST) are in flight; tile loads of
    // fetch_x issued meanwhile only make the count stricter.  lgkmcnt(0): this wave's tile writes (store_x) and table loads.
    {
      int ahead = nch - 2 - ch;
      ahead = ahead > DIST - 1 ? DIST - 1 : ahead;
      if (ahead <= 0) t3_wait_vm<0>();
      else if (DIST < 3 || ahead == 1) t3_wait_vm<WU>();
      else t3_wait_vm<2 * WU>();
    }
    if ((EBEN_T3_DBG & 4) == 0) __builtin_amdgcn_s_barrier();
  }
  if (nch > 0) {
#pragma unroll
    for (int h = 0; h < H2; ++h) mma(bvB[h], aB[h]);
  }
  T3_STAMP(5);

  const bool use_res = P.res != nullptr && (P.res_rows == 0 || b < P.res_rows);
  // ---- epilogue: 32x32 D tile: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) ----
  // Every operand of the fused stages (bias, residual, activation mask, accumulate target) is read for all 16 rows of a tile in ONE
  // batch of loads (clamped addresses, no per-value branch), then the 16 values are formed and stored: written value by value the
  // compiler waits for each load in turn -- ~32 exposed round trips per lane, several times the whole reduction of a short layer.
  const int t = t0 + wn * 32 + (lane & 31);
  if (t >= nt) return;
  if constexpr (BL) {
    // row quad r4 of accumulator tile i: channels m4 .. m4+3 of the group, m4 = m0 + 32 i + 8 r4 + 4 (lane >> 5), i.e. half (lane >> 5)
    // of bundle (m0 >> 3) + 4 i + r4 -- one 8-byte piece per lane, 512 contiguous bytes per wave and quad.
    // Addresses: the bundle row is block-uniform (a scalar base per quad), the lane's place in it -- column and half -- is ONE 32-bit
    // byte offset for the whole epilogue (a group's rows of one batch item span < 2^31 bytes); whether a quad exists (m < Mg) is
    // uniform too, Mg being a multiple of 8.  (The thin layers are bound by the vector-instruction issue: ~660 VALU instructions per
    // wave around 41 MFMAs, a third of them this epilogue's 64-bit per-lane address arithmetic and per-lane predicates.)
    const int hb = lane >> 5;
    const bool pr = P.pr_S > 0;   // phases as rows (uniform)
    const unsigned loff = (((unsigned)t * (unsigned)(pr ? P.pr_S : P.OS) + (unsigned)oo) * 2u + (unsigned)hb) * 8u;   // bytes inside a bundle row
    const long long Lrow = (long long)(pr ? P.pr_Ly : P.Ly) * 16;                                          // bytes per bundle row
    const long long tile0 = pr ? (long long)(g * P.pr_cbg) * Lrow : ((long long)((g * P.Mg + m0) >> 3)) * Lrow;   // this tile's first bundle row
    // byte offset of logical bundle q of this tile from tile0 (uniform): its bundle row, in PR mode plus its phase's 16 bytes; prq: the
    // lane's position t pr_S + phase exists (always, unless the physical length is not a multiple of pr_S)
    auto qrow = [&](int q) -> long long {
      if (!pr) return (long long)q * Lrow;
      const int lb = (m0 >> 3) + q, phs = lb / P.pr_cbg;
      return (long long)(lb - phs * P.pr_cbg) * Lrow + (long long)phs * 16;
    };
    auto qlive = [&](int q) -> bool {
      if (!pr) return true;
      return t * P.pr_S + ((m0 >> 3) + q) / P.pr_cbg < P.pr_Ly;
    };
    const char* ehb = reinterpret_cast<const char*>(P.eh) + (long long)eb * P.CBy * Lrow + tile0;
    const char* elb = reinterpret_cast<const char*>(P.el) + (long long)eb * P.CBy * Lrow + tile0;
    const char* rhb = reinterpret_cast<const char*>(P.eh) + (long long)(b + P.bl_ref_off) * P.CBy * Lrow + tile0;
    const char* rlb = reinterpret_cast<const char*>(P.el) + (long long)(b + P.bl_ref_off) * P.CBy * Lrow + tile0;
    char* yhb = reinterpret_cast<char*>(P.yh) + (long long)b * P.CBy * Lrow + tile0;
    char* ylb = reinterpret_cast<char*>(P.yl) + (long long)b * P.CBy * Lrow + tile0;
    const bool masked = P.eh != nullptr && !(EBEN_T3_DBG & 32);
    // feature-matching rows with a code plane (bl_edge.hip, bl_fm_code): one byte per element at half the hi plane's byte offsets
    const bool fmc = fmr && masked && P.ec != nullptr;
    const char* ecb = reinterpret_cast<const char*>(P.ec) + (((long long)eb * P.CBy * Lrow + tile0) >> 1);
    auto unpack = [](uint2 w, float (&f)[4]) {
      f[0] = __builtin_bit_cast(float, w.x << 16); f[1] = __builtin_bit_cast(float, w.x & 0xffff0000u);
      f[2] = __builtin_bit_cast(float, w.y << 16); f[3] = __builtin_bit_cast(float, w.y & 0xffff0000u);
    };
    auto ld2 = [&](const char* base, long long row) { return *reinterpret_cast<const uint2*>(base + row + loff); };
    const int quads = (P.Mg - m0 + 7) >> 3;   // bundle rows of this tile that exist (uniform)
    if (pr && P.pr_order == 1) {
      // ---- phases as rows, rows ordered (channel bundle, phase, channel in bundle), stride 4 (tap4_kernel.h has the same form): accumulator
      // tile i IS bundle cb0 + i of the group at the four phases of this lane's column -- row quad r4 = phase r4 -- i.e. the 64 contiguous
      // bytes of positions 4 t .. 4 t + 3.  One v_permlane32_swap per dword turns the lane's four half units into two whole units: lane
      // (t, 0) positions 4 t, 4 t + 1, lane (t, 1) positions 4 t + 2, 4 t + 3 -- 32 contiguous bytes per lane, where order 0 reads its mask
      // and writes its result as 8-byte pieces 64 bytes apart.  [MI355X] the mask / feature-matching loads of the order-0 form are 36-43 %
      // of a MelGAN L1 / L2 input-gradient launch (253 / 233 -> 143 / 148 us without them).
      // Stride 2: a 32-row tile is TWO bundles at the two phases; the swap leaves lane (t, 0) with bundle 2 i at positions 2 t, 2 t + 1
      // and lane (t, 1) with bundle 2 i + 1 at the same positions.
      const bool s4 = P.pr_S == 4;
      const int tile0i = m0 >> 5;
      const long long LrowP = (long long)P.pr_Ly * 16;
      const long long tileP = (long long)(g * P.pr_cbg) * LrowP;
      const char* ehp = reinterpret_cast<const char*>(P.eh) + (long long)eb * P.CBy * LrowP + tileP;
      const char* elp = reinterpret_cast<const char*>(P.el) + (long long)eb * P.CBy * LrowP + tileP;
      const char* rhp = reinterpret_cast<const char*>(P.eh) + (long long)(b + P.bl_ref_off) * P.CBy * LrowP + tileP;
      const char* rlp = reinterpret_cast<const char*>(P.el) + (long long)(b + P.bl_ref_off) * P.CBy * LrowP + tileP;
      char* yhp = reinterpret_cast<char*>(P.yh) + (long long)b * P.CBy * LrowP + tileP;
      char* ylp = reinterpret_cast<char*>(P.yl) + (long long)b * P.CBy * LrowP + tileP;
      const int pos0 = s4 ? 4 * t + 2 * hb : 2 * t;
      const bool lv0 = pos0 < P.pr_Ly, lv1 = pos0 + 1 < P.pr_Ly;
      const unsigned poff = (unsigned)(lv0 ? pos0 : 0) * 16u;
      auto bund = [&](int i) { return s4 ? tile0i + i : 2 * (tile0i + i) + hb; };   // this lane's physical bundle (within the group) of tile i
      auto swap32 = [](unsigned& x, unsigned& y) {
        const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
        x = r[0]; y = r[1];
      };
      // memory layout (two whole units per lane) <-> MFMA layout (four half units per lane)
      auto to_halves = [&](const u32x4 (&U)[2], uint2 (&H)[4]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          unsigned x0 = U[j][0], x1 = U[j][1], y0 = U[j][2], y1 = U[j][3];
          swap32(x0, y0); swap32(x1, y1);
          H[j].x = x0; H[j].y = x1; H[2 + j].x = y0; H[2 + j].y = y1;
        }
      };
      auto to_units = [&](const uint2 (&H)[4], u32x4 (&U)[2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          unsigned x0 = H[j].x, x1 = H[j].y, y0 = H[2 + j].x, y1 = H[2 + j].y;
          swap32(x0, y0); swap32(x1, y1);
          U[j] = u32x4{x0, x1, y0, y1};
        }
      };
      auto ldu = [&](const char* base, int i, u32x4 (&U)[2]) {
        const int bd = bund(i);
        const char* q = base + (long long)(bd < P.pr_cbg ? bd : 0) * LrowP + poff;   // a bundle past the group's last (stride 2, odd count): re-read, never stored
        U[0] = *reinterpret_cast<const u32x4*>(q);
        U[1] = *reinterpret_cast<const u32x4*>(q + (lv1 ? 16 : 0));
      };
      auto stu = [&](int i, const uint2 (&H)[4], char* base) {
        u32x4 U[2];
        to_units(H, U);
        const int bd = bund(i);
        char* q = base + (long long)bd * LrowP + poff;
        if ((EBEN_T3_DBG & 128) && U[0][0] != 0x12345u) return;
        if (bd >= P.pr_cbg) return;
        if (lv0) *reinterpret_cast<u32x4*>(q) = U[0];
        if (lv1) *reinterpret_cast<u32x4*>(q + 16) = U[1];
      };
      const int tiles = (P.Mg - m0 + 31) >> 5;   // 32-row tiles of this block that exist (uniform; Mg is a multiple of 32 here)
      const char* ecp = reinterpret_cast<const char*>(P.ec) + (((long long)eb * P.CBy * LrowP + tileP) >> 1);
      u32x4 AU[FM][2];
      if (masked && (!fmr || fmc)) {
#pragma unroll
        for (int i = 0; i < FM; ++i) ldu(ehp, i < tiles ? i : 0, AU[i]);
      }
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        if (i >= tiles) continue;
        uint2 AH[4], AL[4], RH[4], RL[4], OH[4], OL[4];
        unsigned CW[4] = {0u, 0u, 0u, 0u};
        if (masked && (!fmr || fmc)) {
          to_halves(AU[i], AH);
          if (fmc) {
            // the two whole code units of this lane (positions pos0, pos0 + 1: 16 contiguous bytes) -> the four half units of its row quads
            const int bd = bund(i);
            const char* q = ecp + (((long long)(bd < P.pr_cbg ? bd : 0) * LrowP + poff) >> 1);
            const uint2 c0 = *reinterpret_cast<const uint2*>(q);
            const uint2 c1 = *reinterpret_cast<const uint2*>(q + (lv1 ? 8 : 0));
            unsigned x0 = c0.x, y0 = c0.y, x1 = c1.x, y1 = c1.y;
            swap32(x0, y0); swap32(x1, y1);
            CW[0] = x0; CW[2] = y0; CW[1] = x1; CW[3] = y1;
          }
        } else if (masked) {
          u32x4 A2[2], L2[2], RH2[2], RL2[2];
          ldu(ehp, i, A2); ldu(elp, i, L2); ldu(rhp, i, RH2); ldu(rlp, i, RL2);
          to_halves(A2, AH); to_halves(L2, AL); to_halves(RH2, RH); to_halves(RL2, RL);
        }
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          float v[4], a0[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[i][4 * r4 + e];
          if (masked) {
            unpack(AH[r4], a0);
            if (fmc) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const unsigned c = CW[r4] >> (8 * e);
                v[e] += fk1 * (float)((int)(c & 3u) - 1) - fk2 * (float)((int)((c >> 2) & 3u) - 1);
              }
            } else if (fmr) {
              float a1[4], r0[4], r1[4];
              unpack(AL[r4], a1); unpack(RH[r4], r0); unpack(RL[r4], r1);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float av = a0[e] + a1[e], dv = av - (r0[e] + r1[e]);
                v[e] += fk1 * (float)((dv > 0.f) - (dv < 0.f)) - fk2 * (float)((av > 0.f) - (av < 0.f));
              }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= dlrelu(a0[e], P.emask_slope);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = lrelu(v[e], P.out_slope);
          }
          OH[r4].x = pack_bf16(v[0], v[1]); OH[r4].y = pack_bf16(v[2], v[3]);
          float hf[4];
          unpack(OH[r4], hf);
          OL[r4].x = pack_bf16(v[0] - hf[0], v[1] - hf[1]); OL[r4].y = pack_bf16(v[2] - hf[2], v[3] - hf[3]);
        }
        stu(i, OH, yhp);
        if (P.yl) stu(i, OL, ylp);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      uint2 ah[4], al[4], rh[4], rl[4];
      unsigned cw[4];
      float bz[4][4];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int q = 4 * i + r4;
        long long row = qrow(q < quads ? q : 0);   // uniform; a missing quad re-reads the tile's first row
        if (pr && !qlive(q < quads ? q : 0)) row = -(long long)loff;   // (per lane) a position past the row's end re-reads the tile's first unit
        {
          const float4 bq = *reinterpret_cast<const float4*>(Bs + i * 32 + 8 * r4 + 4 * hb);
          bz[r4][0] = bq.x; bz[r4][1] = bq.y; bz[r4][2] = bq.z; bz[r4][3] = bq.w;
        }
        if (masked) {
          if constexpr (PREF) ah[r4] = pr ? ld2(ehb, row) : pah[i][r4];
          else ah[r4] = ld2(ehb, row);
          if (fmc) cw[r4] = *reinterpret_cast<const unsigned*>(ecb + ((row + (long long)loff) >> 1));   // the half unit's four code bytes
          else if (fmr) { al[r4] = ld2(elb, row); rh[r4] = ld2(rhb, row); rl[r4] = ld2(rlb, row); }
        }
      }
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int q = 4 * i + r4;
        if (q >= quads) continue;   // uniform
        const long long row = qrow(q);
        if (pr && !qlive(q)) continue;   // per lane
        float v[4], a0[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][4 * r4 + e] + bz[r4][e];
        if (masked) {
          unpack(ah[r4], a0);
          if (fmc) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const unsigned c = cw[r4] >> (8 * e);
              v[e] += fk1 * (float)((int)(c & 3u) - 1) - fk2 * (float)((int)((c >> 2) & 3u) - 1);
            }
          } else if (fmr) {
            float a1[4], r0[4], r1[4];
            unpack(al[r4], a1); unpack(rh[r4], r0); unpack(rl[r4], r1);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float av = a0[e] + a1[e], dv = av - (r0[e] + r1[e]);
              v[e] += fk1 * (float)((dv > 0.f) - (dv < 0.f)) - fk2 * (float)((av > 0.f) - (av < 0.f));
            }
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= dlrelu(a0[e], P.emask_slope);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = lrelu(v[e], P.out_slope);
        }
        uint2 h;
        h.x = pack_bf16(v[0], v[1]); h.y = pack_bf16(v[2], v[3]);
        if (!((EBEN_T3_DBG & 128) && h.x != 0x12345u)) *reinterpret_cast<uint2*>(yhb + row + loff) = h;
        if (P.yl) {
          float hf[4];
          unpack(h, hf);
          uint2 l;
          l.x = pack_bf16(v[0] - hf[0], v[1] - hf[1]); l.y = pack_bf16(v[2] - hf[2], v[3] - hf[3]);
          if (!((EBEN_T3_DBG & 128) && l.x != 0x12345u)) *reinterpret_cast<uint2*>(ylb + row + loff) = l;
        }
      }
    }
    T3_STAMP(6);
#if EBEN_T3_DBG & 512
    __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0): the stores have left
    T3_STAMP(7);
#endif
    return;
  }
  // addresses: a block-uniform 64-bit base per tensor (scalar registers) + a 32-bit per-lane element offset (one group's rows of one
  // batch item span < 2^31 elements), so that every access is the scalar-base form and no 64-bit vector arithmetic is left
  const unsigned col = (unsigned)t * (unsigned)P.OS + (unsigned)oo;
  const unsigned ycol = (EBEN_T3_DBG & 16) ? (unsigned)ph * (unsigned)nt + (unsigned)t : col;   // 16: phase-major (coalesced) stores, scratch ablation
  const long long ubase = ((long long)b * P.Cy + (long long)g * P.Mg) * P.Ly;
  float* __restrict__ yb = P.y + ubase;
  const float* __restrict__ rb = P.res + ubase;
  const float* __restrict__ eb_ = P.emask + ((long long)eb * P.Cy + (long long)g * P.Mg) * P.Ly;
  const float* __restrict__ bb = P.bias + (long long)g * P.Mg;
  const int mlane = m0 + 4 * (lane >> 5);
  const int mlast = P.Mg - 1;
  const bool plain = !use_res && P.emask == nullptr && !P.accumulate;
  const bool lin = P.out_slope == 1.f && P.res_slope == 1.f;   // the engine's input-gradient launches: no activation arithmetic
  // feature-matching rows: the addend is formed from the two embeddings (mask = enhanced rows, res = reference rows) instead of read
  const bool fm = use_res && P.fm_sums != nullptr;
  float fc1 = 0.f, fc2 = 0.f;
  if (fm) { const float s1 = P.fm_sums[0], s2 = P.fm_sums[1]; fc1 = P.fm_gs / s2; fc2 = P.fm_gs * s1 / (s2 * s2); }
  if (plain) {
    // the forward's form: bias + activation only
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      float bz[16];
      unsigned off[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mlane + i * 32 + (r & 3) + 8 * (r >> 2);
        const int mc = m < mlast ? m : mlast;
        off[r] = (unsigned)mc * (unsigned)P.Ly + ycol;
        bz[r] = P.bias ? bb[mc] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mlane + i * 32 + (r & 3) + 8 * (r >> 2);
        if (m < P.Mg) yb[off[r]] = lrelu(acc[i][r] + bz[r], P.out_slope);
      }
    }
    return;
  }
#pragma unroll
  for (int ih = 0; ih < 2 * FM; ++ih) {   // eight rows (half an accumulator tile) per batch of loads: registers stay below the main loop's
    const int i = ih >> 1, r0 = (ih & 1) * 8;
    float bz[8], rz[8], ez[8], az[8];
    unsigned off[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int m = mlane + i * 32 + (r & 3) + 8 * ((r0 + r) >> 2);
      const int mc = m < mlast ? m : mlast;   // rows beyond the group re-read its last row and are not stored
      off[r] = (unsigned)mc * (unsigned)P.Ly;
      bz[r] = P.bias ? bb[mc] : 0.f;
    }
    if (use_res) {
#pragma unroll
      for (int r = 0; r < 8; ++r) rz[r] = rb[off[r] + ycol];
    }
    if (P.emask) {
#pragma unroll
      for (int r = 0; r < 8; ++r) ez[r] = eb_[off[r] + col];
    }
    if (P.accumulate) {
#pragma unroll
      for (int r = 0; r < 8; ++r) az[r] = yb[off[r] + ycol];
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int m = mlane + i * 32 + (r & 3) + 8 * ((r0 + r) >> 2);
      float v = acc[i][r0 + r] + bz[r];
      if (fm) {
        const float av = ez[r], dv = av - rz[r];
        v += fc1 * (float)((dv > 0.f) - (dv < 0.f)) - fc2 * (float)((av > 0.f) - (av < 0.f));
      } else if (!lin) {
        v = lrelu(v, P.out_slope);
        if (use_res) v += lrelu(rz[r], P.res_slope);
      } else if (use_res) {
        v += rz[r];
      }
      if (P.emask) v *= dlrelu(ez[r], P.emask_slope);
      if (P.accumulate) v += az[r];
      if (m < P.Mg) yb[off[r] + ycol] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------

static int gcd3(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

static int env_int3(const char* name, int dflt) {
  const char* s = getenv(name);
  return s ? atoi(s) : dflt;
}

// tap4_kernel (bigtap.hip) for the long reductions of the bundle layout: whole weight panels (up to 256 rows) per block, four waves with
// TM x TN MFMA tiles each, the input of one channel chunk (CI_T channels, all stride phases, BN columns + halo) double-buffered, the
// weights in a ring of RING chunks of KSC k-steps.  Needs: groups of whole 16-channel k-steps, an even number of channel chunks (the
// k-step table alternates two input buffers and a block runs tile after tile), channel chunks of at least (RING + 1) weight chunks (the
// DMA of an input tile is covered by the wait for the weight chunk issued after it).
static bool plan_big(const Canon& c, int dir, int Jmin, Tap3Plan* p) {
  static const int enabled = env_int3("EBEN_BIG", 1);
  // The thin layers stay with tap3_kernel.  [MI355X, 64 / 128 rows] measured and NOT kept, twice: (i) streamed weights, all channels in
  // one input tile alternating between two buffers: PQMF-band L1-L4 forward 0.040 -> 0.041-0.048 ms, their phase-scatter input
  // gradients 0.053-0.087 -> 0.093-0.126, MelGAN L1 / L2 phases-as-rows 0.232 / 0.212 -> 0.223 / 0.220; (ii) the whole weight image of
  // a panel RESIDENT in LDS (25-90 KB, loaded once per block instead of once per tile, one barrier per tile): MelGAN L1 / L2 forward
  // 0.067 / 0.067 -> 0.072 / 0.076, phases-as-rows 0.233 / 0.211 -> 0.214 / 0.218, PQMF-band L1-L4 input gradients 0.052-0.087 ->
  // 0.076-0.124.  These launches are not streams waiting for weight bytes: padded to 64 rows x whole k-steps they carry 2-4x their
  // MFMAs, and their epilogue (mask and feature-matching operands in, 8-byte pieces out) is what one consumer wave per SIMD cannot
  // hide -- tap3's eight to twenty waves per CU overlap it across blocks.
  static const int min_ks = env_int3("EBEN_BIG_MIN_KS", 100);     // MFMA k-steps of one output tile (x 3 for hi + lo operands)
  // (input gradients from 50: the PQMF-band L6 -- 768 -> 768, 5 taps, 60 k-steps, 192-row tiles -- 0.083 -> 0.071 ms now that the rows with
  // a feature-matching term read the code plane; it lost here while they read four operands per value)
  static const int min_ks_dx = env_int3("EBEN_BIG_MIN_KS_DX", 50);
  // [MI355X] the phase-scatter input gradients of the stride-4 layers (one output phase per tile: 8-byte stores 64 bytes apart, the mask
  // read the same way) lose here against tap3's many small blocks -- MelGAN L3 / L4 0.416 / 0.402 -> 0.506 / 0.444 ms: the epilogue's
  // partial lines are what a block per CU cannot hide; they run phases-as-rows (eben_bl_conv1d_bwd_dx_pr) or stay with tap3
  // [MI355X] the input gradients' epilogue is what tap4 hides worst (one consumer wave per SIMD): without its mask / feature-matching loads
  // MelGAN L3 / L4 / L5 dX run 0.376 / 0.375 / 0.149 -> 0.297 / 0.286 / 0.136 ms (scratch build EBEN_T4_DBG=32).  Measured and NOT kept: the
  // mask tile (hi plane under the output tile, 64 KB) staged through LDS by the producers during the reduction, with a ring of two
  // weight slots and 16-32-channel input tiles to make room -- 0.382 / 0.372 / 0.151 -> 0.411 / 0.394 / 0.157: the rows with a
  // feature-matching term (a quarter, four operands per value, one load batch per 32-row tile) are the larger part of the exposed
  // latency, and the smaller ring + 4.5 % more LDS-DMA bytes cost more than the masked rows gain.
  static const int strided_dx = env_int3("EBEN_BIG_STRIDED_DX", 0);   // largest output stride taken in phase-scatter form
  if (!enabled || !c.bl || c.reflect) return false;
  if (p->dense) return false;
  if (dir == 1 && p->OS > 1 && p->OS > strided_dx) return false;
  if (p->npw != p->npx || p->npw > 2 || p->nph > 8 || (p->Cg & 15) || (p->Mg & 63)) return false;
  const long long ks_total = (long long)(p->Cg / 16) * p->J * (p->npw == 2 ? 3 : 1);
  if (ks_total < (dir == 0 ? min_ks : min_ks_dx)) return false;
  // whole row tiles of 256, 192 or 128 rows on a 2 x 2 wave grid, 128 columns
  const int BM = p->Mg % 256 == 0 ? 256 : p->Mg % 192 == 0 ? 192 : 128;
  if (p->Mg % BM) return false;
  const int WM = 2, WN = 2, TM = BM / 64;
  // ([MI355X] a ring of two slots -- 105 KB instead of 137 KB per block, room for a small block of another stream beside it: MelGAN L4
  // forward 0.141 -> 0.145 ms alone, 0.15-0.18 -> 0.19 in the step, step 9.73 -> 9.81 ms: three it stays)
  const int KSC = p->npw == 1 ? 4 : 2, RING = 3;
  const int adstep = p->dstep >= 0 ? p->dstep : -p->dstep;
  const int maxd = ((p->J - 1) * adstep) / p->S + 1;
  for (int TN = 2; TN == 2; --TN) {
    p->BM = BM; p->FM = WM * TM; p->BN = WN * TN * 32;
    p->KSC = KSC;
    p->WCHU = KSC * p->npw * p->FM * 64;
    p->nmt = ceil_div(p->Mg, p->BM);
    p->ntt = ceil_div(p->nt, p->BN);
    p->PLEN = p->BN + maxd + 1;
    p->CSTRIDE = p->S * p->PLEN;
    const int wbytes = RING * p->WCHU * 16;
    int best = 0, best_ncc = 0;
    auto fits = [&](int ct, int ncc) {
      const int ppt = ceil_div((ct / 8) * p->CSTRIDE, 64);
      if (ppt < 4 || ceil_div(ppt, 4) > 12) return false;
      return (size_t)wbytes + (size_t)2 * p->npw * ppt * 64 * 16 + 2048 <= 160 * 1024;
    };
    // an even number of channel chunks (the input tile alternates between two buffers) of at least RING + 1 weight chunks each
    for (int ct = 16; !best && ct <= p->Cg / 2; ct += 16) {
      if (p->Cg % ct) continue;
      const int ncc = p->Cg / ct;
      if (ncc & 1) continue;
      if (Jmin * (ct / 16) < (RING + 1) * KSC) continue;
      if (!fits(ct, ncc)) continue;
      best = ct; best_ncc = ncc;
    }
    if (!best) continue;
    p->CI_T = best; p->CI_B = best / 8; p->CP = best / 16; p->ncc = best_ncc; p->nxbuf = 2; p->XRB = 0;
    p->PPT = ceil_div(p->CI_B * p->CSTRIDE, 64);
    p->XT = p->PPT * 64;
    p->xbuf_stride = p->npw * p->XT;
    const int KSmax = p->ncc * p->J * p->CP;
    p->NCH = ceil_div(KSmax, KSC);
    if (p->NCH < RING) continue;
    p->tab_phase = p->NCH * KSC;
    p->w_tile = (long long)p->NCH * p->WCHU;
    p->w_phase = p->w_tile * p->nmt * p->G;
    p->tab_off_floats = p->w_phase * p->nph * 4;
    p->packed_floats = (size_t)p->tab_off_floats + (size_t)p->tab_phase * p->nph;
    p->lds_bytes = (size_t)wbytes + (size_t)2 * p->npw * p->XT * 16 + 2048;
    p->WM = WM; p->WN = WN; p->TM = TM; p->TN = TN; p->RING = RING;
    p->big = 1;
    p->ok = 1;
    return true;
  }
  return false;
}

static void make_plan3(const Canon& c, int dir, Tap3Plan* p) {
  p->ok = 0;
  p->mode = dir;
  p->G = c.g;
  p->npw = c.np;
  p->npx = c.np > 1 ? c.np : (c.xsplit_dir == dir ? 2 : 1);
  const int ub = 16 * p->npx;          // LDS bytes per staged bundle position
  const int spare = p->npx > 1 ? 16 * p->npx : 0;   // one spare unit per piece region
  if (dir == 0) {
    p->Cg = c.Cin / c.g; p->Mg = c.Cout / c.g;
    p->S = c.s; p->OS = 1; p->dstep = c.d; p->kstep = 1; p->nph = 1; p->J = c.k;
    p->Lx = c.Lin; p->Ly = c.Lout; p->Cx = c.Cin; p->Cy = c.Cout;
    p->off0 = -c.pl; p->nt = c.Lout; p->ps_pad = 0;
  } else {
    p->Cg = c.Cout / c.g; p->Mg = c.Cin / c.g;
    p->S = 1; p->OS = c.s;
    p->kstep = c.s / gcd3(c.s, c.d);
    p->dstep = -(c.d * p->kstep) / c.s;
    p->nph = c.s;
    p->J = ceil_div(c.k, p->kstep);
    p->Lx = c.Lout; p->Cx = c.Cout; p->Cy = c.Cin;
    p->Ly = c.reflect ? c.Lin + c.pl + c.pr : c.Lin;
    p->ps_pad = c.reflect ? 0 : c.pl;
    p->off0 = 0; p->nt = ceil_div(p->Ly, c.s);
  }
  // A group of <= 8 input channels fills at most half of a 16-channel k-step (forward) or 8 of a tile's 32 rows (input
  // gradient), and one block per (group, tile) is mostly fixed cost: such layers (MelGAN L1: 4 channels per group, PQMF-disc
  // L1: 6) run as ONE group over all channels against a block-diagonal weight image (zeros across groups) -- the same number
  // of MFMAs or fewer, the input tile staged once instead of once per group, a quarter of the blocks.
  // [MI355X] at 128 items: MelGAN L1 forward 0.34 -> 0.20 ms, input gradient 0.72 -> 0.38; PQMF-disc L1 0.109 -> 0.069 and
  // 0.172 -> 0.071.  At 12 channels per group (PQMF-disc L2) the forward loses (0.064 -> 0.076: 2.2x the MFMAs) and the
  // input gradient still wins (0.116 -> 0.092: 8 k-steps per block become 24).
  static const int dense_max_c = env_int3("EBEN_TAP3_DENSE_MAX_C", 8);
  static const int dense_max_c_dx = env_int3("EBEN_TAP3_DENSE_MAX_C_DX", 12);
  p->dense = 0;
  // bundle layout: a group's channels must start on a bundle (8 channels) in both operands -- the layers with 4 / 6 / 12 channels per
  // group (MelGAN L1, PQMF-band L1 / L2) run dense in BOTH directions there (L2's forward: 2.2x the MFMAs of a 12 us launch)
  if (c.g > 1 && c.Cin / c.g >= 4 && c.Cin / c.g <= (dir == 0 && !c.bl ? dense_max_c : dense_max_c_dx) && c.Cin <= 64 && c.Cout <= 128) {
    p->dense = 1; p->G = 1; p->Cg *= c.g; p->Mg *= c.g;
  }
  if (c.bl && ((p->Cg & 7) || (p->Mg & 7) || c.np > 2 || c.xsplit_dir >= 0 || c.reflect)) return;
  static const int enabled = env_int3("EBEN_TAP3", 1);
  static const int min_m = env_int3("EBEN_TAP3_MIN_M", 4);
  static const int min_c = env_int3("EBEN_TAP3_MIN_C", 4);
  static const int min_k = env_int3("EBEN_TAP3_MIN_K", 32);   // 128 -> 32: 24.85 -> 24.5 ms/step (the generator's 32-channel layers)
  // the layers below these sizes are staging-bound either way; measured faster here than on the direct kernel from
  // 4 channels / 4 rows per group up (MelGAN L1 forward 0.23 -> 0.15 ms, its input gradient 0.47 -> 0.34 ms), given a
  // reduction of at least two weight chunks
  if (!enabled || p->Cg < min_c || p->Mg < min_m || p->nph > 64 || (long long)round_up(p->Cg, 16) * p->J < min_k) return;
  const int Jmin = dir == 0 ? p->J : (c.k / p->kstep > 0 ? c.k / p->kstep : 1);
  p->big = 0; p->WM = p->WN = p->TM = p->TN = p->RING = p->XT = p->PPT = 0;
  if (plan_big(c, dir, Jmin, p)) return;

  const int cand[4] = {128, 96, 64, 32};
  const double eff[4] = {1.0, 0.9, 0.8, 0.5};
  int best = 0;
  double best_score = -1.0;
  for (int i = 0; i < 4; ++i) {
    const double score = eff[i] * p->Mg / round_up(p->Mg, cand[i]);
    if (score > best_score + 1e-9) { best_score = score; best = cand[i]; }
  }
  {
    const long long cols = (long long)ceil_div(p->nt, 128) * c.B * p->nph * p->G;
    while (best > 32 && cols * ceil_div(p->Mg, best) < 256) {
      const int smaller = best == 128 ? 64 : 32;
      if (round_up(p->Mg, smaller) > round_up(p->Mg, best)) break;
      best = smaller;
    }
  }
  static const int force_bm = env_int3("EBEN_TAP3_BM", 0);
  if (force_bm == 32 || force_bm == 64 || force_bm == 96 || force_bm == 128) best = force_bm;
  p->BM = best; p->FM = best / 32; p->BN = 128;
  p->KSC = t3_ksc(p->npw, p->FM);
  p->WCHU = p->KSC * p->npw * p->FM * 64;
  p->nmt = ceil_div(p->Mg, p->BM);
  p->ntt = ceil_div(p->nt, p->BN);

  const int adstep = p->dstep >= 0 ? p->dstep : -p->dstep;
  const int maxd = ((p->J - 1) * adstep) / p->S + 1;
  p->PLEN = p->BN + maxd + 1;
  p->CSTRIDE = p->S * p->PLEN;
  const int span = (p->BN - 1) * p->S + (p->J - 1) * adstep + 1;
  if (span > 0xffff) return;
  const int NT = 256;
  // [MI355X] alone on the device the kernel is fastest with the largest tile that leaves two blocks per CU (78 KB: whole discriminator
  // forward at 64 items 1.81 ms against 2.03 at 52 KB); inside the step, where three or four streams share the CUs, smaller blocks
  // co-reside with the other streams' kernels: 19.1 / 18.9 / 18.8 ms per step at 78 / 52 / 39 KB, 19.2 at 26 KB
  static const int lds_budget1 = env_int3("EBEN_TAP3_LDS_KB", 48) * 1024;
  static const int lds_budget_split = env_int3("EBEN_TAP3_SPLIT_LDS_KB", 150) * 1024;   // split weights: one MFMA-bound block per CU
  static const int lds_budget_x3 = env_int3("EBEN_TAP3_X3_LDS_KB", 64) * 1024;
  const int wbytes = t3_ring(p->npw, p->FM, c.bl != 0) * p->WCHU * 16;
  const int Cg2 = round_up(p->Cg, 16);
  // input tiles inside `lds_budget` bytes per block (weights included); a three-buffer scheme may go up to `big`
  auto size_tiles = [&](int lds_budget, int big) -> bool {
    const int xbudget = lds_budget - wbytes - 16;
    p->XRB = 2;
    if ((long long)(Cg2 / 8) * p->CSTRIDE * ub + spare <= xbudget) {
      p->CI_T = Cg2; p->ncc = 1; p->nxbuf = 1;
      return true;
    }
    // hand-over rule of tapconv2.hip in weight chunks of KSC k-steps: the tile of the next channel chunk is
    // written one weight chunk before its first use into the buffer of the tile `nxbuf` chunks back
    for (int nbuf = 2; nbuf <= 3; ++nbuf) {
      for (int xrb : {2, 3, 5}) {
        int cap = (xrb * NT) / span * 8;          // channels
        const int cap_lds = ((nbuf == 2 ? xbudget : big - wbytes) - spare) / nbuf / (p->CSTRIDE * ub) * 8;
        if (cap > cap_lds) cap = cap_lds;
        cap -= cap % 16;
        if (cap < 16) continue;
        const int nchk = ceil_div(Cg2, cap);
        p->CI_T = round_up(ceil_div(p->Cg, nchk), 16);
        p->ncc = ceil_div(p->Cg, p->CI_T);
        p->nxbuf = nbuf;
        p->XRB = xrb;
        if (p->ncc > 1 && Jmin * (p->CI_T / 16) < (nbuf == 2 ? 2 * p->KSC : p->KSC)) continue;
        return true;
      }
    }
    return false;
  };
  bool sized;
  // long reductions (MelGAN L3-L5: >= EBEN_TAP3_BIG_KS k-steps per block) keep the two-blocks-per-CU tiles: they are the launches
  // that fill the device by themselves, and lose 40-50 % stand-alone on the small budget
  static const int big_ks = env_int3("EBEN_TAP3_BIG_KS", 128);   // [MI355X] neutral in the step (18.0-18.3 ms either way), MelGAN L4 forward alone 0.30 -> 0.20 ms
  static const int lds_budget_big = env_int3("EBEN_TAP3_BIG_LDS_KB", 78) * 1024;
  const long long ks_total = (long long)ceil_div(p->Cg, 16) * p->J;
  if (p->npw == 1) sized = size_tiles(ks_total >= big_ks ? lds_budget_big : lds_budget1, 110 * 1024) || size_tiles(lds_budget_big, 110 * 1024) ||
                           size_tiles(lds_budget_split, lds_budget_split);
  else if (p->npw == 2) sized = size_tiles(lds_budget_x3, 110 * 1024) || size_tiles(lds_budget_split, lds_budget_split);
  else sized = size_tiles(lds_budget_split, lds_budget_split);
  if (!sized) return;
  p->CI_B = p->CI_T / 8;
  p->CP = p->CI_T / 16;
  p->xbuf_stride = p->CI_B * p->CSTRIDE;
  const int KSmax = p->ncc * p->J * p->CP;
  p->NCH = ceil_div(KSmax, p->KSC);
  p->tab_phase = p->NCH * p->KSC;
  p->w_tile = (long long)p->NCH * p->WCHU;
  p->w_phase = p->w_tile * p->nmt * p->G;
  p->tab_off_floats = p->w_phase * p->nph * 4;
  p->packed_floats = (size_t)p->tab_off_floats + (size_t)p->tab_phase * p->nph;
  p->lds_bytes = (size_t)wbytes + ((size_t)p->nxbuf * p->CI_B * p->CSTRIDE * 16 + 16) * p->npx;   // + the spare unit of every piece
  if (c.bl) p->lds_bytes += (size_t)p->BM * 4;   // the block's bias rows
  if (p->lds_bytes > 160 * 1024) return;
  p->ok = 1;
}

struct Pack3Args {
  const float* w; const float* scale; float* wp;
  int G, Cg, Mg, nmt, BM, FM, WCHU, CI_T, CI_B, CP, ncc, NCH, nph, tab_phase;
  int mode, J0, off0, nt, dstep, OS, S, ps_pad, k, d, kstep, Ly;
  int Cin_g, Cout_g, PLEN, CSTRIDE, nxbuf, dense, NPW, KSC, xbuf_stride;
  long long w_tile, w_phase, wunits;
  // host-side arithmetic of the unit decomposition: ceil(2^32 / d) per divisor (0: d == 1), the tap geometry of every phase
  unsigned m_nmt, m_fm, m_npw, m_cp, m_coutg;
  float r_wphase, r_wtile, r_wchu;
  struct PGp { int J, k0; unsigned m_kscc; int pad; } pg[8];
};

// One thread = one 16-byte unit of the image (every unit in flight at once: the eight weights of a unit are eight scattered loads).
// The unit index is decomposed with 32-bit multiply-high divisions by host-side magic numbers (the image of the largest layer has
// 2^21 units; round 3's form used 64-bit divisions, ~150 instructions each, and searched the phase geometry per unit) and the per-phase
// tap geometry comes from a table in the arguments.
__device__ __forceinline__ unsigned p3_div(unsigned n, unsigned d, unsigned magic) {   // n / d for n * d < 2^32 (small operands)
  return magic ? __umulhi(n, magic) : n;                                                // magic 0: d == 1
}
__device__ __forceinline__ unsigned p3_divf(unsigned n, unsigned d, float rd) {          // n / d for n < 2^24: fp32 estimate + one correction
  unsigned q = (unsigned)((float)n * rd);
  const int r = (int)(n - q * d);
  return r < 0 ? q - 1u : ((unsigned)r >= d ? q + 1u : q);
}
__device__ __forceinline__ void pack3_body(const Pack3Args& P, unsigned bid, unsigned nblk) {
  const unsigned total = (unsigned)P.wunits + (unsigned)(P.tab_phase * P.nph);
  for (unsigned i = bid * 256u + threadIdx.x; i < total; i += nblk * 256u) {
    if (i < (unsigned)P.wunits) {
      unsigned r = i;
      const unsigned ph = p3_divf(r, (unsigned)P.w_phase, P.r_wphase); r -= ph * (unsigned)P.w_phase;
      const unsigned tile = p3_divf(r, (unsigned)P.w_tile, P.r_wtile); r -= tile * (unsigned)P.w_tile;
      const unsigned g = p3_div(tile, (unsigned)P.nmt, P.m_nmt), mt = tile - g * (unsigned)P.nmt;
      const unsigned ch = p3_divf(r, (unsigned)P.WCHU, P.r_wchu);
      const unsigned e = r - ch * (unsigned)P.WCHU;
      const unsigned e64 = e >> 6, lane = e & 63u;
      const unsigned kp = p3_div(e64, (unsigned)P.FM, P.m_fm), fm = e64 - kp * (unsigned)P.FM;       // kp = ks * NPW + piece
      const unsigned ks = p3_div(kp, (unsigned)P.NPW, P.m_npw), piece = kp - ks * (unsigned)P.NPW;
      int qJ, qk0;
      unsigned m_kscc;
      if (ph < 8u) { qJ = P.pg[ph].J; qk0 = P.pg[ph].k0; m_kscc = P.pg[ph].m_kscc; }
      else {   // strides beyond the table (none in EBEN)
        const PhaseGeom q = phase_geom(P.mode, (int)ph, P.J0, P.off0, P.nt, P.dstep, P.OS, P.ps_pad, P.k, P.d, P.kstep, P.Ly);
        qJ = q.J; qk0 = q.k0;
        const unsigned dd = (unsigned)(qJ * P.CP);
        m_kscc = dd <= 1u ? 0u : (unsigned)((0x100000000ull + dd - 1) / dd);
      }
      const int s = (int)(ch * (unsigned)P.KSC + ks);
      const int KS_CC = qJ * P.CP;
      float v[8];
      {
        // all eight weight loads are issued unconditionally (clamped index), the padding is a select afterwards
        const bool live = qJ > 0 && s < P.ncc * KS_CC;
        const unsigned sc = live ? (unsigned)s : 0u;
        const unsigned cc = KS_CC > 0 ? p3_div(sc, (unsigned)KS_CC, m_kscc) : 0u, rem = sc - cc * (unsigned)KS_CC;
        const unsigned j = p3_div(rem, (unsigned)P.CP, P.m_cp), cp = rem - j * (unsigned)P.CP;
        const int m = (int)(mt * (unsigned)P.BM + fm * 32u + (lane & 31u));
        const int kk = qk0 + (int)j * P.kstep;
        float sc8[8];
        bool ok[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int chan = (int)(cc * (unsigned)P.CI_T + 16u * cp + 8u * (lane >> 5)) + u;
          ok[u] = live && chan < P.Cg && m < P.Mg && (P.mode == 0 || kk < P.k);
          const int co = P.mode == 0 ? (int)g * P.Cout_g + m : (int)g * P.Cout_g + chan;   // mode 1: reduction channel = conv output channel
          // conv input channel within its group; dense form (g == 0, channels counted over all groups): zero across groups
          const int ci = (P.mode == 0 ? chan : m) - (P.dense ? (int)p3_div((unsigned)(co < 0 ? 0 : co), (unsigned)P.Cout_g, P.m_coutg) * P.Cin_g : 0);
          ok[u] = ok[u] && ci >= 0 && ci < P.Cin_g;
          const long long idx = ((long long)co * P.Cin_g + ci) * P.k + (P.mode == 0 ? (int)j : kk);
          v[u] = P.w[ok[u] ? idx : 0];
          sc8[u] = P.scale ? P.scale[ok[u] ? co : 0] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ok[u] ? v[u] * sc8[u] : 0.f;
      }
      u32x4 o;
      for (unsigned qq = 0;; ++qq) {   // piece `piece` of the split w = p0 + p1 + ..: p_q = bf16(w - p0 - .. - p(q-1))
        o[0] = pack_bf16(v[0], v[1]); o[1] = pack_bf16(v[2], v[3]); o[2] = pack_bf16(v[4], v[5]); o[3] = pack_bf16(v[6], v[7]);
        if (qq == piece) break;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          v[2 * u] -= __builtin_bit_cast(float, o[u] << 16);
          v[2 * u + 1] -= __builtin_bit_cast(float, o[u] & 0xffff0000u);
        }
      }
      reinterpret_cast<u32x4*>(P.wp)[i] = o;
    } else {
      const int r = (int)(i - (unsigned)P.wunits);
      const int ph = r / P.tab_phase;
      const int s = r - ph * P.tab_phase;
      const PhaseGeom q = phase_geom(P.mode, ph, P.J0, P.off0, P.nt, P.dstep, P.OS, P.ps_pad, P.k, P.d, P.kstep, P.Ly);
      const int KS_CC = q.J * P.CP;
      int o = 0;
      if (q.J > 0 && s < P.ncc * KS_CC) {
        const int cc = s / KS_CC, rem = s - cc * KS_CC;
        const int j = rem / P.CP, cp = rem - j * P.CP;
        const int rel = q.off0 + j * P.dstep - q.minoff;
        const int dd = rel / P.S, pp = rel - dd * P.S;
        o = (P.nxbuf > 1 ? (cc % P.nxbuf) * P.xbuf_stride : 0) + 2 * cp * P.CSTRIDE + pp * P.PLEN + dd;
      }
      reinterpret_cast<int*>(P.wp)[P.wunits * 4 + r] = o;
    }
  }
}

// ---- coalescing form for the big images (MelGAN L3-L5, PQMF-band L5 / L6: ~10^6 weights each, 4/5 of a step's packed bytes) ----------
// In pack3_body a unit's eight weights are eight loads k (gather-strided) or Cin_g k (phase-scatter) floats apart, and the lanes of a wave
// sit on 32 different rows: every load instruction touches 64 cache lines for 256 useful bytes, every line is asked for ~30 times
// ([MI355X] ~100 us per 10^7-weight image, 0.6 TB/s).  Here a block stages the sub-block of w its units come from -- 32 rows x 16
// reduction channels x all k taps, which is 16 k (mode 0) or 32 k (mode 1) CONTIGUOUS floats per row / channel -- in LDS with coalesced
// loads, then emits the units of every (phase, tap): LDS reads at an odd row stride (conflict-free), 1 KB contiguous per wave store.
// Item = (group, row tile, 32-row sub-tile fm, channel chunk cc, channel pair cp).  The padding k-steps behind the last real one of a
// (phase, tile) are zero-filled by the item that owns the last channel pair.  Not for the dense (block-diagonal) images.
__device__ __forceinline__ void pack3_tables(const Pack3Args& P);
__global__ __launch_bounds__(256) void pack3c_kernel(const Pack3Args P) {
  extern __shared__ float p3c[];
  if (blockIdx.x == gridDim.x - 1) pack3_tables(P);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k = P.k;
  const int NCP = P.ncc * P.CP;                       // 16-channel columns of the reduction
  int item = blockIdx.x;
  const int col = item % NCP; item /= NCP;
  const int fm = item % P.FM; item /= P.FM;
  const int mt = item % P.nmt;
  const int g = item / P.nmt;
  const int cc = col / P.CP, cp = col - cc * P.CP;
  const int m0 = mt * P.BM + fm * 32;                 // first row of the sub-tile (within the group)
  const int c0 = cc * P.CI_T + 16 * cp;               // first reduction channel (within the group)
  const int nrow = P.Mg - m0 < 32 ? P.Mg - m0 : 32;   // rows / channels that exist
  const int nch = P.Cg - c0 < 16 ? P.Cg - c0 : 16;
  // mode 0: LDS[row r][c k + j] = w[g Cout_g + m0 + r][c0 + c][j]        (row = conv output channel, 16 k contiguous floats)
  // mode 1: LDS[chan c][r k + j] = w[g Cout_g + c0 + c][m0 + r][j]        (row of the image = conv INPUT channel; 32 k contiguous floats)
  const int RS = (P.mode == 0 ? 16 : 32) * k | 1;
  if (nrow > 0 && nch > 0) {
    const int nlines = P.mode == 0 ? nrow : nch;
    const int run = (P.mode == 0 ? nch : nrow) * k;
    for (int ln = wave; ln < nlines; ln += 4) {
      const long long src = P.mode == 0 ? ((long long)(g * P.Cout_g + m0 + ln) * P.Cin_g + c0) * k
                                        : ((long long)(g * P.Cout_g + c0 + ln) * P.Cin_g + m0) * k;
      const float sc = P.scale ? P.scale[g * P.Cout_g + (P.mode == 0 ? m0 : c0) + ln] : 1.f;
      for (int i = lane; i < run; i += 64) p3c[ln * RS + i] = P.w[src + i] * sc;
    }
  }
  __syncthreads();
  const int r = lane & 31, h = lane >> 5;
  const bool row_ok = r < nrow;
  u32x4* img = reinterpret_cast<u32x4*>(P.wp);
  const long long tile = (long long)g * P.nmt + mt;
  const bool last_col = col == NCP - 1;
  for (int ph = 0; ph < P.nph; ++ph) {
    int qJ, qk0;
    if (ph < 8) { qJ = P.pg[ph].J; qk0 = P.pg[ph].k0; }
    else { const PhaseGeom q = phase_geom(P.mode, ph, P.J0, P.off0, P.nt, P.dstep, P.OS, P.ps_pad, P.k, P.d, P.kstep, P.Ly); qJ = q.J; qk0 = q.k0; }
    const int KS_CC = qJ * P.CP, KS_all = P.ncc * KS_CC;
    const long long base = (long long)ph * P.w_phase + tile * P.w_tile;
    for (int j = wave; j < qJ; j += 4) {
      const int tap = P.mode == 0 ? j : qk0 + j * P.kstep;
      const int s = cc * KS_CC + j * P.CP + cp;
      const int ch = s / P.KSC, ks = s - ch * P.KSC;
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = 8 * h + u;
        const bool ok = row_ok && c < nch && tap < k;
        const int idx = P.mode == 0 ? r * RS + c * k + tap : c * RS + r * k + tap;
        v[u] = ok ? p3c[idx] : 0.f;
      }
      for (int piece = 0; piece < P.NPW; ++piece) {
        u32x4 o;
        o[0] = pack_bf16(v[0], v[1]); o[1] = pack_bf16(v[2], v[3]); o[2] = pack_bf16(v[4], v[5]); o[3] = pack_bf16(v[6], v[7]);
        img[base + (long long)ch * P.WCHU + ((ks * P.NPW + piece) * P.FM + fm) * 64 + lane] = o;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          v[2 * u] -= __builtin_bit_cast(float, o[u] << 16);
          v[2 * u + 1] -= __builtin_bit_cast(float, o[u] & 0xffff0000u);
        }
      }
    }
    if (last_col) {   // the k-steps of the last chunk that stand for nothing
      for (int s = KS_all + wave; s < P.NCH * P.KSC; s += 4) {
        const int ch = s / P.KSC, ks = s - ch * P.KSC;
        for (int piece = 0; piece < P.NPW; ++piece)
          img[base + (long long)ch * P.WCHU + ((ks * P.NPW + piece) * P.FM + fm) * 64 + lane] = u32x4{0u, 0u, 0u, 0u};
      }
    }
  }
}

// the k-step tables (LDS offset of every k-step's B fragment): a few hundred integers per phase, written by the last block
__device__ __forceinline__ void pack3_tables(const Pack3Args& P) {
  const int tabs = P.tab_phase * P.nph;
  for (int r = threadIdx.x; r < tabs; r += 256) {
    const int ph = r / P.tab_phase;
    const int s = r - ph * P.tab_phase;
    const PhaseGeom q = phase_geom(P.mode, ph, P.J0, P.off0, P.nt, P.dstep, P.OS, P.ps_pad, P.k, P.d, P.kstep, P.Ly);
    const int KS_CC = q.J * P.CP;
    int o = 0;
    if (q.J > 0 && s < P.ncc * KS_CC) {
      const int cc = s / KS_CC, rem = s - cc * KS_CC;
      const int j = rem / P.CP, cp = rem - j * P.CP;
      const int rel = q.off0 + j * P.dstep - q.minoff;
      const int dd = rel / P.S, pp = rel - dd * P.S;
      o = (P.nxbuf > 1 ? (cc % P.nxbuf) * P.xbuf_stride : 0) + 2 * cp * P.CSTRIDE + pp * P.PLEN + dd;
    }
    reinterpret_cast<int*>(P.wp)[P.wunits * 4 + r] = o;
  }
}

__global__ __launch_bounds__(256) void pack3_kernel(const Pack3Args P) { pack3_body(P, blockIdx.x, gridDim.x); }

// several layers' images in one launch (eben_conv1d_pack_multi): block -> job by the prefix sums of the jobs' block counts
constexpr int PACK3_MULTI = 16;
struct Pack3Table {
  int n;
  unsigned first[PACK3_MULTI + 1];
  Pack3Args job[PACK3_MULTI];
};
__global__ __launch_bounds__(256) void pack3_multi_kernel(const Pack3Table T) {
  int j = 0;
#pragma unroll 1
  while (j + 1 < T.n && blockIdx.x >= T.first[j + 1]) ++j;
  pack3_body(T.job[j], blockIdx.x - T.first[j], T.first[j + 1] - T.first[j]);
}

template <int FM, int XRB, bool IM, int NPW = 1, int NPX = 1, bool BL = false>
static int launch3_im(const Tap3Args& a, int nblocks, size_t lds, hipStream_t st) {
  static LdsAttrOnce attr_once;
  auto kern = tap3_kernel<FM, XRB, IM, NPW, NPX, BL>;
  {
    const hipError_t e = lds_attr_once(attr_once, reinterpret_cast<const void*>(kern));
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(tap3)");
  }
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), lds, st, a);
  EBEN_CHECK_LAUNCH("tap3_kernel");
  return EBEN_OK;
}
template <int FM, int XRB>
static int launch3_cfg(const Tap3Args& a, int nblocks, size_t lds, int npw, int npx, hipStream_t st) {
  if (a.xh) {   // bundle layout
    if (a.in_mode || npw != npx || npw > 2) return fail(EBEN_EUNSUPPORTED, "tap3: bundle layout with %d / %d operand pieces", npw, npx);
    if (npw == 2) return launch3_im<FM, XRB, false, 2, 2, true>(a, nblocks, lds, st);
    return launch3_im<FM, XRB, false, 1, 1, true>(a, nblocks, lds, st);
  }
  if (npx > 1) {
    if (a.in_mode) return fail(EBEN_EUNSUPPORTED, "tap3: split operand with a mask on load");
    if (npw == 1) return launch3_im<FM, XRB, false, 1, 2>(a, nblocks, lds, st);
    if (npw == 2) return launch3_im<FM, XRB, false, 2, 2>(a, nblocks, lds, st);
    return launch3_im<FM, XRB, false, 3, 3>(a, nblocks, lds, st);
  }
  return a.in_mode ? launch3_im<FM, XRB, true>(a, nblocks, lds, st) : launch3_im<FM, XRB, false>(a, nblocks, lds, st);
}

int tap3_applicable(const Canon& c, int dir) {
  Tap3Plan p;
  make_plan3(c, dir, &p);
  return p.ok;
}

int tap3_is_big(const Canon& c, int dir) {   // served by tap4_kernel (bigtap.hip): kernel generation 6
  Tap3Plan p;
  make_plan3(c, dir, &p);
  return p.ok && p.big;
}

size_t tap3_packed_floats(const Canon& c, int dir) {
  Tap3Plan p;
  make_plan3(c, dir, &p);
  return p.ok ? p.packed_floats : 0;
}

static int tap3_pack_args(const Canon& c, int dir, const float* w, const float* scale, float* wp, Pack3Args* out, unsigned* blocks_out) {
  Tap3Plan p;
  make_plan3(c, dir, &p);
  if (!p.ok) return fail(EBEN_EUNSUPPORTED, "tap3_pack on a layer the bf16 kernel does not cover");
  Pack3Args a;
  a.w = w; a.scale = scale; a.wp = wp;
  a.G = p.G; a.Cg = p.Cg; a.Mg = p.Mg; a.nmt = p.nmt; a.BM = p.BM; a.FM = p.FM; a.WCHU = p.WCHU;
  a.CI_T = p.CI_T; a.CI_B = p.CI_B; a.CP = p.CP; a.ncc = p.ncc; a.NCH = p.NCH; a.nph = p.nph; a.tab_phase = p.tab_phase;
  a.mode = p.mode; a.J0 = p.J; a.off0 = p.off0; a.nt = p.nt; a.dstep = p.dstep; a.OS = p.OS; a.S = p.S; a.ps_pad = p.ps_pad;
  a.k = c.k; a.d = c.d; a.kstep = p.kstep; a.Ly = p.Ly;
  a.Cin_g = c.Cin / c.g; a.Cout_g = c.Cout / c.g; a.PLEN = p.PLEN; a.CSTRIDE = p.CSTRIDE; a.nxbuf = p.nxbuf; a.dense = p.dense;
  a.NPW = p.npw; a.KSC = p.KSC; a.xbuf_stride = p.xbuf_stride;
  a.w_tile = p.w_tile; a.w_phase = p.w_phase; a.wunits = p.w_phase * p.nph;
  long long blocks = (a.wunits + (long long)p.tab_phase * p.nph + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  {
    // p3_divf needs n < 2^24 (the largest image of EBEN, MelGAN L3 / L4, has 1.3 M units); p3_div: operands < 2^16 x 2^16
    const long long nmax = a.wunits + (long long)p.tab_phase * p.nph;
    if (nmax >= (1 << 24)) return fail(EBEN_EUNSUPPORTED, "tap3_pack: an image of %lld units is beyond the pack kernel's unit arithmetic", nmax);
    auto magic = [](long long d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned long long)d - 1) / (unsigned long long)d); };
    a.r_wphase = 1.0f / (float)a.w_phase; a.r_wtile = 1.0f / (float)a.w_tile; a.m_nmt = magic(p.nmt); a.r_wchu = 1.0f / (float)p.WCHU; a.m_fm = magic(p.FM);
    a.m_npw = magic(p.npw); a.m_cp = magic(p.CP); a.m_coutg = magic(c.Cout / c.g);
    for (int ph = 0; ph < 8; ++ph) {
      a.pg[ph].J = 0; a.pg[ph].k0 = 0; a.pg[ph].m_kscc = 0; a.pg[ph].pad = 0;
      if (ph >= p.nph) continue;
      const PhaseGeom q = phase_geom(p.mode, ph, p.J, p.off0, p.nt, p.dstep, p.OS, p.ps_pad, c.k, c.d, p.kstep, p.Ly);
      a.pg[ph].J = q.J; a.pg[ph].k0 = q.k0; a.pg[ph].m_kscc = magic((long long)q.J * p.CP);
    }
  }
  *out = a;
  *blocks_out = (unsigned)blocks;
  return EBEN_OK;
}

// big, not block-diagonal images go through the coalescing kernel (+ a tables launch); returns 1 when it took the job
static int tap3_pack_coalesced(const Pack3Args& a, hipStream_t st, int* rc) {
  static const long long min_w = getenv("EBEN_PACK3C_MIN") ? atoll(getenv("EBEN_PACK3C_MIN")) : 200000;   // weights; 0 = never
  *rc = EBEN_OK;
  const long long weights = (long long)a.G * a.Cout_g * a.Cin_g * a.k;
  const size_t lds = sizeof(float) * (size_t)(a.mode == 0 ? 32 : 16) * (size_t)(((a.mode == 0 ? 16 : 32) * a.k) | 1);
  if (min_w <= 0 || a.dense || weights < min_w || lds > 150 * 1024 || a.nph > 8) return 0;
  static LdsAttrOnce attr_once;
  {
    const hipError_t e = lds_attr_once(attr_once, reinterpret_cast<const void*>(pack3c_kernel));
    if (e != hipSuccess) { *rc = hip_fail(e, "hipFuncSetAttribute(pack3c)"); return 1; }
  }
  const long long items = (long long)a.G * a.nmt * a.FM * a.ncc * a.CP;
  hipLaunchKernelGGL(pack3c_kernel, dim3((unsigned)items), dim3(256), lds, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) *rc = hip_fail(e, "pack3c_kernel");
  return 1;
}

int tap3_pack(const Canon& c, int dir, const float* w, const float* scale, float* wp, hipStream_t st) {
  Pack3Args a;
  unsigned blocks;
  int rc = tap3_pack_args(c, dir, w, scale, wp, &a, &blocks);
  if (rc) return rc;
  if (tap3_pack_coalesced(a, st, &rc)) return rc;
  hipLaunchKernelGGL(pack3_kernel, dim3(blocks), dim3(256), 0, st, a);
  EBEN_CHECK_LAUNCH("pack3_kernel");
  return EBEN_OK;
}

// jobs[i] = (canonical conv, direction, weights, scale, image) of layers this kernel family covers: ceil(n / 16) launches
int tap3_pack_multi(const Canon* cs, const int* dirs, const float* const* ws, const float* const* scales, float* const* wps, int n, hipStream_t st) {
  for (int base = 0; base < n; base += PACK3_MULTI) {
    Pack3Table T;
    T.n = n - base < PACK3_MULTI ? n - base : PACK3_MULTI;
    T.first[0] = 0;
    for (int j = 0; j < T.n; ++j) {
      unsigned blocks;
      int rc = tap3_pack_args(cs[base + j], dirs[base + j], ws[base + j], scales[base + j], wps[base + j], &T.job[j], &blocks);
      if (rc) return rc;
      if (tap3_pack_coalesced(T.job[j], st, &rc)) {   // a big image: its own launches; an empty slot in this table
        if (rc) return rc;
        T.job[j].wunits = 0; T.job[j].tab_phase = 0;
        blocks = 1;
      }
      { static const unsigned cap = getenv("EBEN_PACK3_CAP") ? (unsigned)atoi(getenv("EBEN_PACK3_CAP")) : 512u; if (blocks > cap) blocks = cap; }   // many jobs share the launch: the grid-stride loop takes the rest
      T.first[j + 1] = T.first[j] + blocks;
    }
    hipLaunchKernelGGL(pack3_multi_kernel, dim3(T.first[T.n]), dim3(256), 0, st, T);
    EBEN_CHECK_LAUNCH("pack3_multi_kernel");
  }
  return EBEN_OK;
}

int tap3_launch(const Canon& c, int dir, const TapIO& io, int reflect, hipStream_t st) {
  Tap3Plan p;
  make_plan3(c, dir, &p);
  if (!p.ok) return fail(EBEN_EUNSUPPORTED, "tap3_launch on a layer the bf16 kernel does not cover");
  Tap3Args a;
  a.x = io.x; a.xmask = io.in_mode ? io.xmask : io.x; a.in_mode = io.in_mode; a.wp = reinterpret_cast<const u32x4*>(io.wp); a.tab = reinterpret_cast<const int*>(io.wp + p.tab_off_floats);
  a.bias = io.bias; a.res = io.res; a.emask = io.emask; a.y = io.y;
  a.xh = a.xl = nullptr; a.yh = a.yl = nullptr; a.eh = a.el = nullptr; a.ec = nullptr; a.CBx = a.CBy = a.bl_ref_off = a.bl_pad = 0;
  a.pr_S = a.pr_cbg = a.pr_Ly = a.pr_order = 0;
  if (c.bl) {
    if (!io.xh || !io.yh || (p.npx > 1 && !io.xl)) return fail(EBEN_EINVAL, "tap3: null bundle-layout plane");
    a.xh = static_cast<const u32x4*>(io.xh); a.xl = static_cast<const u32x4*>(io.xl);
    a.yh = static_cast<uint2*>(io.yh); a.yl = static_cast<uint2*>(io.yl);
    a.eh = static_cast<const uint2*>(io.eh); a.el = static_cast<const uint2*>(io.el); a.ec = static_cast<const unsigned*>(io.ec);
    a.CBx = p.Cx >> 3; a.CBy = p.Cy >> 3; a.bl_ref_off = io.bl_ref_off;
    a.x = nullptr; a.xmask = nullptr;
    if (io.pr_S > 0) {
      if (dir != 0 || p.S != 1 || p.nph != 1) return fail(EBEN_EINVAL, "tap3: phases-as-rows output on a launch that is not a stride-1 gather");
      a.pr_S = io.pr_S; a.pr_cbg = io.pr_cbg; a.pr_Ly = io.pr_Ly; a.CBy = io.pr_CBy; a.pr_order = io.pr_order;
      if (p.big && io.pr_order != 1) return fail(EBEN_EINVAL, "tap3: phases-as-rows order %d on a launch plan that is tap4's", io.pr_order);
      if (io.pr_order == 1 && ((io.pr_S != 4 && io.pr_S != 2) || (io.pr_S == 4 && (p.Mg & 31)) || (p.Mg & 15) || !c.bl || (p.big && io.pr_S != 4)))
        return fail(EBEN_EINVAL, "tap3: bundle-major phases as rows need stride 4 (whole 32-row tiles) or 2 (whole 16-row bundles)");
    }
  } else if (io.xh) {
    return fail(EBEN_EINVAL, "tap3: bundle-layout planes on a descriptor without EBEN_LAYOUT_BL");
  }
  a.B = c.B; a.G = p.G; a.Cg = p.Cg; a.Mg = p.Mg; a.Cx = p.Cx; a.Cy = p.Cy; a.Lx = p.Lx; a.Ly = p.Ly;
  a.S = p.S; a.OS = p.OS; a.dstep = p.dstep; a.J0 = p.J; a.mode = p.mode; a.off0 = p.off0; a.nt = p.nt; a.nph = p.nph;
  a.ps_pad = p.ps_pad; a.ps_k = c.k; a.ps_d = c.d; a.ps_kstep = p.kstep;
  a.reflect = reflect; a.accumulate = io.accumulate;
  a.res_rows = io.res_rows; a.em_seg = io.em_seg; a.fm_sums = io.fm_sums; a.fm_gs = io.fm_gs;
  for (int i = 0; i < 4; ++i) a.em_map[i] = io.em_map[i];
  a.in_slope = io.in_slope; a.out_slope = io.out_slope; a.res_slope = io.res_slope; a.emask_slope = io.emask_slope;
  a.CI_T = p.CI_T; a.CI_B = p.CI_B; a.CP = p.CP; a.ncc = p.ncc; a.PLEN = p.PLEN; a.CSTRIDE = p.CSTRIDE; a.nxb = p.nxbuf;
  a.s_magic = p.S > 1 ? (unsigned)((0x100000000ull + p.S - 1) / p.S) : 0u;
  a.ntt = p.ntt; a.nmt = p.nmt; a.tab_phase = p.tab_phase;
  a.w_tile = p.w_tile; a.w_phase = p.w_phase;
  const long long nb = (long long)p.ntt * c.B * p.nph * p.nmt * p.G;
  if (nb <= 0 || nb > 0x7fffffffLL) return fail(EBEN_EINVAL, "tap3 grid of %lld blocks", nb);
  a.big_XT = p.XT; a.big_PPT = p.PPT; a.big_tiles = (int)nb;
  a.m_cstride = (unsigned)((0x100000000ull + (unsigned)p.CSTRIDE - 1) / (unsigned)p.CSTRIDE);
  a.m_plen = p.S > 1 ? (unsigned)((0x100000000ull + (unsigned)p.PLEN - 1) / (unsigned)p.PLEN) : 0u;
  a.xq = (unsigned)(nb / 8); a.xr = (unsigned)(nb % 8);
  {
    auto magic = [](int d) { return (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); };
    int dmax = p.nph > p.ntt ? p.nph : p.ntt;
    if (c.B > dmax) dmax = c.B;
    if (p.nmt > dmax) dmax = p.nmt;
    // q = mulhi(n, ceil(2^32 / d)) is exact while n * d < 2^32 (error term (magic * d - 2^32) * n < 2^32 since magic * d - 2^32 < d)
    a.id_fast = nb * dmax < 0x100000000LL ? 1 : 0;
    a.m_nph = magic(p.nph); a.m_ntt = magic(p.ntt); a.m_B = magic(c.B); a.m_nmt = magic(p.nmt);
    if (p.nph == 1) a.m_nph = 0;   // d = 1: ceil(2^32 / 1) does not fit; handled below
    if (p.ntt == 1) a.m_ntt = 0;
    if (c.B == 1) a.m_B = 0;
    if (p.nmt == 1) a.m_nmt = 0;
    a.pg_n = p.nph <= 8 ? p.nph : 0;
    const int ad = p.dstep >= 0 ? p.dstep : -p.dstep;
    for (int ph = 0; ph < a.pg_n; ++ph) {
      const PhaseGeom q = phase_geom(p.mode, ph, p.J, p.off0, p.nt, p.dstep, p.OS, p.ps_pad, c.k, c.d, p.kstep, p.Ly);
      a.pg[ph].J = q.J; a.pg[ph].off0 = q.off0; a.pg[ph].minoff = q.minoff; a.pg[ph].nt = q.nt; a.pg[ph].oo = q.oo;
      a.pg[ph].span = q.J > 0 ? (p.BN - 1) * p.S + (q.J - 1) * ad + 1 : 0;
      a.pg[ph].span_magic = a.pg[ph].span > 0 ? magic(a.pg[ph].span) : 0u;
      a.pg[ph].pad = 0;
    }
  }
  if (p.big) {
    if (p.nph > 8) return fail(EBEN_EUNSUPPORTED, "tap4: more than 8 output phases");
    return tap4_launch(p, a, st);
  }
#define EBEN_T3_CASE(FMV)                                                          \
  switch (p.XRB) {                                                                 \
    case 2: return launch3_cfg<FMV, 2>(a, (int)nb, p.lds_bytes, p.npw, p.npx, st);               \
    case 3: return launch3_cfg<FMV, 3>(a, (int)nb, p.lds_bytes, p.npw, p.npx, st);               \
    default: return launch3_cfg<FMV, 5>(a, (int)nb, p.lds_bytes, p.npw, p.npx, st);              \
  }
  switch (p.FM) {
    case 1: EBEN_T3_CASE(1)
    case 2: EBEN_T3_CASE(2)
    case 3: EBEN_T3_CASE(3)
    default: EBEN_T3_CASE(4)
  }
#undef EBEN_T3_CASE
}

}  // namespace eben

#if EBEN_T3_DBG & 512
extern "C" __attribute__((visibility("default"))) int eben_debug_t3_stamps(unsigned long long* out, int rows) {
  if (rows > eben::T3_STAMP_ROWS) rows = eben::T3_STAMP_ROWS;
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(eben::t3_stamp), (size_t)rows * 128) != hipSuccess) return 1;
  if (!out) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(eben::t3_stamp)) != hipSuccess || hipMemset(p, 0, sizeof(unsigned long long) * eben::T3_STAMP_ROWS * 16) != hipSuccess) return 1;
  }
  return 0;
}
#endif
