// tapconv2.hip -- second-generation MFMA tap convolution (gfx950), used for every layer whose
// reduction is deep enough to pipeline (the MelGAN / PQMF-band discriminator bodies, the generator's
// residual / strided / transposed convs).  Same mathematics and fused stages as tapconv.hip:
//
//   y[b, g*Mg+m, t*OS+oo] = epi( sum_{c<Cg} sum_{j<J} W[j,c,m] * xin(b, g*Cg+c, t*S + off0 + j*dstep) )
//
// What differs is the machine mapping:
//   * v_mfma_f32_32x32x2_f32 (exact fp32, 64 cycles, one VGPR per operand): a wave owns FM 32x32
//     output tiles of one 32-position column strip; a block = NW waves = 32*FM rows x 32*NW positions.
//     One k-step = one tap x two channels, so a B fragment is 32 consecutive dwords of one staged
//     row per half-wave (conflict-free for any stride / dilation, no padding rules).
//   * weights are packed on the host side of the ABI (pack kernel) directly as the LDS image the A
//     fragments are read from -- [k-step][lane][FM] -- and streamed with global_load_lds_dwordx4
//     (LDS-DMA: no VGPR staging, no ds_write pass), double-buffered in 16-k-step chunks, one barrier
//     per chunk; an A fragment set is ONE ds_read_b128 / b64 / b32 per k-step.
//   * the B-side LDS offset of every k-step (tap phase / index / channel pair / input-tile buffer)
//     is a per-layer table built by the pack kernel and read through the scalar cache
//     (s_load_dwordx16 per chunk): no per-lane offset table, no integer division in the loop.
//   * input tiles are double-buffered: the next channel chunk's tile is fetched into registers under
//     the current chunk's MFMAs and written to the other LDS buffer one weight chunk before its
//     first use; the k-step table switches buffers mid-chunk, so the weight stream never restarts.
#include "common.h"

#include <cstdlib>

namespace eben {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int T2_XR = 16;      // input-tile elements a thread holds in flight
constexpr int T2_XR_BIG = 28;  // ... when 16 would leave fewer than two weight chunks per channel chunk (wide strided / transposed convs)

struct Tap2Args {
  const float* x; const float* xmask; const float* wp; const int* tab;
  const float* bias; const float* res; const float* emask; float* y;
  int B, G, Cg, Mg, Cx, Cy, Lx, Ly;
  int S, OS, dstep, J0, mode, off0, nt, nph;
  int ps_pad, ps_k, ps_d, ps_kstep;
  int reflect, in_mode, accumulate;
  int res_rows, em_seg, em_map[4];
  float in_slope, out_slope, res_slope, emask_slope;
  int CI_T, CP, ncc, PLEN, CSTRIDE, nxb;   // nxb: input-tile buffers the channel chunks rotate through (2 or 3)
  unsigned s_magic;
  int ntt, nmt, tab_phase;
  long long w_tile, w_phase;
};

template <int FI> struct AFrag;
template <> struct AFrag<1> { typedef float type; };
template <> struct AFrag<2> { typedef f32x2 type; };
template <> struct AFrag<4> { typedef f32x4 type; };

template <int FI>
__device__ __forceinline__ float a_elem(const typename AFrag<FI>::type& a, int i) { return a[i]; }
template <>
__device__ __forceinline__ float a_elem<1>(const float& a, int) { return a; }

// H16: 16-row variant on v_mfma_f32_16x16x4_f32 for layers with <= 16 rows per group and a deep reduction
// (MelGAN layer-2 input gradient: 16 rows x 704 taps*channels): a k-step is one tap x FOUR channels, a wave
// owns FNH = 4 fragments of 16 positions (block = 16 rows x 256 positions), everything else -- LDS-DMA weight
// image, scalar k-step table, double-buffered input tiles -- is shared with the 32x32x2 form.
#ifndef EBEN_T2_MINB_FM1
#define EBEN_T2_MINB_FM1 2   // blocks per CU the register allocation of the 32-row (FM = 1) kernels is held to
#endif
// IMT: -1 the mask-on-load mode (Tap2Args.in_mode) is a run-time switch; 0 / 1 compiled in -- the 28-element variants, which would
// otherwise hold 28 mask registers they never use and spill past the 256 registers of two blocks per CU
template <int FM, int NW, int XR, bool H16 = false, int IMT = -1>
__global__ __launch_bounds__(NW * 64, (FM == 1 && XR <= 16) ? EBEN_T2_MINB_FM1 : 2) void tap2_kernel(const Tap2Args P) {
  const bool im_on = IMT < 0 ? P.in_mode != 0 : IMT == 1;
  constexpr int NT = NW * 64;
  constexpr int FNH = 4;
  constexpr int BN = H16 ? NW * 16 * FNH : NW * 32;
  constexpr int BM = H16 ? 16 : FM * 32;
  constexpr int FI = H16 ? 1 : (FM == 3 ? 4 : FM);      // floats per lane per k-step in the A image
  constexpr int WCH = 16 * 64 * FI;         // floats per weight chunk (16 k-steps)
  constexpr int PIECES = WCH / 4 / NT;      // 16-byte LDS-DMA pieces per thread per chunk
  static_assert(PIECES >= 1 && PIECES * NT * 4 == WCH, "weight chunk must split into whole LDS-DMA pieces");
  typedef typename AFrag<FI>::type afrag_t;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;              // 2 x WCH
  float* Xa = smem + 2 * WCH;    // 1 or 2 input tiles of CI_T * CSTRIDE floats (16-byte aligned rows when S == 1)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- tile decode (XCD-aware: tiles sharing a weight panel stay on one XCD's L2) ----
  // (integer division runs on the VALU: readfirstlane brings the wave-uniform results back to SGPRs,
  // which is what lets the k-step table below go through the scalar cache)
  unsigned id = xcd_remap(blockIdx.x, gridDim.x);
  // the OS phases of one output tile interleave in memory (element t*OS + phase): neighbouring block ids -- the same XCD at about
  // the same time -- so that their partial cache lines meet in that XCD's L2 (tapconv3.hip: 10-20 % on the strided input gradients)
  const int ph = __builtin_amdgcn_readfirstlane(id % P.nph); id /= P.nph;
  const int tt = __builtin_amdgcn_readfirstlane(id % P.ntt); id /= P.ntt;
  const int b = __builtin_amdgcn_readfirstlane(id % P.B); id /= P.B;
  const int mt = __builtin_amdgcn_readfirstlane(id % P.nmt);
  const int g = __builtin_amdgcn_readfirstlane(id / P.nmt);
  const int t0 = tt * BN, m0 = mt * BM;

  const PhaseGeom q = phase_geom(P.mode, ph, P.J0, P.off0, P.nt, P.dstep, P.OS, P.ps_pad, P.ps_k, P.ps_d, P.ps_kstep, P.Ly);
  const int J = q.J, nt = q.nt, oo = q.oo;
  if (t0 >= nt) return;  // whole block outside this phase's range (uniform)
  const int adstep = P.dstep >= 0 ? P.dstep : -P.dstep;
  const int span = J > 0 ? (BN - 1) * P.S + (J - 1) * adstep + 1 : 0;
  const int KS_CC = J * P.CP;            // k-steps per channel chunk
  const int KS = P.ncc * KS_CC;
  const int nch = (KS + 15) >> 4;
  const int q0 = t0 * P.S + q.minoff;
  const int xtot = P.CI_T * span;
  const int XBUF = P.CI_T * P.CSTRIDE;
  const unsigned span_magic = span > 0 ? (unsigned)((0x100000000ull + (unsigned)span - 1) / (unsigned)span) : 0u;

  // Stride-1 tiles that lie wholly inside the signal are staged with aligned float4 loads: the tile then
  // starts at the 16-byte boundary below q0 and every canonical slot moves up by xshift = q0 & 3.
  const bool fast_x = P.S == 1 && (P.Lx & 3) == 0 && J > 0 && q0 >= 0 && q0 + span <= P.Lx;
  const int xshift = fast_x ? (q0 & 3) : 0;
  float* Xs = Xa + xshift;
  const float* wsrc = P.wp + (long long)ph * P.w_phase + ((long long)g * P.nmt + mt) * P.w_tile;
  // constant address space: the table is immutable for the kernel's lifetime, which lets the compiler
  // fetch it with s_load (a plain global pointer next to LDS-DMA writes is loaded per lane instead)
  typedef const __attribute__((address_space(4))) int* ctab_t;
  ctab_t tab = (ctab_t)(P.tab + (long long)ph * P.tab_phase);

  f32x16 acc[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f32x4 acch[FNH];
#pragma unroll
  for (int f = 0; f < FNH; ++f) acch[f] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Input-tile fetch, branch-free: every element's address is clamped to a valid one, all loads of a
  // batch are issued back to back (uniform base + 32-bit lane offset), the fused input stage and the
  // zero fill are selects afterwards (a branch per element costs one memory round trip per element).
  const long long xrow0 = ((long long)b * P.Cx + (long long)g * P.Cg) * P.Lx;
  const int refl = P.reflect;
  auto x_off = [&](int c, int r, int cc, int live, int& ok) -> int {
    int qq = q0 + r;
    const int m1 = qq < 0 ? -qq : qq;
    const int m2 = m1 >= P.Lx ? 2 * (P.Lx - 1) - m1 : m1;
    qq = refl ? m2 : qq;
    // bitwise on purpose: '&&' turns into one branch (and one exposed memory round trip) per element
    ok = live & (int)(cc * P.CI_T + c < P.Cg) & (int)(qq >= 0) & (int)(qq < P.Lx);
    return ok ? c * P.Lx + qq : 0;
  };
  auto x_slot = [&](int c, int r) -> int {
    int p = 0, d = r;
    if (P.S != 1) { d = (int)__umulhi((unsigned)r, P.s_magic); p = r - d * P.S; }
    return c * P.CSTRIDE + p * P.PLEN + d;
  };
  // coordinates of the XR tile elements this thread fetches for every channel chunk after the first
  // (registers for the 16-element tiles; the 28-element variants recompute them at every use from a copy of the thread id the
  // compiler cannot see through -- two VALU ops per element instead of 28 registers held across the whole k loop, which is what
  // keeps those kernels inside the 256 registers of two blocks per CU without scratch)
  constexpr bool XG_REG = XR <= T2_XR;
  auto xg_make = [&](int t, int u) -> int {
    const int i = t + u * NT;
    const int c = (int)__umulhi((unsigned)i, span_magic);
    return (P.ncc > 1 && i < xtot) ? ((c << 16) | (i - c * span)) : -1;
  };
  int xg_r[XG_REG ? XR : 1];
  if constexpr (XG_REG) {
#pragma unroll
    for (int u = 0; u < XR; ++u) xg_r[u] = xg_make(tid, u);
  }
  auto opaque_tid = [&]() -> int {
    int t = tid;
    if constexpr (!XG_REG) asm volatile("" : "+v"(t));
    return t;
  };
  // fetch_x only ISSUES the loads (raw values stay in flight in xreg / mreg under the MFMAs);
  // store_x applies the fused input stage and the zero fill when it writes the tile to LDS
  float xreg[XR], mreg[IMT == 0 ? 1 : XR];
  unsigned okmask = 0;
  auto fetch_x = [&](int cc) {
    const float* xp = P.x + xrow0 + (long long)cc * P.CI_T * P.Lx;
    const float* mp = P.xmask + xrow0 + (long long)cc * P.CI_T * P.Lx;
    okmask = 0;
    const int tq = opaque_tid();
#pragma unroll
    for (int u = 0; u < XR; ++u) {
      int ok;
      const int g_ = XG_REG ? xg_r[XG_REG ? u : 0] : xg_make(tq, u);
      const int o = x_off(g_ >> 16, g_ & 0xffff, cc, (int)(g_ >= 0), ok);
      okmask |= (unsigned)ok << u;
      xreg[u] = xp[o];
      if constexpr (IMT != 0) { if (im_on) mreg[u] = mp[o]; }
    }
  };
  const int dead_slot = P.nxb * XBUF;   // one spare float behind the tiles: lanes without an element store there
  auto store_x = [&](int cc) {
    const int bsel = cc % P.nxb;
    float* dst = Xs + bsel * XBUF;
    float t[XR];
    if (!im_on) {
#pragma unroll
      for (int u = 0; u < XR; ++u) t[u] = lrelu(xreg[u], P.in_slope);
    } else if constexpr (IMT != 0) {
#pragma unroll
      for (int u = 0; u < XR; ++u) t[u] = xreg[u] * dlrelu(mreg[u], P.in_slope);
    }
    const int tq = opaque_tid();
#pragma unroll
    for (int u = 0; u < XR; ++u) {
      const int g_ = XG_REG ? xg_r[XG_REG ? u : 0] : xg_make(tq, u);
      const int sl = g_ >= 0 ? x_slot(g_ >> 16, g_ & 0xffff) : dead_slot - bsel * XBUF;
      dst[sl] = ((okmask >> u) & 1u) ? t[u] : 0.f;
    }
  };

  auto issue_w = [&](int ch) {
    const float* src = wsrc + (long long)ch * WCH;
    float* dst = Ws + (ch & 1) * WCH;
#pragma unroll
    for (int u = 0; u < PIECES; ++u) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (u * NT + tid) * 4),
                                       (__attribute__((address_space(3))) void*)(dst + (u * NT + (tid & ~63)) * 4), 16, 0, 0);
    }
  };

  int written = 0;
  if (nch > 0) {
    issue_w(0);
    if (fast_x) {
      // tile 0, vector path: 4 x float4 in flight per thread, ds_write_b128
      const int span4 = (xshift + span + 3) >> 2;
      const int tot4 = P.CI_T * span4;
      const unsigned s4_magic = (unsigned)((0x100000000ull + (unsigned)span4 - 1) / (unsigned)span4);
      const float* xp = P.x + xrow0 + (q0 - xshift);
      const float* mp = im_on ? P.xmask + xrow0 + (q0 - xshift) : xp;
      for (int base = 0; base < tot4; base += 4 * NT) {
        f32x4 v[4], mk[4];
        int sl[4], ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = base + tid + u * NT;
          const int c = (int)__umulhi((unsigned)i, s4_magic);
          const int k4 = i - c * span4;
          ok[u] = (int)(i < tot4) & (int)(c < P.Cg);
          const long long o = ok[u] ? (long long)c * P.Lx + 4 * k4 : 0;
          v[u] = *reinterpret_cast<const f32x4*>(xp + o);
          mk[u] = *reinterpret_cast<const f32x4*>(mp + o);
          sl[u] = i < tot4 ? c * P.CSTRIDE + 4 * k4 : -1;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          f32x4 t;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float w = !im_on ? lrelu(v[u][e], P.in_slope) : v[u][e] * dlrelu(mk[u][e], P.in_slope);
            t[e] = ok[u] ? w : 0.f;
          }
          if (sl[u] >= 0) *reinterpret_cast<f32x4*>(Xa + sl[u]) = t;
        }
      }
    } else {
      const float* xp = P.x + xrow0;
      const float* mp = im_on ? P.xmask + xrow0 : xp;
      for (int base = 0; base < xtot; base += 8 * NT) {
        float v[8], mk[8];
        int sl[8], ok[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = base + tid + u * NT;
          const int c = (int)__umulhi((unsigned)i, span_magic);
          const int r = i - c * span;
          const int o = x_off(c, r, 0, (int)(i < xtot), ok[u]);
          v[u] = xp[o];
          mk[u] = mp[o];
          sl[u] = i < xtot ? x_slot(c, r) : -1;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float t = !im_on ? lrelu(v[u], P.in_slope) : v[u] * dlrelu(mk[u], P.in_slope);
          if (sl[u] >= 0) Xs[sl[u]] = ok[u] ? t : 0.f;
        }
      }
    }
    if (P.ncc > 1) fetch_x(1);
  }
  __syncthreads();

  const int lanebase = H16 ? (lane >> 4) * P.CSTRIDE + wn * 16 * FNH + (lane & 15) : (lane >> 5) * P.CSTRIDE + wn * 32 + (lane & 31);
  int te[16];   // this chunk's k-step offsets (wave-uniform: scalar registers, fetched one chunk ahead)
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) te[ks] = nch > 0 ? tab[ks] : 0;
  int pending = -1;   // tile whose global loads are issued at the top of the next chunk
  for (int ch = 0; ch < nch; ++ch) {
    if (ch + 1 < nch) issue_w(ch + 1);
    // the loads of the tile after the one just written are issued HERE, under this chunk's MFMAs: issued right before
    // the barrier they would be drained by its vmcnt(0) with their whole latency exposed -- once per weight chunk for a
    // pointwise conv (one tile hand-over per chunk), which is what held those layers at a third of the MFMA rate
    if (pending >= 0) { fetch_x(pending); pending = -1; }
    const float* wb = Ws + (ch & 1) * WCH + lane * FI;
    const float* xb = Xs + lanebase;
    // fragments of k-step s+2 are issued before the MFMAs of step s (sched_barrier pins the order:
    // left alone the scheduler sinks every ds_read next to its first use and exposes the LDS latency)
    if constexpr (H16) {
      float ah[16], bh[16][FNH];
      auto rdh = [&](int ks) {
        ah[ks] = wb[ks * 64];
#pragma unroll
        for (int f = 0; f < FNH; ++f) bh[ks][f] = xb[te[ks] + f * 16];
      };
      rdh(0);
      rdh(1);
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        if (ks + 2 < 16) rdh(ks + 2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < FNH; ++f) acch[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[ks], bh[ks][f], acch[f], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      float bv[16];
      afrag_t a[16];
      bv[0] = xb[te[0]]; a[0] = *reinterpret_cast<const afrag_t*>(wb);
      bv[1] = xb[te[1]]; a[1] = *reinterpret_cast<const afrag_t*>(wb + 64 * FI);
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        if (ks + 2 < 16) {
          bv[ks + 2] = xb[te[ks + 2]];
          a[ks + 2] = *reinterpret_cast<const afrag_t*>(wb + (ks + 2) * 64 * FI);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < FM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_elem<FI>(a[ks], i), bv[ks], acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (ch + 1 < nch) {
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) te[ks] = tab[(ch + 1) * 16 + ks];
    }
    // the tile first needed by the NEXT weight chunk goes to LDS now (its buffer was last read two
    // channel chunks ago, >= one barrier back); the tile after it starts loading into the registers
    if (P.ncc > 1 && written + 1 < P.ncc && (written + 1) * KS_CC < (ch + 2) * 16) {
      ++written;
      store_x(written);
      if (written + 1 < P.ncc) pending = written + 1;
    }
    __syncthreads();
  }

  // batched right-hand sides: residual row limit and remapped mask row (wave-uniform)
  const bool use_res = P.res != nullptr && (P.res_rows == 0 || b < P.res_rows);
  const int eb = P.em_seg > 0 ? P.em_map[b / P.em_seg] * P.em_seg + b % P.em_seg : b;
  const long long eoff = (long long)(eb - b) * P.Cy * P.Ly;
  if constexpr (H16) {
    // ---- epilogue, 16x16 D tile: column = lane & 15, row = (lane >> 4) * 4 + r ----
#pragma unroll
    for (int f = 0; f < FNH; ++f) {
      const int t = t0 + wn * 16 * FNH + f * 16 + (lane & 15);
      if (t >= nt) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + (lane >> 4) * 4 + r;
        if (m >= P.Mg) continue;
        const float bias = P.bias ? P.bias[g * P.Mg + m] : 0.f;
        const long long idx = ((long long)b * P.Cy + (long long)g * P.Mg + m) * P.Ly + (long long)t * P.OS + oo;
        float v = acch[f][r] + bias;
        v = lrelu(v, P.out_slope);
        if (use_res) v += lrelu(P.res[idx], P.res_slope);
        if (P.emask) v *= dlrelu(P.emask[idx + eoff], P.emask_slope);
        if (P.accumulate) v += P.y[idx];
        P.y[idx] = v;
      }
    }
    return;
  }
  // ---- epilogue: 32x32 D tile: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) ----
  const int t = t0 + wn * 32 + (lane & 31);
  if (t >= nt) return;
#pragma unroll
  for (int i = 0; i < FM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (m >= P.Mg) continue;
      const float bias = P.bias ? P.bias[g * P.Mg + m] : 0.f;
      const long long idx = ((long long)b * P.Cy + (long long)g * P.Mg + m) * P.Ly + (long long)t * P.OS + oo;
      float v = acc[i][r] + bias;
      v = lrelu(v, P.out_slope);
      if (use_res) v += lrelu(P.res[idx], P.res_slope);
      if (P.emask) v *= dlrelu(P.emask[idx + eoff], P.emask_slope);
      if (P.accumulate) v += P.y[idx];
      P.y[idx] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
struct Tap2Plan {
  int ok;
  int mode, G, Cg, Mg, S, OS, dstep, kstep, nph, J, Lx, Ly, Cx, Cy, off0, nt, ps_pad;
  int FM, NW, BM, BN, FI, WCH, H16, KCH;
  int CI_T, CP, ncc, PLEN, CSTRIDE, nxbuf, XR;
  int nmt, ntt, NCH, tab_phase;
  long long w_tile, w_phase, tab_off;
  size_t packed_floats, lds_bytes;
};

static int gcd2(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

static int env_int(const char* name, int dflt) {
  const char* s = getenv(name);
  return s ? atoi(s) : dflt;
}

static void make_plan2(const Canon& c, int dir, Tap2Plan* p) {
  p->ok = 0;
  p->mode = dir;
  p->G = c.g;
  if (dir == 0) {
    p->Cg = c.Cin / c.g; p->Mg = c.Cout / c.g;
    p->S = c.s; p->OS = 1; p->dstep = c.d; p->kstep = 1; p->nph = 1; p->J = c.k;
    p->Lx = c.Lin; p->Ly = c.Lout; p->Cx = c.Cin; p->Cy = c.Cout;
    p->off0 = -c.pl; p->nt = c.Lout; p->ps_pad = 0;
  } else {
    p->Cg = c.Cout / c.g; p->Mg = c.Cin / c.g;
    p->S = 1; p->OS = c.s;
    p->kstep = c.s / gcd2(c.s, c.d);
    p->dstep = -(c.d * p->kstep) / c.s;
    p->nph = c.s;
    p->J = ceil_div(c.k, p->kstep);
    p->Lx = c.Lout; p->Cx = c.Cout; p->Cy = c.Cin;
    p->Ly = c.reflect ? c.Lin + c.pl + c.pr : c.Lin;
    p->ps_pad = c.reflect ? 0 : c.pl;
    p->off0 = 0; p->nt = ceil_div(p->Ly, c.s);
  }
  static const int enabled = env_int("EBEN_TAP2", 1);
  static const int min_m = env_int("EBEN_TAP2_MIN_M", 24);
  static const int min_c = env_int("EBEN_TAP2_MIN_C", 2);
  // at least one full weight chunk (16 k-steps of one tap x two channels) of real reduction
  if (!enabled || p->Cg < min_c || p->nph > 64 || (long long)p->Cg * p->J < 32) return;
  // 16-row variant (16x16x4 MFMA): few rows per group but a reduction too deep for the direct kernel
  static const int h16_enabled = env_int("EBEN_TAP2_H16", 1);
  p->H16 = h16_enabled && p->Mg <= 16 && p->Mg >= 8 && (long long)p->Cg * p->J > 256;
  p->KCH = p->H16 ? 4 : 2;
  if (!p->H16 && p->Mg < min_m) return;
  // smallest J over the phases that have taps at all (phase-scatter phases differ by at most one tap)
  const int Jmin = dir == 0 ? p->J : (c.k / p->kstep > 0 ? c.k / p->kstep : 1);

  // rows per block: best (measured tile efficiency) x (useful rows / padded rows).  The efficiencies are the
  // ratios seen on MI355X between the four tile heights on MFMA-bound layers (FM = 4: ~110, 3: ~95-100,
  // 2: ~90, 1: ~50 TFLOP/s); pure least-padding picked 32-row tiles for the 514-row STFT analysis conv.
  const int cand[4] = {128, 96, 64, 32};
  const double eff[4] = {1.0, 0.9, 0.8, 0.45};
  int best = 0;
  double best_score = -1.0;
  for (int i = 0; i < 4; ++i) {
    const double score = eff[i] * p->Mg / round_up(p->Mg, cand[i]);
    if (score > best_score + 1e-9) { best_score = score; best = cand[i]; }
  }
  // a grid that cannot give every CU a block (the 125-sample layers of the generator: 32 items x one
  // position tile) is cut into shorter tiles while that does not add padded rows
  {
    const long long cols = (long long)ceil_div(p->nt, 128) * c.B * p->nph * p->G;
    static const int fill = env_int("EBEN_TAP2_FILL", 1);
    while (fill && best > 32 && cols * ceil_div(p->Mg, best) < 256) {
      const int smaller = best == 128 ? 64 : 32;   // 96 -> 32 keeps the row padding of 96
      if (round_up(p->Mg, smaller) > round_up(p->Mg, best)) break;
      best = smaller;
    }
  }
  static const int force_bm = env_int("EBEN_TAP2_BM", 0);
  if (force_bm == 32 || force_bm == 64 || force_bm == 96 || force_bm == 128) best = force_bm;
  p->BM = best; p->FM = best / 32; p->FI = p->FM == 3 ? 4 : p->FM;
  p->NW = 4; p->BN = 128;
  if (p->H16) { p->BM = 16; p->FM = 1; p->FI = 1; p->BN = 256; }
  p->WCH = 16 * 64 * p->FI;
  p->nmt = ceil_div(p->Mg, p->BM);
  p->ntt = ceil_div(p->nt, p->BN);

  const int adstep = p->dstep >= 0 ? p->dstep : -p->dstep;
  const int maxd = ((p->J - 1) * adstep) / p->S + 1;
  p->PLEN = p->BN + maxd + 1;
  // stride 1: rows 16-byte aligned with room for the float4 staging (tile start rounded down, end rounded up)
  if (p->S == 1) p->PLEN = round_up(p->BN + maxd + 5, 4);
  if (p->H16) {
    // the four channel rows of a 16x16x4 B fragment are read 2 x 16 lanes at a time: rows must sit 16 banks apart
    int pl = p->PLEN, tries = 0;
    while (((long long)p->S * pl) % 32 != 16 && tries < 64) { pl += p->S == 1 ? 4 : 1; ++tries; }
    if (((long long)p->S * pl) % 32 != 16) return;
    p->PLEN = pl;
  }
  p->CSTRIDE = p->S * p->PLEN;
  const int span = (p->BN - 1) * p->S + (p->J - 1) * adstep + 1;
  const int NT = p->NW * 64;
  static const int lds_budget = env_int("EBEN_TAP2_LDS_KB", 72) * 1024;   // two blocks per CU
  const int wbytes = 2 * p->WCH * 4;
  const int xbudget = lds_budget - wbytes;
  const int KCH = p->KCH;
  const int Cg2 = round_up(p->Cg, KCH);
  p->XR = T2_XR;
  if ((long long)Cg2 * p->CSTRIDE * 4 <= xbudget) {
    p->CI_T = Cg2; p->ncc = 1; p->nxbuf = 1;
  } else {
    // Tile hand-over: the tile first needed by weight chunk ch+1 is written at the end of chunk ch into the
    // buffer of the tile `nxbuf` channel chunks back, which must be fully consumed before chunk ch starts:
    // >= 32 k-steps per channel chunk with two buffers, >= 16 with three (pointwise convs over many
    // channels: the STFT-backward GEMMs, the 128-channel residual units).
    bool found = false;
    for (int nbuf = 2; nbuf <= 3 && !found; ++nbuf) {
      for (int xr : {T2_XR, T2_XR_BIG}) {
        int cap = (xr * NT) / span;
        // three buffers of a useful size do not fit the two-blocks-per-CU budget: that variant runs one block per CU
        const int cap_lds = (nbuf == 2 ? xbudget : 104 * 1024 - wbytes) / nbuf / (p->CSTRIDE * 4);
        if (cap > cap_lds) cap = cap_lds;
        cap -= cap % KCH;
        if (cap < KCH) continue;
        const int nchk = ceil_div(Cg2, cap);
        p->CI_T = round_up(ceil_div(p->Cg, nchk), KCH);
        p->ncc = ceil_div(p->Cg, p->CI_T);
        p->nxbuf = nbuf;
        p->XR = xr;
        // two buffers are also enough when the tiles change exactly on weight-chunk boundaries (every phase has J taps
        // and J * CI_T / KCH is a multiple of 16): the tile written at the end of chunk ch replaces the one last read in
        // chunk ch - 1.  That is the pointwise GEMMs over many channels (the STFT): 67 KB instead of 104 KB of LDS, two
        // blocks per CU instead of one.
        const int kscc = Jmin * (p->CI_T / KCH);
        const bool aligned = Jmin == p->J && kscc >= 16 && kscc % 16 == 0;
        if (p->ncc > 1 && kscc < (nbuf == 2 ? 32 : 16) && !(nbuf == 2 && aligned)) continue;
        found = true;
        break;
      }
    }
    if (!found) return;
  }
  p->CP = p->CI_T / KCH;
  const int KSmax = p->ncc * p->J * p->CP;
  p->NCH = ceil_div(KSmax, 16);
  p->tab_phase = p->NCH * 16;
  p->w_tile = (long long)p->NCH * p->WCH;
  p->w_phase = p->w_tile * p->nmt * p->G;
  p->tab_off = p->w_phase * p->nph;
  p->packed_floats = (size_t)p->tab_off + (size_t)p->tab_phase * p->nph;
  p->lds_bytes = (size_t)wbytes + (size_t)p->nxbuf * p->CI_T * p->CSTRIDE * 4 + 32;   // + spare slot + tile shift
  if (p->lds_bytes > 160 * 1024) return;
  p->ok = 1;
}

struct Pack2Args {
  const float* w; const float* scale; float* wp;
  int G, Cg, Mg, nmt, BM, FM, FI, WCH, CI_T, CP, ncc, NCH, nph, tab_phase;
  int mode, J0, off0, nt, dstep, OS, S, ps_pad, k, d, kstep, Ly;
  int Cin_g, Cout_g, PLEN, CSTRIDE, nxbuf, H16, KCH;
  long long w_tile, w_phase, tab_off;
};

__global__ __launch_bounds__(256) void pack2_kernel(const Pack2Args P) {
  const long long wtotal = P.tab_off;
  const long long total = wtotal + (long long)P.tab_phase * P.nph;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    if (i < wtotal) {
      long long r = i;
      const int ph = (int)(r / P.w_phase); r -= (long long)ph * P.w_phase;
      const int tile = (int)(r / P.w_tile); r -= (long long)tile * P.w_tile;
      const int g = tile / P.nmt, mt = tile - g * P.nmt;
      const int ch = (int)(r / P.WCH);
      const int e = (int)(r - (long long)ch * P.WCH);
      const int ks = e / (64 * P.FI);
      const int lane = (e / P.FI) & 63;
      const int fi = e % P.FI;
      const PhaseGeom q = phase_geom(P.mode, ph, P.J0, P.off0, P.nt, P.dstep, P.OS, P.ps_pad, P.k, P.d, P.kstep, P.Ly);
      const int s = ch * 16 + ks;
      const int KS_CC = q.J * P.CP;
      float v = 0.f;
      if (q.J > 0 && s < P.ncc * KS_CC && fi < P.FM) {
        const int cc = s / KS_CC, rem = s - cc * KS_CC;
        const int j = rem / P.CP, cp = rem - j * P.CP;
        const int chan = P.H16 ? cc * P.CI_T + 4 * cp + (lane >> 4) : cc * P.CI_T + 2 * cp + (lane >> 5);
        const int m = P.H16 ? mt * 16 + (lane & 15) : mt * P.BM + fi * 32 + (lane & 31);
        if (chan < P.Cg && m < P.Mg) {
          if (P.mode == 0) {
            const int co = g * P.Cout_g + m;
            v = P.w[((long long)co * P.Cin_g + chan) * P.k + j];
            if (P.scale) v *= P.scale[co];
          } else {
            const int kk = q.k0 + j * P.kstep;
            if (kk < P.k) {
              const int co = g * P.Cout_g + chan;  // reduction channel = conv output channel
              v = P.w[((long long)co * P.Cin_g + m) * P.k + kk];
              if (P.scale) v *= P.scale[co];
            }
          }
        }
      }
      P.wp[i] = v;
    } else {
      const long long r = i - wtotal;
      const int ph = (int)(r / P.tab_phase);
      const int s = (int)(r - (long long)ph * P.tab_phase);
      const PhaseGeom q = phase_geom(P.mode, ph, P.J0, P.off0, P.nt, P.dstep, P.OS, P.ps_pad, P.k, P.d, P.kstep, P.Ly);
      const int KS_CC = q.J * P.CP;
      int o = 0;
      if (q.J > 0 && s < P.ncc * KS_CC) {
        const int cc = s / KS_CC, rem = s - cc * KS_CC;
        const int j = rem / P.CP, cp = rem - j * P.CP;
        const int rel = q.off0 + j * P.dstep - q.minoff;
        const int dd = rel / P.S, pp = rel - dd * P.S;
        o = (P.nxbuf > 1 ? (cc % P.nxbuf) * P.CI_T * P.CSTRIDE : 0) + P.KCH * cp * P.CSTRIDE + pp * P.PLEN + dd;
      }
      reinterpret_cast<int*>(P.wp)[i] = o;
    }
  }
}

template <int FM, int NW, int XR, bool H16 = false, int IMT = -1>
static int launch2_cfg(const Tap2Args& a, int nblocks, size_t lds, hipStream_t st) {
  static LdsAttrOnce attr_once;
  auto kern = tap2_kernel<FM, NW, XR, H16, IMT>;
  {
    const hipError_t e = lds_attr_once(attr_once, reinterpret_cast<const void*>(kern));
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(tap2)");
  }
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(NW * 64), lds, st, a);
  EBEN_CHECK_LAUNCH("tap2_kernel");
  return EBEN_OK;
}

int tap2_applicable(const Canon& c, int dir) {
  Tap2Plan p;
  make_plan2(c, dir, &p);
  return p.ok;
}

size_t tap2_packed_floats(const Canon& c, int dir) {
  Tap2Plan p;
  make_plan2(c, dir, &p);
  return p.ok ? p.packed_floats : 0;
}

int tap2_pack(const Canon& c, int dir, const float* w, const float* scale, float* wp, hipStream_t st) {
  Tap2Plan p;
  make_plan2(c, dir, &p);
  if (!p.ok) return fail(EBEN_EUNSUPPORTED, "tap2_pack on a layer the second-generation kernel does not cover");
  Pack2Args a;
  a.w = w; a.scale = scale; a.wp = wp;
  a.G = p.G; a.Cg = p.Cg; a.Mg = p.Mg; a.nmt = p.nmt; a.BM = p.BM; a.FM = p.FM; a.FI = p.FI; a.WCH = p.WCH;
  a.CI_T = p.CI_T; a.CP = p.CP; a.ncc = p.ncc; a.NCH = p.NCH; a.nph = p.nph; a.tab_phase = p.tab_phase;
  a.mode = p.mode; a.J0 = p.J; a.off0 = p.off0; a.nt = p.nt; a.dstep = p.dstep; a.OS = p.OS; a.S = p.S; a.ps_pad = p.ps_pad;
  a.k = c.k; a.d = c.d; a.kstep = p.kstep; a.Ly = p.Ly;
  a.Cin_g = c.Cin / c.g; a.Cout_g = c.Cout / c.g; a.PLEN = p.PLEN; a.CSTRIDE = p.CSTRIDE; a.nxbuf = p.nxbuf; a.H16 = p.H16; a.KCH = p.KCH;
  a.w_tile = p.w_tile; a.w_phase = p.w_phase; a.tab_off = p.tab_off;
  long long blocks = ((long long)p.packed_floats + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(pack2_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
  EBEN_CHECK_LAUNCH("pack2_kernel");
  return EBEN_OK;
}

int tap2_launch(const Canon& c, int dir, const TapIO& io, int reflect, hipStream_t st) {
  Tap2Plan p;
  make_plan2(c, dir, &p);
  if (!p.ok) return fail(EBEN_EUNSUPPORTED, "tap2_launch on a layer the second-generation kernel does not cover");
  Tap2Args a;
  a.x = io.x; a.xmask = io.xmask; a.wp = io.wp; a.tab = reinterpret_cast<const int*>(io.wp + p.tab_off);
  a.bias = io.bias; a.res = io.res; a.emask = io.emask; a.y = io.y;
  a.B = c.B; a.G = p.G; a.Cg = p.Cg; a.Mg = p.Mg; a.Cx = p.Cx; a.Cy = p.Cy; a.Lx = p.Lx; a.Ly = p.Ly;
  a.S = p.S; a.OS = p.OS; a.dstep = p.dstep; a.J0 = p.J; a.mode = p.mode; a.off0 = p.off0; a.nt = p.nt; a.nph = p.nph;
  a.ps_pad = p.ps_pad; a.ps_k = c.k; a.ps_d = c.d; a.ps_kstep = p.kstep;
  a.reflect = reflect; a.in_mode = io.in_mode; a.accumulate = io.accumulate;
  a.res_rows = io.res_rows; a.em_seg = io.em_seg;
  for (int i = 0; i < 4; ++i) a.em_map[i] = io.em_map[i];
  a.in_slope = io.in_slope; a.out_slope = io.out_slope; a.res_slope = io.res_slope; a.emask_slope = io.emask_slope;
  a.CI_T = p.CI_T; a.CP = p.CP; a.ncc = p.ncc; a.PLEN = p.PLEN; a.CSTRIDE = p.CSTRIDE; a.nxb = p.nxbuf < 2 ? 2 : p.nxbuf;
  a.s_magic = p.S > 1 ? (unsigned)((0x100000000ull + p.S - 1) / p.S) : 0u;
  a.ntt = p.ntt; a.nmt = p.nmt; a.tab_phase = p.tab_phase;
  a.w_tile = p.w_tile; a.w_phase = p.w_phase;
  const long long nb = (long long)p.ntt * c.B * p.nph * p.nmt * p.G;
  if (nb <= 0 || nb > 0x7fffffffLL) return fail(EBEN_EINVAL, "tap2 grid of %lld blocks", nb);
  const bool im = io.in_mode != 0;
  if (p.H16) return p.XR == T2_XR_BIG ? (im ? launch2_cfg<1, 4, T2_XR_BIG, true, 1>(a, (int)nb, p.lds_bytes, st)
                                            : launch2_cfg<1, 4, T2_XR_BIG, true, 0>(a, (int)nb, p.lds_bytes, st))
                                       : launch2_cfg<1, 4, T2_XR, true>(a, (int)nb, p.lds_bytes, st);
  if (p.XR == T2_XR_BIG) {
#define EBEN_T2_BIG(FMV) return im ? launch2_cfg<FMV, 4, T2_XR_BIG, false, 1>(a, (int)nb, p.lds_bytes, st) \
                                   : launch2_cfg<FMV, 4, T2_XR_BIG, false, 0>(a, (int)nb, p.lds_bytes, st)
    switch (p.FM) {
      case 1: EBEN_T2_BIG(1);
      case 2: EBEN_T2_BIG(2);
      case 3: EBEN_T2_BIG(3);
      default: EBEN_T2_BIG(4);
    }
#undef EBEN_T2_BIG
  }
  switch (p.FM) {
    case 1: return launch2_cfg<1, 4, T2_XR>(a, (int)nb, p.lds_bytes, st);
    case 2: return launch2_cfg<2, 4, T2_XR>(a, (int)nb, p.lds_bytes, st);
    case 3: return launch2_cfg<3, 4, T2_XR>(a, (int)nb, p.lds_bytes, st);
    default: return launch2_cfg<4, 4, T2_XR>(a, (int)nb, p.lds_bytes, st);
  }
}

}  // namespace eben
