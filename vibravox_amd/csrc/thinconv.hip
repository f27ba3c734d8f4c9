// thinconv.hip -- direct (VALU) tap convolution for the layers an MFMA tile cannot fill (gfx950):
// a handful of channels per group on the output side (Cout/g or, for an input gradient, Cin/g of
// 4..16: the first layers of the PQMF-band and MelGAN discriminators -- eben_discriminator.py:66-90,
// melgan_discriminator.py:89-110 -- their input gradients, the generator's first / last conv).
// Same mathematics and fused stages as tapconv.hip:
//
//   y[b, g*Mg+m, t*OS+oo] = epi( sum_{c<Cg} sum_{j<J} W[j,c,m] * xin(b, g*Cg+c, t*S + off0 + j*dstep) )
//
// A 32x32 (or 16x16) MFMA tile would be 75..88 % padding on these shapes and the per-block set-up of the
// GEMM kernels (weight chunk, offset table) outweighs the arithmetic.  Here one thread owns one output
// position and MT output channels: the input tile is staged in LDS (de-interleaved by stride phase:
// lanes read consecutive dwords), the MT weights of a (tap, channel) pair are wave-uniform and arrive
// in scalar registers through the constant cache (s_load_dwordx4/8/16), so the inner loop is one
// ds_read_b32 + MT v_fmac_f32 with an SGPR operand.  fp32 FMA chain in the same (tap-major,
// channel-minor) order for every output.
#include "common.h"

#include <cstdlib>

namespace eben {

struct ThinArgs {
  const float* x; const float* xmask; const float* wp;
  const float* bias; const float* res; const float* emask; float* y;
  int B, G, Cg, Mg, Cx, Cy, Lx, Ly;
  int S, OS, dstep, J0, mode, off0, nt, nph;
  int ps_pad, ps_k, ps_d, ps_kstep;
  int reflect, in_mode, accumulate;
  int res_rows, em_seg, em_map[4];
  const float* fm_sums; float fm_gs;
  float in_slope, out_slope, res_slope, emask_slope;
  int PLEN, CSTRIDE;
  unsigned s_magic;
  int ntt, nmt;
  long long w_tile;   // floats per (phase, group, m-tile) weight panel: J0 * Cg * MT
  // block prologue arithmetic done on the host (as tapconv3.hip): block-id decomposition by multiply-high (magic 0 = divisor 1),
  // per-phase tap geometry and span magic for up to 8 phases
  unsigned m_ntt, m_B, m_nph, m_nmt;
  int id_fast, pg_n, tile;
  struct PG { int J, off0, minoff, nt, oo, span; unsigned span_magic; int pad; } pg[8];
};

// NP: output positions per thread (block = BN * NP positions, position tid + n * BN: every store stays coalesced).  A block of
// the first discriminator layers is a few dozen FMAs per thread behind a fixed preamble (phase geometry, tile staging, barrier);
// four positions per thread amortise it and reuse each scalar weight four times.  Same FMA order per output: same bits.
template <int MT, int BN, int NP = 1>
__global__ __launch_bounds__(BN) void thin_kernel(const ThinArgs P) {
  constexpr int TILE = BN * NP;
  extern __shared__ __attribute__((aligned(16))) float Xs[];
  const int tid = threadIdx.x;

  unsigned id = blockIdx.x;
  int tt, b, ph, mt, g;
  if (P.id_fast) {
    unsigned qd = P.m_ntt ? __umulhi(id, P.m_ntt) : id; tt = (int)(id - qd * (unsigned)P.ntt); id = qd;
    qd = P.m_B ? __umulhi(id, P.m_B) : id; b = (int)(id - qd * (unsigned)P.B); id = qd;
    qd = P.m_nph ? __umulhi(id, P.m_nph) : id; ph = (int)(id - qd * (unsigned)P.nph); id = qd;
    qd = P.m_nmt ? __umulhi(id, P.m_nmt) : id; mt = (int)(id - qd * (unsigned)P.nmt); g = (int)qd;
  } else {
    tt = id % P.ntt; id /= P.ntt;
    b = id % P.B; id /= P.B;
    ph = id % P.nph; id /= P.nph;
    mt = id % P.nmt;
    g = id / P.nmt;
  }
  tt = __builtin_amdgcn_readfirstlane(tt); b = __builtin_amdgcn_readfirstlane(b); ph = __builtin_amdgcn_readfirstlane(ph);
  mt = __builtin_amdgcn_readfirstlane(mt); g = __builtin_amdgcn_readfirstlane(g);
  const int t0 = tt * TILE, m0 = mt * MT;

  PhaseGeom q;
  int span;
  unsigned span_magic;
  if (ph < P.pg_n) {
    q.J = P.pg[ph].J; q.off0 = P.pg[ph].off0; q.minoff = P.pg[ph].minoff; q.nt = P.pg[ph].nt; q.oo = P.pg[ph].oo; q.k0 = 0;
    span = P.pg[ph].span; span_magic = P.pg[ph].span_magic;
  } else {
    q = phase_geom(P.mode, ph, P.J0, P.off0, P.nt, P.dstep, P.OS, P.ps_pad, P.ps_k, P.ps_d, P.ps_kstep, P.Ly);
    const int adstep = P.dstep >= 0 ? P.dstep : -P.dstep;
    span = q.J > 0 ? (TILE - 1) * P.S + (q.J - 1) * adstep + 1 : 0;
    span_magic = span > 0 ? (unsigned)((0x100000000ull + (unsigned)span - 1) / (unsigned)span) : 0u;
  }
  const int J = q.J, nt = q.nt, oo = q.oo;
  if (t0 >= nt) return;
  const int q0 = t0 * P.S + q.minoff;
  const int xtot = P.Cg * span;

  // ---- stage the input tile: 8 loads in flight per thread, branch-free ----
  {
    const long long xrow0 = ((long long)b * P.Cx + (long long)g * P.Cg) * P.Lx;
    const float* xp = P.x + xrow0;
    const float* mp = P.in_mode ? P.xmask + xrow0 : xp;
    for (int base = 0; base < xtot; base += 8 * BN) {
      float v[8], mk[8];
      int sl[8], ok[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + tid + u * BN;
        const int c = (int)__umulhi((unsigned)i, span_magic);
        const int r = i - c * span;
        int qq = q0 + r;
        const int m1 = qq < 0 ? -qq : qq;
        const int m2 = m1 >= P.Lx ? 2 * (P.Lx - 1) - m1 : m1;
        qq = P.reflect ? m2 : qq;
        const int live = (int)(i < xtot);
        ok[u] = live & (int)(qq >= 0) & (int)(qq < P.Lx);
        const int o = ok[u] ? c * P.Lx + qq : 0;
        v[u] = xp[o];
        mk[u] = P.in_mode ? mp[o] : 0.f;
        int p = 0, d = r;
        if (P.S != 1) { d = (int)__umulhi((unsigned)r, P.s_magic); p = r - d * P.S; }
        sl[u] = live ? c * P.CSTRIDE + p * P.PLEN + d : -1;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float t = P.in_mode == 0 ? (P.in_slope == 1.f ? v[u] : lrelu(v[u], P.in_slope)) : v[u] * dlrelu(mk[u], P.in_slope);
        if (sl[u] >= 0) Xs[sl[u]] = ok[u] ? t : 0.f;
      }
    }
  }
  __syncthreads();

  float acc[NP][MT];
#pragma unroll
  for (int n = 0; n < NP; ++n)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[n][m] = 0.f;

  // weights: [phase][group][m-tile][tap][channel][MT], wave-uniform -> constant address space -> s_load
  typedef const __attribute__((address_space(4))) float* cw_t;
  cw_t w = (cw_t)(P.wp + (((long long)ph * P.G + g) * P.nmt + mt) * P.w_tile);
  for (int j = 0; j < J; ++j) {
    const int rel = q.off0 + j * P.dstep - q.minoff;
    int pp = 0, dd = rel;
    if (P.S != 1) { dd = (int)__umulhi((unsigned)rel, P.s_magic); pp = rel - dd * P.S; }
    const float* xr = Xs + pp * P.PLEN + dd + tid;
    cw_t wj = w + (long long)j * P.Cg * MT;
#pragma unroll 4
    for (int c = 0; c < P.Cg; ++c) {
      float xv[NP];
#pragma unroll
      for (int n = 0; n < NP; ++n) xv[n] = xr[c * P.CSTRIDE + n * BN];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const float wv = wj[c * MT + m];
#pragma unroll
        for (int n = 0; n < NP; ++n) acc[n][m] = fmaf(wv, xv[n], acc[n][m]);
      }
    }
  }

  // batched right-hand sides: residual row limit and remapped mask row (wave-uniform)
  const bool use_res = P.res != nullptr && (P.res_rows == 0 || b < P.res_rows);
  const int eb = P.em_seg > 0 ? P.em_map[b / P.em_seg] * P.em_seg + b % P.em_seg : b;
  const long long eoff = (long long)(eb - b) * P.Cy * P.Ly;
  // the operands of the fused stages are read for all MT rows of a position in one batch each (clamped rows, no per-value
  // branch): value by value the compiler waits for every load in turn
  const long long ubase = ((long long)b * P.Cy + (long long)g * P.Mg) * P.Ly;
  float* __restrict__ yb = P.y + ubase;
  const float* __restrict__ rb = P.res + ubase;
  const float* __restrict__ eb_ = P.emask + ubase + eoff;
  const int mlast = P.Mg - 1;
  // feature-matching rows: the addend is formed from the two embeddings (mask = enhanced rows, res = reference rows) instead of read
  const bool fm = use_res && P.fm_sums != nullptr;
  float fc1 = 0.f, fc2 = 0.f;
  if (fm) { const float s1 = P.fm_sums[0], s2 = P.fm_sums[1]; fc1 = P.fm_gs / s2; fc2 = P.fm_gs * s1 / (s2 * s2); }
#pragma unroll
  for (int n = 0; n < NP; ++n) {
    const int t = t0 + tid + n * BN;
    if (t >= nt) continue;
    const unsigned col = (unsigned)t * (unsigned)P.OS + (unsigned)oo;
    constexpr int MB = MT < 8 ? MT : 8;   // rows per batch of loads (registers)
#pragma unroll
    for (int mb = 0; mb < MT; mb += MB) {
      float bz[MB], rz[MB], ez[MB], az[MB];
      unsigned off[MB];
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        const int mc = m0 + mb + m < mlast ? m0 + mb + m : mlast;
        off[m] = (unsigned)mc * (unsigned)P.Ly + col;
        bz[m] = P.bias ? P.bias[g * P.Mg + mc] : 0.f;
      }
      if (use_res) {
#pragma unroll
        for (int m = 0; m < MB; ++m) rz[m] = rb[off[m]];
      }
      if (P.emask) {
#pragma unroll
        for (int m = 0; m < MB; ++m) ez[m] = eb_[off[m]];
      }
      if (P.accumulate) {
#pragma unroll
        for (int m = 0; m < MB; ++m) az[m] = yb[off[m]];
      }
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        float v = acc[n][mb + m] + bz[m];
        v = lrelu(v, P.out_slope);
        if (fm) {
          const float av = ez[m], dv = av - rz[m];
          v += fc1 * (float)((dv > 0.f) - (dv < 0.f)) - fc2 * (float)((av > 0.f) - (av < 0.f));
        } else if (use_res) {
          v += lrelu(rz[m], P.res_slope);
        }
        if (P.emask) v *= dlrelu(ez[m], P.emask_slope);
        if (P.accumulate) v += az[m];
        if (m0 + mb + m < P.Mg) yb[off[m]] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
struct ThinPlan {
  int ok;
  int mode, G, Cg, Mg, S, OS, dstep, kstep, nph, J, Lx, Ly, Cx, Cy, off0, nt, ps_pad;
  int MT, BN, NP, nmt, ntt, PLEN, CSTRIDE;
  long long w_tile;
  size_t packed_floats, lds_bytes;
};

static int thin_env(const char* name, int dflt) {
  const char* s = getenv(name);
  return s ? atoi(s) : dflt;
}
static int gcd3(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

static void make_thin_plan(const Canon& c, int dir, ThinPlan* p) {
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
    p->kstep = c.s / gcd3(c.s, c.d);
    p->dstep = -(c.d * p->kstep) / c.s;
    p->nph = c.s;
    p->J = ceil_div(c.k, p->kstep);
    p->Lx = c.Lout; p->Cx = c.Cout; p->Cy = c.Cin;
    p->Ly = c.reflect ? c.Lin + c.pl + c.pr : c.Lin;
    p->ps_pad = c.reflect ? 0 : c.pl;
    p->off0 = 0; p->nt = ceil_div(p->Ly, c.s);
  }
  static const int enabled = thin_env("EBEN_THIN", 1);
  static const int max_m = thin_env("EBEN_THIN_MAX_M", 32);
  // deeper reductions belong on the MFMA kernels (MelGAN layer-2 input gradient, 704 taps x channels
  // for 16 rows: 0.87 ms here against 0.50 ms on 16-row MFMA tiles)
  static const int max_k = thin_env("EBEN_THIN_MAX_K", 256);
  // many rows are fine when the reduction is a handful of taps (input gradient of the logits layers:
  // 768 / 1024 rows from ONE channel x 3 taps -- an outer product, bound by the output write)
  const bool tiny_k = (long long)p->Cg * p->J <= 16;
  if (!enabled || p->Mg < 1 || (p->Mg > max_m && !tiny_k) || (long long)p->Cg * p->J > max_k || p->nph > 64) return;
  p->MT = p->Mg == 1 ? 1 : p->Mg <= 4 ? 4 : p->Mg <= 8 ? 8 : 16;
  p->nmt = ceil_div(p->Mg, p->MT);
  const int adstep = p->dstep >= 0 ? p->dstep : -p->dstep;
  const int maxd = ((p->J - 1) * adstep) / p->S + 1;
  p->BN = p->nt <= 128 ? 128 : 256;
  static const int np4 = thin_env("EBEN_THIN_NP", 4);
  static const int np_min_blocks = thin_env("EBEN_THIN_NP_MIN_BLOCKS", 2048);
  p->NP = (np4 == 4 && p->BN == 256 && p->nt >= 4096 &&
           (long long)ceil_div(p->nt, 1024) * c.B * p->nph * p->nmt * p->G >= np_min_blocks) ? 4 : 1;
  for (;;) {
    p->PLEN = p->BN * p->NP + maxd + 1;
    p->CSTRIDE = p->S * p->PLEN;
    p->lds_bytes = 4ull * p->Cg * p->CSTRIDE;
    if (p->lds_bytes <= 72 * 1024 || (p->BN == 128 && p->NP == 1)) break;
    if (p->NP > 1) p->NP = 1; else p->BN = 128;
  }
  if (p->lds_bytes > 150 * 1024) return;
  p->ntt = ceil_div(p->nt, p->BN * p->NP);
  p->w_tile = (long long)p->J * p->Cg * p->MT;
  p->packed_floats = (size_t)p->w_tile * p->nmt * p->G * p->nph;
  p->ok = 1;
}

struct ThinPackArgs {
  const float* w; const float* scale; float* wp;
  int G, Cg, Mg, MT, nmt, nph, J0;
  int mode, off0, nt, dstep, OS, ps_pad, k, d, kstep, Ly, Cin_g, Cout_g;
  long long w_tile;
};

__global__ __launch_bounds__(256) void thin_pack_kernel(const ThinPackArgs P) {
  const long long total = P.w_tile * P.nmt * P.G * P.nph;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    long long r = i;
    const int ml = (int)(r % P.MT); r /= P.MT;
    const int c = (int)(r % P.Cg); r /= P.Cg;
    const int j = (int)(r % P.J0); r /= P.J0;
    const int mt = (int)(r % P.nmt); r /= P.nmt;
    const int g = (int)(r % P.G);
    const int ph = (int)(r / P.G);
    const PhaseGeom q = phase_geom(P.mode, ph, P.J0, P.off0, P.nt, P.dstep, P.OS, P.ps_pad, P.k, P.d, P.kstep, P.Ly);
    const int m = mt * P.MT + ml;
    float v = 0.f;
    if (j < q.J && m < P.Mg) {
      if (P.mode == 0) {
        const int co = g * P.Cout_g + m;
        v = P.w[((long long)co * P.Cin_g + c) * P.k + j];
        if (P.scale) v *= P.scale[co];
      } else {
        const int kk = q.k0 + j * P.kstep;
        if (kk < P.k) {
          const int co = g * P.Cout_g + c;   // reduction channel = conv output channel
          v = P.w[((long long)co * P.Cin_g + m) * P.k + kk];
          if (P.scale) v *= P.scale[co];
        }
      }
    }
    P.wp[i] = v;
  }
}

template <int MT, int BN, int NP = 1>
static int launch_thin_cfg(const ThinArgs& a, int nblocks, size_t lds, hipStream_t st) {
  static LdsAttrOnce attr_once;
  auto kern = thin_kernel<MT, BN, NP>;
  {
    const hipError_t e = lds_attr_once(attr_once, reinterpret_cast<const void*>(kern));
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(thin)");
  }
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(BN), lds, st, a);
  EBEN_CHECK_LAUNCH("thin_kernel");
  return EBEN_OK;
}

int thin_applicable(const Canon& c, int dir) {
  ThinPlan p;
  make_thin_plan(c, dir, &p);
  return p.ok;
}

size_t thin_packed_floats(const Canon& c, int dir) {
  ThinPlan p;
  make_thin_plan(c, dir, &p);
  return p.ok ? p.packed_floats : 0;
}

int thin_pack(const Canon& c, int dir, const float* w, const float* scale, float* wp, hipStream_t st) {
  ThinPlan p;
  make_thin_plan(c, dir, &p);
  if (!p.ok) return fail(EBEN_EUNSUPPORTED, "thin_pack on a layer the direct kernel does not cover");
  ThinPackArgs a;
  a.w = w; a.scale = scale; a.wp = wp;
  a.G = p.G; a.Cg = p.Cg; a.Mg = p.Mg; a.MT = p.MT; a.nmt = p.nmt; a.nph = p.nph; a.J0 = p.J;
  a.mode = p.mode; a.off0 = p.off0; a.nt = p.nt; a.dstep = p.dstep; a.OS = p.OS; a.ps_pad = p.ps_pad;
  a.k = c.k; a.d = c.d; a.kstep = p.kstep; a.Ly = p.Ly; a.Cin_g = c.Cin / c.g; a.Cout_g = c.Cout / c.g;
  a.w_tile = p.w_tile;
  long long blocks = ((long long)p.packed_floats + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(thin_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
  EBEN_CHECK_LAUNCH("thin_pack_kernel");
  return EBEN_OK;
}

int thin_launch(const Canon& c, int dir, const TapIO& io, int reflect, hipStream_t st) {
  ThinPlan p;
  make_thin_plan(c, dir, &p);
  if (!p.ok) return fail(EBEN_EUNSUPPORTED, "thin_launch on a layer the direct kernel does not cover");
  ThinArgs a;
  a.x = io.x; a.xmask = io.xmask; a.wp = io.wp; a.bias = io.bias; a.res = io.res; a.emask = io.emask; a.y = io.y;
  a.B = c.B; a.G = p.G; a.Cg = p.Cg; a.Mg = p.Mg; a.Cx = p.Cx; a.Cy = p.Cy; a.Lx = p.Lx; a.Ly = p.Ly;
  a.S = p.S; a.OS = p.OS; a.dstep = p.dstep; a.J0 = p.J; a.mode = p.mode; a.off0 = p.off0; a.nt = p.nt; a.nph = p.nph;
  a.ps_pad = p.ps_pad; a.ps_k = c.k; a.ps_d = c.d; a.ps_kstep = p.kstep;
  a.reflect = reflect; a.in_mode = io.in_mode; a.accumulate = io.accumulate;
  a.res_rows = io.res_rows; a.em_seg = io.em_seg; a.fm_sums = io.fm_sums; a.fm_gs = io.fm_gs;
  for (int i = 0; i < 4; ++i) a.em_map[i] = io.em_map[i];
  a.in_slope = io.in_slope; a.out_slope = io.out_slope; a.res_slope = io.res_slope; a.emask_slope = io.emask_slope;
  a.PLEN = p.PLEN; a.CSTRIDE = p.CSTRIDE;
  a.s_magic = p.S > 1 ? (unsigned)((0x100000000ull + p.S - 1) / p.S) : 0u;
  a.ntt = p.ntt; a.nmt = p.nmt; a.w_tile = p.w_tile;
  const long long nb = (long long)p.ntt * c.B * p.nph * p.nmt * p.G;
  if (nb <= 0 || nb > 0x7fffffffLL) return fail(EBEN_EINVAL, "thin grid of %lld blocks", nb);
  {
    auto magic = [](int d) { return d == 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); };
    int dmax = p.nph > p.ntt ? p.nph : p.ntt;
    if (c.B > dmax) dmax = c.B;
    if (p.nmt > dmax) dmax = p.nmt;
    a.id_fast = nb * dmax < 0x100000000LL ? 1 : 0;   // mulhi(n, ceil(2^32 / d)) is exact while n * d < 2^32
    a.m_ntt = magic(p.ntt); a.m_B = magic(c.B); a.m_nph = magic(p.nph); a.m_nmt = magic(p.nmt);
    a.pg_n = p.nph <= 8 ? p.nph : 0;
    a.tile = p.BN * p.NP;
    const int ad = p.dstep >= 0 ? p.dstep : -p.dstep;
    for (int ph = 0; ph < a.pg_n; ++ph) {
      const PhaseGeom q = phase_geom(p.mode, ph, p.J, p.off0, p.nt, p.dstep, p.OS, p.ps_pad, c.k, c.d, p.kstep, p.Ly);
      a.pg[ph].J = q.J; a.pg[ph].off0 = q.off0; a.pg[ph].minoff = q.minoff; a.pg[ph].nt = q.nt; a.pg[ph].oo = q.oo;
      a.pg[ph].span = q.J > 0 ? (a.tile - 1) * p.S + (q.J - 1) * ad + 1 : 0;
      a.pg[ph].span_magic = a.pg[ph].span > 0 ? (unsigned)((0x100000000ull + (unsigned)a.pg[ph].span - 1) / (unsigned)a.pg[ph].span) : 0u;
      a.pg[ph].pad = 0;
    }
  }
  if (p.BN == 256 && p.NP == 4) {
    switch (p.MT) {
      case 1: return launch_thin_cfg<1, 256, 4>(a, (int)nb, p.lds_bytes, st);
      case 4: return launch_thin_cfg<4, 256, 4>(a, (int)nb, p.lds_bytes, st);
      case 8: return launch_thin_cfg<8, 256, 4>(a, (int)nb, p.lds_bytes, st);
      default: return launch_thin_cfg<16, 256, 4>(a, (int)nb, p.lds_bytes, st);
    }
  }
  if (p.BN == 256) {
    switch (p.MT) {
      case 1: return launch_thin_cfg<1, 256>(a, (int)nb, p.lds_bytes, st);
      case 4: return launch_thin_cfg<4, 256>(a, (int)nb, p.lds_bytes, st);
      case 8: return launch_thin_cfg<8, 256>(a, (int)nb, p.lds_bytes, st);
      default: return launch_thin_cfg<16, 256>(a, (int)nb, p.lds_bytes, st);
    }
  }
  switch (p.MT) {
    case 1: return launch_thin_cfg<1, 128>(a, (int)nb, p.lds_bytes, st);
    case 4: return launch_thin_cfg<4, 128>(a, (int)nb, p.lds_bytes, st);
    case 8: return launch_thin_cfg<8, 128>(a, (int)nb, p.lds_bytes, st);
    default: return launch_thin_cfg<16, 128>(a, (int)nb, p.lds_bytes, st);
  }
}

}  // namespace eben
