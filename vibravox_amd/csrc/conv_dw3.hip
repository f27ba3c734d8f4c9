// conv_dw3.hip -- bf16-operand weight-gradient kernel (gfx950), EbenConv1dDesc.math == EBEN_MATH_BF16.
//
//   dw[co, c, j] = sum_{b,t} A(b, co, t) * X(b, c, t*S + off0 + j*d)
//
// GEMM per group as in conv_dw2.hip (M = Cout/g rows, N = (c, j) columns + the "ones" column, K = (batch, time)),
// on v_mfma_f32_32x32x16_bf16.  A bf16 MFMA operand wants 8 CONSECUTIVE k per lane.  Along time the X operand
// of a strided / dilated conv is a strided, unaligned gather; along the BATCH it is not: a k-step here is one
// time step x 16 batch items, so a lane's 8 k are 8 batch items at one (channel, position) -- staged once into
// a 16-byte LDS unit and read back with ONE aligned ds_read_b128 for any stride / dilation / tap.
//   1. dw3_pack_a_kernel: A (fp32 gradient) -> bf16 image [group][m-tile][batch group][time][row tile][lane]
//      of 16-byte units (8 batch items each), zero padded in rows, time and batch.
//   2. conv_dw3_kernel: block = 4 waves, 128 columns; A sub-chunks of 4 time steps arrive by LDS-DMA
//      (double-buffered, one barrier each); the X tile of a (batch group, time chunk) -- [8-item half][channel]
//      [position] units -- is register-prefetched one chunk ahead and converted as it is written to LDS;
//      split-K over blocks into private fp32 slabs reduced in a fixed order by eben_wn_bwd.
#include "common.h"

#include <cstdlib>

namespace eben {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned dw3_pack_bf16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

constexpr int DW3_KSC = 4;   // time steps (k-steps of 16 batch items) per A sub-chunk

struct Dw3Args {
  const float* a; const float* amask; int a_mode; float a_slope;   // a_mode 1: a * lrelu'(amask)
  const float* x; float x_slope;
  u32x4* ap;
  float* slabs;
  int B, G, Cg, Mg, Ca, Cx, La, Lx;
  int S, d, off0, J, Ng, has_bias, row_stride, reflect;
  int nsplit, nct, nbg, nchunks, nnt, nmt, XSTR, HS, nch_max, BKT;
  long long slab_stride;
};

// ---- pre-pass: 8 batch items (one operand half) x 32 rows x 32 time steps per block ------------------
// reads: 128-byte row segments; writes: 512 contiguous bytes (32 lanes of one half) per (time step, row tile)
template <int MT>
__global__ __launch_bounds__(256) void dw3_pack_a_kernel(const Dw3Args P) {
  __shared__ float tile[8][32][33];
  unsigned id = blockIdx.x;
  const int nt32 = (P.nct * P.BKT + 31) / 32;   // a 32-step tile may span two 16-step chunks: the image is linear in time per batch group
  const int tg = id % nt32; id /= nt32;
  const int half = id & 1; id >>= 1;
  const int bg = id % P.nbg; id /= P.nbg;
  const int m32 = id % (P.nmt * MT);
  const int g = id / (P.nmt * MT);
  const int t0 = tg * 32;
  // eight independent loads in flight per thread (a load inside the bounds check is issued, waited for and stored one at a
  // time: 32 exposed memory round trips per block, 6-13x the time the 48 KB a block moves should take)
  // element k of this thread: i = tid + 256 k  ->  time t = tid & 31, row m = (tid >> 5) + 8 (k & 3), batch item k >> 2: one 64-bit
  // index per thread, the rest are block-uniform strides (the general form cost ~12 VALU instructions per element)
  const int tl = threadIdx.x & 31, ml = threadIdx.x >> 5;
  const int bb0 = bg * 16 + half * 8, mm0 = m32 * 32 + ml;
  const long long rowstride = P.La, bstride = (long long)P.Ca * P.La;
  const long long idx0 = ((long long)bb0 * P.Ca + (long long)g * P.Mg + mm0) * P.La + t0 + tl;
  const bool t_ok = t0 + tl < P.La;
  for (int base = 0; base < 32; base += 8) {
    float v[8], mk[8];
    int ok[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = base + u;
      const int b = k >> 2, m = 8 * (k & 3);
      ok[u] = (int)(bb0 + b < P.B) & (int)(mm0 + m < P.Mg) & (int)t_ok;
      const long long idx = ok[u] ? idx0 + b * bstride + m * rowstride : 0;
      v[u] = P.a[idx];
      mk[u] = P.a_mode ? P.amask[idx] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = base + u;
      const float w = P.a_mode == 0 ? (P.a_slope == 1.f ? v[u] : lrelu(v[u], P.a_slope)) : v[u] * dlrelu(mk[u], P.a_slope);
      tile[k >> 2][ml + 8 * (k & 3)][tl] = ok[u] ? w : 0.f;
    }
  }
  __syncthreads();
  const int mt = m32 / MT, fm = m32 - mt * MT;
  const int tc = t0 / P.BKT, tin0 = t0 - tc * P.BKT;
  u32x4* dst = P.ap + ((((long long)g * P.nmt + mt) * P.nchunks + (long long)bg * P.nct + tc) * P.BKT + tin0) * (MT * 64);
  for (int u = threadIdx.x; u < 32 * 32; u += 256) {
    const int t = u >> 5, m = u & 31;
    if (t0 + t >= P.nct * P.BKT) continue;
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = dw3_pack_bf16(tile[2 * e][m][t], tile[2 * e + 1][m][t]);
    dst[((long long)t * MT + fm) * 64 + half * 32 + m] = o;
  }
}

// ---- main kernel ---------------------------------------------------------------------------------
// SP (EBEN_MATH_BF16X2): the X operand -- the layer's activations -- is staged as hi = bf16(x) and lo = bf16(x - hi) tiles and
// every k-step issues two MFMAs per fragment pair against the same gradient fragment.
template <int FM, int FN, int WAVES_M, int XRB, bool SP = false>
__global__ __launch_bounds__(256, 2) void conv_dw3_kernel(const Dw3Args P) {
  constexpr int WAVES_N = 4 / WAVES_M;
  constexpr int MT = WAVES_M * FM;
  constexpr int BM = MT * 32;
  constexpr int BN = WAVES_N * FN * 32;
  constexpr int KSC = DW3_KSC;
  constexpr int ACHU = KSC * MT * 64;           // units per A sub-chunk
  constexpr int PIECES = ACHU / 256;
  static_assert(BN == 128, "128 columns per block");
  static_assert(PIECES * 256 == ACHU, "A sub-chunk must split into whole LDS-DMA pieces");

  extern __shared__ __attribute__((aligned(16))) u32x4 smem_dw3[];
  u32x4* As = smem_dw3;                          // 2 x ACHU
  const int XT = 2 * P.HS + 2;                   // X tile units (+ {zero, one} cells)
  u32x4* Xs = smem_dw3 + 2 * ACHU;               // 2 x XT (SP: 2 x XT hi, then 2 x XT lo)
  const int LO = 2 * XT;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  unsigned id = blockIdx.x;
  const int nti = __builtin_amdgcn_readfirstlane(id % P.nnt); id /= P.nnt;
  const int mt = __builtin_amdgcn_readfirstlane(id % P.nmt); id /= P.nmt;
  const int g = __builtin_amdgcn_readfirstlane(id % P.G);
  const int z = __builtin_amdgcn_readfirstlane(id / P.G);
  const int n0 = nti * BN, m0 = mt * BM;
  const int BKT = P.BKT, NSUB = BKT / KSC;

  const int c_lo = n0 / P.J;
  int nch = (BN - 1) / P.J + 2;
  if (nch > P.Cg - c_lo) nch = P.Cg - c_lo;
  if (nch < 0) nch = 0;
  const int span = (BKT - 1) * P.S + (P.J - 1) * P.d + 1;
  const int cell_zero = 2 * P.HS, cell_one = cell_zero + 1;
  if (tid < 4) {
    const unsigned v = (tid & 1) ? 0x3f803f80u : 0u;   // bf16 1.0 pairs
    Xs[(tid >> 1) * XT + cell_zero + (tid & 1)] = u32x4{v, v, v, v};
    if constexpr (SP) Xs[LO + (tid >> 1) * XT + cell_zero + (tid & 1)] = u32x4{0u, 0u, 0u, 0u};   // lo of the constants 0 and 1
  }

  // per-lane column geometry: B fragment n reads unit Xs[xoff[n] + t * xstep[n]] at time step t of the chunk
  int xoff[FN], xstep[FN];
#pragma unroll
  for (int n = 0; n < FN; ++n) {
    const int col = n0 + (wn * FN + n) * 32 + (lane & 31);
    if (col < P.Ng) {
      const int c = col / P.J, j = col - c * P.J;
      xoff[n] = (lane >> 5) * P.HS + (c - c_lo) * P.XSTR + j * P.d;
      xstep[n] = P.S;
    } else {
      xoff[n] = (col == P.Ng && P.has_bias) ? cell_one : cell_zero;
      xstep[n] = 0;
    }
  }

  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int n = 0; n < FN; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][n][r] = 0.f;

  // X-tile units of this thread: (half << 30 | channel << 16 | position), resolved once
  const int xhalf = nch * span, xtot = 2 * xhalf;
  int xpk[XRB];
#pragma unroll
  for (int u = 0; u < XRB; ++u) {
    const int i = tid + u * 256;
    const int h = i >= xhalf ? 1 : 0;
    const int r = i - h * xhalf;
    const int c = r / span;
    xpk[u] = i < xtot ? ((h << 30) | (c << 16) | (r - c * span)) : -1;
  }
  float xreg[XRB][8];
  unsigned okmask = 0;
  const long long bstride = (long long)P.Cx * P.Lx;
  auto fetch_x = [&](int q) {     // issue only: raw values stay in flight
    const int bg = q / P.nct;
    const int t0 = (q - bg * P.nct) * BKT;
    const int qbase = t0 * P.S + P.off0;
    const float* px = P.x + ((long long)g * P.Cg + (c_lo < P.Cg ? c_lo : P.Cg - 1)) * P.Lx;
    okmask = 0;
#pragma unroll
    for (int u = 0; u < XRB; ++u) {
      int p = qbase + (xpk[u] & 0xffff);
      const int m1 = p < 0 ? -p : p;
      const int m2 = m1 >= P.Lx ? 2 * (P.Lx - 1) - m1 : m1;
      p = P.reflect ? m2 : p;
      const int ok = (int)(xpk[u] >= 0) & (int)(p >= 0) & (int)(p < P.Lx);
      const int o = ok ? ((xpk[u] >> 16) & 0x3fff) * P.Lx + p : 0;
      okmask |= (unsigned)ok << u;
      const int b0 = bg * 16 + ((xpk[u] >> 30) & 1) * 8;
      if (bg * 16 + 16 <= P.B) {
        // whole group of 16 batch items (block-uniform): one 64-bit base per unit, then strides -- no clamps
        const float* p8 = px + (long long)b0 * bstride + o;
#pragma unroll
        for (int e = 0; e < 8; ++e) xreg[u][e] = p8[(long long)e * bstride];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int bb = b0 + e < P.B ? b0 + e : P.B - 1;
          xreg[u][e] = px[(long long)bb * bstride + o];
        }
      }
    }
  };
  const bool plain_x = P.x_slope == 1.f;   // the engine's launches: no activation on load (block-uniform)
  auto store_x = [&](int q, int buf) {
    const int bg = q / P.nct;
    u32x4* dst = Xs + buf * XT;
#pragma unroll
    for (int u = 0; u < XRB; ++u) {
      const int ok = (int)((okmask >> u) & 1u);
      const int b0 = bg * 16 + ((xpk[u] >> 30) & 1) * 8;
      float t[8];
      if (plain_x && bg * 16 + 16 <= P.B) {
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = ok ? xreg[u][e] : 0.f;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = (ok && b0 + e < P.B) ? (plain_x ? xreg[u][e] : lrelu(xreg[u][e], P.x_slope)) : 0.f;
      }
      u32x4 o;
      o[0] = dw3_pack_bf16(t[0], t[1]); o[1] = dw3_pack_bf16(t[2], t[3]); o[2] = dw3_pack_bf16(t[4], t[5]); o[3] = dw3_pack_bf16(t[6], t[7]);
      // lanes without a unit rewrite the constant zero cell with zero
      const int sl = xpk[u] >= 0 ? ((xpk[u] >> 30) & 1) * P.HS + ((xpk[u] >> 16) & 0x3fff) * P.XSTR + (xpk[u] & 0xffff) : cell_zero;
      dst[sl] = o;
      if constexpr (SP) {
        u32x4 l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float h0 = __builtin_bit_cast(float, o[e] << 16), h1 = __builtin_bit_cast(float, o[e] & 0xffff0000u);
          l[e] = dw3_pack_bf16(t[2 * e] - h0, t[2 * e + 1] - h1);
        }
        dst[LO + sl] = l;
      }
    }
  };
  const u32x4* asrc = P.ap + (((long long)g * P.nmt + mt) * P.nchunks) * (long long)BKT * (MT * 64);
  auto issue_a = [&](int q, int sub, int buf) {
    const u32x4* src = asrc + ((long long)q * BKT + sub * KSC) * (MT * 64);
    u32x4* dst = As + buf * ACHU;
#pragma unroll
    for (int u = 0; u < PIECES; ++u)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (u * 256 + tid)),
                                       (__attribute__((address_space(3))) void*)(dst + (u * 256 + (tid & ~63))), 16, 0, 0);
  };

  // ---- K loop over this block's chunks q = z, z + nsplit, ... ----
  int xbuf = 0, abuf = 0;
  if (z < P.nchunks) {
    issue_a(z, 0, 0);
    fetch_x(z);
    store_x(z, 0);
  }
  __syncthreads();
  for (int q = z; q < P.nchunks; q += P.nsplit) {
    const int qn = q + P.nsplit;
    const bool more = qn < P.nchunks;
    if (more) fetch_x(qn);
    const u32x4* xb = Xs + xbuf * XT;
    for (int sub = 0; sub < NSUB; ++sub) {
      if (sub + 1 < NSUB) issue_a(q, sub + 1, abuf ^ 1);
      else if (more) issue_a(qn, 0, abuf ^ 1);
      const u32x4* ab = As + abuf * ACHU + wm * FM * 64 + lane;
      u32x4 av[KSC][FM], bv[KSC][FN], bl[SP ? KSC : 1][FN];
      auto rd = [&](int ks) {
        const int t = sub * KSC + ks;
#pragma unroll
        for (int i = 0; i < FM; ++i) av[ks][i] = ab[(ks * MT + i) * 64];
#pragma unroll
        for (int n = 0; n < FN; ++n) {
          bv[ks][n] = xb[xoff[n] + t * xstep[n]];
          if constexpr (SP) bl[ks][n] = xb[LO + xoff[n] + t * xstep[n]];
        }
      };
      rd(0);
      rd(1);
#pragma unroll
      for (int ks = 0; ks < KSC; ++ks) {
        if (ks + 2 < KSC) rd(ks + 2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int n = 0; n < FN; ++n)
            acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[ks][i]), __builtin_bit_cast(bf16x8, bv[ks][n]),
                                                                acc[i][n], 0, 0, 0);
        if constexpr (SP) {
#pragma unroll
          for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int n = 0; n < FN; ++n)
              acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[ks][i]), __builtin_bit_cast(bf16x8, bl[ks][n]),
                                                                  acc[i][n], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (sub == NSUB - 1 && more) store_x(qn, xbuf ^ 1);   // the other X buffer was last read a whole chunk ago
      __syncthreads();
      abuf ^= 1;
    }
    xbuf ^= 1;
  }

  // ---- epilogue: slab[z][g*Mg + m][col] ----
  float* slab = P.slabs + (long long)z * P.slab_stride;
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int n = 0; n < FN; ++n) {
      const int col = n0 + (wn * FN + n) * 32 + (lane & 31);
      if (col >= P.row_stride) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * FM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < P.Mg) slab[((long long)g * P.Mg + m) * P.row_stride + col] = acc[i][n][r];
      }
    }
}

// ---------------------------------------------------------------------------------------------------
struct Dw3Plan {
  int ok;
  int Cg, Mg, G, J, Ng, row_stride, cfg, BM, MT, nnt, nmt, nct, nbg, nchunks, nsplit, XSTR, HS, nch_max, BKT, xrb, split;
  size_t lds_bytes, ap_units;
  long long slab_stride;
};

static int dw3_env(const char* name, int dflt) {
  const char* s = getenv(name);
  return s ? atoi(s) : dflt;
}

static void make_dw3_plan(const Canon& c, Dw3Plan* p) {
  p->ok = 0;
  p->G = c.g; p->Cg = c.Cin / c.g; p->Mg = c.Cout / c.g; p->J = c.k;
  p->Ng = p->Cg * c.k; p->row_stride = p->Ng + 1;
  p->split = c.xsplit_dir == 0;   // Conv1d under EBEN_MATH_BF16X2: the X operand is the activation tensor
  static const int enabled = dw3_env("EBEN_DW3", 1);
  static const int min_m = dw3_env("EBEN_DW3_MIN_M", 4);
  static const int min_n = dw3_env("EBEN_DW3_MIN_N", 12);
  // split operands on BOTH sides (EBEN_MATH_BF16X3 / X6) are a tap-conv form: those weight gradients take the exact-fp32 kernels
  if (!enabled || !c.bf16 || c.np > 1 || p->Mg < min_m || p->Ng < min_n || c.B < 8) return;
  const int cand[4] = {128, 96, 64, 32};
  int best = 0, best_pad = 1 << 30;
  for (int i = 0; i < 4; ++i) {
    const int pad = round_up(p->Mg, cand[i]);
    if (pad < best_pad) { best_pad = pad; best = cand[i]; }
  }
  p->BM = best;
  p->MT = best / 32;
  p->cfg = best == 128 ? 0 : best == 96 ? 1 : best == 64 ? 2 : 3;
  p->nnt = ceil_div(p->row_stride, 128);
  p->nmt = ceil_div(p->Mg, p->BM);
  p->nch_max = 127 / c.k + 2;
  if (p->nch_max > p->Cg) p->nch_max = p->Cg;
  const size_t a_bytes = 2ull * DW3_KSC * p->MT * 64 * 16;
  int chosen = 0;
  for (int pass = 0; pass < 2 && !chosen; ++pass) {
    static const size_t budget0 = (size_t)(getenv("EBEN_DW3_LDS_KB") ? atoi(getenv("EBEN_DW3_LDS_KB")) : 48) * 1024;   // [MI355X] in-step 18.8 -> 18.7 ms at 48 KB (co-residency with the other streams)
    const size_t budget = pass == 0 ? budget0 : 156 * 1024;   // two (or more) blocks per CU, else one
    for (int bkt : {32, 16, 8, 4}) {   // 8, 4: wide X tiles (pointwise convs over 128 channels, dilation 9): fewer time steps per chunk
      if (pass == 1 && bkt == 32) continue;
      const int span = (bkt - 1) * c.s + (c.k - 1) * c.d + 1;
      // X tile must fit the register prefetch (split operand: four units per thread -- the eight-unit kernels have no registers left for the lo fragments)
      if (span > 0xffff || 2 * p->nch_max * span > (p->split ? 4 : 8) * 256) continue;
      int xstr = span + (((c.k - span) % 16) + 16) % 16;                 // rows continue the column sequence mod 16 units
      const size_t lds = a_bytes + 2ull * (2ull * p->nch_max * xstr + 2) * 16 * (p->split ? 2 : 1);
      if (lds > budget) continue;
      p->BKT = bkt; p->XSTR = xstr; p->HS = p->nch_max * xstr; p->lds_bytes = lds;
      p->xrb = 2 * p->nch_max * span > 4 * 256 ? 8 : 4;
      chosen = 1;
      break;
    }
  }
  if (!chosen) return;
  p->nct = ceil_div(c.Lout, p->BKT);
  p->nbg = ceil_div(c.B, 16);
  p->nchunks = p->nbg * p->nct;
  const int tiles = p->nnt * p->nmt * p->G;
  int ns = tiles >= 384 ? 1 : ceil_div(768, tiles);
  if (ns > 512) ns = 512;
  if (ns > p->nchunks) ns = p->nchunks;
  if (ns < 1) ns = 1;
  p->nsplit = ns;
  p->slab_stride = (long long)c.Cout * p->row_stride;
  p->ap_units = (size_t)p->G * p->nmt * p->nchunks * p->BKT * p->MT * 64;
  p->ok = 1;
}

template <int FM, int FN, int WAVES_M, int XRB, bool SP>
static int launch_dw3_sp(const Dw3Args& a, const Dw3Plan& p, hipStream_t st) {
  static LdsAttrOnce attr_once;
  auto kern = conv_dw3_kernel<FM, FN, WAVES_M, XRB, SP>;
  {
    const hipError_t e = lds_attr_once(attr_once, reinterpret_cast<const void*>(kern));
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(conv_dw3)");
  }
  constexpr int MT = WAVES_M * FM;
  const int npack = p.G * p.nmt * MT * p.nbg * 2 * ((p.nct * p.BKT + 31) / 32);
  hipLaunchKernelGGL((dw3_pack_a_kernel<MT>), dim3(npack), dim3(256), 0, st, a);
  EBEN_CHECK_LAUNCH("dw3_pack_a_kernel");
  const int nb = p.nnt * p.nmt * p.G * p.nsplit;
  hipLaunchKernelGGL(kern, dim3(nb), dim3(256), p.lds_bytes, st, a);
  EBEN_CHECK_LAUNCH("conv_dw3_kernel");
  return EBEN_OK;
}

template <int FM, int FN, int WAVES_M, int XRB>
static int launch_dw3(const Dw3Args& a, const Dw3Plan& p, hipStream_t st) {
  if constexpr (XRB <= 4) {
    if (p.split) return launch_dw3_sp<FM, FN, WAVES_M, XRB, true>(a, p, st);
  }
  return launch_dw3_sp<FM, FN, WAVES_M, XRB, false>(a, p, st);
}

int dw3_applicable(const Canon& c) {
  Dw3Plan p;
  make_dw3_plan(c, &p);
  return p.ok;
}

size_t dw3_workspace(const Canon& c, int* nslab, int* row_stride) {
  Dw3Plan p;
  make_dw3_plan(c, &p);
  if (!p.ok) return 0;
  if (nslab) *nslab = p.nsplit;
  if (row_stride) *row_stride = p.row_stride;
  return sizeof(float) * (size_t)p.slab_stride * p.nsplit + 16 * p.ap_units + 16;
}

int dw3_launch(const Canon& c, const float* a, const float* amask, float a_slope, const float* x, float x_slope, int has_bias, float* workspace,
               size_t ws_bytes, hipStream_t st) {
  Dw3Plan p;
  make_dw3_plan(c, &p);
  if (!p.ok) return fail(EBEN_EUNSUPPORTED, "dw3_launch on a layer the bf16 weight-gradient kernel does not cover");
  const size_t slab_bytes = sizeof(float) * (size_t)p.slab_stride * p.nsplit;
  const size_t need = slab_bytes + 16 * p.ap_units + 16;
  if (ws_bytes < need) return fail(EBEN_EWORKSPACE, "bwd_dw needs %zu workspace bytes, got %zu", need, ws_bytes);
  Dw3Args k;
  k.a = a; k.amask = amask; k.a_mode = amask ? 1 : 0; k.a_slope = a_slope; k.x = x; k.x_slope = x_slope;
  k.slabs = workspace;
  k.ap = reinterpret_cast<u32x4*>((reinterpret_cast<uintptr_t>(workspace) + slab_bytes + 15) & ~(uintptr_t)15);
  k.B = c.B; k.G = c.g; k.Cg = p.Cg; k.Mg = p.Mg; k.Ca = c.Cout; k.Cx = c.Cin; k.La = c.Lout; k.Lx = c.Lin;
  k.S = c.s; k.d = c.d; k.off0 = -c.pl; k.J = c.k; k.Ng = p.Ng; k.has_bias = has_bias; k.row_stride = p.row_stride; k.reflect = c.reflect;
  k.nsplit = p.nsplit; k.nct = p.nct; k.nbg = p.nbg; k.nchunks = p.nchunks; k.nnt = p.nnt; k.nmt = p.nmt;
  k.XSTR = p.XSTR; k.HS = p.HS; k.nch_max = p.nch_max; k.BKT = p.BKT;
  k.slab_stride = p.slab_stride;
  if (p.xrb == 8) {
    switch (p.cfg) {
      case 0: return launch_dw3<2, 2, 2, 8>(k, p, st);
      case 1: return launch_dw3<3, 1, 1, 8>(k, p, st);
      case 2: return launch_dw3<2, 1, 1, 8>(k, p, st);
      default: return launch_dw3<1, 1, 1, 8>(k, p, st);
    }
  }
  switch (p.cfg) {
    case 0: return launch_dw3<2, 2, 2, 4>(k, p, st);
    case 1: return launch_dw3<3, 1, 1, 4>(k, p, st);
    case 2: return launch_dw3<2, 1, 1, 4>(k, p, st);
    default: return launch_dw3<1, 1, 1, 4>(k, p, st);
  }
}

}  // namespace eben
