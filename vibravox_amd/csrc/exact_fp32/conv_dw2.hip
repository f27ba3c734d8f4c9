// conv_dw2.hip -- second-generation weight-gradient kernel (gfx950).
//
//   dw[co, c, j] = sum_{b,t} A(b, co, t) * X(b, c, t*S + off0 + j*d)        (conv_dw.hip states the roles)
//
// GEMM per group: M = Cout/g rows, N = (c, j) columns (+ a "ones" column = bias gradient), K = (batch, time).
// The MFMA wants 32 different rows at one k per instruction while memory holds k contiguous per row, so
// the A operand has to be transposed somewhere.  Here that is a separate streaming pre-pass:
//   1. dw_pack_a_kernel: A (= dy * lrelu'(y), or lrelu(x) for a transposed conv) -> the exact LDS image
//      the MFMA A fragments are read from, [group][m-tile][batch][time chunk][t][lane32][FM], zero padded
//      in rows and time.  One extra read + write of the gradient (<10 % of the GEMM time) buys a main
//      loop with no transform, no bounds check and no VALU on the A side: chunks arrive by LDS-DMA
//      (global_load_lds_dwordx4), double-buffered, one barrier per 64-time-step chunk.
//   2. conv_dw2_kernel: v_mfma_f32_32x32x2_f32, block = 4 waves; the X tile (a few channels x receptive
//      span) is register-prefetched one chunk ahead with the fused input stage; split-K over blocks into
//      private slabs, reduced in a fixed order by eben_wn_bwd (deterministic, no float atomics).
#include "common.h"

#include <cstdlib>

namespace eben {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int DW2_BK = 64;     // time steps per K chunk
constexpr int DW2_XR = 12;     // X-tile elements a thread holds in flight (small tiles)
constexpr int DW2_XR_BIG = 32; // ... for wide X tiles (pointwise / strongly dilated convs: many channels per 128 columns)
constexpr int DW2_XR_MID = 20; // ... the same on the 96- / 128-row tiles (register budget)

// ---- pre-pass: transpose / mask / pad A into the LDS image ------------------------------------
template <int FM, int WAVES_M>
__global__ __launch_bounds__(256) void dw_pack_a_kernel(const Dw2Args P) {
  constexpr int FI = FM == 3 ? 4 : FM;
  constexpr int BM = WAVES_M * FM * 32;
  constexpr int BMI = WAVES_M * 32 * FI;       // floats per time step in the image
  constexpr int BK = DW2_BK;
  __shared__ float tile[BM][BK + 1];
  unsigned id = blockIdx.x;
  const int tc = id % P.nct; id /= P.nct;
  const int b = id % P.B; id /= P.B;
  const int mt = id % P.nmt;
  const int g = id / P.nmt;
  const int m0 = mt * BM, t0 = tc * BK;
  const long long abase = ((long long)b * P.Ca + (long long)g * P.Mg) * P.La;
  // eight independent loads in flight per thread (see dw3_pack_a_kernel)
  static_assert((BM * BK) % (8 * 256) == 0, "tile must split into batches of eight loads per thread");
  for (int base = 0; base < BM * BK; base += 8 * 256) {
    float v[8], mk[8];
    int ok[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * 256 + threadIdx.x;
      const int m = i / BK, t = i - m * BK;
      ok[u] = (int)(m0 + m < P.Mg) & (int)(t0 + t < P.La);
      const long long idx = ok[u] ? abase + (long long)(m0 + m) * P.La + t0 + t : abase;
      v[u] = P.a[idx];
      mk[u] = P.a_mode ? P.amask[idx] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * 256 + threadIdx.x;
      const int m = i / BK, t = i - m * BK;
      const float w = P.a_mode == 0 ? lrelu(v[u], P.a_slope) : v[u] * dlrelu(mk[u], P.a_slope);
      tile[m][t] = ok[u] ? w : 0.f;
    }
  }
  __syncthreads();
  float* dst = P.ap + ((((long long)g * P.nmt + mt) * P.B + b) * P.nct + tc) * (long long)(BK * BMI);
  for (int i = threadIdx.x; i < BK * BMI; i += 256) {
    const int t = i / BMI, e = i - t * BMI;
    const int wm = e / (32 * FI), m32 = (e / FI) & 31, fi = e % FI;
    dst[i] = fi < FM ? tile[wm * FM * 32 + fi * 32 + m32][t] : 0.f;
  }
}

template <int FI> struct DwAFrag;
template <> struct DwAFrag<1> { typedef float type; };
template <> struct DwAFrag<2> { typedef f32x2 type; };
template <> struct DwAFrag<4> { typedef f32x4 type; };
template <int FI>
__device__ __forceinline__ float dwa_elem(const typename DwAFrag<FI>::type& a, int i) { return a[i]; }
template <>
__device__ __forceinline__ float dwa_elem<1>(const float& a, int) { return a; }

// ---- main kernel ---------------------------------------------------------------------------------
// block = 4 waves as WAVES_M x WAVES_N; wave tile = FM x FN fragments of 32x32; BN = 128 columns always.
template <int FM, int FN, int WAVES_M, int XR>
__global__ __launch_bounds__(256, 2) void conv_dw2_kernel(const Dw2Args P) {
  constexpr int WAVES_N = 4 / WAVES_M;
  constexpr int FI = FM == 3 ? 4 : FM;
  constexpr int BM = WAVES_M * FM * 32;
  constexpr int BN = WAVES_N * FN * 32;
  constexpr int BMI = WAVES_M * 32 * FI;
  constexpr int BK = DW2_BK;
  constexpr int ACH = BK * BMI;                 // floats per A chunk
  constexpr int PIECES = ACH / 4 / 256;
  static_assert(BN == 128, "128 columns per block");
  static_assert(PIECES * 256 * 4 == ACH, "A chunk must split into whole LDS-DMA pieces");
  typedef typename DwAFrag<FI>::type afrag_t;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                               // 2 x ACH
  const int XT = P.nch_max * P.XSTR + 2;          // X tile floats (+ {0.f, 1.f} cells)
  float* Xs = smem + 2 * ACH;                     // 2 x XT

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  unsigned id = blockIdx.x;
  const int nti = __builtin_amdgcn_readfirstlane(id % P.nnt); id /= P.nnt;
  const int mt = __builtin_amdgcn_readfirstlane(id % P.nmt); id /= P.nmt;
  const int g = __builtin_amdgcn_readfirstlane(id % P.G);
  const int z = __builtin_amdgcn_readfirstlane(id / P.G);
  const int n0 = nti * BN, m0 = mt * BM;

  const int c_lo = n0 / P.J;
  int nch = (BN - 1) / P.J + 2;
  if (nch > P.Cg - c_lo) nch = P.Cg - c_lo;
  if (nch < 0) nch = 0;
  const int span = (BK - 1) * P.S + (P.J - 1) * P.d + 1;
  const int cell_zero = P.nch_max * P.XSTR, cell_one = cell_zero + 1;
  if (tid < 2) { Xs[cell_zero + tid] = (float)tid; Xs[XT + cell_zero + tid] = (float)tid; }

  // per-lane column geometry: B fragment n, k-lane kk reads Xs[xoff[n] + t * xstep[n]] at time t
  int xoff[FN], xstep[FN];
#pragma unroll
  for (int n = 0; n < FN; ++n) {
    const int col = n0 + (wn * FN + n) * 32 + (lane & 31);
    if (col < P.Ng) {
      const int c = col / P.J, j = col - c * P.J;
      xoff[n] = (c - c_lo) * P.XSTR + j * P.d + (lane >> 5) * P.S;
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

  // X-tile elements of this thread: (channel << 16 | position), resolved once
  const int xtot = nch * span;
  int xpk[XR];
#pragma unroll
  for (int u = 0; u < XR; ++u) {
    const int i = tid + u * 256;
    const int c = i / span;
    xpk[u] = i < xtot ? ((c << 16) | (i - c * span)) : -1;
  }
  float xreg[XR], mreg[XR];
  unsigned okmask = 0;
  auto fetch_x = [&](int q) {     // issue only: raw values stay in flight
    const int b = q / P.nct;
    const int t0 = (q - b * P.nct) * BK;
    const int qbase = t0 * P.S + P.off0;
    const long long xbase = ((long long)b * P.Cx + (long long)g * P.Cg + (c_lo < P.Cg ? c_lo : P.Cg - 1)) * P.Lx;   // a bias-only tile has no channel
    const float* px = P.x + xbase;
    const float* pm = P.xmask + xbase;
    okmask = 0;
#pragma unroll
    for (int u = 0; u < XR; ++u) {
      int p = qbase + (xpk[u] & 0xffff);
      const int m1 = p < 0 ? -p : p;
      const int m2 = m1 >= P.Lx ? 2 * (P.Lx - 1) - m1 : m1;
      p = P.reflect ? m2 : p;
      const int ok = (int)(xpk[u] >= 0) & (int)(p >= 0) & (int)(p < P.Lx);
      const int o = ok ? (xpk[u] >> 16) * P.Lx + p : 0;
      okmask |= (unsigned)ok << u;
      xreg[u] = px[o];
      if (P.x_mode != 0) mreg[u] = pm[o];
    }
  };
  auto store_x = [&](int buf) {
    float* dst = Xs + buf * XT;
#pragma unroll
    for (int u = 0; u < XR; ++u) {
      const float t = P.x_mode == 0 ? lrelu(xreg[u], P.x_slope) : xreg[u] * dlrelu(mreg[u], P.x_slope);
      const int sl = xpk[u] >= 0 ? (xpk[u] >> 16) * P.XSTR + (xpk[u] & 0xffff) : cell_zero;
      // lanes without an element rewrite the constant zero cell with zero
      dst[sl] = ((okmask >> u) & 1u) ? t : 0.f;
    }
  };
  const float* asrc = P.ap + (((long long)g * P.nmt + mt) * P.B) * P.nct * (long long)ACH;
  auto issue_a = [&](int q, int buf) {
    const float* src = asrc + (long long)q * ACH;
    float* dst = As + buf * ACH;
#pragma unroll
    for (int u = 0; u < PIECES; ++u)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (u * 256 + tid) * 4),
                                       (__attribute__((address_space(3))) void*)(dst + (u * 256 + (tid & ~63)) * 4), 16, 0, 0);
  };

  // ---- K loop over this block's chunks q = z, z + nsplit, ... ----
  int buf = 0;
  if (z < P.nchunks) {
    issue_a(z, 0);
    fetch_x(z);
    store_x(0);
    if (z + P.nsplit < P.nchunks) fetch_x(z + P.nsplit);
  }
  __syncthreads();
  for (int q = z; q < P.nchunks; q += P.nsplit) {
    const int qn = q + P.nsplit;
    if (qn < P.nchunks) issue_a(qn, buf ^ 1);
    const float* ab = As + buf * ACH + (lane >> 5) * BMI + wm * 32 * FI + (lane & 31) * FI;
    const float* xb = Xs + buf * XT;
#pragma unroll 4
    for (int s = 0; s < BK / 2; ++s) {
      const afrag_t a = *reinterpret_cast<const afrag_t*>(ab + s * 2 * BMI);
      float bv[FN];
#pragma unroll
      for (int n = 0; n < FN; ++n) bv[n] = xb[xoff[n] + s * 2 * xstep[n]];
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int n = 0; n < FN; ++n)
          acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(dwa_elem<FI>(a, i), bv[n], acc[i][n], 0, 0, 0);
    }
    if (qn < P.nchunks) {
      store_x(buf ^ 1);   // the other X buffer was last read one chunk ago (a barrier back)
      if (qn + P.nsplit < P.nchunks) fetch_x(qn + P.nsplit);
    }
    __syncthreads();
    buf ^= 1;
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
struct Dw2Plan {
  int ok;
  int Cg, Mg, G, J, Ng, row_stride, cfg, BM, BMI, nnt, nmt, nct, nchunks, nsplit, XSTR, nch_max, big_x;
  size_t lds_bytes, ap_floats;
  long long slab_stride;
};

static int dw2_env(const char* name, int dflt) {
  const char* s = getenv(name);
  return s ? atoi(s) : dflt;
}

static void make_dw2_plan(const Canon& c, Dw2Plan* p) {
  p->ok = 0;
  p->G = c.g; p->Cg = c.Cin / c.g; p->Mg = c.Cout / c.g; p->J = c.k;
  p->Ng = p->Cg * c.k; p->row_stride = p->Ng + 1;
  static const int enabled = dw2_env("EBEN_DW2", 1);
  static const int min_m = dw2_env("EBEN_DW2_MIN_M", 24);
  if (!enabled || p->Mg < min_m) return;
  // tile height with the least padded rows (larger on ties): 128 (2x2 waves), 96, 64, 32 (1x4 waves)
  const int cand[4] = {128, 96, 64, 32};
  int best = 0, best_pad = 1 << 30;
  for (int i = 0; i < 4; ++i) {
    const int pad = round_up(p->Mg, cand[i]);
    if (pad < best_pad) { best_pad = pad; best = cand[i]; }
  }
  p->BM = best;
  p->cfg = best == 128 ? 0 : best == 96 ? 1 : best == 64 ? 2 : 3;
  p->BMI = best == 96 ? 128 : best;
  p->nnt = ceil_div(p->row_stride, 128);
  p->nmt = ceil_div(p->Mg, p->BM);
  p->nch_max = 127 / c.k + 2;
  if (p->nch_max > p->Cg) p->nch_max = p->Cg;
  const int span = (DW2_BK - 1) * c.s + (c.k - 1) * c.d + 1;
  if ((long long)p->nch_max * span > (p->cfg <= 1 ? DW2_XR_MID : DW2_XR_BIG) * 256) return;   // X tile must fit the register prefetch
  p->big_x = (long long)p->nch_max * span > DW2_XR * 256;
  p->XSTR = span | 1;
  p->nct = ceil_div(c.Lout, DW2_BK);
  p->nchunks = c.B * p->nct;
  const int tiles = p->nnt * p->nmt * p->G;
  int ns = tiles >= 384 ? 1 : ceil_div(768, tiles);
  if (ns > 512) ns = 512;
  if (ns > p->nchunks) ns = p->nchunks;
  if (ns < 1) ns = 1;
  p->nsplit = ns;
  p->lds_bytes = 4ull * (2ull * DW2_BK * p->BMI + 2ull * ((size_t)p->nch_max * p->XSTR + 2));
  if (p->lds_bytes > 160 * 1024) return;
  p->slab_stride = (long long)c.Cout * p->row_stride;
  p->ap_floats = (size_t)p->G * p->nmt * c.B * p->nct * DW2_BK * p->BMI;
  p->ok = 1;
}

template <int FM, int FN, int WAVES_M, int XR>
static int launch_dw2(const Dw2Args& a, const Dw2Plan& p, hipStream_t st) {
  static LdsAttrOnce attr_once;
  auto kern = conv_dw2_kernel<FM, FN, WAVES_M, XR>;
  {
    const hipError_t e = lds_attr_once(attr_once, reinterpret_cast<const void*>(kern));
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(conv_dw2)");
  }
  const int npack = p.G * p.nmt * a.B * p.nct;
  hipLaunchKernelGGL((dw_pack_a_kernel<FM, WAVES_M>), dim3(npack), dim3(256), 0, st, a);
  EBEN_CHECK_LAUNCH("dw_pack_a_kernel");
  const int nb = p.nnt * p.nmt * p.G * p.nsplit;
  hipLaunchKernelGGL(kern, dim3(nb), dim3(256), p.lds_bytes, st, a);
  EBEN_CHECK_LAUNCH("conv_dw2_kernel");
  return EBEN_OK;
}

int dw2_applicable(const Canon& c) {
  Dw2Plan p;
  make_dw2_plan(c, &p);
  return p.ok;
}

size_t dw2_workspace(const Canon& c, int* nslab, int* row_stride) {
  Dw2Plan p;
  make_dw2_plan(c, &p);
  if (!p.ok) return 0;
  if (nslab) *nslab = p.nsplit;
  if (row_stride) *row_stride = p.row_stride;
  return sizeof(float) * ((size_t)p.slab_stride * p.nsplit + p.ap_floats);
}

int dw2_launch(const Canon& c, Dw2Args a, float* workspace, size_t ws_bytes, hipStream_t st) {
  Dw2Plan p;
  make_dw2_plan(c, &p);
  if (!p.ok) return fail(EBEN_EUNSUPPORTED, "dw2_launch on a layer the second-generation kernel does not cover");
  const size_t need = sizeof(float) * ((size_t)p.slab_stride * p.nsplit + p.ap_floats);
  if (ws_bytes < need) return fail(EBEN_EWORKSPACE, "bwd_dw needs %zu workspace bytes, got %zu", need, ws_bytes);
  a.slabs = workspace;
  a.ap = workspace + (size_t)p.slab_stride * p.nsplit;
  a.B = c.B; a.G = c.g; a.Cg = p.Cg; a.Mg = p.Mg; a.Ca = c.Cout; a.Cx = c.Cin; a.La = c.Lout; a.Lx = c.Lin;
  a.S = c.s; a.d = c.d; a.off0 = -c.pl; a.J = c.k; a.Ng = p.Ng; a.row_stride = p.row_stride; a.reflect = c.reflect;
  a.nsplit = p.nsplit; a.nct = p.nct; a.nchunks = p.nchunks; a.nnt = p.nnt; a.nmt = p.nmt; a.XSTR = p.XSTR; a.nch_max = p.nch_max;
  a.slab_stride = p.slab_stride;
  if (p.big_x) {
    switch (p.cfg) {
      case 0: return launch_dw2<2, 2, 2, DW2_XR_MID>(a, p, st);
      case 1: return launch_dw2<3, 1, 1, DW2_XR_MID>(a, p, st);
      case 2: return launch_dw2<2, 1, 1, DW2_XR_BIG>(a, p, st);
      default: return launch_dw2<1, 1, 1, DW2_XR_BIG>(a, p, st);
    }
  }
  switch (p.cfg) {
    case 0: return launch_dw2<2, 2, 2, DW2_XR>(a, p, st);
    case 1: return launch_dw2<3, 1, 1, DW2_XR>(a, p, st);
    case 2: return launch_dw2<2, 1, 1, DW2_XR>(a, p, st);
    default: return launch_dw2<1, 1, 1, DW2_XR>(a, p, st);
  }
}

}  // namespace eben
