// conv_dw.hip -- weight gradients of Conv1d / ConvTranspose1d and the weight-norm kernels.
//
//   dw[co, c, j] = sum_{b,t} A(b, co, t) * X(b, c, t*S + off0 + j*d)
//
// GEMM view per group: M = Cout/g rows, N = (c, j) flattened (+ one "ones" column that yields the
// bias gradient for free), K = (batch, time) -- a 10^4..10^6-long reduction, so the K range is
// split over `nsplit` blocks that each write a private slab; eben_wn_bwd sums the slabs in a fixed
// order (deterministic, no float atomics) and applies the weight-norm chain rule
// (torch_modules/utils.py:4-9 -> torch._weight_norm backward).
// A = dy * lrelu'(y) (Conv1d) or lrelu(x) (ConvTranspose1d); X = lrelu(x) or dy * lrelu'(y).
#include "common.h"

#include <cstdlib>

namespace eben {

struct DwArgs {
  const float* a; const float* amask; int a_mode; float a_slope;
  const float* x; const float* xmask; int x_mode; float x_slope;
  float* slabs;
  int B, G, Cg, Mg, Ca, Cx, La, Lx;
  int S, d, off0, J, Ng, has_bias, row_stride, reflect;
  int nsplit, nct, nchunks, nnt, nmt, XSTR;
  int bk;  // time steps per K-chunk: 32, 64 or 128 (largest whose X tile fits the register prefetch)
  long long slab_stride;
};

constexpr int DW_BK_MAX = 128;
constexpr int DW_XCAP = 5120;  // X-tile elements the register prefetch can hold (per block)

__device__ __forceinline__ float load_op(const float* p, const float* mask, long long idx, int mode, float slope) {
  float v = p[idx];
  return mode == 0 ? lrelu(v, slope) : v * dlrelu(mask[idx], slope);
}

template <int WAVES_M, int WAVES_N, int FM, int FN>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64) void conv_dw_kernel(const DwArgs P) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int BM = WAVES_M * FM * 16;
  constexpr int BN = WAVES_N * FN * 16;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int BK = P.bk;
  const int DSTR = BK + 2;           // 16 rows x 2 k-lanes of a half-wave -> 32 distinct banks
  float* Ds = smem;                  // [BM][DSTR]
  float* Xs = Ds + BM * DSTR;        // [nch][XSTR] then {0.f, 1.f}

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kk = lane >> 4;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  unsigned id = blockIdx.x;
  const int nti = id % P.nnt; id /= P.nnt;
  const int mt = id % P.nmt; id /= P.nmt;
  const int g = id % P.G;
  const int z = id / P.G;
  const int n0 = nti * BN, m0 = mt * BM;

  const int c_lo = n0 / P.J;
  int nch = (BN - 1) / P.J + 2;
  if (nch > P.Cg - c_lo) nch = P.Cg - c_lo;
  if (nch < 0) nch = 0;
  const int span = (BK - 1) * P.S + (P.J - 1) * P.d + 1;
  const int cell_zero = nch * P.XSTR, cell_one = cell_zero + 1;
  if (tid == 0) { Xs[cell_zero] = 0.f; Xs[cell_one] = 1.f; }

  int xoff[FN], xstep[FN];
#pragma unroll
  for (int n = 0; n < FN; ++n) {
    const int col = n0 + wn * FN * 16 + n * 16 + l15;
    if (col < P.Ng) {
      const int c = col / P.J, j = col - c * P.J;
      xoff[n] = (c - c_lo) * P.XSTR + j * P.d + kk * P.S;
      xstep[n] = P.S;
    } else {
      xoff[n] = (col == P.Ng && P.has_bias) ? cell_one : cell_zero;
      xstep[n] = 0;
    }
  }

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int n = 0; n < FN; ++n) acc[i][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Software pipeline: the global loads of chunk q+1 (A tile: 16 values per thread; X tile: up to
  // XR per thread) are issued before the MFMAs of chunk q and land in registers while they run;
  // they are written to LDS after the barrier that retires chunk q.
  constexpr int AR = BM * DW_BK_MAX / NT;    // register slots for the largest chunk
  constexpr int XR = DW_XCAP / NT;
  const int a_rows_per_pass = NT / BK;
  const int a_cnt = BM / a_rows_per_pass;     // slots actually used (<= AR)
  const int xtot = nch * span;
  const bool xfits = xtot <= XR * NT;
  float areg[AR], xreg[XR];

  // per-thread coordinates, computed once: A element u sits at row a_r0 + 8u, column a_tt;
  // X element u at (channel, position) packed as c<<16 | r (or -1 past the tile).
  const int a_tt = tid & (BK - 1), a_r0 = tid / BK;
  int xpk[XR];
#pragma unroll
  for (int u = 0; u < XR; ++u) {
    const int i = tid + u * NT;
    if (xfits && i < xtot) {
      const int c = i / span;
      xpk[u] = (c << 16) | (i - c * span);
    } else {
      xpk[u] = -1;
    }
  }

  auto load_chunk = [&](int q) {
    const int b = q / P.nct;
    const int t0 = (q - b * P.nct) * BK;
    const long long abase = ((long long)b * P.Ca + (long long)g * P.Mg) * P.La + t0;
    const float* pa = P.a + abase;
    const float* pam = P.amask ? P.amask + abase : nullptr;
    const bool tok = t0 + a_tt < P.La;
#pragma unroll
    for (int u = 0; u < AR; ++u) {
      const int m = m0 + a_r0 + a_rows_per_pass * u;
      float v = 0.f;
      if (u < a_cnt && tok && m < P.Mg) {
        const int off = m * P.La + a_tt;
        v = pa[off];
        v = P.a_mode == 0 ? lrelu(v, P.a_slope) : v * dlrelu(pam[off], P.a_slope);
      }
      areg[u] = v;
    }
    if (xfits) {
      const int qbase = t0 * P.S + P.off0;
      const long long xbase = ((long long)b * P.Cx + (long long)g * P.Cg + c_lo) * P.Lx;
      const float* px = P.x + xbase;
      const float* pxm = P.xmask ? P.xmask + xbase : nullptr;
#pragma unroll
      for (int u = 0; u < XR; ++u) {
        float v = 0.f;
        if (xpk[u] >= 0) {
          int p = qbase + (xpk[u] & 0xffff);
          if (P.reflect) {
            p = p < 0 ? -p : p;
            p = p >= P.Lx ? 2 * (P.Lx - 1) - p : p;
          }
          if (p >= 0 && p < P.Lx) {
            const int off = (xpk[u] >> 16) * P.Lx + p;
            v = px[off];
            v = P.x_mode == 0 ? lrelu(v, P.x_slope) : v * dlrelu(pxm[off], P.x_slope);
          }
        }
        xreg[u] = v;
      }
    }
  };
  auto store_chunk = [&](int q) {
#pragma unroll
    for (int u = 0; u < AR; ++u)
      if (u < a_cnt) Ds[(a_r0 + a_rows_per_pass * u) * DSTR + a_tt] = areg[u];
    if (xfits) {
#pragma unroll
      for (int u = 0; u < XR; ++u)
        if (xpk[u] >= 0) Xs[(xpk[u] >> 16) * P.XSTR + (xpk[u] & 0xffff)] = xreg[u];
    } else {  // oversized X tile (not produced by the EBEN layers): direct staging
      const int b = q / P.nct;
      const int t0 = (q - b * P.nct) * BK;
      const int qbase = t0 * P.S + P.off0;
      for (int c = 0; c < nch; ++c) {
        const long long row = ((long long)b * P.Cx + (long long)g * P.Cg + c_lo + c) * P.Lx;
        for (int r = tid; r < span; r += NT) {
          int p = qbase + r;
          if (P.reflect) {
            p = p < 0 ? -p : p;
            p = p >= P.Lx ? 2 * (P.Lx - 1) - p : p;
          }
          float v = 0.f;
          if (p >= 0 && p < P.Lx) v = load_op(P.x, P.xmask, row + p, P.x_mode, P.x_slope);
          Xs[c * P.XSTR + r] = v;
        }
      }
    }
  };

  if (z < P.nchunks) load_chunk(z);
  for (int q = z; q < P.nchunks; q += P.nsplit) {
    __syncthreads();  // MFMAs of the previous chunk are done with Ds / Xs
    store_chunk(q);
    __syncthreads();
    if (q + P.nsplit < P.nchunks) load_chunk(q + P.nsplit);
    const float* drow = Ds + (wm * FM * 16 + l15) * DSTR + kk;
    for (int kb = 0; kb < BK; kb += 32) {   // 8 k-steps per unrolled block, fragments one step ahead
      float a[2][FM], bv[2][FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) a[0][i] = drow[i * 16 * DSTR + kb];
#pragma unroll
      for (int n = 0; n < FN; ++n) bv[0][n] = Xs[xoff[n] + kb * xstep[n]];
#pragma unroll
      for (int ks = 0; ks < 32; ks += 4) {
        const int cur = (ks >> 2) & 1, nxt = cur ^ 1;
        if (ks + 4 < 32) {
#pragma unroll
          for (int i = 0; i < FM; ++i) a[nxt][i] = drow[i * 16 * DSTR + kb + ks + 4];
#pragma unroll
          for (int n = 0; n < FN; ++n) bv[nxt][n] = Xs[xoff[n] + (kb + ks + 4) * xstep[n]];
        }
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int n = 0; n < FN; ++n)
            acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][i], bv[cur][n], acc[i][n], 0, 0, 0);
      }
    }
  }

  float* slab = P.slabs + (long long)z * P.slab_stride;
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wm * FM * 16 + i * 16 + kk * 4 + r;
      if (m >= P.Mg) continue;
      float* orow = slab + ((long long)g * P.Mg + m) * P.row_stride;
#pragma unroll
      for (int n = 0; n < FN; ++n) {
        const int col = n0 + wn * FN * 16 + n * 16 + l15;
        if (col < P.row_stride) orow[col] = acc[i][n][r];
      }
    }
}

struct DwPlan {
  int Cg, Mg, G, J, Ng, row_stride, cfg, BM, BN, nnt, nmt, nct, nchunks, nsplit, XSTR, nch_max, bk;
  size_t lds_bytes;
  long long slab_stride;
};

static void make_dw_plan(const Canon& c, DwPlan* p) {
  p->G = c.g; p->Cg = c.Cin / c.g; p->Mg = c.Cout / c.g; p->J = c.k;
  p->Ng = p->Cg * c.k; p->row_stride = p->Ng + 1;
  // The first-generation kernel is the universal fp32 fallback (weight gradients with fewer than 8 columns per row or one input
  // channel, the logits layers of the fp32-at-rest plans): ONE configuration, 16 x 256 tiles on eight waves.  (Rounds 1-2 chose among
  // seven tile shapes; the layers those were tuned for run on conv_dw2 / conv_dw3 / bl_dw, profiles/r05_kernel_coverage.txt.)
  p->cfg = 6; p->BM = 16; p->BN = 256;
  p->nnt = ceil_div(p->row_stride, p->BN);
  p->nmt = ceil_div(p->Mg, p->BM);
  p->nch_max = (p->BN - 1) / c.k + 2;
  if (p->nch_max > p->Cg) p->nch_max = p->Cg;
  // K-chunk length: as long as possible (fewer barriers per MFMA) while the X tile still fits the
  // register prefetch and the chunk is not mostly padding for short rows
  int bk = DW_BK_MAX;
  static const int env_bk = getenv("EBEN_DW_BK") ? atoi(getenv("EBEN_DW_BK")) : 0;  // tuning aid
  if (env_bk == 32 || env_bk == 64 || env_bk == 128) bk = env_bk;
  while (bk > 32 && ((long long)p->nch_max * ((bk - 1) * c.s + (c.k - 1) * c.d + 1) > DW_XCAP || bk / 2 >= c.Lout)) bk /= 2;
  p->bk = bk;
  p->nct = ceil_div(c.Lout, bk);
  p->nchunks = c.B * p->nct;
  const int tiles = p->nnt * p->nmt * p->G;
  // enough blocks to fill 256 CUs a few times over; no split at all once the tiles alone do that
  static const int target_blocks = getenv("EBEN_DW_BLOCKS") ? atoi(getenv("EBEN_DW_BLOCKS")) : 1536;  // tuning aid (768 -> 1536: +10..30 % on the thin layers)
  int ns = tiles >= 384 ? 1 : ceil_div(target_blocks, tiles);
  if (ns > 512) ns = 512;
  if (ns > p->nchunks) ns = p->nchunks;
  if (ns < 1) ns = 1;
  p->nsplit = ns;
  const int span = (bk - 1) * c.s + (c.k - 1) * c.d + 1;
  p->XSTR = span | 1;  // odd row stride
  p->lds_bytes = 4ull * ((size_t)p->BM * (bk + 2) + (size_t)p->nch_max * p->XSTR + 2);
  p->slab_stride = (long long)c.Cout * p->row_stride;
}

template <int WM, int WN, int FM, int FN>
static int launch_dw_cfg(const DwArgs& a, int nblocks, size_t lds, hipStream_t st) {
  static LdsAttrOnce attr_once;
  auto kern = conv_dw_kernel<WM, WN, FM, FN>;
  {
    const hipError_t e = lds_attr_once(attr_once, reinterpret_cast<const void*>(kern));
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(conv_dw)");
  }
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(WM * WN * 64), lds, st, a);
  EBEN_CHECK_LAUNCH("conv_dw_kernel");
  return EBEN_OK;
}

// ---- streaming weight gradient for layers with a handful of columns per row ---------------------------------------
// PQMF-disc L0 (1 channel x 3 taps per group), MelGAN L0 (1 x 15), the generator's first conv (2 x 3): the GEMM forms
// above spend their time on padding (M x N is 6 x 4 ... 16 x 16 per group, K is 5e5 long), while the layer is a pure
// stream: one gradient row against NA-1 shifted copies of a few input rows.  One block = one (batch item, output row);
// a thread walks the time steps tid, tid+256, ... with NA accumulators (column c*J+j, the last one the bias sum),
// the block reduces them in a fixed order and writes row `row` of slab `b` -- split-K over the batch, reduced by
// eben_wn_bwd like every other slab set.  HBM-bound: the gradient is read once, the input rows hit the cache.
template <int NA>
__global__ __launch_bounds__(256) void tiny_dw_kernel(const DwArgs P) {
  __shared__ float red[4][NA];
  const int row = blockIdx.x;                 // g*Mg + m
  const int b = blockIdx.y;
  const int g = row / P.Mg;
  const float* pa = P.a + ((long long)b * P.Ca + row) * P.La;
  const float* pam = P.a_mode ? P.amask + ((long long)b * P.Ca + row) * P.La : pa;
  const float* px = P.x + ((long long)b * P.Cx + (long long)g * P.Cg) * P.Lx;
  float acc[NA];
#pragma unroll
  for (int n = 0; n < NA; ++n) acc[n] = 0.f;
  int crow[NA - 1], joff[NA - 1];   // column n = (channel c, tap j): row offset and tap offset, resolved once
#pragma unroll
  for (int n = 0; n < NA - 1; ++n) {
    const int c = n < P.Ng ? n / P.J : 0, j = n < P.Ng ? n - c * P.J : 0;
    crow[n] = c * P.Lx;
    joff[n] = j * P.d;
  }
  for (int t = threadIdx.x; t < P.La; t += 256) {
    float a = pa[t];
    a = P.a_mode == 0 ? lrelu(a, P.a_slope) : a * dlrelu(pam[t], P.a_slope);
    const int q0 = t * P.S + P.off0;
#pragma unroll
    for (int n = 0; n < NA - 1; ++n) {
      if (n < P.Ng) {
        int p = q0 + joff[n];
        const int m1 = p < 0 ? -p : p;
        const int m2 = m1 >= P.Lx ? 2 * (P.Lx - 1) - m1 : m1;
        p = P.reflect ? m2 : p;
        const bool ok = p >= 0 && p < P.Lx;
        const float xv = px[crow[n] + (ok ? p : 0)];
        acc[n] = fmaf(a, ok ? lrelu(xv, P.x_slope) : 0.f, acc[n]);
      }
    }
    acc[NA - 1] += a;
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int n = 0; n < NA; ++n) {
    const float v = wave_sum(acc[n]);
    if (lane == 0) red[w][n] = v;
  }
  __syncthreads();
  if (threadIdx.x < NA) {
    const int n = threadIdx.x;
    const float v = (red[0][n] + red[1][n]) + (red[2][n] + red[3][n]);
    float* o = P.slabs + (long long)b * P.slab_stride + (long long)row * P.row_stride;
    if (n < NA - 1) { if (n < P.Ng) o[n] = v; }
    else if (P.has_bias) o[P.Ng] = v;
  }
}

// ---- weight gradient of a single-output-channel conv (the logits layers: 768 -> 1 and 1024 -> 1, k = 3) -----------------
// dw[c, j] = sum_t a[t] * x[c, t + off0 + j*d]: a matrix-vector product per batch item -- the input is read ONCE, the one
// gradient row stays in cache.  One block = one batch item x 16 channels (4 per wave, their 12 loads in flight together),
// lanes walk the time steps, wave_sum in a fixed order; slab b holds item b's row [c*J + j ..., bias] like tiny_dw's.
constexpr int M1DW_CPB = 16;
__global__ __launch_bounds__(256) void m1_dw_kernel(const DwArgs P) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int b = blockIdx.y, c0 = blockIdx.x * M1DW_CPB + w * 4;
  const float* pa = P.a + (long long)b * P.La;
  const float* pam = P.a_mode ? P.amask + (long long)b * P.La : pa;
  const float* px = P.x + (long long)b * P.Cx * P.Lx;
  float acc[4][3], asum = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[u][j] = 0.f;
  for (int t = lane; t < P.La; t += 64) {
    float a = pa[t];
    a = P.a_mode == 0 ? lrelu(a, P.a_slope) : a * dlrelu(pam[t], P.a_slope);
    asum += a;
    int q[3];
    bool ok[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int p = t + P.off0 + j * P.d;
      ok[j] = p >= 0 && p < P.Lx;
      q[j] = ok[j] ? p : 0;
    }
    float xv[4][3];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = c0 + u < P.Cx ? c0 + u : 0;
#pragma unroll
      for (int j = 0; j < 3; ++j) xv[u][j] = px[(long long)c * P.Lx + q[j]];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[u][j] = fmaf(a, ok[j] ? lrelu(xv[u][j], P.x_slope) : 0.f, acc[u][j]);
  }
  float* o = P.slabs + (long long)b * P.slab_stride;
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float v = wave_sum(acc[u][j]);
      if (lane == 0 && c0 + u < P.Cx) o[(c0 + u) * 3 + j] = v;
    }
  if (blockIdx.x == 0 && w == 0 && P.has_bias) {
    const float v = wave_sum(asum);
    if (lane == 0) o[P.Ng] = v;
  }
}

static bool m1_dw_applicable(const Canon& c, const EbenConv1dDesc* d) {
  static const int enabled = getenv("EBEN_M1_DW") ? atoi(getenv("EBEN_M1_DW")) : 1;
  return enabled && !d->transposed && c.Cout == 1 && c.g == 1 && c.s == 1 && c.k == 3 && !c.reflect && c.Cin >= 64;
}

static bool tiny_dw_applicable(const Canon& c, const EbenConv1dDesc* d) {
  static const int enabled = getenv("EBEN_TINY_DW") ? atoi(getenv("EBEN_TINY_DW")) : 1;
  const int Ng = (c.Cin / c.g) * c.k;
  // fused input stage on the X side is lrelu-on-load only (x_mode 0): a Conv1d (the mask, if any, is on the gradient side)
  // measured: 3.3x faster than the 16x16x4 kernel at 4 columns (PQMF-disc L0: 0.19 -> 0.058 ms), slower from 7 columns up
  // (15 shifted re-reads of the input row per step go through the texture path)
  return enabled && !d->transposed && Ng + 1 <= 4 && c.Lout >= 512;
}

// ---- weight norm ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wn_scale_kernel(const float* __restrict__ g, const float* __restrict__ v, int cols,
                                                       float* __restrict__ scale, float* __restrict__ norm) {
  __shared__ float red[4];
  const int r = blockIdx.x;
  const float* row = v + (long long)r * cols;
  float s = 0.f;
  for (int i = threadIdx.x; i < cols; i += 256) { const float t = row[i]; s += t * t; }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) {
    const float n = sqrtf(s);
    norm[r] = n;
    scale[r] = g[r] / n;
  }
}

// slab reduction: out[i] = sum_z slabs[z][i] for i < n.  64 elements x 4 z-groups per block: each
// thread keeps 4 independent partial sums in flight (latency-bound otherwise: tiny layers use up to
// 512 slabs), the 4 z-groups are combined in a fixed order through LDS (deterministic).
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, int nslab, long long slab_stride,
                                                          long long n, float* __restrict__ out) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, zg = threadIdx.x >> 6;
  for (long long base = (long long)blockIdx.x * 64; base < n; base += (long long)gridDim.x * 64) {
    const long long i = base + lane;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < n) {
      int z = zg;
      for (; z + 12 < nslab; z += 16) {
        s0 += slabs[(long long)z * slab_stride + i];
        s1 += slabs[(long long)(z + 4) * slab_stride + i];
        s2 += slabs[(long long)(z + 8) * slab_stride + i];
        s3 += slabs[(long long)(z + 12) * slab_stride + i];
      }
      for (; z < nslab; z += 4) s0 += slabs[(long long)z * slab_stride + i];
    }
    __syncthreads();
    part[zg][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (zg == 0 && i < n) out[i] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
  }
}

// per weight row: dw row (already summed, row stride `row_stride`) -> dg, dv (+ dbias from column `cols`)
__global__ __launch_bounds__(256) void wn_bwd_kernel(const float* __restrict__ dw, int cols, int row_stride,
                                                     const float* __restrict__ g, const float* __restrict__ v,
                                                     const float* __restrict__ norm, float* __restrict__ dg, float* __restrict__ dv,
                                                     float* __restrict__ dbias) {
  __shared__ float red[4];
  const int r = blockIdx.x;
  const float* srow = dw + (long long)r * row_stride;
  float* orow = dv + (long long)r * cols;
  if (dbias && threadIdx.x == 0) dbias[r] = srow[cols];
  if (!g) {
    for (int i = threadIdx.x; i < cols; i += 256) orow[i] = srow[i];
    return;
  }
  const float* vrow = v + (long long)r * cols;
  float dot = 0.f;
  for (int i = threadIdx.x; i < cols; i += 256) dot += srow[i] * vrow[i];
  dot = block_sum_256(dot, red);
  const float n = norm[r], gr = g[r];
  const float dgr = dot / n;
  if (threadIdx.x == 0) dg[r] = dgr;
  const float c1 = gr / n, c2 = gr * dgr / (n * n);
  for (int i = threadIdx.x; i < cols; i += 256) orow[i] = c1 * srow[i] - c2 * vrow[i];
}

// ---- multi-tensor forms: one launch for every layer of a network (the per-layer launches are ~5 us kernels whose
// launch overhead and dependent-launch gaps dominate; 75 + 166 of the ~1070 launches of a train step) -------------
constexpr int WN_CHUNK = 32;
struct WnScaleTable { EbenWnScaleItem t[WN_CHUNK]; };
struct WnBwdTable { EbenWnBwdItem t[WN_CHUNK]; unsigned first[WN_CHUNK + 1]; };   // first[]: prefix sums of the items' slab-reduce blocks

__global__ __launch_bounds__(256) void wn_scale_multi_kernel(const WnScaleTable T) {
  __shared__ float red[4];
  const EbenWnScaleItem e = T.t[blockIdx.y];
  const int r = blockIdx.x;
  if (r >= e.rows) return;
  const float* row = e.v + (long long)r * e.cols;
  float s = 0.f;
  for (int i = threadIdx.x; i < e.cols; i += 256) { const float t = row[i]; s += t * t; }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) {
    const float n = sqrtf(s);
    e.norm[r] = n;
    e.scale[r] = e.g[r] / n;
  }
}

// same arithmetic and summation order as slab_reduce_kernel / wn_bwd_kernel (results are bit-identical).  A block belongs to one item
// (prefix sums of the items' block counts: a grid of max-blocks x items spent most of its blocks on nothing).  Up to 31 slabs -- the
// tap-conv / bundle-layout weight gradients: 2-16 slabs of up to 10.7 M elements -- one thread sums one element with every slab's
// load in flight at once (the four z-groups of the fixed order kept as four partial sums in the thread); from 32 slabs up (the thin
// layers' per-item slabs) the z-groups stay spread over the block's four waves.
// [MI355X] 0.40 ms of kernel time per step before (64 elements per block pass, two barriers, one or two loads in flight per thread)
__global__ __launch_bounds__(256) void slab_reduce_multi_kernel(const WnBwdTable T) {
  __shared__ float part[4][64];
  int it = 0;
#pragma unroll 1
  while (blockIdx.x >= T.first[it + 1]) ++it;
  const EbenWnBwdItem e = T.t[it];
  const unsigned bid = blockIdx.x - T.first[it], nblk = T.first[it + 1] - T.first[it];
  const long long n = (long long)e.rows * e.row_stride;
  const float* slabs = e.slabs;
  float* out = const_cast<float*>(e.slabs);
  if (e.nslab < 32) {
    for (long long i = (long long)bid * 256 + threadIdx.x; i < n; i += (long long)nblk * 256) {
      float p[4];
#pragma unroll
      for (int zg = 0; zg < 4; ++zg) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int z = zg;
        for (; z + 12 < e.nslab; z += 16) {
          s0 += slabs[(long long)z * e.slab_stride + i];
          s1 += slabs[(long long)(z + 4) * e.slab_stride + i];
          s2 += slabs[(long long)(z + 8) * e.slab_stride + i];
          s3 += slabs[(long long)(z + 12) * e.slab_stride + i];
        }
        for (; z < e.nslab; z += 4) s0 += slabs[(long long)z * e.slab_stride + i];
        p[zg] = (s0 + s1) + (s2 + s3);
      }
      out[i] = (p[0] + p[1]) + (p[2] + p[3]);
    }
    return;
  }
  const int lane = threadIdx.x & 63, zg = threadIdx.x >> 6;
  for (long long base = (long long)bid * 64; base < n; base += (long long)nblk * 64) {
    const long long i = base + lane;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < n) {
      int z = zg;
      for (; z + 12 < e.nslab; z += 16) {
        s0 += slabs[(long long)z * e.slab_stride + i];
        s1 += slabs[(long long)(z + 4) * e.slab_stride + i];
        s2 += slabs[(long long)(z + 8) * e.slab_stride + i];
        s3 += slabs[(long long)(z + 12) * e.slab_stride + i];
      }
      for (; z < e.nslab; z += 4) s0 += slabs[(long long)z * e.slab_stride + i];
    }
    __syncthreads();
    part[zg][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (zg == 0 && i < n) out[i] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
  }
}

__global__ __launch_bounds__(256) void wn_bwd_multi_kernel(const WnBwdTable T) {
  __shared__ float red[4];
  const EbenWnBwdItem e = T.t[blockIdx.y];
  const int r = blockIdx.x;
  if (r >= e.rows) return;
  const int cols = e.cols;
  const float* srow = e.slabs + (long long)r * e.row_stride;
  float* orow = e.dv + (long long)r * cols;
  if (e.dbias && threadIdx.x == 0) e.dbias[r] = srow[cols];
  // slab column of weight element i = c k + j: i itself, or the bundle-major order of bl_dw.hip (its stores are contiguous that way):
  // column s = ((c / 8) k + j) 8 + c % 8.  The permuted rows are walked in SLAB order (contiguous reads of the sums; the element index
  // follows by carries instead of a division per element and pass)
  const int pk = e.col_perm_k;
  const float* vrow = e.g ? e.v + (long long)r * cols : nullptr;
  if (pk > 0) {
    const int eo = threadIdx.x & 7;
    int cb0 = (threadIdx.x >> 3) / pk, j0 = (threadIdx.x >> 3) - cb0 * pk;
    auto walk = [&](auto&& f) {
      int cb = cb0, j = j0;
      for (int sc = threadIdx.x; sc < cols; sc += 256) {
        f(sc, (8 * cb + eo) * pk + j);
        j += 32;
        while (j >= pk) { j -= pk; ++cb; }
      }
    };
    if (!e.g) {
      walk([&](int sc, int i) { orow[i] = srow[sc]; });
      return;
    }
    float dot = 0.f;
    walk([&](int sc, int i) { dot += srow[sc] * vrow[i]; });
    dot = block_sum_256(dot, red);
    const float n = e.norm[r], gr = e.g[r];
    const float dgr = dot / n;
    if (threadIdx.x == 0) e.dg[r] = dgr;
    const float c1 = gr / n, c2 = gr * dgr / (n * n);
    walk([&](int sc, int i) { orow[i] = c1 * srow[sc] - c2 * vrow[i]; });
    return;
  }
  if (!e.g) {
    for (int i = threadIdx.x; i < cols; i += 256) orow[i] = srow[i];
    return;
  }
  float dot = 0.f;
  for (int i = threadIdx.x; i < cols; i += 256) dot += srow[i] * vrow[i];
  dot = block_sum_256(dot, red);
  const float n = e.norm[r], gr = e.g[r];
  const float dgr = dot / n;
  if (threadIdx.x == 0) e.dg[r] = dgr;
  const float c1 = gr / n, c2 = gr * dgr / (n * n);
  for (int i = threadIdx.x; i < cols; i += 256) orow[i] = c1 * srow[i] - c2 * vrow[i];
}

}  // namespace eben

using namespace eben;

extern "C" int eben_wn_scale_multi(const EbenWnScaleItem* items, int n, void* stream) {
  EBEN_REQUIRE(items && n > 0, "bad wn_scale_multi arguments");
  for (int base = 0; base < n; base += WN_CHUNK) {
    WnScaleTable T;
    const int cnt = n - base < WN_CHUNK ? n - base : WN_CHUNK;
    int max_rows = 0;
    for (int i = 0; i < cnt; ++i) {
      const EbenWnScaleItem& e = items[base + i];
      EBEN_REQUIRE(e.g && e.v && e.scale && e.norm && e.rows > 0 && e.cols > 0, "bad wn_scale_multi item %d", base + i);
      T.t[i] = e;
      if (e.rows > max_rows) max_rows = e.rows;
    }
    hipLaunchKernelGGL(wn_scale_multi_kernel, dim3(max_rows, cnt), dim3(256), 0, as_stream(stream), T);
    EBEN_CHECK_LAUNCH("wn_scale_multi_kernel");
  }
  return EBEN_OK;
}

extern "C" int eben_wn_bwd_multi(const EbenWnBwdItem* items, int n, void* stream) {
  EBEN_REQUIRE(items && n > 0, "bad wn_bwd_multi arguments");
  for (int base = 0; base < n; base += WN_CHUNK) {
    WnBwdTable T;
    const int cnt = n - base < WN_CHUNK ? n - base : WN_CHUNK;
    int max_rows = 0;
    T.first[0] = 0;
    for (int i = 0; i < cnt; ++i) {
      const EbenWnBwdItem& e = items[base + i];
      EBEN_REQUIRE(e.slabs && e.dv && e.nslab > 0 && e.rows > 0 && e.cols > 0 && e.row_stride >= e.cols, "bad wn_bwd_multi item %d", base + i);
      EBEN_REQUIRE(!e.g || (e.v && e.norm && e.dg), "weight-norm backward needs v, norm and dg (item %d)", base + i);
      EBEN_REQUIRE(!e.dbias || e.row_stride > e.cols, "no bias column in the slabs (item %d)", base + i);
      EBEN_REQUIRE(e.col_perm_k == 0 || (e.col_perm_k > 0 && e.cols % (8 * e.col_perm_k) == 0), "bundle-major slab columns need cols = 8 n k (item %d)", base + i);
      EBEN_REQUIRE(e.nslab == 1 || e.slab_stride >= (long long)e.rows * e.row_stride, "slab stride smaller than a slab (item %d)", base + i);
      T.t[i] = e;
      if (e.rows > max_rows) max_rows = e.rows;
      long long b = 0;
      if (e.nslab > 1) {
        const long long n = (long long)e.rows * e.row_stride;
        b = e.nslab < 32 ? (n + 255) / 256 : (n + 63) / 64;
        if (b > 2048) b = 2048;   // grid-stride beyond
      }
      T.first[i + 1] = T.first[i] + (unsigned)b;
    }
    if (T.first[cnt] > 0) {
      hipLaunchKernelGGL(slab_reduce_multi_kernel, dim3(T.first[cnt]), dim3(256), 0, as_stream(stream), T);
      EBEN_CHECK_LAUNCH("slab_reduce_multi_kernel");
    }
    hipLaunchKernelGGL(wn_bwd_multi_kernel, dim3(max_rows, cnt), dim3(256), 0, as_stream(stream), T);
    EBEN_CHECK_LAUNCH("wn_bwd_multi_kernel");
  }
  return EBEN_OK;
}

extern "C" int eben_wn_scale(const float* g, const float* v, int rows, int cols, float* scale, float* norm, void* stream) {
  EBEN_REQUIRE(g && v && scale && norm && rows > 0 && cols > 0, "bad wn_scale arguments");
  hipLaunchKernelGGL(wn_scale_kernel, dim3(rows), dim3(256), 0, as_stream(stream), g, v, cols, scale, norm);
  EBEN_CHECK_LAUNCH("wn_scale_kernel");
  return EBEN_OK;
}

extern "C" int eben_wn_bwd(const float* dw_slabs, int nslab, size_t slab_stride, int rows, int cols, int row_stride,
                           const float* g, const float* v, const float* norm, float* dg, float* dv, float* dbias, void* stream) {
  EBEN_REQUIRE(dw_slabs && dv && nslab > 0 && rows > 0 && cols > 0 && row_stride >= cols, "bad wn_bwd arguments");
  EBEN_REQUIRE(!g || (v && norm && dg), "weight-norm backward needs v, norm and dg");
  EBEN_REQUIRE(!dbias || row_stride > cols, "no bias column in the slabs");
  EBEN_REQUIRE(nslab == 1 || slab_stride >= (size_t)rows * row_stride, "slab stride smaller than a slab");
  hipStream_t st = as_stream(stream);
  if (nslab > 1) {
    // sum the split-K slabs into slab 0 (in place: slab 0 is read before it is written, element-wise)
    const long long n = (long long)rows * row_stride;
    long long blocks = (n + 63) / 64;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, dw_slabs, nslab, (long long)slab_stride, n,
                       const_cast<float*>(dw_slabs));
    EBEN_CHECK_LAUNCH("slab_reduce_kernel");
  }
  hipLaunchKernelGGL(wn_bwd_kernel, dim3(rows), dim3(256), 0, st, dw_slabs, cols, row_stride, g, v, norm, dg, dv, dbias);
  EBEN_CHECK_LAUNCH("wn_bwd_kernel");
  return EBEN_OK;
}

extern "C" size_t eben_conv1d_bwd_dw_workspace(const EbenConv1dDesc* d, int* nslab, int* row_stride) {
  Canon c;
  if (canon_from_desc(d, &c) != EBEN_OK) return 0;
  if (tiny_dw_applicable(c, d)) {
    const int rs = (c.Cin / c.g) * c.k + 1;
    if (nslab) *nslab = c.B;
    if (row_stride) *row_stride = rs;
    return sizeof(float) * (size_t)c.B * c.Cout * rs;
  }
  if (m1_dw_applicable(c, d)) {
    const int rs = c.Cin * c.k + 1;
    if (nslab) *nslab = c.B;
    if (row_stride) *row_stride = rs;
    return sizeof(float) * (size_t)c.B * rs;
  }
  if ((d->out_slope == 1.f || !d->transposed) && dw3_applicable(c)) return dw3_workspace(c, nslab, row_stride);   // bf16 math (no mask on the X operand)
  if (dw2_applicable(c)) return dw2_workspace(c, nslab, row_stride);
  DwPlan p;
  make_dw_plan(c, &p);
  if (nslab) *nslab = p.nsplit;
  if (row_stride) *row_stride = p.row_stride;
  return sizeof(float) * (size_t)p.slab_stride * p.nsplit;
}

extern "C" int eben_conv1d_bwd_dw(const EbenConv1dDesc* d, const float* dy, const float* y, const float* x, int has_bias,
                                  float* slabs, size_t ws_bytes, void* stream) {
  Canon c;
  int rc = canon_from_desc(d, &c);
  if (rc) return rc;
  EBEN_REQUIRE(dy && x && slabs, "null pointer in conv1d_bwd_dw");
  EBEN_REQUIRE(d->out_slope == 1.f || y, "y is required to differentiate the fused output activation");
  EBEN_REQUIRE(!(d->transposed && has_bias), "ConvTranspose1d bias gradient is not provided by this kernel");
  if (tiny_dw_applicable(c, d)) {
    DwArgs a{};
    a.a = dy; a.amask = y; a.a_mode = d->out_slope != 1.f ? 1 : 0; a.a_slope = a.a_mode ? d->out_slope : 1.f;
    a.x = x; a.xmask = nullptr; a.x_mode = 0; a.x_slope = d->in_slope;
    a.slabs = slabs;
    a.B = c.B; a.G = c.g; a.Cg = c.Cin / c.g; a.Mg = c.Cout / c.g; a.Ca = c.Cout; a.Cx = c.Cin; a.La = c.Lout; a.Lx = c.Lin;
    a.S = c.s; a.d = c.d; a.off0 = -c.pl; a.J = c.k; a.Ng = a.Cg * c.k; a.has_bias = has_bias ? 1 : 0; a.row_stride = a.Ng + 1;
    a.reflect = c.reflect; a.slab_stride = (long long)c.Cout * a.row_stride;
    const size_t need = sizeof(float) * (size_t)c.B * a.slab_stride;
    if (ws_bytes < need) return fail(EBEN_EWORKSPACE, "bwd_dw needs %zu workspace bytes, got %zu", need, ws_bytes);
    const dim3 grid(c.Cout, c.B);
    EBEN_REQUIRE(a.row_stride <= 4, "tiny_dw serves at most 4 columns per row");
    hipLaunchKernelGGL(tiny_dw_kernel<4>, grid, dim3(256), 0, as_stream(stream), a);
    EBEN_CHECK_LAUNCH("tiny_dw_kernel");
    return EBEN_OK;
  }
  if (m1_dw_applicable(c, d)) {
    DwArgs a{};
    a.a = dy; a.amask = y; a.a_mode = d->out_slope != 1.f ? 1 : 0; a.a_slope = a.a_mode ? d->out_slope : 1.f;
    a.x = x; a.xmask = nullptr; a.x_mode = 0; a.x_slope = d->in_slope;
    a.slabs = slabs;
    a.B = c.B; a.G = 1; a.Cg = c.Cin; a.Mg = 1; a.Ca = 1; a.Cx = c.Cin; a.La = c.Lout; a.Lx = c.Lin;
    a.S = 1; a.d = c.d; a.off0 = -c.pl; a.J = 3; a.Ng = c.Cin * 3; a.has_bias = has_bias ? 1 : 0; a.row_stride = a.Ng + 1;
    a.reflect = 0; a.slab_stride = a.row_stride;
    const size_t need = sizeof(float) * (size_t)c.B * a.slab_stride;
    if (ws_bytes < need) return fail(EBEN_EWORKSPACE, "bwd_dw needs %zu workspace bytes, got %zu", need, ws_bytes);
    hipLaunchKernelGGL(m1_dw_kernel, dim3(ceil_div(c.Cin, M1DW_CPB), c.B), dim3(256), 0, as_stream(stream), a);
    EBEN_CHECK_LAUNCH("m1_dw_kernel");
    return EBEN_OK;
  }
  if ((d->out_slope == 1.f || !d->transposed) && dw3_applicable(c)) {
    if (!d->transposed)   // the gradient operand carries the activation derivative of the fused output stage
      return dw3_launch(c, dy, d->out_slope != 1.f ? y : nullptr, d->out_slope, x, d->in_slope, has_bias ? 1 : 0, slabs, ws_bytes, as_stream(stream));
    return dw3_launch(c, x, nullptr, d->in_slope, dy, 1.f, 0, slabs, ws_bytes, as_stream(stream));
  }
  if (dw2_applicable(c)) {
    Dw2Args a2;
    if (!d->transposed) {
      a2.a = dy; a2.amask = y; a2.a_mode = d->out_slope != 1.f ? 1 : 0; a2.a_slope = d->out_slope;
      a2.x = x; a2.xmask = nullptr; a2.x_mode = 0; a2.x_slope = d->in_slope;
    } else {
      a2.a = x; a2.amask = nullptr; a2.a_mode = 0; a2.a_slope = d->in_slope;
      a2.x = dy; a2.xmask = y; a2.x_mode = d->out_slope != 1.f ? 1 : 0; a2.x_slope = d->out_slope;
    }
    if (a2.a_mode == 0 && a2.a == dy) a2.a_slope = 1.f;
    if (a2.x_mode == 0 && a2.x == dy) a2.x_slope = 1.f;
    a2.has_bias = has_bias ? 1 : 0;
    return dw2_launch(c, a2, slabs, ws_bytes, as_stream(stream));
  }
  DwPlan p;
  make_dw_plan(c, &p);
  const size_t need = sizeof(float) * (size_t)p.slab_stride * p.nsplit;
  if (ws_bytes < need) return fail(EBEN_EWORKSPACE, "bwd_dw needs %zu workspace bytes, got %zu", need, ws_bytes);
  if (p.lds_bytes > 160 * 1024) return fail(EBEN_EUNSUPPORTED, "conv_dw tile needs %zu B of LDS", p.lds_bytes);
  DwArgs a;
  if (!d->transposed) {
    a.a = dy; a.amask = y; a.a_mode = d->out_slope != 1.f ? 1 : 0; a.a_slope = d->out_slope;
    a.x = x; a.xmask = nullptr; a.x_mode = 0; a.x_slope = d->in_slope;
  } else {
    a.a = x; a.amask = nullptr; a.a_mode = 0; a.a_slope = d->in_slope;
    a.x = dy; a.xmask = y; a.x_mode = d->out_slope != 1.f ? 1 : 0; a.x_slope = d->out_slope;
  }
  if (a.a_mode == 0 && a.a == dy) a.a_slope = 1.f;
  if (a.x_mode == 0 && a.x == dy) a.x_slope = 1.f;
  a.slabs = slabs;
  a.B = c.B; a.G = c.g; a.Cg = p.Cg; a.Mg = p.Mg; a.Ca = c.Cout; a.Cx = c.Cin; a.La = c.Lout; a.Lx = c.Lin;
  a.S = c.s; a.d = c.d; a.off0 = -c.pl; a.J = c.k; a.Ng = p.Ng; a.has_bias = has_bias ? 1 : 0; a.row_stride = p.row_stride;
  a.reflect = c.reflect;
  a.nsplit = p.nsplit; a.nct = p.nct; a.nchunks = p.nchunks; a.nnt = p.nnt; a.nmt = p.nmt; a.XSTR = p.XSTR; a.bk = p.bk;
  a.slab_stride = p.slab_stride;
  const int nb = p.nnt * p.nmt * p.G * p.nsplit;
  hipStream_t st = as_stream(stream);
  return launch_dw_cfg<1, 8, 1, 2>(a, nb, p.lds_bytes, st);
}
