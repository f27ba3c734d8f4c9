// tapconv.hip -- the MFMA implicit-GEMM "tap convolution" of libeben_hip.so (gfx950).
//
// One kernel covers the two directions of every Conv1d / ConvTranspose1d of the EBEN generator
// and discriminators (reference call sites: eben_generator.py:112-166,241-249,272-280,295-312;
// eben_discriminator.py:66-157; melgan_discriminator.py:89-156) and the STFT of the
// multi-resolution spectral loss (a strided conv against a windowed DFT basis):
//
//   y[b, g*Mg+m, t*OS+oo] = epi( sum_{c<Cg} sum_{j<J} Wp[j,c,m] * xin(b, g*Cg+c, t*S + off0 + j*dstep) )
//
//   GS ("gather-strided", S = conv stride, OS = 1): Conv1d forward / ConvTranspose1d input-gradient.
//   PS ("phase-scatter",  S = 1, OS = conv stride): Conv1d input-gradient / ConvTranspose1d forward,
//        one stride-1 sub-convolution per output phase r = u mod stride (taps k with
//        (r + pad - k*dil) % stride == 0), so no zero-stuffed MACs are ever issued.
//
// GEMM view per (batch item, group): M = output channels, N = output positions, K = (tap, channel).
// fp32 in / fp32 accumulate on v_mfma_f32_16x16x4_f32 (exact fp32, 157 TFLOP/s chip peak).
//   * the input tile (CI_T channels x receptive span) is staged once per channel chunk in LDS,
//     de-interleaved by stride phase so that the 16 lanes of a B fragment read consecutive
//     dwords whatever the stride (conflict-free ds_read_b32);
//   * weights arrive pre-packed [phase][group][chunk][k][Mpad] (weight-norm scale folded in by
//     the pack kernel), so the A tile is a straight float4 copy, register-prefetched one chunk
//     ahead of the MFMAs;
//   * LeakyReLU on load, bias / LeakyReLU / residual / activation-derivative mask in the epilogue;
//   * blockIdx -> tile mapping is XCD-aware (tiles sharing a weight panel stay on one XCD's L2).
#include "common.h"

#include <cstdlib>

namespace eben {

struct TapArgs {
  const float* x;
  const float* xmask;
  const float* wp;
  const float* bias;
  const float* res;
  const float* emask;
  float* y;
  int B, G, Cg, Mg, Mp, Cx, Cy, Lx, Ly;
  int S, OS, dstep, J, KCpad, CI_T, ncc;
  int mode;              // 0 GS, 1 PS
  int off0, nt;          // GS
  int ps_pad, ps_k, ps_d, ps_kstep;  // PS
  int reflect, in_mode;  // in_mode 0: lrelu(x, in_slope); 1: x * lrelu'(xmask, in_slope)
  float in_slope, out_slope, res_slope, emask_slope;
  int accumulate;
  int res_rows, em_seg, em_map[4];
  int PLEN, CSTRIDE;
  unsigned s_magic;      // ceil(2^32 / S)
  int ntt, nmt, nph;
  long long phase_stride;
};

template <int WAVES_M, int WAVES_N, int FM, int FN>
// second launch-bound = waves per SIMD: an 8-wave block needs 4 (<= 128 VGPRs) for two blocks to share
// a CU -- at 133 VGPRs the 128x128 tile silently ran one block per CU
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, WAVES_M * WAVES_N == 8 ? 4 : 3) void tapconv_kernel(const TapArgs P) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int BM = WAVES_M * FM * 16;
  constexpr int BN = WAVES_N * FN * 16;
  constexpr int KT = 4096 / BM;         // rows of the staged weight chunk
  constexpr int WSTR = BM + 16;         // +16: the 2 k-rows of a half-wave hit disjoint banks
  constexpr int W4_PER_ROW = BM / 4;
  constexpr int W4_PER_THREAD = KT * BM / 4 / NT;
  static_assert(NT == 256 || NT == 512, "4 or 8 waves per block");
  static_assert(W4_PER_THREAD * NT * 4 == 4096, "weight chunk is 4096 floats");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  int* koff = reinterpret_cast<int*>(smem);
  const int koff_len = ((P.KCpad + KT - 1) / KT) * KT;   // padded: the k-loop always runs KT/4 steps
  float* Ws = smem + koff_len;                           // multiple of 4 -> 16 B aligned
  float* Xs = Ws + KT * WSTR;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kk = lane >> 4;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // ---- tile decode (XCD-aware) ----
  unsigned id = xcd_remap(blockIdx.x, gridDim.x);
  const int tt = id % P.ntt; id /= P.ntt;
  const int b = id % P.B; id /= P.B;
  const int ph = id % P.nph; id /= P.nph;
  const int mt = id % P.nmt;
  const int g = id / P.nmt;
  const int t0 = tt * BN, m0 = mt * BM;

  // ---- per-phase tap geometry ----
  int J = P.J, off0 = P.off0, nt = P.nt, oo = 0;
  if (P.mode == 1) {
    const int s = P.OS;
    int k0 = -1;
    for (int c = 0; c < P.ps_kstep; ++c) {
      int v = (ph + P.ps_pad - c * P.ps_d) % s;
      if (v < 0) v += s;
      if (v == 0) { k0 = c; break; }
    }
    if (k0 >= 0 && k0 < P.ps_k) {
      J = (P.ps_k - 1 - k0) / P.ps_kstep + 1;
      off0 = (ph + P.ps_pad - k0 * P.ps_d) / s;
    } else {
      J = 0;
      off0 = 0;
    }
    nt = ph < P.Ly ? (P.Ly - ph + s - 1) / s : 0;
    oo = ph;
  }
  if (t0 >= nt) return;  // whole block is outside this phase's range (uniform)
  const int minoff = (P.dstep >= 0 || J == 0) ? off0 : off0 + (J - 1) * P.dstep;
  const int adstep = P.dstep >= 0 ? P.dstep : -P.dstep;
  const int span = J > 0 ? (BN - 1) * P.S + (J - 1) * adstep + 1 : 0;
  const int KC = ((J * P.CI_T + 3) >> 2) << 2;  // flat K entries actually used by this phase

  for (int f = tid; f < koff_len; f += NT) {
    const int j = f / P.CI_T, cl = f - j * P.CI_T;
    int o = 0;
    if (f < P.KCpad && j < J) {
      const int rel = off0 + j * P.dstep - minoff;
      int p = 0, dd = rel;
      if (P.S != 1) { dd = (int)__umulhi((unsigned)rel, P.s_magic); p = rel - dd * P.S; }
      o = cl * P.CSTRIDE + p * P.PLEN + dd;
    }
    koff[f] = o;
  }

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int n = 0; n < FN; ++n) acc[i][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  const float* wbase = P.wp + (long long)ph * P.phase_stride + (long long)g * P.ncc * P.KCpad * P.Mp + m0;
  const int q0 = t0 * P.S + minoff;
  const int nkc = (KC + KT - 1) / KT;

  // ---- main loop.  Both operand streams are software-pipelined through registers:
  //  * weights: one 4096-float chunk ahead, continuously across channel-chunk boundaries (the packed
  //    layout is contiguous in consumption order);
  //  * input tile: when it is small (<= XCAP elements: the k=3..7 layers, which would otherwise
  //    expose a global-load latency every ~100 K-rows) the next channel chunk's tile is fetched into
  //    registers under the current chunk's MFMAs; large tiles (k=41 stride-4 layers) are staged
  //    directly -- there the staging is amortised over ~20 weight chunks.
  constexpr int XR = 10;
  constexpr int XCAP = XR * NT;
  const int xtot = P.CI_T * span;
  const bool xpre = xtot <= XCAP && span < 65536;
  float xreg[XR];
  // tile coordinates of this thread's elements: held in registers by the light variants, recomputed at every use (from a copy of
  // the thread id the compiler cannot see through, so nothing is hoisted back) by the 16-fragment ones, which otherwise spill
  constexpr bool XG_REG = FM * FN < 8;
  const unsigned span_magic = span > 1 ? (unsigned)((0x100000000ull + (unsigned)span - 1) / (unsigned)span) : 0u;
  auto xg_make = [&](int t, int u) -> int {
    const int i = t + u * NT;
    if (!(xpre && i < xtot)) return -1;
    const int c = span > 1 ? (int)__umulhi((unsigned)i, span_magic) : i;
    return (c << 16) | (i - c * span);
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
  auto fetch_x = [&](int r, long long row, bool cv) -> float {
    int q = q0 + r;
    if (P.reflect) {
      q = q < 0 ? -q : q;
      q = q >= P.Lx ? 2 * (P.Lx - 1) - q : q;
    }
    float v = 0.f;
    if (cv && q >= 0 && q < P.Lx) {
      v = P.x[row + q];
      if (P.in_mode == 0) v = lrelu(v, P.in_slope);
      else v *= dlrelu(P.xmask[row + q], P.in_slope);
    }
    return v;
  };
  auto lds_x = [&](int c, int r) -> int {
    int p = 0, i = r;
    if (P.S != 1) { i = (int)__umulhi((unsigned)r, P.s_magic); p = r - i * P.S; }
    return c * P.CSTRIDE + p * P.PLEN + i;
  };
  auto load_x = [&](int cc) {
    const int tq = opaque_tid();
#pragma unroll
    for (int u = 0; u < XR; ++u) {
      float v = 0.f;
      const int g_ = XG_REG ? xg_r[XG_REG ? u : 0] : xg_make(tq, u);
      if (g_ >= 0) {
        const int chan = cc * P.CI_T + (g_ >> 16);
        v = fetch_x(g_ & 0xffff, ((long long)b * P.Cx + (long long)g * P.Cg + chan) * P.Lx, chan < P.Cg);
      }
      xreg[u] = v;
    }
  };
  auto store_x = [&]() {
    const int tq = opaque_tid();
#pragma unroll
    for (int u = 0; u < XR; ++u) {
      const int g_ = XG_REG ? xg_r[XG_REG ? u : 0] : xg_make(tq, u);
      if (g_ >= 0) Xs[lds_x(g_ >> 16, g_ & 0xffff)] = xreg[u];
    }
  };
  auto stage_x_direct = [&](int cc) {
    for (int c = 0; c < P.CI_T; ++c) {
      const int chan = cc * P.CI_T + c;
      const long long row = ((long long)b * P.Cx + (long long)g * P.Cg + chan) * P.Lx;
      for (int r = tid; r < span; r += NT) Xs[lds_x(c, r)] = fetch_x(r, row, chan < P.Cg);
    }
  };
  float4 wreg[W4_PER_THREAD];
  auto prefetch_w = [&](int cc, int kc) {
    const float* wcc = wbase + (long long)cc * P.KCpad * P.Mp;
    const int rows = min(KT, KC - kc * KT);
#pragma unroll
    for (int u = 0; u < W4_PER_THREAD; ++u) {
      const int i4 = tid + u * NT;
      const int rrow = i4 / W4_PER_ROW, c4 = i4 - rrow * W4_PER_ROW;
      wreg[u] = rrow < rows ? *reinterpret_cast<const float4*>(wcc + (long long)(kc * KT + rrow) * P.Mp + c4 * 4)
                            : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };

  if (J > 0) {
    if (xpre) load_x(0);
    prefetch_w(0, 0);
  }
  for (int cc = 0; cc < P.ncc && J > 0; ++cc) {
    __syncthreads();  // previous channel chunk's MFMAs are done with Xs / Ws (and koff is published)
    if (xpre) {
      store_x();
      if (cc + 1 < P.ncc) load_x(cc + 1);
    } else {
      stage_x_direct(cc);
    }
    for (int kc = 0; kc < nkc; ++kc) {
      if (kc > 0) __syncthreads();  // MFMAs of the previous chunk finished reading Ws
#pragma unroll
      for (int u = 0; u < W4_PER_THREAD; ++u) {
        const int i4 = tid + u * NT;
        const int rrow = i4 / W4_PER_ROW, c4 = i4 - rrow * W4_PER_ROW;
        *reinterpret_cast<float4*>(Ws + rrow * WSTR + c4 * 4) = wreg[u];
      }
      __syncthreads();
      if (kc + 1 < nkc) prefetch_w(cc, kc + 1);
      else if (cc + 1 < P.ncc) prefetch_w(cc + 1, 0);
      // fixed-trip k-loop (rows past the chunk end are zero weights against a valid offset 0):
      // tap offsets for the whole chunk are read up front, fragments are loaded one k-step ahead
      // of the MFMAs that consume them.
      const float* wrow = Ws + kk * WSTR + wm * FM * 16 + l15;
      const int* ko = koff + kc * KT + kk;
      const float* xcol = Xs + wn * FN * 16 + l15;
      constexpr int UNR = 4;  // k-steps per unrolled block (16 rows of the weight chunk)
      const int nstep = (min(KT, KC - kc * KT) + 3) >> 2;
      for (int kb = 0; kb < nstep; kb += UNR) {
        int xo[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) xo[u] = ko[(kb + u) * 4];
        float a[2][FM], bv[2][FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) a[0][i] = wrow[(kb * 4) * WSTR + i * 16];
#pragma unroll
        for (int n = 0; n < FN; ++n) bv[0][n] = xcol[xo[0] + n * 16];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int cur = u & 1, nxt = cur ^ 1;
          if (u + 1 < UNR) {
#pragma unroll
            for (int i = 0; i < FM; ++i) a[nxt][i] = wrow[((kb + u + 1) * 4) * WSTR + i * 16];
#pragma unroll
            for (int n = 0; n < FN; ++n) bv[nxt][n] = xcol[xo[u + 1] + n * 16];
          }
#pragma unroll
          for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int n = 0; n < FN; ++n)
              acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][i], bv[cur][n], acc[i][n], 0, 0, 0);
        }
      }
    }
  }

  // batched right-hand sides: residual row limit and remapped mask row (wave-uniform)
  const bool use_res = P.res != nullptr && (P.res_rows == 0 || b < P.res_rows);
  const int eb = P.em_seg > 0 ? P.em_map[b / P.em_seg] * P.em_seg + b % P.em_seg : b;
  const long long eoff = (long long)(eb - b) * P.Cy * P.Ly;
  // ---- epilogue: D fragment = 4 consecutive rows (m) x 1 column (t) per lane ----
#pragma unroll
  for (int i = 0; i < FM; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wm * FM * 16 + i * 16 + kk * 4 + r;
      if (m >= P.Mg) continue;
      const float bias = P.bias ? P.bias[g * P.Mg + m] : 0.f;
      const long long yrow = ((long long)b * P.Cy + (long long)g * P.Mg + m) * P.Ly;
#pragma unroll
      for (int n = 0; n < FN; ++n) {
        const int t = t0 + wn * FN * 16 + n * 16 + l15;
        if (t >= nt) continue;
        const long long idx = yrow + (long long)t * P.OS + oo;
        float v = acc[i][n][r] + bias;
        v = lrelu(v, P.out_slope);
        if (use_res) v += lrelu(P.res[idx], P.res_slope);
        if (P.emask) v *= dlrelu(P.emask[idx + eoff], P.emask_slope);
        if (P.accumulate) v += P.y[idx];
        P.y[idx] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// host side: canonical conv, plans, launchers
// ---------------------------------------------------------------------------------------

int canon_from_desc(const EbenConv1dDesc* d, Canon* c) {
  EBEN_REQUIRE(d != nullptr, "null conv descriptor");
  EBEN_REQUIRE(d->batch > 0 && d->c_in > 0 && d->c_out > 0 && d->l_in > 0 && d->l_out > 0, "non-positive conv dims");
  EBEN_REQUIRE(d->ksize > 0 && d->stride > 0 && d->dilation > 0 && d->groups > 0, "non-positive conv params");
  EBEN_REQUIRE(d->c_in % d->groups == 0 && d->c_out % d->groups == 0, "channels not divisible by groups");
  EBEN_REQUIRE(d->stride <= 1024 && d->pad_l >= 0 && d->pad_r >= 0, "bad stride / padding");
  c->B = d->batch; c->k = d->ksize; c->s = d->stride; c->d = d->dilation; c->g = d->groups;
  c->pl = d->pad_l; c->pr = d->pad_r;
  const int math = d->math & ~EBEN_LAYOUT_BL;
  EBEN_REQUIRE(math >= EBEN_MATH_F32 && math <= EBEN_MATH_BF16X6, "unknown math mode %d", d->math);
  c->bl = (d->math & EBEN_LAYOUT_BL) != 0;
  EBEN_REQUIRE(!c->bl || math == EBEN_MATH_BF16 || math == EBEN_MATH_BF16X3, "the bundle layout carries EBEN_MATH_BF16 / EBEN_MATH_BF16X3 only");
  c->bf16 = math != EBEN_MATH_F32;
  c->np = math == EBEN_MATH_BF16X3 ? 2 : (math == EBEN_MATH_BF16X6 ? 3 : 1);
  c->xsplit_dir = math == EBEN_MATH_BF16X2 ? (d->transposed ? 1 : 0) : -1;
  if (!d->transposed) {
    c->Cin = d->c_in; c->Cout = d->c_out; c->Lin = d->l_in; c->Lout = d->l_out;
    c->reflect = d->pad_mode == EBEN_PAD_REFLECT;
    const int expect = (d->l_in + d->pad_l + d->pad_r - d->dilation * (d->ksize - 1) - 1) / d->stride + 1;
    EBEN_REQUIRE(expect == d->l_out, "Conv1d l_out %d does not match the expected %d", d->l_out, expect);
    if (c->reflect) EBEN_REQUIRE(d->pad_l < d->l_in && d->pad_r < d->l_in, "reflect padding must be smaller than the input");
  } else {
    EBEN_REQUIRE(d->pad_mode == EBEN_PAD_ZERO, "ConvTranspose1d only supports zero padding");
    c->Cin = d->c_out; c->Cout = d->c_in; c->Lin = d->l_out; c->Lout = d->l_in;
    c->reflect = 0;
    c->pr = d->pad_l;
    // l_out = (l_in-1)*s - 2p + d(k-1) + output_padding + 1 with 0 <= output_padding < s
    const int base = (d->l_in - 1) * d->stride - 2 * d->pad_l + d->dilation * (d->ksize - 1) + 1;
    EBEN_REQUIRE(d->l_out >= base && d->l_out < base + d->stride, "ConvTranspose1d l_out %d outside [%d,%d)", d->l_out, base, base + d->stride);
  }
  return EBEN_OK;
}

static int gcd_i(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

struct TapPlan {
  int mode;  // 0 GS, 1 PS
  int Cg, Mg, G, CI_T, ncc, J, KCpad, Mp, cfg, BM, BN;
  int S, OS, dstep, kstep, nph, PLEN, CSTRIDE;
  int Lx, Ly, Cx, Cy, off0, nt, ps_pad;
  int ntt, nmt;
  size_t lds_bytes;
  long long phase_stride;
  size_t packed_floats;
};

// The first-generation kernel is the library's UNIVERSAL fp32 fallback: the shapes no later family takes (one-channel inputs with long
// taps such as the dense STFT form, odd row counts).  One configuration serves them all: 16 x 256 tiles on four waves.  (Rounds 1-2 chose
// among eight tile shapes here; every layer those were tuned for now runs on tapconv2 / tapconv3 / gen_conv / thinconv, and none of
// the other seven was launched by the GPU suite or the bench any more: profiles/r05_kernel_coverage.txt.)
constexpr int kTapBM = 16, kTapBN = 256;

// dir 0: canonical forward direction (GS); dir 1: canonical input-gradient direction (PS).
// For PS with reflect padding the output domain is the padded one (caller folds afterwards).
static void make_plan(const Canon& c, int dir, TapPlan* p) {
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
    p->kstep = c.s / gcd_i(c.s, c.d);
    p->dstep = -(c.d * p->kstep) / c.s;
    p->nph = c.s;
    p->J = ceil_div(c.k, p->kstep);
    p->Lx = c.Lout; p->Cx = c.Cout; p->Cy = c.Cin;
    p->Ly = c.reflect ? c.Lin + c.pl + c.pr : c.Lin;
    p->ps_pad = c.reflect ? 0 : c.pl;
    p->off0 = 0; p->nt = ceil_div(p->Ly, c.s);
  }
  p->cfg = 3;
  p->BM = kTapBM; p->BN = kTapBN;
  p->Mp = round_up(p->Mg, p->BM);
  p->nmt = p->Mp / p->BM;
  p->ntt = ceil_div(p->nt, p->BN);
  // channel chunk: as many input channels per staged tile as 24 KiB of LDS hold (cap 16), evenly
  // split over the group so that no chunk is mostly padding.  Measured on MI355X (tools/layer_bench.py):
  // the kernel is latency-bound, so a small tile (3-4 resident blocks per CU) beats a large one
  // (48 KiB / 64 channels was 10-35 % slower on every heavy layer).
  const int adstep = p->dstep >= 0 ? p->dstep : -p->dstep;
  const int KT = 4096 / p->BM;
  {
    const int maxd = ((p->J - 1) * adstep) / p->S + 1;
    p->PLEN = p->BN + maxd + 1;
    p->CSTRIDE = round_up(p->S * p->PLEN, 32) + 16;
  }
  static const int env_cap = getenv("EBEN_TAP_CI_CAP") ? atoi(getenv("EBEN_TAP_CI_CAP")) : 0;      // tuning aid
  static const int env_kb = getenv("EBEN_TAP_X_KB") ? atoi(getenv("EBEN_TAP_X_KB")) : 0;          // tuning aid
  int cap = (int)(((env_kb > 0 ? env_kb : 24) * 1024) / (4 * (size_t)p->CSTRIDE));
  if (cap > (env_cap > 0 ? env_cap : 16)) cap = env_cap > 0 ? env_cap : 16;
  if (cap < 1) cap = 1;
  int ci;
  if (p->Cg <= cap) ci = p->Cg;
  else {
    const int nch = ceil_div(p->Cg, cap);
    ci = ceil_div(p->Cg, nch);
  }
  for (;;) {
    p->CI_T = ci;
    p->KCpad = round_up(p->J * ci, 4);
    p->lds_bytes = 4ull * ((size_t)round_up(p->KCpad, KT) + (size_t)KT * (p->BM + 16) + (size_t)ci * p->CSTRIDE);
    if (p->lds_bytes <= 160 * 1024 || ci == 1) break;
    ci = ci / 2 > 0 ? ci / 2 : 1;
  }
  p->ncc = ceil_div(p->Cg, p->CI_T);
  p->phase_stride = (long long)p->G * p->ncc * p->KCpad * p->Mp;
  p->packed_floats = (size_t)p->phase_stride * p->nph;
}

// ---- weight packing -----------------------------------------------------------------------
struct PackArgs {
  const float* w;      // canonical (Cout, Cin/g, k)
  const float* scale;  // per Cout row or null
  float* wp;
  int G, Cg, Mg, Mp, CI_T, ncc, KCpad, J, nph;
  int mode, k, s, d, kstep, ps_pad, Cin_g, Cout_g;
  long long phase_stride;
};

__global__ __launch_bounds__(256) void pack_kernel(const PackArgs P) {
  const long long total = P.phase_stride * P.nph;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    long long r = i;
    const int m = r % P.Mp; r /= P.Mp;
    const int f = r % P.KCpad; r /= P.KCpad;
    const int cc = r % P.ncc; r /= P.ncc;
    const int g = r % P.G;
    const int ph = r / P.G;
    const int j = f / P.CI_T, cl = f - j * P.CI_T;
    const int chan = cc * P.CI_T + cl;
    float v = 0.f;
    if (m < P.Mg && chan < P.Cg && j < P.J) {
      if (P.mode == 0) {
        // reduction channel = conv input channel, m = conv output channel, tap j
        const int co = g * P.Cout_g + m;
        v = P.w[((long long)co * P.Cin_g + chan) * P.k + j];
        if (P.scale) v *= P.scale[co];
      } else {
        int k0 = -1;
        for (int c = 0; c < P.kstep; ++c) {
          int q = (ph + P.ps_pad - c * P.d) % P.s;
          if (q < 0) q += P.s;
          if (q == 0) { k0 = c; break; }
        }
        const int kk = k0 + j * P.kstep;
        if (k0 >= 0 && kk < P.k) {
          const int co = g * P.Cout_g + chan;  // reduction channel = conv output channel
          v = P.w[((long long)co * P.Cin_g + m) * P.k + kk];
          if (P.scale) v *= P.scale[co];
        }
      }
    }
    P.wp[i] = v;
  }
}

static int launch_pack(const Canon& c, const TapPlan& p, const float* w, const float* scale, float* wp, hipStream_t st) {
  PackArgs a;
  a.w = w; a.scale = scale; a.wp = wp;
  a.G = p.G; a.Cg = p.Cg; a.Mg = p.Mg; a.Mp = p.Mp; a.CI_T = p.CI_T; a.ncc = p.ncc; a.KCpad = p.KCpad; a.J = p.J; a.nph = p.nph;
  a.mode = p.mode; a.k = c.k; a.s = c.s; a.d = c.d; a.kstep = p.kstep; a.ps_pad = p.ps_pad;
  a.Cin_g = c.Cin / c.g; a.Cout_g = c.Cout / c.g;
  a.phase_stride = p.phase_stride;
  const long long total = p.phase_stride * p.nph;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(pack_kernel, dim3(blocks), dim3(256), 0, st, a);
  EBEN_CHECK_LAUNCH("pack_kernel");
  return EBEN_OK;
}

// ---- tapconv launcher -----------------------------------------------------------------------
template <int WM, int WN, int FM, int FN>
static int launch_cfg(const TapArgs& a, int nblocks, size_t lds, hipStream_t st) {
  static LdsAttrOnce attr_once;
  auto kern = tapconv_kernel<WM, WN, FM, FN>;
  {
    const hipError_t e = lds_attr_once(attr_once, reinterpret_cast<const void*>(kern));
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(tapconv)");
  }
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(WM * WN * 64), lds, st, a);
  EBEN_CHECK_LAUNCH("tapconv_kernel");
  return EBEN_OK;
}

static int launch_tap(const Canon& c, const TapPlan& p, const TapIO& io, int reflect, hipStream_t st) {
  if (p.lds_bytes > 160 * 1024) return fail(EBEN_EUNSUPPORTED, "tapconv tile needs %zu B of LDS", p.lds_bytes);
  TapArgs a;
  a.x = io.x; a.xmask = io.xmask; a.wp = io.wp; a.bias = io.bias; a.res = io.res; a.emask = io.emask; a.y = io.y;
  a.B = c.B; a.G = p.G; a.Cg = p.Cg; a.Mg = p.Mg; a.Mp = p.Mp; a.Cx = p.Cx; a.Cy = p.Cy; a.Lx = p.Lx; a.Ly = p.Ly;
  a.S = p.S; a.OS = p.OS; a.dstep = p.dstep; a.J = p.J; a.KCpad = p.KCpad; a.CI_T = p.CI_T; a.ncc = p.ncc;
  a.mode = p.mode; a.off0 = p.off0; a.nt = p.nt;
  a.ps_pad = p.ps_pad; a.ps_k = c.k; a.ps_d = c.d; a.ps_kstep = p.kstep;
  a.reflect = reflect; a.in_mode = io.in_mode;
  a.in_slope = io.in_slope; a.out_slope = io.out_slope; a.res_slope = io.res_slope; a.emask_slope = io.emask_slope;
  a.accumulate = io.accumulate;
  a.res_rows = io.res_rows; a.em_seg = io.em_seg;
  for (int i = 0; i < 4; ++i) a.em_map[i] = io.em_map[i];
  a.PLEN = p.PLEN; a.CSTRIDE = p.CSTRIDE;
  a.s_magic = p.S > 1 ? (unsigned)((0x100000000ull + p.S - 1) / p.S) : 0u;
  a.ntt = p.ntt; a.nmt = p.nmt; a.nph = p.nph;
  a.phase_stride = p.phase_stride;
  const long long nb = (long long)p.ntt * c.B * p.nph * p.nmt * p.G;
  if (nb <= 0 || nb > 0x7fffffffLL) return fail(EBEN_EINVAL, "tapconv grid of %lld blocks", nb);
  return launch_cfg<1, 4, 1, 4>(a, (int)nb, p.lds_bytes, st);
}

// Single-output-channel convolution (the logits layers: 1024->1 and 768->1, k=3): a channel
// reduction, not a GEMM.  One block = 64 output positions of one batch item, 16 waves splitting the
// channels; lane = position (coalesced rows, taps hit L1), partial sums combined through LDS in a
// fixed order.  Weights are read from the packed layout (scale already folded in).
__global__ __launch_bounds__(1024) void conv_m1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                           const float* __restrict__ bias, float* __restrict__ y, int C, int Lx,
                                                           int Ly, int J, int pad, int dil, int CI_T, int KCpad, int Mp,
                                                           float out_slope) {
  __shared__ float part[16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int b = blockIdx.y, t = blockIdx.x * 64 + lane;
  const float* xb = x + (long long)b * C * Lx;
  float acc = 0.f;
  if (J == 3) {
    // the two logits layers (k = 3): four channels' loads in flight per round trip, addresses clamped instead of branched on,
    // the channel-chunk index carried instead of divided for; same FMA order as the generic loop below
    int q[3], ok[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int qq = t - pad + j * dil;
      ok[j] = (int)(qq >= 0) & (int)(qq < Lx);
      q[j] = ok[j] ? qq : 0;
    }
    int cc = w / CI_T, cl = w - cc * CI_T;
    for (int c = w; c < C; c += 64) {
      float xv[4][3], wv[4][3];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int cu = c + 16 * u;
        const bool live = cu < C;
        const float* xr = xb + (long long)(live ? cu : w) * Lx;
        const float* wc = wp + ((long long)cc * KCpad + cl) * Mp;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          xv[u][j] = xr[q[j]];
          wv[u][j] = live ? wc[(long long)j * CI_T * Mp] : 0.f;
          if (!ok[j]) xv[u][j] = 0.f;
        }
        cl += 16;
        while (cl >= CI_T) { cl -= CI_T; ++cc; }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          if (c + 16 * u < C) acc = fmaf(wv[u][j], xv[u][j], acc);
    }
  } else {
    for (int c = w; c < C; c += 16) {
      const int cc = c / CI_T, cl = c - cc * CI_T;
      const float* xr = xb + (long long)c * Lx;
      const float* wc = wp + ((long long)cc * KCpad + cl) * Mp;
      for (int j = 0; j < J; ++j) {
        const int q = t - pad + j * dil;
        const float xv = (q >= 0 && q < Lx) ? xr[q] : 0.f;
        acc = fmaf(wc[(long long)j * CI_T * Mp], xv, acc);
      }
    }
  }
  part[w][lane] = acc;
  __syncthreads();
  if (w == 0 && t < Ly) {
    float s = bias ? bias[0] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += part[i][lane];
    y[(long long)b * Ly + t] = lrelu(s, out_slope);
  }
}

// reflect-pad fold: dx[u] = D[u+pl] + D[pl-u] (1<=u<=pl) + D[pl+2(L-1)-u] (L-1-pr<=u<=L-2) (+ pre), then mask (+ post)
__global__ __launch_bounds__(256) void fold_kernel(const float* __restrict__ D, const float* __restrict__ mask, float* __restrict__ dx,
                                                   long long rows, int L, int pl, int pr, float slope, int accumulate,
                                                   const float* __restrict__ pre, const float* __restrict__ post) {
  const int Lp = L + pl + pr;
  const long long total = rows * L;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long row = i / L;
    const int u = (int)(i - row * L);
    const float* d = D + row * Lp;
    float v = d[u + pl];
    if (u >= 1 && u <= pl) v += d[pl - u];
    if (u >= L - 1 - pr && u <= L - 2) v += d[pl + 2 * (L - 1) - u];
    if (pre) v += pre[i];
    if (mask) v *= dlrelu(mask[i], slope);
    if (post) v += post[i];
    if (accumulate) v += dx[i];
    dx[i] = v;
  }
}

}  // namespace eben

using namespace eben;

namespace eben {
int tap_generation(const Canon& c, int dir) {
  static const int thin_first = getenv("EBEN_THIN_FIRST") ? atoi(getenv("EBEN_THIN_FIRST")) : 0;
  if (c.bl) return tap3_applicable(c, dir) ? 4 : 0;  // bundle layout: the bf16 tap-conv or nothing (eben_bl_* report EBEN_EUNSUPPORTED)
  if (c.bf16 && tap3_applicable(c, dir)) return 4;   // layers the bf16 kernel does not cover keep their fp32 kernel
  const int t2 = tap2_applicable(c, dir), th = thin_applicable(c, dir);
  if (th && (thin_first || !t2)) return 3;
  if (t2) return 2;
  return 1;
}
}  // namespace eben

extern "C" size_t eben_conv1d_packed_floats(const EbenConv1dDesc* d, int which) {
  Canon c;
  if (canon_from_desc(d, &c) != EBEN_OK) return 0;
  TapPlan p;
  // which: 0 = the layer's forward, 1 = the layer's input gradient
  const int dir = d->transposed ? 1 - which : which;
  if (which == 0 && gc_applicable(c, dir)) return gc_packed_floats(c, dir);
  const int gen = tap_generation(c, dir);
  if (gen == 2) return tap2_packed_floats(c, dir);
  if (gen == 3) return thin_packed_floats(c, dir);
  if (gen == 4) return tap3_packed_floats(c, dir);
  make_plan(c, dir, &p);
  return p.packed_floats;
}

extern "C" int eben_conv1d_kernel_generation(const EbenConv1dDesc* d, int which) {
  Canon c;
  if (canon_from_desc(d, &c) != EBEN_OK) return 0;
  if (which == 0 && gc_applicable(c, d->transposed ? 1 : 0)) return 5;
  const int dir = d->transposed ? 1 - which : which;
  const int gen = tap_generation(c, dir);
  return gen == 4 && tap3_is_big(c, dir) ? 6 : gen;
}

extern "C" int eben_conv1d_pack(const EbenConv1dDesc* d, const float* v, const float* scale, float* wp_fwd, float* wp_bwd, void* stream) {
  Canon c;
  int rc = canon_from_desc(d, &c);
  if (rc) return rc;
  EBEN_REQUIRE(v != nullptr, "null weight");
  for (int which = 0; which < 2; ++which) {
    float* dst = which == 0 ? wp_fwd : wp_bwd;
    if (!dst) continue;
    const int dir = d->transposed ? 1 - which : which;
    if (which == 0 && gc_applicable(c, dir)) {
      if ((rc = gc_pack(c, dir, v, scale, dst, as_stream(stream)))) return rc;
      continue;
    }
    const int gen = tap_generation(c, dir);
    if (gen != 1) {
      rc = gen == 2 ? tap2_pack(c, dir, v, scale, dst, as_stream(stream))
         : gen == 4 ? tap3_pack(c, dir, v, scale, dst, as_stream(stream)) : thin_pack(c, dir, v, scale, dst, as_stream(stream));
      if (rc) return rc;
      continue;
    }
    TapPlan p;
    make_plan(c, dir, &p);
    rc = launch_pack(c, p, v, scale, dst, as_stream(stream));
    if (rc) return rc;
  }
  return EBEN_OK;
}

extern "C" int eben_conv1d_pack_multi(const EbenPackJob* jobs, int n, void* stream) {
  EBEN_REQUIRE(n >= 0 && (n == 0 || jobs != nullptr), "bad pack job list");
  // images of the bf16 tap-conv family (most of a step's ~150 pack launches) are gathered into launches of 16 jobs; the other
  // families keep their own launches
  constexpr int CAP = 512;
  static thread_local Canon cs[CAP];
  static thread_local int dirs[CAP];
  static thread_local const float* ws[CAP];
  static thread_local const float* scs[CAP];
  static thread_local float* wps[CAP];
  int m = 0;
  auto flush = [&]() -> int {
    const int rc = m ? tap3_pack_multi(cs, dirs, ws, scs, wps, m, as_stream(stream)) : EBEN_OK;
    m = 0;
    return rc;
  };
  for (int i = 0; i < n; ++i) {
    const EbenPackJob& jb = jobs[i];
    Canon c;
    int rc = canon_from_desc(&jb.desc, &c);
    if (rc) return rc;
    EBEN_REQUIRE(jb.v != nullptr, "null weight");
    for (int which = 0; which < 2; ++which) {
      float* dst = which == 0 ? jb.wp_fwd : jb.wp_bwd;
      if (!dst) continue;
      const int dir = jb.desc.transposed ? 1 - which : which;
      if (!(which == 0 && gc_applicable(c, dir)) && tap_generation(c, dir) == 4) {
        if (m == CAP && (rc = flush())) return rc;
        cs[m] = c; dirs[m] = dir; ws[m] = jb.v; scs[m] = jb.scale; wps[m] = dst;
        ++m;
      } else {
        rc = eben_conv1d_pack(&jb.desc, jb.v, jb.scale, which == 0 ? dst : nullptr, which == 1 ? dst : nullptr, stream);
        if (rc) return rc;
      }
    }
  }
  return flush();
}

static int conv1d_fwd_impl(const EbenConv1dDesc* d, const float* x, const float* wp_fwd, const float* bias,
                           const float* residual, float res_slope, float* y, void* stream) {
  Canon c;
  int rc = canon_from_desc(d, &c);
  if (rc) return rc;
  EBEN_REQUIRE(x && wp_fwd && y, "null pointer in conv1d_fwd");
  TapPlan p;
  make_plan(c, d->transposed ? 1 : 0, &p);
  if (!d->transposed && c.Cout == 1 && c.g == 1 && c.s == 1 && !c.reflect && d->in_slope == 1.f && !residual && tap_generation(c, 0) == 1) {
    hipLaunchKernelGGL(conv_m1_fwd_kernel, dim3(ceil_div(c.Lout, 64), c.B), dim3(1024), 0, as_stream(stream), x, wp_fwd, bias, y,
                       c.Cin, c.Lin, c.Lout, c.k, c.pl, c.d, p.CI_T, p.KCpad, p.Mp, d->out_slope);
    EBEN_CHECK_LAUNCH("conv_m1_fwd_kernel");
    return EBEN_OK;
  }
  TapIO io{};
  io.x = x; io.in_mode = 0; io.in_slope = d->in_slope; io.wp = wp_fwd; io.bias = bias;
  io.res = residual; io.res_slope = res_slope; io.emask = nullptr; io.emask_slope = 1.f;
  io.out_slope = d->out_slope; io.y = y; io.accumulate = 0;
  if (gc_applicable(c, d->transposed ? 1 : 0)) return gc_launch(c, d->transposed ? 1 : 0, io, as_stream(stream));
  const int fgen = tap_generation(c, d->transposed ? 1 : 0);
  if (fgen == 2) return tap2_launch(c, d->transposed ? 1 : 0, io, c.reflect && !d->transposed, as_stream(stream));
  if (fgen == 3) return thin_launch(c, d->transposed ? 1 : 0, io, c.reflect && !d->transposed, as_stream(stream));
  if (fgen == 4) return tap3_launch(c, d->transposed ? 1 : 0, io, c.reflect && !d->transposed, as_stream(stream));
  return launch_tap(c, p, io, c.reflect && !d->transposed, as_stream(stream));
}

extern "C" int eben_conv1d_fwd(const EbenConv1dDesc* d, const float* x, const float* wp_fwd, const float* bias,
                               const float* residual, float* y, void* stream) {
  return conv1d_fwd_impl(d, x, wp_fwd, bias, residual, 1.f, y, stream);
}

extern "C" int eben_conv1d_fwd_res(const EbenConv1dDesc* d, const float* x, const float* wp_fwd, const float* bias,
                                   const float* residual, float res_slope, float* y, void* stream) {
  EBEN_REQUIRE(residual != nullptr, "conv1d_fwd_res without a residual");
  return conv1d_fwd_impl(d, x, wp_fwd, bias, residual, res_slope, y, stream);
}

extern "C" size_t eben_conv1d_bwd_dx_workspace(const EbenConv1dDesc* d) {
  Canon c;
  if (canon_from_desc(d, &c) != EBEN_OK) return 0;
  if (!d->transposed && c.reflect) return sizeof(float) * (size_t)c.B * c.Cin * (c.Lin + c.pl + c.pr);
  return 0;
}

static int conv1d_bwd_dx_impl(const EbenConv1dDesc* d, const float* dy, const float* y, const float* wp_bwd, const float* x,
                              const float* res_pre, const float* res_post, float* dx, int accumulate, void* workspace, size_t ws_bytes,
                              void* stream) {
  Canon c;
  int rc = canon_from_desc(d, &c);
  if (rc) return rc;
  EBEN_REQUIRE(dy && wp_bwd && dx, "null pointer in conv1d_bwd_dx");
  EBEN_REQUIRE(d->out_slope == 1.f || y, "y is required to differentiate the fused output activation");
  EBEN_REQUIRE(d->in_slope == 1.f || x, "x is required to differentiate the fused input activation");
  hipStream_t st = as_stream(stream);
  TapPlan p;
  const int dir = d->transposed ? 0 : 1;
  const int gen = tap_generation(c, dir);
  make_plan(c, dir, &p);
  TapIO io{};
  io.x = dy; io.wp = wp_bwd; io.bias = nullptr; io.res = nullptr; io.res_slope = 1.f; io.out_slope = 1.f;
  if (d->out_slope != 1.f) { io.in_mode = 1; io.xmask = y; io.in_slope = d->out_slope; }
  else { io.in_mode = 0; io.in_slope = 1.f; }
  const bool fold = !d->transposed && c.reflect;
  if (!fold) {
    if (res_post) return fail(EBEN_EUNSUPPORTED, "bwd_dx: an addend behind the input-activation mask needs the reflect-fold pass");
    io.res = res_pre;
    io.emask = d->in_slope != 1.f ? x : nullptr; io.emask_slope = d->in_slope;
    io.y = dx; io.accumulate = accumulate;
    return gen == 2 ? tap2_launch(c, dir, io, 0, st) : gen == 3 ? thin_launch(c, dir, io, 0, st)
         : gen == 4 ? tap3_launch(c, dir, io, 0, st) : launch_tap(c, p, io, 0, st);
  }
  const size_t need = eben_conv1d_bwd_dx_workspace(d);
  if (!workspace || ws_bytes < need) return fail(EBEN_EWORKSPACE, "bwd_dx needs %zu workspace bytes, got %zu", need, ws_bytes);
  io.emask = nullptr; io.emask_slope = 1.f; io.y = static_cast<float*>(workspace); io.accumulate = 0;
  rc = gen == 2 ? tap2_launch(c, dir, io, 0, st) : gen == 3 ? thin_launch(c, dir, io, 0, st)
     : gen == 4 ? tap3_launch(c, dir, io, 0, st) : launch_tap(c, p, io, 0, st);
  if (rc) return rc;
  const long long rows = (long long)c.B * c.Cin;
  long long blocks = (rows * c.Lin + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(fold_kernel, dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const float*>(workspace),
                     d->in_slope != 1.f ? x : nullptr, dx, rows, c.Lin, c.pl, c.pr, d->in_slope, accumulate, res_pre, res_post);
  EBEN_CHECK_LAUNCH("fold_kernel");
  return EBEN_OK;
}

extern "C" int eben_conv1d_bwd_dx(const EbenConv1dDesc* d, const float* dy, const float* y, const float* wp_bwd,
                                  const float* x, float* dx, int accumulate, void* workspace, size_t ws_bytes, void* stream) {
  return conv1d_bwd_dx_impl(d, dy, y, wp_bwd, x, nullptr, nullptr, dx, accumulate, workspace, ws_bytes, stream);
}

extern "C" int eben_conv1d_bwd_dx_res(const EbenConv1dDesc* d, const float* dy, const float* y, const float* wp_bwd, const float* x,
                                      const float* res_pre, const float* res_post, float* dx, void* workspace, size_t ws_bytes,
                                      void* stream) {
  return conv1d_bwd_dx_impl(d, dy, y, wp_bwd, x, res_pre, res_post, dx, 0, workspace, ws_bytes, stream);
}

// Batched input gradient for several right-hand sides that share one set of saved activations
// (the discriminator backward of the fused train step: feature-matching, adversarial, fake and real
// seeds stacked along the batch axis):
//   dx[b] = ( conv^T(g[b]) + (b < res_rows ? res[b] : 0) ) * lrelu'(mask[map(b)], mask_slope)
// g is consumed as is: the producer already applied its activation derivative in ITS epilogue, so no
// kernel on this path reads a mask on load.
static int bwd_dx_batched(const EbenConv1dDesc* d, const float* g, const float* wp_bwd, const float* res, int res_rows, const float* fm_sums,
                          float fm_gs, const float* mask, float mask_slope, int seg, const int* seg_map, float* dx, void* stream);
extern "C" int eben_conv1d_bwd_dx_ex(const EbenConv1dDesc* d, const float* g, const float* wp_bwd, const float* res, int res_rows,
                                     const float* mask, float mask_slope, int seg, const int* seg_map, float* dx, void* stream) {
  return bwd_dx_batched(d, g, wp_bwd, res, res_rows, nullptr, 0.f, mask, mask_slope, seg, seg_map, dx, stream);
}
// The same launch with the feature-matching gradient of the embedding formed in the epilogue instead of read from a buffer a separate
// kernel wrote (feature_loss.py:40-47: d/da [ sum|a - b| / sum|a| ] = sgn(a - b) / s2 - s1 sgn(a) / s2^2):
//   dx[b] = ( conv^T(g[b]) + (b < fm_rows ? fm_gs (sgn(mask[b] - ref[b]) / s2 - s1 sgn(mask[b]) / s2^2) : 0) ) * lrelu'(mask[map(b)])
// with (s1, s2) = fm_sums[0..1] read on the device.  Only the thin and the bf16 tap-conv kernels (generations 3 / 4, the ones the
// discriminators' input gradients run on) carry this epilogue: EBEN_EUNSUPPORTED otherwise (use eben_fm_bwd + bwd_dx_ex).
extern "C" int eben_conv1d_bwd_dx_fm(const EbenConv1dDesc* d, const float* g, const float* wp_bwd, const float* ref, int fm_rows,
                                     const float* fm_sums, float fm_gs, const float* mask, float mask_slope, int seg, const int* seg_map,
                                     float* dx, void* stream) {
  EBEN_REQUIRE(ref && fm_sums && mask && fm_rows > 0, "bad feature-matching arguments in conv1d_bwd_dx_fm");
  return bwd_dx_batched(d, g, wp_bwd, ref, fm_rows, fm_sums, fm_gs, mask, mask_slope, seg, seg_map, dx, stream);
}
static int bwd_dx_batched(const EbenConv1dDesc* d, const float* g, const float* wp_bwd, const float* res, int res_rows, const float* fm_sums,
                          float fm_gs, const float* mask, float mask_slope, int seg, const int* seg_map, float* dx, void* stream) {
  Canon c;
  int rc = canon_from_desc(d, &c);
  if (rc) return rc;
  EBEN_REQUIRE(g && wp_bwd && dx, "null pointer in conv1d_bwd_dx_ex");
  EBEN_REQUIRE(!(c.reflect && !d->transposed), "bwd_dx_ex does not fold reflect padding (pad explicitly)");
  EBEN_REQUIRE(seg >= 0 && (seg == 0 || (seg_map && c.B <= 4 * seg)), "bad batch segment map");
  const int dir = d->transposed ? 0 : 1;
  TapIO io{};
  io.x = g; io.in_mode = 0; io.in_slope = 1.f; io.wp = wp_bwd; io.out_slope = 1.f;
  io.res = res; io.res_slope = 1.f; io.res_rows = res_rows;
  io.emask = mask; io.emask_slope = mask_slope; io.em_seg = mask ? seg : 0;
  for (int i = 0; i < 4; ++i) io.em_map[i] = (seg > 0 && seg_map) ? seg_map[i] : i;
  io.y = dx; io.accumulate = 0;
  const int gen = tap_generation(c, dir);
  io.fm_sums = fm_sums; io.fm_gs = fm_gs;
  if (fm_sums && gen != 3 && gen != 4) return fail(EBEN_EUNSUPPORTED, "conv1d_bwd_dx_fm: kernel generation %d has no feature-matching epilogue", gen);
  hipStream_t st = as_stream(stream);
  if (gen == 2) return tap2_launch(c, dir, io, 0, st);
  if (gen == 3) return thin_launch(c, dir, io, 0, st);
  if (gen == 4) return tap3_launch(c, dir, io, 0, st);
  TapPlan p;
  make_plan(c, dir, &p);
  return launch_tap(c, p, io, 0, st);
}

// ---- bundle layout (EBEN_LAYOUT_BL): the discriminator layers between the chain heads and the logits ------------------------------
// Input gradient of a STRIDED Conv1d as "phases as rows" (melgan_discriminator.py:97-118: k 41, stride 4, 4 groups).  The phase-scatter
// form (tap3 mode 1) runs one stride-1 sub-convolution per output phase: `stride` blocks stage the same window of dy, each with a
// quarter of the taps, and a layer with few channels per group (16 -> 64: four input channels per group) fills an eighth of its MFMA
// rows.  Written for all phases at once,
//     dx[c, S q + ph] = sum_{co, u} W[co, c, ph + pad - S u] dy[co, q + u],
// the input gradient IS a stride-1 Conv1d from the Cout channels of dy to S Cin output rows (ph, c) with taps u = umin .. umax (11 for
// k 41 / S 4) whose result is stored depth-to-space: MelGAN L1 / L2 become grouped 64 -> 64 / 256 -> 256 k 11 convolutions, dy staged
// once, full row tiles.  The primed weights W'[(ph, c)][co][u] are a gather of the layer's weights (pr_weights_kernel), packed as an
// ordinary forward image of the primed layer; tap3's bundle epilogue maps logical bundle -> (physical bundle, phase) (Tap3Args.pr_*).
namespace eben {
struct PrGeom { int ok, S, umin, kq, Lq, fold, cbg, order; };   // order 1: rows (channel bundle, phase, channel in bundle) -- tap4_kernel's coalesced depth-to-space epilogue

static int pr_floordiv(int a, int b) { int q = a / b; if ((a % b != 0) && ((a < 0) != (b < 0))) --q; return q; }

static PrGeom pr_geometry(const Canon& c, Canon* cp) {
  PrGeom g{};
  static const int min_s = getenv("EBEN_PR_MIN_STRIDE") ? atoi(getenv("EBEN_PR_MIN_STRIDE")) : 4;
  static const int max_d = getenv("EBEN_PR_MAX_DIL") ? atoi(getenv("EBEN_PR_MAX_DIL")) : 1;
  // stride 2 (the PQMF-band layers) in bundle-major row order on tap3_kernel, up to this dilation (0: off).  [MI355X, 128 rows] the five
  // stride-2 input gradients of a chain at dilation 1: 0.063 / 0.070 / 0.055 / 0.062 / 0.070 -> 0.053 / 0.052 / 0.045 / 0.046 / 0.064 ms,
  // at dilation 2 (7 primed taps for 4 + 3): 0.071 / 0.085 / 0.064 / 0.077 / 0.089 -> 0.061 / 0.063 / 0.050 / 0.057 / 0.085; at dilation
  // 3 (10 primed taps) the extra products cost what the whole-unit epilogue saves: 0.057 / 0.066 / 0.054 / 0.063 / 0.071 -> 0.064 /
  // 0.069 / 0.054 / 0.065 / 0.070
  static const int s2_max_d = getenv("EBEN_PR2_S2_MAX_DIL") ? atoi(getenv("EBEN_PR2_S2_MAX_DIL")) : 2;
  const bool s2 = c.s == 2 && c.d <= s2_max_d;
  if (!c.bl || c.np != 1 || c.reflect || (c.d > max_d && !s2) || c.s < 2 || (c.s < min_s && !s2) || c.s > 8 || (c.k - 1) * c.d + 1 <= c.s || c.pl > (c.k - 1) * c.d) return g;
  const int Cg = c.Cin / c.g;
  if ((c.Cin & 7) || (c.Cout & 7)) return g;
  g.S = c.s;
  // dilation d: tap j sits at offset j d, i.e. W' is nonzero where ph + pad - S u is a multiple of d inside [0, (k - 1) d]
  g.umin = -pr_floordiv((c.k - 1) * c.d - c.pl, c.s);     // ceil((pad - (k - 1) d) / S): phase 0, last tap
  const int umax = pr_floordiv(c.s - 1 + c.pl, c.s);      // phase S - 1, tap 0
  g.kq = umax - g.umin + 1;
  g.Lq = ceil_div(c.Lin, c.s);
  g.fold = (Cg & 7) != 0;                                  // groups that do not start on bundles: one dense (block-diagonal) contraction
  g.cbg = g.fold ? c.Cin / 8 : Cg / 8;
  Canon q = c;
  q.Cin = c.Cout; q.Cout = c.s * c.Cin; q.Lin = c.Lout; q.Lout = g.Lq;
  q.k = g.kq; q.s = 1; q.d = 1; q.g = g.fold ? 1 : c.g;
  q.pl = -g.umin;
  q.pr = g.Lq - c.Lout + g.kq - 1 - q.pl;
  if (q.pl < 0 || q.pr < 0) return g;
  q.reflect = 0; q.xsplit_dir = -1;
  // [MI355X, 128 rows] MelGAN L1 / L2 (4 / 16 channels per group: 64 primed rows per group) 0.336 / 0.310 -> 0.239 / 0.210 ms; L3 / L4 (64 /
  // 256 channels per group: 256 / 1024 primed rows, full row tiles in either form) 0.414 / 0.412 -> 0.531 / 0.524: the form is for the layers
  // whose phases leave row tiles empty
  static const int max_rows = getenv("EBEN_PR_MAX_ROWS") ? atoi(getenv("EBEN_PR_MAX_ROWS")) : 64;
  // Wider layers (MelGAN L3 / L4: 256 / 1024 primed rows per group) take the form on tap4_kernel (bigtap.hip) with the rows ordered
  // (channel bundle, phase, channel in bundle): a 32-row MFMA tile is then ONE bundle at the four phases of a position, i.e. 64
  // contiguous bytes per column -- the phase-scatter form writes (and reads its mask as) 8-byte pieces 64 bytes apart, which a
  // block per CU cannot hide ([MI355X] phase-scatter on tap4: 0.416 / 0.402 -> 0.506 / 0.444 ms)
  static const int big_pr = getenv("EBEN_PR_BIG") ? atoi(getenv("EBEN_PR_BIG")) : 1;
  // ... and the narrow ones (MelGAN L1 / L2: 64 primed rows per group, tap3_kernel) take the same row order: their mask / feature-matching
  // operands and results move as whole units too ([MI355X] those loads were 36-43 % of the order-0 launches)
  static const int small_pr2 = getenv("EBEN_PR2_TAP3") ? atoi(getenv("EBEN_PR2_TAP3")) : 1;
  g.order = (big_pr && c.s == 4 && c.d == 1 && (tap3_is_big(q, 0) || (small_pr2 && (q.Cout / q.g) % 32 == 0))) ? 1 : 0;
  if (s2) {   // two bundles per 32-row tile: whole bundles at both phases, tap3_kernel only
    if (tap3_is_big(q, 0) || (q.Cout / q.g) % 16 != 0) return g;
    g.order = 1;
  }
  if (q.Cout / q.g > max_rows && !g.order) return g;
  if (!tap3_applicable(q, 0)) return g;
  if (cp) *cp = q;
  g.ok = 1;
  return g;
}

// W'[(g, ph, c)][co][ui] (groups kept) or [(ph, ci)][co][ui] (folded, zero across groups) = scale[co] v[co][c][ph + pad - S (ui + umin)]
__global__ __launch_bounds__(256) void pr_weights_kernel(const float* __restrict__ v, const float* __restrict__ scale, float* __restrict__ wq,
                                                          int Cin, int Cout, int G, int k, int S, int pad, int umin, int kq, int fold, int dil, int order) {
  const int Cg = Cin / G, Mg = Cout / G;
  const int cin_q = fold ? Cout : Mg;                      // input channels per group of the primed layer
  const long long total = (long long)S * Cin * cin_q * kq;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ui = (int)(i % kq);
    long long r = i / kq;
    const int cq = (int)(r % cin_q);
    const int row = (int)(r / cin_q);
    int ph, ci, co;
    if (fold && order) { const int lb = row >> 3, cb = lb / S; ph = lb - cb * S; ci = cb * 8 + (row & 7); co = cq; }
    else if (fold) { ph = row / Cin; ci = row - ph * Cin; co = cq; }
    else if (order == 0) { const int g = row / (S * Cg), rr = row - g * S * Cg; ph = rr / Cg; ci = g * Cg + (rr - ph * Cg); co = g * Mg + cq; }
    else {   // (group, channel bundle, phase, channel in bundle)
      const int g = row / (S * Cg), rr = row - g * S * Cg, lb = rr >> 3, cb = lb / S;
      ph = lb - cb * S; ci = g * Cg + cb * 8 + (rr & 7); co = g * Mg + cq;
    }
    const int off = ph + pad - S * (ui + umin);              // = j dil for the tap this entry stands for, if any
    const int j = off / dil;
    float w = 0.f;
    if (off >= 0 && j * dil == off && j < k && co / Mg == ci / Cg) w = v[((long long)co * Cg + (ci % Cg)) * k + j] * (scale ? scale[co] : 1.f);
    wq[i] = w;
  }
}

static void pr_desc_from_canon(const EbenConv1dDesc* d, const Canon& q, EbenConv1dDesc* o) {
  *o = *d;
  o->c_in = q.Cin; o->c_out = q.Cout; o->l_in = q.Lin; o->l_out = q.Lout; o->ksize = q.k; o->stride = 1; o->dilation = 1; o->groups = q.g;
  o->pad_l = q.pl; o->pad_r = q.pr; o->pad_mode = EBEN_PAD_ZERO; o->transposed = 0; o->in_slope = 1.f; o->out_slope = 1.f;
}
}  // namespace eben

extern "C" int eben_bl_dx_pr_desc(const EbenConv1dDesc* d, EbenConv1dDesc* primed) {
  Canon c, q;
  int rc = canon_from_desc(d, &c);
  if (rc) return rc;
  if (d->transposed || !pr_geometry(c, &q).ok) return fail(EBEN_EUNSUPPORTED, "eben_bl_dx_pr_desc: the layer's input gradient has no phases-as-rows form");
  EBEN_REQUIRE(primed != nullptr, "null output descriptor");
  pr_desc_from_canon(d, q, primed);
  return EBEN_OK;
}

extern "C" int eben_bl_dx_pr_weights(const EbenConv1dDesc* d, const float* v, const float* scale, float* w_primed, void* stream) {
  Canon c, q;
  int rc = canon_from_desc(d, &c);
  if (rc) return rc;
  const PrGeom g = d->transposed ? PrGeom{} : pr_geometry(c, &q);
  if (!g.ok) return fail(EBEN_EUNSUPPORTED, "eben_bl_dx_pr_weights: the layer's input gradient has no phases-as-rows form");
  EBEN_REQUIRE(v && w_primed, "null pointer in eben_bl_dx_pr_weights");
  const long long total = (long long)q.Cout * (q.Cin / q.g) * q.k;
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pr_weights_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), v, scale, w_primed, c.Cin, c.Cout, c.g, c.k, c.s, c.pl,
                     g.umin, g.kq, g.fold, c.d, g.order);
  EBEN_CHECK_LAUNCH("pr_weights_kernel");
  return EBEN_OK;
}

extern "C" int eben_bl_conv1d_bwd_dx_pr(const EbenConv1dDesc* d, const void* g_hi, const float* wp_primed_fwd, const void* act_hi, const void* act_lo,
                                        float mask_slope, int seg, const int* seg_map, int fm_rows, int ref_row_offset, const float* fm_sums,
                                        float fm_gs, void* dx_hi, void* dx_lo, void* stream) {
  return eben_bl_conv1d_bwd_dx_pr_c(d, g_hi, wp_primed_fwd, act_hi, act_lo, nullptr, mask_slope, seg, seg_map, fm_rows, ref_row_offset, fm_sums, fm_gs, dx_hi,
                                    dx_lo, stream);
}

extern "C" int eben_bl_conv1d_bwd_dx_pr_c(const EbenConv1dDesc* d, const void* g_hi, const float* wp_primed_fwd, const void* act_hi, const void* act_lo,
                                          const void* fm_codes, float mask_slope, int seg, const int* seg_map, int fm_rows, int ref_row_offset,
                                          const float* fm_sums, float fm_gs, void* dx_hi, void* dx_lo, void* stream) {
  Canon c, q;
  int rc = canon_from_desc(d, &c);
  if (rc) return rc;
  const PrGeom g = d->transposed ? PrGeom{} : pr_geometry(c, &q);
  if (!g.ok) return fail(EBEN_EUNSUPPORTED, "eben_bl_conv1d_bwd_dx_pr: the layer's input gradient has no phases-as-rows form");
  EBEN_REQUIRE(g_hi && wp_primed_fwd && dx_hi, "null pointer in eben_bl_conv1d_bwd_dx_pr");
  EBEN_REQUIRE(seg >= 0 && (seg == 0 || (seg_map && c.B <= 4 * seg)), "bad batch segment map");
  EBEN_REQUIRE(fm_rows == 0 || (fm_sums && act_hi && act_lo && fm_rows > 0), "feature-matching rows need the sums and both planes of the embedding");
  TapIO io{};
  io.in_slope = 1.f; io.wp = wp_primed_fwd; io.out_slope = 1.f; io.res_slope = 1.f;
  io.res_rows = fm_rows; io.fm_sums = fm_rows > 0 ? fm_sums : nullptr; io.fm_gs = fm_gs;
  io.emask_slope = mask_slope; io.em_seg = act_hi ? seg : 0;
  for (int i = 0; i < 4; ++i) io.em_map[i] = (seg > 0 && seg_map) ? seg_map[i] : i;
  io.xh = g_hi; io.yh = dx_hi; io.yl = dx_lo; io.eh = act_hi; io.el = act_lo; io.bl_ref_off = ref_row_offset;
  io.ec = fm_rows > 0 ? fm_codes : nullptr;
  io.pr_S = g.S; io.pr_cbg = g.cbg; io.pr_Ly = c.Lin; io.pr_CBy = c.Cin / 8; io.pr_order = g.order;
  return tap3_launch(q, 0, io, 0, as_stream(stream));
}

extern "C" int eben_bl_conv1d_fwd(const EbenConv1dDesc* d, const void* x_hi, const void* x_lo, const float* wp_fwd, const float* bias,
                                  void* y_hi, void* y_lo, void* stream) {
  Canon c;
  int rc = canon_from_desc(d, &c);
  if (rc) return rc;
  EBEN_REQUIRE(c.bl && !d->transposed && d->in_slope == 1.f, "eben_bl_conv1d_fwd: a Conv1d descriptor with EBEN_LAYOUT_BL and no input activation");
  EBEN_REQUIRE(x_hi && wp_fwd && y_hi, "null pointer in eben_bl_conv1d_fwd");
  if (tap_generation(c, 0) != 4) return fail(EBEN_EUNSUPPORTED, "eben_bl_conv1d_fwd: layer not covered by the bundle-layout tap-conv");
  TapIO io{};
  io.in_slope = 1.f; io.wp = wp_fwd; io.bias = bias; io.res_slope = 1.f; io.emask_slope = 1.f; io.out_slope = d->out_slope;
  io.xh = x_hi; io.xl = x_lo; io.yh = y_hi; io.yl = y_lo;
  return tap3_launch(c, 0, io, 0, as_stream(stream));
}

extern "C" int eben_bl_conv1d_bwd_dx(const EbenConv1dDesc* d, const void* g_hi, const float* wp_bwd, const void* act_hi, const void* act_lo,
                                     float mask_slope, int seg, const int* seg_map, int fm_rows, int ref_row_offset, const float* fm_sums,
                                     float fm_gs, void* dx_hi, void* dx_lo, void* stream) {
  return eben_bl_conv1d_bwd_dx_c(d, g_hi, wp_bwd, act_hi, act_lo, nullptr, mask_slope, seg, seg_map, fm_rows, ref_row_offset, fm_sums, fm_gs, dx_hi, dx_lo,
                                 stream);
}

extern "C" int eben_bl_conv1d_bwd_dx_c(const EbenConv1dDesc* d, const void* g_hi, const float* wp_bwd, const void* act_hi, const void* act_lo,
                                       const void* fm_codes, float mask_slope, int seg, const int* seg_map, int fm_rows, int ref_row_offset,
                                       const float* fm_sums, float fm_gs, void* dx_hi, void* dx_lo, void* stream) {
  Canon c;
  int rc = canon_from_desc(d, &c);
  if (rc) return rc;
  EBEN_REQUIRE(c.bl && !d->transposed, "eben_bl_conv1d_bwd_dx: a Conv1d descriptor with EBEN_LAYOUT_BL");
  EBEN_REQUIRE(g_hi && wp_bwd && dx_hi, "null pointer in eben_bl_conv1d_bwd_dx");
  EBEN_REQUIRE(seg >= 0 && (seg == 0 || (seg_map && c.B <= 4 * seg)), "bad batch segment map");
  EBEN_REQUIRE(fm_rows == 0 || (fm_sums && act_hi && act_lo && fm_rows > 0), "feature-matching rows need the sums and both planes of the embedding");
  if (tap_generation(c, 1) != 4) return fail(EBEN_EUNSUPPORTED, "eben_bl_conv1d_bwd_dx: layer not covered by the bundle-layout tap-conv");
  TapIO io{};
  io.in_slope = 1.f; io.wp = wp_bwd; io.out_slope = 1.f; io.res_slope = 1.f;
  io.res_rows = fm_rows; io.fm_sums = fm_rows > 0 ? fm_sums : nullptr; io.fm_gs = fm_gs;
  io.emask_slope = mask_slope; io.em_seg = act_hi ? seg : 0;
  for (int i = 0; i < 4; ++i) io.em_map[i] = (seg > 0 && seg_map) ? seg_map[i] : i;
  io.xh = g_hi; io.yh = dx_hi; io.yl = dx_lo; io.eh = act_hi; io.el = act_lo; io.bl_ref_off = ref_row_offset;
  io.ec = fm_rows > 0 ? fm_codes : nullptr;
  return tap3_launch(c, 1, io, 0, as_stream(stream));
}
