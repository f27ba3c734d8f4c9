// direct.hip -- the HBM-bound kernels of libeben_hip.so (gfx950): PQMF analysis / synthesis FIR
// banks, the A-weighting FIR, elementwise ops, reflection pads, feature-matching / hinge / STFT
// loss reductions and their gradients, multi-tensor Adam.  All of them stream (batch, channel,
// time) float32 rows with coalesced accesses; reductions are two-level and order-deterministic.
#include "common.h"

#include <cstdlib>

namespace eben {

// ---------------------------------------------------------------------------------------------
// FIR banks (pqmf.py:194-213; auraloss FIRFilter).  One block = 256 consecutive outputs of one
// batch item, input span and taps staged in LDS.
// ---------------------------------------------------------------------------------------------
constexpr int FIR_MAX_W = 1024;  // bands * ntaps

__global__ __launch_bounds__(256) void fir_decimate_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           float* __restrict__ y, int lx, int ly, int bands, int ntaps,
                                                           int stride, int off0) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* ws = smem;                  // bands*ntaps
  float* xs = smem + FIR_MAX_W;      // 255*stride + ntaps
  const int b = blockIdx.y, t0 = blockIdx.x * 256, tid = threadIdx.x;
  for (int i = tid; i < bands * ntaps; i += 256) ws[i] = w[i];
  const int span = 255 * stride + ntaps;
  const long long q0 = (long long)t0 * stride + off0;
  const float* xr = x + (long long)b * lx;
  for (int r = tid; r < span; r += 256) {
    const long long q = q0 + r;
    xs[r] = (q >= 0 && q < lx) ? xr[q] : 0.f;
  }
  __syncthreads();
  const int t = t0 + tid;
  if (t >= ly) return;
  const float* xp = xs + tid * stride;
  for (int k = 0; k < bands; ++k) {
    const float* wk = ws + k * ntaps;
    float acc = 0.f;
    for (int j = 0; j < ntaps; ++j) acc = fmaf(wk[j], xp[j], acc);
    y[((long long)b * bands + k) * ly + t] = acc;
  }
}

__global__ __launch_bounds__(256) void fir_interp_sum_kernel(const float* __restrict__ y, const float* __restrict__ w,
                                                             float* __restrict__ x, int lx, int ly, int bands, int ntaps,
                                                             int stride, int off0, int tile_t) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* ws = smem;                 // bands*ntaps
  float* ys = smem + FIR_MAX_W;     // bands*tile_t
  const int b = blockIdx.y, u0 = blockIdx.x * 256, tid = threadIdx.x;
  for (int i = tid; i < bands * ntaps; i += 256) ws[i] = w[i];
  // t = (u - off0 - j)/stride for j in [0, ntaps): floor-division lower bound of the tile
  long long lo = (long long)u0 - off0 - (ntaps - 1);
  const int t_min = (int)(lo >= 0 ? lo / stride : -((-lo + stride - 1) / stride));
  for (int i = tid; i < bands * tile_t; i += 256) {
    const int k = i / tile_t, tl = i - k * tile_t;
    const int t = t_min + tl;
    ys[i] = (t >= 0 && t < ly) ? y[((long long)b * bands + k) * ly + t] : 0.f;
  }
  __syncthreads();
  const int u = u0 + tid;
  if (u >= lx) return;
  const int rel = u - off0;                 // = t*stride + j
  int j0 = rel % stride;
  if (j0 < 0) j0 += stride;
  float acc = 0.f;
  // t = (rel - j) / stride is exact and falls by one per step of j: ONE division per thread (a division per tap was 101 of them
  // per output of the A-weighting filter's adjoint); same taps in the same order
  int tl = (rel - j0) / stride - t_min;
  for (int j = j0; j < ntaps; j += stride, --tl) {
    if (tl < 0 || tl >= tile_t) continue;
    for (int k = 0; k < bands; ++k) acc = fmaf(ws[k * ntaps + j], ys[k * tile_t + tl], acc);
  }
  x[(long long)b * lx + u] = acc;
}

// ---- PQMF banks at M = 4, N = 32 (pqmf.py:194-213; every EBEN configuration): the polyphase form on wave shuffles ---------------
// Analysis: output t of every band reads the 32 samples 4 t + off0 .. + 31 = EIGHT aligned-in-phase quads.  Lane t of a wave loads ONE
// quad -- the window's last, samples 4 t + off0 + 28 .. + 31: a coalesced 16-byte load, no LDS tile, no barrier -- and gets the seven
// others from lanes t - 1 .. t - 7 (ds_bpermute: __shfl_up); the first seven lanes of a wave only supply their quads (57 outputs per
// wave).  All bands from the one window in registers, the taps through the scalar cache, the 32 products of a band accumulated in the
// order j = 0 .. 31 of the LDS form -- bit-identical results.  ([MI355X] the LDS form: 13 us for 6-8 MB, its 128 ds_read_b32 per
// output four ways bank-conflicted at stride 4.)
template <int BANDS>
__global__ __launch_bounds__(256) void pqmf_analysis_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, int lx, int ly, int off0) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.y;
  const int t = (blockIdx.x * 4 + wave) * 57 - 7 + lane;
  const long long q0 = 4LL * t + off0 + 28;
  const float* xr = x + (long long)b * lx;
  float xq[4];
  if (q0 >= 0 && q0 + 3 < lx) {
#pragma unroll
    for (int e = 0; e < 4; ++e) xq[e] = xr[q0 + e];
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) xq[e] = (q0 + e >= 0 && q0 + e < lx) ? xr[q0 + e] : 0.f;
  }
  float win[8][4];   // win[c] = samples 4 t + off0 + 4 c .. + 3
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) win[c][e] = c == 7 ? xq[e] : __shfl_up(xq[e], 7 - c, 64);
  if (lane < 7 || t < 0 || t >= ly) return;
  typedef const __attribute__((address_space(4))) float* cw_t;
  const cw_t wc = (cw_t)w;
#pragma unroll
  for (int k = 0; k < BANDS; ++k) {
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = fmaf(wc[k * 32 + 4 * c + e], win[c][e], acc);
    y[((long long)b * BANDS + k) * ly + t] = acc;
  }
}

// Synthesis + band sum: outputs u = 4 q + off0 + r, r = 0 .. 3, read y_k[q - i] with the taps j = r + 4 i, i = 0 .. 7: lane q loads its
// BANDS samples y_k[q] (coalesced), gets y_k[q - 1 .. q - 7] from its neighbours and writes the four outputs as one 16-byte store.
// Accumulation order of the LDS form (taps ascending, bands inside): bit-identical.
template <int BANDS>
__global__ __launch_bounds__(256) void pqmf_synthesis_kernel(const float* __restrict__ y, const float* __restrict__ w, float* __restrict__ x, int lx, int ly, int off0, int qmin) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.y;
  const int q = qmin + (blockIdx.x * 4 + wave) * 57 - 7 + lane;
  float yq[BANDS];
#pragma unroll
  for (int k = 0; k < BANDS; ++k) yq[k] = (q >= 0 && q < ly) ? y[((long long)b * BANDS + k) * ly + q] : 0.f;
  float win[8][BANDS];   // win[i][k] = y_k[q - i]
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int k = 0; k < BANDS; ++k) win[i][k] = i == 0 ? yq[k] : __shfl_up(yq[k], i, 64);
  if (lane < 7) return;
  typedef const __attribute__((address_space(4))) float* cw_t;
  const cw_t wc = (cw_t)w;
  const long long u0 = 4LL * q + off0;
  float out[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int k = 0; k < BANDS; ++k) acc = fmaf(wc[k * 32 + r + 4 * i], win[i][k], acc);
    out[r] = acc;
  }
  float* xr = x + (long long)b * lx;
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (u0 + r >= 0 && u0 + r < lx) xr[u0 + r] = out[r];
}

// ---------------------------------------------------------------------------------------------
// elementwise
// ---------------------------------------------------------------------------------------------
#define EBEN_GRID_STRIDE(i, n) \
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (n); i += (size_t)gridDim.x * 256)

__global__ __launch_bounds__(256) void lrelu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, float s) {
  EBEN_GRID_STRIDE(i, n) y[i] = lrelu(x[i], s);
}
__global__ __launch_bounds__(256) void lrelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ ref,
                                                        float* __restrict__ dx, size_t n, float s) {
  EBEN_GRID_STRIDE(i, n) dx[i] = dy[i] * dlrelu(ref[i], s);
}
__global__ __launch_bounds__(256) void axpby_kernel(const float* __restrict__ a, float alpha, const float* __restrict__ b,
                                                    float beta, float* __restrict__ out, size_t n) {
  EBEN_GRID_STRIDE(i, n) out[i] = alpha * a[i] + beta * b[i];
}
__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, size_t n) {
  EBEN_GRID_STRIDE(i, n) out[i] = a[i] + b[i];
}
__global__ __launch_bounds__(256) void tanh_lift_kernel(const float* __restrict__ x, const float* __restrict__ lift,
                                                        float* __restrict__ out, int channels, int c_lift, int length, size_t n) {
  EBEN_GRID_STRIDE(i, n) {
    const size_t row = i / length;
    const int t = (int)(i - row * length);
    const int c = (int)(row % channels);
    const size_t b = row / channels;
    float v = x[i];
    if (c < c_lift) v += lift[(b * c_lift + c) * (size_t)length + t];
    out[i] = tanhf(v);
  }
}
__global__ __launch_bounds__(256) void tanh_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                       float* __restrict__ dx, size_t n) {
  EBEN_GRID_STRIDE(i, n) { const float o = out[i]; dx[i] = dout[i] * (1.f - o * o); }
}
__global__ __launch_bounds__(256) void reflect_pad_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t rows,
                                                              int lx, int pl, int pr) {
  const int ly = lx + pl + pr;
  const size_t n = rows * ly;
  EBEN_GRID_STRIDE(i, n) {
    const size_t row = i / ly;
    int q = (int)(i - row * ly) - pl;
    q = q < 0 ? -q : q;
    q = q >= lx ? 2 * (lx - 1) - q : q;
    y[i] = x[row * lx + q];
  }
}
__global__ __launch_bounds__(256) void reflect_pad_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, size_t rows,
                                                              int lx, int pl, int pr) {
  const int ly = lx + pl + pr;
  const size_t n = rows * lx;
  EBEN_GRID_STRIDE(i, n) {
    const size_t row = i / lx;
    const int u = (int)(i - row * lx);
    const float* d = dy + row * ly;
    float v = d[u + pl];
    if (u >= 1 && u <= pl) v += d[pl - u];
    if (u >= lx - 1 - pr && u <= lx - 2) v += d[pl + 2 * (lx - 1) - u];
    dx[i] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// feature matching (feature_loss.py:37-50)
// ---------------------------------------------------------------------------------------------
constexpr int FM_MAX_PAIRS = 32;
constexpr int FM_BLOCKS = 1024;
struct FmTable {
  const float* a[FM_MAX_PAIRS];
  const float* b[FM_MAX_PAIRS];
  float* da[FM_MAX_PAIRS];
  long long n[FM_MAX_PAIRS];
};

// One block = one contiguous 1/FM_BLOCKS slice of one pair, streamed with four 16-byte loads per tensor in flight per thread (a
// dword-per-thread loop with two loads in flight read the 1.6 GB of embeddings at half the rate the gradient kernel below writes).
__global__ __launch_bounds__(256) void fm_partial_kernel(const FmTable T, float* __restrict__ partial) {
  __shared__ float red[4];
  const int p = blockIdx.y;
  const float* a = T.a[p];
  const float* b = T.b[p];
  const long long n = T.n[p];
  // slice boundaries on multiples of 4 elements; 16-byte loads when both tensors are 16-byte aligned
  const long long per = ((n + FM_BLOCKS - 1) / FM_BLOCKS + 3) & ~3LL;
  const long long lo = (long long)blockIdx.x * per;
  long long hi = lo + per;
  if (hi > n) hi = n;
  float s1 = 0.f, s2 = 0.f;
  if (lo < hi) {
    const bool vec = (((unsigned long long)a | (unsigned long long)b) & 15ull) == 0;
    long long i = lo + 4LL * threadIdx.x;
    if (vec) {
      for (; i + 3 * 1024 + 3 < hi; i += 4 * 1024) {
        f32x4 va[4], vb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          va[u] = *reinterpret_cast<const f32x4*>(a + i + u * 1024);
          vb[u] = *reinterpret_cast<const f32x4*>(b + i + u * 1024);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) { s1 += fabsf(va[u][e] - vb[u][e]); s2 += fabsf(va[u][e]); }
      }
    }
    for (; i < hi; i += 1024) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (i + e < hi) { const float av = a[i + e]; s1 += fabsf(av - b[i + e]); s2 += fabsf(av); }
    }
  }
  s1 = block_sum_256(s1, red);
  s2 = block_sum_256(s2, red);
  if (threadIdx.x == 0) {
    partial[(p * FM_BLOCKS + blockIdx.x) * 2 + 0] = s1;
    partial[(p * FM_BLOCKS + blockIdx.x) * 2 + 1] = s2;
  }
}
__global__ __launch_bounds__(64) void fm_final_kernel(const float* __restrict__ partial, float* __restrict__ sums) {
  const int p = blockIdx.x, l = threadIdx.x;
  float s1 = 0.f, s2 = 0.f;
  for (int k = l; k < FM_BLOCKS; k += 64) {   // fixed order: deterministic
    s1 += partial[(p * FM_BLOCKS + k) * 2 + 0];
    s2 += partial[(p * FM_BLOCKS + k) * 2 + 1];
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if (l == 0) { sums[2 * p] = s1; sums[2 * p + 1] = s2; }
}
__global__ __launch_bounds__(256) void fm_bwd_kernel(const FmTable T, const float* __restrict__ sums, const float* __restrict__ gout,
                                                     float inv_count) {
  const int p = blockIdx.y;
  const float* a = T.a[p];
  const float* b = T.b[p];
  float* da = T.da[p];
  const long long n = T.n[p];
  const float s1 = sums[2 * p], s2 = sums[2 * p + 1];
  const float gs = gout[0] * inv_count;
  const float c1 = gs / s2, c2 = gs * s1 / (s2 * s2);
  auto grad = [&](float av, float bv) {
    const float dv = av - bv;
    const float sg1 = (dv > 0.f) - (dv < 0.f), sg2 = (av > 0.f) - (av < 0.f);
    return c1 * sg1 - c2 * sg2;
  };
  // 16-byte accesses, two per tensor in flight per thread (the dword-per-thread loop wrote at ~1.5 TB/s)
  const bool vec = (((unsigned long long)a | (unsigned long long)b | (unsigned long long)da) & 15ull) == 0;
  const long long n4 = vec ? (n >> 2) : 0, stride = (long long)gridDim.x * 256;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  for (; i + stride < n4; i += 2 * stride) {
    const f32x4 a0 = reinterpret_cast<const f32x4*>(a)[i], a1 = reinterpret_cast<const f32x4*>(a)[i + stride];
    const f32x4 b0 = reinterpret_cast<const f32x4*>(b)[i], b1 = reinterpret_cast<const f32x4*>(b)[i + stride];
    f32x4 g0, g1;
#pragma unroll
    for (int e = 0; e < 4; ++e) { g0[e] = grad(a0[e], b0[e]); g1[e] = grad(a1[e], b1[e]); }
    reinterpret_cast<f32x4*>(da)[i] = g0;
    reinterpret_cast<f32x4*>(da)[i + stride] = g1;
  }
  for (; i < n4; i += stride) {
    const f32x4 a0 = reinterpret_cast<const f32x4*>(a)[i], b0 = reinterpret_cast<const f32x4*>(b)[i];
    f32x4 g0;
#pragma unroll
    for (int e = 0; e < 4; ++e) g0[e] = grad(a0[e], b0[e]);
    reinterpret_cast<f32x4*>(da)[i] = g0;
  }
  for (long long j = 4 * n4 + (long long)blockIdx.x * 256 + threadIdx.x; j < n; j += stride) da[j] = grad(a[j], b[j]);
}

// ---------------------------------------------------------------------------------------------
// hinge (hinge_loss.py:35-43) and L2 norm
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hinge_fwd_kernel(const float* __restrict__ x, size_t n, float target, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (size_t i = threadIdx.x; i < n; i += 256) s += fmaxf(1.f - target * x[i], 0.f);
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) out[0] = s / (float)n;
}
// several hinge terms in one launch (the engine's 12 per step: 3 targets x 4 sub-discriminators): block i = term i, the
// single-term kernel's summation order (same bits)
constexpr int HINGE_MULTI = 32;
struct HingeTable { const float* x[HINGE_MULTI]; long long n[HINGE_MULTI]; float target[HINGE_MULTI]; };
__global__ __launch_bounds__(256) void hinge_fwd_multi_kernel(const HingeTable T, float* __restrict__ out) {
  __shared__ float red[4];
  const int k = blockIdx.x;
  const float* __restrict__ x = T.x[k];
  const size_t n = (size_t)T.n[k];
  const float target = T.target[k];
  float s = 0.f;
  for (size_t i = threadIdx.x; i < n; i += 256) s += fmaxf(1.f - target * x[i], 0.f);
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) out[k] = s / (float)n;
}
__global__ __launch_bounds__(256) void hinge_bwd_kernel(const float* __restrict__ x, size_t n, float target,
                                                        const float* __restrict__ gout, float scale, float* __restrict__ dx) {
  const float g = gout[0] * scale / (float)n;
  EBEN_GRID_STRIDE(i, n) dx[i] = (1.f - target * x[i] > 0.f) ? -target * g : 0.f;
}
// The four stacked seed blocks of the discriminator engine's backward in one launch: [0 | hinge'(a, +1) | hinge'(a, -1) | hinge'(b, +1)]
// (rows [fm | adv | fake | real]; a = logits of the enhanced rows, b = of the reference rows; `per` elements each), every block with
// hinge_bwd_kernel's arithmetic -- a memset and three tiny launches at the head of every input-gradient chain otherwise.
__global__ __launch_bounds__(256) void hinge_bwd_stacked_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t per,
                                                                const float* __restrict__ gout, float s0, float s1, float s2, float* __restrict__ seeds) {
  const float g0 = gout[0] * s0 / (float)per, g1 = gout[0] * s1 / (float)per, g2 = gout[0] * s2 / (float)per;
  EBEN_GRID_STRIDE(i, 4 * per) {
    const size_t seg = i / per, j = i - seg * per;
    float v = 0.f;
    if (seg == 1) v = (1.f - a[j] > 0.f) ? -g0 : 0.f;
    else if (seg == 2) v = (1.f + a[j] > 0.f) ? g1 : 0.f;
    else if (seg == 3) v = (1.f - b[j] > 0.f) ? -g2 : 0.f;
    seeds[i] = v;
  }
}
__global__ __launch_bounds__(256) void l2_partial_kernel(const float* __restrict__ x, size_t n, float* __restrict__ partial) {
  __shared__ float red[4];
  float s = 0.f;
  EBEN_GRID_STRIDE(i, n) { const float v = x[i]; s += v * v; }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ __launch_bounds__(64) void l2_final_kernel(const float* __restrict__ partial, int np, float* __restrict__ out) {
  float s = 0.f;
  for (int i = threadIdx.x; i < np; i += 64) s += partial[i];
  s = wave_sum(s);
  if (threadIdx.x == 0) out[0] = sqrtf(s);
}

// ---------------------------------------------------------------------------------------------
// STFT magnitude losses (auraloss STFTLoss: spectral convergence + log-magnitude L1)
// spec: (rows, 2*bins_pad, frames), re in channel k, im in channel bins_pad+k.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float stft_mag(float re, float im, float eps) { return sqrtf(fmaxf(re * re + im * im, eps)); }

// two-level, order-deterministic: STFT_SPLIT blocks per row write partial sums, one wave per row adds them
constexpr int STFT_SPLIT = 32;
// element (row r, bin k, frame f) of a spectrum sits at r*row_stride + k*bin_stride + f (real part; imaginary at + im_off):
// (rows, 2*bins_pad, frames) tensors and the flat (2*bins, rows*frames) GEMM output are both of that form
struct StftLayout { long long row_stride, bin_stride, im_off; };

__global__ __launch_bounds__(256) void stft_sums_kernel(const float* __restrict__ sx, const float* __restrict__ sy, int bins,
                                                        StftLayout L, int frames, float eps, float* __restrict__ partial) {
  __shared__ float red[4];
  const int r = blockIdx.y;
  const long long base0 = (long long)r * L.row_stride;
  const long long im_off = L.im_off;
  const int n = bins * frames;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += STFT_SPLIT * 256) {
    const int k = i / frames;
    const long long base = base0 + (long long)k * (L.bin_stride - frames);   // + i = k*bin_stride + f
    const float xm = stft_mag(sx[base + i], sx[base + im_off + i], eps);
    const float ym = stft_mag(sy[base + i], sy[base + im_off + i], eps);
    const float d = ym - xm;
    s0 += d * d;
    s1 += ym * ym;
    s2 += fabsf(logf(xm) - logf(ym));
  }
  s0 = block_sum_256(s0, red);
  s1 = block_sum_256(s1, red);
  s2 = block_sum_256(s2, red);
  if (threadIdx.x == 0) {
    float* o = partial + ((long long)r * STFT_SPLIT + blockIdx.x) * 3;
    o[0] = s0; o[1] = s1; o[2] = s2;
  }
}
__global__ __launch_bounds__(64) void stft_sums_final_kernel(const float* __restrict__ partial, float* __restrict__ out) {
  const int r = blockIdx.x, lane = threadIdx.x;
  for (int k = 0; k < 3; ++k) {
    float v = lane < STFT_SPLIT ? partial[((long long)r * STFT_SPLIT + lane) * 3 + k] : 0.f;
    v = wave_sum(v);
    if (lane == 0) out[3 * r + k] = v;
  }
}

// The loss value from the per-row sums of all resolutions in one launch (auraloss: per resolution mean_r sqrt(s0 / s1) [spectral
// convergence] + sum_r s2 / (rows bins frames) [log magnitude], then the mean over the resolutions): ~8 one-element torch kernels
// per resolution otherwise, on the main stream.  One wave; fixed reduction order.
constexpr int STFT_TOTAL_MAX = 8;
struct StftTotalTable { const float* sums[STFT_TOTAL_MAX]; float inv_count[STFT_TOTAL_MAX]; };
__global__ __launch_bounds__(64) void stft_total_kernel(const StftTotalTable T, int n, int rows, float* __restrict__ out) {
  const int lane = threadIdx.x;
  float total = 0.f;
  for (int p = 0; p < n; ++p) {
    float sc = 0.f, lg = 0.f;
    for (int r = lane; r < rows; r += 64) {
      sc += sqrtf(T.sums[p][3 * r] / T.sums[p][3 * r + 1]);
      lg += T.sums[p][3 * r + 2];
    }
    sc = wave_sum(sc);
    lg = wave_sum(lg);
    total += sc / (float)rows + lg * T.inv_count[p];
  }
  if (lane == 0) out[0] = total / (float)n;
}

__global__ __launch_bounds__(256) void stft_bwd_kernel(const float* __restrict__ sx, const float* __restrict__ sy, int rows, int bins,
                                                       StftLayout L, StftLayout LO, int frames, float eps, const float* __restrict__ sums,
                                                       const float* __restrict__ gout, float scale, float* __restrict__ dsx) {
  const int r = blockIdx.y;
  const long long im_off = L.im_off;
  const int n = bins * frames;
  const float g = gout[0] * scale;
  const float c_sc = g / ((float)rows * sqrtf(sums[3 * r]) * sqrtf(sums[3 * r + 1]));
  const float c_lg = g / ((float)rows * (float)bins * (float)frames);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int k = i / frames;
    const long long base = (long long)r * L.row_stride + (long long)k * (L.bin_stride - frames);
    const long long obase = (long long)r * LO.row_stride + (long long)k * (LO.bin_stride - frames);
    const float re = sx[base + i], im = sx[base + im_off + i];
    const float p = re * re + im * im;
    const float xm = sqrtf(fmaxf(p, eps));
    const float ym = stft_mag(sy[base + i], sy[base + im_off + i], eps);
    const float dl = logf(xm) - logf(ym);
    const float sg = (dl > 0.f) - (dl < 0.f);
    float dmag = c_sc * (xm - ym) + c_lg * sg / xm;
    dmag = p >= eps ? dmag / xm : 0.f;   // d sqrt(clamp(p)) / dp * 2  -> (re, im) / mag
    dsx[obase + i] = dmag * re;
    dsx[obase + LO.im_off + i] = dmag * im;
  }
}

// ---------------------------------------------------------------------------------------------
// overlap-add: adjoint of framing a reflect-padded signal (STFT backward).
// frames_buf (B, win, frames) holds per-frame sample gradients; padded coordinate q receives
// sum_f buf[q + pad - f*hop, f]; the reflect padding folds q<0 and q>=lx back into [0, lx).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float ola_gather(const float* __restrict__ buf, int q, int win, int frames, int hop, int pad, long long j_stride) {
  const int s = q + pad;  // = f*hop + j
  if (s < 0) return 0.f;
  int fmax = s / hop;
  if (fmax > frames - 1) fmax = frames - 1;
  int fmin = s - win + 1 <= 0 ? 0 : (s - win + hop) / hop;
  float acc = 0.f;
  for (int f = fmin; f <= fmax; ++f) acc += buf[(long long)(s - f * hop) * j_stride + f];
  return acc;
}
// Folded form (eben_stft_frames_folded's adjoint): buf holds 2*(win/2) rows [dE ; dO]; window sample j = h + m gets
// dE[|m|] + sign(m) dO[|m|], the centre dE[0], sample 0 (zero window weight) nothing.
__device__ __forceinline__ float ola_gather_folded(const float* __restrict__ buf, int q, int win, int frames, int hop, int pad, long long j_stride) {
  const int s = q + pad;
  if (s < 0) return 0.f;
  const int h = win >> 1;
  int fmax = s / hop;
  if (fmax > frames - 1) fmax = frames - 1;
  int fmin = s - win + 1 <= 0 ? 0 : (s - win + hop) / hop;
  float acc = 0.f;
  for (int f = fmin; f <= fmax; ++f) {
    const int j = s - f * hop;
    const int m = j - h, am = m < 0 ? -m : m;
    if (j == 0) continue;
    const float e = buf[(long long)am * j_stride + f];
    const float o = m != 0 ? buf[(long long)(h + am) * j_stride + f] : 0.f;
    acc += m < 0 ? e - o : e + o;
  }
  return acc;
}
__global__ __launch_bounds__(256) void overlap_add_folded_kernel(const float* __restrict__ buf, float* __restrict__ x, int lx, int win,
                                                                 int frames, int hop, int pad, int accumulate,
                                                                 long long row_stride, long long j_stride) {
  const int b = blockIdx.y;
  const float* bb = buf + (long long)b * row_stride;
  for (int u = blockIdx.x * 256 + threadIdx.x; u < lx; u += gridDim.x * 256) {
    float v = ola_gather_folded(bb, u, win, frames, hop, pad, j_stride);
    if (u >= 1 && u <= pad) v += ola_gather_folded(bb, -u, win, frames, hop, pad, j_stride);
    if (u <= lx - 2 && 2 * (lx - 1) - u <= lx - 1 + pad) v += ola_gather_folded(bb, 2 * (lx - 1) - u, win, frames, hop, pad, j_stride);
    const long long i = (long long)b * lx + u;
    if (accumulate) v += x[i];
    x[i] = v;
  }
}

// The same sums from an LDS tile.  The kernel above gathers straight from the buffer with the lanes on consecutive samples, i.e. on
// consecutive ROWS of it (a row = one window sample of every frame): 64 cache lines per load, ten loads per output -- 34-50 us for the
// 20 MB of one resolution.  Here a block owns OLA_U consecutive samples of one item: the frames that reach them (+ the ones the reflected
// ends fold in) are read once, row segment by row segment, unfolded into tile[frame][window sample] = dE[|m|] +- dO[|m|], and every output
// sums its <= win / hop + 1 tile entries (lanes = consecutive samples = consecutive tile entries).  Same terms in the same order: bit-identical.
constexpr int OLA_U = 512;
__device__ __forceinline__ float ola_tile_gather(const float* __restrict__ tile, int q, int win, int frames, int hop, int pad, int f_lo) {
  const int s = q + pad;  // = f*hop + j
  if (s < 0) return 0.f;
  int fmax = s / hop;
  if (fmax > frames - 1) fmax = frames - 1;
  const int fmin = s - win + 1 <= 0 ? 0 : (s - win + hop) / hop;
  float acc = 0.f;
  for (int f = fmin; f <= fmax; ++f) acc += tile[(f - f_lo) * win + (s - f * hop)];
  return acc;
}
__global__ __launch_bounds__(256) void overlap_add_folded_t_kernel(const float* __restrict__ buf, float* __restrict__ x, int lx, int win,
                                                                   int frames, int hop, int pad, int accumulate, long long row_stride,
                                                                   long long j_stride, int fcap) {
  extern __shared__ __attribute__((aligned(16))) float ola_tile[];   // fcap x win
  const int b = blockIdx.y, u0 = blockIdx.x * OLA_U, tid = threadIdx.x;
  const int h = win >> 1;
  const int qlo = u0, qhi = (u0 + OLA_U < lx ? u0 + OLA_U : lx) - 1;
  int f_lo = qlo <= pad ? 0 : (qlo + pad - win + hop) / hop;   // = ceil((qlo + pad - win + 1) / hop) for a positive numerator
  if (f_lo < 0) f_lo = 0;
  int smax = qhi + pad;
  if (qhi >= lx - 1 - pad) {   // samples whose mirror image past the end folds back onto them
    const int ulo = qlo > lx - 1 - pad ? qlo : lx - 1 - pad;
    const int sm = 2 * (lx - 1) - ulo + pad;
    if (sm > smax) smax = sm;
  }
  int f_hi = smax / hop;
  if (f_hi > frames - 1) f_hi = frames - 1;
  int nf = f_hi - f_lo + 1;
  if (nf > fcap) nf = fcap;   // never (fcap is sized for the worst tile)
  const float* bb = buf + (long long)b * row_stride;
  // unfold: entry (frame fl, m) of the two row blocks -> window samples h + m and h - m of that frame
  // thread = (frame fl = tid % nfp, rows m = tid / nfp, + 256 / nfp, ...), nfp = the power of two >= nf: no division, the loads of four
  // rows in flight together
  int nfp = 1;
  while (nfp < nf) nfp <<= 1;
  const int fl = tid & (nfp - 1), mstep = 256 / nfp;
  if (fl < nf) {
    const float* col = bb + f_lo + fl;
    float* row = ola_tile + fl * win;
    for (int m0 = tid / nfp; m0 < h; m0 += 4 * mstep) {
      float e[4], o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int m = m0 + k * mstep;
        const int mc = m < h ? m : 0;
        e[k] = col[(long long)mc * j_stride];
        o[k] = col[(long long)(h + mc) * j_stride];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int m = m0 + k * mstep;
        if (m >= h) continue;
        if (m) { row[h + m] = e[k] + o[k]; row[h - m] = e[k] - o[k]; }
        else { row[h] = e[k] + 0.f; row[0] = 0.f; }   // the centre sample has no odd part; window sample 0 carries no weight
      }
    }
  }
  __syncthreads();
  for (int u = u0 + tid; u <= qhi; u += 256) {
    float v = ola_tile_gather(ola_tile, u, win, frames, hop, pad, f_lo);
    if (u >= 1 && u <= pad) v += ola_tile_gather(ola_tile, -u, win, frames, hop, pad, f_lo);
    if (u <= lx - 2 && 2 * (lx - 1) - u <= lx - 1 + pad) v += ola_tile_gather(ola_tile, 2 * (lx - 1) - u, win, frames, hop, pad, f_lo);
    const long long i = (long long)b * lx + u;
    if (accumulate) v += x[i];
    x[i] = v;
  }
}

// buf element (item b, sample j of the window, frame f) sits at b*row_stride + j*j_stride + f
__global__ __launch_bounds__(256) void overlap_add_kernel(const float* __restrict__ buf, float* __restrict__ x, int lx, int win,
                                                          int frames, int hop, int pad, int reflect, int accumulate,
                                                          long long row_stride, long long j_stride) {
  const int b = blockIdx.y;
  const float* bb = buf + (long long)b * row_stride;
  for (int u = blockIdx.x * 256 + threadIdx.x; u < lx; u += gridDim.x * 256) {
    float v = ola_gather(bb, u, win, frames, hop, pad, j_stride);
    if (reflect) {
      if (u >= 1 && u <= pad) v += ola_gather(bb, -u, win, frames, hop, pad, j_stride);
      if (u <= lx - 2 && 2 * (lx - 1) - u <= lx - 1 + pad) v += ola_gather(bb, 2 * (lx - 1) - u, win, frames, hop, pad, j_stride);
    }
    const long long i = (long long)b * lx + u;
    if (accumulate) v += x[i];
    x[i] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// multi-tensor Adam (torch.optim.Adam, amsgrad=False, maximize=False; optimizer/adam.yaml:1-9)
// ---------------------------------------------------------------------------------------------
constexpr int ADAM_CHUNK = 48;
constexpr int ADAM_BLOCK_ELEMS = 4096;   // per block: 256 threads x 4 iterations x 4 elements
// the launch is a flat list of blocks: bend[k] = blocks of tensors 0 .. k (a grid of (largest tensor, tensors) blocks launched ~98k blocks
// per 48 tensors of which a few thousand had work: 188 us for the step's 703 MB, 3.7 TB/s)
struct AdamTable { EbenAdamTensor t[ADAM_CHUNK]; int bend[ADAM_CHUNK]; };

__device__ __forceinline__ void adam_update(float& p, float g, float& m, float& v, float beta1, float beta2, float eps, float wd, float step_size,
                                            float bc2_sqrt, float grad_scale) {
  g *= grad_scale;
  if (wd != 0.f) g = fmaf(wd, p, g);
  m = m + (g - m) * (1.f - beta1);               // exp_avg.lerp_(grad, 1-beta1)
  v = v * beta2 + (1.f - beta2) * g * g;         // mul_(beta2).addcmul_(g, g, 1-beta2)
  const float denom = sqrtf(v) / bc2_sqrt + eps;
  p = p - step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adam_kernel(const AdamTable T, int ntensors, float lr, float beta1, float beta2, float eps, float wd,
                                                   float bc1, float bc2_sqrt, float grad_scale) {
  int k = 0;
  while (k + 1 < ntensors && (int)blockIdx.x >= T.bend[k]) ++k;   // block-uniform scan of <= 48 entries
  const EbenAdamTensor e = T.t[k];
  const long long base = (long long)((int)blockIdx.x - (k ? T.bend[k - 1] : 0)) * ADAM_BLOCK_ELEMS;
  const float step_size = lr / bc1;
  const bool vec = ((reinterpret_cast<unsigned long long>(e.param) | reinterpret_cast<unsigned long long>(e.grad) |
                     reinterpret_cast<unsigned long long>(e.exp_avg) | reinterpret_cast<unsigned long long>(e.exp_avg_sq)) & 15ull) == 0;
  if (vec && base + ADAM_BLOCK_ELEMS <= e.numel) {
    // whole block inside the tensor, 16-byte aligned: the 12 loads of a thread issued before the first use
    f32x4 g[4], p[4], m[4], v[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const long long i = base + (long long)(it * 256 + threadIdx.x) * 4;
      g[it] = *reinterpret_cast<const f32x4*>(e.grad + i);
      p[it] = *reinterpret_cast<const f32x4*>(e.param + i);
      m[it] = *reinterpret_cast<const f32x4*>(e.exp_avg + i);
      v[it] = *reinterpret_cast<const f32x4*>(e.exp_avg_sq + i);
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const long long i = base + (long long)(it * 256 + threadIdx.x) * 4;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float pp = p[it][c], mm = m[it][c], vv = v[it][c];
        adam_update(pp, g[it][c], mm, vv, beta1, beta2, eps, wd, step_size, bc2_sqrt, grad_scale);
        p[it][c] = pp; m[it][c] = mm; v[it][c] = vv;
      }
      *reinterpret_cast<f32x4*>(e.param + i) = p[it];
      *reinterpret_cast<f32x4*>(e.exp_avg + i) = m[it];
      *reinterpret_cast<f32x4*>(e.exp_avg_sq + i) = v[it];
    }
    return;
  }
  const long long end = base + ADAM_BLOCK_ELEMS < e.numel ? base + ADAM_BLOCK_ELEMS : e.numel;
  for (long long i = base + threadIdx.x; i < end; i += 256) {
    float p = e.param[i], m = e.exp_avg[i], v = e.exp_avg_sq[i];
    adam_update(p, e.grad[i], m, v, beta1, beta2, eps, wd, step_size, bc2_sqrt, grad_scale);
    e.param[i] = p;
    e.exp_avg[i] = m;
    e.exp_avg_sq[i] = v;
  }
}

static int grid_for(size_t n, int cap = 4096) {
  size_t b = (n + 255) / 256;
  if (b > (size_t)cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

static thread_local char g_err[512] = "";
char* tls_error_buffer() { return g_err; }

}  // namespace eben

using namespace eben;

extern "C" const char* eben_last_error(void) { return tls_error_buffer(); }
extern "C" int eben_version(void) { return EBEN_ABI_VERSION; }
extern "C" int eben_device_info(char* name, size_t name_bytes) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return -hip_fail(e, "hipGetDevice");
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) return -hip_fail(e, "hipGetDeviceProperties");
  if (name && name_bytes) { strncpy(name, prop.gcnArchName, name_bytes - 1); name[name_bytes - 1] = 0; }
  return prop.multiProcessorCount;
}

namespace eben {
// Single-band stride-1 FIR and its adjoint: out[i] = sum_j w[j] in[i + off + SGN j], j ascending (zero outside the input) -- the A-weighting
// prefilter of the MRSTFT loss (101 taps; auraloss FIRFilter "aw", multi_stft.yaml:15-18) forward (SGN = +1) and backward (SGN = -1).
// The generic kernels above spend two LDS reads per multiply-add on it (27 / 54 us for 64 x 32000 samples).  Here a thread owns four
// consecutive outputs and walks the taps four at a time over a window of eight inputs held in registers: one 16-byte LDS read (lanes
// 16 bytes apart: conflict-free) per sixteen multiply-adds, the taps from the scalar cache.  Same products in the same order as the
// generic kernels: bit-identical results.
constexpr int FIR1_BLOCK = 1024;   // outputs per block
template <int SGN>
__global__ __launch_bounds__(256) void fir1_kernel(const float* __restrict__ in, const float* __restrict__ w, float* __restrict__ out,
                                                   int lin, int lout, int ntaps, int off) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.y, i0 = blockIdx.x * FIR1_BLOCK, tid = threadIdx.x;
  // xs[r + sh] = in[base + r]; sh aligns the windows of the SGN = -1 walk to 16 bytes
  const int base = SGN > 0 ? i0 + off : i0 + off - (ntaps - 1);
  const int sh = SGN > 0 ? 0 : (4 - ((ntaps - 1 - 3) & 3)) & 3;
  const int span = FIR1_BLOCK + ntaps - 1;
  const int total = ((span + sh + 3) & ~3) + 8;
  const float* xr = in + (long long)b * lin;
  for (int r = tid; r < total; r += 256) {
    const int q = base + r - sh;
    smem[r] = (r >= sh && r < span + sh && q >= 0 && q < lin) ? xr[q] : 0.f;
  }
  __syncthreads();
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  float v[8];
  if (SGN > 0) {
    const float* xp = smem + 4 * tid;
    *reinterpret_cast<f32x4*>(v) = *reinterpret_cast<const f32x4*>(xp);
    for (int j0 = 0; j0 < ntaps; j0 += 4) {
      *reinterpret_cast<f32x4*>(v + 4) = *reinterpret_cast<const f32x4*>(xp + j0 + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float wk = j0 + k < ntaps ? w[j0 + k] : 0.f;   // uniform: scalar loads
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = fmaf(wk, v[e + k], acc[e]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e + 4];
    }
  } else {
    // input index of (output e, tap j) in xs: 4 tid + e + (ntaps - 1) - j + sh = w0 + (e - k + 3) with w0 = 4 tid + ntaps - 4 + sh - j0
    const float* xp = smem + 4 * tid + (ntaps - 4 + sh);
    *reinterpret_cast<f32x4*>(v + 4) = *reinterpret_cast<const f32x4*>(xp + 4);
    for (int j0 = 0; j0 < ntaps; j0 += 4) {
      *reinterpret_cast<f32x4*>(v) = *reinterpret_cast<const f32x4*>(xp - j0);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float wk = j0 + k < ntaps ? w[j0 + k] : 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = fmaf(wk, v[e - k + 3], acc[e]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e + 4] = v[e];
    }
  }
  const int i = i0 + 4 * tid;
  float* o = out + (long long)b * lout + i;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (i + e < lout) o[e] = acc[e];
}
static const bool fir1_enabled = !(getenv("EBEN_FIR1") && atoi(getenv("EBEN_FIR1")) == 0);
template <int SGN>
static int fir1_launch(const float* in, const float* w, float* out, int batch, int lin, int lout, int ntaps, int off, hipStream_t st) {
  const size_t lds = sizeof(float) * (size_t)(FIR1_BLOCK + ntaps + 16);
  hipLaunchKernelGGL(fir1_kernel<SGN>, dim3(ceil_div(lout, FIR1_BLOCK), batch), dim3(256), lds, st, in, w, out, lin, lout, ntaps, off);
  EBEN_CHECK_LAUNCH("fir1_kernel");
  return EBEN_OK;
}
}  // namespace eben

extern "C" int eben_fir_decimate(const float* x, const float* w, float* y, int batch, int lx, int ly, int bands, int ntaps,
                                 int stride, int off0, void* stream) {
  EBEN_REQUIRE(x && w && y && batch > 0 && lx > 0 && ly > 0 && bands > 0 && ntaps > 0 && stride > 0, "bad fir_decimate arguments");
  EBEN_REQUIRE(bands * ntaps <= FIR_MAX_W, "fir bank of %d x %d taps exceeds %d", bands, ntaps, FIR_MAX_W);
  static const bool shuffles = !(getenv("EBEN_PQMF_SHUFFLE") && atoi(getenv("EBEN_PQMF_SHUFFLE")) == 0);
  if (shuffles && stride == 4 && ntaps == 32 && (bands == 1 || bands == 2 || bands == 4)) {   // the PQMF banks: polyphase form on wave shuffles
    const dim3 grid(ceil_div(ly, 4 * 57), batch);
    if (bands == 1) hipLaunchKernelGGL(pqmf_analysis_kernel<1>, grid, dim3(256), 0, as_stream(stream), x, w, y, lx, ly, off0);
    else if (bands == 2) hipLaunchKernelGGL(pqmf_analysis_kernel<2>, grid, dim3(256), 0, as_stream(stream), x, w, y, lx, ly, off0);
    else hipLaunchKernelGGL(pqmf_analysis_kernel<4>, grid, dim3(256), 0, as_stream(stream), x, w, y, lx, ly, off0);
    EBEN_CHECK_LAUNCH("pqmf_analysis_kernel");
    return EBEN_OK;
  }
  if (fir1_enabled && stride == 1 && bands == 1 && ntaps >= 4) return fir1_launch<1>(x, w, y, batch, lx, ly, ntaps, off0, as_stream(stream));
  const size_t lds = sizeof(float) * (FIR_MAX_W + (size_t)255 * stride + ntaps);
  EBEN_REQUIRE(lds <= 64 * 1024, "fir_decimate stride %d too large", stride);
  hipLaunchKernelGGL(fir_decimate_kernel, dim3(ceil_div(ly, 256), batch), dim3(256), lds, as_stream(stream), x, w, y, lx, ly,
                     bands, ntaps, stride, off0);
  EBEN_CHECK_LAUNCH("fir_decimate_kernel");
  return EBEN_OK;
}

extern "C" int eben_fir_interp_sum(const float* y, const float* w, float* x, int batch, int lx, int ly, int bands, int ntaps,
                                   int stride, int off0, void* stream) {
  EBEN_REQUIRE(x && w && y && batch > 0 && lx > 0 && ly > 0 && bands > 0 && ntaps > 0 && stride > 0, "bad fir_interp_sum arguments");
  EBEN_REQUIRE(bands * ntaps <= FIR_MAX_W, "fir bank of %d x %d taps exceeds %d", bands, ntaps, FIR_MAX_W);
  static const bool shuffles = !(getenv("EBEN_PQMF_SHUFFLE") && atoi(getenv("EBEN_PQMF_SHUFFLE")) == 0);
  if (shuffles && stride == 4 && ntaps == 32 && (bands == 1 || bands == 2 || bands == 4)) {
    // u - off0 = 4 q + r covers u = 0 .. lx - 1 for q = floor(-off0 / 4) .. floor((lx - 1 - off0) / 4)
    auto fdiv = [](long long a, long long d) { long long qq = a / d; return (a % d != 0 && ((a < 0) != (d < 0))) ? qq - 1 : qq; };
    const int qmin = (int)fdiv(-(long long)off0, 4), qmax = (int)fdiv((long long)lx - 1 - off0, 4);
    const dim3 grid(ceil_div(qmax - qmin + 1, 4 * 57), batch);
    if (bands == 1) hipLaunchKernelGGL(pqmf_synthesis_kernel<1>, grid, dim3(256), 0, as_stream(stream), y, w, x, lx, ly, off0, qmin);
    else if (bands == 2) hipLaunchKernelGGL(pqmf_synthesis_kernel<2>, grid, dim3(256), 0, as_stream(stream), y, w, x, lx, ly, off0, qmin);
    else hipLaunchKernelGGL(pqmf_synthesis_kernel<4>, grid, dim3(256), 0, as_stream(stream), y, w, x, lx, ly, off0, qmin);
    EBEN_CHECK_LAUNCH("pqmf_synthesis_kernel");
    return EBEN_OK;
  }
  if (fir1_enabled && stride == 1 && bands == 1 && ntaps >= 4) return fir1_launch<-1>(y, w, x, batch, ly, lx, ntaps, -off0, as_stream(stream));
  const int tile_t = (255 + ntaps - 1) / stride + 3;
  const size_t lds = sizeof(float) * (FIR_MAX_W + (size_t)bands * tile_t);
  EBEN_REQUIRE(lds <= 64 * 1024, "fir_interp_sum tile too large");
  hipLaunchKernelGGL(fir_interp_sum_kernel, dim3(ceil_div(lx, 256), batch), dim3(256), lds, as_stream(stream), y, w, x, lx, ly,
                     bands, ntaps, stride, off0, tile_t);
  EBEN_CHECK_LAUNCH("fir_interp_sum_kernel");
  return EBEN_OK;
}

extern "C" int eben_lrelu_fwd(const float* x, float* y, size_t n, float slope, void* stream) {
  EBEN_REQUIRE(x && y, "null pointer");
  if (n == 0) return EBEN_OK;
  hipLaunchKernelGGL(lrelu_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), x, y, n, slope);
  EBEN_CHECK_LAUNCH("lrelu_fwd_kernel");
  return EBEN_OK;
}
extern "C" int eben_lrelu_bwd(const float* dy, const float* ref, float* dx, size_t n, float slope, void* stream) {
  EBEN_REQUIRE(dy && ref && dx, "null pointer");
  if (n == 0) return EBEN_OK;
  hipLaunchKernelGGL(lrelu_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), dy, ref, dx, n, slope);
  EBEN_CHECK_LAUNCH("lrelu_bwd_kernel");
  return EBEN_OK;
}
// ---------------------------------------------------------------------------------------------
// Space to depth along time: out[row][r][q] = xp[row][S q + r + off], r in [0, S), q in [0, Lq), where xp is x continued by
// reflection (reflect) or zeros beyond [0, L), optionally times lrelu'(mask[row][.], slope) (the gradient of a fused output
// activation).  A stride-S conv with k = kq S taps over C channels is then a stride-1 conv with kq taps over the C S channels
// (c, r) of `out` -- the form the split-operand tap-conv covers (a stride-8 layer's input tile of 128 outputs is 1032 positions
// wide, beyond its per-thread prefetch; as 8 channels x 129 positions it is an ordinary tile).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void space_to_depth_kernel(const float* __restrict__ x, const float* __restrict__ mask, float* __restrict__ out,
                                                             int rows, int L, int S, int off, int Lq, int reflect, float slope) {
  const int n = S * Lq;
  for (long long row = blockIdx.y; row < rows; row += gridDim.y) {
  const float* xr = x + row * L;
  const float* mr = mask ? mask + row * L : nullptr;
  float* o = out + row * (long long)S * Lq;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int r = i / Lq, q = i - r * Lq;
    int p = S * q + r + off;
    bool ok = true;
    if (reflect) {
      if (p < 0) p = -p;
      if (p >= L) p = 2 * (L - 1) - p;
      ok = p >= 0 && p < L;
    } else {
      ok = p >= 0 && p < L;
    }
    float v = ok ? xr[p] : 0.f;
    if (mr && ok) v *= dlrelu(mr[p], slope);
    o[i] = v;
  }
  }
}
// The same, one row per block through LDS: the S Lq positions a row's output covers are ONE contiguous run of the input (p = i + off,
// i = S q + r), read coalesced, and leave in (r, q) order -- LDS index i + i / S makes the stride-S reads of a wave conflict-free.
// (The form above reads with stride S and runs 4 blocks of <= 256 elements per 1000-element row: [MI355X] 29 us for 2 x 16 MB.)
__global__ __launch_bounds__(256) void space_to_depth_row_kernel(const float* __restrict__ x, const float* __restrict__ mask, float* __restrict__ out,
                                                                 int rows, int L, int S, int off, int Lq, int reflect, float slope) {
  extern __shared__ float s2d_tile[];
  const int n = S * Lq;
  for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
    const float* xr = x + row * L;
    const float* mr = mask ? mask + row * L : nullptr;
    float* o = out + row * (long long)n;
    __syncthreads();   // the previous row's tile has been written out
    // four elements per thread and pass, all loads before the first use: a 1000-element row is ONE round trip instead of four
    for (int base = 0; base < n; base += 1024) {
      float xv[4], mv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = base + u * 256 + (int)threadIdx.x;
        int p = i + off;
        if (reflect) {
          if (p < 0) p = -p;
          if (p >= L) p = 2 * (L - 1) - p;
        }
        const bool ok = i < n && p >= 0 && p < L;
        xv[u] = ok ? xr[p] : 0.f;
        mv[u] = (mr && ok) ? mr[p] : 1.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = base + u * 256 + (int)threadIdx.x;
        if (i < n) s2d_tile[i + i / S] = mr ? xv[u] * dlrelu(mv[u], slope) : xv[u];
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) {
      const int r = i / Lq, q = i - r * Lq;
      const int j = S * q + r;
      o[i] = s2d_tile[j + q];   // j / S = q
    }
  }
}
extern "C" int eben_space_to_depth(const float* x, const float* mask, float mask_slope, float* out, int rows, int L, int S, int off, int Lq,
                                   int reflect, void* stream) {
  EBEN_REQUIRE(x && out && rows > 0 && L > 0 && S > 0 && Lq > 0, "bad space_to_depth arguments");
  EBEN_REQUIRE(!reflect || (-off < L && S * (Lq - 1) + S - 1 + off < 2 * L - 1), "space_to_depth: reflection wider than the signal");
  static const int by_rows = getenv("EBEN_S2D_ROWS") ? atoi(getenv("EBEN_S2D_ROWS")) : 1;
  const size_t tile = sizeof(float) * ((size_t)S * Lq + Lq + 1);
  if (by_rows && tile <= 48 * 1024) {
    hipLaunchKernelGGL(space_to_depth_row_kernel, dim3(rows < 8192 ? rows : 8192), dim3(256), tile, as_stream(stream), x, mask, out, rows, L, S, off, Lq, reflect,
                       mask_slope);
    EBEN_CHECK_LAUNCH("space_to_depth_row_kernel");
    return EBEN_OK;
  }
  hipLaunchKernelGGL(space_to_depth_kernel, dim3(grid_for((size_t)S * Lq, 64), rows < 65535 ? rows : 65535), dim3(256), 0, as_stream(stream), x, mask,
                     out, rows, L, S, off, Lq, reflect, mask_slope);
  EBEN_CHECK_LAUNCH("space_to_depth_kernel");
  return EBEN_OK;
}
extern "C" int eben_add(const float* a, const float* b, float* out, size_t n, void* stream) {
  EBEN_REQUIRE(a && b && out, "null pointer");
  if (n == 0) return EBEN_OK;
  hipLaunchKernelGGL(add_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), a, b, out, n);
  EBEN_CHECK_LAUNCH("add_kernel");
  return EBEN_OK;
}
extern "C" int eben_axpby(const float* a, float alpha, const float* b, float beta, float* out, size_t n, void* stream) {
  EBEN_REQUIRE(a && b && out, "null pointer");
  if (n == 0) return EBEN_OK;
  hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), a, alpha, b, beta, out, n);
  EBEN_CHECK_LAUNCH("axpby_kernel");
  return EBEN_OK;
}
extern "C" int eben_tanh_lift_fwd(const float* x, const float* lift, float* out, int batch, int channels, int c_lift, int length, void* stream) {
  EBEN_REQUIRE(x && out && batch > 0 && channels > 0 && length > 0 && c_lift >= 0 && c_lift <= channels, "bad tanh_lift arguments");
  EBEN_REQUIRE(c_lift == 0 || lift, "null lift");
  const size_t n = (size_t)batch * channels * length;
  hipLaunchKernelGGL(tanh_lift_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), x, lift, out, channels, c_lift, length, n);
  EBEN_CHECK_LAUNCH("tanh_lift_kernel");
  return EBEN_OK;
}
extern "C" int eben_tanh_bwd(const float* dout, const float* out, float* dx, size_t n, void* stream) {
  EBEN_REQUIRE(dout && out && dx, "null pointer");
  if (n == 0) return EBEN_OK;
  hipLaunchKernelGGL(tanh_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), dout, out, dx, n);
  EBEN_CHECK_LAUNCH("tanh_bwd_kernel");
  return EBEN_OK;
}
extern "C" int eben_reflect_pad_fwd(const float* x, float* y, int rows, int lx, int pad_l, int pad_r, void* stream) {
  EBEN_REQUIRE(x && y && rows > 0 && lx > 0 && pad_l >= 0 && pad_r >= 0 && pad_l < lx && pad_r < lx, "bad reflect_pad arguments");
  const size_t n = (size_t)rows * (lx + pad_l + pad_r);
  hipLaunchKernelGGL(reflect_pad_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), x, y, (size_t)rows, lx, pad_l, pad_r);
  EBEN_CHECK_LAUNCH("reflect_pad_fwd_kernel");
  return EBEN_OK;
}
extern "C" int eben_reflect_pad_bwd(const float* dy, float* dx, int rows, int lx, int pad_l, int pad_r, void* stream) {
  EBEN_REQUIRE(dy && dx && rows > 0 && lx > 0 && pad_l >= 0 && pad_r >= 0 && pad_l < lx && pad_r < lx, "bad reflect_pad arguments");
  const size_t n = (size_t)rows * lx;
  hipLaunchKernelGGL(reflect_pad_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), dy, dx, (size_t)rows, lx, pad_l, pad_r);
  EBEN_CHECK_LAUNCH("reflect_pad_bwd_kernel");
  return EBEN_OK;
}

extern "C" size_t eben_fm_sums_workspace(int npairs) { return sizeof(float) * 2 * FM_BLOCKS * (size_t)(npairs > 0 ? npairs : 0); }

static int fill_fm_table(FmTable* T, const void* const* ptrs, void* const* da, const int64_t* numel, int p0, int cnt) {
  for (int i = 0; i < cnt; ++i) {
    T->a[i] = static_cast<const float*>(ptrs[2 * (p0 + i)]);
    T->b[i] = static_cast<const float*>(ptrs[2 * (p0 + i) + 1]);
    T->da[i] = da ? static_cast<float*>(da[p0 + i]) : nullptr;
    T->n[i] = numel[p0 + i];
    if (!T->a[i] || !T->b[i] || T->n[i] <= 0) return fail(EBEN_EINVAL, "feature-matching pair %d is null or empty", p0 + i);
  }
  return EBEN_OK;
}

extern "C" int eben_fm_sums(const void* const* ptrs, const int64_t* numel, int npairs, float* partial_ws, size_t ws_bytes,
                            float* sums, void* stream) {
  EBEN_REQUIRE(ptrs && numel && npairs > 0 && partial_ws && sums, "bad fm_sums arguments");
  if (ws_bytes < eben_fm_sums_workspace(npairs)) return fail(EBEN_EWORKSPACE, "fm_sums workspace too small");
  for (int p0 = 0; p0 < npairs; p0 += FM_MAX_PAIRS) {
    const int cnt = npairs - p0 < FM_MAX_PAIRS ? npairs - p0 : FM_MAX_PAIRS;
    FmTable T;
    int rc = fill_fm_table(&T, ptrs, nullptr, numel, p0, cnt);
    if (rc) return rc;
    float* part = partial_ws + (size_t)p0 * FM_BLOCKS * 2;
    hipLaunchKernelGGL(fm_partial_kernel, dim3(FM_BLOCKS, cnt), dim3(256), 0, as_stream(stream), T, part);
    EBEN_CHECK_LAUNCH("fm_partial_kernel");
    hipLaunchKernelGGL(fm_final_kernel, dim3(cnt), dim3(64), 0, as_stream(stream), part, sums + 2 * p0);
    EBEN_CHECK_LAUNCH("fm_final_kernel");
  }
  return EBEN_OK;
}

extern "C" int eben_fm_bwd(const void* const* ptrs, void* const* da_ptrs, const int64_t* numel, int npairs, const float* sums,
                           const float* gout, float inv_count, void* stream) {
  EBEN_REQUIRE(ptrs && da_ptrs && numel && npairs > 0 && sums && gout, "bad fm_bwd arguments");
  for (int p0 = 0; p0 < npairs; p0 += FM_MAX_PAIRS) {
    const int cnt = npairs - p0 < FM_MAX_PAIRS ? npairs - p0 : FM_MAX_PAIRS;
    FmTable T;
    int rc = fill_fm_table(&T, ptrs, da_ptrs, numel, p0, cnt);
    if (rc) return rc;
    long long mx = 0;
    for (int i = 0; i < cnt; ++i) { if (T.n[i] > mx) mx = T.n[i]; if (!T.da[i]) return fail(EBEN_EINVAL, "null gradient buffer"); }
    hipLaunchKernelGGL(fm_bwd_kernel, dim3(grid_for((size_t)(mx + 7) / 8, 1024), cnt), dim3(256), 0, as_stream(stream), T, sums + 2 * p0, gout, inv_count);
    EBEN_CHECK_LAUNCH("fm_bwd_kernel");
  }
  return EBEN_OK;
}

extern "C" int eben_hinge_fwd(const float* x, size_t n, float target, float* out, void* stream) {
  EBEN_REQUIRE(x && out && n > 0, "bad hinge arguments");
  hipLaunchKernelGGL(hinge_fwd_kernel, dim3(1), dim3(256), 0, as_stream(stream), x, n, target, out);
  EBEN_CHECK_LAUNCH("hinge_fwd_kernel");
  return EBEN_OK;
}
// ---------------------------------------------------------------------------------------------
// Dynamic loss balancing (eben.py:222-240 as restated by EBENLightningModule._update_lambdas): the scalar arithmetic of a step --
// EMA of the gradient norms (initialised with the first norms, the update applied on that same call), lambda = clamp(1 / (ema +
// 1e-4), 0, 1e4), backprop loss = sum loss_i lambda_i -- and the lambda-weighted sum of the seeds, as two launches instead of ~35
// one-element torch kernels on the main stream in front of the generator backward.  Operation by operation torch's order and
// roundings (no contraction).
// ---------------------------------------------------------------------------------------------
// one rounding per product / sum, whatever -ffp-contract says (the pragma does not reach through the grid-stride macro's loop body)
__device__ __forceinline__ float bal_mul(float a, float b) { float r; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float bal_add(float a, float b) { float r; asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
constexpr int BAL_MAX = 8;
struct BalTable { const float* norm[BAL_MAX]; const float* loss[BAL_MAX]; };
__global__ void balance_kernel(const BalTable T, float* __restrict__ old, int n, int init, int ema, float beta, float one_minus_beta,
                               float* __restrict__ lambdas, float* __restrict__ backprop) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float bp = 0.f;
  for (int i = 0; i < n; ++i) {
    const float nv = *T.norm[i];
    float o = init ? nv : old[i];
    if (ema) o = bal_add(bal_mul(beta, o), bal_mul(one_minus_beta, nv));
    old[i] = o;
    const float r = __fdiv_rn(1.f, bal_add(o, 1e-4f));
    const float lam = r != r ? r : fminf(fmaxf(r, 0.f), 1e4f);   // torch.clamp keeps a NaN
    lambdas[i] = lam;
    const float term = bal_mul(*T.loss[i], lam);
    bp = i == 0 ? term : bal_add(bp, term);
  }
  backprop[0] = bp;
}
struct WsumTable { const float* t[BAL_MAX]; };
__global__ __launch_bounds__(256) void weighted_sum_kernel(const WsumTable T, const float* __restrict__ w, int n, size_t numel, float* __restrict__ out) {
  float wv[BAL_MAX];
#pragma unroll
  for (int i = 0; i < BAL_MAX; ++i) wv[i] = i < n ? w[i] : 0.f;
  EBEN_GRID_STRIDE(j, numel) {
    float acc = bal_mul(T.t[0][j], wv[0]);
    for (int i = 1; i < n; ++i) acc = bal_add(acc, bal_mul(T.t[i][j], wv[i]));
    out[j] = acc;
  }
}
// ---- the balancing norms in one pass (vibravox/lightning_modules/eben.py:222-229) ----------------------------------------------------------
// dynamically_balance_losses differentiates every atomic loss down to `generator.last_conv.weight` and takes the norm of that gradient.
// With the seeds s_i = dL_i / d(bands) at hand (the engine step), that is, for the plain k-tap reflect-padded last conv and the tanh
// recomposition bands = tanh(last_conv(pre) + lift) (eben_generator.py:159-166, 203-208):
//     dW_i[co][ci][j] = sum_{b,t} s_i[b,co,t] (1 - bands[b,co,t]^2) pre[b, ci, reflect(t + j - pad)],      norm_i = ||dW_i||_2 .
// Through autograd this is, per loss, a tanh-backward launch, a weight-gradient launch (+ its slab sum) and torch.norm's kernels --
// ~15 tiny dependent launches between the discriminators' input gradients and the generator backward; here the n losses share ONE pass
// over `pre`: partial sums per (item, position range) in fixed order, then one block that finishes the sums and the norms.
constexpr int LCN_T = 256;        // positions per staged chunk
constexpr int LCN_MAXN = 4;       // losses per call
constexpr int LCN_SPLIT = 64;     // position ranges per item, at most (one staged chunk per block: the staging loads are what a block waits for)
struct LcnArgs {
  const float* seeds[LCN_MAXN]; const float* bands; const float* pre; float* partial;
  int n, B, Cin, Cout, L, k, pad, per;   // per: positions per block (a multiple of LCN_T)
};
template <int CIN, int COUT, int K>
__global__ __launch_bounds__(256) void lcn_partial_kernel(const LcnArgs P) {
  constexpr int COLS = CIN * K, XS = LCN_T + K - 1;
  static_assert(COLS <= 128, "one column per thread of a half block");
  extern __shared__ float lcn_smem[];
  float* xs = lcn_smem;                         // [CIN][XS]: pre at positions t0 - pad .. (reflected)
  float* gs = lcn_smem + CIN * XS;              // [n COUT][LCN_T]: s_i (1 - bands^2)
  const int b = blockIdx.y, sp = blockIdx.x;
  const int tid = threadIdx.x, col = tid & 127, th = tid >> 7;
  const int ci = col / K, j = col - ci * K;
  const int R = P.n * COUT;
  float acc[LCN_MAXN * COUT];
#pragma unroll
  for (int r = 0; r < LCN_MAXN * COUT; ++r) acc[r] = 0.f;
  const int lo = sp * P.per, hi = lo + P.per < P.L ? lo + P.per : P.L;
  for (int t0 = lo; t0 < hi; t0 += LCN_T) {
    __syncthreads();
    // loads in batches of eight with clamped addresses (a load inside its bounds check is issued and waited for one at a time)
    for (int i0 = tid; i0 < CIN * XS; i0 += 8 * 256) {
      float v[8];
      bool ok[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 256 * u;
        const int ic = i < CIN * XS ? i : 0;
        const int c = ic / XS, r = ic - c * XS;
        int q = t0 - P.pad + r;
        q = q < 0 ? -q : q;
        q = q >= P.L ? 2 * (P.L - 1) - q : q;
        ok[u] = i < CIN * XS && q >= 0 && q < P.L;
        v[u] = P.pre[((long long)b * CIN + c) * P.L + (ok[u] ? q : 0)];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + 256 * u < CIN * XS) xs[i0 + 256 * u] = ok[u] ? v[u] : 0.f;
    }
    for (int i0 = tid; i0 < R * LCN_T; i0 += 4 * 256) {
      float sv[4], yv[4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + 256 * u;
        const int ic = i < R * LCN_T ? i : 0;
        const int r = ic / LCN_T, t = t0 + (ic - r * LCN_T);
        const int si = r / COUT, co = r - si * COUT;
        ok[u] = i < R * LCN_T && t < hi;
        const long long o = ((long long)b * COUT + co) * P.L + (ok[u] ? t : lo);
        yv[u] = P.bands[o];
        sv[u] = P.seeds[si][o];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (i0 + 256 * u < R * LCN_T) gs[i0 + 256 * u] = ok[u] ? sv[u] * (1.f - yv[u] * yv[u]) : 0.f;
    }
    __syncthreads();
    if (col < COLS) {
      const float* xr = xs + ci * XS + j;
      // four positions per step: the gradient rows are read as float4 (wave-uniform addresses: LDS broadcasts), 16 LDS reads per 48 FMAs
#pragma unroll 2
      for (int t = th * (LCN_T / 2); t < (th + 1) * (LCN_T / 2); t += 4) {
        const float x0 = xr[t], x1 = xr[t + 1], x2 = xr[t + 2], x3 = xr[t + 3];
#pragma unroll
        for (int r = 0; r < LCN_MAXN * COUT; ++r)
          if (r < R) {
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(gs + r * LCN_T + t);
            acc[r] = fmaf(g4[3], x3, fmaf(g4[2], x2, fmaf(g4[1], x1, fmaf(g4[0], x0, acc[r]))));
          }
      }
    }
  }
  // the two time halves meet in LDS in a fixed order; slab [block][row][column]
  __syncthreads();
  float* red = lcn_smem;                        // [R][COLS] of the second half
  if (th == 1 && col < COLS)
    for (int r = 0; r < R; ++r) red[r * COLS + col] = acc[r];
  __syncthreads();
  if (th == 0 && col < COLS) {
    float* out = P.partial + ((long long)b * gridDim.x + sp) * (LCN_MAXN * COUT * COLS);
    for (int r = 0; r < R; ++r) out[r * COLS + col] = acc[r] + red[r * COLS + col];
  }
}
// sums the slabs in a fixed order: block = 64 gradient entries x 4 slab residues (a thread's 64-odd loads are independent and unrolled:
// one block of 1024 threads walking all the slabs alone took ~250 us of dependent round trips), then the squares of the block's entries
__global__ __launch_bounds__(256) void lcn_sum_kernel(const float* __restrict__ partial, int nslab, int total, int slab_stride, float* __restrict__ dw) {
  __shared__ float red[4][64];
  const int tid = threadIdx.x, l = tid & 63, zp = tid >> 6;
  const int idx = blockIdx.x * 64 + l;
  float s = 0.f;
  if (idx < total) {
    int z = zp;
    for (; z + 28 < nslab; z += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partial[(long long)(z + 4 * u) * slab_stride + idx];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; z < nslab; z += 4) s += partial[(long long)z * slab_stride + idx];
  }
  red[zp][l] = s;
  __syncthreads();
  if (zp == 0 && idx < total) dw[idx] = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
}
// norm_i = sqrt(sum of squares of the per_loss entries of dW_i), one wave per loss
__global__ __launch_bounds__(256) void lcn_norm_kernel(const float* __restrict__ dw, int n, int per_loss, float* __restrict__ norms) {
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (w >= n) return;
  float s = 0.f;
  for (int i = l; i < per_loss; i += 64) { const float v = dw[w * per_loss + i]; s = fmaf(v, v, s); }
  s = wave_sum(s);
  if (l == 0) norms[w] = sqrtf(s);
}

extern "C" size_t eben_last_conv_norms_workspace(int batch) {
  return batch > 0 ? sizeof(float) * ((size_t)batch * LCN_SPLIT + 1) * LCN_MAXN * 4 * 96 : 0;   // the slabs + the summed gradients
}

extern "C" int eben_last_conv_norms(const void* const* seeds, int n, const float* bands, const float* pre, int batch, int c_in, int c_out, int length,
                                    int ksize, int pad, float* workspace, size_t ws_bytes, float* norms, void* stream) {
  EBEN_REQUIRE(seeds && bands && pre && workspace && norms && n >= 1 && n <= LCN_MAXN && batch > 0 && length > 0, "bad arguments to eben_last_conv_norms");
  if (c_in != 32 || c_out != 4 || ksize != 3 || pad != 1 || length < 2)
    return fail(EBEN_EUNSUPPORTED, "eben_last_conv_norms: built for EBEN's last conv (32 -> 4, k 3, reflect padding 1), got %d -> %d k %d pad %d", c_in, c_out, ksize, pad);
  EBEN_REQUIRE(ws_bytes >= eben_last_conv_norms_workspace(batch), "eben_last_conv_norms: workspace too small");
  LcnArgs a;
  for (int i = 0; i < LCN_MAXN; ++i) a.seeds[i] = i < n ? static_cast<const float*>(seeds[i]) : nullptr;
  for (int i = 0; i < n; ++i) EBEN_REQUIRE(a.seeds[i] != nullptr, "null seed %d", i);
  a.bands = bands; a.pre = pre; a.partial = workspace;
  a.n = n; a.B = batch; a.Cin = c_in; a.Cout = c_out; a.L = length; a.k = ksize; a.pad = pad;
  a.per = round_up(ceil_div(length, LCN_SPLIT), LCN_T);   // >= one chunk; 8000 positions: 32 blocks of one chunk per item
  const int nsp = ceil_div(length, a.per);
  const size_t lds = sizeof(float) * ((size_t)32 * (LCN_T + 2) + (size_t)LCN_MAXN * 4 * LCN_T);
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL((lcn_partial_kernel<32, 4, 3>), dim3(nsp, batch), dim3(256), lds, st, a);
  EBEN_CHECK_LAUNCH("lcn_partial_kernel");
  float* dw = workspace + (size_t)batch * LCN_SPLIT * LCN_MAXN * 4 * 96;
  hipLaunchKernelGGL(lcn_sum_kernel, dim3(ceil_div(n * 4 * 96, 64)), dim3(256), 0, st, workspace, nsp * batch, n * 4 * 96, LCN_MAXN * 4 * 96, dw);
  EBEN_CHECK_LAUNCH("lcn_sum_kernel");
  hipLaunchKernelGGL(lcn_norm_kernel, dim3(1), dim3(256), 0, st, dw, n, 4 * 96, norms);
  EBEN_CHECK_LAUNCH("lcn_norm_kernel");
  return EBEN_OK;
}

extern "C" int eben_balance(const void* const* norms, const void* const* losses, int n, float* old, int init, int ema, float beta,
                            float one_minus_beta, float* lambdas, float* backprop, void* stream) {
  EBEN_REQUIRE(norms && losses && old && lambdas && backprop && n > 0 && n <= BAL_MAX, "balance: 1..%d losses", BAL_MAX);
  BalTable T;
  for (int i = 0; i < n; ++i) {
    EBEN_REQUIRE(norms[i] && losses[i], "balance: null term %d", i);
    T.norm[i] = static_cast<const float*>(norms[i]); T.loss[i] = static_cast<const float*>(losses[i]);
  }
  hipLaunchKernelGGL(balance_kernel, dim3(1), dim3(64), 0, as_stream(stream), T, old, n, init, ema, beta, one_minus_beta, lambdas, backprop);
  EBEN_CHECK_LAUNCH("balance_kernel");
  return EBEN_OK;
}
extern "C" int eben_weighted_sum(const void* const* tensors, const float* weights, int n, size_t numel, float* out, void* stream) {
  EBEN_REQUIRE(tensors && weights && out && n > 0 && n <= BAL_MAX && numel > 0, "weighted_sum: 1..%d tensors", BAL_MAX);
  WsumTable T;
  for (int i = 0; i < n; ++i) { EBEN_REQUIRE(tensors[i], "weighted_sum: null tensor %d", i); T.t[i] = static_cast<const float*>(tensors[i]); }
  hipLaunchKernelGGL(weighted_sum_kernel, dim3(grid_for(numel)), dim3(256), 0, as_stream(stream), T, weights, n, numel, out);
  EBEN_CHECK_LAUNCH("weighted_sum_kernel");
  return EBEN_OK;
}
// feature-matching value and the three hinge means of a step from the per-pair sums / per-term hinges, one launch:
//   out[0] = inv_count * sum_p s1_p / s2_p;   out[1 + k] = (1 / nchains) sum_i hinge[3 i + k]
__global__ __launch_bounds__(64) void disc_losses_kernel(const float* __restrict__ fm_sums, int npairs, float inv_count, const float* __restrict__ hinge,
                                                         int nchains, float* __restrict__ out) {
  const int lane = threadIdx.x;
  float f = 0.f;
  for (int p = lane; p < npairs; p += 64) f += fm_sums[2 * p] / fm_sums[2 * p + 1];
  f = wave_sum(f);
  if (lane == 0) out[0] = f * inv_count;
  if (lane < 3) {
    float h = 0.f;
    for (int i = 0; i < nchains; ++i) h += hinge[3 * i + lane];
    out[1 + lane] = h / (float)nchains;
  }
}
extern "C" int eben_disc_losses(const float* fm_sums, int npairs, float inv_count, const float* hinge, int nchains, float* out, void* stream) {
  EBEN_REQUIRE(fm_sums && hinge && out && npairs > 0 && nchains > 0, "bad disc_losses arguments");
  hipLaunchKernelGGL(disc_losses_kernel, dim3(1), dim3(64), 0, as_stream(stream), fm_sums, npairs, inv_count, hinge, nchains, out);
  EBEN_CHECK_LAUNCH("disc_losses_kernel");
  return EBEN_OK;
}
extern "C" int eben_hinge_fwd_multi(const void* const* xs, const int64_t* numel, const float* targets, int n, float* out, void* stream) {
  EBEN_REQUIRE(xs && numel && targets && out && n > 0 && n <= HINGE_MULTI, "hinge_fwd_multi: 1..%d terms", HINGE_MULTI);
  HingeTable T;
  for (int i = 0; i < n; ++i) {
    EBEN_REQUIRE(xs[i] && numel[i] > 0, "hinge_fwd_multi: empty term %d", i);
    T.x[i] = static_cast<const float*>(xs[i]); T.n[i] = numel[i]; T.target[i] = targets[i];
  }
  hipLaunchKernelGGL(hinge_fwd_multi_kernel, dim3(n), dim3(256), 0, as_stream(stream), T, out);
  EBEN_CHECK_LAUNCH("hinge_fwd_multi_kernel");
  return EBEN_OK;
}
extern "C" int eben_hinge_bwd(const float* x, size_t n, float target, const float* gout, float scale, float* dx, void* stream) {
  EBEN_REQUIRE(x && gout && dx && n > 0, "bad hinge arguments");
  hipLaunchKernelGGL(hinge_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), x, n, target, gout, scale, dx);
  EBEN_CHECK_LAUNCH("hinge_bwd_kernel");
  return EBEN_OK;
}

extern "C" int eben_hinge_bwd_stacked(const float* enhanced_logits, const float* reference_logits, size_t per, const float* gout, float scale_adv,
                                      float scale_fake, float scale_real, float* seeds, void* stream) {
  EBEN_REQUIRE(enhanced_logits && reference_logits && gout && seeds && per > 0, "bad stacked hinge arguments");
  hipLaunchKernelGGL(hinge_bwd_stacked_kernel, dim3(grid_for(4 * per)), dim3(256), 0, as_stream(stream), enhanced_logits, reference_logits, per, gout,
                     scale_adv, scale_fake, scale_real, seeds);
  EBEN_CHECK_LAUNCH("hinge_bwd_stacked_kernel");
  return EBEN_OK;
}

extern "C" int eben_l2norm(const float* x, size_t n, float* out, void* stream) {
  // out doubles as scratch for up to 256 partials: caller passes >= 257 floats, result in out[0]
  EBEN_REQUIRE(x && out && n > 0, "bad l2norm arguments");
  const int nb = grid_for(n, 256);
  hipLaunchKernelGGL(l2_partial_kernel, dim3(nb), dim3(256), 0, as_stream(stream), x, n, out + 1);
  EBEN_CHECK_LAUNCH("l2_partial_kernel");
  hipLaunchKernelGGL(l2_final_kernel, dim3(1), dim3(64), 0, as_stream(stream), out + 1, nb, out);
  EBEN_CHECK_LAUNCH("l2_final_kernel");
  return EBEN_OK;
}

extern "C" size_t eben_stft_loss_sums_workspace(int rows) { return sizeof(float) * 3 * (size_t)STFT_SPLIT * (rows > 0 ? rows : 0); }
extern "C" int eben_stft_loss_sums_ex(const float* spec_x, const float* spec_y, int rows, int bins, int frames, long long row_stride,
                                      long long bin_stride, long long im_off, float eps, float* partial_ws, size_t ws_bytes, float* out,
                                      void* stream) {
  EBEN_REQUIRE(spec_x && spec_y && out && partial_ws && rows > 0 && bins > 0 && frames > 0 && bin_stride >= frames, "bad stft_loss arguments");
  if (ws_bytes < eben_stft_loss_sums_workspace(rows)) return fail(EBEN_EWORKSPACE, "stft_loss_sums needs %zu workspace bytes", eben_stft_loss_sums_workspace(rows));
  const StftLayout L{row_stride, bin_stride, im_off};
  hipLaunchKernelGGL(stft_sums_kernel, dim3(STFT_SPLIT, rows), dim3(256), 0, as_stream(stream), spec_x, spec_y, bins, L, frames, eps,
                     partial_ws);
  EBEN_CHECK_LAUNCH("stft_sums_kernel");
  hipLaunchKernelGGL(stft_sums_final_kernel, dim3(rows), dim3(64), 0, as_stream(stream), partial_ws, out);
  EBEN_CHECK_LAUNCH("stft_sums_final_kernel");
  return EBEN_OK;
}
extern "C" int eben_stft_loss_total(const void* const* sums, const float* inv_counts, int n, int rows, float* out, void* stream) {
  EBEN_REQUIRE(sums && inv_counts && out && n > 0 && n <= STFT_TOTAL_MAX && rows > 0, "stft_loss_total: 1..%d resolutions", STFT_TOTAL_MAX);
  StftTotalTable T;
  for (int i = 0; i < n; ++i) {
    EBEN_REQUIRE(sums[i], "stft_loss_total: null sums %d", i);
    T.sums[i] = static_cast<const float*>(sums[i]); T.inv_count[i] = inv_counts[i];
  }
  hipLaunchKernelGGL(stft_total_kernel, dim3(1), dim3(64), 0, as_stream(stream), T, n, rows, out);
  EBEN_CHECK_LAUNCH("stft_total_kernel");
  return EBEN_OK;
}
extern "C" int eben_stft_loss_sums(const float* spec_x, const float* spec_y, int rows, int bins, int bins_pad, int frames,
                                   float eps, float* partial_ws, size_t ws_bytes, float* out, void* stream) {
  EBEN_REQUIRE(bins_pad >= bins, "bad stft_loss arguments");
  return eben_stft_loss_sums_ex(spec_x, spec_y, rows, bins, frames, 2LL * bins_pad * frames, frames, (long long)bins_pad * frames, eps,
                                partial_ws, ws_bytes, out, stream);
}
extern "C" int eben_stft_loss_bwd_ex(const float* spec_x, const float* spec_y, int rows, int bins, int frames, long long row_stride,
                                     long long bin_stride, long long im_off, float eps, const float* sums, const float* gout, float scale,
                                     float* dspec_x, long long out_row_stride, long long out_bin_stride, long long out_im_off, void* stream) {
  EBEN_REQUIRE(spec_x && spec_y && sums && gout && dspec_x && rows > 0 && bins > 0 && frames > 0, "bad stft_loss arguments");
  EBEN_REQUIRE(bin_stride >= frames && out_bin_stride >= frames, "bad stft_loss strides");
  const int nb = grid_for((size_t)bins * frames, 64);
  const StftLayout L{row_stride, bin_stride, im_off}, LO{out_row_stride, out_bin_stride, out_im_off};
  hipLaunchKernelGGL(stft_bwd_kernel, dim3(nb, rows), dim3(256), 0, as_stream(stream), spec_x, spec_y, rows, bins, L, LO, frames,
                     eps, sums, gout, scale, dspec_x);
  EBEN_CHECK_LAUNCH("stft_bwd_kernel");
  return EBEN_OK;
}
extern "C" int eben_stft_loss_bwd(const float* spec_x, const float* spec_y, int rows, int bins, int bins_pad, int frames,
                                  float eps, const float* sums, const float* gout, float scale, float* dspec_x, void* stream) {
  EBEN_REQUIRE(bins_pad >= bins, "bad stft_loss arguments");
  const long long rs = 2LL * bins_pad * frames, io = (long long)bins_pad * frames;
  return eben_stft_loss_bwd_ex(spec_x, spec_y, rows, bins, frames, rs, frames, io, eps, sums, gout, scale, dspec_x, rs, frames, io, stream);
}

extern "C" int eben_overlap_add_ex(const float* frames_buf, float* x, int batch, int lx, int win, int frames, int hop, int pad,
                                   int reflect, int accumulate, long long row_stride, long long j_stride, void* stream) {
  EBEN_REQUIRE(frames_buf && x && batch > 0 && lx > 0 && win > 0 && frames > 0 && hop > 0 && pad >= 0, "bad overlap_add arguments");
  EBEN_REQUIRE(!reflect || pad < lx, "reflect padding must be smaller than the signal");
  hipLaunchKernelGGL(overlap_add_kernel, dim3(grid_for((size_t)lx, 256), batch), dim3(256), 0, as_stream(stream), frames_buf, x, lx,
                     win, frames, hop, pad, reflect, accumulate, row_stride, j_stride);
  EBEN_CHECK_LAUNCH("overlap_add_kernel");
  return EBEN_OK;
}
extern "C" int eben_overlap_add(const float* frames_buf, float* x, int batch, int lx, int win, int frames, int hop, int pad,
                                int reflect, int accumulate, void* stream) {
  return eben_overlap_add_ex(frames_buf, x, batch, lx, win, frames, hop, pad, reflect, accumulate, (long long)win * frames, frames, stream);
}

extern "C" int eben_overlap_add_folded(const float* frames_buf, float* x, int batch, int lx, int win, int frames, int hop, int pad,
                                       int accumulate, long long row_stride, long long j_stride, void* stream) {
  EBEN_REQUIRE(frames_buf && x && batch > 0 && lx > 0 && win > 0 && (win & 1) == 0 && frames > 0 && hop > 0 && pad >= 0 && pad < lx,
               "bad overlap_add_folded arguments");
  static const bool tiled = !(getenv("EBEN_OLA_TILED") && atoi(getenv("EBEN_OLA_TILED")) == 0);
  const int fcap = OLA_U / hop + win / hop + pad / hop + 4;
  const size_t lds = sizeof(float) * (size_t)fcap * win;
  if (tiled && lds <= 64 * 1024 && pad < win && lx > 2 * pad) {
    hipLaunchKernelGGL(overlap_add_folded_t_kernel, dim3(ceil_div(lx, OLA_U), batch), dim3(256), lds, as_stream(stream), frames_buf, x, lx, win,
                       frames, hop, pad, accumulate, row_stride, j_stride, fcap);
    EBEN_CHECK_LAUNCH("overlap_add_folded_t_kernel");
    return EBEN_OK;
  }
  hipLaunchKernelGGL(overlap_add_folded_kernel, dim3(grid_for((size_t)lx, 256), batch), dim3(256), 0, as_stream(stream), frames_buf, x,
                     lx, win, frames, hop, pad, accumulate, row_stride, j_stride);
  EBEN_CHECK_LAUNCH("overlap_add_folded_kernel");
  return EBEN_OK;
}

// ---------------------------------------------------------------------------------------------
// STFT framing (im2col of torch.stft(center=True, pad_mode="reflect")): out[j, r*frames + f] = sig[r, reflect(f*hop + j - pad)].
// The (win, rows*frames) matrix makes the windowed DFT ONE dense GEMM over all items' frames (a pointwise tap-conv
// with batch 1): no N-tile padding per item (134 frames per item fill 52 % of two 128-column tiles).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stft_frames_kernel(const float* __restrict__ sig, float* __restrict__ out, int rows, int t,
                                                          int win, int hop, int pad, int frames) {
  const int j = blockIdx.y;
  const long long cols = (long long)rows * frames;
  float* o = out + (long long)j * cols;
  for (long long c = (long long)blockIdx.x * 256 + threadIdx.x; c < cols; c += (long long)gridDim.x * 256) {
    const int r = (int)(c / frames), f = (int)(c - (long long)r * frames);
    int q = f * hop + j - pad;
    q = q < 0 ? -q : q;
    q = q >= t ? 2 * (t - 1) - q : q;
    o[c] = sig[(long long)r * t + q];
  }
}
// Folded framing.  A window symmetric about sample h = win/2 of the frame (hann, periodic: w[0] = 0, w[h+m] = w[h-m]) centred
// in the DFT makes the real part a functional of the EVEN part of the frame about h and the imaginary part one of its ODD part:
//   E[m] = s[h+m] + s[h-m] (m >= 1), E[0] = s[h];  O[m] = s[h+m] - s[h-m], O[0] = 0        (m < h; sample 0 has zero weight)
//   Re X[k] = sum_m basis[k, h+m] E[m],   Im X[k] = sum_m basis[bins+k, h+m] O[m]
// -- a pointwise conv with TWO GROUPS of h channels instead of one of win: half the products.  split = 1 writes each group as
// 3h rows [hi ; lo ; hi] with hi = bf16(v), lo = bf16(v - hi) (both exactly representable in bf16): against weight rows
// [W_hi ; W_hi ; W_lo] a bf16-operand MFMA contraction with fp32 accumulation returns v.W to ~2^-17 relative (the dropped
// lo.lo and residual terms) -- "bf16x3".
__device__ __forceinline__ float bf16_rne(float v) {
  unsigned u = __float_as_uint(v);
  u += 0x7fffu + ((u >> 16) & 1u);
  return __uint_as_float(u & 0xffff0000u);
}
__global__ __launch_bounds__(256) void stft_frames_folded_kernel(const float* __restrict__ sig, float* __restrict__ out, int rows, int t,
                                                                 int win, int hop, int pad, int frames, int split) {
  const int m = blockIdx.y, h = win >> 1;
  const long long cols = (long long)rows * frames;
  const int nsub = split ? 3 : 1;
  float* oe = out + (long long)m * cols;
  float* oo = out + ((long long)nsub * h + m) * cols;
  const long long sub = (long long)h * cols;
  for (long long c = (long long)blockIdx.x * 256 + threadIdx.x; c < cols; c += (long long)gridDim.x * 256) {
    const int r = (int)(c / frames), f = (int)(c - (long long)r * frames);
    int q1 = f * hop + h + m - pad, q2 = f * hop + h - m - pad;
    q1 = q1 < 0 ? -q1 : q1; q1 = q1 >= t ? 2 * (t - 1) - q1 : q1;
    q2 = q2 < 0 ? -q2 : q2; q2 = q2 >= t ? 2 * (t - 1) - q2 : q2;
    const float a = sig[(long long)r * t + q1], b = sig[(long long)r * t + q2];
    const float e = m ? a + b : a, o = m ? a - b : 0.f;
    if (split) {
      const float eh = bf16_rne(e), oh = bf16_rne(o);
      oe[c] = eh; oe[c + sub] = bf16_rne(e - eh); oe[c + 2 * sub] = eh;
      oo[c] = oh; oo[c + sub] = bf16_rne(o - oh); oo[c + 2 * sub] = oh;
    } else {
      oe[c] = e; oo[c] = o;
    }
  }
}
// The same rows through an LDS transpose (split = 0): the kernel above reads the signal with the lanes on consecutive FRAMES, i.e. hop
// samples apart -- 64 cache lines per load instruction, 30-44 us for the 40 MB of one resolution.  Here a block owns (item, 64 frames, 64
// values of m): its loads run along m (consecutive samples: s[.. + m] ascending, s[.. - m] descending), the sums go through a 64 x 65 tile,
// its stores run along the frames (256 contiguous bytes per row).  Same sums of the same samples: bit-identical.
__global__ __launch_bounds__(256) void stft_frames_folded_t_kernel(const float* __restrict__ sig, float* __restrict__ out, int rows, int t,
                                                                   int win, int hop, int pad, int frames) {
  __shared__ float te[64][65], to[64][65];
  const int h = win >> 1;
  const int f0 = blockIdx.x * 64, m0 = blockIdx.y * 64, r = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long cols = (long long)rows * frames;
  const float* s = sig + (long long)r * t;
  const int m = m0 + lane;
  float av[16], bv[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {   // the 32 loads of this thread first: all in flight together
    const int f = f0 + wave + 4 * k;
    const bool ok = f < frames && m < h;
    int q1 = f * hop + h + m - pad, q2 = f * hop + h - m - pad;
    q1 = q1 < 0 ? -q1 : q1; q1 = q1 >= t ? 2 * (t - 1) - q1 : q1;
    q2 = q2 < 0 ? -q2 : q2; q2 = q2 >= t ? 2 * (t - 1) - q2 : q2;
    av[k] = ok ? s[q1] : 0.f;
    bv[k] = ok ? s[q2] : 0.f;
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int fl = wave + 4 * k;
    te[lane][fl] = m ? av[k] + bv[k] : av[k];
    to[lane][fl] = m ? av[k] - bv[k] : 0.f;
  }
  __syncthreads();
  const int f = f0 + lane;
  if (f >= frames) return;
  float* oe = out + (long long)r * frames + f;
  float* oo = oe + (long long)h * cols;
  for (int ml = wave; ml < 64 && m0 + ml < h; ml += 4) {
    oe[(long long)(m0 + ml) * cols] = te[ml][lane];
    oo[(long long)(m0 + ml) * cols] = to[ml][lane];
  }
}
extern "C" int eben_stft_frames_folded(const float* sig, float* out, int rows, int t, int win, int hop, int pad, int frames, int split,
                                       void* stream) {
  EBEN_REQUIRE(sig && out && rows > 0 && t > 1 && win > 1 && (win & 1) == 0 && hop > 0 && pad >= 0 && pad < t && frames > 0,
               "bad stft_frames_folded arguments");
  EBEN_REQUIRE((frames - 1) * hop + win - 1 - pad <= 2 * (t - 1), "stft_frames_folded: frames reach past the reflected signal");
  static const bool transposed = !(getenv("EBEN_STFT_FRAMES_T") && atoi(getenv("EBEN_STFT_FRAMES_T")) == 0);
  if (!split && transposed && rows <= 65535 && ceil_div(win / 2, 64) <= 65535) {
    hipLaunchKernelGGL(stft_frames_folded_t_kernel, dim3(ceil_div(frames, 64), ceil_div(win / 2, 64), rows), dim3(256), 0, as_stream(stream), sig,
                       out, rows, t, win, hop, pad, frames);
    EBEN_CHECK_LAUNCH("stft_frames_folded_t_kernel");
    return EBEN_OK;
  }
  hipLaunchKernelGGL(stft_frames_folded_kernel, dim3(grid_for((size_t)rows * frames, 64), win / 2), dim3(256), 0, as_stream(stream), sig,
                     out, rows, t, win, hop, pad, frames, split);
  EBEN_CHECK_LAUNCH("stft_frames_folded_kernel");
  return EBEN_OK;
}

// out (groups * 3 * R, cols) = per group [hi ; lo ; hi] of in (groups * R, cols): the bf16x3 operand of a gradient GEMM
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ in, float* __restrict__ out, int R, long long cols) {
  const int row = blockIdx.y;               // g * R + r
  const int g = row / R, r = row - g * R;
  const float* src = in + (long long)row * cols;
  float* dst = out + ((long long)g * 3 * R + r) * cols;
  const long long sub = (long long)R * cols;
  for (long long c = (long long)blockIdx.x * 256 + threadIdx.x; c < cols; c += (long long)gridDim.x * 256) {
    const float v = src[c], hi = bf16_rne(v);
    dst[c] = hi; dst[c + sub] = bf16_rne(v - hi); dst[c + 2 * sub] = hi;
  }
}
extern "C" int eben_split3(const float* in, float* out, int groups, int rows_per_group, long long cols, void* stream) {
  EBEN_REQUIRE(in && out && groups > 0 && rows_per_group > 0 && cols > 0 && (long long)groups * rows_per_group <= 65535, "bad split3 arguments");
  hipLaunchKernelGGL(split3_kernel, dim3(grid_for((size_t)cols, 16), groups * rows_per_group), dim3(256), 0, as_stream(stream), in, out,
                     rows_per_group, cols);
  EBEN_CHECK_LAUNCH("split3_kernel");
  return EBEN_OK;
}

extern "C" int eben_stft_frames(const float* sig, float* out, int rows, int t, int win, int hop, int pad, int frames, void* stream) {
  EBEN_REQUIRE(sig && out && rows > 0 && t > 1 && win > 0 && hop > 0 && pad >= 0 && pad < t && frames > 0, "bad stft_frames arguments");
  EBEN_REQUIRE((frames - 1) * hop + win - 1 - pad <= 2 * (t - 1), "stft_frames: frames reach past the reflected signal");
  hipLaunchKernelGGL(stft_frames_kernel, dim3(grid_for((size_t)rows * frames, 64), win), dim3(256), 0, as_stream(stream), sig, out, rows,
                     t, win, hop, pad, frames);
  EBEN_CHECK_LAUNCH("stft_frames_kernel");
  return EBEN_OK;
}

// ---------------------------------------------------------------------------------------------
// Waveform augmentation on the device (vibravox/torch_modules/dsp/data_augmentation.py:38-71):
//   * time masking (dsp/time_masking_waveform.py:18-36): x[..., first : first + count] = 0, in place;
//   * speed perturbation = torchaudio.functional.speed -> resample(source = int(factor * rate), target = rate): the
//     windowed-sinc polyphase interpolation of torchaudio.functional.resample ("sinc_interp_hann", lowpass width 6, rolloff
//     0.99), restated: with the rates reduced by their gcd to orig / new and `width` zero samples of left padding,
//         out[r, q*new + p] = sum_j kernel[p, j] * xpad[r, q*orig + j],   j < taps = 2*width + orig
//     (kernel table built on the host, vibravox_amd/augment.py).  HBM-bound: taps reads per output hit the L1/L2.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void time_mask_kernel(float* __restrict__ x, long long rows, int t, int first, int count) {
  const long long n = rows * count;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const long long r = i / count;
    x[r * t + first + (int)(i - r * count)] = 0.f;
  }
}
extern "C" int eben_time_mask(float* x, long long rows, int t, int first, int count, void* stream) {
  EBEN_REQUIRE(x && rows > 0 && t > 0 && first >= 0 && count >= 0 && first + count <= t, "bad time_mask arguments");
  if (count == 0) return EBEN_OK;
  hipLaunchKernelGGL(time_mask_kernel, dim3(grid_for((size_t)rows * count, 1024)), dim3(256), 0, as_stream(stream), x, rows, t, first, count);
  EBEN_CHECK_LAUNCH("time_mask_kernel");
  return EBEN_OK;
}

__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ x, const float* __restrict__ kernels, float* __restrict__ out,
                                                       int t_in, int t_out, int orig, int nw, int width, int taps) {
  const int r = blockIdx.y;
  const float* xr = x + (long long)r * t_in;
  float* o = out + (long long)r * t_out;
  for (int n = blockIdx.x * 256 + threadIdx.x; n < t_out; n += gridDim.x * 256) {
    const int q = n / nw, p = n - q * nw;
    const float* k = kernels + (long long)p * taps;
    const int base = q * orig - width;
    float acc = 0.f;
    for (int j = 0; j < taps; ++j) {
      const int i = base + j;
      if (i >= 0 && i < t_in) acc = fmaf(k[j], xr[i], acc);
    }
    o[n] = acc;
  }
}
extern "C" int eben_resample(const float* x, const float* kernels, float* out, int rows, int t_in, int t_out, int orig, int nw, int width,
                             void* stream) {
  EBEN_REQUIRE(x && kernels && out && rows > 0 && t_in > 0 && t_out > 0 && orig > 0 && nw > 0 && width >= 0, "bad resample arguments");
  EBEN_REQUIRE((long long)t_out <= ((long long)nw * t_in + orig - 1) / orig, "resample: t_out beyond ceil(new * t_in / orig)");
  hipLaunchKernelGGL(resample_kernel, dim3(grid_for((size_t)t_out, 256), rows), dim3(256), 0, as_stream(stream), x, kernels, out, t_in, t_out,
                     orig, nw, width, 2 * width + orig);
  EBEN_CHECK_LAUNCH("resample_kernel");
  return EBEN_OK;
}

// ---------------------------------------------------------------------------------------------
// Phase vocoder (torchaudio.functional.phase_vocoder as used by T.PitchShift, restated): time-stretch a complex
// spectrogram by `rate` without changing pitch.  Input / output spectra are flat (2*bins, rows*frames) matrices (real
// parts in rows [0, bins), imaginary in [bins, 2*bins)) like the STFT GEMM's.  One thread per (row, bin) walks the output
// frames in order (the phase is a running sum):
//   ts = i*rate, i0 = floor(ts), alpha = ts - i0, s0 = spec[i0], s1 = spec[i0+1] (zero past the end)
//   dphi = wrap(angle(s1) - angle(s0) - adv[k]) + adv[k];  phase_i = angle(spec[0]) + sum_{i' < i} dphi_i'
//   out_i = (alpha*|s1| + (1-alpha)*|s0|) * exp(j*phase_i),   adv[k] = pi*hop*k/(bins-1)
// (torchaudio runs this in the input's precision, complex64 here; float64 inside the walk costs nothing measurable and is the
// side of the reference's own rounding noise the float64 oracle sits on.)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void phase_vocoder_kernel(const float* __restrict__ spec, float* __restrict__ out, int rows, int bins,
                                                            int frames, int frames_out, double rate, float hop) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * bins) return;
  const int r = i / bins, k = i - r * bins;
  const long long in_cols = (long long)rows * frames, out_cols = (long long)rows * frames_out;
  const float* re = spec + (long long)k * in_cols + (long long)r * frames;
  const float* im = re + (long long)bins * in_cols;
  float* ore = out + (long long)k * out_cols + (long long)r * frames_out;
  float* oim = ore + (long long)bins * out_cols;
  // The phase is a running sum over a few hundred frames and each term is a difference of angles: angles, wrap and the
  // running sum are kept in float64 (2 atan2 + 1 sincos per output value — this is the collator's augmentation, a few
  // million values per batch), which holds the result at the fp32 STFT's own 1e-6 instead of 1e-3 of the signal.
  const double two_pi = 6.283185307179586476925286766559;
  const double adv = 3.14159265358979323846264338327950288 * (double)hop * (double)k / (double)(bins - 1);
  double phase = atan2((double)im[0], (double)re[0]);
  double a0 = phase;   // angle(spec[i0]) of the current frame pair, reused while i0 stands still
  int i0_prev = 0;
  for (int f = 0; f < frames_out; ++f) {
    const double ts = (double)f * rate;   // as torch.arange(0, frames, rate) in float64: the frame index must not flip
    const int i0 = (int)ts;
    const double alpha = ts - (double)i0;
    const float r0 = i0 < frames ? re[i0] : 0.f, m0 = i0 < frames ? im[i0] : 0.f;
    const float r1 = i0 + 1 < frames ? re[i0 + 1] : 0.f, m1 = i0 + 1 < frames ? im[i0 + 1] : 0.f;
    const double n0 = sqrt((double)r0 * r0 + (double)m0 * m0), n1 = sqrt((double)r1 * r1 + (double)m1 * m1);
    const double mag = alpha * n1 + (1.0 - alpha) * n0;
    double sn, cs;
    sincos(phase, &sn, &cs);
    ore[f] = (float)(mag * cs);
    oim[f] = (float)(mag * sn);
    if (i0 != i0_prev) { a0 = atan2((double)m0, (double)r0); i0_prev = i0; }
    double d = atan2((double)m1, (double)r1) - a0 - adv;
    d = d - two_pi * rint(d / two_pi);
    phase += d + adv;
  }
}
extern "C" int eben_phase_vocoder(const float* spec, float* out, int rows, int bins, int frames, int frames_out, double rate, float hop,
                                  void* stream) {
  EBEN_REQUIRE(spec && out && rows > 0 && bins > 1 && frames > 0 && frames_out > 0 && rate > 0.0, "bad phase_vocoder arguments");
  hipLaunchKernelGGL(phase_vocoder_kernel, dim3(grid_for((size_t)rows * bins, 1 << 20)), dim3(256), 0, as_stream(stream), spec, out, rows, bins,
                     frames, frames_out, rate, hop);
  EBEN_CHECK_LAUNCH("phase_vocoder_kernel");
  return EBEN_OK;
}

// ---------------------------------------------------------------------------------------------
// Noisy-BWE batch assembly on the device (vibravox/lightning_datamodules/noisybwe.py:219-291 with
// vibravox/utils.py:7-81,195-254): per item  bc[t] = speech[u] + noise[noise_start + u],  air[t] = airborne[u]
// with u = t + shift, zero outside [0, length) -- `shift` >= 0 is the crop offset of set_audio_duration,
// `shift` < 0 the (quirky) left zero run of pad_audio, 0 with `length` < T plain right padding.
// One gather pass, the ragged clips never leave HBM.  Items travel by value (48 per launch).
// ---------------------------------------------------------------------------------------------
constexpr int COLLATE_CHUNK = 48;
struct CollateTable { EbenCollateItem t[COLLATE_CHUNK]; };

__global__ __launch_bounds__(256) void noisy_collate_kernel(const CollateTable T, int samples, float* __restrict__ bc, float* __restrict__ air) {
  const EbenCollateItem it = T.t[blockIdx.y];
  float* obc = bc + (long long)blockIdx.y * samples;
  float* oair = air ? air + (long long)blockIdx.y * samples : nullptr;
  for (int t = blockIdx.x * 256 + threadIdx.x; t < samples; t += gridDim.x * 256) {
    const long long u = (long long)t + it.shift;
    const bool in = u >= 0 && u < it.length;
    float v = 0.f, a = 0.f;
    if (in) {
      v = it.speech[u];
      if (it.noise) v += it.noise[it.noise_start + u];
      if (it.airborne) a = it.airborne[u];
    }
    obc[t] = v;
    if (oair) oair[t] = a;
  }
}

extern "C" int eben_noisy_collate(const EbenCollateItem* items, int nitems, int samples, float* body_conducted, float* airborne, void* stream) {
  EBEN_REQUIRE(items && nitems > 0 && samples > 0 && body_conducted, "bad collate arguments");
  for (int p0 = 0; p0 < nitems; p0 += COLLATE_CHUNK) {
    const int cnt = nitems - p0 < COLLATE_CHUNK ? nitems - p0 : COLLATE_CHUNK;
    CollateTable T;
    for (int i = 0; i < cnt; ++i) {
      T.t[i] = items[p0 + i];
      if (!T.t[i].speech || T.t[i].length < 0 || T.t[i].noise_start < 0) return fail(EBEN_EINVAL, "collate item %d is malformed", p0 + i);
      if (airborne && !T.t[i].airborne) return fail(EBEN_EINVAL, "collate item %d has no airborne clip", p0 + i);
    }
    int gx = (samples + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(noisy_collate_kernel, dim3(gx, cnt), dim3(256), 0, as_stream(stream), T, samples,
                       body_conducted + (long long)p0 * samples, airborne ? airborne + (long long)p0 * samples : nullptr);
    EBEN_CHECK_LAUNCH("noisy_collate_kernel");
  }
  return EBEN_OK;
}

extern "C" int eben_adam_step(const EbenAdamTensor* table, int ntensors, int64_t max_numel, float lr, float beta1, float beta2,
                              float eps, float weight_decay, int step, float grad_scale, void* stream) {
  EBEN_REQUIRE(table && ntensors > 0 && step > 0, "bad adam arguments");
  const double bc1 = 1.0 - pow((double)beta1, step);
  const double bc2 = 1.0 - pow((double)beta2, step);
  for (int p0 = 0; p0 < ntensors; p0 += ADAM_CHUNK) {
    const int cnt = ntensors - p0 < ADAM_CHUNK ? ntensors - p0 : ADAM_CHUNK;
    AdamTable T;
    long long blocks = 0;
    for (int i = 0; i < cnt; ++i) {
      T.t[i] = table[p0 + i];
      if (!T.t[i].param || !T.t[i].grad || !T.t[i].exp_avg || !T.t[i].exp_avg_sq || T.t[i].numel <= 0)
        return fail(EBEN_EINVAL, "adam tensor %d is null or empty", p0 + i);
      blocks += (T.t[i].numel + ADAM_BLOCK_ELEMS - 1) / ADAM_BLOCK_ELEMS;
      if (blocks > 0x7fffffffLL) return fail(EBEN_EINVAL, "adam launch too large");
      T.bend[i] = (int)blocks;
    }
    for (int i = cnt; i < ADAM_CHUNK; ++i) T.bend[i] = (int)blocks;
    (void)max_numel;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), T, cnt, lr, beta1, beta2, eps,
                       weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale);
    EBEN_CHECK_LAUNCH("adam_kernel");
  }
  return EBEN_OK;
}
