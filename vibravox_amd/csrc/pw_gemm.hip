// pw_gemm.hip -- grouped GEMM y[g] = W[g] (M x K) * x[g] (K x N) with split-bf16 operands on the bf16 matrix pipe (gfx950): the
// windowed-DFT contractions of the MRSTFT loss (auraloss MultiResolutionSTFTLoss as configured by multi_stft.yaml; the folded even / odd
// form is a 2-group GEMM over win/2 reduction rows, forward K = 120 / 300 / 600, M = 257 / 513 / 1025, N = 64 items x frames) and their
// transposes in the backward.
//
// As pointwise tap-convs these launches spent their time staging: a (K x 128-column) tile converted into LDS by every row tile's block,
// one block per CU (130 KB of LDS), every phase of the block serial behind a barrier -- 170 us for the 2.5 GMAC of the 512-point
// resolution.  A GEMM needs no tile: a lane's B fragment of a k-step is eight reduction rows at ONE column, i.e. eight coalesced
// dword loads straight from x (row-major, columns contiguous), converted to bf16 pieces in registers.  So:
//   * block = 4 waves x 32 columns, one row tile of BM = 32 FM rows; no LDS for x, no staging barrier; the loads of chunk c + 1 fly
//     under the MFMAs of chunk c (registers are converted to pieces first, then reloaded);
//   * W: packed once (constant bases) as [group][row tile][chunk][k-step][piece][fm][lane] bf16x8 units, streamed through a
//     double-buffered LDS ring by LDS-DMA, one barrier per chunk of KSC k-steps (the tap-conv's weight stream);
//   * blocks of one column tile are adjacent in the grid (row tile fastest): the x columns they all read stay in L2;
//   * NP pieces per operand: 1 plain bf16, 2 hi + lo (three products, ~2^-17), 3 (six products, fp32-grade).
#include "common.h"

namespace eben {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int PG_KSC = 4;   // k-steps (of 16 reduction rows) per weight chunk

__device__ __forceinline__ unsigned pg_pack_bf16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
template <int NP>
__device__ __forceinline__ void pg_split8(float (&r)[8], u32x4 (&p)[NP]) {
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pg_pack_bf16(r[2 * e], r[2 * e + 1]);
    p[q] = o;
    if (q + 1 < NP) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {   // residual against the rounded value (a bf16 is the upper half of its fp32): exact in fp32
        r[2 * e] -= __builtin_bit_cast(float, o[e] << 16);
        r[2 * e + 1] -= __builtin_bit_cast(float, o[e] & 0xffff0000u);
      }
    }
  }
}

struct PwGemmArgs {
  const float* x; const u32x4* wp; float* y;
  int G, M, K, N, nmt, nch, nct;
};

template <int FM, int NP>
__global__ __launch_bounds__(256) void pw_gemm_kernel(const PwGemmArgs P) {
  constexpr int NT = 256, WCHU = PG_KSC * NP * FM * 64;
  extern __shared__ __attribute__((aligned(16))) u32x4 pg_smem[];   // 2 x WCHU
  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned id = blockIdx.x;
  const int mt = __builtin_amdgcn_readfirstlane((int)(id % (unsigned)P.nmt)); id /= (unsigned)P.nmt;
  const int ct = __builtin_amdgcn_readfirstlane((int)(id % (unsigned)P.nct));
  const int g = __builtin_amdgcn_readfirstlane((int)(id / (unsigned)P.nct));
  const int col = ct * 128 + wn * 32 + (lane & 31);
  const int colc = col < P.N ? col : P.N - 1;
  const int kh = lane >> 5;
  const float* __restrict__ xg = P.x + (long long)g * P.K * P.N + colc;
  const u32x4* __restrict__ wsrc = P.wp + ((long long)g * P.nmt + mt) * P.nch * WCHU;

  auto issue_w = [&](int ch) {
    const u32x4* src = wsrc + (long long)ch * WCHU;
    u32x4* dst = pg_smem + (ch & 1) * WCHU;
#pragma unroll
    for (int u = 0; u * NT < WCHU; ++u) {
      const int idx = u * NT + tid;
      if (WCHU % NT == 0 || idx < WCHU)   // wave-uniform
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + idx),
                                         (__attribute__((address_space(3))) void*)(dst + (idx & ~63)), 16, 0, 0);
    }
  };
  float xr[PG_KSC][8];
  auto load_x = [&](int ch) {
#pragma unroll
    for (int ks = 0; ks < PG_KSC; ++ks) {
      const int k0 = (ch * PG_KSC + ks) * 16 + 8 * kh;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = k0 + e < P.K ? k0 + e : 0;   // rows beyond K: clamped address, zeroed at the conversion
        xr[ks][e] = xg[(long long)k * P.N];
      }
    }
  };

  f32x16 acc[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  issue_w(0);
  load_x(0);
  for (int ch = 0; ch < P.nch; ++ch) {
    // chunk ch's x registers -> bf16 pieces (this waits for their loads only: the DMA of chunk ch was issued before them)
    u32x4 bv[PG_KSC][NP];
#pragma unroll
    for (int ks = 0; ks < PG_KSC; ++ks) {
      const int k0 = (ch * PG_KSC + ks) * 16 + 8 * kh;
      float t[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] = k0 + e < P.K ? xr[ks][e] : 0.f;
      pg_split8<NP>(t, bv[ks]);
    }
    __syncthreads();   // chunk ch's weights are in LDS (every wave's DMA pieces have landed), chunk ch - 1's buffer is free
    if (ch + 1 < P.nch) { issue_w(ch + 1); load_x(ch + 1); }
    const u32x4* wb = pg_smem + (ch & 1) * WCHU + lane;
#pragma unroll
    for (int ks = 0; ks < PG_KSC; ++ks) {
      u32x4 a[NP][FM];
#pragma unroll
      for (int q = 0; q < NP; ++q)
#pragma unroll
        for (int i = 0; i < FM; ++i) a[q][i] = wb[((ks * NP + q) * FM + i) * 64];
      // piece products, smallest first: (qw, qx) with qw + qx = lvl
#pragma unroll
      for (int lvl = NP - 1; lvl >= 0; --lvl)
#pragma unroll
        for (int qw = 0; qw <= lvl; ++qw)
#pragma unroll
          for (int i = 0; i < FM; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[qw][i]), __builtin_bit_cast(bf16x8, bv[ks][lvl - qw]),
                                                             acc[i], 0, 0, 0);
    }
  }

  // ---- D tile: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) ----
  if (col >= P.N) return;
  float* __restrict__ yb = P.y + (long long)g * P.M * P.N + col;
  const int m0 = mt * 32 * FM + 4 * kh;
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2);
      if (m < P.M) yb[(long long)m * P.N] = acc[i][r];
    }
}

// W (G M, K) fp32 -> [group][row tile][chunk][k-step][piece][fm][lane] units of eight bf16 (zero beyond M / K)
struct PwPackArgs { const float* w; u32x4* wp; int G, M, K, nmt, nch, FM, NP; long long units; };
__global__ __launch_bounds__(256) void pw_gemm_pack_kernel(const PwPackArgs P) {
  for (long long u = (long long)blockIdx.x * 256 + threadIdx.x; u < P.units; u += (long long)gridDim.x * 256) {
    long long r = u;
    const int lane = (int)(r & 63); r >>= 6;
    const int i = (int)(r % P.FM); r /= P.FM;
    const int q = (int)(r % P.NP); r /= P.NP;
    const int ks = (int)(r % PG_KSC); r /= PG_KSC;
    const int ch = (int)(r % P.nch); r /= P.nch;
    const int mt = (int)(r % P.nmt);
    const int g = (int)(r / P.nmt);
    const int m = (mt * P.FM + i) * 32 + (lane & 31);
    const int k0 = (ch * PG_KSC + ks) * 16 + 8 * (lane >> 5);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (m < P.M && k0 + e < P.K) ? P.w[((long long)g * P.M + m) * P.K + k0 + e] : 0.f;
    u32x4 o;
    for (int qq = 0;; ++qq) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = pg_pack_bf16(v[2 * e], v[2 * e + 1]);
      if (qq == q) break;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[2 * e] -= __builtin_bit_cast(float, o[e] << 16);
        v[2 * e + 1] -= __builtin_bit_cast(float, o[e] & 0xffff0000u);
      }
    }
    P.wp[u] = o;
  }
}

struct PwPlan { int ok, FM, NP, nmt, nch; long long units; };
static PwPlan pw_plan(int math, int G, int M, int K) {
  PwPlan p{};
  p.NP = math == EBEN_MATH_BF16 ? 1 : math == EBEN_MATH_BF16X3 ? 2 : math == EBEN_MATH_BF16X6 ? 3 : 0;
  if (!p.NP || G <= 0 || M <= 0 || K <= 0) return p;
  // 64-row tiles: two blocks per CU by registers with three pieces per operand, and the least padding of M = 257 / 513 / 1025 (or 120 / 300 / 600)
  static const int force = getenv("EBEN_PWGEMM_FM") ? atoi(getenv("EBEN_PWGEMM_FM")) : 0;
  p.FM = force >= 1 && force <= 4 ? force : (M <= 32 ? 1 : 2);
  // hi + lo operands: 96-row tiles where they pad M no more than 64-row ones (+6 %) -- every row tile's block loads and splits the x columns
  // again, so fewer, taller tiles do less of that ([MI355X] forward M = 257 / 513 / 1025: 45 / 66 / 102 -> 38 / 58 / 91 us; the transposes
  // M = 120 / 300 keep 64 rows: 23 / 36 us against 27 / 42)
  if (!force && p.NP == 2 && M > 64 && ceil_div(M, 96) * 3 * 100 <= ceil_div(M, 64) * 2 * 106) p.FM = 3;
  p.nmt = ceil_div(M, 32 * p.FM);
  p.nch = ceil_div(ceil_div(K, 16), PG_KSC);
  p.units = (long long)G * p.nmt * p.nch * PG_KSC * p.NP * p.FM * 64;
  p.ok = 1;
  return p;
}

template <int FM, int NP>
static int pw_launch(const PwGemmArgs& a, long long nb, hipStream_t st) {
  hipLaunchKernelGGL((pw_gemm_kernel<FM, NP>), dim3((unsigned)nb), dim3(256), (size_t)2 * PG_KSC * NP * FM * 64 * 16, st, a);
  EBEN_CHECK_LAUNCH("pw_gemm_kernel");
  return EBEN_OK;
}

}  // namespace eben

using namespace eben;

extern "C" size_t eben_gemm_packed_floats(int math, int groups, int m, int k) {
  const PwPlan p = pw_plan(math, groups, m, k);
  return p.ok ? (size_t)p.units * 4 : 0;
}

extern "C" int eben_gemm_pack(int math, int groups, int m, int k, const float* w, float* wp, void* stream) {
  const PwPlan p = pw_plan(math, groups, m, k);
  EBEN_REQUIRE(p.ok, "gemm_pack: EBEN_MATH_BF16 / BF16X3 / BF16X6 and positive sizes (math %d, %d x %d x %d)", math, groups, m, k);
  EBEN_REQUIRE(w && wp, "null pointer in gemm_pack");
  PwPackArgs a{w, reinterpret_cast<u32x4*>(wp), groups, m, k, p.nmt, p.nch, p.FM, p.NP, p.units};
  long long nb = (p.units + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(pw_gemm_pack_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), a);
  EBEN_CHECK_LAUNCH("pw_gemm_pack_kernel");
  return EBEN_OK;
}

extern "C" int eben_gemm_fwd(int math, int groups, int m, int k, long long n, const float* x, const float* wp, float* y, void* stream) {
  const PwPlan p = pw_plan(math, groups, m, k);
  EBEN_REQUIRE(p.ok, "gemm_fwd: EBEN_MATH_BF16 / BF16X3 / BF16X6 and positive sizes (math %d, %d x %d x %d)", math, groups, m, k);
  EBEN_REQUIRE(x && wp && y && n > 0, "bad gemm_fwd arguments");
  EBEN_REQUIRE(n < (1ll << 31) && (long long)k * n < (1ll << 40), "gemm_fwd: operand too large");
  PwGemmArgs a;
  a.x = x; a.wp = reinterpret_cast<const u32x4*>(wp); a.y = y;
  a.G = groups; a.M = m; a.K = k; a.N = (int)n; a.nmt = p.nmt; a.nch = p.nch; a.nct = (int)((n + 127) / 128);
  const long long nb = (long long)groups * a.nct * a.nmt;
  if (nb > 0x7fffffffLL) return fail(EBEN_EINVAL, "gemm_fwd grid too large");
  hipStream_t st = as_stream(stream);
  if (p.FM == 1) {
    switch (p.NP) { case 1: return pw_launch<1, 1>(a, nb, st); case 2: return pw_launch<1, 2>(a, nb, st); default: return pw_launch<1, 3>(a, nb, st); }
  }
  if (p.FM == 3 && p.NP == 2) return pw_launch<3, 2>(a, nb, st);
  if (p.FM == 4 && p.NP == 2) return pw_launch<4, 2>(a, nb, st);
  if (p.FM > 2) return fail(EBEN_EUNSUPPORTED, "gemm_fwd: 96 / 128-row tiles are built for hi + lo operands only");
  switch (p.NP) { case 1: return pw_launch<2, 1>(a, nb, st); case 2: return pw_launch<2, 2>(a, nb, st); default: return pw_launch<2, 3>(a, nb, st); }
}
