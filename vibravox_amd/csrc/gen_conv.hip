// gen_conv.hip -- the generator's strided / transposed / bottleneck convolutions with fp32 tensors at rest and split-bf16 products (gfx950).
//
// EBENGenerator's eight non-residual convs (eben_generator.py:241-249 EncBlock.conv k = 2S stride S reflect, :272-280 DecBlock.conv_trans
// k = 2S stride S, :295-312 the two k = 7 latent convs) are 2-4 GFLOP each over 4-65 MB: neither the matrix cores nor HBM bound them, the
// shape of the launch does.  The general tap-conv (tapconv3.hip) stages a 128-column input tile per block through LDS as three bf16 pieces
// (100-150 KB: one block per CU), streams the weights in chunks of 2 k-steps with a barrier each, and gives every output phase of a
// transposed conv a block of its own that re-stages the same tile: [MI355X] 36 / 47 / 90 / 56 / 17 / 49 / 75 / 44 us for the eight layers
// of BASELINE config 2 (0.41 of the generator's 1.32 ms forward) against ~5-10 us of matrix work each.
//
// Here the MFMA B operand never touches LDS.  v_mfma_f32_32x32x16_bf16 wants, per lane, eight consecutive reduction elements of ONE
// column; with the reduction ordered (channel, tap) those are CONSECUTIVE SAMPLES of one input row:
//   form 0  Conv1d, k <= 16, dilation 1: a K-group = 8 taps of one channel = x[c, S u + off + 8 h .. + 7]   (two 16-byte loads)
//   form 1  Conv1d, k <= 4:              a K-group = 4 taps of two channels                                    (two 16-byte loads)
//   form 2  ConvTranspose1d k = 2S, padding S / 2: output phase p of column u (= input position) reads the PAIR x[c, u - 1 .. u]
//           (p < S / 2) or x[c, u .. u + 1] (p >= S / 2); a K-group = 4 channels x 2 samples (four 8-byte loads).  The S / 2 phases of a
//           half share the B operand, so they are ROWS of one GEMM: rows (output channel, phase), phase fastest -- a lane's four
//           consecutive accumulator rows are consecutive output samples (one 16-byte store at S = 8).
// Each lane loads its eight fp32 values straight from global memory (L1 / L2 serve the overlap between neighbouring columns and taps),
// splits them into the bf16 pieces in registers and feeds the MFMAs; the fp32 -> pieces conversion happens once per (column, K-group)
// and wave, amortised over RT row tiles x 6 products.  Only the weights go through LDS: image [half][row tile][k-step][piece][lane] of
// 16-byte units built by gc_pack_kernel, streamed in chunks of 8 / RT k-steps (24 KB slots, two of them: three blocks per CU).
// Block = 4 waves = 4 x CT column tiles of 32 against the same RT row tiles; grid = halves x row groups x items x column blocks.
// Tiles that touch the ends of the signal (reflect / zero padding) or the end of the column range take a per-element path.
//
// EBEN_MATH_BF16X6 (three pieces per operand, six products, fp32-grade) is what the generator's forward computes in (gen_engine.py
// CONV_FWD_MATH); the piece count is a template parameter.
#include "common.h"

#include <cstdlib>
#include <type_traits>

namespace eben {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct GcArgs {
  const float* x; float* y; const u32x4* wp; const float* bias; const float* res;
  int B, CX, LX, CY, LY, NU, S;
  int rpc_shift, off0, KSP, nkg;   // conv: runs of 8 taps per channel = 1 << rpc_shift; off0 = -pad_l
  int nrt, nrg, ncb, nh;                  // 32-row tiles per half, row groups of RT tiles, column blocks per item, halves
  int reflect;
  float in_slope, out_slope, res_slope;
  long long w_rt, w_half;                 // image strides in 16-byte units
  unsigned xq, xr;                        // gridDim.x / 8, gridDim.x % 8 (xcd_remap)
};

__device__ __forceinline__ unsigned gc_pack2(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));   // v_cvt_pk_bf16_f32 (RNE)
}

// piece q = bf16(v - p0 - .. - p(q-1)): every residual is exact in fp32 (a bf16 is the upper half of its fp32)
template <int NP>
__device__ __forceinline__ void gc_split(const float (&v)[8], u32x4 (&pc)[NP]) {
  float t[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) t[e] = v[e];
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    u32x4 o;
    o[0] = gc_pack2(t[0], t[1]); o[1] = gc_pack2(t[2], t[3]); o[2] = gc_pack2(t[4], t[5]); o[3] = gc_pack2(t[6], t[7]);
    pc[q] = o;
    if (q + 1 < NP) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        t[2 * e] -= __builtin_bit_cast(float, o[e] << 16);
        t[2 * e + 1] -= __builtin_bit_cast(float, o[e] & 0xffff0000u);
      }
    }
  }
}

#ifndef EBEN_GC_DBG
#define EBEN_GC_DBG 0   // scratch-build ablations: 1 no MFMAs, 2 no input loads, 4 no weight stream, 8 no stores (1-8: wrong results), 16 no slot schedule
#endif

template <int FORM, int RT, int CT, int NP>
__global__ __launch_bounds__(256, 2) void gc_kernel(const GcArgs P) {
  constexpr int KC = (RT == 2 && CT == 2) ? 2 : 8 / RT;   // k-steps per weight chunk (64 x 64 wave tiles: the samples of 4 k-steps in flight beside 64 accumulator registers spill)
  constexpr int RTU = KC * NP * 64;           // units of one row tile's chunk (contiguous in the image)
  constexpr int SLOT = RT * RTU;              // units per LDS slot
  __shared__ __attribute__((aligned(16))) u32x4 Ws[2 * SLOT];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, hk = lane >> 5;

  unsigned id;
  {
    const unsigned bid = blockIdx.x, xcd = bid & 7u, idx = bid >> 3;
    id = (xcd < P.xr ? xcd * (P.xq + 1) : P.xr * (P.xq + 1) + (xcd - P.xr) * P.xq) + idx;
  }
  // row group fastest, then half: the blocks that read the same input window are neighbours in time on one XCD (its L2 serves the re-reads)
  // (the divisions run on the vector unit: readfirstlane brings the block-uniform results back into scalar registers, and with them
  // every pointer derived from them)
  const int rg = __builtin_amdgcn_readfirstlane((int)(id % (unsigned)P.nrg)); id /= (unsigned)P.nrg;
  const int h = __builtin_amdgcn_readfirstlane((int)(id % (unsigned)P.nh)); id /= (unsigned)P.nh;
  const int cb = __builtin_amdgcn_readfirstlane((int)(id % (unsigned)P.ncb));
  const int b = __builtin_amdgcn_readfirstlane((int)(id / (unsigned)P.ncb));

  const float* xb = P.x + (long long)b * P.CX * P.LX;
  const u32x4* wsrc = P.wp + (long long)h * P.w_half + (long long)(rg * RT) * P.w_rt;
  const int nch = P.KSP / KC;

  // weights: chunk ch + 1 is loaded into registers at the top of chunk ch and written to the other LDS slot at its end (one barrier per
  // chunk).  Plain loads on purpose: hipcc counts them together with the sample loads (everything returns in order), so every wait
  // it places is a counted vmcnt -- an LDS-DMA it does not know about would sit between them and turn each wait for a sample into a
  // wait for the DMA issued after it
  constexpr int WPT = SLOT / 256;             // units per thread and chunk
  typedef const __attribute__((address_space(1))) char* gptr_t;   // global address space: a laundered integer must not come back as a flat pointer
  static_assert(SLOT % 256 == 0, "a slot is a whole number of block-wide pieces");
  u32x4 wreg[WPT];
  auto load_w = [&](int ch) {
    if (EBEN_GC_DBG & 4) return;
#pragma unroll
    for (int p = 0; p < WPT; ++p) {
      // unit p 256 + tid of the slot: row tile idx / RTU, then the chunk's units as they lie in the image -- a wave-uniform base (scalar
      // register pair) plus a 32-bit per-thread offset: one address register for all the loads
      if constexpr (RTU % 256 == 0) {
        const u32x4* base = wsrc + (long long)((p * 256) / RTU) * P.w_rt + (long long)ch * RTU + (p * 256) % RTU;
        unsigned long long a = reinterpret_cast<unsigned long long>(base);
        asm volatile("" : "+s"(a));
        wreg[p] = *reinterpret_cast<const __attribute__((address_space(1))) u32x4*>(reinterpret_cast<gptr_t>(a) + (unsigned)tid * 16u);
      } else {   // a block-wide piece straddles two row tiles: the tile is part of the thread's offset
        const int idx = p * 256 + tid;
        const unsigned off = (unsigned)((idx / RTU) * (int)P.w_rt + idx % RTU) * 16u;
        unsigned long long a = reinterpret_cast<unsigned long long>(wsrc + (long long)ch * RTU);
        asm volatile("" : "+s"(a));
        wreg[p] = *reinterpret_cast<const __attribute__((address_space(1))) u32x4*>(reinterpret_cast<gptr_t>(a) + off);
      }
    }
  };
  auto store_w = [&](int ch) {
    if (EBEN_GC_DBG & 4) return;
#pragma unroll
    for (int p = 0; p < WPT; ++p) Ws[(ch & 1) * SLOT + p * 256 + tid] = wreg[p];
  };

  // ---- the wave's column tiles ----
  const int u0w = (cb * 4 + wave) * (32 * CT);
  int ucol[CT];      // this lane's column (clamped into the range: the loads of columns that do not exist re-read the last one)
  bool fast[CT];     // wave-uniform: every lane's eight samples of every K-group lie inside the row
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int u0 = u0w + ct * 32;
    const int u = u0 + n;
    ucol[ct] = u < P.NU ? u : P.NU - 1;
    bool f = u0 + 31 < P.NU;
    if (FORM == 0) f = f && P.S * u0 + P.off0 >= 0 && P.S * (u0 + 31) + P.off0 + (8 << P.rpc_shift) <= P.LX;
    if (FORM == 1) f = f && P.S * u0 + P.off0 >= 0 && P.S * (u0 + 31) + P.off0 + 4 <= P.LX;
    if (FORM == 2) f = f && (h == 0 ? u0 >= 1 : u0 + 32 < P.LX);
    fast[ct] = f;
  }
  const bool act_in = P.in_slope != 1.f;
  const int ks_real = P.nkg >> 1;
  // Address of a sample = row pointer of the k-step (wave-uniform: a scalar base) + a per-lane offset that does not depend on the
  // k-step: k-step ks covers RKS consecutive channel rows, the lane's half (hk) and column pick the row within them and the position.
  const int RKS = FORM == 0 ? (P.rpc_shift ? 1 : 2) : FORM == 1 ? 4 : 8;
  int loff[CT];           // fast path: offset (floats) of the lane's first sample from the k-step's row pointer
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int u = ucol[ct];
    if (FORM == 0) loff[ct] = (P.rpc_shift ? 8 * hk : hk * P.LX) + P.S * u + P.off0;
    else if (FORM == 1) loff[ct] = 2 * hk * P.LX + P.S * u + P.off0;
    else loff[ct] = 4 * hk * P.LX + u - 1 + h;
  }
  auto row_ptr = [&](int ks) -> gptr_t {
    const int kc = ks < ks_real ? ks : ks_real - 1;   // padding k-steps (zero weights) re-read the last real one
    const float* rp = xb + (long long)(kc * RKS) * P.LX;
    // pinned in a scalar register pair and opaque to the loop optimiser, which otherwise turns every load of the unrolled chunk into an
    // induction variable of its own (a 64-bit address pair per load: 32-64 registers, spills in the larger tile shapes)
    unsigned long long a = reinterpret_cast<unsigned long long>(rp);
    asm volatile("" : "+s"(a));
    return reinterpret_cast<gptr_t>(a);
  };
  // FAST (compile-time, chosen once per wave): no branch sits between the loads of the main loop -- hipcc's vmcnt bookkeeping turns
  // conservative at every control-flow join (it drained the whole prefetch at each chunk end when the per-tile fast / slow choice was
  // a branch around the loads).  The slow form (waves whose tiles touch an end of the rows or of the column range) asks for the eight
  // samples one by one at offsets fixed up once per lane: reflected or, for zero padding, clamped and zeroed on arrival.
  struct Slow { int off[8]; unsigned ok; };
  auto slow_init = [&](int ct, Slow& sl) {
    const int u = ucol[ct];
    sl.ok = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int crow, pos;
      if (FORM == 0) { crow = P.rpc_shift ? 0 : hk; pos = P.S * u + P.off0 + (P.rpc_shift ? 8 * hk : 0) + e; }
      else if (FORM == 1) { crow = 2 * hk + (e >> 2); pos = P.S * u + P.off0 + (e & 3); }
      else { crow = 4 * hk + (e >> 1); pos = u - 1 + h + (e & 1); }
      if (P.reflect) {
        pos = pos < 0 ? -pos : pos;
        pos = pos >= P.LX ? 2 * (P.LX - 1) - pos : pos;
      }
      const bool ok = pos >= 0 && pos < P.LX;
      sl.off[e] = crow * P.LX + (ok ? pos : 0);
      sl.ok |= (unsigned)ok << e;
    }
  };
  auto gather = [&](auto fast_tag, int ks, int ct, const Slow& sl, float (&v)[8]) {
    constexpr bool FAST = decltype(fast_tag)::value;
    if (EBEN_GC_DBG & 2) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 1.f;
      return;
    }
    const gptr_t rp = row_ptr(ks);
    // scalar base + 32-bit BYTE offset per lane (the form the global_load saddr addressing takes: one offset register per tile, not
    // an address pair per load); 4-byte aligned vector loads (the rows start anywhere)
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
    typedef const __attribute__((address_space(1))) f4u* g4_t;
    typedef const __attribute__((address_space(1))) f2u* g2_t;
    typedef const __attribute__((address_space(1))) float* g1_t;
    if constexpr (FAST) {
      const unsigned o = (unsigned)loff[ct] * 4u;
      if (FORM == 0) {
        const f4u a0 = *reinterpret_cast<g4_t>(rp + o), a1 = *reinterpret_cast<g4_t>(rp + 16 + o);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = a0[e]; v[4 + e] = a1[e]; }
      } else if (FORM == 1) {
        const f4u a0 = *reinterpret_cast<g4_t>(rp + o), a1 = *reinterpret_cast<g4_t>(rp + (long long)P.LX * 4 + o);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = a0[e]; v[4 + e] = a1[e]; }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const f2u a = *reinterpret_cast<g2_t>(rp + (long long)r * P.LX * 4 + o);
          v[2 * r] = a[0]; v[2 * r + 1] = a[1];
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = *reinterpret_cast<g1_t>(rp + (unsigned)sl.off[e] * 4u);
    }
  };
  // samples outside the row (zero padding), after arrival
  auto slow_mask = [&](const Slow& sl, float (&v)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (sl.ok >> e) & 1u ? v[e] : 0.f;
  };
  // LeakyReLU on load (latent_conv[1] reads the encoder output through the activation): applied when the samples are consumed, not
  // when they are asked for
  auto activate = [&](auto act_tag, float (&v)[8]) {
    if constexpr (decltype(act_tag)::value) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = lrelu(v[e], P.in_slope);
    }
  };

  f32x16 acc[CT][RT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ct][i][r] = 0.f;

  // Software pipeline, one slot per k-step ks (all compile-time unrolled inside a chunk):
  //   * the MFMAs of ks run on pieces split in the PREVIOUS slot,
  //   * the samples of ks + 1 (asked for 8 / RT slots ago) are split into pieces -- ~44 vector instructions per column tile that the
  //     scheduler interleaves with the slot's MFMAs (sched_group_barrier below): with one wave per SIMD (the 125-column layers)
  //     nothing else would run under the matrix pipe, [MI355X] 128-channel stride-8 layer 69 us of which 38 were MFMAs issued back to
  //     back behind the splits,
  //   * the registers of ks + 1 take the load of ks + 1 + 8 / RT.
  // hipcc moves vector work freely inside a basic block: without the sched_barrier between slots it hoisted the consumption of every
  // k-step of a chunk to the top of the chunk (one wait for everything, no load under compute).
  auto run = [&](auto fast_tag, auto act_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
    float v[KC][CT][8];
    Slow sl[CT];
    if constexpr (!FAST) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) slow_init(ct, sl[ct]);
    }
    auto pieces = [&](int j, u32x4 (&out)[CT][NP]) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        if constexpr (!FAST) slow_mask(sl[ct], v[j][ct]);
        activate(act_tag, v[j][ct]);
        gc_split<NP>(v[j][ct], out[ct]);
      }
    };
    load_w(0);
#pragma unroll
    for (int ksl = 0; ksl < KC; ++ksl)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) gather(fast_tag, ksl, ct, sl[ct], v[ksl][ct]);
    store_w(0);
    u32x4 bq[CT][NP];
    pieces(0, bq);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) gather(fast_tag, KC, ct, sl[ct], v[0][ct]);
    __syncthreads();
    auto chunk = [&](int ch, auto more_tag) {
      constexpr bool MORE = decltype(more_tag)::value;
      if constexpr (MORE) load_w(ch + 1);
      const u32x4* wb = Ws + (ch & 1) * SLOT + lane;
#pragma unroll
      for (int ksl = 0; ksl < KC; ++ksl) {
        constexpr int VPM = (44 * CT + 6 * RT * CT - 1) / (6 * RT * CT);   // vector instructions of the split per MFMA of the slot
        const int nx = (ksl + 1) % KC;
        const bool has_next = MORE || ksl + 1 < KC;
        u32x4 aq[NP][RT];
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
          for (int i = 0; i < RT; ++i) aq[q][i] = wb[i * RTU + (ksl * NP + q) * 64];
        u32x4 bqn[CT][NP];
        if (has_next) pieces(nx, bqn);
        if constexpr (MORE) {
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) gather(fast_tag, ch * KC + ksl + 1 + KC, ct, sl[ct], v[nx][ct]);
        }
        // piece products, smallest first: (qw, qx) with qw + qx = lvl
#pragma unroll
        for (int lvl = NP - 1; lvl >= 0; --lvl)
#pragma unroll
          for (int qw = 0; qw < NP; ++qw) {
            const int qx = lvl - qw;
            if (qx < 0 || qx >= NP) continue;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
              for (int i = 0; i < RT; ++i) {
                if (EBEN_GC_DBG & 1) acc[ct][i][0] += __builtin_bit_cast(float, aq[qw][i][0] ^ bq[ct][qx][1]);
                else acc[ct][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aq[qw][i]), __builtin_bit_cast(bf16x8, bq[ct][qx]), acc[ct][i], 0, 0, 0);
              }
          }
        if (has_next) {
#pragma unroll
          for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int q = 0; q < NP; ++q) bq[ct][q] = bqn[ct][q];
        }
        if (!(EBEN_GC_DBG & 16)) {
          // the slot's schedule: the fragment reads, a first run of vector work under their latency, then MFMA / vector alternating
          __builtin_amdgcn_sched_group_barrier(0x100, NP * RT, 0);   // DS reads
          __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
#pragma unroll
          for (int m = 0; m < 6 * RT * CT; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);     // its share of the split
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (MORE) {
        store_w(ch + 1);   // the slot of chunk ch - 1: every wave is past the barrier that ended it
        __syncthreads();
      }
    };
    for (int ch = 0; ch + 1 < nch; ++ch) chunk(ch, std::true_type{});
    chunk(nch - 1, std::false_type{});
  };
  bool allfast = true;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) allfast = allfast && fast[ct];
  if (allfast) {
    if (act_in) run(std::true_type{}, std::true_type{});
    else run(std::true_type{}, std::false_type{});
  } else {
    if (act_in) run(std::false_type{}, std::true_type{});
    else run(std::false_type{}, std::false_type{});
  }

  // ---- epilogue: accumulator r of a tile = row 8 (r / 4) + 4 hk + r % 4, column n ----
  if (EBEN_GC_DBG & 8) {
    float s = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[ct][i][r];
    if (s == 1.2345f) P.y[0] = s;
    return;
  }
  float* yb = P.y + (long long)b * P.CY * P.LY;
  const float* rb = P.res ? P.res + (long long)b * P.CY * P.LY : nullptr;   // + lrelu(residual, res_slope) behind the output activation
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int u = u0w + ct * 32 + n;
    if (u >= P.NU) continue;
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const int mt = (rg * RT + i) * 32;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int m = mt + 8 * r4 + 4 * hk;   // first of the lane's four consecutive rows
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = acc[ct][i][4 * r4 + r];
        if (FORM != 2) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float t = o[r] + (P.bias ? P.bias[m + r] : 0.f);
            const long long yi = (long long)(m + r) * P.LY + u;
            yb[yi] = lrelu(t, P.out_slope) + (rb ? lrelu(rb[yi], P.res_slope) : 0.f);
          }
        } else if (P.S == 8) {          // rows (channel, phase): four phases of one channel = 16 contiguous bytes
          const int co = m >> 2;
          const float bv = P.bias ? P.bias[co] : 0.f;
          f32x4 t;
#pragma unroll
          for (int r = 0; r < 4; ++r) t[r] = lrelu(o[r] + bv, P.out_slope);
          const long long yi = (long long)co * P.LY + 8 * u + 4 * h;
          if (rb) {
            const f32x4 rv = *reinterpret_cast<const f32x4*>(rb + yi);
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] += lrelu(rv[r], P.res_slope);
          }
          *reinterpret_cast<f32x4*>(yb + yi) = t;
        } else if (P.S == 4) {          // two channels x two phases
#pragma unroll
          for (int c2 = 0; c2 < 2; ++c2) {
            const int co = (m >> 1) + c2;
            const float bv = P.bias ? P.bias[co] : 0.f;
            f32x2 t;
            t[0] = lrelu(o[2 * c2] + bv, P.out_slope); t[1] = lrelu(o[2 * c2 + 1] + bv, P.out_slope);
            const long long yi = (long long)co * P.LY + 4 * u + 2 * h;
            if (rb) { t[0] += lrelu(rb[yi], P.res_slope); t[1] += lrelu(rb[yi + 1], P.res_slope); }
            *reinterpret_cast<f32x2*>(yb + yi) = t;
          }
        } else {                        // S = 2: one phase per half, rows = channels
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float t = o[r] + (P.bias ? P.bias[m + r] : 0.f);
            const long long yi = (long long)(m + r) * P.LY + 2 * u + h;
            yb[yi] = lrelu(t, P.out_slope) + (rb ? lrelu(rb[yi], P.res_slope) : 0.f);
          }
        }
      }
    }
  }
}

// ---- weight image: unit (half, row tile, k-step, piece, lane) = the lane's eight reduction elements of row 32 rt + (lane & 31) -----------
struct GcPackArgs {
  const float* w; const float* scale; u32x4* wp;
  int form, CX, CY, k, S, pad, rpc_shift, nkg, KSP, nrt, nh, np;
  long long total;
};

__global__ __launch_bounds__(256) void gc_pack_kernel(const GcPackArgs P) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= P.total) return;
  const int lane = (int)(i & 63);
  long long r = i >> 6;
  const int q = (int)(r % P.np); r /= P.np;
  const int ks = (int)(r % P.KSP); r /= P.KSP;
  const int rt = (int)(r % P.nrt);
  const int h = (int)(r / P.nrt);
  const int mr = rt * 32 + (lane & 31);
  const int kg = 2 * ks + (lane >> 5);
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float t = 0.f;
    if (kg < P.nkg) {
      if (P.form != 2) {   // Conv1d weight (CY, CX, k), weight-norm scale per output channel
        const int chan = P.form == 0 ? kg >> P.rpc_shift : 2 * kg + (e >> 2);
        const int j = P.form == 0 ? 8 * (kg & ((1 << P.rpc_shift) - 1)) + e : (e & 3);
        if (j < P.k && chan < P.CX && mr < P.CY)
          t = P.w[((long long)mr * P.CX + chan) * P.k + j] * (P.scale ? P.scale[mr] : 1.f);
      } else {             // ConvTranspose1d weight (CX, CY, k), scale per INPUT channel (dim 0); rows (channel, phase of this half)
        const int chan = 4 * kg + (e >> 1);
        const int hs = P.S >> 1;
        const int co = mr / hs, p = h * hs + mr % hs;
        // y[S u + p] takes x[i] through tap j = S (u - i) + p + pad: half 0 reads (u - 1, u), half 1 (u, u + 1)
        const int du = h == 0 ? 1 - (e & 1) : -(e & 1);   // u - i
        const int j = P.S * du + p + P.pad;
        if (j >= 0 && j < P.k && chan < P.CX && co < P.CY)
          t = P.w[((long long)chan * P.CY + co) * P.k + j] * (P.scale ? P.scale[chan] : 1.f);
      }
    }
    v[e] = t;
  }
  u32x4 o;
  for (int qq = 0;; ++qq) {
    o[0] = gc_pack2(v[0], v[1]); o[1] = gc_pack2(v[2], v[3]); o[2] = gc_pack2(v[4], v[5]); o[3] = gc_pack2(v[6], v[7]);
    if (qq == q) break;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v[2 * u] -= __builtin_bit_cast(float, o[u] << 16);
      v[2 * u + 1] -= __builtin_bit_cast(float, o[u] & 0xffff0000u);
    }
  }
  P.wp[i] = o;
}

// ---- host side --------------------------------------------------------------------------------------------------------------------
static int gc_env(const char* name, int dflt) {
  const char* s = getenv(name);
  return s ? atoi(s) : dflt;
}

struct GcPlan {
  int ok, form, np;
  int CX, LX, CY, LY, NU, S, k, pad, rpc_shift, nkg, KSP, rows, nrt, nh;
  int RT, CT, nrg, ncb;
};

// dir 0: the canonical conv itself (a Conv1d's forward); dir 1: its adjoint (a ConvTranspose1d's forward).  See Canon (common.h).
static void gc_plan(const Canon& c, int dir, GcPlan* p) {
  static const int enabled = gc_env("EBEN_GC", 1);
  memset(p, 0, sizeof(*p));
  if (!enabled || c.bl || (c.np != 3 && c.np != 2) || c.g != 1 || c.d != 1 || c.xsplit_dir >= 0) return;   // EBEN_MATH_BF16X6 / EBEN_MATH_BF16X3
  p->np = c.np;
  p->S = c.s; p->k = c.k;
  if (dir == 0) {
    // EncBlock.conv (k = 2 S, stride S) and the latent convs (5 <= k <= 8, stride 1); other shapes keep the general tap-conv
    const bool enc = c.k == 2 * c.s && (c.s == 2 || c.s == 4 || c.s == 8);
    const bool lat = c.s == 1 && c.k >= 5 && c.k <= 8;
    if (!(enc || lat) || c.Cin % 16 || c.Cout % 32) return;
    if (c.reflect && (c.pl > c.Lin - 1 || c.pr > c.Lin - 1)) return;   // one reflection per side
    p->form = c.k <= 4 ? 1 : 0;
    p->CX = c.Cin; p->LX = c.Lin; p->CY = c.Cout; p->LY = c.Lout; p->NU = c.Lout;
    p->pad = c.pl;
    p->rpc_shift = c.k > 8 ? 1 : 0;
    p->nkg = p->form == 0 ? c.Cin << p->rpc_shift : c.Cin / 2;
    p->rows = c.Cout; p->nh = 1;
  } else {
    // the ConvTranspose1d: input (B, c.Cout, c.Lout), output (B, c.Cin, c.Lin), weight (c.Cout, c.Cin, k), padding c.pl
    if (c.s != 2 && c.s != 4 && c.s != 8) return;
    if (c.k != 2 * c.s || c.pl != c.s / 2 || c.Lin != c.s * c.Lout) return;
    if (c.Cout % 16 || (c.Cin * (c.s / 2)) % 32) return;
    p->form = 2;
    p->CX = c.Cout; p->LX = c.Lout; p->CY = c.Cin; p->LY = c.Lin; p->NU = c.Lout;
    p->pad = c.pl;
    p->nkg = c.Cout / 4;
    p->rows = c.Cin * (c.s / 2); p->nh = 2;
  }
  p->KSP = round_up(ceil_div(p->nkg, 2), 8);
  p->nrt = p->rows / 32;
  // tile shape: estimated cycles = rounds of 1024 waves x k-steps x column tiles x max(matrix issue of RT tiles x 6 products, the
  // sample loads + piece split of one column tile); more rows per wave amortise the split, more waves hide more latency
  static const int env_rt = gc_env("EBEN_GC_RT", 0), env_ct = gc_env("EBEN_GC_CT", 0);
  double best = 1e30;
  const int prod = p->np == 3 ? 6 : p->np == 2 ? 3 : 1;
  for (int rt : {2, 1}) {
    if (p->nrt % rt) continue;
    for (int ct : {2, 1}) {
      if (rt == 1 && ct == 2) continue;   // (one row tile, two column tiles) has no use: the split is not amortised and the samples take 128 registers
      if ((env_rt && rt != env_rt && p->nrt % env_rt == 0) || (env_ct && ct != env_ct)) continue;
      const int ncb = ceil_div(p->NU, 128 * ct);
      const long long waves = 4ll * p->nh * (p->nrt / rt) * c.B * ncb;
      const double per_wave = (double)p->KSP * ct * (32.0 * prod * rt > 240.0 ? 32.0 * prod * rt : 240.0) + 2500.0 + 600.0 * rt * ct;
      // ([MI355X] equal estimates: the shape with more, smaller waves wins by 3-10 % -- 32 / 64-channel strided layers 25.8 / 35.0 -> 24.5 / 31.9 us)
      const double t = (double)((waves + 1023) / 1024) * per_wave * (ct == 2 ? 1.05 : 1.0);
      if (t < best * 0.98) { best = t; p->RT = rt; p->CT = ct; }
    }
  }
  if (!p->RT) return;
  p->nrg = p->nrt / p->RT;
  p->ncb = ceil_div(p->NU, 128 * p->CT);
  p->ok = 1;
}

int gc_applicable(const Canon& c, int dir) {
  GcPlan p;
  gc_plan(c, dir, &p);
  return p.ok;
}

size_t gc_packed_floats(const Canon& c, int dir) {
  GcPlan p;
  gc_plan(c, dir, &p);
  return p.ok ? (size_t)p.nh * p.nrt * p.KSP * p.np * 64 * 4 : 0;
}

int gc_pack(const Canon& c, int dir, const float* w, const float* scale, float* wp, hipStream_t st) {
  GcPlan p;
  gc_plan(c, dir, &p);
  if (!p.ok) return fail(EBEN_EUNSUPPORTED, "gc_pack on a layer the direct-operand conv does not cover");
  GcPackArgs a{};
  a.w = w; a.scale = scale; a.wp = reinterpret_cast<u32x4*>(wp);
  a.form = p.form; a.CX = p.CX; a.CY = p.CY; a.k = p.k; a.S = p.S; a.pad = p.pad; a.rpc_shift = p.rpc_shift; a.nkg = p.nkg; a.KSP = p.KSP;
  a.nrt = p.nrt; a.nh = p.nh; a.np = p.np;
  a.total = (long long)p.nh * p.nrt * p.KSP * p.np * 64;
  hipLaunchKernelGGL(gc_pack_kernel, dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, st, a);
  EBEN_CHECK_LAUNCH("gc_pack_kernel");
  return EBEN_OK;
}

template <int FORM, int NP>
static void gc_launch_shape(const GcPlan& p, const GcArgs& a, unsigned grid, hipStream_t st) {
  if (p.RT == 2 && p.CT == 2) hipLaunchKernelGGL((gc_kernel<FORM, 2, 2, NP>), dim3(grid), dim3(256), 0, st, a);
  else if (p.RT == 2) hipLaunchKernelGGL((gc_kernel<FORM, 2, 1, NP>), dim3(grid), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((gc_kernel<FORM, 1, 1, NP>), dim3(grid), dim3(256), 0, st, a);
}

int gc_launch(const Canon& c, int dir, const TapIO& io, hipStream_t st) {
  GcPlan p;
  gc_plan(c, dir, &p);
  if (!p.ok) return fail(EBEN_EUNSUPPORTED, "gc_launch on a layer the direct-operand conv does not cover");
  if (io.emask || io.in_mode || io.accumulate || io.res_rows || io.fm_sums) return fail(EBEN_EUNSUPPORTED, "direct-operand conv: forward launches only");
  GcArgs a{};
  a.x = io.x; a.y = io.y; a.wp = reinterpret_cast<const u32x4*>(io.wp); a.bias = io.bias; a.res = io.res;
  a.B = c.B; a.CX = p.CX; a.LX = p.LX; a.CY = p.CY; a.LY = p.LY; a.NU = p.NU; a.S = p.S;
  a.rpc_shift = p.rpc_shift; a.off0 = -p.pad; a.KSP = p.KSP; a.nkg = p.nkg;
  a.nrt = p.nrt; a.nrg = p.nrg; a.ncb = p.ncb; a.nh = p.nh;
  a.reflect = dir == 0 ? c.reflect : 0;
  a.in_slope = io.in_slope; a.out_slope = io.out_slope; a.res_slope = io.res_slope;
  a.w_rt = (long long)p.KSP * p.np * 64; a.w_half = a.w_rt * p.nrt;
  const unsigned grid = (unsigned)(p.nh * p.nrg * c.B * p.ncb);
  a.xq = grid / 8; a.xr = grid % 8;
  if (p.np == 2) {   // hi + lo operands, three piece products (the bf16-mixed plan)
    if (p.form == 0) gc_launch_shape<0, 2>(p, a, grid, st);
    else if (p.form == 1) gc_launch_shape<1, 2>(p, a, grid, st);
    else gc_launch_shape<2, 2>(p, a, grid, st);
  } else if (p.form == 0) gc_launch_shape<0, 3>(p, a, grid, st);
  else if (p.form == 1) gc_launch_shape<1, 3>(p, a, grid, st);
  else gc_launch_shape<2, 3>(p, a, grid, st);
  EBEN_CHECK_LAUNCH("gc_kernel");
  return EBEN_OK;
}

}  // namespace eben
