// tap4_kernel.h -- tap4_kernel: the MFMA-bound bundle-layout launches of the bf16 tap-conv (MelGAN L3-L5, PQMF-band L5 / L6; forward and
// stacked input gradients) as ONE PERSISTENT 512-thread block per CU (gfx950).  Instantiated in bigtap.hip / bigtap_x3.hip.
//
// Same contraction, same packed weight image, same k-step table as tap3_kernel (tapconv3.hip):
//   y[b, g*Mg+m, t*OS+oo] = epi( sum_{c<Cg} sum_{j<J} W[j,c,m] * x(b, g*Cg+c, t*S + off0 + j*dstep) ),   k-step = one tap x 16 channels.
// What differs is how a CU is kept busy:
//   * block tile = 256 / 192 / 128 rows x 128 columns on FOUR CONSUMER WAVES (2 x 2, one per SIMD; wave tile 128 / 96 / 64 rows x 64
//     columns): 0.75-1 LDS fragment read per MFMA (tap3: 1.25), every input tile staged once per panel instead of once per 128-row
//     tile, half the weight bytes per MFMA of a 128-row tile;
//   * FOUR PRODUCER WAVES (one beside each consumer on its SIMD) issue every LDS-DMA of the block: an LDS-DMA piece costs the wave that
//     issues it 60-185 cycles of issue time, which a wave that also multiplies pays in MFMA slots (one-role blocks: 0.49 of peak,
//     split roles: 0.53);
//   * a block walks a contiguous range of tiles and treats (tile, channel chunk, k-step) as ONE stream: weight chunks of KSC k-steps go
//     through a ring of RING LDS slots, DIST = RING - 1 chunks ahead of the chunk being multiplied, across tile boundaries; the input
//     tile of the next channel chunk (or of the next tile) lands in the other input buffer meanwhile; the bias rows too.  Prologue
//     and DMA round trips are paid once per block, not once per tile;
//   * every DMA is inline asm (hipcc does not know of them), every wait on them an explicit counted s_waitcnt vmcnt(N) of the producers in
//     front of the one raw s_barrier per chunk; loads come back in order, so "all but the N youngest" is exactly "the next chunk has landed";
//   * the consumers' fragment reads run HB k-steps ahead of the MFMAs (four fragment register sets at HB = 2), one read behind each MFMA,
//     ONE counted lgkmcnt wait per k-step; the last HB k-steps of a chunk are held in registers across the barrier, so the MFMAs come
//     first behind it and the next chunk's first reads fly under them.
// What bounds it [MI355X]: the LDS-DMA path -- ~39 B / clock / CU alone, ~21 B / clock / CU under the consumers' LDS reads -- against the
// 8 KB of weights per k-step of a 256 x 128 tile: 0.53 of the dense bf16 peak at 2.0 GHz (profiles/r05_l4_waits.txt).
// The epilogue is tap3_kernel's bundle epilogue (bias through LDS, four rows per lane = one 8-byte half unit, LeakyReLU mask and
// feature-matching operands read the same way), plus the phases-as-rows form (whole 16-byte units through v_permlane32_swap).
#pragma once
#include "common.h"
#include "tap3.h"

#include <cstdlib>

namespace eben {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#ifndef EBEN_T4_DBG
#define EBEN_T4_DBG 0   // scratch-build ablations (wrong results): 1 no weight stream, 2 no input stream, 4 no barrier, 8 no MFMA, 16 no epilogue stores, 32 no mask / feature-matching loads, 32 no mask / feature-matching loads
#endif
#ifndef EBEN_T4_HB
#define EBEN_T4_HB 2    // k-steps the fragment reads run ahead of the MFMAs (single-piece launches)
#endif

static __device__ u32x4 t4_zero_unit = {0u, 0u, 0u, 0u};

__device__ __forceinline__ unsigned t4_pack(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));   // v_cvt_pk_bf16_f32 (RNE)
}

// One LDS-DMA piece: lane i copies the 16 bytes at its source address to LDS byte dst + 16 i (dst wave-uniform, through M0).
__device__ __forceinline__ void t4_dma(const void* src, unsigned dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(dst) : "memory");
}
// The weight chunks: a wave copies ONE contiguous range of the chunk, piece after piece.  The instruction's immediate offset moves both
// the source and the LDS address (piece u % 4 of a group: + 1024 u), M0 is set once per group of four pieces (+ 4096 per group), the
// source is a scalar base + one 32-bit lane offset per group: one instruction per piece, no vector arithmetic.
__device__ __forceinline__ void t4_set_m0(unsigned dst) { asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(dst) : "memory"); }
template <int OFF> __device__ __forceinline__ void t4_dma_o(unsigned voff, const void* sbase) {
  asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
// s_waitcnt vmcnt(N) for the DMAs hipcc does not know of (asm: it must not be "optimised"), then lgkmcnt(0) as a builtin hipcc's own
// bookkeeping understands: every fragment read has returned -- behind the barrier it does not wait for last chunk's reads again
template <int N> __device__ __forceinline__ void t4_wait_vm() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit count");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0) only
}
__device__ __forceinline__ void t4_barrier() { asm volatile("s_barrier" ::: "memory"); }

// v_permlane32_swap: x's lanes 32-63 <-> y's lanes 0-31
__device__ __forceinline__ void t4_swap32(unsigned& x, unsigned& y) {
  const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
  x = r[0]; y = r[1];
}

constexpr int T4_MAXP = 12;   // LDS-DMA pieces per wave and input tile piece

struct T4Tile {
  int ph, b, g, mt, t0, q0, nt, oo, kscc, tab;
  long long xrow, wsrc;
};

template <int V> struct T4Int { static constexpr int value = V; };
template <int I, int N, class F> __device__ __forceinline__ void t4_static_for(F&& f) {
  if constexpr (I < N) {
    f(T4Int<I>{});
    t4_static_for<I + 1, N>(f);
  }
}

template <int WM, int WN, int TM, int TN, int NP, int KSC, int RING, int HB>
__global__ __launch_bounds__(512) void tap4_kernel(const Tap3Args P) {
  constexpr int FM = WM * TM, BM = FM * 32, BN = WN * TN * 32, DIST = RING - 1;
  constexpr int WCHU = KSC * NP * FM * 64;   // 16-byte units per weight chunk
  constexpr int WU = WCHU / 256;             // LDS-DMA pieces per producer wave and chunk
  static_assert(WM * WN == 4, "four consumer waves");
  static_assert(WCHU % 256 == 0, "a weight chunk splits into whole pieces per producer wave");
  constexpr int NSET = HB == 1 ? 2 : 4;      // fragment register sets; k-step ks of every chunk lives in set ks % NSET
  static_assert((HB == 1 || HB == 2) && KSC % NSET == 0 && HB < KSC, "fragment reads run HB k-steps ahead of the MFMAs");
  static_assert(DIST >= 1 && DIST <= 3 && DIST * WU < 64, "ring depth");
  typedef const __attribute__((address_space(3))) u32x4* lds_cu4;
  typedef const __attribute__((address_space(3))) f32x4* lds_cf4;
  typedef const __attribute__((address_space(4))) int* ctab_t;

  extern __shared__ __attribute__((aligned(16))) u32x4 smem4[];
  const int XT = P.big_XT, PPT = P.big_PPT;
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)smem4;
  const unsigned ws_byte = lds0;                                   // RING x WCHU units
  const unsigned xs_byte = lds0 + (unsigned)(RING * WCHU) * 16u;   // 2 buffers x NP pieces x XT units
  const unsigned bs_byte = xs_byte + (unsigned)(2 * NP * XT) * 16u;   // 2 x 64 units: the bias rows of the current / the next tile

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave8 >= 4;
  const int wave = wave8 & 3;
  const int wm = wave / WN, wn = wave % WN;

  // ---- this block's tiles: a contiguous range of the launch's tile list (XCD-contiguous: neighbours share the weight panel in L2) ----
  unsigned first, count;
  {
    const unsigned bid = blockIdx.x, xcd = bid & 7u, idx = bid >> 3;
    const unsigned v = (xcd < P.xr ? xcd * (P.xq + 1) : P.xr * (P.xq + 1) + (xcd - P.xr) * P.xq) + idx;
    const unsigned nb = gridDim.x, per = (unsigned)P.big_tiles / nb, rem = (unsigned)P.big_tiles % nb;
    first = v * per + (v < rem ? v : rem);
    count = per + (v < rem ? 1u : 0u);
  }
  first = __builtin_amdgcn_readfirstlane(first);
  count = __builtin_amdgcn_readfirstlane(count);
  if (count == 0) return;
  const int NCH = P.tab_phase / KSC;             // weight chunks per tile (the same for every phase; zero k-steps behind the last real one)

  const int CgB = P.Cg >> 3;
  auto decode = [&](unsigned id, T4Tile& T) {
    int ph, tt, b, mt, g;
    {   // multiply-high divisions by the host's magic numbers (the launch refuses grids they are not exact for)
      unsigned qd = P.m_nph ? __umulhi(id, P.m_nph) : id; ph = (int)(id - qd * (unsigned)P.nph); id = qd;
      qd = P.m_ntt ? __umulhi(id, P.m_ntt) : id; tt = (int)(id - qd * (unsigned)P.ntt); id = qd;
      qd = P.m_B ? __umulhi(id, P.m_B) : id; b = (int)(id - qd * (unsigned)P.B); id = qd;
      qd = P.m_nmt ? __umulhi(id, P.m_nmt) : id; mt = (int)(id - qd * (unsigned)P.nmt); g = (int)qd;
    }
    T.ph = __builtin_amdgcn_readfirstlane(ph); T.b = __builtin_amdgcn_readfirstlane(b);
    T.g = __builtin_amdgcn_readfirstlane(g); T.mt = __builtin_amdgcn_readfirstlane(mt);
    T.t0 = __builtin_amdgcn_readfirstlane(tt) * BN;
    T.nt = P.pg[T.ph].nt; T.oo = P.pg[T.ph].oo;
    T.q0 = T.t0 * P.S + P.pg[T.ph].minoff;
    T.kscc = P.pg[T.ph].J * P.CP;
    T.tab = T.ph * P.tab_phase;
    T.xrow = ((long long)T.b * P.CBx + (long long)T.g * CgB) * P.Lx;
    T.wsrc = (long long)T.ph * P.w_phase + ((long long)T.g * P.nmt + T.mt) * P.w_tile;
  };
  T4Tile cur, nxt;
  decode(first, cur);
  nxt = cur;
  if (count > 1) decode(first + 1, nxt);

  if (producer) {
    // =================================================================================================================================
    // PRODUCER waves (4-7, one beside each consumer on its SIMD): every LDS-DMA of the block.  Iteration n: the input tile that falls
    // due, then chunk n + DIST of the weights into the slot chunk n - 1 was read from; wait until chunk n + 1 has landed -- loads come
    // back in order: all but the (DIST - 1) WU youngest pieces -- and meet the consumers at the barrier.  Nothing else is in these
    // waves' vmcnt.
    // =================================================================================================================================
    // input tile staging: the LDS image [bundle][stride phase][PLEN] is one linear run of units, moved in pieces of 64; lane constants
    // of this wave's pieces (fixed for the launch): position relative to the tile's first, row offset in the plane
    const int PPW = (PPT + 3) >> 2;
    int rel[T4_MAXP], rowoff[T4_MAXP];
#pragma unroll
    for (int k = 0; k < T4_MAXP; ++k) {
      int pi = wave + 4 * k;
      if (pi >= PPT) pi -= 4;   // a wave past the tile's end repeats its previous piece (same bytes to the same place): uniform DMA counts
      const unsigned u = (unsigned)(pi * 64 + lane);
      const unsigned bb = __umulhi(u, P.m_cstride), rr = u - bb * (unsigned)P.CSTRIDE;
      const unsigned p = P.m_plen ? __umulhi(rr, P.m_plen) : 0u, d = rr - p * (unsigned)P.PLEN;
      rel[k] = bb < (unsigned)P.CI_B ? (int)(d * (unsigned)P.S + p) : -0x40000000;
      rowoff[k] = (int)(bb < (unsigned)P.CI_B ? bb : 0u) * P.Lx;
    }
    const u32x4* zunit = &t4_zero_unit;
    asm volatile("" : "+v"(zunit));   // one register pair for the whole kernel instead of a pc-relative address per piece
    auto x_tile = [&](const T4Tile& T, int cc, int buf, int par) {   // one input tile (+ the tile's bias rows with its first channel chunk)
      if (EBEN_T4_DBG & 2) return;
      const long long base = T.xrow + (long long)cc * P.CI_B * P.Lx;
      t4_static_for<0, T4_MAXP>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if (k < PPW) {
          int pi = wave + 4 * k;
          if (pi >= PPT) pi -= 4;
          const int pos = T.q0 + rel[k];
          const bool ok = (unsigned)pos < (unsigned)P.Lx;
          const long long idx = base + rowoff[k] + pos;
          const unsigned dst = xs_byte + (unsigned)((buf * NP * XT + pi * 64) * 16);
          t4_dma(ok ? (const void*)(P.xh + idx) : (const void*)zunit, dst);
          if constexpr (NP > 1) t4_dma(ok ? (const void*)(P.xl + idx) : (const void*)zunit, dst + (unsigned)(XT * 16));
        }
      });
      if (cc == 0) {   // BM floats = BM / 4 units; the lanes behind them re-read the first: every producer wave, the same bytes
        const float* bsrc = P.bias ? P.bias + (long long)T.g * P.Mg + T.mt * BM + 4 * (4 * lane < BM ? lane : 0) : reinterpret_cast<const float*>(zunit);
        t4_dma(bsrc, bs_byte + (unsigned)(par * 1024));
      }
    };
    // The weight chunks: a wave copies ONE contiguous quarter of the chunk, piece after piece; the instruction's immediate offset moves
    // both the source and the LDS address, M0 is set once per group of four pieces
    constexpr int WG = (WU + 3) / 4;
    unsigned wvoff[WG];
#pragma unroll
    for (int gq = 0; gq < WG; ++gq) wvoff[gq] = (unsigned)((wave * WU + gq * 4) * 64 + lane) * 16u;
    int p_ch = 0;
    const u32x4* p_src = P.wp + cur.wsrc;
    unsigned p_dst = ws_byte + (unsigned)(wave * WU * 1024);
    const unsigned p_dst_end = ws_byte + (unsigned)(RING * WCHU * 16 + wave * WU * 1024);
    // the producer's chunk goes out and the producer moves on; when it wraps it enters `nxt` (NCH > RING: by then `nxt` is the tile behind
    // the producer's).  Behind the end of the stream it keeps walking over memory that exists (`nxt` stays the last decoded tile): those
    // pieces land in slots nobody reads any more and keep the counts of the waits the same.
    auto w_chunk = [&]() {
      if ((EBEN_T4_DBG & 1) == 0) {
        t4_static_for<0, WU>([&](auto uc) {
          constexpr int u = decltype(uc)::value;
          if constexpr (u % 4 == 0) t4_set_m0(p_dst + (unsigned)(u * 1024));
          t4_dma_o<(u % 4) * 1024>(wvoff[u / 4], p_src);
        });
      }
      p_src += WCHU; p_dst += (unsigned)(WCHU * 16);
      if (p_dst == p_dst_end) p_dst -= (unsigned)(RING * WCHU * 16);
      if (++p_ch == NCH) { p_ch = 0; p_src = P.wp + nxt.wsrc; }
    };
    // input buffer of (tile i, channel chunk cc): cc & 1 -- the plan keeps an even number of chunks per tile, which is what the k-step
    // table says
    x_tile(cur, 0, 0, 0);
    t4_static_for<0, DIST>([&](auto) { w_chunk(); });
    t4_wait_vm<(DIST - 1) * WU>();               // chunk 0 and the first input tile (asked for first) have landed
    t4_barrier();
    for (int tile_i = 0; tile_i < (int)count; ++tile_i) {
      if (tile_i > 0) {
        cur = nxt;
        if ((unsigned)(tile_i + 1) < count) decode(first + (unsigned)tile_i + 1u, nxt);
      }
      int in_cc = 1, in_ch = 1;   // the next input tile to ask for (in_cc == ncc: the next tile's first) and the chunk at whose head
      for (int ch = 0; ch < NCH; ++ch) {
        if (ch == in_ch) {
          // tile cc + 1 goes into the buffer tile cc - 1 was read from, whose last k-step lies in the chunk BEFORE this one at the latest;
          // it is needed DIST + 1 chunks later at the earliest (the plan keeps channel chunks that long), i.e. it is older than the
          // weight chunk whose landing the consumers are let go on
          if (in_cc < P.ncc) x_tile(cur, in_cc, in_cc & 1, 0);
          else if ((unsigned)(tile_i + 1) < count) x_tile(nxt, 0, 0, (tile_i + 1) & 1);
          ++in_cc;
          in_ch = in_cc <= P.ncc ? ((in_cc - 1) * cur.kscc) / KSC + 1 : 0x7fffffff;
        }
        w_chunk();
        t4_wait_vm<(DIST - 1) * WU>();
        t4_barrier();
      }
    }
    t4_wait_vm<0>();   // no LDS-DMA of this block may land after it has left the CU
    return;
  }

  // ===================================================================================================================================
  // CONSUMER waves (0-3): fragments out of LDS, MFMAs, epilogue.  No DMA and no vmcnt wait on one: what a DMA piece costs the wave that
  // issues it (60-185 cycles of issue time, MI355X_MICROARCH.md) is paid by the producer wave on the same SIMD while this one multiplies.
  // ===================================================================================================================================
#ifdef EBEN_T4_PRIO
  __builtin_amdgcn_s_setprio(EBEN_T4_PRIO);   // scratch knob: the consumers above the producers in the SIMD's arbitration
#endif
  f32x16 acc[TM][TN];
  struct Frag { u32x4 a[NP][TM], b[NP][TN]; };
  Frag FS[NSET];
  constexpr int NPROD = NP == 1 ? 1 : 3;
  constexpr int MPK = TM * TN * NPROD;   // MFMAs per k-step
  constexpr int NRD = NP * (TM + TN);    // fragment reads per k-step
  const ctab_t tab = (ctab_t)P.tab;
  int te[KSC], tn[KSC];   // te: BYTE offsets of the current chunk's B fragments; tn: table entries (units) of the next chunk
  int t_ch = 0, t_idx = cur.tab;                 // table entries asked for next: chunk t_ch of its tile
  auto t_load = [&](int (&t)[KSC]) {
#pragma unroll
    for (int ks = 0; ks < KSC; ++ks) t[ks] = tab[t_idx + ks];
    t_idx += KSC;
    if (++t_ch == NCH) { t_ch = 0; t_idx = nxt.tab; }
  };
  // The MFMAs of one k-step, with a hook behind each.  The next chunk's table entries are asked for behind the second MFMA of the last
  // k-step in front of the barrier: no LDS wait follows until the barrier's lgkmcnt(0) (a scalar load in flight turns every counted LDS
  // wait into lgkmcnt(0): SMEM returns out of order).
  auto mma = [&](const Frag& F, auto lastc, auto&& hook) {
    constexpr bool LAST_A = decltype(lastc)::value != 0;
    t4_static_for<0, MPK>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      constexpr int prod = m / (TM * TN), i = (m / TN) % TM, f = m % TN;
      // piece products, smallest first: NP = 2: (hi, lo), (lo, hi), (hi, hi) -- lo x lo (~2^-18) dropped
      constexpr int qw = NP == 1 ? 0 : (prod == 1 ? 1 : 0), qx = NP == 1 ? 0 : (prod == 0 ? 1 : 0);
      if (EBEN_T4_DBG & 8) acc[i][f][0] += __builtin_bit_cast(float, F.a[qw][i][0] ^ F.b[qx][f][1]);
      else acc[i][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, F.a[qw][i]), __builtin_bit_cast(bf16x8, F.b[qx][f]), acc[i][f], 0, 0, 0);
      hook(mc);
      if constexpr (LAST_A && m == 1) {
        __builtin_amdgcn_sched_barrier(0);
        t_load(tn);
        __builtin_amdgcn_sched_barrier(0);
      }
    });
  };
  auto no_hook = [](auto) {};

  const unsigned xlane = xs_byte + (unsigned)(((lane >> 5) * P.CSTRIDE + wn * TN * 32 + (lane & 31)) * 16);
  const unsigned wlane = ws_byte + (unsigned)((wm * TM * 64 + lane) * 16);
  int cslot = 0;
  auto rd_one = [&](int ks, Frag& F, auto rc) {   // read r of a k-step's NRD: the B fragments first
    constexpr int r = decltype(rc)::value;
    if constexpr (r < NP * TN) {
      constexpr int q = r / TN, f = r % TN;
      F.b[q][f] = ((lds_cu4)(size_t)(xlane + (unsigned)te[ks] + (unsigned)(q ? XT * 16 : 0)))[f * 32];
    } else {
      constexpr int q = (r - NP * TN) / TM, i = (r - NP * TN) % TM;
      F.a[q][i] = ((lds_cu4)(size_t)(wlane + (unsigned)(cslot * WCHU) * 16u))[((ks * NP + q) * FM + i) * 64];
    }
  };
  auto rd = [&](int ks, Frag& F) { t4_static_for<0, NRD>([&](auto rc) { rd_one(ks, F, rc); }); };
  // The fragments of k-step ks + HB are asked for during the MFMAs of ks (one read behind each MFMA where a k-step has enough of
  // them), ONE wait per k-step where hipcc would place one in front of every MFMA that uses a new fragment.
  static_assert(HB * NRD < 16, "lgkmcnt is a 4-bit count");
  constexpr bool IL = HB >= 2 && MPK >= NRD + 2;   // reads one by one behind the MFMAs

  // ---- epilogue of one tile (tap3_kernel's bundle epilogue): row quad r4 of accumulator tile (i, f) = channels m4 .. m4+3 of the group,
  // m4 = m0 + 32 i + 8 r4 + 4 (lane >> 5): half (lane >> 5) of bundle (m0 >> 3) + 4 i + r4 at position t -- one 8-byte piece per lane ----
  float fk1 = 0.f, fk2 = 0.f;
  if (P.fm_sums != nullptr && P.res_rows > 0) {
    typedef const __attribute__((address_space(4))) float* cf_t;
    const float s1 = ((cf_t)P.fm_sums)[0], s2 = ((cf_t)P.fm_sums)[1];
    fk1 = P.fm_gs / s2; fk2 = P.fm_gs * s1 / (s2 * s2);
  }
  auto epilogue = [&](const T4Tile& T, int par) {
    // (the lane id passes through an empty asm: hipcc otherwise hoists the epilogue's per-lane offsets and predicates out of the tile
    // loop, where they live through the reduction beside 224 accumulator / fragment registers -- and spill)
    int lane = tid & 63;
    asm volatile("" : "+v"(lane));
    const int hb = lane >> 5;
    const int b = T.b;
    const int es = P.em_seg > 0 ? (int)(b >= P.em_seg) + (int)(b >= 2 * P.em_seg) + (int)(b >= 3 * P.em_seg) : 0;
    const int eb = P.em_seg > 0 ? P.em_map[es] * P.em_seg + (b - es * P.em_seg) : b;
    const bool fmr = P.fm_sums != nullptr && P.res_rows > 0 && b < P.res_rows;
    const bool masked = P.eh != nullptr && !(EBEN_T4_DBG & 32);
    // feature-matching rows with a code plane (bl_edge.hip, bl_fm_code: one byte per element at half the hi plane's byte offsets): they
    // read the mask and the codes -- one load batch per 128 x 32 columns -- instead of four operands per value in a batch per 32-row tile
    const bool fmc = fmr && masked && P.ec != nullptr;
    const long long Lrow = (long long)P.Ly * 16;                                                    // bytes per bundle row
    const int m0w = T.mt * BM + wm * TM * 32;
    const long long tile0 = ((long long)((T.g * P.Mg + m0w) >> 3)) * Lrow;                         // this wave's first bundle row
    const char* ehb = reinterpret_cast<const char*>(P.eh) + (long long)eb * P.CBy * Lrow + tile0;
    const char* elb = reinterpret_cast<const char*>(P.el) + (long long)eb * P.CBy * Lrow + tile0;
    const char* rhb = reinterpret_cast<const char*>(P.eh) + (long long)(b + P.bl_ref_off) * P.CBy * Lrow + tile0;
    const char* rlb = reinterpret_cast<const char*>(P.el) + (long long)(b + P.bl_ref_off) * P.CBy * Lrow + tile0;
    char* yhb = reinterpret_cast<char*>(P.yh) + (long long)b * P.CBy * Lrow + tile0;
    char* ylb = reinterpret_cast<char*>(P.yl) + (long long)b * P.CBy * Lrow + tile0;
    const lds_cf4 bsp = (lds_cf4)(size_t)(bs_byte + (unsigned)(par * 1024 + (wm * TM * 32 + 4 * hb) * 4));
    auto unpack = [](uint2 w, float (&f)[4]) {
      f[0] = __builtin_bit_cast(float, w.x << 16); f[1] = __builtin_bit_cast(float, w.x & 0xffff0000u);
      f[2] = __builtin_bit_cast(float, w.y << 16); f[3] = __builtin_bit_cast(float, w.y & 0xffff0000u);
    };
    if (P.pr_S > 0) {
      // ---- phases as rows, rows ordered (channel bundle, phase, channel in bundle), stride 4 (eben_bl_conv1d_bwd_dx_pr): accumulator tile
      // i of this wave IS bundle cb0 + i of the group at the four phases of the tile's columns -- row quad r4 = phase r4 -- i.e. the
      // 64 contiguous bytes of positions 4 t .. 4 t + 3.  The MFMA layout gives lane (t, half h) the half units of all four phases; one
      // v_permlane32_swap per dword turns that into whole units: lane (t, 0) positions 4 t, 4 t + 1, lane (t, 1) positions 4 t + 2,
      // 4 t + 3 -- 32 contiguous bytes per lane, two 16-byte accesses where the phase-scatter form makes four of 8 bytes 64 bytes apart
      // (mask and feature-matching operands come in the same way and are swapped back).
      const long long LrowP = (long long)P.pr_Ly * 16;
      const int cb0 = (T.mt * BM + wm * TM * 32) >> 5;
      const long long tileP = (long long)(T.g * P.pr_cbg + cb0) * LrowP;
      const char* ehp = reinterpret_cast<const char*>(P.eh) + (long long)eb * P.CBy * LrowP + tileP;
      const char* elp = reinterpret_cast<const char*>(P.el) + (long long)eb * P.CBy * LrowP + tileP;
      const char* rhp = reinterpret_cast<const char*>(P.eh) + (long long)(b + P.bl_ref_off) * P.CBy * LrowP + tileP;
      const char* rlp = reinterpret_cast<const char*>(P.el) + (long long)(b + P.bl_ref_off) * P.CBy * LrowP + tileP;
      char* yhp = reinterpret_cast<char*>(P.yh) + (long long)b * P.CBy * LrowP + tileP;
      char* ylp = reinterpret_cast<char*>(P.yl) + (long long)b * P.CBy * LrowP + tileP;
      unsigned poff[TN];
      bool lv[TN][2];
#pragma unroll
      for (int f = 0; f < TN; ++f) {
        const int t = T.t0 + (wn * TN + f) * 32 + (lane & 31);
        const int pos0 = 4 * t + 2 * hb;
        lv[f][0] = t < T.nt && pos0 < P.pr_Ly;
        lv[f][1] = t < T.nt && pos0 + 1 < P.pr_Ly;
        poff[f] = (unsigned)(lv[f][0] ? pos0 : 0) * 16u;
      }
      // memory layout (two whole units per lane) <-> MFMA layout (four half units per lane)
      auto to_halves = [&](const u32x4 (&U)[2], uint2 (&H)[4]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          unsigned x0 = U[j][0], x1 = U[j][1], y0 = U[j][2], y1 = U[j][3];
          t4_swap32(x0, y0); t4_swap32(x1, y1);
          H[j].x = x0; H[j].y = x1; H[2 + j].x = y0; H[2 + j].y = y1;
        }
      };
      auto to_units = [&](const uint2 (&H)[4], u32x4 (&U)[2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          unsigned x0 = H[j].x, x1 = H[j].y, y0 = H[2 + j].x, y1 = H[2 + j].y;
          t4_swap32(x0, y0); t4_swap32(x1, y1);
          U[j] = u32x4{x0, x1, y0, y1};
        }
      };
      auto ldu = [&](const char* base, int i, int f, u32x4 (&U)[2]) {
        const char* q = base + (long long)i * LrowP + poff[f];
        U[0] = *reinterpret_cast<const u32x4*>(q);
        U[1] = *reinterpret_cast<const u32x4*>(q + (lv[f][1] ? 16 : 0));
      };
      auto stu = [&](int i, int f, const uint2 (&H)[4], char* base) {
        u32x4 U[2];
        to_units(H, U);
        char* q = base + (long long)i * LrowP + poff[f];
        if ((EBEN_T4_DBG & 16) && U[0][0] != 0x12345u) return;
        if (lv[f][0]) *reinterpret_cast<u32x4*>(q) = U[0];
        if (lv[f][1]) *reinterpret_cast<u32x4*>(q + 16) = U[1];
      };
      auto quad = [&](int i, int f, int r4, bool has_a, bool has_fm, const uint2 (&a_h)[4], const uint2 (&a_l)[4], const uint2 (&r_h)[4], const uint2 (&r_l)[4], uint2& oh_, uint2& ol_,
                      bool has_code = false, unsigned code = 0u) {
        float v[4], a0[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][f][4 * r4 + e];
        if (has_a) {
          unpack(a_h[r4], a0);
          if (has_code) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const unsigned c = code >> (8 * e);
              v[e] += fk1 * (float)((int)(c & 3u) - 1) - fk2 * (float)((int)((c >> 2) & 3u) - 1);
            }
          } else if (has_fm) {
            float a1[4], r0[4], r1[4];
            unpack(a_l[r4], a1); unpack(r_h[r4], r0); unpack(r_l[r4], r1);
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
        oh_.x = t4_pack(v[0], v[1]); oh_.y = t4_pack(v[2], v[3]);
        float hf[4];
        unpack(oh_, hf);
        ol_.x = t4_pack(v[0] - hf[0], v[1] - hf[1]); ol_.y = t4_pack(v[2] - hf[2], v[3] - hf[3]);
      };
      if (masked && !fmr) {
        u32x4 AU[TN][TM][2];
#pragma unroll
        for (int f = 0; f < TN; ++f)
#pragma unroll
          for (int i = 0; i < TM; ++i) ldu(ehp, i, f, AU[f][i]);
#pragma unroll
        for (int f = 0; f < TN; ++f)
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            uint2 AH[4], OH[4], OL[4];
            to_halves(AU[f][i], AH);
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) quad(i, f, r4, true, false, AH, AH, AH, AH, OH[r4], OL[r4]);
            stu(i, f, OH, yhp);
            if (P.yl) stu(i, f, OL, ylp);
          }
      } else if (fmc) {
        const char* ecp = reinterpret_cast<const char*>(P.ec) + (((long long)eb * P.CBy * LrowP + tileP) >> 1);
#pragma unroll
        for (int f = 0; f < TN; ++f) {
          u32x4 AU[TM][2];
          uint2 CU[TM][2];
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            ldu(ehp, i, f, AU[i]);
            const char* q = ecp + (((long long)i * LrowP + poff[f]) >> 1);
            CU[i][0] = *reinterpret_cast<const uint2*>(q);
            CU[i][1] = *reinterpret_cast<const uint2*>(q + (lv[f][1] ? 8 : 0));
          }
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            uint2 AH[4], OH[4], OL[4];
            to_halves(AU[i], AH);
            // two whole code units (positions pos0, pos0 + 1) -> the four half units of this lane's row quads
            unsigned x0 = CU[i][0].x, y0 = CU[i][0].y, x1 = CU[i][1].x, y1 = CU[i][1].y;
            t4_swap32(x0, y0); t4_swap32(x1, y1);
            const unsigned CW[4] = {x0, x1, y0, y1};
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) quad(i, f, r4, true, false, AH, AH, AH, AH, OH[r4], OL[r4], true, CW[r4]);
            stu(i, f, OH, yhp);
            if (P.yl) stu(i, f, OL, ylp);
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int f = 0; f < TN; ++f) {
            uint2 AH[4], AL[4], RH[4], RL[4], OH[4], OL[4];
            if (masked) {
              u32x4 AU[2], LU[2], RHU[2], RLU[2];
              ldu(ehp, i, f, AU); ldu(elp, i, f, LU); ldu(rhp, i, f, RHU); ldu(rlp, i, f, RLU);
              to_halves(AU, AH); to_halves(LU, AL); to_halves(RHU, RH); to_halves(RLU, RL);
            }
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) quad(i, f, r4, masked, masked, AH, AL, RH, RL, OH[r4], OL[r4]);
            stu(i, f, OH, yhp);
            if (P.yl) stu(i, f, OL, ylp);
          }
      }
      return;
    }
    // Loads before stores: with loads and stores both in flight hipcc waits vmcnt(0) for any load result (the two return out of order),
    // i.e. for every store issued so far -- written group by group ((loads, arithmetic, stores) x 8) a tile's epilogue drained its
    // stores eight times (MelGAN L4 input gradients 0.40 -> 0.52 ms).  Forward tiles have no loads: value by value.  Masked tiles: every
    // mask unit of the tile first (64 registers), then arithmetic + stores.  Feature-matching tiles (a quarter of the stacked rows) carry
    // four operands per value: one batch per 32-row tile -- the registers of two waves per SIMD do not hold more next to the accumulators.
    unsigned loffs[TN];
    bool lives[TN];
#pragma unroll
    for (int f = 0; f < TN; ++f) {
      const int t = T.t0 + (wn * TN + f) * 32 + (lane & 31);
      lives[f] = t < T.nt;
      loffs[f] = (((unsigned)(lives[f] ? t : 0) * (unsigned)P.OS + (unsigned)T.oo) * 2u + (unsigned)hb) * 8u;   // bytes inside a bundle row
    }
    auto ld2 = [&](const char* base, long long row, int f) { return *reinterpret_cast<const uint2*>(base + row + loffs[f]); };
    auto finish = [&](int f, int i, int r4, float (&v)[4]) {   // pack + store one row quad
      const long long row = (long long)(4 * i + r4) * Lrow;
      uint2 h;
      h.x = t4_pack(v[0], v[1]); h.y = t4_pack(v[2], v[3]);
      if (EBEN_T4_DBG & 16) { if (h.x == 0x12345u && lives[f]) *reinterpret_cast<uint2*>(yhb + row + loffs[f]) = h; return; }
      if (lives[f]) *reinterpret_cast<uint2*>(yhb + row + loffs[f]) = h;
      if (P.yl) {
        float hf[4];
        unpack(h, hf);
        uint2 l;
        l.x = t4_pack(v[0] - hf[0], v[1] - hf[1]); l.y = t4_pack(v[2] - hf[2], v[3] - hf[3]);
        if (lives[f]) *reinterpret_cast<uint2*>(ylb + row + loffs[f]) = l;
      }
    };
    auto biased = [&](int f, int i, int r4, float (&v)[4]) {
      const f32x4 bz = bsp[(i * 32 + 8 * r4) / 4];
      v[0] = acc[i][f][4 * r4 + 0] + bz[0]; v[1] = acc[i][f][4 * r4 + 1] + bz[1];
      v[2] = acc[i][f][4 * r4 + 2] + bz[2]; v[3] = acc[i][f][4 * r4 + 3] + bz[3];
    };
    if (!masked) {
#pragma unroll
      for (int f = 0; f < TN; ++f)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            float v[4];
            biased(f, i, r4, v);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = lrelu(v[e], P.out_slope);
            finish(f, i, r4, v);
          }
    } else if (!fmr) {
      uint2 ah[TN][TM][4];
#pragma unroll
      for (int f = 0; f < TN; ++f)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) ah[f][i][r4] = ld2(ehb, (long long)(4 * i + r4) * Lrow, f);
#pragma unroll
      for (int f = 0; f < TN; ++f)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            float v[4], a0[4];
            biased(f, i, r4, v);
            unpack(ah[f][i][r4], a0);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= dlrelu(a0[e], P.emask_slope);
            finish(f, i, r4, v);
          }
    } else if (fmc) {
      const char* ecb = reinterpret_cast<const char*>(P.ec) + (((long long)eb * P.CBy * Lrow + tile0) >> 1);
#pragma unroll
      for (int f = 0; f < TN; ++f) {
        uint2 ah[TM][4];
        unsigned cw[TM][4];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const long long row = (long long)(4 * i + r4) * Lrow;
            ah[i][r4] = ld2(ehb, row, f);
            cw[i][r4] = *reinterpret_cast<const unsigned*>(ecb + ((row + (long long)loffs[f]) >> 1));
          }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            float v[4], a0[4];
            biased(f, i, r4, v);
            unpack(ah[i][r4], a0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const unsigned c = cw[i][r4] >> (8 * e);
              v[e] += fk1 * (float)((int)(c & 3u) - 1) - fk2 * (float)((int)((c >> 2) & 3u) - 1);
              v[e] *= dlrelu(a0[e], P.emask_slope);
            }
            finish(f, i, r4, v);
          }
      }
    } else {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        uint2 ah[TN][4], al[TN][4], rh[TN][4], rl[TN][4];
#pragma unroll
        for (int f = 0; f < TN; ++f)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const long long row = (long long)(4 * i + r4) * Lrow;
            ah[f][r4] = ld2(ehb, row, f); al[f][r4] = ld2(elb, row, f); rh[f][r4] = ld2(rhb, row, f); rl[f][r4] = ld2(rlb, row, f);
          }
#pragma unroll
        for (int f = 0; f < TN; ++f)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            float v[4], a0[4], a1[4], r0[4], r1[4];
            biased(f, i, r4, v);
            unpack(ah[f][r4], a0); unpack(al[f][r4], a1); unpack(rh[f][r4], r0); unpack(rl[f][r4], r1);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float av = a0[e] + a1[e], dv = av - (r0[e] + r1[e]);
              v[e] += fk1 * (float)((dv > 0.f) - (dv < 0.f)) - fk2 * (float)((av > 0.f) - (av < 0.f));
              v[e] *= dlrelu(a0[e], P.emask_slope);
            }
            finish(f, i, r4, v);
          }
      }
    }
  };

  // ---- the stream: chunk n is multiplied in iteration n:   A: k-steps 0 .. KSC-HB-1   |   lgkmcnt(0), barrier (producers: chunk n + 1 has
  // landed; consumers: chunk n's slot may be refilled)   |   B: the last HB k-steps -- their fragments are in registers -- with the first
  // fragment reads of chunk n + 1 behind their MFMAs ----
  t_load(te);                                    // chunk 0; chunk 1's entries are asked for in the first A
#pragma unroll
  for (int ks = 0; ks < KSC; ++ks) te[ks] <<= 4;
  t4_barrier();
  t4_static_for<0, HB>([&](auto jc) { rd(decltype(jc)::value, FS[decltype(jc)::value % NSET]); });

  for (int tile_i = 0; tile_i < (int)count; ++tile_i) {
    if (tile_i > 0) {
      cur = nxt;
      if ((unsigned)(tile_i + 1) < count) decode(first + (unsigned)tile_i + 1u, nxt);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int f = 0; f < TN; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][f][r] = 0.f;
    for (int ch = 0; ch < NCH; ++ch) {
      // ---- A ----
      t4_static_for<0, KSC - HB>([&](auto ksc_) {
        constexpr int ks = decltype(ksc_)::value;
        constexpr int last = ks == KSC - HB - 1;
        if constexpr (IL) {
          // at the head of a k-step everything but the reads issued during the previous one has returned
          __builtin_amdgcn_s_waitcnt(0xc07f | (NRD << 8));
          __builtin_amdgcn_sched_barrier(0);
          mma(FS[ks % NSET], T4Int<last>{}, [&](auto mc) {
            if constexpr (decltype(mc)::value < NRD) {
              __builtin_amdgcn_sched_barrier(0);
              rd_one(ks + HB, FS[(ks + HB) % NSET], mc);
              __builtin_amdgcn_sched_barrier(0);
            }
          });
        } else {
          rd(ks + HB, FS[(ks + HB) % NSET]);
          __builtin_amdgcn_s_waitcnt(0xc07f | ((HB * NRD) << 8));   // all but the HB k-steps just asked for
          __builtin_amdgcn_sched_barrier(0);
          mma(FS[ks % NSET], T4Int<last>{}, no_hook);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): every fragment read of this chunk has returned, the next table entries are there
      if ((EBEN_T4_DBG & 4) == 0) t4_barrier();
      // ---- B ----  the MFMAs come FIRST behind the barrier (they need neither LDS nor the table); table entries, ring slot and the
      // first fragment reads of chunk n + 1 behind them
      t4_static_for<0, HB>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        mma(FS[(KSC - HB + j) % NSET], T4Int<0>{}, [&](auto mc) {
          constexpr int m = decltype(mc)::value;
          if constexpr (j == 0 && m == 0) {
            __builtin_amdgcn_sched_barrier(0);
            // (the entries pass through an empty asm: left to itself hipcc shifts them right behind the scalar load, i.e. waits
            // lgkmcnt(0) -- for the load AND the fragments just asked for -- in the middle of the last k-step in front of the barrier)
#pragma unroll
            for (int ks = 0; ks < KSC; ++ks) asm volatile("" : "+s"(tn[ks]));
#pragma unroll
            for (int ks = 0; ks < KSC; ++ks) te[ks] = tn[ks] << 4;
            cslot = cslot + 1 == RING ? 0 : cslot + 1;
            __builtin_amdgcn_sched_barrier(0);
          }
          if constexpr (IL) {
            constexpr int r0 = j == 0 ? 1 : 0;
            if constexpr (m >= r0 && m - r0 < NRD) {
              __builtin_amdgcn_sched_barrier(0);
              rd_one(j, FS[j % NSET], T4Int<m - r0>{});   // chunk n + 1 (behind the end of the stream: whatever the slot holds, never multiplied)
              __builtin_amdgcn_sched_barrier(0);
            }
          } else if constexpr (m == 1) {
            __builtin_amdgcn_sched_barrier(0);
            rd(j, FS[j % NSET]);
            __builtin_amdgcn_sched_barrier(0);
          }
        });
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    epilogue(cur, tile_i & 1);
    // the next tile's first fragments once more: read in the last B they would have to live through the epilogue (48 registers the
    // epilogue needs); their latency is paid once per tile
    t4_static_for<0, HB>([&](auto jc) { rd(decltype(jc)::value, FS[decltype(jc)::value % NSET]); });
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
template <int WM, int WN, int TM, int TN, int NP, int KSC, int RING, int HB>
static int launch4(const Tap3Plan& p, const Tap3Args& a, hipStream_t st) {
  static LdsAttrOnce attr_once;
  auto kern = tap4_kernel<WM, WN, TM, TN, NP, KSC, RING, HB>;
  {
    const hipError_t e = lds_attr_once(attr_once, reinterpret_cast<const void*>(kern));
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(tap4)");
  }
  static const int force = getenv("EBEN_BIG_BLOCKS") ? atoi(getenv("EBEN_BIG_BLOCKS")) : 0;
  const int cus = force > 0 ? force : device_cus();   // of the current device: one persistent block per CU
  Tap3Args b = a;
  const int nb = a.big_tiles < cus ? a.big_tiles : cus;
  b.xq = (unsigned)(nb / 8); b.xr = (unsigned)(nb % 8);
  hipLaunchKernelGGL(kern, dim3(nb), dim3(512), p.lds_bytes, st, b);
  EBEN_CHECK_LAUNCH("tap4_kernel");
  return EBEN_OK;
}

}  // namespace eben
