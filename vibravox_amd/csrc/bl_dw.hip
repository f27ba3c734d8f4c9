// bl_dw.hip -- weight (+ bias) gradient of a Conv1d layer with BOTH operands at rest in the bf16 bundle layout (gfx950).
//
//   dw[co, c, j] = sum_{b, t} dy[b, co, t] * x[b, c, t*S + j*d - pad],      dbias[co] = sum_{b, t} dy[b, co, t]
//
// GEMM per group on v_mfma_f32_32x32x16_bf16: M = output channels, N = (channel bundle, tap, channel in bundle) columns + one bias
// column bundle, K = (batch item, time) -- the reduction runs ALONG TIME, 16 steps per MFMA.  An MFMA operand wants 8 consecutive k
// per lane; memory holds 8 consecutive CHANNELS per 16-byte unit ([batch][channels / 8][length][8]), i.e. the transpose.  gfx950
// transposes on the way out of LDS: ds_read_b64_tr_b16 hands lane l the k-th elements of four 8-byte rows supplied by four other
// lanes of its 16-lane group (mapping checked on the device, tools/ubench/probe_lds.hip), so that
//   * both operand tiles are plain COPIES of units: A rows = 64 consecutive time steps of one bundle of dy (1 KB), X rows = the
//     stride phases of one bundle of x around the chunk (de-interleaved by the source address: consecutive time steps of a tap are
//     consecutive units of row (bundle, (j d - pad) mod S)), moved by global_load_lds_dwordx4 -- no VGPR round trip, no conversion, no
//     packing pre-pass (conv_dw3.hip: dw3_pack_a + fp32 -> bf16 staging of X); lanes outside the row (the end of dy in time, the zero
//     padding of x in position) copy a zero unit instead;
//   * a fragment is two transposing reads (k = 0..3 and 4..7 of the lane's eight) at per-lane base addresses fixed for the whole
//     launch + immediates; row strides = 4 mod 16 units keep the 32 lanes of a half-wave on distinct banks.
// Block = 4 waves as 2 x 2, (64 FM) x (64 FN) outputs; K chunk = one batch item x 64 time steps, double-buffered: the next chunk's
// tiles are in flight while the current one is multiplied, one barrier per chunk.  Split-K over blocks into private fp32 slabs
// (fixed-order reduction by eben_wn_bwd_multi).  Columns are ordered (bundle, tap, channel in bundle), which makes the slab stores
// contiguous; EbenWnBwdItem.col_perm_k tells the reduction.  Layers whose groups do not start on bundles (4 / 6 / 12 channels per
// group) run as ONE dense contraction over all channels and store the block-diagonal part in the standard order.
#include "common.h"

#include <cstdlib>

namespace eben {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifndef EBEN_BLDW_DBG
#define EBEN_BLDW_DBG 0   // scratch-build ablations (wrong results): 1 no A-tile DMA, 2 no X-row DMA, 4 no MFMA, 8 no slab stores
#endif
#ifndef EBEN_BLDW_PIPE_FENCE
#define EBEN_BLDW_PIPE_FENCE 1
#endif
#ifdef EBEN_BLDW_STAMP
// scratch build: wave 0 of every block sums the cycles (s_memtime) its chunks spend waiting for their pieces, at the barrier, issuing the
// next chunk's pieces and in the MFMA loop: row = block, columns = the four sums + chunks + total
__device__ unsigned long long bldw_stamp[8192 * 8];
#define BLDW_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#endif
constexpr int BLDW_BKT = 64;    // time steps per K chunk
constexpr int BLDW_TS = 68;     // A row stride in units (64 + 4: = 4 mod 16)
constexpr int BLDW_RS_MAX = 132;   // largest X row stride in units (two whole 64-unit LDS-DMA pieces + 4)
// X row stride of a layer: the units a chunk reads (64 time steps + the taps' reach), rounded up to 4 mod 16 -- the second LDS-DMA piece
// of a row is issued for its first xneed - 64 lanes only, so a row no longer occupies two whole pieces ([MI355X] PQMF-band layers: 2.1 ->
// 1.1 KB per row, 3 -> 4-5 blocks per CU)
static int bldw_row_stride(int xneed) { int rs = xneed <= 4 ? 4 : xneed; while ((rs & 15) != 4) ++rs; return rs > BLDW_RS_MAX ? BLDW_RS_MAX : rs; }

struct BlDwArgs {
  const u32x4* ah; const u32x4* xh;
  float* slabs;
  int B, G, Mg, Cg, MgB, CgB, CBa, CBx, La, Lx;
  int S, d, k, pad, amin;
  int NQW;                       // weight column bundles (CgB * k); bundle NQW is the bias column
  int has_bias, nnt, nmt, nsplit, nct, nchunks, XR, xneed;   // xneed: units of an X row a chunk reads (64 time steps + the taps' reach)
  int dense, c_in_g, c_out_g, row_stride;
  int RS;                        // X row stride in LDS units (bldw_row_stride)
  int xw;                        // XC: units of an X row a chunk reads (xneed time steps x stride)
  long long slab_stride;
};

template <unsigned STEP4 = 64u>
__device__ __forceinline__ bf16x8 bl_tr_frag(unsigned addr) {
  typedef __attribute__((address_space(3))) s16x4* lds4_t;
  const s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(size_t)(addr));
  const s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(size_t)(addr + STEP4));   // + 4 time steps: k = 4..7 of the lane's eight
  const s16x8 v = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// 16 zero bytes in device memory: where the lanes of an LDS-DMA piece that fall outside their row (time steps past the end of dy,
// positions in x's zero padding) read from
__device__ u32x4 bl_zero_unit = {0u, 0u, 0u, 0u};

// One LDS-DMA piece: lane i copies the 16-byte unit at `src` (per lane) to LDS bytes dst + 16 i (dst wave-uniform).
// Inline asm on purpose: hipcc (ROCm 7.2) puts s_waitcnt vmcnt(0) in front of every ds_read_b64_tr_b16 intrinsic while an LDS-DMA it
// knows about is in flight (no overlap of the next chunk's tiles with this chunk's MFMAs at all).  Hidden from its bookkeeping, the
// pieces are waited for by the explicit vmcnt(0) in front of the chunk barrier.  M0 (the DMA's LDS base) is written in the same
// statement; nothing else in this kernel keeps a value there.  (First version: buffer_load ... lds with one descriptor per row,
// the descriptor's bounds as the zero padding -- 13 scalar instructions per MFMA, the CU's scalar unit was the bottleneck.)
__device__ __forceinline__ void bl_dma_piece(const u32x4* src, unsigned dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(dst) : "memory");
}

// XC (stride 4, dilation 1: MelGAN L1-L4): an X row is ONE channel bundle at consecutive positions, copied as whole 1 KB pieces (the
// phase rows' lanes gather their units 64 bytes apart: 32 cache lines per piece for 8, and a CU's vector-memory path is what a launch waits
// for -- DESIGN_LOG 13.8); the phases are de-interleaved by the fragment addresses instead: time step t of tap j sits at unit
// (t - t0) S + (j - pad - amin S), a lane's four column bundles (consecutive taps) 16 bytes apart and its four time steps 64 bytes apart --
// 32 distinct banks per half-wave.
constexpr int BLDW_XC_S = 4;
// WN: wave columns of a block (2 x WN waves, 64 FM x 32 WN FN outputs).  [MI355X] cycle stamps of the 2 x 2 form (MelGAN L4, per chunk):
// waiting for the pieces 9, barrier 90, ISSUING the next chunk's ~7 pieces 1 470, fragment reads + MFMAs 1 150 -- a CU takes its blocks'
// pieces at the rate L2 delivers them (28 bytes / clock here, 35-40 at best: tools/ubench/dma_rate.hip), waves that only issue
// (producers) change nothing, and two 128 x 192 blocks per CU ask for 51 KB per 1 536 MFMA cycles.  2 x 4 waves on ONE A tile
// (128 x 384 outputs) ask for 30 KB.
template <int FM, int FN, bool XC = false, int WN = 2>
__device__ __forceinline__ void bl_dw_body(const BlDwArgs& P, unsigned bid, unsigned nblk) {
  constexpr int BM = 64 * FM, BMB = BM / 8, BNQ = 4 * WN * FN, BKT = BLDW_BKT, TS = BLDW_TS, NW = 2 * WN;
  const int RS = P.RS;
  extern __shared__ __attribute__((aligned(16))) u32x4 smem_bldw[];
  typedef __attribute__((address_space(3))) void* lds_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  unsigned id = xcd_remap(bid, nblk);
  const int z = __builtin_amdgcn_readfirstlane(id % P.nsplit); id /= P.nsplit;
  const int nti = __builtin_amdgcn_readfirstlane(id % P.nnt); id /= P.nnt;
  const int mt = __builtin_amdgcn_readfirstlane(id % P.nmt);
  const int g = __builtin_amdgcn_readfirstlane(id / P.nmt);

  const int q0 = nti * BNQ;
  int cb_lo = q0 / P.k;
  if (cb_lo > P.CgB - 1) cb_lo = P.CgB - 1;
  int qlast = q0 + BNQ < P.NQW ? q0 + BNQ : P.NQW;
  int cb_hi = (qlast - 1) / P.k;
  if (cb_hi < cb_lo) cb_hi = cb_lo;
  const int xrows = (cb_hi - cb_lo + 1) * (XC ? 1 : P.S);   // X rows of this tile: (channel bundle, stride phase); XC: channel bundles

  const int a_units = BMB * TS, buf_units = a_units + P.XR * RS;
  // LDS: [ones row: TS units of (1, 0, 0, 0 | 0, 0, 0, 0)] [buffer 0: A rows, X rows] [buffer 1]
  for (int i = tid; i < BLDW_TS; i += 64 * NW) smem_bldw[i] = u32x4{0x00003f80u, 0u, 0u, 0u};   // read at units koff .. koff + 3 (+ 4 time steps of an X row) < 64
  const unsigned lds0 = (unsigned)(unsigned long long)(lds_t)smem_bldw;
  const unsigned ones_addr = lds0, buf_addr = lds0 + BLDW_TS * 16;

  // ---- per-lane fragment addresses (bytes inside a buffer), fixed for the launch -----------------------------------------------
  // ds_read_b64_tr_b16: this lane SUPPLIES the 8-byte row (4 consecutive channels at one time step) of
  //   channel quad 4 (2 (G & 1) + (qs >> 1)) + ... i.e. bundle +2 (G & 1) + (qs >> 1), half qs & 1, time step 8 (G >> 1) + js (+ 4 for the second read)
  const int G4 = lane >> 4, js = (lane & 15) >> 2, qs = lane & 3;
  const int koff = 8 * (G4 >> 1) + js;
  unsigned aoff[FM], boff[FN];
  bool bconst[FN];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int bundle = (wm * FM + i) * 4 + 2 * (G4 & 1) + (qs >> 1);
    aoff[i] = (unsigned)((bundle * TS + koff) * 16 + 8 * (qs & 1));
  }
#pragma unroll
  for (int f = 0; f < FN; ++f) {
    const int q = q0 + 4 * (wn * FN + f) + 2 * (G4 & 1) + (qs >> 1);
    if (q < P.NQW) {
      const int cb = q / P.k, j = q - cb * P.k;
      const int off = j * P.d - P.pad;
      if (XC) {
        boff[f] = (unsigned)(a_units * 16 + ((cb - cb_lo) * RS + (off - P.amin * BLDW_XC_S) + koff * BLDW_XC_S) * 16 + 8 * (qs & 1));
      } else {
        int a = off / P.S;
        if (off < 0 && a * P.S != off) --a;                // floor
        const int p = off - a * P.S;
        const int row = (cb - cb_lo) * P.S + p;
        boff[f] = (unsigned)(a_units * 16 + (row * RS + (a - P.amin) + koff) * 16 + 8 * (qs & 1));
      }
      bconst[f] = false;
    } else {                                             // the bias column bundle (and the padding behind it): constant rows
      boff[f] = (unsigned)(koff * 16 + 8 * (qs & 1));
      bconst[f] = true;
    }
  }

  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int f = 0; f < FN; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][f][r] = 0.f;

  // ---- tile movement: every row is a run of 64 (A) / 2 x 64 (X) units, one LDS-DMA piece each ---------------------------------------
  // rows of this wave (fixed for the launch): A rows r = wave, wave + 4, ... (bundle clamped into the tensor: rows past it are
  // computed and not stored), X rows xr = wave, wave + 4, ... = (channel bundle, stride phase)
  const u32x4* zero = &bl_zero_unit;
  constexpr int AR = (BMB + NW - 1) / NW;
  long long arow[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    int bundle = g * P.MgB + mt * BMB + wave + NW * i;
    if (bundle > P.CBa - 1) bundle = P.CBa - 1;
    arow[i] = mt * BMB + wave + NW * i < P.MgB ? (long long)bundle * P.La : -1;   // rows past the group's last: zeros (no traffic), never stored
  }
  // what a chunk's pieces need besides (item, first time step): fixed for the launch (the divisions and 64-bit products of the row
  // addresses were ~240 mostly scalar instructions per chunk and wave, in front of the chunk's 16-32 MFMAs)
  constexpr int XW = 4;                                  // X rows per wave held in registers (XR <= 16); wider tiles walk the rows
  const bool x_regs = XC || P.XR <= NW * XW;
  long long xoff[XW];
  int xlane[XW];
  unsigned xdst[XW];
  bool xact[XW];                                         // XC: this lane of piece i is inside the row's xw units
  const int npr = XC ? (P.xw + 63) / 64 : 1;             // XC: pieces per row; piece wave + 4 i = (row, 64-unit run)
#pragma unroll
  for (int i = 0; i < XW; ++i) {
    const int xr = wave + NW * i;
    if (XC) {
      const int row = xr / npr, pc = xr - row * npr;
      int cb = cb_lo + row;
      if (cb > P.CgB - 1) cb = P.CgB - 1;
      xoff[i] = ((long long)g * P.CgB + cb) * P.Lx;
      xlane[i] = P.amin * BLDW_XC_S + 64 * pc + lane;    // unit u of the row = position (t0 + amin) S + u
      xdst[i] = (unsigned)((a_units + row * RS + 64 * pc) * 16);
      xact[i] = xr < xrows * npr && 64 * pc + lane < P.xw;
    } else {
      const int cbl = xr / P.S, p = xr - cbl * P.S;
      int cb = cb_lo + cbl;
      if (cb > P.CgB - 1) cb = P.CgB - 1;
      xoff[i] = ((long long)g * P.CgB + cb) * P.Lx;
      xlane[i] = (P.amin + lane) * P.S + p;              // unit u of the row = position (t0 + amin + u) S + p
      xdst[i] = (unsigned)((a_units + xr * RS) * 16);
      xact[i] = true;
    }
  }
  const long long a_item = (long long)P.CBa * P.La, x_item = (long long)P.CBx * P.Lx;
  auto issue = [&](int b, int t0, int bsel) {
    const unsigned dst = buf_addr + (unsigned)(bsel * buf_units * 16);
    const u32x4* ab = P.ah + (long long)b * a_item + t0 + lane;
    const bool a_ok = t0 + lane < P.La;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const int r = wave + NW * i;
      if (r < BMB && !(EBEN_BLDW_DBG & 1)) bl_dma_piece((a_ok && arow[i] >= 0) ? ab + arow[i] : zero, __builtin_amdgcn_readfirstlane(dst + (unsigned)(r * TS * 16)));
    }
    const u32x4* xb = P.xh + (long long)b * x_item;
    if (EBEN_BLDW_DBG & 2) return;
    // outside [0, Lx): the zero unit.  The second piece covers the taps' reach behind the 64 time steps: its lanes past xneed are
    // switched off (the row's stride is RS < 128 units)
    if (XC) {
      const int ts = t0 * BLDW_XC_S;
#pragma unroll
      for (int i = 0; i < XW; ++i)
        if (wave + NW * i < xrows * npr) {                // wave-uniform
          const int pos0 = ts + xlane[i];
          if (xact[i]) bl_dma_piece((unsigned)pos0 < (unsigned)P.Lx ? xb + xoff[i] + pos0 : zero, __builtin_amdgcn_readfirstlane(dst + xdst[i]));
        }
      return;
    }
    if (x_regs) {
      const int ts = t0 * P.S;
#pragma unroll
      for (int i = 0; i < XW; ++i)
        if (wave + NW * i < xrows) {
          const u32x4* row = xb + xoff[i];
          const int pos0 = ts + xlane[i];
          const unsigned rdst = dst + xdst[i];
          bl_dma_piece((unsigned)pos0 < (unsigned)P.Lx ? row + pos0 : zero, __builtin_amdgcn_readfirstlane(rdst));
          if (64 + lane < P.xneed) {
            const int pos1 = pos0 + 64 * P.S;
            bl_dma_piece((unsigned)pos1 < (unsigned)P.Lx ? row + pos1 : zero, __builtin_amdgcn_readfirstlane(rdst + (unsigned)(64 * 16)));
          }
        }
      return;
    }
    for (int xr = wave; xr < xrows; xr += NW) {
      const int cbl = xr / P.S, p = xr - cbl * P.S;
      int cb = cb_lo + cbl;
      if (cb > P.CgB - 1) cb = P.CgB - 1;
      const u32x4* row = xb + ((long long)g * P.CgB + cb) * P.Lx;
      const unsigned rdst = dst + (unsigned)((a_units + xr * RS) * 16);
      const int pos0 = (t0 + P.amin + lane) * P.S + p;
      bl_dma_piece((pos0 >= 0 && pos0 < P.Lx) ? row + pos0 : zero, __builtin_amdgcn_readfirstlane(rdst));
      if (64 + lane < P.xneed) {
        const int pos1 = pos0 + 64 * P.S;
        bl_dma_piece((pos1 >= 0 && pos1 < P.Lx) ? row + pos1 : zero, __builtin_amdgcn_readfirstlane(rdst + (unsigned)(64 * 16)));
      }
    }
  };
  // chunk q = (item q / nct, time steps 64 (q mod nct) ..): walked without divisions
  const int dq_b = P.nsplit / P.nct, dq_t = P.nsplit - dq_b * P.nct;
  int qb = z / P.nct, qt = z - qb * P.nct;             // the chunk whose pieces are issued next

  int bsel = 0;
  auto advance = [&]() {
    qb += dq_b; qt += dq_t;
    if (qt >= P.nct) { qt -= P.nct; ++qb; }
  };
  if (z < P.nchunks) { issue(qb, qt * BKT, 0); advance(); }
#ifdef EBEN_BLDW_STAMP
  unsigned long long st_wait = 0, st_bar = 0, st_issue = 0, st_mma = 0, st_n = 0;
  const unsigned long long st_begin = __builtin_amdgcn_s_memtime();
#endif
  for (int q = z; q < P.nchunks; q += P.nsplit) {
#ifdef EBEN_BLDW_STAMP
    BLDW_T(ta);
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of chunk q have landed ...
#ifdef EBEN_BLDW_STAMP
    BLDW_T(tb);
#endif
    __syncthreads();                                   // ... and everybody's; buffer bsel ^ 1 (read during the previous chunk) is free
#ifdef EBEN_BLDW_STAMP
    BLDW_T(tc);
#endif
    if (q + P.nsplit < P.nchunks) { issue(qb, qt * BKT, bsel ^ 1); advance(); }
#ifdef EBEN_BLDW_STAMP
    BLDW_T(td);
#endif
    const unsigned base = buf_addr + (unsigned)(bsel * buf_units * 16);
    // fragments of k-step ks + 1 are read while k-step ks is multiplied (two register sets: left to itself hipcc reads a k-step's
    // fragments into ONE set right in front of its MFMAs -- two exposed LDS round trips per 6-8 MFMAs)
    // ONE transposing read per fragment half for all lanes (the read exchanges rows inside 16-lane groups: a lane of the bias column
    // bundle only reads another address -- every unit of the ones row is the same, so it needs no k-step advance)
    constexpr unsigned BSTEP4 = XC ? 64u * BLDW_XC_S : 64u;   // bytes between time steps t and t + 4 of an X row
    bf16x8 av[2][FM], bv[2][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) av[0][i] = bl_tr_frag(base + aoff[i]);
#pragma unroll
    for (int f = 0; f < FN; ++f) bv[0][f] = bl_tr_frag<BSTEP4>(bconst[f] ? ones_addr + boff[f] : base + boff[f]);
#pragma unroll
    for (int ks = 0; ks < BKT / 16; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      if (ks + 1 < BKT / 16) {
#pragma unroll
        for (int i = 0; i < FM; ++i) av[nxt][i] = bl_tr_frag(base + aoff[i] + (unsigned)((ks + 1) * 256));
#pragma unroll
        for (int f = 0; f < FN; ++f)
          bv[nxt][f] = bl_tr_frag<BSTEP4>(bconst[f] ? ones_addr + boff[f] : base + boff[f] + (unsigned)((ks + 1) * 4 * BSTEP4));
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int f = 0; f < FN; ++f) {
          if (EBEN_BLDW_DBG & 4) acc[i][f][0] += __builtin_bit_cast(float, __builtin_bit_cast(u32x4, av[cur][i])[0] ^ __builtin_bit_cast(u32x4, bv[cur][f])[1]);
          else acc[i][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[cur][i], bv[cur][f], acc[i][f], 0, 0, 0);
        }
#if EBEN_BLDW_PIPE_FENCE
      __builtin_amdgcn_sched_barrier(0);   // k-step ks's MFMAs and k-step ks + 1's reads stay in front of k-step ks + 1's MFMAs
#endif
    }
    bsel ^= 1;
#ifdef EBEN_BLDW_STAMP
    { BLDW_T(te); st_wait += tb - ta; st_bar += tc - tb; st_issue += td - tc; st_mma += te - td; ++st_n; }
#endif
  }
#ifdef EBEN_BLDW_STAMP
  if (tid == 0 && bid < 8192u) {
    unsigned long long* o = bldw_stamp + bid * 8;
    o[0] = st_wait; o[1] = st_bar; o[2] = st_issue; o[3] = st_mma; o[4] = st_n; o[5] = __builtin_amdgcn_s_memtime() - st_begin;
  }
#endif

  // ---- epilogue: slab[z][row][column] ------------------------------------------------------------------------------------------------
  float* slab = P.slabs + (long long)z * P.slab_stride;
  const int m_base = mt * BM + wm * FM * 32 + 4 * (lane >> 5);
#pragma unroll
  for (int f = 0; f < FN; ++f) {
    const int nl = lane & 31;
    const int q = q0 + 4 * (wn * FN + f) + (nl >> 3), e = nl & 7;
    long long col = -1;
    int cgrp = -1;
    if (!P.dense) {
      // bundle-major columns: 8 q + e (contiguous over the lanes); the bias bundle keeps its first column
      if (q < P.NQW || (q == P.NQW && e == 0 && P.has_bias)) col = 8LL * q + e;
    } else if (q < P.NQW) {
      const int cb = q / P.k, j = q - cb * P.k, c = 8 * cb + e;
      cgrp = c / P.c_in_g;
      col = (long long)(c - cgrp * P.c_in_g) * P.k + j;
    } else if (q == P.NQW && e == 0 && P.has_bias) {
      col = (long long)P.c_in_g * P.k;
    }
    if (col < 0) continue;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m_base + i * 32 + (r & 3) + 8 * (r >> 2);
        if (m >= P.Mg) continue;
        if (P.dense && cgrp >= 0 && m / P.c_out_g != cgrp) continue;
        if (!(EBEN_BLDW_DBG & 8) || acc[i][f][r] == 12345.f) slab[((long long)g * P.Mg + m) * P.row_stride + col] = acc[i][f][r];
      }
  }
}

template <int FM, int FN, bool XC = false, int WN = 2>
__global__ __launch_bounds__(128 * WN, 2) void bl_dw_kernel(const BlDwArgs P) { bl_dw_body<FM, FN, XC, WN>(P, blockIdx.x, gridDim.x); }

// Several layers of ONE tile shape in one launch (the same layer index of the three PQMF-band discriminators: same channels and taps,
// their own dilation, length, operands and slabs): block -> problem by the prefix sums of the problems' block counts.  [MI355X] a thin
// layer's launch is mostly fixed cost (64 rows 37-40 us, 192 rows 66-87 us): three problems per launch pay it once.
constexpr int BLDW_MULTI = 4;
struct BlDwTable {
  int n;
  unsigned first[BLDW_MULTI + 1];
  BlDwArgs job[BLDW_MULTI];
};
template <int FM, int FN>
__global__ __launch_bounds__(256, 2) void bl_dw_multi_kernel(const BlDwTable T) {
  int j = 0;
#pragma unroll 1
  while (j + 1 < T.n && blockIdx.x >= T.first[j + 1]) ++j;
  j = __builtin_amdgcn_readfirstlane(j);
  bl_dw_body<FM, FN>(T.job[j], blockIdx.x - T.first[j], T.first[j + 1] - T.first[j]);
}

// ---------------------------------------------------------------------------------------------------------------------------------
struct BlDwPlan {
  int ok, G, Mg, Cg, MgB, CgB, dense, FM, FN, nmt, nnt, NQW, nct, nchunks, nsplit, XR, amin, row_stride, perm_k, xneed, RS, xc, xw, wide;
  size_t lds_bytes;
  long long slab_stride;
};

static int bldw_floordiv(int a, int b) { int q = a / b; if ((a % b != 0) && ((a < 0) != (b < 0))) --q; return q; }

static void make_bldw_plan(const Canon& c, BlDwPlan* p) {
  p->ok = 0;
  if (!c.bl || c.reflect || c.np != 1) return;          // single bf16 operands (EBEN_MATH_BF16)
  p->G = c.g; p->Mg = c.Cout / c.g; p->Cg = c.Cin / c.g; p->dense = 0;
  if ((p->Mg & 7) || (p->Cg & 7)) { p->dense = 1; p->G = 1; p->Mg = c.Cout; p->Cg = c.Cin; }
  if ((p->Mg & 7) || (p->Cg & 7)) return;
  p->MgB = p->Mg / 8; p->CgB = p->Cg / 8;
  p->amin = bldw_floordiv(-c.pl, c.s);
  const int amax = bldw_floordiv((c.k - 1) * c.d - c.pl, c.s);
  if (amax - p->amin > 60 || c.s > 8) return;           // BKT + halo units per X row must fit the two DMA pieces
  p->FM = (p->Mg > 64 && round_up(p->Mg, 128) == round_up(p->Mg, 64)) ? 2 : 1;
  p->NQW = p->CgB * c.k;
  // column bundles per wave: the A tile (64 FM rows x 64 time steps) and the X rows (every stride phase of the ~2 channel bundles a
  // tile's columns touch, 128 units each) are moved per chunk whatever the tile's width -- at 128 columns a 128-row tile asks the
  // L2 for 32 KB per 64 MFMAs, ~50 TB/s chip-wide at the MFMA rate; 256 columns halve that at the same LDS footprint
  static const int fn_max = getenv("EBEN_BLDW_FN") ? atoi(getenv("EBEN_BLDW_FN")) : 4;
  // [MI355X, 64 rows] MelGAN L1-L5 0.124 / 0.106 / 0.286 / 0.331 / 0.175 -> 0.097 / 0.085 / 0.255 / 0.301 / 0.146 ms; the PQMF-band
  // layers (7 taps, 42-84 column bundles: three 256-column tiles where six 128-column ones covered them as well) lose 10-40 %
  p->FN = (fn_max >= 4 && (c.k >= 16 || p->NQW >= 256)) ? 4 : p->NQW + 1 > 8 ? 2 : 1;
  // 192-column tiles where they cover the layer's columns in fewer tiles than 128-column ones (22 column bundles: 1 tile for 2; 43: 2
  // for 3): every tile reads the whole A rows and (these layers: 3-6 channel bundles) nearly all of the X rows, so traffic falls with
  // the tile count
  static const int fn3_thin = getenv("EBEN_BLDW_FN3_THIN") ? atoi(getenv("EBEN_BLDW_FN3_THIN")) : 1;
  if (fn3_thin && p->FN == 2 && ceil_div(p->NQW + 1, 24) < ceil_div(p->NQW + 1, 16)) p->FN = 3;
  // 192-column tiles where 256-column ones leave the second round of block slots mostly empty: MelGAN L4's 328 tiles become 440 on
  // 512 slots ([MI355X] 0.287 -> 0.263 ms; forced on the other layers it loses 3-10 %: more A-tile bytes per MFMA)
  static const int fn3 = getenv("EBEN_BLDW_FN3") ? atoi(getenv("EBEN_BLDW_FN3")) : 1;
  if (fn3 && p->FN == 4 && p->FM == 2) {   // 128-row tiles only (the one shape it pays on; no 64-row instantiation)
    const int t4 = ceil_div(p->NQW + 1, 32) * ceil_div(p->Mg, 64 * p->FM) * p->G, t3 = ceil_div(p->NQW + 1, 24) * ceil_div(p->Mg, 64 * p->FM) * p->G;
    if (fn3 == 2 || (t4 <= 512 && t4 > 256 && t3 <= 512)) p->FN = 3;
  }
  // eight waves on one A tile (bl_dw_body<2, 3, .., 4>: 128 x 384 outputs, one block per CU): the 128-row layers with at least one full
  // tile of columns -- half the A bytes per MFMA through the CU's vector-memory path, which is what these launches wait for
  static const int wide_on = getenv("EBEN_BLDW_WIDE") ? atoi(getenv("EBEN_BLDW_WIDE")) : 1;
  // [MI355X] MelGAN L3 / L4 (stride 4) 196 / 205 -> 185 / 186 us; L5 (stride 1, eleven X rows per tile) 103 -> 112: stride-4 layers only
  p->wide = (wide_on && p->FM == 2 && p->FN >= 3 && p->NQW + 1 >= 48 && c.s == BLDW_XC_S && c.d == 1) ? 1 : 0;
  if (p->wide) p->FN = 3;
  // contiguous X rows (bl_dw_body<.., XC>): stride 4, dilation 1, the wide tiles, at most four pieces per wave and chunk
  static const int xc_on = getenv("EBEN_BLDW_XC") ? atoi(getenv("EBEN_BLDW_XC")) : 1;
  p->nmt = ceil_div(p->Mg, 64 * p->FM);
  p->xneed = BLDW_BKT + amax - p->amin + 1;
  for (;;) {
    const int BNQ = (p->wide ? 16 : 8) * p->FN;
    p->nnt = ceil_div(p->NQW + 1, BNQ);
    int ncb = (BNQ - 1) / c.k + 2;
    if (ncb > p->CgB) ncb = p->CgB;
    p->XR = ncb * c.s;
    p->RS = bldw_row_stride(p->xneed);
    p->xc = 0; p->xw = 0;
    if (xc_on && c.s == BLDW_XC_S && c.d == 1 && p->FN >= 3) {
      const int xw = p->xneed * c.s;
      if (ncb * ceil_div(xw, 64) <= (p->wide ? 32 : 16)) {
        p->xc = 1; p->xw = xw; p->XR = ncb;
        p->RS = xw + 1;                                      // any stride: rows only meet at a tile's channel-bundle boundary
      }
    }
    if (!p->wide || p->xc) break;
    p->wide = 0;                                           // the eight-wave form exists with contiguous X rows only
  }
  p->lds_bytes = 16ull * (BLDW_TS + 2ull * ((64 * p->FM / 8) * BLDW_TS + (unsigned long long)p->XR * p->RS));
  if (p->lds_bytes > 160 * 1024) return;
  p->nct = ceil_div(c.Lout, BLDW_BKT);
  p->nchunks = c.B * p->nct;
  const int tiles = p->nnt * p->nmt * p->G;
  // Split-K factor: the grid runs in rounds of (CUs x resident blocks per CU) blocks and a kernel this short pays for the empty part
  // of its last round (664 tiles on 512 slots = two rounds for 1.3 rounds of work), while every extra slab is one more write of the
  // gradient by the epilogue and one more read by the reduction.  Pick the factor with the least estimated time: rounds x the work of
  // one block + slab traffic at ~4 TB/s.
  static const int cus = getenv("EBEN_BLDW_CUS") ? atoi(getenv("EBEN_BLDW_CUS")) : 256;
  const int per_cu = p->wide ? 1 : (int)(160 * 1024 / p->lds_bytes) < 1 ? 1 : (int)(160 * 1024 / p->lds_bytes);   // wide: 8 waves x ~150 registers
  const int slots = cus * (per_cu > 4 ? 4 : per_cu);
  const double macs = (double)c.B * c.Lout * (double)c.Cout * (c.Cin / c.g) * c.k;
  const double t_all = 2.0 * macs / 1.0e15;                                   // the whole contraction at ~1 PFLOP/s
  const double slab_bytes = 4.0 * (double)c.Cout * ((c.Cin / c.g) * c.k + 1);
  int ns = 1;
  double best = 1e30;
  for (int cand = 1; cand <= 256 && cand <= p->nchunks; ++cand) {
    const long long blocks = (long long)tiles * cand;
    const double rounds = (double)((blocks + slots - 1) / slots);
    // [MI355X] MelGAN L4 (43 MB per slab): one slab 0.327 ms, two 0.339 -- the slab traffic costs more than its 4 TB/s share
    const double t = rounds * (t_all * slots / blocks) + 2.0 * cand * slab_bytes / 2.0e12 + (cand > 1 ? slab_bytes / 2.0e12 + 5e-6 : 0.0);
    if (t < best * 0.97) { best = t; ns = cand; }
  }
  static const int force_ns = getenv("EBEN_BLDW_NSPLIT") ? atoi(getenv("EBEN_BLDW_NSPLIT")) : 0;
  if (force_ns > 0 && force_ns <= p->nchunks) ns = force_ns;
  p->nsplit = ns;
  if (p->dense) { p->row_stride = (c.Cin / c.g) * c.k + 1; p->perm_k = 0; }
  else { p->row_stride = p->Cg * c.k + 1; p->perm_k = c.k; }
  p->slab_stride = (long long)c.Cout * p->row_stride;
  p->ok = 1;
}

template <int FM, int FN, bool XC = false, int WN = 2>
static int launch_bldw(const BlDwArgs& a, const BlDwPlan& p, hipStream_t st) {
  static LdsAttrOnce attr_once;
  auto kern = bl_dw_kernel<FM, FN, XC, WN>;
  {
    const hipError_t e = lds_attr_once(attr_once, reinterpret_cast<const void*>(kern));
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(bl_dw)");
  }
  const long long nb = (long long)p.nnt * p.nmt * p.G * p.nsplit;
  hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(128 * WN), p.lds_bytes, st, a);
  EBEN_CHECK_LAUNCH("bl_dw_kernel");
  return EBEN_OK;
}

}  // namespace eben

using namespace eben;

extern "C" size_t eben_bl_conv1d_bwd_dw_workspace(const EbenConv1dDesc* d, int* nslab, int* row_stride, int* col_perm_k) {
  Canon c;
  if (canon_from_desc(d, &c) != EBEN_OK || d->transposed) return 0;
  BlDwPlan p;
  make_bldw_plan(c, &p);
  if (!p.ok) return 0;
  if (nslab) *nslab = p.nsplit;
  if (row_stride) *row_stride = p.row_stride;
  if (col_perm_k) *col_perm_k = p.perm_k;
  return sizeof(float) * (size_t)p.slab_stride * p.nsplit;
}

static int bldw_args(const EbenConv1dDesc* d, const void* dy_hi, const void* x_hi, int has_bias, float* slabs, size_t ws_bytes, BlDwArgs* out,
                     BlDwPlan* plan) {
  Canon c;
  int rc = canon_from_desc(d, &c);
  if (rc) return rc;
  EBEN_REQUIRE(!d->transposed && dy_hi && x_hi && slabs, "eben_bl_conv1d_bwd_dw: a Conv1d descriptor and non-null operands");
  BlDwPlan p;
  make_bldw_plan(c, &p);
  if (!p.ok) return fail(EBEN_EUNSUPPORTED, "eben_bl_conv1d_bwd_dw: layer not covered by the bundle-layout weight-gradient kernel");
  const size_t need = sizeof(float) * (size_t)p.slab_stride * p.nsplit;
  if (ws_bytes < need) return fail(EBEN_EWORKSPACE, "bl bwd_dw needs %zu workspace bytes, got %zu", need, ws_bytes);
  BlDwArgs a;
  a.ah = static_cast<const u32x4*>(dy_hi); a.xh = static_cast<const u32x4*>(x_hi); a.slabs = slabs;
  a.B = c.B; a.G = p.G; a.Mg = p.Mg; a.Cg = p.Cg; a.MgB = p.MgB; a.CgB = p.CgB; a.CBa = c.Cout / 8; a.CBx = c.Cin / 8; a.La = c.Lout; a.Lx = c.Lin;
  a.S = c.s; a.d = c.d; a.k = c.k; a.pad = c.pl; a.amin = p.amin; a.NQW = p.NQW; a.has_bias = has_bias ? 1 : 0;
  a.nnt = p.nnt; a.nmt = p.nmt; a.nsplit = p.nsplit; a.nct = p.nct; a.nchunks = p.nchunks; a.XR = p.XR;
  a.dense = p.dense; a.c_in_g = c.Cin / c.g; a.c_out_g = c.Cout / c.g; a.row_stride = p.row_stride; a.slab_stride = p.slab_stride;
  a.xneed = p.xneed; a.RS = p.RS; a.xw = p.xw;
  *out = a; *plan = p;
  return EBEN_OK;
}

extern "C" int eben_bl_conv1d_bwd_dw(const EbenConv1dDesc* d, const void* dy_hi, const void* x_hi, int has_bias, float* slabs, size_t ws_bytes,
                                     void* stream) {
  BlDwArgs a;
  BlDwPlan p;
  const int rc = bldw_args(d, dy_hi, x_hi, has_bias, slabs, ws_bytes, &a, &p);
  if (rc) return rc;
  hipStream_t st = as_stream(stream);
  if (p.wide) return launch_bldw<2, 3, true, 4>(a, p, st);
  if (p.xc) return p.FM == 2 ? (p.FN == 4 ? launch_bldw<2, 4, true>(a, p, st) : launch_bldw<2, 3, true>(a, p, st))
                             : (p.FN == 4 ? launch_bldw<1, 4, true>(a, p, st) : launch_bldw<1, 3, true>(a, p, st));
  if (p.FM == 2) return p.FN == 4 ? launch_bldw<2, 4>(a, p, st) : p.FN == 3 ? launch_bldw<2, 3>(a, p, st) : p.FN == 2 ? launch_bldw<2, 2>(a, p, st) : launch_bldw<2, 1>(a, p, st);
  return p.FN == 4 ? launch_bldw<1, 4>(a, p, st) : p.FN == 3 ? launch_bldw<1, 3>(a, p, st) : p.FN == 2 ? launch_bldw<1, 2>(a, p, st) : launch_bldw<1, 1>(a, p, st);
}

namespace eben {
template <int FM, int FN>
static int launch_bldw_multi(const BlDwTable& T, size_t lds, hipStream_t st) {
  static LdsAttrOnce attr_once;
  auto kern = bl_dw_multi_kernel<FM, FN>;
  {
    const hipError_t e = lds_attr_once(attr_once, reinterpret_cast<const void*>(kern));
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(bl_dw_multi)");
  }
  hipLaunchKernelGGL(kern, dim3(T.first[T.n]), dim3(256), lds, st, T);
  EBEN_CHECK_LAUNCH("bl_dw_multi_kernel");
  return EBEN_OK;
}
}  // namespace eben

extern "C" int eben_bl_conv1d_bwd_dw_multi(const EbenConv1dDesc* const* descs, const void* const* dy_hi, const void* const* x_hi, int has_bias,
                                           float* const* slabs, const size_t* ws_bytes, int n, void* stream) {
  EBEN_REQUIRE(descs && dy_hi && x_hi && slabs && ws_bytes && n > 0, "bad eben_bl_conv1d_bwd_dw_multi arguments");
  hipStream_t st = as_stream(stream);
  // runs of consecutive problems with the same tile shape share a launch; anything else gets its own
  int i = 0;
  while (i < n) {
    BlDwTable T;
    T.n = 0; T.first[0] = 0;
    BlDwPlan p0{};
    size_t lds = 0;
    while (i < n && T.n < BLDW_MULTI) {
      BlDwArgs a;
      BlDwPlan p;
      const int rc = bldw_args(descs[i], dy_hi[i], x_hi[i], has_bias, slabs[i], ws_bytes[i], &a, &p);
      if (rc) return rc;
      if (p.xc || p.wide) {   // contiguous-X (stride 4) and eight-wave layers are not grouped: their own launch
        if (T.n > 0) break;
        const int rc1 = eben_bl_conv1d_bwd_dw(descs[i], dy_hi[i], x_hi[i], has_bias, slabs[i], ws_bytes[i], stream);
        if (rc1) return rc1;
        ++i;
        continue;
      }
      if (T.n > 0 && (p.FM != p0.FM || p.FN != p0.FN)) break;
      const long long nb = (long long)p.nnt * p.nmt * p.G * p.nsplit;
      if (nb <= 0 || nb > 0x3fffffffLL - (long long)T.first[T.n]) return fail(EBEN_EINVAL, "bl bwd_dw_multi grid too large");
      if (T.n == 0) p0 = p;
      T.job[T.n] = a;
      T.first[T.n + 1] = T.first[T.n] + (unsigned)nb;
      if (p.lds_bytes > lds) lds = p.lds_bytes;
      ++T.n; ++i;
    }
    if (T.n == 0) continue;
    int rc;
    if (p0.FM == 2) rc = p0.FN == 4 ? launch_bldw_multi<2, 4>(T, lds, st) : p0.FN == 3 ? launch_bldw_multi<2, 3>(T, lds, st) : p0.FN == 2 ? launch_bldw_multi<2, 2>(T, lds, st) : launch_bldw_multi<2, 1>(T, lds, st);
    else rc = p0.FN == 4 ? launch_bldw_multi<1, 4>(T, lds, st) : p0.FN == 3 ? launch_bldw_multi<1, 3>(T, lds, st) : p0.FN == 2 ? launch_bldw_multi<1, 2>(T, lds, st) : launch_bldw_multi<1, 1>(T, lds, st);
    if (rc) return rc;
  }
  return EBEN_OK;
}

#ifdef EBEN_BLDW_STAMP
extern "C" __attribute__((visibility("default"))) int eben_debug_bldw_stamps(unsigned long long* out, size_t n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(eben::bldw_stamp), n * sizeof(unsigned long long));
}
#endif
