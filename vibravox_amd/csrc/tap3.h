// Launch plan and kernel arguments of the bf16 tap-conv family: tapconv3.hip (tap3_kernel, the general form) and bigtap.hip (tap4_kernel, the
// persistent 256-thread form of the MFMA-bound bundle-layout launches).  Both read the same packed weight image + k-step table.
#pragma once
#include "common.h"

namespace eben {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Tap3Args {
  const float* x; const float* xmask; const u32x4* wp; const int* tab;
  const float* bias; const float* res; const float* emask; float* y;
  int B, G, Cg, Mg, Cx, Cy, Lx, Ly;
  int S, OS, dstep, J0, mode, off0, nt, nph;
  int ps_pad, ps_k, ps_d, ps_kstep;
  int reflect, accumulate, in_mode;   // in_mode 1: the input is multiplied by lrelu'(xmask) as it is staged (autograd's mask-on-load)
  int res_rows, em_seg, em_map[4];
  const float* fm_sums; float fm_gs;
  float in_slope, out_slope, res_slope, emask_slope;
  int CI_T, CI_B, CP, ncc, PLEN, CSTRIDE, nxb;   // CI_T channels = CI_B bundles per input tile; CP = CI_B / 2 k-steps per tap
  unsigned s_magic;
  int ntt, nmt, tab_phase;
  long long w_tile, w_phase;                     // in 16-byte units
  // host-side arithmetic of the block prologue (integer divisions are ~40 instructions each on the device, the 64-bit one
  // behind span_magic ~150): the block-id decomposition by multiply-high, and the per-phase tap geometry for up to 8 phases
  // bundle layout (BL: bf16 [batch][channels / 8][length][8] planes hi = bf16(v), lo = bf16(v - hi); template flag BL): the input
  // planes (xl: split input, NPX = 2), the output planes (yl nullable), the saved activation the epilogue reads its mask /
  // feature-matching operands from (eh / el; the reference rows of a feature-matching pair start bl_ref_off batch rows further)
  const u32x4* xh; const u32x4* xl;
  uint2* yh; uint2* yl;
  const uint2* eh; const uint2* el;
  const unsigned* ec;   // nullable: feature-matching codes of the embedding (4 bytes per half unit), read by the rows b < res_rows instead of el and the reference rows
  int CBx, CBy, bl_ref_off, bl_pad;
  // phases as rows (TapIO.pr_S): the logical output rows of group g are (phase, channel of the group) and land in the PHYSICAL planes
  // [row][pr_CB][pr_Ly][8] at bundle g pr_cbg + (logical bundle % pr_cbg), position t pr_S + logical bundle / pr_cbg
  int pr_S, pr_cbg, pr_Ly, pr_order;   // pr_order 1 (tap4_kernel): logical bundle = physical bundle * pr_S + phase
  unsigned xq, xr;                               // gridDim.x / 8, gridDim.x % 8 (xcd_remap)
  unsigned m_nph, m_ntt, m_B, m_nmt;             // ceil(2^32 / d); valid when id_fast
  int id_fast, pg_n;
  // tap4_kernel (Tap3Plan.big): units per staged input tile and piece (a whole number of 64-unit LDS-DMA pieces), pieces per tile,
  // tiles of the launch, divisions of the staging arithmetic (ceil(2^32 / d))
  int big_XT, big_PPT, big_tiles;
  unsigned m_cstride, m_plen;
  struct PG { int J, off0, minoff, nt, oo, span; unsigned span_magic; int pad; } pg[8];
};

struct Tap3Plan {
  int ok;
  int mode, G, Cg, Mg, S, OS, dstep, kstep, nph, J, Lx, Ly, Cx, Cy, off0, nt, ps_pad;
  int FM, BM, BN, WCHU;
  int dense;   // groups folded into ONE block-diagonal contraction (layers with a handful of channels per group)
  int npw, npx, KSC;   // pieces per weight / per input element (tap3_kernel), k-steps per weight chunk
  int CI_T, CI_B, CP, ncc, PLEN, CSTRIDE, nxbuf, XRB;
  int nmt, ntt, NCH, tab_phase;
  long long w_tile, w_phase, tab_off_floats;
  size_t packed_floats, lds_bytes;
  // tap4_kernel (bigtap.hip): wave grid WM x WN, wave tile TM x TN MFMA tiles, weight ring depth; xbuf_stride = units from one input
  // buffer to the next in the k-step table (tap3_kernel: CI_B * CSTRIDE)
  int big, WM, WN, TM, TN, RING, XT, PPT, xbuf_stride;
};

// bigtap.hip
int tap4_launch(const Tap3Plan& p, const Tap3Args& a, hipStream_t st);

}  // namespace eben
