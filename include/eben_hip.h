/*
 * eben_hip.h -- C ABI of libeben_hip.so: the MI355X (gfx950) kernels of the EBEN
 * bandwidth-extension GAN train step (vibravox: EBENGenerator +
 * DiscriminatorEBENMultiScales forward/backward, PQMF, feature/hinge/MRSTFT losses, Adam).
 *
 * The reference has no native code: every "kernel" on its hot path is
 * an ATen call site inside a Python nn.Module.  Each entry point below therefore cites the
 * reference *call site* it replaces (file:line relative to the vibravox checkout); the
 * binding a maintainer adds on the reference side is a ctypes stub.
 *
 * Conventions
 *   - all tensors are contiguous float32 (batch, channel, time) device buffers; pointers are
 *     raw device pointers (tensor.data_ptr()); `stream` is a hipStream_t passed as void*.
 *   - every call is asynchronous on `stream`, re-entrant, allocates nothing and never
 *     synchronises; scratch comes from a caller-provided workspace.
 *   - return 0 on success, a negative EBEN_E* code on a bad argument, or a positive
 *     hipError_t from the launch.  eben_last_error() returns a thread-local message.
 *     Nothing throws across the ABI.
 */
#ifndef EBEN_HIP_H
#define EBEN_HIP_H

#include <stddef.h>
#include <stdint.h>

#if defined(EBEN_BUILDING)
#define EBEN_API __attribute__((visibility("default")))
#else
#define EBEN_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define EBEN_OK 0
#define EBEN_EINVAL (-1)      /* inconsistent descriptor / null pointer */
#define EBEN_EWORKSPACE (-2)  /* workspace too small */
#define EBEN_EUNSUPPORTED (-3)

#define EBEN_PAD_ZERO 0
#define EBEN_PAD_REFLECT 1

/* arithmetic of the contraction (storage, accumulation and the fused element-wise stages are fp32 either way) */
#define EBEN_MATH_F32 0   /* v_mfma_f32_*_f32 / fp32 FMA: bit-exact fp32 products */
#define EBEN_MATH_BF16 1  /* both MFMA operands rounded (RNE) to bf16 -- after the fused input stage (LeakyReLU or its
                           * derivative mask) -- fp32 accumulate; layers the bf16 kernels do not cover run their fp32 kernel */
#define EBEN_MATH_BF16X2 2 /* EBEN_MATH_BF16 with the ACTIVATION operand -- x of the layer's forward and of its weight gradient --
                           * entering as hi + lo, hi = bf16(x), lo = bf16(x - hi): two MFMAs per k-step against the same bf16
                           * weight / gradient fragment, x accurate to ~2^-17.  The discriminator's activations are a large
                           * input-independent component plus a small input-dependent one; rounding them to 8 bits buries the
                           * second, which is all that survives when the fake and real hinge gradients (eben.py:121-125) cancel.
                           * Gradient operands and weights stay single bf16; the input-gradient launch equals EBEN_MATH_BF16. */
#define EBEN_MATH_BF16X3 3 /* fused ResidualUnit launches (eben_ru_*_ex): both operands as hi + lo bf16, three MFMAs per product */
#define EBEN_MATH_BF16X6 4 /* ... as three bf16 pieces each, six MFMAs: every mantissa bit of the fp32 operands, fp32-grade products */
/* OR-ed into EbenConv1dDesc.math (EBEN_MATH_BF16 | EBEN_LAYOUT_BL, EBEN_MATH_BF16X3 | EBEN_LAYOUT_BL): the layer's activations and
 * gradients are at rest in the bf16 BUNDLE LAYOUT (see "bundle layout" below) and the layer is served by the eben_bl_* entry points;
 * the packed weight images come from eben_conv1d_pack / eben_conv1d_packed_floats on the same descriptor. */
#define EBEN_LAYOUT_BL 0x100

/* One Conv1d / ConvTranspose1d layer (nn.Conv1d / nn.ConvTranspose1d semantics).
 * Replaces the F.conv1d / F.conv_transpose1d call sites behind
 *   vibravox/torch_modules/dnn/eben_generator.py:112-166,241-249,272-280,295-312
 *   vibravox/torch_modules/dnn/eben_discriminator.py:66-157
 *   vibravox/torch_modules/dnn/melgan_discriminator.py:89-156
 * weight layout: Conv1d (Cout, Cin/groups, k); ConvTranspose1d (Cin, Cout/groups, k). */
typedef struct EbenConv1dDesc {
  int32_t batch;
  int32_t c_in, c_out;
  int32_t l_in, l_out;      /* l_out must equal the nn.Module's output length */
  int32_t ksize, stride, dilation, groups;
  int32_t pad_l, pad_r;     /* padding on the input (Conv1d) / `padding` of ConvTranspose1d in pad_l */
  int32_t pad_mode;         /* EBEN_PAD_ZERO | EBEN_PAD_REFLECT (Conv1d only) */
  int32_t transposed;       /* 0 Conv1d, 1 ConvTranspose1d */
  float in_slope;           /* LeakyReLU slope fused on the input load (1.0f = none) */
  float out_slope;          /* LeakyReLU slope fused on the output (1.0f = none) */
  int32_t math;             /* EBEN_MATH_F32 | EBEN_MATH_BF16 */
} EbenConv1dDesc;

EBEN_API const char* eben_last_error(void);
/* Bumped whenever a POD structure, an entry point's signature or a table stride changes (2: EbenWnBwdItem.col_perm_k; 3: eben_rubl_*).
 * eben_version() returns the value the library was built with; bindings compare it with the header they were written against. */
#define EBEN_ABI_VERSION 3
EBEN_API int eben_version(void);
/* fills name with the device's gcnArchName; returns compute-unit count (or negative) */
EBEN_API int eben_device_info(char* name, size_t name_bytes);

/* ---- weight-norm (torch_modules/utils.py:4-9 -> torch._weight_norm, dim=0) -------------- */
/* scale[r] = g[r]/||v[r,:]||, norm[r] = ||v[r,:]||  for r < rows */
EBEN_API int eben_wn_scale(const float* g, const float* v, int rows, int cols, float* scale, float* norm, void* stream);
/* dw given as `nslab` partial slabs of (rows, row_stride) floats (first `cols` of each row are
 * the weight gradient, column `cols` -- if has_bias -- the bias gradient).
 *   g != NULL: dg[r] = sum(dw*v)/norm, dv = (g/norm) dw - (g dg/norm^2) v     (weight-norm)
 *   g == NULL: dv = sum of slabs                                                (plain weight)
 * dbias (nullable) = summed bias column.  The slabs are scratch: slab 0 is overwritten by the sum. */
EBEN_API int eben_wn_bwd(const float* dw_slabs, int nslab, size_t slab_stride, int rows, int cols, int row_stride,
                const float* g, const float* v, const float* norm, float* dg, float* dv, float* dbias, void* stream);

/* Multi-tensor forms of the two calls above: one launch (per 36 items) for every layer of a network.  `items` is a HOST
 * array (device pointers inside), passed to the kernels by value.  Bit-identical results. */
typedef struct EbenWnScaleItem {
  const float* g; const float* v; float* scale; float* norm;
  int32_t rows, cols;
} EbenWnScaleItem;
typedef struct EbenWnBwdItem {
  const float* slabs;                 /* as dw_slabs of eben_wn_bwd: slab 0 is overwritten by the sum */
  const float* g; const float* v; const float* norm;   /* g == NULL: plain weight */
  float* dg; float* dv; float* dbias; /* dbias nullable */
  int64_t slab_stride;                /* floats between slabs */
  int32_t nslab, rows, cols, row_stride;
  int32_t col_perm_k, pad_;           /* k > 0: slab column of weight (c, j) = ((c / 8) k + j) 8 + c % 8, bias column = cols (eben_bl_conv1d_bwd_dw) */
} EbenWnBwdItem;
EBEN_API int eben_wn_scale_multi(const EbenWnScaleItem* items, int n, void* stream);
EBEN_API int eben_wn_bwd_multi(const EbenWnBwdItem* items, int n, void* stream);

/* ---- conv layers ------------------------------------------------------------------------ */
/* floats needed for the packed weights of the forward (which=0) / input-gradient (which=1) pass */
EBEN_API size_t eben_conv1d_packed_floats(const EbenConv1dDesc* d, int which);
/* Many layers' images in few launches (the step rebuilds ~150 images after its two optimiser steps): as eben_conv1d_pack per job
 * (wp_fwd / wp_bwd nullable); `jobs` is a HOST array. */
typedef struct EbenPackJob {
  EbenConv1dDesc desc;
  const float* v; const float* scale;
  float* wp_fwd; float* wp_bwd;
} EbenPackJob;
EBEN_API int eben_conv1d_pack_multi(const EbenPackJob* jobs, int n, void* stream);
/* which tap-conv kernel generation serves the forward (which=0) / input-gradient (which=1) pass of
 * this layer: 1 = tapconv.hip (16x16x4 tiles, any shape), 2 = tapconv2.hip (32x32x2 tiles, LDS-DMA
 * weight stream; deep reductions), 3 = thinconv.hip (VALU), 4 = tapconv3.hip (bf16 operands), 5 = gen_conv.hip (forward only: the
 * generator's strided / transposed / latent convs under EBEN_MATH_BF16X6, eben_generator.py:241-312), 6 = bigtap.hip (bundle layout only: the
 * long reductions of melgan_discriminator.py:119-145 / eben_discriminator.py:118-140 as persistent whole-panel blocks).  The packed layouts differ;
 * pack / fwd / bwd_dx agree by construction. */
EBEN_API int eben_conv1d_kernel_generation(const EbenConv1dDesc* d, int which);
/* pack v (optionally scaled per dim-0 row: weight-norm) into the MFMA-friendly layouts */
EBEN_API int eben_conv1d_pack(const EbenConv1dDesc* d, const float* v, const float* scale, float* wp_fwd, float* wp_bwd, void* stream);
/* y = lrelu_out( conv(lrelu_in(x)) + bias ) [+ residual] */
EBEN_API int eben_conv1d_fwd(const EbenConv1dDesc* d, const float* x, const float* wp_fwd, const float* bias,
                    const float* residual, float* y, void* stream);
/* ... + lrelu(residual, res_slope): the ResidualUnit's skip when its input is the previous block's output seen through
 * the shared LeakyReLU (eben_generator.py:187-189, 314-316: `x + nl(pointwise(dilated(x)))` with x = nl(previous)) */
EBEN_API int eben_conv1d_fwd_res(const EbenConv1dDesc* d, const float* x, const float* wp_fwd, const float* bias,
                        const float* residual, float res_slope, float* y, void* stream);
/* workspace bytes for bwd_dx (reflect padding only) and bwd_dw (partial slabs) */
EBEN_API size_t eben_conv1d_bwd_dx_workspace(const EbenConv1dDesc* d);
EBEN_API size_t eben_conv1d_bwd_dw_workspace(const EbenConv1dDesc* d, int* nslab, int* row_stride);
/* dx = conv^T( dy * lrelu_out'(y) ) * lrelu_in'(x).  y may be NULL when out_slope==1, x when in_slope==1.
 * accumulate!=0 adds into dx instead of overwriting. */
EBEN_API int eben_conv1d_bwd_dx(const EbenConv1dDesc* d, const float* dy, const float* y, const float* wp_bwd,
                       const float* x, float* dx, int accumulate, void* workspace, size_t ws_bytes, void* stream);
/* dx = ( conv^T( dy * lrelu_out'(y) ) + res_pre ) * lrelu_in'(x) + res_post  (either addend may be NULL; res_post only for
 * reflect-padded Conv1d layers, whose fold pass applies it): the gradient joins of the generator's ResidualUnit / skip
 * connections (eben_generator.py:251-254, 282-284, 314-316) without separate add kernels. */
EBEN_API int eben_conv1d_bwd_dx_res(const EbenConv1dDesc* d, const float* dy, const float* y, const float* wp_bwd, const float* x,
                           const float* res_pre, const float* res_post, float* dx, void* workspace, size_t ws_bytes, void* stream);
/* Batched input gradient for several right-hand sides that share one set of saved activations (rows of
 * g / dx = d->batch):  dx[b] = ( conv^T(g[b]) + (b < res_rows ? res[b] : 0) ) * lrelu'(mask[map(b)], mask_slope)
 * with map(b) = seg_map[b / seg] * seg + b % seg (seg = 0: map(b) = b; seg_map = HOST array of 4 ints).
 * g is used as is (the producer applied its own activation derivative); res / mask may be NULL.
 * Replaces the four separate backward passes through DiscriminatorEBENMultiScales that
 * vibravox/lightning_modules/eben.py:99-128,222-240 triggers (three through the enhanced branch, one
 * through the reference branch) with one pass over stacked gradients. */
EBEN_API int eben_conv1d_bwd_dx_ex(const EbenConv1dDesc* d, const float* g, const float* wp_bwd, const float* res, int res_rows,
                          const float* mask, float mask_slope, int seg, const int* seg_map, float* dx, void* stream);
/* The same launch with the feature-matching gradient of the embedding formed in the epilogue (feature_loss.py:40-47, the term
 * |a - b|.sum() / |a|.sum() of one (enhanced, reference) embedding pair) instead of read from a buffer eben_fm_bwd wrote:
 *   dx[b] = ( conv^T(g[b]) + (b < fm_rows ? fm_gs * (sgn(mask[b] - ref[b]) / s2 - s1 * sgn(mask[b]) / s2^2) : 0) )
 *           * lrelu'(mask[map(b)], mask_slope),      (s1, s2) = fm_sums[0..1] (device memory, written by eben_fm_sums)
 * ref = the reference rows of the embedding whose enhanced rows are mask rows 0 .. fm_rows-1.  Only the thin and the bf16 tap-conv
 * kernels (eben_conv1d_kernel_generation(d, 1) == 3 or 4) carry this epilogue; EBEN_EUNSUPPORTED otherwise. */
EBEN_API int eben_conv1d_bwd_dx_fm(const EbenConv1dDesc* d, const float* g, const float* wp_bwd, const float* ref, int fm_rows,
                          const float* fm_sums, float fm_gs, const float* mask, float mask_slope, int seg, const int* seg_map,
                          float* dx, void* stream);
/* partial weight (+bias) gradients into `slabs` (layout reported by bwd_dw_workspace);
 * finish with eben_wn_bwd. */
EBEN_API int eben_conv1d_bwd_dw(const EbenConv1dDesc* d, const float* dy, const float* y, const float* x, int has_bias,
                       float* slabs, size_t ws_bytes, void* stream);

/* ---- bf16 BUNDLE LAYOUT of the discriminator engine ------------------------------------------------------------------------
 * A (batch, channels, length) tensor, channels a multiple of 8, at rest as bf16 [batch][channels / 8][length][8]: one 16-byte UNIT =
 * 8 consecutive channels at one position = the unit the bf16 tap-conv stages into LDS (no conversion: the input tile is a copy) and
 * the unit the weight-gradient kernel transposes with ds_read_b64_tr_b16.  Two planes: hi = bf16(v) (round to nearest even) and lo =
 * bf16(v - hi) (optional; hi + lo carries 16 mantissa bits).  The batched discriminator passes that replace the autograd graph of
 * vibravox/torch_modules/dnn/eben_discriminator.py:27-51,66-163 and melgan_discriminator.py:89-169 keep every embedding and every
 * stacked gradient between the chain heads and the logits in this form when the plan's contractions are bf16 (BASELINE config 2):
 * the values the MFMA sees are the values it saw with fp32 tensors at rest (the same RNE of the same fp32 number, at the producer's
 * store instead of the consumer's load).  Descriptors carry EBEN_LAYOUT_BL in `math`. */
EBEN_API int eben_bl_from_f32(const float* x, int batch, int channels, int length, void* hi, void* lo /* nullable */, void* stream);
EBEN_API int eben_bl_to_f32(const void* hi, const void* lo /* nullable */, int batch, int channels, int length, float* x, void* stream);
/* y = lrelu(conv(x) + bias): x planes (x_lo required for EBEN_MATH_BF16X3: operands hi + lo, three MFMAs per product), y planes
 * (y_lo nullable).  wp_fwd from eben_conv1d_pack on the same descriptor.  EBEN_EUNSUPPORTED for layers outside the bf16 tap-conv. */
EBEN_API int eben_bl_conv1d_fwd(const EbenConv1dDesc* d, const void* x_hi, const void* x_lo, const float* wp_fwd, const float* bias,
                       void* y_hi, void* y_lo, void* stream);
/* Batched input gradient (cf. eben_conv1d_bwd_dx_fm):
 *   dx[b] = ( conv^T(g[b]) + (b < fm_rows ? fm_gs (sgn(a[b] - a[b + ref_row_offset]) / s2 - s1 sgn(a[b]) / s2^2) : 0) )
 *           * lrelu'(a_hi[map(b)], mask_slope),     a = act_hi + act_lo: the saved embedding at the layer's INPUT (2B rows),
 * map(b) = seg_map[b / seg] * seg + b % seg, (s1, s2) = fm_sums[0..1] on the device.  act_hi NULL: no mask, no feature matching. */
EBEN_API int eben_bl_conv1d_bwd_dx(const EbenConv1dDesc* d, const void* g_hi, const float* wp_bwd, const void* act_hi, const void* act_lo,
                          float mask_slope, int seg, const int* seg_map, int fm_rows, int ref_row_offset, const float* fm_sums,
                          float fm_gs, void* dx_hi, void* dx_lo /* nullable */, void* stream);

/* "Phases as rows": the input gradient of a strided bundle-layout Conv1d (stride 4..8 at dilation 1; stride 2 up to dilation 2) as ONE
 * stride-1 Conv1d from the Cout channels of dy to stride x Cin rows, stored depth-to-space (csrc/tapconv.hip; MelGAN layers 1-4,
 * vibravox/torch_modules/dnn/melgan_discriminator.py:97-130, and the stride-2 layers of the PQMF-band discriminators,
 * eben_discriminator.py:86-130).  Rows are (phase, channel) or, at strides 4 and 2, (channel bundle, phase, channel in bundle): a 32-row
 * MFMA tile is then whole 16-byte units at consecutive positions, and mask, feature-matching operands and results move 32 contiguous
 * bytes per lane.  eben_bl_dx_pr_desc writes the descriptor of that primed layer (or returns
 * EBEN_EUNSUPPORTED); eben_bl_dx_pr_weights writes its weights (c_out' x c_in' / groups' x ksize' floats, weight-norm scale folded in), to be
 * packed as the primed layer's FORWARD image with eben_conv1d_pack(primed, w_primed, NULL, image, NULL); eben_bl_conv1d_bwd_dx_pr is
 * eben_bl_conv1d_bwd_dx on that image -- same operands, same epilogue, same results up to the order of the fp32 accumulation. */
EBEN_API int eben_bl_dx_pr_desc(const EbenConv1dDesc* d, EbenConv1dDesc* primed);
EBEN_API int eben_bl_dx_pr_weights(const EbenConv1dDesc* d, const float* v, const float* scale, float* w_primed, void* stream);
EBEN_API int eben_bl_conv1d_bwd_dx_pr(const EbenConv1dDesc* d, const void* g_hi, const float* wp_primed_fwd, const void* act_hi, const void* act_lo,
                                      float mask_slope, int seg, const int* seg_map, int fm_rows, int ref_row_offset, const float* fm_sums,
                                      float fm_gs, void* dx_hi, void* dx_lo, void* stream);
/* eben_bl_conv1d_bwd_dx / eben_bl_conv1d_bwd_dx_pr with the feature-matching code plane of the embedding (eben_bl_fm_sums_codes; NULL: the
 * forms above): the rows with a feature-matching term read act_hi and the codes instead of act_hi, act_lo and both planes of the
 * reference rows.  Bit-identical results. */
EBEN_API int eben_bl_conv1d_bwd_dx_c(const EbenConv1dDesc* d, const void* g_hi, const float* wp_bwd, const void* act_hi, const void* act_lo,
                            const void* fm_codes, float mask_slope, int seg, const int* seg_map, int fm_rows, int ref_row_offset,
                            const float* fm_sums, float fm_gs, void* dx_hi, void* dx_lo /* nullable */, void* stream);
EBEN_API int eben_bl_conv1d_bwd_dx_pr_c(const EbenConv1dDesc* d, const void* g_hi, const float* wp_primed_fwd, const void* act_hi, const void* act_lo,
                               const void* fm_codes, float mask_slope, int seg, const int* seg_map, int fm_rows, int ref_row_offset,
                               const float* fm_sums, float fm_gs, void* dx_hi, void* dx_lo /* nullable */, void* stream);
/* Weight (+ bias) gradient with both operands in the bundle layout (csrc/bl_dw.hip): the reduction runs along (batch, time) on
 * v_mfma_f32_32x32x16_bf16, both operand tiles arrive in LDS by buffer_load ... lds (descriptor bounds = the zero padding) and are
 * transposed on read -- no packing pre-pass, no conversion.  Slabs [nslab][c_out][row_stride] as eben_conv1d_bwd_dw's, except that
 * with *col_perm_k = k > 0 the column of weight (c, j) inside a row is ((c / 8) k + j) 8 + c % 8 (EbenWnBwdItem.col_perm_k). */
EBEN_API size_t eben_bl_conv1d_bwd_dw_workspace(const EbenConv1dDesc* d, int* nslab, int* row_stride, int* col_perm_k);
EBEN_API int eben_bl_conv1d_bwd_dw(const EbenConv1dDesc* d, const void* dy_hi, const void* x_hi, int has_bias, float* slabs, size_t ws_bytes,
                          void* stream);
/* n weight gradients in as few launches as their tile shapes allow (consecutive problems of one shape share a launch): the same layer
 * index of the three PQMF-band discriminators (vibravox/torch_modules/dnn/eben_discriminator.py:27-28 -- same channels and taps, their own
 * dilation, length, operands and slabs).  Problem i is exactly eben_bl_conv1d_bwd_dw(descs[i], dy_hi[i], x_hi[i], has_bias, slabs[i],
 * ws_bytes[i]): same slabs, bit for bit. */
EBEN_API int eben_bl_conv1d_bwd_dw_multi(const EbenConv1dDesc* const* descs, const void* const* dy_hi, const void* const* x_hi, int has_bias,
                                float* const* slabs, const size_t* ws_bytes, int n, void* stream);
/* Chain heads: ReflectionPad1d(reflect_pad) + Conv1d(c_in -> c_out, ksize, dilation, groups = c_in, zero padding `pad`, stride 1)
 * + bias + LeakyReLU(out_slope), fp32 (batch, c_in, l_in) in, bundle planes out (eben_discriminator.py:66-76 layer 0,
 * melgan_discriminator.py:89-98 layer 0).  `jobs` is a HOST array of up to 4 heads run by ONE launch (the three PQMF-band chains
 * read the same bands): forward; input gradient dx (rows, c_in, l_in) = SUM over the jobs of conv^T(g) folded through the
 * reflection (y_hi / y_lo = the gradient planes at the heads' outputs, c_in <= 4); weight gradient of one head (slabs
 * [nslab][c_out][ksize + 1], column ksize = bias) from `rows` gradient rows y_hi and the `rows` input rows x paired with them. */
typedef struct EbenBlHeadJob {
  const float* x;                      /* (batch, c_in, l_in) */
  const float* v; const float* scale;  /* weight direction (c_out, 1, ksize), weight-norm scale g / ||v|| per row (nullable) */
  const float* bias;                   /* nullable */
  void* y_hi; void* y_lo;              /* (batch, c_out, l_out) planes; y_lo nullable */
  int32_t c_in, c_out, l_in, l_out, ksize, dilation, pad, reflect_pad;
  float out_slope; int32_t pad_;
} EbenBlHeadJob;
EBEN_API int eben_bl_head_fwd(const EbenBlHeadJob* jobs, int njobs, int batch, void* stream);
EBEN_API int eben_bl_head_dx(const EbenBlHeadJob* jobs, int njobs, int rows, float* dx, void* stream);
EBEN_API size_t eben_bl_head_dw_workspace(const EbenBlHeadJob* job, int rows, int* nslab, int* row_stride);
EBEN_API int eben_bl_head_dw(const EbenBlHeadJob* job, int rows, float* slabs, size_t ws_bytes, void* stream);
/* Chain tails: the logits layer Conv1d(channels -> 1, ksize <= 8, zero padding, stride 1) (eben_discriminator.py:150-157,
 * melgan_discriminator.py:147-156), v (1, channels, ksize), scale / bias one element (nullable): forward from bundle planes to fp32
 * (batch, 1, l_out); input gradient of `rows` stacked seeds (rows, 1, l_out) with the epilogue of eben_bl_conv1d_bwd_dx (act = the
 * layer's input embedding); weight gradient of `nbranch` hinge branches in one launch -- branch br pairs seed rows and input rows
 * [br rows, (br + 1) rows) of the given pointers -- into slabs [nbranch][nslab][channels ksize + 1] (last column = bias).  The built
 * head shapes are the two of DiscriminatorEBENMultiScales (4 -> 24 k 3, 1 -> 16 k 15): EBEN_EUNSUPPORTED otherwise. */
EBEN_API int eben_bl_tail_fwd(const void* x_hi, const void* x_lo, int batch, int channels, int length, int ksize, int pad, const float* v,
                     const float* scale, const float* bias, float out_slope, float* y, void* stream);
EBEN_API int eben_bl_tail_dx(const float* seeds, int rows, int channels, int length, int ksize, int pad, const float* v, const float* scale,
                    const void* act_hi, const void* act_lo, float mask_slope, int seg, const int* seg_map, int fm_rows,
                    int ref_row_offset, const float* fm_sums, float fm_gs, void* g_hi, void* g_lo /* nullable */, void* stream);
EBEN_API size_t eben_bl_tail_dw_workspace(int rows, int channels, int l_out, int ksize, int nbranch, int* nslab, int* row_stride);
EBEN_API int eben_bl_tail_dw(const float* seeds, const void* x_hi, const void* x_lo, int rows, int nbranch, int channels, int length, int ksize, int pad,
                    float* slabs, size_t ws_bytes, void* stream);
/* feature-matching sums (eben_fm_sums) over embeddings in bundle planes: planes = HOST array (hi_0, lo_0, hi_1, lo_1, ...), the
 * enhanced rows are the first units[i] units of pair i's planes and the reference rows the next units[i] (a = hi + lo). */
EBEN_API size_t eben_bl_fm_sums_workspace(int npairs);
EBEN_API int eben_bl_fm_sums(const void* const* planes, const int64_t* units, int npairs, float* partial_ws, size_t ws_bytes, float* sums,
                    void* stream);
/* eben_bl_fm_sums that also writes, for pairs whose codes[p] is not NULL, one byte per element of the pair's enhanced rows:
 * bits 0-1 = sgn(a - r) + 1, bits 2-3 = sgn(a) + 1 (8 bytes per unit, at the unit's index in the hi plane) -- what the feature-matching
 * term of the stacked input gradients needs of the pair (eben_bl_conv1d_bwd_dx_c: one 4-byte load per row quad in place of three
 * 8-byte operand loads).  Same sums. */
EBEN_API int eben_bl_fm_sums_codes(const void* const* planes, const int64_t* units, void* const* codes, int npairs, float* partial_ws,
                                   size_t ws_bytes, float* sums, void* stream);

/* ---- fused ResidualUnit forward (vibravox/torch_modules/dnn/eben_generator.py:287-316) ---------------------------------
 *   y = xin + lrelu( W_pw . ( W_dil (*) xin ), out_slope ),  xin = lrelu(x, in_slope)
 * W_dil (C, C, 3) dilation d with "same" reflect padding, W_pw (C, C, 1), C in {32, 64, 128}, no bias; x, y (batch, C, length).
 * One launch instead of dilated conv + pointwise conv + add: x is read once.  h (nullable) receives W_dil (*) xin and u
 * (nullable) lrelu(z) -- what the backward needs (the pointwise weight gradient, the activation mask).  eben_ru_pack writes
 * the weight image both stages stream (eben_ru_packed_floats(C) floats) from the directions v and the weight-norm scales
 * g/||v|| (eben_wn_scale; NULL = plain weights). */
EBEN_API size_t eben_ru_packed_floats(int channels);
EBEN_API int eben_ru_pack(int channels, const float* v_dil, const float* scale_dil, const float* v_pw, const float* scale_pw, float* wimg,
                 void* stream);
EBEN_API int eben_ru_fwd(int batch, int channels, int length, int dilation, const float* x, float in_slope, float out_slope,
                const float* wimg, float* y, float* h, float* u, void* stream);
/* Input-gradient chain of the same unit in one launch (the reflect padding folds onto the end positions inside the kernel):
 *   g_h = W_pw^T ( g_y * lrelu'(u, out_slope) ),   g_x = ( g_y + fold(W_dil^T (*) g_h) ) * lrelu'(x, in_slope) + post
 * u: the forward's lrelu(z); x: the unit's input (needed when in_slope != 1), post: nullable addend behind the mask (a skip
 * connection's gradient); g_h is written out for the dilated conv's weight gradient.  eben_ru_pack_bwd writes the transposed
 * weight image (eben_ru_packed_floats(C) floats). */
EBEN_API int eben_ru_pack_bwd(int channels, const float* v_dil, const float* scale_dil, const float* v_pw, const float* scale_pw,
                     float* wimg_bwd, void* stream);
EBEN_API int eben_ru_bwd(int batch, int channels, int length, int dilation, const float* gy, const float* u, float out_slope,
                const float* x, float in_slope, const float* post, const float* wimg_bwd, float* gx, float* gh, void* stream);

/* The same two launches on the bf16 matrix pipe with SPLIT operands (csrc/ru_split.hip): every fp32 operand enters as a sum of
 * bf16 pieces and a product is the sum of the piece products that matter.  math:
 *   EBEN_MATH_F32     the exact-fp32 MFMA kernels above;
 *   EBEN_MATH_BF16X6  three pieces per operand, six products: all 24 mantissa bits, dropped terms <= 2^-26 -- fp32 arithmetic at
 *                     6/16 of the fp32 MFMA cost (the generator forward's mode);
 *   EBEN_MATH_BF16X3  two pieces, three products, ~2^-17 per product;   EBEN_MATH_BF16  plain bf16 operands.
 * which: 0 forward image, 1 backward (transposed) image; eben_ru_packed_floats_ex(C, math) floats each.  Dilation <= 9. */
EBEN_API size_t eben_ru_packed_floats_ex(int channels, int math);
EBEN_API int eben_ru_supported(int channels, int dilation, int math);
EBEN_API int eben_ru_pack_ex(int channels, int math, int which, const float* v_dil, const float* scale_dil, const float* v_pw,
                    const float* scale_pw, float* wimg, void* stream);
typedef struct EbenRuPackJob {   /* one image of eben_ru_pack_ex (math != EBEN_MATH_F32); `jobs` is a HOST array */
  int32_t channels, math, which, pad_;
  const float* v_dil; const float* scale_dil; const float* v_pw; const float* scale_pw; float* wimg;
} EbenRuPackJob;
EBEN_API int eben_ru_pack_multi(const EbenRuPackJob* jobs, int n, void* stream);
EBEN_API int eben_ru_fwd_ex(int math, int batch, int channels, int length, int dilation, const float* x, float in_slope, float out_slope,
                   const float* wimg, float* y, float* h, float* u, void* stream);
EBEN_API int eben_ru_bwd_ex(int math, int batch, int channels, int length, int dilation, const float* gy, const float* u, float out_slope,
                   const float* x, float in_slope, const float* post, const float* wimg_bwd, float* gx, float* gh, void* stream);

/* Both weight gradients of the unit in one launch (csrc/ru_dw.hip): split-K slabs [eben_ru_dw_slabs(batch, C, length)][C][C] for the
 * pointwise conv and [..][C][3 C] for the dilated one (row strides C and 3 C), to be summed by eben_wn_bwd / eben_wn_bwd_multi:
 *   dW_pw[m][c] = sum g_z[m] h[c],  g_z = g_y * lrelu'(u, out_slope);   dW_dil[m][c][j] = sum g_h[m] xin[c](. + (j - 1) d, reflected)
 * The reduction runs along time, so the operands are read straight from the (batch, channel, time) tensors (no packing pass).
 * math: EBEN_MATH_BF16 (single bf16 operands) or EBEN_MATH_BF16X6 (three pieces per operand: fp32-grade). */
EBEN_API int eben_ru_dw_slabs(int batch, int channels, int length);
EBEN_API int eben_ru_dw(int math, int batch, int channels, int length, int dilation, const float* gy, const float* u, float out_slope,
               const float* h, const float* gh, const float* x, float in_slope, float* slabs_pw, float* slabs_dil, void* stream);

/* ResidualUnit with what the backward needs AT REST AS bf16 BUNDLES (csrc/ru_bl.hip; the bf16 generator backward of
 * vibravox/torch_modules/dnn/eben_generator.py:287-316).  Bundle plane of a (batch, C, L) tensor: bf16 [batch][C / 8][L][8] (2 batch C L
 * bytes); sign plane: [batch][C / 8][L] bytes, bit e = (value of channel 8 g + e > 0).
 *   eben_rubl_fwd  = eben_ru_fwd_ex (math EBEN_MATH_BF16X6) that writes y (fp32) and, instead of h / u in fp32: xb = bf16(lrelu(x)),
 *                    hb = bf16(h) as bundle planes and umask = the sign plane of u.
 *   eben_rubl_bwd  = eben_ru_bwd_ex (math EBEN_MATH_BF16; wimg_bwd = the EBEN_MATH_BF16 image of eben_ru_pack_ex(which = 1)): g_y fp32 in,
 *                    g_x fp32 out; gzb = bf16(g_y lrelu'(u)) and ghb = bf16(g_h) leave as bundle planes (operands of eben_rubl_dw only).
 *   eben_rubl_dw   = eben_ru_dw on the four bundle planes: slabs [eben_rubl_dw_slabs(batch, C, L)][C][C] and [..][C][3 C] in
 *                    eben_ru_dw's layout. */
EBEN_API int eben_rubl_supported(int channels, int dilation);
EBEN_API int eben_rubl_fwd(int math, int batch, int channels, int length, int dilation, const float* x, float in_slope, float out_slope,
                           const float* wimg, float* y, void* xb, void* hb, void* umask, void* stream);
EBEN_API int eben_rubl_bwd(int batch, int channels, int length, int dilation, const float* gy, const void* umask, float out_slope, const float* x,
                           float in_slope, const float* post, const float* wimg_bwd, float* gx, void* gzb, void* ghb, void* stream);
EBEN_API int eben_rubl_dw_slabs(int batch, int channels, int length);
EBEN_API int eben_rubl_dw(int batch, int channels, int length, int dilation, const void* gzb, const void* hb, const void* ghb, const void* xb,
                          float* slabs_pw, float* slabs_dil, void* stream);

/* ---- PQMF (vibravox/torch_modules/dsp/pqmf.py:194-213, eben_generator.py:209-211) ---------- */
/* decimating FIR bank: y[b,k,t] = sum_j w[k*ntaps+j] * x[b,0,t*stride+off0+j], zero outside [0,lx) */
EBEN_API int eben_fir_decimate(const float* x, const float* w, float* y, int batch, int lx, int ly, int bands, int ntaps,
                      int stride, int off0, void* stream);
/* interpolating FIR bank summed over bands (adjoint of the above):
 *   x[b,0,u] (+)= sum_k sum_{t,j : t*stride+off0+j == u} w[k*ntaps+j] * y[b,k,t] */
EBEN_API int eben_fir_interp_sum(const float* y, const float* w, float* x, int batch, int lx, int ly, int bands, int ntaps,
                        int stride, int off0, void* stream);

/* ---- elementwise ------------------------------------------------------------------------ */
EBEN_API int eben_lrelu_fwd(const float* x, float* y, size_t n, float slope, void* stream);
EBEN_API int eben_lrelu_bwd(const float* dy, const float* x_or_y, float* dx, size_t n, float slope, void* stream);
/* Grouped GEMM y[g] = W[g] (m x k) * x[g] (k x n), row-major with the n columns contiguous, fp32 in / out, products of split bf16
 * operands on the bf16 MFMA (math: EBEN_MATH_BF16, EBEN_MATH_BF16X3 = hi + lo operands / three products, EBEN_MATH_BF16X6 = three
 * pieces / six products, fp32-grade).  The windowed-DFT contractions of auraloss' STFT (torch.stft inside
 * auraloss.freq.STFTLoss.stft, configs/lightning_module/loss_module/multi_stft.yaml:1-18) and their transposes in the backward:
 * groups = 2 is the folded even / odd form.  w is packed once per basis (eben_gemm_pack -> eben_gemm_packed_floats floats). */
EBEN_API size_t eben_gemm_packed_floats(int math, int groups, int m, int k);
EBEN_API int eben_gemm_pack(int math, int groups, int m, int k, const float* w, float* wp, void* stream);
EBEN_API int eben_gemm_fwd(int math, int groups, int m, int k, long long n, const float* x, const float* wp, float* y, void* stream);
/* Space to depth along time: out[row][r][q] = xp[row][S*q + r + off] (r < S, q < Lq; xp = x continued by reflection or by zeros
 * beyond [0, L)), optionally times lrelu'(mask[row][.], mask_slope).  out: (rows, S, Lq).  Turns a stride-S Conv1d with
 * ksize = kq*S (EncBlock.conv, eben_generator.py:251-254) -- or the input gradient of the matching ConvTranspose1d (DecBlock,
 * eben_generator.py:282-284) -- into a stride-1 conv with kq taps over C*S channels, with the weights viewed as
 * (M, C, kq, S) -> (M, C, S, kq). */
EBEN_API int eben_space_to_depth(const float* x, const float* mask, float mask_slope, float* out, int rows, int L, int S, int off, int Lq,
                        int reflect, void* stream);
EBEN_API int eben_add(const float* a, const float* b, float* out, size_t n, void* stream);
EBEN_API int eben_axpby(const float* a, float alpha, const float* b, float beta, float* out, size_t n, void* stream);
/* bands = tanh(x + lift) where lift has `c_lift` leading channels (eben_generator.py:203-208) */
EBEN_API int eben_tanh_lift_fwd(const float* x, const float* lift, float* out, int batch, int channels, int c_lift, int length, void* stream);
EBEN_API int eben_tanh_bwd(const float* dout, const float* out, float* dx, size_t n, void* stream);
/* nn.ReflectionPad1d (eben_discriminator.py:69, melgan_discriminator.py:92) and its adjoint */
EBEN_API int eben_reflect_pad_fwd(const float* x, float* y, int rows, int lx, int pad_l, int pad_r, void* stream);
EBEN_API int eben_reflect_pad_bwd(const float* dy, float* dx, int rows, int lx, int pad_l, int pad_r, void* stream);

/* ---- losses ----------------------------------------------------------------------------- */
/* feature matching (vibravox/torch_modules/losses/feature_loss.py:37-50): per pair i
 * sums[2i] = sum|a-b|, sums[2i+1] = sum|a| (device floats).  ptrs = HOST array of 2*npairs device
 * pointers (a0,b0,a1,b1,...), numel = HOST array of npairs element counts; they travel to the
 * kernels by value in the kernel-argument segment (no device table, no copy). */
EBEN_API int eben_fm_sums(const void* const* ptrs, const int64_t* numel, int npairs, float* partial_ws, size_t ws_bytes,
                 float* sums, void* stream);
EBEN_API size_t eben_fm_sums_workspace(int npairs);
/* da_i = gout * inv_count * ( sign(a-b)/S2 - S1*sign(a)/S2^2 ); gout is a device scalar */
EBEN_API int eben_fm_bwd(const void* const* ptrs, void* const* da_ptrs, const int64_t* numel, int npairs, const float* sums,
                const float* gout, float inv_count, void* stream);
/* hinge (losses/hinge_loss.py:35-43): out[i] = mean(relu(1 - target*x_i)) ; bwd adds nothing else */
EBEN_API int eben_hinge_fwd(const float* x, size_t n, float target, float* out, void* stream);
/* Dynamic loss balancing of the generator step (vibravox/lightning_modules/eben.py:222-240), the scalar part in one launch:
 * old[i] = init ? *norms[i] : old[i];  if (ema) old[i] = beta * old[i] + one_minus_beta * *norms[i];
 * lambdas[i] = clamp(1 / (old[i] + 1e-4), 0, 1e4);  backprop[0] = sum_i *losses[i] * lambdas[i]   (torch's operation order and
 * roundings; norms / losses: n <= 8 device scalars), and the lambda-weighted sum of n tensors out = sum_i weights[i] * tensors[i]
 * (weights on the device) that seeds the generator backward. */
/* The norms ||dL_i / d(last_conv.weight)|| of the dynamic loss balancing (vibravox/lightning_modules/eben.py:222-229) for n <= 4 losses in
 * one pass, from the seeds s_i = dL_i / d(bands): dW_i = sum s_i (1 - bands^2) (*) reflect_pad(pre), bands = tanh(last_conv(pre) + lift)
 * (eben_generator.py:159-166, 203-208; built for that layer: 32 -> 4, k 3, reflect padding 1).  seeds: HOST array of n device pointers to
 * (batch, 4, length) tensors; norms: n floats on the device; workspace: eben_last_conv_norms_workspace(batch) bytes.  Fixed-order sums. */
EBEN_API size_t eben_last_conv_norms_workspace(int batch);
EBEN_API int eben_last_conv_norms(const void* const* seeds, int n, const float* bands, const float* pre, int batch, int c_in, int c_out, int length,
                                  int ksize, int pad, float* workspace, size_t ws_bytes, float* norms, void* stream);
EBEN_API int eben_balance(const void* const* norms, const void* const* losses, int n, float* old, int init, int ema, float beta,
                 float one_minus_beta, float* lambdas, float* backprop, void* stream);
EBEN_API int eben_weighted_sum(const void* const* tensors, const float* weights, int n, size_t numel, float* out, void* stream);
/* The four discriminator-side loss values of a step from their partial results, one launch: out[0] = inv_count * sum_p s1_p / s2_p
 * (feature_loss.py:37-50, (s1, s2) pairs of eben_fm_sums), out[1 + k] = mean over the sub-discriminators of hinge[3 i + k]
 * (eben.py:99-128: generator-side adversarial, fake, real). */
EBEN_API int eben_disc_losses(const float* fm_sums, int npairs, float inv_count, const float* hinge, int nchains, float* out, void* stream);
/* The MRSTFT loss value from the per-row sums of eben_stft_loss_sums_ex of n <= 8 resolutions (auraloss MultiResolutionSTFTLoss:
 * mean over resolutions of  mean_r sqrt(s0 / s1) + sum_r s2 * inv_counts[i],  inv_counts[i] = 1 / (rows bins_i frames_i)). */
EBEN_API int eben_stft_loss_total(const void* const* sums, const float* inv_counts, int n, int rows, float* out, void* stream);
/* n <= 32 hinge terms in one launch: out[i] = mean(max(0, 1 - targets[i] * xs[i][.])) (the 3 targets x 4 sub-discriminators of one
 * step, eben.py:99-128 through hinge_loss.py:35-43); the single-term kernel's summation order. */
EBEN_API int eben_hinge_fwd_multi(const void* const* xs, const int64_t* numel, const float* targets, int n, float* out, void* stream);
EBEN_API int eben_hinge_bwd(const float* x, size_t n, float target, const float* gout, float scale, float* dx, void* stream);
/* The stacked seeds of the batched discriminator backward (hinge_loss.py:38-43 differentiated three times; rows [fm | adv | fake | real],
 * `per` logits each): seeds[0 .. per) = 0, seeds[per ..) = d hinge(enhanced, +1), [2 per ..) = d hinge(enhanced, -1), [3 per ..) =
 * d hinge(reference, +1), each times gout[0] * scale_x / per -- eben_hinge_bwd's arithmetic, one launch. */
EBEN_API int eben_hinge_bwd_stacked(const float* enhanced_logits, const float* reference_logits, size_t per, const float* gout, float scale_adv,
                                    float scale_fake, float scale_real, float* seeds, void* stream);
/* STFT magnitude losses (auraloss.freq.STFTLoss as configured by multi_stft.yaml; call site
 * vibravox/lightning_modules/eben.py:195-198).  spec_* hold (rows, 2*bins_pad, frames) with re in
 * channels [0,bins) and im in [bins_pad, bins_pad+bins); |.| = sqrt(clamp(re^2+im^2, eps)).
 * per row r: out[3r] = sum (|Y|-|X|)^2, out[3r+1] = sum |Y|^2, out[3r+2] = sum |log|X| - log|Y|| */
EBEN_API int eben_stft_loss_sums(const float* spec_x, const float* spec_y, int rows, int bins, int bins_pad, int frames,
                        float eps, float* partial_ws, size_t ws_bytes, float* out, void* stream);
EBEN_API size_t eben_stft_loss_sums_workspace(int rows);
/* dspec_x from d(loss) with loss = mean_rows(sqrt(s0/s1)) + mean(|log X - log Y|); gout device scalar * scale */
EBEN_API int eben_stft_loss_bwd(const float* spec_x, const float* spec_y, int rows, int bins, int bins_pad, int frames,
                       float eps, const float* sums, const float* gout, float scale, float* dspec_x, void* stream);

/* adjoint of framing a (reflect-)padded signal: x[b,u] (+)= fold( sum_f buf[b, u+pad-f*hop, f] ) with
 * buf (batch, win, frames).  Used by the STFT backward: d(frames) = basis^T . d(spec) is a dense
 * pointwise conv (eben_conv1d_fwd, ksize 1), this kernel scatters it back onto the waveform. */
EBEN_API int eben_overlap_add(const float* frames_buf, float* x, int batch, int lx, int win, int frames, int hop, int pad,
                     int reflect, int accumulate, void* stream);

/* The same three with explicit strides -- element (row r, bin k, frame f) of a spectrum at r*row_stride + k*bin_stride + f
 * (imaginary part at + im_off), frame-buffer element (item b, window sample j, frame f) at b*row_stride + j*j_stride + f --
 * so that the windowed DFT of ALL items runs as one dense GEMM over a flat (channels, rows*frames) matrix:
 *   eben_stft_frames: out[j, r*frames + f] = sig[r, reflect(f*hop + j - pad)]   (torch.stft(center=True, pad_mode="reflect") framing)
 *   spec (2*bins, R*frames) = basis (2*bins, win) . frames (win, R*frames)        (eben_conv1d_fwd, ksize 1, batch 1) */
EBEN_API int eben_stft_frames(const float* sig, float* out, int rows, int t, int win, int hop, int pad, int frames, void* stream);
/* Folded framing for a window symmetric about frame sample win/2 (hann): out holds TWO GROUPS of nsub*win/2 rows, the even
 * part E[m] = s[h+m] + s[h-m] (E[0] = s[h]) and the odd part O[m] = s[h+m] - s[h-m] (O[0] = 0) of each frame about h = win/2, so
 * that Re X = basis[:bins, h:] . E and Im X = basis[bins:, h:] . O is a 2-group pointwise conv with half the products of the
 * dense one (auraloss STFTLoss.stft = torch.stft(center=True, hann) as called from eben.py:195-198).  split = 1: nsub = 3, each
 * group written as [hi ; lo ; hi] (hi = bf16(v), lo = bf16(v - hi)) for a bf16-operand contraction against [W_hi ; W_hi ; W_lo]
 * ("bf16x3", ~2^-17 relative); split = 0: nsub = 1, exact fp32.  eben_split3 does the same to the rows of a gradient matrix
 * (groups * R, cols) -> (groups * 3R, cols); eben_overlap_add_folded is the adjoint of the folded framing: buf rows [dE ; dO]. */
EBEN_API int eben_stft_frames_folded(const float* sig, float* out, int rows, int t, int win, int hop, int pad, int frames, int split,
                            void* stream);
EBEN_API int eben_split3(const float* in, float* out, int groups, int rows_per_group, long long cols, void* stream);
EBEN_API int eben_overlap_add_folded(const float* frames_buf, float* x, int batch, int lx, int win, int frames, int hop, int pad,
                            int accumulate, long long row_stride, long long j_stride, void* stream);
EBEN_API int eben_stft_loss_sums_ex(const float* spec_x, const float* spec_y, int rows, int bins, int frames, long long row_stride,
                           long long bin_stride, long long im_off, float eps, float* partial_ws, size_t ws_bytes, float* out,
                           void* stream);
EBEN_API int eben_stft_loss_bwd_ex(const float* spec_x, const float* spec_y, int rows, int bins, int frames, long long row_stride,
                          long long bin_stride, long long im_off, float eps, const float* sums, const float* gout, float scale,
                          float* dspec_x, long long out_row_stride, long long out_bin_stride, long long out_im_off, void* stream);
EBEN_API int eben_overlap_add_ex(const float* frames_buf, float* x, int batch, int lx, int win, int frames, int hop, int pad,
                        int reflect, int accumulate, long long row_stride, long long j_stride, void* stream);

/* ---- optimiser (torch.optim.Adam as configured by configs/lightning_module/optimizer/adam.yaml) --- */
typedef struct EbenAdamTensor {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  int64_t numel;
} EbenAdamTensor;
/* `table` is a HOST array of ntensors entries (device pointers inside); it is passed to the
 * kernel by value, 48 tensors per launch. */
EBEN_API int eben_adam_step(const EbenAdamTensor* table, int ntensors, int64_t max_numel, float lr, float beta1, float beta2,
                   float eps, float weight_decay, int step, float grad_scale, void* stream);

/* ---- noisy-BWE batch assembly (vibravox/lightning_datamodules/noisybwe.py:219-291; utils.py:7-81,195-254) ----
 * per item:  body_conducted[i, 0, t] = speech[u] + noise[noise_start + u],  airborne[i, 0, t] = airborne_clip[u],
 * u = t + shift, zero outside [0, length).  shift >= 0: crop offset (set_audio_duration); shift < 0: left zero run
 * of pad_audio; noise / airborne_clip may be NULL (real noisy data: padding only).  `items` is a HOST array (device
 * pointers inside), passed to the kernel by value; outputs are (nitems, 1, samples). */
typedef struct EbenCollateItem {
  const float* speech;
  const float* airborne;
  const float* noise;
  int64_t length;       /* samples of speech (== airborne) */
  int64_t noise_start;  /* first noise sample mixed in */
  int64_t shift;
} EbenCollateItem;
EBEN_API int eben_noisy_collate(const EbenCollateItem* items, int nitems, int samples, float* body_conducted, float* airborne, void* stream);

/* ---- waveform augmentation (vibravox/torch_modules/dsp/data_augmentation.py:38-71) -------------------------------
 * time masking (dsp/time_masking_waveform.py:18-36): x[r, first : first+count] = 0 in place for r < rows;
 * polyphase windowed-sinc resampling (torchaudio.functional.resample as used by T.SpeedPerturbation, restated):
 *   out[r, q*nw + p] = sum_{j < 2*width+orig} kernels[p, j] * xpad[r, q*orig + j], xpad = x with `width` zeros in front;
 *   kernels (nw, 2*width+orig) fp32 on the device; t_out <= ceil(nw * t_in / orig). */
EBEN_API int eben_time_mask(float* x, long long rows, int t, int first, int count, void* stream);
/* phase vocoder of T.PitchShift (torchaudio.functional.phase_vocoder, restated): spec / out are flat (2*bins, rows*frames) /
 * (2*bins, rows*frames_out) spectra (re rows, then im rows), frames_out = ceil(frames / rate), hop = STFT hop length */
EBEN_API int eben_phase_vocoder(const float* spec, float* out, int rows, int bins, int frames, int frames_out, double rate, float hop,
                       void* stream);
EBEN_API int eben_resample(const float* x, const float* kernels, float* out, int rows, int t_in, int t_out, int orig, int nw,
                  int width, void* stream);

/* ---- misc ------------------------------------------------------------------------------- */
/* out[0] = sqrt(sum x^2) (torch.norm at eben.py:226); `out` must hold 257 floats (scratch) */
EBEN_API int eben_l2norm(const float* x, size_t n, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EBEN_HIP_H */
