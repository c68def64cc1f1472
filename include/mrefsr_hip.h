/*
 * mrefsr_hip.h -- C ABI of libmrefsr_hip.so: the MI355X (gfx950) kernels of the MRefSR
 * multi-reference matching-and-reconstruction hot path.
 *
 * Conventions
 *   - plain C, no torch / ATen types: raw DEVICE pointers + sizes + a hipStream_t passed as void*.
 *   - every function enqueues work on `stream` and returns immediately: 0 = ok, <0 = MREFSR_E_*.
 *     mrefsr_last_error() gives the message for the calling thread.  Nothing allocates device
 *     memory; scratch is caller-provided (size queries are separate entry points).
 *   - thread-safe: no mutable globals besides the thread-local error string.
 *   - tensors are dense row-major fp32 unless stated; "HW" = H*W.
 *
 * Each entry cites the reference interface it replaces (paths relative to the reference repo).
 */
#ifndef MREFSR_HIP_H
#define MREFSR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MREFSR_ABI_VERSION 1

enum {
    MREFSR_OK = 0,
    MREFSR_E_INVALID = -1,     /* bad argument (null pointer, non-positive size, unsupported combo) */
    MREFSR_E_UNSUPPORTED = -2, /* shape outside what the kernels implement (message says which)    */
    MREFSR_E_LAUNCH = -3       /* hipGetLastError() after the launch                               */
};

typedef void *mrefsr_stream_t; /* hipStream_t */

int mrefsr_abi_version(void);
const char *mrefsr_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Correlation path: basicsr/archs/corres_generation_arch.py:53-68 + basicsr/archs/ref_map_util.py
 * --------------------------------------------------------------------------------------------- */

/* Padded channel count of the pixel-major feature layout used by the correlation kernel
 * (multiple of 64).  Returns <0 if C is unsupported (C > 256). */
int mrefsr_corr_padded_channels(int C);

/* Per-pixel channel L2 normalisation + transposition to the pixel-major "split" layout.
 * Replaces F.normalize(feat.reshape(c,-1), dim=0)  (corres_generation_arch.py:57-59) and the
 * unfold of sample_patches (ref_map_util.py:4-23, never materialised here).
 *   x  [N][C][HW]           raw features (VGG16 conv3_1); x_nhwc = 1: [N][HW][C] fp32; x_nhwc = 2: [N][HW][C] bf16 (the
 *                           2-byte activation storage of BASELINE configs[4], passed through the same pointer)
 *   y  [N][HW][Cp]          Cp = mrefsr_corr_padded_channels(C); element (p, c) lives at
 *                           p*Cp + (c&1)*(Cp/2) + (c>>1); channels >= C are zero
 *   n2 [N][HW]              sum over c of y^2 (fmaf chain, c ascending)
 *   ybf                     NULL, or the operand of the pre-filter correlation pass, natural channel order:
 *                           ybf_fmt 0: [N][HW][2][Cp] bf16, two-term split y = hi + lo (hi = bf16(y), lo = bf16(y - hi))
 *                           ybf_fmt 1: [N][HW][Cp] fp16, yh = fp16(y)
 *   normalize               1: y = x / max(||x||, 1e-12) (the path); 0: y = x (layout change only,
 *                           for callers of feature_match_index that pass un-normalised maps)
 *   d2 [N][HW]              NULL, or sum over c of (y - fp16(y))^2: the squared norm of the rounding error of the
 *                           fp16 operand, from which the caller derives the pre-filter window (`tau` below)     */
int mrefsr_pixnorm_f32(const float *x, float *y, float *n2, void *ybf, int N, int C, int HW,
                       int normalize, int x_nhwc, int ybf_fmt, float *d2, mrefsr_stream_t stream);

/* 3x3 patch norms: batch.norm(p=2, dim=(0,1,2)) + 1e-5  (ref_map_util.py:62-63, :79-80).
 *   n2 [N][h][w] -> nrm_eps [N][h-2][w-2] = sqrt(sum of 9) + 1e-5 ; inv = 1 / nrm_eps
 * either output may be NULL. */
int mrefsr_patch_norm_f32(const float *n2, float *nrm_eps, float *inv, int N, int h, int w,
                          mrefsr_stream_t stream);

/* Fused 3x3-patch correlation + top-1:  feature_match_index(feat_in, feat_ref, patch_size=3,
 * input_stride=1, ref_stride=1, is_norm=True, norm_input=True)  (ref_map_util.py:26-86); the
 * (n_ref_patches x n_query) correlation matrix (:64-67) is never written to memory.
 *   y_in     [n_in ][h*w][Cp]   from mrefsr_pixnorm_f32
 *   y_ref    [n_pair][h*w][Cp]
 *   inv_ref  [n_pair][(h-2)(w-2)]  from mrefsr_patch_norm_f32 on the ref n2
 *   nrm_in   [n_in ][(h-2)(w-2)]   nrm_eps of the input n2
 *   max_idx  [n_pair][(h-2)(w-2)] int64, value ry*(w-2)+rx, lowest index on exact ties
 *   max_val  [n_pair][(h-2)(w-2)] fp32 or NULL (max corr / nrm_in, ref_map_util.py:78-84)
 * pair p matches input (p % n_in) against ref p, so refs stacked [K][B] batch in one launch.
 * C <= 256 (Cp = padded), h, w >= 3. */
int mrefsr_corr_top1_f32(const float *y_in, const float *y_ref, const float *inv_ref,
                         const float *nrm_in, int64_t *max_idx, float *max_val, int n_in,
                         int n_pair, int Cp, int h, int w, mrefsr_stream_t stream);

/* feature_match_index in its general form (ref_map_util.py:26-86): any patch_size, input_stride, ref_stride, input and
 * reference maps of different sizes, is_norm / norm_input as in the reference.  feat_in [C][h][w], feat_ref [C][hr][wr] as given
 * (NCHW, not re-laid out); max_idx [nqy*nqx] int64 = ry*nrx + rx (n = (size - patch) / stride + 1), lowest index on exact ties;
 * max_val [nqy*nqx] or NULL.  Same defined operation order as mrefsr_corr_top1_f32 (per-tap fmaf chains over the channels,
 * taps added row-major, one multiply by 1 / (||ref patch|| + 1e-5)): with patch 3, strides 1 and equal sizes it returns that
 * entry's bits.  A scalar-FMA kernel, O(n_q n_r patch^2 C): the general entry, not the benchmark's path.
 * workspace: mrefsr_feature_match_index_workspace_bytes(h, w, hr, wr) bytes of device memory. */
int64_t mrefsr_feature_match_index_workspace_bytes(int h, int w, int hr, int wr);
int mrefsr_feature_match_index_f32(const float *feat_in, const float *feat_ref, int C, int h, int w, int hr, int wr, int patch,
                                   int stride_in, int stride_ref, int is_norm, int norm_input, int64_t *max_idx, float *max_val,
                                   void *workspace, int64_t workspace_bytes, mrefsr_stream_t stream);

/* Same result, fast path: approximate MFMA pre-filter (candidates within a proven error window of
 * the approximate maximum) + exact fp32 re-scoring of the candidates in the canonical operation
 * order + brute force for queries whose candidate set overflows.  Indices and values are
 * bit-identical to mrefsr_corr_top1_f32.  ybf_* from mrefsr_pixnorm_f32 in format ybf_fmt:
 *   0  bf16 hi|lo two-term split, three bf16 MFMAs per term (any Cp)
 *   1  fp16 single plane, one fp16 MFMA per term (Cp = 256 only).  Window: `tau` [n_pair][(h-2)(w-2)], a proven bound
 *      per query on twice the error of an approximate score, i.e. with d = sqrt(d2) per pixel, D = 3x3 patch norm
 *      of d (mrefsr_patch_norm_f32 on d2), rho = max over the pair's reference patches of D_ref * inv_ref:
 *          tau = 2.02 * (D_in + (nrm_in + 3 D_in) * rho) + 2e-4 * nrm_in
 *      (Cauchy-Schwarz on y_a y_b - h_a h_b = d_a.h_b + h_a.d_b + d_a.d_b over channels and taps, + 1e-4 nrm_in for
 *      the fp32 accumulation orders).  tau = NULL: the worst-case window 2.02 * 1.1e-3 * nrm_in, 2-4x
 *      wider -- more candidates, overflow -> brute force on maps full of near-ties.
 * workspace of mrefsr_corr_workspace_bytes(n_pair, h, w) bytes.
 * ybf_ref must be followed by at least (6*w + 16) pixels of readable bytes (6 image rows + 16 pixels,
 * i.e. (6*w + 16)*2*Cp*2 bytes in format 0, (6*w + 16)*Cp*2 in format 1): edge tiles (8 rows x 16
 * pixels from an origin <= (h-3, w-3)) are staged by LDS-DMA without clamping; what is read there
 * never reaches a valid patch. */
int64_t mrefsr_corr_workspace_bytes(int n_pair, int h, int w);
/* Which pre-filter kernel mrefsr_corr_top1_prefilter_f32 launches for these operands and how much matrix work it issues: the
 * MFMA FLOP per (sample, reference) pair (< 0: invalid arguments); kernel_name (may be NULL) receives the kernel's name,
 * *mfma_dtype (may be NULL) 1 for one fp16 MFMA per product, 0 for three bf16 MFMAs.  No reference counterpart: measurement
 * support (bench.py's roofline divides this figure by the measured time of the call). */
int64_t mrefsr_corr_prefilter_info(int ybf_fmt, int Cp, int h, int w, char *kernel_name, int name_len, int *mfma_dtype);
int mrefsr_corr_top1_prefilter_f32(const float *y_in, const float *y_ref, const void *ybf_in,
                                   const void *ybf_ref, const float *inv_ref, const float *nrm_in,
                                   int64_t *max_idx, float *max_val, void *workspace,
                                   int64_t workspace_bytes, int n_in, int n_pair, int Cp, int h,
                                   int w, int ybf_fmt, const float *tau, mrefsr_stream_t stream);

/* index -> flow -> 9 shifted offset planes at scales 1, 2, 4
 * (CorrespondenceGenerationArch.index_to_flow + forward, corres_generation_arch.py:30-47,:70-105;
 * tensor_shift arch_util.py:386-410).
 *   max_idx [N][(h-2)(w-2)] int64
 *   off_s   [N][9][s*h][s*w][2] fp32, last dim [x, y]; any of the three may be NULL */
int mrefsr_offsets_from_idx_f32(const int64_t *max_idx, float *off_s1, float *off_s2,
                                float *off_s4, int N, int h, int w, mrefsr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * DynAgg glue: ref_mrapa_restoration_arch.py:56-73
 *   om  [B][3*dg*9][H][W]   output of conv_offset_mask (o1 | o2 | mask chunks); om_nhwc = 1: [B][H][W][3*dg*9]
 *   om_bias [3*dg*9] or NULL  bias of conv_offset_mask when the convolution ran without it
 *   pre [B][9][H][W][2]     pre-computed offsets, last dim [x, y]
 *   offset [B][dg*18][H][W] = om[:, :dg*18] + pre re-ordered to [y, x] per tap (:59-67)
 *   mask   [B][dg*9][H][W]  = sigmoid(om[:, dg*18:])                             (:69)
 *   abs_sum: device double[1], += sum |om[:, :dg*18]| (the :70-73 guard, no host sync); or NULL */
int mrefsr_dynagg_prep_f32(const float *om, const float *om_bias, const float *pre, float *offset,
                           float *mask, double *abs_sum, int B, int dg, int H, int W, int om_nhwc,
                           mrefsr_stream_t stream);

/* backward of the above: g_om[:, :dg*18] = g_offset ; g_om[:, dg*18:] = g_mask * m * (1 - m) */
int mrefsr_dynagg_prep_bwd_f32(const float *g_offset, const float *g_mask, const float *mask,
                               float *g_om, int B, int dg, int H, int W, mrefsr_stream_t stream);
/* The same with g_om channels-last, [B][H*W][27*dg] -- the layout the input-gradient / weight-gradient kernels of conv_offset_mask
 * read -- and its two reductions in the same pass: bias_grad[27*dg] += per-channel sums, amax[0] = max(amax[0], max |g_om|)
 * (both zero-initialised by the caller; either may be NULL).  27*dg <= 256. */
int mrefsr_dynagg_prep_bwd_nhwc_f32(const float *g_offset, const float *g_mask, const float *mask, float *g_om, float *bias_grad,
                                    float *amax, int B, int dg, int H, int W, mrefsr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * DCNv2 / DCNv1: basicsr/ops/dcn (deform_conv_ext: deform_conv_ext.cpp:52-147) and
 * mmcv.ops.modulated_deform_conv2d as called at ref_mrapa_restoration_arch.py:74-76.
 * Arithmetic spec: deform_conv_cuda_kernel.cu:467-767, deform_conv_cuda.cpp:490-685.
 *   x [B][C][H][W]; offset [B][dg*2*kh*kw][Ho][Wo] ([g][tap][y,x]); mask [B][dg*kh*kw][Ho][Wo]
 *   or NULL (DCNv1); weight [Co][C/groups][kh][kw]; bias [Co] or NULL; out [B][Co][Ho][Wo].
 *   act_slope: fused LeakyReLU slope applied to the output (1.0f = none).
 * --------------------------------------------------------------------------------------------- */
typedef struct {
    int B, C, H, W, Co, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, groups, dg;
} mrefsr_dcn_shape;

/* replaces modulated_deform_conv_forward (deform_conv_ext.cpp:107-125) / deform_conv_forward.
 * One fused kernel (deformable gather -> LDS -> MFMA GEMM -> bias + LeakyReLU) when the shape is
 * MFMA-eligible (groups 1, 3x3, C % 32 == 0, C/dg in {8,16,32,64,..}, Co in {64,128,256}: every
 * DynAgg of the path); a generic kernel otherwise.  `workspace` holds the re-packed weights
 * (mrefsr_dcn_fwd_workspace_bytes(s) bytes, 0 for the generic path).
 * nhwc (MFMA path only): bit 0: x is [B][H][W][C] -- a thread's 8 channels of a bilinear corner are
 * then two 16-byte loads instead of 8 scalar gathers; bit 1: out is written [B][Ho][Wo][Co]
 * (offset / mask stay planar).  With bit 0 the GEMM runs on the 16-bit matrix pipe from exact splits of columns
 * and weights, fp32-equivalent as in mrefsr_conv_nhwc_f32: fp16 two-term / three products (needs |column| < 65504:
 * `range_flag`, an int32 in device memory or NULL, is set to 1 otherwise -- same contract as the convolution's), or
 * the bf16 three-term / six-product split without range limit (bit 3, with bit 0; the re-run path of a caller whose
 * range flag fired).
 * bit 4 (with bits 0, 1, 2): x and out are bf16 tensors (2-byte channels-last storage; offset / mask stay fp32 planar).
 * bit 2 (with bit 0): bf16 ARITHMETIC instead (BASELINE configs[4]): columns and weights rounded to bf16,
 * fp32 accumulation, output rounded to bf16 in its fp32 container. */
int64_t mrefsr_dcn_fwd_workspace_bytes(const mrefsr_dcn_shape *s);
int mrefsr_dcn_fwd_f32(const float *x, const float *offset, const float *mask,
                       const float *weight, const float *bias, float *out,
                       const mrefsr_dcn_shape *s, float act_slope, int nhwc, void *workspace,
                       int64_t workspace_bytes, int *range_flag, mrefsr_stream_t stream);
/* the same with out_amax[0] = max(out_amax[0], max |out|) (device memory, zero-initialised by the caller, may be NULL; channels-last
 * x only): the input scale of the Winograd convolutions that read the aggregated features (MRAPAFusion.conv_emb2 / conv_ass,
 * ref_mrapa_restoration_arch.py:271-304, 321-335) -- see mrefsr_conv_nhwc_amax_f32. */
int mrefsr_dcn_fwd_amax_f32(const float *x, const float *offset, const float *mask,
                            const float *weight, const float *bias, float *out,
                            const mrefsr_dcn_shape *s, float act_slope, int nhwc, void *workspace,
                            int64_t workspace_bytes, int *range_flag, float *out_amax, mrefsr_stream_t stream);

/* columns[B][C*kh*kw][Ho*Wo] = mask * bilinear(x)  (modulated_deformable_im2col, .cu:570-633);
 * used by the backward's weight gradient (deform_conv_cuda.cpp:640-663). */
int mrefsr_dcn_im2col_f32(const float *x, const float *offset, const float *mask, float *columns,
                          const mrefsr_dcn_shape *s, mrefsr_stream_t stream);

/* grad_col [B][C*kh*kw][Ho*Wo] (= W^T grad_out, computed by the caller's GEMM) ->
 * grad_offset, grad_mask (assigned; .cu:695-767) and grad_x (accumulated with atomics into a
 * caller-zeroed buffer; .cu:635-693).  grad_x or grad_mask may be NULL. */
int mrefsr_dcn_col2im_f32(const float *grad_col, const float *x, const float *offset,
                          const float *mask, float *grad_x, float *grad_offset, float *grad_mask,
                          const mrefsr_dcn_shape *s, mrefsr_stream_t stream);

/* The backward of the modulated deformable convolution w.r.t. offset, mask and input as ONE fused launch -- what
 * deform_conv_cuda.cpp:571-685 runs as a GEMM (d columns = W^T . grad_out, :617-620) into a C*9*H*W buffer followed by
 * modulated_deformable_col2im + col2im_coord (deform_conv_cuda_kernel.cu:635-767): the column gradient is formed per tile of 32
 * pixels on the matrix pipe (fp16 two-term split, fp32-equivalent) and consumed from the accumulators.
 *   3x3, stride 1, pad 1, dilation 1, groups 1, C % 32 == 0, Co % 16 == 0, C / dg in {8, 16, 32}
 *   grad_out [B][H][W][Co], x [B][H][W][C] channels-last; offset [B][18 dg][H][W], mask [B][9 dg][H][W] planar (mask may be NULL)
 *   packed_wT: the weight [Co][C][3][3] packed by mrefsr_conv_pack_weight_view_f32(weight, packed, Cout := C, Cin := Co, 3, terms 16,
 *              wscale, stride_o = 9, stride_i = 9 C, flip = 0): the transposed operator, taps in place
 *   g_amax: device float, max |grad_out| (NULL: grad_out is used as it is; it must then lie inside the fp16 range)
 *   grad_x [B][C][H][W] PLANAR, zero-initialised by the caller (float atomics), or NULL; grad_offset / grad_mask like offset / mask */
int mrefsr_dcn_bwd_data_f32(const float *grad_out, const float *x, const float *offset, const float *mask, const void *packed_wT,
                            float wscale, const float *g_amax, float *grad_x, float *grad_offset, float *grad_mask,
                            const mrefsr_dcn_shape *s, mrefsr_stream_t stream);
/* d weight [Co][C][3][3] of the same convolution: grad_out . columns^T (deform_conv_cuda.cpp:640-657) with the columns re-gathered
 * tile by tile instead of read back from a C*9*H*W buffer -- a pixel-K GEMM on the matrix pipe (fp16 two-term split of both
 * operands), partial sums per K split in `workspace` (mrefsr_dcn_bwd_weight_workspace_bytes), added in split order.
 * Layouts as mrefsr_dcn_bwd_data_f32; range_flag (device int, may be NULL) is raised when a column leaves the fp16 range. */
int64_t mrefsr_dcn_bwd_weight_workspace_bytes(const mrefsr_dcn_shape *s);
int mrefsr_dcn_bwd_weight_f32(const float *grad_out, const float *x, const float *offset, const float *mask, const float *g_amax,
                              float *grad_weight, void *workspace, int64_t workspace_bytes, const mrefsr_dcn_shape *s,
                              int *range_flag, mrefsr_stream_t stream);

/* Weight gradient of a 1x1 convolution (conv_emb1, spatial_attn, feat_fusion of ref_mrapa_restoration_arch.py:271-304; what
 * torch.autograd runs as miopenConvolutionBackwardWeights under multi_ref_restoration_model.py:197-279):
 * grad_weight[o][i] = sum over pixels of g[pixel][o] * x[pixel][i], g [pixels][ld_g] and x [pixels][ld_x] channels-last fp32;
 * a pixel-K GEMM on the matrix pipe (fp16 two-term split of both operands, g scaled by max |g| = *g_amax into the fp16 range),
 * partial sums per K split in `workspace`, added in split order (deterministic). */
int64_t mrefsr_conv_wgrad1x1_workspace_bytes(int64_t pixels, int Cout, int Cin);
int mrefsr_conv_wgrad1x1_f32(const float *g, const float *x, const float *g_amax, float *grad_weight, void *workspace,
                             int64_t workspace_bytes, int64_t pixels, int Cout, int ld_g, int Cin, int ld_x, int *range_flag,
                             mrefsr_stream_t stream);
/* The same three operators for every dtype of the reference's dispatch (AT_DISPATCH_FLOATING_TYPES_AND_HALF,
 * deform_conv_cuda_kernel.cu:259,353,451,781,813,846): all tensors of one `dtype` (0 = f32, 1 = f16, 3 = f64), planar
 * NCHW layouts as above, any stride / dilation / groups / kernel size; float accumulation (double for f64).  Portable
 * kernels (one thread per output pixel x 16 channels), not the fused MFMA path: the fp32 product path is
 * mrefsr_dcn_fwd_f32.  mrefsr_dcn_col2im takes f32 / f64 (native atomics); an f16 caller accumulates gradients in f32. */
int mrefsr_dcn_fwd(const void *x, const void *offset, const void *mask, const void *weight, const void *bias,
                   void *out, const mrefsr_dcn_shape *s, float act_slope, int dtype, mrefsr_stream_t stream);
int mrefsr_dcn_im2col(const void *x, const void *offset, const void *mask, void *columns,
                      const mrefsr_dcn_shape *s, int dtype, mrefsr_stream_t stream);
int mrefsr_dcn_col2im(const void *grad_col, const void *x, const void *offset, const void *mask, void *grad_x,
                      void *grad_offset, void *grad_mask, const mrefsr_dcn_shape *s, int dtype,
                      mrefsr_stream_t stream);


/* ---------------------------------------------------------------------------------------------
 * Multi-reference feature-transfer attention core: ref_mrapa_restoration_arch.py:321-335
 *   q   [N][c][HW]       conv_emb1(target) * c^-1/2
 *   emb [N*T][c][HW]     conv_emb2(refs)  (refs stacked [N][T] on dim 0, :318)
 *   ass [N*T][c2][HW]    conv_ass(refs)
 *   out [N][c2][HW]      sum_t softmax_t(<q, emb_t>) * ass_t        T <= 16
 *   prob [N][T][HW]      softmax weights (saved for backward) or NULL
 *   t_major              0: refs stacked [N][T] as above; 1: stacked [T][N] (image t*N + n), the
 *                        layout of the batched-over-references path (no permute copy either way)
 * --------------------------------------------------------------------------------------------- */
int mrefsr_mrattn_fwd_f32(const float *q, const float *emb, const float *ass, float *out,
                          float *prob, int N, int T, int c, int c2, int HW, int t_major,
                          mrefsr_stream_t stream);
/* channels-last forward for the inference path: q [N][HW][c], emb [T*N][HW][c], ass [T*N][HW][2c]
 * (t-major), out [N][HW][2c]; c in {64, 128, 256}, T <= 16. */
int mrefsr_mrattn_fwd_nhwc_f32(const float *q, const float *emb, const float *ass, float *out, int N,
                               int T, int c, int HW, mrefsr_stream_t stream);
/* the same with `q * q_scale` (`self.conv_emb1(target) * self.scale`, ref_mrapa_restoration_arch.py:321) formed on the way in: each product
 * rounded on its own -- the bits of a separate element-wise pass over q, without the pass. */
int mrefsr_mrattn_fwd_nhwc_scaled_f32(const float *q, const float *emb, const float *ass, float *out, int N,
                                      int T, int c, int HW, float q_scale, mrefsr_stream_t stream);
int mrefsr_mrattn_bwd_f32(const float *q, const float *emb, const float *ass, const float *prob,
                          const float *g_out, float *g_q, float *g_emb, float *g_ass, int N,
                          int T, int c, int c2, int HW, int t_major, mrefsr_stream_t stream);
/* The channels-last attention core on bf16 tensors (2-byte activation storage, BASELINE configs[4]): same lanes and
 * operation order as mrefsr_mrattn_fwd_nhwc_f32, fp32 math, the result rounded to bf16 (round-to-nearest-even). */
int mrefsr_mrattn_fwd_nhwc_bf16(const void *q, const void *emb, const void *ass, void *out, int N, int T,
                                int c, int HW, mrefsr_stream_t stream);


/* ---------------------------------------------------------------------------------------------
 * basicsr/ops/fused_act: fused_bias_act(input, bias, refer, act, grad, alpha, scale)
 * (fused_bias_act.cpp:14-26, kernel fused_bias_act_kernel.cu:19-50).  bias / ref may be NULL
 * ("empty tensor" in the reference).  dtype: 0 = f32, 1 = f16, 2 = bf16, 3 = f64 (the reference dispatches
 * AT_DISPATCH_FLOATING_TYPES_AND_HALF, fused_bias_act_kernel.cu:81; bf16 is an extension); math in float, double for f64.
 * --------------------------------------------------------------------------------------------- */
int mrefsr_fused_bias_act(const void *x, const void *bias, const void *ref, void *out,
                          int64_t size_x, int step_b, int size_b, int act, int grad, float alpha,
                          float scale, int dtype, mrefsr_stream_t stream);

/* Tail of the restoration network (ref_mrapa_restoration_arch.py:132-137: base = F.interpolate(x, None, 4, 'bilinear', False); ... out + base):
 * out [B][C][h*scale][w*scale] (NCHW) = y [B][h*scale][w*scale][ld >= C] (channels-last result of the last convolution) +
 * bilinear(x [B][C][h][w]) with align_corners = False, in one pass; the interpolation returns the bits of torch's upsample_bilinear2d. */
int mrefsr_tail_bilinear_add_f32(const float *y_nhwc, const float *x, float *out, int B, int C, int h, int w, int scale, int ld,
                                 mrefsr_stream_t stream);

/* Convolution epilogue of the NCHW fp32 path (the `conv -> (+bias) -> LeakyReLU/ReLU -> (+x)` idiom
 * of ResidualBlockNoBN arch_util.py:113-116, the VGG stacks and the lrelu(conv(.)) chains of
 * ref_mrapa_restoration_arch.py): out = lrelu(x + bias[c] + pre, slope) + residual in one pass.
 * pre [pre_N][C][HW] is added BEFORE the activation and broadcast over N / pre_N groups (image n
 * uses pre[n % pre_N]: the x-half of offset_conv1 shared by the K references); bias / pre /
 * residual may be NULL; out may alias x; slope 1 = identity, 0 = ReLU. */
int mrefsr_bias_act_res_f32(const float *x, const float *bias, const float *pre, int64_t pre_N,
                            const float *residual, float *out, int64_t N, int C, int64_t HW,
                            float slope, mrefsr_stream_t stream);

/* 3x3 (pad 1) / 1x1 stride-1 convolution with fused epilogue, for the residual trunks, VGG stacks,
 * offset convolutions and fusion heads (arch_util.py ResidualBlockNoBN, ref_mrapa_restoration_arch.py
 * :139-348, vgg_arch.py, contras_multi_extractor_arch.py) on channels-last activations.
 * fp32-equivalent arithmetic on the bf16 matrix pipe: operands split exactly into 3 bf16 terms,
 * `terms` = 6 partial products per product (all those >= 2^-24 relative; 3 = two-term split,
 * ~2^-16 relative, for experiments only; 1 = bf16 ARITHMETIC for BASELINE configs[4]: both operands rounded to
 * bf16, one product, fp32 accumulation, result rounded to bf16 in its fp32 container).  `terms` = 16: fp16 two-term split (11 + 11 significand
 * bits, three products, dropped term 2^-22 relative; as accurate as an fp32 convolution whose own
 * accumulation error dominates, at twice the speed of terms = 6).  It needs |activation| < 65504 and
 * a per-layer power-of-two weight scale `wscale` with max|w| * wscale in [2^13, 2^14), given to the
 * pack call and in the descriptor (the epilogue divides it out); `range_flag` (device int, may be
 * NULL) is set to 1 by any block that meets an activation outside +-65000 (or NaN): the result of
 * that launch is then not to be trusted and the caller should rerun with terms = 6.
 * `terms` = 2 (descriptor only; weights packed with terms = 1): the bf16 arithmetic of terms = 1 on bf16 TENSORS -- x1, x2,
 * pre, residual and out are [..][C] arrays of 2-byte bf16 passed through the same pointers (channel counts and leading
 * dimensions multiples of 8; bias and slope stay fp32).  Same values as terms = 1, half the activation bytes.
 * `terms` = 17: the terms-16 arithmetic in Winograd F(2x2, 3x3) form (3x3 kernels, fp32 tensors, epilogues 0 / 1 / 2, at least 17
 * input channels, H W ld 4 < 2^32): Y = A^T [sum_c (G g G^T) . (B^T d B)] A with G g G^T formed (fp64) and split at pack time under the
 * same `wscale`, B^T d B formed in fp32 and then split -- 2.25x fewer MFMAs per output, the same or a smaller error against fp64
 * (the accumulation chains are 9x shorter); the fp16 guard fires when a transform value leaves the fp16 range (|B^T d B| <= 4 max|x|
 * after the input scale: the eight-wave kernel compares |x| with 16000, the four-wave kernel flags the non-finite outputs), the low term of an
 * activation is an fp16 subnormal below |x| = 2^-3 (absolute error <= 2^-25) unless the launch is given the input's maximum
 * (mrefsr_conv_nhwc_scaled_f32).  Weights packed with terms = 17
 * (mrefsr_conv_packed_bytes / mrefsr_conv_pack_weight[_view]_f32) only serve terms = 17 descriptors.
 *   input   = channel concatenation of x1 [N1][H][W][ld1] (first C1 channels used) and, if C2 > 0,
 *             x2 [N2][H][W][ld2]; image n reads x1[n % N1], x2[n % N2] (batch broadcast);
 *             C1, C2, ld1, ld2 multiples of 4; C1 a multiple of 16 when C2 > 0
 *   packed  = weight [Cout][C1+C2][k][k] re-ordered once by mrefsr_conv_pack_weight_f32 into
 *             mrefsr_conv_packed_bytes(Cout, C1+C2, ksize, terms) bytes (a weight with fewer input
 *             channels than C1+C2 packs with zero padding: pass its own Cin to both calls)
 *   v       = conv + bias[c] + pre[n % pre_N][y][x][c]          (bias, pre may be NULL; pre has ld = Cout)
 *   v       = act ? LeakyReLU(v, slope_ptr ? *slope_ptr : slope) : v     (slope 0 = ReLU; *slope_ptr = PReLU)
 *   v      += residual[n][y][x][c]  (ld_res)                    (may be NULL)
 *   epilogue 0: out[n][y][x][c] (ld_out)   1: MaxPool2d(2,2) -> out [N][H/2][W/2][ld_out]
 *            2: PixelShuffle(2)  -> out [N][2H][2W][ld_out], channel c/4 */
typedef struct mrefsr_conv_desc {
    int32_t N, H, W, ksize;
    int32_t C1, ld1, N1;
    int32_t C2, ld2, N2;
    int32_t Cout, ld_out, ld_res, pre_N;
    int32_t act, epilogue, terms;
    float slope;
    float wscale; /* terms == 16 only */
} mrefsr_conv_desc;
int64_t mrefsr_conv_packed_bytes(int Cout, int Cin, int ksize, int terms);
int mrefsr_conv_pack_weight_f32(const float *weight, void *packed, int Cout, int Cin, int ksize, int terms,
                                float wscale, mrefsr_stream_t stream);
/* The same packing of a strided VIEW of a weight: element (o, i, tap) is read at
 * weight[o * stride_o + i * stride_i + (flip ? k*k - 1 - tap : tap)].  stride_o = Cin_total*k*k, stride_i = k*k, flip = 0 and a
 * pointer offset select an input-channel slice in place (the two halves of offset_conv1, ref_mrapa_restoration_arch.py:217);
 * swapped strides with flip = 1 pack the operator of the convolution's INPUT GRADIENT (Cout := the slice's channels,
 * Cin := the original Cout): what torch.autograd runs as miopenConvolutionBackwardData in the reference's training step
 * (multi_ref_restoration_model.py:197-279) is mrefsr_conv_nhwc_f32 on the output gradient with these weights. */
int mrefsr_conv_pack_weight_view_f32(const float *weight, void *packed, int Cout, int Cin, int ksize, int terms, float wscale,
                                     int64_t stride_o, int64_t stride_i, int flip, mrefsr_stream_t stream);
/* n_jobs such packings in ONE launch -- every convolution weight of net_g and its input-gradient operator after an optimiser
 * step (optimizer_g.step(), multi_ref_restoration_model.py:277: the reference's weights change once per step, so do the packed
 * copies).  `jobs` is a table in DEVICE memory, each entry the arguments of mrefsr_conv_pack_weight_view_f32.  range_flag
 * (device int32 or NULL) is set to 1 if a terms-16 entry meets |weight * wscale| > 65000 or a non-finite weight. */
typedef struct mrefsr_conv_pack_job {
    const float *weight;
    void *packed;
    int64_t stride_o, stride_i;
    int32_t Cout, Cin, ksize, terms, flip;
    float wscale;
} mrefsr_conv_pack_job;
int mrefsr_conv_pack_weights_multi_f32(const mrefsr_conv_pack_job *jobs, int n_jobs, int *range_flag, mrefsr_stream_t stream);
int mrefsr_conv_nhwc_f32(const mrefsr_conv_desc *d, const float *x1, const float *x2, const void *packed,
                         const float *bias, const float *slope_ptr, const float *pre, const float *residual,
                         float *out, int *range_flag, mrefsr_stream_t stream);
/* mrefsr_conv_nhwc_f32 (terms = 16) on inputs of unknown, possibly tiny magnitude -- the output gradients of a training step
 * (multi_ref_restoration_model.py:197-279), which sit far below the fp16 normal range: in_amax[0] (device memory, written by
 * mrefsr_act_bwd_nhwc_f32) is max |x| over the input tensor(s); the kernel multiplies x by the power of two that brings it into
 * [2^13, 2^14) before the two-term split and the result by its inverse -- both exact -- so the three-product mode serves the
 * input-gradient convolutions as it serves the forward ones.  in_amax = NULL: exactly mrefsr_conv_nhwc_f32.
 * terms = 17 takes it too: the Winograd kernels split B^T d B after forming it, and the LOW term of a value below 2^-3 is an fp16
 * subnormal; with in_amax they bring max |x| into [2^11, 2^12) first (the transform grows values by at most 4) and the forward
 * convolution of small activations (1e-2 and less) is as accurate as an fp32 one.  The host mirror hands every layer the maximum
 * its producer measured (mrefsr_conv_nhwc_amax_f32 / mrefsr_dcn_fwd_amax_f32: out_amax of one launch = in_amax of the next). */
int mrefsr_conv_nhwc_scaled_f32(const mrefsr_conv_desc *d, const float *x1, const float *x2, const void *packed,
                                const float *bias, const float *slope_ptr, const float *pre, const float *residual,
                                float *out, int *range_flag, const float *in_amax, mrefsr_stream_t stream);
/* mrefsr_conv_nhwc_scaled_f32 that also MEASURES its output: out_amax[0] = max(out_amax[0], max |out|) (device memory, the caller
 * zero-initialises it; may be NULL; terms 16 / 17, fp32 tensors, every epilogue).  The Winograd launch that reads `out` next takes
 * the same word as its in_amax: the input scale of every 3x3 layer of the path is the CURRENT batch's maximum, measured in the
 * producing kernel's epilogue (the values are in registers there) -- no reduction pass, no calibration that could go stale
 * (round 5 measured once per layer with two torch launches; archs/nhwc.py).  Replaces, per layer, what the reference leaves to
 * fp32 arithmetic (arch_util.py:89-117, ref_mrapa_restoration_arch.py:213-259). */
int mrefsr_conv_nhwc_amax_f32(const mrefsr_conv_desc *d, const float *x1, const float *x2, const void *packed,
                              const float *bias, const float *slope_ptr, const float *pre, const float *residual,
                              float *out, int *range_flag, const float *in_amax, float *out_amax, mrefsr_stream_t stream);
/* The input-gradient convolution of a training step (torch.autograd's miopenConvolutionBackwardData in the reference,
 * multi_ref_restoration_model.py:197-279) with the element-wise pass that would follow it folded into its epilogue -- inside a
 * residual block (arch_util.py:45-70) the gradient of conv2's input is masked by the ReLU between the two convolutions and
 * summed per channel for conv1's bias:  residual_is_mask = 1 turns `residual` into that mask source (out = residual > 0 ? conv : 0);
 * stat_sum[Cout] += per-channel sums of out, stat_amax[0] = max(stat_amax[0], max |out|) (zero-initialised by the caller, either
 * may be NULL).  terms = 16, fp32 tensors, plain epilogue, Cout / ld_out / ld_res multiples of 4; in_amax as above. */
int mrefsr_conv_nhwc_bwd_f32(const mrefsr_conv_desc *d, const float *x1, const void *packed, const float *residual,
                             int residual_is_mask, float *out, int *range_flag, const float *in_amax, float *stat_sum,
                             float *stat_amax, mrefsr_stream_t stream);
/* conv_offset_mask of a DynAgg + its glue in one launch (ref_mrapa_restoration_arch.py:56-73: chunk / cat / repeat /
 * re-order / add / sigmoid / mean-abs): the 3x3 convolution `x` [N][H][W][C1] -> 27*dg channels runs as in
 * mrefsr_conv_nhwc_f32 (same packed weights, terms, wscale, range flag; fields N, H, W, C1, ld1, Cout = 27*dg, ksize = 3,
 * terms, wscale of `d` are read), and its epilogue writes what mrefsr_dcn_fwd_f32 reads, PLANAR:
 *   offset [N][18*dg][H][W] = channels [0, 18 dg) + pre_offset (pre_offset [N][9][H][W][2] is [x, y] per tap: channel
 *                              g*18 + 2*tap gets y, + 1 gets x),   mask [N][9*dg][H][W] = sigmoid(channels [18 dg, 27 dg)),
 *   *abs_sum (double, device, may be NULL) += sum |channels [0, 18 dg)|  (the reference's `offset mean > 100` guard, read by
 *   the host when it likes).  Bit-identical to mrefsr_conv_nhwc_f32 followed by mrefsr_dynagg_prep_f32(om_nhwc = 1). */
int mrefsr_conv_dynagg_f32(const mrefsr_conv_desc *d, const float *x, const void *packed, const float *bias,
                           const float *pre_offset, float *offset, float *mask, double *abs_sum, int dg,
                           int *range_flag, mrefsr_stream_t stream);


/* Spatial-attention modulation of MRAPAFusion (ref_mrapa_restoration_arch.py:343-345) in one pass:
 * mul_inout[i] = refs[i] * sigmoid(mul_inout[i]) * 2 + add[i]; n a multiple of 4, any (common) layout. */
int mrefsr_attn_modulate_f32(const float *refs, float *mul_inout, const float *add, int64_t n,
                             mrefsr_stream_t stream);
int mrefsr_attn_modulate_bf16(const void *refs, void *mul_inout, const void *add, int64_t n, mrefsr_stream_t stream);


/* Input normalisation of the feature extractors (vgg_arch.py:150-153: `(x + 1) / 2` when range_norm, then `(x - mean) / std`;
 * contras_multi_extractor_arch.py:41) fused with the engine's channels-last packing: img [N][3][HW] -> out [N][HW][4]
 * (channel 3 = 0; the packed first-layer weights carry a zero fourth input channel).  mean3 / std3: 3 floats in device memory,
 * both NULL = no normalisation.  ATen's operations in ATen's order: bit-identical to the six launches it replaces. */
int mrefsr_image_to_nhwc4_f32(const float *img, float *out, int64_t N, int64_t HW, int range_norm, const float *mean3,
                              const float *std3, mrefsr_stream_t stream);


/* conv -> +bias -> ReLU -> MaxPool2d(2, 2) of the VGG stacks (vgg_arch.py:113-120,
 * contras_multi_extractor_arch.py:14-27) in one pass: out [N][C][H/2][W/2] = relu(max2x2(x) + bias[c])
 * (bit-identical to pooling the biased, rectified map). */
int mrefsr_bias_relu_pool2_f32(const float *x, const float *bias, float *out, int64_t N, int C, int H,
                               int W, mrefsr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * basicsr/ops/upfirdn2d: upfirdn2d(input (major,in_h,in_w,minor), kernel (kh,kw), up, down, pad)
 * (upfirdn2d.cpp:13-24, upfirdn2d_kernel.cu:50-370).  out (major,out_h,out_w,minor) with
 * out_h = (in_h*up_y + pad_y0 + pad_y1 - kh + down_y) / down_y  (.cu:240-243).
 * --------------------------------------------------------------------------------------------- */
int mrefsr_upfirdn2d_f32(const float *in, const float *kernel, float *out, int major, int in_h,
                         int in_w, int minor, int kh, int kw, int up_x, int up_y, int down_x,
                         int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                         mrefsr_stream_t stream);
/* The same for any of the reference's dtypes (AT_DISPATCH_FLOATING_TYPES_AND_HALF, upfirdn2d_kernel.cu:312): input, kernel
 * and output share `dtype` (0 = f32, 1 = f16, 2 = bf16 (extension), 3 = f64); accumulation in float (double for f64). */
int mrefsr_upfirdn2d(const void *in, const void *kernel, void *out, int major, int in_h, int in_w,
                     int minor, int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0,
                     int pad_x1, int pad_y0, int pad_y1, int dtype, mrefsr_stream_t stream);

/* ---- backward glue of the channels-last training engine (net_g under MultiRefRestorationModel.optimize_parameters,
 * multi_ref_restoration_model.py:197-279; what torch.autograd runs as separate elementwise / reduction kernels per layer).
 *
 * mrefsr_act_bwd_nhwc_f32: backward of the fused convolution epilogue  out = act(conv + bias):
 *   g_pre[p][c] (ld_pre; may be NULL) = g_out[p][c] * (out[p][c] > 0 ? 1 : slope)        act 1: LeakyReLU(slope), 0 = ReLU
 *                                                                                      act 2: PReLU(*slope_ptr); act 0: copy
 *   bias_grad[c] (may be NULL)  += sum of g_pre over the pixels: the bias gradient (torch's conv backward bias term).  The
 *                     caller zero-initialises it; blocks add their totals with float atomics (order not fixed, like the
 *                     reference's own atomically accumulating backward kernels, deform_conv_cuda_kernel.cu:330,688)
 *   slope_grad[0] (act 2, may be NULL) += sum of g_out * x over x < 0, x = out / slope (PReLU weight gradient); needs
 *                     slope > 0 -- *flag (int32, device, may be NULL) is set to 1 otherwise
 *   amax[0] (may be NULL) = max(amax[0], max |g_pre|) (zero-initialised by the caller): the input scale of
 *                     mrefsr_conv_nhwc_scaled_f32 for the input-gradient convolution that follows
 *   g_out, out contiguous [npix][C]; C <= 1024, a multiple of 4 when > 256. */
int mrefsr_act_bwd_blocks(int64_t npix, int C);
int mrefsr_act_bwd_nhwc_f32(const float *g_out, const float *out, float *g_pre, int ld_pre, float *bias_grad, float *slope_grad, float *amax,
                            int64_t npix, int C, int act, float slope, const float *slope_ptr, int *flag, mrefsr_stream_t stream);
/* Weight gradient of a 3x3 stride-1 'same' convolution over channels-last tensors (torch's miopenConvolutionBackwardWeights in
 * the reference's training step):  dw[co][ci][ty][tx] (+)= sum_{n,y,x} g[n][y][x][co] * x[n][y+ty-1][x+tx-1][ci].
 *   x [N][H][W][ld_x] (channels [0, Cin) used), g [N][H][W][ld_g] (channels [0, Cout)); dw element (co, ci, tap) at
 *   dw[co*stride_co + ci*stride_ci + tap] -- an OIHW tensor or an input-channel slice of one; accumulate = 0 overwrites it,
 *   1 adds to it.  Blocks leave 64 x 64 x 9 partial sums in `workspace` (mrefsr_conv_wgrad3x3_workspace_bytes), a second
 *   launch adds them in a fixed order: deterministic.
 *   g_amax[0] = max |g| (device memory, from mrefsr_act_bwd_nhwc_f32): the gradient is scaled by an exact power of two into the
 *   fp16 range before its two-term split, as in mrefsr_conv_nhwc_scaled_f32; x must satisfy |x| < 65504 (range_flag, may be
 *   NULL).  fp32-equivalent: three fp16 MFMA products per term, fp32 accumulation. */
int64_t mrefsr_conv_wgrad3x3_workspace_bytes(int N, int H, int W, int Cin, int Cout);
/* The same for n_jobs (<= MREFSR_WGRAD_MAX_JOBS) convolutions of ONE geometry -- the 32 convolutions of a residual trunk
 * (arch_util.py:45-70 stacked 16 times, ref_mrapa_restoration_arch.py) whose weight gradients autograd runs one after the other --
 * in one launch pair: x, g, g_amax, dw are host arrays of n_jobs device pointers. */
#define MREFSR_WGRAD_MAX_JOBS 32
int64_t mrefsr_conv_wgrad3x3_batch_workspace_bytes(int n_jobs, int N, int H, int W, int Cin, int Cout);
int mrefsr_conv_wgrad3x3_batch_f32(int n_jobs, const float *const *x, int ld_x, int Cin, const float *const *g, int ld_g, int Cout,
                                   float *const *dw, int64_t stride_co, int64_t stride_ci, int accumulate, const float *const *g_amax,
                                   int N, int H, int W, void *workspace, int64_t workspace_bytes, int *range_flag, mrefsr_stream_t stream);
int mrefsr_conv_wgrad3x3_f32(const float *x, int ld_x, int Cin, const float *g, int ld_g, int Cout, float *dw, int64_t stride_co,
                             int64_t stride_ci, int accumulate, const float *g_amax, int N, int H, int W, void *workspace,
                             int64_t workspace_bytes, int *range_flag, mrefsr_stream_t stream);
/* gradient of mrefsr_mrattn_fwd_nhwc_f32 (ref_mrapa_restoration_arch.py:321-335 under autograd): same layouts, g_out [N][HW][2c]
 * -> g_q [N][HW][c], g_emb [T*N][HW][c], g_ass [T*N][HW][2c]; the softmax is recomputed, nothing is saved by the forward. */
int mrefsr_mrattn_bwd_nhwc_f32(const float *q, const float *emb, const float *ass, const float *g_out, float *g_q, float *g_emb,
                               float *g_ass, int N, int T, int c, int HW, mrefsr_stream_t stream);
/* gradient of refs * sigmoid(mul) * 2 + add (ref_mrapa_restoration_arch.py:343-345) w.r.t. refs and mul (d/d add = g);
 * `mul` is the value BEFORE mrefsr_attn_modulate_f32 overwrote it. */
int mrefsr_attn_modulate_bwd_f32(const float *g, const float *refs, const float *mul, float *g_refs, float *g_mul, int64_t n,
                                 mrefsr_stream_t stream);

/* Fingerprints of n device tensors of 32-bit words: table[2t] = address, table[2t+1] = word count (device memory);
 * sums[t] = sum_i word_i * (2 i + 1) mod 2^64 (exact integer arithmetic: independent of the summation order); `done` is n words
 * of scratch.  With `ref` given, `*flag |= flag_bits` (device memory) where sums[t] != ref[t].  No reference counterpart: the host
 * side keeps packed copies of the convolution weights and uses this to notice parameters edited behind autograd's back
 * (`.data` writes); one launch per forward pass. */
int mrefsr_weights_checksum(const int64_t *table, int n, uint64_t *sums, uint32_t *done, const uint64_t *ref, int *flag, int flag_bits,
                            mrefsr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MREFSR_HIP_H */
