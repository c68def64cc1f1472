/*
 * mrefsr_oracle.c -- CPU restatement of the MRefSR matching-and-reconstruction hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under mrefsr_amd/ may import, link or call this file.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it (as the checker /
 * the timed CPU port, never as the product).
 *
 * Parity status (see DESIGN.md "Oracle"):
 *   - correlation / top-1 / offsets: PINNED against outputs of the reference's own Python
 *     (basicsr/archs/ref_map_util.py, corres_generation_arch.py) imported in the build container;
 *     vectors under tests/golden/, generator tests/golden/gen_golden.py.
 *   - DCNv2: the reference calls mmcv.ops.modulated_deform_conv2d (third party, un-vendored,
 *     version unpinned: absent from requirements.txt).  Restated here from the vendored
 *     same-lineage spec basicsr/ops/dcn/src/deform_conv_cuda_kernel.cu:467-767 and
 *     deform_conv_cuda.cpp:490-685.  "parity unpinned" at the mmcv boundary.
 *   - fused_act: formula of basicsr/ops/fused_act/src/fused_bias_act_kernel.cu:19-50 (no CPU path
 *     in the reference) -- "parity unpinned"; upfirdn2d: pinned against upfirdn2d_native
 *     (basicsr/ops/upfirdn2d/upfirdn2d.py:162-192) run in the build container.
 *
 * Floating point contract for the correlation path (what makes index parity bit-exact between
 * this file and the HIP kernels): every fp32 operation below is written in a DEFINED order, with
 * explicit fmaf() where a fused multiply-add is meant, and the HIP kernels perform the same
 * operations in the same order (v_mfma_f32_32x32x2_f32 is a k-ordered fmaf chain).  Compile with
 * -ffp-contract=off so the compiler neither fuses nor splits anything.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * Per-pixel channel L2 normalisation.
 * Reference: corres_generation_arch.py:57-59  F.normalize(feat.reshape(c,-1), dim=0)
 *            (torch: x / max(||x||_2, 1e-12)).
 * Defined order: ss = fmaf chain over c ascending from +0; d = max(sqrt(ss), 1e-12); y = x / d.
 * Also returns n2[p] = fmaf chain over c of y[c,p]^2 (used by the patch norms below).
 * x, y: [C][HW]; n2: [HW].
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_pixnorm(const float *x, int C, int HW, float *y, float *n2)
{
#pragma omp parallel for schedule(static)
    for (int p = 0; p < HW; ++p) {
        float ss = 0.0f;
        for (int c = 0; c < C; ++c) {
            float v = x[(size_t)c * HW + p];
            ss = fmaf(v, v, ss);
        }
        float d = sqrtf(ss);
        if (!(d > 1e-12f)) d = 1e-12f;
        float s2 = 0.0f;
        for (int c = 0; c < C; ++c) {
            float v = x[(size_t)c * HW + p] / d;
            y[(size_t)c * HW + p] = v;
            s2 = fmaf(v, v, s2);
        }
        n2[p] = s2;
    }
}

/* n2[p] = fmaf chain over c ascending of y[c,p]^2 (the second chain of orc_pixnorm), for maps
 * that are already normalised. */
ORC_API void orc_sumsq(const float *y, int C, int HW, float *n2)
{
    for (int p = 0; p < HW; ++p) {
        float s2 = 0.0f;
        for (int c = 0; c < C; ++c) {
            float v = y[(size_t)c * HW + p];
            s2 = fmaf(v, v, s2);
        }
        n2[p] = s2;
    }
}

/* ------------------------------------------------------------------------------------------
 * 3x3 patch norms.
 * Reference: ref_map_util.py:62-63  batch.norm(p=2, dim=(0,1,2)) + 1e-5   (ref patches)
 *            ref_map_util.py:79-80  patches_input.norm(...) + 1e-5       (input patches)
 * Defined order: s = sequential fp32 adds of the 9 pixel sums-of-squares in row-major tap order;
 * nrm_eps = sqrtf(s) + 1e-5f; inv = 1.0f / nrm_eps.
 * n2: [h][w]; outputs: [(h-2)][(w-2)].
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_patch_norm(const float *n2, int h, int w, float *nrm_eps, float *inv)
{
    int ph = h - 2, pw = w - 2;
    for (int y = 0; y < ph; ++y)
        for (int x = 0; x < pw; ++x) {
            float s = n2[y * w + x];
            for (int t = 1; t < 9; ++t) s = s + n2[(y + t / 3) * w + (x + t % 3)];
            float ne = sqrtf(s) + 1e-5f;
            if (nrm_eps) nrm_eps[y * pw + x] = ne;
            if (inv) inv[y * pw + x] = 1.0f / ne;
        }
}

/* ------------------------------------------------------------------------------------------
 * Patch correlation + top-1.
 * Reference: ref_map_util.py:26-86 feature_match_index(feat_input, feat_ref, patch_size=3,
 *            input_stride=1, ref_stride=1, is_norm=True, norm_input=True).
 *   corr[r, q] = < in_patch(q), ref_patch(r) / (||ref_patch(r)|| + 1e-5) >   (:62-67)
 *   max over r, lowest r on exact ties (torch CPU max; chunk merge uses strict '>' :74)
 *   max_val / (||in_patch(q)|| + 1e-5)                                        (:78-84)
 * Restated through the algebraically identical pixel Gram:
 *   G[p, s]   = sum_c fin[c,p] * fref[c,s]                 (fmaf chain, c ascending, from +0)
 *   raw[q, r] = sum_{dy,dx} G[q+(dy,dx), r+(dy,dx)]        (8 sequential fp32 adds, row-major)
 *   corr      = raw * inv_ref[r]                           (one fp32 multiply)
 *   (best, idx): v > best, or v == best and r < idx.
 * fin, fref: [C][h][w] already pixel-normalised; inv_ref, nrm_in from orc_patch_norm.
 * idx_out: int64 [(h-2)(w-2)] with value ry*(w-2)+rx; val_out fp32 (may be NULL).
 * ------------------------------------------------------------------------------------------ */
/* G[p, :] for the w pixels of input pixel row `prow` (the fmaf chains of the header comment) */
static void corr_gram_row(const float *fin, const float *fref, int C, int h, int w, int prow, float *Grow)
{
    const int HW = h * w;
#pragma omp parallel for schedule(static)
    for (int px = 0; px < w; ++px) {
        float *g = Grow + (size_t)px * HW;
        const int p = prow * w + px;
        for (int s0 = 0; s0 < HW; s0 += 2048) {
            int s1 = s0 + 2048 < HW ? s0 + 2048 : HW;
            for (int s = s0; s < s1; ++s) g[s] = 0.0f;
            for (int c = 0; c < C; ++c) {
                const float a = fin[(size_t)c * HW + p];
                const float *b = fref + (size_t)c * HW;
                for (int s = s0; s < s1; ++s) g[s] = fmaf(a, b[s], g[s]);
            }
        }
    }
}

/* the queries of patch row qy from the Gram rows of input pixel rows qy, qy + 1, qy + 2: 8 sequential adds row-major, one multiply,
 * running maximum with the lowest index on ties */
static void corr_scan_row(const float *G0, const float *G1, const float *G2, int h, int w, const float *inv_ref,
                          const float *nrm_row /* nrm_in_eps + qy * pw */, int64_t *idx_row, float *val_row)
{
    const int HW = h * w, ph = h - 2, pw = w - 2;
#pragma omp parallel for schedule(static)
    for (int qx = 0; qx < pw; ++qx) {
        float bv = -INFINITY;
        int64_t bi = 0;
        const float *g00 = G0 + (size_t)(qx + 0) * HW, *g01 = G0 + (size_t)(qx + 1) * HW,
                    *g02 = G0 + (size_t)(qx + 2) * HW;
        const float *g10 = G1 + (size_t)(qx + 0) * HW, *g11 = G1 + (size_t)(qx + 1) * HW,
                    *g12 = G1 + (size_t)(qx + 2) * HW;
        const float *g20 = G2 + (size_t)(qx + 0) * HW, *g21 = G2 + (size_t)(qx + 1) * HW,
                    *g22 = G2 + (size_t)(qx + 2) * HW;
        for (int ry = 0; ry < ph; ++ry)
            for (int rx = 0; rx < pw; ++rx) {
                const int s = ry * w + rx;
                float v = g00[s];
                v = v + g01[s + 1];
                v = v + g02[s + 2];
                v = v + g10[s + w];
                v = v + g11[s + w + 1];
                v = v + g12[s + w + 2];
                v = v + g20[s + 2 * w];
                v = v + g21[s + 2 * w + 1];
                v = v + g22[s + 2 * w + 2];
                const int64_t r = (int64_t)ry * pw + rx;
                v = v * inv_ref[r];
                if (v > bv) { bv = v; bi = r; } /* ascending r: strict '>' keeps lowest index */
            }
        idx_row[qx] = bi;
        if (val_row) val_row[qx] = bv / nrm_row[qx];
    }
}

ORC_API void orc_corr_top1(const float *fin, const float *fref, int C, int h, int w,
                           const float *inv_ref, const float *nrm_in_eps,
                           int64_t *idx_out, float *val_out)
{
    const int HW = h * w, pw = w - 2;
    /* rolling window: G rows for 3 pixel rows of the input (3*w pixels x HW) */
    float *G = (float *)malloc((size_t)3 * w * HW * sizeof(float));
    for (int prow = 0; prow < h; ++prow) {
        corr_gram_row(fin, fref, C, h, w, prow, G + (size_t)(prow % 3) * w * HW);
        if (prow < 2) continue;
        const int qy = prow - 2;
        corr_scan_row(G + (size_t)((qy + 0) % 3) * w * HW, G + (size_t)((qy + 1) % 3) * w * HW, G + (size_t)((qy + 2) % 3) * w * HW, h, w,
                      inv_ref, nrm_in_eps + (size_t)qy * pw, idx_out + (size_t)qy * pw, val_out ? val_out + (size_t)qy * pw : NULL);
    }
    free(G);
}

/* The same for a subset of the patch rows (tests at sizes where the whole map takes a minute per pair: BASELINE configs[4]):
 * rows[k] = qy; idx_out / val_out are [n_rows][w - 2].  Identical arithmetic: the same two helpers. */
ORC_API void orc_corr_top1_rows(const float *fin, const float *fref, int C, int h, int w, const float *inv_ref, const float *nrm_in_eps,
                                const int *rows, int n_rows, int64_t *idx_out, float *val_out)
{
    const int HW = h * w, pw = w - 2;
    float *G = (float *)malloc((size_t)3 * w * HW * sizeof(float));
    for (int k = 0; k < n_rows; ++k) {
        const int qy = rows[k];
        for (int j = 0; j < 3; ++j) corr_gram_row(fin, fref, C, h, w, qy + j, G + (size_t)j * w * HW);
        corr_scan_row(G, G + (size_t)w * HW, G + (size_t)2 * w * HW, h, w, inv_ref, nrm_in_eps + (size_t)qy * pw, idx_out + (size_t)k * pw,
                      val_out ? val_out + (size_t)k * pw : NULL);
    }
    free(G);
}

/* ------------------------------------------------------------------------------------------
 * feature_match_index in its general form: ref_map_util.py:26-86 with any patch_size, input_stride, ref_stride and maps of
 * different sizes (sample_patches :4-23 = unfold(1, p, s).unfold(2, p, s), row-major patches).  Same defined order as
 * orc_corr_top1: per tap t (row-major in the p x p window) g_t = fmaf chain over c ascending, raw = g_0 + g_1 + ...,
 * corr = raw * inv[r] with inv[r] = 1 / (sqrtf(sum over the window of the per-pixel sums of squares, row-major) + 1e-5f)
 * when is_norm (:62-63), max over r with the lowest index on ties (:69-76), max_val / (||in patch|| + 1e-5f) when
 * norm_input (:78-84).  fin [C][h][w], fref [C][hr][wr]; idx_out / val_out [nqy*nqx].
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_feature_match_index(const float *fin, const float *fref, int C, int h, int w, int hr, int wr, int P, int si,
                                     int sr, int is_norm, int norm_input, int64_t *idx_out, float *val_out)
{
    const int nqy = (h - P) / si + 1, nqx = (w - P) / si + 1, nry = (hr - P) / sr + 1, nrx = (wr - P) / sr + 1;
    const size_t HW = (size_t)h * w, HWr = (size_t)hr * wr;
    float *n2i = (float *)malloc(HW * sizeof(float)), *n2r = (float *)malloc(HWr * sizeof(float));
    float *inv = (float *)malloc((size_t)nry * nrx * sizeof(float));
    orc_sumsq(fin, C, (int)HW, n2i);
    orc_sumsq(fref, C, (int)HWr, n2r);
    for (int ry = 0; ry < nry; ++ry)
        for (int rx = 0; rx < nrx; ++rx) {
            const float *m = n2r + (size_t)(ry * sr) * wr + rx * sr;
            float s = m[0];
            for (int t = 1; t < P * P; ++t) s = s + m[(t / P) * wr + t % P];
            inv[ry * nrx + rx] = is_norm ? 1.0f / (sqrtf(s) + 1e-5f) : 1.0f;
        }
#pragma omp parallel for schedule(dynamic)
    for (int q = 0; q < nqy * nqx; ++q) {
        const int qy = q / nqx, qx = q % nqx;
        float bv = -INFINITY;
        int64_t bi = 0;
        for (int r = 0; r < nry * nrx; ++r) {
            const int ry = r / nrx, rx = r % nrx;
            float v = 0.0f;
            for (int t = 0; t < P * P; ++t) {
                const float *a = fin + (size_t)(qy * si + t / P) * w + qx * si + t % P;
                const float *b = fref + (size_t)(ry * sr + t / P) * wr + rx * sr + t % P;
                float g = 0.0f;
                for (int c = 0; c < C; ++c) g = fmaf(a[(size_t)c * HW], b[(size_t)c * HWr], g);
                v = t == 0 ? g : v + g;
            }
            v = v * inv[r];
            if (v > bv) { bv = v; bi = r; }
        }
        idx_out[q] = bi;
        if (val_out) {
            if (norm_input) {
                const float *m = n2i + (size_t)(qy * si) * w + qx * si;
                float s = m[0];
                for (int t = 1; t < P * P; ++t) s = s + m[(t / P) * w + t % P];
                bv = bv / (sqrtf(s) + 1e-5f);
            }
            val_out[q] = bv;
        }
    }
    free(n2i); free(n2r); free(inv);
}

/* Exact (fp64) correlation of one query against one ref patch, reference operation order
 * (ref_map_util.py:62-67: normalise the ref patch first, then the dot product).  Used by tests
 * to classify any disagreement with the reference's own fp32 result as a sub-rounding near-tie. */
ORC_API double orc_corr_pair_f64(const float *fin, const float *fref, int C, int h, int w,
                                 int q, int r)
{
    const int HW = h * w, pw = w - 2;
    const int qy = q / pw, qx = q % pw, ry = r / pw, rx = r % pw;
    double nn = 0.0, dot = 0.0;
    for (int c = 0; c < C; ++c)
        for (int t = 0; t < 9; ++t) {
            double b = fref[(size_t)c * HW + (ry + t / 3) * w + rx + t % 3];
            double a = fin[(size_t)c * HW + (qy + t / 3) * w + qx + t % 3];
            nn += b * b;
            dot += a * b;
        }
    return dot / (sqrt(nn) + 1e-5);
}

/* ------------------------------------------------------------------------------------------
 * index -> flow -> 9 shifted offset planes at 3 scales.
 * Reference: corres_generation_arch.py:30-47 (index_to_flow) and :70-105 (+ arch_util.py:386-410
 * tensor_shift).  flow_x = idx % (w-2) - x, flow_y = idx // (w-2) - y, zero-padded by 2 at the
 * bottom/right to (h, w); scale s in {1,2,4}: repeat_interleave s times in both dims, * s, then
 * plane k = 3*i+j is the map shifted down/right by (i*s, j*s) with zero fill.
 * Output layout per scale: [9][s*h][s*w][2], last dim [x, y]  (one batch element).
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_offsets_from_idx(const int64_t *idx, int h, int w, float *off_s1, float *off_s2,
                                  float *off_s4)
{
    const int ph = h - 2, pw = w - 2;
    float *outs[3] = {off_s1, off_s2, off_s4};
    for (int si = 0; si < 3; ++si) {
        const int s = 1 << si, H = h * s, W = w * s;
        float *out = outs[si];
        if (!out) continue;
        for (int k = 0; k < 9; ++k) {
            const int sh = (k / 3) * s, sw = (k % 3) * s;
            for (int Y = 0; Y < H; ++Y)
                for (int X = 0; X < W; ++X) {
                    float fx = 0.0f, fy = 0.0f;
                    const int ys = Y - sh, xs = X - sw; /* source position in the scaled map */
                    if (ys >= 0 && xs >= 0) {
                        const int y = ys / s, x = xs / s; /* repeat_interleave */
                        if (y < ph && x < pw) {
                            const int64_t m = idx[(size_t)y * pw + x];
                            fx = (float)((m % pw) - x) * (float)s;
                            fy = (float)((m / pw) - y) * (float)s;
                        }
                    }
                    float *o = out + (((size_t)k * H + Y) * W + X) * 2;
                    o[0] = fx;
                    o[1] = fy;
                }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * DCNv2 (modulated deformable convolution), forward and backward.
 * Spec: basicsr/ops/dcn/src/deform_conv_cuda_kernel.cu:467-497 (bilinear), :570-633 (im2col),
 *       :635-693 (col2im), :695-767 (coord), host deform_conv_cuda.cpp:490-685.
 * Layouts: x [B][C][H][W]; offset [B][dg*2*kh*kw][Ho][Wo] ordered [g][tap][y,x];
 *          mask [B][dg*kh*kw][Ho][Wo]; weight [Co][C/groups][kh][kw]; bias [Co] or NULL.
 * Accumulation in double (the checker should be more accurate than either implementation).
 * ------------------------------------------------------------------------------------------ */
static inline float dcn_bilinear(const float *im, int H, int W, float h, float w)
{
    int hl = (int)floorf(h), wl = (int)floorf(w);
    int hh = hl + 1, wh = wl + 1;
    float lh = h - hl, lw = w - wl, uh = 1 - lh, uw = 1 - lw;
    float v1 = (hl >= 0 && wl >= 0) ? im[hl * W + wl] : 0.f;
    float v2 = (hl >= 0 && wh <= W - 1) ? im[hl * W + wh] : 0.f;
    float v3 = (hh <= H - 1 && wl >= 0) ? im[hh * W + wl] : 0.f;
    float v4 = (hh <= H - 1 && wh <= W - 1) ? im[hh * W + wh] : 0.f;
    return uh * uw * v1 + uh * lw * v2 + lh * uw * v3 + lh * lw * v4;
}

ORC_API void orc_dcnv2_fwd(const float *x, const float *offset, const float *mask,
                           const float *weight, const float *bias, float *out, int B, int C, int H,
                           int W, int Co, int kh, int kw, int stride, int pad, int dil, int groups,
                           int dg)
{
    const int Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1;
    const int Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) / stride + 1;
    const int cpg = C / dg, cig = C / groups, cog = Co / groups, KK = kh * kw;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int ho = 0; ho < Ho; ++ho) {
            float *col = (float *)malloc((size_t)C * KK * sizeof(float));
            for (int wo = 0; wo < Wo; ++wo) {
                for (int c = 0; c < C; ++c) {
                    const int g = c / cpg;
                    const float *im = x + ((size_t)b * C + c) * H * W;
                    for (int t = 0; t < KK; ++t) {
                        const int i = t / kw, j = t % kw;
                        const float oh = offset[(((size_t)b * dg + g) * 2 * KK + 2 * t) * Ho * Wo + (size_t)ho * Wo + wo];
                        const float ow = offset[(((size_t)b * dg + g) * 2 * KK + 2 * t + 1) * Ho * Wo + (size_t)ho * Wo + wo];
                        const float m = mask ? mask[(((size_t)b * dg + g) * KK + t) * Ho * Wo + (size_t)ho * Wo + wo] : 1.f;
                        const float hi = ho * stride - pad + i * dil + oh;
                        const float wi = wo * stride - pad + j * dil + ow;
                        float v = 0.f;
                        if (hi > -1 && wi > -1 && hi < H && wi < W) v = dcn_bilinear(im, H, W, hi, wi);
                        col[c * KK + t] = v * m;
                    }
                }
                for (int o = 0; o < Co; ++o) {
                    const int gr = o / cog;
                    double acc = bias ? bias[o] : 0.0;
                    const float *wr = weight + (size_t)o * cig * KK;
                    const float *cr = col + (size_t)gr * cig * KK;
                    for (int k = 0; k < cig * KK; ++k) acc += (double)wr[k] * cr[k];
                    out[(((size_t)b * Co + o) * Ho + ho) * Wo + wo] = (float)acc;
                }
            }
            free(col);
        }
}

/* columns[B][C*kh*kw][Ho*Wo] = mask * bilinear(x)  (modulated_deformable_im2col, kernel.cu:570-633).
 * The CPU port's DCN forward = this + one GEMM per sample, i.e. the structure of
 * deform_conv_cuda.cpp:539-561 (used by oracle/pipeline.py for the timed CPU baseline). */
ORC_API void orc_dcnv2_im2col(const float *x, const float *offset, const float *mask, float *col,
                              int B, int C, int H, int W, int kh, int kw, int stride, int pad,
                              int dil, int dg)
{
    const int Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1;
    const int Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) / stride + 1;
    const int cpg = C / dg, KK = kh * kw;
    const size_t HWo = (size_t)Ho * Wo;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            const int g = c / cpg;
            const float *im = x + ((size_t)b * C + c) * H * W;
            for (int t = 0; t < KK; ++t) {
                const int i = t / kw, j = t % kw;
                const float *oh = offset + (((size_t)b * dg + g) * 2 * KK + 2 * t) * HWo;
                const float *ow = oh + HWo;
                const float *mk = mask ? mask + (((size_t)b * dg + g) * KK + t) * HWo : NULL;
                float *dst = col + (((size_t)b * C + c) * KK + t) * HWo;
                for (int ho = 0; ho < Ho; ++ho)
                    for (int wo = 0; wo < Wo; ++wo) {
                        const size_t p = (size_t)ho * Wo + wo;
                        const float hi = ho * stride - pad + i * dil + oh[p];
                        const float wi = wo * stride - pad + j * dil + ow[p];
                        float v = 0.f;
                        if (hi > -1 && wi > -1 && hi < H && wi < W) v = dcn_bilinear(im, H, W, hi, wi);
                        dst[p] = v * (mk ? mk[p] : 1.f);
                    }
            }
        }
}

/* Backward.  Any of the grad outputs may be NULL.  grad_x/grad_w/grad_b are ACCUMULATED into
 * (caller zeroes them, as deform_conv.py:161-165 does); grad_offset/grad_mask are assigned. */
ORC_API void orc_dcnv2_bwd(const float *x, const float *offset, const float *mask,
                           const float *weight, const float *gout, float *grad_x,
                           float *grad_offset, float *grad_mask, float *grad_w, float *grad_b,
                           int B, int C, int H, int W, int Co, int kh, int kw, int stride, int pad,
                           int dil, int groups, int dg)
{
    const int Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1;
    const int Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) / stride + 1;
    const int cpg = C / dg, cig = C / groups, cog = Co / groups, KK = kh * kw;
    double *gw = grad_w ? (double *)calloc((size_t)Co * cig * KK, sizeof(double)) : NULL;
    double *gb = grad_b ? (double *)calloc((size_t)Co, sizeof(double)) : NULL;
    double *gx = grad_x ? (double *)calloc((size_t)B * C * H * W, sizeof(double)) : NULL;
    float *col = (float *)malloc((size_t)C * KK * sizeof(float));
    double *gcol = (double *)malloc((size_t)C * KK * sizeof(double));
    for (int b = 0; b < B; ++b)
        for (int ho = 0; ho < Ho; ++ho)
            for (int wo = 0; wo < Wo; ++wo) {
                const size_t pix = (size_t)ho * Wo + wo;
                /* grad_col = W^T gout */
                for (int k = 0; k < C * KK; ++k) gcol[k] = 0.0;
                for (int o = 0; o < Co; ++o) {
                    const int gr = o / cog;
                    const double go = gout[((size_t)b * Co + o) * Ho * Wo + pix];
                    if (gb) gb[o] += go;
                    const float *wr = weight + (size_t)o * cig * KK;
                    for (int k = 0; k < cig * KK; ++k) gcol[(size_t)gr * cig * KK + k] += go * wr[k];
                }
                for (int g = 0; g < dg; ++g)
                    for (int t = 0; t < KK; ++t) {
                        const int i = t / kw, j = t % kw;
                        const size_t oi = (((size_t)b * dg + g) * 2 * KK + 2 * t) * Ho * Wo + pix;
                        const size_t mi = (((size_t)b * dg + g) * KK + t) * Ho * Wo + pix;
                        const float oh = offset[oi], ow = offset[oi + (size_t)Ho * Wo];
                        const float m = mask ? mask[mi] : 1.f;
                        const float hi = ho * stride - pad + i * dil + oh;
                        const float wi = wo * stride - pad + j * dil + ow;
                        const int inside = (hi > -1 && wi > -1 && hi < H && wi < W);
                        double g_oh = 0.0, g_ow = 0.0, g_m = 0.0;
                        const int hl = (int)floorf(hi), wl = (int)floorf(wi), hh = hl + 1, wh = wl + 1;
                        const float lh = hi - hl, lw = wi - wl, uh = 1 - lh, uw = 1 - lw;
                        for (int cc = 0; cc < cpg; ++cc) {
                            const int c = g * cpg + cc;
                            const float *im = x + ((size_t)b * C + c) * H * W;
                            float v = 0.f;
                            const double gc = gcol[(size_t)c * KK + t];
                            if (inside) {
                                const float v1 = (hl >= 0 && wl >= 0) ? im[hl * W + wl] : 0.f;
                                const float v2 = (hl >= 0 && wh <= W - 1) ? im[hl * W + wh] : 0.f;
                                const float v3 = (hh <= H - 1 && wl >= 0) ? im[hh * W + wl] : 0.f;
                                const float v4 = (hh <= H - 1 && wh <= W - 1) ? im[hh * W + wh] : 0.f;
                                v = uh * uw * v1 + uh * lw * v2 + lh * uw * v3 + lh * lw * v4;
                                /* d val / d h, d val / d w  (kernel.cu:526-568) */
                                g_oh += gc * m * ((double)(-uw) * v1 - (double)lw * v2 + (double)uw * v3 + (double)lw * v4);
                                g_ow += gc * m * ((double)(-uh) * v1 + (double)uh * v2 - (double)lh * v3 + (double)lh * v4);
                                g_m += gc * v;
                                if (gx) {
                                    double *gi = gx + ((size_t)b * C + c) * H * W;
                                    const double gv = gc * m;
                                    if (hl >= 0 && wl >= 0) gi[hl * W + wl] += gv * uh * uw;
                                    if (hl >= 0 && wh <= W - 1) gi[hl * W + wh] += gv * uh * lw;
                                    if (hh <= H - 1 && wl >= 0) gi[hh * W + wl] += gv * lh * uw;
                                    if (hh <= H - 1 && wh <= W - 1) gi[hh * W + wh] += gv * lh * lw;
                                }
                            }
                            col[c * KK + t] = v * m;
                        }
                        if (grad_offset) {
                            grad_offset[oi] = (float)g_oh;
                            grad_offset[oi + (size_t)Ho * Wo] = (float)g_ow;
                        }
                        if (grad_mask && mask) grad_mask[mi] = (float)g_m;
                    }
                if (gw)
                    for (int o = 0; o < Co; ++o) {
                        const int gr = o / cog;
                        const double go = gout[((size_t)b * Co + o) * Ho * Wo + pix];
                        double *gwr = gw + (size_t)o * cig * KK;
                        const float *cr = col + (size_t)gr * cig * KK;
                        for (int k = 0; k < cig * KK; ++k) gwr[k] += go * cr[k];
                    }
            }
    if (gw) { for (size_t k = 0; k < (size_t)Co * cig * KK; ++k) grad_w[k] += (float)gw[k]; free(gw); }
    if (gb) { for (int o = 0; o < Co; ++o) grad_b[o] += (float)gb[o]; free(gb); }
    if (gx) { for (size_t k = 0; k < (size_t)B * C * H * W; ++k) grad_x[k] += (float)gx[k]; free(gx); }
    free(col); free(gcol);
}

/* ------------------------------------------------------------------------------------------
 * Multi-reference feature-transfer attention core.
 * Reference: ref_mrapa_restoration_arch.py:321-335.
 *   s_t = < q[:,p], emb_t[:,p] >;  a = softmax_t(s);  out[:,p] = sum_t a_t * ass_t[:,p]
 * q [N][c][HW] (already scaled by c^-1/2, :322), emb [N][T][c][HW], ass [N][T][c2][HW],
 * out [N][c2][HW].
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_mrattn_fwd(const float *q, const float *emb, const float *ass, float *out,
                            float *prob, int N, int T, int c, int c2, int HW)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int p = 0; p < HW; ++p) {
            double s[64], mx = -1e300, den = 0.0;
            for (int t = 0; t < T; ++t) {
                double a = 0.0;
                for (int k = 0; k < c; ++k)
                    a += (double)q[((size_t)n * c + k) * HW + p] * emb[(((size_t)n * T + t) * c + k) * HW + p];
                s[t] = a;
                if (a > mx) mx = a;
            }
            for (int t = 0; t < T; ++t) { s[t] = exp(s[t] - mx); den += s[t]; }
            for (int t = 0; t < T; ++t) {
                s[t] /= den;
                if (prob) prob[((size_t)n * T + t) * HW + p] = (float)s[t];
            }
            for (int k = 0; k < c2; ++k) {
                double a = 0.0;
                for (int t = 0; t < T; ++t) a += s[t] * ass[(((size_t)n * T + t) * c2 + k) * HW + p];
                out[((size_t)n * c2 + k) * HW + p] = (float)a;
            }
        }
}

ORC_API void orc_mrattn_bwd(const float *q, const float *emb, const float *ass, const float *gout,
                            float *gq, float *gemb, float *gass, int N, int T, int c, int c2, int HW)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int p = 0; p < HW; ++p) {
            double s[64], da[64], mx = -1e300, den = 0.0, dot = 0.0;
            for (int t = 0; t < T; ++t) {
                double a = 0.0;
                for (int k = 0; k < c; ++k)
                    a += (double)q[((size_t)n * c + k) * HW + p] * emb[(((size_t)n * T + t) * c + k) * HW + p];
                s[t] = a;
                if (a > mx) mx = a;
            }
            for (int t = 0; t < T; ++t) { s[t] = exp(s[t] - mx); den += s[t]; }
            for (int t = 0; t < T; ++t) s[t] /= den;
            for (int t = 0; t < T; ++t) {
                double a = 0.0;
                for (int k = 0; k < c2; ++k) {
                    const double g = gout[((size_t)n * c2 + k) * HW + p];
                    a += g * ass[(((size_t)n * T + t) * c2 + k) * HW + p];
                    gass[(((size_t)n * T + t) * c2 + k) * HW + p] = (float)(g * s[t]);
                }
                da[t] = a;
                dot += a * s[t];
            }
            for (int k = 0; k < c; ++k) {
                double a = 0.0;
                for (int t = 0; t < T; ++t) {
                    const double ds = s[t] * (da[t] - dot);
                    a += ds * emb[(((size_t)n * T + t) * c + k) * HW + p];
                    gemb[(((size_t)n * T + t) * c + k) * HW + p] = (float)(ds * q[((size_t)n * c + k) * HW + p]);
                }
                gq[((size_t)n * c + k) * HW + p] = (float)a;
            }
        }
}

/* ------------------------------------------------------------------------------------------
 * fused bias + activation.  Spec: basicsr/ops/fused_act/src/fused_bias_act_kernel.cu:19-50.
 * act*10+grad: 10,11 linear; 12,32 zero; 30 lrelu fwd; 31 lrelu bwd on sign of ref.
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_fused_bias_act(const float *x, const float *b, const float *ref, float *out,
                                int64_t size_x, int step_b, int size_b, int act, int grad,
                                float alpha, float scale)
{
    for (int64_t i = 0; i < size_x; ++i) {
        float v = x[i];
        if (b) v += b[(i / step_b) % size_b];
        const float r = ref ? ref[i] : 0.f;
        float y;
        switch (act * 10 + grad) {
        default:
        case 10: case 11: y = v; break;
        case 12: case 32: y = 0.f; break;
        case 30: y = (v > 0.f) ? v : v * alpha; break;
        case 31: y = (r > 0.f) ? v : v * alpha; break;
        }
        out[i] = y * scale;
    }
}

/* ------------------------------------------------------------------------------------------
 * upfirdn2d.  Spec: basicsr/ops/upfirdn2d/upfirdn2d.py:162-192 (upfirdn2d_native) and
 * src/upfirdn2d_kernel.cu:50-106.  input [major][in_h][in_w][minor], kernel [kh][kw].
 * ------------------------------------------------------------------------------------------ */
static inline int floor_div(int a, int b) { int c = a / b; if (c * b > a) c--; return c; }

ORC_API void orc_upfirdn2d(const float *in, const float *k, float *out, int major, int in_h,
                           int in_w, int minor, int kh, int kw, int up_x, int up_y, int down_x,
                           int down_y, int px0, int px1, int py0, int py1)
{
    const int out_h = (in_h * up_y + py0 + py1 - kh + down_y) / down_y;
    const int out_w = (in_w * up_x + px0 + px1 - kw + down_x) / down_x;
    for (int mj = 0; mj < major; ++mj)
        for (int oy = 0; oy < out_h; ++oy)
            for (int ox = 0; ox < out_w; ++ox)
                for (int mn = 0; mn < minor; ++mn) {
                    double v = 0.0;
                    /* out = sum_{ky,kx} upsampled_padded[oy*down + ky][ox*down + kx] * k_flipped */
                    for (int ky = 0; ky < kh; ++ky) {
                        const int uy = oy * down_y + ky - py0; /* position in the zero-inserted map */
                        if (uy < 0 || uy % up_y) continue;
                        const int iy = uy / up_y;
                        if (iy >= in_h) continue;
                        for (int kx = 0; kx < kw; ++kx) {
                            const int ux = ox * down_x + kx - px0;
                            if (ux < 0 || ux % up_x) continue;
                            const int ix = ux / up_x;
                            if (ix >= in_w) continue;
                            v += (double)in[(((size_t)mj * in_h + iy) * in_w + ix) * minor + mn] *
                                 k[(kh - 1 - ky) * kw + (kw - 1 - kx)];
                        }
                    }
                    out[(((size_t)mj * out_h + oy) * out_w + ox) * minor + mn] = (float)v;
                }
    (void)floor_div;
}

ORC_API void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

ORC_API int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
