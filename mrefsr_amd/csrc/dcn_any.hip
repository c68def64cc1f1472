// DCNv2 / DCNv1 for the other dtypes of the reference's dispatch (AT_DISPATCH_FLOATING_TYPES_AND_HALF,
// deform_conv_cuda_kernel.cu:259,353,451,781,813,846): portable kernels templated on the storage type, any stride /
// dilation / groups / kernel size / channel count -- the arithmetic of deform_conv_cuda_kernel.cu:467-497, :570-633, :635-767
// restated once for T = float / double / __half, with float accumulation (double for T = double).  The fp32 fast paths
// (fused gather + MFMA) stay in dcn.hip; these are what `mrefsr_dcn_fwd_f32` falls back to for shapes outside them, offered
// for every dtype: one thread = one output pixel x 16 output channels, sampling on the fly.
#include <hip/hip_fp16.h>

#include "common.h"

namespace {

struct GeoA {
    int B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, groups, dg, Ho, Wo;
};

int make_geo_a(const mrefsr_dcn_shape *s, GeoA &g, const char *who)
{
    if (!s) return mrefsr::fail(MREFSR_E_INVALID, "%s: null shape", who);
    g = GeoA{s->B, s->C, s->H, s->W, s->Co, s->kh, s->kw, s->stride_h, s->stride_w, s->pad_h, s->pad_w, s->dil_h, s->dil_w, s->groups, s->dg, 0, 0};
    if (g.B <= 0 || g.C <= 0 || g.H <= 0 || g.W <= 0 || g.Co <= 0 || g.kh <= 0 || g.kw <= 0 || g.sh <= 0 || g.sw <= 0 || g.dh <= 0 ||
        g.dw <= 0 || g.groups <= 0 || g.dg <= 0 || g.ph < 0 || g.pw < 0)
        return mrefsr::fail(MREFSR_E_INVALID, "%s: non-positive dimension in shape", who);
    if (g.C % g.groups || g.Co % g.groups || g.C % g.dg)
        return mrefsr::fail(MREFSR_E_INVALID, "%s: C=%d / Co=%d not divisible by groups=%d / dg=%d", who, g.C, g.Co, g.groups, g.dg);
    g.Ho = (g.H + 2 * g.ph - (g.dh * (g.kh - 1) + 1)) / g.sh + 1;
    g.Wo = (g.W + 2 * g.pw - (g.dw * (g.kw - 1) + 1)) / g.sw + 1;
    if (g.Ho <= 0 || g.Wo <= 0) return mrefsr::fail(MREFSR_E_INVALID, "%s: empty output %dx%d", who, g.Ho, g.Wo);
    return 0;
}

template <typename T> struct AccOf { typedef float type; };
template <> struct AccOf<double> { typedef double type; };
template <typename T> __device__ __forceinline__ typename AccOf<T>::type ldt(const T *p, size_t i) { return (typename AccOf<T>::type)p[i]; }
template <> __device__ __forceinline__ float ldt<__half>(const __half *p, size_t i) { return __half2float(p[i]); }
template <typename T, typename A> __device__ __forceinline__ void stt(T *p, size_t i, A v) { p[i] = (T)v; }
template <> __device__ __forceinline__ void stt<__half, float>(__half *p, size_t i, float v) { p[i] = __float2half(v); }

// bilinear setup: clamped corner offsets, corner validity (position in range and sample inside (-1,H) x (-1,W)), fractions
template <typename A>
struct Bil {
    int o[4];
    bool v[4];
    A lh, lw;
    __device__ __forceinline__ Bil(A hi, A wi, int H, int W)
    {
        const bool inside = (hi > (A)-1) && (wi > (A)-1) && (hi < (A)H) && (wi < (A)W);
        const A fh = floor(hi), fw = floor(wi);
        const int hl = (int)fh, wl = (int)fw, hh = hl + 1, wh = wl + 1;
        lh = hi - fh;
        lw = wi - fw;
        const int chl = min(max(hl, 0), H - 1), chh = min(max(hh, 0), H - 1), cwl = min(max(wl, 0), W - 1), cwh = min(max(wh, 0), W - 1);
        o[0] = chl * W + cwl, o[1] = chl * W + cwh, o[2] = chh * W + cwl, o[3] = chh * W + cwh;
        v[0] = inside && hl >= 0 && wl >= 0, v[1] = inside && hl >= 0 && wh <= W - 1;
        v[2] = inside && hh <= H - 1 && wl >= 0, v[3] = inside && hh <= H - 1 && wh <= W - 1;
    }
    __device__ __forceinline__ A w(int k) const
    {
        const A uh = (A)1 - lh, uw = (A)1 - lw;
        return k == 0 ? uh * uw : k == 1 ? uh * lw : k == 2 ? lh * uw : lh * lw;
    }
    template <typename T>
    __device__ __forceinline__ A sample(const T *im) const
    {
        A s = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (v[k]) s += w(k) * ldt(im, (size_t)o[k]);
        return s;
    }
};

template <typename T>
__global__ __launch_bounds__(256) void dcn_fwd_any_kernel(const T *__restrict__ x, const T *__restrict__ offset, const T *__restrict__ mask,
                                                          const T *__restrict__ wgt, const T *__restrict__ bias, T *__restrict__ out, GeoA g,
                                                          float slope)
{
    typedef typename AccOf<T>::type A;
    const int HWo = g.Ho * g.Wo, HWi = g.H * g.W, KK = g.kh * g.kw;
    const int cig = g.C / g.groups, cog = g.Co / g.groups, cpg = g.C / g.dg;
    const int otiles = (cog + 15) / 16;  // 16-wide output tiles never straddle a conv group
    const long total = (long)g.B * g.groups * otiles * HWo;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int pix = (int)(e % HWo);
        long t = e / HWo;
        const int ot = (int)(t % otiles);
        t /= otiles;
        const int gr = (int)(t % g.groups), b = (int)(t / g.groups);
        const int ho = pix / g.Wo, wo = pix - ho * g.Wo;
        const int o0 = gr * cog + ot * 16;
        const int no = min(16, gr * cog + cog - o0);
        A acc[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = 0;
        for (int cc = 0; cc < cig; ++cc) {
            const int c = gr * cig + cc, dgi = c / cpg;
            const T *im = x + ((size_t)b * g.C + c) * HWi;
            for (int tap = 0; tap < KK; ++tap) {
                const int ti = tap / g.kw, tj = tap - ti * g.kw;
                const size_t oi = (((size_t)b * g.dg + dgi) * 2 * KK + 2 * tap) * HWo + pix;
                const A oh = ldt(offset, oi), ow = ldt(offset, oi + HWo);
                const A m = mask ? ldt(mask, (((size_t)b * g.dg + dgi) * KK + tap) * HWo + pix) : (A)1;
                const Bil<A> bl((A)(ho * g.sh - g.ph + ti * g.dh) + oh, (A)(wo * g.sw - g.pw + tj * g.dw) + ow, g.H, g.W);
                const A v = bl.sample(im) * m;
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if (k < no) acc[k] += ldt(wgt, ((size_t)(o0 + k) * cig + cc) * KK + tap) * v;
            }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (k < no) {
                A v = acc[k] + (bias ? ldt(bias, (size_t)(o0 + k)) : (A)0);
                v = v > 0 ? v : v * (A)slope;
                stt(out, ((size_t)b * g.Co + o0 + k) * HWo + pix, v);
            }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void dcn_im2col_any_kernel(const T *__restrict__ x, const T *__restrict__ offset, const T *__restrict__ mask,
                                                             T *__restrict__ col, GeoA g)
{
    typedef typename AccOf<T>::type A;
    const int HWo = g.Ho * g.Wo, HWi = g.H * g.W, KK = g.kh * g.kw, cpg = g.C / g.dg;
    const long total = (long)g.B * g.C * HWo;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int pix = (int)(e % HWo);
        const long t = e / HWo;
        const int c = (int)(t % g.C), b = (int)(t / g.C);
        const int ho = pix / g.Wo, wo = pix - ho * g.Wo, dgi = c / cpg;
        const T *im = x + ((size_t)b * g.C + c) * HWi;
        for (int tap = 0; tap < KK; ++tap) {
            const int ti = tap / g.kw, tj = tap - ti * g.kw;
            const size_t oi = (((size_t)b * g.dg + dgi) * 2 * KK + 2 * tap) * HWo + pix;
            const A oh = ldt(offset, oi), ow = ldt(offset, oi + HWo);
            const A m = mask ? ldt(mask, (((size_t)b * g.dg + dgi) * KK + tap) * HWo + pix) : (A)1;
            const Bil<A> bl((A)(ho * g.sh - g.ph + ti * g.dh) + oh, (A)(wo * g.sw - g.pw + tj * g.dw) + ow, g.H, g.W);
            stt(col, (((size_t)b * g.C + c) * KK + tap) * HWo + pix, bl.sample(im) * m);
        }
    }
}

// one thread = (b, deformable group, tap, pixel): loops the group's channels; assigns grad_offset (y, x) and grad_mask,
// scatters grad_x with atomics (deform_conv_cuda_kernel.cu:635-767).  T = float / double (native atomics).
template <typename T>
__global__ __launch_bounds__(256) void dcn_col2im_any_kernel(const T *__restrict__ gcol, const T *__restrict__ x, const T *__restrict__ offset,
                                                             const T *__restrict__ mask, T *__restrict__ gx, T *__restrict__ goff,
                                                             T *__restrict__ gmask, GeoA g)
{
    typedef typename AccOf<T>::type A;
    const int HWo = g.Ho * g.Wo, HWi = g.H * g.W, KK = g.kh * g.kw, cpg = g.C / g.dg;
    const long total = (long)g.B * g.dg * KK * HWo;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int pix = (int)(e % HWo);
        long t = e / HWo;
        const int tap = (int)(t % KK);
        t /= KK;
        const int dgi = (int)(t % g.dg), b = (int)(t / g.dg);
        const int ho = pix / g.Wo, wo = pix - ho * g.Wo;
        const int ti = tap / g.kw, tj = tap - ti * g.kw;
        const size_t oi = (((size_t)b * g.dg + dgi) * 2 * KK + 2 * tap) * HWo + pix;
        const size_t mi = (((size_t)b * g.dg + dgi) * KK + tap) * HWo + pix;
        const A oh = ldt(offset, oi), ow = ldt(offset, oi + HWo);
        const A m = mask ? ldt(mask, mi) : (A)1;
        const Bil<A> bl((A)(ho * g.sh - g.ph + ti * g.dh) + oh, (A)(wo * g.sw - g.pw + tj * g.dw) + ow, g.H, g.W);
        const A uh = (A)1 - bl.lh, uw = (A)1 - bl.lw;
        A g_oh = 0, g_ow = 0, g_m = 0;
        for (int cc = 0; cc < cpg; ++cc) {
            const int c = dgi * cpg + cc;
            const A gc = ldt(gcol, (((size_t)b * g.C + c) * KK + tap) * HWo + pix);
            const T *im = x + ((size_t)b * g.C + c) * HWi;
            A xv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) xv[k] = bl.v[k] ? ldt(im, (size_t)bl.o[k]) : (A)0;
            g_m += gc * (bl.w(0) * xv[0] + bl.w(1) * xv[1] + bl.w(2) * xv[2] + bl.w(3) * xv[3]);
            g_oh += gc * m * (-uw * xv[0] - bl.lw * xv[1] + uw * xv[2] + bl.lw * xv[3]);
            g_ow += gc * m * (-uh * xv[0] + uh * xv[1] - bl.lh * xv[2] + bl.lh * xv[3]);
            if (gx) {
                T *gi = gx + ((size_t)b * g.C + c) * HWi;
                const A gv = gc * m;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (bl.v[k]) atomicAdd(gi + bl.o[k], (T)(gv * bl.w(k)));
            }
        }
        stt(goff, oi, g_oh);
        stt(goff, oi + HWo, g_ow);
        if (gmask && mask) stt(gmask, mi, g_m);
    }
}

inline dim3 blocks_for(long total) { const long b = (total + 255) / 256; return dim3((unsigned)(b < 65536 ? b : 65536)); }

}  // namespace

MREFSR_EXPORT int mrefsr_dcn_fwd(const void *x, const void *offset, const void *mask, const void *weight, const void *bias, void *out,
                                 const mrefsr_dcn_shape *s, float act_slope, int dtype, mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(x && offset && weight && out, "dcn_fwd: null pointer");
    MREFSR_REQUIRE(dtype == 0 || dtype == 1 || dtype == 3, "dcn_fwd: dtype=%d (0 f32, 1 f16, 3 f64)", dtype);
    GeoA g;
    if (int e = make_geo_a(s, g, "dcn_fwd")) return e;
    const long total = (long)g.B * g.groups * ((g.Co / g.groups + 15) / 16) * g.Ho * g.Wo;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0)
        hipLaunchKernelGGL(dcn_fwd_any_kernel<float>, blocks_for(total), dim3(256), 0, st, (const float *)x, (const float *)offset, (const float *)mask,
                           (const float *)weight, (const float *)bias, (float *)out, g, act_slope);
    else if (dtype == 1)
        hipLaunchKernelGGL(dcn_fwd_any_kernel<__half>, blocks_for(total), dim3(256), 0, st, (const __half *)x, (const __half *)offset,
                           (const __half *)mask, (const __half *)weight, (const __half *)bias, (__half *)out, g, act_slope);
    else
        hipLaunchKernelGGL(dcn_fwd_any_kernel<double>, blocks_for(total), dim3(256), 0, st, (const double *)x, (const double *)offset,
                           (const double *)mask, (const double *)weight, (const double *)bias, (double *)out, g, act_slope);
    return mrefsr::check_launch("dcn_fwd(any dtype)");
}

MREFSR_EXPORT int mrefsr_dcn_im2col(const void *x, const void *offset, const void *mask, void *columns, const mrefsr_dcn_shape *s, int dtype,
                                    mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(x && offset && columns, "dcn_im2col: null pointer");
    MREFSR_REQUIRE(dtype == 0 || dtype == 1 || dtype == 3, "dcn_im2col: dtype=%d (0 f32, 1 f16, 3 f64)", dtype);
    GeoA g;
    if (int e = make_geo_a(s, g, "dcn_im2col")) return e;
    const long total = (long)g.B * g.C * g.Ho * g.Wo;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0)
        hipLaunchKernelGGL(dcn_im2col_any_kernel<float>, blocks_for(total), dim3(256), 0, st, (const float *)x, (const float *)offset,
                           (const float *)mask, (float *)columns, g);
    else if (dtype == 1)
        hipLaunchKernelGGL(dcn_im2col_any_kernel<__half>, blocks_for(total), dim3(256), 0, st, (const __half *)x, (const __half *)offset,
                           (const __half *)mask, (__half *)columns, g);
    else
        hipLaunchKernelGGL(dcn_im2col_any_kernel<double>, blocks_for(total), dim3(256), 0, st, (const double *)x, (const double *)offset,
                           (const double *)mask, (double *)columns, g);
    return mrefsr::check_launch("dcn_im2col(any dtype)");
}

MREFSR_EXPORT int mrefsr_dcn_col2im(const void *grad_col, const void *x, const void *offset, const void *mask, void *grad_x, void *grad_offset,
                                    void *grad_mask, const mrefsr_dcn_shape *s, int dtype, mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(grad_col && x && offset && grad_offset, "dcn_col2im: null pointer");
    MREFSR_REQUIRE(dtype == 0 || dtype == 3, "dcn_col2im: dtype=%d (0 f32, 3 f64; an f16 caller accumulates in f32)", dtype);
    GeoA g;
    if (int e = make_geo_a(s, g, "dcn_col2im")) return e;
    const long total = (long)g.B * g.dg * g.kh * g.kw * g.Ho * g.Wo;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0)
        hipLaunchKernelGGL(dcn_col2im_any_kernel<float>, blocks_for(total), dim3(256), 0, st, (const float *)grad_col, (const float *)x,
                           (const float *)offset, (const float *)mask, (float *)grad_x, (float *)grad_offset, (float *)grad_mask, g);
    else
        hipLaunchKernelGGL(dcn_col2im_any_kernel<double>, blocks_for(total), dim3(256), 0, st, (const double *)grad_col, (const double *)x,
                           (const double *)offset, (const double *)mask, (double *)grad_x, (double *)grad_offset, (double *)grad_mask, g);
    return mrefsr::check_launch("dcn_col2im(any dtype)");
}
