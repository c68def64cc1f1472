// DCNv2 / DCNv1 for gfx950.
//
// Replaces basicsr/ops/dcn/src/{deform_conv_cuda.cpp,deform_conv_cuda_kernel.cu} and the
// mmcv.ops.modulated_deform_conv2d call of ref_mrapa_restoration_arch.py:74-76.  The reference
// forward is, per sample in a host loop, an im2col kernel that writes C*9*H*W floats to memory
// followed by a cuBLAS GEMM that reads them back (deform_conv_cuda.cpp:539-561).  Here the
// forward is ONE kernel: the deformable gather produces 32-row K-chunks of the column matrix
// straight into LDS (software-pipelined one chunk ahead), and v_mfma_f32_32x32x2_f32 consumes
// them against pre-packed weights; bias and the following LeakyReLU are fused into the epilogue.
// The column matrix never exists in HBM.
//
// Sampling semantics (deform_conv_cuda_kernel.cu:467-497, :570-633): position
// (ho*stride - pad + i*dil + off_y, wo*stride - pad + j*dil + off_x); contributes only if inside
// (-1,H) x (-1,W); bilinear with out-of-range corners = 0; value * mask.
#include <stdlib.h>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Geo {
    int B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, groups, dg, Ho, Wo;
};

__host__ int make_geo(const mrefsr_dcn_shape *s, Geo &g, const char *who)
{
    if (!s) return mrefsr::fail(MREFSR_E_INVALID, "%s: null shape", who);
    g = Geo{s->B, s->C, s->H, s->W, s->Co, s->kh, s->kw, s->stride_h, s->stride_w, s->pad_h, s->pad_w,
            s->dil_h, s->dil_w, s->groups, s->dg, 0, 0};
    if (g.B <= 0 || g.C <= 0 || g.H <= 0 || g.W <= 0 || g.Co <= 0 || g.kh <= 0 || g.kw <= 0 || g.sh <= 0 || g.sw <= 0 ||
        g.dh <= 0 || g.dw <= 0 || g.groups <= 0 || g.dg <= 0 || g.ph < 0 || g.pw < 0)
        return mrefsr::fail(MREFSR_E_INVALID, "%s: non-positive dimension in shape", who);
    if (g.C % g.groups || g.Co % g.groups || g.C % g.dg)
        return mrefsr::fail(MREFSR_E_INVALID, "%s: C=%d / Co=%d not divisible by groups=%d / dg=%d", who, g.C, g.Co,
                            g.groups, g.dg);
    g.Ho = (g.H + 2 * g.ph - (g.dh * (g.kh - 1) + 1)) / g.sh + 1;
    g.Wo = (g.W + 2 * g.pw - (g.dw * (g.kw - 1) + 1)) / g.sw + 1;
    if (g.Ho <= 0 || g.Wo <= 0) return mrefsr::fail(MREFSR_E_INVALID, "%s: empty output %dx%d", who, g.Ho, g.Wo);
    return 0;
}

// bilinear setup shared by every kernel: 4 clamped corner offsets + 4 weights (0 where the
// corner is out of range or the sample is outside the validity window)
struct Tap {
    int o1, o2, o3, o4;
    float w1, w2, w3, w4;
    float lh, lw;
    bool inside;
};

__device__ __forceinline__ Tap make_tap(float hi, float wi, int H, int W)
{
    Tap t;
    t.inside = (hi > -1.f) && (wi > -1.f) && (hi < (float)H) && (wi < (float)W);
    const float fh = floorf(hi), fw = floorf(wi);
    const int hl = (int)fh, wl = (int)fw, hh = hl + 1, wh = wl + 1;
    const float lh = hi - fh, lw = wi - fw, uh = 1.f - lh, uw = 1.f - lw;
    t.lh = lh;
    t.lw = lw;
    const bool okhl = t.inside && hl >= 0, okhh = t.inside && hh <= H - 1;
    const bool okwl = wl >= 0, okwh = wh <= W - 1;
    const int chl = min(max(hl, 0), H - 1), chh = min(max(hh, 0), H - 1);
    const int cwl = min(max(wl, 0), W - 1), cwh = min(max(wh, 0), W - 1);
    t.o1 = chl * W + cwl;
    t.o2 = chl * W + cwh;
    t.o3 = chh * W + cwl;
    t.o4 = chh * W + cwh;
    // Four SCALAR products.  Left to itself the SLP vectoriser pairs (uh*lw, lh*uw) into
    //     v_pk_mul_f32 vD, vA, vB op_sel:[0,1] op_sel_hi:[1,0]          (crossed halves)
    // and on gfx950 that instruction returned wrong values in lanes 48..63 whenever waves of OTHER workgroups
    // were issuing bf16 MFMAs on the same SIMD (run-to-run different DCN results at full size, 0.2-2 % of the
    // pixels; DESIGN 3.2 has the bisection: one workgroup per CU, no MFMA, -fno-slp-vectorize and this form
    // are clean, the hand-written instruction fails with or without wait states around it).  The empty asm
    // statements make the products opaque to the vectoriser; tests/test_boundary.py checks the built library
    // for the instruction form.
    float p1 = uh * uw, p2 = uh * lw, p3 = lh * uw, p4 = lh * lw;
    asm volatile("" : "+v"(p1));
    asm volatile("" : "+v"(p2));
    asm volatile("" : "+v"(p3));
    asm volatile("" : "+v"(p4));
    t.w1 = (okhl && okwl) ? p1 : 0.f;
    t.w2 = (okhl && okwh) ? p2 : 0.f;
    t.w3 = (okhh && okwl) ? p3 : 0.f;
    t.w4 = (okhh && okwh) ? p4 : 0.f;
    return t;
}

// ---------------------------------------------------------------------------------------------
// weight packing for the MFMA forward:  Wp[chunk = tap*(C/32) + cb][o][kh2][16]
//   = W[o][32*cb + 2*t + kh2][tap],  t = 0..15   (k-step t of the chunk uses rows 2t, 2t+1)
// ---------------------------------------------------------------------------------------------
__global__ void dcn_pack_weight_kernel(const float *__restrict__ w, float *__restrict__ wp, int Co, int C)
{
    const long total = (long)Co * C * 9;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int pos = (int)(e & 31);
        long t = e >> 5;
        const int o = (int)(t % Co);
        const int chunk = (int)(t / Co);
        const int ncb = C >> 5, tap = chunk / ncb, cb = chunk - tap * ncb;
        const int kh2 = pos >> 4, tt = pos & 15;
        wp[e] = w[((size_t)o * C + 32 * cb + 2 * tt + kh2) * 9 + tap];
    }
}

// ---------------------------------------------------------------------------------------------
// fused forward.  Block = 256 threads = 4 waves, tile = 64 consecutive output pixels x all Co.
// ---------------------------------------------------------------------------------------------
constexpr int CL_LD = 36;  // floats per pixel in a staged column chunk: [kh2][16] + 4 pad
constexpr int CL_BUF = 64 * CL_LD;

// XL = 0: x is NCHW (8 channels x 4 corners = 32 scalar gathers per thread and chunk);
// XL = 1: x is NHWC (the thread's 8 channels are contiguous: 4 corners x 2 x 16-byte loads).
template <int MB, int NB, int XL>
__global__ __launch_bounds__(256) void dcn_fwd_mfma_kernel(const float *__restrict__ x, const float *__restrict__ offset,
                                                           const float *__restrict__ mask, const float *__restrict__ wp,
                                                           const float *__restrict__ bias, float *__restrict__ out, Geo g,
                                                           float slope, int out_nhwc, int xcd_order)
{
    __shared__ __attribute__((aligned(16))) float cols[2 * CL_BUF];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int HWo = g.Ho * g.Wo, HWi = g.H * g.W;
    // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so
    // workgroup id -> (id % 8) * per_xcd + id / 8 gives every XCD one contiguous run of pixel tiles:
    // the bilinear corners of neighbouring tiles then hit the same L2 instead of being fetched 8 times.
    const int tiles = (HWo + 63) >> 6;
    int b, p0;
    {
        const long total = (long)tiles * g.B;
        long id = blockIdx.x;
        if (xcd_order) {
            const long per = gridDim.x >> 3;
            id = (id & 7) * per + (id >> 3);
        }
        if (id >= total) return;
        b = (int)(id / tiles);
        p0 = (int)(id - (long)b * tiles) * 64;
    }
    const int cpg = g.C / g.dg;
    const int ncb = g.C >> 5, nchunk = 9 * ncb;

    // gather role.  XL = 0 (NCHW x): one pixel (tid & 63) x 8 channels (8 * (tid >> 6)) per thread.
    // XL = 1 (NHWC x): 8 consecutive lanes read the 8 x 16 bytes = one 128-byte line of a corner's 32
    // channels, each thread 4 channels (4 * (tid & 7)) of 2 pixels (tid >> 3, + 32): every fetched
    // cache line is used in full (one pixel x 8 channels per thread used a quarter of it).
    constexpr int NPT = XL ? 2 : 1, NCT = XL ? 4 : 8;
    const int gch = XL ? 4 * (tid & 7) : 8 * (tid >> 6);   // first K row (channel within the chunk) of this thread
    int gpx[NPT], ho[NPT], wo[NPT];
    bool pvalid[NPT];
    const float *offb[NPT], *mskb[NPT];
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        gpx[j] = XL ? (tid >> 3) + 32 * j : (tid & 63);
        const int pix = p0 + gpx[j];
        pvalid[j] = pix < HWo;
        ho[j] = pvalid[j] ? pix / g.Wo : 0;
        wo[j] = pvalid[j] ? pix - ho[j] * g.Wo : 0;
        offb[j] = offset + (size_t)b * g.dg * 18 * HWo + (pvalid[j] ? pix : 0);
        mskb[j] = mask ? mask + (size_t)b * g.dg * 9 * HWo + (pvalid[j] ? pix : 0) : nullptr;
    }
    const float *xb = x + (size_t)b * g.C * HWi;

    // MFMA role
    const int mb0 = (MB == 2) ? 2 * wv : (NB == 2 ? wv : (wv & 1));
    const int nb0 = (NB == 2) ? 0 : (wv >> 1);
    f32x16 acc[MB][NB];
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int ni = 0; ni < NB; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

    float cv[NPT][NCT][4];  // raw corner values of the chunk in flight
    Tap tp[NPT];
    float mval[NPT];

    // three-stage software pipeline: offsets / mask of chunk+2 (tiny loads, one memory latency), bilinear
    // corners of chunk+1 (addresses depend on those offsets: a second latency), MFMAs of chunk
    float oh_n[NPT], ow_n[NPT], mv_n[NPT];
    auto offs_issue = [&](int chunk) {
        const int tap = chunk / ncb, cb = chunk - tap * ncb;
        const int grp = (32 * cb + gch) / cpg;
#pragma unroll
        for (int j = 0; j < NPT; ++j) {
            oh_n[j] = pvalid[j] ? offb[j][(size_t)(grp * 18 + 2 * tap) * HWo] : 0.f;
            ow_n[j] = pvalid[j] ? offb[j][(size_t)(grp * 18 + 2 * tap + 1) * HWo] : 0.f;
            mv_n[j] = pvalid[j] ? (mskb[j] ? mskb[j][(size_t)(grp * 9 + tap) * HWo] : 1.f) : 0.f;
        }
    };
    auto gather_issue = [&](int chunk) {
        const int tap = chunk / ncb, cb = chunk - tap * ncb;
        const int c0 = 32 * cb + gch;
        const int ti = tap / 3, tj = tap - ti * 3;
#pragma unroll
        for (int j = 0; j < NPT; ++j) {
            mval[j] = mv_n[j];
            const float hi = (float)(ho[j] * g.sh - g.ph + ti * g.dh) + oh_n[j];
            const float wi = (float)(wo[j] * g.sw - g.pw + tj * g.dw) + ow_n[j];
            tp[j] = make_tap(hi, wi, g.H, g.W);
            if (XL == 0) {
                const float *xc = xb + (size_t)c0 * HWi;
#pragma unroll
                for (int i = 0; i < NCT; ++i) {
                    const float *im = xc + (size_t)i * HWi;
                    cv[j][i][0] = im[tp[j].o1];
                    cv[j][i][1] = im[tp[j].o2];
                    cv[j][i][2] = im[tp[j].o3];
                    cv[j][i][3] = im[tp[j].o4];
                }
            } else {
                const float *xc = xb + c0;
                const int offs[4] = {tp[j].o1, tp[j].o2, tp[j].o3, tp[j].o4};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f32x4 v4 = *reinterpret_cast<const f32x4 *>(xc + (size_t)offs[k] * g.C);
#pragma unroll
                    for (int i = 0; i < 4; ++i) cv[j][i][k] = v4[i];
                }
            }
        }
    };
    auto gather_commit = [&](float *buf) {
#pragma unroll
        for (int j = 0; j < NPT; ++j) {
            float v[NCT];
#pragma unroll
            for (int i = 0; i < NCT; ++i)
                v[i] = (tp[j].w1 * cv[j][i][0] + tp[j].w2 * cv[j][i][1] + tp[j].w3 * cv[j][i][2] + tp[j].w4 * cv[j][i][3]) * mval[j];
            // K row r of the chunk lives at [r & 1][r >> 1] of the pixel's 36-float record
            float *dst = buf + gpx[j] * CL_LD + (gch >> 1);
            if (XL == 0) {
                *reinterpret_cast<f32x4 *>(dst) = f32x4{v[0], v[2], v[4], v[6]};       // even K rows -> kh2 = 0
                *reinterpret_cast<f32x4 *>(dst + 16) = f32x4{v[1], v[3], v[5], v[7]};  // odd K rows  -> kh2 = 1
            } else {
                *reinterpret_cast<float2 *>(dst) = make_float2(v[0], v[2]);
                *reinterpret_cast<float2 *>(dst + 16) = make_float2(v[1], v[3]);
            }
        }
    };

    offs_issue(0);
    gather_issue(0);
    if (nchunk > 1) offs_issue(1);
    gather_commit(cols);
    __syncthreads();

    for (int chunk = 0; chunk < nchunk; ++chunk) {
        const int buf = chunk & 1;
        const bool has_next = chunk + 1 < nchunk;
        if (has_next) gather_issue(chunk + 1);
        if (chunk + 2 < nchunk) offs_issue(chunk + 2);

        // A: packed weights of this chunk
        f32x4 a[MB][4];
#pragma unroll
        for (int mi = 0; mi < MB; ++mi) {
            const float *wsrc = wp + ((size_t)chunk * g.Co + (mb0 + mi) * 32 + (lane & 31)) * 32 + (lane >> 5) * 16;
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) a[mi][t4] = *reinterpret_cast<const f32x4 *>(wsrc + 4 * t4);
        }
        const float *bb = cols + buf * CL_BUF + (lane & 31) * CL_LD + (lane >> 5) * 16;
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            f32x4 bv[NB];
#pragma unroll
            for (int ni = 0; ni < NB; ++ni) bv[ni] = *reinterpret_cast<const f32x4 *>(bb + (nb0 + ni) * 32 * CL_LD + 4 * t4);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mi = 0; mi < MB; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NB; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][t4][j], bv[ni][j], acc[mi][ni], 0, 0, 0);
        }
        if (has_next) gather_commit(cols + (buf ^ 1) * CL_BUF);
        __syncthreads();
    }

    // epilogue: + bias, LeakyReLU(slope); NHWC: a lane's 4 consecutive couts of its pixel as one 16-byte store
    if (out_nhwc) {
#pragma unroll
        for (int mi = 0; mi < MB; ++mi)
#pragma unroll
            for (int ni = 0; ni < NB; ++ni) {
                const int px = p0 + (nb0 + ni) * 32 + (lane & 31);
                if (px < HWo) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int o = (mb0 + mi) * 32 + 8 * q + 4 * (lane >> 5);
                        float4 v = make_float4(acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2], acc[mi][ni][4 * q + 3]);
                        if (bias) v.x += bias[o], v.y += bias[o + 1], v.z += bias[o + 2], v.w += bias[o + 3];
                        v.x = v.x > 0.f ? v.x : v.x * slope, v.y = v.y > 0.f ? v.y : v.y * slope;
                        v.z = v.z > 0.f ? v.z : v.z * slope, v.w = v.w > 0.f ? v.w : v.w * slope;
                        *reinterpret_cast<float4 *>(out + ((size_t)b * HWo + px) * g.Co + o) = v;
                    }
                }
            }
        return;
    }
    // NCHW: coalesced stores along pixels
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int ni = 0; ni < NB; ++ni) {
            const int px = p0 + (nb0 + ni) * 32 + (lane & 31);
            if (px < HWo) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int o = (mb0 + mi) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    float v = acc[mi][ni][e] + (bias ? bias[o] : 0.f);
                    v = v > 0.f ? v : v * slope;
                    out[((size_t)b * g.Co + o) * HWo + px] = v;
                }
            }
        }
}


// ---------------------------------------------------------------------------------------------
// Channels-last forward on the bf16 matrix pipe (the inference path).  Same tile, gather mapping
// and three-stage pipeline as dcn_fwd_mfma_kernel<.., XL = 1>, but the sampled column values and
// the weights are split exactly into three bf16 terms and every product is the six partial
// products >= 2^-24 (csrc/conv_nhwc.hip explains the arithmetic): fp32-equivalent results at
// 2.7x the fp32 matrix rate.  Weights: Wq[chunk = tap*(C/32)+cb][kstep 2][split 3][Co][16] bf16.
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned int pk_bf16(float a, float b)
{
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2{a, b}, bf16x2));
}

__global__ void dcn_pack_weight_bf16_kernel(const float *__restrict__ w, unsigned short *__restrict__ wq, int Co, int C)
{
    const long total = (long)Co * C * 9;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int k = (int)(e & 15);
        long t = e >> 4;
        const int o = (int)(t % Co);
        t /= Co;
        const int ks = (int)(t & 1);
        const int chunk = (int)(t >> 1);
        const int ncb = C >> 5, tap = chunk / ncb, cb = chunk - tap * ncb;
        float v = w[((size_t)o * C + 32 * cb + 16 * ks + k) * 9 + tap];
        const size_t base = ((((size_t)chunk * 2 + ks) * 3) * Co + o) * 16 + k;
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) {
            const unsigned int p = pk_bf16(v, 0.f);
            wq[base + (size_t)sp * Co * 16] = (unsigned short)(p & 0xffffu);
            v -= __uint_as_float(p << 16);
        }
    }
}

// fp16 two-term split (csrc/conv_nhwc.hip MODE 2): w * S = wh + wl with S = 2^s chosen so that max|w| * S is in
// [2^13, 2^14); planes wh, wl, WH2 = wh * 2^-11.  The scale is found on the device: scal[0] = max|w| (as uint bits),
// scal[1] = S, scal[2] = 1 / S; no host synchronisation.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int pk_f16(float a, float b)
{
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2{a, b}, f16x2));
}
__device__ __forceinline__ f32x2 un_f16(unsigned int p) { return __builtin_convertvector(__builtin_bit_cast(f16x2, p), f32x2); }

__global__ void dcn_weight_amax_kernel(const float *__restrict__ w, unsigned int *__restrict__ scal, long total)
{
    float m = 0.f;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(w[e]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(scal, __float_as_uint(m));  // non-negative floats order like their bits
}

__global__ void dcn_pack_weight_f16_kernel(const float *__restrict__ w, unsigned short *__restrict__ wq, float *__restrict__ scal, int Co, int C)
{
    const float amax = __uint_as_float(reinterpret_cast<const unsigned int *>(scal)[0]);
    int ex = 0;
    if (amax > 0.f && amax < 3.0e38f) (void)frexpf(amax, &ex);     // amax = m * 2^ex, m in [0.5, 1)
    const float S = amax > 0.f ? ldexpf(1.f, 14 - ex) : 1.f;        // amax * S in [2^13, 2^14)
    if (blockIdx.x == 0 && threadIdx.x == 0) scal[1] = S, scal[2] = 1.f / S;
    const long total = (long)Co * C * 9;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int k = (int)(e & 15);
        long t = e >> 4;
        const int o = (int)(t % Co);
        t /= Co;
        const int ks = (int)(t & 1);
        const int chunk = (int)(t >> 1);
        const int ncb = C >> 5, tap = chunk / ncb, cb = chunk - tap * ncb;
        const float v = w[((size_t)o * C + 32 * cb + 16 * ks + k) * 9 + tap] * S;
        const size_t base = ((((size_t)chunk * 2 + ks) * 3) * Co + o) * 16 + k;
        const unsigned int ph = pk_f16(v, 0.f);
        const float h = un_f16(ph)[0];
        wq[base] = (unsigned short)(ph & 0xffffu);
        wq[base + (size_t)Co * 16] = (unsigned short)(pk_f16(v - h, 0.f) & 0xffffu);
        wq[base + (size_t)2 * Co * 16] = (unsigned short)(pk_f16(h * (1.0f / 2048.f), 0.f) & 0xffffu);
    }
}

// the bilinear blend as ONE defined operation order (both 16-bit kernels: the same bits whatever the compiler would contract)
__device__ __forceinline__ float blend4(const Tap &t, const float (&c)[4])
{
    return __builtin_fmaf(t.w4, c[3], __builtin_fmaf(t.w3, c[2], __builtin_fmaf(t.w2, c[1], t.w1 * c[0])));
}
// epilogue arithmetic: acc * (1 / S) + bias as one fma (bias 0 when absent), then LeakyReLU
__device__ __forceinline__ float4 epi4(float4 v, float oscale, const float *bias, int o, float slope)
{
    const float4 bz = bias ? *reinterpret_cast<const float4 *>(bias + o) : make_float4(0.f, 0.f, 0.f, 0.f);
    v.x = __builtin_fmaf(v.x, oscale, bz.x), v.y = __builtin_fmaf(v.y, oscale, bz.y);
    v.z = __builtin_fmaf(v.z, oscale, bz.z), v.w = __builtin_fmaf(v.w, oscale, bz.w);
    v.x = v.x > 0.f ? v.x : v.x * slope, v.y = v.y > 0.f ? v.y : v.y * slope;
    v.z = v.z > 0.f ? v.z : v.z * slope, v.w = v.w > 0.f ? v.w : v.w * slope;
    return v;
}

constexpr int CQ_LD = 80;                  // bytes per pixel per split plane: 32 bf16 + 16 pad (conflict-free b128 reads)
constexpr int CQ_PLANE = 64 * CQ_LD;       // one split plane of a 64-pixel chunk
constexpr int CQ_BUF = 3 * CQ_PLANE;

// NT = 16 (default): fp16 two-term split, three products (fp32-equivalent; columns must stay inside the fp16 range:
// range_flag).  NT = 6: bf16 three-term split, six products (fp32-equivalent, no range limit).  NT = 1: bf16 ARITHMETIC
// (BASELINE configs[4]): columns and weights rounded to one bf16 plane, fp32 accumulation, the output rounded to bf16.
// IO16 (with NT = 1): x and out are bf16 tensors (2-byte channels-last storage): a corner's 8 channels are one 16-byte load.
template <int MB, int NB, int NT, bool MAP8, bool IO16 = false>
__global__ __launch_bounds__(256) void dcn_fwd_bf16_kernel(const float *__restrict__ x, const float *__restrict__ offset,
                                                           const float *__restrict__ mask, const unsigned short *__restrict__ wq,
                                                           const float *__restrict__ bias, float *__restrict__ out, Geo g,
                                                           float slope, int out_nhwc, int xcd_order, const float *__restrict__ scal,
                                                           int *__restrict__ range_flag, unsigned int *__restrict__ out_amax)
{
    __shared__ __attribute__((aligned(16))) unsigned char cols[2 * CQ_BUF];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int HWo = g.Ho * g.Wo;
    const int tiles = (HWo + 63) >> 6;
    int b, p0;
    {
        const long total = (long)tiles * g.B;
        long id = blockIdx.x;
        if (xcd_order) {
            const long per = gridDim.x >> 3;
            id = (id & 7) * per + (id >> 3);
        }
        if (id >= total) return;
        b = (int)(id / tiles);
        p0 = (int)(id - (long)b * tiles) * 64;
    }
    const int cpg = g.C / g.dg;
    const int ncb = g.C >> 5, nchunk = 9 * ncb;

    // gather role.  MAP8: the 8 channels 8 * (tid & 3) .. +7 of pixel tid >> 2 -- at most one deformable group (C / dg is a
    // multiple of 8), so the bilinear setup is computed once per thread and chunk (4-7 % faster at C = 128 / 256).
    // Otherwise 4 channels (4 * (tid & 7)) of pixels (tid >> 3) and (tid >> 3) + 32: every load instruction covers whole
    // 128-byte lines (2 % faster at C = 64, where the gather itself is the bound).
    constexpr int NJ = MAP8 ? 1 : 2, NCH = MAP8 ? 8 : 4;
    const int gch = MAP8 ? 8 * (tid & 3) : 4 * (tid & 7);
    int gpx[NJ], ho[NJ], wo[NJ];
    bool pvalid[NJ];
    const float *offb[NJ], *mskb[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        gpx[j] = NJ == 1 ? (tid >> 2) : (tid >> 3) + 32 * j;
        const int pix = p0 + gpx[j];
        pvalid[j] = pix < HWo;
        ho[j] = pvalid[j] ? pix / g.Wo : 0;
        wo[j] = pvalid[j] ? pix - ho[j] * g.Wo : 0;
        offb[j] = offset + (size_t)b * g.dg * 18 * HWo + (pvalid[j] ? pix : 0);
        mskb[j] = mask ? mask + (size_t)b * g.dg * 9 * HWo + (pvalid[j] ? pix : 0) : nullptr;
    }
    const float *xb = IO16 ? reinterpret_cast<const float *>(reinterpret_cast<const unsigned short *>(x) + (size_t)b * g.C * g.H * g.W)
                           : x + (size_t)b * g.C * g.H * g.W;

    // NB == 1: wave = one 32-pixel tile (wv & 1) x MB cout tiles; NB == 2 (both pixel tiles per wave): MB cout tiles
    const int mb0 = (NB == 1) ? (wv >> 1) * MB : wv * MB;
    const int nb0 = (NB == 1) ? (wv & 1) : 0;
    f32x16 acc[MB][NB];
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int ni = 0; ni < NB; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

    float cv[NJ][NCH][4];
    Tap tp[NJ];
    float mval[NJ], oh_n[NJ], ow_n[NJ], mv_n[NJ];
    auto offs_issue = [&](int chunk) {
        const int tap = chunk / ncb, cb = chunk - tap * ncb;
        const int grp = (32 * cb + gch) / cpg;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            oh_n[j] = pvalid[j] ? offb[j][(size_t)(grp * 18 + 2 * tap) * HWo] : 0.f;
            ow_n[j] = pvalid[j] ? offb[j][(size_t)(grp * 18 + 2 * tap + 1) * HWo] : 0.f;
            mv_n[j] = pvalid[j] ? (mskb[j] ? mskb[j][(size_t)(grp * 9 + tap) * HWo] : 1.f) : 0.f;
        }
    };
    auto gather_issue = [&](int chunk) {
        const int tap = chunk / ncb, cb = chunk - tap * ncb;
        const float *xc = xb + 32 * cb + gch;
        const unsigned short *xh = reinterpret_cast<const unsigned short *>(xb) + 32 * cb + gch;
        const int ti = tap / 3, tj = tap - ti * 3;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            mval[j] = mv_n[j];
            tp[j] = make_tap((float)(ho[j] * g.sh - g.ph + ti * g.dh) + oh_n[j], (float)(wo[j] * g.sw - g.pw + tj * g.dw) + ow_n[j],
                             g.H, g.W);
            const int offs[4] = {tp[j].o1, tp[j].o2, tp[j].o3, tp[j].o4};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (IO16) {   // NCH bf16 channels: one 16-byte (NCH = 8) or 8-byte (NCH = 4) load
                    unsigned int r[NCH / 2];
                    if (NCH == 8) {
                        const u32x4 t = *reinterpret_cast<const u32x4 *>(xh + (size_t)offs[k] * g.C);
                        r[0] = t[0], r[1] = t[1], r[NCH / 2 - 2] = t[2], r[NCH / 2 - 1] = t[3];
                    } else {
                        const u32x2 t = *reinterpret_cast<const u32x2 *>(xh + (size_t)offs[k] * g.C);
                        r[0] = t[0], r[1] = t[1];
                    }
#pragma unroll
                    for (int i = 0; i < NCH / 2; ++i) {
                        cv[j][2 * i][k] = __uint_as_float(r[i] << 16);
                        cv[j][2 * i + 1][k] = __uint_as_float(r[i] & 0xffff0000u);
                    }
                    continue;
                }
#pragma unroll
                for (int q = 0; q < NCH / 4; ++q) {
                    const f32x4 v4 = *reinterpret_cast<const f32x4 *>(xc + (size_t)offs[k] * g.C + 4 * q);
#pragma unroll
                    for (int i = 0; i < 4; ++i) cv[j][4 * q + i][k] = v4[i];
                }
            }
        }
    };
    auto gather_commit = [&](unsigned char *buf) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float v[NCH];
#pragma unroll
            for (int i = 0; i < NCH; ++i)
                v[i] = blend4(tp[j], cv[j][i]) * mval[j];
            unsigned char *dst = buf + gpx[j] * CQ_LD + gch * 2;
            if (NT == 16) {
                float amx = 0.f;
#pragma unroll
                for (int i = 0; i < NCH; ++i) amx = fmaxf(amx, fabsf(v[i]));
                if (range_flag && !(amx <= 65000.f)) atomicOr(range_flag, 1);
                unsigned int qh[NCH / 2], ql[NCH / 2];
#pragma unroll
                for (int i = 0; i < NCH / 2; ++i) {
                    qh[i] = pk_f16(v[2 * i], v[2 * i + 1]);
                    // (v - h) * 2048 as fma(h, -2048, v * 2048): the fp16 half is read in place (v_fma_mix_f32); both
                    // forms are exact (v - h is representable, powers of two scale exactly)
                    const f16x2 h = __builtin_bit_cast(f16x2, qh[i]);
                    ql[i] = pk_f16(__builtin_fmaf((float)h[0], -2048.f, v[2 * i] * 2048.f), __builtin_fmaf((float)h[1], -2048.f, v[2 * i + 1] * 2048.f));
                }
                if (NCH == 8) {
                    *reinterpret_cast<u32x4 *>(dst) = u32x4{qh[0], qh[1], qh[NCH / 2 - 2], qh[NCH / 2 - 1]};
                    *reinterpret_cast<u32x4 *>(dst + CQ_PLANE) = u32x4{ql[0], ql[1], ql[NCH / 2 - 2], ql[NCH / 2 - 1]};
                } else {
                    *reinterpret_cast<u32x2 *>(dst) = u32x2{qh[0], qh[1]};
                    *reinterpret_cast<u32x2 *>(dst + CQ_PLANE) = u32x2{ql[0], ql[1]};
                }
                continue;
            }
#pragma unroll
            for (int sp = 0; sp < (NT == 1 ? 1 : 3); ++sp) {
                unsigned int q[NCH / 2];
#pragma unroll
                for (int i = 0; i < NCH / 2; ++i) q[i] = pk_bf16(v[2 * i], v[2 * i + 1]);
                if (NCH == 8)
                    *reinterpret_cast<u32x4 *>(dst + sp * CQ_PLANE) = u32x4{q[0], q[1], q[NCH / 2 - 2], q[NCH / 2 - 1]};
                else
                    *reinterpret_cast<u32x2 *>(dst + sp * CQ_PLANE) = u32x2{q[0], q[1]};
                if (sp < 2) {
#pragma unroll
                    for (int i = 0; i < NCH / 2; ++i) {
                        v[2 * i] -= __uint_as_float(q[i] << 16);
                        v[2 * i + 1] -= __uint_as_float(q[i] & 0xffff0000u);
                    }
                }
            }
        }
    };

    offs_issue(0);
    gather_issue(0);
    if (nchunk > 1) offs_issue(1);
    gather_commit(cols);
    __syncthreads();

    constexpr int TA[6] = {1, 2, 0, 1, 0, 0}, TB[6] = {1, 0, 2, 0, 1, 0};  // (weight split, column split), smallest first
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        const int buf = chunk & 1;
        const bool has_next = chunk + 1 < nchunk;
        if (has_next) gather_issue(chunk + 1);
        if (chunk + 2 < nchunk) offs_issue(chunk + 2);
        // All operand fragments of the chunk (both 16-channel k-steps, every split plane) are fetched into their
        // own registers before the first MFMA, so the 12 * MB * NB MFMAs issue back to back.
        constexpr int NP = NT == 1 ? 1 : 3, NPB = NT == 1 ? 1 : (NT == 16 ? 2 : 3);  // weight / column planes
        u32x4 a[2][MB][NP], bv[2][NB][NPB];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int mi = 0; mi < MB; ++mi)
#pragma unroll
                for (int sp = 0; sp < (NT == 16 ? 2 : NP); ++sp)
                    a[ks][mi][sp] = *reinterpret_cast<const u32x4 *>(
                        wq + ((((size_t)chunk * 2 + ks) * 3 + sp) * g.Co + (mb0 + mi) * 32 + (lane & 31)) * 16 + (lane >> 5) * 8);
#pragma unroll
            for (int ni = 0; ni < NB; ++ni)
#pragma unroll
                for (int sp = 0; sp < NPB; ++sp)
                    bv[ks][ni][sp] = *reinterpret_cast<const u32x4 *>(cols + buf * CQ_BUF + sp * CQ_PLANE +
                                                                      ((nb0 + ni) * 32 + (lane & 31)) * CQ_LD + ks * 32 + (lane >> 5) * 16);
        }
        if (NT == 16) {  // WH2 = wh * 2^-11 (exact while normal, RNE into the denormals like the pack kernel): a third fewer weight loads
            const f16x2 k11 = {(_Float16)0.00048828125f, (_Float16)0.00048828125f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int mi = 0; mi < MB; ++mi)
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const unsigned int d = a[ks][mi][0][w];
                        a[ks][mi][2][w] = __builtin_bit_cast(unsigned int, __builtin_bit_cast(f16x2, d) * k11);
                    }
        }
        __builtin_amdgcn_sched_barrier(0);   // keep every load above, every MFMA below
        if (NT == 16) {
            constexpr int WA[3] = {1, 2, 0}, CB[3] = {0, 1, 0};  // wl*ah, WH2*AL, wh*ah: smallest first
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NB; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[ks][mi][WA[t]]),
                                                                                 __builtin_bit_cast(f16x8, bv[ks][ni][NT == 16 ? CB[t] : 0]),
                                                                                 acc[mi][ni], 0, 0, 0);
        } else {
#pragma unroll
        for (int t = (NT == 1 ? 5 : 0); t < 6; ++t)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int mi = 0; mi < MB; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NB; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[ks][mi][NT == 6 ? TA[t] : 0]),
                                                                              __builtin_bit_cast(bf16x8, bv[ks][ni][NT == 6 ? TB[t] : 0]),
                                                                              acc[mi][ni], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (has_next) gather_commit(cols + (buf ^ 1) * CQ_BUF);
        __syncthreads();
    }

    const float oscale = NT == 16 ? scal[2] : 1.f;  // 1 / S of the weight scaling
    float oamx = 0.f;   // max |out| of the launch -> out_amax
    auto publish = [&]() {
        if (!out_amax) return;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) oamx = fmaxf(oamx, __shfl_xor(oamx, o, 64));
        if (lane == 0 && oamx > 0.f && oamx < 3.0e38f && __float_as_uint(oamx) > __builtin_nontemporal_load(out_amax)) atomicMax(out_amax, __float_as_uint(oamx));
    };
    if (out_nhwc) {
#pragma unroll
        for (int mi = 0; mi < MB; ++mi)
#pragma unroll
            for (int ni = 0; ni < NB; ++ni) {
                const int px = p0 + (nb0 + ni) * 32 + (lane & 31);
                if (px < HWo) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int o = (mb0 + mi) * 32 + 8 * q + 4 * (lane >> 5);
                        float4 v = make_float4(acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2], acc[mi][ni][4 * q + 3]);
                        v = epi4(v, oscale, bias, o, slope);
                        oamx = fmaxf(fmaxf(fmaxf(oamx, fabsf(v.x)), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
                        if (NT == 1) {
                            const unsigned int r0 = pk_bf16(v.x, v.y), r1 = pk_bf16(v.z, v.w);
                            if (IO16) {
                                *reinterpret_cast<u32x2 *>(reinterpret_cast<unsigned short *>(out) + ((size_t)b * HWo + px) * g.Co + o) = u32x2{r0, r1};
                                continue;
                            }
                            v = make_float4(__uint_as_float(r0 << 16), __uint_as_float(r0 & 0xffff0000u), __uint_as_float(r1 << 16),
                                            __uint_as_float(r1 & 0xffff0000u));
                        }
                        *reinterpret_cast<float4 *>(out + ((size_t)b * HWo + px) * g.Co + o) = v;
                    }
                }
            }
        publish();
        return;
    }
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int ni = 0; ni < NB; ++ni) {
            const int px = p0 + (nb0 + ni) * 32 + (lane & 31);
            if (px < HWo) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int o = (mb0 + mi) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    float v = __builtin_fmaf(acc[mi][ni][e], oscale, bias ? bias[o] : 0.f);
                    v = v > 0.f ? v : v * slope;
                    if (NT == 1) v = __uint_as_float(pk_bf16(v, 0.f) << 16);
                    oamx = fmaxf(oamx, fabsf(v));
                    out[((size_t)b * g.Co + o) * HWo + px] = v;
                }
            }
        }
    publish();
}

// ---------------------------------------------------------------------------------------------
// Channels-last forward, T pixel tiles per block with the K chunk as the OUTER loop (round 6: the default wherever a tap's
// 3 dg offset / mask planes fit LDS -- every DynAgg of the path).  What dcn_fwd_bf16_kernel spends its time on at C = 64 is the
// vector-memory path, and only 44 % of what it sends down that path is the gather (profiles/r3_dcn_l1_pmc.json: 72 wave loads
// per tile and chunk = 32 gather + 24 four-byte offset / mask loads, issued again for every channel chunk + 16 weight
// fragments, 147 KB re-read per 64-pixel tile).  Here
//   * a tap's offset / mask planes are staged ONCE per (tap, tile) into LDS by LDS-DMA (one 256-byte row per plane and tile,
//     6 instructions per wave), one tap ahead, and the threads read them with ds_read_b32: 6 instead of 24 * C/32 VMEM
//     instructions per tile and tap;
//   * a chunk's weight fragments are loaded once per chunk and serve the T tiles of the block from registers (the next
//     chunk's arrive during the chunk's last tile): 16 / T instead of 16 VMEM instructions per tile and chunk;
//   * gather mapping, split arithmetic, MFMA order and epilogue are dcn_fwd_bf16_kernel's: the same bits.
// Work item w = chunk * T + tile: gather of w + 1 in flight, offsets of w + 2 read, MFMAs of w, one barrier.
// LDS-DMA completion: the planes of tile t of tap + 1 are requested at the top of work item (first chunk of tap, t); the same
// item's gather_commit waits for loads issued AFTER them (vmcnt counts in order), the item's barrier publishes them; their
// first reader runs in a later item.  Buffer (tap + 1) & 1 was last read two items before tap began.
// ---------------------------------------------------------------------------------------------
template <int T>
struct PtCfg {
    static constexpr int PS = 64 * T + 4;   // floats per staged plane: + 4 puts the four deformable groups of a wave load on different banks
};
constexpr int pt_cols_bytes(int nt) { return (nt == 16 ? 2 : nt == 1 ? 1 : 3) * CQ_PLANE; }   // one column buffer: the split planes in use

// lane pair helpers of dcn_fwd_pt_kernel: the value held by the even / odd lane of this lane's pair (DPP quad_perm [0,0,2,2] / [1,1,3,3])
__device__ __forceinline__ int pair_lo(int v) { return __builtin_amdgcn_mov_dpp(v, 0xA0, 0xf, 0xf, true); }
__device__ __forceinline__ int pair_hi(int v) { return __builtin_amdgcn_mov_dpp(v, 0xF5, 0xf, 0xf, true); }
__device__ __forceinline__ float pair_lo(float v) { return __builtin_bit_cast(float, pair_lo(__builtin_bit_cast(int, v))); }
__device__ __forceinline__ float pair_hi(float v) { return __builtin_bit_cast(float, pair_hi(__builtin_bit_cast(int, v))); }
__device__ __forceinline__ const void *mrefsr_dcn_scalar_ptr(const void *p)
{
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)v), hi = __builtin_amdgcn_readfirstlane((unsigned int)(v >> 32));
    return reinterpret_cast<const void *>(((unsigned long long)hi << 32) | lo);
}

template <int MB, int NB, int NT, bool MAP8, bool IO16, int T>
__global__ __launch_bounds__(256, T == 2 && MB * NB == 1 ? 3 : 2) void dcn_fwd_pt_kernel(const float *__restrict__ x, const float *__restrict__ offset,
                                                         const float *__restrict__ mask, const unsigned short *__restrict__ wq,
                                                         const float *__restrict__ bias, float *__restrict__ out, Geo g,
                                                         float slope, int out_nhwc, int xcd_order, const float *__restrict__ scal,
                                                         int *__restrict__ range_flag, int cpg_shift, int n_pl, int cshift, unsigned int *__restrict__ out_amax)
{
    static_assert(T == 2 || T == 4, "dcn_fwd_pt_kernel: T");
    constexpr int PS = PtCfg<T>::PS;
    extern __shared__ __attribute__((aligned(16))) unsigned char pt_smem[];
    constexpr int CBUF = pt_cols_bytes(NT);
    unsigned char *cols = pt_smem;                                        // [2][CBUF]
    float *ofs = reinterpret_cast<float *>(pt_smem + 2 * CBUF);           // [2][n_pl][PS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HWo = g.Ho * g.Wo;
    const int tiles = (HWo + 63) >> 6, tgroups = (tiles + T - 1) / T;
    int b, p0;
    {
        const long total = (long)tgroups * g.B;
        long id = blockIdx.x;
        if (xcd_order) {
            const long per = gridDim.x >> 3;
            id = (id & 7) * per + (id >> 3);
        }
        if (id >= total) return;
        b = (int)(id / tgroups);
        p0 = (int)(id - (long)b * tgroups) * (64 * T);
    }
    const int ncb = g.C >> 5, nchunk = 9 * ncb;

    // PAIR (4-channel mapping, fp32 x): the two lanes of a pair hold the two 16-byte halves of a deformable group's 32 bytes for
    // pixels tid >> 3 and (tid >> 3) + 32.  The bilinear setup of (pixel, group, tap) is formed ONCE per pair -- the even lane
    // owns the first pixel's, the odd lane the second's -- and reaches the other lane as the DPP operand (quad_perm) of the address
    // add and of the blend's multiply-adds: the loads stay pair-contiguous (32 bytes per corner), the setup is not done twice.
    constexpr bool PAIR = !MAP8 && !IO16;
    constexpr int NJ = MAP8 ? 1 : 2, NCH = MAP8 ? 8 : 4, NS = PAIR ? 1 : NJ;   // NS: bilinear setups per thread and work item
    const int gch = MAP8 ? 8 * (tid & 3) : 4 * (tid & 7);
    int gpx[NJ], spx[NS];
    float hb[T][NS], wb[T][NS];   // sampling position of tap (0, 0) without the learned offset, per tile
#pragma unroll
    for (int j = 0; j < NJ; ++j) gpx[j] = NJ == 1 ? (tid >> 2) : (tid >> 3) + 32 * j;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        spx[j] = PAIR ? (tid >> 3) + 32 * (tid & 1) : gpx[j];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            int pix = p0 + 64 * t + spx[j];
            pix = pix < HWo ? pix : HWo - 1;   // (pixels past the end shadow the last one: staged planes and gather stay finite; never stored)
            const int ho = pix / g.Wo, wo = pix - ho * g.Wo;
            hb[t][j] = (float)(ho * g.sh - g.ph);
            wb[t][j] = (float)(wo * g.sw - g.pw);
        }
    }
    const float *xb = IO16 ? reinterpret_cast<const float *>(reinterpret_cast<const unsigned short *>(x) + (size_t)b * g.C * g.H * g.W)
                           : x + (size_t)b * g.C * g.H * g.W;
    const float *offb = offset + (size_t)b * g.dg * 18 * HWo;
    const float *mskb = mask ? mask + (size_t)b * g.dg * 9 * HWo : nullptr;
    const int n_op = 2 * g.dg;   // offset planes of a tap; the mask planes follow

    // ---- staging of a tap's planes for one tile: plane pl = 2 grp + (y | x) for pl < 2 dg, 2 dg + grp for the masks ----
    auto stage = [&](int tap, int tile) {
        int px = p0 + 64 * tile + lane;
        px = px < HWo ? px : HWo - 1;
        float *dst0 = ofs + ((tap & 1) * n_pl) * PS + 64 * tile;
        for (int pl = wv; pl < n_pl; pl += 4) {
            const float *src = pl < n_op ? offb + (size_t)((pl >> 1) * 18 + 2 * tap + (pl & 1)) * HWo
                                         : mskb + (size_t)((pl - n_op) * 9 + tap) * HWo;
            __builtin_amdgcn_global_load_lds(src + px, (__attribute__((address_space(3))) void *)(dst0 + pl * PS), 4, 0, 0);
        }
    };

    const int mb0 = (NB == 1) ? (wv >> 1) * MB : wv * MB;
    const int nb0 = (NB == 1) ? (wv & 1) : 0;
    f32x16 acc[T][MB][NB];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int mi = 0; mi < MB; ++mi)
#pragma unroll
            for (int ni = 0; ni < NB; ++ni)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[t][mi][ni][e] = 0.f;

    float cv[NJ][NCH][4];
    Tap tp[NS];
    float mval[NS], oh_n[NS], ow_n[NS], mv_n[NS], amx_run = 0.f;
    const __amdgpu_buffer_rsrc_t x_srd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void *>(mrefsr_dcn_scalar_ptr(xb)), 0, PAIR ? (unsigned int)((size_t)g.H * g.W * g.C * 4) : 0u, 0x00020000);
    // offsets / mask of work item (tap, cb, tile) from the staged planes
    auto offs_read = [&](int tap, int cb, int tile) {
        const int grp = (32 * cb + gch) >> cpg_shift;
        const float *o = ofs + ((tap & 1) * n_pl) * PS + 64 * tile;
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            oh_n[j] = o[(2 * grp) * PS + spx[j]];
            ow_n[j] = o[(2 * grp + 1) * PS + spx[j]];
            mv_n[j] = mskb ? o[(n_op + grp) * PS + spx[j]] : 1.f;
        }
    };
    auto gather_issue = [&](int tap, int cb, const float (&hbt)[NS], const float (&wbt)[NS]) {
        const float *xc = xb + 32 * cb + gch;
        const unsigned short *xh = reinterpret_cast<const unsigned short *>(xb) + 32 * cb + gch;
        const int ti = tap / 3, tj = tap - ti * 3;
        if constexpr (PAIR) {
            mval[0] = mv_n[0];
            tp[0] = make_tap(hbt[0] + (float)(ti * g.dh) + oh_n[0], wbt[0] + (float)(tj * g.dw) + ow_n[0], g.H, g.W);
            const unsigned int choff = (unsigned int)(32 * cb + gch) * 4u;
            const int offs[4] = {tp[0].o1, tp[0].o2, tp[0].o3, tp[0].o4};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ob = offs[k] << cshift;   // byte offset of the corner pixel inside the image (< 2^32: checked by the launcher)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const unsigned int vo = (unsigned int)(j == 0 ? pair_lo(ob) : pair_hi(ob)) + choff;
                    const f32x4 v4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_srd, vo, 0, 0));
#pragma unroll
                    for (int i = 0; i < 4; ++i) cv[j][i][k] = v4[i];
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            mval[j] = mv_n[j];
            tp[j] = make_tap(hbt[j] + (float)(ti * g.dh) + oh_n[j], wbt[j] + (float)(tj * g.dw) + ow_n[j], g.H, g.W);
            const int offs[4] = {tp[j].o1, tp[j].o2, tp[j].o3, tp[j].o4};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (IO16) {
                    unsigned int r[NCH / 2];
                    if (NCH == 8) {
                        const u32x4 t = *reinterpret_cast<const u32x4 *>(xh + (size_t)offs[k] * g.C);
                        r[0] = t[0], r[1] = t[1], r[NCH / 2 - 2] = t[2], r[NCH / 2 - 1] = t[3];
                    } else {
                        const u32x2 t = *reinterpret_cast<const u32x2 *>(xh + (size_t)offs[k] * g.C);
                        r[0] = t[0], r[1] = t[1];
                    }
#pragma unroll
                    for (int i = 0; i < NCH / 2; ++i) {
                        cv[j][2 * i][k] = __uint_as_float(r[i] << 16);
                        cv[j][2 * i + 1][k] = __uint_as_float(r[i] & 0xffff0000u);
                    }
                    continue;
                }
#pragma unroll
                for (int q = 0; q < NCH / 4; ++q) {
                    const f32x4 v4 = *reinterpret_cast<const f32x4 *>(xc + (size_t)offs[k] * g.C + 4 * q);
#pragma unroll
                    for (int i = 0; i < 4; ++i) cv[j][4 * q + i][k] = v4[i];
                }
            }
        }
    };
    auto gather_commit = [&](unsigned char *buf) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float v[NCH];
            if constexpr (PAIR) {   // blend4's operation order, the setup read from the pair's lane that formed it
#pragma unroll
                for (int i = 0; i < NCH; ++i) {
                    float a;
                    if (j == 0)
                        a = __builtin_fmaf(pair_lo(tp[0].w4), cv[j][i][3], __builtin_fmaf(pair_lo(tp[0].w3), cv[j][i][2],
                            __builtin_fmaf(pair_lo(tp[0].w2), cv[j][i][1], pair_lo(tp[0].w1) * cv[j][i][0]))) * pair_lo(mval[0]);
                    else
                        a = __builtin_fmaf(pair_hi(tp[0].w4), cv[j][i][3], __builtin_fmaf(pair_hi(tp[0].w3), cv[j][i][2],
                            __builtin_fmaf(pair_hi(tp[0].w2), cv[j][i][1], pair_hi(tp[0].w1) * cv[j][i][0]))) * pair_hi(mval[0]);
                    v[i] = a;
                }
            } else {
#pragma unroll
                for (int i = 0; i < NCH; ++i)
                    v[i] = blend4(tp[j], cv[j][i]) * mval[j];
            }
            unsigned char *dst = buf + gpx[j] * CQ_LD + gch * 2;
            if (NT == 16) {
#pragma unroll
                for (int i = 0; i < NCH; i += 2) amx_run = fmaxf(fmaxf(amx_run, fabsf(v[i])), fabsf(v[i + 1]));   // (checked once, after the last chunk)
                unsigned int qh[NCH / 2], ql[NCH / 2];
#pragma unroll
                for (int i = 0; i < NCH / 2; ++i) {
                    qh[i] = pk_f16(v[2 * i], v[2 * i + 1]);
                    const f16x2 h = __builtin_bit_cast(f16x2, qh[i]);
                    ql[i] = pk_f16(__builtin_fmaf((float)h[0], -2048.f, v[2 * i] * 2048.f), __builtin_fmaf((float)h[1], -2048.f, v[2 * i + 1] * 2048.f));
                }
                if (NCH == 8) {
                    *reinterpret_cast<u32x4 *>(dst) = u32x4{qh[0], qh[1], qh[NCH / 2 - 2], qh[NCH / 2 - 1]};
                    *reinterpret_cast<u32x4 *>(dst + CQ_PLANE) = u32x4{ql[0], ql[1], ql[NCH / 2 - 2], ql[NCH / 2 - 1]};
                } else {
                    *reinterpret_cast<u32x2 *>(dst) = u32x2{qh[0], qh[1]};
                    *reinterpret_cast<u32x2 *>(dst + CQ_PLANE) = u32x2{ql[0], ql[1]};
                }
                continue;
            }
#pragma unroll
            for (int sp = 0; sp < (NT == 1 ? 1 : 3); ++sp) {
                unsigned int q[NCH / 2];
#pragma unroll
                for (int i = 0; i < NCH / 2; ++i) q[i] = pk_bf16(v[2 * i], v[2 * i + 1]);
                if (NCH == 8)
                    *reinterpret_cast<u32x4 *>(dst + sp * CQ_PLANE) = u32x4{q[0], q[1], q[NCH / 2 - 2], q[NCH / 2 - 1]};
                else
                    *reinterpret_cast<u32x2 *>(dst + sp * CQ_PLANE) = u32x2{q[0], q[1]};
                if (sp < 2) {
#pragma unroll
                    for (int i = 0; i < NCH / 2; ++i) {
                        v[2 * i] -= __uint_as_float(q[i] << 16);
                        v[2 * i + 1] -= __uint_as_float(q[i] & 0xffff0000u);
                    }
                }
            }
        }
    };

    // weight fragments of a chunk: planes wh, wl (NT = 16; WH2 = wh * 2^-11 derived), or the bf16 planes
    constexpr int NP = NT == 1 ? 1 : 3, NPB = NT == 1 ? 1 : (NT == 16 ? 2 : 3), NPL = NT == 16 ? 2 : NP;
    u32x4 a[2][MB][NP], an[2][MB][NPL];
    auto wload = [&](int chunk, u32x4(&dstp)[2][MB][NT == 16 ? 2 : NP]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mi = 0; mi < MB; ++mi)
#pragma unroll
                for (int sp = 0; sp < NPL; ++sp)
                    dstp[ks][mi][sp] = *reinterpret_cast<const u32x4 *>(
                        wq + ((((size_t)chunk * 2 + ks) * 3 + sp) * g.Co + (mb0 + mi) * 32 + (lane & 31)) * 16 + (lane >> 5) * 8);
    };
    auto wtake = [&]() {   // the prefetched chunk becomes the current one
        const _Float16 k = (_Float16)0.00048828125f;   // WH2 = wh * 2^-11 (exact while normal, RNE into the denormals like the pack kernel)
        const f16x8 k11 = {k, k, k, k, k, k, k, k};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mi = 0; mi < MB; ++mi) {
#pragma unroll
                for (int sp = 0; sp < NPL; ++sp) a[ks][mi][sp] = an[ks][mi][sp];
                if (NT == 16) a[ks][mi][2] = __builtin_bit_cast(u32x4, __builtin_bit_cast(f16x8, an[ks][mi][0]) * k11);
            }
    };

    // ---- prologue: tap 0 staged for every tile, work item 0 gathered ----
#pragma unroll
    for (int t = 0; t < T; ++t) stage(0, t);
    wload(0, an);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    wtake();
    offs_read(0, 0, 0);
    gather_issue(0, 0, hb[0], wb[0]);
    offs_read(0, 0, 1);
    gather_commit(cols);
    __syncthreads();

    constexpr int TA[6] = {1, 2, 0, 1, 0, 0}, TB[6] = {1, 0, 2, 0, 1, 0};
    int tap = 0, cb = 0;   // of `chunk`
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        int tap1 = tap, cb1 = cb + 1;   // of chunk + 1
        if (cb1 == ncb) cb1 = 0, ++tap1;
        const bool more = chunk + 1 < nchunk;
#pragma unroll
        for (int tile = 0; tile < T; ++tile) {
            // (1) next tap's planes of this tile (requested during the tap's first chunk)
            if (cb == 0 && tap + 1 < 9) stage(tap + 1, tile);
            // (2) gather of the next work item, offsets of the one after it
            if (tile + 1 < T) gather_issue(tap, cb, hb[(tile + 1) % T], wb[(tile + 1) % T]);
            else if (more) gather_issue(tap1, cb1, hb[0], wb[0]);
            if (tile + 2 < T) offs_read(tap, cb, (tile + 2) % T);
            else if (more) offs_read(tap1, cb1, (tile + 2) % T);
            // (3) next chunk's weight fragments during the chunk's last tile
            if (tile == T - 1 && more) wload(chunk + 1, an);
            // (4) this item's column fragments, MFMAs
            u32x4 bv[2][NB][NPB];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int ni = 0; ni < NB; ++ni)
#pragma unroll
                    for (int sp = 0; sp < NPB; ++sp)
                        bv[ks][ni][sp] = *reinterpret_cast<const u32x4 *>(cols + (tile & 1) * CBUF + sp * CQ_PLANE +
                                                                          ((nb0 + ni) * 32 + (lane & 31)) * CQ_LD + ks * 32 + (lane >> 5) * 16);
            __builtin_amdgcn_sched_barrier(0);
            if (NT == 16) {
                constexpr int WA[3] = {1, 2, 0}, CB[3] = {0, 1, 0};  // wl*ah, WH2*AL, wh*ah: smallest first
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int mi = 0; mi < MB; ++mi)
#pragma unroll
                            for (int ni = 0; ni < NB; ++ni)
                                acc[tile][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[ks][mi][WA[t]]),
                                                                                           __builtin_bit_cast(f16x8, bv[ks][ni][NT == 16 ? CB[t] : 0]),
                                                                                           acc[tile][mi][ni], 0, 0, 0);
            } else {
#pragma unroll
                for (int t = (NT == 1 ? 5 : 0); t < 6; ++t)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int mi = 0; mi < MB; ++mi)
#pragma unroll
                            for (int ni = 0; ni < NB; ++ni)
                                acc[tile][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[ks][mi][NT == 6 ? TA[t] : 0]),
                                                                                            __builtin_bit_cast(bf16x8, bv[ks][ni][NT == 6 ? TB[t] : 0]),
                                                                                            acc[tile][mi][ni], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // (5) the next item's columns
            if (tile + 1 < T || more) gather_commit(cols + ((tile + 1) & 1) * CBUF);
            if (tile == T - 1 && more) wtake();
            __syncthreads();
        }
        tap = tap1, cb = cb1;
    }

    if (NT == 16 && range_flag && !(amx_run <= 65000.f)) atomicOr(range_flag, 1);   // a sampled column left the fp16 range
    const float oscale = NT == 16 ? scal[2] : 1.f;  // 1 / S of the weight scaling
    float oamx = 0.f;   // max |out| of the launch -> out_amax (the input scale of the Winograd layers that read `out`, archs/nhwc.py)
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int mi = 0; mi < MB; ++mi)
#pragma unroll
            for (int ni = 0; ni < NB; ++ni) {
                const int px = p0 + 64 * t + (nb0 + ni) * 32 + (lane & 31);
                if (px >= HWo) continue;
                if (out_nhwc) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int o = (mb0 + mi) * 32 + 8 * q + 4 * (lane >> 5);
                        float4 v = make_float4(acc[t][mi][ni][4 * q], acc[t][mi][ni][4 * q + 1], acc[t][mi][ni][4 * q + 2], acc[t][mi][ni][4 * q + 3]);
                        v = epi4(v, oscale, bias, o, slope);
                        oamx = fmaxf(fmaxf(fmaxf(oamx, fabsf(v.x)), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
                        if (NT == 1) {
                            const unsigned int r0 = pk_bf16(v.x, v.y), r1 = pk_bf16(v.z, v.w);
                            if (IO16) {
                                *reinterpret_cast<u32x2 *>(reinterpret_cast<unsigned short *>(out) + ((size_t)b * HWo + px) * g.Co + o) = u32x2{r0, r1};
                                continue;
                            }
                            v = make_float4(__uint_as_float(r0 << 16), __uint_as_float(r0 & 0xffff0000u), __uint_as_float(r1 << 16),
                                            __uint_as_float(r1 & 0xffff0000u));
                        }
                        *reinterpret_cast<float4 *>(out + ((size_t)b * HWo + px) * g.Co + o) = v;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int o = (mb0 + mi) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                        float v = __builtin_fmaf(acc[t][mi][ni][e], oscale, bias ? bias[o] : 0.f);
                        v = v > 0.f ? v : v * slope;
                        if (NT == 1) v = __uint_as_float(pk_bf16(v, 0.f) << 16);
                        oamx = fmaxf(oamx, fabsf(v));
                        out[((size_t)b * g.Co + o) * HWo + px] = v;
                    }
                }
            }
    if (out_amax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) oamx = fmaxf(oamx, __shfl_xor(oamx, o, 64));
        if (lane == 0 && oamx > 0.f && oamx < 3.0e38f && __float_as_uint(oamx) > __builtin_nontemporal_load(out_amax)) atomicMax(out_amax, __float_as_uint(oamx));
    }
}

// ---------------------------------------------------------------------------------------------
// generic forward (any stride / dilation / groups / kernel size / channel count): one thread =
// one pixel x 16 output channels, sampling on the fly.  Correct everywhere, fast nowhere.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dcn_fwd_generic_kernel(const float *__restrict__ x, const float *__restrict__ offset,
                                                              const float *__restrict__ mask, const float *__restrict__ w,
                                                              const float *__restrict__ bias, float *__restrict__ out, Geo g,
                                                              float slope)
{
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
        float acc[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = 0.f;
        for (int cc = 0; cc < cig; ++cc) {
            const int c = gr * cig + cc, dgi = c / cpg;
            const float *im = x + ((size_t)b * g.C + c) * HWi;
            for (int tap = 0; tap < KK; ++tap) {
                const int ti = tap / g.kw, tj = tap - ti * g.kw;
                const float oh = offset[(((size_t)b * g.dg + dgi) * 2 * KK + 2 * tap) * HWo + pix];
                const float ow = offset[(((size_t)b * g.dg + dgi) * 2 * KK + 2 * tap + 1) * HWo + pix];
                const float m = mask ? mask[(((size_t)b * g.dg + dgi) * KK + tap) * HWo + pix] : 1.f;
                const Tap tp = make_tap((float)(ho * g.sh - g.ph + ti * g.dh) + oh, (float)(wo * g.sw - g.pw + tj * g.dw) + ow,
                                        g.H, g.W);
                const float v = (tp.w1 * im[tp.o1] + tp.w2 * im[tp.o2] + tp.w3 * im[tp.o3] + tp.w4 * im[tp.o4]) * m;
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if (k < no) acc[k] = fmaf(w[((size_t)(o0 + k) * cig + cc) * KK + tap], v, acc[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (k < no) {
                float v = acc[k] + (bias ? bias[o0 + k] : 0.f);
                v = v > 0.f ? v : v * slope;
                out[((size_t)b * g.Co + o0 + k) * HWo + pix] = v;
            }
    }
}

// ---------------------------------------------------------------------------------------------
// im2col (for the weight gradient GEMM) and the column -> (offset, mask, input) gradients
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dcn_im2col_kernel(const float *__restrict__ x, const float *__restrict__ offset,
                                                         const float *__restrict__ mask, float *__restrict__ col, Geo g)
{
    const int HWo = g.Ho * g.Wo, HWi = g.H * g.W, KK = g.kh * g.kw, cpg = g.C / g.dg;
    const long total = (long)g.B * g.C * HWo;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int pix = (int)(e % HWo);
        const long t = e / HWo;
        const int c = (int)(t % g.C), b = (int)(t / g.C);
        const int ho = pix / g.Wo, wo = pix - ho * g.Wo, dgi = c / cpg;
        const float *im = x + ((size_t)b * g.C + c) * HWi;
        for (int tap = 0; tap < KK; ++tap) {
            const int ti = tap / g.kw, tj = tap - ti * g.kw;
            const float oh = offset[(((size_t)b * g.dg + dgi) * 2 * KK + 2 * tap) * HWo + pix];
            const float ow = offset[(((size_t)b * g.dg + dgi) * 2 * KK + 2 * tap + 1) * HWo + pix];
            const float m = mask ? mask[(((size_t)b * g.dg + dgi) * KK + tap) * HWo + pix] : 1.f;
            const Tap tp = make_tap((float)(ho * g.sh - g.ph + ti * g.dh) + oh, (float)(wo * g.sw - g.pw + tj * g.dw) + ow,
                                    g.H, g.W);
            const float v = (tp.w1 * im[tp.o1] + tp.w2 * im[tp.o2] + tp.w3 * im[tp.o3] + tp.w4 * im[tp.o4]) * m;
            col[(((size_t)b * g.C + c) * KK + tap) * HWo + pix] = v;
        }
    }
}

// one thread = (b, deformable group, tap, pixel): loops the group's channels; assigns grad_offset
// (y, x) and grad_mask, scatters grad_x with float atomics (deform_conv_cuda_kernel.cu:635-767)
__global__ __launch_bounds__(256) void dcn_col2im_kernel(const float *__restrict__ gcol, const float *__restrict__ x,
                                                         const float *__restrict__ offset, const float *__restrict__ mask,
                                                         float *__restrict__ gx, float *__restrict__ goff,
                                                         float *__restrict__ gmask, Geo g)
{
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
        const float oh = offset[oi], ow = offset[oi + HWo];
        const float m = mask ? mask[mi] : 1.f;
        const Tap tp = make_tap((float)(ho * g.sh - g.ph + ti * g.dh) + oh, (float)(wo * g.sw - g.pw + tj * g.dw) + ow, g.H,
                                g.W);
        // d(sample)/dh, d(sample)/dw as corner coefficient sets (kernel.cu:526-568)
        const float uh = 1.f - tp.lh, uw = 1.f - tp.lw;
        float g_oh = 0.f, g_ow = 0.f, g_m = 0.f;
        // validity of each corner = its position is in range (its weight may still be 0 by value)
        const float fh = floorf((float)(ho * g.sh - g.ph + ti * g.dh) + oh);
        const float fw = floorf((float)(wo * g.sw - g.pw + tj * g.dw) + ow);
        const int hl = (int)fh, wl = (int)fw, hh = hl + 1, wh = wl + 1;
        const bool v1 = tp.inside && hl >= 0 && wl >= 0, v2 = tp.inside && hl >= 0 && wh <= g.W - 1;
        const bool v3 = tp.inside && hh <= g.H - 1 && wl >= 0, v4 = tp.inside && hh <= g.H - 1 && wh <= g.W - 1;
        for (int cc = 0; cc < cpg; ++cc) {
            const int c = dgi * cpg + cc;
            const float gc = gcol[(((size_t)b * g.C + c) * KK + tap) * HWo + pix];
            const float *im = x + ((size_t)b * g.C + c) * HWi;
            const float x1 = v1 ? im[tp.o1] : 0.f, x2 = v2 ? im[tp.o2] : 0.f, x3 = v3 ? im[tp.o3] : 0.f,
                        x4 = v4 ? im[tp.o4] : 0.f;
            g_m = fmaf(gc, tp.w1 * x1 + tp.w2 * x2 + tp.w3 * x3 + tp.w4 * x4, g_m);
            g_oh = fmaf(gc * m, -uw * x1 - tp.lw * x2 + uw * x3 + tp.lw * x4, g_oh);
            g_ow = fmaf(gc * m, -uh * x1 + uh * x2 - tp.lh * x3 + tp.lh * x4, g_ow);
            if (gx) {
                float *gi = gx + ((size_t)b * g.C + c) * HWi;
                const float gv = gc * m;
                if (v1) atomicAdd(gi + tp.o1, gv * tp.w1);
                if (v2) atomicAdd(gi + tp.o2, gv * tp.w2);
                if (v3) atomicAdd(gi + tp.o3, gv * tp.w3);
                if (v4) atomicAdd(gi + tp.o4, gv * tp.w4);
            }
        }
        goff[oi] = g_oh;
        goff[oi + HWo] = g_ow;
        if (gmask && mask) gmask[mi] = g_m;
    }
}

bool mfma_eligible(const Geo &g)
{
    const int cpg = g.C / g.dg;
    return g.groups == 1 && g.kh == 3 && g.kw == 3 && (g.C % 32 == 0) && (cpg % 8 == 0) && (32 % cpg == 0 || cpg % 32 == 0) &&
           (g.Co == 64 || g.Co == 128 || g.Co == 256);
}

}  // namespace

MREFSR_EXPORT int64_t mrefsr_dcn_fwd_workspace_bytes(const mrefsr_dcn_shape *s)
{
    Geo g;
    if (make_geo(s, g, "dcn_fwd_workspace_bytes")) return -1;
    return mfma_eligible(g) ? (int64_t)g.Co * g.C * 9 * 6 + 64 : 0;  // fp32 repack (4 B) or three 16-bit planes (6 B) per weight + the fp16 scale
}

static int dcn_fwd_entry(const float *x, const float *offset, const float *mask, const float *weight, const float *bias, float *out,
                         const mrefsr_dcn_shape *s, float act_slope, int nhwc, void *workspace, int64_t workspace_bytes, int *range_flag,
                         unsigned int *out_amax, mrefsr_stream_t stream);

MREFSR_EXPORT int mrefsr_dcn_fwd_f32(const float *x, const float *offset, const float *mask, const float *weight,
                                     const float *bias, float *out, const mrefsr_dcn_shape *s, float act_slope,
                                     int nhwc, void *workspace, int64_t workspace_bytes, int *range_flag, mrefsr_stream_t stream)
{
    return dcn_fwd_entry(x, offset, mask, weight, bias, out, s, act_slope, nhwc, workspace, workspace_bytes, range_flag, nullptr, stream);
}

// mrefsr_dcn_fwd_f32 + out_amax[0] = max(out_amax[0], max |out|) (device memory, zero-initialised by the caller): the input scale of
// the Winograd convolutions that read `out` (MRAPAFusion's conv_emb2 / conv_ass, ref_mrapa_restoration_arch.py:271-304).  Channels-last
// x on the 16-bit matrix pipe only (nhwc bit 0).
MREFSR_EXPORT int mrefsr_dcn_fwd_amax_f32(const float *x, const float *offset, const float *mask, const float *weight,
                                          const float *bias, float *out, const mrefsr_dcn_shape *s, float act_slope,
                                          int nhwc, void *workspace, int64_t workspace_bytes, int *range_flag, float *out_amax,
                                          mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(!out_amax || (nhwc & 1), "dcn_fwd_amax: out_amax is written by the channels-last kernels (nhwc bit 0)");
    return dcn_fwd_entry(x, offset, mask, weight, bias, out, s, act_slope, nhwc, workspace, workspace_bytes, range_flag,
                         reinterpret_cast<unsigned int *>(out_amax), stream);
}

static int dcn_fwd_entry(const float *x, const float *offset, const float *mask, const float *weight, const float *bias, float *out,
                         const mrefsr_dcn_shape *s, float act_slope, int nhwc, void *workspace, int64_t workspace_bytes, int *range_flag,
                         unsigned int *out_amax, mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(x && offset && weight && out, "dcn_fwd: null pointer");
    Geo g;
    if (int e = make_geo(s, g, "dcn_fwd")) return e;
    const int x_nhwc = nhwc & 1, out_nhwc = (nhwc >> 1) & 1, bf16_arith = (nhwc >> 2) & 1, range_free = (nhwc >> 3) & 1, io16 = (nhwc >> 4) & 1;
    MREFSR_REQUIRE(!io16 || (bf16_arith && x_nhwc && out_nhwc), "dcn_fwd: bf16 storage (nhwc bit 4) goes with bits 0, 1 and 2");
    MREFSR_REQUIRE(!bf16_arith || x_nhwc, "dcn_fwd: bf16 arithmetic (nhwc bit 2) is implemented for channels-last input only");
    MREFSR_REQUIRE(!nhwc || mfma_eligible(g), "dcn_fwd: NHWC x / out is only implemented by the MFMA path (see mrefsr_dcn_fwd_workspace_bytes > 0)");
    hipStream_t st = (hipStream_t)stream;
    const int HWo = g.Ho * g.Wo;
    if (mfma_eligible(g)) {
        const int64_t need = (int64_t)g.Co * g.C * 9 * 6 + 64;
        MREFSR_REQUIRE(workspace && workspace_bytes >= need, "dcn_fwd: workspace of %ld bytes required (got %ld)", (long)need,
                       (long)workspace_bytes);
        float *wp = (float *)workspace;
        const long tot = (long)g.Co * g.C * 9;
        static const int xcd_order = [] { const char *e = getenv("MREFSR_DCN_XCD"); return (e && e[0] == '0') ? 0 : 1; }();
        const long nblk = (long)mrefsr::cdiv(HWo, 64) * g.B;
        dim3 grid((unsigned)(xcd_order ? ((nblk + 7) / 8) * 8 : nblk));
        // Channels-last input runs on the 16-bit matrix pipe from exact operand splits (fp32-equivalent results):
        // fp16 two-term / three products by default, MREFSR_DCN_TERMS=6 the bf16 three-term / six-product split (no range
        // limit); MREFSR_DCN_BF16=0 selects the fp32-MFMA kernel below instead (A/B measurements).
        static const int use_bf16 = [] { const char *e = getenv("MREFSR_DCN_BF16"); return (e && e[0] == '0') ? 0 : 1; }();
        static const int terms = [] { const char *e = getenv("MREFSR_DCN_TERMS"); return (e && e[0] == '6') ? 6 : 16; }();
        if (x_nhwc && (use_bf16 || bf16_arith)) {
            unsigned short *wq = (unsigned short *)workspace;
            float *scal = (float *)((char *)workspace + (int64_t)g.Co * g.C * 9 * 6);
            const int nt = bf16_arith ? 1 : range_free ? 6 : terms;
            if (nt == 16) {
                if (hipMemsetAsync(scal, 0, 16, st) != hipSuccess) return mrefsr::check_launch("dcn_fwd(fp16 split): memset");
                hipLaunchKernelGGL(dcn_weight_amax_kernel, dim3(64), dim3(256), 0, st, weight, (unsigned int *)scal, tot);
                hipLaunchKernelGGL(dcn_pack_weight_f16_kernel, dim3((int)((tot + 255) / 256)), dim3(256), 0, st, weight, wq, scal, g.Co, g.C);
            } else {
                hipLaunchKernelGGL(dcn_pack_weight_bf16_kernel, dim3((int)((tot + 255) / 256)), dim3(256), 0, st, weight, wq, g.Co, g.C);
            }
            // Round 6: T pixel tiles per block, chunk-outer (dcn_fwd_pt_kernel): a tap's 3 dg offset / mask planes must fit LDS
            // beside the column buffers and C / dg be a power of two.  MREFSR_DCN_PT=0: the one-tile kernel (A/B runs).
            const char *e_pt = getenv("MREFSR_DCN_PT"), *e_t = getenv("MREFSR_DCN_T");   // (read per call: the tests flip them)
            const int use_pt = !(e_pt && e_pt[0] == '0'), env_t = e_t ? atoi(e_t) : 0;
            const int cpg = g.C / g.dg, n_pl = (mask ? 3 : 2) * g.dg;
            // (Co = 256: two tiles of a 64 x 64 wave tile are 128 accumulator registers -- one wave per SIMD; the one-tile kernel stays)
            if (use_pt && (cpg & (cpg - 1)) == 0 && (g.C & (g.C - 1)) == 0 && (size_t)g.H * g.W * g.C * 4 < ((size_t)1 << 32) && n_pl <= 24 &&
                g.C >= 64 && (g.Co < 256 || (e_pt && e_pt[0] == '2'))) {
                int cpg_shift = 0, cshift = 0;
                while ((1 << cpg_shift) < cpg) ++cpg_shift;
                while ((1 << cshift) < g.C * 4) ++cshift;   // (log2 of a pixel's bytes: the paired gather shifts pixel indices into byte offsets)
                const int T = (g.Co == 64 && env_t == 4) ? 4 : 2;   // (two tiles: three blocks per CU -- measured faster than four tiles at two, tools/dcn_ab.py)
                const long ngrp = (long)mrefsr::cdiv(mrefsr::cdiv(HWo, 64), T) * g.B;
                dim3 pgrid((unsigned)(xcd_order ? ((ngrp + 7) / 8) * 8 : ngrp));
                const size_t lds = 2 * pt_cols_bytes(nt) + (size_t)2 * n_pl * (64 * T + 4) * sizeof(float);
#define MREFSR_DCNPT_K(MB, NB, M8, NT, IO, TT)                                                                                              \
    do {                                                                                                                                  \
        static unsigned long long attr_done = 0;                                                                                          \
        if (mrefsr::first_use_on_device(attr_done))                                                                                       \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(dcn_fwd_pt_kernel<MB, NB, NT, M8, IO, TT>),                          \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 2 * pt_cols_bytes(NT) + 2 * 24 * (64 * TT + 4) * 4);                \
        hipLaunchKernelGGL((dcn_fwd_pt_kernel<MB, NB, NT, M8, IO, TT>), pgrid, dim3(256), lds, st, x, offset, mask, wq, bias, out, g,     \
                           act_slope, out_nhwc, xcd_order, scal, range_flag, cpg_shift, n_pl, cshift, out_amax);                                            \
    } while (0)
#define MREFSR_DCNPT_T(MB, NB, M8, NT, IO)        \
    do {                                          \
        if (T == 4) MREFSR_DCNPT_K(MB, NB, M8, NT, IO, 4); \
        else MREFSR_DCNPT_K(MB, NB, M8, NT, IO, 2);        \
    } while (0)
#define MREFSR_DCNPT(MB, NB, M8, TSEL)                        \
    do {                                                      \
        if (nt == 1 && io16) TSEL(MB, NB, M8, 1, true);       \
        else if (nt == 1) TSEL(MB, NB, M8, 1, false);         \
        else if (nt == 6) TSEL(MB, NB, M8, 6, false);         \
        else TSEL(MB, NB, M8, 16, false);                     \
    } while (0)
#define MREFSR_DCNPT_T2(MB, NB, M8, NT, IO) MREFSR_DCNPT_K(MB, NB, M8, NT, IO, 2)
                // (four tiles = 64 accumulator registers per 32 x 32 wave tile: only the Co = 64 shapes have room for them)
                // (the paired four-channel mapping also where a deformable group is 16 channels: the lanes of a quad then form their two
                //  setups twice, which costs no instruction, and every corner is one 64-byte request: 6.4 -> 6.0 ms at C = 128;
                //  MREFSR_DCN_MAP8=1: the eight-channel mapping, A/B runs)
                const char *e_m8 = getenv("MREFSR_DCN_MAP8");
                const bool map8 = e_m8 && e_m8[0] == '1';
                if (g.Co == 256) MREFSR_DCNPT(2, 2, true, MREFSR_DCNPT_T2);
                else if (g.Co == 128 && map8) MREFSR_DCNPT(1, 2, true, MREFSR_DCNPT_T2);
                else if (g.Co == 128) MREFSR_DCNPT(1, 2, false, MREFSR_DCNPT_T2);
                else if (g.C >= 128) MREFSR_DCNPT(1, 1, true, MREFSR_DCNPT_T);
                else MREFSR_DCNPT(1, 1, false, MREFSR_DCNPT_T);
#undef MREFSR_DCNPT_T2
#undef MREFSR_DCNPT
#undef MREFSR_DCNPT_T
#undef MREFSR_DCNPT_K
                return mrefsr::check_launch("dcn_fwd(pt)");
            }
#define MREFSR_DCN16_NT(MB, NB, M8, NT)                                                                                            \
    hipLaunchKernelGGL((dcn_fwd_bf16_kernel<MB, NB, NT, M8>), grid, dim3(256), 0, st, x, offset, mask, wq, bias, out, g, act_slope, \
                       out_nhwc, xcd_order, scal, range_flag, out_amax)
#define MREFSR_DCN16(MB, NB, M8)                      \
    do {                                              \
        if (nt == 1 && io16)                          \
            hipLaunchKernelGGL((dcn_fwd_bf16_kernel<MB, NB, 1, M8, true>), grid, dim3(256), 0, st, x, offset, mask, wq, bias, out, g, act_slope, \
                               out_nhwc, xcd_order, scal, range_flag, out_amax);                                                                       \
        else if (nt == 1) MREFSR_DCN16_NT(MB, NB, M8, 1);  \
        else if (nt == 6) MREFSR_DCN16_NT(MB, NB, M8, 6); \
        else MREFSR_DCN16_NT(MB, NB, M8, 16);         \
    } while (0)
            // a wave covers both 32-pixel tiles (NB = 2) of Co / 128 cout tiles where Co allows
            if (g.Co == 256) MREFSR_DCN16(2, 2, true);
            else if (g.Co == 128) MREFSR_DCN16(1, 2, true);
            else if (g.C >= 128) MREFSR_DCN16(1, 1, true);
            else MREFSR_DCN16(1, 1, false);
#undef MREFSR_DCN16
#undef MREFSR_DCN16_NT
            return mrefsr::check_launch("dcn_fwd(bf16 split)");
        }
        hipLaunchKernelGGL(dcn_pack_weight_kernel, dim3((int)((tot + 255) / 256)), dim3(256), 0, st, weight, wp, g.Co, g.C);
#define MREFSR_DCN_LAUNCH(MB, NB)                                                                                          \
    do {                                                                                                                  \
        if (x_nhwc)                                                                                                       \
            hipLaunchKernelGGL((dcn_fwd_mfma_kernel<MB, NB, 1>), grid, dim3(256), 0, st, x, offset, mask, wp, bias, out, g, act_slope, out_nhwc, xcd_order); \
        else                                                                                                              \
            hipLaunchKernelGGL((dcn_fwd_mfma_kernel<MB, NB, 0>), grid, dim3(256), 0, st, x, offset, mask, wp, bias, out, g, act_slope, out_nhwc, xcd_order); \
    } while (0)
        if (g.Co == 256) MREFSR_DCN_LAUNCH(2, 2);
        else if (g.Co == 128) MREFSR_DCN_LAUNCH(1, 2);
        else MREFSR_DCN_LAUNCH(1, 1);
#undef MREFSR_DCN_LAUNCH
        return mrefsr::check_launch("dcn_fwd(mfma)");
    }
    const long total = (long)g.B * g.groups * ((g.Co / g.groups + 15) / 16) * HWo;
    const long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(dcn_fwd_generic_kernel, dim3((int)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, st, x, offset, mask,
                       weight, bias, out, g, act_slope);
    return mrefsr::check_launch("dcn_fwd(generic)");
}

MREFSR_EXPORT int mrefsr_dcn_im2col_f32(const float *x, const float *offset, const float *mask, float *columns,
                                        const mrefsr_dcn_shape *s, mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(x && offset && columns, "dcn_im2col: null pointer");
    Geo g;
    if (int e = make_geo(s, g, "dcn_im2col")) return e;
    const long total = (long)g.B * g.C * g.Ho * g.Wo;
    const long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(dcn_im2col_kernel, dim3((int)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, (hipStream_t)stream, x,
                       offset, mask, columns, g);
    return mrefsr::check_launch("dcn_im2col");
}

MREFSR_EXPORT int mrefsr_dcn_col2im_f32(const float *grad_col, const float *x, const float *offset, const float *mask,
                                        float *grad_x, float *grad_offset, float *grad_mask, const mrefsr_dcn_shape *s,
                                        mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(grad_col && x && offset && grad_offset, "dcn_col2im: null pointer");
    Geo g;
    if (int e = make_geo(s, g, "dcn_col2im")) return e;
    const long total = (long)g.B * g.dg * g.kh * g.kw * g.Ho * g.Wo;
    const long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(dcn_col2im_kernel, dim3((int)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, (hipStream_t)stream,
                       grad_col, x, offset, mask, grad_x, grad_offset, grad_mask, g);
    return mrefsr::check_launch("dcn_col2im");
}
