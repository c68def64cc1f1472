// Backward of the modulated deformable convolution (DCNv2, 3x3, stride 1, pad 1, one weight group) as TWO fused kernels --
// what the reference runs as im2col / GEMM / GEMM / col2im around a C*9*H*W column buffer per sample
// (basicsr/ops/dcn/src/deform_conv_cuda.cpp:571-685, deform_conv_cuda_kernel.cu:635-767; mmcv's twin in the shipped model):
//
//   mrefsr_dcn_bwd_data_f32    d columns = W^T . g_out on the matrix pipe per tile of 32 pixels, and -- as that GEMM's epilogue,
//                              straight from the accumulators -- the gradients of offset, mask and input: the column gradient never
//                              exists in memory.  MFMA result layout: lane = pixel, registers = channels of one tap, so a lane
//                              re-forms the bilinear set-up of its (pixel, tap, deformable group) once, reads the four corners of its
//                              channels as 16-byte vectors of the channels-last input, and scatters the input gradient with lanes
//                              of a wave on neighbouring pixels of one channel plane (float atomics coalesce there; with the lanes
//                              on the channels of a pixel they ran 2.4-3x slower).
//   mrefsr_dcn_bwd_weight_f32  d W = g_out . columns^T with the columns re-gathered tile by tile (as the forward does) instead of
//                              read back from memory: pixel-K GEMM, both operands transposed through LDS.
// Arithmetic: the fp16 two-term split of conv_nhwc.hip (three MFMAs per product, fp32-equivalent); the output gradient is scaled
// into the fp16 range by the power of two that brings max |g| (device memory, from the activation backward) to [2^13, 2^14).
#include "conv_common.h"

namespace {
using namespace mrefsr_conv;

__device__ __forceinline__ f32x16 mma16(u32x4 a, u32x4 b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

struct BwdArgs {
    const float *gy, *x, *offset, *mask;   // g_out [B][H][W][Co], x [B][H][W][C] channels-last; offset [B][18 dg][H][W], mask [B][9 dg][H][W] planar
    const unsigned short *wp;              // W^T packed as the (unmirrored) input-gradient operator: conv_nhwc.hip's terms-16 layout
    const float *g_amax;
    float *gx, *goff, *gmask;              // gx planar [B][C][H][W] (zero-initialised, atomics) or NULL; goff / gmask as offset / mask
    int B, C, Co, H, W, dg, n_ch;
    float inv_wscale;
};

// the bilinear set-up of one (pixel, tap, deformable group): deform_conv_cuda_kernel.cu:467-497 (value) and :526-568 (derivatives)
struct Corner {
    int o[4];        // pixel index of the four corners, clamped into the image
    float w[4];      // bilinear weights, 0 where the corner is out of range or the sample outside the validity window
    float dh[4], dw[4];   // d(sample) / d(h), d(w) coefficients of the four corner values (0 for a corner out of range)
};
__device__ __forceinline__ Corner make_corner(const float hi, const float wi, const int H, const int W)
{
    Corner c;
    const bool inside = (hi > -1.f) && (wi > -1.f) && (hi < (float)H) && (wi < (float)W);
    const float fh = floorf(hi), fw = floorf(wi);
    const int hl = (int)fh, wl = (int)fw, hh = hl + 1, wh = wl + 1;
    const float lh = hi - fh, lw = wi - fw, uh = 1.f - lh, uw = 1.f - lw;
    const bool v1 = inside && hl >= 0 && wl >= 0, v2 = inside && hl >= 0 && wh <= W - 1;
    const bool v3 = inside && hh <= H - 1 && wl >= 0, v4 = inside && hh <= H - 1 && wh <= W - 1;
    const int chl = min(max(hl, 0), H - 1), chh = min(max(hh, 0), H - 1), cwl = min(max(wl, 0), W - 1), cwh = min(max(wh, 0), W - 1);
    c.o[0] = chl * W + cwl, c.o[1] = chl * W + cwh, c.o[2] = chh * W + cwl, c.o[3] = chh * W + cwh;
    // (four scalar products kept apart: the crossed packed form the vectoriser would build is unsafe beside 16-bit MFMAs, dcn.hip)
    float p1 = uh * uw, p2 = uh * lw, p3 = lh * uw, p4 = lh * lw;
    asm volatile("" : "+v"(p1));
    asm volatile("" : "+v"(p2));
    asm volatile("" : "+v"(p3));
    asm volatile("" : "+v"(p4));
    c.w[0] = v1 ? p1 : 0.f, c.w[1] = v2 ? p2 : 0.f, c.w[2] = v3 ? p3 : 0.f, c.w[3] = v4 ? p4 : 0.f;
    c.dh[0] = v1 ? -uw : 0.f, c.dh[1] = v2 ? -lw : 0.f, c.dh[2] = v3 ? uw : 0.f, c.dh[3] = v4 ? lw : 0.f;
    c.dw[0] = v1 ? -uh : 0.f, c.dw[1] = v2 ? uh : 0.f, c.dw[2] = v3 ? -lh : 0.f, c.dw[3] = v4 ? lh : 0.f;
    return c;
}

// CPG = channels per deformable group (8, 16 or 32): a 32-channel row block of the GEMM holds 32 / CPG whole groups
template <int CPG>
__global__ __launch_bounds__(256) void dcn_bwd_data_kernel(const BwdArgs A)
{
    extern __shared__ __align__(16) unsigned char smem[];   // g_out tile, split: [plane 2][co chunk][k half 2][pixel 32][8 co] fp16
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, kh = lane >> 5;
    const int HW = A.H * A.W, tiles = (HW + 31) / 32;
    const int b = blockIdx.x / tiles, p0 = (blockIdx.x - b * tiles) * 32;
    const int Co = A.Co, C = A.C, n_ch = A.n_ch;
    float in_s = 1.f, oscale = A.inv_wscale;
    if (A.g_amax) {
        const float am = *A.g_amax;
        if (am > 1.0e-30f && am < 3.0e38f) {
            int e;
            (void)frexpf(am, &e);
            in_s = ldexpf(1.f, 14 - e);
            oscale = A.inv_wscale * ldexpf(1.f, e - 14);
        }
    }
    // ---- the tile's output gradient -> two fp16 planes in LDS (rows of g_out are contiguous: coalesced 16-byte loads)
    const int q_per_px = Co >> 2;
    for (int i = tid; i < 32 * q_per_px; i += 256) {
        const int px = i / q_per_px, co = (i - px * q_per_px) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p0 + px < HW) v = *reinterpret_cast<const float4 *>(A.gy + ((size_t)b * HW + p0 + px) * Co + co);
        v.x *= in_s, v.y *= in_s, v.z *= in_s, v.w *= in_s;
        const unsigned int h0 = pk_f16(v.x, v.y), h1 = pk_f16(v.z, v.w);
        const f32x2 f0 = un_f16(h0), f1 = un_f16(h1);
        const u32x2 hi = {h0, h1}, lo = {pk_f16((v.x - f0[0]) * 2048.f, (v.y - f0[1]) * 2048.f), pk_f16((v.z - f1[0]) * 2048.f, (v.w - f1[1]) * 2048.f)};
        unsigned char *d = smem + (co >> 4) * 1024 + ((co >> 3) & 1) * 512 + px * 16 + ((co >> 2) & 1) * 8;
        *reinterpret_cast<u32x2 *>(d) = hi;
        *reinterpret_cast<u32x2 *>(d + n_ch * 1024) = lo;
    }
    __syncthreads();

    const int p = p0 + l31;
    const bool pok = p < HW;
    const int ph = pok ? p / A.W : 0, pw = pok ? p - ph * A.W : 0;
    const float *xb = A.x + (size_t)b * HW * C;
    const int n_cblk = C >> 5;
    for (int rb = wv; rb < 9 * n_cblk; rb += 4) {   // row block = 32 channels of one tap
        const int tap = rb / n_cblk, cblk = rb - tap * n_cblk;
        // ---- d columns [32 channels x 32 pixels] = W^T . g_out, three partial products per k step
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        const unsigned short *wrow = A.wp + ((size_t)(cblk >> 1) * n_ch * 9 + tap) * (3 * NB * KC) + ((cblk & 1) * 32 + l31) * KC + kh * 8;
        const unsigned char *grow = smem + kh * 512 + l31 * 16;
        for (int ch = 0; ch < n_ch; ++ch) {
            const unsigned short *w = wrow + (size_t)ch * 9 * (3 * NB * KC);
            const u32x4 wh = *reinterpret_cast<const u32x4 *>(w), wl = *reinterpret_cast<const u32x4 *>(w + NB * KC),
                        wh2 = *reinterpret_cast<const u32x4 *>(w + 2 * NB * KC);
            const u32x4 gh = *reinterpret_cast<const u32x4 *>(grow + ch * 1024), gl = *reinterpret_cast<const u32x4 *>(grow + (n_ch + ch) * 1024);
            acc = mma16(wl, gh, acc);
            acc = mma16(wh2, gl, acc);
            acc = mma16(wh, gh, acc);
        }
        // ---- epilogue: register e = channel 32 cblk + 8 (e >> 2) + 4 kh + (e & 3) of pixel l31
        const int ti = tap / 3, tj = tap - 3 * ti;
        float g_m = 0.f, g_oh = 0.f, g_ow = 0.f, mval = 1.f;
        Corner cn;
        int grp = -1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c0 = cblk * 32 + 8 * q + 4 * kh;
            const int g = c0 / CPG;
            if (g != grp) {   // (compile-time pattern: once per row block for CPG 32, twice for 16, every q for 8)
                grp = g;
                const size_t oi = (((size_t)b * A.dg + g) * 18 + 2 * tap) * HW + (pok ? p : 0);
                const float oh = A.offset[oi], ow = A.offset[oi + HW];
                mval = A.mask ? A.mask[(((size_t)b * A.dg + g) * 9 + tap) * HW + (pok ? p : 0)] : 1.f;
                cn = make_corner((float)(ph - 1 + ti) + oh, (float)(pw - 1 + tj) + ow, A.H, A.W);
                g_m = g_oh = g_ow = 0.f;
            }
            float4 xc[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) xc[k] = *reinterpret_cast<const float4 *>(xb + (size_t)cn.o[k] * C + c0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float gc = acc[4 * q + j] * oscale;
                const float x1 = j == 0 ? xc[0].x : (j == 1 ? xc[0].y : (j == 2 ? xc[0].z : xc[0].w));
                const float x2 = j == 0 ? xc[1].x : (j == 1 ? xc[1].y : (j == 2 ? xc[1].z : xc[1].w));
                const float x3 = j == 0 ? xc[2].x : (j == 1 ? xc[2].y : (j == 2 ? xc[2].z : xc[2].w));
                const float x4 = j == 0 ? xc[3].x : (j == 1 ? xc[3].y : (j == 2 ? xc[3].z : xc[3].w));
                g_m = fmaf(gc, cn.w[0] * x1 + cn.w[1] * x2 + cn.w[2] * x3 + cn.w[3] * x4, g_m);
                const float gv = gc * mval;
                g_oh = fmaf(gv, cn.dh[0] * x1 + cn.dh[1] * x2 + cn.dh[2] * x3 + cn.dh[3] * x4, g_oh);
                g_ow = fmaf(gv, cn.dw[0] * x1 + cn.dw[1] * x2 + cn.dw[2] * x3 + cn.dw[3] * x4, g_ow);
                if (A.gx && pok) {
                    float *gi = A.gx + ((size_t)b * C + c0 + j) * HW;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (cn.w[k] != 0.f) atomicAdd(gi + cn.o[k], gv * cn.w[k]);
                }
            }
            const bool last_of_group = ((8 * (q + 1)) % CPG) == 0;
            if (last_of_group) {   // the group's other channels of this pixel: the partner lane (k half)
                g_m += __shfl_xor(g_m, 32, 64), g_oh += __shfl_xor(g_oh, 32, 64), g_ow += __shfl_xor(g_ow, 32, 64);
                if (kh == 0 && pok) {
                    const size_t oi = (((size_t)b * A.dg + g) * 18 + 2 * tap) * HW + p;
                    A.goff[oi] = g_oh, A.goff[oi + HW] = g_ow;
                    if (A.gmask && A.mask) A.gmask[(((size_t)b * A.dg + g) * 9 + tap) * HW + p] = g_m;
                }
                grp = -1;
            }
        }
    }
}


// ---- d W[o][c][tap] = sum over (b, pixel) of g_out[b][pixel][o] * column[b][pixel][c][tap]: a GEMM whose K dimension is the pixel
// index.  Block = 64 couts x (32 channels x 9 taps), K tiles of 32 pixels: per tile the output gradient goes TRANSPOSED into LDS
// ([o][pixel] fp16 planes gh | gl of g S) and the columns are gathered -- bilinear set-up per (pixel, tap, group), four corners of
// 4 channels as 16-byte vectors of the channels-last input, times the mask -- and stored transposed as well ([tap][c][pixel] planes
// xh | XL = fp16((x - xh) 2^11)); a thread owns two adjacent pixels, so every LDS store is one packed pair.  Wave (cout half, tap
// group 0-4 / 5-8) multiplies gh xh + gl xh + GH2 XL (GH2 = gh 2^-11 in registers).  A block walks every S-th tile and leaves its
// 64 x 288 partial in the workspace ([split][tap][o][c]); dcn_bwd_weight_reduce_kernel adds the partials in split order
// (deterministic) and scales.
struct BwdWArgs {
    const float *gy, *x, *offset, *mask, *g_amax;
    float *partial;
    int B, C, Co, H, W, dg, splits;
    int *range_flag;
};
constexpr int WG_BYTES = 2 * 2 * 2 * 64 * 16;        // [plane][k step][k half][o 64][8 px]
constexpr int WC_BYTES = 2 * 9 * 2 * 2 * 32 * 16;    // [plane][tap][k step][k half][c 32][8 px]

__global__ __launch_bounds__(256) void dcn_bwd_weight_kernel(const BwdWArgs A)
{
    __shared__ __align__(16) unsigned char smem[WG_BYTES + WC_BYTES];
    unsigned char *const sg = smem, *const sc = smem + WG_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, kh = lane >> 5;
    const int HW = A.H * A.W, tiles_img = (HW + 31) / 32, tiles = A.B * tiles_img;
    const int c0 = blockIdx.y * 32, o0 = blockIdx.z * 64, C = A.C, Co = A.Co, cpg = C / A.dg;
    float gs = 1.f;
    if (A.g_amax) {
        const float am = *A.g_amax;
        if (am > 1.0e-30f && am < 3.0e38f) {
            int e;
            (void)frexpf(am, &e);
            gs = ldexpf(1.f, 14 - e);
        }
    }
    const int ot = wv & 1, tg = wv >> 1, t_lo = tg ? 5 : 0, n_t = tg ? 4 : 5;   // wave: cout half, taps [t_lo, t_lo + n_t)
    f32x16 acc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    // staging roles: the gradient -- thread = (cout quad gq, pixel pair pj); the columns -- thread = (channel quad cq, pixel pair pj, tap half)
    const int gq = tid & 15, gpj = tid >> 4;
    const int cq = tid & 7, cpj = (tid >> 3) & 15, cth = tid >> 7;
    float xmax = 0.f;
    for (int kt = blockIdx.x; kt < tiles; kt += A.splits) {
        const int b = kt / tiles_img, p0 = (kt - b * tiles_img) * 32;
        {   // ---- g_out^T: pixels p0 + 2 gpj, + 1; couts o0 + 4 gq .. + 3
            float4 v[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int p = p0 + 2 * gpj + k;
                v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p < HW && o0 + 4 * gq < Co) v[k] = *reinterpret_cast<const float4 *>(A.gy + ((size_t)b * HW + p) * Co + o0 + 4 * gq);
            }
            const float a0[4] = {v[0].x * gs, v[0].y * gs, v[0].z * gs, v[0].w * gs}, a1[4] = {v[1].x * gs, v[1].y * gs, v[1].z * gs, v[1].w * gs};
            const int px = 2 * gpj;
            unsigned char *d = sg + (((px >> 4) * 2 + ((px >> 3) & 1)) * 1024) + (px & 7) * 2;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned int h = pk_f16(a0[i], a1[i]);
                const f32x2 f = un_f16(h);
                *reinterpret_cast<unsigned int *>(d + (4 * gq + i) * 16) = h;
                *reinterpret_cast<unsigned int *>(d + 4096 + (4 * gq + i) * 16) = pk_f16(a0[i] - f[0], a1[i] - f[1]);
            }
        }
        {   // ---- columns: pixels p0 + 2 cpj, + 1; channels c0 + 4 cq .. + 3; taps of this thread's half
            const int g = (c0 + 4 * cq) / cpg;
            const float *xb = A.x + (size_t)b * HW * C + c0 + 4 * cq;
            const int px = 2 * cpj;
            unsigned char *dbase = sc + (((px >> 4) * 2 + ((px >> 3) & 1)) * 512) + (4 * cq) * 16 + (px & 7) * 2;
            for (int tap = cth ? 5 : 0; tap < (cth ? 9 : 5); ++tap) {
                const int ti = tap / 3, tj = tap - 3 * ti;
                float val[2][4];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int p = p0 + px + k;
                    const bool pok = p < HW;
                    const int pp = pok ? p : 0, ph = pp / A.W, pw = pp - ph * A.W;
                    const size_t oi = (((size_t)b * A.dg + g) * 18 + 2 * tap) * HW + pp;
                    const float oh = A.offset[oi], ow = A.offset[oi + HW];
                    const float m = pok ? (A.mask ? A.mask[(((size_t)b * A.dg + g) * 9 + tap) * HW + pp] : 1.f) : 0.f;
                    const Corner cn = make_corner((float)(ph - 1 + ti) + oh, (float)(pw - 1 + tj) + ow, A.H, A.W);
                    float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 xv = *reinterpret_cast<const float4 *>(xb + (size_t)cn.o[q] * C);
                        s4.x = fmaf(cn.w[q], xv.x, s4.x), s4.y = fmaf(cn.w[q], xv.y, s4.y), s4.z = fmaf(cn.w[q], xv.z, s4.z), s4.w = fmaf(cn.w[q], xv.w, s4.w);
                    }
                    val[k][0] = s4.x * m, val[k][1] = s4.y * m, val[k][2] = s4.z * m, val[k][3] = s4.w * m;
                }
                unsigned char *d = dbase + tap * 2048;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    xmax = fmaxf(xmax, fmaxf(fabsf(val[0][i]), fabsf(val[1][i])));
                    const unsigned int h = pk_f16(val[0][i], val[1][i]);
                    const f32x2 f = un_f16(h);
                    *reinterpret_cast<unsigned int *>(d + i * 16) = h;
                    *reinterpret_cast<unsigned int *>(d + 9 * 2048 + i * 16) = pk_f16((val[0][i] - f[0]) * 2048.f, (val[1][i] - f[1]) * 2048.f);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const unsigned char *ga = sg + ((ks * 2 + kh) * 1024) + (ot * 32 + l31) * 16;
            const u32x4 gh = *reinterpret_cast<const u32x4 *>(ga), gl = *reinterpret_cast<const u32x4 *>(ga + 4096);
            const u32x4 gh2 = scale_wh(gh);
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                if (t < n_t) {
                    const unsigned char *ca = sc + (t_lo + t) * 2048 + ((ks * 2 + kh) * 512) + l31 * 16;
                    const u32x4 xh = *reinterpret_cast<const u32x4 *>(ca), xl = *reinterpret_cast<const u32x4 *>(ca + 9 * 2048);
                    acc[t] = mma16(gl, xh, acc[t]);
                    acc[t] = mma16(gh2, xl, acc[t]);
                    acc[t] = mma16(gh, xh, acc[t]);
                }
            }
        }
        __syncthreads();
    }
    if (A.range_flag && !(xmax <= 65000.f)) atomicOr(A.range_flag, 1);
    // partial [split][tap][Co][C]: lane = channel (column), register e = cout 32 ot + 8 (e >> 2) + 4 kh + (e & 3)
    float *part = A.partial + (size_t)blockIdx.x * 9 * Co * C;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        if (t < n_t) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int o = o0 + 32 * ot + 8 * (e >> 2) + 4 * kh + (e & 3);
                if (o < Co) part[((size_t)(t_lo + t) * Co + o) * C + c0 + l31] = acc[t][e];
            }
        }
    }
}

__global__ void dcn_bwd_weight_reduce_kernel(const float *__restrict__ partial, const float *__restrict__ g_amax, float *__restrict__ dw, int splits,
                                             int Co, int C)
{
    float oscale = 1.f;
    if (g_amax) {
        const float am = *g_amax;
        if (am > 1.0e-30f && am < 3.0e38f) {
            int e;
            (void)frexpf(am, &e);
            oscale = ldexpf(1.f, e - 14);
        }
    }
    const long total = (long)Co * C * 9;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int tap = (int)(i % 9);
        const long oc = i / 9;   // o * C + c
        float s = 0.f;
        for (int k = 0; k < splits; ++k) s += partial[((size_t)k * 9 + tap) * Co * C + oc];
        dw[i] = s * oscale;
    }
}


// ---- weight gradient of a 1x1 convolution, d W[o][i] = sum over pixels of g[pixel][o] * x[pixel][i] (conv_emb1, spatial_attn,
// feat_fusion of ref_mrapa_restoration_arch.py:271-304): the same pixel-K GEMM without a gather.  Block = 64 couts x 64 cins,
// wave = one 32 x 32 quarter; both operands transposed into LDS by threads that own (channel quad, pixel pair).
struct Wg1Args {
    const float *g, *x, *g_amax;
    float *partial;
    long P;          // pixels (all images)
    int Cout, Cin, ld_g, ld_x, splits;
    int *range_flag;
};
__global__ __launch_bounds__(256) void conv_wgrad1x1_kernel(const Wg1Args A)
{
    __shared__ __align__(16) unsigned char smem[2 * WG_BYTES];   // g^T | x^T, each [plane 2][k step 2][k half 2][channel 64][8 px]
    unsigned char *const sg = smem, *const sx = smem + WG_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, kh = lane >> 5;
    const int i0 = blockIdx.y * 64, o0 = blockIdx.z * 64;
    float gs = 1.f;
    if (A.g_amax) {
        const float am = *A.g_amax;
        if (am > 1.0e-30f && am < 3.0e38f) {
            int e;
            (void)frexpf(am, &e);
            gs = ldexpf(1.f, 14 - e);
        }
    }
    const int ot = wv & 1, it = wv >> 1;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int cq = tid & 15, pj = tid >> 4;
    const long tiles = (A.P + 31) / 32;
    float xmax = 0.f;
    auto load4 = [&](const float *row, const int c, const int Cn, const int ld) {   // 4 channels of a pixel; ragged / unaligned tails element-wise
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c + 3 < Cn && (ld & 3) == 0) {
            v = *reinterpret_cast<const float4 *>(row + c);
        } else {
            if (c < Cn) v.x = row[c];
            if (c + 1 < Cn) v.y = row[c + 1];
            if (c + 2 < Cn) v.z = row[c + 2];
            if (c + 3 < Cn) v.w = row[c + 3];
        }
        return v;
    };
    for (long kt = blockIdx.x; kt < tiles; kt += A.splits) {
        const long p0 = kt * 32;
        float4 gv[2], xv[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const long pp = p0 + 2 * pj + k;
            gv[k] = xv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pp < A.P) {
                gv[k] = load4(A.g + (size_t)pp * A.ld_g, o0 + 4 * cq, A.Cout, A.ld_g);
                xv[k] = load4(A.x + (size_t)pp * A.ld_x, i0 + 4 * cq, A.Cin, A.ld_x);
            }
        }
        const int px = 2 * pj;
        const int off = (((px >> 4) * 2 + ((px >> 3) & 1)) * 1024) + (px & 7) * 2;
        const float g0[4] = {gv[0].x * gs, gv[0].y * gs, gv[0].z * gs, gv[0].w * gs}, g1[4] = {gv[1].x * gs, gv[1].y * gs, gv[1].z * gs, gv[1].w * gs};
        const float x0[4] = {xv[0].x, xv[0].y, xv[0].z, xv[0].w}, x1[4] = {xv[1].x, xv[1].y, xv[1].z, xv[1].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned int h = pk_f16(g0[i], g1[i]);
            f32x2 f = un_f16(h);
            *reinterpret_cast<unsigned int *>(sg + off + (4 * cq + i) * 16) = h;
            *reinterpret_cast<unsigned int *>(sg + off + 4096 + (4 * cq + i) * 16) = pk_f16(g0[i] - f[0], g1[i] - f[1]);
            xmax = fmaxf(xmax, fmaxf(fabsf(x0[i]), fabsf(x1[i])));
            h = pk_f16(x0[i], x1[i]);
            f = un_f16(h);
            *reinterpret_cast<unsigned int *>(sx + off + (4 * cq + i) * 16) = h;
            *reinterpret_cast<unsigned int *>(sx + off + 4096 + (4 * cq + i) * 16) = pk_f16((x0[i] - f[0]) * 2048.f, (x1[i] - f[1]) * 2048.f);
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const unsigned char *ga = sg + ((ks * 2 + kh) * 1024) + (ot * 32 + l31) * 16, *xa = sx + ((ks * 2 + kh) * 1024) + (it * 32 + l31) * 16;
            const u32x4 gh = *reinterpret_cast<const u32x4 *>(ga), gl = *reinterpret_cast<const u32x4 *>(ga + 4096);
            const u32x4 xh = *reinterpret_cast<const u32x4 *>(xa), xl = *reinterpret_cast<const u32x4 *>(xa + 4096);
            acc = mma16(gl, xh, acc);
            acc = mma16(scale_wh(gh), xl, acc);
            acc = mma16(gh, xh, acc);
        }
        __syncthreads();
    }
    if (A.range_flag && !(xmax <= 65000.f)) atomicOr(A.range_flag, 1);
    float *part = A.partial + (size_t)blockIdx.x * A.Cout * A.Cin;   // [split][Cout][Cin]; lane = cin (column), register e = cout row
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int o = o0 + 32 * ot + 8 * (e >> 2) + 4 * kh + (e & 3), i = i0 + 32 * it + l31;
        if (o < A.Cout && i < A.Cin) part[(size_t)o * A.Cin + i] = acc[e];
    }
}

__global__ void conv_wgrad1x1_reduce_kernel(const float *__restrict__ partial, const float *__restrict__ g_amax, float *__restrict__ dw, int splits, long n)
{
    float oscale = 1.f;
    if (g_amax) {
        const float am = *g_amax;
        if (am > 1.0e-30f && am < 3.0e38f) {
            int e;
            (void)frexpf(am, &e);
            oscale = ldexpf(1.f, e - 14);
        }
    }
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < splits; ++k) s += partial[(size_t)k * n + i];
        dw[i] = s * oscale;
    }
}

}  // namespace

MREFSR_EXPORT int mrefsr_dcn_bwd_data_f32(const float *grad_out, const float *x, const float *offset, const float *mask, const void *packed_wT,
                                          float wscale, const float *g_amax, float *grad_x, float *grad_offset, float *grad_mask,
                                          const mrefsr_dcn_shape *s, mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(grad_out && x && offset && packed_wT && grad_offset && s, "dcn_bwd_data: null pointer");
    MREFSR_REQUIRE(s->kh == 3 && s->kw == 3 && s->stride_h == 1 && s->stride_w == 1 && s->pad_h == 1 && s->pad_w == 1 && s->dil_h == 1 && s->dil_w == 1 &&
                       s->groups == 1,
                   "dcn_bwd_data: 3x3, stride 1, pad 1, dilation 1, one weight group (the shapes of the restoration networks)");
    MREFSR_REQUIRE(s->B > 0 && s->H > 0 && s->W > 0 && s->dg > 0 && s->C % 32 == 0 && s->Co % 16 == 0 && s->C % s->dg == 0, "dcn_bwd_data: B=%d C=%d Co=%d dg=%d",
                   s->B, s->C, s->Co, s->dg);
    const int cpg = s->C / s->dg;
    MREFSR_REQUIRE(cpg == 8 || cpg == 16 || cpg == 32, "dcn_bwd_data: %d channels per deformable group (8, 16 or 32)", cpg);
    MREFSR_REQUIRE(wscale > 0.f && wscale < 3.0e38f, "dcn_bwd_data: the wscale the weights were packed with");
    BwdArgs a{};
    a.gy = grad_out, a.x = x, a.offset = offset, a.mask = mask, a.wp = reinterpret_cast<const unsigned short *>(packed_wT), a.g_amax = g_amax;
    a.gx = grad_x, a.goff = grad_offset, a.gmask = grad_mask;
    a.B = s->B, a.C = s->C, a.Co = s->Co, a.H = s->H, a.W = s->W, a.dg = s->dg, a.n_ch = s->Co / 16, a.inv_wscale = 1.0f / wscale;
    const long blocks = (long)s->B * ((s->H * s->W + 31) / 32);
    MREFSR_REQUIRE(blocks < ((long)1 << 31), "dcn_bwd_data: %ld blocks", blocks);
    const size_t lds = (size_t)2 * a.n_ch * 1024;
    hipStream_t st = (hipStream_t)stream;
    static unsigned long long attr = 0;
    if (mrefsr::first_use_on_device(attr)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(dcn_bwd_data_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(dcn_bwd_data_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(dcn_bwd_data_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    }
    MREFSR_REQUIRE(lds <= 64 * 1024, "dcn_bwd_data: Co=%d exceeds the tile buffer", s->Co);
    if (cpg == 8) hipLaunchKernelGGL(dcn_bwd_data_kernel<8>, dim3((unsigned)blocks), dim3(256), lds, st, a);
    else if (cpg == 16) hipLaunchKernelGGL(dcn_bwd_data_kernel<16>, dim3((unsigned)blocks), dim3(256), lds, st, a);
    else hipLaunchKernelGGL(dcn_bwd_data_kernel<32>, dim3((unsigned)blocks), dim3(256), lds, st, a);
    return mrefsr::check_launch("dcn_bwd_data");
}

MREFSR_EXPORT int64_t mrefsr_dcn_bwd_weight_workspace_bytes(const mrefsr_dcn_shape *s)
{
    if (!s || s->C <= 0 || s->Co <= 0) return -1;
    const long tiles = (long)s->B * ((s->H * s->W + 31) / 32);
    // ~1500 blocks per launch (three per CU: a block alternates gather and MFMA phases), every block at least 8 tiles, at most 512
    // partial sums per weight (the reduction reads splits x 36 Co C bytes: with 1024 splits of a 64 x 64 layer it cost more than the GEMM)
    long splits = 1536 / ((long)(s->C / 32) * ((s->Co + 63) / 64));
    splits = splits > tiles / 8 ? tiles / 8 : splits;
    splits = splits < 1 ? 1 : (splits > 512 ? 512 : splits);
    return (int64_t)splits * 9 * s->Co * s->C * 4;
}

MREFSR_EXPORT int mrefsr_dcn_bwd_weight_f32(const float *grad_out, const float *x, const float *offset, const float *mask, const float *g_amax,
                                            float *grad_weight, void *workspace, int64_t workspace_bytes, const mrefsr_dcn_shape *s,
                                            int *range_flag, mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(grad_out && x && offset && grad_weight && workspace && s, "dcn_bwd_weight: null pointer");
    MREFSR_REQUIRE(s->kh == 3 && s->kw == 3 && s->stride_h == 1 && s->stride_w == 1 && s->pad_h == 1 && s->pad_w == 1 && s->dil_h == 1 && s->dil_w == 1 &&
                       s->groups == 1,
                   "dcn_bwd_weight: 3x3, stride 1, pad 1, dilation 1, one weight group (the shapes of the restoration networks)");
    MREFSR_REQUIRE(s->B > 0 && s->H > 0 && s->W > 0 && s->dg > 0 && s->C % 32 == 0 && s->Co % 4 == 0 && s->C % s->dg == 0 && (s->C / s->dg) % 4 == 0,
                   "dcn_bwd_weight: B=%d C=%d Co=%d dg=%d", s->B, s->C, s->Co, s->dg);
    const int64_t need = mrefsr_dcn_bwd_weight_workspace_bytes(s);
    MREFSR_REQUIRE(workspace_bytes >= need, "dcn_bwd_weight: workspace of %ld bytes required (got %ld)", (long)need, (long)workspace_bytes);
    BwdWArgs a{};
    a.gy = grad_out, a.x = x, a.offset = offset, a.mask = mask, a.g_amax = g_amax, a.partial = reinterpret_cast<float *>(workspace);
    a.B = s->B, a.C = s->C, a.Co = s->Co, a.H = s->H, a.W = s->W, a.dg = s->dg, a.range_flag = range_flag;
    a.splits = (int)(need / ((int64_t)9 * s->Co * s->C * 4));
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(dcn_bwd_weight_kernel, dim3(a.splits, s->C / 32, (s->Co + 63) / 64), dim3(256), 0, st, a);
    if (int e = mrefsr::check_launch("dcn_bwd_weight")) return e;
    const long total = (long)s->Co * s->C * 9;
    hipLaunchKernelGGL(dcn_bwd_weight_reduce_kernel, dim3((unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048)), dim3(256), 0, st, a.partial, g_amax,
                       grad_weight, a.splits, s->Co, s->C);
    return mrefsr::check_launch("dcn_bwd_weight_reduce");
}

static long wgrad1x1_splits(long P, int Cout, int Cin)
{
    const long tiles = (P + 31) / 32;
    long splits = 512 / ((long)((Cin + 63) / 64) * ((Cout + 63) / 64));   // ~512 blocks per launch, at most 64 partial sums per weight
    splits = splits < 4 ? 4 : (splits > 64 ? 64 : splits);
    return splits > tiles ? tiles : splits;
}

MREFSR_EXPORT int64_t mrefsr_conv_wgrad1x1_workspace_bytes(int64_t pixels, int Cout, int Cin)
{
    if (pixels <= 0 || Cout <= 0 || Cin <= 0) return -1;
    return (int64_t)wgrad1x1_splits((long)pixels, Cout, Cin) * Cout * Cin * 4;
}

MREFSR_EXPORT int mrefsr_conv_wgrad1x1_f32(const float *g, const float *x, const float *g_amax, float *grad_weight, void *workspace,
                                           int64_t workspace_bytes, int64_t pixels, int Cout, int ld_g, int Cin, int ld_x, int *range_flag,
                                           mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(g && x && grad_weight && workspace, "conv_wgrad1x1: null pointer");
    MREFSR_REQUIRE(pixels > 0 && Cout > 0 && Cin > 0 && ld_g >= Cout && ld_x >= Cin, "conv_wgrad1x1: pixels=%ld Cout=%d ld_g=%d Cin=%d ld_x=%d", (long)pixels,
                   Cout, ld_g, Cin, ld_x);
    const int64_t need = mrefsr_conv_wgrad1x1_workspace_bytes(pixels, Cout, Cin);
    MREFSR_REQUIRE(workspace_bytes >= need, "conv_wgrad1x1: workspace of %ld bytes required (got %ld)", (long)need, (long)workspace_bytes);
    Wg1Args a{};
    a.g = g, a.x = x, a.g_amax = g_amax, a.partial = reinterpret_cast<float *>(workspace), a.P = (long)pixels;
    a.Cout = Cout, a.Cin = Cin, a.ld_g = ld_g, a.ld_x = ld_x, a.range_flag = range_flag;
    a.splits = (int)wgrad1x1_splits((long)pixels, Cout, Cin);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(conv_wgrad1x1_kernel, dim3(a.splits, (Cin + 63) / 64, (Cout + 63) / 64), dim3(256), 0, st, a);
    if (int e = mrefsr::check_launch("conv_wgrad1x1")) return e;
    const long n = (long)Cout * Cin;
    hipLaunchKernelGGL(conv_wgrad1x1_reduce_kernel, dim3((unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024)), dim3(256), 0, st, a.partial, g_amax, grad_weight,
                       a.splits, n);
    return mrefsr::check_launch("conv_wgrad1x1_reduce");
}
