// 3x3 / stride 1 / pad 1 convolution of the restoration and VGG trunks (arch_util.py ResidualBlockNoBN,
// ref_mrapa_restoration_arch.py:139-259, vgg_arch.py) as an implicit GEMM on the bf16 matrix pipe with
// fp32-equivalent arithmetic:
//
//   every fp32 operand is split exactly into three bf16 terms  v = hi + mid + lo  (8+8+8 mantissa
//   bits, round-to-nearest-even at each step, remainders exact in fp32) and a product a*b is
//   evaluated as the six partial products whose weight is >= 2^-24 relative:
//       hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid
//   each accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The dropped terms (mid*lo, lo*mid,
//   lo*lo) are < 2^-23 relative, i.e. below fp32 rounding of the product itself, so the result
//   is as accurate as an fp32 FMA chain -- while the bf16 pipe is 16x faster than the fp32 MFMA
//   (6 instructions per product: 2.7x the fp32 matrix peak).
//
// Data layout: activations NHWC fp32 (channels-last), weights pre-packed once per layer into the
// B-fragment order  [cout block of 64][cin chunk of 16][tap 9][split 3][cout 64][cin 16]  bf16.
// One block = 256 threads = 4 waves computes 16 rows x 32 cols x 64 couts; per cin chunk the
// (18 x 34)-pixel halo tile is split into LDS once ([split][pixel][16 ch] bf16, 58.75 KB), then each
// wave runs 9 taps x (4 row tiles x 2 cout tiles x 6 terms) MFMAs reading A fragments straight
// from LDS (conflict-free: a pixel's 16 channels are 32 contiguous bytes) and B fragments from
// the packed weights (L1/L2 resident, 1 KB contiguous per fragment).
// Epilogue fused: + bias, LeakyReLU(slope) / ReLU, + residual, NHWC store (128-byte segments).
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int TH = 16, TW = 32, PH = TH + 2, PW = TW + 2, NPIX = PH * PW, KC = 16, NB = 64;

__device__ __forceinline__ bf16x8 as_bf(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// packed round-to-nearest-even bf16 of two floats, and back
__device__ __forceinline__ unsigned int pk_bf16(float a, float b)
{
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ float bf_lo(unsigned int p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(unsigned int p) { return __uint_as_float(p & 0xffff0000u); }

// 4 floats -> NS planes of 4 bf16 (8 bytes each)
template <int NS>
__device__ __forceinline__ void split4(const float4 v, u32x2 *out)
{
    float a = v.x, b = v.y, c = v.z, d = v.w;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const unsigned int p0 = pk_bf16(a, b), p1 = pk_bf16(c, d);
        out[s] = u32x2{p0, p1};
        if (s + 1 < NS) {
            a -= bf_lo(p0);
            b -= bf_hi(p0);
            c -= bf_lo(p1);
            d -= bf_hi(p1);
        }
    }
}

// OIHW fp32 -> [cout block][cin chunk][tap][split][64 cout][16 cin] bf16 (zero padded)
template <int NS>
__global__ void conv3x3_pack_kernel(const float *__restrict__ w, unsigned short *__restrict__ wp, int Cout, int Cin,
                                    int n_cb, int n_ch)
{
    const long total = (long)n_cb * n_ch * 9 * NB * KC;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(e % KC);
        long t = e / KC;
        const int co = (int)(t % NB);
        t /= NB;
        const int tap = (int)(t % 9);
        t /= 9;
        const int ch = (int)(t % n_ch), cb = (int)(t / n_ch);
        const int o = cb * NB + co, i = ch * KC + ci;
        float v = (o < Cout && i < Cin) ? w[((size_t)o * Cin + i) * 9 + tap] : 0.f;
        const size_t base = ((((size_t)cb * n_ch + ch) * 9 + tap) * NS) * NB * KC + (size_t)co * KC + ci;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const unsigned int p = pk_bf16(v, 0.f);
            wp[base + (size_t)s * NB * KC] = (unsigned short)(p & 0xffffu);
            v -= bf_lo(p);
        }
    }
}

template <int NS>
__global__ __launch_bounds__(256, 2) void conv3x3_nhwc_kernel(const float *__restrict__ x, const unsigned short *__restrict__ wp,
                                                              const float *__restrict__ bias, const float *__restrict__ residual,
                                                              float *__restrict__ out, int H, int W, int Cin, int Cout,
                                                              int n_ch, int n_cb, float slope, int act)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int cb = blockIdx.z % n_cb, n = blockIdx.z / n_cb;
    const int y0 = blockIdx.y * TH, x0 = blockIdx.x * TW;
    constexpr int PLANE = NPIX * KC * 2;  // bytes per split plane

    f32x16 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][j][e] = 0.f;

    const float *xn = x + (size_t)n * H * W * Cin;
    const unsigned short *wcb = wp + (size_t)cb * n_ch * 9 * NS * NB * KC;

    for (int ch = 0; ch < n_ch; ++ch) {
        if (ch) __syncthreads();
        // ---- halo tile of this 16-channel chunk -> NS bf16 planes in LDS
        for (int i = tid; i < NPIX * 4; i += 256) {
            const int p = i >> 2, q = i & 3;
            const int py = p / PW, px = p - py * PW;
            const int gy = y0 + py - 1, gx = x0 + px - 1, c = ch * KC + 4 * q;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy >= 0 && gy < H && gx >= 0 && gx < W && c < Cin)
                v = *reinterpret_cast<const float4 *>(xn + ((size_t)gy * W + gx) * Cin + c);
            u32x2 sp[NS];
            split4<NS>(v, sp);
#pragma unroll
            for (int s = 0; s < NS; ++s) *reinterpret_cast<u32x2 *>(smem + s * PLANE + p * (KC * 2) + q * 8) = sp[s];
        }
        __syncthreads();
        const unsigned short *wch = wcb + (size_t)ch * 9 * NS * NB * KC + (size_t)l31 * KC + kh * 8;
        u32x4 b[2][NS];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int s = 0; s < NS; ++s) b[j][s] = *reinterpret_cast<const u32x4 *>(wch + ((size_t)s * NB + j * 32) * KC);
#pragma nounroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = (tap * 11) >> 5, dx = tap - 3 * dy;
            // B fragments of the next tap travel while this tap's MFMAs run
            u32x4 bn[2][NS];
            const int tn = tap < 8 ? tap + 1 : 8;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    bn[j][s] = *reinterpret_cast<const u32x4 *>(wch + ((size_t)(tn * NS + s) * NB + j * 32) * KC);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int p = (wv * 4 + m + dy) * PW + l31 + dx;
                u32x4 a[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s) a[s] = *reinterpret_cast<const u32x4 *>(smem + s * PLANE + p * (KC * 2) + kh * 16);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    // smallest partial products first
                    if (NS >= 3) {
                        acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a[1]), as_bf(b[j][1]), acc[m][j], 0, 0, 0);
                        acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a[2]), as_bf(b[j][0]), acc[m][j], 0, 0, 0);
                        acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a[0]), as_bf(b[j][2]), acc[m][j], 0, 0, 0);
                    }
                    if (NS >= 2) {
                        acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a[1]), as_bf(b[j][0]), acc[m][j], 0, 0, 0);
                        acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a[0]), as_bf(b[j][1]), acc[m][j], 0, 0, 0);
                    }
                    acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a[0]), as_bf(b[j][0]), acc[m][j], 0, 0, 0);
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int s = 0; s < NS; ++s) b[j][s] = bn[j][s];
        }
    }

    // ---- epilogue: lane holds cout (j*32 + l31) for pixels x = (e&3) + 8*(e>>2) + 4*kh of row m
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int co = cb * NB + j * 32 + l31;
        if (co >= Cout) continue;
        const float bv = bias ? bias[co] : 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int gy = y0 + wv * 4 + m;
            if (gy >= H) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int gx = x0 + (e & 3) + 8 * (e >> 2) + 4 * kh;
                if (gx >= W) continue;
                const size_t o = (((size_t)n * H + gy) * W + gx) * Cout + co;
                float v = acc[m][j][e] + bv;
                if (act) v = v > 0.f ? v : v * slope;
                if (residual) v += residual[o];
                out[o] = v;
            }
        }
    }
}

template <int NS>
int launch(const float *x, const void *wp, const float *bias, const float *residual, float *out, int N, int H, int W, int Cin,
           int Cout, float slope, int act, hipStream_t stream)
{
    const int n_ch = (Cin + KC - 1) / KC, n_cb = (Cout + NB - 1) / NB;
    const size_t lds = (size_t)NS * NPIX * KC * 2;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_nhwc_kernel<NS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
        attr_done = true;
    }
    dim3 grid((W + TW - 1) / TW, (H + TH - 1) / TH, N * n_cb);
    hipLaunchKernelGGL(conv3x3_nhwc_kernel<NS>, grid, dim3(256), lds, stream, x, reinterpret_cast<const unsigned short *>(wp), bias,
                       residual, out, H, W, Cin, Cout, n_ch, n_cb, slope, act);
    return mrefsr::check_launch("conv3x3_nhwc");
}

}  // namespace

MREFSR_EXPORT int64_t mrefsr_conv3x3_packed_bytes(int Cout, int Cin, int terms)
{
    const int ns = terms == 3 ? 2 : 3;
    const long n_ch = (Cin + KC - 1) / KC, n_cb = (Cout + NB - 1) / NB;
    return n_cb * n_ch * 9 * ns * NB * KC * 2;
}

MREFSR_EXPORT int mrefsr_conv3x3_pack_weight_f32(const float *weight, void *packed, int Cout, int Cin, int terms,
                                                 mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(weight && packed, "conv3x3_pack_weight: null pointer");
    MREFSR_REQUIRE(Cout > 0 && Cin > 0 && (terms == 6 || terms == 3), "conv3x3_pack_weight: Cout=%d Cin=%d terms=%d", Cout, Cin,
                   terms);
    const int n_ch = (Cin + KC - 1) / KC, n_cb = (Cout + NB - 1) / NB;
    const long total = (long)n_cb * n_ch * 9 * NB * KC;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (terms == 6)
        hipLaunchKernelGGL(conv3x3_pack_kernel<3>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, weight,
                           reinterpret_cast<unsigned short *>(packed), Cout, Cin, n_cb, n_ch);
    else
        hipLaunchKernelGGL(conv3x3_pack_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, weight,
                           reinterpret_cast<unsigned short *>(packed), Cout, Cin, n_cb, n_ch);
    return mrefsr::check_launch("conv3x3_pack_weight");
}

MREFSR_EXPORT int mrefsr_conv3x3_nhwc_f32(const float *x, const void *packed, const float *bias, const float *residual, float *out,
                                          int N, int H, int W, int Cin, int Cout, int terms, int act, float slope,
                                          mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(x && packed && out, "conv3x3_nhwc: null pointer");
    MREFSR_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv3x3_nhwc: N=%d H=%d W=%d Cin=%d Cout=%d", N, H, W, Cin, Cout);
    MREFSR_REQUIRE(Cin % 4 == 0, "conv3x3_nhwc: Cin=%d must be a multiple of 4", Cin);
    MREFSR_REQUIRE(terms == 6 || terms == 3, "conv3x3_nhwc: terms=%d (6 or 3)", terms);
    MREFSR_REQUIRE((long)N * ((Cout + NB - 1) / NB) <= 65535, "conv3x3_nhwc: N*cout blocks too large");
    if (terms == 6) return launch<3>(x, packed, bias, residual, out, N, H, W, Cin, Cout, slope, act, (hipStream_t)stream);
    return launch<2>(x, packed, bias, residual, out, N, H, W, Cin, Cout, slope, act, (hipStream_t)stream);
}
