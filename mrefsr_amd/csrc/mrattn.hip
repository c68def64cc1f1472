// Multi-reference feature-transfer attention core (ref_mrapa_restoration_arch.py:321-335).
//
// The reference materialises three permuted copies ((n*h*w,1,c), (n*h*w,c,t), (n*h*w,t,2c)) and
// runs two batched matmuls with inner dims 1 x c x t.  Here one thread owns one pixel and reads
// the NCHW tensors in place: every load is coalesced along W, the T logits and the softmax live
// in registers, nothing is permuted.  HBM-bound: (c + T*c + T*c2 + c2) * 4 bytes per pixel.
#include "common.h"

namespace {

constexpr int MAX_T = 16;

__global__ __launch_bounds__(256) void mrattn_fwd_kernel(const float *__restrict__ q, const float *__restrict__ emb,
                                                         const float *__restrict__ ass, float *__restrict__ out,
                                                         float *__restrict__ prob, int N, int T, int c, int c2, int HW, int t_major)
{
    const long total = (long)N * HW;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int n = (int)(e / HW), p = (int)(e - (long)n * HW);
        float s[MAX_T];
#pragma unroll
        for (int t = 0; t < MAX_T; ++t) s[t] = 0.f;
        // image index of reference t of sample n: n*T + t (refs stacked [N][T], the reference's
        // torch.stack(refs, dim=1).flatten(0, 1)) or t*N + n (refs stacked [T][N], the batched path)
        const size_t i0 = t_major ? (size_t)n : (size_t)n * T, is = t_major ? (size_t)N : 1;
        const float *qp = q + (size_t)n * c * HW + p;
        const float *ep = emb + i0 * c * HW + p;
        const size_t est = is * c * HW;
        for (int k = 0; k < c; ++k) {
            const float qv = qp[(size_t)k * HW];
#pragma unroll
            for (int t = 0; t < MAX_T; ++t)
                if (t < T) s[t] = fmaf(qv, ep[t * est + (size_t)k * HW], s[t]);
        }
        float mx = s[0];
#pragma unroll
        for (int t = 1; t < MAX_T; ++t)
            if (t < T) mx = fmaxf(mx, s[t]);
        float den = 0.f;
#pragma unroll
        for (int t = 0; t < MAX_T; ++t)
            if (t < T) { s[t] = expf(s[t] - mx); den += s[t]; }
        const float rden = 1.0f / den;
#pragma unroll
        for (int t = 0; t < MAX_T; ++t)
            if (t < T) {
                s[t] *= rden;
                if (prob) prob[((size_t)n * T + t) * HW + p] = s[t];
            }
        const float *ap = ass + i0 * c2 * HW + p;
        const size_t ast = is * c2 * HW;
        float *op = out + (size_t)n * c2 * HW + p;
        for (int k = 0; k < c2; ++k) {
            float a = 0.f;
#pragma unroll
            for (int t = 0; t < MAX_T; ++t)
                if (t < T) a = fmaf(s[t], ap[t * ast + (size_t)k * HW], a);
            op[(size_t)k * HW] = a;
        }
    }
}

__global__ __launch_bounds__(256) void mrattn_bwd_kernel(const float *__restrict__ q, const float *__restrict__ emb,
                                                         const float *__restrict__ ass, const float *__restrict__ prob,
                                                         const float *__restrict__ g_out, float *__restrict__ g_q,
                                                         float *__restrict__ g_emb, float *__restrict__ g_ass, int N,
                                                         int T, int c, int c2, int HW, int t_major)
{
    const long total = (long)N * HW;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int n = (int)(e / HW), p = (int)(e - (long)n * HW);
        const size_t i0 = t_major ? (size_t)n : (size_t)n * T, is = t_major ? (size_t)N : 1;
        float a[MAX_T], da[MAX_T];
#pragma unroll
        for (int t = 0; t < MAX_T; ++t) {
            a[t] = (t < T) ? prob[((size_t)n * T + t) * HW + p] : 0.f;
            da[t] = 0.f;
        }
        const float *ap = ass + i0 * c2 * HW + p;
        float *gap = g_ass + i0 * c2 * HW + p;
        const size_t ast = is * c2 * HW, est = is * c * HW;
        const float *gp = g_out + (size_t)n * c2 * HW + p;
        for (int k = 0; k < c2; ++k) {
            const float g = gp[(size_t)k * HW];
#pragma unroll
            for (int t = 0; t < MAX_T; ++t)
                if (t < T) {
                    const size_t o = t * ast + (size_t)k * HW;
                    da[t] = fmaf(g, ap[o], da[t]);
                    gap[o] = g * a[t];
                }
        }
        float dot = 0.f;
#pragma unroll
        for (int t = 0; t < MAX_T; ++t)
            if (t < T) dot = fmaf(a[t], da[t], dot);
#pragma unroll
        for (int t = 0; t < MAX_T; ++t) da[t] = a[t] * (da[t] - dot);  // d logits
        const float *qp = q + (size_t)n * c * HW + p;
        const float *ep = emb + i0 * c * HW + p;
        float *gqp = g_q + (size_t)n * c * HW + p;
        float *gep = g_emb + i0 * c * HW + p;
        for (int k = 0; k < c; ++k) {
            const float qv = qp[(size_t)k * HW];
            float acc = 0.f;
#pragma unroll
            for (int t = 0; t < MAX_T; ++t)
                if (t < T) {
                    const size_t o = t * est + (size_t)k * HW;
                    acc = fmaf(da[t], ep[o], acc);
                    gep[o] = da[t] * qv;
                }
            gqp[(size_t)k * HW] = acc;
        }
    }
}

int check(const char *who, int N, int T, int c, int c2, int HW)
{
    if (N <= 0 || T <= 0 || c <= 0 || c2 <= 0 || HW <= 0)
        return mrefsr::fail(MREFSR_E_INVALID, "%s: N=%d T=%d c=%d c2=%d HW=%d", who, N, T, c, c2, HW);
    if (T > MAX_T) return mrefsr::fail(MREFSR_E_UNSUPPORTED, "%s: T=%d references > %d", who, T, MAX_T);
    return 0;
}

}  // namespace

namespace {

// channels-last forward (inference path): q [N][HW][c], emb [T*N][HW][c], ass [T*N][HW][2c] (t-major),
// out [N][HW][2c].  c/4 lanes share a pixel: 16-byte channel vectors, the T dot products reduced with
// xor shuffles inside the lane group, softmax in registers, the assembly as two 16-byte streams per lane.
// IO16: q / emb / ass / out are bf16 tensors (2-byte storage of BASELINE configs[4]; same lanes, same operation order, the
// result rounded to bf16 on the store).
template <bool IO16> struct Io4;
template <> struct Io4<false> {
    typedef float T;
    static __device__ __forceinline__ float4 ld(const float *p) { return *reinterpret_cast<const float4 *>(p); }
    static __device__ __forceinline__ void st(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
};
template <> struct Io4<true> {
    typedef unsigned short T;
    static __device__ __forceinline__ float4 ld(const unsigned short *p)
    {
        const uint2 r = *reinterpret_cast<const uint2 *>(p);
        return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                           __uint_as_float(r.y & 0xffff0000u));
    }
    static __device__ __forceinline__ unsigned int rne(float a, float b)
    {
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2{a, b}, bf16x2));
    }
    static __device__ __forceinline__ void st(unsigned short *p, float4 v) { *reinterpret_cast<uint2 *>(p) = make_uint2(rne(v.x, v.y), rne(v.z, v.w)); }
};

template <int CH, bool IO16>
__global__ __launch_bounds__(256) void mrattn_fwd_nhwc_kernel(const typename Io4<IO16>::T *__restrict__ q, const typename Io4<IO16>::T *__restrict__ emb,
                                                              const typename Io4<IO16>::T *__restrict__ ass, typename Io4<IO16>::T *__restrict__ out,
                                                              int N, int T, long HW, float q_scale)
{
    typedef Io4<IO16> IO;
    constexpr int L = CH / 4, PPW = 64 / L;
    const int lane = threadIdx.x & 63, sub = lane % L, pw = lane / L;
    const long total = (long)N * HW;
    const long wave = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6, nwave = ((long)gridDim.x * blockDim.x) >> 6;
    for (long base = wave * PPW; base < total; base += nwave * PPW) {
        const long gp = base + pw;
        const bool ok = gp < total;
        const long g = ok ? gp : total - 1;
        const long n = g / HW, p = g - n * HW;
        float4 qv = IO::ld(q + g * CH + 4 * sub);
        if (!IO16) qv.x *= q_scale, qv.y *= q_scale, qv.z *= q_scale, qv.w *= q_scale;   // (q * scale of ref :323 as its own rounded product: the bits of the separate pass; 1 = none)
        float logit[16];
        float mx = -3.4e38f;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            if (t < T) {
                const float4 e = IO::ld(emb + (((long)t * N + n) * HW + p) * CH + 4 * sub);
                float d = qv.x * e.x + qv.y * e.y + qv.z * e.z + qv.w * e.w;
#pragma unroll
                for (int o = L / 2; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
                logit[t] = d;
                mx = fmaxf(mx, d);
            }
        }
        float den = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t)
            if (t < T) {
                logit[t] = expf(logit[t] - mx);
                den += logit[t];
            }
        const float inv = 1.0f / den;
        float4 o0 = make_float4(0.f, 0.f, 0.f, 0.f), o1 = o0;
#pragma unroll
        for (int t = 0; t < 16; ++t)
            if (t < T) {
                const float w = logit[t] * inv;
                const typename IO::T *ap = ass + (((long)t * N + n) * HW + p) * (2 * CH) + 4 * sub;
                const float4 a0 = IO::ld(ap), a1 = IO::ld(ap + CH);
                o0.x += w * a0.x, o0.y += w * a0.y, o0.z += w * a0.z, o0.w += w * a0.w;
                o1.x += w * a1.x, o1.y += w * a1.y, o1.z += w * a1.z, o1.w += w * a1.w;
            }
        if (ok) {
            typename IO::T *op = out + g * (2 * CH) + 4 * sub;
            IO::st(op, o0);
            IO::st(op + CH, o1);
        }
    }
}

template <bool IO16>
int launch_mrattn_nhwc(const void *q, const void *emb, const void *ass, void *out, int N, int T, int c, int HW, float q_scale, hipStream_t st)
{
    typedef typename Io4<IO16>::T E;
    const long waves = ((long)N * HW * (c / 4) + 63) / 64;
    const long blocks = (waves + 3) / 4;
    const dim3 grid((int)(blocks < 65536 ? blocks : 65536));
    const E *q_ = (const E *)q, *e_ = (const E *)emb, *a_ = (const E *)ass;
    E *o_ = (E *)out;
    if (c == 256) hipLaunchKernelGGL((mrattn_fwd_nhwc_kernel<256, IO16>), grid, dim3(256), 0, st, q_, e_, a_, o_, N, T, (long)HW, q_scale);
    else if (c == 128) hipLaunchKernelGGL((mrattn_fwd_nhwc_kernel<128, IO16>), grid, dim3(256), 0, st, q_, e_, a_, o_, N, T, (long)HW, q_scale);
    else if (c == 64) hipLaunchKernelGGL((mrattn_fwd_nhwc_kernel<64, IO16>), grid, dim3(256), 0, st, q_, e_, a_, o_, N, T, (long)HW, q_scale);
    else return mrefsr::fail(MREFSR_E_UNSUPPORTED, "mrattn_fwd_nhwc: c=%d (64, 128 or 256: the three MRAPAFusion heads)", c);
    return mrefsr::check_launch("mrattn_fwd_nhwc");
}

}  // namespace

MREFSR_EXPORT int mrefsr_mrattn_fwd_nhwc_f32(const float *q, const float *emb, const float *ass, float *out, int N, int T, int c,
                                             int HW, mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(q && emb && ass && out, "mrattn_fwd_nhwc: null pointer");
    MREFSR_REQUIRE(N > 0 && T > 0 && T <= 16 && HW > 0, "mrattn_fwd_nhwc: N=%d T=%d HW=%d (T <= 16)", N, T, HW);
    return launch_mrattn_nhwc<false>(q, emb, ass, out, N, T, c, HW, 1.0f, (hipStream_t)stream);
}

// the same with `q * q_scale` (the 1 / sqrt(c) of ref_mrapa_restoration_arch.py:323) formed on the way in -- each product rounded on its
// own, i.e. the bits of a separate element-wise pass over q, without the pass
MREFSR_EXPORT int mrefsr_mrattn_fwd_nhwc_scaled_f32(const float *q, const float *emb, const float *ass, float *out, int N, int T, int c,
                                                    int HW, float q_scale, mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(q && emb && ass && out, "mrattn_fwd_nhwc_scaled: null pointer");
    MREFSR_REQUIRE(N > 0 && T > 0 && T <= 16 && HW > 0, "mrattn_fwd_nhwc_scaled: N=%d T=%d HW=%d (T <= 16)", N, T, HW);
    return launch_mrattn_nhwc<false>(q, emb, ass, out, N, T, c, HW, q_scale, (hipStream_t)stream);
}

MREFSR_EXPORT int mrefsr_mrattn_fwd_nhwc_bf16(const void *q, const void *emb, const void *ass, void *out, int N, int T, int c, int HW,
                                              mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(q && emb && ass && out, "mrattn_fwd_nhwc_bf16: null pointer");
    MREFSR_REQUIRE(N > 0 && T > 0 && T <= 16 && HW > 0, "mrattn_fwd_nhwc_bf16: N=%d T=%d HW=%d (T <= 16)", N, T, HW);
    return launch_mrattn_nhwc<true>(q, emb, ass, out, N, T, c, HW, 1.0f, (hipStream_t)stream);
}

MREFSR_EXPORT int mrefsr_mrattn_fwd_f32(const float *q, const float *emb, const float *ass, float *out, float *prob,
                                        int N, int T, int c, int c2, int HW, int t_major, mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(q && emb && ass && out, "mrattn_fwd: null pointer");
    if (int e = check("mrattn_fwd", N, T, c, c2, HW)) return e;
    const long blocks = ((long)N * HW + 255) / 256;
    hipLaunchKernelGGL(mrattn_fwd_kernel, dim3((int)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, (hipStream_t)stream,
                       q, emb, ass, out, prob, N, T, c, c2, HW, t_major);
    return mrefsr::check_launch("mrattn_fwd");
}

MREFSR_EXPORT int mrefsr_mrattn_bwd_f32(const float *q, const float *emb, const float *ass, const float *prob,
                                        const float *g_out, float *g_q, float *g_emb, float *g_ass, int N, int T, int c,
                                        int c2, int HW, int t_major, mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(q && emb && ass && prob && g_out && g_q && g_emb && g_ass, "mrattn_bwd: null pointer");
    if (int e = check("mrattn_bwd", N, T, c, c2, HW)) return e;
    const long blocks = ((long)N * HW + 255) / 256;
    hipLaunchKernelGGL(mrattn_bwd_kernel, dim3((int)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, (hipStream_t)stream,
                       q, emb, ass, prob, g_out, g_q, g_emb, g_ass, N, T, c, c2, HW, t_major);
    return mrefsr::check_launch("mrattn_bwd");
}
