// basicsr.ops.fused_act: y = act(x + bias[channel]) * scale, forward / backward / double-backward
// selected by act*10+grad (fused_bias_act_kernel.cu:19-50 of the reference).  One read + one
// write per element: HBM-bound.  fp32 / fp16 / bf16 storage, fp32 math.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include "common.h"

namespace {

template <typename T> struct Acc { typedef float type; };
template <> struct Acc<double> { typedef double type; };
template <typename T> __device__ __forceinline__ typename Acc<T>::type ld(const T *p, long i) { return (typename Acc<T>::type)p[i]; }
template <> __device__ __forceinline__ float ld<__half>(const __half *p, long i) { return __half2float(p[i]); }
template <> __device__ __forceinline__ float ld<__hip_bfloat16>(const __hip_bfloat16 *p, long i) { return __bfloat162float(p[i]); }
template <typename T, typename A> __device__ __forceinline__ void st(T *p, long i, A v) { p[i] = (T)v; }
template <> __device__ __forceinline__ void st<__half, float>(__half *p, long i, float v) { p[i] = __float2half(v); }
template <> __device__ __forceinline__ void st<__hip_bfloat16, float>(__hip_bfloat16 *p, long i, float v) { p[i] = __float2bfloat16(v); }

template <typename T>
__global__ __launch_bounds__(256) void fused_bias_act_kernel(const T *__restrict__ x, const T *__restrict__ b,
                                                             const T *__restrict__ ref, T *__restrict__ out, long size_x,
                                                             int step_b, int size_b, int mode, float alpha, float scale)
{
    typedef typename Acc<T>::type A;   // float math, double for T = double (the reference computes in scalar_t)
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < size_x; i += (long)gridDim.x * blockDim.x) {
        A v = ld(x, i);
        if (b) v += ld(b, (i / step_b) % size_b);
        const A r = ref ? ld(ref, i) : (A)0;
        A y;
        switch (mode) {
        default:
        case 10: case 11: y = v; break;
        case 12: case 32: y = 0; break;
        case 30: y = (v > 0) ? v : v * (A)alpha; break;
        case 31: y = (r > 0) ? v : v * (A)alpha; break;
        }
        st(out, i, y * (A)scale);
    }
}

// 2-byte fast path (f16 / bf16): 16 B = 8 elements per lane when the bias index is constant over them
template <typename T>
__global__ __launch_bounds__(256) void fused_bias_act_x8_kernel(const uint4 *__restrict__ x, const T *__restrict__ b, const uint4 *__restrict__ ref,
                                                                uint4 *__restrict__ out, long n8, int step_b, int size_b, int mode, float alpha,
                                                                float scale)
{
    // With a bias the grid is (pieces of a plane, planes): the channel of a plane is blockIdx.y-derived -- no 64-bit division per
    // 16-byte piece (it cost the forward a third of its bandwidth: 4.75 TB/s against the bias-free backward's 7.2).
    // launch(): step_b is a multiple of 8 there, pv = pieces per plane; without a bias pv = n8 and the grid is one row.
    const long pv = b ? step_b / 8 : n8;
    for (long pl = blockIdx.y; pl * pv < n8; pl += gridDim.y)
    for (long j = blockIdx.x * (long)blockDim.x + threadIdx.x; j < pv; j += (long)gridDim.x * blockDim.x) {
        const long i = pl * pv + j;
        typedef unsigned int u32x4n __attribute__((ext_vector_type(4)));
        union { u32x4n u; T e[8]; } xv, rv, ov;
        xv.u = __builtin_nontemporal_load(reinterpret_cast<const u32x4n *>(x) + i);
        if (ref) rv.u = __builtin_nontemporal_load(reinterpret_cast<const u32x4n *>(ref) + i);
        const float bv = b ? ld(b, (long)(pl % size_b)) : 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float v = ld(xv.e, k) + bv;
            const float r = ref ? ld(rv.e, k) : 0.f;
            float y;
            switch (mode) {
            default:
            case 10: case 11: y = v; break;
            case 12: case 32: y = 0.f; break;
            case 30: y = (v > 0.f) ? v : v * alpha; break;
            case 31: y = (r > 0.f) ? v : v * alpha; break;
            }
            st(ov.e, k, y * scale);
        }
        __builtin_nontemporal_store(ov.u, reinterpret_cast<u32x4n *>(out) + i);
    }
}

// fp32 fast path: 16 B per lane when the bias index is constant over the 4 elements
__global__ __launch_bounds__(256) void fused_bias_act_f32x4_kernel(const float4 *__restrict__ x, const float *__restrict__ b,
                                                                   const float4 *__restrict__ ref, float4 *__restrict__ out,
                                                                   long n4, int step_b, int size_b, int mode, float alpha,
                                                                   float scale)
{
    typedef float f32x4n __attribute__((ext_vector_type(4)));
    const long pv = b ? step_b / 4 : n4;   // (plane, piece) decomposition as in fused_bias_act_x8_kernel
    for (long pl = blockIdx.y; pl * pv < n4; pl += gridDim.y)
    for (long j = blockIdx.x * (long)blockDim.x + threadIdx.x; j < pv; j += (long)gridDim.x * blockDim.x) {
        const long i = pl * pv + j;
        const f32x4n xv = __builtin_nontemporal_load(reinterpret_cast<const f32x4n *>(x) + i);
        float4 v = make_float4(xv[0], xv[1], xv[2], xv[3]);
        if (b) {
            const float bv = b[pl % size_b];
            v.x += bv; v.y += bv; v.z += bv; v.w += bv;
        }
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ref) {
            const f32x4n rv = __builtin_nontemporal_load(reinterpret_cast<const f32x4n *>(ref) + i);
            r = make_float4(rv[0], rv[1], rv[2], rv[3]);
        }
        float vv[4] = {v.x, v.y, v.z, v.w}, rr[4] = {r.x, r.y, r.z, r.w}, yy[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float y;
            switch (mode) {
            default:
            case 10: case 11: y = vv[k]; break;
            case 12: case 32: y = 0.f; break;
            case 30: y = (vv[k] > 0.f) ? vv[k] : vv[k] * alpha; break;
            case 31: y = (rr[k] > 0.f) ? vv[k] : vv[k] * alpha; break;
            }
            yy[k] = y * scale;
        }
        __builtin_nontemporal_store(f32x4n{yy[0], yy[1], yy[2], yy[3]}, reinterpret_cast<f32x4n *>(out) + i);
    }
}

// conv epilogue for the NCHW fp32 path: out = lrelu(x + bias[c], slope) (+ residual), one pass.
// (PyTorch-ROCm runs MIOpen convolutions without bias and then launches a broadcast add and an
// activation as two more full passes over the tensor; this is one.)  slope 1 = identity, 0 = ReLU.
__global__ __launch_bounds__(256) void bias_act_res_v4_kernel(const float4 *__restrict__ x, const float *__restrict__ bias,
                                                              const float4 *__restrict__ pre, long pre_n4,
                                                              const float4 *__restrict__ res, float4 *__restrict__ out,
                                                              long n4, int hw4, int C, float slope)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 v = x[i];
        const float b = bias ? bias[(i / hw4) % C] : 0.f;
        v.x += b; v.y += b; v.z += b; v.w += b;
        if (pre) {
            const float4 q = pre[i % pre_n4];
            v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
        }
        v.x = v.x > 0.f ? v.x : v.x * slope;
        v.y = v.y > 0.f ? v.y : v.y * slope;
        v.z = v.z > 0.f ? v.z : v.z * slope;
        v.w = v.w > 0.f ? v.w : v.w * slope;
        if (res) {
            const float4 r = res[i];
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        out[i] = v;
    }
}

__global__ __launch_bounds__(256) void bias_act_res_kernel(const float *__restrict__ x, const float *__restrict__ bias,
                                                           const float *__restrict__ pre, long pre_n,
                                                           const float *__restrict__ res, float *__restrict__ out, long n,
                                                           long hw, int C, float slope)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float v = x[i] + (bias ? bias[(i / hw) % C] : 0.f);
        if (pre) v += pre[i % pre_n];
        v = v > 0.f ? v : v * slope;
        if (res) v += res[i];
        out[i] = v;
    }
}

// conv -> +bias -> ReLU -> MaxPool2d(2,2) of the VGG stacks in one pass: relu(max4(x) + b) equals
// max4(relu(x + b)) exactly (fp add and ReLU are monotone), so the full-resolution activation is
// never written.  One thread = two horizontally adjacent outputs (one 16-byte load per input row).
__global__ __launch_bounds__(256) void bias_relu_pool2_kernel(const float *__restrict__ x, const float *__restrict__ bias,
                                                              float *__restrict__ out, long NC, int C, int H, int W)
{
    const int Ho = H >> 1, Wo = W >> 1, Wo2 = (Wo + 1) >> 1;
    const long total = NC * Ho * Wo2;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int xo2 = (int)(e % Wo2);
        long t = e / Wo2;
        const int yo = (int)(t % Ho);
        const long nc = t / Ho;
        const float b = bias ? bias[nc % C] : 0.f;
        const float *r0 = x + (nc * H + 2 * yo) * (long)W + 4 * xo2;
        const float *r1 = r0 + W;
        float *o = out + (nc * Ho + yo) * (long)Wo + 2 * xo2;
        if (4 * xo2 + 3 < W && (((uintptr_t)r0 | (uintptr_t)r1) & 15) == 0) {
            const float4 a = *reinterpret_cast<const float4 *>(r0), c = *reinterpret_cast<const float4 *>(r1);
            const float m0 = fmaxf(fmaxf(a.x, a.y), fmaxf(c.x, c.y)) + b, m1 = fmaxf(fmaxf(a.z, a.w), fmaxf(c.z, c.w)) + b;
            o[0] = m0 > 0.f ? m0 : 0.f;
            o[1] = m1 > 0.f ? m1 : 0.f;
        } else {
            for (int k = 0; k < 2; ++k) {
                const int xo = 2 * xo2 + k;
                if (xo < Wo) {
                    const float m = fmaxf(fmaxf(r0[2 * k], r0[2 * k + 1]), fmaxf(r1[2 * k], r1[2 * k + 1])) + b;
                    o[k] = m > 0.f ? m : 0.f;
                }
            }
        }
    }
}

// spatial-attention modulation of MRAPAFusion (ref_mrapa_restoration_arch.py:343-345): refs * sigmoid(mul) * 2 + add
// in one pass (three ATen launches otherwise), same operation order; layout-agnostic, in place on `mul`.
__global__ __launch_bounds__(256) void attn_modulate_kernel(const float4 *__restrict__ refs, float4 *__restrict__ mul,
                                                            const float4 *__restrict__ add, long n4)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 r = refs[i], m = mul[i], a = add[i];
        float4 o;
        o.x = r.x * (1.0f / (1.0f + expf(-m.x))) * 2.0f + a.x;
        o.y = r.y * (1.0f / (1.0f + expf(-m.y))) * 2.0f + a.y;
        o.z = r.z * (1.0f / (1.0f + expf(-m.z))) * 2.0f + a.z;
        o.w = r.w * (1.0f / (1.0f + expf(-m.w))) * 2.0f + a.w;
        mul[i] = o;
    }
}

// image -> the extractors' first-layer input (vgg_arch.py:150-153, contras_multi_extractor_arch.py:41 of the reference + the engine's
// channels-last packing): [N][3][H][W] -> [N][H][W][4] = ((x + 1) / 2 if range_norm, then (x - mean[c]) / std[c] if mean) in channels
// 0..2, zero in channel 3, in ONE pass (ATen: add, mul, sub, div, fill, strided copy -- six launches over the 40 reference images of a
// step).  Same operations in the same order as ATen's (x / 2 is its multiplication by the reciprocal, the division by std a division).
__global__ __launch_bounds__(256) void image_to_nhwc4_kernel(const float *__restrict__ img, float4 *__restrict__ out, long n_px, long HW,
                                                             int range_norm, const float *__restrict__ mean, const float *__restrict__ stdv)
{
    float m[3] = {0.f, 0.f, 0.f}, sd[3] = {1.f, 1.f, 1.f};
    if (mean) {
#pragma unroll
        for (int c = 0; c < 3; ++c) m[c] = mean[c], sd[c] = stdv[c];
    }
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n_px; i += (long)gridDim.x * blockDim.x) {
        const long n = i / HW, p = i - n * HW;
        const float *src = img + n * 3 * HW + p;
        float v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            v[c] = src[c * HW];
            if (range_norm) v[c] = (v[c] + 1.0f) * 0.5f;
            if (mean) v[c] = (v[c] - m[c]) / sd[c];
        }
        out[i] = make_float4(v[0], v[1], v[2], 0.0f);
    }
}

// the same on bf16 tensors (fp32 math, result rounded to bf16)
__global__ __launch_bounds__(256) void attn_modulate_bf16_kernel(const uint2 *__restrict__ refs, uint2 *__restrict__ mul,
                                                                 const uint2 *__restrict__ add, long n4)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const uint2 r = refs[i], m = mul[i], a = add[i];
        const unsigned int rr[2] = {r.x, r.y}, mm[2] = {m.x, m.y}, aa[2] = {a.x, a.y};
        unsigned int oo[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float r0 = __uint_as_float(rr[k] << 16), r1 = __uint_as_float(rr[k] & 0xffff0000u);
            const float m0 = __uint_as_float(mm[k] << 16), m1 = __uint_as_float(mm[k] & 0xffff0000u);
            const float a0 = __uint_as_float(aa[k] << 16), a1 = __uint_as_float(aa[k] & 0xffff0000u);
            const float o0 = r0 * (1.0f / (1.0f + expf(-m0))) * 2.0f + a0, o1 = r1 * (1.0f / (1.0f + expf(-m1))) * 2.0f + a1;
            const __hip_bfloat16 b0 = __float2bfloat16(o0), b1 = __float2bfloat16(o1);   // round-to-nearest-even
            oo[k] = (unsigned int)__builtin_bit_cast(unsigned short, b0) | ((unsigned int)__builtin_bit_cast(unsigned short, b1) << 16);
        }
        mul[i] = make_uint2(oo[0], oo[1]);
    }
}

// ---------------------------------------------------------------------------------------------
// Tail of the restoration net (ref_mrapa_restoration_arch.py:132-137: `base = F.interpolate(x, None, 4, 'bilinear', False)` ...
// `out + base`): out[b][c][Y][X] = y[b][Y][X][c] (channels-last, row stride ld) + bilinear(x)[b][c][Y][X], written NCHW -- one pass
// instead of ATen's upsample kernel, a strided add and a layout copy.  The interpolation restates upsample_bilinear2d_out_frame
// (align_corners = False, scale factors given: source index max(0, (dst + 0.5) / scale - 0.5), neighbour +1 unless on the last row /
// column, lambda1 = index - floor, lambda0 = 1 - lambda1; the blend h0 (w0 a + w1 b) + h1 (w0 c + w1 d) in the contraction the HIP
// build of ATen has: tests/test_kernels_gpu.py compares the bits with F.interpolate).
__global__ __launch_bounds__(256) void tail_bilinear_add_kernel(const float *__restrict__ y, const float *__restrict__ x, float *__restrict__ out, int B,
                                                                int C, int h, int w, int scale, int ld)
{
    const int H = h * scale, W = w * scale;
    const long total = (long)B * C * H * W;
    const float r = 1.0f / (float)scale;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int X = (int)(e % W);
        long t = e / W;
        const int Y = (int)(t % H);
        t /= H;
        const int c = (int)(t % C), b = (int)(t / C);
        float sy = r * ((float)Y + 0.5f) - 0.5f, sx = r * ((float)X + 0.5f) - 0.5f;
        sy = sy < 0.f ? 0.f : sy, sx = sx < 0.f ? 0.f : sx;
        const int y1 = (int)sy, x1 = (int)sx;
        const int yp = y1 < h - 1 ? 1 : 0, xp = x1 < w - 1 ? 1 : 0;
        const float h1 = sy - (float)y1, h0 = 1.f - h1, w1 = sx - (float)x1, w0 = 1.f - w1;
        const float *p = x + ((size_t)(b * C + c) * h + y1) * w + x1;
        const float a00 = p[0], a01 = p[xp], a10 = p[(size_t)yp * w], a11 = p[(size_t)yp * w + xp];
        const float top = __builtin_fmaf(w0, a00, w1 * a01), bot = __builtin_fmaf(w0, a10, w1 * a11);
        const float base = __builtin_fmaf(h0, top, h1 * bot);
        out[e] = y[((size_t)(b * H + Y) * W + X) * ld + c] + base;
    }
}

}  // namespace

MREFSR_EXPORT int mrefsr_attn_modulate_bf16(const void *refs, void *mul_inout, const void *add, int64_t n, mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(refs && mul_inout && add, "attn_modulate_bf16: null pointer");
    MREFSR_REQUIRE(n > 0 && (n & 3) == 0, "attn_modulate_bf16: n=%ld must be a positive multiple of 4", (long)n);
    const long n4 = n / 4, blocks = (n4 + 255) / 256;
    hipLaunchKernelGGL(attn_modulate_bf16_kernel, dim3((int)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const uint2 *>(refs), reinterpret_cast<uint2 *>(mul_inout), reinterpret_cast<const uint2 *>(add), n4);
    return mrefsr::check_launch("attn_modulate_bf16");
}

MREFSR_EXPORT int mrefsr_attn_modulate_f32(const float *refs, float *mul_inout, const float *add, int64_t n, mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(refs && mul_inout && add, "attn_modulate: null pointer");
    MREFSR_REQUIRE(n > 0 && (n & 3) == 0, "attn_modulate: n=%ld must be a positive multiple of 4", (long)n);
    const long n4 = n / 4, blocks = (n4 + 255) / 256;
    hipLaunchKernelGGL(attn_modulate_kernel, dim3((int)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(refs), reinterpret_cast<float4 *>(mul_inout),
                       reinterpret_cast<const float4 *>(add), n4);
    return mrefsr::check_launch("attn_modulate");
}

MREFSR_EXPORT int mrefsr_image_to_nhwc4_f32(const float *img, float *out, int64_t N, int64_t HW, int range_norm, const float *mean3,
                                            const float *std3, mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(img && out, "image_to_nhwc4: null pointer");
    MREFSR_REQUIRE(N > 0 && HW > 0, "image_to_nhwc4: N=%ld HW=%ld", (long)N, (long)HW);
    MREFSR_REQUIRE((mean3 == nullptr) == (std3 == nullptr), "image_to_nhwc4: mean and std go together");
    const long n_px = (long)N * HW, blocks = (n_px + 255) / 256;
    hipLaunchKernelGGL(image_to_nhwc4_kernel, dim3((int)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, (hipStream_t)stream, img,
                       reinterpret_cast<float4 *>(out), n_px, (long)HW, range_norm, mean3, std3);
    return mrefsr::check_launch("image_to_nhwc4");
}

MREFSR_EXPORT int mrefsr_bias_relu_pool2_f32(const float *x, const float *bias, float *out, int64_t N, int C, int H, int W,
                                             mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(x && out, "bias_relu_pool2: null pointer");
    MREFSR_REQUIRE(N > 0 && C > 0 && H >= 2 && W >= 2, "bias_relu_pool2: N=%ld C=%d H=%d W=%d", (long)N, C, H, W);
    const long total = (long)N * C * (H >> 1) * (((W >> 1) + 1) >> 1);
    const long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(bias_relu_pool2_kernel, dim3((int)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, (hipStream_t)stream, x,
                       bias, out, (long)N * C, C, H, W);
    return mrefsr::check_launch("bias_relu_pool2");
}

MREFSR_EXPORT int mrefsr_bias_act_res_f32(const float *x, const float *bias, const float *pre, int64_t pre_N,
                                          const float *residual, float *out, int64_t N, int C, int64_t HW, float slope,
                                          mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(x && out, "bias_act_res: null pointer");
    MREFSR_REQUIRE(N > 0 && C > 0 && HW > 0, "bias_act_res: N=%ld C=%d HW=%ld", (long)N, C, (long)HW);
    MREFSR_REQUIRE(!pre || (pre_N > 0 && N % pre_N == 0), "bias_act_res: pre_N=%ld must divide N=%ld", (long)pre_N, (long)N);
    const long n = (long)N * C * HW;
    const long pre_n = pre ? (long)pre_N * C * HW : 1;
    const bool v4 = (HW % 4 == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 16 == 0) &&
                    (!residual || (uintptr_t)residual % 16 == 0) && (!pre || (uintptr_t)pre % 16 == 0);
    if (v4) {
        const long n4 = n / 4, blocks = (n4 + 255) / 256;
        hipLaunchKernelGGL(bias_act_res_v4_kernel, dim3((int)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, (hipStream_t)stream,
                           (const float4 *)x, bias, (const float4 *)pre, pre_n / 4, (const float4 *)residual, (float4 *)out, n4,
                           (int)(HW / 4), C, slope);
    } else {
        const long blocks = (n + 255) / 256;
        hipLaunchKernelGGL(bias_act_res_kernel, dim3((int)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, (hipStream_t)stream, x,
                           bias, pre, pre_n, residual, out, n, (long)HW, C, slope);
    }
    return mrefsr::check_launch("bias_act_res");
}

MREFSR_EXPORT int mrefsr_fused_bias_act(const void *x, const void *bias, const void *ref, void *out, int64_t size_x,
                                        int step_b, int size_b, int act, int grad, float alpha, float scale, int dtype,
                                        mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(x && out, "fused_bias_act: null pointer");
    MREFSR_REQUIRE(size_x >= 0, "fused_bias_act: size_x=%ld", (long)size_x);
    MREFSR_REQUIRE(!bias || (step_b > 0 && size_b > 0), "fused_bias_act: bias with step_b=%d size_b=%d", step_b, size_b);
    MREFSR_REQUIRE(dtype >= 0 && dtype <= 3, "fused_bias_act: dtype=%d (0 f32, 1 f16, 2 bf16, 3 f64)", dtype);
    if (size_x == 0) return MREFSR_OK;
    const int mode = act * 10 + grad;
    hipStream_t st_ = (hipStream_t)stream;
    // the (plane, piece) grids below walk whole planes of step_b elements: anything else (a trailing partial plane, size_x < step_b)
    // takes the flat-index kernel at the end, which has no such precondition
    const bool whole_planes = !bias || (size_x % step_b == 0 && size_x >= step_b);
    if (dtype == 0 && whole_planes && (size_x % 4 == 0) && (!bias || step_b % 4 == 0) && ((uintptr_t)x % 16 == 0) &&
        ((uintptr_t)out % 16 == 0) && (!ref || (uintptr_t)ref % 16 == 0)) {
        const long n4 = size_x / 4;
        const long pv = bias ? step_b / 4 : n4, planes = n4 / pv, bx = (pv + 255) / 256;
        const dim3 g4((unsigned)(bx < 8192 ? bx : 8192), (unsigned)(planes < 32768 ? planes : 32768));
        hipLaunchKernelGGL(fused_bias_act_f32x4_kernel, g4, dim3(256), 0, st_,
                           (const float4 *)x, (const float *)bias, (const float4 *)ref, (float4 *)out, n4, step_b, size_b,
                           mode, alpha, scale);
        return mrefsr::check_launch("fused_bias_act");
    }
    if ((dtype == 1 || dtype == 2) && whole_planes && (size_x % 8 == 0) && (!bias || step_b % 8 == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 16 == 0) &&
        (!ref || (uintptr_t)ref % 16 == 0)) {
        const long n8 = size_x / 8;
        const long pv = bias ? step_b / 8 : n8, planes = n8 / pv, bx = (pv + 255) / 256;
        const dim3 g8((unsigned)(bx < 8192 ? bx : 8192), (unsigned)(planes < 32768 ? planes : 32768));
        if (dtype == 1)
            hipLaunchKernelGGL(fused_bias_act_x8_kernel<__half>, g8, dim3(256), 0, st_, (const uint4 *)x, (const __half *)bias, (const uint4 *)ref,
                               (uint4 *)out, n8, step_b, size_b, mode, alpha, scale);
        else
            hipLaunchKernelGGL(fused_bias_act_x8_kernel<__hip_bfloat16>, g8, dim3(256), 0, st_, (const uint4 *)x, (const __hip_bfloat16 *)bias,
                               (const uint4 *)ref, (uint4 *)out, n8, step_b, size_b, mode, alpha, scale);
        return mrefsr::check_launch("fused_bias_act");
    }
    const long blocks = (size_x + 255) / 256;
    const dim3 grid((int)(blocks < 16384 ? blocks : 16384));
    if (dtype == 0)
        hipLaunchKernelGGL(fused_bias_act_kernel<float>, grid, dim3(256), 0, st_, (const float *)x, (const float *)bias,
                           (const float *)ref, (float *)out, (long)size_x, step_b, size_b, mode, alpha, scale);
    else if (dtype == 1)
        hipLaunchKernelGGL(fused_bias_act_kernel<__half>, grid, dim3(256), 0, st_, (const __half *)x, (const __half *)bias,
                           (const __half *)ref, (__half *)out, (long)size_x, step_b, size_b, mode, alpha, scale);
    else if (dtype == 3)
        hipLaunchKernelGGL(fused_bias_act_kernel<double>, grid, dim3(256), 0, st_, (const double *)x, (const double *)bias,
                           (const double *)ref, (double *)out, (long)size_x, step_b, size_b, mode, alpha, scale);
    else
        hipLaunchKernelGGL(fused_bias_act_kernel<__hip_bfloat16>, grid, dim3(256), 0, st_, (const __hip_bfloat16 *)x,
                           (const __hip_bfloat16 *)bias, (const __hip_bfloat16 *)ref, (__hip_bfloat16 *)out, (long)size_x,
                           step_b, size_b, mode, alpha, scale);
    return mrefsr::check_launch("fused_bias_act");
}

MREFSR_EXPORT int mrefsr_tail_bilinear_add_f32(const float *y_nhwc, const float *x, float *out, int B, int C, int h, int w, int scale, int ld,
                                               mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(y_nhwc && x && out, "tail_bilinear_add: null pointer");
    MREFSR_REQUIRE(B > 0 && C > 0 && h > 0 && w > 0 && scale > 0 && ld >= C, "tail_bilinear_add: B=%d C=%d h=%d w=%d scale=%d ld=%d", B, C, h, w, scale, ld);
    const long total = (long)B * C * h * scale * w * scale;
    const long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(tail_bilinear_add_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, (hipStream_t)stream, y_nhwc, x, out, B, C, h,
                       w, scale, ld);
    return mrefsr::check_launch("tail_bilinear_add");
}
