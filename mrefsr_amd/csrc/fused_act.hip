// basicsr.ops.fused_act: y = act(x + bias[channel]) * scale, forward / backward / double-backward
// selected by act*10+grad (fused_bias_act_kernel.cu:19-50 of the reference).  One read + one
// write per element: HBM-bound.  fp32 / fp16 / bf16 storage, fp32 math.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include "common.h"

namespace {

template <typename T> __device__ __forceinline__ float ld(const T *p, long i);
template <> __device__ __forceinline__ float ld<float>(const float *p, long i) { return p[i]; }
template <> __device__ __forceinline__ float ld<__half>(const __half *p, long i) { return __half2float(p[i]); }
template <> __device__ __forceinline__ float ld<__hip_bfloat16>(const __hip_bfloat16 *p, long i) { return __bfloat162float(p[i]); }
template <typename T> __device__ __forceinline__ void st(T *p, long i, float v);
template <> __device__ __forceinline__ void st<float>(float *p, long i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void st<__half>(__half *p, long i, float v) { p[i] = __float2half(v); }
template <> __device__ __forceinline__ void st<__hip_bfloat16>(__hip_bfloat16 *p, long i, float v) { p[i] = __float2bfloat16(v); }

template <typename T>
__global__ __launch_bounds__(256) void fused_bias_act_kernel(const T *__restrict__ x, const T *__restrict__ b,
                                                             const T *__restrict__ ref, T *__restrict__ out, long size_x,
                                                             int step_b, int size_b, int mode, float alpha, float scale)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < size_x; i += (long)gridDim.x * blockDim.x) {
        float v = ld(x, i);
        if (b) v += ld(b, (i / step_b) % size_b);
        const float r = ref ? ld(ref, i) : 0.f;
        float y;
        switch (mode) {
        default:
        case 10: case 11: y = v; break;
        case 12: case 32: y = 0.f; break;
        case 30: y = (v > 0.f) ? v : v * alpha; break;
        case 31: y = (r > 0.f) ? v : v * alpha; break;
        }
        st(out, i, y * scale);
    }
}

// fp32 fast path: 16 B per lane when the bias index is constant over the 4 elements
__global__ __launch_bounds__(256) void fused_bias_act_f32x4_kernel(const float4 *__restrict__ x, const float *__restrict__ b,
                                                                   const float4 *__restrict__ ref, float4 *__restrict__ out,
                                                                   long n4, int step_b, int size_b, int mode, float alpha,
                                                                   float scale)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 v = x[i];
        if (b) {
            const float bv = b[((i * 4) / step_b) % size_b];
            v.x += bv; v.y += bv; v.z += bv; v.w += bv;
        }
        float4 r = ref ? ref[i] : make_float4(0.f, 0.f, 0.f, 0.f);
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
        out[i] = make_float4(yy[0], yy[1], yy[2], yy[3]);
    }
}

}  // namespace

MREFSR_EXPORT int mrefsr_fused_bias_act(const void *x, const void *bias, const void *ref, void *out, int64_t size_x,
                                        int step_b, int size_b, int act, int grad, float alpha, float scale, int dtype,
                                        mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(x && out, "fused_bias_act: null pointer");
    MREFSR_REQUIRE(size_x >= 0, "fused_bias_act: size_x=%ld", (long)size_x);
    MREFSR_REQUIRE(!bias || (step_b > 0 && size_b > 0), "fused_bias_act: bias with step_b=%d size_b=%d", step_b, size_b);
    MREFSR_REQUIRE(dtype >= 0 && dtype <= 2, "fused_bias_act: dtype=%d (0 f32, 1 f16, 2 bf16)", dtype);
    if (size_x == 0) return MREFSR_OK;
    const int mode = act * 10 + grad;
    hipStream_t st_ = (hipStream_t)stream;
    if (dtype == 0 && (size_x % 4 == 0) && (!bias || step_b % 4 == 0) && ((uintptr_t)x % 16 == 0) &&
        ((uintptr_t)out % 16 == 0) && (!ref || (uintptr_t)ref % 16 == 0)) {
        const long n4 = size_x / 4, blocks = (n4 + 255) / 256;
        hipLaunchKernelGGL(fused_bias_act_f32x4_kernel, dim3((int)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, st_,
                           (const float4 *)x, (const float *)bias, (const float4 *)ref, (float4 *)out, n4, step_b, size_b,
                           mode, alpha, scale);
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
    else
        hipLaunchKernelGGL(fused_bias_act_kernel<__hip_bfloat16>, grid, dim3(256), 0, st_, (const __hip_bfloat16 *)x,
                           (const __hip_bfloat16 *)bias, (const __hip_bfloat16 *)ref, (__hip_bfloat16 *)out, (long)size_x,
                           step_b, size_b, mode, alpha, scale);
    return mrefsr::check_launch("fused_bias_act");
}
