// DynAgg glue (ref_mrapa_restoration_arch.py:56-73) as one HBM-bound pass:
//   offset = conv_offset_mask(...)[:, :dg*18] + pre_offset (re-ordered [x,y] -> per-tap [y,x])
//   mask   = sigmoid(conv_offset_mask(...)[:, dg*18:])
//   guard  = sum |learned offset|  (accumulated on device: the reference's `.mean() > 100` test
//            forces a device->host sync per call; here the host may read it whenever it likes)
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void dynagg_prep_kernel(const float *__restrict__ om, const float *__restrict__ om_bias,
                                                          const float2 *__restrict__ pre,
                                                          float *__restrict__ offset, float *__restrict__ mask,
                                                          double *__restrict__ abs_sum, int B, int dg, int HW)
{
    const int n_i = dg * 9;
    const long total = (long)B * n_i * HW;
    float local = 0.f;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int p = (int)(e % HW);
        const long t = e / HW;
        const int i = (int)(t % n_i), b = (int)(t / n_i);
        const int tap = i % 9;
        const float2 pr = pre[((size_t)b * 9 + tap) * HW + p];  // [x, y]
        const size_t ob = ((size_t)b * 3 * n_i) * HW;
        float oy = om[ob + (size_t)(2 * i) * HW + p];
        float ox = om[ob + (size_t)(2 * i + 1) * HW + p];
        float mv = om[ob + (size_t)(2 * n_i + i) * HW + p];
        if (om_bias) {  // bias of conv_offset_mask, when the convolution was run without it
            oy += om_bias[2 * i];
            ox += om_bias[2 * i + 1];
            mv += om_bias[2 * n_i + i];
        }
        local += fabsf(oy) + fabsf(ox);
        const size_t fb = ((size_t)b * 2 * n_i) * HW;
        offset[fb + (size_t)(2 * i) * HW + p] = oy + pr.y;
        offset[fb + (size_t)(2 * i + 1) * HW + p] = ox + pr.x;
        mask[((size_t)b * n_i + i) * HW + p] = 1.0f / (1.0f + expf(-mv));
    }
    if (abs_sum) {
        __shared__ float red[4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(abs_sum, (double)(red[0] + red[1] + red[2] + red[3]));
    }
}

// Same glue for a channels-last conv_offset_mask output om [B][HW][27*dg]: 32 pixels per block go
// through LDS (coalesced 16-byte row loads in, 128-byte pixel runs out to the planar offset / mask
// tensors the DCN gather reads).
__global__ __launch_bounds__(256) void dynagg_prep_nhwc_kernel(const float *__restrict__ om, const float *__restrict__ om_bias,
                                                               const float2 *__restrict__ pre, float *__restrict__ offset,
                                                               float *__restrict__ mask, double *__restrict__ abs_sum, int dg,
                                                               int HW)
{
    extern __shared__ __align__(16) float dyn_tile[];  // [32][27*dg + 1]
    const int n_i = dg * 9, Com = 3 * n_i, LD = Com + 1;
    const int b = blockIdx.y, p0 = blockIdx.x * 32, tid = threadIdx.x;
    const int npx = HW - p0 < 32 ? HW - p0 : 32;
    const float *src = om + ((size_t)b * HW + p0) * Com;
    if ((Com & 3) == 0) {
        const int nv = npx * Com / 4;
        for (int i = tid; i < nv; i += 256) {
            const float4 v = reinterpret_cast<const float4 *>(src)[i];
            const int e = 4 * i, px = e / Com, c = e - px * Com;  // Com % 4 == 0: the 4 values stay in one pixel row
            float *d = dyn_tile + px * LD + c;
            d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = v.w;
        }
    } else {
        for (int e = tid; e < npx * Com; e += 256) dyn_tile[(e / Com) * LD + e % Com] = src[e];
    }
    __syncthreads();
    float local = 0.f;
    for (int e = tid; e < n_i * 32; e += 256) {
        const int px = e & 31, i = e >> 5;
        if (px >= npx) continue;
        const int p = p0 + px, tap = i % 9;
        const float *t = dyn_tile + px * LD;
        float oy = t[2 * i], ox = t[2 * i + 1], mv = t[2 * n_i + i];
        if (om_bias) oy += om_bias[2 * i], ox += om_bias[2 * i + 1], mv += om_bias[2 * n_i + i];
        local += fabsf(oy) + fabsf(ox);
        const float2 pr = pre[((size_t)b * 9 + tap) * HW + p];  // [x, y]
        const size_t fb = ((size_t)b * 2 * n_i) * HW;
        offset[fb + (size_t)(2 * i) * HW + p] = oy + pr.y;
        offset[fb + (size_t)(2 * i + 1) * HW + p] = ox + pr.x;
        mask[((size_t)b * n_i + i) * HW + p] = 1.0f / (1.0f + expf(-mv));
    }
    if (abs_sum) {
        __shared__ float red[4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(abs_sum, (double)(red[0] + red[1] + red[2] + red[3]));
    }
}

__global__ __launch_bounds__(256) void dynagg_prep_bwd_kernel(const float *__restrict__ g_offset,
                                                              const float *__restrict__ g_mask,
                                                              const float *__restrict__ mask, float *__restrict__ g_om,
                                                              int B, int dg, int HW)
{
    const int n_i = dg * 9;
    const long per_b = (long)3 * n_i * HW;
    const long total = (long)B * per_b;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int b = (int)(e / per_b);
        const long r = e - (long)b * per_b;
        const long n_off = (long)2 * n_i * HW;
        float v;
        if (r < n_off) {
            v = g_offset[(size_t)b * n_off + r];
        } else {
            const size_t mi = (size_t)b * n_i * HW + (r - n_off);
            const float m = mask[mi];
            v = g_mask[mi] * m * (1.0f - m);
        }
        g_om[e] = v;
    }
}

// The same with the result channels-last, [B][HW][27 dg] -- what the input-gradient and weight-gradient kernels of
// conv_offset_mask read -- plus the two reductions its consumers need: per-channel sums (the bias gradient) and max |g_om| (the
// scale of the fp16-split kernels).  32 pixels x 27 dg channels pass through an LDS tile (planar reads along the pixels,
// channels-last writes along the channels); a block walks `groups` such tiles and adds its sums with one atomic per channel.
// (Planar result + ATen's transposing copy + a reduction pass: 587 + 100 us at 20 x 160^2 x 216.)
constexpr int PXT = 32;   // pixels per tile (64: 55 KB of LDS and 64 values per thread in flight -- slower, 0.95 against 0.73 ms per step)

__global__ __launch_bounds__(256) void dynagg_prep_bwd_nhwc_kernel(const float *__restrict__ g_offset, const float *__restrict__ g_mask,
                                                                   const float *__restrict__ mask, float *__restrict__ g_om,
                                                                   float *__restrict__ bias_grad, unsigned int *__restrict__ amax_bits, int dg, int HW,
                                                                   int groups)
{
    extern __shared__ float tile[];   // [PXT][nc + 1]
    const int n_i = dg * 9, nc = 3 * n_i, ld = nc + 1, b = blockIdx.y, t = threadIdx.x, px = t & (PXT - 1), c0 = t / PXT;
    const float *go = g_offset + (size_t)b * 2 * n_i * HW, *gm = g_mask + (size_t)b * n_i * HW, *mk = mask + (size_t)b * n_i * HW;
    float sum = 0.f, amx = 0.f;   // thread t < nc: running sum of channel t
    constexpr int CR = 256 / PXT, MAXR = 256 / CR;      // channel rows per pass; channels per thread: 27 dg / CR <= MAXR
    float r[MAXR];
    auto load = [&](const int gi) {   // the tile's planar values into registers (the next tile's are in flight while this one is written)
        const int p = (blockIdx.x * groups + gi) * PXT + px;
#pragma unroll
        for (int k = 0; k < MAXR; ++k) {
            const int c = c0 + CR * k;
            float v = 0.f;
            if (c < nc && p < HW) {
                if (c < 2 * n_i) {
                    v = go[(size_t)c * HW + p];
                } else {
                    const size_t mi = (size_t)(c - 2 * n_i) * HW + p;
                    const float m = mk[mi];
                    v = gm[mi] * m * (1.0f - m);
                }
            }
            r[k] = v;
        }
    };
    load(0);
    for (int gi = 0; gi < groups; ++gi) {
        const int p0 = (blockIdx.x * groups + gi) * PXT;
        if (p0 >= HW) break;
#pragma unroll
        for (int k = 0; k < MAXR; ++k) {
            const int c = c0 + CR * k;
            if (c < nc) {
                tile[px * ld + c] = r[k];
                amx = fmaxf(amx, fabsf(r[k]));
            }
        }
        __syncthreads();
        if (gi + 1 < groups && p0 + PXT < HW) load(gi + 1);
        const int np = HW - p0 < PXT ? HW - p0 : PXT;
        if (t < nc) {   // channel t of every pixel of the tile: one coalesced row per pixel, summed on the way
            float *dst = g_om + ((size_t)b * HW + p0) * nc + t;
            float s = 0.f;
#pragma unroll 8
            for (int q = 0; q < np; ++q) {
                const float v = tile[q * ld + t];
                dst[(size_t)q * nc] = v;
                s += v;
            }
            sum += s;
        }
        __syncthreads();
    }
    if (bias_grad && t < nc) atomicAdd(bias_grad + t, sum);
    if (amax_bits) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amx = fmaxf(amx, __shfl_xor(amx, o, 64));
        if ((t & 63) == 0 && amx > 0.f && amx < 3.0e38f) atomicMax(amax_bits, __float_as_uint(amx));
    }
}

}  // namespace

MREFSR_EXPORT int mrefsr_dynagg_prep_f32(const float *om, const float *om_bias, const float *pre, float *offset, float *mask,
                                         double *abs_sum,
                                         int B, int dg, int H, int W, int om_nhwc, mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(om && pre && offset && mask, "dynagg_prep: null pointer");
    MREFSR_REQUIRE(B > 0 && dg > 0 && H > 0 && W > 0, "dynagg_prep: B=%d dg=%d H=%d W=%d", B, dg, H, W);
    if (om_nhwc) {
        MREFSR_REQUIRE(B <= 65535 && dg <= 32, "dynagg_prep: B=%d dg=%d out of range for the channels-last kernel", B, dg);
        const size_t lds = (size_t)32 * (27 * dg + 1) * sizeof(float);
        hipLaunchKernelGGL(dynagg_prep_nhwc_kernel, dim3(mrefsr::cdiv((long)H * W, 32), B), dim3(256), lds, (hipStream_t)stream, om,
                           om_bias, reinterpret_cast<const float2 *>(pre), offset, mask, abs_sum, dg, H * W);
        return mrefsr::check_launch("dynagg_prep(nhwc)");
    }
    const long total = (long)B * dg * 9 * H * W;
    const long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(dynagg_prep_kernel, dim3((int)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream,
                       om, om_bias, reinterpret_cast<const float2 *>(pre), offset, mask, abs_sum, B, dg, H * W);
    return mrefsr::check_launch("dynagg_prep");
}

MREFSR_EXPORT int mrefsr_dynagg_prep_bwd_f32(const float *g_offset, const float *g_mask, const float *mask, float *g_om,
                                             int B, int dg, int H, int W, mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(g_offset && g_mask && mask && g_om, "dynagg_prep_bwd: null pointer");
    MREFSR_REQUIRE(B > 0 && dg > 0 && H > 0 && W > 0, "dynagg_prep_bwd: B=%d dg=%d H=%d W=%d", B, dg, H, W);
    const long total = (long)B * 3 * dg * 9 * H * W;
    const long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(dynagg_prep_bwd_kernel, dim3((int)(blocks < 8192 ? blocks : 8192)), dim3(256), 0,
                       (hipStream_t)stream, g_offset, g_mask, mask, g_om, B, dg, H * W);
    return mrefsr::check_launch("dynagg_prep_bwd");
}

MREFSR_EXPORT int mrefsr_dynagg_prep_bwd_nhwc_f32(const float *g_offset, const float *g_mask, const float *mask, float *g_om, float *bias_grad,
                                                  float *amax, int B, int dg, int H, int W, mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(g_offset && g_mask && mask && g_om, "dynagg_prep_bwd_nhwc: null pointer");
    MREFSR_REQUIRE(B > 0 && B <= 65535 && dg > 0 && 27 * dg <= 256 && H > 0 && W > 0, "dynagg_prep_bwd_nhwc: B=%d dg=%d H=%d W=%d (27 dg <= 256)", B, dg,
                   H, W);
    const long HW = (long)H * W;
    long groups = HW * B / (PXT * 512);
    groups = groups < 1 ? 1 : (groups > 32 ? 32 : groups);
    const size_t lds = (size_t)PXT * (27 * dg + 1) * sizeof(float);
    hipLaunchKernelGGL(dynagg_prep_bwd_nhwc_kernel, dim3(mrefsr::cdiv(HW, PXT * groups), B), dim3(256), lds, (hipStream_t)stream, g_offset, g_mask, mask,
                       g_om, bias_grad, reinterpret_cast<unsigned int *>(amax), dg, (int)HW, (int)groups);
    return mrefsr::check_launch("dynagg_prep_bwd_nhwc");
}
