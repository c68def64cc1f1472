// basicsr.ops.upfirdn2d: zero-insert upsample, pad/crop, FIR with the flipped kernel, decimate
// (upfirdn2d.py:162-192 / upfirdn2d_kernel.cu:50-106 of the reference).  One thread per output
// sample gathers only the non-zero taps (kh/up_y x kw/up_x of them); the FIR sits in LDS.
// HBM-bound: (in + out) * 4 bytes; input re-reads are served by L1/L2.
#include "common.h"

namespace {

__device__ __forceinline__ int floor_div(int a, int b)
{
    int c = a / b;
    if (c * b > a) --c;
    return c;
}

struct UpParams {
    int major, in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y, px0, py0, out_h, out_w;
};

__global__ __launch_bounds__(256) void upfirdn2d_kernel(const float *__restrict__ in, const float *__restrict__ kernel,
                                                        float *__restrict__ out, UpParams p)
{
    extern __shared__ float sk[];  // flipped FIR
    for (int i = threadIdx.x; i < p.kh * p.kw; i += blockDim.x) {
        const int ky = i / p.kw, kx = i - ky * p.kw;
        sk[i] = kernel[(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)];
    }
    __syncthreads();
    const long total = (long)p.major * p.out_h * p.out_w * p.minor;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int mn = (int)(e % p.minor);
        long t = e / p.minor;
        const int ox = (int)(t % p.out_w);
        t /= p.out_w;
        const int oy = (int)(t % p.out_h), mj = (int)(t / p.out_h);
        // taps ky with (oy*down + ky - pad0) = iy*up, 0 <= iy < in_h
        const int by = oy * p.down_y - p.py0, bx = ox * p.down_x - p.px0;
        int iy0 = floor_div(by + p.up_y - 1, p.up_y);  // ceil(by / up)
        if (iy0 < 0) iy0 = 0;
        int iy1 = floor_div(by + p.kh - 1, p.up_y);
        if (iy1 > p.in_h - 1) iy1 = p.in_h - 1;
        int ix0 = floor_div(bx + p.up_x - 1, p.up_x);
        if (ix0 < 0) ix0 = 0;
        int ix1 = floor_div(bx + p.kw - 1, p.up_x);
        if (ix1 > p.in_w - 1) ix1 = p.in_w - 1;
        float v = 0.f;
        for (int iy = iy0; iy <= iy1; ++iy) {
            const int ky = iy * p.up_y - by;
            const float *row = in + (((size_t)mj * p.in_h + iy) * p.in_w) * p.minor + mn;
            for (int ix = ix0; ix <= ix1; ++ix) v = fmaf(row[(size_t)ix * p.minor], sk[ky * p.kw + ix * p.up_x - bx], v);
        }
        out[e] = v;
    }
}

}  // namespace

MREFSR_EXPORT int mrefsr_upfirdn2d_f32(const float *in, const float *kernel, float *out, int major, int in_h, int in_w,
                                       int minor, int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0,
                                       int pad_x1, int pad_y0, int pad_y1, mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(in && kernel && out, "upfirdn2d: null pointer");
    MREFSR_REQUIRE(major > 0 && in_h > 0 && in_w > 0 && minor > 0 && kh > 0 && kw > 0, "upfirdn2d: bad sizes");
    MREFSR_REQUIRE(up_x > 0 && up_y > 0 && down_x > 0 && down_y > 0, "upfirdn2d: up/down must be positive");
    UpParams p;
    p.major = major; p.in_h = in_h; p.in_w = in_w; p.minor = minor; p.kh = kh; p.kw = kw;
    p.up_x = up_x; p.up_y = up_y; p.down_x = down_x; p.down_y = down_y; p.px0 = pad_x0; p.py0 = pad_y0;
    p.out_h = (in_h * up_y + pad_y0 + pad_y1 - kh + down_y) / down_y;
    p.out_w = (in_w * up_x + pad_x0 + pad_x1 - kw + down_x) / down_x;
    MREFSR_REQUIRE(p.out_h > 0 && p.out_w > 0, "upfirdn2d: empty output (%d x %d)", p.out_h, p.out_w);
    const long total = (long)major * p.out_h * p.out_w * minor;
    const long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(upfirdn2d_kernel, dim3((int)(blocks < 16384 ? blocks : 16384)), dim3(256),
                       (size_t)kh * kw * sizeof(float), (hipStream_t)stream, in, kernel, out, p);
    return mrefsr::check_launch("upfirdn2d");
}
