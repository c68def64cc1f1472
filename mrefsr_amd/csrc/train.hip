// Backward glue of the channels-last training engine (mrefsr_amd/archs/nhwc_train.py): what autograd would run as a dozen
// elementwise / reduction launches per layer of net_g (ref_mrapa_restoration_arch.py:101-348 under
// MultiRefRestorationModel.optimize_parameters, multi_ref_restoration_model.py:197-279) is one pass each here.
//   act_bwd_nhwc      g_pre = g_out * act'(out), per-channel sums of g_pre (= bias gradient), PReLU slope gradient
//   mrattn_bwd_nhwc   gradient of the multi-reference attention core (:321-335) on [N,H,W,C] tensors, probabilities recomputed
//   attn_modulate_bwd gradient of refs * sigmoid(mul) * 2 + add (:343-345)
// All HBM-bound.  Channel sums: fixed order inside a block, one float atomic per channel and block.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// Thread t owns channel unit u = t % U (V consecutive channels, V = 4 when C % 4 == 0 else 1) and walks the pixels
// po, po + ppp * grid, ... (ppp = 256 / U pixels per block and pass), so its channel sums live in registers; the ppp
// threads of a unit are then added in LDS in pixel order.
// act: 0 none (g_pre may be NULL: sums only), 1 LeakyReLU(slope) (slope 0 = ReLU), 2 PReLU(*slope_ptr) (+ slope gradient:
// x = out / slope for out < 0, which needs slope > 0 -- `flag` is raised otherwise and the caller re-runs unfused)
// ---------------------------------------------------------------------------------------------------------------
template <int V>
__global__ __launch_bounds__(256) void act_bwd_nhwc_kernel(const float *__restrict__ g_out, const float *__restrict__ out,
                                                           float *__restrict__ g_pre, int ld_pre, float *__restrict__ bias_grad,
                                                           float *__restrict__ slope_grad, unsigned int *__restrict__ amax_bits, long npix,
                                                           int C, int act, float slope, const float *__restrict__ slope_ptr,
                                                           int *__restrict__ flag)
{
    __shared__ float red[1024 + 256];
    float amx = 0.f;
    const int U = C / V, ppp = 256 / U;
    const int t = threadIdx.x, u = t % U, po = t / U;
    const bool active = po < ppp;
    float sl = slope;
    if (act == 2) {
        sl = *slope_ptr;
        if (!(sl > 0.f) && flag && blockIdx.x == 0 && t == 0) *flag = 1;
    }
    const float inv_sl = (act == 2 && sl != 0.f) ? 1.0f / sl : 0.f;
    float acc[V], ps = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) acc[i] = 0.f;
    if (active) {
        // four pixels per thread and round, their loads issued together (one pixel per round left every round a full memory
        // latency long: 46 us for a 26-MB tensor); fewer, longer blocks also mean fewer atomics on the few accumulator lines
        const long stride = (long)gridDim.x * ppp;
        auto body = [&](float (&g)[V], const float (&o)[V], const long p) {
#pragma unroll
            for (int i = 0; i < V; ++i) {
                if (act && !(o[i] > 0.f)) {
                    if (act == 2) ps = fmaf(g[i], o[i] * inv_sl, ps);
                    g[i] *= sl;
                }
                acc[i] += g[i];
                amx = fmaxf(amx, fabsf(g[i]));
            }
            if (g_pre) {
                if (V == 4) *reinterpret_cast<float4 *>(g_pre + p * ld_pre + 4 * u) = make_float4(g[0], g[1], g[2 % V], g[3 % V]);
                else g_pre[p * ld_pre + u] = g[0];
            }
        };
        auto load = [&](float (&g)[V], float (&o)[V], const long p) {
            if (V == 4) {
                const float4 gv = *reinterpret_cast<const float4 *>(g_out + p * C + 4 * u);
                g[0] = gv.x, g[1] = gv.y, g[2 % V] = gv.z, g[3 % V] = gv.w;
                if (act) {
                    const float4 ov = *reinterpret_cast<const float4 *>(out + p * C + 4 * u);
                    o[0] = ov.x, o[1] = ov.y, o[2 % V] = ov.z, o[3 % V] = ov.w;
                }
            } else {
                g[0] = g_out[p * C + u];
                if (act) o[0] = out[p * C + u];
            }
        };
        long p = (long)blockIdx.x * ppp + po;
        for (; p + 3 * stride < npix; p += 4 * stride) {
            float g0[V], g1[V], g2[V], g3[V], o0[V] = {}, o1[V] = {}, o2[V] = {}, o3[V] = {};
            load(g0, o0, p), load(g1, o1, p + stride), load(g2, o2, p + 2 * stride), load(g3, o3, p + 3 * stride);
            body(g0, o0, p), body(g1, o1, p + stride), body(g2, o2, p + 2 * stride), body(g3, o3, p + 3 * stride);
        }
        for (; p < npix; p += stride) {
            float g0[V], o0[V] = {};
            load(g0, o0, p);
            body(g0, o0, p);
        }
    }
    if (active) {
#pragma unroll
        for (int i = 0; i < V; ++i) red[po * C + V * u + i] = acc[i];
    }
    red[1024 + t] = ps;
    __syncthreads();
    // block totals (fixed order inside the block), then one float atomic per channel and block into the zero-initialised
    // gradient: the order of the blocks' contributions is not fixed -- as in the reference's own backward kernels, which
    // accumulate with atomics (deform_conv_cuda_kernel.cu:330,688) -- and costs neither a second launch nor a device-wide fence
    if (bias_grad)
        for (int c = t; c < C; c += 256) {
            float s = 0.f;
            for (int q = 0; q < ppp; ++q) s += red[q * C + c];
            atomicAdd(bias_grad + c, s);
        }
    if (slope_grad && t == 0) {
        float s = 0.f;
        for (int q = 0; q < 256; ++q) s += red[1024 + q];
        atomicAdd(slope_grad, s);
    }
    if (amax_bits) {   // max |g_pre| (non-negative floats order like their bit patterns): the scale of the fp16-split dgrad
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amx = fmaxf(amx, __shfl_xor(amx, o, 64));
        if ((t & 63) == 0 && amx > 0.f && amx < 3.0e38f) atomicMax(amax_bits, __float_as_uint(amx));
    }
}

// ---------------------------------------------------------------------------------------------------------------
// attention backward, channels-last: q [N][HW][c], emb [T*N][HW][c], ass [T*N][HW][2c] (t-major), g_out [N][HW][2c]
// -> g_q, g_emb, g_ass.  c/4 lanes share a pixel (as mrattn_fwd_nhwc_kernel); the softmax is recomputed from q and emb
// (they are read for their gradients anyway), the T embeddings of the pixel stay in registers.
// ---------------------------------------------------------------------------------------------------------------
template <int CH>
__global__ __launch_bounds__(256) void mrattn_bwd_nhwc_kernel(const float *__restrict__ q, const float *__restrict__ emb,
                                                              const float *__restrict__ ass, const float *__restrict__ g_out,
                                                              float *__restrict__ g_q, float *__restrict__ g_emb, float *__restrict__ g_ass,
                                                              int N, int T, long HW)
{
    constexpr int L = CH / 4, PPW = 64 / L, MT = 16;
    const int lane = threadIdx.x & 63, sub = lane % L, pw = lane / L;
    const long total = (long)N * HW;
    const long wave = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6, nwave = ((long)gridDim.x * blockDim.x) >> 6;
    for (long base = wave * PPW; base < total; base += nwave * PPW) {
        const long gp = base + pw;
        const bool ok = gp < total;
        const long g = ok ? gp : total - 1;
        const long n = g / HW, p = g - n * HW;
        const float4 qv = *reinterpret_cast<const float4 *>(q + g * CH + 4 * sub);
        const float4 g0 = *reinterpret_cast<const float4 *>(g_out + g * (2 * CH) + 4 * sub);
        const float4 g1 = *reinterpret_cast<const float4 *>(g_out + g * (2 * CH) + CH + 4 * sub);
        float4 e[MT];
        float a[MT], da[MT];
        float mx = -3.4e38f;
#pragma unroll
        for (int t = 0; t < MT; ++t)
            if (t < T) {
                e[t] = *reinterpret_cast<const float4 *>(emb + (((long)t * N + n) * HW + p) * CH + 4 * sub);
                float d = qv.x * e[t].x + qv.y * e[t].y + qv.z * e[t].z + qv.w * e[t].w;
#pragma unroll
                for (int o = L / 2; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
                a[t] = d;
                mx = fmaxf(mx, d);
            }
        float den = 0.f;
#pragma unroll
        for (int t = 0; t < MT; ++t)
            if (t < T) {
                a[t] = expf(a[t] - mx);
                den += a[t];
            }
        const float inv = 1.0f / den;
        float dot = 0.f;
#pragma unroll
        for (int t = 0; t < MT; ++t)
            if (t < T) {
                a[t] *= inv;
                const size_t o = (size_t)((((long)t * N + n) * HW + p) * (2 * CH) + 4 * sub);
                const float4 a0 = *reinterpret_cast<const float4 *>(ass + o), a1 = *reinterpret_cast<const float4 *>(ass + o + CH);
                float d = g0.x * a0.x + g0.y * a0.y + g0.z * a0.z + g0.w * a0.w + g1.x * a1.x + g1.y * a1.y + g1.z * a1.z + g1.w * a1.w;
#pragma unroll
                for (int s = L / 2; s > 0; s >>= 1) d += __shfl_xor(d, s, 64);
                da[t] = d;
                dot = fmaf(a[t], d, dot);
                if (ok) {
                    *reinterpret_cast<float4 *>(g_ass + o) = make_float4(g0.x * a[t], g0.y * a[t], g0.z * a[t], g0.w * a[t]);
                    *reinterpret_cast<float4 *>(g_ass + o + CH) = make_float4(g1.x * a[t], g1.y * a[t], g1.z * a[t], g1.w * a[t]);
                }
            }
        float4 gq = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int t = 0; t < MT; ++t)
            if (t < T) {
                const float dl = a[t] * (da[t] - dot);   // d logit_t
                gq.x = fmaf(dl, e[t].x, gq.x), gq.y = fmaf(dl, e[t].y, gq.y), gq.z = fmaf(dl, e[t].z, gq.z), gq.w = fmaf(dl, e[t].w, gq.w);
                if (ok)
                    *reinterpret_cast<float4 *>(g_emb + (((long)t * N + n) * HW + p) * CH + 4 * sub) =
                        make_float4(dl * qv.x, dl * qv.y, dl * qv.z, dl * qv.w);
            }
        if (ok) *reinterpret_cast<float4 *>(g_q + g * CH + 4 * sub) = gq;
    }
}

__global__ __launch_bounds__(256) void attn_modulate_bwd_kernel(const float4 *__restrict__ g, const float4 *__restrict__ refs,
                                                                const float4 *__restrict__ mul, float4 *__restrict__ g_refs,
                                                                float4 *__restrict__ g_mul, long n4)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 gv = g[i], r = refs[i], m = mul[i];
        float4 gr, gm;
#define MREFSR_MOD1(f)                                   \
    {                                                    \
        const float s = 1.0f / (1.0f + expf(-m.f));      \
        gr.f = gv.f * 2.0f * s;                          \
        gm.f = gv.f * r.f * 2.0f * s * (1.0f - s);       \
    }
        MREFSR_MOD1(x) MREFSR_MOD1(y) MREFSR_MOD1(z) MREFSR_MOD1(w)
#undef MREFSR_MOD1
        g_refs[i] = gr;
        g_mul[i] = gm;
    }
}

}  // namespace

#ifndef ACT_BWD_PIX
#define ACT_BWD_PIX 16      // pixels per thread before the grid is capped
#endif
#ifndef ACT_BWD_MAXB
#define ACT_BWD_MAXB 512
#endif
MREFSR_EXPORT int mrefsr_act_bwd_blocks(int64_t npix, int C)
{
    if (npix <= 0 || C <= 0 || C > 1024) return -1;
    const int V = (C % 4 == 0) ? 4 : 1, U = C / V;
    if (U > 256) return -1;
    const int ppp = 256 / U;
    const long want = (npix + (long)ppp * ACT_BWD_PIX - 1) / ((long)ppp * ACT_BWD_PIX);
    return (int)(want < 1 ? 1 : (want > ACT_BWD_MAXB ? ACT_BWD_MAXB : want));
}

MREFSR_EXPORT int mrefsr_act_bwd_nhwc_f32(const float *g_out, const float *out, float *g_pre, int ld_pre, float *bias_grad, float *slope_grad,
                                          float *amax, int64_t npix, int C, int act, float slope, const float *slope_ptr, int *flag,
                                          mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(g_out, "act_bwd_nhwc: null pointer");
    MREFSR_REQUIRE(act >= 0 && act <= 2 && (act == 0 || out) && (act != 2 || slope_ptr), "act_bwd_nhwc: act=%d needs out%s", act,
                   act == 2 ? " and slope_ptr" : "");
    const int blocks = mrefsr_act_bwd_blocks(npix, C);
    MREFSR_REQUIRE(blocks > 0, "act_bwd_nhwc: npix=%ld C=%d (C <= 1024; C <= 256 unless a multiple of 4)", (long)npix, C);
    MREFSR_REQUIRE(!g_pre || ld_pre >= C, "act_bwd_nhwc: ld_pre=%d < C=%d", ld_pre, C);
    if (C % 4 == 0) {
        MREFSR_REQUIRE(!g_pre || ld_pre % 4 == 0, "act_bwd_nhwc: ld_pre=%d must be a multiple of 4", ld_pre);
        hipLaunchKernelGGL(act_bwd_nhwc_kernel<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g_out, out, g_pre, ld_pre, bias_grad, slope_grad,
                           reinterpret_cast<unsigned int *>(amax), (long)npix, C, act, slope, slope_ptr, flag);
    } else {
        hipLaunchKernelGGL(act_bwd_nhwc_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g_out, out, g_pre, ld_pre, bias_grad, slope_grad,
                           reinterpret_cast<unsigned int *>(amax), (long)npix, C, act, slope, slope_ptr, flag);
    }
    return mrefsr::check_launch("act_bwd_nhwc");
}

MREFSR_EXPORT int mrefsr_mrattn_bwd_nhwc_f32(const float *q, const float *emb, const float *ass, const float *g_out, float *g_q,
                                             float *g_emb, float *g_ass, int N, int T, int c, int HW, mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(q && emb && ass && g_out && g_q && g_emb && g_ass, "mrattn_bwd_nhwc: null pointer");
    MREFSR_REQUIRE(N > 0 && T > 0 && T <= 16 && HW > 0, "mrattn_bwd_nhwc: N=%d T=%d HW=%d (T <= 16)", N, T, HW);
    const long waves = ((long)N * HW * (c / 4) + 63) / 64;
    const long blocks = (waves + 3) / 4;
    const dim3 grid((int)(blocks < 65536 ? blocks : 65536));
    hipStream_t st = (hipStream_t)stream;
    if (c == 256) hipLaunchKernelGGL(mrattn_bwd_nhwc_kernel<256>, grid, dim3(256), 0, st, q, emb, ass, g_out, g_q, g_emb, g_ass, N, T, (long)HW);
    else if (c == 128) hipLaunchKernelGGL(mrattn_bwd_nhwc_kernel<128>, grid, dim3(256), 0, st, q, emb, ass, g_out, g_q, g_emb, g_ass, N, T, (long)HW);
    else if (c == 64) hipLaunchKernelGGL(mrattn_bwd_nhwc_kernel<64>, grid, dim3(256), 0, st, q, emb, ass, g_out, g_q, g_emb, g_ass, N, T, (long)HW);
    else return mrefsr::fail(MREFSR_E_UNSUPPORTED, "mrattn_bwd_nhwc: c=%d (64, 128 or 256: the three MRAPAFusion heads)", c);
    return mrefsr::check_launch("mrattn_bwd_nhwc");
}

MREFSR_EXPORT int mrefsr_attn_modulate_bwd_f32(const float *g, const float *refs, const float *mul, float *g_refs, float *g_mul, int64_t n,
                                               mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(g && refs && mul && g_refs && g_mul, "attn_modulate_bwd: null pointer");
    MREFSR_REQUIRE(n > 0 && n % 4 == 0, "attn_modulate_bwd: n=%ld must be a positive multiple of 4", (long)n);
    const long n4 = n / 4, blocks = (n4 + 255) / 256;
    hipLaunchKernelGGL(attn_modulate_bwd_kernel, dim3((int)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(g), reinterpret_cast<const float4 *>(refs), reinterpret_cast<const float4 *>(mul),
                       reinterpret_cast<float4 *>(g_refs), reinterpret_cast<float4 *>(g_mul), n4);
    return mrefsr::check_launch("attn_modulate_bwd");
}
