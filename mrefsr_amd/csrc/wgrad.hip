// Weight gradient of the 3x3 stride-1 'same' convolutions of net_g's training step (what torch.autograd runs as
// miopenConvolutionBackwardWeights under multi_ref_restoration_model.py:197-279):
//     dW[co][ci][ty][tx] = sum over n, y, x of  g[n][y][x][co] * in[n][y + ty - 1][x + tx - 1][ci]        (zero outside the image)
// on the 16-bit matrix pipe with the operand splits of conv_nhwc.hip (fp32-equivalent: three fp16 products per term).  It is a
// GEMM whose K dimension is the PIXEL index, so both operands have to be transposed on their way into LDS ([channel][pixel]
// fp16 planes) and the 3 x 3 shifts become operand offsets:
//   * work is partitioned by the INPUT column q = x + tx - 1 (32 per block) and the gradient row y: every term has a unique
//     (y, q), the input rows y - 1, y, y + 1 sit unshifted in a 4-slot LDS ring, and the gradient row is staged three times,
//     shifted by tx (g_tx[q] = g[q - tx + 1], one halo column on either side) -- so every MFMA fragment is an ALIGNED 16-byte
//     read of 8 consecutive pixels;
//   * block = 64 output channels x 64 input channels x 9 taps; wave w owns the (32 co, 32 ci) quarter (w & 1, w >> 1) of all
//     nine taps: 9 accumulator tiles, 6 + 6 fragment reads per 16-pixel k-step for 27 MFMAs;
//   * the gradient is scaled by the power of two that brings its maximum (in_amax, from mrefsr_act_bwd_nhwc_f32) into
//     [2^13, 2^14), as in mrefsr_conv_nhwc_scaled_f32: g S = gh + gl (+ GH2 = gh 2^-11 derived in registers), in = xh + XL 2^-11;
//     dW S = sum  gh xh + gl xh + GH2 XL;
//   * a block walks RG gradient rows (the next row's global loads in flight during a row's MFMAs), then leaves its 64 x 64 x 9
//     partial in the caller's workspace; conv_wgrad_reduce_kernel adds the partials in block order (deterministic -- float
//     atomics straight into dW serialise per cache line: 30 M of them per trunk convolution made the first version 6x slower
//     than MIOpen).
#include "common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int WQ = 32;              // LDS columns per block (two 16-pixel k-steps)
constexpr int WQV = 30;             // of which input columns: with one halo column either side the gradient row is 32 wide -- 16 pairs
                                    // per thread set, no 17th (its two registers per stage were the ones that spilled)
constexpr int ROW_LD = 80;          // bytes per channel row: 32 fp16 + 16 pad (conflict-free 16-byte fragment reads)
constexpr int PLANE = 64 * ROW_LD;  // one fp16 plane of 64 channels
constexpr int X_SLOT = 2 * PLANE;   // xh | XL of one input row
constexpr int G_OFF = 4 * X_SLOT;   // gradient copies behind the 4-slot input ring
constexpr int G_COPY = 2 * PLANE;   // gh | gl of one shifted copy
constexpr int G_BUF = 3 * G_COPY;    // the three shifted copies of one gradient row
constexpr int LDS_BYTES = G_OFF + 2 * G_BUF;   // double-buffered: the staging waves fill row y + 1 while the MFMA waves read row y

__device__ __forceinline__ unsigned short h_bits(_Float16 h) { return __builtin_bit_cast(unsigned short, h); }

// Wave-specialised: waves 0-3 issue the MFMAs of gradient row y while waves 4-7 load, split, transpose and store row y + 1
// (input row y + 2) -- one barrier per row; the MFMA waves never wait for global memory, the staging waves hold two rows of raw
// loads in flight.  (All eight waves staging, then all computing, left the matrix pipe 20 % busy.)
// Up to MREFSR_WGRAD_MAX_JOBS weight gradients of identical geometry in one launch (the 32 convolutions of a residual trunk):
// the operands of job j = blockIdx.x / units travel by value in the kernel arguments, no table in device memory to upload.
struct WgradJobs {
    const float *x[MREFSR_WGRAD_MAX_JOBS], *g[MREFSR_WGRAD_MAX_JOBS], *g_amax[MREFSR_WGRAD_MAX_JOBS];
    float *dw[MREFSR_WGRAD_MAX_JOBS];
};

__global__ __launch_bounds__(512, 1) void conv_wgrad3x3_kernel(const WgradJobs J, int units, int ld_x, int Cin, int ld_g, int Cout,
                                                               float *__restrict__ partial_all, int H, int W, int n_qt, int n_rg, int RG,
                                                               int *__restrict__ range_flag)
{
    const int job = blockIdx.x / units, unit = blockIdx.x - job * units;
    const float *__restrict__ x = J.x[job], *__restrict__ g = J.g[job], *__restrict__ g_amax = J.g_amax[job];
    float *__restrict__ partial = partial_all + (size_t)job * units * gridDim.y * gridDim.z * 9 * 4096;
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wvb = tid >> 6, l31 = lane & 31, kh = lane >> 5;
    const bool mfma_wave = wvb < 4;
    const int wv = wvb & 3;            // index within the role
    const int cs = wv & 1, is = wv >> 1;
    int u = unit;
    const int qt = u % n_qt;
    u /= n_qt;
    const int rg = u % n_rg, n = u / n_rg;
    const int q0 = qt * WQV, y0 = rg * RG, ci0 = blockIdx.y * 64, co0 = blockIdx.z * 64;
    const int y1 = (y0 + RG < H) ? y0 + RG : H;

    float gs = 1.f, oscale = 1.f;
    {
        const float am = g_amax ? *g_amax : 0.f;
        if (am > 1.0e-30f && am < 3.0e38f) {
            int e;
            (void)frexpf(am, &e);
            gs = ldexpf(1.f, 14 - e);
            oscale = ldexpf(1.f, e - 14);
        }
    }

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    // staging role: 4 channels (cq) of two adjacent pixels (pair pj); a thread writes both pixels of a channel as one 4-byte LDS
    // store at dword (4 cq + i) * 20 + pj.  A wave takes the channel quads 4 w .. 4 w + 3 of all 16 pairs: 80 cq mod 64 =
    // 16 (cq & 3), so its 64 lanes hit 64 different banks (with cq = tid & 15 the 16 quads of a pair fell on 4 banks: 4-way
    // conflicts on every store); 4 consecutive lanes still read 64 contiguous bytes of a pixel
    const int cq = 4 * wv + (lane & 3), pj = lane >> 2;
    auto load4 = [&](const float *base, const bool ok, const int c, const int C, const int ld) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) {
            if (c + 3 < C && (ld & 3) == 0) {
                v = *reinterpret_cast<const float4 *>(base + c);
            } else {
                if (c < C) v.x = base[c];
                if (c + 1 < C) v.y = base[c + 1];
                if (c + 2 < C) v.z = base[c + 2];
                if (c + 3 < C) v.w = base[c + 3];
            }
        }
        return v;
    };
    // input row r, columns q0 + 2 pj, + 1
    auto load_x = [&](const int r, float4 *xr) {
        const bool row_ok = r >= 0 && r < H;
        const float *base = x + (((size_t)n * H + (row_ok ? r : 0)) * W) * ld_x;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int gq = q0 + 2 * pj + k;
            const bool ok = row_ok && 2 * pj + k < WQV && gq < W;   // columns 30, 31 of the tile stay zero: they belong to the next block
            xr[k] = load4(base + (size_t)(ok ? gq : 0) * ld_x, ok, ci0 + cq * 4, Cin, ld_x);
        }
    };
    auto commit_x = [&](const int r, const float4 *xr) {
        unsigned char *slot = smem + (r & 3) * X_SLOT;
        const float a[2][4] = {{xr[0].x, xr[0].y, xr[0].z, xr[0].w}, {xr[1].x, xr[1].y, xr[1].z, xr[1].w}};
        float mx = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            mx = fmaxf(mx, fmaxf(fabsf(a[0][i]), fabsf(a[1][i])));
            const _Float16 h0 = (_Float16)a[0][i], h1 = (_Float16)a[1][i];
            const _Float16 l0 = (_Float16)((a[0][i] - (float)h0) * 2048.f), l1 = (_Float16)((a[1][i] - (float)h1) * 2048.f);
            unsigned char *d = slot + (cq * 4 + i) * ROW_LD + pj * 4;
            *reinterpret_cast<unsigned int *>(d) = (unsigned int)h_bits(h0) | ((unsigned int)h_bits(h1) << 16);
            *reinterpret_cast<unsigned int *>(d + PLANE) = (unsigned int)h_bits(l0) | ((unsigned int)h_bits(l1) << 16);
        }
        if (range_flag && !(mx <= 65000.f)) atomicOr(range_flag, 1);
    };
    // gradient row y, columns q0 - 1 + p for the pair p = 2 j, 2 j + 1 (j = pj): p = 0 .. 31 is all the 30 input columns need
    auto load_g = [&](const int y, const int j, float4 *gr) {
        const float *base = g + (((size_t)n * H + y) * W) * ld_g;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int gx = q0 - 1 + 2 * j + k;
            const bool ok = gx >= 0 && gx < W;
            gr[k] = load4(base + (size_t)(ok ? gx : 0) * ld_g, ok, co0 + cq * 4, Cout, ld_g);
        }
    };
    auto commit_g = [&](const int j, const float4 *gr, const int y) {
        const float a[2][4] = {{gr[0].x, gr[0].y, gr[0].z, gr[0].w}, {gr[1].x, gr[1].y, gr[1].z, gr[1].w}};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float s0 = a[0][i] * gs, s1 = a[1][i] * gs;
            const _Float16 h0 = (_Float16)s0, h1 = (_Float16)s1;
            const _Float16 l0 = (_Float16)(s0 - (float)h0), l1 = (_Float16)(s1 - (float)h1);
            const unsigned int hp = (unsigned int)h_bits(h0) | ((unsigned int)h_bits(h1) << 16);
            const unsigned int lp = (unsigned int)h_bits(l0) | ((unsigned int)h_bits(l1) << 16);
            unsigned char *row = smem + G_OFF + (y & 1) * G_BUF + (cq * 4 + i) * ROW_LD;
            // copy tx holds column p at q = p + tx - 2:   tx = 0: q = 2 j - 2, 2 j - 1 (a pair);  tx = 2: q = 2 j, 2 j + 1 (a pair);
            //                                             tx = 1: q = 2 j - 1 and 2 j (two halves of different pairs)
            if (j >= 1) {
                *reinterpret_cast<unsigned int *>(row + (2 * j - 2) * 2) = hp;
                *reinterpret_cast<unsigned int *>(row + PLANE + (2 * j - 2) * 2) = lp;
                *reinterpret_cast<unsigned short *>(row + G_COPY + (2 * j - 1) * 2) = h_bits(h0);
                *reinterpret_cast<unsigned short *>(row + G_COPY + PLANE + (2 * j - 1) * 2) = h_bits(l0);
            }
            if (j <= 15) {
                *reinterpret_cast<unsigned int *>(row + 2 * G_COPY + (2 * j) * 2) = hp;
                *reinterpret_cast<unsigned int *>(row + 2 * G_COPY + PLANE + (2 * j) * 2) = lp;
                *reinterpret_cast<unsigned short *>(row + G_COPY + (2 * j) * 2) = h_bits(h1);
                *reinterpret_cast<unsigned short *>(row + G_COPY + PLANE + (2 * j) * 2) = h_bits(l1);
            }
        }
    };

    struct Stage {
        float4 x[2], g[2];
    };
    auto load_stage = [&](Stage &st, const int y) {   // what is staged for iteration y: input row y + 1, gradient row y
        load_x(y + 1, st.x);
        load_g(y, pj, st.g);
    };
    // columns 30 / 31 of the gradient copies are never written (the matching input columns are zero): they must not hold NaNs
    for (int i = tid; i < 2 * G_BUF / 16; i += 512) *reinterpret_cast<u32x4 *>(smem + G_OFF + i * 16) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();

    if (!mfma_wave) {
        // ---- staging waves.  Before iteration y's barrier: input rows <= y + 1 and gradient row y are in LDS.
        Stage sa, sb;
        {   // the block's first three input rows and first gradient row: every load requested before the first commit
            float4 x0[2], x1[2];
            load_x(y0 - 1, x0);
            load_x(y0, x1);
            load_stage(sa, y0);
            if (y0 + 1 < y1) load_stage(sb, y0 + 1);
            commit_x(y0 - 1, x0);
            commit_x(y0, x1);
        }
        commit_x(y0 + 1, sa.x);
        commit_g(pj, sa.g, y0);
        if (y0 + 2 < y1) load_stage(sa, y0 + 2);
        // iteration y stages row y + 1 from the stage loaded two iterations ago, then refills that stage with row y + 3
        auto stage_row = [&](const int y, Stage &st) {
            __syncthreads();                       // MFMA waves are done with row y - 1: its buffers are free
            if (y + 1 < y1) {
                commit_x(y + 2, st.x);
                commit_g(pj, st.g, y + 1);
                if (y + 3 < y1) load_stage(st, y + 3);
            }
        };
        for (int y = y0; y < y1; y += 2) {
            stage_row(y, sb);
            if (y + 1 < y1) stage_row(y + 1, sa);
        }
        return;
    }

    // ---- MFMA waves
    for (int y = y0; y < y1; ++y) {
        __syncthreads();                           // row y is staged
        const unsigned char *gbuf = smem + G_OFF + (y & 1) * G_BUF;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int koff = (h * 16 + kh * 8) * 2;
            u32x4 ga[3][3], xb[3][2];
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) {
                const unsigned char *s = gbuf + tx * G_COPY + (cs * 32 + l31) * ROW_LD + koff;
                ga[tx][0] = *reinterpret_cast<const u32x4 *>(s);
                ga[tx][1] = *reinterpret_cast<const u32x4 *>(s + PLANE);
                const f16x2 k11 = {(_Float16)0.00048828125f, (_Float16)0.00048828125f};
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const unsigned int d = ga[tx][0][w];
                    ga[tx][2][w] = __builtin_bit_cast(unsigned int, __builtin_bit_cast(f16x2, d) * k11);
                }
            }
#pragma unroll
            for (int ty = 0; ty < 3; ++ty) {
                const unsigned char *s = smem + ((y + ty - 1) & 3) * X_SLOT + (is * 32 + l31) * ROW_LD + koff;
                xb[ty][0] = *reinterpret_cast<const u32x4 *>(s);
                xb[ty][1] = *reinterpret_cast<const u32x4 *>(s + PLANE);
            }
#pragma unroll
            for (int ty = 0; ty < 3; ++ty)
#pragma unroll
                for (int tx = 0; tx < 3; ++tx) {
                    f32x16 c = acc[ty * 3 + tx];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ga[tx][1]), __builtin_bit_cast(f16x8, xb[ty][0]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ga[tx][2]), __builtin_bit_cast(f16x8, xb[ty][1]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ga[tx][0]), __builtin_bit_cast(f16x8, xb[ty][0]), c, 0, 0, 0);
                    acc[ty * 3 + tx] = c;
                }
        }
    }

    // this block's 64 x 64 x 9 partial: [unit][ci tile][co tile][tap][co 64][ci 64], 128 contiguous bytes per half-wave and register
    float *pp = partial + ((((size_t)unit * gridDim.y + blockIdx.y) * gridDim.z + blockIdx.z) * 9) * 4096 + is * 32 + l31;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int col = cs * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
            pp[(size_t)t * 4096 + col * 64] = acc[t][e] * oscale;
        }
}

// dw[co][ci][tap] (+)= sum over the units of partial[unit][ci tile][co tile][tap][co][ci], in unit order (deterministic)
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float *__restrict__ partial_all, const WgradJobs J, long stride_co, long stride_ci,
                                                                int Cin, int Cout, int units, int n_cit, int n_cot, int accumulate)
{
    const long total = (long)n_cit * n_cot * 9 * 4096;
    const float *__restrict__ partial = partial_all + (size_t)blockIdx.y * units * total;
    float *__restrict__ dw = J.dw[blockIdx.y];
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cil = (int)(i & 63), col = (int)((i >> 6) & 63);
        long t = i >> 12;
        const int tap = (int)(t % 9);
        t /= 9;
        const int cot = (int)(t % n_cot), cit = (int)(t / n_cot);
        const int ci = cit * 64 + cil, co = cot * 64 + col;
        if (ci >= Cin || co >= Cout) continue;
        const float *p = partial + (((size_t)cit * n_cot + cot) * 9 + tap) * 4096 + col * 64 + cil;
        const size_t ustride = (size_t)n_cit * n_cot * 9 * 4096;
        float s = 0.f;
        int uu = 0;
        for (; uu + 3 < units; uu += 4) {
            const float a0 = p[(size_t)uu * ustride], a1 = p[(size_t)(uu + 1) * ustride], a2 = p[(size_t)(uu + 2) * ustride],
                        a3 = p[(size_t)(uu + 3) * ustride];
            s += a0, s += a1, s += a2, s += a3;
        }
        for (; uu < units; ++uu) s += p[(size_t)uu * ustride];
        float *d = dw + (size_t)co * stride_co + (size_t)ci * stride_ci + tap;
        *d = accumulate ? *d + s : s;
    }
}

}  // namespace

namespace {
void wgrad_plan(int N, int H, int W, int Cin, int Cout, int n_jobs, int &RG, int &n_qt, int &n_rg, int &n_cit, int &n_cot)
{
    n_qt = (W + WQV - 1) / WQV, n_cit = (Cin + 63) / 64, n_cot = (Cout + 63) / 64;
    // Rows per block.  One 512-thread block per CU: the launch runs in ceil(blocks / 256) rounds of (RG + fixed) row times --
    // the fixed part (three-row prologue, the 147 KB partial and its share of the reduction) is worth about 8 rows.  The split of
    // the image rows that minimises rounds x (RG + 8) is taken (a power-of-two RG with "at least 224 blocks" left most launches at
    // 257-447 blocks: two rounds, the second half empty -- 360 blocks x 64 rows where 240 x 80 do: 294 -> 190 us).
    const long base = (long)n_qt * N * n_cit * n_cot * n_jobs;
    long best = -1;
    n_rg = 1, RG = H;
    for (int k = 1; k <= H; ++k) {
        const int rg = (H + k - 1) / k;
        if (rg < 4 && k > 1) break;
        const int kk = (H + rg - 1) / rg;   // row groups this RG really makes
        const long rounds = (base * kk + 255) / 256, cost = rounds * (rg + 8);
        if (best < 0 || cost < best) best = cost, n_rg = kk, RG = rg;
    }
}

int wgrad_launch(const WgradJobs &J, int n_jobs, int ld_x, int Cin, int ld_g, int Cout, int64_t stride_co, int64_t stride_ci, int accumulate, int N,
                 int H, int W, void *workspace, int64_t workspace_bytes, int *range_flag, mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && ld_x >= Cin && ld_g >= Cout, "conv_wgrad3x3: N=%d H=%d W=%d Cin=%d/%d Cout=%d/%d", N,
                   H, W, Cin, ld_x, Cout, ld_g);
    MREFSR_REQUIRE(stride_co > 0 && stride_ci > 0, "conv_wgrad3x3: strides");
    int RG, n_qt, n_rg, n_cit, n_cot;
    wgrad_plan(N, H, W, Cin, Cout, n_jobs, RG, n_qt, n_rg, n_cit, n_cot);
    const long units = (long)n_qt * n_rg * N;
    MREFSR_REQUIRE(units * n_jobs < 0x7fffffffL && n_cit <= 65535 && n_cot <= 65535, "conv_wgrad3x3: grid too large");
    MREFSR_REQUIRE(workspace_bytes >= mrefsr_conv_wgrad3x3_batch_workspace_bytes(n_jobs, N, H, W, Cin, Cout), "conv_wgrad3x3: workspace too small");
    static unsigned long long attr_done = 0;
    if (mrefsr::first_use_on_device(attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv_wgrad3x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(conv_wgrad3x3_kernel, dim3((unsigned)(units * n_jobs), n_cit, n_cot), dim3(512), LDS_BYTES, st, J, (int)units, ld_x, Cin, ld_g, Cout,
                       (float *)workspace, H, W, n_qt, n_rg, RG, range_flag);
    const long total = (long)n_cit * n_cot * 9 * 4096;
    const long rb = (total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048;
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)(n_jobs > 8 ? (rb < 256 ? rb : 256) : rb), n_jobs), dim3(256), 0, st, (const float *)workspace, J,
                       (long)stride_co, (long)stride_ci, Cin, Cout, (int)units, n_cit, n_cot, accumulate);
    return mrefsr::check_launch("conv_wgrad3x3");
}
}  // namespace

MREFSR_EXPORT int64_t mrefsr_conv_wgrad3x3_batch_workspace_bytes(int n_jobs, int N, int H, int W, int Cin, int Cout)
{
    if (n_jobs <= 0 || n_jobs > MREFSR_WGRAD_MAX_JOBS || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return -1;
    int RG, n_qt, n_rg, n_cit, n_cot;
    wgrad_plan(N, H, W, Cin, Cout, n_jobs, RG, n_qt, n_rg, n_cit, n_cot);
    return (int64_t)n_jobs * n_qt * n_rg * N * n_cit * n_cot * 9 * 4096 * 4;
}

MREFSR_EXPORT int64_t mrefsr_conv_wgrad3x3_workspace_bytes(int N, int H, int W, int Cin, int Cout)
{
    return mrefsr_conv_wgrad3x3_batch_workspace_bytes(1, N, H, W, Cin, Cout);
}

MREFSR_EXPORT int mrefsr_conv_wgrad3x3_f32(const float *x, int ld_x, int Cin, const float *g, int ld_g, int Cout, float *dw, int64_t stride_co,
                                           int64_t stride_ci, int accumulate, const float *g_amax, int N, int H, int W, void *workspace,
                                           int64_t workspace_bytes, int *range_flag, mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(x && g && dw && g_amax && workspace, "conv_wgrad3x3: null pointer");
    WgradJobs J{};
    J.x[0] = x, J.g[0] = g, J.g_amax[0] = g_amax, J.dw[0] = dw;
    return wgrad_launch(J, 1, ld_x, Cin, ld_g, Cout, stride_co, stride_ci, accumulate, N, H, W, workspace, workspace_bytes, range_flag, stream);
}

// n_jobs weight gradients of ONE geometry (N, H, W, Cin, Cout, leading dimensions, dw strides) in one launch pair: x, g, g_amax,
// dw are HOST arrays of n_jobs device pointers (copied into the kernel arguments).  The blocks of all jobs fill the chip
// together, so each walks more rows and leaves fewer partials than n_jobs separate calls would.
MREFSR_EXPORT int mrefsr_conv_wgrad3x3_batch_f32(int n_jobs, const float *const *x, int ld_x, int Cin, const float *const *g, int ld_g, int Cout,
                                                 float *const *dw, int64_t stride_co, int64_t stride_ci, int accumulate, const float *const *g_amax,
                                                 int N, int H, int W, void *workspace, int64_t workspace_bytes, int *range_flag, mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(x && g && dw && g_amax && workspace, "conv_wgrad3x3_batch: null pointer");
    MREFSR_REQUIRE(n_jobs > 0 && n_jobs <= MREFSR_WGRAD_MAX_JOBS, "conv_wgrad3x3_batch: n_jobs=%d (1..%d)", n_jobs, MREFSR_WGRAD_MAX_JOBS);
    WgradJobs J{};
    for (int j = 0; j < n_jobs; ++j) {
        MREFSR_REQUIRE(x[j] && g[j] && dw[j] && g_amax[j], "conv_wgrad3x3_batch: null pointer in job %d", j);
        J.x[j] = x[j], J.g[j] = g[j], J.g_amax[j] = g_amax[j], J.dw[j] = dw[j];
    }
    return wgrad_launch(J, n_jobs, ld_x, Cin, ld_g, Cout, stride_co, stride_ci, accumulate, N, H, W, workspace, workspace_bytes, range_flag, stream);
}
