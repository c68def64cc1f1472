// Winograd F(2x2, 3x3) form of the 3x3 (pad 1, stride 1) convolutions of the restoration and VGG trunks (arch_util.py:89-117
// ResidualBlockNoBN, ref_mrapa_restoration_arch.py:217-225,271-304, vgg_arch.py) with the arithmetic of conv_nhwc.hip's terms-16
// mode: fp32-equivalent results from exact fp16 two-term operand splits, three v_mfma_f32_32x32x16_f16 per product.
//
// Why: the direct kernel keeps the matrix pipe 0.5-0.67 busy and the chip answers by dropping its clock to ~1.7 GHz (power):
// the lever left is matrix FLOP per output.  Y = A^T [ sum_c (G g G^T) . (B^T d B) ] A  turns the 36 multiply-adds of a 2 x 2
// output tile (per input / output channel pair) into 16: 2.25x fewer MFMAs for the same outputs.
//   V = B^T d B   of every 4 x 4 input patch (stride 2) is formed in fp32 and THEN split  V = vh + vl  (vl stored as fp16(vl 2^11)),
//   U = G g G^T   is formed at pack time (fp64), scaled by the layer's power of two S (conv_nhwc.hip: max|w| S in [2^13, 2^14),
//                 so |U| S < 2.25 * 2^14 stays inside fp16) and split  U S = uh + ul,  with  UH2 = uh 2^-11  derived in registers,
//   M[xi]        = sum_c  ul vh + UH2 VL + uh vh       (16 independent GEMMs, xi = (i, j) the position in the 4 x 4 transform)
//   Y = A^T M A / S,  then the epilogue of the direct kernel (bias, broadcast pre-activation term, LeakyReLU / PReLU, residual,
//                 NHWC store, MaxPool2d(2,2), PixelShuffle(2)).
// Accuracy: the split adds nothing to an fp32 Winograd transform (22-bit operands, fp32 accumulation); against fp64 the result
// is as close as an fp32 direct convolution's (tests/test_kernels_gpu.py::test_conv_wino_fp32_equivalent).  |V| <= 4 max|x|:
// the fp16 range guard fires at |x| > 16000.
//
// Block = 512 threads = 8 waves = 8 x 8 Winograd tiles (16 x 16 output pixels) x 64 couts, K walks 16-channel chunks:
//   1. the (18 x 18)-pixel patch of the chunk goes global -> registers (one chunk ahead) -> a raw fp32 tile in LDS laid out
//      [k half][row][4-channel piece parity][col] so that the transform's 16-byte reads are conflict-free;
//   2. transform: thread = (tile, 4-channel group, half of the j's): 12 ds_read_b128, 20 x 4 adds, 32 splits, 16 ds_write_b64 into
//      V[buffer][xi][plane vh | VL][k half][tile][8 ch] (double-buffered: chunk c + 1 is transformed while chunk c is multiplied;
//      waves 0-3 transform first, waves 4-7 multiply first: on every SIMD one wave's VALU work sits beside its partner's MFMAs);
//   3. multiply: wave (i = wave & 3, h = wave >> 2) owns xi = (i, 0..3) for all 64 tiles and couts [32 h, 32 h + 32): 4 x 2
//      accumulator tiles (128 VGPRs), per chunk 8 weight-fragment loads (L2), 16 ds_read_b128, 24 MFMAs.  A = weights (rows =
//      couts), B = V (columns = tiles): a lane ends up with 4 consecutive couts of a tile in consecutive registers;
//   4. output transform: each wave folds its four j's (columns of M) in registers, the four i's meet through LDS
//      ([i][b][tile][64 + 4 couts] fp32, 16-byte accesses both ways), then thread = (tile, 4 couts) forms the 2 x 2 pixels and runs
//      the epilogue: 256 contiguous bytes per pixel and 16 lanes.
#include <cstdlib>

#include "conv_common.h"

namespace {
using namespace mrefsr_conv;

constexpr int TT = 8;                 // Winograd tiles per side of a block's tile
constexpr int NTILE = TT * TT;        // 64 tiles = 16 x 16 output pixels
constexpr int PP = 2 * TT + 2;        // 18: side of the input patch
// raw fp32 patch of one 16-channel chunk, in 16-byte units: [k half][row PP][piece parity g0][col], col stride 1,
// g0 stride RAW_CS (odd), row stride RAW_RS (a multiple of 8: two rows = 0 mod 16 slots), k-half stride RAW_PLANE
constexpr int RAW_CS = 19, RAW_RS = 40, RAW_PLANE = PP * RAW_RS, RAW_BYTES = 2 * RAW_PLANE * 16;
// transformed, split tile: [buffer][xi 16][plane 2][k half 2][tile 64][8 ch] fp16
constexpr int V_KH = NTILE * 16, V_PL = 2 * V_KH, V_XI = 2 * V_PL, V_BUF = 16 * V_XI;
constexpr int X_LD = NB + 4;          // floats per tile row of the exchange buffer ([i 4][b 2][tile 64][X_LD])
constexpr int X_BYTES = 4 * 2 * NTILE * X_LD * 4;
constexpr int LDS_BYTES = 2 * V_BUF + RAW_BYTES + 16;   // + a 16-byte sink for the patch pieces a thread does not have
static_assert(X_BYTES <= LDS_BYTES, "conv_wino: the exchange buffer reuses the tile buffers");
static_assert(LDS_BYTES <= 160 * 1024, "conv_wino: LDS budget");
constexpr int NPF = (PP * PP * 4 + 511) / 512;   // 16-byte pieces of the patch per thread (3)
constexpr size_t WCH_HALVES = (size_t)16 * 2 * NB * KC;   // packed 16-bit values per (cout block, chunk): [xi][plane][cout 64][cin 16]

__device__ __forceinline__ f32x16 mma16(u32x4 a, u32x4 b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// 4 floats -> (vh, VL = fp16((v - vh) 2^11)) as 2 x 8 bytes
__device__ __forceinline__ void split_hl(const float a, const float b, const float c, const float d, u32x2 &hi, u32x2 &lo)
{
    const unsigned int p0 = pk_f16(a, b), p1 = pk_f16(c, d);
    hi = u32x2{p0, p1};
    const f32x2 h0 = un_f16(p0), h1 = un_f16(p1);
    lo = u32x2{pk_f16((a - h0[0]) * 2048.f, (b - h0[1]) * 2048.f), pk_f16((c - h1[0]) * 2048.f, (d - h1[1]) * 2048.f)};
}

// OIHW fp32 (strided view, optionally point-mirrored: conv_nhwc.hip's pack) -> U = G g G^T, scaled and split:
// [cout block][cin chunk][xi 16][plane uh | ul][cout 64][cin 16] fp16, zero padded
__global__ void wino_pack_kernel(const float *__restrict__ w, unsigned short *__restrict__ wp, int Cout, int Cin, int n_cb, int n_ch, float wscale,
                                 long so, long si, int flip, int *range_flag)
{
    const long total = (long)n_cb * n_ch * NB * KC;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(e % KC);
        long t = e / KC;
        const int co = (int)(t % NB);
        t /= NB;
        const int ch = (int)(t % n_ch), cb = (int)(t / n_ch);
        const int o = cb * NB + co, i = ch * KC + ci;
        double g[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int tap = a * 3 + b;
                g[a][b] = (o < Cout && i < Cin) ? (double)w[(size_t)o * so + (size_t)i * si + (flip ? 8 - tap : tap)] : 0.0;
            }
        // G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
        double gg[4][3];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            gg[0][b] = g[0][b];
            gg[1][b] = 0.5 * (g[0][b] + g[1][b] + g[2][b]);
            gg[2][b] = 0.5 * (g[0][b] - g[1][b] + g[2][b]);
            gg[3][b] = g[2][b];
        }
        const size_t base = ((size_t)cb * n_ch + ch) * WCH_HALVES + (size_t)co * KC + ci;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const double u4[4] = {gg[a][0], 0.5 * (gg[a][0] + gg[a][1] + gg[a][2]), 0.5 * (gg[a][0] - gg[a][1] + gg[a][2]), gg[a][2]};
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const float v = (float)(u4[b] * (double)wscale);
                if (range_flag && !(fabsf(v) <= 65000.f)) atomicOr(range_flag, 1);
                const unsigned int ph = pk_f16(v, 0.f);
                const float h = un_f16(ph)[0];
                const size_t at = base + (size_t)((a * 4 + b) * 2) * NB * KC;
                wp[at] = (unsigned short)(ph & 0xffffu);
                wp[at + (size_t)NB * KC] = (unsigned short)(pk_f16(v - h, 0.f) & 0xffffu);
            }
        }
    }
}

template <bool RES>
__global__ __launch_bounds__(512, 2) void conv_wino_kernel(const ConvArgs A)
{
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char *const vbuf = smem;
    unsigned char *const raw = smem + 2 * V_BUF;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (A.xcd_bands) {   // one contiguous eighth of the launch per XCD (conv_nhwc.hip)
        const unsigned gx = gridDim.x, gy = gridDim.y, lin = bx + gx * (by + gy * bz), per = (gx * gy * gridDim.z) / 8;
        if (lin < per * 8) {
            const unsigned l2 = (lin & 7) * per + (lin >> 3);
            bx = l2 % gx;
            const unsigned t2 = l2 / gx;
            by = t2 % gy, bz = t2 / gy;
        }
    }
    const int cb = bx % A.n_cb, n = bz;
    const int y0 = by * (2 * TT), x0 = (bx / A.n_cb) * (2 * TT);
    const int H = A.H, W = A.W;

    // multiply role: xi row `wi`, cout half `wh`
    const int wi = wv & 3, whf = wv >> 2;
    f32x16 acc[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][t][e] = 0.f;

    // ---- stage 1: global -> registers.  Piece idx = tid + 512 k: pixel idx >> 2 of the 18 x 18 patch, quarter idx & 3 of its 16
    // channels (4 lanes read the 64 contiguous bytes of a pixel)
    float4 pf[NPF];
    unsigned int praw[NPF];   // byte offset of the piece in the raw tile (RAW_BYTES, the sink: no such piece)
    unsigned int pgl[NPF];    // pixel index in an image, clamped into it
    unsigned int okmask = 0;  // bit k: piece k lies inside the image
#pragma unroll
    for (int k = 0; k < NPF; ++k) {
        const int idx = tid + k * 512, p = idx >> 2, q = idx & 3;
        const int py = p / PP, px = p - py * PP;
        const int gy = y0 + py - 1, gx = x0 + px - 1;
        const bool have = idx < PP * PP * 4;
        praw[k] = have ? (unsigned int)(((q >> 1) * RAW_PLANE + py * RAW_RS + (q & 1) * RAW_CS + px) * 16) : (unsigned int)RAW_BYTES;
        const bool ok = have && gy >= 0 && gy < H && gx >= 0 && gx < W;
        pgl[k] = ok ? (unsigned int)(gy * W + gx) : 0u;
        okmask |= ok ? (1u << k) : 0u;
    }
    const int q4 = (tid & 3) * 4;
    bool chunk_ok = true;
    auto fetch = [&](const int ch) {
        const bool first = ch < A.n_ch1;
        const int cl = (first ? ch * KC : (ch - A.n_ch1) * KC) + q4;
        const int Cs = first ? A.C1 : A.C2, ld = first ? A.ld1 : A.ld2;
        const float *xs = first ? A.x1 + (size_t)(n % A.N1) * H * W * A.ld1 : A.x2 + (size_t)(n % A.N2) * H * W * A.ld2;
        const int clc = cl < Cs ? cl : 0;   // (a ragged last chunk: clamped address, the piece is zeroed in raw_store)
        chunk_ok = cl < Cs;
#pragma unroll
        for (int k = 0; k < NPF; ++k) pf[k] = *reinterpret_cast<const float4 *>(xs + (size_t)pgl[k] * ld + clc);
    };
    unsigned int amax_bits = 0;   // fp16 range guard: largest |x| seen, as an IEEE bit pattern (Inf / NaN sort above every finite value)
    auto raw_store = [&]() {
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const bool ok = chunk_ok && ((okmask >> k) & 1u);
            float4 v;
            v.x = ok ? pf[k].x : 0.f, v.y = ok ? pf[k].y : 0.f, v.z = ok ? pf[k].z : 0.f, v.w = ok ? pf[k].w : 0.f;
            const unsigned int m01 = max(__float_as_uint(v.x) & 0x7fffffffu, __float_as_uint(v.y) & 0x7fffffffu);
            const unsigned int m23 = max(__float_as_uint(v.z) & 0x7fffffffu, __float_as_uint(v.w) & 0x7fffffffu);
            amax_bits = max(amax_bits, max(m01, m23));
            *reinterpret_cast<float4 *>(raw + praw[k]) = v;
        }
    };

    // ---- stage 2: transform role.  lane = (g0, tx, ty low 2 bits), wave = (jh, k half, ty high bit)
    const int t_g0 = lane & 1, t_tx = (lane >> 1) & 7, t_ty = (wv >> 2) * 4 + (lane >> 4);
    const int t_jh = wv & 1, t_kh = (wv >> 1) & 1;
    const unsigned char *const t_src = raw + (t_kh * RAW_PLANE + (2 * t_ty) * RAW_RS + t_g0 * RAW_CS + 2 * t_tx) * 16;
    // the wave's two columns of (B^T d) B from three tile columns (a, b, c):  v0 = a - c,  v1 = sg b + c
    //   jh 0: (x0, x1, x2), sg +1: j 0 = x0 - x2, j 1 = x1 + x2  |  jh 1: (x2, x3, x1), sg -1: j 2 = x2 - x1, j 3 = x1 - x3
    const int t_ca = (t_jh ? 2 : 0) * 16, t_cb = (t_jh ? 3 : 1) * 16, t_cc = (t_jh ? 1 : 2) * 16;
    const float t_sg = t_jh ? -1.f : 1.f;
    const unsigned int t_dst = (unsigned int)(t_kh * V_KH + (t_ty * TT + t_tx) * 16 + t_g0 * 8);
    auto transform = [&](const int buf) {
        unsigned char *const dst = vbuf + buf * V_BUF + t_dst;
        // one row i of B^T d at a time (i = 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3), then the two columns
        auto emit = [&](const int i, const float4 (&ra)[3], const float4 (&rb)[3], const bool add) {
            float q[3][4];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float4 a = ra[c], b = rb[c];
                if (add) q[c][0] = a.x + b.x, q[c][1] = a.y + b.y, q[c][2] = a.z + b.z, q[c][3] = a.w + b.w;
                else q[c][0] = a.x - b.x, q[c][1] = a.y - b.y, q[c][2] = a.z - b.z, q[c][3] = a.w - b.w;
            }
            float v0[4], v1[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v0[e] = q[0][e] - q[2][e], v1[e] = fmaf(t_sg, q[1][e], q[2][e]);
            u32x2 hi, lo;
            unsigned char *const dx = dst + (i * 4 + 2 * t_jh) * V_XI;
            split_hl(v0[0], v0[1], v0[2], v0[3], hi, lo);
            *reinterpret_cast<u32x2 *>(dx) = hi;
            *reinterpret_cast<u32x2 *>(dx + V_PL) = lo;
            split_hl(v1[0], v1[1], v1[2], v1[3], hi, lo);
            *reinterpret_cast<u32x2 *>(dx + V_XI) = hi;
            *reinterpret_cast<u32x2 *>(dx + V_XI + V_PL) = lo;
        };
        float4 d1[3], d2[3], dz[3];   // rows 1 and 2 serve three of the four combinations; rows 0 and 3 pass through dz
        auto row = [&](float4 (&d)[3], const int r) {
            d[0] = *reinterpret_cast<const float4 *>(t_src + r * RAW_RS * 16 + t_ca);
            d[1] = *reinterpret_cast<const float4 *>(t_src + r * RAW_RS * 16 + t_cb);
            d[2] = *reinterpret_cast<const float4 *>(t_src + r * RAW_RS * 16 + t_cc);
        };
        row(d1, 1);
        row(d2, 2);
        row(dz, 0);
        emit(1, d1, d2, true);
        emit(2, d2, d1, false);
        emit(0, dz, d2, false);
        row(dz, 3);
        emit(3, d1, dz, false);
    };

    // ---- stage 3: multiply role
    const unsigned short *const wcb = A.wp + (size_t)cb * A.n_ch * WCH_HALVES + (size_t)(whf * 32 + l31) * KC + kh * 8;
    const unsigned char *const m_src = vbuf + (wi * 4) * V_XI + kh * V_KH + l31 * 16;
    auto multiply = [&](const int ch, const int buf) {
        const unsigned short *wch = wcb + (size_t)ch * WCH_HALVES + (size_t)(wi * 4) * 2 * NB * KC;
        const unsigned char *src = m_src + buf * V_BUF;
        u32x4 uh = *reinterpret_cast<const u32x4 *>(wch), ul = *reinterpret_cast<const u32x4 *>(wch + (size_t)NB * KC);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u32x4 uhn = uh, uln = ul;
            if (j + 1 < 4) {   // the next j's fragments are requested before this j's MFMAs
                uhn = *reinterpret_cast<const u32x4 *>(wch + (size_t)((j + 1) * 2) * NB * KC);
                uln = *reinterpret_cast<const u32x4 *>(wch + (size_t)((j + 1) * 2 + 1) * NB * KC);
            }
            const u32x4 uh2 = scale_wh(uh);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const u32x4 vh = *reinterpret_cast<const u32x4 *>(src + j * V_XI + t * 512);
                const u32x4 vl = *reinterpret_cast<const u32x4 *>(src + j * V_XI + V_PL + t * 512);
                // partial products, smallest first (the order of conv_nhwc.hip's terms-16 mode)
                acc[j][t] = mma16(ul, vh, acc[j][t]);
                acc[j][t] = mma16(uh2, vl, acc[j][t]);
                acc[j][t] = mma16(uh, vh, acc[j][t]);
            }
            uh = uhn, ul = uln;
        }
    };

    // ---- the chunk pipeline: step s stores chunk s to the raw tile, requests chunk s + 1, transforms chunk s and multiplies chunk s - 1
    const int n_ch = A.n_ch;
    fetch(0);
    raw_store();
    if (1 < n_ch) fetch(1);
    __syncthreads();
    transform(0);
    __syncthreads();
    for (int s = 1; s < n_ch; ++s) {
        raw_store();
        if (s + 1 < n_ch) fetch(s + 1);
        __syncthreads();   // raw tile of chunk s complete
        // (the multiply sits outside every branch: the accumulators must not pass through a join)
        if (wv < 4) transform(s & 1);
        multiply(s - 1, (s - 1) & 1);
        if (wv >= 4) transform(s & 1);
        __syncthreads();   // V[s & 1] complete; the raw tile and V[(s - 1) & 1] are free
    }
    multiply(n_ch - 1, (n_ch - 1) & 1);
    __syncthreads();       // every wave is done with the tile buffers: the exchange buffer reuses them
    if (A.range_flag && amax_bits > __float_as_uint(16000.f)) atomicOr(A.range_flag, 1);

    // ---- stage 4: output transform.  Columns (j) in registers:  Z[b] = sum_j A^T[b][j] M[i][j]  =  b 0: m0 + m1 + m2,  1: m1 - m2 - m3;
    // accumulator register e of tile t holds cout 32 whf + 8 (e >> 2) + 4 kh + (e & 3) of Winograd tile 32 t + l31
    float *const xb = reinterpret_cast<float *>(smem);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float *const xrow = xb + ((size_t)(wi * 2) * NTILE + t * 32 + l31) * X_LD + whf * 32 + 4 * kh;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            float4 z0, z1;
            const int e = 4 * qd;
            z0.x = acc[0][t][e] + acc[1][t][e] + acc[2][t][e], z1.x = acc[1][t][e] - acc[2][t][e] - acc[3][t][e];
            z0.y = acc[0][t][e + 1] + acc[1][t][e + 1] + acc[2][t][e + 1], z1.y = acc[1][t][e + 1] - acc[2][t][e + 1] - acc[3][t][e + 1];
            z0.z = acc[0][t][e + 2] + acc[1][t][e + 2] + acc[2][t][e + 2], z1.z = acc[1][t][e + 2] - acc[2][t][e + 2] - acc[3][t][e + 2];
            z0.w = acc[0][t][e + 3] + acc[1][t][e + 3] + acc[2][t][e + 3], z1.w = acc[1][t][e + 3] - acc[2][t][e + 3] - acc[3][t][e + 3];
            *reinterpret_cast<float4 *>(xrow + 8 * qd) = z0;
            *reinterpret_cast<float4 *>(xrow + (size_t)NTILE * X_LD + 8 * qd) = z1;
        }
    }
    __syncthreads();

    // rows (i) from LDS, then the epilogue: thread = (Winograd tile T, 4 couts), two such items per thread
    const float slope = A.slope_ptr ? *A.slope_ptr : A.slope;
    const float oscale = A.out_scale;
    const int Cout = A.Cout;
    const int c4 = (tid & 15) * 4, co = cb * NB + c4;
    const bool cok = co < Cout;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (A.bias) {
        if (co + 0 < Cout) bv.x = A.bias[co + 0];
        if (co + 1 < Cout) bv.y = A.bias[co + 1];
        if (co + 2 < Cout) bv.z = A.bias[co + 2];
        if (co + 3 < Cout) bv.w = A.bias[co + 3];
    }
    const bool vec = (co + 3 < Cout) && ((A.ld_out & 3) == 0) && ((Cout & 3) == 0);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int T = (tid >> 4) + 32 * k, ty = T >> 3, tx = T & 7;
        const int gy0 = y0 + 2 * ty, gx0 = x0 + 2 * tx;
        float4 rq[2][2];
        if constexpr (RES) {   // residual of the tile's four pixels, requested before the LDS round (clamped addresses, no branch)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int gy = gy0 + a < H ? gy0 + a : H - 1, gx = gx0 + b < W ? gx0 + b : W - 1;
                    rq[a][b] = ld_f4(A.residual + (((size_t)n * H + gy) * W + gx) * A.ld_res + (cok ? co : 0), A.stream_out);
                }
        }
        float4 y[2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            float4 z[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) z[i] = *reinterpret_cast<const float4 *>(xb + ((size_t)(i * 2 + b) * NTILE + T) * X_LD + c4);
            y[0][b] = make_float4(z[0].x + z[1].x + z[2].x, z[0].y + z[1].y + z[2].y, z[0].z + z[1].z + z[2].z, z[0].w + z[1].w + z[2].w);
            y[1][b] = make_float4(z[1].x - z[2].x - z[3].x, z[1].y - z[2].y - z[3].y, z[1].z - z[2].z - z[3].z, z[1].w - z[2].w - z[3].w);
        }
        if (A.epilogue == 1) {   // MaxPool2d(2,2) of act(conv + bias) = act(max4 + bias): the tile IS the pooling window
            float4 v;
            v.x = fmaxf(fmaxf(y[0][0].x, y[0][1].x), fmaxf(y[1][0].x, y[1][1].x)) * oscale + bv.x;
            v.y = fmaxf(fmaxf(y[0][0].y, y[0][1].y), fmaxf(y[1][0].y, y[1][1].y)) * oscale + bv.y;
            v.z = fmaxf(fmaxf(y[0][0].z, y[0][1].z), fmaxf(y[1][0].z, y[1][1].z)) * oscale + bv.z;
            v.w = fmaxf(fmaxf(y[0][0].w, y[0][1].w), fmaxf(y[1][0].w, y[1][1].w)) * oscale + bv.w;
            if (A.act) {
                v.x = v.x > 0.f ? v.x : v.x * slope, v.y = v.y > 0.f ? v.y : v.y * slope;
                v.z = v.z > 0.f ? v.z : v.z * slope, v.w = v.w > 0.f ? v.w : v.w * slope;
            }
            const int Ho = H >> 1, Wo = W >> 1, py = gy0 >> 1, px = gx0 >> 1;
            if (cok && py < Ho && px < Wo) {
                float *o = A.out + (((size_t)n * Ho + py) * Wo + px) * A.ld_out + co;
                if (vec) {
                    st_f4(o, v, A.stream_out);
                } else {
                    o[0] = v.x;
                    if (co + 1 < Cout) o[1] = v.y;
                    if (co + 2 < Cout) o[2] = v.z;
                    if (co + 3 < Cout) o[3] = v.w;
                }
            }
            continue;
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int gy = gy0 + a, gx = gx0 + b;
                if (!(cok && gy < H && gx < W)) continue;
                const size_t pix = ((size_t)n * H + gy) * W + gx;
                float4 v = y[a][b];
                v.x = v.x * oscale + bv.x, v.y = v.y * oscale + bv.y, v.z = v.z * oscale + bv.z, v.w = v.w * oscale + bv.w;
                if (A.pre) {
                    const float *pp = A.pre + (((size_t)(n % A.pre_N) * H + gy) * W + gx) * Cout + co;
                    if (vec) {
                        const float4 t = *reinterpret_cast<const float4 *>(pp);
                        v.x += t.x, v.y += t.y, v.z += t.z, v.w += t.w;
                    } else {
                        v.x += pp[0];
                        if (co + 1 < Cout) v.y += pp[1];
                        if (co + 2 < Cout) v.z += pp[2];
                        if (co + 3 < Cout) v.w += pp[3];
                    }
                }
                if (A.act) {
                    v.x = v.x > 0.f ? v.x : v.x * slope, v.y = v.y > 0.f ? v.y : v.y * slope;
                    v.z = v.z > 0.f ? v.z : v.z * slope, v.w = v.w > 0.f ? v.w : v.w * slope;
                }
                if constexpr (RES) {
                    v.x += rq[a][b].x, v.y += rq[a][b].y, v.z += rq[a][b].z, v.w += rq[a][b].w;
                } else if (A.residual) {
                    const float *rp = A.residual + pix * A.ld_res + co;
                    if (vec && (A.ld_res & 3) == 0) {
                        const float4 t = ld_f4(rp, A.stream_out);
                        v.x += t.x, v.y += t.y, v.z += t.z, v.w += t.w;
                    } else {
                        v.x += rp[0];
                        if (co + 1 < Cout) v.y += rp[1];
                        if (co + 2 < Cout) v.z += rp[2];
                        if (co + 3 < Cout) v.w += rp[3];
                    }
                }
                if (A.epilogue == 2) {   // PixelShuffle(2): cout = 4c + 2i + j -> out[2y+i][2x+j][c]   (Cout % 4 == 0)
                    float *o = A.out + (((size_t)n * 2 * H + 2 * gy) * 2 * W + 2 * gx) * A.ld_out + (co >> 2);
                    o[0] = v.x;
                    o[A.ld_out] = v.y;
                    o[(size_t)2 * W * A.ld_out] = v.z;
                    o[(size_t)(2 * W + 1) * A.ld_out] = v.w;
                } else {
                    float *o = A.out + pix * A.ld_out + co;
                    if (vec) {
                        st_f4(o, v, A.stream_out);
                    } else {
                        o[0] = v.x;
                        if (co + 1 < Cout) o[1] = v.y;
                        if (co + 2 < Cout) o[2] = v.z;
                        if (co + 3 < Cout) o[3] = v.w;
                    }
                }
            }
    }
}

}  // namespace

namespace mrefsr_conv {

int64_t wino_packed_bytes(int Cout, int Cin)
{
    const long n_ch = (Cin + KC - 1) / KC, n_cb = (Cout + NB - 1) / NB;
    return n_cb * n_ch * (long)WCH_HALVES * 2;
}

int wino_pack(const float *weight, void *packed, int Cout, int Cin, float wscale, long stride_o, long stride_i, int flip, int *range_flag,
              hipStream_t stream)
{
    const int n_ch = (Cin + KC - 1) / KC, n_cb = (Cout + NB - 1) / NB;
    const long total = (long)n_cb * n_ch * NB * KC;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(wino_pack_kernel, dim3(blocks), dim3(256), 0, stream, weight, reinterpret_cast<unsigned short *>(packed), Cout, Cin, n_cb, n_ch,
                       wscale, stride_o, stride_i, flip, range_flag);
    return mrefsr::check_launch("conv_wino_pack");
}

// ConvArgs as conv_nhwc.hip's conv_entry fills them (terms 16 semantics: out_scale = 1 / wscale); epilogues 0 / 1 / 2
int wino_launch(const ConvArgs &a, int N, hipStream_t stream)
{
    if (a.in_amax || a.res_mask || a.stat_sum || a.io16 || a.epilogue == 3)
        return mrefsr::fail(MREFSR_E_UNSUPPORTED, "conv_wino: input scaling / training statistics / bf16 storage / DynAgg epilogue are the direct kernel's");
    static unsigned long long attr = 0;
    if (mrefsr::first_use_on_device(attr)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv_wino_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv_wino_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    }
    ConvArgs b = a;
    const int tw = 2 * TT;
    dim3 grid(((a.W + tw - 1) / tw) * a.n_cb, (a.H + tw - 1) / tw, N);
    b.stream_out = (size_t)N * a.H * a.W * a.ld_out * sizeof(float) > ((size_t)256 << 20);
    {
        const char *ex = getenv("MREFSR_CONV_XCD");
        b.xcd_bands = (ex ? ex[0] != '0' : 1) && (long)grid.x * grid.y * grid.z >= 128;
    }
    const bool res = a.residual && a.epilogue == 0 && (a.Cout & 3) == 0 && (a.ld_res & 3) == 0 && (a.ld_out & 3) == 0;
    if (res) hipLaunchKernelGGL((conv_wino_kernel<true>), grid, dim3(512), LDS_BYTES, stream, b);
    else hipLaunchKernelGGL((conv_wino_kernel<false>), grid, dim3(512), LDS_BYTES, stream, b);
    return mrefsr::check_launch("conv_wino");
}

}  // namespace mrefsr_conv
