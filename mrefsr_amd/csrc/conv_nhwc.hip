// 3x3 (pad 1) and 1x1 stride-1 convolutions of the restoration and VGG trunks (arch_util.py
// ResidualBlockNoBN, ref_mrapa_restoration_arch.py:139-348, vgg_arch.py, contras_multi_extractor_arch.py)
// as an implicit GEMM on the 16-bit matrix pipe with fp32-equivalent arithmetic.
//
// Arithmetic (the `terms` argument; ModeTraits below):
//   terms 16 (DEFAULT, MODE 2): every fp32 operand is split exactly into two fp16 terms -- a = ah + al with the remainder
//     stored as AL = fp16(al * 2^11); weights scaled per layer by a power of two S and split w S = wh + wl, plus the derived
//     plane WH2 = wh * 2^-11 -- and a product is the THREE partial products ah*wh + ah*wl + AL*WH2, each accumulated in fp32
//     by v_mfma_f32_32x32x16_f16 (22 significand bits per operand, dropped term 2^-22 relative; needs |a| < 65504: range flag);
//   terms 6 (MODE 0): three bf16 terms per operand, the SIX partial products >= 2^-24 (no range limit, 1.5x slower: the
//     automatic re-run path when the range flag fires);
//   terms 1 / 2 (MODE 3): bf16 arithmetic (BASELINE configs[4]), fp32 containers / bf16 tensors;  terms 3 (MODE 1): A/B builds only.
//
// Data layout: activations NHWC fp32 (channels-last), weights pre-packed once per layer into the
// B-fragment order  [cout block of 64][cin chunk of 16][tap 9][plane 3][cout 64][cin 16]  16-bit.
// Two kernels share the arithmetic, the epilogue (conv_epilogue) and, for every output, the accumulation order (bit-identical
// results):
//   conv_nhwc_kernel    256 threads = 4 waves compute 16 rows x 32 cols x 64 couts (4- / 8-row tiles for small launches); per cin
//                       chunk the (18 x 34)-pixel halo tile is split into LDS once ([plane][k half][pixel][8 ch] fp16: a fragment
//                       read is a conflict-free 16-byte-stride ds_read_b128), then each wave runs 9 taps x (4 row tiles x 2 cout
//                       tiles x 3 products) MFMAs reading A fragments from LDS and B fragments from the packed weights (L1 / L2);
//   conv_nhwc8_kernel   512 threads: TWO cout blocks share one split halo tile (Cout >= 128, chip-filling launches): half the
//                       input traffic and split work, a double-buffered LDS tile filled beside the partner wave's MFMAs, a
//                       three-tap weight-fragment ring with hand-counted s_waitcnt (see the kernel).
// The input may be the channel concatenation of two tensors (torch.cat of ref :217/:339/:346 never
// materialises), either of them broadcast over the batch (n % N1); the epilogue fuses + bias,
// + a broadcast pre-activation term, LeakyReLU / ReLU / PReLU(slope from device memory), + residual,
// and stores NHWC (128-byte segments), optionally through MaxPool2d(2,2) or PixelShuffle(2), or as the planar offset / mask
// tensors of a DynAgg (conv_offset_mask + its glue: epilogue 3).
#include <cstdlib>
#include <type_traits>

#include "conv_common.h"

namespace {
using namespace mrefsr_conv;

constexpr int TH = 16, TW = 32;
#ifndef MREFSR_CONV_T_EARLY
#define MREFSR_CONV_T_EARLY 6
#endif
constexpr int T_EARLY = MREFSR_CONV_T_EARLY;  // 3x3: taps [0, T_EARLY) fetch the B fragments of tap+1 ahead of their MFMAs
constexpr int EP_LD = NB + 8;                      // epilogue slab row stride (floats): conflict-free both ways
constexpr int EP_BYTES = 4 * 32 * EP_LD * 4;       // 4 waves x [32 px][EP_LD]
constexpr int DYN_T_LD = 36;                       // DynAgg epilogue: floats per channel row of a wave's transposing slab (32 px + 4)


// Arithmetic modes (the `terms` argument of the C ABI):
//   MODE 0 (terms 6):  bf16, activations and weights as hi + mid + lo, six partial products >= 2^-24
//   MODE 1 (terms 3):  bf16, hi + lo, three products (~2^-16 relative; experiments only)
//   MODE 2 (terms 16): fp16 (11-bit significands), two-term split = 22 bits, three products:
//        a = ah + al,  al stored as AL = fp16(al * 2^11)            (same binade as a: never denormal before a is)
//        w * S = wh + wl, S = 2^s chosen at pack time so that max|w| * S is in [2^13, 2^14)
//                         (wl ~ 2^-11 wh stays a normal fp16 for every weight within 2^-17 of the largest);
//                         a third plane WH2 = wh * 2^-11 (exact) pairs with AL
//        a * w * S  ~=  ah*wh + ah*wl + AL*WH2        (dropped al*wl: 2^-22 relative), fp32 accumulation,
//        the epilogue multiplies by 1/S.  Requires |a| < 65504 (fp16 range).
//   MODE 3 (terms 1):  bf16 ARITHMETIC (BASELINE configs[4]): activations and weights rounded to bf16 (one plane
//        each, one MFMA per product), fp32 accumulation, the result rounded to bf16 again (kept in an fp32 container)
template <int MODE> struct ModeTraits;
template <> struct ModeTraits<0> { static constexpr int NA = 3, NW = 3, NT = 6; };
template <> struct ModeTraits<1> { static constexpr int NA = 2, NW = 2, NT = 3; };
template <> struct ModeTraits<2> { static constexpr int NA = 2, NW = 3, NT = 3; };
template <> struct ModeTraits<3> { static constexpr int NA = 1, NW = 1, NT = 1; };
// (activation plane, weight plane) of each partial product, smallest first
__device__ constexpr int TERM_A[4][6] = {{1, 2, 0, 1, 0, 0}, {1, 0, 0, 0, 0, 0}, {0, 1, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}};
__device__ constexpr int TERM_W[4][6] = {{1, 0, 2, 0, 1, 0}, {0, 1, 0, 0, 0, 0}, {1, 2, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}};

// 4 floats -> activation planes of 4 x 16-bit (8 bytes each)
template <int MODE>
__device__ __forceinline__ void split4(const float4 v, u32x2 *out)
{
    float a = v.x, b = v.y, c = v.z, d = v.w;
    if (MODE == 2) {
        const unsigned int p0 = pk_f16(a, b), p1 = pk_f16(c, d);
        out[0] = u32x2{p0, p1};
        const f32x2 h0 = un_f16(p0), h1 = un_f16(p1);
        out[1] = u32x2{pk_f16((a - h0[0]) * 2048.f, (b - h0[1]) * 2048.f), pk_f16((c - h1[0]) * 2048.f, (d - h1[1]) * 2048.f)};
        return;
    }
    constexpr int NS = ModeTraits<MODE>::NA;
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

__device__ __forceinline__ void round4_bf16(float4 &v)
{
    const unsigned int p0 = pk_bf16(v.x, v.y), p1 = pk_bf16(v.z, v.w);
    v.x = bf_lo(p0), v.y = bf_hi(p0), v.z = bf_lo(p1), v.w = bf_hi(p1);
}


template <int MODE>
__device__ __forceinline__ f32x16 mma(u32x4 a, u32x4 b, f32x16 c)
{
    if (MODE == 2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// OIHW fp32 -> [cout block][cin chunk][tap][split][64 cout][16 cin] bf16 (zero padded)
template <int MODE>
__device__ __forceinline__ void conv_pack_elements(const float *__restrict__ w, unsigned short *__restrict__ wp, int Cout, int Cin, int taps,
                                                   int n_cb, int n_ch, float wscale, long so, long si, int flip, int *range_flag = nullptr)
{
    constexpr int NS = ModeTraits<MODE>::NW;
    const long total = (long)n_cb * n_ch * taps * NB * KC;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(e % KC);
        long t = e / KC;
        const int co = (int)(t % NB);
        t /= NB;
        const int tap = (int)(t % taps);
        t /= taps;
        const int ch = (int)(t % n_ch), cb = (int)(t / n_ch);
        const int o = cb * NB + co, i = ch * KC + ci;
        float v = (o < Cout && i < Cin) ? w[(size_t)o * so + (size_t)i * si + (flip ? taps - 1 - tap : tap)] : 0.f;
        const size_t base = ((((size_t)cb * n_ch + ch) * taps + tap) * NS) * NB * KC + (size_t)co * KC + ci;
        if (MODE == 2) {
            v *= wscale;
            if (range_flag && !(fabsf(v) <= 65000.f)) atomicOr(range_flag, 1);   // a weight that has outgrown its cached power-of-two scale (or is not finite)
            const unsigned int ph = pk_f16(v, 0.f);
            const float h = un_f16(ph)[0];
            wp[base] = (unsigned short)(ph & 0xffffu);
            wp[base + (size_t)NB * KC] = (unsigned short)(pk_f16(v - h, 0.f) & 0xffffu);
            wp[base + (size_t)2 * NB * KC] = (unsigned short)(pk_f16(h * (1.0f / 2048.f), 0.f) & 0xffffu);
            continue;
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const unsigned int p = pk_bf16(v, 0.f);
            wp[base + (size_t)s * NB * KC] = (unsigned short)(p & 0xffffu);
            v -= bf_lo(p);
        }
    }
}

template <int MODE>
__global__ void conv_pack_kernel(const float *__restrict__ w, unsigned short *__restrict__ wp, int Cout, int Cin, int taps,
                                 int n_cb, int n_ch, float wscale, long so, long si, int flip)
{
    conv_pack_elements<MODE>(w, wp, Cout, Cin, taps, n_cb, n_ch, wscale, so, si, flip);
}

// every weight of a training step in ONE launch: blockIdx.y picks the job (a table in device memory), blockIdx.x strides over
// its elements exactly like the single-tensor kernel (346 launches of 4 us per step before)
__global__ void conv_pack_multi_kernel(const mrefsr_conv_pack_job *__restrict__ jobs, int *__restrict__ range_flag)
{
    const mrefsr_conv_pack_job j = jobs[blockIdx.y];
    const int taps = j.ksize * j.ksize, n_ch = (j.Cin + KC - 1) / KC, n_cb = (j.Cout + NB - 1) / NB;
    unsigned short *wp = reinterpret_cast<unsigned short *>(j.packed);
    const long so = (long)j.stride_o, si = (long)j.stride_i;
    switch (j.terms) {
    case 6: conv_pack_elements<0>(j.weight, wp, j.Cout, j.Cin, taps, n_cb, n_ch, 1.f, so, si, j.flip); break;
    case 3: conv_pack_elements<1>(j.weight, wp, j.Cout, j.Cin, taps, n_cb, n_ch, 1.f, so, si, j.flip); break;
    case 1: conv_pack_elements<3>(j.weight, wp, j.Cout, j.Cin, taps, n_cb, n_ch, 1.f, so, si, j.flip); break;
    default: conv_pack_elements<2>(j.weight, wp, j.Cout, j.Cin, taps, n_cb, n_ch, j.wscale, so, si, j.flip, range_flag); break;
    }
}

#ifndef MREFSR_CONV_XCD_DEFAULT
#define MREFSR_CONV_XCD_DEFAULT 1
#endif


// 4 consecutive bf16 <-> float4 (8-byte accesses)
__device__ __forceinline__ float4 ld_bf16x4(const void *p)
{
    const u32x2 r = *reinterpret_cast<const u32x2 *>(p);
    return make_float4(bf_lo(r[0]), bf_hi(r[0]), bf_lo(r[1]), bf_hi(r[1]));
}
__device__ __forceinline__ void st_bf16x4(void *p, const float4 v) { *reinterpret_cast<u32x2 *>(p) = u32x2{pk_bf16(v.x, v.y), pk_bf16(v.z, v.w)}; }
__device__ __forceinline__ float ld_bf16(const void *p) { return __uint_as_float((unsigned int)*reinterpret_cast<const unsigned short *>(p) << 16); }
__device__ __forceinline__ void st_bf16(void *p, const float v) { *reinterpret_cast<unsigned short *>(p) = (unsigned short)(pk_bf16(v, 0.f) & 0xffffu); }

#ifdef MREFSR_CONV_STAMP
// instrumentation build (tools/conv_stamp.py): shader-clock totals per phase of a wave's life, summed over all waves
//   0 prologue | 1 barrier before the fill | 2 wait for the halo prefetch | 3 split + LDS stores | 4 barrier after the fill |
//   5 taps (MFMA phase) | 6 epilogue | 7 waves
__device__ unsigned long long g_conv_stamp[1024][8];   // 1024 slots: no hot spot from the waves' final adds
#define STAMP(i)                                                      \
    {                                                                 \
        const unsigned long long t_now = __builtin_readcyclecounter(); \
        st_acc[i] += t_now - t_last;                                  \
        t_last = t_now;                                               \
    }
#else
#define STAMP(i)
#endif

// max |out| of a launch (ConvArgs::stat_amax as the forward launches' `out_amax`: the input scale of the Winograd layer that reads
// this tensor next, archs/nhwc.py): a lane's running maximum over the values it stores, one wave reduction and one atomic per wave
__device__ __forceinline__ float amax4(float m, const float4 &v, int co, int Cout)
{
    if (co + 3 < Cout) return fmaxf(fmaxf(fmaxf(m, fabsf(v.x)), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));   // (two v_max3 with |.| modifiers)
    m = fmaxf(m, fabsf(v.x));
    if (co + 1 < Cout) m = fmaxf(m, fabsf(v.y));
    if (co + 2 < Cout) m = fmaxf(m, fabsf(v.z));
    return m;
}
__device__ __forceinline__ void publish_amax(unsigned int *dst, float m)
{
    if (!dst) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    // (one plain read first: after the first blocks most waves' maxima are below the word already -- tens of thousands of atomics
    //  on ONE address cost the bandwidth-bound launches 0.3 ms)
    if ((threadIdx.x & 63) == 0 && m > 0.f && m < 3.0e38f && __float_as_uint(m) > __builtin_nontemporal_load(dst)) atomicMax(dst, __float_as_uint(m));
}

// The epilogue of both convolution kernels.  `acc`: the wave's RPW x 2 accumulator tiles (rows wrow * RPW .. + RPW of the block's
// tile, couts cb * 64 .. + 64); `wslab`: which 32 x EP_LD-float slab of `smem` the wave turns its rows through; NTHR threads
// per block (the pre-offset staging of epilogue 3 is a block-wide loop).
template <int MODE, bool IO16, bool RES, int RPW, int NTHR>
__device__ __forceinline__ void conv_epilogue(const ConvArgs &A, f32x16 (&acc)[RPW][2], unsigned char *smem, const int tid, const int wslab,
                                              const int wrow, const int cb, const int n, const int y0, const int x0, const float oscale)
{
    const int lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    const int H = A.H, W = A.W;
    // ---- epilogue.  MFMA result: lane holds cout (j*32 + l31) for pixels x = (e&3) + 8*(e>>2) + 4*kh of
    // row m.  Each wave turns one row at a time through its own LDS slab ([32 px][64 + 8 cout] fp32) so
    // that a lane owns 4 consecutive couts of a pixel: bias / pre / residual / out move as 16-byte
    // vectors, 256 contiguous bytes per pixel.
    __syncthreads();  // every wave is done reading the input tile
    float *slab = reinterpret_cast<float *>(smem) + wslab * (32 * EP_LD);
    const float slope = A.slope_ptr ? *A.slope_ptr : A.slope;
    const int Cout = A.Cout;
    const int c4 = (lane & 15) * 4, co = cb * NB + c4, psub = lane >> 4;
    const bool cok = co < Cout;  // Cout % 4 == 0 is not required: the tail is handled per element
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (A.bias) {
        if (co + 0 < Cout) bv.x = A.bias[co + 0];
        if (co + 1 < Cout) bv.y = A.bias[co + 1];
        if (co + 2 < Cout) bv.z = A.bias[co + 2];
        if (co + 3 < Cout) bv.w = A.bias[co + 3];
    }
    const bool vec = (co + 3 < Cout) && ((A.ld_out & 3) == 0) && ((Cout & 3) == 0);

    if constexpr (RPW == 4) {   // (the DynAgg and max-pool epilogues exist for 16-row tiles only: launch())
    if (A.epilogue == 3) {
        // conv_offset_mask + the DynAgg glue in one pass: channel c < 2 n_i is offset component (c & 1 ? x : y) of (group, tap) =
        // (c / 18, (c / 2) % 9) and gets the pre-computed correspondence offset of that tap added, channel 2 n_i + i is mask i
        // and goes through the sigmoid; both are written PLANAR ([n][channel][H][W]: what the DCN gather reads), and
        // sum |learned offset| is accumulated for the reference's "offset mean > 100" guard.  No slab: the accumulator layout
        // already is "lane = channel, 4 consecutive registers = 4 consecutive pixels", i.e. one 16-byte store per lane into
        // its channel's plane; the tile's pre-offsets are staged once in LDS as [tap][x|y][16 rows][32 px] planes.
        constexpr int PLD = TH * TW + 4;   // plane stride (floats): + 4 spreads the 18 planes over the banks
        float *pre_t = reinterpret_cast<float *>(smem);
        // W % 4 == 0: a wave turns each (row, 32-channel half) through a [32 channels][32 px (+4)] slab behind the pre-offset tile, so
        // that 8 lanes write the 128 contiguous bytes of one plane row (8 planes x 128 B per store instruction; straight from the
        // accumulator layout it was 32 planes x 32 B -- the 14 GB of planar output of the 640^2 launch left in quarter lines)
        float *tsl = pre_t + 18 * PLD + wslab * (32 * DYN_T_LD);
        const int n_i = A.dyn_ni, n_off = 2 * n_i;
        const size_t HW = (size_t)H * W;
        for (int i = tid; i < 9 * TH * TW; i += NTHR) {
            const int px = i & (TW - 1), r = (i / TW) % TH, tap = i / (TH * TW);
            const int gy = y0 + r, gx = x0 + px;
            float2 pr = make_float2(0.f, 0.f);
            if (gy < H && gx < W) pr = A.dyn_pre[((size_t)n * 9 + tap) * HW + (size_t)gy * W + gx];   // [x, y]
            pre_t[(tap * 2 + 0) * PLD + r * TW + px] = pr.x;
            pre_t[(tap * 2 + 1) * PLD + r * TW + px] = pr.y;
        }
        __syncthreads();
        const bool vec4 = (W & 3) == 0;
        float local = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (cb * NB + j * 32 >= Cout) continue;              // (wave-uniform: no channel of this half exists)
            const int c_raw = cb * NB + j * 32 + l31;
            const bool cval = c_raw < Cout;                      // lanes beyond the last channel still take part in the slab turns
            if (!vec4 && !cval) continue;
            const int c = cval ? c_raw : Cout - 1;
            const float bc = A.bias ? A.bias[c] : 0.f;
            const bool is_off = c < n_off;
            const float *pp = pre_t + (((c >> 1) % 9) * 2 + ((c & 1) ? 0 : 1)) * PLD;   // odd channel: x, even: y
            float *plane = is_off ? A.out + ((size_t)n * n_off + c) * HW : A.dyn_mask + ((size_t)n * n_i + (c - n_off)) * HW;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int r = wrow * 4 + m, gy = y0 + r;
                if (gy >= H) continue;
                if (vec4) __builtin_amdgcn_wave_barrier();   // (the slab's previous readers are done)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int px = 8 * q + 4 * kh, gx = x0 + px;
                    if (gx >= W) continue;
                    float v[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        v[k] = acc[m][j][4 * q + k];
                        if (MODE == 2) v[k] *= oscale;
                        v[k] += bc;
                        if (MODE == 3) v[k] = bf_lo(pk_bf16(v[k], 0.f));   // bf16 arithmetic: the layer output is a bf16 value
                    }
                    if (is_off) {
                        const float4 pr = *reinterpret_cast<const float4 *>(pp + r * TW + px);
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if ((vec4 || gx + k < W) && cval) local += fabsf(v[k]);   // (pixels beyond a ragged right edge are not part of the map)
                        v[0] += pr.x, v[1] += pr.y, v[2] += pr.z, v[3] += pr.w;
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] = 1.0f / (1.0f + expf(-v[k]));
                    }
                    float *o = plane + (size_t)gy * W + gx;
                    if (vec4) {   // W % 4 == 0: gx + 3 < W; through the slab (below)
                        *reinterpret_cast<float4 *>(tsl + l31 * DYN_T_LD + px) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (gx + k < W) o[k] = v[k];
                    }
                }
                if (vec4) {
                    __builtin_amdgcn_wave_barrier();
                    const int px4 = (lane & 7) * 4, gx4 = x0 + px4;
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int rc = (lane >> 3) + 8 * it, c2 = cb * NB + j * 32 + rc;
                        const float4 t = *reinterpret_cast<const float4 *>(tsl + rc * DYN_T_LD + px4);
                        if (c2 < Cout && gx4 < W) {
                            float *pl2 = c2 < n_off ? A.out + ((size_t)n * n_off + c2) * HW : A.dyn_mask + ((size_t)n * n_i + (c2 - n_off)) * HW;
                            *reinterpret_cast<float4 *>(pl2 + (size_t)gy * W + gx4) = t;
                        }
                    }
                }
            }
        }
        if (A.dyn_abs) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o, 64);
            if (lane == 0 && local != 0.f) atomicAdd(A.dyn_abs, (double)local);
        }
        return;
    }

    if (A.epilogue == 1) {  // MaxPool2d(2,2) of act(conv + bias) = act(max4 + bias): both monotone
        const int Ho = H >> 1, Wo = W >> 1;
        float pamx = 0.f;
#pragma unroll
        for (int m = 0; m < 4; m += 2) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const int xh = (((e & 3) + 8 * (e >> 2)) >> 1) + 2 * kh;  // pooled column 0..15
                    slab[xh * EP_LD + j * 32 + l31] =
                        fmaxf(fmaxf(acc[m][j][e], acc[m][j][e + 1]), fmaxf(acc[m + 1][j][e], acc[m + 1][j][e + 1]));
                }
            __builtin_amdgcn_wave_barrier();
            const int gy = (y0 + wrow * 4 + m) >> 1;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int px = it * 4 + psub, gx = (x0 >> 1) + px;
                float4 v = *reinterpret_cast<const float4 *>(slab + px * EP_LD + c4);
                if (MODE == 2) v.x *= oscale, v.y *= oscale, v.z *= oscale, v.w *= oscale;
                v.x += bv.x, v.y += bv.y, v.z += bv.z, v.w += bv.w;
                if (A.act) {
                    v.x = v.x > 0.f ? v.x : v.x * slope, v.y = v.y > 0.f ? v.y : v.y * slope;
                    v.z = v.z > 0.f ? v.z : v.z * slope, v.w = v.w > 0.f ? v.w : v.w * slope;
                }
                if (MODE == 3) round4_bf16(v);
                if (MODE == 2 && !IO16 && cok && gy < Ho && gx < Wo) pamx = amax4(pamx, v, co, Cout);
                if (IO16) {
                    if (cok && gy < Ho && gx < Wo) {
                        unsigned short *o = reinterpret_cast<unsigned short *>(A.out) + (((size_t)n * Ho + gy) * Wo + gx) * A.ld_out + co;
                        if (vec) {
                            st_bf16x4(o, v);
                        } else {
                            st_bf16(o, v.x);
                            if (co + 1 < Cout) st_bf16(o + 1, v.y);
                            if (co + 2 < Cout) st_bf16(o + 2, v.z);
                            if (co + 3 < Cout) st_bf16(o + 3, v.w);
                        }
                    }
                } else if (cok && gy < Ho && gx < Wo) {
                    float *o = A.out + (((size_t)n * Ho + gy) * Wo + gx) * A.ld_out + co;
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
            __builtin_amdgcn_wave_barrier();
        }
        if (MODE == 2 && !IO16) publish_amax(A.stat_amax, pamx);
        return;
    }
    }
    // RES (fp32 residual, Cout and ld_res multiples of 4; chosen by launch()): a residual row is requested whole (8 x 16 B
    // per lane), from clamped addresses and without a branch, BEFORE the row's accumulators go through the slab.  Inside
    // the per-pixel `if` below every load was its own basic block with its own vmcnt(0): 32 serial HBM round trips per
    // wave, 44 % of a wave's life in the 64 -> 64 trunk layers (tools/conv_stamp.py).  A separate instantiation, because the
    // 32 extra registers cost the layers without a residual 2-3 %.
    constexpr bool res_fast = RES;
    float4 ssum = make_float4(0.f, 0.f, 0.f, 0.f);   // A.stat_sum: this lane's four couts over its pixels
    float samx = 0.f;
#pragma unroll
    for (int m = 0; m < RPW; ++m) {
        const int gy = y0 + wrow * RPW + m;
        float4 rq[8];
        if constexpr (res_fast) {
            const float *rrow = A.residual + ((size_t)n * H + (gy < H ? gy : H - 1)) * W * A.ld_res + (cok ? co : 0);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int gx = x0 + it * 4 + psub;
                rq[it] = ld_f4(rrow + (size_t)(gx < W ? gx : W - 1) * A.ld_res, A.stream_out);
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) slab[((e & 3) + 8 * (e >> 2) + 4 * kh) * EP_LD + j * 32 + l31] = acc[m][j][e];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int px = it * 4 + psub, gx = x0 + px;
            float4 v = *reinterpret_cast<const float4 *>(slab + px * EP_LD + c4);
            if (cok && gy < H && gx < W) {
                const size_t pix = ((size_t)n * H + gy) * W + gx;
                if (MODE == 2) v.x *= oscale, v.y *= oscale, v.z *= oscale, v.w *= oscale;
                v.x += bv.x, v.y += bv.y, v.z += bv.z, v.w += bv.w;
                if (A.pre && IO16) {
                    const unsigned short *pp = reinterpret_cast<const unsigned short *>(A.pre) + (((size_t)(n % A.pre_N) * H + gy) * W + gx) * Cout + co;
                    if (vec) {
                        const float4 t = ld_bf16x4(pp);
                        v.x += t.x, v.y += t.y, v.z += t.z, v.w += t.w;
                    } else {
                        v.x += ld_bf16(pp);
                        if (co + 1 < Cout) v.y += ld_bf16(pp + 1);
                        if (co + 2 < Cout) v.z += ld_bf16(pp + 2);
                        if (co + 3 < Cout) v.w += ld_bf16(pp + 3);
                    }
                } else if (A.pre) {
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
                if (A.residual && IO16) {
                    const unsigned short *rp = reinterpret_cast<const unsigned short *>(A.residual) + pix * A.ld_res + co;
                    if (vec && (A.ld_res & 3) == 0) {
                        const float4 t = ld_bf16x4(rp);
                        v.x += t.x, v.y += t.y, v.z += t.z, v.w += t.w;
                    } else {
                        v.x += ld_bf16(rp);
                        if (co + 1 < Cout) v.y += ld_bf16(rp + 1);
                        if (co + 2 < Cout) v.z += ld_bf16(rp + 2);
                        if (co + 3 < Cout) v.w += ld_bf16(rp + 3);
                    }
                } else if constexpr (res_fast) {
                    if (A.res_mask) {
                        v.x = rq[it].x > 0.f ? v.x : 0.f, v.y = rq[it].y > 0.f ? v.y : 0.f;
                        v.z = rq[it].z > 0.f ? v.z : 0.f, v.w = rq[it].w > 0.f ? v.w : 0.f;
                    } else {
                        v.x += rq[it].x, v.y += rq[it].y, v.z += rq[it].z, v.w += rq[it].w;
                    }
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
                if (MODE == 3) round4_bf16(v);
                if (MODE == 2 && !IO16 && A.stat_sum) {   // (launch() takes this path only with Cout % 4 == 0)
                    ssum.x += v.x, ssum.y += v.y, ssum.z += v.z, ssum.w += v.w;
                    samx = fmaxf(fmaxf(samx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                } else if (MODE == 2 && !IO16) {
                    samx = amax4(samx, v, co, Cout);   // (out_amax of a forward launch: the next layer's input scale)
                }
                if (IO16) {
                    unsigned short *oh = reinterpret_cast<unsigned short *>(A.out);
                    if (A.epilogue == 2) {
                        unsigned short *o = oh + (((size_t)n * 2 * H + 2 * gy) * 2 * W + 2 * gx) * A.ld_out + (co >> 2);
                        st_bf16(o, v.x);
                        st_bf16(o + A.ld_out, v.y);
                        st_bf16(o + (size_t)2 * W * A.ld_out, v.z);
                        st_bf16(o + (size_t)(2 * W + 1) * A.ld_out, v.w);
                    } else {
                        unsigned short *o = oh + pix * A.ld_out + co;
                        if (vec) {
                            st_bf16x4(o, v);
                        } else {
                            st_bf16(o, v.x);
                            if (co + 1 < Cout) st_bf16(o + 1, v.y);
                            if (co + 2 < Cout) st_bf16(o + 2, v.z);
                            if (co + 3 < Cout) st_bf16(o + 3, v.w);
                        }
                    }
                } else if (A.epilogue == 2) {  // PixelShuffle(2): cout = 4c + 2i + j -> out[2y+i][2x+j][c]   (Cout % 4 == 0)
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
        __builtin_amdgcn_wave_barrier();
    }
    if (MODE == 2 && !IO16 && A.stat_sum) {
        // the four lanes that share a cout quad (lane, +16, +32, +48), then the block's waves through LDS, then one float atomic
        // per cout and block -- what mrefsr_act_bwd_nhwc_f32 would compute in a pass of its own over this launch's output
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            ssum.x += __shfl_xor(ssum.x, o, 64), ssum.y += __shfl_xor(ssum.y, o, 64);
            ssum.z += __shfl_xor(ssum.z, o, 64), ssum.w += __shfl_xor(ssum.w, o, 64);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) samx = fmaxf(samx, __shfl_xor(samx, o, 64));
        constexpr int NWV = NTHR / 64;
        float *red = reinterpret_cast<float *>(smem);   // [NWV][64] sums, then [NWV] maxima
        __syncthreads();                                 // every wave is done with its slab
        const int wv = tid >> 6;
        if (lane < 16) *reinterpret_cast<float4 *>(red + wv * 64 + c4) = ssum;
        if (lane == 0) red[NWV * 64 + wv] = samx;
        __syncthreads();
        if (tid < 64 && cb * NB + tid < Cout) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NWV; ++w) t += red[w * 64 + tid];
            atomicAdd(A.stat_sum + cb * NB + tid, t);
        }
        if (tid == 0 && A.stat_amax) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NWV; ++w) t = fmaxf(t, red[NWV * 64 + w]);
            if (t > 0.f && t < 3.0e38f) atomicMax(A.stat_amax, __float_as_uint(t));
        }
    } else if (MODE == 2 && !IO16) {
        publish_amax(A.stat_amax, samx);
    }
}

// RPW = output rows per wave: 4 (block = 16 x 32 pixels, the throughput shape) or 1 / 2 (4 x 32 / 8 x 32 pixels: a quarter /
// half of the serial work per block, for launches that cannot fill the chip with 16-row tiles -- the 40^2 .. 160^2 maps of the
// training step and of single-image inference, where a launch lasted one block's lifetime whatever its size)
template <int MODE, int KS, bool IO16 = false, bool RES = false, int RPW = 4>
__global__ __launch_bounds__(256, RPW == 4 ? 2 : (RPW == 2 ? 3 : 4)) void conv_nhwc_kernel(const ConvArgs A)
{
    constexpr int THB = 4 * RPW;
#ifdef MREFSR_CONV_STAMP
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last = __builtin_readcyclecounter();
#endif
    constexpr int NS = ModeTraits<MODE>::NA, NW = ModeTraits<MODE>::NW, NT = ModeTraits<MODE>::NT;
#ifdef MREFSR_CONV_LOAD_WH2
    constexpr int NWL = NW;
#else
    constexpr int NWL = MODE == 2 ? 2 : NW;  // weight planes loaded; MODE 2 derives WH2 from wh in registers
#endif
    constexpr int HALO = KS / 2, PH = THB + 2 * HALO, PW = TW + 2 * HALO, NPIX = PH * PW, TAPS = KS * KS;
    constexpr int PLANE = NPIX * KC * 2;  // bytes per split plane
    // a plane is two half-planes [k half][pixel][8 channels]: lane (pixel l31, k half kh) of an A fragment reads 16 bytes at
    // kh * HPLANE + pixel * 16 -- 16-byte stride over the 32 pixels of a half-wave = the one stride `ds_read_b128` serves without
    // bank conflicts (with [pixel][16 channels], 32-byte stride, every read was a 2-way conflict: SQ_LDS_BANK_CONFLICT)
    constexpr int HPLANE = NPIX * 16;
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    // cout block fastest: the n_cb blocks that share an input tile are dispatched together (its halo tile
    // is fetched from HBM once and found in L2 by the others)
    // Workgroups go to the 8 XCDs round-robin in dispatch order (x fastest): with the plain mapping the tiles an XCD's L2 sees
    // are every 8th of a row of tiles -- no two of them share a halo column or row.  xcd_bands re-labels the blocks so that each
    // XCD walks one contiguous eighth of the launch (bands of tile rows of one image): neighbouring tiles meet in one L2.
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (A.xcd_bands) {
        const unsigned gx = gridDim.x, gy = gridDim.y, lin = bx + gx * (by + gy * bz), per = (gx * gy * gridDim.z) / 8;
        if (lin < per * 8) {
            const unsigned l2 = (lin & 7) * per + (lin >> 3);
            bx = l2 % gx;
            const unsigned t2 = l2 / gx;
            by = t2 % gy, bz = t2 / gy;
        }
    }
    const int cb = bx % A.n_cb, n = bz;
    const int y0 = by * THB, x0 = (bx / A.n_cb) * TW;
    const int H = A.H, W = A.W;
    float in_s = 1.f, oscale = A.out_scale;
    if (MODE == 2 && A.in_amax) {
        const float am = *A.in_amax;
        if (am > 1.0e-30f && am < 3.0e38f) {
            int e;
            (void)frexpf(am, &e);                      // am = m 2^e, m in [0.5, 1): floor(log2 am) = e - 1
            in_s = ldexpf(1.f, 14 - e);                // exact powers of two either way
            oscale = A.out_scale * ldexpf(1.f, e - 14);
        }
    }

    f32x16 acc[RPW][2];
#pragma unroll
    for (int m = 0; m < RPW; ++m)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][j][e] = 0.f;

    const unsigned short *wcb = A.wp + (size_t)cb * A.n_ch * TAPS * NW * NB * KC;

    // The halo tile of chunk ch+1 is fetched into registers while the MFMAs of chunk ch run
    // (NPF 16-byte loads per thread, consumed -- split + LDS store -- after the barrier).
    constexpr int NPF = IO16 ? (NPIX * 2 + 255) / 256 : (NPIX * 4 + 255) / 256;   // 16-byte pieces of the halo tile per thread
    float4 pf[NPF];
    auto fetch = [&](const int ch) {
        const bool first = ch < A.n_ch1;
        const int cl = first ? ch * KC : (ch - A.n_ch1) * KC;
        const int Cs = first ? A.C1 : A.C2, ld = first ? A.ld1 : A.ld2;
        const float *xs = first ? A.x1 + (size_t)(n % A.N1) * H * W * A.ld1 : A.x2 + (size_t)(n % A.N2) * H * W * A.ld2;
        if (IO16) {   // bf16 storage: a pixel's 16 channels are 32 bytes = two 16-byte pieces, no split needed
            const unsigned short *xh = reinterpret_cast<const unsigned short *>(first ? A.x1 : A.x2) +
                                       (size_t)(n % (first ? A.N1 : A.N2)) * H * W * ld;
#pragma unroll
            for (int k = 0; k < NPF; ++k) {
                const int i = tid + k * 256;
                const int p = i >> 1, q = i & 1;
                const int py = p / PW, px = p - py * PW;
                const int gy = y0 + py - HALO, gx = x0 + px - HALO, c = cl + 8 * q;
                pf[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < NPIX * 2 && gy >= 0 && gy < H && gx >= 0 && gx < W && c < Cs)
                    pf[k] = *reinterpret_cast<const float4 *>(xh + ((size_t)gy * W + gx) * ld + c);
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int i = tid + k * 256;
            const int p = i >> 2, q = i & 3;
            const int py = p / PW, px = p - py * PW;
            const int gy = y0 + py - HALO, gx = x0 + px - HALO, c = cl + 4 * q;
            pf[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < NPIX * 4 && gy >= 0 && gy < H && gx >= 0 && gx < W && c < Cs)
                pf[k] = *reinterpret_cast<const float4 *>(xs + ((size_t)gy * W + gx) * ld + c);
        }
    };
    fetch(0);
    // Launches of a few hundred blocks (the 40^2 ... 160^2 maps of a training step) run one short chunk loop per block, all
    // blocks in step: a weight slab that is not in the cache (packed at the top of the step, 70 MB of activations ago) costs
    // one memory latency per tap, serially.  Up to 2048 128-byte lines of the slab are requested here at once instead (plain
    // loads, xor-ed into a value nothing depends on; the wait the compiler puts in front of the xor is the one the halo tile
    // of chunk 0 needs anyway).  The blocks of one XCD (every 8th in dispatch order) share its L2: each takes 1 / warm_w
    // of the slab.  (Plain loads, not inline asm into a dead register: nothing would keep the allocator from reusing such a
    // register while its load is still in flight.)
    if (A.warm_w) {
        const int lines = A.n_ch * TAPS * NW * (NB * KC * 2 / 128);
        const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const int part = (lin / (8 * A.n_cb)) % A.warm_w, per = (lines + A.warm_w - 1) / A.warm_w;
        const int lo = part * per, hi = (part + 1) * per < lines ? (part + 1) * per : lines;
        const char *wb = reinterpret_cast<const char *>(wcb);
        unsigned int sink = 0;
        if (lo < hi) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = lo + tid + k * 256;
                sink ^= *reinterpret_cast<const unsigned int *>(wb + (size_t)(i < hi ? i : hi - 1) * 128);
            }
        }
        asm volatile("" : : "v"(sink));
    }
    const int nj = (A.Cout - cb * NB > 32) ? 2 : 1;  // a last cout block of <= 32 channels skips its second MFMA column

    u32x4 ball[3][2][2];   // 3x3, ring path: B fragments (<= 2 loaded planes) of three taps in flight
    (void)ball;
    STAMP(0)
    for (int ch = 0; ch < A.n_ch; ++ch) {
        if (ch) __syncthreads();
        STAMP(1)
#ifdef MREFSR_CONV_STAMP
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(2)
#endif
        // ---- prefetched halo tile of this 16-channel chunk -> NS bf16 planes in LDS
        if (IO16) {
#pragma unroll
            for (int k = 0; k < NPF; ++k) {
                const int i = tid + k * 256;
                if (i < NPIX * 2) *reinterpret_cast<float4 *>(smem + (i & 1) * HPLANE + (i >> 1) * 16) = pf[k];
            }
        } else
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int i = tid + k * 256;
            if (i < NPIX * 4) {
                const int p = i >> 2, q = i & 3;
                u32x2 sp[NS];
                float4 raw = pf[k];
                if (MODE == 2 && A.in_amax) raw.x *= in_s, raw.y *= in_s, raw.z *= in_s, raw.w *= in_s;
                if (MODE == 2 && A.range_flag) {  // fp16 range guard: the host reads the flag whenever it likes
                    const float4 v = raw;
                    if (!(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))) <= 65000.f)) atomicOr(A.range_flag, 1);
                }
                split4<MODE>(raw, sp);
#pragma unroll
                for (int s = 0; s < NS; ++s) *reinterpret_cast<u32x2 *>(smem + s * PLANE + (q >> 1) * HPLANE + p * 16 + (q & 1) * 8) = sp[s];
            }
        }
        STAMP(3)
        __syncthreads();
        STAMP(4)
        const unsigned short *wch = wcb + (size_t)ch * TAPS * NW * NB * KC + (size_t)l31 * KC + kh * 8;
#ifndef MREFSR_CONV_RING
#define MREFSR_CONV_RING 0
#endif
        if constexpr (KS == 3 && (IO16 || (MODE == 2 && MREFSR_CONV_RING))) {
            // B fragments (packed weights, L2-resident) run ahead of their MFMAs in a small register ring that carries over
            // from chunk to chunk, refilled by the tap that has just consumed a slot:
            //   bf16 arithmetic (8 MFMAs = 256 cycles per tap and wave): three slots, fragments of tap + 3 (slot = dx);
            //   fp16 two-term split (24 MFMAs per tap): two slots, fragments of tap + 2 (slot = global tap parity; the rows of
            //   taps alternate between two instantiations of the row body so that the slot index stays a constant).
            // The scheme of the other modes below (next tap's fragments fetched early for taps < 6, just in time after the
            // halo prefetch has taken its registers) left three taps per chunk waiting for the L2.
            constexpr int NL = IO16 ? 1 : NWL;       // weight planes loaded per fragment
            constexpr int RING = IO16 ? 3 : 2;
            auto bload = [&](const unsigned short *wc, const int tap, const int j, const int sp) {
                return *reinterpret_cast<const u32x4 *>(wc + ((size_t)(tap * NW + sp) * NB + j * 32) * KC);
            };
            if (ch == 0) {
#pragma unroll
                for (int tap = 0; tap < RING; ++tap)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int sp = 0; sp < NL; ++sp) ball[tap][j][sp] = bload(wch, tap, j, sp);
            }
            auto row = [&](const int dy, auto parity) {
                constexpr int P = decltype(parity)::value;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    constexpr int dummy = 0;
                    (void)dummy;
                    const int slot = RING == 3 ? dx : ((P + dx) & 1);
                    u32x4 bw[2][NW];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
#pragma unroll
                        for (int sp = 0; sp < NL; ++sp) bw[j][sp] = ball[slot][j][sp];
                        if (MODE == 2 && NL < NW) bw[j][2] = scale_wh(bw[j][0]);   // the third weight plane is derived, not loaded
                    }
#pragma unroll
                    for (int m = 0; m < RPW; ++m) {
                        u32x4 a[NS];
#pragma unroll
                        for (int sp = 0; sp < NS; ++sp)
                            a[sp] = *reinterpret_cast<const u32x4 *>(smem + sp * PLANE + kh * HPLANE + ((wv * RPW + m + dy) * PW + l31 + dx) * 16);
#pragma unroll
                        for (int t = 0; t < NT; ++t)
#pragma unroll
                            for (int j = 0; j < 2; ++j)
                                if (j < nj) acc[m][j] = mma<MODE>(a[TERM_A[MODE][t]], bw[j][TERM_W[MODE][t]], acc[m][j]);
                    }
                    // refill this slot with the fragments RING taps ahead (possibly in the next chunk)
                    int nt = 3 * dy + dx + RING;
                    const unsigned short *wn = wch;
                    bool have = true;
                    if (nt >= TAPS) {
                        nt -= TAPS;
                        wn = wch + (size_t)TAPS * NW * NB * KC;
                        have = ch + 1 < A.n_ch;
                    }
                    if (have) {
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int sp = 0; sp < NL; ++sp) ball[slot][j][sp] = bload(wn, nt, j, sp);
                    }
                }
            };
#pragma nounroll
            for (int dy = 0; dy < 3; ++dy) {
                if (RING == 3 || ((3 * ch + dy) & 1) == 0) row(dy, std::integral_constant<int, 0>{});
                else row(dy, std::integral_constant<int, 1>{});
                if (dy == 0) {   // the next chunk's halo tile: requested once the first taps are under way
                    if (ch + 1 < A.n_ch) {
                        fetch(ch + 1);
                    } else {
#pragma unroll
                        for (int k = 0; k < NPF; ++k) pf[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
            }
            continue;
        }
        u32x4 b[2][NW];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int s = 0; s < NWL; ++s) b[j][s] = *reinterpret_cast<const u32x4 *>(wch + ((size_t)s * NB + j * 32) * KC);
        // One tap = 48 MFMAs per wave.  Register budget (<= 256 for two waves per SIMD) decides who may be
        // in flight: during taps [0, T_SPLIT) the B fragments of tap+1 are fetched early (24 VGPRs) so no
        // tap starts by waiting for the L2; then the next chunk's halo tile is requested (40 VGPRs) and
        // the remaining taps load their B fragments just in time.
        auto tap_body = [&](const int tap, auto early_b) {
            constexpr bool EARLY = decltype(early_b)::value;
            const int dy = KS == 3 ? (tap * 11) >> 5 : 0, dx = tap - 3 * dy;
            u32x4 bn[2][NWL];
            if (EARLY) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int s = 0; s < NWL; ++s)
                        bn[j][s] = *reinterpret_cast<const u32x4 *>(wch + ((size_t)((tap + 1) * NW + s) * NB + j * 32) * KC);
            }
            if (NWL < NW) {  // the third weight plane is derived, not loaded: a third fewer B loads
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j][2] = scale_wh(b[j][0]);
            }
#pragma unroll
            for (int m = 0; m < RPW; ++m) {
                const int p = (wv * RPW + m + dy) * PW + l31 + dx;
                u32x4 a[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s) a[s] = *reinterpret_cast<const u32x4 *>(smem + s * PLANE + kh * HPLANE + p * 16);
                // partial products, smallest first; consecutive MFMAs alternate between the two cout
                // columns so that no MFMA waits for the accumulator of the one issued just before it
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        if (j < nj) acc[m][j] = mma<MODE>(a[TERM_A[MODE][t]], b[j][TERM_W[MODE][t]], acc[m][j]);
            }
            if (EARLY) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int s = 0; s < NWL; ++s) b[j][s] = bn[j][s];
            } else if (tap + 1 < TAPS) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int s = 0; s < NWL; ++s)
                        b[j][s] = *reinterpret_cast<const u32x4 *>(wch + ((size_t)((tap + 1) * NW + s) * NB + j * 32) * KC);
            }
        };
        constexpr int T_SPLIT = (KS == 3) ? T_EARLY : 0;
#pragma nounroll
        for (int tap = 0; tap < T_SPLIT; ++tap) tap_body(tap, std::true_type{});
        if (ch + 1 < A.n_ch) {
            fetch(ch + 1);
        } else {  // (defines pf on every path, so its registers are free during the early taps)
#pragma unroll
            for (int k = 0; k < NPF; ++k) pf[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma nounroll
        for (int tap = T_SPLIT; tap < TAPS; ++tap) tap_body(tap, std::false_type{});
        STAMP(5)
    }
#ifdef MREFSR_CONV_STAMP
    struct StampOut {
        unsigned long long *acc, *last;
        __device__ ~StampOut()
        {
            acc[6] += __builtin_readcyclecounter() - *last;
            acc[7] = 1;
            const unsigned slot = (((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)) & 1023u;
            if ((threadIdx.x & 63) == 0)
                for (int i = 0; i < 8; ++i) atomicAdd(&g_conv_stamp[slot][i], acc[i]);
        }
    } stamp_out{st_acc, &t_last};
#endif

    conv_epilogue<MODE, IO16, RES, RPW, 256>(A, acc, smem, tid, wv, wv, cb, n, y0, x0, oscale);
}

// ---- the 8-wave form: TWO cout blocks of 64 share one split halo tile (MODE 2 = fp16 two-term split, fp32 tensors).
// Why: with 64 couts per block a Cout >= 128 layer fetches, range-checks and splits every halo tile once per cout block (11 % of a
// wave's life, and twice / four times / eight times the input traffic of the layer), and the 40 registers of the register-staged
// halo prefetch leave no room to run the weight fragments ahead of their MFMAs (taps 6-8 of every chunk waited for the L2).
// Here a block is 512 threads: waves 0-3 own the four row groups for couts [128 g, 128 g + 64), waves 4-7 the same rows for the
// next 64 couts (wave w and w + 4 sit on the same SIMD).  Per thread the halo tile is 5 instead of 10 sixteen-byte pieces, which
// pays for (a) a three-slot B-fragment ring that runs three taps ahead across chunk boundaries and (b) the LDS tile being
// double-buffered (2 x 38.25 KB, one block per CU): chunk ch + 1 is split and stored WHILE chunk ch is multiplied -- by waves 0-3
// after the first row of taps and by waves 4-7 after the second, so that on every SIMD one wave's VALU work sits beside its
// partner's MFMAs -- and requested from memory a whole chunk earlier.  One barrier per chunk instead of two.  The accumulation
// order of every output is the one of conv_nhwc_kernel (chunks, taps, partial products): bit-identical results.
// Memory operations of the chunk loop are issued by hand (inline asm) and waited for by hand (`s_waitcnt vmcnt(N)` with N
// counted below): left to the compiler, a weight fragment's first use waits with vmcnt(0) -- for the two refills and the halo
// request issued after it as well -- and the ring's three-tap lead collapses to nothing.
__device__ __forceinline__ void vm_load16(u32x4 &dst, const unsigned int voff, const void *sbase, const int imm)
{   // (imm: one of four literal offsets -- the asm template needs a literal, not a register)
    if (imm == 0) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
    else if (imm == 1024) asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
    else if (imm == 2048) asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, %2 offset:3072" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void vm_load16(float4 &dst, const void *vaddr)
{
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(vaddr) : "memory");
}
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// The registers an asm load writes look "ready" to the compiler from the load on: nothing but data flow keeps it from moving
// a plain VALU read of them above the wait.  These forms pass the guarded registers THROUGH the wait (read-write operands), so
// every later use depends on it.
template <int N> __device__ __forceinline__ void vm_wait_for(u32x4 &a, u32x4 &b, u32x4 &c, u32x4 &d)
{
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}
__device__ __forceinline__ void vm_arrived(f32x4 &a, f32x4 &b, f32x4 &c, f32x4 &d, f32x4 &e)
{   // (no instruction: an earlier vm_wait has covered these loads; volatile asm statements keep their order)
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e)::"memory");
}
__device__ __forceinline__ void vm_arrived4(u32x4 &a, u32x4 &b, u32x4 &c, u32x4 &d)
{
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
}
__device__ __forceinline__ const void *scalar_ptr(const void *p)
{
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)v), hi = __builtin_amdgcn_readfirstlane((unsigned int)(v >> 32));
    return reinterpret_cast<const void *>(((unsigned long long)hi << 32) | lo);
}

template <int KS, bool RES, bool ALLTWO>
__global__ __launch_bounds__(512, 2) void conv_nhwc8_kernel(const ConvArgs A)
{
    static_assert(KS == 3, "conv_nhwc8: 3x3 only");
    constexpr int MODE = 2, NS = 2, NW = 3, RPW = 4;
    constexpr int HALO = KS / 2, PH = TH + 2 * HALO, PW = TW + 2 * HALO, NPIX = PH * PW, TAPS = KS * KS;
    constexpr int PLANE = NPIX * KC * 2, HPLANE = NPIX * 16, BUF = NS * PLANE;   // plane = [k half][pixel][8 channels], see conv_nhwc_kernel
    constexpr int NPF = (NPIX * 4 + 511) / 512;   // 5 sixteen-byte pieces of the halo tile per thread
    constexpr int NREF = 4;                       // loads per weight-fragment refill: 2 cout halves x 2 loaded planes
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), wrow = wv & 3, half = wv >> 2;
    const int l31 = lane & 31, kh = lane >> 5;
    const int n_cg = A.n_cb >> 1;     // (launch() takes this kernel for an even number of cout blocks only)
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (A.xcd_bands) {   // one contiguous eighth of the launch per XCD (see conv_nhwc_kernel)
        const unsigned gx = gridDim.x, gy = gridDim.y, lin = bx + gx * (by + gy * bz), per = (gx * gy * gridDim.z) / 8;
        if (lin < per * 8) {
            const unsigned l2 = (lin & 7) * per + (lin >> 3);
            bx = l2 % gx;
            const unsigned t2 = l2 / gx;
            by = t2 % gy, bz = t2 / gy;
        }
    }
    const int cb = 2 * (bx % n_cg) + half, n = bz;
    const int y0 = by * TH, x0 = (bx / n_cg) * TW;
    const int H = A.H, W = A.W;
    float in_s = 1.f, oscale = A.out_scale;
    if (A.in_amax) {
        const float am = *A.in_amax;
        if (am > 1.0e-30f && am < 3.0e38f) {
            int e;
            (void)frexpf(am, &e);
            in_s = ldexpf(1.f, 14 - e);
            oscale = A.out_scale * ldexpf(1.f, e - 14);
        }
    }
    f32x16 acc[RPW][2];
#pragma unroll
    for (int m = 0; m < RPW; ++m)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][j][e] = 0.f;

    // ---- halo tile: request (asm loads on clamped addresses, always NPF of them) / split + store (after the data has arrived).
    // launch() guarantees C1, C2 multiples of 16 (no ragged channel chunk) and H * W * ld * 4 < 2^32: a piece's validity is
    // spatial only and its address a 32-bit byte offset from the image's (scalar) base.
    f32x4 pf[NPF];
    unsigned int poff[NPF], okmask = 0;
#pragma unroll
    for (int k = 0; k < NPF; ++k) {
        const int i = tid + k * 512;
        const int p = i >> 2;
        const int py = p / PW, px = p - py * PW;
        const int gy = y0 + py - HALO, gx = x0 + px - HALO;
        const bool ok = i < NPIX * 4 && gy >= 0 && gy < H && gx >= 0 && gx < W;
        poff[k] = ok ? (unsigned int)(gy * W + gx) : 0u;
        okmask |= ok ? (1u << k) : 0u;
    }
    const unsigned int q16 = (unsigned int)(tid & 3) * 16;   // (512 is a multiple of 4: piece k of a thread is always quarter tid & 3)
    auto fetch = [&](int ch) {
        ch = ch < A.n_ch ? ch : A.n_ch - 1;   // past the last chunk: a valid address, the data is never used
        const bool first = ch < A.n_ch1;
        const int cl = first ? ch * KC : (ch - A.n_ch1) * KC;
        const unsigned int ldb = (unsigned int)(first ? A.ld1 : A.ld2) * 4u;
        const float *xs = (first ? A.x1 + (size_t)(n % A.N1) * H * W * A.ld1 : A.x2 + (size_t)(n % A.N2) * H * W * A.ld2) + cl;
        const void *sb = scalar_ptr(xs);
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const unsigned int vo = __umul24(poff[k], ldb) + q16;
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(pf[k]) : "v"(vo), "s"(sb) : "memory");
        }
    };
    // fp16 range guard without a branch: the largest |x| seen, compared as IEEE bit patterns (unsigned order = magnitude order
    // for non-negative floats; Inf and every NaN sort above all finite values), one flag store at the end of the kernel
    unsigned int amax_bits = 0;
    unsigned char *const st_lane = smem + ((tid & 3) >> 1) * HPLANE + (tid >> 2) * 16 + (tid & 1) * 8;   // piece k of this thread lands at + k * 2048
    auto fill = [&](const int buf_off) {   // pf = one chunk of the halo tile -> the two fp16 planes of the buffer at smem + buf_off
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const bool ok = (okmask >> k) & 1u;
            float4 raw;
            raw.x = ok ? pf[k][0] * in_s : 0.f, raw.y = ok ? pf[k][1] * in_s : 0.f, raw.z = ok ? pf[k][2] * in_s : 0.f, raw.w = ok ? pf[k][3] * in_s : 0.f;
            const unsigned int m01 = max(__float_as_uint(raw.x) & 0x7fffffffu, __float_as_uint(raw.y) & 0x7fffffffu);
            const unsigned int m23 = max(__float_as_uint(raw.z) & 0x7fffffffu, __float_as_uint(raw.w) & 0x7fffffffu);
            amax_bits = max(amax_bits, max(m01, m23));
            u32x2 sp[NS];
            split4<MODE>(raw, sp);
            if (tid + k * 512 < NPIX * 4) {
#pragma unroll
                for (int s = 0; s < NS; ++s) *reinterpret_cast<u32x2 *>(st_lane + buf_off + s * PLANE + k * (512 / 4) * 16) = sp[s];
            }
        }
    };
    // ---- weight fragments: ring of three taps, [slot][cout half j][plane wh | wl]
    constexpr size_t WCH_BYTES = (size_t)TAPS * NW * NB * KC * 2;   // packed bytes per (cout block, chunk)
    const unsigned char *wcb = reinterpret_cast<const unsigned char *>(A.wp) + (size_t)cb * A.n_ch * WCH_BYTES;
    const unsigned int voff = (unsigned int)(l31 * KC + kh * 8) * 2;
    u32x4 ring[3][2][2];
    // a last cout block of <= 32 channels skips its second MFMA column (wave-uniform branch; ALLTWO: launch() saw no such block)
    const bool two = ALLTWO || A.Cout - cb * NB > 32;
    auto refill = [&](const int slot, const unsigned char *chunk_base, const int tap) {   // always NREF loads: the waits count them
        const void *sb = scalar_ptr(chunk_base + (size_t)tap * NW * NB * KC * 2);
        vm_load16(ring[slot][0][0], voff, sb, 0);       // j = 0, wh
        vm_load16(ring[slot][1][0], voff, sb, 1024);    // j = 1, wh
        vm_load16(ring[slot][0][1], voff, sb, 2048);    // j = 0, wl
        vm_load16(ring[slot][1][1], voff, sb, 3072);    // j = 1, wl
    };
    // Order of the memory operations, per wave (loads return in issue order, so a wait for load X also waits for everything older):
    //   prologue   halo(0) | wait | split -> buffer 0 | halo(1) | ring: taps 0, 1, 2 of chunk 0
    //   chunk c    tap t: wait for slot t % 3 (leaving the loads issued after it in flight: the two later refills, + the halo request
    //              for taps 6-8) | 24 MFMAs | refill the slot with tap t + 3 (of chunk c + 1 for t >= 6)
    //              after tap 0 (waves 0-3) / tap 3 (waves 4-7): split halo(c + 1) -> the other buffer   [it arrived with tap 0's wait:
    //              the refill tap 0 waits for was issued after it] -- staggered so that one wave's VALU runs beside its SIMD
    //              partner's MFMAs; the branch holds no memory operation and no wait
    //              after tap 5: request halo(c + 2): taps 6, 7, 8 still run on fragments requested before it (three taps = ~4 k cycles
    //              for the HBM round trip), the next chunk's tap 0 is the first wait that includes it
    static_assert(NPF == 5, "vm_arrived takes the five pieces of a thread");
    fetch(0);
    vm_wait<0>();
    vm_arrived(pf[0], pf[1], pf[2], pf[3], pf[4]);
    fill(0);
    fetch(1);
#pragma unroll
    for (int t = 0; t < 3; ++t) refill(t, wcb, t);
    __syncthreads();

    const unsigned char *a_lane = smem + kh * HPLANE + ((wrow * RPW) * PW + l31) * 16;   // this lane's A fragment of (row 0, tap 0)
    for (int ch = 0; ch < A.n_ch; ++ch) {
        const unsigned char *cur = a_lane + (ch & 1) * BUF;
        const int nxt = ((ch + 1) & 1) * BUF;
        const unsigned char *wch = wcb + (size_t)ch * WCH_BYTES;
        const unsigned char *wnx = wcb + (size_t)(ch + 1 < A.n_ch ? ch + 1 : ch) * WCH_BYTES;   // past the end: clamped, unused
        u32x4 a[NS], an[NS];
#pragma unroll
        for (int sp = 0; sp < NS; ++sp) a[sp] = *reinterpret_cast<const u32x4 *>(cur + sp * PLANE);
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int dx = tap % 3;
            if (tap >= 6) vm_wait_for<2 * NREF + NPF>(ring[dx][0][0], ring[dx][1][0], ring[dx][0][1], ring[dx][1][1]);
            else vm_wait_for<2 * NREF>(ring[dx][0][0], ring[dx][1][0], ring[dx][0][1], ring[dx][1][1]);
            u32x4 bw[2][NW];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bw[j][0] = ring[dx][j][0], bw[j][1] = ring[dx][j][1];
                bw[j][2] = scale_wh(bw[j][0]);
            }
#pragma unroll
            for (int m = 0; m < RPW; ++m) {
                // the next A fragments (next row of this tap, or row 0 of the next tap) are requested before this row's MFMAs
                const int nm = m + 1 < RPW ? m + 1 : 0, ntap = m + 1 < RPW ? tap : tap + 1;
                if (ntap < TAPS) {
                    const int ndy = ntap / 3, ndx = ntap - 3 * ndy;
#pragma unroll
                    for (int sp = 0; sp < NS; ++sp) an[sp] = *reinterpret_cast<const u32x4 *>(cur + sp * PLANE + ((nm + ndy) * PW + ndx) * 16);
                }
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    acc[m][0] = mma<MODE>(a[TERM_A[MODE][t]], bw[0][TERM_W[MODE][t]], acc[m][0]);
                    if (two) acc[m][1] = mma<MODE>(a[TERM_A[MODE][t]], bw[1][TERM_W[MODE][t]], acc[m][1]);
                }
#pragma unroll
                for (int sp = 0; sp < NS; ++sp) a[sp] = an[sp];
            }
            if (tap + 3 < TAPS) refill(dx, wch, tap + 3);
            else refill(dx, wnx, tap + 3 - TAPS);
            if (tap == 0) {
                vm_arrived(pf[0], pf[1], pf[2], pf[3], pf[4]);   // (every wave: the data flow of pf passes through here once per chunk)
                if (half == 0) fill(nxt);
            }
            if (tap == 3 && half == 1) fill(nxt);
            if (tap == 5) fetch(ch + 2);
        }
        __syncthreads();
    }
    // requests past the last chunk are still in flight: wait, and keep their destination registers "in use" up to here -- to the
    // compiler they were dead after the loop, and anything it had placed in them before the wait would be overwritten on arrival
    vm_wait<0>();
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) asm volatile("" ::"v"(ring[t][j][sp]) : "memory");
#pragma unroll
    for (int k = 0; k < NPF; ++k) asm volatile("" ::"v"(pf[k]) : "memory");
    if (A.range_flag && amax_bits > __float_as_uint(65000.f)) atomicOr(A.range_flag, 1);
    conv_epilogue<MODE, false, RES, RPW, 512>(A, acc, smem, tid, wv, wrow, cb, n, y0, x0, oscale);
}

#ifndef MREFSR_CONV_SMALL
#define MREFSR_CONV_SMALL 1
#endif
#ifndef MREFSR_CONV_NT
#define MREFSR_CONV_NT 1
#endif
static long xcd_min_blocks()
{
    static const long v = getenv("MREFSR_CONV_XCD_MIN") ? atol(getenv("MREFSR_CONV_XCD_MIN")) : 128;   // (2048 at first: the 40^2 ... 160^2 launches of the training step gain too, 38.5 -> 37.9 ms)
    return v;
}

// ---- 1 x 1 convolutions of fp32 tensors (MODE 2): the channel-reduction layers in front of the trunks (the DynAgg fusion layers and
// heads of ref_mrapa_restoration_arch.py:217-225,271-304: 576 -> 512 ... 192 -> 64 channels at 160^2 .. 640^2).  A 1 x 1 chunk is ONE tap
// = 24 MFMAs per wave (768 clocks) between the two barriers of conv_nhwc_kernel's chunk loop, with the weight fragments requested just in
// time and the next chunk's tile one MFMA phase ahead: 2.7 TB/s on layers whose only real cost is reading the input once and writing the
// output.  Here: 8 x 32 pixel tiles (64 accumulator registers: room for two tiles in flight), the tile of a chunk requested TWO chunks
// ahead (two register sets, the loop unrolled by two), the split tile double-buffered in LDS (one barrier per chunk), the weight
// fragments of chunk ch + 1 requested IN FRONT of the tile of chunk ch + 2 (they are consumed a chunk earlier and the return queue is in
// order: requested behind the tile, their wait would be the tile's as well).  Requests and waits by hand (see conv_nhwc8_kernel: the
// compiler's own waits in a loop with loads in flight across the back edge are vmcnt(0)); every chunk issues the same eight requests -- past
// the last chunk the tile's lanes are out of bounds (zeros, no traffic) and the fragments are the last chunk's again -- so that the counts
// are constants:  queue at the top of chunk ch: [tile ch: 4] [weights ch: 4] [tile ch + 1: 4].
// Same cout block (64), same accumulation order (chunks, partial products smallest first), same epilogue as conv_nhwc_kernel:
// bit-identical results.
template <bool RES>
__global__ __launch_bounds__(256, 2) void conv1x1_kernel(const ConvArgs A)
{
    constexpr int MODE = 2, NS = 2, NW = 3, NT = 3, RPW = 2, THB = 4 * RPW, NPIX = THB * TW, PLANE = NPIX * KC * 2, HPLANE = NPIX * 16, BUF = NS * PLANE;
    constexpr int NPF = NPIX * 4 / 256;   // 16-byte pieces of a chunk's tile per thread
    static_assert(NPF == 4, "conv1x1: four pieces per thread (the waits pass four registers through)");
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (A.xcd_bands) {   // (conv_nhwc_kernel: an XCD walks a contiguous band of tiles, the cout blocks of a tile meet in one L2)
        const unsigned gx = gridDim.x, gy = gridDim.y, lin = bx + gx * (by + gy * bz), per = (gx * gy * gridDim.z) / 8;
        if (lin < per * 8) {
            const unsigned l2 = (lin & 7) * per + (lin >> 3);
            bx = l2 % gx;
            const unsigned t2 = l2 / gx;
            by = t2 % gy, bz = t2 / gy;
        }
    }
    const int cb = bx % A.n_cb, n = bz;
    const int y0 = by * THB, x0 = (bx / A.n_cb) * TW;
    const int H = A.H, W = A.W, n_ch = A.n_ch;
    float in_s = 1.f, oscale = A.out_scale;
    if (A.in_amax) {
        const float am = *A.in_amax;
        if (am > 1.0e-30f && am < 3.0e38f) {
            int e;
            (void)frexpf(am, &e);
            in_s = ldexpf(1.f, 14 - e);
            oscale = A.out_scale * ldexpf(1.f, e - 14);
        }
    }
    f32x16 acc[RPW][2];
#pragma unroll
    for (int m = 0; m < RPW; ++m)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][j][e] = 0.f;

    // piece k of a thread: quarter tid & 3 of pixel (tid >> 2) + 64 k of the tile = row 2 k + (tid >> 7), column (tid >> 2) & 31.  Requests
    // are buffer loads on a descriptor of the source image: rows below the image lie beyond the descriptor by themselves, a lane right
    // of the image or past a ragged chunk's channels ORs an offset beyond it into its own -- zeros either way, no branches in the chunk
    // loop (launch(): an image is smaller than 2^31 bytes)
    constexpr unsigned int OOB = 0xffff0000u;
    const int p_q = tid & 3, p_px = (tid >> 2) & 31, p_r0 = tid >> 7;
    const unsigned int p_xmask = x0 + p_px < W ? 0u : OOB;
    const unsigned int p_pix = (unsigned int)((y0 + p_r0) * W + x0 + p_px);
    const float *const img1 = A.x1 + (size_t)(n % A.N1) * H * W * A.ld1, *const img2 = A.x2 ? A.x2 + (size_t)(n % A.N2) * H * W * A.ld2 : A.x1;
    auto fetch = [&](const int ch, f32x4 (&pf)[NPF]) {
        const bool past = ch >= n_ch, first = past || ch < A.n_ch1;
        const int cl = first ? ch * KC : (ch - A.n_ch1) * KC;
        const int Cs = past ? 0 : (first ? A.C1 : A.C2), ld = first ? A.ld1 : A.ld2;
        const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(scalar_ptr(first ? img1 : img2)), 0,
                                                                             __builtin_amdgcn_readfirstlane((unsigned int)((size_t)H * W * ld * 4)), 0x00020000);
        const unsigned int cmask = (cl + 4 * p_q < Cs ? 0u : OOB) | p_xmask;
        const unsigned int base = (p_pix * (unsigned int)ld + (unsigned int)(cl + 4 * p_q)) * 4u;
        const unsigned int step = __builtin_amdgcn_readfirstlane((unsigned int)(2 * W * ld) * 4u);
        // (the whole offset travels in the VECTOR operand -- the one the descriptor's range check is sure to see -- and the mask goes
        // on last: a row below the image lies beyond the descriptor whatever the hardware does with a scalar offset)
#pragma unroll
        for (int k = 0; k < NPF; ++k) {   // (s_nop: the descriptor may have been written by the instruction in front, conv_wino4.hip)
            const unsigned int vo = (base + (unsigned int)k * step) | cmask;
            asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(pf[k]) : "v"(vo), "s"(srd) : "memory");
        }
    };
    float rmax = 0.f;   // fp16 range guard: the largest |x| this thread has split (one atomic at the end instead of a branch per piece)
    const unsigned int st_off = (unsigned int)((p_q >> 1) * HPLANE + (tid >> 2) * 16 + (p_q & 1) * 8);
    auto fill = [&](const f32x4 (&pf)[NPF], unsigned char *const buf) {
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            float4 raw = make_float4(pf[k][0] * in_s, pf[k][1] * in_s, pf[k][2] * in_s, pf[k][3] * in_s);   // (1 without an input scale)
            rmax = fmaxf(fmaxf(rmax, fabsf(raw.x)), fmaxf(fabsf(raw.y), fmaxf(fabsf(raw.z), fabsf(raw.w))));
            u32x2 sp[NS];
            split4<MODE>(raw, sp);
#pragma unroll
            for (int s = 0; s < NS; ++s) *reinterpret_cast<u32x2 *>(buf + s * PLANE + st_off + k * (64 * 16)) = sp[s];
        }
    };
    // fragment (cout half j, plane s) of a chunk: 1 KB at s * 2 KB + j * 1 KB of the chunk's 6 KB in the packed weights
    const unsigned int w_voff = (unsigned int)(l31 * KC + kh * 8) * 2u;
    const unsigned short *const wcb = A.wp + (size_t)cb * n_ch * NW * NB * KC;
    auto wload = [&](const int ch, u32x4 (&b)[2][2]) {
        const void *const sb = scalar_ptr(wcb + (size_t)(ch < n_ch ? ch : n_ch - 1) * NW * NB * KC);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int s = 0; s < 2; ++s) vm_load16(b[j][s], w_voff, sb, s * 2048 + j * 1024);
    };
    const unsigned int a_off = (unsigned int)(kh * HPLANE + (wv * RPW * TW + l31) * 16);
    auto body = [&](const int ch, f32x4 (&pf)[NPF], unsigned char *const buf, u32x4 (&b)[2][2], u32x4 (&bn)[2][2]) {
        asm volatile("s_waitcnt vmcnt(8)" : "+v"(pf[0]), "+v"(pf[1]), "+v"(pf[2]), "+v"(pf[3]) : : "memory");   // tile ch; younger: weights ch, tile ch + 1
        fill(pf, buf);
        __builtin_amdgcn_sched_barrier(0);
        wload(ch + 1, bn);
        fetch(ch + 2, pf);
        __syncthreads();   // the tile of chunk ch is complete; its buffer's readers of chunk ch - 2 are a barrier behind
        vm_wait_for<12>(b[0][0], b[0][1], b[1][0], b[1][1]);   // weights ch; younger: tile ch + 1, weights ch + 1, tile ch + 2
        u32x4 bw[2][NW];
#pragma unroll
        for (int j = 0; j < 2; ++j) bw[j][0] = b[j][0], bw[j][1] = b[j][1], bw[j][2] = scale_wh(b[j][0]);
        // (the fragments of row m + 1 are read beside the MFMAs of row m)
        u32x4 a[2][NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) a[0][s] = *reinterpret_cast<const u32x4 *>(buf + s * PLANE + a_off);
#pragma unroll
        for (int m = 0; m < RPW; ++m) {
            if (m + 1 < RPW) {
#pragma unroll
                for (int s = 0; s < NS; ++s) a[(m + 1) & 1][s] = *reinterpret_cast<const u32x4 *>(buf + s * PLANE + a_off + (m + 1) * (TW * 16));
            }
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[m][j] = mma<MODE>(a[m & 1][TERM_A[MODE][t]], bw[j][TERM_W[MODE][t]], acc[m][j]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    f32x4 pfa[NPF], pfb[NPF];
    u32x4 b0[2][2], b1[2][2];
    fetch(0, pfa);
    wload(0, b0);
    fetch(1, pfb);
    for (int ch = 0; ch < n_ch; ch += 2) {
        body(ch, pfa, smem, b0, b1);
        if (ch + 1 < n_ch) body(ch + 1, pfb, smem + BUF, b1, b0);
    }
    // (requests past the last chunk are still in flight: nothing may take their registers before they have landed)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(pfa[0]), "+v"(pfa[1]), "+v"(pfa[2]), "+v"(pfa[3]), "+v"(pfb[0]), "+v"(pfb[1]), "+v"(pfb[2]), "+v"(pfb[3]) : : "memory");
    vm_arrived4(b0[0][0], b0[0][1], b0[1][0], b0[1][1]);
    vm_arrived4(b1[0][0], b1[0][1], b1[1][0], b1[1][1]);
    if (A.range_flag && !(rmax <= 65000.f)) atomicOr(A.range_flag, 1);
    conv_epilogue<MODE, false, RES, RPW, 256>(A, acc, smem, tid, wv, wv, cb, n, y0, x0, oscale);
}

template <bool RES>
int launch1x1(const ConvArgs &a, int N, hipStream_t stream)
{
    constexpr int THB = 8;
    constexpr size_t fill = (size_t)2 * 2 * THB * TW * KC * 2, lds1 = fill > (size_t)EP_BYTES ? fill : (size_t)EP_BYTES;
    static unsigned long long attr1 = 0;
    if (mrefsr::first_use_on_device(attr1))
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv1x1_kernel<RES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
    dim3 grid(((a.W + TW - 1) / TW) * a.n_cb, (a.H + THB - 1) / THB, N);
    ConvArgs b = a;
    b.stream_out = MREFSR_CONV_NT && (size_t)N * a.H * a.W * a.ld_out * sizeof(float) > ((size_t)256 << 20);
    const char *ex = getenv("MREFSR_CONV_XCD");
    b.xcd_bands = (ex ? ex[0] != '0' : MREFSR_CONV_XCD_DEFAULT) && (long)grid.x * grid.y * grid.z >= xcd_min_blocks();
    b.warm_w = 0;
    hipLaunchKernelGGL((conv1x1_kernel<RES>), grid, dim3(256), lds1, stream, b);
    return mrefsr::check_launch("conv1x1");
}

template <int MODE, int KS, bool IO16 = false, bool RES = false, int RPW = 4>
int launch(const ConvArgs &a, int N, hipStream_t stream)
{
    if constexpr (MODE == 3 && !IO16) {
        if (a.io16) return launch<3, KS, true>(a, N, stream);
    }
    if constexpr (MODE == 2 && !RES) {
        if (a.residual && a.epilogue != 1 && a.epilogue != 3 && (a.Cout & 3) == 0 && (a.ld_res & 3) == 0 && (a.ld_out & 3) == 0)
            return launch<2, KS, false, true>(a, N, stream);
    }
    if constexpr (MREFSR_CONV_SMALL && RPW == 4 && !IO16 && (MODE == 0 || MODE == 2)) {
        // 16-row tiles: 512 block slots on the chip (2 per CU).  A launch that leaves most of them empty takes one block's
        // lifetime however small it is: give it 4-row (<= 128 such blocks) or 8-row (<= 256) tiles instead.  (Smaller tiles
        // load their weight fragments 4x / 2x as often: once the chip is half full the 16-row tile wins again -- measured on
        // the training step's shapes: 24 blocks 54 -> 24 us, 60 blocks 56 -> 26 us, but 480 blocks 352 -> 387 us.)
        const long blocks = (long)((a.W + TW - 1) / TW) * a.n_cb * ((a.H + TH - 1) / TH) * N;
        if (a.epilogue != 1 && a.epilogue != 3) {
            if (blocks <= 128) return launch<MODE, KS, IO16, RES, 1>(a, N, stream);
            if (blocks <= 256) return launch<MODE, KS, IO16, RES, 2>(a, N, stream);
            // 64-cout layers of a few rounds of 16-row blocks (the 320^2 trunk: 1600 blocks = 3.1 rounds of 512 slots): 8-row
            // tiles run three blocks per CU and end on a shorter tail: +8-12 % there, nothing at 6400 blocks (round 3)
            if (a.n_cb == 1 && blocks >= 1024 && blocks < 4096) return launch<MODE, KS, IO16, RES, 2>(a, N, stream);
        }
    }
    if constexpr (MODE == 2 && !IO16 && KS == 3 && RPW == 4) {
        // Cout > 64 on a launch that fills the chip with 512-thread blocks: two cout blocks share one split halo tile
        // (conv_nhwc8_kernel).  MREFSR_CONV8=0 keeps the 4-wave kernel (A/B runs; same bits either way).
        const char *e8 = getenv("MREFSR_CONV8");   // (read per call: tests flip it inside one process)
        const bool conv8 = !(e8 && e8[0] == '0');
        const long groups = (long)((a.W + TW - 1) / TW) * (a.n_cb / 2) * ((a.H + TH - 1) / TH) * N;
        const bool fits = (a.C1 % KC) == 0 && (a.C2 % KC) == 0 && (size_t)a.H * a.W * (size_t)(a.ld1 > a.ld2 ? a.ld1 : a.ld2) * 4 < ((size_t)1 << 32) &&
                          (size_t)a.H * a.W < ((size_t)1 << 24);
        if (conv8 && fits && a.n_cb >= 2 && (a.n_cb & 1) == 0 && groups >= 256 && !a.stat_sum && !a.res_mask) {
            constexpr int NPIX8 = (TH + 2) * (TW + 2);
            constexpr size_t lds8 = (size_t)2 * 2 * NPIX8 * KC * 2;   // two buffers x two fp16 planes (> 8 epilogue slabs, > the pre-offset tile)
            static_assert(lds8 >= (size_t)2 * EP_BYTES && lds8 >= (size_t)18 * (TH * TW + 4) * 4 + (size_t)8 * 32 * DYN_T_LD * 4, "conv_nhwc8: LDS budget");
            static unsigned long long attr8 = 0;
            if (mrefsr::first_use_on_device(attr8))
            {
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv_nhwc8_kernel<KS, RES, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8);
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv_nhwc8_kernel<KS, RES, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8);
            }
            ConvArgs b = a;
            b.stream_out = MREFSR_CONV_NT && (size_t)N * a.H * a.W * a.ld_out * sizeof(float) > ((size_t)256 << 20);
            dim3 grid(((a.W + TW - 1) / TW) * (a.n_cb / 2), (a.H + TH - 1) / TH, N);
            {
                const char *ex = getenv("MREFSR_CONV_XCD");
                b.xcd_bands = (ex ? ex[0] != '0' : MREFSR_CONV_XCD_DEFAULT) && (long)grid.x * grid.y * grid.z >= xcd_min_blocks();
            }
            const int tail = a.Cout % NB;   // couts of the last block: 1..32 -> that block runs one MFMA column
            if (tail == 0 || tail > 32) hipLaunchKernelGGL((conv_nhwc8_kernel<KS, RES, true>), grid, dim3(512), lds8, stream, b);
            else hipLaunchKernelGGL((conv_nhwc8_kernel<KS, RES, false>), grid, dim3(512), lds8, stream, b);
            return mrefsr::check_launch("conv_nhwc8");
        }
    }
    if constexpr (MODE == 2 && !IO16 && KS == 1 && (RPW == 4 || RPW == 2)) {
        // the 1 x 1 kernel (two chunks of input in flight, double-buffered tile); MREFSR_CONV1X1=0 keeps conv_nhwc_kernel (A/B runs:
        // same bits either way)
        const char *e1 = getenv("MREFSR_CONV1X1");   // (read per call: tests flip it inside one process)
        if (!(e1 && e1[0] == '0') && a.epilogue != 1 && a.epilogue != 3 &&
            (size_t)a.H * a.W * (size_t)(a.ld1 > a.ld2 ? a.ld1 : a.ld2) * 4 < ((size_t)1 << 31) && (a.Cout % NB == 0 || a.Cout % NB > 32)) {
            return launch1x1<RES>(a, N, stream);
        }
    }
    constexpr int THB = 4 * RPW;
    constexpr int NS = ModeTraits<MODE>::NA;
    constexpr int HALO = KS / 2, NPIX = (THB + 2 * HALO) * (TW + 2 * HALO);
    constexpr size_t DYN_BYTES = (size_t)18 * (TH * TW + 4) * 4 + (size_t)4 * 32 * DYN_T_LD * 4;   // pre-offset tile + transposing slabs of epilogue 3 (16-row tiles only)
    const size_t fill = (size_t)NS * NPIX * KC * 2, ep = (RPW == 4 && DYN_BYTES > (size_t)EP_BYTES) ? DYN_BYTES : (size_t)EP_BYTES;
    size_t lds = fill > ep ? fill : ep;   // input tile | epilogue slab | pre-offset tile of epilogue 3
#ifdef MREFSR_CONV_STAMP
    if (const char *e = getenv("MREFSR_CONV_LDS_PAD")) lds += (size_t)atoi(e);   // > 80 KB in total: one block per CU
#endif
    static unsigned long long attr_done = 0;
    if (mrefsr::first_use_on_device(attr_done))
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv_nhwc_kernel<MODE, KS, IO16, RES, RPW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
    dim3 grid(((a.W + TW - 1) / TW) * a.n_cb, (a.H + THB - 1) / THB, N);
    ConvArgs b = a;
    b.stream_out = MREFSR_CONV_NT && (size_t)N * a.H * a.W * a.ld_out * sizeof(float) > ((size_t)256 << 20);
    {
        const char *ex = getenv("MREFSR_CONV_XCD");   // (read per call: A/B runs flip it inside one process)
        b.xcd_bands = (ex ? ex[0] != '0' : MREFSR_CONV_XCD_DEFAULT) && (long)grid.x * grid.y * grid.z >= xcd_min_blocks();
    }
    static const long warm_max = getenv("MREFSR_CONV_WARM") ? atol(getenv("MREFSR_CONV_WARM")) : 1024;
    {   // blocks per XCD and cout block share the slab's lines between them (1 = every block requests all of it, 0 = off)
        const long nblk = (long)grid.x * grid.y * grid.z, share = nblk / (8 * a.n_cb);
        b.warm_w = nblk <= warm_max ? (int)(share < 1 ? 1 : (share > 8 ? 8 : share)) : 0;
    }
    hipLaunchKernelGGL((conv_nhwc_kernel<MODE, KS, IO16, RES, RPW>), grid, dim3(256), lds, stream, b);
    return mrefsr::check_launch("conv_nhwc");
}

int dispatch(const ConvArgs &a, const mrefsr_conv_desc *d, mrefsr_stream_t stream);

}  // namespace

MREFSR_EXPORT int64_t mrefsr_conv_packed_bytes(int Cout, int Cin, int ksize, int terms)
{
    if (terms == 17) return ksize == 3 ? wino_packed_bytes(Cout, Cin) : 0;   // Winograd F(2x2, 3x3) form of terms 16 (conv_wino.hip)
    const int ns = terms == 1 ? 1 : (terms == 3 ? 2 : 3);  // weight planes: bf16 | bf16 hi/lo | bf16 hi/mid/lo | fp16 wh/wl/wh*2^-11
    const long n_ch = (Cin + KC - 1) / KC, n_cb = (Cout + NB - 1) / NB;
    return n_cb * n_ch * ksize * ksize * ns * NB * KC * 2;
}

// weight element (o, i, tap) at weight[o * stride_o + i * stride_i + (flip ? taps - 1 - tap : tap)]: the plain OIHW tensor
// (stride_o = Cin_total * taps, stride_i = taps), an input-channel slice of it (pointer offset), or -- strides swapped and
// the taps flipped -- the operator of its input gradient (dgrad of a stride-1 'same' convolution = the convolution of the
// output gradient with the transposed, point-mirrored kernel)
MREFSR_EXPORT int mrefsr_conv_pack_weight_view_f32(const float *weight, void *packed, int Cout, int Cin, int ksize, int terms, float wscale,
                                                   int64_t stride_o, int64_t stride_i, int flip, mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(weight && packed, "conv_pack_weight: null pointer");
    MREFSR_REQUIRE(Cout > 0 && Cin > 0 && (terms == 6 || terms == 3 || terms == 16 || terms == 1 || terms == 17) && (ksize == 1 || ksize == 3),
                   "conv_pack_weight: Cout=%d Cin=%d ksize=%d terms=%d", Cout, Cin, ksize, terms);
    MREFSR_REQUIRE((terms != 16 && terms != 17) || (wscale > 0.f && wscale < 3.0e38f), "conv_pack_weight: terms=16 / 17 need a positive finite wscale");
    MREFSR_REQUIRE(stride_o > 0 && stride_i > 0, "conv_pack_weight: strides %ld / %ld", (long)stride_o, (long)stride_i);
    if (terms == 17) {
        MREFSR_REQUIRE(ksize == 3, "conv_pack_weight: terms=17 (Winograd F(2x2, 3x3)) is for 3x3 kernels");
        return wino_pack(weight, packed, Cout, Cin, wscale, (long)stride_o, (long)stride_i, flip, nullptr, (hipStream_t)stream);
    }
    const int n_ch = (Cin + KC - 1) / KC, n_cb = (Cout + NB - 1) / NB, taps = ksize * ksize;
    const long total = (long)n_cb * n_ch * taps * NB * KC;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    unsigned short *wp = reinterpret_cast<unsigned short *>(packed);
    hipStream_t st = (hipStream_t)stream;
    const long so = (long)stride_o, si = (long)stride_i;
    if (terms == 6) hipLaunchKernelGGL(conv_pack_kernel<0>, dim3(blocks), dim3(256), 0, st, weight, wp, Cout, Cin, taps, n_cb, n_ch, 1.f, so, si, flip);
    else if (terms == 3) hipLaunchKernelGGL(conv_pack_kernel<1>, dim3(blocks), dim3(256), 0, st, weight, wp, Cout, Cin, taps, n_cb, n_ch, 1.f, so, si, flip);
    else if (terms == 1) hipLaunchKernelGGL(conv_pack_kernel<3>, dim3(blocks), dim3(256), 0, st, weight, wp, Cout, Cin, taps, n_cb, n_ch, 1.f, so, si, flip);
    else hipLaunchKernelGGL(conv_pack_kernel<2>, dim3(blocks), dim3(256), 0, st, weight, wp, Cout, Cin, taps, n_cb, n_ch, wscale, so, si, flip);
    return mrefsr::check_launch("conv_pack_weight");
}

// n_jobs packings as one launch; `jobs` is a table in DEVICE memory (the caller keeps it alive until the launch has run),
// every entry as the arguments of mrefsr_conv_pack_weight_view_f32; the entries are not validated on the host.  range_flag
// (device int32, may be NULL) is set to 1 when a terms-16 job meets |weight * wscale| > 65000 or a non-finite weight: the packed
// copy would hold Inf -- the flag the fp16-split convolutions raise for their activations (mrefsr_conv_nhwc_f32).
MREFSR_EXPORT int mrefsr_conv_pack_weights_multi_f32(const mrefsr_conv_pack_job *jobs, int n_jobs, int *range_flag, mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(jobs && n_jobs > 0 && n_jobs <= 65535, "conv_pack_weights_multi: jobs=%p n_jobs=%d", (const void *)jobs, n_jobs);
    hipLaunchKernelGGL(conv_pack_multi_kernel, dim3(48, n_jobs), dim3(256), 0, (hipStream_t)stream, jobs, range_flag);
    return mrefsr::check_launch("conv_pack_weights_multi");
}

MREFSR_EXPORT int mrefsr_conv_pack_weight_f32(const float *weight, void *packed, int Cout, int Cin, int ksize, int terms,
                                              float wscale, mrefsr_stream_t stream)
{
    return mrefsr_conv_pack_weight_view_f32(weight, packed, Cout, Cin, ksize, terms, wscale, (int64_t)Cin * ksize * ksize, (int64_t)ksize * ksize, 0,
                                            stream);
}

MREFSR_EXPORT int mrefsr_conv_nhwc_f32(const mrefsr_conv_desc *d, const float *x1, const float *x2, const void *packed,
                                       const float *bias, const float *slope_ptr, const float *pre, const float *residual, float *out,
                                       int *range_flag, mrefsr_stream_t stream)
{
    return mrefsr_conv_nhwc_scaled_f32(d, x1, x2, packed, bias, slope_ptr, pre, residual, out, range_flag, nullptr, stream);
}

namespace {
int conv_entry(const mrefsr_conv_desc *d, const float *x1, const float *x2, const void *packed, const float *bias, const float *slope_ptr,
               const float *pre, const float *residual, float *out, int *range_flag, const float *in_amax, int res_mask, float *stat_sum,
               float *stat_amax, mrefsr_stream_t stream);
}

// mrefsr_conv_nhwc_scaled_f32 + out_amax[0] = max(out_amax[0], max |out|) (device memory, zero-initialised by the caller, may be NULL):
// what the Winograd launch that reads `out` next takes as its in_amax -- always the current batch's, no reduction pass.
MREFSR_EXPORT int mrefsr_conv_nhwc_amax_f32(const mrefsr_conv_desc *d, const float *x1, const float *x2, const void *packed, const float *bias,
                                            const float *slope_ptr, const float *pre, const float *residual, float *out, int *range_flag,
                                            const float *in_amax, float *out_amax, mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(!out_amax || (d && (d->terms == 16 || d->terms == 17)), "conv_nhwc_amax: out_amax belongs to the fp32-equivalent modes (terms = 16, 17)");
    return conv_entry(d, x1, x2, packed, bias, slope_ptr, pre, residual, out, range_flag, in_amax, 0, nullptr, out_amax, stream);
}

MREFSR_EXPORT int mrefsr_conv_nhwc_scaled_f32(const mrefsr_conv_desc *d, const float *x1, const float *x2, const void *packed,
                                              const float *bias, const float *slope_ptr, const float *pre, const float *residual, float *out,
                                              int *range_flag, const float *in_amax, mrefsr_stream_t stream)
{
    return conv_entry(d, x1, x2, packed, bias, slope_ptr, pre, residual, out, range_flag, in_amax, 0, nullptr, nullptr, stream);
}

// The input-gradient convolution of a training step with the element-wise pass that would follow it folded into its epilogue:
// residual_is_mask = 1 turns the `residual` operand into the ReLU mask of the layer below (out = residual > 0 ? conv : 0 -- what
// mrefsr_act_bwd_nhwc_f32 does to this launch's output with act = 1, slope = 0); stat_sum[Cout] += per-channel sums of `out`
// (the bias gradient of the layer below) and stat_amax[0] = max(stat_amax[0], max |out|) (the input scale of the launches that
// read `out` next), both zero-initialised by the caller, either may be NULL.  terms = 16, fp32 tensors, plain epilogue.
MREFSR_EXPORT int mrefsr_conv_nhwc_bwd_f32(const mrefsr_conv_desc *d, const float *x1, const void *packed, const float *residual,
                                           int residual_is_mask, float *out, int *range_flag, const float *in_amax, float *stat_sum,
                                           float *stat_amax, mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(d && d->terms == 16 && d->epilogue == 0 && d->act == 0 && d->C2 == 0, "conv_nhwc_bwd: terms 16, plain epilogue, one source, no activation");
    MREFSR_REQUIRE(d->Cout % 4 == 0 && d->ld_out % 4 == 0, "conv_nhwc_bwd: Cout=%d ld_out=%d (multiples of 4)", d->Cout, d->ld_out);
    MREFSR_REQUIRE(!residual_is_mask || (residual && d->ld_res % 4 == 0), "conv_nhwc_bwd: the mask source is the residual operand (ld_res a multiple of 4)");
    return conv_entry(d, x1, nullptr, packed, nullptr, nullptr, nullptr, residual, out, range_flag, in_amax, residual_is_mask ? 1 : 0, stat_sum, stat_amax,
                      stream);
}

namespace {
int conv_entry(const mrefsr_conv_desc *d, const float *x1, const float *x2, const void *packed, const float *bias, const float *slope_ptr,
               const float *pre, const float *residual, float *out, int *range_flag, const float *in_amax, int res_mask, float *stat_sum,
               float *stat_amax, mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(d && x1 && packed && out, "conv_nhwc: null pointer");
    MREFSR_REQUIRE(!in_amax || d->terms == 16 || d->terms == 17, "conv_nhwc_scaled: the input scale belongs to the fp16 two-term modes (terms = 16, 17)");
    MREFSR_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->C1 > 0 && d->Cout > 0 && d->C2 >= 0,
                   "conv_nhwc: N=%d H=%d W=%d C1=%d C2=%d Cout=%d", d->N, d->H, d->W, d->C1, d->C2, d->Cout);
    MREFSR_REQUIRE(d->ksize == 1 || d->ksize == 3, "conv_nhwc: ksize=%d (1 or 3)", d->ksize);
    MREFSR_REQUIRE(d->terms == 6 || d->terms == 3 || d->terms == 16 || d->terms == 1 || d->terms == 2 || d->terms == 17,
                   "conv_nhwc: terms=%d (16, 17, 6, 3, 1 or 2)", d->terms);
    MREFSR_REQUIRE((d->terms != 16 && d->terms != 17) || (d->wscale > 0.f && d->wscale < 3.0e38f),
                   "conv_nhwc: terms=16 / 17 need the wscale the weights were packed with");
    MREFSR_REQUIRE(d->terms != 17 || (d->ksize == 3 && !res_mask && !stat_sum),
                   "conv_nhwc: terms=17 (Winograd F(2x2, 3x3)) is the plain 3x3 forward convolution");
    MREFSR_REQUIRE(d->C1 % 4 == 0 && d->ld1 % 4 == 0 && d->ld1 >= d->C1 && d->N1 > 0,
                   "conv_nhwc: first input C=%d ld=%d N=%d (C, ld multiples of 4)", d->C1, d->ld1, d->N1);
    if (d->C2 > 0) {
        MREFSR_REQUIRE(x2, "conv_nhwc: second input missing");
        MREFSR_REQUIRE(d->C1 % KC == 0, "conv_nhwc: with a second input C1=%d must be a multiple of %d", d->C1, KC);
        MREFSR_REQUIRE(d->C2 % 4 == 0 && d->ld2 % 4 == 0 && d->ld2 >= d->C2 && d->N2 > 0,
                       "conv_nhwc: second input C=%d ld=%d N=%d", d->C2, d->ld2, d->N2);
    }
    MREFSR_REQUIRE(d->epilogue >= 0 && d->epilogue <= 2, "conv_nhwc: epilogue=%d", d->epilogue);
    if (d->epilogue == 1)
        MREFSR_REQUIRE(d->H % 2 == 0 && d->W % 2 == 0 && !pre && !residual, "conv_nhwc: pooled epilogue needs even H, W, no pre/residual");
    if (d->epilogue == 2) MREFSR_REQUIRE(d->Cout % 4 == 0, "conv_nhwc: pixel-shuffle epilogue needs Cout %% 4 == 0");
    MREFSR_REQUIRE(d->ld_out >= (d->epilogue == 2 ? d->Cout / 4 : d->Cout), "conv_nhwc: ld_out=%d too small", d->ld_out);
    MREFSR_REQUIRE(!residual || d->ld_res >= d->Cout, "conv_nhwc: ld_res=%d too small", d->ld_res);
    MREFSR_REQUIRE(!pre || d->pre_N > 0, "conv_nhwc: pre_N=%d", d->pre_N);
    ConvArgs a{};
    a.x1 = x1, a.x2 = x2, a.wp = reinterpret_cast<const unsigned short *>(packed);
    a.bias = bias, a.slope_ptr = slope_ptr, a.pre = pre, a.residual = residual, a.out = out, a.range_flag = range_flag;
    a.in_amax = in_amax;
    a.res_mask = res_mask, a.stat_sum = stat_sum, a.stat_amax = reinterpret_cast<unsigned int *>(stat_amax);
    a.H = d->H, a.W = d->W, a.C1 = d->C1, a.ld1 = d->ld1, a.N1 = d->N1;
    a.C2 = d->C2, a.ld2 = d->C2 > 0 ? d->ld2 : 4, a.N2 = d->C2 > 0 ? d->N2 : 1;
    a.Cout = d->Cout, a.ld_out = d->ld_out, a.ld_res = d->ld_res, a.pre_N = pre ? d->pre_N : 1;
    a.n_ch1 = (d->C1 + KC - 1) / KC;
    a.n_ch = a.n_ch1 + (d->C2 + KC - 1) / KC;
    a.n_cb = (d->Cout + NB - 1) / NB;
    a.act = d->act, a.epilogue = d->epilogue, a.slope = d->slope;
    a.out_scale = (d->terms == 16 || d->terms == 17) ? 1.0f / d->wscale : 1.0f;
    a.dyn_pre = nullptr, a.dyn_mask = nullptr, a.dyn_abs = nullptr, a.dyn_ni = 0;
    a.io16 = d->terms == 2;
    if (a.io16)
        MREFSR_REQUIRE(d->C1 % 8 == 0 && d->ld1 % 8 == 0 && d->C2 % 8 == 0 && (d->C2 == 0 || d->ld2 % 8 == 0),
                       "conv_nhwc(bf16 storage): C1=%d ld1=%d C2=%d ld2=%d must be multiples of 8", d->C1, d->ld1, d->C2, d->ld2);
    MREFSR_REQUIRE(d->N <= 65535, "conv_nhwc: N = %d exceeds the grid limit", d->N);
    return dispatch(a, d, stream);
}
}  // namespace

MREFSR_EXPORT int mrefsr_conv_dynagg_f32(const mrefsr_conv_desc *d, const float *x, const void *packed, const float *bias,
                                         const float *pre_offset, float *offset, float *mask, double *abs_sum, int dg,
                                         int *range_flag, mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(d && x && packed && pre_offset && offset && mask, "conv_dynagg: null pointer");
    MREFSR_REQUIRE(d->N > 0 && d->N <= 65535 && d->H > 0 && d->W > 0 && d->C1 > 0 && d->C2 == 0, "conv_dynagg: N=%d H=%d W=%d C1=%d C2=%d", d->N,
                   d->H, d->W, d->C1, d->C2);
    MREFSR_REQUIRE(dg > 0 && d->Cout == 27 * dg && d->ksize == 3, "conv_dynagg: Cout=%d must be 27 * dg (dg=%d), ksize=%d must be 3", d->Cout, dg,
                   d->ksize);
    MREFSR_REQUIRE(d->terms == 6 || d->terms == 16 || d->terms == 1 || d->terms == 2, "conv_dynagg: terms=%d (16, 6, 1 or 2)", d->terms);
    MREFSR_REQUIRE(d->terms != 16 || (d->wscale > 0.f && d->wscale < 3.0e38f), "conv_dynagg: terms=16 needs the wscale the weights were packed with");
    MREFSR_REQUIRE(d->C1 % 4 == 0 && d->ld1 % 4 == 0 && d->ld1 >= d->C1, "conv_dynagg: input C=%d ld=%d (multiples of 4)", d->C1, d->ld1);
    ConvArgs a{};
    a.x1 = x, a.x2 = nullptr, a.wp = reinterpret_cast<const unsigned short *>(packed);
    a.bias = bias, a.slope_ptr = nullptr, a.pre = nullptr, a.residual = nullptr, a.out = offset, a.range_flag = range_flag;
    a.in_amax = nullptr;
    a.H = d->H, a.W = d->W, a.C1 = d->C1, a.ld1 = d->ld1, a.N1 = d->N, a.C2 = 0, a.ld2 = 4, a.N2 = 1;
    a.Cout = d->Cout, a.ld_out = d->Cout, a.ld_res = 0, a.pre_N = 1;
    a.n_ch1 = (d->C1 + KC - 1) / KC, a.n_ch = a.n_ch1, a.n_cb = (d->Cout + NB - 1) / NB;
    a.act = 0, a.epilogue = 3, a.slope = 0.f, a.out_scale = d->terms == 16 ? 1.0f / d->wscale : 1.0f;
    a.dyn_pre = reinterpret_cast<const float2 *>(pre_offset), a.dyn_mask = mask, a.dyn_abs = abs_sum, a.dyn_ni = 9 * dg;
    a.io16 = d->terms == 2;
    if (a.io16) MREFSR_REQUIRE(d->C1 % 8 == 0 && d->ld1 % 8 == 0, "conv_dynagg(bf16 storage): C1=%d ld1=%d must be multiples of 8", d->C1, d->ld1);
    return dispatch(a, d, stream);
}

#ifdef MREFSR_CONV_STAMP
// read-and-reset of the phase clocks (instrumentation builds only)
MREFSR_EXPORT int mrefsr_dbg_conv_stamps(unsigned long long *out8)
{
    static unsigned long long h[1024][8];
    if (hipDeviceSynchronize() != hipSuccess) return mrefsr::fail(MREFSR_E_LAUNCH, "conv_stamps: sync failed");
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_conv_stamp), sizeof(h)) != hipSuccess) return mrefsr::fail(MREFSR_E_LAUNCH, "conv_stamps: read failed");
    for (int i = 0; i < 8; ++i) out8[i] = 0;
    for (int s = 0; s < 1024; ++s)
        for (int i = 0; i < 8; ++i) out8[i] += h[s][i], h[s][i] = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_conv_stamp), h, sizeof(h)) != hipSuccess) return mrefsr::fail(MREFSR_E_LAUNCH, "conv_stamps: reset failed");
    return 0;
}
#endif

namespace {
int dispatch(const ConvArgs &a, const mrefsr_conv_desc *d, mrefsr_stream_t stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (d->terms == 17) return wino_launch(a, d->N, st);
    if (d->terms == 6) return d->ksize == 3 ? launch<0, 3>(a, d->N, st) : launch<0, 1>(a, d->N, st);
#ifdef MREFSR_AB_KERNELS
    if (d->terms == 3) return d->ksize == 3 ? launch<1, 3>(a, d->N, st) : launch<1, 1>(a, d->N, st);
#else
    if (d->terms == 3) return mrefsr::fail(MREFSR_E_UNSUPPORTED, "conv_nhwc: terms=3 (bf16 two-term split, ~2^-16: experiments) needs a -DMREFSR_AB_KERNELS build");
#endif
    if (d->terms == 1 || d->terms == 2) return d->ksize == 3 ? launch<3, 3>(a, d->N, st) : launch<3, 1>(a, d->N, st);
    return d->ksize == 3 ? launch<2, 3>(a, d->N, st) : launch<2, 1>(a, d->N, st);
}
}  // namespace
