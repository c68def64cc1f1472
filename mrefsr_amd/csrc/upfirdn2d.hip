// basicsr.ops.upfirdn2d: zero-insert upsample, pad/crop, FIR with the flipped kernel, decimate
// (upfirdn2d.py:162-192 is the definition; upfirdn2d_kernel.cu:50-208 the reference's kernels).  HBM-bound:
// (in + out) * sizeof(T) bytes per call.
//
//   tiled kernel (minor == 1, the only layout the callers use: upfirdn2d.py:38 reshapes to (-1, H, W, 1)):
//     a block owns a 16 x 64 output tile of one image plane, stages the input rectangle the tile needs in LDS
//     (coalesced row reads, every input sample fetched once per tile) and gathers only the non-zero taps
//     (ceil(kh/up_y) x ceil(kw/up_x) per output) from there; the flipped FIR sits in LDS too;
//   generic kernel (minor > 1, or a tile whose input rectangle does not fit): one thread per output sample, taps
//     straight from global memory (L1/L2 serve the re-reads).
// T = float / double / __half / __hip_bfloat16 (the reference dispatches AT_DISPATCH_FLOATING_TYPES_AND_HALF,
// upfirdn2d_kernel.cu:312); accumulation in float (double for T = double), taps ascending in (y, x).
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace {

__device__ __forceinline__ int floor_div(int a, int b)
{
    int c = a / b;
    if (c * b > a) --c;
    return c;
}

template <typename T> struct Acc { typedef float type; };
template <> struct Acc<double> { typedef double type; };
template <typename T> __device__ __forceinline__ typename Acc<T>::type ldv(const T *p, size_t i) { return (typename Acc<T>::type)p[i]; }
template <> __device__ __forceinline__ float ldv<__half>(const __half *p, size_t i) { return __half2float(p[i]); }
template <> __device__ __forceinline__ float ldv<__hip_bfloat16>(const __hip_bfloat16 *p, size_t i) { return __bfloat162float(p[i]); }
template <typename T, typename A> __device__ __forceinline__ void stv(T *p, size_t i, A v) { p[i] = (T)v; }
template <typename T, typename A> __device__ __forceinline__ void stv_nt(T *p, size_t i, A v)
{
    T t;
    stv(&t, 0, v);
    if constexpr (sizeof(T) == 2) {
        unsigned short u;
        __builtin_memcpy(&u, &t, 2);
        __builtin_nontemporal_store(u, reinterpret_cast<unsigned short *>(p + i));
    } else {
        __builtin_nontemporal_store(t, p + i);
    }
}
template <> __device__ __forceinline__ void stv<__half, float>(__half *p, size_t i, float v) { p[i] = __float2half(v); }
template <> __device__ __forceinline__ void stv<__hip_bfloat16, float>(__hip_bfloat16 *p, size_t i, float v) { p[i] = __float2bfloat16(v); }

struct UpParams {
    int major, in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y, px0, py0, out_h, out_w;
    int reg_h, reg_w;   // tiled kernel: rows / columns of the staged input rectangle
};

constexpr int TILE_H = 16, TILE_W = 64;
constexpr int TILE_LDS_MAX = 12288;   // staged samples + FIR taps (accumulator-typed words) a block may hold

// first / last input index that contributes to output index o along one axis: taps k with o*down + k - pad0 = i*up
__device__ __forceinline__ int first_in(int o, int down, int pad0, int up) { return floor_div(o * down - pad0 + up - 1, up); }
__device__ __forceinline__ int last_in(int o, int down, int pad0, int up, int k) { return floor_div(o * down - pad0 + k - 1, up); }

template <typename T>
__global__ __launch_bounds__(256) void upfirdn2d_tile_kernel(const T *__restrict__ in, const T *__restrict__ kernel, T *__restrict__ out,
                                                             UpParams p, int tiles_x, int tiles_y)
{
    typedef typename Acc<T>::type A;
    extern __shared__ __attribute__((aligned(8))) unsigned char smem_raw[];
    A *sk = reinterpret_cast<A *>(smem_raw);     // flipped FIR [kh][kw]
    A *sx = sk + p.kh * p.kw;                     // input rectangle [reg_h][reg_w], zero outside the image
    int b = blockIdx.x;
    const int tx = b % tiles_x;
    b /= tiles_x;
    const int ty = b % tiles_y, mj = b / tiles_y;
    const int oy0 = ty * TILE_H, ox0 = tx * TILE_W;
    const int iy_lo = first_in(oy0, p.down_y, p.py0, p.up_y), ix_lo = first_in(ox0, p.down_x, p.px0, p.up_x);
    for (int i = threadIdx.x; i < p.kh * p.kw; i += 256) {
        const int ky = i / p.kw, kx = i - ky * p.kw;
        sk[i] = ldv(kernel, (size_t)(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx));
    }
    const T *plane = in + (size_t)mj * p.in_h * p.in_w;
    for (int i = threadIdx.x; i < p.reg_h * p.reg_w; i += 256) {
        const int ry = i / p.reg_w, rx = i - ry * p.reg_w;
        const int iy = iy_lo + ry, ix = ix_lo + rx;
        sx[i] = (iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w) ? ldv(plane, (size_t)iy * p.in_w + ix) : (A)0;
    }
    __syncthreads();
    const int lx = threadIdx.x & (TILE_W - 1), ly0 = threadIdx.x / TILE_W;
    const int ox = ox0 + lx;
    if (ox >= p.out_w) return;
    const int bx = ox * p.down_x - p.px0;
    const int ix0 = first_in(ox, p.down_x, p.px0, p.up_x), ix1 = last_in(ox, p.down_x, p.px0, p.up_x, p.kw);
    for (int ly = ly0; ly < TILE_H; ly += 256 / TILE_W) {
        const int oy = oy0 + ly;
        if (oy >= p.out_h) break;
        const int by = oy * p.down_y - p.py0;
        const int iy0 = first_in(oy, p.down_y, p.py0, p.up_y), iy1 = last_in(oy, p.down_y, p.py0, p.up_y, p.kh);
        A v = 0;
        for (int iy = iy0; iy <= iy1; ++iy) {
            const A *row = sx + (iy - iy_lo) * p.reg_w - ix_lo;
            const A *krow = sk + (iy * p.up_y - by) * p.kw - bx;
            for (int ix = ix0; ix <= ix1; ++ix) v += row[ix] * krow[ix * p.up_x];
        }
        stv(out, ((size_t)mj * p.out_h + oy) * p.out_w + ox, v);
    }
}

// StyleGAN2's cases ([1,3,3,1] FIR: 4 x 4 taps; blur, up x2, down x2) with the rates and the tap count as compile-time
// constants.  Measured on the generic tile kernel: the same 0.4 ms for fp32 and fp16 -- not HBM; a first version with four
// CONSECUTIVE outputs per thread was no faster because lanes 4 words apart read LDS with a 4-way bank conflict.  So:
//   * a wave's 64 lanes own 64 consecutive output columns (conflict-free ds_read_b32; stride 2 for down x2), a thread
//     computes XN columns 64 apart and ROWS CONSECUTIVE rows, so the fully unrolled tap loops share the input rows between
//     vertically adjacent outputs (blur: 11 x 4 reads for 8 rows instead of 8 x 16);
//   * up x2 touches only the 2 x 2 taps whose phase matches the output (selected with v_cndmask, no dynamic register index);
//   * taps in registers; the input rectangle is staged with a division-free 2-D loop.
#ifndef UP2_ROWS
#define UP2_ROWS 16
#endif
template <int UP, int DOWN, int QX = 1> struct FastTile {
    // QX: column pairs per lane of the up x2 quad path (2 for the 2-byte types: the same bytes per lane as a 4-byte type)
    static constexpr int W = (UP == 2 ? 128 * QX : 128 / DOWN), XN = W / 64, ROWS = (UP == 2 ? UP2_ROWS : 8) / DOWN, H = 4 * ROWS;   // (up x2: rows per wave, see the quad path)
    // staged input rectangle (worst case over the phase of the tile origin), row pitch RW words
    static constexpr int RH = ((H - 1) * DOWN + 3) / UP + 2, RW = ((W - 1) * DOWN + 3) / UP + 2;
};

template <typename T, int UP, int DOWN, int K>
__global__ __launch_bounds__(256) void upfirdn2d_fast_kernel(const T *__restrict__ in, const T *__restrict__ kernel, T *__restrict__ out, UpParams p,
                                                             int tiles_x, int tiles_y)
{
    static_assert(K == 4 && (UP == 1 || (UP == 2 && DOWN == 1)), "specialised for StyleGAN2's resamplers");
    typedef typename Acc<T>::type A;
    constexpr int QX = (UP == 2 && sizeof(T) == 2) ? 2 : 1;
    typedef FastTile<UP, DOWN, QX> FT;
    extern __shared__ __attribute__((aligned(8))) unsigned char smem_raw[];
    A *sx = reinterpret_cast<A *>(smem_raw);   // input rectangle [reg_h][reg_w], zero outside the image
    int b = blockIdx.x;
    const int tx = b % tiles_x;
    b /= tiles_x;
    const int ty = b % tiles_y, mj = b / tiles_y;
    const int oy0 = ty * FT::H, ox0 = tx * FT::W;
    const int iy_lo = first_in(oy0, DOWN, p.py0, UP), ix_lo = first_in(ox0, DOWN, p.px0, UP);
    const T *plane = in + (size_t)mj * p.in_h * p.in_w;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    {
        // compile-time trip counts: every load of the rectangle is issued before the first one is waited for (a run-time
        // loop serialises ~30 HBM latencies per wave -- measured: 3x the kernel's time)
        constexpr int NR = (FT::RH + 3) / 4, NC = (FT::RW + 63) / 64;
        T v[NR][NC];   // raw samples from CLAMPED addresses: unconditional loads, no branch (and no wait) between them
#pragma unroll
        for (int a = 0; a < NR; ++a) {
            const int iy = iy_lo + wv + 4 * a;
            const T *src = plane + (size_t)(iy < 0 ? 0 : (iy >= p.in_h ? p.in_h - 1 : iy)) * p.in_w;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int ix = ix_lo + lane + 64 * c;
                v[a][c] = src[ix < 0 ? 0 : (ix >= p.in_w ? p.in_w - 1 : ix)];
            }
        }
#pragma unroll
        for (int a = 0; a < NR; ++a)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int ry = wv + 4 * a, rx = lane + 64 * c, iy = iy_lo + ry, ix = ix_lo + rx;
                const bool inside = iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w;
                if (ry < FT::RH && rx < FT::RW) sx[ry * FT::RW + rx] = inside ? ldv(&v[a][c], 0) : (A)0;
            }
    }
    A kf[K][K];   // flipped FIR
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int kx = 0; kx < K; ++kx) kf[ky][kx] = ldv(kernel, (size_t)(K - 1 - ky) * K + (K - 1 - kx));
    __syncthreads();
    const int oyb = oy0 + wv * FT::ROWS;
    if constexpr (UP == 2) {
        // up x2: a thread owns a column PAIR (2 lane, 2 lane + 1) of the wave's ROWS rows = ROWS / 2 output quads (2 x 2).  The quads of
        // a thread share their input rows ((ROWS / 2 + 2) rows x 3 columns of LDS words; one output at a time read 4 each), the
        // tap weights of the four output phases are chosen once per thread, and a row's pair leaves as one 8-byte store.  The
        // sums keep the order of the generic path (taps ascending in y, then x): the same bits.
        const int ox = ox0 + 2 * QX * lane;
        if (ox >= p.out_w) return;
        // tile origins are even, so which taps and which input columns / rows an output phase uses depends on the parity of the
        // pads alone: four instantiations of the body with the selections as constants (run-time selects were 12 of the 16 VALU
        // instructions per output)
        auto body = [&](auto pxo_t, auto pyo_t) {
            constexpr bool PXO = decltype(pxo_t)::value, PYO = decltype(pyo_t)::value;
            // column phase b: taps kx = (b ^ PXO) + {0, 2}, first input column c0 + (b && !PXO); rows alike
            const int c0 = ((ox - p.px0 + (PXO ? 1 : 0)) >> 1) - ix_lo, r0 = ((oyb - p.py0 + (PYO ? 1 : 0)) >> 1) - iy_lo;
            constexpr int NQ = FT::ROWS / 2;   // quad rows per thread
            A in[NQ + 2][2 * QX + 1];
#pragma unroll
            for (int r = 0; r < NQ + 2; ++r)
#pragma unroll
                for (int c = 0; c < 2 * QX + 1; ++c) {
                    const int rr = r0 + r, cc = c0 + c;
                    in[r][c] = (rr < FT::RH && cc < FT::RW) ? sx[rr * FT::RW + cc] : (A)0;
                }
            const int nvalid = p.out_w - ox;   // columns of this lane inside the image (>= 1)
#pragma unroll
            for (int qr = 0; qr < NQ; ++qr)
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int oy = oyb + 2 * qr + a;
                    if (oy >= p.out_h) return;
                    constexpr int dyA[2] = {0, PYO ? 0 : 1};
                    const int ky = a ^ (PYO ? 1 : 0), ri = qr + dyA[a];
                    A o[2 * QX];
#pragma unroll
                    for (int b2 = 0; b2 < 2 * QX; ++b2) {
                        constexpr int dxB[2] = {0, PXO ? 0 : 1};
                        const int kx = (b2 & 1) ^ (PXO ? 1 : 0), ci = (b2 >> 1) + dxB[b2 & 1];
                        A acc = in[ri][ci] * kf[ky][kx];
                        acc += in[ri][ci + 1] * kf[ky][kx + 2];
                        acc += in[ri + 1][ci] * kf[ky + 2][kx];
                        acc += in[ri + 1][ci + 1] * kf[ky + 2][kx + 2];
                        o[b2] = acc;
                    }
                    T *dst = out + ((size_t)mj * p.out_h + oy) * p.out_w + ox;
                    if (nvalid >= 2 * QX && ((reinterpret_cast<size_t>(dst) & 7) == 0)) {   // the lane's columns as one 8-byte store
                        if constexpr (sizeof(T) == 4) {
                            typedef float f32x2 __attribute__((ext_vector_type(2)));
                            __builtin_nontemporal_store(f32x2{(float)o[0], (float)o[1]}, reinterpret_cast<f32x2 *>(dst));
                            continue;
                        }
                        if constexpr (sizeof(T) == 2) {
                            typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                            T q4[4];
#pragma unroll
                            for (int b2 = 0; b2 < 4; ++b2) stv(q4, b2, o[b2]);
                            u32x2 u;
                            __builtin_memcpy(&u, q4, 8);
                            __builtin_nontemporal_store(u, reinterpret_cast<u32x2 *>(dst));
                            continue;
                        }
                    }
#pragma unroll
                    for (int b2 = 0; b2 < 2 * QX; ++b2)
                        if (b2 < nvalid) stv_nt(dst, b2, o[b2]);
                }
        };
        typedef std::integral_constant<bool, true> yes;
        typedef std::integral_constant<bool, false> no;
        if (p.px0 & 1) {
            if (p.py0 & 1) body(yes{}, yes{});
            else body(yes{}, no{});
        } else {
            if (p.py0 & 1) body(no{}, yes{});
            else body(no{}, no{});
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < FT::XN; ++j) {
        const int ox = ox0 + lane + 64 * j;
        if (ox >= p.out_w) break;
        const int bx = ox * DOWN - p.px0;
        if constexpr (UP == 1) {
            // rows of the rectangle this thread's ROWS outputs read: (oyb + r) * DOWN - py0 + ky - iy_lo, r and ky constants
            const A *base = sx + (oyb * DOWN - p.py0 - iy_lo) * FT::RW + (bx - ix_lo);
            A acc[FT::ROWS];
#pragma unroll
            for (int r = 0; r < FT::ROWS; ++r) acc[r] = 0;
#pragma unroll
            for (int rr = 0; rr < (FT::ROWS - 1) * DOWN + K; ++rr) {   // input row of the window, each read once
                A v[K];
#pragma unroll
                for (int kx = 0; kx < K; ++kx) v[kx] = base[rr * FT::RW + kx];
#pragma unroll
                for (int r = 0; r < FT::ROWS; ++r) {
                    const int ky = rr - r * DOWN;
                    if (ky < 0 || ky >= K) continue;
#pragma unroll
                    for (int kx = 0; kx < K; ++kx) acc[r] += v[kx] * kf[ky][kx];
                }
            }
#pragma unroll
            for (int r = 0; r < FT::ROWS; ++r)
                if (oyb + r < p.out_h) stv_nt(out, ((size_t)mj * p.out_h + oyb + r) * p.out_w + ox, acc[r]);
        } else {
            const int kx0 = bx & 1, cx = ((bx + kx0) >> 1) - ix_lo;
            A wx[K][2];   // the two horizontal taps of this column's phase, per tap row
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                wx[ky][0] = kx0 ? kf[ky][1] : kf[ky][0];
                wx[ky][1] = kx0 ? kf[ky][3] : kf[ky][2];
            }
#pragma unroll
            for (int r = 0; r < FT::ROWS; ++r) {
                const int oy = oyb + r;
                if (oy >= p.out_h) break;
                const int by = oy - p.py0, ky0 = by & 1;
                const A *row = sx + (((by + ky0) >> 1) - iy_lo) * FT::RW + cx;
                const A w00 = ky0 ? wx[1][0] : wx[0][0], w01 = ky0 ? wx[1][1] : wx[0][1];
                const A w10 = ky0 ? wx[3][0] : wx[2][0], w11 = ky0 ? wx[3][1] : wx[2][1];
                A acc = row[0] * w00;
                acc += row[1] * w01;
                acc += row[FT::RW] * w10;
                acc += row[FT::RW + 1] * w11;
                stv_nt(out, ((size_t)mj * p.out_h + oy) * p.out_w + ox, acc);
            }
        }
    }
}

// 2-byte types, UP == 1 (blur, down x2) with 4 x 4 taps: a lane owns a COLUMN PAIR.  The one-column-per-lane body above runs the
// 2-byte types at 1.35x the fp32 kernel's element rate and no further: 16 FMAs + 5.5 LDS words + a 2-byte load and a 2-byte store
// per output fill the VALU.  Here a pair's sixteen taps are eight v_pk_fma_f32 (operand pairs = two LDS words DOWN apart, one ds_read2_b32), inputs arrive as 4-byte loads of two samples (the staged
// rectangle starts at an EVEN input column and in_w is even: a pair is inside the image or outside it as a whole), a row's pair
// leaves as one 4-byte store.  Same tap order per output as every other path
// (ky ascending, kx ascending; each step one fma): the same bits.
#ifndef UP_PAIR_ROWS
#define UP_PAIR_ROWS 8
#endif
template <int DOWN> struct PairTile {
    static constexpr int W = 256 / DOWN, XN = W / 128, ROWS = UP_PAIR_ROWS / DOWN, H = 4 * ROWS;
    static constexpr int RH = (H - 1) * DOWN + 4, RW = ((W - 1) * DOWN + 4 + 1 + 1) & ~1, NP = RW / 2;   // staged rows, columns (from an even column), pairs per row
};

template <typename T, int DOWN>
__global__ __launch_bounds__(256) void upfirdn2d_pair_kernel(const T *__restrict__ in, const T *__restrict__ kernel, T *__restrict__ out, UpParams p,
                                                             int tiles_x, int tiles_y, int tiles_per_block)
{
    static_assert(sizeof(T) == 2 && (DOWN == 1 || DOWN == 2), "2-byte types, blur / down x2");
    typedef PairTile<DOWN> PT;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    constexpr int K = 4;
    extern __shared__ __attribute__((aligned(8))) unsigned char smem_raw[];
    float *sx = reinterpret_cast<float *>(smem_raw);   // [RH][RW] input rectangle as floats, zero outside the image
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // A block walks `tiles_per_block` consecutive tiles (down a tile column, on into the next plane) and requests tile t + 1's input
    // before it computes tile t, so that a block has loads in flight while it computes.  (What was measured, fp16 blur of 8 x 256 planes
    // of 256 x 256: one column per lane 144-148 us; column pairs 138; operand pairs by ds_read2_b32 instead of register moves 134.5; taps
    // in SGPR pairs (126 -> 66 VGPRs) 140; 16-row tiles for eight resident blocks 155; this loop 134 at two tiles per block.  The
    // instruction count, the occupancy and the pipelining move it by a few per cent: at 4.0 TB/s the kernel is at 80 % of the fp32
    // kernel's byte rate, and a plain copy reaches 5.0 on this chip.)
    const long total = (long)tiles_x * tiles_y * p.major;
    const long id0 = (long)blockIdx.x * tiles_per_block;
    struct Tile { int mj, oy0, ox0, iy_lo, ixe; };
    auto decode = [&](const long id) {
        Tile t;
        const int ty = (int)(id % tiles_y);
        const long r = id / tiles_y;
        const int tx = (int)(r % tiles_x);
        t.mj = (int)(r / tiles_x);
        t.oy0 = ty * PT::H, t.ox0 = tx * PT::W;
        t.iy_lo = t.oy0 * DOWN - p.py0;
        const int ix_lo = t.ox0 * DOWN - p.px0;
        t.ixe = ix_lo - (ix_lo & 1);   // (two's complement: -1 & 1 = 1, ixe = -2)
        return t;
    };
    // thread = (row wv + 4 a, pairs lane and lane + 64) + one of the 2 RH leftover pairs (NP = 130): every load issued before the first
    // is used, clamped addresses, no branch; row pointers and column clamps are computed once
    constexpr int NR = (PT::RH + 3) / 4;
    static_assert(PT::NP == 130 && 2 * PT::RH <= 256, "staging map");
    unsigned int v[NR][2], vx;
    const int xrow = threadIdx.x >> 1, xpr = 128 + (threadIdx.x & 1);   // the leftover pair of this thread (threads < 2 RH)
    auto issue = [&](const Tile &t) {
        const T *plane = in + (size_t)t.mj * p.in_h * p.in_w;
        int cxs[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int ix = t.ixe + 2 * (lane + 64 * c);
            cxs[c] = ix < 0 ? 0 : (ix >= p.in_w ? p.in_w - 2 : ix);
        }
        {
            const int iy = t.iy_lo + xrow, ix = t.ixe + 2 * xpr;
            const int cy = iy < 0 ? 0 : (iy >= p.in_h ? p.in_h - 1 : iy), cx = ix < 0 ? 0 : (ix >= p.in_w ? p.in_w - 2 : ix);
            vx = *reinterpret_cast<const unsigned int *>(plane + (size_t)cy * p.in_w + cx);
        }
#pragma unroll
        for (int a = 0; a < NR; ++a) {
            const int iy = t.iy_lo + wv + 4 * a;
            const T *src = plane + (size_t)(iy < 0 ? 0 : (iy >= p.in_h ? p.in_h - 1 : iy)) * p.in_w;
#pragma unroll
            for (int c = 0; c < 2; ++c) v[a][c] = *reinterpret_cast<const unsigned int *>(src + cxs[c]);
        }
    };
    auto put = [&](const unsigned int raw, const bool inside, const int row, const int pr) {
        T two[2];
        __builtin_memcpy(two, &raw, 4);
        const float q0 = inside ? ldv(two, 0) : 0.f, q1 = inside ? ldv(two, 1) : 0.f;
        *reinterpret_cast<f32x2 *>(sx + row * PT::RW + 2 * pr) = f32x2{q0, q1};
    };
    auto commit = [&](const Tile &t) {
        bool in_x[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int ix = t.ixe + 2 * (lane + 64 * c);
            in_x[c] = ix >= 0 && ix < p.in_w;
        }
#pragma unroll
        for (int a = 0; a < NR; ++a) {
            const int row = wv + 4 * a, iy = t.iy_lo + row;
            const bool in_y = iy >= 0 && iy < p.in_h;
            if (row < PT::RH) {
#pragma unroll
                for (int c = 0; c < 2; ++c) put(v[a][c], in_y && in_x[c], row, lane + 64 * c);
            }
        }
        if (xrow < PT::RH) {
            const int iy = t.iy_lo + xrow, ix = t.ixe + 2 * xpr;
            put(vx, iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w, xrow, xpr);
        }
    };
    unsigned long long kf[K][K];   // flipped FIR, each tap twice in a scalar register pair (the packed FMA's broadcast operand)
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const unsigned int kb = __builtin_amdgcn_readfirstlane(__float_as_uint(ldv(kernel, (size_t)(K - 1 - ky) * K + (K - 1 - kx))));
            kf[ky][kx] = ((unsigned long long)kb << 32) | kb;
        }
    const int ry0 = wv * PT::ROWS * DOWN;   // first staged row of this wave's outputs
    auto compute = [&](const Tile &t, auto d_t) {
        constexpr int D = decltype(d_t)::value;
        const int oyb = t.oy0 + wv * PT::ROWS;
#pragma unroll
        for (int j = 0; j < PT::XN; ++j) {
            const int ox = t.ox0 + 2 * lane + 128 * j;
            if (ox >= p.out_w) break;
            const int P = 2 * lane + 128 * j;   // (= ox - ox0: the pair's window starts at word P * DOWN + D of a staged row)
            f32x2 acc[PT::ROWS];
#pragma unroll
            for (int r = 0; r < PT::ROWS; ++r) acc[r] = f32x2{0.f, 0.f};
#pragma unroll
            for (int rr = 0; rr < (PT::ROWS - 1) * DOWN + K; ++rr) {
                // X[kx] = the tap column's input for the pair's two outputs: words D + kx and D + kx + DOWN of the window -- one
                // ds_read2_b32 each (two dwords at independent offsets: no register shuffling for the pairs that start at an odd word)
                f32x2 X[K];
                const float *row = sx + (ry0 + rr) * PT::RW + P * DOWN;
#pragma unroll
                for (int kx = 0; kx < K; ++kx) X[kx] = f32x2{row[D + kx], row[D + kx + DOWN]};
#pragma unroll
                for (int r = 0; r < PT::ROWS; ++r) {
                    const int ky = rr - r * DOWN;
                    if (ky < 0 || ky >= K) continue;
#pragma unroll
                    for (int kx = 0; kx < K; ++kx) asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[r]) : "s"(kf[ky][kx]), "v"(X[kx]));
                }
            }
#pragma unroll
            for (int r = 0; r < PT::ROWS; ++r) {
                const int oy = oyb + r;
                if (oy >= p.out_h) break;
                T q2[2];
                stv(q2, 0, acc[r].x);
                stv(q2, 1, acc[r].y);
                unsigned int u;
                __builtin_memcpy(&u, q2, 4);
                __builtin_nontemporal_store(u, reinterpret_cast<unsigned int *>(out + ((size_t)t.mj * p.out_h + oy) * p.out_w + ox));   // (out_w even)
            }
        }
    };
    if (id0 >= total) return;
    Tile cur = decode(id0);
    issue(cur);
    for (int i = 0; i < tiles_per_block; ++i) {
        const long id = id0 + i;
        if (id >= total) break;
        commit(cur);   // (waits for the tile's loads)
        __syncthreads();
        Tile nxt = cur;
        if (i + 1 < tiles_per_block && id + 1 < total) {
            nxt = decode(id + 1);
            issue(nxt);
        }
        asm volatile("" ::: "memory");   // the next tile's requests stay in front of this tile's arithmetic
        if (p.px0 & 1) compute(cur, std::integral_constant<int, 1>{});   // (tile origins are even: the window's parity is the pad's)
        else compute(cur, std::integral_constant<int, 0>{});
        __syncthreads();   // the rectangle has been read: the next tile may overwrite it
        cur = nxt;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void upfirdn2d_kernel(const T *__restrict__ in, const T *__restrict__ kernel, T *__restrict__ out, UpParams p)
{
    typedef typename Acc<T>::type A;
    extern __shared__ __attribute__((aligned(8))) unsigned char smem_raw[];
    A *sk = reinterpret_cast<A *>(smem_raw);  // flipped FIR
    for (int i = threadIdx.x; i < p.kh * p.kw; i += blockDim.x) {
        const int ky = i / p.kw, kx = i - ky * p.kw;
        sk[i] = ldv(kernel, (size_t)(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx));
    }
    __syncthreads();
    const long total = (long)p.major * p.out_h * p.out_w * p.minor;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int mn = (int)(e % p.minor);
        long t = e / p.minor;
        const int ox = (int)(t % p.out_w);
        t /= p.out_w;
        const int oy = (int)(t % p.out_h), mj = (int)(t / p.out_h);
        const int by = oy * p.down_y - p.py0, bx = ox * p.down_x - p.px0;
        int iy0 = first_in(oy, p.down_y, p.py0, p.up_y), iy1 = last_in(oy, p.down_y, p.py0, p.up_y, p.kh);
        int ix0 = first_in(ox, p.down_x, p.px0, p.up_x), ix1 = last_in(ox, p.down_x, p.px0, p.up_x, p.kw);
        iy0 = iy0 < 0 ? 0 : iy0;
        ix0 = ix0 < 0 ? 0 : ix0;
        iy1 = iy1 > p.in_h - 1 ? p.in_h - 1 : iy1;
        ix1 = ix1 > p.in_w - 1 ? p.in_w - 1 : ix1;
        A v = 0;
        for (int iy = iy0; iy <= iy1; ++iy) {
            const size_t row = (((size_t)mj * p.in_h + iy) * p.in_w) * p.minor + mn;
            for (int ix = ix0; ix <= ix1; ++ix) v += ldv(in, row + (size_t)ix * p.minor) * sk[(iy * p.up_y - by) * p.kw + ix * p.up_x - bx];
        }
        stv(out, (size_t)e, v);
    }
}

template <typename T>
int launch_upfirdn2d(const void *in, const void *kernel, void *out, UpParams p, hipStream_t st)
{
    typedef typename Acc<T>::type A;
    // input rectangle of a full tile: rows first_in(oy0) .. last_in(oy0 + TILE_H - 1); its extent does not depend on oy0
    // beyond the rounding, so take the worst case over the phase of oy0 * down modulo up
    const int reg_h = ((TILE_H - 1) * p.down_y + p.kh - 1) / p.up_y + 2, reg_w = ((TILE_W - 1) * p.down_x + p.kw - 1) / p.up_x + 2;
    if constexpr (sizeof(T) == 2) {   // blur / down x2 of the 2-byte types: column pairs per lane (4-byte loads and stores)
        if (p.minor == 1 && p.kh == 4 && p.kw == 4 && p.up_x == 1 && p.up_y == 1 && p.down_x == p.down_y && p.down_x <= 2 && p.in_w % 2 == 0 &&
            p.in_w >= 2 && p.out_w % 2 == 0 && (reinterpret_cast<uintptr_t>(in) & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 3) == 0) {
            const int tw = p.down_x == 1 ? PairTile<1>::W : PairTile<2>::W, th = p.down_x == 1 ? PairTile<1>::H : PairTile<2>::H;
            const int tiles_x = mrefsr::cdiv(p.out_w, tw), tiles_y = mrefsr::cdiv(p.out_h, th);
            const long tiles = (long)tiles_x * tiles_y * p.major;
            // tiles per block: 2 once the launch has more than four rounds of resident blocks (measured at 16384 tiles: 1 / 2 / 4 / 8 tiles per
            // block = 140 / 134 / 135 / 141 us for the blur, 98 / 98 / 105 / 114 for down x2); MREFSR_UP_TPB overrides (A/B)
            static const int tpb_env = getenv("MREFSR_UP_TPB") ? atoi(getenv("MREFSR_UP_TPB")) : 0;
            int tpb = tpb_env > 0 ? tpb_env : (tiles >= 256L * 4 * 4 ? 2 : 1);
            tpb = tpb < 1 ? 1 : (tpb > 8 ? 8 : tpb);
            const long blocks = (tiles + tpb - 1) / tpb;
            if (blocks < 0x7fffffffL) {
                if (p.down_x == 1)
                    hipLaunchKernelGGL((upfirdn2d_pair_kernel<T, 1>), dim3((unsigned)blocks), dim3(256), (size_t)PairTile<1>::RH * PairTile<1>::RW * 4, st,
                                       (const T *)in, (const T *)kernel, (T *)out, p, tiles_x, tiles_y, tpb);
                else
                    hipLaunchKernelGGL((upfirdn2d_pair_kernel<T, 2>), dim3((unsigned)blocks), dim3(256), (size_t)PairTile<2>::RH * PairTile<2>::RW * 4, st,
                                       (const T *)in, (const T *)kernel, (T *)out, p, tiles_x, tiles_y, tpb);
                return mrefsr::check_launch("upfirdn2d(pair)");
            }
        }
    }
    if (p.minor == 1 && p.kh == 4 && p.kw == 4 && p.up_x == p.up_y && p.down_x == p.down_y &&
        ((p.up_x <= 2 && p.down_x == 1) || (p.up_x == 1 && p.down_x == 2))) {
#define MREFSR_UPFIRDN_FAST(U, D)                                                                                                       \
    {                                                                                                                                   \
        typedef FastTile<U, D, (U == 2 && sizeof(T) == 2) ? 2 : 1> FT;                                                                                                  \
        p.reg_h = FT::RH;                                                                                                               \
        p.reg_w = FT::RW;                                                                                                               \
        const int tiles_x = mrefsr::cdiv(p.out_w, FT::W), tiles_y = mrefsr::cdiv(p.out_h, FT::H);                                       \
        const long blocks = (long)tiles_x * tiles_y * p.major;                                                                          \
        if (blocks < 0x7fffffffL) {                                                                                                     \
            hipLaunchKernelGGL((upfirdn2d_fast_kernel<T, U, D, 4>), dim3((unsigned)blocks), dim3(256),                                  \
                               (size_t)p.reg_h * p.reg_w * sizeof(A), st, (const T *)in, (const T *)kernel, (T *)out, p, tiles_x, tiles_y); \
            return mrefsr::check_launch("upfirdn2d(fast)");                                                                             \
        }                                                                                                                               \
    }
        if (p.up_x == 1 && p.down_x == 1) MREFSR_UPFIRDN_FAST(1, 1)
        if (p.up_x == 2 && p.down_x == 1) MREFSR_UPFIRDN_FAST(2, 1)
        if (p.up_x == 1 && p.down_x == 2) MREFSR_UPFIRDN_FAST(1, 2)
#undef MREFSR_UPFIRDN_FAST
    }
    if (p.minor == 1 && (long)reg_h * reg_w + (long)p.kh * p.kw <= TILE_LDS_MAX) {
        p.reg_h = reg_h;
        p.reg_w = reg_w;
        const int tiles_x = mrefsr::cdiv(p.out_w, TILE_W), tiles_y = mrefsr::cdiv(p.out_h, TILE_H);
        const long blocks = (long)tiles_x * tiles_y * p.major;
        if (blocks < 0x7fffffffL) {
            hipLaunchKernelGGL(upfirdn2d_tile_kernel<T>, dim3((unsigned)blocks), dim3(256), (size_t)(reg_h * reg_w + p.kh * p.kw) * sizeof(A), st,
                               (const T *)in, (const T *)kernel, (T *)out, p, tiles_x, tiles_y);
            return mrefsr::check_launch("upfirdn2d(tile)");
        }
    }
    const long total = (long)p.major * p.out_h * p.out_w * p.minor;
    const long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(upfirdn2d_kernel<T>, dim3((int)(blocks < 16384 ? blocks : 16384)), dim3(256), (size_t)p.kh * p.kw * sizeof(A), st,
                       (const T *)in, (const T *)kernel, (T *)out, p);
    return mrefsr::check_launch("upfirdn2d");
}

}  // namespace

MREFSR_EXPORT int mrefsr_upfirdn2d(const void *in, const void *kernel, void *out, int major, int in_h, int in_w, int minor, int kh,
                                   int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                                   int dtype, mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(in && kernel && out, "upfirdn2d: null pointer");
    MREFSR_REQUIRE(major > 0 && in_h > 0 && in_w > 0 && minor > 0 && kh > 0 && kw > 0, "upfirdn2d: bad sizes");
    MREFSR_REQUIRE(up_x > 0 && up_y > 0 && down_x > 0 && down_y > 0, "upfirdn2d: up/down must be positive");
    MREFSR_REQUIRE(dtype >= 0 && dtype <= 3, "upfirdn2d: dtype=%d (0 f32, 1 f16, 2 bf16, 3 f64)", dtype);
    UpParams p;
    p.major = major; p.in_h = in_h; p.in_w = in_w; p.minor = minor; p.kh = kh; p.kw = kw;
    p.up_x = up_x; p.up_y = up_y; p.down_x = down_x; p.down_y = down_y; p.px0 = pad_x0; p.py0 = pad_y0;
    p.out_h = (in_h * up_y + pad_y0 + pad_y1 - kh + down_y) / down_y;
    p.out_w = (in_w * up_x + pad_x0 + pad_x1 - kw + down_x) / down_x;
    p.reg_h = p.reg_w = 0;
    MREFSR_REQUIRE(p.out_h > 0 && p.out_w > 0, "upfirdn2d: empty output (%d x %d)", p.out_h, p.out_w);
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
    case 0: return launch_upfirdn2d<float>(in, kernel, out, p, st);
    case 1: return launch_upfirdn2d<__half>(in, kernel, out, p, st);
    case 2: return launch_upfirdn2d<__hip_bfloat16>(in, kernel, out, p, st);
    default: return launch_upfirdn2d<double>(in, kernel, out, p, st);
    }
}

MREFSR_EXPORT int mrefsr_upfirdn2d_f32(const float *in, const float *kernel, float *out, int major, int in_h, int in_w,
                                       int minor, int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0,
                                       int pad_x1, int pad_y0, int pad_y1, mrefsr_stream_t stream)
{
    return mrefsr_upfirdn2d(in, kernel, out, major, in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0,
                            pad_y1, 0, stream);
}
