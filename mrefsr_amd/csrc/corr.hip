// Correlation path of MRefSR on gfx950: pixel normalisation, patch norms, fused 3x3-patch
// correlation + top-1 (MFMA pixel-Gram + LDS box-sum, the correlation matrix never leaves the CU),
// and index -> offset planes.
//
// Replaces basicsr/archs/ref_map_util.py:4-86 and corres_generation_arch.py:30-105 of the
// reference (see include/mrefsr_hip.h for the per-entry citations).
//
// Arithmetic contract (bit-exact with oracle/mrefsr_oracle.c, compile with -ffp-contract=off):
//   G[p,s]   = fmaf chain over channels ascending, from +0   (v_mfma_f32_32x32x2_f32 is exactly
//              D = fma(a_k1, b_k1, fma(a_k0, b_k0, C)); k0 <- even channel, k1 <- odd channel)
//   raw[q,r] = 8 sequential fp32 adds of G[q+d, r+d] over the 3x3 taps d in row-major order
//   corr     = raw * inv_ref[r];   best: v > best or (v == best and r < idx)
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------
// pixnorm: x [N][C][HW] -> y [N][HW][Cp] (split even/odd layout), n2 [N][HW]
// One block = 64 pixels.  The per-pixel chains are sequential by contract (c ascending); the
// kernel is HBM-bound (reads C*4 B, writes Cp*4 B per pixel), the chain runs out of LDS.
// ---------------------------------------------------------------------------------------------
constexpr int PN_PIX = 64;
constexpr int PN_LD = PN_PIX + 1;

__global__ __launch_bounds__(256) void pixnorm_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                      float *__restrict__ n2, unsigned short *__restrict__ ybf, int C, int Cp, int HW,
                                                      int normalize, int x_nhwc, int ybf_fmt, float *__restrict__ d2)
{
    extern __shared__ __attribute__((aligned(16))) float tile[];  // [C][65]
    const int tid = threadIdx.x;
    const int n = blockIdx.y;
    const int p0 = blockIdx.x * PN_PIX;
    const float *xs = x + (size_t)n * C * HW;
    const int pix = tid & 63, grp = tid >> 6;
    const bool pvalid = p0 + pix < HW;
    if (x_nhwc == 2) {  // channels-last bf16 input (2-byte storage of BASELINE configs[4])
        const unsigned short *xr = reinterpret_cast<const unsigned short *>(x) + ((size_t)n * HW + p0) * C;
        const int npx = HW - p0 < PN_PIX ? HW - p0 : PN_PIX;
        for (int e = tid; e < PN_PIX * C; e += 256) {
            const int px = e / C, c = e - px * C;
            tile[c * PN_LD + px] = px < npx ? __uint_as_float((unsigned int)xr[e] << 16) : 0.0f;
        }
    } else if (x_nhwc && (C & (C - 1)) == 0 && C >= 64 && C <= 256 && HW - p0 >= PN_PIX) {
        // channels-last input, a whole tile, C a power of two (the path's 256 / 128 / 64): the 64 x C block is contiguous -- every thread
        // requests its C / 16 sixteen-byte pieces AT ONCE (the whole 64 KB of a C = 256 block in flight: one memory round trip per block;
        // the general path below has eight of them with four bytes per lane) and scatters them into the [channel][pixel] tile
        const float4 *xr4 = reinterpret_cast<const float4 *>(x + ((size_t)n * HW + p0) * C);
        const int qs = 31 - __builtin_clz(C >> 2), np = C >> 4;   // pieces per pixel = 2^qs; pieces per thread
        float4 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i < np) v[i] = xr4[tid + 256 * i];
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i < np) {
                const int e = tid + 256 * i, px = e >> qs, c4 = (e & ((1 << qs) - 1)) * 4;
                tile[(c4 + 0) * PN_LD + px] = v[i].x;
                tile[(c4 + 1) * PN_LD + px] = v[i].y;
                tile[(c4 + 2) * PN_LD + px] = v[i].z;
                tile[(c4 + 3) * PN_LD + px] = v[i].w;
            }
    } else if (x_nhwc) {  // channels-last input: the 64 pixels x C block is contiguous
        const float *xr = x + ((size_t)n * HW + p0) * C;
        const int npx = HW - p0 < PN_PIX ? HW - p0 : PN_PIX;
        // eight coalesced 1-KB rows requested before the first LDS store (one load per iteration left every round trip
        // exposed: 64 of them per block, 44 us of a block's life)
        const int total = PN_PIX * C;
        for (int e0 = tid; e0 < total; e0 += 256 * 8) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = e0 + i * 256;
                v[i] = (e < total && e / C < npx) ? xr[e] : 0.0f;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = e0 + i * 256;
                if (e < total) {
                    const int px = e / C, c = e - px * C;
                    tile[c * PN_LD + px] = v[i];
                }
            }
        }
    } else {
        for (int c = grp; c < C; c += 4) tile[c * PN_LD + pix] = pvalid ? xs[(size_t)c * HW + p0 + pix] : 0.0f;
    }
    __syncthreads();
    // the two per-pixel sums are sequential fmaf chains over the channels by contract (oracle/mrefsr_oracle.c); the divisions
    // between them are not: all 256 threads share them
    __shared__ float pn_den[PN_PIX];
    float ss = 0.0f;
    if (tid < PN_PIX) {
        // (unrolled: 16 LDS reads in flight per batch -- the chain itself stays sequential; one read per iteration left the
        // LDS latency exposed 256 times per chain)
#pragma unroll 16
        for (int c = 0; c < C; ++c) {
            const float v = tile[c * PN_LD + tid];
            ss = __builtin_fmaf(v, v, ss);
        }
        float d = __builtin_sqrtf(ss);
        if (!(d > 1e-12f)) d = 1e-12f;
        pn_den[tid] = d;
    }
    if (normalize) {
        __syncthreads();
        for (int e = tid; e < PN_PIX * C; e += 256) {
            const int c = e >> 6, px = e & 63;
            tile[c * PN_LD + px] = tile[c * PN_LD + px] / pn_den[px];
        }
        __syncthreads();
    }
    if (tid < PN_PIX) {
        float s2 = ss, dd = 0.0f;   // dd: squared norm of the fp16 rounding error of the pixel vector (pre-filter window)
        if (normalize) {
            s2 = 0.0f;
#pragma unroll 16
            for (int c = 0; c < C; ++c) {
                const float v = tile[c * PN_LD + tid];
                s2 = __builtin_fmaf(v, v, s2);
                if (d2) {
                    const float r = v - (float)(_Float16)v;
                    dd = __builtin_fmaf(r, r, dd);
                }
            }
        } else if (d2) {
            for (int c = 0; c < C; ++c) {
                const float v = tile[c * PN_LD + tid];
                const float r = v - (float)(_Float16)v;
                dd = __builtin_fmaf(r, r, dd);
            }
        }
        if (pvalid) n2[(size_t)n * HW + p0 + tid] = s2;
        if (d2 && pvalid) d2[(size_t)n * HW + p0 + tid] = dd;
    }
    __syncthreads();
    const int half = Cp >> 1;
    const bool cp2 = (Cp & (Cp - 1)) == 0;   // (uniform) Cp a power of two: shifts instead of the integer divisions below
    const int cps = 31 - __builtin_clz(Cp);
    for (int e = tid; e < PN_PIX * Cp; e += 256) {
        const int px = cp2 ? e >> cps : e / Cp, pos = e - px * Cp;
        const int kh = pos >= half, tt = pos - kh * half;
        const int c = 2 * tt + kh;
        if (p0 + px < HW) y[((size_t)n * HW + p0 + px) * Cp + pos] = (c < C) ? tile[c * PN_LD + px] : 0.0f;
    }
    if (ybf && ybf_fmt == 1) {
        // single fp16 plane for the fp16 pre-filter: yh = fp16(y), round-to-nearest-even, [pixel][Cp]
        const int hp = Cp >> 1;   // two channels per thread: 4-byte stores, 256 contiguous bytes per wave
        for (int e = tid; e < PN_PIX * hp; e += 256) {
            const int px = cp2 ? e >> (cps - 1) : e / hp, c = 2 * (e - px * hp);
            if (p0 + px >= HW) continue;
            const _Float16 h0 = (_Float16)((c < C) ? tile[c * PN_LD + px] : 0.0f), h1 = (_Float16)((c + 1 < C) ? tile[(c + 1) * PN_LD + px] : 0.0f);
            *reinterpret_cast<unsigned int *>(ybf + ((size_t)n * HW + p0 + px) * Cp + c) =
                (unsigned int)__builtin_bit_cast(unsigned short, h0) | ((unsigned int)__builtin_bit_cast(unsigned short, h1) << 16);
        }
    } else if (ybf) {
        // two-term bf16 split for the pre-filter's matrix pass: hi = bf16(v), lo = bf16(v - hi)
        // (round-to-nearest-even), natural channel order, [pixel][hi Cp | lo Cp]
        for (int e = tid; e < PN_PIX * Cp; e += 256) {
            const int px = e / Cp, c = e - px * Cp;
            if (p0 + px >= HW) continue;
            const float v = (c < C) ? tile[c * PN_LD + px] : 0.0f;
            unsigned int u = __float_as_uint(v);
            const unsigned int hi = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
            const float rem = v - __uint_as_float(hi << 16);
            u = __float_as_uint(rem);
            const unsigned int lo = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
            unsigned short *dst = ybf + ((size_t)n * HW + p0 + px) * 2 * Cp + c;
            dst[0] = (unsigned short)hi;
            dst[Cp] = (unsigned short)lo;
        }
    }
}

__global__ void patch_norm_kernel(const float *__restrict__ n2, float *__restrict__ nrm_eps,
                                  float *__restrict__ inv, int N, int h, int w)
{
    const int pw = w - 2, P = (h - 2) * pw;
    const long total = (long)N * P;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i / P), r = (int)(i - (long)n * P);
        const int ry = r / pw, rx = r - ry * pw;
        const float *m = n2 + (size_t)n * h * w + ry * w + rx;
        float s = m[0];
        s = s + m[1];
        s = s + m[2];
        s = s + m[w];
        s = s + m[w + 1];
        s = s + m[w + 2];
        s = s + m[2 * w];
        s = s + m[2 * w + 1];
        s = s + m[2 * w + 2];
        const float ne = __builtin_sqrtf(s) + 1e-5f;
        if (nrm_eps) nrm_eps[i] = ne;
        if (inv) inv[i] = 1.0f / ne;
    }
}

// ---------------------------------------------------------------------------------------------
// corr_top1: one block = one 6x14 tile of query patches (8x16 pixels) of one (input, ref) pair,
// streaming every 6x14 ref-patch tile (8x16 pixels) through LDS.
//   4 waves, one per SIMD.  Wave v keeps the A operand (its 32 query pixels x up to 256 channels)
//   in 128 registers for the whole block; per ref tile it accumulates the 32x128 pixel Gram
//   block in 4 MFMA 32x32 accumulators, K streamed in 64-channel chunks (double-buffered LDS).
//   The 128x128 Gram tile then goes to LDS, 252 threads each own (query, third of the ref rows)
//   and do the 9-tap diagonal box-sum, the inv-norm multiply and the running argmax.
// ---------------------------------------------------------------------------------------------
constexpr int T_PY = 8, T_PX = 16;   // pixel tile
constexpr int T_QY = 6, T_QX = 14;   // patches per tile
constexpr int T_NQ = T_QY * T_QX;    // 84
constexpr int BS_LD = 68;            // floats per pixel in a staged chunk: [kh][32] + 4 pad
constexpr int BS_BUF = 128 * BS_LD;
constexpr int GS_LD = 132;         // Gram tile row stride: multiple of 4 floats so box-sum segments are 16-B aligned
constexpr int CORR_LDS_FLOATS = 2 * BS_BUF + 128 * GS_LD + 8 * T_NQ;

__device__ __forceinline__ void stage_load(f32x4 (&r)[8], const float *__restrict__ yref, int Cp, int h, int w,
                                           int ry0, int rx0, int ch, int tid)
{
    const int pixel = tid & 127, kh = tid >> 7;
    const int py = ry0 + (pixel >> 4), px = rx0 + (pixel & 15);
    if (py < h && px < w) {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(yref + ((size_t)py * w + px) * Cp + kh * (Cp >> 1) + ch * 32);
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = src[j];
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

__device__ __forceinline__ void stage_store(const f32x4 (&r)[8], float *bs, int tid)
{
    const int pixel = tid & 127, kh = tid >> 7;
    f32x4 *dst = reinterpret_cast<f32x4 *>(bs + pixel * BS_LD + kh * 32);
#pragma unroll
    for (int j = 0; j < 8; ++j) dst[j] = r[j];
}

__global__ __launch_bounds__(256) void corr_top1_kernel(
    const float *__restrict__ y_in, const float *__restrict__ y_ref, const float *__restrict__ inv_ref,
    const float *__restrict__ nrm_in, int64_t *__restrict__ max_idx, float *__restrict__ max_val, int n_in,
    int Cp, int h, int w, int tiles_x, int tiles_y, const int *__restrict__ tile_flag, const int *__restrict__ flag_count,
    int min_flags)
{
    // fallback mode of the pre-filter path: only the query tiles it flagged, and only when more than
    // `min_flags` queries overflowed (fewer are brute-forced one by one by the re-scoring kernel)
    if (tile_flag && (*flag_count <= min_flags || !tile_flag[blockIdx.y * gridDim.x + blockIdx.x])) return;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Bs = smem;
    float *Gs = smem + 2 * BS_BUF;
    float *redv = Gs + 128 * GS_LD;
    int *redi = reinterpret_cast<int *>(redv + 3 * T_NQ);
    float *invs = redv + 6 * T_NQ;  // [2][84]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int pair = blockIdx.y;
    const int qy0 = (blockIdx.x / tiles_x) * T_QY, qx0 = (blockIdx.x % tiles_x) * T_QX;
    const int ph = h - 2, pw = w - 2, P = ph * pw;
    const int nch = Cp >> 6;
    const int in_i = pair % n_in;
    const float *yin = y_in + (size_t)in_i * h * w * Cp;
    const float *yref = y_ref + (size_t)pair * h * w * Cp;
    const float *inv = inv_ref + (size_t)pair * P;

    // ---- A operand: 32 query pixels of this wave, all channels, in registers ----
    f32x4 A[4][8];
    {
        const int pi = wv * 32 + (lane & 31), kh = lane >> 5;
        const int py = qy0 + (pi >> 4), px = qx0 + (pi & 15);
        const bool ok = py < h && px < w;
        const float *src = yin + ((size_t)(ok ? py : 0) * w + (ok ? px : 0)) * Cp + kh * (Cp >> 1);
#pragma unroll
        for (int ch = 0; ch < 4; ++ch)
#pragma unroll
            for (int t4 = 0; t4 < 8; ++t4)
                A[ch][t4] = (ok && ch < nch) ? *reinterpret_cast<const f32x4 *>(src + ch * 32 + t4 * 4)
                                             : f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // ---- box-sum role of this thread ----
    const int bq = tid % T_NQ, bpart = tid / T_NQ;  // bpart 3 = idle (tid >= 252)
    const int bqy = bq / T_QX, bqx = bq - bqy * T_QX;
    const bool bq_valid = bpart < 3 && (qy0 + bqy < ph) && (qx0 + bqx < pw);
    float best_v = -__builtin_inff();
    int best_i = 0x7fffffff;

    const int n_rt = tiles_x * tiles_y;
    const long total = (long)n_rt * nch;
    f32x4 stg[8];
    stage_load(stg, yref, Cp, h, w, 0, 0, 0, tid);
    stage_store(stg, Bs, tid);
    __syncthreads();

    long s = 0;
    for (int rt = 0; rt < n_rt; ++rt) {
        const int rty = rt / tiles_x;
        const int ry0 = rty * T_QY, rx0 = (rt - rty * tiles_x) * T_QX;
        f32x16 acc[4];
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[n][e] = 0.0f;
        if (tid < T_NQ) {  // 1/(||ref patch||+1e-5) of this tile's 6x14 patches -> LDS (parity-buffered)
            const int ryl = tid / T_QX, rxl = tid - ryl * T_QX;
            const int ry = ry0 + ryl, rx = rx0 + rxl;
            invs[(rt & 1) * T_NQ + tid] = (ry < ph && rx < pw) ? inv[(size_t)ry * pw + rx] : 0.0f;
        }

#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            if (ch < nch) {
                const int buf = (int)(s & 1);
                const bool has_next = s + 1 < total;
                if (has_next) {
                    int nrt = rt, nchk = ch + 1;
                    if (nchk == nch) { nchk = 0; nrt = rt + 1; }
                    const int nty = nrt / tiles_x;
                    stage_load(stg, yref, Cp, h, w, nty * T_QY, (nrt - nty * tiles_x) * T_QX, nchk, tid);
                }
                const float *bb = Bs + buf * BS_BUF + (lane & 31) * BS_LD + (lane >> 5) * 32;
#pragma unroll
                for (int t4 = 0; t4 < 8; ++t4) {
                    f32x4 b[4];
#pragma unroll
                    for (int n = 0; n < 4; ++n) b[n] = *reinterpret_cast<const f32x4 *>(bb + n * 32 * BS_LD + t4 * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int n = 0; n < 4; ++n)
                            acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[ch][t4][j], b[n][j], acc[n], 0, 0, 0);
                }
                if (has_next) stage_store(stg, Bs + (buf ^ 1) * BS_BUF, tid);
                __syncthreads();
                ++s;
            }
        }

        // ---- Gram tile -> LDS ----
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = wv * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                Gs[row * GS_LD + n * 32 + (lane & 31)] = acc[n][e];
            }
        __syncthreads();

        // ---- 9-tap diagonal box-sum, inv-norm, running argmax ----
        // For a fixed query and reference row, tap (dy,dx) needs Gram row (q + (dy,dx)) at columns
        // (ryl+dy)*16 + dx + rxl, rxl = 0..13: one aligned 16-float segment per tap, fetched with
        // four ds_read_b128; all 14 outputs of the row are then independent add chains (ILP instead
        // of a latency-bound scalar loop).  Sum order per output stays row-major over (dy,dx).
        if (bq_valid) {
            const float *g0 = Gs + (bqy * T_PX + bqx) * GS_LD + bpart * 2 * T_PX;
            const int nrx = (pw - rx0) < T_QX ? (pw - rx0) : T_QX;
            const int ry_a = ry0 + bpart * 2;      // this thread's two reference rows: ry_a, ry_a + 1
            float v[2][T_QX];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                f32x4 sg[2][3][4];
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const f32x4 *sp = reinterpret_cast<const f32x4 *>(g0 + (dy * T_PX + dx) * GS_LD + (r2 + dy) * T_PX);
#pragma unroll
                        for (int k4 = 0; k4 < 4; ++k4) sg[r2][dx][k4] = sp[k4];
                    }
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                        for (int rxl = 0; rxl < T_QX; ++rxl) {
                            const float g = sg[r2][dx][(rxl + dx) >> 2][(rxl + dx) & 3];
                            if (dy == 0 && dx == 0) v[r2][rxl] = g;
                            else v[r2][rxl] = v[r2][rxl] + g;
                        }
            }
            const float *ivs = invs + (rt & 1) * T_NQ + bpart * 2 * T_QX;
#pragma unroll
            for (int r2 = 0; r2 < 2; ++r2) {
                if (ry_a + r2 < ph) {
                    // four independent partial arg-maxes, merged at the end (total order => any order)
                    float pv[4];
                    int pi[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) { pv[k] = -__builtin_inff(); pi[k] = 0x7fffffff; }
                    const int rbase = (ry_a + r2) * pw + rx0;
#pragma unroll
                    for (int rxl = 0; rxl < T_QX; ++rxl) {
                        if (rxl < nrx) {
                            const float vv = v[r2][rxl] * ivs[r2 * T_QX + rxl];
                            const int r = rbase + rxl;
                            if (vv > pv[rxl & 3] || (vv == pv[rxl & 3] && r < pi[rxl & 3])) { pv[rxl & 3] = vv; pi[rxl & 3] = r; }
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (pv[k] > best_v || (pv[k] == best_v && pi[k] < best_i)) { best_v = pv[k]; best_i = pi[k]; }
                }
            }
        }
        // next tile's Gs writes are separated from these reads by the per-chunk barriers above
    }

    // ---- merge the three row-parts of each query ----
    if (bpart < 3) { redv[bpart * T_NQ + bq] = best_v; redi[bpart * T_NQ + bq] = best_i; }
    __syncthreads();
    if (tid < T_NQ && bq_valid) {
        float v = redv[tid];
        int i = redi[tid];
#pragma unroll
        for (int p = 1; p < 3; ++p) {
            const float v2 = redv[p * T_NQ + tid];
            const int i2 = redi[p * T_NQ + tid];
            if (v2 > v || (v2 == v && i2 < i)) { v = v2; i = i2; }
        }
        if (i == 0x7fffffff) i = 0;
        const size_t o = (size_t)pair * P + (size_t)(qy0 + bqy) * pw + qx0 + bqx;
        max_idx[o] = (int64_t)i;
        if (max_val) max_val[o] = v / nrm_in[(size_t)in_i * P + (size_t)(qy0 + bqy) * pw + qx0 + bqx];
    }
}

// ---------------------------------------------------------------------------------------------
// index -> 9 shifted offset planes at one scale.  Pure HBM write stream (8 B per thread, coalesced).
// ---------------------------------------------------------------------------------------------
__global__ void offsets_kernel(const int64_t *__restrict__ idx, float2 *__restrict__ out, int N, int h, int w, int sh)
{
    // grid: x strides over the H x W pixels of one plane, y = n * 9 + tap; s = 2^sh (1, 2, 4): no division by a run-time value but the two
    // of the row decode, in 32 bits (the one-dimensional form spent eight 64-bit divisions per 8-byte store: VALU-bound at 2.3 TB/s)
    const int s = 1 << sh, ph = h - 2, pw = w - 2, H = h << sh, W = w << sh;
    const int k = blockIdx.y % 9, n = blockIdx.y / 9, ky = (k * 11) >> 5, kx = k - 3 * ky;
    const int64_t *const ip = idx + (size_t)n * ph * pw;
    float2 *const op = out + (size_t)blockIdx.y * H * W;
    const float fs = (float)s;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < H * W; i += gridDim.x * blockDim.x) {
        const int Y = i / W, X = i - Y * W;
        const int ys = Y - (ky << sh), xs = X - (kx << sh);
        float2 o = make_float2(0.f, 0.f);
        if (ys >= 0 && xs >= 0) {
            const int y = ys >> sh, x = xs >> sh;
            if (y < ph && x < pw) {
                const int m = (int)ip[y * pw + x];   // (an index into a (h - 2) x (w - 2) map)
                const int my = m / pw, mx = m - my * pw;
                o.x = (float)(mx - x) * fs;
                o.y = (float)(my - y) * fs;
            }
        }
        op[i] = o;
    }
}

}  // namespace

// ============================================ C ABI ============================================
MREFSR_EXPORT int mrefsr_corr_padded_channels(int C)
{
    if (C <= 0 || C > 256) return mrefsr::fail(MREFSR_E_UNSUPPORTED, "corr: C=%d outside (0, 256]", C);
    return ((C + 63) / 64) * 64;
}

MREFSR_EXPORT int mrefsr_pixnorm_f32(const float *x, float *y, float *n2, void *ybf, int N, int C, int HW, int normalize,
                                     int x_nhwc, int ybf_fmt, float *d2, mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(x && y && n2, "pixnorm: null pointer");
    MREFSR_REQUIRE(N > 0 && HW > 0, "pixnorm: N=%d HW=%d", N, HW);
    const int Cp = mrefsr_corr_padded_channels(C);
    if (Cp < 0) return Cp;
    const size_t lds = (size_t)C * PN_LD * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(pixnorm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid(mrefsr::cdiv(HW, PN_PIX), N);
    hipLaunchKernelGGL(pixnorm_kernel, grid, dim3(256), lds, (hipStream_t)stream, x, y, n2, (unsigned short *)ybf, C, Cp, HW,
                       normalize, x_nhwc, ybf_fmt, d2);
    return mrefsr::check_launch("pixnorm");
}

MREFSR_EXPORT int mrefsr_patch_norm_f32(const float *n2, float *nrm_eps, float *inv, int N, int h, int w,
                                        mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(n2 && (nrm_eps || inv), "patch_norm: null pointer");
    MREFSR_REQUIRE(N > 0 && h >= 3 && w >= 3, "patch_norm: N=%d h=%d w=%d (need h,w >= 3)", N, h, w);
    const long total = (long)N * (h - 2) * (w - 2);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(patch_norm_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, n2, nrm_eps, inv, N, h, w);
    return mrefsr::check_launch("patch_norm");
}

MREFSR_EXPORT int mrefsr_corr_top1_f32(const float *y_in, const float *y_ref, const float *inv_ref,
                                       const float *nrm_in, int64_t *max_idx, float *max_val, int n_in, int n_pair,
                                       int Cp, int h, int w, mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(y_in && y_ref && inv_ref && max_idx, "corr_top1: null pointer");
    MREFSR_REQUIRE(!max_val || nrm_in, "corr_top1: max_val requested without nrm_in");
    MREFSR_REQUIRE(n_in > 0 && n_pair > 0, "corr_top1: n_in=%d n_pair=%d", n_in, n_pair);
    MREFSR_REQUIRE(h >= 3 && w >= 3, "corr_top1: h=%d w=%d (3x3 patches need h,w >= 3)", h, w);
    if (Cp <= 0 || Cp > 256 || (Cp & 63))
        return mrefsr::fail(MREFSR_E_UNSUPPORTED, "corr_top1: Cp=%d must be 64, 128, 192 or 256", Cp);
    if ((long)(h - 2) * (w - 2) >= 0x7fffffffL)
        return mrefsr::fail(MREFSR_E_UNSUPPORTED, "corr_top1: too many patches");
    const int tiles_y = mrefsr::cdiv(h - 2, T_QY), tiles_x = mrefsr::cdiv(w - 2, T_QX);
    const size_t lds = (size_t)CORR_LDS_FLOATS * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(corr_top1_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid(tiles_x * tiles_y, n_pair);
    hipLaunchKernelGGL(corr_top1_kernel, grid, dim3(256), lds, (hipStream_t)stream, y_in, y_ref, inv_ref, nrm_in,
                       max_idx, max_val, n_in, Cp, h, w, tiles_x, tiles_y, (const int *)nullptr, (const int *)nullptr, 0);
    return mrefsr::check_launch("corr_top1");
}

// the exact kernel restricted to flagged query tiles (called by mrefsr_corr_top1_prefilter_f32)
int mrefsr::launch_corr_top1_flagged(const float *y_in, const float *y_ref, const float *inv_ref, const float *nrm_in,
                                     int64_t *max_idx, float *max_val, int n_in, int n_pair, int Cp, int h, int w,
                                     const int *tile_flag, const int *flag_count, int min_flags, hipStream_t stream)
{
    const int tiles_y = mrefsr::cdiv(h - 2, T_QY), tiles_x = mrefsr::cdiv(w - 2, T_QX);
    const size_t lds = (size_t)CORR_LDS_FLOATS * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(corr_top1_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(corr_top1_kernel, dim3(tiles_x * tiles_y, n_pair), dim3(256), lds, stream, y_in, y_ref, inv_ref, nrm_in,
                       max_idx, max_val, n_in, Cp, h, w, tiles_x, tiles_y, tile_flag, flag_count, min_flags);
    return mrefsr::check_launch("corr_top1(flagged tiles)");
}

MREFSR_EXPORT int mrefsr_offsets_from_idx_f32(const int64_t *max_idx, float *off_s1, float *off_s2, float *off_s4,
                                              int N, int h, int w, mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(max_idx, "offsets_from_idx: null index pointer");
    MREFSR_REQUIRE(N > 0 && h >= 3 && w >= 3, "offsets_from_idx: N=%d h=%d w=%d", N, h, w);
    float *outs[3] = {off_s1, off_s2, off_s4};
    for (int si = 0; si < 3; ++si) {
        if (!outs[si]) continue;
        const long plane = (long)(h << si) * (w << si);
        MREFSR_REQUIRE(plane < (1L << 31) && (long)N * 9 <= 65535, "offsets_from_idx: N=%d h=%d w=%d beyond the grid", N, h, w);
        const long blocks = (plane + 255) / 256;
        hipLaunchKernelGGL(offsets_kernel, dim3((int)(blocks < 2048 ? blocks : 2048), N * 9), dim3(256), 0, (hipStream_t)stream, max_idx,
                           reinterpret_cast<float2 *>(outs[si]), N, h, w, si);
    }
    return mrefsr::check_launch("offsets_from_idx");
}
