// Correlation + top-1, fast path: bf16x3-split MFMA pre-filter + exact fp32 re-scoring.
//
// Same contract as corr_top1_kernel (csrc/corr.hip): the returned index is the arg-max of the
// CANONICAL fp32 correlation (fmaf chains over channels ascending, 8 sequential tap adds,
// * inv[r]; lowest index on ties) -- bit-identical to oracle/mrefsr_oracle.c.  What changes is how
// it is found:
//
//   pass A  (corr_prefilter_kernel)  the pixel Gram is computed APPROXIMATELY on the bf16 matrix
//           pipe (16x the fp32 MFMA rate) from a two-term split y = hi + lo (hi = bf16(y),
//           lo = bf16(y - hi)):  G~ = hi.hi + hi.lo + lo.hi, fp32 accumulate.  Per pixel pair
//           |G~ - G| <= KAPPA * |a||b|  (split truncation 3*2^-18 + accumulation),
//           so after the 9-tap box-sum and the inv-norm multiply
//               |v~(q,r) - v(q,r)| <= KAPPA * nrm_in[q]            (Cauchy-Schwarz over the taps).
//           Every r with v~ >= max v~ - TAU(q), TAU = 2*KAPPA*nrm_in[q] (+slack), is kept as a
//           candidate: the canonical arg-max and all its exact ties are provably among them.
//   pass B  (corr_rescore_kernel)    canonical fp32 evaluation of the (typically 1-2) candidates of
//           each query, arg-max with the tie rule.
//   pass B' (tail of the same kernel) queries whose candidate set overflowed (e.g. many exact ties)
//           are evaluated canonically against every reference patch.  Worst case (everything
//           overflows) costs about as much as the exact kernel; it is never wrong.
#include "common.h"
#include "corr_cfg.h"
#include <stdlib.h>

namespace {

using namespace mrefsr_corr;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int T_NQ = T_QY * T_QX;   // 84
constexpr int PB_LD = 68;           // dwords per pixel per staged chunk: 64 ch hi | 64 ch lo | 4 pad
constexpr int PB_BUF = 128 * PB_LD;
constexpr int GS_LD = 132;
constexpr int CAP = 4;              // candidates kept per (query, third of the reference rows)
// (BRUTE_MAX overflowed queries are brute-forced one by one: 235 MB of reference reads each, in parallel: ~2 ms; more ->
// the exact kernel on their tiles, >= 6.5 ms: one block per flagged tile)
constexpr float KAPPA = 1.220703125e-4f;  // 2^-13: bound on |G~ - G| / (|a||b|), ~3x the analytic estimate
constexpr float TAU_SCALE = 2.0f * 1.01f * KAPPA;
constexpr int PRE_LDS_DWORDS = 2 * PB_BUF + 128 * GS_LD + 2 * T_NQ + 3 * T_NQ * (2 * CAP + 3);
constexpr int DB_BUF = 128 * 64;    // wave-specialised kernel: unpadded, XOR-swizzled chunk buffer (LDS-DMA target), dwords
constexpr int PIPE_LDS_DWORDS = 2 * DB_BUF + 128 * GS_LD + 2 * T_NQ + 3 * T_NQ * (2 * CAP + 3);

__device__ __forceinline__ bf16x8 as_bf(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

__device__ __forceinline__ void pre_stage_load(u32x4 (&r)[8], const unsigned short *__restrict__ ybf, int Cp, int h, int w,
                                               int ry0, int rx0, int ch, int tid)
{
    const int pixel = tid & 127, half = tid >> 7;  // half 0 = hi block, 1 = lo block
    const int py = ry0 + (pixel >> 4), px = rx0 + (pixel & 15);
    if (py < h && px < w) {
        const u32x4 *src = reinterpret_cast<const u32x4 *>(ybf + (((size_t)py * w + px) * 2 + half) * Cp + ch * 64);
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = src[j];
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = u32x4{0u, 0u, 0u, 0u};
    }
}

__device__ __forceinline__ void pre_stage_store(const u32x4 (&r)[8], unsigned int *bs, int tid)
{
    const int pixel = tid & 127, half = tid >> 7;
    u32x4 *dst = reinterpret_cast<u32x4 *>(bs + pixel * PB_LD + half * 32);
#pragma unroll
    for (int j = 0; j < 8; ++j) dst[j] = r[j];
}

__global__ __launch_bounds__(256) void corr_prefilter_kernel(
    const unsigned short *__restrict__ ybf_in, const unsigned short *__restrict__ ybf_ref,
    const float *__restrict__ inv_ref, const float *__restrict__ nrm_in, int *__restrict__ cand_r_out,
    int *__restrict__ cand_n_out, int *__restrict__ flag_count, int *__restrict__ flag_list, int *__restrict__ tile_flag, int n_in, int Cp, int h,
    int w, int tiles_x, int tiles_y)
{
    extern __shared__ __attribute__((aligned(16))) unsigned int smem_u[];
    unsigned int *Bs = smem_u;
    float *Gs = reinterpret_cast<float *>(smem_u + 2 * PB_BUF);
    float *invs = Gs + 128 * GS_LD;                               // [2][84]
    float *cv = invs + 2 * T_NQ;                                  // [3*84][CAP]  candidate values
    int *cr = reinterpret_cast<int *>(cv + 3 * T_NQ * CAP);       // [3*84][CAP]  candidate indices
    float *pmax = reinterpret_cast<float *>(cr + 3 * T_NQ * CAP);  // [3*84]
    int *pcnt = reinterpret_cast<int *>(pmax + 3 * T_NQ);         // [3*84]
    float *povf = reinterpret_cast<float *>(pcnt + 3 * T_NQ);     // [3*84]  running max at the last list overflow

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int pair = blockIdx.y;
    const int qy0 = (blockIdx.x / tiles_x) * T_QY, qx0 = (blockIdx.x % tiles_x) * T_QX;
    const int ph = h - 2, pw = w - 2, P = ph * pw;
    const int nch = Cp >> 6;
    const int in_i = pair % n_in;
    const unsigned short *yin = ybf_in + (size_t)in_i * h * w * 2 * Cp;
    const unsigned short *yref = ybf_ref + (size_t)pair * h * w * 2 * Cp;
    const float *inv = inv_ref + (size_t)pair * P;

    // ---- A operand (hi and lo halves of this wave's 32 query pixels) in registers ----
    u32x4 Ah[4][4], Al[4][4];  // [chunk][step within chunk]
    {
        const int pi = wv * 32 + (lane & 31), kb = lane >> 5;
        const int py = qy0 + (pi >> 4), px = qx0 + (pi & 15);
        const bool ok = py < h && px < w;
        const unsigned short *src = yin + ((size_t)(ok ? py : 0) * w + (ok ? px : 0)) * 2 * Cp + kb * 8;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch)
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const bool on = ok && ch < nch;
                Ah[ch][s4] = on ? *reinterpret_cast<const u32x4 *>(src + ch * 64 + s4 * 16) : u32x4{0u, 0u, 0u, 0u};
                Al[ch][s4] = on ? *reinterpret_cast<const u32x4 *>(src + Cp + ch * 64 + s4 * 16) : u32x4{0u, 0u, 0u, 0u};
            }
    }

    // ---- box-sum / candidate role of this thread ----
    const int bq = tid % T_NQ, bpart = tid / T_NQ;
    const int bqy = bq / T_QX, bqx = bq - bqy * T_QX;
    const bool bq_valid = bpart < 3 && (qy0 + bqy < ph) && (qx0 + bqx < pw);
    const int slot = bpart * T_NQ + bq;   // only meaningful for bpart < 3
    float run_max = -__builtin_inff(), thr = -__builtin_inff();
    float ovf_max = -__builtin_inff();  // running max at the time the list last overflowed (entries <= it were dropped)
    int cnt = 0;
    const float tau = bq_valid ? TAU_SCALE * nrm_in[(size_t)in_i * P + (size_t)(qy0 + bqy) * pw + qx0 + bqx] : 0.f;

    const int n_rt = tiles_x * tiles_y;
    const long total = (long)n_rt * nch;
    u32x4 stg[8];
    pre_stage_load(stg, yref, Cp, h, w, 0, 0, 0, tid);
    pre_stage_store(stg, Bs, tid);
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
        if (tid < T_NQ) {
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
                    pre_stage_load(stg, yref, Cp, h, w, nty * T_QY, (nrt - nty * tiles_x) * T_QX, nchk, tid);
                }
                const unsigned int *bb = Bs + buf * PB_BUF + (lane & 31) * PB_LD + (lane >> 5) * 4;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    u32x4 bh[4], bl[4];
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        bh[n] = *reinterpret_cast<const u32x4 *>(bb + n * 32 * PB_LD + s4 * 8);
                        bl[n] = *reinterpret_cast<const u32x4 *>(bb + n * 32 * PB_LD + s4 * 8 + 32);
                    }
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(Ah[ch][s4]), as_bf(bh[n]), acc[n], 0, 0, 0);
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(Ah[ch][s4]), as_bf(bl[n]), acc[n], 0, 0, 0);
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(Al[ch][s4]), as_bf(bh[n]), acc[n], 0, 0, 0);
                }
                if (has_next) pre_stage_store(stg, Bs + (buf ^ 1) * PB_BUF, tid);
                __syncthreads();
                ++s;
            }
        }

        // ---- approximate Gram tile -> LDS ----
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = wv * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                Gs[row * GS_LD + n * 32 + (lane & 31)] = acc[n][e];
            }
        __syncthreads();

        // ---- box-sum (same shape as the exact kernel) + candidate collection ----
        if (bq_valid) {
            const float *g0 = Gs + (bqy * T_PX + bqx) * GS_LD + bpart * 2 * T_PX;
            const int nrx = (pw - rx0) < T_QX ? (pw - rx0) : T_QX;
            const int ry_a = ry0 + bpart * 2;
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
            // tile-local maximum first (cheap, branch-free); the rare candidate path runs only when
            // something in this tile reaches the running threshold
            float tmax = -__builtin_inff();
#pragma unroll
            for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
                for (int rxl = 0; rxl < T_QX; ++rxl) {
                    const bool ok = (ry_a + r2 < ph) && (rxl < nrx);
                    v[r2][rxl] = ok ? v[r2][rxl] * ivs[r2 * T_QX + rxl] : -__builtin_inff();
                    tmax = fmaxf(tmax, v[r2][rxl]);
                }
            if (tmax >= thr) {
                if (tmax > run_max) { run_max = tmax; thr = run_max - tau; }
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
                    for (int rxl = 0; rxl < T_QX; ++rxl) {
                        const float vv = v[r2][rxl];
                        if (vv >= thr) {
                            if (cnt == CAP) {  // prune against the current threshold, then retry
                                int m = 0;
                                for (int k = 0; k < CAP; ++k) {
                                    const float cvk = cv[slot * CAP + k];
                                    const int crk = cr[slot * CAP + k];
                                    if (cvk >= thr) { cv[slot * CAP + m] = cvk; cr[slot * CAP + m] = crk; ++m; }
                                }
                                cnt = m;
                            }
                            if (cnt == CAP) {
                                // overflow: drop the list but remember how high the dropped entries could
                                // be; if that level still matters at the end the query is brute-forced
                                ovf_max = run_max;
                                cnt = 0;
                            }
                            cv[slot * CAP + cnt] = vv;
                            cr[slot * CAP + cnt] = (ry_a + r2) * pw + rx0 + rxl;
                            ++cnt;
                        }
                    }
            }
        }
    }

    // ---- merge the three row-parts of each query, publish candidates ----
    if (bpart < 3) { pmax[slot] = run_max; pcnt[slot] = cnt; povf[slot] = ovf_max; }
    __syncthreads();
    if (tid < T_NQ && bq_valid) {
        const float gmax = fmaxf(fmaxf(pmax[tid], pmax[T_NQ + tid]), pmax[2 * T_NQ + tid]);
        const float gthr = gmax - tau;
        const size_t qo = (size_t)pair * P + (size_t)(qy0 + bqy) * pw + qx0 + bqx;
        int n = 0;
        bool over = false;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int c = pcnt[p * T_NQ + tid];
            // entries dropped at an overflow were all <= povf: they matter only if povf reaches the final window
            if (povf[p * T_NQ + tid] >= gthr) over = true;
            for (int k = 0; k < c; ++k) {
                if (cv[(p * T_NQ + tid) * CAP + k] >= gthr) {
                    if (n < SLOTS) cand_r_out[qo * SLOTS + n] = cr[(p * T_NQ + tid) * CAP + k];
                    ++n;
                }
            }
        }
        if (over || n > SLOTS) {
            cand_n_out[qo] = -1;
            flag_list[atomicAdd(flag_count, 1)] = (int)qo;
            tile_flag[blockIdx.y * gridDim.x + blockIdx.x] = 1;
        } else {
            cand_n_out[qo] = n;
        }
    }
}

// fp16 single-plane operand: window constants (the kernels: corr_rowstream.hip; A/B tile variant: tools/ab/corr_prefilter_ab.inc)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
[[maybe_unused]] constexpr int DB16_BUF = 128 * 32;  // dwords: 128 pixels x 64 fp16 channels, unpadded, XOR-swizzled
#ifndef MREFSR_KAPPA16
#define MREFSR_KAPPA16 1.1e-3f
#endif
constexpr float KAPPA16 = MREFSR_KAPPA16;
constexpr float TAU_SCALE16 = 2.0f * 1.01f * KAPPA16;
constexpr int CAP16 = 8;             // wider window: more candidates per (query, third) survive until the merge
constexpr int PIPE16_LDS_DWORDS = 2 * DB16_BUF + 128 * GS_LD + 2 * T_NQ + 3 * T_NQ * (2 * CAP16 + 3);

#ifdef MREFSR_AB_KERNELS
#include "../../tools/ab/corr_prefilter_ab.inc"   // (retired generations: not product source, -DMREFSR_AB_KERNELS builds only)
#endif

// canonical correlation of query patch (qy,qx) with reference patch (ry,rx): bit-identical to
// oracle/mrefsr_oracle.c:orc_corr_top1 (and to corr_top1_kernel).  y maps are in the split layout.
// value of lane j of the caller's quad (DPP quad_perm [j, j, j, j]: a register move, no LDS)
__device__ __forceinline__ float quad_bcast(const float x, const int j)
{
    const int xi = __builtin_bit_cast(int, x);
    int r;
    switch (j) {   // (every lane is written: row / bank masks 0xf, all lanes of a quad active -- no "old" value to keep, no copy in front)
    case 0: r = __builtin_amdgcn_mov_dpp(xi, 0x00, 0xf, 0xf, true); break;
    case 1: r = __builtin_amdgcn_mov_dpp(xi, 0x55, 0xf, 0xf, true); break;
    case 2: r = __builtin_amdgcn_mov_dpp(xi, 0xaa, 0xf, 0xf, true); break;
    default: r = __builtin_amdgcn_mov_dpp(xi, 0xff, 0xf, 0xf, true); break;
    }
    return __builtin_bit_cast(float, r);
}

// fma(lane j of the quad's x, y, g) as ONE instruction: the broadcast rides on the multiply-add as its DPP operand (the compiler
// emits a v_mov_b32_dpp per operand and a v_fma: three instructions where two do).  v_fmac_f32 is the fused multiply-add.
__device__ __forceinline__ float quad_fma(const float x, const float y, float g, const int j)
{
    switch (j) {
    case 0: asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf" : "+v"(g) : "v"(x), "v"(y)); break;
    case 1: asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "+v"(g) : "v"(x), "v"(y)); break;
    case 2: asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf" : "+v"(g) : "v"(x), "v"(y)); break;
    default: asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf" : "+v"(g) : "v"(x), "v"(y)); break;
    }
    return g;
}

__device__ __forceinline__ float canon_corr(const float *__restrict__ yin, const float *__restrict__ yref, int Cp, int w,
                                            int qy, int qx, int ry, int rx, float inv_r)
{
    const int half = Cp >> 1;
    float g[9];
    const float *a[9], *b[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        g[t] = 0.0f;
        a[t] = yin + ((size_t)(qy + t / 3) * w + qx + t % 3) * Cp;
        b[t] = yref + ((size_t)(ry + t / 3) * w + rx + t % 3) * Cp;
    }
    for (int tt = 0; tt < half; tt += 4) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const f32x4 ae = *reinterpret_cast<const f32x4 *>(a[t] + tt), ao = *reinterpret_cast<const f32x4 *>(a[t] + half + tt);
            const f32x4 be = *reinterpret_cast<const f32x4 *>(b[t] + tt), bo = *reinterpret_cast<const f32x4 *>(b[t] + half + tt);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                g[t] = __builtin_fmaf(ae[k], be[k], g[t]);  // channel 2(tt+k)
                g[t] = __builtin_fmaf(ao[k], bo[k], g[t]);  // channel 2(tt+k)+1
            }
        }
    }
    float v = g[0];
#pragma unroll
    for (int t = 1; t < 9; ++t) v = v + g[t];
    return v * inv_r;
}

// pass B' of the re-scoring kernels: queries whose candidate set overflowed: canonical evaluation against every
// reference patch (usually none: the loop bound is read from memory)
__device__ __forceinline__ void rescore_brute(const float *__restrict__ y_in, const float *__restrict__ y_ref, const float *__restrict__ inv_ref,
                                              const float *__restrict__ nrm_in, const int *__restrict__ flag_count,
                                              const int *__restrict__ flag_list, int64_t *__restrict__ max_idx, float *__restrict__ max_val,
                                              int n_in, int Cp, int h, int w, unsigned long long *__restrict__ brute, float *rv, int *ri)
{
    const int pw = w - 2, P = (h - 2) * pw;
    const int nflag = *flag_count <= BRUTE_MAX ? *flag_count : 0;   // more: the exact kernel re-does the flagged tiles
    // each flagged query is split into BRUTE_SEG reference segments handled by different blocks (one block alone
    // would stream the 235 MB reference map for ~5 ms); partial results merge through a 64-bit atomic max on
    // (order-preserving value bits, ~index) = the canonical total order, the last segment to finish writes the result
    for (int wk = blockIdx.x; wk < nflag * BRUTE_SEG; wk += gridDim.x) {
        const int f = wk / BRUTE_SEG, seg = wk - f * BRUTE_SEG;
        const long e = flag_list[f];
        const int pair = (int)(e / P), q = (int)(e - (long)pair * P);
        const int in_i = pair % n_in;
        const float *yin = y_in + (size_t)in_i * h * w * Cp;
        const float *yref = y_ref + (size_t)pair * h * w * Cp;
        const float *inv = inv_ref + (size_t)pair * P;
        const int qy = q / pw, qx = q - qy * pw;
        const int r_lo = (int)((long)P * seg / BRUTE_SEG), r_hi = (int)((long)P * (seg + 1) / BRUTE_SEG);
        float bv = -__builtin_inff();
        int bi = 0x7fffffff;
        for (int r = r_lo + threadIdx.x; r < r_hi; r += 256) {
            const float v = canon_corr(yin, yref, Cp, w, qy, qx, r / pw, r % pw, inv[r]);
            if (v > bv || (v == bv && r < bi)) { bv = v; bi = r; }
        }
        rv[threadIdx.x] = bv;
        ri[threadIdx.x] = bi;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (threadIdx.x < o) {
                const float v2 = rv[threadIdx.x + o];
                const int i2 = ri[threadIdx.x + o];
                if (v2 > rv[threadIdx.x] || (v2 == rv[threadIdx.x] && i2 < ri[threadIdx.x])) { rv[threadIdx.x] = v2; ri[threadIdx.x] = i2; }
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            unsigned long long *best = brute + 2 * f;
            int *done = reinterpret_cast<int *>(brute + 2 * f + 1);
            if (ri[0] != 0x7fffffff) {
                const unsigned int u = __float_as_uint(rv[0]);
                const unsigned int ord = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
                atomicMax(best, ((unsigned long long)ord << 32) | (0xffffffffu - (unsigned int)ri[0]));
            }
            __threadfence();
            if (atomicAdd(done, 1) == BRUTE_SEG - 1) {
                __threadfence();
                const unsigned long long key = atomicMax(best, 0ull);   // atomic read
                int i = 0;
                float v = -__builtin_inff();
                if (key) {
                    const unsigned int ord = (unsigned int)(key >> 32);
                    v = __uint_as_float((ord & 0x80000000u) ? (ord & 0x7fffffffu) : ~ord);
                    i = (int)(0xffffffffu - (unsigned int)(key & 0xffffffffu));
                }
                max_idx[e] = (int64_t)i;
                if (max_val) max_val[e] = v / nrm_in[(size_t)in_i * P + q];
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void corr_rescore_kernel(const float *__restrict__ y_in, const float *__restrict__ y_ref,
                                                           const float *__restrict__ inv_ref, const float *__restrict__ nrm_in,
                                                           const int *__restrict__ cand_r, const int *__restrict__ cand_n,
                                                           const int *__restrict__ flag_count, const int *__restrict__ flag_list,
                                                           int64_t *__restrict__ max_idx, float *__restrict__ max_val, int n_in,
                                                           int n_pair, int Cp, int h, int w, unsigned long long *__restrict__ brute)
{
    __shared__ float rv[256];
    __shared__ int ri[256];
    const int pw = w - 2, P = (h - 2) * pw;
    const long total = (long)n_pair * P;
    // One wave per query, FOUR lanes per tap: quad t < 9 owns tap t of the 3x3 patch.  The canonical arithmetic is one 256-channel fmaf
    // chain per tap (channels ascending, then the nine sums added in tap order: canon_corr above), and every lane of the quad runs that
    // whole chain -- redundantly, the same bits four times -- but the quad LOADS cooperatively: lane j fetches the j-th 16 bytes of each
    // 64-byte run of the two pixel vectors and the pieces are broadcast inside the quad (DPP quad_perm, no LDS).  Why: the CU's address
    // path takes one 64-byte request per clock; with one lane per tap (round 1-4: 16 lanes per query) the four lanes of a hardware quad
    // read four DIFFERENT pixel vectors, i.e. four requests per 64 bytes -- the kernel was bound by that (4.7 ms per call for 28 GB of
    // mostly L2-resident reads), not by its fmaf chains.  Same bits as before: the chain order is untouched.
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, quad = lane >> 2, jq = lane & 3;
    const bool live = quad < 9;
    const int tap = live ? quad : 8, ty = tap / 3, tx = tap - 3 * ty;
    const int half = Cp >> 1;
    for (long e = blockIdx.x * 4L + wv; e < total; e += (long)gridDim.x * 4L) {
        const int n = cand_n[e];
        if (n < 0) continue;  // brute-force pass owns this query
        if (n == 1 && !max_val) {   // the window kept ONE candidate: it is the arg-max (and its only exact tie) -- nothing to evaluate when
                                    // the caller does not ask for the value (the path: ref_map_util.py:78-84's max_val is unused there)
            if (lane == 0) max_idx[e] = (int64_t)cand_r[e * SLOTS];
            continue;
        }
        const int pair = (int)(e / P), q = (int)(e - (long)pair * P);
        const int in_i = pair % n_in;
        const float *yin = y_in + (size_t)in_i * h * w * Cp;
        const float *yref = y_ref + (size_t)pair * h * w * Cp;
        const float *inv = inv_ref + (size_t)pair * P;
        const int qy = q / pw, qx = q - qy * pw;
        const float *a = yin + ((size_t)(qy + ty) * w + qx + tx) * Cp + 4 * jq;
        float bv = -__builtin_inff();
        int bi = 0x7fffffff;
        for (int k = 0; k < n; ++k) {
            const int r = cand_r[e * SLOTS + k];
            const int ry = r / pw, rx = r - ry * pw;
            const float *b = yref + ((size_t)(ry + ty) * w + rx + tx) * Cp + 4 * jq;
            float g = 0.0f;
            for (int tt = 0; tt < half; tt += 16) {   // 16 even + 16 odd channels per round: 4 x 64 contiguous bytes per quad
                f32x4 ae = {0.f, 0.f, 0.f, 0.f}, ao = ae, be = ae, bo = ae;
                if (live) {
                    ae = *reinterpret_cast<const f32x4 *>(a + tt), ao = *reinterpret_cast<const f32x4 *>(a + half + tt);
                    be = *reinterpret_cast<const f32x4 *>(b + tt), bo = *reinterpret_cast<const f32x4 *>(b + half + tt);
                }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {   // channel 2 (tt + 4 jj + c), then 2 (tt + 4 jj + c) + 1
                        g = quad_fma(ae[c], quad_bcast(be[c], jj), g, jj);
                        g = quad_fma(ao[c], quad_bcast(bo[c], jj), g, jj);
                    }
                }
            }
            const int gi = __builtin_bit_cast(int, g);   // (readlane moves bits: an int builtin)
            float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(gi, 0));
#pragma unroll
            for (int t = 1; t < 9; ++t) v = v + __builtin_bit_cast(float, __builtin_amdgcn_readlane(gi, 4 * t));
            v = v * inv[r];
            if (v > bv || (v == bv && r < bi)) { bv = v; bi = r; }
        }
        if (lane == 0) {
            if (bi == 0x7fffffff) bi = 0;
            max_idx[e] = (int64_t)bi;
            if (max_val) max_val[e] = bv / nrm_in[(size_t)in_i * P + q];
        }
    }
    rescore_brute(y_in, y_ref, inv_ref, nrm_in, flag_count, flag_list, max_idx, max_val, n_in, Cp, h, w, brute, rv, ri);
}

// ---- pass B, second form (round 6): SEVEN (query, candidate) evaluations per wave, one LANE per (evaluation, tap).
// corr_rescore_kernel above spends a whole wave on one evaluation -- 36 live lanes, every chain run four times over: 550 wave instructions
// per evaluation, 738 M per call on the benchmark's maps = the kernel's duration (VALU issue 70-80 % busy, profiles/r6_bench_pmc_per_step.json).
// Here lane = 9 u + t runs the 256-channel chain of tap t of evaluation u ONCE (lane 63 idles), and the operands reach it through a
// wave-private LDS tile: per stage (16 even + 16 odd channels = 2 x 64 bytes of each of the 63 query-side and 63 reference-side pixel
// vectors) eight lanes fetch a vector's 128 bytes (16 fully coalesced b128 requests per lane and stage, the next stage's in flight
// during the chains), store them, and every lane reads its own two vectors back with 16-byte reads: ~110 wave instructions per
// evaluation, 1.7 -> 0.9 ms per call in the benchmark step (bound by moving the operands now: 19 GB through the address path, 22 GB of
// LDS writes).  (Stages of 8 + 8 channels -- half the tile, sixteen waves per CU instead of eight, -DMREFSR_RS_CH=8 -- measured slower:
// -0.3 instead of -1.25 ms per call against the quad kernel.)  The arithmetic is canon_corr's: per tap one fmaf chain over the channels ascending (even plane, then odd, per channel
// pair), the nine sums added in tap order, times inv_ref; candidates of a query merge under (value, then smaller index) -- the same bits.
// A wave takes 32 consecutive queries at a time: their candidate counts become a unit list in LDS (queries with one candidate and no
// value asked for are answered on the spot), the list is walked seven units at a time.
constexpr int RS_Q = 32;                       // queries per group
constexpr int RS_U = 7;                        // evaluations per batch
#ifndef MREFSR_RS_CH
#define MREFSR_RS_CH 16
#endif
constexpr int RS_CH = MREFSR_RS_CH;            // channels of each parity per stage: a staged vector is 2 x RS_CH floats (16: 128 bytes; 8: A/B builds)
constexpr int RS_NP = 2 * RS_CH / 4;           // its 16-byte pieces = loader lanes per vector
constexpr bool RS_ROT = RS_CH == 8;            // 64-byte vectors: unpadded, the pieces of vector v rotated by v / 4; 128-byte vectors: 16 bytes of padding
constexpr int RS_VLD = RS_NP * 16 + (RS_ROT ? 0 : 16);   // bytes per staged vector
constexpr int RS_NLD = 128 * RS_NP / 64;       // requests per lane and stage
constexpr int RS_VPI = 64 / RS_NP;             // vectors per request instruction (a multiple of 4: the rotation repeats)
constexpr int RS_TILE = 128 * RS_VLD;          // 64 query-side + 64 reference-side vectors (63 used of each)
constexpr int RS_TAB = RS_Q * SLOTS * 2;       // unit list: (query of the group << 4 | candidate) as 16-bit entries
constexpr int RS_WAVE_LDS = RS_TILE + RS_TAB + RS_Q * 8;   // + best value / best index per query
constexpr int RS_BLOCKS = RS_ROT ? 4 : 2;      // thread blocks per CU
static_assert(SLOTS <= 16 && (RS_NP == 4 || RS_NP == 8) && (RS_WAVE_LDS & 15) == 0 && RS_BLOCKS * (4 * RS_WAVE_LDS + 2048) <= 160 * 1024,
              "corr_rescore_lds: LDS budget");
// byte offset of piece p of vector v in a wave's tile: rotated so that 16 lanes reading the same piece of 16 consecutive vectors
// (64-byte stride) cover all 64 banks
__device__ __forceinline__ unsigned int rs_at(const int v, const int p) { return (unsigned int)(v * RS_VLD + (RS_ROT ? ((p + (v >> 2)) & (RS_NP - 1)) : p) * 16); }

__global__ __launch_bounds__(256, RS_BLOCKS) void corr_rescore_lds_kernel(const float *__restrict__ y_in, const float *__restrict__ y_ref,
                                                                  const float *__restrict__ inv_ref, const float *__restrict__ nrm_in,
                                                                  const int *__restrict__ cand_r, const int *__restrict__ cand_n,
                                                                  const int *__restrict__ flag_count, const int *__restrict__ flag_list,
                                                                  int64_t *__restrict__ max_idx, float *__restrict__ max_val, int n_in,
                                                                  int n_pair, int Cp, int h, int w, unsigned long long *__restrict__ brute)
{
    __shared__ float rv[256];
    __shared__ int ri[256];
    extern __shared__ __align__(16) unsigned char rs_smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned char *const tile = rs_smem + wv * RS_WAVE_LDS;
    unsigned short *const tab = reinterpret_cast<unsigned short *>(tile + RS_TILE);
    float *const bestv = reinterpret_cast<float *>(tile + RS_TILE + RS_TAB);
    int *const besti = reinterpret_cast<int *>(bestv + RS_Q);
    const int pw = w - 2, P = (h - 2) * pw, half = Cp >> 1;
    const long total = (long)n_pair * P;
    const int slot = lane / 9, tap = lane - 9 * slot, ty = tap / 3, tx = tap - 3 * ty;
    const bool lane_live = lane < 9 * RS_U;
    // loader role: request i of a stage fetches piece lane % RS_NP of vector RS_VPI i + lane / RS_NP -- vectors 0..63 query side (vector =
    // the lane that owns it, 63 shadows 62), 64..127 reference side; the first half of the pieces: the even channels of the stage
    const int ld_src = lane / RS_NP, ld_piece = lane & (RS_NP - 1);
    const unsigned int ld_poff = (unsigned int)((ld_piece < RS_NP / 2 ? 0 : half) + 4 * (ld_piece & (RS_NP / 2 - 1)));
    const unsigned int ld_dst = rs_at(ld_src, ld_piece);
    const int rd_v = lane_live ? lane : 62;
    unsigned int rd_off[RS_NP];
#pragma unroll
    for (int j = 0; j < RS_NP; ++j) rd_off[j] = rs_at(rd_v, j);
    const __amdgpu_buffer_rsrc_t srd_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(y_in), 0, 0xffffffffu, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_ref = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(y_ref), 0, 0xffffffffu, 0x00020000);
    const long n_groups = (total + RS_Q - 1) / RS_Q;
    for (long grp = blockIdx.x * 4L + wv; grp < n_groups; grp += (long)gridDim.x * 4L) {
        const long e0 = grp * RS_Q;
        // ---- the group's unit list
        int cnt = 0;
        if (lane < RS_Q && e0 + lane < total) {
            const int n = cand_n[e0 + lane];   // (< 0: the brute-force pass owns this query)
            if (n == 1 && !max_val) max_idx[e0 + lane] = (int64_t)cand_r[(e0 + lane) * SLOTS];
            else if (n > 0) cnt = n;
            bestv[lane] = -__builtin_inff();
            besti[lane] = 0x7fffffff;
        }
        int pre = cnt;   // inclusive prefix sum over the 32 lanes
#pragma unroll
        for (int o = 1; o < RS_Q; o <<= 1) {
            const int up = __shfl_up(pre, o, 64);
            if (lane >= o) pre += up;
        }
        const int U = __builtin_amdgcn_readlane(pre, RS_Q - 1);
        if (U == 0) continue;
        for (int k = 0; k < cnt; ++k) tab[pre - cnt + k] = (unsigned short)((lane << 4) | k);
        __builtin_amdgcn_wave_barrier();
        for (int u0 = 0; u0 < U; u0 += RS_U) {
            // ---- this lane's evaluation (past the list's end: the last one again, not merged)
            const int ui = u0 + (slot < RS_U ? slot : RS_U - 1);
            const bool valid = lane_live && ui < U;
            const unsigned int ent = tab[ui < U ? ui : U - 1];
            const long e = e0 + (ent >> 4);
            const int pair = (int)(e / P), q = (int)(e - (long)pair * P);
            const int in_i = pair % n_in, qy = q / pw, qx = q - qy * pw;
            const int r = cand_r[e * SLOTS + (ent & 15)];
            const int ry = r / pw, rx = r - ry * pw;
            const unsigned int a_off = (unsigned int)(((in_i * h + qy + ty) * w + qx + tx) * Cp);   // (launcher: the maps hold < 2^30 floats)
            const unsigned int b_off = (unsigned int)(((pair * h + ry + ty) * w + rx + tx) * Cp);
            // the loader's sources: vectors RS_VPI i + lane / RS_NP of the query side, then of the reference side
            unsigned int src[RS_NLD];   // byte offsets into y_in (the first half of the requests) / y_ref
#pragma unroll
            for (int i = 0; i < RS_NLD / 2; ++i) {
                const int v = RS_VPI * i + ld_src, owner = v < 63 ? v : 62;
                src[i] = (__shfl(a_off, owner, 64) + ld_poff) * 4u;
                src[RS_NLD / 2 + i] = (__shfl(b_off, owner, 64) + ld_poff) * 4u;
            }
            auto request = [&](f32x4 (&ld)[RS_NLD], const int tt) {
#pragma unroll
                for (int i = 0; i < RS_NLD; ++i)
                    ld[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(i < RS_NLD / 2 ? srd_in : srd_ref, src[i], (unsigned int)tt * 4u, 0));
            };
            f32x4 ld[RS_NLD];
            request(ld, 0);
            float g = 0.0f;
            for (int tt = 0; tt < half; tt += RS_CH) {
#pragma unroll
                for (int i = 0; i < RS_NLD; ++i)   // (vector RS_VPI i + ld_src: the rotation of its pieces does not depend on i)
                    *reinterpret_cast<f32x4 *>(tile + ld_dst + (i < RS_NLD / 2 ? i * RS_VPI : 64 + (i - RS_NLD / 2) * RS_VPI) * RS_VLD) = ld[i];
                if (tt + RS_CH < half) request(ld, tt + RS_CH);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int j = 0; j < RS_NP / 2; ++j) {
                    const f32x4 ae = *reinterpret_cast<const f32x4 *>(tile + rd_off[j]), ao = *reinterpret_cast<const f32x4 *>(tile + rd_off[RS_NP / 2 + j]);
                    const f32x4 be = *reinterpret_cast<const f32x4 *>(tile + rd_off[j] + 64 * RS_VLD), bo = *reinterpret_cast<const f32x4 *>(tile + rd_off[RS_NP / 2 + j] + 64 * RS_VLD);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {   // channel 2 (tt + 4 j + c), then 2 (tt + 4 j + c) + 1
                        g = __builtin_fmaf(ae[c], be[c], g);
                        g = __builtin_fmaf(ao[c], bo[c], g);
                    }
                }
                __builtin_amdgcn_wave_barrier();   // (the tile is read: the next stage may overwrite it)
            }
            // ---- the nine taps in order, times 1 / |ref patch|; merge into the query's best
            const int l0 = 9 * (slot < RS_U ? slot : RS_U - 1);
            float v = __shfl(g, l0, 64);
#pragma unroll
            for (int t = 1; t < 9; ++t) v = v + __shfl(g, l0 + t, 64);
            v = v * inv_ref[(size_t)pair * P + r];
            for (int sl = 0; sl < RS_U; ++sl) {   // (one evaluation at a time: two of a batch may belong to the same query)
                if (lane == 9 * sl && valid) {
                    const int ql = (int)(ent >> 4);
                    const float bv = bestv[ql];
                    const int bi = besti[ql];
                    if (v > bv || (v == bv && r < bi)) { bestv[ql] = v; besti[ql] = r; }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (lane < RS_Q && cnt > 0) {
            const long e = e0 + lane;
            const int bi = besti[lane];
            max_idx[e] = (int64_t)(bi == 0x7fffffff ? 0 : bi);
            if (max_val) {
                const int pair = (int)(e / P), q = (int)(e - (long)pair * P);
                max_val[e] = bestv[lane] / nrm_in[(size_t)(pair % n_in) * P + q];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    rescore_brute(y_in, y_ref, inv_ref, nrm_in, flag_count, flag_list, max_idx, max_val, n_in, Cp, h, w, brute, rv, ri);
}

}  // namespace

// the fixed part of the workspace, rounded up to 16 bytes; the per-lane candidate lists of the row-stationary fp16 pre-filter follow
static int64_t corr_ws_fixed_bytes(int n_pair, int h, int w)
{
    const int64_t P = (int64_t)(h - 2) * (w - 2);
    const int64_t tiles = (int64_t)mrefsr::cdiv(h - 2, T_QY) * mrefsr::cdiv(w - 2, T_QX);
    const int64_t b = (n_pair * P * (SLOTS + 2) + 4 + n_pair * tiles + 2 + 4 * BRUTE_MAX) * (int64_t)sizeof(int);
    return (b + 15) / 16 * 16;
}

// workspace: [cand_r n_pair*P*SLOTS][cand_n n_pair*P][flag_list n_pair*P][flag_count 1 (+3 pad)][tile_flag n_pair*tiles][pad to 8 bytes][brute (u64 best, int done, pad) x BRUTE_MAX] int32
MREFSR_EXPORT int64_t mrefsr_corr_workspace_bytes(int n_pair, int h, int w)
{
    if (n_pair <= 0 || h < 3 || w < 3) return -1;
    const int64_t P = (int64_t)(h - 2) * (w - 2);
    const int64_t tiles = (int64_t)mrefsr::cdiv(h - 2, T_QY) * mrefsr::cdiv(w - 2, T_QX);
    return corr_ws_fixed_bytes(n_pair, h, w) + mrefsr::corr_prefilter_rs16_scratch_bytes(n_pair, h, w);
}

// What mrefsr_corr_top1_prefilter_f32 would launch as its pre-filter for these operands, and the matrix work that kernel issues
// per (sample, reference) pair -- asked of the LIBRARY (not re-derived by the caller from environment variables), so that a
// benchmark's "executed FLOP" always belongs to the code that ran.  Returns the MFMA FLOP per pair (< 0: invalid arguments);
// `kernel_name` (optional, `name_len` bytes) receives the kernel's name; *mfma_dtype: 1 fp16 single plane, 0 bf16 x 3 products.
MREFSR_EXPORT int64_t mrefsr_corr_prefilter_info(int ybf_fmt, int Cp, int h, int w, char *kernel_name, int name_len, int *mfma_dtype)
{
    if (h < 3 || w < 3 || Cp <= 0 || (ybf_fmt != 0 && ybf_fmt != 1)) return -1;
    const char *name = "corr_prefilter_kernel";
    int64_t flop;
    int ab = 0;
#ifdef MREFSR_AB_KERNELS
    {
        const char *e16 = getenv("MREFSR_CORR_PREFILTER_WS16"), *ews = getenv("MREFSR_CORR_PREFILTER_WS"), *est = getenv("MREFSR_CORR_PREFILTER_STREAM");
        if (ybf_fmt == 1 && e16 && e16[0] == '1') ab = 1, name = "corr_prefilter_ws16_kernel";
        else if (ybf_fmt == 0 && Cp == 256 && ews && ews[0] == '1') ab = 2, name = "corr_prefilter_ws_kernel";
        else if (ybf_fmt == 0 && (Cp & 127) == 0 && est && est[0] == '1') ab = 3, name = "corr_prefilter_stream_kernel";
    }
#endif
    if (ab == 0 && ybf_fmt == 1) {
        flop = mrefsr::corr_prefilter_rs16_mfma_flop(h, w, &name);
    } else {   // 128 x 128 x Cp pixel-Gram tiles, every query tile against every reference tile
        const int64_t tiles = (int64_t)mrefsr::cdiv(h - 2, T_QY) * mrefsr::cdiv(w - 2, T_QX);
        flop = 2LL * 128 * 128 * Cp * tiles * tiles * (ybf_fmt == 1 ? 1 : 3);
    }
    if (kernel_name && name_len > 0) snprintf(kernel_name, (size_t)name_len, "%s", name);
    if (mfma_dtype) *mfma_dtype = ybf_fmt;
    return flop;
}

MREFSR_EXPORT int mrefsr_corr_top1_prefilter_f32(const float *y_in, const float *y_ref, const void *ybf_in,
                                                 const void *ybf_ref, const float *inv_ref, const float *nrm_in,
                                                 int64_t *max_idx, float *max_val, void *workspace, int64_t workspace_bytes,
                                                 int n_in, int n_pair, int Cp, int h, int w, int ybf_fmt, const float *tau,
                                                 mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(ybf_fmt == 0 || ybf_fmt == 1, "corr_top1_prefilter: ybf_fmt=%d (0 bf16 hi|lo, 1 fp16)", ybf_fmt);
    if (ybf_fmt == 1 && Cp != 256)
        return mrefsr::fail(MREFSR_E_UNSUPPORTED, "corr_top1_prefilter: the fp16 single-plane pre-filter needs Cp=256 (got %d)", Cp);
    MREFSR_REQUIRE(y_in && y_ref && ybf_in && ybf_ref && inv_ref && nrm_in && max_idx, "corr_top1_prefilter: null pointer");
    MREFSR_REQUIRE(n_in > 0 && n_pair > 0, "corr_top1_prefilter: n_in=%d n_pair=%d", n_in, n_pair);
    MREFSR_REQUIRE(h >= 3 && w >= 3, "corr_top1_prefilter: h=%d w=%d (3x3 patches need h,w >= 3)", h, w);
    if (Cp <= 0 || Cp > 256 || (Cp & 63))
        return mrefsr::fail(MREFSR_E_UNSUPPORTED, "corr_top1_prefilter: Cp=%d must be 64, 128, 192 or 256", Cp);
    const int64_t P = (int64_t)(h - 2) * (w - 2);
    if (n_pair * P >= 0x7fffffffL) return mrefsr::fail(MREFSR_E_UNSUPPORTED, "corr_top1_prefilter: too many queries");
    const int64_t need = mrefsr_corr_workspace_bytes(n_pair, h, w);
    MREFSR_REQUIRE(workspace && workspace_bytes >= need, "corr_top1_prefilter: workspace of %ld bytes required (got %ld)",
                   (long)need, (long)workspace_bytes);
    hipStream_t st = (hipStream_t)stream;
    int *cand_r = (int *)workspace;
    int *cand_n = cand_r + n_pair * P * SLOTS;
    int *flag_list = cand_n + n_pair * P;
    int *flag_count = flag_list + n_pair * P;
    const int tiles_y = mrefsr::cdiv(h - 2, T_QY), tiles_x = mrefsr::cdiv(w - 2, T_QX);
    int *tile_flag = flag_count + 4;
    const size_t n_tf = (size_t)n_pair * tiles_x * tiles_y;
    int *brute_i = tile_flag + n_tf + ((reinterpret_cast<uintptr_t>(tile_flag + n_tf) & 7) ? 1 : 0);   // 8-byte aligned
    unsigned long long *brute = reinterpret_cast<unsigned long long *>(brute_i);
    if (hipMemsetAsync(flag_count, 0, (size_t)((char *)(brute_i + 4 * BRUTE_MAX) - (char *)flag_count), st) != hipSuccess)
        return mrefsr::check_launch("corr_top1_prefilter(memset)");
    // Product kernels: the row-stationary kernel for the fp16 operand (corr_rowstream.hip), the generic tile kernel for the
    // bf16 two-term operand (any Cp).  A -DMREFSR_AB_KERNELS build also carries the earlier generations
    // (tools/ab/corr_prefilter_ab.inc), selected by MREFSR_CORR_PREFILTER_{WS16,WS,STREAM}=1.
    int ab = 0;
#ifdef MREFSR_AB_KERNELS
    {
        const char *e16 = getenv("MREFSR_CORR_PREFILTER_WS16"), *ews = getenv("MREFSR_CORR_PREFILTER_WS"), *est = getenv("MREFSR_CORR_PREFILTER_STREAM");
        if (ybf_fmt == 1 && e16 && e16[0] == '1') ab = 1;
        else if (ybf_fmt == 0 && Cp == 256 && ews && ews[0] == '1') ab = 2;
        else if (ybf_fmt == 0 && (Cp & 127) == 0 && est && est[0] == '1') ab = 3;
    }
    if (ab == 1) {
        const size_t lds = (size_t)PIPE16_LDS_DWORDS * sizeof(int);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(corr_prefilter_ws16_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(corr_prefilter_ws16_kernel, dim3(tiles_x * tiles_y, n_pair), dim3(512), lds, st,
                           (const unsigned short *)ybf_in, (const unsigned short *)ybf_ref, inv_ref, nrm_in, cand_r, cand_n,
                           flag_count, flag_list, tile_flag, n_in, h, w, tiles_x, tiles_y, tau);
    } else if (ab == 2) {
        const size_t lds = (size_t)PIPE_LDS_DWORDS * sizeof(int);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(corr_prefilter_ws_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(corr_prefilter_ws_kernel, dim3(tiles_x * tiles_y, n_pair), dim3(512), lds, st,
                           (const unsigned short *)ybf_in, (const unsigned short *)ybf_ref, inv_ref, nrm_in, cand_r, cand_n,
                           flag_count, flag_list, tile_flag, n_in, h, w, tiles_x, tiles_y);
    } else if (ab == 3) {
        const size_t lds = (size_t)STR_LDS_DWORDS * sizeof(int);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(corr_prefilter_stream_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(corr_prefilter_stream_kernel, dim3(tiles_x * tiles_y, n_pair), dim3(256), lds, st,
                           (const unsigned short *)ybf_in, (const unsigned short *)ybf_ref, inv_ref, nrm_in, cand_r, cand_n,
                           flag_count, flag_list, tile_flag, n_in, Cp, h, w, tiles_x, tiles_y);
    }
#endif
    if (ab == 0 && ybf_fmt == 1) {
        const PrefilterOut po{cand_r, cand_n, flag_count, flag_list, tile_flag};
        void *scratch = reinterpret_cast<char *>(workspace) + corr_ws_fixed_bytes(n_pair, h, w);
        if (int e = mrefsr::launch_corr_prefilter_rs16(ybf_in, ybf_ref, inv_ref, nrm_in, tau, po, n_in, n_pair, h, w, TAU_SCALE16,
                                                       nullptr, scratch, st))
            return e;
    } else if (ab == 0) {
        const size_t lds = (size_t)PRE_LDS_DWORDS * sizeof(int);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(corr_prefilter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(corr_prefilter_kernel, dim3(tiles_x * tiles_y, n_pair), dim3(256), lds, st,
                           (const unsigned short *)ybf_in, (const unsigned short *)ybf_ref, inv_ref, nrm_in, cand_r, cand_n,
                           flag_count, flag_list, tile_flag, n_in, Cp, h, w, tiles_x, tiles_y);
    }
    if (int e = mrefsr::check_launch("corr_prefilter")) return e;
    const long total = (long)n_pair * P;
    // re-scoring: seven evaluations per wave through a wave-private LDS tile (corr_rescore_lds_kernel); MREFSR_CORR_RESCORE=quad keeps the
    // one-evaluation-per-wave kernel (A/B runs: same bits), which also takes maps of 4 GB and more
    const char *ers = getenv("MREFSR_CORR_RESCORE");   // (read per call: tests flip it inside one process)
    const bool small_maps = (size_t)n_in * h * w * Cp < ((size_t)1 << 30) && (size_t)n_pair * h * w * Cp < ((size_t)1 << 30);   // (32-bit byte offsets)
    if (small_maps && !(ers && ers[0] == 'q')) {
        const size_t lds = (size_t)4 * RS_WAVE_LDS;
        static unsigned long long rs_attr = 0;
        if (mrefsr::first_use_on_device(rs_attr))
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(corr_rescore_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        const long groups = (total + RS_Q - 1) / RS_Q, rs_blocks = (groups + 3) / 4;
        hipLaunchKernelGGL(corr_rescore_lds_kernel, dim3((int)(rs_blocks < 4096 ? (rs_blocks < 1024 ? 1024 : rs_blocks) : 4096)), dim3(256), lds, st, y_in,
                           y_ref, inv_ref, nrm_in, cand_r, cand_n, flag_count, flag_list, max_idx, max_val, n_in, n_pair, Cp, h, w, brute);
    } else {
        const long rs_blocks = (total + 3) / 4;   // 4 queries (a wave each) per block and round
        hipLaunchKernelGGL(corr_rescore_kernel, dim3((int)(rs_blocks < 65536 ? (rs_blocks < 1024 ? 1024 : rs_blocks) : 65536)), dim3(256), 0, st, y_in, y_ref,
                           inv_ref, nrm_in, cand_r, cand_n, flag_count, flag_list, max_idx, max_val, n_in, n_pair, Cp, h, w, brute);
    }
    if (int e = mrefsr::check_launch("corr_rescore")) return e;
    // queries whose candidate lists overflowed (maps full of near-ties): when there are more than a
    // handful, the exact single-pass kernel re-does their query tiles (same canonical bits); its
    // blocks return at once otherwise
    return mrefsr::launch_corr_top1_flagged(y_in, y_ref, inv_ref, nrm_in, max_idx, max_val, n_in, n_pair, Cp, h, w, tile_flag,
                                            flag_count, BRUTE_MAX, st);
}
