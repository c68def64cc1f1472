// Correlation + top-1, pass A (pre-filter) in ROW-STATIONARY form: the default for fp16 operands (Cp == 256).
//
// Same contract as the tile pre-filters of corr_prefilter.hip (candidates within a proven window of the approximate
// maximum -> corr_rescore_kernel; ref_map_util.py:54-84 is the specification), different dataflow.  The tile kernels
// compute a 128 x 128 pixel-Gram tile, park it in LDS and box-sum it from there: per useful score that is ~95 bytes
// of LDS traffic (Gram store, nine 16-float segment reads, the B operand read by four waves) and the kernel sits on
// the LDS pipe (LDS busy 45 %, matrix pipe 22 %).  Here the 3x3 box-sum never leaves the registers:
//
//   * a wave keeps R consecutive query pixel rows x 16 pixels x 256 channels (fp16) as the MFMA B operand for its
//     whole life (R * 32 VGPRs) and streams the reference map as 16-pixel row segments, strip by strip (a strip =
//     16 pixel columns, walked top to bottom), through v_mfma_f32_16x16x32_f16:  G_m = <query row a0+m, ref row b>,
//     a 16 x 16 tile with the reference pixel along (lane group, register) and the query pixel along the lane;
//   * the VERTICAL taps are rolling register sums down the strip:  P2 + G_{i+2} completes patch row b-2 of query
//     row a0+i, P2 <- P1 + G_{i+1}, P1 <- G_i   (R - 2 output rows per wave);
//   * the HORIZONTAL taps are the (m+1, n+1), (m+2, n+2) diagonal of the accumulator tile: two DPP row shifts per
//     element (the next lane's next register), with two cross-lane-group fetches per tile (ds_bpermute);
//   * * inv[r], tile maximum against the running threshold, rare candidate path -- all on registers.
//
// The only LDS traffic left is the operand stream itself: one 1 KB fragment read per R MFMAs and wave, staged by
// LDS-DMA (global_load_lds, 16 B per lane) into an 8-deep ring shared by the 8 waves of a block (8 different query
// tiles, one reference stream).  14 of 16 columns of a tile are valid on either side and R - 2 of R query rows:
// (14/16)^2 * (R-2)/R = 38 % of the issued MFMA work is useful at R = 4 (the 128 x 128 tile kernels: 43 %), but
// nothing else competes with the matrix pipe.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "corr_cfg.h"

namespace {

using namespace mrefsr_corr;

__host__ __device__ constexpr int cdiv_i(int a, int b) { return (a + b - 1) / b; }

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int RS_NV = 14;      // valid columns of a 16-pixel tile (two are the halo of the 3-tap sum along x)
constexpr int RS_D = 8;        // operand ring depth, in segments
constexpr int RS_SEG = 2048;   // dwords per segment: 16 pixels x 256 fp16 channels = 8 KB
constexpr int RS_CAP = 8;      // candidates kept per (query, lane group); near-ties cluster in neighbouring reference patches, i.e. in one list
constexpr int RS_WAVES = 8;

template <int R>
struct RsCfg {
    static constexpr int RO = R - 2;                       // output (patch) rows per wave
    static constexpr int NSLOT = RS_WAVES * RO * 64;       // candidate lists per block
    static constexpr int LDS_DWORDS = RS_D * RS_SEG + RS_D * 64 + 2 * RS_CAP * NSLOT + 3 * NSLOT;
};

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v)   // bound_ctrl: lanes whose source is outside the 16-lane row read 0
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// four scalar adds: v_pk_add_f32 (what a vector add becomes) does not overlap the matrix pipe at all (profiles/r5_issue_costs.txt)
__device__ __forceinline__ f32x4 add4(const f32x4 &a, const f32x4 &b)
{
#ifdef MREFSR_RX_PKADD
    return a + b;   // (A/B builds: the packed form)
#endif
    float r0 = a[0] + b[0], r1 = a[1] + b[1], r2 = a[2] + b[2], r3 = a[3] + b[3];
    asm volatile("" : "+v"(r0), "+v"(r1));   // (opaque to the packer)
    asm volatile("" : "+v"(r2), "+v"(r3));
    return f32x4{r0, r1, r2, r3};
}
__device__ __forceinline__ float row_shl1(float v) { return dpp_f<0x101>(v); }   // lane i <- lane i+1 of its row
__device__ __forceinline__ float row_shl2(float v) { return dpp_f<0x102>(v); }   // lane i <- lane i+2
__device__ __forceinline__ float from_lane(float v, int byte_addr)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_addr, __builtin_bit_cast(int, v)));
}

template <int R>
__global__ __launch_bounds__(512) void corr_prefilter_rs16_kernel(
    const unsigned short *__restrict__ yh_in, const unsigned short *__restrict__ yh_ref, const float *__restrict__ inv_ref,
    const float *__restrict__ nrm_in, const float *__restrict__ tau_q, PrefilterOut out, int n_in, int h, int w, int ntx, int nty,
    int tiles_x, int n_tf, float tau_scale, float *__restrict__ dbg)
{
    constexpr int RO = RsCfg<R>::RO, Cp = 256, NSLOT = RsCfg<R>::NSLOT;
    extern __shared__ __attribute__((aligned(16))) unsigned int smem_u[];
    unsigned int *ring = smem_u;                                       // [RS_D][piece 32][pixel 16] x 16 B
    float *invr = reinterpret_cast<float *>(smem_u + RS_D * RS_SEG);   // [RS_D][64]: inverse norms of the segment's patch row
    float *cv = invr + RS_D * 64;                                      // [RS_CAP][NSLOT]
    int *cr = reinterpret_cast<int *>(cv + RS_CAP * NSLOT);            // [RS_CAP][NSLOT]
    float *pmax = reinterpret_cast<float *>(cr + RS_CAP * NSLOT);      // [NSLOT]
    int *pcnt = reinterpret_cast<int *>(pmax + NSLOT);
    float *povf = reinterpret_cast<float *>(pcnt + NSLOT);

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, g = lane >> 4;      // query column of this lane; lane group = reference columns 4g .. 4g+3
    const int pair = blockIdx.y;
    const int ph = h - 2, pw = w - 2, P = ph * pw;
    const int in_i = pair % n_in;
    const unsigned short *yin = yh_in + (size_t)in_i * h * w * Cp;
    const unsigned short *yref = yh_ref + (size_t)pair * h * w * Cp;
    const float *inv = inv_ref + (size_t)pair * P;

    // ---- this wave's query tile: pixel rows a0 .. a0+R-1, pixel columns qx0 .. qx0+15 ----
    const int nwt = ntx * nty;
    int wt = blockIdx.x * RS_WAVES + wv;
    const bool owner = wt < nwt;               // surplus waves of the last block shadow the last tile (they stage and sync)
    if (!owner) wt = nwt - 1;
    const int ty = wt / ntx, tx = wt - ty * ntx;
    const int a0 = ty * RO, qx0 = tx * RS_NV;

    u32x4 A[R][8];   // B operand of the MFMA: column = query pixel n, k = 8 channels of k-group g, per 32-channel step
    {
        const int px = qx0 + n;
#pragma unroll
        for (int m = 0; m < R; ++m) {
            const int py = a0 + m;
            const bool ok = py < h && px < w;
            const unsigned short *src = yin + ((size_t)(ok ? py : 0) * w + (ok ? px : 0)) * Cp + g * 8;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
                A[m][ks] = ok ? *reinterpret_cast<const u32x4 *>(src + ks * 32) : u32x4{0u, 0u, 0u, 0u};
        }
    }

    // ---- per output row: running maximum, threshold, candidate list ----
    float gm[RO], thr[RO], ovf[RO], tau[RO];
    int cnt[RO];
    bool live[RO];
#pragma unroll
    for (int i = 0; i < RO; ++i) {
        const int qy = a0 + i;
        live[i] = owner && n < RS_NV && qx0 + n < pw && qy < ph;
        const size_t q = (size_t)(live[i] ? qy : 0) * pw + (live[i] ? qx0 + n : 0);
        tau[i] = !live[i] ? 0.f : tau_q ? tau_q[(size_t)pair * P + q] : tau_scale * nrm_in[(size_t)in_i * P + q];
        asm volatile("" : "+v"(tau[i]));   // retire this load here: its first real use is inside the streaming loop, where the
                                           // compiler's s_waitcnt vmcnt(0) for it would drain the whole LDS-DMA ring
        gm[i] = -__builtin_inff();
        thr[i] = live[i] ? -__builtin_inff() : __builtin_inff();
        ovf[i] = -__builtin_inff();
        cnt[i] = 0;
    }

    // ---- operand stream: segment (strip sx, pixel row b) = 16 pixels x 512 B, contiguous in the pixel-major map.
    // Two LDS-DMA instructions per wave and segment (exactly two: the s_waitcnt vmcnt arithmetic below counts them):
    // wave wv stages the four 16-byte pieces of k-step wv for the 16 pixels -- LDS image [piece][pixel], so a fragment
    // read (k-step ks: lane = pixel + 16 * piece) is 1 KB contiguous -- and 64 inverse norms of the patch row the
    // segment completes (row b-2 from column 14 sx; every wave writes the same 256 bytes: no wave is special).
    const unsigned int dma_lane_off = (unsigned int)((lane & 15) * (Cp * 2) + (4 * wv + (lane >> 4)) * 16);
    int d_sx = 0, d_b = 0, d_slot = 0;
    auto dma_issue = [&]() {
        const char *src = reinterpret_cast<const char *>(yref) + ((size_t)d_b * w + d_sx * RS_NV) * (Cp * 2) + dma_lane_off;
        __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void *)(ring + d_slot * RS_SEG + wv * 256), 16, 0, 0);
        int ii = (d_b >= 2 ? d_b - 2 : 0) * pw + d_sx * RS_NV + lane;
        ii = ii < P ? ii : P - 1;
        __builtin_amdgcn_global_load_lds(inv + ii, (__attribute__((address_space(3))) void *)(invr + d_slot * 64), 4, 0, 0);
        d_slot = d_slot + 1 == RS_D ? 0 : d_slot + 1;
        if (d_b + 1 < h) ++d_b;
        else if (d_sx + 1 < ntx) { ++d_sx; d_b = 0; }   // past the end: the last segment is harmlessly re-staged
    };
#pragma unroll
    for (int d = 0; d < RS_D; ++d) dma_issue();
    // segment 0 landed (mine: the requests of the RS_D-1 younger segments may still be in flight; everybody's: barrier)
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * (RS_D - 1)) : "memory");
    u32x4 f[8];   // A operand of the MFMA: row = reference pixel (lane & 15), k-group g
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) f[ks] = *reinterpret_cast<const u32x4 *>(ring + (ks * 64 + lane) * 4);

    int slot = 0;
    for (int sx = 0; sx < ntx; ++sx) {
        const int rx_base = sx * RS_NV + 4 * g;
        bool val[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) val[e] = (4 * g + e < RS_NV) && (rx_base + e < pw);
        f32x4 P1[RO], P2[RO];
#pragma unroll
        for (int i = 0; i < RO; ++i) P1[i] = P2[i] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int b = 0; b < h; ++b) {
            // segment s+1 complete (mine: the RS_D-2 younger segments' requests may be in flight; all waves: barrier);
            // every wave's reads of segment s have returned (lgkmcnt), so its slot can be refilled
            const f32x4 iv = *reinterpret_cast<const f32x4 *>(invr + slot * 64 + 4 * g);   // 1 / (|ref patch| + eps): patch row b-2, columns rx_base ..
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * (RS_D - 2)) : "memory");
            dma_issue();   // segment s + RS_D -> the slot of segment s
            const int nslot = slot + 1 == RS_D ? 0 : slot + 1;
            f32x4 G[R];
#pragma unroll
            for (int m = 0; m < R; ++m) G[m] = f32x4{0.f, 0.f, 0.f, 0.f};
            const unsigned int *nb = ring + nslot * RS_SEG + lane * 4;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
                for (int m = 0; m < R; ++m)
                    G[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, f[ks]), __builtin_bit_cast(f16x8, A[m][ks]),
                                                                  G[m], 0, 0, 0);
                f[ks] = *reinterpret_cast<const u32x4 *>(nb + ks * 256);   // the same k-step of the next segment
            }
            slot = nslot;

#pragma unroll
            for (int i = 0; i < RO; ++i) {
                const f32x4 c = P2[i] + G[i + 2];
                P2[i] = P1[i] + G[i + 1];
                P1[i] = G[i];
                if (b < 2) continue;   // (wave-uniform) the first two rows of a strip only fill the partial sums
                // horizontal taps: element (m, n) + (m+1, n+1) + (m+2, n+2); m = 4g + e, so m+1 / m+2 leave the lane
                // group for e >= 2: registers 0 and 1 of the next group, fetched once
                const int up = ((lane + 16) & 63) * 4;
                const float y0 = from_lane(c[0], up), y1 = from_lane(c[1], up);
                float sc[4];
                sc[0] = c[0] + row_shl1(c[1]) + row_shl2(c[2]);
                sc[1] = c[1] + row_shl1(c[2]) + row_shl2(c[3]);
                sc[2] = c[2] + row_shl1(c[3]) + row_shl2(y0);
                sc[3] = c[3] + row_shl1(y0) + row_shl2(y1);
                float tmax = -__builtin_inff();
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sc[e] = val[e] ? sc[e] * iv[e] : -__builtin_inff();
                    tmax = fmaxf(tmax, sc[e]);
                }
#ifdef MREFSR_CORR_DEBUG
                if (dbg && pair == 0 && live[i])
                    for (int e = 0; e < 4; ++e)
                        if (val[e]) dbg[((size_t)(a0 + i) * pw + qx0 + n) * P + (size_t)(b - 2) * pw + rx_base + e] = sc[e];
#endif
                if (tmax >= thr[i]) {   // rare per lane: a new running maximum of this lane's columns, or a near-tie of it
                    if (tmax > gm[i]) { gm[i] = tmax; thr[i] = fmaxf(thr[i], tmax - tau[i]); }
                    const int ls = (wv * RO + i) * 64 + lane;
                    unsigned int todo = (sc[0] >= thr[i] ? 1u : 0u) | (sc[1] >= thr[i] ? 2u : 0u) | (sc[2] >= thr[i] ? 4u : 0u) |
                                        (sc[3] >= thr[i] ? 8u : 0u);
#pragma unroll 1
                    while (todo) {   // one copy of the list code, not four: the registers of the hot loop matter more than this path
                        const int e = __builtin_ctz(todo);
                        todo &= todo - 1;
                        const float vv = e == 0 ? sc[0] : e == 1 ? sc[1] : e == 2 ? sc[2] : sc[3];
                        if (cnt[i] == RS_CAP) {   // prune against the current threshold, then retry
                            int mm = 0;
#pragma unroll 1
                            for (int k = 0; k < RS_CAP; ++k) {
                                const float cvk = cv[k * NSLOT + ls];
                                const int crk = cr[k * NSLOT + ls];
                                if (cvk >= thr[i]) { cv[mm * NSLOT + ls] = cvk; cr[mm * NSLOT + ls] = crk; ++mm; }
                            }
                            cnt[i] = mm;
                        }
                        if (cnt[i] == RS_CAP) { ovf[i] = gm[i]; cnt[i] = 0; }   // overflow: remember how high the dropped entries could be
                        cv[cnt[i] * NSLOT + ls] = vv;
                        cr[cnt[i] * NSLOT + ls] = (b - 2) * pw + rx_base + e;
                        ++cnt[i];
                    }
                }
            }
        }
        // end of a strip: the four lane groups of a query share their maxima (a tighter threshold for the next strip)
#pragma unroll
        for (int i = 0; i < RO; ++i) {
            float o = fmaxf(gm[i], from_lane(gm[i], (lane ^ 16) * 4));
            o = fmaxf(o, from_lane(o, (lane ^ 32) * 4));
            if (live[i]) thr[i] = fmaxf(thr[i], o - tau[i]);
        }
    }

    // ---- merge the four lane groups of each query, publish the candidates ----
#pragma unroll
    for (int i = 0; i < RO; ++i) {
        const int ls = (wv * RO + i) * 64 + lane;
        pmax[ls] = gm[i];
        pcnt[ls] = cnt[i];
        povf[ls] = ovf[i];
    }
    __syncthreads();
    if (g == 0) {
#pragma unroll
        for (int i = 0; i < RO; ++i) {
            if (!live[i]) continue;
            const int l0 = (wv * RO + i) * 64 + n;
            const float gmax = fmaxf(fmaxf(pmax[l0], pmax[l0 + 16]), fmaxf(pmax[l0 + 32], pmax[l0 + 48]));
            const float gthr = gmax - tau[i];
            const int qy = a0 + i, qx = qx0 + n;
            const size_t qo = (size_t)pair * P + (size_t)qy * pw + qx;
            int nn = 0;
            bool over = false;
            for (int gg = 0; gg < 4; ++gg) {
                const int l2 = l0 + gg * 16;
                const int c = pcnt[l2];
                if (povf[l2] >= gthr) over = true;   // entries dropped at an overflow were all <= povf
                for (int k = 0; k < c; ++k)
                    if (cv[k * NSLOT + l2] >= gthr) {
                        if (nn < SLOTS) out.cand_r[qo * SLOTS + nn] = cr[k * NSLOT + l2];
                        ++nn;
                    }
            }
            if (over || nn > SLOTS) {
                out.cand_n[qo] = -1;
                out.flag_list[atomicAdd(out.flag_count, 1)] = (int)qo;
                out.tile_flag[(size_t)pair * n_tf + (qy / T_QY) * tiles_x + qx / T_QX] = 1;
            } else {
                out.cand_n[qo] = nn;
            }
        }
    }
}


// ---- Row-stationary pre-filter with EXCHANGED boundary products (round 3, the default) -----------------------------------
// In corr_prefilter_rs16_kernel a wave with R = 4 query rows finishes only R - 2 = 2 patch rows: the vertical taps of rows 2 and
// 3 need the products of the two query rows BELOW, which a neighbouring wave computes as well -- half of all MFMAs are done
// twice.  Here the 8 waves of a block are stacked vertically on one 16-pixel query column (wave v: query pixel rows a0 + 4v ..
// a0 + 4v + 3) and every reference segment's products are formed ONCE: after its MFMAs of segment b wave v publishes two
// 16 x 16 tiles through LDS,
//     S2(b) = G_2(b-1) + G_3(b)     (patch row 4v+2 after two of its three vertical taps)
//     S3(b) = G_3(b)                (patch row 4v+3 after one)
// and wave v + 1 -- which owns the missing terms G'_0, G'_1 -- completes them one step later, at the same moment and for the same
// reference patch row b - 2 as its own rows 0 and 1:
//     row 4v+2:  S2(b-1) + G'_0(b)                       row 4(v+1)  :  G'_0(b-2) + G'_1(b-1) + G'_2(b)
//     row 4v+3:  S3(b-2) + G'_0(b-1) + G'_1(b)           row 4(v+1)+1:  G'_1(b-2) + G'_2(b-1) + G'_3(b)
// i.e. every wave but the first finishes FOUR patch rows per step with the same 32 MFMAs (wave 0 two): 30 patch rows per block of
// 32 query rows, 94 % instead of 50 % of the issued rows useful ((14/16)^2 * 30/32 = 72 % of the MFMA work, 38 % before).  The
// exchange costs 2 + 2 ds_*_b128 per lane and step and no barrier of its own: the per-step ring barrier orders it (published in
// step b, read in step b + 1; two parities).  The per-lane candidate lists (rare path) live in a global scratch area now -- four
// rows x 8 entries per lane do not fit beside the ring -- everything else (operand ring, window, candidate protocol, flags) is
// the previous kernel's, and so are the results: the same candidate sets reach corr_rescore_kernel.
#ifndef RX_SHARE_MASK
#define RX_SHARE_MASK 15   // the lane groups of a query share their maxima every RX_SHARE_MASK + 1 steps (A/B: 7 and 31 measured no better)
#endif
#ifndef RX_RING8
#define RX_RING8 6
#endif
#ifndef RX_DEFAULT_W
#define RX_DEFAULT_W 8
#endif
constexpr int RX_ROWS = 4;                                 // query pixel rows per wave
constexpr int RX_GCAP = 8;                                 // entries per list in a global spill area: a list that is still full after
                                                           // pruning (more near-ties in one lane's four columns than the LDS list holds:
                                                           // neighbouring patches of smooth maps) moves its entries there instead of overflowing
// W = waves per block.  8 (round 3): one block per CU, the two waves of a SIMD half a step out of phase by construction.
// 4 (round 6): TWO independent blocks per CU (one wave of each on every SIMD): when a block waits at its step barrier for the wave
// that is in the candidate path, the other block's waves have the SIMDs -- at the price of 14 instead of 30 patch rows per
// 16 / 32 pixel rows (the same 192 pixel rows per 158-row map) and a ring of 4 segments.
template <int W>
struct Rx {
    static constexpr int OUT = W * RX_ROWS - 2;            // patch rows finished per block: 30 / 14
    static constexpr int NSLOT = W * 4 * 64;               // candidate lists per block: (wave, output row, lane)
    static constexpr int XCH = 2 * W * 2 * 256;            // dwords: [parity][wave][S2 | S3][64 lanes x 4]
    static constexpr int D = W == 8 ? RX_RING8 : 4;        // operand ring depth (segments)
    static constexpr int CAP = W == 8 ? 3 : 2;             // near-tie candidates per (query, lane group) in LDS (the running maximum itself
                                                           // lives in a register + one LDS word; the lists take what lies inside the window BESIDE it)
    static constexpr int PW = RS_WAVES / W;                // 1-KB pieces of a segment each wave stages
    static constexpr int LDS_DWORDS = D * RS_SEG + D * 64 + XCH + NSLOT + NSLOT / 4 + NSLOT + 2 * CAP * NSLOT;
    static_assert(LDS_DWORDS * 4 * (W == 8 ? 1 : 2) <= 160 * 1024, "corr_prefilter_rx16: LDS budget");
    static_assert(NSLOT <= D * RS_SEG, "the end-of-kernel merge array aliases the operand ring");
};
// The last block row of a map.  A block finishes OUT patch rows; when the rows left for the last block row fit the first HALF of the
// waves' chain (4 * 4 - 2 = 14 rows at W = 8: 158 = 5 * 30 + 8 at the benchmark's 160^2) a block of that row takes TWO column tiles --
// the first half of its waves one, the second half the next, each half its own exchange chain -- instead of parking most of its waves:
// ceil(ntx / 2) blocks instead of ntx in that row (2640 instead of 2880 blocks per 40 pairs at W = 8).
template <int W> __host__ __device__ inline bool rx_split_last_row(int ph)
{
    const int rem = ph - (cdiv_i(ph, Rx<W>::OUT) - 1) * Rx<W>::OUT;
    return rem <= 4 * (W / 2) - 2;
}
// The last column tile of a map.  pw = 11 * 14 + 4 at 160^2: the twelfth column tile has 4 valid queries (+ 2 halo columns) in its 16
// lanes -- a twelfth of the blocks doing 29 % of a block's work.  When two such (valid + halo) groups fit the 16 lanes, a block of that
// column takes TWO block rows side by side: lanes 0 .. vw+1 the block row 2 j, lanes vw+2 .. 2 vw+3 the block row 2 j + 1 (the
// horizontal taps of a group's invalid halo lanes read the neighbouring group: never used).  With the split last row this makes
// 64 blocks per pair at 160^2: 2560 per call = exactly ten rounds of 256 CUs (2640: eleven).
__host__ __device__ inline int rx_packed_width(int pw)   // valid columns of the last column tile if it is lane-packed, else 0
{
    const int ntx = cdiv_i(pw, RS_NV), vw = pw - (ntx - 1) * RS_NV;
    return (ntx >= 2 && 2 * (vw + 2) <= 16) ? vw : 0;
}
template <int W> __host__ __device__ inline int rx_blocks_per_pair(int ph, int pw)
{
    const int ntx = cdiv_i(pw, RS_NV), nty = cdiv_i(ph, Rx<W>::OUT);
    const int ncol = rx_packed_width(pw) ? ntx - 1 : ntx;   // column tiles in the normal tiling
    const int normal = rx_split_last_row<W>(ph) ? ncol * (nty - 1) + (ncol + 1) / 2 : ncol * nty;
    return normal + (rx_packed_width(pw) ? (nty + 1) / 2 : 0);
}

template <int W>
__global__ __launch_bounds__(64 * W, W == 8 ? 1 : 2) void corr_prefilter_rx16_kernel(
    const unsigned short *__restrict__ yh_in, const unsigned short *__restrict__ yh_ref, const float *__restrict__ inv_ref,
    const float *__restrict__ nrm_in, const float *__restrict__ tau_q, PrefilterOut out, float *__restrict__ ovf_g, float *__restrict__ spill_g,
    int n_in, int h, int w, int ntx, int nty, int tiles_x, int n_tf, float tau_scale, float *__restrict__ dbg, int xcd_bands)
{
    // (the block's geometry under the names the body was written with)
    constexpr int RS_WAVES = W, RX_OUT = Rx<W>::OUT, RX_NSLOT = Rx<W>::NSLOT, RX_XCH = Rx<W>::XCH, RX_D = Rx<W>::D, RX_CAP = Rx<W>::CAP, PW = Rx<W>::PW;
    constexpr int R = RX_ROWS, RO = 4, Cp = 256, NSLOT = RX_NSLOT;
    extern __shared__ __attribute__((aligned(16))) unsigned int smem_u[];
    constexpr int RS_D = RX_D, RS_CAP = RX_CAP;                        // (this kernel's ring depth / list capacity)
    // LDS layout: the small per-list arrays FIRST.  A DS instruction's immediate offset reaches 64 KB: with l_tau / l_gidx at the bottom
    // one per-lane base register (+ immediates) addresses every row's window and arg-max word; behind the ring and the exchange
    // area each (array, row) address was a register of its own -- spilled, and a spill reload inside the candidate path is an
    // `s_waitcnt vmcnt(0)` that drains the whole LDS-DMA ring.
    float *l_tau = reinterpret_cast<float *>(smem_u);                  // [wave][lane][4 rows] window of the lane's four queries (one ds_read_b128)
    int *l_gidx = reinterpret_cast<int *>(l_tau + RX_NSLOT);           // [wave][lane][4 rows] reference patch of each list's running maximum
    unsigned char *l_cnt = reinterpret_cast<unsigned char *>(l_gidx + RX_NSLOT);   // [NSLOT] list lengths, written once for the final merge
    unsigned int *ring = smem_u + 2 * RX_NSLOT + RX_NSLOT / 4;         // [RS_D][piece 32][pixel 16] x 16 B
    float *invr = reinterpret_cast<float *>(ring + RS_D * RS_SEG);     // [RS_D][64]: inverse norms of the segment's patch row
    float *xch = invr + RS_D * 64;                                     // [2][RS_WAVES][2][256]
    float *cv = xch + RX_XCH;                                          // [RS_CAP][NSLOT] candidate values ...
    int *cr = reinterpret_cast<int *>(cv + RS_CAP * NSLOT);            // ... and reference patch indices
    // how high the entries dropped at a list overflow could be: written at overflows only, global scratch
    float *l_ovf = ovf_g + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * NSLOT;
    // spill area of this block: [RX_GCAP][NSLOT] values, then [RX_GCAP][NSLOT] reference patch indices
    float *gv = spill_g + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (2 * RX_GCAP * NSLOT);
    int *gr = reinterpret_cast<int *>(gv + RX_GCAP * NSLOT);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, g = lane >> 4;      // query column of this lane; lane group = reference columns 4g .. 4g+3
    // (wave-uniform) which side of its MFMAs this wave's step barrier sits on.  W = 4: every wave before them -- its SIMD partner
    // belongs to the CU's other block and is not synchronised with it
    const bool late = W == 8 && wv >= RS_WAVES / 2;
    const bool sub_before_barrier = W == 8 && wv > RS_WAVES / 2;   // reader and publisher both late: the tiles are fetched before the barrier
    // Workgroups go to the 8 XCDs round-robin in dispatch order: the blocks of one pair -- which all stream the same reference
    // map -- would be spread over all eight L2s.  Re-labelled so that each XCD takes a contiguous eighth of the (tile, pair)
    // list: the blocks that share an L2 work on the same one or two pairs.  (MREFSR_CORR_XCD=0: dispatch order.)
    unsigned lbx = blockIdx.x, lby = blockIdx.y;
    if (xcd_bands) {
        const unsigned gx = gridDim.x, lin = lbx + gx * lby, per = (gx * gridDim.y) / 8;
        if (lin < per * 8) {
            const unsigned l2 = (lin & 7) * per + (lin >> 3);
            lbx = l2 % gx, lby = l2 / gx;
        }
    }
    const int pair = lby;
    const int ph = h - 2, pw = w - 2, P = ph * pw;
    const int in_i = pair % n_in;
    const unsigned short *yin = yh_in + (size_t)in_i * h * w * Cp;
    const unsigned short *yref = yh_ref + (size_t)pair * h * w * Cp;
    const float *inv = inv_ref + (size_t)pair * P;

    // ---- this wave's query rows: pixel rows pr0 .. pr0+3 of the block's column tile ----
    // (last block row in split form -- rx_split_last_row: two column tiles per block, chain position = wv & 3)
    // (last column tile lane-packed -- rx_packed_width: two block rows per block, side by side in the 16 query lanes)
    const int vw = rx_packed_width(pw), ncol = vw ? ntx - 1 : ntx;
    const bool split = rx_split_last_row<W>(ph);
    const int n_full = ncol * (split ? nty - 1 : nty), n_normal = n_full + (split ? (ncol + 1) / 2 : 0);
    int ty, tx, cp = wv;
    int qcol = n, rshift = 0;   // this lane's query column inside the tile; pixel-row shift of its group (packed tile)
    bool lane_ok = true;
    if ((int)lbx < n_full) {
        ty = (int)lbx / ncol, tx = (int)lbx - ty * ncol;
    } else if ((int)lbx < n_normal) {
        ty = nty - 1;
        tx = 2 * ((int)lbx - n_full) + wv / (W / 2);
        cp = wv % (W / 2);
        lane_ok = tx < ncol;
    } else {
        ty = 2 * ((int)lbx - n_normal), tx = ntx - 1;
        const int pk = n >= vw + 2 ? 1 : 0;
        qcol = n - pk * (vw + 2);
        rshift = pk * RX_OUT;
        lane_ok = n < 2 * (vw + 2);
    }
    const int a0 = ty * RX_OUT, qx0 = tx * RS_NV, pr0 = a0 + R * cp;
    const bool active = pr0 < h && tx < ntx;       // (wave-uniform) query rows inside the map: else no MFMAs, only staging + barriers
    const int nvalid = (vw && tx == ntx - 1 && (int)lbx >= n_normal) ? vw : RS_NV;   // valid query columns of this lane's group
    u32x4 A[R][8];
    {
        const int px = qx0 + qcol;
#pragma unroll
        for (int m = 0; m < R; ++m) {
            const int py = pr0 + rshift + m;
            const bool ok = lane_ok && py < h && px < w;
            const unsigned short *src = yin + ((size_t)(ok ? py : 0) * w + (ok ? px : 0)) * Cp + g * 8;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
                A[m][ks] = ok ? *reinterpret_cast<const u32x4 *>(src + ks * 32) : u32x4{0u, 0u, 0u, 0u};
        }
    }
    // ---- the four patch rows this wave finishes: qy = pr0 - 2 + i (i = 0, 1: the wave above's rows 2, 3; none for wave 0) ----
    float gm[RO], thr[RO];
    bool live[RO];
#pragma unroll
    for (int i = 0; i < RO; ++i) {
        const int qy = pr0 + rshift - 2 + i;
        live[i] = (cp > 0 || i >= 2) && lane_ok && tx < ntx && qcol < nvalid && qx0 + qcol < pw && qy < ph;
        const size_t q = (size_t)(live[i] ? qy : 0) * pw + (live[i] ? qx0 + qcol : 0);
        float t = !live[i] ? 0.f : tau_q ? tau_q[(size_t)pair * P + q] : tau_scale * nrm_in[(size_t)in_i * P + q];
        asm volatile("" : "+v"(t));   // retire this load here (see corr_prefilter_rs16_kernel)
        const int ls = (wv * RO + i) * 64 + lane;
        l_tau[(wv * 64 + lane) * 4 + i] = t;
        l_gidx[(wv * 64 + lane) * 4 + i] = 0;
        l_ovf[ls] = -__builtin_inff();
        gm[i] = -__builtin_inff();
        thr[i] = live[i] ? -__builtin_inff() : __builtin_inff();
    }

    // ---- operand stream (as in corr_prefilter_rs16_kernel: two LDS-DMA instructions per wave and segment) ----
    const unsigned int dma_lane_off = (unsigned int)((lane & 15) * (Cp * 2) + (4 * (PW * wv) + (lane >> 4)) * 16);   // (+ 64 j for piece j)
    // The request cursor is kept as what the two instructions consume -- a uniform source pointer, the patch-row base of the inverse
    // norms, the slot's LDS offset -- and ADVANCED by additions (round 6: it used to be re-derived from (strip, row, slot) with
    // multiplies and 64-bit shifts in every step: ~35 scalar + 6 vector instructions between the barrier and the MFMAs).
    int d_sx = 0, d_b = 0;
    const char *d_src = reinterpret_cast<const char *>(yref);   // + ((d_b * w + d_sx * RS_NV) * 512)
    int d_ii = 0;                                               // (d_b >= 2 ? d_b - 2 : 0) * pw + d_sx * RS_NV
    unsigned int d_roff = 0;                                    // d_slot * RS_SEG (dwords); the inverse norms' slot is d_roff / 32
    const size_t d_row = (size_t)w * (Cp * 2);
    auto dma_issue = [&]() {
#pragma unroll
        for (int j = 0; j < PW; ++j)
            __builtin_amdgcn_global_load_lds(d_src + dma_lane_off + 64 * j, (__attribute__((address_space(3))) void *)(ring + d_roff + (PW * wv + j) * 256),
                                             16, 0, 0);
        int ii = d_ii + lane;
        ii = ii < P ? ii : P - 1;
        __builtin_amdgcn_global_load_lds(inv + ii, (__attribute__((address_space(3))) void *)(invr + (d_roff >> 5)), 4, 0, 0);
        d_roff = d_roff + RS_SEG == RS_D * RS_SEG ? 0u : d_roff + RS_SEG;
        if (d_b + 1 < h) {
            ++d_b;
            d_src += d_row;
            if (d_b > 2) d_ii += pw;
        } else if (d_sx + 1 < ntx) {   // next strip (past the end: the last segment is harmlessly re-staged)
            ++d_sx;
            d_b = 0;
            d_src = reinterpret_cast<const char *>(yref) + (size_t)(d_sx * RS_NV) * (Cp * 2);
            d_ii = d_sx * RS_NV;
        }
    };
    // Ring protocol (one slot shallower than corr_prefilter_rs16_kernel's: the second half of a segment's fragments is read in
    // the segment's own step, so that only four fragments are held across the epilogue): segments 0 .. RS_D-2 requested up front;
    // step s waits for segment s+1 (RS_D-3 younger segments' requests may stay in flight), passes the barrier -- every wave has
    // finished step s-1, i.e. all reads of segment s-1 -- and refills that slot with segment s + RS_D - 1.
#pragma unroll
    for (int d = 0; d < RS_D - 1; ++d) dma_issue();
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((PW + 1) * (RX_D - 2)) : "memory");
    u32x4 f[8];   // A operand of the MFMA: row = reference pixel (lane & 15), k-group g; f[0..3] of a segment are read a step ahead
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) f[ks] = *reinterpret_cast<const u32x4 *>(ring + (ks * 64 + lane) * 4);

    // The streaming loop is bound by the SIMD's VALU issue slots (one per four cycles, MFMAs included, shared by its two waves),
    // not by the matrix pipe: every instruction in it counts.  Hence
    //   * the rolling vertical state is not copied: two steps are unrolled and the accumulator sets GA / GB swap the roles
    //     "this step's products" / "the previous step's" (P1[0] = G_0(b-1), P1[1] = G_1(b-1), X2 = G_2(b-1) are simply prev[0..2]);
    //   * the two horizontal taps of a score are two v_add_f32_dpp (inline asm: the compiler emits v_mov_b32_dpp + v_add_f32);
    //   * scaling by the inverse norm and masking of the tile's invalid columns is one fma with a per-strip bias (0 / -inf);
    //   * waves whose query rows lie below the map (last block of a column) only stage and synchronise, in a loop of their own.
    const int T = ntx * h;   // steps = reference segments
    if (!active) {
        for (int t = 0; t < T; ++t) {
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((PW + 1) * (RX_D - 3)) : "memory");
            dma_issue();
        }
    } else {
        f32x4 GA[R], GB[R], P2_0, P2_1, S3p;
#pragma unroll
        for (int m = 0; m < R; ++m) GB[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        P2_0 = P2_1 = S3p = f32x4{0.f, 0.f, 0.f, 0.f};
        float *const x_pub = xch + (wv * 2) * 256 + lane * 4;                    // + parity * RS_WAVES * 512: my S2 | S3 (+256)
        const float *const x_sub = xch + ((wv > 0 ? wv - 1 : 0) * 2) * 256 + lane * 4;   // the wave above's
        const int up = ((lane + 16) & 63) * 4;
        int slot = 0, par = 0, sx = 0, bb = 0;
        unsigned int cnt4 = 0;   // the four list lengths of this lane, 4 bits each (a register: the candidate path runs in most steps)
        unsigned int gcnt4 = 0;  // ... and how many entries each has moved to the global spill area
        float bias[4];   // 0 for the valid reference columns of this lane in the current strip, -inf for the others
        auto strip_setup = [&]() {
#pragma unroll
            for (int e = 0; e < 4; ++e) bias[e] = ((4 * g + e < RS_NV) && (sx * RS_NV + 4 * g + e < pw)) ? 0.f : -__builtin_inff();
        };
        strip_setup();
        // dst = src0 shifted by one / two lanes within its row of 16 (0 beyond the row) + src1.  (The DPP operand must not have
        // been written by one of the two preceding VALU instructions: every use below reads registers produced earlier.)
        auto add_shl1 = [](float shifted, float plain) {
            float r;
            asm volatile("v_add_f32_dpp %0, %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(r) : "v"(shifted), "v"(plain));
            return r;
        };
        auto add_shl2 = [](float shifted, float plain) {
            float r;
            asm volatile("v_add_f32_dpp %0, %1, %2 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(r) : "v"(shifted), "v"(plain));
            return r;
        };
        auto step = [&](f32x4(&cur)[R], f32x4(&prev)[R]) {
            // Waves 0-3 pass the step's barrier BEFORE their MFMAs, waves 4-7 (their SIMD partners) AFTER: between two barriers the
            // first group runs [MFMAs(s), epilogue(s)] and the second [epilogue(s), MFMAs(s+1)] -- on every SIMD one wave's matrix
            // work sits beside the other's VALU work.  What the barrier orders still holds: segment s+1 has landed for everybody
            // (the second group's MFMAs(s+1) need it one barrier earlier than the first group's, which waits for it there anyway
            // for its half-segment prefetch); segment s-1's slot is refilled after barrier s by either group.
            if (!late) {
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((PW + 1) * (RX_D - 3)) : "memory");
                dma_issue();   // segment s + RS_D - 1 -> the slot of segment s - 1
            }
            const int nslot = slot + 1 == RS_D ? 0 : slot + 1;
            const unsigned int *cb_ = ring + slot * RS_SEG + lane * 4, *nb = ring + nslot * RS_SEG + lane * 4;
#pragma unroll
            for (int ks = 4; ks < 8; ++ks) f[ks] = *reinterpret_cast<const u32x4 *>(cb_ + ks * 256);   // this segment's second half
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
                for (int m = 0; m < R; ++m)
                    cur[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, f[ks]), __builtin_bit_cast(f16x8, A[m][ks]),
                                                                    ks == 0 ? zero : cur[m], 0, 0, 0);
                if (ks < 4) f[ks] = *reinterpret_cast<const u32x4 *>(nb + ks * 256);   // the same k-step of the next segment
            }
            const int islot = slot;
            slot = nslot;
            // publish for the wave below: its row 2 after two of three vertical taps, its row 3 after one
            *reinterpret_cast<f32x4 *>(x_pub + par * (RS_WAVES * 512)) = add4(prev[2], cur[3]);
            *reinterpret_cast<f32x4 *>(x_pub + par * (RS_WAVES * 512) + 256) = cur[3];
            par ^= 1;
            // What the wave above published in the PREVIOUS step (now parity `par`; stable from the barrier after its publication
            // until the publisher's step after next).  Waves 5-7 -- reader and publisher both in the late group, where that
            // rewrite falls into the same barrier interval as the reader's epilogue -- fetch it before their barrier, the others
            // after their MFMAs (early group) / after the barrier (wave 4, whose publisher is the early wave 3).  The exchange
            // never runs from a late wave to an early one.
            f32x4 S2n, S3n;
            if (sub_before_barrier) {
                S2n = *reinterpret_cast<const f32x4 *>(x_sub + par * (RS_WAVES * 512));
                S3n = *reinterpret_cast<const f32x4 *>(x_sub + par * (RS_WAVES * 512) + 256);
            }
            if (late) {
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((PW + 1) * (RX_D - 3)) : "memory");
                dma_issue();
            }
            const f32x4 iv = *reinterpret_cast<const f32x4 *>(invr + islot * 64 + 4 * g);   // 1 / (|ref patch| + eps): patch row b-2, columns 14 sx + 4g ..
            if (!sub_before_barrier) {
                S2n = *reinterpret_cast<const f32x4 *>(x_sub + par * (RS_WAVES * 512));
                S3n = *reinterpret_cast<const f32x4 *>(x_sub + par * (RS_WAVES * 512) + 256);
            }
            // the four finished patch rows: the wave above's rows 2, 3 and this wave's rows 0, 1 (all for reference patch row b-2)
            f32x4 c[RO];
            c[2] = add4(P2_0, cur[2]);
            c[3] = add4(P2_1, cur[3]);
            P2_0 = add4(prev[0], cur[1]);
            P2_1 = add4(prev[1], cur[2]);
            c[0] = add4(S2n, cur[0]);
            c[1] = add4(S3p, P2_0);
            S3p = S3n;
#ifdef MREFSR_RX_NOEPI
            if (bb >= 2 && tau_scale < 0.f) {   // (timing experiment)
#else
            if (bb >= 2) {   // (wave-uniform) the first two rows of a strip only fill the partial sums
#endif
                // The eight cross-lane-group fetches (ds_bpermute: an LDS round trip each) go out FIRST and stay there (the scheduler used to
                // sink each pair next to its use: four exposed LDS latencies per step); the five adds of a row that need only this
                // lane group's registers run under their latency, rows 0 and 1 before the first fetched value is touched.
                float y0[RO], y1[RO];
#pragma unroll
                for (int i = 0; i < RO; ++i) y0[i] = from_lane(c[i][0], up), y1[i] = from_lane(c[i][1], up);
                // (the four windows of this lane's queries, for the candidate path: requested here so that their latency is not its)
                const f32x4 tau4 = *reinterpret_cast<const f32x4 *>(l_tau + (wv * 64 + lane) * 4);
                __builtin_amdgcn_sched_barrier(0);
                float sc[RO][4], tmax[RO];
                bool hit = false;
                asm volatile("s_nop 1" ::: "memory");   // (c[] -> first DPP read: two wait states, whatever the compiler placed in between)
                float t0[RO], t1[RO], t2[RO], u0[RO], u1[RO];
                // element (m, n) + (m+1, n+1) + (m+2, n+2): the next lane's next register, twice
                auto own = [&](int i) {    // what needs no neighbouring lane group
                    t0[i] = add_shl1(c[i][1], c[i][0]), t1[i] = add_shl1(c[i][2], c[i][1]), t2[i] = add_shl1(c[i][3], c[i][2]);
                    u0[i] = add_shl2(c[i][2], t0[i]), u1[i] = add_shl2(c[i][3], t1[i]);
                };
                auto finish = [&](int i) {
                    const float t3 = add_shl1(y0[i], c[i][3]);
                    const float u2 = add_shl2(y0[i], t2[i]), u3 = add_shl2(y1[i], t3);
                    sc[i][0] = __builtin_fmaf(u0[i], iv[0], bias[0]);
                    sc[i][1] = __builtin_fmaf(u1[i], iv[1], bias[1]);
                    sc[i][2] = __builtin_fmaf(u2, iv[2], bias[2]);
                    sc[i][3] = __builtin_fmaf(u3, iv[3], bias[3]);
                    tmax[i] = fmaxf(fmaxf(sc[i][0], sc[i][1]), fmaxf(sc[i][2], sc[i][3]));
                    hit |= tmax[i] >= thr[i];   // (rows without a query -- wave 0's first two, rows off the map -- have thr = +inf)
                };
                own(0), own(1);
                finish(0), finish(1);
                own(2), finish(2);
                own(3), finish(3);
#ifdef MREFSR_CORR_DEBUG
#pragma unroll
                for (int i = 0; i < RO; ++i)
                    if (dbg && pair == 0 && live[i])
                        for (int e = 0; e < 4; ++e)
                            if (bias[e] == 0.f) dbg[((size_t)(pr0 + rshift - 2 + i) * pw + qx0 + qcol) * P + (size_t)(bb - 2) * pw + sx * RS_NV + 4 * g + e] = sc[i][e];
#endif
#ifdef MREFSR_RX_NORARE
                hit = false;   // (timing experiment: results are wrong)
#endif
                if (hit) {   // some lane has a new running maximum of its columns in some row, or a near-tie of it.  Taken by SOME wave
                             // of the block in nearly every step (a streaming maximum sets ~ln N records per list, and the per-step
                             // barrier makes every wave wait for the one that is in here), so it is kept short: the four windows come
                             // with one LDS read, the list lengths live in a register, list entries are written, never read back
                             // (except when a list is full).
                    const int rbase = (bb - 2) * pw + sx * RS_NV + 4 * g;
                    // Round 6: a list's running maximum is (gm, l_gidx) -- a register and one LDS word, no list traffic: the COMMON trip
                    // in here is a clear new maximum (a streaming maximum sets ~ln N of them per list), which now costs a threshold
                    // update, the arg-max element and one ds_write.  Only what lies inside the window BESIDE the maximum goes to the
                    // candidate lists (near-ties: the dethroned maximum if the new window still holds it, further elements of the row,
                    // hits below the maximum) -- the same candidate set as before reaches the merge: every list entry and every
                    // maximum is filtered there by the query's final threshold.
#pragma unroll
                    for (int i = 0; i < RO; ++i) {
                        if (tmax[i] >= thr[i]) {
                            const int ls = (wv * RO + i) * 64 + lane;
                            const float ogm = gm[i];
                            const bool newmax = tmax[i] > ogm;
                            if (newmax) { gm[i] = tmax[i]; thr[i] = fmaxf(thr[i], tmax[i] - tau4[i]); }
                            // the row's arg-max element (lowest e on equal values)
                            const int em = sc[i][0] == tmax[i] ? 0 : sc[i][1] == tmax[i] ? 1 : sc[i][2] == tmax[i] ? 2 : 3;
                            unsigned int todo = (sc[i][0] >= thr[i] ? 1u : 0u) | (sc[i][1] >= thr[i] ? 2u : 0u) | (sc[i][2] >= thr[i] ? 4u : 0u) |
                                                (sc[i][3] >= thr[i] ? 8u : 0u);
                            if (newmax) todo &= ~(1u << em);
                            const bool push_old = newmax && ogm >= thr[i];   // (ogm = -inf before the first maximum: never)
                            if (todo != 0u || push_old) {   // near-ties: rare
                                int cn = (cnt4 >> (4 * i)) & 15;
                                float vv = ogm;
                                int rr = push_old ? l_gidx[(wv * 64 + lane) * 4 + i] : 0;
                                bool pending = push_old;
#pragma unroll 1
                                while (pending || todo) {
                                    if (!pending) {
                                        const int e = __builtin_ctz(todo);
                                        todo &= todo - 1;
                                        vv = e == 0 ? sc[i][0] : e == 1 ? sc[i][1] : e == 2 ? sc[i][2] : sc[i][3];
                                        rr = rbase + e;
                                    }
                                    pending = false;
                                    if (cn == RS_CAP) {   // prune against the current threshold, then retry
                                        int mm = 0;
#pragma unroll 1
                                        for (int k = 0; k < RS_CAP; ++k) {
                                            const float cvk = cv[k * NSLOT + ls];
                                            const int crk = cr[k * NSLOT + ls];
                                            if (cvk >= thr[i]) { cv[mm * NSLOT + ls] = cvk; cr[mm * NSLOT + ls] = crk; ++mm; }
                                        }
                                        cn = mm;
                                    }
                                    if (cn == RS_CAP) {   // still full: move the list to the spill area (kept until the final merge, which
                                                          // filters by the final threshold); no room there either -> overflow: drop it
                                                          // and remember how high the dropped entries could be
                                        const int gn = (gcnt4 >> (4 * i)) & 15;
                                        if (gn + RS_CAP <= RX_GCAP) {
#pragma unroll 1
                                            for (int k = 0; k < RS_CAP; ++k) {
                                                gv[(gn + k) * NSLOT + ls] = cv[k * NSLOT + ls];
                                                gr[(gn + k) * NSLOT + ls] = cr[k * NSLOT + ls];
                                            }
                                            gcnt4 += (unsigned int)RS_CAP << (4 * i);
                                        } else {
                                            l_ovf[ls] = gm[i];
                                        }
                                        cn = 0;
                                    }
                                    cv[cn * NSLOT + ls] = vv;
                                    cr[cn * NSLOT + ls] = rr;
                                    ++cn;
                                }
                                cnt4 = (cnt4 & ~(15u << (4 * i))) | ((unsigned int)cn << (4 * i));
                            }
                            if (newmax) l_gidx[(wv * 64 + lane) * 4 + i] = rbase + em;
                        }
                    }
                }
            }
            if ((bb & RX_SHARE_MASK) == RX_SHARE_MASK || bb + 1 == h) {   // every 16 steps: the four lane groups of a query share their maxima (each is a lower
                                                    // bound of the query's: a tighter threshold, fewer trips into the candidate path)
                const f32x4 tq = *reinterpret_cast<const f32x4 *>(l_tau + (wv * 64 + lane) * 4);
#pragma unroll
                for (int i = 0; i < RO; ++i) {
                    float o = fmaxf(gm[i], from_lane(gm[i], (lane ^ 16) * 4));
                    o = fmaxf(o, from_lane(o, (lane ^ 32) * 4));
                    if (live[i]) thr[i] = fmaxf(thr[i], o - tq[i]);
                }
            }
            if (++bb == h) {   // next strip
                bb = 0;
                ++sx;
                strip_setup();
            }
        };
        int t = 0;
        for (; t + 1 < T; t += 2) {
            step(GA, GB);
            step(GB, GA);
        }
        if (t < T) step(GA, GB);
#pragma unroll
        for (int i = 0; i < RO; ++i)   // (LDS count in the low nibble, spilled count in the high one)
            l_cnt[(wv * RO + i) * 64 + lane] = (unsigned char)(((cnt4 >> (4 * i)) & 15) | (((gcnt4 >> (4 * i)) & 15) << 4));
    }

    // ---- merge the four lane groups of each query, publish the candidates (the merge array aliases the drained ring) ----
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    float *pmax = reinterpret_cast<float *>(ring);
    const unsigned char *pcnt = l_cnt;
    const float *povf = l_ovf;   // (own lanes' stores of this wave: ordered by the vmcnt(0) above)
#pragma unroll
    for (int i = 0; i < RO; ++i) pmax[(wv * RO + i) * 64 + lane] = gm[i];
    __syncthreads();
    if (g == 0) {
#pragma unroll
        for (int i = 0; i < RO; ++i) {
            if (!live[i]) continue;
            const int l0 = (wv * RO + i) * 64 + n;
            const float gmax = fmaxf(fmaxf(pmax[l0], pmax[l0 + 16]), fmaxf(pmax[l0 + 32], pmax[l0 + 48]));
            const float gthr = gmax - l_tau[(wv * 64 + n) * 4 + i];
            const int qy = pr0 + rshift - 2 + i, qx = qx0 + qcol;
            const size_t qo = (size_t)pair * P + (size_t)qy * pw + qx;
            int nn = 0;
            bool over = false;
            for (int gg = 0; gg < 4; ++gg) {
                const int l2 = l0 + gg * 16;
                const int c = pcnt[l2] & 15, gc = pcnt[l2] >> 4;
                if (povf[l2] >= gthr) over = true;   // entries dropped at an overflow were all <= povf
                if (pmax[l2] >= gthr) {   // the lane group's running maximum (-inf: the group never scored a valid column)
                    if (nn < SLOTS) out.cand_r[qo * SLOTS + nn] = l_gidx[(wv * 64 + n + gg * 16) * 4 + i];
                    ++nn;
                }
                for (int k = 0; k < c; ++k)
                    if (cv[k * NSLOT + l2] >= gthr) {
                        if (nn < SLOTS) out.cand_r[qo * SLOTS + nn] = cr[k * NSLOT + l2];
                        ++nn;
                    }
                for (int k = 0; k < gc; ++k)   // (written by lanes of this wave; ordered by the vmcnt(0) + barrier above)
                    if (gv[k * NSLOT + l2] >= gthr) {
                        if (nn < SLOTS) out.cand_r[qo * SLOTS + nn] = gr[k * NSLOT + l2];
                        ++nn;
                    }
            }
            if (over || nn > SLOTS) {
                out.cand_n[qo] = -1;
                out.flag_list[atomicAdd(out.flag_count, 1)] = (int)qo;
                out.tile_flag[(size_t)pair * n_tf + (qy / T_QY) * tiles_x + qx / T_QX] = 1;
            } else {
                out.cand_n[qo] = nn;
            }
        }
    }
}

}  // namespace

namespace mrefsr {

// waves per block of corr_prefilter_rx16_kernel: MREFSR_CORR_W=8 / 4 (A/B runs; read per call)
static int rx_waves()
{
    const char *e = getenv("MREFSR_CORR_W");
    return (e && e[0] == '8') ? 8 : (e && e[0] == '4') ? 4 : RX_DEFAULT_W;
}

int64_t corr_prefilter_rs16_scratch_bytes(int n_pair, int h, int w)
{   // corr_prefilter_rx16_kernel: per block [NSLOT] floats (the overflow marks of its candidate lists) + the spill area
    // [2][RX_GCAP][NSLOT] words; the larger of the two block geometries
    const int64_t ntx = cdiv(w - 2, RS_NV);
    const int64_t b8 = ntx * cdiv(h - 2, Rx<8>::OUT) * Rx<8>::NSLOT, b4 = ntx * cdiv(h - 2, Rx<4>::OUT) * Rx<4>::NSLOT;
    return (int64_t)n_pair * (b8 > b4 ? b8 : b4) * 4 * (1 + 2 * RX_GCAP);
}

static bool rx_enabled()
{
    const char *e = getenv("MREFSR_CORR_RS_DUP");   // =1: the previous form (every wave recomputes its two halo rows), A/B runs
    return !(e && e[0] == '1');
}

// MFMA FLOP the fp16 pre-filter issues per (sample, reference) pair at this map size, and the kernel's name (bench.py's roofline)
int64_t corr_prefilter_rs16_mfma_flop(int h, int w, const char **name)
{
    const int ph = h - 2, pw = w - 2;
    const int64_t ntx = cdiv(pw, RS_NV), per_wave_step = 32LL * 16384;   // 8 k-steps x 4 query rows of v_mfma_f32_16x16x32_f16
    if (rx_enabled()) {
        int64_t waves = 0;   // waves with query rows inside the map: per normal column tile ...
        const int W = rx_waves(), out = W == 8 ? Rx<8>::OUT : Rx<4>::OUT;
        const bool split = W == 8 ? rx_split_last_row<8>(ph) : rx_split_last_row<4>(ph);
        const int nty = cdiv(ph, out), ncol = rx_packed_width(pw) ? ntx - 1 : ntx;
        for (int ty = 0; ty < nty; ++ty)
            for (int v = 0; v < (ty == nty - 1 && split ? W / 2 : W); ++v) waves += ty * out + RX_ROWS * v < h;
        waves *= ncol;
        if (rx_packed_width(pw))   // ... and of the lane-packed last column tile (two block rows per block)
            for (int ty = 0; ty < nty; ty += 2)
                for (int v = 0; v < W; ++v) waves += ty * out + RX_ROWS * v < h;
        if (name) *name = W == 8 ? "corr_prefilter_rx16_kernel<8>" : "corr_prefilter_rx16_kernel<4>";
        return waves * ntx * h * per_wave_step;
    }
    const int64_t nwt = ntx * cdiv(ph, RsCfg<4>::RO);
    if (name) *name = "corr_prefilter_rs16_kernel<4>";
    return (nwt + RS_WAVES - 1) / RS_WAVES * RS_WAVES * ntx * h * per_wave_step;
}

int launch_corr_prefilter_rs16(const void *yh_in, const void *yh_ref, const float *inv_ref, const float *nrm_in, const float *tau,
                               const mrefsr_corr::PrefilterOut &out, int n_in, int n_pair, int h, int w, float tau_scale,
                               float *dbg, void *scratch, hipStream_t st)
{
    const int ph = h - 2, pw = w - 2;
    const int tiles_x = cdiv(pw, T_QX), tiles_y = cdiv(ph, T_QY);
    if (scratch && rx_enabled()) {
        const int ntx = cdiv(pw, RS_NV);
        const char *ex = getenv("MREFSR_CORR_XCD");
        auto go = [&](auto wtag) {
            constexpr int W = decltype(wtag)::value;
            const int nty = cdiv(ph, Rx<W>::OUT);
            const size_t lds = (size_t)Rx<W>::LDS_DWORDS * sizeof(int);
            const int nb = rx_blocks_per_pair<W>(ph, pw);
            const int xcd = (ex ? ex[0] != '0' : 1) && (long)nb * n_pair >= 512;
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(corr_prefilter_rx16_kernel<W>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(corr_prefilter_rx16_kernel<W>, dim3(nb, n_pair), dim3(64 * W), lds, st, (const unsigned short *)yh_in,
                               (const unsigned short *)yh_ref, inv_ref, nrm_in, tau, out, reinterpret_cast<float *>(scratch),
                               reinterpret_cast<float *>(scratch) + (size_t)n_pair * nb * Rx<W>::NSLOT, n_in, h, w, ntx, nty, tiles_x, tiles_x * tiles_y,
                               tau_scale, dbg, xcd);
        };
        if (rx_waves() == 8) go(std::integral_constant<int, 8>{});
        else go(std::integral_constant<int, 4>{});
        return check_launch("corr_prefilter_rx16");
    }
    constexpr int R = 4;
    const int ntx = cdiv(pw, RS_NV), nty = cdiv(ph, RsCfg<R>::RO);
    const size_t lds = (size_t)RsCfg<R>::LDS_DWORDS * sizeof(int);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(corr_prefilter_rs16_kernel<R>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    hipLaunchKernelGGL(corr_prefilter_rs16_kernel<R>, dim3(cdiv((long)ntx * nty, RS_WAVES), n_pair), dim3(512), lds, st,
                       (const unsigned short *)yh_in, (const unsigned short *)yh_ref, inv_ref, nrm_in, tau, out, n_in, h, w, ntx, nty,
                       tiles_x, tiles_x * tiles_y, tau_scale, dbg);
    return check_launch("corr_prefilter_rs16");
}

}  // namespace mrefsr

#ifdef MREFSR_CORR_DEBUG
// Debug build only (make EXTRA=-DMREFSR_CORR_DEBUG; tools/corr_rs_debug.py): the approximate scores the row-stationary
// pre-filter forms for pair 0, scores[q * P + r] (caller pre-fills with NaN: untouched entries = (q, r) never scored).
MREFSR_EXPORT int mrefsr_dbg_corr_rs16_scores(const void *yh_in, const void *yh_ref, const float *inv_ref, const float *nrm_in,
                                              const float *tau, float *scores, void *workspace, int h, int w, void *stream)
{
    using namespace mrefsr_corr;
    const long P = (long)(h - 2) * (w - 2);
    int *ws = (int *)workspace;   // layout of mrefsr_corr_top1_prefilter_f32 for n_pair = 1
    const PrefilterOut po{ws, ws + P * SLOTS, ws + P * (SLOTS + 2), ws + P * (SLOTS + 1), ws + P * (SLOTS + 2) + 4};
    if (hipMemsetAsync(po.flag_count, 0, 4 * sizeof(int), (hipStream_t)stream) != hipSuccess) return MREFSR_E_LAUNCH;
    return mrefsr::launch_corr_prefilter_rs16(yh_in, yh_ref, inv_ref, nrm_in, tau, po, 1, 1, h, w, 2.0f * 1.01f * 1.1e-3f, scores,
                                              nullptr, (hipStream_t)stream);
}
#endif
