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
#include "common.h"
#include "corr_cfg.h"

namespace {

using namespace mrefsr_corr;

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

}  // namespace

namespace mrefsr {

int launch_corr_prefilter_rs16(const void *yh_in, const void *yh_ref, const float *inv_ref, const float *nrm_in, const float *tau,
                               const mrefsr_corr::PrefilterOut &out, int n_in, int n_pair, int h, int w, float tau_scale,
                               float *dbg, hipStream_t st)
{
    constexpr int R = 4;
    const int ph = h - 2, pw = w - 2;
    const int ntx = cdiv(pw, RS_NV), nty = cdiv(ph, RsCfg<R>::RO);
    const int tiles_x = cdiv(pw, T_QX), tiles_y = cdiv(ph, T_QY);
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
                                              (hipStream_t)stream);
}
#endif
