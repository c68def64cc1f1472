// Winograd F(2x2, 3x3) convolution, second generation: FOUR waves of 512 registers, one per SIMD, each owning one ROW of the 4 x 4
// transform (xi = (i, 0..3)) for all 64 tiles and all 64 couts of the block's tile -- same arithmetic, packed weights, tile list and
// epilogues as conv_wino.hip (terms 17: fp32-equivalent results from exact fp16 two-term splits, three v_mfma_f32_32x32x16_f16 per
// product; arch_util.py:89-117, ref_mrapa_restoration_arch.py:217-225,271-304, vgg_arch.py), results bit-identical to it.
//
// Why a second kernel.  conv_wino_kernel (8 waves x 2 transform positions, 256 registers each) keeps the matrix pipe 0.18-0.27 busy:
// a chunk step costs 4.5 k clocks for 1.5 k of MFMA time and the output transform 11-14 k clocks per tile, because with two positions
// per wave Y = A^T M A needs the partial sums of all EIGHT waves through LDS (262 KB per tile) and the chunk step serialises transform,
// weight requests (8 waves x 8 loads in a burst), patch stores and a barrier.  What the counters of tools/ubench/issue_costs.hip say
// about gfx950 (clocks per v_mfma_f32_32x32x16_f16 with N other instructions per MFMA, one wave per SIMD):
//   plain fp32 VALU (v_add / v_fma / v_max3): free up to ~6 per MFMA, then 4.8 each (one wave issues one instruction per ~4.8 clocks)
//   v_pk_add_f32 / v_pk_fma_f32: +10 each from the first one on -- packed fp32 does NOT overlap the matrix pipe: not used here
//   v_cvt_pk_f16_f32: free up to 4, then 8 each;  v_fma_mixlo_f16: +5 each, 9 at high density
//   ds_read_b128: 4 LDS clocks per KB per CU;  ds_write_b128: 13.5;  any 1-KB VMEM instruction: 16 clocks of the CU's address path
// The weight fragments (64 KB per 16-channel chunk) + the patch (21 KB) are 85 KB per chunk through that address path = 1.36 k clocks,
// the MFMAs of a chunk 1.54 k: neither operand may go through the address path twice, and LDS cannot take the weights on top of the
// raw patch.  That pins the decomposition: a wave must own ALL tiles and ALL couts of the block's tile for its share of the transform
// positions (weight fragments straight from L2 into registers, read by exactly one wave; every V value formed exactly once), and
// 256 accumulator registers per lane make that share a whole row of the transform:
//   * the column half of the output transform (Z[b] = sum_j A^T[b][j] M[i][j]) is private to the wave, the row half meets FOUR
//     partial sums instead of eight: 131 KB through LDS per tile instead of 262, in a buffer of its own (no aliasing with the patch);
//   * stage one of the input transform (row i of B^T d) is formed once instead of by two waves;
//   * all four waves run the same instruction stream (no early / late groups): every load of the chunk loop has a fixed position in
//     the in-order return queue, the waits are counted per weight-fragment group (j), and the stream is software-pipelined by hand --
//     transform of tile half 1 beside the MFMAs of half 0, transform of the NEXT chunk's half 0 beside the MFMAs of half 1;
//   * the first product of every accumulator takes the constant 0 as C (no 256-register clear per tile).
// Block = 256 threads, persistent (one per CU, XCD-contiguous band of the tile list), accumulators in the AGPR half of the file.
#include <cstdlib>
#include <type_traits>

#include "conv_wino_common.h"

namespace {
using namespace mrefsr_conv;
using namespace mrefsr_wino;

constexpr int NSLOT = 4;                                  // raw 16-channel chunks resident in LDS (slot = chunk counter & 3)
constexpr int X_LD = 32 + 4;                              // floats per tile row of the exchange buffer [wave 4][tile 64][X_LD]
constexpr int X_OFF = NSLOT * RAW_BYTES;
constexpr int X_BYTES = 4 * NTILE * X_LD * 4;
constexpr int SINK_OFF = X_OFF + X_BYTES;                 // 16 bytes for the threads without patch pieces
constexpr int BIAS_OFF = SINK_OFF + 16;
constexpr int LDS_BYTES = BIAS_OFF + BIAS_MAX * 4;
static_assert(LDS_BYTES <= 160 * 1024 && (X_OFF & 15) == 0, "conv_wino4: LDS budget");
constexpr int PROWS = 3, NPF = 6;                         // thread tid < 216 stages 6 pieces: patch rows py, py + 3, ... of one column and quarter
static_assert(NPF * PROWS == PP && PROWS * PP * 4 <= 256, "conv_wino4: patch piece assignment");

__device__ __forceinline__ f32x16 mma16(const u32x4 a, const u32x4 b, const f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// counted waits; the guarded registers pass through as read-write operands (conv_wino_common.h)
template <int N> __device__ __forceinline__ void vm_wait4(u32x4 &a, u32x4 &b, u32x4 &c, u32x4 &d)
{
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void vm_wait4p6(u32x4 &a, u32x4 &b, u32x4 &c, u32x4 &d, f32x4 (&p)[NPF])
{
    asm volatile("s_waitcnt vmcnt(%10)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]) : "n"(N) : "memory");
}
__device__ __forceinline__ void vm_drain6(f32x4 (&p)[NPF])
{
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]) : : "memory");
}

template <int I, int N, class F> __device__ __forceinline__ void sfor(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, N>(f);
    }
}
// patch pieces by buffer loads: a lane outside the image (or past a ragged chunk's channels) carries an offset beyond the descriptor's
// size and reads zeros -- no select per value in front of the LDS store
constexpr unsigned int OOB = 0xffff0000u;   // (wino_launch: an image of the input is smaller than this)
__device__ __forceinline__ void bload16(f32x4 &dst, const unsigned int voff, const __amdgpu_buffer_rsrc_t srd, const unsigned int soff)
{
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(srd), "s"(soff) : "memory");
}

// The transform of one tile half as 128 micro-operations in dependency order (a lane: 8 channels = two quarters g0 of a chunk):
//   [0, 8) reads of quarter 0 (rows ra / rb x 4 columns)   [8, 24) stage 1 of quarter 0: t[c] = a[c] + sr b[c]
//   [24, 32) reads of quarter 1                              [32, 48) stage 2 of quarter 0: the four columns of (B^T d) B
//   [48, 72) splits of quarter 0 (8 pairs x cvt, mixlo, mixhi)   [72, 88) stage 1, [88, 104) stage 2, [104, 128) splits of quarter 1
// so that the step can deal them out between its MFMAs (a wave issues one instruction per ~4.8 clocks: 5-7 beside every MFMA).
struct TState {
    f32x4 a[4], b[4], t[4], v[4];
};
template <int K>
__device__ __forceinline__ void top(TState &s, const unsigned char *const pa, const unsigned char *const pb, const float sr, u32x4 (&vh)[4], u32x4 (&vl)[4])
{
    constexpr int g0 = (K < 24 || (K >= 32 && K < 72)) ? 0 : 1;
    if constexpr (K < 8 || (K >= 24 && K < 32)) {
        constexpr int r = K < 8 ? K : K - 24, c = r >> 1;
        constexpr int co = ((c & 1) * RAW_CP + (c >> 1)) * 16 + g0 * (RAW_Q * 16);
        if constexpr (r & 1) s.b[c] = *reinterpret_cast<const f32x4 *>(pb + co);
        else s.a[c] = *reinterpret_cast<const f32x4 *>(pa + co);
    } else if constexpr ((K >= 8 && K < 24) || (K >= 72 && K < 88)) {
        constexpr int r = K < 24 ? K - 8 : K - 72, c = r >> 2, e = r & 3;
        s.t[c][e] = __builtin_fmaf(sr, s.b[c][e], s.a[c][e]);
    } else if constexpr ((K >= 32 && K < 48) || (K >= 88 && K < 104)) {
        constexpr int r = K < 48 ? K - 32 : K - 88, j = r >> 2, e = r & 3;
        if constexpr (j == 0) s.v[0][e] = s.t[0][e] - s.t[2][e];
        else if constexpr (j == 1) s.v[1][e] = s.t[1][e] + s.t[2][e];
        else if constexpr (j == 2) s.v[2][e] = s.t[2][e] - s.t[1][e];
        else s.v[3][e] = s.t[1][e] - s.t[3][e];
    } else {
        constexpr int r = K < 72 ? K - 48 : K - 104, u = r / 3, sub = r % 3, j = u >> 1, p = u & 1, w = 2 * g0 + p;
        if constexpr (sub == 0) vh[j][w] = pk_f16(s.v[j][2 * p], s.v[j][2 * p + 1]);
        else if constexpr (sub == 1) {
            unsigned int lo;
            asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(vh[j][w]), "v"(s.v[j][2 * p]));
            vl[j][w] = lo;
        } else {
            unsigned int lo = vl[j][w];
            asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(vh[j][w]), "v"(s.v[j][2 * p + 1]));
            vl[j][w] = lo;
        }
    }
}
// micro-operations dealt to each of the 24 MFMA slots of a half step (the rest of a slot: waits, patch stores / requests, fragment requests)
//   first half : slot 0, 6, 12, 18 wait for a fragment group; 6..11 store a patch piece (2 max3 + 1 ds_write); 12..17 request one
//   second half: slot 0 holds the barrier; slots 6 j + 4, 6 j + 5 request two fragments each
constexpr int QA[24] = {4, 4, 7, 7, 7, 7, 3, 3, 3, 3, 3, 3, 4, 6, 6, 6, 6, 6, 5, 7, 7, 7, 7, 7};
constexpr int QB[24] = {4, 4, 7, 7, 4, 4, 7, 7, 6, 6, 4, 4, 6, 6, 6, 6, 4, 4, 6, 6, 6, 6, 4, 4};
constexpr int qsum(const int (&q)[24], const int n)
{
    int t = 0;
    for (int i = 0; i < n; ++i) t += q[i];
    return t;
}
static_assert(qsum(QA, 24) == 128 && qsum(QB, 24) == 128, "conv_wino4: every micro-operation of a transform is dealt exactly once");

#ifdef WINO_STAMP
// instrumentation build (tools/conv_wino_stamp.py): shader-clock totals per phase of a wave's life, summed over all waves
//   first half: 0 up to and including the wait for fragment group 0 | 1 slots 0-5 | 2 slots 6-11 (wait j1, patch stores) | 3 slots 12-17 (wait j2,
//   patch requests) | 4 slots 18-23 (wait j3) || second half: 5 first MFMA + barrier | 6 slots 0-11 | 7 slots 12-23 | 8 cursors ||
//   9 output exchange + epilogue | 10 tile bookkeeping | 11 waves
constexpr int NSTAMP = 12;
__device__ unsigned long long g_wino4_stamp[1024][NSTAMP];
#define W4STAMP(i)                                                     \
    {                                                                 \
        const unsigned long long t_now = __builtin_readcyclecounter(); \
        st_acc[i] += t_now - t_last;                                  \
        t_last = t_now;                                               \
    }
#else
#define W4STAMP(i)
#endif

// timing experiments (results wrong): -DW4_ABL=1 no fragment refills | 2 no patch stores / requests | 3 no transform | 4 no MFMAs |
// 5 no output exchange / epilogue | 6 no barrier in the step
#ifndef W4_ABL
#define W4_ABL 0
#endif

template <int RES>   // 0: no tensor added in the epilogue, 1: residual (after the activation), 2: pre (before it) on the fast path
__global__ __launch_bounds__(256, 1) void conv_wino4_kernel(const ConvArgs A)
{
#ifdef WINO_STAMP
    unsigned long long st_acc[NSTAMP] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_last = __builtin_readcyclecounter();
#endif
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wi = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave = row i of the transform
    const int l31 = lane & 31, kh = lane >> 5;
    const int H = A.H, W = A.W;
    // Workgroups go to the 8 XCDs round-robin: block b works in the contiguous band b & 7 of the tile list (cout block fastest, then
    // x, y, image: neighbouring tiles meet in one L2).
    const int tiles_x = (W + 2 * TT - 1) / (2 * TT), tiles_y = (H + 2 * TT - 1) / (2 * TT);
    const int n_tiles = A.n_cb * tiles_x * tiles_y * A.wino_N, per = (n_tiles + 7) >> 3;
    const int band0 = (blockIdx.x & 7) * per, band1 = min(band0 + per, n_tiles), tstep = gridDim.x >> 3;
    int tile = band0 + (blockIdx.x >> 3);
    if (tile >= band1) return;
    for (int i = tid; i < A.n_cb * NB; i += 256)   // (visible after the prologue's barrier)
        reinterpret_cast<float *>(smem + BIAS_OFF)[i] = (A.bias && i < A.Cout) ? A.bias[i] : 0.f;
    struct Tile { int n, cb, y0, x0; };
    auto decode = [&](const int lin) {
        Tile t;
        t.cb = lin % A.n_cb;
        int r = lin / A.n_cb;
        t.x0 = (r % tiles_x) * (2 * TT);
        r /= tiles_x;
        t.y0 = (r % tiles_y) * (2 * TT);
        t.n = r / tiles_y;
        return t;
    };
    const int n_ch = A.n_ch;

    f32x16 acc[4][2][2];   // [j][tile half][cout half]

    // ---- stage 1: global -> registers -> raw patch in LDS.  Thread tid < 216 owns quarter tid & 3 (4 channels) of patch pixel
    // (py, px) = divmod(tid >> 2, 18) and of the pixels 3, 6, ... 15 rows below it.  The pieces are buffer loads: a lane outside the
    // image or past the channels of a ragged chunk carries the offset OOB and receives zeros.
    f32x4 pf[NPF];
    const int p_q = tid & 3, p_py = (tid >> 2) / PP, p_px = (tid >> 2) - p_py * PP;
    const bool p_have = tid < PROWS * PP * 4;
    const unsigned int praw0 = p_have ? (unsigned int)((p_q * RAW_Q + p_py * RAW_RS + (p_px & 1) * RAW_CP + (p_px >> 1)) * 16) : (unsigned int)SINK_OFF;
    const unsigned int praw_step = p_have ? (unsigned int)(PROWS * RAW_RS * 16) : 0u, praw_slot = p_have ? (unsigned int)RAW_BYTES : 0u;
    const int q4 = (tid & 3) * 4;
    // The request stream of the patch runs three chunks ahead of the multiply, across tile boundaries, with its own cursor
    // (rq_tile, rq_ch); `prepare` sets up the request of the cursor's chunk a step before it is issued and moves the cursor on:
    //   rq_srd   buffer descriptor of the source image (x1 or x2 of the tile's sample: H W ld 4 bytes)
    //   rq_soff  byte offset of the chunk inside a pixel's channels
    //   rq_vo[k] byte offset of piece k's pixel and quarter -- recomputed only at the first chunk of a source and at a ragged last one
    int rq_tile = tile, rq_ch = 0;
    int pg0 = 0;              // pixel index (in an image) of piece 0; piece k lies k PROWS rows below
    unsigned int f_ok = 0;    // bit k: piece k lies inside the image
    const float *xs1 = A.x1, *xs2 = A.x1;
    auto aim = [&](const int lin) {
        if (lin >= band1) {   // past the block's last tile: every piece out of bounds (zeros, never used)
            f_ok = 0;
            return;
        }
        const Tile t = decode(lin);
        const int gy = t.y0 + p_py - 1, gx = t.x0 + p_px - 1;
        pg0 = gy * W + gx;
        f_ok = 0;
#pragma unroll
        for (int k = 0; k < NPF; ++k) f_ok |= (p_have && gy + k * PROWS >= 0 && gy + k * PROWS < H && gx >= 0 && gx < W) ? (1u << k) : 0u;
        xs1 = A.x1 + (size_t)(t.n % A.N1) * H * W * A.ld1;
        xs2 = A.x2 ? A.x2 + (size_t)(t.n % A.N2) * H * W * A.ld2 : A.x1;
    };
    __amdgpu_buffer_rsrc_t rq_srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A.x1), 0, 0, 0x00020000);
    unsigned int rq_soff = 0, rq_vo[NPF];
#pragma unroll
    for (int k = 0; k < NPF; ++k) rq_vo[k] = OOB;
    auto prepare = [&]() {
        const int ch = rq_ch;
        const bool first = ch < A.n_ch1;
        const int cl = first ? ch : ch - A.n_ch1, Cs = first ? A.C1 : A.C2;   // chunk within its source
        rq_soff = __builtin_amdgcn_readfirstlane((unsigned int)(cl * KC * 4));
        if (cl == 0 || (cl + 1) * KC > Cs) {   // (uniform) first chunk of a source: new offsets; ragged last chunk: some quarters end
            if (ch == 0) aim(rq_tile);
            const int ld = first ? A.ld1 : A.ld2;
            const unsigned int bytes = __builtin_amdgcn_readfirstlane((unsigned int)((size_t)H * W * ld * 4));   // (wino_launch: < OOB)
            rq_srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(scalar_ptr(first ? xs1 : xs2)), 0, bytes, 0x00020000);
            const bool lane_ok = cl * KC + q4 < Cs;
#pragma unroll
            for (int k = 0; k < NPF; ++k)
                rq_vo[k] = (lane_ok && ((f_ok >> k) & 1u)) ? ((unsigned int)(pg0 + k * PROWS * W) * (unsigned int)ld + (unsigned int)q4) * 4u : OOB;
        }
        if (++rq_ch == n_ch) rq_ch = 0, rq_tile += tstep;
    };
    float amax = 0.f;          // fp16 range guard: largest |x| seen
    auto request_piece = [&](const int k) { bload16(pf[k], rq_vo[k], rq_srd, rq_soff); };
    auto store_piece = [&](const int slot, const int k) {   // (the caller's counted wait has passed pf through)
        const f32x4 v = pf[k];
        asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(amax) : "v"(v[0]), "v"(v[1]));   // (a NaN input is not caught here: it reaches the output)
        asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(amax) : "v"(v[2]), "v"(v[3]));
        *reinterpret_cast<f32x4 *>(smem + praw0 + slot * praw_slot + k * praw_step) = v;
    };

    // ---- stage 2: the wave's transform.  Row i of B^T d = d[ra] + sr d[rb]:  i 0: d0 - d2,  1: d1 + d2,  2: d2 - d1,  3: d1 - d3;
    // its four columns:  j 0: t0 - t2,  1: t1 + t2,  2: t2 - t1,  3: t1 - t3.  Plain fp32 instructions only (see the header); the
    // micro-operations are `top<K>` above.
    const int t_ra = wi == 0 ? 0 : (wi == 2 ? 2 : 1), t_rb = wi == 0 ? 2 : (wi == 1 ? 2 : (wi == 2 ? 1 : 3));
    const float t_sr = __uint_as_float(__builtin_amdgcn_readfirstlane(wi == 1 ? 0x3f800000u : 0xbf800000u));
    // lane = (tx = l31 & 7, ty low bits = l31 >> 3, k half): quarter 2 kh (+ g0) of pixel (2 ty, 2 tx) of tile half 0
    const unsigned int t_base = (unsigned int)(((2 * kh) * RAW_Q + (2 * (l31 >> 3)) * RAW_RS + (l31 & 7)) * 16);
    const unsigned int t_a = t_base + t_ra * RAW_RS * 16, t_b = t_base + t_rb * RAW_RS * 16;
    auto transform = [&](const int slot, const int tt, u32x4 (&vh)[4], u32x4 (&vl)[4]) {   // all of it at once (prologue)
        TState ts;
        const unsigned char *const pa = smem + t_a + slot * RAW_BYTES + tt * (8 * RAW_RS * 16);
        const unsigned char *const pb = smem + t_b + slot * RAW_BYTES + tt * (8 * RAW_RS * 16);
        sfor<0, 128>([&](auto kc) { top<decltype(kc)::value>(ts, pa, pb, t_sr, vh, vl); });
    };

    // ---- stage 3: the wave's 16 weight fragments of a chunk ((j, cout half) x (uh, ul)), in registers; each is requested again right
    // behind its last MFMA of the chunk (tile half 1) and is used a whole multiply later.  The request stream has its own cursor.
    unsigned int u_voff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) u_voff[j] = (unsigned int)(l31 * KC + kh * 8) * 2u + j * 4096u;   // xi (i, j): 2 planes x 64 couts x 16 cin x 2 bytes each
    const unsigned short *const u_wave = A.wp + (size_t)(wi * 4) * 2 * NB * KC;
    u32x4 uq[4][2][2];   // [j][cout half][uh | ul]
    int uq_tile = tile, uq_ch = 0;
    const void *u_s = nullptr;
    auto frag_cursor = [&]() {   // scalar base of the cursor's chunk of the fragment stream, then the cursor moves on
        if (uq_ch == 0) {
            const int cbn = (uq_tile < band1 ? uq_tile : tile) % A.n_cb;   // (past the last tile: any valid fragments, never used)
            u_s = scalar_ptr(u_wave + (size_t)cbn * n_ch * WCH_HALVES);
        } else {
            u_s = reinterpret_cast<const unsigned char *>(u_s) + WCH_HALVES * 2;
        }
        if (++uq_ch == n_ch) uq_ch = 0, uq_tile += tstep;
    };
    auto frag_pair = [&](const int j, const int ct) {   // 2 loads; the 8 pairs of a chunk always in the order (0,0) (0,1) (1,0) ... (3,1)
        if (ct == 0) gload16u<0>(uq[j][0][0], u_voff[j], u_s), gload16u<2048>(uq[j][0][1], u_voff[j], u_s);
        else gload16u<1024>(uq[j][1][0], u_voff[j], u_s), gload16u<3072>(uq[j][1][1], u_voff[j], u_s);
    };
    // one of the three products of an accumulator (smallest first); FIRST: the accumulator starts at the constant 0
    auto mul1 = [&](const int j, const int tt, const int ct, const int prod, const u32x4 vh, const u32x4 vl, const bool first) {
        f32x16 z;
#pragma unroll
        for (int e = 0; e < 16; ++e) z[e] = 0.f;
        if (W4_ABL == 4) {
            if (first && prod == 0) acc[j][tt][ct] = z;
            acc[j][tt][ct][prod] += __uint_as_float(vh[0] ^ vl[1] ^ uq[j][ct][prod & 1][prod]);
            return;
        }
        if (prod == 0) acc[j][tt][ct] = mma16(uq[j][ct][1], vh, first ? z : acc[j][tt][ct]);
        else if (prod == 1) acc[j][tt][ct] = mma16(uq[j][ct][0], vl, acc[j][tt][ct]);
        else acc[j][tt][ct] = mma16(uq[j][ct][0], vh, acc[j][tt][ct]);
    };

    // ---- the chunk stream.  Chunk counter g (per block, across tiles); its raw patch lives in slot g & 3.  A step is two half steps
    // of 24 slots, a slot = one MFMA + its share of everything else, pinned in this order (sched_barrier: left to itself the compiler
    // emits a whole transform in front of the MFMAs that wait for fragments, and the wave -- alone on its SIMD -- idles the matrix pipe):
    //   first half : MFMAs of tile half 0 (operands from the previous step: slot 6 j + k = product k >> 1 of cout half k & 1 of group j)
    //                beside the transform of tile half 1; a counted wait in front of each fragment group; slots 6..11 store the
    //                patch of chunk g + 2 (requested a step ago), slots 12..17 request the patch of chunk g + 3
    //   second half: MFMAs of tile half 1 beside the transform of tile half 0 of chunk g + 1; the barrier sits behind the first MFMA
    //                (chunk g + 1's patch, stored a step ago, is complete; slot (g + 2) & 3 may be overwritten by the NEXT step's stores);
    //                every fragment pair is requested again (chunk g + 1) behind its last MFMA
    //   then       : the cursors move on (the only branches of a step)
    // In-order return queue at the start of a step:  [patch g + 2: 6] [fragments g: j0 x4, j1 x4, j2 x4, j3 x4]
    //   wait j0: 12 younger may fly | j1: 8 | j2: 4 | (the 6 patch requests of g + 3 join) | j3: 6
    // Hazards on the raw slots: slot (g + 2) & 3 was last read in step g - 2 (two barriers ago); a slot is read a barrier or more
    // after its stores.
    Tile cur = decode(tile);
    int g = 0;
    prepare();                       // chunk 0
#pragma unroll
    for (int k = 0; k < NPF; ++k) request_piece(k);
    vm_drain6(pf);
#pragma unroll
    for (int k = 0; k < NPF; ++k) store_piece(0, k);
    prepare();                       // chunk 1
#pragma unroll
    for (int k = 0; k < NPF; ++k) request_piece(k);
    vm_drain6(pf);
#pragma unroll
    for (int k = 0; k < NPF; ++k) store_piece(1, k);
    prepare();                       // chunk 2: in flight into the loop, OLDER than the fragments of chunk 0
    frag_cursor();
#pragma unroll
    for (int k = 0; k < NPF; ++k) request_piece(k);
#pragma unroll
    for (int j = 0; j < 4; ++j) frag_pair(j, 0), frag_pair(j, 1);
    prepare();                       // chunk 3: requested by the first step
    frag_cursor();                   // fragments of chunk 1: requested by the first step
    __syncthreads();
    const float slope = A.slope_ptr ? *A.slope_ptr : A.slope;
    u32x4 a0h[4], a0l[4];            // operands of tile half 0 of the current chunk
    transform(0, 0, a0h, a0l);

    auto step = [&](auto first_c) {
        constexpr bool first = decltype(first_c)::value;
        u32x4 b1h[4], b1l[4];
        {
            TState ts;
            const unsigned char *const pa = smem + t_a + (g & 3) * RAW_BYTES + 8 * RAW_RS * 16;
            const unsigned char *const pb = smem + t_b + (g & 3) * RAW_BYTES + 8 * RAW_RS * 16;
            const int st_slot = (g + 2) & 3;
            sfor<0, 24>([&](auto mc) {
                constexpr int m = decltype(mc)::value, j = m / 6, k = m % 6;
                if constexpr (m == 0) {
                    vm_wait4p6<12>(uq[0][0][0], uq[0][0][1], uq[0][1][0], uq[0][1][1], pf);
                    W4STAMP(0)
                }
                if constexpr (m == 6) W4STAMP(1)
                if constexpr (m == 12) W4STAMP(2)
                if constexpr (m == 18) W4STAMP(3)
                if constexpr (m == 6) vm_wait4<8>(uq[1][0][0], uq[1][0][1], uq[1][1][0], uq[1][1][1]);
                if constexpr (m == 12) vm_wait4<4>(uq[2][0][0], uq[2][0][1], uq[2][1][0], uq[2][1][1]);
                if constexpr (m == 18) vm_wait4<6>(uq[3][0][0], uq[3][0][1], uq[3][1][0], uq[3][1][1]);
                mul1(j, 0, k & 1, k >> 1, a0h[j], a0l[j], first);
                if constexpr (W4_ABL != 3) sfor<qsum(QA, m), qsum(QA, m + 1)>([&](auto kc) { top<decltype(kc)::value>(ts, pa, pb, t_sr, b1h, b1l); });
                if constexpr (W4_ABL != 2 && m >= 6 && m < 12) store_piece(st_slot, m - 6);
                if constexpr (W4_ABL != 2 && m >= 12 && m < 18) request_piece(m - 12);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        W4STAMP(4)
        {
            TState ts;
            const unsigned char *const pa = smem + t_a + ((g + 1) & 3) * RAW_BYTES;
            const unsigned char *const pb = smem + t_b + ((g + 1) & 3) * RAW_BYTES;
            sfor<0, 24>([&](auto mc) {
                constexpr int m = decltype(mc)::value, j = m / 6, k = m % 6;
                mul1(j, 1, k & 1, k >> 1, b1h[j], b1l[j], first);
                if constexpr (m == 0) {
                    if constexpr (W4_ABL != 6) __syncthreads();
                    W4STAMP(5)
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (m == 12) W4STAMP(6)
                if constexpr (W4_ABL != 3) sfor<qsum(QB, m), qsum(QB, m + 1)>([&](auto kc) { top<decltype(kc)::value>(ts, pa, pb, t_sr, a0h, a0l); });
                if constexpr (W4_ABL != 1 && k == 4) frag_pair(j, 0);
                if constexpr (W4_ABL != 1 && k == 5) frag_pair(j, 1);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        W4STAMP(7)
        prepare();
        frag_cursor();
        ++g;
        W4STAMP(8)
    };

    for (;;) {
        const int tile_n = tile + tstep;
        const bool more = tile_n < band1;
        W4STAMP(10)
        step(std::true_type{});
        for (int s = 1; s < n_ch; ++s) step(std::false_type{});

        // ---- stage 4: output transform.  Columns in registers: Z[b] = sum_j A^T[b][j] M[i][j]  (A^T = [1 1 1 0; 0 1 -1 -1]):
        //   b 0: m0 + m1 + m2,  b 1: m1 - m2 - m3
        // accumulator register e of [j][t][ct] holds cout 32 ct + 8 (e >> 2) + 4 kh + (e & 3) of Winograd tile 32 t + l31.
        // Four passes (cout half hc x output column parity b) through the exchange buffer [wave = row i][tile][X_LD]; thread = (tile T,
        // 4 couts) for T = tid >> 3 and T + 32 adds the four rows up with the row signs -- y[a] = Z0 + Z1 + Z2 (a = 0), Z1 - Z2 - Z3
        // (a = 1) -- and runs the direct kernel's epilogue on the pixels (2 ty + a, 2 tx + b) of the pass.
        const int cb = cur.cb, n = cur.n, y0 = cur.y0, x0 = cur.x0;
        float *const xb = reinterpret_cast<float *>(smem + X_OFF);
        const int Cout = A.Cout;
        const float oscale = A.out_scale;
        const int c4 = (tid & 7) * 4;
        const bool fast = y0 + 2 * TT <= H && x0 + 2 * TT <= W && cb * NB + NB <= Cout && (A.ld_out & 3) == 0 && (Cout & 3) == 0 && A.epilogue == 0 &&
                          (RES == 1 ? !A.pre && (A.ld_res & 3) == 0 : RES == 2 ? A.pre && !A.residual : !A.pre && !A.residual);
        const int ld_r = RES == 2 ? Cout : A.ld_res;   // (pre is dense [pre_N][H][W][Cout], batch-broadcast)
        float4 pool[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};   // epilogue 1: running maximum of a tile's four pixels
        if (W4_ABL == 5) {   // (keep the accumulators alive: without a reader the MFMAs would be dead code)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < 2; ++t) asm volatile("" ::"a"(acc[j][t][0]), "a"(acc[j][t][1]));
        }
#pragma unroll
        for (int hc = 0; hc < (W4_ABL == 5 ? 0 : 2); ++hc) {
            const int co = cb * NB + hc * 32 + c4;
            const bool cok = co < Cout;
            const bool vec = (co + 3 < Cout) && ((A.ld_out & 3) == 0) && ((Cout & 3) == 0);
            const float4 bv = *reinterpret_cast<const float4 *>(smem + BIAS_OFF + (cb * NB + hc * 32 + c4) * 4);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                if (hc | b) __syncthreads();   // the previous pass has been read
                float4 rq[2][2];
                if constexpr (RES != 0) {   // residual / pre term of the pass's pixels, requested at the START of the pass: its latency lies
                                            // under the exchange writes and the barrier
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        const int T = (tid >> 3) + 32 * it, ty = T >> 3, tx = T & 7, gy0 = y0 + 2 * ty, gx = x0 + 2 * tx + b;
                        if (fast) {
                            const float *const r00 = (RES == 1 ? A.residual + ((size_t)(n * H + gy0) * W + gx) * ld_r
                                                               : A.pre + ((size_t)((n % A.pre_N) * H + gy0) * W + gx) * ld_r) + co;
                            rq[it][0] = ld_f4(r00, A.stream_out);
                            rq[it][1] = ld_f4(r00 + (size_t)W * ld_r, A.stream_out);
                        } else if (RES == 1) {
#pragma unroll
                            for (int a = 0; a < 2; ++a) {
                                const int gy = gy0 + a < H ? gy0 + a : H - 1, gxc = gx < W ? gx : W - 1;
                                rq[it][a] = ld_f4(A.residual + (((size_t)n * H + gy) * W + gxc) * A.ld_res + (cok ? co : 0), A.stream_out);
                            }
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    float *const xrow = xb + ((size_t)wi * NTILE + t * 32 + l31) * X_LD + 4 * kh;
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        float4 z;
                        const int e = 4 * qd;
                        const f32x16 &m0 = acc[0][t][hc], &m1 = acc[1][t][hc], &m2 = acc[2][t][hc], &m3 = acc[3][t][hc];
                        if (b == 0) z = make_float4(m0[e] + m1[e] + m2[e], m0[e + 1] + m1[e + 1] + m2[e + 1], m0[e + 2] + m1[e + 2] + m2[e + 2], m0[e + 3] + m1[e + 3] + m2[e + 3]);
                        else z = make_float4(m1[e] + (-m2[e] - m3[e]), m1[e + 1] + (-m2[e + 1] - m3[e + 1]), m1[e + 2] + (-m2[e + 2] - m3[e + 2]), m1[e + 3] + (-m2[e + 3] - m3[e + 3]));   // (conv_wino_kernel's association: same bits)
                        *reinterpret_cast<float4 *>(xrow + 8 * qd) = z;
                    }
                }
                __syncthreads();
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int T = (tid >> 3) + 32 * it, ty = T >> 3, tx = T & 7, gy0 = y0 + 2 * ty, gx = x0 + 2 * tx + b;
                    float4 z[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) z[i] = *reinterpret_cast<const float4 *>(xb + ((size_t)i * NTILE + T) * X_LD + c4);
                    float4 y[2];
                    y[0] = make_float4(z[0].x + z[1].x + z[2].x, z[0].y + z[1].y + z[2].y, z[0].z + z[1].z + z[2].z, z[0].w + z[1].w + z[2].w);
                    y[1] = make_float4(z[1].x - z[2].x - z[3].x, z[1].y - z[2].y - z[3].y, z[1].z - z[2].z - z[3].z, z[1].w - z[2].w - z[3].w);
                    if (fast) {
                        float *const o00 = A.out + ((size_t)(n * H + gy0) * W + gx) * A.ld_out + co;
#pragma unroll
                        for (int a = 0; a < 2; ++a) {
                            float4 v = y[a];
                            v.x = v.x * oscale + bv.x, v.y = v.y * oscale + bv.y, v.z = v.z * oscale + bv.z, v.w = v.w * oscale + bv.w;
                            if constexpr (RES == 2) v.x += rq[it][a].x, v.y += rq[it][a].y, v.z += rq[it][a].z, v.w += rq[it][a].w;
                            if (A.act) {
                                v.x = v.x > 0.f ? v.x : v.x * slope, v.y = v.y > 0.f ? v.y : v.y * slope;
                                v.z = v.z > 0.f ? v.z : v.z * slope, v.w = v.w > 0.f ? v.w : v.w * slope;
                            }
                            if constexpr (RES == 1) v.x += rq[it][a].x, v.y += rq[it][a].y, v.z += rq[it][a].z, v.w += rq[it][a].w;
                            st_f4(o00 + (size_t)a * W * A.ld_out, v, A.stream_out);
                        }
                        continue;
                    }
                    if (A.epilogue == 1) {   // MaxPool2d(2,2) of act(conv + bias) = act(max4 + bias): the tile IS the pooling window
                        float4 m = make_float4(fmaxf(y[0].x, y[1].x), fmaxf(y[0].y, y[1].y), fmaxf(y[0].z, y[1].z), fmaxf(y[0].w, y[1].w));
                        if (b == 0) {
                            pool[it] = m;
                            continue;
                        }
                        m = make_float4(fmaxf(m.x, pool[it].x), fmaxf(m.y, pool[it].y), fmaxf(m.z, pool[it].z), fmaxf(m.w, pool[it].w));
                        float4 v = make_float4(m.x * oscale + bv.x, m.y * oscale + bv.y, m.z * oscale + bv.z, m.w * oscale + bv.w);
                        if (A.act) {
                            v.x = v.x > 0.f ? v.x : v.x * slope, v.y = v.y > 0.f ? v.y : v.y * slope;
                            v.z = v.z > 0.f ? v.z : v.z * slope, v.w = v.w > 0.f ? v.w : v.w * slope;
                        }
                        const int Ho = H >> 1, Wo = W >> 1, py = gy0 >> 1, px = (x0 >> 1) + tx;
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
                    for (int a = 0; a < 2; ++a) {
                        const int gy = gy0 + a;
                        if (!(cok && gy < H && gx < W)) continue;
                        const size_t pix = ((size_t)n * H + gy) * W + gx;
                        float4 v = y[a];
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
                        if constexpr (RES == 1) {
                            v.x += rq[it][a].x, v.y += rq[it][a].y, v.z += rq[it][a].z, v.w += rq[it][a].w;
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
        }
        W4STAMP(9)
        if (!more) break;
        tile = tile_n;
        cur = decode(tile);
    }
    // requests past the last chunk are still in flight: wait, and keep their destination registers "in use" up to here -- to the
    // compiler they were dead after the loop, and anything it had placed in them before the wait would be overwritten on arrival
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < NPF; ++k) asm volatile("" ::"v"(pf[k]) : "memory");
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) asm volatile("" ::"v"(uq[j][ct][sp]) : "memory");
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(a0h[j]), "v"(a0l[j]) : "memory");
    if (A.range_flag && !(amax <= 16000.f)) atomicOr(A.range_flag, 1);
#ifdef WINO_STAMP
    W4STAMP(10)
    st_acc[11] = 1;
    if (lane == 0)
        for (int i = 0; i < NSTAMP; ++i) atomicAdd(&g_wino4_stamp[(blockIdx.x * 4 + wi) & 1023][i], st_acc[i]);
#endif
}

}  // namespace

namespace mrefsr_conv {

// ConvArgs as conv_nhwc.hip's conv_entry fills them (terms 16 semantics: out_scale = 1 / wscale); epilogues 0 / 1 / 2.
// Called by wino_launch (conv_wino.hip) after its descriptor checks.
int wino4_launch(const ConvArgs &b, int blocks_cu, hipStream_t stream)
{
    static unsigned long long attr = 0;
    if (mrefsr::first_use_on_device(attr)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv_wino4_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv_wino4_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv_wino4_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    }
    const bool plain = b.epilogue == 0 && (b.Cout & 3) == 0 && (b.ld_out & 3) == 0;
    if (plain && b.residual && (b.ld_res & 3) == 0) hipLaunchKernelGGL((conv_wino4_kernel<1>), dim3(blocks_cu), dim3(256), LDS_BYTES, stream, b);
    else if (plain && b.pre && !b.residual) hipLaunchKernelGGL((conv_wino4_kernel<2>), dim3(blocks_cu), dim3(256), LDS_BYTES, stream, b);
    else hipLaunchKernelGGL((conv_wino4_kernel<0>), dim3(blocks_cu), dim3(256), LDS_BYTES, stream, b);
    return mrefsr::check_launch("conv_wino4");
}

}  // namespace mrefsr_conv

#ifdef WINO_STAMP
// read-and-reset of the phase clocks (instrumentation builds only)
MREFSR_EXPORT int mrefsr_dbg_wino4_stamps(unsigned long long *out8)   // (NSTAMP = 12 values)
{
    static unsigned long long h[1024][NSTAMP];
    if (hipDeviceSynchronize() != hipSuccess) return mrefsr::fail(MREFSR_E_LAUNCH, "wino4_stamps: sync failed");
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_wino4_stamp), sizeof(h)) != hipSuccess) return mrefsr::fail(MREFSR_E_LAUNCH, "wino4_stamps: read failed");
    for (int i = 0; i < NSTAMP; ++i) out8[i] = 0;
    for (int s = 0; s < 1024; ++s)
        for (int i = 0; i < NSTAMP; ++i) out8[i] += h[s][i], h[s][i] = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_wino4_stamp), h, sizeof(h)) != hipSuccess) return mrefsr::fail(MREFSR_E_LAUNCH, "wino4_stamps: reset failed");
    return 0;
}
#endif
