// Winograd F(2x2, 3x3) convolution, second generation: FOUR waves of 512 registers, one per SIMD, each owning one ROW of the 4 x 4
// transform (xi = (i, 0..3)) for all 64 tiles and all 64 couts of the block's tile -- same arithmetic, packed weights and tile list as
// conv_wino.hip (terms 17: fp32-equivalent results from exact fp16 two-term splits, three v_mfma_f32_32x32x16_f16 per product;
// arch_util.py:89-117, ref_mrapa_restoration_arch.py:217-225,271-304, vgg_arch.py), results bit-identical to it.
//
// Why a second kernel.  conv_wino_kernel (8 waves x 2 transform positions, 256 registers each) keeps the matrix pipe 0.18-0.27 busy:
// a chunk step costs 4.5 k clocks for 1.5 k of MFMA time and the output transform 11-14 k clocks per tile, because with two positions
// per wave Y = A^T M A needs the partial sums of all EIGHT waves through LDS (262 KB per tile) and the chunk step serialises transform,
// weight requests (8 waves x 8 loads in a burst), patch stores and a barrier.  What tools/ubench/issue_costs.hip measured on gfx950
// (profiles/r5_issue_costs.txt: clocks per v_mfma_f32_32x32x16_f16 with N other instructions per MFMA, one wave per SIMD):
//   plain fp32 VALU (v_add / v_fma / v_max3): free up to ~6 per MFMA, then 4.8 each (one wave issues one instruction per ~4.8 clocks)
//   v_pk_add_f32 / v_pk_fma_f32: +10 each from the first one on -- packed fp32 does NOT overlap the matrix pipe: not used here
//   v_cvt_pk_f16_f32: free up to 4, then 8 each;  v_fma_mixlo_f16: +5 each, 9 at high density
//   ds_read_b128: 4 LDS clocks per KB per CU;  ds_write_b128: 13.5;  any 1-KB VMEM instruction: 16 clocks of the CU's address path
// The weight fragments (64 KB per 16-channel chunk) + the patch (21 KB) are 85 KB per chunk through that address path = 1.36 k clocks,
// the MFMAs of a chunk 1.54 k: neither operand may go through the address path twice.  That pins the decomposition: a wave owns ALL
// tiles and ALL couts of the block's tile for its share of the transform positions (every weight fragment is fetched by exactly one
// wave, every V value formed exactly once), and 256 accumulator registers per lane make that share a whole row of the transform:
//   * the column half of the output transform (Z[b] = sum_j A^T[b][j] M[i][j]) is private to the wave, the row half meets FOUR
//     partial sums instead of eight: 131 KB through LDS per tile instead of 262;
//   * stage one of the input transform (row i of B^T d) is formed once instead of by two waves;
//   * all four waves run the same instruction stream: every memory operation of the chunk loop has a fixed position in the in-order
//     return queue, the waits are counted per weight-fragment group (j), and the stream is software-pipelined by hand -- transform of
//     tile half 1 beside the MFMAs of half 0, transform of the NEXT chunk's half 0 beside the MFMAs of half 1, one MFMA per slot;
//   * the first product of every accumulator takes the constant 0 as C (no 256-register clear per tile).
// Data movement of the loop (profiles/r5_lds_dma_semantics.txt: the instruction offset moves the LDS address as well, M0 reaches all
// 160 KB): the WEIGHT fragments arrive by LDS-DMA (`global_load_lds_dwordx4`) into a ring private to each wave -- the wave reads only
// what it fetched itself, its own counted vmcnt orders it, no barrier, no VGPR destination; the PATCH pieces are hand-waited REGISTER
// loads (`buffer_load_dwordx4 ... offen`, four lanes per pixel = one 64-byte request, bload16 -> pf[] -> store_piece: a lane outside
// the image or past a ragged chunk's channels carries an offset beyond the descriptor and reads zeros) that pass through the counted
// waits as read-write operands and are stored into the transform's layout by the wave itself.  The first version of this kernel
// fetched the weight fragments into registers too, as conv_wino_kernel does: with ~120 registers of requests in flight across the
// epilogue the register allocator spilled them around it -- i.e. stored them BEFORE their data had arrived and handed the registers to
// the epilogue, where the arriving data then landed in somebody else's values (wrong outputs), and every spill reload is a
// `vmcnt(0)` behind the previous pass's output stores (16 k clocks per tile).  With the fragments on LDS-DMA only the six patch
// registers are in flight, and they are pinned through every wait.
// Block = 256 threads, persistent (one per CU, XCD-contiguous band of the tile list), accumulators in the AGPR half of the file.
// Served: whole tiles (H, W multiples of 16), whole cout blocks, the plain and the max-pool epilogue (wino4_serves); everything else is
// conv_wino_kernel's.
#include <cstdlib>
#include <type_traits>

#include "conv_wino_common.h"

namespace {
using namespace mrefsr_conv;
using namespace mrefsr_wino;

constexpr int NSLOT = 2;                                  // raw 16-channel chunks resident in LDS (slot = chunk counter mod 2)
constexpr int SLOT_BYTES = RAW_BYTES;
constexpr int RING_OFF = 0;                               // weight fragments: [wave 4][group j 4][uh ct0 | uh ct1 | ul ct0 | ul ct1] x 1 KB
                                                          // (first: the LDS base of a DMA -- M0 -- stays a multiple of 4 KB)
constexpr int RING_BYTES = 4 * 4 * 4096;
constexpr int X_LD = 32 + 4;                              // floats per tile row of the exchange buffer [wave 4][tile 64][X_LD]
constexpr int X_BYTES = 4 * NTILE * X_LD * 4;
constexpr int SLOT_OFF = RING_OFF + RING_BYTES;
constexpr int BIAS_OFF = SLOT_OFF + NSLOT * SLOT_BYTES;
constexpr int SINK_OFF = BIAS_OFF + BIAS_MAX * 4;         // 16 bytes per thread for the stores of the threads without patch pieces
constexpr int X_OFF = SINK_OFF + 256 * 16;                // the exchange buffer of the epilogue, a region of its own: the fragments of the
                                                          // NEXT tile's first chunk land in the ring while the epilogue runs
constexpr int LDS_BYTES = X_OFF + X_BYTES;
static_assert(LDS_BYTES <= 160 * 1024 && (RING_OFF & 4095) == 0 && (SLOT_OFF & 15) == 0 && (X_OFF & 15) == 0, "conv_wino4: LDS budget");
constexpr int NPD = 6, PROWS = 3;                         // thread tid < 216 stages 6 pieces: patch rows py, py + 3, ... of one column and quarter
static_assert(NPD * PROWS == PP && PROWS * PP * 4 <= 256, "conv_wino4: patch piece assignment");
constexpr unsigned int OOB = 0xffff0000u;                 // (wino_launch: an image of the input is smaller than this)

__device__ __forceinline__ f32x16 mma16(const u32x4 a, const u32x4 b, const f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <int I, int N, class F> __device__ __forceinline__ void sfor(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, N>(f);
    }
}

// ---- LDS-DMA.  M0 (the LDS base of a DMA) is the compiler's: it is written in the statement that uses it.
// one piece of the patch into a register (hand-waited: the counted waits pass the registers through); voff = the lane's byte offset
// in the source image (OOB: zeros), soff = the chunk's
__device__ __forceinline__ void bload16(f32x4 &dst, const unsigned int voff, const __amdgpu_buffer_rsrc_t srd, const unsigned int soff)
{
    // (s_nop: the descriptor / offset SGPRs may have been written by the instruction in front -- a VMEM instruction reading an SGPR
    // needs 5 wait states behind its writer, and the compiler's hazard recognizer does not look into inline assembly: without them
    // the first request behind `prepare` went out with the OLD descriptor)
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(srd), "s"(soff) : "memory");
}
template <int N> __device__ __forceinline__ void vm_wait_p(f32x4 (&p)[NPD])
{
    asm volatile("s_waitcnt vmcnt(%6)" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]) : "n"(N) : "memory");
}
// the four fragments of one group (uh ct0 | uh ct1 | ul ct0 | ul ct1: 4 KB contiguous in the packed weights and in the ring): the
// instruction offset moves source and destination alike
__device__ __forceinline__ void dma_group(const unsigned int m0v, const unsigned int voff, const void *sbase)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072"
                 : : "s"(m0v), "v"(voff), "s"(sbase) : "memory");
}
// one of them (the step deals them out one per slot: the CU's address path takes 16 clocks per 1-KB instruction, and a wave whose
// request finds it busy stands still -- four waves x four requests in one slot are 256 clocks in which nothing else issues)
// -- M0 is set by the first of a group and stands for the other three: nothing the compiler emits in between writes it (no GWS, no
// s_movrel, no other DMA; tests/test_isa.py looks)
template <int OFF> __device__ __forceinline__ void dma_frag(const unsigned int m0v, const unsigned int voff, const void *sbase)
{
    if constexpr (OFF == 0)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(m0v), "v"(voff), "s"(sbase) : "memory");
    else
        asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" : : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

// The transform of one tile half as 128 micro-operations in dependency order (a lane: 8 channels = two quarters g0 of a chunk):
//   [0, 8) reads of quarter 0 (rows ra / rb x 4 columns)   [8, 24) stage 1 of quarter 0: t[c] = a[c] + sr b[c]
//   [24, 32) reads of quarter 1                              [32, 48) stage 2 of quarter 0: the four columns of (B^T d) B
//   [48, 72) splits of quarter 0 (8 pairs x cvt, mixlo, mixhi)   [72, 88) stage 1, [88, 104) stage 2, [104, 128) splits of quarter 1
// so that the step can deal them out between its MFMAs (a wave issues one instruction per ~4.8 clocks: 5-7 beside every MFMA).
// Plain fp32 instructions only (see the header).  Row i of B^T d = d[ra] + sr d[rb]:  i 0: d0 - d2,  1: d1 + d2,  2: d2 - d1,  3: d1 - d3;
// its four columns:  j 0: t0 - t2,  1: t1 + t2,  2: t2 - t1,  3: t1 - t3.
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
// micro-operations of the transform dealt to each of the 24 MFMA slots of a half step.  What else a slot carries:
//   first half : slots 6 j: the counted wait for fragment group j; 6 j, 6 j + 1: the group's fragment reads (ul, then uh, for both cout
//                halves); slots 12..17: one patch DMA each
//   second half: slot 0: the barrier; 6 j, 6 j + 1: fragment reads; 6 j + 2 .. 6 j + 5: one DMA each of the group's next fragments
constexpr int QA[24] = {4, 4, 6, 6, 6, 6, 5, 6, 5, 6, 6, 6, 4, 5, 4, 5, 5, 5, 5, 6, 5, 6, 6, 6};
constexpr int QB[24] = {4, 4, 5, 6, 6, 5, 5, 6, 5, 6, 6, 5, 5, 6, 5, 6, 6, 5, 5, 6, 5, 6, 5, 5};
constexpr int qsum(const int (&q)[24], const int n)
{
    int t = 0;
    for (int i = 0; i < n; ++i) t += q[i];
    return t;
}
static_assert(qsum(QA, 24) == 128 && qsum(QB, 24) == 128, "conv_wino4: every micro-operation of a transform is dealt exactly once");

#ifdef WINO_STAMP
// instrumentation build (tools/conv_wino4_stamp.py): shader-clock totals per phase of a wave's life, summed over all waves
//   first half: 0 up to and including the wait for fragment group 0 | 1 slots 0-5 | 2 slots 6-11 | 3 slots 12-17 (patch DMAs) | 4 slots 18-23 ||
//   second half: 5 first MFMA + barrier | 6 slots 0-11 | 7 slots 12-23 | 8 cursors || 9 output exchange + epilogue | 10 tile bookkeeping |
//   11 waves
constexpr int NSTAMP = 12;
__device__ unsigned long long g_wino4_stamp[1024][NSTAMP];
#define W4STAMP(i)                                                     \
    {                                                                 \
        const unsigned long long t_now = __builtin_readcyclecounter(); \
        st_acc[i] += t_now - t_last;                                  \
        t_last = t_now;                                               \
    }
#else
#define W4STAMP(i) {}   // (a statement: `if constexpr (...) W4STAMP(n)` must not swallow the line that follows)
#endif

// timing experiments (results wrong): -DW4_ABL=1 no fragment DMAs | 2 no patch DMAs | 3 no transform | 4 no MFMAs |
// 5 no output exchange / epilogue | 6 no barrier in the step | 7 = 1 + 2 | 8 no fragment reads
#ifndef W4_ABL
#define W4_ABL 0
#endif

template <int RES, bool NT, bool POOL, bool SCALED>   // RES 0: no tensor added in the epilogue, 1: residual (after the activation), 2: pre (before it);
                                         // NT: the output is larger than the last-level cache and is streamed (non-temporal stores /
                                         // loads of the added tensor); POOL: epilogue 1 (MaxPool2d(2,2); RES 0 only); SCALED: the
                                         // input is multiplied by a power of two on its way into LDS (A.in_amax, conv_wino.hip)
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
    const int tiles_x = W / (2 * TT), tiles_y = H / (2 * TT);
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

    // ---- stage 1: the raw fp32 patch of a chunk, global -> registers -> LDS.  Thread tid < 216 owns quarter tid & 3 (4 channels) of
    // patch pixel (py, px) = divmod(tid >> 2, 18) and of the pixels 3, 6, ... 15 rows below it: four neighbouring lanes read the 64
    // contiguous bytes of a pixel -- the address path takes a 64-byte request per clock; with one pixel per lane (what a DMA
    // straight into the transform's layout [quarter][pixel] needs) every lane is a request of its own, 4 x the time: 1.5 k clocks
    // per chunk, measured.  A lane outside the image or past a ragged chunk's channels carries an offset beyond the buffer descriptor
    // and receives zeros.  The registers are in flight from the request (first half, slots 12..17) to the step's-end wait of the
    // next step's fragment group 0; tools/asm_inflight_check_wino4.py checks that nothing touches them in between.
    f32x4 pf[NPD];
    const int p_q = tid & 3, p_py = (tid >> 2) / PP, p_px = (tid >> 2) - p_py * PP;
    const bool p_have = tid < PROWS * PP * 4;
    const unsigned int praw0 = p_have ? (unsigned int)(SLOT_OFF + (p_q * RAW_Q + p_py * RAW_RS + (p_px & 1) * RAW_CP + (p_px >> 1)) * 16) : (unsigned int)(SINK_OFF + tid * 16);
    const unsigned int praw_step = p_have ? (unsigned int)(PROWS * RAW_RS * 16) : 0u, praw_slot = p_have ? (unsigned int)SLOT_BYTES : 0u;
    // The request stream of the patch runs two chunks ahead of the multiply, across tile boundaries, with its own cursor (rq_tile,
    // rq_ch); `prepare` sets up the request of the cursor's chunk a step before it is issued and moves the cursor on:
    //   rq_srd   buffer descriptor of the source image (x1 or x2 of the tile's sample: H W ld 4 bytes)
    //   rq_soff  byte offset of the chunk inside a pixel's channels
    //   rq_vo[k] byte offset of piece k's pixel and quarter -- recomputed only at the first chunk of a source and at a ragged last one
    Tile nxt = {0, 0, 0, 0};   // the request cursor's tile, decoded
    int rq_tile = tile, rq_ch = 0, rq_run = 0;
    int t_n = 0, t_y0 = 0, t_x0 = 0;   // the request cursor's tile (t_n = -1: past the block's last tile -- every lane out of bounds)
    __amdgpu_buffer_rsrc_t rq_srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A.x1), 0, 0, 0x00020000);
    unsigned int rq_soff = 0, rq_vo[NPD];
#pragma unroll
    for (int k = 0; k < NPD; ++k) rq_vo[k] = OOB;
    // (the cursor variables are updated OUTSIDE the branch, by selects: similar load-add-store sequences on different variables in the
    // two arms get merged by the compiler into one on a selected POINTER, and the variables then live in scratch memory)
    auto prepare = [&]() {
        const bool edge = rq_run == 0;   // (uniform) the first chunk of a tile or of its second source (new offsets / descriptor), or a
                                         // ragged last chunk (some quarters past the source's channels); else: 64 bytes further per pixel
        const bool wrap = edge && rq_ch == n_ch;
        rq_tile += wrap ? tstep : 0;
        const int ch = wrap ? 0 : rq_ch;
        rq_ch = ch + 1;
        int run = rq_run - 1;
        unsigned int soff = rq_soff + KC * 4;
        if (edge) {
            if (ch == 0) {   // a new tile: decoded ONCE, here (three integer divisions) -- the fragment cursor and the tile loop take
                             // its coordinates from `nxt` when they get there (the request stream is the furthest ahead: two chunks, and
                             // a tile has at least three)
                nxt = decode(rq_tile < band1 ? rq_tile : band0);
                t_n = rq_tile < band1 ? nxt.n : -1, t_y0 = nxt.y0, t_x0 = nxt.x0;
            }
            const bool first = ch < A.n_ch1;
            const int cl = first ? ch : ch - A.n_ch1, Cs = first ? A.C1 : A.C2;   // chunk within its source
            soff = (unsigned int)(cl * KC * 4);
            const int ld = first ? A.ld1 : A.ld2, nn = t_n < 0 ? 0 : t_n;
            const float *const img = first ? A.x1 + (size_t)(nn % A.N1) * H * W * A.ld1 : A.x2 + (size_t)(nn % A.N2) * H * W * A.ld2;
            const unsigned int bytes = __builtin_amdgcn_readfirstlane((unsigned int)((size_t)H * W * ld * 4));   // (wino_launch: < OOB)
            rq_srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(scalar_ptr(img)), 0, bytes, 0x00020000);
            // (branch-free: the conditions are combined bitwise -- as short-circuit tests they were 13 divergent branches per edge --;
            // the six pieces of a thread share their column, quarter and base offset and lie PROWS rows apart)
            const int gy0 = t_y0 + p_py - 1, gx = t_x0 + p_px - 1;
            const bool cok = p_have & (t_n >= 0) & ((unsigned int)gx < (unsigned int)W) & (cl * KC + 4 * p_q < Cs);
            const unsigned int base = ((unsigned int)(gy0 * W + gx) * (unsigned int)ld + 4u * p_q) * 4u;   // (used only where gy >= 0)
            const unsigned int step3 = (unsigned int)(PROWS * W * ld) * 4u;
#pragma unroll
            for (int k = 0; k < NPD; ++k)
                rq_vo[k] = (cok & ((unsigned int)(gy0 + k * PROWS) < (unsigned int)H)) ? base + (unsigned int)k * step3 : OOB;
            const int full = Cs / KC;                      // whole chunks of the source: the chunks cl + 1 .. full - 1 share this one's offsets
            run = cl < full ? full - 1 - cl : 0;
        }
        rq_run = __builtin_amdgcn_readfirstlane(run);
        rq_soff = __builtin_amdgcn_readfirstlane(soff);
    };
    auto request_piece = [&](const int k) { bload16(pf[k], rq_vo[k], rq_srd, __builtin_amdgcn_readfirstlane(rq_soff)); };
    float in_s = 1.f, oscale_in = A.out_scale;
    if constexpr (SCALED) {   // (conv_wino.hip: max |x| into [2^11, 2^12))
        const float am = *A.in_amax;
        if (am > 1.0e-30f && am < 3.0e38f) {
            int e;
            (void)frexpf(am, &e);
            in_s = ldexpf(1.f, 12 - e);
            oscale_in = A.out_scale * ldexpf(1.f, e - 12);
        }
        in_s = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(in_s)));
    }
    auto store_piece = [&](const int slot, const int k) {   // (the caller's counted wait has passed pf through)
        f32x4 v = pf[k];
        if constexpr (SCALED) v[0] *= in_s, v[1] *= in_s, v[2] *= in_s, v[3] *= in_s;
        *reinterpret_cast<f32x4 *>(smem + praw0 + slot * praw_slot + k * praw_step) = v;
    };

    // ---- stage 2: the wave's transform (micro-operations `top<K>` above)
    const int t_ra = wi == 0 ? 0 : (wi == 2 ? 2 : 1), t_rb = wi == 0 ? 2 : (wi == 1 ? 2 : (wi == 2 ? 1 : 3));
    const float t_sr = __uint_as_float(__builtin_amdgcn_readfirstlane(wi == 1 ? 0x3f800000u : 0xbf800000u));
    // lane = (tx = l31 & 7, ty low bits = l31 >> 3, k half): quarter 2 kh (+ g0) of pixel (2 ty, 2 tx) of tile half 0
    const unsigned int t_base = (unsigned int)(SLOT_OFF + ((2 * kh) * RAW_Q + (2 * (l31 >> 3)) * RAW_RS + (l31 & 7)) * 16);
    const unsigned int t_a = t_base + t_ra * RAW_RS * 16, t_b = t_base + t_rb * RAW_RS * 16;

    // ---- stage 3: the wave's 16 weight fragments of a chunk ((j, cout half) x (uh, ul)) in its quarter of the ring; a group is
    // fetched again (next chunk) right behind its last MFMA of the chunk (tile half 1) and is read a whole multiply later.
    const unsigned int u_voff = (unsigned int)(l31 * KC + kh * 8) * 2u;   // the lane's 16 bytes of a fragment: cout l31, cin 8 kh ..
    const unsigned short *const u_wave = A.wp + (size_t)(wi * 4) * 2 * NB * KC;
    const unsigned int ring_w = (unsigned int)(RING_OFF + wi * 16384);
    const unsigned char *const frag_l = smem + ring_w + lane * 16;
    int uq_tile = tile, uq_left = 0;   // chunks of the fragment cursor's tile still to come
    const unsigned char *u_s = nullptr;
    auto frag_cursor = [&]() {   // scalar base of the cursor's chunk of the fragment stream, then the cursor moves on
        const bool edge = uq_left == 0;   // (uniform) a new tile: its cout block's fragments; else the next chunk, 64 KB further
        const unsigned char *nb = u_s + WCH_HALVES * 2;
        if (edge) {
            // (the request cursor has decoded this tile already; past the block's last tile: any valid fragments, never used)
            nb = reinterpret_cast<const unsigned char *>(scalar_ptr(u_wave + (size_t)nxt.cb * n_ch * WCH_HALVES));
        }
        u_s = nb;
        uq_tile += edge ? tstep : 0;
        uq_left = edge ? n_ch - 1 : uq_left - 1;
    };
    auto frag_dma = [&](const int j) { dma_group(ring_w + j * 4096, u_voff, u_s + j * 4096); };   // xi (i, j): 4 KB per group in the packed weights
    auto frag_dma1 = [&](const int j, auto pc) { dma_frag<decltype(pc)::value * 1024>(ring_w + j * 4096, u_voff, u_s + j * 4096); };
    auto frag_read = [&](const int j, const int ct, const int plane) {   // plane 0: uh, 1: ul
        if constexpr (W4_ABL == 8) return u32x4{(unsigned int)(j + ct), (unsigned int)plane, 0x3c003c00u, 0x3c003c00u};
        else return *reinterpret_cast<const u32x4 *>(frag_l + j * 4096 + plane * 2048 + ct * 1024);
    };

    // ---- the chunk stream.  Chunk counter g (per block, across tiles); its raw patch lives in slot g mod 3.  A step is two half steps
    // of 24 slots, a slot = one MFMA + its share of everything else, pinned in this order (sched_barrier: left to itself the compiler
    // emits a whole transform in front of the MFMAs that wait for fragments, and the wave -- alone on its SIMD -- idles the matrix pipe):
    //   first half : MFMAs of tile half 0 (operands from the previous step: slot 6 j + k = product k >> 1 of cout half k & 1 of group j)
    //                beside the transform of tile half 1; a counted wait in front of each fragment group (the first one also hands
    //                over the patch registers of chunk g + 1); slots 6..11 store them into the other slot (chunk g - 1's, whose last
    //                reader passed the previous step's barrier); slots 12..17 request the patch of chunk g + 2
    //   second half: MFMAs of tile half 1 beside the transform of tile half 0 of chunk g + 1; the barrier sits behind the first MFMA
    //                (every wave has stored its share of chunk g + 1's patch); every
    //                fragment group is requested again (chunk g + 1), one DMA per slot behind the slots that read it (a tile's last step requests the NEXT tile's first chunk: it lands under the epilogue)
    //   then       : the cursors move on (the only branches of a step)
    // In-order return queue at the start of a step:  [patch g + 1: 6] [fragments g: j0 x4, j1 x4, j2 x4, j3 x4]
    //   waits (each in front of the ul read of its group, two slots before the group's first MFMA):
    //   second half slot 22 of the previous step: group 0, 10 younger may fly | first half slot 4: group 1, 8 | slot 10: group 2, 4 |
    //   slot 16: group 3, 4 (the patch requests of slots 12..15) | a tile's first step waits for group 0 at its start (12)
    int sl = 0;   // slot of the current chunk (chunk counter mod 2)
    prepare();                       // chunk 0
    Tile cur = nxt;
#pragma unroll
    for (int k = 0; k < NPD; ++k) request_piece(k);
    prepare();                       // chunk 1: in flight into the loop, OLDER than the fragments of chunk 0
    frag_cursor();
    vm_wait_p<0>(pf);
#pragma unroll
    for (int k = 0; k < NPD; ++k) store_piece(0, k);
#pragma unroll
    for (int k = 0; k < NPD; ++k) request_piece(k);
#pragma unroll
    for (int j = 0; j < 4; ++j) frag_dma(j);
    prepare();                       // chunk 2: requested by the first step
    frag_cursor();                   // fragments of chunk 1: requested by the first step
    __syncthreads();
    vm_wait_p<12>(pf);               // fragment group 0 of chunk 0 has landed (and with it the older patch pieces of chunk 1)
    // (a scalar: as a vector register it is one more value to keep across the chunk loop; no activation = slope 1: v * 1 is v, the
    // epilogue has no branch)
    const float slope = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(!A.act ? 1.0f : A.slope_ptr ? *A.slope_ptr : A.slope)));
    u32x4 a0h[4], a0l[4];            // operands of tile half 0 of the current chunk
    {
        TState ts;
        const unsigned char *const pa = smem + t_a, *const pb = smem + t_b;
        sfor<0, 128>([&](auto kc) { top<decltype(kc)::value>(ts, pa, pb, t_sr, a0h, a0l); });
    }
    float nonfin = 0.f;              // fp16 range guard: becomes NaN when an output is not finite (an operand beyond the fp16 range)
    float oamx = 0.f;                // max |out| of what this lane stores (A.stat_amax: the next Winograd layer's input scale)
    // The fragment registers of the group in work: `ul` serves a group's first two MFMAs (ul . vh for the two cout halves), `uh` the
    // other four.  Each is requested from the ring while the OTHER one is in use -- ul of group j + 1 at slot 6 j + 4, uh of group j at
    // slot 6 j -- so that no MFMA waits for an LDS read issued in its own slot (8 exposed LDS latencies per step in the first version).
    // ul therefore lives across half steps and steps.
    u32x4 ul[2], uh[2];
    ul[0] = frag_read(0, 0, 1), ul[1] = frag_read(0, 1, 1);   // (the block's first step; every later one finds them read by its predecessor)

    auto step = [&](auto first_c) {
        constexpr bool first = decltype(first_c)::value;   // a tile's first step: the accumulators start from zero
        asm volatile("; W4MARK step_begin");
        u32x4 b1h[4], b1l[4];
        const int sl1 = sl ^ 1;   // slot of chunk g + 1
        {
            TState ts;
            const unsigned char *const pa = smem + t_a + sl * SLOT_BYTES + 8 * RAW_RS * 16;
            const unsigned char *const pb = smem + t_b + sl * SLOT_BYTES + 8 * RAW_RS * 16;
            sfor<0, 24>([&](auto mc) {
                constexpr int m = decltype(mc)::value, j = m / 6, k = m % 6, ct = k & 1, prod = k >> 1;
                if constexpr (m == 0) W4STAMP(0)
                if constexpr (m == 6) W4STAMP(1)
                if constexpr (m == 12) W4STAMP(2)
                if constexpr (m == 18) W4STAMP(3)
                if constexpr (k == 0) uh[0] = frag_read(j, 0, 0), uh[1] = frag_read(j, 1, 0);
                if constexpr (W4_ABL != 4) {
                    f32x16 z;
#pragma unroll
                    for (int e = 0; e < 16; ++e) z[e] = 0.f;
                    if constexpr (prod == 0) acc[j][0][ct] = mma16(ul[ct], a0h[j], first ? z : acc[j][0][ct]);
                    else if constexpr (prod == 1) acc[j][0][ct] = mma16(uh[ct], a0l[j], acc[j][0][ct]);
                    else acc[j][0][ct] = mma16(uh[ct], a0h[j], acc[j][0][ct]);
                } else {
                    asm volatile("" : : "v"(prod == 0 ? ul[ct] : uh[ct]), "v"(prod == 1 ? a0l[j] : a0h[j]));
                }
                if constexpr (W4_ABL != 3) sfor<qsum(QA, m), qsum(QA, m + 1)>([&](auto kc) { top<decltype(kc)::value>(ts, pa, pb, t_sr, b1h, b1l); });
                if constexpr (W4_ABL != 2 && W4_ABL != 7 && m >= 6 && m < 12) store_piece(sl1, m - 6);
                if constexpr (k == 4) {   // the next group's ul (behind the MFMAs that used this group's: slots 6 j, 6 j + 1)
                    if constexpr (j == 0) vm_wait<8>();
                    if constexpr (j == 1) vm_wait<4>();
                    if constexpr (j == 2) vm_wait<4>();   // (fragment group 3; the four patch requests of slots 12..15 are younger)
                    constexpr int jn = j == 3 ? 0 : j + 1;   // (slot 22: group 0 again, for the second half step)
                    ul[0] = frag_read(jn, 0, 1), ul[1] = frag_read(jn, 1, 1);
                }
                if constexpr (W4_ABL != 2 && W4_ABL != 7 && m >= 12 && m < 18) request_piece(m - 12);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        W4STAMP(4)
        {
            TState ts;
            const unsigned char *const pa = smem + t_a + sl1 * SLOT_BYTES;
            const unsigned char *const pb = smem + t_b + sl1 * SLOT_BYTES;
            sfor<0, 24>([&](auto mc) {
                constexpr int m = decltype(mc)::value, j = m / 6, k = m % 6, ct = k & 1, prod = k >> 1;
                if constexpr (k == 0) uh[0] = frag_read(j, 0, 0), uh[1] = frag_read(j, 1, 0);
                if constexpr (W4_ABL != 4) {
                    f32x16 z;
#pragma unroll
                    for (int e = 0; e < 16; ++e) z[e] = 0.f;
                    if constexpr (prod == 0) acc[j][1][ct] = mma16(ul[ct], b1h[j], first ? z : acc[j][1][ct]);
                    else if constexpr (prod == 1) acc[j][1][ct] = mma16(uh[ct], b1l[j], acc[j][1][ct]);
                    else acc[j][1][ct] = mma16(uh[ct], b1h[j], acc[j][1][ct]);
                } else {
                    asm volatile("" : : "v"(prod == 0 ? ul[ct] : uh[ct]), "v"(prod == 1 ? b1l[j] : b1h[j]));
                }
                if constexpr (m == 0) {
                    if constexpr (W4_ABL != 6) __syncthreads();
                    W4STAMP(5)
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (m == 12) W4STAMP(6)
                if constexpr (W4_ABL != 3) sfor<qsum(QB, m), qsum(QB, m + 1)>([&](auto kc) { top<decltype(kc)::value>(ts, pa, pb, t_sr, a0h, a0l); });
                // (the group's fragments were read into registers by slots 6 j, 6 j + 1 and consumed by the MFMAs up to this one: the DMA
                // may overwrite them in LDS from here on)
                if constexpr (k == 4 && j < 3) ul[0] = frag_read(j + 1, 0, 1), ul[1] = frag_read(j + 1, 1, 1);
                if constexpr (m == 22) {
                    // group 0 of the NEXT chunk (requested in slots 2..5: 4 + 4 + 2 younger requests may fly) for the next step's first
                    // MFMAs; the patch registers of chunk g + 1, older than every fragment request of this half step, are handed over here
                    vm_wait_p<10>(pf);
                    ul[0] = frag_read(0, 0, 1), ul[1] = frag_read(0, 1, 1);
                }
                if constexpr (W4_ABL != 1 && W4_ABL != 7 && k >= 2) frag_dma1(j, std::integral_constant<int, k - 2>{});
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        // (no patch register is in flight here: the pieces requested in this step were handed over in slot 22 -- across a tile
        // boundary the compiler copies them between the step instances' registers, which it may only do with values that have arrived)
        asm volatile("; W4MARK step_cursors");
        W4STAMP(7)
        prepare();
        frag_cursor();
        sl = sl1;
        asm volatile("; W4MARK step_end");
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
        // The exchange buffer is a region of its own (the ring is receiving the next tile's first fragments meanwhile); its readers of
        // the previous tile's last pass are n_ch step barriers behind.
        asm volatile("; W4MARK fast_begin");
        const int cb = cur.cb, n = cur.n, y0 = cur.y0, x0 = cur.x0;
        float *const xb = reinterpret_cast<float *>(smem + X_OFF);
        const int Cout = A.Cout;
        const float oscale = oscale_in;
        // (per-thread addresses of the epilogue are formed here, per tile, from an opaque copy of the thread index instead of being kept
        // across the chunk loop)
        int tid_e = tid;
        asm volatile("" : "+v"(tid_e));
        const int c4 = (tid_e & 7) * 4, T0 = tid_e >> 3;
        float *const xrow0 = xb + ((size_t)wi * NTILE + (tid_e & 31)) * X_LD + 4 * ((tid_e >> 5) & 1);
        // the column fold of one pass into this wave's row of the exchange buffer.  The accumulators are read where they are used, by
        // hand: left to the compiler, all 256 of them are copied into vector registers at the top of the epilogue (and everything else
        // that lives across it is spilled to make the room)
        auto rd = [](const float &a) {
            float v;
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
            return v;
        };
        auto publish = [&](const int hc, const int b) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float *const xrow = xrow0 + t * 32 * X_LD;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    float z[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int e = 4 * qd + r;
                        if (b == 0) {
                            const float m0 = rd(acc[0][t][hc][e]), m1 = rd(acc[1][t][hc][e]), m2 = rd(acc[2][t][hc][e]);
                            z[r] = m0 + m1 + m2;
                        } else {
                            const float m1 = rd(acc[1][t][hc][e]), m2 = rd(acc[2][t][hc][e]), m3 = rd(acc[3][t][hc][e]);
                            z[r] = m1 + (-m2 - m3);   // (conv_wino_kernel's association: same bits)
                        }
                    }
                    *reinterpret_cast<float4 *>(xrow + 8 * qd) = make_float4(z[0], z[1], z[2], z[3]);
                }
            }
        };
        auto gather = [&](const int T, float4 (&y)[2]) {   // the row fold for tile T, couts c4 .. c4 + 3 of the pass
            float4 z[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) z[i] = *reinterpret_cast<const float4 *>(xb + ((size_t)i * NTILE + T) * X_LD + c4);
            y[0] = make_float4(z[0].x + z[1].x + z[2].x, z[0].y + z[1].y + z[2].y, z[0].z + z[1].z + z[2].z, z[0].w + z[1].w + z[2].w);
            y[1] = make_float4(z[1].x - z[2].x - z[3].x, z[1].y - z[2].y - z[3].y, z[1].z - z[2].z - z[3].z, z[1].w - z[2].w - z[3].w);
        };
        // Every address is a scalar base (the tile's first pixel in `out` / the added tensor, a buffer descriptor without bounds) + one
        // per-thread offset that does not depend on the tile + a scalar offset per (pass, row, tile half): no 64-bit vector arithmetic.
        {
            constexpr bool pooled = POOL;
            const int Ho = pooled ? H >> 1 : H, Wo = pooled ? W >> 1 : W, yo = pooled ? y0 >> 1 : y0, xo = pooled ? x0 >> 1 : x0;
            const int ld_r = RES == 2 ? Cout : A.ld_res;   // (pre is dense [pre_N][H][W][Cout], batch-broadcast)
            const __amdgpu_buffer_rsrc_t srd_o = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<void *>(scalar_ptr(A.out + ((size_t)(n * Ho + yo) * Wo + xo) * A.ld_out + cb * NB)), 0, 0xffffffffu, 0x00020000);
            const float *const rbase = RES == 1 ? A.residual + ((size_t)(n * H + y0) * W + x0) * ld_r + cb * NB
                                     : RES == 2 ? A.pre + ((size_t)((n % A.pre_N) * H + y0) * W + x0) * ld_r + cb * NB : A.out;
            const __amdgpu_buffer_rsrc_t srd_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(scalar_ptr(rbase)), 0, 0xffffffffu, 0x00020000);
            // item 0: tile T0 (ty = T0 >> 3 < 4); item 1: tile T0 + 32 = four tile rows further down
            const int pstep = pooled ? 1 : 2;   // output pixels per tile and direction
            const unsigned int vo_o = (unsigned int)(((pstep * (T0 >> 3)) * Wo + pstep * (T0 & 7)) * A.ld_out + c4) * 4u;
            const unsigned int vo_r = (unsigned int)(((2 * (T0 >> 3)) * W + 2 * (T0 & 7)) * ld_r + c4) * 4u;
            const unsigned int row_o = (unsigned int)(Wo * A.ld_out) * 4u, row_r = (unsigned int)(W * ld_r) * 4u;
            constexpr int aux = NT ? 2 : 0;   // (nt: outputs beyond the last-level cache are streamed, conv_common.h)
            if constexpr (W4_ABL == 5) {   // (the accumulators stay alive)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int t = 0; t < 4; ++t) nonfin = __builtin_fmaf(rd(acc[j][t >> 1][t & 1][0]), 0.f, nonfin);
            }
#pragma unroll
            for (int hc = 0; hc < (W4_ABL == 5 ? 0 : 2); ++hc) {
                const float4 bv = *reinterpret_cast<const float4 *>(smem + BIAS_OFF + (cb * NB + hc * 32 + c4) * 4);
                float4 pool[2];   // epilogue 1: maximum of the tile's column 0 pixels, kept over the b = 1 pass
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    if (hc | b) __syncthreads();   // the previous pass has been read
                    f32x4 rq[2][2];
                    if constexpr (RES != 0) {   // the added tensor's pixels of the pass, requested at its START: their latency lies under the
                                                // exchange writes and the barrier
#pragma unroll
                        for (int it = 0; it < 2; ++it)
#pragma unroll
                            for (int a = 0; a < 2; ++a) {
                                const unsigned int so = (unsigned int)(8 * it + a) * row_r + (unsigned int)(b * ld_r + hc * 32) * 4u;
                                rq[it][a] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_r, vo_r, so, aux));
                            }
                    }
                    publish(hc, b);
                    __syncthreads();
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        __builtin_amdgcn_sched_barrier(0);
                        float4 y[2];
                        gather(T0 + 32 * it, y);
                        if constexpr (pooled) {   // MaxPool2d(2,2) of act(conv + bias) = act(max4 + bias): both monotone
                            // the range guard BEFORE the maximum: fmaxf drops a NaN, and a transform value that enters one output of the
                            // 2 x 2 tile only (V[0][0] -> Y[0][0]) makes just that pixel non-finite (inf - inf = NaN keeps the flag up)
                            nonfin = __builtin_fmaf(y[0].x + y[1].x, 0.f, nonfin);
                            float4 m = make_float4(fmaxf(y[0].x, y[1].x), fmaxf(y[0].y, y[1].y), fmaxf(y[0].z, y[1].z), fmaxf(y[0].w, y[1].w));
                            if (b == 0) {
                                pool[it] = m;
                            } else {
                                m = make_float4(fmaxf(m.x, pool[it].x), fmaxf(m.y, pool[it].y), fmaxf(m.z, pool[it].z), fmaxf(m.w, pool[it].w));
                                float4 v = make_float4(m.x * oscale + bv.x, m.y * oscale + bv.y, m.z * oscale + bv.z, m.w * oscale + bv.w);
                                nonfin = __builtin_fmaf(v.x, 0.f, nonfin);   // (one cout of the pixel: see the plain epilogue)
                                v.x = v.x > 0.f ? v.x : v.x * slope, v.y = v.y > 0.f ? v.y : v.y * slope;
                                v.z = v.z > 0.f ? v.z : v.z * slope, v.w = v.w > 0.f ? v.w : v.w * slope;
                                const unsigned int so = (unsigned int)(4 * it) * row_o + (unsigned int)(hc * 32) * 4u;
                                oamx = fmaxf(fmaxf(fmaxf(oamx, fabsf(v.x)), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v.x, v.y, v.z, v.w}), srd_o, vo_o, so, aux);
                            }
                        } else {
#pragma unroll
                            for (int a = 0; a < 2; ++a) {
                                float4 v = y[a];
                                v.x = v.x * oscale + bv.x, v.y = v.y * oscale + bv.y, v.z = v.z * oscale + bv.z, v.w = v.w * oscale + bv.w;
                                // the range guard: (inf, NaN) * 0 = NaN.  One cout of the pixel is enough: a transform value beyond the
                                // fp16 range is an inf operand of EVERY cout's products (inf * u = +-inf, inf * 0 = NaN)
                                nonfin = __builtin_fmaf(v.x, 0.f, nonfin);
                                if constexpr (RES == 2) v.x += rq[it][a][0], v.y += rq[it][a][1], v.z += rq[it][a][2], v.w += rq[it][a][3];
                                v.x = v.x > 0.f ? v.x : v.x * slope, v.y = v.y > 0.f ? v.y : v.y * slope;
                                v.z = v.z > 0.f ? v.z : v.z * slope, v.w = v.w > 0.f ? v.w : v.w * slope;
                                if constexpr (RES == 1) v.x += rq[it][a][0], v.y += rq[it][a][1], v.z += rq[it][a][2], v.w += rq[it][a][3];
                                const unsigned int so = (unsigned int)(8 * it + a) * row_o + (unsigned int)(b * A.ld_out + hc * 32) * 4u;
                                oamx = fmaxf(fmaxf(fmaxf(oamx, fabsf(v.x)), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v.x, v.y, v.z, v.w}), srd_o, vo_o, so, aux);
                            }
                        }
                    }
                }
            }
        }
        asm volatile("; W4MARK fast_end");
        W4STAMP(9)
        if (!more) break;
        tile = tile_n;
        cur = nxt;   // (decoded by the request stream when it reached this tile's first chunk)
    }
    vm_wait_p<0>(pf);   // (DMAs past the last chunk land in LDS: nothing of the block may leave before them)
#pragma unroll
    for (int k = 0; k < NPD; ++k) asm volatile("" ::"v"(pf[k]) : "memory");
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(a0h[j]), "v"(a0l[j]) : "memory");
    // the fp16 range guard: an activation so large that a transform value leaves the fp16 range makes that value +-inf and every output it
    // enters inf or NaN (the eight-wave kernel compares the raw activations with 16000 instead: here they never pass through registers)
    if (A.range_flag && !(nonfin == nonfin)) atomicOr(A.range_flag, 1);
    if (A.stat_amax) {   // (a non-finite output raised the flag above: the batch is re-run, this value is not used)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) oamx = fmaxf(oamx, __shfl_xor(oamx, o, 64));
        if (lane == 0 && oamx > 0.f && oamx < 3.0e38f && __float_as_uint(oamx) > __builtin_nontemporal_load(A.stat_amax)) atomicMax(A.stat_amax, __float_as_uint(oamx));
    }
#ifdef WINO_STAMP
    W4STAMP(10)
    st_acc[11] = 1;
    if (lane == 0)
        for (int i = 0; i < NSTAMP; ++i) atomicAdd(&g_wino4_stamp[(blockIdx.x * 4 + wi) & 1023][i], st_acc[i]);
#endif
}

}  // namespace

namespace mrefsr_conv {

// What the four-wave kernel takes: whole tiles, whole cout blocks, 16-byte rows, the plain or the max-pool epilogue; everything else
// stays on the eight-wave kernel.
bool wino4_serves(const ConvArgs &a)
{
    if ((a.H & 15) || (a.W & 15) || (a.Cout & 63) || (a.ld_out & 3) || (a.ld1 & 3) || (a.C1 & 3) || (a.x2 && ((a.ld2 & 3) || (a.C2 & 3)))) return false;
    if (a.epilogue == 1) return !a.residual && !a.pre;
    if (a.epilogue != 0) return false;
    if (a.residual && a.pre) return false;
    if (a.residual && (a.ld_res & 3)) return false;
    return true;
}

// ConvArgs as conv_nhwc.hip's conv_entry fills them (terms 16 semantics: out_scale = 1 / wscale).
// Called by wino_launch (conv_wino.hip) after its descriptor checks and wino4_serves().
int wino4_launch(const ConvArgs &b, int blocks_cu, hipStream_t stream)
{
    const int res = b.residual ? 1 : b.pre ? 2 : 0;
    const bool pool = b.epilogue == 1, nt = b.stream_out != 0;
    const bool sc = b.in_amax != nullptr;
    using K = void (*)(const ConvArgs);
    static const K table[4][2][2] = {   // [RES 0 / 1 / 2 / pooled][NT][SCALED]
        {{conv_wino4_kernel<0, false, false, false>, conv_wino4_kernel<0, false, false, true>},
         {conv_wino4_kernel<0, true, false, false>, conv_wino4_kernel<0, true, false, true>}},
        {{conv_wino4_kernel<1, false, false, false>, conv_wino4_kernel<1, false, false, true>},
         {conv_wino4_kernel<1, true, false, false>, conv_wino4_kernel<1, true, false, true>}},
        {{conv_wino4_kernel<2, false, false, false>, conv_wino4_kernel<2, false, false, true>},
         {conv_wino4_kernel<2, true, false, false>, conv_wino4_kernel<2, true, false, true>}},
        {{conv_wino4_kernel<0, false, true, false>, conv_wino4_kernel<0, false, true, true>},
         {conv_wino4_kernel<0, true, true, false>, conv_wino4_kernel<0, true, true, true>}}};
    const int ki = pool ? 3 : res;
    const K kern = table[ki][nt ? 1 : 0][sc ? 1 : 0];
    static unsigned long long attr[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (mrefsr::first_use_on_device(attr[(ki * 2 + (nt ? 1 : 0)) * 2 + (sc ? 1 : 0)]))
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipLaunchKernelGGL(kern, dim3(blocks_cu), dim3(256), LDS_BYTES, stream, b);
    return mrefsr::check_launch("conv_wino4");
}

}  // namespace mrefsr_conv

#ifdef WINO_STAMP
// read-and-reset of the phase clocks (instrumentation builds only)
MREFSR_EXPORT int mrefsr_dbg_wino4_stamps(unsigned long long *out12)
{
    static unsigned long long h[1024][NSTAMP];
    if (hipDeviceSynchronize() != hipSuccess) return mrefsr::fail(MREFSR_E_LAUNCH, "wino4_stamps: sync failed");
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_wino4_stamp), sizeof(h)) != hipSuccess) return mrefsr::fail(MREFSR_E_LAUNCH, "wino4_stamps: read failed");
    for (int i = 0; i < NSTAMP; ++i) out12[i] = 0;
    for (int s = 0; s < 1024; ++s)
        for (int i = 0; i < NSTAMP; ++i) out12[i] += h[s][i], h[s][i] = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_wino4_stamp), h, sizeof(h)) != hipSuccess) return mrefsr::fail(MREFSR_E_LAUNCH, "wino4_stamps: reset failed");
    return 0;
}
#endif
