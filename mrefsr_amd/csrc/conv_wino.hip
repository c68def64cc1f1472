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
#include <type_traits>

#include "conv_wino_common.h"

namespace {
using namespace mrefsr_conv;
using namespace mrefsr_wino;

constexpr int X_LD = 32 + 4;          // floats per tile row of the exchange buffer ([wave 8][tile 64][X_LD]: one cout half per pass)
constexpr int X_BYTES = 8 * NTILE * X_LD * 4;
// Two raw buffers of TWO chunks each (one barrier per pair of chunks).  At a tile's end one buffer holds the next tile's first pair
// and the other is free: the exchange buffer is that free one plus the extra area on its side --  [extra | buffer 0 | buffer 1 | extra]
constexpr int RAW_BUF = 2 * RAW_BYTES;
constexpr int X_EXTRA = X_BYTES - RAW_BUF;
static_assert(X_EXTRA > 0, "conv_wino: exchange buffer smaller than a raw buffer");
constexpr int RAW_OFF = X_EXTRA;
constexpr int SINK_OFF = 2 * X_EXTRA + 2 * RAW_BUF;   // 16 bytes for the threads without patch pieces
constexpr int BIAS_OFF = SINK_OFF + 16;                // the launch's bias vector, zero-padded to whole cout blocks (no global
                                                       // loads the compiler would have to wait for inside the tile loop; BIAS_MAX floats)
constexpr int LDS_BYTES = BIAS_OFF + BIAS_MAX * 4;
static_assert(LDS_BYTES <= 160 * 1024, "conv_wino: LDS budget");
constexpr int NPF = 3;   // 16-byte pieces of the patch per thread

__device__ __forceinline__ f32x16 mma16(u32x4 a, u32x4 b, f32x16 c)
{
#if defined(WINO_ABL) && WINO_ABL == 2
    c[0] += __uint_as_float(a[0] ^ b[0]);
    return c;
#endif
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
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

// The same for one of the two wave groups only (`on`: wave-uniform).  The branch lives INSIDE the asm statement: a C++ branch
// around a hand-waited load or its wait makes the register allocator copy the destination registers at the join -- while the
// loads are in flight (seen: v_mov of the three patch registers right behind their global_loads, wrong tiles).
__device__ __forceinline__ void vm_wait3_if(const int on, f32x4 &a, f32x4 &b, f32x4 &c)
{
    const int son = __builtin_amdgcn_readfirstlane(on);
    asm volatile("s_cmp_eq_u32 %3, 0\n\ts_cbranch_scc1 .Lwino_w%=\n\ts_waitcnt vmcnt(8)\n.Lwino_w%=:" : "+v"(a), "+v"(b), "+v"(c) : "s"(son) : "memory", "scc");
}
__device__ __forceinline__ void gload3_if(const int on, f32x4 &a, f32x4 &b, f32x4 &c, const unsigned int v0, const unsigned int v1, const unsigned int v2,
                                          const void *sbase)
{
    const int son = __builtin_amdgcn_readfirstlane(on);
    asm volatile("s_cmp_eq_u32 %7, 0\n\ts_cbranch_scc1 .Lwino_g%=\n\tglobal_load_dwordx4 %0, %3, %6\n\tglobal_load_dwordx4 %1, %4, %6\n\t"
                 "global_load_dwordx4 %2, %5, %6\n.Lwino_g%=:"
                 : "+v"(a), "+v"(b), "+v"(c)
                 : "v"(v0), "v"(v1), "v"(v2), "s"(sbase), "s"(son)
                 : "memory", "scc");
}

#ifdef WINO_STAMP
// instrumentation build (tools/conv_wino_stamp.py): shader-clock totals per phase of a wave's life, summed over all waves
//   0 transform + multiply of a chunk | 1 next weight fragments | 2 barrier | 3 raw store (wait for the patch) | 4 next patch request |
//   5 output exchange + epilogue | 6 tile bookkeeping | 7 waves
__device__ unsigned long long g_wino_stamp[1024][8];
#define WSTAMP(i)                                                      \
    {                                                                 \
        const unsigned long long t_now = __builtin_readcyclecounter(); \
        st_acc[i] += t_now - t_last;                                  \
        t_last = t_now;                                               \
    }
#else
#define WSTAMP(i)
#endif

#if defined(MREFSR_AB_KERNELS) && !defined(WINO_ARRIVAL_CHECK)
#define WINO_ARRIVAL_CHECK 1
#endif
#ifdef WINO_ARRIVAL_CHECK
// Checking build (the A/B library): a software shadow of the in-order vmcnt counter per wave -- loads issued so far, loads known to
// have returned (a wait vmcnt(N) proves "all but the N youngest"; a drain proves all), and the issue count behind the patch pieces /
// the weight fragments a wait guards.  A guarded load that the wait's count does not reach is counted here: deterministic, independent
// of how long the loads really took (the fault of the first sub-step needed slow loads to show in the results, about one run in four;
// a build with it put back, -DWINO_BUG_FIRST_WAIT, counts one late piece per block and wave on every launch).  Stores and the
// compiler's own loads are not in the shadow: they only make the hardware's wait cover more.
// tests/test_kernels_gpu.py::test_conv_wino_counted_waits_cover_their_loads reads the counters after launches of every instantiation.
__device__ unsigned int g_wino_late[2];   // [0] patch pieces, [1] weight fragments that arrived after their counted wait
#endif

// Block = 512 threads = 8 waves = 8 x 8 Winograd tiles (16 x 16 output pixels) x 64 couts; persistent (one block per CU walks its
// share of the tile list, the chunk stream runs on across tile boundaries).  Wave (i = wave & 3, jh = wave >> 2) owns the transform
// positions xi = (i, 2 jh), (i, 2 jh + 1) for all 64 tiles and all 64 couts: 2 xi x 2 tile halves x 2 cout halves = 8 accumulator
// tiles (128 VGPRs).  Everything an MFMA reads is private to the wave:
//   A = weight fragments (rows = couts) straight from the packed weights (8 loads per chunk, requested a chunk ahead; no two
//       waves read the same fragment);
//   B = the wave's OWN transform output: lane (tile, k half) forms row i of B^T d (two patch rows, wave-uniform choice) and the two
//       columns 2 jh, 2 jh + 1 of (B^T d) B for its tile and 8 channels from 12 ds_read_b128 of the raw patch, splits them and feeds
//       the registers to the MFMAs -- the transformed tile never exists in memory.  (Passing V through LDS, one transform per
//       block, cost 64 KB of LDS stores + 128 KB of loads per chunk and a second barrier: 5.8 k clocks per chunk against 1.5 k of
//       MFMA time.)
// Only the raw fp32 patch is shared: global -> registers (requested two steps ahead) -> LDS (double-buffered: one barrier per
// chunk).  Waves 4-7 store / request before they compute, waves 0-3 after (wave w and w + 4 share a SIMD: one wave's selects and
// LDS stores sit beside its partner's MFMAs); the accumulators pass through no branch.
// Output: each wave folds its two columns, the partial sums of the 8 waves meet through LDS in four passes (output column parity x
// cout half), thread = (tile, 4 couts) adds them up with the row signs and runs the epilogue on its two pixels of the pass.
template <int RES>   // 0: no tensor added in the epilogue, 1: residual (after the activation), 2: pre (before it) on the fast path
__global__ __launch_bounds__(512, 2) void conv_wino_kernel(const ConvArgs A)
{
#ifdef WINO_STAMP
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last = __builtin_readcyclecounter();
#endif
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char *const raw = smem + RAW_OFF;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int H = A.H, W = A.W;
    // Workgroups go to the 8 XCDs round-robin: block b works in the contiguous band b & 7 of the tile list (cout block fastest, then
    // x, y, image: neighbouring tiles meet in one L2).
    const int tiles_x = (W + 2 * TT - 1) / (2 * TT), tiles_y = (H + 2 * TT - 1) / (2 * TT);
    const int n_tiles = A.n_cb * tiles_x * tiles_y * A.wino_N, per = (n_tiles + 7) >> 3;
    const int band0 = (blockIdx.x & 7) * per, band1 = min(band0 + per, n_tiles), tstep = gridDim.x >> 3;
    int tile = band0 + (blockIdx.x >> 3);
    if (tile >= band1) return;
    for (int i = tid; i < A.n_cb * NB; i += 512)   // (visible after the prologue's barrier)
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

    const int wi = wv & 3, wjh = wv >> 2;
    f32x16 acc[2][2][2];   // [jj][tile half][cout half]

    // ---- stage 1: global -> registers -> raw tile.  Thread tid < 432 owns quarter tid & 3 (4 channels) of patch pixel (py0, px) =
    // divmod(tid >> 2, 18) and of the pixels 6 and 12 rows below it (4 lanes read the 64 contiguous bytes of a pixel; one base
    // address and one LDS offset per thread, the other two pieces at constant strides)
    constexpr int PROWS = 6;
    static_assert(NPF * PROWS == PP && PROWS * PP * 4 <= 512, "conv_wino: patch piece assignment");
    f32x4 pf[NPF];
    const int p_q = tid & 3, p_py = (tid >> 2) / PP, p_px = (tid >> 2) - p_py * PP;
    const bool p_have = tid < PROWS * PP * 4;
    const unsigned int praw0 = p_have ? (unsigned int)(RAW_OFF + (p_q * RAW_Q + p_py * RAW_RS + (p_px & 1) * RAW_CP + (p_px >> 1)) * 16) : (unsigned int)SINK_OFF;
    const unsigned int praw_step = p_have ? (unsigned int)(PROWS * RAW_RS * 16) : 0u, praw_sub = p_have ? (unsigned int)RAW_BYTES : 0u;
    // what fetch() reads: the tile whose chunks are being requested (runs ahead of the compute state at a tile's end)
    int pg0 = 0;              // pixel index (in an image) of piece 0; pieces 1, 2 lie PROWS, 2 PROWS rows below
    unsigned int f_ok = 0;    // bit k: piece k lies inside the image
    const float *xs1, *xs2;
    auto aim = [&](const Tile &t) {
        const int gy = t.y0 + p_py - 1, gx = t.x0 + p_px - 1;
        pg0 = gy * W + gx;
        f_ok = 0;
#pragma unroll
        for (int k = 0; k < NPF; ++k) f_ok |= (p_have && gy + k * PROWS >= 0 && gy + k * PROWS < H && gx >= 0 && gx < W) ? (1u << k) : 0u;
        xs1 = A.x1 + (size_t)(t.n % A.N1) * H * W * A.ld1;
        xs2 = A.x2 ? A.x2 + (size_t)(t.n % A.N2) * H * W * A.ld2 : A.x1;
    };
    const int q4 = (tid & 3) * 4;
    unsigned int okmask = 0;   // validity of the pieces in pf (f_ok of the fetch that filled them; 0: the chunk does not have these channels)
#ifdef WINO_ARRIVAL_CHECK
    int sw_issued = 0, sw_done = 0, mark_pf = 0, mark_uq = 0;
#endif
    // the request of a chunk in two parts: addresses (plain code, once per step) and the three loads (of the wave group whose turn it is)
    unsigned int f_vo[NPF], f_mask = 0;
    const void *f_sb = nullptr;
    auto fetch_prepare = [&](const int slot) {   // slot >= n_ch: the zero chunk that pads an odd chunk count to a pair
        const int ch = slot < A.n_ch ? slot : A.n_ch - 1;
        const bool first = ch < A.n_ch1;
        const int c0 = first ? ch * KC : (ch - A.n_ch1) * KC;
        const int Cs = first ? A.C1 : A.C2, ld = first ? A.ld1 : A.ld2;
        f_sb = scalar_ptr((first ? xs1 : xs2) + c0);
        const bool lane_ok = c0 + q4 < Cs;   // (a ragged last chunk: clamped address, the piece is zeroed in raw_store)
        f_mask = (lane_ok && slot < A.n_ch) ? f_ok : 0u;
        const unsigned int coff = lane_ok ? (unsigned int)q4 : 0u;
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const unsigned int pix = ((f_ok >> k) & 1u) ? (unsigned int)(pg0 + k * PROWS * W) : 0u;   // (clamped into the image)
            f_vo[k] = (pix * (unsigned int)ld + coff) * 4u;   // (wino_launch: H W ld 4 < 2^32)
        }
    };
    auto fetch_issue = [&](const int on) {   // NPF loads (if `on`): the waits count them
        if (on) okmask = f_mask;
        gload3_if(on, pf[0], pf[1], pf[2], f_vo[0], f_vo[1], f_vo[2], f_sb);
#ifdef WINO_ARRIVAL_CHECK
        if (on) sw_issued += 3, mark_pf = sw_issued;
#endif
    };
    float amax = 0.f;   // fp16 range guard: largest |x| seen (after the input scale)
    float oamx = 0.f;   // max |out| of what this lane stores (A.stat_amax)
    // Input scale (A.in_amax: max |x| of the input tensor(s), device memory, may be NULL): x is multiplied by the power of two that
    // brings that maximum into [2^11, 2^12) -- the transform's growth of 4 stays far inside the fp16 range and the LOW term of an
    // activation, fp16(v - fp16(v)), is a normal number down to |v| = 2^-14 of the maximum instead of a subnormal below |x| = 2^-3
    // (absolute error 2^-25: visible against activations of 1e-2 and less) -- and the result by its inverse.  Exact either way.
    float in_s = 1.f, oscale_in = A.out_scale;
    if (A.in_amax) {
        const float am = *A.in_amax;
        if (am > 1.0e-30f && am < 3.0e38f) {
            int e;
            (void)frexpf(am, &e);                      // am = m 2^e, m in [0.5, 1)
            in_s = ldexpf(1.f, 12 - e);
            oscale_in = A.out_scale * ldexpf(1.f, e - 12);
        }
    }
    in_s = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(in_s)));
    auto raw_store = [&](const int slot, const int on) {   // slot = 2 buffer + sub-chunk
        vm_wait3_if(on, pf[0], pf[1], pf[2]);   // younger: the 8 weight fragments requested after this patch
#ifdef WINO_ARRIVAL_CHECK
        if (on) {
            sw_done = sw_issued - 8 > sw_done ? sw_issued - 8 : sw_done;
            if (mark_pf > sw_done && lane == 0) atomicAdd(&g_wino_late[0], 1u);
        }
#endif
        if (on) {
#pragma unroll
            for (int k = 0; k < NPF; ++k) {
                const bool ok = (okmask >> k) & 1u;
                float4 v;
                v.x = ok ? pf[k][0] : 0.f, v.y = ok ? pf[k][1] : 0.f, v.z = ok ? pf[k][2] : 0.f, v.w = ok ? pf[k][3] : 0.f;
                // (the input scale as volatile asm: a plain multiply is speculated out of `if (on)` and reads the registers of the
                // other wave group's loads while they are in flight -- harmless, but the static checker rightly refuses it)
                asm volatile("v_mul_f32 %0, %4, %0\n\tv_mul_f32 %1, %4, %1\n\tv_mul_f32 %2, %4, %2\n\tv_mul_f32 %3, %4, %3"
                             : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w) : "s"(in_s));
                asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(amax) : "v"(v.x), "v"(v.y));   // (a NaN input is not caught here: it reaches the output)
                asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(amax) : "v"(v.z), "v"(v.w));
                *reinterpret_cast<float4 *>(smem + praw0 + slot * praw_sub + k * praw_step) = v;
            }
        }
    };

    // ---- stage 2: the wave's transform.  Row i of B^T d = d[ra] + sr d[rb]:  i 0: d0 - d2,  1: d1 + d2,  2: d2 - d1,  3: d1 - d3;
    // its two columns of (B^T d) B from three tile columns (a, b, c):  v0 = a - c,  v1 = sc b + c
    //   jh 0: (x0, x1, x2), sc +1: j 0 = x0 - x2, j 1 = x1 + x2  |  jh 1: (x2, x3, x1), sc -1: j 2 = x2 - x1, j 3 = x1 - x3
    const int t_ra = wi == 0 ? 0 : (wi == 2 ? 2 : 1), t_rb = wi == 0 ? 2 : (wi == 1 ? 2 : (wi == 2 ? 1 : 3));
    const unsigned int t_srb = __builtin_amdgcn_readfirstlane(wi == 1 ? 0x3f800000u : 0xbf800000u), t_scb = __builtin_amdgcn_readfirstlane(wjh ? 0xbf800000u : 0x3f800000u);
    const unsigned long long t_sr = ((unsigned long long)t_srb << 32) | t_srb, t_sc = ((unsigned long long)t_scb << 32) | t_scb;   // +-1.0 twice
    auto col_off = [&](const int c) { return ((c & 1) * RAW_CP + (c >> 1)) * 16; };
    const int t_ca = col_off(wjh ? 2 : 0), t_cb = col_off(wjh ? 3 : 1), t_cc = col_off(wjh ? 1 : 2);
    // lane = (tx = l31 & 7, ty low bits = l31 >> 3, k half): quarter 2 kh (+ g0) of pixel (2 ty, 2 tx) of tile half 0
    const unsigned char *const t_src = raw + ((2 * kh) * RAW_Q + (2 * (l31 >> 3)) * RAW_RS + (l31 & 7)) * 16;
    const int t_oa = t_ra * RAW_RS * 16, t_ob = t_rb * RAW_RS * 16;
    // -> B operands of tile half tt: vh[jj], vl[jj] = 8 channels (k half kh) of V[(i, 2 jh + jj)] for tile 32 tt + l31, split
    auto transform = [&](const int slot, const int tt, u32x4 (&vh)[2], u32x4 (&vl)[2]) {
        const unsigned char *const src = t_src + slot * RAW_BYTES + tt * (8 * RAW_RS * 16);
#pragma unroll
        for (int g0 = 0; g0 < 2; ++g0) {
            const unsigned char *const sg = src + g0 * (RAW_Q * 16);
            const f32x4 a0 = *reinterpret_cast<const f32x4 *>(sg + t_oa + t_ca), b0 = *reinterpret_cast<const f32x4 *>(sg + t_ob + t_ca);
            const f32x4 a1 = *reinterpret_cast<const f32x4 *>(sg + t_oa + t_cb), b1 = *reinterpret_cast<const f32x4 *>(sg + t_ob + t_cb);
            const f32x4 a2 = *reinterpret_cast<const f32x4 *>(sg + t_oa + t_cc), b2 = *reinterpret_cast<const f32x4 *>(sg + t_ob + t_cc);
            // packed fp32 arithmetic, two channels per instruction: plain and scalar-source forms only (the forms that read src.hi
            // through op_sel return wrong values beside 16-bit MFMAs on gfx950: profiles/r1_pk_mul_hazard.txt)
#pragma unroll
            for (int h = 0; h < 2; ++h) {   // channel pairs (0, 1), (2, 3) of the quarter
                unsigned int hi, lo;
                const f32x2 pa = pk_fma_s(t_sr, lo2(b0, h), lo2(a0, h)), pb = pk_fma_s(t_sr, lo2(b1, h), lo2(a1, h)), pc = pk_fma_s(t_sr, lo2(b2, h), lo2(a2, h));
                const f32x2 v0 = pk_sub(pa, pc), v1 = pk_fma_s(t_sc, pb, pc);
                split_pair(v0[0], v0[1], hi, lo);
                vh[0][2 * g0 + h] = hi, vl[0][2 * g0 + h] = lo;
                split_pair(v1[0], v1[1], hi, lo);
                vh[1][2 * g0 + h] = hi, vl[1][2 * g0 + h] = lo;
            }
        }
    };

    // ---- stage 3: the wave's 8 weight fragments of a chunk ((jj, cout half) x (uh, ul)), requested a whole step before their MFMAs
    const unsigned int u_voff = (unsigned int)(l31 * KC + kh * 8) * 2u;
    const unsigned short *const u_wave = A.wp + (size_t)(wi * 4 + 2 * wjh) * 2 * NB * KC;
    u32x4 uq[2][2][2];   // [jj][cout half][uh | ul]
    const void *u_s0 = nullptr, *u_s1 = nullptr;
    auto fetch_u_prepare = [&](const int cb, const int ch) {
        const unsigned char *b0 = reinterpret_cast<const unsigned char *>(u_wave + ((size_t)cb * A.n_ch + ch) * WCH_HALVES);
        u_s0 = scalar_ptr(b0), u_s1 = scalar_ptr(b0 + 4096);   // xi jj = 1 lies 2 planes x 64 couts x 16 cin x 2 bytes further
    };
    auto fetch_u_pair = [&](const int jj, const int ct) {   // 2 loads; the 4 pairs of a chunk always in the order (0,0) (0,1) (1,0) (1,1)
        const void *sb = jj ? u_s1 : u_s0;
        if (ct == 0) gload16u<0>(uq[jj][0][0], u_voff, sb), gload16u<2048>(uq[jj][0][1], u_voff, sb);
        else gload16u<1024>(uq[jj][1][0], u_voff, sb), gload16u<3072>(uq[jj][1][1], u_voff, sb);
#ifdef WINO_ARRIVAL_CHECK
        sw_issued += 2, mark_uq = sw_issued;
#endif
    };
    auto fetch_u = [&](const int cb, const int ch) {   // all 8 at once (the first chunk of a block)
        fetch_u_prepare(cb, ch);
        fetch_u_pair(0, 0), fetch_u_pair(0, 1), fetch_u_pair(1, 0), fetch_u_pair(1, 1);
    };
    // (younger than the 8 fragments: the 3 patch pieces requested after them.  One wait for all eight -- they were requested a whole
    // step ago -- so that nothing stands between the MFMAs of tile half 0 and the transform of tile half 1: the scheduler may fill
    // the shadows of the one with the other)
    auto frags_arrived = [&]() {
        asm volatile("s_waitcnt vmcnt(3)"
                     : "+v"(uq[0][0][0]), "+v"(uq[0][0][1]), "+v"(uq[0][1][0]), "+v"(uq[0][1][1]), "+v"(uq[1][0][0]), "+v"(uq[1][0][1]), "+v"(uq[1][1][0]),
                       "+v"(uq[1][1][1])
                     :
                     : "memory");
#ifdef WINO_ARRIVAL_CHECK
        sw_done = sw_issued - 3 > sw_done ? sw_issued - 3 : sw_done;
        if (mark_uq > sw_done && lane == 0) atomicAdd(&g_wino_late[1], 1u);
#endif
    };
    // tile half 1 is the fragments' last use: the pair of the NEXT chunk is requested right behind its three MFMAs -- eight loads in a
    // burst at the end of the step queued behind the other waves' bursts (0.8-1.2 k clocks per wave and step in the issue of 8 loads)
    auto multiply = [&](const int tt, const u32x4 (&vh)[2], const u32x4 (&vl)[2], const bool refill) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const u32x4 uh = uq[jj][ct][0], ul = uq[jj][ct][1];
                // partial products, smallest first
                acc[jj][tt][ct] = mma16(ul, vh[jj], acc[jj][tt][ct]);
                acc[jj][tt][ct] = mma16(uh, vl[jj], acc[jj][tt][ct]);
                acc[jj][tt][ct] = mma16(uh, vh[jj], acc[jj][tt][ct]);
                if (refill) fetch_u_pair(jj, ct);
            }
    };

    // ---- the chunk stream, in PAIRS of chunks (slots 0 .. S - 1 of a tile, S = n_ch rounded up to even; a slot past n_ch is a zero
    // chunk).  Sub-step s: [waves 4-7: advance] transform + multiply slot s from raw buffer `par`, request the weight fragments of
    // slot s + 1, [waves 0-3: advance];  advance = store slot s + 2 (in registers since the previous sub-step) to the other raw
    // buffer (same half) and request the patch of slot s + 3.  One barrier per pair.  "Next" runs on into the next tile of the block.
    const int n_ch = A.n_ch, S = (n_ch + 1) & ~1;   // (n_ch >= 3: wino_launch)
    Tile cur = decode(tile), nxt = cur;
    int par = 0;
#ifndef WINO_VARIANT
#define WINO_VARIANT 0
#endif
    const int early = WINO_VARIANT == 1 ? 0 : (WINO_VARIANT == 2 ? 1 : wv >= 4);   // (variants 1 / 2: every wave stores / requests after / before computing)
    if (early && WINO_VARIANT != 3) __builtin_amdgcn_s_setprio(1);   // (the second-dispatched half loses the VALU arbitration by age otherwise)
    auto advance = [&](const int sub, int on) {   // (the addresses of the request: fetch_prepare, once per sub-step)
#if defined(WINO_ABL) && WINO_ABL == 3
        on = 0;
#endif
        WSTAMP(1)
        raw_store(2 * (par ^ 1) + sub, on);
        WSTAMP(3)
        fetch_issue(on);
        WSTAMP(4)
    };
    aim(cur);
    fetch_prepare(0), fetch_issue(1);
    fetch_u(cur.cb, 0);
    raw_store(0, 1);
    fetch_prepare(1), fetch_issue(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the prologue's only: the patch of slot 1 was requested AFTER the weight fragments)
#ifdef WINO_ARRIVAL_CHECK
    sw_done = sw_issued;
#endif
    raw_store(1, 1);
    fetch_prepare(2), fetch_issue(1);
    // The chunk loop's counted waits rely on what is outstanding at the start of a sub-step: for waves 4-7 the patch of slot s + 2 and,
    // YOUNGER, the 8 weight fragments of slot s -- "vmcnt(8)" in front of their store.  Nothing younger exists yet at the first
    // sub-step of a block, so that wait would pass with the patch of slot 2 still in flight (wrong first tiles whenever the loads were
    // slow: launches whose tensors miss the caches; tests/test_kernels_gpu.py::test_conv_wino_large_launch_*).  Once per block:
#ifndef WINO_BUG_FIRST_WAIT   // (defined only to show that the checking build sees the fault: tools/conv_wino_arrival_check.py)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef WINO_ARRIVAL_CHECK
    sw_done = sw_issued;
#endif
#endif
    __syncthreads();
    const float slope = A.slope_ptr ? *A.slope_ptr : A.slope;
    for (;;) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[jj][t][ct][e] = 0.f;
        const int tile_n = tile + tstep;
        const bool more = tile_n < band1;
        nxt = decode(more ? tile_n : tile);
        WSTAMP(6)
        for (int s0 = 0; s0 < S; s0 += 2) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const int s = s0 + sub;
                // the patch request of this sub-step: slot s + 3 of the stream (aimed at the next tile from its first slot on)
                if (s + 3 == S) aim(nxt);
                fetch_prepare(s + 3 < S ? s + 3 : s + 3 - S);
                advance(sub, early);
                WSTAMP(6)
                {
                    u32x4 vh[2], vl[2];
#ifndef WINO_ABL
#define WINO_ABL 0
#endif
                    u32x4 wh[2], wl[2];
                    if (WINO_ABL == 1) vh[0] = vh[1] = vl[0] = vl[1] = wh[0] = wh[1] = wl[0] = wl[1] = u32x4{(unsigned)s, (unsigned)tid, 3u, 4u};
                    if (WINO_ABL != 1) transform(2 * par + sub, 0, vh, vl);
                    frags_arrived();
                    {   // the fragments of slot s + 1 (a zero slot: any valid chunk -- its products vanish with the zero patch)
                        const int sn = s + 1 < S ? s + 1 : 0;
                        fetch_u_prepare(s + 1 < S ? cur.cb : nxt.cb, sn < n_ch ? sn : n_ch - 1);
                    }
                    multiply(0, vh, vl, false);
                    if (WINO_ABL != 1) transform(2 * par + sub, 1, wh, wl);
                    multiply(1, wh, wl, WINO_ABL != 5);
                }
                WSTAMP(0)
                advance(sub, !early);
                WSTAMP(1)
            }
            __syncthreads();   // the other raw buffer is complete, this one is free
            WSTAMP(2)
            par ^= 1;
        }

        // ---- stage 4: output transform.  Columns in registers: the wave's share of Z[b] = sum_j A^T[b][j] M[i][j]
        //   (A^T = [1 1 1 0; 0 1 -1 -1]):   jh 0: b 0: m0 + m1, b 1: m1   |   jh 1: b 0: m2, b 1: -m2 - m3
        // accumulator register e of [jj][t][ct] holds cout 32 ct + 8 (e >> 2) + 4 kh + (e & 3) of Winograd tile 32 t + l31.
        // Four passes (cout half hc x output column parity b) through the exchange buffer [wave][tile][X_LD]; thread = (tile T, 4 couts)
        // adds the 8 partial sums up with the row signs -- y[a] = Z0 + Z1 + Z2 (a = 0), Z1 - Z2 - Z3 (a = 1), Z_i = jh 0 part + jh 1
        // part -- and runs the direct kernel's epilogue on the pixels (2 ty + a, 2 tx + b) of the pass.
        const int cb = cur.cb, n = cur.n, y0 = cur.y0, x0 = cur.x0;
        // (buffer `par` holds the next tile's first pair by now; the other one has just been multiplied and is free)
        float *const xb = reinterpret_cast<float *>(smem + ((par ^ 1) ? X_EXTRA + RAW_BUF : 0));
        const int Cout = A.Cout;
        const float oscale = oscale_in;
        const int T = tid >> 3, ty = T >> 3, tx = T & 7, gy0 = y0 + 2 * ty;
        // the common case -- a tile inside the image, a full cout block, plain epilogue -- without per-pixel bounds tests and with
        // every address an offset from one pointer of the tile (the general path below spent ~150 VALU instructions per pass)
        const bool fast = y0 + 2 * TT <= H && x0 + 2 * TT <= W && cb * NB + NB <= Cout && (A.ld_out & 3) == 0 && (Cout & 3) == 0 && A.epilogue == 0 &&
                          (RES == 1 ? !A.pre && (A.ld_res & 3) == 0 : RES == 2 ? A.pre && !A.residual : !A.pre && !A.residual);
        float *const o00 = A.out + ((size_t)(n * H + gy0) * W + (x0 + 2 * tx)) * A.ld_out + cb * NB + (tid & 7) * 4;
        const int ld_r = RES == 2 ? Cout : A.ld_res;   // (pre is dense [pre_N][H][W][Cout], batch-broadcast)
        const float *const r00 = RES == 1 ? A.residual + ((size_t)(n * H + gy0) * W + (x0 + 2 * tx)) * ld_r + cb * NB + (tid & 7) * 4 :
                                 RES == 2 ? A.pre + ((size_t)((n % A.pre_N) * H + gy0) * W + (x0 + 2 * tx)) * ld_r + cb * NB + (tid & 7) * 4 : nullptr;
#pragma unroll
        for (int hc = 0; hc < (WINO_ABL == 4 ? 0 : 2); ++hc) {
            const int c4 = (tid & 7) * 4, co = cb * NB + hc * 32 + c4;
            const bool cok = co < Cout;
            const bool vec = (co + 3 < Cout) && ((A.ld_out & 3) == 0) && ((Cout & 3) == 0);
            const float4 bv = *reinterpret_cast<const float4 *>(smem + BIAS_OFF + (cb * NB + hc * 32 + c4) * 4);
            float4 pool = make_float4(0.f, 0.f, 0.f, 0.f);   // epilogue 1: running maximum of the tile's four pixels
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                if (hc | b) __syncthreads();   // the previous pass has been read
                const int gx = x0 + 2 * tx + b;
                float4 rq[2];
                if constexpr (RES != 0) {   // residual / pre term of the pass's two pixels, requested at the START of the pass: its latency lies
                                            // under the exchange writes and the barrier (requested after the barrier it cost 3.2-3.8 k clocks per tile)
                    if (fast) {
                        rq[0] = ld_f4(r00 + (size_t)b * ld_r + hc * 32, A.stream_out);
                        rq[1] = ld_f4(r00 + (size_t)(W + b) * ld_r + hc * 32, A.stream_out);
                    } else if (RES == 1) {
#pragma unroll
                        for (int a = 0; a < 2; ++a) {
                            const int gy = gy0 + a < H ? gy0 + a : H - 1, gxc = gx < W ? gx : W - 1;
                            rq[a] = ld_f4(A.residual + (((size_t)n * H + gy) * W + gxc) * A.ld_res + (cok ? co : 0), A.stream_out);
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    float *const xrow = xb + ((size_t)wv * NTILE + t * 32 + l31) * X_LD + 4 * kh;
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        float4 z;
                        const int e = 4 * qd;
                        const f32x16 &m0 = acc[0][t][hc], &m1 = acc[1][t][hc];
                        if (wjh == 0) {
                            if (b == 0) z = make_float4(m0[e] + m1[e], m0[e + 1] + m1[e + 1], m0[e + 2] + m1[e + 2], m0[e + 3] + m1[e + 3]);
                            else z = make_float4(m1[e], m1[e + 1], m1[e + 2], m1[e + 3]);
                        } else {
                            if (b == 0) z = make_float4(m0[e], m0[e + 1], m0[e + 2], m0[e + 3]);
                            else z = make_float4(-m0[e] - m1[e], -m0[e + 1] - m1[e + 1], -m0[e + 2] - m1[e + 2], -m0[e + 3] - m1[e + 3]);
                        }
                        *reinterpret_cast<float4 *>(xrow + 8 * qd) = z;
                    }
                }
                __syncthreads();
                float4 z[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 p = *reinterpret_cast<const float4 *>(xb + ((size_t)i * NTILE + T) * X_LD + c4);
                    const float4 q = *reinterpret_cast<const float4 *>(xb + ((size_t)(4 + i) * NTILE + T) * X_LD + c4);
                    z[i] = make_float4(p.x + q.x, p.y + q.y, p.z + q.z, p.w + q.w);
                }
                float4 y[2];
                y[0] = make_float4(z[0].x + z[1].x + z[2].x, z[0].y + z[1].y + z[2].y, z[0].z + z[1].z + z[2].z, z[0].w + z[1].w + z[2].w);
                y[1] = make_float4(z[1].x - z[2].x - z[3].x, z[1].y - z[2].y - z[3].y, z[1].z - z[2].z - z[3].z, z[1].w - z[2].w - z[3].w);
                if (fast) {
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        float4 v = y[a];
                        v.x = v.x * oscale + bv.x, v.y = v.y * oscale + bv.y, v.z = v.z * oscale + bv.z, v.w = v.w * oscale + bv.w;
                        if constexpr (RES == 2) v.x += rq[a].x, v.y += rq[a].y, v.z += rq[a].z, v.w += rq[a].w;
                        if (A.act) {
                            v.x = v.x > 0.f ? v.x : v.x * slope, v.y = v.y > 0.f ? v.y : v.y * slope;
                            v.z = v.z > 0.f ? v.z : v.z * slope, v.w = v.w > 0.f ? v.w : v.w * slope;
                        }
                        if constexpr (RES == 1) v.x += rq[a].x, v.y += rq[a].y, v.z += rq[a].z, v.w += rq[a].w;
                        oamx = fmaxf(fmaxf(fmaxf(oamx, fabsf(v.x)), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
                        st_f4(o00 + (size_t)(a * W + b) * A.ld_out + hc * 32, v, A.stream_out);
                    }
                    continue;
                }
                if (A.epilogue == 1) {   // MaxPool2d(2,2) of act(conv + bias) = act(max4 + bias): the tile IS the pooling window
                    float4 m = make_float4(fmaxf(y[0].x, y[1].x), fmaxf(y[0].y, y[1].y), fmaxf(y[0].z, y[1].z), fmaxf(y[0].w, y[1].w));
                    if (b == 0) {
                        pool = m;
                        continue;
                    }
                    m = make_float4(fmaxf(m.x, pool.x), fmaxf(m.y, pool.y), fmaxf(m.z, pool.z), fmaxf(m.w, pool.w));
                    float4 v = make_float4(m.x * oscale + bv.x, m.y * oscale + bv.y, m.z * oscale + bv.z, m.w * oscale + bv.w);
                    if (A.act) {
                        v.x = v.x > 0.f ? v.x : v.x * slope, v.y = v.y > 0.f ? v.y : v.y * slope;
                        v.z = v.z > 0.f ? v.z : v.z * slope, v.w = v.w > 0.f ? v.w : v.w * slope;
                    }
                    const int Ho = H >> 1, Wo = W >> 1, py = gy0 >> 1, px = (x0 >> 1) + tx;
                    if (cok && py < Ho && px < Wo) {
                        oamx = fmaxf(oamx, fabsf(v.x));
                        if (co + 1 < Cout) oamx = fmaxf(oamx, fabsf(v.y));
                        if (co + 2 < Cout) oamx = fmaxf(oamx, fabsf(v.z));
                        if (co + 3 < Cout) oamx = fmaxf(oamx, fabsf(v.w));
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
                        v.x += rq[a].x, v.y += rq[a].y, v.z += rq[a].z, v.w += rq[a].w;
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
                    oamx = fmaxf(oamx, fabsf(v.x));
                    if (co + 1 < Cout) oamx = fmaxf(oamx, fabsf(v.y));
                    if (co + 2 < Cout) oamx = fmaxf(oamx, fabsf(v.z));
                    if (co + 3 < Cout) oamx = fmaxf(oamx, fabsf(v.w));
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
        WSTAMP(5)
        if (!more) break;
        // the exchange buffer lies over the raw buffer the next tile's first sub-steps store into: every wave has to be through with
        // the last pass's reads first (without this barrier a wave that ran ahead overwrote them -- seen only on launches whose
        // tensors miss the caches, where the waves of a block drift apart: tests/test_kernels_gpu.py::test_conv_wino_large_launch_*)
        __syncthreads();
        tile = tile_n;
        cur = nxt;
    }
    // requests past the last chunk are still in flight: wait, and keep their destination registers "in use" up to here -- to the
    // compiler they were dead after the loop, and anything it had placed in them before the wait would be overwritten on arrival
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < NPF; ++k) asm volatile("" ::"v"(pf[k]) : "memory");
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) asm volatile("" ::"v"(uq[jj][ct][sp]) : "memory");
    if (A.range_flag && !(amax <= 16000.f)) atomicOr(A.range_flag, 1);
    if (A.stat_amax) {   // max |out| of the launch: the next Winograd layer's input scale (ConvArgs::stat_amax as out_amax)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) oamx = fmaxf(oamx, __shfl_xor(oamx, o, 64));
        if (lane == 0 && oamx > 0.f && oamx < 3.0e38f && __float_as_uint(oamx) > __builtin_nontemporal_load(A.stat_amax)) atomicMax(A.stat_amax, __float_as_uint(oamx));
    }
#ifdef WINO_STAMP
    WSTAMP(6)
    st_acc[7] = 1;
    if (lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&g_wino_stamp[(blockIdx.x * 8 + wv) & 1023][i], st_acc[i]);
#endif
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
    if (a.res_mask || a.stat_sum || a.io16 || a.epilogue == 3)
        return mrefsr::fail(MREFSR_E_UNSUPPORTED, "conv_wino: training statistics / bf16 storage / DynAgg epilogue are the direct kernel's");
    if (a.n_cb * NB > BIAS_MAX) return mrefsr::fail(MREFSR_E_UNSUPPORTED, "conv_wino: more than %d output channels: the direct kernel's", BIAS_MAX);
    if (a.n_ch < 3) return mrefsr::fail(MREFSR_E_UNSUPPORTED, "conv_wino: fewer than 33 input channels (two K chunks): the direct kernel's");
    // (0xffff0000 = conv_wino4.hip's OOB: the offset it gives out-of-image lanes must lie beyond the buffer's num_records)
    if ((size_t)a.H * a.W * (size_t)(a.ld1 > a.ld2 ? a.ld1 : a.ld2) * 4 >= (size_t)0xffff0000u)
        return mrefsr::fail(MREFSR_E_UNSUPPORTED, "conv_wino: an image of the input exceeds 0xffff0000 bytes (32-bit patch offsets): the direct kernel's");
    static unsigned long long attr = 0;
    static int n_cu[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (mrefsr::first_use_on_device(attr)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv_wino_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv_wino_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv_wino_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        int cu = 0;
        if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0) cu = 256;
        if (dev >= 0 && dev < 64) n_cu[dev] = cu;
    }
    ConvArgs b = a;
    b.wino_N = N;
    const int tw = 2 * TT;
    const long tiles = (long)((a.W + tw - 1) / tw) * ((a.H + tw - 1) / tw) * a.n_cb * N;
    if (tiles >= ((long)1 << 31)) return mrefsr::fail(MREFSR_E_INVALID, "conv_wino: %ld tiles", tiles);
    // one persistent block per CU (the LDS footprint allows no second one), a multiple of 8 so that every XCD gets the same number
    int blocks = (dev >= 0 && dev < 64 && n_cu[dev] > 0) ? n_cu[dev] : 256;
    blocks = (blocks + 7) & ~7;
    const long need = ((tiles + 7) / 8) * 8;   // (a band of the tile list per XCD: at most ceil(tiles / 8) useful blocks in each)
    if (need < blocks) blocks = (int)need;
    b.stream_out = (size_t)N * a.H * a.W * a.ld_out * sizeof(float) > ((size_t)256 << 20);
    {   // The four-wave kernel (conv_wino4.hip) takes what it serves -- whole 16 x 16 tiles, whole cout blocks, the plain and the
        // max-pool epilogue -- and is 1.01-1.09 x this file's on every benchmark shape (profiles/r5_conv_wino4_check.txt), with the
        // same bits.  MREFSR_WINO_WAVES=8 keeps everything here (A/B runs and the tests flip it: read per call).
        const char *ew = getenv("MREFSR_WINO_WAVES");
        if (!(ew && ew[0] == '8') && wino4_serves(b)) return wino4_launch(b, blocks, stream);
    }
    const bool plain = a.epilogue == 0 && (a.Cout & 3) == 0 && (a.ld_out & 3) == 0;
    if (plain && a.residual && (a.ld_res & 3) == 0) hipLaunchKernelGGL((conv_wino_kernel<1>), dim3(blocks), dim3(512), LDS_BYTES, stream, b);
    else if (plain && a.pre && !a.residual) hipLaunchKernelGGL((conv_wino_kernel<2>), dim3(blocks), dim3(512), LDS_BYTES, stream, b);
    else hipLaunchKernelGGL((conv_wino_kernel<0>), dim3(blocks), dim3(512), LDS_BYTES, stream, b);
    return mrefsr::check_launch("conv_wino");
}

}  // namespace mrefsr_conv

#ifdef WINO_ARRIVAL_CHECK
// read-and-reset of the late-arrival counters (checking builds only): out2[0] patch pieces, out2[1] weight fragments
MREFSR_EXPORT int mrefsr_dbg_wino_late(unsigned int *out2)
{
    unsigned int z[2] = {0u, 0u};
    if (hipDeviceSynchronize() != hipSuccess) return mrefsr::fail(MREFSR_E_LAUNCH, "wino_late: sync failed");
    if (hipMemcpyFromSymbol(out2, HIP_SYMBOL(g_wino_late), sizeof(z)) != hipSuccess) return mrefsr::fail(MREFSR_E_LAUNCH, "wino_late: read failed");
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_wino_late), z, sizeof(z)) != hipSuccess) return mrefsr::fail(MREFSR_E_LAUNCH, "wino_late: reset failed");
    return 0;
}
#endif

#ifdef WINO_STAMP
// read-and-reset of the phase clocks (instrumentation builds only)
MREFSR_EXPORT int mrefsr_dbg_wino_stamps(unsigned long long *out8)
{
    static unsigned long long h[1024][8];
    if (hipDeviceSynchronize() != hipSuccess) return mrefsr::fail(MREFSR_E_LAUNCH, "wino_stamps: sync failed");
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_wino_stamp), sizeof(h)) != hipSuccess) return mrefsr::fail(MREFSR_E_LAUNCH, "wino_stamps: read failed");
    for (int i = 0; i < 8; ++i) out8[i] = 0;
    for (int s = 0; s < 1024; ++s)
        for (int i = 0; i < 8; ++i) out8[i] += h[s][i], h[s][i] = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_wino_stamp), h, sizeof(h)) != hipSuccess) return mrefsr::fail(MREFSR_E_LAUNCH, "wino_stamps: reset failed");
    return 0;
}
#endif
