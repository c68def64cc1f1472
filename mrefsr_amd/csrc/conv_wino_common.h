// Pieces shared by the two Winograd F(2x2, 3x3) kernels (conv_wino.hip: eight waves, two transform positions each;
// conv_wino4.hip: four 512-register waves, one transform row each): tile geometry, the raw-patch LDS layout, the packed-weight
// geometry, the exact fp16 split, and the hand-issued / hand-waited memory operations.
#pragma once
#include "conv_common.h"

namespace mrefsr_wino {
using namespace mrefsr_conv;

constexpr int TT = 8;                 // Winograd tiles per side of a block's tile
constexpr int NTILE = TT * TT;        // 64 tiles = 16 x 16 output pixels
constexpr int PP = 2 * TT + 2;        // 18: side of the input patch
// raw fp32 patch of one 16-channel chunk in LDS, in 16-byte units (one unit = 4 channels of a pixel):
//   unit(q, row, col) = q RAW_Q + row RAW_RS + (col & 1) RAW_CP + (col >> 1)        q = quarter of the chunk's 16 channels
// A lane of the transform is a Winograd tile (tx = lane & 7, ty = lane >> 3 & 3) and reads pixel (2 ty + r, 2 tx + c): with the
// columns split by parity consecutive tx are consecutive units, and two tile rows are 2 RAW_RS = 8 (mod 16) units apart, so the 16
// lanes ds_read_b128 serves per LDS cycle always hit 16 different 16-byte slots of the 256-byte bank window.
constexpr int RAW_CP = 10, RAW_RS = 20, RAW_Q = PP * RAW_RS + 1, RAW_BYTES = 4 * RAW_Q * 16;
constexpr int BIAS_MAX = 1024;   // bias vector of a launch staged in LDS, zero-padded to whole cout blocks
constexpr size_t WCH_HALVES = (size_t)16 * 2 * NB * KC;   // packed 16-bit values per (cout block, chunk): [xi][plane][cout 64][cin 16]

// 2 floats -> packed (vh, vl = fp16(v - vh)): one packed conversion, then the exact remainder v - vh formed and rounded by
// v_fma_mixlo / mixhi_f16 (fp16 source read in place, fp32 accumulate, result rounded to nearest even into one half of the
// destination): 3 instructions per pair.  vl is NOT scaled into v's binade here (conv_nhwc.hip stores fp16(vl 2^11) and pairs it
// with a weight plane uh 2^-11): below |v| = 2^-3 it is an fp16 subnormal (the MFMA honours them: tools/hazard/mfma_f16_denorm.hip)
// with an ABSOLUTE error <= 2^-25 instead of 2^-22 |v| -- far below the fp32 accumulation error of sums whose terms are O(1), and
// it saves the kernel, which is bound by VALU issue (13 VALU instructions per MFMA before; tools/ubench/mfma_valu_overlap.hip: an
// MFMA covers 4), the scaling multiplies of both operands: 64 of 310 instructions per wave and chunk.
__device__ __forceinline__ void split_pair(const float a, const float b, unsigned int &hi, unsigned int &lo)
{
    hi = pk_f16(a, b);
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(a));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(b));
}
// packed fp32: d = s * b + a with a scalar multiplier, d = a - b
__device__ __forceinline__ f32x2 pk_fma_s(const unsigned long long sgn2, const f32x2 b, const f32x2 a)   // sgn2: the multiplier twice (a scalar pair)
{
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "s"(sgn2), "v"(b), "v"(a));
    return d;
}
__device__ __forceinline__ f32x2 pk_sub(const f32x2 a, const f32x2 b)
{
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 lo2(const f32x4 v, const int h) { return h ? f32x2{v[2], v[3]} : f32x2{v[0], v[1]}; }
__device__ __forceinline__ void split_hl(const float a, const float b, const float c, const float d, u32x2 &hi, u32x2 &lo)
{
    unsigned int h0, h1, l0, l1;
    split_pair(a, b, h0, l0);
    split_pair(c, d, h1, l1);
    hi = u32x2{h0, h1};
    lo = u32x2{l0, l1};
}

// ---- memory operations of the chunk loop, issued and waited for by hand (as in conv_nhwc8_kernel): a patch piece or weight
// fragment is requested one or two steps before its use and the wait in front of the use names how many YOUNGER loads may stay in
// flight (loads return in issue order).  Left to the compiler the counts are merged over paths that cannot occur (the two wave
// groups run the steps' halves in opposite order) and every wait degenerates to "everything".
__device__ __forceinline__ void gload16(f32x4 &dst, const unsigned int voff, const void *sbase)
{
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
template <int OFF> __device__ __forceinline__ void gload16u(u32x4 &dst, const unsigned int voff, const void *sbase)
{
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
// (the guarded registers pass THROUGH the wait as read-write operands: every later use depends on it)
template <int N> __device__ __forceinline__ void vm_wait2(u32x4 &a, u32x4 &b) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void vm_wait3(f32x4 &a, f32x4 &b, f32x4 &c)
{
    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N) : "memory");
}
__device__ __forceinline__ const void *scalar_ptr(const void *p)
{
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)v), hi = __builtin_amdgcn_readfirstlane((unsigned int)(v >> 32));
    return reinterpret_cast<const void *>(((unsigned long long)hi << 32) | lo);
}

}  // namespace mrefsr_wino
