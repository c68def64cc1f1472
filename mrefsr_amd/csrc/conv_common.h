// Pieces shared by the convolution kernels (conv_nhwc.hip: direct implicit GEMM; conv_wino.hip: Winograd F(2x2, 3x3)):
// vector typedefs, the packed 16-bit conversions of the operand splits, the launch arguments and the streaming accessors.
#pragma once
#include "common.h"

namespace mrefsr_conv {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));


constexpr int KC = 16, NB = 64;   // input channels per K chunk, output channels per block column

__device__ __forceinline__ bf16x8 as_bf(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// packed round-to-nearest-even bf16 of two floats, and back
__device__ __forceinline__ unsigned int pk_bf16(float a, float b)
{
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ float bf_lo(unsigned int p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(unsigned int p) { return __uint_as_float(p & 0xffff0000u); }

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int pk_f16(float a, float b)   // round-to-nearest-even
{
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2{a, b}, f16x2));
}
__device__ __forceinline__ f32x2 un_f16(unsigned int p) { return __builtin_convertvector(__builtin_bit_cast(f16x2, p), f32x2); }

// WH2 = wh * 2^-11 of 8 packed fp16: exact while the result is a normal fp16, round-to-nearest-even into the
// denormals exactly like the pack kernel's conversion (plain v_pk_mul_f16: safe next to MFMAs, tools/hazard/)
__device__ __forceinline__ u32x4 scale_wh(u32x4 v)
{
    const f16x2 k = {(_Float16)0.00048828125f, (_Float16)0.00048828125f};
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned int d = v[i];
        r[i] = __builtin_bit_cast(unsigned int, __builtin_bit_cast(f16x2, d) * k);
    }
    return r;
}

struct ConvArgs {
    const float *x1, *x2;
    const unsigned short *wp;
    const float *bias, *slope_ptr, *pre, *residual;
    float *out;
    int *range_flag;
    const float *in_amax;   // MODE 2, may be NULL: max |x| of the input tensor(s), in device memory -- the kernel scales x by 2^s
                            // (max 2^s in [2^13, 2^14)) before the fp16 split and the result by 2^-s: gradients, whose
                            // magnitudes would sit in the fp16 subnormals, through the three-product mode (training dgrad)
    int H, W, C1, ld1, N1, C2, ld2, N2, Cout, ld_out, ld_res, pre_N, n_ch1, n_ch, n_cb, act, epilogue;
    float slope, out_scale;
    // epilogue 3 (DynAgg glue, ref_mrapa_restoration_arch.py:56-73): planar offset (`out`) / mask outputs, pre-offsets, |offset| sum
    const float2 *dyn_pre;
    float *dyn_mask;
    double *dyn_abs;
    int dyn_ni;   // deformable groups x 9 taps
    int io16;     // MODE 3 only: x1 / x2 / pre / residual / out are bf16 tensors (2-byte storage, BASELINE configs[4])
    int stream_out;   // output larger than the last-level cache: non-temporal stores / residual loads
    // training (mrefsr_conv_nhwc_bwd_f32): the `residual` operand is a ReLU mask source (out = residual > 0 ? v : 0) instead of an
    // addend; per-channel sums (+=) and max |out| (atomic max of the bit pattern) of the launch's output, both zero-initialised
    int res_mask;
    float *stat_sum;
    unsigned int *stat_amax;
    int xcd_bands;    // 4-wave kernel: re-label the blocks so that an XCD works on a contiguous band of tiles
    int warm_w;       // 4-wave kernel, launches of few blocks: request 1 / warm_w of the block's weight slab before the chunk loop (0 = off)
    int wino_N;       // conv_wino.hip (persistent blocks): images of the launch
};

// Output tensors beyond the MALL (256 MB; the 640^2 layers write 0.8-4 GB) are streamed: non-temporal output stores and
// residual loads do not evict the halo tiles and weight fragments the blocks share (-1 % on those layers; on tensors
// that fit, the next layer finds its input in the cache and the plain store is 10 % better: 160^2 x 64 channels)
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_f4(float *p, const float4 v, const bool stream)
{
    if (stream) __builtin_nontemporal_store(f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4 *>(p));
    else *reinterpret_cast<float4 *>(p) = v;
}
__device__ __forceinline__ float4 ld_f4(const float *p, const bool stream)
{
    if (stream) {
        const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p));
        return make_float4(t[0], t[1], t[2], t[3]);
    }
    return *reinterpret_cast<const float4 *>(p);
}

// conv_wino.hip: the Winograd form of the terms-16 3x3 convolution (descriptor terms = 17)
int64_t wino_packed_bytes(int Cout, int Cin);
int wino_pack(const float *weight, void *packed, int Cout, int Cin, float wscale, long stride_o, long stride_i, int flip, int *range_flag,
              hipStream_t stream);
int wino_launch(const ConvArgs &a, int N, hipStream_t stream);
bool wino4_serves(const ConvArgs &a);
int wino4_launch(const ConvArgs &b, int blocks, hipStream_t stream);   // conv_wino4.hip: the four-wave kernel (b: wino_launch's checked arguments)

}  // namespace mrefsr_conv
