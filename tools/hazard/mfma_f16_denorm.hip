// Does v_mfma_f32_32x32x16_f16 honour fp16 subnormal inputs?  (The fp16 pre-filter's error bound, DESIGN 3.1 Step 4, and the
// derived weight plane WH2 of the fp16 convolution mode assume the operands are used as stored, subnormals included.)
//   hipcc --offload-arch=gfx950 -O2 -o tools/hazard/mfma_f16_denorm tools/hazard/mfma_f16_denorm.hip && ./tools/hazard/mfma_f16_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k(float *out)
{
    const _Float16 tiny = (_Float16)9.5367431640625e-07f;   // 2^-20: an fp16 subnormal (smallest normal 2^-14)
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) a[e] = tiny, b[e] = (_Float16)1024.0f;
    f32x16 c;
    for (int e = 0; e < 16; ++e) c[e] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
    f16x8 a2, b2;
    for (int e = 0; e < 8; ++e) a2[e] = (_Float16)1024.0f, b2[e] = tiny;
    f32x16 d;
    for (int e = 0; e < 16; ++e) d[e] = 0.f;
    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b2, d, 0, 0, 0);
    if (threadIdx.x == 0) out[1] = d[0];
}

int main()
{
    float *o, h[2];
    (void)hipMalloc(&o, 8);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o);
    (void)hipMemcpy(h, o, 8, hipMemcpyDeviceToHost);
    const float want = 16 * 9.5367431640625e-07f * 1024.0f;   // 16 products of 2^-20 * 2^10
    printf("subnormal A: %.9g   subnormal B: %.9g   expected %.9g   -> %s\n", h[0], h[1], want,
           (h[0] == want && h[1] == want) ? "subnormal fp16 inputs are honoured" : "SUBNORMAL INPUTS ARE FLUSHED");
    return !(h[0] == want && h[1] == want);
}
