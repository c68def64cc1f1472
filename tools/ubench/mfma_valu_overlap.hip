// Does VALU work hide in the shadow of MFMAs on gfx950?  One wave (or two) per SIMD runs a loop of 1 v_mfma_f32_32x32x16_f16 followed
// by N independent VALU instructions (v_fma_f32 / v_cvt_pk_f16_f32 / v_fma_mix_f32); cycles per iteration against N tell how many
// VALU slots an 8-pass MFMA covers.   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int N, int KIND, int NACC>
__global__ void k(float *out, unsigned long long *clk, int iters)
{
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    f16x8 x, y;
    for (int e = 0; e < 8; ++e) x[e] = (_Float16)(threadIdx.x * 0.01f + e), y[e] = (_Float16)(e * 0.5f);
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = threadIdx.x + e;
    unsigned int h[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[a], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < N; ++n) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[n & 7]) : "v"(v[(n + 1) & 7]), "v"(v[(n + 2) & 7]));
                else if (KIND == 1) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h[n & 7]) : "v"(v[n & 7]), "v"(v[(n + 1) & 7]));
                else if (KIND == 2) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(v[n & 7]) : "v"(h[n & 7]), "v"(v[(n + 3) & 7]));
                else asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(h[n & 7]) : "v"(h[(n + 1) & 7]), "v"(h[(n + 2) & 7]));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a)
        for (int e = 0; e < 16; ++e) s += acc[a][e];
    for (int e = 0; e < 8; ++e) s += v[e] + (float)h[e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *clk = t1 - t0;
}

template <int N, int KIND, int NACC>
void run(int threads, const char *kind)
{
    float *out;
    unsigned long long *clk, h = 0;
    hipMalloc(&out, 1024 * 1024 * 4);
    hipMalloc(&clk, 8);
    const int iters = 2000;
    hipLaunchKernelGGL((k<N, KIND, NACC>), dim3(256), dim3(threads), 0, 0, out, clk, iters);
    hipLaunchKernelGGL((k<N, KIND, NACC>), dim3(256), dim3(threads), 0, 0, out, clk, iters);
    hipDeviceSynchronize();
    hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    printf("%-18s waves/SIMD %d  chains %d  VALU per MFMA %2d : %6.1f clk per MFMA\n", kind, threads / 256, NACC, N, (double)h / iters / NACC);
    hipFree(out);
    hipFree(clk);
}

template <int KIND> void sweep(const char *kind)
{
    run<0, KIND, 4>(256, kind), run<2, KIND, 4>(256, kind), run<4, KIND, 4>(256, kind), run<6, KIND, 4>(256, kind), run<8, KIND, 4>(256, kind),
        run<12, KIND, 4>(256, kind), run<16, KIND, 4>(256, kind);
    run<0, KIND, 4>(512, kind), run<4, KIND, 4>(512, kind), run<8, KIND, 4>(512, kind), run<16, KIND, 4>(512, kind);
}

int main()
{
    sweep<0>("v_fma_f32");
    sweep<1>("v_cvt_pk_f16_f32");
    sweep<2>("v_fma_mix_f32");
    sweep<3>("v_pk_mul_f16");
    run<0, 0, 1>(256, "dependent chain"), run<6, 0, 1>(256, "dependent chain"), run<0, 0, 2>(256, "two chains"), run<6, 0, 2>(256, "two chains");
    return 0;
}
