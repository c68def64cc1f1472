// What one instruction of each kind costs a wave that is also feeding the matrix pipe (gfx950).  A loop of 1 v_mfma_f32_32x32x16_f16
// + N instructions of one kind, 16 independent accumulators (the 512-register, one-wave-per-SIMD regime of conv_wino2) or 4; one or two
// waves per SIMD.  Output: shader clocks per MFMA against N.   hipcc --offload-arch=gfx950 -O3 -o issue_costs issue_costs.hip && ./issue_costs
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { K_ADD, K_PKADD, K_CVTPK, K_MIXLO, K_MAX3, K_DSR128, K_DMA, K_GLOAD, K_DSW128, K_PKFMA_S };

template <int N, int KIND, int NACC, int THREADS>
__global__ __launch_bounds__(THREADS) void k(float *out, unsigned long long *clk, const float *src, int iters)
{
    extern __shared__ __align__(16) unsigned char smem[];
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    f16x8 x, y;
    for (int e = 0; e < 8; ++e) x[e] = (_Float16)(threadIdx.x * 0.01f + e), y[e] = (_Float16)(e * 0.5f);
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = threadIdx.x + e;
    f32x2 p[8];
    for (int e = 0; e < 8; ++e) p[e] = f32x2{(float)threadIdx.x, (float)e};
    unsigned int h[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    f32x4 q[8];
    for (int e = 0; e < 8; ++e) q[e] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned int lds_a = threadIdx.x * 16;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float *gsrc = src + (size_t)blockIdx.x * 16384 + threadIdx.x * 4;   // 64 KB per block: L2 / MALL hot after the first pass
    const unsigned long long one2 = 0x3f8000003f800000ull;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[a], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < N; ++n) {
                if (KIND == K_ADD) asm volatile("v_add_f32 %0, %1, %2" : "=v"(v[n & 7]) : "v"(v[(n + 1) & 7]), "v"(v[(n + 2) & 7]));
                else if (KIND == K_PKADD) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p[n & 7]) : "v"(p[(n + 1) & 7]), "v"(p[(n + 2) & 7]));
                else if (KIND == K_PKFMA_S) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(p[n & 7]) : "s"(one2), "v"(p[(n + 1) & 7]), "v"(p[(n + 2) & 7]));
                else if (KIND == K_CVTPK) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h[n & 7]) : "v"(v[n & 7]), "v"(v[(n + 1) & 7]));
                else if (KIND == K_MIXLO) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "+v"(h[n & 7]) : "v"(h[(n + 1) & 7]), "v"(v[(n + 3) & 7]));
                else if (KIND == K_MAX3) asm volatile("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(v[n & 7]) : "v"(v[(n + 1) & 7]), "v"(v[(n + 2) & 7]));
                else if (KIND == K_DSR128) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[n & 7]) : "v"(lds_a), "n"((n & 7) * 4096));
                else if (KIND == K_DSW128) asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(lds_a), "v"(q[n & 7]), "n"((n & 7) * 4096) : "memory");
                else if (KIND == K_GLOAD) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(q[n & 7]) : "v"(gsrc), "n"((n & 3) * 1024));
                else if (KIND == K_DMA) {
                    // LDS-DMA: 1 KB per wave-instruction into the wave's own slot of a small ring
                    const unsigned int m0v = __builtin_amdgcn_readfirstlane((unsigned int)(wv * 8192 + (n & 7) * 1024));
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:%2" : : "v"(gsrc), "s"(m0v), "n"((n & 3) * 1024) : "memory");
                }
            }
        }
        if (KIND == K_DSR128 || KIND == K_DSW128) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (KIND == K_GLOAD || KIND == K_DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a)
        for (int e = 0; e < 16; ++e) s += acc[a][e];
    for (int e = 0; e < 8; ++e) s += v[e] + (float)h[e] + p[e][0] + p[e][1] + q[e][0] + q[e][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + smem[threadIdx.x];
    if (threadIdx.x == 0 && blockIdx.x == 0) *clk = t1 - t0;
}

static float *g_out, *g_src;
static unsigned long long *g_clk;

template <int N, int KIND, int NACC, int THREADS>
double run()
{
    unsigned long long h = 0;
    const int iters = 400;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<N, KIND, NACC, THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<N, KIND, NACC, THREADS>), dim3(256), dim3(THREADS), 65536, 0, g_out, g_clk, g_src, iters);
    hipDeviceSynchronize();
    hipMemcpy(&h, g_clk, 8, hipMemcpyDeviceToHost);
    return (double)h / iters / NACC;
}

template <int KIND, int NACC, int THREADS> void sweep(const char *kind)
{
    printf("%-22s waves/SIMD %d acc %2d | N=0 %5.1f  1 %5.1f  2 %5.1f  3 %5.1f  4 %5.1f  6 %5.1f  8 %5.1f  12 %5.1f  16 %5.1f  clk per MFMA\n", kind, THREADS / 256, NACC,
           run<0, KIND, NACC, THREADS>(), run<1, KIND, NACC, THREADS>(), run<2, KIND, NACC, THREADS>(), run<3, KIND, NACC, THREADS>(), run<4, KIND, NACC, THREADS>(),
           run<6, KIND, NACC, THREADS>(), run<8, KIND, NACC, THREADS>(), run<12, KIND, NACC, THREADS>(), run<16, KIND, NACC, THREADS>());
    fflush(stdout);
}

#define BOTH(KIND, name)               \
    sweep<KIND, 16, 256>(name);        \
    sweep<KIND, 4, 512>(name);

int main()
{
    hipMalloc(&g_out, 1024 * 1024 * 4);
    hipMalloc(&g_clk, 8);
    hipMalloc(&g_src, (size_t)256 * 65536 + 65536);
    hipMemset(g_src, 0, (size_t)256 * 65536 + 65536);
    BOTH(K_ADD, "v_add_f32")
    BOTH(K_PKADD, "v_pk_add_f32")
    BOTH(K_PKFMA_S, "v_pk_fma_f32 (sgpr)")
    BOTH(K_CVTPK, "v_cvt_pk_f16_f32")
    BOTH(K_MIXLO, "v_fma_mixlo_f16")
    BOTH(K_MAX3, "v_max3_f32")
    BOTH(K_DSR128, "ds_read_b128")
    BOTH(K_DSW128, "ds_write_b128")
    BOTH(K_GLOAD, "global_load_dwordx4")
    BOTH(K_DMA, "global_load_lds_x4")
    return 0;
}
