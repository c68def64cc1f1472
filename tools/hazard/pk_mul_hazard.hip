// Stand-alone check of the gfx950 behaviour described in DESIGN.md 3.2: packed fp32 VALU instructions with
// swizzled source halves, e.g.
//     v_pk_mul_f32 vD, vA, vB op_sel:[0,1] op_sel_hi:[1,0]      (lo = A.lo * B.hi, hi = A.hi * B.lo)
// compared, lane by lane, with the scalar products of the same inputs while the other waves of the CU (this
// workgroup's and co-resident workgroups') issue MFMAs.
//   hipcc --offload-arch=gfx950 -O3 -o tools/hazard/pk_mul_hazard tools/hazard/pk_mul_hazard.hip && ./tools/hazard/pk_mul_hazard
// Prints the number of mismatching results per quarter-wave for each (instruction form, neighbour) pair.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define SCALAR(op, d, x, y) asm volatile(op " %0, %1, %2" : "=&v"(d) : "v"(x), "v"(y))

// FORM: which packed instruction; NEIGH: 0 = bf16 MFMAs in the other waves, 1 = no MFMA, 2 = fp16 MFMAs, 3 = fp32 MFMAs
template <int FORM, int NEIGH>
__global__ __launch_bounds__(256) void probe(unsigned long long *bad, float *sink, int iters, float sc)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    bf16x8 fa, fb;
    f16x8 ha, hb;
    for (int e = 0; e < 8; ++e) {
        fa[e] = (__bf16)(0.001f * (lane + e)), fb[e] = (__bf16)(0.002f * (lane - e));
        ha[e] = (_Float16)(0.001f * (lane + e)), hb[e] = (_Float16)(0.002f * (lane - e));
    }
    unsigned int seed = blockIdx.x * 2654435761u + threadIdx.x * 40503u + 17u;
    unsigned long long nbad = 0;
    for (int it = 0; it < iters; ++it) {
        // a burst of MFMAs whose length differs per wave and iteration: the waves of a SIMD drift out of phase
        const int n = 4 + ((it * 7 + wv * 3 + blockIdx.x) % 13);
        for (int k = 0; k < n; ++k) {
            if (NEIGH == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
            if (NEIGH == 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc, 0, 0, 0);
            if (NEIGH == 3) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(0.001f * lane, 0.002f * lane, acc, 0, 0, 0);
        }
        // a stretch of VALU work containing the packed instruction
        for (int k = 0; k < 24; ++k) {
            seed = seed * 1664525u + 1013904223u;
            const float a0 = (float)(seed & 0xffff) * (1.0f / 65536.f), a1 = (float)(seed >> 16) * (1.0f / 65536.f);
            const float b0 = 1.f - a0, b1 = 1.f - a1;
            f32x2 pa = {a0, a1}, pb = {b0, b1}, pc = {a1, b0}, pr;
            float s0, s1;
            if (FORM == 0) {  // crossed multiply
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(pr) : "v"(pa), "v"(pb));
                SCALAR("v_mul_f32", s0, a0, b1);
                SCALAR("v_mul_f32", s1, a1, b0);
            } else if (FORM == 1) {  // plain multiply
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=&v"(pr) : "v"(pa), "v"(pb));
                SCALAR("v_mul_f32", s0, a0, b0);
                SCALAR("v_mul_f32", s1, a1, b1);
            } else if (FORM == 2) {  // src1.lo broadcast
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=&v"(pr) : "v"(pa), "v"(pb));
                SCALAR("v_mul_f32", s0, a0, b0);
                SCALAR("v_mul_f32", s1, a1, b0);
            } else if (FORM == 3) {  // both sources swapped
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0]" : "=&v"(pr) : "v"(pa), "v"(pb));
                SCALAR("v_mul_f32", s0, a1, b1);
                SCALAR("v_mul_f32", s1, a0, b0);
            } else if (FORM == 4) {  // src1.hi broadcast
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=&v"(pr) : "v"(pa), "v"(pb));
                SCALAR("v_mul_f32", s0, a0, b1);
                SCALAR("v_mul_f32", s1, a1, b1);
            } else if (FORM == 5) {  // crossed add
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(pr) : "v"(pa), "v"(pb));
                SCALAR("v_add_f32", s0, a0, b1);
                SCALAR("v_add_f32", s1, a1, b0);
            } else if (FORM == 6) {  // crossed fma (sources 0 / 1 crossed, addend plain)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=&v"(pr) : "v"(pa), "v"(pb), "v"(pc));
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(s0) : "v"(a0), "v"(b1), "v"(a1));
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(s1) : "v"(a1), "v"(b0), "v"(b0));
            } else if (FORM == 7) {  // scalar-register source broadcast (the conv epilogue's form)
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=&v"(pr) : "v"(pa), "s"(f32x2{sc, 0.f}));
                SCALAR("v_mul_f32", s0, a0, sc);
                SCALAR("v_mul_f32", s1, a1, sc);
            } else if (FORM == 9) {  // mixed-precision fma reading the HIGH fp16 half of a packed pair (and the low half)
                unsigned int ph;
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=&v"(ph) : "v"(a0), "v"(a1));
                const float m0 = b0 * 2048.f, m1 = b1 * 2048.f, c = -2048.f;
                asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=&v"(pr[0]) : "v"(ph), "v"(c), "v"(m0));
                asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=&v"(pr[1]) : "v"(ph), "v"(c), "v"(m1));
                unsigned int hi16;
                float f0, f1;
                asm volatile("v_cvt_f32_f16 %0, %1" : "=&v"(f0) : "v"(ph));
                asm volatile("v_lshrrev_b32 %0, 16, %1" : "=&v"(hi16) : "v"(ph));
                asm volatile("v_cvt_f32_f16 %0, %1" : "=&v"(f1) : "v"(hi16));
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(s0) : "v"(f0), "v"(c), "v"(m0));
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(s1) : "v"(f1), "v"(c), "v"(m1));
            } else if (FORM == 10) {  // plain packed fp16 multiply (the DCN kernel derives WH2 = wh * 2^-11 with it)
                unsigned int ph, pq, prod, e0, e1;
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=&v"(ph) : "v"(a0), "v"(a1));
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=&v"(pq) : "v"(b0), "v"(b1));
                asm volatile("v_pk_mul_f16 %0, %1, %2" : "=&v"(prod) : "v"(ph), "v"(pq));
                asm volatile("v_mul_f16 %0, %1, %2" : "=&v"(e0) : "v"(ph), "v"(pq));
                asm volatile("v_lshrrev_b32 %0, 16, %1" : "=&v"(e1) : "v"(ph));
                unsigned int q1;
                asm volatile("v_lshrrev_b32 %0, 16, %1" : "=&v"(q1) : "v"(pq));
                asm volatile("v_mul_f16 %0, %1, %2" : "=&v"(e1) : "v"(e1), "v"(q1));
                pr[0] = __uint_as_float(prod & 0xffffu), pr[1] = __uint_as_float(prod >> 16);
                s0 = __uint_as_float(e0 & 0xffffu), s1 = __uint_as_float(e1 & 0xffffu);
            } else if (FORM == 11) {  // packed move with crossed halves (what the compiler forms for f32x2{a.hi, b.lo}): lo = src0.hi, hi = src1.lo
                asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=&v"(pr) : "v"(pa), "v"(pb));
                asm volatile("v_mov_b32 %0, %1" : "=&v"(s0) : "v"(a1));
                asm volatile("v_mov_b32 %0, %1" : "=&v"(s1) : "v"(b0));
            } else {  // the other cross
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=&v"(pr) : "v"(pa), "v"(pb));
                SCALAR("v_mul_f32", s0, a1, b0);
                SCALAR("v_mul_f32", s1, a0, b1);
            }
            nbad += (__float_as_uint(pr[0]) != __float_as_uint(s0)) + (__float_as_uint(pr[1]) != __float_as_uint(s1));
        }
    }
    if (nbad) atomicAdd(&bad[lane >> 4], nbad);
    float t = 0.f;
    for (int e = 0; e < 16; ++e) t += acc[e];
    if (t == 12345.678f) sink[0] = t;  // keeps the MFMAs alive
}

static const char *FORMS[] = {"v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0] (crossed)", "v_pk_mul_f32 (plain)", "v_pk_mul_f32 op_sel_hi:[1,0] (src1.lo twice)",
                              "v_pk_mul_f32 op_sel:[1,1] op_sel_hi:[0,0] (swapped)", "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,1] (src1.hi twice)",
                              "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0] (crossed)", "v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[1,0,1] (crossed)",
                              "v_pk_mul_f32 v, v, s op_sel_hi:[1,0] (scalar source)", "v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[0,1] (crossed, src0)",
                              "v_fma_mix_f32 (fp16 lo / hi half of a packed pair)", "v_pk_mul_f16 (plain)",
                              "v_pk_mov_b32 op_sel:[1,0] (lo = src0.hi, hi = src1.lo)"};
static const char *NEIGHS[] = {"bf16 MFMA", "no MFMA", "fp16 MFMA", "fp32 MFMA"};

template <int FORM, int NEIGH> static void run(int blocks, int iters)
{
    unsigned long long *bad;
    float *sink;
    (void)hipMalloc(&bad, 4 * sizeof(*bad));
    (void)hipMalloc(&sink, 4);
    (void)hipMemset(bad, 0, 4 * sizeof(*bad));
    hipLaunchKernelGGL((probe<FORM, NEIGH>), dim3(blocks), dim3(256), 0, 0, bad, sink, iters, 0.37f);
    hipError_t e = hipDeviceSynchronize();
    unsigned long long h[4];
    (void)hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-58s | %-9s | %s | mismatches by quarter-wave: %llu %llu %llu %llu of %.3g\n", FORMS[FORM], NEIGHS[NEIGH], hipGetErrorString(e), h[0],
           h[1], h[2], h[3], (double)blocks * 256 * iters * 24 * 2);
    (void)hipFree(bad);
    (void)hipFree(sink);
}

int main(int argc, char **argv)
{
    const int blocks = argc > 1 ? atoi(argv[1]) : 256 * 8 * 4, iters = argc > 2 ? atoi(argv[2]) : 200;
    run<0, 0>(blocks, iters);
    run<0, 1>(blocks, iters);
    run<0, 2>(blocks, iters);
    run<0, 3>(blocks, iters);
    run<1, 0>(blocks, iters);
    run<2, 0>(blocks, iters);
    run<3, 0>(blocks, iters);
    run<4, 0>(blocks, iters);
    run<5, 0>(blocks, iters);
    run<6, 0>(blocks, iters);
    run<7, 0>(blocks, iters);
    run<7, 2>(blocks, iters);
    run<8, 0>(blocks, iters);
    run<9, 0>(blocks, iters);
    run<9, 2>(blocks, iters);
    run<10, 0>(blocks, iters);
    run<10, 2>(blocks, iters);
    run<11, 0>(blocks, iters);
    run<11, 2>(blocks, iters);
    return 0;
}
