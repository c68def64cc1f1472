// LDS-DMA semantics that conv_wino4 relies on (gfx950), checked on the device:
//  (1) global_load_lds_dwordx4 with an instruction offset: does the offset move the LDS destination as well as the global source?
//  (2) buffer_load_dwordx4 ... lds: lanes whose offset lies beyond the descriptor's size -- do they write zeros into LDS, or nothing?
//  (3) the same with an SGPR soffset: is soffset part of the range check?
//  (4) M0 values beyond 64 KB (LDS is 160 KB): does the destination follow M0's upper bits?
//    hipcc --offload-arch=gfx950 -O2 -o lds_dma_semantics lds_dma_semantics.hip && ./lds_dma_semantics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k(const float *src, float *out, int n_bytes)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];   // 40960 floats = 160 KB
    const int lane = threadIdx.x;
    for (int i = lane; i < 40960; i += 64) lds[i] = -1.f;
    __syncthreads();
    // (1) global_load_lds with offset:1024: M0 = 0
    {
        const unsigned int voff = lane * 16;
        const unsigned int m0v = __builtin_amdgcn_readfirstlane(0u);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\ts_waitcnt vmcnt(0)" : : "v"(voff), "s"(src), "s"(m0v) : "memory");
    }
    // (4) M0 = 100 KB
    {
        const unsigned int voff = lane * 16;
        const unsigned int m0v = __builtin_amdgcn_readfirstlane(102400u);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_waitcnt vmcnt(0)" : : "v"(voff), "s"(src), "s"(m0v) : "memory");
    }
    // (2) buffer_load lds, odd lanes out of bounds: M0 = 8192
    {
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, n_bytes, 0x00020000);
        unsigned int voff = lane * 16;
        if (lane & 1) voff = 0xffff0000u;
        const unsigned int m0v = __builtin_amdgcn_readfirstlane(8192u);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds\n\ts_waitcnt vmcnt(0)" : : "v"(voff), "s"(r), "s"(m0v) : "memory");
    }
    // (3) buffer_load lds with soffset = n_bytes - 512: lanes 0..31 end inside, lanes 32..63 beyond the size if soffset counts: M0 = 16384
    {
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, n_bytes, 0x00020000);
        const unsigned int voff = lane * 16;
        const unsigned int m0v = __builtin_amdgcn_readfirstlane(16384u), so = __builtin_amdgcn_readfirstlane((unsigned int)(n_bytes - 512));
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds\n\ts_waitcnt vmcnt(0)" : : "v"(voff), "s"(r), "s"(m0v), "s"(so) : "memory");
    }
    __syncthreads();
    for (int i = lane; i < 40960; i += 64) out[i] = lds[i];
}

int main()
{
    const int n = 4096;   // floats in the source
    std::vector<float> h(n + 1024);
    for (int i = 0; i < n + 1024; ++i) h[i] = 1000.f + i;
    float *src, *out;
    hipMalloc(&src, (n + 1024) * 4);
    hipMalloc(&out, 40960 * 4);
    hipMemcpy(src, h.data(), (n + 1024) * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 163840, 0, src, out, n * 4);
    std::vector<float> o(40960);
    hipMemcpy(o.data(), out, 40960 * 4, hipMemcpyDeviceToHost);
    // (1): expected source floats 256.. (offset 1024 bytes) -- where did they land?
    printf("(1) offset:1024, M0 = 0:   lds[0] = %.0f   lds[256] = %.0f   (source[256] = %.0f; -1 = untouched)\n", o[0], o[256], h[256]);
    printf("    => the instruction offset %s the LDS address\n", o[256] == h[256] && o[0] == -1.f ? "MOVES" : (o[0] == h[256] ? "does NOT move" : "??"));
    printf("(4) M0 = 100 KB:           lds[25600] = %.0f (expected %.0f)   lds[(102400 & 65535) / 4 = 9216] = %.0f\n", o[25600], h[0], o[9216]);
    printf("(2) buffer lds, odd lanes OOB (M0 = 8 KB): lane 0 -> %.0f %.0f (expected %.0f), lane 1 -> %.0f %.0f %.0f %.0f, lane 2 -> %.0f\n", o[2048], o[2049], h[0], o[2052],
           o[2053], o[2054], o[2055], o[2056]);
    printf("(3) soffset = size - 512 (M0 = 16 KB): lane 31 -> %.0f (expected %.0f), lane 32 -> %.0f (source beyond the size: %.0f), lane 63 -> %.0f\n", o[4096 + 31 * 4],
           h[n - 128 + 31 * 4], o[4096 + 32 * 4], h[n - 128 + 32 * 4], o[4096 + 63 * 4]);
    return 0;
}
