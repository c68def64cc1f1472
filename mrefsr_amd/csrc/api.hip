// ABI bookkeeping of libmrefsr_hip.so: version + thread-local error string.
#include "common.h"

namespace mrefsr {
char *err_buf()
{
    static thread_local char buf[512] = "";
    return buf;
}
}  // namespace mrefsr

MREFSR_EXPORT int mrefsr_abi_version(void) { return MREFSR_ABI_VERSION; }
MREFSR_EXPORT const char *mrefsr_last_error(void) { return mrefsr::err_buf(); }

// ---- fingerprints of parameter tensors -----------------------------------------------------------------------------------
// The host caches packed (split, re-laid-out) copies of the convolution weights and re-packs them when a parameter's autograd
// version or storage changes.  Writes through `.data` change neither, so the cache is also checked on the device: one launch
// per forward pass sums every cached parameter's words, position-weighted, in 64-bit integer arithmetic (exact and independent
// of the summation order, unlike a floating-point norm) and compares the sums with those taken when the copies were packed.
namespace {
__global__ __launch_bounds__(256) void weights_checksum_kernel(const long long *__restrict__ table, unsigned long long *__restrict__ sums,
                                                               unsigned int *__restrict__ done, const unsigned long long *__restrict__ ref,
                                                               int *__restrict__ flag, int flag_bits)
{
    const int t = blockIdx.y;
    const unsigned int *w = reinterpret_cast<const unsigned int *>(table[2 * t]);
    const long long n = table[2 * t + 1];
    unsigned long long s = 0;
    const long long n4 = ((reinterpret_cast<size_t>(w) & 15) == 0) ? n >> 2 : 0;   // 16-byte pieces, then the scalar tail
    typedef unsigned int u32x4c __attribute__((ext_vector_type(4)));
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const u32x4c v = reinterpret_cast<const u32x4c *>(w)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) s += (unsigned long long)v[k] * (unsigned long long)(2 * (4 * i + k) + 1);
    }
    for (long long i = 4 * n4 + blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        s += (unsigned long long)w[i] * (unsigned long long)(2 * i + 1);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    __shared__ unsigned long long part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&sums[t], part[0] + part[1] + part[2] + part[3]);
        __threadfence();
        if (atomicAdd(&done[t], 1u) == gridDim.x - 1 && ref) {   // the last block of this tensor: the sum is complete
            const unsigned long long total = atomicAdd(&sums[t], 0ull);
            if (total != ref[t]) atomicOr(flag, flag_bits);
        }
    }
}
}  // namespace

MREFSR_EXPORT int mrefsr_weights_checksum(const int64_t *table, int n, uint64_t *sums, uint32_t *done, const uint64_t *ref, int *flag,
                                          int flag_bits, mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(table && sums && done && n > 0, "weights_checksum: null pointer / n=%d", n);
    MREFSR_REQUIRE(!ref || flag, "weights_checksum: a reference needs a flag to report into");
    MREFSR_REQUIRE(n <= 65535, "weights_checksum: n=%d exceeds the grid limit", n);
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(sums, 0, (size_t)n * 8, st) != hipSuccess || hipMemsetAsync(done, 0, (size_t)n * 4, st) != hipSuccess)
        return mrefsr::check_launch("weights_checksum(memset)");
    hipLaunchKernelGGL(weights_checksum_kernel, dim3(48, n), dim3(256), 0, st, (const long long *)table, (unsigned long long *)sums, done,
                       (const unsigned long long *)ref, flag, flag_bits);
    return mrefsr::check_launch("weights_checksum");
}
