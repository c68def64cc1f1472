// feature_match_index for ANY patch size / strides / map sizes (ref_map_util.py:26-86) -- the general form of the entry the
// path calls with patch_size = 3, stride 1 (every shipped yml; csrc/corr*.hip are the kernels for that case).  Same defined
// fp32 operation order as the 3x3 kernels and oracle/mrefsr_oracle.c (built with -ffp-contract=off):
//   g_t   = fmaf chain over channels ascending of in[c, q + t] * ref[c, r + t]            for each tap t of the p x p window
//   raw   = g_0 + g_1 + ... in row-major tap order (sequential fp32 adds)
//   corr  = raw * inv[r],  inv[r] = 1 / (sqrt(sum of the window's per-pixel sums of squares, row-major) + 1e-5)   (is_norm)
//   best  = max over r, lowest r on exact ties;  max_val = best / (||in patch|| + 1e-5)                          (norm_input)
// A block owns QT query patches (their p*p*C values staged in LDS, read as broadcasts); its 256 threads walk the
// reference patches (lanes along rx: coalesced), QT running sums per tap in registers.  VALU-bound by design: O(n_q n_r p^2 C)
// scalar FMAs -- a correct general entry, not the benchmark's path.
#include "common.h"

namespace {

__global__ void fmi_sumsq_kernel(const float *__restrict__ x, float *__restrict__ n2, int C, int HW)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    float s = 0.f;
    for (int c = 0; c < C; ++c) {
        const float v = x[(size_t)c * HW + p];
        s = fmaf(v, v, s);
    }
    n2[p] = s;
}

__global__ void fmi_patch_norm_kernel(const float *__restrict__ n2, float *__restrict__ nrm_eps, float *__restrict__ inv, int w,
                                      int P, int stride, int npx, int np)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= np) return;
    const int py = i / npx, px = i - py * npx;
    const float *m = n2 + (size_t)(py * stride) * w + px * stride;
    float s = m[0];
    for (int t = 1; t < P * P; ++t) s = s + m[(t / P) * w + (t % P)];
    const float ne = __builtin_sqrtf(s) + 1e-5f;
    if (nrm_eps) nrm_eps[i] = ne;
    if (inv) inv[i] = 1.0f / ne;
}

template <int QT>
__global__ __launch_bounds__(256) void fmi_match_kernel(const float *__restrict__ fin, const float *__restrict__ fref,
                                                        const float *__restrict__ inv_ref, const float *__restrict__ nrm_in,
                                                        int64_t *__restrict__ max_idx, float *__restrict__ max_val, int C, int h, int w,
                                                        int hr, int wr, int P, int si, int sr, int nqx, int nq, int nrx, int nr)
{
    extern __shared__ float qs[];   // [QT][P*P][C]
    __shared__ float red_v[256];
    __shared__ int red_i[256];
    const int tid = threadIdx.x, PP = P * P, q0 = blockIdx.x * QT;
    const size_t HW = (size_t)h * w, HWr = (size_t)hr * wr;
    for (int e = tid; e < QT * PP * C; e += 256) {
        const int c = e % C, t = (e / C) % PP, qi = e / (C * PP);
        const int q = q0 + qi < nq ? q0 + qi : nq - 1;
        const int qy = q / nqx, qx = q - qy * nqx;
        qs[e] = fin[(size_t)c * HW + (size_t)(qy * si + t / P) * w + qx * si + t % P];
    }
    __syncthreads();
    float best[QT];
    int bidx[QT];
#pragma unroll
    for (int qi = 0; qi < QT; ++qi) best[qi] = -__builtin_inff(), bidx[qi] = 0x7fffffff;
    for (int r = tid; r < nr; r += 256) {
        const int ry = r / nrx, rx = r - ry * nrx;
        const float *rb = fref + (size_t)(ry * sr) * wr + rx * sr;
        float v[QT];
        for (int t = 0; t < PP; ++t) {
            const float *rp = rb + (size_t)(t / P) * wr + t % P;
            const float *qp = qs + (size_t)t * C;
            float g[QT];
#pragma unroll
            for (int qi = 0; qi < QT; ++qi) g[qi] = 0.f;
#pragma unroll 4
            for (int c = 0; c < C; ++c) {
                const float rv = rp[(size_t)c * HWr];
#pragma unroll
                for (int qi = 0; qi < QT; ++qi) g[qi] = fmaf(qp[(size_t)qi * PP * C + c], rv, g[qi]);
            }
#pragma unroll
            for (int qi = 0; qi < QT; ++qi) v[qi] = t == 0 ? g[qi] : v[qi] + g[qi];
        }
        const float iv = inv_ref ? inv_ref[r] : 1.0f;
#pragma unroll
        for (int qi = 0; qi < QT; ++qi) {
            const float cv = v[qi] * iv;
            if (cv > best[qi]) best[qi] = cv, bidx[qi] = r;   // r ascends within a thread: '>' keeps the lowest index
        }
    }
#pragma unroll
    for (int qi = 0; qi < QT; ++qi) {
        __syncthreads();
        red_v[tid] = best[qi];
        red_i[tid] = bidx[qi];
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) {
                const float a = red_v[tid], b = red_v[tid + s];
                const int ia = red_i[tid], ib = red_i[tid + s];
                if (b > a || (b == a && ib < ia)) red_v[tid] = b, red_i[tid] = ib;
            }
            __syncthreads();
        }
        if (tid == 0 && q0 + qi < nq) {
            max_idx[q0 + qi] = red_i[0];
            if (max_val) max_val[q0 + qi] = nrm_in ? red_v[0] / nrm_in[q0 + qi] : red_v[0];
        }
    }
}

}  // namespace

MREFSR_EXPORT int64_t mrefsr_feature_match_index_workspace_bytes(int h, int w, int hr, int wr)
{
    if (h <= 0 || w <= 0 || hr <= 0 || wr <= 0) return -1;
    return ((int64_t)h * w * 2 + (int64_t)hr * wr * 2) * 4;   // n2_in, nrm_in (<= h*w), n2_ref, inv_ref (<= hr*wr)
}

MREFSR_EXPORT int mrefsr_feature_match_index_f32(const float *feat_in, const float *feat_ref, int C, int h, int w, int hr, int wr, int patch,
                                                 int stride_in, int stride_ref, int is_norm, int norm_input, int64_t *max_idx, float *max_val,
                                                 void *workspace, int64_t workspace_bytes, mrefsr_stream_t stream)
{
    MREFSR_REQUIRE(feat_in && feat_ref && max_idx && workspace, "feature_match_index: null pointer");
    MREFSR_REQUIRE(C > 0 && patch > 0 && stride_in > 0 && stride_ref > 0, "feature_match_index: C=%d patch=%d strides %d / %d", C, patch, stride_in,
                   stride_ref);
    MREFSR_REQUIRE(h >= patch && w >= patch && hr >= patch && wr >= patch, "feature_match_index: maps %dx%d / %dx%d smaller than the %d-patch", h, w,
                   hr, wr, patch);
    MREFSR_REQUIRE(workspace_bytes >= mrefsr_feature_match_index_workspace_bytes(h, w, hr, wr), "feature_match_index: workspace too small");
    const int nqy = (h - patch) / stride_in + 1, nqx = (w - patch) / stride_in + 1, nq = nqy * nqx;
    const int nry = (hr - patch) / stride_ref + 1, nrx = (wr - patch) / stride_ref + 1, nr = nry * nrx;
    hipStream_t st = (hipStream_t)stream;
    float *n2_in = reinterpret_cast<float *>(workspace), *nrm_in = n2_in + (size_t)h * w, *n2_ref = nrm_in + (size_t)h * w,
          *inv_ref = n2_ref + (size_t)hr * wr;
    if (is_norm) {
        hipLaunchKernelGGL(fmi_sumsq_kernel, dim3((hr * wr + 255) / 256), dim3(256), 0, st, feat_ref, n2_ref, C, hr * wr);
        hipLaunchKernelGGL(fmi_patch_norm_kernel, dim3((nr + 255) / 256), dim3(256), 0, st, n2_ref, (float *)nullptr, inv_ref, wr, patch, stride_ref, nrx,
                           nr);
    }
    if (norm_input && max_val) {
        hipLaunchKernelGGL(fmi_sumsq_kernel, dim3((h * w + 255) / 256), dim3(256), 0, st, feat_in, n2_in, C, h * w);
        hipLaunchKernelGGL(fmi_patch_norm_kernel, dim3((nq + 255) / 256), dim3(256), 0, st, n2_in, nrm_in, (float *)nullptr, w, patch, stride_in, nqx, nq);
    }
    const size_t per_q = (size_t)patch * patch * C * sizeof(float);
    const float *iv = is_norm ? inv_ref : nullptr, *nq_ = (norm_input && max_val) ? nrm_in : nullptr;
#define MREFSR_FMI_LAUNCH(QT)                                                                                                                    \
    {                                                                                                                                            \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fmi_match_kernel<QT>), hipFuncAttributeMaxDynamicSharedMemorySize,               \
                                  (int)(per_q * QT));                                                                                            \
        hipLaunchKernelGGL(fmi_match_kernel<QT>, dim3((nq + QT - 1) / QT), dim3(256), per_q * QT, st, feat_in, feat_ref, iv, nq_, max_idx, max_val, C, \
                           h, w, hr, wr, patch, stride_in, stride_ref, nqx, nq, nrx, nr);                                                         \
    }
    if (per_q * 4 <= (size_t)96 * 1024) MREFSR_FMI_LAUNCH(4)
    else if (per_q * 2 <= (size_t)96 * 1024) MREFSR_FMI_LAUNCH(2)
    else if (per_q <= (size_t)150 * 1024) MREFSR_FMI_LAUNCH(1)
    else return mrefsr::fail(MREFSR_E_UNSUPPORTED, "feature_match_index: patch %d x %d x %d channels does not fit the LDS", patch, patch, C);
#undef MREFSR_FMI_LAUNCH
    return mrefsr::check_launch("feature_match_index");
}
