// Shared helpers for the gfx950 kernels of libmrefsr_hip.so (no torch, no CUDA shims).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/mrefsr_hip.h"

#define MREFSR_EXPORT extern "C" __attribute__((visibility("default")))

namespace mrefsr {

char *err_buf();  // thread-local, 512 bytes (defined in api.hip)

inline int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

inline int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MREFSR_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return MREFSR_OK;
}

// "set this kernel's function attribute once" is once PER DEVICE: a process that drives a second GPU needs it there too.
// `done` is a per-kernel bit mask indexed by the current device id (racing first calls both set the attribute: harmless).
inline bool first_use_on_device(unsigned long long &done)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
    const unsigned long long bit = 1ull << dev;
    if (done & bit) return false;
    done |= bit;
    return true;
}

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// corr.hip: exact correlation kernel on the query tiles flagged by the pre-filter (corr_prefilter.hip)
int launch_corr_top1_flagged(const float *y_in, const float *y_ref, const float *inv_ref, const float *nrm_in, int64_t *max_idx,
                             float *max_val, int n_in, int n_pair, int Cp, int h, int w, const int *tile_flag, const int *flag_count,
                             int min_flags, hipStream_t stream);

}  // namespace mrefsr

namespace mrefsr_corr { struct PrefilterOut; }
namespace mrefsr {
// corr_rowstream.hip: row-stationary fp16 pre-filter (pass A of mrefsr_corr_top1_prefilter_f32, Cp = 256)
int launch_corr_prefilter_rs16(const void *yh_in, const void *yh_ref, const float *inv_ref, const float *nrm_in, const float *tau,
                               const mrefsr_corr::PrefilterOut &out, int n_in, int n_pair, int h, int w, float tau_scale, float *dbg,
                               void *scratch, hipStream_t stream);
// bytes of `scratch` (per-lane candidate lists of the exchanged-products kernel; NULL scratch = the previous kernel, lists in LDS)
int64_t corr_prefilter_rs16_scratch_bytes(int n_pair, int h, int w);
int64_t corr_prefilter_rs16_mfma_flop(int h, int w, const char **name);
}  // namespace mrefsr

#define MREFSR_REQUIRE(cond, ...) \
    do { if (!(cond)) return mrefsr::fail(MREFSR_E_INVALID, __VA_ARGS__); } while (0)
