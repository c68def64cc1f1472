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
