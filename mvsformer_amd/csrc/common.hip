// Error channel + ABI version of libmvs_hip.so.
#include "common.h"

namespace mvs {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace mvs

extern "C" int mvs_version(void) { return MVS_ABI_VERSION; }
extern "C" const char* mvs_last_error(void) { return mvs::g_err; }
