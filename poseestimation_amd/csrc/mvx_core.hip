// Error reporting and version of the mvx C ABI.
#include <stdarg.h>

#include "mvx_common.h"

namespace mvx {

namespace {
thread_local char g_error[512] = "";
}

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

}  // namespace mvx

extern "C" int mvx_version(void) { return MVX_VERSION; }

extern "C" const char* mvx_last_error(void) { return mvx::g_error; }
