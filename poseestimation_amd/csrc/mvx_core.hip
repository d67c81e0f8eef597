// Error reporting, version and the kernel-selection switch of the mvx C ABI.
#include <stdarg.h>

#include <atomic>

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

namespace {
std::atomic<unsigned> g_kernel_selection{0};
}
unsigned kernel_selection() { return g_kernel_selection.load(std::memory_order_relaxed); }

}  // namespace mvx

extern "C" void mvx_set_kernel_selection(unsigned mask) {
    mvx::g_kernel_selection.store(mask, std::memory_order_relaxed);
}

extern "C" int mvx_version(void) { return MVX_VERSION; }

extern "C" const char* mvx_last_error(void) { return mvx::g_error; }
