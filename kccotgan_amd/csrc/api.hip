// Library-level entry points: version and the thread-local error string.
#include "common.h"
#include <string.h>

namespace kccot {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
}  // namespace kccot

extern "C" int kccot_version(void) { return KCCOT_VERSION; }
extern "C" const char* kccot_last_error(void) { return kccot::g_err; }
