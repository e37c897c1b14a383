#include <stdarg.h>
#include "shg_common.h"

namespace shg {
static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}
}  // namespace shg

extern "C" int shg_abi_version(void) { return SHG_ABI_VERSION; }
extern "C" const char* shg_last_error_string(void) { return shg::g_error; }
