// libmarlhip.so: version / error text / device probe.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace marl {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace marl

extern "C" int marlhip_version(void) { return MARLHIP_VERSION; }
extern "C" const char* marlhip_last_error(void) { return marl::g_err; }
extern "C" int marlhip_device_available(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n > 0 ? 1 : 0;
}
