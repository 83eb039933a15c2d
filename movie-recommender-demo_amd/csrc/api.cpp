// libamdrec: error plumbing and version entry points.
#include <stdarg.h>

#include "../../include/amdrec.h"
#include "common.hpp"

namespace amdrec {
thread_local char g_err[512] = "";

int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace amdrec

extern "C" int amdrec_abi_version(void) { return AMDREC_ABI_VERSION; }
extern "C" const char* amdrec_last_error(void) { return amdrec::g_err; }
