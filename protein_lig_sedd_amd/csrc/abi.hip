// Error plumbing and version of libsedd_hip (see include/sedd_hip.h).
#include "common.h"

static thread_local char g_last_error[512] = "";

void sedd_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}

extern "C" int sedd_abi_version(void) { return SEDD_ABI_VERSION; }
extern "C" const char* sedd_last_error(void) { return g_last_error; }
