// ABI version + thread-local error text for libssi_hip.so.
#include <stdarg.h>
#include "common_hip.h"

static thread_local char g_err[512] = "";

void ssi_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int ssi_abi_version(void) { return SSI_ABI_VERSION; }
extern "C" const char* ssi_last_error(void) { return g_err; }
