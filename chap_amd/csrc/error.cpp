#include <stdarg.h>
#include <stdio.h>
#include "chap_hip.h"

static thread_local char g_err[512] = "";

void chap_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* chap_last_error(void) { return g_err; }
extern "C" int chap_abi_version(void) { return CHAP_ABI_VERSION; }
