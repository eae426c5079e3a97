#include <cstdarg>
#include <cstdio>

#include "ibl_common.h"

static thread_local char g_err[512] = "";

int ibl_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* ibl_last_error(void) { return g_err; }
extern "C" int ibl_version(void) { return 100; }
