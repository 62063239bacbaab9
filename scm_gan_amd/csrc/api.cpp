// Error slot, version and HIP error translation for libscmgan_hip.so.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void scm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int scm_check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return SCM_OK;
    scm_set_error("%s: %s", what, hipGetErrorString(e));
    return SCM_EHIP;
}

extern "C" int scm_version(void) { return 100; }
extern "C" const char* scm_last_error(void) { return g_err; }
