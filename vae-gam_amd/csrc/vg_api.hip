// vg_api.hip -- library identity and error reporting for the C ABI (include/vaegam.h).
#include "vg_common.h"
#include "../../include/vaegam.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void vg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int vg_check_launch(const char* what) {
    auto e = hipGetLastError();
    if (e != 0) { vg_set_error("%s: %s", what, hipGetErrorString(e)); return VG_ERR_LAUNCH; }
    return VG_OK;
}

extern "C" int vg_version(void) { return 100; }          // 0.1.0
extern "C" const char* vg_last_error(void) { return g_err; }
