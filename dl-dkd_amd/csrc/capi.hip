// Error plumbing + ABI version for libdldkd_hip.so (see include/dldkd_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include "common.hpp"

namespace dldkd {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return DLDKD_ELAUNCH;
    }
    return DLDKD_OK;
}

}  // namespace dldkd

extern "C" {
int dldkd_abi_version(void) { return 1; }
const char* dldkd_last_error(void) { return dldkd::g_err; }
}
