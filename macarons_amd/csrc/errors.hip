// Error plumbing + library identity for the C-ABI (include/macarons_hip.h).
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

namespace mcr {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return 0;
    set_error("%s: HIP error %d (%s)", what, (int)e, hipGetErrorString(e));
    return 2;
}
}  // namespace mcr

extern "C" {
const char* mcr_last_error(void) { return mcr::g_err; }
int mcr_abi_version(void) { return 1; }
const char* mcr_target_arch(void) { return "gfx950"; }
}
