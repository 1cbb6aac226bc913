// lec_common.hip -- version and error plumbing of the C ABI (include/lec_hip.h).
#include <stdio.h>
#include <string.h>

#include "../../include/lec_hip.h"
#include "lec_internal.h"

static thread_local char g_err[512] = "";

int lec_set_error(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg ? msg : "");
    return code;
}

extern "C" int lec_version(void) { return LEC_ABI_VERSION; }

extern "C" const char* lec_last_error(void) { return g_err; }
