// Error channel + version of the C ABI (include/pyflow_hip.h).
#include <stdio.h>
#include "pyflow_hip.h"

static thread_local char g_err[512];

int pf_set_err(const char* m) {
    snprintf(g_err, sizeof(g_err), "%s", m ? m : "unknown error");
    return -1;
}

extern "C" const char* pf_last_error(void) { return g_err; }
extern "C" int pf_version(void) { return 1; }
