// Error channel + version of the C ABI (include/pyflow_hip.h).
#include <stdio.h>
#include "pyflow_hip.h"

static thread_local char g_err[512];

int pf_set_err(const char* m) {
    snprintf(g_err, sizeof(g_err), "%s", m ? m : "unknown error");
    return -1;
}

extern "C" const char* pf_last_error(void) { return g_err; }
// 4: pf_gemm_desc grew (qk_*); 3: pf_attn_desc grew (workspace, workspace_bytes) and pf_conv_desc grew (gn_stats, gn_C) in round 3 -- a caller built against
// a version-2 header passes shorter structs; check pf_version() == PF_ABI_VERSION (and pf_struct_size) before the first call
extern "C" int pf_version(void) { return PF_ABI_VERSION; }
extern "C" int pf_struct_size(int which) {
    switch (which) {
        case 0: return (int)sizeof(pf_gemm_desc);
        case 1: return (int)sizeof(pf_conv_desc);
        case 2: return (int)sizeof(pf_attn_desc);
        case 3: return (int)sizeof(pf_attn_small_desc);
        default: return -1;
    }
}
