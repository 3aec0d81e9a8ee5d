// Host-side glue of libstgcma_hip.so: version + thread-local error string (see include/stgcma.h).
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include "../../include/stgcma.h"

static thread_local char g_err[512] = "";

void stg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int stg_version(void) { return STG_VERSION; }
extern "C" const char* stg_last_error(void) { return g_err; }

// ---- dispatch options (declared in common.h)
std::atomic<int> stg_opt_gemm_8ph{1}, stg_opt_gemm_8phm{1}, stg_opt_gemm_nx{1}, stg_opt_gemm_d8m{2}, stg_opt_gemm_dbg{0}, stg_opt_xattn{1}, stg_opt_wgrad_plan{2}, stg_opt_upln_cap{256};

extern "C" int stg_set_option(const char* name, int value) {
    if (name == nullptr) { stg_set_error("stg_set_option: null name"); return -1; }
    if (!strcmp(name, "gemm_8ph")) stg_opt_gemm_8ph = value;
    else if (!strcmp(name, "gemm_8phm")) stg_opt_gemm_8phm = value;
    else if (!strcmp(name, "gemm_nx")) stg_opt_gemm_nx = value;
    else if (!strcmp(name, "gemm_d8m")) stg_opt_gemm_d8m = value;
    else if (!strcmp(name, "gemm_dbg")) stg_opt_gemm_dbg = value;     // read by the diagnostics build only
    else if (!strcmp(name, "xattn")) stg_opt_xattn = value;
    else if (!strcmp(name, "wgrad_plan")) stg_opt_wgrad_plan = value;
    else if (!strcmp(name, "upln_cap")) stg_opt_upln_cap = value < 2 ? 2 : value;
    else { stg_set_error("stg_set_option: unknown option '%s'", name); return -2; }
    return 0;
}
