// Host-side glue of libstgcma_hip.so: version + thread-local error string (see include/stgcma.h).
#include <cstdarg>
#include <cstdio>
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
