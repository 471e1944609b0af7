// Error reporting and version of libmic_hip.
#include "mic_common.h"

static thread_local char g_err[512] = "";

void mic_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* mic_last_error(void) { return g_err; }
extern "C" int mic_version(void) { return 1; }
