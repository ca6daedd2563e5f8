// api.hip — ABI version, thread-local error string.
#include "g2s_common.h"

namespace g2s {
static thread_local char g_err[512] = "";
char *error_buf() { return g_err; }
int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace g2s

extern "C" int g2s_abi_version(void) { return G2S_ABI_VERSION; }
extern "C" const char *g2s_last_error(void) { return g2s::error_buf(); }
