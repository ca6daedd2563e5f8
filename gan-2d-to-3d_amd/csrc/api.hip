// api.hip — ABI version, thread-local error string.
#include "g2s_common.h"
#include <atomic>

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
static std::atomic<int> g_deterministic{0};
bool deterministic() { return g_deterministic.load(std::memory_order_relaxed) != 0; }
static thread_local int t_precleared = 0;
bool precleared() { return t_precleared != 0; }
}  // namespace g2s

extern "C" int g2s_set_precleared(int on) {
    const int prev = g2s::t_precleared;
    g2s::t_precleared = on ? 1 : 0;
    return prev;
}

extern "C" int g2s_set_deterministic(int on) {
    g2s::g_deterministic.store(on ? 1 : 0, std::memory_order_relaxed);
    return G2S_OK;
}
extern "C" int g2s_get_deterministic(void) { return g2s::deterministic() ? 1 : 0; }

extern "C" int g2s_abi_version(void) { return G2S_ABI_VERSION; }
extern "C" const char *g2s_last_error(void) { return g2s::error_buf(); }
