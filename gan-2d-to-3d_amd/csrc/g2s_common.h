// g2s_common.h — shared helpers of libg2s.so (error string, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "../../include/g2s.h"

namespace g2s {

char *error_buf();                     // thread-local, defined in api.hip
int fail(int code, const char *fmt, ...);

// g2s_set_deterministic (api.hip): launches choose partitions whose fp32 sums have a fixed order.
bool deterministic();

// g2s_set_precleared (api.hip, per thread): the caller guarantees that the accumulators a function would clear
// itself (small gradient sums, fixed-point workspaces, scatter targets) are already zero: their memsets are skipped.
bool precleared();

inline hipStream_t as_stream(g2s_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Checks the launch (not the execution: no synchronisation inside the library).
inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(G2S_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return G2S_OK;
}

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace g2s

#define G2S_REQUIRE(cond, ...) \
    do { if (!(cond)) return g2s::fail(G2S_ERR_INVALID, __VA_ARGS__); } while (0)
