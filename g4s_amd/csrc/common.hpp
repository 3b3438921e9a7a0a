// common.hpp — shared host-side helpers of libg4s_hip.so (error reporting, HIP call checking, stream cast).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "g4s.h"

#define G4S_API extern "C" __attribute__((visibility("default")))

namespace g4s {

// Thread-local message behind g4s_last_error().
char *last_error_buf();
int set_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// 8 XCDs, each with its own L2: block b and b+8 share one (MI355X_MICROARCH.md, Workgroup dispatch).
constexpr int kXcds = 8;

} // namespace g4s

#define G4S_HIP_TRY(expr)                                                                            \
    do {                                                                                             \
        hipError_t e__ = (expr);                                                                     \
        if (e__ != hipSuccess)                                                                       \
            return g4s::set_error(G4S_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                                  __FILE__, __LINE__);                                               \
    } while (0)

#define G4S_TRY(expr)                      \
    do {                                   \
        int s__ = (expr);                  \
        if (s__ != G4S_OK) return s__;     \
    } while (0)

#define G4S_REQUIRE(cond, msg)                                                   \
    do {                                                                         \
        if (!(cond)) return g4s::set_error(G4S_ERR_INVALID, "%s: %s", __func__, msg); \
    } while (0)
