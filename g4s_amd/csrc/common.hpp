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

// stream-ordered scratch from the library's own never-trimming memory pool (runtime.cpp)
int scratch_alloc(void **p, size_t bytes, hipStream_t s);
void scratch_free(void *p, hipStream_t s);
int scratch_shutdown();
// per-call arena over the big-block cache (runtime.cpp): between arena_enter and the matching arena_leave of a thread, scratch_alloc carves from cached chunks
// and scratch_free does nothing; the outermost leave returns the chunks (idle: the call's stream has been synchronised)
void arena_enter();
void arena_leave(hipStream_t s, bool idle);
void spgemm_release_cache();   // spgemm.hip
// caching allocator for large device blocks (runtime.cpp); big_free returns false for a pointer it does not own
int big_alloc(void **p, size_t bytes);
bool big_free(void *p, bool idle = false);   // idle: every stream that used the block has been synchronised by the caller
void big_release_all();
void release_cached_device_memory();   // big_release_all + the SpGEMM column scratch (what g4s_trim does)
hipError_t device_malloc(void **p, size_t bytes);   // hipMalloc that drops the library's caches and retries once on out-of-memory

// 8 XCDs, each with its own L2: block b and b+8 share one (MI355X_MICROARCH.md, Workgroup dispatch).
constexpr int kXcds = 8;

// Device-side tables of an element operator (graph.hip owns them): node → (element, local node) terms in ascending element
// order, equation ids per element unknown and per (node, dof).
struct ElemOpView {
    int nel, npe, dof, nno, neq;
    const int *node_ptr, *node_terms, *elem_eq, *node_eq;
};

} // namespace g4s
int g4s_elem_op_view(g4s_elem_op_t op, g4s::ElemOpView *out);   // graph.hip

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
