// readback.hpp — small device→host reads (a count, a total, a flag) through a pinned block of the calling thread (runtime.cpp).
// hipMemcpyAsync into pageable memory — a stack variable — goes through the runtime's staging buffer and costs 40–100 µs of host time per read on this stack; a
// SpGEMM call makes ten of them (round 5: profiles/r05_small_sizes.txt). read_small enqueues the copy into the pinned block and notes where the value belongs;
// reads_sync(s) / reads_sync_event(ev, s) wait for the stream / for an event recorded on it BEHIND the reads, and hand out every value noted for that stream.
// The destination must be alive until then — or be taken back with reads_forget (an object that holds one and is freed first).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
namespace g4s {
hipError_t read_small(void *dst, const void *src, size_t bytes, hipStream_t s);
hipError_t reads_sync(hipStream_t s);
hipError_t reads_sync_event(hipEvent_t ev, hipStream_t s);
void reads_forget(const void *lo, const void *hi);   // drops the calling thread's noted reads whose destination lies in [lo, hi)
} // namespace g4s
