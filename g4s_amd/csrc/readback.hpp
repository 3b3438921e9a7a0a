// readback.hpp — small device→host reads (a count, a total, a flag) through a pinned block of the calling thread (runtime.cpp).
// hipMemcpyAsync into pageable memory — a stack variable — goes through the runtime's staging buffer and costs 40–100 µs of host time per read on this stack; a
// SpGEMM call makes ten of them (round 5: profiles/r05_small_sizes.txt). read_small enqueues the copy into the pinned block and notes where the value belongs;
// reads_sync / reads_sync_event wait for the stream / the event and hand EVERY noted value out (all of them were enqueued earlier on the same stream). The
// destination must be alive until then.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
namespace g4s {
hipError_t read_small(void *dst, const void *src, size_t bytes, hipStream_t s);
hipError_t reads_sync(hipStream_t s);
hipError_t reads_sync_event(hipEvent_t ev);
} // namespace g4s
