// Dynamic hand-out of 64-slot segments (8x8 pixel tiles / lists of queued paths) to the waves of a
// persistent kernel.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace cpt {

constexpr uint32_t kSegmentSlots = 64;      // one segment = one 8x8 pixel tile = one wave of path slots
constexpr uint32_t kCursorRanges = 8;       // segment ranges with a fetch cursor each (one per XCD)
constexpr uint32_t kCursorStride = 16;      // words between cursors: one 64-byte line each
constexpr uint32_t kCursorSetWords = kCursorRanges * kCursorStride;

#ifdef __HIPCC__
// A wave's view of the segments it consumes.  The segments are split into kCursorRanges
// contiguous ranges with one fetch cursor each; a wave starts in the range of its XCD (workgroups
// are dealt to the XCDs round robin), so neighbouring tiles are traced next to each other and the
// cursor atomics are spread over several addresses, and it moves on to the other ranges when its
// own is drained.  All members are wave-uniform.
struct SegmentFetch {
    uint32_t next = 0, end = 0, visited = 0;
    bool exhausted = false;

    // takes the next non-empty segment; false when every range is drained.  counts: entries per
    // segment, or null when every segment is full.
    __device__ inline bool advance(const uint32_t *counts, uint32_t *cursors, uint32_t segments, uint32_t home, uint32_t lane) {
        while (visited < kCursorRanges) {
            const uint32_t r = (home + visited) % kCursorRanges;
            const uint32_t lo = (uint32_t)((unsigned long long)segments * r / kCursorRanges);
            const uint32_t size = (uint32_t)((unsigned long long)segments * (r + 1) / kCursorRanges) - lo;
            uint32_t s = 0;
            if (lane == 0) {
                // a drained range is recognised with a load: only waves that can still get a
                // segment pay for a read-modify-write on the shared cursor
                s = __hip_atomic_load(&cursors[r * kCursorStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (s < size) s = atomicAdd(&cursors[r * kCursorStride], 1u);
            }
            s = __shfl(s, 0, 64);
            if (s >= size) { visited++; continue; }
            const uint32_t c = counts ? counts[lo + s] : kSegmentSlots;
            if (c == 0) continue;
            next = (lo + s) * kSegmentSlots;
            end = next + c;
            return true;
        }
        exhausted = true;
        return false;
    }
};

// rewinds one set of cursors; call from every thread of workgroup 0 while no kernel uses the set
__device__ inline void zero_cursors(uint32_t *cursors) {
    if (blockIdx.x == 0 && threadIdx.x < kCursorRanges) cursors[threadIdx.x * kCursorStride] = 0;
}

#endif  // __HIPCC__

}  // namespace cpt
