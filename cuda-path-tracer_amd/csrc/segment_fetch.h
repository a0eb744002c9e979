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

constexpr uint32_t kNoEntry = 0xffffffffu;  // unused tail of a segment's list

#ifdef __HIPCC__
// A wave's view of the segments it consumes.  The segments are split into kCursorRanges
// contiguous ranges with one fetch cursor each; a wave starts in the range of its XCD (workgroups
// are dealt to the XCDs round robin), so neighbouring tiles are traced next to each other and the
// cursor atomics are spread over several addresses, and it moves on to the other ranges when its
// own is drained.
//
// A segment's list is 64 words, entries packed at the front, kNoEntry behind them; a wave reads
// it with one load (one entry per lane) and hands entries to idle lanes with a shuffle.  Without
// a list (`queue` null) segment s simply holds the slots [64 s, 64 s + 64).
//
// AHEAD = true: the fetch runs two segments ahead of the consumer: while segment k is handed
// out, the list of segment k+1 is already loaded (or on its way) and the cursor atomic for segment
// k+2 is in flight, so a refill in the middle of a short traversal loop does not wait for a round
// trip to the memory-side atomic unit followed by a dependent load.  AHEAD = false claims a
// segment only when the wave needs one: nothing is hoarded, which balances better when a segment
// is a long piece of work (whole paths) and there are only a few segments per wave.
template <bool AHEAD>
struct SegmentFetchT {
    // wave-uniform
    uint32_t consumed = 0, count = 0;       // entries of the segment in hand: handed out / total
    uint32_t visited = 0;                   // ranges this wave has seen drained
    bool exhausted = false, next_valid = false, claimed = false;
    // per lane
    uint32_t ids = kNoEntry;                // this lane's entry of the segment in hand
    uint32_t next_ids = kNoEntry;           // ... of the segment after it
    uint32_t claim = 0;                     // lane 0: result of the cursor atomic issued ahead

    __device__ inline uint32_t remaining() const { return count - consumed; }
    // entry number (consumed + rank), for the lanes that take one
    __device__ inline uint32_t entry(uint32_t rank) const { return __shfl(ids, (int)(consumed + rank), 64); }

    // repeat (lists absent only): every range is walked `repeat` times over, pass after pass; the slots of pass k
    // of segment s are numbered (k * segments + s) * 64 + lane
    __device__ inline bool fill_next(const uint32_t *queue, uint32_t *cursors, uint32_t segments, uint32_t home, uint32_t lane, uint32_t repeat = 1) {
        while (visited < kCursorRanges) {
            const uint32_t r = (home + visited) % kCursorRanges;
            const uint32_t lo = (uint32_t)((unsigned long long)segments * r / kCursorRanges);
            const uint32_t size = (uint32_t)((unsigned long long)segments * (r + 1) / kCursorRanges) - lo;
            const uint32_t turns = size * repeat;
            uint32_t *cursor = &cursors[r * kCursorStride];
            if (!claimed) {
                // a drained range is recognised with a load, so that the waves arriving late from
                // other ranges do not queue read-modify-writes on its cursor
                uint32_t seen = 0;
                if (lane == 0) {
                    seen = __hip_atomic_load(cursor, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (seen < turns) claim = atomicAdd(cursor, 1u);
                }
                seen = __shfl(seen, 0, 64);
                if (seen >= turns) { visited++; continue; }
            }
            const uint32_t s = __shfl(claim, 0, 64);
            claimed = false;
            if (s >= turns) { visited++; continue; }
            next_ids = queue ? queue[(lo + s) * kSegmentSlots + lane] : ((s / size) * segments + lo + s % size) * kSegmentSlots + lane;
            next_valid = true;
            if (AHEAD) {
                if (lane == 0) claim = atomicAdd(cursor, 1u);      // consumed by the next call
                claimed = true;
            }
            return true;
        }
        return false;
    }

    // takes the next non-empty segment; false (and exhausted) when every range is drained
    __device__ inline bool advance(const uint32_t *queue, uint32_t *cursors, uint32_t segments, uint32_t home, uint32_t lane, uint32_t repeat = 1) {
        for (;;) {
            if (!next_valid && !fill_next(queue, cursors, segments, home, lane, repeat)) { exhausted = true; return false; }
            ids = next_ids;
            next_valid = false;
            if (AHEAD) fill_next(queue, cursors, segments, home, lane, repeat);
            count = (uint32_t)__popcll(__ballot(ids != kNoEntry));
            consumed = 0;
            if (count != 0) return true;
        }
    }
};

// a wave replaces its segment's list: the flagged lanes' values packed at the front, kNoEntry behind
__device__ inline uint32_t segment_publish(uint32_t *queue, uint32_t seg, bool flag, uint32_t value) {
    const unsigned long long m = __ballot(flag);
    const uint32_t lane = threadIdx.x & 63, total = (uint32_t)__popcll(m);
    uint32_t *list = queue + seg * kSegmentSlots;
    if (lane >= total) list[lane] = kNoEntry;
    if (flag) list[__popcll(m & ((1ull << lane) - 1ull))] = value;
    return total;
}

// rewinds one set of cursors; call from every thread of workgroup 0 while no kernel uses the set
__device__ inline void zero_cursors(uint32_t *cursors) {
    if (blockIdx.x == 0 && threadIdx.x < kCursorRanges) cursors[threadIdx.x * kCursorStride] = 0;
}

#endif  // __HIPCC__

}  // namespace cpt
