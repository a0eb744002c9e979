// Host interface of the wavefront (ray-queue) bounce loop (wavefront.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "device_types.h"
#include "segment_fetch.h"

namespace cpt {

// words of WavefrontBuffers::control
constexpr uint32_t kControlExtendCursors = 0;
constexpr uint32_t kControlShadowCursors = kCursorRanges * kCursorStride;
constexpr uint32_t kControlFinishCursors = 2 * kCursorRanges * kCursorStride;
constexpr uint32_t kControlLiveMarker = 3 * kCursorRanges * kCursorStride;
constexpr uint32_t kControlWords = kControlLiveMarker + kCursorStride;

// Path state in HBM, one 16-byte field per array, `capacity` = wavefront_slots(W, H) entries.
// A path never leaves its slot and a slot never leaves its segment, so a segment's list of live
// paths is written by exactly one wave per kernel: no atomics on the producer side, and the order
// of every list is the same from run to run.
struct WavefrontBuffers {
    float4 *ray_o;          // origin.xyz | pixel index + flags (bit 31 primary ray, bit 30 last lobe specular)
    float4 *ray_d;          // direction.xyz | sampler state
    float4 *thr;            // throughput.xyz | depth
    float4 *col;            // colour.xyz | pdf of the last BSDF sample
    float4 *hit;            // instance, primitive (0xffffffff = miss), barycentrics
    float4 *sh_o;           // shadow ray origin.xyz | tmax
    float4 *sh_d;           // shadow ray direction.xyz | 1 when the path ends after this ray
    float4 *pending;        // light contribution added when the shadow ray is unoccluded
    uint32_t *queue;        // per segment: ids of the paths whose next ray waits to be traced, packed at the front, kNoEntry behind
    uint32_t *shadow_queue; // per segment: ids of the paths with a shadow ray, same layout
    uint32_t *control;      // kControlWords words: fetch cursors + live marker
    uint32_t capacity;      // multiple of kSegmentSlots
    uint32_t segments;
};

size_t wavefront_slots(uint32_t width, uint32_t height);
// workgroups of a queue traversal kernel the device holds at once; dense = the build with more waves per
// SIMD and a shorter LDS stack, meant for scenes whose BVH does not fit the vector L1
int wavefront_resident_blocks(int device, bool dense, bool wide);
int wavefront_finish_resident_blocks(int device, bool wide);   // same for the finishing kernel
// camera rays + fresh path state for sample p.spp; fills the queue
int wavefront_begin_sample(const SceneDev &sc, const RenderParams &p, const WavefrontBuffers &wb, hipStream_t stream);
// one bounce: extend the queued rays, shade (refills the queue and the shadow queue), trace shadow
// rays.  `bounce_id` (non-zero, different for every bounce of a sample) is stored to the live
// marker when at least one path goes on.
int wavefront_bounce(const SceneDev &sc, const RenderParams &p, const WavefrontBuffers &wb, uint32_t bounce_id, uint32_t resident_blocks, bool dense,
                     hipStream_t stream);
// runs every path still queued to its end in registers (one persistent launch)
int wavefront_finish(const SceneDev &sc, const RenderParams &p, const WavefrontBuffers &wb, uint32_t resident_blocks, hipStream_t stream);

}  // namespace cpt
