// Host interface of the wavefront (ray-queue) bounce loop (wavefront.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "device_types.h"

namespace cpt {

// Path state in HBM, one 16-byte field per array, `capacity` = wavefront_slots(W, H) entries.
struct WavefrontBuffers {
    float4 *ray_o;          // origin.xyz | pixel index + flags (bit 31 primary ray, bit 30 last lobe specular)
    float4 *ray_d;          // direction.xyz | sampler state
    float4 *thr;            // throughput.xyz | depth
    float4 *col;            // colour.xyz | pdf of the last BSDF sample
    float4 *hit;            // instance, primitive (0xffffffff = miss), barycentrics
    float4 *sh_o;           // shadow ray origin.xyz | tmax
    float4 *sh_d;           // shadow ray direction.xyz | 1 when the path ends after this ray
    float4 *pending;        // light contribution added when the shadow ray is unoccluded
    uint32_t *queue[2];     // live path ids, ping-pong
    uint32_t *shadow_queue;
    uint32_t *counts;       // 8 words: [0],[1] queue sizes, [2] shadow queue size, [3],[4] fetch cursors
    uint32_t capacity;
};

size_t wavefront_slots(uint32_t width, uint32_t height);
int wavefront_resident_blocks(int device);
// camera rays + fresh path state for sample p.spp; fills queue 0
int wavefront_begin_sample(const SceneDev &sc, const RenderParams &p, const WavefrontBuffers &wb, hipStream_t stream);
// one bounce: extend queue[parity], shade (fills queue[parity ^ 1] and the shadow queue), trace shadow rays, reset
int wavefront_bounce(const SceneDev &sc, const RenderParams &p, const WavefrontBuffers &wb, uint32_t parity, uint32_t resident_blocks, hipStream_t stream);

}  // namespace cpt
