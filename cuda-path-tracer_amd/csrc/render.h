// Host launchers of the traversal / bounce-loop kernels (render.hip).  All return 0 or a hipError_t.
#pragma once
#include <hip/hip_runtime.h>

#include "device_types.h"
#include "segment_fetch.h"

namespace cpt {

// One sample per pixel into p.output: kernel::PathTracer::Render (src/kernels/integrator/path.cuh:22)
// without the trailing device synchronisation.
int launch_render(const SceneDev &sc, const RenderParams &p, hipStream_t stream);

// The same sample through the persistent, lane-regenerating kernel.  `queue` is a device array of
// kRegenQueueWords uint32 owned by the caller: two sets of tile cursors.  Set `parity` must be zero
// when the launch starts (the kernel zeroes the other set for the next launch, so alternate parity
// 0,1,0,.. on one stream after zeroing everything once).  max_blocks: render_regen_resident_blocks() of the device.
// dense: the build with more waves per SIMD and a shorter LDS stack, for scenes whose BVH does not fit the vector L1.
constexpr uint32_t kRegenQueueWords = 2 * kCursorSetWords;
// variant: 0 = k_render_regen; 1 = the build that keeps traversal state across shading and re-enters the loop with partly
// finished waves (k_render_flow); 2 = the build whose finished lanes walk the wave's pooled shadow rays (k_render_pool).
// sc.fast_traversal selects the builds of variants 1 and 2 with the ordered traversal (variant 0 has none and runs exact).
// p.spp_count > 1 (variants 1 and 2 only): one launch renders that many consecutive samples; p.turn_in_alpha is set (running
// mean into a film whose alpha is 0 or 1 everywhere) or p.order must point to
// width * height zeroed words.  The film ends up bit for bit as after one launch per sample.
int launch_render_regen(const SceneDev &sc, const RenderParams &p, uint32_t *queue, uint32_t parity, uint32_t max_blocks, bool dense, int variant, hipStream_t stream);
// tiles (of 64 path slots) one sample of a width x height film takes in those kernels: whole blocks of 16 x 16 tiles
uint32_t render_tile_count(uint32_t width, uint32_t height);
int render_regen_resident_blocks(int device, bool dense, int variant, bool fast, bool wide, bool scene_in_lds);
// `records` two-level node records in the form the LDS copy of a small scene holds them (SceneDev::lds_scene): kLdsNodeWords 16-byte words each
int launch_lds_scene_nodes(float *dst, const PackedNode *wide, uint32_t records, hipStream_t stream);

// Compares the traversal's reciprocal-based exact division with the hardware division on at least
// `pairs` operand pairs; adds the number of differing results to *d_mismatches (device memory).
int selftest_division(uint64_t pairs, uint32_t seed, unsigned long long *d_mismatches, hipStream_t stream);

// Copies bytes / 16 * 16 bytes from src to dst (both 16-byte aligned device memory) with a plain streaming kernel.
int launch_stream_copy(const void *src, void *dst, uint64_t bytes, hipStream_t stream);
int launch_valu_rate(int mix, int waves_per_simd, int compute_units, uint32_t iterations, float *sink, unsigned long long *cycles, hipStream_t stream);

// Films of sharded renders (W*H float4 each): dst += src; sums with the sample count in alpha -> mean with alpha 1.
int launch_film_add(float *dst, const float *src, uint64_t pixels, hipStream_t stream);
int launch_film_finish(float *film, uint64_t pixels, hipStream_t stream);

// rays: device array of n * 8 floats (origin.xyz, tmin, direction.xyz, tmax).
// inst/prim receive 0xffffffff on a miss; t receives the hit distance (or the input tmax);
// t and uv may be null.
int launch_first_hit(const SceneDev &sc, const float *rays, uint64_t n, uint32_t *inst, uint32_t *prim, float *t, float *uv, hipStream_t stream);
int launch_occlude(const SceneDev &sc, const float *rays, uint64_t n, uint8_t *out, hipStream_t stream);
// cpt_tracer_first_hit as a persistent kernel whose lanes take a new ray when theirs is finished; `queue` as for launch_render_regen
int launch_first_hit_stream(const SceneDev &sc, const float *rays, uint64_t n, uint32_t *inst, uint32_t *prim, float *t, float *uv, uint32_t *queue, uint32_t parity,
                            uint32_t max_blocks, hipStream_t stream);
int first_hit_stream_resident_blocks(int device, bool large, bool fast, bool wide);

// Camera rays of 1-based sample index spp for every pixel, stored in pixel-index order (row 0 = top).
int launch_camera_rays(const CameraDev &cam, uint32_t W, uint32_t H, uint32_t spp, float *rays, hipStream_t stream);

}  // namespace cpt
