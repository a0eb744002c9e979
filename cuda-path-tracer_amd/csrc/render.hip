// Megakernel path tracer for gfx950: one lane = one pixel = one path per launch, the direct
// counterpart of RenderKernel/Trace (src/kernels/integrator/path.cu:9-105), plus the batch
// first-hit / any-hit / camera-ray kernels used by the parity tests and the benchmark.
#include "render.h"

#include <hip/hip_runtime.h>

#include "flow_lane.h"
#include "path_lane.h"
#include "segment_fetch.h"

#include <type_traits>

namespace cpt {

namespace {

// Tiles of 8 x 8 pixels, one wave each, in the order the persistent kernels hand them out: blocks of 16 x 16 tiles, row by
// row inside a block and block after block (the tile count is rounded up to whole blocks; tiles outside the film hand out
// no pixels), so that the few hundred tiles the waves of one XCD hold at a time form a patch of the image and not one row of
// it, and their rays meet in the caches more often.  Against plain rows of tiles, Mpaths/s: heightfield 10M 617 -> 623, soup 1M
// 225 -> 226, Cornell box 1741 -> 1748; blocks of 4 / 8 / 32 tiles: 617 / 618 / 624 on the heightfield.
constexpr uint32_t kTileBlock = 16;
__host__ __device__ inline uint32_t tile_count(uint32_t width, uint32_t height) {
    const uint32_t tx = (width + 7) / 8, ty = (height + 7) / 8;
    return ((tx + kTileBlock - 1) / kTileBlock) * ((ty + kTileBlock - 1) / kTileBlock) * kTileBlock * kTileBlock;
}
__device__ inline void tile_pixel(uint32_t tile, uint32_t k, uint32_t width, uint32_t &px, uint32_t &py) {
    const uint32_t blocks_x = ((width + 7) / 8 + kTileBlock - 1) / kTileBlock;
    const uint32_t block = tile / (kTileBlock * kTileBlock), in = tile % (kTileBlock * kTileBlock);
    // the tile's 64 pixels in Morton order: four neighbouring lanes hold a 2 x 2 block of pixels, whose camera rays read the
    // same records (the address unit takes the lanes of a group of four that stand in one line in one cycle); Cornell box
    // 1744 -> 1756 Mpaths/s against rows of eight, the large scenes unchanged
    const uint32_t kx = (k & 1u) | ((k >> 1) & 2u) | ((k >> 2) & 4u), ky = ((k >> 1) & 1u) | ((k >> 2) & 2u) | ((k >> 3) & 4u);
    px = ((block % blocks_x) * kTileBlock + in % kTileBlock) * 8 + kx;
    py = ((block / blocks_x) * kTileBlock + in / kTileBlock) * 8 + ky;
}

// src/kernels/integrator/path.cu:9-83
template <bool MAPS, class Stack>
__device__ inline V3 trace_path(const SceneDev &sc, V3 ro, V3 rd, Sampler &sampler, uint32_t max_depth, int channel, Stack &st, LaneCounters &cnt) {
    Hit hit;
    float tmax = kRayTMax;
    if (!trace<false>(sc, true, ro, rd, kRayTMin, tmax, hit, st, cnt)) return v3(0.0f);
    Surface surface = shading_surface<MAPS>(sc, hit, cnt);
    if (channel == 1) return surface.vertex.normal * 0.5f + v3(0.5f);
    V3 color = surface.bsdf.emission;
    if (!is_zero(color)) return color;
    V3 throughput = v3(1.0f);
    for (uint32_t depth = 0; depth < max_depth; depth++) {
        Frame frame = Frame::from_normal(surface.vertex.normal);
        V3 wo = frame.to_local(-rd);

        if (!bsdf_is_delta(surface.bsdf)) {
            float r1 = sampler.next1d();
            uint32_t index = min(f2u_sat(r1 * (float)sc.num_lights), sc.num_lights - 1);   // light/light.cuh:39-43
            V2 r2 = sampler.next2d();
            LightSample ls = light_sample<MAPS>(sc, sc.lights[index], surface.vertex.position, r2);
            ls.pdf *= sc.light_pick_pdf;
            ls.weight = ls.weight / sc.light_pick_pdf;
            if (ls.pdf > 0.0f) {
                Hit dummy;
                float smax = ls.dist - kShadowRayEps;
                if (!trace<true>(sc, true, surface.vertex.position, ls.dir, kRayTMin, smax, dummy, st, cnt)) {
                    V3 wi = frame.to_local(ls.dir);
                    float bp = bsdf_pdf(surface.bsdf, wo, wi);
                    float mis_weight = power_heuristic(ls.pdf, bp);      // area lights are never delta
                    color = color + throughput * mis_weight * ls.weight * bsdf_eval(surface.bsdf, wo, wi);
                }
            }
        }

        float b1 = sampler.next1d();
        V2 b2 = sampler.next2d();
        BsdfSample bs = bsdf_sample(surface.bsdf, wo, b1, b2);
        if (bs.pdf == 0.0f) break;
        throughput = throughput * bs.weight;
        ro = surface.vertex.position;
        rd = frame.to_world(bs.wi);
        tmax = kRayTMax;
        if (!trace<false>(sc, true, ro, rd, kRayTMin, tmax, hit, st, cnt)) break;

        surface = shading_surface<MAPS>(sc, hit, cnt);
        if (!is_zero(surface.bsdf.emission) && dot(rd, surface.vertex.normal) < 0.0f) {
            float mis_weight = 1.0f;
            if (bs.lobe != kLobeSpecular) {
                float light_pdf = surface.vertex.pdf * path_to_solid_angle_jacobian(ro, surface.vertex.position, surface.vertex.normal) *
                                  sc.light_pick_pdf;
                mis_weight = power_heuristic(bs.pdf, light_pdf);
            }
            color = color + throughput * mis_weight * surface.bsdf.emission;
            break;
        }

        float rr_prop = gmin(gmax(luminance(throughput), 0.01f), 0.95f);
        if (sampler.next1d() > rr_prop) break;
        throughput = throughput / rr_prop;
    }
    return color;
}

// Workgroups are dealt to the 8 XCDs round robin; give each XCD one contiguous band of tiles so
// that neighbouring tiles, which walk the same part of the BVH, share an L2.
__device__ inline uint32_t xcd_band_remap(uint32_t b, uint32_t nb) {
    uint32_t q = nb / 8, r = nb % 8, xcd = b % 8, k = b / 8;
    uint32_t start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + k;
}

// one workgroup = one 16x16 pixel tile, one wave = one 8x8 sub-tile
template <bool MAPS, bool WIDE>
__global__ void __launch_bounds__(kTraceBlock) k_render(SceneDev sc, RenderParams p) {
    __shared__ uint2 s_stack[kLdsStackEntries * kTraceBlock];
    __shared__ unsigned long long s_cnt[5];
    uint2 spill_entries[kSpillEntries];
    TraceStackT<kLdsStackEntries, 1, false, WIDE> st(s_stack, spill_entries, sc.counters);
    LaneCounters cnt;
    const uint32_t tiles_x = (p.width + 15) / 16;
    const uint32_t tile = xcd_band_remap(blockIdx.x, gridDim.x);
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t px = (tile % tiles_x) * 16 + (w & 1) * 8 + (lane & 7);
    const uint32_t py = (tile / tiles_x) * 16 + (w >> 1) * 8 + (lane >> 3);
    const bool active = px < p.width && py < p.height;
    if (active) {
        Sampler sampler;
        uint32_t pixel_index;
        V3 ro, rd;
        pixel_ray(sc.camera, p.width, p.height, px, py, p.spp, sampler, pixel_index, ro, rd);
        V3 color = trace_path<MAPS>(sc, ro, rd, sampler, p.max_depth, p.channel, st, cnt);
        film_accumulate(p, pixel_index, color);
    }
    flush_counters(sc, cnt, active ? 1u : 0u, s_cnt);
}

// Persistent variant of the same integrator: every lane still follows exactly one path at a
// time with the reference's operation order, but a lane whose path ends picks up the next pixel
// at once instead of idling until the slowest path of its wave is done.  Pixel slots are handed
// out one 8x8 tile at a time (SegmentFetch: a cursor per XCD over a contiguous band of tiles, so
// neighbouring tiles share an L2); idle lanes are found with a ballot and numbered with a popcount
// prefix.  One trip of the loop is path_trip().
// DENSE: the build for scenes whose BVH misses the vector L1: 5 waves per SIMD and 12 LDS stack entries
// instead of 4 and 16 (a few registers spill, the extra wave hides more than that costs: heightfield 10M
// 187 -> 199 Mpaths/s; cornell-box-glass, all L1 hits, is 10 % slower that way and keeps the other build).
template <bool DENSE> struct RegenShape { static constexpr int kWaves = DENSE ? 5 : 4, kLdsEntries = DENSE ? 12 : 16; };

template <bool MAPS, bool DENSE, bool WIDE>
__global__ void __launch_bounds__(kTraceBlock, RegenShape<DENSE>::kWaves) k_render_regen(SceneDev sc, RenderParams p, uint32_t *queue, uint32_t parity) {
    __shared__ uint2 s_stack[RegenShape<DENSE>::kLdsEntries * kTraceBlock];
    __shared__ unsigned long long s_cnt[5];
    uint2 spill_entries[kSpillEntries];
    TraceStackT<RegenShape<DENSE>::kLdsEntries, DENSE ? kLeafWeightLargeScene : 1, false, WIDE> st(s_stack, spill_entries, sc.counters);
    LaneCounters cnt;
    uint32_t paths = 0;
    const uint32_t lane = threadIdx.x & 63;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const uint32_t tiles = tile_count(p.width, p.height);
    uint32_t *cursors = queue + parity * kCursorSetWords;
    zero_cursors(queue + (parity ^ 1u) * kCursorSetWords);      // the next launch's cursors

    SegmentFetchT<false> f;
    PathLane l;

    for (;;) {
        // ---- hand new pixels to idle lanes
        unsigned long long idle = __ballot(!l.alive);
        while (idle != 0ull && !f.exhausted) {
            if (f.remaining() == 0 && !f.advance(nullptr, cursors, tiles, blockIdx.x, lane)) break;
            const uint32_t rank = __popcll(idle & lt_mask);
            const uint32_t take = min((uint32_t)__popcll(idle), f.remaining());
            const uint32_t slot = f.entry(rank);
            if (!l.alive && rank < take) {
                const uint32_t tile = slot >> 6, k = slot & 63u;
                uint32_t px, py;
                tile_pixel(tile, k, p.width, px, py);
                if (px < p.width && py < p.height) {
                    pixel_ray(sc.camera, p.width, p.height, px, py, p.spp, l.sampler, l.pixel_index, l.ro, l.rd);
                    l.alive = true; l.primary = true; l.last_specular = false;
                    l.depth = 0; l.color = v3(0.0f); l.throughput = v3(1.0f);
                    paths++;
                }
            }
            f.consumed += take;
            idle = __ballot(!l.alive);
        }
        if (__ballot(l.alive) == 0ull) break;        // nothing in flight and every tile is handed out

        path_trip<MAPS>(sc, p, l, st, cnt);
    }
    flush_counters(sc, cnt, paths, s_cnt);
}

// The persistent kernel again, but a wave no longer waits for its slowest ray before it shades:
// traversal state stays in registers across the shading step (flow_lane.h), lanes with a finished
// extension ray are shaded and given their next rays as soon as they are five eighths of the
// wave's live lanes, and shadow rays run in the same loop as extension rays.  Lanes that enter the loop at
// different times vote less often for the same kind of step, which costs more than the idle lanes
// did on a scene that sits in the vector L1 (cornell-box-glass: 1.66 against 1.42 ms per sample);
// with long rays through a large BVH it wins (heightfield 10M: 248 -> 395 Mpaths/s, soup 1M 73 -> 104).
// Four waves per SIMD and 16 LDS stack entries for every scene: five waves measured the same, six spill.
// The kernel's shape and thresholds; each can be overridden on the compiler's command line for an A/B build
// (scripts/build_variant.sh).  Measured on two-level records, heightfield 10M 3840 x 2160 depth 5, Mpaths/s:
//   waves per SIMD / LDS stack entries   4 / 16: 549   5 / 12: 404 (spills)   4 / 12: no gain   (3 waves, no spills: 525)
//   shade threshold in eighths           2: 524   3: 549   4: 563   5: 570   6: 551
//   leaf weight of the step vote         1: 568   2: 576   3: 571   4: 563   6: 536   (each with threshold 4)
//   trips between looks at ended rays    8: 535   16: 549   32: 561 with threshold 4, where 16 gives 563
//   threshold 5 with weight 3: 575 (soup 1M 219 against 220, heightfield 1M 670 against 648); with weight 2: 578, but soup 1M 211
// Three things that were built, measured on the same scene and taken out again.  (1) Asking early for the record of a subtree
// that waits on the stack, by a four-byte load straight into a scrap area of LDS (no register, nothing to wait for), so
// that the fetch at pop time finds the line in the caches: when the subtree is pushed 493 against 550 Mpaths/s; for the
// top of the stack, issued by hand BEHIND the eight loads of the record being visited with the wait written out so that
// the touch alone stays in flight (loads return in order: a touch holds back every load issued after it until it has
// arrived itself) 541 against 575.  The next record's loads queue behind a touch that went to memory, and that costs more
// than the pop saves.  (2) Shading a hit from one 64-byte surface record per triangle slot (normals and texture
// coordinates as first value + differences, next to the triangle record) instead of three indices and then the vertex
// attributes: 574 against 576, the same film bit for bit -- the fetches of the shading step are not what the kernel waits for.
// (3) The camera rays of a launch traced beforehand by lockstep waves of their own (one per 8 x 8 tile and sample, first hits
// kept in 16 bytes per pixel slot) with the bounce loop starting every path at its first shading step: 568 against 563 with the
// branch compiled in and 575 without it; soup 1M 198 against 211 -- camera rays cost this kernel what they cost a lockstep
// kernel (about 0.32 ns each), and the pass only adds its launch and the registers of one more branch.
#ifndef CPT_FLOW_WAVES
#define CPT_FLOW_WAVES 4
#endif
#ifndef CPT_FLOW_LDS
#define CPT_FLOW_LDS kLdsStackEntries
#endif
#ifndef CPT_FLOW_PERIOD
#define CPT_FLOW_PERIOD 15u
#endif
#ifndef CPT_FLOW_LEAF_WEIGHT
#define CPT_FLOW_LEAF_WEIGHT 3
#endif
#ifndef FLOW_NUM
#define FLOW_NUM 5
#endif
constexpr int kFlowWaves = CPT_FLOW_WAVES;
constexpr int kFlowLeafWeight = CPT_FLOW_LEAF_WEIGHT;      // this kernel's weight of a lane at a triangle or an instance in the step vote (Traversal::vote_step)
constexpr int kFlowShadeEighths = FLOW_NUM;      // shade when the lanes waiting for it are this many eighths of the wave's live lanes

#ifdef CPT_STEP_STATS
__device__ unsigned long long g_step_stats[18];      // development build: step statistics of k_render_flow / k_render_pool (trace.h: StepStats)
#endif
template <bool MAPS, bool FAST, bool WIDE>
__global__ void __launch_bounds__(kTraceBlock, kFlowWaves) k_render_flow(SceneDev sc, RenderParams p, uint32_t *queue, uint32_t parity) {
    __shared__ uint2 s_stack[CPT_FLOW_LDS * kTraceBlock];
    __shared__ unsigned long long s_cnt[5];
    __shared__ float s_next[4 * kTraceBlock];      // per lane: reciprocal direction and flags of the extension ray that follows its shadow ray
    uint2 spill_entries[kSpillEntries];
    TraceStackT<CPT_FLOW_LDS, kFlowLeafWeight, FAST, WIDE> st(s_stack, spill_entries, sc.counters);      // the build for large scenes
    LaneCounters cnt;
#ifdef CPT_STEP_STATS
    StepStats stats;
    cnt.stats = &stats;
    const unsigned long long stat_begin = clock64();
#endif
    uint32_t paths = 0;
    const uint32_t lane = threadIdx.x & 63;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const uint32_t tiles = tile_count(p.width, p.height);
    uint32_t *cursors = queue + parity * kCursorSetWords;
    zero_cursors(queue + (parity ^ 1u) * kCursorSetWords);      // the next launch's cursors

    SegmentFetchT<false> fetch;
    FlowLane s;
    Flow t;

    for (;;) {
        // ---- shade the lanes whose extension ray is done
#ifdef CPT_STEP_STATS
        const unsigned long long stat_s0 = clock64();
        stats.lanes[4] += lanes_in(__ballot(s.l.alive && !t.walking()));
#endif
        if (s.l.alive && !t.walking())
            flow_shade<MAPS>(sc, p, s, t, cnt, [&](V3, V3 dir, float reach, V3 contribution) {
                s.shadow_dir = dir; s.shadow_tmax = reach; s.pending = contribution;
                return true;
            });
#ifdef CPT_STEP_STATS
        stats.time[4] += clock64() - stat_s0; stats.trips[4]++;
#endif

        // ---- hand new pixels to idle lanes
        unsigned long long idle = __ballot(!s.l.alive);
        while (idle != 0ull && !fetch.exhausted) {
            if (fetch.remaining() == 0 && !fetch.advance(nullptr, cursors, tiles, blockIdx.x, lane, p.spp_count)) break;
            const uint32_t rank = __popcll(idle & lt_mask);
            const uint32_t take = min((uint32_t)__popcll(idle), fetch.remaining());
            const uint32_t slot = fetch.entry(rank);
            if (!s.l.alive && rank < take) {
                const uint32_t pass = (slot >> 6) / tiles, tile = (slot >> 6) % tiles, k = slot & 63u;
                uint32_t px, py;
                tile_pixel(tile, k, p.width, px, py);
                if (px < p.width && py < p.height) {
                    PathLane &l = s.l;
                    l.pass = pass; l.film_pending = false;
                    pixel_ray(sc.camera, p.width, p.height, px, py, p.spp + pass, l.sampler, l.pixel_index, l.ro, l.rd);
                    l.alive = true; l.primary = true; l.last_specular = false;
                    l.depth = 0; l.color = v3(0.0f); l.throughput = v3(1.0f);
                    s.has_ext = true;
                    paths++;
                }
            }
            fetch.consumed += take;
            idle = __ballot(!s.l.alive);
        }
        if (__ballot(s.l.alive) == 0ull) break;        // nothing in flight and every tile is handed out

        // ---- begin the rays of the lanes just shaded: the shadow ray if there is one, else the extension
        // ray; an extension ray that follows a shadow ray is prepared now, while the lanes are together,
        // and begun inside the loop from what LDS keeps
        auto ray_ended = [&]() {
            if (s.has_shadow) {
                if (!t.found) s.l.color = s.l.color + s.pending;
                s.has_shadow = false;
                if (s.has_ext) {
                    Flow::Prepared q;
                    q.r = v3(s_next[threadIdx.x], s_next[kTraceBlock + threadIdx.x], s_next[2 * kTraceBlock + threadIdx.x]);
                    const uint32_t flags = __float_as_uint(s_next[3 * kTraceBlock + threadIdx.x]);
                    q.fast = flags & 1u; q.enters = flags & 2u;
                    t.template begin<WIDE>(sc, s.l.ro, s.l.rd, kRayTMax, q, cnt);
                }
            }
            if (!s.has_shadow && !t.walking()) { s.ext_traced = s.has_ext; s.has_ext = false; }
        };
        if (s.l.alive && !t.walking() && !s.l.film_pending) {
            const Flow::Prepared ext = Flow::prepare<FAST>(sc, s.l.ro, s.l.rd, kRayTMax);
            if (s.has_shadow) {
                if (s.has_ext) {
                    s_next[threadIdx.x] = ext.r.x; s_next[kTraceBlock + threadIdx.x] = ext.r.y; s_next[2 * kTraceBlock + threadIdx.x] = ext.r.z;
                    s_next[3 * kTraceBlock + threadIdx.x] = __uint_as_float((ext.fast ? 1u : 0u) | (ext.enters ? 2u : 0u));
                }
                t.template begin<WIDE>(sc, s.l.ro, s.shadow_dir, s.shadow_tmax, Flow::prepare<FAST>(sc, s.l.ro, s.shadow_dir, s.shadow_tmax), cnt);
            } else {
                t.template begin<WIDE>(sc, s.l.ro, s.l.rd, kRayTMax, ext, cnt);
            }
            if (!t.walking()) ray_ended();
        }

        // ---- walk until enough lanes want shading.  Rays end at scattered trips; looking after them (light
        // contribution, the extension ray that follows a shadow ray) and counting lanes every trip would put that
        // code into nearly every trip for one or two lanes, so it runs every sixteenth trip and when nothing moves.
        for (uint32_t trip = 0;; trip++) {
            const bool shadow = s.has_shadow;
            auto world = [&](V3 &ow, V3 &dw) { ow = s.l.ro; dw = shadow ? s.shadow_dir : s.l.rd; };
            const bool moved = t.step(sc, st, cnt, world, shadow);
            if (moved && (trip & CPT_FLOW_PERIOD) != CPT_FLOW_PERIOD) continue;
            if (!t.walking() && (s.has_shadow || s.has_ext)) ray_ended();
            const int walking = lanes_in(__ballot(t.walking()));
            const int waiting = lanes_in(__ballot(!t.walking() && (s.l.alive || !fetch.exhausted)));
            if (walking == 0 || (waiting > 0 && waiting * 8 >= (walking + waiting) * kFlowShadeEighths)) break;
        }
    }
#ifdef CPT_STEP_STATS
    stats.time[5] = clock64() - stat_begin; stats.trips[5] = 1;
    if (lane == 0) for (int k = 0; k < 6; k++) { atomicAdd(&g_step_stats[k], stats.time[k]); atomicAdd(&g_step_stats[6 + k], stats.trips[k]); atomicAdd(&g_step_stats[12 + k], stats.lanes[k]); }
#endif
    flush_counters(sc, cnt, paths, s_cnt);
}

// The lane-regenerating kernel with its shadow rays pooled per wave.  Measured with a probe build on
// cornell-box-glass, 72 % of the lane slots of k_render_regen's traversal trips belong to lanes whose ray
// is already finished (5 % to lanes that lost the vote): every trace call lasts as long as its longest
// ray.  Here the wave still shades all its lanes together, once per bounce, but the shadow rays that
// shading produces go to a list in LDS, and in the next traversal loop a lane whose own extension ray is
// finished takes shadow rays from that list (any lane's), so they fill the slots that would idle and the
// separate shadow trace disappears.  The owner adds its light contribution when the next shading step
// finds its ray marked as having reached the light: same order of additions per path as before.
// SCENE_IN_LDS: the node and triangle records of a scene of a few KB (SceneDev::lds_scene) are copied into LDS by every
// workgroup, and the traversal reads them there: no record visit goes through the vector memory pipeline, whose address
// unit the 36-triangle Cornell box keeps busy 63 % of the time with everything in the L1.  The stack gives up three LDS
// entries per lane for the room.
template <bool DENSE, bool SCENE_IN_LDS = false> struct PoolShape { static constexpr int kWaves = DENSE ? 5 : 4, kLdsEntries = DENSE ? 7 : SCENE_IN_LDS ? 8 : 11; };
#ifndef POOL_MASK
#define POOL_MASK 7u
#endif
constexpr uint32_t kPoolPeriodMask = POOL_MASK;         // lanes are re-assigned every eighth trip

template <bool MAPS, bool DENSE, bool FAST, bool WIDE, bool SCENE_IN_LDS = false>
__global__ void __launch_bounds__(kTraceBlock, PoolShape<DENSE>::kWaves) k_render_pool(SceneDev sc, RenderParams p, uint32_t *queue, uint32_t parity) {
    __shared__ uint2 s_stack[PoolShape<DENSE, SCENE_IN_LDS>::kLdsEntries * kTraceBlock];
    __shared__ RawF4 s_scene[SCENE_IN_LDS ? kLdsSceneBytes / 16 : 1];
    __shared__ unsigned long long s_cnt[5];
    // the shadow ray of each lane's path, four 16-byte fields next to each other (64 bytes per lane): (origin, reach),
    // (direction, fast-division flag), (1 / direction, -), (light contribution, reached the light).  Field-major storage
    // ([field][lane], no bank conflicts between the lanes of a wave) was measured too and is 3.5 % slower on the Cornell
    // box (1219 against 1263 Mpaths/s): a lane reads its three fields with one address and immediate offsets here.
    __shared__ float4 s_pool[4 * kTraceBlock];
    auto pool = [&](uint32_t field, uint32_t slot) -> float4 & { return s_pool[4 * slot + field]; };
    __shared__ uint8_t s_order[kTraceBlock];         // per wave: the lanes that have a shadow ray to be walked, in lane order
    uint2 spill_entries[kSpillEntries];
    TraceStackT<PoolShape<DENSE, SCENE_IN_LDS>::kLdsEntries, DENSE ? kLeafWeightLargeScene : 1, FAST, WIDE, SCENE_IN_LDS> st(s_stack, spill_entries, sc.counters);
    if constexpr (SCENE_IN_LDS) {
        const RawF4 *src = reinterpret_cast<const RawF4 *>(sc.lds_scene);
        for (uint32_t w = threadIdx.x; w < sc.lds_scene_words; w += kTraceBlock) s_scene[w] = src[w];
        st.scene = (LdsF4)s_scene;
        __syncthreads();
    }
    LaneCounters cnt;
#ifdef CPT_STEP_STATS
    StepStats stats;
    cnt.stats = &stats;
    const unsigned long long stat_begin = clock64();
#endif
    uint32_t paths = 0;
    const uint32_t lane = threadIdx.x & 63, wave_base = threadIdx.x & ~63u;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const uint32_t tiles = tile_count(p.width, p.height);
    uint32_t *cursors = queue + parity * kCursorSetWords;
    zero_cursors(queue + (parity ^ 1u) * kCursorSetWords);      // the next launch's cursors

    SegmentFetchT<false> fetch;
    FlowLane s;
    Flow t;
    bool ext_found = false;                 // result of the lane's own extension ray, kept while it walks other lanes' shadow rays
    Hit ext_hit{0u, 0u, 0.0f, 0.0f};
    uint32_t job = 0xffffffffu;             // slot of the shadow ray the lane walks, in its wave's list
    uint32_t pool_count = 0;                // wave-uniform: shadow rays in the list

    for (;;) {
        // ---- hand new pixels to idle lanes
        unsigned long long idle = __ballot(!s.l.alive);
        while (idle != 0ull && !fetch.exhausted) {
            if (fetch.remaining() == 0 && !fetch.advance(nullptr, cursors, tiles, blockIdx.x, lane, p.spp_count)) break;
            const uint32_t rank = __popcll(idle & lt_mask);
            const uint32_t take = min((uint32_t)__popcll(idle), fetch.remaining());
            const uint32_t slot = fetch.entry(rank);
            if (!s.l.alive && rank < take) {
                const uint32_t pass = (slot >> 6) / tiles, tile = (slot >> 6) % tiles, k = slot & 63u;
                uint32_t px, py;
                tile_pixel(tile, k, p.width, px, py);
                if (px < p.width && py < p.height) {
                    PathLane &l = s.l;
                    l.pass = pass; l.film_pending = false;
                    pixel_ray(sc.camera, p.width, p.height, px, py, p.spp + pass, l.sampler, l.pixel_index, l.ro, l.rd);
                    l.alive = true; l.primary = true; l.last_specular = false;
                    l.depth = 0; l.color = v3(0.0f); l.throughput = v3(1.0f);
                    s.has_ext = true; s.has_shadow = false;
                    paths++;
                }
            }
            fetch.consumed += take;
            idle = __ballot(!s.l.alive);
        }
        if (__ballot(s.l.alive) == 0ull) break;        // nothing in flight and every tile is handed out

        // ---- every lane begins its own extension ray
        job = 0xffffffffu;
        s.ext_traced = s.l.alive && s.has_ext;
        t = Flow();                                     // nothing of the last bounce's walks is carried over
        if (s.ext_traced) t.template begin<WIDE>(sc, s.l.ro, s.l.rd, kRayTMax, Flow::prepare<FAST>(sc, s.l.ro, s.l.rd, kRayTMax), cnt);
        bool own = s.ext_traced;                        // the ray in `t` is the lane's own extension ray
        uint32_t pool_next = 0;

        // ---- walk; lanes without a ray take shadow rays from the list
        for (uint32_t trip = 0;; trip++) {
            const bool shadow = !own;
            auto world = [&](V3 &ow, V3 &dw) {
                if (own) { ow = s.l.ro; dw = s.l.rd; }
                else { float4 a = pool(0, wave_base + job), b = pool(1, wave_base + job); ow = v3(a.x, a.y, a.z); dw = v3(b.x, b.y, b.z); }
            };
            const bool moved = t.step(sc, st, cnt, world, shadow);
            if (moved && (trip & kPoolPeriodMask) != kPoolPeriodMask) continue;
            const bool free_lane = !t.walking();
            if (free_lane && own) { ext_found = t.found; ext_hit = t.hit; own = false; }
            if (free_lane && job != 0xffffffffu) {
                if (!t.found) pool(3, wave_base + job).w = 1.0f;      // reached the light
                job = 0xffffffffu;
            }
            const unsigned long long free_mask = __ballot(free_lane);
            const uint32_t avail = pool_count - pool_next, rank = __popcll(free_mask & lt_mask);
            if (free_lane && rank < avail) {
                job = s_order[wave_base + pool_next + rank];
                const float4 a = pool(0, wave_base + job), b = pool(1, wave_base + job), c = pool(2, wave_base + job);
                Flow::Prepared q;
                q.r = v3(c.x, c.y, c.z);
                q.fast = __float_as_uint(b.w) != 0u;
                q.enters = true;                        // rays that miss the scene's box are not listed
                t.template begin<WIDE>(sc, v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), a.w, q, cnt, false);      // counted when it was listed
            }
            pool_next += min((uint32_t)__popcll(free_mask), avail);
            if (__ballot(t.walking()) == 0ull && pool_next == pool_count) break;
        }

        // ---- shade: last bounce's light contribution first, then this bounce's hit.  The shadow ray goes to the
        // lane's slot of the list as soon as it is known, with what another lane needs to begin it; one that misses
        // the scene's box has nothing between the point and the light.
#ifdef CPT_STEP_STATS
        const unsigned long long stat_s0 = clock64();
        stats.lanes[4] += lanes_in(__ballot(s.l.alive));
#endif
        if (s.l.alive) {
            if (s.has_shadow) {
                const float4 c = pool(3, threadIdx.x);
                if (c.w != 0.0f) s.l.color = s.l.color + v3(c.x, c.y, c.z);
                s.has_shadow = false;
            }
            Flow result;
            result.found = ext_found;
            result.hit = ext_hit;
            flow_shade<MAPS>(sc, p, s, result, cnt, [&](V3 pos, V3 dir, float reach, V3 contribution) {
                const Flow::Prepared q = Flow::prepare<FAST>(sc, pos, dir, reach);
                cnt.rays++;
                if (!q.enters) { s.l.color = s.l.color + contribution; return false; }
                pool(0, threadIdx.x) = make_float4(pos.x, pos.y, pos.z, reach);
                pool(1, threadIdx.x) = make_float4(dir.x, dir.y, dir.z, __uint_as_float(q.fast ? 1u : 0u));
                pool(2, threadIdx.x) = make_float4(q.r.x, q.r.y, q.r.z, 0.0f);
                pool(3, threadIdx.x) = make_float4(contribution.x, contribution.y, contribution.z, 0.0f);
                return true;
            });
        }
#ifdef CPT_STEP_STATS
        stats.time[4] += clock64() - stat_s0; stats.trips[4]++;
#endif
        const unsigned long long with_shadow = __ballot(s.l.alive && s.has_shadow);
        pool_count = __popcll(with_shadow);
        if (s.l.alive && s.has_shadow) s_order[wave_base + __popcll(with_shadow & lt_mask)] = (uint8_t)lane;
    }
#ifdef CPT_STEP_STATS
    stats.time[5] = clock64() - stat_begin; stats.trips[5] = 1;
    if (lane == 0) for (int k = 0; k < 6; k++) { atomicAdd(&g_step_stats[k], stats.time[k]); atomicAdd(&g_step_stats[6 + k], stats.trips[k]); atomicAdd(&g_step_stats[12 + k], stats.lanes[k]); }
#endif
    flush_counters(sc, cnt, paths, s_cnt);
}

// The batch query kernels hold one ray and no path state: few registers, so more waves per SIMD when
// the LDS stack is shortened to match (deeper entries go to private memory).
constexpr int kQueryWaves = 8;
#ifndef WIDE_WAVES
#define WIDE_WAVES 5
#endif
constexpr int kWideQueryWaves = WIDE_WAVES;
constexpr int kQueryLdsEntries = 8;
// batches are often coherent camera rays, which lose with the bounce loops' weight of 4 (heightfield 10M: 2.59 Grays/s
// with 1 or 2, 2.46 with 4) while diverging ones gain (soup 1M: 0.67 with 1, 0.75 with 2, 0.81 with 4)
constexpr int kQueryLeafWeightLargeScene = 2;

template <bool LARGE, bool FAST, bool WIDE>
__global__ void __launch_bounds__(kTraceBlock, WIDE ? kWideQueryWaves : kQueryWaves) k_first_hit(SceneDev sc, const float *__restrict__ rays, unsigned long long n, uint32_t *__restrict__ inst_out,
                                                            uint32_t *__restrict__ prim_out, float *__restrict__ t_out, float *__restrict__ uv_out) {
    __shared__ uint2 s_stack[kQueryLdsEntries * kTraceBlock];
    __shared__ unsigned long long s_cnt[5];
    uint2 spill_entries[kSpillEntries];
    TraceStackT<kQueryLdsEntries, LARGE ? kQueryLeafWeightLargeScene : 1, FAST, WIDE> st(s_stack, spill_entries, sc.counters);
    LaneCounters cnt;
    unsigned long long i = (unsigned long long)blockIdx.x * kTraceBlock + threadIdx.x;
    if (i < n) {
        const float4 *r = reinterpret_cast<const float4 *>(rays) + 2 * i;
        float4 a = r[0], b = r[1];
        Hit hit;
        float tmax = b.w;
        bool ok = trace<false>(sc, true, v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), a.w, tmax, hit, st, cnt);
        inst_out[i] = ok ? hit.inst : 0xffffffffu;
        prim_out[i] = ok ? hit.prim : 0xffffffffu;
        if (t_out) t_out[i] = tmax;
        if (uv_out) { uv_out[2 * i] = ok ? hit.u : 0.0f; uv_out[2 * i + 1] = ok ? hit.v : 0.0f; }
    }
    flush_counters(sc, cnt, 0u, s_cnt);
}

// The same query as a persistent kernel: the grid is what the device holds at once, a wave takes the rays in segments of 64
// consecutive indices from the per-XCD cursors (SegmentFetch, fetching two segments ahead) and hands new rays to lanes whose
// ray is finished as soon as kQueryRefillLanes of them are idle, inside the traversal loop, instead of holding every lane
// until the wave's longest ray is done.  A ray's own sequence of steps is the sequential one either way: same results.
// It pays for rays that diverge and loses for coherent ones, whatever the refill threshold (measured, exact traversal,
// Grays/s plain -> persistent: rays from inside the heightfield 10M 1.37 -> 1.78, heightfield 1M 2.00 -> 2.48, soup 1M
// 0.52 -> 0.66; camera rays 2.99 -> 1.85, 3.01 -> 1.42, 0.86 -> 0.50; the ordered traversal, whose build at eight waves
// spills, loses everywhere), so the caller asks for it (cpt_tracer_set_query_kernel) and the default stays the plain kernel.
constexpr int kQueryRefillLanes = 16;      // 16 / 32 / 48 / 60 measured on rays from inside the 10M heightfield: 1.30 / 1.34 / 1.26 / 1.09 x the plain kernel (soup 1M: 1.27 / 1.27 / 1.10 / 0.88)
template <bool LARGE, bool FAST, bool WIDE>
__global__ void __launch_bounds__(kTraceBlock, WIDE ? kWideQueryWaves : kQueryWaves) k_first_hit_stream(SceneDev sc, const float *__restrict__ rays, uint32_t n, uint32_t *__restrict__ inst_out,
                                                            uint32_t *__restrict__ prim_out, float *__restrict__ t_out, float *__restrict__ uv_out, uint32_t *queue, uint32_t parity) {
    __shared__ uint2 s_stack[kQueryLdsEntries * kTraceBlock];
    __shared__ unsigned long long s_cnt[5];
    uint2 spill_entries[kSpillEntries];
    TraceStackT<kQueryLdsEntries, LARGE ? kQueryLeafWeightLargeScene : 1, FAST, WIDE> st(s_stack, spill_entries, sc.counters);
    LaneCounters cnt;
    const uint32_t lane = threadIdx.x & 63;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t *cursors = queue + parity * kCursorSetWords;
    zero_cursors(queue + (parity ^ 1u) * kCursorSetWords);      // the next launch's cursors
    const uint32_t segments = (n + kSegmentSlots - 1) / kSegmentSlots;

    SegmentFetchT<true> f;
    Traversal t;
    bool have = false;
    uint32_t index = 0;
    for (;;) {
        unsigned long long idle = __ballot(!have);
        if (!f.exhausted && (__popcll(idle) >= kQueryRefillLanes || idle == ~0ull)) {
            while (idle != 0ull) {
                if (f.remaining() == 0 && !f.advance(nullptr, cursors, segments, blockIdx.x, lane)) break;
                const uint32_t rank = __popcll(idle & lt_mask);
                const uint32_t take = min((uint32_t)__popcll(idle), f.remaining());
                const uint32_t entry = f.entry(rank);
                if (!have && rank < take && entry < n) {
                    index = entry;
                    const float4 *r = reinterpret_cast<const float4 *>(rays) + 2 * (size_t)index;
                    const float4 a = r[0], b = r[1];
                    t.template begin<FAST, WIDE>(sc, v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), a.w, b.w, cnt);
                    have = true;
                }
                f.consumed += take;
                idle = __ballot(!have);
            }
        }
        if (__ballot(have) == 0ull) break;
        t.template vote_step<false>(sc, st, cnt);
        if (have && !t.live()) {
            inst_out[index] = t.found ? t.hit.inst : 0xffffffffu;
            prim_out[index] = t.found ? t.hit.prim : 0xffffffffu;
            if (t_out) t_out[index] = t.tmax;
            if (uv_out) { uv_out[2 * (size_t)index] = t.found ? t.hit.u : 0.0f; uv_out[2 * (size_t)index + 1] = t.found ? t.hit.v : 0.0f; }
            have = false;
        }
    }
    flush_counters(sc, cnt, 0u, s_cnt);
}

template <bool LARGE, bool FAST, bool WIDE>
__global__ void __launch_bounds__(kTraceBlock, WIDE ? kWideQueryWaves : kQueryWaves) k_occlude(SceneDev sc, const float *__restrict__ rays, unsigned long long n, uint8_t *__restrict__ out) {
    __shared__ uint2 s_stack[kQueryLdsEntries * kTraceBlock];
    __shared__ unsigned long long s_cnt[5];
    uint2 spill_entries[kSpillEntries];
    TraceStackT<kQueryLdsEntries, LARGE ? kQueryLeafWeightLargeScene : 1, FAST, WIDE> st(s_stack, spill_entries, sc.counters);
    LaneCounters cnt;
    unsigned long long i = (unsigned long long)blockIdx.x * kTraceBlock + threadIdx.x;
    if (i < n) {
        const float4 *r = reinterpret_cast<const float4 *>(rays) + 2 * i;
        float4 a = r[0], b = r[1];
        Hit hit;
        float tmax = b.w;
        out[i] = trace<true>(sc, true, v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), a.w, tmax, hit, st, cnt) ? 1 : 0;
    }
    flush_counters(sc, cnt, 0u, s_cnt);
}

__global__ void __launch_bounds__(kTraceBlock) k_camera_rays(CameraDev cam, uint32_t W, uint32_t H, uint32_t spp, float *__restrict__ rays) {
    uint32_t i = blockIdx.x * kTraceBlock + threadIdx.x;
    if (i >= W * H) return;
    uint32_t px = i % W, py = i / W;
    Sampler s;
    uint32_t pixel_index;
    V3 o, d;
    pixel_ray(cam, W, H, px, py, spp, s, pixel_index, o, d);
    float4 *r = reinterpret_cast<float4 *>(rays) + 2 * (size_t)pixel_index;
    r[0] = make_float4(o.x, o.y, o.z, kRayTMin);
    r[1] = make_float4(d.x, d.y, d.z, kRayTMax);
}

// Self-test of div_by_recip against the hardware's correctly rounded division on pseudo-random
// operands drawn from the whole range the traversal's fast path admits (|d| in [2^-50, 2^50],
// a zero or |a| in [2^-74, 2^51]), with extra weight on mantissas that are hard for Newton-type
// division: all ones, powers of two, one ulp above or below a power of two.
__device__ inline float selftest_operand(uint32_t &state, int emin, int emax) {
    uint32_t r1 = rand_pcg(state), r2 = rand_pcg(state);
    uint32_t mant = r1 & 0x7fffffu;
    switch ((r2 >> 8) & 7u) {
        case 0: mant = 0x7fffffu; break;
        case 1: mant = 0u; break;
        case 2: mant = 1u; break;
        case 3: mant = 0x7ffffeu; break;
        case 4: mant &= 0x7ff000u; break;      // short mantissas (exact quotients are common)
        default: break;
    }
    int e = emin + (int)((r2 >> 16) % (uint32_t)(emax - emin + 1));
    uint32_t bits = ((r2 & 1u) << 31) | ((uint32_t)(e + 127) << 23) | mant;
    return __uint_as_float(bits);
}

__global__ void __launch_bounds__(256) k_div_selftest(uint32_t seed, uint32_t per_thread, unsigned long long *mismatches) {
    uint32_t state = rand_init_tea(blockIdx.x * 256u + threadIdx.x, seed);
    uint32_t bad = 0;
    for (uint32_t k = 0; k < per_thread; k++) {
        float d = selftest_operand(state, -50, 50);
        float a = selftest_operand(state, -74, 51);
        if ((rand_pcg(state) & 63u) == 0u) a = 0.0f;
        float r = 1.0f / d;
        float q = div_by_recip(a, d, r);
        float t = a / d;
        bad += __float_as_uint(q) != __float_as_uint(t) ? 1u : 0u;
    }
    if (bad) atomicAdd(mismatches, (unsigned long long)bad);
}

// Plain copy: the rate the memory system gives a kernel that does nothing but stream, to set next to the 8 TB/s nominal
// peak.  One 16-byte element per lane and one workgroup per 4 KB, no loop: measured against 2, 4 and 8 elements in flight
// per lane and against persistent grids, this shape is the fastest on MI355X (6.2 TB/s read + written; 4 in flight 5.6,
// 8 in flight 4.3, grid-stride loops 4.9-5.5), so a streaming kernel here wants many short workgroups, not long lanes.
__global__ void __launch_bounds__(256) k_stream_copy(const RawF4 *__restrict__ src, RawF4 *__restrict__ dst, unsigned long long n16) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 256ull + threadIdx.x;
    if (i < n16) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

// Vector-instruction issue rate of the device, measured: every wave runs `iterations` trips of 64 independent vector
// instructions on 16 accumulators (no memory access, no dependence shorter than 16 instructions) and notes the shader
// clock before and after.  MIX 0: v_fma_f32 only.  MIX 1: the kinds a slab test is made of (subtract, multiply,
// compare + select, max, min).  MIX 2: v_pk_fma_f32 only (two fp32 operations per lane and instruction, on register
// pairs).  The host sizes the grid so that a chosen number of waves shares each SIMD; the quotient
// instructions / time over the whole device is the ceiling a traversal kernel's instruction count is set against.
template <int MIX>
__global__ void __launch_bounds__(256) k_valu_rate(uint32_t iterations, float seed, float *sink, unsigned long long *cycles) {
    float a[16];
#pragma unroll
    for (int k = 0; k < 16; k++) a[k] = seed + (float)(threadIdx.x + k);
    const float x = 1.0f + seed * 1e-9f, y = seed * 1e-3f;
    typedef float Pair __attribute__((ext_vector_type(2)));
    Pair p[16];
#pragma unroll
    for (int k = 0; k < 16; k++) p[k] = Pair{a[k], a[k] + 0.25f};
    const Pair px = Pair{x, x}, py = Pair{y, y};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (uint32_t i = 0; i < iterations; i++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (MIX == 0) {
#pragma unroll
                for (int k = 0; k < 16; k++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(x), "v"(y));
            } else if (MIX == 2) {
#pragma unroll
                for (int k = 0; k < 16; k++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(px), "v"(py));
            } else {
#pragma unroll
                for (int k = 0; k < 16; k += 8) {
                    asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[k]) : "v"(x));
                    asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k + 1]) : "v"(x));
                    asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %1, %2, vcc" : "+v"(a[k + 2]) : "v"(a[k + 3]), "v"(y) : "vcc");
                    asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[k + 4]) : "v"(y));
                    asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[k + 5]) : "v"(x));
                    asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[k + 6]) : "v"(y));
                    asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k + 7]) : "v"(x));
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float total = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; k++) total += a[k] + (MIX == 2 ? p[k].x + p[k].y : 0.0f);
    if (total == 123.456f) sink[0] = total;            // keeps the accumulators alive; never true for the seeds used
    if ((threadIdx.x & 63u) == 0u) cycles[(blockIdx.x * 256u + threadIdx.x) >> 6] = t1 - t0;
}

// d_dst += d_src, one float4 (pixel) per lane and trip
__global__ void __launch_bounds__(256) k_film_add(RawF4 *__restrict__ dst, const RawF4 *__restrict__ src, unsigned long long pixels) {
    const unsigned long long stride = (unsigned long long)gridDim.x * 256ull;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256ull + threadIdx.x; i < pixels; i += stride) dst[i] = dst[i] + src[i];
}

// sums with the sample count in alpha -> mean radiance, alpha 1 (a pixel without samples stays as it is)
__global__ void __launch_bounds__(256) k_film_finish(RawF4 *__restrict__ film, unsigned long long pixels) {
    const unsigned long long stride = (unsigned long long)gridDim.x * 256ull;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256ull + threadIdx.x; i < pixels; i += stride) {
        RawF4 v = film[i];
        if (v.w > 0.0f) film[i] = RawF4{v.x / v.w, v.y / v.w, v.z / v.w, 1.0f};
    }
}

}  // namespace

int launch_film_add(float *dst, const float *src, uint64_t pixels, hipStream_t stream) {
    if (pixels == 0) return 0;
    const unsigned long long want = (pixels + 255) / 256;
    hipLaunchKernelGGL(k_film_add, dim3((uint32_t)(want > 8192 ? 8192 : want)), dim3(256), 0, stream, (RawF4 *)dst, (const RawF4 *)src, (unsigned long long)pixels);
    return (int)hipGetLastError();
}

int launch_film_finish(float *film, uint64_t pixels, hipStream_t stream) {
    if (pixels == 0) return 0;
    const unsigned long long want = (pixels + 255) / 256;
    hipLaunchKernelGGL(k_film_finish, dim3((uint32_t)(want > 8192 ? 8192 : want)), dim3(256), 0, stream, (RawF4 *)film, (unsigned long long)pixels);
    return (int)hipGetLastError();
}

int launch_stream_copy(const void *src, void *dst, uint64_t bytes, hipStream_t stream) {
    const unsigned long long n16 = bytes / 16;
    if (n16 == 0) return 0;
    const unsigned long long want = (n16 + 255) / 256;
    if (want > 0x7fffffffull) return (int)hipErrorInvalidValue;
    const uint32_t blocks = (uint32_t)want;
    hipLaunchKernelGGL(k_stream_copy, dim3(blocks), dim3(256), 0, stream, (const RawF4 *)src, (RawF4 *)dst, n16);
    return (int)hipGetLastError();
}

// `waves_per_simd` waves on every SIMD of the device: workgroups of four waves (one per SIMD of a CU), and so much LDS
// asked for per workgroup (64 KB at most) that no more than waves_per_simd of them fit the 160 KB of a CU; the grid holds
// just that many for every CU.  With one wave per SIMD two workgroups would fit; the dispatcher spreads a grid of one
// workgroup per CU, and the per-wave clock counts the kernel returns show whether a wave had its SIMD to itself.
int launch_valu_rate(int mix, int waves_per_simd, int compute_units, uint32_t iterations, float *sink, unsigned long long *cycles, hipStream_t stream) {
    if (waves_per_simd < 1 || waves_per_simd > 8 || compute_units < 1) return (int)hipErrorInvalidValue;
    uint32_t lds = (160u * 1024u / (uint32_t)waves_per_simd) & ~1023u;
    if (lds > 64u * 1024u) lds = 64u * 1024u;
    void (*kernel)(uint32_t, float, float *, unsigned long long *) = mix == 0 ? k_valu_rate<0> : mix == 1 ? k_valu_rate<1> : k_valu_rate<2>;
    hipLaunchKernelGGL(kernel, dim3((uint32_t)(compute_units * waves_per_simd)), dim3(256), lds, stream, iterations, 0.5f, sink, cycles);
    return (int)hipGetLastError();
}

int selftest_division(uint64_t pairs, uint32_t seed, unsigned long long *d_mismatches, hipStream_t stream) {
    const uint32_t per_thread = 4096;
    uint64_t threads = (pairs + per_thread - 1) / per_thread;
    uint32_t blocks = (uint32_t)((threads + 255) / 256);
    if (blocks == 0) return 0;
    hipLaunchKernelGGL(k_div_selftest, dim3(blocks), dim3(256), 0, stream, seed, per_thread, d_mismatches);
    return (int)hipGetLastError();
}

// ---- kernel builds by (material maps, large scene, traversal order, node records).  Both orders walk the two-level
// records when the scene has them, except the ordered batch queries: those are bound by arithmetic, and ordering two
// children before their children are looked at tests fewer boxes than ordering four grandchildren (heightfield 10M camera
// rays 3.41 against 2.84 Grays/s, soup 1M 1.06 against 0.95), while the bounce loops gain from half as many trips (Cornell
// box 1423 -> 1525 Mpaths/s, heightfield 10M 544 -> 556).  Integrators without an ordered build run exact.
template <class F> auto pick(bool v, F &&f) { return v ? f(std::true_type{}) : f(std::false_type{}); }

using RegenKernel = void (*)(SceneDev, RenderParams, uint32_t *, uint32_t);
RegenKernel regen_kernel(int variant, bool maps, bool dense, bool fast, bool wide, bool scene_in_lds) {
    return pick(maps, [&](auto M) {
        return pick(dense, [&](auto D) -> RegenKernel {
            constexpr bool m = decltype(M)::value, d = decltype(D)::value;
            if (variant == 2 && scene_in_lds && !dense && !fast && wide) return k_render_pool<m, false, false, true, true>;
            if (variant == 2) return fast ? (wide ? k_render_pool<m, d, true, true> : k_render_pool<m, d, true, false>) : wide ? k_render_pool<m, d, false, true> : k_render_pool<m, d, false, false>;
            if (variant == 1) return fast ? (wide ? k_render_flow<m, true, true> : k_render_flow<m, true, false>) : wide ? k_render_flow<m, false, true> : k_render_flow<m, false, false>;
            return wide ? k_render_regen<m, d, true> : k_render_regen<m, d, false>;
        });
    });
}

int launch_render(const SceneDev &sc, const RenderParams &p, hipStream_t stream) {
    uint32_t tiles = ((p.width + 15) / 16) * ((p.height + 15) / 16);
    if (tiles == 0) return 0;
    void (*kernel)(SceneDev, RenderParams) = pick(sc.textures != nullptr, [&](auto M) {
        return sc.wide_nodes ? k_render<decltype(M)::value, true> : k_render<decltype(M)::value, false>;
    });
    hipLaunchKernelGGL(kernel, dim3(tiles), dim3(kTraceBlock), 0, stream, sc, p);
    return (int)hipGetLastError();
}

uint32_t render_tile_count(uint32_t width, uint32_t height) { return tile_count(width, height); }

int launch_render_regen(const SceneDev &sc, const RenderParams &p, uint32_t *queue, uint32_t parity, uint32_t max_blocks, bool dense, int variant, hipStream_t stream) {
    uint32_t tiles = tile_count(p.width, p.height);
    if (tiles == 0) return 0;
    if (p.spp_count == 0 || (p.spp_count > 1 && (variant == 0 || (!p.order && !p.turn_in_alpha)))) return (int)hipErrorInvalidValue;
    if (p.turn_in_alpha && (p.spp_count > (1u << 23) || (p.sum_mode && (unsigned long long)p.alpha_base + p.spp_count > (1ull << 24)))) return (int)hipErrorInvalidValue;
    if ((unsigned long long)tiles * p.spp_count * kSegmentSlots > 0xffffffffull) return (int)hipErrorInvalidValue;
    unsigned long long want = ((unsigned long long)tiles * p.spp_count + kTraceBlock / 64 - 1) / (kTraceBlock / 64);
    uint32_t blocks = want < max_blocks ? (uint32_t)want : max_blocks;
    RegenKernel kernel = regen_kernel(variant, sc.textures != nullptr, dense, sc.fast_traversal != 0, sc.wide_nodes != 0, sc.lds_scene != nullptr);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kTraceBlock), 0, stream, sc, p, queue, parity);
    return (int)hipGetLastError();
}

int render_regen_resident_blocks(int device, bool dense, int variant, bool fast, bool wide, bool scene_in_lds) {
    int per_cu = 0, other = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)regen_kernel(variant, true, dense, fast, wide, scene_in_lds), kTraceBlock, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&other, (const void *)regen_kernel(variant, false, dense, fast, wide, scene_in_lds), kTraceBlock, 0) != hipSuccess || other < 1) other = 1;
    if (other < per_cu) per_cu = other;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 1) cus = 256;
    return per_cu * cus;
}

int launch_first_hit(const SceneDev &sc, const float *rays, uint64_t n, uint32_t *inst, uint32_t *prim, float *t, float *uv, hipStream_t stream) {
    if (n == 0) return 0;
    uint64_t blocks = (n + kTraceBlock - 1) / kTraceBlock;
    if (blocks > 0x7fffffffull) return (int)hipErrorInvalidValue;
    const unsigned long long count = n;
    const bool fast = sc.fast_traversal != 0, wide = sc.wide_nodes != 0;
    void (*kernel)(SceneDev, const float *, unsigned long long, uint32_t *, uint32_t *, float *, float *) = pick(sc.large_scene != 0, [&](auto L) {
        constexpr bool l = decltype(L)::value;
        return fast ? k_first_hit<l, true, false> : wide ? k_first_hit<l, false, true> : k_first_hit<l, false, false>;
    });
    hipLaunchKernelGGL(kernel, dim3((uint32_t)blocks), dim3(kTraceBlock), 0, stream, sc, rays, count, inst, prim, t, uv);
    return (int)hipGetLastError();
}

using StreamQueryKernel = void (*)(SceneDev, const float *, uint32_t, uint32_t *, uint32_t *, float *, float *, uint32_t *, uint32_t);
StreamQueryKernel first_hit_stream_kernel(bool large, bool fast, bool wide) {
    return pick(large, [&](auto L) -> StreamQueryKernel {
        constexpr bool l = decltype(L)::value;
        return fast ? k_first_hit_stream<l, true, false> : wide ? k_first_hit_stream<l, false, true> : k_first_hit_stream<l, false, false>;
    });
}

// n below 2^32 - 64 (segment numbers are 32 bits); queue: kRegenQueueWords words, zeroed once, parity alternating per launch
int launch_first_hit_stream(const SceneDev &sc, const float *rays, uint64_t n, uint32_t *inst, uint32_t *prim, float *t, float *uv, uint32_t *queue, uint32_t parity,
                            uint32_t max_blocks, hipStream_t stream) {
    if (n == 0) return 0;
    if (n >= 0xffffffc0ull || !queue || max_blocks == 0) return (int)hipErrorInvalidValue;
    const uint64_t want = (n + kTraceBlock - 1) / kTraceBlock;
    const uint32_t blocks = want < max_blocks ? (uint32_t)want : max_blocks;
    StreamQueryKernel kernel = first_hit_stream_kernel(sc.large_scene != 0, sc.fast_traversal != 0, sc.fast_traversal == 0 && sc.wide_nodes != 0);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kTraceBlock), 0, stream, sc, rays, (uint32_t)n, inst, prim, t, uv, queue, parity);
    return (int)hipGetLastError();
}

int first_hit_stream_resident_blocks(int device, bool large, bool fast, bool wide) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)first_hit_stream_kernel(large, fast, wide), kTraceBlock, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 1) cus = 256;
    return per_cu * cus;
}

int launch_occlude(const SceneDev &sc, const float *rays, uint64_t n, uint8_t *out, hipStream_t stream) {
    if (n == 0) return 0;
    uint64_t blocks = (n + kTraceBlock - 1) / kTraceBlock;
    if (blocks > 0x7fffffffull) return (int)hipErrorInvalidValue;
    const unsigned long long count = n;
    const bool fast = sc.fast_traversal != 0, wide = sc.wide_nodes != 0;
    void (*kernel)(SceneDev, const float *, unsigned long long, uint8_t *) = pick(sc.large_scene != 0, [&](auto L) {
        constexpr bool l = decltype(L)::value;
        return fast ? k_occlude<l, true, false> : wide ? k_occlude<l, false, true> : k_occlude<l, false, false>;
    });
    hipLaunchKernelGGL(kernel, dim3((uint32_t)blocks), dim3(kTraceBlock), 0, stream, sc, rays, count, out);
    return (int)hipGetLastError();
}

// Node records as the LDS copy of a small scene holds them: the kLdsNodeWords 16-byte words of each two-level record that a visit reads
__global__ void __launch_bounds__(kTraceBlock) k_lds_scene_nodes(RawF4 *__restrict__ dst, const RawF4 *__restrict__ wide, uint32_t words) {
    const uint32_t w = blockIdx.x * kTraceBlock + threadIdx.x;
    if (w < words) dst[w] = wide[8u * (w / kLdsNodeWords) + w % kLdsNodeWords];
}
int launch_lds_scene_nodes(float *dst, const PackedNode *wide, uint32_t records, hipStream_t stream) {
    const uint32_t words = records * kLdsNodeWords;
    if (words == 0) return 0;
    hipLaunchKernelGGL(k_lds_scene_nodes, dim3((words + kTraceBlock - 1) / kTraceBlock), dim3(kTraceBlock), 0, stream, reinterpret_cast<RawF4 *>(dst),
                       reinterpret_cast<const RawF4 *>(wide), words);
    return (int)hipGetLastError();
}

int launch_camera_rays(const CameraDev &cam, uint32_t W, uint32_t H, uint32_t spp, float *rays, hipStream_t stream) {
    uint64_t n = (uint64_t)W * H;
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_camera_rays, dim3((uint32_t)((n + kTraceBlock - 1) / kTraceBlock)), dim3(kTraceBlock), 0, stream, cam, W, H, spp, rays);
    return (int)hipGetLastError();
}

}  // namespace cpt

#ifdef CPT_STEP_STATS
// development build only: reads (and optionally clears) the step statistics of k_render_pool on the current device
extern "C" __attribute__((visibility("default"))) int cpt_debug_step_stats(unsigned long long *out18, int reset) {
    int status = (int)hipDeviceSynchronize();
    if (status == 0 && out18) status = (int)hipMemcpyFromSymbol(out18, HIP_SYMBOL(cpt::g_step_stats), sizeof(unsigned long long) * 18);
    if (status == 0 && reset) {
        unsigned long long zero[18] = {};
        status = (int)hipMemcpyToSymbol(HIP_SYMBOL(cpt::g_step_stats), zero, sizeof(zero));
    }
    return status;
}
#endif
