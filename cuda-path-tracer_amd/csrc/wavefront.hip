// Wavefront (ray-queue) variant of the bounce loop for gfx950.
//
// The path state lives in HBM (structure of arrays, 16-byte fields) and one sample per pixel is
// rendered by a short sequence of kernels per bounce:
//     generate -> [ extend -> shade -> shadow -> reset ] x (max_depth + 1)
// extend / shadow are persistent traversal kernels: a lane whose ray is finished takes the next
// ray from the queue INSIDE the traversal loop (ballot + popcount prefix over idle lanes), so the
// wave-voted node step keeps most of its 64 lanes busy however uneven the rays are.  shade runs
// the integrator between two rays of a path with the reference's operation and random-draw order
// (src/kernels/integrator/path.cu:9-83), so a path computes bit for bit what it computes in the
// megakernels; only the order in which different paths are processed changes.
#include "wavefront.h"

#include <hip/hip_runtime.h>

#include "path_lane.h"
#include "segment_fetch.h"

#include <type_traits>

namespace cpt {

namespace {

constexpr uint32_t kPrimaryBit = 0x80000000u;
constexpr uint32_t kSpecularBit = 0x40000000u;
constexpr uint32_t kPixelMask = 0x3fffffffu;
constexpr int kRefillIdleLanes = 16;       // refill as soon as this many lanes of a wave are idle
// Waves per SIMD and LDS stack entries per lane of the queue traversal kernels, two builds.  They
// keep no path state in registers and fit 6 waves per SIMD with 12 entries (24 KB per workgroup).
// On scenes whose BVH misses the vector L1 the extra waves hide the misses (heightfield 10M:
// 229 -> 259 Mpaths/s); on cornell-box-glass, where every node is an L1 hit, they only add
// contention (3.8 -> 5.4 ms per sample), so small scenes keep 4 waves and 16 entries.
template <bool DENSE> struct QueueTraceShape { static constexpr int kWaves = DENSE ? 6 : 4, kLdsEntries = DENSE ? 12 : 16; };

// one lane per pixel slot (8x8 tiles): camera ray + fresh path state, src/kernels/integrator/path.cu:86-96
__global__ void __launch_bounds__(256) k_wf_generate(SceneDev sc, RenderParams p, WavefrontBuffers wb) {
    __shared__ unsigned long long s_cnt[5];
    const uint32_t slot = blockIdx.x * 256 + threadIdx.x;
    const uint32_t tiles_x = (p.width + 7) / 8;
    const uint32_t tile = slot >> 6, k = slot & 63u;
    const uint32_t px = (tile % tiles_x) * 8 + (k & 7u), py = (tile / tiles_x) * 8 + (k >> 3);
    const bool valid = slot < wb.capacity && px < p.width && py < p.height;
    if (valid) {
        Sampler sampler;
        uint32_t pixel_index;
        V3 ro, rd;
        pixel_ray(sc.camera, p.width, p.height, px, py, p.spp, sampler, pixel_index, ro, rd);
        wb.ray_o[slot] = make_float4(ro.x, ro.y, ro.z, __uint_as_float(pixel_index | kPrimaryBit));
        wb.ray_d[slot] = make_float4(rd.x, rd.y, rd.z, __uint_as_float(sampler.state));
        wb.thr[slot] = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(0u));
        wb.col[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    if (tile < wb.segments) segment_publish(wb.queue, tile, valid, slot);
    flush_counters(sc, LaneCounters{}, valid ? 1u : 0u, s_cnt);
}

// Persistent traversal over the queued paths.  ANY = false: closest hit of the path's current
// ray, result to wb.hit.  ANY = true: the path's shadow ray; an unoccluded ray adds the pending
// light contribution to the path's colour, and a path that was only waiting for this ray is
// written to the film.
template <bool ANY, bool DENSE, bool WIDE>
__global__ void __launch_bounds__(kTraceBlock, WIDE ? 5 : QueueTraceShape<DENSE>::kWaves) k_wf_trace(SceneDev sc, RenderParams p, WavefrontBuffers wb) {
    __shared__ uint2 s_stack[QueueTraceShape<DENSE>::kLdsEntries * kTraceBlock];
    __shared__ unsigned long long s_cnt[5];
    uint2 spill_entries[kSpillEntries];
    TraceStackT<QueueTraceShape<DENSE>::kLdsEntries, DENSE ? kLeafWeightLargeScene : 1, false, WIDE> st(s_stack, spill_entries, sc.counters);
    LaneCounters cnt;
    const uint32_t lane = threadIdx.x & 63;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const uint32_t *queue = ANY ? wb.shadow_queue : wb.queue;
    uint32_t *cursors = wb.control + (ANY ? kControlShadowCursors : kControlExtendCursors);
    // the shadow pass of the previous bounce is over, the one of this bounce has not begun
    if (!ANY) zero_cursors(wb.control + kControlShadowCursors);

    SegmentFetchT<true> f;
    Traversal t;
    bool have = false;
    uint32_t path = 0;
    for (;;) {
        unsigned long long idle = __ballot(!have);
        if (!f.exhausted && (__popcll(idle) >= kRefillIdleLanes || idle == ~0ull)) {
            while (idle != 0ull) {
                if (f.remaining() == 0 && !f.advance(queue, cursors, wb.segments, blockIdx.x, lane)) break;
                const uint32_t rank = __popcll(idle & lt_mask);
                const uint32_t take = min((uint32_t)__popcll(idle), f.remaining());
                const uint32_t entry = f.entry(rank);
                if (!have && rank < take) {
                    path = entry;
                    if (ANY) {
                        float4 o = wb.sh_o[path], d = wb.sh_d[path];
                        t.template begin<false, WIDE>(sc, v3(o.x, o.y, o.z), v3(d.x, d.y, d.z), kRayTMin, o.w, cnt);
                    } else {
                        float4 o = wb.ray_o[path], d = wb.ray_d[path];
                        t.template begin<false, WIDE>(sc, v3(o.x, o.y, o.z), v3(d.x, d.y, d.z), kRayTMin, kRayTMax, cnt);
                    }
                    have = true;
                }
                f.consumed += take;
                idle = __ballot(!have);
            }
        }
        if (__ballot(have) == 0ull) break;
        t.template vote_step<ANY>(sc, st, cnt);
        if (have && !t.live()) {
            if (ANY) {
                float4 c = wb.col[path];
                if (!t.found) {
                    float4 pend = wb.pending[path];
                    c.x = c.x + pend.x; c.y = c.y + pend.y; c.z = c.z + pend.z;
                    wb.col[path] = c;
                }
                if (__float_as_uint(wb.sh_d[path].w) != 0u) film_accumulate(p, __float_as_uint(wb.ray_o[path].w) & kPixelMask, v3(c.x, c.y, c.z));
            } else {
                wb.hit[path] = t.found ? make_float4(__uint_as_float(t.hit.inst), __uint_as_float(t.hit.prim), t.hit.u, t.hit.v)
                                       : make_float4(__uint_as_float(0xffffffffu), 0.0f, 0.0f, 0.0f);
            }
            have = false;
        }
    }
    flush_counters(sc, cnt, 0u, s_cnt);
}

// Everything the integrator does between two rays of one path (path.cu:12-80 in the order the
// statements run for one path): primary-hit special cases, emitter hit with MIS, Russian roulette,
// depth test, light sample (shadow ray + pending contribution), BSDF sample (next ray).
template <bool MAPS>
__global__ void __launch_bounds__(256) k_wf_shade(SceneDev sc, RenderParams p, WavefrontBuffers wb, uint32_t bounce_id) {
    __shared__ unsigned long long s_cnt[5];
    LaneCounters cnt;
    zero_cursors(wb.control + kControlExtendCursors);       // the extend pass of this bounce is over
    const uint32_t seg = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (seg < wb.segments) {
        uint32_t path = wb.queue[seg * kSegmentSlots + lane];
        const bool valid = path != kNoEntry;
        bool cont = false, need_shadow = false;
        if (valid) {
            float4 fo = wb.ray_o[path], fd = wb.ray_d[path], ft = wb.thr[path], fc = wb.col[path], fh = wb.hit[path];
            uint32_t tag = __float_as_uint(fo.w);
            const uint32_t pixel_index = tag & kPixelMask;
            bool primary = (tag & kPrimaryBit) != 0, last_specular = (tag & kSpecularBit) != 0;
            Sampler sampler{__float_as_uint(fd.w)};
            V3 ro = v3(fo.x, fo.y, fo.z), rd = v3(fd.x, fd.y, fd.z);
            V3 throughput = v3(ft.x, ft.y, ft.z), color = v3(fc.x, fc.y, fc.z);
            uint32_t depth = __float_as_uint(ft.w);
            float last_pdf = fc.w;
            bool finish = false;
            V3 contrib = v3(0.0f), ldir = v3(0.0f), pos = v3(0.0f);
            float smax = 0.0f;

            Hit hit;
            hit.inst = __float_as_uint(fh.x); hit.prim = __float_as_uint(fh.y); hit.u = fh.z; hit.v = fh.w;
            if (hit.inst == 0xffffffffu) {
                finish = true;
            } else {
                Surface surface = shading_surface<MAPS>(sc, hit, cnt);
                if (primary) {
                    primary = false;
                    if (p.channel == 1) { color = surface.vertex.normal * 0.5f + v3(0.5f); finish = true; }
                    else { color = surface.bsdf.emission; if (!is_zero(color)) finish = true; }
                } else if (!is_zero(surface.bsdf.emission) && dot(rd, surface.vertex.normal) < 0.0f) {
                    float mis_weight = 1.0f;
                    if (!last_specular) {
                        float light_pdf = surface.vertex.pdf * path_to_solid_angle_jacobian(ro, surface.vertex.position, surface.vertex.normal) *
                                          sc.light_pick_pdf;
                        mis_weight = power_heuristic(last_pdf, light_pdf);
                    }
                    color = color + throughput * mis_weight * surface.bsdf.emission;
                    finish = true;
                } else {
                    float rr_prop = gmin(gmax(luminance(throughput), 0.01f), 0.95f);
                    if (sampler.next1d() > rr_prop) finish = true;
                    else { throughput = throughput / rr_prop; depth++; }
                }
                if (!finish && !(depth < p.max_depth)) finish = true;
                if (!finish) {
                    pos = surface.vertex.position;
                    Frame frame = Frame::from_normal(surface.vertex.normal);
                    V3 wo = frame.to_local(-rd);
                    if (!bsdf_is_delta(surface.bsdf)) {
                        float r1 = sampler.next1d();
                        uint32_t index = min(f2u_sat(r1 * (float)sc.num_lights), sc.num_lights - 1);
                        V2 r2 = sampler.next2d();
                        LightSample ls = light_sample<MAPS>(sc, sc.lights[index], pos, r2);
                        ls.pdf *= sc.light_pick_pdf;
                        ls.weight = ls.weight / sc.light_pick_pdf;
                        if (ls.pdf > 0.0f) {
                            need_shadow = true;
                            ldir = ls.dir;
                            smax = ls.dist - kShadowRayEps;
                            V3 wi = frame.to_local(ls.dir);
                            float bp = bsdf_pdf(surface.bsdf, wo, wi);
                            float mis_weight = power_heuristic(ls.pdf, bp);
                            contrib = throughput * mis_weight * ls.weight * bsdf_eval(surface.bsdf, wo, wi);
                        }
                    }
                    float b1 = sampler.next1d();
                    V2 b2 = sampler.next2d();
                    BsdfSample bs = bsdf_sample(surface.bsdf, wo, b1, b2);
                    if (bs.pdf == 0.0f) {
                        finish = true;
                    } else {
                        throughput = throughput * bs.weight;
                        ro = pos;
                        rd = frame.to_world(bs.wi);
                        last_pdf = bs.pdf;
                        last_specular = bs.lobe == kLobeSpecular;
                    }
                }
            }
            cont = !finish;
            tag = pixel_index | (primary ? kPrimaryBit : 0u) | (last_specular ? kSpecularBit : 0u);
            if (need_shadow) {
                wb.sh_o[path] = make_float4(pos.x, pos.y, pos.z, smax);
                wb.sh_d[path] = make_float4(ldir.x, ldir.y, ldir.z, __uint_as_float(finish ? 1u : 0u));
                wb.pending[path] = make_float4(contrib.x, contrib.y, contrib.z, 0.0f);
            }
            if (cont || need_shadow) {
                wb.ray_o[path] = make_float4(ro.x, ro.y, ro.z, __uint_as_float(tag));
                wb.col[path] = make_float4(color.x, color.y, color.z, last_pdf);
            }
            if (cont) {
                wb.ray_d[path] = make_float4(rd.x, rd.y, rd.z, __uint_as_float(sampler.state));
                wb.thr[path] = make_float4(throughput.x, throughput.y, throughput.z, __uint_as_float(depth));
            }
            if (finish && !need_shadow) film_accumulate(p, pixel_index, color);
        }
        // every lane has read its entry of the old list by now (one wave, program order)
        const uint32_t live = segment_publish(wb.queue, seg, cont, path);
        segment_publish(wb.shadow_queue, seg, need_shadow, path);
        if (live != 0 && lane == 0) wb.control[kControlLiveMarker] = bounce_id;    // same value from every wave
    }
    flush_counters(sc, cnt, 0u, s_cnt);
}

// Finishes what the queue still holds: lanes load a queued path's state and run it to its end
// in registers, picking up another queued path whenever one ends.
template <bool MAPS, bool WIDE>
__global__ void __launch_bounds__(kTraceBlock, 4) k_wf_finish(SceneDev sc, RenderParams p, WavefrontBuffers wb) {
    __shared__ uint2 s_stack[kLdsStackEntries * kTraceBlock];
    __shared__ unsigned long long s_cnt[5];
    uint2 spill_entries[kSpillEntries];
    TraceStackT<kLdsStackEntries, 1, false, WIDE> st(s_stack, spill_entries, sc.counters);
    LaneCounters cnt;
    const uint32_t lane = threadIdx.x & 63;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t *cursors = wb.control + kControlFinishCursors;
    SegmentFetchT<false> f;
    PathLane l;
    for (;;) {
        unsigned long long idle = __ballot(!l.alive);
        while (idle != 0ull && !f.exhausted) {
            if (f.remaining() == 0 && !f.advance(wb.queue, cursors, wb.segments, blockIdx.x, lane)) break;
            const uint32_t rank = __popcll(idle & lt_mask);
            const uint32_t take = min((uint32_t)__popcll(idle), f.remaining());
            const uint32_t path = f.entry(rank);
            if (!l.alive && rank < take) {
                const float4 fo = wb.ray_o[path], fd = wb.ray_d[path], ft = wb.thr[path], fc = wb.col[path];
                const uint32_t tag = __float_as_uint(fo.w);
                l.alive = true;
                l.primary = (tag & kPrimaryBit) != 0;
                l.last_specular = (tag & kSpecularBit) != 0;
                l.pixel_index = tag & kPixelMask;
                l.depth = __float_as_uint(ft.w);
                l.last_pdf = fc.w;
                l.sampler.state = __float_as_uint(fd.w);
                l.ro = v3(fo.x, fo.y, fo.z); l.rd = v3(fd.x, fd.y, fd.z);
                l.throughput = v3(ft.x, ft.y, ft.z); l.color = v3(fc.x, fc.y, fc.z);
            }
            f.consumed += take;
            idle = __ballot(!l.alive);
        }
        if (__ballot(l.alive) == 0ull) break;
        path_trip<MAPS>(sc, p, l, st, cnt);
    }
    flush_counters(sc, cnt, 0u, s_cnt);
}

}  // namespace

size_t wavefront_slots(uint32_t width, uint32_t height) { return (size_t)((width + 7) / 8) * ((height + 7) / 8) * kSegmentSlots; }

static int resident_blocks_of(const void *const *kernels, int count, int device) {
    int best = 1 << 30, cus = 0;
    for (int i = 0; i < count; i++) {
        int a = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, kernels[i], kTraceBlock, 0) != hipSuccess || a < 1) a = 1;
        if (a < best) best = a;
    }
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 1) cus = 256;
    return best * cus;
}

template <class F> static auto pick(bool v, F &&f) { return v ? f(std::true_type{}) : f(std::false_type{}); }
using QueueTraceKernel = void (*)(SceneDev, RenderParams, WavefrontBuffers);
static QueueTraceKernel trace_kernel(bool any, bool dense, bool wide) {
    return pick(any, [&](auto A) {
        return pick(dense, [&](auto D) -> QueueTraceKernel {
            constexpr bool a = decltype(A)::value, d = decltype(D)::value;
            return wide ? k_wf_trace<a, d, true> : k_wf_trace<a, d, false>;
        });
    });
}
static QueueTraceKernel finish_kernel(bool maps, bool wide) {
    return pick(maps, [&](auto M) -> QueueTraceKernel { return wide ? k_wf_finish<decltype(M)::value, true> : k_wf_finish<decltype(M)::value, false>; });
}

int wavefront_resident_blocks(int device, bool dense, bool wide) {
    const void *kernels[2] = {(const void *)trace_kernel(false, dense, wide), (const void *)trace_kernel(true, dense, wide)};
    return resident_blocks_of(kernels, 2, device);
}

int wavefront_finish_resident_blocks(int device, bool wide) {
    const void *kernels[2] = {(const void *)finish_kernel(false, wide), (const void *)finish_kernel(true, wide)};
    return resident_blocks_of(kernels, 2, device);
}

static uint32_t persistent_blocks(const WavefrontBuffers &wb, uint32_t resident_blocks) {
    uint32_t want = (wb.segments + kTraceBlock / 64 - 1) / (kTraceBlock / 64);
    uint32_t blocks = want < resident_blocks ? want : resident_blocks;
    return blocks ? blocks : 1;
}

int wavefront_begin_sample(const SceneDev &sc, const RenderParams &p, const WavefrontBuffers &wb, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(wb.control, 0, kControlWords * sizeof(uint32_t), stream);
    if (e != hipSuccess) return (int)e;
    uint32_t blocks = (wb.capacity + 255) / 256;
    hipLaunchKernelGGL(k_wf_generate, dim3(blocks), dim3(256), 0, stream, sc, p, wb);
    return (int)hipGetLastError();
}

int wavefront_bounce(const SceneDev &sc, const RenderParams &p, const WavefrontBuffers &wb, uint32_t bounce_id, uint32_t resident_blocks, bool dense,
                     hipStream_t stream) {
    const uint32_t blocks = persistent_blocks(wb, resident_blocks);
    const uint32_t shade_blocks = (wb.segments + 3) / 4;
    hipLaunchKernelGGL(trace_kernel(false, dense, sc.wide_nodes != 0), dim3(blocks), dim3(kTraceBlock), 0, stream, sc, p, wb);
    if (sc.textures) hipLaunchKernelGGL(k_wf_shade<true>, dim3(shade_blocks), dim3(256), 0, stream, sc, p, wb, bounce_id);
    else hipLaunchKernelGGL(k_wf_shade<false>, dim3(shade_blocks), dim3(256), 0, stream, sc, p, wb, bounce_id);
    hipLaunchKernelGGL(trace_kernel(true, dense, sc.wide_nodes != 0), dim3(blocks), dim3(kTraceBlock), 0, stream, sc, p, wb);
    return (int)hipGetLastError();
}

int wavefront_finish(const SceneDev &sc, const RenderParams &p, const WavefrontBuffers &wb, uint32_t resident_blocks, hipStream_t stream) {
    hipLaunchKernelGGL(finish_kernel(sc.textures != nullptr, sc.wide_nodes != 0), dim3(persistent_blocks(wb, resident_blocks)), dim3(kTraceBlock), 0, stream, sc, p, wb);
    return (int)hipGetLastError();
}

}  // namespace cpt
