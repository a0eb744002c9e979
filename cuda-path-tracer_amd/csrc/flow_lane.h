// One path per lane with the traversal state kept across the shading step: the wave leaves the
// traversal loop as soon as enough of its lanes have a finished extension ray, shades those lanes
// (new directions, new camera rays for ended paths) and goes back in, while the other lanes simply
// resume the ray they were walking.  A lane traces its shadow ray and then its extension ray
// inside the same loop, so the per-path order of operations is still the one of Trace
// (src/kernels/integrator/path.cu:9-83): light sample, shadow ray, BSDF sample, extension ray.
#pragma once
#include "path_lane.h"

namespace cpt {

// Traversal of trace.h with the ray kind chosen per lane (closest hit for extension rays, any hit
// for shadow rays).  The world-space ray is not kept here: its origin and direction are the
// caller's path state.
struct Flow {
    uint32_t cur = kNoNode;
    uint32_t sp = 0, sp_base = 0, inst_id = 0;
    bool in_blas = false, found = false, world_fast = false;
    uint32_t inst_pos = 0, hit_pos = 0, hit_slot = 0;   // ordered traversal: top-level leaf position of the instance being walked; of the hit held; its sorted-leaf slot
    float tmax = 0.0f;
    uint32_t top_ref = kNoNode;        // the entry on top of the stack, when it is held here instead of in the stack's memory
    float top_t0 = 0.0f;
    V3 world_r = v3(0.0f);
    RayFrame f;
    const PackedNode *nodes = nullptr;
    const TriRecord *tris = nullptr;
    uint32_t nodes_at = 0, tris_at = 0;       // the same two as 16-byte word positions in the LDS copy of the records (kernels that hold one)
    Hit hit;

    __device__ inline bool walking() const { return cur != kNoNode; }

    // What a ray needs before its first trip: the reciprocal direction (when the fast division applies)
    // and the test of the top-level root box, which AccelTop::Intersect / Occlude do when they pop node 0.
    struct Prepared { V3 r; bool fast, enters; };
    template <bool FAST = false>
    static __device__ inline Prepared prepare(const SceneDev &sc, V3 ow, V3 dw, float reach) {
        const AccelRoot troot = *sc.top_root;
        RayFrame w;
        w.set_for<FAST>(ow, dw, troot.flags == 0);
        float t0;
        Prepared q;
        q.r = w.r;
        q.fast = w.fast;
        q.enters = slab_of<FAST>(troot.bmin[0], troot.bmin[1], troot.bmin[2], troot.bmax[0], troot.bmax[1], troot.bmax[2], w, kRayTMin, t0) && t0 < reach;
        return q;
    }
    // Begins the ray (ow, dw) of reach `reach`; afterwards walking() is false if it missed everything at once.
    template <bool WIDE = false>
    __device__ inline void begin(const SceneDev &sc, V3 ow, V3 dw, float reach, const Prepared &q, LaneCounters &cnt, bool count = true) {
        if (count) cnt.rays++;
        tmax = reach;
        sp = 0; sp_base = 0;
        top_ref = kNoNode;
        in_blas = false; found = false;
        nodes = WIDE ? sc.top_wide : sc.top_nodes;
        nodes_at = sc.lds_top;
        f.o = ow; f.d = dw; f.r = q.r; f.fast = q.fast;
        world_r = q.r;
        world_fast = q.fast;
        cur = q.enters ? sc.top_root->ref : kNoNode;
    }

    // One trip for the wave.  world(ow, dw) gives the lane's world-space ray (asked for only when the lane
    // enters or leaves an instance); any: it is a shadow ray (AccelTop::Occlude, accel.cuh:152-173) and not
    // an extension ray (AccelTop::Intersect, accel.cuh:124-150).  A lane whose ray ends is left with
    // walking() false.  Returns false, in every lane, when no lane of the wave had a step to take.
    template <class Stack, class World>
    __device__ inline bool step(const SceneDev &sc, Stack &st, LaneCounters &cnt, World world, bool any) {
        constexpr bool FAST = Stack::kFast, WIDE = Stack::kWide;        // the traversals of trace.h (Traversal), same rules
        // The entry on top of the stack stays in two registers: the last subtree pushed at a node is the first one popped,
        // and a pop that finds it there does not wait for LDS; it moves to the stack's memory when the next one is pushed
        // (Cornell box 1577 -> 1675 Mpaths/s, heightfield 10M 574 -> 582).
        auto push = [&](uint32_t ref, float t0) {
            if (top_ref != kNoNode) st.push(sp - 1, top_ref, top_t0);
            top_ref = ref; top_t0 = t0; sp++;
        };
        auto pop = [&]() -> uint2 {
            --sp;
            if (top_ref != kNoNode) { const uint2 e = make_uint2(top_ref, __float_as_uint(top_t0)); top_ref = kNoNode; return e; }
            return st.at(sp);
        };
        const bool alive = cur != kNoNode;
        const bool is_node = alive && !(cur & kLeafBit);
        const bool is_tri = alive && (cur & kLeafBit) && in_blas;
        const bool is_inst = alive && (cur & kLeafBit) && !in_blas;
        const int cn = lanes_in(__ballot(is_node)), ct = lanes_in(__ballot(is_tri)), ci = lanes_in(__ballot(is_inst));
        bool need_pop = false;
#ifdef CPT_STEP_STATS
        const unsigned long long stat_t0 = clock64();
#endif
        if (cn >= Stack::kLeafWeight * ct && cn >= Stack::kLeafWeight * ci) {
            if constexpr (WIDE) {
                if (is_node) {                          // two levels per record, as in Traversal::vote_step
                    float4 a0, a1, a2, a3, b0, b1, b2;       // the first half's last word has all four references; the second half's is not read
                    if constexpr (Stack::kSceneInLds) {
                        LdsF4 q = st.scene + nodes_at + kLdsNodeWords * cur;
                        a0 = load_f4(q, 0); a1 = load_f4(q, 1); a2 = load_f4(q, 2); a3 = load_f4(q, 3);
                        b0 = load_f4(q, 4); b1 = load_f4(q, 5); b2 = load_f4(q, 6);
                    } else {
                        GlobalF4 q = (GlobalF4)(nodes + 2 * (size_t)cur);
                        a0 = load_f4(q, 0); a1 = load_f4(q, 1); a2 = load_f4(q, 2); a3 = load_f4(q, 3);
                        b0 = load_f4(q, 4); b1 = load_f4(q, 5); b2 = load_f4(q, 6);
                    }
                    cnt.nodes++;
                    const uint32_t r0 = __float_as_uint(a3.x), r1 = __float_as_uint(a3.y), r2 = __float_as_uint(a3.z), r3 = __float_as_uint(a3.w);
                    if constexpr (FAST) {
                        const float inf = __uint_as_float(0x7f800000u);
                        float t[4];
                        uint32_t r[4] = {r0, r1, r2, r3};
                        const bool h0 = slab_fast(a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, f, kRayTMin, t[0]) && within_reach<true>(t[0], tmax, any);
                        const bool h1 = slab_fast(a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, f, kRayTMin, t[1]) && within_reach<true>(t[1], tmax, any);
                        const bool h2 = r2 != kNoNode && slab_fast(b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, f, kRayTMin, t[2]) && within_reach<true>(t[2], tmax, any);
                        const bool h3 = r3 != kNoNode && slab_fast(b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, f, kRayTMin, t[3]) && within_reach<true>(t[3], tmax, any);
                        if (!h0) { t[0] = inf; r[0] = kNoNode; }
                        if (!h1) { t[1] = inf; r[1] = kNoNode; }
                        if (!h2) { t[2] = inf; r[2] = kNoNode; }
                        if (!h3) { t[3] = inf; r[3] = kNoNode; }
                        order_slots(t, r);
                        if (r[3] != kNoNode) push(r[3], t[3]);
                        if (r[2] != kNoNode) push(r[2], t[2]);
                        if (r[1] != kNoNode) push(r[1], t[1]);
                        if (r[0] != kNoNode) cur = r[0]; else need_pop = true;
                    } else {
                    float t0, t1, t2, t3;
                    const bool h0 = slab(a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, f, kRayTMin, t0) && t0 < tmax;
                    const bool h1 = slab(a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, f, kRayTMin, t1) && t1 < tmax;       // a record has at least two slots
                    const bool h2 = r2 != kNoNode && slab(b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, f, kRayTMin, t2) && t2 < tmax;
                    const bool h3 = r3 != kNoNode && slab(b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, f, kRayTMin, t3) && t3 < tmax;
                    if (h3 && (h0 || h1 || h2)) push(r3, t3);
                    if (h2 && (h0 || h1)) push(r2, t2);
                    if (h1 && h0) push(r1, t1);
                    if (h0 || h1 || h2 || h3) cur = h0 ? r0 : h1 ? r1 : h2 ? r2 : r3;
                    else need_pop = true;
                    }
                }
            } else if (is_node) {
                GlobalF4 q = (GlobalF4)(nodes + cur);
                float4 q0 = load_f4(q, 0), q1 = load_f4(q, 1), q2 = load_f4(q, 2), q3 = load_f4(q, 3);
                cnt.nodes++;
                float t0l, t0r;
                if constexpr (FAST) {
                    const bool hl = slab_fast(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, f, kRayTMin, t0l) && within_reach<true>(t0l, tmax, any);
                    const bool hr = slab_fast(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, f, kRayTMin, t0r) && within_reach<true>(t0r, tmax, any);
                    const uint32_t lc = __float_as_uint(q3.x), rc = __float_as_uint(q3.y);
                    const bool left_first = t0l < t0r;
                    if (hl && hr) push(left_first ? rc : lc, left_first ? t0r : t0l);
                    if (hl || hr) cur = (hl && (left_first || !hr)) ? lc : rc;
                    else need_pop = true;
                } else {
                    // accel.cuh:84-91 pushes a left child that is hit and tests its entry distance against the reach when it
                    // is popped; the reach only shrinks, so one that is out of reach now never needs the stack, and one
                    // that would be popped at once (the right child is not walked) is walked without going through it
                    const bool hl = slab(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, f, kRayTMin, t0l) && t0l < tmax;
                    const bool hr = slab(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, f, kRayTMin, t0r) && t0r < tmax;
                    if (hl && hr) push(__float_as_uint(q3.x), t0l);
                    if (hl || hr) cur = __float_as_uint(hr ? q3.y : q3.x);
                    else need_pop = true;
                }
            }
        } else if (ct >= ci) {
            // A lane at a triangle goes on to the next pending entry when that is a triangle of the same mesh too
            // (the leaves under one two-level record), up to kLeafRun tests in one trip: its own order of tests and
            // pops is unchanged, and it does not wait for the vote between them.  On binary records one test per trip.
            if (is_tri) {
                auto exact_tri_test = [&](uint32_t slot, float &t, float &bv, float &bw, uint32_t &prim) -> bool {
                    if constexpr (Stack::kSceneInLds) {
                        LdsF4 q = st.scene + tris_at + 3u * slot;
                        return tri_test_words(load_f4(q, 0), load_f4(q, 1), load_f4(q, 2), f.o, f.d, kRayTMin, tmax, t, bv, bw, prim);
                    } else {
                        return tri_test(tris + slot, f.o, f.d, kRayTMin, tmax, t, bv, bw, prim);
                    }
                };
                for (int run = 0; run < (WIDE ? kLeafRun : 1); run++) {
                    float t, bv, bw;
                    uint32_t prim;
                    cnt.tris++;
                    bool ended = false;
                    if constexpr (FAST) {
                        bool tie;
                        const uint32_t slot = cur & ~kLeafBit;
                        if (tri_test_or_tie(tris + slot, f.o, f.d, kRayTMin, tmax, t, bv, bw, prim, tie)) {
                            if (any) {
                                if (!tie) { found = true; ended = true; }       // any hit is strictly inside the reach (accel.cuh:96-110)
                            } else {
                                const bool later = inst_pos > hit_pos || (inst_pos == hit_pos && slot > hit_slot);
                                if (!tie || !found || later) {
                                    found = true;
                                    tmax = t;
                                    hit.inst = inst_id; hit.prim = prim; hit.u = bv; hit.v = bw;
                                    hit_pos = inst_pos; hit_slot = slot;
                                }
                            }
                        }
                    } else if (exact_tri_test(cur & ~kLeafBit, t, bv, bw, prim)) {
                        found = true;
                        if (any) {
                            ended = true;
                        } else {
                            tmax = t;
                            hit.inst = inst_id; hit.prim = prim; hit.u = bv; hit.v = bw;
                        }
                    }
                    cur = kNoNode;
                    if (ended) break;
                    // the next pending entry of this mesh that is still within reach
                    while (sp != sp_base) {
                        const uint2 e = pop();
                        if (within_reach<FAST>(__uint_as_float(e.y), tmax, any)) { cur = e.x; break; }
                    }
                    if (cur == kNoNode) { need_pop = true; break; }       // the mesh is done: leave it below
                    if (!(cur & kLeafBit)) break;
                }
            }
        } else {
            if (is_inst) {
                inst_id = cur & ~kLeafBit;
                const InstanceDev &in = sc.instances[inst_id];
                const MeshDev &m = sc.meshes[in.mesh];
                nodes = WIDE ? m.wide : m.nodes;
                tris = m.tris;
                nodes_at = m.lds_nodes;
                tris_at = m.lds_tris;
                AccelRoot r = m.root;
                if constexpr (FAST) inst_pos = in.tlas_position;
                V3 ow, dw;
                world(ow, dw);
                enter_instance<FAST>(f, in, ow, dw, world_r, world_fast, r.flags == 0);
                in_blas = true;
                sp_base = sp;
                float t0;
                if (slab_of<FAST>(r.bmin[0], r.bmin[1], r.bmin[2], r.bmax[0], r.bmax[1], r.bmax[2], f, kRayTMin, t0) && within_reach<FAST>(t0, tmax, any)) cur = r.ref;
                else need_pop = true;
            }
        }
#ifdef CPT_STEP_STATS
        const unsigned long long stat_t1 = clock64();
        const int stat_pops = lanes_in(__ballot(need_pop));
#endif
        if (need_pop) {
            cur = kNoNode;
            for (;;) {
                if (in_blas && sp == sp_base) { in_blas = false; world(f.o, f.d); f.r = world_r; f.fast = world_fast; nodes = WIDE ? sc.top_wide : sc.top_nodes; nodes_at = sc.lds_top; }
                if (sp == 0) break;
                uint2 e = pop();
                if (within_reach<FAST>(__uint_as_float(e.y), tmax, any)) { cur = e.x; break; }
            }
        }
#ifdef CPT_STEP_STATS
        if (cnt.stats && cn + ct + ci != 0) {
            const unsigned long long stat_t2 = clock64();
            const int kind = (cn >= Stack::kLeafWeight * ct && cn >= Stack::kLeafWeight * ci) ? 0 : (ct >= ci ? 1 : 2);
            cnt.stats->time[kind] += stat_t1 - stat_t0; cnt.stats->trips[kind]++; cnt.stats->lanes[kind] += kind == 0 ? cn : kind == 1 ? ct : ci;
            cnt.stats->lanes[5] += cn + ct + ci;      // lanes with a ray, whatever its kind of step
            cnt.stats->time[3] += stat_t2 - stat_t1; cnt.stats->trips[3] += stat_pops != 0; cnt.stats->lanes[3] += stat_pops;
        }
#endif
        return cn + ct + ci != 0;
    }
};

// Path state beyond PathLane: the shadow ray in flight and what follows it
struct FlowLane {
    PathLane l;
    V3 shadow_dir = v3(0.0f), pending = v3(0.0f);     // pending: light contribution, added when the shadow ray ends unoccluded
    float shadow_tmax = 0.0f;
    bool has_shadow = false, has_ext = false;         // rays still to be traced for the current path vertex
    bool ext_traced = false;                          // Flow holds the result of this path's extension ray
};

// Shading for the lanes whose extension ray is finished (or whose path has nothing left to trace):
// the part of Trace between two traversals (path.cu:13-55 for the first hit, :57-80 + :27-55 later).
// Leaves the lane either ended (film written, l.alive false) or with has_shadow / has_ext set.
// emit(pos, dir, reach, contribution) is called for the shadow ray of this path vertex, if there is one, before
// the BSDF is sampled; it returns whether the ray has to be traced (false: the caller has dealt with it).
template <bool MAPS, class Emit>
__device__ inline void flow_shade(const SceneDev &sc, const RenderParams &p, FlowLane &s, const Flow &t, LaneCounters &cnt, Emit emit) {
    PathLane &l = s.l;
    bool finish = l.film_pending || !s.ext_traced || !t.found;
    if (!finish) {
        Surface surface = shading_surface<MAPS>(sc, t.hit, cnt);
        if (l.primary) {
            l.primary = false;
            if (p.channel == 1) { l.color = surface.vertex.normal * 0.5f + v3(0.5f); finish = true; }
            else { l.color = surface.bsdf.emission; if (!is_zero(l.color)) finish = true; }
        } else if (!is_zero(surface.bsdf.emission) && dot(l.rd, surface.vertex.normal) < 0.0f) {
            float mis_weight = 1.0f;
            if (!l.last_specular) {
                float light_pdf = surface.vertex.pdf * path_to_solid_angle_jacobian(l.ro, surface.vertex.position, surface.vertex.normal) *
                                  sc.light_pick_pdf;
                mis_weight = power_heuristic(l.last_pdf, light_pdf);
            }
            l.color = l.color + l.throughput * mis_weight * surface.bsdf.emission;
            finish = true;
        } else {
            float rr_prop = gmin(gmax(luminance(l.throughput), 0.01f), 0.95f);
            if (l.sampler.next1d() > rr_prop) finish = true;
            else { l.throughput = l.throughput / rr_prop; l.depth++; }
        }
        if (!finish && !(l.depth < p.max_depth)) finish = true;
        if (!finish) {
            const V3 pos = surface.vertex.position;
            Frame frame = Frame::from_normal(surface.vertex.normal);
            V3 wo = frame.to_local(-l.rd);
            s.has_shadow = false;
            if (!bsdf_is_delta(surface.bsdf)) {
                float r1 = l.sampler.next1d();
                uint32_t index = min(f2u_sat(r1 * (float)sc.num_lights), sc.num_lights - 1);
                V2 r2 = l.sampler.next2d();
                LightSample ls = light_sample<MAPS>(sc, sc.lights[index], pos, r2);
                ls.pdf *= sc.light_pick_pdf;
                ls.weight = ls.weight / sc.light_pick_pdf;
                if (ls.pdf > 0.0f) {
                    V3 wi = frame.to_local(ls.dir);
                    float bp = bsdf_pdf(surface.bsdf, wo, wi);
                    float mis_weight = power_heuristic(ls.pdf, bp);
                    s.has_shadow = emit(pos, ls.dir, ls.dist - kShadowRayEps, l.throughput * mis_weight * ls.weight * bsdf_eval(surface.bsdf, wo, wi));
                }
            }
            float b1 = l.sampler.next1d();
            V2 b2 = l.sampler.next2d();
            BsdfSample bs = bsdf_sample(surface.bsdf, wo, b1, b2);
            l.ro = pos;
            s.has_ext = bs.pdf != 0.0f;
            if (s.has_ext) {
                l.throughput = l.throughput * bs.weight;
                l.rd = frame.to_world(bs.wi);
                l.last_pdf = bs.pdf;
                l.last_specular = bs.lobe == kLobeSpecular;
            }
            s.ext_traced = false;
            if (!s.has_shadow && !s.has_ext) finish = true;
        }
    }
    if (finish) {
        s.has_shadow = false;
        s.has_ext = false;
        s.ext_traced = false;
        if (p.order || p.turn_in_alpha) {
            l.film_pending = !film_accumulate_in_order(p, l.pixel_index, l.pass, l.color);
            l.alive = l.film_pending;
        } else {
            film_accumulate(p, l.pixel_index, l.color);
            l.alive = false;
        }
    }
}

}  // namespace cpt
