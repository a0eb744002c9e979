// One path per lane, kept in registers between rays: the state and the loop body shared by the
// persistent kernels (the lane-regenerating megakernel and the kernel that finishes the paths a
// ray-queue run leaves alive).
#pragma once
#include "integrate.h"

namespace cpt {

// Everything a path carries from one ray to the next (src/kernels/integrator/path.cu:9-83 keeps the
// same values in locals of Trace).
struct PathLane {
    bool alive = false, primary = false, last_specular = false;
    bool film_pending = false;          // the path is finished and waits for its pixel's earlier samples (several samples per launch)
    uint32_t pixel_index = 0, depth = 0;
    uint32_t pass = 0;                  // which of the launch's samples the path is: sample index - RenderParams::spp
    float last_pdf = 0.0f;
    Sampler sampler{0};
    V3 ro = v3(0.0f), rd = v3(0.0f), throughput = v3(0.0f), color = v3(0.0f);
};

// One trip for the whole wave: closest-hit trace for every live lane (camera rays and bounce rays
// mixed), shading + light sampling, any-hit trace for the lanes with a shadow ray, BSDF sampling,
// film write for the lanes whose path ended.  Every lane of the wave must call it.
template <bool MAPS, class Stack>
__device__ inline void path_trip(const SceneDev &sc, const RenderParams &p, PathLane &l, Stack &st, LaneCounters &cnt) {
    // ---- extend
    Hit hit;
    float tmax = kRayTMax;
    const bool got = trace<false>(sc, l.alive, l.ro, l.rd, kRayTMin, tmax, hit, st, cnt);

    // ---- shade, pick a light sample (src/kernels/integrator/path.cu:9-48, :57-80)
    bool finish = false, need_shadow = false;
    V3 pos = v3(0.0f), normal = v3(0.0f, 0.0f, 1.0f), ldir = v3(0.0f), lweight = v3(0.0f);
    float lpdf = 0.0f, ldist = 0.0f;
    uint32_t material = 0;
    V2 uv = V2{0.0f, 0.0f};
    if (l.alive) {
        if (!got) {
            finish = true;
        } else {
            Surface surface = shading_surface<MAPS>(sc, hit, cnt);
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
                pos = surface.vertex.position;
                normal = surface.vertex.normal;
                material = sc.instances[hit.inst].material;
                if (MAPS) uv = surface.vertex.texcoord;
                if (!bsdf_is_delta(surface.bsdf)) {
                    float r1 = l.sampler.next1d();
                    uint32_t index = min(f2u_sat(r1 * (float)sc.num_lights), sc.num_lights - 1);
                    V2 r2 = l.sampler.next2d();
                    LightSample ls = light_sample<MAPS>(sc, sc.lights[index], pos, r2);
                    ls.pdf *= sc.light_pick_pdf;
                    ls.weight = ls.weight / sc.light_pick_pdf;
                    if (ls.pdf > 0.0f) { need_shadow = true; ldir = ls.dir; lweight = ls.weight; lpdf = ls.pdf; ldist = ls.dist; }
                }
            }
        }
    }

    // ---- shadow rays
    Hit dummy;
    float smax = ldist - kShadowRayEps;
    const bool occluded = trace<true>(sc, need_shadow, pos, ldir, kRayTMin, smax, dummy, st, cnt);

    // ---- light contribution and next direction (path.cu:38-55)
    if (l.alive && !finish) {
        const BsdfDev bsdf = MAPS ? material_at(sc.materials[material], sc.textures, sc.srgb_to_linear, uv) : sc.materials[material];
        Frame frame = Frame::from_normal(normal);
        V3 wo = frame.to_local(-l.rd);
        if (need_shadow && !occluded) {
            V3 wi = frame.to_local(ldir);
            float bp = bsdf_pdf(bsdf, wo, wi);
            float mis_weight = power_heuristic(lpdf, bp);
            l.color = l.color + l.throughput * mis_weight * lweight * bsdf_eval(bsdf, wo, wi);
        }
        float b1 = l.sampler.next1d();
        V2 b2 = l.sampler.next2d();
        BsdfSample bs = bsdf_sample(bsdf, wo, b1, b2);
        if (bs.pdf == 0.0f) {
            finish = true;
        } else {
            l.throughput = l.throughput * bs.weight;
            l.ro = pos;
            l.rd = frame.to_world(bs.wi);
            l.last_pdf = bs.pdf;
            l.last_specular = bs.lobe == kLobeSpecular;
        }
    }

    // ---- film (path.cu:99-104)
    if (l.alive && finish) {
        film_accumulate(p, l.pixel_index, l.color);
        l.alive = false;
    }
}

}  // namespace cpt
