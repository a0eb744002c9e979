// Pieces shared by the bounce-loop kernels (megakernels in render.hip, ray-queue kernels in
// wavefront.hip): shading surface, area-light sampling, film accumulation, counter flush.
#pragma once
#include "shading.h"
#include "trace.h"

namespace cpt {

struct Surface { Vertex vertex; BsdfDev bsdf; int32_t light; };     // vertex.texcoord is valid when the material reads it

// src/kernels/scene/instance.cuh:19-26.  MAPS = false is the variant for scenes without material maps: no
// texture coordinates, no per-hit material evaluation (kernels are built both ways, the host picks).
template <bool MAPS>
__device__ inline Surface shading_surface(const SceneDev &sc, const Hit &hit, LaneCounters &cnt) {
    cnt.shades++;
    const InstanceDev &in = sc.instances[hit.inst];
    const MeshDev &m = sc.meshes[in.mesh];
    Surface s;
    const BsdfDev &material = sc.materials[in.material];
    const bool mapped = MAPS && material_reads_texcoord(material);
    s.vertex = get_vertex(m, hit.prim, hit.u, hit.v, mapped);
    M4 tr = in.transform;
    vertex_transform_pdf(s.vertex, tr);
    s.vertex.position = xyz(mul(tr, v4(s.vertex.position, 1.0f)));
    s.vertex.normal = xyz(normalize4(mul(in.transform_it, v4(s.vertex.normal, 0.0f))));   // 4-component normalise, as the reference writes it
    s.bsdf = mapped ? material_at(material, sc.textures, sc.srgb_to_linear, s.vertex.texcoord) : material;
    s.light = in.light;
    return s;
}

struct LightSample { V3 dir; float pdf; V3 weight; float dist; };

// src/kernels/geometry/geometry.cuh:27-39 + src/kernels/light/geometry.cuh:17-30
template <bool MAPS>
__device__ inline LightSample light_sample(const SceneDev &sc, uint32_t inst_id, V3 pos, V2 rand) {
    const InstanceDev &in = sc.instances[inst_id];
    const MeshDev &m = sc.meshes[in.mesh];
    const BsdfDev &material = sc.materials[in.material];
    const bool mapped = MAPS && material.tex_emission != 0u;
    Vertex vert = sample_vertex(m, rand, mapped);
    M4 tr = in.transform;
    vertex_transform_pdf(vert, tr);
    vert.position = xyz(mul(tr, v4(vert.position, 1.0f)));
    vert.normal = xyz(normalize4(mul(in.transform_it, v4(vert.normal, 0.0f))));
    V3 vec = vert.position - pos;
    float dist_sqr = dot(vec, vec);
    float dist = sqrtf(dist_sqr);
    V3 dir = vec / dist;
    LightSample ls;
    ls.dir = dir;
    ls.dist = dist;
    float cos_theta = dot(-dir, vert.normal);
    ls.pdf = cos_theta > 0.0f ? vert.pdf * dist_sqr / cos_theta : 0.0f;
    V3 emission = material.emission;       // material.ptr->emission.At(vert.texcoord)
    if (mapped) emission = texture_fetch(sc.textures[material.tex_emission - 1], sc.srgb_to_linear, vert.texcoord.x, 1.0f - vert.texcoord.y);
    ls.weight = emission / ls.pdf;
    return ls;
}

constexpr float kRayTMin = 0.0001f;          // src/kernels/basic/ray.cuh:8-10
constexpr float kRayTMax = 3.402823466e+38f;
constexpr float kShadowRayEps = 0.001f;

// NaN/Inf guard + accumulation of one finished path (src/kernels/integrator/path.cu:99-104)
__device__ inline void film_accumulate(const RenderParams &p, uint32_t pixel_index, V3 color) {
    if (isnan(color.x) || isnan(color.y) || isnan(color.z) || isinf(color.x) || isinf(color.y) || isinf(color.z)) color = v3(0.0f);
    float4 *out = reinterpret_cast<float4 *>(p.output) + pixel_index;
    float4 prev = *out;
    float4 res;
    if (p.sum_mode) {
        res = make_float4(prev.x + color.x, prev.y + color.y, prev.z + color.z, prev.w + 1.0f);
    } else {
        float a = 1.0f / (float)p.spp;      // glm::mix(prev, new, a) = prev * (1 - a) + new * a
        res = make_float4(prev.x * (1.0f - a) + color.x * a, prev.y * (1.0f - a) + color.y * a, prev.z * (1.0f - a) + color.z * a,
                          prev.w * (1.0f - a) + 1.0f * a);
    }
    *out = res;
}

// The same for a launch that renders several samples per pixel: the path of the launch's k-th sample may write
// only when the pixel has received k of them, so the pixel's samples are accumulated in sample order whatever
// order their paths end in, and the film holds bit for bit what one launch per sample leaves.  Returns false when
// it is not this path's turn yet (the caller tries again later; it must not spin, the earlier path may be a lane
// of the same wave).
//
// A pixel's paths may run on different XCDs, whose L2s do not see each other's plain stores, so the pixel and its
// turn word are accessed at device scope.  Default build: the pixel moves as ONE 16-byte device-scope load and ONE
// 16-byte write-through store (a quarter of the memory transactions of four dword accesses; 1 % faster on the Cornell
// box), the turn word as
// relaxed device-scope atomics, and the order between them is kept by hand: the pixel load is issued after the turn
// word has arrived (the branch needs it), and the turn store after the pixel store has been acknowledged (the wait
// behind it).  An acquire or release at this scope would invalidate or write back whole caches, and the traversal
// lives on its L1.  That hand ordering is outside the language's memory model; a build with
// -DCPT_FILM_ORDER_BY_FENCES uses an acquire load and a release store of the turn word instead (slower, same film),
// so a toolchain or target change that breaks the hand-ordered form can be told apart by comparing the two.
typedef float FilmF4 __attribute__((ext_vector_type(4)));

//
// Running mean into a film whose alpha is 0 or 1 (p.turn_in_alpha): no turn words at all.  The reference's alpha is
// prev * (1 - 1/n) + 1/n, which is exactly 1 for every n up to the 65536 samples the executable renders once the
// first sample has written it, so during a launch the alpha of a pixel is free to carry the count: sample k > 0 of
// the launch writes when it reads alpha == k + 1 and leaves k + 2, the first sample writes unconditionally (nobody
// precedes it) and the last one leaves the true alpha.  Pixel and count are then ONE 16-byte word, read by one
// device-scope load and written by one write-through store: nothing to order, half the transactions again.
__device__ inline bool film_accumulate_in_order(const RenderParams &p, uint32_t pixel_index, uint32_t pass, V3 color) {
    FilmF4 *pixel = reinterpret_cast<FilmF4 *>(p.output) + pixel_index;
    if (p.turn_in_alpha) {
        FilmF4 prev;
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(prev)
                     : "v"(pixel)
                     : "memory");
        if (p.sum_mode) {
            // sums: the alpha of the film is the count of samples it holds (device_types.h: alpha_base)
            if (prev.w != (float)(p.alpha_base + pass)) return false;
            if (isnan(color.x) || isnan(color.y) || isnan(color.z) || isinf(color.x) || isinf(color.y) || isinf(color.z)) color = v3(0.0f);
            const FilmF4 sum = FilmF4{prev.x + color.x, prev.y + color.y, prev.z + color.z, prev.w + 1.0f};
            asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(pixel), "v"(sum) : "memory");
            return true;
        }
        if (pass != 0u && prev.w != (float)(pass + 1u)) return false;
        if (isnan(color.x) || isnan(color.y) || isnan(color.z) || isinf(color.x) || isinf(color.y) || isinf(color.z)) color = v3(0.0f);
        const float alpha_in = pass == 0u ? prev.w : 1.0f;
        const float a = 1.0f / (float)(p.spp + pass);
        FilmF4 res = FilmF4{prev.x * (1.0f - a) + color.x * a, prev.y * (1.0f - a) + color.y * a, prev.z * (1.0f - a) + color.z * a, alpha_in * (1.0f - a) + 1.0f * a};
        if (pass + 1u != p.spp_count) res.w = (float)(pass + 2u);
        asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(pixel), "v"(res) : "memory");
        return true;
    }
    uint32_t *turn = p.order + pixel_index;
#ifdef CPT_FILM_ORDER_BY_FENCES
    if (__hip_atomic_load(turn, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != pass) return false;
    FilmF4 prev = *pixel;
#else
    if (__hip_atomic_load(turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != pass) return false;
    FilmF4 prev;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(prev)
                 : "v"(pixel)
                 : "memory");
#endif
    if (isnan(color.x) || isnan(color.y) || isnan(color.z) || isinf(color.x) || isinf(color.y) || isinf(color.z)) color = v3(0.0f);
    FilmF4 res;
    if (p.sum_mode) {
        res = FilmF4{prev.x + color.x, prev.y + color.y, prev.z + color.z, prev.w + 1.0f};
    } else {
        float a = 1.0f / (float)(p.spp + pass);
        res = FilmF4{prev.x * (1.0f - a) + color.x * a, prev.y * (1.0f - a) + color.y * a, prev.z * (1.0f - a) + color.z * a, prev.w * (1.0f - a) + 1.0f * a};
    }
#ifdef CPT_FILM_ORDER_BY_FENCES
    *pixel = res;
    __hip_atomic_store(turn, pass + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#else
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 :
                 : "v"(pixel), "v"(res)
                 : "memory");
    __hip_atomic_store(turn, pass + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    return true;
}

// sum over the 64 lanes of a wave, result in every lane
__device__ inline unsigned long long wave_sum(uint32_t v) {
    unsigned long long s = v;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        uint32_t lo = __shfl_xor((uint32_t)s, off, 64), hi = __shfl_xor((uint32_t)(s >> 32), off, 64);
        s += ((unsigned long long)hi << 32) | lo;
    }
    return s;
}

// end of a kernel: per-wave sums -> per-workgroup sums in LDS -> five atomics into the workgroup's
// counter slot.  Every thread of the workgroup must call it.
__device__ inline void flush_counters(const SceneDev &sc, const LaneCounters &cnt, uint32_t paths, unsigned long long *block_cnt) {
    if (!sc.counters) return;
    if (threadIdx.x < 5) block_cnt[threadIdx.x] = 0;
    __syncthreads();
    const unsigned long long w0 = wave_sum(cnt.rays), w1 = wave_sum(cnt.nodes), w2 = wave_sum(cnt.tris), w3 = wave_sum(paths),
                             w4 = wave_sum(cnt.shades);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&block_cnt[0], w0);
        atomicAdd(&block_cnt[1], w1);
        atomicAdd(&block_cnt[2], w2);
        atomicAdd(&block_cnt[3], w3);
        atomicAdd(&block_cnt[4], w4);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        Counters *slot = sc.counters + (blockIdx.x & (kCounterSlots - 1));
        if (block_cnt[0]) atomicAdd(&slot->rays, block_cnt[0]);
        if (block_cnt[1]) atomicAdd(&slot->node_visits, block_cnt[1]);
        if (block_cnt[2]) atomicAdd(&slot->tri_tests, block_cnt[2]);
        if (block_cnt[3]) atomicAdd(&slot->paths, block_cnt[3]);
        if (block_cnt[4]) atomicAdd(&slot->shades, block_cnt[4]);
    }
}

}  // namespace cpt
