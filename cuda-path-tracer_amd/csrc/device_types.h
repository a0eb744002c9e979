// Data layout in HBM for the three device subsystems (LBVH build, traversal, bounce loop).
#pragma once
#include "vecmath.h"

namespace cpt {

// ---- build-side layout: identical to the reference's arrays so kernel::BuildAccel callers can
// be pointed at the same buffers (src/kernels/accel/accel.cuh:15-23).
struct Bbox { V4 pmin, pmax; };                 // 32 B; w = 1 on leaves
struct AccelNode { uint32_t lc_or_id, rc; };    // 8 B; leaf <=> rc == ~0u, then lc_or_id = original primitive index
static_assert(sizeof(Bbox) == 32 && sizeof(AccelNode) == 8, "reference layout");

// ---- traversal-side layout.
// One 64-byte record per internal node holding BOTH children's boxes and references, fetched
// with four 16-byte loads from one half cache line.  A child reference with bit 31 set is a
// leaf: its low bits are a triangle-record slot (bottom level) or an instance id (top level).
struct alignas(64) PackedNode {
    float lmin[3], lmax[3];
    float rmin[3], rmax[3];
    uint32_t lc, rc;
    uint32_t pad[2];
};
static_assert(sizeof(PackedNode) == 64, "one record per 64 B");
constexpr uint32_t kLeafBit = 0x80000000u;

// 48-byte triangle record in sorted-leaf order: p0, e1 = p1 - p0, e2 = p2 - p0 (single fp32
// subtractions, so the values equal what src/kernels/geometry/trimesh.cuh:21-22 computes per
// test) plus the original primitive index.
struct alignas(16) TriRecord {
    float p0[3]; uint32_t prim;
    float e1[3]; uint32_t pad0;
    float e2[3]; uint32_t pad1;
};
static_assert(sizeof(TriRecord) == 48, "three 16 B loads");

// Root of one packed BVH: the root's own box is tested once, as the reference does when it
// pops node 0 (accel.cuh:73-75); ref is a leaf reference when the BVH has a single primitive.
struct AccelRoot {
    float bmin[3], bmax[3];
    uint32_t ref;
    uint32_t num_prims;
    uint32_t flags;      // bit 0: some box coordinate is neither zero nor of magnitude in [2^-50, 2^50]
    uint32_t pad;
};

struct MeshDev {
    const PackedNode *nodes;    // binary records; null until something that walks them asks (scenes built with two-level records)
    const TriRecord *tris;
    const uint32_t *indices;
    const V3 *positions;
    const V3 *normals;
    const V2 *texcoords;
    AccelRoot root;
    uint32_t num_triangles;
    uint32_t pad;
    const PackedNode *wide;     // two-level records: two PackedNode-shaped halves per internal node (lbvh.hip: pack_from_memory), or null
    uint32_t lds_nodes, lds_tris;     // where this mesh's records lie in SceneDev::lds_scene (16-byte words), when that is set
};

struct InstanceDev {
    M4 transform;        // object -> world
    M4 transform_inv;    // world -> object
    M4 transform_it;     // transpose(transform_inv)
    uint32_t mesh;
    uint32_t material;
    int32_t light;       // index into the light table or -1
    uint32_t tlas_position;  // position of the instance among the sorted leaves of the top-level BVH (the reference visits leaves in descending position)
    uint32_t flags;          // kInstanceIdentity
    uint32_t zero_sign[3];   // kInstanceIdentity: per component of a transformed point, bits 31..29 = sign bits of the three zeros of transform_inv that meet its other terms
};
// transform_inv is the identity matrix up to the signs of its zeros (what the inverse of an untransformed object's matrix
// comes out as).  Its product with a finite point p, in the evaluation order of mul(), (1*x + z1*y) + (z2*z + z3*1) for
// the first component, returns a component that is not zero unchanged; a zero comes out negative only if it went in
// negative and the three other terms are negative zeros too, which depends on the signs of z1, z2, z3 (zero_sign)
// and of the other two components.  So the traversal can enter such an instance without the matrix products.
constexpr uint32_t kInstanceIdentity = 1u;

enum BsdfType : int { kLambert = 0, kPhong = 1, kBlinnPhong = 2, kMicrofacet = 3, kGlass = 4 };   // src/kernels/bsdf/bsdf.cuh:14-20
enum LobeType : int { kLobeNone = 0, kLobeDiffuse = 1, kLobeGlossy = 2, kLobeSpecular = 3 };      // src/kernels/bsdf/common.cuh:8-13

// BSDF parameter block per material, evaluated once on the host from the MTL values
// (src/kernels/material/mtl.cuh:16-66 has no uv dependence without textures).
// Glass keeps reflectance in `diffuse` and transmittance in `specular`.
// Texture numbers are 1-based indices into SceneDev::textures, 0 = constant.  A material with a
// diffuse or specular map keeps the raw constants here and is finished per hit (material_at()).
struct BsdfDev {
    int type;
    V3 emission;
    V3 diffuse;
    V3 specular;
    V3 transmittance;
    float shininess, s_norm, ior, roughness, opacity;
    uint32_t tex_diffuse_specular;     // diffuse map | specular map << 16
    uint32_t tex_emission;
};

// RGBA8 image, sRGB encoded, row 0 = first row of the file (CuTexture, src/cuda_helpers/texture.cpp:5-27)
struct TextureDev {
    const uint32_t *texels;
    uint32_t width, height;
};
static_assert(sizeof(BsdfDev) == 80, "five 16 B loads");

struct CameraDev { V3 pos, frame_x, frame_y, frame_z, x_dir, y_dir; };   // src/kernels/camera/pinhole.cuh:8-14

// One slot of the work counters.  The device holds kCounterSlots of them, one cache line each, and
// a workgroup adds to slot (blockIdx.x mod kCounterSlots): atomics on ONE address retire one at a
// time at the memory side (about 13 ns each on this part), atomics on different lines do not queue
// behind each other.  The host sums the slots.
struct alignas(64) Counters {
    unsigned long long rays;        // closest-hit + any-hit queries
    unsigned long long node_visits; // 64-byte records fetched
    unsigned long long tri_tests;   // triangle records fetched
    unsigned long long paths;       // camera paths started
    unsigned long long shades;      // shading surfaces evaluated (vertex fetch + material read)
    unsigned long long stack_overflows; // traversal stack entries dropped for lack of room (slot 0 only); 0 = every traversal was complete
};
constexpr uint32_t kCounterSlots = 1024;

struct SceneDev {
    const PackedNode *top_nodes;
    const AccelRoot *top_root;
    const InstanceDev *instances;
    const MeshDev *meshes;
    const BsdfDev *materials;
    const TextureDev *textures; // may be null when no material names a map
    const float *srgb_to_linear; // 256 entries, shared by every texture fetch
    const uint32_t *lights;     // instance index of each light
    uint32_t num_lights;
    float light_pick_pdf;
    CameraDev camera;
    Counters *counters;         // kCounterSlots slots; may be null
    int fast_traversal;         // 0: the reference's traversal order and arithmetic (bit-exact results); 1: ordered traversal (trace.h)
    int large_scene;            // the BVH is far larger than the caches next to the CUs: batch queries use the kernel builds for such scenes
    const PackedNode *top_wide; // the top level in two-level records, or null
    int wide_nodes;             // every BVH of the scene has its two-level records
    // A scene whose traversal records fit kLdsSceneBytes has them once more in one piece, for kernels that keep them in LDS:
    // the top level's two-level records, then per mesh its two-level records and its triangle records.  A node record is
    // kLdsNodeWords 16-byte words here (the seven a visit reads, section "traversal-side layout"), a triangle record three.
    const float *lds_scene;      // or null
    uint32_t lds_scene_words;    // 16-byte words in it
    uint32_t lds_top;      // first word of the top level's records (0)
};
constexpr uint32_t kLdsSceneBytes = 6144;
constexpr uint32_t kLdsNodeWords = 7;

struct RenderParams {
    float *output;              // W*H float4, row 0 = top of the image
    uint32_t width, height;
    uint32_t spp;               // 1-based sample index of this pass
    uint32_t max_depth;         // 0xffffffff = unbounded
    int channel;                // 0 colour, 1 normal
    int sum_mode;               // 0: running mean (reference semantics); 1: plain sum (sharded renders)
    // Several samples in one launch (the persistent kernels that take it): samples spp .. spp + spp_count - 1 of
    // every pixel.  `order` then holds one word per pixel, zero at launch: the number of this launch's samples the
    // pixel has received, which is what keeps a pixel's samples arriving in sample order.
    uint32_t spp_count = 1;
    uint32_t *order = nullptr;
    // Running-mean launches of several samples into a film whose alpha is 0 or 1 everywhere (a film only ever written by
    // running-mean samples and clears) need no `order` array: the count travels in the pixel's alpha, see film_accumulate_in_order.
    int turn_in_alpha = 0;
    // The same for sums into a film that only clears and sums have written: its alpha IS the number of samples summed so
    // far, alpha_base of them before this launch, so sample k of the launch writes when it reads alpha == alpha_base + k.
    uint32_t alpha_base = 0;
};

}  // namespace cpt
