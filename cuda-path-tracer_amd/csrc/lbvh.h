// Host entry points of the gfx950 LBVH builder (lbvh.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "device_types.h"

namespace cpt {

// Scratch memory of one build, reusable across builds (grown on demand, never shrunk).
struct BuildWorkspace {
    void *base = nullptr;
    size_t capacity = 0;
    // radix sort variant: 0 = tile histograms + scan + scatter per pass (default, faster on this part),
    // 1 = onesweep (histograms with the keys, one look-back kernel per pass)
    int sort_variant = 0;
    // nodes per workgroup of the first hierarchy pass that may be handed to the second pass (at most 64, the
    // default); the rest climbs inside the first pass.  Lower values exist to exercise that path in tests.
    uint32_t carrier_limit = 64;
    bool borrowed = false;          // `base` belongs to someone else: never freed, never grown
    static size_t bytes_for(uint32_t num_primitives);
    int reserve(uint32_t num_primitives);
    void release();
};

// Traversal records a build may write next to the reference-format arrays (all device pointers): the 64-byte binary
// records (n - 1) and the 128-byte two-level records (two PackedNode per internal node), either of which may be null
// (not wanted), and the root record.  top_level: leaves are instances (references carry the primitive id) instead of triangles
// (references carry the sorted-leaf position).
struct RecordOutputs {
    PackedNode *binary = nullptr;
    PackedNode *wide = nullptr;
    AccelRoot *root = nullptr;
    int top_level = 0;
};

// Drop-in for kernel::BuildAccel (src/kernels/accel/accel_build.cuh:7): `nodes` and `bboxes` are
// device arrays of 2n-1 entries; the leaf boxes occupy bboxes[n-1 .. 2n-2] on entry and are
// reordered in place; every node and every internal box is written.
//   ws          optional reusable scratch; when null, scratch is allocated and freed inside and
//               the call returns only after the build has finished (the reference's behaviour)
//   leaf_src    optional device array of the n unsorted leaf boxes living OUTSIDE `bboxes`;
//               when given, bboxes[n-1..] need not be filled and one device copy is saved
//   keys_out    optional device array of n sorted 64-bit keys (code << 32 | primitive)
//   parents_out optional device array of 2n-1 parent indices (entry 0 is never written)
//   stage_ms    optional host array of 5 floats: morton, sort, hierarchy pass one (leaf gather + everything inside a workgroup's leaves), hierarchy pass two (nodes spanning workgroups) with the records of those nodes, 0 (forces a sync)
//   records     optional traversal records, written by the hierarchy passes themselves (a node finished inside a
//               workgroup has its children and grandchildren in LDS at that moment) and a short kernel for the rest
// Returns 0 or a hipError_t value.
int lbvh_build(AccelNode *nodes, Bbox *bboxes, const Bbox &merged, uint32_t num_primitives, hipStream_t stream, BuildWorkspace *ws,
               const Bbox *leaf_src, uint64_t *keys_out, uint32_t *parents_out, float *stage_ms, const RecordOutputs *records = nullptr);

// The same build (in place, leaf boxes pre-filled, returns when it has finished) with all scratch in a device
// array of fixed size inside the library: no allocation of any kind.  At most kStaticMaxPrimitives primitives
// (hipErrorInvalidValue beyond).  Calls on one GPU share that array: two of them at the same time race.
constexpr uint32_t kStaticMaxPrimitives = 1u << 20;
uint32_t lbvh_static_max_primitives();
int lbvh_build_static(AccelNode *nodes, Bbox *bboxes, const Bbox &merged, uint32_t num_primitives, hipStream_t stream);

// Leaf boxes of a triangle mesh (w = 1), the device counterpart of src/scene/mesh.cpp:88-98.
int lbvh_triangle_boxes(Bbox *leaf, const uint32_t *indices, const V3 *positions, uint32_t num_triangles, hipStream_t stream);

// The binary records of a finished build from its reference-format arrays, for callers that did not ask the build for them.
int lbvh_pack_binary(PackedNode *out_binary, const AccelNode *nodes, const Bbox *bboxes, uint32_t num_primitives, bool top_level, hipStream_t stream);

// 48-byte triangle records in sorted-leaf order from a finished bottom-level build.
int lbvh_pack_triangles(TriRecord *out_tris, const AccelNode *nodes, const uint32_t *indices, const V3 *positions, uint32_t num_triangles, hipStream_t stream);

}  // namespace cpt
