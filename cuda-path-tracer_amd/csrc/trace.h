// Two-level BVH traversal and ray/triangle intersection for gfx950 (device code).
//
// Bit-exact against the reference traversal (src/kernels/accel/accel.cuh:33-173 and
// src/kernels/geometry/trimesh.cuh:14-43):
//  * same slab arithmetic: six correctly rounded IEEE divisions per box, compare-and-swap
//    ordering, fmax/fmin NaN rules, hit <=> t0 <= t1 && t0 < tmax && t1 > tmin;
//  * same visit order: the reference pushes the left child, then the right one, and pops the
//    right one first, testing a node's box when it is POPPED (accel.cuh:73-86).  Here both child
//    boxes arrive with their parent's 64-byte record, so the tmax-independent part of the test
//    (t0 <= t1 && t1 > tmin) is evaluated at the parent and the entry point t0 is kept with a
//    deferred left child; `t0 < tmax` is re-checked when that child is popped, which is the
//    decision the reference takes at that moment because t0 and t1 depend only on box and ray;
//  * same Moeller-Trumbore expression order, two-sided, strict tmin < t < tmax acceptance.
// Rays are not renormalised when they enter an instance, so t is shared between spaces
// (accel.cuh:54-59).
#pragma once
#include "device_types.h"

namespace cpt {

constexpr int kLdsStackEntries = 16;   // per lane, in LDS: 16 * 8 B * 256 lanes = 32 KB per workgroup
constexpr int kSpillEntries = 48;      // per lane, private memory, touched only by very deep trees
constexpr int kTraceBlock = 256;

#ifdef CPT_STEP_STATS
// Development build only (make FLAGS+=-DCPT_STEP_STATS): per wave, shader-clock time, trips and participating lanes of
// every kind of step of the two persistent megakernels that use Flow::step (0 node, 1 triangle, 2 instance, 3 stack
// pops, 4 shading, 5 whole kernel; lanes[5]: lanes that had a ray in a traversal trip).
struct StepStats { unsigned long long time[6] = {0, 0, 0, 0, 0, 0}, trips[6] = {0, 0, 0, 0, 0, 0}, lanes[6] = {0, 0, 0, 0, 0, 0}; };
struct LaneCounters { uint32_t rays = 0, nodes = 0, tris = 0, shades = 0; StepStats *stats = nullptr; };
#else
struct LaneCounters { uint32_t rays = 0, nodes = 0, tris = 0, shades = 0; };
#endif

// Per-lane stack of (node reference, entry distance) pairs.  Entry e of lane l lives at
// lds[e * 256 + l]: lanes of a wave always touch consecutive 8-byte slots, whatever their depths.
// Pointers with their address space in the type: an access through a generic pointer is a FLAT
// instruction, which has to find out at run time whether the address is LDS, private or global
// memory and counts against both the memory and the LDS wait counters.  BVH records are always
// global memory and the stack window is always LDS, so the hot loop says so.
typedef float RawF4 __attribute__((ext_vector_type(4)));        // plain vectors: the HIP float4 / uint2 classes only bind generic references
typedef uint32_t RawU2 __attribute__((ext_vector_type(2)));
typedef const RawF4 __attribute__((address_space(1))) *GlobalF4;
typedef RawU2 __attribute__((address_space(3))) *LdsStack;
typedef RawU2 __attribute__((address_space(5))) *PrivateStack;
typedef const RawF4 __attribute__((address_space(3))) *LdsF4;
__device__ inline float4 load_f4(GlobalF4 p, int i) { RawF4 v = p[i]; return make_float4(v.x, v.y, v.z, v.w); }
__device__ inline float4 load_f4(LdsF4 p, int i) { RawF4 v = p[i]; return make_float4(v.x, v.y, v.z, v.w); }

// LDS_ENTRIES is a kernel's trade between LDS per workgroup (occupancy) and how deep a tree goes
// before entries land in private memory.  The two arrays live in the kernel (a __shared__ array of
// LDS_ENTRIES * kTraceBlock entries and a local array of kSpillEntries); this holds their addresses.
// LEAF_WEIGHT rides along because it is, like the stack's shape, a choice each kernel build makes for the
// scenes it is meant for (see Traversal::vote_step).
// FAST selects the ordered traversal (see Traversal) and rides along for the same reason; so does WIDE, which selects
// the two-level node records (exact traversal only), and SCENE_IN_LDS: the kernel holds the scene's node and triangle records
// in LDS (SceneDev::lds_scene, copied in by every workgroup) and `scene` points at them (Flow::step reads them there).
template <int LDS_ENTRIES, int LEAF_WEIGHT = 1, bool FAST = false, bool WIDE = false, bool SCENE_IN_LDS = false>
struct TraceStackT {
    static constexpr int kLdsEntries = LDS_ENTRIES;
    static constexpr int kLeafWeight = LEAF_WEIGHT;
    static constexpr bool kFast = FAST;
    static constexpr bool kWide = WIDE;
    static constexpr bool kSceneInLds = SCENE_IN_LDS;
    static_assert(!SCENE_IN_LDS || (WIDE && !FAST), "the resident copy holds two-level records");
    LdsF4 scene = nullptr;
    LdsStack lds;
    PrivateStack spill;
    Counters *counters;       // where an entry that does not fit is reported (may be null)
    __device__ inline TraceStackT(uint2 *shared_entries, uint2 *private_entries, Counters *report)
        : lds((LdsStack)shared_entries), spill((PrivateStack)private_entries), counters(report) {}
    // An entry beyond LDS_ENTRIES + kSpillEntries pending subtrees is dropped: that subtree is never walked (at()
    // returns an entry that is never entered) and the drop is counted, so a caller can tell a complete traversal
    // (count 0) from one that may have missed a hit.  The reference keeps 32 entries and writes past its array
    // beyond them (accel.cuh:69).
    __device__ inline void push(uint32_t sp, uint32_t node, float t0) {
        RawU2 e = {node, __float_as_uint(t0)};
        if (sp < LDS_ENTRIES) lds[sp * kTraceBlock + threadIdx.x] = e;
        else if (sp < LDS_ENTRIES + kSpillEntries) spill[sp - LDS_ENTRIES] = e;
        else if (counters) atomicAdd(&counters->stack_overflows, 1ull);
    }
    __device__ inline uint2 at(uint32_t sp) const {
        RawU2 e = {0u, 0x7fc00000u};          // NaN entry distance: never entered
        if (sp < LDS_ENTRIES) e = lds[sp * kTraceBlock + threadIdx.x];
        else if (sp < LDS_ENTRIES + kSpillEntries) e = spill[sp - LDS_ENTRIES];
        return make_uint2(e.x, e.y);
    }
};
using TraceStack = TraceStackT<kLdsStackEntries>;

struct Hit { uint32_t inst, prim; float u, v; };

// Lane count of a ballot as a 32-bit scalar: the 64-bit population count is carried as a 64-bit value, and
// ordering two of those is a vector compare on this target (there is no scalar 64-bit less-than).
__device__ inline int lanes_in(unsigned long long mask) { return __builtin_popcount((uint32_t)mask) + __builtin_popcount((uint32_t)(mask >> 32)); }

// ---- correctly rounded a / d from a correctly rounded reciprocal ---------------------------------
// With r = RN(1/d):  q0 = RN(a*r) is within 2 ulp of a/d; one residual step
// q1 = RN(q0 + RN(a - q0*d) * r) makes it faithful (error < 1 ulp); for a faithful q1 the residual
// a - q1*d is exact in one FMA and q2 = RN(q1 + (a - q1*d) * r) IS RN(a/d) (Markstein 1990).
// The argument needs every intermediate to stay a normal number, which `RayFrame::fast`
// guarantees: |d| in [2^-50, 2^50], and a = plane - origin with plane and origin each zero or of
// magnitude in [2^-50, 2^50], so a is zero or |a| >= 2^-74 and the smallest residual, about
// 2^-46 |a|, is still normal.  Five VALU operations instead of the ~10 of the IEEE divide
// expansion; rays or scenes outside the range take the plain divisions.
__device__ inline float div_by_recip(float a, float d, float r) {
    float q = a * r;
    float e = fmaf(-q, d, a);
    q = fmaf(e, r, q);
    e = fmaf(-q, d, a);
    return fmaf(e, r, q);
}

__device__ inline bool moderate(float x) { float a = fabsf(x); return a >= 0x1p-50f && a <= 0x1p50f; }
__device__ inline bool zero_or_moderate(float x) { return x == 0.0f || moderate(x); }
__device__ inline bool coords_in_fast_range(float a, float b, float c) { return zero_or_moderate(a) && zero_or_moderate(b) && zero_or_moderate(c); }

// A ray in one space (world or the space of the instance being traversed)
struct RayFrame {
    V3 o, d, r;      // origin, direction, RN(1/direction) (only meaningful when fast)
    bool fast;
    __device__ inline void set(V3 origin, V3 dir, bool boxes_ok) {
        o = origin;
        d = dir;
        fast = boxes_ok && moderate(dir.x) && moderate(dir.y) && moderate(dir.z) && coords_in_fast_range(origin.x, origin.y, origin.z);
        r = fast ? v3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z) : v3(0.0f);
    }
    // the ordered traversal multiplies by the reciprocal whatever the magnitudes (a zero component gives an infinity,
    // which orders the slab of that axis like the quotient by zero does)
    __device__ inline void set_recip(V3 origin, V3 dir) {
        o = origin;
        d = dir;
        fast = true;
        r = v3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
    }
    template <bool FAST> __device__ inline void set_for(V3 origin, V3 dir, bool boxes_ok) {
        if constexpr (FAST) set_recip(origin, dir); else set(origin, dir, boxes_ok);
    }
};

// One component of transform_inv * (p, 1) for a transform_inv that is the identity up to the signs of its zeros: c is
// the component, a and b the other two in the order they appear in the sum, signs the three zero signs (bits 31, 30, 29).
__device__ inline float identity_component(float c, float a, float b, uint32_t signs) {
    const uint32_t top = 0x80000000u;
    const uint32_t negative = __float_as_uint(c) & (signs ^ __float_as_uint(a)) & ((signs << 1) ^ __float_as_uint(b)) & (signs << 2) & top;
    return c == 0.0f ? __uint_as_float(negative) : c;
}

// The ray (ow, dw) of the world frame (reciprocal direction wr, fast division applicable: wfast) in the space of instance
// `in` (AccelTop::Intersect, accel.cuh:131-133: origin and direction through the inverse transform).  Where the
// inverse transform is the identity up to the signs of its zeros, the matrix products return every component that
// is not zero unchanged and a zero as a zero whose sign follows from the signs involved.  A direction the fast division
// applies to has no zero, so it and its reciprocals are the world frame's; the origin's zeros get their signs from
// identity_component.  The ordered traversal makes no promise about signs of zeros and keeps the world frame as it is.
template <bool FAST>
__device__ inline void enter_instance(RayFrame &f, const InstanceDev &in, V3 ow, V3 dw, V3 wr, bool wfast, bool boxes_ok) {
    if ((in.flags & kInstanceIdentity) != 0u && (FAST || wfast)) {
        f.o = ow;
        f.d = dw;
        f.r = wr;
        f.fast = FAST ? true : boxes_ok;
        if constexpr (!FAST) {
            // first component: (1*x + z*y) + (z*z' + z*1); second: (z*x + 1*y) + ...; third: (z*x + z*y) + (1*z + z*1)
            f.o = v3(identity_component(ow.x, ow.y, ow.z, in.zero_sign[0]), identity_component(ow.y, ow.x, ow.z, in.zero_sign[1]),
                     identity_component(ow.z, ow.x, ow.y, in.zero_sign[2]));
        }
    } else {
        const M4 inv = in.transform_inv;
        f.set_for<FAST>(xyz(mul(inv, v4(ow, 1.0f))), xyz(mul(inv, v4(dw, 0.0f))), boxes_ok);
    }
}

// accel.cuh:33-52 without the tmax clause; returns the entry distance in t0
__device__ inline bool slab(float bminx, float bminy, float bminz, float bmaxx, float bmaxy, float bmaxz, const RayFrame &f, float tmin, float &t0) {
    float t1;
    if (f.fast) {
        // no quotient is a NaN here (finite numerators, directions without zeros), so ordering a pair is a min and a
        // max: the values the compare-and-swap below produces, up to the sign of a zero, which no comparison sees
        const float x0 = div_by_recip(bminx - f.o.x, f.d.x, f.r.x), x1 = div_by_recip(bmaxx - f.o.x, f.d.x, f.r.x);
        const float y0 = div_by_recip(bminy - f.o.y, f.d.y, f.r.y), y1 = div_by_recip(bmaxy - f.o.y, f.d.y, f.r.y);
        const float z0 = div_by_recip(bminz - f.o.z, f.d.z, f.r.z), z1 = div_by_recip(bmaxz - f.o.z, f.d.z, f.r.z);
        t0 = fmaxf(fminf(x0, x1), fmaxf(fminf(y0, y1), fminf(z0, z1)));
        t1 = fminf(fmaxf(x0, x1), fminf(fmaxf(y0, y1), fmaxf(z0, z1)));
    } else {
        const float x0 = (bminx - f.o.x) / f.d.x, x1 = (bmaxx - f.o.x) / f.d.x;
        const float y0 = (bminy - f.o.y) / f.d.y, y1 = (bmaxy - f.o.y) / f.d.y;
        const float z0 = (bminz - f.o.z) / f.d.z, z1 = (bmaxz - f.o.z) / f.d.z;
        const float xa = x0 > x1 ? x1 : x0, xb = x0 > x1 ? x0 : x1;
        const float ya = y0 > y1 ? y1 : y0, yb = y0 > y1 ? y0 : y1;
        const float za = z0 > z1 ? z1 : z0, zb = z0 > z1 ? z0 : z1;
        t0 = fmaxf(xa, fmaxf(ya, za));
        t1 = fminf(xb, fminf(yb, zb));
    }
    return t0 <= t1 && t1 > tmin;
}

// The slab test of the ordered ("fast") traversal: distances by multiplication with the reciprocal direction (two
// operations per plane instead of a correctly rounded quotient), the exit distance widened by four units in the
// last place so that the rounding of the products never rejects a box the exact quotients admit.  It departs from
// accel.cuh:33-52 only in that rounding: a box may be entered that the reference skips, never the other way round
// (up to entry distances that tie with the current hit distance, which the callers admit on equality).
__device__ inline bool slab_fast(float bminx, float bminy, float bminz, float bmaxx, float bmaxy, float bmaxz, const RayFrame &f, float tmin, float &t0) {
    const float x0 = (bminx - f.o.x) * f.r.x, x1 = (bmaxx - f.o.x) * f.r.x;
    const float y0 = (bminy - f.o.y) * f.r.y, y1 = (bmaxy - f.o.y) * f.r.y;
    const float z0 = (bminz - f.o.z) * f.r.z, z1 = (bmaxz - f.o.z) * f.r.z;
    t0 = fmaxf(fminf(x0, x1), fmaxf(fminf(y0, y1), fminf(z0, z1)));
    const float t1 = fminf(fmaxf(x0, x1), fminf(fmaxf(y0, y1), fmaxf(z0, z1))) * 1.0000005f;
    return t0 <= t1 && t1 > tmin;
}
template <bool FAST>
__device__ inline bool slab_of(float bminx, float bminy, float bminz, float bmaxx, float bmaxy, float bmaxz, const RayFrame &f, float tmin, float &t0) {
    if constexpr (FAST) return slab_fast(bminx, bminy, bminz, bmaxx, bmaxy, bmaxz, f, tmin, t0);
    else return slab(bminx, bminy, bminz, bmaxx, bmaxy, bmaxz, f, tmin, t0);
}
// Is a box whose entry distance is t0 still of interest with the current reach?  The reference's test is strict
// (accel.cuh:51).  The ordered traversal of closest hits admits equality: a triangle at exactly the current hit
// distance may still win the tie (see Traversal), and it may sit in a box that begins at that distance.
// Its entry distances carry the rounding of slab_fast (up to 1.5 ulp too large), so its reach is widened like the exit
// distance: a box is never skipped whose exact entry distance lies within the reach.
template <bool FAST> __device__ inline bool within_reach(float t0, float tmax, bool any) {
    if constexpr (FAST) return any ? t0 < tmax * 1.0000005f : t0 <= tmax * 1.0000005f;
    else return t0 < tmax;
}

// trimesh.cuh:14-43 on a pre-gathered record: its three 16-byte words a, b, c
__device__ inline bool tri_test_words(float4 a, float4 b, float4 c, V3 o, V3 d, float tmin, float tmax, float &t, float &bv, float &bw, uint32_t &prim) {
    V3 p0 = v3(a.x, a.y, a.z), e1 = v3(b.x, b.y, b.z), e2 = v3(c.x, c.y, c.z);
    V3 qv = cross(d, e2);
    float det = dot(e1, qv);
    if (det != 0.0f) {
        det = 1.0f / det;
        V3 s = o - p0;
        float v = dot(s, qv) * det;
        if (v >= 0.0f) {
            V3 r = cross(s, e1);
            float w = dot(d, r) * det;
            float u = 1.0f - v - w;
            if (w >= 0.0f && u >= 0.0f) {
                float tt = dot(e2, r) * det;
                if (tt > tmin && tt < tmax) {
                    t = tt; bv = v; bw = w; prim = __float_as_uint(a.w);
                    return true;
                }
            }
        }
    }
    return false;
}
__device__ inline bool tri_test(const TriRecord *rec, V3 o, V3 d, float tmin, float tmax, float &t, float &bv, float &bw, uint32_t &prim) {
    GlobalF4 q = (GlobalF4)rec;
    const float4 a = load_f4(q, 0), b = load_f4(q, 1), c = load_f4(q, 2);
    return tri_test_words(a, b, c, o, d, tmin, tmax, t, bv, bw, prim);
}

// The same test for the ordered traversal: a hit at exactly the current reach is reported too (`tie`), the caller
// decides whether it replaces the one it holds.
__device__ inline bool tri_test_or_tie(const TriRecord *rec, V3 o, V3 d, float tmin, float tmax, float &t, float &bv, float &bw, uint32_t &prim, bool &tie) {
    GlobalF4 q = (GlobalF4)rec;
    float4 a = load_f4(q, 0), b = load_f4(q, 1), c = load_f4(q, 2);
    V3 p0 = v3(a.x, a.y, a.z), e1 = v3(b.x, b.y, b.z), e2 = v3(c.x, c.y, c.z);
    V3 qv = cross(d, e2);
    float det = dot(e1, qv);
    tie = false;
    if (det != 0.0f) {
        det = 1.0f / det;
        V3 s = o - p0;
        float v = dot(s, qv) * det;
        if (v >= 0.0f) {
            V3 r = cross(s, e1);
            float w = dot(d, r) * det;
            float u = 1.0f - v - w;
            if (w >= 0.0f && u >= 0.0f) {
                float tt = dot(e2, r) * det;
                if (tt > tmin && tt <= tmax) {
                    t = tt; bv = v; bw = w; prim = __float_as_uint(a.w);
                    tie = tt == tmax;
                    return true;
                }
            }
        }
    }
    return false;
}

constexpr uint32_t kNoNode = 0xffffffffu;
// triangle tests a lane may make in one trip on two-level records (see Traversal::vote_step; 1, 2 and 4 measured: Cornell box
// 1546 / 1576 / 1565 Mpaths/s, heightfield 10M 535 / 550 / 529; the ordered batch queries on binary records lose 15 % with 2)
#ifndef CPT_LEAF_RUN
#define CPT_LEAF_RUN 2
#endif
constexpr int kLeafRun = CPT_LEAF_RUN;

// Ordered traversal over a two-level record: the (up to four) slots that are hit, nearest first.  On return r[0] is
// the slot to walk next (kNoNode: none was hit) and r[1..3] the ones to push, r[3] first; slots that were missed or
// are empty come last with reference kNoNode and distance +inf.
__device__ inline void order_slots(float t[4], uint32_t r[4]) {
    auto nearer_first = [&](int a, int b) {
        const bool swap = t[b] < t[a];
        const float ta = swap ? t[b] : t[a], tb = swap ? t[a] : t[b];
        const uint32_t ra = swap ? r[b] : r[a], rb = swap ? r[a] : r[b];
        t[a] = ta; t[b] = tb; r[a] = ra; r[b] = rb;
    };
    nearer_first(0, 1); nearer_first(2, 3); nearer_first(0, 2); nearer_first(1, 3); nearer_first(1, 2);
}

// The wave runs the kind of step that has the most lanes, with lanes at a triangle or at an instance counted
// Stack::kLeafWeight times: those are the rarer kinds, and a lane that waits at one until its kind outnumbers
// the node walkers holds its ray back for many trips.  Measured with the weights 1, 2, 3, 4, 8: on scenes whose
// rays diverge 4 is best (soup 1M 102 -> 126 Mpaths/s, heightfield 10M 396 -> 428), on the Cornell box and on
// coherent camera rays 1 is (4 costs 2-4 % there), so the kernel builds for large scenes use 4, the others 1.
#ifndef CPT_LEAF_WEIGHT
#define CPT_LEAF_WEIGHT 4
#endif
constexpr int kLeafWeightLargeScene = CPT_LEAF_WEIGHT;

// Traversal state of one lane: where it stands (`cur`: internal node, triangle slot, instance
// leaf, or kNoNode when the ray is finished), its stack window and the ray in world and instance
// space.  Closest hit (ANY = false) is AccelTop::Intersect (accel.cuh:124-150), any hit
// (ANY = true) is AccelTop::Occlude (accel.cuh:152-173).
//
// vote_step() is one trip of the traversal loop for the whole wave: each live lane is at an
// internal node, at a triangle or at an instance leaf; the wave votes (three ballots + popcounts)
// for the state most lanes are in and executes only that step, the other lanes keep their state
// and wait.  A lane's own sequence of box tests, triangle tests and stack operations is exactly
// the sequential one, so results do not depend on the vote, but the expensive node step always
// runs with as many lanes as possible instead of being serialised with the other two every trip.
//
// Two traversal orders, chosen by the stack type a kernel is built with (Stack::kFast):
//  * exact (default): the reference's order and arithmetic, as described at the top of this file; results are the
//    reference's bit for bit.
//  * ordered ("fast"): of two children that are hit, the one the ray enters first is walked first and the other
//    waits with its entry distance, so a hit found early culls most of what is behind it (the reference walks the
//    right child first whatever the ray, accel.cuh:84-85, about twice the node visits on surfaces and several times
//    on scattered triangles); slab distances by reciprocal multiplication (slab_fast).  The triangle test is the
//    same, so a hit's distance and barycentrics are the same numbers.  What can differ is WHICH of several triangles
//    at the same or nearly the same distance is reported: the reference reports the first one its order meets whose
//    box it entered; the ordered traversal reports the nearest, and among hits at exactly equal distance the one
//    later in (top-level leaf position, sorted-leaf slot) order, which is the one the reference meets first and keeps.
struct Traversal {
    uint32_t cur = kNoNode;
    uint32_t sp = 0, sp_base = 0, inst_id = 0;
    uint32_t inst_pos = 0, hit_pos = 0, hit_slot = 0;   // ordered traversal: top-level leaf position of the instance being walked; of the hit held; its sorted-leaf slot
    bool in_blas = false, found = false;
    float tmin = 0.0f, tmax = 0.0f;
    uint32_t top_ref = kNoNode;        // the entry on top of the stack, when it is held here instead of in the stack's memory
    float top_t0 = 0.0f;
    RayFrame world, f;
    const PackedNode *nodes = nullptr;
    const TriRecord *tris = nullptr;
    Hit hit;

    // tests the top-level root box; afterwards live() tells whether there is anything to walk
    template <bool FAST = false, bool WIDE = false>
    __device__ inline void begin(const SceneDev &sc, V3 ow, V3 dw, float tmin_, float tmax_, LaneCounters &cnt) {
        cnt.rays++;
        cur = kNoNode;
        sp = 0; sp_base = 0;
        top_ref = kNoNode;
        in_blas = false; found = false;
        tmin = tmin_; tmax = tmax_;
        nodes = WIDE ? sc.top_wide : sc.top_nodes;
        const AccelRoot troot = *sc.top_root;
        world.set_for<FAST>(ow, dw, troot.flags == 0);
        f = world;
        float t0;
        if (slab_of<FAST>(troot.bmin[0], troot.bmin[1], troot.bmin[2], troot.bmax[0], troot.bmax[1], troot.bmax[2], world, tmin, t0) && t0 < tmax) cur = troot.ref;
    }
    __device__ inline bool live() const { return cur != kNoNode; }

    // returns false when no lane of the wave that called it has work left
    template <bool ANY, class Stack>
    __device__ inline bool vote_step(const SceneDev &sc, Stack &st, LaneCounters &cnt) {
        constexpr bool FAST = Stack::kFast, WIDE = Stack::kWide;
        // The entry on top of the stack stays in two registers: the last subtree pushed at a node is the first one popped, and
        // a pop that finds it there does not wait for the stack's memory; it moves there when the next one is pushed (camera
        // rays of the 10M heightfield 3.10 -> 3.14 Grays/s, wavefront integrator 424 -> 432 Mpaths/s; Flow::step does the same).
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
        const unsigned long long mn = __ballot(is_node), mt = __ballot(is_tri), mi = __ballot(is_inst);
        if ((mn | mt | mi) == 0ull) return false;
        const int cn = lanes_in(mn), ct = lanes_in(mt), ci = lanes_in(mi);
        bool need_pop = false;
        if (cn >= Stack::kLeafWeight * ct && cn >= Stack::kLeafWeight * ci) {
            if constexpr (WIDE) {
                // Up to four nodes from two or three levels below per record, slots in the order the reference reaches
                // them (right child first).  A slot is walked iff its own box is hit within the reach: the boxes of
                // the ancestors the record skipped contain it, and every operation of the slab test is monotonic in
                // the box coordinates, so an ancestor's test (which the reference makes first) cannot fail where the
                // slot's succeeds, nor can the ancestor's entry distance be larger.
                if (is_node) {
                    GlobalF4 q = (GlobalF4)(nodes + 2 * (size_t)cur);
                    const float4 a0 = load_f4(q, 0), a1 = load_f4(q, 1), a2 = load_f4(q, 2), a3 = load_f4(q, 3);
                    const float4 b0 = load_f4(q, 4), b1 = load_f4(q, 5), b2 = load_f4(q, 6);      // the first half's last quad has all four references
                    cnt.nodes++;
                    const uint32_t r0 = __float_as_uint(a3.x), r1 = __float_as_uint(a3.y), r2 = __float_as_uint(a3.z), r3 = __float_as_uint(a3.w);
                    if constexpr (FAST) {
                        const float inf = __uint_as_float(0x7f800000u);
                        float t[4];
                        uint32_t r[4] = {r0, r1, r2, r3};
                        const bool h0 = slab_fast(a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, f, tmin, t[0]) && within_reach<true>(t[0], tmax, ANY);
                        const bool h1 = slab_fast(a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, f, tmin, t[1]) && within_reach<true>(t[1], tmax, ANY);
                        const bool h2 = r2 != kNoNode && slab_fast(b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, f, tmin, t[2]) && within_reach<true>(t[2], tmax, ANY);
                        const bool h3 = r3 != kNoNode && slab_fast(b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, f, tmin, t[3]) && within_reach<true>(t[3], tmax, ANY);
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
                    const bool h0 = slab(a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, f, tmin, t0) && t0 < tmax;
                    const bool h1 = slab(a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, f, tmin, t1) && t1 < tmax;       // a record has at least two slots
                    const bool h2 = r2 != kNoNode && slab(b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, f, tmin, t2) && t2 < tmax;
                    const bool h3 = r3 != kNoNode && slab(b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, f, tmin, t3) && t3 < tmax;
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
                    const bool hl = slab_fast(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, f, tmin, t0l) && within_reach<true>(t0l, tmax, ANY);
                    const bool hr = slab_fast(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, f, tmin, t0r) && within_reach<true>(t0r, tmax, ANY);
                    const uint32_t lc = __float_as_uint(q3.x), rc = __float_as_uint(q3.y);
                    const bool left_first = t0l < t0r;              // equal entry distances: the right child, as the reference
                    if (hl && hr) push(left_first ? rc : lc, left_first ? t0r : t0l);
                    if (hl || hr) cur = (hl && (left_first || !hr)) ? lc : rc;
                    else need_pop = true;
                } else {
                    // accel.cuh:84-91 pushes a left child that is hit and tests its entry distance against the reach when it
                    // is popped; the reach only shrinks, so one that is out of reach now never needs the stack, and one
                    // that would be popped at once (the right child is not walked) is walked without going through it
                    const bool hl = slab(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, f, tmin, t0l) && t0l < tmax;
                    const bool hr = slab(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, f, tmin, t0r) && t0r < tmax;
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
                for (int run = 0; run < (WIDE ? kLeafRun : 1); run++) {
                    float t, bv, bw;
                    uint32_t prim;
                    cnt.tris++;
                    bool ended = false;
                    if constexpr (FAST && !ANY) {
                        bool tie;
                        const uint32_t slot = cur & ~kLeafBit;
                        if (tri_test_or_tie(tris + slot, f.o, f.d, tmin, tmax, t, bv, bw, prim, tie)) {
                            // at equal distance the hit later in (top-level leaf, sorted-leaf) order stays: the one the
                            // reference, which walks leaves in descending order, meets first and keeps
                            const bool later = inst_pos > hit_pos || (inst_pos == hit_pos && slot > hit_slot);
                            if (!tie || !found || later) {
                                found = true;
                                tmax = t;
                                hit.inst = inst_id; hit.prim = prim; hit.u = bv; hit.v = bw;
                                hit_pos = inst_pos; hit_slot = slot;
                            }
                        }
                    } else if (tri_test(tris + (cur & ~kLeafBit), f.o, f.d, tmin, tmax, t, bv, bw, prim)) {
                        found = true;
                        if (ANY) {
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
                        if (within_reach<FAST>(__uint_as_float(e.y), tmax, ANY)) { cur = e.x; break; }
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
                AccelRoot r = m.root;
                if constexpr (FAST) inst_pos = in.tlas_position;
                enter_instance<FAST>(f, in, world.o, world.d, world.r, world.fast, r.flags == 0);
                in_blas = true;
                sp_base = sp;
                float t0;
                if (slab_of<FAST>(r.bmin[0], r.bmin[1], r.bmin[2], r.bmax[0], r.bmax[1], r.bmax[2], f, tmin, t0) && within_reach<FAST>(t0, tmax, ANY)) cur = r.ref;
                else need_pop = true;
            }
        }
        if (need_pop) {
            cur = kNoNode;
            for (;;) {
                if (in_blas && sp == sp_base) { in_blas = false; f = world; nodes = WIDE ? sc.top_wide : sc.top_nodes; }
                if (sp == 0) break;
                uint2 e = pop();
                if (within_reach<FAST>(__uint_as_float(e.y), tmax, ANY)) { cur = e.x; break; }
            }
        }
        return true;
    }
};

// One ray per lane, all lanes of the wave that reach the call together (`active` = false for
// lanes with nothing to trace); returns when every lane's ray is finished.  tmax is updated to
// the hit distance on closest hits.
template <bool ANY, class Stack>
__device__ inline bool trace(const SceneDev &sc, bool active, V3 ow, V3 dw, float tmin, float &tmax, Hit &hit, Stack &st, LaneCounters &cnt) {
    Traversal t;
    if (active) t.begin<Stack::kFast, Stack::kWide>(sc, ow, dw, tmin, tmax, cnt);
    while (t.vote_step<ANY>(sc, st, cnt)) {}
    if (!ANY && t.found) { hit = t.hit; tmax = t.tmax; }
    return t.found;
}

}  // namespace cpt
