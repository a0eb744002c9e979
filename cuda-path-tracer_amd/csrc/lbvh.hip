// Linear BVH builder for gfx950: 30-bit Morton keys, LDS-staged stable LSD radix sort,
// Karras internal-node split, bottom-up refit with per-node arrival counters, and packing into
// 64-byte traversal records.
//
// Results are bit-identical to the reference builder (src/kernels/accel/accel_build.cu):
//   keys      : accel_build.cu:25-41  (same IEEE expression, 0/0 on a flat axis -> 1023)
//   order     : accel_build.cu:193    (ascending 64-bit keys; keys are unique because the low
//               word is the primitive index, and the indices enter in ascending order, so a
//               STABLE sort on the 30 code bits alone yields the same permutation)
//   topology  : accel_build.cu:57-138 (Karras 2012 range/split searches on the 64-bit keys)
//   boxes     : accel_build.cu:140-180 (min/max are exact and order independent; the sign-split
//               integer atomics there order -0 below +0, reproduced here with a total order)
#include "lbvh.h"

#include <hip/hip_runtime.h>

namespace cpt {

namespace {

constexpr int kBlock = 256;                 // 4 waves of 64
constexpr int kItems = 16;                  // keys per thread in the sort kernels
constexpr int kTile = kBlock * kItems;      // 4096 keys = 32 KB of LDS staging per workgroup
constexpr int kWaves = kBlock / 64;
constexpr int kRadixBits = 8;
constexpr int kBins = 1 << kRadixBits;

#define HIP_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; } while (0)

__device__ inline uint32_t morton_spread(uint32_t x) {     // accel_build.cu:17-23
    x = (x ^ (x << 16)) & 0xff0000ffu;
    x = (x ^ (x << 8)) & 0x0300f00fu;
    x = (x ^ (x << 4)) & 0x030c30c3u;
    x = (x ^ (x << 2)) & 0x09249249u;
    return x;
}

// float -> uint32 truncation with saturation (what the GPU conversion instruction does)
__device__ inline uint32_t f2u_sat(float f) {
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)f;
}

__global__ void __launch_bounds__(kBlock) k_morton(uint64_t *__restrict__ keys, const Bbox *__restrict__ leaf, Bbox merged, uint32_t n) {
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4 *p = reinterpret_cast<const float4 *>(leaf + i);
    float4 lo = p[0], hi = p[1];
    float cx = (lo.x + hi.x) * 0.5f, cy = (lo.y + hi.y) * 0.5f, cz = (lo.z + hi.z) * 0.5f;
    float sx = merged.pmax.x - merged.pmin.x, sy = merged.pmax.y - merged.pmin.y, sz = merged.pmax.z - merged.pmin.z;
    float px = (cx - merged.pmin.x) / sx, py = (cy - merged.pmin.y) / sy, pz = (cz - merged.pmin.z) / sz;
    uint32_t x = morton_spread(f2u_sat(fminf(px * 1024.0f, 1023.0f)));
    uint32_t y = morton_spread(f2u_sat(fminf(py * 1024.0f, 1023.0f)));
    uint32_t z = morton_spread(f2u_sat(fminf(pz * 1024.0f, 1023.0f)));
    uint64_t code = (x << 2) | (y << 1) | z;
    keys[i] = (code << 32) | i;
}

// ------------------------------------------------------------------------------------------
// Radix sort pass = tile histogram -> per-bin scan over tiles -> stable LDS-staged scatter.
// ------------------------------------------------------------------------------------------
__device__ inline uint32_t wave_incl_scan(uint32_t v) {
    int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// exclusive scan of one value per thread over a 256-thread workgroup; tmp needs kWaves + 1 words
__device__ inline uint32_t block_excl_scan(uint32_t v, uint32_t *tmp, uint32_t &total) {
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = wave_incl_scan(v);
    __syncthreads();
    if (lane == 63) tmp[w] = inc;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int i = 0; i < kWaves; i++) {
        uint32_t t = tmp[i];
        if (i < w) base += t;
    }
    total = tmp[0] + tmp[1] + tmp[2] + tmp[3];
    return base + inc - v;
}

__global__ void __launch_bounds__(kBlock) k_hist(const uint64_t *__restrict__ keys, uint32_t n, int shift, uint32_t *__restrict__ tile_hist,
                                                  uint32_t num_tiles) {
    __shared__ uint32_t hist[kBins];
    hist[threadIdx.x] = 0;
    __syncthreads();
    uint32_t base = blockIdx.x * kTile;
#pragma unroll
    for (int k = 0; k < kItems; k++) {
        uint32_t i = base + k * kBlock + threadIdx.x;
        if (i < n) {
            uint32_t d = (uint32_t)(keys[i] >> shift) & (kBins - 1);
            atomicAdd(&hist[d], 1u);
        }
    }
    __syncthreads();
    tile_hist[(size_t)threadIdx.x * num_tiles + blockIdx.x] = hist[threadIdx.x];   // bin-major
}

// one workgroup per bin: exclusive scan of that bin's counts over all tiles, total to bin_total
__global__ void __launch_bounds__(kBlock) k_scan_bins(uint32_t *__restrict__ tile_hist, uint32_t num_tiles, uint32_t *__restrict__ bin_total) {
    __shared__ uint32_t tmp[kWaves + 1];
    uint32_t *row = tile_hist + (size_t)blockIdx.x * num_tiles;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < num_tiles; base += kBlock) {
        uint32_t i = base + threadIdx.x;
        uint32_t v = i < num_tiles ? row[i] : 0u;
        uint32_t total;
        uint32_t ex = block_excl_scan(v, tmp, total);
        if (i < num_tiles) row[i] = carry + ex;
        carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) bin_total[blockIdx.x] = carry;
}

__global__ void __launch_bounds__(kBlock) k_scatter(const uint64_t *__restrict__ keys_in, uint64_t *__restrict__ keys_out, uint32_t n, int shift,
                                                     const uint32_t *__restrict__ tile_hist, const uint32_t *__restrict__ bin_total,
                                                     uint32_t num_tiles) {
    __shared__ uint64_t staged[kTile];           // keys in tile-sorted order
    __shared__ uint32_t cnt[kWaves][kBins];      // per-wave digit counts, then exclusive over waves
    __shared__ uint32_t tile_off[kBins];         // first staged slot of each digit
    __shared__ uint32_t gbase[kBins];            // global destination of staged slot 0 of each digit
    __shared__ uint32_t tmp[kWaves + 1];

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const uint32_t tile_base = blockIdx.x * kTile;
    const uint32_t tile_n = min((uint32_t)kTile, n - tile_base);
#pragma unroll
    for (int i = 0; i < kWaves; i++) cnt[i][t] = 0;
    __syncthreads();

    // Each wave owns a contiguous quarter of the tile and walks it 64 keys at a time, so key
    // order within the tile is (wave, round, lane): ranking in that order keeps the sort stable.
    uint64_t key[kItems];
    uint32_t rank[kItems];
    const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < kItems; r++) {
        uint32_t li = w * (64 * kItems) + r * 64 + lane;
        bool valid = li < tile_n;
        uint64_t k = valid ? keys_in[tile_base + li] : ~0ull;
        key[r] = k;
        uint32_t d = (uint32_t)(k >> shift) & (kBins - 1);
        uint64_t same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < kRadixBits; b++) {
            bool bit = (d >> b) & 1u;
            uint64_t bal = __ballot(bit);
            same &= bit ? bal : ~bal;
        }
        uint32_t before = __popcll(same & lt_mask);
        uint32_t old = cnt[w][d];
        if (valid && before == 0) cnt[w][d] = old + __popcll(same);
        rank[r] = old + before;
    }
    __syncthreads();

    // digit t: exclusive scan over waves, tile total, then scan over digits
    uint32_t total_d = 0;
#pragma unroll
    for (int i = 0; i < kWaves; i++) {
        uint32_t c = cnt[i][t];
        cnt[i][t] = total_d;
        total_d += c;
    }
    uint32_t all;
    uint32_t off = block_excl_scan(total_d, tmp, all);
    tile_off[t] = off;
    __syncthreads();
    uint32_t bt = bin_total[t];
    uint32_t bin_base = block_excl_scan(bt, tmp, all);
    gbase[t] = bin_base + tile_hist[(size_t)t * num_tiles + blockIdx.x] - off;
    __syncthreads();

#pragma unroll
    for (int r = 0; r < kItems; r++) {
        uint32_t li = w * (64 * kItems) + r * 64 + lane;
        if (li < tile_n) {
            uint32_t d = (uint32_t)(key[r] >> shift) & (kBins - 1);
            staged[tile_off[d] + cnt[w][d] + rank[r]] = key[r];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kItems; k++) {
        uint32_t i = k * kBlock + t;
        if (i < tile_n) {
            uint64_t kk = staged[i];
            uint32_t d = (uint32_t)(kk >> shift) & (kBins - 1);
            keys_out[gbase[d] + i] = kk;
        }
    }
}

// sorted leaf boxes into place + leaf nodes (accel_build.cu:43-51)
__global__ void __launch_bounds__(kBlock) k_gather_leaves(const uint64_t *__restrict__ keys, const Bbox *__restrict__ leaf_src,
                                                          Bbox *__restrict__ bboxes, AccelNode *__restrict__ nodes, uint32_t n) {
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    uint32_t prim = (uint32_t)keys[i];
    const float4 *s = reinterpret_cast<const float4 *>(leaf_src + prim);
    float4 a = s[0], b = s[1];
    float4 *d = reinterpret_cast<float4 *>(bboxes + (n - 1) + i);
    d[0] = a;
    d[1] = b;
    nodes[(n - 1) + i] = AccelNode{prim, ~0u};
}

constexpr uint32_t kLocalBit = 0x80000000u;

__device__ inline uint32_t lcp(uint64_t a, uint64_t b) {           // accel_build.cu:53-55
    uint64_t x = a ^ b;
    return x ? (uint32_t)__builtin_clzll(x) : 64u;
}

// Karras 2012 as written at accel_build.cu:57-138
__global__ void __launch_bounds__(kBlock) k_karras(AccelNode *__restrict__ nodes, uint32_t *__restrict__ parents, const uint64_t *__restrict__ codes,
                                                    uint32_t n) {
    uint32_t index = blockIdx.x * kBlock + threadIdx.x;
    if (index >= n - 1) return;
    uint32_t first, last;
    if (index == 0) {
        first = 0;
        last = n - 1;
    } else {
        uint64_t code = codes[index];
        uint32_t l_lcp = lcp(code, codes[index - 1]);
        uint32_t r_lcp = lcp(code, codes[index + 1]);
        int d = l_lcp > r_lcp ? -1 : 1;
        uint32_t min_lcp = min(l_lcp, r_lcp);
        uint32_t step = 1;
        uint32_t j_lcp;
        do {
            step <<= 1;
            int j = (int)(index + (uint32_t)d * step);
            j_lcp = 0;
            if (j >= 0 && (uint32_t)j < n) j_lcp = lcp(code, codes[j]);
        } while (j_lcp > min_lcp);
        uint32_t l = step >> 1;
        uint32_t r = min(step, d == -1 ? index : n - 1 - index);
        while (l < r) {
            uint32_t mid = l + ((r - l) >> 1) + 1;
            uint32_t j = index + (uint32_t)d * mid;
            if (lcp(code, codes[j]) < min_lcp) r = mid - 1; else l = mid;
        }
        r = index + (uint32_t)d * l;
        l = index;
        if (l <= r) { first = l; last = r; } else { first = r; last = l; }
    }
    uint64_t l_code = codes[first];
    uint32_t node_lcp = lcp(l_code, codes[last]);
    uint32_t l = first, r = last;
    while (l < r) {
        uint32_t mid = l + ((r - l) >> 1);
        if (lcp(l_code, codes[mid]) > node_lcp) l = mid + 1; else r = mid;
    }
    uint32_t lc = l - 1, rc = l;
    if (lc == first) lc += n - 1;
    if (rc == last) rc += n - 1;
    nodes[index] = AccelNode{lc, rc};
    // bit 31 marks a parent whose whole leaf range lies inside one refit workgroup's 256 leaves
    uint32_t tag = (first / kBlock == last / kBlock) ? (index | kLocalBit) : index;
    parents[lc] = tag;
    parents[rc] = tag;
}

// Total order on floats in which -0 sorts below +0: the order the reference's integer-atomic
// min/max realise (accel_build.cu:150-158).  No NaNs reach the refit.
__device__ inline uint32_t order_key(float f) {
    uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ inline float tmin(float a, float b) { return order_key(b) < order_key(a) ? b : a; }
__device__ inline float tmax(float a, float b) { return order_key(b) > order_key(a) ? b : a; }

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Device-scope 16-byte accesses that bypass the per-CU cache (loads) or write through to memory
// (stores).  A box is two of them; one wide access costs the fabric a quarter of what four
// dword accesses of the same kind do.  Inline asm is outside the compiler's wait-count
// bookkeeping, so each statement carries its own wait.
__device__ inline void load_box_sc1(const Bbox *p, f32x4 &lo, f32x4 &hi) {
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\t"
                 "global_load_dwordx4 %1, %2, off offset:16 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(lo), "=&v"(hi)
                 : "v"(p)
                 : "memory");
}
__device__ inline void store_box_sc1(Bbox *p, f32x4 lo, f32x4 hi) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\t"
                 "global_store_dwordx4 %0, %2, off offset:16 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 :
                 : "v"(p), "v"(lo), "v"(hi)
                 : "memory");
}

// Bottom-up refit: one lane per leaf climbs towards the root; at every parent the first arrival
// stops and the second merges both children.
//
// Leaves are in sorted order and an internal node's index is one end of its leaf range, so a
// node whose range lies inside the 256 leaves of one workgroup is only ever touched by that
// workgroup and owns the LDS slot (index - first leaf of the workgroup).  Those nodes -- all but
// a few per workgroup -- hand their boxes over through LDS with workgroup-scope atomics.  Only
// nodes whose range crosses a workgroup boundary use memory: the child box is written with
// device-scope write-through stores, drained, and then announced by a device-scope arrival
// counter; the merging lane reads the sibling with device-scope loads that bypass the per-CU
// cache.  No cache-wide fence is needed and the result does not depend on where lanes run.
__global__ void __launch_bounds__(kBlock) k_refit(const uint32_t *__restrict__ parents, const AccelNode *__restrict__ nodes, Bbox *bboxes,
                                                   uint32_t *arrivals, Bbox merged, uint32_t n) {
    __shared__ float s_box[kBlock][6];          // boxes of finished internal nodes, slot = node index - block_first
    __shared__ float s_leaf[kBlock][6];         // this workgroup's leaf boxes
    __shared__ uint2 s_nodes[kBlock];           // children of internal nodes block_first .. block_first + 255
    __shared__ uint32_t s_parent[kBlock];       // their parent tags
    __shared__ uint32_t s_arrivals[kBlock];
    const uint32_t block_first = blockIdx.x * kBlock;
    const uint32_t i = block_first + threadIdx.x;
    s_arrivals[threadIdx.x] = 0;
    float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    if (i < n) {
        const float4 *lb = reinterpret_cast<const float4 *>(bboxes + (n - 1) + i);
        float4 lo4 = lb[0], hi4 = lb[1];
        lo[0] = lo4.x; lo[1] = lo4.y; lo[2] = lo4.z; hi[0] = hi4.x; hi[1] = hi4.y; hi[2] = hi4.z;
        float *sl = s_leaf[threadIdx.x];
        sl[0] = lo[0]; sl[1] = lo[1]; sl[2] = lo[2]; sl[3] = hi[0]; sl[4] = hi[1]; sl[5] = hi[2];
    }
    if (i < n - 1) {
        AccelNode nd = nodes[i];
        s_nodes[threadIdx.x] = make_uint2(nd.lc_or_id, nd.rc);
        s_parent[threadIdx.x] = i > 0 ? parents[i] : 0u;
    }
    __syncthreads();
    if (i >= n) return;
    uint32_t cur = (n - 1) + i;
    bool cur_internal = false;
    // internal boxes start from the inverted merged box (accel_build.cu:140-148)
    const float init_lo[3] = {merged.pmax.x, merged.pmax.y, merged.pmax.z};
    const float init_hi[3] = {merged.pmin.x, merged.pmin.y, merged.pmin.z};
    uint32_t tag = parents[cur];
    for (;;) {
        const uint32_t cslot = cur - block_first;
        const uint32_t pa = tag & ~kLocalBit;
        const bool local = (tag & kLocalBit) != 0;
        // The parent's children and its own parent tag are read-only here (written by earlier
        // kernels): fetch them now so that the loads overlap the hand-off below.
        const uint32_t pslot = pa - block_first;
        uint2 nd;
        uint32_t next_tag = 0;
        if (pslot < kBlock) {
            nd = s_nodes[pslot];
            next_tag = s_parent[pslot];
        } else {
            AccelNode g = nodes[pa];
            nd = make_uint2(g.lc_or_id, g.rc);
            if (pa != 0) next_tag = parents[pa];
        }
        uint32_t seen;
        if (local) {
            if (cur_internal) {
                float *sb = s_box[cslot];
                sb[0] = lo[0]; sb[1] = lo[1]; sb[2] = lo[2]; sb[3] = hi[0]; sb[4] = hi[1]; sb[5] = hi[2];
                float4 *gb = reinterpret_cast<float4 *>(bboxes + cur);
                gb[0] = make_float4(lo[0], lo[1], lo[2], merged.pmax.w);
                gb[1] = make_float4(hi[0], hi[1], hi[2], merged.pmin.w);
            }
            seen = __hip_atomic_fetch_add(&s_arrivals[pslot], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            if (cur_internal) {
                f32x4 olo = {lo[0], lo[1], lo[2], merged.pmax.w}, ohi = {hi[0], hi[1], hi[2], merged.pmin.w};
                store_box_sc1(bboxes + cur, olo, ohi);
            }
            seen = __hip_atomic_fetch_add(&arrivals[pa], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (seen == 0) return;
        const uint32_t sib = nd.x == cur ? nd.y : nd.x;
        float slo[3], shi[3];
        if (sib >= n - 1) {                       // a leaf: written by an earlier kernel
            const uint32_t lslot = sib - (n - 1) - block_first;
            if (lslot < kBlock) {
                const float *sl = s_leaf[lslot];
                slo[0] = sl[0]; slo[1] = sl[1]; slo[2] = sl[2]; shi[0] = sl[3]; shi[1] = sl[4]; shi[2] = sl[5];
            } else {
                const float4 *sp = reinterpret_cast<const float4 *>(bboxes + sib);
                float4 a4 = sp[0], b4 = sp[1];
                slo[0] = a4.x; slo[1] = a4.y; slo[2] = a4.z; shi[0] = b4.x; shi[1] = b4.y; shi[2] = b4.z;
            }
        } else if (local) {
            const float *sb = s_box[sib - block_first];
            slo[0] = sb[0]; slo[1] = sb[1]; slo[2] = sb[2]; shi[0] = sb[3]; shi[1] = sb[4]; shi[2] = sb[5];
        } else {
            f32x4 a4, b4;
            load_box_sc1(bboxes + sib, a4, b4);
            slo[0] = a4[0]; slo[1] = a4[1]; slo[2] = a4[2]; shi[0] = b4[0]; shi[1] = b4[1]; shi[2] = b4[2];
        }
#pragma unroll
        for (int a = 0; a < 3; a++) {
            lo[a] = tmin(init_lo[a], tmin(lo[a], slo[a]));
            hi[a] = tmax(init_hi[a], tmax(hi[a], shi[a]));
        }
        cur = pa;
        cur_internal = true;
        tag = next_tag;
        if (cur == 0) {
            float4 *gb = reinterpret_cast<float4 *>(bboxes);
            gb[0] = make_float4(lo[0], lo[1], lo[2], merged.pmax.w);
            gb[1] = make_float4(hi[0], hi[1], hi[2], merged.pmin.w);
            return;
        }
    }
}

// parent links without the locality tag, for callers that ask for the reference's parents array
__global__ void __launch_bounds__(kBlock) k_strip_parents(uint32_t *__restrict__ out, const uint32_t *__restrict__ parents, uint32_t count) {
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= 1 && i < count) out[i] = parents[i] & ~kLocalBit;
}

// ---- packing into traversal records --------------------------------------------------------
// The traversal may replace its divisions by a reciprocal-based sequence that is exact only while
// every box coordinate is zero or of magnitude in [2^-50, 2^50]; one flag per BVH records that.
__device__ inline bool coord_in_fast_range(float x) {
    float a = fabsf(x);
    return x == 0.0f || (a >= 0x1p-50f && a <= 0x1p50f);
}
__device__ inline bool box_in_fast_range(const Bbox &b) {
    return coord_in_fast_range(b.pmin.x) && coord_in_fast_range(b.pmin.y) && coord_in_fast_range(b.pmin.z) && coord_in_fast_range(b.pmax.x) &&
           coord_in_fast_range(b.pmax.y) && coord_in_fast_range(b.pmax.z);
}

__device__ inline uint32_t child_ref(uint32_t child, const AccelNode *nodes, uint32_t n, int top_level) {
    if (child < n - 1) return child;
    return kLeafBit | (top_level ? nodes[child].lc_or_id : child - (n - 1));
}

__global__ void __launch_bounds__(kBlock) k_pack_nodes(PackedNode *__restrict__ out, const AccelNode *__restrict__ nodes, const Bbox *__restrict__ bboxes,
                                                        uint32_t n, int top_level, AccelRoot *__restrict__ root) {
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    // root->flags was zeroed by the host before this launch; lanes only ever OR bits into it
    if (i == 0) {
        Bbox b = bboxes[0];
        root->bmin[0] = b.pmin.x; root->bmin[1] = b.pmin.y; root->bmin[2] = b.pmin.z;
        root->bmax[0] = b.pmax.x; root->bmax[1] = b.pmax.y; root->bmax[2] = b.pmax.z;
        root->ref = n == 1 ? (kLeafBit | (top_level ? nodes[0].lc_or_id : 0u)) : 0u;
        root->num_prims = n;
        if (!box_in_fast_range(b)) atomicOr(&root->flags, 1u);
    }
    if (n < 2 || i >= n - 1) return;
    AccelNode nd = nodes[i];
    Bbox l = bboxes[nd.lc_or_id], r = bboxes[nd.rc];
    if (!box_in_fast_range(l) || !box_in_fast_range(r)) atomicOr(&root->flags, 1u);
    float4 q0 = make_float4(l.pmin.x, l.pmin.y, l.pmin.z, l.pmax.x);
    float4 q1 = make_float4(l.pmax.y, l.pmax.z, r.pmin.x, r.pmin.y);
    float4 q2 = make_float4(r.pmin.z, r.pmax.x, r.pmax.y, r.pmax.z);
    uint32_t lref = child_ref(nd.lc_or_id, nodes, n, top_level), rref = child_ref(nd.rc, nodes, n, top_level);
    float4 q3 = make_float4(__uint_as_float(lref), __uint_as_float(rref), 0.0f, 0.0f);
    float4 *o = reinterpret_cast<float4 *>(out + i);
    o[0] = q0; o[1] = q1; o[2] = q2; o[3] = q3;
}

__global__ void __launch_bounds__(kBlock) k_pack_tris(TriRecord *__restrict__ out, const AccelNode *__restrict__ nodes, const uint32_t *__restrict__ indices,
                                                       const V3 *__restrict__ positions, uint32_t n) {
    uint32_t s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= n) return;
    uint32_t prim = nodes[(n - 1) + s].lc_or_id;
    uint32_t i0 = indices[3 * prim], i1 = indices[3 * prim + 1], i2 = indices[3 * prim + 2];
    V3 p0 = positions[i0], p1 = positions[i1], p2 = positions[i2];
    V3 e1 = p1 - p0, e2 = p2 - p0;
    float4 *o = reinterpret_cast<float4 *>(out + s);
    o[0] = make_float4(p0.x, p0.y, p0.z, __uint_as_float(prim));
    o[1] = make_float4(e1.x, e1.y, e1.z, 0.0f);
    o[2] = make_float4(e2.x, e2.y, e2.z, 0.0f);
}

// per-triangle leaf boxes (what src/scene/mesh.cpp:88-98 computes on the host) + mesh bounds
__global__ void __launch_bounds__(kBlock) k_tri_boxes(Bbox *__restrict__ leaf, const uint32_t *__restrict__ indices, const V3 *__restrict__ positions, uint32_t n) {
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    uint32_t i0 = indices[3 * i], i1 = indices[3 * i + 1], i2 = indices[3 * i + 2];
    V3 p0 = positions[i0], p1 = positions[i1], p2 = positions[i2];
    V3 lo = gmin3(p0, gmin3(p1, p2)), hi = gmax3(p0, gmax3(p1, p2));
    float4 *o = reinterpret_cast<float4 *>(leaf + i);
    o[0] = make_float4(lo.x, lo.y, lo.z, 1.0f);
    o[1] = make_float4(hi.x, hi.y, hi.z, 1.0f);
}

inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

}  // namespace

size_t BuildWorkspace::bytes_for(uint32_t n) {
    if (n == 0) return 0;
    size_t tiles = cdiv(n, kTile);
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    return up(8 * (size_t)n) * 2               // keys ping/pong
         + up(32 * (size_t)n)                  // unsorted leaf boxes
         + up(4 * (2 * (size_t)n))             // parents
         + up(4 * (size_t)n)                   // arrival counters
         + up(4 * tiles * kBins)               // tile histograms (bin-major)
         + up(4 * kBins);                      // bin totals
}

int BuildWorkspace::reserve(uint32_t n) {
    size_t need = bytes_for(n);
    if (need <= capacity) return 0;
    if (base) { HIP_OK(hipFree(base)); base = nullptr; capacity = 0; }
    HIP_OK(hipMalloc(&base, need));
    capacity = need;
    return 0;
}

void BuildWorkspace::release() {
    if (base) (void)hipFree(base);
    base = nullptr;
    capacity = 0;
}

int lbvh_build(AccelNode *nodes, Bbox *bboxes, const Bbox &merged, uint32_t n, hipStream_t stream, BuildWorkspace *ws_in,
               const Bbox *leaf_src_in, uint64_t *keys_out, uint32_t *parents_out, float *stage_ms) {
    if (n == 0) return 0;
    BuildWorkspace local;
    BuildWorkspace *ws = ws_in ? ws_in : &local;
    int rc = ws->reserve(n);
    if (rc) return rc;
    const size_t tiles = cdiv(n, kTile);
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    char *p = static_cast<char *>(ws->base);
    uint64_t *keys_a = reinterpret_cast<uint64_t *>(p); p += up(8 * (size_t)n);
    uint64_t *keys_b = reinterpret_cast<uint64_t *>(p); p += up(8 * (size_t)n);
    Bbox *leaf_tmp = reinterpret_cast<Bbox *>(p); p += up(32 * (size_t)n);
    uint32_t *parents = reinterpret_cast<uint32_t *>(p); p += up(4 * (2 * (size_t)n));
    uint32_t *arrivals = reinterpret_cast<uint32_t *>(p); p += up(4 * (size_t)n);
    uint32_t *tile_hist = reinterpret_cast<uint32_t *>(p); p += up(4 * tiles * kBins);
    uint32_t *bin_total = reinterpret_cast<uint32_t *>(p);

    hipEvent_t ev[6];
    if (stage_ms) for (auto &e : ev) HIP_OK(hipEventCreate(&e));
    auto mark = [&](int i) { if (stage_ms) (void)hipEventRecord(ev[i], stream); };

    const uint32_t grid_n = cdiv(n, kBlock);
    // The leaf boxes are permuted by the sort; a parallel gather needs a source that is not the
    // destination.  Callers that produce the boxes on the device hand them over in leaf_src_in,
    // the in-place entry point (kernel::BuildAccel's contract) copies them aside first.
    const Bbox *leaf_src = leaf_src_in;
    mark(0);
    if (!leaf_src) {
        HIP_OK(hipMemcpyAsync(leaf_tmp, bboxes + (n - 1), sizeof(Bbox) * (size_t)n, hipMemcpyDeviceToDevice, stream));
        leaf_src = leaf_tmp;
    }
    hipLaunchKernelGGL(k_morton, dim3(grid_n), dim3(kBlock), 0, stream, keys_a, leaf_src, merged, n);
    mark(1);
    uint64_t *src = keys_a, *dst = keys_b;
    if (n > 1) {
        for (int pass = 0; pass < 4; pass++) {
            int shift = 32 + pass * kRadixBits;
            hipLaunchKernelGGL(k_hist, dim3((uint32_t)tiles), dim3(kBlock), 0, stream, src, n, shift, tile_hist, (uint32_t)tiles);
            hipLaunchKernelGGL(k_scan_bins, dim3(kBins), dim3(kBlock), 0, stream, tile_hist, (uint32_t)tiles, bin_total);
            hipLaunchKernelGGL(k_scatter, dim3((uint32_t)tiles), dim3(kBlock), 0, stream, src, dst, n, shift, tile_hist, bin_total, (uint32_t)tiles);
            uint64_t *t = src; src = dst; dst = t;
        }
    }
    mark(2);
    hipLaunchKernelGGL(k_gather_leaves, dim3(grid_n), dim3(kBlock), 0, stream, src, leaf_src, bboxes, nodes, n);
    mark(3);
    if (n > 1) {
        HIP_OK(hipMemsetAsync(arrivals, 0, 4 * (size_t)(n - 1), stream));
        hipLaunchKernelGGL(k_karras, dim3(cdiv(n - 1, kBlock)), dim3(kBlock), 0, stream, nodes, parents, src, n);
        mark(4);
        hipLaunchKernelGGL(k_refit, dim3(grid_n), dim3(kBlock), 0, stream, parents, nodes, bboxes, arrivals, merged, n);
    } else {
        mark(4);
    }
    mark(5);
    if (keys_out) HIP_OK(hipMemcpyAsync(keys_out, src, 8 * (size_t)n, hipMemcpyDeviceToDevice, stream));
    if (parents_out && n > 1) hipLaunchKernelGGL(k_strip_parents, dim3(cdiv(2 * n - 1, kBlock)), dim3(kBlock), 0, stream, parents_out, parents, 2 * n - 1);
    HIP_OK(hipGetLastError());
    if (stage_ms) {
        HIP_OK(hipStreamSynchronize(stream));
        for (int i = 0; i < 5; i++) HIP_OK(hipEventElapsedTime(&stage_ms[i], ev[i], ev[i + 1]));
        for (auto &e : ev) (void)hipEventDestroy(e);
    }
    if (!ws_in) {
        HIP_OK(hipStreamSynchronize(stream));   // the temporaries die with this call, as in the reference
        local.release();
    }
    return 0;
}

int lbvh_triangle_boxes(Bbox *leaf, const uint32_t *indices, const V3 *positions, uint32_t n, hipStream_t stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_tri_boxes, dim3(cdiv(n, kBlock)), dim3(kBlock), 0, stream, leaf, indices, positions, n);
    return (int)hipGetLastError();
}

int lbvh_pack(PackedNode *out_nodes, TriRecord *out_tris, AccelRoot *root, const AccelNode *nodes, const Bbox *bboxes, uint32_t n,
              const uint32_t *indices, const V3 *positions, hipStream_t stream) {
    if (n == 0) return 0;
    int top_level = out_tris == nullptr;
    HIP_OK(hipMemsetAsync(root, 0, sizeof(AccelRoot), stream));
    hipLaunchKernelGGL(k_pack_nodes, dim3(cdiv(n > 1 ? n - 1 : 1, kBlock)), dim3(kBlock), 0, stream, out_nodes, nodes, bboxes, n, top_level, root);
    if (!top_level) hipLaunchKernelGGL(k_pack_tris, dim3(cdiv(n, kBlock)), dim3(kBlock), 0, stream, out_tris, nodes, indices, positions, n);
    return (int)hipGetLastError();
}

}  // namespace cpt
