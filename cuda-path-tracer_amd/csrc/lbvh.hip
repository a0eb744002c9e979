// Linear BVH builder for gfx950: 30-bit Morton keys, LDS-staged stable LSD radix sort, one
// bottom-up pass for leaf order, topology and boxes (in LDS inside a workgroup's leaves, through
// memory above them), and packing into 64-byte traversal records.
//
// Results are bit-identical to the reference builder (src/kernels/accel/accel_build.cu):
//   keys      : accel_build.cu:25-41  (same IEEE expression, 0/0 on a flat axis -> 1023)
//   order     : accel_build.cu:193    (ascending 64-bit keys; keys are unique because the low
//               word is the primitive index, and the indices enter in ascending order, so a
//               STABLE sort on the 30 code bits alone yields the same permutation)
//   topology  : accel_build.cu:57-138 (Karras 2012: the binary radix tree of the 64-bit keys with
//               its node numbering, grown here from the leaves instead of searched from the top)
//   boxes     : accel_build.cu:140-180 (min/max are exact and order independent; the sign-split
//               integer atomics there order -0 below +0, reproduced here with a total order)
#include "lbvh.h"

#include <hip/hip_runtime.h>

#include <mutex>

namespace cpt {

namespace {

constexpr int kBlock = 256;                 // 4 waves of 64
constexpr int kItems = 16;                  // keys per thread in the sort kernels (8 for small inputs: more, shorter workgroups)
constexpr int kSmallItems = 8;
constexpr uint32_t kSmallSort = 3000000;    // inputs up to this size sort with the small tile
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

// The key kernel of the default sort variant: one workgroup per tile of the first sort pass.  With FUSED it also
// writes that tile's digit counts for the first pass (the keys are in registers anyway: the pass then needs no
// histogram kernel) and clears the exchange words of the hierarchy pass.  Two launches fewer pay while the build
// is bound by launches and their tails (1M keys: 0.170 -> 0.164 ms); at 10M the shared-memory atomics and the
// extra stores cost this kernel what the two removed ones took, so large inputs run it plain, one key per lane
// (16 keys per lane, as the sort's tiles are cut, is slower there: 0.088 against 0.078 ms).
template <int ITEMS, bool FUSED>
__global__ void __launch_bounds__(kBlock) k_morton_tile(uint64_t *__restrict__ keys, const Bbox *__restrict__ leaf, Bbox merged, uint32_t n,
                                                         uint32_t *__restrict__ tile_hist, uint32_t num_tiles, uint32_t *__restrict__ slots) {
    __shared__ uint32_t hist[kBins];
    if (FUSED) {
        hist[threadIdx.x] = 0;
        __syncthreads();
    }
    const float sx = merged.pmax.x - merged.pmin.x, sy = merged.pmax.y - merged.pmin.y, sz = merged.pmax.z - merged.pmin.z;
    const uint32_t base = blockIdx.x * (kBlock * ITEMS);
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const uint32_t i = base + k * kBlock + threadIdx.x;
        if (i < n) {
            const float4 *p = reinterpret_cast<const float4 *>(leaf + i);
            float4 lo = p[0], hi = p[1];
            float cx = (lo.x + hi.x) * 0.5f, cy = (lo.y + hi.y) * 0.5f, cz = (lo.z + hi.z) * 0.5f;
            float px = (cx - merged.pmin.x) / sx, py = (cy - merged.pmin.y) / sy, pz = (cz - merged.pmin.z) / sz;
            uint32_t x = morton_spread(f2u_sat(fminf(px * 1024.0f, 1023.0f)));
            uint32_t y = morton_spread(f2u_sat(fminf(py * 1024.0f, 1023.0f)));
            uint32_t z = morton_spread(f2u_sat(fminf(pz * 1024.0f, 1023.0f)));
            uint32_t code = (x << 2) | (y << 1) | z;
            keys[i] = ((uint64_t)code << 32) | i;
            if (FUSED) {
                slots[i] = 0;               // n - 1 exchange words are used; the array holds n
                atomicAdd(&hist[code & (kBins - 1)], 1u);
            }
        }
    }
    if (FUSED) {
        __syncthreads();
        tile_hist[(size_t)threadIdx.x * num_tiles + blockIdx.x] = hist[threadIdx.x];   // bin-major, as k_hist leaves it
    }
}

constexpr int kHistWide = 4;                // the large configuration's histogram kernel: 4 x 256 threads, a quarter of the keys per lane
constexpr int kPasses = 4;                  // 8-bit digits of key bits 32..63 (the 30 code bits)
constexpr uint32_t kMortonBlocks = 1024;    // persistent workgroups of the key kernel

// Keys (accel_build.cu:25-41) and, on the way, the global histogram of each of the four digits the
// sort will use: the sort itself then reads the keys once per pass and never just to count them.
__global__ void __launch_bounds__(kBlock) k_morton_hist(uint64_t *__restrict__ keys, const Bbox *__restrict__ leaf, Bbox merged, uint32_t n,
                                                         uint32_t *__restrict__ digit_hist) {
    __shared__ uint32_t hist[kPasses][kBins];
#pragma unroll
    for (int q = 0; q < kPasses; q++) hist[q][threadIdx.x] = 0;
    __syncthreads();
    const float sx = merged.pmax.x - merged.pmin.x, sy = merged.pmax.y - merged.pmin.y, sz = merged.pmax.z - merged.pmin.z;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const float4 *p = reinterpret_cast<const float4 *>(leaf + i);
        float4 lo = p[0], hi = p[1];
        float cx = (lo.x + hi.x) * 0.5f, cy = (lo.y + hi.y) * 0.5f, cz = (lo.z + hi.z) * 0.5f;
        float px = (cx - merged.pmin.x) / sx, py = (cy - merged.pmin.y) / sy, pz = (cz - merged.pmin.z) / sz;
        uint32_t x = morton_spread(f2u_sat(fminf(px * 1024.0f, 1023.0f)));
        uint32_t y = morton_spread(f2u_sat(fminf(py * 1024.0f, 1023.0f)));
        uint32_t z = morton_spread(f2u_sat(fminf(pz * 1024.0f, 1023.0f)));
        uint32_t code = (x << 2) | (y << 1) | z;
        keys[i] = ((uint64_t)code << 32) | i;
#pragma unroll
        for (int q = 0; q < kPasses; q++) atomicAdd(&hist[q][(code >> (q * kRadixBits)) & (kBins - 1)], 1u);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kPasses; q++) {
        uint32_t c = hist[q][threadIdx.x];
        if (c) atomicAdd(&digit_hist[q * kBins + threadIdx.x], c);
    }
}

// ------------------------------------------------------------------------------------------
// Radix sort, "onesweep" variant: digit histograms are known up front (k_morton_hist), so each of the four
// 8-bit passes is ONE kernel that reads every key once and writes it once.  A tile of 4096 keys
// ranks its keys in LDS, publishes its 256 digit counts, and finds the number of equal digits in
// all earlier tiles by looking back over the published words of its predecessors (decoupled
// look-back: a word is either a tile's own count or already the inclusive sum up to that tile).
// Stable, which the LBVH needs: indices enter ascending, so sorting the code bits stably equals
// sorting the unique 64-bit keys.
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

// one workgroup per pass: digit histogram -> first output position of each digit
__global__ void __launch_bounds__(kBlock) k_digit_bases(uint32_t *__restrict__ digit_hist) {
    __shared__ uint32_t tmp[kWaves + 1];
    uint32_t total;
    uint32_t *h = digit_hist + blockIdx.x * kBins;
    uint32_t ex = block_excl_scan(h[threadIdx.x], tmp, total);
    h[threadIdx.x] = ex;
}

constexpr uint32_t kSortBlocks = 1024;            // persistent workgroups of a sort pass: 4 per CU, what the 38 KB of LDS each allow
constexpr uint32_t kTileCount = 0x40000000u;      // the word is this tile's own digit count
constexpr uint32_t kTilePrefix = 0x80000000u;     // the word is the count of this and all earlier tiles
constexpr uint32_t kTileValue = 0x3fffffffu;
constexpr int kLookBackWindow = 4;                // predecessor tiles examined per round trip (1, 4, 8, 16 measured: 0.450, 0.429, 0.437, 0.455 ms for the 10M sort)
constexpr uint32_t kLookBackSpins = 1u << 21;     // far beyond any wait seen: a lost predecessor becomes a reported error, not a hang

__global__ void __launch_bounds__(kBlock) k_onesweep(const uint64_t *__restrict__ keys_in, uint64_t *__restrict__ keys_out, uint32_t n, int shift,
                                                      const uint32_t *__restrict__ digit_base, uint32_t *tile_status, uint32_t *ticket, uint32_t *failed) {
    __shared__ uint64_t staged[kTile];           // keys in tile-sorted order
    __shared__ uint32_t cnt[kWaves][kBins];      // per-wave digit counts, then exclusive over waves
    __shared__ uint32_t tile_off[kBins];         // first staged slot of each digit
    __shared__ uint32_t gbase[kBins];            // global destination of staged slot 0 of each digit
    __shared__ uint32_t tmp[kWaves + 1];
    __shared__ uint32_t s_tile;

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const uint32_t num_tiles = (n + kTile - 1) / kTile;
    // Workgroups are persistent and take tile numbers from a ticket counter: a tile's predecessors
    // have all been taken by workgroups that are running (or done), whatever order the hardware
    // dispatches workgroups in, and the launch cost of a workgroup is paid once, not per tile.
    for (;;) {
    __syncthreads();                              // LDS of the previous tile is no longer read
    if (t == 0) s_tile = atomicAdd(ticket, 1u);
#pragma unroll
    for (int i = 0; i < kWaves; i++) cnt[i][t] = 0;
    __syncthreads();
    const uint32_t tile = s_tile;
    if (tile >= num_tiles) return;
    const uint32_t tile_base = tile * kTile;
    const uint32_t tile_n = min((uint32_t)kTile, n - tile_base);

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

    // digit t: exclusive scan over waves, tile total
    uint32_t total_d = 0;
#pragma unroll
    for (int i = 0; i < kWaves; i++) {
        uint32_t c = cnt[i][t];
        cnt[i][t] = total_d;
        total_d += c;
    }
    // publish this tile's count of digit t, then sum the predecessors' words
    uint32_t *mine = tile_status + (size_t)tile * kBins + t;
    __hip_atomic_store(mine, (tile == 0 ? kTilePrefix : kTileCount) | total_d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t before_tiles = 0;
    bool lost = false;
    if (tile != 0) {
        // A look-back step is a device-scope load, a round trip past this XCD's L2 of a microsecond or more.  The words of
        // kLookBackWindow predecessors are therefore asked for together and consumed nearest first, up to the first one that
        // is a prefix (done) or not yet published (asked for again, with the ones behind it).
        int next = (int)tile - 1;
        uint32_t spins = 0;
        for (;;) {
            uint32_t word[kLookBackWindow];
#pragma unroll
            for (int k = 0; k < kLookBackWindow; k++) {
                const int idx = next - k;
                word[k] = idx >= 0 ? __hip_atomic_load(tile_status + (size_t)idx * kBins + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kTilePrefix;
            }
            int used = 0;
            bool done = false;
#pragma unroll
            for (int k = 0; k < kLookBackWindow; k++) {
                if (!done && used == k && word[k] != 0u) {
                    before_tiles += word[k] & kTileValue;
                    used = k + 1;
                    done = (word[k] & kTilePrefix) != 0u;
                }
            }
            if (done) break;
            next -= used;
            if (used == 0 && ++spins > kLookBackSpins) { lost = true; break; }
        }
        // published even after a give-up, so that no successor waits for this tile in turn
        __hip_atomic_store(mine, kTilePrefix | ((before_tiles + total_d) & kTileValue), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (__syncthreads_or(lost)) {               // nothing is written from offsets that are not known
        if (t == 0) *failed = 1u;
        continue;
    }
    uint32_t all;
    uint32_t off = block_excl_scan(total_d, tmp, all);
    tile_off[t] = off;
    gbase[t] = digit_base[t] + before_tiles - off;
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
}

// ------------------------------------------------------------------------------------------
// Radix sort, default variant: per pass a tile histogram, a per-digit scan over the tiles and the
// same stable LDS-staged scatter.  It reads the keys twice per pass, but no workgroup ever waits
// for another one; on this part that is faster than the look-back of the one-kernel pass above,
// whose every step is a round trip past the per-XCD L2s (10M keys: 0.35 ms against 0.36-0.49 ms).
// ------------------------------------------------------------------------------------------
template <int ITEMS, int THREADS>
__global__ void __launch_bounds__(THREADS) k_hist(const uint64_t *__restrict__ keys, uint32_t n, int shift, uint32_t *__restrict__ tile_hist,
                                                  uint32_t num_tiles) {
    __shared__ uint32_t hist[kBins];
    if (threadIdx.x < kBins) hist[threadIdx.x] = 0;
    __syncthreads();
    uint32_t base = blockIdx.x * (THREADS * ITEMS);
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        uint32_t i = base + k * THREADS + threadIdx.x;
        if (i < n) {
            uint32_t d = (uint32_t)(keys[i] >> shift) & (kBins - 1);       // plain load: the scatter kernel reads the same keys again, out of the caches (non-temporal: 0.37 instead of 0.31 ms)
            atomicAdd(&hist[d], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < kBins) tile_hist[(size_t)threadIdx.x * num_tiles + blockIdx.x] = hist[threadIdx.x];   // bin-major
}

// one workgroup per bin: exclusive scan of that bin's counts over all tiles, total to bin_total.  A lane takes kScanRun
// consecutive counts, all loaded before the first is used, so up to 4096 tiles (16.7M keys) are one round of loads and
// one scan over the workgroup instead of a round per 256 tiles.
constexpr int kScanRun = 16;
//
// The scan keeps one workgroup per CU busy for a few microseconds and nothing else runs beside it, so the large
// configuration hands it a share of a chore that is pure bandwidth: workgroups past the first kBins zero 4 KB each of
// `clear` (the hierarchy pass's exchange words, a quarter of them per sort pass) instead of a 40 MB memset between the
// sort and the hierarchy pass.
__global__ void __launch_bounds__(kBlock) k_scan_bins(uint32_t *__restrict__ tile_hist, uint32_t num_tiles, uint32_t *__restrict__ bin_total, uint4 *__restrict__ clear,
                                                      uint32_t clear_count) {
    if (blockIdx.x >= kBins) {
        const uint32_t i = (blockIdx.x - kBins) * kBlock + threadIdx.x;
        if (i < clear_count) clear[i] = make_uint4(0u, 0u, 0u, 0u);
        return;
    }
    __shared__ uint32_t tmp[kWaves + 1];
    uint32_t *row = tile_hist + (size_t)blockIdx.x * num_tiles;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < num_tiles; base += kBlock * kScanRun) {
        const uint32_t first = base + threadIdx.x * kScanRun;
        uint32_t v[kScanRun];
#pragma unroll
        for (int j = 0; j < kScanRun; j++) v[j] = first + j < num_tiles ? row[first + j] : 0u;
        uint32_t sum = 0;
#pragma unroll
        for (int j = 0; j < kScanRun; j++) { const uint32_t x = v[j]; v[j] = sum; sum += x; }
        uint32_t total;
        const uint32_t ex = carry + block_excl_scan(sum, tmp, total);
#pragma unroll
        for (int j = 0; j < kScanRun; j++) if (first + j < num_tiles) row[first + j] = ex + v[j];
        carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) bin_total[blockIdx.x] = carry;
}

__device__ inline uint32_t xcd_band(uint32_t b, uint32_t nb) {
    const uint32_t q = nb / 8, r = nb % 8, xcd = b % 8, k = b / 8;
    const uint32_t start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + k;
}

template <int ITEMS, bool BANDS>
__global__ void __launch_bounds__(kBlock, 4) k_scatter(const uint64_t *__restrict__ keys_in, uint64_t *__restrict__ keys_out, uint32_t n, int shift,
                                                     const uint32_t *__restrict__ tile_hist, const uint32_t *__restrict__ bin_total,
                                                     uint32_t num_tiles) {
    __shared__ uint64_t staged[(kBlock * ITEMS)];           // keys in tile-sorted order
    __shared__ uint32_t cnt[kWaves][kBins];      // per-wave digit counts, then exclusive over waves
    __shared__ uint32_t tile_off[kBins];         // first staged slot of each digit
    __shared__ uint32_t gbase[kBins];            // global destination of staged slot 0 of each digit
    __shared__ uint32_t tmp[kWaves + 1];

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    // Workgroups are dealt to the 8 XCDs round robin.  A tile's run of a digit is followed in memory by the next tile's run
    // of that digit, and the two share a cache line; with one contiguous band of tiles per XCD (BANDS) both halves of that
    // line are written through the same L2, which can then write it out whole: 10M keys sort in 0.301 instead of 0.321 ms.
    // Small inputs, whose keys stay in the caches anyway, lose by it (1M: 0.080 against 0.074 ms) and keep the plain order.
    const uint32_t tile = BANDS ? xcd_band(blockIdx.x, gridDim.x) : blockIdx.x;
    const uint32_t tile_base = tile * (kBlock * ITEMS);
    const uint32_t tile_n = min((uint32_t)(kBlock * ITEMS), n - tile_base);
    // digit t's totals are not needed before the scatter: fetch them while the ranking runs
    const uint32_t bt = bin_total[t];
    const uint32_t earlier_tiles = tile_hist[(size_t)t * num_tiles + tile];
#pragma unroll
    for (int i = 0; i < kWaves; i++) cnt[i][t] = 0;
    __syncthreads();

    // Each wave owns a contiguous quarter of the tile and walks it 64 keys at a time, so key
    // order within the tile is (wave, round, lane): ranking in that order keeps the sort stable.
    uint64_t key[ITEMS];
    uint32_t rank[ITEMS];
    const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        uint32_t li = w * (64 * ITEMS) + r * 64 + lane;
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
    uint32_t bin_base = block_excl_scan(bt, tmp, all);
    gbase[t] = bin_base + earlier_tiles - off;
    __syncthreads();

#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        uint32_t li = w * (64 * ITEMS) + r * 64 + lane;
        if (li < tile_n) {
            uint32_t d = (uint32_t)(key[r] >> shift) & (kBins - 1);
            staged[tile_off[d] + cnt[w][d] + rank[r]] = key[r];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        uint32_t i = k * kBlock + t;
        if (i < tile_n) {
            uint64_t kk = staged[i];
            uint32_t d = (uint32_t)(kk >> shift) & (kBins - 1);
            keys_out[gbase[d] + i] = kk;
        }
    }
}

// The same scatter for large inputs with WAVES waves per workgroup (a tile is still WAVES * 64 * ITEMS keys): a streaming
// kernel on this part runs fastest with few elements per lane and many waves in flight (see k_stream_copy in render.hip),
// so the 4096-key tile of the large configuration is walked by 8 waves of 8 keys per lane instead of 4 waves of 16: the 10M
// sort takes 0.267 ms instead of 0.292 (16 waves of 4 keys: 0.281).  The histogram kernel of these passes runs 16 waves of
// 4 keys for the same reason (0.306 -> 0.293 ms).
constexpr int kWideWaves = 8;
template <int ITEMS, int WAVES, bool BANDS>
__global__ void __launch_bounds__(WAVES * 64) k_scatter_wide(const uint64_t *__restrict__ keys_in, uint64_t *__restrict__ keys_out, uint32_t n, int shift,
                                                             const uint32_t *__restrict__ tile_hist, const uint32_t *__restrict__ bin_total, uint32_t num_tiles) {
    constexpr int THREADS = WAVES * 64, TILE = THREADS * ITEMS;
    __shared__ uint64_t staged[TILE];            // keys in tile-sorted order
    __shared__ uint32_t cnt[WAVES][kBins];       // per-wave digit counts, then exclusive over waves
    __shared__ uint32_t tile_off[kBins];         // first staged slot of each digit
    __shared__ uint32_t gbase[kBins];            // global destination of staged slot 0 of each digit
    __shared__ uint32_t tmp[2][kWaves];

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const uint32_t tile = BANDS ? xcd_band(blockIdx.x, gridDim.x) : blockIdx.x;
    const uint32_t tile_base = tile * TILE;
    const uint32_t tile_n = min((uint32_t)TILE, n - tile_base);
    uint32_t bt = 0, earlier_tiles = 0;
    if (t < kBins) {
        bt = bin_total[t];
        earlier_tiles = tile_hist[(size_t)t * num_tiles + tile];
    }
    for (int i = t; i < WAVES * kBins; i += THREADS) (&cnt[0][0])[i] = 0;
    __syncthreads();

    uint64_t key[ITEMS];
    uint32_t rank[ITEMS];
    const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        uint32_t li = w * (64 * ITEMS) + r * 64 + lane;
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

    // digit t (the first 256 threads): exclusive scan over waves, tile total, then the two scans over digits, done by the
    // first four waves with one value per thread
    uint32_t total_d = 0, inc_tile = 0, inc_bins = 0;
    if (t < kBins) {
#pragma unroll
        for (int i = 0; i < WAVES; i++) {
            uint32_t c = cnt[i][t];
            cnt[i][t] = total_d;
            total_d += c;
        }
        inc_tile = wave_incl_scan(total_d);
        inc_bins = wave_incl_scan(bt);
        if (lane == 63) { tmp[0][w] = inc_tile; tmp[1][w] = inc_bins; }
    }
    __syncthreads();
    if (t < kBins) {
        uint32_t base_tile = 0, base_bins = 0;
#pragma unroll
        for (int i = 0; i < kWaves; i++) {
            if (i < w) { base_tile += tmp[0][i]; base_bins += tmp[1][i]; }
        }
        const uint32_t off = base_tile + inc_tile - total_d;
        tile_off[t] = off;
        gbase[t] = (base_bins + inc_bins - bt) + earlier_tiles - off;
    }
    __syncthreads();

#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        uint32_t li = w * (64 * ITEMS) + r * 64 + lane;
        if (li < tile_n) {
            uint32_t d = (uint32_t)(key[r] >> shift) & (kBins - 1);
            staged[tile_off[d] + cnt[w][d] + rank[r]] = key[r];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        uint32_t i = k * THREADS + t;
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

__device__ inline uint32_t lcp(uint64_t a, uint64_t b) {           // accel_build.cu:53-55
    uint64_t x = a ^ b;
    return x ? (uint32_t)__builtin_clzll(x) : 64u;
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

// ------------------------------------------------------------------------------------------
// Hierarchy in one bottom-up pass: leaf gather + topology + boxes.
//
// The tree the reference builds top-down (accel_build.cu:57-138) is the binary radix tree of the
// sorted keys, which is unique because the keys are; only the numbering is Karras': the left child
// of a split between sorted positions g and g+1 is node g (leaf n-1+g when its range is the single
// key g), the right child is node g+1 (leaf n-1+g+1), the root is node 0.  The same tree grows from
// the leaves (Apetrei 2014): a node covering sorted keys [l, r] is merged first with the neighbour
// it shares the longer prefix with, d(j) = lcp(key j, key j+1): if d(r) > d(l-1) it is the LEFT
// child of the split g = r, otherwise the RIGHT child of the split g = l-1 (d(-1) = d(n-1) = -1;
// d(r) = d(l-1) cannot happen with distinct keys).  That fixes the node's own number (r or l), so
// it can write its record and box, and hand over to the sibling through one exchange word per
// split: the first child to arrive leaves the far end of its range there and stops, the second
// one picks it up -- it then knows the parent's range, both child numbers, and where the
// sibling's box is -- merges, and goes on as the parent.  No binary searches, no separate parents
// array, one kernel instead of gather + topology + refit.
//
// A split whose parent covers only leaves of one workgroup is exchanged through LDS.  Both
// children must take the same decision knowing only their own half of the parent's range, so
// the workgroup precomputes it per split from the prefix lengths it can see (one more than it has
// leaves): the parent of split g stays inside the workgroup's leaves [b0, b1] iff some d(j) < d(g)
// with b0-1 <= j < g and some d(j) < d(g) with g < j <= b1 (two running minima); the exchange word
// of every other split starts as a marker that sends the child to the memory pass.  Inside the
// workgroup a node is a pair of leaf positions with the two prefix lengths just outside it, and the
// exchange word carries the far end together with the prefix length beyond it, so a level is one
// LDS exchange and one 32-byte LDS read (the pass is bound by its gather of the leaf boxes: with
// random primitive order a 32-byte read per leaf moves a whole memory line).  All other splits use
// memory: device-scope write-through stores of the child box, a device-scope exchange, and
// device-scope loads of the sibling box.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kSlotValid = 0x80000000u;
// Leaves per workgroup of the bottom-up pass.  512 for inputs up to kSmallSort keys (256 / 512 / 1024: 0.157 / 0.154 / 0.156 ms
// for the 1M build), 1024 above: with 35 KB of LDS per 512 leaves two workgroups of 1024 share a CU, half as many nodes are
// left for the pass through memory, and the 10M build takes 0.803 instead of 0.834 ms (256: 0.903).  The scratch is sized
// for the smaller groups.
constexpr int kUpBlock = 512;
constexpr int kUpBlockLarge = 1024;

__device__ inline int wave_incl_min(int v) {
    int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (lane >= d) v = min(v, t);
    }
    return v;
}
__device__ inline int wave_incl_min_rev(int v) {       // suffix minimum
    int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_down(v, d, 64);
        if (lane + d < 64) v = min(v, t);
    }
    return v;
}

// Boxes travel through the pass as order keys (order_key(): unsigned integers whose order is the
// total order above), so a merge is six integer min/max operations; they are turned back into
// floats, and folded with the inverted merged box every internal box starts from
// (accel_build.cu:140-148), only where a box is written.
__device__ inline float key_to_float(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

// A node that still has to meet its sibling, handed from the workgroup pass to the memory pass.
struct Carrier { uint32_t l, r, idx, pad; };
constexpr uint32_t kCarriersPerGroup = 64;       // per workgroup of the first pass; the (rare) rest climbs in place

// The climb through splits whose parent spans several workgroups: exchange words and boxes in
// memory.  On entry node `idx` covers sorted keys [l, r], its record is written and its box is in
// lo/hi; `in_memory` tells whether the box is already visible in bboxes[idx] to every CU (written
// by an earlier kernel), otherwise it is stored write-through (w components lw, hw) first.
template <bool PARENTS>
__device__ inline void climb_global(uint32_t l, uint32_t r, uint32_t idx, uint32_t lo[3], uint32_t hi[3], bool in_memory, float lw, float hw,
                                    const uint64_t *__restrict__ keys, Bbox *bboxes, AccelNode *nodes, uint32_t *parents, uint32_t *slots,
                                    const Bbox &merged, uint32_t n) {
    const uint32_t init_lo[3] = {order_key(merged.pmax.x), order_key(merged.pmax.y), order_key(merged.pmax.z)};
    const uint32_t init_hi[3] = {order_key(merged.pmin.x), order_key(merged.pmin.y), order_key(merged.pmin.z)};
    auto delta_before = [&](uint32_t j1) -> int {          // prefix shared by sorted keys j1-1 and j1
        if (j1 == 0 || j1 >= n) return -1;
        return (int)lcp(keys[j1 - 1], keys[j1]);
    };
    uint32_t lc = 0, rc = 0;
    bool have_record = true;
    // While n < 2^24 an exchange word has room for the prefix length beyond the range's far end next to the far
    // end itself, so the node that merges two ranges knows both of its outer prefix lengths without reading keys
    // again: one memory round trip less per level of a climb whose levels are nothing but round trips.
    const bool packed = n < (1u << 24);
    int dl = delta_before(l), dr = delta_before(r + 1);
    for (;;) {
        const bool left_child = dr > dl;
        const uint32_t g = left_child ? r : l - 1;
        if (!have_record) {
            idx = left_child ? r : l;
            nodes[idx] = AccelNode{lc, rc};
            if (PARENTS) { parents[lc] = idx; parents[rc] = idx; }
        }
        if (!in_memory) {
            f32x4 olo, ohi;
            if (have_record) {          // the entry node as it is (a leaf keeps its own box and w)
                olo = f32x4{key_to_float(lo[0]), key_to_float(lo[1]), key_to_float(lo[2]), lw};
                ohi = f32x4{key_to_float(hi[0]), key_to_float(hi[1]), key_to_float(hi[2]), hw};
            } else {
                olo = f32x4{key_to_float(min(init_lo[0], lo[0])), key_to_float(min(init_lo[1], lo[1])), key_to_float(min(init_lo[2], lo[2])), merged.pmax.w};
                ohi = f32x4{key_to_float(max(init_hi[0], hi[0])), key_to_float(max(init_hi[1], hi[1])), key_to_float(max(init_hi[2], hi[2])), merged.pmin.w};
            }
            store_box_sc1(bboxes + idx, olo, ohi);
        }
        uint32_t far = (left_child ? l : r) | kSlotValid;
        if (packed) far |= (uint32_t)((left_child ? dl : dr) + 1) << 24;
        const uint32_t prev = __hip_atomic_exchange(&slots[g], far, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == 0) return;
        const uint32_t sib_far = packed ? (prev & 0x00ffffffu) : (prev & ~kSlotValid);
        uint32_t sib;
        if (left_child) {                                  // sibling = right child, keys [g+1, sib_far]
            sib = sib_far == g + 1 ? (n - 1) + g + 1 : g + 1;
            lc = idx; rc = sib; r = sib_far;
            dr = packed ? (int)((prev >> 24) & 0x7fu) - 1 : delta_before(r + 1);
        } else {                                           // sibling = left child, keys [sib_far, g]
            sib = sib_far == g ? (n - 1) + g : g;
            lc = sib; rc = idx; l = sib_far;
            dl = packed ? (int)((prev >> 24) & 0x7fu) - 1 : delta_before(l);
        }
        f32x4 a4, b4;
        load_box_sc1(bboxes + sib, a4, b4);
        lo[0] = min(lo[0], order_key(a4[0])); lo[1] = min(lo[1], order_key(a4[1])); lo[2] = min(lo[2], order_key(a4[2]));
        hi[0] = max(hi[0], order_key(b4[0])); hi[1] = max(hi[1], order_key(b4[1])); hi[2] = max(hi[2], order_key(b4[2]));
        have_record = false;
        in_memory = false;
        if (l == 0 && r == n - 1) {                       // the root is node 0
            nodes[0] = AccelNode{lc, rc};
            if (PARENTS) { parents[lc] = 0; parents[rc] = 0; }
            float4 *gb = reinterpret_cast<float4 *>(bboxes);
            gb[0] = make_float4(key_to_float(min(init_lo[0], lo[0])), key_to_float(min(init_lo[1], lo[1])), key_to_float(min(init_lo[2], lo[2])), merged.pmax.w);
            gb[1] = make_float4(key_to_float(max(init_hi[0], hi[0])), key_to_float(max(init_hi[1], hi[1])), key_to_float(max(init_hi[2], hi[2])), merged.pmin.w);
            return;
        }
    }
}

// second pass: one lane per carrier of the first pass
template <bool PARENTS>
__global__ void __launch_bounds__(kBlock) k_climb(const Carrier *__restrict__ carriers, const uint32_t *__restrict__ counts, uint32_t groups,
                                                   const uint64_t *__restrict__ keys, Bbox *bboxes, AccelNode *nodes, uint32_t *parents, uint32_t *slots,
                                                   Bbox merged, uint32_t n) {
    const uint32_t group = blockIdx.x * (kBlock / kCarriersPerGroup) + threadIdx.x / kCarriersPerGroup, k = threadIdx.x % kCarriersPerGroup;
    if (group >= groups || k >= counts[group]) return;
    const Carrier c = carriers[(size_t)group * kCarriersPerGroup + k];
    const float4 *b = reinterpret_cast<const float4 *>(bboxes + c.idx);       // stored by the first pass
    const float4 a4 = b[0], b4 = b[1];
    uint32_t lo[3] = {order_key(a4.x), order_key(a4.y), order_key(a4.z)}, hi[3] = {order_key(b4.x), order_key(b4.y), order_key(b4.z)};
    climb_global<PARENTS>(c.l, c.r, c.idx, lo, hi, true, a4.w, b4.w, keys, bboxes, nodes, parents, slots, merged, n);
}

// LDS exchange word of a split inside a workgroup: bit 31 = a child has arrived, bits 16-23 = the prefix length
// beyond that child's far end plus one, bits 0-15 = the far end as a leaf position of the workgroup.
constexpr uint32_t kSlotGlobal = 0x40000000u;       // initial word of a split whose parent spans several workgroups

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

constexpr uint32_t kNoChild = 0xffffffffu;       // reference of an empty slot of a two-level record

// Traversal records straight from the hierarchy pass: a node finished inside a workgroup has its children and their
// children in LDS at that moment, so its 64-byte binary record and its 128-byte two-level record (see pack_wide_global
// for the layout) are written then, without reading anything back; local_mask gets one bit per internal node that was
// served this way, and k_pack_rest makes the records of the others (the nodes that span workgroups) from memory.
__device__ inline void store_half(float4 *o, const float a[6], uint32_t ra, const float b[6], uint32_t rb) {
    o[0] = make_float4(a[0], a[1], a[2], a[3]);
    o[1] = make_float4(a[4], a[5], b[0], b[1]);
    o[2] = make_float4(b[2], b[3], b[4], b[5]);
    o[3] = make_float4(__uint_as_float(ra), __uint_as_float(rb), 0.0f, 0.0f);
}
__device__ inline bool six_in_fast_range(const float b[6]) {
    return coord_in_fast_range(b[0]) && coord_in_fast_range(b[1]) && coord_in_fast_range(b[2]) && coord_in_fast_range(b[3]) && coord_in_fast_range(b[4]) &&
           coord_in_fast_range(b[5]);
}

// ---- the four slots of a node's two-level record -------------------------------------------------------------
// A record holds a cut through the subtree below its node: it starts as (right child, left child) and, twice, the
// internal member with the largest box surface is replaced by its two children, right one first -- so the members
// stay in the order the reference's traversal reaches them (right child first, accel.cuh:84-91), and the boxes a ray
// is most likely to enter are the ones that are opened.  Two levels down on both sides is the cut this gives on a
// balanced tree; where one child is small or a leaf the other side is opened three levels deep instead.  Measured
// against the fixed cut of both children's children (a leaf child standing for itself, its half's second slot empty):
// heightfield 10M 27.1 -> 25.3 records per ray and 591 -> 617 Mpaths/s, soup 1M 103.4 -> 101.8 and 222 -> 226; the
// Cornell box, whose trees are a dozen triangles, loses 1.2 % (1760 -> 1740).  Opening the node's other child first
// whenever it is internal and choosing by surface only then: 26.9 records per ray and 594 Mpaths/s, so it is the
// one-sided openings that pay.  Any cut is valid: a member is walked iff its own box is hit within
// the reach, which implies that the boxes of the ancestors the cut skipped are (see Traversal::vote_step).
// Tree: internal(id), children(id, left, right), slot(id, box) -> reference.  Writes the record's seven words that
// are read and the second half's own reference word (device_types.h: PackedNode halves).
__device__ inline float box_surface(const float b[6]) {
    const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    return dx * dy + dy * dz + dz * dx;
}
template <class Tree>
__device__ inline void store_cut_record(float4 *o, const Tree &tree, uint32_t left, uint32_t right) {
    constexpr uint32_t kEmpty = 0xffffffffu;
    uint32_t cut[4] = {right, left, kEmpty, kEmpty};
    float surface[4] = {-1.0f, -1.0f, -1.0f, -1.0f};       // of internal members; leaves and empty places never win
    float box[6];
    auto surface_of = [&](uint32_t id) -> float {
        if (!tree.internal(id)) return -1.0f;
        tree.slot(id, box);
        return box_surface(box);
    };
    surface[0] = surface_of(right);
    surface[1] = surface_of(left);
#pragma unroll
    for (int step = 0; step < 2; step++) {
        int best = -1;
        float most = -1.0f;
#pragma unroll
        for (int i = 0; i < 4; i++) if (surface[i] > most) { best = i; most = surface[i]; }      // equal surfaces: the earlier member
        if (best < 0) break;
        uint32_t opened = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) if (i == best) opened = cut[i];
        uint32_t l, r;
        tree.children(opened, l, r);
        const float sr = surface_of(r), sl = surface_of(l);
#pragma unroll
        for (int i = 3; i > 0; i--) {
            if (i > best + 1) { cut[i] = cut[i - 1]; surface[i] = surface[i - 1]; }
            else if (i == best + 1) { cut[i] = l; surface[i] = sl; }
        }
#pragma unroll
        for (int i = 0; i < 3; i++) if (i == best) { cut[i] = r; surface[i] = sr; }
    }
    const float none[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    float a[6], b[6];
    uint32_t ref[4];
    // second half first: its references are repeated in the first half's last word
    if (cut[2] != kEmpty) ref[2] = tree.slot(cut[2], a); else { ref[2] = kNoChild; for (int i = 0; i < 6; i++) a[i] = none[i]; }
    if (cut[3] != kEmpty) ref[3] = tree.slot(cut[3], b); else { ref[3] = kNoChild; for (int i = 0; i < 6; i++) b[i] = none[i]; }
    store_half(o + 4, a, ref[2], b, ref[3]);
    ref[0] = tree.slot(cut[0], a);
    ref[1] = tree.slot(cut[1], b);
    o[0] = make_float4(a[0], a[1], a[2], a[3]);
    o[1] = make_float4(a[4], a[5], b[0], b[1]);
    o[2] = make_float4(b[2], b[3], b[4], b[5]);
    o[3] = make_float4(__uint_as_float(ref[0]), __uint_as_float(ref[1]), __uint_as_float(ref[2]), __uint_as_float(ref[3]));
}

template <bool PARENTS, int BLOCK, bool RECORDS>
__global__ void __launch_bounds__(BLOCK) k_build_up(const uint64_t *__restrict__ keys, const Bbox *__restrict__ leaf_src, Bbox *bboxes, AccelNode *nodes,
                                                    uint32_t *parents, uint32_t *slots, Carrier *carriers, uint32_t *carrier_counts, uint32_t carrier_limit, Bbox merged, uint32_t n,
                                                    RecordOutputs rec, unsigned long long *local_mask) {
    __shared__ Carrier s_park[kCarriersPerGroup];
    __shared__ uint32_t s_park_count;
    __shared__ int s_wave_min[2][BLOCK / 64];
    __shared__ uint32_t s_slot[BLOCK + 1];       // exchange word of split b0-1+k
    constexpr int kFirstLevels = 4;
    __shared__ uint2 s_queue[BLOCK / (kFirstLevels + 1) + 1];
    __shared__ uint32_t s_queue_count;
    // Entry e < BLOCK: internal node b0+e; entry BLOCK+e: leaf b0+e.  (lo.xyz, hi.x) (hi.yz, left child, right child):
    // boxes as order keys, children as entry numbers; a right child of 0 marks a node this workgroup did not finish
    // (a leaf entry keeps its primitive where an internal one has its left child).
    __shared__ uint4 s_node[2 * BLOCK][2];
    // What the set-up needs and the climb does not -- keys, prefix lengths, their running minima -- lives in the internal
    // entries' half of s_node, which nothing writes before the climb: 35 KB of LDS per workgroup instead of 45, so that
    // four workgroups of 512 leaves share a CU instead of three.
    uint64_t *const s_key = reinterpret_cast<uint64_t *>(&s_node[0][0]);      // keys b0-1 .. b1+1
    int *const s_delta = reinterpret_cast<int *>(s_key + BLOCK + 2);            // d(b0-1) .. d(b1)
    int *const s_lmin = s_delta + BLOCK + 1, *const s_rmin = s_lmin + BLOCK;    // BLOCK and BLOCK + 2 entries
    static_assert(sizeof(uint64_t) * (BLOCK + 2) + sizeof(int) * (3 * BLOCK + 3) <= sizeof(uint4) * 2 * BLOCK, "set-up arrays exceed the internal entries");
    const uint32_t t = threadIdx.x, lane = t & 63, w = t >> 6;
    const uint32_t b0 = blockIdx.x * BLOCK;
    const uint32_t i = b0 + t;
    auto global_id = [&](uint32_t e) { return e < (uint32_t)BLOCK ? b0 + e : (n - 1) + b0 + (e - BLOCK); };

    // ---- keys, leaf gather (accel_build.cu:43-51 and the payload permutation of :193).  The gather's two loads are
    // issued here and their values used after the set-up below, which needs the keys only: with primitives in random
    // order each of them is a memory round trip of its own, and the set-up's four barrier phases run under it.
    uint64_t key = 0;
    float4 box_a = make_float4(0.0f, 0.0f, 0.0f, 0.0f), box_b = box_a;
    if (i < n) {
        key = keys[i];
        const float4 *src = reinterpret_cast<const float4 *>(leaf_src + (uint32_t)key);
        box_a = src[0];
        box_b = src[1];
    }
    s_key[t + 1] = key;
    if (t == 0) s_key[0] = b0 > 0 ? keys[b0 - 1] : 0ull;
    if (t == 1) s_key[BLOCK + 1] = (b0 + BLOCK < n) ? keys[b0 + BLOCK] : 0ull;
    if (t == 0) s_park_count = 0;
    __syncthreads();
    // d(j) for j = b0-1+k, k = 0 .. BLOCK: -1 outside the key array
    auto delta_of_window = [&](uint32_t k) -> int {
        const uint32_t j1 = b0 + k;                         // j + 1
        if (j1 == 0 || j1 >= n) return -1;
        return (int)lcp(s_key[k], s_key[k + 1]);
    };
    const int d_here = delta_of_window(t);
    s_delta[t] = d_here;
    if (t == 0) s_delta[BLOCK] = delta_of_window(BLOCK);
    // running minima: s_lmin[k] = min d(window 0..k), s_rmin[k] = min d(window k..BLOCK)
    int lmin = wave_incl_min(d_here), rmin = wave_incl_min_rev(d_here);
    if (lane == 63) s_wave_min[0][w] = lmin;
    if (lane == 0) s_wave_min[1][w] = rmin;
    __syncthreads();
    for (uint32_t k = 0; k < (uint32_t)(BLOCK / 64); k++) {
        if (k < w) lmin = min(lmin, s_wave_min[0][k]);
        if (k > w) rmin = min(rmin, s_wave_min[1][k]);
    }
    s_lmin[t] = lmin;
    s_rmin[t] = min(rmin, s_delta[BLOCK]);
    if (t == 0) { s_rmin[BLOCK] = s_delta[BLOCK]; s_rmin[BLOCK + 1] = 0x7fffffff; }
    __syncthreads();
    int dl = s_delta[t], dr = s_delta[t + 1];
    {
        // split g = b0 + t has d(g) = window entry t+1; left candidates are entries 0..t, right ones t+2..BLOCK
        s_slot[t + 1] = (dr >= 0 && s_lmin[t] < dr && s_rmin[t + 2] < dr) ? 0u : kSlotGlobal;
        if (t == 0) s_slot[0] = kSlotGlobal;
    }
    __syncthreads();                                        // the set-up arrays have been read for the last time
    s_node[t][1] = make_uint4(0u, 0u, 0u, 0u);
    // The node a lane holds, in the layout of its s_node entry: (lo.xyz, hi.x) (hi.yz, left child, right child).
    uint4 e0 = make_uint4(0u, 0u, 0u, 0u), e1 = e0;
    if (i < n) {
        const uint32_t prim = (uint32_t)key;
        float4 *dst = reinterpret_cast<float4 *>(bboxes + (n - 1) + i);
        dst[0] = box_a;
        dst[1] = box_b;
        nodes[(n - 1) + i] = AccelNode{prim, ~0u};
        e0 = make_uint4(order_key(box_a.x), order_key(box_a.y), order_key(box_a.z), order_key(box_b.x));
        e1 = make_uint4(order_key(box_b.y), order_key(box_b.z), prim, 0u);
    }
    if (t == 0) s_queue_count = 0;
    __syncthreads();

    // ---- the climb inside the workgroup.  A lane holds a node as leaf positions [lr, rr] of this workgroup with the
    // prefix lengths just outside the range (dl, dr: window entries lr and rr + 1); while the parent is a split of this
    // workgroup both stay inside the window, and the exchange word hands the sibling's far end and the prefix
    // length beyond it to the lane that goes on, so a level costs one exchange and one 32-byte LDS read.
    uint32_t lr = t, rr = t, me = BLOCK + t;
    // publish: the node leaves record and box in its LDS entry (an internal node's number lies in this workgroup's
    // range because its keys do); all entries are written to memory together after the climb, by every lane and in
    // index order, instead of one scattered store per level by the few lanes that are still climbing.
    auto publish = [&]() {
        me = (dr > dl ? rr : lr) + (lr == rr ? (uint32_t)BLOCK : 0u);
        s_node[me][0] = e0;
        s_node[me][1] = e1;
    };
    // meet: one exchange with the sibling.  kGoesOn: this lane holds the parent now; kStops: the sibling will (or the
    // node is the root, node 0); kLeaves: the parent spans several workgroups.
    enum { kGoesOn, kStops, kLeaves };
    auto meet = [&]() -> int {
        if ((dl & dr) < 0) return kStops;                   // both are -1: the root
        const bool left_child = dr > dl;
        const uint32_t gs = left_child ? rr + 1 : lr;
        const uint32_t word = kSlotValid | ((uint32_t)((left_child ? dl : dr) + 1) << 16) | (left_child ? lr : rr);
        // LDS executes a wave's operations in issue order, so the entry written by publish() is in place when another
        // wave sees the exchange and reads it afterwards; only the compiler has to be kept from moving the accesses.
        asm volatile("" ::: "memory");
        const uint32_t prev = __hip_atomic_exchange(&s_slot[gs], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");
        if ((int)prev >= 0) return prev == 0u ? kStops : kLeaves;      // 0 or kSlotGlobal: arrivals set the top bit
        const uint32_t sib_far = prev & 0xffffu;
        const int d_far = (int)((prev >> 16) & 0xffu) - 1;
        const uint32_t next = left_child ? rr + 1 : lr - 1;              // the sibling's leaves are [rr + 1, sib_far] or [sib_far, lr - 1]
        const uint32_t sib = next + (sib_far == next ? (uint32_t)BLOCK : 0u);
        const uint4 a = s_node[sib][0];
        const uint2 b = *reinterpret_cast<const uint2 *>(&s_node[sib][1]);
        e0 = make_uint4(min(e0.x, a.x), min(e0.y, a.y), min(e0.z, a.z), max(e0.w, a.w));
        e1 = make_uint4(max(e1.x, b.x), max(e1.y, b.y), left_child ? me : sib, left_child ? sib : me);
        lr = left_child ? lr : sib_far;
        rr = left_child ? sib_far : rr;
        dl = left_child ? dl : d_far;
        dr = left_child ? d_far : dr;
        return kGoesOn;
    };
    // Half of the lanes stop at every level, and a wave runs a level for its last lane as it does for 64.  After
    // kFirstLevels levels the lanes that still hold a node -- of more than kFirstLevels leaves, so there are at most
    // BLOCK / (kFirstLevels + 1) of them -- queue it, and the first waves of the workgroup take the queue from there.
    int state = kStops;
    if (i < n) {
        state = kGoesOn;
        for (int level = 0; level < kFirstLevels && state == kGoesOn; level++) {
            publish();
            state = meet();
        }
        if (state == kGoesOn) {
            publish();
            s_queue[atomicAdd(&s_queue_count, 1u)] = make_uint2(lr | (rr << 16), (uint32_t)(dl + 1) | ((uint32_t)(dr + 1) << 8));
        }
    }
    const uint32_t init_lo[3] = {order_key(merged.pmax.x), order_key(merged.pmax.y), order_key(merged.pmax.z)};
    const uint32_t init_hi[3] = {order_key(merged.pmin.x), order_key(merged.pmin.y), order_key(merged.pmin.z)};
    auto park = [&]() {
        // The node meets its sibling in the second pass, which starts after this kernel has ended, so that the memory
        // round trips of that exchange do not stall the lanes of this wave that still work in LDS.
        const uint32_t slot = atomicAdd(&s_park_count, 1u);
        if (slot < carrier_limit) { s_park[slot] = Carrier{b0 + lr, b0 + rr, global_id(me), 0u}; return; }
        // No room (more than kCarriersPerGroup nodes of one workgroup meet their sibling outside it: rare): the node
        // climbs through memory here and now.
        float lw = merged.pmax.w, hw = merged.pmin.w;
        uint32_t lo[3] = {e0.x, e0.y, e0.z}, hi[3] = {e0.w, e1.x, e1.y};
        if (me >= (uint32_t)BLOCK) {                        // a leaf: this lane's own
            lw = box_a.w; hw = box_b.w;
        } else {                                            // the stored box is the folded one
            for (int a = 0; a < 3; a++) { lo[a] = min(init_lo[a], lo[a]); hi[a] = max(init_hi[a], hi[a]); }
        }
        climb_global<PARENTS>(b0 + lr, b0 + rr, global_id(me), lo, hi, false, lw, hw, keys, bboxes, nodes, parents, slots, merged, n);
    };
    if (state == kLeaves) park();
    __syncthreads();
    if (t < s_queue_count) {
        const uint2 q = s_queue[t];
        lr = q.x & 0xffffu; rr = q.x >> 16;
        dl = (int)(q.y & 0xffu) - 1; dr = (int)(q.y >> 8) - 1;
        me = dr > dl ? rr : lr;
        e0 = s_node[me][0];
        e1 = s_node[me][1];
        for (;;) {
            state = meet();
            if (state != kGoesOn) break;
            publish();
        }
        if (state == kLeaves) park();
    }
    __syncthreads();
    {
        const uint4 b = s_node[t][1];
        if (b.w != 0u) {                                    // node b0 + t was finished here
            const uint4 a = s_node[t][0];
            const uint32_t idx = b0 + t, lcg = global_id(b.z), rcg = global_id(b.w);
            nodes[idx] = AccelNode{lcg, rcg};
            if (PARENTS) { parents[lcg] = idx; parents[rcg] = idx; }
            float4 *gb = reinterpret_cast<float4 *>(bboxes + idx);
            gb[0] = make_float4(key_to_float(min(init_lo[0], a.x)), key_to_float(min(init_lo[1], a.y)), key_to_float(min(init_lo[2], a.z)), merged.pmax.w);
            gb[1] = make_float4(key_to_float(max(init_hi[0], a.w)), key_to_float(max(init_hi[1], b.x)), key_to_float(max(init_hi[2], b.y)), merged.pmin.w);
        }
    }
    if constexpr (RECORDS) {
        const uint4 own = s_node[t][1];
        const bool finished = own.w != 0u;
        const unsigned long long served = __ballot(finished);
        if (lane == 0) local_mask[(b0 + t) >> 6] = served;
        if (finished) {
            // entry e as a traversal slot: box as floats (an internal node's folded with the box every internal box starts from), reference
            auto slot_of = [&](uint32_t e, float box[6]) -> uint32_t {
                const uint4 a = s_node[e][0], b = s_node[e][1];
                if (e < (uint32_t)BLOCK) {
                    box[0] = key_to_float(min(init_lo[0], a.x)); box[1] = key_to_float(min(init_lo[1], a.y)); box[2] = key_to_float(min(init_lo[2], a.z));
                    box[3] = key_to_float(max(init_hi[0], a.w)); box[4] = key_to_float(max(init_hi[1], b.x)); box[5] = key_to_float(max(init_hi[2], b.y));
                    return b0 + e;
                }
                box[0] = key_to_float(a.x); box[1] = key_to_float(a.y); box[2] = key_to_float(a.z);
                box[3] = key_to_float(a.w); box[4] = key_to_float(b.x); box[5] = key_to_float(b.y);
                const uint32_t leaf = e - BLOCK;
                return kLeafBit | (rec.top_level ? b.z : b0 + leaf);
            };
            float lbox[6], rbox[6];
            const uint32_t lref = slot_of(own.z, lbox), rref = slot_of(own.w, rbox);
            if (!six_in_fast_range(lbox) || !six_in_fast_range(rbox)) atomicOr(&rec.root->flags, 1u);
            if (rec.binary) store_half(reinterpret_cast<float4 *>(rec.binary + (b0 + t)), lbox, lref, rbox, rref);
            if (rec.wide) {
                struct {
                    decltype(slot_of) &slot_fn;
                    uint4 (*entries)[2];
                    __device__ bool internal(uint32_t e) const { return e < (uint32_t)BLOCK; }
                    __device__ void children(uint32_t e, uint32_t &l, uint32_t &r) const { const uint4 c = entries[e][1]; l = c.z; r = c.w; }
                    __device__ uint32_t slot(uint32_t e, float box[6]) const { return slot_fn(e, box); }
                } tree{slot_of, s_node};
                store_cut_record(reinterpret_cast<float4 *>(rec.wide + 2 * (size_t)(b0 + t)), tree, own.z, own.w);
            }
        }
    }
    const uint32_t parked = min(s_park_count, carrier_limit);
    if (t < parked) carriers[(size_t)blockIdx.x * kCarriersPerGroup + t] = s_park[t];
    if (t == 0) carrier_counts[blockIdx.x] = parked;
}

// Records of one internal node from the finished arrays in memory.  Binary: both children's boxes and references.
// Two-level: see store_cut_record.
__device__ inline void box6(const Bbox &b, float out[6]) {
    out[0] = b.pmin.x; out[1] = b.pmin.y; out[2] = b.pmin.z; out[3] = b.pmax.x; out[4] = b.pmax.y; out[5] = b.pmax.z;
}
__device__ inline void pack_from_memory(uint32_t i, const RecordOutputs &rec, const AccelNode *__restrict__ nodes, const Bbox *__restrict__ bboxes, uint32_t n) {
    const AccelNode nd = nodes[i];
    float lbox[6], rbox[6];
    box6(bboxes[nd.lc_or_id], lbox);
    box6(bboxes[nd.rc], rbox);
    if (rec.root && (!six_in_fast_range(lbox) || !six_in_fast_range(rbox))) atomicOr(&rec.root->flags, 1u);
    const uint32_t lref = child_ref(nd.lc_or_id, nodes, n, rec.top_level), rref = child_ref(nd.rc, nodes, n, rec.top_level);
    if (rec.binary) store_half(reinterpret_cast<float4 *>(rec.binary + i), lbox, lref, rbox, rref);
    if (!rec.wide) return;
    struct {
        const AccelNode *nodes; const Bbox *bboxes; uint32_t n; int top_level;
        __device__ bool internal(uint32_t id) const { return id < n - 1; }
        __device__ void children(uint32_t id, uint32_t &l, uint32_t &r) const { const AccelNode c = nodes[id]; l = c.lc_or_id; r = c.rc; }
        __device__ uint32_t slot(uint32_t id, float box[6]) const { box6(bboxes[id], box); return child_ref(id, nodes, n, top_level); }
    } tree{nodes, bboxes, n, rec.top_level};
    store_cut_record(reinterpret_cast<float4 *>(rec.wide + 2 * (size_t)i), tree, nd.lc_or_id, nd.rc);
}

// The root record of a BVH (when rec.root is given), and the records of the internal nodes the hierarchy pass did not
// serve (local_mask null: of all of them).  root->flags was zeroed before the build; lanes only ever OR bits into it.
__global__ void __launch_bounds__(kBlock) k_pack_rest(RecordOutputs rec, const AccelNode *__restrict__ nodes, const Bbox *__restrict__ bboxes, uint32_t n,
                                                       const unsigned long long *__restrict__ local_mask) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i == 0 && rec.root) {
        const Bbox b = bboxes[0];
        AccelRoot *root = rec.root;
        root->bmin[0] = b.pmin.x; root->bmin[1] = b.pmin.y; root->bmin[2] = b.pmin.z;
        root->bmax[0] = b.pmax.x; root->bmax[1] = b.pmax.y; root->bmax[2] = b.pmax.z;
        root->ref = n == 1 ? (kLeafBit | (rec.top_level ? nodes[0].lc_or_id : 0u)) : 0u;
        root->num_prims = n;
        if (!box_in_fast_range(b)) atomicOr(&root->flags, 1u);
    }
    if (n < 2 || i >= n - 1) return;
    if (local_mask && ((local_mask[i >> 6] >> (i & 63u)) & 1ull)) return;
    pack_from_memory(i, rec, nodes, bboxes, n);
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
    size_t tiles = cdiv(n, kBlock * kSmallItems);      // the smaller tile of the two sort configurations
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    return up(8 * (size_t)n) * 2               // keys ping/pong
         + up(32 * (size_t)n)                  // unsorted leaf boxes
         + up(sizeof(Carrier) * kCarriersPerGroup * cdiv(n, kUpBlock))   // carriers between the two hierarchy passes
         + up(4 * (size_t)cdiv(n, kUpBlock))   // their counts
         + up(4 * (size_t)n)                   // one exchange word per split
         + up(8 * (size_t)cdiv(n, kUpBlock) * (kUpBlock / 64))   // one bit per internal node (one word per wave of the first hierarchy pass): its records were written by that pass
         + up(4 * ((size_t)kPasses * kBins + 64 + (size_t)kPasses * tiles * kBins));   // digit histograms, tile tickets, look-back words
}

int BuildWorkspace::reserve(uint32_t n) {
    size_t need = bytes_for(n);
    if (need <= capacity) return 0;
    if (borrowed) return (int)hipErrorOutOfMemory;       // memory that is not this workspace's to grow
    if (base) { HIP_OK(hipFree(base)); base = nullptr; capacity = 0; }
    HIP_OK(hipMalloc(&base, need));
    capacity = need;
    return 0;
}

void BuildWorkspace::release() {
    if (base && !borrowed) (void)hipFree(base);
    base = nullptr;
    capacity = 0;
}

int lbvh_build(AccelNode *nodes, Bbox *bboxes, const Bbox &merged, uint32_t n, hipStream_t stream, BuildWorkspace *ws_in,
               const Bbox *leaf_src_in, uint64_t *keys_out, uint32_t *parents_out, float *stage_ms, const RecordOutputs *records) {
    if (n == 0) return 0;
    if (records && (!records->root || (!records->binary && !records->wide))) return (int)hipErrorInvalidValue;
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
    Carrier *carriers = reinterpret_cast<Carrier *>(p); p += up(sizeof(Carrier) * kCarriersPerGroup * cdiv(n, kUpBlock));
    uint32_t *carrier_counts = reinterpret_cast<uint32_t *>(p); p += up(4 * (size_t)cdiv(n, kUpBlock));
    uint32_t *slots = reinterpret_cast<uint32_t *>(p); p += up(4 * (size_t)n);
    unsigned long long *local_mask = reinterpret_cast<unsigned long long *>(p); p += up(8 * (size_t)cdiv(n, kUpBlock) * (kUpBlock / 64));
    uint32_t *digit_hist = reinterpret_cast<uint32_t *>(p);            // kPasses x kBins, then first positions
    uint32_t *tickets = digit_hist + kPasses * kBins;                  // one per pass
    uint32_t *tile_status = tickets + 64;                              // kPasses x tiles x kBins
    const size_t sort_ctl_bytes = 4 * ((size_t)kPasses * kBins + 64 + (size_t)kPasses * tiles * kBins);

    struct StageEvents {                        // destroyed on every way out of this function
        hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        ~StageEvents() { for (auto e : ev) if (e) (void)hipEventDestroy(e); }
    } stage_events;
    hipEvent_t *ev = stage_events.ev;
    if (stage_ms) for (int i = 0; i < 6; i++) HIP_OK(hipEventCreate(&ev[i]));
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
    uint64_t *src = keys_a, *dst = keys_b;
    const bool onesweep = ws->sort_variant == 1;
    bool slots_cleared = false;
    if (onesweep) {
        HIP_OK(hipMemsetAsync(digit_hist, 0, sort_ctl_bytes, stream));
        hipLaunchKernelGGL(k_morton_hist, dim3(grid_n < kMortonBlocks ? grid_n : kMortonBlocks), dim3(kBlock), 0, stream, keys_a, leaf_src, merged, n, digit_hist);
        mark(1);
        if (n > 1) {
            hipLaunchKernelGGL(k_digit_bases, dim3(kPasses), dim3(kBlock), 0, stream, digit_hist);
            for (int pass = 0; pass < kPasses; pass++) {
                hipLaunchKernelGGL(k_onesweep, dim3((uint32_t)(tiles < kSortBlocks ? tiles : kSortBlocks)), dim3(kBlock), 0, stream, src, dst, n,
                                   32 + pass * kRadixBits, digit_hist + pass * kBins, tile_status + (size_t)pass * tiles * kBins, tickets + pass, tickets + 8);
                uint64_t *t = src; src = dst; dst = t;
            }
        }
    } else {
        uint32_t *tile_hist = tile_status, *bin_total = digit_hist;     // same scratch, other use
        const bool small = n <= kSmallSort;
        const uint32_t nt = small ? cdiv(n, kBlock * kSmallItems) : (uint32_t)tiles;
        if (small) hipLaunchKernelGGL((k_morton_tile<kSmallItems, true>), dim3(nt), dim3(kBlock), 0, stream, keys_a, leaf_src, merged, n, tile_hist, nt, slots);
        else hipLaunchKernelGGL((k_morton_tile<1, false>), dim3(grid_n), dim3(kBlock), 0, stream, keys_a, leaf_src, merged, n, tile_hist, nt, slots);     // one key per lane
        slots_cleared = n > 1;       // small: by the key kernel; large: beside the scans of the sort passes below
        mark(1);
        if (n > 1) {
            for (int pass = 0; pass < kPasses; pass++) {
                int shift = 32 + pass * kRadixBits;
                if (pass == 0 && small) {}       // the key kernel counted this pass's digits
                else if (small) hipLaunchKernelGGL((k_hist<kSmallItems, kBlock>), dim3(nt), dim3(kBlock), 0, stream, src, n, shift, tile_hist, nt);
                else hipLaunchKernelGGL((k_hist<kItems / kHistWide, kBlock * kHistWide>), dim3(nt), dim3(kBlock * kHistWide), 0, stream, src, n, shift, tile_hist, nt);
                // large inputs: the exchange words (4 bytes per split, the allocation rounded up to 256) are zeroed beside the scans, a quarter per pass
                uint32_t clear_count = 0;
                uint4 *clear = nullptr;
                if (!small) {
                    const uint32_t words16 = (uint32_t)(((size_t)4 * n + 255) / 256 * 16), share = cdiv(words16, (uint32_t)kPasses);
                    const uint32_t begin = share * (uint32_t)pass;
                    clear_count = begin < words16 ? (words16 - begin < share ? words16 - begin : share) : 0;
                    clear = reinterpret_cast<uint4 *>(slots) + begin;
                }
                hipLaunchKernelGGL(k_scan_bins, dim3(kBins + cdiv(clear_count, (uint32_t)kBlock)), dim3(kBlock), 0, stream, tile_hist, nt, bin_total, clear, clear_count);
                if (small) hipLaunchKernelGGL((k_scatter<kSmallItems, false>), dim3(nt), dim3(kBlock), 0, stream, src, dst, n, shift, tile_hist, bin_total, nt);
                                else hipLaunchKernelGGL((k_scatter_wide<kTile / (kWideWaves * 64), kWideWaves, true>), dim3(nt), dim3(kWideWaves * 64), 0, stream, src, dst, n, shift, tile_hist, bin_total, nt);
                uint64_t *t = src; src = dst; dst = t;
            }
        }
    }
    mark(2);
    if (n > 1) {
        const bool large = n > kSmallSort;
        const uint32_t up_block = large ? kUpBlockLarge : kUpBlock;
        const uint32_t groups = cdiv(n, up_block);
        const uint32_t carrier_limit = ws->carrier_limit < kCarriersPerGroup ? ws->carrier_limit : kCarriersPerGroup;
        if (!slots_cleared) HIP_OK(hipMemsetAsync(slots, 0, 4 * (size_t)(n - 1), stream));      // one exchange word per split
        const RecordOutputs rec = records ? *records : RecordOutputs{};
        if (records) HIP_OK(hipMemsetAsync(rec.root, 0, sizeof(AccelRoot), stream));
        auto first_pass = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, dim3(groups), dim3(up_block), 0, stream, src, leaf_src, bboxes, nodes, parents_out, slots, carriers, carrier_counts,
                               carrier_limit, merged, n, rec, local_mask);
        };
        if (parents_out) {
            if (large) { if (records) first_pass(k_build_up<true, kUpBlockLarge, true>); else first_pass(k_build_up<true, kUpBlockLarge, false>); }
            else if (records) first_pass(k_build_up<true, kUpBlock, true>); else first_pass(k_build_up<true, kUpBlock, false>);
            mark(3);
            hipLaunchKernelGGL(k_climb<true>, dim3(cdiv(groups, kBlock / kCarriersPerGroup)), dim3(kBlock), 0, stream, carriers, carrier_counts, groups, src,
                               bboxes, nodes, parents_out, slots, merged, n);
        } else {
            if (large) { if (records) first_pass(k_build_up<false, kUpBlockLarge, true>); else first_pass(k_build_up<false, kUpBlockLarge, false>); }
            else if (records) first_pass(k_build_up<false, kUpBlock, true>); else first_pass(k_build_up<false, kUpBlock, false>);
            mark(3);
            hipLaunchKernelGGL(k_climb<false>, dim3(cdiv(groups, kBlock / kCarriersPerGroup)), dim3(kBlock), 0, stream, carriers, carrier_counts, groups, src,
                               bboxes, nodes, (uint32_t *)nullptr, slots, merged, n);
        }
        if (records) hipLaunchKernelGGL(k_pack_rest, dim3(cdiv(n - 1, kBlock)), dim3(kBlock), 0, stream, rec, nodes, bboxes, n, local_mask);
    } else {
        hipLaunchKernelGGL(k_gather_leaves, dim3(grid_n), dim3(kBlock), 0, stream, src, leaf_src, bboxes, nodes, n);
        mark(3);
        if (records) {
            HIP_OK(hipMemsetAsync(records->root, 0, sizeof(AccelRoot), stream));
            hipLaunchKernelGGL(k_pack_rest, dim3(1), dim3(kBlock), 0, stream, *records, nodes, bboxes, n, (const unsigned long long *)nullptr);
        }
    }
    mark(4);
    mark(5);
    int copy_status = keys_out ? (int)hipMemcpyAsync(keys_out, src, 8 * (size_t)n, hipMemcpyDeviceToDevice, stream) : 0;
    // The look-back sort reports a predecessor tile it gave up waiting for through a flag; a build that used it
    // always reads the flag back (and so waits for the build), a silently incomplete scatter is never returned.
    uint32_t sort_failed = 0;
    int status = copy_status ? copy_status : (int)hipGetLastError();
    if (status == 0 && (stage_ms || !ws_in || onesweep)) {
        if (onesweep) status = (int)hipMemcpyAsync(&sort_failed, tickets + 8, 4, hipMemcpyDeviceToHost, stream);
        if (status == 0) status = (int)hipStreamSynchronize(stream);
    }
    if (stage_ms) {
        for (int i = 0; i < 5 && status == 0; i++) status = (int)hipEventElapsedTime(&stage_ms[i], ev[i], ev[i + 1]);
    }
    if (!ws_in) local.release();                // the temporaries die with this call, as in the reference
    if (status) return status;
    return sort_failed ? (int)hipErrorLaunchFailure : 0;
}

// ---- the build with its scratch in a fixed device array: no allocation, ever ------------------------------
// Every GPU this library runs on holds its own copy of the array (at its own address) from the moment the code
// object is loaded until the process ends.  The name `scratch` is only a host-side handle for that symbol: the
// device address is asked for once per device and kept, and only that address goes to kernels and memsets.
constexpr size_t kStaticScratchBytes = 60u << 20;        // what 2^20 primitives need (56.2 MiB), rounded up
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

constexpr int kMaxDevices = 64;
static std::mutex g_static_mutex;
static void *g_static_ptr[kMaxDevices];                  // per device: address of that device's `scratch`

uint32_t lbvh_static_max_primitives() { return kStaticMaxPrimitives; }

int lbvh_build_static(AccelNode *nodes, Bbox *bboxes, const Bbox &merged, uint32_t n, hipStream_t stream) {
    if (n == 0) return 0;
    if (n > kStaticMaxPrimitives || BuildWorkspace::bytes_for(n) > kStaticScratchBytes) return (int)hipErrorInvalidValue;
    int device = -1;
    HIP_OK(hipGetDevice(&device));
    if (device < 0 || device >= kMaxDevices) return (int)hipErrorInvalidDevice;
    void *ptr = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_static_mutex);
        if (!g_static_ptr[device]) {
            void *p = nullptr;
            HIP_OK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));        // with `device` current: that device's copy
            if (!p) return (int)hipErrorInvalidSymbol;
            g_static_ptr[device] = p;
        }
        ptr = g_static_ptr[device];
    }
    BuildWorkspace ws;
    ws.base = ptr;
    ws.capacity = kStaticScratchBytes;
    ws.borrowed = true;
    // Everything the kernels expect to start at zero (exchange words, tile counts) is cleared or fully rewritten by
    // the build itself, as for any reused workspace, so the previous call's contents do not matter.
    int rc = lbvh_build(nodes, bboxes, merged, n, stream, &ws, nullptr, nullptr, nullptr, nullptr);
    if (rc) return rc;
    HIP_OK(hipStreamSynchronize(stream));
    return 0;
}

int lbvh_triangle_boxes(Bbox *leaf, const uint32_t *indices, const V3 *positions, uint32_t n, hipStream_t stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_tri_boxes, dim3(cdiv(n, kBlock)), dim3(kBlock), 0, stream, leaf, indices, positions, n);
    return (int)hipGetLastError();
}

int lbvh_pack_binary(PackedNode *out_binary, const AccelNode *nodes, const Bbox *bboxes, uint32_t n, bool top_level, hipStream_t stream) {
    if (n < 2) return 0;
    RecordOutputs rec;
    rec.binary = out_binary;
    rec.top_level = top_level ? 1 : 0;
    hipLaunchKernelGGL(k_pack_rest, dim3(cdiv(n - 1, kBlock)), dim3(kBlock), 0, stream, rec, nodes, bboxes, n, (const unsigned long long *)nullptr);
    return (int)hipGetLastError();
}

int lbvh_pack_triangles(TriRecord *out_tris, const AccelNode *nodes, const uint32_t *indices, const V3 *positions, uint32_t n, hipStream_t stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_pack_tris, dim3(cdiv(n, kBlock)), dim3(kBlock), 0, stream, out_tris, nodes, indices, positions, n);
    return (int)hipGetLastError();
}

}  // namespace cpt
