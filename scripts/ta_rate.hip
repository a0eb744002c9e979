// Development aid: how many cycles the address unit of a CU's vector memory pipeline spends on one wave64
// global_load_dwordx4, by where the lanes' 16 bytes lie.  Every wave of the device reads the same 8 KB (L1 hits after the
// first pass), eight independent loads per loop trip, four workgroups of four waves per CU.
//   hipcc -O3 --offload-arch=gfx950 scripts/ta_rate.hip -o scripts/build/ta_rate && scripts/build/ta_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef const f4 __attribute__((address_space(1))) *GlobalF4;

// pattern -> byte offset of the lane's 16 bytes inside the 8 KB window, for load k of trip it
__device__ inline uint32_t lane_offset(int pattern, uint32_t lane, uint32_t it, uint32_t k) {
    const uint32_t turn = (it * 8u + k) * 5u;
    switch (pattern) {
        case 0: return (((lane + turn) & 63u) * 128u) + ((turn >> 3) & 7u) * 16u;                         // 64 lanes, 64 lines
        case 1: return ((((lane >> 1) + turn) & 63u) * 128u) + (lane & 1u) * 64u;                          // pairs: one line, its two halves
        case 2: return ((((lane >> 1) + turn) & 63u) * 128u) + (lane & 1u) * 16u;                          // pairs: 32 contiguous bytes
        case 3: return ((((lane >> 2) + turn) & 63u) * 128u) + (lane & 3u) * 16u;                          // quads: 64 contiguous bytes
        case 4: return ((((lane >> 3) + turn) & 63u) * 128u) + (lane & 7u) * 16u;                          // octets: one whole line
        case 5: return ((lane * 16u + (it + k * 3u) * 1024u) & 8191u);                                               // the wave: 1 KB contiguous
        default: return (((lane + turn) & 63u) * 128u);                                                     // 6, 7: as 0 with some lanes off
    }
}

template <int PATTERN>
__global__ void __launch_bounds__(256) k_ta(const float *window, uint32_t trips, float *sink) {
    const uint32_t lane = threadIdx.x & 63u;
    GlobalF4 base = (GlobalF4)window;
    f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    const bool on = PATTERN == 6 ? (lane * 13u % 64u) < 26u : PATTERN == 7 ? lane < 26u : true;           // 6: 26 lanes spread over the wave; 7: the first 26
    if (on) {
        for (uint32_t it = 0; it < trips; it++) {
            f4 v[8];
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) v[k] = base[lane_offset(PATTERN, lane, it, k) >> 4];
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) acc += v[k];
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[threadIdx.x] = acc.x;
}

int main() {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float *window, *sink;
    hipMalloc(&window, 8192);
    hipMalloc(&sink, 1024);
    hipMemset(window, 0, 8192);
    const uint32_t trips = 20000;
    const char *names[8] = {"64 lanes in 64 lines", "pairs in the two halves of a line", "pairs in 32 contiguous bytes", "quads in 64 contiguous bytes",
                            "octets in one 128-byte line", "the wave in 1 KB contiguous", "26 of 64 lanes (spread), each in a line of its own", "26 of 64 lanes (the first), each in a line of its own"};
    void (*kernels[8])(const float *, uint32_t, float *) = {k_ta<0>, k_ta<1>, k_ta<2>, k_ta<3>, k_ta<4>, k_ta<5>, k_ta<6>, k_ta<7>};
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    printf("{\"compute_units\": %d, \"waves_per_cu\": 16, \"loads_per_wave\": %u, \"clock_mhz_assumed\": 2400, \"patterns\": {", cus, trips * 8);
    for (int p = 0; p < 8; p++) {
        hipLaunchKernelGGL(kernels[p], dim3(cus * 4), dim3(256), 0, 0, window, 200u, sink);
        hipDeviceSynchronize();
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(kernels[p], dim3(cus * 4), dim3(256), 0, 0, window, trips, sink);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0.0f;
        hipEventElapsedTime(&ms, a, b);
        const double wave_loads_per_cu = 16.0 * trips * 8.0;
        const double cycles = ms * 1e-3 * 2.4e9 / wave_loads_per_cu;
        printf("%s\"%s\": {\"ms\": %.3f, \"cu_cycles_per_wave_load\": %.2f}", p ? ", " : "", names[p], ms, cycles);
    }
    printf("}}\n");
    return 0;
}
