// Microbenchmark 2 (round 2): what a trie step costs by record width (4 / 8 / 16 bytes per lane), by where
// the table lives (vector L1/L2 by table size, or LDS) and by the number of independent chains per lane.
// Every lane runs CH independent dependent chains of random gathers (like trie walks); 13/16 of the
// gathers go to a hot prefix of the table (the real trie: 4096 slots take 81 %).
//   hipcc --offload-arch=gfx950 -O3 -o gather2_bench gather2_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int WIDTH, int CH>
__global__ __launch_bounds__(256) void gather(const uint32_t* __restrict__ tab, uint32_t mask_slots, int steps, int active,
                                              uint32_t hot_mask, uint32_t* out) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t s[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) s[c] = (blockIdx.x * blockDim.x + threadIdx.x + 977u * c) * 2654435761u;
    uint32_t acc = 0;
    if ((int)lane < active) {
        for (int i = 0; i < steps; ++i) {
            uint32_t r[CH], idx[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                r[c] = s[c] * 1664525u + 1013904223u;
                idx[c] = (r[c] >> 4) & (((r[c] >> 28) < 13u) ? hot_mask : mask_slots);
            }
            if (WIDTH == 16) {
                uint4 v[CH];
#pragma unroll
                for (int c = 0; c < CH; ++c) v[c] = reinterpret_cast<const uint4*>(tab)[idx[c]];
#pragma unroll
                for (int c = 0; c < CH; ++c) { s[c] = v[c].x ^ r[c]; acc += v[c].w; }
            } else if (WIDTH == 8) {
                uint2 v[CH];
#pragma unroll
                for (int c = 0; c < CH; ++c) v[c] = reinterpret_cast<const uint2*>(tab)[idx[c]];
#pragma unroll
                for (int c = 0; c < CH; ++c) { s[c] = v[c].x ^ r[c]; acc += v[c].y; }
            } else {
                uint32_t v[CH];
#pragma unroll
                for (int c = 0; c < CH; ++c) v[c] = tab[idx[c]];
#pragma unroll
                for (int c = 0; c < CH; ++c) { s[c] = v[c] ^ r[c]; acc += v[c]; }
            }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// the same chains over a table copied into LDS (4-byte records, lds_slots a power of two)
template <int CH>
__global__ __launch_bounds__(1024) void gather_lds(const uint32_t* __restrict__ tab, uint32_t lds_slots, int steps, int active,
                                                   uint32_t hot_mask, uint32_t* out) {
    extern __shared__ uint32_t lt[];
    for (uint32_t i = threadIdx.x; i < lds_slots; i += blockDim.x) lt[i] = tab[i];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t mask_slots = lds_slots - 1u;
    uint32_t s[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) s[c] = (blockIdx.x * blockDim.x + threadIdx.x + 977u * c) * 2654435761u;
    uint32_t acc = 0;
    if ((int)lane < active) {
        for (int i = 0; i < steps; ++i) {
            uint32_t r[CH], v[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                r[c] = s[c] * 1664525u + 1013904223u;
                v[c] = lt[(r[c] >> 4) & (((r[c] >> 28) < 13u) ? hot_mask : mask_slots)];
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) { s[c] = v[c] ^ r[c]; acc += v[c]; }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// rows: the 16 lanes of a row read 8 bytes each of ONE random 128-byte line (a dense per-node score row)
__global__ __launch_bounds__(256) void gather_rows(const double* __restrict__ tab, uint32_t mask_lines, int steps, uint32_t hot_mask,
                                                   uint32_t* out) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t s = (blockIdx.x * blockDim.x + (threadIdx.x & ~15u)) * 2654435761u;
    double acc = 0;
    for (int i = 0; i < steps; ++i) {
        const uint32_t r = s * 1664525u + 1013904223u;
        const uint32_t line = (r >> 4) & (((r >> 28) < 13u) ? hot_mask : mask_lines);
        const double v = tab[(size_t)line * 16u + (lane & 15u)];
        acc += v;
        s = r ^ (uint32_t)__double2uint_rz(v);
    }
    if (acc == 1.2345) out[0] = 1;
}

int main() {
    const uint32_t max_bytes = 16u << 20;
    std::vector<uint32_t> h(max_bytes / 4);
    for (uint32_t i = 0; i < h.size(); ++i) h[i] = i * 747796405u + 2891336453u;
    uint32_t* d; uint32_t* out;
    CK(hipMalloc(&d, max_bytes)); CK(hipMalloc(&out, 4));
    CK(hipMemcpy(d, h.data(), max_bytes, hipMemcpyHostToDevice));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int steps = 2000;
    auto report = [&](const char* what, int width, uint32_t slots, uint32_t hot, int wpc, int ch, int active, float best) {
        const double wave_loads_per_cu = (double)wpc * steps * ch;
        const double cyc = best * 1e-3 * 2.4e9;
        printf("%-6s width %2d slots %7u (%5u KiB) hot %5u waves/CU %2d chains %d active %2d: %7.3f ms  %6.1f cyc/wave-load/CU  %5.2f lanes/cyc/CU  step latency %5.0f cyc\n",
               what, width, slots, slots * (uint32_t)width / 1024u, hot, wpc, ch, active, best, cyc / wave_loads_per_cu,
               active * wave_loads_per_cu / cyc, cyc / steps);
        fflush(stdout);
    };
#define RUN(expr)                                                                   \
    ([&]() {                                                                        \
        float best = 1e9f;                                                          \
        for (int rep = 0; rep < 3; ++rep) {                                         \
            CK(hipEventRecord(a)); expr; CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); \
            float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; \
        }                                                                           \
        return best;                                                                \
    })()
    // (1) global tables: width x table size x waves x chains
    for (int width : {4, 8, 16})
        for (uint32_t slots : {16384u, 65536u, 262144u})   // 4-byte: 64 KiB .. 1 MiB; 16-byte: 256 KiB .. 4 MiB
            for (int wpc : {16, 32})
                for (int ch : {1, 2, 4})
                    for (int active : {64, 24}) {
                        const int blocks = 256 * wpc / 4;
                        const uint32_t hot = 4095u;
                        float ms = 0;
                        if (width == 4) {
                            if (ch == 1) ms = RUN((gather<4, 1><<<blocks, 256>>>(d, slots - 1, steps, active, hot, out)));
                            if (ch == 2) ms = RUN((gather<4, 2><<<blocks, 256>>>(d, slots - 1, steps, active, hot, out)));
                            if (ch == 4) ms = RUN((gather<4, 4><<<blocks, 256>>>(d, slots - 1, steps, active, hot, out)));
                        } else if (width == 8) {
                            if (ch == 1) ms = RUN((gather<8, 1><<<blocks, 256>>>(d, slots - 1, steps, active, hot, out)));
                            if (ch == 2) ms = RUN((gather<8, 2><<<blocks, 256>>>(d, slots - 1, steps, active, hot, out)));
                            if (ch == 4) ms = RUN((gather<8, 4><<<blocks, 256>>>(d, slots - 1, steps, active, hot, out)));
                        } else {
                            if (ch == 1) ms = RUN((gather<16, 1><<<blocks, 256>>>(d, slots - 1, steps, active, hot, out)));
                            if (ch == 2) ms = RUN((gather<16, 2><<<blocks, 256>>>(d, slots - 1, steps, active, hot, out)));
                            if (ch == 4) ms = RUN((gather<16, 4><<<blocks, 256>>>(d, slots - 1, steps, active, hot, out)));
                        }
                        report("global", width, slots, hot + 1, wpc, ch, active, ms);
                    }
    // (2) LDS table of 4-byte records: 64 KiB and 128 KiB per block of 16 waves, one block per CU
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gather_lds<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gather_lds<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gather_lds<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (uint32_t slots : {16384u, 32768u})
        for (int ch : {1, 2, 4})
            for (int active : {64, 24}) {
                float ms = 0;
                if (ch == 1) ms = RUN((gather_lds<1><<<256, 1024, slots * 4>>>(d, slots, steps, active, 4095u, out)));
                if (ch == 2) ms = RUN((gather_lds<2><<<256, 1024, slots * 4>>>(d, slots, steps, active, 4095u, out)));
                if (ch == 4) ms = RUN((gather_lds<4><<<256, 1024, slots * 4>>>(d, slots, steps, active, 4095u, out)));
                report("lds", 4, slots, 4096, 16, ch, active, ms);
            }
    // (3) row gathers: one 128-byte line per 16-lane row
    for (uint32_t lines : {4096u, 32768u, 131072u})
        for (int wpc : {16, 32}) {
            const int blocks = 256 * wpc / 4;
            const float ms = RUN((gather_rows<<<blocks, 256>>>(reinterpret_cast<const double*>(d), lines - 1, steps, 255u, out)));
            report("rows", 8, lines * 16, 256 * 16, wpc, 1, 64, ms);
        }
    return 0;
}
