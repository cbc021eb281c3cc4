// Microbenchmark: cost of divergent gathers from an L2-resident table as a function of the number of
// active lanes and the access width — decides whether the trie walk is bound per load instruction or per
// lane.  Each wave runs CHAINS independent dependent-chains of random gathers (like trie steps).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int WIDTH>  // bytes per lane: 8 or 16
__global__ __launch_bounds__(256) void gather(const uint4* __restrict__ tab, uint32_t mask_slots, int steps, int active,
                                              uint32_t hot_slots, uint32_t* out) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u;
    uint32_t acc = 0;
    if ((int)lane < active) {
        for (int i = 0; i < steps; ++i) {
            // 80 % of the gathers go to the hot region like the real trie
            const uint32_t r = s * 1664525u + 1013904223u;
            const uint32_t idx = ((r >> 28) < 13u) ? (r >> 4) % hot_slots : (r >> 4) & mask_slots;
            if (WIDTH == 16) {
                const uint4 v = tab[idx];
                s = v.x ^ r;  // dependent chain
                acc += v.w;
            } else {
                const uint2 v = reinterpret_cast<const uint2*>(tab)[idx];
                s = v.x ^ r;
                acc += v.y;
            }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const uint32_t slots = 1u << 16;  // 1 MiB of 16-B records (L2 resident)
    std::vector<uint4> h(slots);
    for (uint32_t i = 0; i < slots; ++i) h[i] = make_uint4(i * 747796405u + 2891336453u, i, i * 3u, i * 7u);
    uint4* d; uint32_t* out;
    CK(hipMalloc(&d, slots * sizeof(uint4))); CK(hipMalloc(&out, 4));
    CK(hipMemcpy(d, h.data(), slots * sizeof(uint4), hipMemcpyHostToDevice));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int steps = 2000;
    for (int width : {16, 8})
        for (int wpc : {8, 16, 20, 32})           // waves per CU
            for (int active : {64, 32, 16, 4}) {
                const int blocks = 256 * wpc / 4;
                float best = 1e9;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipEventRecord(a));
                    if (width == 16) gather<16><<<blocks, 256>>>(d, slots - 1, steps, active, 2048, out);
                    else gather<8><<<blocks, 256>>>(d, slots - 1, steps, active, 2048, out);
                    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                    float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
                }
                const double wave_loads_per_cu = (double)wpc * steps;
                const double cyc = best * 1e-3 * 2.4e9;
                printf("width %2d waves/CU %2d active %2d: %7.3f ms  %6.1f cyc per wave-load per CU  %5.2f lanes/cyc/CU  step latency %6.0f cyc\n",
                       width, wpc, active, best, cyc / wave_loads_per_cu, active * wave_loads_per_cu / cyc, cyc / steps);
            }
    return 0;
}
