// Microbenchmark 3 (round 3): what does the texture path (TA) charge for the lanes of a gather that do NOT walk?
// encode5_kernel's walks die at different depths; a dead lane keeps issuing its gather from slot 0 (all dead lanes
// the same address) so that every load is issued on every path (in-order vmcnt).  Modes, for A walking lanes out of 64
// doing dependent random 8-byte gathers from an L2-resident table:
//   off   : the other lanes are EXEC-masked off
//   slot0 : the other lanes are active and load record 0
//   self  : the other lanes are active and re-load the record they loaded last (divergent, L1-resident)
// plus the cost of a gather instruction in which 0 / 1 / 2 / 8 / 64 lanes read, beside the walking wave's own loads.
//   hipcc --offload-arch=gfx950 -O3 -o gather3_bench gather3_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE, int CH>
__global__ __launch_bounds__(1024) void gather(const uint2* __restrict__ tab, uint32_t mask_slots, int steps, int active, uint32_t hot_mask, uint32_t* out) {
    const uint32_t lane = threadIdx.x & 63u;
    // the walking lanes are spread over the wave (every 64/active-th lane), as dying walks are
    const bool walker = ((lane * (uint32_t)active) >> 6) != (((lane + 63u) & 63u) * (uint32_t)active >> 6) || (lane == 0 && active > 0);
    uint32_t s[CH], last[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) { s[c] = (blockIdx.x * blockDim.x + threadIdx.x + 977u * c) * 2654435761u; last[c] = 0; }
    uint32_t acc = 0;
    for (int i = 0; i < steps; ++i) {
        uint32_t r[CH], idx[CH];
        uint2 v[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            r[c] = s[c] * 1664525u + 1013904223u;
            idx[c] = (r[c] >> 4) & (((r[c] >> 28) < 13u) ? hot_mask : mask_slots);
            if (MODE == 1) idx[c] = walker ? idx[c] : 0u;
            if (MODE == 2) idx[c] = walker ? idx[c] : last[c];
        }
        if (MODE == 0) {
            if (walker) {
#pragma unroll
                for (int c = 0; c < CH; ++c) v[c] = tab[idx[c]];
            } else {
#pragma unroll
                for (int c = 0; c < CH; ++c) v[c] = make_uint2(0, 0);
            }
        } else {
#pragma unroll
            for (int c = 0; c < CH; ++c) v[c] = tab[idx[c]];
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) { s[c] = v[c].x ^ r[c]; acc += v[c].y; if (walker) last[c] = idx[c]; }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// a full wave of dependent gathers (64 walkers, CH chains) plus EXTRA gather instructions per step in which only `few`
// lanes read anything.  MODE 0: buffer loads with the other lanes' offsets OUT OF RANGE (all lanes EXEC-active; a raw
// buffer load returns 0 for an offset beyond num_records without a memory access); MODE 1: the other lanes EXEC-masked
// off by a branch.  (A first version issued the loads from inline asm and faulted: the compiler, which does not know
// that an asm load's destination registers stay pending, reused a dead half of one as an address temporary.)
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
template <int CH, int EXTRA, int MODE>
__global__ __launch_bounds__(1024) void gather_extra(const uint2* __restrict__ tab, uint32_t bytes, uint32_t mask_slots, int steps, int few, uint32_t hot_mask, uint32_t* out) {
    const uint32_t lane = threadIdx.x & 63u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint2*>(tab), 0, (int)bytes, 0x00020000);
    uint32_t s[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) s[c] = (blockIdx.x * blockDim.x + threadIdx.x + 977u * c) * 2654435761u;
    uint32_t acc = 0;
    const bool mine = (int)lane >= 5 && (int)lane < 5 + few;
    for (int i = 0; i < steps; ++i) {
        uint32_t r[CH];
        u32x2_t v[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            r[c] = s[c] * 1664525u + 1013904223u;
            v[c] = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(((r[c] >> 4) & (((r[c] >> 28) < 13u) ? hot_mask : mask_slots)) * 8u), 0, 0);
        }
        u32x2_t x[EXTRA > 0 ? EXTRA : 1];
#pragma unroll
        for (int e = 0; e < EXTRA; ++e) {
            const uint32_t off = (((r[0] >> 7) + 131u * e) & mask_slots) * 8u;
            x[e] = u32x2_t{0u, 0u};
            if (MODE == 0) x[e] = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(mine ? off : 0xFFFFFFF0u), 0, 0);
            else if (mine) x[e] = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < EXTRA; ++e) acc += x[e].y ^ x[e].x;
#pragma unroll
        for (int c = 0; c < CH; ++c) { s[c] = v[c].x ^ r[c]; acc += v[c].y; }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const uint32_t slots = 65536;  // 512 KiB of 8-byte records: L2-resident
    std::vector<uint32_t> h(slots * 2);
    for (uint32_t i = 0; i < h.size(); ++i) h[i] = i * 747796405u + 2891336453u;
    uint2* d; uint32_t* out;
    CK(hipMalloc(&d, slots * 8)); CK(hipMalloc(&out, 4));
    CK(hipMemcpy(d, h.data(), slots * 8, hipMemcpyHostToDevice));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int steps = 2000, wpc = 16, blocks = 256;
#define RUN(expr)                                                                   \
    ([&]() {                                                                        \
        float best = 1e9f;                                                          \
        for (int rep = 0; rep < 3; ++rep) {                                         \
            CK(hipEventRecord(a)); expr; CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); \
            float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; \
        }                                                                           \
        return best;                                                                \
    })()
    const char* names[3] = {"off", "slot0", "self"};
    for (int active : {64, 32, 16, 8}) {
        for (int mode = 0; mode < 3; ++mode) {
            float ms = 0;
            if (mode == 0) ms = RUN((gather<0, 4><<<blocks, 1024>>>(d, slots - 1, steps, active, 4095u, out)));
            if (mode == 1) ms = RUN((gather<1, 4><<<blocks, 1024>>>(d, slots - 1, steps, active, 4095u, out)));
            if (mode == 2) ms = RUN((gather<2, 4><<<blocks, 1024>>>(d, slots - 1, steps, active, 4095u, out)));
            const double cyc = ms * 1e-3 * 2.4e9, wl = (double)wpc * steps * 4;
            printf("walkers %2d of 64, others %-5s: %7.3f ms  %6.1f cyc per wave-load and CU  (16 waves/CU, 4 chains per lane)\n", active, names[mode], ms, cyc / wl);
            fflush(stdout);
        }
    }
    for (int mode = 0; mode < 2; ++mode)
        for (int few : {0, 1, 2, 8, 64}) {
            float base, with;
            if (mode == 0) {
                base = RUN((gather_extra<4, 0, 0><<<blocks, 1024>>>(d, slots * 8, slots - 1, steps, few, 4095u, out)));
                with = RUN((gather_extra<4, 4, 0><<<blocks, 1024>>>(d, slots * 8, slots - 1, steps, few, 4095u, out)));
            } else {
                base = RUN((gather_extra<4, 0, 1><<<blocks, 1024>>>(d, slots * 8, slots - 1, steps, few, 4095u, out)));
                with = RUN((gather_extra<4, 4, 1><<<blocks, 1024>>>(d, slots * 8, slots - 1, steps, few, 4095u, out)));
            }
            const double cyc = (with - base) * 1e-3 * 2.4e9, wl = (double)wpc * steps * 4;
            printf("extra buffer load with %2d lanes reading (%s) beside 4 full gathers: +%6.1f cyc per extra wave-load and CU (base %.3f ms, with %.3f ms)\n",
                   few, mode == 0 ? "others out of range" : "others EXEC-masked ", cyc / wl, base, with);
            fflush(stdout);
        }
    return 0;
}
