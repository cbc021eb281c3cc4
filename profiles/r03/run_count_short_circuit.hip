// Reconstruction (round 3) of the short-circuit form of run_count_kernel that DESIGN.md section 4.6 blames for the
// round-2 fault (HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION in tgx_substring_df).  The faulting source itself was
// never committed; this follows the description there ("short-circuit loads of entry i - 1").  Compile only:
//   hipcc -O3 --offload-arch=gfx950 -S --cuda-device-only -o run_count_short_circuit.s run_count_short_circuit.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void run_count_short_circuit(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ vals,
                                                              const uint32_t* __restrict__ run_id, uint64_t n,
                                                              uint64_t* __restrict__ rep, uint32_t* __restrict__ df) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = run_id[i] - 1u;
    const bool head = i == 0 || keys[i] != keys[i - 1];
    const bool newdoc = head || (vals[i] >> 37) != (vals[i - 1] >> 37);
    if (head) rep[r] = vals[i];
    if (newdoc) atomicAdd(&df[r], 1u);
}

// the same with the run index read AFTER the conditions (another plausible order of the lost source)
__global__ __launch_bounds__(256) void run_count_short_circuit_late_index(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ vals,
                                                                         const uint32_t* __restrict__ run_id, uint64_t n,
                                                                         uint64_t* __restrict__ rep, uint32_t* __restrict__ df) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const bool head = i == 0 || keys[i] != keys[i - 1];
    const bool newdoc = head || (vals[i] >> 37) != (vals[i - 1] >> 37);
    const uint32_t r = run_id[i] - 1u;
    if (head) rep[r] = vals[i];
    if (newdoc) atomicAdd(&df[r], 1u);
}

// with the run-count guard that the committed kernel has
__global__ __launch_bounds__(256) void run_count_short_circuit_guarded(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ vals,
                                                                      const uint32_t* __restrict__ run_id, uint64_t n, uint32_t n_runs,
                                                                      uint64_t* __restrict__ rep, uint32_t* __restrict__ df) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = run_id[i] - 1u;
    const bool head = i == 0 || keys[i] != keys[i - 1];
    const bool newdoc = head || (vals[i] >> 37) != (vals[i - 1] >> 37);
    if (r >= n_runs) return;
    if (head) rep[r] = vals[i];
    if (newdoc) atomicAdd(&df[r], 1u);
}

// a 64-bit run index (size_t r), which keeps the index in a register PAIR as section 4.6 describes
__global__ __launch_bounds__(256) void run_count_short_circuit_index64(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ vals,
                                                                      const uint32_t* __restrict__ run_id, uint64_t n,
                                                                      uint64_t* __restrict__ rep, uint32_t* __restrict__ df) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint64_t r = (uint64_t)run_id[i] - 1u;
    const bool head = i == 0 || keys[i] != keys[i - 1];
    if (head) rep[r] = vals[i];
    if (head || (vals[i] >> 37) != (vals[i - 1] >> 37)) atomicAdd(&df[r], 1u);
}
