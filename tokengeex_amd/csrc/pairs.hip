// Pair scan of `merge`: adjacent token-id pairs inside each sample.
//
// Replaces the per-chunk FnvHashMap<(u32,u32),usize> + RwLock merge of the reference
// (src/merge.rs:53-76) with a deterministic device pipeline over the ids produced by
// the encode kernel: one u64 key (a << 32 | b) per token slot (the last token of a
// sample gets a sentinel), radix sort (rocPRIM), run-length encode.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdint.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_run_length_encode.hpp>

#include "kernels.h"

namespace tgx {

// keys[o + i] = (ids[o+i] << shift) | ids[o+i+1] for i < cnt - 1 (merge.rs:60-63), `sentinel` (larger than any
// pair) for i = cnt - 1.  shift = bits of the largest id: the radix sort then runs over 2 shift + 1 bits instead
// of 64 (31 bits, four passes instead of eight, for a 32 000-entry vocabulary).
__global__ __launch_bounds__(256) void pair_keys_kernel(const uint32_t* __restrict__ ids,
                                                        const uint64_t* __restrict__ out_offs,
                                                        uint64_t n_samples, uint32_t shift,
                                                        unsigned long long sentinel,
                                                        unsigned long long* __restrict__ keys) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t n_waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    const uint64_t wave_id = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    for (uint64_t s = wave_id; s < n_samples; s += n_waves) {
        const uint64_t o = out_offs[s];
        const uint64_t cnt = out_offs[s + 1] - o;
        for (uint64_t i = lane; i < cnt; i += 64) {
            const unsigned long long a = ids[o + i];
            keys[o + i] = (i + 1 < cnt) ? ((a << shift) | (unsigned long long)ids[o + i + 1]) : sentinel;
        }
    }
}

hipError_t launch_pair_keys(const uint32_t* ids, const uint64_t* out_offs, uint64_t n_samples, uint32_t shift,
                            unsigned long long sentinel, unsigned long long* keys, uint32_t blocks, hipStream_t stream) {
    hipLaunchKernelGGL(pair_keys_kernel, dim3(blocks), dim3(256), 0, stream, ids, out_offs, n_samples, shift, sentinel, keys);
    return hipGetLastError();
}

hipError_t pair_sort_temp_bytes(uint64_t n, size_t* bytes) {
    unsigned long long* p = nullptr;
    size_t b = 0;
    hipError_t e = rocprim::radix_sort_keys(nullptr, b, p, p, (size_t)n);
    *bytes = b;
    return e;
}
hipError_t pair_sort(void* temp, size_t temp_bytes, const unsigned long long* in, unsigned long long* out,
                     uint64_t n, unsigned int end_bit, hipStream_t stream) {
    return rocprim::radix_sort_keys(temp, temp_bytes, in, out, (size_t)n, 0, end_bit, stream);
}
hipError_t pair_rle_temp_bytes(uint64_t n, size_t* bytes) {
    unsigned long long* p = nullptr;
    unsigned int* c = nullptr;
    size_t b = 0;
    hipError_t e = rocprim::run_length_encode(nullptr, b, p, (unsigned int)n, p, c, c);
    *bytes = b;
    return e;
}
hipError_t pair_rle(void* temp, size_t temp_bytes, const unsigned long long* sorted, uint64_t n,
                    unsigned long long* unique_out, unsigned int* counts_out, unsigned int* n_runs_out,
                    hipStream_t stream) {
    return rocprim::run_length_encode(temp, temp_bytes, sorted, (unsigned int)n, unique_out, counts_out,
                                      n_runs_out, stream);
}

// ---- token histogram (frequency pass, src/prune.rs:205-244) by sort + run-length encode:
// global atomics on a Zipf-distributed id stream serialise on the hot tokens.
// (unique key, count) table ordered by count descending; the sort is stable, so equal counts keep the
// ascending key order of the input
hipError_t pair_count_sort_temp_bytes(uint64_t n, size_t* bytes) {
    unsigned int* c = nullptr;
    unsigned long long* k = nullptr;
    size_t b = 0;
    hipError_t e = rocprim::radix_sort_pairs_desc(nullptr, b, c, c, k, k, (size_t)n);
    *bytes = b;
    return e;
}
hipError_t pair_count_sort(void* temp, size_t temp_bytes, const unsigned int* counts_in, unsigned int* counts_out,
                           const unsigned long long* keys_in, unsigned long long* keys_out, uint64_t n,
                           hipStream_t stream) {
    return rocprim::radix_sort_pairs_desc(temp, temp_bytes, counts_in, counts_out, keys_in, keys_out, (size_t)n, 0, 32,
                                          stream);
}

// (sorted key, count) -> the ABI's (a << 32 | b, count as u64): the whole table leaves the device in its final form
__global__ __launch_bounds__(256) void pair_expand_kernel(const unsigned long long* __restrict__ keys, const unsigned int* __restrict__ cnt,
                                                          uint64_t n, uint32_t shift, unsigned long long* __restrict__ out_keys,
                                                          unsigned long long* __restrict__ out_counts) {
    const unsigned long long low = shift < 32 ? (1ull << shift) - 1ull : 0xFFFFFFFFull;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long k = keys[i];
        out_keys[i] = ((k >> shift) << 32) | (k & low);
        out_counts[i] = cnt[i];
    }
}
hipError_t launch_pair_expand(const unsigned long long* keys, const unsigned int* cnt, uint64_t n, uint32_t shift,
                              unsigned long long* out_keys, unsigned long long* out_counts, hipStream_t stream) {
    const uint32_t blocks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((n + 255) / 256, 4096));
    hipLaunchKernelGGL(pair_expand_kernel, dim3(blocks), dim3(256), 0, stream, keys, cnt, n, shift, out_keys, out_counts);
    return hipGetLastError();
}

// ---- frequency pass of a vocabulary that fits LDS: one private histogram per block -------------------------
// u32 counters for every id in the block's LDS (at most kHistMaxVocab ids: 144 KiB), ds_add per id, then one
// global u64 add per non-zero counter and block.  The ids are read once (4.6 GB/s per CU is all it takes); the
// radix sort + run-length encode it replaces moved them four times (3.9 ms per GiB of text against 0.4).
__global__ __launch_bounds__(1024) void ids_histogram_kernel(const uint32_t* __restrict__ ids, uint64_t n, uint32_t vocab,
                                                             unsigned long long* __restrict__ out) {
    extern __shared__ uint32_t hist[];
    for (uint32_t i = threadIdx.x; i < vocab; i += blockDim.x) hist[i] = 0u;
    __syncthreads();
    const uint64_t n4 = n / 4u;
    const uint4* __restrict__ ids4 = reinterpret_cast<const uint4*>(ids);  // pool buffers are 256-byte aligned
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 v = ids4[i];
        if (v.x < vocab) atomicAdd(&hist[v.x], 1u);
        if (v.y < vocab) atomicAdd(&hist[v.y], 1u);
        if (v.z < vocab) atomicAdd(&hist[v.z], 1u);
        if (v.w < vocab) atomicAdd(&hist[v.w], 1u);
    }
    if (blockIdx.x == 0 && threadIdx.x < (uint32_t)(n - 4u * n4)) {
        const uint32_t v = ids[4u * n4 + threadIdx.x];
        if (v < vocab) atomicAdd(&hist[v], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < vocab; i += blockDim.x) {
        const uint32_t c = hist[i];
        if (c) atomicAdd(&out[i], (unsigned long long)c);
    }
}
hipError_t launch_ids_histogram(const uint32_t* ids, uint64_t n, uint32_t vocab, unsigned long long* out, uint32_t blocks,
                                hipStream_t stream) {
    if (vocab > kHistMaxVocab) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ids_histogram_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(ids_histogram_kernel, dim3(blocks), dim3(1024), vocab * 4u, stream, ids, n, vocab, out);
    return hipGetLastError();
}

hipError_t ids_sort_temp_bytes(uint64_t n, size_t* bytes) {
    uint32_t* p = nullptr;
    size_t b = 0;
    hipError_t e = rocprim::radix_sort_keys(nullptr, b, p, p, (size_t)n);
    *bytes = b;
    return e;
}
hipError_t ids_sort(void* temp, size_t temp_bytes, const uint32_t* in, uint32_t* out, uint64_t n,
                    unsigned int end_bit, hipStream_t stream) {
    return rocprim::radix_sort_keys(temp, temp_bytes, in, out, (size_t)n, 0, end_bit, stream);
}
hipError_t ids_rle_temp_bytes(uint64_t n, size_t* bytes) {
    uint32_t* p = nullptr;
    unsigned int* c = nullptr;
    size_t b = 0;
    hipError_t e = rocprim::run_length_encode(nullptr, b, p, (unsigned int)n, p, c, c);
    *bytes = b;
    return e;
}
hipError_t ids_rle(void* temp, size_t temp_bytes, const uint32_t* sorted, uint64_t n, uint32_t* unique_out,
                   unsigned int* counts_out, unsigned int* n_runs_out, hipStream_t stream) {
    return rocprim::run_length_encode(temp, temp_bytes, sorted, (unsigned int)n, unique_out, counts_out,
                                      n_runs_out, stream);
}

}  // namespace tgx
