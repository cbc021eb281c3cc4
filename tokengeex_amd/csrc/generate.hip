// Device side of `generate` (reference src/generate.rs:54-139, VocabularyGenerator::feed): the DOCUMENT
// frequency of every char-aligned substring of at most max_token_length bytes — in how many samples it occurs
// at least once — over a packed batch.  The reference enumerates the windows per sample on rayon workers into
// a HashSet<&str> and merges per-chunk maps; here
//
//   window kernels : one lane per byte position; a position that starts a character extends a window
//                    character by character, hashing as it goes (FNV-1a 64), and keeps a window with the
//                    seeded insert-probability rule (the reference draws from an unseeded thread RNG,
//                    src/generate.rs:88,112: a counter hash of (seed, sample, window hash) stands in, the same
//                    in the Python mirror).  Pass 1 counts the kept windows per block of 256 positions, an
//                    exclusive scan gives every block its output range, pass 2 writes {hash, sample | position
//                    | length} in position order — hence in ascending sample order.
//   sort           : rocPRIM radix sort of the pairs by hash (stable: samples stay ascending inside a hash run).
//   runs           : a run of equal hashes is one substring; its document frequency is the number of distinct
//                    samples in the run (adjacent duplicates after the stable sort).  Run heads are numbered by
//                    an inclusive scan; every entry compares its window's BYTES with its run's first entry,
//                    so a 64-bit hash collision is detected, never silently merged.
//
// What stays on the host, as in the reference: the split regex (parts arrive as byte ranges), the allow regex
// (a pure function of the candidate: applied to the distinct substrings that come back instead of to every
// window) and the added / suggested tokens.  HBM-bound byte and integer work; no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "../../include/tgx.h"

tgx_status tgx_set_error(tgx_status st, const char* msg);  // tgx_api.cpp

namespace tgx {

constexpr uint64_t kFnvOffset = 0xCBF29CE484222325ULL, kFnvPrime = 0x100000001B3ULL;

// same function as tgx_generate_u01 below and tokengeex_amd/generate.py::_u01
__host__ __device__ inline double generate_u01(uint64_t seed, uint64_t sample, uint64_t h) {
    uint64_t x = seed ^ (sample * 0x9E3779B97F4A7C15ULL) ^ (h * 0xC2B2AE3D27D4EB4FULL);
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBULL;
    x ^= x >> 31;
    return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}

struct WindowParams {
    const uint8_t* text;          // padded: reads up to 32 bytes past a position are safe
    uint64_t n_bytes;
    const uint64_t* part_begin;   // parts (samples, or the split regex's matches) sorted, disjoint
    const uint64_t* part_end;
    const uint32_t* part_sample;
    const uint64_t* part_origin;  // offset in `text` of the first byte of the part's SAMPLE (the keep rule hashes the window's offset in its sample)
    uint64_t n_parts;
    const uint32_t* blk_part;     // per block of 256 positions: first part whose end lies beyond the block's start
    uint32_t max_len;
    double prob;
    uint64_t seed;
    uint64_t salt;                // 0: the sort key is the window's FNV-1a hash; else a second, salted hash of its bytes (the retry
                                  // after two different substrings met in one key)
    uint64_t key_mask;            // ~0; fewer bits only in the forced-collision test
    uint32_t* blk_count;          // pass 1: kept windows per block
    const uint64_t* blk_offs;     // pass 2: exclusive scan of blk_count
    uint64_t* keys;               // pass 2: window hash
    uint64_t* vals;               // pass 2: sample << 37 | position << 5 | (length - 1)
};

template <bool EMIT>
__global__ __launch_bounds__(256) void window_kernel(WindowParams P) {
    __shared__ uint32_t wave_cnt[4];
    const uint64_t p = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t sample = 0;
    uint64_t origin = 0;
    uint64_t end = 0;  // end of this position's part; 0 = the position starts no window
    if (p < P.n_bytes) {
        uint64_t k = P.blk_part[blockIdx.x];
        while (k < P.n_parts && P.part_end[k] <= p) k++;
        if (k < P.n_parts && P.part_begin[k] <= p && (P.text[p] & 0xC0u) != 0x80u) {  // inside a part, at a character start
            end = P.part_end[k];
            sample = P.part_sample[k];
            origin = P.part_origin[k];
        }
    }
    // the windows of this position in ascending length; `emit` is called for the kept ones (src/generate.rs:99-120)
    auto windows = [&](auto emit) {
        uint32_t kept = 0;
        uint64_t h = kFnvOffset, g = kFnvOffset ^ P.salt;
        for (uint32_t len = 1; len <= P.max_len && p + len <= end; ++len) {
            const uint64_t c = (uint64_t)P.text[p + len - 1];
            h = (h ^ c) * kFnvPrime;
            if (P.salt) g = ((g ^ c) * 0x9E3779B97F4A7C15ULL) ^ (g >> 29);  // an unrelated mixing chain: keys that met under h part here
            const bool boundary = (p + len == end) || (P.text[p + len] & 0xC0u) != 0x80u;
            // one draw per OCCURRENCE (src/generate.rs:84-89, 108-113: `rng.gen_range(0.0..1.0) < insert_probability` inside the loops
            // over positions and lengths): the stand-in for the thread RNG hashes (seed, sample, offset in the sample, length)
            if (boundary && (P.prob >= 1.0 || generate_u01(P.seed, sample, ((p - origin) << 8) | len) < P.prob)) {
                emit(kept, (P.salt ? g : h) & P.key_mask, len);
                kept++;
            }
        }
        return kept;
    };
    const uint32_t kept = windows([](uint32_t, uint64_t, uint32_t) {});
    // block-level exclusive prefix of `kept` (position order)
    uint32_t incl = kept;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t y = __shfl_up(incl, off);
        if ((int)lane >= off) incl += y;
    }
    if (lane == 63) wave_cnt[wave] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < wave; ++w) before += wave_cnt[w];
    if (!EMIT) {
        if (threadIdx.x == 255) P.blk_count[blockIdx.x] = before + incl;
        return;
    }
    const uint64_t at = P.blk_offs[blockIdx.x] + before + (incl - kept);
    windows([&](uint32_t i, uint64_t h, uint32_t len) {
        P.keys[at + i] = h;
        P.vals[at + i] = ((uint64_t)sample << 37) | (p << 5) | (uint64_t)(len - 1u);
    });
}

// head[i] = 1 where a new hash run starts
__global__ __launch_bounds__(256) void run_heads_kernel(const uint64_t* __restrict__ keys, uint64_t n, uint32_t* __restrict__ head) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint64_t j = i ? i - 1 : 0;
    head[i] = ((i == 0) | (keys[i] != keys[j])) ? 1u : 0u;
}

// rep[run] = the run's first entry; df[run] += 1 for every entry that opens a new sample inside its run.
// A run id outside [1, n_runs] cannot come out of an inclusive scan of the heads; it is never used as an index,
// and it is COUNTED (bad_runs): the host fails the call with TGX_ERR_DEVICE instead of returning frequencies with
// entries missing.  (Round 2 met a wild atomic address in an earlier, short-circuit form of this kernel whose
// source was not kept; the reconstructions under profiles/r03/run_count_short_circuit.* do not show a compiler
// defect, so the cause of that fault is NOT established — hence the loud guard.)
__global__ __launch_bounds__(256) void run_count_kernel(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ vals,
                                                        const uint32_t* __restrict__ run_id, uint64_t n, uint32_t n_runs,
                                                        uint64_t* __restrict__ rep, uint32_t* __restrict__ df,
                                                        unsigned long long* __restrict__ bad_runs) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint64_t j = i ? i - 1 : 0;
    const uint64_t k = keys[i], kp = keys[j], v = vals[i], vp = vals[j];
    const uint32_t r = run_id[i] - 1u;
    const bool head = (i == 0) | (k != kp);
    const bool newdoc = head | ((v >> 37) != (vp >> 37));
    if (r >= n_runs) {
        atomicAdd(bad_runs, 1ull);
        return;
    }
    if (head) rep[r] = v;
    if (newdoc) atomicAdd(&df[r], 1u);
}

// every entry's window against its run's representative, byte by byte
__global__ __launch_bounds__(256) void run_check_kernel(const uint8_t* __restrict__ text, const uint64_t* __restrict__ vals,
                                                        const uint32_t* __restrict__ run_id, uint64_t n, uint32_t n_runs,
                                                        const uint64_t* __restrict__ rep, unsigned long long* __restrict__ collisions,
                                                        unsigned long long* __restrict__ bad_runs) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    if (run_id[i] - 1u >= n_runs) {
        atomicAdd(bad_runs, 1ull);
        return;
    }
    const uint64_t a = vals[i], b = rep[run_id[i] - 1u];
    if (a == b) return;
    const uint32_t la = (uint32_t)(a & 31u) + 1u, lb = (uint32_t)(b & 31u) + 1u;
    bool same = la == lb;
    const uint64_t pa = (a >> 5) & 0xFFFFFFFFull, pb = (b >> 5) & 0xFFFFFFFFull;
    for (uint32_t j = 0; same && j < la; ++j) same = text[pa + j] == text[pb + j];
    if (!same) atomicAdd(collisions, 1ull);
}

}  // namespace tgx

extern "C" {

double tgx_generate_u01(uint64_t seed, uint64_t sample, uint64_t window_hash) { return tgx::generate_u01(seed, sample, window_hash); }

// Document frequencies of the char-aligned substrings of at most max_token_length (<= 32) bytes of the parts
// text[part_begin[k], part_end[k]) (sorted, disjoint; part_sample[k] non-decreasing), every OCCURRENCE kept with probability
// insert_probability (src/generate.rs:84-89, 108-113: a substring with k occurrences in a sample is counted for it with
// probability 1 - (1 - p)^k); part_origin[k]: where the part's sample begins in `text` (NULL: the part is its own sample) —
// the draw of an occurrence is tgx_generate_u01(seed, sample, offset in the sample << 8 | length).  Out: one entry per distinct substring — the position and length
// of one occurrence and the number of samples it occurs in — malloc'd (tgx_free), in ascending order of the
// substrings' sort keys (their FNV-1a hashes).  Two different substrings with one 64-bit key are DETECTED (every entry of
// a run is compared with the run's first, byte by byte) and resolved: the pass is sorted again under a second, salted hash
// of the windows' bytes, up to three times; *n_collisions reports how many entries met a foreign run in the attempts that
// were discarded.  text must be < 4 GiB.
static tgx_status substring_df_impl(int device, const uint8_t* text, uint64_t n_bytes, const uint64_t* part_begin,
                                    const uint64_t* part_end, const uint32_t* part_sample, const uint64_t* part_origin, uint64_t n_parts,
                                    uint32_t max_token_length, double insert_probability, uint64_t seed, uint64_t top_k,
                                    uint64_t** out_pos, uint32_t** out_len, uint32_t** out_df, uint64_t* n_out,
                                    uint64_t* n_windows, uint64_t* n_collisions, uint64_t* n_distinct, uint32_t* cutoff_df) {
    using namespace tgx;
    if (n_distinct) *n_distinct = 0;
    if (cutoff_df) *cutoff_df = 0;
    if (!out_pos || !out_len || !out_df || !n_out) return tgx_set_error(TGX_ERR_INVALID, "tgx_substring_df: NULL argument");
    *out_pos = nullptr;
    *out_len = nullptr;
    *out_df = nullptr;
    *n_out = 0;
    if (n_windows) *n_windows = 0;
    if (n_collisions) *n_collisions = 0;
    if (n_parts && (!text || !part_begin || !part_end || !part_sample)) return tgx_set_error(TGX_ERR_INVALID, "tgx_substring_df: NULL argument");
    // (windows of up to 32 bytes: the reference's CLI default is 24, src/cli.rs:675; a value carries length - 1 in 5 bits)
    if (max_token_length < 1 || max_token_length > 32) return tgx_set_error(TGX_ERR_UNSUPPORTED, "tgx_substring_df: max_token_length must be 1..32");
    if (n_bytes >= (1ull << 32)) return tgx_set_error(TGX_ERR_UNSUPPORTED, "tgx_substring_df: feed at most 4 GiB per call");
    if (n_parts == 0 || n_bytes == 0) return TGX_OK;
    for (uint64_t k = 0; k < n_parts; k++) {
        if (part_origin && part_origin[k] > part_begin[k]) return tgx_set_error(TGX_ERR_INVALID, "tgx_substring_df: a part begins before its sample");
        if (part_end[k] < part_begin[k] || part_end[k] > n_bytes || (k && part_begin[k] < part_end[k - 1]) ||
            (k && part_sample[k] < part_sample[k - 1]) || part_sample[k] >= (1u << 27))
            return tgx_set_error(TGX_ERR_INVALID, "tgx_substring_df: parts must be sorted, disjoint, inside the text, with ascending sample ids below 2^27");
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        (void)hipGetLastError();
        return tgx_set_error(TGX_ERR_DEVICE, "no usable HIP device (gfx950 required)");
    }
    std::vector<void*> owned;
    auto fail = [&](tgx_status st, const char* msg) {
        (void)hipDeviceSynchronize();
        for (void* p : owned) (void)hipFree(p);
        return tgx_set_error(st, msg);
    };
    auto dalloc = [&](size_t bytes) -> void* {
        void* p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(bytes, 256)) != hipSuccess) return nullptr;
        owned.push_back(p);
        return p;
    };
    // (TGX_DEBUG as tgx_api.cpp reads it: set, not empty, not "0")
    const char* dbg_env = getenv("TGX_DEBUG");
    const bool dbg = dbg_env && *dbg_env && *dbg_env != '0';
#define G_TRY(expr)                                                       \
    do {                                                                  \
        if (dbg) { fprintf(stderr, "[tgx] generate: %s\n", #expr); fflush(stderr); } \
        if ((expr) != hipSuccess) return fail(TGX_ERR_DEVICE, "HIP error in tgx_substring_df: " #expr); \
        if (dbg && hipDeviceSynchronize() != hipSuccess) return fail(TGX_ERR_DEVICE, "HIP error after " #expr); \
    } while (0)
    G_TRY(hipSetDevice(device));
    const uint64_t n_blocks = (n_bytes + 255) / 256;
    // first part whose end lies beyond each block's start
    std::vector<uint32_t> blk_part(n_blocks);
    {
        uint64_t k = 0;
        for (uint64_t b = 0; b < n_blocks; b++) {
            while (k < n_parts && part_end[k] <= b * 256) k++;
            blk_part[b] = (uint32_t)std::min<uint64_t>(k, 0xFFFFFFFFull);
        }
    }
    uint8_t* d_text = (uint8_t*)dalloc(n_bytes + 64);
    uint64_t* d_pb = (uint64_t*)dalloc(n_parts * 8);
    uint64_t* d_pe = (uint64_t*)dalloc(n_parts * 8);
    uint32_t* d_ps = (uint32_t*)dalloc(n_parts * 4);
    uint32_t* d_bp = (uint32_t*)dalloc(n_blocks * 4);
    uint32_t* d_bc = (uint32_t*)dalloc(n_blocks * 4);
    uint64_t* d_bo = (uint64_t*)dalloc((n_blocks + 1) * 8);
    unsigned long long* d_ctr = (unsigned long long*)dalloc(64);
    if (!d_text || !d_pb || !d_pe || !d_ps || !d_bp || !d_bc || !d_bo || !d_ctr) return fail(TGX_ERR_DEVICE, "out of device memory (generate)");
    G_TRY(hipMemset(d_text + n_bytes, 0, 64));
    G_TRY(hipMemcpy(d_text, text, n_bytes, hipMemcpyHostToDevice));
    G_TRY(hipMemcpy(d_pb, part_begin, n_parts * 8, hipMemcpyHostToDevice));
    G_TRY(hipMemcpy(d_pe, part_end, n_parts * 8, hipMemcpyHostToDevice));
    G_TRY(hipMemcpy(d_ps, part_sample, n_parts * 4, hipMemcpyHostToDevice));
    uint64_t* d_po = (uint64_t*)dalloc(n_parts * 8);
    if (!d_po) return fail(TGX_ERR_DEVICE, "out of device memory (generate)");
    G_TRY(hipMemcpy(d_po, part_origin ? part_origin : part_begin, n_parts * 8, hipMemcpyHostToDevice));
    G_TRY(hipMemcpy(d_bp, blk_part.data(), n_blocks * 4, hipMemcpyHostToDevice));
    G_TRY(hipMemset(d_ctr, 0, 64));
    WindowParams P{};
    P.text = d_text;
    P.n_bytes = n_bytes;
    P.part_begin = d_pb;
    P.part_end = d_pe;
    P.part_sample = d_ps;
    P.part_origin = d_po;
    P.n_parts = n_parts;
    P.blk_part = d_bp;
    P.max_len = max_token_length;
    P.prob = insert_probability;
    P.seed = seed;
    P.blk_count = d_bc;
    hipLaunchKernelGGL(window_kernel<false>, dim3((uint32_t)n_blocks), dim3(256), 0, 0, P);
    G_TRY(hipGetLastError());
    // exclusive scan of the block counts (u32 -> u64)
    {
        size_t tb = 0;
        G_TRY((rocprim::exclusive_scan(nullptr, tb, d_bc, d_bo, (uint64_t)0, (size_t)n_blocks, rocprim::plus<uint64_t>())));
        void* tmp = dalloc(tb);
        if (!tmp) return fail(TGX_ERR_DEVICE, "out of device memory (generate)");
        G_TRY((rocprim::exclusive_scan(tmp, tb, d_bc, d_bo, (uint64_t)0, (size_t)n_blocks, rocprim::plus<uint64_t>())));
    }
    uint64_t last_off = 0;
    uint32_t last_cnt = 0;
    G_TRY(hipMemcpy(&last_off, d_bo + (n_blocks - 1), 8, hipMemcpyDeviceToHost));
    G_TRY(hipMemcpy(&last_cnt, d_bc + (n_blocks - 1), 4, hipMemcpyDeviceToHost));
    const uint64_t M = last_off + last_cnt;
    if (n_windows) *n_windows = M;
    if (M == 0) {
        for (void* p : owned) (void)hipFree(p);
        return TGX_OK;
    }
    if (M >= (1ull << 32)) return fail(TGX_ERR_UNSUPPORTED, "tgx_substring_df: more than 2^32 windows in one call: feed smaller batches");
    uint64_t* d_keys = (uint64_t*)dalloc(M * 8);
    uint64_t* d_vals = (uint64_t*)dalloc(M * 8);
    uint64_t* d_keys2 = (uint64_t*)dalloc(M * 8);
    uint64_t* d_vals2 = (uint64_t*)dalloc(M * 8);
    if (!d_keys || !d_vals || !d_keys2 || !d_vals2) return fail(TGX_ERR_DEVICE, "out of device memory (generate: windows)");
    P.blk_offs = d_bo;
    P.keys = d_keys;
    P.vals = d_vals;
    // forced-collision test (TGX_KNOBS=1 TGX_GENERATE_COLLIDE=bits): the first attempt keeps that many key bits only
    uint64_t first_mask = ~0ull;
    {
        const char* kn = getenv("TGX_KNOBS");
        const char* fc = (kn && *kn && *kn != '0') ? getenv("TGX_GENERATE_COLLIDE") : nullptr;
        if (fc && atoi(fc) > 0 && atoi(fc) < 64) first_mask = (1ull << atoi(fc)) - 1ull;
    }
    uint32_t* d_head = reinterpret_cast<uint32_t*>(d_keys);  // (the unsorted key buffer is free once the sort is done)
    uint32_t* d_run = reinterpret_cast<uint32_t*>(d_keys) + M;
    const uint32_t mblocks = (uint32_t)((M + 255) / 256);
    uint64_t* d_rep = d_vals;  // the unsorted values are free after the sort: n_runs <= M
    uint32_t* d_df = nullptr;
    uint32_t n_runs = 0;
    unsigned long long coll_total = 0, coll = 0;
    for (int attempt = 0; attempt < 4; attempt++) {
        P.salt = attempt ? 0xD6E8FEB86659FD93ULL * (uint64_t)attempt : 0ull;
        P.key_mask = attempt ? ~0ull : first_mask;
        G_TRY(hipMemset(d_ctr, 0, 64));
        hipLaunchKernelGGL(window_kernel<true>, dim3((uint32_t)n_blocks), dim3(256), 0, 0, P);
        G_TRY(hipGetLastError());
        {
            size_t tb = 0;
            G_TRY(rocprim::radix_sort_pairs(nullptr, tb, d_keys, d_keys2, d_vals, d_vals2, (size_t)M));
            void* tmp = dalloc(tb);
            if (!tmp) return fail(TGX_ERR_DEVICE, "out of device memory (generate: sort)");
            G_TRY(rocprim::radix_sort_pairs(tmp, tb, d_keys, d_keys2, d_vals, d_vals2, (size_t)M));
        }
        // run ids: inclusive scan of the run heads
        hipLaunchKernelGGL(run_heads_kernel, dim3(mblocks), dim3(256), 0, 0, d_keys2, M, d_head);
        G_TRY(hipGetLastError());
        {
            size_t tb = 0;
            G_TRY((rocprim::inclusive_scan(nullptr, tb, d_head, d_run, (size_t)M, rocprim::plus<uint32_t>())));
            void* tmp = dalloc(tb);
            if (!tmp) return fail(TGX_ERR_DEVICE, "out of device memory (generate)");
            G_TRY((rocprim::inclusive_scan(tmp, tb, d_head, d_run, (size_t)M, rocprim::plus<uint32_t>())));
        }
        G_TRY(hipMemcpy(&n_runs, d_run + (M - 1), 4, hipMemcpyDeviceToHost));
        if (dbg) fprintf(stderr, "[tgx] generate: attempt %d M=%llu n_runs=%u\n", attempt, (unsigned long long)M, n_runs);
        if (n_runs == 0 || n_runs > M) return fail(TGX_ERR_DEVICE, "tgx_substring_df: inconsistent run count");
        d_df = (uint32_t*)dalloc((size_t)n_runs * 4);
        if (!d_df) return fail(TGX_ERR_DEVICE, "out of device memory (generate)");
        G_TRY(hipMemset(d_df, 0, (size_t)n_runs * 4));
        hipLaunchKernelGGL(run_count_kernel, dim3(mblocks), dim3(256), 0, 0, d_keys2, d_vals2, d_run, M, n_runs, d_rep, d_df, d_ctr + 1);
        G_TRY(hipGetLastError());
        hipLaunchKernelGGL(run_check_kernel, dim3(mblocks), dim3(256), 0, 0, d_text, d_vals2, d_run, M, n_runs, d_rep, d_ctr, d_ctr + 1);
        G_TRY(hipGetLastError());
        unsigned long long ctr[2] = {0, 0};  // [0] entries that met a foreign run (a key collision), [1] entries whose run id was out of range
        G_TRY(hipMemcpy(ctr, d_ctr, 16, hipMemcpyDeviceToHost));
        if (ctr[1]) return fail(TGX_ERR_DEVICE, "tgx_substring_df: entries with a run id outside [1, n_runs] (run ids corrupt)");
        coll = ctr[0];
        coll_total += coll;
        if (!coll) break;  // every run is one substring: the counts are exact
    }
    if (n_collisions) *n_collisions = coll_total;
    if (coll) return fail(TGX_ERR_UNSUPPORTED, "tgx_substring_df: different substrings shared a 64-bit key under four independent hashes");
    if (n_distinct) *n_distinct = n_runs;
    // top_k: only the top_k most frequent substrings leave the device (VocabularyGenerator::generate keeps the most
    // frequent ones, src/generate.rs:150-152, 199-213): a stable descending radix sort of (df, representative) —
    // equal frequencies stay in ascending hash order —, and the frequency of the first substring that is cut off
    uint64_t n_ret = n_runs;
    if (top_k && top_k < n_runs) {
        uint32_t* d_df2 = (uint32_t*)dalloc((size_t)n_runs * 4);
        uint64_t* d_rep2 = (uint64_t*)dalloc((size_t)n_runs * 8);
        if (!d_df2 || !d_rep2) return fail(TGX_ERR_DEVICE, "out of device memory (generate: top-k)");
        size_t tb = 0;
        G_TRY(rocprim::radix_sort_pairs_desc(nullptr, tb, d_df, d_df2, d_rep, d_rep2, (size_t)n_runs));
        void* tmp = dalloc(tb);
        if (!tmp) return fail(TGX_ERR_DEVICE, "out of device memory (generate: top-k sort)");
        G_TRY(rocprim::radix_sort_pairs_desc(tmp, tb, d_df, d_df2, d_rep, d_rep2, (size_t)n_runs));
        uint32_t cut = 0;
        G_TRY(hipMemcpy(&cut, d_df2 + top_k, 4, hipMemcpyDeviceToHost));
        if (cutoff_df) *cutoff_df = cut;
        d_df = d_df2;
        d_rep = d_rep2;
        n_ret = top_k;
    }
    const uint32_t n_runs_all = n_runs;
    (void)n_runs_all;
    n_runs = (uint32_t)n_ret;
    std::vector<uint64_t> rep(n_runs);
    uint64_t* pos = (uint64_t*)malloc(sizeof(uint64_t) * std::max<uint32_t>(n_runs, 1));
    uint32_t* len = (uint32_t*)malloc(sizeof(uint32_t) * std::max<uint32_t>(n_runs, 1));
    uint32_t* df = (uint32_t*)malloc(sizeof(uint32_t) * std::max<uint32_t>(n_runs, 1));
    if (!pos || !len || !df) {
        free(pos);
        free(len);
        free(df);
        return fail(TGX_ERR_INVALID, "tgx_substring_df: out of host memory");
    }
    if (hipMemcpy(rep.data(), d_rep, (size_t)n_runs * 8, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(df, d_df, (size_t)n_runs * 4, hipMemcpyDeviceToHost) != hipSuccess) {
        free(pos);
        free(len);
        free(df);
        return fail(TGX_ERR_DEVICE, "tgx_substring_df: copy of the result failed");
    }
    for (uint32_t r = 0; r < n_runs; r++) {
        pos[r] = (rep[r] >> 5) & 0xFFFFFFFFull;
        len[r] = (uint32_t)(rep[r] & 31u) + 1u;
    }
    for (void* p : owned) (void)hipFree(p);
#undef G_TRY
    *out_pos = pos;
    *out_len = len;
    *out_df = df;
    *n_out = n_runs;
    return TGX_OK;
}

tgx_status tgx_substring_df(int device, const uint8_t* text, uint64_t n_bytes, const uint64_t* part_begin,
                            const uint64_t* part_end, const uint32_t* part_sample, const uint64_t* part_origin, uint64_t n_parts,
                            uint32_t max_token_length, double insert_probability, uint64_t seed, uint64_t** out_pos,
                            uint32_t** out_len, uint32_t** out_df, uint64_t* n_out, uint64_t* n_windows,
                            uint64_t* n_collisions) {
    return substring_df_impl(device, text, n_bytes, part_begin, part_end, part_sample, part_origin, n_parts, max_token_length,
                             insert_probability, seed, 0, out_pos, out_len, out_df, n_out, n_windows, n_collisions, nullptr, nullptr);
}

// The same with only the top_k most frequent substrings returned (descending frequency; 0 = all): *n_distinct = the
// number of distinct substrings counted, *cutoff_df = the frequency of the most frequent substring that was NOT
// returned (0 when nothing was cut) — every substring that is not in the output occurs in at most that many
// samples, which is what lets a caller that accumulates several calls decide whether its selection is exact.
tgx_status tgx_substring_df_top(int device, const uint8_t* text, uint64_t n_bytes, const uint64_t* part_begin,
                                const uint64_t* part_end, const uint32_t* part_sample, const uint64_t* part_origin, uint64_t n_parts,
                                uint32_t max_token_length, double insert_probability, uint64_t seed, uint64_t top_k,
                                uint64_t** out_pos, uint32_t** out_len, uint32_t** out_df, uint64_t* n_out,
                                uint64_t* n_windows, uint64_t* n_collisions, uint64_t* n_distinct, uint32_t* cutoff_df) {
    return substring_df_impl(device, text, n_bytes, part_begin, part_end, part_sample, part_origin, n_parts, max_token_length,
                             insert_probability, seed, top_k, out_pos, out_len, out_df, n_out, n_windows, n_collisions, n_distinct, cutoff_df);
}

}  // extern "C"
