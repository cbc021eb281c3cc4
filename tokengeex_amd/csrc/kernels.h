// Kernel parameter blocks and launchers shared by kernels.hip and tgx_api.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tgx {

struct EncodeParams {
    const uint8_t* text;            // u8[N (+ pad)]
    const uint64_t* offs;           // u64[S+1]
    const uint32_t* order;          // u32[S] sample processing order (longest first)
    uint64_t n_samples;
    const void* trie;               // TrieRec[n_slots] (16 B each)
    const uint32_t* tokid;          // u32[n_slots]
    uint32_t root_base;
    uint32_t n_slots;               // trie slots (guards the handle -> id lookup)
    uint32_t lm;                    // max token length rounded up (<= 64)
    uint32_t* bp;                   // u32[N] back-pointer scratch
    uint32_t* tmp;                  // u32[N] right-aligned ids per sample
    uint32_t* counts;               // u32[S] tokens per sample
    uint32_t* status;               // u32[S] 1 = end of sample reachable (encode4 -> trace)
    uint8_t* bp8;                   // u8[N] rows4 back-pointers: len - 1, 0xFF = unreachable
    const void* tokhash;            // TokHashEntry[mask + 1]: token bytes -> id (rows4 trace)
    uint32_t tokhash_mask;
    uint32_t tokhash_seed;
    unsigned long long* err_sample; // min failing sample (init ~0)
    unsigned long long* queue;      // rows4: next unclaimed position of `order` (init 0)
    unsigned long long* redo_count; // encode4l_kernel: samples whose wave ran out of overflow entries (init 0) ...
    uint32_t* redo_list;            // ... and their indices, u32[S]: encode2_kernel redoes exactly those
    double dropout;
    uint64_t seed;
    uint32_t flags;                 // timing experiments only (TGX_FLAGS env): see kernels.hip
    unsigned long long* stamps;     // diagnostic build only (TGX_STAMPS=1): 8 u64 per wave
    // round 4 (trace2.hip): ids in their final place
    unsigned long long* endmask;    // u64[mask_words + 1]: sample s owns words mword[s] .. mword[s + 1); bit j of its word k = "a token ends
                                    // with the sample's byte 64 k + j" (every word is written by mark_kernel; the last is padding, 0)
    uint32_t trace_carry;           // trace kernels: waiting tokens carry over from sample to sample (short samples: trace_body.h)
    const uint64_t* mword;          // u64[S + 1]: exclusive scan of ceil(n / 64) over the samples
    uint64_t mask_words;            // mword[S]
    const uint64_t* prefix;         // u64[mask_words + 1]: set bits before each word (launch_mask_scan)
    uint32_t* ids_out;              // u32[T] the result's ids (emit_kernel)
};

// encode5_kernel / encode6_kernel (encode5.hip): 8-byte label-checked records, score values by rank
struct Encode5Params {
    const void* trie8;              // Trie8Rec[n_slots]
    uint32_t trie_bytes;            // 8 * n_slots
    const double* values;           // f64[n_values + 1]: values[0] = -inf, values[r] = the score value of rank r
    uint32_t root_base;
    uint32_t n_values;              // distinct score values of the vocabulary (<= 65 535)
    uint32_t n_hot;                 // ranks 1..n_hot are copied into the block's LDS; COLD builds read the rest from `values`
    uint32_t claim_chunk;           // consecutive samples of the order a row claims per atomic (>= 1)
    uint32_t list_off, root_off, idx_off;  // LDS layout (set by the launcher)
    uint32_t ctrl_off, ring_off, ring_slots;  // encode6_kernel: control words, ring of match-index buffers
    uint32_t pool;                            // encode6_kernel, COLD builds: pool entries per ring slot for values its walkers fetch (0: none)
    unsigned int* started;                    // encode5_kernel: when not null, every block adds 1 here once it is resident (host-visible memory:
                                              // the co-run launches the long-sample kernel only after all of them are)
};

struct CompactParams {
    const uint64_t* offs;
    const uint32_t* order;
    uint64_t n_samples;
    const uint32_t* tmp;
    const uint64_t* out_offs;
    uint32_t* ids;
};

struct EstepParams {
    const uint8_t* text;
    const uint64_t* offs;
    const uint32_t* order;
    uint64_t n_samples;
    const void* trie_fwd;           // tokens as written (forward sweep)
    const void* trie_rev;           // reversed tokens (backward sweep)
    uint32_t root_fwd, root_rev;
    uint32_t lm;                    // max token length rounded up (<= 64)
    uint64_t snippet_len;           // 81920 in the reference (prune.rs:75)
    double* alpha;                  // f64[N + S + pad] forward values, sample s at offs[s] + s
    double* expected_slot;          // f64[n_replicas][n_slots_rev] expected counts per reversed-trie slot
    uint32_t n_slots_rev, n_replicas;
    double* logz_sum;
    unsigned long long* err_sample; // min sample whose z is not normal (init ~0)
    double dropout;
    uint64_t seed;
};

hipError_t launch_pair_keys(const uint32_t* ids, const uint64_t* out_offs, uint64_t n_samples, uint32_t shift,
                            unsigned long long sentinel, unsigned long long* keys, uint32_t blocks, hipStream_t stream);
hipError_t pair_sort_temp_bytes(uint64_t n, size_t* bytes);
hipError_t pair_sort(void* temp, size_t temp_bytes, const unsigned long long* in, unsigned long long* out,
                     uint64_t n, unsigned int end_bit, hipStream_t stream);
hipError_t pair_rle_temp_bytes(uint64_t n, size_t* bytes);
hipError_t pair_rle(void* temp, size_t temp_bytes, const unsigned long long* sorted, uint64_t n,
                    unsigned long long* unique_out, unsigned int* counts_out, unsigned int* n_runs_out,
                    hipStream_t stream);

hipError_t pair_count_sort_temp_bytes(uint64_t n, size_t* bytes);
hipError_t pair_count_sort(void* temp, size_t temp_bytes, const unsigned int* counts_in, unsigned int* counts_out,
                           const unsigned long long* keys_in, unsigned long long* keys_out, uint64_t n,
                           hipStream_t stream);
hipError_t launch_pair_expand(const unsigned long long* keys, const unsigned int* cnt, uint64_t n, uint32_t shift,
                              unsigned long long* out_keys, unsigned long long* out_counts, hipStream_t stream);
constexpr uint32_t kHistMaxVocab = 36864;  // ids whose u32 counters fit one block's LDS (144 KiB)
hipError_t launch_ids_histogram(const uint32_t* ids, uint64_t n, uint32_t vocab, unsigned long long* out, uint32_t blocks,
                                hipStream_t stream);
hipError_t ids_sort_temp_bytes(uint64_t n, size_t* bytes);
hipError_t ids_sort(void* temp, size_t temp_bytes, const uint32_t* in, uint32_t* out, uint64_t n,
                    unsigned int end_bit, hipStream_t stream);
hipError_t ids_rle_temp_bytes(uint64_t n, size_t* bytes);
hipError_t ids_rle(void* temp, size_t temp_bytes, const uint32_t* sorted, uint64_t n, uint32_t* unique_out,
                   unsigned int* counts_out, unsigned int* n_runs_out, hipStream_t stream);

// four-snippets-per-wave E-step (estep4.hip): the work list is the list of snippets
struct Estep4Params {
    const uint8_t* text;            // 256-byte front pad: the backward sweep reads 16 bytes before a position
    const uint64_t* soffs;          // u64[K+1] snippet k = text[soffs[k] .. soffs[k+1])
    const uint32_t* order;          // u32[K] longest first
    uint64_t n_snips;
    const uint32_t* snip_sample;    // u32[K] sample of the snippet        (dropout hash only)
    const uint64_t* snip_base;      // u64[K] its byte offset in the sample (dropout hash only)
    const void* trie_fwd;
    const void* trie_rev;
    uint32_t root_fwd, root_rev;
    double* alpha;                  // f64[N + K + pad]: snippet k's A[0..n] at soffs[k] + k
    double* zarr;                   // f64[K] z = A[n] per snippet
    int32_t* alpha_exp;             // linear-domain kernels (estep4l.hip): power-of-two exponent of each block of 16
                                    // alpha values, snippet k's blocks at (soffs[k] >> 4) + k
    double* expected_slot;          // f64[n_replicas][n_slots_rev]
    uint32_t n_slots_rev, n_replicas;
    uint32_t n_hot;                 // estep4l_bwd_kernel: slots summed in the block's LDS (set by its launcher)
    double* logz_sum;
    unsigned long long* err_snip;   // min snippet whose z is not normal (init ~0)
    uint32_t flags;                 // timing experiments only (TGX_FLAGS with TGX_DEBUG=1): 8 = no cold-slot atomics
    unsigned long long* range_flag; // linear-domain forward kernel: set when the pass needs the log-domain kernels (init 0)
    unsigned long long* queue_fwd;  // next unclaimed position of `order`, forward / backward kernel (init 0)
    unsigned long long* queue_bwd;
    double dropout;
    uint64_t seed;
};
// cuts.hip: positions no token match crosses (the lattice factorises there): long snippets are cut into pieces
struct CutParams {
    const uint8_t* text;
    const uint64_t* soffs;          // u64[K+1] snippets
    const uint32_t* snip_sample;    // u32[K], u64[K]: dropout hash bookkeeping of the snippets
    const uint64_t* snip_base;
    const uint32_t* win_snip;       // u32[W] snippet of window w (windows in text order)
    const uint32_t* win_k;          // u32[W] its number within the snippet; window 0 stands for the snippet's start
    uint64_t n_windows;
    uint32_t window;                // bytes per window
    const void* trie;               // 16-byte records (check, base | terminal << 31, ...)
    uint32_t root;
    uint32_t lmx;                   // longest token, rounded up to 16 or 32
    double dropout;
    uint64_t seed;
    uint64_t* bound;                // out u64[W]: the window's boundary (absolute byte offset) or ~0
    uint32_t* flag;                 // out u32[W]: 1 iff bound[w] != ~0
};
hipError_t launch_cut_windows(const CutParams& p, hipStream_t stream);
hipError_t launch_cut_scatter(const CutParams& p, const uint64_t* pos, uint64_t n_bytes, uint64_t* poffs, uint32_t* psample,
                              uint64_t* pbase, uint32_t* psnip, hipStream_t stream);
hipError_t launch_piece_len(const uint64_t* poffs, const uint64_t* n_pieces, uint32_t* len, uint32_t* idx, unsigned long long* longest,
                            uint64_t cap, hipStream_t stream);
hipError_t piece_sort_temp_bytes(uint64_t n, size_t* bytes);
hipError_t piece_sort(void* temp, size_t temp_bytes, const uint32_t* len_in, uint32_t* len_out, const uint32_t* idx_in, uint32_t* idx_out,
                      uint64_t n, hipStream_t stream);
hipError_t launch_piece_z_check(const double* zarr, const uint32_t* psnip, uint64_t n_pieces, double* zsnip, uint64_t n_snips,
                                unsigned long long* err_snip, hipStream_t stream);

// estep7_kernel (estep7.hip): the E-step with one walk per position — every trip of a row is a lattice of its own
// between two positions no match crosses; match entries are token ranks (trie_build.h: Trie8T)
// (what a row needs once per piece or less often lives behind a pointer, in device memory: as kernel arguments these
// twenty scalar-register pairs pushed the buffer resources of the walk out to spilled lanes — 4 v_readlane and their
// hazard nops per walk level)
struct Estep7Work {
    const uint64_t* soffs;          // u64[K+1] pieces (or snippets): k = text[soffs[k] .. soffs[k+1])
    const uint32_t* order;          // u32[K]
    uint64_t n_snips;
    const uint32_t* snip_sample;    // u32[K], u64[K]: sample and offset in the sample (dropout hash only)
    const uint64_t* snip_base;
    const uint32_t* snip_of;        // u32[K] snippet a piece belongs to (z is summed per snippet); null: k itself
    double* zsnip;                  // f64[snippets] log Z per snippet (sum of its pieces' trips)
    double* logz_sum;
    unsigned long long* range_flag; // 1: a position nothing reaches or a value out of range; 4: redo list full
    unsigned long long* queue;
    unsigned long long* redo_count; // stretches the kernel could not do (no cut within a trip) ...
    uint64_t* redo_offs;            // ... u64[2 cap]: [2 i], [2 i + 1] = begin and end of stretch i
    uint32_t* redo_sample;          // u32[2 cap], u64[2 cap], u32[2 cap]: sample, offset in the sample, snippet (entries 2 i and 2 i + 1 alike)
    uint64_t* redo_base;
    uint32_t* redo_snip;
    uint64_t redo_cap;
};
struct Estep7Params {
    const uint8_t* text;
    const Estep7Work* work;         // device memory (the launcher uploads `host_work` there)
    Estep7Work host_work;           // filled by the caller; not read by the kernel
    const void* trie8t;             // Trie8TRec[n_slots]
    uint32_t n_slots;
    uint32_t root_base;
    const double* wtab;             // f64[n_tok + 1]: [0] = 0, [r] = exp(score of the token of rank r)
    uint32_t n_tok;                 // tokens that can match (ranks 1 .. n_tok)
    uint32_t ovf_limit;             // 16-bit builds: a match of a rank beyond it sends its trip to the redo kernel (0xFFFFFFFF: none can occur)
    uint32_t n_hot;                 // ranks 1 .. n_hot have {sum, w} in the block's LDS
    double* expected;               // f64[n_tok + 1] expected counts by rank
    double dropout;
    uint64_t seed;
    uint32_t claim_chunk;
    uint32_t root_off, zero_off, idx_off;  // LDS layout (set by the launcher)
    uint32_t flags;                 // timing experiments only (TGX_FLAGS with TGX_DEBUG=1): results are WRONG when set
    unsigned long long* stamps;     // diagnostic runs only (TGX_STAMPS=7 with TGX_DEBUG=1): 8 u64 per wave
};
struct Estep7RedoParams {
    const uint8_t* text;
    const uint64_t* redo_offs;      // as written by estep7_kernel
    const uint32_t* redo_sample;
    const uint64_t* redo_base;
    const uint32_t* redo_snip;
    uint64_t n_redo;
    const uint64_t* tbase;          // u64[n_redo]: trips of the stretches before stretch i (the scratch layout)
    const void* trie8t;
    uint32_t n_slots;
    uint32_t root_base;
    const double* wtab;
    uint32_t n_tok, n_hot;
    double* expected;
    double* zsnip;
    double* logz_sum;
    unsigned long long* range_flag;
    unsigned long long* queue;
    double* alpha;                  // f64[16 trips]
    int32_t* aexp;                  // i32[trips]
    unsigned char* mscratch;        // row images of the trips' match entries: trips x 512 (1024) bytes
    double dropout;
    uint64_t seed;
    uint32_t root_off, zero_off, idx_off;  // LDS layout (set by the launcher)
};
uint32_t estep7_redo_max_hot(bool wide);
hipError_t launch_estep7_redo(Estep7RedoParams p, bool wide, uint32_t num_cus, hipStream_t stream);
uint32_t estep7_lds_layout(uint32_t n_hot, bool wide, int waves, int ppl, uint32_t* root_off, uint32_t* zero_off, uint32_t* idx_off);
uint32_t estep7_max_hot(bool wide, int waves, int ppl, uint32_t budget);
hipError_t estep7_waves_per_simd(bool dropout, bool cold, bool wide, int ppl, int* out);
hipError_t launch_estep7(Estep7Params p, bool wide, int ppl, int waves, uint32_t blocks, hipStream_t stream);
// How often every token MATCHES in a sample of the text (chunks of `chunk` bytes every `stride` bytes): counts[rank] += 1
// per position of the sample and token that is a prefix of the text there.  The E-step adds to a token's expected count
// once per match, whatever its score: the ranks that stay in LDS are chosen by these counts (tgx_api.cpp).
hipError_t launch_match_count(const uint8_t* text, uint64_t n_bytes, uint32_t chunk, uint64_t stride, const void* trie8t, uint32_t n_slots,
                              uint32_t root_base, uint32_t n_tok, unsigned int* counts, uint32_t num_cus, hipStream_t stream);
// the same over encode5_kernel's records: counts[rank of the score value] (walks of up to max_len bytes)
hipError_t launch_value_count(const uint8_t* text, uint64_t n_bytes, uint32_t chunk, uint64_t stride, const void* trie8, uint32_t n_slots,
                              uint32_t root_base, uint32_t n_values, uint32_t max_len, unsigned int* counts, uint32_t num_cus, hipStream_t stream);
// rec[t].tok = perm[rec[t].tok] for every slot of the 8-byte E-step records (and encode5_kernel's: the second word is the rank there too)
hipError_t launch_rank_remap(void* trie8t, uint32_t n_slots, const uint32_t* perm, hipStream_t stream);
hipError_t launch_snip_z_check(const double* zsnip, uint64_t n_snips, unsigned long long* err_snip, hipStream_t stream);
hipError_t launch_piece_z_add(const double* zarr, const uint32_t* psnip, const uint32_t* order, uint64_t n_order, double* zsnip, hipStream_t stream);

hipError_t estep4_prepare();
hipError_t launch_estep4_fwd(const Estep4Params& p, uint32_t num_cus, hipStream_t stream);
hipError_t launch_estep4_bwd(const Estep4Params& p, uint32_t num_cus, hipStream_t stream);
hipError_t estep4l_prepare();
hipError_t launch_estep4l_fwd(const Estep4Params& p, int ppl, bool long_tokens, uint32_t num_cus, hipStream_t stream);
hipError_t launch_estep4l_bwd(const Estep4Params& p, int ppl, bool long_tokens, uint32_t num_cus, uint32_t groups_wanted, hipStream_t stream);
hipError_t launch_estep4_reduce(const double* rep, double* out, uint32_t n_slots, uint32_t n_replicas,
                                hipStream_t stream);

uint32_t estep_lds_bytes_per_block(uint32_t lm);
uint32_t estep_waves_per_block(uint32_t lm);
hipError_t estep_max_blocks_per_cu(uint32_t lm, int* out);
hipError_t launch_estep(const EstepParams& p, uint32_t blocks, hipStream_t stream);

uint32_t encode_lds_bytes_per_block(uint32_t lm);
uint32_t encode_waves_per_block(uint32_t lm);
hipError_t encode_max_blocks_per_cu(uint32_t lm, int* out);
hipError_t launch_encode(const EncodeParams& p, uint32_t blocks, hipStream_t stream);
uint32_t encode4_group_bytes();
hipError_t encode4_waves_per_simd(bool dropout, int ppl, bool root, int* out);
hipError_t launch_encode4(const EncodeParams& p, int ppl, int waves, uint32_t blocks, bool root, hipStream_t stream);
uint32_t encode4_lds_bytes(int waves, int ppl, bool root);
hipError_t launch_trace(const EncodeParams& p, uint32_t blocks, hipStream_t stream);
// trace2.hip: mark_kernel (the hop chain alone: one bit per text byte), popcount scan, sample offsets, emit_kernel
hipError_t launch_mark(const EncodeParams& p, uint32_t blocks, uint32_t lm, bool permuted, hipStream_t stream);
hipError_t mask_scan_temp_bytes(uint64_t n_words, size_t* bytes);
hipError_t launch_mask_scan(const unsigned long long* mask, uint64_t* prefix, uint64_t n_words, void* temp, size_t temp_bytes, hipStream_t stream);
hipError_t launch_sample_offs(const uint64_t* mword, uint64_t n_samples, const uint64_t* prefix, uint64_t* out_offs, hipStream_t stream);
hipError_t launch_emit(const EncodeParams& p, uint32_t lm, uint32_t num_cus, hipStream_t stream);
uint32_t encode5_lds_layout(uint32_t n_hot, bool long_tokens, int waves, int ppl, uint32_t* list_off, uint32_t* root_off, uint32_t* idx_off);
uint32_t encode5_max_hot(bool long_tokens, int waves, int ppl, uint32_t budget);
hipError_t encode5_waves_per_simd(bool dropout, bool cold, int ppl, bool long_tokens, int* out);
uint32_t encode6_lds_layout(uint32_t n_hot, uint32_t pool, uint32_t* root_off, uint32_t* ctrl_off, uint32_t* ring_off);
uint32_t encode6_max_hot(uint32_t budget, uint32_t pool);
uint32_t encode6_pool_total(uint32_t pool);  // pool entries of a block (index space they take)
// estep5_fwd_kernel (encode5.hip): the E-step's forward sweep over the 8-byte ranked records; q.values = exp(score value) by rank
hipError_t estep5_waves_per_simd(bool dropout, bool cold, int ppl, int* out);
hipError_t launch_estep5_fwd(const struct Estep4Params& p, Encode5Params q, bool cold, int ppl, int waves, uint32_t blocks, hipStream_t stream);
hipError_t launch_encode6(const EncodeParams& p, Encode5Params q, bool cold, uint32_t blocks, hipStream_t stream);
hipError_t launch_encode5(const EncodeParams& p, Encode5Params q, bool cold, int ppl, bool long_tokens, int waves, uint32_t blocks,
                          uint32_t min_lds, hipStream_t stream);
hipError_t launch_encode2(const EncodeParams& p, uint32_t num_cus, bool permuted, hipStream_t stream);   // encode2.hip
hipError_t launch_trace32(const EncodeParams& p, uint32_t blocks, bool permuted, hipStream_t stream);
hipError_t launch_encode4l(const EncodeParams& p, uint32_t num_cus, hipStream_t stream);  // encode4l.hip
hipError_t scan_temp_bytes(uint64_t n, size_t* bytes);
hipError_t launch_scan(uint32_t* counts, uint64_t* offsets, uint64_t n, void* temp, size_t temp_bytes, hipStream_t stream);
hipError_t launch_compact(const CompactParams& p, uint32_t blocks, bool rows, hipStream_t stream);

}  // namespace tgx
