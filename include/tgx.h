/*
 * tgx.h — C ABI of the MI355X-native TokenGeeX Unigram encode / E-step path.
 *
 * This is the drop-in boundary: a Rust `extern "C"` block, a
 * C++ host or Python/ctypes bind exactly these symbols.  Plain pointers and
 * sizes only; no exceptions or unwinding cross it; every function returns a
 * tgx_status.  All reference citations are relative to /root/reference.
 *
 * Batch format (replaces Vec<&str> / &[&str] of the reference's batch loops):
 *   text  : uint8_t[N]      all samples' bytes back to back (already processed
 *                           by the tokenizer's processors, as in cli.rs:276-287)
 *   offs  : uint64_t[S+1]   sample i = text[offs[i] .. offs[i+1])
 * Result format (replaces Vec<Vec<u32>>):
 *   ids   : uint32_t[T]     token ids of all samples back to back
 *   ooffs : uint64_t[S+1]   sample i's ids = ids[ooffs[i] .. ooffs[i+1])
 *
 * The library fails loudly (TGX_ERR_DEVICE) when no gfx950 device is usable;
 * there is no CPU fallback behind these entry points.
 */
#ifndef TGX_H
#define TGX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TGX_ABI_VERSION 1

/* tokengeex::Error — src/lib.rs:219-224 (+ build-specific codes >= 5) */
typedef enum tgx_status {
    TGX_OK = 0,
    TGX_ERR_IO = 1,                 /* Error::IO                                   */
    TGX_ERR_JSON = 2,               /* Error::SerdeJSON                            */
    TGX_ERR_TOKEN_ID_OOB = 3,       /* Error::TokenIdOutOfBounds(id)               */
    TGX_ERR_NO_PATH = 4,            /* Error::NoPath(pos, len), src/model.rs:119   */
    TGX_ERR_DEVICE = 5,             /* HIP runtime / no GPU / kernel failure       */
    TGX_ERR_Z_NOT_NORMAL = 6,       /* the panic at src/prune.rs:90-96             */
    TGX_ERR_INVALID = 7,            /* bad argument                                */
    TGX_ERR_UNSUPPORTED = 8         /* e.g. a token longer than TGX_MAX_TOKEN_LEN  */
} tgx_status;

/* Longest vocabulary token (bytes) the device lattice handles.  The reference's
 * recipes use 16 (README.md:161) to 24 (cli.rs:723). */
#define TGX_MAX_TOKEN_LEN 64

/* E-step snippet length: const MAX_SAMPLE_LENGTH = 8192 * 10, src/prune.rs:75 */
#define TGX_ESTEP_SNIPPET_LEN 81920

typedef struct tgx_model tgx_model;   /* Model: vocab + trie, src/model.rs:8-12   */
typedef struct tgx_corpus tgx_corpus; /* a packed batch resident in HBM            */
typedef struct tgx_result tgx_result; /* ids + offsets of one encode pass          */

/* ---- error reporting ----------------------------------------------------- */
/* Message of the last failure on the calling thread, formatted like the
 * reference's Display impl (src/lib.rs:238-249), e.g.
 * "no path to position 12/12", "token id 7 is out of bounds". */
const char *tgx_last_error(void);
/* Details of the last TGX_ERR_NO_PATH / TGX_ERR_Z_NOT_NORMAL: the LOWEST failing
 * sample index (the reference leaves it unspecified, src/tokenizer.rs:107-110),
 * and (pos, len) of Error::NoPath. */
void tgx_last_error_detail(uint64_t *sample, uint64_t *pos, uint64_t *len);
int tgx_abi_version(void);
/* Number of usable gfx950 devices (0 when none; never fails). */
int tgx_device_count(void);

/* ---- Model ---------------------------------------------------------------- */
/* Model::from(vocab) — src/model.rs:16-30.  Token i = bytes[offs[i]..offs[i+1]),
 * id = i, scores[i] = vocab[i].score passed as raw f64 (never re-parsed).
 * Later duplicates overwrite earlier ones (src/trie.rs:19); empty tokens never
 * match (src/trie.rs:53-61).  Flattens the byte trie into an XOR double-array in
 * HBM on `device`.  The handle is immutable and may be used concurrently; mutation
 * = build a new handle (as `*model = Model::from(vocab)`, src/prune.rs:48,53). */
tgx_status tgx_model_create(const uint8_t *bytes, const uint64_t *offs, const double *scores,
                            uint32_t vocab_size, int device, tgx_model **out);
/* As tgx_model_create; flags: TGX_MODEL_FOR_ESTEP builds the tables of the E-step's backward sweep on a second
 * host thread during creation instead of at the first tgx_estep call (Model::from is rebuilt for every EM
 * sub-iteration, src/prune.rs:48). */
#define TGX_MODEL_FOR_ESTEP 1u
tgx_status tgx_model_create_ex(const uint8_t *bytes, const uint64_t *offs, const double *scores,
                               uint32_t vocab_size, int device, uint32_t flags, tgx_model **out);

/* A model for a SUBSET of `parent`'s vocabulary (keep_ids: ascending ids of the parent's tokens; the new id of a token
 * is its position in keep_ids) with new scores, on the parent's double-arrays: nothing is rebuilt, the tokens that are
 * gone lose their terminal marks — a trie with dead branches gives the same matches.  What `prune` needs three times
 * per iteration (src/prune.rs:36-56: a model per EM sub-iteration and one for the pruning step, each a subset of the one
 * before).  Same flags as tgx_model_create_ex; the parent stays valid.  TGX_ERR_UNSUPPORTED when the parent's vocabulary
 * has duplicate tokens (build the model with tgx_model_create_ex then). */
tgx_status tgx_model_create_derived(const tgx_model *parent, const uint32_t *keep_ids, uint32_t n_keep,
                                    const double *scores, uint32_t flags, tgx_model **out);
void tgx_model_destroy(tgx_model *m);
uint32_t tgx_model_vocab_size(const tgx_model *m);     /* Model::vocab_size, src/model.rs:179 */
uint32_t tgx_model_max_token_len(const tgx_model *m);
uint64_t tgx_model_trie_bytes(const tgx_model *m);     /* size of the device table */
int tgx_model_device(const tgx_model *m);

/* Model::common_prefix_search — src/model.rs:132-138 / src/trie.rs:44-64 (host
 * twin of the device walk over the same flattened table).  Writes up to cap
 * (id, len) pairs in ascending length; *count = number found. */
tgx_status tgx_common_prefix_search(const tgx_model *m, const uint8_t *s, uint64_t n,
                                    uint32_t *ids, uint32_t *lens, uint64_t cap, uint64_t *count);

/* ---- Tokenizer front and back over packed buffers (host code, no device; csrc/frontback.cpp) ----------
 * The reference runs these per sample on its rayon workers: the special-token splitter
 * (src/tokenizer.rs:299-347), CrlfProcessor::preprocess (src/processor.rs:46-54), the assembly of a
 * sample's ids from its segments (src/tokenizer.rs:65-90) and decode_batch (src/tokenizer.rs:126-187 over
 * src/model.rs:146-160, String::from_utf8_lossy per run of base ids).  Strings cross as UTF-8 bytes +
 * u64 offsets.  Arrays returned through pointer-to-pointer arguments are malloc'd: tgx_free. */
tgx_status tgx_split_specials(const uint8_t *text, const uint64_t *offs, uint64_t n_samples,
                              const uint8_t *special_bytes, const uint64_t *special_offs, uint32_t n_specials,
                              uint64_t *seg_offs, uint64_t **seg_begin, uint64_t **seg_end,
                              int32_t **seg_special, uint64_t *n_segments);
tgx_status tgx_pack_segments(const uint8_t *text, const uint64_t *seg_begin, const uint64_t *seg_end,
                             const int32_t *seg_special, uint64_t n, int crlf, uint8_t *out_text,
                             uint64_t *out_offs, uint64_t *n_out);
/* UnicodeProcessor::preprocess (src/processor.rs:124-137) over packed segments: form 0 NFD, 1 NFC, 2 NFKD, 3 NFKC (UAX #15;
 * tables of Unicode tgx_unidata_version()).  Bytes that are not UTF-8 pass through unchanged.  out_text / out_offs are
 * malloc'd (tgx_free): n_segs segments back to back, offsets u64[n_segs + 1]. */
tgx_status tgx_normalize_segments(uint32_t form, const uint8_t *text, const uint64_t *seg_begin, const uint64_t *seg_end,
                                  uint64_t n_segs, uint8_t **out_text, uint64_t **out_offs);
const char *tgx_unidata_version(void);
tgx_status tgx_assemble_ids(const uint64_t *seg_offs, const int32_t *seg_special, uint64_t n_samples,
                            const uint32_t *ids, const uint64_t *id_offs, uint32_t vocab_size,
                            uint32_t *out_ids, uint64_t *out_offs);
tgx_status tgx_decode_batch(const uint8_t *vocab_bytes, const uint64_t *vocab_offs, uint32_t vocab_size,
                            const uint8_t *special_bytes, const uint64_t *special_offs, uint32_t n_specials,
                            const uint32_t *ids, const uint64_t *id_offs, uint64_t n_samples,
                            int include_special, uint8_t **out_text, uint64_t *out_offs,
                            uint64_t *bad_sample, uint64_t *bad_id);
uint64_t tgx_utf8_lossy(const uint8_t *s, uint64_t n, uint8_t *out);

/* ---- `generate`: document frequencies of substrings on the device (csrc/generate.hip) ----------------
 * VocabularyGenerator::feed (src/generate.rs:54-139): in how many samples does every char-aligned substring of
 * at most max_token_length (<= 32) bytes occur?  Parts are the byte ranges the windows may lie in (the samples,
 * or the matches of the split regex), sorted, disjoint, with ascending sample ids; part_origin[k] = where the part's
 * sample begins in `text` (NULL: every part is a sample of its own).  Every OCCURRENCE is kept with probability
 * insert_probability, as in the reference's loops (src/generate.rs:84-89, 108-113; a substring with k occurrences in a
 * sample counts for it with probability 1 - (1 - p)^k): occurrence (sample, offset o in the sample, length l) is kept iff
 * tgx_generate_u01(seed, sample, o << 8 | l) < insert_probability (the reference draws from an unseeded thread RNG).  Out: one entry per distinct substring — position and length of one occurrence, number
 * of samples — malloc'd (tgx_free).  Two different substrings in one 64-bit sort key are detected (every entry of a run is
 * compared with the run's first, byte by byte) and resolved: the pass is sorted again under a second, independent hash of
 * the windows' bytes (up to three times; *n_collisions = entries that had met a foreign run in the discarded attempts).
 * More than 2^32 - 1 kept windows in one call: TGX_ERR_UNSUPPORTED ("feed smaller batches"). */
tgx_status tgx_substring_df(int device, const uint8_t *text, uint64_t n_bytes, const uint64_t *part_begin,
                            const uint64_t *part_end, const uint32_t *part_sample, const uint64_t *part_origin, uint64_t n_parts,
                            uint32_t max_token_length, double insert_probability, uint64_t seed,
                            uint64_t **out_pos, uint32_t **out_len, uint32_t **out_df, uint64_t *n_out,
                            uint64_t *n_windows, uint64_t *n_collisions);
/* The same with only the top_k most frequent substrings leaving the device, in descending frequency (top_k = 0:
 * all): VocabularyGenerator::generate keeps the most frequent substrings (src/generate.rs:150-152, 199-213), and
 * copying every distinct substring of a corpus to the host (254 M for 64 MiB of text) is what bounded `feed`.
 * *n_distinct = distinct substrings counted; *cutoff_df = the frequency of the most frequent substring NOT
 * returned (0 when nothing was cut): whatever is missing from the output occurs in at most that many samples. */
tgx_status tgx_substring_df_top(int device, const uint8_t *text, uint64_t n_bytes, const uint64_t *part_begin,
                                const uint64_t *part_end, const uint32_t *part_sample, const uint64_t *part_origin, uint64_t n_parts,
                                uint32_t max_token_length, double insert_probability, uint64_t seed,
                                uint64_t top_k, uint64_t **out_pos, uint32_t **out_len, uint32_t **out_df,
                                uint64_t *n_out, uint64_t *n_windows, uint64_t *n_collisions,
                                uint64_t *n_distinct, uint32_t *cutoff_df);
double tgx_generate_u01(uint64_t seed, uint64_t sample, uint64_t window_hash);

/* ---- host-only trie introspection (no device needed) ------------------------
 * The same flattening tgx_model_create uploads, built on the host alone, so that
 * the layout can be validated (and inspected) on machines without a GPU. */
typedef struct tgx_flat_trie tgx_flat_trie;
tgx_status tgx_flat_trie_build(const uint8_t *bytes, const uint64_t *offs, const double *scores,
                               uint32_t vocab_size, tgx_flat_trie **out);
void tgx_flat_trie_free(tgx_flat_trie *t);
/* as tgx_common_prefix_search; returns the number of matches */
uint64_t tgx_flat_trie_search(const tgx_flat_trie *t, const uint8_t *s, uint64_t n, uint32_t *ids,
                              uint32_t *lens, uint64_t cap);
void tgx_flat_trie_stats(const tgx_flat_trie *t, uint64_t *n_slots, uint64_t *n_nodes,
                         uint32_t *max_token_len);
/* The same search over the 8-byte label-checked records encode5_kernel walks (built from the same slot
 * assignment: the walk keeps only the record and compares its label with the text byte), plus figures of the
 * score table for a copy of max_hot values in LDS: *n_hot = values in that copy (ranks 1..n_hot of the
 * vocabulary's distinct score values), *n_cold terminal slots whose value is outside it (read from HBM / L2 by
 * the kernels), *hot_coverage the expected share of matches the copy serves.  Any out pointer may be NULL.
 * Returns the number of matches, or UINT64_MAX when the records cannot be built (more than 65 535 distinct
 * score values, or more than 2^21 slots). */
uint64_t tgx_flat_trie_search8(const tgx_flat_trie *t, const uint8_t *bytes, const uint64_t *offs,
                               const double *scores, uint32_t max_hot, const uint8_t *s, uint64_t n,
                               uint32_t *ids, uint32_t *lens, uint64_t cap, uint32_t *n_hot,
                               uint64_t *n_cold, double *hot_coverage);
/* copies the slot table: check[n_slots], base_flags[n_slots] (bit 31 = terminal), tokid[n_slots] */
void tgx_flat_trie_copy(const tgx_flat_trie *t, uint32_t *check, uint32_t *base_flags, uint32_t *tokid);
/* Builds the token-bytes -> id table the trace kernel probes and looks every token up again through it
 * (host only).  TGX_OK and *mismatches == 0 when every token of <= 32 bytes maps back to the id the trie
 * gives it; TGX_ERR_UNSUPPORTED when the table cannot be built (a token longer than 32 bytes, or no
 * collision-free seed) — the model then keeps the one-sample-per-wave kernel. */
tgx_status tgx_tok_hash_selftest(const uint8_t *bytes, const uint64_t *offs, uint32_t vocab_size,
                                 uint32_t *seed, uint64_t *mismatches);
/* out-of-line copy of tgx_dropout_u01 (below) for bindings that cannot inline C */
double tgx_dropout_u01_host(uint64_t seed, uint64_t sample, uint64_t pos, uint32_t len);

/* ---- encode: Tokenizer::encode_ordinary_batch / the rayon loop ----------- */
/* src/tokenizer.rs:102-123 over src/model.rs:59-129.  Host buffers in, result
 * handle out (host-readable).  dropout/seed: see tgx_dropout_u01.  On
 * TGX_ERR_NO_PATH *out is NULL and tgx_last_error_detail names the sample. */
tgx_status tgx_encode_batch(tgx_model *m, const uint8_t *text, const uint64_t *offs,
                            uint64_t n_samples, double dropout, uint64_t seed, tgx_result **out);
/* The same with host buffers on both sides: the ids are written to ids_out (room for ids_cap entries; the
 * number of bytes of the batch always suffices) and the exclusive token offsets to offs_out[n_samples + 1].
 * Large batches are cut at sample boundaries into chunks that go through three stages — upload, kernels,
 * download — on three host threads and two copy streams, so the link works in both directions beside the kernels.  Errors as tgx_encode_batch (the lowest failing sample of the batch is reported). */
tgx_status tgx_encode_batch_host(tgx_model *m, const uint8_t *text, const uint64_t *offs, uint64_t n_samples,
                                 double dropout, uint64_t seed, uint32_t *ids_out, uint64_t ids_cap,
                                 uint64_t *offs_out, uint64_t *n_tokens);

/* The same from ONE host process over SEVERAL devices: `models` holds one handle per GPU (tgx_model_create with
 * device = 0 .. n - 1, the same vocabulary), the batch is cut at sample boundaries into byte-balanced shards, every shard
 * runs tgx_encode_batch_host on a host thread of its own, and ids / offsets come back packed in sample order — what the
 * reference's single-call batch (src/tokenizer.rs:102-111) becomes on a node with several GPUs.  No collective.
 * ids_out must hold one id per input byte (ids_cap >= bytes).  With dropout > 0 the keep decisions hash a sample's index
 * in its shard.  The lowest failing sample of the batch is reported. */
tgx_status tgx_encode_batch_multi(tgx_model *const *models, uint32_t n_models, const uint8_t *text, const uint64_t *offs,
                                  uint64_t n_samples, double dropout, uint64_t seed, uint32_t *ids_out, uint64_t ids_cap,
                                  uint64_t *offs_out, uint64_t *n_tokens);

uint64_t tgx_result_num_samples(const tgx_result *r);
uint64_t tgx_result_num_tokens(const tgx_result *r);
/* Host pointers (copied from the device on first use); valid until tgx_result_free. */
const uint32_t *tgx_result_ids(tgx_result *r);
const uint64_t *tgx_result_offsets(tgx_result *r);
/* The same arrays copied from the device straight into caller-owned memory (a binding's Vec<u32> / numpy
 * array: no intermediate host copy); `cap` = elements available at dst, at least num_tokens resp.
 * num_samples + 1, else TGX_ERR_INVALID. */
tgx_status tgx_result_copy_ids(const tgx_result *r, uint32_t *dst, uint64_t cap);
tgx_status tgx_result_copy_offsets(const tgx_result *r, uint64_t *dst, uint64_t cap);
/* Device pointers of the same arrays (for callers that keep ids in HBM). */
const void *tgx_result_ids_device(const tgx_result *r);
const void *tgx_result_offsets_device(const tgx_result *r);
void tgx_result_free(tgx_result *r);

/* ---- resident corpus: the prune / merge training loops -------------------- */
/* The reference holds `samples: &[&str]` in RAM across all EM / merge passes
 * (src/prune.rs:23, src/merge.rs:33); here the batch is uploaded once and stays
 * in HBM while models change.  Samples are processed longest-first. */
tgx_status tgx_corpus_upload(int device, const uint8_t *text, const uint64_t *offs,
                             uint64_t n_samples, tgx_corpus **out);
void tgx_corpus_free(tgx_corpus *c);
uint64_t tgx_corpus_num_samples(const tgx_corpus *c);
uint64_t tgx_corpus_num_bytes(const tgx_corpus *c);

/* Model::encode over every sample of the corpus; result stays in HBM until the
 * host pointers are asked for. */
tgx_status tgx_encode_corpus(tgx_model *m, tgx_corpus *c, double dropout, uint64_t seed,
                             tgx_result **out);

/* Frequency pass of prune_vocab — src/prune.rs:205-244: freq[id] += 1 for every
 * Viterbi token (dropout 0.0).  freq[vocab_size] is ACCUMULATED into (host). */
tgx_status tgx_count_tokens(tgx_model *m, tgx_corpus *c, uint64_t *freq);

/* Pair scan of merge — src/merge.rs:53-76: adjacent id pairs inside each sample.
 * Returns malloc'd arrays sorted by key = (a << 32) | b; free with tgx_free. */
tgx_status tgx_count_pairs(tgx_model *m, tgx_corpus *c, uint64_t **keys, uint64_t **counts,
                           uint64_t *n_pairs);
/* The same scan, but only the max_pairs MOST FREQUENT pairs come back, ordered by descending count and
 * ascending key among equal counts — the order in which the merge loop consumes them (src/merge.rs:84-126);
 * *n_total = number of distinct pairs.  Saves the copy and the host sort of a table of millions. */
tgx_status tgx_count_pairs_top(tgx_model *m, tgx_corpus *c, uint64_t max_pairs, uint64_t **keys,
                               uint64_t **counts, uint64_t *n_pairs, uint64_t *n_total);

/* run_e_step — src/prune.rs:64-120 over src/model.rs:34-55 + src/lattice.rs:245-333:
 * each sample cut into <= snippet_len-byte snippets, forward/backward in f64
 * log-space, expected[vocab_size] ACCUMULATED into (host), *logz_sum = sum of z.
 * TGX_ERR_Z_NOT_NORMAL where the reference would panic (nothing of the failed
 * pass is added to expected[]).  Tokens of up to 16 bytes, and of up to 32 bytes
 * (vocabularies after `merge`), run the four-snippets-per-wave kernels; longer
 * ones the generic kernel. */
tgx_status tgx_estep(tgx_model *m, tgx_corpus *c, uint64_t snippet_len, double dropout,
                     uint64_t seed, double *expected, double *logz_sum);

void tgx_free(void *p);
/* Scratch and result buffers are recycled through a per-process pool of device buffers (at most a quarter of
 * the device's memory; flushed and retried automatically when an allocation fails).  tgx_pool_trim returns every
 * pooled buffer of `device` (all devices if negative) to the HIP runtime, e.g. before another library needs
 * the memory. */
void tgx_pool_trim(int device);
/* Page-locked host memory for the buffers a caller hands to tgx_encode_batch_host / tgx_corpus_upload and
 * receives ids in: copies from and to such memory are DMA transfers at the link's rate, copies from and to
 * ordinary (pageable) memory go through the driver's staging buffers at about half of it.  NULL (and
 * tgx_last_error) on failure.  (The reference has no counterpart: its tokenizer runs on the host.) */
void *tgx_host_alloc(uint64_t bytes);
void tgx_host_free(void *p);

/* ---- prune host logic (no device needed) ---------------------
 * The O(V) steps of ModelVocabularyPruner between the corpus passes above. */
double tgx_digamma(double x);                                        /* src/prune.rs:322-335 */
/* run_m_step — src/prune.rs:124-170: tokens with expected < 0.5 and !keep are dropped, the
 * others get score = digamma(max(freq, 0.5)) - digamma(sum).  out_* need room for V entries. */
tgx_status tgx_prune_m_step(const double *expected, const uint8_t *keep, uint32_t vocab_size,
                            uint32_t *out_idx, double *out_score, uint32_t *out_n);
/* prune_vocab, first half — src/prune.rs:179-203 over Lattice::nbest(2) (src/lattice.rs:152-238):
 * always_keep[V]; alternatives in CSR form, *alt_ids malloc'd (tgx_free). */
tgx_status tgx_prune_alternatives(const tgx_flat_trie *trie, const uint8_t *bytes, const uint64_t *offs,
                                  const double *scores, uint32_t vocab_size, uint8_t *always_keep,
                                  uint32_t *alt_offs, uint32_t **alt_ids);
/* The same for the vocabulary of a model, over the model's own table (no second trie is built). */
tgx_status tgx_model_prune_alternatives(const tgx_model *m, uint8_t *always_keep, uint32_t *alt_offs, uint32_t **alt_ids);
/* prune_vocab, second half — src/prune.rs:246-318: ids of the pruned vocabulary in its final
 * order (score descending).  out_idx needs room for V entries. */
tgx_status tgx_prune_select(const uint64_t *freq, const uint8_t *keep, const uint8_t *always_keep,
                            const uint32_t *alt_offs, const uint32_t *alt_ids, const double *scores,
                            uint32_t vocab_size, uint64_t n_samples, uint32_t pruned_size,
                            uint32_t *out_idx, uint32_t *out_n);

/* ---- measurement ---------------------------------------------------------- */
/* Per-kernel GPU time of the last pass on this model's stream, measured with
 * hipEvents recorded on that stream around each launch.  names[i] are static
 * strings.  Returns the number of kernels written (<= cap). */
int tgx_last_kernel_times(const tgx_model *m, const char **names, float *ms, int cap);
/* Algorithmic bytes of the last pass, SURVEY.md §8(d): encode N + 4T + 16(S+1). */
uint64_t tgx_last_algorithmic_bytes(const tgx_model *m);
/* waves per CU the last four-samples-per-wave encode launch really had resident (occupancy query for
 * its kernel variant, block size and LDS): a self-check that the launch geometry fits the device. */
uint32_t tgx_last_encode_waves_per_cu(const tgx_model *m);
/* samples the last encode of a vocabulary with tokens of 17..32 bytes had to redo with the
 * two-samples-per-wave kernel because a wave ran out of overflow entries for long matches (0 = none). */
uint64_t tgx_last_encode_redo_samples(const tgx_model *m);
/* samples of the last encode pass that had a block of their own (the long-sample kernel) */
uint64_t tgx_last_encode_long_samples(const tgx_model *m);
/* pieces the last tgx_estep pass cut its snippets into at positions no token match crosses (the lattice factorises
 * there, so expected counts and log Z of the pieces add up to the snippets': csrc/cuts.hip); 0: snippets uncut. */
uint64_t tgx_last_estep_pieces(const tgx_model *m);
/* stretches of text the last tgx_estep pass on estep7_kernel (one walk per position: every trip of 16 .. 64 positions a
 * lattice of its own between two positions no match crosses) could not close within a trip and handed to its redo
 * pass (csrc/estep7.hip); 0: none. */
uint64_t tgx_last_estep_redo(const tgx_model *m);
/* CUs the long-sample kernel had to itself while encode5_kernel ran on the others in the last encode pass
 * (batches of a few hundred MB whose longest samples bound either kernel alone); 0: the kernels ran one after the other. */
uint32_t tgx_last_encode_corun_cus(const tgx_model *m);
/* co-run passes of this model whose host-side wait for the resident blocks of encode5_kernel ran into its 2 ms limit
 * (0 in a healthy setup; after the first one the model launches its two encode kernels one after the other). */
uint32_t tgx_encode_corun_timeouts(const tgx_model *m);
/* distinct score values of the vocabulary as the rows5 encode kernels rank them (0: the model has no 8-byte
 * records — tokens longer than 16 bytes, non-finite scores, more than 65 535 distinct values — or has not
 * encoded yet when it was created for E-step passes), and how many of them the last encode5_kernel launch
 * kept in its block's LDS (the rest are read from L2 by the relaxing lanes). */
uint32_t tgx_model_score_values(const tgx_model *m);
uint32_t tgx_last_encode_hot_values(const tgx_model *m);

/* ---- dropout ---------------------------------------------------------------
 * The reference draws rand::random::<f64>() from an unseeded thread RNG
 * (src/model.rs:48,100), so dropout > 0 is not reproducible there.  This build
 * replaces it with a counter hash of (seed, sample index, byte position, token
 * length) so that runs are reproducible and the CPU oracle and the HIP kernels
 * make identical decisions.  encode keeps a multi-byte match iff dropout < u
 * (model.rs:100); populate_nodes skips it iff u < dropout (model.rs:48). */
static inline double tgx_dropout_u01(uint64_t seed, uint64_t sample, uint64_t pos, uint32_t len) {
    uint64_t x = seed ^ (sample * 0x9E3779B97F4A7C15ULL) ^ (pos * 0xC2B2AE3D27D4EB4FULL) ^
                 ((uint64_t)len * 0x165667B19E3779F9ULL);
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBULL;
    x ^= x >> 31;
    return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}

#ifdef __cplusplus
}
#endif
#endif /* TGX_H */
