/*
 * tgx_oracle.h — CPU restatement of TokenGeeX's Unigram encode / E-step path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle and the timed CPU
 * baseline ("port").  Nothing under tokengeex_amd/ may include, link or call
 * it; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * Every function names the reference file:line (relative to /root/reference)
 * whose behaviour it restates.  The reference is Rust and cannot be compiled
 * in this image (no cargo/rustc), so parity of this restatement is pinned by
 * the reference's own known-answer tests (tests/test_oracle_kat.py):
 *   src/model.rs:208-215, 217-236, 242-252, src/lattice.rs:425-452,
 *   src/tokenizer.rs:443-469
 * plus an independent cross-check against HuggingFace `tokenizers` Unigram
 * (the upstream this code was forked from) on ASCII inputs (tests/golden/).
 */
#ifndef TGX_ORACLE_H
#define TGX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes — same numbering as include/tgx.h */
#define ORC_OK 0
#define ORC_ERR_NO_PATH 4       /* src/lib.rs:223  Error::NoPath(pos,len)  */
#define ORC_ERR_Z_NOT_NORMAL 6  /* src/prune.rs:90-96 panic                  */

typedef struct orc_model orc_model;

/* Model::from — src/model.rs:16-30.  Token i = bytes[offs[i]..offs[i+1]),
 * id = i; later duplicates overwrite earlier ones (src/trie.rs:19). */
orc_model *orc_model_new(const uint8_t *bytes, const uint64_t *offs,
                         const double *scores, uint32_t vocab_size);
void orc_model_free(orc_model *m);
uint32_t orc_model_vocab_size(const orc_model *m);

/* Deterministic stand-in for rand::random::<f64>() (src/model.rs:48,100): the
 * reference draws from an unseeded thread RNG, so there is nothing to be
 * bit-compatible with; both the oracle and the HIP path use this counter
 * hash so that dropout runs are reproducible and comparable.  Returns a value
 * in [0,1) with 53 random bits. */
double orc_dropout_u01(uint64_t seed, uint64_t sample, uint64_t pos, uint32_t len);

/* Model::encode — src/model.rs:59-129.  Returns ORC_OK or ORC_ERR_NO_PATH
 * (err_pos = err_len = n).  *ids is malloc'd (caller frees with orc_free). */
int orc_encode(const orc_model *m, const uint8_t *text, size_t n, double dropout,
               uint64_t seed, uint64_t sample_index, uint32_t **ids, size_t *n_ids,
               size_t *err_pos, size_t *err_len);

/* Tokenizer::encode_ordinary_batch over a packed batch — src/tokenizer.rs:114-123
 * (processors already applied).  Output: flat ids + out_offs[S+1].  On error
 * returns the status of the LOWEST failing sample and its index. */
int orc_encode_batch(const orc_model *m, const uint8_t *text, const uint64_t *offs,
                     uint64_t n_samples, double dropout, uint64_t seed, int n_threads,
                     uint32_t **ids, uint64_t *out_offs, uint64_t *err_sample,
                     uint64_t *err_pos);

/* Model::common_prefix_search — src/model.rs:132-138 + src/trie.rs:44-64.
 * Writes up to cap (id,len) pairs in ascending length; returns the count. */
size_t orc_common_prefix_search(const orc_model *m, const uint8_t *s, size_t n,
                                uint32_t *ids, uint32_t *lens, size_t cap);

/* Model::populate_nodes + Lattice::populate_marginal on ONE snippet —
 * src/model.rs:34-55, src/lattice.rs:78-110, 245-333.  expected[V] is
 * accumulated into; returns z. */
double orc_marginal(const orc_model *m, const uint8_t *snippet, size_t n, double dropout,
                    uint64_t seed, uint64_t sample_index, uint64_t snippet_base,
                    double *expected);

/* run_e_step — src/prune.rs:64-120: every sample cut into <= snippet_len-byte
 * snippets (81920 in the reference), expected[V] accumulated in sample order,
 * logz_sum = sum of z.  Returns ORC_ERR_Z_NOT_NORMAL (and the sample) where the
 * reference would panic. */
int orc_estep(const orc_model *m, const uint8_t *text, const uint64_t *offs,
              uint64_t n_samples, uint64_t snippet_len, double dropout, uint64_t seed,
              int n_threads, double *expected, double *logz_sum, uint64_t *err_sample);

/* frequency pass — src/prune.rs:205-244: freq[id] += 1 over Viterbi ids. */
int orc_count_tokens(const orc_model *m, const uint8_t *text, const uint64_t *offs,
                     uint64_t n_samples, int n_threads, uint64_t *freq,
                     uint64_t *err_sample, uint64_t *err_pos);

/* pair scan — src/merge.rs:53-76: adjacent id pairs within each sample.
 * Output sorted by key = (a << 32) | b; arrays malloc'd. */
int orc_count_pairs(const orc_model *m, const uint8_t *text, const uint64_t *offs,
                    uint64_t n_samples, int n_threads, uint64_t **keys, uint64_t **counts,
                    uint64_t *n_pairs, uint64_t *err_sample, uint64_t *err_pos);

/* SpecialTokenSplitter — src/tokenizer.rs:299-347.  Specials are packed like a
 * vocab.  Emits segments as (start,end,special_index or -1); returns count. */
size_t orc_split_specials(const uint8_t *text, size_t n, const uint8_t *sp_bytes,
                          const uint64_t *sp_offs, uint32_t n_specials, uint64_t *seg_start,
                          uint64_t *seg_end, int32_t *seg_special, size_t cap);

void orc_free(void *p);

/* ---- host half of `prune` (tgx_prune_oracle.c) ---- */
double orc_digamma(double x);                                   /* src/prune.rs:322-335 */
int orc_m_step(const double *expected, const uint8_t *keep, uint32_t vocab_size, uint32_t *out_idx,
               double *out_score, uint32_t *out_n);             /* src/prune.rs:124-170 */
int orc_prune_alternatives(const orc_model *m, const uint8_t *bytes, const uint64_t *offs,
                           const double *scores, uint32_t vocab_size, uint8_t *always_keep,
                           uint32_t *alt_offs, uint32_t **alt_ids); /* src/prune.rs:179-203 */
int orc_prune_select(const uint64_t *freq, const uint8_t *keep, const uint8_t *always_keep,
                     const uint32_t *alt_offs, const uint32_t *alt_ids, const double *scores,
                     uint32_t vocab_size, uint64_t n_samples, uint32_t pruned_size, uint32_t *out_idx,
                     uint32_t *out_n);                          /* src/prune.rs:246-318 */

#ifdef __cplusplus
}
#endif
#endif
