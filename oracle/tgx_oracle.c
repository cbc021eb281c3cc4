/*
 * tgx_oracle.c — CPU restatement of the reference's Unigram hot path.
 * TEST INFRASTRUCTURE ONLY (see tgx_oracle.h).  Plain C11, no dependencies.
 *
 * It follows the reference's data structures and loop orders literally so that
 * (a) results are the reference's results, including tie-breaks and quirks, and
 * (b) timing it is a fair "port" CPU baseline: one FNV-1a hash map per trie
 * node (src/trie.rs:75-78), a 32-byte DP node per position allocated per call
 * (src/model.rs:72-81), per-node lattice with begin/end buckets and per-node
 * alpha/beta vectors (src/lattice.rs:50-64, 255-256).
 *
 * Build with -O2/-O3 but WITHOUT -ffast-math and with -ffp-contract=off: the
 * f64 operation order below is the reference's.
 */
#include "tgx_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ trie -- */
/* src/trie.rs:75-78  Node { data: Option<Data>, children: HashMap<u8, Node, Fnv> } */
typedef struct obucket obucket;
typedef struct onode {
    uint32_t has_data; /* Option<(TokenID, u32)> */
    uint32_t id;
    uint32_t len;
    uint32_t cap; /* bucket count: 0 or a power of two */
    uint32_t count;
    obucket *buckets;
} onode;
struct obucket {
    uint8_t used;
    uint8_t key;
    onode child; /* values live inline in the table, as in Rust's HashMap */
};

static inline uint64_t fnv1a_u8(uint8_t b) {
    /* fnv::FnvHasher over a single byte key */
    return (0xcbf29ce484222325ULL ^ (uint64_t)b) * 0x100000001b3ULL;
}

static onode *node_get(const onode *n, uint8_t key) {
    if (n->cap == 0) return NULL;
    uint32_t mask = n->cap - 1;
    uint32_t i = (uint32_t)fnv1a_u8(key) & mask;
    for (;;) {
        obucket *b = &n->buckets[i];
        if (!b->used) return NULL;
        if (b->key == key) return &b->child;
        i = (i + 1) & mask;
    }
}

static void node_grow(onode *n) {
    uint32_t ncap = n->cap ? n->cap * 2 : 4;
    obucket *nb = (obucket *)calloc(ncap, sizeof(obucket));
    uint32_t mask = ncap - 1;
    for (uint32_t j = 0; j < n->cap; j++) {
        if (!n->buckets[j].used) continue;
        uint32_t i = (uint32_t)fnv1a_u8(n->buckets[j].key) & mask;
        while (nb[i].used) i = (i + 1) & mask;
        nb[i] = n->buckets[j];
    }
    free(n->buckets);
    n->buckets = nb;
    n->cap = ncap;
}

/* children.entry(b).or_default() — src/trie.rs:16 */
static onode *node_entry(onode *n, uint8_t key) {
    onode *c = node_get(n, key);
    if (c) return c;
    if ((uint64_t)(n->count + 1) * 8 > (uint64_t)n->cap * 7) node_grow(n);
    uint32_t mask = n->cap - 1;
    uint32_t i = (uint32_t)fnv1a_u8(key) & mask;
    while (n->buckets[i].used) i = (i + 1) & mask;
    n->buckets[i].used = 1;
    n->buckets[i].key = key;
    memset(&n->buckets[i].child, 0, sizeof(onode));
    n->count++;
    return &n->buckets[i].child;
}

static void node_destroy(onode *n) {
    for (uint32_t j = 0; j < n->cap; j++)
        if (n->buckets[j].used) node_destroy(&n->buckets[j].child);
    free(n->buckets);
}

/* Trie::push — src/trie.rs:12-20: walk/create, then OVERWRITE data. */
static void trie_push(onode *root, const uint8_t *tok, size_t len, uint32_t id) {
    onode *n = root;
    for (size_t i = 0; i < len; i++) n = node_entry(n, tok[i]);
    n->has_data = 1;
    n->id = id;
    n->len = (uint32_t)len;
}

/* ----------------------------------------------------------------- model -- */
/* src/model.rs:8-12 (token_to_ids is not on the hot path and is omitted). */
struct orc_model {
    uint32_t vocab_size;
    double *score; /* vocab[id].score */
    onode root;
};

orc_model *orc_model_new(const uint8_t *bytes, const uint64_t *offs, const double *scores,
                         uint32_t vocab_size) {
    orc_model *m = (orc_model *)calloc(1, sizeof(orc_model));
    m->vocab_size = vocab_size;
    m->score = (double *)malloc(sizeof(double) * (vocab_size ? vocab_size : 1));
    memcpy(m->score, scores, sizeof(double) * vocab_size);
    /* src/model.rs:20-23: ids are vocab positions, pushed in order */
    for (uint32_t id = 0; id < vocab_size; id++)
        trie_push(&m->root, bytes + offs[id], (size_t)(offs[id + 1] - offs[id]), id);
    return m;
}

void orc_model_free(orc_model *m) {
    if (!m) return;
    node_destroy(&m->root);
    free(m->score);
    free(m);
}

uint32_t orc_model_vocab_size(const orc_model *m) { return m->vocab_size; }
void orc_free(void *p) { free(p); }

/* our own reproducible replacement for the reference's unseeded RNG */
double orc_dropout_u01(uint64_t seed, uint64_t sample, uint64_t pos, uint32_t len) {
    uint64_t x = seed ^ (sample * 0x9E3779B97F4A7C15ULL) ^ (pos * 0xC2B2AE3D27D4EB4FULL) ^
                 ((uint64_t)len * 0x165667B19E3779F9ULL);
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBULL;
    x ^= x >> 31;
    return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}

/* TrieIterator::next — src/trie.rs:51-63, fused into the callers' loops:
 * consume one byte, descend, stop at the first missing child, report data at
 * every terminal node passed.  Callers keep (node, depth). */

size_t orc_common_prefix_search(const orc_model *m, const uint8_t *s, size_t n, uint32_t *ids,
                                uint32_t *lens, size_t cap) {
    const onode *node = &m->root;
    size_t out = 0;
    for (size_t i = 0; i < n; i++) {
        node = node_get(node, s[i]);
        if (!node) break;
        if (node->has_data) {
            if (out < cap) {
                ids[out] = node->id;
                lens[out] = node->len;
            }
            out++;
        }
    }
    return out;
}

/* ---------------------------------------------------------------- encode -- */
/* src/model.rs:63-68: struct Node { id: u32, score: f64, start: Option<usize> } */
typedef struct {
    uint64_t has_start;
    size_t start;
    double score;
    uint32_t id;
    uint32_t _pad;
} dpnode;

int orc_encode(const orc_model *m, const uint8_t *text, size_t n, double dropout, uint64_t seed,
               uint64_t sample_index, uint32_t **ids_out, size_t *n_ids, size_t *err_pos,
               size_t *err_len) {
    /* src/model.rs:72-81 */
    dpnode *dp = (dpnode *)malloc(sizeof(dpnode) * (n + 1));
    for (size_t i = 0; i <= n; i++) {
        dp[i].has_start = 0;
        dp[i].start = 0;
        dp[i].score = 0.0;
        dp[i].id = 0;
    }
    dp[0].has_start = 1;
    dp[0].start = 0;

    for (size_t pos = 0; pos < n; pos++) {
        if (!dp[pos].has_start) continue; /* src/model.rs:85-87 */
        const onode *node = &m->root;
        for (size_t j = pos; j < n; j++) { /* src/model.rs:92-94 → trie.rs:51-63 */
            node = node_get(node, text[j]);
            if (!node) break;
            if (!node->has_data) continue;
            uint32_t id = node->id;
            size_t len = node->len;
            dpnode *tgt = &dp[pos + len];
            double score = dp[pos].score + m->score[id]; /* src/model.rs:98 */
            /* src/model.rs:100-101, short-circuit order kept */
            int keep = (dropout <= 0.0) || (len <= 1) ||
                       (dropout < orc_dropout_u01(seed, sample_index, pos, (uint32_t)len));
            if (keep && (!tgt->has_start || score > tgt->score)) {
                tgt->id = id;
                tgt->score = score;
                tgt->has_start = 1;
                tgt->start = pos;
            }
        }
    }

    /* src/model.rs:113-128 */
    size_t cap = n / 2 + 1, cnt = 0;
    uint32_t *ids = (uint32_t *)malloc(sizeof(uint32_t) * cap);
    size_t pos = n;
    while (pos > 0) {
        if (!dp[pos].has_start) {
            if (err_pos) *err_pos = pos;
            if (err_len) *err_len = n;
            free(ids);
            free(dp);
            *ids_out = NULL;
            *n_ids = 0;
            return ORC_ERR_NO_PATH;
        }
        if (cnt == cap) {
            cap *= 2;
            ids = (uint32_t *)realloc(ids, sizeof(uint32_t) * cap);
        }
        ids[cnt++] = dp[pos].id;
        pos = dp[pos].start;
    }
    for (size_t i = 0; i < cnt / 2; i++) { /* ids.reverse() */
        uint32_t t = ids[i];
        ids[i] = ids[cnt - 1 - i];
        ids[cnt - 1 - i] = t;
    }
    free(dp);
    *ids_out = ids;
    *n_ids = cnt;
    return ORC_OK;
}

/* ------------------------------------------------------- batch machinery -- */
/* The reference's batch loops are rayon par_chunks with chunk =
 * max(1, n / threads / f) (src/task.rs:134-137); here: pthreads pulling chunk
 * indices from an atomic counter. */
typedef void (*chunk_fn)(void *ctx, uint64_t lo, uint64_t hi, int tid);
typedef struct {
    chunk_fn fn;
    void *ctx;
    uint64_t n, chunk;
    atomic_ullong next;
    int tid;
} par_shared;
typedef struct {
    par_shared *sh;
    int tid;
} par_arg;

static void *par_worker(void *p) {
    par_arg *a = (par_arg *)p;
    par_shared *sh = a->sh;
    for (;;) {
        uint64_t c = atomic_fetch_add(&sh->next, 1);
        uint64_t lo = c * sh->chunk;
        if (lo >= sh->n) break;
        uint64_t hi = lo + sh->chunk;
        if (hi > sh->n) hi = sh->n;
        sh->fn(sh->ctx, lo, hi, a->tid);
    }
    return NULL;
}

static void par_chunks(uint64_t n, int n_threads, uint64_t f, chunk_fn fn, void *ctx) {
    if (n_threads < 1) n_threads = 1;
    par_shared sh;
    sh.fn = fn;
    sh.ctx = ctx;
    sh.n = n;
    sh.chunk = n / (uint64_t)n_threads / f;
    if (sh.chunk < 1) sh.chunk = 1;
    atomic_init(&sh.next, 0);
    if (n_threads == 1) {
        par_arg a = {&sh, 0};
        par_worker(&a);
        return;
    }
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    par_arg *args = (par_arg *)malloc(sizeof(par_arg) * (size_t)n_threads);
    for (int t = 0; t < n_threads; t++) {
        args[t].sh = &sh;
        args[t].tid = t;
        pthread_create(&th[t], NULL, par_worker, &args[t]);
    }
    for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    free(th);
    free(args);
}

/* lowest failing sample wins (the reference leaves it unspecified,
 * src/tokenizer.rs:107-110) */
typedef struct {
    pthread_mutex_t mu;
    int status;
    uint64_t sample, pos;
} err_slot;

static void err_record(err_slot *e, int status, uint64_t sample, uint64_t pos) {
    pthread_mutex_lock(&e->mu);
    if (e->status == ORC_OK || sample < e->sample) {
        e->status = status;
        e->sample = sample;
        e->pos = pos;
    }
    pthread_mutex_unlock(&e->mu);
}

/* ---- encode_batch ---- */
typedef struct {
    const orc_model *m;
    const uint8_t *text;
    const uint64_t *offs;
    double dropout;
    uint64_t seed;
    uint32_t **per_ids;
    size_t *per_n;
    err_slot err;
} encb_ctx;

static void encb_chunk(void *p, uint64_t lo, uint64_t hi, int tid) {
    (void)tid;
    encb_ctx *c = (encb_ctx *)p;
    for (uint64_t s = lo; s < hi; s++) {
        size_t ep = 0, el = 0;
        int st = orc_encode(c->m, c->text + c->offs[s], (size_t)(c->offs[s + 1] - c->offs[s]),
                            c->dropout, c->seed, s, &c->per_ids[s], &c->per_n[s], &ep, &el);
        if (st != ORC_OK) err_record(&c->err, st, s, ep);
    }
}

int orc_encode_batch(const orc_model *m, const uint8_t *text, const uint64_t *offs,
                     uint64_t n_samples, double dropout, uint64_t seed, int n_threads,
                     uint32_t **ids, uint64_t *out_offs, uint64_t *err_sample,
                     uint64_t *err_pos) {
    encb_ctx c;
    c.m = m;
    c.text = text;
    c.offs = offs;
    c.dropout = dropout;
    c.seed = seed;
    c.per_ids = (uint32_t **)calloc(n_samples ? n_samples : 1, sizeof(uint32_t *));
    c.per_n = (size_t *)calloc(n_samples ? n_samples : 1, sizeof(size_t));
    pthread_mutex_init(&c.err.mu, NULL);
    c.err.status = ORC_OK;
    c.err.sample = c.err.pos = 0;
    par_chunks(n_samples, n_threads, 4, encb_chunk, &c);
    int st = c.err.status;
    *ids = NULL;
    if (st == ORC_OK) {
        uint64_t total = 0;
        for (uint64_t s = 0; s < n_samples; s++) {
            out_offs[s] = total;
            total += c.per_n[s];
        }
        out_offs[n_samples] = total;
        uint32_t *flat = (uint32_t *)malloc(sizeof(uint32_t) * (total ? total : 1));
        for (uint64_t s = 0; s < n_samples; s++)
            if (c.per_n[s]) memcpy(flat + out_offs[s], c.per_ids[s], sizeof(uint32_t) * c.per_n[s]);
        *ids = flat;
    } else {
        if (err_sample) *err_sample = c.err.sample;
        if (err_pos) *err_pos = c.err.pos;
    }
    for (uint64_t s = 0; s < n_samples; s++) free(c.per_ids[s]);
    free(c.per_ids);
    free(c.per_n);
    pthread_mutex_destroy(&c.err.mu);
    return st;
}

/* --------------------------------------------------------------- lattice -- */
/* src/lattice.rs:13-26 (prev / backtrack_score are Viterbi-only, not used by
 * populate_marginal, and are left out) */
typedef struct {
    size_t pos;
    uint32_t token_id;
    size_t token_len;
    double score;
} lnode;

typedef struct {
    size_t *d;
    uint32_t n, cap;
} ivec;

static inline void ivec_push(ivec *v, size_t x) {
    if (v->n == v->cap) {
        v->cap = v->cap ? v->cap * 2 : 16; /* VecPool::with_capacity(_, 16), src/prune.rs:80 */
        v->d = (size_t *)realloc(v->d, sizeof(size_t) * v->cap);
    }
    v->d[v->n++] = x;
}

/* src/lattice.rs:50-64.  begin/end buckets are kept across snippets and only
 * reset (n = 0), which is what VecPool recycling amounts to. */
typedef struct {
    const uint8_t *sentence;
    size_t len;
    ivec *begin_nodes, *end_nodes;
    size_t buckets_cap;
    lnode *nodes;
    size_t n_nodes, nodes_cap;
    double *alpha, *beta;
    size_t ab_cap;
} lattice;

static void lattice_destroy(lattice *l) {
    for (size_t i = 0; i < l->buckets_cap; i++) {
        free(l->begin_nodes[i].d);
        free(l->end_nodes[i].d);
    }
    free(l->begin_nodes);
    free(l->end_nodes);
    free(l->nodes);
    free(l->alpha);
    free(l->beta);
}

static inline void lattice_push_node(lattice *l, size_t pos, uint32_t id, size_t len, double score) {
    if (l->n_nodes == l->nodes_cap) {
        l->nodes_cap = l->nodes_cap ? l->nodes_cap * 2 : 1024;
        l->nodes = (lnode *)realloc(l->nodes, sizeof(lnode) * l->nodes_cap);
    }
    lnode *nd = &l->nodes[l->n_nodes++];
    nd->pos = pos;
    nd->token_id = id;
    nd->token_len = len;
    nd->score = score;
}

/* Lattice::from — src/lattice.rs:78-103 */
static void lattice_from(lattice *l, const uint8_t *sentence, size_t len) {
    l->sentence = sentence;
    l->len = len;
    l->n_nodes = 0;
    if (len + 1 > l->buckets_cap) {
        l->begin_nodes = (ivec *)realloc(l->begin_nodes, sizeof(ivec) * (len + 1));
        l->end_nodes = (ivec *)realloc(l->end_nodes, sizeof(ivec) * (len + 1));
        memset(l->begin_nodes + l->buckets_cap, 0, sizeof(ivec) * (len + 1 - l->buckets_cap));
        memset(l->end_nodes + l->buckets_cap, 0, sizeof(ivec) * (len + 1 - l->buckets_cap));
        l->buckets_cap = len + 1;
    }
    for (size_t i = 0; i <= len; i++) {
        l->begin_nodes[i].n = 0;
        l->end_nodes[i].n = 0;
    }
    lattice_push_node(l, 0, UINT32_MAX - 1, 0, 0.0); /* BOS, idx 0 */
    lattice_push_node(l, len, UINT32_MAX, 0, 0.0);   /* EOS, idx 1 */
    ivec_push(&l->end_nodes[0], 0);
    ivec_push(&l->begin_nodes[len], 1);
}

/* Lattice::insert — src/lattice.rs:105-110 */
static inline void lattice_insert(lattice *l, size_t pos, uint32_t id, size_t len, double score) {
    size_t idx = l->n_nodes;
    ivec_push(&l->begin_nodes[pos], idx);
    ivec_push(&l->end_nodes[pos + len], idx);
    lattice_push_node(l, pos, id, len, score);
}

/* Model::populate_nodes — src/model.rs:34-55 */
static void populate_nodes(const orc_model *m, lattice *l, double dropout, uint64_t seed,
                           uint64_t sample_index, uint64_t snippet_base) {
    const uint8_t *input = l->sentence;
    size_t n = l->len;
    for (size_t pos = 0; pos < n; pos++) {
        const onode *node = &m->root;
        for (size_t j = pos; j < n; j++) {
            node = node_get(node, input[j]);
            if (!node) break;
            if (!node->has_data) continue;
            uint32_t id = node->id;
            uint32_t len = node->len;
            /* src/model.rs:48 */
            if (len > 1 && dropout > 0.0 &&
                orc_dropout_u01(seed, sample_index, snippet_base + pos, len) < dropout)
                continue;
            lattice_insert(l, pos, id, len, m->score[id]);
        }
    }
}

/* log_sum_exp — src/lattice.rs:321-333 */
static inline double log_sum_exp(double x, double y, int init_mode) {
    if (init_mode) return y;
    double vmin, vmax;
    if (x > y) {
        vmin = y;
        vmax = x;
    } else {
        vmin = x;
        vmax = y;
    }
    const double k_minus_log_epsilon = 50.0;
    if (vmax > vmin + k_minus_log_epsilon) return vmax;
    return vmax + log(exp(vmin - vmax) + 1.0);
}

/* Lattice::populate_marginal — src/lattice.rs:245-312 */
static double populate_marginal(lattice *l, double *expected) {
    size_t len = l->len, num_nodes = l->n_nodes;
    if (num_nodes > l->ab_cap) {
        l->ab_cap = num_nodes * 2;
        l->alpha = (double *)realloc(l->alpha, sizeof(double) * l->ab_cap);
        l->beta = (double *)realloc(l->beta, sizeof(double) * l->ab_cap);
    }
    double *alpha = l->alpha, *beta = l->beta;
    for (size_t i = 0; i < num_nodes; i++) alpha[i] = 0.0, beta[i] = 0.0;

    for (size_t pos = 0; pos <= len; pos++) { /* :259-272 */
        const ivec *bn = &l->begin_nodes[pos], *en = &l->end_nodes[pos];
        for (uint32_t r = 0; r < bn->n; r++) {
            size_t rid = bn->d[r];
            for (uint32_t q = 0; q < en->n; q++) {
                size_t lid = en->d[q];
                alpha[rid] = log_sum_exp(alpha[rid], l->nodes[lid].score + alpha[lid],
                                         lid == en->d[0]);
            }
        }
    }
    for (size_t pos = len + 1; pos-- > 0;) { /* :275-287 */
        const ivec *bn = &l->begin_nodes[pos], *en = &l->end_nodes[pos];
        for (uint32_t q = 0; q < en->n; q++) {
            size_t lid = en->d[q];
            for (uint32_t r = 0; r < bn->n; r++) {
                size_t rid = bn->d[r];
                beta[lid] = log_sum_exp(beta[lid], l->nodes[rid].score + beta[rid],
                                        rid == bn->d[0]);
            }
        }
    }
    double z = alpha[1]; /* eos_idx = 1, :290-291 */
    for (size_t pos = 0; pos < len; pos++) { /* :295-309 */
        const ivec *bn = &l->begin_nodes[pos];
        for (uint32_t r = 0; r < bn->n; r++) {
            size_t idx = bn->d[r];
            uint32_t id = l->nodes[idx].token_id;
            double score = l->nodes[idx].score;
            double a = alpha[idx], b = beta[idx];
            double total = a + score + b - z; /* ((a + score) + b) - z */
            expected[id] += exp(total);
        }
    }
    return z;
}

double orc_marginal(const orc_model *m, const uint8_t *snippet, size_t n, double dropout,
                    uint64_t seed, uint64_t sample_index, uint64_t snippet_base,
                    double *expected) {
    lattice l;
    memset(&l, 0, sizeof(l));
    lattice_from(&l, snippet, n);
    populate_nodes(m, &l, dropout, seed, sample_index, snippet_base);
    double z = populate_marginal(&l, expected);
    lattice_destroy(&l);
    return z;
}

/* ---- run_e_step: src/prune.rs:64-120 ---- */
typedef struct {
    const orc_model *m;
    const uint8_t *text;
    const uint64_t *offs;
    uint64_t snippet_len;
    double dropout;
    uint64_t seed;
    int n_threads;
    double **local_expected; /* per thread */
    double *local_z;
    lattice *lat;
    err_slot err;
} estep_ctx;

static void estep_chunk(void *p, uint64_t lo, uint64_t hi, int tid) {
    estep_ctx *c = (estep_ctx *)p;
    lattice *l = &c->lat[tid];
    double *ex = c->local_expected[tid];
    for (uint64_t s = lo; s < hi; s++) {
        const uint8_t *sample = c->text + c->offs[s];
        uint64_t n = c->offs[s + 1] - c->offs[s];
        /* sample.as_bytes().chunks(MAX_SAMPLE_LENGTH) — src/prune.rs:83 */
        for (uint64_t base = 0; base < n; base += c->snippet_len) {
            uint64_t sn = n - base < c->snippet_len ? n - base : c->snippet_len;
            lattice_from(l, sample + base, (size_t)sn);
            populate_nodes(c->m, l, c->dropout, c->seed, s, base);
            double z = populate_marginal(l, ex);
            /* !z.is_normal() panics — src/prune.rs:90-96 */
            if (fpclassify(z) != FP_NORMAL) err_record(&c->err, ORC_ERR_Z_NOT_NORMAL, s, base);
            c->local_z[tid] += z;
        }
    }
}

int orc_estep(const orc_model *m, const uint8_t *text, const uint64_t *offs, uint64_t n_samples,
              uint64_t snippet_len, double dropout, uint64_t seed, int n_threads,
              double *expected, double *logz_sum, uint64_t *err_sample) {
    if (n_threads < 1) n_threads = 1;
    uint32_t V = m->vocab_size;
    estep_ctx c;
    memset(&c, 0, sizeof(c));
    c.m = m;
    c.text = text;
    c.offs = offs;
    c.snippet_len = snippet_len ? snippet_len : 81920; /* 8192 * 10, src/prune.rs:75 */
    c.dropout = dropout;
    c.seed = seed;
    c.n_threads = n_threads;
    c.local_expected = (double **)calloc((size_t)n_threads, sizeof(double *));
    c.local_z = (double *)calloc((size_t)n_threads, sizeof(double));
    c.lat = (lattice *)calloc((size_t)n_threads, sizeof(lattice));
    for (int t = 0; t < n_threads; t++)
        c.local_expected[t] = (double *)calloc(V ? V : 1, sizeof(double));
    pthread_mutex_init(&c.err.mu, NULL);
    c.err.status = ORC_OK;
    par_chunks(n_samples, n_threads, 8, estep_chunk, &c); /* f = 8, src/prune.rs:66 */
    /* fixed thread order (the reference merges in completion order) */
    double zsum = 0.0;
    for (int t = 0; t < n_threads; t++) {
        for (uint32_t i = 0; i < V; i++) expected[i] += c.local_expected[t][i];
        zsum += c.local_z[t];
        free(c.local_expected[t]);
        lattice_destroy(&c.lat[t]);
    }
    if (logz_sum) *logz_sum = zsum;
    if (c.err.status != ORC_OK && err_sample) *err_sample = c.err.sample;
    free(c.local_expected);
    free(c.local_z);
    free(c.lat);
    pthread_mutex_destroy(&c.err.mu);
    return c.err.status;
}

/* ---- frequency pass: src/prune.rs:205-244 ---- */
typedef struct {
    const orc_model *m;
    const uint8_t *text;
    const uint64_t *offs;
    uint64_t **local; /* per-thread usize[V] */
    err_slot err;
} freq_ctx;

static void freq_chunk(void *p, uint64_t lo, uint64_t hi, int tid) {
    freq_ctx *c = (freq_ctx *)p;
    uint64_t *f = c->local[tid];
    for (uint64_t s = lo; s < hi; s++) {
        uint32_t *ids = NULL;
        size_t n_ids = 0, ep = 0, el = 0;
        int st = orc_encode(c->m, c->text + c->offs[s], (size_t)(c->offs[s + 1] - c->offs[s]), 0.0,
                            0, s, &ids, &n_ids, &ep, &el);
        if (st != ORC_OK) {
            err_record(&c->err, st, s, ep);
            continue;
        }
        for (size_t i = 0; i < n_ids; i++) f[ids[i]] += 1;
        free(ids);
    }
}

int orc_count_tokens(const orc_model *m, const uint8_t *text, const uint64_t *offs,
                     uint64_t n_samples, int n_threads, uint64_t *freq, uint64_t *err_sample,
                     uint64_t *err_pos) {
    if (n_threads < 1) n_threads = 1;
    uint32_t V = m->vocab_size;
    freq_ctx c;
    c.m = m;
    c.text = text;
    c.offs = offs;
    c.local = (uint64_t **)calloc((size_t)n_threads, sizeof(uint64_t *));
    for (int t = 0; t < n_threads; t++) c.local[t] = (uint64_t *)calloc(V ? V : 1, sizeof(uint64_t));
    pthread_mutex_init(&c.err.mu, NULL);
    c.err.status = ORC_OK;
    par_chunks(n_samples, n_threads, 2, freq_chunk, &c); /* f = 2, src/prune.rs:206 */
    for (int t = 0; t < n_threads; t++) {
        for (uint32_t i = 0; i < V; i++) freq[i] += c.local[t][i];
        free(c.local[t]);
    }
    free(c.local);
    if (c.err.status != ORC_OK) {
        if (err_sample) *err_sample = c.err.sample;
        if (err_pos) *err_pos = c.err.pos;
    }
    pthread_mutex_destroy(&c.err.mu);
    return c.err.status;
}

/* ---- pair scan: src/merge.rs:53-76 ---- */
typedef struct {
    uint64_t *keys, *vals; /* key+1 stored so that 0 = empty */
    uint64_t cap, count;
} pairmap;

static inline uint64_t mix64(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

static void pairmap_add(pairmap *pm, uint64_t key, uint64_t delta);

static void pairmap_grow(pairmap *pm) {
    pairmap n;
    n.cap = pm->cap ? pm->cap * 2 : 1024;
    n.count = 0;
    n.keys = (uint64_t *)calloc(n.cap, sizeof(uint64_t));
    n.vals = (uint64_t *)calloc(n.cap, sizeof(uint64_t));
    for (uint64_t i = 0; i < pm->cap; i++)
        if (pm->keys[i]) pairmap_add(&n, pm->keys[i] - 1, pm->vals[i]);
    free(pm->keys);
    free(pm->vals);
    *pm = n;
}

static void pairmap_add(pairmap *pm, uint64_t key, uint64_t delta) {
    if ((pm->count + 1) * 2 > pm->cap) pairmap_grow(pm);
    uint64_t mask = pm->cap - 1, i = mix64(key) & mask;
    for (;;) {
        if (pm->keys[i] == 0) {
            pm->keys[i] = key + 1;
            pm->vals[i] = delta;
            pm->count++;
            return;
        }
        if (pm->keys[i] == key + 1) {
            pm->vals[i] += delta;
            return;
        }
        i = (i + 1) & mask;
    }
}

typedef struct {
    const orc_model *m;
    const uint8_t *text;
    const uint64_t *offs;
    pairmap *local;
    err_slot err;
} pair_ctx;

static void pair_chunk(void *p, uint64_t lo, uint64_t hi, int tid) {
    pair_ctx *c = (pair_ctx *)p;
    pairmap *pm = &c->local[tid];
    for (uint64_t s = lo; s < hi; s++) {
        uint32_t *ids = NULL;
        size_t n_ids = 0, ep = 0, el = 0;
        int st = orc_encode(c->m, c->text + c->offs[s], (size_t)(c->offs[s + 1] - c->offs[s]), 0.0,
                            0, s, &ids, &n_ids, &ep, &el);
        if (st != ORC_OK) { /* the reference unwraps (panics), src/merge.rs:58 */
            err_record(&c->err, st, s, ep);
            continue;
        }
        for (size_t i = 1; i < n_ids; i++) /* src/merge.rs:60-63 */
            pairmap_add(pm, ((uint64_t)ids[i - 1] << 32) | ids[i], 1);
        free(ids);
    }
}

typedef struct {
    uint64_t k, v;
} kv;
static int kv_cmp(const void *a, const void *b) {
    uint64_t x = ((const kv *)a)->k, y = ((const kv *)b)->k;
    return x < y ? -1 : (x > y ? 1 : 0);
}

int orc_count_pairs(const orc_model *m, const uint8_t *text, const uint64_t *offs,
                    uint64_t n_samples, int n_threads, uint64_t **keys, uint64_t **counts,
                    uint64_t *n_pairs, uint64_t *err_sample, uint64_t *err_pos) {
    if (n_threads < 1) n_threads = 1;
    pair_ctx c;
    c.m = m;
    c.text = text;
    c.offs = offs;
    c.local = (pairmap *)calloc((size_t)n_threads, sizeof(pairmap));
    pthread_mutex_init(&c.err.mu, NULL);
    c.err.status = ORC_OK;
    par_chunks(n_samples, n_threads, 4, pair_chunk, &c); /* f = 4, src/merge.rs:39 */
    pairmap total;
    memset(&total, 0, sizeof(total));
    for (int t = 0; t < n_threads; t++) {
        for (uint64_t i = 0; i < c.local[t].cap; i++)
            if (c.local[t].keys[i]) pairmap_add(&total, c.local[t].keys[i] - 1, c.local[t].vals[i]);
        free(c.local[t].keys);
        free(c.local[t].vals);
    }
    free(c.local);
    kv *arr = (kv *)malloc(sizeof(kv) * (total.count ? total.count : 1));
    uint64_t k = 0;
    for (uint64_t i = 0; i < total.cap; i++)
        if (total.keys[i]) {
            arr[k].k = total.keys[i] - 1;
            arr[k].v = total.vals[i];
            k++;
        }
    qsort(arr, (size_t)k, sizeof(kv), kv_cmp);
    *keys = (uint64_t *)malloc(sizeof(uint64_t) * (k ? k : 1));
    *counts = (uint64_t *)malloc(sizeof(uint64_t) * (k ? k : 1));
    for (uint64_t i = 0; i < k; i++) {
        (*keys)[i] = arr[i].k;
        (*counts)[i] = arr[i].v;
    }
    *n_pairs = k;
    free(arr);
    free(total.keys);
    free(total.vals);
    if (c.err.status != ORC_OK) {
        if (err_sample) *err_sample = c.err.sample;
        if (err_pos) *err_pos = c.err.pos;
    }
    pthread_mutex_destroy(&c.err.mu);
    return c.err.status;
}

/* ---- SpecialTokenSplitter: src/tokenizer.rs:299-347 ---- */
size_t orc_split_specials(const uint8_t *text, size_t n, const uint8_t *sp_bytes,
                          const uint64_t *sp_offs, uint32_t n_specials, uint64_t *seg_start,
                          uint64_t *seg_end, int32_t *seg_special, size_t cap) {
    for (uint32_t k = 0; k < n_specials; k++)
        if (sp_offs[k + 1] == sp_offs[k]) return (size_t)-1; /* would never advance */
    size_t out = 0, cursor = 0;
    while (cursor < n) { /* one Iterator::next per turn */
        const uint8_t *input = text + cursor;
        size_t rem = n - cursor;
        int emitted = 0;
        for (size_t i = 0; i < rem && !emitted; i++) {
            if ((input[i] & 0xC0) == 0x80) continue; /* char_indices: char starts only */
            for (uint32_t k = 0; k < n_specials; k++) {
                size_t sl = (size_t)(sp_offs[k + 1] - sp_offs[k]);
                if (sl <= rem - i && memcmp(input + i, sp_bytes + sp_offs[k], sl) == 0) {
                    if (out < cap) {
                        seg_start[out] = cursor;
                        seg_end[out] = cursor + (i > 0 ? i : sl);
                        seg_special[out] = i > 0 ? -1 : (int32_t)k;
                    }
                    out++;
                    cursor += i > 0 ? i : sl;
                    emitted = 1;
                    break;
                }
            }
        }
        if (!emitted) {
            if (out < cap) {
                seg_start[out] = cursor;
                seg_end[out] = n;
                seg_special[out] = -1;
            }
            out++;
            cursor = n;
        }
    }
    return out;
}
