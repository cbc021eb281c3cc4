/* TEST INFRASTRUCTURE ONLY — CPU restatement of the host half of `prune`
 * (reference src/prune.rs run_m_step / prune_vocab / digamma, src/lattice.rs viterbi / nbest).
 * Written as the reference writes it: a node vector, per-position begin/end index vectors,
 * a binary heap of reference-counted hypotheses.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may use this.
 *
 * Pinning: the reference holds no golden vector for this code (its test_digamma only prints);
 * digamma is checked against scipy.special.digamma, the rest against hand-worked cases in
 * tests/test_prune_cpu.py.  sort_unstable_by tie order is unspecified upstream; a stable
 * order is used here. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "tgx_oracle.h"

/* src/prune.rs:322-335 */
double orc_digamma(double x) {
    double result = 0.0;
    while (x < 7.0) {
        result -= 1.0 / x;
        x += 1.0;
    }
    x -= 1.0 / 2.0;
    double xx = 1.0 / x;
    double xx2 = xx * xx;
    double xx4 = xx2 * xx2;
    result += log(x) + (1.0 / 24.0) * xx2 - 7.0 / 960.0 * xx4 + (31.0 / 8064.0) * xx4 * xx2 -
              (127.0 / 30720.0) * xx4 * xx4;
    return result;
}

/* src/prune.rs:124-170; returns 0, or -1 where the reference panics */
int orc_m_step(const double *expected, const uint8_t *keep, uint32_t vocab_size, uint32_t *out_idx,
               double *out_score, uint32_t *out_n) {
    uint32_t n = 0;
    for (uint32_t i = 0; i < vocab_size; i++) {
        double freq = expected[i];
        if (freq < 0.5 && !keep[i]) continue;
        out_idx[n] = i;
        out_score[n] = freq != freq ? 0.5 : (freq > 0.5 ? freq : 0.5); /* f64::max */
        n++;
    }
    double sum = 0.0;
    for (uint32_t i = 0; i < n; i++) sum += out_score[i];
    double logsum = orc_digamma(sum);
    int bad = 0;
    for (uint32_t i = 0; i < n; i++) {
        out_score[i] = orc_digamma(out_score[i]) - logsum;
        if (isnan(out_score[i]) || isinf(out_score[i])) bad = 1;
    }
    *out_n = n;
    return bad ? -1 : 0;
}

/* ---- src/lattice.rs:13-110: Node, Lattice ---- */
typedef struct {
    size_t pos;
    uint32_t token_id;
    size_t token_len;
    double score;
    long prev; /* Option<usize>, -1 = None */
    double backtrack_score;
} lnode;
typedef struct {
    size_t *v, n, cap;
} idxvec;
static void iv_push(idxvec *a, size_t x) {
    if (a->n == a->cap) {
        a->cap = a->cap ? 2 * a->cap : 4;
        a->v = (size_t *)realloc(a->v, sizeof(size_t) * a->cap);
    }
    a->v[a->n++] = x;
}
typedef struct {
    size_t len;
    lnode *nodes;
    size_t n_nodes, cap_nodes;
    idxvec *begin_nodes, *end_nodes;
} plattice;

static size_t lat_push(plattice *L, size_t pos, uint32_t id, size_t len, double score) {
    if (L->n_nodes == L->cap_nodes) {
        L->cap_nodes = L->cap_nodes ? 2 * L->cap_nodes : 64;
        L->nodes = (lnode *)realloc(L->nodes, sizeof(lnode) * L->cap_nodes);
    }
    lnode nd = {pos, id, len, score, -1, 0.0};
    L->nodes[L->n_nodes] = nd;
    return L->n_nodes++;
}
static void lat_from(plattice *L, size_t len) {
    memset(L, 0, sizeof(*L));
    L->len = len;
    L->begin_nodes = (idxvec *)calloc(len + 1, sizeof(idxvec));
    L->end_nodes = (idxvec *)calloc(len + 1, sizeof(idxvec));
    size_t bos = lat_push(L, 0, UINT32_MAX - 1, 0, 0.0);
    size_t eos = lat_push(L, len, UINT32_MAX, 0, 0.0);
    iv_push(&L->end_nodes[0], bos);
    iv_push(&L->begin_nodes[len], eos);
}
static void lat_free(plattice *L) {
    for (size_t i = 0; i <= L->len; i++) {
        free(L->begin_nodes[i].v);
        free(L->end_nodes[i].v);
    }
    free(L->begin_nodes);
    free(L->end_nodes);
    free(L->nodes);
}
/* src/lattice.rs:105-110 */
static void lat_insert(plattice *L, size_t pos, uint32_t id, size_t len, double score) {
    size_t idx = lat_push(L, pos, id, len, score);
    iv_push(&L->begin_nodes[pos], idx);
    iv_push(&L->end_nodes[pos + len], idx);
}

/* src/lattice.rs:112-150 (return value unused by nbest) */
static void lat_viterbi(plattice *L) {
    for (size_t pos = 0; pos <= L->len; pos++) {
        for (size_t bi = 0; bi < L->begin_nodes[pos].n; bi++) {
            size_t r = L->begin_nodes[pos].v[bi];
            L->nodes[r].prev = -1;
            double best_score = 0.0;
            long best_node = -1;
            for (size_t ei = 0; ei < L->end_nodes[pos].n; ei++) {
                size_t l = L->end_nodes[pos].v[ei];
                double score = L->nodes[l].backtrack_score + L->nodes[r].score;
                if (best_node < 0 || score > best_score) {
                    best_node = (long)l;
                    best_score = score;
                }
            }
            if (best_node < 0) return;
            L->nodes[r].prev = best_node;
            L->nodes[r].backtrack_score = best_score;
        }
    }
}

/* src/lattice.rs:335-378: Hypothesis; Ord::cmp = fx < other.fx ? Less : Greater */
typedef struct hyp {
    size_t node_idx;
    struct hyp *next;
    double fx, gx;
} hyp;
/* Rust std BinaryHeap<T> (max-heap): push = append + sift_up; pop = take last, swap with
 * root, sift_down_to_bottom, sift_up.  `a <= b` under the Ord above is fx_a < fx_b. */
typedef struct {
    hyp **d;
    size_t n, cap;
} heap;
static int hyp_le(const hyp *a, const hyp *b) { return a->fx < b->fx; }
static void heap_sift_up(heap *h, size_t start, size_t pos) {
    hyp *e = h->d[pos];
    while (pos > start) {
        size_t parent = (pos - 1) / 2;
        if (hyp_le(e, h->d[parent])) break;
        h->d[pos] = h->d[parent];
        pos = parent;
    }
    h->d[pos] = e;
}
static void heap_push(heap *h, hyp *x) {
    if (h->n == h->cap) {
        h->cap = h->cap ? 2 * h->cap : 64;
        h->d = (hyp **)realloc(h->d, sizeof(hyp *) * h->cap);
    }
    h->d[h->n++] = x;
    heap_sift_up(h, 0, h->n - 1);
}
static hyp *heap_pop(heap *h) {
    hyp *item = h->d[--h->n];
    if (h->n > 0) {
        hyp *t = item;
        item = h->d[0];
        h->d[0] = t;
        size_t end = h->n, pos = 0, child = 1;
        hyp *e = h->d[0];
        while (child + 1 < end) {
            if (hyp_le(h->d[child], h->d[child + 1])) child++;
            h->d[pos] = h->d[child];
            pos = child;
            child = 2 * pos + 1;
        }
        if (child + 1 == end) {
            h->d[pos] = h->d[child];
            pos = child;
        }
        h->d[pos] = e;
        heap_sift_up(h, 0, pos);
    }
    return item;
}

/* src/lattice.rs:152-238 with n = 2.  paths: node indices; returns number of paths found. */
static size_t lat_nbest2(plattice *L, size_t *path0, size_t *n0, size_t *path1, size_t *n1) {
    heap agenda = {0, 0, 0};
    hyp **all = NULL; /* arena so everything can be freed */
    size_t n_all = 0, cap_all = 0;
#define NEWHYP(ptr, ni, nx, f, g)                                             \
    do {                                                                      \
        ptr = (hyp *)malloc(sizeof(hyp));                                     \
        ptr->node_idx = (ni); ptr->next = (nx); ptr->fx = (f); ptr->gx = (g); \
        if (n_all == cap_all) {                                               \
            cap_all = cap_all ? 2 * cap_all : 256;                            \
            all = (hyp **)realloc(all, sizeof(hyp *) * cap_all);              \
        }                                                                     \
        all[n_all++] = ptr;                                                   \
    } while (0)
    size_t found = 0;
    hyp *h0;
    NEWHYP(h0, 1, NULL, L->nodes[1].score, L->nodes[1].score);
    heap_push(&agenda, h0);
    lat_viterbi(L);
    while (agenda.n > 0) {
        hyp *top = heap_pop(&agenda);
        size_t node_idx = top->node_idx;
        if (L->nodes[node_idx].token_id == L->nodes[0].token_id) {
            size_t *dst = found == 0 ? path0 : path1, k = 0;
            hyp *next = top->next;
            while (next->next != NULL) {
                dst[k++] = next->node_idx;
                next = next->next;
            }
            if (found == 0) *n0 = k; else *n1 = k;
            found++;
            if (found == 2) break;
        } else {
            size_t node_pos = L->nodes[node_idx].pos;
            for (size_t ei = 0; ei < L->end_nodes[node_pos].n; ei++) {
                size_t l = L->end_nodes[node_pos].v[ei];
                hyp *nh;
                NEWHYP(nh, l, top, L->nodes[l].backtrack_score + top->gx, L->nodes[l].score + top->gx);
                heap_push(&agenda, nh);
            }
            if (agenda.n > 100000) { /* k_max_agenda_size; keep min(512, n*10) = 20 */
                heap fresh = {0, 0, 0};
                for (int i = 0; i < 20; i++) heap_push(&fresh, heap_pop(&agenda));
                free(agenda.d);
                agenda = fresh;
            }
        }
    }
    for (size_t i = 0; i < n_all; i++) free(all[i]);
    free(all);
    free(agenda.d);
    return found;
#undef NEWHYP
}

/* src/prune.rs:179-203.  alt_ids is malloc'd (orc_free). */
int orc_prune_alternatives(const orc_model *m, const uint8_t *bytes, const uint64_t *offs,
                           const double *scores, uint32_t vocab_size, uint8_t *always_keep,
                           uint32_t *alt_offs, uint32_t **alt_ids) {
    size_t cap = 1024, n_alt = 0;
    uint32_t *alts = (uint32_t *)malloc(sizeof(uint32_t) * cap);
    uint32_t ids[256], lens[256];
    for (uint32_t id = 0; id < vocab_size; id++) {
        const uint8_t *s = bytes + offs[id];
        size_t n = (size_t)(offs[id + 1] - offs[id]);
        alt_offs[id] = (uint32_t)n_alt;
        always_keep[id] = 1;
        plattice L;
        lat_from(&L, n);
        /* src/model.rs:34-55 populate_nodes with dropout 0.0 */
        for (size_t pos = 0; pos < n; pos++) {
            size_t k = orc_common_prefix_search(m, s + pos, n - pos, ids, lens, 256);
            for (size_t j = 0; j < k; j++) lat_insert(&L, pos, ids[j], lens[j], scores[ids[j]]);
        }
        size_t *p0 = (size_t *)malloc(sizeof(size_t) * (n + 2)), *p1 = (size_t *)malloc(sizeof(size_t) * (n + 2));
        size_t n0 = 0, n1 = 0;
        size_t found = lat_nbest2(&L, p0, &n0, p1, &n1);
        if (found > 1 && n0 > 1) always_keep[id] = 0;
        if (found > 1 && n0 == 1) {
            for (size_t j = 0; j < n1; j++) {
                if (n_alt == cap) {
                    cap *= 2;
                    alts = (uint32_t *)realloc(alts, sizeof(uint32_t) * cap);
                }
                alts[n_alt++] = L.nodes[p1[j]].token_id;
            }
        }
        free(p0);
        free(p1);
        lat_free(&L);
    }
    alt_offs[vocab_size] = (uint32_t)n_alt;
    *alt_ids = alts;
    return 0;
}

typedef struct {
    uint32_t id;
    double key;
    uint32_t seq;
} sortrec;
static int cmp_desc(const void *a, const void *b) {
    const sortrec *x = (const sortrec *)a, *y = (const sortrec *)b;
    if (x->key > y->key) return -1;
    if (x->key < y->key) return 1;
    return x->seq < y->seq ? -1 : (x->seq > y->seq ? 1 : 0); /* stable */
}

/* src/prune.rs:246-318; returns 0, or -1 where the reference panics */
int orc_prune_select(const uint64_t *freq, const uint8_t *keep, const uint8_t *always_keep,
                     const uint32_t *alt_offs, const uint32_t *alt_ids, const double *scores,
                     uint32_t vocab_size, uint64_t n_samples, uint32_t pruned_size, uint32_t *out_idx,
                     uint32_t *out_n) {
    uint64_t total = 0;
    for (uint32_t i = 0; i < vocab_size; i++) total += freq[i];
    double sum_token_frequencies = (double)total;
    double logsum_token_frequencies = log(sum_token_frequencies);
    sortrec *cand = (sortrec *)malloc(sizeof(sortrec) * (vocab_size ? vocab_size : 1));
    sortrec *pruned = (sortrec *)malloc(sizeof(sortrec) * (vocab_size ? vocab_size : 1));
    uint32_t n_cand = 0, n_pruned = 0;
    int bad = 0;
    for (uint32_t id = 0; id < vocab_size; id++) {
        uint32_t n_alt = alt_offs[id + 1] - alt_offs[id];
        if (keep[id]) {
            pruned[n_pruned].id = id; pruned[n_pruned].seq = n_pruned; n_pruned++;
            continue;
        }
        if (freq[id] == 0 && !always_keep[id]) {
            continue;
        } else if (n_alt == 0) {
            pruned[n_pruned].id = id; pruned[n_pruned].seq = n_pruned; n_pruned++;
        } else if (freq[id] != 0) {
            double f = (double)freq[id];
            double logprob = log(f) - logsum_token_frequencies;
            /* alternatives.len() is the outer vector's length = vocab_size */
            double alt_logsum = log(sum_token_frequencies + f * (double)(vocab_size - 1));
            double alt_logprob = 0.0;
            for (uint32_t k = alt_offs[id]; k < alt_offs[id + 1]; k++)
                alt_logprob += log((double)freq[alt_ids[k]] + f) - alt_logsum;
            double loss = (f / (double)n_samples) * (logprob - alt_logprob);
            if (!isnormal(loss)) bad = 1;
            cand[n_cand].id = id; cand[n_cand].key = loss; cand[n_cand].seq = n_cand; n_cand++;
        }
    }
    if (!bad) {
        qsort(cand, n_cand, sizeof(sortrec), cmp_desc);
        for (uint32_t i = 0; i < n_cand; i++) {
            if (n_pruned == pruned_size) break;
            pruned[n_pruned].id = cand[i].id; pruned[n_pruned].seq = n_pruned; n_pruned++;
        }
        for (uint32_t i = 0; i < n_pruned; i++) pruned[i].key = scores[pruned[i].id];
        qsort(pruned, n_pruned, sizeof(sortrec), cmp_desc);
        for (uint32_t i = 0; i < n_pruned; i++) out_idx[i] = pruned[i].id;
        *out_n = n_pruned;
    }
    free(cand);
    free(pruned);
    return bad ? -1 : 0;
}
