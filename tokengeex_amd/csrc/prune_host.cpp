// Host side of `prune` (reference src/prune.rs) around the E-step / frequency-pass
// kernels: M-step, per-token second-best segmentations, loss-based selection.  These are
// O(V) scalar loops between corpus passes; they need no device.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/tgx.h"
#include "trie_build.h"


tgx_status tgx_set_error(tgx_status st, const char* msg);  // tgx_api.cpp: the message tgx_last_error() returns

namespace {

tgx_status perr(tgx_status st, const char* fmt, ...) {
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return tgx_set_error(st, buf);
}

// digamma — src/prune.rs:322-335
double digamma_ref(double x) {
    double result = 0.0;
    while (x < 7.0) {
        result -= 1.0 / x;
        x += 1.0;
    }
    x -= 1.0 / 2.0;
    const double xx = 1.0 / x;
    const double xx2 = xx * xx;
    const double xx4 = xx2 * xx2;
    result += std::log(x) + (1.0 / 24.0) * xx2 - 7.0 / 960.0 * xx4 + (31.0 / 8064.0) * xx4 * xx2 -
              (127.0 / 30720.0) * xx4 * xx4;
    return result;
}

// ---- the tiny lattice of one token's bytes: src/lattice.rs:13-110 ----
struct LNode {
    uint32_t pos, id, len;
    double score;
    int32_t prev;  // Option<usize>
    double backtrack;
};
struct TokLattice {
    std::vector<LNode> nodes;
    std::vector<std::vector<uint32_t>> begin_nodes, end_nodes;
    uint32_t len = 0;
};

// Lattice::from + Model::populate_nodes(dropout 0.0) — src/lattice.rs:78-110, src/model.rs:34-55
void build_lattice(const tgx::FlatTrie& t, const double* scores, const uint8_t* s, uint32_t n, TokLattice* L) {
    L->len = n;
    L->nodes.clear();
    L->begin_nodes.assign(n + 1, {});
    L->end_nodes.assign(n + 1, {});
    L->nodes.push_back(LNode{0, 0xFFFFFFFEu, 0, 0.0, -1, 0.0});  // BOS, idx 0
    L->nodes.push_back(LNode{n, 0xFFFFFFFFu, 0, 0.0, -1, 0.0});  // EOS, idx 1
    L->end_nodes[0].push_back(0);
    L->begin_nodes[n].push_back(1);
    uint32_t ids[TGX_MAX_TOKEN_LEN + 1], lens[TGX_MAX_TOKEN_LEN + 1];
    for (uint32_t pos = 0; pos < n; pos++) {
        const uint64_t k = tgx::flat_common_prefix_search(t, s + pos, n - pos, ids, lens, TGX_MAX_TOKEN_LEN + 1);
        for (uint64_t j = 0; j < k; j++) {
            const uint32_t idx = (uint32_t)L->nodes.size();
            L->begin_nodes[pos].push_back(idx);
            L->end_nodes[pos + lens[j]].push_back(idx);
            L->nodes.push_back(LNode{pos, ids[j], lens[j], scores[ids[j]], -1, 0.0});
        }
    }
}

// Lattice::viterbi — src/lattice.rs:112-150 (only its side effects on prev / backtrack_score
// matter to nbest; it stops early when a node has no left neighbour)
void viterbi_fill(TokLattice* L) {
    for (uint32_t pos = 0; pos <= L->len; pos++) {
        for (uint32_t r : L->begin_nodes[pos]) {
            L->nodes[r].prev = -1;
            double best_score = 0.0;
            int32_t best_node = -1;
            for (uint32_t l : L->end_nodes[pos]) {
                const double score = L->nodes[l].backtrack + L->nodes[r].score;
                if (best_node < 0 || score > best_score) {
                    best_node = (int32_t)l;
                    best_score = score;
                }
            }
            if (best_node < 0) return;
            L->nodes[r].prev = best_node;
            L->nodes[r].backtrack = best_score;
        }
    }
}

// Hypothesis + BinaryHeap<Hypothesis> — src/lattice.rs:335-378.  The Ord impl never returns
// Equal (fx < other.fx ? Less : Greater); the heap below performs the same comparisons in the
// same order as Rust's std::collections::BinaryHeap (push = sift_up, pop = swap with the last
// element, sift_down_to_bottom, sift_up), so ties resolve the same way.
struct Hyp {
    uint32_t node;
    int32_t next;  // index into the arena, -1 = None
    double fx, gx;
};
struct HypHeap {
    std::vector<uint32_t> data;  // indices into the arena
    const std::vector<Hyp>* arena;
    bool le(uint32_t a, uint32_t b) const { return (*arena)[a].fx < (*arena)[b].fx; }  // a <= b  <=>  cmp != Greater
    void sift_up(size_t start, size_t pos) {
        const uint32_t elt = data[pos];
        while (pos > start) {
            const size_t parent = (pos - 1) / 2;
            if (le(elt, data[parent])) break;
            data[pos] = data[parent];
            pos = parent;
        }
        data[pos] = elt;
    }
    void push(uint32_t h) {
        data.push_back(h);
        sift_up(0, data.size() - 1);
    }
    uint32_t pop() {
        uint32_t item = data.back();
        data.pop_back();
        if (!data.empty()) {
            std::swap(item, data[0]);
            // sift_down_to_bottom(0)
            const size_t end = data.size();
            size_t pos = 0;
            const uint32_t elt = data[0];
            size_t child = 1;
            while (child + 1 < end) {
                if (le(data[child], data[child + 1])) child += 1;
                data[pos] = data[child];
                pos = child;
                child = 2 * pos + 1;
            }
            if (child == end - 1) {
                data[pos] = data[child];
                pos = child;
            }
            data[pos] = elt;
            sift_up(0, pos);
        }
        return item;
    }
};

// Lattice::nbest(2) — src/lattice.rs:152-238.  paths[k] = node indices of the k-th best path.
void nbest2(TokLattice* L, std::vector<std::vector<uint32_t>>* paths) {
    paths->clear();
    std::vector<Hyp> arena;
    HypHeap agenda;
    agenda.arena = &arena;
    arena.push_back(Hyp{1, -1, L->nodes[1].score, L->nodes[1].score});  // EOS
    agenda.push(0);
    viterbi_fill(L);
    while (!agenda.data.empty()) {
        const uint32_t top = agenda.pop();
        const uint32_t node = arena[top].node;
        if (L->nodes[node].id == L->nodes[0].id) {  // reached BOS
            std::vector<uint32_t> hyp;
            int32_t nxt = arena[top].next;
            while (arena[nxt].next >= 0) {
                hyp.push_back(arena[nxt].node);
                nxt = arena[nxt].next;
            }
            paths->push_back(hyp);
            if (paths->size() == 2) return;
        } else {
            for (uint32_t l : L->end_nodes[L->nodes[node].pos]) {
                const double top_gx = arena[top].gx;
                arena.push_back(Hyp{l, (int32_t)top, L->nodes[l].backtrack + top_gx, L->nodes[l].score + top_gx});
                agenda.push((uint32_t)arena.size() - 1);
            }
            if (agenda.data.size() > 100000) {  // k_max_agenda_size: keep the min(512, n * 10) = 20 best
                HypHeap fresh;
                fresh.arena = &arena;
                for (int i = 0; i < 20; i++) fresh.push(agenda.pop());
                agenda.data.swap(fresh.data);
            }
        }
    }
}

}  // namespace

extern "C" {

double tgx_digamma(double x) { return digamma_ref(x); }

// run_m_step — src/prune.rs:124-170.  out_idx / out_score need room for V entries.
tgx_status tgx_prune_m_step(const double* expected, const uint8_t* keep, uint32_t vocab_size, uint32_t* out_idx,
                            double* out_score, uint32_t* out_n) {
    if (!expected || !keep || !out_idx || !out_score || !out_n) return perr(TGX_ERR_INVALID, "tgx_prune_m_step: NULL argument");
    const double threshold = 0.5;  // EXPECTED_FREQUENCY_THRESHOLD
    uint32_t n = 0;
    for (uint32_t i = 0; i < vocab_size; i++) {
        if (expected[i] < threshold && !keep[i]) continue;
        out_idx[n] = i;
        out_score[n] = std::fmax(expected[i], threshold);  // f64::max
        n++;
    }
    double sum = 0.0;
    for (uint32_t i = 0; i < n; i++) sum += out_score[i];
    const double logsum = digamma_ref(sum);
    // (digamma of half a million values: 12 of the M-step's 14 ms at 500 000 tokens — on the host's threads; every value is
    // computed exactly as before, only by another thread)
    unsigned hw = std::thread::hardware_concurrency();
    const uint32_t n_threads = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>({(uint64_t)(hw ? hw : 1u), 8, (uint64_t)n / 16384 + 1}));
    std::vector<uint32_t> first_bad(n_threads, 0xFFFFFFFFu);
    auto work = [&](uint32_t t) {
        const uint32_t a = (uint32_t)((uint64_t)n * t / n_threads), b = (uint32_t)((uint64_t)n * (t + 1) / n_threads);
        for (uint32_t i = a; i < b; i++) {
            out_score[i] = digamma_ref(out_score[i]) - logsum;
            if ((std::isnan(out_score[i]) || std::isinf(out_score[i])) && first_bad[t] == 0xFFFFFFFFu) first_bad[t] = i;
        }
    };
    {
        std::vector<std::thread> th;
        for (uint32_t t = 1; t < n_threads; t++) th.emplace_back(work, t);
        work(0);
        for (auto& x : th) x.join();
    }
    for (uint32_t t = 0; t < n_threads; t++)
        if (first_bad[t] != 0xFFFFFFFFu) {  // the reference panics (src/prune.rs:152-158)
            const uint32_t i = first_bad[t];
            return perr(TGX_ERR_INVALID, "M-step: score of token %u is not finite (expected frequency %.17g, sum %.17g)", out_idx[i], std::fmax(expected[out_idx[i]], threshold), sum);
        }
    *out_n = n;
    return TGX_OK;
}

// First half of prune_vocab — src/prune.rs:179-203: how would each token be segmented without
// itself?  always_keep[V]; alternatives as CSR (alt_offs[V+1], *alt_ids malloc'd, free with tgx_free).
tgx_status tgx_prune_alternatives(const tgx_flat_trie* trie, const uint8_t* bytes, const uint64_t* offs,
                                  const double* scores, uint32_t vocab_size, uint8_t* always_keep,
                                  uint32_t* alt_offs, uint32_t** alt_ids) {
    if (!trie) return perr(TGX_ERR_INVALID, "tgx_prune_alternatives: NULL argument");
    return (tgx_status)tgx_prune_alternatives_flat(&trie->flat, bytes, offs, scores, vocab_size, always_keep, alt_offs, alt_ids);
}

// the same over a table that already exists (tgx_model_prune_alternatives, tgx_api.cpp: the model's own trie)
int tgx_prune_alternatives_flat(const tgx::FlatTrie* flat_trie, const uint8_t* bytes, const uint64_t* offs,
                                       const double* scores, uint32_t vocab_size, uint8_t* always_keep,
                                       uint32_t* alt_offs, uint32_t** alt_ids) {
    if (!flat_trie || !offs || !scores || !always_keep || !alt_offs || !alt_ids) return perr(TGX_ERR_INVALID, "tgx_prune_alternatives: NULL argument");
    // The reference walks the vocabulary serially; the tokens are independent (each gets its own lattice), so
    // contiguous id ranges go to host threads and the per-range lists are concatenated in id order: the
    // result does not depend on the number of threads.  (1.0 s at 500 000 tokens on one core.)
    uint32_t n_threads = std::thread::hardware_concurrency();
    if (const char* e = getenv("TGX_HOST_THREADS")) n_threads = (uint32_t)atoi(e);
    n_threads = std::max(1u, std::min({n_threads, 32u, vocab_size / 2048u + 1u}));
    std::vector<std::vector<uint32_t>> part(n_threads);
    auto work = [&](uint32_t t) {
        const uint32_t lo = (uint32_t)((uint64_t)vocab_size * t / n_threads);
        const uint32_t hi = (uint32_t)((uint64_t)vocab_size * (t + 1) / n_threads);
        std::vector<uint32_t>& flat_alts = part[t];
        TokLattice L;
        std::vector<std::vector<uint32_t>> paths;
        for (uint32_t id = lo; id < hi; id++) {
            alt_offs[id] = (uint32_t)flat_alts.size();  // relative to the range; rebased below
            always_keep[id] = 1;
            const uint32_t n = (uint32_t)(offs[id + 1] - offs[id]);
            build_lattice(*flat_trie, scores, bytes + offs[id], n, &L);
            nbest2(&L, &paths);
            if (paths.size() > 1 && paths[0].size() > 1) always_keep[id] = 0;
            if (paths.size() > 1 && paths[0].size() == 1)
                for (uint32_t nd : paths[1]) flat_alts.push_back(L.nodes[nd].id);
        }
    };
    {
        std::vector<std::thread> pool;
        for (uint32_t t = 1; t < n_threads; t++) pool.emplace_back(work, t);
        work(0);
        for (std::thread& th : pool) th.join();
    }
    size_t total = 0;
    for (uint32_t t = 0; t < n_threads; t++) total += part[t].size();
    *alt_ids = (uint32_t*)malloc(sizeof(uint32_t) * (total ? total : 1));
    if (!*alt_ids) return perr(TGX_ERR_INVALID, "tgx_prune_alternatives: out of host memory");
    size_t base = 0;
    for (uint32_t t = 0; t < n_threads; t++) {
        const uint32_t lo = (uint32_t)((uint64_t)vocab_size * t / n_threads);
        const uint32_t hi = (uint32_t)((uint64_t)vocab_size * (t + 1) / n_threads);
        for (uint32_t id = lo; id < hi; id++) alt_offs[id] += (uint32_t)base;
        if (!part[t].empty()) memcpy(*alt_ids + base, part[t].data(), sizeof(uint32_t) * part[t].size());
        base += part[t].size();
    }
    alt_offs[vocab_size] = (uint32_t)total;
    return TGX_OK;
}

// Second half of prune_vocab — src/prune.rs:246-318: loss of removing each token, keep the
// pruned_size best, order by score.  out_idx needs room for V entries.  (sort_unstable_by in the
// reference leaves the order of equal keys unspecified; std::stable_sort here.)
tgx_status tgx_prune_select(const uint64_t* freq, const uint8_t* keep, const uint8_t* always_keep,
                            const uint32_t* alt_offs, const uint32_t* alt_ids, const double* scores,
                            uint32_t vocab_size, uint64_t n_samples, uint32_t pruned_size, uint32_t* out_idx,
                            uint32_t* out_n) {
    if (!freq || !keep || !always_keep || !alt_offs || !scores || !out_idx || !out_n) return perr(TGX_ERR_INVALID, "tgx_prune_select: NULL argument");
    uint64_t total = 0;
    for (uint32_t i = 0; i < vocab_size; i++) total += freq[i];
    const double sum_f = (double)total;
    const double logsum = std::log(sum_f);
    std::vector<std::pair<uint32_t, double>> candidates;
    std::vector<uint32_t> pruned;
    for (uint32_t id = 0; id < vocab_size; id++) {
        const uint32_t na = alt_offs[id + 1] - alt_offs[id];
        if (keep[id]) {
            pruned.push_back(id);
            continue;
        }
        if (freq[id] == 0 && !always_keep[id]) {
            continue;
        } else if (na == 0) {
            pruned.push_back(id);
        } else if (freq[id] != 0) {
            const double f = (double)freq[id];
            const double logprob = std::log(f) - logsum;
            // `alternatives.len() - 1` in the reference is the length of the OUTER vector (= V)
            const double alt_logsum = std::log(sum_f + f * (double)(vocab_size - 1));
            double alt_logprob = 0.0;
            for (uint32_t k = alt_offs[id]; k < alt_offs[id + 1]; k++)
                alt_logprob += std::log((double)freq[alt_ids[k]] + f) - alt_logsum;
            const double loss = (f / (double)n_samples) * (logprob - alt_logprob);
            if (!std::isnormal(loss))  // the reference panics (src/prune.rs:287-293)
                return perr(TGX_ERR_INVALID, "prune: loss of token %u is not a normal number (frequency %.0f, loss %.17g)", id, f, loss);
            candidates.emplace_back(id, loss);
        }
    }
    std::stable_sort(candidates.begin(), candidates.end(),
                     [](const std::pair<uint32_t, double>& a, const std::pair<uint32_t, double>& b) { return a.second > b.second; });
    for (const auto& c : candidates) {
        if (pruned.size() == pruned_size) break;  // '==' as in the reference: no cut if already above
        pruned.push_back(c.first);
    }
    std::stable_sort(pruned.begin(), pruned.end(), [scores](uint32_t a, uint32_t b) { return scores[a] > scores[b]; });
    for (size_t i = 0; i < pruned.size(); i++) out_idx[i] = pruned[i];
    *out_n = (uint32_t)pruned.size();
    return TGX_OK;
}

}  // extern "C"
