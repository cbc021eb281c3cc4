#include "trie_build.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <queue>
#include <thread>

namespace tgx {

static double host_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
HostPhases::HostPhases(const char* scope_) : scope(scope_) {
    static int enabled = -1;
    if (enabled < 0) {
        const char* k = getenv("TGX_KNOBS");
        const char* e = getenv("TGX_HOST_TIMES");
        enabled = (k && *k && *k != '0' && e && *e && *e != '0') ? 1 : 0;
    }
    on = enabled == 1;
    t0 = last = on ? host_now() : 0.0;
}
void HostPhases::mark(const char* what) {
    if (!on) return;
    const double t = host_now();
    fprintf(stderr, "[tgx host] %-28s %-28s %8.2f ms\n", scope, what, (t - last) * 1e3);
    last = t;
}
HostPhases::~HostPhases() {
    if (on) fprintf(stderr, "[tgx host] %-28s %-28s %8.2f ms\n", scope, "(total)", (host_now() - t0) * 1e3);
}

namespace {

// std::sort on `threads` chunks at once, then pairwise std::inplace_merge (the levels' merges at once too): the two big
// sorts of build_flat_trie (tokens, expansion order) were 67 of its 124 ms at 500 000 tokens
template <typename It, typename Cmp>
void parallel_sort(It begin, It end, Cmp cmp) {
    const size_t n = (size_t)(end - begin);
    unsigned hw = std::thread::hardware_concurrency();
    size_t parts = 1;
    while (parts < 8 && parts * 2 <= (hw ? hw : 1u) && n / (parts * 2) >= 32768) parts *= 2;
    if (parts == 1) {
        std::sort(begin, end, cmp);
        return;
    }
    std::vector<size_t> cut(parts + 1);
    for (size_t i = 0; i <= parts; i++) cut[i] = n * i / parts;
    {
        std::vector<std::thread> th;
        for (size_t i = 1; i < parts; i++) th.emplace_back([&, i]() { std::sort(begin + (long)cut[i], begin + (long)cut[i + 1], cmp); });
        std::sort(begin, begin + (long)cut[1], cmp);
        for (auto& t : th) t.join();
    }
    for (size_t width = 1; width < parts; width *= 2) {
        std::vector<std::thread> th;
        for (size_t i = 0; i + width < parts; i += 2 * width) {
            const size_t a = cut[i], b = cut[i + width], c = cut[std::min(parts, i + 2 * width)];
            if (i == 0) continue;
            th.emplace_back([&, a, b, c]() { std::inplace_merge(begin + (long)a, begin + (long)b, begin + (long)c, cmp); });
        }
        std::inplace_merge(begin, begin + (long)cut[width], begin + (long)cut[std::min(parts, 2 * width)], cmp);
        for (auto& t : th) t.join();
    }
}

// fn(begin, end) over [0, n) in contiguous ranges on up to 8 host threads (ranges of at least `grain`)
template <typename Fn>
void parallel_ranges(size_t n, size_t grain, Fn fn) {
    unsigned hw = std::thread::hardware_concurrency();
    const size_t parts = std::max<size_t>(1, std::min<size_t>({(size_t)(hw ? hw : 1u), (size_t)8, n / std::max<size_t>(grain, 1) + 1}));
    if (parts == 1) {
        fn((size_t)0, n);
        return;
    }
    std::vector<std::thread> th;
    for (size_t i = 1; i < parts; i++) th.emplace_back([&, i]() { fn(n * i / parts, n * (i + 1) / parts); });
    fn((size_t)0, n / parts);
    for (auto& t : th) t.join();
}

struct Edge {
    uint32_t parent;
    uint32_t child;
    uint8_t byte;
};

struct BlockAlloc {
    std::vector<uint64_t> used;      // 4 words per 256-slot block
    std::vector<uint64_t> base_used; // 4 words per block: bit x = some node's children live at (block << 8 | x) ^ byte
    std::vector<uint16_t> free_cnt;  // per block
    uint32_t head1 = 0;              // first block that may have a free slot
    uint32_t headk = 0;              // first block worth trying for multi-child nodes

    uint32_t n_blocks() const { return (uint32_t)free_cnt.size(); }
    void add_block() {
        used.insert(used.end(), 4, 0ULL);
        // low bytes 0xFE and 0xFF are never the base of a node with children: the label-checked 8-byte
        // records (build_trie8) point leaves at 0xFE and give unused slot t the label (t ^ 0xFF) & 0xFF
        base_used.insert(base_used.end(), {0ULL, 0ULL, 0ULL, 3ULL << 62});
        free_cnt.push_back(256);
    }
    bool is_used(uint32_t t) const { return (used[t >> 6] >> (t & 63)) & 1ULL; }
    void mark(uint32_t t) {
        used[t >> 6] |= 1ULL << (t & 63);
        free_cnt[t >> 8]--;
    }
    int first_free_in_block(uint32_t b) const {
        for (int w = 0; w < 4; w++) {
            uint64_t inv = ~used[(size_t)b * 4 + w];
            if (inv) return w * 64 + __builtin_ctzll(inv);
        }
        return -1;
    }
};

}  // namespace

void build_flat_trie(const uint8_t* bytes, const uint64_t* offs, const double* scores,
                     uint32_t vocab_size, FlatTrie* out) {
    // 1. pointer-free trie: nodes are indices.  The tokens are visited in lexicographic order (ties: ascending
    // id, so that the last duplicate wins), which turns insertion into "keep the common prefix with the previous
    // token, append the rest": no edge map, sequential memory.  (A hash map of (node, byte) edges took half of
    // the build at 500 000 tokens, nearly every probe a cache miss.)
    struct Key {
        uint64_t prefix;  // first 8 bytes, big-endian, zero-padded
        uint32_t id;
    };
    HostPhases hp("build_flat_trie");
    std::vector<Key> order;
    order.reserve(vocab_size);
    uint32_t max_len = 0;
    for (uint32_t id = 0; id < vocab_size; id++) {
        const uint64_t b = offs[id], e = offs[id + 1];
        if (e == b) continue;  // empty token: root payload, never matched
        uint64_t k = 0;
        for (uint64_t i = 0; i < 8; i++) k = (k << 8) | (b + i < e ? bytes[b + i] : 0u);
        order.push_back(Key{k, id});
        max_len = std::max<uint32_t>(max_len, (uint32_t)(e - b));
    }
    parallel_sort(order.begin(), order.end(), [&](const Key& x, const Key& y) {
        if (x.prefix != y.prefix) return x.prefix < y.prefix;
        const uint64_t lx = offs[x.id + 1] - offs[x.id], ly = offs[y.id + 1] - offs[y.id];
        const int c = std::memcmp(bytes + offs[x.id], bytes + offs[y.id], (size_t)std::min(lx, ly));
        if (c != 0) return c < 0;
        if (lx != ly) return lx < ly;
        return x.id < y.id;
    });
    hp.mark("sort tokens");
    uint64_t total_bytes = vocab_size ? offs[vocab_size] - offs[0] : 0;
    std::vector<Edge> created;  // in creation (depth-first) order
    created.reserve(total_bytes / 4 + 16);
    std::vector<uint32_t> node_tok(1, kNoToken);  // node 0 = root
    std::vector<uint32_t> path(max_len + 1, 0);    // path[d] = node of the current token's first d bytes
    const uint8_t* prev = nullptr;
    uint64_t prev_len = 0;
    for (const Key& key : order) {
        const uint8_t* cur = bytes + offs[key.id];
        const uint64_t len = offs[key.id + 1] - offs[key.id];
        uint64_t l = 0;
        const uint64_t m = std::min(len, prev_len);
        while (l < m && cur[l] == prev[l]) l++;
        uint32_t node = path[l];
        for (uint64_t i = l; i < len; i++) {
            const uint32_t nxt = (uint32_t)node_tok.size();
            node_tok.push_back(kNoToken);
            created.push_back(Edge{node, nxt, cur[i]});
            path[i + 1] = nxt;
            node = nxt;
        }
        node_tok[node] = key.id;  // overwrite: the last duplicate wins
        prev = cur;
        prev_len = len;
    }
    uint32_t n_nodes = (uint32_t)node_tok.size();
    hp.mark("nodes");

    // 2. children in CSR form, sorted by (parent, byte): the children of a node were created in ascending byte
    // order, so a stable counting sort by parent is all it takes.
    std::vector<uint32_t> row(n_nodes + 1, 0);
    for (const Edge& e : created) row[e.parent + 1]++;
    for (uint32_t i = 0; i < n_nodes; i++) row[i + 1] += row[i];
    std::vector<Edge> edges(created.size());
    {
        std::vector<uint32_t> fill(row.begin(), row.end() - 1);
        for (const Edge& e : created) edges[fill[e.parent]++] = e;
    }
    created = std::vector<Edge>();

    // 3. hottest-first slot assignment.  A node's weight is the probability mass of the
    // tokens below it (sum of exp(score)), a proxy for how often a walk passes through it;
    // nodes are expanded in descending weight, so the children blocks of the most
    // visited nodes land in the first slots (the kernels keep those in LDS).
    std::vector<double> weight(n_nodes, 0.0);
    for (uint32_t i = 0; i < n_nodes; i++)
        if (node_tok[i] != kNoToken) {
            const double w = std::exp(scores[node_tok[i]]);
            weight[i] = (w == w && w < 1e300) ? w : 0.0;
        }
    {
        std::vector<uint32_t> parent(n_nodes, 0);
        for (const Edge& e : edges) parent[e.child] = e.parent;
        for (uint32_t i = n_nodes; i-- > 1;) weight[parent[i]] += weight[i];  // children are created after parents
    }
    hp.mark("csr + weights");
    BlockAlloc ba;
    ba.add_block();
    ba.mark(0);  // root
    ba.free_cnt[0] = 0;  // block 0 holds the root only: leaves of the 8-byte records point into it (base 0xFE)
    std::vector<uint32_t> slot(n_nodes, 0), base(n_nodes, 0);
    // Expansion order: descending weight, the older node first among equals.  A parent weighs at least as
    // much as any of its children and was created before them, so one sort gives an order in which every
    // node comes after its parent (the order a priority queue over the frontier would pop them in).
    // (sorted as contiguous (key, node) pairs: weights are finite and >= 0, so their bit patterns order like
    // the values; the key is the complement for a descending order)
    std::vector<std::pair<uint64_t, uint32_t>> keyed(n_nodes);
    for (uint32_t i = 0; i < n_nodes; i++) {
        const double w = weight[i] + 0.0;  // -0.0 -> +0.0
        uint64_t bits;
        std::memcpy(&bits, &w, 8);
        keyed[i] = {~bits, i};
    }
    parallel_sort(keyed.begin(), keyed.end(), std::less<std::pair<uint64_t, uint32_t>>());
    std::vector<uint32_t> expand(n_nodes);
    for (uint32_t i = 0; i < n_nodes; i++) expand[i] = keyed[i].second;
    hp.mark("expansion order (sort)");
    uint32_t head1_by_byte[256] = {};  // single-child nodes: first block that may still hold (free slot f, unclaimed base f ^ byte)
    keyed = std::vector<std::pair<uint64_t, uint32_t>>();
    for (uint32_t node : expand) {
        uint32_t lo = row[node], hi = row[node + 1], k = hi - lo;
        if (k == 0) continue;
        uint32_t chosen = 0;
        bool found = false;
        if (k == 1) {
            // lowest free slot f of the first block that has one whose base f ^ byte is still unclaimed (bases are
            // unique, so that a record can be checked by its edge byte alone: build_trie8)
            // (a block that has no such pair for byte c now never will — slots only fill up, bases only get
            // claimed — so every byte keeps the first block that may still serve it: without it the scan over
            // blocks with free slots but claimed bases was two thirds of the build at 500 000 tokens)
            const uint32_t c = edges[lo].byte;
            uint32_t& head_c = head1_by_byte[c];
            for (uint32_t b = head_c; !found; b++) {
                if (b == ba.n_blocks()) ba.add_block();
                if (ba.free_cnt[b] == 0) {
                    if (b == head_c) head_c++;
                    continue;
                }
                for (uint32_t w = 0; w < 4 && !found; w++) {
                    // slot f = x ^ c is free and base x is unclaimed: permute the claimed-base mask like the slots below
                    uint64_t bu = ba.base_used[(size_t)b * 4 + (w ^ (c >> 6))];
                    if (c & 1u) bu = ((bu >> 1) & 0x5555555555555555ULL) | ((bu & 0x5555555555555555ULL) << 1);
                    if (c & 2u) bu = ((bu >> 2) & 0x3333333333333333ULL) | ((bu & 0x3333333333333333ULL) << 2);
                    if (c & 4u) bu = ((bu >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((bu & 0x0F0F0F0F0F0F0F0FULL) << 4);
                    if (c & 8u) bu = ((bu >> 8) & 0x00FF00FF00FF00FFULL) | ((bu & 0x00FF00FF00FF00FFULL) << 8);
                    if (c & 16u) bu = ((bu >> 16) & 0x0000FFFF0000FFFFULL) | ((bu & 0x0000FFFF0000FFFFULL) << 16);
                    if (c & 32u) bu = (bu >> 32) | (bu << 32);
                    const uint64_t ok = ~ba.used[(size_t)b * 4 + w] & ~bu;  // bit f: slot f free and base f ^ c unclaimed
                    if (ok) {
                        const uint32_t f = w * 64u + (uint32_t)__builtin_ctzll(ok);
                        chosen = (b << 8) | (f ^ c);
                        found = true;
                    }
                }
                if (!found && b == head_c) head_c++;
            }
        } else {
            while (ba.headk < ba.n_blocks() && ba.free_cnt[ba.headk] < 24) ba.headk++;
            uint32_t tried = 0;
            for (uint32_t b = ba.headk; b < ba.n_blocks() && tried < 24 && !found; b++) {
                if (ba.free_cnt[b] < k) continue;
                tried++;
                // x fits iff used[x ^ c] is clear for every child byte c: OR the block's mask permuted by each
                // c (bit x of the permuted mask = bit x ^ c of the mask) and take the lowest clear bit
                const uint64_t* u = &ba.used[(size_t)b * 4];
                const uint64_t* bu = &ba.base_used[(size_t)b * 4];
                uint64_t forbidden[4] = {bu[0], bu[1], bu[2], bu[3]};  // bases are unique (build_trie8)
                for (uint32_t j = lo; j < hi; j++) {
                    const uint32_t c = edges[j].byte;
                    for (uint32_t w = 0; w < 4; w++) {
                        uint64_t v = u[w ^ (c >> 6)];
                        if (c & 1u) v = ((v >> 1) & 0x5555555555555555ULL) | ((v & 0x5555555555555555ULL) << 1);
                        if (c & 2u) v = ((v >> 2) & 0x3333333333333333ULL) | ((v & 0x3333333333333333ULL) << 2);
                        if (c & 4u) v = ((v >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((v & 0x0F0F0F0F0F0F0F0FULL) << 4);
                        if (c & 8u) v = ((v >> 8) & 0x00FF00FF00FF00FFULL) | ((v & 0x00FF00FF00FF00FFULL) << 8);
                        if (c & 16u) v = ((v >> 16) & 0x0000FFFF0000FFFFULL) | ((v & 0x0000FFFF0000FFFFULL) << 16);
                        if (c & 32u) v = (v >> 32) | (v << 32);
                        forbidden[w] |= v;
                    }
                }
                for (uint32_t w = 0; w < 4 && !found; w++)
                    if (~forbidden[w]) {
                        chosen = (b << 8) | (w * 64u + (uint32_t)__builtin_ctzll(~forbidden[w]));
                        found = true;
                    }
            }
            if (!found) {
                ba.add_block();
                chosen = (ba.n_blocks() - 1) << 8;
                found = true;
            }
        }
        base[node] = chosen;
        ba.base_used[chosen >> 6] |= 1ULL << (chosen & 63u);
        for (uint32_t j = lo; j < hi; j++) {
            uint32_t t = chosen ^ edges[j].byte;
            ba.mark(t);
            slot[edges[j].child] = t;
        }
        {   // a block whose 256 base values are all claimed (or reserved) can take no further sibling set
            const uint64_t* bu = &ba.base_used[(size_t)(chosen >> 8) * 4];
            if ((bu[0] & bu[1] & bu[2] & bu[3]) == ~0ULL) ba.free_cnt[chosen >> 8] = 0;
        }
    }

    hp.mark("slot assignment");
    // 4. records.
    uint32_t n_slots = ba.n_blocks() * 256;
    out->table.assign(n_slots, TrieRec{kNoParent, 0, 0});
    out->tokid.assign(n_slots, kNoToken);
    out->label.assign(n_slots, 0);
    out->inner.assign(n_slots, 0);
    out->table[0].base = base[0];
    out->inner[0] = row[1] > row[0];
    for (const Edge& e : edges) {
        uint32_t t = slot[e.child];
        out->label[t] = e.byte;
        out->inner[t] = row[e.child + 1] > row[e.child];
        TrieRec& r = out->table[t];
        r.check = slot[e.parent];
        r.base = base[e.child];
        uint32_t id = node_tok[e.child];
        if (id != kNoToken) {
            r.base |= kTerminalBit;
            std::memcpy(&r.score_bits, &scores[id], 8);
            out->tokid[t] = id;
        }
    }
    out->max_token_len = max_len;
    out->n_nodes = n_nodes;
    hp.mark("records");
}

// ---- 8-byte label-checked records + score table (encode5_kernel) ----------------------------------------
void build_trie8(const FlatTrie& ft, const uint64_t* offs, const double* scores, Trie8* out) {
    const uint32_t n_slots = (uint32_t)ft.table.size();
    out->rec.assign(n_slots, Trie8Rec{0, 0});
    out->values.clear();
    out->coverage.clear();
    out->ok = false;
    // distinct score values (by bit pattern) of the tokens that can match, weighted by how often their
    // tokens are expected to match: a token of probability mass w and length l starts at about w / l of
    // the positions (its Viterbi share; the proxy only has to rank the values)
    struct Val {
        uint64_t bits;
        double weight;
    };
    std::vector<std::pair<uint64_t, uint32_t>> by_bits;  // (score bits, slot) of terminal slots
    by_bits.reserve(n_slots / 2);
    for (uint32_t t = 0; t < n_slots; t++)
        if (ft.tokid[t] != kNoToken) by_bits.push_back({ft.table[t].score_bits, t});
    std::sort(by_bits.begin(), by_bits.end());
    std::vector<Val> vals;
    std::vector<uint32_t> val_of(by_bits.size());  // by_bits index -> vals index
    double total_w = 0.0;
    for (size_t i = 0; i < by_bits.size(); i++) {
        if (i == 0 || by_bits[i].first != by_bits[i - 1].first) vals.push_back(Val{by_bits[i].first, 0.0});
        const uint32_t id = ft.tokid[by_bits[i].second];
        const double len = (double)std::max<uint64_t>(1, offs[id + 1] - offs[id]);
        double w = std::exp(scores[id]) / len;
        if (!(w == w) || w > 1e300) w = 0.0;
        vals.back().weight += w;
        total_w += w;
        val_of[i] = (uint32_t)vals.size() - 1;
    }
    if (vals.size() > kTrie8MaxValues || n_slots > kTrie8MaxSlots) return;  // ranks are 16 bits, 8 * base 24
    std::vector<uint32_t> order(vals.size());
    for (uint32_t i = 0; i < order.size(); i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        if (vals[a].weight != vals[b].weight) return vals[a].weight > vals[b].weight;
        return vals[a].bits < vals[b].bits;
    });
    std::vector<uint32_t> rank(vals.size(), 0u);
    out->values.resize(order.size() + 1);
    out->coverage.resize(order.size() + 1);
    {
        const double ninf = -__builtin_huge_val();
        std::memcpy(&out->values[0], &ninf, 8);
        out->coverage[0] = 0.0;
    }
    double seen_w = 0.0;
    for (uint32_t r = 0; r < order.size(); r++) {
        rank[order[r]] = r + 1u;
        out->values[r + 1u] = vals[order[r]].bits;
        seen_w += vals[order[r]].weight;
        out->coverage[r + 1u] = total_w > 0.0 ? std::min(seen_w / total_w, 1.0) : 1.0;
    }
    out->coverage.back() = 1.0;  // all values (the running sum need not end exactly on the total)
    std::vector<uint32_t> sref(n_slots, 0);
    for (size_t i = 0; i < by_bits.size(); i++) sref[by_bits[i].second] = rank[val_of[i]];
    for (uint32_t t = 0; t < n_slots; t++) {
        Trie8Rec& q = out->rec[t];
        const bool used = t != 0 && ft.table[t].check != kNoParent;
        if (used) {
            const uint32_t base = ft.inner[t] ? (ft.table[t].base & ~kTerminalBit) : kTrie8LeafBase;
            q.rec = (base << 3) | ((uint32_t)ft.label[t] << 24);
            q.sref = sref[t];
        } else {  // unused slots (and the root's own slot) can never pass the label check: see BlockAlloc::add_block
            q.rec = (kTrie8LeafBase << 3) | (((t ^ 0xFFu) & 0xFFu) << 24);
            q.sref = 0;
        }
    }
    out->root_base = ft.inner[0] ? (ft.table[0].base & ~kTerminalBit) : kTrie8LeafBase;
    out->ok = true;
}

// host twin of the device walk over the 8-byte records
uint64_t trie8_common_prefix_search(const Trie8& t8, const FlatTrie& ft, const uint8_t* s, uint64_t n, uint32_t* ids,
                                    uint32_t* lens, uint64_t cap) {
    uint32_t off = (t8.root_base ^ (n ? s[0] : 0u)) << 3;  // byte offset of the record to read
    uint64_t found = 0;
    for (uint64_t i = 0; i < n; i++) {
        const uint32_t t = off >> 3;
        if (t >= t8.rec.size()) break;
        const uint32_t r = t8.rec[t].rec;
        if ((r >> 24) != s[i]) break;
        if (t8.rec[t].sref & kTrie8RankMask) {
            if (found < cap) {
                ids[found] = ft.tokid[t];
                lens[found] = (uint32_t)(i + 1);
            }
            found++;
        }
        if (i + 1 < n) off = (r ^ ((uint32_t)s[i + 1] << 3)) & 0xFFFFFFu;
    }
    return found;
}

// ---- token-ranked 8-byte records (estep7_kernel) -----------------------------------------------------------------
void build_trie8t(const FlatTrie& ft, const uint64_t* offs, const double* scores, Trie8T* out) {
    const uint32_t n_slots = (uint32_t)ft.table.size();
    out->ok = false;
    out->rec.clear();
    out->w.clear();
    out->id_of_rank.clear();
    out->n_tok = 0;
    if (n_slots > kTrie8TMaxSlots) return;
    // the tokens that can match (those that own a terminal slot): the kTrie8TSortedRanks with the largest exp(score) / length
    // in that order, ties by id; the others follow in id order
    struct Key {
        double weight;
        uint32_t id;
    };
    std::vector<Key> keys;
    keys.reserve(n_slots / 2);
    uint32_t id_bound = 0;
    for (uint32_t t = 0; t < n_slots; t++) {
        const uint32_t id = ft.tokid[t];
        if (id == kNoToken) continue;
        keys.push_back(Key{0.0, id});
        id_bound = std::max(id_bound, id + 1u);
    }
    std::vector<double> exp_of(id_bound, 0.0);  // exp(score) by id: computed once (the weight table below wants the same values)
    parallel_ranges(keys.size(), 32768, [&](size_t a, size_t b) {  // (half a million exp: 8 of this function's 17 ms on one thread)
        for (size_t i = a; i < b; i++) {
            const uint32_t id = keys[i].id;
            const double len = (double)std::max<uint64_t>(1, offs[id + 1] - offs[id]);
            const double e = std::exp(scores[id]);
            double w = e / len;
            if (!(w == w)) w = 0.0;
            keys[i].weight = w;
            exp_of[id] = e;
        }
    });
    const auto hotter = [](const Key& a, const Key& b) { return a.weight != b.weight ? a.weight > b.weight : a.id < b.id; };
    const size_t head = std::min<size_t>(keys.size(), kTrie8TSortedRanks);
    if (head < keys.size()) {
        std::nth_element(keys.begin(), keys.begin() + (long)head, keys.end(), hotter);
        // the tail in id order without a sort: mark its ids, then read the marks in order
        std::vector<uint8_t> in_tail(id_bound, 0);
        for (size_t i = head; i < keys.size(); i++) in_tail[keys[i].id] = 1;
        size_t k = head;
        for (uint32_t id = 0; id < id_bound; id++)
            if (in_tail[id]) keys[k++] = Key{0.0, id};
    }
    std::sort(keys.begin(), keys.begin() + (long)head, hotter);
    const uint32_t n_tok = (uint32_t)keys.size();
    out->n_tok = n_tok;
    out->w.assign((size_t)n_tok + 1, 0.0);
    out->id_of_rank.assign((size_t)n_tok + 1, kNoToken);
    uint32_t max_id = 0;
    for (const Key& k : keys) max_id = std::max(max_id, k.id);
    std::vector<uint32_t> rank_of((size_t)max_id + 1, 0u);
    for (uint32_t r = 0; r < n_tok; r++) {
        rank_of[keys[r].id] = r + 1u;
        out->id_of_rank[r + 1u] = keys[r].id;
        out->w[r + 1u] = exp_of[keys[r].id];  // std::exp(score): the same function of the same double as the 16-byte tables' weights
    }
    out->rec.assign(n_slots, Trie8TRec{0, 0});
    parallel_ranges(n_slots, 131072, [&](size_t ta, size_t tb) {
    for (uint32_t t = (uint32_t)ta; t < (uint32_t)tb; t++) {
        Trie8TRec& q = out->rec[t];
        const bool used = t != 0 && ft.table[t].check != kNoParent;
        if (used) {
            const uint32_t base = ft.inner[t] ? (ft.table[t].base & ~kTerminalBit) : kTrie8LeafBase;
            q.rec = base | ((uint32_t)ft.label[t] << 24);
            q.tok = ft.tokid[t] != kNoToken ? rank_of[ft.tokid[t]] : 0u;
        } else {  // unused slots (and the root's own) can never pass the label check: see BlockAlloc::add_block
            q.rec = kTrie8LeafBase | (((t ^ 0xFFu) & 0xFFu) << 24);
            q.tok = 0;
        }
    }
    });
    out->root_base = ft.inner[0] ? (ft.table[0].base & ~kTerminalBit) : kTrie8LeafBase;
    out->ok = true;
}

// host twin of estep7_kernel's walk: ids of the tokens that are prefixes of s
uint64_t trie8t_common_prefix_search(const Trie8T& t8, const uint8_t* s, uint64_t n, uint32_t* ids, uint32_t* lens, uint64_t cap) {
    uint32_t t = n ? (t8.root_base ^ s[0]) : 0u;
    uint64_t found = 0;
    for (uint64_t i = 0; i < n; i++) {
        if (t >= t8.rec.size()) break;
        const uint32_t r = t8.rec[t].rec;
        if ((r >> 24) != s[i]) break;
        if (t8.rec[t].tok) {
            if (found < cap) {
                ids[found] = t8.id_of_rank[t8.rec[t].tok];
                lens[found] = (uint32_t)(i + 1);
            }
            found++;
        }
        if (i + 1 < n) t = (r ^ (uint32_t)s[i + 1]) & 0xFFFFFFu;
    }
    return found;
}

static bool fill_tok_hash(const uint8_t* bytes, const uint64_t* offs, uint32_t vocab_size, TokHashTable* out) {
    std::fill(out->slots.begin(), out->slots.end(), TokHashEntry{0, 0, 0});
    for (uint32_t id = 0; id < vocab_size; id++) {
        const uint64_t b = offs[id], e = offs[id + 1];
        const uint32_t len = (uint32_t)(e - b);
        if (len == 0) continue;  // never matched
        uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        std::memcpy(w, bytes + b, len);
        const uint64_t h = tok_hash64_long(w, len, out->seed);
        uint32_t i = (uint32_t)h & out->mask;
        for (;;) {
            TokHashEntry& s = out->slots[i];
            if (!s.used) {
                s = TokHashEntry{h, id, 1};
                break;
            }
            if (s.hash == h) {
                // same hash: must be the same bytes (a duplicate token: the later id wins, trie.rs:19)
                const uint64_t ob = offs[s.id], oe = offs[s.id + 1];
                if (oe - ob != e - b || std::memcmp(bytes + ob, bytes + b, len) != 0) return false;  // a genuine collision
                s.id = id;
                break;
            }
            i = (i + 1) & out->mask;
        }
    }
    return true;
}

void build_tok_hash(const uint8_t* bytes, const uint64_t* offs, uint32_t vocab_size, TokHashTable* out) {
    uint32_t cap = 1024;
    while (cap < vocab_size * 8u) cap <<= 1;  // load factor <= 1/8: a probe chain is as long as its slowest lane
    out->slots.assign(cap, TokHashEntry{0, 0, 0});
    out->mask = cap - 1;
    out->ok = false;
    for (uint32_t id = 0; id < vocab_size; id++)
        if (offs[id + 1] - offs[id] > 32) return;  // longer tokens: callers keep trie handles
    for (uint32_t attempt = 0; attempt < 8 && !out->ok; attempt++) {
        out->seed = attempt * 0x9E3779B1u;
        out->ok = fill_tok_hash(bytes, offs, vocab_size, out);
    }
}

uint64_t flat_common_prefix_search(const FlatTrie& t, const uint8_t* s, uint64_t n, uint32_t* ids,
                                   uint32_t* lens, uint64_t cap) {
    uint32_t cur = 0, base = t.table[0].base & ~kTerminalBit;
    uint64_t found = 0;
    for (uint64_t i = 0; i < n; i++) {
        uint32_t nxt = base ^ s[i];
        const TrieRec& r = t.table[nxt];
        if (r.check != cur) break;
        cur = nxt;
        base = r.base & ~kTerminalBit;
        if (r.base & kTerminalBit) {
            if (found < cap) {
                ids[found] = t.tokid[nxt];
                lens[found] = (uint32_t)(i + 1);
            }
            found++;
        }
    }
    return found;
}

}  // namespace tgx
