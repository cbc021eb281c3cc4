// Flattened byte trie: XOR double-array, the HBM-resident replacement of the
// reference's pointer/HashMap trie (reference src/trie.rs:7-87, built by
// Model::from, src/model.rs:16-30).
//
// One 16-byte record per slot t:
//   x = check  : slot of the parent node (0xFFFFFFFF = unused slot / root)
//   y = base   : children of this node live at slot (base ^ byte); bit 31 set
//                when this node is terminal (a vocabulary token ends here)
//   z,w = score: f64 bits of vocab[id].score for terminal nodes
// Child of node s by byte c is t = base[s] ^ c, valid iff check[t] == s, so one
// 16-byte gather per trie step, and all children of a node share one 256-slot
// (4 KiB) block.  The root is slot 0.  tokid[t] gives the token id of a
// terminal slot (read once per emitted token, not per match).
#pragma once
#include <cstdint>
#include <memory>
#include <vector>

namespace tgx {

// Wall time of host phases to stderr when TGX_KNOBS=1 TGX_HOST_TIMES=1 (tools/prune_bench.py): where a prune iteration's
// host time goes.  mark("x") prints the time since the previous mark (or construction).
struct HostPhases {
    const char* scope;
    bool on;
    double t0, last;
    explicit HostPhases(const char* scope_);
    void mark(const char* what);
    ~HostPhases();
};

struct TrieRec {
    uint32_t check;
    uint32_t base;  // bit 31: terminal
    uint64_t score_bits;
};
static_assert(sizeof(TrieRec) == 16, "TrieRec must be 16 bytes");

constexpr uint32_t kTerminalBit = 0x80000000u;
constexpr uint32_t kNoParent = 0xFFFFFFFFu;
constexpr uint32_t kNoToken = 0xFFFFFFFFu;

struct FlatTrie {
    std::vector<TrieRec> table;   // n_slots records (multiple of 256)
    std::vector<uint32_t> tokid;  // n_slots, kNoToken when not terminal
    std::vector<uint8_t> label;   // n_slots: byte of the edge that leads to the slot (used slots other than the root)
    std::vector<uint8_t> inner;   // n_slots: the node has children
    uint32_t max_token_len = 0;
    uint32_t n_nodes = 0;
};

// Model::from: token i = bytes[offs[i]..offs[i+1]), id = i; a later duplicate
// overwrites the payload of an earlier one (src/trie.rs:19); empty tokens are
// stored on the root and can never match (src/trie.rs:53-61), so they are skipped.
void build_flat_trie(const uint8_t* bytes, const uint64_t* offs, const double* scores,
                     uint32_t vocab_size, FlatTrie* out);

// Label-checked 8-byte records over the SAME slot assignment (encode5_kernel): the walk keeps only the record,
//   rec = table[off / 8];  valid iff (rec >> 24) == byte;  next off = (rec ^ (next byte << 3)) & 0xFFFFFF
// i.e. `rec` = 8 * base (the BYTE offset of the node's child block: (base ^ c) * 8 = 8 base ^ 8 c) in its low 24
// bits and the label — the byte of the edge that leads to the slot — in its top byte: one compare, one shift of
// the text byte and one three-input bit operation per step (round 3; the first layout, label | base << 8, took
// four instructions for the address alone, and the kernel is bound by vector-instruction issue).  Exact because
// build_flat_trie gives every node with children a base of its own (if slot t passes the check for (base, byte) its
// owner's base is t ^ byte = base) whose low byte is neither 0xFE nor 0xFF: leaves point at base 0xFE of block 0
// (the root's block, otherwise empty) and an unused slot t carries the label (t ^ 0xFF) & 0xFF, so neither can
// pass.  8 base < 2^24 limits the table to 2^21 slots (a 65 536-entry vocabulary has ~3 * 10^5).
// `sref`: the RANK of the token's score value, 0 for a slot no token ends at — the distinct score values of the
// vocabulary are ranked by how often their tokens are expected to match (1 = hottest), and `values[rank]` is the
// table the kernels read: its first entries from a copy in LDS, the rest from HBM / L2.  A vocabulary after an
// M-step has one value per token (src/prune.rs:143-151), so ranks must cover 65 535 values; a vocabulary with more
// distinct values than that has no Trie8 (ok == false: the 16-byte records serve it).
// (Round 3 also tried a 16-bit CHILD MASK in the upper half of sref — bit (c >> 4) set iff the node has a child
// in that byte class — which ends 87 % of the walks' failing probes without a gather (3.46 -> 2.55 gathers per
// position below the root level): the kernel got 4 % SLOWER, the five instructions of the test cost more than the
// gathers they save — DESIGN.md section 6.)
struct Trie8Rec {
    uint32_t rec;   // 8 * base | label << 24
    uint32_t sref;  // rank of the score value (16 bits), 0: no token ends here
};
constexpr uint32_t kTrie8RankMask = 0xFFFFu;
constexpr uint32_t kTrie8MaxValues = 65535u;
constexpr uint32_t kTrie8MaxSlots = 1u << 21;
constexpr uint32_t kTrie8LeafBase = 0xFEu;
struct Trie8 {
    std::vector<Trie8Rec> rec;     // n_slots
    std::vector<uint64_t> values;  // bit patterns: values[0] = -inf ("no token"), values[r] = the value of rank r
    std::vector<double> coverage;  // coverage[k] = expected share of the matches whose value has rank <= k (k = 0 .. n)
    uint32_t root_base = 0;
    bool ok = false;
};
void build_trie8(const FlatTrie& ft, const uint64_t* offs, const double* scores, Trie8* out);
uint64_t trie8_common_prefix_search(const Trie8& t8, const FlatTrie& ft, const uint8_t* s, uint64_t n, uint32_t* ids,
                                    uint32_t* lens, uint64_t cap);

// Records of estep7_kernel (estep7.hip) over the SAME slot assignment: {base | label << 24, rank of the token that ends
// here (0: none)}.  (The order below is the one build_trie8t leaves; tgx_api.cpp re-ranks by MEASURED match counts at the
// model's first E-step — round 4 also tried the probability mass of the tokens below a token's node as the proxy for how
// often it matches: worse than exp(score) / length for both this table and encode5's value table.)  The child by byte c of a node is the record at byte offset 8 * ((rec ^ c) & 0xFFFFFF), valid iff its
// label is c (exact for the reasons given above; 2^24 slots).  Tokens are ranked by exp(score) / length — how often they
// are expected to match — because the first ranks' weights and expected-count sums live in the blocks' LDS; only the
// first `sorted_ranks` ranks are in that order (the kernels never keep more in LDS), the others follow in id order.
// w[r] = exp(score of the token of rank r) (w[0] = 0: "no token"), id_of_rank[r] its vocabulary id.
struct Trie8TRec {
    uint32_t rec;  // base | label << 24
    uint32_t tok;  // rank of the token, 0: no token ends here
};
constexpr uint32_t kTrie8TMaxSlots = 1u << 24;
constexpr uint32_t kTrie8TSortedRanks = 16384u;
struct Trie8T {
    std::vector<Trie8TRec> rec;        // n_slots
    std::vector<double> w;             // n_tok + 1
    std::vector<uint32_t> id_of_rank;  // n_tok + 1 ([0] unused)
    uint32_t root_base = 0;
    uint32_t n_tok = 0;                // tokens that own a slot (duplicates and empty tokens do not)
    bool ok = false;
};
void build_trie8t(const FlatTrie& ft, const uint64_t* offs, const double* scores, Trie8T* out);
uint64_t trie8t_common_prefix_search(const Trie8T& t8, const uint8_t* s, uint64_t n, uint32_t* ids, uint32_t* lens, uint64_t cap);

// Token-bytes -> id hash table (tokens of 1..32 bytes): lets the trace kernel turn a
// (start, length) pair straight into a token id with one probe instead of carrying a trie
// handle through the DP.  Open addressing, linear probing, capacity a power of two >= 2V;
// entry = {hash64 of the bytes, id, used}.  The 64-bit hash is verified to be collision-free
// over the vocabulary at build time (ok == false otherwise: callers keep trie handles).
struct TokHashEntry {
    uint64_t hash;
    uint32_t id;
    uint32_t used;
};
struct TokHashTable {
    std::vector<TokHashEntry> slots;
    uint32_t mask = 0;
    uint32_t seed = 0;
    bool ok = false;
};
// Hash of a token of len (1..16) bytes given as four little-endian zero-padded dwords: two 32-bit
// words.  `lo` (slot index and half of the tag) is a multiply-xorshift chain over the dwords, `hi` folds
// the chain's intermediate states with the raw dwords.  32-bit multiplies only: the trace kernel is
// bound by VALU issue and 64-bit multiplies run at a quarter of the rate four times over.  It only has to
// be injective on the vocabulary, which build_tok_hash verifies (another seed is tried otherwise).
inline uint32_t tok_rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
inline uint64_t tok_hash64(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t len, uint32_t seed) {
    uint32_t a = (w0 ^ (len << 27) ^ seed) * 0x85EBCA6Bu;
    a ^= a >> 15;
    uint32_t b = a;
    a = (a + w1) * 0xC2B2AE35u;
    a ^= a >> 13;
    b = tok_rotl32(b, 11) ^ a;
    a = (a + w2) * 0x27D4EB2Fu;
    a ^= a >> 16;
    b = tok_rotl32(b, 11) ^ a;
    a = (a + w3) * 0x165667B1u;
    a ^= a >> 15;
    b = tok_rotl32(b, 11) + (w0 ^ tok_rotl32(w1, 8) ^ tok_rotl32(w2, 16) ^ tok_rotl32(w3, 24));
    return ((uint64_t)b << 32) | a;
}
// tok_hash64 continued over four more dwords for tokens of 17..32 bytes (eight zero-padded dwords);
// equal to tok_hash64 for len <= 16
inline uint64_t tok_hash64_long(const uint32_t* w, uint32_t len, uint32_t seed) {
    const uint64_t h = tok_hash64(w[0], w[1], w[2], w[3], len, seed);
    if (len <= 16) return h;
    uint32_t a = (uint32_t)h, b = (uint32_t)(h >> 32);
    a = (a + w[4]) * 0x85EBCA6Bu;
    a ^= a >> 15;
    b = tok_rotl32(b, 11) ^ a;
    a = (a + w[5]) * 0xC2B2AE35u;
    a ^= a >> 13;
    b = tok_rotl32(b, 11) ^ a;
    a = (a + w[6]) * 0x27D4EB2Fu;
    a ^= a >> 16;
    b = tok_rotl32(b, 11) ^ a;
    a = (a + w[7]) * 0x165667B1u;
    a ^= a >> 15;
    b = tok_rotl32(b, 11) + (w[4] ^ tok_rotl32(w[5], 8) ^ tok_rotl32(w[6], 16) ^ tok_rotl32(w[7], 24));
    return ((uint64_t)b << 32) | a;
}
void build_tok_hash(const uint8_t* bytes, const uint64_t* offs, uint32_t vocab_size, TokHashTable* out);

// Host twin of the device walk (TrieIterator::next, src/trie.rs:51-63).
uint64_t flat_common_prefix_search(const FlatTrie& t, const uint8_t* s, uint64_t n, uint32_t* ids,
                                   uint32_t* lens, uint64_t cap);

}  // namespace tgx

// prune_host.cpp: tgx_prune_alternatives over a table that already exists (a handle's, or a model's own)
// prune_host.cpp: tgx_prune_alternatives over a table that already exists (a handle's, or a model's own);
// returns a tgx_status
extern "C" int tgx_prune_alternatives_flat(const tgx::FlatTrie* flat_trie, const uint8_t* bytes, const uint64_t* offs,
                                const double* scores, uint32_t vocab_size, uint8_t* always_keep,
                                uint32_t* alt_offs, uint32_t** alt_ids);

// host-only handle of include/tgx.h's tgx_flat_trie_* functions (tgx_api.cpp, prune_host.cpp)
struct tgx_flat_trie {
    tgx::FlatTrie flat;
    std::unique_ptr<tgx::Trie8> t8;  // tgx_flat_trie_search8: built on first use
};
