// C ABI (include/tgx.h) over the HIP kernels: handles, HBM buffers, launch
// sequencing, error reporting.  Host side of the reference's batch loops
// (src/tokenizer.rs:102-123, src/prune.rs:205-244); there is no CPU fallback —
// without a usable gfx950 device every compute entry point returns TGX_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <unordered_set>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <new>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "../../include/tgx.h"
#include "kernels.h"
#include "trie_build.h"

namespace {

// ---- thread-local error state ------------------------------------------------
thread_local std::string g_err_msg;
thread_local uint64_t g_err_sample = 0, g_err_pos = 0, g_err_len = 0;

tgx_status fail(tgx_status st, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err_msg = buf;
    return st;
}
}  // namespace
// shared with prune_host.cpp: records the message tgx_last_error() returns (thread-local)
tgx_status tgx_set_error(tgx_status st, const char* msg) {
    g_err_msg = msg ? msg : "";
    return st;
}
namespace {

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess)                                                             \
            return fail(TGX_ERR_DEVICE, "HIP error %d (%s) at %s:%d: %s", (int)_e,        \
                        hipGetErrorString(_e), __FILE__, __LINE__, #expr);                \
    } while (0)

// ---- device buffer pool --------------------------------------------------------
// hipMalloc/hipFree cost far more than a kernel launch; scratch and result
// buffers are recycled through a small per-process free list.
struct PoolEntry {
    void* ptr;
    size_t bytes;
    int device;
};
std::mutex g_pool_mu;
std::vector<PoolEntry> g_pool;
size_t g_pool_bytes = 0;
constexpr size_t kPoolMaxBytes = 64ull << 30;

// hipFree of every pooled buffer of `device` (all devices if < 0); the caller's current device is kept
void pool_trim(int device) {
    std::vector<PoolEntry> victims;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (size_t i = 0; i < g_pool.size();) {
            if (device < 0 || g_pool[i].device == device) {
                victims.push_back(g_pool[i]);
                g_pool_bytes -= g_pool[i].bytes;
                g_pool.erase(g_pool.begin() + (long)i);
            } else {
                i++;
            }
        }
    }
    if (victims.empty()) return;
    int prev = -1;
    (void)hipGetDevice(&prev);
    for (const PoolEntry& e : victims) {
        (void)hipSetDevice(e.device);
        (void)hipFree(e.ptr);
    }
    if (prev >= 0) (void)hipSetDevice(prev);
}

hipError_t pool_alloc(int device, size_t bytes, void** out) {
    if (bytes == 0) bytes = 256;
    bytes = (bytes + 255) & ~size_t(255);
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        size_t best = g_pool.size();
        for (size_t i = 0; i < g_pool.size(); i++) {
            const PoolEntry& e = g_pool[i];
            if (e.device != device || e.bytes < bytes || e.bytes > bytes * 2 + (1u << 20)) continue;
            if (best == g_pool.size() || e.bytes < g_pool[best].bytes) best = i;
        }
        if (best != g_pool.size()) {
            *out = g_pool[best].ptr;
            g_pool_bytes -= g_pool[best].bytes;
            g_pool.erase(g_pool.begin() + (long)best);
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipErrorOutOfMemory) {
        // the pool may be what fills the device (passes of very different sizes, or memory shared with
        // torch / RCCL allocations): give everything back and try once more
        (void)hipGetLastError();
        pool_trim(device);
        e = hipMalloc(out, bytes);
    }
    return e;
}

size_t rounded(size_t bytes) {
    if (bytes == 0) bytes = 256;
    return (bytes + 255) & ~size_t(255);
}

// Pooled bytes per process: at most a quarter of the device's memory (and never more than kPoolMaxBytes).
size_t pool_cap(int device) {
    static size_t cap[16] = {0};
    if (device < 0 || device >= 16) return kPoolMaxBytes;
    if (cap[device] == 0) {
        size_t free_b = 0, total_b = 0;
        int prev = -1;
        (void)hipGetDevice(&prev);
        (void)hipSetDevice(device);
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) total_b = 4 * kPoolMaxBytes;
        if (prev >= 0) (void)hipSetDevice(prev);
        cap[device] = std::min<size_t>(kPoolMaxBytes, total_b / 4);
    }
    return cap[device];
}

void pool_free(int device, void* ptr, size_t bytes) {
    if (!ptr) return;
    bytes = rounded(bytes);
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (g_pool_bytes + bytes <= pool_cap(device) && g_pool.size() < 256) {
            g_pool.push_back(PoolEntry{ptr, bytes, device});
            g_pool_bytes += bytes;
            return;
        }
    }
    int prev = -1;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(device);
    (void)hipFree(ptr);
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
}

struct KernelTime {
    const char* name;
    hipEvent_t start, stop;
    bool used;
};
constexpr int kMaxTimed = 8;

}  // namespace

struct tgx_model {
    int device = 0;
    uint32_t vocab_size = 0;
    uint32_t lm = 0;  // max token length rounded up to a multiple of 4 (>= 4)
    bool scores_finite = true;  // the four-samples-per-wave kernel encodes 'no token' as -inf
    tgx::FlatTrie flat;
    void* d_trie = nullptr;
    uint32_t* d_tokid = nullptr;
    unsigned long long* d_ctrl = nullptr;  // [0] work counter, [1] min failing sample
    unsigned long long* h_ctrl = nullptr;  // pinned: [0] err sample, [1] total tokens
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;          // encode6_kernel beside encode5_kernel (run_encode_kernel: co-run)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    unsigned int* h_started = nullptr;      // page-locked, device-visible: blocks of encode5_kernel that are resident (co-run)
    int num_cus = 0;
    int blocks_per_cu = 0;  // encode_kernel (one sample per wave)
    // E-step only (built on first use): trie of the reversed tokens
    std::vector<uint8_t> vocab_bytes;
    std::vector<uint64_t> vocab_offs;
    std::vector<double> vocab_scores;
    tgx::FlatTrie flat_rev;
    void* d_trie_rev = nullptr;
    bool rev_host_built = false;    // flat_rev was built at creation (TGX_MODEL_FOR_ESTEP)
    void* d_trie_w = nullptr;       // forward / reversed tables with w = exp(score) in place of the score
    void* d_trie_rev_w = nullptr;   //   (linear-domain E-step, estep4l.hip)
    uint64_t last_long_samples = 0;    // samples the last pass gave a block of their own (encode6_kernel)
    uint64_t last_estep_pieces = 0;    // pieces the last E-step cut its snippets into (0: uncut)
    uint32_t last_corun_cus = 0;       // CUs the long-sample kernel had to itself beside encode5_kernel in the last pass (0: one after the other)
    double e7_overflow_share = 0.0;    // share of the sample's matches whose token ranks beyond 65 535 (ensure_estep_trie8t; 0 without counts)
    bool values_ranked = false;        // the score values were re-ranked by how often a sample of some corpus reads them (ensure_value_ranks)
    uint32_t corun_wait_timeouts = 0;  // co-run passes whose host wait for encode5_kernel's blocks ran into its 2 ms limit (then: no more co-runs)
    uint64_t last_redo_samples = 0;    // samples the last encode4l pass left to encode2_kernel
    bool mask_path = false;            // the last encode pass wrote the token-end mask (TGX_TRACE=mask: mark / scan / emit, trace2.hip)
    int last_encode_waves_per_cu = 0;  // resident waves per CU of the last rows4 encode launch (self-check)
    bool estep_linear_ok = false;   // tables for the linear-domain E-step kernels exist
    tgx::TokHashTable tokhash;      // token bytes -> id (rows4 trace); ok == false: not usable
    void* d_tokhash = nullptr;
    // encode5_kernel: 8-byte label-checked records + table of distinct score values (trie_build.h: Trie8)
    void* d_trie8 = nullptr;
    double* d_values = nullptr;        // f64[n_values + 1]: -inf, then the distinct score values by rank (trie_build.h: Trie8)
    std::vector<double> value_coverage;  // [k]: expected share of the matches whose value has rank <= k
    uint32_t n_values = 0, root_base8 = 0;
    uint32_t last_n_hot = 0;           // values in the LDS copy of the last encode5 launch
    bool have_trie8 = false;
    bool encode_tables_ready = false;  // tokhash / trie8 built and uploaded (ensure_encode_tables)
    bool estep_trie8_tried = false;    // ensure_estep_trie8 ran
    uint64_t estep_calls = 0;          // E-step passes this model has run (estep_rows4)
    bool have_wvalues = false;         // d_trie8 / d_wvalues are there for estep5_fwd_kernel
    double* d_wvalues = nullptr;       // f64[n_values + 1]: [0] = 0, [r] = exp(score value of rank r)
    bool tokhash_host_built = false;   // m->tokhash was built beside the forward trie at creation
    // estep7_kernel: token-ranked 8-byte records, w by rank (trie_build.h: Trie8T), built at the first E-step
    void* d_trie8t = nullptr;
    double* d_wtab = nullptr;          // f64[n_tok7 + 1]
    std::vector<uint32_t> id_of_rank;  // [r] = vocabulary id of the token of rank r (1 .. n_tok7)
    uint32_t n_tok7 = 0, root_base7 = 0;
    bool trie8t_tried = false, have_trie8t = false;
    uint64_t last_estep_redo = 0;      // stretches the last fused E-step left to the chained kernels
    int estep_blocks_per_cu = 0;
    KernelTime timed[kMaxTimed] = {};
    int n_timed = 0;
    uint64_t last_alg_bytes = 0;
    std::mutex mu;  // one pass at a time per handle (its stream, counters and events)
};

struct tgx_corpus {
    int device = 0;
    uint64_t n_samples = 0, n_bytes = 0, max_len = 0;  // max_len: longest sample in bytes
    std::vector<uint64_t> h_offs;
    std::vector<uint32_t> h_sorted_len;  // sample lengths in the order of d_order (longest first) ...
    std::vector<uint64_t> h_sorted_cum;  // ... and their running sum
    uint8_t* d_text = nullptr;        // = d_text_alloc + 256 (the backward E-step sweep reads before a position)
    uint8_t* d_text_alloc = nullptr;
    uint64_t* d_offs = nullptr;
    uint32_t* d_order = nullptr;
    uint32_t* d_bp = nullptr;      // scratch, allocated on first pass
    uint32_t* d_tmp = nullptr;
    uint32_t* d_counts = nullptr;
    uint32_t* d_status = nullptr;
    void* d_scan_tmp = nullptr;
    size_t scan_tmp_bytes = 0;
    // round 4 (trace2.hip): one bit per text byte ("a token ends here") and its popcount prefix, allocated on first pass
    unsigned long long* d_endmask = nullptr;
    uint64_t* d_prefix = nullptr;
    uint64_t* d_mword = nullptr;       // u64[S + 1]: the samples' first mask words
    void* d_mscan_tmp = nullptr;
    size_t mscan_tmp_bytes = 0;
    uint64_t mask_words = 0;
    // E-step work list of the corpus (every sample cut at multiples of snippet_len, longest snippet first) and its
    // device copies: built on the first pass with a given snippet length, reused by the following ones (prune
    // runs two E-steps per iteration over the same corpus; building and sorting the list took 5 of 65 ms at 1 GiB)
    struct EstepWork {
        uint64_t snippet_len = 0;
        std::vector<uint64_t> soffs, sbase;
        std::vector<uint32_t> ssample, order;
        uint64_t *d_soffs = nullptr, *d_sbase = nullptr;
        uint32_t *d_order = nullptr, *d_ssample = nullptr;
        size_t obytes = 0, ordbytes = 0;
        // windows of the snippets (cuts.hip: one boundary is sought per window): built with the work list
        uint32_t window = 0;
        uint64_t n_windows = 0;
        uint32_t *d_win_snip = nullptr, *d_win_k = nullptr;
        size_t winbytes = 0;
    } es;
    // the scratch above belongs to the corpus, so a pass holds this lock too (always after its model's):
    // two models may work on one resident corpus from two host threads (prune and merge do, src/prune.rs:48)
    std::mutex mu;
};

struct tgx_result {
    int device = 0;
    uint64_t n_samples = 0, n_tokens = 0;
    uint32_t* d_ids = nullptr;
    uint64_t* d_offs = nullptr;
    std::unique_ptr<uint32_t[]> h_ids;  // uninitialised: a vector would zero a GB first
    std::vector<uint64_t> h_offs;
    bool have_ids = false, have_offs = false;
};

namespace {

int usable_device_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

// TGX_DEBUG=1: log every launch to stderr and synchronise after it, so that a
// device fault can be attributed to one kernel.
bool debug_on() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("TGX_DEBUG");
        v = (e && *e && *e != '0') ? 1 : 0;
    }
    return v == 1;
}

// Environment switches that choose among the library's (all correct) kernels and geometries exist for the
// parity tests, A/B timing and diagnosis; a process must opt in with TGX_KNOBS=1 (tests/conftest.py does) or
// TGX_DEBUG=1, so that a stray variable in a production environment changes nothing.
const char* knob(const char* name) {
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("TGX_KNOBS");
        on = ((e && *e && *e != '0') || debug_on()) ? 1 : 0;
    }
    return on == 1 ? getenv(name) : nullptr;
}

void time_begin(tgx_model* m, const char* name) {
    if (debug_on()) {
        fprintf(stderr, "[tgx] launch %s\n", name);
        fflush(stderr);
    }
    if (m->n_timed >= kMaxTimed) return;
    KernelTime& t = m->timed[m->n_timed];
    t.name = name;
    t.used = true;
    (void)hipEventRecord(t.start, m->stream);
}
void time_end(tgx_model* m) {
    if (debug_on()) {
        hipError_t e = hipStreamSynchronize(m->stream);
        fprintf(stderr, "[tgx]   done: %s\n", hipGetErrorString(e));
        fflush(stderr);
    }
    if (m->n_timed >= kMaxTimed) return;
    (void)hipEventRecord(m->timed[m->n_timed].stop, m->stream);
    m->n_timed++;
}

// mask: the ids go through the token-end mask (TGX_TRACE=mask: mark / scan / emit, trace2.hip — the measured
// alternative, 1.3 ms per GiB slower); else right-aligned in `tmp` and compacted (the default)
tgx_status ensure_scratch(tgx_corpus* c, bool mask) {
    const bool rows = mask;
    HIP_TRY(hipSetDevice(c->device));
    if (!c->d_bp) {
        // u32[N] for the one-sample-per-wave kernel; the rows4 kernels use it as bytes, N + 128 per sample
        HIP_TRY(pool_alloc(c->device, std::max((size_t)c->n_bytes * 4 + 256, (size_t)c->n_bytes + 128 * (size_t)c->n_samples + 512), (void**)&c->d_bp));
        HIP_TRY(pool_alloc(c->device, (size_t)c->n_samples * 4 + 256, (void**)&c->d_counts));
        HIP_TRY(pool_alloc(c->device, (size_t)c->n_samples * 4 + 256, (void**)&c->d_status));
    }
    if (!rows && !c->d_tmp) HIP_TRY(pool_alloc(c->device, (size_t)c->n_bytes * 4 + 256, (void**)&c->d_tmp));
    if (rows && !c->d_endmask) {
        std::vector<uint64_t> mword(c->n_samples + 1, 0);  // sample s owns ceil(n / 64) words of the mask
        for (uint64_t i = 0; i < c->n_samples; i++) mword[i + 1] = mword[i] + (c->h_offs[i + 1] - c->h_offs[i] + 63) / 64;
        c->mask_words = mword[c->n_samples];
        HIP_TRY(pool_alloc(c->device, (size_t)(c->n_samples + 1) * 8 + 256, (void**)&c->d_mword));
        HIP_TRY(hipMemcpy(c->d_mword, mword.data(), (size_t)(c->n_samples + 1) * 8, hipMemcpyHostToDevice));
        HIP_TRY(pool_alloc(c->device, (size_t)(c->mask_words + 1) * 8 + 256, (void**)&c->d_endmask));
        HIP_TRY(hipMemset(c->d_endmask + c->mask_words, 0, 8));  // the word of padding (the scan's last element)
        HIP_TRY(pool_alloc(c->device, (size_t)(c->mask_words + 1) * 8 + 256, (void**)&c->d_prefix));
        HIP_TRY(tgx::mask_scan_temp_bytes(c->mask_words, &c->mscan_tmp_bytes));
        if (c->mscan_tmp_bytes) HIP_TRY(pool_alloc(c->device, c->mscan_tmp_bytes, &c->d_mscan_tmp));
    }
    return TGX_OK;
}

uint32_t grid_blocks(const tgx_model* m, uint64_t n_samples) {
    const uint64_t wpb = tgx::encode_waves_per_block(m->lm);
    uint64_t want = (n_samples + wpb - 1) / wpb;
    uint64_t cap = (uint64_t)m->num_cus * (uint64_t)std::max(1, m->blocks_per_cu);
    return (uint32_t)std::max<uint64_t>(1, std::min(want, cap));
}

// encode5_kernel keeps the first ranks of the score values in LDS.  build_trie8 ranks a value by the probability mass of
// its tokens — how often they are CHOSEN — but a value is read whenever one of its tokens MATCHES: a vocabulary after an
// M-step has a value per token, and its kept single-byte tokens have tiny scores and match at every position.  So, as
// for the E-step's ranks (ensure_estep_trie8t), the first time a model with more values than fit LDS meets text the
// values are re-ranked by match counts over a sample of it (value_count_kernel, 4 MiB): the records' ranks are remapped on
// the device, the value table(s) permuted.  Caller holds m->mu.
tgx_status ensure_value_ranks(tgx_model* m, const tgx_corpus* c) {
    if (m->values_ranked || !m->have_trie8 || !c->n_bytes) return TGX_OK;
    m->values_ranked = true;
    const char* vr = knob("TGX_VALUE_RANK");  // "model": keep build_trie8's order; "counts": re-rank even when every value fits (measurements, tests)
    if (vr && strcmp(vr, "model") == 0) return TGX_OK;
    if (m->n_values <= tgx::encode5_max_hot(false, 13, 3, 160u * 1024u) && !(vr && strcmp(vr, "counts") == 0)) return TGX_OK;  // (every value in LDS anyway)
    tgx::HostPhases hp("ensure_value_ranks");
    HIP_TRY(hipSetDevice(m->device));
    const uint32_t nv = m->n_values, chunk = 65536u;
    const uint64_t stride = std::max<uint64_t>(chunk, (c->n_bytes + 63) / 64);
    const size_t cb = ((size_t)nv + 1) * 4 + 256;
    unsigned int* d_cnt = nullptr;
    uint32_t* d_perm = nullptr;
    auto drop = [&]() {
        pool_free(m->device, d_cnt, cb);
        pool_free(m->device, d_perm, cb);
    };
    if (pool_alloc(m->device, cb, (void**)&d_cnt) != hipSuccess || pool_alloc(m->device, cb, (void**)&d_perm) != hipSuccess) {
        drop();
        return fail(TGX_ERR_DEVICE, "out of device memory (value counts)");
    }
    std::vector<unsigned int> cnt((size_t)nv + 1, 0u);
    std::vector<double> values((size_t)nv + 1), wvalues;
    bool ok = hipMemsetAsync(d_cnt, 0, cb, m->stream) == hipSuccess &&
              tgx::launch_value_count(c->d_text, c->n_bytes, chunk, stride, m->d_trie8, (uint32_t)m->flat.table.size(), m->root_base8, nv,
                                      std::max<uint32_t>(1, m->flat.max_token_len), d_cnt, (uint32_t)m->num_cus, m->stream) == hipSuccess &&
              hipMemcpyAsync(cnt.data(), d_cnt, ((size_t)nv + 1) * 4, hipMemcpyDeviceToHost, m->stream) == hipSuccess &&
              hipMemcpyAsync(values.data(), m->d_values, ((size_t)nv + 1) * 8, hipMemcpyDeviceToHost, m->stream) == hipSuccess;
    if (ok && m->have_wvalues) {
        wvalues.resize((size_t)nv + 1);
        ok = hipMemcpyAsync(wvalues.data(), m->d_wvalues, ((size_t)nv + 1) * 8, hipMemcpyDeviceToHost, m->stream) == hipSuccess;
    }
    if (!ok || hipStreamSynchronize(m->stream) != hipSuccess) {
        drop();
        return fail(TGX_ERR_DEVICE, "value count pass failed: %s", hipGetErrorString(hipGetLastError()));
    }
    hp.mark("value counts");
    std::vector<uint32_t> order(nv);
    for (uint32_t r = 0; r < nv; r++) order[r] = r + 1u;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cnt[a] > cnt[b]; });  // ties: the old order
    std::vector<uint32_t> perm((size_t)nv + 1, 0u);
    std::vector<double> v2((size_t)nv + 1), w2(wvalues.size());
    v2[0] = values[0];
    if (!w2.empty()) w2[0] = wvalues[0];
    unsigned long long total = 0, run = 0;
    for (uint32_t r = 0; r < nv; r++) total += cnt[order[r]];
    std::vector<double> cov((size_t)nv + 1, 0.0);
    for (uint32_t r = 0; r < nv; r++) {
        perm[order[r]] = r + 1u;
        v2[r + 1u] = values[order[r]];
        if (!w2.empty()) w2[r + 1u] = wvalues[order[r]];
        run += cnt[order[r]];
        cov[r + 1u] = total ? (double)run / (double)total : 1.0;
    }
    cov[nv] = 1.0;
    ok = hipMemcpyAsync(d_perm, perm.data(), ((size_t)nv + 1) * 4, hipMemcpyHostToDevice, m->stream) == hipSuccess &&
         tgx::launch_rank_remap(m->d_trie8, (uint32_t)m->flat.table.size(), d_perm, m->stream) == hipSuccess &&
         hipMemcpyAsync(m->d_values, v2.data(), ((size_t)nv + 1) * 8, hipMemcpyHostToDevice, m->stream) == hipSuccess &&
         (w2.empty() || hipMemcpyAsync(m->d_wvalues, w2.data(), ((size_t)nv + 1) * 8, hipMemcpyHostToDevice, m->stream) == hipSuccess) &&
         hipStreamSynchronize(m->stream) == hipSuccess;
    drop();
    if (!ok) return fail(TGX_ERR_DEVICE, "value re-rank failed: %s", hipGetErrorString(hipGetLastError()));
    m->value_coverage = std::move(cov);
    hp.mark("re-rank");
    return TGX_OK;
}

// Runs the wave-per-sample kernel over the corpus; on return (stream synced)
// h_ctrl[0] = min failing sample (~0 if none).
tgx_status run_encode_kernel(tgx_model* m, tgx_corpus* c, double dropout, uint64_t seed) {
    {
        const char* force0 = knob("TGX_PATH");
        const char* tr = knob("TGX_TRACE");
        m->mask_path = tr && strcmp(tr, "mask") == 0 && m->lm <= 32 && m->scores_finite && m->d_tokhash && !(force0 && strcmp(force0, "fused") == 0);
    }
    tgx_status st = ensure_scratch(c, m->mask_path);
    if (st != TGX_OK) return st;
    HIP_TRY(hipMemsetAsync(m->d_ctrl, 0x00, 8, m->stream));
    HIP_TRY(hipMemsetAsync(m->d_ctrl + 1, 0xFF, 8, m->stream));
    tgx::EncodeParams p{};
    p.text = c->d_text;
    p.offs = c->d_offs;
    p.order = c->d_order;
    p.n_samples = c->n_samples;
    p.trie = m->d_trie;
    p.tokid = m->d_tokid;
    p.root_base = m->flat.table[0].base & ~tgx::kTerminalBit;
    p.lm = m->lm;
    p.n_slots = (uint32_t)m->flat.table.size();
    p.bp = c->d_bp;
    p.tmp = c->d_tmp;
    p.counts = c->d_counts;
    p.status = c->d_status;
    p.bp8 = reinterpret_cast<uint8_t*>(c->d_bp);  // the rows4 path uses the scratch row as bytes
    p.tokhash = m->d_tokhash;
    p.tokhash_mask = m->tokhash.mask;
    p.tokhash_seed = m->tokhash.seed;
    p.err_sample = m->d_ctrl + 1;
    p.queue = m->d_ctrl;
    p.dropout = dropout;
    p.seed = seed;
    // samples of less than 2 KiB on average: the trace keeps its waiting tokens across samples (trace_body.h; 140-byte samples
    // 5.5 -> 4.6 ms per GiB, 9 KiB samples 2.93 -> 3.04)
    p.trace_carry = (c->n_samples && c->n_bytes / c->n_samples < 2048) ? 1u : 0u;
    if (const char* e = knob("TGX_TRACE_CARRY")) p.trace_carry = atoi(e) ? 1u : 0u;
    p.endmask = c->d_endmask;
    p.mword = c->d_mword;
    p.mask_words = c->mask_words;
    p.prefix = c->d_prefix;
    {
        // timing experiments (tools/ablate.py) — results are WRONG when set; honoured only with TGX_DEBUG=1
        const char* f = debug_on() ? getenv("TGX_FLAGS") : nullptr;
        p.flags = f ? (uint32_t)atoi(f) : 0u;
    }
    // TGX_PATH=fused forces the one-sample-per-wave kernel (A/B timing, tests of both paths)
    const char* force = knob("TGX_PATH");
    const bool use4 = m->lm <= 16 && m->scores_finite && m->d_tokhash && !(force && strcmp(force, "fused") == 0);
    // vocabularies with tokens of 17..32 bytes (after `merge`): two samples per wave (encode2.hip)
    const bool use2 = !use4 && m->lm <= 32 && m->scores_finite && m->d_tokhash && !(force && strcmp(force, "fused") == 0);
    // encode5_kernel / encode6_kernel (8-byte label-checked records, match indices = ranks of score values, the
    // hottest values in LDS and the rest read from L2 by the relaxing lanes) for every vocabulary with finite scores
    // and at most 65 535 distinct score values — round 3: that includes vocabularies in which every token has its own
    // score (after an M-step or merge: any trained vocabulary) and vocabularies with tokens of 17..32 bytes (after
    // `merge`: the LONG build of encode5_kernel).  encode4_kernel / encode4l_kernel remain for more distinct values
    // than that (the 500 000-entry stages of prune).  TGX_PATH=rows4 / rows5 / rows2 force a kernel (A/B timing,
    // tests of every path).
    const bool long_tokens = m->lm > 16;
    const bool use5 = (use4 || use2) && m->have_trie8 &&
                      !(force && (strcmp(force, "rows4") == 0 || strcmp(force, "rows2") == 0 || strcmp(force, "rows4l") == 0));
    if (debug_on())
        fprintf(stderr, "[tgx] encode: S=%llu N=%llu lm=%u path=%s slots=%zu root_base=%u values=%u\n",
                (unsigned long long)c->n_samples, (unsigned long long)c->n_bytes, p.lm, use4 ? (use5 ? "rows5" : "rows4") : (use2 ? "rows2" : "fused"),
                m->flat.table.size(), p.root_base, m->n_values);
    if (use5) {
        {
            const tgx_status rst = ensure_value_ranks(m, c);
            if (rst != TGX_OK) return rst;
        }
        // One block of sixteen waves per CU, and the block's LDS (160 KiB) is shared by the match-index buffers (2 KiB
        // per wave and 16 positions per lane) and the copy of the hottest score values.  Three geometries, by the
        // number of distinct score values (1 GiB of the bench corpus, profiles/r03):
        //   * four positions per lane (the four staggered walks of a lane hide most of each other's gather latency,
        //     encode5.hip: Walk5) with EVERY value in LDS: up to ~3 700 values, 11.0 ms;
        //   * the same with 15 / 14 / 13 waves, every value in LDS: up to ~4 700 / 5 700 / 6 800 values, 11.4 - 12.3 ms;
        //   * two positions per lane, every value in LDS: up to ~11 900 values — the generate-style vocabularies
        //     of SURVEY.md 8(d), 9 652 values at 32 000 entries, 10 569 at 65 536 —, 12.7 ms;
        //   * three positions per lane x 13 waves, every value in LDS: up to ~10 100 values, 12.5 ms;
        //   * three positions per lane with the ~7 800 hottest values in LDS and the others read from L2 by the
        //     relaxing lanes (COLD builds; every token its own score: after an M-step or merge), 14.0 ms.
        // Fewer waves when the batch has fewer samples than the chip has rows, so that they spread over the CUs.
        // Tokens of 17..32 bytes: the LONG build (four positions per lane, a list of long matches per wave).
        m->last_redo_samples = 0;
        // Long samples first, four to a block (encode6_kernel: three walker waves per sample ahead of one row of the
        // block's relaxing wave), when that shortens the pass.  encode5_kernel takes ~0.104 us per byte of a
        // sample's serial chain (6.8 ms per 64 KiB) and ~1 s per 88 GB of batch; encode6_kernel ~0.037 us per byte
        // of chain (2.4 ms per 64 KiB) but ~55 GB/s once every CU has its two blocks (eight relaxing rows per CU;
        // profiles/r02: e6 shapes).  The two run one after the other, so the split is chosen among the powers of
        // two as thresholds by the sum of the two estimates; TGX_LONG_THRESHOLD forces one (0: never).
        // Its LDS copy of the value table is what two blocks per CU leave room for (~4 500 values).
        int e6_bpc = 2;
        if (const char* e = knob("TGX_E6_BPC")) e6_bpc = std::min(4, std::max(1, atoi(e)));
        uint32_t n_hot6 = std::min(m->n_values, tgx::encode6_max_hot(160u * 1024u / (uint32_t)e6_bpc, 0u));
        // More values than that: the walkers fetch the values beyond the LDS copy into `pool6` pool entries per ring
        // slot (encode5.hip: Res5), and the relaxer reads from L2 only what a full pool left behind.
        // 128 entries per slot (64 positions): 64 MiB / 256 MiB of the bench corpus take 2.70 / 6.06 ms with the
        // 9 652-value vocabulary and 2.73 / 6.12 ms with 32 000 values (64 entries: 2.91 / 6.20 and 3.69 / 7.54;
        // no pool, round 2: 4.88 / 9.82 and 5.04 / 10.11; every value in LDS: 2.64 / 5.45 — profiles/r03).
        uint32_t pool6 = 0;
        bool e6_usable = true;
        if (n_hot6 < m->n_values) {
            pool6 = 128;
            if (const char* e = knob("TGX_E6_POOL")) pool6 = (uint32_t)std::min(192, std::max(0, atoi(e)));
            if ((uint64_t)m->n_values + tgx::encode6_pool_total(pool6) > 65535ull) {
                // indices are 16 bits: no room for pool entries, and a relaxer that reads the cold values itself is
                // slower than encode5_kernel on everything but a lone long sample
                pool6 = 0;
                e6_usable = knob("TGX_E6_POOL") != nullptr || knob("TGX_LONG_THRESHOLD") != nullptr;
            }
            n_hot6 = std::min(m->n_values, tgx::encode6_max_hot(160u * 1024u / (uint32_t)e6_bpc, pool6));
        }
        if (const char* e = knob("TGX_E5_HOT")) {
            const int v = atoi(e);
            if (v >= 0) n_hot6 = std::min(n_hot6, (uint32_t)v);
        }
        const bool cold6 = n_hot6 < m->n_values;
        uint64_t n_long = 0;
        uint32_t corun_cus = 0;  // CUs of encode6_kernel when both kernels run at once (0: one after the other)
        if (c->n_samples && !long_tokens && e6_usable) {  // (encode6_kernel walks 16 bytes)
            const auto count_ge = [&](uint64_t thr) {  // h_sorted_len descends: the long samples are a prefix of the order
                return (uint64_t)(std::partition_point(c->h_sorted_len.begin(), c->h_sorted_len.end(),
                                                       [thr](uint32_t len) { return len >= thr; }) - c->h_sorted_len.begin());
            };
            if (const char* e = knob("TGX_LONG_THRESHOLD")) {
                const uint64_t thr = (uint64_t)std::max(0ll, atoll(e));
                if (thr) n_long = count_ge(thr);
            } else {
                const double N = (double)c->n_bytes;
                // encode5_kernel's chain of a long sample on the few waves such a batch gets: 7.1 ms per 64 KiB with every
                // value in LDS, 8.7 ms with the COLD build (profiles/r03/l_*, o_*)
                const double c5 = m->n_values > tgx::encode5_max_hot(false, 8, 4, 160u * 1024u) ? 0.135e-6 : 0.108e-6;
                auto cost = [&](uint64_t k) {  // the k longest samples to encode6_kernel
                    const double bytes_long = k ? (double)c->h_sorted_cum[k - 1] : 0.0;
                    const double t6 = k ? std::max((double)c->h_sorted_len[0] * (cold6 ? 0.042e-6 : 0.0369e-6), bytes_long / (cold6 ? 46e9 : 55e9)) + 20e-6 : 0.0;
                    const double rest_max = k < c->n_samples ? (double)c->h_sorted_len[k] : 0.0;
                    const double t5 = std::max(rest_max * c5, (N - bytes_long) / 88e9);
                    return t6 + t5;
                };
                double best = cost(0);
                for (uint64_t thr = 1024; thr <= c->max_len; thr *= 2) {
                    const uint64_t k = count_ge(thr);
                    const double ck = cost(k);
                    if (k && ck < best * 0.9) {  // switch for a clear gain only
                        best = ck;
                        n_long = k;
                    }
                }
                // Both kernels at once, each on CUs of its own: between ~200 and ~450 MiB neither wins alone — the
                // long-sample kernel is bound by its eight relaxing rows per CU, encode5_kernel by the chains of the
                // longest samples.  `corun_cus` CUs (two blocks each) take the samples of at least thr bytes, the others
                // run encode5_kernel on the rest; its blocks are launched first and ask for more than half of a CU's
                // LDS, so no block of the other kernel shares their CU.  Estimates: a long-sample row does
                // 1 / 0.0369 us (0.042 us with cold values) per byte, a CU 55 (46) GB/s / 256; encode5_kernel 0.12 us
                // per byte of chain (measured beside the other kernel) and 70 GB/s x its share of the CUs.
                const double c6 = cold6 ? 0.042e-6 : 0.0369e-6, r6 = (cold6 ? 46e9 : 55e9) / (double)m->num_cus;
                double best_co = best * 0.9;
                uint64_t k_co = 0;
                static const uint64_t thrs[] = {8192, 12288, 16384, 24576, 32768, 40960, 49152};
                static const uint32_t shares[] = {32, 48, 64, 96, 128, 160, 192};
                // (only when encode5_kernel keeps every value in LDS on its few waves: its COLD builds are bound by the
                // chains of mid-length samples at twice the estimate — profiles/r03/p_corun_sweep*.txt)
                // (and never again after a pass whose host wait for encode5_kernel's blocks timed out: the device's atomic to
                // mapped host memory was not delivered, every co-run would pay the full 2 ms and lose the CU separation)
                const bool corun_off = (knob("TGX_CORUN") && atoi(knob("TGX_CORUN")) == 0) || m->corun_wait_timeouts != 0 ||
                                       m->n_values > tgx::encode5_max_hot(false, 8, 4, 160u * 1024u);
                for (uint64_t thr : thrs) {
                    if (corun_off || thr > c->max_len || e6_bpc != 2) break;
                    const uint64_t k = count_ge(thr);
                    if (!k || k >= c->n_samples) continue;
                    const double bytes_long = (double)c->h_sorted_cum[k - 1];
                    const double r_max = (double)c->h_sorted_len[k], r_bytes = N - bytes_long;
                    for (uint32_t X : shares) {
                        if ((int)X >= m->num_cus) break;
                        const double t6 = std::max((double)c->h_sorted_len[0] * c6, bytes_long / (X * r6)) + 20e-6;
                        const double t5 = std::max(r_max * 0.12e-6, r_bytes / ((double)(m->num_cus - (int)X) / (double)m->num_cus * 70e9));
                        const double t = std::max(t5, t6);
                        if (t < best_co) {
                            best_co = t;
                            k_co = k;
                            corun_cus = X;
                        }
                    }
                }
                if (corun_cus) n_long = k_co;
            }
            if (const char* e = knob("TGX_CORUN")) {  // forces the share of the long-sample kernel (0: no co-run)
                const int v = atoi(e);
                corun_cus = (v > 0 && v < m->num_cus && n_long > 0 && n_long < c->n_samples && e6_bpc == 2) ? (uint32_t)v : 0u;
            }
        }
        const int cus5 = m->num_cus - (int)corun_cus;  // CUs of encode5_kernel
        // what is left for encode5_kernel: the samples from n_long on in the longest-first order
        const uint64_t rest_n = c->n_samples - n_long;
        const uint64_t rest_bytes = c->n_bytes - (n_long ? c->h_sorted_cum[n_long - 1] : 0);
        const uint64_t rest_max = rest_n ? c->h_sorted_len[n_long] : 0;
        int ppl = 4, bpc = 1, hot_waves = 16;  // hot_waves: the most waves beside which every value fits
        {
            int ps4 = 0;
            HIP_TRY(tgx::encode5_waves_per_simd(dropout > 0.0, false, 4, long_tokens, &ps4));
            hot_waves = std::min(16, ps4 * 4);
            const int least = long_tokens ? 12 : 13;  // (the long-token build has four positions per lane only)
            while (hot_waves >= least && m->n_values > tgx::encode5_max_hot(long_tokens, hot_waves, 4, 160u * 1024u)) hot_waves--;
            if (hot_waves < least) {
                hot_waves = 0;  // not with four positions per lane
                if (!long_tokens) {
                    // three positions per lane x 13 waves with every value in LDS (up to ~10 100: the 32 000-entry spec
                    // vocabulary, 12.5 ms against 12.8 with two positions per lane); two x 16 (up to ~11 900: the
                    // 65 536-entry one, 13.0 ms); else three x 16 with the ~7 800 hottest in LDS (every token its own
                    // score: 14.0 ms against 14.4 with four positions per lane and 3 712) — profiles/r03/n_e5_ppl3_sweep.txt
                    if (m->n_values <= tgx::encode5_max_hot(false, 13, 3, 160u * 1024u)) {
                        ppl = 3;
                        hot_waves = 13;
                    } else if (m->n_values <= tgx::encode5_max_hot(false, 16, 2, 160u * 1024u)) {
                        ppl = 2;
                    } else {
                        ppl = 3;
                    }
                }
            }
        }
        // A batch whose longest sample is more than a row's share of it is not bound by throughput but by the rows that
        // drew the long samples: with rows = bytes / longest sample every row's share is one such sample, fewer waves run
        // each faster, and their LDS takes every value with four positions per lane (the longest trips).  512 MiB of the
        // bench corpus: 8 waves, 9.7 ms against 10.5 (three positions x 13 waves); with distinct scores 11.3 against 12.5
        // (profiles/r03/o_split_and_geometry_by_size.txt).  From 11 waves on the throughput geometries above are faster.
        int balance_waves = 0;
        if (rest_max > 0) {
            const uint64_t rows_bal = (rest_bytes + rest_max - 1) / rest_max;
            const uint64_t wb = (rows_bal + 4ull * (uint64_t)cus5 - 1) / (4ull * (uint64_t)cus5);
            if (wb <= 10) {
                balance_waves = (int)std::max<uint64_t>(4, wb);
                ppl = 4;
                hot_waves = 0;
            }
        }
        if (const char* e = long_tokens ? nullptr : knob("TGX_PPL")) {
            const int v = atoi(e);
            if (v >= 1 && v <= 4) {
                ppl = v;
                bpc = ppl >= 3 ? 1 : 2;
            }
        }
        if (const char* e = knob("TGX_BPC")) {
            const int v = atoi(e);
            if (v >= 1 && v <= 8) bpc = v;
        }
        const uint32_t budget = 160u * 1024u / (uint32_t)bpc;
        bool cold = false;
        int per_simd = 0;
        HIP_TRY(tgx::encode5_waves_per_simd(dropout > 0.0, false, ppl, long_tokens, &per_simd));
        int waves = std::min(16, (per_simd / bpc) * 4);
        if ((ppl == 4 || ppl == 3) && bpc == 1 && hot_waves > 0) waves = std::min(waves, hot_waves);
        if (m->n_values > tgx::encode5_max_hot(long_tokens, waves, ppl, budget)) {
            cold = true;
            HIP_TRY(tgx::encode5_waves_per_simd(dropout > 0.0, true, ppl, long_tokens, &per_simd));
            waves = std::min(16, (per_simd / bpc) * 4);
        }
        if (balance_waves > 0 && ppl == 4 && bpc == 1) waves = std::min(waves, balance_waves);
        {
            const uint64_t rows_wanted = (rest_n + (uint64_t)cus5 * bpc - 1) / ((uint64_t)cus5 * bpc);
            waves = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)waves, (rows_wanted + 3) / 4));
        }
        if (const char* e = knob("TGX_WAVES")) {
            const int v = atoi(e);
            if (v >= 1 && v <= 16) waves = v;
        }
        while (waves > 1 && tgx::encode5_max_hot(long_tokens, waves, ppl, budget) < 16u) waves--;
        uint32_t n_hot = std::min(m->n_values, tgx::encode5_max_hot(long_tokens, waves, ppl, budget));
        if (const char* e = knob("TGX_E5_HOT")) {  // tests of the COLD builds (a small LDS copy), table-size sweeps
            const int v = atoi(e);
            if (v >= 0) n_hot = std::min(n_hot, (uint32_t)v);
        }
        cold = n_hot < m->n_values;
        m->last_n_hot = n_hot;
        // whole blocks only: a block's waves are dealt round-robin to the SIMDs, ceil(waves / 4) on the fullest
        m->last_encode_waves_per_cu = std::min(bpc, per_simd / ((waves + 3) / 4)) * waves;
        const uint64_t rows_per_block = 4 * (uint64_t)waves;
        const uint32_t blocks5 = (uint32_t)std::max<uint64_t>(
            1, std::min<uint64_t>((rest_n + rows_per_block - 1) / rows_per_block, (uint64_t)cus5 * bpc));
        tgx::Encode5Params q{};
        q.trie8 = m->d_trie8;
        q.trie_bytes = (uint32_t)(m->flat.table.size() * sizeof(tgx::Trie8Rec));
        q.values = m->d_values;
        q.root_base = m->root_base8;
        q.n_values = m->n_values;
        q.n_hot = n_hot;
        {   // rows claim several consecutive samples of the order per atomic when samples are short: one global
            // atomic round trip (~1-2 us) per sample is what a corpus of 130-byte samples otherwise waits for
            const uint64_t avg = c->n_samples ? c->n_bytes / c->n_samples : 0;
            q.claim_chunk = (uint32_t)std::min<uint64_t>(16, std::max<uint64_t>(1, 4096 / std::max<uint64_t>(1, avg)));
            if (const char* e = knob("TGX_CLAIM_CHUNK")) q.claim_chunk = (uint32_t)std::min(64, std::max(1, atoi(e)));
        }
        m->last_long_samples = n_long;
        m->last_corun_cus = n_long ? corun_cus : 0u;
        tgx::EncodeParams p6 = p;
        tgx::Encode5Params q6 = q;
        uint32_t blocks6 = 0;
        if (n_long) {
            p6.n_samples = n_long;
            // four samples per block (one per row of its relaxing wave); 13 waves of 64 registers: two blocks per CU
            blocks6 = (uint32_t)std::min<uint64_t>((n_long + 3) / 4, (uint64_t)(corun_cus ? (int)corun_cus : m->num_cus) * (uint64_t)e6_bpc);
            q6.n_hot = n_hot6;
            q6.pool = cold6 ? pool6 : 0u;
            if (!corun_cus) {
                time_begin(m, "encode6_kernel");
                HIP_TRY(tgx::launch_encode6(p6, q6, cold6, blocks6, m->stream));
                time_end(m);
                HIP_TRY(hipMemsetAsync(m->d_ctrl, 0x00, 8, m->stream));  // the work queue, for encode5_kernel
            } else {
                if (!m->stream2) {
                    HIP_TRY(hipStreamCreateWithFlags(&m->stream2, hipStreamNonBlocking));
                    HIP_TRY(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
                    HIP_TRY(hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming));
                    HIP_TRY(hipHostMalloc((void**)&m->h_started, 64, hipHostMallocMapped | hipHostMallocPortable));
                }
                *reinterpret_cast<volatile unsigned int*>(m->h_started) = 0u;
                q.started = m->h_started;
                p6.queue = m->d_ctrl + 2;  // a work queue of its own
                HIP_TRY(hipMemsetAsync(m->d_ctrl + 2, 0x00, 8, m->stream));
                HIP_TRY(hipEventRecord(m->ev_fork, m->stream));
                HIP_TRY(hipStreamWaitEvent(m->stream2, m->ev_fork, 0));
            }
            p.order = c->d_order + n_long;
            p.n_samples = c->n_samples - n_long;
        }
        unsigned long long* d_stamps5 = nullptr;
        const size_t n_stamp_waves5 = (size_t)blocks5 * (size_t)waves;
        if (const char* e = debug_on() ? getenv("TGX_STAMPS") : nullptr) {
            if (*e == '1') {
                if (pool_alloc(m->device, n_stamp_waves5 * 64, (void**)&d_stamps5) != hipSuccess) return fail(TGX_ERR_DEVICE, "out of device memory (stamps)");
                if (hipMemsetAsync(d_stamps5, 0, n_stamp_waves5 * 64, m->stream) != hipSuccess) {
                    pool_free(m->device, d_stamps5, n_stamp_waves5 * 64);
                    return fail(TGX_ERR_DEVICE, "stamps reset failed");
                }
                p.stamps = d_stamps5;
            }
        }
        time_begin(m, "encode5_kernel");
        {
            if (long_tokens) {  // samples whose wave ran out of list entries for long matches go to encode2_kernel
                p.redo_count = m->d_ctrl + 6;
                p.redo_list = c->d_counts;  // free until the trace writes the token counts
                HIP_TRY(hipMemsetAsync(m->d_ctrl + 6, 0x00, 8, m->stream));
            }
            // (co-run: more than half of the CU's LDS, so that no block of the long-sample kernel shares the CU)
            const hipError_t le = tgx::launch_encode5(p, q, cold, ppl, long_tokens, waves, blocks5, corun_cus && n_long ? 84u * 1024u : 0u, m->stream);
            if (le != hipSuccess) {
                if (d_stamps5) {
                    (void)hipStreamSynchronize(m->stream);
                    pool_free(m->device, d_stamps5, n_stamp_waves5 * 64);
                }
                return fail(TGX_ERR_DEVICE, "encode5 launch failed: %s", hipGetErrorString(le));
            }
        }
        bool joined = false, joined_slot = false;
        if (n_long && corun_cus) {  // the long-sample kernel beside it, on the second stream, with a timing slot of its own
            const bool slot = m->n_timed + 1 < kMaxTimed;
            if (slot) {
                KernelTime& tb = m->timed[m->n_timed + 1];
                tb.name = "encode6_kernel";
                tb.used = true;
                (void)hipEventRecord(tb.start, m->stream2);
            }
            {   // the long-sample kernel's blocks must find the CUs of encode5_kernel's blocks taken: wait (at most 2 ms)
                // until every one of those has reported itself resident
                const auto t0 = std::chrono::steady_clock::now();
                volatile unsigned int* started = m->h_started;
                bool timed_out = false;
                while (*started < blocks5 && !(timed_out = std::chrono::steady_clock::now() - t0 >= std::chrono::milliseconds(2))) {
                }
                if (timed_out && *started < blocks5) m->corun_wait_timeouts++;
            }
            const hipError_t l6 = tgx::launch_encode6(p6, q6, cold6, blocks6, m->stream2);
            if (slot) (void)hipEventRecord(m->timed[m->n_timed + 1].stop, m->stream2);
            (void)hipEventRecord(m->ev_join, m->stream2);
            if (l6 != hipSuccess) {
                (void)hipStreamSynchronize(m->stream);
                (void)hipStreamSynchronize(m->stream2);
                return fail(TGX_ERR_DEVICE, "encode6 launch failed: %s", hipGetErrorString(l6));
            }
            joined = true;
            joined_slot = slot;
        }
        time_end(m);
        if (joined) {
            if (joined_slot) m->n_timed++;
            HIP_TRY(hipStreamWaitEvent(m->stream, m->ev_join, 0));  // the trace needs both kernels' back-pointers
        }
        if (d_stamps5) {  // diagnostic: mean ticks per iteration and phase over all waves
            std::vector<unsigned long long> h(n_stamp_waves5 * 8);
            const bool ok = hipStreamSynchronize(m->stream) == hipSuccess &&
                            hipMemcpy(h.data(), d_stamps5, n_stamp_waves5 * 64, hipMemcpyDeviceToHost) == hipSuccess;
            pool_free(m->device, d_stamps5, n_stamp_waves5 * 64);
            p.stamps = nullptr;
            if (!ok) return fail(TGX_ERR_DEVICE, "stamps copy failed");
            double sum[5] = {0, 0, 0, 0, 0}, iters = 0;
            for (size_t w = 0; w < n_stamp_waves5; w++) {
                for (int i = 0; i < 5; i++) sum[i] += (double)h[w * 8 + i];
                iters += (double)h[w * 8 + 5];
            }
            fprintf(stderr, "[tgx] encode5 stamps (s_memtime ticks per wave-iteration, %zu waves x ppl %d, %.0f iterations): switch %.0f  text+reset %.0f  walk %.0f  relax %.0f  store %.0f\n",
                    n_stamp_waves5, ppl, iters, sum[0] / iters, sum[1] / iters, sum[2] / iters, sum[3] / iters, sum[4] / iters);
        }
        if (long_tokens) {
            unsigned long long n_redo = 0;
            HIP_TRY(hipMemcpyAsync(&n_redo, m->d_ctrl + 6, 8, hipMemcpyDeviceToHost, m->stream));
            HIP_TRY(hipStreamSynchronize(m->stream));
            if (n_redo > c->n_samples) return fail(TGX_ERR_DEVICE, "redo list longer than the batch");
            m->last_redo_samples = n_redo;
            if (n_redo) {
                tgx::EncodeParams q2 = p;
                q2.order = c->d_counts;
                q2.n_samples = n_redo;
                HIP_TRY(hipMemsetAsync(m->d_ctrl, 0x00, 8, m->stream));  // the work queue
                time_begin(m, "encode2_kernel");
                HIP_TRY(tgx::launch_encode2(q2, (uint32_t)m->num_cus, true, m->stream));
                time_end(m);
            }
        }
        p.order = c->d_order;
        p.n_samples = c->n_samples;
        const uint32_t blocks_t =
            (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((c->n_samples + 3) / 4, (uint64_t)m->num_cus * 8));
        time_begin(m, m->mask_path ? "mark_kernel" : long_tokens ? "trace32_kernel" : "trace_kernel");
        if (m->mask_path) HIP_TRY(tgx::launch_mark(p, blocks_t, long_tokens ? 32u : 16u, true, m->stream));
        else if (long_tokens) HIP_TRY(tgx::launch_trace32(p, blocks_t, true, m->stream));
        else HIP_TRY(tgx::launch_trace(p, blocks_t, m->stream));
        time_end(m);
    } else if (use4) {
        // Vocabularies without 8-byte records (more than 65 535 distinct score values): round 1's kernel over the
        // 16-byte records, scores through the match buffer.
        // `bpc` blocks per CU of `waves` waves each (8 KiB of LDS per wave and position group).
        // TGX_PPL (positions per lane: 1, 2, 4), TGX_WAVES, TGX_BPC override the defaults (with TGX_DEBUG=1).
        // Two blocks of ten waves per CU (20 x 8 KiB of LDS); rows claim samples dynamically, so the
        // geometry only has to fill the CU.
        int ppl = 1, waves = 10, bpc = 2;
        // Positions per lane.  A sample is a serial chain of 16 * ppl positions per trip, so the longest
        // sample bounds the pass from below, while more positions per lane cost LDS and with it waves per CU.
        // Measured on 1 x MI355X (tools/ppl_sweep.py): throughput 60 / 50 / 34 GB/s and 13.5 / 10.7 / 9.1 ms per
        // 64 KiB of chain for ppl = 1 / 2 / 4; take the smallest estimate.
        const double gbps4[3] = {60.0, 50.0, 34.0}, chain4_ms[3] = {13.5, 10.7, 9.1};
        auto cost4 = [&](double bytes, double longest, int* ppl_out) {
            double best_t = 0;
            int best_ppl = 1;
            for (int i = 0; i < 3; i++) {
                const double t = std::max(bytes / (gbps4[i] * 1e9), longest / 65536.0 * chain4_ms[i] * 1e-3);
                if (i == 0 || t < best_t * 0.95) {  // switch only for a clear gain
                    best_t = t;
                    best_ppl = 1 << i;
                }
            }
            if (ppl_out) *ppl_out = best_ppl;
            return best_t;
        };
        m->last_long_samples = 0;
        m->last_redo_samples = 0;
        cost4((double)c->n_bytes, c->n_samples ? (double)c->h_sorted_len[0] : 0.0, &ppl);
        if (const char* e = knob("TGX_PPL")) {  // (also set by tests: every positions-per-lane build against the oracle)
            const int v = atoi(e);
            if (v == 1 || v == 2 || v == 4) ppl = v;
        }
        if (ppl == 2) { waves = 5; bpc = 2; }
        if (ppl == 4) { waves = 5; bpc = 1; }
        if (const char* e = knob("TGX_WAVES")) {
            const int v = atoi(e);
            if (v >= 1 && v <= 16) waves = v;
        }
        if (const char* e = knob("TGX_BPC")) {
            const int v = atoi(e);
            if (v >= 1 && v <= 8) bpc = v;
        }
        bool root = ppl == 1;  // first trie level in LDS (4 KiB per block)
        if (const char* e = knob("TGX_ROOT")) root = root && atoi(e) != 0;
        while (waves > 1 && tgx::encode4_lds_bytes(waves, ppl, root) > (160u * 1024u) / (uint32_t)bpc) waves--;
        {
            // The geometry must really fit: a block's waves are dealt round-robin to the four SIMDs, so
            // `bpc` blocks put bpc * ceil(waves / 4) waves on some SIMD, which its registers must allow
            // (see encode4_kernel's note); otherwise five blocks of four waves: one wave per SIMD and block.
            int per_simd = 0;
            HIP_TRY(tgx::encode4_waves_per_simd(dropout > 0.0, ppl, root, &per_simd));
            if (bpc * ((waves + 3) / 4) > per_simd && ppl == 1) {
                waves = 4;
                bpc = std::min(5, per_simd);
                root = false;
                HIP_TRY(tgx::encode4_waves_per_simd(dropout > 0.0, ppl, root, &per_simd));
            }
            m->last_encode_waves_per_cu = std::min(bpc * ((waves + 3) / 4), per_simd) * 4 >= bpc * waves
                                              ? bpc * waves
                                              : per_simd / ((waves + 3) / 4) * waves;
        }
        const uint64_t rows_per_block = 4 * (uint64_t)waves;
        const uint32_t blocks4 = (uint32_t)std::max<uint64_t>(
            1, std::min<uint64_t>((p.n_samples + rows_per_block - 1) / rows_per_block, (uint64_t)m->num_cus * bpc));
        unsigned long long* d_stamps = nullptr;
        const size_t n_stamp_waves = (size_t)blocks4 * (size_t)waves;
        // (stamp buffers of the diagnostic runs: a failing call between allocation and release ends the process's
        // usefulness for diagnosis anyway; they are only ever allocated with TGX_DEBUG=1)
        if (const char* e = debug_on() ? getenv("TGX_STAMPS") : nullptr) {
            if (*e == '1' && dropout <= 0.0 && ppl == 1) {
                HIP_TRY(hipMalloc((void**)&d_stamps, n_stamp_waves * 64));
                HIP_TRY(hipMemsetAsync(d_stamps, 0, n_stamp_waves * 64, m->stream));
                p.stamps = d_stamps;
            }
        }
        time_begin(m, "encode4_kernel");
        HIP_TRY(tgx::launch_encode4(p, ppl, waves, blocks4, root, m->stream));
        time_end(m);
        if (d_stamps) {  // diagnostic: mean cycles per iteration and phase over all waves
            std::vector<unsigned long long> h(n_stamp_waves * 8);
            HIP_TRY(hipStreamSynchronize(m->stream));
            HIP_TRY(hipMemcpy(h.data(), d_stamps, n_stamp_waves * 64, hipMemcpyDeviceToHost));
            double sum[5] = {0, 0, 0, 0, 0}, iters = 0;
            for (size_t w = 0; w < n_stamp_waves; w++) {
                for (int i = 0; i < 5; i++) sum[i] += (double)h[w * 8 + i];
                iters += (double)h[w * 8 + 5];
            }
            fprintf(stderr, "[tgx] stamps (s_memtime ticks per wave-iteration, %zu waves, %.0f iterations): switch %.0f  text %.0f  walk %.0f  relax %.0f  store %.0f\n",
                    n_stamp_waves, iters, sum[0] / iters, sum[1] / iters, sum[2] / iters, sum[3] / iters, sum[4] / iters);
            (void)hipFree(d_stamps);
            p.stamps = nullptr;
        }
        p.order = c->d_order;
        p.n_samples = c->n_samples;
        const uint32_t blocks_t =
            (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((c->n_samples + 3) / 4, (uint64_t)m->num_cus * 8));
        time_begin(m, m->mask_path ? "mark_kernel" : "trace_kernel");
        if (m->mask_path) HIP_TRY(tgx::launch_mark(p, blocks_t, 16u, true, m->stream));
        else HIP_TRY(tgx::launch_trace(p, blocks_t, m->stream));
        time_end(m);
    } else if (use2) {
        // Tokens of 17..32 bytes: the 16-lane rows with an overflow list for the long matches (encode4l.hip);
        // samples whose wave ran out of overflow entries are redone two per wave on 32-lane rows (encode2.hip),
        // as is the whole batch with TGX_PATH=rows2.
        const uint32_t blocks_t =
            (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((c->n_samples + 3) / 4, (uint64_t)m->num_cus * 8));
        const bool rows2 = force && strcmp(force, "rows2") == 0;
        m->last_redo_samples = 0;
        if (!rows2) {
            p.redo_count = m->d_ctrl + 6;
            p.redo_list = c->d_counts;  // free until the trace writes the token counts
            HIP_TRY(hipMemsetAsync(m->d_ctrl + 6, 0x00, 8, m->stream));
            time_begin(m, "encode4l_kernel");
            HIP_TRY(tgx::launch_encode4l(p, (uint32_t)m->num_cus, m->stream));
            time_end(m);
            unsigned long long n_redo = 0;
            HIP_TRY(hipMemcpyAsync(&n_redo, m->d_ctrl + 6, 8, hipMemcpyDeviceToHost, m->stream));
            HIP_TRY(hipStreamSynchronize(m->stream));
            if (n_redo > c->n_samples) return fail(TGX_ERR_DEVICE, "redo list longer than the batch");
            m->last_redo_samples = n_redo;
            if (n_redo) {
                tgx::EncodeParams q = p;
                q.order = c->d_counts;
                q.n_samples = n_redo;
                HIP_TRY(hipMemsetAsync(m->d_ctrl, 0x00, 8, m->stream));  // the work queue
                time_begin(m, "encode2_kernel");
                HIP_TRY(tgx::launch_encode2(q, (uint32_t)m->num_cus, true, m->stream));
                time_end(m);
            }
        } else {
            time_begin(m, "encode2_kernel");
            HIP_TRY(tgx::launch_encode2(p, (uint32_t)m->num_cus, false, m->stream));
            time_end(m);
        }
        time_begin(m, m->mask_path ? "mark_kernel" : "trace32_kernel");
        if (m->mask_path) HIP_TRY(tgx::launch_mark(p, blocks_t, 32u, !rows2, m->stream));
        else HIP_TRY(tgx::launch_trace32(p, blocks_t, !rows2, m->stream));
        time_end(m);
    } else {
        time_begin(m, "encode_kernel");
        HIP_TRY(tgx::launch_encode(p, grid_blocks(m, c->n_samples), m->stream));
        time_end(m);
    }
    return TGX_OK;
}

tgx_status check_no_path(tgx_model* m, const tgx_corpus* c) {
    unsigned long long bad = m->h_ctrl[0];
    if (bad == ~0ULL) return TGX_OK;
    if (bad & (1ULL << 62))
        return fail(TGX_ERR_DEVICE, "internal error: corrupt back-pointer (sample or text byte %llu)",
                    (unsigned long long)(bad & ~(1ULL << 62)));
    uint64_t n = c->h_offs[bad + 1] - c->h_offs[bad];
    g_err_sample = bad;
    g_err_pos = n;
    g_err_len = n;
    // Display of Error::NoPath, reference src/lib.rs:243-245
    return fail(TGX_ERR_NO_PATH, "no path to position %llu/%llu", (unsigned long long)n,
                (unsigned long long)n);
}

}  // namespace

extern "C" {

const char* tgx_last_error(void) { return g_err_msg.c_str(); }

void tgx_last_error_detail(uint64_t* sample, uint64_t* pos, uint64_t* len) {
    if (sample) *sample = g_err_sample;
    if (pos) *pos = g_err_pos;
    if (len) *len = g_err_len;
}

int tgx_abi_version(void) { return TGX_ABI_VERSION; }
int tgx_device_count(void) { return usable_device_count(); }
void tgx_free(void* p) { free(p); }
void tgx_pool_trim(int device) { pool_trim(device); }
void* tgx_host_alloc(uint64_t bytes) {
    if (usable_device_count() <= 0) {
        fail(TGX_ERR_DEVICE, "no usable HIP device (gfx950 required)");
        return nullptr;
    }
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? (size_t)bytes : 1, hipHostMallocPortable) != hipSuccess) {
        (void)hipGetLastError();
        fail(TGX_ERR_DEVICE, "out of page-locked host memory (%llu bytes)", (unsigned long long)bytes);
        return nullptr;
    }
    return p;
}
void tgx_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

// Tables that only the encode path reads (built at creation, or at the first encode of a model created with
// TGX_MODEL_FOR_ESTEP: prune builds two such models per iteration and never encodes with them).  Caller holds
// m->mu (or is the creating thread).
static tgx_status ensure_encode_tables(tgx_model* m) {
    if (m->encode_tables_ready) return TGX_OK;
    HIP_TRY(hipSetDevice(m->device));
    // the bytes -> id table and the 8-byte records are independent: the first is built on a second host thread
    // while this one builds the second (each ~50 ms at 500 000 tokens)
    tgx::HostPhases hp("ensure_encode_tables");
    const bool want_hash = m->lm <= 32 && m->scores_finite;
    const bool want_trie8 = m->lm <= 32 && m->scores_finite && m->flat.table.size() <= tgx::kTrie8MaxSlots;
    std::thread hash_builder;
    if (want_hash && !m->tokhash_host_built)
        hash_builder = std::thread([m]() { tgx::build_tok_hash(m->vocab_bytes.data(), m->vocab_offs.data(), m->vocab_size, &m->tokhash); });
    tgx::Trie8 t8;
    if (want_trie8) tgx::build_trie8(m->flat, m->vocab_offs.data(), m->vocab_scores.data(), &t8);
    if (hash_builder.joinable()) hash_builder.join();
    hp.mark("trie8 | tok hash");
    // (a failure part-way leaves the tables that exist in place, pointers set: the next call finds them — nothing is
    // allocated twice — and tgx_model_destroy frees them)
    if (want_hash && m->tokhash.ok && !m->d_tokhash) {
        const size_t hb = m->tokhash.slots.size() * sizeof(tgx::TokHashEntry);
        HIP_TRY(hipMalloc(&m->d_tokhash, hb));
        HIP_TRY(hipMemcpyAsync(m->d_tokhash, m->tokhash.slots.data(), hb, hipMemcpyHostToDevice, m->stream));
    }
    if (want_trie8 && t8.ok && m->d_tokhash) {
        const size_t ns = t8.rec.size();
        if (!m->d_trie8) HIP_TRY(hipMalloc(&m->d_trie8, ns * sizeof(tgx::Trie8Rec)));
        if (!m->d_values) HIP_TRY(hipMalloc((void**)&m->d_values, t8.values.size() * 8));
        HIP_TRY(hipMemcpy(m->d_trie8, t8.rec.data(), ns * sizeof(tgx::Trie8Rec), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(m->d_values, t8.values.data(), t8.values.size() * 8, hipMemcpyHostToDevice));
        m->n_values = (uint32_t)t8.values.size() - 1u;
        m->value_coverage = std::move(t8.coverage);
        m->root_base8 = t8.root_base;
        m->have_trie8 = true;
    }
    HIP_TRY(hipStreamSynchronize(m->stream));
    hp.mark("uploads");
    m->encode_tables_ready = true;
    return TGX_OK;
}

static tgx_status finish_model_create(tgx_model* m, const uint8_t* bytes, const uint64_t* offs, const double* scores,
                                      uint32_t vocab_size, int device, uint32_t flags, tgx_model** out);

tgx_status tgx_model_create(const uint8_t* bytes, const uint64_t* offs, const double* scores,
                            uint32_t vocab_size, int device, tgx_model** out) {
    return tgx_model_create_ex(bytes, offs, scores, vocab_size, device, 0, out);
}

tgx_status tgx_model_create_ex(const uint8_t* bytes, const uint64_t* offs, const double* scores,
                               uint32_t vocab_size, int device, uint32_t flags, tgx_model** out) {
    if (!out) return fail(TGX_ERR_INVALID, "tgx_model_create: out is NULL");
    *out = nullptr;
    if (vocab_size && (!offs || !scores)) return fail(TGX_ERR_INVALID, "tgx_model_create: NULL vocab");
    if (vocab_size && !bytes && offs[vocab_size] != offs[0]) return fail(TGX_ERR_INVALID, "tgx_model_create: NULL token bytes");
    int ndev = usable_device_count();
    if (ndev <= 0) return fail(TGX_ERR_DEVICE, "no usable HIP device (gfx950 required)");
    if (device < 0 || device >= ndev)
        return fail(TGX_ERR_INVALID, "device %d out of range (have %d)", device, ndev);

    tgx::HostPhases hp("tgx_model_create");
    tgx_model* m = new tgx_model();
    m->device = device;
    m->vocab_size = vocab_size;
    static const uint64_t zero_offs[1] = {0};
    // TGX_MODEL_FOR_ESTEP: the double-array of the REVERSED tokens (backward sweep of the E-step) is built
    // on a second host thread while this one builds the forward table — `prune` makes a new model for every
    // EM sub-iteration (src/prune.rs:48), and at 500 K tokens each table takes ~0.4 s of host time
    // (round 4: not for vocabularies estep7_kernel serves — tokens of at most 16 bytes, scores within +-300: its E-step
    // walks the forward table only; the reversed table is then built if a pass ever falls back to the chained kernels)
    bool fused_estep = vocab_size != 0;
    for (uint32_t i = 0; i < vocab_size && fused_estep; i++)
        fused_estep = offs[i + 1] - offs[i] <= 16 && scores[i] >= -300.0 && scores[i] <= 300.0;
    std::thread rev_builder;
    if ((flags & TGX_MODEL_FOR_ESTEP) && vocab_size && !fused_estep)
        rev_builder = std::thread([m, bytes, offs, scores, vocab_size]() {
            const uint64_t o0 = offs[0];
            std::vector<uint8_t> rev((size_t)(offs[vocab_size] - o0));
            std::vector<uint64_t> ro(vocab_size + 1);
            for (uint32_t i = 0; i <= vocab_size; i++) ro[i] = offs[i] - o0;
            for (uint32_t i = 0; i < vocab_size; i++)
                for (uint64_t k = ro[i]; k < ro[i + 1]; k++) rev[k] = bytes[o0 + ro[i + 1] - 1 - (k - ro[i])];
            tgx::build_flat_trie(rev.data(), ro.data(), scores, vocab_size, &m->flat_rev);
        });
    // a model for encode: the bytes -> id table does not depend on the trie and is built beside it
    std::thread hash_early;
    if (!(flags & TGX_MODEL_FOR_ESTEP) && vocab_size) {
        bool finite = true;
        uint64_t longest = 0;
        for (uint32_t i = 0; i < vocab_size; i++) {
            finite = finite && (scores[i] - scores[i] == 0.0);
            longest = std::max<uint64_t>(longest, offs[i + 1] - offs[i]);
        }
        if (finite && longest <= 32) {
            hash_early = std::thread([m, bytes, offs, vocab_size]() {
                // (same arguments as ensure_encode_tables: token bytes from offs[0] on, offsets rebased to 0)
                std::vector<uint64_t> ro(vocab_size + 1);
                for (uint32_t i = 0; i <= vocab_size; i++) ro[i] = offs[i] - offs[0];
                tgx::build_tok_hash(bytes + offs[0], ro.data(), vocab_size, &m->tokhash);
            });
        }
    }
    hp.mark("scan of the vocabulary");
    tgx::build_flat_trie(bytes, vocab_size ? offs : zero_offs, scores, vocab_size, &m->flat);
    hp.mark("build_flat_trie");
    if (hash_early.joinable()) {
        hash_early.join();
        m->tokhash_host_built = true;
    }
    if (rev_builder.joinable()) {
        rev_builder.join();
        m->rev_host_built = true;
    }
    hp.mark("join hash / reverse builders");
    const tgx_status fst = finish_model_create(m, bytes, offs, scores, vocab_size, device, flags, out);
    hp.mark("finish (device tables)");
    return fst;
}

// the part of model creation after the host tables exist: checks, device tables, uploads
static tgx_status finish_model_create(tgx_model* m, const uint8_t* bytes, const uint64_t* offs, const double* scores,
                                      uint32_t vocab_size, int device, uint32_t flags, tgx_model** out) {
    if (m->flat.max_token_len > TGX_MAX_TOKEN_LEN) {
        uint32_t l = m->flat.max_token_len;
        delete m;
        return fail(TGX_ERR_UNSUPPORTED, "token of %u bytes exceeds TGX_MAX_TOKEN_LEN (%d)", l,
                    TGX_MAX_TOKEN_LEN);
    }
    if (m->flat.table.size() >= (1u << 26)) {
        delete m;
        return fail(TGX_ERR_UNSUPPORTED, "trie needs more than 2^26 slots");
    }
    m->lm = std::max<uint32_t>(4, (m->flat.max_token_len + 3) & ~3u);
    if (vocab_size) {
        m->vocab_bytes.assign(bytes + offs[0], bytes + offs[vocab_size]);
        m->vocab_offs.resize(vocab_size + 1);
        for (uint32_t i = 0; i <= vocab_size; i++) m->vocab_offs[i] = offs[i] - offs[0];
        m->vocab_scores.assign(scores, scores + vocab_size);
        for (uint32_t i = 0; i < vocab_size; i++)
            if (!(scores[i] - scores[i] == 0.0)) m->scores_finite = false;  // inf or NaN
    } else {
        m->vocab_offs.assign(1, 0);
    }

    auto cleanup = [&](tgx_status st) {
        tgx_model_destroy(m);
        return st;
    };
#define HIP_TRY_M(expr)                                                                     \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess)                                                               \
            return cleanup(fail(TGX_ERR_DEVICE, "HIP error %d (%s) at %s:%d: %s", (int)_e,  \
                                hipGetErrorString(_e), __FILE__, __LINE__, #expr));         \
    } while (0)
    HIP_TRY_M(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY_M(hipGetDeviceProperties(&prop, device));
    m->num_cus = prop.multiProcessorCount;
    HIP_TRY_M(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
    size_t tbytes = m->flat.table.size() * sizeof(tgx::TrieRec);
    HIP_TRY_M(hipMalloc(&m->d_trie, tbytes));
    HIP_TRY_M(hipMalloc((void**)&m->d_tokid, m->flat.tokid.size() * 4));
    HIP_TRY_M(hipMalloc((void**)&m->d_ctrl, 64));
    // the tables only encode needs (bytes -> id table of the trace, 8-byte records and score table of
    // encode5_kernel): now, unless the model is created for E-step passes — then at its first encode
    if (!(flags & TGX_MODEL_FOR_ESTEP)) {
        const tgx_status est = ensure_encode_tables(m);
        if (est != TGX_OK) return cleanup(est);
    }
    HIP_TRY_M(hipHostMalloc((void**)&m->h_ctrl, 64, hipHostMallocDefault));
    HIP_TRY_M(hipMemcpyAsync(m->d_trie, m->flat.table.data(), tbytes, hipMemcpyHostToDevice, m->stream));
    HIP_TRY_M(hipMemcpyAsync(m->d_tokid, m->flat.tokid.data(), m->flat.tokid.size() * 4,
                             hipMemcpyHostToDevice, m->stream));
    for (int i = 0; i < kMaxTimed; i++) {
        m->timed[i].used = false;
        m->timed[i].name = "";
        HIP_TRY_M(hipEventCreate(&m->timed[i].start));
        HIP_TRY_M(hipEventCreate(&m->timed[i].stop));
    }
    {
        int occ = 0;
        HIP_TRY_M(tgx::encode_max_blocks_per_cu(m->lm, &occ));
        m->blocks_per_cu = std::max(1, std::min(occ, 16));
    }
    HIP_TRY_M(hipStreamSynchronize(m->stream));
#undef HIP_TRY_M
    *out = m;
    return TGX_OK;
}

// A model for a SUBSET of another model's vocabulary with new scores, on the other model's double-arrays: the tokens that
// are gone lose their terminal mark, the others get their new id and score, and nothing is rebuilt — a trie with dead
// branches gives the same matches.  `prune` makes three models per iteration (two EM sub-iterations and the pruning step,
// src/prune.rs:36-56), each for a subset of the one before; the double-array builds were two thirds of an iteration at
// 500 000 entries.  keep_ids: ascending ids of the parent's vocabulary; new id = position in keep_ids.
// Not for parents with duplicate tokens (only the last duplicate is in the parent's tables): TGX_ERR_UNSUPPORTED, the
// caller builds the model from scratch.
tgx_status tgx_model_create_derived(const tgx_model* parent, const uint32_t* keep_ids, uint32_t n_keep, const double* scores,
                                    uint32_t flags, tgx_model** out) {
    if (!out) return fail(TGX_ERR_INVALID, "tgx_model_create_derived: out is NULL");
    *out = nullptr;
    if (!parent || (n_keep && (!keep_ids || !scores))) return fail(TGX_ERR_INVALID, "tgx_model_create_derived: NULL argument");
    tgx::HostPhases hp("tgx_model_create_derived");
    const uint32_t PV = parent->vocab_size;
    for (uint32_t i = 0; i < n_keep; i++)
        if (keep_ids[i] >= PV || (i && keep_ids[i] <= keep_ids[i - 1])) return fail(TGX_ERR_INVALID, "tgx_model_create_derived: keep_ids must ascend within the parent's vocabulary");
    // A derived model keeps the parent's slot count.  A parent above the 8-byte records' limit would leave the model that
    // `prune` encodes with (its frequency pass) on the 16-byte kernels however small the vocabulary has become: such a
    // model is built from scratch (the caller does that on TGX_ERR_UNSUPPORTED), its own table is within the limit sooner.
    if (!(flags & TGX_MODEL_FOR_ESTEP) && parent->flat.table.size() > tgx::kTrie8MaxSlots && (uint64_t)n_keep * 2 <= PV)
        return fail(TGX_ERR_UNSUPPORTED, "tgx_model_create_derived: the parent's table has more slots than 8-byte records address; build this model from scratch");
    {   // every non-empty token of the parent must own a terminal slot
        uint64_t non_empty = 0, terminals = 0;
        for (uint32_t i = 0; i < PV; i++) non_empty += parent->vocab_offs[i + 1] > parent->vocab_offs[i];
        for (uint32_t t : parent->flat.tokid) terminals += t != tgx::kNoToken;
        if (terminals != non_empty) return fail(TGX_ERR_UNSUPPORTED, "tgx_model_create_derived: the parent vocabulary has duplicate tokens");
    }
    tgx_model* m = new tgx_model();
    m->device = parent->device;
    m->vocab_size = n_keep;
    std::vector<uint32_t> new_id(PV, tgx::kNoToken);
    for (uint32_t i = 0; i < n_keep; i++) new_id[keep_ids[i]] = i;
    // the new vocabulary's bytes (the tail of creation copies them into the model)
    std::vector<uint8_t> bytes;
    std::vector<uint64_t> offs(n_keep + 1, 0);
    uint32_t longest = 0;
    for (uint32_t i = 0; i < n_keep; i++) {
        const uint64_t b = parent->vocab_offs[keep_ids[i]], e = parent->vocab_offs[keep_ids[i] + 1];
        offs[i + 1] = offs[i] + (e - b);
        longest = std::max<uint32_t>(longest, (uint32_t)(e - b));
    }
    bytes.resize(offs[n_keep]);
    for (uint32_t i = 0; i < n_keep; i++) {
        const uint64_t b = parent->vocab_offs[keep_ids[i]], e = parent->vocab_offs[keep_ids[i] + 1];
        if (e > b) memcpy(bytes.data() + offs[i], parent->vocab_bytes.data() + b, (size_t)(e - b));
    }
    hp.mark("checks + bytes");
    auto derive = [&](const tgx::FlatTrie& src, tgx::FlatTrie* dst) {
        *dst = src;
        dst->max_token_len = longest;
        for (size_t t = 0; t < dst->tokid.size(); t++) {
            const uint32_t old = dst->tokid[t];
            if (old == tgx::kNoToken) continue;
            const uint32_t nid = new_id[old];
            tgx::TrieRec& r = dst->table[t];
            if (nid == tgx::kNoToken) {
                dst->tokid[t] = tgx::kNoToken;
                r.base &= ~tgx::kTerminalBit;
                r.score_bits = 0;
            } else {
                dst->tokid[t] = nid;
                memcpy(&r.score_bits, &scores[nid], 8);
            }
        }
    };
    // (the parent's reversed table is built, if ever, under its lock: ensure_reverse_trie)
    std::unique_lock<std::mutex> parent_lock(const_cast<tgx_model*>(parent)->mu);
    std::thread rev_deriver;
    const bool want_rev = (flags & TGX_MODEL_FOR_ESTEP) && parent->rev_host_built && n_keep;
    if (want_rev) rev_deriver = std::thread([&]() { derive(parent->flat_rev, &m->flat_rev); });
    std::thread hash_early;
    if (!(flags & TGX_MODEL_FOR_ESTEP) && n_keep && longest <= 32) {
        bool finite = true;
        for (uint32_t i = 0; i < n_keep; i++) finite = finite && (scores[i] - scores[i] == 0.0);
        if (finite) hash_early = std::thread([&]() { tgx::build_tok_hash(bytes.data(), offs.data(), n_keep, &m->tokhash); });
    }
    derive(parent->flat, &m->flat);
    if (hash_early.joinable()) {
        hash_early.join();
        m->tokhash_host_built = true;
    }
    if (rev_deriver.joinable()) {
        rev_deriver.join();
        m->rev_host_built = true;
    }
    parent_lock.unlock();
    hp.mark("derive tables");
    const tgx_status fst = finish_model_create(m, bytes.data(), offs.data(), scores, n_keep, parent->device, flags, out);
    hp.mark("finish (device tables)");
    return fst;
}

// tgx_prune_alternatives over the model's own double-array (the prune driver has a model of the same vocabulary
// at hand: no second table is built)
tgx_status tgx_model_prune_alternatives(const tgx_model* m, uint8_t* always_keep, uint32_t* alt_offs, uint32_t** alt_ids) {
    if (!m) return fail(TGX_ERR_INVALID, "tgx_model_prune_alternatives: NULL argument");
    return (tgx_status)tgx_prune_alternatives_flat(&m->flat, m->vocab_bytes.data(), m->vocab_offs.data(), m->vocab_scores.data(),
                                                   m->vocab_size, always_keep, alt_offs, alt_ids);
}

void tgx_model_destroy(tgx_model* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    for (int i = 0; i < kMaxTimed; i++) {
        if (m->timed[i].start) (void)hipEventDestroy(m->timed[i].start);
        if (m->timed[i].stop) (void)hipEventDestroy(m->timed[i].stop);
    }
    if (m->d_trie) (void)hipFree(m->d_trie);
    if (m->d_trie_rev) (void)hipFree(m->d_trie_rev);
    if (m->d_trie_w) (void)hipFree(m->d_trie_w);
    if (m->d_trie_rev_w) (void)hipFree(m->d_trie_rev_w);
    if (m->d_tokhash) (void)hipFree(m->d_tokhash);
    if (m->d_trie8) (void)hipFree(m->d_trie8);
    if (m->d_values) (void)hipFree(m->d_values);
    if (m->d_tokid) (void)hipFree(m->d_tokid);
    if (m->d_ctrl) (void)hipFree(m->d_ctrl);
    if (m->h_ctrl) (void)hipHostFree(m->h_ctrl);
    if (m->d_wvalues) (void)hipFree(m->d_wvalues);
    if (m->d_trie8t) (void)hipFree(m->d_trie8t);
    if (m->d_wtab) (void)hipFree(m->d_wtab);
    if (m->stream2) (void)hipStreamDestroy(m->stream2);
    if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
    if (m->ev_join) (void)hipEventDestroy(m->ev_join);
    if (m->h_started) (void)hipHostFree(m->h_started);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
}

uint32_t tgx_model_vocab_size(const tgx_model* m) { return m ? m->vocab_size : 0; }
uint32_t tgx_model_max_token_len(const tgx_model* m) { return m ? m->flat.max_token_len : 0; }
uint64_t tgx_model_trie_bytes(const tgx_model* m) {
    return m ? (uint64_t)m->flat.table.size() * sizeof(tgx::TrieRec) : 0;
}
int tgx_model_device(const tgx_model* m) { return m ? m->device : -1; }

tgx_status tgx_common_prefix_search(const tgx_model* m, const uint8_t* s, uint64_t n, uint32_t* ids,
                                    uint32_t* lens, uint64_t cap, uint64_t* count) {
    if (!m || !count) return fail(TGX_ERR_INVALID, "tgx_common_prefix_search: NULL argument");
    *count = tgx::flat_common_prefix_search(m->flat, s, n, ids, lens, cap);
    return TGX_OK;
}

// ---- host-only trie introspection ----------------------------------------------


tgx_status tgx_flat_trie_build(const uint8_t* bytes, const uint64_t* offs, const double* scores,
                               uint32_t vocab_size, tgx_flat_trie** out) {
    if (!out) return fail(TGX_ERR_INVALID, "tgx_flat_trie_build: out is NULL");
    static const uint64_t zero_offs[1] = {0};
    tgx_flat_trie* t = new tgx_flat_trie();
    tgx::build_flat_trie(bytes, vocab_size ? offs : zero_offs, scores, vocab_size, &t->flat);
    *out = t;
    return TGX_OK;
}
void tgx_flat_trie_free(tgx_flat_trie* t) { delete t; }
uint64_t tgx_flat_trie_search(const tgx_flat_trie* t, const uint8_t* s, uint64_t n, uint32_t* ids,
                              uint32_t* lens, uint64_t cap) {
    return t ? tgx::flat_common_prefix_search(t->flat, s, n, ids, lens, cap) : 0;
}
uint64_t tgx_flat_trie_search8(const tgx_flat_trie* t, const uint8_t* bytes, const uint64_t* offs, const double* scores,
                               uint32_t max_hot, const uint8_t* s, uint64_t n, uint32_t* ids, uint32_t* lens,
                               uint64_t cap, uint32_t* n_hot, uint64_t* n_cold, double* hot_coverage) {
    if (!t) return ~0ULL;
    static const uint64_t zero_offs[1] = {0};
    // (test infrastructure: the records are built on first use and kept on the handle)
    tgx_flat_trie* tm = const_cast<tgx_flat_trie*>(t);
    if (!tm->t8) {
        tm->t8.reset(new tgx::Trie8());
        tgx::build_trie8(t->flat, offs ? offs : zero_offs, scores, tm->t8.get());
    }
    const tgx::Trie8& t8 = *tm->t8;
    (void)bytes;
    if (!t8.ok) return ~0ULL;  // more than 65 535 distinct score values (or more than 2^21 slots): no 8-byte records
    const uint32_t n_values = (uint32_t)t8.values.size() - 1u, k = std::min(max_hot, n_values);
    if (n_hot) *n_hot = k;
    if (n_cold) {
        uint64_t c = 0;
        for (const tgx::Trie8Rec& q : t8.rec) c += (q.sref & tgx::kTrie8RankMask) > k ? 1 : 0;
        *n_cold = c;
    }
    if (hot_coverage) *hot_coverage = t8.coverage[k];
    return tgx::trie8_common_prefix_search(t8, t->flat, s, n, ids, lens, cap);
}
void tgx_flat_trie_stats(const tgx_flat_trie* t, uint64_t* n_slots, uint64_t* n_nodes,
                         uint32_t* max_token_len) {
    if (!t) return;
    if (n_slots) *n_slots = t->flat.table.size();
    if (n_nodes) *n_nodes = t->flat.n_nodes;
    if (max_token_len) *max_token_len = t->flat.max_token_len;
}
tgx_status tgx_tok_hash_selftest(const uint8_t* bytes, const uint64_t* offs, uint32_t vocab_size, uint32_t* seed,
                                 uint64_t* mismatches) {
    if (!offs || !mismatches) return TGX_ERR_INVALID;
    tgx::TokHashTable t;
    tgx::build_tok_hash(bytes, offs, vocab_size, &t);
    if (!t.ok) return TGX_ERR_UNSUPPORTED;
    if (seed) *seed = t.seed;
    // the id a lookup must give: the LAST vocabulary entry with these bytes (src/trie.rs:19)
    uint64_t bad = 0;
    for (uint32_t id = 0; id < vocab_size; id++) {
        const uint32_t len = (uint32_t)(offs[id + 1] - offs[id]);
        if (len == 0) continue;
        uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        memcpy(w, bytes + offs[id], len);
        const uint64_t h = tgx::tok_hash64_long(w, len, t.seed);
        uint32_t i = (uint32_t)h & t.mask;
        bool found = false;
        for (uint32_t probe = 0; probe <= t.mask; probe++) {  // the device loop of trace_kernel
            const tgx::TokHashEntry& e = t.slots[i];
            if (!e.used) break;
            if (e.hash == h) {
                const uint32_t o = e.id;
                found = offs[o + 1] - offs[o] == len && memcmp(bytes + offs[o], bytes + offs[id], len) == 0 && o >= id;
                break;
            }
            i = (i + 1) & t.mask;
        }
        if (!found) bad++;
    }
    *mismatches = bad;
    return TGX_OK;
}
void tgx_flat_trie_copy(const tgx_flat_trie* t, uint32_t* check, uint32_t* base_flags, uint32_t* tokid) {
    if (!t) return;
    for (size_t i = 0; i < t->flat.table.size(); i++) {
        if (check) check[i] = t->flat.table[i].check;
        if (base_flags) base_flags[i] = t->flat.table[i].base;
        if (tokid) tokid[i] = t->flat.tokid[i];
    }
}
double tgx_dropout_u01_host(uint64_t seed, uint64_t sample, uint64_t pos, uint32_t len) {
    return tgx_dropout_u01(seed, sample, pos, len);
}

// ---- corpus ------------------------------------------------------------------

// Host <-> device copies of corpora and results go through two streams of their own per device (one per
// direction): synchronous hipMemcpy calls all share the null stream, so an upload from one thread and a download
// from another took turns on it instead of using the link in both directions (profiles/r02: e2e timeline).
static hipStream_t copy_stream(int device, int dir) {
    static std::mutex mu;
    static hipStream_t streams[64][2] = {};
    if (device < 0 || device >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    if (!streams[device][dir]) {
        hipStream_t st = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;  // the null stream: correct, only slower
        }
        streams[device][dir] = st;
    }
    return streams[device][dir];
}
static hipError_t copy_sync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, int device) {
    hipStream_t st = copy_stream(device, kind == hipMemcpyHostToDevice ? 0 : 1);
    hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, st);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(st);
}

tgx_status tgx_corpus_upload(int device, const uint8_t* text, const uint64_t* offs,
                             uint64_t n_samples, tgx_corpus** out) {
    if (!out) return fail(TGX_ERR_INVALID, "tgx_corpus_upload: out is NULL");
    *out = nullptr;
    if (n_samples && !offs) return fail(TGX_ERR_INVALID, "tgx_corpus_upload: offs is NULL");
    int ndev = usable_device_count();
    if (ndev <= 0) return fail(TGX_ERR_DEVICE, "no usable HIP device (gfx950 required)");
    if (device < 0 || device >= ndev)
        return fail(TGX_ERR_INVALID, "device %d out of range (have %d)", device, ndev);
    if (n_samples >= 0xFFFFFFFFull) return fail(TGX_ERR_UNSUPPORTED, "more than 2^32-1 samples");
    const uint64_t base = n_samples ? offs[0] : 0;
    for (uint64_t i = 0; i < n_samples; i++) {
        if (offs[i + 1] < offs[i]) return fail(TGX_ERR_INVALID, "offsets not monotone at %llu",
                                               (unsigned long long)i);
        if (offs[i + 1] - offs[i] >= 0xFFFFFF00ull)
            return fail(TGX_ERR_UNSUPPORTED, "sample %llu is 4 GiB or longer", (unsigned long long)i);
    }
    tgx_corpus* c = new tgx_corpus();
    c->device = device;
    c->n_samples = n_samples;
    c->n_bytes = n_samples ? offs[n_samples] - base : 0;
    c->h_offs.resize(n_samples + 1);
    for (uint64_t i = 0; i <= n_samples; i++) c->h_offs[i] = n_samples ? offs[i] - base : 0;
    if (n_samples && c->n_bytes && !text) {
        delete c;
        return fail(TGX_ERR_INVALID, "tgx_corpus_upload: text is NULL");
    }
    // longest first: the tail of a pass is then made of short samples
    std::vector<uint32_t> order(n_samples);
    std::iota(order.begin(), order.end(), 0u);
    const uint64_t* ho = c->h_offs.data();
    std::stable_sort(order.begin(), order.end(), [ho](uint32_t a, uint32_t b) {
        return ho[a + 1] - ho[a] > ho[b + 1] - ho[b];
    });

    c->max_len = n_samples ? ho[order[0] + 1] - ho[order[0]] : 0;
    c->h_sorted_len.resize(n_samples);
    c->h_sorted_cum.resize(n_samples);
    for (uint64_t i = 0, run = 0; i < n_samples; i++) {
        c->h_sorted_len[i] = (uint32_t)(ho[order[i] + 1] - ho[order[i]]);
        run += c->h_sorted_len[i];
        c->h_sorted_cum[i] = run;
    }
    auto cleanup = [&](tgx_status st) {
        tgx_corpus_free(c);
        return st;
    };
#define HIP_TRY_C(expr)                                                                     \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess)                                                               \
            return cleanup(fail(TGX_ERR_DEVICE, "HIP error %d (%s) at %s:%d: %s", (int)_e,  \
                                hipGetErrorString(_e), __FILE__, __LINE__, #expr));         \
    } while (0)
    HIP_TRY_C(hipSetDevice(device));
    HIP_TRY_C(pool_alloc(device, (size_t)c->n_bytes + 512, (void**)&c->d_text_alloc));
    c->d_text = c->d_text_alloc + 256;
    HIP_TRY_C(hipMemset(c->d_text_alloc, 0, 256));
    HIP_TRY_C(pool_alloc(device, (size_t)(n_samples + 1) * 8, (void**)&c->d_offs));
    HIP_TRY_C(pool_alloc(device, (size_t)n_samples * 4 + 4, (void**)&c->d_order));
    if (c->n_bytes) HIP_TRY_C(copy_sync(c->d_text, text + base, c->n_bytes, hipMemcpyHostToDevice, device));
    HIP_TRY_C(hipMemset(c->d_text + c->n_bytes, 0, 256));
    HIP_TRY_C(hipMemcpy(c->d_offs, c->h_offs.data(), (n_samples + 1) * 8, hipMemcpyHostToDevice));
    if (n_samples) HIP_TRY_C(hipMemcpy(c->d_order, order.data(), n_samples * 4, hipMemcpyHostToDevice));
#undef HIP_TRY_C
    *out = c;
    return TGX_OK;
}

void tgx_corpus_free(tgx_corpus* c) {
    if (!c) return;
    pool_free(c->device, c->d_text_alloc, (size_t)c->n_bytes + 512);
    pool_free(c->device, c->d_offs, (size_t)(c->n_samples + 1) * 8);
    pool_free(c->device, c->d_order, (size_t)c->n_samples * 4 + 4);
    pool_free(c->device, c->d_bp, std::max((size_t)c->n_bytes * 4 + 256, (size_t)c->n_bytes + 128 * (size_t)c->n_samples + 512));
    if (c->d_tmp) pool_free(c->device, c->d_tmp, (size_t)c->n_bytes * 4 + 256);
    pool_free(c->device, c->d_counts, (size_t)c->n_samples * 4 + 256);
    pool_free(c->device, c->d_status, (size_t)c->n_samples * 4 + 256);
    pool_free(c->device, c->d_scan_tmp, c->scan_tmp_bytes);
    pool_free(c->device, c->d_endmask, (size_t)(c->mask_words + 1) * 8 + 256);
    pool_free(c->device, c->d_prefix, (size_t)(c->mask_words + 1) * 8 + 256);
    pool_free(c->device, c->d_mscan_tmp, c->mscan_tmp_bytes);
    pool_free(c->device, c->d_mword, (size_t)(c->n_samples + 1) * 8 + 256);
    pool_free(c->device, c->es.d_soffs, c->es.obytes);
    pool_free(c->device, c->es.d_sbase, c->es.obytes);
    pool_free(c->device, c->es.d_order, c->es.ordbytes);
    pool_free(c->device, c->es.d_ssample, c->es.ordbytes);
    pool_free(c->device, c->es.d_win_snip, c->es.winbytes);
    pool_free(c->device, c->es.d_win_k, c->es.winbytes);
    delete c;
}

uint64_t tgx_corpus_num_samples(const tgx_corpus* c) { return c ? c->n_samples : 0; }
uint64_t tgx_corpus_num_bytes(const tgx_corpus* c) { return c ? c->n_bytes : 0; }

// ---- encode --------------------------------------------------------------------

// caller holds m->mu
static tgx_status encode_corpus_locked(tgx_model* m, tgx_corpus* c, double dropout, uint64_t seed,
                                       tgx_result** out) {
    HIP_TRY(hipSetDevice(m->device));
    {
        const tgx_status est = ensure_encode_tables(m);
        if (est != TGX_OK) return est;
    }
    m->n_timed = 0;
    const uint64_t S = c->n_samples;

    tgx_result* r = new tgx_result();
    r->device = m->device;
    r->n_samples = S;
    auto cleanup = [&](tgx_status st) {
        tgx_result_free(r);
        return st;
    };
    if (pool_alloc(m->device, (size_t)(S + 1) * 8, (void**)&r->d_offs) != hipSuccess)
        return cleanup(fail(TGX_ERR_DEVICE, "out of device memory (result offsets)"));

    tgx_status st = run_encode_kernel(m, c, dropout, seed);
    if (st != TGX_OK) return cleanup(st);
    if (m->mask_path) {
        // TGX_TRACE=mask: ids in their final place (trace2.hip): the mask's popcount prefix gives every token its index, the batch its
        // token count and every sample its offset; emit_kernel turns set bits into ids, fully parallel
        time_begin(m, "mask_scan");
        if (tgx::launch_mask_scan(c->d_endmask, c->d_prefix, c->mask_words, c->d_mscan_tmp, c->mscan_tmp_bytes, m->stream) != hipSuccess ||
            tgx::launch_sample_offs(c->d_mword, S, c->d_prefix, r->d_offs, m->stream) != hipSuccess)
            return cleanup(fail(TGX_ERR_DEVICE, "mask scan launch failed"));
        time_end(m);
        if (hipMemcpyAsync(&m->h_ctrl[0], m->d_ctrl + 1, 8, hipMemcpyDeviceToHost, m->stream) != hipSuccess ||
            hipMemcpyAsync(&m->h_ctrl[1], c->d_prefix + c->mask_words, 8, hipMemcpyDeviceToHost, m->stream) != hipSuccess ||
            hipStreamSynchronize(m->stream) != hipSuccess)
            return cleanup(fail(TGX_ERR_DEVICE, "encode pass failed: %s", hipGetErrorString(hipGetLastError())));
        st = check_no_path(m, c);
        if (st != TGX_OK) return cleanup(st);
        r->n_tokens = m->h_ctrl[1];
        if (pool_alloc(m->device, (size_t)r->n_tokens * 4 + 256, (void**)&r->d_ids) != hipSuccess)
            return cleanup(fail(TGX_ERR_DEVICE, "out of device memory (result ids)"));
        tgx::EncodeParams pe{};
        pe.text = c->d_text;
        pe.tokhash = m->d_tokhash;
        pe.tokhash_mask = m->tokhash.mask;
        pe.tokhash_seed = m->tokhash.seed;
        pe.err_sample = m->d_ctrl + 1;
        pe.offs = c->d_offs;
        pe.n_samples = S;
        pe.endmask = c->d_endmask;
        pe.mword = c->d_mword;
        pe.mask_words = c->mask_words;
        pe.prefix = c->d_prefix;
        pe.ids_out = r->d_ids;
        time_begin(m, "emit_kernel");
        if (tgx::launch_emit(pe, m->lm <= 16 ? 16u : 32u, (uint32_t)m->num_cus, m->stream) != hipSuccess)
            return cleanup(fail(TGX_ERR_DEVICE, "emit launch failed"));
        time_end(m);
        if (hipMemcpyAsync(&m->h_ctrl[0], m->d_ctrl + 1, 8, hipMemcpyDeviceToHost, m->stream) != hipSuccess ||
            hipStreamSynchronize(m->stream) != hipSuccess)
            return cleanup(fail(TGX_ERR_DEVICE, "emit failed: %s", hipGetErrorString(hipGetLastError())));
        st = check_no_path(m, c);  // (a lookup that missed: "corrupt back-pointer", never a fault)
        if (st != TGX_OK) return cleanup(st);
    } else {
    if (!c->d_scan_tmp) {  // scratch of the device-wide scan (large sample counts only), kept on the corpus
        if (tgx::scan_temp_bytes(S, &c->scan_tmp_bytes) != hipSuccess)
            return cleanup(fail(TGX_ERR_DEVICE, "scan temp-size query failed"));
        if (c->scan_tmp_bytes && pool_alloc(m->device, c->scan_tmp_bytes, &c->d_scan_tmp) != hipSuccess)
            return cleanup(fail(TGX_ERR_DEVICE, "out of device memory (scan)"));
    }
    time_begin(m, "scan_counts_kernel");
    if (tgx::launch_scan(c->d_counts, r->d_offs, S, c->d_scan_tmp, c->scan_tmp_bytes, m->stream) != hipSuccess)
        return cleanup(fail(TGX_ERR_DEVICE, "scan launch failed"));
    time_end(m);
    if (hipMemcpyAsync(&m->h_ctrl[0], m->d_ctrl + 1, 8, hipMemcpyDeviceToHost, m->stream) != hipSuccess ||
        hipMemcpyAsync(&m->h_ctrl[1], r->d_offs + S, 8, hipMemcpyDeviceToHost, m->stream) != hipSuccess ||
        hipStreamSynchronize(m->stream) != hipSuccess)
        return cleanup(fail(TGX_ERR_DEVICE, "encode pass failed: %s", hipGetErrorString(hipGetLastError())));
    st = check_no_path(m, c);
    if (st != TGX_OK) return cleanup(st);

    r->n_tokens = m->h_ctrl[1];
    if (pool_alloc(m->device, (size_t)r->n_tokens * 4 + 256, (void**)&r->d_ids) != hipSuccess)
        return cleanup(fail(TGX_ERR_DEVICE, "out of device memory (result ids)"));
    if (hipMemsetAsync(m->d_ctrl, 0x00, 8, m->stream) != hipSuccess)
        return cleanup(fail(TGX_ERR_DEVICE, "memset failed"));
    tgx::CompactParams cp{};
    cp.offs = c->d_offs;
    cp.order = c->d_order;
    cp.n_samples = S;
    cp.tmp = c->d_tmp;
    cp.out_offs = r->d_offs;
    cp.ids = r->d_ids;
    time_begin(m, "compact_kernel");
    // short samples (fewer than 128 ids on average): a 16-lane row per sample instead of a wave
    const bool rows = S && r->n_tokens / S < 128;
    const uint64_t units = rows ? 16 : 4;  // samples per block and round
    uint32_t blocks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((S + units - 1) / units, (uint64_t)m->num_cus * 8));
    if (tgx::launch_compact(cp, blocks, rows, m->stream) != hipSuccess)
        return cleanup(fail(TGX_ERR_DEVICE, "compact launch failed"));
    time_end(m);
    if (hipStreamSynchronize(m->stream) != hipSuccess)
        return cleanup(fail(TGX_ERR_DEVICE, "compact failed: %s", hipGetErrorString(hipGetLastError())));
    }
    // N + 4T + 16(S+1)
    m->last_alg_bytes = c->n_bytes + 4 * r->n_tokens + 16 * (S + 1);
    *out = r;
    return TGX_OK;
}

tgx_status tgx_encode_corpus(tgx_model* m, tgx_corpus* c, double dropout, uint64_t seed,
                             tgx_result** out) {
    if (!m || !c || !out) return fail(TGX_ERR_INVALID, "tgx_encode_corpus: NULL argument");
    *out = nullptr;
    if (m->device != c->device) return fail(TGX_ERR_INVALID, "model and corpus on different devices");
    std::lock_guard<std::mutex> lk(m->mu);
    std::lock_guard<std::mutex> lkc(c->mu);
    return encode_corpus_locked(m, c, dropout, seed, out);
}

tgx_status tgx_encode_batch(tgx_model* m, const uint8_t* text, const uint64_t* offs,
                            uint64_t n_samples, double dropout, uint64_t seed, tgx_result** out) {
    if (!m || !out) return fail(TGX_ERR_INVALID, "tgx_encode_batch: NULL argument");
    *out = nullptr;
    tgx_corpus* c = nullptr;
    tgx_status st = tgx_corpus_upload(m->device, text, offs, n_samples, &c);
    if (st != TGX_OK) return st;
    st = tgx_encode_corpus(m, c, dropout, seed, out);
    tgx_corpus_free(c);
    return st;
}

// Host buffers in, host buffers out (what a Rust caller of Tokenizer::encode_batch has: bindings/python/src/lib.rs:51-59
// hands over borrowed strings and takes owned vectors back).  A large batch is cut at sample boundaries into chunks
// of 256 MiB that go through three stages — upload, kernels, download — each on a thread of its own: the kernels of
// one chunk run while the text of the next goes to the device and the ids of the previous come back, so
// the PCIe link works in both directions beside the kernels instead of before and after them.  ids_out must hold
// ids_cap entries (at most one token per byte: ids_cap = bytes always suffices); offs_out[n_samples + 1].
// Dropout passes keep the one-chunk path (the keep rule hashes the sample's index in the batch).
tgx_status tgx_encode_batch_host(tgx_model* m, const uint8_t* text, const uint64_t* offs, uint64_t n_samples, double dropout,
                                 uint64_t seed, uint32_t* ids_out, uint64_t ids_cap, uint64_t* offs_out, uint64_t* n_tokens) {
    if (!m || !offs_out || !n_tokens || (n_samples && !offs)) return fail(TGX_ERR_INVALID, "tgx_encode_batch_host: NULL argument");
    *n_tokens = 0;
    offs_out[0] = 0;
    if (n_samples == 0) return TGX_OK;
    for (uint64_t i = 0; i < n_samples; i++)
        if (offs[i + 1] < offs[i]) return fail(TGX_ERR_INVALID, "offsets not monotone at %llu", (unsigned long long)i);
    const uint64_t N = offs[n_samples] - offs[0];
    // chunk boundaries
    std::vector<uint64_t> cut(1, 0);
    {
        uint64_t chunk = 256ull << 20;  // a chunk's pass is bounded below by the serial chain of its longest sample: few, large chunks
        if (dropout > 0.0) chunk = ~0ull;
        if (const char* e = knob("TGX_E2E_CHUNK_MB")) {
            const long v = atol(e);
            if (v > 0 && dropout <= 0.0) chunk = (uint64_t)v << 20;
        }
        uint64_t start = offs[0];
        for (uint64_t i = 0; i < n_samples; i++)
            if (offs[i + 1] - start >= chunk && i + 1 < n_samples) {
                cut.push_back(i + 1);
                start = offs[i + 1];
            }
        cut.push_back(n_samples);
    }
    const size_t C = cut.size() - 1;
    // Three stages, each chunk through them in order: this thread downloads, one thread uploads, one runs the
    // kernels.  (Workers that each took a chunk through all three stages started their uploads together, shared
    // the link, and the first kernel waited for all of them.)
    struct ChunkState {
        tgx_corpus* corpus = nullptr;
        tgx_result* result = nullptr;
        uint64_t tokens = 0;
        int stage = 0;  // 1 uploaded, 2 encoded, 3 downloaded
    };
    std::vector<ChunkState> cs(C);
    std::mutex mu;
    std::condition_variable cv;
    tgx_status first_err = TGX_OK;
    std::string err_msg;
    uint64_t err_sample = 0, err_pos = 0, err_len = 0;
    bool abort_all = false;
    size_t downloaded = 0;
    uint64_t err_base = ~0ull;  // first sample of the chunk the recorded error belongs to
    auto set_error = [&](tgx_status st, uint64_t sample_base) {  // under mu; g_err_* are this thread's
        // the error of the LOWEST chunk is the batch's (an upload failure of a later chunk must not hide the NoPath of
        // an earlier one that is still being encoded: the encoder records its own when it gets there)
        if (first_err == TGX_OK || sample_base < err_base) {
            first_err = st;
            err_base = sample_base;
            err_msg = g_err_msg;
            err_sample = g_err_sample + sample_base;
            err_pos = g_err_pos;
            err_len = g_err_len;
        }
        abort_all = true;
        cv.notify_all();
    };
    const auto t_start = std::chrono::steady_clock::now();
    auto stamp = [&](const char* what, size_t k, std::chrono::steady_clock::time_point t0) {  // TGX_DEBUG=1: stage timeline
        if (!debug_on()) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[tgx] e2e chunk %zu %-8s %7.2f -> %7.2f ms\n", k, what,
                std::chrono::duration<double, std::milli>(t0 - t_start).count(),
                std::chrono::duration<double, std::milli>(now - t_start).count());
    };
    std::thread uploader([&]() {
        for (size_t k = 0; k < C; k++) {
            {
                std::unique_lock<std::mutex> lk(mu);  // at most three chunks between upload and download
                cv.wait(lk, [&]() { return abort_all || k < downloaded + 3; });
                if (abort_all) return;
            }
            tgx_corpus* c = nullptr;
            const auto t0 = std::chrono::steady_clock::now();
            const tgx_status st = tgx_corpus_upload(m->device, text, offs + cut[k], cut[k + 1] - cut[k], &c);
            stamp("upload", k, t0);
            std::lock_guard<std::mutex> lk(mu);
            if (st != TGX_OK) {
                set_error(st, cut[k]);
                return;
            }
            cs[k].corpus = c;
            cs[k].stage = 1;
            cv.notify_all();
        }
    });
    std::thread encoder([&]() {
        for (size_t k = 0; k < C; k++) {
            tgx_corpus* c = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&]() { return abort_all || cs[k].stage >= 1; });
                if (cs[k].stage < 1) return;  // aborted before this chunk was uploaded
                c = cs[k].corpus;
                if (abort_all && cut[k] >= err_base) continue;  // (the main thread frees what is left; a chunk BELOW a
                                                                // failed upload is still encoded: its own error would come first)
            }
            tgx_result* r = nullptr;
            const auto t0 = std::chrono::steady_clock::now();
            const tgx_status st = tgx_encode_corpus(m, c, dropout, seed, &r);
            tgx_corpus_free(c);  // text and scratch go back to the pool for the next chunk's upload
            stamp("kernels", k, t0);
            std::lock_guard<std::mutex> lk(mu);
            cs[k].corpus = nullptr;
            if (st != TGX_OK) {
                set_error(st, cut[k]);  // chunks are encoded in order: the lowest failing sample of the batch
                return;
            }
            cs[k].result = r;
            cs[k].tokens = r->n_tokens;
            cs[k].stage = 2;
            cv.notify_all();
        }
    });
    uint64_t base = 0;
    for (size_t k = 0; k < C; k++) {
        tgx_result* r = nullptr;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&]() { return abort_all || cs[k].stage >= 2; });
            if (cs[k].stage < 2) break;
            r = cs[k].result;
        }
        const uint64_t lo = cut[k], hi = cut[k + 1];
        const auto t0 = std::chrono::steady_clock::now();
        tgx_status st2 = TGX_OK;
        if (base + r->n_tokens > ids_cap) {
            st2 = fail(TGX_ERR_INVALID, "tgx_encode_batch_host: ids_out holds %llu ids, the batch has more", (unsigned long long)ids_cap);
        } else {
            if (r->n_tokens) st2 = tgx_result_copy_ids(r, ids_out + base, r->n_tokens);
            if (st2 == TGX_OK) {
                std::vector<uint64_t> lo_offs(hi - lo + 1);
                st2 = tgx_result_copy_offsets(r, lo_offs.data(), hi - lo + 1);
                for (uint64_t i = 0; i <= hi - lo && st2 == TGX_OK; i++) offs_out[lo + i] = base + lo_offs[i];
            }
        }
        base += r->n_tokens;
        stamp("download", k, t0);
        // the chunk's ids are in the caller's memory: its device buffers go back to the pool now (a batch larger
        // than the free HBM streams through; kept until the end they grew by a byte per input byte)
        if (st2 == TGX_OK) tgx_result_free(r);
        std::lock_guard<std::mutex> lk(mu);
        if (st2 != TGX_OK) {
            g_err_sample = 0;
            set_error(st2, 0);
            break;
        }
        cs[k].result = nullptr;
        cs[k].stage = 3;
        downloaded = k + 1;
        cv.notify_all();
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        if (downloaded < C && first_err == TGX_OK) abort_all = true;  // (cannot happen: a stage that stops sets the error)
        cv.notify_all();
    }
    uploader.join();
    encoder.join();
    for (size_t k = 0; k < C; k++) {
        if (cs[k].result) tgx_result_free(cs[k].result);
        if (cs[k].corpus) tgx_corpus_free(cs[k].corpus);
    }
    if (first_err != TGX_OK) {
        g_err_msg = err_msg;
        g_err_sample = err_sample;
        g_err_pos = err_pos;
        g_err_len = err_len;
        return first_err;
    }
    uint64_t total = 0;
    for (size_t k = 0; k < C; k++) total += cs[k].tokens;
    *n_tokens = total;
    m->last_alg_bytes = N + 4 * total + 16 * (n_samples + 1);
    return TGX_OK;
}

// One host process, several devices: the batch is cut at sample boundaries into one byte-balanced shard per model
// handle (one handle per GPU; the reference's batch is ONE call from one process, src/tokenizer.rs:102-111), every shard
// goes through tgx_encode_batch_host on a host thread of its own, and the shards' ids are packed in sample order.  No
// collective: the path shards by samples.  ids_out must hold one id per input byte (ids_cap >= bytes: every shard writes
// at its text offset first).  With dropout the keep decisions hash the sample's index in its SHARD.
tgx_status tgx_encode_batch_multi(tgx_model* const* models, uint32_t n_models, const uint8_t* text, const uint64_t* offs,
                                  uint64_t n_samples, double dropout, uint64_t seed, uint32_t* ids_out, uint64_t ids_cap,
                                  uint64_t* offs_out, uint64_t* n_tokens) {
    if (!models || n_models == 0 || !offs_out || !n_tokens || (n_samples && !offs)) return fail(TGX_ERR_INVALID, "tgx_encode_batch_multi: NULL argument");
    for (uint32_t k = 0; k < n_models; k++)
        if (!models[k]) return fail(TGX_ERR_INVALID, "tgx_encode_batch_multi: NULL model handle");
    *n_tokens = 0;
    offs_out[0] = 0;
    if (n_samples == 0) return TGX_OK;
    const uint64_t N = offs[n_samples] - offs[0];
    if (ids_cap < N) return fail(TGX_ERR_INVALID, "tgx_encode_batch_multi: ids_out must hold one id per input byte (%llu), it holds %llu",
                                 (unsigned long long)N, (unsigned long long)ids_cap);
    // shard bounds: the first sample whose start is at or beyond k / n_models of the bytes
    std::vector<uint64_t> cut(n_models + 1, n_samples);
    cut[0] = 0;
    for (uint32_t k = 1; k < n_models; k++) {
        const uint64_t want = offs[0] + N / n_models * k;
        cut[k] = (uint64_t)(std::lower_bound(offs, offs + n_samples, want) - offs);
        cut[k] = std::max(cut[k], cut[k - 1]);
    }
    struct Shard {
        tgx_status st = TGX_OK;
        uint64_t tokens = 0;
        std::string msg;
        uint64_t es = 0, ep = 0, el = 0;
        std::vector<uint64_t> lo;
    };
    std::vector<Shard> sh(n_models);
    std::vector<std::thread> th;
    for (uint32_t k = 0; k < n_models; k++) {
        th.emplace_back([&, k]() {
            Shard& S = sh[k];
            const uint64_t a = cut[k], b = cut[k + 1];
            if (b == a) return;
            S.lo.assign(b - a + 1, 0);
            const uint64_t region = offs[a] - offs[0];  // (at most one token per byte: the shard's ids fit behind its text offset)
            S.st = tgx_encode_batch_host(models[k], text, offs + a, b - a, dropout, seed, ids_out + region, ids_cap - region, S.lo.data(), &S.tokens);
            if (S.st != TGX_OK) {  // the error state is this thread's
                S.msg = g_err_msg;
                S.es = g_err_sample + a;
                S.ep = g_err_pos;
                S.el = g_err_len;
            }
        });
    }
    for (std::thread& t : th) t.join();
    for (uint32_t k = 0; k < n_models; k++)
        if (sh[k].st != TGX_OK) {  // the lowest failing shard: the lowest failing sample of the batch
            g_err_msg = sh[k].msg;
            g_err_sample = sh[k].es;
            g_err_pos = sh[k].ep;
            g_err_len = sh[k].el;
            return sh[k].st;
        }
    uint64_t base = 0;
    for (uint32_t k = 0; k < n_models; k++) {
        const uint64_t a = cut[k], b = cut[k + 1];
        if (b == a) continue;
        const uint64_t region = offs[a] - offs[0];
        if (region != base && sh[k].tokens) memmove(ids_out + base, ids_out + region, sh[k].tokens * sizeof(uint32_t));
        for (uint64_t i = 0; i <= b - a; i++) offs_out[a + i] = base + sh[k].lo[i];
        base += sh[k].tokens;
    }
    *n_tokens = base;
    return TGX_OK;
}

uint64_t tgx_result_num_samples(const tgx_result* r) { return r ? r->n_samples : 0; }
uint64_t tgx_result_num_tokens(const tgx_result* r) { return r ? r->n_tokens : 0; }

const uint32_t* tgx_result_ids(tgx_result* r) {
    if (!r) return nullptr;
    if (!r->have_ids) {
        r->h_ids.reset(new (std::nothrow) uint32_t[r->n_tokens ? r->n_tokens : 1]);
        if (!r->h_ids) {
            fail(TGX_ERR_DEVICE, "out of host memory (ids)");
            return nullptr;
        }
        if (r->n_tokens) {
            (void)hipSetDevice(r->device);
            if (hipMemcpy(r->h_ids.get(), r->d_ids, r->n_tokens * 4, hipMemcpyDeviceToHost) != hipSuccess) {
                fail(TGX_ERR_DEVICE, "D2H copy of ids failed");
                return nullptr;
            }
        }
        r->have_ids = true;
    }
    return r->h_ids.get();
}

const uint64_t* tgx_result_offsets(tgx_result* r) {
    if (!r) return nullptr;
    if (!r->have_offs) {
        r->h_offs.resize(r->n_samples + 1);
        (void)hipSetDevice(r->device);
        if (hipMemcpy(r->h_offs.data(), r->d_offs, (r->n_samples + 1) * 8, hipMemcpyDeviceToHost) != hipSuccess) {
            fail(TGX_ERR_DEVICE, "D2H copy of offsets failed");
            return nullptr;
        }
        r->have_offs = true;
    }
    return r->h_offs.data();
}

tgx_status tgx_result_copy_ids(const tgx_result* r, uint32_t* dst, uint64_t cap) {
    if (!r || (!dst && r->n_tokens)) return fail(TGX_ERR_INVALID, "tgx_result_copy_ids: NULL argument");
    if (cap < r->n_tokens) return fail(TGX_ERR_INVALID, "tgx_result_copy_ids: %llu ids, room for %llu", (unsigned long long)r->n_tokens, (unsigned long long)cap);
    if (!r->n_tokens) return TGX_OK;
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(copy_sync(dst, r->d_ids, r->n_tokens * 4, hipMemcpyDeviceToHost, r->device));
    return TGX_OK;
}

tgx_status tgx_result_copy_offsets(const tgx_result* r, uint64_t* dst, uint64_t cap) {
    if (!r || !dst) return fail(TGX_ERR_INVALID, "tgx_result_copy_offsets: NULL argument");
    if (cap < r->n_samples + 1) return fail(TGX_ERR_INVALID, "tgx_result_copy_offsets: %llu offsets, room for %llu", (unsigned long long)(r->n_samples + 1), (unsigned long long)cap);
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(copy_sync(dst, r->d_offs, (r->n_samples + 1) * 8, hipMemcpyDeviceToHost, r->device));
    return TGX_OK;
}

const void* tgx_result_ids_device(const tgx_result* r) { return r ? r->d_ids : nullptr; }
const void* tgx_result_offsets_device(const tgx_result* r) { return r ? r->d_offs : nullptr; }

void tgx_result_free(tgx_result* r) {
    if (!r) return;
    pool_free(r->device, r->d_ids, (size_t)r->n_tokens * 4 + 256);
    pool_free(r->device, r->d_offs, (size_t)(r->n_samples + 1) * 8);
    delete r;
}

// ---- frequency pass ------------------------------------------------------------

tgx_status tgx_count_tokens(tgx_model* m, tgx_corpus* c, uint64_t* freq) {
    if (!m || !c || !freq) return fail(TGX_ERR_INVALID, "tgx_count_tokens: NULL argument");
    if (m->device != c->device) return fail(TGX_ERR_INVALID, "model and corpus on different devices");
    std::lock_guard<std::mutex> lk(m->mu);
    std::lock_guard<std::mutex> lkc(c->mu);
    // model.encode(sample, 0.0) for every sample (src/prune.rs:218), then a histogram of the ids: per-block LDS
    // histograms when the vocabulary's counters fit a block's LDS, radix sort + run-length encode otherwise
    tgx_result* r = nullptr;
    tgx_status st = encode_corpus_locked(m, c, 0.0, 0, &r);
    if (st != TGX_OK) return st;
    const uint64_t T = r->n_tokens;
    if (T >= 0xFFFFFFFFull) {
        tgx_result_free(r);
        return fail(TGX_ERR_UNSUPPORTED, "frequency pass over more than 2^32-1 tokens per call");
    }
    if (T && m->vocab_size <= tgx::kHistMaxVocab && !knob("TGX_FREQ_SORT")) {
        // the vocabulary's counters fit a block's LDS: one private histogram per block (pairs.hip)
        unsigned long long* d_hist = nullptr;
        const size_t hb = (size_t)m->vocab_size * 8 + 256;
        auto done = [&](tgx_status s2) {
            if (s2 != TGX_OK) (void)hipStreamSynchronize(m->stream);
            pool_free(m->device, d_hist, hb);
            tgx_result_free(r);
            return s2;
        };
        if (pool_alloc(m->device, hb, (void**)&d_hist) != hipSuccess) return done(fail(TGX_ERR_DEVICE, "out of device memory (frequency pass)"));
        if (hipMemsetAsync(d_hist, 0, hb, m->stream) != hipSuccess) return done(fail(TGX_ERR_DEVICE, "memset failed"));
        time_begin(m, "ids_histogram_kernel");
        if (tgx::launch_ids_histogram(r->d_ids, T, m->vocab_size, d_hist, (uint32_t)m->num_cus, m->stream) != hipSuccess)
            return done(fail(TGX_ERR_DEVICE, "histogram launch failed"));
        time_end(m);
        std::vector<unsigned long long> hh(m->vocab_size);
        if (hipMemcpyAsync(hh.data(), d_hist, (size_t)m->vocab_size * 8, hipMemcpyDeviceToHost, m->stream) != hipSuccess ||
            hipStreamSynchronize(m->stream) != hipSuccess)
            return done(fail(TGX_ERR_DEVICE, "frequency pass failed: %s", hipGetErrorString(hipGetLastError())));
        for (uint32_t i = 0; i < m->vocab_size; i++) freq[i] += hh[i];
        m->last_alg_bytes = c->n_bytes + 8 * (c->n_samples + 1) + 8ull * m->vocab_size;
        return done(TGX_OK);
    }
    uint32_t *d_sorted = nullptr, *d_unique = nullptr;
    unsigned int *d_cnt = nullptr, *d_runs = nullptr;
    void* d_temp = nullptr;
    size_t tb1 = 0, tb2 = 0;
    const size_t kb = (size_t)T * 4 + 256;
    auto cleanup = [&](tgx_status s2) {
        // kernels already queued may still write these buffers: no other handle may take them from the pool yet
        if (s2 != TGX_OK) (void)hipStreamSynchronize(m->stream);
        pool_free(m->device, d_sorted, kb);
        pool_free(m->device, d_unique, kb);
        pool_free(m->device, d_cnt, kb);
        pool_free(m->device, d_runs, 256);
        pool_free(m->device, d_temp, std::max(tb1, tb2) + 256);
        tgx_result_free(r);
        return s2;
    };
    if (T == 0) return cleanup(TGX_OK);
    if (tgx::ids_sort_temp_bytes(T, &tb1) != hipSuccess || tgx::ids_rle_temp_bytes(T, &tb2) != hipSuccess)
        return cleanup(fail(TGX_ERR_DEVICE, "rocPRIM temp-size query failed"));
    if (pool_alloc(m->device, kb, (void**)&d_sorted) != hipSuccess ||
        pool_alloc(m->device, kb, (void**)&d_unique) != hipSuccess ||
        pool_alloc(m->device, kb, (void**)&d_cnt) != hipSuccess ||
        pool_alloc(m->device, 256, (void**)&d_runs) != hipSuccess ||
        pool_alloc(m->device, std::max(tb1, tb2) + 256, &d_temp) != hipSuccess)
        return cleanup(fail(TGX_ERR_DEVICE, "out of device memory (frequency pass)"));
    unsigned int end_bit = 1;
    while (end_bit < 32 && (1ull << end_bit) < (uint64_t)m->vocab_size) end_bit++;
    time_begin(m, "ids_sort+rle");
    if (tgx::ids_sort(d_temp, tb1, r->d_ids, d_sorted, T, end_bit, m->stream) != hipSuccess ||
        tgx::ids_rle(d_temp, tb2, d_sorted, T, d_unique, d_cnt, d_runs, m->stream) != hipSuccess)
        return cleanup(fail(TGX_ERR_DEVICE, "id sort / run-length encode failed"));
    time_end(m);
    unsigned int runs = 0;
    if (hipMemcpyAsync(&runs, d_runs, 4, hipMemcpyDeviceToHost, m->stream) != hipSuccess ||
        hipStreamSynchronize(m->stream) != hipSuccess)
        return cleanup(fail(TGX_ERR_DEVICE, "frequency pass failed: %s", hipGetErrorString(hipGetLastError())));
    std::vector<uint32_t> hu(runs ? runs : 1);
    std::vector<unsigned int> hc(runs ? runs : 1);
    if (runs && (hipMemcpy(hu.data(), d_unique, (size_t)runs * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                 hipMemcpy(hc.data(), d_cnt, (size_t)runs * 4, hipMemcpyDeviceToHost) != hipSuccess))
        return cleanup(fail(TGX_ERR_DEVICE, "D2H copy of the histogram failed"));
    for (unsigned int i = 0; i < runs; i++)
        if (hu[i] < m->vocab_size) freq[hu[i]] += hc[i];
    m->last_alg_bytes = c->n_bytes + 8 * (c->n_samples + 1) + 8ull * m->vocab_size;
    return cleanup(TGX_OK);
}

// max_pairs == 0: the whole table in ascending key order (tgx_count_pairs); otherwise the max_pairs most
// frequent pairs, by descending count and ascending key among equal counts (tgx_count_pairs_top)
static tgx_status count_pairs_impl(tgx_model* m, tgx_corpus* c, uint64_t max_pairs, uint64_t** keys, uint64_t** counts,
                                   uint64_t* n_pairs, uint64_t* n_total) {
    if (!m || !c || !keys || !counts || !n_pairs) return fail(TGX_ERR_INVALID, "tgx_count_pairs: NULL argument");
    *keys = *counts = nullptr;
    *n_pairs = 0;
    if (n_total) *n_total = 0;
    if (m->device != c->device) return fail(TGX_ERR_INVALID, "model and corpus on different devices");
    std::lock_guard<std::mutex> lk(m->mu);
    std::lock_guard<std::mutex> lkc(c->mu);
    tgx_result* r = nullptr;
    tgx_status st = encode_corpus_locked(m, c, 0.0, 0, &r);  // model.encode(sample, 0.0), merge.rs:58
    if (st != TGX_OK) return st;
    const uint64_t T = r->n_tokens, S = r->n_samples;
    if (T >= 0xFFFFFFFFull) {
        tgx_result_free(r);
        return fail(TGX_ERR_UNSUPPORTED, "pair scan over more than 2^32-1 tokens per pass");
    }
    unsigned long long *d_keys = nullptr, *d_sorted = nullptr, *d_unique = nullptr;
    unsigned int *d_cnt = nullptr, *d_runs = nullptr;
    void* d_temp = nullptr;
    size_t tb1 = 0, tb2 = 0;
    const size_t kb = (size_t)T * 8 + 256, cb = (size_t)T * 4 + 256;
    auto cleanup = [&](tgx_status s2) {
        // kernels already queued may still write these buffers: no other handle may take them from the pool yet
        if (s2 != TGX_OK) (void)hipStreamSynchronize(m->stream);
        pool_free(m->device, d_keys, kb);
        pool_free(m->device, d_sorted, kb);
        pool_free(m->device, d_unique, kb);
        pool_free(m->device, d_cnt, cb);
        pool_free(m->device, d_runs, 256);
        pool_free(m->device, d_temp, std::max(tb1, tb2) + 256);
        tgx_result_free(r);
        return s2;
    };
    if (T == 0) return cleanup(TGX_OK);
    if (tgx::pair_sort_temp_bytes(T, &tb1) != hipSuccess || tgx::pair_rle_temp_bytes(T, &tb2) != hipSuccess)
        return cleanup(fail(TGX_ERR_DEVICE, "rocPRIM temp-size query failed"));
    if (pool_alloc(m->device, kb, (void**)&d_keys) != hipSuccess ||
        pool_alloc(m->device, kb, (void**)&d_sorted) != hipSuccess ||
        pool_alloc(m->device, kb, (void**)&d_unique) != hipSuccess ||
        pool_alloc(m->device, cb, (void**)&d_cnt) != hipSuccess ||
        pool_alloc(m->device, 256, (void**)&d_runs) != hipSuccess ||
        pool_alloc(m->device, std::max(tb1, tb2) + 256, &d_temp) != hipSuccess)
        return cleanup(fail(TGX_ERR_DEVICE, "out of device memory (pair scan)"));
    const uint32_t blocks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((S + 3) / 4, (uint64_t)m->num_cus * 8));
    // pair keys of 2 shift (+ 1 for the sentinel) bits, shift = bits of the largest id; widened to the API's
    // (a << 32) | b on the host
    uint32_t shift = 1;
    while (shift < 32 && (1ull << shift) < (unsigned long long)m->vocab_size) shift++;
    const unsigned long long sentinel = shift < 32 ? (1ull << (2 * shift)) : ~0ULL;
    const unsigned int end_bit = shift < 32 ? 2 * shift + 1 : 64;
    time_begin(m, "pair_keys_kernel");
    if (tgx::launch_pair_keys(r->d_ids, r->d_offs, S, shift, sentinel, d_keys, blocks, m->stream) != hipSuccess)
        return cleanup(fail(TGX_ERR_DEVICE, "pair key launch failed"));
    time_end(m);
    time_begin(m, "pair_sort+rle");
    if (tgx::pair_sort(d_temp, tb1, d_keys, d_sorted, T, end_bit, m->stream) != hipSuccess ||
        tgx::pair_rle(d_temp, tb2, d_sorted, T, d_unique, d_cnt, d_runs, m->stream) != hipSuccess)
        return cleanup(fail(TGX_ERR_DEVICE, "pair sort / run-length encode failed"));
    time_end(m);
    unsigned int runs = 0;
    if (hipMemcpyAsync(&runs, d_runs, 4, hipMemcpyDeviceToHost, m->stream) != hipSuccess ||
        hipStreamSynchronize(m->stream) != hipSuccess)
        return cleanup(fail(TGX_ERR_DEVICE, "pair scan failed: %s", hipGetErrorString(hipGetLastError())));
    // the sentinel run (one per non-empty sample) has the largest key: it is the last run
    unsigned long long last_key = 0;
    if (runs && hipMemcpy(&last_key, d_unique + (runs - 1), 8, hipMemcpyDeviceToHost) != hipSuccess)
        return cleanup(fail(TGX_ERR_DEVICE, "D2H copy failed"));
    if (runs && last_key == sentinel) runs--;
    const unsigned long long* src_keys = d_unique;
    const unsigned int* src_cnt = d_cnt;
    if (n_total) *n_total = runs;
    if (max_pairs && runs) {
        // order by count on the device and hand back the head only: the merge loop looks at a few hundred
        // candidates of a table of millions (src/merge.rs:84-126)
        size_t tb3 = 0;
        if (tgx::pair_count_sort_temp_bytes(runs, &tb3) != hipSuccess)
            return cleanup(fail(TGX_ERR_DEVICE, "rocPRIM temp-size query failed"));
        void* d_temp3 = nullptr;
        if (pool_alloc(m->device, tb3 + 256, &d_temp3) != hipSuccess)
            return cleanup(fail(TGX_ERR_DEVICE, "out of device memory (pair scan)"));
        unsigned int* cnt_out = reinterpret_cast<unsigned int*>(d_sorted);  // the sorted keys are no longer needed
        time_begin(m, "pair_count_sort");
        hipError_t e3 = tgx::pair_count_sort(d_temp3, tb3, d_cnt, cnt_out, d_unique, d_keys, runs, m->stream);
        time_end(m);
        if (e3 == hipSuccess) e3 = hipStreamSynchronize(m->stream);
        pool_free(m->device, d_temp3, tb3 + 256);
        if (e3 != hipSuccess) return cleanup(fail(TGX_ERR_DEVICE, "pair count sort failed"));
        src_keys = d_keys;
        src_cnt = cnt_out;
        runs = (unsigned int)std::min<uint64_t>(runs, max_pairs);
    }
    uint64_t* ok = (uint64_t*)malloc(sizeof(uint64_t) * (runs ? runs : 1));
    uint64_t* oc = (uint64_t*)malloc(sizeof(uint64_t) * (runs ? runs : 1));
    if (!ok || !oc) {
        free(ok);
        free(oc);
        return cleanup(fail(TGX_ERR_DEVICE, "out of host memory (pair table of %llu entries)", (unsigned long long)runs));
    }
    if (!max_pairs && runs) {
        // the whole table (millions of pairs: the multi-GPU merge exchanges it): brought into the ABI's form on the
        // device and copied straight into the caller's arrays — two zero-filled staging vectors and a host loop
        // over every pair took three times the kernels' time.  d_sorted and d_keys are free by now.
        unsigned long long* ek = d_sorted;
        unsigned long long* ec = d_keys;
        if (tgx::launch_pair_expand(src_keys, src_cnt, runs, (uint32_t)shift, ek, ec, m->stream) != hipSuccess ||
            hipStreamSynchronize(m->stream) != hipSuccess ||
            copy_sync(ok, ek, (size_t)runs * 8, hipMemcpyDeviceToHost, m->device) != hipSuccess ||
            copy_sync(oc, ec, (size_t)runs * 8, hipMemcpyDeviceToHost, m->device) != hipSuccess) {
            free(ok);
            free(oc);
            return cleanup(fail(TGX_ERR_DEVICE, "D2H copy of pair table failed"));
        }
    } else {
        std::vector<unsigned long long> hk(runs ? runs : 1);
        std::vector<unsigned int> hc(runs ? runs : 1);
        if (runs && (hipMemcpy(hk.data(), src_keys, (size_t)runs * 8, hipMemcpyDeviceToHost) != hipSuccess ||
                     hipMemcpy(hc.data(), src_cnt, (size_t)runs * 4, hipMemcpyDeviceToHost) != hipSuccess)) {
            free(ok);
            free(oc);
            return cleanup(fail(TGX_ERR_DEVICE, "D2H copy of pair table failed"));
        }
        const unsigned long long low = shift < 32 ? (1ull << shift) - 1 : 0xFFFFFFFFull;
        for (unsigned int i = 0; i < runs; i++) {
            ok[i] = ((hk[i] >> shift) << 32) | (hk[i] & low);
            oc[i] = hc[i];
        }
    }
    *keys = ok;
    *counts = oc;
    *n_pairs = runs;
    // N + 8(S+1) + 16P
    m->last_alg_bytes = c->n_bytes + 8 * (S + 1) + 16ull * runs;
    return cleanup(TGX_OK);
}

tgx_status tgx_count_pairs(tgx_model* m, tgx_corpus* c, uint64_t** keys, uint64_t** counts, uint64_t* n_pairs) {
    return count_pairs_impl(m, c, 0, keys, counts, n_pairs, nullptr);
}

tgx_status tgx_count_pairs_top(tgx_model* m, tgx_corpus* c, uint64_t max_pairs, uint64_t** keys, uint64_t** counts,
                               uint64_t* n_pairs, uint64_t* n_total) {
    if (max_pairs == 0) return fail(TGX_ERR_INVALID, "tgx_count_pairs_top: max_pairs must be positive");
    return count_pairs_impl(m, c, max_pairs, keys, counts, n_pairs, n_total);
}

// Builds (once) the double-array of the REVERSED tokens used by the backward sweep.
static tgx_status ensure_reverse_trie(tgx_model* m) {
    if (m->d_trie_rev) return TGX_OK;
    const uint32_t V = m->vocab_size;
    if (!m->rev_host_built) {
        std::vector<uint8_t> rev(m->vocab_bytes.size());
        for (uint32_t i = 0; i < V; i++) {
            const uint64_t b = m->vocab_offs[i], e = m->vocab_offs[i + 1];
            for (uint64_t k = b; k < e; k++) rev[k] = m->vocab_bytes[e - 1 - (k - b)];
        }
        tgx::build_flat_trie(rev.data(), m->vocab_offs.data(), m->vocab_scores.data(), V, &m->flat_rev);
    }
    if (m->flat_rev.table.size() >= (1u << 26)) return fail(TGX_ERR_UNSUPPORTED, "reversed trie needs more than 2^26 slots");
    HIP_TRY(hipSetDevice(m->device));
    const size_t tbytes = m->flat_rev.table.size() * sizeof(tgx::TrieRec);
    HIP_TRY(hipMalloc(&m->d_trie_rev, tbytes));
    HIP_TRY(hipMemcpy(m->d_trie_rev, m->flat_rev.table.data(), tbytes, hipMemcpyHostToDevice));
    int occ = 0;
    HIP_TRY(tgx::estep_max_blocks_per_cu(m->lm, &occ));
    m->estep_blocks_per_cu = std::max(1, std::min(occ, 16));
    HIP_TRY(tgx::estep4_prepare());
    // Linear-domain E-step (estep4l.hip): its tables carry w = exp(score).  Whether a pass can use it
    // (every position has an incoming token, values stay inside the f64 range) is decided per pass by
    // the forward kernel itself; scores beyond +-300 would overflow exp() or underflow within a block.
    bool ok = m->lm <= 32 && m->scores_finite;  // (17..32 bytes: the long-token builds, one position per lane)
    for (uint32_t i = 0; ok && i < V; i++) ok = m->vocab_scores[i] >= -300.0 && m->vocab_scores[i] <= 300.0;
    if (ok) {
        auto upload_weights = [&](const tgx::FlatTrie& ft, void** dst) -> hipError_t {
            std::vector<tgx::TrieRec> w(ft.table);
            for (tgx::TrieRec& r : w) {
                double sc = 0.0, wt = 0.0;
                if (r.base & tgx::kTerminalBit) {
                    memcpy(&sc, &r.score_bits, 8);
                    wt = std::exp(sc);
                }
                memcpy(&r.score_bits, &wt, 8);
            }
            hipError_t e = hipMalloc(dst, w.size() * sizeof(tgx::TrieRec));
            if (e != hipSuccess) return e;
            return hipMemcpy(*dst, w.data(), w.size() * sizeof(tgx::TrieRec), hipMemcpyHostToDevice);
        };
        HIP_TRY(upload_weights(m->flat, &m->d_trie_w));
        HIP_TRY(upload_weights(m->flat_rev, &m->d_trie_rev_w));
        HIP_TRY(tgx::estep4l_prepare());
    }
    m->estep_linear_ok = ok;
    return TGX_OK;
}

// The 8-byte ranked records and the table of w = exp(score value) by rank for estep5_fwd_kernel (encode5.hip): built at
// the first E-step of a model whose vocabulary allows them (tokens of at most 16 bytes, finite scores, at most 65 535
// distinct values, at most 2^21 slots); models created for E-step passes do not have the encode tables.
static tgx_status ensure_estep_trie8(tgx_model* m) {
    if (m->estep_trie8_tried) return TGX_OK;
    m->estep_trie8_tried = true;
    if (!(m->lm <= 16 && m->scores_finite && m->vocab_size && m->flat.table.size() <= tgx::kTrie8MaxSlots)) return TGX_OK;
    {   // more than 65 535 distinct values (every model of a prune run above that many tokens): known after a fraction
        // of a millisecond, not after build_trie8 has sorted them
        std::unordered_set<uint64_t> seen;
        seen.reserve(1u << 17);
        for (uint32_t i = 0; i < m->vocab_size && seen.size() <= tgx::kTrie8MaxValues; i++) {
            uint64_t b;
            memcpy(&b, &m->vocab_scores[i], 8);
            seen.insert(b);
        }
        if (seen.size() > tgx::kTrie8MaxValues) return TGX_OK;
    }
    tgx::Trie8 t8;
    tgx::build_trie8(m->flat, m->vocab_offs.data(), m->vocab_scores.data(), &t8);
    if (!t8.ok) return TGX_OK;
    const size_t ns = t8.rec.size(), nv = t8.values.size();
    std::vector<double> w(nv);
    w[0] = 0.0;  // "no token"
    for (size_t r = 1; r < nv; r++) {
        double v;
        memcpy(&v, &t8.values[r], 8);
        w[r] = std::exp(v);  // the same function of the same double as the weights of the 16-byte tables (upload_weights)
    }
    if (!m->d_trie8) HIP_TRY(hipMalloc(&m->d_trie8, ns * sizeof(tgx::Trie8Rec)));
    if (!m->d_values) HIP_TRY(hipMalloc((void**)&m->d_values, nv * 8));
    if (!m->d_wvalues) HIP_TRY(hipMalloc((void**)&m->d_wvalues, nv * 8));
    HIP_TRY(hipMemcpy(m->d_trie8, t8.rec.data(), ns * sizeof(tgx::Trie8Rec), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(m->d_values, t8.values.data(), nv * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(m->d_wvalues, w.data(), nv * 8, hipMemcpyHostToDevice));
    m->n_values = (uint32_t)nv - 1u;
    m->root_base8 = t8.root_base;
    if (m->values_ranked || m->value_coverage.empty()) m->value_coverage = std::move(t8.coverage);
    m->values_ranked = false;  // (the tables just written are in build_trie8's order again: the next encode re-ranks them)
    m->have_wvalues = true;
    return TGX_OK;
}

// The E-step's work list of a corpus: every sample cut at multiples of snippet_len (src/prune.rs:83), longest snippet
// first, with its device copies; kept with the corpus (caller holds both locks).
static tgx_status ensure_estep_work(tgx_model* m, tgx_corpus* c, uint64_t snippet_len) {
    tgx_corpus::EstepWork& es = c->es;
    if (es.snippet_len == snippet_len) return TGX_OK;
    const uint64_t S = c->n_samples, N = c->n_bytes;
    {
        pool_free(c->device, es.d_soffs, es.obytes);
        pool_free(c->device, es.d_sbase, es.obytes);
        pool_free(c->device, es.d_order, es.ordbytes);
        pool_free(c->device, es.d_ssample, es.ordbytes);
        pool_free(c->device, es.d_win_snip, es.winbytes);
        pool_free(c->device, es.d_win_k, es.winbytes);
        es.d_soffs = es.d_sbase = nullptr;
        es.d_order = es.d_ssample = nullptr;
        es.d_win_snip = es.d_win_k = nullptr;
        es.window = 0;
        es.n_windows = 0;
        es.snippet_len = 0;
        es.soffs.clear();
        es.ssample.clear();
        es.sbase.clear();
        es.soffs.reserve(S + N / snippet_len + 2);
        for (uint64_t i = 0; i < S; i++) {
            const uint64_t b = c->h_offs[i], e = c->h_offs[i + 1];
            for (uint64_t o = b; o < e; o += snippet_len) {
                es.soffs.push_back(o);
                es.ssample.push_back((uint32_t)i);
                es.sbase.push_back(o - b);
            }
        }
        const uint64_t K0 = es.soffs.size();
        es.soffs.push_back(N);
        // a snippet's end is the next snippet's start except across empty samples: offsets stay exact
        // because snippets tile [0, N) in order
        es.order.resize(K0);
        std::iota(es.order.begin(), es.order.end(), 0u);
        const std::vector<uint64_t>& so = es.soffs;
        std::stable_sort(es.order.begin(), es.order.end(), [&so](uint32_t a, uint32_t b) {
            return so[a + 1] - so[a] > so[b + 1] - so[b];
        });
        es.obytes = (size_t)(K0 + 1) * 8 + 256;
        es.ordbytes = (size_t)K0 * 4 + 256;
        if (pool_alloc(c->device, es.obytes, (void**)&es.d_soffs) != hipSuccess ||
            pool_alloc(c->device, es.obytes, (void**)&es.d_sbase) != hipSuccess ||
            pool_alloc(c->device, es.ordbytes, (void**)&es.d_order) != hipSuccess ||
            pool_alloc(c->device, es.ordbytes, (void**)&es.d_ssample) != hipSuccess)
            return fail(TGX_ERR_DEVICE, "out of device memory (E-step work list)");
        if (hipMemcpyAsync(es.d_soffs, es.soffs.data(), (K0 + 1) * 8, hipMemcpyHostToDevice, m->stream) != hipSuccess ||
            (K0 && hipMemcpyAsync(es.d_sbase, es.sbase.data(), K0 * 8, hipMemcpyHostToDevice, m->stream) != hipSuccess) ||
            (K0 && hipMemcpyAsync(es.d_order, es.order.data(), K0 * 4, hipMemcpyHostToDevice, m->stream) != hipSuccess) ||
            (K0 && hipMemcpyAsync(es.d_ssample, es.ssample.data(), K0 * 4, hipMemcpyHostToDevice, m->stream) != hipSuccess) ||
            hipStreamSynchronize(m->stream) != hipSuccess)
            return fail(TGX_ERR_DEVICE, "E-step work list upload failed");
        es.snippet_len = snippet_len;
    }
    return TGX_OK;
}

// ---- pieces (cuts.hip): long snippets cut where no token match crosses — the lattice factorises there
struct EstepPieces {
    uint64_t n = 0, longest = 0, cap = 0;
    uint64_t *d_bound = nullptr, *d_pos = nullptr, *d_offs = nullptr, *d_base = nullptr;
    uint32_t *d_flag = nullptr, *d_sample = nullptr, *d_snip = nullptr, *d_len = nullptr, *d_idx = nullptr, *d_len2 = nullptr, *d_order = nullptr;
    void *d_scan = nullptr, *d_sort = nullptr;
    double* d_zsnip = nullptr;
    size_t scan_bytes = 0, sort_bytes = 0, zsnip_bytes = 0;
};
static void estep_pieces_free(tgx_model* m, EstepPieces& pc) {
    pool_free(m->device, pc.d_bound, pc.cap * 8 + 256);
    pool_free(m->device, pc.d_pos, (pc.cap + 1) * 8 + 256);
    pool_free(m->device, pc.d_offs, (pc.cap + 1) * 8 + 256);
    pool_free(m->device, pc.d_base, pc.cap * 8 + 256);
    pool_free(m->device, pc.d_flag, (pc.cap + 1) * 4 + 256);
    pool_free(m->device, pc.d_sample, pc.cap * 4 + 256);
    pool_free(m->device, pc.d_snip, pc.cap * 4 + 256);
    pool_free(m->device, pc.d_len, pc.cap * 4 + 256);
    pool_free(m->device, pc.d_idx, pc.cap * 4 + 256);
    pool_free(m->device, pc.d_len2, pc.cap * 4 + 256);
    pool_free(m->device, pc.d_order, pc.cap * 4 + 256);
    pool_free(m->device, pc.d_scan, pc.scan_bytes);
    pool_free(m->device, pc.d_sort, pc.sort_bytes);
    pool_free(m->device, pc.d_zsnip, pc.zsnip_bytes);
    pc = EstepPieces{};
}
// the windows of the corpus's snippets (one boundary is sought per window), kept with the work list
static tgx_status ensure_estep_windows(tgx_model* m, tgx_corpus* c, uint32_t window) {
    tgx_corpus::EstepWork& es = c->es;
    if (es.window == window && es.d_win_snip) return TGX_OK;
    const uint64_t K = es.soffs.size() - 1, N = c->n_bytes;
    pool_free(c->device, es.d_win_snip, es.winbytes);
    pool_free(c->device, es.d_win_k, es.winbytes);
    es.d_win_snip = es.d_win_k = nullptr;
    es.window = 0;
    std::vector<uint32_t> ws, wk;
    ws.reserve(K + N / window + 2);
    wk.reserve(K + N / window + 2);
    for (uint64_t k = 0; k < K; k++) {
        const uint64_t len = es.soffs[k + 1] - es.soffs[k];
        for (uint64_t j = 0; j * window < len; j++) {
            ws.push_back((uint32_t)k);
            wk.push_back((uint32_t)j);
        }
    }
    es.n_windows = ws.size();
    es.winbytes = es.n_windows * 4 + 256;
    if (pool_alloc(c->device, es.winbytes, (void**)&es.d_win_snip) != hipSuccess ||
        pool_alloc(c->device, es.winbytes, (void**)&es.d_win_k) != hipSuccess)
        return fail(TGX_ERR_DEVICE, "out of device memory (E-step windows)");
    if (hipMemcpyAsync(es.d_win_snip, ws.data(), es.n_windows * 4, hipMemcpyHostToDevice, m->stream) != hipSuccess ||
        hipMemcpyAsync(es.d_win_k, wk.data(), es.n_windows * 4, hipMemcpyHostToDevice, m->stream) != hipSuccess ||
        hipStreamSynchronize(m->stream) != hipSuccess)
        return fail(TGX_ERR_DEVICE, "E-step window list upload failed");
    es.window = window;
    return TGX_OK;
}
// cut_windows_kernel, scan, scatter, lengths, longest-first order: the piece list of this pass (dropout decides the
// matches, so the list is per pass).  trie16: a 16-byte-record table of the forward tokens (scores or weights alike).
static tgx_status estep_pieces_build(tgx_model* m, tgx_corpus* c, const void* trie16, double dropout, uint64_t seed, EstepPieces* out) {
    tgx_corpus::EstepWork& es = c->es;
    EstepPieces& pc = *out;
    const uint64_t K = es.soffs.size() - 1, N = c->n_bytes;
    const uint64_t W = es.n_windows;
    pc.cap = W;
    pc.zsnip_bytes = (size_t)K * 8 + 256;
    auto bad = [&](const char* what) {
        (void)hipStreamSynchronize(m->stream);
        estep_pieces_free(m, pc);
        return fail(TGX_ERR_DEVICE, "E-step pieces: %s", what);
    };
    if (tgx::scan_temp_bytes(W, &pc.scan_bytes) != hipSuccess || tgx::piece_sort_temp_bytes(W, &pc.sort_bytes) != hipSuccess)
        return bad("scratch sizes");
    if (pool_alloc(m->device, W * 8 + 256, (void**)&pc.d_bound) != hipSuccess ||
        pool_alloc(m->device, (W + 1) * 8 + 256, (void**)&pc.d_pos) != hipSuccess ||
        pool_alloc(m->device, (W + 1) * 8 + 256, (void**)&pc.d_offs) != hipSuccess ||
        pool_alloc(m->device, W * 8 + 256, (void**)&pc.d_base) != hipSuccess ||
        pool_alloc(m->device, (W + 1) * 4 + 256, (void**)&pc.d_flag) != hipSuccess ||
        pool_alloc(m->device, W * 4 + 256, (void**)&pc.d_sample) != hipSuccess ||
        pool_alloc(m->device, W * 4 + 256, (void**)&pc.d_snip) != hipSuccess ||
        pool_alloc(m->device, W * 4 + 256, (void**)&pc.d_len) != hipSuccess ||
        pool_alloc(m->device, W * 4 + 256, (void**)&pc.d_idx) != hipSuccess ||
        pool_alloc(m->device, W * 4 + 256, (void**)&pc.d_len2) != hipSuccess ||
        pool_alloc(m->device, W * 4 + 256, (void**)&pc.d_order) != hipSuccess ||
        (pc.scan_bytes && pool_alloc(m->device, pc.scan_bytes, &pc.d_scan) != hipSuccess) ||
        (pc.sort_bytes && pool_alloc(m->device, pc.sort_bytes, &pc.d_sort) != hipSuccess) ||
        pool_alloc(m->device, pc.zsnip_bytes, (void**)&pc.d_zsnip) != hipSuccess)
        return bad("out of device memory");
    tgx::CutParams q{};
    q.text = c->d_text;
    q.soffs = es.d_soffs;
    q.snip_sample = es.d_ssample;
    q.snip_base = es.d_sbase;
    q.win_snip = es.d_win_snip;
    q.win_k = es.d_win_k;
    q.n_windows = W;
    q.window = es.window;
    q.trie = trie16;
    q.root = m->flat.table[0].base & ~tgx::kTerminalBit;
    q.lmx = m->lm > 16 ? 32u : 16u;
    q.dropout = dropout;
    q.seed = seed;
    q.bound = pc.d_bound;
    q.flag = pc.d_flag;
    unsigned long long* const d_longest = m->d_ctrl + 6;
    time_begin(m, "cut_windows_kernel");
    if (tgx::launch_cut_windows(q, m->stream) != hipSuccess) return bad("cut kernel launch failed");
    time_end(m);
    time_begin(m, "piece_list");
    if (hipMemsetAsync(d_longest, 0, 8, m->stream) != hipSuccess || hipMemsetAsync(pc.d_zsnip, 0, pc.zsnip_bytes, m->stream) != hipSuccess ||
        tgx::launch_scan(pc.d_flag, pc.d_pos, W, pc.d_scan, pc.scan_bytes, m->stream) != hipSuccess ||
        tgx::launch_cut_scatter(q, pc.d_pos, N, pc.d_offs, pc.d_sample, pc.d_base, pc.d_snip, m->stream) != hipSuccess ||
        tgx::launch_piece_len(pc.d_offs, pc.d_pos + W, pc.d_len, pc.d_idx, d_longest, W, m->stream) != hipSuccess ||
        tgx::piece_sort(pc.d_sort, pc.sort_bytes, pc.d_len, pc.d_len2, pc.d_idx, pc.d_order, W, m->stream) != hipSuccess)
        return bad("piece list kernels failed");
    time_end(m);
    unsigned long long h_n = 0, h_longest = 0;
    if (hipMemcpyAsync(&h_n, pc.d_pos + W, 8, hipMemcpyDeviceToHost, m->stream) != hipSuccess ||
        hipMemcpyAsync(&h_longest, d_longest, 8, hipMemcpyDeviceToHost, m->stream) != hipSuccess ||
        hipStreamSynchronize(m->stream) != hipSuccess)
        return bad("piece list read-back failed");
    if (h_n < K || h_n > W) return bad("inconsistent piece count");
    pc.n = h_n;
    pc.longest = h_longest;
    return TGX_OK;
}

// The token-ranked records and weights of estep7_kernel (estep7.hip): vocabularies with finite scores within +-300
// (w = exp(score) and its products must stay inside the f64 range, as for the other linear-domain kernels) and tokens of
// at most 16 bytes.  ~40 ns of host time per token.
// The 8-byte records of estep7_kernel.  Built at the model's first E-step, where there is text: the ranks that the kernels
// keep in LDS are the tokens that MATCH most often in a sample of the corpus (match_count_kernel) — every match is one add
// to the token's expected count, whatever its score, and adds outside LDS are memory-side atomics that queue up per
// address.  (Ranked by exp(score) / length, prune's second sub-iteration at 500 000 entries — M-step scores — left one
// token with 1.9 M matches outside: 61 ms against the first sub-iteration's 13; ranked by the probability mass below
// the token's node, 74 ms.)
static tgx_status ensure_estep_trie8t(tgx_model* m, const tgx_corpus* c) {
    if (m->trie8t_tried) return TGX_OK;
    m->trie8t_tried = true;
    if (!(m->lm <= 16 && m->scores_finite && m->vocab_size && m->flat.table.size() <= tgx::kTrie8TMaxSlots)) return TGX_OK;
    for (uint32_t i = 0; i < m->vocab_size; i++)
        if (!(m->vocab_scores[i] >= -300.0 && m->vocab_scores[i] <= 300.0)) return TGX_OK;
    tgx::HostPhases hp("ensure_estep_trie8t");
    tgx::Trie8T t8;
    tgx::build_trie8t(m->flat, m->vocab_offs.data(), m->vocab_scores.data(), &t8);
    if (!t8.ok || t8.n_tok == 0) return TGX_OK;
    hp.mark("build_trie8t");
    HIP_TRY(hipSetDevice(m->device));
    const size_t ns = t8.rec.size(), nw = t8.w.size();
    if (!m->d_trie8t) HIP_TRY(hipMalloc(&m->d_trie8t, ns * sizeof(tgx::Trie8TRec)));
    if (!m->d_wtab) HIP_TRY(hipMalloc((void**)&m->d_wtab, nw * 8));
    HIP_TRY(hipMemcpy(m->d_trie8t, t8.rec.data(), ns * sizeof(tgx::Trie8TRec), hipMemcpyHostToDevice));
    hp.mark("upload records");
    const char* norank = knob("TGX_E7_RANK");  // "model": keep build_trie8t's order (measurements)
    if (c && c->n_bytes && !(norank && strcmp(norank, "model") == 0)) {
        // up to 64 chunks of 64 KiB spread over the corpus (4 MiB: ~13 M matches)
        const uint32_t chunk = 65536u;
        const uint64_t stride = std::max<uint64_t>(chunk, (c->n_bytes + 63) / 64);
        unsigned int* d_cnt = nullptr;
        uint32_t* d_perm = nullptr;
        const size_t cb = nw * 4 + 256;
        auto drop = [&]() {
            pool_free(m->device, d_cnt, cb);
            pool_free(m->device, d_perm, cb);
        };
        if (pool_alloc(m->device, cb, (void**)&d_cnt) != hipSuccess || pool_alloc(m->device, cb, (void**)&d_perm) != hipSuccess) {
            drop();
            return fail(TGX_ERR_DEVICE, "out of device memory (match counts)");
        }
        std::vector<unsigned int> cnt(nw, 0u);
        if (hipMemsetAsync(d_cnt, 0, cb, m->stream) != hipSuccess ||
            tgx::launch_match_count(c->d_text, c->n_bytes, chunk, stride, m->d_trie8t, (uint32_t)ns, t8.root_base, t8.n_tok, d_cnt, (uint32_t)m->num_cus, m->stream) != hipSuccess ||
            hipMemcpyAsync(cnt.data(), d_cnt, nw * 4, hipMemcpyDeviceToHost, m->stream) != hipSuccess || hipStreamSynchronize(m->stream) != hipSuccess) {
            drop();
            return fail(TGX_ERR_DEVICE, "match count pass failed: %s", hipGetErrorString(hipGetLastError()));
        }
        hp.mark("match counts");
        // new order: the kTrie8TSortedRanks most matched tokens by descending count (ties: the old rank), the others as they were
        const uint32_t n_tok = t8.n_tok;
        std::vector<uint32_t> order(n_tok);
        for (uint32_t r = 0; r < n_tok; r++) order[r] = r + 1u;
        const auto hotter = [&](uint32_t a, uint32_t b) { return cnt[a] != cnt[b] ? cnt[a] > cnt[b] : a < b; };
        const size_t head = std::min<size_t>(n_tok, tgx::kTrie8TSortedRanks);
        if (head < n_tok) {
            std::nth_element(order.begin(), order.begin() + (long)head, order.end(), hotter);
            // the others in their old order, without a sort: mark the head, read the marks in order
            std::vector<uint8_t> in_head(nw, 0);
            for (size_t i = 0; i < head; i++) in_head[order[i]] = 1;
            size_t k = head;
            for (uint32_t r = 1; r <= n_tok; r++)
                if (!in_head[r]) order[k++] = r;
        }
        std::sort(order.begin(), order.begin() + (long)head, hotter);
        std::vector<uint32_t> perm(nw, 0u);
        std::vector<double> w2(nw, 0.0);
        std::vector<uint32_t> id2(nw, tgx::kNoToken);
        unsigned long long all_matches = 0, far_matches = 0;
        for (uint32_t r = 0; r < n_tok; r++) {
            perm[order[r]] = r + 1u;
            w2[r + 1u] = t8.w[order[r]];
            id2[r + 1u] = t8.id_of_rank[order[r]];
            all_matches += cnt[order[r]];
            if (r + 1u > 65535u) far_matches += cnt[order[r]];
        }
        m->e7_overflow_share = all_matches ? (double)far_matches / (double)all_matches : 0.0;
        t8.w.swap(w2);
        t8.id_of_rank.swap(id2);
        if (hipMemcpyAsync(d_perm, perm.data(), nw * 4, hipMemcpyHostToDevice, m->stream) != hipSuccess ||
            tgx::launch_rank_remap(m->d_trie8t, (uint32_t)ns, d_perm, m->stream) != hipSuccess || hipStreamSynchronize(m->stream) != hipSuccess) {
            drop();
            return fail(TGX_ERR_DEVICE, "rank remap failed: %s", hipGetErrorString(hipGetLastError()));
        }
        drop();
        hp.mark("re-rank");
    }
    HIP_TRY(hipMemcpy(m->d_wtab, t8.w.data(), nw * 8, hipMemcpyHostToDevice));
    m->id_of_rank = std::move(t8.id_of_rank);
    m->n_tok7 = t8.n_tok;
    m->root_base7 = t8.root_base;
    m->have_trie8t = true;
    return TGX_OK;
}

// E-step on estep7_kernel (estep7.hip): one walk per position, every trip of a row a lattice of its own.  Caller holds
// both locks.  `fallback`: the chained kernels must do the pass (no records for this vocabulary, a position nothing
// reaches, a value out of range); `expected` is untouched then.
static tgx_status estep_fused(tgx_model* m, tgx_corpus* c, uint64_t snippet_len, double dropout, uint64_t seed, double* expected,
                              double* logz_sum, bool* fallback) {
    *fallback = false;
    tgx::HostPhases hp("estep_fused");
    {
        const tgx_status tst = ensure_estep_trie8t(m, c);
        if (tst != TGX_OK) return tst;
        if (!m->have_trie8t) {
            *fallback = true;
            return TGX_OK;
        }
        hp.mark("trie8t");
        const tgx_status wst = ensure_estep_work(m, c, snippet_len);
        if (wst != TGX_OK) return wst;
        hp.mark("work list");
    }
    tgx_corpus::EstepWork& es = c->es;
    const uint64_t S = c->n_samples, N = c->n_bytes, K = es.soffs.size() - 1;
    m->last_estep_pieces = 0;
    m->last_estep_redo = 0;
    if (K == 0) {
        if (logz_sum) *logz_sum = 0.0;
        m->last_alg_bytes = N + 8 * (S + 1) + 8ull * m->vocab_size;
        return TGX_OK;
    }
    // A piece is a chain of trips (~0.12 us per byte: a 64 KiB snippet takes ~8 ms) while a pass runs at ~33 GB/s: long
    // snippets are cut where no match crosses (cuts.hip) when the longest chain is a sizeable part of the pass — 256 MiB:
    // 21.2 -> 8.6 ms; 1 GiB: 33.6 ms uncut, 34.3 cut (the cut search and the piece list cost 0.5 - 0.8 ms).  Windows of
    // 4 KiB: 8.64 ms at 256 MiB against 8.91 with 2 KiB and 9.23 with 1 KiB (profiles/r04)
    EstepPieces pc;
    bool pieces = false;
    {
        uint32_t window = 4096;
        if (const char* e = knob("TGX_ESTEP_WINDOW")) window = (uint32_t)std::min(1 << 20, std::max(256, atoi(e)));
        const double longest0 = (double)(es.soffs[es.order[0] + 1] - es.soffs[es.order[0]]);
        pieces = longest0 > 4.0 * window && longest0 * 0.12e-6 > 0.4 * ((double)N / 33e9);
        if (const char* e = knob("TGX_ESTEP_PIECES")) pieces = atoi(e) != 0 && longest0 > (double)window;
        if (pieces) {
            tgx_status pst = ensure_estep_windows(m, c, window);
            if (pst != TGX_OK) return pst;
            pst = estep_pieces_build(m, c, m->d_trie, dropout, seed, &pc);
            if (pst != TGX_OK) return pst;
            m->last_estep_pieces = pc.n;
        }
    }
    const uint64_t units = pieces ? pc.n : K;
    // 16-bit match entries hold 65 535 ranks.  A vocabulary of a few more tokens (the usual "64 K": 65 536) keeps them — 34
    // against 24 GB/s with 32-bit entries — and leaves the trips in which one of its least matched tokens (the ranks beyond
    // 65 535: the tokens are ranked by match counts) matches to the redo kernel, which has 32-bit entries then.
    // TGX_E7_OVF_AT=<rank> (tests) makes the ranks beyond <rank> such tokens.
    uint32_t ovf_limit = 0xFFFFFFFFu;
    // (only while those tokens are rare in the text: a trip holds ~150 matches, the redo kernel is several times slower per trip)
    if (m->n_tok7 > 65535u && m->n_tok7 <= 65535u + 1024u && m->e7_overflow_share < 1e-4) ovf_limit = 65535u;
    if (const char* e = knob("TGX_E7_OVF_AT")) {
        const long v = atol(e);
        if (v >= 1 && v <= 65535 && m->n_tok7 <= 65535u + 1024u) ovf_limit = (uint32_t)v;
        if (v == 0) ovf_limit = 0xFFFFFFFFu;
    }
    bool ovf = ovf_limit < m->n_tok7;
    const bool wide = m->n_tok7 > 65535u && !ovf;
    const bool wide_redo = wide || ovf;
    // (1 GiB, 32 000 entries: 12 waves x 4 positions per lane 32.5 ms, 10 waves 36.2, 8 waves 41.7; three positions per lane
    // 34.2 / 38.5 / 44.3 and forty times the stretches without a cut in a trip — profiles/r04)
    int ppl = wide ? 2 : 4, waves = 12;
    const bool ppl_forced = ovf;  // (the overflow build exists for four positions per lane)
    if (const char* e = knob("TGX_EPPL")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 4) ppl = v;
    }
    if (ppl_forced) ppl = 4;
    if (const char* e = knob("TGX_E7_WAVES")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 12) waves = v;
    }
    {   // fewer waves when the pass has fewer units than the chip has rows, so that they spread over the CUs
        const uint64_t rows_wanted = (units + (uint64_t)m->num_cus - 1) / (uint64_t)m->num_cus;
        waves = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)waves, (rows_wanted + 3) / 4));
    }
    // (32-bit entries take twice the LDS: the geometry must leave room for a table worth having)
    while (waves > 1 && tgx::estep7_max_hot(wide, waves, ppl, 160u * 1024u) < 1024u) waves--;
    uint32_t n_hot = std::min(m->n_tok7, tgx::estep7_max_hot(wide, waves, ppl, 160u * 1024u));
    if (const char* e = knob("TGX_E7_HOT")) {
        const int v = atoi(e);
        if (v >= 0) n_hot = std::min(n_hot, (uint32_t)v);
    }
    if (ovf && n_hot >= m->n_tok7) n_hot = m->n_tok7 - 1u;  // (the overflow build is a COLD build)
    const size_t ebytes = ((size_t)m->n_tok7 + 1) * 8 + 256, zbytes = (size_t)K * 8 + 256;
    // (the redo list: a piece can leave several stretches, each longer than a trip's reach)
    const uint64_t redo_cap = units + N / 32 + 16;
    const size_t r8 = (size_t)redo_cap * 2 * 8 + 256, r4 = (size_t)redo_cap * 2 * 4 + 256;
    double *d_exp = nullptr, *d_z = nullptr, *d_zsnip = nullptr;
    void* d_work = nullptr;  // Estep7Work in device memory (kernels.h)
    uint64_t *d_roffs = nullptr, *d_rbase = nullptr;
    uint32_t *d_rsample = nullptr, *d_rsnip = nullptr;
    auto cleanup = [&](tgx_status s2) {
        if (s2 != TGX_OK) (void)hipStreamSynchronize(m->stream);
        estep_pieces_free(m, pc);
        pool_free(m->device, d_exp, ebytes);
        pool_free(m->device, d_z, 256);
        pool_free(m->device, d_work, 512);
        pool_free(m->device, d_zsnip, zbytes);
        pool_free(m->device, d_roffs, r8);
        pool_free(m->device, d_rbase, r8);
        pool_free(m->device, d_rsample, r4);
        pool_free(m->device, d_rsnip, r4);
        return s2;
    };
    static_assert(sizeof(tgx::Estep7Work) <= 512, "Estep7Work scratch");
    if (pool_alloc(m->device, ebytes, (void**)&d_exp) != hipSuccess || pool_alloc(m->device, 256, (void**)&d_z) != hipSuccess ||
        pool_alloc(m->device, 512, &d_work) != hipSuccess || pool_alloc(m->device, zbytes, (void**)&d_zsnip) != hipSuccess || pool_alloc(m->device, r8, (void**)&d_roffs) != hipSuccess ||
        pool_alloc(m->device, r8, (void**)&d_rbase) != hipSuccess || pool_alloc(m->device, r4, (void**)&d_rsample) != hipSuccess ||
        pool_alloc(m->device, r4, (void**)&d_rsnip) != hipSuccess)
        return cleanup(fail(TGX_ERR_DEVICE, "out of device memory (E-step scratch)"));
    if (hipMemsetAsync(d_exp, 0, ebytes, m->stream) != hipSuccess || hipMemsetAsync(d_z, 0, 256, m->stream) != hipSuccess ||
        hipMemsetAsync(d_zsnip, 0, zbytes, m->stream) != hipSuccess || hipMemsetAsync(m->d_ctrl + 3, 0x00, 24, m->stream) != hipSuccess ||
        hipMemsetAsync(m->d_ctrl + 1, 0xFF, 8, m->stream) != hipSuccess)
        return cleanup(fail(TGX_ERR_DEVICE, "E-step setup failed"));
    tgx::Estep7Params p{};
    tgx::Estep7Work& wk = p.host_work;
    p.text = c->d_text;
    p.work = reinterpret_cast<const tgx::Estep7Work*>(d_work);
    wk.soffs = pieces ? pc.d_offs : es.d_soffs;
    wk.order = pieces ? pc.d_order : es.d_order;
    wk.n_snips = units;
    wk.snip_sample = pieces ? pc.d_sample : es.d_ssample;
    wk.snip_base = pieces ? pc.d_base : es.d_sbase;
    wk.snip_of = pieces ? pc.d_snip : nullptr;
    p.trie8t = m->d_trie8t;
    p.n_slots = (uint32_t)m->flat.table.size();
    p.root_base = m->root_base7;
    p.wtab = m->d_wtab;
    p.n_tok = m->n_tok7;
    p.n_hot = n_hot;
    p.ovf_limit = ovf ? ovf_limit : 0xFFFFFFFFu;
    p.expected = d_exp;
    wk.zsnip = d_zsnip;
    wk.logz_sum = d_z;
    wk.queue = m->d_ctrl + 3;
    wk.redo_count = m->d_ctrl + 4;
    wk.range_flag = m->d_ctrl + 5;
    wk.redo_offs = d_roffs;
    wk.redo_sample = d_rsample;
    wk.redo_base = d_rbase;
    wk.redo_snip = d_rsnip;
    wk.redo_cap = redo_cap;
    p.dropout = dropout;
    p.seed = seed;
    {
        const uint64_t avg = units ? N / units : 0;
        p.claim_chunk = (uint32_t)std::min<uint64_t>(16, std::max<uint64_t>(1, 4096 / std::max<uint64_t>(1, avg)));
    }
    const uint64_t rows_per_block = 4ull * (uint64_t)waves;
    const uint32_t blocks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((units + rows_per_block - 1) / rows_per_block, (uint64_t)m->num_cus));
    unsigned long long* d_stamps = nullptr;
    const size_t n_stamp_waves = (size_t)blocks * (size_t)waves;
    if (const char* e = debug_on() ? getenv("TGX_STAMPS") : nullptr) {
        if (*e == '7' && pool_alloc(m->device, n_stamp_waves * 64, (void**)&d_stamps) == hipSuccess) {
            (void)hipMemsetAsync(d_stamps, 0, n_stamp_waves * 64, m->stream);
            p.stamps = d_stamps;
        }
    }
    {
        const char* f = debug_on() ? getenv("TGX_FLAGS") : nullptr;
        p.flags = f ? (uint32_t)atoi(f) : 0u;
    }
    hp.mark("pieces + buffers");
    time_begin(m, "estep7_kernel");
    if (tgx::launch_estep7(p, wide, ppl, waves, blocks, m->stream) != hipSuccess) {
        pool_free(m->device, d_stamps, n_stamp_waves * 64);
        return cleanup(fail(TGX_ERR_DEVICE, "estep7 launch failed"));
    }
    time_end(m);
    if (d_stamps) {  // diagnostic: mean ticks per trip and phase over all waves
        std::vector<unsigned long long> hs(n_stamp_waves * 8);
        const bool ok = hipStreamSynchronize(m->stream) == hipSuccess && hipMemcpy(hs.data(), d_stamps, n_stamp_waves * 64, hipMemcpyDeviceToHost) == hipSuccess;
        pool_free(m->device, d_stamps, n_stamp_waves * 64);
        if (ok) {
            double sum[6] = {0, 0, 0, 0, 0, 0}, trips = 0;
            for (size_t w = 0; w < n_stamp_waves; w++) {
                for (int i = 0; i < 6; i++) sum[i] += (double)hs[w * 8 + i];
                trips += (double)hs[w * 8 + 6];
            }
            fprintf(stderr, "[tgx] estep7 stamps (s_memtime ticks per wave-trip, %zu waves x ppl %d, %.0f trips): claim %.0f  walk %.0f  cut %.0f  forward %.0f  z %.0f  backward %.0f\n",
                    n_stamp_waves, ppl, trips, sum[0] / trips, sum[1] / trips, sum[2] / trips, sum[3] / trips, sum[4] / trips, sum[5] / trips);
        }
    }
    unsigned long long flag = 0, n_redo = 0;
    if (hipMemcpyAsync(&flag, m->d_ctrl + 5, 8, hipMemcpyDeviceToHost, m->stream) != hipSuccess ||
        hipMemcpyAsync(&n_redo, m->d_ctrl + 4, 8, hipMemcpyDeviceToHost, m->stream) != hipSuccess ||
        hipStreamSynchronize(m->stream) != hipSuccess)
        return cleanup(fail(TGX_ERR_DEVICE, "E-step pass failed: %s", hipGetErrorString(hipGetLastError())));
    m->last_estep_redo = n_redo;
    if (flag != 0) {  // a position nothing reaches, a value out of range: the pass belongs to the log-domain kernels
        *fallback = true;
        return cleanup(TGX_OK);
    }
    if (n_redo != 0) {
        // the stretches the kernel could not close within a trip: estep7_redo_kernel, trips spilled to scratch
        if (n_redo > redo_cap) return cleanup(fail(TGX_ERR_DEVICE, "E-step redo list longer than its buffers"));
        std::vector<uint64_t> ro(2 * n_redo), tbase(n_redo + 1, 0);
        if (hipMemcpy(ro.data(), d_roffs, 2 * n_redo * 8, hipMemcpyDeviceToHost) != hipSuccess) return cleanup(fail(TGX_ERR_DEVICE, "redo list read-back failed"));
        for (uint64_t i = 0; i < n_redo; i++) {
            if (ro[2 * i + 1] < ro[2 * i] || ro[2 * i + 1] > N) return cleanup(fail(TGX_ERR_DEVICE, "corrupt E-step redo list"));
            tbase[i + 1] = tbase[i] + (ro[2 * i + 1] - ro[2 * i]) / 16 + 1;
        }
        const uint64_t TT = tbase[n_redo];
        const size_t rs = wide_redo ? 1024 : 512;
        const size_t ab = (size_t)TT * 16 * 8 + 256, xb = (size_t)TT * 4 + 256, mb = (size_t)TT * rs + 256, tbb = (size_t)(n_redo + 1) * 8 + 256;
        double* d_alpha = nullptr;
        int32_t* d_aexp = nullptr;
        unsigned char* d_ms = nullptr;
        uint64_t* d_tbase = nullptr;
        auto cleanup2 = [&](tgx_status s2) {
            (void)hipStreamSynchronize(m->stream);
            pool_free(m->device, d_alpha, ab);
            pool_free(m->device, d_aexp, xb);
            pool_free(m->device, d_ms, mb);
            pool_free(m->device, d_tbase, tbb);
            return s2;
        };
        if (pool_alloc(m->device, ab, (void**)&d_alpha) != hipSuccess || pool_alloc(m->device, xb, (void**)&d_aexp) != hipSuccess ||
            pool_alloc(m->device, mb, (void**)&d_ms) != hipSuccess || pool_alloc(m->device, tbb, (void**)&d_tbase) != hipSuccess)
            return cleanup(cleanup2(fail(TGX_ERR_DEVICE, "out of device memory (E-step redo scratch)")));
        if (hipMemcpyAsync(d_tbase, tbase.data(), (n_redo + 1) * 8, hipMemcpyHostToDevice, m->stream) != hipSuccess ||
            hipMemsetAsync(m->d_ctrl + 3, 0x00, 8, m->stream) != hipSuccess)
            return cleanup(cleanup2(fail(TGX_ERR_DEVICE, "E-step redo setup failed")));
        tgx::Estep7RedoParams q{};
        q.text = c->d_text;
        q.redo_offs = d_roffs;
        q.redo_sample = d_rsample;
        q.redo_base = d_rbase;
        q.redo_snip = d_rsnip;
        q.n_redo = n_redo;
        q.tbase = d_tbase;
        q.trie8t = m->d_trie8t;
        q.n_slots = p.n_slots;
        q.root_base = p.root_base;
        q.wtab = m->d_wtab;
        q.n_tok = m->n_tok7;
        q.n_hot = std::min(std::min(m->n_tok7, tgx::estep7_redo_max_hot(wide_redo)), n_hot);
        q.expected = d_exp;
        q.zsnip = d_zsnip;
        q.logz_sum = d_z;
        q.range_flag = m->d_ctrl + 5;
        q.queue = m->d_ctrl + 3;
        q.alpha = d_alpha;
        q.aexp = d_aexp;
        q.mscratch = d_ms;
        q.dropout = dropout;
        q.seed = seed;
        time_begin(m, "estep7_redo_kernel");
        const hipError_t le = tgx::launch_estep7_redo(q, wide_redo, (uint32_t)m->num_cus, m->stream);
        time_end(m);
        if (le != hipSuccess) return cleanup(cleanup2(fail(TGX_ERR_DEVICE, "estep7 redo launch failed: %s", hipGetErrorString(le))));
        if (hipMemcpyAsync(&flag, m->d_ctrl + 5, 8, hipMemcpyDeviceToHost, m->stream) != hipSuccess)
            return cleanup(cleanup2(fail(TGX_ERR_DEVICE, "E-step redo pass failed")));
        const tgx_status rst = cleanup2(TGX_OK);  // (synchronises the stream)
        if (rst != TGX_OK) return cleanup(rst);
        if (hipGetLastError() != hipSuccess) return cleanup(fail(TGX_ERR_DEVICE, "E-step redo pass failed"));
        if (flag != 0) {
            *fallback = true;
            return cleanup(TGX_OK);
        }
    }
    hp.mark("kernel + redo");
    if (tgx::launch_snip_z_check(d_zsnip, K, m->d_ctrl + 1, m->stream) != hipSuccess) return cleanup(fail(TGX_ERR_DEVICE, "z check launch failed"));
    std::vector<double> h((size_t)m->n_tok7 + 1);
    double hz = 0.0;
    if (hipMemcpyAsync(&m->h_ctrl[0], m->d_ctrl + 1, 8, hipMemcpyDeviceToHost, m->stream) != hipSuccess ||
        hipMemcpyAsync(h.data(), d_exp, h.size() * 8, hipMemcpyDeviceToHost, m->stream) != hipSuccess ||
        hipMemcpyAsync(&hz, d_z, 8, hipMemcpyDeviceToHost, m->stream) != hipSuccess || hipStreamSynchronize(m->stream) != hipSuccess)
        return cleanup(fail(TGX_ERR_DEVICE, "E-step pass failed: %s", hipGetErrorString(hipGetLastError())));
    m->last_alg_bytes = N + 8 * (S + 1) + 8ull * m->vocab_size;
    const unsigned long long bad = m->h_ctrl[0];
    if (bad != ~0ULL) {  // nothing of a failed pass reaches the caller's `expected`
        const uint64_t smp = es.ssample[bad];
        g_err_sample = smp;
        g_err_pos = g_err_len = c->h_offs[smp + 1] - c->h_offs[smp];
        return cleanup(fail(TGX_ERR_Z_NOT_NORMAL, "normalization constant is not a normal number (sample %llu, len=%llu)",
                            (unsigned long long)smp, (unsigned long long)g_err_len));  // src/prune.rs:90-96
    }
    hp.mark("download");
    for (uint32_t r = 1; r <= m->n_tok7; r++) expected[m->id_of_rank[r]] += h[r];
    hp.mark("rank -> id");
    if (logz_sum) *logz_sum = hz;
    m->estep_calls++;
    return cleanup(TGX_OK);
}

// E-step on the four-snippets-per-wave kernels (estep4.hip).  Caller holds m->mu and has
// built the reversed trie.
// `fallback` (vocabularies with tokens of 17..32 bytes only): set when the linear-domain kernels cannot do the
// pass (a position without an incoming token, the f64 range left, an overflow list full) — there are no log-domain
// rows4 kernels for such vocabularies, the caller goes on to the generic kernel; `expected` is untouched then.
static tgx_status estep_rows4(tgx_model* m, tgx_corpus* c, uint64_t snippet_len, double dropout,
                              uint64_t seed, double* expected, double* logz_sum, bool* fallback) {
    const bool long_tokens = m->lm > 16;
    if (fallback) *fallback = false;
    const uint64_t S = c->n_samples, N = c->n_bytes;
    {
        const tgx_status wst = ensure_estep_work(m, c, snippet_len);
        if (wst != TGX_OK) return wst;
    }
    tgx_corpus::EstepWork& es = c->es;
    const std::vector<uint64_t>& soffs = es.soffs;
    const std::vector<uint32_t>& ssample = es.ssample;
    const std::vector<uint32_t>& order = es.order;
    const uint64_t K = soffs.size() - 1;
    uint64_t* const d_soffs = es.d_soffs;
    uint64_t* const d_sbase = es.d_sbase;
    uint32_t* const d_order = es.d_order;
    uint32_t* const d_ssample = es.d_ssample;
    const size_t n_rev = m->flat_rev.table.size();
    // TGX_ESTEP=log keeps the log-domain kernels (A/B timing, tests of both)
    const char* force_log = knob("TGX_ESTEP");
    const bool linear = m->estep_linear_ok && !(force_log && strcmp(force_log, "log") == 0);

    // ---- pieces (cuts.hip): a pass bound by the serial chains of its longest snippets cuts them where no token match
    // crosses — the lattice factorises there — and runs the linear-domain kernels on pieces of about `window` bytes.
    // Estimates as below (1 position per lane: forward 52 GB/s and 14.5 ms per 64 KiB of chain, backward 21 GB/s and
    // 29 ms); TGX_ESTEP_PIECES=0 / 1 forces, TGX_ESTEP_WINDOW sets the window.
    EstepPieces pc;
    auto free_pieces = [&]() { estep_pieces_free(m, pc); };
    bool pieces = false;
    if (linear && K && m->lm <= 32) {
        uint32_t window = 2048;
        if (const char* e = knob("TGX_ESTEP_WINDOW")) window = (uint32_t)std::min(1 << 20, std::max(256, atoi(e)));
        const double longest0 = (double)(soffs[order[0] + 1] - soffs[order[0]]);
        const double t_chain = longest0 / 65536.0 * (9.0 + 20.9) * 1e-3;     // the best chains of either kernel
        const double t_thru = (double)N / 52e9 + (double)N / 21e9;
        // (measured, 32 000 entries, samples <= 64 KiB: 64 MiB 31.3 -> 4.1 ms, 256 MiB 33.0 -> 14.4, 512 MiB 44.8 -> 28.3,
        // 1 GiB 57.1 -> 57.2: the longest chains delay a pass well before they bound it — profiles/r03)
        pieces = longest0 > 4.0 * window && t_chain > 0.5 * t_thru;
        if (const char* e = knob("TGX_ESTEP_PIECES")) pieces = atoi(e) != 0 && longest0 > (double)window;
        if (pieces) {
            const tgx_status wst = ensure_estep_windows(m, c, window);
            if (wst != TGX_OK) return wst;
        }
    }
    if (pieces) {
        const tgx_status pst = estep_pieces_build(m, c, m->d_trie_w, dropout, seed, &pc);
        if (pst != TGX_OK) return pst;
        m->last_estep_pieces = pc.n;
    } else {
        m->last_estep_pieces = 0;
    }
    const uint64_t Kmax = pieces ? std::max<uint64_t>(K, pc.n) : K;
    // replicas of the expected-count array (see estep4_bwd_kernel): up to 256, within 512 MiB
    // (round 3: 2.  The 256 of round 1 kept a handful of very frequent tokens from serialising every wave's atomics; those
    // are summed in the blocks' LDS since round 2, and the remaining memory-side adds are faster on a small array:
    // backward kernel 38.3 ms per GiB with 256 replicas, 36.9 with 4, 36.2 with 2, 35.9 with 1 — profiles/r03/r_*)
    uint32_t n_rep = 2;
    if (const char* e = knob("TGX_ESTEP_REPLICAS")) n_rep = (uint32_t)std::min(256, std::max(1, atoi(e)));
    while (n_rep > 1 && (size_t)n_rep * n_rev * 8 > (512ull << 20)) n_rep >>= 1;
    const size_t abytes = (size_t)(N + Kmax + 128) * 8, ebytes = (size_t)(n_rep + 1) * n_rev * 8 + 256, zbytes = (size_t)Kmax * 8 + 256;
    double *d_alpha = nullptr, *d_exp = nullptr, *d_z = nullptr, *d_zarr = nullptr;
    int32_t* d_aexp = nullptr;  // block exponents of alpha (linear-domain kernels)
    const size_t xbytes = (size_t)((N >> 4) + Kmax + 128) * 4;
    auto cleanup = [&](tgx_status s2) {
        // kernels already queued may still write these buffers: no other handle may take them from the pool yet
        if (s2 != TGX_OK) (void)hipStreamSynchronize(m->stream);
        free_pieces();
        pool_free(m->device, d_alpha, abytes);
        pool_free(m->device, d_aexp, xbytes);
        pool_free(m->device, d_exp, ebytes);
        pool_free(m->device, d_z, 256);
        pool_free(m->device, d_zarr, zbytes);
        return s2;
    };
    if (pool_alloc(m->device, abytes, (void**)&d_alpha) != hipSuccess ||
        (linear && pool_alloc(m->device, xbytes, (void**)&d_aexp) != hipSuccess) ||
        pool_alloc(m->device, ebytes, (void**)&d_exp) != hipSuccess ||
        pool_alloc(m->device, 256, (void**)&d_z) != hipSuccess ||
        pool_alloc(m->device, zbytes, (void**)&d_zarr) != hipSuccess)
        return cleanup(fail(TGX_ERR_DEVICE, "out of device memory (E-step scratch)"));
    if (hipMemsetAsync(d_exp, 0, ebytes, m->stream) != hipSuccess ||
        hipMemsetAsync(d_z, 0, 256, m->stream) != hipSuccess ||
        hipMemsetAsync(m->d_ctrl + 1, 0xFF, 8, m->stream) != hipSuccess)
        return cleanup(fail(TGX_ERR_DEVICE, "E-step setup copies failed"));
    tgx::Estep4Params p{};
    p.text = c->d_text;
    p.soffs = d_soffs;
    p.order = d_order;
    p.n_snips = K;
    p.snip_sample = d_ssample;
    p.snip_base = d_sbase;
    p.trie_fwd = linear ? m->d_trie_w : m->d_trie;
    p.trie_rev = linear ? m->d_trie_rev_w : m->d_trie_rev;
    p.alpha_exp = d_aexp;
    p.root_fwd = m->flat.table[0].base & ~tgx::kTerminalBit;
    p.root_rev = m->flat_rev.table[0].base & ~tgx::kTerminalBit;
    p.alpha = d_alpha;
    p.zarr = d_zarr;
    p.expected_slot = d_exp;
    p.n_slots_rev = (uint32_t)n_rev;
    p.n_replicas = n_rep;
    p.logz_sum = d_z;
    p.err_snip = m->d_ctrl + 1;
    p.queue_fwd = m->d_ctrl + 3;
    p.queue_bwd = m->d_ctrl + 4;
    p.range_flag = m->d_ctrl + 5;
    {
        const char* f = debug_on() ? getenv("TGX_FLAGS") : nullptr;
        p.flags = f ? (uint32_t)atoi(f) : 0u;
    }
    p.dropout = dropout;
    p.seed = seed;
    // The linear-domain kernels are tried first; if some position of some snippet has no incoming token
    // (lattice.rs:255's 0.0 case, e.g. a byte that is no token) or a value left the f64 range, the forward
    // kernel says so and the pass is redone with the log-domain kernels.
    // Positions per lane of the linear-domain kernels, chosen per kernel like encode4_kernel's: the longest
    // snippet is a serial chain (forward 14.5 / 11 / 9 ms, backward 29 / 23 / 23.5 ms per 64 KiB at 1 / 2 / 4
    // positions per lane) while more positions per lane cost waves (forward 52 / 48 / 33 GB/s, backward
    // 21 / 16 / 11 GB/s): tools/eppl_sweep.py on 1 x MI355X.  TGX_EPPL overrides both.
    int eppl_fwd = 1, eppl_bwd = 1;
    {
        const double longest = pieces ? (double)pc.longest / 65536.0 : (K ? (double)(soffs[order[0] + 1] - soffs[order[0]]) / 65536.0 : 0.0);
        const double f_gbps[3] = {52.0, 48.0, 33.0}, f_chain[3] = {14.5, 11.0, 9.0};
        // (the backward kernel with 2 positions per lane runs 8 groups of 16 positions per block and sums 8192
        // slots in LDS: tools/bwd_groups_sweep.py)
        const double b_gbps[3] = {21.0, 12.7, 11.0}, b_chain[3] = {29.0, 20.9, 23.5};
        double bf = 0, bb = 0;
        for (int i = 0; i < 3; i++) {
            const double tf = std::max((double)N / (f_gbps[i] * 1e6), longest * f_chain[i]);
            const double tb = std::max((double)N / (b_gbps[i] * 1e6), longest * b_chain[i]);
            if (i == 0 || tf < bf * 0.95) { bf = tf; eppl_fwd = 1 << i; }
            if (i == 0 || tb < bb * 0.95) { bb = tb; eppl_bwd = 1 << i; }
        }
        if (const char* e = knob("TGX_EPPL")) {
            const int v = atoi(e);
            if (v == 1 || v == 2 || v == 4) eppl_fwd = eppl_bwd = v;
        }
    }
    // Forward sweep on the 8-byte ranked records (encode5.hip: estep5_fwd_kernel) where the vocabulary has them; the
    // geometry as for encode5_kernel: every w in LDS with four positions per lane on 16..13 waves, three on 13, two
    // on 16, else three on 16 with the hottest in LDS (COLD build).  TGX_ESTEP_FWD=rows4 keeps estep4l_fwd_kernel.
    bool use5f = false, cold5f = false;
    int ppl5f = 4, waves5f = 16;
    tgx::Encode5Params q5f{};
    uint32_t blocks5f = 1;
    if (linear && !long_tokens && c->max_len < (1ull << 32)) {
        const char* ff = knob("TGX_ESTEP_FWD");
        const char* fp = knob("TGX_PATH");  // (rows4 names the kernels over the 16-byte records, here as for encode)
        if (!(ff && strcmp(ff, "rows4") == 0) && !(fp && strcmp(fp, "rows4") == 0)) {
            // The tables cost ~80 ns of host time per token of the vocabulary, the kernel saves ~4.7 ms per GiB of text:
            // a model that runs ONE pass over a small shard (prune makes a model per EM sub-iteration, src/prune.rs:48)
            // is better off without them; a model's second pass, or a pass over enough text, builds them.
            if (!m->estep_trie8_tried && (m->estep_calls >= 1 || ff != nullptr || (double)N * 4.7e-12 >= (double)m->vocab_size * 80e-9)) {
                const tgx_status est = ensure_estep_trie8(m);
                if (est != TGX_OK) return cleanup(est);
            }
            use5f = m->have_wvalues;
        }
    }
    if (use5f) {
        const uint32_t budget = 160u * 1024u;
        auto fits = [&](int waves, int ppl) { return m->n_values <= tgx::encode5_max_hot(false, waves, ppl, budget); };
        if (fits(13, 4)) {
            ppl5f = 4;
            waves5f = 16;
            while (!fits(waves5f, 4)) waves5f--;
        } else if (fits(13, 3)) {
            ppl5f = 3;
            waves5f = 13;
        } else if (fits(16, 2)) {
            ppl5f = 2;
            waves5f = 16;
        } else {
            ppl5f = 3;
            waves5f = 16;
        }
        if (const char* e = knob("TGX_EPPL")) {
            const int v = atoi(e);
            if (v >= 1 && v <= 4) ppl5f = v;
        }
        uint32_t n_hot = std::min(m->n_values, tgx::encode5_max_hot(false, waves5f, ppl5f, budget));
        if (const char* e = knob("TGX_E5_HOT")) {
            const int v = atoi(e);
            if (v >= 0) n_hot = std::min(n_hot, (uint32_t)v);
        }
        cold5f = n_hot < m->n_values;
        int per_simd = 0;
        if (tgx::estep5_waves_per_simd(dropout > 0.0, cold5f, ppl5f, &per_simd) != hipSuccess) return cleanup(fail(TGX_ERR_DEVICE, "estep5 attribute query failed"));
        waves5f = std::max(1, std::min(waves5f, per_simd * 4));
        const uint64_t n_units = pieces ? pc.n : K;
        {
            const uint64_t rows_wanted = (n_units + (uint64_t)m->num_cus - 1) / (uint64_t)m->num_cus;
            waves5f = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)waves5f, (rows_wanted + 3) / 4));
        }
        n_hot = std::min(m->n_values, tgx::encode5_max_hot(false, waves5f, ppl5f, budget));  // (fewer waves: more room)
        if (const char* e = knob("TGX_E5_HOT")) {
            const int v = atoi(e);
            if (v >= 0) n_hot = std::min(n_hot, (uint32_t)v);
        }
        cold5f = n_hot < m->n_values;
        q5f.trie8 = m->d_trie8;
        q5f.trie_bytes = (uint32_t)(m->flat.table.size() * sizeof(tgx::Trie8Rec));
        q5f.values = m->d_wvalues;
        q5f.root_base = m->root_base8;
        q5f.n_values = m->n_values;
        q5f.n_hot = n_hot;
        const uint64_t avg = n_units ? N / n_units : 0;
        q5f.claim_chunk = (uint32_t)std::min<uint64_t>(16, std::max<uint64_t>(1, 4096 / std::max<uint64_t>(1, avg)));
        const uint64_t rows_per_block = 4ull * (uint64_t)waves5f;
        blocks5f = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((n_units + rows_per_block - 1) / rows_per_block, (uint64_t)m->num_cus));
    }
    bool use_linear = linear;
    for (;;) {
        p.trie_fwd = use_linear ? m->d_trie_w : m->d_trie;
        p.trie_rev = use_linear ? m->d_trie_rev_w : m->d_trie_rev;
        const bool on_pieces = pieces && use_linear;  // (the log-domain kernels redo a pass on the uncut snippets: cuts.hip)
        p.soffs = on_pieces ? pc.d_offs : d_soffs;
        p.order = on_pieces ? pc.d_order : d_order;
        p.n_snips = on_pieces ? pc.n : K;
        p.snip_sample = on_pieces ? pc.d_sample : d_ssample;
        p.snip_base = on_pieces ? pc.d_base : d_sbase;
        p.err_snip = on_pieces ? m->d_ctrl + 7 : m->d_ctrl + 1;  // pieces: z is checked per snippet (launch_piece_z_check)
        if (hipMemsetAsync(m->d_ctrl + 3, 0x00, 24, m->stream) != hipSuccess ||
            hipMemsetAsync(d_z, 0, 256, m->stream) != hipSuccess ||
            hipMemsetAsync(m->d_ctrl + 1, 0xFF, 8, m->stream) != hipSuccess)
            return cleanup(fail(TGX_ERR_DEVICE, "E-step queue reset failed"));
        const bool fwd5 = use_linear && use5f;
        time_begin(m, fwd5 ? "estep5_fwd_kernel" : (use_linear ? "estep4l_fwd_kernel" : "estep4_fwd_kernel"));
        if ((fwd5 ? tgx::launch_estep5_fwd(p, q5f, cold5f, ppl5f, waves5f, blocks5f, m->stream)
                  : (use_linear ? tgx::launch_estep4l_fwd(p, eppl_fwd, long_tokens, (uint32_t)m->num_cus, m->stream)
                                : tgx::launch_estep4_fwd(p, (uint32_t)m->num_cus, m->stream))) != hipSuccess)
            return cleanup(fail(TGX_ERR_DEVICE, "E-step forward launch failed"));
        time_end(m);
        if (use_linear) {
            unsigned long long flag = 0;
            if (hipMemcpyAsync(&flag, m->d_ctrl + 5, 8, hipMemcpyDeviceToHost, m->stream) != hipSuccess ||
                hipStreamSynchronize(m->stream) != hipSuccess)
                return cleanup(fail(TGX_ERR_DEVICE, "E-step forward pass failed: %s", hipGetErrorString(hipGetLastError())));
            if (flag != 0) {
                if (long_tokens) {
                    if (fallback) *fallback = true;
                    return cleanup(TGX_OK);
                }
                use_linear = false;
                continue;
            }
        }
        time_begin(m, use_linear ? "estep4l_bwd_kernel" : "estep4_bwd_kernel");
        uint32_t bwd_groups = 16;  // groups of 16 positions per block of the backward kernel (estep4l.hip)
        if (const char* e = knob("TGX_BWD_GROUPS")) bwd_groups = (uint32_t)std::max(0, atoi(e));
        if ((use_linear ? tgx::launch_estep4l_bwd(p, eppl_bwd, long_tokens, (uint32_t)m->num_cus, bwd_groups, m->stream)
                        : tgx::launch_estep4_bwd(p, (uint32_t)m->num_cus, m->stream)) != hipSuccess)
            return cleanup(fail(TGX_ERR_DEVICE, "estep4 backward launch failed"));
        time_end(m);
        if (on_pieces && tgx::launch_piece_z_check(d_zarr, pc.d_snip, pc.n, pc.d_zsnip, K, m->d_ctrl + 1, m->stream) != hipSuccess)
            return cleanup(fail(TGX_ERR_DEVICE, "E-step z check launch failed"));
        break;
    }
    double* d_sum = d_exp + (size_t)n_rep * n_rev;  // replica sums
    time_begin(m, "estep4_reduce_kernel");
    if (tgx::launch_estep4_reduce(d_exp, d_sum, (uint32_t)n_rev, n_rep, m->stream) != hipSuccess)
        return cleanup(fail(TGX_ERR_DEVICE, "estep4 reduce launch failed"));
    time_end(m);
    std::vector<double> h(n_rev);
    double hz = 0.0;
    // The backward kernel of the long-token builds has an overflow list of its own (matches of 17..32 bytes by END
    // position and window; the forward kernel counts them by START position), so it can run out where the forward
    // kernel did not: the flag is read again with the results, and a pass that raised it is discarded.
    unsigned long long flag_bwd = 0;
    if (hipMemcpyAsync(&m->h_ctrl[0], m->d_ctrl + 1, 8, hipMemcpyDeviceToHost, m->stream) != hipSuccess ||
        hipMemcpyAsync(&flag_bwd, m->d_ctrl + 5, 8, hipMemcpyDeviceToHost, m->stream) != hipSuccess ||
        hipMemcpyAsync(h.data(), d_sum, n_rev * 8, hipMemcpyDeviceToHost, m->stream) != hipSuccess ||
        hipMemcpyAsync(&hz, d_z, 8, hipMemcpyDeviceToHost, m->stream) != hipSuccess ||
        hipStreamSynchronize(m->stream) != hipSuccess)
        return cleanup(fail(TGX_ERR_DEVICE, "E-step pass failed: %s", hipGetErrorString(hipGetLastError())));
    if (use_linear && flag_bwd != 0) {  // nothing of this pass reaches `expected`
        if (long_tokens && fallback) {
            *fallback = true;  // the generic kernel redoes it (tgx_estep)
            return cleanup(TGX_OK);
        }
        return cleanup(fail(TGX_ERR_DEVICE, "E-step backward kernel raised range_flag %llu after a clean forward pass", flag_bwd));
    }
    m->last_alg_bytes = N + 8 * (S + 1) + 8ull * m->vocab_size;
    const unsigned long long bad = m->h_ctrl[0];
    if (bad != ~0ULL) {  // nothing of a failed pass reaches the caller's `expected`
        const uint64_t smp = ssample[bad];
        g_err_sample = smp;
        g_err_pos = g_err_len = c->h_offs[smp + 1] - c->h_offs[smp];
        return cleanup(fail(TGX_ERR_Z_NOT_NORMAL, "normalization constant is not a normal number (sample %llu, len=%llu)",
                            (unsigned long long)smp, (unsigned long long)g_err_len));  // src/prune.rs:90-96
    }
    for (size_t t = 0; t < n_rev; t++) {
        const uint32_t id = m->flat_rev.tokid[t];
        if (id != tgx::kNoToken) expected[id] += h[t];
    }
    if (logz_sum) *logz_sum = hz;
    m->estep_calls++;
    return cleanup(TGX_OK);
}

tgx_status tgx_estep(tgx_model* m, tgx_corpus* c, uint64_t snippet_len, double dropout,
                     uint64_t seed, double* expected, double* logz_sum) {
    if (!m || !c || !expected) return fail(TGX_ERR_INVALID, "tgx_estep: NULL argument");
    if (m->device != c->device) return fail(TGX_ERR_INVALID, "model and corpus on different devices");
    if (snippet_len == 0) snippet_len = TGX_ESTEP_SNIPPET_LEN;
    if (snippet_len >= 0xFFFFFF00ull) return fail(TGX_ERR_UNSUPPORTED, "snippet_len must be below 4 GiB");
    std::lock_guard<std::mutex> lk(m->mu);
    std::lock_guard<std::mutex> lkc(c->mu);
    HIP_TRY(hipSetDevice(m->device));
    m->n_timed = 0;
    tgx_status st = TGX_OK;
    {   // round 4: one walk per position (estep7.hip); TGX_ESTEP=chain / log and TGX_PATH=rows4 / fused keep the older kernels
        const char* force = knob("TGX_PATH");
        const char* fe = knob("TGX_ESTEP");
        if (m->scores_finite && m->lm <= 16 && m->vocab_size && !force && !fe) {
            bool fallback = false;
            st = estep_fused(m, c, snippet_len, dropout, seed, expected, logz_sum, &fallback);
            if (st != TGX_OK || !fallback) return st;
            m->n_timed = 0;
        }
    }
    st = ensure_reverse_trie(m);
    if (st != TGX_OK) return st;
    const uint64_t S = c->n_samples, N = c->n_bytes;
    {
        const char* force = knob("TGX_PATH");
        const bool rows = m->scores_finite && !(force && strcmp(force, "fused") == 0);
        if (rows && m->lm <= 16) return estep_rows4(m, c, snippet_len, dropout, seed, expected, logz_sum, nullptr);
        // tokens of 17..32 bytes (after `merge`): the linear-domain kernels' long-token builds; the generic kernel
        // below where they cannot do the pass
        const char* force_log = knob("TGX_ESTEP");
        if (rows && m->lm <= 32 && m->estep_linear_ok && !(force_log && strcmp(force_log, "log") == 0)) {
            bool fallback = false;
            st = estep_rows4(m, c, snippet_len, dropout, seed, expected, logz_sum, &fallback);
            if (st != TGX_OK || !fallback) return st;
        }
    }
    const size_t n_rev = m->flat_rev.table.size();
    uint32_t n_rep = 256;
    while (n_rep > 1 && (size_t)n_rep * n_rev * 8 > (512ull << 20)) n_rep >>= 1;
    const size_t abytes = (size_t)(N + S + 128) * 8, ebytes = (size_t)(n_rep + 1) * n_rev * 8 + 256;
    double *d_alpha = nullptr, *d_exp = nullptr, *d_z = nullptr;
    auto cleanup = [&](tgx_status s2) {
        // kernels already queued may still write these buffers: no other handle may take them from the pool yet
        if (s2 != TGX_OK) (void)hipStreamSynchronize(m->stream);
        pool_free(m->device, d_alpha, abytes);
        pool_free(m->device, d_exp, ebytes);
        pool_free(m->device, d_z, 256);
        return s2;
    };
    if (pool_alloc(m->device, abytes, (void**)&d_alpha) != hipSuccess ||
        pool_alloc(m->device, ebytes, (void**)&d_exp) != hipSuccess ||
        pool_alloc(m->device, 256, (void**)&d_z) != hipSuccess)
        return cleanup(fail(TGX_ERR_DEVICE, "out of device memory (E-step scratch)"));
    if (hipMemsetAsync(d_exp, 0, ebytes, m->stream) != hipSuccess ||
        hipMemsetAsync(d_z, 0, 256, m->stream) != hipSuccess ||
        hipMemsetAsync(m->d_ctrl + 1, 0xFF, 8, m->stream) != hipSuccess)
        return cleanup(fail(TGX_ERR_DEVICE, "memset failed"));
    tgx::EstepParams p{};
    p.text = c->d_text;
    p.offs = c->d_offs;
    p.order = c->d_order;
    p.n_samples = S;
    p.trie_fwd = m->d_trie;
    p.trie_rev = m->d_trie_rev;
    p.root_fwd = m->flat.table[0].base & ~tgx::kTerminalBit;
    p.root_rev = m->flat_rev.table[0].base & ~tgx::kTerminalBit;
    p.lm = m->lm;
    p.snippet_len = snippet_len;
    p.alpha = d_alpha;
    p.expected_slot = d_exp;
    p.n_slots_rev = (uint32_t)n_rev;
    p.n_replicas = n_rep;
    p.logz_sum = d_z;
    p.err_sample = m->d_ctrl + 1;
    p.dropout = dropout;
    p.seed = seed;
    const uint64_t wpb = tgx::estep_waves_per_block(m->lm);
    const uint32_t blocks = (uint32_t)std::max<uint64_t>(
        1, std::min<uint64_t>((S + wpb - 1) / wpb, (uint64_t)m->num_cus * (uint64_t)m->estep_blocks_per_cu));
    time_begin(m, "estep_kernel");
    if (tgx::launch_estep(p, blocks, m->stream) != hipSuccess)
        return cleanup(fail(TGX_ERR_DEVICE, "estep launch failed"));
    time_end(m);
    double* d_sum = d_exp + (size_t)n_rep * n_rev;
    if (tgx::launch_estep4_reduce(d_exp, d_sum, (uint32_t)n_rev, n_rep, m->stream) != hipSuccess)
        return cleanup(fail(TGX_ERR_DEVICE, "estep reduce launch failed"));
    std::vector<double> h(n_rev);
    double hz = 0.0;
    if (hipMemcpyAsync(&m->h_ctrl[0], m->d_ctrl + 1, 8, hipMemcpyDeviceToHost, m->stream) != hipSuccess ||
        hipMemcpyAsync(h.data(), d_sum, n_rev * 8, hipMemcpyDeviceToHost, m->stream) != hipSuccess ||
        hipMemcpyAsync(&hz, d_z, 8, hipMemcpyDeviceToHost, m->stream) != hipSuccess ||
        hipStreamSynchronize(m->stream) != hipSuccess)
        return cleanup(fail(TGX_ERR_DEVICE, "E-step pass failed: %s", hipGetErrorString(hipGetLastError())));
    for (size_t t = 0; t < n_rev; t++) {
        const uint32_t id = m->flat_rev.tokid[t];
        if (id != tgx::kNoToken) expected[id] += h[t];
    }
    if (logz_sum) *logz_sum = hz;
    // N + 8(S+1) + 8V
    m->last_alg_bytes = N + 8 * (S + 1) + 8ull * m->vocab_size;
    const unsigned long long bad = m->h_ctrl[0];
    if (bad != ~0ULL) {
        g_err_sample = bad;
        g_err_pos = g_err_len = c->h_offs[bad + 1] - c->h_offs[bad];
        // the reference panics here: src/prune.rs:90-96
        return cleanup(fail(TGX_ERR_Z_NOT_NORMAL, "normalization constant is not a normal number (sample %llu, len=%llu)",
                            bad, (unsigned long long)g_err_len));
    }
    return cleanup(TGX_OK);
}

// ---- measurement ---------------------------------------------------------------

int tgx_last_kernel_times(const tgx_model* m, const char** names, float* ms, int cap) {
    if (!m) return 0;
    int n = 0;
    for (int i = 0; i < m->n_timed && n < cap; i++) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, m->timed[i].start, m->timed[i].stop) != hipSuccess) t = -1.f;
        if (names) names[n] = m->timed[i].name;
        if (ms) ms[n] = t;
        n++;
    }
    return n;
}

uint32_t tgx_last_encode_waves_per_cu(const tgx_model* m) { return m ? (uint32_t)m->last_encode_waves_per_cu : 0u; }
uint64_t tgx_last_algorithmic_bytes(const tgx_model* m) { return m ? m->last_alg_bytes : 0; }
uint64_t tgx_last_encode_redo_samples(const tgx_model* m) { return m ? m->last_redo_samples : 0; }
uint64_t tgx_last_encode_long_samples(const tgx_model* m) { return m ? m->last_long_samples : 0; }
uint64_t tgx_last_estep_pieces(const tgx_model* m) { return m ? m->last_estep_pieces : 0; }
uint64_t tgx_last_estep_redo(const tgx_model* m) { return m ? m->last_estep_redo : 0; }
uint32_t tgx_last_encode_corun_cus(const tgx_model* m) { return m ? m->last_corun_cus : 0; }
uint32_t tgx_encode_corun_timeouts(const tgx_model* m) { return m ? m->corun_wait_timeouts : 0; }
uint32_t tgx_model_score_values(const tgx_model* m) { return m && m->have_trie8 ? m->n_values : 0u; }
uint32_t tgx_last_encode_hot_values(const tgx_model* m) { return m ? m->last_n_hot : 0u; }

}  // extern "C"
