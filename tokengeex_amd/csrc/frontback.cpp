// Front and back of the Tokenizer around the encode path, over packed buffers and on host threads
// (SURVEY.md section 8f rank 4): the special-token splitter (reference src/tokenizer.rs:299-347), the CRLF
// processor (src/processor.rs:46-54) and decode / decode_batch (src/tokenizer.rs:126-187 over
// src/model.rs:146-160, each run of base ids through String::from_utf8_lossy).  The reference does these per
// sample on rayon workers; here they are byte loops over the batch's flat buffers, so that the Python
// surface no longer spends its time in per-sample Python code.  No device is needed.
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/tgx.h"

tgx_status tgx_set_error(tgx_status st, const char* msg);  // tgx_api.cpp

namespace {

tgx_status ferr(tgx_status st, const char* fmt, ...) {
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return tgx_set_error(st, buf);
}

unsigned host_threads(uint64_t work_items) {
    unsigned n = std::thread::hardware_concurrency();
    if (const char* e = getenv("TGX_HOST_THREADS")) {
        const int v = atoi(e);
        if (v >= 1) n = (unsigned)v;
    }
    n = std::max(1u, std::min(n, 64u));
    return (unsigned)std::min<uint64_t>(n, std::max<uint64_t>(1, work_items / 64));
}

// runs fn(lo, hi) over [0, n) split into contiguous ranges, one per host thread
template <class F>
void parallel_ranges(uint64_t n, F fn) {
    const unsigned t = host_threads(n);
    if (t <= 1) {
        fn(0, n, 0u);
        return;
    }
    std::vector<std::thread> th;
    for (unsigned k = 0; k < t; k++) th.emplace_back([=]() { fn(n * k / t, n * (k + 1) / t, k); });
    for (auto& x : th) x.join();
}

struct Seg {
    uint64_t begin, end;
    int32_t special;
};

// String::from_utf8_lossy (what src/model.rs:156-159 applies to every run of base ids): every maximal
// ill-formed subpart becomes one U+FFFD (Unicode 15 section 3.9, "U+FFFD substitution of maximal subparts";
// core::str::lossy::Utf8Chunks).  Appends to out.
void append_utf8_lossy(const uint8_t* s, size_t n, std::vector<uint8_t>& out) {
    size_t i = 0;
    while (i < n) {
        const uint8_t b = s[i];
        if (b < 0x80) {
            out.push_back(b);
            i++;
            continue;
        }
        size_t need = 0;
        uint8_t lo = 0x80, hi = 0xBF;  // allowed range of the SECOND byte
        if (b >= 0xC2 && b <= 0xDF) need = 1;
        else if (b == 0xE0) { need = 2; lo = 0xA0; }
        else if (b >= 0xE1 && b <= 0xEC) need = 2;
        else if (b == 0xED) { need = 2; hi = 0x9F; }
        else if (b >= 0xEE && b <= 0xEF) need = 2;
        else if (b == 0xF0) { need = 3; lo = 0x90; }
        else if (b >= 0xF1 && b <= 0xF3) need = 3;
        else if (b == 0xF4) { need = 3; hi = 0x8F; }
        size_t got = 0;  // continuation bytes accepted
        if (need) {
            while (got < need && i + 1 + got < n) {
                const uint8_t c = s[i + 1 + got];
                const uint8_t l = got == 0 ? lo : 0x80, h = got == 0 ? hi : 0xBF;
                if (c < l || c > h) break;
                got++;
            }
        }
        if (need && got == need) {
            out.insert(out.end(), s + i, s + i + 1 + need);
            i += 1 + need;
        } else {  // the maximal subpart: the lead byte and the continuation bytes that were acceptable
            out.push_back(0xEF);
            out.push_back(0xBF);
            out.push_back(0xBD);
            i += 1 + got;
        }
    }
}

}  // namespace

extern "C" {

// SpecialTokenSplitter over a packed batch (src/tokenizer.rs:299-347): the earliest position wins, at one position
// the first special token in list order that the text starts with (not the longest); the text in front of it is a
// segment of its own.  Specials are valid UTF-8 and so is the text, so a match can only begin at a character
// boundary and bytes can be scanned instead of chars.  Segments come back in CSR form: sample i owns segments
// seg_offs[i] .. seg_offs[i+1]; segment k is text[seg_begin[k], seg_end[k]) and seg_special[k] is the index of
// the special token or -1.  The three segment arrays are malloc'd (tgx_free).
tgx_status tgx_split_specials(const uint8_t* text, const uint64_t* offs, uint64_t n_samples, const uint8_t* special_bytes,
                              const uint64_t* special_offs, uint32_t n_specials, uint64_t* seg_offs, uint64_t** seg_begin,
                              uint64_t** seg_end, int32_t** seg_special, uint64_t* n_segments) {
    if (!offs || !seg_offs || !seg_begin || !seg_end || !seg_special || !n_segments || (n_specials && !special_offs))
        return ferr(TGX_ERR_INVALID, "tgx_split_specials: NULL argument");
    *seg_begin = *seg_end = nullptr;
    *seg_special = nullptr;
    *n_segments = 0;
    for (uint32_t k = 0; k < n_specials; k++)
        if (special_offs[k + 1] == special_offs[k])
            return ferr(TGX_ERR_INVALID, "empty special token (the reference's splitter would never advance)");
    // candidates by first byte, in list order
    std::vector<std::vector<uint32_t>> by_first(256);
    for (uint32_t k = 0; k < n_specials; k++) by_first[special_bytes[special_offs[k]]].push_back(k);
    bool first_mask[256];
    for (int b = 0; b < 256; b++) first_mask[b] = !by_first[b].empty();
    const unsigned T = host_threads(n_samples);
    std::vector<std::vector<Seg>> parts(std::max(1u, T));
    std::vector<uint32_t> per_sample(n_samples, 0);
    parallel_ranges(n_samples, [&](uint64_t lo, uint64_t hi, unsigned tid) {
        std::vector<Seg>& out = parts[tid];
        for (uint64_t i = lo; i < hi; i++) {
            const uint64_t b = offs[i], e = offs[i + 1];
            uint64_t cursor = b;
            uint32_t cnt = 0;
            for (uint64_t p = b; p < e;) {
                int32_t hit = -1;
                uint64_t len = 0;
                if (first_mask[text[p]]) {
                    for (uint32_t k : by_first[text[p]]) {
                        const uint64_t l = special_offs[k + 1] - special_offs[k];
                        if (l <= e - p && memcmp(text + p, special_bytes + special_offs[k], (size_t)l) == 0) {
                            hit = (int32_t)k;
                            len = l;
                            break;
                        }
                    }
                }
                if (hit >= 0) {
                    if (p > cursor) {
                        out.push_back(Seg{cursor, p, -1});
                        cnt++;
                    }
                    out.push_back(Seg{p, p + len, hit});
                    cnt++;
                    p += len;
                    cursor = p;
                } else {
                    p++;
                }
            }
            if (cursor < e) {
                out.push_back(Seg{cursor, e, -1});
                cnt++;
            }
            per_sample[i] = cnt;
        }
    });
    uint64_t total = 0;
    seg_offs[0] = 0;
    for (uint64_t i = 0; i < n_samples; i++) {
        total += per_sample[i];
        seg_offs[i + 1] = total;
    }
    uint64_t* sb = (uint64_t*)malloc(sizeof(uint64_t) * std::max<uint64_t>(1, total));
    uint64_t* se = (uint64_t*)malloc(sizeof(uint64_t) * std::max<uint64_t>(1, total));
    int32_t* ss = (int32_t*)malloc(sizeof(int32_t) * std::max<uint64_t>(1, total));
    if (!sb || !se || !ss) {
        free(sb);
        free(se);
        free(ss);
        return ferr(TGX_ERR_INVALID, "tgx_split_specials: out of host memory");
    }
    uint64_t k = 0;
    for (const auto& part : parts)  // thread ranges are contiguous and ascending: concatenation is sample order
        for (const Seg& s : part) {
            sb[k] = s.begin;
            se[k] = s.end;
            ss[k] = s.special;
            k++;
        }
    *seg_begin = sb;
    *seg_end = se;
    *seg_special = ss;
    *n_segments = total;
    return TGX_OK;
}

// Packs segments text[seg_begin[k], seg_end[k]) back to back (the batch format of the encode entry points), with
// CrlfProcessor::preprocess applied on the way when crlf != 0 (src/processor.rs:46-54: plain
// replace("\r\n", "\n")).  Segments with seg_special[k] >= 0 are skipped when seg_special is given: the result
// holds the non-special segments only, in order.  out_text must hold as many bytes as the segments have;
// out_offs[m + 1] for m output segments; *n_out = m.
tgx_status tgx_pack_segments(const uint8_t* text, const uint64_t* seg_begin, const uint64_t* seg_end, const int32_t* seg_special,
                             uint64_t n, int crlf, uint8_t* out_text, uint64_t* out_offs, uint64_t* n_out) {
    if (!seg_begin || !seg_end || !out_offs || !n_out) return ferr(TGX_ERR_INVALID, "tgx_pack_segments: NULL argument");
    std::vector<uint64_t> keep;
    keep.reserve(n);
    for (uint64_t k = 0; k < n; k++)
        if (!seg_special || seg_special[k] < 0) keep.push_back(k);
    const uint64_t m = keep.size();
    std::vector<uint64_t> lens(m);
    parallel_ranges(m, [&](uint64_t lo, uint64_t hi, unsigned) {
        for (uint64_t i = lo; i < hi; i++) {
            const uint64_t b = seg_begin[keep[i]], e = seg_end[keep[i]];
            uint64_t drop = 0;
            if (crlf)
                for (uint64_t p = b; p + 1 < e; p++)
                    if (text[p] == '\r' && text[p + 1] == '\n') drop++;
            lens[i] = (e - b) - drop;
        }
    });
    out_offs[0] = 0;
    for (uint64_t i = 0; i < m; i++) out_offs[i + 1] = out_offs[i] + lens[i];
    parallel_ranges(m, [&](uint64_t lo, uint64_t hi, unsigned) {
        for (uint64_t i = lo; i < hi; i++) {
            const uint64_t b = seg_begin[keep[i]], e = seg_end[keep[i]];
            uint8_t* o = out_text + out_offs[i];
            if (!crlf) {
                if (e > b) memcpy(o, text + b, (size_t)(e - b));
                continue;
            }
            for (uint64_t p = b; p < e; p++) {
                if (text[p] == '\r' && p + 1 < e && text[p + 1] == '\n') continue;
                *o++ = text[p];
            }
        }
    });
    *n_out = m;
    return TGX_OK;
}

// Puts a batch's ids back together (src/tokenizer.rs:65-90): sample i is the concatenation, over its segments, of
// the special token's id (vocab_size + index, src/tokenizer.rs:72-76) or the ids of the next encoded segment
// (ids / id_offs: the encode result over the non-special segments in order).  out_ids must hold
// n_ids + (number of special segments) entries; out_offs[n_samples + 1].
tgx_status tgx_assemble_ids(const uint64_t* seg_offs, const int32_t* seg_special, uint64_t n_samples, const uint32_t* ids,
                            const uint64_t* id_offs, uint32_t vocab_size, uint32_t* out_ids, uint64_t* out_offs) {
    if (!seg_offs || !id_offs || !out_offs || (seg_offs[n_samples] && !seg_special))
        return ferr(TGX_ERR_INVALID, "tgx_assemble_ids: NULL argument");
    uint64_t enc = 0, at = 0;
    out_offs[0] = 0;
    for (uint64_t i = 0; i < n_samples; i++) {
        for (uint64_t k = seg_offs[i]; k < seg_offs[i + 1]; k++) {
            if (seg_special[k] >= 0) {
                out_ids[at++] = vocab_size + (uint32_t)seg_special[k];
            } else {
                const uint64_t b = id_offs[enc], e = id_offs[enc + 1];
                if (e > b) memcpy(out_ids + at, ids + b, (size_t)(e - b) * 4);
                at += e - b;
                enc++;
            }
        }
        out_offs[i + 1] = at;
    }
    return TGX_OK;
}

// Tokenizer::decode_batch over packed ids (src/tokenizer.rs:126-187): ids >= vocab_size are special tokens
// (id - vocab_size indexes special_*; emitted only with include_special), each run of base ids between them is
// byte-concatenated and passed through String::from_utf8_lossy on its own (src/model.rs:146-160).  Postprocessors
// are the identity in the reference (src/processor.rs:52-54, 134-136).  An id outside both ranges fails the
// batch with TGX_ERR_TOKEN_ID_OOB, "token id {id} is out of bounds" (src/lib.rs:246-248): the lowest failing
// sample is reported (tgx_last_error_detail: sample, id).  *out_text is malloc'd (tgx_free), out_offs[n + 1].
tgx_status tgx_decode_batch(const uint8_t* vocab_bytes, const uint64_t* vocab_offs, uint32_t vocab_size,
                            const uint8_t* special_bytes, const uint64_t* special_offs, uint32_t n_specials,
                            const uint32_t* ids, const uint64_t* id_offs, uint64_t n_samples, int include_special,
                            uint8_t** out_text, uint64_t* out_offs, uint64_t* bad_sample, uint64_t* bad_id) {
    if (!vocab_offs || !id_offs || !out_text || !out_offs || (n_specials && !special_offs))
        return ferr(TGX_ERR_INVALID, "tgx_decode_batch: NULL argument");
    *out_text = nullptr;
    const unsigned T = host_threads(n_samples);
    std::vector<std::vector<uint8_t>> parts(std::max(1u, T));
    std::vector<uint64_t> lens(n_samples, 0);
    std::vector<uint64_t> bad_s(std::max(1u, T), ~0ULL), bad_i(std::max(1u, T), 0);
    parallel_ranges(n_samples, [&](uint64_t lo, uint64_t hi, unsigned tid) {
        std::vector<uint8_t>& out = parts[tid];
        std::vector<uint8_t> run;
        for (uint64_t i = lo; i < hi; i++) {
            const size_t start = out.size();
            run.clear();
            for (uint64_t k = id_offs[i]; k < id_offs[i + 1]; k++) {
                const uint32_t id = ids[k];
                if (id < vocab_size) {
                    run.insert(run.end(), vocab_bytes + vocab_offs[id], vocab_bytes + vocab_offs[id + 1]);
                    continue;
                }
                append_utf8_lossy(run.data(), run.size(), out);
                run.clear();
                const uint64_t sp = (uint64_t)id - vocab_size;
                if (sp >= n_specials) {
                    if (bad_s[tid] == ~0ULL) {
                        bad_s[tid] = i;
                        bad_i[tid] = id;
                    }
                    continue;
                }
                if (include_special) out.insert(out.end(), special_bytes + special_offs[sp], special_bytes + special_offs[sp + 1]);
            }
            append_utf8_lossy(run.data(), run.size(), out);
            lens[i] = out.size() - start;
        }
    });
    for (unsigned k = 0; k < bad_s.size(); k++)
        if (bad_s[k] != ~0ULL) {  // thread ranges ascend: the first hit is the lowest sample
            if (bad_sample) *bad_sample = bad_s[k];
            if (bad_id) *bad_id = bad_i[k];
            return ferr(TGX_ERR_TOKEN_ID_OOB, "token id %llu is out of bounds", (unsigned long long)bad_i[k]);
        }
    out_offs[0] = 0;
    for (uint64_t i = 0; i < n_samples; i++) out_offs[i + 1] = out_offs[i] + lens[i];
    uint8_t* buf = (uint8_t*)malloc((size_t)std::max<uint64_t>(1, out_offs[n_samples]));
    if (!buf) return ferr(TGX_ERR_INVALID, "tgx_decode_batch: out of host memory");
    size_t at = 0;
    for (const auto& part : parts) {
        if (!part.empty()) memcpy(buf + at, part.data(), part.size());
        at += part.size();
    }
    *out_text = buf;
    return TGX_OK;
}

// String::from_utf8_lossy of one buffer (tests: against Python's bytes.decode("utf-8", "replace"), which follows
// the same substitution practice).  out must hold 3 n bytes; returns the output length.
uint64_t tgx_utf8_lossy(const uint8_t* s, uint64_t n, uint8_t* out) {
    std::vector<uint8_t> v;
    v.reserve((size_t)n);
    append_utf8_lossy(s, (size_t)n, v);
    if (!v.empty()) memcpy(out, v.data(), v.size());
    return v.size();
}

}  // extern "C"
