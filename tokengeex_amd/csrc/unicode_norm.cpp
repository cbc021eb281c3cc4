// UnicodeProcessor::preprocess over packed buffers (src/processor.rs:124-137: s.nfc() / nfd() / nfkc() / nfkd() of the
// unicode-normalization crate): the four UAX #15 normalisation forms in native code, a batch of segments at a time on the
// host's threads.  The data (combining classes, full decompositions, primary composites) is generated from Python's
// unicodedata by tools/make_unicode_tables.py; Hangul syllables are handled arithmetically.  Bytes that are not UTF-8
// (the flat surface accepts any bytes) pass through unchanged and act as starters.
//
// Checked against unicodedata.normalize on every code point and on random sequences with combining marks
// (tests/test_unicode_norm_cpu.py).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/tgx.h"
#include "unicode_tables.h"

tgx_status tgx_set_error(tgx_status st, const char* msg);  // tgx_api.cpp

namespace {

constexpr uint32_t S_BASE = 0xAC00, L_BASE = 0x1100, V_BASE = 0x1161, T_BASE = 0x11A7;
constexpr uint32_t L_COUNT = 19, V_COUNT = 21, T_COUNT = 28, N_COUNT = V_COUNT * T_COUNT, S_COUNT = L_COUNT * N_COUNT;
constexpr uint32_t kRawByte = 0x80000000u;  // an invalid byte b travels as kRawByte | b

inline int find_cp(const uint32_t* cps, uint32_t n, uint32_t cp) {
    const uint32_t* p = std::lower_bound(cps, cps + n, cp);
    return (p != cps + n && *p == cp) ? (int)(p - cps) : -1;
}
inline uint32_t ccc_of(uint32_t cp) {
    if (cp < 0x300 || (cp & kRawByte)) return 0;
    const int i = find_cp(kCccCp, kCccN, cp);
    return i < 0 ? 0u : kCccVal[i];
}

// UTF-8 of one scalar value appended to out
inline void put_utf8(std::vector<uint8_t>& out, uint32_t cp) {
    if (cp & kRawByte) {
        out.push_back((uint8_t)cp);
    } else if (cp < 0x80) {
        out.push_back((uint8_t)cp);
    } else if (cp < 0x800) {
        out.push_back((uint8_t)(0xC0 | (cp >> 6)));
        out.push_back((uint8_t)(0x80 | (cp & 63)));
    } else if (cp < 0x10000) {
        out.push_back((uint8_t)(0xE0 | (cp >> 12)));
        out.push_back((uint8_t)(0x80 | ((cp >> 6) & 63)));
        out.push_back((uint8_t)(0x80 | (cp & 63)));
    } else {
        out.push_back((uint8_t)(0xF0 | (cp >> 18)));
        out.push_back((uint8_t)(0x80 | ((cp >> 12) & 63)));
        out.push_back((uint8_t)(0x80 | ((cp >> 6) & 63)));
        out.push_back((uint8_t)(0x80 | (cp & 63)));
    }
}

// next scalar value of s[i, n) (strict UTF-8: no overlong forms, no surrogates, <= 0x10FFFF); an offending byte is
// returned as kRawByte | byte and consumed alone
inline uint32_t next_cp(const uint8_t* s, uint64_t n, uint64_t& i) {
    const uint8_t b0 = s[i];
    if (b0 < 0x80) {
        i++;
        return b0;
    }
    auto cont = [&](uint64_t k) { return k < n && (s[k] & 0xC0) == 0x80; };
    if (b0 >= 0xC2 && b0 <= 0xDF && cont(i + 1)) {
        const uint32_t cp = ((uint32_t)(b0 & 0x1F) << 6) | (s[i + 1] & 63u);
        i += 2;
        return cp;
    }
    if (b0 >= 0xE0 && b0 <= 0xEF && cont(i + 1) && cont(i + 2)) {
        const uint32_t cp = ((uint32_t)(b0 & 0x0F) << 12) | ((uint32_t)(s[i + 1] & 63u) << 6) | (s[i + 2] & 63u);
        if (cp >= 0x800 && !(cp >= 0xD800 && cp <= 0xDFFF)) {
            i += 3;
            return cp;
        }
    }
    if (b0 >= 0xF0 && b0 <= 0xF4 && cont(i + 1) && cont(i + 2) && cont(i + 3)) {
        const uint32_t cp = ((uint32_t)(b0 & 0x07) << 18) | ((uint32_t)(s[i + 1] & 63u) << 12) | ((uint32_t)(s[i + 2] & 63u) << 6) | (s[i + 3] & 63u);
        if (cp >= 0x10000 && cp <= 0x10FFFF) {
            i += 4;
            return cp;
        }
    }
    i++;
    return kRawByte | b0;
}

// full decomposition of cp appended to buf
inline void decompose(uint32_t cp, bool compat, std::vector<uint32_t>& buf) {
    if (cp < 0xA0 || (cp & kRawByte)) {
        buf.push_back(cp);
        return;
    }
    if (cp >= S_BASE && cp < S_BASE + S_COUNT) {  // Hangul syllable: L V (T)
        const uint32_t si = cp - S_BASE;
        buf.push_back(L_BASE + si / N_COUNT);
        buf.push_back(V_BASE + (si % N_COUNT) / T_COUNT);
        if (si % T_COUNT) buf.push_back(T_BASE + si % T_COUNT);
        return;
    }
    if (compat) {
        const int i = find_cp(kCompatCp, kCompatN, cp);
        if (i >= 0) {
            buf.insert(buf.end(), kCompatSeq + kCompatOff[i], kCompatSeq + kCompatOff[i] + kCompatLen[i]);
            return;
        }
    } else {
        const int i = find_cp(kCanonCp, kCanonN, cp);
        if (i >= 0) {
            buf.insert(buf.end(), kCanonSeq + kCanonOff[i], kCanonSeq + kCanonOff[i] + kCanonLen[i]);
            return;
        }
    }
    buf.push_back(cp);
}

inline uint32_t compose_pair(uint32_t a, uint32_t b) {
    if (a >= L_BASE && a < L_BASE + L_COUNT && b >= V_BASE && b < V_BASE + V_COUNT)  // L + V
        return S_BASE + ((a - L_BASE) * V_COUNT + (b - V_BASE)) * T_COUNT;
    if (a >= S_BASE && a < S_BASE + S_COUNT && (a - S_BASE) % T_COUNT == 0 && b > T_BASE && b < T_BASE + T_COUNT)  // LV + T
        return a + (b - T_BASE);
    if ((a | b) & kRawByte) return 0;
    // pairs are sorted by (first, second)
    uint32_t lo = 0, hi = kPairN;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) / 2;
        if (kPairFirst[mid] < a || (kPairFirst[mid] == a && kPairSecond[mid] < b)) lo = mid + 1;
        else hi = mid;
    }
    return (lo < kPairN && kPairFirst[lo] == a && kPairSecond[lo] == b) ? kPairComposite[lo] : 0u;
}

// one segment: form bit 0 = compose (NFC / NFKC), bit 1 = compatibility (NFKC / NFKD)
void normalize_segment(const uint8_t* s, uint64_t n, uint32_t form, std::vector<uint8_t>& out, std::vector<uint32_t>& buf,
                       std::vector<uint8_t>& cls) {
    const bool compose = form & 1u, compat = form & 2u;
    uint64_t i = 0;
    while (i < n) {
        // ASCII run: nothing decomposes, reorders or composes with what follows unless the next scalar is a combining mark
        // or composes with the run's last byte — so the run goes out except for its last byte
        uint64_t j = i;
        while (j < n && s[j] < 0x80) j++;
        if (j > i + 1) {
            out.insert(out.end(), s + i, s + j - 1);
            i = j - 1;
        }
        // a stretch of scalars up to (not including) the next "safe" starter: decomposed, reordered, composed as one unit.
        // The unit ends before a starter of the decomposed stream that cannot compose with what precedes it; taking it
        // at an ASCII byte that follows a non-ASCII stretch (or at the end) is always safe (no ASCII scalar is the second
        // of a composition pair, and none decomposes).
        buf.clear();
        bool first = true;
        while (i < n) {
            if (!first && s[i] < 0x80) break;
            first = false;
            const uint32_t cp = next_cp(s, n, i);
            decompose(cp, compat, buf);
        }
        // canonical ordering: stable sort of every run of non-starters by combining class
        cls.resize(buf.size());
        for (size_t k = 0; k < buf.size(); k++) cls[k] = (uint8_t)ccc_of(buf[k]);
        for (size_t a = 0; a < buf.size();) {
            if (cls[a] == 0) {
                a++;
                continue;
            }
            size_t b = a;
            while (b < buf.size() && cls[b] != 0) b++;
            if (b - a > 1) {  // insertion sort, stable (runs are a few marks long)
                for (size_t x = a + 1; x < b; x++) {
                    const uint32_t cv = buf[x];
                    const uint8_t cc = cls[x];
                    size_t y = x;
                    while (y > a && cls[y - 1] > cc) {
                        buf[y] = buf[y - 1];
                        cls[y] = cls[y - 1];
                        y--;
                    }
                    buf[y] = cv;
                    cls[y] = cc;
                }
            }
            a = b;
        }
        if (compose && !buf.empty()) {
            size_t starter = 0, w = 1;
            uint32_t starter_cp = buf[0];
            int last = cls[0] ? 256 : 0;  // (a string that starts with a mark: nothing composes with that mark)
            for (size_t k = 1; k < buf.size(); k++) {
                const uint32_t ch = buf[k];
                const int cc = cls[k];
                const uint32_t comp = (last < cc || last == 0) ? compose_pair(starter_cp, ch) : 0u;
                if (comp) {
                    buf[starter] = starter_cp = comp;
                    continue;
                }
                if (cc == 0) {
                    starter = w;
                    starter_cp = ch;
                }
                last = cc;
                buf[w] = ch;
                cls[w] = (uint8_t)cc;
                w++;
            }
            buf.resize(w);
        }
        for (uint32_t cp : buf) put_utf8(out, cp);
    }
}

}  // namespace

extern "C" {

const char* tgx_unidata_version(void) { return TGX_UNIDATA_VERSION; }

// form: 0 NFD, 1 NFC, 2 NFKD, 3 NFKC.  Segments text[seg_begin[k], seg_end[k]) -> a packed batch (malloc'd: tgx_free).
tgx_status tgx_normalize_segments(uint32_t form, const uint8_t* text, const uint64_t* seg_begin, const uint64_t* seg_end, uint64_t n_segs,
                                  uint8_t** out_text, uint64_t** out_offs) {
    if (!out_text || !out_offs) return tgx_set_error(TGX_ERR_INVALID, "tgx_normalize_segments: NULL argument");
    *out_text = nullptr;
    *out_offs = nullptr;
    if (form > 3) return tgx_set_error(TGX_ERR_INVALID, "tgx_normalize_segments: form must be 0 (NFD), 1 (NFC), 2 (NFKD) or 3 (NFKC)");
    if (n_segs && (!seg_begin || !seg_end)) return tgx_set_error(TGX_ERR_INVALID, "tgx_normalize_segments: NULL argument");
    for (uint64_t k = 0; k < n_segs; k++)
        if (seg_end[k] < seg_begin[k] || (seg_end[k] > seg_begin[k] && !text)) return tgx_set_error(TGX_ERR_INVALID, "tgx_normalize_segments: bad segment");
    uint64_t total = 0;
    for (uint64_t k = 0; k < n_segs; k++) total += seg_end[k] - seg_begin[k];
    unsigned hw = std::thread::hardware_concurrency();
    const uint64_t n_threads = std::max<uint64_t>(1, std::min<uint64_t>({(uint64_t)(hw ? hw : 1u), 16, n_segs, total / (256u << 10) + 1}));
    // contiguous ranges of segments of about equal bytes per thread
    std::vector<uint64_t> cut(n_threads + 1, n_segs);
    cut[0] = 0;
    {
        uint64_t acc = 0, t = 1;
        for (uint64_t k = 0; k < n_segs && t < n_threads; k++) {
            acc += seg_end[k] - seg_begin[k];
            if (acc >= total * t / n_threads) cut[t++] = k + 1;
        }
    }
    std::vector<std::vector<uint8_t>> bytes(n_threads);
    std::vector<std::vector<uint64_t>> lens(n_threads);
    auto work = [&](uint64_t t) {
        std::vector<uint32_t> buf;
        std::vector<uint8_t> cls;
        std::vector<uint8_t>& o = bytes[t];
        for (uint64_t k = cut[t]; k < cut[t + 1]; k++) {
            const size_t before = o.size();
            normalize_segment(text + seg_begin[k], seg_end[k] - seg_begin[k], form, o, buf, cls);
            lens[t].push_back(o.size() - before);
        }
    };
    {
        std::vector<std::thread> th;
        for (uint64_t t = 1; t < n_threads; t++) th.emplace_back(work, t);
        work(0);
        for (auto& x : th) x.join();
    }
    uint64_t out_total = 0;
    for (auto& b : bytes) out_total += b.size();
    uint8_t* ot = (uint8_t*)malloc(std::max<uint64_t>(out_total, 1) + 64);  // (padded: the encode kernels read a few bytes past the end)
    uint64_t* oo = (uint64_t*)malloc((n_segs + 1) * sizeof(uint64_t));
    if (!ot || !oo) {
        free(ot);
        free(oo);
        return tgx_set_error(TGX_ERR_INVALID, "tgx_normalize_segments: out of host memory");
    }
    uint64_t pos = 0, k = 0;
    oo[0] = 0;
    for (uint64_t t = 0; t < n_threads; t++) {
        if (!bytes[t].empty()) memcpy(ot + pos, bytes[t].data(), bytes[t].size());
        uint64_t p = pos;
        for (uint64_t l : lens[t]) {
            p += l;
            oo[++k] = p;
        }
        pos += bytes[t].size();
    }
    memset(ot + out_total, 0, 64);
    *out_text = ot;
    *out_offs = oo;
    return TGX_OK;
}

}  // extern "C"
