// Back-trace + id emission shared by trace_kernel (kernels.hip, tokens <= 16 bytes) and trace32_kernel
// (encode2.hip, tokens <= 32 bytes): one wave per sample, 1-byte back-pointers (token length - 1), ids
// recovered from the token's bytes through the hash table.
//
// The kernels are bound by instruction issue, not by memory: a window of 64 positions holds ~14 tokens, so a
// lookup (token bytes, hash, table probe, store: ~100 instructions) executed per window would run with a
// quarter of its lanes.  The hop chain therefore only appends (end position, length) of the tokens it finds to
// a per-wave ring in LDS, and the lookups run once 64 tokens are waiting: all lanes busy, one pass per ~4.6
// windows; the token's bytes come straight from the text with unaligned 16-byte loads.  The ring's entries are absolute
// (text position of the token's end, its length, the slot of `tmp` its id goes to), so waiting tokens carry over from one
// sample to the next (round 4: a flush per sample — two dependent memory round trips with half of the lanes — was a
// quarter of the kernel on a corpus of 140-byte samples).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_common.h"
#include "kernels.h"

namespace tgx {

constexpr uint32_t kTraceRing = 128;  // entries per wave: < 64 waiting + up to 64 from one window

// STAMP = true is a diagnostic build only (TGX_STAMPS=2): s_memtime stamps around the phases of a window
#define TGX_TRACE_STAMP(i)                                             \
    if (STAMP) {                                                       \
        __builtin_amdgcn_sched_barrier(0);                             \
        const uint64_t _now = (uint64_t)__builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_s_waitcnt(0xC07F);                            \
        __builtin_amdgcn_sched_barrier(0);                             \
        seg[i] += _now - t_last;                                       \
        t_last = _now;                                                 \
    }

// LM: longest token, 16 or 32 (back-pointer bytes hold length - 1 in their low 4 / 5 bits).
// PERM: the back-pointer bytes are in encode4_kernel's permuted layout (bp8_perm), else plain.
// `ring`: kTraceRing entries of LDS owned by the calling wave.
// CARRY: ring entries are absolute and waiting tokens carry over from one sample to the next (corpora of short samples);
// else they are relative to the sample and the ring is flushed at its end (8-byte entries: 0.1 ms per GiB less on samples
// of kilobytes, where a flush per sample costs nothing).
template <bool CARRY> struct TraceRingEntry { using type = uint2; };
template <> struct TraceRingEntry<true> { using type = uint4; };
template <uint32_t LM, bool PERM, bool STAMP, bool CARRY>
__device__ __forceinline__ void trace_body(const EncodeParams& P, typename TraceRingEntry<CARRY>::type* ring) {
    static_assert(LM == 16 || LM == 32, "token lengths of up to 16 or 32 bytes");
    constexpr uint32_t LMASK = LM - 1u;
    constexpr int NW = (int)LM / 4;  // dwords of token bytes
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wpb = blockDim.x >> 6;
    const uint32_t n_waves = gridDim.x * wpb;
    const uint32_t wave_id = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * wpb + (threadIdx.x >> 6)));
    const uint4* __restrict__ table = reinterpret_cast<const uint4*>(P.tokhash);
    uint64_t seg[6] = {0, 0, 0, 0, 0, 0};
    uint64_t t_last = STAMP ? (uint64_t)__builtin_amdgcn_s_memtime() : 0;
    uint32_t iters = 0;
    // Static round-robin over the longest-first order.  (Claiming samples from a global counter, as
    // encode4_kernel does per row, gained nothing here on 64 KiB samples and cost 2-5x on corpora of
    // short samples: one atomic per sample on one address serialises the waves.)
    // Rounds alternate direction (wave w takes w, 2W - 1 - w, 2W + w, ...): the order descends by length, so a
    // plain stride gives wave 0 the longest sample of every round.
    // A sample's description is three dependent loads (order -> offsets / status -> first window of back-pointers): on a
    // corpus of 140-byte samples a wave met that chain 930 times, ~3 us each, half of the kernel.  So the loop is a
    // pipeline two samples deep (round 4): at the top of a round the index of the round after next and the offsets of
    // the next one are requested; its first window is requested before this round's last lookups.
    const uint64_t n_rounds = (P.n_samples + n_waves - 1) / n_waves;
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    // (the loads return into vector registers and are read — v_readfirstlane, hence waited for — only where they are used:
    // a lambda that returned the scalar would wait on the spot)
    auto sample_of_round = [&](uint64_t rnd) -> uint32_t {  // raw: not waited for
        const uint64_t k = rnd * n_waves + ((rnd & 1u) ? (uint64_t)(n_waves - 1u - wave_id) : (uint64_t)wave_id);
        return (rnd < n_rounds && k < P.n_samples) ? P.order[k] : kNone;
    };
    struct MetaRaw {
        uint64_t beg, end;
        uint32_t reach;
    };
    struct Meta {
        uint32_t s;
        uint64_t beg;
        uint32_t n, reach_n;
    };
    auto meta_request = [&](uint32_t smp) -> MetaRaw {  // smp: scalar
        MetaRaw r{0, 0, 1u};
        if (smp != kNone) {
            r.beg = P.offs[smp];
            r.end = P.offs[smp + 1];
            r.reach = P.status[smp];
        }
        return r;
    };
    auto meta_read = [&](uint32_t smp, const MetaRaw& r) -> Meta {
        Meta mt{smp, 0, 0, 1u};
        if (smp != kNone) {
            mt.beg = first_u64(r.beg);
            mt.n = (uint32_t)(first_u64(r.end) - mt.beg);
            mt.reach_n = (mt.n == 0) ? 1u : (uint32_t)__builtin_amdgcn_readfirstlane((int)r.reach);
        }
        return mt;
    };
    auto first_window = [&](const Meta& mt) -> uint32_t {  // back-pointers of the window that holds the sample's last byte
        if (mt.s == kNone || mt.n == 0 || !mt.reach_n || (P.flags & 4u)) return 0u;
        const uint8_t* __restrict__ bpn = P.bp8 + bp8_base(mt.beg, mt.s);
        const uint32_t wq0 = (mt.n - 1u) & ~63u;
        return (wq0 + lane < mt.n) ? (uint32_t)bpn[PERM ? bp8_perm(wq0 + lane) : wq0 + lane] : 0u;
    };
    uint32_t head = 0, fill = 0;  // ring: `fill` tokens wait from entry `head` on
    uint64_t flush_beg = 0, flush_out_top = 0;  // !CARRY: the current sample's first byte, one past the tmp slot of its next token looked up
    // ids of the first m waiting tokens: lane i takes the i-th
    auto lookup = [&](uint32_t m) {
        if (lane < m && !(P.flags & 16u)) {  // flags 16: timing experiment, no lookups
            const auto e = ring[(head + lane) & (kTraceRing - 1u)];
            uint32_t len;
            uint64_t end, out;
            if constexpr (CARRY) {
                len = e.y >> 16;
                end = ((uint64_t)(e.y & 0xFFFFu) << 32) | e.x;
                out = ((uint64_t)e.w << 32) | e.z;
            } else {
                len = e.y;
                end = flush_beg + e.x;
                out = flush_out_top - 1u - lane;
            }
            // token = text[end - len .. end): LM bytes from its start (the text is padded), cut to len
            struct __attribute__((packed, aligned(1))) Bytes16 { uint32_t w[4]; };
            const Bytes16* __restrict__ src = reinterpret_cast<const Bytes16*>(P.text + (end - len));
            uint32_t b[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < NW; j += 4) {
                const Bytes16 raw = src[j / 4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t nb = len > 4u * (j + i) ? len - 4u * (j + i) : 0u;  // bytes of the token in this dword
                    b[j + i] = nb >= 4u ? raw.w[i] : (raw.w[i] & ((1u << (8u * nb)) - 1u));
                }
            }
            const uint64_t hk = LM == 16 ? tok_hash64_dev(b[0], b[1], b[2], b[3], len, P.tokhash_seed)
                                         : tok_hash64_long_dev(b, len, P.tokhash_seed);
            uint32_t slot = (uint32_t)hk & P.tokhash_mask;
            uint32_t id = 0;
            bool found = false;
            for (uint32_t probe = 0; probe <= P.tokhash_mask && !found; ++probe) {
                const uint4 t = load_rec(table, slot);  // one 16-byte load, not two dependent ones
                if (t.w == 0u) break;  // empty slot: the back-pointer does not name a vocabulary token
                if (t.x == (uint32_t)hk && t.y == (uint32_t)(hk >> 32)) {
                    id = t.z;
                    found = true;
                }
                slot = (slot + 1u) & P.tokhash_mask;
            }
            // cannot happen unless a kernel bug corrupted a back-pointer: report (the text position), do not fault
            if (!found) atomicMin(P.err_sample, (unsigned long long)((end - len) & ((1ULL << 62) - 1ULL)) | (1ULL << 62));
            P.tmp[out] = id;
        }
        if constexpr (!CARRY) flush_out_top -= m;
        head = (head + m) & (kTraceRing - 1u);
        fill -= m;
    };
    Meta cur;
    {
        const uint32_t s0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)sample_of_round(0));
        cur = meta_read(s0, meta_request(s0));
    }
    uint32_t s_next_raw = sample_of_round(1);
    uint32_t h_first = first_window(cur);
    for (uint64_t rnd = 0; rnd < n_rounds; ++rnd) {
        const uint32_t s_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_next_raw);  // (requested a round ago)
        const uint32_t s_after_raw = sample_of_round(rnd + 2);  // requested now, read at the top of the next round
        const MetaRaw nxt_raw = meta_request(s_next);           // requested now, read before this round's last lookups
        const uint32_t s = cur.s;
        const uint64_t beg = cur.beg;
        const uint32_t n = cur.n;
        const uint32_t reach_n = cur.reach_n;
        const bool have = s != kNone;
        const uint8_t* __restrict__ bp = P.bp8 + (have ? bp8_base(beg, s) : 0);
        uint32_t total = 0;
        uint64_t out_top = beg + n;   // one past the tmp slot of the sample's next token (ids are right-aligned)
        flush_beg = beg;
        flush_out_top = out_top;
        int64_t q = (have && reach_n) ? (int64_t)n - 1 : (int64_t)-1;  // Error::NoPath(n, n) otherwise (model.rs:119)
        if (P.flags & 4u) q = -1;

        // Windows are visited top-down and a token is shorter than a window, so the next window is always the
        // one below: its back-pointers are requested right after the wait for the current window's and have
        // the whole hop chain to arrive (the compiler waits for every outstanding load at once).
        uint32_t h_cur = h_first;  // (requested a round ago)
        TGX_TRACE_STAMP(0)  // sample setup
        while (q >= 0) {
            iters++;
            const uint32_t wq = (uint32_t)q & ~63u;
            const uint32_t idx = wq + lane;
            const uint32_t h = h_cur;
            // Every lane holds where ITS position's token starts (relative to the window), so a hop is one
            // v_readlane.  (Tried without gain: the 64 back-pointers packed into four 64-bit scalars for a
            // pure-SALU chain; four hops per taken branch.)
            uint32_t prev = lane - ((h & LMASK) + 1u);
            asm volatile("" : "+v"(prev) : : "memory");  // the wait for h_cur lands above the next request
            uint32_t h_next = 0;
            if (wq >= 64u) h_next = (uint32_t)bp[PERM ? bp8_perm(wq - 64u + lane) : wq - 64u + lane];
            TGX_TRACE_STAMP(1)  // window loads issued / consumed
            uint64_t ends = 0;
            int32_t qq = (int32_t)((uint32_t)q - wq);
            while (qq >= 0) {  // model.rs:113-126, 64 positions per load
                asm("s_bitset1_b64 %0, %1" : "+s"(ends) : "s"(qq));  // ends |= 1 << qq
                qq = (int32_t)readlane_u32(prev, (uint32_t)qq);
            }
            q = (int64_t)wq + qq;
            const uint32_t cnt = (uint32_t)__popcll(ends);
            TGX_TRACE_STAMP(2)  // hops
            if ((ends >> lane) & 1ULL) {
                const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(ends >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ends, 0u));
                if constexpr (CARRY) {
                    const uint64_t end = beg + idx + 1u, out = out_top - 1u - (cnt - 1u - below);
                    ring[(head + fill + (cnt - 1u - below)) & (kTraceRing - 1u)] =
                        make_uint4((uint32_t)end, (uint32_t)(end >> 32) | (((h & LMASK) + 1u) << 16), (uint32_t)out, (uint32_t)(out >> 32));
                } else {
                    ring[(head + fill + (cnt - 1u - below)) & (kTraceRing - 1u)] = make_uint2(idx + 1u, (h & LMASK) + 1u);
                }
            }
            fill += cnt;
            total += cnt;
            out_top -= cnt;
            __builtin_amdgcn_wave_barrier();
            if (fill >= 64u) lookup(64u);
            TGX_TRACE_STAMP(3)  // ring append, lookups
            h_cur = h_next;
        }
        const Meta nxt = meta_read(s_next, nxt_raw);
        h_first = first_window(nxt);  // (!CARRY: the next sample's first window travels while this one's last tokens are looked up)
        if constexpr (!CARRY) {
            if (fill) lookup(fill);
        }
        if (have && lane == 0) {
            P.counts[s] = total;
            if (!reach_n) atomicMin(P.err_sample, (unsigned long long)s);
        }
        cur = nxt;
        s_next_raw = s_after_raw;
    }
    if (fill) lookup(fill);  // (< 64 are left)
    if (STAMP && lane == 0 && P.stamps) {
        unsigned long long* o = P.stamps + (size_t)wave_id * 8u;
        for (int i = 0; i < 5; ++i) o[i] = seg[i];
        o[5] = iters;
    }
}

}  // namespace tgx
