// Ids in their final place (round 4): mark_kernel -> popcount scan -> emit_kernel replace trace_kernel -> tmp ->
// scan of the counts -> compact_kernel (reference src/model.rs:113-126: one back-trace and a reverse).
//
// Rounds 1 - 3 walked a sample's back-pointers from its end, looked every token's id up on the way and wrote the ids
// right-aligned into a scratch row `tmp` (4 bytes per input byte), because where a sample's ids belong is only known
// once every sample's token count is; a scan and a copy (compact_kernel) then packed them: 6.1 GB of traffic and 3.4
// of a pass's 16 ms for putting ids in order.  Now
//
//   mark_kernel   the hop chain alone (trace_body.h's: 64 positions per load, one v_readlane per hop): the ONLY thing it
//                 records is one bit per text byte — "a token ends with this byte" — in a bitmask over the whole batch
//                 (N / 8 bytes + a word of padding per sample).  Samples are back to back and every byte belongs to
//                 exactly one token, so the set bits, in position order, ARE the batch's tokens in output order, across
//                 sample boundaries.
//   scan          an exclusive scan of the mask's 64-bit words' popcounts (rocPRIM): the final index of every word's
//                 first token; the last entry is the batch's token count, and a sample's first id sits at the number
//                 of set bits before its first byte (sample_offs_kernel): the result's offsets for free.
//   emit_kernel   fully parallel, no per-sample chain: a wave takes 256 positions, compacts the set bits into a list,
//                 and every lane turns one token — the bytes after the previous set bit up to its own — into its id
//                 (bytes -> multiply-xorshift hash -> one probe of the bytes -> id table, trace_body.h's lookup) and stores
//                 it at its final place.  Short samples cost nothing extra: nothing here is per sample.
//
// `tmp` and the counts are gone for every vocabulary with tokens of at most 32 bytes and finite scores (the generic
// one-sample-per-wave kernel keeps its fused trace).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>

#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "device_common.h"
#include "kernels.h"

namespace tgx {

// The mask is laid out by SAMPLE: sample s owns the words mword[s] .. mword[s + 1) (ceil(n / 64) of them; the host scans
// the samples' word counts at upload), bit j of its word k = "a token ends with the sample's byte 64 k + j".  A window of
// the hop chain is then exactly one word — a plain store, no atomics, nothing to clear — and padded position order is
// still text order, so the popcount prefix over the words numbers the batch's tokens.
//
// LM: longest token, 16 or 32 (back-pointer bytes hold length - 1 in their low 4 / 5 bits).
// PERM: the back-pointer bytes are in encode4_kernel's permuted layout (bp8_perm), else plain.
template <uint32_t LM, bool PERM>
__global__ __launch_bounds__(256) void mark_kernel(EncodeParams P) {
    constexpr uint32_t LMASK = LM - 1u;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wpb = blockDim.x >> 6;
    const uint32_t n_waves = gridDim.x * wpb;
    const uint32_t wave_id = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * wpb + (threadIdx.x >> 6)));
    // static round-robin over the longest-first order, alternating direction (trace_body.h)
    for (uint64_t base = 0, rnd = 0; base < P.n_samples; base += n_waves, ++rnd) {
        const uint64_t k = base + ((rnd & 1u) ? (uint64_t)(n_waves - 1u - wave_id) : (uint64_t)wave_id);
        if (k >= P.n_samples) continue;
        const uint32_t s = (uint32_t)__builtin_amdgcn_readfirstlane((int)P.order[k]);
        const uint64_t beg = first_u64(P.offs[s]);
        const uint32_t n = (uint32_t)(first_u64(P.offs[s + 1]) - beg);
        const uint32_t reach_n = (n == 0) ? 1u : (uint32_t)__builtin_amdgcn_readfirstlane((int)P.status[s]);
        const uint8_t* __restrict__ bp = P.bp8 + bp8_base(beg, s);
        unsigned long long* const mw = P.endmask + first_u64(P.mword[s]);  // this sample's words
        int64_t q = reach_n ? (int64_t)n - 1 : (int64_t)-1;  // Error::NoPath(n, n) otherwise (model.rs:119)
        uint32_t h_cur = 0;
        if (q >= 0) {
            const uint32_t wq0 = (uint32_t)q & ~63u;
            h_cur = (wq0 + lane < n) ? (uint32_t)bp[PERM ? bp8_perm(wq0 + lane) : wq0 + lane] : 0u;
        }
        // The words of up to 64 consecutive windows wait in one register, window w in lane w & 63, and leave with ONE
        // coalesced store (a store — or an atomic — per window sits in the same in-order queue as the next window's
        // back-pointer load: the chain then waits a memory round trip per window, 2.4 ms per GiB instead of 1.3).
        uint32_t acc_lo = 0, acc_hi = 0;
        while (q >= 0) {
            const uint32_t wq = (uint32_t)q & ~63u;
            const uint32_t h = h_cur;
            // every lane holds where ITS position's token starts (relative to the window): a hop is one v_readlane
            uint32_t prev = lane - ((h & LMASK) + 1u);
            asm volatile("" : "+v"(prev) : : "memory");  // the wait for h_cur lands above the next request
            uint32_t h_next = 0;
            if (wq >= 64u) h_next = (uint32_t)bp[PERM ? bp8_perm(wq - 64u + lane) : wq - 64u + lane];
            uint64_t ends = 0;  // bit j: a token ends with the sample's byte wq + j
            int32_t qq = (int32_t)((uint32_t)q - wq);
            while (qq >= 0) {  // model.rs:113-126, 64 positions per load
                asm("s_bitset1_b64 %0, %1" : "+s"(ends) : "s"(qq));
                qq = (int32_t)readlane_u32(prev, (uint32_t)qq);
            }
            q = (int64_t)wq + qq;
            const uint32_t win = wq >> 6;
            acc_lo = (lane == (win & 63u)) ? (uint32_t)ends : acc_lo;
            acc_hi = (lane == (win & 63u)) ? (uint32_t)(ends >> 32) : acc_hi;
            if ((win & 63u) == 0u) {  // (wave-uniform) the group's lowest window: lanes 0 .. 63 hold windows win .. win + 63
                const uint32_t top = (n - 1u) >> 6;  // the sample's highest window
                if (win + lane <= top) mw[win + lane] = ((unsigned long long)acc_hi << 32) | acc_lo;
            }
            h_cur = h_next;
        }
        if (lane == 0u && !reach_n) atomicMin(P.err_sample, (unsigned long long)s);
    }
}

struct PopcU64 {
    __host__ __device__ uint64_t operator()(unsigned long long x) const {
#if defined(__HIP_DEVICE_COMPILE__)
        return (uint64_t)__popcll(x);
#else
        return (uint64_t)__builtin_popcountll(x);
#endif
    }
};

// out_offs[s] = set bits before sample s's first word (out_offs[S] = the batch's tokens)
__global__ __launch_bounds__(256) void sample_offs_kernel(const uint64_t* __restrict__ mword, uint64_t n_samples, const uint64_t* __restrict__ prefix,
                                                          uint64_t* __restrict__ out_offs) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s > n_samples) return;
    out_offs[s] = prefix[mword[s]];
}

constexpr uint32_t kEmitWords = 4;    // 64-bit mask words (256 positions) appended per round
constexpr uint32_t kEmitRing = 512;   // waiting tokens per wave: < 128 + 256

// one token: text[start, start + len) -> its id (trace_body.h's lookup); false: the bytes are no vocabulary token
template <uint32_t LM>
__device__ __forceinline__ bool emit_lookup(const EncodeParams& P, const uint4* __restrict__ table, uint64_t start, uint32_t len, uint32_t* id_out) {
    constexpr int NW = (int)LM / 4;
    struct __attribute__((packed, aligned(1))) Bytes16 { uint32_t w[4]; };
    const Bytes16* __restrict__ src = reinterpret_cast<const Bytes16*>(P.text + start);
    uint32_t b[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < NW; j += 4) {
        const Bytes16 raw = src[j / 4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t nb = len > 4u * (j + t) ? len - 4u * (j + t) : 0u;  // bytes of the token in this dword
            b[j + t] = nb >= 4u ? raw.w[t] : (raw.w[t] & ((1u << (8u * nb)) - 1u));
        }
    }
    const uint64_t hk = LM == 16 ? tok_hash64_dev(b[0], b[1], b[2], b[3], len, P.tokhash_seed) : tok_hash64_long_dev(b, len, P.tokhash_seed);
    uint32_t slot = (uint32_t)hk & P.tokhash_mask;
    for (uint32_t probe = 0; probe <= P.tokhash_mask; ++probe) {
        const uint4 t = load_rec(table, slot);
        if (t.w == 0u) break;  // empty slot
        if (t.x == (uint32_t)hk && t.y == (uint32_t)(hk >> 32)) {
            *id_out = t.z;
            return true;
        }
        slot = (slot + 1u) & P.tokhash_mask;
    }
    *id_out = 0u;
    return false;
}

// LM = 16 or 32: bytes of a token the hash covers.  Every wave owns a contiguous range of mask words; the set bits of
// the range go through a ring in LDS (as text positions) and are looked up 128 at a time — two tokens per lane, their
// two chains of dependent loads (token bytes, then the table probe) in flight together — in text order, so a token's
// first byte is the one after the token before it (across rounds and samples: the text is contiguous) and its id's final
// index is the range's first index plus its number in the range.
template <uint32_t LM>
__global__ __launch_bounds__(256) void emit_kernel(EncodeParams P) {
    __shared__ uint32_t ring_all[4][kEmitRing];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wpb = blockDim.x >> 6;
    uint32_t* const ring = ring_all[threadIdx.x >> 6];
    const uint64_t n_waves = (uint64_t)gridDim.x * wpb;
    const uint64_t per = ((P.mask_words + n_waves - 1) / n_waves + kEmitWords - 1) / kEmitWords * kEmitWords;  // words per wave
    const uint64_t wave_id = (uint64_t)blockIdx.x * wpb + (threadIdx.x >> 6);
    const uint64_t wa = wave_id * per, wb = (wa + per < P.mask_words) ? wa + per : P.mask_words;
    if (wa >= P.mask_words) return;  // (wave-uniform)
    const uint4* __restrict__ table = reinterpret_cast<const uint4*>(P.tokhash);
    // the sample of word wa: the last s with mword[s] <= wa and words of its own (binary search; empty samples own none)
    uint64_t s = 0;
    {
        uint64_t lo = 0, hi = P.n_samples;  // invariant: mword[lo] <= wa < mword[hi] (mword[n_samples] = mask_words > wa)
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (first_u64(P.mword[mid]) <= wa) lo = mid;
            else hi = mid;
        }
        s = lo;  // (mword[s + 1] > wa: hi = s + 1)
    }
    uint64_t mb = first_u64(P.mword[s]), mb_next = first_u64(P.mword[s + 1]), tbeg = first_u64(P.offs[s]);
    const uint64_t tp0 = tbeg + 64u * (wa - mb);  // text position of the range's first bit: ring entries are relative to it
    // the byte before the range's first token: the sample's start, or the last set bit of the word before (a word holds
    // 64 positions, a token at most LM <= 32: never empty in a sample that was encoded)
    int64_t prev_end = (int64_t)tp0 - 1;
    if (wa > mb) {
        const uint64_t m = first_u64(P.endmask[wa - 1]);
        if (m) prev_end = (int64_t)(tp0 - 64u + 63u - (uint32_t)__builtin_clzll(m));
    }
    uint64_t out = first_u64(P.prefix[wa]);
    uint32_t head = 0, fill = 0;

    auto lookup = [&](uint32_t m) {  // the first m (<= 128) waiting tokens: lane i takes the i-th and the (i + 64)-th
        uint32_t last = 0;
        const bool has0 = lane < m, has1 = lane + 64u < m;
        uint64_t st0 = 0, st1 = 0;
        uint32_t ln0 = 0, ln1 = 0;
        if (has0) {
            const uint32_t xr = ring[(head + lane) & (kEmitRing - 1u)];
            last = xr;
            st0 = lane ? tp0 + ring[(head + lane - 1u) & (kEmitRing - 1u)] + 1u : (uint64_t)(prev_end + 1);
            ln0 = (uint32_t)(tp0 + xr + 1u - st0);
        }
        if (has1) {
            const uint32_t xr = ring[(head + lane + 64u) & (kEmitRing - 1u)];
            last = xr;
            st1 = tp0 + ring[(head + lane + 63u) & (kEmitRing - 1u)] + 1u;
            ln1 = (uint32_t)(tp0 + xr + 1u - st1);
        }
        if (ln0 > LM) ln0 = LM;  // (only in a batch that fails anyway)
        if (ln1 > LM) ln1 = LM;
        uint32_t id0 = 0, id1 = 0;
        bool ok0 = true, ok1 = true;
        if (has0) ok0 = emit_lookup<LM>(P, table, st0, ln0, &id0);
        if (has1) ok1 = emit_lookup<LM>(P, table, st1, ln1, &id1);
        // cannot happen unless a kernel bug corrupted a back-pointer: report, do not fault
        if (!ok0 || !ok1) atomicMin(P.err_sample, (unsigned long long)((ok0 ? st1 : st0) & ((1ULL << 62) - 1ULL)) | (1ULL << 62));
        if (has0) P.ids_out[out + lane] = id0;
        if (has1) P.ids_out[out + lane + 64u] = id1;
        prev_end = (int64_t)(tp0 + readlane_u32(last, (m - 1u) & 63u));
        out += m;
        head = (head + m) & (kEmitRing - 1u);
        fill -= m;
    };

    for (uint64_t w0 = wa; w0 < wb; w0 += kEmitWords) {
#pragma unroll
        for (uint32_t j = 0; j < kEmitWords; ++j) {
            const uint64_t w = w0 + j;
            if (w >= wb) break;  // (wave-uniform)
            while (w >= mb_next) {  // the next sample with words of its own
                s++;
                mb = mb_next;
                mb_next = first_u64(P.mword[s + 1]);
                tbeg = first_u64(P.offs[s]);
            }
            const uint64_t m = first_u64(P.endmask[w]);
            if ((m >> lane) & 1ull) {
                const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                ring[(head + fill + below) & (kEmitRing - 1u)] = (uint32_t)(tbeg + 64u * (w - mb) + lane - tp0);
            }
            fill += (uint32_t)__popcll(m);
        }
        __builtin_amdgcn_wave_barrier();
        while (fill >= 128u) lookup(128u);
        __builtin_amdgcn_wave_barrier();
    }
    while (fill) lookup(fill < 128u ? fill : 128u);
}

hipError_t launch_mark(const EncodeParams& p, uint32_t blocks, uint32_t lm, bool permuted, hipStream_t stream) {
    if (lm <= 16u) hipLaunchKernelGGL((mark_kernel<16, true>), dim3(blocks), dim3(256), 0, stream, p);
    else if (permuted) hipLaunchKernelGGL((mark_kernel<32, true>), dim3(blocks), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((mark_kernel<32, false>), dim3(blocks), dim3(256), 0, stream, p);
    return hipGetLastError();
}
hipError_t mask_scan_temp_bytes(uint64_t n_words, size_t* bytes) {
    auto in = rocprim::make_transform_iterator((const unsigned long long*)nullptr, PopcU64());
    return rocprim::exclusive_scan(nullptr, *bytes, in, (uint64_t*)nullptr, (uint64_t)0, (size_t)(n_words + 1), rocprim::plus<uint64_t>());
}
// prefix[w] = set bits in words 0 .. w - 1, w = 0 .. n_words (the mask has one zero word of padding at its end)
hipError_t launch_mask_scan(const unsigned long long* mask, uint64_t* prefix, uint64_t n_words, void* temp, size_t temp_bytes, hipStream_t stream) {
    auto in = rocprim::make_transform_iterator(mask, PopcU64());
    return rocprim::exclusive_scan(temp, temp_bytes, in, prefix, (uint64_t)0, (size_t)(n_words + 1), rocprim::plus<uint64_t>(), stream);
}
hipError_t launch_sample_offs(const uint64_t* mword, uint64_t n_samples, const uint64_t* prefix, uint64_t* out_offs, hipStream_t stream) {
    const uint32_t blocks = (uint32_t)((n_samples + 1 + 255) / 256);
    hipLaunchKernelGGL(sample_offs_kernel, dim3(blocks), dim3(256), 0, stream, mword, n_samples, prefix, out_offs);
    return hipGetLastError();
}
hipError_t launch_emit(const EncodeParams& p, uint32_t lm, uint32_t num_cus, hipStream_t stream) {
    // (contiguous ranges of at least 64 words — 4 KiB of text — per wave)
    const uint32_t blocks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((p.mask_words + 255) / 256, (uint64_t)num_cus * 8u));
    if (lm <= 16u) hipLaunchKernelGGL(emit_kernel<16>, dim3(blocks), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(emit_kernel<32>, dim3(blocks), dim3(256), 0, stream, p);
    return hipGetLastError();
}

}  // namespace tgx
