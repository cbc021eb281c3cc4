// HIP kernels (gfx950 / CDNA4, wave64) for the Unigram encode path.
//
// Replaces the reference's Model::encode (src/model.rs:59-129) + the rayon batch
// loop (src/tokenizer.rs:102-123).  One WAVEFRONT per sample:
//
//   match  : lane i walks the flattened trie from byte position p0+i (64
//            positions in parallel; one 16-byte gather per trie step), leaving a
//            bitmask of matching lengths in a register and (score, handle) per
//            match in LDS.
//   relax  : positions are then finalised in order.  Lane j is the accumulator of
//            the end position e with e % 64 == j; finalising position p pushes
//            best[p] + score(p, L) into lane (p + L) % 64 for every match of
//            length L at once (one f64 add + compare per lane).  Candidates reach
//            each accumulator in ascending start order and replace it only on a
//            strict '>', i.e. exactly the reference's relaxation order
//            (model.rs:83-110): bit-exact scores, longest-token-wins ties.
//   trace  : back-pointers (slot << 6 | len - 1) go to an HBM scratch row per
//            sample; the same wave walks them back 64 positions at a time with
//            scalar readlane hops and writes the token ids right-aligned into the
//            sample's slice of a temporary row.
//
// No MFMA: this is byte indexing plus f64 add/compare.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "kernels.h"
#include "trace_body.h"
#include "device_common.h"

namespace tgx {

struct WaveCtx {
    const uint4* __restrict__ trie;
    uint32_t root_base;
    double dropout;
    uint64_t seed;
    bool use_dropout;
    uint32_t flags;
};

// ---- generic block: any LM <= 64, any number of positions (also the sample tail) ----
// Processes positions p0 .. min(p0 + 63, n): match, relax, store the block's
// back-pointers.  Sets reach_n when position n lies in this block.
__device__ __forceinline__ void block_generic(const WaveCtx& C, const uint8_t* __restrict__ text, uint32_t n,
                                              uint32_t s, uint32_t p0, uint32_t lane, uint32_t LM,
                                              double* sc, uint32_t* hl, uint8_t* txt, uint32_t* __restrict__ bp,
                                              double& acc, uint32_t& bpv, uint64_t& reach, uint32_t& reach_n) {
    const uint32_t p = p0 + lane;
    // stage the block's text (+ LM look-ahead) in LDS
    txt[lane] = (p < n) ? text[p] : (uint8_t)0;
    txt[lane + 64] = (p + 64 < n) ? text[p + 64] : (uint8_t)0;
    __builtin_amdgcn_wave_barrier();

    // match: TrieIterator::next (trie.rs:51-63) for 64 start positions at once
    const uint32_t rem = (p < n) ? (n - p) : 0u;
    const uint32_t maxd = rem < LM ? rem : LM;
    uint32_t cur = 0, base = C.root_base;
    uint64_t m = 0;
    bool alive = maxd > 0;
    if (C.flags & 1u) {  // experiment: no trie walk, fake 3 matches per position
        alive = false;
        m = maxd >= 3 ? 7ULL : (maxd ? 1ULL : 0ULL);
    }
    for (uint32_t d = 0; d < LM; ++d) {
        alive = alive && (d < maxd);
        if (__builtin_amdgcn_ballot_w64(alive) == 0) break;
        if (alive) {
            const uint32_t c = txt[lane + d];
            const uint32_t t = base ^ c;
            const uint4 r = load_rec(C.trie, t);
            if (r.x == cur) {
                cur = t;
                base = r.y & 0x7FFFFFFFu;
                if (r.y >> 31) {
                    bool keep = true;
                    // model.rs:100: kept iff dropout <= 0 || len <= 1 || dropout < rand
                    if (C.use_dropout && d >= 1) keep = C.dropout < dropout_u01(C.seed, s, p, d + 1);
                    if (keep) {
                        m |= 1ULL << d;
                        sc[kFront + lane * LM + d] = __hiloint2double((int)r.w, (int)r.z);
                        hl[kFront + lane * LM + d] = (t << 6) | d;
                    }
                }
            } else {
                alive = false;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();

    // relax: finalise positions p0 .. p0+63 in order (model.rs:83-110)
    const uint32_t left = n - p0;  // positions < n in this block: min(64, left)
    const uint32_t steps = left < 64u ? left : 64u;
    uint32_t fin = 0;
    for (uint32_t i = 0; i < ((C.flags & 2u) ? 0u : steps); ++i) {
        fin = (lane == i) ? bpv : fin;  // back-pointer of position p0+i is final now
        if (!((reach >> i) & 1ULL)) continue;  // model.rs:85-87 unreachable: skip
        const double best = readlane_f64(acc, i);
        const uint64_t mi = readlane_u64(m, i);
        reach &= ~(1ULL << i);  // lane i now accumulates position p0 + i + 64
        if (mi == 0) continue;
        const uint64_t active = rotl64(mi, i + 1);  // length L -> lane (i + L) % 64
        const uint32_t tj = (lane - i - 1u) & 63u;   // L - 1 for this lane
        const double sv = sc[kFront + i * LM + tj];
        const uint32_t hv = hl[kFront + i * LM + tj];
        const double cand = best + sv;  // model.rs:98
        const uint64_t gt = __builtin_amdgcn_ballot_w64(cand > acc);
        const uint64_t take = active & (~reach | gt);  // model.rs:101: empty, or strict '>'
        acc = sel_f64(take, cand, acc);
        bpv = sel_u32(take, hv, bpv);
        reach |= active;
    }
    if (left < 64u) {  // position n itself sits in this block
        fin = (lane == left) ? bpv : fin;
        reach_n = (uint32_t)((reach >> left) & 1ULL);
    }
    const uint32_t e = p0 + lane;
    if (e >= 1 && e <= n) bp[e - 1] = fin;
}

// ---- fast block: LMT in {16, 32}, a FULL block (all 64 positions < n) --------------
// Same arithmetic as block_generic with everything that can be static made static:
// the text window lives in registers, the trie walk and the 64 relaxation steps are
// fully unrolled (lane selects, rotate amounts and LDS offsets are immediates), masks
// stay in SGPR pairs.  Back-pointers of lanes 0..31 are stored mid-block, 32..63 at
// the end: with LMT <= 32 a lane is only overwritten (as accumulator of position
// e + 64) after those points.
template <int LMT>
__device__ __forceinline__ void block_fast(const WaveCtx& C, const uint8_t* __restrict__ text, uint32_t n,
                                           uint32_t s, uint32_t p0, uint32_t lane, double* sc, uint32_t* hl,
                                           uint32_t* __restrict__ bp, double& acc, uint32_t& bpv,
                                           uint64_t& reach) {
    static_assert(LMT == 16 || LMT == 32, "fast path handles max token length <= 32");
    const uint32_t p = p0 + lane;
    // text window: bytes p .. p+LMT-1 from aligned dwords (the buffer is padded)
    const uintptr_t addr = reinterpret_cast<uintptr_t>(text + p);
    const uint32_t sh = (uint32_t)(addr & 3u);
    const uint32_t* __restrict__ wp = reinterpret_cast<const uint32_t*>(addr & ~uintptr_t(3));
    uint32_t w[LMT / 4 + 1];
#pragma unroll
    for (int k = 0; k <= LMT / 4; ++k) w[k] = wp[k];
    uint32_t bytes[LMT / 4];
#pragma unroll
    for (int k = 0; k < LMT / 4; ++k) bytes[k] = __builtin_amdgcn_alignbyte(w[k + 1], w[k], sh);

    const uint32_t rem = n - p;  // > 0 in a full block
    const uint32_t maxd = rem < (uint32_t)LMT ? rem : (uint32_t)LMT;
    uint32_t cur = 0, base = C.root_base;
    uint32_t m = 0;
    bool alive = true;
    if (C.flags & 1u) {
        alive = false;
        m = maxd >= 3 ? 7u : 1u;
    }
    double* scw = sc + kFront + lane * LMT;
    uint32_t* hlw = hl + kFront + lane * LMT;
#pragma unroll
    for (int d = 0; d < LMT; ++d) {
        alive = alive && ((uint32_t)d < maxd);
        if (__builtin_amdgcn_ballot_w64(alive) == 0) break;
        if (alive) {
            const uint32_t c = (bytes[d >> 2] >> ((d & 3) * 8)) & 0xFFu;
            const uint32_t t = base ^ c;
            const uint4 r = load_rec(C.trie, t);
            if (r.x == cur) {
                cur = t;
                base = r.y & 0x7FFFFFFFu;
                if (r.y >> 31) {
                    bool keep = true;
                    if (C.use_dropout && d >= 1) keep = C.dropout < dropout_u01(C.seed, s, p, d + 1);
                    if (keep) {
                        m |= 1u << d;
                        scw[d] = __hiloint2double((int)r.w, (int)r.z);
                        hlw[d] = (t << 6) | (uint32_t)d;
                    }
                }
            } else {
                alive = false;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (C.flags & 2u) return;

    // relax, 64 static steps.  Lane j reads row u at column (j - u - 1) & 63:
    // entry index u*LMT + j - u - 1 (+64 for the lanes that wrapped, j <= u).
    const double* scr = sc + kFront + lane - 1;  // + u*(LMT-1) [+64]
    const uint32_t* hlr = hl + kFront + lane - 1;
#pragma unroll
    for (int u = 0; u < 64; ++u) {
        if (u == 32) {
            if (lane < 32u && p >= 1u) bp[p - 1] = bpv;  // positions p0 .. p0+31 are final
        }
        const uint64_t ubit = 1ULL << u;
        const bool reachable = (reach & ubit) != 0;  // model.rs:85-87
        uint32_t mi = readlane_u32(m, (uint32_t)u);
        mi = reachable ? mi : 0u;
        const double best = readlane_f64(acc, (uint32_t)u);
        reach &= ~ubit;  // lane u now accumulates position p0 + u + 64
        const uint64_t active = rotl64((uint64_t)mi, (uint32_t)(u + 1));
        double sv;
        uint32_t hv;
        if (u + LMT >= 64) {  // some targets wrap around the lane ring (u + L >= 64)
            const uint64_t wrapped = (u == 63) ? ~0ULL : ((1ULL << (u + 1)) - 1ULL);  // lanes <= u
            const uint32_t off = sel_u32(wrapped, 64u, 0u);
            sv = scr[u * (LMT - 1) + (int)off];
            hv = hlr[u * (LMT - 1) + (int)off];
        } else {
            sv = scr[u * (LMT - 1)];
            hv = hlr[u * (LMT - 1)];
        }
        const double cand = best + sv;  // model.rs:98
        const uint64_t gt = __builtin_amdgcn_ballot_w64(cand > acc);
        const uint64_t take = active & (~reach | gt);  // model.rs:101
        acc = sel_f64(take, cand, acc);
        bpv = sel_u32(take, hv, bpv);
        reach |= active;
    }
    if (lane >= 32u) bp[p - 1] = bpv;  // positions p0+32 .. p0+63
}

__device__ __forceinline__ void trace_sample(const EncodeParams& P, uint32_t s, uint64_t beg, uint32_t n,
                                             uint32_t lane, const uint32_t* __restrict__ bp, uint32_t reach_n) {
    // follow the back-pointers from n (model.rs:113-126), 64 positions per hop group
    __threadfence_block();  // this wave's bp stores are read back below
    uint32_t total = 0;
    uint64_t cursor = beg + n;  // one past this sample's slice of tmp
    // Error::NoPath(n, n), model.rs:119: nothing to trace, the sample is reported below
    int64_t q = reach_n ? (int64_t)n - 1 : (int64_t)-1;  // index into bp of the current end position
    if (P.flags & 4u) q = -1;  // experiment: no trace
    while (q >= 0) {
        const uint32_t wq = (uint32_t)q & ~63u;
        const uint32_t idx = wq + lane;
        const uint32_t h = (idx < n) ? bp[idx] : 0u;
        uint64_t ends = 0;
        int32_t qq = (int32_t)((uint32_t)q - wq);
        while (qq >= 0) {
            const uint32_t hh = readlane_u32(h, (uint32_t)qq);
            ends |= 1ULL << qq;
            qq -= (int32_t)(hh & 63u) + 1;
        }
        q = (int64_t)wq + qq;
        const uint32_t cnt = (uint32_t)__popcll(ends);
        if ((ends >> lane) & 1ULL) {
            // a handle outside the table can only come from a kernel bug; report it as a
            // failed sample instead of faulting the device
            const uint32_t slot = h >> 6;
            if (slot >= P.n_slots) atomicMin(P.err_sample, (unsigned long long)s | (1ULL << 62));
            const uint32_t id = slot < P.n_slots ? P.tokid[slot] : 0u;
            const uint32_t above = (uint32_t)__popcll((ends >> lane) >> 1);
            P.tmp[cursor - 1 - above] = id;
        }
        cursor -= cnt;
        total += cnt;
    }
    if (lane == 0) {
        P.counts[s] = total;
        if (!reach_n) atomicMin(P.err_sample, (unsigned long long)s);
    }
}

// LMT = 0: generic (runtime LM <= 64); LMT = 16 / 32: fast path for full blocks.
template <int LMT>
__global__ __launch_bounds__(256) void encode_kernel(EncodeParams P) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t LM = LMT ? (uint32_t)LMT : P.lm;
    const uint32_t entries = wave_lds_entries(LM);
    unsigned char* wbase = smem + (size_t)wave * wave_lds_bytes(LM);
    double* sc = reinterpret_cast<double*>(wbase);
    uint32_t* hl = reinterpret_cast<uint32_t*>(wbase + (size_t)entries * 8u);
    uint8_t* txt = reinterpret_cast<uint8_t*>(wbase + (size_t)entries * 12u);
    WaveCtx C;
    C.trie = reinterpret_cast<const uint4*>(P.trie);
    C.root_base = P.root_base;
    C.dropout = P.dropout;
    C.seed = P.seed;
    C.use_dropout = P.dropout > 0.0;
    C.flags = P.flags;

    // Static round-robin over the longest-first order: wave w takes samples w, w + W,
    // w + 2W, ...  (no work queue: every wave's trip count is known at launch).
    const uint32_t wpb = blockDim.x >> 6;  // waves per block: 4 for short tokens, fewer when LDS-bound
    const uint32_t n_waves = gridDim.x * wpb;
    const uint32_t wave_id = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * wpb + wave));
    for (uint64_t k = wave_id; k < P.n_samples; k += n_waves) {
        const uint32_t s = (uint32_t)__builtin_amdgcn_readfirstlane((int)P.order[k]);
        const uint64_t beg = first_u64(P.offs[s]);
        const uint32_t n = (uint32_t)(first_u64(P.offs[s + 1]) - beg);
        const uint8_t* __restrict__ text = P.text + beg;
        uint32_t* __restrict__ bp = P.bp + beg;  // bp[e - 1] for end position e in 1..n

        double acc = 0.0;    // best[e] of the end position this lane accumulates
        uint32_t bpv = 0;    // its back-pointer: slot << 6 | (len - 1)
        uint64_t reach = 1;  // wave-uniform: bit j = lane j holds a value (position 0: score 0)
        uint32_t reach_n = (n == 0) ? 1u : 0u;

        uint32_t p0 = 0;
        if (LMT != 0) {
            for (; p0 + 64u <= n; p0 += 64u)
                block_fast<(LMT ? LMT : 16)>(C, text, n, s, p0, lane, sc, hl, bp, acc, bpv, reach);
        }
        for (; p0 <= n; p0 += 64u)
            block_generic(C, text, n, s, p0, lane, LM, sc, hl, txt, bp, acc, bpv, reach, reach_n);

        trace_sample(P, s, beg, n, lane, bp, reach_n);
    }
}

// ---- four samples per wave (max token length <= 16) -------------------------------
// The relaxation of one sample keeps only <= 16 lanes busy, so the wave is split into
// four 16-lane rows, each with its own sample, advancing in lock-step over blocks of
// 16 positions.  Per step every VALU instruction now finalises FOUR positions: the
// source position's best score and match mask are broadcast inside each row with DPP
// (row_newbcast), lane l of a row accumulates the end position e ≡ l (mod 16), and a
// token of length L pushes into lane (u + L) mod 16.  Same candidate order and strict
// '>' as block_generic, hence the same bits.  Back-pointers and the per-sample
// "end reachable" flag go to HBM; trace_kernel turns them into ids.

// LDS layout of a 16-position group (rows4): entry (row, len - 1) lives at column
// (len - 1 + row) & 15 of its row (row = lane that matched it).  The target lane
// l = (u + len) & 15 of step u then always reads column (l - 1) & 15 — a per-lane
// constant, so relax reads need no address arithmetic — and the 64 match-phase writers
// of one depth spread over 16 columns instead of hitting one bank.
//
// "No token" is encoded as a score of -inf (rows are reset before every walk) and "no
// value yet" as an accumulator of -inf, so one comparison `cand > acc` is the whole
// relaxation rule of model.rs:100-101 — empty targets take the first finite candidate,
// later ones only when strictly greater, absent tokens and unreachable sources give
// cand = -inf and never win.  This needs every vocabulary score to be finite; models
// with +-inf / NaN scores use the generic kernel instead.
// (kNoStep and relax4_step live in device_common.h: encode5.hip shares them)

// rows4 LDS: 1024 f64 scores (8 KiB) per wave and 16-position group — 20 waves per CU.
constexpr uint32_t kRows4Entries = 1024;
constexpr uint32_t kRows4GroupBytes = kRows4Entries * 8u;  // 8192

// PPL = positions per lane and iteration: a row advances 16*PPL positions per trip of
// the dependent gather chain.
// STAMP = true is a diagnostic build only (TGX_STAMPS=1): s_memtime stamps around the phases
// of an iteration, summed per wave into P.stamps; its run time is not representative.
#define TGX_STAMP(i)                                                   \
    if (STAMP) {                                                       \
        __builtin_amdgcn_sched_barrier(0);                             \
        const uint64_t _now = (uint64_t)__builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_s_waitcnt(0xC07F);                            \
        __builtin_amdgcn_sched_barrier(0);                             \
        seg[i] += _now - t_last;                                       \
        t_last = _now;                                                 \
    }

// ROOT: the 256 records of the root's children (one 4 KiB block of the double array) are copied into
// the block's LDS, and the first step of every walk — always a full 64-lane gather — reads them there.
// launch_bounds(.., 6 waves per SIMD) for PPL = 1: at most 80 VGPRs.  Two blocks of ten waves put SIX waves on some SIMD; with 81+
// registers (which the compiler picks on small source changes: launch_bounds(1024) alone allows 128) the
// second block does not fit and the kernel silently runs at half occupancy (17.8 -> 28.1 ms).
template <bool DROPOUT, int PPL, bool STAMP, bool ROOT>
__global__ __launch_bounds__(1024, (PPL == 1 ? 6 : 1)) void encode4_kernel(EncodeParams P) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr uint32_t LM = 16;
    constexpr uint32_t SPAN = 16u * PPL;  // positions per row and iteration
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t l = lane & 15u, r = lane >> 4;
    const uint32_t wave = threadIdx.x >> 6;
    const uint4* __restrict__ trie = reinterpret_cast<const uint4*>(P.trie);
    // (An LDS copy of the hottest trie slots — 2048 slots take 3/4 of all gathers — was tried twice, per
    // block and shared by one 16-wave block per CU, and lost both times: 21.2 ms against 19.7 ms at 1 GiB.
    // A walk step still waits for its slowest lane, which goes to L2, and pays an extra LDS read and a
    // select; the LDS pipe is as loaded as the gather path.)
    const uint4* rootc = reinterpret_cast<const uint4*>(smem);
    unsigned char* wbase = smem + (ROOT ? 4096u : 0u) + (size_t)wave * (PPL * kRows4GroupBytes);
    if (ROOT) {
        uint4* rw = reinterpret_cast<uint4*>(smem);
        for (uint32_t i = threadIdx.x; i < 256u; i += blockDim.x) rw[i] = trie[(P.root_base & ~255u) + i];
        __syncthreads();
    }
    const uint32_t wpb = blockDim.x >> 6;

    // per-row state (identical in the 16 lanes of a row)
    uint32_t s = 0, n = 0, p0 = 0;
    uint64_t beg = 0;
    bool live = false, need_new = true;
    const double ninf = -__builtin_huge_val();
    double acc = ninf;            // best[e] of the end position this lane accumulates; -inf = none yet
    uint32_t bpv = kNoStep;       // step (source position mod 16) that pushed the current best
    uint32_t wn[4 * PPL + 1];  // prefetched text window of the next block
    uint32_t pk = 0, pk_j = 0;  // back-pointer bytes of this lane's current group of 64 positions, and its last index
    bool pk_dirty = false;
#pragma unroll
    for (int q = 0; q <= 4 * PPL; ++q) wn[q] = 0;
    uint64_t seg[6] = {0, 0, 0, 0, 0, 0};
    uint64_t t_last = STAMP ? (uint64_t)__builtin_amdgcn_s_memtime() : 0;
    uint32_t iters = 0;

    for (;;) {
        // ---- rows that finished their sample take the next one
        // ---- rows that finished their sample claim the next one (device_common.h: claim_rows)
        {
            const uint64_t k = claim_rows(P.queue, need_new, r);
            if (need_new) {
                live = k < P.n_samples;
                if (live) {
                    s = P.order[k];
                    beg = P.offs[s];
                    n = (uint32_t)(P.offs[s + 1] - beg);
                }
            }
        }
        if (need_new) {
            p0 = 0;
            acc = (l == 0u) ? 0.0 : ninf;  // position 0: score 0, reachable
            bpv = kNoStep;
            pk_dirty = false;
        }
        const bool fresh_row = need_new;
        need_new = false;
        if (__builtin_amdgcn_ballot_w64(live) == 0) break;  // every row ran out of samples
        iters++;
        TGX_STAMP(0)  // sample switching

        // ---- match: 64*PPL trie walks; lane (r, l) owns positions p0 + 16*g + l, g < PPL.
        // The text window of a row's NEXT block was requested one iteration ago (wn); only
        // rows that just switched samples load theirs now.
        const uintptr_t addr = reinterpret_cast<uintptr_t>(P.text + (live ? beg + p0 + l : 0));
        const uint32_t sh = (uint32_t)(addr & 3u);
        const uint32_t* __restrict__ wp = reinterpret_cast<const uint32_t*>(addr & ~uintptr_t(3));
        uint32_t w[4 * PPL + 1];
        if (fresh_row) {
#pragma unroll
            for (int q = 0; q <= 4 * PPL; ++q) w[q] = wp[q];
        } else {
#pragma unroll
            for (int q = 0; q <= 4 * PPL; ++q) w[q] = wn[q];
        }
        uint32_t bytes[PPL][4];
#pragma unroll
        for (int g = 0; g < PPL; ++g)
#pragma unroll
            for (int q = 0; q < 4; ++q) bytes[g][q] = __builtin_amdgcn_alignbyte(w[4 * g + q + 1], w[4 * g + q], sh);

        TGX_STAMP(1)  // text window
        // every (row, len) starts as "no token"
#pragma unroll
        for (int g = 0; g < PPL; ++g) {
            // the wave sweeps its 8 KiB of scores linearly (1 KiB per instruction, conflict-free)
            double2* grp = reinterpret_cast<double2*>(wbase + g * kRows4GroupBytes);
#pragma unroll
            for (int q = 0; q < 8; ++q) grp[q * 64 + lane] = make_double2(ninf, ninf);
        }
        uint32_t pg[PPL], maxd[PPL], cur[PPL], base[PPL], m[PPL];
        bool alive[PPL];
#pragma unroll
        for (int g = 0; g < PPL; ++g) {
            pg[g] = p0 + 16u * g + l;
            const uint32_t rem = (live && pg[g] < n) ? (n - pg[g]) : 0u;
            maxd[g] = rem < LM ? rem : LM;
            cur[g] = 0;
            base[g] = P.root_base;
            m[g] = 0;
            alive[g] = maxd[g] > 0;
            if (P.flags & 1u) {
                alive[g] = false;
                m[g] = maxd[g] >= 3 ? 7u : (maxd[g] ? 1u : 0u);
            }
        }
#pragma unroll
        for (int d = 0; d < (int)LM; ++d) {
            bool any = false;
#pragma unroll
            for (int g = 0; g < PPL; ++g) {
                alive[g] = alive[g] && ((uint32_t)d < maxd[g]);
                any = any || alive[g];
            }
            if (__builtin_amdgcn_ballot_w64(any) == 0) break;
            if (any) {
                // all of this lane's gathers of the step are issued before any is consumed
                uint4 rec[PPL];
                uint32_t t[PPL];
#pragma unroll
                for (int g = 0; g < PPL; ++g) {
                    const uint32_t c = (bytes[g][d >> 2] >> ((d & 3) * 8)) & 0xFFu;
                    t[g] = alive[g] ? (base[g] ^ c) : 0u;
                    if (ROOT && d == 0)
                        rec[g] = rootc[(P.root_base ^ c) & 255u];  // idle lanes read a valid entry too; `alive` masks it
                    else
                        rec[g] = trie[t[g]];
                }
#pragma unroll
                for (int g = 0; g < PPL; ++g)
                    if (!(ROOT && d == 0)) asm volatile("" : "+v"(rec[g].x), "+v"(rec[g].y), "+v"(rec[g].z), "+v"(rec[g].w));
#pragma unroll
                for (int g = 0; g < PPL; ++g) {
                    alive[g] = alive[g] && rec[g].x == cur[g];
                    if (alive[g]) {
                        cur[g] = t[g];
                        base[g] = rec[g].y & 0x7FFFFFFFu;
                        bool term = (rec[g].y >> 31) != 0u;
                        if (DROPOUT) {  // model.rs:100: kept iff len <= 1 || dropout < rand
                            if (term && d >= 1) term = P.dropout < dropout_u01(P.seed, s, pg[g], (uint32_t)d + 1u);
                        }
                        if (term) {
                            double* scw = reinterpret_cast<double*>(wbase + g * kRows4GroupBytes) + lane * LM;
                            const uint32_t col = ((uint32_t)d + l) & 15u;  // (len - 1 + row) & 15
                            m[g] |= 1u << d;
                            scw[col] = __hiloint2double((int)rec[g].w, (int)rec[g].z);
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        TGX_STAMP(2)  // trie walk
        {   // request the following block's text window now (vector loads return in order, so
            // this must come after the walk's last gather): it lands while the relax runs
            const uint32_t* __restrict__ np = wp + 4 * PPL;  // + SPAN bytes, same alignment
#pragma unroll
            for (int q = 0; q <= 4 * PPL; ++q) wn[q] = np[q];
        }

        // ---- relax: 16*PPL static steps, four positions (one per row) per step.  All 16 LDS
        // reads of a group are issued before the dependent chain starts.
        uint32_t fin[PPL];   // winner step of this lane's position
        bool reached[PPL];   // its best score is not -inf
#pragma unroll
        for (int g = 0; g < PPL; ++g) {
            fin[g] = kNoStep;
            reached[g] = false;
            if (P.flags & 2u) continue;
            // row (r*16 + U), column (l - 1) & 15: see relax4_step
            const double* scr = reinterpret_cast<const double*>(wbase + g * kRows4GroupBytes) + r * 256u + ((l - 1u) & 15u);
            double sv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) sv[u] = scr[u * 16];
            uint32_t fhi = 0xFFF00000u;
            relax5_step<0>(sv[0], acc, bpv, fin[g], fhi);
            relax5_step<1>(sv[1], acc, bpv, fin[g], fhi);
            relax5_step<2>(sv[2], acc, bpv, fin[g], fhi);
            relax5_step<3>(sv[3], acc, bpv, fin[g], fhi);
            relax5_step<4>(sv[4], acc, bpv, fin[g], fhi);
            relax5_step<5>(sv[5], acc, bpv, fin[g], fhi);
            relax5_step<6>(sv[6], acc, bpv, fin[g], fhi);
            relax5_step<7>(sv[7], acc, bpv, fin[g], fhi);
            relax5_step<8>(sv[8], acc, bpv, fin[g], fhi);
            relax5_step<9>(sv[9], acc, bpv, fin[g], fhi);
            relax5_step<10>(sv[10], acc, bpv, fin[g], fhi);
            relax5_step<11>(sv[11], acc, bpv, fin[g], fhi);
            relax5_step<12>(sv[12], acc, bpv, fin[g], fhi);
            relax5_step<13>(sv[13], acc, bpv, fin[g], fhi);
            relax5_step<14>(sv[14], acc, bpv, fin[g], fhi);
            relax5_step<15>(sv[15], acc, bpv, fin[g], fhi);
            reached[g] = fhi != 0xFFF00000u;  // high word of -inf
        }
        __builtin_amdgcn_wave_barrier();
        TGX_STAMP(3)  // relax

        // ---- back-pointers of the SPAN positions (streamed past the caches), next block.  Index j =
        // end position - 1; a lane's bytes of one 64-position group are packed into a dword (bp8_perm)
        uint8_t* const bpw = P.bp8 + bp8_base(beg, s);
#pragma unroll
        for (int g = 0; g < PPL; ++g)
            if (live && pg[g] >= 1u && pg[g] <= n) {
                // winner pushed at step U = fin into lane l: token length ((l - U - 1) & 15) + 1
                const uint32_t b = reached[g] ? ((l - fin[g] - 1u) & 15u) : 0xFFu;
                const uint32_t j = pg[g] - 1u, kq = (j >> 4) & 3u;
                pk = (kq == 0u) ? b : (pk | (b << (8u * kq)));
                pk_j = j;
                pk_dirty = true;
                if (kq == 3u) {
                    __builtin_nontemporal_store(pk, reinterpret_cast<uint32_t*>(bpw + (bp8_perm(j) & ~3u)));
                    pk_dirty = false;
                }
            }
        if (live && n - p0 < SPAN && pk_dirty) {  // the sample ends inside a group: flush the partial dword
            __builtin_nontemporal_store(pk, reinterpret_cast<uint32_t*>(bpw + (bp8_perm(pk_j) & ~3u)));
            pk_dirty = false;
        }
        if (live) {
            const uint32_t left = n - p0;
            if (left < SPAN) {  // position n lies in this iteration: the sample is done
#pragma unroll
                for (int g = 0; g < PPL; ++g)
                    if (left == 16u * g + l) P.status[s] = (n == 0u || reached[g]) ? 1u : 0u;
                need_new = true;
            } else {
                p0 += SPAN;
            }
        }
        TGX_STAMP(4)  // stores, bookkeeping
    }
    if (STAMP && lane == 0 && P.stamps) {
        unsigned long long* o = P.stamps + (size_t)(blockIdx.x * wpb + wave) * 8u;
        for (int i = 0; i < 5; ++i) o[i] = seg[i];
        o[5] = iters;
    }
}

// Back-trace + id emission for the rows4 path (trace_body.h).
template <bool STAMP, bool CARRY>
__global__ __launch_bounds__(256) void trace_kernel(EncodeParams P) {
    __shared__ typename TraceRingEntry<CARRY>::type ring_all[4][kTraceRing];
    trace_body<16, true, STAMP, CARRY>(P, ring_all[threadIdx.x >> 6]);
}

// counts[S] -> offsets[S+1] (exclusive prefix sum), one workgroup.
__global__ __launch_bounds__(1024) void scan_counts_kernel(const uint32_t* __restrict__ counts,
                                                           uint64_t* __restrict__ offsets,
                                                           uint64_t n) {
    __shared__ uint64_t wsum[16];
    __shared__ uint64_t carry_s;
    constexpr uint32_t E = 8;  // consecutive counts per thread and round: 8192 per round
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (uint64_t base = 0; base < n; base += 1024u * E) {
        const uint64_t i0 = base + (uint64_t)tid * E;
        uint32_t c[E];
        uint64_t v = 0;
#pragma unroll
        for (uint32_t e = 0; e < E; ++e) {
            c[e] = (i0 + e < n) ? counts[i0 + e] : 0u;
            v += c[e];
        }
        uint64_t x = v;
        for (int off = 1; off < 64; off <<= 1) {
            uint64_t y = __shfl_up(x, off);
            if ((int)lane >= off) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        uint64_t wprefix = 0;
        for (uint32_t w = 0; w < wave; ++w) wprefix += wsum[w];
        const uint64_t carry = carry_s;
        uint64_t run = carry + wprefix + x - v;
#pragma unroll
        for (uint32_t e = 0; e < E; ++e) {
            if (i0 + e < n) offsets[i0 + e] = run;
            run += c[e];
        }
        __syncthreads();
        if (tid == 1023) carry_s = run;
        __syncthreads();
    }
    if (tid == 0) offsets[n] = carry_s;
}

// tmp (right-aligned per sample) -> ids (packed).  One wave per sample, or — ROWS: batches of short samples, a few
// dozen ids each — one 16-lane row per sample (four samples per wave; neighbours in the longest-first order have
// similar lengths).
template <bool ROWS>
__global__ __launch_bounds__(256) void compact_kernel(CompactParams P) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t sub = ROWS ? (lane & 15u) : lane, width = ROWS ? 16u : 64u;
    const uint64_t n_units = (uint64_t)gridDim.x * 4u * (ROWS ? 4u : 1u);
    const uint32_t wave_id = blockIdx.x * 4u + (threadIdx.x >> 6);
    for (uint64_t k = ROWS ? (uint64_t)wave_id * 4u + (lane >> 4) : (uint64_t)wave_id; k < P.n_samples; k += n_units) {
        // (rows: samples of a few dozen ids — any order does, and the samples' own saves a dependent load per sample and
        // makes the rows of a wave read neighbouring offsets: 1.39 -> ms per GiB of 140-byte samples)
        const uint32_t s = ROWS ? (uint32_t)k : P.order[k];
        const uint64_t end = P.offs[s + 1];
        const uint64_t o0 = P.out_offs[s];
        const uint32_t cnt = (uint32_t)(P.out_offs[s + 1] - o0);
        const uint32_t* __restrict__ src = P.tmp + (end - cnt);
        uint32_t* __restrict__ dst = P.ids + o0;
        // 16 bytes per lane (neither side is 16-byte aligned: unaligned dwordx4), the last 0..3 ids one by one
        struct __attribute__((packed, aligned(4))) Ids4 { uint32_t w[4]; };
        const uint32_t body = cnt & ~3u;
        for (uint32_t i = sub * 4u; i < body; i += width * 4u)
            *reinterpret_cast<Ids4*>(dst + i) = *reinterpret_cast<const Ids4*>(src + i);
        if (body + sub < cnt) dst[body + sub] = src[body + sub];
    }
}

// ---- launchers -------------------------------------------------------------

// kernel specialisation for a (rounded) max token length: fast paths need LM == LMT
static int lmt_for(uint32_t lm) { return lm <= 16u ? 16 : (lm <= 32u ? 32 : 0); }
uint32_t encode_effective_lm(uint32_t lm) {
    const int t = lmt_for(lm);
    return t ? (uint32_t)t : lm;
}

// Waves per block: as many (<= 4) as keep one block's LDS within 64 KiB.
uint32_t encode_waves_per_block(uint32_t lm) {
    uint32_t w = (64u * 1024u) / wave_lds_bytes(encode_effective_lm(lm));
    return w < 1u ? 1u : (w > 4u ? 4u : w);
}
uint32_t encode_lds_bytes_per_block(uint32_t lm) {
    return encode_waves_per_block(lm) * wave_lds_bytes(encode_effective_lm(lm));
}

typedef void (*encode_fn)(EncodeParams);
static encode_fn pick_kernel(uint32_t lm) {
    const int t = lmt_for(lm);
    if (t == 16) return encode_kernel<16>;
    if (t == 32) return encode_kernel<32>;
    return encode_kernel<0>;
}

hipError_t launch_encode(const EncodeParams& p, uint32_t blocks, hipStream_t stream) {
    const uint32_t lds = encode_lds_bytes_per_block(p.lm);
    const dim3 block(64u * encode_waves_per_block(p.lm));
    hipLaunchKernelGGL(pick_kernel(p.lm), dim3(blocks), block, lds, stream, p);
    return hipGetLastError();
}

hipError_t encode_max_blocks_per_cu(uint32_t lm, int* out) {
    const uint32_t lds = encode_lds_bytes_per_block(lm);
    const int threads = (int)(64u * encode_waves_per_block(lm));
    if (lds > 160u * 1024u) return hipErrorInvalidValue;
    encode_fn fn = pick_kernel(lm);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(out, fn, threads, lds);
}

// four-samples-per-wave path (max token length <= 16): one block per CU
typedef void (*encode4_fn)(EncodeParams);
static encode4_fn pick_encode4(bool dropout, int ppl, bool stamp, bool root) {
    if (stamp) return encode4_kernel<false, 1, true, false>;
    if (root) return dropout ? encode4_kernel<true, 1, false, true> : encode4_kernel<false, 1, false, true>;
    if (ppl == 1) return dropout ? encode4_kernel<true, 1, false, false> : encode4_kernel<false, 1, false, false>;
    if (ppl == 2) return dropout ? encode4_kernel<true, 2, false, false> : encode4_kernel<false, 2, false, false>;
    return dropout ? encode4_kernel<true, 4, false, false> : encode4_kernel<false, 4, false, false>;
}
uint32_t encode4_group_bytes() { return kRows4GroupBytes; }
uint32_t encode4_lds_bytes(int waves, int ppl, bool root) {
    return (root ? 4096u : 0u) + (uint32_t)waves * (uint32_t)ppl * kRows4GroupBytes;
}
// root = true (first trie level in LDS) exists for ppl == 1 only
// Waves of this variant that one SIMD can hold by its vector registers (512 per lane and SIMD, allocated
// in units of 8, at most 8 waves): what the host checks its geometry against, so that a register-count
// surprise cannot silently halve the occupancy.  (hipOccupancyMaxActiveBlocksPerMultiprocessor is of no
// use here: it does not know the CU's 160 KiB of LDS.)
hipError_t encode4_waves_per_simd(bool dropout, int ppl, bool root, int* out) {
    root = root && ppl == 1;
    hipFuncAttributes attr;
    hipError_t e = hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(pick_encode4(dropout, ppl, false, root)));
    if (e != hipSuccess) return e;
    const int regs = (attr.numRegs + 7) & ~7;
    *out = regs > 0 ? (512 / regs > 8 ? 8 : 512 / regs) : 8;
    return hipSuccess;
}
hipError_t launch_encode4(const EncodeParams& p, int ppl, int waves, uint32_t blocks, bool root, hipStream_t stream) {
    const bool stamp = p.stamps != nullptr;
    root = root && !stamp && ppl == 1;
    encode4_fn fn = pick_encode4(p.dropout > 0.0, ppl, stamp, root);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(64u * (uint32_t)waves), encode4_lds_bytes(waves, stamp ? 1 : ppl, root), stream, p);
    return hipGetLastError();
}
hipError_t launch_trace(const EncodeParams& p, uint32_t blocks, hipStream_t stream) {
    if (p.stamps)
        hipLaunchKernelGGL((trace_kernel<true, false>), dim3(blocks), dim3(256), 0, stream, p);
    else if (p.trace_carry)
        hipLaunchKernelGGL((trace_kernel<false, true>), dim3(blocks), dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL((trace_kernel<false, false>), dim3(blocks), dim3(256), 0, stream, p);
    return hipGetLastError();
}

// counts[n] -> offsets[n + 1].  One workgroup up to 64 K samples; a device-wide scan (rocPRIM) beyond: at 7 M
// samples (1 GiB of samples <= 256 bytes) the single workgroup took 7.1 ms of a 39 ms pass.  `temp` must hold
// scan_temp_bytes(n) bytes; counts must have room for one more entry (it is zeroed and scanned, so that
// offsets[n] is the total).
struct WidenU32 {
    __host__ __device__ uint64_t operator()(uint32_t x) const { return (uint64_t)x; }
};
hipError_t scan_temp_bytes(uint64_t n, size_t* bytes) {
    *bytes = 0;
    if (n <= 65536) return hipSuccess;
    auto in = rocprim::make_transform_iterator((const uint32_t*)nullptr, WidenU32());
    return rocprim::exclusive_scan(nullptr, *bytes, in, (uint64_t*)nullptr, (uint64_t)0, (size_t)(n + 1), rocprim::plus<uint64_t>());
}
hipError_t launch_scan(uint32_t* counts, uint64_t* offsets, uint64_t n, void* temp, size_t temp_bytes, hipStream_t stream) {
    if (n <= 65536 || !temp) {
        hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(1024), 0, stream, counts, offsets, n);
        return hipGetLastError();
    }
    hipError_t e = hipMemsetAsync(counts + n, 0, 4, stream);
    if (e != hipSuccess) return e;
    auto in = rocprim::make_transform_iterator((const uint32_t*)counts, WidenU32());
    return rocprim::exclusive_scan(temp, temp_bytes, in, offsets, (uint64_t)0, (size_t)(n + 1), rocprim::plus<uint64_t>(), stream);
}

hipError_t launch_compact(const CompactParams& p, uint32_t blocks, bool rows, hipStream_t stream) {
    if (rows)
        hipLaunchKernelGGL(compact_kernel<true>, dim3(blocks), dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL(compact_kernel<false>, dim3(blocks), dim3(256), 0, stream, p);
    return hipGetLastError();
}

}  // namespace tgx
