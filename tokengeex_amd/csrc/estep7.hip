// estep7_kernel (gfx950 / CDNA4, wave64) — round 4: the E-step (src/lattice.rs:245-312 populate_marginal over the
// nodes of src/model.rs:34-55 populate_nodes, per snippet of src/prune.rs:64-120) with ONE trie walk per position.
//
// Rounds 1 - 3 ran the reference's three loops as two chained kernels: a forward sweep that writes alpha[] to HBM
// (estep5_fwd_kernel / estep4l_fwd_kernel) and a backward sweep on the reversed text over a second double-array of
// the REVERSED tokens that also adds every match's marginal (estep4l_bwd_kernel: 36.5 of the pass's 49 ms per GiB —
// a second walk, 28 vector instructions per step, 888 M memory-side f64 atomics).  Round 3 found that the lattice
// FACTORISES at a position no token match crosses (cuts.hip) and used it to cut long snippets into pieces of ~2 KiB.
// Such positions are 4.5 bytes apart on average — so this kernel makes every TRIP of a row (16 PPL positions) a
// lattice of its own:
//
//   * a row walks the 16 PPL positions from its current cut p0 (the walk of encode5_kernel: 8-byte label-checked
//     records, staggered walks, match entries in LDS) and remembers every position's longest match;
//   * the last position E in (p0, p0 + 16 PPL] that no match crosses and some match ends at (the criterion of
//     cuts.hip) closes the trip: [p0, E) is an independent lattice — alpha[p0] = 1, beta[E] = 1, Z = alpha[E];
//   * forward steps, backward steps and the marginals a[p] w b[q] / Z of all its matches run from the SAME match
//     entries in LDS: nothing goes to HBM but the expected counts, no reversed trie exists, and no chain runs
//     through a piece (a trip depends on the one before only through its start);
//   * the row goes on at E; the positions after E were walked in vain (~6 % at 64 positions per trip).
// log Z of a piece is the sum of its trips', expected counts add (src/prune.rs:99-117 only ever sums them).
//
// Match entries are TOKEN RANKS (Trie8T, trie_build.h: tokens ranked by exp(score) / length, i.e. by how often they
// are expected to match), 16 bits for vocabularies of at most 65 535 tokens and 32 bits beyond (the 500 000-entry
// stages of prune).  The first n_hot ranks have {sum, w = exp(score)} in the block's LDS: their weights are read and
// their marginals added there (ds_add_f64); the others read w from the table in L2 and add to HBM (~9 % of the
// matches with 6 000 ranks in LDS on the bench corpus; the slot-ranked table of estep4l_bwd_kernel left 26 % out).
//
// A trip WITHOUT such a position (a run of blanks under a vocabulary with long blank tokens; 1 trip in ~10 000 on
// the bench corpus) hands the rest of its piece to the redo list: the host runs the chained kernels on exactly those
// stretches.  A position nothing reaches (lattice.rs:255's corner, a value out of the f64 range) raises range_flag
// like the other linear-domain kernels, and the pass is redone in the log domain.
//
// Linear domain with exact power-of-two rescaling per group of 16 positions, as estep4l.hip: same products in the
// same order within a trip.  Tokens of at most 16 bytes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <type_traits>

#include "device_common.h"
#include "kernels.h"

namespace tgx {

template <bool WIDE>
struct E7Lay {
    using ET = typename std::conditional<WIDE, uint32_t, uint16_t>::type;
    static constexpr uint32_t ES = WIDE ? 4u : 2u;  // bytes per entry
    static constexpr uint32_t CS = 16u * ES;        // one column: 16 start positions
    static constexpr uint32_t RS = 16u * CS;        // one row (sample): 16 columns
    static constexpr uint32_t GB = 4u * RS;         // one group of 16 positions of a wave's four rows
};

struct Walk7Ctx {
    __amdgpu_buffer_rsrc_t trie;  // Trie8TRec[], byte offsets
    uint32_t smp;                 // sample of the row's piece (dropout hash)
    uint32_t lcs;                 // l * CS, opaque per trip
    double dropout;
    uint64_t seed;
};

// The PPL staggered walks of a lane (encode5.hip: Walk5) over records {base | label << 24, token rank}: the child by byte
// c of a node is the record at byte offset 8 ((rec ^ c) & 0xFFFFFF), valid iff its label is c.  A match stores its
// token's rank at (start, length) of the group's entry buffer and is the walk's longest so far.
// OVF (16-bit entries for a vocabulary of a few more than 65 535 tokens): `seen` collects the ranks of the matches — a rank
// beyond the entries' range shows there after the walk and the trip is handed to the redo kernel (32-bit entries).
template <bool DROPOUT, bool WIDE, int PPL, int D, bool OVF = false>
struct Walk7 {
    using L = E7Lay<WIDE>;
    static __device__ __forceinline__ void run(const Walk7Ctx& W, const uint32_t (&bytes)[PPL][4], const uint32_t (&maxd)[PPL],
                                               const uint32_t (&pgh)[PPL], const uint32_t (&wlane)[PPL], bool (&alive)[PPL],
                                               uint2 (&rec)[PPL], uint32_t (&c)[PPL], uint32_t (&mlen)[PPL], uint32_t& seen) {
        bool any = false;
#pragma unroll
        for (int g = 0; g < PPL; ++g) {
            asm("" : "+v"(rec[g].x), "+v"(rec[g].y));
            alive[g] = alive[g] && ((rec[g].x >> 24) == c[g]);
            const uint32_t tok = rec[g].y;  // 0: no token ends here
            bool term = alive[g] && tok != 0u;
            if (DROPOUT) {  // model.rs:48: a node is skipped iff len > 1 && rand < dropout
                if (term && D >= 1) term = !(dropout_u01(W.seed, W.smp, pgh[g], (uint32_t)D + 1u) < W.dropout);
            }
            if (term) {
                lds_st<typename L::ET>(wlane[g] | ((W.lcs + L::CS * (uint32_t)D) & (15u * L::CS)), (typename L::ET)tok);
                mlen[g] = (uint32_t)D + 1u;
                if (OVF) seen = max(seen, tok);
            }
            if (D + 1 < 16) {
                constexpr int e = D + 1 < 16 ? D + 1 : 15;
                c[g] = (bytes[g][e >> 2] >> ((e & 3) * 8)) & 0xFFu;
                alive[g] = alive[g] && ((uint32_t)e < maxd[g]);
                const uint32_t off = ((rec[g].x ^ c[g]) << 8) >> 5;  // 8 * ((rec ^ c) & 0xFFFFFF)
                rec[g] = buf_ld8(W.trie, alive[g] ? off : 0u);
            } else {
                alive[g] = false;
            }
            any = any || alive[g];
        }
        if (__builtin_amdgcn_ballot_w64(any) == 0) return;
        Walk7<DROPOUT, WIDE, PPL, D + 1, OVF>::run(W, bytes, maxd, pgh, wlane, alive, rec, c, mlen, seen);
    }
};
template <bool DROPOUT, bool WIDE, int PPL, bool OVF>
struct Walk7<DROPOUT, WIDE, PPL, 16, OVF> {
    static __device__ __forceinline__ void run(const Walk7Ctx&, const uint32_t (&)[PPL][4], const uint32_t (&)[PPL], const uint32_t (&)[PPL],
                                               const uint32_t (&)[PPL], bool (&)[PPL], uint2 (&)[PPL], uint32_t (&)[PPL], uint32_t (&)[PPL], uint32_t&) {}
};

// A match entry in registers is the LDS byte offset `ha` = 16 * rank of its token's {sum, w} (entry 0 = {0, 0}: "no
// token"): w is read at ha + 8, the marginal added at ha, both without further address arithmetic.  Ranks beyond the
// block's table (ha > hot_lim = 16 n_hot; COLD builds) read w from the table in L2 — the LDS read is unconditional at a
// clamped address and exists before the conditional load, as in e5_scores_cold — and add to HBM.
template <bool COLD, int FIRST>
__device__ __forceinline__ void e7_weights4(__amdgpu_buffer_rsrc_t wtab, uint32_t hot_lim, const uint32_t (&ha)[16], double (&sv)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) sv[u] = lds_ld<double>(((!COLD || ha[FIRST + u] <= hot_lim) ? ha[FIRST + u] : 0u) + 8u);
    if (COLD) {
        asm volatile("" : "+v"(sv[0]), "+v"(sv[1]), "+v"(sv[2]), "+v"(sv[3]));
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (ha[FIRST + u] > hot_lim) sv[u] = buf_ld_f64(wtab, ha[FIRST + u] >> 1);
    }
}

__device__ __forceinline__ double sel0_f64(uint64_t mask, double v) {  // mask ? 0.0 : v
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = sel_imm_u32<0>(mask, (uint32_t)b), hi = sel_imm_u32<0>(mask, (uint32_t)(b >> 32));
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

// Forward step U: a[x] of the start position x = (group, U) is final (captured in `fin`); every lane L = end position
// adds a[x] w for its match of length L - x.  The finalised lane restarts from zero, so the value path of a step is
// broadcast -> fused multiply-add (e4l_fwd_step: broadcast -> multiply -> add -> select): the steps are a dependent
// chain, and a trip's 64 forward + 65 backward steps are two thirds of its time.
template <int U>
__device__ __forceinline__ void e7_fwd_step(double sv, double& acc, double& fin) {
    constexpr uint64_t MU = kRowLane0 << U;
    fin = sel_f64(MU, acc, fin);
    const double best = row_bcast_f64<U>(acc);
    const double acc_r = sel0_f64(MU, acc);
    acc = __builtin_fma(best, sv, acc_r);
}

// Backward step U (descending): b[x] of the end position x = (group, U) is final; every lane P = start position takes
// cand = w b[x] for its match of length x - P (lanes P >= U: the start one group below, length U - P + 16) and adds the
// match's marginal a[P] w b[x] / Z (lattice.rs:305-307) to its token's sum.  cs: a[P] / Z of the lane's start position,
// scaled by the power of two that the exponents of a, b and Z leave — lane U switches to the position one group below
// at its own step (cs_low), so the steps need the sixteen one-lane masks only (with a second family of sixteen masks
// "lanes >= U" the kernel's 64-bit constants no longer fit the scalar registers: 1 700 v_readlane of spilled masks).
template <int U, bool COLD>
__device__ __forceinline__ void e7_bwd_step(double sv, uint32_t ha, double& cs, double cs_low, uint32_t hot_lim,
                                            __attribute__((address_space(1))) double* expected, double& acc) {
    constexpr uint64_t MU = kRowLane0 << U;
    const double best = row_bcast_f64<U>(acc);
    const double acc_r = sel0_f64(MU, acc);
    const double cand = best * sv;
    acc = __builtin_fma(best, sv, acc_r);
    cs = sel_f64(MU, cs_low, cs);
    // (COLD builds: the comparison is made HERE — kept from the weight loads, its sixteen lane masks occupy 32 scalar
    // registers through the group's steps and the walk's buffer resources end up in spilled lanes)
    if (COLD) asm volatile("" : "+v"(ha));
    if (cand != 0.0) {
        const double mg = cand * cs;
        if (!COLD || ha <= hot_lim)
            __hip_atomic_fetch_add((__attribute__((address_space(3))) double*)(uintptr_t)ha, mg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else
            __hip_atomic_fetch_add((__attribute__((address_space(1))) double*)((__attribute__((address_space(1))) char*)expected + (ha >> 1)), mg,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// the sixteen entries a START lane needs for the steps of one group, read in descending step order: the lane's address
// switches from this group's buffer to the one below at its own step (entries (start P, end U): column (U - 1) & 15)
template <bool WIDE, int U>
struct E7BwdEntries {
    static __device__ __forceinline__ void run(uint32_t& va, uint32_t vb_low, uint32_t (&ha)[16]) {
        using L = E7Lay<WIDE>;
        constexpr uint64_t MU = kRowLane0 << U;
        va = sel_u32(MU, vb_low, va);
        ha[U] = (uint32_t)lds_ld<typename L::ET>(va + (uint32_t)((U - 1) & 15) * L::CS) << 4;
        E7BwdEntries<WIDE, U - 1>::run(va, vb_low, ha);
    }
};
template <bool WIDE>
struct E7BwdEntries<WIDE, -1> {
    static __device__ __forceinline__ void run(uint32_t&, uint32_t, uint32_t (&)[16]) {}
};

// the sixteen entries an END lane needs for the forward steps of one group: its column, 16 starts x one entry
template <bool WIDE>
__device__ __forceinline__ void e7_fwd_entries(uint32_t col, uint32_t (&ha)[16]) {
    if (WIDE) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const u32x4_t t = lds_ld<u32x4_t>(col + 16u * q);
            ha[4 * q] = t.x << 4; ha[4 * q + 1] = t.y << 4; ha[4 * q + 2] = t.z << 4; ha[4 * q + 3] = t.w << 4;
        }
    } else {
        const u32x4_t ia = lds_ld<u32x4_t>(col), ib = lds_ld<u32x4_t>(col + 16u);
        const uint32_t iw[8] = {ia.x, ia.y, ia.z, ia.w, ib.x, ib.y, ib.z, ib.w};
#pragma unroll
        for (int u = 0; u < 16; ++u) ha[u] = (u & 1) ? ((iw[u >> 1] >> 16) << 4) : ((iw[u >> 1] & 0xFFFFu) << 4);
    }
}

#ifndef TGX_E7_QUARTERS
#define TGX_E7_QUARTERS 0
#endif
#if TGX_E7_QUARTERS
// the sixteen steps of a group with their weights a quarter group at a time, requested a quarter ahead of the steps
// that use them (all sixteen at once keep 32 more registers live through the steps)
#define E7_FWD_GROUP(wtab, hot_lim, ha, acc, fin)                                                                                   \
    {                                                                                                                               \
        double sa[4], sb[4];                                                                                                        \
        e7_weights4<COLD, 0>(wtab, hot_lim, ha, sa);                                                                                \
        e7_weights4<COLD, 4>(wtab, hot_lim, ha, sb);                                                                                \
        e7_fwd_step<0>(sa[0], acc, fin); e7_fwd_step<1>(sa[1], acc, fin); e7_fwd_step<2>(sa[2], acc, fin); e7_fwd_step<3>(sa[3], acc, fin);     \
        e7_weights4<COLD, 8>(wtab, hot_lim, ha, sa);                                                                                \
        e7_fwd_step<4>(sb[0], acc, fin); e7_fwd_step<5>(sb[1], acc, fin); e7_fwd_step<6>(sb[2], acc, fin); e7_fwd_step<7>(sb[3], acc, fin);     \
        e7_weights4<COLD, 12>(wtab, hot_lim, ha, sb);                                                                               \
        e7_fwd_step<8>(sa[0], acc, fin); e7_fwd_step<9>(sa[1], acc, fin); e7_fwd_step<10>(sa[2], acc, fin); e7_fwd_step<11>(sa[3], acc, fin);   \
        e7_fwd_step<12>(sb[0], acc, fin); e7_fwd_step<13>(sb[1], acc, fin); e7_fwd_step<14>(sb[2], acc, fin); e7_fwd_step<15>(sb[3], acc, fin); \
    }
#define E7_BWD_GROUP(wtab, hot_lim, ha, cs, cs_low, expected, accb)                                                                 \
    {                                                                                                                               \
        double sa[4], sb[4];                                                                                                        \
        e7_weights4<COLD, 12>(wtab, hot_lim, ha, sa);                                                                               \
        e7_weights4<COLD, 8>(wtab, hot_lim, ha, sb);                                                                                \
        e7_bwd_step<15, COLD>(sa[3], ha[15], cs, cs_low, hot_lim, expected, accb); e7_bwd_step<14, COLD>(sa[2], ha[14], cs, cs_low, hot_lim, expected, accb); \
        e7_bwd_step<13, COLD>(sa[1], ha[13], cs, cs_low, hot_lim, expected, accb); e7_bwd_step<12, COLD>(sa[0], ha[12], cs, cs_low, hot_lim, expected, accb); \
        e7_weights4<COLD, 4>(wtab, hot_lim, ha, sa);                                                                                \
        e7_bwd_step<11, COLD>(sb[3], ha[11], cs, cs_low, hot_lim, expected, accb); e7_bwd_step<10, COLD>(sb[2], ha[10], cs, cs_low, hot_lim, expected, accb); \
        e7_bwd_step<9, COLD>(sb[1], ha[9], cs, cs_low, hot_lim, expected, accb); e7_bwd_step<8, COLD>(sb[0], ha[8], cs, cs_low, hot_lim, expected, accb);     \
        e7_weights4<COLD, 0>(wtab, hot_lim, ha, sb);                                                                                \
        e7_bwd_step<7, COLD>(sa[3], ha[7], cs, cs_low, hot_lim, expected, accb); e7_bwd_step<6, COLD>(sa[2], ha[6], cs, cs_low, hot_lim, expected, accb);     \
        e7_bwd_step<5, COLD>(sa[1], ha[5], cs, cs_low, hot_lim, expected, accb); e7_bwd_step<4, COLD>(sa[0], ha[4], cs, cs_low, hot_lim, expected, accb);     \
        e7_bwd_step<3, COLD>(sb[3], ha[3], cs, cs_low, hot_lim, expected, accb); e7_bwd_step<2, COLD>(sb[2], ha[2], cs, cs_low, hot_lim, expected, accb);     \
        e7_bwd_step<1, COLD>(sb[1], ha[1], cs, cs_low, hot_lim, expected, accb); e7_bwd_step<0, COLD>(sb[0], ha[0], cs, cs_low, hot_lim, expected, accb);     \
    }

#else
// the sixteen steps of a group; all sixteen weights are requested before the first step: one exposed round trip per
// group (requested a quarter group at a time they keep 32 fewer registers live — and every quarter waits for its own
// loads, COLD builds for an L2 round trip: 32.5 -> 34.0 ms per GiB)
template <bool COLD>
__device__ __forceinline__ void e7_weights(__amdgpu_buffer_rsrc_t wtab, uint32_t hot_lim, const uint32_t (&ha)[16], double (&sv)[16]) {
#pragma unroll
    for (int u = 0; u < 16; ++u) sv[u] = lds_ld<double>(((!COLD || ha[u] <= hot_lim) ? ha[u] : 0u) + 8u);
    if (COLD) {
        asm volatile("" : "+v"(sv[0]), "+v"(sv[1]), "+v"(sv[2]), "+v"(sv[3]), "+v"(sv[4]), "+v"(sv[5]), "+v"(sv[6]), "+v"(sv[7]));
        asm volatile("" : "+v"(sv[8]), "+v"(sv[9]), "+v"(sv[10]), "+v"(sv[11]), "+v"(sv[12]), "+v"(sv[13]), "+v"(sv[14]), "+v"(sv[15]));
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (ha[u] > hot_lim) sv[u] = buf_ld_f64(wtab, ha[u] >> 1);
    }
}
#define E7_FWD_GROUP(wtab, hot_lim, ha, acc, fin)                                                                                   \
    {                                                                                                                               \
        double sv[16];                                                                                                              \
        e7_weights<COLD>(wtab, hot_lim, ha, sv);                                                                                    \
        e7_fwd_step<0>(sv[0], acc, fin); e7_fwd_step<1>(sv[1], acc, fin); e7_fwd_step<2>(sv[2], acc, fin); e7_fwd_step<3>(sv[3], acc, fin);     \
        e7_fwd_step<4>(sv[4], acc, fin); e7_fwd_step<5>(sv[5], acc, fin); e7_fwd_step<6>(sv[6], acc, fin); e7_fwd_step<7>(sv[7], acc, fin);     \
        e7_fwd_step<8>(sv[8], acc, fin); e7_fwd_step<9>(sv[9], acc, fin); e7_fwd_step<10>(sv[10], acc, fin); e7_fwd_step<11>(sv[11], acc, fin); \
        e7_fwd_step<12>(sv[12], acc, fin); e7_fwd_step<13>(sv[13], acc, fin); e7_fwd_step<14>(sv[14], acc, fin); e7_fwd_step<15>(sv[15], acc, fin); \
    }
#define E7_BWD_GROUP(wtab, hot_lim, ha, cs, cs_low, expected, accb)                                                                 \
    {                                                                                                                               \
        double sv[16];                                                                                                              \
        e7_weights<COLD>(wtab, hot_lim, ha, sv);                                                                                    \
        e7_bwd_step<15, COLD>(sv[15], ha[15], cs, cs_low, hot_lim, expected, accb); e7_bwd_step<14, COLD>(sv[14], ha[14], cs, cs_low, hot_lim, expected, accb); \
        e7_bwd_step<13, COLD>(sv[13], ha[13], cs, cs_low, hot_lim, expected, accb); e7_bwd_step<12, COLD>(sv[12], ha[12], cs, cs_low, hot_lim, expected, accb); \
        e7_bwd_step<11, COLD>(sv[11], ha[11], cs, cs_low, hot_lim, expected, accb); e7_bwd_step<10, COLD>(sv[10], ha[10], cs, cs_low, hot_lim, expected, accb); \
        e7_bwd_step<9, COLD>(sv[9], ha[9], cs, cs_low, hot_lim, expected, accb); e7_bwd_step<8, COLD>(sv[8], ha[8], cs, cs_low, hot_lim, expected, accb);     \
        e7_bwd_step<7, COLD>(sv[7], ha[7], cs, cs_low, hot_lim, expected, accb); e7_bwd_step<6, COLD>(sv[6], ha[6], cs, cs_low, hot_lim, expected, accb);     \
        e7_bwd_step<5, COLD>(sv[5], ha[5], cs, cs_low, hot_lim, expected, accb); e7_bwd_step<4, COLD>(sv[4], ha[4], cs, cs_low, hot_lim, expected, accb);     \
        e7_bwd_step<3, COLD>(sv[3], ha[3], cs, cs_low, hot_lim, expected, accb); e7_bwd_step<2, COLD>(sv[2], ha[2], cs, cs_low, hot_lim, expected, accb);     \
        e7_bwd_step<1, COLD>(sv[1], ha[1], cs, cs_low, hot_lim, expected, accb); e7_bwd_step<0, COLD>(sv[0], ha[0], cs, cs_low, hot_lim, expected, accb);     \
    }
#endif

// TGX_STAMPS=7 (diagnostic runs only): s_memtime stamps around the phases of a trip, summed per wave
#ifdef TGX_E7_STAMPS
#define E7_STAMP(i)                                                    \
    if (P.stamps) {                                                    \
        __builtin_amdgcn_sched_barrier(0);                             \
        const uint64_t _now = (uint64_t)__builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_s_waitcnt(0xC07F);                            \
        __builtin_amdgcn_sched_barrier(0);                             \
        seg[i] += _now - t_last;                                       \
        t_last = _now;                                                 \
    }
#else
#define E7_STAMP(i)
#endif

template <bool DROPOUT, bool COLD, bool WIDE, int PPL, bool OVF = false>
__global__ __launch_bounds__(768) void estep7_kernel(Estep7Params P) {
    using L = E7Lay<WIDE>;
    using ET = typename L::ET;
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t l = lane & 15u, r = lane >> 4;
    const uint32_t wave = threadIdx.x >> 6;
    const uint2* __restrict__ trie = reinterpret_cast<const uint2*>(P.trie8t);
    const __amdgpu_buffer_rsrc_t trie_b = make_rsrc(P.trie8t, P.n_slots * 8u);
    const __amdgpu_buffer_rsrc_t wtab_b = make_rsrc(P.wtab, (P.n_tok + 1u) * 8u);
    const Estep7Work& Wk = *P.work;  // read where it is used (once per piece or less often), not kept in registers
    // (the dynamic LDS starts at LDS address 0 — the kernel has no static LDS —: as a constant it drops an add per LDS access)
    constexpr uint32_t lds0 = 0u;
    const uint32_t n_hot = P.n_hot;
    // the pointer of the cold-path atomics in a vector register pair: read from the parameter block, every use reloaded the
    // block's 16-register tuple from spilled lanes (16 v_readlane per step)
    typedef __attribute__((address_space(1))) double gdouble;
    gdouble* expected = (gdouble*)P.expected;
    asm volatile("" : "+v"(expected));  // (a vector pair: scalar registers are what this kernel runs out of)
    // ---- LDS: [0, 16 (n_hot + 1)) {sum, w} by token rank | root records | one all-zero group | match entries
    {
        double2* hot = reinterpret_cast<double2*>(smem);
        for (uint32_t i = threadIdx.x; i <= n_hot; i += blockDim.x) hot[i] = make_double2(0.0, P.wtab[i]);
        uint2* rw = reinterpret_cast<uint2*>(smem + P.root_off);
        for (uint32_t i = threadIdx.x; i < 256u; i += blockDim.x) rw[i] = trie[(P.root_base & ~255u) + i];
        uint4* zg = reinterpret_cast<uint4*>(smem + P.zero_off);
        for (uint32_t i = threadIdx.x; i < L::GB / 16u; i += blockDim.x) zg[i] = make_uint4(0, 0, 0, 0);
        __syncthreads();
    }
    const uint2* rootc = reinterpret_cast<const uint2*>(smem + P.root_off);
    const uint32_t wbase_off = lds0 + P.idx_off + wave * (PPL * L::GB);
    const uint32_t wr_off = wbase_off + r * L::RS + l * L::ES;                   // this lane's entries as a walker (column bits clear) ...
    const uint32_t col_off = wbase_off + r * L::RS + ((l - 1u) & 15u) * L::CS;   // ... its column as the END lane of the forward steps ...
    const uint32_t zero_rd = lds0 + P.zero_off + r * L::RS + l * L::ES;          // ... and "the group below group 0" of the backward steps
    unsigned char* const wbase = smem + P.idx_off + (size_t)wave * (PPL * L::GB);

    const uint32_t hot_lim = n_hot << 4;  // LDS byte offset of the last rank's entry
    uint32_t s = 0, n = 0, p0 = 0, smp = 0;
    uint64_t beg = 0, sbase = 0;
    bool live = false, need_new = true;
    // log Z of the row's piece = log of the product of its trips' Z: kept as mantissa x 2^exponent, one log per piece
    double zm = 1.0, zsum = 0.0;
    int ze = 0;
    // probe mode: a trip found no cut — the row walks on (16 positions back, so that every start that can cross a
    // candidate is in the trip) until it finds one; [redo_p0, that cut) goes to the redo list, the row goes on behind it
    bool probe = false;
    uint32_t redo_p0 = 0;
    uint64_t local_next = 0, local_end = 0;
#ifdef TGX_E7_STAMPS
    uint64_t seg[6] = {0, 0, 0, 0, 0, 0};
    uint64_t t_last = P.stamps ? (uint64_t)__builtin_amdgcn_s_memtime() : 0;
    uint32_t trips = 0;
#endif
    for (;;) {
        {
            uint64_t k = ~0ull;
            const bool have_local = need_new && local_next < local_end;
            if (have_local) k = local_next++;
            unsigned long long* const queue = reinterpret_cast<unsigned long long*>(first_u64((uint64_t)reinterpret_cast<uintptr_t>(Wk.queue)));
            const uint64_t kc = claim_rows_chunk(queue, need_new && !have_local, r, P.claim_chunk);
            if (need_new && !have_local) {
                k = kc;
                local_next = kc + 1u;
                local_end = kc + P.claim_chunk;
            }
            if (need_new) {
                live = k < Wk.n_snips;
                if (live) {
                    s = Wk.order[k];
                    beg = Wk.soffs[s];
                    n = (uint32_t)(Wk.soffs[s + 1] - beg);
                    if (DROPOUT) {
                        smp = Wk.snip_sample[s];
                        sbase = Wk.snip_base[s];
                    }
                }
                p0 = 0;
                zm = 1.0;
                ze = 0;
                probe = false;
            }
        }
        need_new = false;
        if (__builtin_amdgcn_ballot_w64(live) == 0) break;
#ifdef TGX_E7_STAMPS
        trips++;
#endif
        E7_STAMP(0)  // claiming pieces

        // ---- text window of this trip, reset of this lane's columns
        const uintptr_t addr = reinterpret_cast<uintptr_t>(P.text + (live ? beg + p0 + l : 0));
        const uint32_t sh = (uint32_t)(addr & 3u);
        const uint32_t* __restrict__ wp = reinterpret_cast<const uint32_t*>(addr & ~uintptr_t(3));
        uint32_t w[4 * PPL + 1];
#pragma unroll
        for (int q = 0; q <= 4 * PPL; ++q) w[q] = wp[q];
        uint32_t bytes[PPL][4];
#pragma unroll
        for (int g = 0; g < PPL; ++g)
#pragma unroll
            for (int q = 0; q < 4; ++q) bytes[g][q] = __builtin_amdgcn_alignbyte(w[4 * g + q + 1], w[4 * g + q], sh);
#pragma unroll
        for (int g = 0; g < PPL; ++g) {
            uint4* mine = reinterpret_cast<uint4*>(wbase + g * L::GB + r * L::RS + ((l - 1u) & 15u) * L::CS);
#pragma unroll
            for (uint32_t q = 0; q < L::CS / 16u; ++q) mine[q] = make_uint4(0, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();

        // ---- match: 64 PPL walks; mlen = the longest match of every start position
        const uint32_t nrel = live ? n - p0 : 0u;  // positions of the piece from p0 on
        uint32_t pgh[PPL], maxd[PPL], wlane[PPL], c[PPL], mlen[PPL];
        bool alive[PPL];
        uint2 rec[PPL];
#pragma unroll
        for (int g = 0; g < PPL; ++g) {
            const uint32_t rel = 16u * g + l;
            pgh[g] = (uint32_t)sbase + p0 + rel;  // position in the SAMPLE: what the keep rule hashes
            const uint32_t rem = rel < nrel ? nrel - rel : 0u;
            maxd[g] = rem < 16u ? rem : 16u;
            alive[g] = maxd[g] > 0u;
            wlane[g] = wr_off + g * L::GB;
            c[g] = bytes[g][0] & 0xFFu;
            rec[g] = rootc[(P.root_base ^ c[g]) & 255u];
            mlen[g] = 0u;
        }
        {
            uint32_t lcs = l * L::CS;
            asm volatile("" : "+v"(lcs));
            Walk7Ctx W{trie_b, smp, lcs, P.dropout, P.seed};
            uint32_t seen = 0;
            Walk7<DROPOUT, WIDE, PPL, 0, OVF>::run(W, bytes, maxd, pgh, wlane, alive, rec, c, mlen, seen);
            if (OVF) {
                // a match whose rank the 16-bit entries cannot hold (one of the vocabulary's least matched tokens): the stretch
                // from here to the next cut goes to the redo kernel, exactly as a stretch without a cut does
                const uint64_t ob = __builtin_amdgcn_ballot_w64(seen > P.ovf_limit);
                if (live && !probe && ((ob >> (16u * r)) & 0xFFFFull) != 0ull) {
                    probe = true;
                    redo_p0 = p0;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        E7_STAMP(1)  // text window, reset, walk

        // ---- the trip's end E (relative to p0): the last q in (0, min(SPAN, nrel)] with max over the starts before q of
        // (start + longest match) == q — nothing crosses q and something ends there (cuts.hip) — or the piece's end
        uint32_t e_rel = 0;
        {
            uint32_t carry = 0;
#pragma unroll
            for (int g = 0; g < PPL; ++g) {
                uint32_t v = 16u * g + l + mlen[g];
                v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false));  // row_shr:1
                v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false));  // row_shr:2
                v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false));  // row_shr:4
                v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false));  // row_shr:8
                v = max(v, carry);
                carry = row_bcast_u32<0x15F>(v);  // row_newbcast:15
                const uint32_t q = 16u * g + l + 1u;
                const bool ok = q <= nrel && (q == nrel || (v == q && (!probe || q > 16u)));
                const uint64_t bal = __builtin_amdgcn_ballot_w64(ok);
                const uint32_t mrow = (uint32_t)(bal >> (16u * r)) & 0xFFFFu;
                if (mrow) e_rel = 16u * g + 32u - (uint32_t)__builtin_clz(mrow);
            }
        }
        const bool trip_ok = live && e_rel != 0u && !probe;
        const uint32_t ge = e_rel >> 4, le = e_rel & 15u;
        E7_STAMP(2)  // cut

        // ---- forward: a[p0] = 1; fin_a[g] = a of this lane's position of group g under the exponent ea[g]
        double fin_a[PPL];
        int ea[PPL];
        double acc = (l == 0u) ? 1.0 : 0.0;
        int erow = 0;
        bool bad = false;  // a position of the trip nothing reaches, or a value out of range
#pragma unroll
        for (int g = 0; g < PPL; ++g) {
            uint32_t ha[16];
            e7_fwd_entries<WIDE>(col_off + g * L::GB, ha);
            double fin = 0.0;
            E7_FWD_GROUP(wtab_b, hot_lim, ha, acc, fin)
            fin_a[g] = fin;
            ea[g] = erow;
            if (16u * g + l <= e_rel && !(fin > 0.0 && fin <= 1.7976931348623157e308)) bad = true;
            const int e = row_max_exponent(acc);
            if (e > -100000) {
                acc = ldexp(acc, -e);
                erow += e;
            }
        }
        E7_STAMP(3)  // forward steps
        // Z = a[E]: lane le of group ge (lane 0 of the ring after the last group when E = p0 + SPAN)
        double zval = 1.0;
        int ez = 0;
        {
            const int src = (int)((r * 16u + le) * 4u);
#pragma unroll
            for (int g = 0; g < PPL; ++g) {
                const uint64_t fb = (uint64_t)__double_as_longlong(fin_a[g]);
                const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)fb);
                const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)(fb >> 32));
                if (ge == (uint32_t)g) {
                    zval = __hiloint2double((int)hi, (int)lo);
                    ez = ea[g];
                }
            }
            const double top = row_bcast_f64<0>(acc);
            if (ge == (uint32_t)PPL) {
                zval = top;
                ez = erow;
                if (!(top > 0.0 && top <= 1.7976931348623157e308)) bad = true;
            }
        }
        if (trip_ok && bad) atomicMax(Wk.range_flag, 1ULL);
        if (trip_ok && zval > 0.0) {  // Z of the piece so far: mantissa in [0.5, 1), exponent apart (lattice.rs:290-291 at the piece's end)
            const int e1 = __builtin_amdgcn_frexp_exp(zval);
            zm *= ldexp(zval, -e1);
            const int e2 = __builtin_amdgcn_frexp_exp(zm);
            zm = ldexp(zm, -e2);
            ze += ez + e1 + e2;
        }
        const double inv_z = 1.0 / zval;
        E7_STAMP(4)  // Z

        // ---- backward + marginals: b[E] = 1, every position above E stays 0
        {
            double accb = 0.0;
            int eb = 0;
            {   // the step of position p0 + SPAN alone (group PPL, lane 0): only E itself can be there
                if (trip_ok && ge == (uint32_t)PPL && l == 0u) accb = 1.0;
                uint32_t ha = (uint32_t)lds_ld<ET>(wr_off + (PPL - 1) * L::GB + 15u * L::CS) << 4;
                double sv = lds_ld<double>(((!COLD || ha <= hot_lim) ? ha : 0u) + 8u);
                if (COLD) {
                    asm volatile("" : "+v"(sv));
                    if (ha > hot_lim) sv = buf_ld_f64(wtab_b, ha >> 1);
                }
                const double cs_low = ldexp(fin_a[PPL - 1] * inv_z, ea[PPL - 1] - ez);
                double cs = cs_low;  // (every lane's start lies one group below)
                e7_bwd_step<0, COLD>(sv, ha, cs, cs_low, hot_lim, expected, accb);
            }
#pragma unroll
            for (int g = PPL - 1; g >= 0; --g) {
                if (trip_ok && ge == (uint32_t)g && l == le) accb = 1.0;
                const uint32_t vb_low = g > 0 ? wr_off + (g - 1) * L::GB : zero_rd;
                uint32_t va = wr_off + g * L::GB;
                uint32_t ha[16];
                E7BwdEntries<WIDE, 15>::run(va, vb_low, ha);
                double cs = ldexp(fin_a[g] * inv_z, ea[g] - ez + eb);
                const double cs_low = g > 0 ? ldexp(fin_a[g > 0 ? g - 1 : 0] * inv_z, ea[g > 0 ? g - 1 : 0] - ez + eb) : 0.0;
                E7_BWD_GROUP(wtab_b, hot_lim, ha, cs, cs_low, expected, accb)
                const int e = row_max_exponent(accb);
                if (e > -100000) {
                    accb = ldexp(accb, -e);
                    eb += e;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        E7_STAMP(5)  // backward steps, marginals

        // ---- next trip, end of the piece, or no cut in this trip (probe mode, see above)
        if (live) {
            if (nrel == 0u) {
                need_new = true;
            } else if (e_rel == 0u && PPL > 1) {  // (nrel > SPAN: the piece's end would have closed the trip)
                if (!probe) {
                    probe = true;
                    redo_p0 = p0;
                }
                p0 += 16u * PPL - 16u;  // > 0: every trip moves on, and a piece's last trip always closes
            } else {
                if (PPL == 1 && e_rel == 0u) {  // (a trip of 16 positions cannot probe: the rest of the piece is the stretch)
                    probe = true;
                    redo_p0 = p0;
                    e_rel = nrel;
                }
                if (probe) {
                    if (l == 0u) {
                        const unsigned long long i = atomicAdd(Wk.redo_count, 1ULL);
                        if (i < Wk.redo_cap) {
                            Wk.redo_offs[2 * i] = beg + redo_p0;
                            Wk.redo_offs[2 * i + 1] = beg + p0 + e_rel;
                            Wk.redo_sample[2 * i] = Wk.redo_sample[2 * i + 1] = DROPOUT ? smp : 0u;
                            Wk.redo_base[2 * i] = Wk.redo_base[2 * i + 1] = sbase + redo_p0;
                            Wk.redo_snip[2 * i] = Wk.redo_snip[2 * i + 1] = Wk.snip_of ? Wk.snip_of[s] : s;
                        } else {
                            atomicMax(Wk.range_flag, 4ULL);
                        }
                    }
                    probe = false;
                }
                if (e_rel >= nrel) need_new = true;
                else p0 += e_rel;
            }
            if (need_new && l == 0u) {
                const double z = log(zm) + (double)ze * 0.6931471805599453;
                atomicAdd(&Wk.zsnip[Wk.snip_of ? Wk.snip_of[s] : s], z);
                zsum += z;
            }
        }
    }
    if (zsum != 0.0) atomicAdd(Wk.logz_sum, zsum);
#ifdef TGX_E7_STAMPS
    if (P.stamps && lane == 0u) {
        unsigned long long* o = P.stamps + (size_t)(blockIdx.x * (blockDim.x >> 6) + wave) * 8u;
        for (int i = 0; i < 6; ++i) o[i] = seg[i];
        o[6] = trips;
    }
#endif
    __syncthreads();
    {
        const double2* hot = reinterpret_cast<const double2*>(smem);
        for (uint32_t i = 1u + threadIdx.x; i <= n_hot; i += blockDim.x) {
            const double v = hot[i].x;
            if (v != 0.0) __hip_atomic_fetch_add(&expected[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---- estep7_redo_kernel: the stretches estep7_kernel could not close within a trip ------------------------------------
// A stretch [b, e) of the redo list starts at a cut and ends with its piece, so it is a lattice of its own, of any
// length.  One row per stretch, 16 positions per trip, two phases with the trips' state spilled to HBM scratch:
//   F  ascending trips: walk, forward steps with the ring carried from trip to trip; a[p] and its block exponent go to
//      `alpha` / `aexp`, the row's match entries of the trip (512 bytes, or 1 KiB with 32-bit entries) to `mscratch`;
//   B  descending trips: the entries of trip t and of trip t - 1 come back into two LDS slots, backward steps with the
//      ring carried, marginals exactly as in estep7_kernel (cs from the stored a[p]).
// No reversed trie and no second walk here either; the rate does not matter (1 trip in ~10 000 ends up here).
// The rows of a wave run each phase in lockstep over the longest of their four stretches.
template <bool DROPOUT, bool COLD, bool WIDE>
__global__ __launch_bounds__(256) void estep7_redo_kernel(Estep7RedoParams P) {
    using L = E7Lay<WIDE>;
    using ET = typename L::ET;
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t l = lane & 15u, r = lane >> 4;
    const uint32_t wave = threadIdx.x >> 6;
    const uint2* __restrict__ trie = reinterpret_cast<const uint2*>(P.trie8t);
    const __amdgpu_buffer_rsrc_t trie_b = make_rsrc(P.trie8t, P.n_slots * 8u);
    const __amdgpu_buffer_rsrc_t wtab_b = make_rsrc(P.wtab, (P.n_tok + 1u) * 8u);
    constexpr uint32_t lds0 = 0u;
    const uint32_t n_hot = P.n_hot;
    typedef __attribute__((address_space(1))) double gdouble;
    gdouble* expected = (gdouble*)P.expected;
    asm volatile("" : "+v"(expected));
    {
        double2* hot = reinterpret_cast<double2*>(smem);
        for (uint32_t i = threadIdx.x; i <= n_hot; i += blockDim.x) hot[i] = make_double2(0.0, P.wtab[i]);
        uint2* rw = reinterpret_cast<uint2*>(smem + P.root_off);
        for (uint32_t i = threadIdx.x; i < 256u; i += blockDim.x) rw[i] = trie[(P.root_base & ~255u) + i];
        uint4* zg = reinterpret_cast<uint4*>(smem + P.zero_off);
        for (uint32_t i = threadIdx.x; i < L::GB / 16u; i += blockDim.x) zg[i] = make_uint4(0, 0, 0, 0);
        __syncthreads();
    }
    const uint2* rootc = reinterpret_cast<const uint2*>(smem + P.root_off);
    const uint32_t hot_lim = n_hot << 4;
    const uint32_t cidx = (l - 1u) & 15u;
    const uint32_t slot_off = lds0 + P.idx_off + wave * (2u * L::GB);   // this wave's two slots
    const uint32_t my_col = r * L::RS + cidx * L::CS;                   // this lane's column within a slot (and within a spilled row image: - r RS)
    const uint32_t my_ent = r * L::RS + l * L::ES;                      // this lane's entries as a walker / as the START lane
    const uint32_t zero_rd = lds0 + P.zero_off + my_ent;
    double zsum = 0.0;
    for (;;) {
        // ---- four stretches, one per row
        const uint64_t i = claim_rows(P.queue, true, r);
        const bool live = i < P.n_redo;
        uint64_t beg = 0, tb = 0;
        uint32_t n = 0, smp = 0;
        uint64_t sbase = 0;
        if (live) {
            beg = P.redo_offs[2 * i];
            n = (uint32_t)(P.redo_offs[2 * i + 1] - beg);
            tb = P.tbase[i];
            smp = P.redo_sample[2 * i];
            sbase = P.redo_base[2 * i];
        }
        if (__builtin_amdgcn_ballot_w64(live) == 0) break;
        const uint32_t T = live ? n / 16u + 1u : 0u;  // trips: position n lies in the last one
        uint32_t t_max = T;
        t_max = max(t_max, (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane ^ 16u) * 4u), (int)t_max));
        t_max = max(t_max, (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane ^ 32u) * 4u), (int)t_max));
        t_max = (uint32_t)__builtin_amdgcn_readfirstlane((int)t_max);
        const uint32_t tn = n >> 4, ln = n & 15u;
        double* const alpha = P.alpha + tb * 16u;
        int32_t* const aexp = P.aexp + tb;
        unsigned char* const mrow = P.mscratch + tb * L::RS;

        // ---- phase F
        double acc = (l == 0u) ? 1.0 : 0.0, zval = 1.0;
        int erow = 0, ez = 0;
        bool bad = false;
        for (uint32_t t = 0; t < t_max; ++t) {
            const bool act = live && t < T;
            const uint32_t p0 = 16u * t;
            const uintptr_t addr = reinterpret_cast<uintptr_t>(P.text + (act ? beg + p0 + l : 0));
            const uint32_t sh = (uint32_t)(addr & 3u);
            const uint32_t* __restrict__ wp = reinterpret_cast<const uint32_t*>(addr & ~uintptr_t(3));
            uint32_t w[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) w[q] = wp[q];
            uint32_t bytes[1][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bytes[0][q] = __builtin_amdgcn_alignbyte(w[q + 1], w[q], sh);
#pragma unroll
            for (uint32_t q = 0; q < L::CS / 16u; ++q) lds_st<u32x4_t>(slot_off + my_col + 16u * q, u32x4_t{0u, 0u, 0u, 0u});
            __builtin_amdgcn_wave_barrier();
            uint32_t pgh[1], maxd[1], wlane[1], c[1], mlen[1];
            bool alive[1];
            uint2 rec[1];
            const uint32_t rel = p0 + l;
            pgh[0] = (uint32_t)sbase + rel;
            const uint32_t rem = (act && rel < n) ? n - rel : 0u;
            maxd[0] = rem < 16u ? rem : 16u;
            alive[0] = maxd[0] > 0u;
            wlane[0] = slot_off + my_ent;
            c[0] = bytes[0][0] & 0xFFu;
            rec[0] = rootc[(P.root_base ^ c[0]) & 255u];
            mlen[0] = 0u;
            {
                uint32_t lcs = l * L::CS;
                asm volatile("" : "+v"(lcs));
                Walk7Ctx W{trie_b, smp, lcs, P.dropout, P.seed};
                uint32_t seen_unused = 0;
                Walk7<DROPOUT, WIDE, 1, 0>::run(W, bytes, maxd, pgh, wlane, alive, rec, c, mlen, seen_unused);
            }
            __builtin_amdgcn_wave_barrier();
            uint32_t ha[16];
            u32x4_t colw[L::CS / 16u];
#pragma unroll
            for (uint32_t q = 0; q < L::CS / 16u; ++q) colw[q] = lds_ld<u32x4_t>(slot_off + my_col + 16u * q);
            e7_fwd_entries<WIDE>(slot_off + my_col, ha);
            if (act) {  // the row's entries of this trip -> scratch (this lane's column)
#pragma unroll
                for (uint32_t q = 0; q < L::CS / 16u; ++q)
                    *reinterpret_cast<u32x4_t*>(mrow + (size_t)t * L::RS + cidx * L::CS + 16u * q) = colw[q];
            }
            double fin = 0.0;
            E7_FWD_GROUP(wtab_b, hot_lim, ha, acc, fin)
            if (act) {
                alpha[p0 + l] = fin;
                if (l == 0u) aexp[t] = erow;
                if (rel <= n && !(fin > 0.0 && fin <= 1.7976931348623157e308)) bad = true;
            }
            {   // Z = a[n]: lane ln of trip tn
                const uint64_t fb = (uint64_t)__double_as_longlong(fin);
                const int src = (int)((r * 16u + ln) * 4u);
                const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)fb);
                const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)(fb >> 32));
                if (act && t == tn) {
                    zval = __hiloint2double((int)hi, (int)lo);
                    ez = erow;
                }
            }
            const int e = row_max_exponent(acc);
            if (e > -100000) {
                acc = ldexp(acc, -e);
                erow += e;
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (live && bad) atomicMax(P.range_flag, 1ULL);
        if (live && l == 0u) {
            const double z = log(zval) + (double)ez * 0.6931471805599453;
            atomicAdd(&P.zsnip[P.redo_snip[2 * i]], z);
            zsum += z;
        }
        const double inv_z = 1.0 / zval;
        // the spilled entries and a[] come back through the caches of this CU: make the stores visible first
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");

        // ---- phase B: slot (t & 1) holds the entries of trip t, slot ((t - 1) & 1) those of trip t - 1
        auto fetch = [&](uint32_t t) {  // trip t's entries of this row -> slot (t & 1); zeros for a row without that trip
            const bool act = live && t < T;
#pragma unroll
            for (uint32_t q = 0; q < L::CS / 16u; ++q) {
                u32x4_t v{0u, 0u, 0u, 0u};
                if (act) v = *reinterpret_cast<const u32x4_t*>(mrow + (size_t)t * L::RS + cidx * L::CS + 16u * q);
                lds_st<u32x4_t>(slot_off + (t & 1u) * L::GB + my_col + 16u * q, v);
            }
        };
        double accb = 0.0;
        int eb = 0;
        fetch(t_max - 1u);
        for (uint32_t t = t_max; t-- > 0u;) {
            if (t > 0u) fetch(t - 1u);
            __builtin_amdgcn_wave_barrier();
            const bool act = live && t < T;
            if (act && t == tn && l == ln) accb = 1.0;  // b[n] = 1
            const uint32_t vb_low = t > 0u ? slot_off + ((t - 1u) & 1u) * L::GB + my_ent : zero_rd;
            uint32_t va = slot_off + (t & 1u) * L::GB + my_ent;
            uint32_t ha[16];
            E7BwdEntries<WIDE, 15>::run(va, vb_low, ha);
            double c_cur = 0.0, c_low = 0.0;
            int e_cur = 0, e_low = 0;
            if (act) {
                c_cur = alpha[16u * t + l] * inv_z;
                e_cur = aexp[t];
                if (t > 0u) {
                    c_low = alpha[16u * (t - 1u) + l] * inv_z;
                    e_low = aexp[t - 1u];
                }
            }
            double cs = ldexp(c_cur, e_cur - ez + eb);
            const double cs_low = ldexp(c_low, e_low - ez + eb);
            E7_BWD_GROUP(wtab_b, hot_lim, ha, cs, cs_low, expected, accb)
            const int e = row_max_exponent(accb);
            if (e > -100000) {
                accb = ldexp(accb, -e);
                eb += e;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (zsum != 0.0) atomicAdd(P.logz_sum, zsum);
    __syncthreads();
    {
        const double2* hot = reinterpret_cast<const double2*>(smem);
        for (uint32_t i = 1u + threadIdx.x; i <= n_hot; i += blockDim.x) {
            const double v = hot[i].x;
            if (v != 0.0) __hip_atomic_fetch_add(&expected[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

uint32_t estep7_redo_max_hot(bool wide) {
    const uint32_t gb = wide ? 4096u : 2048u;
    const uint32_t fixed = 2048u + gb + gb + 4u * 2u * gb;  // root, alignment slack, zero group, four waves x two slots
    return (64u * 1024u - fixed) / 16u - 1u;
}
hipError_t launch_estep7_redo(Estep7RedoParams p, bool wide, uint32_t num_cus, hipStream_t stream) {
    if (p.n_redo == 0) return hipSuccess;
    const uint32_t gb = wide ? 4096u : 2048u;
    if (p.n_hot > p.n_tok || p.n_hot > estep7_redo_max_hot(wide) || (!wide && p.n_tok > 65535u)) return hipErrorInvalidValue;
    const bool cold = p.n_hot < p.n_tok;
    p.root_off = 16u * (p.n_hot + 1u);
    p.zero_off = (p.root_off + 2048u + gb - 1u) & ~(gb - 1u);
    p.idx_off = p.zero_off + gb;
    const uint32_t lds = p.idx_off + 4u * 2u * gb;
    const bool d = p.dropout > 0.0;
    auto fn = wide ? (d ? (cold ? estep7_redo_kernel<true, true, true> : estep7_redo_kernel<true, false, true>)
                        : (cold ? estep7_redo_kernel<false, true, true> : estep7_redo_kernel<false, false, true>))
                   : (d ? (cold ? estep7_redo_kernel<true, true, false> : estep7_redo_kernel<true, false, false>)
                        : (cold ? estep7_redo_kernel<false, true, false> : estep7_redo_kernel<false, false, false>));
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    const uint32_t blocks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((p.n_redo + 15) / 16, (uint64_t)num_cus * 2u));
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), lds, stream, p);
    return hipGetLastError();
}

typedef void (*estep7_fn)(Estep7Params);
static estep7_fn pick_estep7(bool dropout, bool cold, bool wide, int ppl, bool ovf = false) {
    if (ovf) return dropout ? estep7_kernel<true, true, false, 4, true> : estep7_kernel<false, true, false, 4, true>;  // (cold, 16-bit entries, four positions per lane)
#define TGX_E7P(D, C, W) (ppl == 1 ? estep7_kernel<D, C, W, 1> : ppl == 2 ? estep7_kernel<D, C, W, 2> : ppl == 3 ? estep7_kernel<D, C, W, 3> : estep7_kernel<D, C, W, 4>)
#define TGX_E7W(D, C) (wide ? TGX_E7P(D, C, true) : TGX_E7P(D, C, false))
    if (dropout) return cold ? TGX_E7W(true, true) : TGX_E7W(true, false);
    return cold ? TGX_E7W(false, true) : TGX_E7W(false, false);
#undef TGX_E7W
#undef TGX_E7P
}

// LDS of one block: {sum, w} of n_hot + 1 ranks, root records, the all-zero group, waves x ppl groups of match entries
uint32_t estep7_lds_layout(uint32_t n_hot, bool wide, int waves, int ppl, uint32_t* root_off, uint32_t* zero_off, uint32_t* idx_off) {
    const uint32_t gb = wide ? 4096u : 2048u;
    const uint32_t ro = 16u * (n_hot + 1u);
    const uint32_t zo = (ro + 2048u + gb - 1u) & ~(gb - 1u);  // group-aligned: the walkers OR column offsets into entry addresses
    const uint32_t io = zo + gb;
    if (root_off) *root_off = ro;
    if (zero_off) *zero_off = zo;
    if (idx_off) *idx_off = io;
    return io + (uint32_t)waves * (uint32_t)ppl * gb;
}
// ---- match counts over a sample of the text (which ranks deserve the LDS) -------------------------------------------
// One thread per position: the walk of estep7_kernel, one count per token met.  The first kCountLds ranks (the provisional
// order puts the likely winners there) are counted in the block's LDS and flushed once; the others — and there may be a
// frequent one among them, which is the point of counting — go to memory directly: the sample is small.
constexpr uint32_t kCountLds = 16384;
__global__ __launch_bounds__(256) void match_count_kernel(const uint8_t* __restrict__ text, uint64_t n_bytes, uint32_t chunk, uint64_t stride,
                                                          const uint2* __restrict__ trie, uint32_t n_slots, uint32_t root_base, uint32_t n_tok,
                                                          unsigned int* __restrict__ counts) {
    __shared__ unsigned int hist[kCountLds];
    for (uint32_t i = threadIdx.x; i < kCountLds; i += blockDim.x) hist[i] = 0u;
    __syncthreads();
    const uint64_t n_chunks = (n_bytes + stride - 1) / stride;
    const uint64_t total = n_chunks * chunk;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t p = (i / chunk) * stride + (i % chunk);
        if (p >= n_bytes) continue;
        const uint32_t reach = (uint32_t)(n_bytes - p < 16u ? n_bytes - p : 16u);
        uint32_t t = root_base ^ text[p];
        for (uint32_t k = 0; k < reach; ++k) {
            if (t >= n_slots) break;
            const uint2 rec = trie[t];
            if ((rec.x >> 24) != text[p + k]) break;
            if (rec.y) {
                if (rec.y < kCountLds) atomicAdd(&hist[rec.y], 1u);
                else if (rec.y <= n_tok) atomicAdd(&counts[rec.y], 1u);
            }
            if (k + 1u < reach) t = (rec.x ^ (uint32_t)text[p + k + 1u]) & 0xFFFFFFu;
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < kCountLds; i += blockDim.x)
        if (hist[i] && i <= n_tok) atomicAdd(&counts[i], hist[i]);
}
// the same count over encode5_kernel's records {8 base | label << 24, rank of the SCORE VALUE}: how often every value is read
__global__ __launch_bounds__(256) void value_count_kernel(const uint8_t* __restrict__ text, uint64_t n_bytes, uint32_t chunk, uint64_t stride,
                                                          const uint2* __restrict__ trie, uint32_t n_slots, uint32_t root_base, uint32_t n_values,
                                                          uint32_t max_len, unsigned int* __restrict__ counts) {
    __shared__ unsigned int hist[kCountLds];
    for (uint32_t i = threadIdx.x; i < kCountLds; i += blockDim.x) hist[i] = 0u;
    __syncthreads();
    const uint64_t n_chunks = (n_bytes + stride - 1) / stride;
    const uint64_t total = n_chunks * chunk;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t p = (i / chunk) * stride + (i % chunk);
        if (p >= n_bytes) continue;
        const uint32_t reach = (uint32_t)(n_bytes - p < max_len ? n_bytes - p : max_len);
        uint32_t t = root_base ^ text[p];
        for (uint32_t k = 0; k < reach; ++k) {
            if (t >= n_slots) break;
            const uint2 rec = trie[t];
            if ((rec.x >> 24) != text[p + k]) break;
            const uint32_t rank = rec.y & 0xFFFFu;
            if (rank) {
                if (rank < kCountLds) atomicAdd(&hist[rank], 1u);
                else if (rank <= n_values) atomicAdd(&counts[rank], 1u);
            }
            if (k + 1u < reach) t = ((rec.x ^ ((uint32_t)text[p + k + 1u] << 3)) & 0xFFFFFFu) >> 3;
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < kCountLds; i += blockDim.x)
        if (hist[i] && i <= n_values) atomicAdd(&counts[i], hist[i]);
}
__global__ __launch_bounds__(256) void rank_remap_kernel(uint2* __restrict__ trie, uint32_t n_slots, const uint32_t* __restrict__ perm) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_slots) return;
    const uint32_t r = trie[t].y;
    if (r) trie[t].y = perm[r];
}
hipError_t launch_match_count(const uint8_t* text, uint64_t n_bytes, uint32_t chunk, uint64_t stride, const void* trie8t, uint32_t n_slots,
                              uint32_t root_base, uint32_t n_tok, unsigned int* counts, uint32_t num_cus, hipStream_t stream) {
    if (!n_bytes || !chunk || stride < chunk) return hipErrorInvalidValue;
    hipLaunchKernelGGL(match_count_kernel, dim3(num_cus * 4u), dim3(256), 0, stream, text, n_bytes, chunk, stride,
                       reinterpret_cast<const uint2*>(trie8t), n_slots, root_base, n_tok, counts);
    return hipGetLastError();
}
hipError_t launch_value_count(const uint8_t* text, uint64_t n_bytes, uint32_t chunk, uint64_t stride, const void* trie8, uint32_t n_slots,
                              uint32_t root_base, uint32_t n_values, uint32_t max_len, unsigned int* counts, uint32_t num_cus, hipStream_t stream) {
    if (!n_bytes || !chunk || stride < chunk) return hipErrorInvalidValue;
    hipLaunchKernelGGL(value_count_kernel, dim3(num_cus * 4u), dim3(256), 0, stream, text, n_bytes, chunk, stride,
                       reinterpret_cast<const uint2*>(trie8), n_slots, root_base, n_values, max_len, counts);
    return hipGetLastError();
}
hipError_t launch_rank_remap(void* trie8t, uint32_t n_slots, const uint32_t* perm, hipStream_t stream) {
    hipLaunchKernelGGL(rank_remap_kernel, dim3((n_slots + 255u) / 256u), dim3(256), 0, stream, reinterpret_cast<uint2*>(trie8t), n_slots, perm);
    return hipGetLastError();
}

uint32_t estep7_max_hot(bool wide, int waves, int ppl, uint32_t budget) {
    const uint32_t fixed = estep7_lds_layout(0u, wide, waves, ppl, nullptr, nullptr, nullptr) + (wide ? 4096u : 2048u);  // alignment slack
    return budget > fixed + 64u ? (budget - fixed) / 16u : 0u;
}
hipError_t estep7_waves_per_simd(bool dropout, bool cold, bool wide, int ppl, int* out) {
    hipFuncAttributes attr;
    hipError_t e = hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(pick_estep7(dropout, cold, wide, ppl)));
    if (e != hipSuccess) return e;
    const int regs = (attr.numRegs + 7) & ~7;
    *out = regs > 0 ? (512 / regs > 8 ? 8 : 512 / regs) : 8;
    return hipSuccess;
}
// p.ovf_limit < n_tok (and !wide): the overflow build — ranks beyond ovf_limit (<= 65 535) are left to the redo kernel
hipError_t launch_estep7(Estep7Params p, bool wide, int ppl, int waves, uint32_t blocks, hipStream_t stream) {
    const bool cold = p.n_hot < p.n_tok;
    const bool ovf = !wide && p.ovf_limit < p.n_tok;
    const uint32_t lds = estep7_lds_layout(p.n_hot, wide, waves, ppl, &p.root_off, &p.zero_off, &p.idx_off);
    if (lds > 160u * 1024u || p.n_hot > p.n_tok || ppl < 1 || ppl > 4 || waves < 1 || waves > 12 || (!wide && !ovf && p.n_tok > 65535u)) return hipErrorInvalidValue;
    if (ovf && (p.ovf_limit > 65535u || ppl != 4 || !cold)) return hipErrorInvalidValue;
    if (!p.work) return hipErrorInvalidValue;
    estep7_fn fn = pick_estep7(p.dropout > 0.0, cold, wide, ppl, ovf);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(const_cast<Estep7Work*>(p.work), &p.host_work, sizeof(Estep7Work), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(64u * (uint32_t)waves), lds, stream, p);
    return hipGetLastError();
}

}  // namespace tgx
