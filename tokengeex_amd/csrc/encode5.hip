// encode5_kernel (gfx950 / CDNA4, wave64): Model::encode (reference src/model.rs:59-129) for vocabularies
// whose tokens have at most 16 bytes and finite scores — four samples per wave on 16-lane rows like
// encode4_kernel (kernels.hip), with two changes that attack what bounded that kernel (DESIGN.md section 6:
// the texture-address path at 1.6 cycles per 16-byte gather lane, and 8 KiB of LDS per wave and 64
// positions capping the waves per CU):
//
//   * 8-byte label-checked trie records (trie_build.h: Trie8Rec).  A walk keeps only the record; a step is one
//     8-byte gather {8 * base | label << 24, rank of the score value} (0.9 cycles per lane, measured:
//     profiles/r02/a_gather_8byte_records.txt), one compare of the label with the text byte and — round 3 — three
//     instructions for the next address ((rec ^ byte << 3) & 0xFFFFFF, dead walks to record 0): the kernel is
//     bound by vector-instruction issue, every instruction of a step counts 43 times per
//     256 positions.
//   * scores are not carried through the match buffer.  The distinct score VALUES of the vocabulary are ranked
//     by how often their tokens are expected to match (trie_build.h: Trie8) and a match is the 16-bit RANK of its
//     value ("no token" = rank 0 = a -inf entry): 2 KiB per wave and 64 positions instead of 8 KiB.  The first
//     n_hot values sit in a table in the block's LDS; round 3: a rank beyond the table is read from the same
//     table in HBM / L2 by the RELAXING lane (COLD builds: one exec-masked 8-byte load per cold match, issued with
//     the LDS reads of its group).  Every vocabulary with at most 65 535 distinct values runs this kernel — after
//     an M-step every token has its own (src/prune.rs:143-151) — and nothing can overflow: round 2 sent cold
//     values through per-wave pools in LDS whose overflow cost a second pass over the samples concerned.
//   * the PPL walks of a lane are staggered (Walk5), and the relaxation runs on relax5_step (device_common.h).
//
// Same candidate order and strict '>' as encode4_kernel, so the back-pointer bytes and the per-sample status
// are bit-identical and trace_kernel is shared.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "device_common.h"
#include "kernels.h"

namespace tgx {

// match-index buffer of one 16-position group: four sample rows of 16 columns x 16 start positions x 2 bytes
// = 2 KiB per wave and group, 512-byte aligned.  Entry (start u, len) of row r: r * 512 + ((len - 1 + u) & 15) * 32
// + u * 2 — lane l of the row, which at step u accumulates end position u + len with len - 1 = (l - u - 1) & 15,
// finds its 16 entries (u = 0..15) contiguous in column (l - 1) & 15: two ds_read_b128.  The walker's address is
// (lane constant) | ((l * 32 + d * 32) & 0x1E0): the alignment makes it one add and one and-or.  (A layout with
// 32 columns and no wrap-around needs no address arithmetic at all but twice the LDS, and LDS is what limits the
// waves per CU here.)
constexpr uint32_t kE5RowStride = 512;
constexpr uint32_t kE5GroupBytes = 4 * kE5RowStride;  // 2048

template <int PPL>
struct WalkCtx {
    __amdgpu_buffer_rsrc_t trie;  // Trie8Rec[], addressed by byte offset
    uint32_t s, l32;
    double dropout;
    uint64_t seed;
    bool estep_rule = false;  // the keep rule of populate_nodes (model.rs:48: skipped iff rand < dropout) instead of encode's (model.rs:100)
};

// RES walks (the walker waves of encode6_kernel, COLD builds): a match whose value is not in the block's LDS copy is
// RESOLVED by the walker — it reads the value from the table in L2 and parks it in a pool entry of its ring slot, the
// match index becomes that entry's — so the relaxer, whose steps are the sample's serial chain, finds every value in
// LDS.  The load is issued at the depth that finds the match, ahead of the walk's next gather, and completed at the
// next depth (vector loads return in order: by then it has arrived).  Indices: 0 = "no token", 1 .. shift = pool
// entries, rank + shift = the value of that rank (in the LDS copy iff <= hot_max).  A pool that is full leaves the
// rank in place: the relaxer reads those from L2 itself, as before.
struct Res5 {
    __amdgpu_buffer_rsrc_t cold;  // the value table from rank n_hot + 1 on
    uint32_t tab;                 // LDS byte offset of the block's table
    uint32_t shift, hot_max;      // see above
    uint32_t used_off;            // LDS byte offset of the slot's count of pool entries handed out
    uint32_t first, cap;          // the slot's pool entries: indices first .. first + cap - 1
    bool pend;                    // per lane: a cold match of the depth before awaits its value
    uint32_t pend_addr;
    double pend_val;
    __device__ __forceinline__ void complete() {
        if (pend) {
            const uint32_t e = __hip_atomic_fetch_add((__attribute__((address_space(3))) uint32_t*)(uintptr_t)used_off, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (e < cap) {
                lds_st<double>(tab + 8u * (first + e), pend_val);
                lds_st<uint16_t>(pend_addr, (uint16_t)(first + e));
            }
        }
        pend = false;
    }
};

// ---- the trie walks of a lane (PPL start positions), unrolled over the depth D by template recursion (with
// its early exit inside, hipcc's unroller gives up on the plain loop and the text bytes end up selected by
// v_cndmask chains).  The PPL walks are STAGGERED: a level consumes the record of walk g and at once requests
// walk g's next record, so while one walk's record is examined the gathers of the other walks are in flight
// (vector loads return in order: the wait for walk g's record leaves the PPL - 1 younger ones outstanding).
// With all gathers of a depth issued together and waited for together a step of two walks took 890 cycles
// against 540 for one.
// (Every lane issues every gather, finished walks from record 0: a load that only some paths issue would force
// the compiler to wait for ALL outstanding loads at every use — in-order counters cannot name a load that may
// not exist — and the stagger would be lost; lanes reading record 0 cost the texture path next to nothing,
// profiles/r03/d_gather3_dead_lanes.txt.  For the same reason the hottest slots are NOT read from an LDS copy:
// tried, 2 % at best, a step still waits for its slowest lane, which goes to L2.)
// SKIP (the COLD builds of encode5_kernel, round 4): from depth 11 on a walk whose lanes are all finished issues no
// gather — 5 of a trip's 43 gathers have no live lane at all (the trip goes on while ANY of its walks is alive).  The
// conditional load costs the stagger at those depths (see above), so the build with every value in LDS, whose vector and
// texture units are in balance, loses (12.63 -> 13.07 ms per GiB); the COLD build, whose cold values go through the texture
// path as well, gains (14.30 -> 13.70).  From depth 7 or 9 on it loses in both.
template <bool DROPOUT, bool LONG, bool RES, int PPL, int D, bool SKIP = false>
struct Walk5 {
    // rec[g], c[g]: record and text byte of depth D of walk g (the load may still be in flight)
    static __device__ __forceinline__ void run(const WalkCtx<PPL>& W, Res5& R, const uint32_t (&bytes)[PPL][4], const uint32_t (&maxd)[PPL],
                                               const uint32_t (&pg)[PPL], const uint32_t (&wlane)[PPL], bool (&alive)[PPL],
                                               uint2 (&rec)[PPL], uint32_t (&c)[PPL]) {
        static_assert(!RES || PPL == 1, "resolving walks keep one pending value per lane");
        constexpr int d = D;
        bool any = false;
#pragma unroll
        for (int g = 0; g < PPL; ++g) {
            asm("" : "+v"(rec[g].x), "+v"(rec[g].y));  // two 32-bit words (else the 64-bit load is picked apart with 64-bit ops)
            alive[g] = alive[g] && ((rec[g].x >> 24) == c[g]);
            const uint32_t rank = rec[g].y & 0xFFFFu;  // 0: no token ends here
            bool term = alive[g] && rank != 0u;
            if (DROPOUT) {  // model.rs:100: kept iff len <= 1 || dropout < rand
                if (term && d >= 1) {
                    const double u01 = dropout_u01(W.seed, W.s, pg[g], (uint32_t)d + 1u);
                    term = W.estep_rule ? !(u01 < W.dropout) : (W.dropout < u01);
                }
            }
            if (RES) {
                if (D > 0) R.complete();  // the cold match of the depth before: its value was requested ahead of this depth's record
                const uint32_t addr = wlane[g] | ((W.l32 + 32u * (uint32_t)d) & 0x1E0u);
                const uint32_t idx = rank + R.shift;
                if (term) lds_st<uint16_t>(addr, (uint16_t)idx);
                R.pend = term && idx > R.hot_max && R.cap != 0u;
                if (R.pend) {
                    R.pend_val = buf_ld_f64(R.cold, (idx - R.hot_max - 1u) << 3);
                    R.pend_addr = addr;
                }
            } else {
                if (term) lds_st<uint16_t>(wlane[g] | ((W.l32 + 32u * (uint32_t)d) & 0x1E0u), (uint16_t)rank);
            }
            // this walk's next record
            if (D + 1 < 16) {
                constexpr int e = D + 1 < 16 ? D + 1 : 15;
                c[g] = (bytes[g][e >> 2] >> ((e & 3) * 8)) & 0xFFu;
                alive[g] = alive[g] && ((uint32_t)e < maxd[g]);
                const uint32_t off = (rec[g].x ^ (c[g] << 3)) & 0xFFFFFFu;
                if (SKIP && D >= 11) {
                    if (__builtin_amdgcn_ballot_w64(alive[g]) != 0) rec[g] = buf_ld8(W.trie, alive[g] ? off : 0u);
                } else {
                    rec[g] = buf_ld8(W.trie, alive[g] ? off : 0u);
                }
            } else if (!LONG) {
                alive[g] = false;
            }  // LONG: alive[g] says whether the walk goes on past 16 bytes (rec[g] is its record of depth 15): e5_long_tail
            any = any || alive[g];
        }
        if (__builtin_amdgcn_ballot_w64(any) == 0) {
            if (RES) R.complete();
            return;
        }
        Walk5<DROPOUT, LONG, RES, PPL, D + 1, SKIP>::run(W, R, bytes, maxd, pg, wlane, alive, rec, c);
    }
};
template <bool DROPOUT, bool LONG, bool RES, int PPL, bool SKIP>
struct Walk5<DROPOUT, LONG, RES, PPL, 16, SKIP> {
    static __device__ __forceinline__ void run(const WalkCtx<PPL>&, Res5& R, const uint32_t (&)[PPL][4], const uint32_t (&)[PPL],
                                               const uint32_t (&)[PPL], const uint32_t (&)[PPL], bool (&)[PPL], uint2 (&)[PPL],
                                               uint32_t (&)[PPL]) {
        if (RES) R.complete();
    }
};

// The score values of N consecutive steps (from step `first`) of a lane, whose sixteen match indices are the 16-bit
// words in iw[8]: index * 8 is the byte offset of the value in the block's LDS — the copy of the first hot_bytes of the
// value table (at LDS byte offset `tab`).  Requested a quarter group ahead of use.
template <int N>
__device__ __forceinline__ void e5_scores_issue(uint32_t tab, const uint32_t (&iw)[8], int first, u32x2_t (&sp)[N]) {
#pragma unroll
    for (int u = 0; u < N; ++u) {
        const int v = first + u;
        const uint32_t a = (v & 1) ? ((iw[v >> 1] >> 16) << 3) : ((iw[v >> 1] & 0xFFFFu) << 3);
        sp[u] = lds_ld<u32x2_t>(tab + a);
    }
}
__device__ __forceinline__ double e5_score_value(const u32x2_t& p) { return __hiloint2double((int)p.y, (int)p.x); }

// COLD builds (vocabularies with more distinct score values than the LDS copy holds): an index beyond the copy is
// the RANK of a value that lives in the rest of the table in HBM / L2 (`cold`: the values from rank n_hot + 1 on),
// read under the cold lanes' EXEC mask — one exec-masked buffer load per entry in which some lane is cold (~33 of 64
// per 256 positions when a tenth of the matches are cold), a batch of N entries at a time.  (The LDS reads are
// unconditional, at clamped addresses, and exist before the conditional loads — the empty asm: written as
// `hot ? lds : hbm`, or with the LDS read sinkable, the compiler merges the two into one FLAT load of a selected
// address, which goes through the texture path for every lane.)
// Three other ways of getting the cold values were built and measured in round 3 (profiles/r03/README.md): moving
// them into pool slots in LDS before the relaxation, with or without its L2 round trip hidden behind the group
// before (e5_resolve: +0.4 ... +2.3 ms per GiB against this); unconditional loads whose hot lanes fall outside the
// buffer, merged by OR (+2.7 ms).  Each removes instructions of one kind and adds more of another.
template <int N>
__device__ __forceinline__ void e5_scores_cold(uint32_t tab, __amdgpu_buffer_rsrc_t cold, const uint32_t (&iw)[8], int first,
                                               uint32_t hot_bytes, double (&sv)[N]) {
    uint32_t a[N];
#pragma unroll
    for (int u = 0; u < N; ++u) {
        const int v = first + u;
        a[u] = (v & 1) ? ((iw[v >> 1] >> 16) << 3) : ((iw[v >> 1] & 0xFFFFu) << 3);
        sv[u] = lds_ld<double>(tab + (a[u] < hot_bytes ? a[u] : 0u));
    }
    static_assert(N == 4 || N == 8, "e5_scores_cold: four or eight values at a time");
    if (N == 8) asm volatile("" : "+v"(sv[0]), "+v"(sv[1]), "+v"(sv[2]), "+v"(sv[3]), "+v"(sv[4 % N]), "+v"(sv[5 % N]), "+v"(sv[6 % N]), "+v"(sv[7 % N]));
    else asm volatile("" : "+v"(sv[0]), "+v"(sv[1]), "+v"(sv[2]), "+v"(sv[3]));
#pragma unroll
    for (int u = 0; u < N; ++u)
        if (a[u] >= hot_bytes) sv[u] = buf_ld_f64(cold, a[u] - hot_bytes);
}

// one score by its rank (the rare paths of the LONG builds): LDS copy, or the rest of the table under a branch
template <bool COLD>
__device__ __forceinline__ double e5_score_one(uint32_t tab, __amdgpu_buffer_rsrc_t cold, uint32_t rank, uint32_t hot_bytes) {
    const uint32_t a = rank << 3;
    double v = lds_ld<double>(tab + ((!COLD || a < hot_bytes) ? a : 0u));
    if (COLD) {
        asm volatile("" : "+v"(v));
        if (a >= hot_bytes) v = buf_ld_f64(cold, a - hot_bytes);
    }
    return v;
}

// ---- LONG builds: vocabularies whose longest token has 17..32 bytes (after `merge`: src/cli.rs:723 defaults to 24) ----
// Such tokens are RARE matches (23 of the 32 600 tokens of a merged vocabulary), so the kernel keeps its 16-column
// structure and adds what encode4l_kernel (encode4l.hip, round 1) adds to encode4_kernel, but only on the paths that
// meet a long match: the walk goes on past depth 16 only where the trie continues (e5_long_tail, a plain loop:
// rarely entered); a match of 17..32 bytes is appended to a per-wave list in LDS instead of the match-index buffer;
// every lane has a second accumulator `far` for the end position 17..32 ahead, which the lane takes over when its
// step restarts it; after the 16 steps of the group a long match STARTS in, its entry is applied — best[start] +
// score into `acc` (end position 17..31 ahead of the group's first position) or `far` (32..47).  A group of 16
// positions runs the ordinary steps (relax5_step) unless some lane of the wave holds a pending `far` value or the
// list holds an entry: the slow steps (relax5l_step) are the exception, and a vocabulary with a few hundred long
// tokens encodes at the speed of one without.
// Order of candidates (model.rs:96-108: ascending starts, strict '>', so the LONGEST token wins among equal scores):
// a long candidate is applied out of order and replaces an equal score iff its token is longer; `far` wins ties
// against near candidates at a restart (everything in `far` starts earlier than any near start of that position).
// A list that fills up puts the wave's samples on P.redo_list: the host redoes exactly those with encode2_kernel.
constexpr uint32_t kE5LongCap = 62;                         // list entries per wave and iteration
constexpr uint32_t kE5LongBytes = 8u + kE5LongCap * 4u;     // {count, pad} + entries {lane | group << 6 | depth << 8 | rank << 16}
constexpr uint32_t kFarCode = 0x80u;                        // winner code of a long token: kFarCode | (length - 1)

__device__ __forceinline__ uint32_t e5_winner_len_m1(uint32_t code, uint32_t l) {  // near winners: the step that pushed them
    return (code & kFarCode) ? (code & 31u) : ((l - code - 1u) & 15u);
}

template <int U>
__device__ __forceinline__ void relax5l_step(double sv, double& acc, uint32_t& bpv, double& far, uint32_t& fbp, uint32_t& fin, double& fval) {
    constexpr uint64_t MU = kRowLane0 << U;  // lanes with l == U
    fin = sel_u32(MU, bpv, fin);             // winner of position p0 + U is final now
    fval = sel_f64(MU, acc, fval);           // and its score (the long matches that start there need it)
    const double best = row_bcast_f64<U>(acc);
    const double cur = sel_f64(MU, far, acc);          // lane U restarts on position + 16: from its long candidates
    const uint32_t curbp = sel_u32(MU, fbp, bpv);
    far = sel_f64(MU, -__builtin_huge_val(), far);     // `far` of lane U now stands for position + 32
    const double cand = best + sv;                     // model.rs:98
    const uint64_t take = __builtin_amdgcn_fcmp(cand, cur, 2 /* OGT: model.rs:101; far wins ties */);
    asm("v_max_f64 %0, %1, %2" : "=v"(acc) : "v"(cur), "v"(cand));
    bpv = sel_imm_u32<U>(take, curbp);
}

// the walk of group g's lanes past 16 bytes (alive: the walk matched 16 bytes and the trie goes on): appends the
// matches of 17..32 bytes to the wave's list.  rec_x: the walk's record of depth 15.
template <bool DROPOUT>
__device__ __forceinline__ void e5_long_tail(__amdgpu_buffer_rsrc_t trie, const uint8_t* __restrict__ text16, uint32_t rec_x, bool alive,
                                             uint32_t maxd, uint32_t lane, uint32_t g, uint32_t list_off, double dropout, uint64_t seed,
                                             uint32_t s, uint32_t pg) {
    bool more = alive && maxd > 16u;
    if (__builtin_amdgcn_ballot_w64(more) == 0) return;  // wave-uniform: the ordinary case
    struct __attribute__((packed, aligned(1))) Bytes16 { uint32_t w[4]; };
    Bytes16 b{{0u, 0u, 0u, 0u}};
    if (more) b = *reinterpret_cast<const Bytes16*>(text16);  // text bytes 16..31 of this lane's position (the text is padded)
    uint32_t x = rec_x;
#pragma unroll 1
    for (uint32_t d = 16u; d < 32u; ++d) {
        more = more && d < maxd;
        if (__builtin_amdgcn_ballot_w64(more) == 0) break;
        const uint32_t bi = d - 16u;
        const uint32_t word = (bi & 8u) ? ((bi & 4u) ? b.w[3] : b.w[2]) : ((bi & 4u) ? b.w[1] : b.w[0]);
        const uint32_t c = (word >> ((bi & 3u) * 8u)) & 0xFFu;
        const uint2 rec = buf_ld8(trie, more ? ((x ^ (c << 3)) & 0xFFFFFFu) : 0u);
        more = more && (rec.x >> 24) == c;
        const uint32_t rank = rec.y & 0xFFFFu;
        bool term = more && rank != 0u;
        if (DROPOUT) {
            if (term) term = dropout < dropout_u01(seed, s, pg, d + 1u);
        }
        if (term) {
            const uint32_t slot = __hip_atomic_fetch_add((__attribute__((address_space(3))) uint32_t*)(uintptr_t)list_off, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (slot < kE5LongCap) lds_st<uint32_t>(list_off + 8u + slot * 4u, lane | (g << 6) | (d << 8) | (rank << 16));
        }
        x = rec.x;
    }
}

// TGX_STAMPS=1 (diagnostic runs only): s_memtime stamps around the phases of an iteration, summed per wave
#define E5_STAMP(i)                                                    \
    if (P.stamps) {                                                    \
        __builtin_amdgcn_sched_barrier(0);                             \
        const uint64_t _now = (uint64_t)__builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_s_waitcnt(0xC07F);                            \
        __builtin_amdgcn_sched_barrier(0);                             \
        seg[i] += _now - t_last;                                       \
        t_last = _now;                                                 \
    }

template <bool DROPOUT, bool COLD, int PPL, bool LONG>
__global__ __launch_bounds__(1024, (PPL == 1 ? 8 : (PPL == 2 ? 6 : (PPL == 3 ? 4 : 1)))) void encode5_kernel(EncodeParams P, Encode5Params Q) {
    static_assert(!LONG || PPL == 4, "the long-token build runs four positions per lane");
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr uint32_t LM = LONG ? 32 : 16;
    constexpr uint32_t SPAN = 16u * PPL;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t l = lane & 15u, r = lane >> 4;
    const uint32_t wave = threadIdx.x >> 6;
    const uint2* __restrict__ trie = reinterpret_cast<const uint2*>(Q.trie8);
    const __amdgpu_buffer_rsrc_t trie_b = make_rsrc(Q.trie8, Q.trie_bytes);
    const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>(smem);  // LDS byte offset of the dynamic LDS (0 here)
    // ---- LDS: [0, 8 (n_hot + 1)) score table (entry 0 = -inf: rank 0 = "no token") | root records | match indices
    double* const score_tab = reinterpret_cast<double*>(smem);
    const uint32_t hot_bytes = 8u * (Q.n_hot + 1u);
    const __amdgpu_buffer_rsrc_t values = make_rsrc(reinterpret_cast<const unsigned char*>(Q.values) + hot_bytes, 8u * (Q.n_values - Q.n_hot));  // ranks beyond the LDS copy
    const uint2* rootc = reinterpret_cast<const uint2*>(smem + Q.root_off);
    unsigned char* wbase = smem + Q.idx_off + (size_t)wave * (PPL * kE5GroupBytes);
    {
        for (uint32_t i = threadIdx.x; i <= Q.n_hot; i += blockDim.x) score_tab[i] = Q.values[i];
        uint2* rw = reinterpret_cast<uint2*>(smem + Q.root_off);
        for (uint32_t i = threadIdx.x; i < 256u; i += blockDim.x) rw[i] = trie[(Q.root_base & ~255u) + i];
        if (Q.started && threadIdx.x == 0u) __hip_atomic_fetch_add(Q.started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __syncthreads();
    }
    const __amdgpu_buffer_rsrc_t cold_values = values;

    uint32_t s = 0, n = 0, p0 = 0;
    uint64_t beg = 0;
    bool live = false, need_new = true;
    const double ninf = -__builtin_huge_val();
    double acc = ninf, far = ninf;       // LONG: `far` collects the long candidates of the position 17..32 ahead
    uint32_t bpv = kNoStep, fbp = kFarCode;
    bool redo = false;                   // LONG: this row's sample is on the redo list already
    uint32_t wn[4 * PPL + 1];
    uint32_t pk = 0, pk_j = 0;
    bool pk_dirty = false;
#pragma unroll
    for (int q = 0; q <= 4 * PPL; ++q) wn[q] = 0;
    // this lane's two 32-byte columns of every group's index buffer (what it reads, and what it resets) and
    // the entry of its own start position's 1-byte token (what it writes, + 32 bytes per further byte)
    const uint32_t my_col = r * kE5RowStride + ((l - 1u) & 15u) * 32u;
    const uint32_t wbase_off = lds0 + Q.idx_off + wave * (PPL * kE5GroupBytes);  // this wave's match-index buffers, LDS byte offset
    const uint32_t wr_off = wbase_off + r * kE5RowStride + l * 2u;                // ... this lane's entries as a walker, column bits clear
    const uint32_t col_off = wbase_off + my_col;                                  // ... and its 32 contiguous bytes as the relaxing lane
    const uint32_t list_off = lds0 + Q.list_off + wave * ((kE5LongBytes + 15u) & ~15u);  // LONG: this wave's list of long matches
    uint64_t seg[5] = {0, 0, 0, 0, 0};
    uint64_t t_last = P.stamps ? (uint64_t)__builtin_amdgcn_s_memtime() : 0;
    uint32_t iters = 0;

    uint64_t local_next = 0, local_end = 0;  // sample indices this row has claimed and not started yet
    for (;;) {
        {
            uint64_t k = ~0ull;
            const bool have_local = need_new && local_next < local_end;
            if (have_local) k = local_next++;
            const uint64_t kc = claim_rows_chunk(P.queue, need_new && !have_local, r, Q.claim_chunk);
            if (need_new && !have_local) {
                k = kc;
                local_next = kc + 1u;
                local_end = kc + Q.claim_chunk;
            }
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
            acc = (l == 0u) ? 0.0 : ninf;
            bpv = kNoStep;
            pk_dirty = false;
            if (LONG) {
                far = ninf;
                fbp = kFarCode;
                redo = false;
            }
        }
        const bool fresh_row = need_new;
        need_new = false;
        if (__builtin_amdgcn_ballot_w64(live) == 0) break;
        iters++;
        E5_STAMP(0)  // sample switching

        // ---- text window (as encode4_kernel)
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

        // every (start, len) starts as "no token": index 0 = the table's -inf entry
#pragma unroll
        for (int g = 0; g < PPL; ++g) {
            uint4* mine = reinterpret_cast<uint4*>(wbase + g * kE5GroupBytes + my_col);
            mine[0] = make_uint4(0, 0, 0, 0);
            mine[1] = make_uint4(0, 0, 0, 0);
        }
        if (LONG && lane == 0u) lds_st<uint32_t>(list_off, 0u);  // this iteration's long matches
        __builtin_amdgcn_wave_barrier();
        E5_STAMP(1)  // text window, reset

        // ---- match: 64 * PPL walks over the 8-byte records
        uint32_t pg[PPL], maxd[PPL];
        uint32_t wlane[PPL];
        bool alive[PPL];
        uint2 rec[PPL];   // record of every walk's current depth (the root's children come from LDS) ...
        uint32_t c[PPL];  // ... and the text byte it must carry
#pragma unroll
        for (int g = 0; g < PPL; ++g) {
            pg[g] = p0 + 16u * g + l;
            const uint32_t rem = (live && pg[g] < n) ? (n - pg[g]) : 0u;
            maxd[g] = rem < LM ? rem : LM;
            alive[g] = maxd[g] > 0 && !(P.flags & 1u);
            wlane[g] = wr_off + g * kE5GroupBytes;
            c[g] = bytes[g][0] & 0xFFu;
            rec[g] = rootc[(Q.root_base ^ c[g]) & 255u];
        }
        {
            // (opaque per trip: as a loop invariant the sixteen column offsets of a lane get hoisted out of the
            // sample loop, kept in registers and spilled)
            uint32_t l32 = l * 32u;
            asm volatile("" : "+v"(l32));
            WalkCtx<PPL> W{trie_b, s, l32, P.dropout, P.seed};
            Res5 no_res{};
            Walk5<DROPOUT, LONG, false, PPL, 0, COLD && PPL < 4>::run(W, no_res, bytes, maxd, pg, wlane, alive, rec, c);  // (four positions per lane: the geometry of batches bound by their chains, where the stagger is worth more — 512 MiB 9.67 -> 10.11 ms with SKIP)
        }
        if (LONG) {
#pragma unroll
            for (int g = 0; g < PPL; ++g)
                e5_long_tail<DROPOUT>(trie_b, P.text + (live ? beg + pg[g] + 16u : 0), rec[g].x, alive[g], maxd[g], lane, (uint32_t)g, list_off,
                                      P.dropout, P.seed, s, pg[g]);
        }
        __builtin_amdgcn_wave_barrier();
        E5_STAMP(2)  // walk
        {   // the following block's text window lands while the relax runs
            const uint32_t* __restrict__ np = wp + 4 * PPL;
#pragma unroll
            for (int q = 0; q <= 4 * PPL; ++q) wn[q] = np[q];
        }

        // ---- relax: a lane's 16 match indices (ranks) are 32 contiguous bytes; the scores come from the table
        // (the walks — latency chains — issue ahead of the relaxing waves of their SIMD, which fill the gaps:
        // 11.28 -> 11.06 ms; the other way round 11.23 -> 11.53; TGX_FLAGS=32 with TGX_DEBUG=1 switches it off)
        if (!(P.flags & 32u)) __builtin_amdgcn_s_setprio(0);
        uint32_t fin[PPL];
        bool reached[PPL];
        if (!(P.flags & 2u)) {
            // the scores of four steps at a time, requested a quarter group ahead of the steps that use them
            uint32_t iw[8], iwn[8];
            auto load_iw = [&](int g, uint32_t (&w)[8]) {
                const u32x4_t ia = lds_ld<u32x4_t>(col_off + g * kE5GroupBytes), ib = lds_ld<u32x4_t>(col_off + g * kE5GroupBytes + 16u);
                w[0] = ia.x; w[1] = ia.y; w[2] = ia.z; w[3] = ia.w;
                w[4] = ib.x; w[5] = ib.y; w[6] = ib.z; w[7] = ib.w;
            };
            u32x2_t sa[4], sb[4];
            load_iw(0, iw);
            if (!COLD) e5_scores_issue<4>(lds0, iw, 0, sa);
#pragma unroll
            for (int g = 0; g < PPL; ++g) {
                uint32_t (&cw)[8] = (g & 1) ? iwn : iw;
                uint32_t (&nw)[8] = (g & 1) ? iw : iwn;
                fin[g] = kNoStep;
                uint32_t fhi = 0xFFF00000u;
                bool slow = false;   // LONG: some lane holds a long candidate, or a long match waits in the list (wave-uniform)
                uint32_t n_list = 0;
                if (LONG) {
                    n_list = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_ld<uint32_t>(list_off));
                    slow = n_list != 0u || __builtin_amdgcn_ballot_w64((uint32_t)((uint64_t)__double_as_longlong(far) >> 32) != 0xFFF00000u) != 0;
                }
                if (LONG && slow) {
                    // ---- the steps with the second accumulator, then the long matches that start in this group
                    auto rank_of = [&](int u) { return (u & 1) ? (cw[u >> 1] >> 16) : (cw[u >> 1] & 0xFFFFu); };
                    double fval = ninf;
                    relax5l_step<0>(e5_score_one<COLD>(lds0, cold_values, rank_of(0), hot_bytes), acc, bpv, far, fbp, fin[g], fval);
                    relax5l_step<1>(e5_score_one<COLD>(lds0, cold_values, rank_of(1), hot_bytes), acc, bpv, far, fbp, fin[g], fval);
                    relax5l_step<2>(e5_score_one<COLD>(lds0, cold_values, rank_of(2), hot_bytes), acc, bpv, far, fbp, fin[g], fval);
                    relax5l_step<3>(e5_score_one<COLD>(lds0, cold_values, rank_of(3), hot_bytes), acc, bpv, far, fbp, fin[g], fval);
                    relax5l_step<4>(e5_score_one<COLD>(lds0, cold_values, rank_of(4), hot_bytes), acc, bpv, far, fbp, fin[g], fval);
                    relax5l_step<5>(e5_score_one<COLD>(lds0, cold_values, rank_of(5), hot_bytes), acc, bpv, far, fbp, fin[g], fval);
                    relax5l_step<6>(e5_score_one<COLD>(lds0, cold_values, rank_of(6), hot_bytes), acc, bpv, far, fbp, fin[g], fval);
                    relax5l_step<7>(e5_score_one<COLD>(lds0, cold_values, rank_of(7), hot_bytes), acc, bpv, far, fbp, fin[g], fval);
                    relax5l_step<8>(e5_score_one<COLD>(lds0, cold_values, rank_of(8), hot_bytes), acc, bpv, far, fbp, fin[g], fval);
                    relax5l_step<9>(e5_score_one<COLD>(lds0, cold_values, rank_of(9), hot_bytes), acc, bpv, far, fbp, fin[g], fval);
                    relax5l_step<10>(e5_score_one<COLD>(lds0, cold_values, rank_of(10), hot_bytes), acc, bpv, far, fbp, fin[g], fval);
                    relax5l_step<11>(e5_score_one<COLD>(lds0, cold_values, rank_of(11), hot_bytes), acc, bpv, far, fbp, fin[g], fval);
                    relax5l_step<12>(e5_score_one<COLD>(lds0, cold_values, rank_of(12), hot_bytes), acc, bpv, far, fbp, fin[g], fval);
                    relax5l_step<13>(e5_score_one<COLD>(lds0, cold_values, rank_of(13), hot_bytes), acc, bpv, far, fbp, fin[g], fval);
                    relax5l_step<14>(e5_score_one<COLD>(lds0, cold_values, rank_of(14), hot_bytes), acc, bpv, far, fbp, fin[g], fval);
                    relax5l_step<15>(e5_score_one<COLD>(lds0, cold_values, rank_of(15), hot_bytes), acc, bpv, far, fbp, fin[g], fval);
                    // After the 16 steps lane j of a row accumulates position + 16 + j in `acc` and + 32 + j in `far`
                    // (positions counted from this group's first); entry (start lane, depth d) is the token of d + 1
                    // bytes that starts at the start lane's position of group eg: it ends 17..47 positions on.
                    const uint32_t cnt = n_list < kE5LongCap ? n_list : kE5LongCap;
#pragma unroll 1
                    for (uint32_t i = 0; i < cnt; ++i) {
                        const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_ld<uint32_t>(list_off + 8u + i * 4u));
                        if (((e >> 6) & 3u) != (uint32_t)g) continue;  // wave-uniform: a match that starts in another group
                        const uint32_t src = e & 63u, d = (e >> 8) & 31u;
                        const double score = e5_score_one<COLD>(lds0, cold_values, e >> 16, hot_bytes);
                        const double cand = readlane_f64(fval, src) + score;  // model.rs:98
                        const uint32_t tt = (src & 15u) + d + 1u;             // end position - this group's first: 17 .. 47
                        const bool mine = (lane >> 4) == (src >> 4) && l == (tt & 15u);
                        const bool to_far = tt >= 32u;
                        const double curv = to_far ? far : acc;
                        const uint32_t curc = to_far ? fbp : bpv;
                        // longer token = earlier start: it wins an equal score; a near winner is always shorter
                        const bool longer = !(curc & kFarCode) || d > (curc & 31u);
                        const bool take = mine && (cand > curv || (cand == curv && longer));
                        if (take && to_far) {
                            far = cand;
                            fbp = kFarCode | d;
                        }
                        if (take && !to_far) {
                            acc = cand;
                            bpv = kFarCode | d;
                        }
                    }
                    reached[g] = fval > ninf;
                    if (g + 1 < PPL) {  // what the ordinary steps of the next group expect to find requested
                        load_iw(g + 1, nw);
                        if (!COLD) e5_scores_issue<4>(lds0, nw, 0, sa);
                    }
                    continue;
                }
                if (COLD) {
                    // both halves' cold values are requested before the first step: the loads sit behind branches, so
                    // the compiler waits for all of them at the first use — one exposed L2 round trip per group, not two
                    double sv[8], sw[8];
                    e5_scores_cold<8>(lds0, cold_values, cw, 0, hot_bytes, sv);
                    e5_scores_cold<8>(lds0, cold_values, cw, 8, hot_bytes, sw);
                    if (g + 1 < PPL) load_iw(g + 1, nw);
                    relax5_step<0>(sv[0], acc, bpv, fin[g], fhi);
                    relax5_step<1>(sv[1], acc, bpv, fin[g], fhi);
                    relax5_step<2>(sv[2], acc, bpv, fin[g], fhi);
                    relax5_step<3>(sv[3], acc, bpv, fin[g], fhi);
                    relax5_step<4>(sv[4], acc, bpv, fin[g], fhi);
                    relax5_step<5>(sv[5], acc, bpv, fin[g], fhi);
                    relax5_step<6>(sv[6], acc, bpv, fin[g], fhi);
                    relax5_step<7>(sv[7], acc, bpv, fin[g], fhi);
                    relax5_step<8>(sw[0], acc, bpv, fin[g], fhi);
                    relax5_step<9>(sw[1], acc, bpv, fin[g], fhi);
                    relax5_step<10>(sw[2], acc, bpv, fin[g], fhi);
                    relax5_step<11>(sw[3], acc, bpv, fin[g], fhi);
                    relax5_step<12>(sw[4], acc, bpv, fin[g], fhi);
                    relax5_step<13>(sw[5], acc, bpv, fin[g], fhi);
                    relax5_step<14>(sw[6], acc, bpv, fin[g], fhi);
                    relax5_step<15>(sw[7], acc, bpv, fin[g], fhi);
                } else {
                    e5_scores_issue<4>(lds0, cw, 4, sb);
                    if (g + 1 < PPL) load_iw(g + 1, nw);
                    relax5_step<0>(e5_score_value(sa[0]), acc, bpv, fin[g], fhi);
                    relax5_step<1>(e5_score_value(sa[1]), acc, bpv, fin[g], fhi);
                    relax5_step<2>(e5_score_value(sa[2]), acc, bpv, fin[g], fhi);
                    relax5_step<3>(e5_score_value(sa[3]), acc, bpv, fin[g], fhi);
                    e5_scores_issue<4>(lds0, cw, 8, sa);
                    relax5_step<4>(e5_score_value(sb[0]), acc, bpv, fin[g], fhi);
                    relax5_step<5>(e5_score_value(sb[1]), acc, bpv, fin[g], fhi);
                    relax5_step<6>(e5_score_value(sb[2]), acc, bpv, fin[g], fhi);
                    relax5_step<7>(e5_score_value(sb[3]), acc, bpv, fin[g], fhi);
                    e5_scores_issue<4>(lds0, cw, 12, sb);
                    relax5_step<8>(e5_score_value(sa[0]), acc, bpv, fin[g], fhi);
                    relax5_step<9>(e5_score_value(sa[1]), acc, bpv, fin[g], fhi);
                    relax5_step<10>(e5_score_value(sa[2]), acc, bpv, fin[g], fhi);
                    relax5_step<11>(e5_score_value(sa[3]), acc, bpv, fin[g], fhi);
                    if (g + 1 < PPL) e5_scores_issue<4>(lds0, nw, 0, sa);
                    relax5_step<12>(e5_score_value(sb[0]), acc, bpv, fin[g], fhi);
                    relax5_step<13>(e5_score_value(sb[1]), acc, bpv, fin[g], fhi);
                    relax5_step<14>(e5_score_value(sb[2]), acc, bpv, fin[g], fhi);
                    relax5_step<15>(e5_score_value(sb[3]), acc, bpv, fin[g], fhi);
                }
                reached[g] = fhi != 0xFFF00000u;
            }
        } else {
#pragma unroll
            for (int g = 0; g < PPL; ++g) {
                fin[g] = kNoStep;
                reached[g] = true;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (!(P.flags & 32u)) __builtin_amdgcn_s_setprio(2);
        E5_STAMP(3)  // relax

        // ---- back-pointer bytes (encode4_kernel's packed, permuted layout), next block
        uint8_t* const bpw = P.bp8 + bp8_base(beg, s);
#pragma unroll
        for (int g = 0; g < PPL; ++g)
            if (live && pg[g] >= 1u && pg[g] <= n) {
                const uint32_t b = reached[g] ? (LONG ? e5_winner_len_m1(fin[g], l) : ((l - fin[g] - 1u) & 15u)) : 0xFFu;
                const uint32_t j = pg[g] - 1u, kq = (j >> 4) & 3u;
                pk = (kq == 0u) ? b : (pk | (b << (8u * kq)));
                pk_j = j;
                pk_dirty = true;
                if (kq == 3u) {
                    __builtin_nontemporal_store(pk, reinterpret_cast<uint32_t*>(bpw + (bp8_perm(j) & ~3u)));
                    pk_dirty = false;
                }
            }
        if (LONG) {  // a list that filled up dropped matches, whichever row they belonged to: encode2_kernel redoes these samples
            const uint32_t n_all = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_ld<uint32_t>(list_off));
            if (n_all > kE5LongCap) {
                if (live && !redo && l == 0u) P.redo_list[atomicAdd(P.redo_count, 1ULL)] = s;
                redo = true;
            }
        }
        if (live && n - p0 < SPAN && pk_dirty) {
            __builtin_nontemporal_store(pk, reinterpret_cast<uint32_t*>(bpw + (bp8_perm(pk_j) & ~3u)));
            pk_dirty = false;
        }
        if (live) {
            const uint32_t left = n - p0;
            if (left < SPAN) {
#pragma unroll
                for (int g = 0; g < PPL; ++g)
                    if (left == 16u * g + l) P.status[s] = (n == 0u || reached[g]) ? 1u : 0u;
                need_new = true;
            } else {
                p0 += SPAN;
            }
        }
        E5_STAMP(4)  // stores, bookkeeping
    }
    if (P.stamps && lane == 0) {
        unsigned long long* o = P.stamps + (size_t)(blockIdx.x * (blockDim.x >> 6) + wave) * 8u;
        for (int i = 0; i < 5; ++i) o[i] = seg[i];
        o[5] = iters;
    }
}

// ---- encode6_kernel: long samples ---------------------------------------------------------------------------
// A sample is a serial chain: in encode5_kernel a row walks 64 positions, relaxes them, walks the next 64, so a
// 64 KiB sample takes 1024 x (walk + relax) = 6.8 ms however idle the chip is, and batches of a few hundred MB
// (the per-GPU shards of the prune and merge passes, a 10 MB batch) are bound by their longest samples.  Only
// the relaxation is serial — the walks of different positions are independent — so this kernel keeps FOUR long
// samples per block, one per 16-lane row of the RELAXER wave (wave 0), and gives each of them kE6Walkers WALKER
// waves that fill a ring of match-index buffers (64 positions per trip, the walk of encode5_kernel) ahead of
// the relaxer, which consumes the trips in order: a sample's chain is 16 relax steps per 16 positions and
// nothing else.  Hand-over inside the block through LDS words, one set per row (E6Ctrl):
//   epoch        relaxer -> walkers of the row: a new sample's (index, begin, length, first trip) is published
//   walk_done[i] walker -> relaxer: ring slot i holds trip (value - 1)
//   relax_done   relaxer -> walkers: trips consumed so far (slot t % G is free once t < relax_done + G)
//   ack[j]       walker j -> relaxer: the epoch it has read; the next sample's fields are written only after all
//                of the row's walkers have read the current ones (a walker without a trip in a short sample
//                could otherwise still be reading them)
// Trip ids of a row run on across its samples, so the words never need resetting.  LDS operations of a wave are
// performed in order and the LDS is one pipeline per CU, so a flag written after the data is seen after the
// data.  The four rows advance in lockstep (the relaxer waits until every live row's next trip is there): rows
// have walkers of their own and the same cost per trip, and the order of the samples is longest-first, so
// rows of a block finish close to each other.  Every wait is on a word that another wave of the same block
// advances without waiting for the waiter: walkers wait for `epoch` and `relax_done` (the relaxer advances both
// after waiting only for walk_done words of trips whose slots are free), the relaxer for `walk_done`.
// Same back-pointer bytes and status as encode5_kernel (its relaxation), same trace kernel.
constexpr uint32_t kE6Walkers = 3;               // walker waves per sample
constexpr uint32_t kE6Slots = kE6Walkers + 2u;   // ring slots per sample
struct E6Ctrl {
    uint32_t epoch;       // 0 = nothing yet, 0xFFFFFFFF = no more samples for this row
    uint32_t s, n, trip0;
    uint64_t beg;
    uint32_t relax_done;
    uint32_t pad_;
    uint32_t walk_done[8];
    uint32_t ack[4];      // walker j -> relaxer: the epoch whose (s, n, beg, trip0) it has read
    uint32_t pool_used[8];  // COLD builds: pool entries of ring slot i handed out by its walker (Res5)
    uint32_t pad2_[4];
};
constexpr uint32_t kE6Done = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t lds_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// the four score values of a quarter group (steps 4 q .. 4 q + 3) of a lane: 16-bit ranks in iw
template <bool COLD>
__device__ __forceinline__ void e6_load_quarter(uint32_t tab, __amdgpu_buffer_rsrc_t cold, uint32_t hot_bytes,
                                                const uint32_t (&iw)[8], int q, double (&sv)[4]) {
    if (COLD) {
        e5_scores_cold<4>(tab, cold, iw, 4 * q, hot_bytes, sv);
    } else {
        u32x2_t sp[4];
        e5_scores_issue<4>(tab, iw, 4 * q, sp);
#pragma unroll
        for (int u = 0; u < 4; ++u) sv[u] = e5_score_value(sp[u]);
    }
}

template <bool DROPOUT, bool COLD>
__global__ __launch_bounds__(64 * (1 + 4 * kE6Walkers), 8) void encode6_kernel(EncodeParams P, Encode5Params Q) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr uint32_t LM = 16;
    constexpr uint32_t K = kE6Walkers, G = kE6Slots;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t l = lane & 15u, r = lane >> 4;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint2* __restrict__ trie = reinterpret_cast<const uint2*>(Q.trie8);
    const __amdgpu_buffer_rsrc_t trie_b = make_rsrc(Q.trie8, Q.trie_bytes);
    const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>(smem);  // LDS byte offset of the dynamic LDS (0 here)
    double* const score_tab = reinterpret_cast<double*>(smem);
    // LDS table: entry 0 = -inf ("no token"), COLD builds: 4 * G * Q.pool pool entries (Res5), then the n_hot hottest values
    const uint32_t shift = COLD ? 4u * G * Q.pool : 0u;
    const uint32_t hot_bytes = 8u * (1u + shift + Q.n_hot);
    const __amdgpu_buffer_rsrc_t values = make_rsrc(reinterpret_cast<const unsigned char*>(Q.values) + 8u * (Q.n_hot + 1u), 8u * (Q.n_values - Q.n_hot));  // ranks beyond the LDS copy
    const uint2* rootc = reinterpret_cast<const uint2*>(smem + Q.root_off);
    E6Ctrl* const ctrl_all = reinterpret_cast<E6Ctrl*>(smem + Q.ctrl_off);
    {
        if (threadIdx.x == 0u) score_tab[0] = Q.values[0];
        for (uint32_t i = threadIdx.x; i < shift; i += blockDim.x) score_tab[1u + i] = Q.values[0];
        for (uint32_t i = threadIdx.x; i < Q.n_hot; i += blockDim.x) score_tab[1u + shift + i] = Q.values[1u + i];
        uint2* rw = reinterpret_cast<uint2*>(smem + Q.root_off);
        for (uint32_t i = threadIdx.x; i < 256u; i += blockDim.x) rw[i] = trie[(Q.root_base & ~255u) + i];
        uint32_t* cw = reinterpret_cast<uint32_t*>(ctrl_all);
        for (uint32_t i = threadIdx.x; i < 4u * sizeof(E6Ctrl) / 4u; i += blockDim.x) cw[i] = 0u;
        // the ring starts as "no token" everywhere: rows without a sample read it while the others relax
        uint4* ring = reinterpret_cast<uint4*>(smem + Q.ring_off);
        for (uint32_t i = threadIdx.x; i < 4u * G * kE5GroupBytes / 16u; i += blockDim.x) ring[i] = make_uint4(0, 0, 0, 0);
        __syncthreads();
    }
    const double ninf = -__builtin_huge_val();

    if (wave == 0u) {
        // ================= relaxer: row r of this wave owns one sample at a time =================
        // (its chain of dependent steps is what a sample waits for: it issues ahead of the walkers on its SIMD)
        __builtin_amdgcn_s_setprio(3);
        E6Ctrl* const ctrl = ctrl_all + r;
        uint32_t s = 0, n = 0, n_trips = 0, tau = 0, trip0 = 0, epoch = 0, slot = 0;
        uint64_t beg = 0;
        bool live = false, need_new = true, exhausted = false;
        double acc = ninf;
        uint32_t bpv = kNoStep, pk = 0, pk_j = 0;
        bool pk_dirty = false;
        for (;;) {
            if (__builtin_amdgcn_ballot_w64(need_new) != 0) {
                for (;;) {  // the row's walkers have read the sample that is being replaced
                    const bool read = !need_new || epoch == 0u || l >= K || lds_load(&ctrl->ack[l]) == epoch;
                    if (__builtin_amdgcn_ballot_w64(!read) == 0) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                const uint64_t k = claim_rows(P.queue, need_new && !exhausted, r);
                if (need_new) {
                    live = !exhausted && k < P.n_samples;
                    if (live) {
                        s = P.order[k];
                        beg = P.offs[s];
                        n = (uint32_t)(P.offs[s + 1] - beg);
                        n_trips = n / 64u + 1u;
                        tau = 0;
                        acc = (l == 0u) ? 0.0 : ninf;
                        bpv = kNoStep;
                        pk_dirty = false;
                        if (l == 0u) {
                            ctrl->s = s;
                            ctrl->n = n;
                            ctrl->beg = beg;
                            ctrl->trip0 = trip0;
                        }
                    } else {
                        exhausted = true;  // the queue only grows: nothing more for this row
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (need_new) {
                    if (live) epoch++;
                    if (l == 0u) lds_store(&ctrl->epoch, live ? epoch : kE6Done);
                }
                need_new = false;
            }
            if (__builtin_amdgcn_ballot_w64(live) == 0) break;

            // ---- every live row's next trip must be in its ring slot
            const uint32_t t = trip0 + tau;
            for (;;) {
                const bool ready = !live || lds_load(&ctrl->walk_done[slot]) == t + 1u;
                if (__builtin_amdgcn_ballot_w64(!ready) == 0) break;
                __builtin_amdgcn_s_sleep(1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

            const uint32_t p0 = tau * 64u;
            const unsigned char* sbase = smem + Q.ring_off + (r * G + slot) * kE5GroupBytes + ((l - 1u) & 15u) * 32u;
            // index words one group ahead, score values half a group ahead of the steps that use them
            auto load_iw = [&](int g, uint32_t (&iw)[8]) {
                const uint4* ip = reinterpret_cast<const uint4*>(sbase + g * kE5RowStride);
                const uint4 ia = ip[0], ib = ip[1];
                iw[0] = ia.x; iw[1] = ia.y; iw[2] = ia.z; iw[3] = ia.w;
                iw[4] = ib.x; iw[5] = ib.y; iw[6] = ib.z; iw[7] = ib.w;
            };
            const double acc_in = acc;
            const uint32_t bpv_in = bpv;
            uint32_t fin[4];
            bool reached[4];
            // COLD builds: the walkers have moved the trip's cold values into the slot's pool (Res5) unless the pool ran
            // out — only then (or without a pool) the steps run on the variant that reads values from L2, whose
            // clamp-and-branch per entry costs the chain ~1.7x even when nothing is cold.
            auto relax_trip = [&](auto cold_tag) {
                constexpr bool C = decltype(cold_tag)::value;
                uint32_t iwa[8], iwb[8];
                double sva[4], svb[4];
                load_iw(0, iwa);
                e6_load_quarter<C>(lds0, values, hot_bytes, iwa, 0, sva);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint32_t (&iw)[8] = (g & 1) ? iwb : iwa;
                    uint32_t (&iwn)[8] = (g & 1) ? iwa : iwb;
                    fin[g] = kNoStep;
                    uint32_t fhi = 0xFFF00000u;
                    e6_load_quarter<C>(lds0, values, hot_bytes, iw, 1, svb);
                    if (g < 3) load_iw(g + 1, iwn);
                    relax5_step<0>(sva[0], acc, bpv, fin[g], fhi);
                    relax5_step<1>(sva[1], acc, bpv, fin[g], fhi);
                    relax5_step<2>(sva[2], acc, bpv, fin[g], fhi);
                    relax5_step<3>(sva[3], acc, bpv, fin[g], fhi);
                    e6_load_quarter<C>(lds0, values, hot_bytes, iw, 2, sva);
                    relax5_step<4>(svb[0], acc, bpv, fin[g], fhi);
                    relax5_step<5>(svb[1], acc, bpv, fin[g], fhi);
                    relax5_step<6>(svb[2], acc, bpv, fin[g], fhi);
                    relax5_step<7>(svb[3], acc, bpv, fin[g], fhi);
                    e6_load_quarter<C>(lds0, values, hot_bytes, iw, 3, svb);
                    relax5_step<8>(sva[0], acc, bpv, fin[g], fhi);
                    relax5_step<9>(sva[1], acc, bpv, fin[g], fhi);
                    relax5_step<10>(sva[2], acc, bpv, fin[g], fhi);
                    relax5_step<11>(sva[3], acc, bpv, fin[g], fhi);
                    if (g < 3) e6_load_quarter<C>(lds0, values, hot_bytes, iwn, 0, sva);
                    relax5_step<12>(svb[0], acc, bpv, fin[g], fhi);
                    relax5_step<13>(svb[1], acc, bpv, fin[g], fhi);
                    relax5_step<14>(svb[2], acc, bpv, fin[g], fhi);
                    relax5_step<15>(svb[3], acc, bpv, fin[g], fhi);
                    reached[g] = fhi != 0xFFF00000u;
                }
            };
            bool unresolved = false;
            if (COLD) unresolved = live && (Q.pool == 0u || lds_load(&ctrl->pool_used[slot]) > Q.pool);
            if (COLD && __builtin_expect(__builtin_amdgcn_ballot_w64(unresolved) != 0, 0)) relax_trip(std::true_type{});
            else relax_trip(std::false_type{});
            if (!live) {  // a row without a sample went through the motions on an all-"no token" slot
                acc = acc_in;
                bpv = bpv_in;
            }
            // the slot's entries are consumed: its walker-to-be may reset and refill it
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (live && l == 0u) lds_store(&ctrl->relax_done, t + 1u);
            // back-pointer bytes (encode5_kernel's packed, permuted layout)
            if (live) {
                uint8_t* const bpw = P.bp8 + bp8_base(beg, s);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint32_t pgq = p0 + 16u * g + l;
                    if (pgq >= 1u && pgq <= n) {
                        const uint32_t b = reached[g] ? ((l - fin[g] - 1u) & 15u) : 0xFFu;
                        const uint32_t j = pgq - 1u, kq = (j >> 4) & 3u;
                        pk = (kq == 0u) ? b : (pk | (b << (8u * kq)));
                        pk_j = j;
                        pk_dirty = true;
                        if (kq == 3u) {
                            __builtin_nontemporal_store(pk, reinterpret_cast<uint32_t*>(bpw + (bp8_perm(j) & ~3u)));
                            pk_dirty = false;
                        }
                    }
                }
                const uint32_t left = n - p0;
                if (left < 64u) {  // the sample's last trip
                    if (pk_dirty) __builtin_nontemporal_store(pk, reinterpret_cast<uint32_t*>(bpw + (bp8_perm(pk_j) & ~3u)));
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        if (left == 16u * g + l) P.status[s] = (n == 0u || reached[g]) ? 1u : 0u;
                    trip0 += n_trips;
                    need_new = true;
                } else {
                    tau++;
                }
                slot = (slot + 1u == G) ? 0u : slot + 1u;
            }
        }
        return;
    }

    // ================= walkers: wave w serves row (w - 1) & 3 as its walker number (w - 1) >> 2 =================
    const uint32_t rw = (wave - 1u) & 3u, jw = (wave - 1u) >> 2;
    E6Ctrl* const ctrl = ctrl_all + rw;
    uint32_t seen = 0;
    for (;;) {
        uint32_t e;
        while ((e = lds_load(&ctrl->epoch)) == seen) __builtin_amdgcn_s_sleep(2);
        if (e == kE6Done) break;
        seen = e;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const uint32_t s = ctrl->s, n = ctrl->n, trip0 = ctrl->trip0;
        const uint64_t beg = ctrl->beg;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the reads above stay above the acknowledgement
        if (lane == 0u) lds_store(&ctrl->ack[jw], e);
        const uint32_t n_trips = n / 64u + 1u;
        const uint32_t tau0 = (jw + K - trip0 % K) % K;
        uint32_t slot = (trip0 + tau0) % G;
        for (uint32_t tau = tau0; tau < n_trips; tau += K, slot = (slot + K >= G) ? slot + K - G : slot + K) {
            const uint32_t t = trip0 + tau;
            while (t >= lds_load(&ctrl->relax_done) + G) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const uint32_t p0 = tau * 64u, pgl = p0 + lane;
            // text window of this lane's position
            const uintptr_t addr = reinterpret_cast<uintptr_t>(P.text + beg + pgl);
            const uint32_t sh = (uint32_t)(addr & 3u);
            const uint32_t* __restrict__ wp = reinterpret_cast<const uint32_t*>(addr & ~uintptr_t(3));
            uint32_t wv[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) wv[q] = wp[q];
            uint32_t bytes[1][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bytes[0][q] = __builtin_amdgcn_alignbyte(wv[q + 1], wv[q], sh);
            // reset this lane's reader column of its group, then walk
            const uint32_t slot_off = Q.ring_off + (rw * G + slot) * kE5GroupBytes;
            {
                uint4* mine = reinterpret_cast<uint4*>(smem + slot_off + r * kE5RowStride + ((l - 1u) & 15u) * 32u);
                mine[0] = make_uint4(0, 0, 0, 0);
                mine[1] = make_uint4(0, 0, 0, 0);
                if (COLD && lane == 0u) lds_store(&ctrl->pool_used[slot], 0u);
            }
            __builtin_amdgcn_wave_barrier();
            uint32_t pg[1] = {pgl}, maxd[1], wlane[1], c[1];
            bool alive[1];
            uint2 rec[1];
            const uint32_t rem = pgl < n ? (n - pgl) : 0u;
            maxd[0] = rem < LM ? rem : LM;
            alive[0] = maxd[0] > 0;
            wlane[0] = lds0 + slot_off + r * kE5RowStride + l * 2u;
            c[0] = bytes[0][0] & 0xFFu;
            rec[0] = rootc[(Q.root_base ^ c[0]) & 255u];
            uint32_t l32 = l * 32u;
            asm volatile("" : "+v"(l32));
            WalkCtx<1> W{trie_b, s, l32, P.dropout, P.seed};
            Res5 R{};
            if (COLD) {
                R.cold = values;
                R.tab = lds0;
                R.shift = shift;
                R.hot_max = shift + Q.n_hot;
                R.used_off = lds0 + Q.ctrl_off + rw * (uint32_t)sizeof(E6Ctrl) + (uint32_t)offsetof(E6Ctrl, pool_used) + slot * 4u;
                R.first = 1u + (rw * G + slot) * Q.pool;
                R.cap = Q.pool;
            }
            Walk5<DROPOUT, false, COLD, 1, 0>::run(W, R, bytes, maxd, pg, wlane, alive, rec, c);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0u) lds_store(&ctrl->walk_done[slot], t + 1u);
        }
    }
}


// ---- estep5_fwd_kernel: the forward sweep of the E-step (src/lattice.rs:259-291) on this file's structure ----------
// estep4l_fwd_kernel (estep4l.hip) walks 16-byte records at one position per lane and carries f64 weights through
// an 8 KiB match buffer per wave.  For vocabularies that have 8-byte records (at most 65 535 distinct score values,
// tokens of at most 16 bytes) the forward sweep is this kernel instead: encode5_kernel's staggered walk over the
// ranked records and its 2 KiB match-index buffers, the table holding w = exp(score value) by rank (entry 0 = 0: "no
// token"), and estep4l_fwd_kernel's steps and epilogue — a[p] = sum a[start] * w in the linear domain with a power-of-
// two exponent per block of 16 positions, alpha / alpha_exp / z written in its layout, the same range check.  Same
// matches, same weights (exp of the same doubles, taken on the host), same operations in the same order: the arrays
// are bit-identical to estep4l_fwd_kernel's (tests/test_estep_pairs_gpu.py), the backward kernel is unchanged.
template <bool DROPOUT, bool COLD, int PPL>
__global__ __launch_bounds__(1024, (PPL == 4 ? 1 : 4)) void estep5_fwd_kernel(Estep4Params P, Encode5Params Q) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr uint32_t LM = 16;
    constexpr uint32_t SPAN = 16u * PPL;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t l = lane & 15u, r = lane >> 4;
    const uint32_t wave = threadIdx.x >> 6;
    const uint2* __restrict__ trie = reinterpret_cast<const uint2*>(Q.trie8);
    const __amdgpu_buffer_rsrc_t trie_b = make_rsrc(Q.trie8, Q.trie_bytes);
    const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>(smem);
    double* const w_tab = reinterpret_cast<double*>(smem);  // [0] = 0.0 ("no token"), [r] = exp(score value of rank r)
    const uint32_t hot_bytes = 8u * (Q.n_hot + 1u);
    const __amdgpu_buffer_rsrc_t cold_values = make_rsrc(reinterpret_cast<const unsigned char*>(Q.values) + hot_bytes, 8u * (Q.n_values - Q.n_hot));
    const uint2* rootc = reinterpret_cast<const uint2*>(smem + Q.root_off);
    unsigned char* wbase = smem + Q.idx_off + (size_t)wave * (PPL * kE5GroupBytes);
    {
        for (uint32_t i = threadIdx.x; i <= Q.n_hot; i += blockDim.x) w_tab[i] = Q.values[i];
        uint2* rw = reinterpret_cast<uint2*>(smem + Q.root_off);
        for (uint32_t i = threadIdx.x; i < 256u; i += blockDim.x) rw[i] = trie[(Q.root_base & ~255u) + i];
        __syncthreads();
    }
    uint32_t s = 0, n = 0, p0 = 0, smp = 0;
    uint64_t beg = 0, sbase = 0, ebase = 0;
    bool live = false, need_new = true;
    double acc = 0.0;
    int erow = 0;  // alpha_true = acc * 2^erow for every accumulator of the row
    double zsum = 0.0;
    uint32_t wn[4 * PPL + 1];
#pragma unroll
    for (int q = 0; q <= 4 * PPL; ++q) wn[q] = 0;
    const uint32_t my_col = r * kE5RowStride + ((l - 1u) & 15u) * 32u;
    const uint32_t wbase_off = lds0 + Q.idx_off + wave * (PPL * kE5GroupBytes);
    const uint32_t wr_off = wbase_off + r * kE5RowStride + l * 2u;
    const uint32_t col_off = wbase_off + my_col;
    uint64_t local_next = 0, local_end = 0;
    for (;;) {
        {
            uint64_t k = ~0ull;
            const bool have_local = need_new && local_next < local_end;
            if (have_local) k = local_next++;
            const uint64_t kc = claim_rows_chunk(P.queue_fwd, need_new && !have_local, r, Q.claim_chunk);
            if (need_new && !have_local) {
                k = kc;
                local_next = kc + 1u;
                local_end = kc + Q.claim_chunk;
            }
            if (need_new) {
                live = k < P.n_snips;
                if (live) {
                    s = P.order[k];
                    beg = P.soffs[s];
                    n = (uint32_t)(P.soffs[s + 1] - beg);
                    ebase = (beg >> 4) + s;
                    if (DROPOUT) {
                        smp = P.snip_sample[s];
                        sbase = P.snip_base[s];
                    }
                }
                p0 = 0;
                acc = (l == 0u) ? 1.0 : 0.0;  // BOS (lattice.rs:96-101, 267)
                erow = 0;
            }
        }
        const bool fresh_row = need_new;
        need_new = false;
        if (__builtin_amdgcn_ballot_w64(live) == 0) break;

        // ---- text window, reset of this lane's columns (as encode5_kernel)
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
#pragma unroll
        for (int g = 0; g < PPL; ++g) {
            uint4* mine = reinterpret_cast<uint4*>(wbase + g * kE5GroupBytes + my_col);
            mine[0] = make_uint4(0, 0, 0, 0);
            mine[1] = make_uint4(0, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();

        // ---- match
        uint32_t pg[PPL], pgh[PPL], maxd[PPL], wlane[PPL], c[PPL];
        bool alive[PPL];
        uint2 rec[PPL];
#pragma unroll
        for (int g = 0; g < PPL; ++g) {
            pg[g] = p0 + 16u * g + l;
            pgh[g] = (uint32_t)sbase + pg[g];  // position in the SAMPLE: what the keep rule hashes (samples < 4 GiB: host)
            const uint32_t rem = (live && pg[g] < n) ? (n - pg[g]) : 0u;
            maxd[g] = rem < LM ? rem : LM;
            alive[g] = maxd[g] > 0;
            wlane[g] = wr_off + g * kE5GroupBytes;
            c[g] = bytes[g][0] & 0xFFu;
            rec[g] = rootc[(Q.root_base ^ c[g]) & 255u];
        }
        {
            uint32_t l32 = l * 32u;
            asm volatile("" : "+v"(l32));
            WalkCtx<PPL> W{trie_b, smp, l32, P.dropout, P.seed, true};
            Res5 no_res{};
            Walk5<DROPOUT, false, false, PPL, 0>::run(W, no_res, bytes, maxd, pgh, wlane, alive, rec, c);
        }
        __builtin_amdgcn_wave_barrier();
        {   // the following trip's text window lands while the steps run
            const uint32_t* __restrict__ np = wp + 4 * PPL;
#pragma unroll
            for (int q = 0; q <= 4 * PPL; ++q) wn[q] = np[q];
        }

        // ---- forward recursion (estep4l_fwd_kernel's steps): 16 static steps per group of 16 positions, then the row is
        // rescaled; the values of a group are stored under the exponent in effect while they were finalised
#pragma unroll
        for (int g = 0; g < PPL; ++g) {
            uint32_t iw[8];
            {
                const u32x4_t ia = lds_ld<u32x4_t>(col_off + g * kE5GroupBytes), ib = lds_ld<u32x4_t>(col_off + g * kE5GroupBytes + 16u);
                iw[0] = ia.x; iw[1] = ia.y; iw[2] = ia.z; iw[3] = ia.w;
                iw[4] = ib.x; iw[5] = ib.y; iw[6] = ib.z; iw[7] = ib.w;
            }
            double sv[8], sw[8];
            if (COLD) {
                e5_scores_cold<8>(lds0, cold_values, iw, 0, hot_bytes, sv);
                e5_scores_cold<8>(lds0, cold_values, iw, 8, hot_bytes, sw);
            } else {
                u32x2_t sa[8], sb[8];
                e5_scores_issue<8>(lds0, iw, 0, sa);
                e5_scores_issue<8>(lds0, iw, 8, sb);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    sv[u] = e5_score_value(sa[u]);
                    sw[u] = e5_score_value(sb[u]);
                }
            }
            double fin = 0.0;
            e4l_fwd_step<0>(sv[0], acc, fin);
            e4l_fwd_step<1>(sv[1], acc, fin);
            e4l_fwd_step<2>(sv[2], acc, fin);
            e4l_fwd_step<3>(sv[3], acc, fin);
            e4l_fwd_step<4>(sv[4], acc, fin);
            e4l_fwd_step<5>(sv[5], acc, fin);
            e4l_fwd_step<6>(sv[6], acc, fin);
            e4l_fwd_step<7>(sv[7], acc, fin);
            e4l_fwd_step<8>(sw[0], acc, fin);
            e4l_fwd_step<9>(sw[1], acc, fin);
            e4l_fwd_step<10>(sw[2], acc, fin);
            e4l_fwd_step<11>(sw[3], acc, fin);
            e4l_fwd_step<12>(sw[4], acc, fin);
            e4l_fwd_step<13>(sw[5], acc, fin);
            e4l_fwd_step<14>(sw[6], acc, fin);
            e4l_fwd_step<15>(sw[7], acc, fin);
            if (live && pg[g] <= n) {
                P.alpha[beg + s + pg[g]] = fin;
                // a position nothing was pushed to (lattice.rs:255), an underflow or an overflow: the pass belongs to
                // the log-domain kernels
                if (!(fin > 0.0 && fin <= 1.7976931348623157e308)) atomicMax(P.range_flag, 1ULL);
                if (l == 0u) P.alpha_exp[ebase + (pg[g] >> 4)] = erow;
                if (pg[g] == n) {  // z = log alpha_true[n] (lattice.rs:290-291)
                    const double z = log(fin) + (double)erow * 0.6931471805599453;
                    P.zarr[s] = z;
                    zsum += z;
                    const double az = fabs(z);  // !z.is_normal() panics in the reference (prune.rs:90-96)
                    if (!(az >= 2.2250738585072014e-308 && az <= 1.7976931348623157e308)) atomicMin(P.err_snip, (unsigned long long)s);
                }
            }
            const int e = row_max_exponent(acc);
            if (e > -100000) {
                acc = ldexp(acc, -e);
                erow += e;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (live) {
            if (n - p0 < SPAN) need_new = true;  // position n lies in this trip: the snippet is done
            else p0 += SPAN;
        }
    }
    if (zsum != 0.0) atomicAdd(P.logz_sum, zsum);
}

typedef void (*estep5_fn)(Estep4Params, Encode5Params);
static estep5_fn pick_estep5(bool dropout, bool cold, int ppl) {
#define TGX_E5F(D, C)                                                                        \
    (ppl == 1 ? estep5_fwd_kernel<D, C, 1> : ppl == 2 ? estep5_fwd_kernel<D, C, 2> : ppl == 3 ? estep5_fwd_kernel<D, C, 3> : estep5_fwd_kernel<D, C, 4>)
    if (dropout) return cold ? TGX_E5F(true, true) : TGX_E5F(true, false);
    return cold ? TGX_E5F(false, true) : TGX_E5F(false, false);
#undef TGX_E5F
}
hipError_t estep5_waves_per_simd(bool dropout, bool cold, int ppl, int* out) {
    hipFuncAttributes attr;
    hipError_t e = hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(pick_estep5(dropout, cold, ppl)));
    if (e != hipSuccess) return e;
    const int regs = (attr.numRegs + 7) & ~7;
    *out = regs > 0 ? (512 / regs > 8 ? 8 : 512 / regs) : 8;
    return hipSuccess;
}
hipError_t launch_estep5_fwd(const Estep4Params& p, Encode5Params q, bool cold, int ppl, int waves, uint32_t blocks, hipStream_t stream) {
    const uint32_t lds = encode5_lds_layout(q.n_hot, false, waves, ppl, &q.list_off, &q.root_off, &q.idx_off);
    if (lds > 160u * 1024u || q.n_hot > q.n_values || (!cold && q.n_hot != q.n_values) || ppl < 1 || ppl > 4) return hipErrorInvalidValue;
    estep5_fn fn = pick_estep5(p.dropout > 0.0, cold, ppl);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(64u * (uint32_t)waves), lds, stream, p, q);
    return hipGetLastError();
}

typedef void (*encode5_fn)(EncodeParams, Encode5Params);
static encode5_fn pick_encode5(bool dropout, bool cold, int ppl, bool long_tokens) {
    if (long_tokens) {  // tokens of 17..32 bytes: four positions per lane only
        if (cold) return dropout ? encode5_kernel<true, true, 4, true> : encode5_kernel<false, true, 4, true>;
        return dropout ? encode5_kernel<true, false, 4, true> : encode5_kernel<false, false, 4, true>;
    }
    if (cold) {
        if (ppl == 1) return dropout ? encode5_kernel<true, true, 1, false> : encode5_kernel<false, true, 1, false>;
        if (ppl == 2) return dropout ? encode5_kernel<true, true, 2, false> : encode5_kernel<false, true, 2, false>;
        if (ppl == 3) return dropout ? encode5_kernel<true, true, 3, false> : encode5_kernel<false, true, 3, false>;
        return dropout ? encode5_kernel<true, true, 4, false> : encode5_kernel<false, true, 4, false>;
    }
    if (ppl == 1) return dropout ? encode5_kernel<true, false, 1, false> : encode5_kernel<false, false, 1, false>;
    if (ppl == 2) return dropout ? encode5_kernel<true, false, 2, false> : encode5_kernel<false, false, 2, false>;
    if (ppl == 3) return dropout ? encode5_kernel<true, false, 3, false> : encode5_kernel<false, false, 3, false>;
    return dropout ? encode5_kernel<true, false, 4, false> : encode5_kernel<false, false, 4, false>;
}

// LDS of one block of `waves` waves: score table (-inf and n_hot values), the waves' lists of long matches (LONG
// builds), root records, match indices
uint32_t encode5_lds_layout(uint32_t n_hot, bool long_tokens, int waves, int ppl, uint32_t* list_off, uint32_t* root_off, uint32_t* idx_off) {
    const uint32_t lo = (8u * (n_hot + 1u) + 15u) & ~15u;
    const uint32_t score_bytes = lo + (long_tokens ? (uint32_t)waves * ((kE5LongBytes + 15u) & ~15u) : 0u);
    if (list_off) *list_off = lo;
    const uint32_t ro = (score_bytes + 15u) & ~15u;
    const uint32_t io = (ro + 2048u + 511u) & ~511u;  // 512-byte aligned: see kE5RowStride
    if (root_off) *root_off = ro;
    if (idx_off) *idx_off = io;
    return io + (uint32_t)waves * (uint32_t)ppl * kE5GroupBytes;
}
// the largest table that leaves a block of `waves` waves within `budget` bytes of LDS
uint32_t encode5_max_hot(bool long_tokens, int waves, int ppl, uint32_t budget) {
    const uint32_t fixed = encode5_lds_layout(0u, long_tokens, waves, ppl, nullptr, nullptr, nullptr) + 512u;  // alignment slack
    return budget > fixed + 64u ? (budget - fixed) / 8u : 0u;
}

hipError_t encode5_waves_per_simd(bool dropout, bool cold, int ppl, bool long_tokens, int* out) {
    hipFuncAttributes attr;
    hipError_t e = hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(pick_encode5(dropout, cold, ppl, long_tokens)));
    if (e != hipSuccess) return e;
    const int regs = (attr.numRegs + 7) & ~7;
    *out = regs > 0 ? (512 / regs > 8 ? 8 : 512 / regs) : 8;
    return hipSuccess;
}

// LDS of an encode6_kernel block: score table, root records, control words of the four rows, four rings of
// kE6Slots match-index buffers
uint32_t encode6_lds_layout(uint32_t n_hot, uint32_t pool, uint32_t* root_off, uint32_t* ctrl_off, uint32_t* ring_off) {
    const uint32_t score_bytes = 8u * (1u + 4u * kE6Slots * pool + n_hot);
    const uint32_t ro = (score_bytes + 15u) & ~15u;
    const uint32_t co = ro + 2048u;
    const uint32_t go = (co + 4u * (uint32_t)sizeof(E6Ctrl) + 511u) & ~511u;
    if (root_off) *root_off = ro;
    if (ctrl_off) *ctrl_off = co;
    if (ring_off) *ring_off = go;
    return go + 4u * kE6Slots * kE5GroupBytes;
}
uint32_t encode6_pool_total(uint32_t pool) { return 4u * kE6Slots * pool; }
uint32_t encode6_max_hot(uint32_t budget, uint32_t pool) {
    const uint32_t fixed = encode6_lds_layout(0u, pool, nullptr, nullptr, nullptr) + 512u;
    return budget > fixed + 64u ? (budget - fixed) / 8u : 0u;
}
hipError_t launch_encode6(const EncodeParams& p, Encode5Params q, bool cold, uint32_t blocks, hipStream_t stream) {
    q.ring_slots = kE6Slots;
    if (!cold) q.pool = 0u;
    const uint32_t lds = encode6_lds_layout(q.n_hot, q.pool, &q.root_off, &q.ctrl_off, &q.ring_off);
    if (lds > 160u * 1024u || q.n_hot > q.n_values || (!cold && q.n_hot != q.n_values) ||
        (uint64_t)q.n_values + 4ull * kE6Slots * q.pool > 65535ull)
        return hipErrorInvalidValue;
    auto fn = cold ? (p.dropout > 0.0 ? encode6_kernel<true, true> : encode6_kernel<false, true>)
                   : (p.dropout > 0.0 ? encode6_kernel<true, false> : encode6_kernel<false, false>);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(64u * (1u + 4u * kE6Walkers)), lds, stream, p, q);
    return hipGetLastError();
}

hipError_t launch_encode5(const EncodeParams& p, Encode5Params q, bool cold, int ppl, bool long_tokens, int waves, uint32_t blocks, uint32_t min_lds, hipStream_t stream) {
    uint32_t lds = encode5_lds_layout(q.n_hot, long_tokens, waves, ppl, &q.list_off, &q.root_off, &q.idx_off);
    if (lds > 160u * 1024u || q.n_hot > q.n_values || (!cold && q.n_hot != q.n_values) || (long_tokens && ppl != 4)) return hipErrorInvalidValue;
    encode5_fn fn = pick_encode5(p.dropout > 0.0, cold, ppl, long_tokens);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    if (lds < min_lds && min_lds <= 160u * 1024u) lds = min_lds;  // (a launch that wants its CUs to itself)
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(64u * (uint32_t)waves), lds, stream, p, q);
    return hipGetLastError();
}

}  // namespace tgx
