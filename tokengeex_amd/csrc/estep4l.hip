// E-step, four snippets per wave, in the LINEAR domain with exact power-of-two rescaling.
//
// estep4.hip follows the reference literally: alpha / beta are log-probabilities folded with
// log_sum_exp (src/lattice.rs:259-287, 321-333), i.e. one exp and one log per (position, token length)
// — about 250 VALU instructions per relaxation step, so that its kernels are bound by the serial
// arithmetic of the longest snippet.  The same quantities can be carried as PROBABILITIES:
//
//     a[p] = sum over the tokens ending at p of a[start] * w(token),   w = exp(score),
//
// one multiply-add per (position, length), if the common scale of a row of 16 accumulators is kept in an
// integer exponent that is adjusted once per block of 16 positions (a multiplication by 2^k is exact).
// With alpha_true[p] = a[p] * 2^Ea(block of p) and beta_true[q] = b[q] * 2^Eb:
//
//     z        = log(a[n]) + Ea(n) * ln 2                              (lattice.rs:290-291)
//     marginal = alpha_true[p] * w * beta_true[q] / alpha_true[n]       (lattice.rs:305-307)
//              = ldexp((a[p] / a[n]) * (w * b[q]), Ea(p) - Ea(n) + Eb)
//
// The results differ from the reference's only by rounding, and by less than two log-domain
// evaluations differ from each other: every operation here is exact to 1 ulp of a PROBABILITY, while a
// log_sum_exp rounds a log-probability of magnitude |z| ~ 2.4 x bytes.  The reference's
// `vmax > vmin + 50` shortcut drops terms below e^-50 of the running sum; here they are added
// (a relative difference below 2e-22).
//
// What this representation cannot express is lattice.rs:255's corner: a position NO token ends at keeps
// alpha = 0.0 there, a log-probability of zero in the middle of values around -2.4 x bytes.  It needs a
// text byte that is no token by itself (0xFF in the recipes' vocabularies, i.e. invalid UTF-8).  The
// forward kernel therefore checks every finalised value (0 = nothing pushed or underflow, inf = overflow)
// and raises range_flag; the host then redoes the pass with the log-domain kernels of estep4.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_common.h"
#include "kernels.h"

namespace tgx {

constexpr uint32_t kE4LEntries = 1024;  // 64 rows x 16 columns

// ---- tokens of 17..32 bytes (vocabularies after `merge`): LONG = true builds ---------------------------------
// Such tokens are rare matches, so the kernels keep their 16-column structure and add, per lane, a second
// accumulator `far` for the position 17..32 ahead (the scheme of encode4l_kernel; sums commute, so no order
// has to be kept here): the walk goes on past depth 16 only where the trie continues, a match of 17..32 bytes
// goes to a per-wave overflow list in LDS instead of the match buffer, lane U restarts from `far` at step U, and
// after the 16 steps of a block the list is applied: (value of the source position) x w into `acc` (target
// 17..31 ahead of the block's first position) or `far` (32..47).  A list that fills up raises range_flag = 2
// and the host redoes the pass with the generic kernel.
constexpr uint32_t kE4LOvfCap = 62;                            // entries per wave and block
constexpr uint32_t kE4LOvfBytes = 16u + kE4LOvfCap * 16u;      // {pad} + entries {lane, len | slot << 8 (unused fwd), f64 w}

template <int U>
__device__ __forceinline__ void e4l_fwd_step_long(double sv, double& acc, double& far, double& fin) {
    constexpr uint64_t MU = kRowLane0 << U;
    fin = sel_f64(MU, acc, fin);
    const double best = row_bcast_f64<U>(acc);
    const double cand = best * sv;
    acc = sel_f64(MU, far, acc) + cand;      // lane U restarts from the long contributions to position p0 + U + 16
    far = sel_f64(MU, 0.0, far);             // its `far` now stands for position p0 + U + 32
}

// Hot slots (P.n_hot): expected counts of the first (hottest-first order) slots of the reversed trie are summed in
// the block's LDS and flushed once at the end; the global f64 atomics of the others (executed at the memory
// side, not in L2: ~30 G scattered adds per second chip-wide whatever their locality — 888 M of them per GiB
// with 2048 hot slots, TCC_EA0_ATOMIC in profiles/r02/t_passes_1GiB) are what bounds the backward kernel.
// So the LDS goes to the hot set: a hot entry is {sum, w = exp(score)} and the match buffer holds the SLOT of a
// match only (4 bytes per (position, length) instead of 12: the weight is read from the hot entry, or from the
// trie record in HBM for a cold slot), which leaves room for 7 000 hot slots instead of 2 000.
constexpr uint32_t kNoSlot = 0xFFFFFFFFu;

// (cs_cur, cs_nxt: a[p] / a[n] of the lane's two start positions, already scaled by 2^(Ea(p) - Ea(n) + Eb) — a power of
// two, so cand * cs is ldexp(cand * c, e + eb) bit for bit while both stay in the normal range; one ldexp per block and
// start position instead of one per step: round 3)
template <int U>
__device__ __forceinline__ void e4l_bwd_step(double sv, uint32_t hv, double cs_cur, double cs_nxt,
                                             double* __restrict__ expected_slot, double2* hot, uint32_t n_hot, bool cold_ok, double& acc) {
    constexpr uint64_t MU = kRowLane0 << U;
    constexpr uint64_t WRAPPED = (uint64_t)((1u << (U + 1)) - 1u) * kRowLane0;  // lanes l <= U: position y0 + 16 + l
    const double best = row_bcast_f64<U>(acc);  // b[q] of the source (end) position
    const double cand = best * sv;              // w * b[q]
    const double cs = sel_f64(WRAPPED, cs_nxt, cs_cur);
    if (sv != 0.0) {  // lattice.rs:305-307
        const double mg = cand * cs;
        if (hv < n_hot)
            atomicAdd(&hot[hv].x, mg);  // ds_add_f64
        else if (cold_ok)
            atomicAdd(&expected_slot[hv], mg);
    }
    acc = sel_f64(MU, cand, acc + cand);
}

template <int U>
__device__ __forceinline__ void e4l_bwd_step_long(double sv, uint32_t hv, double c_cur, double c_nxt, int e_cur, int e_nxt,
                                                  int eb, double* __restrict__ expected_slot, double2* hot, uint32_t n_hot, bool cold_ok,
                                                  double& acc, double& far, double& fin) {
    constexpr uint64_t MU = kRowLane0 << U;
    constexpr uint64_t WRAPPED = (uint64_t)((1u << (U + 1)) - 1u) * kRowLane0;
    fin = sel_f64(MU, acc, fin);                // b[q] of the source position is final now (the long matches use it)
    const double best = row_bcast_f64<U>(acc);
    const double cand = best * sv;
    const double c = sel_f64(WRAPPED, c_nxt, c_cur);
    const int e = (int)sel_u32(WRAPPED, (uint32_t)e_nxt, (uint32_t)e_cur);
    if (sv != 0.0) {
        const double mg = ldexp(cand * c, e + eb);
        if (hv < n_hot)
            atomicAdd(&hot[hv].x, mg);
        else if (cold_ok)
            atomicAdd(&expected_slot[hv], mg);
    }
    acc = sel_f64(MU, far, acc) + cand;         // lane U restarts from the long contributions to its next position
    far = sel_f64(MU, 0.0, far);
}

// PPL = positions per lane and trip (1, 2, 4): a row advances 16 * PPL positions per trip of the dependent
// gather chain.  The relaxation is cheap here (one multiply-add per step), so more positions per lane shorten
// the serial chain of a long snippet almost proportionally, at the price of LDS (8 KiB * PPL per wave
// forward, 12 KiB * PPL backward) and with it waves per CU; the host picks PPL from the shape of the pass.
template <bool DROPOUT, int PPL, bool LONG>
__global__ __launch_bounds__(1024) void estep4l_fwd_kernel(Estep4Params P) {
    static_assert(!LONG || PPL == 1, "the long-token build runs one position per lane");
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr uint32_t LM = 16;
    constexpr uint32_t SPAN = 16u * PPL;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t l = lane & 15u, r = lane >> 4;
    const uint32_t wave = threadIdx.x >> 6;
    const uint4* __restrict__ trie = reinterpret_cast<const uint4*>(P.trie_fwd);  // records carry w = exp(score)
    double* sc = reinterpret_cast<double*>(smem + (size_t)wave * (PPL * kE4LEntries * 8u));
    // LONG: this wave's overflow list, behind the match buffers of the block's waves
    unsigned char* ovf = smem + (size_t)(blockDim.x >> 6) * (PPL * kE4LEntries * 8u) + (size_t)wave * kE4LOvfBytes;
    constexpr uint32_t LMX = LONG ? 32u : 16u;  // longest token

    uint32_t s = 0, n = 0, p0 = 0, smp = 0;
    uint64_t beg = 0, sbase = 0, ebase = 0;
    bool live = false, need_new = true;
    double acc = 0.0, far = 0.0;
    int erow = 0;  // alpha_true = acc * 2^erow for every accumulator of the row
    double zsum = 0.0;

    for (;;) {
        const uint64_t k = claim_rows(P.queue_fwd, need_new, r);  // longest-first, dynamic
        if (need_new) {
            live = k < P.n_snips;
            if (live) {
                s = P.order[k];
                beg = P.soffs[s];
                n = (uint32_t)(P.soffs[s + 1] - beg);
                ebase = (beg >> 4) + s;  // this snippet's slice of the block-exponent array
                if (DROPOUT) {
                    smp = P.snip_sample[s];
                    sbase = P.snip_base[s];
                }
            }
            p0 = 0;
            acc = (l == 0u) ? 1.0 : 0.0;  // BOS: alpha = 0 in the log domain (lattice.rs:96-101, 267)
            far = 0.0;
            erow = 0;
        }
        need_new = false;
        if (__builtin_amdgcn_ballot_w64(live) == 0) break;

        // ---- match (forward trie), as in encode4_kernel; "no token" = weight 0
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
            double2* grp = reinterpret_cast<double2*>(sc + g * kE4LEntries);
#pragma unroll
            for (int q = 0; q < 8; ++q) grp[q * 64 + lane] = make_double2(0.0, 0.0);
        }
        uint32_t pg[PPL], maxd[PPL], cur[PPL], base[PPL];
        bool alive[PPL];
#pragma unroll
        for (int g = 0; g < PPL; ++g) {
            pg[g] = p0 + 16u * g + l;
            const uint32_t rem = (live && pg[g] < n) ? (n - pg[g]) : 0u;
            maxd[g] = rem < LMX ? rem : LMX;
            cur[g] = 0;
            base[g] = P.root_fwd;
            alive[g] = maxd[g] > 0;
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
                uint4 rec[PPL];
                uint32_t t[PPL];
#pragma unroll
                for (int g = 0; g < PPL; ++g) {  // all of this lane's gathers of the step are issued before any is consumed
                    const uint32_t c = (bytes[g][d >> 2] >> ((d & 3) * 8)) & 0xFFu;
                    t[g] = alive[g] ? (base[g] ^ c) : 0u;
                    rec[g] = trie[t[g]];
                }
#pragma unroll
                for (int g = 0; g < PPL; ++g) asm volatile("" : "+v"(rec[g].x), "+v"(rec[g].y), "+v"(rec[g].z), "+v"(rec[g].w));
#pragma unroll
                for (int g = 0; g < PPL; ++g) {
                    alive[g] = alive[g] && rec[g].x == cur[g];
                    if (alive[g]) {
                        cur[g] = t[g];
                        base[g] = rec[g].y & 0x7FFFFFFFu;
                        bool term = (rec[g].y >> 31) != 0u;
                        if (DROPOUT) {  // model.rs:48: skipped iff len > 1 && rand < dropout
                            if (term && d >= 1) term = !(dropout_u01(P.seed, smp, sbase + pg[g], (uint32_t)d + 1u) < P.dropout);
                        }
                        if (term) (sc + g * kE4LEntries + lane * LM)[((uint32_t)d + l) & 15u] = __hiloint2double((int)rec[g].w, (int)rec[g].z);
                    }
                }
            }
        }
        uint32_t n_ovf = 0;  // wave-uniform: overflow entries of this block
        if (LONG) {
            // walks that are still alive after 16 bytes (the trie continues: a token of 17..32 bytes may follow)
            bool more = alive[0] && maxd[0] > 16u;
            if (__builtin_amdgcn_ballot_w64(more) != 0) {
                uint32_t b2[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) b2[q] = __builtin_amdgcn_alignbyte(wp[4 + q + 1], wp[4 + q], sh);  // text bytes 16..31
                for (uint32_t d = 16; d < 32u; ++d) {
                    more = more && d < maxd[0];
                    if (__builtin_amdgcn_ballot_w64(more) == 0) break;
                    const uint32_t word = (d & 8u) ? ((d & 4u) ? b2[3] : b2[2]) : ((d & 4u) ? b2[1] : b2[0]);
                    const uint32_t c = (word >> ((d & 3u) * 8u)) & 0xFFu;
                    const uint32_t t = more ? (base[0] ^ c) : 0u;
                    const uint4 rec = load_rec(trie, t);
                    more = more && rec.x == cur[0];
                    bool term = more && (rec.y >> 31) != 0u;
                    if (more) {
                        cur[0] = t;
                        base[0] = rec.y & 0x7FFFFFFFu;
                    }
                    if (DROPOUT) {
                        if (term) term = !(dropout_u01(P.seed, smp, sbase + pg[0], d + 1u) < P.dropout);
                    }
                    const uint64_t tm = __builtin_amdgcn_ballot_w64(term);
                    if (tm != 0) {  // wave-uniform
                        const uint32_t e = n_ovf + __builtin_amdgcn_mbcnt_hi((uint32_t)(tm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)tm, 0u));
                        n_ovf += (uint32_t)__builtin_popcountll(tm);
                        if (term && e < kE4LOvfCap) *reinterpret_cast<uint4*>(ovf + 16u + e * 16u) = make_uint4(lane, d + 1u, rec.z, rec.w);
                    }
                }
                if (n_ovf > kE4LOvfCap) {  // the pass goes to the generic kernel
                    atomicMax(P.range_flag, 2ULL);
                    n_ovf = kE4LOvfCap;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();

        // ---- forward recursion: 16 static steps per group of 16 positions, then the row is rescaled (its
        // largest accumulator goes to [0.5, 1)) — the values of a group are stored under the exponent
        // that was in effect while they were finalised
#pragma unroll
        for (int g = 0; g < PPL; ++g) {
            double fin = 0.0;
            const double* scr = sc + g * kE4LEntries + r * 256u + ((l - 1u) & 15u);
            double sv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) sv[u] = scr[u * 16];
            if (LONG) {
                e4l_fwd_step_long<0>(sv[0], acc, far, fin);
                e4l_fwd_step_long<1>(sv[1], acc, far, fin);
                e4l_fwd_step_long<2>(sv[2], acc, far, fin);
                e4l_fwd_step_long<3>(sv[3], acc, far, fin);
                e4l_fwd_step_long<4>(sv[4], acc, far, fin);
                e4l_fwd_step_long<5>(sv[5], acc, far, fin);
                e4l_fwd_step_long<6>(sv[6], acc, far, fin);
                e4l_fwd_step_long<7>(sv[7], acc, far, fin);
                e4l_fwd_step_long<8>(sv[8], acc, far, fin);
                e4l_fwd_step_long<9>(sv[9], acc, far, fin);
                e4l_fwd_step_long<10>(sv[10], acc, far, fin);
                e4l_fwd_step_long<11>(sv[11], acc, far, fin);
                e4l_fwd_step_long<12>(sv[12], acc, far, fin);
                e4l_fwd_step_long<13>(sv[13], acc, far, fin);
                e4l_fwd_step_long<14>(sv[14], acc, far, fin);
                e4l_fwd_step_long<15>(sv[15], acc, far, fin);
                // the long matches of this block: a[start] * w into the position 17..47 ahead of the block's first
                for (uint32_t e = 0; e < n_ovf; ++e) {
                    const uint4 ent = *reinterpret_cast<const uint4*>(ovf + 16u + e * 16u);  // same address in every lane
                    const uint32_t src = (uint32_t)__builtin_amdgcn_readfirstlane((int)ent.x);
                    const uint32_t tgt = (src & 15u) + (uint32_t)__builtin_amdgcn_readfirstlane((int)ent.y);  // 17..47
                    const double contrib = readlane_f64(fin, src) * __hiloint2double((int)ent.w, (int)ent.z);
                    const bool mine = r == (src >> 4) && l == (tgt & 15u);
                    if (tgt < 32u)
                        acc += mine ? contrib : 0.0;
                    else
                        far += mine ? contrib : 0.0;
                }
            } else {
                e4l_fwd_step<0>(sv[0], acc, fin);
                e4l_fwd_step<1>(sv[1], acc, fin);
                e4l_fwd_step<2>(sv[2], acc, fin);
                e4l_fwd_step<3>(sv[3], acc, fin);
                e4l_fwd_step<4>(sv[4], acc, fin);
                e4l_fwd_step<5>(sv[5], acc, fin);
                e4l_fwd_step<6>(sv[6], acc, fin);
                e4l_fwd_step<7>(sv[7], acc, fin);
                e4l_fwd_step<8>(sv[8], acc, fin);
                e4l_fwd_step<9>(sv[9], acc, fin);
                e4l_fwd_step<10>(sv[10], acc, fin);
                e4l_fwd_step<11>(sv[11], acc, fin);
                e4l_fwd_step<12>(sv[12], acc, fin);
                e4l_fwd_step<13>(sv[13], acc, fin);
                e4l_fwd_step<14>(sv[14], acc, fin);
                e4l_fwd_step<15>(sv[15], acc, fin);
            }
            // a[pg] and the block's exponent -> scratch (snippet s: n + 1 values at soffs[s] + s)
            if (live && pg[g] <= n) {
                P.alpha[beg + s + pg[g]] = fin;
                // a position nothing was pushed to (lattice.rs:255: it counts as log-probability 0.0 there), an
                // underflow or an overflow: this pass belongs to the log-domain kernels
                if (!(fin > 0.0 && fin <= 1.7976931348623157e308)) atomicMax(P.range_flag, 1ULL);
                if (l == 0u) P.alpha_exp[ebase + (pg[g] >> 4)] = erow;
                if (pg[g] == n) {  // z = log alpha_true[n] (lattice.rs:290-291)
                    const double z = log(fin) + (double)erow * 0.6931471805599453;
                    P.zarr[s] = z;
                    zsum += z;
                    // !z.is_normal() panics in the reference (prune.rs:90-96)
                    const double az = fabs(z);
                    if (!(az >= 2.2250738585072014e-308 && az <= 1.7976931348623157e308))
                        atomicMin(P.err_snip, (unsigned long long)s);
                }
            }
            const int e = row_max_exponent(acc);
            if (e > -100000) {
                acc = ldexp(acc, -e);
                if (LONG) far = ldexp(far, -e);
                erow += e;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (live) {
            if (n - p0 < SPAN)  // position n lies in this trip: the snippet is done
                need_new = true;
            else
                p0 += SPAN;
        }
    }
    if (zsum != 0.0) atomicAdd(P.logz_sum, zsum);
}

template <bool DROPOUT, int PPL, bool LONG>
__global__ __launch_bounds__(1024) void estep4l_bwd_kernel(Estep4Params P) {
    static_assert(!LONG || PPL == 1, "the long-token build runs one position per lane");
    constexpr uint32_t LMX = LONG ? 32u : 16u;  // longest token
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr uint32_t LM = 16;
    constexpr uint32_t SPAN = 16u * PPL;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t l = lane & 15u, r = lane >> 4;
    const uint32_t wave = threadIdx.x >> 6;
    const uint4* __restrict__ trie = reinterpret_cast<const uint4*>(P.trie_rev);  // records carry w = exp(score)
    double2* hot = reinterpret_cast<double2*>(smem);  // n_hot + 1 entries {partial sum, w}, shared by the block
    const uint32_t n_hot = P.n_hot;  // slots summed in LDS (wave-uniform); entry n_hot is {0, 0}: "no token"
    unsigned char* wbase = smem + (n_hot + 1u) * 16u + (size_t)wave * (PPL * kE4LEntries * 4u);
    uint32_t* hl = reinterpret_cast<uint32_t*>(wbase);  // PPL groups of slots
    // LONG: this wave's overflow list, behind the match buffers of the block's waves
    unsigned char* ovf = smem + (n_hot + 1u) * 16u + (size_t)(blockDim.x >> 6) * (PPL * kE4LEntries * 4u) + (size_t)wave * kE4LOvfBytes;
    // expected counts go to one of n_replicas copies of the slot array (reduced afterwards):
    // a handful of very frequent tokens would otherwise serialise every wave's atomics
    double* __restrict__ expected_slot = P.expected_slot + (size_t)(blockIdx.x % P.n_replicas) * P.n_slots_rev;
    constexpr bool cold_ok = true;  // (a run-time switch here — TGX_FLAGS=8, the experiment of y_estep_bwd_without_cold_atomics.txt — cost four instructions per step: a spilled scalar flag)
    for (uint32_t i = threadIdx.x; i <= n_hot; i += blockDim.x)
        hot[i] = make_double2(0.0, (i < n_hot && i < P.n_slots_rev) ? reinterpret_cast<const double*>(P.trie_rev)[2u * i + 1u] : 0.0);
    __syncthreads();

    uint32_t s = 0, n = 0, y0 = 0, smp = 0;
    uint64_t beg = 0, sbase = 0, ebase = 0;
    bool live = false, need_new = true;
    double acc = 0.0, far = 0.0, inv_an = 0.0;
    double cfw[PPL + 1];  // a[p] / a[n] of this lane's start positions y0 + 16 g + l, g = 0 .. PPL
    int efw[PPL + 1];     // Ea(p) - Ea(n)
    int eb = 0, ean = 0;
#pragma unroll
    for (int g = 0; g <= PPL; ++g) {
        cfw[g] = 0.0;
        efw[g] = 0;
    }

    // (a[p] / a[n], Ea(p) - Ea(n)) of the start position with distance y from the end, (0, 0) past the start
    auto load_fwd = [&](uint32_t y, double& c, int& e) {
        if (live && y <= n) {
            const uint32_t p = n - y;
            c = P.alpha[beg + s + p] * inv_an;
            e = P.alpha_exp[ebase + (p >> 4)] - ean;
        } else {
            c = 0.0;
            e = 0;
        }
    };

    for (;;) {
        const uint64_t k = claim_rows(P.queue_bwd, need_new, r);  // longest-first, dynamic
        if (need_new) {
            live = k < P.n_snips;
            if (live) {
                s = P.order[k];
                beg = P.soffs[s];
                n = (uint32_t)(P.soffs[s + 1] - beg);
                ebase = (beg >> 4) + s;
                inv_an = 1.0 / P.alpha[beg + s + n];
                ean = P.alpha_exp[ebase + (n >> 4)];
                if (DROPOUT) {
                    smp = P.snip_sample[s];
                    sbase = P.snip_base[s];
                }
            }
            y0 = 0;
            acc = (l == 0u) ? 1.0 : 0.0;  // EOS: beta = 0 in the log domain
            far = 0.0;
            eb = 0;
#pragma unroll
            for (int g = 0; g <= PPL; ++g) load_fwd(16u * g + l, cfw[g], efw[g]);
        }
        need_new = false;
        if (__builtin_amdgcn_ballot_w64(live) == 0) break;

        // ---- match on the reversed text with the reversed-token trie: lane (r, l) owns the end positions
        // q = n - y, y = y0 + 16 g + l, and reads text[q - 1], text[q - 2], ...  The 16-byte windows
        // text[q - 16 .. q) of a lane's PPL positions are adjacent (16 bytes apart, descending with g), so
        // one span of 4 PPL + 1 dwords starting at the lowest covers them (the buffer has a 256-byte front
        // pad; windows of positions before the snippet's start are read but never used)
        uint32_t yq[PPL], qq[PPL], maxd[PPL], cur[PPL], base[PPL];
        bool alive[PPL];
#pragma unroll
        for (int g = 0; g < PPL; ++g) {
            yq[g] = y0 + 16u * g + l;
            qq[g] = (live && yq[g] < n) ? (n - yq[g]) : 0u;  // bytes available before q
            maxd[g] = qq[g] < LMX ? qq[g] : LMX;
            cur[g] = 0;
            base[g] = P.root_rev;
            alive[g] = maxd[g] > 0;
        }
        const int64_t low = live ? (int64_t)beg + (int64_t)n - (int64_t)(y0 + 16u * (PPL - 1) + l) - 16 : 0;
        const uintptr_t addr = reinterpret_cast<uintptr_t>(P.text) + (uintptr_t)(low < -240 ? -240 : low);
        const uint32_t sh = (uint32_t)(addr & 3u);
        const uint32_t* __restrict__ wp = reinterpret_cast<const uint32_t*>(addr & ~uintptr_t(3));
        uint32_t w[4 * PPL + 1];
#pragma unroll
        for (int j = 0; j <= 4 * PPL; ++j) w[j] = wp[j];
        uint32_t bytes[PPL][4];  // group g's window sits 16 (PPL - 1 - g) bytes above the lowest
#pragma unroll
        for (int g = 0; g < PPL; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                bytes[g][j] = __builtin_amdgcn_alignbyte(w[4 * (PPL - 1 - g) + j + 1], w[4 * (PPL - 1 - g) + j], sh);
#pragma unroll
        for (int g = 0; g < PPL; ++g) {
            uint4* grp = reinterpret_cast<uint4*>(hl + g * kE4LEntries);
#pragma unroll
            for (int j = 0; j < 4; ++j) grp[j * 64 + lane] = make_uint4(kNoSlot, kNoSlot, kNoSlot, kNoSlot);
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
                uint4 rec[PPL];
                uint32_t t[PPL];
#pragma unroll
                for (int g = 0; g < PPL; ++g) {
                    const uint32_t c = (bytes[g][(15 - d) >> 2] >> (((15 - d) & 3) * 8)) & 0xFFu;  // text[q - 1 - d]
                    t[g] = alive[g] ? (base[g] ^ c) : 0u;
                    rec[g] = trie[t[g]];
                }
#pragma unroll
                for (int g = 0; g < PPL; ++g) asm volatile("" : "+v"(rec[g].x), "+v"(rec[g].y), "+v"(rec[g].z), "+v"(rec[g].w));
#pragma unroll
                for (int g = 0; g < PPL; ++g) {
                    alive[g] = alive[g] && rec[g].x == cur[g];
                    if (alive[g]) {
                        cur[g] = t[g];
                        base[g] = rec[g].y & 0x7FFFFFFFu;
                        bool term = (rec[g].y >> 31) != 0u;
                        if (DROPOUT) {  // keyed by the token's START byte in the sample, as in the forward sweep
                            if (term && d >= 1)
                                term = !(dropout_u01(P.seed, smp, sbase + (uint64_t)(qq[g] - (uint32_t)d - 1u), (uint32_t)d + 1u) < P.dropout);
                        }
                        if (term) (hl + g * kE4LEntries + lane * LM)[((uint32_t)d + l) & 15u] = t[g];
                    }
                }
            }
        }
        uint32_t n_ovf = 0;  // wave-uniform: overflow entries of this block
        if (LONG) {
            bool more = alive[0] && maxd[0] > 16u;
            if (__builtin_amdgcn_ballot_w64(more) != 0) {
                // text[q - 32 .. q - 16): the 16 bytes below the window (front pad of 256 bytes: `low` >= -240)
                const uint32_t* __restrict__ wl = wp - 4;
                uint32_t b2[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) b2[j] = __builtin_amdgcn_alignbyte(wl[j + 1], wl[j], sh);
                for (uint32_t d = 16; d < 32u; ++d) {
                    more = more && d < maxd[0];
                    if (__builtin_amdgcn_ballot_w64(more) == 0) break;
                    const uint32_t bi = 31u - d;  // text[q - 1 - d] is byte 31 - d of [q - 32, q)
                    const uint32_t word = (bi & 8u) ? ((bi & 4u) ? b2[3] : b2[2]) : ((bi & 4u) ? b2[1] : b2[0]);
                    const uint32_t c = (word >> ((bi & 3u) * 8u)) & 0xFFu;
                    const uint32_t t = more ? (base[0] ^ c) : 0u;
                    const uint4 rec = load_rec(trie, t);
                    more = more && rec.x == cur[0];
                    bool term = more && (rec.y >> 31) != 0u;
                    if (more) {
                        cur[0] = t;
                        base[0] = rec.y & 0x7FFFFFFFu;
                    }
                    if (DROPOUT) {
                        if (term) term = !(dropout_u01(P.seed, smp, sbase + (uint64_t)(qq[0] - d - 1u), d + 1u) < P.dropout);
                    }
                    const uint64_t tm = __builtin_amdgcn_ballot_w64(term);
                    if (tm != 0) {  // wave-uniform
                        const uint32_t e = n_ovf + __builtin_amdgcn_mbcnt_hi((uint32_t)(tm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)tm, 0u));
                        n_ovf += (uint32_t)__builtin_popcountll(tm);
                        if (term && e < kE4LOvfCap) *reinterpret_cast<uint4*>(ovf + 16u + e * 16u) = make_uint4(lane | ((d + 1u) << 8), t, rec.z, rec.w);
                    }
                }
                if (n_ovf > kE4LOvfCap) {  // the pass goes to the generic kernel
                    atomicMax(P.range_flag, 2ULL);
                    n_ovf = kE4LOvfCap;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();

        // ---- backward recursion + marginals: 16 static steps per group, then the row is rescaled
#pragma unroll
        for (int g = 0; g < PPL; ++g) {
            const uint32_t* hlr = hl + g * kE4LEntries + r * 256u + ((l - 1u) & 15u);
            const double c_cur = cfw[g], c_nxt = cfw[g + 1];
            const int e_cur = efw[g], e_nxt = efw[g + 1];
            uint32_t hvs[16];
            double sv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) hvs[u] = hlr[u * 16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {  // w of the match: hot entry (entry n_hot = 0.0 for "no token") ...
                sv[u] = hot[hvs[u] < n_hot ? hvs[u] : n_hot].y;
                // ... or, for a slot outside the hot set, the trie record in HBM
                if (hvs[u] != kNoSlot && hvs[u] >= n_hot) sv[u] = reinterpret_cast<const double*>(P.trie_rev)[2u * hvs[u] + 1u];
            }
            if (LONG) {
                double fin = 0.0;
                e4l_bwd_step_long<0>(sv[0], hvs[0], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, n_hot, cold_ok, acc, far, fin);
                e4l_bwd_step_long<1>(sv[1], hvs[1], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, n_hot, cold_ok, acc, far, fin);
                e4l_bwd_step_long<2>(sv[2], hvs[2], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, n_hot, cold_ok, acc, far, fin);
                e4l_bwd_step_long<3>(sv[3], hvs[3], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, n_hot, cold_ok, acc, far, fin);
                e4l_bwd_step_long<4>(sv[4], hvs[4], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, n_hot, cold_ok, acc, far, fin);
                e4l_bwd_step_long<5>(sv[5], hvs[5], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, n_hot, cold_ok, acc, far, fin);
                e4l_bwd_step_long<6>(sv[6], hvs[6], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, n_hot, cold_ok, acc, far, fin);
                e4l_bwd_step_long<7>(sv[7], hvs[7], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, n_hot, cold_ok, acc, far, fin);
                e4l_bwd_step_long<8>(sv[8], hvs[8], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, n_hot, cold_ok, acc, far, fin);
                e4l_bwd_step_long<9>(sv[9], hvs[9], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, n_hot, cold_ok, acc, far, fin);
                e4l_bwd_step_long<10>(sv[10], hvs[10], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, n_hot, cold_ok, acc, far, fin);
                e4l_bwd_step_long<11>(sv[11], hvs[11], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, n_hot, cold_ok, acc, far, fin);
                e4l_bwd_step_long<12>(sv[12], hvs[12], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, n_hot, cold_ok, acc, far, fin);
                e4l_bwd_step_long<13>(sv[13], hvs[13], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, n_hot, cold_ok, acc, far, fin);
                e4l_bwd_step_long<14>(sv[14], hvs[14], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, n_hot, cold_ok, acc, far, fin);
                e4l_bwd_step_long<15>(sv[15], hvs[15], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, n_hot, cold_ok, acc, far, fin);
                // the long matches of this block: b[q] * w into the position 17..47 further from the end, and their
                // marginals (the lane that owns the match's end position has everything of its snippet at hand)
                for (uint32_t e = 0; e < n_ovf; ++e) {
                    const uint4 ent = *reinterpret_cast<const uint4*>(ovf + 16u + e * 16u);  // same address in every lane
                    const uint32_t src = (uint32_t)__builtin_amdgcn_readfirstlane((int)(ent.x & 0xFFu));
                    const uint32_t tgt = (src & 15u) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(ent.x >> 8));  // 17..47
                    const double w = __hiloint2double((int)ent.w, (int)ent.z);
                    const double contrib = readlane_f64(fin, src) * w;
                    if (lane == src) {  // lattice.rs:305-307 for this match: its start lies tgt positions before the block's last
                        const uint32_t pp = n - (y0 + tgt);
                        const double cp = P.alpha[beg + s + pp] * inv_an;
                        const int ep = P.alpha_exp[ebase + (pp >> 4)] - ean;
                        const double mg = ldexp(fin * w * cp, ep + eb);
                        if (ent.y < n_hot)
                            atomicAdd(&hot[ent.y].x, mg);
                        else if (cold_ok)
                            atomicAdd(&expected_slot[ent.y], mg);
                    }
                    const bool mine = r == (src >> 4) && l == (tgt & 15u);
                    if (tgt < 32u)
                        acc += mine ? contrib : 0.0;
                    else
                        far += mine ? contrib : 0.0;
                }
            } else {
                const double cs_cur = ldexp(c_cur, e_cur + eb), cs_nxt = ldexp(c_nxt, e_nxt + eb);
                e4l_bwd_step<0>(sv[0], hvs[0], cs_cur, cs_nxt, expected_slot, hot, n_hot, cold_ok, acc);
                e4l_bwd_step<1>(sv[1], hvs[1], cs_cur, cs_nxt, expected_slot, hot, n_hot, cold_ok, acc);
                e4l_bwd_step<2>(sv[2], hvs[2], cs_cur, cs_nxt, expected_slot, hot, n_hot, cold_ok, acc);
                e4l_bwd_step<3>(sv[3], hvs[3], cs_cur, cs_nxt, expected_slot, hot, n_hot, cold_ok, acc);
                e4l_bwd_step<4>(sv[4], hvs[4], cs_cur, cs_nxt, expected_slot, hot, n_hot, cold_ok, acc);
                e4l_bwd_step<5>(sv[5], hvs[5], cs_cur, cs_nxt, expected_slot, hot, n_hot, cold_ok, acc);
                e4l_bwd_step<6>(sv[6], hvs[6], cs_cur, cs_nxt, expected_slot, hot, n_hot, cold_ok, acc);
                e4l_bwd_step<7>(sv[7], hvs[7], cs_cur, cs_nxt, expected_slot, hot, n_hot, cold_ok, acc);
                e4l_bwd_step<8>(sv[8], hvs[8], cs_cur, cs_nxt, expected_slot, hot, n_hot, cold_ok, acc);
                e4l_bwd_step<9>(sv[9], hvs[9], cs_cur, cs_nxt, expected_slot, hot, n_hot, cold_ok, acc);
                e4l_bwd_step<10>(sv[10], hvs[10], cs_cur, cs_nxt, expected_slot, hot, n_hot, cold_ok, acc);
                e4l_bwd_step<11>(sv[11], hvs[11], cs_cur, cs_nxt, expected_slot, hot, n_hot, cold_ok, acc);
                e4l_bwd_step<12>(sv[12], hvs[12], cs_cur, cs_nxt, expected_slot, hot, n_hot, cold_ok, acc);
                e4l_bwd_step<13>(sv[13], hvs[13], cs_cur, cs_nxt, expected_slot, hot, n_hot, cold_ok, acc);
                e4l_bwd_step<14>(sv[14], hvs[14], cs_cur, cs_nxt, expected_slot, hot, n_hot, cold_ok, acc);
                e4l_bwd_step<15>(sv[15], hvs[15], cs_cur, cs_nxt, expected_slot, hot, n_hot, cold_ok, acc);
            }
            const int e = row_max_exponent(acc);
            if (e > -100000) {
                acc = ldexp(acc, -e);
                if (LONG) far = ldexp(far, -e);
                eb += e;
            }
        }
        __builtin_amdgcn_wave_barrier();

        if (live) {
            if (n - y0 < SPAN) {  // position 0 lies in this trip: the snippet is done
                need_new = true;
            } else {
                y0 += SPAN;
                cfw[0] = cfw[PPL];
                efw[0] = efw[PPL];
#pragma unroll
                for (int g = 1; g <= PPL; ++g) load_fwd(y0 + 16u * g + l, cfw[g], efw[g]);
            }
        }
    }
    __syncthreads();
    const uint32_t n_flush = P.n_slots_rev < n_hot ? P.n_slots_rev : n_hot;
    for (uint32_t i = threadIdx.x; i < n_flush; i += blockDim.x) {
        const double v = hot[i].x;
        if (v != 0.0) atomicAdd(&expected_slot[i], v);
    }
}

typedef void (*estep4l_fn)(Estep4Params);
static estep4l_fn pick_fwd(bool dropout, int ppl, bool long_tokens) {
    if (long_tokens) return dropout ? estep4l_fwd_kernel<true, 1, true> : estep4l_fwd_kernel<false, 1, true>;
    if (ppl == 4) return dropout ? estep4l_fwd_kernel<true, 4, false> : estep4l_fwd_kernel<false, 4, false>;
    if (ppl == 2) return dropout ? estep4l_fwd_kernel<true, 2, false> : estep4l_fwd_kernel<false, 2, false>;
    return dropout ? estep4l_fwd_kernel<true, 1, false> : estep4l_fwd_kernel<false, 1, false>;
}
static estep4l_fn pick_bwd(bool dropout, int ppl, bool long_tokens) {
    if (long_tokens) return dropout ? estep4l_bwd_kernel<true, 1, true> : estep4l_bwd_kernel<false, 1, true>;
    if (ppl == 4) return dropout ? estep4l_bwd_kernel<true, 4, false> : estep4l_bwd_kernel<false, 4, false>;
    if (ppl == 2) return dropout ? estep4l_bwd_kernel<true, 2, false> : estep4l_bwd_kernel<false, 2, false>;
    return dropout ? estep4l_bwd_kernel<true, 1, false> : estep4l_bwd_kernel<false, 1, false>;
}

hipError_t estep4l_prepare() {
    for (int d = 0; d < 2; d++)
        for (int ppl = 1; ppl <= 4; ppl *= 2)
            for (int lg = 0; lg < (ppl == 1 ? 2 : 1); lg++) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pick_fwd(d == 1, ppl, lg == 1)),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e != hipSuccess) return e;
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(pick_bwd(d == 1, ppl, lg == 1)),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e != hipSuccess) return e;
            }
    return hipSuccess;
}

// forward: 8 KiB * ppl of LDS per wave: 5 blocks x 4 waves / 2 x 5 / 1 x 5 per CU for ppl = 1 / 2 / 4
// (long_tokens: vocabularies with tokens of 17..32 bytes — one position per lane, an overflow list per wave)
hipError_t launch_estep4l_fwd(const Estep4Params& p, int ppl, bool long_tokens, uint32_t num_cus, hipStream_t stream) {
    if (long_tokens) ppl = 1;
    const uint32_t waves = ppl == 1 ? 4u : 5u, bpc = ppl == 1 ? (long_tokens ? 4u : 5u) : (ppl == 2 ? 2u : 1u);
    const uint64_t want = (p.n_snips + 4 * waves - 1) / (4 * waves);
    const uint32_t blocks = (uint32_t)(want < (uint64_t)num_cus * bpc ? (want ? want : 1) : (uint64_t)num_cus * bpc);
    const uint32_t lds = waves * (uint32_t)ppl * kE4LEntries * 8u + (long_tokens ? waves * kE4LOvfBytes : 0u);
    hipLaunchKernelGGL(pick_fwd(p.dropout > 0.0, ppl, long_tokens), dim3(blocks), dim3(64u * waves), lds, stream, p);
    return hipGetLastError();
}
// backward: ONE block per CU: 16 / 8 / 4 waves x 4 KiB * ppl of match buffers (slots), the rest of the 160 KiB hot entries
hipError_t launch_estep4l_bwd(const Estep4Params& p0, int ppl, bool long_tokens, uint32_t num_cus, uint32_t groups_wanted, hipStream_t stream) {
    if (long_tokens) ppl = 1;
    // groups of 16 positions per block: 16 (x 4 KiB of match buffer) leave 96 KiB = 6 143 hot entries of 16
    // bytes; fewer groups, more slots summed in LDS instead of by memory-side atomics but fewer waves to hide
    // the gathers: 44.3 / 40.9 / 39.5 ms per GiB with 12 / 14 / 16 groups, 51.0 ms with the 12 groups and
    // 2 048 hot slots that 12-byte match entries allowed (profiles/r02; groups_wanted: TGX_BWD_GROUPS, read by the caller)
    const uint32_t groups = (groups_wanted >= 4u && groups_wanted <= 16u) ? groups_wanted : 16u;
    Estep4Params p = p0;
    const uint32_t waves = std::max(1u, groups / (uint32_t)ppl);
    const uint32_t wave_bytes = (uint32_t)ppl * kE4LEntries * 4u + (long_tokens ? kE4LOvfBytes : 0u);
    p.n_hot = (160u * 1024u - waves * wave_bytes) / 16u - 1u;
    const uint64_t want = (p.n_snips + 4 * waves - 1) / (4 * waves);
    const uint32_t blocks = (uint32_t)(want < (uint64_t)num_cus ? (want ? want : 1) : (uint64_t)num_cus);
    const uint32_t lds = (p.n_hot + 1u) * 16u + waves * wave_bytes;
    hipLaunchKernelGGL(pick_bwd(p.dropout > 0.0, ppl, long_tokens), dim3(blocks), dim3(64u * waves), lds, stream, p);
    return hipGetLastError();
}

}  // namespace tgx
