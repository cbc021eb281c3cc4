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

// exponent of the largest of a row's 16 accumulators (all lanes of the row get it); zeros do not count
__device__ __forceinline__ int row_max_exponent(double acc) {
    int e = (acc == 0.0) ? -100000 : __builtin_amdgcn_frexp_exp(acc);  // acc = m * 2^e, 0.5 <= |m| < 1
    e = max(e, __builtin_amdgcn_update_dpp(e, e, 0x121, 0xF, 0xF, false));  // row_ror:1
    e = max(e, __builtin_amdgcn_update_dpp(e, e, 0x122, 0xF, 0xF, false));  // row_ror:2
    e = max(e, __builtin_amdgcn_update_dpp(e, e, 0x124, 0xF, 0xF, false));  // row_ror:4
    e = max(e, __builtin_amdgcn_update_dpp(e, e, 0x128, 0xF, 0xF, false));  // row_ror:8
    return e;
}

template <int U>
__device__ __forceinline__ void e4l_fwd_step(double sv, double& acc, double& fin) {
    constexpr uint64_t MU = kRowLane0 << U;  // lanes with l == U
    fin = sel_f64(MU, acc, fin);             // a[p0 + U] is final now
    const double best = row_bcast_f64<U>(acc);
    const double cand = best * sv;           // sv = 0 where no token of this length starts at p0 + U
    acc = sel_f64(MU, cand, acc + cand);     // lane U starts accumulating position p0 + U + 16
}

// Hot slots (P.n_hot): expected counts of the first (hottest-first order) slots of the reversed trie are summed in
// the block's LDS and flushed once at the end; the global f64 atomics of the others (executed at the memory
// side, not in L2: ~30 G scattered adds per second chip-wide whatever their locality — 888 M of them per GiB
// with 2048 hot slots, TCC_EA0_ATOMIC in profiles/r02/t_passes_1GiB) are what bounds the backward kernel.
// So the LDS goes to the hot set: a hot entry is {sum, w = exp(score)} and the match buffer holds the SLOT of a
// match only (4 bytes per (position, length) instead of 12: the weight is read from the hot entry, or from the
// trie record in HBM for a cold slot), which leaves room for 7 000 hot slots instead of 2 000.
constexpr uint32_t kNoSlot = 0xFFFFFFFFu;

template <int U>
__device__ __forceinline__ void e4l_bwd_step(double sv, uint32_t hv, double c_cur, double c_nxt, int e_cur, int e_nxt,
                                             int eb, double* __restrict__ expected_slot, double2* hot, uint32_t n_hot, bool cold_ok, double& acc) {
    constexpr uint64_t MU = kRowLane0 << U;
    constexpr uint64_t WRAPPED = (uint64_t)((1u << (U + 1)) - 1u) * kRowLane0;  // lanes l <= U: position y0 + 16 + l
    const double best = row_bcast_f64<U>(acc);  // b[q] of the source (end) position
    const double cand = best * sv;              // w * b[q]
    const double c = sel_f64(WRAPPED, c_nxt, c_cur);                     // a[p] / a[n] of this lane's start position
    const int e = (int)sel_u32(WRAPPED, (uint32_t)e_nxt, (uint32_t)e_cur);  // Ea(p) - Ea(n)
    if (sv != 0.0) {  // lattice.rs:305-307
        const double mg = ldexp(cand * c, e + eb);
        if (hv < n_hot)
            atomicAdd(&hot[hv].x, mg);  // ds_add_f64
        else if (cold_ok)
            atomicAdd(&expected_slot[hv], mg);
    }
    acc = sel_f64(MU, cand, acc + cand);
}

// PPL = positions per lane and trip (1, 2, 4): a row advances 16 * PPL positions per trip of the dependent
// gather chain.  The relaxation is cheap here (one multiply-add per step), so more positions per lane shorten
// the serial chain of a long snippet almost proportionally, at the price of LDS (8 KiB * PPL per wave
// forward, 12 KiB * PPL backward) and with it waves per CU; the host picks PPL from the shape of the pass.
template <bool DROPOUT, int PPL>
__global__ __launch_bounds__(1024) void estep4l_fwd_kernel(Estep4Params P) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr uint32_t LM = 16;
    constexpr uint32_t SPAN = 16u * PPL;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t l = lane & 15u, r = lane >> 4;
    const uint32_t wave = threadIdx.x >> 6;
    const uint4* __restrict__ trie = reinterpret_cast<const uint4*>(P.trie_fwd);  // records carry w = exp(score)
    double* sc = reinterpret_cast<double*>(smem + (size_t)wave * (PPL * kE4LEntries * 8u));

    uint32_t s = 0, n = 0, p0 = 0, smp = 0;
    uint64_t beg = 0, sbase = 0, ebase = 0;
    bool live = false, need_new = true;
    double acc = 0.0;
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
            maxd[g] = rem < LM ? rem : LM;
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

template <bool DROPOUT, int PPL>
__global__ __launch_bounds__(1024) void estep4l_bwd_kernel(Estep4Params P) {
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
    // expected counts go to one of n_replicas copies of the slot array (reduced afterwards):
    // a handful of very frequent tokens would otherwise serialise every wave's atomics
    double* __restrict__ expected_slot = P.expected_slot + (size_t)(blockIdx.x % P.n_replicas) * P.n_slots_rev;
    const bool cold_ok = (P.flags & 8u) == 0u;
    for (uint32_t i = threadIdx.x; i <= n_hot; i += blockDim.x)
        hot[i] = make_double2(0.0, (i < n_hot && i < P.n_slots_rev) ? reinterpret_cast<const double*>(P.trie_rev)[2u * i + 1u] : 0.0);
    __syncthreads();

    uint32_t s = 0, n = 0, y0 = 0, smp = 0;
    uint64_t beg = 0, sbase = 0, ebase = 0;
    bool live = false, need_new = true;
    double acc = 0.0, inv_an = 0.0;
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
            maxd[g] = qq[g] < LM ? qq[g] : LM;
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
            e4l_bwd_step<0>(sv[0], hvs[0], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, n_hot, cold_ok, acc);
            e4l_bwd_step<1>(sv[1], hvs[1], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, n_hot, cold_ok, acc);
            e4l_bwd_step<2>(sv[2], hvs[2], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, n_hot, cold_ok, acc);
            e4l_bwd_step<3>(sv[3], hvs[3], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, n_hot, cold_ok, acc);
            e4l_bwd_step<4>(sv[4], hvs[4], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, n_hot, cold_ok, acc);
            e4l_bwd_step<5>(sv[5], hvs[5], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, n_hot, cold_ok, acc);
            e4l_bwd_step<6>(sv[6], hvs[6], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, n_hot, cold_ok, acc);
            e4l_bwd_step<7>(sv[7], hvs[7], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, n_hot, cold_ok, acc);
            e4l_bwd_step<8>(sv[8], hvs[8], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, n_hot, cold_ok, acc);
            e4l_bwd_step<9>(sv[9], hvs[9], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, n_hot, cold_ok, acc);
            e4l_bwd_step<10>(sv[10], hvs[10], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, n_hot, cold_ok, acc);
            e4l_bwd_step<11>(sv[11], hvs[11], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, n_hot, cold_ok, acc);
            e4l_bwd_step<12>(sv[12], hvs[12], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, n_hot, cold_ok, acc);
            e4l_bwd_step<13>(sv[13], hvs[13], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, n_hot, cold_ok, acc);
            e4l_bwd_step<14>(sv[14], hvs[14], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, n_hot, cold_ok, acc);
            e4l_bwd_step<15>(sv[15], hvs[15], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, n_hot, cold_ok, acc);
            const int e = row_max_exponent(acc);
            if (e > -100000) {
                acc = ldexp(acc, -e);
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
static estep4l_fn pick_fwd(bool dropout, int ppl) {
    if (ppl == 4) return dropout ? estep4l_fwd_kernel<true, 4> : estep4l_fwd_kernel<false, 4>;
    if (ppl == 2) return dropout ? estep4l_fwd_kernel<true, 2> : estep4l_fwd_kernel<false, 2>;
    return dropout ? estep4l_fwd_kernel<true, 1> : estep4l_fwd_kernel<false, 1>;
}
static estep4l_fn pick_bwd(bool dropout, int ppl) {
    if (ppl == 4) return dropout ? estep4l_bwd_kernel<true, 4> : estep4l_bwd_kernel<false, 4>;
    if (ppl == 2) return dropout ? estep4l_bwd_kernel<true, 2> : estep4l_bwd_kernel<false, 2>;
    return dropout ? estep4l_bwd_kernel<true, 1> : estep4l_bwd_kernel<false, 1>;
}

hipError_t estep4l_prepare() {
    for (int d = 0; d < 2; d++)
        for (int ppl = 1; ppl <= 4; ppl *= 2) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pick_fwd(d == 1, ppl)),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(pick_bwd(d == 1, ppl)),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
        }
    return hipSuccess;
}

// forward: 8 KiB * ppl of LDS per wave: 5 blocks x 4 waves / 2 x 5 / 1 x 5 per CU for ppl = 1 / 2 / 4
hipError_t launch_estep4l_fwd(const Estep4Params& p, int ppl, uint32_t num_cus, hipStream_t stream) {
    const uint32_t waves = ppl == 1 ? 4u : 5u, bpc = ppl == 1 ? 5u : (ppl == 2 ? 2u : 1u);
    const uint64_t want = (p.n_snips + 4 * waves - 1) / (4 * waves);
    const uint32_t blocks = (uint32_t)(want < (uint64_t)num_cus * bpc ? (want ? want : 1) : (uint64_t)num_cus * bpc);
    hipLaunchKernelGGL(pick_fwd(p.dropout > 0.0, ppl), dim3(blocks), dim3(64u * waves), waves * (uint32_t)ppl * kE4LEntries * 8u,
                       stream, p);
    return hipGetLastError();
}
// backward: ONE block per CU: 16 / 8 / 4 waves x 4 KiB * ppl of match buffers (slots), the rest of the 160 KiB hot entries
hipError_t launch_estep4l_bwd(const Estep4Params& p0, int ppl, uint32_t num_cus, hipStream_t stream) {
    // groups of 16 positions per block: 16 (x 4 KiB of match buffer) leave 96 KiB = 6 143 hot entries of 16
    // bytes; fewer groups, more slots summed in LDS instead of by memory-side atomics but fewer waves to hide
    // the gathers: 44.3 / 40.9 / 39.5 ms per GiB with 12 / 14 / 16 groups, 51.0 ms with the 12 groups and
    // 2 048 hot slots that 12-byte match entries allowed (profiles/r02; TGX_BWD_GROUPS overrides)
    uint32_t groups = 16;
    if (const char* e = getenv("TGX_BWD_GROUPS")) {
        const int v = atoi(e);
        if (v >= 4 && v <= 16) groups = (uint32_t)v;
    }
    Estep4Params p = p0;
    const uint32_t waves = std::max(1u, groups / (uint32_t)ppl);
    p.n_hot = (160u * 1024u - waves * (uint32_t)ppl * kE4LEntries * 4u) / 16u - 1u;
    const uint64_t want = (p.n_snips + 4 * waves - 1) / (4 * waves);
    const uint32_t blocks = (uint32_t)(want < (uint64_t)num_cus ? (want ? want : 1) : (uint64_t)num_cus);
    const uint32_t lds = (p.n_hot + 1u) * 16u + waves * (uint32_t)ppl * kE4LEntries * 4u;
    hipLaunchKernelGGL(pick_bwd(p.dropout > 0.0, ppl), dim3(blocks), dim3(64u * waves), lds, stream, p);
    return hipGetLastError();
}

}  // namespace tgx
