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

// kHotSlots: expected counts of the first (hottest-first order) slots of the reversed trie are summed in
// the block's LDS and flushed once at the end; 2048 slots take about three quarters of all matches, and
// the global f64 atomics are what bounds the backward kernel.
constexpr uint32_t kHotSlots = 2048;

template <int U>
__device__ __forceinline__ void e4l_bwd_step(double sv, uint32_t hv, double c_cur, double c_nxt, int e_cur, int e_nxt,
                                             int eb, double* __restrict__ expected_slot, double* hot, bool cold_ok, double& acc) {
    constexpr uint64_t MU = kRowLane0 << U;
    constexpr uint64_t WRAPPED = (uint64_t)((1u << (U + 1)) - 1u) * kRowLane0;  // lanes l <= U: position y0 + 16 + l
    const double best = row_bcast_f64<U>(acc);  // b[q] of the source (end) position
    const double cand = best * sv;              // w * b[q]
    const double c = sel_f64(WRAPPED, c_nxt, c_cur);                     // a[p] / a[n] of this lane's start position
    const int e = (int)sel_u32(WRAPPED, (uint32_t)e_nxt, (uint32_t)e_cur);  // Ea(p) - Ea(n)
    if (sv != 0.0) {  // lattice.rs:305-307
        const double mg = ldexp(cand * c, e + eb);
        if (hv < kHotSlots)
            atomicAdd(&hot[hv], mg);  // ds_add_f64
        else if (cold_ok)
            atomicAdd(&expected_slot[hv], mg);
    }
    acc = sel_f64(MU, cand, acc + cand);
}

template <bool DROPOUT>
__global__ __launch_bounds__(256) void estep4l_fwd_kernel(Estep4Params P) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr uint32_t LM = 16;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t l = lane & 15u, r = lane >> 4;
    const uint32_t wave = threadIdx.x >> 6;
    const uint4* __restrict__ trie = reinterpret_cast<const uint4*>(P.trie_fwd);  // records carry w = exp(score)
    double* sc = reinterpret_cast<double*>(smem + (size_t)wave * (kE4LEntries * 8u));

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
        const uint32_t p = p0 + l;
        const uintptr_t addr = reinterpret_cast<uintptr_t>(P.text + (live ? beg + p : 0));
        const uint32_t sh = (uint32_t)(addr & 3u);
        const uint32_t* __restrict__ wp = reinterpret_cast<const uint32_t*>(addr & ~uintptr_t(3));
        uint32_t w[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) w[q] = wp[q];
        uint32_t bytes[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bytes[q] = __builtin_amdgcn_alignbyte(w[q + 1], w[q], sh);
        {
            double2* grp = reinterpret_cast<double2*>(sc);
#pragma unroll
            for (int q = 0; q < 8; ++q) grp[q * 64 + lane] = make_double2(0.0, 0.0);
        }
        const uint32_t rem = (live && p < n) ? (n - p) : 0u;
        const uint32_t maxd = rem < LM ? rem : LM;
        uint32_t cur = 0, base = P.root_fwd;
        bool alive = maxd > 0;
        double* scw = sc + lane * LM;
#pragma unroll
        for (int d = 0; d < (int)LM; ++d) {
            alive = alive && ((uint32_t)d < maxd);
            if (__builtin_amdgcn_ballot_w64(alive) == 0) break;
            if (alive) {
                const uint32_t c = (bytes[d >> 2] >> ((d & 3) * 8)) & 0xFFu;
                const uint32_t t = base ^ c;
                const uint4 rec = load_rec(trie, t);
                alive = rec.x == cur;
                if (alive) {
                    cur = t;
                    base = rec.y & 0x7FFFFFFFu;
                    bool term = (rec.y >> 31) != 0u;
                    if (DROPOUT) {  // model.rs:48: skipped iff len > 1 && rand < dropout
                        if (term && d >= 1) term = !(dropout_u01(P.seed, smp, sbase + p, (uint32_t)d + 1u) < P.dropout);
                    }
                    if (term) scw[((uint32_t)d + l) & 15u] = __hiloint2double((int)rec.w, (int)rec.z);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();

        // ---- forward recursion, 16 static steps
        double fin = 0.0;
        {
            const double* scr = sc + r * 256u + ((l - 1u) & 15u);
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
        }
        __builtin_amdgcn_wave_barrier();

        // ---- a[p0 .. p0+15] and the block's exponent -> scratch (snippet s: n + 1 values at soffs[s] + s)
        if (live && p <= n) {
            P.alpha[beg + s + p] = fin;
            // a position nothing was pushed to (lattice.rs:255: it counts as log-probability 0.0 there), an
            // underflow or an overflow: this pass belongs to the log-domain kernels
            if (!(fin > 0.0 && fin <= 1.7976931348623157e308)) atomicMax(P.range_flag, 1ULL);
        }
        if (live && l == 0u) P.alpha_exp[ebase + (p0 >> 4)] = erow;
        if (live) {
            const uint32_t left = n - p0;
            if (left < 16u) {  // position n lies in this block: z = log alpha_true[n] (lattice.rs:290-291)
                if (l == left) {
                    const double z = log(fin) + (double)erow * 0.6931471805599453;
                    P.zarr[s] = z;
                    zsum += z;
                    // !z.is_normal() panics in the reference (prune.rs:90-96)
                    const double az = fabs(z);
                    if (!(az >= 2.2250738585072014e-308 && az <= 1.7976931348623157e308))
                        atomicMin(P.err_snip, (unsigned long long)s);
                }
                need_new = true;
            } else {
                p0 += 16u;
            }
        }
        // ---- rescale the row: the largest accumulator goes to [0.5, 1)
        {
            const int e = row_max_exponent(acc);
            if (e > -100000) {
                acc = ldexp(acc, -e);
                erow += e;
            }
        }
    }
    if (zsum != 0.0) atomicAdd(P.logz_sum, zsum);
}

template <bool DROPOUT>
__global__ __launch_bounds__(768) void estep4l_bwd_kernel(Estep4Params P) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr uint32_t LM = 16;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t l = lane & 15u, r = lane >> 4;
    const uint32_t wave = threadIdx.x >> 6;
    const uint4* __restrict__ trie = reinterpret_cast<const uint4*>(P.trie_rev);  // records carry w = exp(score)
    double* hot = reinterpret_cast<double*>(smem);  // kHotSlots partial sums, shared by the block
    unsigned char* wbase = smem + kHotSlots * 8u + (size_t)wave * (kE4LEntries * 12u);
    double* sc = reinterpret_cast<double*>(wbase);
    uint32_t* hl = reinterpret_cast<uint32_t*>(wbase + kE4LEntries * 8u);
    // expected counts go to one of n_replicas copies of the slot array (reduced afterwards):
    // a handful of very frequent tokens would otherwise serialise every wave's atomics
    double* __restrict__ expected_slot = P.expected_slot + (size_t)(blockIdx.x % P.n_replicas) * P.n_slots_rev;
    const bool cold_ok = (P.flags & 8u) == 0u;
    for (uint32_t i = threadIdx.x; i < kHotSlots; i += blockDim.x) hot[i] = 0.0;
    __syncthreads();

    uint32_t s = 0, n = 0, y0 = 0, smp = 0;
    uint64_t beg = 0, sbase = 0, ebase = 0;
    bool live = false, need_new = true;
    double acc = 0.0, inv_an = 0.0, c_cur = 0.0, c_nxt = 0.0;
    int eb = 0, ean = 0, e_cur = 0, e_nxt = 0;

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
            load_fwd(l, c_cur, e_cur);
            load_fwd(16u + l, c_nxt, e_nxt);
        }
        need_new = false;
        if (__builtin_amdgcn_ballot_w64(live) == 0) break;

        // ---- match on the reversed text with the reversed-token trie: lane (r, l) owns end
        // position q = n - y, y = y0 + l, and reads text[q - 1], text[q - 2], ...
        const uint32_t y = y0 + l;
        const uint32_t q = (live && y < n) ? (n - y) : 0u;  // bytes available before q
        const uint32_t maxd = q < LM ? q : LM;
        // the 16 bytes text[q - 16 .. q) (the buffer has a 256-byte front pad)
        const uintptr_t addr = reinterpret_cast<uintptr_t>(P.text + (live ? beg + (uint64_t)(n - (y < n ? y : n)) : 16u)) - 16u;
        const uint32_t sh = (uint32_t)(addr & 3u);
        const uint32_t* __restrict__ wp = reinterpret_cast<const uint32_t*>(addr & ~uintptr_t(3));
        uint32_t w[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) w[j] = wp[j];
        uint32_t bytes[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bytes[j] = __builtin_amdgcn_alignbyte(w[j + 1], w[j], sh);
        {
            double2* grp = reinterpret_cast<double2*>(sc);
#pragma unroll
            for (int j = 0; j < 8; ++j) grp[j * 64 + lane] = make_double2(0.0, 0.0);
        }
        uint32_t cur = 0, base = P.root_rev;
        bool alive = maxd > 0;
        double* scw = sc + lane * LM;
        uint32_t* hlw = hl + lane * LM;
#pragma unroll
        for (int d = 0; d < (int)LM; ++d) {
            alive = alive && ((uint32_t)d < maxd);
            if (__builtin_amdgcn_ballot_w64(alive) == 0) break;
            if (alive) {
                const uint32_t c = (bytes[(15 - d) >> 2] >> (((15 - d) & 3) * 8)) & 0xFFu;  // text[q - 1 - d]
                const uint32_t t = base ^ c;
                const uint4 rec = load_rec(trie, t);
                alive = rec.x == cur;
                if (alive) {
                    cur = t;
                    base = rec.y & 0x7FFFFFFFu;
                    bool term = (rec.y >> 31) != 0u;
                    if (DROPOUT) {  // keyed by the token's START byte in the sample, as in the forward sweep
                        if (term && d >= 1)
                            term = !(dropout_u01(P.seed, smp, sbase + (uint64_t)(q - (uint32_t)d - 1u), (uint32_t)d + 1u) < P.dropout);
                    }
                    if (term) {
                        const uint32_t col = ((uint32_t)d + l) & 15u;
                        scw[col] = __hiloint2double((int)rec.w, (int)rec.z);
                        hlw[col] = t;
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();

        // ---- backward recursion + marginals, 16 static steps
        {
            const double* scr = sc + r * 256u + ((l - 1u) & 15u);
            const uint32_t* hlr = hl + r * 256u + ((l - 1u) & 15u);
            double sv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) sv[u] = scr[u * 16];
            e4l_bwd_step<0>(sv[0], hlr[0 * 16], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, cold_ok, acc);
            e4l_bwd_step<1>(sv[1], hlr[1 * 16], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, cold_ok, acc);
            e4l_bwd_step<2>(sv[2], hlr[2 * 16], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, cold_ok, acc);
            e4l_bwd_step<3>(sv[3], hlr[3 * 16], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, cold_ok, acc);
            e4l_bwd_step<4>(sv[4], hlr[4 * 16], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, cold_ok, acc);
            e4l_bwd_step<5>(sv[5], hlr[5 * 16], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, cold_ok, acc);
            e4l_bwd_step<6>(sv[6], hlr[6 * 16], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, cold_ok, acc);
            e4l_bwd_step<7>(sv[7], hlr[7 * 16], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, cold_ok, acc);
            e4l_bwd_step<8>(sv[8], hlr[8 * 16], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, cold_ok, acc);
            e4l_bwd_step<9>(sv[9], hlr[9 * 16], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, cold_ok, acc);
            e4l_bwd_step<10>(sv[10], hlr[10 * 16], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, cold_ok, acc);
            e4l_bwd_step<11>(sv[11], hlr[11 * 16], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, cold_ok, acc);
            e4l_bwd_step<12>(sv[12], hlr[12 * 16], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, cold_ok, acc);
            e4l_bwd_step<13>(sv[13], hlr[13 * 16], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, cold_ok, acc);
            e4l_bwd_step<14>(sv[14], hlr[14 * 16], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, cold_ok, acc);
            e4l_bwd_step<15>(sv[15], hlr[15 * 16], c_cur, c_nxt, e_cur, e_nxt, eb, expected_slot, hot, cold_ok, acc);
        }
        __builtin_amdgcn_wave_barrier();

        if (live) {
            const uint32_t left = n - y0;
            if (left < 16u) {  // position 0 lies in this block: the snippet is done
                need_new = true;
            } else {
                y0 += 16u;
                c_cur = c_nxt;
                e_cur = e_nxt;
                load_fwd(y0 + 16u + l, c_nxt, e_nxt);
            }
        }
        {
            const int e = row_max_exponent(acc);
            if (e > -100000) {
                acc = ldexp(acc, -e);
                eb += e;
            }
        }
    }
    __syncthreads();
    const uint32_t n_hot = P.n_slots_rev < kHotSlots ? P.n_slots_rev : kHotSlots;
    for (uint32_t i = threadIdx.x; i < n_hot; i += blockDim.x) {
        const double v = hot[i];
        if (v != 0.0) atomicAdd(&expected_slot[i], v);
    }
}

hipError_t estep4l_prepare() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(estep4l_fwd_kernel<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(estep4l_fwd_kernel<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(estep4l_bwd_kernel<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(estep4l_bwd_kernel<true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

// forward: 8 KiB of LDS per wave (5 blocks x 4 waves per CU); backward: 12 KiB (3 x 4)
hipError_t launch_estep4l_fwd(const Estep4Params& p, uint32_t num_cus, hipStream_t stream) {
    const uint64_t want = (p.n_snips + 15) / 16;
    const uint32_t blocks = (uint32_t)(want < (uint64_t)num_cus * 5 ? (want ? want : 1) : (uint64_t)num_cus * 5);
    if (p.dropout > 0.0)
        hipLaunchKernelGGL(estep4l_fwd_kernel<true>, dim3(blocks), dim3(256), 4u * kE4LEntries * 8u, stream, p);
    else
        hipLaunchKernelGGL(estep4l_fwd_kernel<false>, dim3(blocks), dim3(256), 4u * kE4LEntries * 8u, stream, p);
    return hipGetLastError();
}
// backward: ONE block of 12 waves per CU: 12 x 12 KiB of match buffers + 16 KiB of hot-slot sums = 160 KiB
hipError_t launch_estep4l_bwd(const Estep4Params& p, uint32_t num_cus, hipStream_t stream) {
    const uint32_t waves = 12;
    const uint64_t want = (p.n_snips + 4 * waves - 1) / (4 * waves);
    const uint32_t blocks = (uint32_t)(want < (uint64_t)num_cus ? (want ? want : 1) : (uint64_t)num_cus);
    const uint32_t lds = kHotSlots * 8u + waves * kE4LEntries * 12u;
    if (p.dropout > 0.0)
        hipLaunchKernelGGL(estep4l_bwd_kernel<true>, dim3(blocks), dim3(64u * waves), lds, stream, p);
    else
        hipLaunchKernelGGL(estep4l_bwd_kernel<false>, dim3(blocks), dim3(64u * waves), lds, stream, p);
    return hipGetLastError();
}

}  // namespace tgx
