// E-step, four snippets per wave (max token length <= 16, finite scores).
//
// Same arithmetic as estep.hip (which stays as the generic path and documents the
// mapping to the reference: src/model.rs:34-55, src/lattice.rs:245-333,
// src/prune.rs:64-120), on the machinery of encode4_kernel: a wave is four 16-lane rows,
// each with its own snippet, advancing in lock-step over blocks of 16 positions; 64 trie
// walks per block; matches as f64 scores in a swizzled LDS buffer with -inf for "no
// token"; DPP row broadcasts; 16 fully unrolled steps.  The work list is the list of
// SNIPPETS (every sample cut at multiples of snippet_len, src/prune.rs:83), so a snippet
// is to these kernels what a sample is to encode4_kernel.
//
//   estep4_fwd_kernel : A[p] (alpha) by pushes in ascending start order with the
//                       reference's log_sum_exp (lattice.rs:259-272, 321-333); A[] of the
//                       snippet goes to the alpha scratch row, z = A[n] to zarr.
//   estep4_bwd_kernel : B[q] (beta) as the same recursion on the reversed text with the
//                       trie of the reversed tokens; each push also adds the token's
//                       marginal exp(((A[p] + s) + B[q]) - z) (lattice.rs:305-307) to the
//                       expected count of its (reversed-trie) slot.
//
// A position nothing was pushed to contributes 0.0, as the reference's zero-initialised
// vectors do (lattice.rs:255-256).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_common.h"
#include "kernels.h"

namespace tgx {

constexpr uint32_t kE4Entries = 1024;  // 64 rows x 16 columns

// log_sum_exp(x, y) — reference src/lattice.rs:321-333; x may be -inf ("nothing yet": the
// reference's init_mode assigns y, and so does this: vmax > vmin + 50)
__device__ __forceinline__ double lse_pair(double x, double y) {
    double vmin, vmax;
    if (x > y) {
        vmin = y;
        vmax = x;
    } else {
        vmin = x;
        vmax = y;
    }
    if (vmax > vmin + 50.0) return vmax;
    return vmax + log(exp(vmin - vmax) + 1.0);
}

template <int U>
__device__ __forceinline__ void e4_fwd_step(double sv, double& acc, double& fin) {
    constexpr uint64_t MU = kRowLane0 << U;  // lanes with l == U
    const double ninf = -__builtin_huge_val();
    const double mine = (acc == ninf) ? 0.0 : acc;  // lattice.rs:255: nothing pushed -> 0.0
    fin = sel_f64(MU, mine, fin);                   // A[p0 + U] is final now
    const double best = row_bcast_f64<U>(mine);
    acc = sel_f64(MU, ninf, acc);                   // lane U now accumulates position p0 + U + 16
    const double cand = sv + best;                  // lattice.rs:267: score + alpha
    const double merged = lse_pair(acc, cand);
    acc = (sv == ninf) ? acc : merged;              // no token of this length here
}

template <int U>
__device__ __forceinline__ void e4_bwd_step(double sv, uint32_t hv, double a_cur, double a_nxt, double z,
                                            double* __restrict__ expected_slot, double& acc) {
    constexpr uint64_t MU = kRowLane0 << U;
    constexpr uint64_t WRAPPED = (uint64_t)((1u << (U + 1)) - 1u) * kRowLane0;  // lanes l <= U: position y0 + 16 + l
    const double ninf = -__builtin_huge_val();
    const double mine = (acc == ninf) ? 0.0 : acc;
    const double best = row_bcast_f64<U>(mine);     // B[q] of the source position
    acc = sel_f64(MU, ninf, acc);
    const double a = sel_f64(WRAPPED, a_nxt, a_cur);  // A[p] of this lane's (start) position
    const double cand = sv + best;                  // lattice.rs:282: score + beta
    if (sv != ninf) {                               // lattice.rs:305-307
        const double total = ((a + sv) + best) - z;
        atomicAdd(&expected_slot[hv], exp(total));
    }
    const double merged = lse_pair(acc, cand);
    acc = (sv == ninf) ? acc : merged;
}

template <bool DROPOUT>
__global__ __launch_bounds__(256) void estep4_fwd_kernel(Estep4Params P) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr uint32_t LM = 16;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t l = lane & 15u, r = lane >> 4;
    const uint32_t wave = threadIdx.x >> 6;
    const uint4* __restrict__ trie = reinterpret_cast<const uint4*>(P.trie_fwd);
    double* sc = reinterpret_cast<double*>(smem + (size_t)wave * (kE4Entries * 8u));

    const double ninf = -__builtin_huge_val();
    uint32_t s = 0, n = 0, p0 = 0, smp = 0;
    uint64_t beg = 0, sbase = 0;
    bool live = false, need_new = true;
    double acc = ninf;
    double zsum = 0.0;

    for (;;) {
        const uint64_t k = claim_rows(P.queue_fwd, need_new, r);  // longest-first, dynamic
        if (need_new) {
            live = k < P.n_snips;
            if (live) {
                s = P.order[k];
                beg = P.soffs[s];
                n = (uint32_t)(P.soffs[s + 1] - beg);
                if (DROPOUT) {
                    smp = P.snip_sample[s];
                    sbase = P.snip_base[s];
                }
            }
            p0 = 0;
            acc = (l == 0u) ? 0.0 : ninf;  // BOS: alpha = 0 (lattice.rs:96-101, 267)
        }
        need_new = false;
        if (__builtin_amdgcn_ballot_w64(live) == 0) break;

        // ---- match (forward trie), as in encode4_kernel
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
            for (int q = 0; q < 8; ++q) grp[q * 64 + lane] = make_double2(ninf, ninf);
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
            e4_fwd_step<0>(sv[0], acc, fin);
            e4_fwd_step<1>(sv[1], acc, fin);
            e4_fwd_step<2>(sv[2], acc, fin);
            e4_fwd_step<3>(sv[3], acc, fin);
            e4_fwd_step<4>(sv[4], acc, fin);
            e4_fwd_step<5>(sv[5], acc, fin);
            e4_fwd_step<6>(sv[6], acc, fin);
            e4_fwd_step<7>(sv[7], acc, fin);
            e4_fwd_step<8>(sv[8], acc, fin);
            e4_fwd_step<9>(sv[9], acc, fin);
            e4_fwd_step<10>(sv[10], acc, fin);
            e4_fwd_step<11>(sv[11], acc, fin);
            e4_fwd_step<12>(sv[12], acc, fin);
            e4_fwd_step<13>(sv[13], acc, fin);
            e4_fwd_step<14>(sv[14], acc, fin);
            e4_fwd_step<15>(sv[15], acc, fin);
        }
        __builtin_amdgcn_wave_barrier();

        // ---- A[p0 .. p0+15] -> scratch (snippet s at soffs[s] + s: n + 1 values), next block
        if (live && p <= n) P.alpha[beg + s + p] = fin;
        if (live) {
            const uint32_t left = n - p0;
            if (left < 16u) {  // position n lies in this block: z = A[n] (lattice.rs:290-291)
                if (l == left) {
                    P.zarr[s] = fin;
                    zsum += fin;
                    // !z.is_normal() panics in the reference (prune.rs:90-96)
                    const double az = fabs(fin);
                    if (!(az >= 2.2250738585072014e-308 && az <= 1.7976931348623157e308))
                        atomicMin(P.err_snip, (unsigned long long)s);
                }
                need_new = true;
            } else {
                p0 += 16u;
            }
        }
    }
    if (zsum != 0.0) atomicAdd(P.logz_sum, zsum);
}

template <bool DROPOUT>
__global__ __launch_bounds__(256) void estep4_bwd_kernel(Estep4Params P) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr uint32_t LM = 16;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t l = lane & 15u, r = lane >> 4;
    const uint32_t wave = threadIdx.x >> 6;
    const uint4* __restrict__ trie = reinterpret_cast<const uint4*>(P.trie_rev);
    unsigned char* wbase = smem + (size_t)wave * (kE4Entries * 12u);
    double* sc = reinterpret_cast<double*>(wbase);
    uint32_t* hl = reinterpret_cast<uint32_t*>(wbase + kE4Entries * 8u);
    // expected counts go to one of n_replicas copies of the slot array (reduced afterwards):
    // a handful of very frequent tokens would otherwise serialise every wave's atomics
    double* __restrict__ expected_slot = P.expected_slot + (size_t)(blockIdx.x % P.n_replicas) * P.n_slots_rev;

    const double ninf = -__builtin_huge_val();
    uint32_t s = 0, n = 0, y0 = 0, smp = 0;
    uint64_t beg = 0, sbase = 0;
    bool live = false, need_new = true;
    double acc = ninf, z = 0.0, a_cur = 0.0, a_nxt = 0.0;

    for (;;) {
        const uint64_t k = claim_rows(P.queue_bwd, need_new, r);  // longest-first, dynamic
        if (need_new) {
            live = k < P.n_snips;
            if (live) {
                s = P.order[k];
                beg = P.soffs[s];
                n = (uint32_t)(P.soffs[s + 1] - beg);
                z = P.zarr[s];
                if (DROPOUT) {
                    smp = P.snip_sample[s];
                    sbase = P.snip_base[s];
                }
            }
            y0 = 0;
            acc = (l == 0u) ? 0.0 : ninf;  // EOS: beta = 0
            // forward values of the positions this lane accumulates: y = l and y = 16 + l  (p = n - y)
            a_cur = (live && l <= n) ? P.alpha[beg + s + (n - l)] : 0.0;
            a_nxt = (live && 16u + l <= n) ? P.alpha[beg + s + (n - 16u - l)] : 0.0;
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
            for (int j = 0; j < 8; ++j) grp[j * 64 + lane] = make_double2(ninf, ninf);
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
            e4_bwd_step<0>(sv[0], hlr[0 * 16], a_cur, a_nxt, z, expected_slot, acc);
            e4_bwd_step<1>(sv[1], hlr[1 * 16], a_cur, a_nxt, z, expected_slot, acc);
            e4_bwd_step<2>(sv[2], hlr[2 * 16], a_cur, a_nxt, z, expected_slot, acc);
            e4_bwd_step<3>(sv[3], hlr[3 * 16], a_cur, a_nxt, z, expected_slot, acc);
            e4_bwd_step<4>(sv[4], hlr[4 * 16], a_cur, a_nxt, z, expected_slot, acc);
            e4_bwd_step<5>(sv[5], hlr[5 * 16], a_cur, a_nxt, z, expected_slot, acc);
            e4_bwd_step<6>(sv[6], hlr[6 * 16], a_cur, a_nxt, z, expected_slot, acc);
            e4_bwd_step<7>(sv[7], hlr[7 * 16], a_cur, a_nxt, z, expected_slot, acc);
            e4_bwd_step<8>(sv[8], hlr[8 * 16], a_cur, a_nxt, z, expected_slot, acc);
            e4_bwd_step<9>(sv[9], hlr[9 * 16], a_cur, a_nxt, z, expected_slot, acc);
            e4_bwd_step<10>(sv[10], hlr[10 * 16], a_cur, a_nxt, z, expected_slot, acc);
            e4_bwd_step<11>(sv[11], hlr[11 * 16], a_cur, a_nxt, z, expected_slot, acc);
            e4_bwd_step<12>(sv[12], hlr[12 * 16], a_cur, a_nxt, z, expected_slot, acc);
            e4_bwd_step<13>(sv[13], hlr[13 * 16], a_cur, a_nxt, z, expected_slot, acc);
            e4_bwd_step<14>(sv[14], hlr[14 * 16], a_cur, a_nxt, z, expected_slot, acc);
            e4_bwd_step<15>(sv[15], hlr[15 * 16], a_cur, a_nxt, z, expected_slot, acc);
        }
        __builtin_amdgcn_wave_barrier();

        if (live) {
            const uint32_t left = n - y0;
            if (left < 16u) {  // position 0 lies in this block: the snippet is done
                need_new = true;
            } else {
                y0 += 16u;
                a_cur = a_nxt;
                a_nxt = (y0 + 16u + l <= n) ? P.alpha[beg + s + (n - y0 - 16u - l)] : 0.0;
            }
        }
    }
}

// expected[slot] = sum over replicas
__global__ __launch_bounds__(256) void estep4_reduce_kernel(const double* __restrict__ rep, double* __restrict__ out,
                                                           uint32_t n_slots, uint32_t n_replicas) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_slots) return;
    double acc = 0.0;
    for (uint32_t k = 0; k < n_replicas; ++k) acc += rep[(size_t)k * n_slots + i];  // fixed order
    out[i] = acc;
}
hipError_t launch_estep4_reduce(const double* rep, double* out, uint32_t n_slots, uint32_t n_replicas,
                                hipStream_t stream) {
    hipLaunchKernelGGL(estep4_reduce_kernel, dim3((n_slots + 255u) / 256u), dim3(256), 0, stream, rep, out, n_slots,
                       n_replicas);
    return hipGetLastError();
}

hipError_t estep4_prepare() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(estep4_fwd_kernel<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(estep4_fwd_kernel<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(estep4_bwd_kernel<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(estep4_bwd_kernel<true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

// forward: 8 KiB of LDS per wave (5 blocks x 4 waves per CU); backward: 12 KiB (3 x 4)
hipError_t launch_estep4_fwd(const Estep4Params& p, uint32_t num_cus, hipStream_t stream) {
    const uint64_t want = (p.n_snips + 15) / 16;
    const uint32_t blocks = (uint32_t)(want < (uint64_t)num_cus * 5 ? (want ? want : 1) : (uint64_t)num_cus * 5);
    if (p.dropout > 0.0)
        hipLaunchKernelGGL(estep4_fwd_kernel<true>, dim3(blocks), dim3(256), 4u * kE4Entries * 8u, stream, p);
    else
        hipLaunchKernelGGL(estep4_fwd_kernel<false>, dim3(blocks), dim3(256), 4u * kE4Entries * 8u, stream, p);
    return hipGetLastError();
}
hipError_t launch_estep4_bwd(const Estep4Params& p, uint32_t num_cus, hipStream_t stream) {
    const uint64_t want = (p.n_snips + 15) / 16;
    const uint32_t blocks = (uint32_t)(want < (uint64_t)num_cus * 3 ? (want ? want : 1) : (uint64_t)num_cus * 3);
    if (p.dropout > 0.0)
        hipLaunchKernelGGL(estep4_bwd_kernel<true>, dim3(blocks), dim3(256), 4u * kE4Entries * 12u, stream, p);
    else
        hipLaunchKernelGGL(estep4_bwd_kernel<false>, dim3(blocks), dim3(256), 4u * kE4Entries * 12u, stream, p);
    return hipGetLastError();
}

}  // namespace tgx
