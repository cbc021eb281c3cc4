// Viterbi encode for vocabularies whose longest token has 17..32 bytes (vocabularies after `merge`:
// 20-24 bytes in the reference's recipes): TWO samples per wave on 32-lane rows.
//
// Same algorithm and the same bits as encode4_kernel (kernels.hip: four samples per wave on 16-lane DPP
// rows, which needs max token <= 16): a row advances over blocks of 32 positions, lane l of a row owns
// position p0 + l in the match phase and accumulates the end position e == l (mod 32) in the relaxation,
// "no token" is a score of -inf in the LDS match buffer (16 KiB per wave: 64 positions x 32 lengths x 8 B,
// entry (row, len - 1) at column (len - 1 + row) & 31 so that the reader's column is a per-lane constant),
// the lane whose position is finalised is restarted by forcing it to take that step's candidate, and only
// the length of the winning token is remembered (1 byte per position).  What differs is the broadcast of
// best[p0 + U] inside a row: DPP row_newbcast stops at 16 lanes, so it is two v_readlane pairs (one per
// row) and a select.  trace32_kernel is trace_kernel for tokens of up to 32 bytes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_common.h"
#include "kernels.h"
#include "trace_body.h"

namespace tgx {

constexpr uint32_t kNoStep2 = 0xFFu;
constexpr uint32_t kRows2Bytes = 64u * 32u * 8u;  // 16 KiB per wave
constexpr uint64_t kHiHalf = 0xFFFFFFFF00000000ULL;  // lanes of row 1

template <int U>
__device__ __forceinline__ void relax2_step(double sv, double& acc, uint32_t& bpv, uint32_t& fin, uint32_t& fhi) {
    constexpr uint64_t MU = (1ULL << U) | (1ULL << (32 + U));  // lanes with l == U
    fin = sel_u32(MU, bpv, fin);                               // winner step of position p0 + U is final now
    fhi = sel_u32(MU, (uint32_t)((uint64_t)__double_as_longlong(acc) >> 32), fhi);
    const double b0 = readlane_f64(acc, (uint32_t)U), b1 = readlane_f64(acc, 32u + (uint32_t)U);
    const double best = sel_f64(kHiHalf, b1, b0);
    const double cand = best + sv;  // model.rs:98
    const uint64_t take = __builtin_amdgcn_fcmp(cand, acc, 2 /* OGT: model.rs:101 */) | MU;
    acc = sel_f64(take, cand, acc);
    bpv = sel_imm_u32<U>(take, bpv);
}

// PERM: back-pointer bytes in encode4_kernel's permuted layout (the samples encode4l_kernel left to this kernel)
template <bool DROPOUT, bool PERM>
__global__ __launch_bounds__(640) void encode2_kernel(EncodeParams P) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr uint32_t LM = 32;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t l = lane & 31u, r = lane >> 5;
    const uint32_t wave = threadIdx.x >> 6;
    const uint4* __restrict__ trie = reinterpret_cast<const uint4*>(P.trie);
    double* sc = reinterpret_cast<double*>(smem + (size_t)wave * kRows2Bytes);

    uint32_t s = 0, n = 0, p0 = 0;
    uint64_t beg = 0;
    bool live = false, need_new = true;
    const double ninf = -__builtin_huge_val();
    double acc = ninf;
    uint32_t bpv = kNoStep2;
    uint32_t wn[9];  // prefetched text window of the next block
#pragma unroll
    for (int q = 0; q < 9; ++q) wn[q] = 0;

    for (;;) {
        // ---- rows that finished their sample claim the next one of the longest-first order
        {
            const uint64_t want = __builtin_amdgcn_ballot_w64(need_new) & 0x0000000100000001ULL;
            uint64_t k = ~0ull;
            if (want != 0) {  // wave-uniform
                const uint64_t base = wave_fetch_add(P.queue, (uint32_t)__builtin_popcountll(want));
                if (need_new) k = base + (uint64_t)__builtin_popcountll(want & ((1ull << (r * 32u)) - 1ull));
            }
            if (need_new) {
                live = k < P.n_samples;
                if (live) {
                    s = P.order[k];
                    beg = P.offs[s];
                    n = (uint32_t)(P.offs[s + 1] - beg);
                }
                p0 = 0;
                acc = (l == 0u) ? 0.0 : ninf;  // position 0: score 0, reachable
                bpv = kNoStep2;
            }
        }
        const bool fresh_row = need_new;
        need_new = false;
        if (__builtin_amdgcn_ballot_w64(live) == 0) break;

        // ---- match: 64 trie walks of up to 32 steps; lane (r, l) owns position p0 + l
        const uintptr_t addr = reinterpret_cast<uintptr_t>(P.text + (live ? beg + p0 + l : 0));
        const uint32_t sh = (uint32_t)(addr & 3u);
        const uint32_t* __restrict__ wp = reinterpret_cast<const uint32_t*>(addr & ~uintptr_t(3));
        uint32_t w[9];
        if (fresh_row) {
#pragma unroll
            for (int q = 0; q < 9; ++q) w[q] = wp[q];
        } else {
#pragma unroll
            for (int q = 0; q < 9; ++q) w[q] = wn[q];
        }
        uint32_t bytes[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) bytes[q] = __builtin_amdgcn_alignbyte(w[q + 1], w[q], sh);
        {
            double2* grp = reinterpret_cast<double2*>(sc);  // the wave sweeps its 16 KiB linearly
#pragma unroll
            for (int q = 0; q < 16; ++q) grp[q * 64 + lane] = make_double2(ninf, ninf);
        }
        const uint32_t pg = p0 + l;
        const uint32_t rem = (live && pg < n) ? (n - pg) : 0u;
        const uint32_t maxd = rem < LM ? rem : LM;
        uint32_t cur = 0, base = P.root_base;
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
                    if (DROPOUT) {  // model.rs:100: kept iff len <= 1 || dropout < rand
                        if (term && d >= 1) term = P.dropout < dropout_u01(P.seed, s, pg, (uint32_t)d + 1u);
                    }
                    if (term) scw[((uint32_t)d + l) & 31u] = __hiloint2double((int)rec.w, (int)rec.z);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        {   // the following block's text window, requested after the walk's last gather: it lands during the relax
            const uint32_t* __restrict__ np = wp + 8;  // + 32 bytes, same alignment
#pragma unroll
            for (int q = 0; q < 9; ++q) wn[q] = np[q];
        }

        // ---- relax: 32 static steps, two positions (one per row) per step; the LDS reads of a half are
        // issued before its dependent chain starts
        uint32_t fin = kNoStep2, fhi = 0xFFF00000u;
        {
            const double* scr = sc + r * 1024u + ((l - 1u) & 31u);  // row (r*32 + U), column (l - 1) & 31
            double sv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) sv[u] = scr[u * 32];
            relax2_step<0>(sv[0], acc, bpv, fin, fhi);
            relax2_step<1>(sv[1], acc, bpv, fin, fhi);
            relax2_step<2>(sv[2], acc, bpv, fin, fhi);
            relax2_step<3>(sv[3], acc, bpv, fin, fhi);
            relax2_step<4>(sv[4], acc, bpv, fin, fhi);
            relax2_step<5>(sv[5], acc, bpv, fin, fhi);
            relax2_step<6>(sv[6], acc, bpv, fin, fhi);
            relax2_step<7>(sv[7], acc, bpv, fin, fhi);
            relax2_step<8>(sv[8], acc, bpv, fin, fhi);
            relax2_step<9>(sv[9], acc, bpv, fin, fhi);
            relax2_step<10>(sv[10], acc, bpv, fin, fhi);
            relax2_step<11>(sv[11], acc, bpv, fin, fhi);
            relax2_step<12>(sv[12], acc, bpv, fin, fhi);
            relax2_step<13>(sv[13], acc, bpv, fin, fhi);
            relax2_step<14>(sv[14], acc, bpv, fin, fhi);
            relax2_step<15>(sv[15], acc, bpv, fin, fhi);
#pragma unroll
            for (int u = 0; u < 16; ++u) sv[u] = scr[(u + 16) * 32];
            relax2_step<16>(sv[0], acc, bpv, fin, fhi);
            relax2_step<17>(sv[1], acc, bpv, fin, fhi);
            relax2_step<18>(sv[2], acc, bpv, fin, fhi);
            relax2_step<19>(sv[3], acc, bpv, fin, fhi);
            relax2_step<20>(sv[4], acc, bpv, fin, fhi);
            relax2_step<21>(sv[5], acc, bpv, fin, fhi);
            relax2_step<22>(sv[6], acc, bpv, fin, fhi);
            relax2_step<23>(sv[7], acc, bpv, fin, fhi);
            relax2_step<24>(sv[8], acc, bpv, fin, fhi);
            relax2_step<25>(sv[9], acc, bpv, fin, fhi);
            relax2_step<26>(sv[10], acc, bpv, fin, fhi);
            relax2_step<27>(sv[11], acc, bpv, fin, fhi);
            relax2_step<28>(sv[12], acc, bpv, fin, fhi);
            relax2_step<29>(sv[13], acc, bpv, fin, fhi);
            relax2_step<30>(sv[14], acc, bpv, fin, fhi);
            relax2_step<31>(sv[15], acc, bpv, fin, fhi);
        }
        __builtin_amdgcn_wave_barrier();
        const bool reached = fhi != 0xFFF00000u;  // high word of -inf

        // ---- back-pointer of this lane's position (plain bytes: index = end position - 1), next block
        if (live && pg >= 1u && pg <= n) {
            // winner pushed at step U = fin into lane l: token length ((l - U - 1) & 31) + 1
            const uint8_t b = reached ? (uint8_t)((l - fin - 1u) & 31u) : (uint8_t)0xFF;
            __builtin_nontemporal_store(b, P.bp8 + bp8_base(beg, s) + (PERM ? bp8_perm(pg - 1u) : pg - 1u));
        }
        if (live) {
            const uint32_t left = n - p0;
            if (left < 32u) {  // position n lies in this block: the sample is done
                if (left == l) P.status[s] = (n == 0u || reached) ? 1u : 0u;
                need_new = true;
            } else {
                p0 += 32u;
            }
        }
    }
}

// trace_kernel (kernels.hip) for tokens of up to 32 bytes (trace_body.h).
// PERM: the back-pointer bytes are in encode4_kernel's permuted layout (bp8_perm; encode4l_kernel), else plain
template <bool PERM, bool CARRY>
__global__ __launch_bounds__(256) void trace32_kernel(EncodeParams P) {
    __shared__ typename TraceRingEntry<CARRY>::type ring_all[4][kTraceRing];
    trace_body<32, PERM, false, CARRY>(P, ring_all[threadIdx.x >> 6]);
}

// two blocks of five waves per CU: 10 x 16 KiB of LDS
hipError_t launch_encode2(const EncodeParams& p, uint32_t num_cus, bool permuted, hipStream_t stream) {
    const uint32_t waves = 5, bpc = 2;
    const uint64_t want = (p.n_samples + 2 * waves - 1) / (2 * waves);
    const uint32_t blocks = (uint32_t)(want < (uint64_t)num_cus * bpc ? (want ? want : 1) : (uint64_t)num_cus * bpc);
    auto fn = p.dropout > 0.0 ? (permuted ? encode2_kernel<true, true> : encode2_kernel<true, false>)
                              : (permuted ? encode2_kernel<false, true> : encode2_kernel<false, false>);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(64u * waves), waves * kRows2Bytes, stream, p);
    return hipGetLastError();
}
hipError_t launch_trace32(const EncodeParams& p, uint32_t blocks, bool permuted, hipStream_t stream) {
    auto fn = permuted ? (p.trace_carry ? trace32_kernel<true, true> : trace32_kernel<true, false>)
                       : (p.trace_carry ? trace32_kernel<false, true> : trace32_kernel<false, false>);
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, stream, p);
    return hipGetLastError();
}

}  // namespace tgx
