// Viterbi encode on 16-lane rows (four samples per wave) for vocabularies whose longest token has 17..32 bytes.
//
// encode4_kernel (kernels.hip) needs max token <= 16: lane l of a row accumulates the one end position
// e == l (mod 16) that lies 1..16 positions ahead.  Tokens of 17..32 bytes are RARE matches (a merged
// vocabulary has a few hundred of them), so this kernel keeps that structure and adds, per lane, a second
// accumulator `far` for the position 17..32 ahead:
//
//   * the walk goes on past depth 16 only where the trie continues (the ballot ends it for the wave); a match
//     of 17..32 bytes is not written to the LDS match buffer but appended to a small per-wave overflow list;
//   * when lane U is restarted at step U (it starts accumulating position p0 + U + 16) it starts from `far`
//     — the best long candidate pushed to that position so far — instead of nothing, and `far` restarts empty;
//   * after the 16 steps of a block the overflow entries of the block are applied, one at a time: candidate
//     best[start] + score into `acc` (target 17..31 ahead of the block's first position) or `far` (32..47).
//
// Order of candidates: the reference relaxes starts in ascending order with a strict '>' (model.rs:96-108), so
// among candidates of equal score for one end position the EARLIEST start, i.e. the LONGEST token, wins.  The
// 16 steps deliver near candidates in that order; a long candidate is applied out of order, so it replaces
// an equal score iff its token is longer than the current winner's (and `far` wins ties against near
// candidates when a lane restarts: everything in `far` starts earlier than any near start of that position).
//
// An overflow list that fills up (kOvfCap entries per wave and block; e.g. a run of blanks when the vocabulary
// has several long blank tokens) puts the wave's current samples on P.redo_list: the host then redoes exactly
// those samples with the two-samples-per-wave kernel (encode2.hip), which has no such limit.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_common.h"
#include "kernels.h"

namespace tgx {

constexpr uint32_t kL4GroupBytes = 64u * 16u * 8u;  // match buffer, 8 KiB per wave
constexpr uint32_t kOvfCap = 62;                    // overflow entries per wave and block
constexpr uint32_t kL4WaveBytes = kL4GroupBytes + 1024u;  // + {count, pad} + 62 x 16 B + pad
constexpr uint32_t kFarFlag = 0x80u;                // winner code: kFarFlag | (length - 1) for tokens of 17..32 bytes

// winner code -> token length - 1; near winners are remembered as the step that pushed them (as in encode4_kernel)
__device__ __forceinline__ uint32_t winner_len_m1(uint32_t code, uint32_t l) {
    return (code & kFarFlag) ? (code & 31u) : ((l - code - 1u) & 15u);
}

template <int U>
__device__ __forceinline__ void relax4l_step(double sv, double& acc, uint32_t& bpv, double& far, uint32_t& fbp,
                                             uint32_t& fin, double& fval) {
    constexpr uint64_t MU = kRowLane0 << U;  // lanes with l == U
    const double ninf = -__builtin_huge_val();
    fin = sel_u32(MU, bpv, fin);             // winner of position p0 + U is final now
    fval = sel_f64(MU, acc, fval);           // and its score (needed by the long candidates of this block)
    const double best = row_bcast_f64<U>(acc);
    // lane U restarts on position p0 + U + 16: from the long candidates collected for it
    const double cur = sel_f64(MU, far, acc);
    const uint32_t curbp = sel_u32(MU, fbp, bpv);
    far = sel_f64(MU, ninf, far);            // `far` of lane U now stands for position p0 + U + 32
    const double cand = best + sv;           // model.rs:98
    const uint64_t take = __builtin_amdgcn_fcmp(cand, cur, 2 /* OGT: model.rs:101; far wins ties */);
    asm("v_max_f64 %0, %1, %2" : "=v"(acc) : "v"(cur), "v"(cand));  // = take ? cand : cur (finite or -inf values), off the compare (relax5_step)
    bpv = sel_imm_u32<U>(take, curbp);
}

template <bool DROPOUT>
__global__ __launch_bounds__(512, 4) void encode4l_kernel(EncodeParams P) {  // two blocks of eight waves: four per SIMD, <= 128 VGPRs
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr uint32_t LM = 32;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t l = lane & 15u, r = lane >> 4;
    const uint32_t wave = threadIdx.x >> 6;
    const uint4* __restrict__ trie = reinterpret_cast<const uint4*>(P.trie);
    // the 256 records of the root's children in LDS: the first step of every walk reads them there (as encode4_kernel)
    const uint4* rootc = reinterpret_cast<const uint4*>(smem);
    {
        uint4* rw = reinterpret_cast<uint4*>(smem);
        for (uint32_t i = threadIdx.x; i < 256u; i += blockDim.x) rw[i] = trie[(P.root_base & ~255u) + i];
        __syncthreads();
    }
    unsigned char* wbase = smem + 4096u + (size_t)wave * kL4WaveBytes;
    double* sc = reinterpret_cast<double*>(wbase);
    uint32_t* ovf_cnt = reinterpret_cast<uint32_t*>(wbase + kL4GroupBytes);
    uint4* ovf = reinterpret_cast<uint4*>(wbase + kL4GroupBytes + 16u);  // {lane | depth << 8, -, score lo, score hi}

    uint32_t s = 0, n = 0, p0 = 0;
    uint64_t beg = 0;
    bool live = false, need_new = true;
    const double ninf = -__builtin_huge_val();
    double acc = ninf, far = ninf;
    uint32_t bpv = 0, fbp = kFarFlag;
    uint32_t wn[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) wn[q] = 0;
    uint32_t pk = 0, pk_j = 0;
    bool pk_dirty = false;
    bool redo = false;  // this row's sample is on the redo list already

    for (;;) {
        {
            const uint64_t k = claim_rows(P.queue, need_new, r);
            if (need_new) {
                live = k < P.n_samples;
                if (live) {
                    s = P.order[k];
                    beg = P.offs[s];
                    n = (uint32_t)(P.offs[s + 1] - beg);
                }
                p0 = 0;
                acc = (l == 0u) ? 0.0 : ninf;  // position 0: score 0, reachable
                far = ninf;
                bpv = 0;
                fbp = kFarFlag;
                pk_dirty = false;
                redo = false;
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
            double2* grp = reinterpret_cast<double2*>(sc);
#pragma unroll
            for (int q = 0; q < 8; ++q) grp[q * 64 + lane] = make_double2(ninf, ninf);
            if (lane == 0) *ovf_cnt = 0;
        }
        __builtin_amdgcn_wave_barrier();
        const uint32_t pg = p0 + l;
        const uint32_t rem = (live && pg < n) ? (n - pg) : 0u;
        const uint32_t maxd = rem < LM ? rem : LM;
        uint32_t cur = 0, base = P.root_base;
        bool alive = maxd > 0;
        double* scw = sc + lane * 16u;
#pragma unroll
        for (int d = 0; d < (int)LM; ++d) {
            alive = alive && ((uint32_t)d < maxd);
            if (__builtin_amdgcn_ballot_w64(alive) == 0) break;
            if (alive) {
                const uint32_t c = (bytes[d >> 2] >> ((d & 3) * 8)) & 0xFFu;
                const uint32_t t = base ^ c;
                const uint4 rec = d == 0 ? rootc[t & 255u] : load_rec(trie, t);
                alive = rec.x == cur;
                if (alive) {
                    cur = t;
                    base = rec.y & 0x7FFFFFFFu;
                    bool term = (rec.y >> 31) != 0u;
                    if (DROPOUT) {  // model.rs:100: kept iff len <= 1 || dropout < rand
                        if (term && d >= 1) term = P.dropout < dropout_u01(P.seed, s, pg, (uint32_t)d + 1u);
                    }
                    if (term) {
                        if (d < 16) {
                            scw[((uint32_t)d + l) & 15u] = __hiloint2double((int)rec.w, (int)rec.z);
                        } else {  // a token of 17..32 bytes: overflow list
                            const uint32_t slot = atomicAdd(ovf_cnt, 1u);
                            if (slot < kOvfCap) ovf[slot] = make_uint4(lane | ((uint32_t)d << 8), 0u, rec.z, rec.w);
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        {   // the following block's text window: it lands while the relax runs
            const uint32_t* __restrict__ np = wp + 4;  // + 16 bytes, same alignment
#pragma unroll
            for (int q = 0; q < 9; ++q) wn[q] = np[q];
        }

        // ---- relax: 16 static steps, four positions (one per row) per step
        uint32_t fin = 0;
        double fval = ninf;
        {
            const double* scr = sc + r * 256u + ((l - 1u) & 15u);
            double sv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) sv[u] = scr[u * 16];
            relax4l_step<0>(sv[0], acc, bpv, far, fbp, fin, fval);
            relax4l_step<1>(sv[1], acc, bpv, far, fbp, fin, fval);
            relax4l_step<2>(sv[2], acc, bpv, far, fbp, fin, fval);
            relax4l_step<3>(sv[3], acc, bpv, far, fbp, fin, fval);
            relax4l_step<4>(sv[4], acc, bpv, far, fbp, fin, fval);
            relax4l_step<5>(sv[5], acc, bpv, far, fbp, fin, fval);
            relax4l_step<6>(sv[6], acc, bpv, far, fbp, fin, fval);
            relax4l_step<7>(sv[7], acc, bpv, far, fbp, fin, fval);
            relax4l_step<8>(sv[8], acc, bpv, far, fbp, fin, fval);
            relax4l_step<9>(sv[9], acc, bpv, far, fbp, fin, fval);
            relax4l_step<10>(sv[10], acc, bpv, far, fbp, fin, fval);
            relax4l_step<11>(sv[11], acc, bpv, far, fbp, fin, fval);
            relax4l_step<12>(sv[12], acc, bpv, far, fbp, fin, fval);
            relax4l_step<13>(sv[13], acc, bpv, far, fbp, fin, fval);
            relax4l_step<14>(sv[14], acc, bpv, far, fbp, fin, fval);
            relax4l_step<15>(sv[15], acc, bpv, far, fbp, fin, fval);
        }
        // ---- the block's long candidates.  After the 16 steps lane j of a row accumulates position
        // p0 + 16 + j in `acc` and p0 + 32 + j in `far`; entry (start lane, depth d) is the token of d + 1
        // bytes starting at p0 + (start lane & 15): it ends 17..47 positions after p0.
        {
            const uint32_t cnt_all = (uint32_t)__builtin_amdgcn_readfirstlane((int)*ovf_cnt);
            const uint32_t cnt = cnt_all < kOvfCap ? cnt_all : kOvfCap;
            if (cnt_all > kOvfCap) {  // wave-uniform: entries were dropped, whichever row they belonged to
                if (live && !redo && l == 0u) P.redo_list[atomicAdd(P.redo_count, 1ULL)] = s;
                redo = true;
            }
            for (uint32_t i = 0; i < cnt; ++i) {
                const uint4 e = ovf[i];
                const uint32_t src = (uint32_t)__builtin_amdgcn_readfirstlane((int)(e.x & 63u));
                const uint32_t d = (uint32_t)__builtin_amdgcn_readfirstlane((int)(e.x >> 8));
                const double score = __hiloint2double((int)e.w, (int)e.z);
                const double cand = readlane_f64(fval, src) + score;  // model.rs:98
                const uint32_t tt = (src & 15u) + d + 1u;             // end position - p0: 17 .. 47
                const bool mine = (lane >> 4) == (src >> 4) && l == (tt & 15u);
                const bool to_far = tt >= 32u;
                const double curv = to_far ? far : acc;
                const uint32_t curc = to_far ? fbp : bpv;
                // longer token = earlier start: it wins an equal score (see the header); a near winner is
                // always shorter than a long candidate
                const bool longer = !(curc & kFarFlag) || d > (curc & 31u);
                const bool take = mine && (cand > curv || (cand == curv && longer));
                if (take && to_far) {
                    far = cand;
                    fbp = kFarFlag | d;
                }
                if (take && !to_far) {
                    acc = cand;
                    bpv = kFarFlag | d;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        const bool reached = fval > ninf;

        // ---- back-pointers (length - 1, 0..31; permuted and packed as in encode4_kernel), next block
        uint8_t* const bpw = P.bp8 + bp8_base(beg, s);
        if (live && pg >= 1u && pg <= n) {
            const uint32_t b = reached ? winner_len_m1(fin, l) : 0xFFu;
            const uint32_t j = pg - 1u, kq = (j >> 4) & 3u;
            pk = (kq == 0u) ? b : (pk | (b << (8u * kq)));
            pk_j = j;
            pk_dirty = true;
            if (kq == 3u) {
                __builtin_nontemporal_store(pk, reinterpret_cast<uint32_t*>(bpw + (bp8_perm(j) & ~3u)));
                pk_dirty = false;
            }
        }
        if (live && n - p0 < 16u && pk_dirty) {
            __builtin_nontemporal_store(pk, reinterpret_cast<uint32_t*>(bpw + (bp8_perm(pk_j) & ~3u)));
            pk_dirty = false;
        }
        if (live) {
            const uint32_t left = n - p0;
            if (left < 16u) {  // position n lies in this block: the sample is done
                if (left == l) P.status[s] = (n == 0u || reached) ? 1u : 0u;
                need_new = true;
            } else {
                p0 += 16u;
            }
        }
    }
}

// two blocks of eight waves per CU: 16 x 9 KiB of LDS
hipError_t launch_encode4l(const EncodeParams& p, uint32_t num_cus, hipStream_t stream) {
    const uint32_t waves = 8, bpc = 2;
    const uint64_t want = (p.n_samples + 4 * waves - 1) / (4 * waves);
    const uint32_t blocks = (uint32_t)(want < (uint64_t)num_cus * bpc ? (want ? want : 1) : (uint64_t)num_cus * bpc);
    auto fn = p.dropout > 0.0 ? encode4l_kernel<true> : encode4l_kernel<false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(64u * waves), 4096u + waves * kL4WaveBytes, stream, p);
    return hipGetLastError();
}

}  // namespace tgx
