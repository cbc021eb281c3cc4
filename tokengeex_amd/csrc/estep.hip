// E-step kernel: expected token counts by forward/backward over the full lattice.
//
// Replaces Model::populate_nodes (reference src/model.rs:34-55) +
// Lattice::populate_marginal (src/lattice.rs:245-333) + the snippet loop of
// run_e_step (src/prune.rs:64-120).  One wavefront per sample, snippets of
// <= snippet_len bytes in turn; per snippet two sweeps with the same machinery as
// the encode kernel (64 positions matched in parallel, then finalised in order with
// lane j accumulating the position congruent to j mod 64), max replaced by the
// reference's log_sum_exp:
//
//   forward  : A[p]   = LSE over tokens ending at p   of (score + A[start])   (alpha)
//              pushes arrive in ascending start order, first one assigned
//              (lattice.rs:259-272, 321-333); A[p] goes to an HBM scratch row; z = A[n].
//   backward : B[q]   = LSE over tokens starting at q of (score + B[end])     (beta)
//              run as the SAME push recursion on the reversed text with a trie of the
//              reversed tokens: finalising end position q pushes score + B[q] to every
//              start p = q - L.  While pushing, each token (p, L) adds
//              exp(((A[p] + score) + B[q]) - z) to its expected count (lattice.rs:295-309).
//
// Positions without incoming tokens keep the reference's 0.0 (its vectors are
// zero-initialised, lattice.rs:255-256).  Expected counts are accumulated per slot of
// the reversed trie with f64 atomics and mapped to token ids on the host.  The
// backward fold visits a position's tokens in descending instead of ascending length
// and exp/log are the device's: results agree with the reference to rounding
// (tests: 1e-9 relative), not bitwise.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_common.h"
#include "kernels.h"

namespace tgx {

// log_sum_exp(x, y, init_mode = false) — reference src/lattice.rs:321-333
__device__ __forceinline__ double log_sum_exp(double x, double y) {
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

// LDS per wave: the encode layout (scores + handles) plus a row of forward values.
__host__ __device__ inline uint32_t estep_wave_lds_bytes(uint32_t lm) {
    return wave_lds_entries(lm) * 12u + 128u + (64u + 64u + 2u) * 8u;
}

// One sweep over a snippet.  BACKWARD = false: forward values to Arow[0..sn].
// BACKWARD = true: sweep coordinate y = sn - q, reversed text / reversed trie, adds
// the marginals.  Returns (wave-uniform) the value at the far end: A[sn] or B[0].
template <bool BACKWARD>
__device__ __forceinline__ double estep_sweep(const EstepParams& P, const uint8_t* __restrict__ snip, uint32_t sn,
                                              uint32_t s, uint64_t snippet_base, uint32_t lane, uint32_t LM,
                                              double* sc, uint32_t* hl, uint8_t* txt, double* abuf,
                                              double* __restrict__ Arow, double z, bool use_dropout) {
    const uint4* __restrict__ trie = reinterpret_cast<const uint4*>(BACKWARD ? P.trie_rev : P.trie_fwd);
    // one of n_replicas copies of the slot array (summed afterwards): hot tokens would
    // otherwise serialise every wave's atomics
    double* __restrict__ expected_slot = P.expected_slot + (size_t)(blockIdx.x % P.n_replicas) * P.n_slots_rev;
    const uint32_t root_base = BACKWARD ? P.root_rev : P.root_fwd;
    double acc = 0.0;    // value of the position this lane accumulates
    uint64_t reach = 1;  // bit j: lane j has received a push (coordinate 0: BOS / EOS, value 0.0)
    double end_value = 0.0;

    for (uint32_t x0 = 0; x0 <= sn; x0 += 64) {
        const uint32_t x = x0 + lane;
        // stage the block's bytes in sweep order: forward snip[x + k], backward snip[sn - 1 - (x + k)]
        {
            const uint32_t k0 = x, k1 = x + 64;
            txt[lane] = (k0 < sn) ? (BACKWARD ? snip[sn - 1 - k0] : snip[k0]) : (uint8_t)0;
            txt[lane + 64] = (k1 < sn) ? (BACKWARD ? snip[sn - 1 - k1] : snip[k1]) : (uint8_t)0;
            if (BACKWARD) {
                // forward values of the start positions this block can reach:
                // abuf[k] = A[sn - (x0 + k)], k = 1 .. 64 + LM
                for (uint32_t k = lane; k <= 64u + LM; k += 64u) {
                    const uint32_t yy = x0 + k;
                    abuf[k] = (k >= 1u && yy <= sn) ? Arow[sn - yy] : 0.0;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();

        // match: every token that begins (forward) / ends (backward) at this lane's position
        const uint32_t rem = (x < sn) ? (sn - x) : 0u;
        const uint32_t maxd = rem < LM ? rem : LM;
        uint32_t cur = 0, base = root_base;
        uint64_t m = 0;
        bool alive = maxd > 0;
        for (uint32_t d = 0; d < LM; ++d) {
            alive = alive && (d < maxd);
            if (__builtin_amdgcn_ballot_w64(alive) == 0) break;
            if (alive) {
                const uint32_t c = txt[lane + d];
                const uint32_t t = base ^ c;
                const uint4 r = load_rec(trie, t);
                if (r.x == cur) {
                    cur = t;
                    base = r.y & 0x7FFFFFFFu;
                    if (r.y >> 31) {
                        bool keep = true;
                        if (use_dropout && d >= 1) {
                            // model.rs:48: node skipped iff len > 1 && dropout > 0 && rand < dropout;
                            // the draw is keyed by the token's START byte in the sample
                            const uint32_t len = d + 1;
                            const uint64_t start = BACKWARD ? (uint64_t)(sn - x - len) : (uint64_t)x;
                            keep = !(dropout_u01(P.seed, s, snippet_base + start, len) < P.dropout);
                        }
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

        // finalise coordinates x0 .. min(x0 + 63, sn) in order, pushing to x + L
        const uint32_t left = sn - x0;
        const uint32_t steps = left < 64u ? left : 64u;
        double fin = 0.0;
        for (uint32_t i = 0; i < steps; ++i) {
            const bool had = (reach >> i) & 1ULL;
            // a position nothing was pushed to keeps 0.0 (lattice.rs:255-256)
            const double val = had ? readlane_f64(acc, i) : 0.0;
            fin = (lane == i) ? val : fin;
            const uint64_t mi = readlane_u64(m, i);
            reach &= ~(1ULL << i);
            if (mi == 0) continue;
            const uint64_t active = rotl64(mi, i + 1);
            const uint32_t tj = (lane - i - 1u) & 63u;  // len - 1 for this lane
            const double sv = sc[kFront + i * LM + tj];
            const double pushed = sv + val;  // lattice.rs:267 / :282: score + alpha|beta
            const bool act = (active >> lane) & 1ULL;
            if (BACKWARD && act) {
                // marginal of token (p = q - len, len): lattice.rs:305-307
                const uint32_t hv = hl[kFront + i * LM + tj];
                const double a = abuf[i + tj + 1u];
                const double total = ((a + sv) + val) - z;
                atomicAdd(&expected_slot[hv >> 6], exp(total));
            }
            const bool first = !((reach >> lane) & 1ULL);
            const double merged = first ? pushed : log_sum_exp(acc, pushed);  // init_mode, lattice.rs:322-323
            acc = act ? merged : acc;
            reach |= active;
        }
        if (left < 64u) {  // the far end (forward: n, backward: position 0) lies in this block
            const bool had = (reach >> left) & 1ULL;
            const double val = had ? readlane_f64(acc, left) : 0.0;
            fin = (lane == left) ? val : fin;
            end_value = val;
        }
        if (!BACKWARD && x <= sn) Arow[x] = fin;
        __builtin_amdgcn_wave_barrier();
    }
    return end_value;
}

__global__ __launch_bounds__(256) void estep_kernel(EstepParams P) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t LM = P.lm;
    const uint32_t entries = wave_lds_entries(LM);
    unsigned char* wbase = smem + (size_t)wave * estep_wave_lds_bytes(LM);
    double* sc = reinterpret_cast<double*>(wbase);
    uint32_t* hl = reinterpret_cast<uint32_t*>(wbase + (size_t)entries * 8u);
    uint8_t* txt = reinterpret_cast<uint8_t*>(wbase + (size_t)entries * 12u);
    double* abuf = reinterpret_cast<double*>(wbase + (size_t)entries * 12u + 128u);
    const bool use_dropout = P.dropout > 0.0;

    const uint32_t wpb = blockDim.x >> 6;
    const uint32_t n_waves = gridDim.x * wpb;
    const uint32_t wave_id = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * wpb + wave));
    double zsum = 0.0;
    for (uint64_t k = wave_id; k < P.n_samples; k += n_waves) {
        const uint32_t s = (uint32_t)__builtin_amdgcn_readfirstlane((int)P.order[k]);
        const uint64_t beg = first_u64(P.offs[s]);
        const uint64_t n = first_u64(P.offs[s + 1]) - beg;
        // sample.as_bytes().chunks(MAX_SAMPLE_LENGTH) — prune.rs:83 (no paths cross a cut)
        for (uint64_t b0 = 0; b0 < n; b0 += P.snippet_len) {
            const uint32_t sn = (uint32_t)((n - b0 < P.snippet_len) ? (n - b0) : P.snippet_len);
            const uint8_t* __restrict__ snip = P.text + beg + b0;
            double* __restrict__ Arow = P.alpha + beg + b0 + s;  // sn + 1 values; + s: one extra slot per sample
            const double z = estep_sweep<false>(P, snip, sn, s, b0, lane, LM, sc, hl, txt, abuf, Arow, 0.0, use_dropout);
            __threadfence_block();  // Arow is read back by this wave in the backward sweep
            // !z.is_normal() panics in the reference (prune.rs:90-96): zero, subnormal, inf, NaN
            const double az = fabs(z);
            if (!(az >= 2.2250738585072014e-308 && az <= 1.7976931348623157e308)) {
                if (lane == 0) atomicMin(P.err_sample, (unsigned long long)s);
            }
            zsum += z;
            estep_sweep<true>(P, snip, sn, s, b0, lane, LM, sc, hl, txt, abuf, Arow, z, use_dropout);
        }
    }
    if (lane == 0 && zsum != 0.0) atomicAdd(P.logz_sum, zsum);
}

uint32_t estep_waves_per_block(uint32_t lm) {
    uint32_t w = (64u * 1024u) / estep_wave_lds_bytes(lm);
    return w < 1u ? 1u : (w > 4u ? 4u : w);
}
uint32_t estep_lds_bytes_per_block(uint32_t lm) { return estep_waves_per_block(lm) * estep_wave_lds_bytes(lm); }

hipError_t estep_max_blocks_per_cu(uint32_t lm, int* out) {
    const uint32_t lds = estep_lds_bytes_per_block(lm);
    if (lds > 160u * 1024u) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(estep_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(out, estep_kernel, (int)(64u * estep_waves_per_block(lm)), lds);
}

hipError_t launch_estep(const EstepParams& p, uint32_t blocks, hipStream_t stream) {
    hipLaunchKernelGGL(estep_kernel, dim3(blocks), dim3(64u * estep_waves_per_block(p.lm)),
                       estep_lds_bytes_per_block(p.lm), stream, p);
    return hipGetLastError();
}

}  // namespace tgx
