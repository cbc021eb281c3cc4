// Cut points of the token lattice (gfx950 / CDNA4, wave64) — round 3.
//
// A snippet of the E-step (src/prune.rs:83-86: a sample, in chunks of 81 920 bytes) is a serial chain: alpha[p] needs
// alpha of the 16 positions before it (src/lattice.rs:259-287), so one row of 16 lanes walks a 64 KiB sample for
// 9 - 29 ms however idle the chip is, and a 256 MiB shard is bound by its longest samples (forward 9.2 + backward
// 22.9 ms where throughput gives 5 + 10; profiles/r02).  But the lattice FACTORISES at a position q that no token
// match crosses: every path from the start to the end goes through the node q, so
//     alpha[x] = alpha[q] * alpha_q[x]  (x >= q),    beta[x] = beta_q'[x] * beta[q]  (x <= q),    Z = Z_left * Z_right,
// and the marginal of a token left of q, alpha[s] w beta[e] / Z (lattice.rs:305-307), is alpha[s] w beta_q'[e] / Z_left:
// the two sides are independent lattices whose expected counts and log Z simply add.  Such positions are frequent —
// on the bench corpus one every 4.5 bytes on average and at worst 72 bytes apart, with the 32 000- and the
// 500 000-entry vocabulary alike (a position is crossed iff a vocabulary token matches the text across it) — so a
// long snippet is cut into PIECES of about `window` bytes (2 048), which the kernels of estep4l.hip take as
// snippets of their own: the pass becomes bound by throughput whatever the sample lengths.
//
// Whether q is a cut depends only on the matches that start in [q - lmx, q): max over them of (start + longest
// match) must be exactly q — not larger (a token would cross), and not smaller: then some token ENDS at q, and it
// starts at or after the previous cut, so the piece's end is reachable inside the piece whenever it is in the whole
// snippet.  (A position nothing reaches is lattice.rs:255's corner: the linear-domain kernels raise range_flag for
// it with or without pieces, and the host redoes such a pass with the log-domain kernels on the UNCUT snippets.)
// The same dropout decisions as the E-step's walks are applied (model.rs:48), so the matches seen here are exactly the
// lattice's.  One wave per window of a snippet: it walks 64 positions at a time from `lmx` before the window and
// stops at the first cut it finds — about one trip per window, 3 % of a full walk of the text.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>

#include <rocprim/rocprim.hpp>

#include "device_common.h"
#include "kernels.h"

namespace tgx {

template <bool DROPOUT>
__global__ __launch_bounds__(256) void cut_windows_kernel(CutParams P) {
    __shared__ uint32_t reach_s[4][64];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t w = (uint64_t)blockIdx.x * 4ull + wave;
    if (w >= P.n_windows) return;  // wave-uniform
    const uint4* __restrict__ trie = reinterpret_cast<const uint4*>(P.trie);
    const uint32_t s = P.win_snip[w], k = P.win_k[w];
    const uint64_t b = P.soffs[s], e = P.soffs[s + 1];
    uint64_t found = ~0ull;
    if (k == 0u) {
        found = b;  // a snippet's start is a boundary as it is
    } else {
        const uint32_t lmx = P.lmx;
        const uint64_t q_lo = b + (uint64_t)k * P.window;                      // candidates q_lo <= q < q_hi
        const uint64_t q_hi = (q_lo + P.window < e) ? q_lo + P.window : e;
        uint32_t smp = 0;
        uint64_t sbase = 0;
        if (DROPOUT) {
            smp = P.snip_sample[s];
            sbase = P.snip_base[s];
        }
        for (uint64_t pos0 = q_lo - lmx; pos0 + lmx < q_hi; pos0 += 64u - lmx) {
            const uint64_t pp = pos0 + lane;  // >= b: windows 1.. start `window` > lmx bytes into the snippet
            bool alive = pp < e;
            const uint32_t maxd = alive ? (uint32_t)((e - pp < lmx) ? e - pp : lmx) : 0u;
            uint32_t cur = 0, base = P.root, longest = 0;
            for (uint32_t d = 0; d < lmx; ++d) {
                alive = alive && d < maxd;
                if (__builtin_amdgcn_ballot_w64(alive) == 0) break;
                const uint32_t c = alive ? (uint32_t)P.text[pp + d] : 0u;
                const uint32_t t = alive ? (base ^ c) : 0u;
                const uint4 rec = load_rec(trie, t);
                alive = alive && rec.x == cur;
                if (alive) {
                    cur = t;
                    base = rec.y & 0x7FFFFFFFu;
                    bool term = (rec.y >> 31) != 0u;
                    if (DROPOUT) {  // model.rs:48: skipped iff len > 1 && rand < dropout
                        if (term && d >= 1u) term = !(dropout_u01(P.seed, smp, sbase + (pp - b), d + 1u) < P.dropout);
                    }
                    if (term) longest = d + 1u;
                }
            }
            reach_s[wave][lane] = lane + longest;  // relative to pos0
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            bool valid = false;
            const uint64_t q = pos0 + lane;
            if (lane >= lmx && q >= q_lo && q < q_hi) {
                uint32_t m = 0;
                for (uint32_t i = 1; i <= lmx; ++i) {
                    const uint32_t v = reach_s[wave][lane - i];
                    m = v > m ? v : m;
                }
                valid = m == lane;
            }
            __builtin_amdgcn_wave_barrier();
            const uint64_t hits = __builtin_amdgcn_ballot_w64(valid);
            if (hits != 0) {
                found = pos0 + (uint64_t)__builtin_ctzll(hits);
                break;
            }
        }
    }
    if (lane == 0u) {
        P.bound[w] = found;
        P.flag[w] = found != ~0ull ? 1u : 0u;
    }
}

hipError_t launch_cut_windows(const CutParams& p, hipStream_t stream) {
    if (p.n_windows == 0) return hipSuccess;
    if (p.lmx > 32u || p.window < 4u * p.lmx || p.window < 128u) return hipErrorInvalidValue;
    const uint32_t blocks = (uint32_t)((p.n_windows + 3) / 4);
    if (p.dropout > 0.0) hipLaunchKernelGGL(cut_windows_kernel<true>, dim3(blocks), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(cut_windows_kernel<false>, dim3(blocks), dim3(256), 0, stream, p);
    return hipGetLastError();
}

// boundaries found -> the piece list: offsets (text order), dropout-hash bookkeeping, snippet of each piece
__global__ __launch_bounds__(256) void cut_scatter_kernel(CutParams P, const uint64_t* __restrict__ pos, uint64_t n_bytes,
                                                          uint64_t* __restrict__ poffs, uint32_t* __restrict__ psample,
                                                          uint64_t* __restrict__ pbase, uint32_t* __restrict__ psnip) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w == 0) poffs[pos[P.n_windows]] = n_bytes;
    if (w >= P.n_windows || P.flag[w] == 0u) return;
    const uint64_t j = pos[w], q = P.bound[w];
    const uint32_t s = P.win_snip[w];
    poffs[j] = q;
    psample[j] = P.snip_sample[s];
    pbase[j] = P.snip_base[s] + (q - P.soffs[s]);
    psnip[j] = s;
}
hipError_t launch_cut_scatter(const CutParams& p, const uint64_t* pos, uint64_t n_bytes, uint64_t* poffs, uint32_t* psample,
                              uint64_t* pbase, uint32_t* psnip, hipStream_t stream) {
    const uint32_t blocks = (uint32_t)((p.n_windows + 255) / 256 + 1);
    hipLaunchKernelGGL(cut_scatter_kernel, dim3(blocks), dim3(256), 0, stream, p, pos, n_bytes, poffs, psample, pbase, psnip);
    return hipGetLastError();
}

// piece lengths (the keys of the longest-first order) and the longest one; `n_pieces` is read on the device: the host
// has not seen it yet
__global__ __launch_bounds__(256) void piece_len_kernel(const uint64_t* __restrict__ poffs, const uint64_t* __restrict__ n_pieces,
                                                        uint32_t* __restrict__ len, uint32_t* __restrict__ idx,
                                                        unsigned long long* __restrict__ longest, uint64_t cap) {
    const uint64_t n = *n_pieces;
    unsigned long long mx = 0;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < cap; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t v = j < n ? (uint32_t)(poffs[j + 1] - poffs[j]) : 0u;  // (entries past the list sort to its end)
        len[j] = v;
        idx[j] = (uint32_t)j;
        mx = v > mx ? v : mx;
    }
    if (mx) atomicMax(longest, mx);
}
hipError_t launch_piece_len(const uint64_t* poffs, const uint64_t* n_pieces, uint32_t* len, uint32_t* idx, unsigned long long* longest,
                            uint64_t cap, hipStream_t stream) {
    const uint32_t blocks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((cap + 255) / 256, 2048));
    hipLaunchKernelGGL(piece_len_kernel, dim3(blocks), dim3(256), 0, stream, poffs, n_pieces, len, idx, longest, cap);
    return hipGetLastError();
}
hipError_t piece_sort_temp_bytes(uint64_t n, size_t* bytes) {
    uint32_t* k = nullptr;
    size_t b = 0;
    hipError_t e = rocprim::radix_sort_pairs_desc(nullptr, b, k, k, k, k, (size_t)n);
    *bytes = b;
    return e;
}
hipError_t piece_sort(void* temp, size_t temp_bytes, const uint32_t* len_in, uint32_t* len_out, const uint32_t* idx_in, uint32_t* idx_out,
                      uint64_t n, hipStream_t stream) {
    return rocprim::radix_sort_pairs_desc(temp, temp_bytes, len_in, len_out, idx_in, idx_out, (size_t)n, 0, 32, stream);
}

// z of a snippet = sum of its pieces' (log Z adds over a cut), then the reference's check of every snippet's z
// (src/prune.rs:90-96: !z.is_normal() panics)
__global__ __launch_bounds__(256) void piece_z_kernel(const double* __restrict__ zarr, const uint32_t* __restrict__ psnip, uint64_t n_pieces,
                                                      double* __restrict__ zsnip) {
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_pieces; j += (uint64_t)gridDim.x * blockDim.x)
        atomicAdd(&zsnip[psnip[j]], zarr[j]);
}
__global__ __launch_bounds__(256) void snip_z_check_kernel(const double* __restrict__ zsnip, uint64_t n_snips, unsigned long long* __restrict__ err_snip) {
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n_snips; s += (uint64_t)gridDim.x * blockDim.x) {
        const double az = fabs(zsnip[s]);
        if (!(az >= 2.2250738585072014e-308 && az <= 1.7976931348623157e308)) atomicMin(err_snip, (unsigned long long)s);
    }
}
hipError_t launch_piece_z_check(const double* zarr, const uint32_t* psnip, uint64_t n_pieces, double* zsnip, uint64_t n_snips,
                                unsigned long long* err_snip, hipStream_t stream) {
    const uint32_t b1 = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((n_pieces + 255) / 256, 2048));
    const uint32_t b2 = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((n_snips + 255) / 256, 2048));
    hipLaunchKernelGGL(piece_z_kernel, dim3(b1), dim3(256), 0, stream, zarr, psnip, n_pieces, zsnip);
    hipLaunchKernelGGL(snip_z_check_kernel, dim3(b2), dim3(256), 0, stream, zsnip, n_snips, err_snip);
    return hipGetLastError();
}

// the two halves of launch_piece_z_check on their own (estep7.hip: the kernel sums z per snippet itself; the stretches
// its redo list hands to the chained kernels are listed by `order` — every second entry of the pseudo-snippet list)
__global__ __launch_bounds__(256) void piece_z_order_kernel(const double* __restrict__ zarr, const uint32_t* __restrict__ psnip,
                                                            const uint32_t* __restrict__ order, uint64_t n_order, double* __restrict__ zsnip) {
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_order; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t k = order[j];
        atomicAdd(&zsnip[psnip[k]], zarr[k]);
    }
}
hipError_t launch_piece_z_add(const double* zarr, const uint32_t* psnip, const uint32_t* order, uint64_t n_order, double* zsnip, hipStream_t stream) {
    if (n_order == 0) return hipSuccess;
    const uint32_t b = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((n_order + 255) / 256, 2048));
    hipLaunchKernelGGL(piece_z_order_kernel, dim3(b), dim3(256), 0, stream, zarr, psnip, order, n_order, zsnip);
    return hipGetLastError();
}
hipError_t launch_snip_z_check(const double* zsnip, uint64_t n_snips, unsigned long long* err_snip, hipStream_t stream) {
    if (n_snips == 0) return hipSuccess;
    const uint32_t b = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((n_snips + 255) / 256, 2048));
    hipLaunchKernelGGL(snip_z_check_kernel, dim3(b), dim3(256), 0, stream, zsnip, n_snips, err_snip);
    return hipGetLastError();
}

}  // namespace tgx
