// Device-side helpers shared by the encode and E-step kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tgx {

__device__ __forceinline__ uint32_t readlane_u32(uint32_t v, uint32_t lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane);
}
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, uint32_t lane) {
    uint32_t lo = readlane_u32((uint32_t)v, lane);
    uint32_t hi = readlane_u32((uint32_t)(v >> 32), lane);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ double readlane_f64(double v, uint32_t lane) {
    return __longlong_as_double((long long)readlane_u64((uint64_t)__double_as_longlong(v), lane));
}
__device__ __forceinline__ uint64_t first_u64(uint64_t v) {
    uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t rotl64(uint64_t x, uint32_t r) {
    r &= 63u;
    return (x << r) | (x >> ((64u - r) & 63u));
}

// same function as tgx_dropout_u01 in include/tgx.h
__device__ __forceinline__ double dropout_u01(uint64_t seed, uint64_t sample, uint64_t pos, uint32_t len) {
    uint64_t x = seed ^ (sample * 0x9E3779B97F4A7C15ULL) ^ (pos * 0xC2B2AE3D27D4EB4FULL) ^
                 ((uint64_t)len * 0x165667B19E3779F9ULL);
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBULL;
    x ^= x >> 31;
    return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}

// ---- LDS by byte offset, buffer resources (encode5.hip, estep7.hip) ----
// LDS accesses by byte offset (32-bit LDS pointers): no 64-bit flat address arithmetic, no "+ base of the dynamic
// LDS" per access — the kernel folds that base (0 when a kernel has no static LDS) into its per-lane constants once
template <class T>
__device__ __forceinline__ T lds_ld(uint32_t off) { return *(const __attribute__((address_space(3))) T*)(uintptr_t)off; }
template <class T>
__device__ __forceinline__ void lds_st(uint32_t off, T v) { *(__attribute__((address_space(3))) T*)(uintptr_t)off = v; }

// The trie records and the value table are read through buffer resources: `buffer_load_dwordx2 v, voffset, rsrc, 0
// offen` takes a 32-bit byte offset per lane (a global_load of base + offset needs a 64-bit add per lane, or a form
// the compiler only picks for scaled indices), and an offset beyond the table reads zeros instead of faulting.
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);  // raw buffer, 32-bit data format
}
__device__ __forceinline__ uint2 buf_ld8(__amdgpu_buffer_rsrc_t rs, uint32_t off) {
    const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, 0);
    return make_uint2(v.x, v.y);
}
__device__ __forceinline__ double buf_ld_f64(__amdgpu_buffer_rsrc_t rs, uint32_t off) {
    const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, 0);
    return __hiloint2double((int)v.y, (int)v.x);
}


// ---- LDS layout (per wave) ----------------------------------------------------
// sc[row u = start position in the 64-block][col = len - 1] : f64 score
// hl[same]                                                    : slot << 6 | (len - 1)
// Row stride = LM entries; FRONT entries of padding in front and 64 behind, because
// lanes that take no part in a relaxation step still issue their (ignored) read.
constexpr uint32_t kFront = 2;
__host__ __device__ inline uint32_t wave_lds_entries(uint32_t lm) { return 64u * lm + 64u + kFront; }
__host__ __device__ inline uint32_t wave_lds_bytes(uint32_t lm) {
    return wave_lds_entries(lm) * 12u + 128u;  // + 128 B text staging (generic path)
}

// D = mask[lane] ? t : f, with the 64-bit lane mask in an SGPR pair (no per-lane bit test)
__device__ __forceinline__ uint32_t sel_u32(uint64_t mask, uint32_t t, uint32_t f) {
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(f), "v"(t), "s"(mask));
    return r;
}
template <int IMM>  // mask ? IMM : f, IMM an inline constant (0..64)
__device__ __forceinline__ uint32_t sel_imm_u32(uint64_t mask, uint32_t f) {
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(f), "n"(IMM), "s"(mask));
    return r;
}
__device__ __forceinline__ double sel_f64(uint64_t mask, double t, double f) {
    const uint64_t tb = (uint64_t)__double_as_longlong(t), fb = (uint64_t)__double_as_longlong(f);
    const uint32_t lo = sel_u32(mask, (uint32_t)tb, (uint32_t)fb);
    const uint32_t hi = sel_u32(mask, (uint32_t)(tb >> 32), (uint32_t)(fb >> 32));
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}


// ---- 16-lane rows (four samples per wave) ----------------------------------------------
template <int CTRL>
__device__ __forceinline__ uint32_t row_bcast_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}
// One v_mov_b64 with DPP (gfx90a+ allow row_newbcast on the 64-bit move) instead of two
// 32-bit DPP moves plus the two moves that initialise their destinations.  The s_nop covers
// the "VALU write -> DPP read" hazard, which the compiler does not track through inline asm.
template <int U>
__device__ __forceinline__ double row_bcast_f64(double v) {
    double r;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf"
                 : "=v"(r) : "v"(v), "n"(U));
    return r;
}

constexpr uint64_t kRowLane0 = 0x0001000100010001ULL;  // lane 0 of each 16-lane row

// Work queues.  One lane adds to a global counter and the whole wave reads the result.  Written as
// `if (lane == 0) got = atomicAdd(..)` followed by readfirstlane, the compiler's control-flow
// structurizer has twice produced loops that re-use a stale value instead of repeating the atomic
// (the wave then processes unit 0 forever), so the EXEC mask is narrowed by hand and the compiler sees
// one opaque instruction.  Lane 0 of a wave is always active here (full waves, uniform control flow).
__device__ __forceinline__ uint64_t wave_fetch_add(unsigned long long* counter, uint32_t n) {
    uint64_t got, saved;
    const uint64_t inc = n;
    const uint32_t zero = 0;
    asm volatile(
        "s_mov_b64 %1, exec\n\t"
        "s_mov_b64 exec, 1\n\t"
        "global_atomic_add_x2 %0, %2, %3, %4 sc0\n\t"
        "s_mov_b64 exec, %1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(got), "=&s"(saved) : "v"(zero), "v"(inc), "s"(counter) : "memory");
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)got);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(got >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// Work distribution of the four-rows-per-wave kernels: rows that finished their unit claim the next
// index of the longest-first order — one atomic per wave and trip.  Rows therefore run out of work
// together whatever the mix of lengths (a static split leaves the rows that drew the long units
// running alone at the end).  Returns the claimed index for rows with need_new set, ~0 for the others.
__device__ __forceinline__ uint64_t claim_rows(unsigned long long* queue, bool need_new, uint32_t r) {
    const uint64_t want = __builtin_amdgcn_ballot_w64(need_new) & kRowLane0;  // bit 16 r: row r needs one
    uint64_t k = ~0ull;
    if (want != 0) {  // wave-uniform
        const uint64_t base = wave_fetch_add(queue, (uint32_t)__builtin_popcountll(want));
        if (need_new) k = base + (uint64_t)__builtin_popcountll(want & ((1ull << (r * 16u)) - 1ull));
    }
    return k;
}

// The same with `chunk` consecutive indices per claiming row: returns the first, the row hands out the rest itself
__device__ __forceinline__ uint64_t claim_rows_chunk(unsigned long long* queue, bool need_new, uint32_t r, uint32_t chunk) {
    const uint64_t want = __builtin_amdgcn_ballot_w64(need_new) & kRowLane0;
    uint64_t k = ~0ull;
    if (want != 0) {  // wave-uniform
        const uint64_t base = wave_fetch_add(queue, (uint32_t)__builtin_popcountll(want) * chunk);
        if (need_new) k = base + (uint64_t)__builtin_popcountll(want & ((1ull << (r * 16u)) - 1ull)) * chunk;
    }
    return k;
}

// One 16-byte trie record in ONE load: the empty asm makes all four words live at
// once (otherwise the compiler splits the load into three dependent round trips).
__device__ __forceinline__ uint4 load_rec(const uint4* __restrict__ trie, uint32_t t) {
    uint4 r = trie[t];
    asm volatile("" : "+v"(r.x), "+v"(r.y), "+v"(r.z), "+v"(r.w));
    return r;
}

// rows4 back-pointer bytes live in a per-sample region of the scratch row that starts at bp8_base(beg, s)
// and are permuted inside every group of 64 positions so that the four bytes one lane produces in four
// consecutive trips are adjacent: a row of 16 lanes then writes one full 64-byte segment per group instead
// of four 16-byte pieces (HBM write traffic of encode4_kernel 3.99 -> ~1.1 GB per GiB of text).
__device__ __forceinline__ uint64_t bp8_base(uint64_t beg, uint32_t s) { return (beg + 128ull * s) & ~3ull; }
__device__ __forceinline__ uint32_t bp8_perm(uint32_t j) { return (j & ~63u) | ((j & 15u) << 2) | ((j >> 4) & 3u); }


// ---- relaxation step of the four-rows-per-wave encode kernels (encode4_kernel, encode5_kernel) ----
constexpr uint32_t kNoStep = 0xFFu;  // "nothing pushed into this accumulator yet"

// The winner is remembered as the step U that pushed it; with the lane's own index that
// gives the token length ((l - U - 1) & 15) + 1, the only thing the trace needs: the
// token id is looked up from the token's bytes (hash table), so no trie handle travels
// through LDS and the match buffer is 8 bytes per (position, length).
//
// Step U finalises position p0 + U (lane U of each row) and starts position p0 + U + 16 in
// the same lane.  The final (winner step, high word of the score) pair of that lane is kept
// in `fin` / `fhi`, and the lane is restarted by FORCING it to take this step's candidate
// (the 16-byte token starting at p0 + U, or -inf) instead of resetting it first.  A forced
// take of -inf leaves a meaningless winner step behind; the position is "not reached" iff its
// final score is -inf, which is what the kept high word tells.  (Parking the pair in LDS with
// an EXEC-masked store instead of two v_cndmask was tried: the LDS pipe is per CU, the VALUs
// per SIMD, and the kernel got slower.)
template <int U>
__device__ __forceinline__ void relax4_step(double sv, double& acc, uint32_t& bpv, uint32_t& fin, uint32_t& fhi) {
    constexpr uint64_t MU = kRowLane0 << U;  // lanes with l == U
    const double best = row_bcast_f64<U>(acc);
    fin = sel_u32(MU, bpv, fin);
    fhi = sel_u32(MU, (uint32_t)((uint64_t)__double_as_longlong(acc) >> 32), fhi);
    const double cand = best + sv;           // model.rs:98
    const uint64_t take = __builtin_amdgcn_fcmp(cand, acc, 2 /* OGT: model.rs:101 */) | MU;
    acc = sel_f64(take, cand, acc);
    bpv = sel_imm_u32<U>(take, bpv);
}

// The same step with a shorter dependent chain (encode5_kernel, encode6_kernel).  The lane that was just
// finalised is reset to -inf before the new candidate meets it, so the value path is one v_max_f64 — the
// maximum of two finite-or--inf values is the value the strict '>' selection keeps (model.rs:101) — and the
// comparison only steers the winner bookkeeping, off the chain that the next step waits for:
// broadcast -> add -> max instead of broadcast -> add -> compare -> mask OR -> two selects.  Same count of
// VALU instructions, about a third of the latency per position (a lone 64 KiB sample: 7.1 -> ms per pass).
// (Round 4 removed two of the step's nine vector instructions as an experiment — no `fhi`, the reset by the high word
// alone with a finite sentinel: 150 of the kernel's 1 252 static VALU instructions gone, ids still bit-exact, 12.60 ->
// 12.66 ms per GiB.  The vector ALU is 79 % busy but it is not what the pass waits for: profiles/r04.)
template <int U>
__device__ __forceinline__ void relax5_step(double sv, double& acc, uint32_t& bpv, uint32_t& fin, uint32_t& fhi) {
    constexpr uint64_t MU = kRowLane0 << U;  // lanes with l == U
    const double best = row_bcast_f64<U>(acc);
    fin = sel_u32(MU, bpv, fin);
    fhi = sel_u32(MU, (uint32_t)((uint64_t)__double_as_longlong(acc) >> 32), fhi);
    const double acc_r = sel_f64(MU, -__builtin_huge_val(), acc);  // the finalised lane starts position + 16
    const double cand = best + sv;           // model.rs:98
    const uint64_t take = __builtin_amdgcn_fcmp(cand, acc_r, 2 /* OGT: model.rs:101 */);
    asm("v_max_f64 %0, %1, %2" : "=v"(acc) : "v"(acc_r), "v"(cand));  // (fmax() adds a canonicalising v_max of its own)
    bpv = sel_imm_u32<U>(take, bpv);
}


// same function as tgx::tok_hash64 (trie_build.h)
__device__ __forceinline__ uint32_t rotl32_dev(uint32_t x, int r) { return __builtin_rotateleft32(x, (uint32_t)r); }
__device__ __forceinline__ uint64_t tok_hash64_dev(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t len, uint32_t seed) {
    uint32_t a = (w0 ^ (len << 27) ^ seed) * 0x85EBCA6Bu;
    a ^= a >> 15;
    uint32_t b = a;
    a = (a + w1) * 0xC2B2AE35u;
    a ^= a >> 13;
    b = rotl32_dev(b, 11) ^ a;
    a = (a + w2) * 0x27D4EB2Fu;
    a ^= a >> 16;
    b = rotl32_dev(b, 11) ^ a;
    a = (a + w3) * 0x165667B1u;
    a ^= a >> 15;
    b = rotl32_dev(b, 11) + (w0 ^ rotl32_dev(w1, 8) ^ rotl32_dev(w2, 16) ^ rotl32_dev(w3, 24));
    return ((uint64_t)b << 32) | a;
}


// same function as tgx::tok_hash64_long (trie_build.h): tok_hash64 continued over four more dwords
__device__ __forceinline__ uint64_t tok_hash64_long_dev(const uint32_t* w, uint32_t len, uint32_t seed) {
    uint32_t a = (w[0] ^ (len << 27) ^ seed) * 0x85EBCA6Bu;
    a ^= a >> 15;
    uint32_t b = a;
    a = (a + w[1]) * 0xC2B2AE35u;
    a ^= a >> 13;
    b = rotl32_dev(b, 11) ^ a;
    a = (a + w[2]) * 0x27D4EB2Fu;
    a ^= a >> 16;
    b = rotl32_dev(b, 11) ^ a;
    a = (a + w[3]) * 0x165667B1u;
    a ^= a >> 15;
    b = rotl32_dev(b, 11) + (w[0] ^ rotl32_dev(w[1], 8) ^ rotl32_dev(w[2], 16) ^ rotl32_dev(w[3], 24));
    if (len > 16u) {
        a = (a + w[4]) * 0x85EBCA6Bu;
        a ^= a >> 15;
        b = rotl32_dev(b, 11) ^ a;
        a = (a + w[5]) * 0xC2B2AE35u;
        a ^= a >> 13;
        b = rotl32_dev(b, 11) ^ a;
        a = (a + w[6]) * 0x27D4EB2Fu;
        a ^= a >> 16;
        b = rotl32_dev(b, 11) ^ a;
        a = (a + w[7]) * 0x165667B1u;
        a ^= a >> 15;
        b = rotl32_dev(b, 11) + (w[4] ^ rotl32_dev(w[5], 8) ^ rotl32_dev(w[6], 16) ^ rotl32_dev(w[7], 24));
    }
    return ((uint64_t)b << 32) | a;
}


// ---- linear-domain E-step (estep4l.hip, encode5.hip: estep5_fwd_kernel) ----
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


}  // namespace tgx
