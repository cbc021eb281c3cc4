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
template <int U>
__device__ __forceinline__ double row_bcast_f64(double v) {
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = row_bcast_u32<0x150 + U>((uint32_t)b);
    const uint32_t hi = row_bcast_u32<0x150 + U>((uint32_t)(b >> 32));
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

constexpr uint64_t kRowLane0 = 0x0001000100010001ULL;  // lane 0 of each 16-lane row

// One 16-byte trie record in ONE load: the empty asm makes all four words live at
// once (otherwise the compiler splits the load into three dependent round trips).
__device__ __forceinline__ uint4 load_rec(const uint4* __restrict__ trie, uint32_t t) {
    uint4 r = trie[t];
    asm volatile("" : "+v"(r.x), "+v"(r.y), "+v"(r.z), "+v"(r.w));
    return r;
}

}  // namespace tgx
