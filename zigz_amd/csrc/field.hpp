// BabyBear arithmetic for gfx950 kernels and their host launchers.
//
// The reference stores BabyBear as canonical u64 and multiplies through `u128 %`
// (src/core/field.zig:123-129).  In HBM we keep canonical values packed as u32 (4 B/element).
// Multiplications use a 32-bit Montgomery reduction with ONE operand in Montgomery form:
//   mont_mul(a*R mod p, b) = a*b mod p   (canonical in, canonical out)
// so tables never need converting -- only the per-round scalar (challenge / eq weight) does.
// Field arithmetic is exact, hence results equal the reference's canonical values bit for bit.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define ZK_HD __host__ __device__ __forceinline__
#else
#define ZK_HD inline
#endif

namespace zk {

constexpr uint32_t P = 2013265921u;   // 2^31 - 2^27 + 1, src/core/field_presets.zig:19
constexpr uint32_t MU = 0x88000001u;  // p^-1 mod 2^32
constexpr uint32_t R_MOD_P = 268435454u;    // 2^32 mod p  (Montgomery form of 1)
constexpr uint32_t R2_MOD_P = 1172168163u;  // 2^64 mod p
static_assert((uint32_t)(MU * P) == 1u, "MU must be p^-1 mod 2^32");

// t < p * 2^32  ->  t * 2^-32 mod p, canonical
ZK_HD uint32_t monty_reduce(uint64_t t) {
    uint32_t m = (uint32_t)t * MU;
    uint32_t u = (uint32_t)(((uint64_t)m * P) >> 32);
    uint32_t hi = (uint32_t)(t >> 32);
    uint32_t x = hi - u;
    return hi < u ? x + P : x;
}

// a < 2^32 (Montgomery-form scalar, canonical representative), b < p
ZK_HD uint32_t mont_mul(uint32_t a, uint32_t b) { return monty_reduce((uint64_t)a * b); }

ZK_HD uint32_t add_mod(uint32_t a, uint32_t b) {
    uint32_t s = a + b;  // < 2^32 since a, b < 2^31
    return s >= P ? s - P : s;
}
ZK_HD uint32_t sub_mod(uint32_t a, uint32_t b) { return a >= b ? a - b : a + P - b; }

// canonical -> Montgomery form (a * 2^32 mod p)
ZK_HD uint32_t to_mont(uint32_t a) { return mont_mul(R2_MOD_P, a); }

// (1-r)*a0 + r*a1 == a0 + r*(a1 - a0)   (src/poly/multilinear.zig:166-173), r_m = to_mont(r)
ZK_HD uint32_t bind1(uint32_t a0, uint32_t a1, uint32_t r_m) {
    return add_mod(a0, mont_mul(r_m, sub_mod(a1, a0)));
}

}  // namespace zk
