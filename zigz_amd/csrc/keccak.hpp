// Keccak-f[1600] / SHA3-256 single-block hashing for gfx950: one thread = one hash, the 25-lane
// state lives in 50 VGPRs.  Both hashes on the Merkle path fit one rate block (136 B), i.e. exactly
// one permutation each:
//   leaf  = SHA3-256(LE64(value))        src/core/hash.zig:135-147  (hashFieldElementSHA3)
//   node  = SHA3-256(left || right)      src/core/hash.zig:187-195  (mergeHashesSHA3)
// This work is integer-ALU bound (about 4.4k 32-bit VALU ops per permutation vs 40-96 B of traffic).
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#else  // host-only unit test build of the same code (tests/test_host_units.py)
#define __device__
#define __constant__
#define __forceinline__ inline
#endif

namespace zk {

__constant__ const uint64_t KECCAK_RC[24] = {
    0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull,
    0x000000000000808bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
    0x000000000000008aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000aull,
    0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull,
    0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
    0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};

__device__ __forceinline__ uint64_t rotl64(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }

// One round, lanes indexed a[x + 5*y].
#define ZK_KECCAK_ROUND(a, rc)                                                                      \
    do {                                                                                            \
        uint64_t c0 = a[0] ^ a[5] ^ a[10] ^ a[15] ^ a[20];                                          \
        uint64_t c1 = a[1] ^ a[6] ^ a[11] ^ a[16] ^ a[21];                                          \
        uint64_t c2 = a[2] ^ a[7] ^ a[12] ^ a[17] ^ a[22];                                          \
        uint64_t c3 = a[3] ^ a[8] ^ a[13] ^ a[18] ^ a[23];                                          \
        uint64_t c4 = a[4] ^ a[9] ^ a[14] ^ a[19] ^ a[24];                                          \
        uint64_t d0 = c4 ^ rotl64(c1, 1), d1 = c0 ^ rotl64(c2, 1), d2 = c1 ^ rotl64(c3, 1);         \
        uint64_t d3 = c2 ^ rotl64(c4, 1), d4 = c3 ^ rotl64(c0, 1);                                  \
        /* theta + rho + pi: b[y + 5*((2x+3y)%5)] = rot(a[x+5y] ^ d[x], r[x][y]) */                 \
        uint64_t b00 = a[0] ^ d0;                                                                   \
        uint64_t b10 = rotl64(a[1] ^ d1, 1), b20 = rotl64(a[2] ^ d2, 62);                           \
        uint64_t b05 = rotl64(a[3] ^ d3, 28), b15 = rotl64(a[4] ^ d4, 27);                          \
        uint64_t b16 = rotl64(a[5] ^ d0, 36), b01 = rotl64(a[6] ^ d1, 44);                          \
        uint64_t b11 = rotl64(a[7] ^ d2, 6), b21 = rotl64(a[8] ^ d3, 55);                           \
        uint64_t b06 = rotl64(a[9] ^ d4, 20), b07 = rotl64(a[10] ^ d0, 3);                          \
        uint64_t b17 = rotl64(a[11] ^ d1, 10), b02 = rotl64(a[12] ^ d2, 43);                        \
        uint64_t b12 = rotl64(a[13] ^ d3, 25), b22 = rotl64(a[14] ^ d4, 39);                        \
        uint64_t b23 = rotl64(a[15] ^ d0, 41), b08 = rotl64(a[16] ^ d1, 45);                        \
        uint64_t b18 = rotl64(a[17] ^ d2, 15), b03 = rotl64(a[18] ^ d3, 21);                        \
        uint64_t b13 = rotl64(a[19] ^ d4, 8), b14 = rotl64(a[20] ^ d0, 18);                         \
        uint64_t b24 = rotl64(a[21] ^ d1, 2), b09 = rotl64(a[22] ^ d2, 61);                         \
        uint64_t b19 = rotl64(a[23] ^ d3, 56), b04 = rotl64(a[24] ^ d4, 14);                        \
        /* chi (+ iota on lane 0); bXY = b[index XY] after pi */                                   \
        a[0] = b00 ^ (~b01 & b02) ^ (rc);                                                           \
        a[1] = b01 ^ (~b02 & b03); a[2] = b02 ^ (~b03 & b04);                                       \
        a[3] = b03 ^ (~b04 & b00); a[4] = b04 ^ (~b00 & b01);                                       \
        a[5] = b05 ^ (~b06 & b07); a[6] = b06 ^ (~b07 & b08); a[7] = b07 ^ (~b08 & b09);            \
        a[8] = b08 ^ (~b09 & b05); a[9] = b09 ^ (~b05 & b06);                                       \
        a[10] = b10 ^ (~b11 & b12); a[11] = b11 ^ (~b12 & b13); a[12] = b12 ^ (~b13 & b14);         \
        a[13] = b13 ^ (~b14 & b10); a[14] = b14 ^ (~b10 & b11);                                     \
        a[15] = b15 ^ (~b16 & b17); a[16] = b16 ^ (~b17 & b18); a[17] = b17 ^ (~b18 & b19);         \
        a[18] = b18 ^ (~b19 & b15); a[19] = b19 ^ (~b15 & b16);                                     \
        a[20] = b20 ^ (~b21 & b22); a[21] = b21 ^ (~b22 & b23); a[22] = b22 ^ (~b23 & b24);         \
        a[23] = b23 ^ (~b24 & b20); a[24] = b24 ^ (~b20 & b21);                                     \
    } while (0)

#ifndef ZK_KECCAK_UNROLL
#define ZK_KECCAK_UNROLL 24
#endif

__device__ __forceinline__ void keccak_f1600(uint64_t a[25]) {
#pragma unroll ZK_KECCAK_UNROLL
    for (int r = 0; r < 24; r++) ZK_KECCAK_ROUND(a, KECCAK_RC[r]);
}

struct Digest {
    uint64_t w[4];
};

// SHA3-256 of the 8 LE bytes of a canonical field element (pad: 0x06 at byte 8, 0x80 at byte 135)
__device__ __forceinline__ Digest sha3_leaf(uint64_t value) {
    uint64_t a[25];
#pragma unroll
    for (int i = 0; i < 25; i++) a[i] = 0;
    a[0] = value;
    a[1] = 0x06ull;
    a[16] = 0x8000000000000000ull;
    keccak_f1600(a);
    return Digest{{a[0], a[1], a[2], a[3]}};
}

// SHA3-256 of left || right (64 bytes; pad: 0x06 at byte 64, 0x80 at byte 135)
__device__ __forceinline__ Digest sha3_node(const Digest &l, const Digest &r) {
    uint64_t a[25];
#pragma unroll
    for (int i = 0; i < 25; i++) a[i] = 0;
    a[0] = l.w[0]; a[1] = l.w[1]; a[2] = l.w[2]; a[3] = l.w[3];
    a[4] = r.w[0]; a[5] = r.w[1]; a[6] = r.w[2]; a[7] = r.w[3];
    a[8] = 0x06ull;
    a[16] = 0x8000000000000000ull;
    keccak_f1600(a);
    return Digest{{a[0], a[1], a[2], a[3]}};
}

}  // namespace zk
