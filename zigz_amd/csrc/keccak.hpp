// Keccak-f[1600] / SHA3-256 single-block hashing: one thread = one hash.  Both hashes on the Merkle path
// fit one rate block (136 B), i.e. exactly one permutation each:
//   leaf  = SHA3-256(LE64(value))        src/core/hash.zig:135-147  (hashFieldElementSHA3)
//   node  = SHA3-256(left || right)      src/core/hash.zig:187-195  (mergeHashesSHA3)
// This work is integer-ALU bound.  The device permutation keeps the 25 lanes as 50 32-bit VGPRs and is
// written for the gfx950 VALU: 3-input boolean ops through v_bitop3_b32 (XOR3 = 0x96, chi a^(~b&c) = 0xD2)
// and 64-bit rotations as two v_alignbit_b32 -- 180 VALU ops per round, 4.3 k per permutation (hipcc's
// own lowering of the 64-bit formulation needs 6.5 k: 2-input XORs, v_bfi+xor for chi, 64-bit shifts).
// The same source builds on the host (plain C fallbacks for the three primitives) for unit tests, and
// the 64-bit macro formulation below is what the host-side sponge uses.
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#else  // host-only build of the same code (host sponge, unit tests)
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

// ---- 64-bit formulation (host sponge) ----
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


// ---- 32-bit formulation (device kernels) ----
#if defined(__HIP_DEVICE_COMPILE__)
#define ZK_X3(a, b, c) __builtin_amdgcn_bitop3_b32((a), (b), (c), 0x96)
#define ZK_CHI(a, b, c) __builtin_amdgcn_bitop3_b32((a), (b), (c), 0xD2)
#define ZK_ALIGN(hi, lo, s) __builtin_amdgcn_alignbit((hi), (lo), (s))
#else
#define ZK_X3(a, b, c) ((a) ^ (b) ^ (c))
#define ZK_CHI(a, b, c) ((a) ^ (~(b) & (c)))
#define ZK_ALIGN(hi, lo, s) ((uint32_t)(((((uint64_t)(hi)) << 32) | (uint64_t)(lo)) >> (s)))
#endif

#ifndef ZK_KECCAK_UNROLL
#define ZK_KECCAK_UNROLL 24
#endif

// l[i], h[i]: low / high 32 bits of lane i = x + 5y
__device__ __forceinline__ void keccak_f1600_32(uint32_t l[25], uint32_t h[25]) {
#pragma unroll ZK_KECCAK_UNROLL
    for (int r = 0; r < 24; r++) {
        uint32_t bl[25], bh[25];
// theta: column parities (two 3-input XORs per half)
        const uint32_t cl0 = ZK_X3(ZK_X3(l[0], l[5], l[10]), l[15], l[20]);
        const uint32_t ch0 = ZK_X3(ZK_X3(h[0], h[5], h[10]), h[15], h[20]);
        const uint32_t cl1 = ZK_X3(ZK_X3(l[1], l[6], l[11]), l[16], l[21]);
        const uint32_t ch1 = ZK_X3(ZK_X3(h[1], h[6], h[11]), h[16], h[21]);
        const uint32_t cl2 = ZK_X3(ZK_X3(l[2], l[7], l[12]), l[17], l[22]);
        const uint32_t ch2 = ZK_X3(ZK_X3(h[2], h[7], h[12]), h[17], h[22]);
        const uint32_t cl3 = ZK_X3(ZK_X3(l[3], l[8], l[13]), l[18], l[23]);
        const uint32_t ch3 = ZK_X3(ZK_X3(h[3], h[8], h[13]), h[18], h[23]);
        const uint32_t cl4 = ZK_X3(ZK_X3(l[4], l[9], l[14]), l[19], l[24]);
        const uint32_t ch4 = ZK_X3(ZK_X3(h[4], h[9], h[14]), h[19], h[24]);
        // rot(C[x], 1)
        const uint32_t rl0 = ZK_ALIGN(cl0, ch0, 31), rh0 = ZK_ALIGN(ch0, cl0, 31);
        const uint32_t rl1 = ZK_ALIGN(cl1, ch1, 31), rh1 = ZK_ALIGN(ch1, cl1, 31);
        const uint32_t rl2 = ZK_ALIGN(cl2, ch2, 31), rh2 = ZK_ALIGN(ch2, cl2, 31);
        const uint32_t rl3 = ZK_ALIGN(cl3, ch3, 31), rh3 = ZK_ALIGN(ch3, cl3, 31);
        const uint32_t rl4 = ZK_ALIGN(cl4, ch4, 31), rh4 = ZK_ALIGN(ch4, cl4, 31);
        // theta apply (A ^ C[x-1] ^ rot(C[x+1],1)) fused with rho (rotate) and pi (destination index)
        { const uint32_t tl = ZK_X3(l[0], cl4, rl1), th = ZK_X3(h[0], ch4, rh1);
          bl[0] = tl; bh[0] = th; }
        { const uint32_t tl = ZK_X3(l[1], cl0, rl2), th = ZK_X3(h[1], ch0, rh2);
          bh[10] = ZK_ALIGN(th, tl, 31); bl[10] = ZK_ALIGN(tl, th, 31); }
        { const uint32_t tl = ZK_X3(l[2], cl1, rl3), th = ZK_X3(h[2], ch1, rh3);
          bh[20] = ZK_ALIGN(tl, th, 2); bl[20] = ZK_ALIGN(th, tl, 2); }
        { const uint32_t tl = ZK_X3(l[3], cl2, rl4), th = ZK_X3(h[3], ch2, rh4);
          bh[5] = ZK_ALIGN(th, tl, 4); bl[5] = ZK_ALIGN(tl, th, 4); }
        { const uint32_t tl = ZK_X3(l[4], cl3, rl0), th = ZK_X3(h[4], ch3, rh0);
          bh[15] = ZK_ALIGN(th, tl, 5); bl[15] = ZK_ALIGN(tl, th, 5); }
        { const uint32_t tl = ZK_X3(l[5], cl4, rl1), th = ZK_X3(h[5], ch4, rh1);
          bh[16] = ZK_ALIGN(tl, th, 28); bl[16] = ZK_ALIGN(th, tl, 28); }
        { const uint32_t tl = ZK_X3(l[6], cl0, rl2), th = ZK_X3(h[6], ch0, rh2);
          bh[1] = ZK_ALIGN(tl, th, 20); bl[1] = ZK_ALIGN(th, tl, 20); }
        { const uint32_t tl = ZK_X3(l[7], cl1, rl3), th = ZK_X3(h[7], ch1, rh3);
          bh[11] = ZK_ALIGN(th, tl, 26); bl[11] = ZK_ALIGN(tl, th, 26); }
        { const uint32_t tl = ZK_X3(l[8], cl2, rl4), th = ZK_X3(h[8], ch2, rh4);
          bh[21] = ZK_ALIGN(tl, th, 9); bl[21] = ZK_ALIGN(th, tl, 9); }
        { const uint32_t tl = ZK_X3(l[9], cl3, rl0), th = ZK_X3(h[9], ch3, rh0);
          bh[6] = ZK_ALIGN(th, tl, 12); bl[6] = ZK_ALIGN(tl, th, 12); }
        { const uint32_t tl = ZK_X3(l[10], cl4, rl1), th = ZK_X3(h[10], ch4, rh1);
          bh[7] = ZK_ALIGN(th, tl, 29); bl[7] = ZK_ALIGN(tl, th, 29); }
        { const uint32_t tl = ZK_X3(l[11], cl0, rl2), th = ZK_X3(h[11], ch0, rh2);
          bh[17] = ZK_ALIGN(th, tl, 22); bl[17] = ZK_ALIGN(tl, th, 22); }
        { const uint32_t tl = ZK_X3(l[12], cl1, rl3), th = ZK_X3(h[12], ch1, rh3);
          bh[2] = ZK_ALIGN(tl, th, 21); bl[2] = ZK_ALIGN(th, tl, 21); }
        { const uint32_t tl = ZK_X3(l[13], cl2, rl4), th = ZK_X3(h[13], ch2, rh4);
          bh[12] = ZK_ALIGN(th, tl, 7); bl[12] = ZK_ALIGN(tl, th, 7); }
        { const uint32_t tl = ZK_X3(l[14], cl3, rl0), th = ZK_X3(h[14], ch3, rh0);
          bh[22] = ZK_ALIGN(tl, th, 25); bl[22] = ZK_ALIGN(th, tl, 25); }
        { const uint32_t tl = ZK_X3(l[15], cl4, rl1), th = ZK_X3(h[15], ch4, rh1);
          bh[23] = ZK_ALIGN(tl, th, 23); bl[23] = ZK_ALIGN(th, tl, 23); }
        { const uint32_t tl = ZK_X3(l[16], cl0, rl2), th = ZK_X3(h[16], ch0, rh2);
          bh[8] = ZK_ALIGN(tl, th, 19); bl[8] = ZK_ALIGN(th, tl, 19); }
        { const uint32_t tl = ZK_X3(l[17], cl1, rl3), th = ZK_X3(h[17], ch1, rh3);
          bh[18] = ZK_ALIGN(th, tl, 17); bl[18] = ZK_ALIGN(tl, th, 17); }
        { const uint32_t tl = ZK_X3(l[18], cl2, rl4), th = ZK_X3(h[18], ch2, rh4);
          bh[3] = ZK_ALIGN(th, tl, 11); bl[3] = ZK_ALIGN(tl, th, 11); }
        { const uint32_t tl = ZK_X3(l[19], cl3, rl0), th = ZK_X3(h[19], ch3, rh0);
          bh[13] = ZK_ALIGN(th, tl, 24); bl[13] = ZK_ALIGN(tl, th, 24); }
        { const uint32_t tl = ZK_X3(l[20], cl4, rl1), th = ZK_X3(h[20], ch4, rh1);
          bh[14] = ZK_ALIGN(th, tl, 14); bl[14] = ZK_ALIGN(tl, th, 14); }
        { const uint32_t tl = ZK_X3(l[21], cl0, rl2), th = ZK_X3(h[21], ch0, rh2);
          bh[24] = ZK_ALIGN(th, tl, 30); bl[24] = ZK_ALIGN(tl, th, 30); }
        { const uint32_t tl = ZK_X3(l[22], cl1, rl3), th = ZK_X3(h[22], ch1, rh3);
          bh[9] = ZK_ALIGN(tl, th, 3); bl[9] = ZK_ALIGN(th, tl, 3); }
        { const uint32_t tl = ZK_X3(l[23], cl2, rl4), th = ZK_X3(h[23], ch2, rh4);
          bh[19] = ZK_ALIGN(tl, th, 8); bl[19] = ZK_ALIGN(th, tl, 8); }
        { const uint32_t tl = ZK_X3(l[24], cl3, rl0), th = ZK_X3(h[24], ch3, rh0);
          bh[4] = ZK_ALIGN(th, tl, 18); bl[4] = ZK_ALIGN(tl, th, 18); }
        // chi
        l[0] = ZK_CHI(bl[0], bl[1], bl[2]); h[0] = ZK_CHI(bh[0], bh[1], bh[2]);
        l[1] = ZK_CHI(bl[1], bl[2], bl[3]); h[1] = ZK_CHI(bh[1], bh[2], bh[3]);
        l[2] = ZK_CHI(bl[2], bl[3], bl[4]); h[2] = ZK_CHI(bh[2], bh[3], bh[4]);
        l[3] = ZK_CHI(bl[3], bl[4], bl[0]); h[3] = ZK_CHI(bh[3], bh[4], bh[0]);
        l[4] = ZK_CHI(bl[4], bl[0], bl[1]); h[4] = ZK_CHI(bh[4], bh[0], bh[1]);
        l[5] = ZK_CHI(bl[5], bl[6], bl[7]); h[5] = ZK_CHI(bh[5], bh[6], bh[7]);
        l[6] = ZK_CHI(bl[6], bl[7], bl[8]); h[6] = ZK_CHI(bh[6], bh[7], bh[8]);
        l[7] = ZK_CHI(bl[7], bl[8], bl[9]); h[7] = ZK_CHI(bh[7], bh[8], bh[9]);
        l[8] = ZK_CHI(bl[8], bl[9], bl[5]); h[8] = ZK_CHI(bh[8], bh[9], bh[5]);
        l[9] = ZK_CHI(bl[9], bl[5], bl[6]); h[9] = ZK_CHI(bh[9], bh[5], bh[6]);
        l[10] = ZK_CHI(bl[10], bl[11], bl[12]); h[10] = ZK_CHI(bh[10], bh[11], bh[12]);
        l[11] = ZK_CHI(bl[11], bl[12], bl[13]); h[11] = ZK_CHI(bh[11], bh[12], bh[13]);
        l[12] = ZK_CHI(bl[12], bl[13], bl[14]); h[12] = ZK_CHI(bh[12], bh[13], bh[14]);
        l[13] = ZK_CHI(bl[13], bl[14], bl[10]); h[13] = ZK_CHI(bh[13], bh[14], bh[10]);
        l[14] = ZK_CHI(bl[14], bl[10], bl[11]); h[14] = ZK_CHI(bh[14], bh[10], bh[11]);
        l[15] = ZK_CHI(bl[15], bl[16], bl[17]); h[15] = ZK_CHI(bh[15], bh[16], bh[17]);
        l[16] = ZK_CHI(bl[16], bl[17], bl[18]); h[16] = ZK_CHI(bh[16], bh[17], bh[18]);
        l[17] = ZK_CHI(bl[17], bl[18], bl[19]); h[17] = ZK_CHI(bh[17], bh[18], bh[19]);
        l[18] = ZK_CHI(bl[18], bl[19], bl[15]); h[18] = ZK_CHI(bh[18], bh[19], bh[15]);
        l[19] = ZK_CHI(bl[19], bl[15], bl[16]); h[19] = ZK_CHI(bh[19], bh[15], bh[16]);
        l[20] = ZK_CHI(bl[20], bl[21], bl[22]); h[20] = ZK_CHI(bh[20], bh[21], bh[22]);
        l[21] = ZK_CHI(bl[21], bl[22], bl[23]); h[21] = ZK_CHI(bh[21], bh[22], bh[23]);
        l[22] = ZK_CHI(bl[22], bl[23], bl[24]); h[22] = ZK_CHI(bh[22], bh[23], bh[24]);
        l[23] = ZK_CHI(bl[23], bl[24], bl[20]); h[23] = ZK_CHI(bh[23], bh[24], bh[20]);
        l[24] = ZK_CHI(bl[24], bl[20], bl[21]); h[24] = ZK_CHI(bh[24], bh[20], bh[21]);
        l[0] ^= (uint32_t)KECCAK_RC[r];
        h[0] ^= (uint32_t)(KECCAK_RC[r] >> 32);
    }
}

struct Digest {
    uint64_t w[4];
};

__device__ __forceinline__ Digest digest_of(const uint32_t l[25], const uint32_t h[25]) {
    return Digest{{((uint64_t)h[0] << 32) | l[0], ((uint64_t)h[1] << 32) | l[1], ((uint64_t)h[2] << 32) | l[2],
                   ((uint64_t)h[3] << 32) | l[3]}};
}

// SHA3-256 of the 8 LE bytes of a canonical field element (pad: 0x06 at byte 8, 0x80 at byte 135)
__device__ __forceinline__ Digest sha3_leaf(uint64_t value) {
    uint32_t l[25], h[25];
#pragma unroll
    for (int i = 0; i < 25; i++) { l[i] = 0; h[i] = 0; }
    l[0] = (uint32_t)value;
    h[0] = (uint32_t)(value >> 32);
    l[1] = 0x06u;
    h[16] = 0x80000000u;
    keccak_f1600_32(l, h);
    return digest_of(l, h);
}

// SHA3-256 of left || right (64 bytes; pad: 0x06 at byte 64, 0x80 at byte 135)
__device__ __forceinline__ Digest sha3_node(const Digest &a, const Digest &b) {
    uint32_t l[25], h[25];
#pragma unroll
    for (int i = 0; i < 25; i++) { l[i] = 0; h[i] = 0; }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        l[i] = (uint32_t)a.w[i]; h[i] = (uint32_t)(a.w[i] >> 32);
        l[4 + i] = (uint32_t)b.w[i]; h[4 + i] = (uint32_t)(b.w[i] >> 32);
    }
    l[8] = 0x06u;
    h[16] = 0x80000000u;
    keccak_f1600_32(l, h);
    return digest_of(l, h);
}

}  // namespace zk
