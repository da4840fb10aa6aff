// Keccak-f[1600] / SHA3-256 single-block hashing: one thread = one hash.  Both hashes on the Merkle path
// fit one rate block (136 B), i.e. exactly one permutation each:
//   leaf  = SHA3-256(LE64(value))        src/core/hash.zig:135-147  (hashFieldElementSHA3)
//   node  = SHA3-256(left || right)      src/core/hash.zig:187-195  (mergeHashesSHA3)
// This work is integer-ALU bound.  The device permutation keeps the 25 lanes as 50 32-bit VGPRs (bit-interleaved,
// see below) and is written for the gfx950 VALU: 3-input boolean ops through v_bitop3_b32 (XOR3 = 0x96,
// chi a^(~b&c) = 0xD2) and rotations as single-source v_alignbit_b32 -- about 172 VALU ops per round, 4.1 k per
// permutation (hipcc's own lowering of the 64-bit formulation needs 6.5 k: 2-input XORs, v_bfi+xor for chi, 64-bit
// shifts).  The same source builds on the host (plain C fallbacks for the three primitives) for unit tests
// (tests/c_driver/keccak_forms.cpp), and the 64-bit macro formulation below is what the host-side sponge uses.
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


// ---- 32-bit bit-interleaved formulation (device kernels) ----
// Lane L is held as two 32-bit words: e = its even bits (bit 2i of L -> bit i of e), o = its odd bits.  A 64-bit
// rotation by 2k is then rot32(e,k), rot32(o,k); by 2k+1 it is e' = rot32(o,k+1), o' = rot32(e,k): theta's rot-by-1
// and rho's rot-by-1 need one v_alignbit_b32 instead of two, i.e. 52 instead of 58 rotate instructions per round (the
// rotate is the slow instruction: 35 T lane-ops/s against 55 T for v_bitop3_b32, tools/valu_rate.hip).  XOR3 / chi are
// bitwise, so they are unchanged: v_bitop3_b32 0x96 / 0xD2.  Per round: 120 v_bitop3 + 52 v_alignbit + iota.
// The round is emitted phase by phase (tools/gen_keccak_il.py generates the body) -- 20 XOR3 | 5 rotates | 50 XOR3 |
// 47 rotates | 50 chi -- with scheduling barriers so each phase stays one instruction class, and the wave sleeps ~100
// cycles (s_sleep 2) after each rotate phase: v_bitop3_b32 can issue on both VALU pipes (every second one on the second
// pipe) but a wave that has just issued rotates keeps its following v_bitop3 on one pipe; the pause restores the pairing
// (SQ_ACTIVE_INST_VALU2 / SQ_INSTS_VALU 0.12 -> 0.27, 10.2 -> 11.4 G permutations/s).
// Digests stay in this form inside the tree (node input = output of the children, no conversion); they are
// converted to canonical SHA3 bytes only where they leave the device (roots, authentication paths).
#if defined(__HIP_DEVICE_COMPILE__)
// v_bitop3_b32 through the builtin: the compiler neither folds it with constant operands nor merges it with its neighbours,
// while it does both for plain C operators (but then breaks XOR3 into two v_xor).  So: plain operators exactly where an
// operand is a compile-time constant after unrolling (the zero / padding lanes of the first round of a leaf or node hash
// fold away), the builtin everywhere else.
#ifdef ZK_BITOP3_ALWAYS  // A/B: the round-1 form (builtin everywhere)
#define ZK_ANYCONST(a, b, c) 0
#else
#define ZK_ANYCONST(a, b, c) (__builtin_constant_p(a) || __builtin_constant_p(b) || __builtin_constant_p(c))
#endif
#define ZK_X3(a, b, c) (ZK_ANYCONST(a, b, c) ? ((a) ^ (b) ^ (c)) : __builtin_amdgcn_bitop3_b32((a), (b), (c), 0x96))
#define ZK_CHI(a, b, c) (ZK_ANYCONST(a, b, c) ? ((a) ^ (~(b) & (c))) : __builtin_amdgcn_bitop3_b32((a), (b), (c), 0xD2))
#define ZK_ROT32(x, k) __builtin_amdgcn_alignbit((x), (x), 32 - (k))
#define ZK_STR2(x) #x
#define ZK_STR(x) ZK_STR2(x)
#ifndef ZK_SLEEP_A
#define ZK_SLEEP_A 4  // s_sleep argument after the 5 rotates of theta (64 clocks each)
#endif
#ifndef ZK_SLEEP_B
#define ZK_SLEEP_B 4  // ... after the 47 rotates of rho (round 2 sweep on two boxes: 2 -> 4 is 2-2.5 % faster now that
                      // the leaf kernel holds 7 waves per SIMD; 1 is 6 % slower, 8 is 1 % slower, 12 is 3 % slower)
#endif
#ifndef ZK_REARM_A_ASM
#define ZK_REARM_A_ASM "s_sleep " ZK_STR(ZK_SLEEP_A)
#endif
#ifndef ZK_REARM_B_ASM
#define ZK_REARM_B_ASM "s_sleep " ZK_STR(ZK_SLEEP_B)
#endif
#ifndef ZK_KECCAK_PHASED
#define ZK_KECCAK_PHASED 1  // 0 = leave the instruction order to the compiler (A/B experiments)
#endif
#if ZK_KECCAK_PHASED
#define ZK_PHASE() __builtin_amdgcn_sched_barrier(0)
#define ZK_REARM_(text)                             \
    do {                                            \
        __builtin_amdgcn_sched_barrier(0);          \
        asm volatile(text);                         \
        __builtin_amdgcn_sched_barrier(0);          \
    } while (0)
#define ZK_REARM_A() ZK_REARM_(ZK_REARM_A_ASM)
#define ZK_REARM_B() ZK_REARM_(ZK_REARM_B_ASM)
#else
#define ZK_PHASE()
#define ZK_REARM_A()
#define ZK_REARM_B()
#endif
#else
#define ZK_PHASE()
#define ZK_REARM_A()
#define ZK_REARM_B()
#define ZK_X3(a, b, c) ((a) ^ (b) ^ (c))
#define ZK_CHI(a, b, c) ((a) ^ (~(b) & (c)))
#define ZK_ROT32(x, k) ((uint32_t)(((x) << (k)) | ((x) >> (32 - (k)))))
#endif

// round constants, even / odd bits
__constant__ const uint32_t KECCAK_RC_E[24] = {0x00000001u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000001u, 0x00000001u, 0x00000001u, 0x00000001u, 0x00000000u, 0x00000000u, 0x00000001u, 0x00000000u, 0x00000001u, 0x00000001u, 0x00000001u, 0x00000001u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000001u, 0x00000000u, 0x00000001u, 0x00000000u};
__constant__ const uint32_t KECCAK_RC_O[24] = {0x00000000u, 0x00000089u, 0x8000008bu, 0x80008080u, 0x0000008bu, 0x00008000u, 0x80008088u, 0x80000082u, 0x0000000bu, 0x0000000au, 0x00008082u, 0x00008003u, 0x0000808bu, 0x8000000bu, 0x8000008au, 0x80000081u, 0x80000081u, 0x80000008u, 0x00000083u, 0x80008003u, 0x80008088u, 0x80000088u, 0x00008000u, 0x80008082u};

// e[i], o[i]: even / odd bits of lane i = x + 5y
#ifndef ZK_KECCAK_UNROLL
#define ZK_KECCAK_UNROLL 24
#endif
// PAUSE = false: no re-arm pauses -- for launches whose waves run (almost) alone on their SIMD (the top of a tree, where a
// hash is a link in a dependent chain and the ~12 k sleep cycles per permutation are pure latency)
template <bool PAUSE = true>
__device__ __forceinline__ void keccak_f1600_il(uint32_t e[25], uint32_t o[25]) {
#pragma unroll ZK_KECCAK_UNROLL
    for (int r = 0; r < 24; r++) {
        uint32_t te[25], to[25], be[25], bo[25];
        // <generated by tools/gen_keccak_il.py>
        // phase 1: column parities
        const uint32_t ce0 = ZK_X3(ZK_X3(e[0], e[5], e[10]), e[15], e[20]);
        const uint32_t co0 = ZK_X3(ZK_X3(o[0], o[5], o[10]), o[15], o[20]);
        const uint32_t ce1 = ZK_X3(ZK_X3(e[1], e[6], e[11]), e[16], e[21]);
        const uint32_t co1 = ZK_X3(ZK_X3(o[1], o[6], o[11]), o[16], o[21]);
        const uint32_t ce2 = ZK_X3(ZK_X3(e[2], e[7], e[12]), e[17], e[22]);
        const uint32_t co2 = ZK_X3(ZK_X3(o[2], o[7], o[12]), o[17], o[22]);
        const uint32_t ce3 = ZK_X3(ZK_X3(e[3], e[8], e[13]), e[18], e[23]);
        const uint32_t co3 = ZK_X3(ZK_X3(o[3], o[8], o[13]), o[18], o[23]);
        const uint32_t ce4 = ZK_X3(ZK_X3(e[4], e[9], e[14]), e[19], e[24]);
        const uint32_t co4 = ZK_X3(ZK_X3(o[4], o[9], o[14]), o[19], o[24]);
        ZK_PHASE();  // phase 2: rot64(C[x], 1): even <- rot32(odd, 1), odd <- even
        const uint32_t re0 = ZK_ROT32(co0, 1);
        const uint32_t re1 = ZK_ROT32(co1, 1);
        const uint32_t re2 = ZK_ROT32(co2, 1);
        const uint32_t re3 = ZK_ROT32(co3, 1);
        const uint32_t re4 = ZK_ROT32(co4, 1);
        if (PAUSE) ZK_REARM_A();  // phase 3: theta applied, t = a ^ C[x-1] ^ rot(C[x+1], 1)
        te[0] = ZK_X3(e[0], ce4, re1); to[0] = ZK_X3(o[0], co4, ce1);
        te[1] = ZK_X3(e[1], ce0, re2); to[1] = ZK_X3(o[1], co0, ce2);
        te[2] = ZK_X3(e[2], ce1, re3); to[2] = ZK_X3(o[2], co1, ce3);
        te[3] = ZK_X3(e[3], ce2, re4); to[3] = ZK_X3(o[3], co2, ce4);
        te[4] = ZK_X3(e[4], ce3, re0); to[4] = ZK_X3(o[4], co3, ce0);
        te[5] = ZK_X3(e[5], ce4, re1); to[5] = ZK_X3(o[5], co4, ce1);
        te[6] = ZK_X3(e[6], ce0, re2); to[6] = ZK_X3(o[6], co0, ce2);
        te[7] = ZK_X3(e[7], ce1, re3); to[7] = ZK_X3(o[7], co1, ce3);
        te[8] = ZK_X3(e[8], ce2, re4); to[8] = ZK_X3(o[8], co2, ce4);
        te[9] = ZK_X3(e[9], ce3, re0); to[9] = ZK_X3(o[9], co3, ce0);
        te[10] = ZK_X3(e[10], ce4, re1); to[10] = ZK_X3(o[10], co4, ce1);
        te[11] = ZK_X3(e[11], ce0, re2); to[11] = ZK_X3(o[11], co0, ce2);
        te[12] = ZK_X3(e[12], ce1, re3); to[12] = ZK_X3(o[12], co1, ce3);
        te[13] = ZK_X3(e[13], ce2, re4); to[13] = ZK_X3(o[13], co2, ce4);
        te[14] = ZK_X3(e[14], ce3, re0); to[14] = ZK_X3(o[14], co3, ce0);
        te[15] = ZK_X3(e[15], ce4, re1); to[15] = ZK_X3(o[15], co4, ce1);
        te[16] = ZK_X3(e[16], ce0, re2); to[16] = ZK_X3(o[16], co0, ce2);
        te[17] = ZK_X3(e[17], ce1, re3); to[17] = ZK_X3(o[17], co1, ce3);
        te[18] = ZK_X3(e[18], ce2, re4); to[18] = ZK_X3(o[18], co2, ce4);
        te[19] = ZK_X3(e[19], ce3, re0); to[19] = ZK_X3(o[19], co3, ce0);
        te[20] = ZK_X3(e[20], ce4, re1); to[20] = ZK_X3(o[20], co4, ce1);
        te[21] = ZK_X3(e[21], ce0, re2); to[21] = ZK_X3(o[21], co0, ce2);
        te[22] = ZK_X3(e[22], ce1, re3); to[22] = ZK_X3(o[22], co1, ce3);
        te[23] = ZK_X3(e[23], ce2, re4); to[23] = ZK_X3(o[23], co2, ce4);
        te[24] = ZK_X3(e[24], ce3, re0); to[24] = ZK_X3(o[24], co3, ce0);
        ZK_PHASE();  // phase 4: rho (rotate) and pi (destination index)
        be[0] = te[0]; bo[0] = to[0];  // lane 0, rho 0
        be[10] = ZK_ROT32(to[1], 1); bo[10] = te[1];  // lane 1, rho 1
        be[20] = ZK_ROT32(te[2], 31); bo[20] = ZK_ROT32(to[2], 31);  // lane 2, rho 62
        be[5] = ZK_ROT32(te[3], 14); bo[5] = ZK_ROT32(to[3], 14);  // lane 3, rho 28
        be[15] = ZK_ROT32(to[4], 14); bo[15] = ZK_ROT32(te[4], 13);  // lane 4, rho 27
        be[16] = ZK_ROT32(te[5], 18); bo[16] = ZK_ROT32(to[5], 18);  // lane 5, rho 36
        be[1] = ZK_ROT32(te[6], 22); bo[1] = ZK_ROT32(to[6], 22);  // lane 6, rho 44
        be[11] = ZK_ROT32(te[7], 3); bo[11] = ZK_ROT32(to[7], 3);  // lane 7, rho 6
        be[21] = ZK_ROT32(to[8], 28); bo[21] = ZK_ROT32(te[8], 27);  // lane 8, rho 55
        be[6] = ZK_ROT32(te[9], 10); bo[6] = ZK_ROT32(to[9], 10);  // lane 9, rho 20
        be[7] = ZK_ROT32(to[10], 2); bo[7] = ZK_ROT32(te[10], 1);  // lane 10, rho 3
        be[17] = ZK_ROT32(te[11], 5); bo[17] = ZK_ROT32(to[11], 5);  // lane 11, rho 10
        be[2] = ZK_ROT32(to[12], 22); bo[2] = ZK_ROT32(te[12], 21);  // lane 12, rho 43
        be[12] = ZK_ROT32(to[13], 13); bo[12] = ZK_ROT32(te[13], 12);  // lane 13, rho 25
        be[22] = ZK_ROT32(to[14], 20); bo[22] = ZK_ROT32(te[14], 19);  // lane 14, rho 39
        be[23] = ZK_ROT32(to[15], 21); bo[23] = ZK_ROT32(te[15], 20);  // lane 15, rho 41
        be[8] = ZK_ROT32(to[16], 23); bo[8] = ZK_ROT32(te[16], 22);  // lane 16, rho 45
        be[18] = ZK_ROT32(to[17], 8); bo[18] = ZK_ROT32(te[17], 7);  // lane 17, rho 15
        be[3] = ZK_ROT32(to[18], 11); bo[3] = ZK_ROT32(te[18], 10);  // lane 18, rho 21
        be[13] = ZK_ROT32(te[19], 4); bo[13] = ZK_ROT32(to[19], 4);  // lane 19, rho 8
        be[14] = ZK_ROT32(te[20], 9); bo[14] = ZK_ROT32(to[20], 9);  // lane 20, rho 18
        be[24] = ZK_ROT32(te[21], 1); bo[24] = ZK_ROT32(to[21], 1);  // lane 21, rho 2
        be[9] = ZK_ROT32(to[22], 31); bo[9] = ZK_ROT32(te[22], 30);  // lane 22, rho 61
        be[19] = ZK_ROT32(te[23], 28); bo[19] = ZK_ROT32(to[23], 28);  // lane 23, rho 56
        be[4] = ZK_ROT32(te[24], 7); bo[4] = ZK_ROT32(to[24], 7);  // lane 24, rho 14
        if (PAUSE) ZK_REARM_B();  // phase 5: chi
        e[0] = ZK_CHI(be[0], be[1], be[2]); o[0] = ZK_CHI(bo[0], bo[1], bo[2]);
        e[1] = ZK_CHI(be[1], be[2], be[3]); o[1] = ZK_CHI(bo[1], bo[2], bo[3]);
        e[2] = ZK_CHI(be[2], be[3], be[4]); o[2] = ZK_CHI(bo[2], bo[3], bo[4]);
        e[3] = ZK_CHI(be[3], be[4], be[0]); o[3] = ZK_CHI(bo[3], bo[4], bo[0]);
        e[4] = ZK_CHI(be[4], be[0], be[1]); o[4] = ZK_CHI(bo[4], bo[0], bo[1]);
        e[5] = ZK_CHI(be[5], be[6], be[7]); o[5] = ZK_CHI(bo[5], bo[6], bo[7]);
        e[6] = ZK_CHI(be[6], be[7], be[8]); o[6] = ZK_CHI(bo[6], bo[7], bo[8]);
        e[7] = ZK_CHI(be[7], be[8], be[9]); o[7] = ZK_CHI(bo[7], bo[8], bo[9]);
        e[8] = ZK_CHI(be[8], be[9], be[5]); o[8] = ZK_CHI(bo[8], bo[9], bo[5]);
        e[9] = ZK_CHI(be[9], be[5], be[6]); o[9] = ZK_CHI(bo[9], bo[5], bo[6]);
        e[10] = ZK_CHI(be[10], be[11], be[12]); o[10] = ZK_CHI(bo[10], bo[11], bo[12]);
        e[11] = ZK_CHI(be[11], be[12], be[13]); o[11] = ZK_CHI(bo[11], bo[12], bo[13]);
        e[12] = ZK_CHI(be[12], be[13], be[14]); o[12] = ZK_CHI(bo[12], bo[13], bo[14]);
        e[13] = ZK_CHI(be[13], be[14], be[10]); o[13] = ZK_CHI(bo[13], bo[14], bo[10]);
        e[14] = ZK_CHI(be[14], be[10], be[11]); o[14] = ZK_CHI(bo[14], bo[10], bo[11]);
        e[15] = ZK_CHI(be[15], be[16], be[17]); o[15] = ZK_CHI(bo[15], bo[16], bo[17]);
        e[16] = ZK_CHI(be[16], be[17], be[18]); o[16] = ZK_CHI(bo[16], bo[17], bo[18]);
        e[17] = ZK_CHI(be[17], be[18], be[19]); o[17] = ZK_CHI(bo[17], bo[18], bo[19]);
        e[18] = ZK_CHI(be[18], be[19], be[15]); o[18] = ZK_CHI(bo[18], bo[19], bo[15]);
        e[19] = ZK_CHI(be[19], be[15], be[16]); o[19] = ZK_CHI(bo[19], bo[15], bo[16]);
        e[20] = ZK_CHI(be[20], be[21], be[22]); o[20] = ZK_CHI(bo[20], bo[21], bo[22]);
        e[21] = ZK_CHI(be[21], be[22], be[23]); o[21] = ZK_CHI(bo[21], bo[22], bo[23]);
        e[22] = ZK_CHI(be[22], be[23], be[24]); o[22] = ZK_CHI(bo[22], bo[23], bo[24]);
        e[23] = ZK_CHI(be[23], be[24], be[20]); o[23] = ZK_CHI(bo[23], bo[24], bo[20]);
        e[24] = ZK_CHI(be[24], be[20], be[21]); o[24] = ZK_CHI(bo[24], bo[20], bo[21]);
        // </generated>
        e[0] ^= KECCAK_RC_E[r];
        o[0] ^= KECCAK_RC_O[r];
    }
}

// even bits of x -> low 16 bits
__device__ __forceinline__ uint32_t compress_even(uint32_t x) {
    x &= 0x55555555u;
    x = (x | (x >> 1)) & 0x33333333u;
    x = (x | (x >> 2)) & 0x0f0f0f0fu;
    x = (x | (x >> 4)) & 0x00ff00ffu;
    x = (x | (x >> 8)) & 0x0000ffffu;
    return x;
}
// low 16 bits of x -> even bit positions
__device__ __forceinline__ uint32_t expand_even(uint32_t x) {
    x &= 0x0000ffffu;
    x = (x | (x << 8)) & 0x00ff00ffu;
    x = (x | (x << 4)) & 0x0f0f0f0fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}

// A SHA3-256 digest.  Inside the tree ("tree form"): w[i] = (odd bits of lane i) << 32 | (even bits of lane i).
// canonical_digest() gives the 4 little-endian lanes of the standard byte string.
struct Digest {
    uint64_t w[4];
};

__device__ __forceinline__ Digest digest_of(const uint32_t e[25], const uint32_t o[25]) {
    return Digest{{((uint64_t)o[0] << 32) | e[0], ((uint64_t)o[1] << 32) | e[1], ((uint64_t)o[2] << 32) | e[2],
                   ((uint64_t)o[3] << 32) | e[3]}};
}

__device__ __forceinline__ Digest canonical_digest(const Digest &t) {
    Digest c;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t e = (uint32_t)t.w[i], o = (uint32_t)(t.w[i] >> 32);
        const uint32_t lo = expand_even(e) | (expand_even(o) << 1);
        const uint32_t hi = expand_even(e >> 16) | (expand_even(o >> 16) << 1);
        c.w[i] = ((uint64_t)hi << 32) | lo;
    }
    return c;
}

// SHA3-256 of the 8 LE bytes of a canonical field element (pad: 0x06 at byte 8, 0x80 at byte 135), tree form
template <bool PAUSE = true>
__device__ __forceinline__ Digest sha3_leaf(uint64_t value) {
    uint32_t e[25], o[25];
#pragma unroll
    for (int i = 0; i < 25; i++) { e[i] = 0; o[i] = 0; }
    const uint32_t lo = (uint32_t)value, hi = (uint32_t)(value >> 32);
    e[0] = compress_even(lo) | (compress_even(hi) << 16);
    o[0] = compress_even(lo >> 1) | (compress_even(hi >> 1) << 16);
    e[1] = 0x2u;  // lane 1 = 0x06: bit 1 (odd bit 0), bit 2 (even bit 1)
    o[1] = 0x1u;
    o[16] = 0x80000000u;  // lane 16 = 1 << 63: odd bit 31
    keccak_f1600_il<PAUSE>(e, o);
    return digest_of(e, o);
}

// SHA3-256 of left || right (64 bytes; pad: 0x06 at byte 64, 0x80 at byte 135); inputs and output in tree form
template <bool PAUSE = true>
__device__ __forceinline__ Digest sha3_node(const Digest &a, const Digest &b) {
    uint32_t e[25], o[25];
#pragma unroll
    for (int i = 0; i < 25; i++) { e[i] = 0; o[i] = 0; }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        e[i] = (uint32_t)a.w[i]; o[i] = (uint32_t)(a.w[i] >> 32);
        e[4 + i] = (uint32_t)b.w[i]; o[4 + i] = (uint32_t)(b.w[i] >> 32);
    }
    e[8] = 0x2u;
    o[8] = 0x1u;
    o[16] = 0x80000000u;
    keccak_f1600_il<PAUSE>(e, o);
    return digest_of(e, o);
}

}  // namespace zk
