// Keccak-f[1600] for the host Fiat-Shamir sponge with AVX-512VL, ONE LANE PER VECTOR REGISTER: the 25 lanes live in the
// low qwords of 25 xmm registers for the whole permutation (x86-64 has 16 general registers but 32 vector registers),
// every XOR3 and chi is one vpternlogq (the CPU's counterpart of gfx950's v_bitop3_b32 -- this is the formulation of
// keccak.hpp's device round), every rho rotation one vprolq, and pi is register renaming in the fully unrolled code.
// 90 one-cycle-latency operations per round with no memory traffic and short dependency chains, against ~130 scalar
// operations plus spills for the 64-bit BMI2 form: throughput-bound on the vector ALUs instead of latency-bound on
// cross-lane permutes like the five-plane form (host_keccak_avx512.cpp).  Chosen at load time by calibration when it is
// the fastest supported variant (host_hash.cpp).
#include <immintrin.h>
#include <stdint.h>

namespace zk {

static const uint64_t RCVL[24] = {
    0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull,
    0x000000000000808bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
    0x000000000000008aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000aull,
    0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull,
    0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
    0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};

#define VL_X3(a, b, c) _mm_ternarylogic_epi64((a), (b), (c), 0x96)
#define VL_CHI(a, b, c) _mm_ternarylogic_epi64((a), (b), (c), 0xD2)
#define VL_ROL(a, n) _mm_rol_epi64((a), (n))

// one round on lanes a[x + 5y]; b?? named by destination index after pi (as ZK_KECCAK_ROUND in keccak.hpp)
#define VL_ROUND(rc)                                                                                              \
    do {                                                                                                          \
        const __m128i c0 = VL_X3(VL_X3(a0, a5, a10), a15, a20), c1 = VL_X3(VL_X3(a1, a6, a11), a16, a21);          \
        const __m128i c2 = VL_X3(VL_X3(a2, a7, a12), a17, a22), c3 = VL_X3(VL_X3(a3, a8, a13), a18, a23);          \
        const __m128i c4 = VL_X3(VL_X3(a4, a9, a14), a19, a24);                                                    \
        const __m128i r0 = VL_ROL(c0, 1), r1 = VL_ROL(c1, 1), r2 = VL_ROL(c2, 1), r3 = VL_ROL(c3, 1), r4 = VL_ROL(c4, 1); \
        /* theta applied (a ^ C[x-1] ^ rol(C[x+1], 1)) + rho + pi: b[y + 5*((2x+3y)%5)] = rol(t[x+5y], r[x][y]) */   \
        const __m128i b00 = VL_X3(a0, c4, r1);                                                                     \
        const __m128i b10 = VL_ROL(VL_X3(a1, c0, r2), 1), b20 = VL_ROL(VL_X3(a2, c1, r3), 62);                     \
        const __m128i b05 = VL_ROL(VL_X3(a3, c2, r4), 28), b15 = VL_ROL(VL_X3(a4, c3, r0), 27);                    \
        const __m128i b16 = VL_ROL(VL_X3(a5, c4, r1), 36), b01 = VL_ROL(VL_X3(a6, c0, r2), 44);                    \
        const __m128i b11 = VL_ROL(VL_X3(a7, c1, r3), 6), b21 = VL_ROL(VL_X3(a8, c2, r4), 55);                     \
        const __m128i b06 = VL_ROL(VL_X3(a9, c3, r0), 20), b07 = VL_ROL(VL_X3(a10, c4, r1), 3);                    \
        const __m128i b17 = VL_ROL(VL_X3(a11, c0, r2), 10), b02 = VL_ROL(VL_X3(a12, c1, r3), 43);                  \
        const __m128i b12 = VL_ROL(VL_X3(a13, c2, r4), 25), b22 = VL_ROL(VL_X3(a14, c3, r0), 39);                  \
        const __m128i b23 = VL_ROL(VL_X3(a15, c4, r1), 41), b08 = VL_ROL(VL_X3(a16, c0, r2), 45);                  \
        const __m128i b18 = VL_ROL(VL_X3(a17, c1, r3), 15), b03 = VL_ROL(VL_X3(a18, c2, r4), 21);                  \
        const __m128i b13 = VL_ROL(VL_X3(a19, c3, r0), 8), b14 = VL_ROL(VL_X3(a20, c4, r1), 18);                   \
        const __m128i b24 = VL_ROL(VL_X3(a21, c0, r2), 2), b09 = VL_ROL(VL_X3(a22, c1, r3), 61);                   \
        const __m128i b19 = VL_ROL(VL_X3(a23, c2, r4), 56), b04 = VL_ROL(VL_X3(a24, c3, r0), 14);                  \
        /* chi (+ iota on lane 0) */                                                                              \
        a0 = _mm_xor_si128(VL_CHI(b00, b01, b02), _mm_cvtsi64_si128((long long)(rc)));                             \
        a1 = VL_CHI(b01, b02, b03); a2 = VL_CHI(b02, b03, b04); a3 = VL_CHI(b03, b04, b00); a4 = VL_CHI(b04, b00, b01); \
        a5 = VL_CHI(b05, b06, b07); a6 = VL_CHI(b06, b07, b08); a7 = VL_CHI(b07, b08, b09);                        \
        a8 = VL_CHI(b08, b09, b05); a9 = VL_CHI(b09, b05, b06);                                                    \
        a10 = VL_CHI(b10, b11, b12); a11 = VL_CHI(b11, b12, b13); a12 = VL_CHI(b12, b13, b14);                     \
        a13 = VL_CHI(b13, b14, b10); a14 = VL_CHI(b14, b10, b11);                                                  \
        a15 = VL_CHI(b15, b16, b17); a16 = VL_CHI(b16, b17, b18); a17 = VL_CHI(b17, b18, b19);                     \
        a18 = VL_CHI(b18, b19, b15); a19 = VL_CHI(b19, b15, b16);                                                  \
        a20 = VL_CHI(b20, b21, b22); a21 = VL_CHI(b21, b22, b23); a22 = VL_CHI(b22, b23, b24);                     \
        a23 = VL_CHI(b23, b24, b20); a24 = VL_CHI(b24, b20, b21);                                                  \
    } while (0)

#define VL_LD(i) _mm_loadl_epi64((const __m128i *)(st + (i)))
#define VL_ST(i, v) _mm_storel_epi64((__m128i *)(st + (i)), (v))

__attribute__((target("avx512f,avx512vl"))) void keccak_f1600_avx512vl(uint64_t st[25]) {
    __m128i a0 = VL_LD(0), a1 = VL_LD(1), a2 = VL_LD(2), a3 = VL_LD(3), a4 = VL_LD(4), a5 = VL_LD(5), a6 = VL_LD(6);
    __m128i a7 = VL_LD(7), a8 = VL_LD(8), a9 = VL_LD(9), a10 = VL_LD(10), a11 = VL_LD(11), a12 = VL_LD(12), a13 = VL_LD(13);
    __m128i a14 = VL_LD(14), a15 = VL_LD(15), a16 = VL_LD(16), a17 = VL_LD(17), a18 = VL_LD(18), a19 = VL_LD(19);
    __m128i a20 = VL_LD(20), a21 = VL_LD(21), a22 = VL_LD(22), a23 = VL_LD(23), a24 = VL_LD(24);
#pragma unroll 2
    for (int r = 0; r < 24; r++) VL_ROUND(RCVL[r]);
    VL_ST(0, a0); VL_ST(1, a1); VL_ST(2, a2); VL_ST(3, a3); VL_ST(4, a4); VL_ST(5, a5); VL_ST(6, a6); VL_ST(7, a7);
    VL_ST(8, a8); VL_ST(9, a9); VL_ST(10, a10); VL_ST(11, a11); VL_ST(12, a12); VL_ST(13, a13); VL_ST(14, a14);
    VL_ST(15, a15); VL_ST(16, a16); VL_ST(17, a17); VL_ST(18, a18); VL_ST(19, a19); VL_ST(20, a20); VL_ST(21, a21);
    VL_ST(22, a22); VL_ST(23, a23); VL_ST(24, a24);
}

bool cpu_has_avx512vl() { return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl"); }

}  // namespace zk
