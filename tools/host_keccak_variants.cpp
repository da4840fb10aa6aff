// A/B harness for the one-lane-per-xmm host Keccak-f[1600] (zigz_amd/csrc/host_keccak_avx512vl.cpp): which of the round's
// three-input booleans should be ONE vpternlogq and which two plain vpxor / vpandn?  On Zen 4/5 vpternlogq issues on fewer
// pipes than vpxor, so trading some of the 60 ternlogs per round for pairs of plain ops can shorten the round even though
// it adds instructions.  Build + run on the target host:  clang++ -O3 -std=c++17 -mavx512f -mavx512vl tools/host_keccak_variants.cpp -o /tmp/hkv && /tmp/hkv
//   THETA_C   0: column parities by 2 ternlogs   1: by 4 xors
//   THETA_A   0: a ^ c[x-1] ^ rol(c[x+1]) by 1 ternlog per lane   1: d[x] first (5 xors), then 1 xor per lane
//   CHI       0: 1 ternlog per lane   1: vpandn + vpxor
#include <immintrin.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <chrono>

static const uint64_t RC[24] = {
    0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull,
    0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull,
    0x0000000080008009ull, 0x000000008000000aull, 0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull,
    0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
    0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};

#define BAR(v) __asm__("" : "+x"(v))  // keeps the compiler from re-fusing two plain ops into a ternlog
static inline __m128i X2(__m128i a, __m128i b) { __m128i t = _mm_xor_si128(a, b); BAR(t); return t; }
static inline __m128i T3(__m128i a, __m128i b, __m128i c) { return _mm_ternarylogic_epi64(a, b, c, 0x96); }
static inline __m128i TCHI(__m128i a, __m128i b, __m128i c) { return _mm_ternarylogic_epi64(a, b, c, 0xD2); }
static inline __m128i PCHI(__m128i a, __m128i b, __m128i c) { __m128i t = _mm_andnot_si128(b, c); BAR(t); return X2(a, t); }
#define ROL(a, n) _mm_rol_epi64((a), (n))

template <int THETA_C, int THETA_A, int CHI>
__attribute__((target("avx512f,avx512vl"), noinline)) void perm(uint64_t st[25]) {
#define LD(i) _mm_loadl_epi64((const __m128i *)(st + (i)))
    __m128i a0 = LD(0), a1 = LD(1), a2 = LD(2), a3 = LD(3), a4 = LD(4), a5 = LD(5), a6 = LD(6), a7 = LD(7), a8 = LD(8), a9 = LD(9);
    __m128i a10 = LD(10), a11 = LD(11), a12 = LD(12), a13 = LD(13), a14 = LD(14), a15 = LD(15), a16 = LD(16), a17 = LD(17);
    __m128i a18 = LD(18), a19 = LD(19), a20 = LD(20), a21 = LD(21), a22 = LD(22), a23 = LD(23), a24 = LD(24);
#define PAR(p, q, r, s, t) (THETA_C ? X2(X2(X2(p, q), X2(r, s)), t) : T3(T3(p, q, r), s, t))
#define CHIF(a, b, c) (CHI ? PCHI(a, b, c) : TCHI(a, b, c))
#pragma unroll 2
    for (int r = 0; r < 24; r++) {
        const __m128i c0 = PAR(a0, a5, a10, a15, a20), c1 = PAR(a1, a6, a11, a16, a21), c2 = PAR(a2, a7, a12, a17, a22);
        const __m128i c3 = PAR(a3, a8, a13, a18, a23), c4 = PAR(a4, a9, a14, a19, a24);
        const __m128i r0 = ROL(c0, 1), r1 = ROL(c1, 1), r2 = ROL(c2, 1), r3 = ROL(c3, 1), r4 = ROL(c4, 1);
        __m128i d0, d1, d2, d3, d4;
        if (THETA_A) { d0 = X2(c4, r1); d1 = X2(c0, r2); d2 = X2(c1, r3); d3 = X2(c2, r4); d4 = X2(c3, r0); }
#define TH(a, cm, rp, d) (THETA_A ? X2(a, d) : T3(a, cm, rp))
        const __m128i b00 = TH(a0, c4, r1, d0);
        const __m128i b10 = ROL(TH(a1, c0, r2, d1), 1), b20 = ROL(TH(a2, c1, r3, d2), 62);
        const __m128i b05 = ROL(TH(a3, c2, r4, d3), 28), b15 = ROL(TH(a4, c3, r0, d4), 27);
        const __m128i b16 = ROL(TH(a5, c4, r1, d0), 36), b01 = ROL(TH(a6, c0, r2, d1), 44);
        const __m128i b11 = ROL(TH(a7, c1, r3, d2), 6), b21 = ROL(TH(a8, c2, r4, d3), 55);
        const __m128i b06 = ROL(TH(a9, c3, r0, d4), 20), b07 = ROL(TH(a10, c4, r1, d0), 3);
        const __m128i b17 = ROL(TH(a11, c0, r2, d1), 10), b02 = ROL(TH(a12, c1, r3, d2), 43);
        const __m128i b12 = ROL(TH(a13, c2, r4, d3), 25), b22 = ROL(TH(a14, c3, r0, d4), 39);
        const __m128i b23 = ROL(TH(a15, c4, r1, d0), 41), b08 = ROL(TH(a16, c0, r2, d1), 45);
        const __m128i b18 = ROL(TH(a17, c1, r3, d2), 15), b03 = ROL(TH(a18, c2, r4, d3), 21);
        const __m128i b13 = ROL(TH(a19, c3, r0, d4), 8), b14 = ROL(TH(a20, c4, r1, d0), 18);
        const __m128i b24 = ROL(TH(a21, c0, r2, d1), 2), b09 = ROL(TH(a22, c1, r3, d2), 61);
        const __m128i b19 = ROL(TH(a23, c2, r4, d3), 56), b04 = ROL(TH(a24, c3, r0, d4), 14);
        a0 = _mm_xor_si128(CHIF(b00, b01, b02), _mm_cvtsi64_si128((long long)RC[r]));
        a1 = CHIF(b01, b02, b03); a2 = CHIF(b02, b03, b04); a3 = CHIF(b03, b04, b00); a4 = CHIF(b04, b00, b01);
        a5 = CHIF(b05, b06, b07); a6 = CHIF(b06, b07, b08); a7 = CHIF(b07, b08, b09); a8 = CHIF(b08, b09, b05); a9 = CHIF(b09, b05, b06);
        a10 = CHIF(b10, b11, b12); a11 = CHIF(b11, b12, b13); a12 = CHIF(b12, b13, b14); a13 = CHIF(b13, b14, b10); a14 = CHIF(b14, b10, b11);
        a15 = CHIF(b15, b16, b17); a16 = CHIF(b16, b17, b18); a17 = CHIF(b17, b18, b19); a18 = CHIF(b18, b19, b15); a19 = CHIF(b19, b15, b16);
        a20 = CHIF(b20, b21, b22); a21 = CHIF(b21, b22, b23); a22 = CHIF(b22, b23, b24); a23 = CHIF(b23, b24, b20); a24 = CHIF(b24, b20, b21);
    }
#define ST(i, v) _mm_storel_epi64((__m128i *)(st + (i)), (v))
    ST(0, a0); ST(1, a1); ST(2, a2); ST(3, a3); ST(4, a4); ST(5, a5); ST(6, a6); ST(7, a7); ST(8, a8); ST(9, a9); ST(10, a10);
    ST(11, a11); ST(12, a12); ST(13, a13); ST(14, a14); ST(15, a15); ST(16, a16); ST(17, a17); ST(18, a18); ST(19, a19);
    ST(20, a20); ST(21, a21); ST(22, a22); ST(23, a23); ST(24, a24);
}

template <int C, int A, int X>
static void run(const uint64_t ref[25]) {
    uint64_t s[25] = {1};
    const int n = 3000000;
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; i++) perm<C, A, X>(s);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    uint64_t one[25] = {1};
    perm<C, A, X>(one);
    printf("THETA_C=%d THETA_A=%d CHI=%d  %.4f us/perm  %s\n", C, A, X, dt / n * 1e6, memcmp(one, ref, 200) == 0 ? "ok" : "MISMATCH");
}

int main() {
    uint64_t ref[25] = {1};
    perm<0, 0, 0>(ref);
    for (int rep = 0; rep < 2; rep++) {
        run<0, 0, 0>(ref); run<0, 1, 0>(ref); run<1, 0, 0>(ref); run<1, 1, 0>(ref);
        run<0, 0, 1>(ref); run<0, 1, 1>(ref); run<1, 1, 1>(ref); run<1, 0, 1>(ref);
    }
    return 0;
}
