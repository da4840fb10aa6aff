// How many independent Keccak-f[1600] states does one core advance per unit time when the 25 lanes sit in xmm / ymm / zmm
// registers (1 / 4 / 8 states per register, the same instruction stream)?  Decides whether a "sponge server" that advances
// several proofs' transcripts in lock step would pay, and calibrates the core clock with a dependent add chain.
//   clang++ -O3 -std=c++17 -mavx512f -mavx512vl tools/host_keccak_wide.cpp -o /tmp/hkw && /tmp/hkw
#include <immintrin.h>
#include <stdint.h>
#include <stdio.h>
#include <chrono>

static const uint64_t RC[24] = {
    0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull,
    0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull,
    0x0000000080008009ull, 0x000000008000000aull, 0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull,
    0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
    0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};

struct V128 { typedef __m128i T; static constexpr int W = 2;
    static T ld(const uint64_t *p) { return _mm_loadu_si128((const __m128i *)p); } static void st(uint64_t *p, T v) { _mm_storeu_si128((__m128i *)p, v); }
    static T x3(T a, T b, T c) { return _mm_ternarylogic_epi64(a, b, c, 0x96); } static T chi(T a, T b, T c) { return _mm_ternarylogic_epi64(a, b, c, 0xD2); }
    template <int N> static T rol(T a) { return _mm_rol_epi64(a, N); } static T x2(T a, T b) { return _mm_xor_si128(a, b); } static T bc(uint64_t v) { return _mm_set1_epi64x((long long)v); } };
struct V256 { typedef __m256i T; static constexpr int W = 4;
    static T ld(const uint64_t *p) { return _mm256_loadu_si256((const __m256i *)p); } static void st(uint64_t *p, T v) { _mm256_storeu_si256((__m256i *)p, v); }
    static T x3(T a, T b, T c) { return _mm256_ternarylogic_epi64(a, b, c, 0x96); } static T chi(T a, T b, T c) { return _mm256_ternarylogic_epi64(a, b, c, 0xD2); }
    template <int N> static T rol(T a) { return _mm256_rol_epi64(a, N); } static T x2(T a, T b) { return _mm256_xor_si256(a, b); } static T bc(uint64_t v) { return _mm256_set1_epi64x((long long)v); } };
struct V512 { typedef __m512i T; static constexpr int W = 8;
    static T ld(const uint64_t *p) { return _mm512_loadu_si512(p); } static void st(uint64_t *p, T v) { _mm512_storeu_si512(p, v); }
    static T x3(T a, T b, T c) { return _mm512_ternarylogic_epi64(a, b, c, 0x96); } static T chi(T a, T b, T c) { return _mm512_ternarylogic_epi64(a, b, c, 0xD2); }
    template <int N> static T rol(T a) { return _mm512_rol_epi64(a, N); } static T x2(T a, T b) { return _mm512_xor_si512(a, b); } static T bc(uint64_t v) { return _mm512_set1_epi64((long long)v); } };

template <class V>
__attribute__((target("avx512f,avx512vl"), noinline)) void perm(uint64_t *st) {  // st[25][W]
    typedef typename V::T T;
    constexpr int W = V::W;
#define LD(i) V::ld(st + (i) * W)
    T a0 = LD(0), a1 = LD(1), a2 = LD(2), a3 = LD(3), a4 = LD(4), a5 = LD(5), a6 = LD(6), a7 = LD(7), a8 = LD(8), a9 = LD(9);
    T a10 = LD(10), a11 = LD(11), a12 = LD(12), a13 = LD(13), a14 = LD(14), a15 = LD(15), a16 = LD(16), a17 = LD(17);
    T a18 = LD(18), a19 = LD(19), a20 = LD(20), a21 = LD(21), a22 = LD(22), a23 = LD(23), a24 = LD(24);
#pragma unroll 2
    for (int r = 0; r < 24; r++) {
        const T c0 = V::x3(V::x3(a0, a5, a10), a15, a20), c1 = V::x3(V::x3(a1, a6, a11), a16, a21), c2 = V::x3(V::x3(a2, a7, a12), a17, a22);
        const T c3 = V::x3(V::x3(a3, a8, a13), a18, a23), c4 = V::x3(V::x3(a4, a9, a14), a19, a24);
        const T r0 = V::template rol<1>(c0), r1 = V::template rol<1>(c1), r2 = V::template rol<1>(c2), r3 = V::template rol<1>(c3), r4 = V::template rol<1>(c4);
#define B(n, a, cm, rp) V::template rol<n>(V::x3(a, cm, rp))
        const T b00 = V::x3(a0, c4, r1);
        const T b10 = B(1, a1, c0, r2), b20 = B(62, a2, c1, r3), b05 = B(28, a3, c2, r4), b15 = B(27, a4, c3, r0);
        const T b16 = B(36, a5, c4, r1), b01 = B(44, a6, c0, r2), b11 = B(6, a7, c1, r3), b21 = B(55, a8, c2, r4);
        const T b06 = B(20, a9, c3, r0), b07 = B(3, a10, c4, r1), b17 = B(10, a11, c0, r2), b02 = B(43, a12, c1, r3);
        const T b12 = B(25, a13, c2, r4), b22 = B(39, a14, c3, r0), b23 = B(41, a15, c4, r1), b08 = B(45, a16, c0, r2);
        const T b18 = B(15, a17, c1, r3), b03 = B(21, a18, c2, r4), b13 = B(8, a19, c3, r0), b14 = B(18, a20, c4, r1);
        const T b24 = B(2, a21, c0, r2), b09 = B(61, a22, c1, r3), b19 = B(56, a23, c2, r4), b04 = B(14, a24, c3, r0);
        a0 = V::x2(V::chi(b00, b01, b02), V::bc(RC[r]));
        a1 = V::chi(b01, b02, b03); a2 = V::chi(b02, b03, b04); a3 = V::chi(b03, b04, b00); a4 = V::chi(b04, b00, b01);
        a5 = V::chi(b05, b06, b07); a6 = V::chi(b06, b07, b08); a7 = V::chi(b07, b08, b09); a8 = V::chi(b08, b09, b05); a9 = V::chi(b09, b05, b06);
        a10 = V::chi(b10, b11, b12); a11 = V::chi(b11, b12, b13); a12 = V::chi(b12, b13, b14); a13 = V::chi(b13, b14, b10); a14 = V::chi(b14, b10, b11);
        a15 = V::chi(b15, b16, b17); a16 = V::chi(b16, b17, b18); a17 = V::chi(b17, b18, b19); a18 = V::chi(b18, b19, b15); a19 = V::chi(b19, b15, b16);
        a20 = V::chi(b20, b21, b22); a21 = V::chi(b21, b22, b23); a22 = V::chi(b22, b23, b24); a23 = V::chi(b23, b24, b20); a24 = V::chi(b24, b20, b21);
    }
#define ST(i, v) V::st(st + (i) * W, v)
    ST(0, a0); ST(1, a1); ST(2, a2); ST(3, a3); ST(4, a4); ST(5, a5); ST(6, a6); ST(7, a7); ST(8, a8); ST(9, a9); ST(10, a10);
    ST(11, a11); ST(12, a12); ST(13, a13); ST(14, a14); ST(15, a15); ST(16, a16); ST(17, a17); ST(18, a18); ST(19, a19);
    ST(20, a20); ST(21, a21); ST(22, a22); ST(23, a23); ST(24, a24);
}

template <class V>
static void run(const char *name) {
    alignas(64) uint64_t s[25 * 8] = {1, 2, 3, 4, 5, 6, 7, 8};
    const int n = 2000000;
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; i++) perm<V>(s);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%s: %.4f us per call = %.4f us per state (%d states per call)  [%llx]\n", name, dt / n * 1e6, dt / n * 1e6 / V::W, V::W,
           (unsigned long long)s[0]);
}

int main() {
    {   // core clock: 2e9 dependent 1-cycle adds
        uint64_t x = 1;
        auto t0 = std::chrono::steady_clock::now();
        for (long i = 0; i < 500000000L; i++) { __asm__ volatile("add %0, %0\n add %0, %0\n add %0, %0\n add %0, %0" : "+r"(x)); }
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("core clock ~ %.2f GHz (dependent add chain)\n", 2e9 / dt / 1e9);
    }
    for (int rep = 0; rep < 2; rep++) { run<V128>("xmm (2 states)"); run<V256>("ymm (4 states)"); run<V512>("zmm (8 states)"); }
    return 0;
}
