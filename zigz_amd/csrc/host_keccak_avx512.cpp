// Keccak-f[1600] for the host Fiat-Shamir sponge with AVX-512F: the state is held as five "planes"
// (plane y = lanes A[0..4][y] in the low five qwords of one zmm register).  Per round:
//   theta  column parities and D by two lane rotations of the parity vector
//   rho    one variable rotate (vprolvq) per plane
//   pi     one in-plane permute per plane gives Q_X[Y] = B[Y][X] (the new state transposed)
//   chi    lane-wise ternary logic ACROSS the five Q registers (no shuffles)
//   a 5x5 qword transpose (4 unpacks + 5 two-source permutes + 5 masked permutes) restores plane order
// The sponge is sequential by construction, so this single-state permutation is the
// critical path of Prover.prove once the Merkle work runs on the GPU.
#include <immintrin.h>
#include <stdint.h>

namespace zk {

static const uint64_t RC512[24] = {
    0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull,
    0x000000000000808bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
    0x000000000000008aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000aull,
    0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull,
    0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
    0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};

__attribute__((target("avx512f"))) void keccak_f1600_avx512(uint64_t st[25]) {
    const __mmask8 M5 = 0x1f;
    __m512i p0 = _mm512_maskz_loadu_epi64(M5, st + 0), p1 = _mm512_maskz_loadu_epi64(M5, st + 5);
    __m512i p2 = _mm512_maskz_loadu_epi64(M5, st + 10), p3 = _mm512_maskz_loadu_epi64(M5, st + 15);
    __m512i p4 = _mm512_maskz_loadu_epi64(M5, st + 20);
    const __m512i left = _mm512_setr_epi64(4, 0, 1, 2, 3, 5, 6, 7);   // lane x <- C[x-1]
    const __m512i right = _mm512_setr_epi64(1, 2, 3, 4, 0, 5, 6, 7);  // lane x <- C[x+1]
    const __m512i rho0 = _mm512_setr_epi64(0, 1, 62, 28, 27, 0, 0, 0), rho1 = _mm512_setr_epi64(36, 44, 6, 55, 20, 0, 0, 0);
    const __m512i rho2 = _mm512_setr_epi64(3, 10, 43, 25, 39, 0, 0, 0), rho3 = _mm512_setr_epi64(41, 45, 15, 21, 8, 0, 0, 0);
    const __m512i rho4 = _mm512_setr_epi64(18, 2, 61, 56, 14, 0, 0, 0);
    // Q_X[Y] = P_X[(X + 3Y) % 5]
    const __m512i pi0 = _mm512_setr_epi64(0, 3, 1, 4, 2, 5, 6, 7), pi1 = _mm512_setr_epi64(1, 4, 2, 0, 3, 5, 6, 7);
    const __m512i pi2 = _mm512_setr_epi64(2, 0, 3, 1, 4, 5, 6, 7), pi3 = _mm512_setr_epi64(3, 1, 4, 2, 0, 5, 6, 7);
    const __m512i pi4 = _mm512_setr_epi64(4, 2, 0, 3, 1, 5, 6, 7);
    const __m512i t01 = _mm512_setr_epi64(0, 1, 8, 9, 0, 0, 0, 0), t23 = _mm512_setr_epi64(2, 3, 10, 11, 0, 0, 0, 0);
    const __m512i t45 = _mm512_setr_epi64(4, 5, 12, 13, 0, 0, 0, 0);
    const __m512i l0 = _mm512_set1_epi64(0), l1 = _mm512_set1_epi64(1), l2 = _mm512_set1_epi64(2);
    const __m512i l3 = _mm512_set1_epi64(3), l4 = _mm512_set1_epi64(4);
    for (int r = 0; r < 24; r++) {
        // theta
        __m512i c = _mm512_ternarylogic_epi64(_mm512_ternarylogic_epi64(p0, p1, p2, 0x96), p3, p4, 0x96);
        __m512i d = _mm512_xor_si512(_mm512_permutexvar_epi64(left, c), _mm512_rol_epi64(_mm512_permutexvar_epi64(right, c), 1));
        // theta apply + rho + pi (in-plane part)
        __m512i q0 = _mm512_permutexvar_epi64(pi0, _mm512_rolv_epi64(_mm512_xor_si512(p0, d), rho0));
        __m512i q1 = _mm512_permutexvar_epi64(pi1, _mm512_rolv_epi64(_mm512_xor_si512(p1, d), rho1));
        __m512i q2 = _mm512_permutexvar_epi64(pi2, _mm512_rolv_epi64(_mm512_xor_si512(p2, d), rho2));
        __m512i q3 = _mm512_permutexvar_epi64(pi3, _mm512_rolv_epi64(_mm512_xor_si512(p3, d), rho3));
        __m512i q4 = _mm512_permutexvar_epi64(pi4, _mm512_rolv_epi64(_mm512_xor_si512(p4, d), rho4));
        // chi across registers: e_X[Y] = new A[X][Y];  f(a,b,c) = a ^ (~b & c) = 0xD2
        __m512i e0 = _mm512_ternarylogic_epi64(q0, q1, q2, 0xD2), e1 = _mm512_ternarylogic_epi64(q1, q2, q3, 0xD2);
        __m512i e2 = _mm512_ternarylogic_epi64(q2, q3, q4, 0xD2), e3 = _mm512_ternarylogic_epi64(q3, q4, q0, 0xD2);
        __m512i e4 = _mm512_ternarylogic_epi64(q4, q0, q1, 0xD2);
        // transpose back: p_y[x] = e_x[y]
        __m512i u0 = _mm512_unpacklo_epi64(e0, e1), u1 = _mm512_unpackhi_epi64(e0, e1);
        __m512i v0 = _mm512_unpacklo_epi64(e2, e3), v1 = _mm512_unpackhi_epi64(e2, e3);
        p0 = _mm512_mask_permutexvar_epi64(_mm512_permutex2var_epi64(u0, t01, v0), 0x10, l0, e4);
        p1 = _mm512_mask_permutexvar_epi64(_mm512_permutex2var_epi64(u1, t01, v1), 0x10, l1, e4);
        p2 = _mm512_mask_permutexvar_epi64(_mm512_permutex2var_epi64(u0, t23, v0), 0x10, l2, e4);
        p3 = _mm512_mask_permutexvar_epi64(_mm512_permutex2var_epi64(u1, t23, v1), 0x10, l3, e4);
        p4 = _mm512_mask_permutexvar_epi64(_mm512_permutex2var_epi64(u0, t45, v0), 0x10, l4, e4);
        // iota
        p0 = _mm512_xor_si512(p0, _mm512_maskz_set1_epi64(1, (long long)RC512[r]));
    }
    _mm512_mask_storeu_epi64(st + 0, M5, p0);
    _mm512_mask_storeu_epi64(st + 5, M5, p1);
    _mm512_mask_storeu_epi64(st + 10, M5, p2);
    _mm512_mask_storeu_epi64(st + 15, M5, p3);
    _mm512_mask_storeu_epi64(st + 20, M5, p4);
}

bool cpu_has_avx512f() { return __builtin_cpu_supports("avx512f"); }

}  // namespace zk
