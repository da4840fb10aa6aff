/*
 * zigz_oracle.c -- CPU restatement of the zigz reference hot path.  TEST INFRASTRUCTURE ONLY.
 * See zigz_oracle.h for the parity-pin statement.  Plain C11, single thread, u64 canonical field
 * elements, `%` reduction, naive O(v*2^v) eval, recompute-on-open Merkle: the reference's own
 * algorithms and quirks, restated loop for loop.  Citations are reference file:line.
 */
#include "zigz_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ======================================================================== */
/* Field: src/core/field.zig                                                 */
/* ======================================================================== */

uint64_t orc_f_init(uint64_t p, uint64_t v) { return v % p; } /* field.zig:36-38 */

uint64_t orc_f_add(uint64_t p, uint64_t a, uint64_t b) { /* field.zig:73-88 */
    uint64_t s = a + b;
    if (s < a) return s % p; /* overflow branch, :76-80 (unreachable for p < 2^63) */
    if (s >= p) return s - p;
    return s;
}

uint64_t orc_f_sub(uint64_t p, uint64_t a, uint64_t b) { /* field.zig:91-98 */
    if (a >= b) return a - b;
    return p - (b - a);
}

uint64_t orc_f_neg(uint64_t p, uint64_t a) { return a == 0 ? 0 : p - a; } /* field.zig:101-106 */

uint64_t orc_f_mul(uint64_t p, uint64_t a, uint64_t b) { /* field.zig:123-129 */
    return (uint64_t)(((u128)a * (u128)b) % (u128)p);
}

int orc_f_inv(uint64_t p, uint64_t a, uint64_t *out) { /* field.zig:157-191 */
    if (a == 0) return 1;
    __int128 t = 0, new_t = 1, r = (__int128)p, new_r = (__int128)a;
    while (new_r != 0) {
        __int128 q = r / new_r; /* both positive: divFloor == trunc */
        __int128 tt = t; t = new_t; new_t = tt - q * new_t;
        __int128 tr = r; r = new_r; new_r = tr - q * new_r;
    }
    if (r > 1) return 1;
    if (t < 0) t += (__int128)p;
    *out = (uint64_t)t;
    return 0;
}

uint64_t orc_f_pow(uint64_t p, uint64_t a, uint64_t e) { /* field.zig:204-225 */
    if (e == 0) return 1 % p;
    if (e == 1) return a;
    uint64_t result = 1 % p, base = a;
    while (e > 0) {
        if (e & 1) result = orc_f_mul(p, result, base);
        base = orc_f_mul(p, base, base);
        e >>= 1;
    }
    return result;
}

/* ======================================================================== */
/* Keccak-f[1600] / SHA3-256 (FIPS 202) == Zig std.crypto.hash.sha3.Sha3_256 */
/* ======================================================================== */

static const uint64_t KECCAK_RC[24] = {
    0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull,
    0x000000000000808bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
    0x000000000000008aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000aull,
    0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull,
    0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
    0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
static inline uint64_t rotl64(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }

static void le64(uint64_t v, uint8_t out[8]) {
    for (int b = 0; b < 8; b++) out[b] = (uint8_t)(v >> (8 * b));
}
static uint64_t rd64(const uint8_t *in) {
    uint64_t v = 0;
    for (int b = 0; b < 8; b++) v |= (uint64_t)in[b] << (8 * b);
    return v;
}
static uint32_t rd32(const uint8_t *in) {
    return (uint32_t)in[0] | ((uint32_t)in[1] << 8) | ((uint32_t)in[2] << 16) | ((uint32_t)in[3] << 24);
}



/* One round with theta, rho+pi and chi fully unrolled (lanes st[x + 5y]); rotation offsets and the pi
 * lane permutation are the FIPS 202 tables written out: b[y + 5((2x+3y)%5)] = rot(a[x+5y], r[x][y]). */
#define TH(i) (st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20])
#define RP(dst, src, rot) b[dst] = rotl64(st[src] ^ d[(src) % 5], rot)
#define CH(y) do { \
        st[y] = b[y] ^ (~b[y + 1] & b[y + 2]); st[y + 1] = b[y + 1] ^ (~b[y + 2] & b[y + 3]); \
        st[y + 2] = b[y + 2] ^ (~b[y + 3] & b[y + 4]); st[y + 3] = b[y + 3] ^ (~b[y + 4] & b[y]); \
        st[y + 4] = b[y + 4] ^ (~b[y] & b[y + 1]); } while (0)

static void keccak_f1600(uint64_t st[25]) {
    uint64_t b[25], c[5], d[5];
    for (int round = 0; round < 24; round++) {
        c[0] = TH(0); c[1] = TH(1); c[2] = TH(2); c[3] = TH(3); c[4] = TH(4);
        d[0] = c[4] ^ rotl64(c[1], 1); d[1] = c[0] ^ rotl64(c[2], 1); d[2] = c[1] ^ rotl64(c[3], 1);
        d[3] = c[2] ^ rotl64(c[4], 1); d[4] = c[3] ^ rotl64(c[0], 1);
        b[0] = st[0] ^ d[0];
        RP(10, 1, 1);  RP(20, 2, 62); RP(5, 3, 28);  RP(15, 4, 27);
        RP(16, 5, 36); RP(1, 6, 44);  RP(11, 7, 6);  RP(21, 8, 55); RP(6, 9, 20);
        RP(7, 10, 3);  RP(17, 11, 10); RP(2, 12, 43); RP(12, 13, 25); RP(22, 14, 39);
        RP(23, 15, 41); RP(8, 16, 45); RP(18, 17, 15); RP(3, 18, 21); RP(13, 19, 8);
        RP(14, 20, 18); RP(24, 21, 2); RP(9, 22, 61); RP(19, 23, 56); RP(4, 24, 14);
        CH(0); CH(5); CH(10); CH(15); CH(20);
        st[0] ^= KECCAK_RC[round];
    }
}
#undef TH
#undef RP
#undef CH

#define SHA3_RATE 136 /* SHA3-256: r = 1088 bits */

typedef struct {
    uint64_t st[25];
    size_t pos; /* bytes absorbed into the current block */
} sha3_ctx;

static void sha3_init(sha3_ctx *c) { memset(c, 0, sizeof(*c)); }

static void sha3_update(sha3_ctx *c, const uint8_t *data, size_t len) {
    for (size_t i = 0; i < len; i++) {
        c->st[c->pos >> 3] ^= (uint64_t)data[i] << (8 * (c->pos & 7));
        if (++c->pos == SHA3_RATE) {
            keccak_f1600(c->st);
            c->pos = 0;
        }
    }
}

static void sha3_final(sha3_ctx *c, uint8_t out[32]) {
    c->st[c->pos >> 3] ^= (uint64_t)0x06 << (8 * (c->pos & 7));
    c->st[(SHA3_RATE - 1) >> 3] ^= (uint64_t)0x80 << (8 * ((SHA3_RATE - 1) & 7));
    keccak_f1600(c->st);
    for (int i = 0; i < 4; i++)
        for (int b = 0; b < 8; b++) out[i * 8 + b] = (uint8_t)(c->st[i] >> (8 * b));
}

void orc_sha3_256(const uint8_t *data, size_t len, uint8_t out[32]) {
    sha3_ctx c;
    sha3_init(&c);
    sha3_update(&c, data, len);
    sha3_final(&c, out);
}

/* hashFieldElementSHA3, hash.zig:135-147: SHA3-256 over the 8 LE bytes of the canonical value */
void orc_hash_leaf(uint64_t value, uint8_t out[32]) {
    uint64_t st[25] = {0};
    st[0] = value;                    /* 8 message bytes, little endian */
    st[1] = 0x06;                     /* SHA3 domain/pad byte at offset 8 */
    st[16] = 0x8000000000000000ull;   /* final pad bit at byte 135 */
    keccak_f1600(st);
    for (int i = 0; i < 4; i++) le64(st[i], out + 8 * i);
}

/* mergeHashesSHA3, hash.zig:187-195 */
void orc_hash_internal(const uint8_t l[32], const uint8_t r[32], uint8_t out[32]) {
    uint64_t st[25] = {0};
    for (int i = 0; i < 4; i++) { st[i] = rd64(l + 8 * i); st[4 + i] = rd64(r + 8 * i); }
    st[8] = 0x06;
    st[16] = 0x8000000000000000ull;
    keccak_f1600(st);
    for (int i = 0; i < 4; i++) le64(st[i], out + 8 * i);
}

/* ======================================================================== */
/* SHA-256 (FIPS 180-4) == Zig std.crypto.hash.sha2.Sha256 (prover.zig:98-100) */
/* ======================================================================== */

static const uint32_t SHA256_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

static inline uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

static void sha256_block(uint32_t h[8], const uint8_t blk[64]) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++)
        w[i] = ((uint32_t)blk[4 * i] << 24) | ((uint32_t)blk[4 * i + 1] << 16) |
               ((uint32_t)blk[4 * i + 2] << 8) | (uint32_t)blk[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = rotr32(w[i - 15], 7) ^ rotr32(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = rotr32(w[i - 2], 17) ^ rotr32(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
        uint32_t S1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25);
        uint32_t ch = (e & f) ^ (~e & g);
        uint32_t t1 = hh + S1 + ch + SHA256_K[i] + w[i];
        uint32_t S0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22);
        uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
        uint32_t t2 = S0 + mj;
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

void orc_sha256(const uint8_t *data, size_t len, uint8_t out[32]) {
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a,
                     0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    size_t full = len / 64;
    for (size_t i = 0; i < full; i++) sha256_block(h, data + 64 * i);
    uint8_t tail[128];
    size_t rem = len - 64 * full;
    memset(tail, 0, sizeof(tail));
    if (rem) memcpy(tail, data + 64 * full, rem);
    tail[rem] = 0x80;
    size_t tl = (rem < 56) ? 64 : 128;
    uint64_t bits = (uint64_t)len * 8;
    for (int b = 0; b < 8; b++) tail[tl - 1 - b] = (uint8_t)(bits >> (8 * b));
    sha256_block(h, tail);
    if (tl == 128) sha256_block(h, tail + 64);
    for (int i = 0; i < 8; i++) {
        out[4 * i] = (uint8_t)(h[i] >> 24); out[4 * i + 1] = (uint8_t)(h[i] >> 16);
        out[4 * i + 2] = (uint8_t)(h[i] >> 8); out[4 * i + 3] = (uint8_t)h[i];
    }
}

/* ======================================================================== */
/* XXH3-64, 4..8-byte input branch == std.hash.XxHash3.hash(seed, 8 bytes)   */
/* (lasso_prover.zig:213,218,230,235).  Published algorithm: XXH3_len_4to8_64b. */
/* ======================================================================== */

uint64_t orc_xxh3_64(uint64_t seed, const uint8_t *data, size_t len) {
    /* default kSecret bytes 8..23 */
    static const uint8_t sec[16] = {0x7c, 0x01, 0x81, 0x2c, 0xf7, 0x21, 0xad, 0x1c,
                                    0xde, 0xd4, 0x6d, 0xe9, 0x83, 0x90, 0x97, 0xdb};
    uint32_t s32 = (uint32_t)seed;
    uint32_t sw = ((s32 & 0xff) << 24) | ((s32 & 0xff00) << 8) | ((s32 >> 8) & 0xff00) | (s32 >> 24);
    seed ^= (uint64_t)sw << 32;
    uint32_t in1 = rd32(data);
    uint32_t in2 = rd32(data + len - 4);
    uint64_t bitflip = (rd64(sec) ^ rd64(sec + 8)) - seed;
    uint64_t in64 = (uint64_t)in2 + ((uint64_t)in1 << 32);
    uint64_t h = in64 ^ bitflip;
    /* XXH3_rrmxmx */
    h ^= rotl64(h, 49) ^ rotl64(h, 24);
    h *= 0x9FB21C651E98DF25ull;
    h ^= (h >> 35) + (uint64_t)len;
    h *= 0x9FB21C651E98DF25ull;
    return h ^ (h >> 28);
}

/* ======================================================================== */
/* Fiat-Shamir transcript: src/core/hash.zig:255-324                         */
/* ======================================================================== */

struct orc_transcript {
    sha3_ctx h;
};

orc_transcript *orc_tr_new(void) {
    orc_transcript *t = (orc_transcript *)malloc(sizeof(*t));
    if (t) sha3_init(&t->h);
    return t;
}
void orc_tr_free(orc_transcript *t) { free(t); }

void orc_tr_append_bytes(orc_transcript *t, const uint8_t *data, size_t len) { /* :293-295 */
    sha3_update(&t->h, data, len);
}

void orc_tr_append_field(orc_transcript *t, uint64_t v) { /* :279-283 */
    uint8_t b[8];
    le64(v, b);
    sha3_update(&t->h, b, 8);
}

uint64_t orc_digest_to_field(uint64_t p, const uint8_t digest[32]) { /* :228-242, T = u64 => 8 bytes */
    return rd64(digest) % p;
}

uint64_t orc_tr_challenge(orc_transcript *t, uint64_t p) { /* :301-316 */
    uint8_t digest[32];
    sha3_ctx copy = t->h;       /* clone, finalize the clone */
    sha3_final(&copy, digest);
    uint64_t r = orc_digest_to_field(p, digest);
    sha3_update(&t->h, digest, 32); /* absorb the digest into the live sponge */
    return r;
}

void orc_tr_finalize(orc_transcript *t, uint8_t out[32]) { sha3_final(&t->h, out); } /* :319-323 */

static void tr_append_str(orc_transcript *t, const char *s) {
    orc_tr_append_bytes(t, (const uint8_t *)s, strlen(s));
}

/* ======================================================================== */
/* Multilinear: src/poly/multilinear.zig                                     */
/* ======================================================================== */

static int is_pow2(size_t n) { return n > 0 && (n & (n - 1)) == 0; }
static size_t log2_floor(size_t n) { size_t l = 0; while (n > 1) { n >>= 1; l++; } return l; }

size_t orc_log2_ceil(size_t n) { /* std.math.log2_int_ceil */
    size_t l = log2_floor(n);
    return ((size_t)1 << l) == n ? l : l + 1;
}

size_t orc_ceil_pow2(size_t n) { /* std.math.ceilPowerOfTwo */
    size_t v = 1;
    while (v < n) v <<= 1;
    return v;
}

int orc_mle_check(size_t n) { /* :36-44 */
    if (n == 0) return ORC_ERR_EMPTY_EVALUATIONS;
    if (!is_pow2(n)) return ORC_ERR_LENGTH_NOT_POWER_OF_TWO;
    return ORC_OK;
}

/* eval, :110-144 -- point[0] is bound to the LEAST significant index bit */
int orc_mle_eval(uint64_t p, const uint64_t *ev, size_t n, const uint64_t *pt, size_t npt, uint64_t *out) {
    int rc = orc_mle_check(n);
    if (rc) return rc;
    size_t nv = log2_floor(n);
    if (npt != nv) return ORC_ERR_WRONG_NUMBER_OF_VARIABLES;
    uint64_t (*basis)[2] = (uint64_t (*)[2])malloc(sizeof(uint64_t[2]) * (nv ? nv : 1));
    if (!basis) return ORC_ERR_OUT_OF_MEMORY;
    for (size_t i = 0; i < nv; i++) {
        basis[i][0] = orc_f_sub(p, 1 % p, pt[i]);
        basis[i][1] = pt[i];
    }
    uint64_t result = 0;
    for (size_t idx = 0; idx < n; idx++) {
        uint64_t term = ev[idx];
        size_t index = idx;
        for (size_t v = 0; v < nv; v++) {
            term = orc_f_mul(p, term, basis[v][index & 1]);
            index >>= 1;
        }
        result = orc_f_add(p, result, term);
    }
    free(basis);
    *out = result;
    return ORC_OK;
}

/* partialEval, :154-180 -- binds the MOST significant index bit */
int orc_mle_partial_eval(uint64_t p, const uint64_t *ev, size_t n, uint64_t r, uint64_t *out) {
    int rc = orc_mle_check(n);
    if (rc) return rc;
    if (n == 1) return ORC_ERR_NO_VARIABLES_TO_FIX;
    size_t half = n / 2;
    for (size_t i = 0; i < half; i++) {
        uint64_t one_minus_r = orc_f_sub(p, 1 % p, r);
        out[i] = orc_f_add(p, orc_f_mul(p, one_minus_r, ev[i]), orc_f_mul(p, r, ev[i + half]));
    }
    return ORC_OK;
}

uint64_t orc_mle_sum(uint64_t p, const uint64_t *ev, size_t n) { /* :188-194 */
    uint64_t s = 0;
    for (size_t i = 0; i < n; i++) s = orc_f_add(p, s, ev[i]);
    return s;
}

int orc_mle_round_poly(uint64_t p, const uint64_t *ev, size_t n, uint64_t out[2]) { /* :205-232 */
    int rc = orc_mle_check(n);
    if (rc) return rc;
    if (n == 1) return ORC_ERR_NO_VARIABLES;
    size_t half = n / 2;
    uint64_t s0 = 0, s1 = 0;
    for (size_t i = 0; i < half; i++) {
        s0 = orc_f_add(p, s0, ev[i]);
        s1 = orc_f_add(p, s1, ev[i + half]);
    }
    out[0] = s0;
    out[1] = orc_f_sub(p, s1, s0);
    return ORC_OK;
}

/* ======================================================================== */
/* Sumcheck: src/proofs/sumcheck_protocol.zig, sumcheck_prover.zig           */
/* ======================================================================== */

uint64_t orc_eval_univariate(uint64_t p, const uint64_t *c, size_t n, uint64_t x) { /* protocol:113-123 */
    if (n == 0) return 0;
    uint64_t result = c[n - 1];
    for (size_t i = n - 1; i > 0; i--) result = orc_f_add(p, orc_f_mul(p, result, x), c[i - 1]);
    return result;
}

static int sumcheck_core(uint64_t p, const uint64_t *ev, size_t n, const uint64_t *fixed_challenges,
                         uint64_t *rounds, uint64_t *point, uint64_t *final_eval) {
    size_t nv = log2_floor(n);
    uint64_t *cur = (uint64_t *)malloc(n * sizeof(uint64_t));
    uint64_t *nxt = (uint64_t *)malloc((n / 2) * sizeof(uint64_t));
    orc_transcript *tr = orc_tr_new(); /* fresh transcript per sumcheck, protocol:161 */
    if (!cur || !nxt || !tr) { free(cur); free(nxt); orc_tr_free(tr); return ORC_ERR_OUT_OF_MEMORY; }
    memcpy(cur, ev, n * sizeof(uint64_t));
    /* claimed sum is computed (prover:39) but neither absorbed nor stored in the proof */
    (void)orc_mle_sum(p, ev, n);
    size_t len = n;
    for (size_t round = 0; round < nv; round++) {
        uint64_t coeffs[2];
        orc_mle_round_poly(p, cur, len, coeffs); /* prover:52 */
        rounds[2 * round] = coeffs[0];
        rounds[2 * round + 1] = coeffs[1];
        uint64_t ch;
        if (fixed_challenges) {
            ch = fixed_challenges[round]; /* proveInteractive, prover:129 */
        } else {
            orc_tr_append_field(tr, coeffs[0]); /* generateChallenge, protocol:176-184 */
            orc_tr_append_field(tr, coeffs[1]);
            ch = orc_tr_challenge(tr, p);
        }
        point[round] = ch;
        orc_mle_partial_eval(p, cur, len, ch, nxt); /* prover:74 */
        len /= 2;
        uint64_t *tmp = cur; cur = nxt; nxt = tmp;
    }
    *final_eval = cur[0];
    free(cur); free(nxt); orc_tr_free(tr);
    return ORC_OK;
}

int orc_sumcheck_prove(uint64_t p, const uint64_t *ev, size_t n, uint64_t *rounds, uint64_t *point,
                       uint64_t *final_eval) { /* prover:26-91 */
    int rc = orc_mle_check(n);
    if (rc) return rc;
    if (n == 1) return ORC_ERR_NO_VARIABLES;
    return sumcheck_core(p, ev, n, NULL, rounds, point, final_eval);
}

int orc_sumcheck_prove_interactive(uint64_t p, const uint64_t *ev, size_t n, const uint64_t *challenges,
                                   size_t n_challenges, uint64_t *rounds, uint64_t *point,
                                   uint64_t *final_eval) { /* prover:97-144 */
    int rc = orc_mle_check(n);
    if (rc) return rc;
    if (n == 1) return ORC_ERR_NO_VARIABLES;
    if (n_challenges != log2_floor(n)) return ORC_ERR_WRONG_NUMBER_OF_CHALLENGES;
    return sumcheck_core(p, ev, n, challenges, rounds, point, final_eval);
}

size_t orc_sumcheck_to_bytes(size_t nv, const uint64_t *rounds, const uint64_t *point,
                             uint64_t final_eval, uint8_t *out) { /* protocol:76-107 */
    size_t off = 0;
    le64((uint64_t)nv, out + off); off += 8;
    for (size_t i = 0; i < 2 * nv; i++) { le64(rounds[i], out + off); off += 8; }
    for (size_t i = 0; i < nv; i++) { le64(point[i], out + off); off += 8; }
    le64(final_eval, out + off); off += 8;
    return off;
}

/* SumcheckVerifier.verify with oracle = poly.eval(final_point): sumcheck_verifier.zig:48-108.
 * Because partialEval binds MSB-first while eval is LSB-first this rejects
 * honest proofs in general; the oracle reproduces that behaviour rather than "fixing" it. */
int orc_sumcheck_verify(uint64_t p, const uint64_t *ev, size_t n, uint64_t claimed_sum,
                        const uint64_t *rounds, const uint64_t *point, uint64_t final_eval) {
    size_t nv = log2_floor(n);
    orc_transcript *tr = orc_tr_new();
    uint64_t claim = claimed_sum;
    for (size_t round = 0; round < nv; round++) {
        const uint64_t *c = rounds + 2 * round;
        uint64_t e0 = orc_eval_univariate(p, c, 2, 0);
        uint64_t e1 = orc_eval_univariate(p, c, 2, 1 % p);
        if (orc_f_add(p, e0, e1) != claim) { orc_tr_free(tr); return 0; }
        orc_tr_append_field(tr, c[0]);
        orc_tr_append_field(tr, c[1]);
        uint64_t ch = orc_tr_challenge(tr, p);
        claim = orc_eval_univariate(p, c, 2, ch);
    }
    orc_tr_free(tr);
    uint64_t oracle_eval = 0;
    if (orc_mle_eval(p, ev, n, point, nv, &oracle_eval)) return 0;
    return oracle_eval == claim && oracle_eval == final_eval;
}

/* ======================================================================== */
/* SimpleMerkleTree(F, SHA3Hasher): src/commitments/merkle_tree.zig:273-401  */
/* ======================================================================== */

static uint8_t *merkle_leaf_hashes(const uint64_t *values, size_t n, size_t npad) { /* :294-306 */
    uint8_t *leaves = (uint8_t *)malloc(npad * 32);
    if (!leaves) return NULL;
    for (size_t i = 0; i < n; i++) orc_hash_leaf(values[i], leaves + 32 * i);
    uint8_t zero_hash[32];
    orc_hash_leaf(0, zero_hash);
    for (size_t i = n; i < npad; i++) memcpy(leaves + 32 * i, zero_hash, 32);
    return leaves;
}

int orc_merkle_build(const uint64_t *values, size_t n, uint8_t root[32], size_t *height) { /* :283-318 */
    if (n == 0) return ORC_ERR_EMPTY_VALUES;
    if (n > ((size_t)1 << 62)) return ORC_ERR_TOO_MANY_VALUES;
    size_t npad = orc_ceil_pow2(n);
    if (height) *height = log2_floor(npad);
    uint8_t *cur = merkle_leaf_hashes(values, n, npad);
    if (!cur) return ORC_ERR_OUT_OF_MEMORY;
    size_t len = npad; /* computeRoot, :380-400 */
    while (len > 1) {
        size_t next = len / 2;
        for (size_t i = 0; i < next; i++) {
            uint8_t tmp[32];
            orc_hash_internal(cur + 64 * i, cur + 64 * i + 32, tmp);
            memcpy(cur + 32 * i, tmp, 32); /* in place is safe: writes index i <= 2i */
        }
        len = next;
    }
    memcpy(root, cur, 32);
    free(cur);
    return ORC_OK;
}

int orc_merkle_open(const uint64_t *values, size_t n, size_t index, uint8_t *siblings, uint8_t *dirs,
                    uint64_t *leaf_value) { /* :324-360 -- rebuilds every level */
    if (n == 0) return ORC_ERR_EMPTY_VALUES;
    if (index >= n) return ORC_ERR_INDEX_OUT_OF_BOUNDS; /* :325 checks values.len, not padded len */
    size_t npad = orc_ceil_pow2(n);
    size_t height = log2_floor(npad);
    uint8_t *cur = merkle_leaf_hashes(values, n, npad);
    if (!cur) return ORC_ERR_OUT_OF_MEMORY;
    size_t len = npad, ci = index;
    for (size_t level = 0; level < height; level++) {
        int is_right = (ci % 2) == 1;
        size_t sib = is_right ? ci - 1 : ci + 1;
        memcpy(siblings + 32 * level, cur + 32 * sib, 32);
        dirs[level] = (uint8_t)is_right;
        size_t next = len / 2;
        for (size_t i = 0; i < next; i++) {
            uint8_t tmp[32];
            orc_hash_internal(cur + 64 * i, cur + 64 * i + 32, tmp);
            memcpy(cur + 32 * i, tmp, 32);
        }
        len = next;
        ci /= 2;
    }
    *leaf_value = values[index];
    free(cur);
    return ORC_OK;
}

int orc_merkle_verify(const uint8_t root[32], uint64_t value, const uint8_t *siblings,
                      const uint8_t *dirs, size_t height) { /* :362-373 */
    uint8_t cur[32], tmp[32];
    orc_hash_leaf(value, cur);
    for (size_t l = 0; l < height; l++) {
        if (dirs[l]) orc_hash_internal(siblings + 32 * l, cur, tmp);
        else orc_hash_internal(cur, siblings + 32 * l, tmp);
        memcpy(cur, tmp, 32);
    }
    return memcmp(cur, root, 32) == 0;
}

/* All levels, bottom-up, concatenated: level 0 (npad nodes), level 1 (npad/2) ... root.  Not a
 * reference routine: the same tree, kept instead of recomputed (used for large-size tests/timing). */
int orc_merkle_levels(const uint64_t *values, size_t n, uint8_t *levels, size_t *height) {
    if (n == 0) return ORC_ERR_EMPTY_VALUES;
    size_t npad = orc_ceil_pow2(n);
    size_t h = log2_floor(npad);
    if (height) *height = h;
    for (size_t i = 0; i < n; i++) orc_hash_leaf(values[i], levels + 32 * i);
    uint8_t zero_hash[32];
    orc_hash_leaf(0, zero_hash);
    for (size_t i = n; i < npad; i++) memcpy(levels + 32 * i, zero_hash, 32);
    uint8_t *cur = levels;
    size_t len = npad;
    while (len > 1) {
        uint8_t *nxt = cur + 32 * len;
        for (size_t i = 0; i < len / 2; i++) orc_hash_internal(cur + 64 * i, cur + 64 * i + 32, nxt + 32 * i);
        cur = nxt;
        len /= 2;
    }
    return ORC_OK;
}

/* ======================================================================== */
/* CommitmentScheme: src/commitments/polynomial_commit.zig                   */
/* ======================================================================== */

size_t orc_point_to_index(const uint64_t *pt, size_t npt) { /* :178-183 */
    if (npt == 0) return 0;
    return (size_t)(pt[0] % ((uint64_t)1 << npt));
}

int orc_commit_open(uint64_t p, const uint64_t *ev, size_t n, const uint64_t *pt, size_t npt,
                    uint64_t *value, uint64_t *index, uint8_t *siblings, uint8_t *dirs,
                    uint64_t *leaf_value) { /* :86-115 */
    int rc = orc_mle_check(n);
    if (rc) return rc;
    if (npt != log2_floor(n)) return ORC_ERR_POINT_DIMENSION_MISMATCH;
    rc = orc_mle_eval(p, ev, n, pt, npt, value); /* :97 (second naive eval in prove) */
    if (rc) return rc;
    size_t idx = orc_point_to_index(pt, npt);
    *index = idx;
    return orc_merkle_open(ev, n, idx, siblings, dirs, leaf_value);
}

/* ======================================================================== */
/* Lasso (simplified): src/lookups/lasso_prover.zig                          */
/* ======================================================================== */

uint64_t orc_lasso_hash_row(uint64_t p, const uint64_t *fields, size_t n_fields) { /* :208-239 */
    uint64_t h = 0;
    for (size_t i = 0; i < n_fields; i++) {
        h ^= fields[i];
        uint8_t b[8];
        le64(h, b);
        h = orc_xxh3_64(0, b, 8);
    }
    return h % p;
}

void orc_lasso_commit(const uint64_t *ev, size_t n, uint8_t out[32]) { /* :242-252 one flat sponge */
    sha3_ctx c;
    sha3_init(&c);
    for (size_t i = 0; i < n; i++) {
        uint8_t b[8];
        le64(ev[i], b);
        sha3_update(&c, b, 8);
    }
    sha3_final(&c, out);
}

int orc_lasso_prove(uint64_t p, const uint64_t *table, size_t table_rows, const uint64_t *queries,
                    size_t n_queries, size_t n_in, size_t n_out, size_t *nv_out, uint64_t *rounds,
                    uint64_t *point, uint64_t *final_eval, uint8_t query_commit[32],
                    uint8_t table_commit[32]) { /* :103-173 */
    if (n_queries == 0) return ORC_ERR_NO_QUERIES;
    size_t w = n_in + n_out;
    int rc = orc_mle_check(table_rows); /* Multilinear.init(table_evals), :124 */
    if (rc) return rc;
    uint64_t *tev = (uint64_t *)malloc(table_rows * sizeof(uint64_t));
    size_t padded = orc_ceil_pow2(n_queries);
    uint64_t *qev = (uint64_t *)malloc(padded * sizeof(uint64_t));
    if (!tev || !qev) { free(tev); free(qev); return ORC_ERR_OUT_OF_MEMORY; }
    for (size_t i = 0; i < table_rows; i++) tev[i] = orc_lasso_hash_row(p, table + i * w, w);
    for (size_t j = 0; j < n_queries; j++) qev[j] = orc_lasso_hash_row(p, queries + j * w, w);
    for (size_t j = n_queries; j < padded; j++) qev[j] = 0;
    *nv_out = log2_floor(padded);
    rc = orc_sumcheck_prove(p, qev, padded, rounds, point, final_eval); /* :160, NoVariables if 1 query */
    if (rc == ORC_OK) {
        orc_lasso_commit(qev, padded, query_commit);
        orc_lasso_commit(tev, table_rows, table_commit);
    }
    free(tev); free(qev);
    return rc;
}

int orc_lasso_prove_with_mapping(uint64_t p, const uint64_t *table, size_t table_rows,
                                 const uint64_t *queries, size_t n_queries, size_t n_in, size_t n_out,
                                 const uint64_t *mapping, size_t n_mapping, size_t *nv_out,
                                 uint64_t *rounds, uint64_t *point, uint64_t *final_eval,
                                 uint8_t query_commit[32], uint8_t table_commit[32]) { /* :179-205 */
    if (n_queries != n_mapping) return ORC_ERR_MAPPING_LENGTH_MISMATCH;
    size_t w = n_in + n_out;
    for (size_t j = 0; j < n_queries; j++) {
        if (mapping[j] >= table_rows) return ORC_ERR_INVALID_MAPPING;
        if (memcmp(queries + j * w, table + mapping[j] * w, w * sizeof(uint64_t)) != 0)
            return ORC_ERR_QUERY_TABLE_MISMATCH;
    }
    return orc_lasso_prove(p, table, table_rows, queries, n_queries, n_in, n_out, nv_out, rounds, point,
                           final_eval, query_commit, table_commit);
}

void orc_build_table(uint64_t p, int kind, size_t bits, uint64_t *out) { /* table_builder.zig:126-213 */
    uint64_t max_val = (uint64_t)1 << bits;
    size_t idx = 0;
    for (uint64_t a = 0; a < max_val; a++)
        for (uint64_t b = 0; b < max_val; b++) {
            uint64_t r = kind == 0 ? (a + b) % max_val : kind == 1 ? (a ^ b) : (a & b);
            out[idx++] = a % p; out[idx++] = b % p; out[idx++] = r % p;
        }
}

/* ======================================================================== */
/* VM: src/vm/state.zig, src/vm/memory.zig, src/isa/rv64i.zig                */
/* ======================================================================== */

/* sparse byte memory (memory.zig): open-addressing hash map addr -> byte; unmapped reads 0 */
typedef struct {
    uint64_t *keys;
    uint8_t *vals;
    uint8_t *used;
    size_t cap, count;
} orc_mem;

static int mem_init(orc_mem *m, size_t cap) {
    m->cap = cap; m->count = 0;
    m->keys = (uint64_t *)calloc(cap, sizeof(uint64_t));
    m->vals = (uint8_t *)calloc(cap, 1);
    m->used = (uint8_t *)calloc(cap, 1);
    return (m->keys && m->vals && m->used) ? 0 : 1;
}
static void mem_free(orc_mem *m) { free(m->keys); free(m->vals); free(m->used); }
static size_t mem_slot(const orc_mem *m, uint64_t a) {
    uint64_t h = a * 0x9E3779B97F4A7C15ull;
    size_t i = (size_t)(h >> 20) & (m->cap - 1);
    while (m->used[i] && m->keys[i] != a) i = (i + 1) & (m->cap - 1);
    return i;
}
static uint8_t mem_lb(const orc_mem *m, uint64_t a) {
    size_t i = mem_slot(m, a);
    return m->used[i] ? m->vals[i] : 0;
}
static int mem_grow(orc_mem *m);
static int mem_sb(orc_mem *m, uint64_t a, uint8_t v) {
    /* storing 0 removes the key in the reference (memory.zig:41-47); a stored 0 reads back 0 either way */
    size_t i = mem_slot(m, a);
    if (!m->used[i]) {
        if ((m->count + 1) * 2 > m->cap) { if (mem_grow(m)) return 1; i = mem_slot(m, a); }
        m->used[i] = 1; m->keys[i] = a; m->count++;
    }
    m->vals[i] = v;
    return 0;
}
static int mem_grow(orc_mem *m) {
    orc_mem n;
    if (mem_init(&n, m->cap * 2)) return 1;
    for (size_t i = 0; i < m->cap; i++)
        if (m->used[i]) { size_t j = mem_slot(&n, m->keys[i]); n.used[j] = 1; n.keys[j] = m->keys[i]; n.vals[j] = m->vals[i]; n.count++; }
    mem_free(m);
    *m = n;
    return 0;
}
static uint64_t mem_load(const orc_mem *m, uint64_t a, int bytes) { /* little endian, memory.zig:50-86 */
    uint64_t v = 0;
    for (int b = 0; b < bytes; b++) v |= (uint64_t)mem_lb(m, a + (uint64_t)b) << (8 * b);
    return v;
}
static int mem_store(orc_mem *m, uint64_t a, uint64_t v, int bytes) {
    for (int b = 0; b < bytes; b++)
        if (mem_sb(m, a + (uint64_t)b, (uint8_t)(v >> (8 * b)))) return 1;
    return 0;
}

typedef struct {
    uint8_t opcode, rd, funct3, rs1, rs2, funct7;
    int64_t imm;
} orc_inst;

enum { OP_LOAD = 0x03, OP_LOAD_FP = 0x07, OP_MISC_MEM = 0x0f, OP_OP_IMM = 0x13, OP_AUIPC = 0x17,
       OP_OP_IMM_32 = 0x1b, OP_STORE = 0x23, OP_STORE_FP = 0x27, OP_AMO = 0x2f, OP_OP = 0x33,
       OP_LUI = 0x37, OP_OP_32 = 0x3b, OP_MADD = 0x43, OP_MSUB = 0x47, OP_NMSUB = 0x4b,
       OP_NMADD = 0x4f, OP_OP_FP = 0x53, OP_BRANCH = 0x63, OP_JALR = 0x67, OP_JAL = 0x6f,
       OP_SYSTEM = 0x73 };

static int64_t sext(uint32_t v, int bits) { /* sign-extend the low `bits` bits */
    uint32_t m = 1u << (bits - 1);
    return (int64_t)(int32_t)((v ^ m) - m);
}

/* Instruction.decode, rv64i.zig:124-233; returns 1 on InvalidInstruction (opcode bits == 0) */
static int decode(uint32_t w, orc_inst *in) {
    uint8_t op = w & 0x7f;
    if (op == 0) return 1;
    in->opcode = op;
    in->rd = (w >> 7) & 0x1f;
    in->funct3 = (w >> 12) & 7;
    in->rs1 = (w >> 15) & 0x1f;
    in->rs2 = (w >> 20) & 0x1f;
    in->funct7 = (w >> 25) & 0x7f;
    switch (op) { /* instructionFormat, rv64i.zig:61-73 */
    case OP_OP_IMM: case OP_OP_IMM_32: case OP_JALR: case OP_LOAD: case OP_LOAD_FP: case OP_MISC_MEM: case OP_SYSTEM:
        in->imm = sext((w >> 20) & 0xfff, 12); break;                                   /* I */
    case OP_STORE: case OP_STORE_FP:
        in->imm = sext((((w >> 25) & 0x7f) << 5) | ((w >> 7) & 0x1f), 12); break;       /* S */
    case OP_BRANCH:
        in->imm = sext((((w >> 31) & 1) << 12) | (((w >> 7) & 1) << 11) | (((w >> 25) & 0x3f) << 5) |
                       (((w >> 8) & 0xf) << 1), 13); break;                             /* B */
    case OP_LUI: case OP_AUIPC:
        in->imm = (int64_t)(int32_t)(w & 0xfffff000u); break;                           /* U */
    case OP_JAL:
        in->imm = sext((((w >> 31) & 1) << 20) | (((w >> 12) & 0xff) << 12) | (((w >> 20) & 1) << 11) |
                       (((w >> 21) & 0x3ff) << 1), 21); break;                          /* J */
    default:
        in->imm = 0; break; /* R-type and unknown opcodes, rv64i.zig:71,231 */
    }
    return 0;
}

/* getTableMetadata(inst) != null, instruction_table.zig:243-274 */
static int is_lookup(const orc_inst *in) {
    return in->opcode == OP_OP || in->opcode == OP_OP_IMM || in->opcode == OP_LOAD ||
           in->opcode == OP_STORE || in->opcode == OP_BRANCH;
}

typedef struct {
    uint64_t pc;
    uint64_t regs[32];
    orc_mem mem;
    int halted;
    const uint64_t *input; size_t n_input, input_pos;
} orc_vm;

static uint64_t rr(const orc_vm *vm, unsigned r) { return r == 0 ? 0 : vm->regs[r]; } /* registers.zig:38-48 */
static void wr(orc_vm *vm, unsigned r, uint64_t v) { if (r != 0) vm->regs[r] = v; }

static int trace_reserve(orc_trace *t, size_t need) {
    if (need <= t->capacity) return 0;
    size_t cap = t->capacity ? t->capacity : 1024;
    while (cap < need) cap *= 2;
#define GROW(field, type, mult) do { \
        type *np_ = (type *)realloc(t->field, cap * (mult) * sizeof(type)); \
        if (!np_) { return 1; } \
        t->field = np_; \
    } while (0)
    GROW(pc, uint64_t, 1); GROW(regs_after, uint64_t, 32);
    GROW(opcode, uint8_t, 1); GROW(rd, uint8_t, 1); GROW(rs1, uint8_t, 1); GROW(rs2, uint8_t, 1);
    GROW(funct3, uint8_t, 1); GROW(funct7, uint8_t, 1); GROW(imm, int64_t, 1);
    GROW(mem_kind, uint8_t, 1); GROW(mem_addr, uint64_t, 1); GROW(mem_value, uint64_t, 1);
    GROW(is_lookup, uint8_t, 1);
#undef GROW
    t->capacity = cap;
    return 0;
}

orc_trace *orc_trace_new(void) { return (orc_trace *)calloc(1, sizeof(orc_trace)); }
void orc_trace_free(orc_trace *t) {
    if (!t) return;
    free(t->pc); free(t->regs_after); free(t->opcode); free(t->rd); free(t->rs1); free(t->rs2);
    free(t->funct3); free(t->funct7); free(t->imm); free(t->mem_kind); free(t->mem_addr);
    free(t->mem_value); free(t->is_lookup); free(t->outputs); free(t);
}

static uint64_t mulhu64(uint64_t a, uint64_t b) { return (uint64_t)(((u128)a * (u128)b) >> 64); }

/* execute(), state.zig:188-597.  Returns ORC_OK with *next_pc, or an error code. */
static int vm_execute(orc_vm *vm, const orc_inst *in, uint64_t *next_pc, uint8_t *mem_kind,
                      uint64_t *mem_addr, uint64_t *mem_value, orc_trace *tr) {
    uint64_t pc = vm->pc;
    uint64_t a = rr(vm, in->rs1), b = rr(vm, in->rs2);
    uint64_t imm = (uint64_t)in->imm;
    switch (in->opcode) {
    case OP_OP: { /* :221-317 */
        uint64_t r;
        if (in->funct7 == 1) { /* RV64M */
            int64_t sa = (int64_t)a, sb = (int64_t)b;
            switch (in->funct3) {
            case 0: r = a * b; break;
            case 1: r = (uint64_t)(int64_t)(((__int128)sa * (__int128)sb) >> 64); break;
            case 2: r = (uint64_t)(int64_t)(((__int128)sa * (__int128)(u128)b) >> 64); break;
            case 3: r = mulhu64(a, b); break;
            case 4: r = (sb == 0) ? ~0ull : (sa == INT64_MIN && sb == -1) ? a : (uint64_t)(sa / sb); break;
            case 5: r = (b == 0) ? ~0ull : a / b; break;
            case 6: r = (sb == 0) ? a : (sa == INT64_MIN && sb == -1) ? 0 : (uint64_t)(sa % sb); break;
            default: r = (b == 0) ? a : a % b; break;
            }
        } else {
            unsigned sh = (unsigned)(b & 0x3f);
            switch (in->funct3) {
            case 0: r = (in->funct7 == 0x20) ? a - b : a + b; break;
            case 1: r = a << sh; break;
            case 2: r = ((int64_t)a < (int64_t)b); break;
            case 3: r = (a < b); break;
            case 4: r = a ^ b; break;
            case 5: r = (in->funct7 == 0x20) ? (uint64_t)((int64_t)a >> sh) : a >> sh; break;
            case 6: r = a | b; break;
            default: r = a & b; break;
            }
        }
        wr(vm, in->rd, r);
        *next_pc = pc + 4;
        return ORC_OK;
    }
    case OP_OP_32: { /* :319-397 */
        uint32_t x = (uint32_t)a, y = (uint32_t)b, r32;
        if (in->funct7 == 1) {
            int32_t sx = (int32_t)x, sy = (int32_t)y;
            switch (in->funct3) {
            case 0: r32 = x * y; break;
            case 4: r32 = (sy == 0) ? 0xffffffffu : (sx == INT32_MIN && sy == -1) ? x : (uint32_t)(sx / sy); break;
            case 5: r32 = (y == 0) ? 0xffffffffu : x / y; break;
            case 6: r32 = (sy == 0) ? x : (sx == INT32_MIN && sy == -1) ? 0 : (uint32_t)(sx % sy); break;
            case 7: r32 = (y == 0) ? x : x % y; break;
            default: return ORC_ERR_INVALID_OP32;
            }
        } else {
            unsigned sh = y & 0x1f;
            switch (in->funct3) {
            case 0: r32 = (in->funct7 == 0x20) ? x - y : x + y; break;
            case 1: r32 = x << sh; break;
            case 5: r32 = (in->funct7 == 0x20) ? (uint32_t)((int32_t)x >> sh) : x >> sh; break;
            default: return ORC_ERR_INVALID_OP32;
            }
        }
        wr(vm, in->rd, (uint64_t)(int64_t)(int32_t)r32);
        *next_pc = pc + 4;
        return ORC_OK;
    }
    case OP_OP_IMM: { /* :399-425 */
        uint64_t r;
        unsigned sh = (unsigned)(imm & 0x3f);
        switch (in->funct3) {
        case 0: r = a + imm; break;
        case 1: r = a << sh; break;
        case 2: r = ((int64_t)a < in->imm); break;
        case 3: r = (a < imm); break;
        case 4: r = a ^ imm; break;
        case 5: r = (in->funct7 == 0x20) ? (uint64_t)((int64_t)a >> sh) : a >> sh; break;
        case 6: r = a | imm; break;
        default: r = a & imm; break;
        }
        wr(vm, in->rd, r);
        *next_pc = pc + 4;
        return ORC_OK;
    }
    case OP_OP_IMM_32: { /* :427-450 */
        uint32_t x = (uint32_t)a, r32;
        unsigned sh = (unsigned)(imm & 0x1f);
        switch (in->funct3) {
        case 0: r32 = x + (uint32_t)imm; break;
        case 1: r32 = x << sh; break;
        case 5: r32 = (in->funct7 == 0x20) ? (uint32_t)((int32_t)x >> sh) : x >> sh; break;
        default: return ORC_ERR_INVALID_OP32;
        }
        wr(vm, in->rd, (uint64_t)(int64_t)(int32_t)r32);
        *next_pc = pc + 4;
        return ORC_OK;
    }
    case OP_LOAD: { /* :452-482 */
        uint64_t addr = a + imm, r;
        switch (in->funct3) {
        case 0: r = (uint64_t)(int64_t)(int8_t)mem_load(&vm->mem, addr, 1); break;
        case 1: r = (uint64_t)(int64_t)(int16_t)mem_load(&vm->mem, addr, 2); break;
        case 2: r = (uint64_t)(int64_t)(int32_t)mem_load(&vm->mem, addr, 4); break;
        case 3: r = mem_load(&vm->mem, addr, 8); break;
        case 4: r = mem_load(&vm->mem, addr, 1); break;
        case 5: r = mem_load(&vm->mem, addr, 2); break;
        case 6: r = mem_load(&vm->mem, addr, 4); break;
        default: return ORC_ERR_INVALID_LOAD_FUNCT3;
        }
        *mem_kind = 1; *mem_addr = addr; *mem_value = r;
        wr(vm, in->rd, r);
        *next_pc = pc + 4;
        return ORC_OK;
    }
    case OP_STORE: { /* :484-507 */
        uint64_t addr = a + imm;
        int bytes;
        switch (in->funct3) {
        case 0: bytes = 1; break; case 1: bytes = 2; break; case 2: bytes = 4; break; case 3: bytes = 8; break;
        default: return ORC_ERR_INVALID_STORE_FUNCT3;
        }
        if (mem_store(&vm->mem, addr, b, bytes)) return ORC_ERR_OUT_OF_MEMORY;
        *mem_kind = 2; *mem_addr = addr; *mem_value = b; /* full rs2, :499-504 */
        *next_pc = pc + 4;
        return ORC_OK;
    }
    case OP_BRANCH: { /* :509-528 */
        int taken;
        switch (in->funct3) {
        case 0: taken = a == b; break;
        case 1: taken = a != b; break;
        case 4: taken = (int64_t)a < (int64_t)b; break;
        case 5: taken = (int64_t)a >= (int64_t)b; break;
        case 6: taken = a < b; break;
        case 7: taken = a >= b; break;
        default: return ORC_ERR_INVALID_BRANCH_FUNCT3;
        }
        *next_pc = taken ? pc + imm : pc + 4;
        return ORC_OK;
    }
    case OP_JAL: wr(vm, in->rd, pc + 4); *next_pc = pc + imm; return ORC_OK;          /* :530-536 */
    case OP_JALR: { uint64_t base = a; wr(vm, in->rd, pc + 4); *next_pc = (base + imm) & ~1ull; return ORC_OK; } /* :538-547 */
    case OP_LUI: wr(vm, in->rd, imm); *next_pc = pc + 4; return ORC_OK;                /* :549-555 */
    case OP_AUIPC: wr(vm, in->rd, pc + imm); *next_pc = pc + 4; return ORC_OK;         /* :557-562 */
    case OP_SYSTEM: /* :564-597 */
        if (in->funct3 == 0) {
            if (in->imm == 0) {
                uint64_t sc = rr(vm, 17);
                if (sc == 1) { /* ECALL_COMMIT */
                    if (tr->n_outputs == tr->cap_outputs) {
                        size_t nc = tr->cap_outputs ? tr->cap_outputs * 2 : 16;
                        uint64_t *np_ = (uint64_t *)realloc(tr->outputs, nc * sizeof(uint64_t));
                        if (!np_) return ORC_ERR_OUT_OF_MEMORY;
                        tr->outputs = np_; tr->cap_outputs = nc;
                    }
                    tr->outputs[tr->n_outputs++] = rr(vm, 10);
                } else if (sc == 2) { /* ECALL_READ */
                    if (vm->input_pos < vm->n_input) wr(vm, 10, vm->input[vm->input_pos++]);
                    else wr(vm, 10, 0);
                }
                *next_pc = pc + 4;
                return ORC_OK;
            } else if (in->imm == 1) { /* EBREAK */
                vm->halted = 1;
                *next_pc = pc;
                return ORC_OK;
            }
        }
        return ORC_ERR_UNIMPLEMENTED_SYSTEM;
    case OP_MISC_MEM: *next_pc = pc + 4; return ORC_OK; /* FENCE no-op, :202-205 */
    default: return ORC_ERR_UNIMPLEMENTED_INSTRUCTION;
    }
}

/* one VMState.step(), state.zig:128-167.  rc: 0 ok, -1 InvalidInstruction (halts, no step), >0 error */
static int vm_step(orc_vm *vm, orc_trace *t) {
    uint32_t w = (uint32_t)mem_load(&vm->mem, vm->pc, 4);
    orc_inst in;
    if (decode(w, &in)) { vm->halted = 1; return -1; }
    uint8_t mk = 0; uint64_t ma = 0, mv = 0, next_pc = 0;
    uint64_t pc_before = vm->pc;
    int rc = vm_execute(vm, &in, &next_pc, &mk, &ma, &mv, t);
    if (rc) return rc;
    if (trace_reserve(t, t->num_steps + 1)) return ORC_ERR_OUT_OF_MEMORY;
    size_t i = t->num_steps++;
    t->pc[i] = pc_before;
    for (int r = 0; r < 32; r++) t->regs_after[i * 32 + r] = rr(vm, (unsigned)r);
    t->opcode[i] = in.opcode; t->rd[i] = in.rd; t->rs1[i] = in.rs1; t->rs2[i] = in.rs2;
    t->funct3[i] = in.funct3; t->funct7[i] = in.funct7; t->imm[i] = in.imm;
    t->mem_kind[i] = mk; t->mem_addr[i] = ma; t->mem_value[i] = mv;
    t->is_lookup[i] = (uint8_t)is_lookup(&in);
    vm->pc = next_pc;
    return 0;
}

static int vm_setup(orc_vm *vm, const uint8_t *program, size_t len, uint64_t entry_pc,
                    const uint64_t *input, size_t n_input) {
    memset(vm, 0, sizeof(*vm));
    size_t cap = 1024;
    while (cap < len * 4) cap *= 2;
    if (mem_init(&vm->mem, cap)) return ORC_ERR_OUT_OF_MEMORY;
    for (size_t i = 0; i < len; i++) /* loadProgram: storeByte per byte */
        if (program[i] != 0 && mem_sb(&vm->mem, entry_pc + i, program[i])) return ORC_ERR_OUT_OF_MEMORY;
    vm->pc = entry_pc;
    vm->input = input; vm->n_input = n_input;
    return ORC_OK;
}

int orc_vm_run(const uint8_t *program, size_t program_len, uint64_t entry_pc, const uint64_t *initial_regs,
               size_t n_initial_regs, size_t max_steps, const uint64_t *input, size_t n_input,
               orc_trace *out) { /* prover.zig:117-142 */
    orc_vm vm;
    int rc = vm_setup(&vm, program, program_len, entry_pc, input, n_input);
    if (rc) { mem_free(&vm.mem); return rc; }
    for (size_t i = 0; i < n_initial_regs && i < 32; i++) wr(&vm, (unsigned)i, initial_regs[i]);
    size_t step_count = 0;
    rc = ORC_OK;
    while (!vm.halted && step_count < max_steps) {
        int s = vm_step(&vm, out);
        if (s == -1) break; /* InvalidInstruction: normal termination */
        if (s > 0) { rc = s; break; }
        step_count++;
    }
    out->final_pc = vm.pc;
    for (int r = 0; r < 32; r++) out->final_regs[r] = rr(&vm, (unsigned)r);
    out->halted = vm.halted;
    mem_free(&vm.mem);
    return rc;
}

int orc_vm_run_kat(const uint8_t *program, size_t program_len, uint64_t entry_pc, size_t max_steps,
                   uint64_t final_regs[32], uint64_t *final_pc, size_t *steps) { /* state.zig:172-184 */
    orc_trace *t = orc_trace_new();
    if (!t) return ORC_ERR_OUT_OF_MEMORY;
    int rc = orc_vm_run(program, program_len, entry_pc, NULL, 0, max_steps, NULL, 0, t);
    if (rc == ORC_OK && !t->halted && t->num_steps >= max_steps) rc = ORC_ERR_MAX_STEPS_EXCEEDED;
    memcpy(final_regs, t->final_regs, sizeof(t->final_regs));
    *final_pc = t->final_pc;
    *steps = t->num_steps;
    orc_trace_free(t);
    return rc;
}

/* ======================================================================== */
/* Witness: src/constraints/witness.zig:29-270; order prover.zig:376-390     */
/* ======================================================================== */

int orc_witness(uint64_t p, const orc_trace *t, uint64_t *cols, size_t *nv_out) {
    size_t ns = t->num_steps;
    size_t nv = ns == 0 ? 0 : orc_log2_ceil(ns);
    size_t N = (size_t)1 << nv;
    *nv_out = nv;
    if (ns == 0) { memset(cols, 0, 43 * N * sizeof(uint64_t)); return ORC_OK; }
    for (size_t i = 0; i < N; i++) {
        size_t s = i < ns ? i : ns - 1; /* pc and registers repeat the last value, :80-87,116-123 */
        cols[0 * N + i] = t->pc[s] % p;
        for (int r = 0; r < 32; r++) cols[(size_t)(1 + r) * N + i] = t->regs_after[s * 32 + (size_t)r] % p;
        int live = i < ns; /* everything else pads with zero, :174-182,249-253 */
        cols[33 * N + i] = live ? t->opcode[i] % p : 0;
        cols[34 * N + i] = live ? t->rd[i] % p : 0;
        cols[35 * N + i] = live ? t->rs1[i] % p : 0;
        cols[36 * N + i] = live ? t->rs2[i] % p : 0;
        cols[37 * N + i] = live ? t->funct3[i] % p : 0;
        cols[38 * N + i] = live ? t->funct7[i] % p : 0;
        cols[39 * N + i] = live ? ((uint64_t)t->imm[i]) % p : 0; /* u64(bitcast i64), :170 */
        int m = live && t->mem_kind[i] != 0;
        cols[40 * N + i] = m ? t->mem_addr[i] % p : 0;
        cols[41 * N + i] = m ? t->mem_value[i] % p : 0;
        cols[42 * N + i] = (m && t->mem_kind[i] == 1) ? 1 % p : 0; /* is_read = Load, :239 */
    }
    return ORC_OK;
}

/* ======================================================================== */
/* Prover.generateCommitments: src/prover/prover.zig:366-467                 */
/* ======================================================================== */

int orc_generate_commitments(uint64_t p, orc_transcript *t, const uint64_t *cols, size_t nv, uint8_t *roots,
                             uint64_t *points, uint64_t *values, uint64_t *indices, uint64_t *leaves,
                             uint8_t *siblings, uint8_t *dirs) {
    size_t N = (size_t)1 << nv;
    for (int c = 0; c < 43; c++) { /* phase 1, :405-410 */
        int rc = orc_merkle_build(cols + (size_t)c * N, N, roots + 32 * c, NULL);
        if (rc) return rc;
    }
    tr_append_str(t, "POLY_COMMITMENTS"); /* phase 2, :413-416 */
    for (int c = 0; c < 43; c++) orc_tr_append_bytes(t, roots + 32 * c, 32);
    for (int c = 0; c < 43; c++) { /* phase 3, :420-442 */
        uint64_t *pt = points + (size_t)c * nv;
        for (size_t j = 0; j < nv; j++) pt[j] = orc_tr_challenge(t, p);
        uint64_t v1 = 0, v2 = 0;
        int rc = orc_mle_eval(p, cols + (size_t)c * N, N, pt, nv, &v1); /* :427 */
        if (rc) return rc;
        rc = orc_commit_open(p, cols + (size_t)c * N, N, pt, nv, &v2, &indices[c],
                             siblings + (size_t)c * nv * 32, dirs + (size_t)c * nv, &leaves[c]); /* :431 */
        if (rc) return rc;
        values[c] = v1; /* v2 == v1: same function, same inputs */
        (void)v2;
    }
    tr_append_str(t, "OPENING_CLAIMS"); /* phase 4, :463-466 */
    for (int c = 0; c < 43; c++) orc_tr_append_field(t, values[c]);
    return ORC_OK;
}

int orc_commit_column_literal(uint64_t p, const uint64_t *col, size_t nv, const uint64_t *point, uint8_t root[32],
                              uint64_t *value, uint64_t *index, uint8_t *siblings, uint8_t *dirs, uint64_t *leaf) {
    size_t N = (size_t)1 << nv;
    int rc = orc_merkle_build(col, N, root, NULL); /* prover.zig:406 */
    if (rc) return rc;
    uint64_t v1 = 0, v2 = 0;
    rc = orc_mle_eval(p, col, N, point, nv, &v1); /* prover.zig:427 */
    if (rc) return rc;
    rc = orc_commit_open(p, col, N, point, nv, &v2, index, siblings, dirs, leaf); /* prover.zig:431 */
    if (rc) return rc;
    *value = v1;
    return v1 == v2 ? ORC_OK : ORC_ERR_PROTOCOL_ERROR;
}

/* eval by MSB-first folds with the point reversed: exact field arithmetic => the value of :110-144 */
static uint64_t eval_by_folds(uint64_t p, const uint64_t *ev, size_t N, const uint64_t *pt, size_t nv) {
    uint64_t *cur = (uint64_t *)malloc(N * sizeof(uint64_t));
    memcpy(cur, ev, N * sizeof(uint64_t));
    size_t len = N;
    for (size_t k = 0; k < nv; k++) {
        uint64_t r = pt[nv - 1 - k];
        size_t half = len / 2;
        for (size_t i = 0; i < half; i++) {
            uint64_t d = orc_f_sub(p, cur[i + half], cur[i]);
            cur[i] = orc_f_add(p, cur[i], orc_f_mul(p, r, d));
        }
        len = half;
    }
    uint64_t v = cur[0];
    free(cur);
    return v;
}

int orc_generate_commitments_fast(uint64_t p, orc_transcript *t, const uint64_t *cols, size_t nv,
                                  uint8_t *roots, uint64_t *points, uint64_t *values, uint64_t *indices,
                                  uint64_t *leaves, uint8_t *siblings, uint8_t *dirs) {
    size_t N = (size_t)1 << nv;
    uint8_t *levels = (uint8_t *)malloc((size_t)43 * 2 * N * 32);
    if (!levels) return ORC_ERR_OUT_OF_MEMORY;
    for (int c = 0; c < 43; c++) {
        uint8_t *lv = levels + (size_t)c * 2 * N * 32;
        orc_merkle_levels(cols + (size_t)c * N, N, lv, NULL);
        memcpy(roots + 32 * c, lv + (2 * N - 2) * 32, 32);
    }
    tr_append_str(t, "POLY_COMMITMENTS");
    for (int c = 0; c < 43; c++) orc_tr_append_bytes(t, roots + 32 * c, 32);
    for (int c = 0; c < 43; c++) {
        uint64_t *pt = points + (size_t)c * nv;
        for (size_t j = 0; j < nv; j++) pt[j] = orc_tr_challenge(t, p);
        values[c] = eval_by_folds(p, cols + (size_t)c * N, N, pt, nv);
        size_t idx = orc_point_to_index(pt, nv);
        indices[c] = idx;
        leaves[c] = cols[(size_t)c * N + idx];
        const uint8_t *lv = levels + (size_t)c * 2 * N * 32;
        size_t off = 0, len = N, ci = idx;
        for (size_t l = 0; l < nv; l++) {
            memcpy(siblings + ((size_t)c * nv + l) * 32, lv + (off + (ci ^ 1)) * 32, 32);
            dirs[(size_t)c * nv + l] = (uint8_t)(ci & 1);
            off += len; len /= 2; ci /= 2;
        }
    }
    tr_append_str(t, "OPENING_CLAIMS");
    for (int c = 0; c < 43; c++) orc_tr_append_field(t, values[c]);
    free(levels);
    return ORC_OK;
}

/* ======================================================================== */
/* Prover.prove + BinarySerializer: prover.zig:73-226, serialization.zig     */
/* ======================================================================== */

size_t orc_proof_size(size_t nv, size_t n_initial_regs, size_t n_outputs, size_t n_lookups) {
    /* header 32; public IO 32+8+8+4+8r+4+256+8+4+8o; constraint 40v+8;
     * lasso 4+24L; openings 43*(32+8v+8+8+8+8+4+32v+v) */
    return 32 + (324 + 8 * n_initial_regs + 8 * n_outputs) + (40 * nv + 8) + (4 + 24 * n_lookups) +
           43 * (68 + 41 * nv);
}

typedef struct { uint8_t *b; size_t pos; } wbuf;
static void w_bytes(wbuf *w, const void *d, size_t n) { memcpy(w->b + w->pos, d, n); w->pos += n; }
static void w_u8(wbuf *w, uint8_t v) { w->b[w->pos++] = v; }
static void w_u32(wbuf *w, uint32_t v) { for (int i = 0; i < 4; i++) w->b[w->pos++] = (uint8_t)(v >> (8 * i)); }
static void w_u64(wbuf *w, uint64_t v) { le64(v, w->b + w->pos); w->pos += 8; }

/* Steps [4/6] and [5/6] of Prover.prove as far as they are observable: generateSumcheckProof (prover.zig:229-289:
 * zero round polynomials, nv challenges) and generateLassoProofs (prover.zig:292-363: one "LASSO_TABLE" + LE64(i)
 * absorption per lookup step, placeholders with num_lookups = 1 => 0 variables).  cpoint may be NULL. */
static void prove_transcript_steps_4_5(uint64_t p, orc_transcript *tr, size_t ns, size_t nv, size_t L, uint64_t *cpoint) {
    tr_append_str(tr, "SUMCHECK_BEGIN");
    orc_tr_append_field(tr, (uint64_t)ns % p);
    orc_tr_append_field(tr, (uint64_t)nv % p);
    for (size_t r = 0; r < nv; r++) {
        for (int k = 0; k < 4; k++) orc_tr_append_field(tr, 0);
        uint64_t ch = orc_tr_challenge(tr, p);
        if (cpoint) cpoint[r] = ch;
    }
    tr_append_str(tr, "LASSO_BEGIN");
    for (size_t i = 0; i < L; i++) {
        tr_append_str(tr, "LASSO_TABLE");
        orc_tr_append_field(tr, (uint64_t)(uint32_t)i % p);
    }
}

/* The same two steps on a fresh transcript bound to (program_hash, entry_pc): the sequential sponge work of one proof,
 * exported so bench.py's cpu_baseline leg can time it for real (returns the next challenge so the work cannot be
 * optimised away).  Not a reference function by itself: it is the part of orc_prove above, run alone. */
uint64_t orc_prove_transcript_only(uint64_t p, const uint8_t program_hash[32], uint64_t entry_pc, size_t ns, size_t nv,
                                   size_t L) {
    orc_transcript *tr = orc_tr_new();
    if (!tr) return 0;
    orc_tr_append_bytes(tr, program_hash, 32);
    orc_tr_append_field(tr, entry_pc % p);
    prove_transcript_steps_4_5(p, tr, ns, nv, L, NULL);
    uint64_t ch = orc_tr_challenge(tr, p);
    orc_tr_free(tr);
    return ch;
}

int orc_prove(uint64_t p, const uint8_t *program, size_t program_len, uint64_t entry_pc,
              const uint64_t *initial_regs, size_t n_initial_regs, int has_initial_regs, size_t max_steps,
              const uint64_t *input, size_t n_input, uint8_t **proof_out, size_t *proof_len,
              size_t *num_steps_out) {
    int rc;
    orc_transcript *tr = orc_tr_new();
    orc_trace *t = orc_trace_new();
    uint64_t *cols = NULL, *points = NULL, *cpoint = NULL;
    uint8_t *roots = NULL, *siblings = NULL, *dirs = NULL, *out = NULL;
    uint64_t values[43], indices[43], leaves[43];
    uint8_t program_hash[32];
    if (!tr || !t) { rc = ORC_ERR_OUT_OF_MEMORY; goto done; }
    if (!has_initial_regs) n_initial_regs = 0;

    /* transcript binding of public inputs, :91-110 */
    orc_sha256(program, program_len, program_hash);
    orc_tr_append_bytes(tr, program_hash, 32);
    orc_tr_append_field(tr, entry_pc % p);
    for (size_t i = 0; i < n_initial_regs; i++) orc_tr_append_field(tr, initial_regs[i] % p);

    /* [1/6] execute, :117-149 */
    rc = orc_vm_run(program, program_len, entry_pc, initial_regs, n_initial_regs, max_steps, input, n_input, t);
    if (rc) goto done;
    size_t ns = t->num_steps;
    if (ns == 0) { rc = ORC_ERR_EMPTY_TRACE; goto done; }
    if (num_steps_out) *num_steps_out = ns;

    /* [2/6] witness, :156-162 */
    size_t nv = orc_log2_ceil(ns), N = (size_t)1 << nv;
    cols = (uint64_t *)malloc((size_t)43 * N * sizeof(uint64_t));
    points = (uint64_t *)malloc(((size_t)43 * nv + 1) * sizeof(uint64_t));
    cpoint = (uint64_t *)malloc((nv + 1) * sizeof(uint64_t));
    roots = (uint8_t *)malloc(43 * 32);
    siblings = (uint8_t *)malloc((size_t)43 * nv * 32 + 1);
    dirs = (uint8_t *)malloc((size_t)43 * nv + 1);
    if (!cols || !points || !cpoint || !roots || !siblings || !dirs) { rc = ORC_ERR_OUT_OF_MEMORY; goto done; }
    size_t nv2;
    orc_witness(p, t, cols, &nv2);

    /* [3/6] constraint system: only the lookup-step count L matters, builder.zig:253-267 */
    size_t L = 0;
    for (size_t i = 0; i < ns; i++) L += t->is_lookup[i];

    /* [4/6] + [5/6]: transcript-only steps (zero round polys, Lasso placeholders) */
    prove_transcript_steps_4_5(p, tr, ns, nv, L, cpoint);

    /* [6/6] generateCommitments */
    rc = orc_generate_commitments(p, tr, cols, nv, roots, points, values, indices, leaves, siblings, dirs);
    if (rc) goto done;

    /* packagePublicIO + BinarySerializer.serialize with an exact-size buffer */
    size_t size = orc_proof_size(nv, n_initial_regs, t->n_outputs, L);
    out = (uint8_t *)malloc(size);
    if (!out) { rc = ORC_ERR_OUT_OF_MEMORY; goto done; }
    wbuf w = {out, 0};
    w_bytes(&w, "ZIGZ", 4); w_u32(&w, 1); w_u64(&w, p); w_u64(&w, ns); w_u32(&w, (uint32_t)nv); w_u32(&w, 0); /* :175-182 */
    w_bytes(&w, program_hash, 32); w_u64(&w, entry_pc); w_u64(&w, t->final_pc);   /* :209-245 */
    w_u32(&w, (uint32_t)n_initial_regs);
    for (size_t i = 0; i < n_initial_regs; i++) w_u64(&w, initial_regs[i]);
    w_u32(&w, 32);
    for (int r = 0; r < 32; r++) w_u64(&w, t->final_regs[r]);
    w_u64(&w, ns);
    w_u32(&w, (uint32_t)t->n_outputs);
    for (size_t i = 0; i < t->n_outputs; i++) w_u64(&w, t->outputs[i]);
    for (size_t r = 0; r < nv; r++) for (int k = 0; k < 4; k++) w_u64(&w, 0);     /* :296-311 */
    for (size_t r = 0; r < nv; r++) w_u64(&w, cpoint[r]);
    w_u64(&w, 0);
    w_u32(&w, (uint32_t)L);                                                         /* :333-344 */
    for (size_t i = 0; i < L; i++) { w_u32(&w, (uint32_t)i); w_u64(&w, 1); w_u32(&w, 0); w_u64(&w, 0); }
    for (int c = 0; c < 43; c++) {                                                  /* :374-429 */
        w_bytes(&w, roots + 32 * c, 32);
        for (size_t j = 0; j < nv; j++) w_u64(&w, points[(size_t)c * nv + j]);
        w_u64(&w, values[c]);
        w_u64(&w, values[c]);   /* OpeningProof.value */
        w_u64(&w, indices[c]);
        w_u64(&w, leaves[c]);
        w_u32(&w, (uint32_t)nv);
        w_bytes(&w, siblings + (size_t)c * nv * 32, nv * 32);
        for (size_t j = 0; j < nv; j++) w_u8(&w, dirs[(size_t)c * nv + j] ? 1 : 0);
    }
    if (w.pos != size) { rc = ORC_ERR_PROTOCOL_ERROR; goto done; }
    *proof_out = out; out = NULL;
    *proof_len = size;
    rc = ORC_OK;
done:
    free(out); free(cols); free(points); free(cpoint); free(roots); free(siblings); free(dirs);
    orc_trace_free(t); orc_tr_free(tr);
    return rc;
}

/* ======================================================================== */
/* BinarySerializer.deserialize + Verifier.verify: verifier.zig:49-294       */
/* ======================================================================== */

typedef struct { const uint8_t *b; size_t len, pos; int bad; } rbuf;
static const uint8_t *r_bytes(rbuf *r, size_t n) {
    if (r->bad || r->len - r->pos < n) { r->bad = 1; return NULL; }
    const uint8_t *q = r->b + r->pos; r->pos += n; return q;
}
static uint64_t r_u64(rbuf *r) { const uint8_t *q = r_bytes(r, 8); return q ? rd64(q) : 0; }
static uint32_t r_u32(rbuf *r) { const uint8_t *q = r_bytes(r, 4); return q ? rd32(q) : 0; }

/* verifySumcheckProof, verifier.zig:182-238 (coeffs already reduced by deserialize's F.init) */
static int verify_sumcheck(uint64_t p, orc_transcript *tr, size_t nv, size_t ncoef,
                           const uint64_t *rounds, uint64_t final_eval) {
    tr_append_str(tr, "SUMCHECK_BEGIN");
    orc_tr_append_field(tr, (uint64_t)nv % p);
    for (size_t round = 0; round < nv; round++) {
        const uint64_t *c = rounds + round * ncoef;
        uint64_t g1 = 0;
        for (size_t k = 0; k < ncoef; k++) g1 = orc_f_add(p, g1, c[k]);
        if (round == 0 && orc_f_add(p, c[0], g1) != final_eval) return ORC_REJECT_INVALID_SUMCHECK;
        uint64_t ch = orc_tr_challenge(tr, p);
        uint64_t ev = 0, pw = 1 % p;
        for (size_t k = 0; k < ncoef; k++) { ev = orc_f_add(p, ev, orc_f_mul(p, c[k], pw)); pw = orc_f_mul(p, pw, ch); }
        orc_tr_append_field(tr, ev);
    }
    return ORC_ACCEPT;
}

int orc_verify(uint64_t p, const uint8_t *proof, size_t proof_len, const uint8_t *program,
               size_t program_len, int *result) {
    rbuf r = {proof, proof_len, 0, 0};
    const uint8_t *magic = r_bytes(&r, 4);
    if (!magic) return ORC_ERR_INVALID_DATA;
    if (memcmp(magic, "ZIGZ", 4) != 0) return ORC_ERR_INVALID_MAGIC;
    if (r_u32(&r) != 1) return r.bad ? ORC_ERR_INVALID_DATA : ORC_ERR_UNSUPPORTED_VERSION;
    uint64_t modulus = r_u64(&r);
    uint64_t num_steps = r_u64(&r);
    uint32_t nv_hdr = r_u32(&r);
    (void)r_u32(&r);
    if (r.bad) return ORC_ERR_INVALID_DATA;
    if (modulus != p) return ORC_ERR_FIELD_MISMATCH;
    /* Proof.init(allocator, metadata.num_steps): shapes come from num_steps, serialization.zig:113 */
    if (num_steps == 0) return ORC_ERR_INVALID_DATA;
    size_t nv = orc_log2_ceil((size_t)num_steps);
    (void)nv_hdr;

    const uint8_t *phash = r_bytes(&r, 32);
    uint64_t initial_pc = r_u64(&r);
    (void)r_u64(&r); /* final_pc */
    uint32_t n_init = r_u32(&r);
    if (r.bad) return ORC_ERR_INVALID_DATA;
    const uint8_t *init_regs = r_bytes(&r, (size_t)n_init * 8);
    uint32_t n_final = r_u32(&r);
    (void)r_bytes(&r, (size_t)n_final * 8);
    (void)r_u64(&r);
    uint32_t n_out = r_u32(&r);
    (void)r_bytes(&r, (size_t)n_out * 8);
    if (r.bad) return ORC_ERR_INVALID_DATA;

    uint64_t *crounds = (uint64_t *)malloc((4 * nv + 1) * sizeof(uint64_t));
    if (!crounds) return ORC_ERR_OUT_OF_MEMORY;
    for (size_t i = 0; i < 4 * nv; i++) crounds[i] = r_u64(&r) % p;
    for (size_t i = 0; i < nv; i++) (void)r_u64(&r);
    uint64_t cfinal = r_u64(&r) % p;
    uint32_t n_lasso = r_u32(&r);
    if (r.bad) { free(crounds); return ORC_ERR_INVALID_DATA; }
    size_t lasso_start = r.pos;
    /* first pass over Lasso entries only to find the openings; verified after PHASE 4 below */
    for (uint32_t i = 0; i < n_lasso; i++) {
        (void)r_u32(&r); (void)r_u64(&r);
        uint32_t lv = r_u32(&r);
        if (r.bad) break;
        (void)r_bytes(&r, ((size_t)lv * 3 + (size_t)lv + 1) * 8);
    }
    if (r.bad) { free(crounds); return ORC_ERR_INVALID_DATA; }
    size_t openings_start = r.pos;

    int rc = ORC_OK;
    orc_transcript *tr = orc_tr_new();
    uint8_t program_hash[32];
    orc_sha256(program, program_len, program_hash);
    if (memcmp(program_hash, phash, 32) != 0) { rc = ORC_ERR_PROGRAM_HASH_MISMATCH; goto out; } /* :105-107 */
    orc_tr_append_bytes(tr, program_hash, 32);
    orc_tr_append_field(tr, initial_pc % p);
    for (uint32_t i = 0; i < n_init; i++) orc_tr_append_field(tr, rd64(init_regs + 8 * i) % p);

    /* parse the 43 openings */
    struct { const uint8_t *root; uint64_t value, pvalue, leaf; uint32_t plen; const uint8_t *sib, *dir; } op[43];
    for (int c = 0; c < 43; c++) {
        op[c].root = r_bytes(&r, 32);
        (void)r_bytes(&r, nv * 8);
        op[c].value = r_u64(&r) % p;
        op[c].pvalue = r_u64(&r) % p;
        (void)r_u64(&r);
        op[c].leaf = r_u64(&r) % p;
        op[c].plen = r_u32(&r);
        if (r.bad) break;
        op[c].sib = r_bytes(&r, (size_t)op[c].plen * 32);
        op[c].dir = r_bytes(&r, op[c].plen);
    }
    if (r.bad) { rc = ORC_ERR_INVALID_DATA; goto out; }
    (void)openings_start;

    tr_append_str(tr, "POLY_COMMITMENTS"); /* :126-137 */
    for (int c = 0; c < 43; c++) orc_tr_append_bytes(tr, op[c].root, 32);
    for (int c = 0; c < 43; c++) for (size_t j = 0; j < nv; j++) (void)orc_tr_challenge(tr, p); /* :152-156 */
    tr_append_str(tr, "OPENING_CLAIMS");
    for (int c = 0; c < 43; c++) orc_tr_append_field(tr, op[c].value);

    *result = verify_sumcheck(p, tr, nv, 4, crounds, cfinal); /* PHASE 4 */
    if (*result != ORC_ACCEPT) goto out;

    { /* PHASE 5, :241-267 */
        rbuf lr = {proof, proof_len, lasso_start, 0};
        for (uint32_t i = 0; i < n_lasso; i++) {
            uint32_t table_id = r_u32(&lr);
            (void)r_u64(&lr);
            uint32_t lv = r_u32(&lr);
            uint64_t *lrounds = (uint64_t *)malloc(((size_t)3 * lv + 1) * sizeof(uint64_t));
            for (size_t k = 0; k < (size_t)3 * lv; k++) lrounds[k] = r_u64(&lr) % p;
            for (size_t k = 0; k < lv; k++) (void)r_u64(&lr);
            uint64_t lfinal = r_u64(&lr) % p;
            tr_append_str(tr, "LASSO_BEGIN");
            tr_append_str(tr, "LASSO_TABLE");
            orc_tr_append_field(tr, (uint64_t)table_id % p);
            int res = verify_sumcheck(p, tr, lv, 3, lrounds, lfinal);
            free(lrounds);
            if (res != ORC_ACCEPT) { *result = ORC_REJECT_INVALID_LOOKUP; goto out; }
        }
    }
    for (int c = 0; c < 43; c++) { /* PHASE 6, :270-294 */
        if (op[c].value != op[c].pvalue) { *result = ORC_REJECT_INVALID_COMMITMENT; goto out; }
        /* Scheme.verify: proof.point.len (= nv) vs commitment.num_vars (= point.len) always equal */
        if (!orc_merkle_verify(op[c].root, op[c].leaf, op[c].sib, op[c].dir, op[c].plen)) {
            *result = ORC_REJECT_INVALID_COMMITMENT; goto out;
        }
    }
    *result = ORC_ACCEPT;
out:
    free(crounds);
    orc_tr_free(tr);
    return rc;
}

void orc_free(void *ptr) { free(ptr); }
