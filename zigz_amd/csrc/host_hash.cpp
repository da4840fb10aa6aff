#include "host_hash.hpp"

#include <stdlib.h>
#include <string.h>

#include <chrono>

#include "field.hpp"
#include "keccak.hpp"  // host build: the same round macro as the device kernels

namespace zk {

void keccak_f1600_avx512(uint64_t st[25]);  // host_keccak_avx512.cpp
bool cpu_has_avx512f();
void keccak_f1600_bmi(uint64_t st[25]);     // host_keccak_bmi.cpp
bool cpu_has_bmi2();
void keccak_f1600_avx512vl(uint64_t st[25]);  // host_keccak_avx512vl.cpp (one lane per vector register)
bool cpu_has_avx512vl();

static void permute_scalar(uint64_t st[25]) {
    for (int r = 0; r < 24; r++) ZK_KECCAK_ROUND(st, KECCAK_RC[r]);
}

// The sponge is the sequential critical path of a proof, and which single-state permutation is fastest
// depends on the micro-architecture (the AVX-512 plane form wins on Intel server cores, the scalar/BMI
// forms on Zen 5), so the choice is made once at load time by timing each supported variant (~1 ms).
namespace {
struct Variant { const char *name; void (*fn)(uint64_t *); };
Variant pick_variant() {
    Variant cands[4];
    int n = 0;
    cands[n++] = {"scalar", permute_scalar};
    if (cpu_has_bmi2()) cands[n++] = {"bmi2", keccak_f1600_bmi};
    if (cpu_has_avx512f()) cands[n++] = {"avx512f", keccak_f1600_avx512};
    if (cpu_has_avx512vl()) cands[n++] = {"avx512vl", keccak_f1600_avx512vl};
    const char *force = getenv("ZIGZ_HOST_KECCAK");
    if (force)
        for (int i = 0; i < n; i++)
            if (strcmp(force, cands[i].name) == 0) return cands[i];
    int best = 0;
    double best_t = 1e30;
    for (int i = 0; i < n; i++) {
        uint64_t st[25];
        for (int k = 0; k < 25; k++) st[k] = 0x9E3779B97F4A7C15ull * (uint64_t)(k + 1);
        for (int k = 0; k < 200; k++) cands[i].fn(st);  // warm up
        double t = 1e30;
        for (int rep = 0; rep < 3; rep++) {
            auto t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < 1000; k++) cands[i].fn(st);
            double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (dt < t) t = dt;
        }
        volatile uint64_t sink = st[0];
        (void)sink;
        if (t < best_t) { best_t = t; best = i; }
    }
    return cands[best];
}
const Variant g_variant = pick_variant();
}  // namespace

static inline void permute(uint64_t st[25]) { g_variant.fn(st); }

const char *host_keccak_impl() { return g_variant.name; }
void host_keccak_permute(uint64_t st[25], int which) {
    if (which == 1) permute_scalar(st);
    else if (which == 2 && cpu_has_bmi2()) keccak_f1600_bmi(st);
    else if (which == 3 && cpu_has_avx512f()) keccak_f1600_avx512(st);
    else if (which == 4 && cpu_has_avx512vl()) keccak_f1600_avx512vl(st);
    else g_variant.fn(st);
}

void Sha3_256::reset() {
    memset(st_, 0, sizeof(st_));
    pos_ = 0;
}

static inline void xor_into(uint8_t *dst, const uint8_t *src, size_t n) {
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t a, b;
        memcpy(&a, dst + i, 8);
        memcpy(&b, src + i, 8);
        a ^= b;
        memcpy(dst + i, &a, 8);
    }
    for (; i < n; i++) dst[i] ^= src[i];
}

// Absorb: XOR message bytes into the byte image of the state (little-endian host), block by block.
void Sha3_256::update(const uint8_t *data, size_t len) {
    constexpr size_t RATE = 136;
    uint8_t *sb = reinterpret_cast<uint8_t *>(st_);
    while (len) {
        size_t n = RATE - pos_;
        if (n > len) n = len;
        xor_into(sb + pos_, data, n);
        pos_ += n;
        data += n;
        len -= n;
        if (pos_ == RATE) {
            permute(st_);
            pos_ = 0;
        }
    }
}

void Sha3_256::update_le64(uint64_t v) {
    if ((pos_ & 7) == 0) {
        st_[pos_ >> 3] ^= v;
        pos_ += 8;
        if (pos_ == 136) { permute(st_); pos_ = 0; }
    } else {
        uint8_t b[8];
        memcpy(b, &v, 8);
        update(b, 8);
    }
}

static void finish(uint64_t st[25], size_t pos, uint8_t out[32]) {
    st[pos >> 3] ^= (uint64_t)0x06 << (8 * (pos & 7));
    st[16] ^= 0x8000000000000000ull;  // byte 135
    permute(st);
    memcpy(out, st, 32);
}

void Sha3_256::digest_copy(uint8_t out[32]) const {
    uint64_t st[25];
    memcpy(st, st_, sizeof(st));
    finish(st, pos_, out);
}

void Sha3_256::finalize(uint8_t out[32]) { finish(st_, pos_, out); }

uint64_t Transcript::challenge() {
    uint8_t d[32];
    h_.digest_copy(d);
    uint64_t v;
    memcpy(&v, d, 8);
    uint64_t r = v % (uint64_t)P;  // digestToFieldElement, hash.zig:228-242
    h_.update(d, 32);
    return r;
}

void Transcript::append_tagged_counter(const uint8_t *tag, size_t tag_len, uint64_t start, uint64_t count) {
    // materialise the records (tag || LE64((start+k) mod p)) in chunks and absorb them in bulk
    constexpr size_t CHUNK = 512;
    const size_t rec = tag_len + 8;
    // a long run goes to the sponge service when one is running (8 transcripts per core in lock step; same bytes absorbed)
    if (rec <= 256 && count * rec >= (64u << 10) && host_sponge_batching() &&
        host_sponge_absorb_tagged(h_.raw_state(), h_.raw_pos(), tag, tag_len, start, count))
        return;
    if (rec > 256) {
        for (uint64_t k = 0; k < count; k++) { h_.update(tag, tag_len); h_.update_le64((start + k) % (uint64_t)P); }
        return;
    }
    uint8_t buf[CHUNK * 64 + 32];  // + slack: every record is written as one 32-byte template copy
    const size_t per = (sizeof(buf) - 32) / rec;
    uint8_t tmpl[32] = {0};
    memcpy(tmpl, tag, tag_len < 24 ? tag_len : 24);
    uint64_t k = 0, v = start % (uint64_t)P;  // (start + k) mod p kept incrementally
    while (k < count) {
        size_t m = count - k < per ? (size_t)(count - k) : per;
        uint8_t *q = buf;
        if (tag_len <= 24) {
            for (size_t j = 0; j < m; j++, q += rec) {
                memcpy(q, tmpl, 32);  // tag (the tail is overwritten by the counter and the next record)
                memcpy(q + tag_len, &v, 8);
                if (++v == (uint64_t)P) v = 0;
            }
        } else {
            for (size_t j = 0; j < m; j++, q += rec) {
                memcpy(q, tag, tag_len);
                memcpy(q + tag_len, &v, 8);
                if (++v == (uint64_t)P) v = 0;
            }
        }
        h_.update(buf, m * rec);
        k += m;
    }
}

void sha3_256(const uint8_t *data, size_t len, uint8_t out[32]) {
    Sha3_256 h;
    h.update(data, len);
    h.finalize(out);
}

// ---------------------------------------------------------------- SHA-256 (FIPS 180-4)
static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

static inline uint32_t ror(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

static void compress(uint32_t h[8], const uint8_t *blk) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++)
        w[i] = ((uint32_t)blk[4 * i] << 24) | ((uint32_t)blk[4 * i + 1] << 16) | ((uint32_t)blk[4 * i + 2] << 8) | blk[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = ror(w[i - 15], 7) ^ ror(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = ror(w[i - 2], 17) ^ ror(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t v[8];
    memcpy(v, h, sizeof(v));
    for (int i = 0; i < 64; i++) {
        uint32_t t1 = v[7] + (ror(v[4], 6) ^ ror(v[4], 11) ^ ror(v[4], 25)) + ((v[4] & v[5]) ^ (~v[4] & v[6])) + K256[i] + w[i];
        uint32_t t2 = (ror(v[0], 2) ^ ror(v[0], 13) ^ ror(v[0], 22)) + ((v[0] & v[1]) ^ (v[0] & v[2]) ^ (v[1] & v[2]));
        v[7] = v[6]; v[6] = v[5]; v[5] = v[4]; v[4] = v[3] + t1; v[3] = v[2]; v[2] = v[1]; v[1] = v[0]; v[0] = t1 + t2;
    }
    for (int i = 0; i < 8; i++) h[i] += v[i];
}

void sha256(const uint8_t *data, size_t len, uint8_t out[32]) {
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    size_t full = len / 64;
    for (size_t i = 0; i < full; i++) compress(h, data + 64 * i);
    uint8_t tail[128] = {0};
    size_t rem = len - 64 * full;
    if (rem) memcpy(tail, data + 64 * full, rem);
    tail[rem] = 0x80;
    size_t tl = rem < 56 ? 64 : 128;
    uint64_t bits = (uint64_t)len * 8;
    for (int b = 0; b < 8; b++) tail[tl - 1 - b] = (uint8_t)(bits >> (8 * b));
    compress(h, tail);
    if (tl == 128) compress(h, tail + 64);
    for (int i = 0; i < 8; i++) {
        out[4 * i] = (uint8_t)(h[i] >> 24); out[4 * i + 1] = (uint8_t)(h[i] >> 16);
        out[4 * i + 2] = (uint8_t)(h[i] >> 8); out[4 * i + 3] = (uint8_t)h[i];
    }
}

}  // namespace zk
