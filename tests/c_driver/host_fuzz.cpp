// Sanitizer-built fuzz driver for the host mirror's untrusted-input surface (built by tests/test_host_sanitized.py
// with -fsanitize=address,undefined from zigz_amd/csrc/host/*.cpp; CPU only -- GPU AddressSanitizer is not available).
//   host_fuzz <proof.bin> <program.bin> <iterations>
// 1. the intact proof must verify (Accept) and re-serialize to the same bytes;
// 2. every mutated proof (bit flips, truncations, length-field corruption, random splices) must come back as a clean
//    result -- Accept / Reject* / an error code -- never a crash, an out-of-bounds access or a leak;
// 3. random byte strings run through the VM (zigzh_vm_run) as programs: any outcome but memory errors is fine.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "zigz_host.h"

static std::vector<uint8_t> slurp(const char *path) {
    std::vector<uint8_t> v;
    FILE *f = fopen(path, "rb");
    if (!f) return v;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
    fclose(f);
    return v;
}

static uint64_t rng_state = 0x5A49475Aull;
static uint64_t rnd() {
    rng_state += 0x9E3779B97F4A7C15ull;
    uint64_t z = rng_state;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main(int argc, char **argv) {
    if (argc < 4) return 2;
    const std::vector<uint8_t> proof = slurp(argv[1]), prog = slurp(argv[2]);
    const int iters = atoi(argv[3]);
    if (proof.empty() || prog.empty()) { printf("cannot read inputs\n"); return 2; }
    int result = -1;
    int rc = zigzh_verify(proof.data(), proof.size(), prog.data(), prog.size(), &result);
    if (rc != 0 || result != 0) { printf("intact proof: rc %d result %d (%s)\n", rc, result, zigzh_last_error()); return 1; }
    uint8_t *again = nullptr;
    size_t again_len = 0;
    rc = zigzh_reserialize(proof.data(), proof.size(), &again, &again_len);
    if (rc != 0 || again_len != proof.size() || memcmp(again, proof.data(), again_len) != 0) { printf("round trip differs\n"); return 1; }
    zigzh_free(again);
    int accepted = 0, rejected = 0, errors = 0;
    for (int it = 0; it < iters; it++) {
        std::vector<uint8_t> m = proof;
        switch (rnd() % 5) {
        case 0:  // bit flips
            for (int k = 0, n = 1 + (int)(rnd() % 4); k < n; k++) m[rnd() % m.size()] ^= (uint8_t)(1u << (rnd() % 8));
            break;
        case 1:  // truncation
            m.resize(rnd() % m.size());
            break;
        case 2: {  // a 32-bit little-endian field overwritten with an extreme value (counts / lengths live in those)
            const size_t off = rnd() % (m.size() - 4);
            const uint32_t v = (rnd() & 1) ? 0xFFFFFFFFu : (uint32_t)rnd();
            memcpy(&m[off], &v, 4);
            break;
        }
        case 3: {  // splice: a random chunk copied over another place
            const size_t len = 1 + rnd() % 64, a = rnd() % (m.size() - len), b = rnd() % (m.size() - len);
            memmove(&m[a], &m[b], len);
            break;
        }
        default:  // extension with garbage
            for (int k = 0, n = 1 + (int)(rnd() % 128); k < n; k++) m.push_back((uint8_t)rnd());
        }
        result = -1;
        rc = zigzh_verify(m.empty() ? (const uint8_t *)"" : m.data(), m.size(), prog.data(), prog.size(), &result);
        if (rc != 0) errors++;
        else if (result == 0) accepted++;
        else rejected++;
        uint8_t *o = nullptr;
        size_t ol = 0;
        if (zigzh_reserialize(m.empty() ? (const uint8_t *)"" : m.data(), m.size(), &o, &ol) == 0) zigzh_free(o);
    }
    for (int it = 0; it < iters / 4; it++) {  // random "programs"
        std::vector<uint8_t> p(4 * (1 + rnd() % 64));
        for (auto &b : p) b = (uint8_t)rnd();
        uint64_t regs[32], pc = 0;
        size_t steps = 0;
        (void)zigzh_vm_run(p.data(), p.size(), 0x1000, 2000, regs, &pc, &steps);
    }
    printf("ok: %d mutated proofs -> %d accepted, %d rejected, %d errors\n", iters, accepted, rejected, errors);
    return 0;
}
