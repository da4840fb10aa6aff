// Host build of the DEVICE Keccak source (zigz_amd/csrc/keccak.hpp with its plain-C fallbacks for v_bitop3 /
// v_alignbit): checks the bit-interleaved permutation against the 64-bit formulation on random states and prints
// the canonical bytes of sha3_leaf / sha3_node for the values given on the command line, so the Python test can
// compare them with hashlib.  Usage: keccak_forms v0 v1 ...   -> "leaf <v> <hex>" per value, "node <hex>" per pair.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "keccak.hpp"

using namespace zk;

static void il(uint64_t v, uint32_t &e, uint32_t &o) {
    e = o = 0;
    for (int b = 0; b < 32; b++) {
        e |= (uint32_t)((v >> (2 * b)) & 1) << b;
        o |= (uint32_t)((v >> (2 * b + 1)) & 1) << b;
    }
}
static uint64_t dil(uint32_t e, uint32_t o) {
    uint64_t v = 0;
    for (int b = 0; b < 32; b++) {
        v |= (uint64_t)((e >> b) & 1) << (2 * b);
        v |= (uint64_t)((o >> b) & 1) << (2 * b + 1);
    }
    return v;
}
static void print_digest(const char *tag, const Digest &tree_form) {
    const Digest c = canonical_digest(tree_form);
    printf("%s", tag);
    for (int i = 0; i < 4; i++)
        for (int b = 0; b < 8; b++) printf("%02x", (unsigned)((c.w[i] >> (8 * b)) & 0xff));
    printf("\n");
}

int main(int argc, char **argv) {
    uint64_t seed = 0x5A49475Aull;
    auto next = [&]() { seed += 0x9E3779B97F4A7C15ull; uint64_t z = seed; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
                        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
    for (int it = 0; it < 200; it++) {
        uint64_t a[25];
        uint32_t e[25], o[25];
        for (int i = 0; i < 25; i++) { a[i] = next(); il(a[i], e[i], o[i]); }
        for (int r = 0; r < 24; r++) ZK_KECCAK_ROUND(a, KECCAK_RC[r]);
        keccak_f1600_il(e, o);
        for (int i = 0; i < 25; i++)
            if (dil(e[i], o[i]) != a[i]) { printf("permutation mismatch: lane %d, state %d\n", i, it); return 1; }
    }
    for (int it = 0; it < 1000; it++) {  // compress/expand are inverse on every lane
        const uint64_t v = next();
        const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
        const uint32_t e = compress_even(lo) | (compress_even(hi) << 16), o = compress_even(lo >> 1) | (compress_even(hi >> 1) << 16);
        uint32_t e2, o2;
        il(v, e2, o2);
        if (e != e2 || o != o2) { printf("compress mismatch\n"); return 1; }
        const Digest c = canonical_digest(Digest{{((uint64_t)o << 32) | e, 0, 0, 0}});
        if (c.w[0] != v) { printf("expand mismatch\n"); return 1; }
    }
    printf("permutation ok\n");
    Digest prev{};
    for (int k = 1; k < argc; k++) {
        const uint64_t v = strtoull(argv[k], nullptr, 10);
        const Digest d = sha3_leaf(v);
        char tag[64];
        snprintf(tag, sizeof tag, "leaf %llu ", (unsigned long long)v);
        print_digest(tag, d);
        if (k % 2 == 0) print_digest("node ", sha3_node(prev, d));
        prev = d;
    }
    return 0;
}
