/* Plain-C99 consumer of include/zigz_hip.h: what a cgo-style / Zig `extern` binding sees ("the C ABI
 * exercised from a C/C++ driver is the verifiable boundary").  No Python, no torch, no oracle.
 *
 *   abi_driver host   -- host-only entry points (status names, SHA3, transcript); runs without a GPU and checks
 *                        that zigz_ctx_create fails loudly (NoDevice) when there is none
 *   abi_driver gpu    -- the reference's own small known answers through the GPU path:
 *                        [1,2,3,4] sum = 10, roundPolynomial = [3,4], partialEval(0) = [1,2]
 *                        (src/poly/multilinear.zig:436-506), the derived sumcheck vector of SURVEY s8c,
 *                        Merkle commit / open of 5 values (pad to 8, height 3; merkle_tree.zig:425-571)
 * Exit code 0 = every check passed; each failure prints a line.
 */
#include <stdio.h>
#include <string.h>
#include <stdint.h>

#include "zigz_hip.h"

static int failures = 0;
#define CHECK(cond, ...)                                                                                   \
    do {                                                                                                   \
        if (!(cond)) {                                                                                     \
            failures++;                                                                                    \
            printf("FAIL %s:%d: ", __FILE__, __LINE__);                                                    \
            printf(__VA_ARGS__);                                                                           \
            printf("\n");                                                                                  \
        }                                                                                                  \
    } while (0)

static void hex(const uint8_t *b, size_t n, char *out) {
    static const char *d = "0123456789abcdef";
    size_t i;
    for (i = 0; i < n; i++) { out[2 * i] = d[b[i] >> 4]; out[2 * i + 1] = d[b[i] & 15]; }
    out[2 * n] = 0;
}

static int host_checks(void) {
    uint8_t dg[32];
    char hx[65];
    zigz_transcript *t;
    uint64_t c0, c1;
    int ndev = -1;
    zigz_status st;

    CHECK(zigz_abi_version() == 1, "abi version %u", zigz_abi_version());
    CHECK(strcmp(zigz_status_name(ZIGZ_ERR_LENGTH_NOT_POWER_OF_TWO), "LengthNotPowerOfTwo") == 0, "status name");
    CHECK(strcmp(zigz_status_name(ZIGZ_ERR_NO_DEVICE), "NoDevice") == 0, "status name");
    zigz_sha3_256((const uint8_t *)"", 0, dg); /* FIPS 202 known answer */
    hex(dg, 32, hx);
    CHECK(strcmp(hx, "a7ffc6f8bf1ed76651c14756a061d662f580ff4de43b49fa82d80a4b80f8434a") == 0, "sha3(\"\") = %s", hx);
    zigz_sha256((const uint8_t *)"abc", 3, dg); /* FIPS 180-4 known answer */
    hex(dg, 32, hx);
    CHECK(strcmp(hx, "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad") == 0, "sha256(abc) = %s", hx);
    /* fresh transcript, absorb the round polynomial [3,4] of [1,2,3,4]: first challenge of the SURVEY s8c vector */
    t = zigz_transcript_new();
    zigz_transcript_append_field(t, 3);
    zigz_transcript_append_field(t, 4);
    c0 = zigz_transcript_challenge(t);
    zigz_transcript_append_field(t, 42686404);
    zigz_transcript_append_field(t, 1);
    c1 = zigz_transcript_challenge(t);
    zigz_transcript_free(t);
    CHECK(c0 == 1027976162ull && c1 == 792614669ull, "transcript challenges %llu %llu", (unsigned long long)c0,
          (unsigned long long)c1);
    st = zigz_device_count(&ndev);
    if (st != ZIGZ_OK || ndev == 0) { /* no GPU: the product must refuse, there is no CPU path */
        zigz_ctx *ctx = NULL;
        st = zigz_ctx_create(0, &ctx);
        CHECK(st == ZIGZ_ERR_NO_DEVICE && ctx == NULL, "ctx_create without a GPU returned %d", (int)st);
    }
    return failures;
}

static int gpu_checks(void) {
    zigz_ctx *ctx = NULL;
    const uint64_t evals[4] = {1, 2, 3, 4};
    uint64_t out[4], rounds[4], point[2], fe = 0, sum = 0;
    zigz_status st = zigz_ctx_create(0, &ctx);
    if (st != ZIGZ_OK) {
        printf("FAIL: zigz_ctx_create -> %s\n", zigz_status_name(st));
        return 1;
    }
    st = zigz_mle_sum(ctx, evals, 4, &sum);
    CHECK(st == ZIGZ_OK && sum == 10, "sum %llu (%s)", (unsigned long long)sum, zigz_status_name(st));
    st = zigz_mle_round_poly(ctx, evals, 4, out);
    CHECK(st == ZIGZ_OK && out[0] == 3 && out[1] == 4, "roundPolynomial [%llu,%llu]", (unsigned long long)out[0],
          (unsigned long long)out[1]);
    st = zigz_mle_bind(ctx, evals, 4, 0, out);
    CHECK(st == ZIGZ_OK && out[0] == 1 && out[1] == 2, "partialEval(0)");
    st = zigz_mle_bind(ctx, evals, 3, 0, out); /* error mapping: LengthNotPowerOfTwo, multilinear.zig:37-44 */
    CHECK(st == ZIGZ_ERR_LENGTH_NOT_POWER_OF_TWO, "n = 3 -> %s", zigz_status_name(st));
    st = zigz_sumcheck_prove(ctx, evals, 4, rounds, point, &fe);
    CHECK(st == ZIGZ_OK && rounds[0] == 3 && rounds[1] == 4 && rounds[2] == 42686404ull && rounds[3] == 1 &&
              point[0] == 1027976162ull && point[1] == 792614669ull && fe == 835301073ull,
          "sumcheck [[%llu,%llu],[%llu,%llu]] point [%llu,%llu] final %llu", (unsigned long long)rounds[0],
          (unsigned long long)rounds[1], (unsigned long long)rounds[2], (unsigned long long)rounds[3],
          (unsigned long long)point[0], (unsigned long long)point[1], (unsigned long long)fe);
    {
        /* final_eval == eval(reverse(point)): MSB-first bind vs LSB-first eval */
        uint64_t rev[2], v = 0;
        rev[0] = point[1];
        rev[1] = point[0];
        st = zigz_mle_eval(ctx, evals, 4, rev, 2, &v);
        CHECK(st == ZIGZ_OK && v == fe, "eval(reverse(point)) = %llu", (unsigned long long)v);
    }
    {
        const uint64_t vals[5] = {10, 20, 30, 40, 50};
        uint8_t root[32], root2[32], sib[3 * 32], dirs[3], leaf_dg[32], node[64], le[8];
        size_t height = 0, idx;
        zigz_merkle *tree = NULL;
        st = zigz_merkle_commit(ctx, vals, 5, root, &height, &tree);
        CHECK(st == ZIGZ_OK && height == 3 && tree != NULL, "commit: %s height %zu", zigz_status_name(st), height);
        for (idx = 0; tree && idx < 5; idx++) {
            uint64_t leaf = 0;
            size_t lvl, pos = idx;
            int k;
            st = zigz_merkle_open(ctx, tree, idx, sib, dirs, &leaf);
            CHECK(st == ZIGZ_OK && leaf == vals[idx], "open %zu: %s leaf %llu", idx, zigz_status_name(st),
                  (unsigned long long)leaf);
            /* recompute the root on the host exactly as SimpleMerkleTree.verify does (merkle_tree.zig:362-373) */
            for (k = 0; k < 8; k++) le[k] = (uint8_t)(leaf >> (8 * k));
            zigz_sha3_256(le, 8, leaf_dg);
            for (lvl = 0; lvl < height; lvl++, pos >>= 1) {
                if (pos & 1) { memcpy(node, sib + 32 * lvl, 32); memcpy(node + 32, leaf_dg, 32); }
                else         { memcpy(node, leaf_dg, 32); memcpy(node + 32, sib + 32 * lvl, 32); }
                zigz_sha3_256(node, 64, leaf_dg);
            }
            memcpy(root2, leaf_dg, 32);
            CHECK(memcmp(root, root2, 32) == 0, "path of index %zu does not hash to the root", idx);
        }
        if (tree) {
            uint64_t leaf;
            st = zigz_merkle_open(ctx, tree, 8, sib, dirs, &leaf); /* IndexOutOfBounds, merkle_tree.zig:325 */
            CHECK(st == ZIGZ_ERR_INDEX_OUT_OF_BOUNDS, "open(8) -> %s", zigz_status_name(st));
            zigz_merkle_destroy(ctx, tree);
        }
    }
    zigz_ctx_destroy(ctx);
    return failures;
}

int main(int argc, char **argv) {
    int gpu = argc > 1 && strcmp(argv[1], "gpu") == 0;
    host_checks();
    if (gpu) gpu_checks();
    printf("%s: %d failure(s)\n", gpu ? "gpu" : "host", failures);
    return failures ? 1 : 0;
}
