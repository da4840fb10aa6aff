// Host-side sequential hashing that stays on the CPU by construction:
// the SHA3-256 Fiat-Shamir sponge (src/core/hash.zig:255-324), flat SHA3 commitments
// (src/lookups/lasso_prover.zig:242-252) and the SHA-256 program binding (src/prover/prover.zig:98-100).
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace zk {

class Sha3_256 {
  public:
    Sha3_256() { reset(); }
    void reset();
    void update(const uint8_t *data, size_t len);
    void update_le64(uint64_t v);
    // finalises a COPY (the object stays usable), as hash.zig:305-306 does
    void digest_copy(uint8_t out[32]) const;
    void finalize(uint8_t out[32]);
    // the sponge itself, for the batched absorber (host_sponge_batch.cpp)
    uint64_t *raw_state() { return st_; }
    size_t *raw_pos() { return &pos_; }

  private:
    uint64_t st_[25];
    size_t pos_;  // bytes absorbed into the current 136-byte block
};

// FiatShamirTranscript over BabyBear
class Transcript {
  public:
    void append_bytes(const uint8_t *d, size_t n) { h_.update(d, n); }
    void append_field(uint64_t canonical) { h_.update_le64(canonical); }
    uint64_t challenge();  // hash.zig:301-316
    void append_tagged_counter(const uint8_t *tag, size_t tag_len, uint64_t start, uint64_t count);

  private:
    Sha3_256 h_;
};

void sha3_256(const uint8_t *data, size_t len, uint8_t out[32]);
const char *host_keccak_impl();  // "scalar", "bmi2", "avx512f" or "avx512vl": the variant picked at load time
// one Keccak-f[1600] through a chosen variant: 0 = picked, 1 = scalar, 2 = bmi2, 3 = avx512f, 4 = avx512vl (if supported)
void host_keccak_permute(uint64_t st[25], int which);
void sha256(const uint8_t *data, size_t len, uint8_t out[32]);

// Sponge service (host_sponge_batch.cpp): n server threads, each advancing up to 8 transcripts' tagged-counter absorptions in
// lock step with one 8-way AVX-512 permutation per block (n = 0: off, every transcript absorbs on its own thread; ignored
// on CPUs without AVX-512F).  Call it while no transcript is absorbing.
void host_sponge_servers(int n);
bool host_sponge_batching();
// false when no server accepted the job: the caller absorbs sequentially
bool host_sponge_absorb_tagged(uint64_t st[25], size_t *pos, const uint8_t *tag, size_t tag_len, uint64_t start, uint64_t count);
void host_keccak_permute_x8(uint64_t st[25][8]);  // diagnostics: the 8-way permutation (requires AVX-512F)

}  // namespace zk
