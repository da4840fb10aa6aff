// Sponge service: several proofs' Fiat-Shamir transcripts advanced in LOCK STEP by one host thread.
//
// A proof's transcript is a chain of ~147 k dependent Keccak-f[1600] at a 2^20 trace (one "LASSO_TABLE" record of 19 bytes per
// lookup step, prover.zig:302-312): nothing inside one proof can be done in parallel, and one permutation costs ~880 cycles
// on Zen 5 whether its 25 lanes sit in xmm, ymm or zmm registers (tools/host_keccak_wide.cpp: 0.176 / 0.176 / 0.182 us per
// call for 1-2 / 4 / 8 states).  So a core that advances 8 independent sponges in the 8 qword lanes of the zmm registers does
// the work of ~7 cores advancing them one by one.  A proving service has that many independent proofs in flight per GPU --
// it is how the GPU is kept busy at all -- and with one core per proof the host, not the GPU, bounds the throughput
// (DESIGN.md s4, s9).
//
// zigz_host_sponge_servers(k) starts k server threads.  A transcript that has to absorb a long tagged-counter run hands its
// sponge state to the service and sleeps; a server owns 8 slots, each iteration XORs every active slot's next 136-byte block
// into its column of the interleaved state and runs ONE 8-way permutation; requests join and leave at block boundaries.  The
// bytes absorbed, and therefore every challenge, are exactly those of the sequential code (tests/test_host_mirror.py).
#include <immintrin.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "field.hpp"
#include "host_hash.hpp"

namespace zk {

bool cpu_has_avx512f();

namespace {

constexpr int SLOTS = 8;
constexpr size_t RATE = 136;

const uint64_t RC8[24] = {
    0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull,
    0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull,
    0x0000000080008009ull, 0x000000008000000aull, 0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull,
    0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
    0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};

// Keccak-f[1600] on 8 states at once: st[lane][slot], one zmm register per lane (the formulation of
// host_keccak_avx512vl.cpp: vpternlogq for XOR3 and chi, vprolq for rho, pi by renaming)
__attribute__((target("avx512f"))) void keccak_f1600_x8(uint64_t st[25][SLOTS]) {
#define X8_X3(a, b, c) _mm512_ternarylogic_epi64((a), (b), (c), 0x96)
#define X8_CHI(a, b, c) _mm512_ternarylogic_epi64((a), (b), (c), 0xD2)
#define X8_ROL(a, n) _mm512_rol_epi64((a), (n))
#define X8_B(n, a, cm, rp) X8_ROL(X8_X3(a, cm, rp), n)
#define X8_LD(i) _mm512_load_si512(st[i])
    __m512i a0 = X8_LD(0), a1 = X8_LD(1), a2 = X8_LD(2), a3 = X8_LD(3), a4 = X8_LD(4), a5 = X8_LD(5), a6 = X8_LD(6), a7 = X8_LD(7);
    __m512i a8 = X8_LD(8), a9 = X8_LD(9), a10 = X8_LD(10), a11 = X8_LD(11), a12 = X8_LD(12), a13 = X8_LD(13), a14 = X8_LD(14);
    __m512i a15 = X8_LD(15), a16 = X8_LD(16), a17 = X8_LD(17), a18 = X8_LD(18), a19 = X8_LD(19), a20 = X8_LD(20), a21 = X8_LD(21);
    __m512i a22 = X8_LD(22), a23 = X8_LD(23), a24 = X8_LD(24);
#pragma unroll 2
    for (int r = 0; r < 24; r++) {
        const __m512i c0 = X8_X3(X8_X3(a0, a5, a10), a15, a20), c1 = X8_X3(X8_X3(a1, a6, a11), a16, a21);
        const __m512i c2 = X8_X3(X8_X3(a2, a7, a12), a17, a22), c3 = X8_X3(X8_X3(a3, a8, a13), a18, a23);
        const __m512i c4 = X8_X3(X8_X3(a4, a9, a14), a19, a24);
        const __m512i r0 = X8_ROL(c0, 1), r1 = X8_ROL(c1, 1), r2 = X8_ROL(c2, 1), r3 = X8_ROL(c3, 1), r4 = X8_ROL(c4, 1);
        const __m512i b00 = X8_X3(a0, c4, r1);
        const __m512i b10 = X8_B(1, a1, c0, r2), b20 = X8_B(62, a2, c1, r3), b05 = X8_B(28, a3, c2, r4), b15 = X8_B(27, a4, c3, r0);
        const __m512i b16 = X8_B(36, a5, c4, r1), b01 = X8_B(44, a6, c0, r2), b11 = X8_B(6, a7, c1, r3), b21 = X8_B(55, a8, c2, r4);
        const __m512i b06 = X8_B(20, a9, c3, r0), b07 = X8_B(3, a10, c4, r1), b17 = X8_B(10, a11, c0, r2), b02 = X8_B(43, a12, c1, r3);
        const __m512i b12 = X8_B(25, a13, c2, r4), b22 = X8_B(39, a14, c3, r0), b23 = X8_B(41, a15, c4, r1), b08 = X8_B(45, a16, c0, r2);
        const __m512i b18 = X8_B(15, a17, c1, r3), b03 = X8_B(21, a18, c2, r4), b13 = X8_B(8, a19, c3, r0), b14 = X8_B(18, a20, c4, r1);
        const __m512i b24 = X8_B(2, a21, c0, r2), b09 = X8_B(61, a22, c1, r3), b19 = X8_B(56, a23, c2, r4), b04 = X8_B(14, a24, c3, r0);
        a0 = _mm512_xor_si512(X8_CHI(b00, b01, b02), _mm512_set1_epi64((long long)RC8[r]));
        a1 = X8_CHI(b01, b02, b03); a2 = X8_CHI(b02, b03, b04); a3 = X8_CHI(b03, b04, b00); a4 = X8_CHI(b04, b00, b01);
        a5 = X8_CHI(b05, b06, b07); a6 = X8_CHI(b06, b07, b08); a7 = X8_CHI(b07, b08, b09); a8 = X8_CHI(b08, b09, b05);
        a9 = X8_CHI(b09, b05, b06); a10 = X8_CHI(b10, b11, b12); a11 = X8_CHI(b11, b12, b13); a12 = X8_CHI(b12, b13, b14);
        a13 = X8_CHI(b13, b14, b10); a14 = X8_CHI(b14, b10, b11); a15 = X8_CHI(b15, b16, b17); a16 = X8_CHI(b16, b17, b18);
        a17 = X8_CHI(b17, b18, b19); a18 = X8_CHI(b18, b19, b15); a19 = X8_CHI(b19, b15, b16); a20 = X8_CHI(b20, b21, b22);
        a21 = X8_CHI(b21, b22, b23); a22 = X8_CHI(b22, b23, b24); a23 = X8_CHI(b23, b24, b20); a24 = X8_CHI(b24, b20, b21);
    }
#define X8_ST(i, v) _mm512_store_si512(st[i], (v))
    X8_ST(0, a0); X8_ST(1, a1); X8_ST(2, a2); X8_ST(3, a3); X8_ST(4, a4); X8_ST(5, a5); X8_ST(6, a6); X8_ST(7, a7); X8_ST(8, a8);
    X8_ST(9, a9); X8_ST(10, a10); X8_ST(11, a11); X8_ST(12, a12); X8_ST(13, a13); X8_ST(14, a14); X8_ST(15, a15); X8_ST(16, a16);
    X8_ST(17, a17); X8_ST(18, a18); X8_ST(19, a19); X8_ST(20, a20); X8_ST(21, a21); X8_ST(22, a22); X8_ST(23, a23); X8_ST(24, a24);
}

struct Job {  // absorb `count` records (tag || LE64((start + k) mod p)) into the sponge (st, pos)
    uint64_t *st;
    size_t *pos;
    const uint8_t *tag;
    size_t tag_len;
    uint64_t start, count;
    std::mutex m;
    std::condition_variable cv;
    bool done = false;
};

// the byte stream of one job, materialised a chunk of records at a time (as Transcript::append_tagged_counter does)
struct Stream {
    static constexpr size_t CHUNK_RECORDS = 272;  // 38 blocks of 19-byte records per refill
    Job *job = nullptr;
    uint64_t v = 0, left = 0;  // next counter value (mod p), records not yet materialised
    size_t rec = 0, rd = 0, wr = 0;
    uint8_t tmpl[32];  // the tag, zero-padded: a record is one fixed-size copy of it + the counter (tags of <= 24 bytes)
    std::vector<uint8_t> buf;

    void open(Job *j) {
        job = j;
        rec = j->tag_len + 8;
        v = j->start % (uint64_t)P;
        left = j->count;
        rd = wr = 0;
        memset(tmpl, 0, sizeof(tmpl));
        memcpy(tmpl, j->tag, j->tag_len < 24 ? j->tag_len : 24);
        buf.resize(CHUNK_RECORDS * rec + RATE + 64);
    }
    size_t avail() const { return wr - rd; }
    bool exhausted() const { return left == 0 && rd == wr; }
    // makes at least min(RATE, everything that is left) bytes readable at buf[rd]
    void fill() {
        if (avail() >= RATE || left == 0) return;
        if (rd) {
            memmove(buf.data(), buf.data() + rd, wr - rd);
            wr -= rd;
            rd = 0;
        }
        size_t m = (buf.size() - wr - 32) / rec;
        if (m > left) m = (size_t)left;
        uint8_t *q = buf.data() + wr;
        const size_t tl = job->tag_len;
        if (tl <= 24) {  // (the buffer has 32 bytes of slack: the tail of each copy is overwritten by the counter and the next record)
            for (size_t k = 0; k < m; k++, q += rec) {
                memcpy(q, tmpl, 32);
                memcpy(q + tl, &v, 8);
                if (++v == (uint64_t)P) v = 0;
            }
        } else {
            for (size_t k = 0; k < m; k++, q += rec) {
                memcpy(q, job->tag, tl);
                memcpy(q + tl, &v, 8);
                if (++v == (uint64_t)P) v = 0;
            }
        }
        wr += m * rec;
        left -= m;
    }
};

class Service {
  public:
    ~Service() { resize(0); }

    void resize(int n) {
        std::unique_lock<std::mutex> lk(m_);
        if (n < 0) n = 0;
        if (n > 0 && !cpu_has_avx512f()) n = 0;  // no 8-way permutation on this CPU: the sequential path stays
        if ((int)threads_.size() == n) return;
        // stop the running servers (they finish the jobs they hold and the queue; run() refuses new ones meanwhile), then
        // start the new set
        enabled_.store(false, std::memory_order_release);
        stop_ = true;
        cv_.notify_all();
        std::vector<std::thread> old;
        old.swap(threads_);
        lk.unlock();
        for (auto &t : old) t.join();
        lk.lock();
        stop_ = false;
        for (int i = 0; i < n; i++) threads_.emplace_back([this] { serve(); });
        enabled_.store(n > 0, std::memory_order_release);
    }
    bool enabled() const { return enabled_.load(std::memory_order_acquire); }

    // false: no server is accepting work (none started, or being stopped) -- the caller absorbs sequentially
    bool run(Job &j) {
        {
            std::lock_guard<std::mutex> lk(m_);
            if (threads_.empty() || stop_) return false;
            queue_.push_back(&j);
            pending_.fetch_add(1, std::memory_order_release);
        }
        cv_.notify_one();
        std::unique_lock<std::mutex> lk(j.m);
        j.cv.wait(lk, [&] { return j.done; });
        return true;
    }

  private:
    void serve() {
        alignas(64) uint64_t S[25][SLOTS];
        memset(S, 0, sizeof(S));
        Stream slot[SLOTS];
        size_t pos[SLOTS] = {0};
        int active = 0;
        for (;;) {
            if (active == 0 || (active < SLOTS && pending_.load(std::memory_order_acquire) > 0)) {
                // admit waiting jobs into free slots; sleep when there is nothing to do
                std::unique_lock<std::mutex> lk(m_);
                if (active == 0) cv_.wait(lk, [&] { return stop_ || !queue_.empty(); });
                if (stop_ && active == 0 && queue_.empty()) return;
                for (int s = 0; s < SLOTS && !queue_.empty(); s++) {
                    if (slot[s].job) continue;
                    Job *j = queue_.front();
                    queue_.pop_front();
                    pending_.fetch_sub(1, std::memory_order_relaxed);
                    slot[s].open(j);
                    pos[s] = *j->pos;
                    for (int w = 0; w < 25; w++) S[w][s] = j->st[w];
                    active++;
                }
            }
            if (active == 1 && pending_.load(std::memory_order_acquire) == 0) {
                // a lone transcript: whole blocks through the single-state permutation (the 8-way one costs ~15 % more per
                // call, slot bookkeeping included), a batch at a time so that a newcomer is admitted within microseconds
                int s = 0;
                while (!slot[s].job) s++;
                Stream &st = slot[s];
                if (pos[s] == 0) {
                    uint64_t one[25];
                    for (int w = 0; w < 25; w++) one[w] = S[w][s];
                    for (int k = 0; k < 64; k++) {
                        st.fill();
                        if (st.avail() < RATE) break;
                        const uint8_t *src = st.buf.data() + st.rd;
                        for (int w = 0; w < 17; w++) {
                            uint64_t x;
                            memcpy(&x, src + 8 * w, 8);
                            one[w] ^= x;
                        }
                        st.rd += RATE;
                        host_keccak_permute(one, 0);
                    }
                    for (int w = 0; w < 25; w++) S[w][s] = one[w];
                }
            }
            bool permute = false;
            for (int s = 0; s < SLOTS; s++) {
                Stream &st = slot[s];
                if (!st.job) continue;
                st.fill();
                const size_t need = RATE - pos[s];
                size_t n = st.avail() < need ? st.avail() : need;
                const uint8_t *src = st.buf.data() + st.rd;
                if (pos[s] == 0 && n == RATE) {  // a whole block: 17 words into the slot's column (eight blocks by one
                    // vpgatherqq per state row instead: 39.0 against 38.0 ms per batch of 8 -- tools/sponge_service_rate.py)
                    for (int w = 0; w < 17; w++) {
                        uint64_t x;
                        memcpy(&x, src + 8 * w, 8);
                        S[w][s] ^= x;
                    }
                } else if (n) {  // the first or the last, partial block
                    uint8_t blk[RATE] = {0};
                    memcpy(blk + pos[s], src, n);
                    for (int w = 0; w < 17; w++) {
                        uint64_t x;
                        memcpy(&x, blk + 8 * w, 8);
                        S[w][s] ^= x;
                    }
                }
                st.rd += n;
                pos[s] += n;
                if (pos[s] == RATE) {
                    pos[s] = 0;
                    permute = true;  // this slot's block is complete
                } else {             // the stream ended inside a block: hand the sponge back as it is
                    retire(S, s, slot, pos);
                    active--;
                }
            }
            if (permute) keccak_f1600_x8(S);
            for (int s = 0; s < SLOTS; s++)  // streams that ended exactly on a block boundary
                if (slot[s].job && slot[s].exhausted() && pos[s] == 0) {
                    retire(S, s, slot, pos);
                    active--;
                }
        }
    }

    static void retire(uint64_t S[25][SLOTS], int s, Stream *slot, size_t *pos) {
        Job *j = slot[s].job;
        for (int w = 0; w < 25; w++) j->st[w] = S[w][s];
        *j->pos = pos[s];
        slot[s].job = nullptr;
        std::lock_guard<std::mutex> lk(j->m);  // the job lives on the waiter's stack: notify before the lock is released,
        j->done = true;                        // the waiter cannot return (and destroy it) until then
        j->cv.notify_one();
    }

    std::mutex m_;
    std::condition_variable cv_;
    std::deque<Job *> queue_;
    std::vector<std::thread> threads_;
    bool stop_ = false;
    std::atomic<int> pending_{0};
    std::atomic<bool> enabled_{false};
};

Service &service() {
    static Service s;
    return s;
}

}  // namespace

void host_sponge_servers(int n) { service().resize(n); }
bool host_sponge_batching() { return service().enabled(); }

bool host_sponge_absorb_tagged(uint64_t st[25], size_t *pos, const uint8_t *tag, size_t tag_len, uint64_t start, uint64_t count) {
    Job j;
    j.st = st;
    j.pos = pos;
    j.tag = tag;
    j.tag_len = tag_len;
    j.start = start;
    j.count = count;
    return service().run(j);
}

void host_keccak_permute_x8(uint64_t st[25][8]) { keccak_f1600_x8(st); }

}  // namespace zk
