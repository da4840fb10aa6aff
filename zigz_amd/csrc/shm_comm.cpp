// Same-node all-gather over POSIX shared memory: the built-in zigz_allgather_fn of include/zigz_hip.h.
//
// The exchanges of the sharded prover are host-resident by construction -- block sums, roots and openings are read
// back for the SHA3 transcript before they are exchanged -- and tiny (16 B .. 64 KiB), i.e. latency-bound.  Between
// the <= 8 ranks of one node a mailbox in shared memory moves them in a few microseconds; a device collective would
// add an H2D and a D2H around a transfer this small.  (Hosts that span nodes bind the hook to RCCL / MPI instead:
// rccl_comm.cpp.)
//
// Layout: header | world sequence counters (one cache line each) | 2 x world slots of max_bytes.  All-gather number s
// uses slot set s & 1: write my slot, publish seq = s, wait until every rank has published >= s, copy all slots out.
// A rank can only start s + 2 (reusing the slot set of s) after all ranks published s + 1, i.e. after they all finished
// reading s -- two slot sets suffice and one wait per all-gather.  Waits time out (a dead rank does not hang the rest).
//
// Creation is a collective with a HANDSHAKE, because names get reused (a port number, a job id) and a crashed job leaves its
// segment behind: an attaching rank writes a random token into its hello word of whatever segment it found under the name
// and accepts that segment only when rank 0 -- the LIVE rank 0 of this create call, which polls the hello words of the
// segment it has just made -- echoes the token.  A leftover segment has nobody to answer: the attacher lets go of it after
// a short wait and opens the name again.  Rank 0 returns when every rank has been acknowledged, so nobody's unlink (rank 0
// removes the name it finds before creating its own) can take a segment away from ranks that are still attaching to it.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <new>

#include "zigz_hip.h"

namespace {
constexpr uint32_t MAGIC = 0x5A49475Au;  // "ZIGZ"
struct alignas(64) Seq {
    std::atomic<uint64_t> v;
};
constexpr int MAX_WORLD = 64;
struct Header {
    std::atomic<uint32_t> ready;
    uint32_t world;
    uint64_t max_bytes;
    std::atomic<uint64_t> hello[MAX_WORLD];  // rank r: a random token, written by the attacher
    std::atomic<uint64_t> ack[MAX_WORLD];    // rank 0 echoes the token: "this segment is the live one"
};
static_assert(sizeof(Header) <= 4096, "the header has a page of its own");
uint64_t random_token() {
    uint64_t t = 0;
    int fd = open("/dev/urandom", O_RDONLY);
    if (fd >= 0) {
        if (read(fd, &t, sizeof(t)) != (ssize_t)sizeof(t)) t = 0;
        close(fd);
    }
    timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    t ^= ((uint64_t)getpid() << 32) ^ (uint64_t)ts.tv_nsec ^ ((uint64_t)ts.tv_sec << 20);
    return t ? t : 1;
}
double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
}  // namespace

struct zigz_shm_comm {
    char name[96];
    int rank, world;
    size_t max_bytes, map_bytes;
    uint8_t *base;
    Header *hdr;
    Seq *seq;
    uint8_t *slots;
    uint64_t next;  // sequence number of the next all-gather (starts at 1)
    double timeout_s;
    ino_t ino;  // rank 0: the segment it created (destroy unlinks the name only while it still refers to it)
    dev_t dev;
};

static size_t layout_bytes(int world, size_t max_bytes) {
    return 4096 + (size_t)world * sizeof(Seq) + 2 * (size_t)world * max_bytes;
}

extern "C" zigz_status zigz_shm_comm_create(const char *name, int rank, int world, size_t max_bytes, double timeout_s,
                                            zigz_shm_comm **out) {
    if (!name || !out || world < 1 || world > MAX_WORLD || rank < 0 || rank >= world || max_bytes == 0 || strlen(name) > 80)
        return ZIGZ_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    zigz_shm_comm *c = new (std::nothrow) zigz_shm_comm();
    if (!c) return ZIGZ_ERR_OUT_OF_MEMORY;
    snprintf(c->name, sizeof(c->name), "/%s", name);
    c->rank = rank;
    c->world = world;
    c->max_bytes = (max_bytes + 63) & ~(size_t)63;
    c->map_bytes = layout_bytes(world, c->max_bytes);
    c->next = 1;
    c->timeout_s = timeout_s > 0 ? timeout_s : 60.0;
    const double deadline = now_s() + c->timeout_s;
    auto bind = [&](void *m) {
        c->base = (uint8_t *)m;
        c->hdr = (Header *)c->base;
        c->seq = (Seq *)(c->base + 4096);
        c->slots = c->base + 4096 + (size_t)world * sizeof(Seq);
    };
    if (rank == 0) {
        shm_unlink(c->name);  // whatever is left under the name is not ours (its ranks, if alive, have all attached already)
        int fd = shm_open(c->name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) {
            if (fd >= 0) { close(fd); shm_unlink(c->name); }
            delete c;
            return ZIGZ_ERR_INVALID_ARGUMENT;
        }
        struct stat st;
        if (fstat(fd, &st) == 0) { c->ino = st.st_ino; c->dev = st.st_dev; }
        void *m = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (m == MAP_FAILED) {
            shm_unlink(c->name);
            delete c;
            return ZIGZ_ERR_OUT_OF_MEMORY;
        }
        bind(m);  // ftruncate zero-fills: counters, hello and ack words start at 0
        c->hdr->world = (uint32_t)world;
        c->hdr->max_bytes = c->max_bytes;
        c->hdr->ready.store(MAGIC, std::memory_order_release);
        // acknowledge every attacher; return when all of them hold THIS segment
        int acked = 1;
        while (acked < world) {
            acked = 1;
            for (int r = 1; r < world; r++) {
                const uint64_t h = c->hdr->hello[r].load(std::memory_order_acquire);
                if (h) {
                    c->hdr->ack[r].store(h, std::memory_order_release);
                    acked++;
                }
            }
            if (acked == world) break;
            if (now_s() > deadline) {
                shm_unlink(c->name);
                munmap(c->base, c->map_bytes);
                delete c;
                return ZIGZ_ERR_BAD_STATE;
            }
            usleep(50);
        }
    } else {
        const uint64_t token = random_token();
        zigz_status why = ZIGZ_ERR_BAD_STATE;  // what to report if no live segment turns up
        for (;;) {
            if (now_s() > deadline) {
                delete c;
                return why;
            }
            int fd = shm_open(c->name, O_RDWR, 0600);
            if (fd < 0) { usleep(200); continue; }
            struct stat st;
            if (fstat(fd, &st) != 0 || (size_t)st.st_size < c->map_bytes) {  // not (yet) sized, or somebody else's
                close(fd);
                usleep(200);
                continue;
            }
            void *m = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            close(fd);
            if (m == MAP_FAILED) {
                delete c;
                return ZIGZ_ERR_OUT_OF_MEMORY;
            }
            bind(m);
            bool live = false;
            if (c->hdr->ready.load(std::memory_order_acquire) == MAGIC) {
                if (c->hdr->world != (uint32_t)world || c->hdr->max_bytes != c->max_bytes) {
                    why = ZIGZ_ERR_INVALID_ARGUMENT;  // a mismatch with a LIVE segment is a caller error; with a leftover it is not
                } else {
                    c->hdr->hello[rank].store(token, std::memory_order_release);
                    const double until = now_s() + 0.05;  // a live rank 0 answers within microseconds; then ask the name again
                    while (now_s() < until) {
                        if (c->hdr->ack[rank].load(std::memory_order_acquire) == token) { live = true; break; }
                        usleep(20);
                    }
                }
            }
            if (live) break;
            munmap(c->base, c->map_bytes);
            c->base = nullptr;
            usleep(200);
        }
    }
    *out = c;
    return ZIGZ_OK;
}

// zigz_allgather_fn: user = zigz_shm_comm*
extern "C" int zigz_shm_allgather(void *user, const void *send, size_t bytes, void *recv) {
    zigz_shm_comm *c = (zigz_shm_comm *)user;
    if (!c || !send || !recv || bytes > c->max_bytes) return 1;
    const uint64_t s = c->next++;
    uint8_t *set = c->slots + (size_t)(s & 1) * (size_t)c->world * c->max_bytes;
    memcpy(set + (size_t)c->rank * c->max_bytes, send, bytes);
    c->seq[c->rank].v.store(s, std::memory_order_release);
    const double deadline = now_s() + c->timeout_s;
    for (int r = 0; r < c->world; r++) {
        unsigned spins = 0;
        while (c->seq[r].v.load(std::memory_order_acquire) < s) {
            if ((++spins & 1023) == 0) {
                if (now_s() > deadline) return 2;  // a rank is gone: fail instead of hanging
                if (spins > (1u << 16)) usleep(50);
            }
        }
        memcpy((uint8_t *)recv + (size_t)r * bytes, set + (size_t)r * c->max_bytes, bytes);
    }
    return 0;
}

extern "C" void zigz_shm_comm_destroy(zigz_shm_comm *c) {
    if (!c) return;
    if (c->rank == 0) {  // mappings of the other ranks stay valid until they unmap
        int fd = shm_open(c->name, O_RDWR, 0600);
        if (fd >= 0) {
            struct stat st;
            const bool mine = fstat(fd, &st) == 0 && st.st_ino == c->ino && st.st_dev == c->dev;
            close(fd);
            if (mine) shm_unlink(c->name);  // (a later job may have taken the name: leave that one alone)
        }
    }
    munmap(c->base, c->map_bytes);
    delete c;
}
