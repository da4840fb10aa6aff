// ThreadSanitizer driver for the shared-memory all-gather (zigz_amd/csrc/shm_comm.cpp): `world` threads of ONE process
// each attach to the same segment as a rank and run `iters` back-to-back exchanges of varying size, checking every byte
// received.  Built by tests/test_shard_gloo.py with -fsanitize=thread: the slot reuse (two slot sets, one wait per
// exchange) and the release / acquire pairs on the sequence counters are what is under test.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <atomic>
#include <thread>
#include <vector>

#include "zigz_hip.h"

static std::atomic<int> failures{0};

static void rank_main(const char *name, int rank, int world, int iters) {
    zigz_shm_comm *c = nullptr;
    if (zigz_shm_comm_create(name, rank, world, 4096, 30.0, &c) != ZIGZ_OK) {
        failures++;
        return;
    }
    std::vector<uint8_t> send(4096), recv((size_t)world * 4096);
    for (int it = 0; it < iters; it++) {
        const size_t n = 1 + (size_t)((it * 131) % 4000);
        for (size_t j = 0; j < n; j++) send[j] = (uint8_t)(rank * 31 + it + (int)j);
        if (zigz_shm_allgather(c, send.data(), n, recv.data()) != 0) {
            failures++;
            break;
        }
        for (int r = 0; r < world; r++)
            for (size_t j = 0; j < n; j++)
                if (recv[(size_t)r * n + j] != (uint8_t)(r * 31 + it + (int)j)) {
                    failures++;
                    j = n;
                }
    }
    zigz_shm_comm_destroy(c);
}

int main(int argc, char **argv) {
    const int world = argc > 1 ? atoi(argv[1]) : 4, iters = argc > 2 ? atoi(argv[2]) : 2000;
    char name[64];
    snprintf(name, sizeof name, "zigz_tsan_%d", (int)getpid());
    std::vector<std::thread> th;
    for (int r = 0; r < world; r++) th.emplace_back(rank_main, name, r, world, iters);
    for (auto &t : th) t.join();
    printf("%s: world %d, %d exchanges, %d failure(s)\n", failures.load() ? "FAIL" : "ok", world, iters, failures.load());
    return failures.load() ? 1 : 0;
}
