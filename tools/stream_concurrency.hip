// How many dependent chains of SMALL kernels does the GPU run side by side?  (round 4: the non-hash launches of a proof --
// structure passes, small levels, tops, evaluations -- each use a fraction of the chip for 5-90 us; twelve contexts' chains
// should overlap freely.  In-order streams share a few hardware queues, and a packet with the barrier bit waits for everything
// before it in ITS QUEUE, other streams' packets included.)
//
// S streams x L dependent launches of G workgroups that spin T us each.  Unbounded concurrency: wall = L * (T + gap) whatever S.
// Capped at Q: wall ~ S / Q times that.  Run with GPU_MAX_HW_QUEUES = 2, 4 (default), 8, 16 in the environment.
//   hipcc --offload-arch=gfx950 -O2 -o stream_concurrency tools/stream_concurrency.hip && ./stream_concurrency
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <chrono>
#include <thread>
#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

__global__ void k_spin(unsigned long long ticks, unsigned long long *sink) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (ticks == ~0ull) *sink = t0;
}

// prio_classes > 1: the streams take turns through that many priorities (hipDeviceGetStreamPriorityRange), which have
// hardware queues of their own
static double run(int S, int L, unsigned G, double T_us, bool threads, int prio_classes = 1) {
    std::vector<hipStream_t> st(S);
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));  // lo: least priority (largest number), hi: greatest
    for (int k = 0; k < S; k++) {
        if (prio_classes <= 1) CK(hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking));
        else CK(hipStreamCreateWithPriority(&st[k], hipStreamNonBlocking, hi + (k % prio_classes) * (lo - hi) / (prio_classes - 1)));
    }
    const unsigned long long ticks = (unsigned long long)(T_us * 100.0);
    auto chain = [&](int k) {
        for (int l = 0; l < L; l++) hipLaunchKernelGGL(k_spin, dim3(G), dim3(256), 0, st[k], ticks, nullptr);
    };
    for (int k = 0; k < S; k++) chain(k);  // warm-up
    for (auto &s : st) CK(hipStreamSynchronize(s));
    const auto t0 = std::chrono::steady_clock::now();
    if (threads) {
        std::vector<std::thread> th;
        for (int k = 0; k < S; k++) th.emplace_back([&, k]() { chain(k); CK(hipStreamSynchronize(st[k])); });
        for (auto &t : th) t.join();
    } else {
        for (int l = 0; l < L; l++)
            for (int k = 0; k < S; k++) hipLaunchKernelGGL(k_spin, dim3(G), dim3(256), 0, st[k], ticks, nullptr);
        for (auto &s : st) CK(hipStreamSynchronize(s));
    }
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    for (auto &s : st) CK(hipStreamDestroy(s));
    return us;
}

int main(int argc, char **argv) {
    CK(hipSetDevice(0));
    const char *q = getenv("GPU_MAX_HW_QUEUES");
    printf("GPU_MAX_HW_QUEUES=%s\n", q ? q : "(default)");
    const int L = 50;
    for (int threads = 0; threads < 2; threads++)
        for (double T : {20.0, 80.0})
            for (unsigned G : {16u, 64u}) {
                printf("%s, %d launches per stream of %u workgroups x %.0f us:", threads ? "a thread per stream" : "one thread", L, G, T);
                for (int S : {1, 2, 4, 6, 8, 12, 16, 24}) {
                    const double us = run(S, L, G, T, threads != 0);
                    printf("  S=%d %.0f us (x%.1f side by side)", S, us, S * L * T / us);
                }
                printf("\n");
            }
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    printf("stream priorities: least %d .. greatest %d\n", lo, hi);
    for (int pc : {2, 3})
        for (double T : {20.0, 80.0}) {
            printf("one thread, %d priority classes in turn, %d launches per stream of 64 workgroups x %.0f us:", pc, L, T);
            for (int S : {2, 4, 6, 8, 12, 16, 24}) {
                const double us = run(S, L, 64, T, false, pc);
                printf("  S=%d %.0f us (x%.1f side by side)", S, us, S * L * T / us);
            }
            printf("\n");
        }
    return 0;
}
