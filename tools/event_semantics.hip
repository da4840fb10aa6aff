// What do the two events of hipExtLaunchKernelGGL measure, and can they be placed on ONE time axis across streams?
// (the in-process event pairs read 1.2-1.5x longer than rocprofv3's kernel durations.)
//
// Ground truth inside the kernel: every workgroup takes wall_clock64() (the 100 MHz constant clock, the same for every CU,
// stream and process) when it starts and when it ends; the launch's true interval is [min start, max end].  Compared with
//   D  = elapsed(start_ev, stop_ev)                 what zigz_kernel_stats sums
//   A  = elapsed(epoch, start_ev), B = elapsed(epoch, stop_ev)   candidates for absolute times (epoch: a recorded event)
// for back-to-back launches in one stream (does a launch's "start" include the wait behind its predecessor?), for launches
// on several streams at once (does it include the wait for wave slots?), and -- run the same binary under
// `rocprofv3 --kernel-trace` -- with the profiler's start / end of the same dispatches (kernel names carry the launch id).
//
//   hipcc --offload-arch=gfx950 -O2 -o event_semantics tools/event_semantics.hip && ./event_semantics
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

// every workgroup spins for `ticks` of the constant clock (10 ns each); slot[0] = min start, slot[1] = max end
template <int ID>
__global__ void k_spin(unsigned long long *slot, unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) atomicMin(&slot[0], t0);
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) atomicMax(&slot[1], wall_clock64());
}

// (e) the bench's shape: many streams, chains of dependent launches that oversubscribe the chip.  Every workgroup stores its own
// begin / end (no atomics: 1024 same-address atomics at the end of a 25 us kernel are 12 us of kernel, case d).
__global__ void k_spin_log(unsigned long long *log, unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        log[2 * blockIdx.x] = t0;
        log[2 * blockIdx.x + 1] = wall_clock64();
    }
}

struct Launch {
    const char *what;
    int stream;
    hipEvent_t s, e;
    unsigned long long *slot;
};
static std::vector<Launch> ls;
static void launch(hipStream_t *st, unsigned long long *slots, const char *what, int stream, int id, unsigned grid,
                   unsigned long long ticks) {
    Launch l{what, stream, nullptr, nullptr, slots + 2 * ls.size()};
    CK(hipEventCreate(&l.s));
    CK(hipEventCreate(&l.e));
    switch (id) {
    case 1: hipExtLaunchKernelGGL(k_spin<1>, dim3(grid), dim3(256), 0, st[stream], l.s, l.e, 0, l.slot, ticks); break;
    case 2: hipExtLaunchKernelGGL(k_spin<2>, dim3(grid), dim3(256), 0, st[stream], l.s, l.e, 0, l.slot, ticks); break;
    case 3: hipExtLaunchKernelGGL(k_spin<3>, dim3(grid), dim3(256), 0, st[stream], l.s, l.e, 0, l.slot, ticks); break;
    default: hipExtLaunchKernelGGL(k_spin<4>, dim3(grid), dim3(256), 0, st[stream], l.s, l.e, 0, l.slot, ticks); break;
    }
    CK(hipGetLastError());
    ls.push_back(l);
}

int main() {
    CK(hipSetDevice(0));
    const int NS = 6;
    hipStream_t st[NS];
    for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    // (the slots live in HBM: atomics on pinned host memory cross PCIe, ~1 us each on one address, and a launch is not over
    // until they are -- the first version of this tool measured exactly that: 75 us of "end" for 64 workgroups)
    unsigned long long *slots, h_slots[128];
    CK(hipMalloc((void **)&slots, 64 * 16));
    for (int i = 0; i < 128; i++) h_slots[i] = (i & 1) ? 0ull : ~0ull;
    CK(hipMemcpy(slots, h_slots, sizeof(h_slots), hipMemcpyHostToDevice));
    hipEvent_t epoch;
    CK(hipEventCreate(&epoch));
    // the constant clock at the epoch: a one-thread kernel right behind the epoch record
    CK(hipEventRecord(epoch, st[0]));
    hipLaunchKernelGGL(k_spin<0>, dim3(1), dim3(64), 0, st[0], slots + 126, 1ull);
    CK(hipStreamSynchronize(st[0]));
    auto sync = [&]() { for (auto &s : st) CK(hipStreamSynchronize(s)); };
    // (a) one stream, three launches back to back: 200, 100, 50 us, small grids (no contention for wave slots)
    launch(st, slots, "a: stream 0, 200 us, 64 wgs", 0, 1, 64, 20000);
    launch(st, slots, "a: stream 0, 100 us, 64 wgs (behind the 200 us launch)", 0, 2, 64, 10000);
    launch(st, slots, "a: stream 0,  50 us, 64 wgs (behind both)", 0, 3, 64, 5000);
    sync();
    // (b) five streams at once, each a grid that alone fills the chip several times over (8192 workgroups x 256 threads x
    // 20 us): launches overlap and wait for wave slots
    for (int k = 0; k < 5; k++) launch(st, slots, "b: one of five streams at once, 8192 wgs x 20 us", 1 + k, 4, 8192, 2000);
    sync();
    // (c) the same grid alone
    launch(st, slots, "c: alone, 8192 wgs x 20 us", 1, 4, 8192, 2000);
    sync();
    // (d) short launches like the bench's: 1024 workgroups x 25 us alone, then three of them on three streams at once
    launch(st, slots, "d: alone, 1024 wgs x 25 us", 1, 4, 1024, 2500);
    sync();
    launch(st, slots, "d: alone again, 1024 wgs x 25 us", 1, 4, 1024, 2500);
    sync();
    for (int k = 0; k < 3; k++) launch(st, slots, "d: one of three streams at once, 1024 wgs x 25 us", 1 + k, 4, 1024, 2500);
    sync();
    {   // (e) 12 streams x 20 dependent launches of 1024 workgroups x 40 us: ~6 launches' worth of workgroups resident at a time
        const int S = 12, L = 20, WG = 1024;
        hipStream_t ss[S];
        for (auto &x : ss) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
        unsigned long long *log;
        CK(hipMalloc((void **)&log, (size_t)S * L * WG * 16));
        std::vector<hipEvent_t> es(S * L), ee(S * L);
        for (auto &x : es) CK(hipEventCreate(&x));
        for (auto &x : ee) CK(hipEventCreate(&x));
        for (int l = 0; l < L; l++)
            for (int k = 0; k < S; k++)
                hipExtLaunchKernelGGL(k_spin_log, dim3(WG), dim3(256), 0, ss[k], es[k * L + l], ee[k * L + l], 0, log + (size_t)(k * L + l) * WG * 2,
                                      4000ull);
        for (auto &x : ss) CK(hipStreamSynchronize(x));
        std::vector<unsigned long long> h((size_t)S * L * WG * 2);
        CK(hipMemcpy(h.data(), log, h.size() * 8, hipMemcpyDeviceToHost));
        double sum_true = 0, sum_d = 0, max_ratio = 0;
        std::vector<std::pair<double, double>> iv_true, iv_ev;
        for (int i = 0; i < S * L; i++) {
            unsigned long long a = ~0ull, b = 0;
            for (int w = 0; w < WG; w++) {
                a = h[((size_t)i * WG + w) * 2] < a ? h[((size_t)i * WG + w) * 2] : a;
                b = h[((size_t)i * WG + w) * 2 + 1] > b ? h[((size_t)i * WG + w) * 2 + 1] : b;
            }
            float d = 0, e0 = 0, e1 = 0;
            CK(hipEventElapsedTime(&d, es[i], ee[i]));
            CK(hipEventElapsedTime(&e0, epoch, es[i]));
            CK(hipEventElapsedTime(&e1, epoch, ee[i]));
            const double t = (double)(b - a) * 0.01;
            sum_true += t;
            sum_d += d * 1e3;
            if (d * 1e3 / t > max_ratio) max_ratio = d * 1e3 / t;
            iv_true.push_back({(double)a * 0.01, (double)b * 0.01});
            iv_ev.push_back({e0 * 1e3, e1 * 1e3});
        }
        auto uni = [](std::vector<std::pair<double, double>> v) {
            std::sort(v.begin(), v.end());
            double tot = 0, cs = v[0].first, ce = v[0].second;
            for (auto &x : v) {
                if (x.first > ce) { tot += ce - cs; cs = x.first; ce = x.second; }
                else if (x.second > ce) ce = x.second;
            }
            return tot + ce - cs;
        };
        printf("e: %d streams x %d dependent launches (1024 wgs x 40 us): per launch first-wave-start..last-wave-end %.1f us, event pair %.1f us "
               "(x%.2f, worst x%.2f); union of the intervals: in-kernel clock %.0f us, events %.0f us\n",
               S, L, sum_true / (S * L), sum_d / (S * L), sum_d / sum_true, max_ratio, uni(iv_true), uni(iv_ev));
    }
    CK(hipMemcpy(h_slots, slots, sizeof(h_slots), hipMemcpyDeviceToHost));
    const double tick_us = 0.01;
    const unsigned long long t_epoch = h_slots[126];
    printf("%-58s %10s %10s | %10s %10s %10s | %10s %10s\n", "launch", "true_us", "D_us", "true_t0", "A_us", "B-D_us", "true_t1", "B_us");
    for (auto &l : ls) {
        float d = 0, a = 0, b = 0;
        CK(hipEventElapsedTime(&d, l.s, l.e));
        hipError_t ea = hipEventElapsedTime(&a, epoch, l.s), eb = hipEventElapsedTime(&b, epoch, l.e);
        if (ea != hipSuccess) a = -1;
        if (eb != hipSuccess) b = -1;
        (void)hipGetLastError();
        const unsigned long long *hs = h_slots + (l.slot - slots);
        const double t0 = (double)(long long)(hs[0] - t_epoch) * tick_us, t1 = (double)(long long)(hs[1] - t_epoch) * tick_us;
        printf("%-58s %10.1f %10.1f | %10.1f %10.1f %10.1f | %10.1f %10.1f\n", l.what, t1 - t0, d * 1e3, t0, a * 1e3, (b - d) * 1e3, t1, b * 1e3);
    }
    return 0;
}
