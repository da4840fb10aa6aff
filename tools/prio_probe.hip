// Does s_setprio let a lone hashing wave through on SIMDs that are full of other hashing waves?  (round 4: the small Merkle levels
// -- a few waves, one permutation each -- take 35-85 us next to other proofs' big levels and 16-25 us alone.)
// Background: 3 x 3072 workgroups of chained node hashes on three streams (the chip stays full for ~100 ms).  Foreground, on a fourth
// stream, one after the other: a launch of G workgroups x 256 threads doing ONE permutation chain of C hashes, at priority 0 and 3,
// with and without the re-arm pauses.  Reported: the foreground launch's duration (events), alone and in company.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Izigz_amd/csrc -Iinclude -o tools/bin/prio_probe tools/prio_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>

#include "keccak.hpp"

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)
using namespace zk;

__global__ __launch_bounds__(256) void k_bg(unsigned long long *out, int iters) {
    Digest a{{blockIdx.x, threadIdx.x, 3, 4}}, b{{5, 6, 7, 8}};
#pragma unroll 1
    for (int i = 0; i < iters; i++) { a = sha3_node<true>(a, b); b.w[0] += 1; }
    if (a.w[0] == 1) out[0] = a.w[1];
}
template <int PRIO, bool PAUSE>
__global__ __launch_bounds__(256) void k_fg(unsigned long long *out, int iters) {
    __builtin_amdgcn_s_setprio(PRIO);
    Digest a{{blockIdx.x, threadIdx.x, 3, 4}}, b{{5, 6, 7, 8}};
#pragma unroll 1
    for (int i = 0; i < iters; i++) { a = sha3_node<PAUSE>(a, b); b.w[0] += 1; }
    if (a.w[0] == 1) out[1] = a.w[1];
}

template <int PRIO, bool PAUSE>
static float fg(hipStream_t s, unsigned G, int chain, unsigned long long *out) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f, sum = 0;
    const int reps = 8;
    for (int r = 0; r < reps; r++) {
        hipExtLaunchKernelGGL((k_fg<PRIO, PAUSE>), dim3(G), dim3(256), 0, s, e0, e1, 0, out, chain);
        CK(hipStreamSynchronize(s));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        sum += ms;
        best = ms < best ? ms : best;
    }
    return sum / reps * 1e3f;
}

int main() {
    CK(hipSetDevice(0));
    unsigned long long *out;
    CK(hipMalloc((void **)&out, 64));
    hipStream_t bg[3], f;  // (three + one: the process has four hardware queues, and a launch waits for everything before it in its queue)
    for (auto &s : bg) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&f, hipStreamNonBlocking));
    for (int company = 0; company < 2; company++) {
        if (company)
            for (auto &s : bg) hipLaunchKernelGGL(k_bg, dim3(3072), dim3(256), 0, s, out, 1600);  // ~0.4 s of full chip
        for (unsigned G : {4u, 48u}) {
            printf("%s, %u workgroups x 1 hash:  prio 0 pause %.1f us | prio 3 pause %.1f us | prio 0 no pause %.1f us | prio 3 no pause %.1f us\n",
                   company ? "in company" : "alone", G, fg<0, true>(f, G, 1, out), fg<3, true>(f, G, 1, out), fg<0, false>(f, G, 1, out),
                   fg<3, false>(f, G, 1, out));
            printf("%s, %u workgroups x 4 hashes: prio 0 pause %.1f us | prio 3 pause %.1f us | prio 0 no pause %.1f us | prio 3 no pause %.1f us\n",
                   company ? "in company" : "alone", G, fg<0, true>(f, G, 4, out), fg<3, true>(f, G, 4, out), fg<0, false>(f, G, 4, out),
                   fg<3, false>(f, G, 4, out));
        }
        CK(hipDeviceSynchronize());
    }
    return 0;
}
