// Do two DIFFERENT fully unrolled Keccak kernels that share the chip slow each other down through the instruction cache?
// (A permutation is ~4 000 instructions, ~33 KB of code per kernel variant; the instruction cache is 64 KB per pair of CUs.
// In a batch of proofs k_level_hash<leaf>, k_level_hash<node>, its variant without pauses and k_merkle_top run side by side.)
//
// Each kernel hashes in registers (no memory traffic to speak of): A = leaf hashes chained, B = node hashes chained, C = node
// hashes without re-arm pauses.  Alone; two launches of the SAME kernel on two streams; two DIFFERENT kernels on two streams; three.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Izigz_amd/csrc -Iinclude -o tools/bin/icache_corun tools/icache_corun.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <chrono>
#include <vector>

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

__global__ __launch_bounds__(256) void k_A(unsigned long long *out, int iters) {
    unsigned long long v = blockIdx.x * 256 + threadIdx.x;
#pragma unroll 1
    for (int i = 0; i < iters; i++) v = sha3_leaf<true>(v).w[0];
    if (v == 1) out[0] = v;
}
__global__ __launch_bounds__(256) void k_B(unsigned long long *out, int iters) {
    Digest a{{blockIdx.x, threadIdx.x, 3, 4}}, b{{5, 6, 7, 8}};
#pragma unroll 1
    for (int i = 0; i < iters; i++) { a = sha3_node<true>(a, b); b.w[0] += 1; }
    if (a.w[0] == 1) out[0] = a.w[1];
}
__global__ __launch_bounds__(256) void k_C(unsigned long long *out, int iters) {
    Digest a{{blockIdx.x, threadIdx.x, 3, 4}}, b{{5, 6, 7, 8}};
#pragma unroll 1
    for (int i = 0; i < iters; i++) { a = sha3_node<false>(a, b); b.w[1] += 1; }
    if (a.w[0] == 1) out[0] = a.w[1];
}

typedef void (*kern_t)(unsigned long long *, int);

static double run(std::vector<kern_t> ks, unsigned wgs, int iters, unsigned long long *out) {
    std::vector<hipStream_t> st(ks.size());
    for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int rep = 0; rep < 2; rep++) {  // first round: warm-up
        if (rep == 1) CK(hipDeviceSynchronize());
        static std::chrono::steady_clock::time_point t0;
        t0 = std::chrono::steady_clock::now();
        for (size_t i = 0; i < ks.size(); i++) hipLaunchKernelGGL(ks[i], dim3(wgs), dim3(256), 0, st[i], out, iters);
        for (auto &s : st) CK(hipStreamSynchronize(s));
        if (rep == 1) {
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            for (auto &s : st) CK(hipStreamDestroy(s));
            return (double)ks.size() * wgs * 256 * iters / us * 1e-3;  // G permutations / s, all launches together
        }
    }
    return 0;
}

int main(int argc, char **argv) {
    CK(hipSetDevice(0));
    unsigned long long *out;
    CK(hipMalloc((void **)&out, 64));
    if (argc > 1) {  // sustained: B x 4 streams for `argv[1]` chained hashes per thread (40 = 10 ms, 4000 = 1 s): does the rate hold?
        for (int rep = 0; rep < (argc > 2 ? atoi(argv[2]) : 3); rep++)
            printf("sustained, %d chained hashes per thread, 4 streams x 3072 workgroups: %.2f G permutations/s\n", atoi(argv[1]),
                   run({k_B, k_B, k_B, k_B}, 3072, atoi(argv[1]), out));
        return 0;
    }
    const int iters = 40;
    // wgs per launch: the chip holds 256 CUs x 6 workgroups of these kernels; every launch alone would fill it
    for (unsigned wgs : {1536u, 3072u}) {
        printf("%u workgroups per launch, %d chained hashes per thread; G permutations/s over all launches:\n", wgs, iters);
        printf("  A alone %.2f   B alone %.2f   C alone %.2f\n", run({k_A}, wgs, iters, out), run({k_B}, wgs, iters, out), run({k_C}, wgs, iters, out));
        printf("  A+A %.2f   B+B %.2f   C+C %.2f\n", run({k_A, k_A}, wgs, iters, out), run({k_B, k_B}, wgs, iters, out), run({k_C, k_C}, wgs, iters, out));
        printf("  A+B %.2f   B+C %.2f   A+C %.2f\n", run({k_A, k_B}, wgs, iters, out), run({k_B, k_C}, wgs, iters, out), run({k_A, k_C}, wgs, iters, out));
        printf("  A+B+C %.2f   B+B+B %.2f   A+B+C+B %.2f   B+B+B+B %.2f\n", run({k_A, k_B, k_C}, wgs, iters, out), run({k_B, k_B, k_B}, wgs, iters, out),
               run({k_A, k_B, k_C, k_B}, wgs, iters, out), run({k_B, k_B, k_B, k_B}, wgs, iters, out));
    }
    return 0;
}
