// Per-launch timing of the Keccak Merkle kernels on the bench shape (43 columns x 2^20 leaves), outside the prover:
// the numbers DESIGN.md quotes for K5/K6.  Builds the kernels from source, so kernel experiments (-D flags) need no
// library rebuild.
//   hipcc --offload-arch=gfx950 -O3 -I zigz_amd/csrc -I include tools/merkle_rate.hip -o /tmp/merkle_rate && /tmp/merkle_rate
#include "../zigz_amd/csrc/kernels.hip"

#include <stdio.h>
#include <chrono>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char **argv) {
    const size_t ncols = 43;
    const unsigned nv = argc > 1 ? (unsigned)atoi(argv[1]) : 20;
    const size_t N = (size_t)1 << nv, nodes = 2 * N;
    uint32_t *d_vals;
    uint8_t *d_tree;
    CK(hipMalloc(&d_vals, ncols * N * 4));
    CK(hipMalloc(&d_tree, ncols * nodes * 32));
    std::vector<uint32_t> h(ncols * N);
    uint64_t s = 0x5A49475Aull;
    for (auto &x : h) { s += 0x9E3779B97F4A7C15ull; uint64_t z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; x = (uint32_t)((z ^ (z >> 31)) % zk::P); }
    CK(hipMemcpy(d_vals, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    // the eval pass of the prover reads the same columns right after the tree build: time it in that cache state
    const size_t fm = 1024, fnb = N / fm, fgroups = zk::radix_fold_groups(fnb);
    uint32_t *d_w;
    unsigned long long *d_part;
    CK(hipMalloc(&d_w, ncols * fnb * 4));
    CK(hipMemset(d_w, 0x11, ncols * fnb * 4));
    CK(hipMalloc(&d_part, ncols * fgroups * fm * 8));
    hipEvent_t f0, f1;
    CK(hipEventCreate(&f0));
    CK(hipEventCreate(&f1));
    hipEvent_t ev[64];
    for (auto &e : ev) CK(hipEventCreate(&e));
    {   // the fold issued 1 ms into another stream's leaves kernel (what an eval meets when other proofs are in flight)
        hipStream_t s2;
        CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        for (int rep = 0; rep < 4; rep++) {
            zk::launch_keccak_leaves(d_vals, N, N, N, d_tree, nodes, ncols, 0);
            const auto t0 = std::chrono::steady_clock::now();
            while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 1e-3 * (1 + rep % 2)) {}
            zk::launch_radix_fold(d_vals, N, fm, fnb, d_w, fnb, d_part, fgroups * fm, ncols, s2, f0, f1);
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, f0, f1));
            printf("fold inside a running leaves kernel: %.1f us\n", ms * 1e3);
        }
    }
    for (int rep = 0; rep < 3; rep++) {
        int k = 0;
        CK(hipEventRecord(ev[k++], 0));
        zk::launch_keccak_leaves(d_vals, N, N, N, d_tree, nodes, ncols, 0);
        CK(hipEventRecord(ev[k++], 0));
        unsigned l = 0;
        for (; l < nv; l++) {
            const size_t n_out = N >> (l + 1);
            if (n_out < 512) break;
            zk::launch_keccak_level(d_tree, nodes, 2 * N - 2 * (N >> l), 2 * N - 2 * (N >> (l + 1)), n_out, ncols, 0);
            CK(hipEventRecord(ev[k++], 0));
        }
        zk::launch_keccak_top(d_tree, nodes, N, l, nv, ncols, 0);
        CK(hipEventRecord(ev[k++], 0));
        zk::launch_radix_fold(d_vals, N, fm, fnb, d_w, fnb, d_part, fgroups * fm, ncols, 0, f0, f1);
        CK(hipDeviceSynchronize());
        if (rep == 0) continue;
        float fold_ms = 0;
        CK(hipEventElapsedTime(&fold_ms, f0, f1));
        printf("fold after the build: %.1f us | ", fold_ms * 1e3);
        float total = 0;
        CK(hipEventElapsedTime(&total, ev[0], ev[k - 1]));
        printf("rep %d: total %.3f ms = %.2f Gperm/s |", rep, total, (double)ncols * (2 * N - 1) / total / 1e6);
        for (int j = 1; j < k; j++) {
            float ms;
            CK(hipEventElapsedTime(&ms, ev[j - 1], ev[j]));
            const double perms = j == 1 ? (double)ncols * N : (j == k - 1 ? (double)ncols * ((N >> l) - 1) : (double)ncols * (N >> (j - 1)));
            printf(" %s %.3f ms (%.2f G/s)", j == 1 ? "leaves" : (j == k - 1 ? "top" : "lvl"), ms, perms / ms / 1e6);
        }
        printf("\n");
    }
    return 0;
}
