// How many small dependent commands per second does the stack sustain?  T host threads, one stream each, every thread launches
// chains of K trivial kernels (each depends on the previous one in its stream), eagerly or as one captured hipGraph per chain.
//   hipcc --offload-arch=gfx950 -O2 -pthread -o cmd_rate tools/cmd_rate.hip && ./cmd_rate [threads] [chain] [reps]
// The commit path of a proof is ~48 such commands; tools/gpu_bound_rate.py --debug-skip shows its floor (0.15 ms per proof
// without any hashing) is the command count, not the kernels' work.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

__global__ void k_tiny(unsigned *p, unsigned v) {
    if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += v;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 14, K = argc > 2 ? atoi(argv[2]) : 48, R = argc > 3 ? atoi(argv[3]) : 200;
    for (int mode = 0; mode < 2; mode++) {
        std::vector<std::thread> th;
        std::vector<double> secs(T);
        const double t0 = now();
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t] {
                hipStream_t s;
                CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
                unsigned *d;
                CK(hipMalloc(&d, 256));
                CK(hipMemsetAsync(d, 0, 256, s));
                hipGraphExec_t ge = nullptr;
                if (mode == 1) {
                    hipGraph_t g;
                    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                    for (int k = 0; k < K; k++) hipLaunchKernelGGL(k_tiny, dim3(64), dim3(256), 0, s, d, 1u);
                    CK(hipStreamEndCapture(s, &g));
                    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                }
                CK(hipStreamSynchronize(s));
                const double a = now();
                for (int r = 0; r < R; r++) {
                    if (mode == 0) for (int k = 0; k < K; k++) hipLaunchKernelGGL(k_tiny, dim3(64), dim3(256), 0, s, d, 1u);
                    else CK(hipGraphLaunch(ge, s));
                    if ((r & 3) == 3) CK(hipStreamSynchronize(s));  // a proof waits for its roots
                }
                CK(hipStreamSynchronize(s));
                secs[t] = now() - a;
            });
        for (auto &x : th) x.join();
        const double wall = now() - t0;
        double mx = 0;
        for (double x : secs) mx = x > mx ? x : mx;
        printf("%s: %d threads x %d chains of %d kernels: %.1f us per chain (all threads together), %.2f us per command; wall %.2f s\n",
               mode ? "graph" : "eager", T, R, K, mx / R / T * 1e6, mx / R / T / K * 1e6, wall);
    }
    return 0;
}
