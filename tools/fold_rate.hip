// Micro-benchmark behind DESIGN.md's k_radix_fold numbers: is the eq-weighted fold of 43 x 2^20 u32 bound by HBM reads
// or by the integer multiplier?  Times variants of the fold body on the real shape (per-kernel timestamps through
// hipExtLaunchKernelGGL's start/stop events, a 1 GiB memset between launches so nothing is served from the MALL),
// and the issue rate of the 32-bit multiply instructions.
//   hipcc --offload-arch=gfx950 -O3 -I zigz_amd/csrc tools/fold_rate.hip -o /tmp/fold_rate && /tmp/fold_rate
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>
#include "field.hpp"

using namespace zk;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// VARIANT 0: mont_mul per term (the shipped body)   1: plain sum, no multiply (read ceiling)
//         2: 96-bit accumulation of w*v, one reduction per output     3: split lo/hi u64 accumulators
template <int VARIANT, int RB_, int RLOOPS_>
__global__ __launch_bounds__(256) void k_fold(const uint32_t *__restrict__ in, size_t in_stride, size_t m, size_t nb,
                                              const uint32_t *__restrict__ w_m, size_t w_stride,
                                              unsigned long long *__restrict__ part, size_t part_col_stride) {
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q * 4 >= m) return;
    const size_t col = blockIdx.z;
    const uint4 *p = reinterpret_cast<const uint4 *>(in + col * in_stride) + q;
    const uint32_t *w = w_m + col * w_stride;
    const size_t mq = m / 4;
    unsigned long long s[4] = {0, 0, 0, 0};
    uint32_t h[4] = {0, 0, 0, 0};
    unsigned long long t[4] = {0, 0, 0, 0};
#pragma unroll 1
    for (int l = 0; l < RLOOPS_; l++) {
        const size_t b0 = ((size_t)blockIdx.y * RLOOPS_ + l) * RB_;
        uint4 v[RB_];
#pragma unroll
        for (int j = 0; j < RB_; j++) v[j] = p[(b0 + j) * mq];
#pragma unroll
        for (int j = 0; j < RB_; j++) {
            const uint32_t wj = w[b0 + j];
            const uint32_t e[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
            for (int c = 0; c < 4; c++) {
                if (VARIANT == 0) s[c] += mont_mul(wj, e[c]);
                if (VARIANT == 1) s[c] += e[c];
                if (VARIANT == 2) {
                    const unsigned long long pr = (unsigned long long)wj * e[c];
                    s[c] += pr;
                    h[c] += (s[c] < pr);
                }
                if (VARIANT == 3) {
                    const unsigned long long pr = (unsigned long long)wj * e[c];
                    s[c] += (uint32_t)pr;
                    t[c] += pr >> 32;
                }
            }
        }
    }
    if (VARIANT == 2) {
        // X = h*2^64 + s = 2^32 * sum(w*v)  (w in Montgomery form)  =>  sum(w*v) = h*2^32 + hi(s) + lo(s)*2^-32 (mod p)
#pragma unroll
        for (int c = 0; c < 4; c++)
            s[c] = (unsigned long long)h[c] * R_MOD_P + (s[c] >> 32) + monty_reduce((unsigned long long)(uint32_t)s[c]);
    }
    if (VARIANT == 3) {
#pragma unroll
        for (int c = 0; c < 4; c++) s[c] += t[c];
    }
    ulonglong2 *o = reinterpret_cast<ulonglong2 *>(part + col * part_col_stride + (size_t)blockIdx.y * m + q * 4);
    o[0] = make_ulonglong2(s[0], s[1]);
    o[1] = make_ulonglong2(s[2], s[3]);
}

// flush by READING a large clean buffer: a memset would leave dirty lines whose write-back competes with the fold's reads
__global__ __launch_bounds__(256) void k_flush_read(const uint4 *__restrict__ p, size_t n16, uint32_t *out) {
    uint32_t s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const uint4 v = p[i];
        s += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (s == 0x12345678u) out[0] = s;
}

template <int OP>
__global__ __launch_bounds__(256) void k_rate(uint32_t *out, int iters) {
    uint32_t a[8];
    unsigned long long acc[8];
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 2654435761u + i * 40503u + blockIdx.x; acc[i] = i; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                uint32_t x = a[i], y = a[(i + 1) & 7];
                if (OP == 0) a[i] = x * y;                                              // v_mul_lo_u32
                if (OP == 1) a[i] = __umulhi(x, y);                                     // v_mul_hi_u32
                if (OP == 2) { acc[i] = (unsigned long long)x * (uint32_t)acc[(i + 1) & 7] + acc[i]; }  // v_mad_u64_u32
                if (OP == 3) a[i] = __umul24(x, y);                                     // v_mul_u32_u24
                if (OP == 4) a[i] = __umul24(x, y) + a[(i + 3) & 7];                    // v_mad_u32_u24
                if (OP == 5) a[i] = x + y;                                              // v_add_u32
            }
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s ^= a[i] ^ (uint32_t)acc[i] ^ (uint32_t)(acc[i] >> 32);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
static int rate(const char *name) {
    const int blocks = 256 * 8, iters = 2000;
    uint32_t *d;
    CK(hipMalloc(&d, blocks * 256 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k_rate<OP><<<blocks, 256>>>(d, 10);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k_rate<OP><<<blocks, 256>>>(d, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    double ops = (double)blocks * 256 * iters * 32;
    printf("%-28s %8.2f Tops/s  (%.1f%% of 78.6 T lane-ops/s)\n", name, ops / ms / 1e9, ops / ms / 1e9 / 78.64 * 100);
    CK(hipFree(d));
    return 0;
}

static uint32_t *d_in, *d_w;
static unsigned long long *d_part;
static void *d_flush;
static int g_flush = 1;  // 0 none (MALL-warm), 1 read 1 GiB, 2 memset 1 GiB
static const size_t NCOLS = 43, NV = 20, N = (size_t)1 << NV, M = 1024, NB = N / M, FLUSH = (size_t)1 << 30;

template <int VARIANT, int RB_, int RLOOPS_>
static int fold(const char *name) {
    const size_t groups = NB / (RB_ * RLOOPS_);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> t;
    for (int it = 0; it < 12; it++) {
        if (g_flush == 2) CK(hipMemsetAsync(d_flush, it, FLUSH, 0));
        if (g_flush == 1) k_flush_read<<<2048, 256>>>((const uint4 *)d_flush, FLUSH / 16, (uint32_t *)d_part);
        dim3 grid((unsigned)(M / 4 / 256 ? M / 4 / 256 : 1), (unsigned)groups, (unsigned)NCOLS);
        hipExtLaunchKernelGGL((k_fold<VARIANT, RB_, RLOOPS_>), grid, dim3(256), 0, 0, e0, e1, 0, (const uint32_t *)d_in, N, M, NB,
                              (const uint32_t *)d_w, NB, d_part, groups * M);
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 2) t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    const double bytes = (double)NCOLS * (N * 4 + groups * M * 8);
    printf("%-44s grid %4zu wg  min %6.1f us  median %6.1f us  -> %5.2f TB/s (median)\n", name, groups * NCOLS, t[0] * 1e3,
           t[t.size() / 2] * 1e3, bytes / (t[t.size() / 2] * 1e-3) / 1e12);
    return 0;
}

int main() {
    rate<0>("v_mul_lo_u32");
    rate<1>("v_mul_hi_u32");
    rate<2>("v_mad_u64_u32");
    rate<3>("v_mul_u32_u24");
    rate<4>("v_mad_u32_u24");
    rate<5>("v_add_u32");
    CK(hipMalloc(&d_in, NCOLS * N * 4));
    CK(hipMalloc(&d_w, NCOLS * NB * 4));
    CK(hipMalloc(&d_part, NCOLS * 64 * M * 8));
    CK(hipMalloc(&d_flush, FLUSH));
    CK(hipMemset(d_in, 0x5a, NCOLS * N * 4));
    CK(hipMemset(d_w, 0x11, NCOLS * NB * 4));
    for (g_flush = 0; g_flush < 3; g_flush++) {
        printf("-- flush mode %d (0 none: table may sit in the 256 MB MALL, 1 read 1 GiB, 2 memset 1 GiB) --\n", g_flush);
        fold<0, 16, 4>("mont_mul per term, 16 loads x 4 (shipped)");
        fold<1, 16, 4>("plain sum (read ceiling), 16 x 4");
        fold<1, 16, 1>("plain sum, 16 x 1");
        fold<1, 8, 2>("plain sum, 8 x 2");
        fold<2, 16, 4>("96-bit accumulate, 16 x 4");
        fold<2, 16, 1>("96-bit accumulate, 16 x 1");
        fold<3, 16, 4>("split lo/hi accumulate, 16 x 4");
        fold<0, 16, 1>("mont_mul per term, 16 x 1");
    }
    return 0;
}
