// Probe kernel: Keccak-f[800]-shaped rounds (see gen.py) at 8 waves per SIMD, compiler-scheduled vs run-scheduled.
//   python3 tools/k800/gen.py > /tmp/k800_plain.inc; python3 tools/k800/gen.py --scheduled > /tmp/k800_sched.inc
//   hipcc --offload-arch=gfx950 -O3 -I /tmp tools/k800/k800.hip -o /tmp/k800
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ZK_X3(a, b, c) __builtin_amdgcn_bitop3_b32((a), (b), (c), 0x96)
#define ZK_CHI(a, b, c) __builtin_amdgcn_bitop3_b32((a), (b), (c), 0xD2)
#define ZK_ROT32(x, k) __builtin_amdgcn_alignbit((x), (x), 32 - (k))
#define ZK_SCHED() __builtin_amdgcn_sched_barrier(0)

template <int V>
__global__ __launch_bounds__(256) void k(uint32_t *out, int perms) {
    uint32_t a[25];
    for (int i = 0; i < 25; i++) a[i] = threadIdx.x * 2654435761u + i * 40503u + blockIdx.x;
    for (int p = 0; p < perms; p++) {
#pragma unroll 22
        for (int r = 0; r < 22; r++) {
            uint32_t n[25];
            if (V == 0) {
#include "k800_plain.inc"
            } else {
#include "k800_sched.inc"
            }
#pragma unroll
            for (int i = 0; i < 25; i++) a[i] = n[i];
            a[0] ^= 0x80008081u + r;
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 25; i++) s ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int V>
static void run(const char *name) {
    const int blocks = 256 * 16, perms = 256;
    uint32_t *d;
    (void)hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    k<V><<<blocks, 256>>>(d, 2);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<V><<<blocks, 256>>>(d, perms);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // 60 v_bitop3 + 29 v_alignbit + 1 xor per round
    printf("%-28s %.2f G rounds/s = %.2f T lane-ops/s (90 instructions per round)\n", name, (double)blocks * 256 * perms * 22 / ms / 1e6,
           (double)blocks * 256 * perms * 22 * 90 / ms / 1e9);
    (void)hipFree(d);
}

int main() {
    run<0>("compiler order");
    run<1>("runs of 12 : 5");
    run<0>("compiler order");
    run<1>("runs of 12 : 5");
    return 0;
}
