// Micro-benchmark: issue rate of the integer VALU instructions the Keccak kernels are made of, on gfx950.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters) {
    uint32_t a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 2654435761u + i * 40503u + blockIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                uint32_t x = a[i], y = a[(i + 1) & 7], z = a[(i + 3) & 7];
                if (OP == 0) a[i] = x ^ y;                                            // v_xor_b32
                if (OP == 1) a[i] = __builtin_amdgcn_bitop3_b32(x, y, z, 0x96);        // v_bitop3_b32
                if (OP == 2) a[i] = __builtin_amdgcn_alignbit(x, y, 13);               // v_alignbit_b32 (imm shift)
                if (OP == 3) a[i] = __builtin_amdgcn_bitop3_b32(x, y, z, 0xD2);
                if (OP == 4) a[i] = (x & y) | (~x & z);                                // v_bfi_b32
                if (OP == 5) a[i] = x + y;                                             // v_add_u32
                if (OP == 6) a[i] = __builtin_amdgcn_alignbit(x, y, z);                // v_alignbit_b32 (reg shift)
                // inline asm so the compiler cannot fold chains of rotations
                if (OP == 7) asm volatile("v_alignbit_b32 %0, %1, %1, 13" : "=v"(a[i]) : "v"(y));          // rot32: one source register
                if (OP == 8) asm volatile("v_alignbit_b32 %0, %1, %2, 13" : "=v"(a[i]) : "v"(x), "v"(y));  // funnel: two source registers
                if (OP == 9) asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(a[i]) : "v"(x), "v"(y), "v"(z));
                if (OP == 10) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(a[i]) : "v"(x), "v"(y));
            }
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
static void run(const char *name) {
    const int blocks = 256 * 8, iters = 4000;
    uint32_t *d;
    hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)blocks * 256 * iters * 32;
    printf("%-28s %8.2f Tops/s  (%.1f%% of 78.6 T = 256CU*128 lanes*2.4GHz)\n", name, ops / ms / 1e9, ops / ms / 1e9 / 78.64 * 100);
    hipFree(d);
}

int main() {
    run<0>("v_xor_b32");
    run<1>("v_bitop3_b32 (xor3)");
    run<3>("v_bitop3_b32 (chi)");
    run<2>("v_alignbit_b32 imm");
    run<6>("v_alignbit_b32 reg");
    run<4>("v_bfi_b32");
    run<5>("v_add_u32");
    run<7>("asm v_alignbit rot32 (1 src)");
    run<8>("asm v_alignbit funnel (2 src)");
    run<9>("asm v_bitop3 (3 src)");
    run<10>("asm v_xor (2 src)");
    return 0;
}
