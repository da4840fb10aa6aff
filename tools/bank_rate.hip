// Does the VGPR bank (register number mod 4) of the source operands change the issue rate of v_bitop3_b32 /
// v_alignbit_b32 / v_xor_b32 on gfx950?  Fixed registers through inline asm; destinations v40..v47 are never read, so
// there are no dependency stalls.  Result on MI355X: profiles/r01_bank_rate.txt.
//   hipcc --offload-arch=gfx950 -O3 tools/bank_rate.hip -o /tmp/bank_rate && /tmp/bank_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CLOB "v20", "v21", "v22", "v24", "v28", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47"

template <int V>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters) {
    uint32_t acc = threadIdx.x;
    for (int it = 0; it < iters; it++) {
        if (V == 0)
            asm volatile("v_bitop3_b32 v40, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v41, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v42, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v43, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v44, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v45, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v46, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v47, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v40, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v41, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v42, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v43, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v44, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v45, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v46, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v47, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v40, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v41, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v42, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v43, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v44, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v45, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v46, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v47, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v40, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v41, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v42, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v43, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v44, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v45, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v46, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v47, v20, v21, v22 bitop3:0x96\n"
                         : "+v"(acc) : : CLOB);
        if (V == 1)
            asm volatile("v_bitop3_b32 v40, v20, v24, v28 bitop3:0x96\n"
                         "v_bitop3_b32 v41, v20, v24, v28 bitop3:0x96\n"
                         "v_bitop3_b32 v42, v20, v24, v28 bitop3:0x96\n"
                         "v_bitop3_b32 v43, v20, v24, v28 bitop3:0x96\n"
                         "v_bitop3_b32 v44, v20, v24, v28 bitop3:0x96\n"
                         "v_bitop3_b32 v45, v20, v24, v28 bitop3:0x96\n"
                         "v_bitop3_b32 v46, v20, v24, v28 bitop3:0x96\n"
                         "v_bitop3_b32 v47, v20, v24, v28 bitop3:0x96\n"
                         "v_bitop3_b32 v40, v20, v24, v28 bitop3:0x96\n"
                         "v_bitop3_b32 v41, v20, v24, v28 bitop3:0x96\n"
                         "v_bitop3_b32 v42, v20, v24, v28 bitop3:0x96\n"
                         "v_bitop3_b32 v43, v20, v24, v28 bitop3:0x96\n"
                         "v_bitop3_b32 v44, v20, v24, v28 bitop3:0x96\n"
                         "v_bitop3_b32 v45, v20, v24, v28 bitop3:0x96\n"
                         "v_bitop3_b32 v46, v20, v24, v28 bitop3:0x96\n"
                         "v_bitop3_b32 v47, v20, v24, v28 bitop3:0x96\n"
                         "v_bitop3_b32 v40, v20, v24, v28 bitop3:0x96\n"
                         "v_bitop3_b32 v41, v20, v24, v28 bitop3:0x96\n"
                         "v_bitop3_b32 v42, v20, v24, v28 bitop3:0x96\n"
                         "v_bitop3_b32 v43, v20, v24, v28 bitop3:0x96\n"
                         "v_bitop3_b32 v44, v20, v24, v28 bitop3:0x96\n"
                         "v_bitop3_b32 v45, v20, v24, v28 bitop3:0x96\n"
                         "v_bitop3_b32 v46, v20, v24, v28 bitop3:0x96\n"
                         "v_bitop3_b32 v47, v20, v24, v28 bitop3:0x96\n"
                         "v_bitop3_b32 v40, v20, v24, v28 bitop3:0x96\n"
                         "v_bitop3_b32 v41, v20, v24, v28 bitop3:0x96\n"
                         "v_bitop3_b32 v42, v20, v24, v28 bitop3:0x96\n"
                         "v_bitop3_b32 v43, v20, v24, v28 bitop3:0x96\n"
                         "v_bitop3_b32 v44, v20, v24, v28 bitop3:0x96\n"
                         "v_bitop3_b32 v45, v20, v24, v28 bitop3:0x96\n"
                         "v_bitop3_b32 v46, v20, v24, v28 bitop3:0x96\n"
                         "v_bitop3_b32 v47, v20, v24, v28 bitop3:0x96\n"
                         : "+v"(acc) : : CLOB);
        if (V == 2)
            asm volatile("v_bitop3_b32 v40, v20, v24, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v41, v20, v24, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v42, v20, v24, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v43, v20, v24, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v44, v20, v24, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v45, v20, v24, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v46, v20, v24, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v47, v20, v24, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v40, v20, v24, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v41, v20, v24, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v42, v20, v24, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v43, v20, v24, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v44, v20, v24, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v45, v20, v24, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v46, v20, v24, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v47, v20, v24, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v40, v20, v24, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v41, v20, v24, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v42, v20, v24, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v43, v20, v24, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v44, v20, v24, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v45, v20, v24, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v46, v20, v24, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v47, v20, v24, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v40, v20, v24, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v41, v20, v24, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v42, v20, v24, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v43, v20, v24, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v44, v20, v24, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v45, v20, v24, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v46, v20, v24, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v47, v20, v24, v21 bitop3:0x96\n"
                         : "+v"(acc) : : CLOB);
        if (V == 3)
            asm volatile("v_bitop3_b32 v40, v20, v21, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v41, v20, v21, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v42, v20, v21, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v43, v20, v21, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v44, v20, v21, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v45, v20, v21, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v46, v20, v21, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v47, v20, v21, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v40, v20, v21, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v41, v20, v21, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v42, v20, v21, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v43, v20, v21, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v44, v20, v21, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v45, v20, v21, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v46, v20, v21, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v47, v20, v21, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v40, v20, v21, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v41, v20, v21, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v42, v20, v21, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v43, v20, v21, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v44, v20, v21, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v45, v20, v21, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v46, v20, v21, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v47, v20, v21, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v40, v20, v21, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v41, v20, v21, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v42, v20, v21, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v43, v20, v21, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v44, v20, v21, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v45, v20, v21, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v46, v20, v21, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v47, v20, v21, v24 bitop3:0x96\n"
                         : "+v"(acc) : : CLOB);
        if (V == 4)
            asm volatile("v_bitop3_b32 v40, v21, v20, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v41, v21, v20, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v42, v21, v20, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v43, v21, v20, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v44, v21, v20, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v45, v21, v20, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v46, v21, v20, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v47, v21, v20, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v40, v21, v20, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v41, v21, v20, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v42, v21, v20, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v43, v21, v20, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v44, v21, v20, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v45, v21, v20, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v46, v21, v20, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v47, v21, v20, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v40, v21, v20, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v41, v21, v20, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v42, v21, v20, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v43, v21, v20, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v44, v21, v20, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v45, v21, v20, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v46, v21, v20, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v47, v21, v20, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v40, v21, v20, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v41, v21, v20, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v42, v21, v20, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v43, v21, v20, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v44, v21, v20, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v45, v21, v20, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v46, v21, v20, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v47, v21, v20, v24 bitop3:0x96\n"
                         : "+v"(acc) : : CLOB);
        if (V == 5)
            asm volatile("v_bitop3_b32 v40, v20, v20, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v41, v20, v20, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v42, v20, v20, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v43, v20, v20, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v44, v20, v20, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v45, v20, v20, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v46, v20, v20, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v47, v20, v20, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v40, v20, v20, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v41, v20, v20, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v42, v20, v20, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v43, v20, v20, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v44, v20, v20, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v45, v20, v20, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v46, v20, v20, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v47, v20, v20, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v40, v20, v20, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v41, v20, v20, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v42, v20, v20, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v43, v20, v20, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v44, v20, v20, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v45, v20, v20, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v46, v20, v20, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v47, v20, v20, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v40, v20, v20, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v41, v20, v20, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v42, v20, v20, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v43, v20, v20, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v44, v20, v20, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v45, v20, v20, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v46, v20, v20, v21 bitop3:0x96\n"
                         "v_bitop3_b32 v47, v20, v20, v21 bitop3:0x96\n"
                         : "+v"(acc) : : CLOB);
        if (V == 6)
            asm volatile("v_bitop3_b32 v40, v20, v20, v20 bitop3:0x96\n"
                         "v_bitop3_b32 v41, v20, v20, v20 bitop3:0x96\n"
                         "v_bitop3_b32 v42, v20, v20, v20 bitop3:0x96\n"
                         "v_bitop3_b32 v43, v20, v20, v20 bitop3:0x96\n"
                         "v_bitop3_b32 v44, v20, v20, v20 bitop3:0x96\n"
                         "v_bitop3_b32 v45, v20, v20, v20 bitop3:0x96\n"
                         "v_bitop3_b32 v46, v20, v20, v20 bitop3:0x96\n"
                         "v_bitop3_b32 v47, v20, v20, v20 bitop3:0x96\n"
                         "v_bitop3_b32 v40, v20, v20, v20 bitop3:0x96\n"
                         "v_bitop3_b32 v41, v20, v20, v20 bitop3:0x96\n"
                         "v_bitop3_b32 v42, v20, v20, v20 bitop3:0x96\n"
                         "v_bitop3_b32 v43, v20, v20, v20 bitop3:0x96\n"
                         "v_bitop3_b32 v44, v20, v20, v20 bitop3:0x96\n"
                         "v_bitop3_b32 v45, v20, v20, v20 bitop3:0x96\n"
                         "v_bitop3_b32 v46, v20, v20, v20 bitop3:0x96\n"
                         "v_bitop3_b32 v47, v20, v20, v20 bitop3:0x96\n"
                         "v_bitop3_b32 v40, v20, v20, v20 bitop3:0x96\n"
                         "v_bitop3_b32 v41, v20, v20, v20 bitop3:0x96\n"
                         "v_bitop3_b32 v42, v20, v20, v20 bitop3:0x96\n"
                         "v_bitop3_b32 v43, v20, v20, v20 bitop3:0x96\n"
                         "v_bitop3_b32 v44, v20, v20, v20 bitop3:0x96\n"
                         "v_bitop3_b32 v45, v20, v20, v20 bitop3:0x96\n"
                         "v_bitop3_b32 v46, v20, v20, v20 bitop3:0x96\n"
                         "v_bitop3_b32 v47, v20, v20, v20 bitop3:0x96\n"
                         "v_bitop3_b32 v40, v20, v20, v20 bitop3:0x96\n"
                         "v_bitop3_b32 v41, v20, v20, v20 bitop3:0x96\n"
                         "v_bitop3_b32 v42, v20, v20, v20 bitop3:0x96\n"
                         "v_bitop3_b32 v43, v20, v20, v20 bitop3:0x96\n"
                         "v_bitop3_b32 v44, v20, v20, v20 bitop3:0x96\n"
                         "v_bitop3_b32 v45, v20, v20, v20 bitop3:0x96\n"
                         "v_bitop3_b32 v46, v20, v20, v20 bitop3:0x96\n"
                         "v_bitop3_b32 v47, v20, v20, v20 bitop3:0x96\n"
                         : "+v"(acc) : : CLOB);
        if (V == 7)
            asm volatile("v_bitop3_b32 v40, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v41, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v42, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v43, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v44, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v45, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v46, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v47, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v40, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v41, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v42, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v43, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v44, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v45, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v46, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v47, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v40, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v41, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v42, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v43, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v44, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v45, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v46, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v47, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v40, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v41, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v42, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v43, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v44, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v45, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v46, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v47, v20, v21, v22 bitop3:0x96\n"
                         : "+v"(acc) : : CLOB);
        if (V == 8)
            asm volatile("v_alignbit_b32 v40, v20, v21, 13\n"
                         "v_alignbit_b32 v41, v20, v21, 13\n"
                         "v_alignbit_b32 v42, v20, v21, 13\n"
                         "v_alignbit_b32 v43, v20, v21, 13\n"
                         "v_alignbit_b32 v44, v20, v21, 13\n"
                         "v_alignbit_b32 v45, v20, v21, 13\n"
                         "v_alignbit_b32 v46, v20, v21, 13\n"
                         "v_alignbit_b32 v47, v20, v21, 13\n"
                         "v_alignbit_b32 v40, v20, v21, 13\n"
                         "v_alignbit_b32 v41, v20, v21, 13\n"
                         "v_alignbit_b32 v42, v20, v21, 13\n"
                         "v_alignbit_b32 v43, v20, v21, 13\n"
                         "v_alignbit_b32 v44, v20, v21, 13\n"
                         "v_alignbit_b32 v45, v20, v21, 13\n"
                         "v_alignbit_b32 v46, v20, v21, 13\n"
                         "v_alignbit_b32 v47, v20, v21, 13\n"
                         "v_alignbit_b32 v40, v20, v21, 13\n"
                         "v_alignbit_b32 v41, v20, v21, 13\n"
                         "v_alignbit_b32 v42, v20, v21, 13\n"
                         "v_alignbit_b32 v43, v20, v21, 13\n"
                         "v_alignbit_b32 v44, v20, v21, 13\n"
                         "v_alignbit_b32 v45, v20, v21, 13\n"
                         "v_alignbit_b32 v46, v20, v21, 13\n"
                         "v_alignbit_b32 v47, v20, v21, 13\n"
                         "v_alignbit_b32 v40, v20, v21, 13\n"
                         "v_alignbit_b32 v41, v20, v21, 13\n"
                         "v_alignbit_b32 v42, v20, v21, 13\n"
                         "v_alignbit_b32 v43, v20, v21, 13\n"
                         "v_alignbit_b32 v44, v20, v21, 13\n"
                         "v_alignbit_b32 v45, v20, v21, 13\n"
                         "v_alignbit_b32 v46, v20, v21, 13\n"
                         "v_alignbit_b32 v47, v20, v21, 13\n"
                         : "+v"(acc) : : CLOB);
        if (V == 9)
            asm volatile("v_alignbit_b32 v40, v20, v24, 13\n"
                         "v_alignbit_b32 v41, v20, v24, 13\n"
                         "v_alignbit_b32 v42, v20, v24, 13\n"
                         "v_alignbit_b32 v43, v20, v24, 13\n"
                         "v_alignbit_b32 v44, v20, v24, 13\n"
                         "v_alignbit_b32 v45, v20, v24, 13\n"
                         "v_alignbit_b32 v46, v20, v24, 13\n"
                         "v_alignbit_b32 v47, v20, v24, 13\n"
                         "v_alignbit_b32 v40, v20, v24, 13\n"
                         "v_alignbit_b32 v41, v20, v24, 13\n"
                         "v_alignbit_b32 v42, v20, v24, 13\n"
                         "v_alignbit_b32 v43, v20, v24, 13\n"
                         "v_alignbit_b32 v44, v20, v24, 13\n"
                         "v_alignbit_b32 v45, v20, v24, 13\n"
                         "v_alignbit_b32 v46, v20, v24, 13\n"
                         "v_alignbit_b32 v47, v20, v24, 13\n"
                         "v_alignbit_b32 v40, v20, v24, 13\n"
                         "v_alignbit_b32 v41, v20, v24, 13\n"
                         "v_alignbit_b32 v42, v20, v24, 13\n"
                         "v_alignbit_b32 v43, v20, v24, 13\n"
                         "v_alignbit_b32 v44, v20, v24, 13\n"
                         "v_alignbit_b32 v45, v20, v24, 13\n"
                         "v_alignbit_b32 v46, v20, v24, 13\n"
                         "v_alignbit_b32 v47, v20, v24, 13\n"
                         "v_alignbit_b32 v40, v20, v24, 13\n"
                         "v_alignbit_b32 v41, v20, v24, 13\n"
                         "v_alignbit_b32 v42, v20, v24, 13\n"
                         "v_alignbit_b32 v43, v20, v24, 13\n"
                         "v_alignbit_b32 v44, v20, v24, 13\n"
                         "v_alignbit_b32 v45, v20, v24, 13\n"
                         "v_alignbit_b32 v46, v20, v24, 13\n"
                         "v_alignbit_b32 v47, v20, v24, 13\n"
                         : "+v"(acc) : : CLOB);
        if (V == 10)
            asm volatile("v_alignbit_b32 v40, v20, v20, 13\n"
                         "v_alignbit_b32 v41, v20, v20, 13\n"
                         "v_alignbit_b32 v42, v20, v20, 13\n"
                         "v_alignbit_b32 v43, v20, v20, 13\n"
                         "v_alignbit_b32 v44, v20, v20, 13\n"
                         "v_alignbit_b32 v45, v20, v20, 13\n"
                         "v_alignbit_b32 v46, v20, v20, 13\n"
                         "v_alignbit_b32 v47, v20, v20, 13\n"
                         "v_alignbit_b32 v40, v20, v20, 13\n"
                         "v_alignbit_b32 v41, v20, v20, 13\n"
                         "v_alignbit_b32 v42, v20, v20, 13\n"
                         "v_alignbit_b32 v43, v20, v20, 13\n"
                         "v_alignbit_b32 v44, v20, v20, 13\n"
                         "v_alignbit_b32 v45, v20, v20, 13\n"
                         "v_alignbit_b32 v46, v20, v20, 13\n"
                         "v_alignbit_b32 v47, v20, v20, 13\n"
                         "v_alignbit_b32 v40, v20, v20, 13\n"
                         "v_alignbit_b32 v41, v20, v20, 13\n"
                         "v_alignbit_b32 v42, v20, v20, 13\n"
                         "v_alignbit_b32 v43, v20, v20, 13\n"
                         "v_alignbit_b32 v44, v20, v20, 13\n"
                         "v_alignbit_b32 v45, v20, v20, 13\n"
                         "v_alignbit_b32 v46, v20, v20, 13\n"
                         "v_alignbit_b32 v47, v20, v20, 13\n"
                         "v_alignbit_b32 v40, v20, v20, 13\n"
                         "v_alignbit_b32 v41, v20, v20, 13\n"
                         "v_alignbit_b32 v42, v20, v20, 13\n"
                         "v_alignbit_b32 v43, v20, v20, 13\n"
                         "v_alignbit_b32 v44, v20, v20, 13\n"
                         "v_alignbit_b32 v45, v20, v20, 13\n"
                         "v_alignbit_b32 v46, v20, v20, 13\n"
                         "v_alignbit_b32 v47, v20, v20, 13\n"
                         : "+v"(acc) : : CLOB);
        if (V == 11)
            asm volatile("v_xor_b32 v40, v20, v21\n"
                         "v_xor_b32 v41, v20, v21\n"
                         "v_xor_b32 v42, v20, v21\n"
                         "v_xor_b32 v43, v20, v21\n"
                         "v_xor_b32 v44, v20, v21\n"
                         "v_xor_b32 v45, v20, v21\n"
                         "v_xor_b32 v46, v20, v21\n"
                         "v_xor_b32 v47, v20, v21\n"
                         "v_xor_b32 v40, v20, v21\n"
                         "v_xor_b32 v41, v20, v21\n"
                         "v_xor_b32 v42, v20, v21\n"
                         "v_xor_b32 v43, v20, v21\n"
                         "v_xor_b32 v44, v20, v21\n"
                         "v_xor_b32 v45, v20, v21\n"
                         "v_xor_b32 v46, v20, v21\n"
                         "v_xor_b32 v47, v20, v21\n"
                         "v_xor_b32 v40, v20, v21\n"
                         "v_xor_b32 v41, v20, v21\n"
                         "v_xor_b32 v42, v20, v21\n"
                         "v_xor_b32 v43, v20, v21\n"
                         "v_xor_b32 v44, v20, v21\n"
                         "v_xor_b32 v45, v20, v21\n"
                         "v_xor_b32 v46, v20, v21\n"
                         "v_xor_b32 v47, v20, v21\n"
                         "v_xor_b32 v40, v20, v21\n"
                         "v_xor_b32 v41, v20, v21\n"
                         "v_xor_b32 v42, v20, v21\n"
                         "v_xor_b32 v43, v20, v21\n"
                         "v_xor_b32 v44, v20, v21\n"
                         "v_xor_b32 v45, v20, v21\n"
                         "v_xor_b32 v46, v20, v21\n"
                         "v_xor_b32 v47, v20, v21\n"
                         : "+v"(acc) : : CLOB);
        if (V == 12)
            asm volatile("v_xor_b32 v40, v20, v24\n"
                         "v_xor_b32 v41, v20, v24\n"
                         "v_xor_b32 v42, v20, v24\n"
                         "v_xor_b32 v43, v20, v24\n"
                         "v_xor_b32 v44, v20, v24\n"
                         "v_xor_b32 v45, v20, v24\n"
                         "v_xor_b32 v46, v20, v24\n"
                         "v_xor_b32 v47, v20, v24\n"
                         "v_xor_b32 v40, v20, v24\n"
                         "v_xor_b32 v41, v20, v24\n"
                         "v_xor_b32 v42, v20, v24\n"
                         "v_xor_b32 v43, v20, v24\n"
                         "v_xor_b32 v44, v20, v24\n"
                         "v_xor_b32 v45, v20, v24\n"
                         "v_xor_b32 v46, v20, v24\n"
                         "v_xor_b32 v47, v20, v24\n"
                         "v_xor_b32 v40, v20, v24\n"
                         "v_xor_b32 v41, v20, v24\n"
                         "v_xor_b32 v42, v20, v24\n"
                         "v_xor_b32 v43, v20, v24\n"
                         "v_xor_b32 v44, v20, v24\n"
                         "v_xor_b32 v45, v20, v24\n"
                         "v_xor_b32 v46, v20, v24\n"
                         "v_xor_b32 v47, v20, v24\n"
                         "v_xor_b32 v40, v20, v24\n"
                         "v_xor_b32 v41, v20, v24\n"
                         "v_xor_b32 v42, v20, v24\n"
                         "v_xor_b32 v43, v20, v24\n"
                         "v_xor_b32 v44, v20, v24\n"
                         "v_xor_b32 v45, v20, v24\n"
                         "v_xor_b32 v46, v20, v24\n"
                         "v_xor_b32 v47, v20, v24\n"
                         : "+v"(acc) : : CLOB);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_vary(uint32_t *out, int iters) {
    uint32_t acc = threadIdx.x;
    for (int it = 0; it < iters; it++)
        asm volatile("v_bitop3_b32 v40, v20, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v23, v24, v25 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v26, v27, v28 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v29, v30, v31 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v32, v33, v34 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v35, v36, v37 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v22, v23, v24 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v25, v26, v27 bitop3:0x96\n"
                     "v_bitop3_b32 v40, v28, v29, v30 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v31, v32, v33 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v34, v35, v36 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v21, v22, v23 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v24, v25, v26 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v27, v28, v29 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v30, v31, v32 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v33, v34, v35 bitop3:0x96\n"
                     "v_bitop3_b32 v40, v20, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v23, v24, v25 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v26, v27, v28 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v29, v30, v31 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v32, v33, v34 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v35, v36, v37 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v22, v23, v24 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v25, v26, v27 bitop3:0x96\n"
                     "v_bitop3_b32 v40, v28, v29, v30 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v31, v32, v33 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v34, v35, v36 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v21, v22, v23 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v24, v25, v26 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v27, v28, v29 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v30, v31, v32 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v33, v34, v35 bitop3:0x96\n"
                     : "+v"(acc) : : CLOB, "v23", "v25", "v26", "v27", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37");
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_dep(uint32_t *out, int iters) {
    uint32_t acc = threadIdx.x;
    for (int it = 0; it < iters; it++)
        asm volatile("v_bitop3_b32 v40, v48, v47, v45 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v49, v48, v46 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v50, v49, v47 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v51, v50, v48 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v52, v51, v49 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v53, v52, v50 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v54, v53, v51 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v55, v54, v52 bitop3:0x96\n"
                     "v_bitop3_b32 v48, v40, v55, v53 bitop3:0x96\n"
                     "v_bitop3_b32 v49, v41, v40, v54 bitop3:0x96\n"
                     "v_bitop3_b32 v50, v42, v41, v55 bitop3:0x96\n"
                     "v_bitop3_b32 v51, v43, v42, v40 bitop3:0x96\n"
                     "v_bitop3_b32 v52, v44, v43, v41 bitop3:0x96\n"
                     "v_bitop3_b32 v53, v45, v44, v42 bitop3:0x96\n"
                     "v_bitop3_b32 v54, v46, v45, v43 bitop3:0x96\n"
                     "v_bitop3_b32 v55, v47, v46, v44 bitop3:0x96\n"
                     "v_bitop3_b32 v40, v48, v47, v45 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v49, v48, v46 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v50, v49, v47 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v51, v50, v48 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v52, v51, v49 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v53, v52, v50 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v54, v53, v51 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v55, v54, v52 bitop3:0x96\n"
                     "v_bitop3_b32 v48, v40, v55, v53 bitop3:0x96\n"
                     "v_bitop3_b32 v49, v41, v40, v54 bitop3:0x96\n"
                     "v_bitop3_b32 v50, v42, v41, v55 bitop3:0x96\n"
                     "v_bitop3_b32 v51, v43, v42, v40 bitop3:0x96\n"
                     "v_bitop3_b32 v52, v44, v43, v41 bitop3:0x96\n"
                     "v_bitop3_b32 v53, v45, v44, v42 bitop3:0x96\n"
                     "v_bitop3_b32 v54, v46, v45, v43 bitop3:0x96\n"
                     "v_bitop3_b32 v55, v47, v46, v44 bitop3:0x96\n"
                     : "+v"(acc) : : CLOB, "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// Even workgroups run a pure v_bitop3 stream, odd ones a pure v_alignbit stream: every SIMD hosts waves of both
// kinds, each wave's own stream is a single instruction class.  If the SIMD simply time-shares, the run takes
// N_b / R_b + N_a / R_a.
__global__ __launch_bounds__(256) void k_two_streams(uint32_t *out, int iters_b, int iters_a) {
    uint32_t acc = threadIdx.x;
    if ((blockIdx.x & 1) == 0) {
        for (int it = 0; it < iters_b; it++)
            asm volatile("v_bitop3_b32 v40, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v41, v23, v24, v25 bitop3:0x96\n"
                         "v_bitop3_b32 v42, v26, v27, v28 bitop3:0x96\n"
                         "v_bitop3_b32 v43, v29, v30, v31 bitop3:0x96\n"
                         "v_bitop3_b32 v44, v32, v33, v34 bitop3:0x96\n"
                         "v_bitop3_b32 v45, v35, v36, v37 bitop3:0x96\n"
                         "v_bitop3_b32 v46, v22, v23, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v47, v25, v26, v27 bitop3:0x96\n"
                         "v_bitop3_b32 v40, v28, v29, v30 bitop3:0x96\n"
                         "v_bitop3_b32 v41, v31, v32, v33 bitop3:0x96\n"
                         "v_bitop3_b32 v42, v34, v35, v36 bitop3:0x96\n"
                         "v_bitop3_b32 v43, v21, v22, v23 bitop3:0x96\n"
                         "v_bitop3_b32 v44, v24, v25, v26 bitop3:0x96\n"
                         "v_bitop3_b32 v45, v27, v28, v29 bitop3:0x96\n"
                         "v_bitop3_b32 v46, v30, v31, v32 bitop3:0x96\n"
                         "v_bitop3_b32 v47, v33, v34, v35 bitop3:0x96\n"
                         "v_bitop3_b32 v40, v20, v21, v22 bitop3:0x96\n"
                         "v_bitop3_b32 v41, v23, v24, v25 bitop3:0x96\n"
                         "v_bitop3_b32 v42, v26, v27, v28 bitop3:0x96\n"
                         "v_bitop3_b32 v43, v29, v30, v31 bitop3:0x96\n"
                         "v_bitop3_b32 v44, v32, v33, v34 bitop3:0x96\n"
                         "v_bitop3_b32 v45, v35, v36, v37 bitop3:0x96\n"
                         "v_bitop3_b32 v46, v22, v23, v24 bitop3:0x96\n"
                         "v_bitop3_b32 v47, v25, v26, v27 bitop3:0x96\n"
                         "v_bitop3_b32 v40, v28, v29, v30 bitop3:0x96\n"
                         "v_bitop3_b32 v41, v31, v32, v33 bitop3:0x96\n"
                         "v_bitop3_b32 v42, v34, v35, v36 bitop3:0x96\n"
                         "v_bitop3_b32 v43, v21, v22, v23 bitop3:0x96\n"
                         "v_bitop3_b32 v44, v24, v25, v26 bitop3:0x96\n"
                         "v_bitop3_b32 v45, v27, v28, v29 bitop3:0x96\n"
                         "v_bitop3_b32 v46, v30, v31, v32 bitop3:0x96\n"
                         "v_bitop3_b32 v47, v33, v34, v35 bitop3:0x96\n"
                         : "+v"(acc) : : CLOB, "v23", "v25", "v26", "v27", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37");
    } else {
        for (int it = 0; it < iters_a; it++)
            asm volatile("v_alignbit_b32 v40, v20, v20, 13\n"
                         "v_alignbit_b32 v41, v23, v23, 13\n"
                         "v_alignbit_b32 v42, v26, v26, 13\n"
                         "v_alignbit_b32 v43, v29, v29, 13\n"
                         "v_alignbit_b32 v44, v32, v32, 13\n"
                         "v_alignbit_b32 v45, v35, v35, 13\n"
                         "v_alignbit_b32 v46, v22, v22, 13\n"
                         "v_alignbit_b32 v47, v25, v25, 13\n"
                         "v_alignbit_b32 v40, v28, v28, 13\n"
                         "v_alignbit_b32 v41, v31, v31, 13\n"
                         "v_alignbit_b32 v42, v34, v34, 13\n"
                         "v_alignbit_b32 v43, v21, v21, 13\n"
                         "v_alignbit_b32 v44, v24, v24, 13\n"
                         "v_alignbit_b32 v45, v27, v27, 13\n"
                         "v_alignbit_b32 v46, v30, v30, 13\n"
                         "v_alignbit_b32 v47, v33, v33, 13\n"
                         "v_alignbit_b32 v40, v20, v20, 13\n"
                         "v_alignbit_b32 v41, v23, v23, 13\n"
                         "v_alignbit_b32 v42, v26, v26, 13\n"
                         "v_alignbit_b32 v43, v29, v29, 13\n"
                         "v_alignbit_b32 v44, v32, v32, 13\n"
                         "v_alignbit_b32 v45, v35, v35, 13\n"
                         "v_alignbit_b32 v46, v22, v22, 13\n"
                         "v_alignbit_b32 v47, v25, v25, 13\n"
                         "v_alignbit_b32 v40, v28, v28, 13\n"
                         "v_alignbit_b32 v41, v31, v31, 13\n"
                         "v_alignbit_b32 v42, v34, v34, 13\n"
                         "v_alignbit_b32 v43, v21, v21, 13\n"
                         "v_alignbit_b32 v44, v24, v24, 13\n"
                         "v_alignbit_b32 v45, v27, v27, 13\n"
                         "v_alignbit_b32 v46, v30, v30, 13\n"
                         "v_alignbit_b32 v47, v33, v33, 13\n"
                         : "+v"(acc) : : CLOB, "v23", "v25", "v26", "v27", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37");
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_mix(uint32_t *out, int iters) {
    uint32_t acc = threadIdx.x;
    for (int it = 0; it < iters; it++)
        asm volatile("v_bitop3_b32 v40, v20, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v20, v21, v22 bitop3:0x96\n"
                     "v_alignbit_b32 v42, v21, v21, 13\n"
                     "v_bitop3_b32 v43, v20, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v20, v21, v22 bitop3:0x96\n"
                     "v_alignbit_b32 v45, v21, v21, 13\n"
                     "v_bitop3_b32 v46, v20, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v20, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v40, v20, v21, v22 bitop3:0x96\n"
                     "v_alignbit_b32 v41, v21, v21, 13\n"
                     "v_bitop3_b32 v42, v20, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v20, v21, v22 bitop3:0x96\n"
                     "v_alignbit_b32 v44, v21, v21, 13\n"
                     "v_bitop3_b32 v45, v20, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v20, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v20, v21, v22 bitop3:0x96\n"
                     "v_alignbit_b32 v40, v21, v21, 13\n"
                     "v_bitop3_b32 v41, v20, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v20, v21, v22 bitop3:0x96\n"
                     "v_alignbit_b32 v43, v21, v21, 13\n"
                     "v_bitop3_b32 v44, v20, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v20, v21, v22 bitop3:0x96\n"
                     "v_alignbit_b32 v46, v21, v21, 13\n"
                     "v_bitop3_b32 v47, v20, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v40, v20, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v20, v21, v22 bitop3:0x96\n"
                     "v_alignbit_b32 v42, v21, v21, 13\n"
                     "v_bitop3_b32 v43, v20, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v20, v21, v22 bitop3:0x96\n"
                     "v_alignbit_b32 v45, v21, v21, 13\n"
                     "v_bitop3_b32 v46, v20, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v20, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v40, v20, v21, v22 bitop3:0x96\n"
                     "v_alignbit_b32 v41, v21, v21, 13\n"
                     : "+v"(acc) : : CLOB);
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_one0(uint32_t *out, int iters) {
    uint32_t acc = threadIdx.x;
    for (int it = 0; it < iters; it++)
        asm volatile("v_lshlrev_b32 v40, 13, v21\n"
                     "v_lshlrev_b32 v41, 13, v21\n"
                     "v_lshlrev_b32 v42, 13, v21\n"
                     "v_lshlrev_b32 v43, 13, v21\n"
                     "v_lshlrev_b32 v44, 13, v21\n"
                     "v_lshlrev_b32 v45, 13, v21\n"
                     "v_lshlrev_b32 v46, 13, v21\n"
                     "v_lshlrev_b32 v47, 13, v21\n"
                     "v_lshlrev_b32 v40, 13, v21\n"
                     "v_lshlrev_b32 v41, 13, v21\n"
                     "v_lshlrev_b32 v42, 13, v21\n"
                     "v_lshlrev_b32 v43, 13, v21\n"
                     "v_lshlrev_b32 v44, 13, v21\n"
                     "v_lshlrev_b32 v45, 13, v21\n"
                     "v_lshlrev_b32 v46, 13, v21\n"
                     "v_lshlrev_b32 v47, 13, v21\n"
                     "v_lshlrev_b32 v40, 13, v21\n"
                     "v_lshlrev_b32 v41, 13, v21\n"
                     "v_lshlrev_b32 v42, 13, v21\n"
                     "v_lshlrev_b32 v43, 13, v21\n"
                     "v_lshlrev_b32 v44, 13, v21\n"
                     "v_lshlrev_b32 v45, 13, v21\n"
                     "v_lshlrev_b32 v46, 13, v21\n"
                     "v_lshlrev_b32 v47, 13, v21\n"
                     "v_lshlrev_b32 v40, 13, v21\n"
                     "v_lshlrev_b32 v41, 13, v21\n"
                     "v_lshlrev_b32 v42, 13, v21\n"
                     "v_lshlrev_b32 v43, 13, v21\n"
                     "v_lshlrev_b32 v44, 13, v21\n"
                     "v_lshlrev_b32 v45, 13, v21\n"
                     "v_lshlrev_b32 v46, 13, v21\n"
                     "v_lshlrev_b32 v47, 13, v21\n"
                     : "+v"(acc) : : CLOB);
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_one1(uint32_t *out, int iters) {
    uint32_t acc = threadIdx.x;
    for (int it = 0; it < iters; it++)
        asm volatile("v_lshrrev_b32 v40, 19, v21\n"
                     "v_lshrrev_b32 v41, 19, v21\n"
                     "v_lshrrev_b32 v42, 19, v21\n"
                     "v_lshrrev_b32 v43, 19, v21\n"
                     "v_lshrrev_b32 v44, 19, v21\n"
                     "v_lshrrev_b32 v45, 19, v21\n"
                     "v_lshrrev_b32 v46, 19, v21\n"
                     "v_lshrrev_b32 v47, 19, v21\n"
                     "v_lshrrev_b32 v40, 19, v21\n"
                     "v_lshrrev_b32 v41, 19, v21\n"
                     "v_lshrrev_b32 v42, 19, v21\n"
                     "v_lshrrev_b32 v43, 19, v21\n"
                     "v_lshrrev_b32 v44, 19, v21\n"
                     "v_lshrrev_b32 v45, 19, v21\n"
                     "v_lshrrev_b32 v46, 19, v21\n"
                     "v_lshrrev_b32 v47, 19, v21\n"
                     "v_lshrrev_b32 v40, 19, v21\n"
                     "v_lshrrev_b32 v41, 19, v21\n"
                     "v_lshrrev_b32 v42, 19, v21\n"
                     "v_lshrrev_b32 v43, 19, v21\n"
                     "v_lshrrev_b32 v44, 19, v21\n"
                     "v_lshrrev_b32 v45, 19, v21\n"
                     "v_lshrrev_b32 v46, 19, v21\n"
                     "v_lshrrev_b32 v47, 19, v21\n"
                     "v_lshrrev_b32 v40, 19, v21\n"
                     "v_lshrrev_b32 v41, 19, v21\n"
                     "v_lshrrev_b32 v42, 19, v21\n"
                     "v_lshrrev_b32 v43, 19, v21\n"
                     "v_lshrrev_b32 v44, 19, v21\n"
                     "v_lshrrev_b32 v45, 19, v21\n"
                     "v_lshrrev_b32 v46, 19, v21\n"
                     "v_lshrrev_b32 v47, 19, v21\n"
                     : "+v"(acc) : : CLOB);
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_one2(uint32_t *out, int iters) {
    uint32_t acc = threadIdx.x;
    for (int it = 0; it < iters; it++)
        asm volatile("v_lshl_or_b32 v40, v20, 13, v21\n"
                     "v_lshl_or_b32 v41, v20, 13, v21\n"
                     "v_lshl_or_b32 v42, v20, 13, v21\n"
                     "v_lshl_or_b32 v43, v20, 13, v21\n"
                     "v_lshl_or_b32 v44, v20, 13, v21\n"
                     "v_lshl_or_b32 v45, v20, 13, v21\n"
                     "v_lshl_or_b32 v46, v20, 13, v21\n"
                     "v_lshl_or_b32 v47, v20, 13, v21\n"
                     "v_lshl_or_b32 v40, v20, 13, v21\n"
                     "v_lshl_or_b32 v41, v20, 13, v21\n"
                     "v_lshl_or_b32 v42, v20, 13, v21\n"
                     "v_lshl_or_b32 v43, v20, 13, v21\n"
                     "v_lshl_or_b32 v44, v20, 13, v21\n"
                     "v_lshl_or_b32 v45, v20, 13, v21\n"
                     "v_lshl_or_b32 v46, v20, 13, v21\n"
                     "v_lshl_or_b32 v47, v20, 13, v21\n"
                     "v_lshl_or_b32 v40, v20, 13, v21\n"
                     "v_lshl_or_b32 v41, v20, 13, v21\n"
                     "v_lshl_or_b32 v42, v20, 13, v21\n"
                     "v_lshl_or_b32 v43, v20, 13, v21\n"
                     "v_lshl_or_b32 v44, v20, 13, v21\n"
                     "v_lshl_or_b32 v45, v20, 13, v21\n"
                     "v_lshl_or_b32 v46, v20, 13, v21\n"
                     "v_lshl_or_b32 v47, v20, 13, v21\n"
                     "v_lshl_or_b32 v40, v20, 13, v21\n"
                     "v_lshl_or_b32 v41, v20, 13, v21\n"
                     "v_lshl_or_b32 v42, v20, 13, v21\n"
                     "v_lshl_or_b32 v43, v20, 13, v21\n"
                     "v_lshl_or_b32 v44, v20, 13, v21\n"
                     "v_lshl_or_b32 v45, v20, 13, v21\n"
                     "v_lshl_or_b32 v46, v20, 13, v21\n"
                     "v_lshl_or_b32 v47, v20, 13, v21\n"
                     : "+v"(acc) : : CLOB);
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_one3(uint32_t *out, int iters) {
    uint32_t acc = threadIdx.x;
    for (int it = 0; it < iters; it++)
        asm volatile("v_lshl_add_u32 v40, v20, 13, v21\n"
                     "v_lshl_add_u32 v41, v20, 13, v21\n"
                     "v_lshl_add_u32 v42, v20, 13, v21\n"
                     "v_lshl_add_u32 v43, v20, 13, v21\n"
                     "v_lshl_add_u32 v44, v20, 13, v21\n"
                     "v_lshl_add_u32 v45, v20, 13, v21\n"
                     "v_lshl_add_u32 v46, v20, 13, v21\n"
                     "v_lshl_add_u32 v47, v20, 13, v21\n"
                     "v_lshl_add_u32 v40, v20, 13, v21\n"
                     "v_lshl_add_u32 v41, v20, 13, v21\n"
                     "v_lshl_add_u32 v42, v20, 13, v21\n"
                     "v_lshl_add_u32 v43, v20, 13, v21\n"
                     "v_lshl_add_u32 v44, v20, 13, v21\n"
                     "v_lshl_add_u32 v45, v20, 13, v21\n"
                     "v_lshl_add_u32 v46, v20, 13, v21\n"
                     "v_lshl_add_u32 v47, v20, 13, v21\n"
                     "v_lshl_add_u32 v40, v20, 13, v21\n"
                     "v_lshl_add_u32 v41, v20, 13, v21\n"
                     "v_lshl_add_u32 v42, v20, 13, v21\n"
                     "v_lshl_add_u32 v43, v20, 13, v21\n"
                     "v_lshl_add_u32 v44, v20, 13, v21\n"
                     "v_lshl_add_u32 v45, v20, 13, v21\n"
                     "v_lshl_add_u32 v46, v20, 13, v21\n"
                     "v_lshl_add_u32 v47, v20, 13, v21\n"
                     "v_lshl_add_u32 v40, v20, 13, v21\n"
                     "v_lshl_add_u32 v41, v20, 13, v21\n"
                     "v_lshl_add_u32 v42, v20, 13, v21\n"
                     "v_lshl_add_u32 v43, v20, 13, v21\n"
                     "v_lshl_add_u32 v44, v20, 13, v21\n"
                     "v_lshl_add_u32 v45, v20, 13, v21\n"
                     "v_lshl_add_u32 v46, v20, 13, v21\n"
                     "v_lshl_add_u32 v47, v20, 13, v21\n"
                     : "+v"(acc) : : CLOB);
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_one4(uint32_t *out, int iters) {
    uint32_t acc = threadIdx.x;
    for (int it = 0; it < iters; it++)
        asm volatile("v_and_or_b32 v40, v20, v21, v22\n"
                     "v_and_or_b32 v41, v20, v21, v22\n"
                     "v_and_or_b32 v42, v20, v21, v22\n"
                     "v_and_or_b32 v43, v20, v21, v22\n"
                     "v_and_or_b32 v44, v20, v21, v22\n"
                     "v_and_or_b32 v45, v20, v21, v22\n"
                     "v_and_or_b32 v46, v20, v21, v22\n"
                     "v_and_or_b32 v47, v20, v21, v22\n"
                     "v_and_or_b32 v40, v20, v21, v22\n"
                     "v_and_or_b32 v41, v20, v21, v22\n"
                     "v_and_or_b32 v42, v20, v21, v22\n"
                     "v_and_or_b32 v43, v20, v21, v22\n"
                     "v_and_or_b32 v44, v20, v21, v22\n"
                     "v_and_or_b32 v45, v20, v21, v22\n"
                     "v_and_or_b32 v46, v20, v21, v22\n"
                     "v_and_or_b32 v47, v20, v21, v22\n"
                     "v_and_or_b32 v40, v20, v21, v22\n"
                     "v_and_or_b32 v41, v20, v21, v22\n"
                     "v_and_or_b32 v42, v20, v21, v22\n"
                     "v_and_or_b32 v43, v20, v21, v22\n"
                     "v_and_or_b32 v44, v20, v21, v22\n"
                     "v_and_or_b32 v45, v20, v21, v22\n"
                     "v_and_or_b32 v46, v20, v21, v22\n"
                     "v_and_or_b32 v47, v20, v21, v22\n"
                     "v_and_or_b32 v40, v20, v21, v22\n"
                     "v_and_or_b32 v41, v20, v21, v22\n"
                     "v_and_or_b32 v42, v20, v21, v22\n"
                     "v_and_or_b32 v43, v20, v21, v22\n"
                     "v_and_or_b32 v44, v20, v21, v22\n"
                     "v_and_or_b32 v45, v20, v21, v22\n"
                     "v_and_or_b32 v46, v20, v21, v22\n"
                     "v_and_or_b32 v47, v20, v21, v22\n"
                     : "+v"(acc) : : CLOB);
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_one5(uint32_t *out, int iters) {
    uint32_t acc = threadIdx.x;
    for (int it = 0; it < iters; it++)
        asm volatile("v_or3_b32 v40, v20, v21, v22\n"
                     "v_or3_b32 v41, v20, v21, v22\n"
                     "v_or3_b32 v42, v20, v21, v22\n"
                     "v_or3_b32 v43, v20, v21, v22\n"
                     "v_or3_b32 v44, v20, v21, v22\n"
                     "v_or3_b32 v45, v20, v21, v22\n"
                     "v_or3_b32 v46, v20, v21, v22\n"
                     "v_or3_b32 v47, v20, v21, v22\n"
                     "v_or3_b32 v40, v20, v21, v22\n"
                     "v_or3_b32 v41, v20, v21, v22\n"
                     "v_or3_b32 v42, v20, v21, v22\n"
                     "v_or3_b32 v43, v20, v21, v22\n"
                     "v_or3_b32 v44, v20, v21, v22\n"
                     "v_or3_b32 v45, v20, v21, v22\n"
                     "v_or3_b32 v46, v20, v21, v22\n"
                     "v_or3_b32 v47, v20, v21, v22\n"
                     "v_or3_b32 v40, v20, v21, v22\n"
                     "v_or3_b32 v41, v20, v21, v22\n"
                     "v_or3_b32 v42, v20, v21, v22\n"
                     "v_or3_b32 v43, v20, v21, v22\n"
                     "v_or3_b32 v44, v20, v21, v22\n"
                     "v_or3_b32 v45, v20, v21, v22\n"
                     "v_or3_b32 v46, v20, v21, v22\n"
                     "v_or3_b32 v47, v20, v21, v22\n"
                     "v_or3_b32 v40, v20, v21, v22\n"
                     "v_or3_b32 v41, v20, v21, v22\n"
                     "v_or3_b32 v42, v20, v21, v22\n"
                     "v_or3_b32 v43, v20, v21, v22\n"
                     "v_or3_b32 v44, v20, v21, v22\n"
                     "v_or3_b32 v45, v20, v21, v22\n"
                     "v_or3_b32 v46, v20, v21, v22\n"
                     "v_or3_b32 v47, v20, v21, v22\n"
                     : "+v"(acc) : : CLOB);
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_one6(uint32_t *out, int iters) {
    uint32_t acc = threadIdx.x;
    for (int it = 0; it < iters; it++)
        asm volatile("v_perm_b32 v40, v20, v21, v22\n"
                     "v_perm_b32 v41, v20, v21, v22\n"
                     "v_perm_b32 v42, v20, v21, v22\n"
                     "v_perm_b32 v43, v20, v21, v22\n"
                     "v_perm_b32 v44, v20, v21, v22\n"
                     "v_perm_b32 v45, v20, v21, v22\n"
                     "v_perm_b32 v46, v20, v21, v22\n"
                     "v_perm_b32 v47, v20, v21, v22\n"
                     "v_perm_b32 v40, v20, v21, v22\n"
                     "v_perm_b32 v41, v20, v21, v22\n"
                     "v_perm_b32 v42, v20, v21, v22\n"
                     "v_perm_b32 v43, v20, v21, v22\n"
                     "v_perm_b32 v44, v20, v21, v22\n"
                     "v_perm_b32 v45, v20, v21, v22\n"
                     "v_perm_b32 v46, v20, v21, v22\n"
                     "v_perm_b32 v47, v20, v21, v22\n"
                     "v_perm_b32 v40, v20, v21, v22\n"
                     "v_perm_b32 v41, v20, v21, v22\n"
                     "v_perm_b32 v42, v20, v21, v22\n"
                     "v_perm_b32 v43, v20, v21, v22\n"
                     "v_perm_b32 v44, v20, v21, v22\n"
                     "v_perm_b32 v45, v20, v21, v22\n"
                     "v_perm_b32 v46, v20, v21, v22\n"
                     "v_perm_b32 v47, v20, v21, v22\n"
                     "v_perm_b32 v40, v20, v21, v22\n"
                     "v_perm_b32 v41, v20, v21, v22\n"
                     "v_perm_b32 v42, v20, v21, v22\n"
                     "v_perm_b32 v43, v20, v21, v22\n"
                     "v_perm_b32 v44, v20, v21, v22\n"
                     "v_perm_b32 v45, v20, v21, v22\n"
                     "v_perm_b32 v46, v20, v21, v22\n"
                     "v_perm_b32 v47, v20, v21, v22\n"
                     : "+v"(acc) : : CLOB);
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_one7(uint32_t *out, int iters) {
    uint32_t acc = threadIdx.x;
    for (int it = 0; it < iters; it++)
        asm volatile("v_alignbyte_b32 v40, v20, v21, 1\n"
                     "v_alignbyte_b32 v41, v20, v21, 1\n"
                     "v_alignbyte_b32 v42, v20, v21, 1\n"
                     "v_alignbyte_b32 v43, v20, v21, 1\n"
                     "v_alignbyte_b32 v44, v20, v21, 1\n"
                     "v_alignbyte_b32 v45, v20, v21, 1\n"
                     "v_alignbyte_b32 v46, v20, v21, 1\n"
                     "v_alignbyte_b32 v47, v20, v21, 1\n"
                     "v_alignbyte_b32 v40, v20, v21, 1\n"
                     "v_alignbyte_b32 v41, v20, v21, 1\n"
                     "v_alignbyte_b32 v42, v20, v21, 1\n"
                     "v_alignbyte_b32 v43, v20, v21, 1\n"
                     "v_alignbyte_b32 v44, v20, v21, 1\n"
                     "v_alignbyte_b32 v45, v20, v21, 1\n"
                     "v_alignbyte_b32 v46, v20, v21, 1\n"
                     "v_alignbyte_b32 v47, v20, v21, 1\n"
                     "v_alignbyte_b32 v40, v20, v21, 1\n"
                     "v_alignbyte_b32 v41, v20, v21, 1\n"
                     "v_alignbyte_b32 v42, v20, v21, 1\n"
                     "v_alignbyte_b32 v43, v20, v21, 1\n"
                     "v_alignbyte_b32 v44, v20, v21, 1\n"
                     "v_alignbyte_b32 v45, v20, v21, 1\n"
                     "v_alignbyte_b32 v46, v20, v21, 1\n"
                     "v_alignbyte_b32 v47, v20, v21, 1\n"
                     "v_alignbyte_b32 v40, v20, v21, 1\n"
                     "v_alignbyte_b32 v41, v20, v21, 1\n"
                     "v_alignbyte_b32 v42, v20, v21, 1\n"
                     "v_alignbyte_b32 v43, v20, v21, 1\n"
                     "v_alignbyte_b32 v44, v20, v21, 1\n"
                     "v_alignbyte_b32 v45, v20, v21, 1\n"
                     "v_alignbyte_b32 v46, v20, v21, 1\n"
                     "v_alignbyte_b32 v47, v20, v21, 1\n"
                     : "+v"(acc) : : CLOB);
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_one8(uint32_t *out, int iters) {
    uint32_t acc = threadIdx.x;
    for (int it = 0; it < iters; it++)
        asm volatile("v_bfe_u32 v40, v20, 5, 13\n"
                     "v_bfe_u32 v41, v20, 5, 13\n"
                     "v_bfe_u32 v42, v20, 5, 13\n"
                     "v_bfe_u32 v43, v20, 5, 13\n"
                     "v_bfe_u32 v44, v20, 5, 13\n"
                     "v_bfe_u32 v45, v20, 5, 13\n"
                     "v_bfe_u32 v46, v20, 5, 13\n"
                     "v_bfe_u32 v47, v20, 5, 13\n"
                     "v_bfe_u32 v40, v20, 5, 13\n"
                     "v_bfe_u32 v41, v20, 5, 13\n"
                     "v_bfe_u32 v42, v20, 5, 13\n"
                     "v_bfe_u32 v43, v20, 5, 13\n"
                     "v_bfe_u32 v44, v20, 5, 13\n"
                     "v_bfe_u32 v45, v20, 5, 13\n"
                     "v_bfe_u32 v46, v20, 5, 13\n"
                     "v_bfe_u32 v47, v20, 5, 13\n"
                     "v_bfe_u32 v40, v20, 5, 13\n"
                     "v_bfe_u32 v41, v20, 5, 13\n"
                     "v_bfe_u32 v42, v20, 5, 13\n"
                     "v_bfe_u32 v43, v20, 5, 13\n"
                     "v_bfe_u32 v44, v20, 5, 13\n"
                     "v_bfe_u32 v45, v20, 5, 13\n"
                     "v_bfe_u32 v46, v20, 5, 13\n"
                     "v_bfe_u32 v47, v20, 5, 13\n"
                     "v_bfe_u32 v40, v20, 5, 13\n"
                     "v_bfe_u32 v41, v20, 5, 13\n"
                     "v_bfe_u32 v42, v20, 5, 13\n"
                     "v_bfe_u32 v43, v20, 5, 13\n"
                     "v_bfe_u32 v44, v20, 5, 13\n"
                     "v_bfe_u32 v45, v20, 5, 13\n"
                     "v_bfe_u32 v46, v20, 5, 13\n"
                     "v_bfe_u32 v47, v20, 5, 13\n"
                     : "+v"(acc) : : CLOB);
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_one9(uint32_t *out, int iters) {
    uint32_t acc = threadIdx.x;
    for (int it = 0; it < iters; it++)
        asm volatile("v_bfi_b32 v40, v20, v21, v22\n"
                     "v_bfi_b32 v41, v20, v21, v22\n"
                     "v_bfi_b32 v42, v20, v21, v22\n"
                     "v_bfi_b32 v43, v20, v21, v22\n"
                     "v_bfi_b32 v44, v20, v21, v22\n"
                     "v_bfi_b32 v45, v20, v21, v22\n"
                     "v_bfi_b32 v46, v20, v21, v22\n"
                     "v_bfi_b32 v47, v20, v21, v22\n"
                     "v_bfi_b32 v40, v20, v21, v22\n"
                     "v_bfi_b32 v41, v20, v21, v22\n"
                     "v_bfi_b32 v42, v20, v21, v22\n"
                     "v_bfi_b32 v43, v20, v21, v22\n"
                     "v_bfi_b32 v44, v20, v21, v22\n"
                     "v_bfi_b32 v45, v20, v21, v22\n"
                     "v_bfi_b32 v46, v20, v21, v22\n"
                     "v_bfi_b32 v47, v20, v21, v22\n"
                     "v_bfi_b32 v40, v20, v21, v22\n"
                     "v_bfi_b32 v41, v20, v21, v22\n"
                     "v_bfi_b32 v42, v20, v21, v22\n"
                     "v_bfi_b32 v43, v20, v21, v22\n"
                     "v_bfi_b32 v44, v20, v21, v22\n"
                     "v_bfi_b32 v45, v20, v21, v22\n"
                     "v_bfi_b32 v46, v20, v21, v22\n"
                     "v_bfi_b32 v47, v20, v21, v22\n"
                     "v_bfi_b32 v40, v20, v21, v22\n"
                     "v_bfi_b32 v41, v20, v21, v22\n"
                     "v_bfi_b32 v42, v20, v21, v22\n"
                     "v_bfi_b32 v43, v20, v21, v22\n"
                     "v_bfi_b32 v44, v20, v21, v22\n"
                     "v_bfi_b32 v45, v20, v21, v22\n"
                     "v_bfi_b32 v46, v20, v21, v22\n"
                     "v_bfi_b32 v47, v20, v21, v22\n"
                     : "+v"(acc) : : CLOB);
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_one10(uint32_t *out, int iters) {
    uint32_t acc = threadIdx.x;
    for (int it = 0; it < iters; it++)
        asm volatile("v_mov_b32 v40, v20\n"
                     "v_mov_b32 v41, v20\n"
                     "v_mov_b32 v42, v20\n"
                     "v_mov_b32 v43, v20\n"
                     "v_mov_b32 v44, v20\n"
                     "v_mov_b32 v45, v20\n"
                     "v_mov_b32 v46, v20\n"
                     "v_mov_b32 v47, v20\n"
                     "v_mov_b32 v40, v20\n"
                     "v_mov_b32 v41, v20\n"
                     "v_mov_b32 v42, v20\n"
                     "v_mov_b32 v43, v20\n"
                     "v_mov_b32 v44, v20\n"
                     "v_mov_b32 v45, v20\n"
                     "v_mov_b32 v46, v20\n"
                     "v_mov_b32 v47, v20\n"
                     "v_mov_b32 v40, v20\n"
                     "v_mov_b32 v41, v20\n"
                     "v_mov_b32 v42, v20\n"
                     "v_mov_b32 v43, v20\n"
                     "v_mov_b32 v44, v20\n"
                     "v_mov_b32 v45, v20\n"
                     "v_mov_b32 v46, v20\n"
                     "v_mov_b32 v47, v20\n"
                     "v_mov_b32 v40, v20\n"
                     "v_mov_b32 v41, v20\n"
                     "v_mov_b32 v42, v20\n"
                     "v_mov_b32 v43, v20\n"
                     "v_mov_b32 v44, v20\n"
                     "v_mov_b32 v45, v20\n"
                     "v_mov_b32 v46, v20\n"
                     "v_mov_b32 v47, v20\n"
                     : "+v"(acc) : : CLOB);
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_one11(uint32_t *out, int iters) {
    uint32_t acc = threadIdx.x;
    for (int it = 0; it < iters; it++)
        asm volatile("v_add_u32 v40, v20, v21\n"
                     "v_add_u32 v41, v20, v21\n"
                     "v_add_u32 v42, v20, v21\n"
                     "v_add_u32 v43, v20, v21\n"
                     "v_add_u32 v44, v20, v21\n"
                     "v_add_u32 v45, v20, v21\n"
                     "v_add_u32 v46, v20, v21\n"
                     "v_add_u32 v47, v20, v21\n"
                     "v_add_u32 v40, v20, v21\n"
                     "v_add_u32 v41, v20, v21\n"
                     "v_add_u32 v42, v20, v21\n"
                     "v_add_u32 v43, v20, v21\n"
                     "v_add_u32 v44, v20, v21\n"
                     "v_add_u32 v45, v20, v21\n"
                     "v_add_u32 v46, v20, v21\n"
                     "v_add_u32 v47, v20, v21\n"
                     "v_add_u32 v40, v20, v21\n"
                     "v_add_u32 v41, v20, v21\n"
                     "v_add_u32 v42, v20, v21\n"
                     "v_add_u32 v43, v20, v21\n"
                     "v_add_u32 v44, v20, v21\n"
                     "v_add_u32 v45, v20, v21\n"
                     "v_add_u32 v46, v20, v21\n"
                     "v_add_u32 v47, v20, v21\n"
                     "v_add_u32 v40, v20, v21\n"
                     "v_add_u32 v41, v20, v21\n"
                     "v_add_u32 v42, v20, v21\n"
                     "v_add_u32 v43, v20, v21\n"
                     "v_add_u32 v44, v20, v21\n"
                     "v_add_u32 v45, v20, v21\n"
                     "v_add_u32 v46, v20, v21\n"
                     "v_add_u32 v47, v20, v21\n"
                     : "+v"(acc) : : CLOB);
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_dist1(uint32_t *out, int iters) {
    uint32_t acc = threadIdx.x;
    for (int it = 0; it < iters; it++)
        asm volatile("v_bitop3_b32 v40, v55, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v40, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v41, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v42, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v43, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v44, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v45, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v46, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v48, v47, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v49, v48, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v50, v49, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v51, v50, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v52, v51, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v53, v52, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v54, v53, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v55, v54, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v40, v55, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v40, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v41, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v42, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v43, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v44, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v45, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v46, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v48, v47, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v49, v48, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v50, v49, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v51, v50, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v52, v51, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v53, v52, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v54, v53, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v55, v54, v21, v22 bitop3:0x96\n"
                     : "+v"(acc) : : CLOB, "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_dist2(uint32_t *out, int iters) {
    uint32_t acc = threadIdx.x;
    for (int it = 0; it < iters; it++)
        asm volatile("v_bitop3_b32 v40, v54, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v55, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v40, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v41, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v42, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v43, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v44, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v45, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v48, v46, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v49, v47, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v50, v48, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v51, v49, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v52, v50, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v53, v51, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v54, v52, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v55, v53, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v40, v54, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v55, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v40, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v41, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v42, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v43, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v44, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v45, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v48, v46, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v49, v47, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v50, v48, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v51, v49, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v52, v50, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v53, v51, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v54, v52, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v55, v53, v21, v22 bitop3:0x96\n"
                     : "+v"(acc) : : CLOB, "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_dist3(uint32_t *out, int iters) {
    uint32_t acc = threadIdx.x;
    for (int it = 0; it < iters; it++)
        asm volatile("v_bitop3_b32 v40, v53, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v54, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v55, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v40, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v41, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v42, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v43, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v44, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v48, v45, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v49, v46, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v50, v47, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v51, v48, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v52, v49, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v53, v50, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v54, v51, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v55, v52, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v40, v53, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v54, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v55, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v40, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v41, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v42, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v43, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v44, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v48, v45, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v49, v46, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v50, v47, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v51, v48, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v52, v49, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v53, v50, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v54, v51, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v55, v52, v21, v22 bitop3:0x96\n"
                     : "+v"(acc) : : CLOB, "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_dist4(uint32_t *out, int iters) {
    uint32_t acc = threadIdx.x;
    for (int it = 0; it < iters; it++)
        asm volatile("v_bitop3_b32 v40, v52, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v53, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v54, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v55, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v40, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v41, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v42, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v43, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v48, v44, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v49, v45, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v50, v46, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v51, v47, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v52, v48, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v53, v49, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v54, v50, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v55, v51, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v40, v52, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v53, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v54, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v55, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v40, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v41, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v42, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v43, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v48, v44, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v49, v45, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v50, v46, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v51, v47, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v52, v48, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v53, v49, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v54, v50, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v55, v51, v21, v22 bitop3:0x96\n"
                     : "+v"(acc) : : CLOB, "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_dist6(uint32_t *out, int iters) {
    uint32_t acc = threadIdx.x;
    for (int it = 0; it < iters; it++)
        asm volatile("v_bitop3_b32 v40, v50, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v51, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v52, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v53, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v54, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v55, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v40, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v41, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v48, v42, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v49, v43, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v50, v44, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v51, v45, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v52, v46, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v53, v47, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v54, v48, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v55, v49, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v40, v50, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v51, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v52, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v53, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v54, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v55, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v40, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v41, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v48, v42, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v49, v43, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v50, v44, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v51, v45, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v52, v46, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v53, v47, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v54, v48, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v55, v49, v21, v22 bitop3:0x96\n"
                     : "+v"(acc) : : CLOB, "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_runs2(uint32_t *out, int iters) {
    uint32_t acc = threadIdx.x;
    for (int it = 0; it < iters; it++)
        asm volatile("v_bitop3_b32 v40, v20, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v23, v24, v25 bitop3:0x96\n"
                     "v_alignbit_b32 v42, v26, v26, 13\n"
                     : "+v"(acc) : : CLOB, "v23", "v25", "v26", "v27", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37");
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_runs5(uint32_t *out, int iters) {
    uint32_t acc = threadIdx.x;
    for (int it = 0; it < iters; it++)
        asm volatile("v_bitop3_b32 v40, v20, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v23, v24, v25 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v26, v27, v28 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v29, v30, v31 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v32, v33, v34 bitop3:0x96\n"
                     "v_alignbit_b32 v45, v35, v35, 13\n"
                     "v_alignbit_b32 v46, v22, v22, 13\n"
                     : "+v"(acc) : : CLOB, "v23", "v25", "v26", "v27", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37");
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_runs12(uint32_t *out, int iters) {
    uint32_t acc = threadIdx.x;
    for (int it = 0; it < iters; it++)
        asm volatile("v_bitop3_b32 v40, v20, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v23, v24, v25 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v26, v27, v28 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v29, v30, v31 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v32, v33, v34 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v35, v36, v37 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v22, v23, v24 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v25, v26, v27 bitop3:0x96\n"
                     "v_bitop3_b32 v40, v28, v29, v30 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v31, v32, v33 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v34, v35, v36 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v21, v22, v23 bitop3:0x96\n"
                     "v_alignbit_b32 v44, v24, v24, 13\n"
                     "v_alignbit_b32 v45, v27, v27, 13\n"
                     "v_alignbit_b32 v46, v30, v30, 13\n"
                     "v_alignbit_b32 v47, v33, v33, 13\n"
                     "v_alignbit_b32 v40, v20, v20, 13\n"
                     : "+v"(acc) : : CLOB, "v23", "v25", "v26", "v27", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37");
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_runs24(uint32_t *out, int iters) {
    uint32_t acc = threadIdx.x;
    for (int it = 0; it < iters; it++)
        asm volatile("v_bitop3_b32 v40, v20, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v23, v24, v25 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v26, v27, v28 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v29, v30, v31 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v32, v33, v34 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v35, v36, v37 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v22, v23, v24 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v25, v26, v27 bitop3:0x96\n"
                     "v_bitop3_b32 v40, v28, v29, v30 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v31, v32, v33 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v34, v35, v36 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v21, v22, v23 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v24, v25, v26 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v27, v28, v29 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v30, v31, v32 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v33, v34, v35 bitop3:0x96\n"
                     "v_bitop3_b32 v40, v20, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v23, v24, v25 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v26, v27, v28 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v29, v30, v31 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v32, v33, v34 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v35, v36, v37 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v22, v23, v24 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v25, v26, v27 bitop3:0x96\n"
                     "v_alignbit_b32 v40, v28, v28, 13\n"
                     "v_alignbit_b32 v41, v31, v31, 13\n"
                     "v_alignbit_b32 v42, v34, v34, 13\n"
                     "v_alignbit_b32 v43, v21, v21, 13\n"
                     "v_alignbit_b32 v44, v24, v24, 13\n"
                     "v_alignbit_b32 v45, v27, v27, 13\n"
                     "v_alignbit_b32 v46, v30, v30, 13\n"
                     "v_alignbit_b32 v47, v33, v33, 13\n"
                     "v_alignbit_b32 v40, v20, v20, 13\n"
                     "v_alignbit_b32 v41, v23, v23, 13\n"
                     : "+v"(acc) : : CLOB, "v23", "v25", "v26", "v27", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37");
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_runs48(uint32_t *out, int iters) {
    uint32_t acc = threadIdx.x;
    for (int it = 0; it < iters; it++)
        asm volatile("v_bitop3_b32 v40, v20, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v23, v24, v25 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v26, v27, v28 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v29, v30, v31 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v32, v33, v34 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v35, v36, v37 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v22, v23, v24 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v25, v26, v27 bitop3:0x96\n"
                     "v_bitop3_b32 v40, v28, v29, v30 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v31, v32, v33 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v34, v35, v36 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v21, v22, v23 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v24, v25, v26 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v27, v28, v29 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v30, v31, v32 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v33, v34, v35 bitop3:0x96\n"
                     "v_bitop3_b32 v40, v20, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v23, v24, v25 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v26, v27, v28 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v29, v30, v31 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v32, v33, v34 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v35, v36, v37 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v22, v23, v24 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v25, v26, v27 bitop3:0x96\n"
                     "v_bitop3_b32 v40, v28, v29, v30 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v31, v32, v33 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v34, v35, v36 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v21, v22, v23 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v24, v25, v26 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v27, v28, v29 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v30, v31, v32 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v33, v34, v35 bitop3:0x96\n"
                     "v_bitop3_b32 v40, v20, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v23, v24, v25 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v26, v27, v28 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v29, v30, v31 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v32, v33, v34 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v35, v36, v37 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v22, v23, v24 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v25, v26, v27 bitop3:0x96\n"
                     "v_bitop3_b32 v40, v28, v29, v30 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v31, v32, v33 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v34, v35, v36 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v21, v22, v23 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v24, v25, v26 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v27, v28, v29 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v30, v31, v32 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v33, v34, v35 bitop3:0x96\n"
                     "v_alignbit_b32 v40, v20, v20, 13\n"
                     "v_alignbit_b32 v41, v23, v23, 13\n"
                     "v_alignbit_b32 v42, v26, v26, 13\n"
                     "v_alignbit_b32 v43, v29, v29, 13\n"
                     "v_alignbit_b32 v44, v32, v32, 13\n"
                     "v_alignbit_b32 v45, v35, v35, 13\n"
                     "v_alignbit_b32 v46, v22, v22, 13\n"
                     "v_alignbit_b32 v47, v25, v25, 13\n"
                     "v_alignbit_b32 v40, v28, v28, 13\n"
                     "v_alignbit_b32 v41, v31, v31, 13\n"
                     "v_alignbit_b32 v42, v34, v34, 13\n"
                     "v_alignbit_b32 v43, v21, v21, 13\n"
                     "v_alignbit_b32 v44, v24, v24, 13\n"
                     "v_alignbit_b32 v45, v27, v27, 13\n"
                     "v_alignbit_b32 v46, v30, v30, 13\n"
                     "v_alignbit_b32 v47, v33, v33, 13\n"
                     "v_alignbit_b32 v40, v20, v20, 13\n"
                     "v_alignbit_b32 v41, v23, v23, 13\n"
                     "v_alignbit_b32 v42, v26, v26, 13\n"
                     "v_alignbit_b32 v43, v29, v29, 13\n"
                     : "+v"(acc) : : CLOB, "v23", "v25", "v26", "v27", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37");
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_runs120(uint32_t *out, int iters) {
    uint32_t acc = threadIdx.x;
    for (int it = 0; it < iters; it++)
        asm volatile("v_bitop3_b32 v40, v20, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v23, v24, v25 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v26, v27, v28 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v29, v30, v31 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v32, v33, v34 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v35, v36, v37 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v22, v23, v24 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v25, v26, v27 bitop3:0x96\n"
                     "v_bitop3_b32 v40, v28, v29, v30 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v31, v32, v33 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v34, v35, v36 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v21, v22, v23 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v24, v25, v26 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v27, v28, v29 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v30, v31, v32 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v33, v34, v35 bitop3:0x96\n"
                     "v_bitop3_b32 v40, v20, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v23, v24, v25 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v26, v27, v28 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v29, v30, v31 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v32, v33, v34 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v35, v36, v37 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v22, v23, v24 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v25, v26, v27 bitop3:0x96\n"
                     "v_bitop3_b32 v40, v28, v29, v30 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v31, v32, v33 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v34, v35, v36 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v21, v22, v23 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v24, v25, v26 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v27, v28, v29 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v30, v31, v32 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v33, v34, v35 bitop3:0x96\n"
                     "v_bitop3_b32 v40, v20, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v23, v24, v25 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v26, v27, v28 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v29, v30, v31 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v32, v33, v34 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v35, v36, v37 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v22, v23, v24 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v25, v26, v27 bitop3:0x96\n"
                     "v_bitop3_b32 v40, v28, v29, v30 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v31, v32, v33 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v34, v35, v36 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v21, v22, v23 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v24, v25, v26 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v27, v28, v29 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v30, v31, v32 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v33, v34, v35 bitop3:0x96\n"
                     "v_bitop3_b32 v40, v20, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v23, v24, v25 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v26, v27, v28 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v29, v30, v31 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v32, v33, v34 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v35, v36, v37 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v22, v23, v24 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v25, v26, v27 bitop3:0x96\n"
                     "v_bitop3_b32 v40, v28, v29, v30 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v31, v32, v33 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v34, v35, v36 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v21, v22, v23 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v24, v25, v26 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v27, v28, v29 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v30, v31, v32 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v33, v34, v35 bitop3:0x96\n"
                     "v_bitop3_b32 v40, v20, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v23, v24, v25 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v26, v27, v28 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v29, v30, v31 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v32, v33, v34 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v35, v36, v37 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v22, v23, v24 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v25, v26, v27 bitop3:0x96\n"
                     "v_bitop3_b32 v40, v28, v29, v30 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v31, v32, v33 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v34, v35, v36 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v21, v22, v23 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v24, v25, v26 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v27, v28, v29 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v30, v31, v32 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v33, v34, v35 bitop3:0x96\n"
                     "v_bitop3_b32 v40, v20, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v23, v24, v25 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v26, v27, v28 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v29, v30, v31 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v32, v33, v34 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v35, v36, v37 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v22, v23, v24 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v25, v26, v27 bitop3:0x96\n"
                     "v_bitop3_b32 v40, v28, v29, v30 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v31, v32, v33 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v34, v35, v36 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v21, v22, v23 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v24, v25, v26 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v27, v28, v29 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v30, v31, v32 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v33, v34, v35 bitop3:0x96\n"
                     "v_bitop3_b32 v40, v20, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v23, v24, v25 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v26, v27, v28 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v29, v30, v31 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v32, v33, v34 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v35, v36, v37 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v22, v23, v24 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v25, v26, v27 bitop3:0x96\n"
                     "v_bitop3_b32 v40, v28, v29, v30 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v31, v32, v33 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v34, v35, v36 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v21, v22, v23 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v24, v25, v26 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v27, v28, v29 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v30, v31, v32 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v33, v34, v35 bitop3:0x96\n"
                     "v_bitop3_b32 v40, v20, v21, v22 bitop3:0x96\n"
                     "v_bitop3_b32 v41, v23, v24, v25 bitop3:0x96\n"
                     "v_bitop3_b32 v42, v26, v27, v28 bitop3:0x96\n"
                     "v_bitop3_b32 v43, v29, v30, v31 bitop3:0x96\n"
                     "v_bitop3_b32 v44, v32, v33, v34 bitop3:0x96\n"
                     "v_bitop3_b32 v45, v35, v36, v37 bitop3:0x96\n"
                     "v_bitop3_b32 v46, v22, v23, v24 bitop3:0x96\n"
                     "v_bitop3_b32 v47, v25, v26, v27 bitop3:0x96\n"
                     "v_alignbit_b32 v40, v28, v28, 13\n"
                     "v_alignbit_b32 v41, v31, v31, 13\n"
                     "v_alignbit_b32 v42, v34, v34, 13\n"
                     "v_alignbit_b32 v43, v21, v21, 13\n"
                     "v_alignbit_b32 v44, v24, v24, 13\n"
                     "v_alignbit_b32 v45, v27, v27, 13\n"
                     "v_alignbit_b32 v46, v30, v30, 13\n"
                     "v_alignbit_b32 v47, v33, v33, 13\n"
                     "v_alignbit_b32 v40, v20, v20, 13\n"
                     "v_alignbit_b32 v41, v23, v23, 13\n"
                     "v_alignbit_b32 v42, v26, v26, 13\n"
                     "v_alignbit_b32 v43, v29, v29, 13\n"
                     "v_alignbit_b32 v44, v32, v32, 13\n"
                     "v_alignbit_b32 v45, v35, v35, 13\n"
                     "v_alignbit_b32 v46, v22, v22, 13\n"
                     "v_alignbit_b32 v47, v25, v25, 13\n"
                     "v_alignbit_b32 v40, v28, v28, 13\n"
                     "v_alignbit_b32 v41, v31, v31, 13\n"
                     "v_alignbit_b32 v42, v34, v34, 13\n"
                     "v_alignbit_b32 v43, v21, v21, 13\n"
                     "v_alignbit_b32 v44, v24, v24, 13\n"
                     "v_alignbit_b32 v45, v27, v27, 13\n"
                     "v_alignbit_b32 v46, v30, v30, 13\n"
                     "v_alignbit_b32 v47, v33, v33, 13\n"
                     "v_alignbit_b32 v40, v20, v20, 13\n"
                     "v_alignbit_b32 v41, v23, v23, 13\n"
                     "v_alignbit_b32 v42, v26, v26, 13\n"
                     "v_alignbit_b32 v43, v29, v29, 13\n"
                     "v_alignbit_b32 v44, v32, v32, 13\n"
                     "v_alignbit_b32 v45, v35, v35, 13\n"
                     "v_alignbit_b32 v46, v22, v22, 13\n"
                     "v_alignbit_b32 v47, v25, v25, 13\n"
                     "v_alignbit_b32 v40, v28, v28, 13\n"
                     "v_alignbit_b32 v41, v31, v31, 13\n"
                     "v_alignbit_b32 v42, v34, v34, 13\n"
                     "v_alignbit_b32 v43, v21, v21, 13\n"
                     "v_alignbit_b32 v44, v24, v24, 13\n"
                     "v_alignbit_b32 v45, v27, v27, 13\n"
                     "v_alignbit_b32 v46, v30, v30, 13\n"
                     "v_alignbit_b32 v47, v33, v33, 13\n"
                     "v_alignbit_b32 v40, v20, v20, 13\n"
                     "v_alignbit_b32 v41, v23, v23, 13\n"
                     "v_alignbit_b32 v42, v26, v26, 13\n"
                     "v_alignbit_b32 v43, v29, v29, 13\n"
                     "v_alignbit_b32 v44, v32, v32, 13\n"
                     "v_alignbit_b32 v45, v35, v35, 13\n"
                     "v_alignbit_b32 v46, v22, v22, 13\n"
                     "v_alignbit_b32 v47, v25, v25, 13\n"
                     "v_alignbit_b32 v40, v28, v28, 13\n"
                     "v_alignbit_b32 v41, v31, v31, 13\n"
                     "v_alignbit_b32 v42, v34, v34, 13\n"
                     "v_alignbit_b32 v43, v21, v21, 13\n"
                     : "+v"(acc) : : CLOB, "v23", "v25", "v26", "v27", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37");
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

static void run_kernel_n(const char *name, void (*kern)(uint32_t *, int), int per_iter) {
    const int blocks = 256 * 8, iters = 128000 / per_iter;
    uint32_t *d;
    (void)hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    kern<<<blocks, 256>>>(d, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    kern<<<blocks, 256>>>(d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-56s %8.2f T lane-ops/s\n", name, (double)blocks * 256 * iters * per_iter / ms / 1e9);
    (void)hipFree(d);
}

static void run_kernel(const char *name, void (*kern)(uint32_t *, int)) {
    const int blocks = 256 * 8, iters = 4000;
    uint32_t *d;
    (void)hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    kern<<<blocks, 256>>>(d, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    kern<<<blocks, 256>>>(d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-56s %8.2f T lane-ops/s\n", name, (double)blocks * 256 * iters * 32 / ms / 1e9);
    (void)hipFree(d);
}

template <int V>
static void run(const char *name) {
    const int blocks = 256 * 8, iters = 4000;
    uint32_t *d;
    (void)hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    k<V><<<blocks, 256>>>(d, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<V><<<blocks, 256>>>(d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-56s %8.2f T lane-ops/s\n", name, (double)blocks * 256 * iters * 32 / ms / 1e9);
    (void)hipFree(d);
}

int main() {
    run<0>("bitop3 banks 0,1,2 (v20,v21,v22)");
    run<1>("bitop3 banks 0,0,0 (v20,v24,v28)");
    run<2>("bitop3 banks 0,0,1 (v20,v24,v21)");
    run<3>("bitop3 banks 0,1,0 (v20,v21,v24)");
    run<4>("bitop3 banks 1,0,0 (v21,v20,v24)");
    run<5>("bitop3 same register twice (v20,v20,v21)");
    run<6>("bitop3 same register x3 (v20,v20,v20)");
    run<7>("bitop3 0,1,2 dst bank = src bank pattern (dst v48..)");
    run<8>("alignbit banks 0,1 (v20,v21)");
    run<9>("alignbit banks 0,0 (v20,v24)");
    run<10>("alignbit same register (v20,v20)");
    run<11>("xor banks 0,1 (v20,v21)");
    run<12>("xor banks 0,0 (v20,v24)");
    {   // distinct (conflict-free) source registers in every instruction: no operand reuse between neighbours
        const int blocks = 256 * 8, iters = 4000;
        uint32_t *d;
        (void)hipMalloc(&d, blocks * 256 * 4);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        k_vary<<<blocks, 256>>>(d, 10);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        k_vary<<<blocks, 256>>>(d, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-56s %8.2f T lane-ops/s\n", "bitop3, different conflict-free sources each time", (double)blocks * 256 * iters * 32 / ms / 1e9);
    }
    run_kernel("v_lshlrev_b32 (shift by constant)", k_one0);
    run_kernel("v_lshrrev_b32 (shift by constant)", k_one1);
    run_kernel("v_lshl_or_b32", k_one2);
    run_kernel("v_lshl_add_u32", k_one3);
    run_kernel("v_and_or_b32", k_one4);
    run_kernel("v_or3_b32", k_one5);
    run_kernel("v_perm_b32", k_one6);
    run_kernel("v_alignbyte_b32", k_one7);
    run_kernel("v_bfe_u32", k_one8);
    run_kernel("v_bfi_b32", k_one9);
    run_kernel("v_mov_b32", k_one10);
    run_kernel("v_add_u32", k_one11);
    run_kernel("bitop3 reading the result of 1 instruction(s) back", k_dist1);
    run_kernel("bitop3 reading the result of 2 instruction(s) back", k_dist2);
    run_kernel("bitop3 reading the result of 3 instruction(s) back", k_dist3);
    run_kernel("bitop3 reading the result of 4 instruction(s) back", k_dist4);
    run_kernel("bitop3 reading the result of 6 instruction(s) back", k_dist6);
    run_kernel_n("runs of 2 bitop3 then 1 alignbit", k_runs2, 3);
    run_kernel_n("runs of 5 bitop3 then 2 alignbit", k_runs5, 7);
    run_kernel_n("runs of 12 bitop3 then 5 alignbit", k_runs12, 17);
    run_kernel_n("runs of 24 bitop3 then 10 alignbit", k_runs24, 34);
    run_kernel_n("runs of 48 bitop3 then 20 alignbit", k_runs48, 68);
    run_kernel_n("runs of 120 bitop3 then 52 alignbit", k_runs120, 172);
    for (int blocks : {256, 512, 1024, 2048}) {  // occupancy: 1, 2, 4, 8 waves per SIMD
        const int iters = 4000;
        uint32_t *d;
        (void)hipMalloc(&d, blocks * 256 * 4);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        for (int which = 0; which < 2; which++) {
            if (which == 0) k_vary<<<blocks, 256>>>(d, 10); else k_dep<<<blocks, 256>>>(d, 10);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            if (which == 0) k_vary<<<blocks, 256>>>(d, iters); else k_dep<<<blocks, 256>>>(d, iters);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("%d waves/SIMD, bitop3 %-34s %8.2f T lane-ops/s\n", blocks / 256, which == 0 ? "independent" : "reading results 8-11 instrs back", (double)blocks * 256 * iters * 32 / ms / 1e9);
        }
        (void)hipFree(d);
    }
    {   // two single-class streams sharing every SIMD
        const int blocks = 256 * 8, ib = 4000 * 59 / 35, ia = 4000;
        uint32_t *d;
        (void)hipMalloc(&d, blocks * 256 * 4);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        k_two_streams<<<blocks, 256>>>(d, 10, 10);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        k_two_streams<<<blocks, 256>>>(d, ib, ia);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double nb = (double)blocks / 2 * 256 * ib * 32, na = (double)blocks / 2 * 256 * ia * 32;
        printf("two streams per SIMD (bitop3 waves + alignbit waves): %.3f ms; time-sharing at 59 / 35 T/s would take %.3f ms\n",
               ms, (nb / 59e12 + na / 35e12) * 1e3);
    }
    {   // the Keccak mix (24 bitop3 : 10 alignbit), independent and conflict-free: what the issue logic allows at best
        const int blocks = 256 * 8, iters = 4000;
        uint32_t *d;
        (void)hipMalloc(&d, blocks * 256 * 4);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        k_mix<<<blocks, 256>>>(d, 10);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        k_mix<<<blocks, 256>>>(d, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double tops = (double)blocks * 256 * iters * 34 / ms / 1e9;
        printf("%-56s %8.2f T lane-ops/s  (= %.2f G Keccak-f/s at 4021 instructions)\n", "mix 24 bitop3 : 10 alignbit, interleaved", tops, tops * 1e3 / 4021);
    }
    return 0;
}
