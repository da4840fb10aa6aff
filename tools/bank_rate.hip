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

// runs of 12 bitop3 + 5 alignbit with operands spread over 64 source and 16 destination registers (as in the real kernel)
__global__ __launch_bounds__(256) void k_spread(uint32_t *out, int iters) {
    uint32_t acc = threadIdx.x;
    for (int it = 0; it < iters; it++)
        asm volatile("v_bitop3_b32 v85, v61, v39, v70 bitop3:0x96\n"
                     "v_bitop3_b32 v85, v29, v32, v66 bitop3:0x96\n"
                     "v_bitop3_b32 v91, v75, v73, v28 bitop3:0x96\n"
                     "v_bitop3_b32 v93, v73, v38, v35 bitop3:0x96\n"
                     "v_bitop3_b32 v95, v43, v33, v44 bitop3:0x96\n"
                     "v_bitop3_b32 v95, v60, v79, v78 bitop3:0x96\n"
                     "v_bitop3_b32 v97, v56, v29, v35 bitop3:0x96\n"
                     "v_bitop3_b32 v95, v29, v60, v63 bitop3:0x96\n"
                     "v_bitop3_b32 v86, v83, v78, v28 bitop3:0x96\n"
                     "v_bitop3_b32 v89, v22, v79, v65 bitop3:0x96\n"
                     "v_bitop3_b32 v93, v82, v43, v53 bitop3:0x96\n"
                     "v_bitop3_b32 v95, v20, v38, v73 bitop3:0x96\n"
                     "v_alignbit_b32 v88, v60, v60, 13\n"
                     "v_alignbit_b32 v98, v26, v26, 13\n"
                     "v_alignbit_b32 v96, v70, v70, 13\n"
                     "v_alignbit_b32 v96, v71, v71, 13\n"
                     "v_alignbit_b32 v99, v33, v33, 13\n"
                     "v_bitop3_b32 v85, v40, v34, v63 bitop3:0x96\n"
                     "v_bitop3_b32 v87, v33, v20, v39 bitop3:0x96\n"
                     "v_bitop3_b32 v90, v66, v23, v29 bitop3:0x96\n"
                     "v_bitop3_b32 v95, v58, v31, v53 bitop3:0x96\n"
                     "v_bitop3_b32 v90, v50, v71, v49 bitop3:0x96\n"
                     "v_bitop3_b32 v86, v77, v64, v66 bitop3:0x96\n"
                     "v_bitop3_b32 v90, v80, v45, v63 bitop3:0x96\n"
                     "v_bitop3_b32 v90, v30, v35, v69 bitop3:0x96\n"
                     "v_bitop3_b32 v94, v81, v42, v75 bitop3:0x96\n"
                     "v_bitop3_b32 v89, v71, v30, v40 bitop3:0x96\n"
                     "v_bitop3_b32 v95, v79, v38, v80 bitop3:0x96\n"
                     "v_bitop3_b32 v84, v39, v36, v22 bitop3:0x96\n"
                     "v_alignbit_b32 v88, v33, v33, 13\n"
                     "v_alignbit_b32 v90, v75, v75, 13\n"
                     "v_alignbit_b32 v84, v47, v47, 13\n"
                     "v_alignbit_b32 v90, v52, v52, 13\n"
                     "v_alignbit_b32 v91, v57, v57, 13\n"
                     "v_bitop3_b32 v98, v36, v27, v65 bitop3:0x96\n"
                     "v_bitop3_b32 v84, v73, v36, v39 bitop3:0x96\n"
                     "v_bitop3_b32 v98, v55, v25, v32 bitop3:0x96\n"
                     "v_bitop3_b32 v87, v29, v47, v58 bitop3:0x96\n"
                     "v_bitop3_b32 v91, v52, v37, v79 bitop3:0x96\n"
                     "v_bitop3_b32 v84, v63, v78, v76 bitop3:0x96\n"
                     "v_bitop3_b32 v87, v28, v34, v49 bitop3:0x96\n"
                     "v_bitop3_b32 v88, v25, v43, v54 bitop3:0x96\n"
                     "v_bitop3_b32 v88, v74, v53, v71 bitop3:0x96\n"
                     "v_bitop3_b32 v86, v22, v31, v53 bitop3:0x96\n"
                     "v_bitop3_b32 v94, v35, v78, v21 bitop3:0x96\n"
                     "v_bitop3_b32 v85, v73, v54, v36 bitop3:0x96\n"
                     "v_alignbit_b32 v87, v50, v50, 13\n"
                     "v_alignbit_b32 v92, v40, v40, 13\n"
                     "v_alignbit_b32 v89, v26, v26, 13\n"
                     "v_alignbit_b32 v93, v45, v45, 13\n"
                     "v_alignbit_b32 v90, v59, v59, 13\n"
                     "v_bitop3_b32 v88, v71, v64, v26 bitop3:0x96\n"
                     "v_bitop3_b32 v92, v78, v43, v40 bitop3:0x96\n"
                     "v_bitop3_b32 v90, v30, v80, v55 bitop3:0x96\n"
                     "v_bitop3_b32 v96, v53, v31, v38 bitop3:0x96\n"
                     "v_bitop3_b32 v88, v30, v23, v25 bitop3:0x96\n"
                     "v_bitop3_b32 v98, v66, v33, v68 bitop3:0x96\n"
                     "v_bitop3_b32 v98, v82, v53, v20 bitop3:0x96\n"
                     "v_bitop3_b32 v86, v78, v83, v68 bitop3:0x96\n"
                     "v_bitop3_b32 v93, v32, v47, v82 bitop3:0x96\n"
                     "v_bitop3_b32 v99, v45, v59, v30 bitop3:0x96\n"
                     "v_bitop3_b32 v95, v31, v38, v53 bitop3:0x96\n"
                     "v_bitop3_b32 v95, v36, v55, v34 bitop3:0x96\n"
                     "v_alignbit_b32 v99, v49, v49, 13\n"
                     "v_alignbit_b32 v96, v82, v82, 13\n"
                     "v_alignbit_b32 v89, v23, v23, 13\n"
                     "v_alignbit_b32 v99, v20, v20, 13\n"
                     "v_alignbit_b32 v96, v77, v77, 13\n"
                     "v_bitop3_b32 v94, v35, v62, v20 bitop3:0x96\n"
                     "v_bitop3_b32 v85, v26, v55, v33 bitop3:0x96\n"
                     "v_bitop3_b32 v90, v54, v75, v60 bitop3:0x96\n"
                     "v_bitop3_b32 v88, v26, v72, v77 bitop3:0x96\n"
                     "v_bitop3_b32 v93, v73, v63, v56 bitop3:0x96\n"
                     "v_bitop3_b32 v91, v52, v53, v71 bitop3:0x96\n"
                     "v_bitop3_b32 v86, v35, v41, v40 bitop3:0x96\n"
                     "v_bitop3_b32 v98, v46, v83, v48 bitop3:0x96\n"
                     "v_bitop3_b32 v86, v37, v44, v51 bitop3:0x96\n"
                     "v_bitop3_b32 v92, v60, v50, v67 bitop3:0x96\n"
                     "v_bitop3_b32 v96, v45, v22, v72 bitop3:0x96\n"
                     "v_bitop3_b32 v96, v54, v51, v69 bitop3:0x96\n"
                     "v_alignbit_b32 v97, v77, v77, 13\n"
                     "v_alignbit_b32 v84, v59, v59, 13\n"
                     "v_alignbit_b32 v85, v36, v36, 13\n"
                     "v_alignbit_b32 v99, v74, v74, 13\n"
                     "v_alignbit_b32 v84, v82, v82, 13\n"
                     "v_bitop3_b32 v98, v29, v70, v79 bitop3:0x96\n"
                     "v_bitop3_b32 v88, v51, v33, v48 bitop3:0x96\n"
                     "v_bitop3_b32 v86, v39, v33, v78 bitop3:0x96\n"
                     "v_bitop3_b32 v88, v49, v24, v58 bitop3:0x96\n"
                     "v_bitop3_b32 v87, v52, v75, v34 bitop3:0x96\n"
                     "v_bitop3_b32 v96, v29, v58, v44 bitop3:0x96\n"
                     "v_bitop3_b32 v84, v80, v50, v51 bitop3:0x96\n"
                     "v_bitop3_b32 v97, v22, v44, v83 bitop3:0x96\n"
                     "v_bitop3_b32 v97, v30, v52, v49 bitop3:0x96\n"
                     "v_bitop3_b32 v95, v24, v63, v73 bitop3:0x96\n"
                     "v_bitop3_b32 v93, v70, v45, v20 bitop3:0x96\n"
                     "v_bitop3_b32 v90, v28, v46, v83 bitop3:0x96\n"
                     "v_alignbit_b32 v90, v59, v59, 13\n"
                     "v_alignbit_b32 v98, v49, v49, 13\n"
                     "v_alignbit_b32 v92, v48, v48, 13\n"
                     "v_alignbit_b32 v87, v57, v57, 13\n"
                     "v_alignbit_b32 v89, v83, v83, 13\n"
                     "v_bitop3_b32 v85, v48, v82, v73 bitop3:0x96\n"
                     "v_bitop3_b32 v89, v73, v26, v27 bitop3:0x96\n"
                     "v_bitop3_b32 v87, v70, v77, v60 bitop3:0x96\n"
                     "v_bitop3_b32 v89, v67, v62, v76 bitop3:0x96\n"
                     "v_bitop3_b32 v92, v33, v20, v30 bitop3:0x96\n"
                     "v_bitop3_b32 v87, v30, v64, v73 bitop3:0x96\n"
                     "v_bitop3_b32 v93, v46, v68, v65 bitop3:0x96\n"
                     "v_bitop3_b32 v98, v80, v45, v67 bitop3:0x96\n"
                     "v_bitop3_b32 v99, v44, v61, v66 bitop3:0x96\n"
                     "v_bitop3_b32 v85, v71, v25, v68 bitop3:0x96\n"
                     "v_bitop3_b32 v84, v60, v55, v58 bitop3:0x96\n"
                     "v_bitop3_b32 v87, v28, v23, v49 bitop3:0x96\n"
                     "v_alignbit_b32 v98, v80, v80, 13\n"
                     "v_alignbit_b32 v92, v69, v69, 13\n"
                     "v_alignbit_b32 v99, v75, v75, 13\n"
                     "v_alignbit_b32 v99, v36, v36, 13\n"
                     "v_alignbit_b32 v84, v43, v43, 13\n"
                     "v_bitop3_b32 v95, v61, v60, v78 bitop3:0x96\n"
                     "v_bitop3_b32 v87, v61, v40, v74 bitop3:0x96\n"
                     "v_bitop3_b32 v99, v46, v32, v73 bitop3:0x96\n"
                     "v_bitop3_b32 v93, v50, v35, v57 bitop3:0x96\n"
                     "v_bitop3_b32 v98, v51, v49, v32 bitop3:0x96\n"
                     "v_bitop3_b32 v90, v49, v35, v26 bitop3:0x96\n"
                     "v_bitop3_b32 v89, v44, v29, v67 bitop3:0x96\n"
                     "v_bitop3_b32 v85, v33, v64, v47 bitop3:0x96\n"
                     "v_bitop3_b32 v85, v25, v46, v52 bitop3:0x96\n"
                     "v_bitop3_b32 v85, v59, v29, v46 bitop3:0x96\n"
                     "v_bitop3_b32 v97, v83, v81, v28 bitop3:0x96\n"
                     "v_bitop3_b32 v86, v32, v70, v39 bitop3:0x96\n"
                     "v_alignbit_b32 v96, v40, v40, 13\n"
                     "v_alignbit_b32 v97, v54, v54, 13\n"
                     "v_alignbit_b32 v93, v56, v56, 13\n"
                     "v_alignbit_b32 v85, v73, v73, 13\n"
                     "v_alignbit_b32 v95, v59, v59, 13\n"
                     : "+v"(acc) : : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99");
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_aonly(uint32_t *out, int iters) {
    uint32_t acc = threadIdx.x;
    for (int it = 0; it < iters; it++)
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
    run_kernel_n("runs of 12 bitop3 + 5 alignbit, 80 registers in use", k_spread, 136);
    for (int w : {1, 2, 3, 4, 5, 6, 7, 8}) {  // resident waves per SIMD (256-thread workgroups: one wave per SIMD each)
        const int blocks = 256 * w;
        uint32_t *d;
        (void)hipMalloc(&d, blocks * 256 * 4);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        double rate[3];
        for (int which = 0; which < 3; which++) {
            const int per = which == 2 ? 17 : 32, iters = 128000 / per;
            auto launch = [&](int it) {
                if (which == 0) k_vary<<<blocks, 256>>>(d, it);
                else if (which == 1) k_aonly<<<blocks, 256>>>(d, it);
                else k_runs12<<<blocks, 256>>>(d, it);
            };
            launch(10);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            launch(iters);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            rate[which] = (double)blocks * 256 * iters * per / ms / 1e9;
        }
        printf("%d waves/SIMD: bitop3 %6.2f   alignbit %6.2f   runs of 12 bitop3 + 5 alignbit %6.2f  T lane-ops/s\n", w, rate[0], rate[1], rate[2]);
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
