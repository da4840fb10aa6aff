// What decides whether a v_bitop3_b32 goes to the second VALU pipe (SQ_ACTIVE_INST_VALU2) on gfx950?  Streams of
// 12 v_bitop3 + 5 v_alignbit runs whose operands are drawn from register sets of different sizes / layouts, all
// kernels with the same VGPR allocation (same occupancy).  Run under
//   rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU2 --kernel-trace -- /tmp/valu2_rate
// and plain for the rates.  The kernels are generated: python3 tools/gen_valu2_rate.py writes valu2_rate_kernels.inc /
// valu2_rate_calls.inc next to this file (not committed).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "valu2_rate_kernels.inc"

static void run(const char *name, void (*kern)(uint32_t *, int), int per_iter) {
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
    printf("%-64s %8.2f T lane-ops/s\n", name, (double)blocks * 256 * iters * per_iter / ms / 1e9);
    (void)hipFree(d);
}

int main() {
#include "valu2_rate_calls.inc"
    return 0;
}
