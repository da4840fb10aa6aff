#!/usr/bin/env python3
"""Generates tools/valu2_rate_kernels.inc / valu2_rate_calls.inc for tools/valu2_rate.hip."""
import os
import random

random.seed(11)
HERE = os.path.dirname(os.path.abspath(__file__))
ALL = list(range(20, 100))
CLOB = ", ".join('"v%d"' % r for r in ALL)
kernels, calls = [], []


def three(pool):
    while True:
        r = random.sample(pool, 3)
        if len({x % 4 for x in r}) == 3:
            return r


def emit(name, title, src_pool, dst_pool, groups=8, seq=False, same_within_run=False):
    lines, n, k = [], 0, 0
    for g in range(groups):
        fixed = three(src_pool)
        for r in range(12):
            if seq:
                a = src_pool[(k * 3) % (len(src_pool) - 2)]
                srcs = [a, a + 1, a + 2]
            elif same_within_run:
                srcs = fixed
            else:
                srcs = three(src_pool)
            d = dst_pool[k % len(dst_pool)]
            lines.append('"v_bitop3_b32 v%d, v%d, v%d, v%d bitop3:0x96\\n"' % (d, *srcs))
            n += 1; k += 1
        for r in range(5):
            a = random.choice(src_pool)
            d = dst_pool[k % len(dst_pool)]
            lines.append('"v_alignbit_b32 v%d, v%d, v%d, 13\\n"' % (d, a, a))
            n += 1; k += 1
    kernels.append("__global__ __launch_bounds__(256) void %s(uint32_t *out, int iters) {\n    uint32_t acc = threadIdx.x;\n"
                   "    for (int it = 0; it < iters; it++)\n        asm volatile(%s\n                     : \"+v\"(acc) : : %s);\n"
                   "    out[blockIdx.x * 256 + threadIdx.x] = acc;\n}\n" % (name, "\n                     ".join(lines), CLOB))
    calls.append('    run("%s", %s, %d);' % (title, name, n))


D8, D16 = list(range(92, 100)), list(range(84, 100))
emit("k_s8", "sources from 8 registers (v20-v27), random", list(range(20, 28)), D8)
emit("k_s16", "sources from 16 registers, random", list(range(20, 36)), D8)
emit("k_s32", "sources from 32 registers, random", list(range(20, 52)), D8)
emit("k_s64", "sources from 64 registers, random", list(range(20, 84)), D8)
emit("k_s64seq", "sources from 64 registers, consecutive triples walking up", list(range(20, 84)), D8, seq=True)
emit("k_s64run", "sources from 64 registers, one triple per run of 12", list(range(20, 84)), D8, same_within_run=True)
emit("k_s16d16", "sources from 16 registers, 16 destinations", list(range(20, 36)), D16)
emit("k_s64d16", "sources from 64 registers, 16 destinations", list(range(20, 84)), D16)
open(os.path.join(HERE, "valu2_rate_kernels.inc"), "w").write("\n".join(kernels))
open(os.path.join(HERE, "valu2_rate_calls.inc"), "w").write("\n".join(calls) + "\n")
