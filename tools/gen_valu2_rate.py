#!/usr/bin/env python3
"""Generates tools/valu2_rate_kernels.inc / valu2_rate_calls.inc for tools/valu2_rate.hip: streams of 12 v_bitop3 +
5 v_alignbit runs that differ in ONE respect each -- register-set size (at 5 waves per SIMD), and, at 8 waves per SIMD,
whether the v_bitop3 read results of v_alignbit instructions and how long ago those were produced."""
import os
import random

random.seed(11)
HERE = os.path.dirname(os.path.abspath(__file__))
kernels, calls = [], []


def clob(lo, hi):
    return ", ".join('"v%d"' % r for r in range(lo, hi))


def three(pool):
    while True:
        r = random.sample(pool, 3)
        if len({x % 4 for x in r}) == 3:
            return r


def kernel(name, title, lines, clobbers):
    kernels.append("__global__ __launch_bounds__(256) void %s(uint32_t *out, int iters) {\n    uint32_t acc = threadIdx.x;\n"
                   "    for (int it = 0; it < iters; it++)\n        asm volatile(%s\n                     : \"+v\"(acc) : : %s);\n"
                   "    out[blockIdx.x * 256 + threadIdx.x] = acc;\n}\n" % (name, "\n                     ".join(lines), clobbers))
    calls.append('    run("%s", %s, %d);' % (title, name, len(lines)))


def emit_sets(name, title, src_pool, dst_pool):  # 100 VGPRs -> 5 waves per SIMD
    lines, k = [], 0
    for g in range(8):
        for r in range(12):
            d = dst_pool[k % len(dst_pool)]
            lines.append('"v_bitop3_b32 v%d, v%d, v%d, v%d bitop3:0x96\\n"' % (d, *three(src_pool))); k += 1
        for r in range(5):
            a = random.choice(src_pool)
            lines.append('"v_alignbit_b32 v%d, v%d, v%d, 13\\n"' % (dst_pool[k % len(dst_pool)], a, a)); k += 1
    kernel(name, title, lines, clob(20, 100))


D8 = list(range(92, 100))
emit_sets("k_s8", "5 waves/SIMD: sources from 8 registers", list(range(20, 28)), D8)
emit_sets("k_s64", "5 waves/SIMD: sources from 64 registers", list(range(20, 84)), D8)


def emit_flow(name, title, lag, b_reads_a, a_reads_b):
    """8 waves per SIMD (registers v20..v63).  Rotating result files: B results in v24..v43 (20), A results in v44..v63 (20).
    b_reads_a: every v_bitop3 takes one source from the A file, written `lag` A-instructions ago (else from constants).
    a_reads_b: every v_alignbit rotates a B result (else a constant)."""
    lines, nb, na = [], 0, 0
    for g in range(12):
        for r in range(12):
            d = 24 + nb % 20
            s0 = (44 + (na - 1 - lag) % 20) if b_reads_a else 20
            lines.append('"v_bitop3_b32 v%d, v%d, v21, v22 bitop3:0x96\\n"' % (d, s0)); nb += 1
        for r in range(5):
            d = 44 + na % 20
            s = (24 + (nb - 1 - r) % 20) if a_reads_b else 23
            lines.append('"v_alignbit_b32 v%d, v%d, v%d, 13\\n"' % (d, s, s)); na += 1
    kernel(name, title, lines, clob(20, 64))


emit_flow("k_f00", "8 waves/SIMD: no data flow between the classes", 0, False, False)
emit_flow("k_f01", "8 waves/SIMD: alignbit rotates bitop3 results", 0, False, True)
emit_flow("k_f10", "8 waves/SIMD: bitop3 reads the newest alignbit result", 0, True, False)
emit_flow("k_f10l5", "8 waves/SIMD: bitop3 reads an alignbit result 5 rotates old", 5, True, False)
emit_flow("k_f10l15", "8 waves/SIMD: bitop3 reads an alignbit result 15 rotates old", 15, True, False)
emit_flow("k_f11", "8 waves/SIMD: both directions (Keccak-like)", 0, True, True)
emit_flow("k_f11l15", "8 waves/SIMD: both directions, alignbit results 15 rotates old", 15, True, True)
def emit_reset(name, title, sep):
    """12 groups of 12 v_bitop3 + 5 v_alignbit per loop iteration, `sep` inserted after every v_alignbit run"""
    lines, nb, na = [], 0, 0
    for g in range(12):
        for r in range(12):
            lines.append('"v_bitop3_b32 v%d, v20, v21, v22 bitop3:0x96\\n"' % (24 + nb % 20)); nb += 1
        for r in range(5):
            lines.append('"v_alignbit_b32 v%d, v23, v23, 13\\n"' % (44 + na % 20)); na += 1
        if sep:
            lines.append('"%s\\n"' % sep.replace("@", str(g)))
    kernels.append("__global__ __launch_bounds__(256) void %s(uint32_t *out, int iters) {\n    uint32_t acc = threadIdx.x;\n"
                   "    for (int it = 0; it < iters; it++)\n        asm volatile(%s\n                     : \"+v\"(acc) : : %s);\n"
                   "    out[blockIdx.x * 256 + threadIdx.x] = acc;\n}\n" % (name, "\n                     ".join(lines), clob(20, 64)))
    calls.append('    run("%s", %s, %d);' % (title, name, 204))


emit_reset("k_r_none", "8 waves/SIMD: 12x(12 bitop3 + 5 alignbit), nothing in between", "")
emit_reset("k_r_nop", "  ... s_nop 0 after every alignbit run", "s_nop 0")
emit_reset("k_r_nop7", "  ... s_nop 7 after every alignbit run", "s_nop 7")
emit_reset("k_r_mov", "  ... s_mov_b32 s20, 0 after every alignbit run", "s_mov_b32 s20, 0")
emit_reset("k_r_br", "  ... s_branch to the next instruction after every alignbit run", "s_branch 0")
emit_reset("k_r_vnop", "  ... v_nop after every alignbit run", "v_nop")
emit_reset("k_r_wait", "  ... s_waitcnt vmcnt(0) after every alignbit run", "s_waitcnt vmcnt(0)")


def emit_len(n):
    lines = ['"v_bitop3_b32 v%d, v%d, v%d, v%d bitop3:0x96\\n"' % (40 + (i % 8), 20 + (i * 3) % 16, 21 + (i * 3) % 16, 22 + (i * 3) % 16)
             for i in range(n)]
    kernel("k_len%d" % n, "8 waves/SIMD: pure v_bitop3, %d instructions per loop iteration" % n, lines, clob(20, 64))


for n in (8, 16, 32, 64, 128, 256, 512, 1024, 2048):
    emit_len(n)
open(os.path.join(HERE, "valu2_rate_kernels.inc"), "w").write("\n".join(kernels))
open(os.path.join(HERE, "valu2_rate_calls.inc"), "w").write("\n".join(calls) + "\n")
