#!/usr/bin/env python3
"""Keccak-f[800]-shaped round (25 lanes of 32 bits: theta / rho / pi / chi with the f[1600] offsets mod 32) as a probe:
real Keccak data flow that fits 8 waves per SIMD (about 40 VGPRs), emitted (a) in plain phase order for the compiler to
schedule and (b) list-scheduled into runs of RUN_B v_bitop3 : RUN_A v_alignbit fenced by scheduling barriers.
Only the instruction mix and the dependency structure matter here; the output is not a standard hash."""
import sys

RUN_B, RUN_A = 12, 5
RHO = {}
x, y = 1, 0
for t in range(24):
    RHO[(x, y)] = (((t + 1) * (t + 2) // 2) % 64) % 32
    x, y = y, (2 * x + 3 * y) % 5
RHO[(0, 0)] = 0
ops = []


def op(name, cls, deps, stmt):
    ops.append((name, cls, tuple(deps), stmt))


for x in range(5):
    op(f"p{x}", "b", [], f"const uint32_t p{x} = ZK_X3(a[{x}], a[{x + 5}], a[{x + 10}]);")
    op(f"c{x}", "b", [f"p{x}"], f"const uint32_t c{x} = ZK_X3(p{x}, a[{x + 15}], a[{x + 20}]);")
for x in range(5):
    op(f"r{x}", "a", [f"c{x}"], f"const uint32_t r{x} = ZK_ROT32(c{x}, 1);")
lanes = sorted(((x, y) for y in range(5) for x in range(5)), key=lambda l: ((2 * l[0] + 3 * l[1]) % 5, l[1]))
for (x, y) in lanes:
    i = x + 5 * y
    xm, xp = (x + 4) % 5, (x + 1) % 5
    op(f"t{i}", "b", [f"c{xm}", f"r{xp}"], f"const uint32_t t{i} = ZK_X3(a[{i}], c{xm}, r{xp});")
    dst = y + 5 * ((2 * x + 3 * y) % 5)
    k = RHO[(x, y)]
    if k == 0:
        op(f"b{dst}", None, [f"t{i}"], f"const uint32_t b{dst} = t{i};")
    else:
        op(f"b{dst}", "a", [f"t{i}"], f"const uint32_t b{dst} = ZK_ROT32(t{i}, {k});")
for row in range(5):
    for xx in range(5):
        i = 5 * row + xx
        op(f"n{i}", "b", [f"b{i}", f"b{5 * row + (xx + 1) % 5}", f"b{5 * row + (xx + 2) % 5}"],
           f"n[{i}] = ZK_CHI(b{i}, b{5 * row + (xx + 1) % 5}, b{5 * row + (xx + 2) % 5});")


def plain():
    return "\n".join("        " + o[3] for o in ops)


def scheduled():
    done, out = set(), []
    pending = list(ops)

    def flush():
        moved = True
        while moved:
            moved = False
            for o in list(pending):
                if o[1] is None and all(d in done for d in o[2]):
                    pending.remove(o); done.add(o[0]); out.append("        " + o[3]); moved = True

    want = "b"
    while any(o[1] is not None for o in pending):
        flush()
        quota = RUN_B if want == "b" else RUN_A
        batch = [o for o in pending if o[1] == want and all(d in done for d in o[2])][:quota]
        if not batch:
            want = "a" if want == "b" else "b"
            batch = [o for o in pending if o[1] == want and all(d in done for d in o[2])][:RUN_B if want == "b" else RUN_A]
        for o in batch:
            pending.remove(o); out.append("        " + o[3])
        for o in batch:
            done.add(o[0])
        out.append("        ZK_SCHED();")
        want = "a" if want == "b" else "b"
    flush()
    return "\n".join(out)


print(scheduled() if "--scheduled" in sys.argv else plain())
