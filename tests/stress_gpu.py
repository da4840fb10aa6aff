#!/usr/bin/env python3
"""One-off GPU soak (not collected by pytest): random LOOPING RV64IM programs with ragged step counts up to 2^17, proved
(every fifth case: a straight-line program that never loops) through every build variant of the HIP path -- packed rows /
compact device trace; Merkle build dense, with the small-domain tables, with run-aware register columns, with every column
run-aware, with the content-addressed group -- ALL ON ONE CONTEXT, so the room it has learnt for its lists, its repeated builds
and its dropped groups carry over from case to case -- and compared byte for byte with the oracle's literal proof.

    python tests/stress_gpu.py [--cases 12] [--seed 7] [--max-log 17]

The pytest suite covers the same paths at fixed sizes; this widens the input space (loop bodies drawn from
tests/programs.random_program's instruction mix, so columns have runs of every length, loads/stores, wrapped values).
Prints one line per case and a final "OK n cases".
"""
import argparse
import os
import struct
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402
import programs  # noqa: E402


def looping_program(rng, target_steps):
    """prologue; x8 = 0; x9 = K; loop: <random body over x1..x7>; x8 += 1; BNE x8, x9, loop; EBREAK"""
    body = programs.random_program(rng, n_insts=int(rng.integers(1, 40)))
    words = list(struct.unpack("<%dI" % (len(body) // 4), body))
    # random_program starts with an x1..x7 initialisation block: keep it as the prologue, loop over the rest
    n_init = 0
    while n_init < len(words) and (words[n_init] & 0x7f) in (0x37, 0x13) and n_init < 21:
        n_init += 1
    n_init = min(n_init, len(words) - 1)
    pro, loop = words[:n_init], words[n_init:]
    per_iter = len(loop) + 2
    k = max(1, (target_steps - len(pro) - 4) // per_iter)
    out = list(pro)
    out.append(programs._I(0x13, 8, 0, 0, 0))
    out += programs._li(9, k) if k >= 2048 else [programs._I(0x13, 9, 0, 0, k)]
    out += loop
    out.append(programs._I(0x13, 8, 0, 8, 1))
    out.append(programs._B(1, 8, 9, -4 * (len(loop) + 1)))
    out.append(0x00100073)
    return programs._pack(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=12)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--max-log", type=int, default=17)
    args = ap.parse_args()
    import zigz_amd
    from zigz_amd import host
    ctx = zigz_amd.Context(0)
    rng = np.random.default_rng(args.seed)
    P = O.P_BB
    for case in range(args.cases):
        lg = int(rng.integers(8, args.max_log + 1)) if case % 3 else args.max_log
        target = int(rng.integers((1 << (lg - 1)) + 1, (1 << lg) + 1))
        # every fifth case never loops (the content-addressed group is dropped on the device, its columns rebuilt with slabs,
        # and after two such jobs the context stops trying the group for a while): the same context then meets loops again
        prog = programs.straight_line_program(int(rng.integers(1 << 30)), target) if case % 5 == 4 else looping_program(rng, target)
        iregs = None if case % 2 else [0] + [int(x) for x in rng.integers(0, 2**63, size=int(rng.integers(1, 31)), dtype=np.int64)]
        t0 = time.time()
        oproof, ons = O.prove(P, prog, 0x1000, iregs, 1 << 20)
        t_or = time.time() - t0
        variants = 0
        for mode in ("dense", "off", "regs", "struct", "all", "cons"):  # Merkle build: dense / small-domain tables / + run-aware registers
            # / + the memory columns / + every column / struct + the content-addressed instruction group (the default)
            os.environ.pop("ZIGZ_DENSE_MERKLE", None)
            os.environ.pop("ZIGZ_RUN_AWARE", None)
            if mode == "dense":
                os.environ["ZIGZ_DENSE_MERKLE"] = "1"
            else:
                os.environ["ZIGZ_RUN_AWARE"] = mode
            proof, ns = host.prove(ctx, prog, 0x1000, iregs, 1 << 20)
            assert ns == ons and proof == oproof, ("rows", case, mode)
            tr = host.Trace(prog, 0x1000, iregs, 1 << 20)
            n = 1 << tr.num_vars
            d = ctx.dev_alloc(43 * max(n, 4) * 4)
            try:
                tr.witness_to_device(ctx, d, max(n, 4))
                assert tr.prove(ctx, d, max(n, 4)) == oproof, ("compact", case, mode)
            finally:
                ctx.dev_free(d)
            variants += 2
        os.environ.pop("ZIGZ_DENSE_MERKLE", None)
        os.environ.pop("ZIGZ_RUN_AWARE", None)
        assert host.verify(oproof, prog) == "Accept"
        print("case %2d  steps %7d  nv %2d  proof %8d B  oracle %.1f s  %d variants identical" %
              (case, ons, (ons - 1).bit_length() if ons > 1 else 0, len(oproof), t_or, variants), flush=True)
    ctx.close()
    print("OK %d cases" % args.cases)


if __name__ == "__main__":
    main()
