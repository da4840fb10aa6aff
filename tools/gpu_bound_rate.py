#!/usr/bin/env python3
"""GPU-bound rate of the commit path: L lanes (thread + context + HIP stream each) run generateCommitments' device work
-- Merkle build of the 43 resident columns of the bench trace, roots, 43 evals + openings -- back to back WITHOUT the host
transcript that bounds a real proof.  What the GPU could take if the host sponge were free.

    python tools/gpu_bound_rate.py [--lanes 14] [--iters 20] [--trace add_xor|round_robin]
"""
import argparse
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import programs  # noqa: E402
import zigz_amd  # noqa: E402
from zigz_amd import host  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lanes", type=int, default=14)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--trace", default="add_xor", help="add_xor | round_robin | mixed (RV64IM mix with loads / stores, BASELINE config 4's loop)")
ap.add_argument("--hint", default="cons", help="run-aware hint: regs | regs+mem | all | cons (regs+mem and the ten "
                "instruction-determined columns as a content-addressed group)")
ap.add_argument("--debug-skip", type=int, default=0, help="measurement only (wrong trees): 1 = no level hashing / top kernel, "
                "2 = no structure passes (the hash launches then find the previous job's lists)")
ap.add_argument("--phases", action="store_true", help="also print the host wall time of each call of the commit job")
ap.add_argument("--blocking-sync", action="store_true", help="waiting host threads sleep instead of spinning")
ap.add_argument("--batch", type=int, default=1, help="proofs per commit job (zigz_commit_begin_batch: the arena form at 2^20)")
args = ap.parse_args()
if args.blocking_sync:
    zigz_amd._ffi.lib.zigz_device_set_blocking_sync(0, 1)
nv = 20
N = 1 << nv
REGS = 0x7fffffff << 2
SMALL = (1 << 1) | (0x3f << 33) | (1 << 42)
rng = np.random.default_rng(1)


STRAIGHT = None
if args.trace == "straight":  # no loop at all: ~2^20 different instructions, executed once each (the worst case for a
    STRAIGHT = programs.random_program(np.random.default_rng(3), n_insts=int(0.83 * N))  # content-addressed group)


class Lane:
    def __init__(self, k):
        self.ctx = zigz_amd.Context(0)
        prog = (STRAIGHT if args.trace == "straight" else
                programs.add_xor_loop((N - 3) // 4 - k) if args.trace == "add_xor" else
                programs.mixed_loop((N - 8) // 12 - k) if args.trace == "mixed" else
                programs.register_round_robin((N - 2) // 31 - k))
        self.tr = host.Trace(prog, 0x1000, None, 2 * N)
        self.d = self.ctx.dev_alloc(43 * N * 4)
        self.tr.witness_to_device(self.ctx, self.d, N)
        self.ds = [self.d]
        for j in range(1, args.batch):  # (the same witness again: what a job costs does not depend on whose columns they are)
            d2 = self.ctx.dev_alloc(43 * N * 4)
            self.tr.witness_to_device(self.ctx, d2, N)
            self.ds.append(d2)
        self.points = rng.integers(0, 2013265921, size=(43 * args.batch, nv), dtype=np.uint64)
        self.ctx.set_option("small_domain_mask", SMALL)
        self.ctx.set_option("run_aware_mask", {"regs": REGS, "regs+mem": REGS | (3 << 40), "all": ((1 << 43) - 1) & ~SMALL,
                                               "cons": REGS | (3 << 40)}[args.hint])
        if args.hint == "cons":
            self.ctx.set_option("cons_group_mask", 1 | (1 << 1) | (0x7f << 33) | (1 << 42))
        self.debug_skip = args.debug_skip

    def once(self):
        t0 = time.perf_counter()
        if args.batch > 1:
            job = zigz_amd.CommitJob(self.ctx, d_cols_list=self.ds, ncols=43, nv=nv, col_stride=N)
        else:
            job = zigz_amd.CommitJob(self.ctx, d_cols=self.d, ncols=43, nv=nv, col_stride=N)
        t1 = time.perf_counter()
        job.roots()
        t2 = time.perf_counter()
        job.open_all(self.points)
        t3 = time.perf_counter()
        job.end()
        t4 = time.perf_counter()
        self.t = [a + b for a, b in zip(getattr(self, "t", [0, 0, 0, 0]), (t1 - t0, t2 - t1, t3 - t2, t4 - t3))]


lanes = [Lane(k) for k in range(args.lanes)]
for l in lanes:
    l.once()
    l.once()
    if l.debug_skip:
        l.ctx.set_option("debug_skip", l.debug_skip)


def loop(l):
    for _ in range(args.iters):
        l.once()


t0 = time.perf_counter()
th = [threading.Thread(target=loop, args=(l,)) for l in lanes]
[t.start() for t in th]
[t.join() for t in th]
dt = time.perf_counter() - t0
n = args.lanes * args.iters * args.batch
if args.phases:
    tot = [sum(l.t[i] for l in lanes) / (args.lanes * (args.iters + 1)) * 1e3 for i in range(4)]
    print("host wall per proof and lane: begin %.3f  roots %.3f  open_all %.3f  end %.3f ms" % tuple(tot))
print("%s (hint %s): %d lanes x %d proofs per job: %.3f ms per proof's GPU work = %.1f M steps/s if nothing else bounded it" %
      (args.trace, args.hint, args.lanes, args.batch, dt / n * 1e3, n * lanes[0].tr.num_steps / dt / 1e6))
