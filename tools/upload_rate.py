#!/usr/bin/env python3
"""Aggregate rate of the trace upload + witness build alone: L lanes (thread + context + stream each) push their pinned
2^20-step compact trace (50 MB) and expand it to the 43 columns, K times each.  Tells the link / copy-engine limit apart from
everything else in bench.py's PCIe-inclusive leg.   python tools/upload_rate.py [--lanes 16] [--iters 20]"""
import argparse
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import programs  # noqa: E402
import zigz_amd  # noqa: E402
from zigz_amd import host  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lanes", type=int, default=16)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--blocking", type=int, default=0)
args = ap.parse_args()
N = 1 << 20
if args.blocking:
    zigz_amd._ffi.lib.zigz_device_set_blocking_sync(0, 1)


class Lane:
    def __init__(self, k):
        self.ctx = zigz_amd.Context(0)
        self.tr = host.Trace(programs.add_xor_loop((N - 3) // 4 - k), 0x1000, None, 2 * N)
        self.d = self.ctx.dev_alloc(43 * N * 4)
        self.tr.pin(self.ctx)
        self.tr.witness_to_device(self.ctx, self.d, N)
        self.ctx.synchronize()

    def loop(self):
        for _ in range(args.iters):
            self.tr.witness_to_device(self.ctx, self.d, N, wait=False)
            self.ctx.synchronize()


lanes = [Lane(k) for k in range(args.lanes)]
t0 = time.perf_counter()
th = [threading.Thread(target=l.loop) for l in lanes]
[t.start() for t in th]
[t.join() for t in th]
dt = time.perf_counter() - t0
n = args.lanes * args.iters
print("%d lanes: %.3f ms per upload + witness build = %.1f GB/s of trace records = %.0f M steps/s" %
      (args.lanes, dt / n * 1e3, n * lanes[0].tr.num_steps * 48 / dt / 1e9, n * lanes[0].tr.num_steps / dt / 1e6))
