#!/usr/bin/env python3
"""Host-only: time of a 2^20-record "LASSO_TABLE" absorption per transcript when J transcripts run at once through K sponge
servers (zigz_host_sponge_servers), against one transcript on its own thread.   python tools/sponge_service_rate.py"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zigz_amd.hip import Transcript  # noqa: E402
from zigz_amd._ffi import lib  # noqa: E402

L = 1 << 20


def run(jobs):
    def one():
        t = Transcript()
        t.append_tagged_counter(b"LASSO_TABLE", 0, L)
    th = [threading.Thread(target=one) for _ in range(jobs)]
    t0 = time.perf_counter()
    [t.start() for t in th]
    [t.join() for t in th]
    return time.perf_counter() - t0


print("own thread, 1 transcript: %.1f ms" % (min(run(1) for _ in range(3)) * 1e3))
for servers, jobs in ((1, 1), (1, 4), (1, 8), (2, 16)):
    lib.zigz_host_sponge_servers(servers)
    dt = min(run(jobs) for _ in range(3))
    lib.zigz_host_sponge_servers(0)
    print("%d server(s), %2d transcripts at once: %.1f ms each batch = %.3f us per block and slot-row" %
          (servers, jobs, dt * 1e3, dt / (L * 19 / 136) * 1e6))
