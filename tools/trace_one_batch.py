#!/usr/bin/env python3
"""Lone batched commit jobs (K proofs of the 2^20 bench trace in one zigz_commit_begin_batch) beside lone single jobs, for a
per-launch view:  rocprofv3 --kernel-trace --output-format csv -d <dir> -o t -- python3 tools/trace_one_batch.py [K]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import zigz_amd, programs
from zigz_amd import host
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
nv = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctx = zigz_amd.Context(0)
N = 1 << nv
bufs = []
for i in range(K):
    prog = programs.add_xor_loop((N - 3) // 4 - i)
    tr = host.Trace(prog, 0x1000, None, 2 * N)
    d = ctx.dev_alloc(43 * N * 4)
    tr.witness_to_device(ctx, d, N)
    bufs.append(d)
small = (1 << 1) | (0x3f << 33) | (1 << 42)
for k, v in {"small_domain_mask": small, "run_aware_mask": (0x7fffffff << 2) | (3 << 40), "cons_group_mask": 1 | (1 << 1) | (0x7f << 33) | (1 << 42)}.items():
    ctx.set_option(k, v)
pts = np.random.default_rng(1).integers(0, 2013265921, size=(K * 43, nv), dtype=np.uint64)
import time
for rep in range(3):
    t0 = time.perf_counter()
    job = zigz_amd.CommitJob(ctx, d_cols=bufs[0], ncols=43, nv=nv, col_stride=N)
    job.roots(); t1 = time.perf_counter(); job.open_all(pts[:43]); job.end()
    t2 = time.perf_counter()
    sys.stderr.write("single: roots %.3f ms, open_all %.3f ms\n" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
for rep in range(3):
    t0 = time.perf_counter()
    job = zigz_amd.CommitJob(ctx, d_cols_list=bufs, ncols=43, nv=nv, col_stride=N)
    job.roots(); t1 = time.perf_counter(); job.open_all(pts); job.end()
    t2 = time.perf_counter()
    sys.stderr.write("batch of %d: roots %.3f ms, open_all %.3f ms\n" % (K, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
ctx.synchronize()
