#!/usr/bin/env python3
"""Secondary measurements quoted in DESIGN.md (not the bench contract): device-resident sumcheck latency,
Prover.prove end to end from program bytes (VM + witness build + hot path + serialisation), PCIe-inclusive
commit from host columns.  Run on the GPU box:  python tools/measure_extra.py"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import zigz_amd  # noqa: E402
from zigz_amd import host  # noqa: E402
import programs  # noqa: E402

P = 2013265921


def best(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts), sorted(ts)[len(ts) // 2]


def main():
    out = {}
    ctx = zigz_amd.Context(0)
    rng = np.random.default_rng(1)
    # (1) device-resident sumcheck (A5): one table of 2^nv elements already in HBM
    for nv in (12, 16, 20, 22, 24):
        n = 1 << nv
        ev = rng.integers(0, P, n, dtype=np.uint64)
        d = ctx.dev_alloc(n * 4)
        ctx.upload(ev, d)
        ctx.enable_timing(False)
        tmin, tmed = best(lambda: ctx.dev_sumcheck_prove(d, n))
        ctx.set_option("per_round_sumcheck", 1)
        tmin_pr, _ = best(lambda: ctx.dev_sumcheck_prove(d, n))
        ctx.set_option("per_round_sumcheck", 0)
        ctx.enable_timing(True)
        ctx.dev_sumcheck_prove(d, n)
        st = ctx.stats()
        ctx.enable_timing(False)
        out["sumcheck_2^%d" % nv] = dict(ms_min=tmin * 1e3, ms_median=tmed * 1e3, ms_min_per_round_form=tmin_pr * 1e3, us_per_round=tmin * 1e6 / nv,
                                         algorithmic_MB=16 * n / 1e6, hbm_GBs_overall=16 * n / tmin / 1e9,
                                         bind_vec_GBs=(st["bind_vec_bytes"] / 1e9) / (st["bind_vec_us"] / 1e6) if st["bind_vec_us"] else None,
                                         bind_vec_launches=st["bind_vec_launches"])
        ctx.dev_free(d)
    if "--sumcheck-only" in sys.argv:
        print(json.dumps(out, indent=1))
        ctx.close()
        return
    # (2) Prover.prove end to end from program bytes at a 2^20 trace (VM + compact-trace H2D + K8 + hot path + serialise)
    for nv in (16, 20):
        prog = programs.add_xor_loop(((1 << nv) - 3) // 4)
        tmin, tmed = best(lambda: host.prove(ctx, prog, 0x1000, None, 1 << (nv + 1)), reps=3)
        t0 = time.perf_counter(); tr = host.Trace(prog, 0x1000, None, 1 << (nv + 1)); t_vm = time.perf_counter() - t0
        out["prove_from_program_2^%d" % nv] = dict(ms_min=tmin * 1e3, vm_ms=t_vm * 1e3, steps_per_s=tr.num_steps / tmin)
        # (3) PCIe-inclusive: witness columns on the host (canonical u64), uploaded inside the call
        tmin2, _ = best(lambda: tr.prove(ctx, want_bytes="borrow"), reps=3)
        out["prove_trace_host_witness_2^%d" % nv] = dict(ms_min=tmin2 * 1e3, steps_per_s=tr.num_steps / tmin2)
        N = 1 << nv
        d = ctx.dev_alloc(43 * N * 4)
        tr.witness_to_device(ctx, d, N)
        tmin3, _ = best(lambda: tr.prove(ctx, d, N, want_bytes="borrow"), reps=5)
        out["prove_trace_resident_2^%d" % nv] = dict(ms_min=tmin3 * 1e3, steps_per_s=tr.num_steps / tmin3)
        t0 = time.perf_counter(); tr.witness_to_device(ctx, d, N); out["witness_steps_to_device_pageable_2^%d_ms" % nv] = (time.perf_counter() - t0) * 1e3
        tr.pin(ctx)
        tr.witness_to_device(ctx, d, N)
        t0 = time.perf_counter(); tr.witness_to_device(ctx, d, N); out["witness_steps_to_device_pinned_2^%d_ms" % nv] = (time.perf_counter() - t0) * 1e3
        rows = tr.rows()
        ctx.witness_from_rows(rows, nv, d, N)
        t0 = time.perf_counter(); ctx.witness_from_rows(rows, nv, d, N); out["witness_rows_to_device_2^%d_ms" % nv] = (time.perf_counter() - t0) * 1e3
        del rows
        ctx.dev_free(d)
    print(json.dumps(out, indent=1))
    ctx.close()


if __name__ == "__main__":
    main()
