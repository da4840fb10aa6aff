#!/usr/bin/env python3
"""Three lone 2^20 proofs of the bench trace (resident witness), for a per-launch view of one Merkle build:

    cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d <dir> -o t -- python3 tools/trace_one_proof.py

then sort <dir>/t_kernel_trace.csv by Start_Timestamp (this is how the run-aware level launches were timed)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import zigz_amd, programs
from zigz_amd import host
ctx = zigz_amd.Context(0)
N = 1 << 20
prog = programs.add_xor_loop((N - 3) // 4)
tr = host.Trace(prog, 0x1000, None, 2 * N)
d = ctx.dev_alloc(43 * N * 4)
tr.witness_to_device(ctx, d, N)
import json
for _ in range(3):
    tr.prove(ctx, d, N, want_bytes="borrow")
ctx.synchronize()
print(json.dumps({k: v for k, v in ctx.stats().items() if not k.endswith("_us")}))  # what the last build hashed
