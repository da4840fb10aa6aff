#!/usr/bin/env python3
"""A/B of Keccak kernel variants on ONE box in ONE process (box-to-box differences are ~10 %): interleaved cold launches of
k_keccak_leaves / k_keccak_level (43 x 2^20) per variant (ctx option "keccak_variant"), plus a tree-identity check."""
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import zigz_amd

variants = [int(v) for v in (sys.argv[1:] or ["0", "1"])]
ctx = zigz_amd.Context(0)
res = {v: {"leaves": [], "level": []} for v in variants}
for rep in range(4):
    for v in variants:
        ctx.set_option("keccak_variant", v)
        res[v]["leaves"].append(ctx.bench_kernel("k_keccak_leaves", 20, 43, 6, True)["avg_us"])
        res[v]["level"].append(ctx.bench_kernel("k_keccak_level", 20, 43, 6, True)["avg_us"])
# identical trees: roots of a ragged tree and a 2^13 x 43 commit under every variant
vals = (np.arange(5000, dtype=np.uint64) * 2654435761) % 2013265921
cols = ((np.arange(43 * 8192, dtype=np.uint64) * 40503 + 7) % 2013265921).reshape(43, 8192)
roots = {}
for v in variants:
    ctx.set_option("keccak_variant", v)
    t = zigz_amd.SimpleMerkleTree(ctx, vals)
    job = zigz_amd.CommitJob(ctx, cols=cols)
    roots[v] = (t.getRoot().hex(), job.roots().tobytes().hex())
    job.end(); t.deinit()
same = all(roots[v] == roots[variants[0]] for v in variants)
for v in variants:
    l, k = res[v]["leaves"], res[v]["level"]
    print("variant %d: leaves %s -> min %.1f us (%.2f Gperm/s); level %s -> min %.1f us (%.2f Gperm/s)" %
          (v, [round(x) for x in l], min(l), 43 * 2**20 / min(l) / 1e3, [round(x) for x in k], min(k), 43 * 2**19 / min(k) / 1e3))
print("identical trees:", same)
ctx.close()
sys.exit(0 if same else 1)
