import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, zigz_amd
ctx = zigz_amd.Context(0)
N = 1 << 20
d = ctx.dev_alloc(43 * N * 4)
ctx.upload((np.arange(43 * N, dtype=np.uint64) * 2654435761) % 2013265921, d)
ctx.enable_timing(True)
res = {0: [], 1: []}
roots = {}
for rep in range(6):
    for unf in (1, 0):
        ctx.set_option("unfused_levels", unf)
        job = zigz_amd.CommitJob(ctx, d_cols=d, ncols=43, nv=20)
        r = job.roots()
        st = ctx.stats()
        job.end()
        if rep:
            res[unf].append((st["merkle_build_us"], st["keccak_level_small_us"]))
        roots[unf] = r.tobytes()
print("identical roots:", roots[0] == roots[1])
for unf in (1, 0):
    a = np.array(res[unf])
    print("unfused" if unf else "fused  ", "build us", a[:, 0].round(), "small-level kernels us", a[:, 1].round(), "mean", a.mean(axis=0).round(1))
