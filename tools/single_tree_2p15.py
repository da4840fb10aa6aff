import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, zigz_amd, oracle_lib as O
ctx = zigz_amd.Context(0)
N = 1 << 15
rng = np.random.default_rng(1)
other = np.repeat(rng.integers(0, 2013265921, size=N // 8, dtype=np.uint64), 8)
lv, _ = O.merkle_levels(other)
want = lv[(2 * N - 2) * 32:(2 * N - 1) * 32].tobytes()
for name, ra, cg in (("plain", 0, 0), ("run-aware", 1, 0), ("group", 0, 1)):
    ctx.set_option("run_aware_mask", ra); ctx.set_option("cons_group_mask", cg)
    print(name, flush=True)
    t = zigz_amd.SimpleMerkleTree(ctx, other)
    print(name, t.root_hash == want, flush=True)
    t.deinit()
ctx.close()
