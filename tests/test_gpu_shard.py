"""-m gpu: the sharded paths of zigz_amd.shard on the real HIP engines, world_size 2 (both ranks on the one
GPU of the test box, gloo for the tiny exchanges), against the unsharded oracle."""
import os
import socket

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
P = O.P_BB
torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import zigz_amd
    from zigz_amd import shard
    ctx = zigz_amd.Context(0)
    try:
        nv = 11
        cols = O.splitmix64_field(177, 43 * (1 << nv)).reshape(43, 1 << nv)
        c0, c1 = shard.column_partition(43, world)[rank]
        tr = zigz_amd.Transcript(); tr.append_bytes(b"prefix")
        res = shard.generate_commitments_sharded(shard.GpuEngine(ctx), tr, cols[c0:c1], nv, dist)
        res["next_challenge"] = tr.challenge()
        table = O.splitmix64_field(188, 1 << 16)
        ops = shard.GpuOps(ctx)
        local = ops.upload(shard.interleave_rows(table, rank, world))
        r, p, fe = shard.sumcheck_prove_row_sharded(ops, local, 1 << 16, dist, zigz_amd.Transcript)
        ops.close()
        nm = 1 << 14
        vals = O.splitmix64_field(199, nm)
        t = shard.RowShardedMerkle(shard.GpuTreeOps(ctx), vals[rank * nm // world:(rank + 1) * nm // world], nm, dist)
        opens = []
        for idx in (0, nm // 2 - 1, nm // 2, nm - 1, 12345):
            sib, dirs, leaf = t.open(idx)
            opens.append((idx, sib.tobytes().hex(), dirs.tobytes().hex(), int(leaf)))
        t.close()
        # one whole proof, sharded by column through the C++ host mirror (zigzh_prove_trace_sharded)
        import hashlib
        import programs
        from zigz_amd import host
        prog, inp = programs.fibonacci(700)
        tr = host.Trace(prog, 0x1000, None, 1 << 14, inp)
        N = 1 << tr.num_vars
        d = ctx.dev_alloc(43 * N * 4)
        tr.witness_to_device(ctx, d, N)
        sharded = tr.prove_sharded(ctx, d, N, dist).tobytes()
        ctx.dev_free(d)
        q.put((rank, {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in res.items()},
               dict(rounds=r.tolist(), point=p.tolist(), fe=fe), dict(root=t.root.hex(), height=t.height, opens=opens),
               dict(sha3=hashlib.sha3_256(sharded).hexdigest(), n=len(sharded), nv=tr.num_vars)))
    finally:
        ctx.close()
        dist.destroy_process_group()


def test_sharded_paths_world2_on_gpu():
    world = 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    outs = [q.get(timeout=300) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    nv = 11
    cols = O.splitmix64_field(177, 43 * (1 << nv)).reshape(43, 1 << nv)
    tr = O.Transcript(); tr.append_bytes(b"prefix")
    exp = O.generate_commitments(P, tr, cols, fast=True)
    nxt = tr.challenge(P)
    table = O.splitmix64_field(188, 1 << 16)
    r, p, fe = O.sumcheck_prove(P, table)
    vals = O.splitmix64_field(199, 1 << 14)
    mroot, mheight = O.merkle_build(vals)
    import hashlib
    import programs
    prog, inp = programs.fibonacci(700)
    oproof, _ = O.prove(P, prog, 0x1000, None, 1 << 14, inp)
    for rank, got, sc, mk, pr in outs:
        assert pr["nv"] >= 11 and pr["n"] == len(oproof) and pr["sha3"] == hashlib.sha3_256(oproof).hexdigest(), rank
        assert mk["root"] == mroot.hex() and mk["height"] == mheight
        for idx, sib, dirs, leaf in mk["opens"]:
            esib, edirs, eleaf = O.merkle_open(vals, idx)
            assert (sib, dirs, leaf) == (esib.hex(), edirs.hex(), eleaf), (rank, idx)
        for k in ("roots", "points", "values", "indices", "leaves", "siblings", "dirs"):
            assert np.array_equal(np.array(got[k], dtype=exp[k].dtype), exp[k]), (rank, k)
        assert got["next_challenge"] == nxt
        assert sc["rounds"] == [int(x) for x in r] and sc["point"] == [int(x) for x in p] and sc["fe"] == fe
