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
        # the product path: radix form orchestrated in C++ (2-3 all-gathers per proof), same table + a 2^22 table timed
        d_loc = ctx.dev_alloc((1 << 16) // world * 4)
        ctx.upload(shard.interleave_rows(table, rank, world), d_loc)
        cb = shard.make_allgather(dist)
        r2, p2, fe2 = shard.sumcheck_prove_row_sharded_radix(ctx, d_loc, (1 << 16) // world, dist, cb)
        ctx.dev_free(d_loc)
        big = O.splitmix64_field(1888, 1 << 22)
        d_big = ctx.dev_alloc((1 << 22) // world * 4)
        ctx.upload(shard.interleave_rows(big, rank, world), d_big)
        import time
        shard.sumcheck_prove_row_sharded_radix(ctx, d_big, (1 << 22) // world, dist, cb)  # warm-up (workspaces)
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(5):
            rb, pb, feb = shard.sumcheck_prove_row_sharded_radix(ctx, d_big, (1 << 22) // world, dist, cb)
        ms_sharded = (time.perf_counter() - t0) / 5 * 1e3
        # ... and through the built-in same-node transport (shared-memory mailbox) instead of torch.distributed
        comm = shard.ShmComm("zigz_gpu_shard_%d" % port, rank, world)
        shard.sumcheck_prove_row_sharded_radix(ctx, d_big, (1 << 22) // world, dist, comm)
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(20):
            rs, ps, fes = shard.sumcheck_prove_row_sharded_radix(ctx, d_big, (1 << 22) // world, dist, comm)
        ms_shm = (time.perf_counter() - t0) / 20 * 1e3
        comm.close()
        assert np.array_equal(rs, rb) and np.array_equal(ps, pb) and fes == feb
        ctx.dev_free(d_big)
        radix = dict(ms_shm=ms_shm, rounds=r2.tolist(), point=p2.tolist(), fe=fe2, big_rounds=rb.tolist(), big_point=pb.tolist(), big_fe=feb,
                     ms_2p22=ms_sharded)
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
               dict(sha3=hashlib.sha3_256(sharded).hexdigest(), n=len(sharded), nv=tr.num_vars), radix))
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
    big = O.splitmix64_field(1888, 1 << 22)
    import zigz_amd
    with zigz_amd.Context(0) as c1:  # the unsharded GPU prover on the same 2^22 table (itself oracle-checked elsewhere)
        rb, pb, feb = c1.sumcheck_prove(big)
        import time
        d = c1.dev_alloc((1 << 22) * 4)
        c1.upload(big, d)
        c1.dev_sumcheck_prove(d, 1 << 22)
        t0 = time.perf_counter()
        for _ in range(5):
            c1.dev_sumcheck_prove(d, 1 << 22)
        ms_single = (time.perf_counter() - t0) / 5 * 1e3
        c1.dev_free(d)
    for rank, got, sc, mk, pr, rx in outs:
        assert rx["rounds"] == [int(x) for x in r] and rx["point"] == [int(x) for x in p] and rx["fe"] == fe, rank
        assert rx["big_rounds"] == [int(x) for x in rb] and rx["big_point"] == [int(x) for x in pb] and rx["big_fe"] == feb
        print("rank %d: row-sharded radix sumcheck 2^22 over 2 ranks on one GPU: %.3f ms (torch/gloo hook), %.3f ms "
              "(shared-memory hook); unsharded %.3f ms" % (rank, rx["ms_2p22"], rx["ms_shm"], ms_single))
        assert pr["nv"] >= 11 and pr["n"] == len(oproof) and pr["sha3"] == hashlib.sha3_256(oproof).hexdigest(), rank
        assert mk["root"] == mroot.hex() and mk["height"] == mheight
        for idx, sib, dirs, leaf in mk["opens"]:
            esib, edirs, eleaf = O.merkle_open(vals, idx)
            assert (sib, dirs, leaf) == (esib.hex(), edirs.hex(), eleaf), (rank, idx)
        for k in ("roots", "points", "values", "indices", "leaves", "siblings", "dirs"):
            assert np.array_equal(np.array(got[k], dtype=exp[k].dtype), exp[k]), (rank, k)
        assert got["next_challenge"] == nxt
        assert sc["rounds"] == [int(x) for x in r] and sc["point"] == [int(x) for x in p] and sc["fe"] == fe


def test_rccl_comm_one_rank_on_the_gpu():
    """The native RCCL transport (zigz_rccl_comm, csrc/rccl_comm.cpp; no torch involved) with the ONE rank this box allows:
    communicator init from a unique id, all-gather and all-reduce of host payloads through the staging buffers, the in-place
    device all-reduce, and zigz_dev_sumcheck_prove_sharded / _rccl through it against the unsharded oracle.  (With one GPU per
    box RCCL cannot carry a payload between two ranks here; world > 1 is covered over gloo / shared memory on the CPU.)"""
    import time
    import zigz_amd
    from zigz_amd import shard
    uid = shard.RcclComm.unique_id()
    assert len(uid) == 128
    comm = shard.RcclComm(0, uid, 0, 1, max_bytes=1 << 16)
    ctx = zigz_amd.Context(0)
    try:
        for n in (1, 16, 8191, 1 << 16):
            payload = bytes((7 * i + n) & 255 for i in range(n))
            assert comm.all_gather(payload) == [payload]
        words = np.arange(1, 1025, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        assert np.array_equal(comm.all_reduce_u64(words), words)
        table = O.splitmix64_field(4242, 1 << 16)
        d = ctx.dev_alloc((1 << 16) * 4)
        ctx.upload(table, d)
        r0, p0, fe0 = O.sumcheck_prove(P, table)
        r, p, fe = shard.sumcheck_prove_row_sharded_radix(ctx, d, 1 << 16, None, comm)    # the hook form
        assert np.array_equal(r, r0) and np.array_equal(p, p0) and fe == fe0
        # the in-place device all-reduce on the context's stream (with one rank: RCCL's identity), called directly ...
        # (torch only to put raw 64-bit words into HBM and read them back: the ABI's uploads narrow to field elements)
        import torch
        hw = np.concatenate([words, np.array([0], dtype=np.uint64)])
        tw = torch.from_numpy(hw.view(np.int64).copy()).to("cuda:0")
        torch.cuda.synchronize()
        comm.all_reduce_u64_dev(ctx, tw.data_ptr(), 1025)
        assert np.array_equal(tw.cpu().numpy().view(np.uint64), hw)
        del tw
        # ... and through the native form, which now issues it at world 1 as well: every radix stage's block sums + the
        # failure word go through ncclAllReduce on the context's stream and the deadline wait behind it (two stages at 2^22)
        r, p, fe = ctx.dev_sumcheck_prove_rccl(d, 1 << 16, comm)
        assert np.array_equal(r, r0) and np.array_equal(p, p0) and fe == fe0
        big = O.splitmix64_field(4243, 1 << 22)
        db = ctx.dev_alloc((1 << 22) * 4)
        ctx.upload(big, db)
        rb, pb, feb = ctx.dev_sumcheck_prove(db, 1 << 22)
        r, p, fe = ctx.dev_sumcheck_prove_rccl(db, 1 << 22, comm)
        assert np.array_equal(r, rb) and np.array_equal(p, pb) and fe == feb
        ctx.dev_free(db)
        t0 = time.perf_counter()
        for _ in range(200):
            comm.all_gather(b"x" * 8192)
        us = (time.perf_counter() - t0) / 200 * 1e6
        print("zigz_rccl_allgather, 8 KiB, one rank: %.1f us per exchange" % us)
        ctx.dev_free(d)
    finally:
        ctx.close()
        comm.close()
