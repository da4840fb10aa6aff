"""N > 1 path on CPU: world_size-2/4 gloo runs of zigz_amd.shard's column-sharded generateCommitments and
row-sharded sumcheck (exchange logic), with oracle-backed compute stand-ins, against the unsharded oracle."""
import os
import time
import socket

import numpy as np
import pytest

import oracle_lib as O

P = O.P_BB
torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _OrcTranscript(O.Transcript):
    def challenge(self):  # BabyBear, like zigz_amd.Transcript
        return super().challenge(P)


def _worker(rank, world, port, kind, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from zigz_amd import shard
    import fake_engine
    try:
        if kind == "columns":
            nv = 5
            cols = O.splitmix64_field(77, 43 * (1 << nv)).reshape(43, 1 << nv)
            c0, c1 = shard.column_partition(43, world)[rank]
            tr = _OrcTranscript(); tr.append_bytes(b"prefix")
            res = shard.generate_commitments_sharded(fake_engine.FakeEngine(), tr, cols[c0:c1], nv, dist)
            res["next_challenge"] = tr.challenge()
            q.put((rank, {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in res.items()}))
        elif kind.startswith("merkle"):
            n = int(kind.split(":")[1])
            vals = O.splitmix64_field(99, n)
            nl = n // world
            t = shard.RowShardedMerkle(fake_engine.FakeTreeOps(), vals[rank * nl:(rank + 1) * nl], n, dist)
            opens = []
            for idx in sorted({0, 1, n // 2, n - 1, (n * 5) // 7}):
                sib, dirs, leaf = t.open(idx)
                opens.append((idx, sib.tobytes().hex(), dirs.tobytes().hex(), leaf))
            t.close()
            q.put((rank, dict(root=t.root.hex(), height=t.height, opens=opens)))
        elif kind.startswith("radix"):
            nv = int(kind.split(":")[1])
            table = O.splitmix64_field(880 + nv, 1 << nv)
            local = shard.interleave_rows(table, rank, world)
            r, p, fe = shard.sumcheck_radix_run(fake_engine.FakeRadixOps(local), len(local), dist)
            q.put((rank, dict(rounds=r.tolist(), point=p.tolist(), fe=fe)))
        else:
            nv = 9
            table = O.splitmix64_field(88, 1 << nv)
            local = shard.interleave_rows(table, rank, world)
            r, p, fe = shard.sumcheck_prove_row_sharded(fake_engine.FakeOps(), local, 1 << nv, dist, _OrcTranscript)
            q.put((rank, dict(rounds=r.tolist(), point=p.tolist(), fe=fe)))
    finally:
        dist.destroy_process_group()


def _run(world, kind):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, q)) for r in range(world)]
    [p.start() for p in procs]
    out = dict(q.get(timeout=120) for _ in range(world))
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    return out


def test_column_partition():
    from zigz_amd import shard
    assert [b - a for a, b in shard.column_partition(43, 8)] == [6, 6, 6, 5, 5, 5, 5, 5]
    assert shard.column_partition(43, 1) == [(0, 43)]
    for w in (2, 3, 4, 8):
        part = shard.column_partition(43, w)
        assert part[0][0] == 0 and part[-1][1] == 43 and all(part[i][1] == part[i + 1][0] for i in range(w - 1))


@pytest.mark.parametrize("world", [2, 3])
def test_column_sharded_commitments_gloo(world):
    out = _run(world, "columns")
    nv = 5
    cols = O.splitmix64_field(77, 43 * (1 << nv)).reshape(43, 1 << nv)
    tr = O.Transcript(); tr.append_bytes(b"prefix")
    exp = O.generate_commitments(P, tr, cols)
    nxt = tr.challenge(P)
    for rank in range(world):
        got = out[rank]
        for k in ("roots", "points", "values", "indices", "leaves", "siblings", "dirs"):
            assert np.array_equal(np.array(got[k], dtype=exp[k].dtype), exp[k]), (rank, k)
        assert got["next_challenge"] == nxt  # transcripts stayed in lockstep on every rank


@pytest.mark.parametrize("world", [2, 4])
def test_row_sharded_sumcheck_gloo(world):
    out = _run(world, "rows")
    table = O.splitmix64_field(88, 1 << 9)
    r, p, fe = O.sumcheck_prove(P, table)
    for rank in range(world):
        assert out[rank]["rounds"] == [int(x) for x in r] and out[rank]["point"] == [int(x) for x in p] and out[rank]["fe"] == fe


@pytest.mark.parametrize("world,nv", [(2, 13), (4, 13), (2, 12), (4, 14), (2, 3), (4, 2), (2, 1)])
def test_row_sharded_radix_sumcheck_gloo(world, nv):
    """The C++ radix orchestration of zigz_dev_sumcheck_prove_sharded (zigz_sumcheck_radix_run: same code, stand-in
    data passes) with world ranks over gloo: 2-3 all-gathers per proof; rounds, challenges and final_eval equal the
    unsharded reference prover's on every rank -- with device stages (local tables >= 2^11) and without."""
    out = _run(world, "radix:%d" % nv)
    table = O.splitmix64_field(880 + nv, 1 << nv)
    r, p, fe = O.sumcheck_prove(P, table)
    for rank in range(world):
        assert out[rank]["rounds"] == [int(x) for x in r] and out[rank]["point"] == [int(x) for x in p] and out[rank]["fe"] == fe


def test_radix_run_single_rank_matches_oracle():
    """world = 1 through the same entry (no hook): the unsharded radix sumcheck, 1 and 2 device stages."""
    from zigz_amd import shard
    import fake_engine
    for nv in (11, 12, 15):
        table = O.splitmix64_field(990 + nv, 1 << nv)
        r, p, fe = shard.sumcheck_radix_run(fake_engine.FakeRadixOps(table), 1 << nv, None)
        r0, p0, fe0 = O.sumcheck_prove(P, table)
        assert np.array_equal(r, r0) and np.array_equal(p, p0) and fe == fe0
        chs = O.splitmix64_field(5, nv)
        r, p, fe = shard.sumcheck_radix_run(fake_engine.FakeRadixOps(table), 1 << nv, None, challenges=chs)
        r0, p0, fe0 = O.sumcheck_prove(P, table, chs)
        assert np.array_equal(r, r0) and np.array_equal(p, p0) and fe == fe0


def _shm_worker(rank, world, name, nv, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from zigz_amd import shard
    import fake_engine
    comm = shard.ShmComm(name, rank, world, max_bytes=1 << 14, timeout_s=60)
    try:
        echo = []
        for it in range(50):  # many back-to-back exchanges of varying size: the two slot sets are reused correctly
            n = 1 + (it * 37) % 300
            got = comm.all_gather(bytes([(rank * 17 + it + j) & 255 for j in range(n)]))
            echo.append(all(got[r] == bytes([(r * 17 + it + j) & 255 for j in range(n)]) for r in range(world)))
        table = O.splitmix64_field(660 + nv, 1 << nv)
        local = shard.interleave_rows(table, rank, world)
        r, p, fe = shard.sumcheck_radix_run(fake_engine.FakeRadixOps(local), len(local), None, allgather=comm)
        q.put((rank, dict(echo=all(echo), rounds=r.tolist(), point=p.tolist(), fe=fe)))
    finally:
        comm.close()


def _failing_rank_worker(rank, world, name, nv, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from zigz_amd import shard
    import zigz_amd
    import fake_engine

    class Broken(fake_engine.FakeRadixOps):
        def fold(self, k, weights, k_next):
            if rank == 1:
                raise ValueError("this rank's data pass fails")
            return super().fold(k, weights, k_next)
    comm = shard.ShmComm(name, rank, world, max_bytes=1 << 14, timeout_s=60)
    t0 = time.perf_counter()
    try:
        local = shard.interleave_rows(O.splitmix64_field(5, 1 << nv), rank, world)
        shard.sumcheck_radix_run(Broken(local), len(local), None, allgather=comm)
        q.put((rank, "no error", 0.0))
    except zigz_amd.ZigzError as e:
        q.put((rank, e.name, time.perf_counter() - t0))
    except ValueError:
        q.put((rank, "own error", time.perf_counter() - t0))
    finally:
        comm.close()


def test_sharded_sumcheck_fails_on_all_ranks_together():
    """a rank whose local pass fails must not leave its peers in the next all-gather until the transport's
    timeout (60 s here).  Its status rides in the next exchange: it returns its own error, the others CommError, at once."""
    world, nv = 4, 14
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = "zigz_test_fail_%d" % os.getpid()
    procs = [ctx.Process(target=_failing_rank_worker, args=(r, world, name, nv, q)) for r in range(world)]
    [p.start() for p in procs]
    out = {r: (what, dt) for r, what, dt in (q.get(timeout=120) for _ in range(world))}
    [p.join(timeout=60) for p in procs]
    assert out[1][0] == "own error"
    for r in (0, 2, 3):
        assert out[r][0] == "CommError", out
    assert max(dt for _, dt in out.values()) < 20.0, out


class _ReducedOps:
    """Stand-in passes with the semantics of the RCCL passes of zigz_dev_sumcheck_prove_rccl (csrc/api_mle.cpp: sums_out): a
    pass returns the sums over ALL ranks -- reduced inside the pass by a collective of its own, which carries one more word,
    the number of ranks whose local pass failed, and which EVERY rank takes part in whatever happened locally."""

    def __init__(self, inner, comm, fail_in=None):
        self.inner, self.comm, self.fail_in = inner, comm, fail_in

    def _reduce(self, fn, n):
        failed, v = 0, [0] * n
        try:
            v = fn()
        except ValueError:
            failed = 1
        words = [int(x) for x in v] + [failed]
        got = self.comm.all_gather(b"".join(int(w).to_bytes(16, "little") for w in words))  # (the collective: always issued)
        tot = [sum(int.from_bytes(g[16 * i:16 * i + 16], "little") for g in got) for i in range(n + 1)]
        if failed:
            raise ValueError("this rank's data pass fails")
        if tot[n]:
            raise RuntimeError("a peer reported a failure inside the collective")
        return tot[:n]

    def block_sums(self, k):
        def local():
            if self.fail_in == "block_sums":
                raise ValueError
            return self.inner.block_sums(k)
        return self._reduce(local, 1 << k)

    def fold(self, k, weights, k_next):
        def local():
            if self.fail_in == "fold":
                raise ValueError
            return self.inner.fold(k, weights, k_next) or []
        if not k_next:
            return local()
        return self._reduce(local, 1 << k_next)

    def read_tail(self, m):
        return self.inner.read_tail(m)


def _reduced_worker(rank, world, name, nv, fail_in, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from zigz_amd import shard
    import zigz_amd
    import fake_engine
    comm = shard.ShmComm(name, rank, world, max_bytes=1 << 15, timeout_s=60)
    t0 = time.perf_counter()
    try:
        local = shard.interleave_rows(O.splitmix64_field(41 + nv, 1 << nv), rank, world)
        ops = _ReducedOps(fake_engine.FakeRadixOps(local), comm, fail_in if rank == 1 else None)
        r, p, fe = shard.sumcheck_radix_run(ops, len(local), None, allgather=comm, reduced=True)
        q.put((rank, "ok", dict(rounds=r.tolist(), point=p.tolist(), fe=fe), time.perf_counter() - t0))
    except zigz_amd.ZigzError as e:
        q.put((rank, e.name, None, time.perf_counter() - t0))
    except ValueError:
        q.put((rank, "own error", None, time.perf_counter() - t0))
    except RuntimeError:
        q.put((rank, "peer failed", None, time.perf_counter() - t0))
    finally:
        comm.close()


@pytest.mark.parametrize("fail_in", [None, "block_sums", "fold"])
def test_radix_run_with_reduced_sums_stays_in_step(fail_in):
    """with the sums reduced inside the passes (the RCCL form) there is no host exchange between the stages, so
    a rank whose pass fails must still take part in that pass's collective, and its peers must learn of the failure THERE:
    otherwise the failing rank goes straight to the tail exchange while the others enqueue the next stage's all-reduce, and
    RCCL -- which has no timeout -- never returns.  zigz_sumcheck_radix_run_reduced over stand-in passes with exactly the
    protocol of csrc/api_mle.cpp: sums_out, 2^15 rows over 4 ranks (two stages): without a failure the proof is the unsharded
    oracle's; with one, every rank returns at once -- rank 1 with its own error, the others with what the collective told them."""
    world, nv = 4, 15
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = "zigz_test_red_%d_%s" % (os.getpid(), fail_in)
    procs = [ctx.Process(target=_reduced_worker, args=(r, world, name, nv, fail_in, q)) for r in range(world)]
    [p.start() for p in procs]
    out = {r: (what, res, dt) for r, what, res, dt in (q.get(timeout=120) for _ in range(world))}
    [p.join(timeout=60) for p in procs]
    assert max(dt for _, _, dt in out.values()) < 30.0, out
    if fail_in is None:
        r, p, fe = O.sumcheck_prove(P, O.splitmix64_field(41 + nv, 1 << nv))
        for rank in range(world):
            what, res, _ = out[rank]
            assert what == "ok" and res["rounds"] == [int(x) for x in r] and res["point"] == [int(x) for x in p] and res["fe"] == fe
    else:
        assert out[1][0] == "own error", out
        # (at this size the table has ONE device stage: a failing fold has no collective of its own and the peers learn of it
        # in the tail exchange -- CommError --; a failing block-sums pass tells them inside its collective)
        for rank in (0, 2, 3):
            assert out[rank][0] == ("peer failed" if fail_in == "block_sums" else "CommError"), out


@pytest.mark.parametrize("world,nv", [(2, 13), (4, 14), (8, 14)])
def test_row_sharded_radix_sumcheck_shm(world, nv):
    """The same orchestration with the built-in same-node transport (zigz_shm_comm, no torch, no sockets):
    world processes, 2-3 shared-memory all-gathers per proof, results equal the unsharded reference prover's."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = "zigz_test_%d_%d_%d" % (os.getpid(), world, nv)
    procs = [ctx.Process(target=_shm_worker, args=(r, world, name, nv, q)) for r in range(world)]
    [p.start() for p in procs]
    out = dict(q.get(timeout=180) for _ in range(world))
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    table = O.splitmix64_field(660 + nv, 1 << nv)
    r, p, fe = O.sumcheck_prove(P, table)
    for rank in range(world):
        assert out[rank]["echo"]
        assert out[rank]["rounds"] == [int(x) for x in r] and out[rank]["point"] == [int(x) for x in p] and out[rank]["fe"] == fe


def test_shm_comm_times_out_instead_of_hanging():
    """A rank whose peers never arrive gets an error after timeout_s, not a hang ('the others hang').  Creation is
    a collective with a handshake, so it is the create call that times out -- on either side."""
    from zigz_amd import shard
    import zigz_amd
    with pytest.raises(zigz_amd.ZigzError):      # rank 1 of 2, rank 0 never creates the segment
        shard.ShmComm("zigz_test_absent_%d" % os.getpid(), 1, 2, timeout_s=0.3)
    with pytest.raises(zigz_amd.ZigzError):      # rank 0 of 2, rank 1 never attaches
        shard.ShmComm("zigz_test_alone_%d" % os.getpid(), 0, 2, timeout_s=0.3)
    assert not os.path.exists("/dev/shm/zigz_test_alone_%d" % os.getpid())  # ... and it takes its segment with it


def _stale_segment_rank(rank, world, name, q):
    from zigz_amd import shard
    try:
        comm = shard.ShmComm(name, rank, world, max_bytes=256, timeout_s=20)
        outs = [comm.all_gather(bytes([rank + 1]) * 16 + bytes([i]) * 16) for i in range(5)]
        comm.close()
        q.put((rank, outs))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))


def test_shm_comm_ignores_a_leftover_segment():
    """names get reused, and a crashed job leaves its segment behind -- initialised (MAGIC set, matching world and
    max_bytes) and with sequence counters far ahead.  An attacher that opens it before the new rank 0 has replaced it must
    not take it for the live one (it would return at once from every all-gather with old slot bytes): nobody answers its
    hello there, so it lets go and finds the new segment."""
    import multiprocessing as mp
    import struct
    world, name = 3, "zigz_test_stale_%d" % os.getpid()
    max_bytes = 256
    size = 4096 + world * 64 + 2 * world * max_bytes
    blob = bytearray(size)
    struct.pack_into("<IIQ", blob, 0, 0x5A49475A, world, max_bytes)        # ready = MAGIC, world, max_bytes
    for r in range(world):
        struct.pack_into("<Q", blob, 4096 + 64 * r, 1 << 40)               # seq[r]: every future all-gather "already done"
    for i in range(4096 + world * 64, size):
        blob[i] = 0xEE                                                      # old slot bytes
    with open("/dev/shm/" + name, "wb") as f:
        f.write(blob)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_stale_segment_rank, args=(r, world, name, q)) for r in (1, 2)]
    for p in procs:
        p.start()
    time.sleep(0.5)                                # ranks 1 and 2 are looking at the leftover now
    p0 = ctx.Process(target=_stale_segment_rank, args=(0, world, name, q))
    p0.start()
    res = dict(q.get(timeout=60) for _ in range(world))
    for p in procs + [p0]:
        p.join(30)
    for r in range(world):
        assert isinstance(res[r], list), res[r]
        for i, out in enumerate(res[r]):
            assert b"".join(out) == b"".join(bytes([k + 1]) * 16 + bytes([i]) * 16 for k in range(world)), (r, i)
    assert not os.path.exists("/dev/shm/" + name)


@pytest.mark.parametrize("world,n", [(2, 64), (4, 64), (4, 4), (2, 2)])
def test_row_sharded_merkle_gloo(world, n):
    """Contiguous row ownership: subtree per rank, G roots all-gathered, top levels on every rank; root and openings
    equal the unsharded SimpleMerkleTree's (also when a rank holds a single leaf)."""
    out = _run(world, "merkle:%d" % n)
    vals = O.splitmix64_field(99, n)
    root, height = O.merkle_build(vals)
    for rank in range(world):
        got = out[rank]
        assert got["root"] == root.hex() and got["height"] == height
        for idx, sib, dirs, leaf in got["opens"]:
            esib, edirs, eleaf = O.merkle_open(vals, idx)
            assert (sib, dirs, leaf) == (esib.hex(), edirs.hex(), eleaf), (rank, idx)
            assert O.merkle_verify(root, leaf, bytes.fromhex(sib), bytes.fromhex(dirs))


def test_shm_comm_under_thread_sanitizer(tmp_path):
    """The shared-memory transport's protocol (two slot sets, release / acquire on the sequence counters) with the ranks as
    threads of one process under ThreadSanitizer: 2 000 back-to-back exchanges of varying size, every byte checked."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "shm_tsan")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "c_driver", "shm_tsan.cpp"),
                           os.path.join(root, "zigz_amd", "csrc", "shm_comm.cpp"), "-o", exe, "-lrt"])
    for world in (2, 4):
        r = subprocess.run([exe, str(world), "2000"], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
        assert r.returncode == 0 and r.stdout.startswith("ok:"), r.stdout[-2000:] + r.stderr[-4000:]
