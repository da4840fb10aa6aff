"""Sharding ONE proof over several GPUs (one process per GPU, torch.distributed; backend "nccl" = RCCL
over xGMI on a node, "gloo" in the CPU tests).  SURVEY.md s8(e):

* columns  -- `generate_commitments_sharded`: the 43 (Merkle tree, eval, opening) jobs of
  Prover.generateCommitments (src/prover/prover.zig:366-467) are independent; rank r takes a contiguous
  block of columns.  Exchange steps: one all-gather of the roots (43 x 32 B) before the transcript absorbs
  them, one all-gather of the openings (value, index, leaf, v x 33 B) after.  Every rank runs the same
  deterministic transcript, so challenges need no broadcast.
* rows     -- a single large table is owned interleaved (global index i lives on rank i mod G), which keeps every
  MSB-bind pair (i, i + n/2) on one rank for the first v - log2(G) rounds.  `sumcheck_prove_row_sharded_radix` (the
  product path) hands the whole proof to the C++ radix orchestration of libzigz_hip.so: 2-3 exchanges of <= 1024 u64
  block sums per proof.  `sumcheck_prove_row_sharded` is the per-round form kept as a second, independent statement
  of the same proof for the tests: ONE all-reduce(sum) of the two partial half sums (16 bytes) per round, the last
  log2(G) rounds on the all-gathered G-element table.

* rows (Merkle) -- `RowShardedMerkle`: one large column owned by CONTIGUOUS slices (top log2 G index bits), the
  layout a Merkle tree prefers: each rank builds the subtree over its slice, the G subtree roots (G x 32 B) are
  all-gathered, and every rank finishes the top log2 G levels itself (G - 1 host SHA3 calls).  An opening is the
  owner's local path (all-gathered, only the owner contributes) followed by the siblings of the top levels.
  Both row layouts coexist because the committed columns and the sumcheck table are different buffers.

Messages are 16 B - a few KiB: latency-bound, far below xGMI link bandwidth.  The compute is delegated to an
`ops` object (GpuOps below = libzigz_hip.so; the CPU tests inject an oracle-backed stand-in), so this file
holds only the partitioning and exchange logic.
"""
import numpy as np

P = 2013265921
NUM_COLUMNS = 43


def column_partition(ncols, world):
    """Contiguous blocks, sizes differing by at most one: 43 over 8 -> 6,6,6,5,5,5,5,5."""
    base, extra = divmod(ncols, world)
    out, c = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((c, c + n))
        c += n
    return out


def _dist_device(dist):
    import torch
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def _all_gather_bytes(dist, payload, nbytes_max):
    """all-gather of one fixed-size byte buffer per rank -> list of numpy uint8 arrays."""
    import torch
    dev = _dist_device(dist)
    buf = np.zeros(nbytes_max, dtype=np.uint8)
    buf[: len(payload)] = np.frombuffer(payload, dtype=np.uint8)
    t = torch.from_numpy(buf).to(dev)
    outs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, t)
    return [o.cpu().numpy() for o in outs]


def _all_reduce_u64(dist, values):
    import torch
    dev = _dist_device(dist)
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=dev)  # sums < G * 2^31: no overflow
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(v) for v in t.cpu().tolist()]


def generate_commitments_sharded(engine, transcript, local_cols, nv, dist, ncols=NUM_COLUMNS):
    """Column-sharded Prover.generateCommitments.  `local_cols`: this rank's block of columns (whatever the
    engine's `begin` accepts: a [n_local, 2^nv] uint64 array, or a device handle).  `transcript`: a
    FiatShamirTranscript-like object (append_bytes / append_field / challenge) in the state just before
    "POLY_COMMITMENTS" -- identical on every rank.  Returns the full 43-column result on every rank."""
    rank, world = dist.get_rank(), dist.get_world_size()
    part = column_partition(ncols, world)
    c0, c1 = part[rank]
    nmax = max(b - a for a, b in part)
    job = engine.begin(local_cols, c1 - c0, nv)
    try:
        roots_local = np.ascontiguousarray(job.roots(), dtype=np.uint8).reshape(-1)          # PHASE 1
        gathered = _all_gather_bytes(dist, roots_local.tobytes(), nmax * 32)                    # exchange 1
        roots = np.concatenate([gathered[r][: (part[r][1] - part[r][0]) * 32] for r in range(world)]).reshape(ncols, 32)
        transcript.append_bytes(b"POLY_COMMITMENTS")                                            # PHASE 2
        for c in range(ncols):
            transcript.append_bytes(roots[c].tobytes())
        points = np.array([[transcript.challenge() for _ in range(nv)] for _ in range(ncols)],
                          dtype=np.uint64).reshape(ncols, nv)                                   # PHASE 3
        o = job.open_all(points[c0:c1])
    finally:
        job.end()
    rec = 24 + 33 * nv  # value, index, leaf (3 x u64) + siblings + dirs per column
    payload = bytearray()
    for k in range(c1 - c0):
        payload += np.array([o["values"][k], o["indices"][k], o["leaves"][k]], dtype="<u8").tobytes()
        payload += np.ascontiguousarray(o["siblings"][k], dtype=np.uint8).tobytes()
        payload += np.ascontiguousarray(o["dirs"][k], dtype=np.uint8).tobytes()
    gathered = _all_gather_bytes(dist, bytes(payload), nmax * rec)                              # exchange 2
    values = np.zeros(ncols, dtype=np.uint64)
    indices = np.zeros(ncols, dtype=np.uint64)
    leaves = np.zeros(ncols, dtype=np.uint64)
    siblings = np.zeros((ncols, nv, 32), dtype=np.uint8)
    dirs = np.zeros((ncols, nv), dtype=np.uint8)
    for r in range(world):
        a, b = part[r]
        for k in range(b - a):
            chunk = gathered[r][k * rec:(k + 1) * rec]
            values[a + k], indices[a + k], leaves[a + k] = np.frombuffer(chunk[:24].tobytes(), dtype="<u8")
            siblings[a + k] = chunk[24:24 + 32 * nv].reshape(nv, 32)
            dirs[a + k] = chunk[24 + 32 * nv:24 + 33 * nv]
    transcript.append_bytes(b"OPENING_CLAIMS")                                                  # PHASE 4
    for c in range(ncols):
        transcript.append_field(int(values[c]))
    return dict(roots=roots, points=points, values=values, indices=indices, leaves=leaves, siblings=siblings, dirs=dirs)


def interleave_rows(table, rank, world):
    """The slice of a global table owned by `rank` under interleaved ownership (index mod world)."""
    return np.ascontiguousarray(np.asarray(table)[rank::world])


def make_allgather(dist):
    """zigz_allgather_fn on torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" in the tests): every rank
    contributes `nbytes` from `send`, `recv` receives world * nbytes in rank order.  Keep the returned object alive
    while C code may call it."""
    import ctypes as C
    import torch
    from ._ffi import ALLGATHER_FN
    world = dist.get_world_size()
    dev = _dist_device(dist)

    def hook(_user, send, nbytes, recv):
        try:
            src = torch.frombuffer((C.c_uint8 * nbytes).from_address(send), dtype=torch.uint8).to(dev)
            outs = [torch.empty_like(src) for _ in range(world)]
            dist.all_gather(outs, src)
            dst = (C.c_uint8 * (world * nbytes)).from_address(recv)
            for r, o in enumerate(outs):
                C.memmove(C.addressof(dst) + r * nbytes, o.cpu().numpy().ctypes.data, nbytes)
            return 0
        except Exception:  # never unwind into C
            return 1

    return ALLGATHER_FN(hook)


class ShmComm:
    """zigz_shm_comm: the built-in same-node all-gather (shared-memory mailbox).  `hook` / `user` are what the C entry
    points take as (allgather, user); `all_gather(bytes)` is the same exchange from Python."""

    def __init__(self, name, rank, world, max_bytes=1 << 16, timeout_s=60.0):
        import ctypes as C
        from . import _ffi, errors
        h = _ffi.vp()
        rc = _ffi.lib.zigz_shm_comm_create(name.encode(), rank, world, max_bytes, timeout_s, C.byref(h))
        if rc != 0:
            raise errors.ZigzError(rc, _ffi.lib.zigz_status_name(rc).decode(), "zigz_shm_comm_create(%s)" % name)
        self.h, self.rank, self.world = h, rank, world
        self.hook = C.cast(_ffi.lib.zigz_shm_allgather, _ffi.ALLGATHER_FN)
        self.user = h

    def all_gather(self, payload):
        import ctypes as C
        from . import _ffi
        n = len(payload)
        send = (C.c_uint8 * max(n, 1)).from_buffer_copy(bytes(payload) or b"\0")
        recv = (C.c_uint8 * max(n * self.world, 1))()
        if _ffi.lib.zigz_shm_allgather(self.h, send, n, recv) != 0:
            raise RuntimeError("zigz_shm_allgather failed (a rank timed out)")
        return [bytes(recv[r * n:(r + 1) * n]) for r in range(self.world)]

    def close(self):
        if self.h:
            from . import _ffi
            _ffi.lib.zigz_shm_comm_destroy(self.h)
            self.h = None


class RcclComm:
    """zigz_rccl_comm: RCCL as the exchange transport, bound natively (no torch in the loop).  `unique_id`: the 128 bytes of
    RcclComm.unique_id() from ONE rank, distributed by the caller (torch.distributed.broadcast_object_list, a file, MPI ...).
    `hook` / `user` are what the C entry points take as (allgather, user)."""

    @staticmethod
    def unique_id():
        import ctypes as C
        from . import _ffi, errors
        buf = (C.c_uint8 * 128)()
        rc = _ffi.lib.zigz_rccl_unique_id(buf)
        if rc != 0:
            raise errors.ZigzError(rc, _ffi.lib.zigz_status_name(rc).decode(), "zigz_rccl_unique_id")
        return bytes(buf)

    def __init__(self, device, unique_id, rank, world, max_bytes=1 << 16):
        import ctypes as C
        from . import _ffi, errors
        h = _ffi.vp()
        idb = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        rc = _ffi.lib.zigz_rccl_comm_create(device, idb, rank, world, max_bytes, C.byref(h))
        if rc != 0:
            raise errors.ZigzError(rc, _ffi.lib.zigz_status_name(rc).decode(), "zigz_rccl_comm_create")
        self.h, self.rank, self.world = h, rank, world
        self.hook = C.cast(_ffi.lib.zigz_rccl_allgather, _ffi.ALLGATHER_FN)
        self.user = h

    def all_gather(self, payload):
        import ctypes as C
        from . import _ffi
        n = len(payload)
        send = (C.c_uint8 * max(n, 1)).from_buffer_copy(bytes(payload) or b"\0")
        recv = (C.c_uint8 * max(n * self.world, 1))()
        rc = _ffi.lib.zigz_rccl_allgather(self.h, send, n, recv)
        if rc != 0:
            raise RuntimeError("zigz_rccl_allgather failed (%d)" % rc)
        return [bytes(recv[r * n:(r + 1) * n]) for r in range(self.world)]

    def all_reduce_u64(self, words):
        import numpy as np
        from . import _ffi
        a = np.ascontiguousarray(words, dtype=np.uint64)
        out = np.zeros_like(a)
        rc = _ffi.lib.zigz_rccl_allreduce_u64(self.h, a.ctypes.data_as(_ffi.u64p), a.size, out.ctypes.data_as(_ffi.u64p))
        if rc != 0:
            raise RuntimeError("zigz_rccl_allreduce_u64 failed (%d)" % rc)
        return out

    def all_reduce_u64_dev(self, ctx, d_words, n):
        """zigz_rccl_allreduce_u64_dev: n u64 words already in HBM, summed over the ranks in place on the context's stream,
        then the deadline wait that belongs to it (zigz_rccl_stream_wait)."""
        from . import _ffi
        s = _ffi.lib.zigz_ctx_get_stream(ctx.h)
        rc = _ffi.lib.zigz_rccl_allreduce_u64_dev(self.h, d_words, n, s)
        if rc == 0:
            rc = _ffi.lib.zigz_rccl_stream_wait(self.h, s)
        if rc != 0:
            raise RuntimeError("zigz_rccl_allreduce_u64_dev failed (%d)" % rc)

    def set_timeout(self, seconds):
        from . import _ffi
        _ffi.lib.zigz_rccl_comm_set_timeout(self.h, float(seconds))

    def close(self):
        if self.h:
            from . import _ffi
            _ffi.lib.zigz_rccl_comm_destroy(self.h)
            self.h = None


def sumcheck_prove_row_sharded_radix(ctx, d_local, n_local, dist, allgather=None):
    """SumcheckProver.prove (src/proofs/sumcheck_prover.zig:26-91) over a table sharded by rows (interleaved), radix
    form, orchestrated in C++ (zigz_dev_sumcheck_prove_sharded): 2-3 exchanges of <= 1024 u64 per proof through the
    all-gather hook, no per-round collective, no Python in the loop.  d_local: this rank's n_local elements in HBM
    (16-byte aligned).  Returns (rounds[2v], point[v], final_eval), identical on every rank and to the unsharded proof."""
    if isinstance(allgather, (ShmComm, RcclComm)):
        return ctx.dev_sumcheck_prove_sharded(d_local, n_local, allgather.rank, allgather.world, allgather.hook, allgather.user)
    cb = allgather or make_allgather(dist)
    return ctx.dev_sumcheck_prove_sharded(d_local, n_local, dist.get_rank(), dist.get_world_size(), cb)


def sumcheck_radix_run(py_ops, n_local, dist, challenges=None, allgather=None, reduced=False):
    """The C++ orchestration of the radix sumcheck (zigz_sumcheck_radix_run) over Python data passes:
    py_ops.block_sums(k) -> 2^k ints, py_ops.fold(k, weights, k_next) -> 2^k_next ints or None, py_ops.read_tail(m) -> m
    ints.  Used by the multi-process CPU tests to drive the exchange logic of the sharded prover without a GPU.
    reduced: the passes return the sums over ALL ranks (they reduce them in a collective of their own, as the RCCL passes of
    zigz_dev_sumcheck_prove_rccl do): zigz_sumcheck_radix_run_reduced, no exchange per stage."""
    import ctypes as C
    from . import _ffi
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    comm_user = None
    if isinstance(allgather, (ShmComm, RcclComm)):
        cb, comm_user, world, rank = allgather.hook, allgather.user, allgather.world, allgather.rank
    else:
        cb = allgather or (make_allgather(dist) if world > 1 else _ffi.ALLGATHER_FN(lambda *a: 1))
    err = []

    def guard(fn):
        def w(*a):
            try:
                fn(*a)
                return 0
            except Exception as e:  # never unwind into C
                err.append(e)
                return 103
        return w

    def c_block_sums(_u, k, out):
        v = py_ops.block_sums(k)
        for i in range(1 << k):
            out[i] = int(v[i])

    def c_fold(_u, k, w, k_next, nxt):
        v = py_ops.fold(k, [int(w[i]) for i in range(1 << k)], k_next)
        for i in range((1 << k_next) if k_next else 0):
            nxt[i] = int(v[i])

    def c_read_tail(_u, m, out):
        v = py_ops.read_tail(m)
        for i in range(m):
            out[i] = int(v[i])

    ops = _ffi.RadixOps(None, _ffi.RADIX_BLOCK_SUMS_FN(guard(c_block_sums)), _ffi.RADIX_FOLD_FN(guard(c_fold)),
                        _ffi.RADIX_READ_TAIL_FN(guard(c_read_tail)))
    nv = (n_local * world).bit_length() - 1
    r = (C.c_uint64 * max(2 * nv, 1))()
    pt = (C.c_uint64 * max(nv, 1))()
    fe = C.c_uint64()
    ch = None
    if challenges is not None:
        ch = (C.c_uint64 * max(nv, 1))(*[int(c) for c in challenges])
    run = _ffi.lib.zigz_sumcheck_radix_run_reduced if reduced else _ffi.lib.zigz_sumcheck_radix_run
    rc = run(C.byref(ops), n_local, rank, world, cb, comm_user, ch, r, pt, C.byref(fe))
    if err:
        raise err[0]
    if rc != 0:
        from . import errors
        raise errors.ZigzError(rc, _ffi.lib.zigz_status_name(rc).decode())
    return np.array(r[: 2 * nv], dtype=np.uint64), np.array(pt[:nv], dtype=np.uint64), fe.value


def sumcheck_prove_row_sharded(ops, local_table, n_global, dist, transcript_factory):
    """SumcheckProver.prove (src/proofs/sumcheck_prover.zig:26-91) over a table sharded by rows (interleaved).
    `local_table`: ops-specific handle of this rank's n_global/G elements.  One all-reduce of 2 words per round.
    Returns (rounds[2v], point[v], final_eval) on every rank -- identical to the unsharded proof."""
    rank, world = dist.get_rank(), dist.get_world_size()
    assert n_global & (n_global - 1) == 0 and world & (world - 1) == 0 and n_global >= 2 * world
    nv = n_global.bit_length() - 1
    lg = world.bit_length() - 1
    tr = transcript_factory()  # fresh transcript per sumcheck (sumcheck_protocol.zig:161)
    rounds, point = [], []
    cur, n_local = local_table, n_global // world
    s = _all_reduce_u64(dist, ops.half_sums(cur, n_local))
    for _ in range(nv - lg):                       # rounds whose bind pairs are rank-local
        s0, s1 = s[0] % P, s[1] % P
        c0, c1 = s0, (s1 - s0) % P                 # roundPolynomial, multilinear.zig:228-229
        rounds += [c0, c1]
        tr.append_field(c0); tr.append_field(c1)
        ch = tr.challenge()
        point.append(ch)
        if n_local > 1:
            if n_local > 2:
                cur, part = ops.bind_sums(cur, n_local, ch)
                s = _all_reduce_u64(dist, part)
            else:
                cur = ops.bind(cur, n_local, ch)
            n_local //= 2
    # n_local == 1 now: global table of `world` elements, element g on rank g
    mine = int(ops.download(cur, 1)[0])
    tail = np.array(_all_gather_u64(dist, mine), dtype=np.uint64)
    if lg:
        t = ops.upload(tail)
        r2, p2, fe = ops.sumcheck_tail(t, world, tr)
        rounds += r2
        point += p2
    else:
        fe = int(tail[0])
    return np.array(rounds, dtype=np.uint64), np.array(point, dtype=np.uint64), int(fe)


class RowShardedMerkle:
    """SimpleMerkleTree.build / open (src/commitments/merkle_tree.zig:283-360) of ONE column sharded by contiguous
    rows: rank r holds values[r*n/G : (r+1)*n/G] (n and G powers of two, n >= G).  `tree_ops` supplies the local
    work: commit(values) -> (root32 bytes, handle), open(handle, index) -> (siblings[h,32], dirs[h], leaf),
    sha3(bytes) -> 32 bytes (GpuTreeOps below = libzigz_hip.so).  Root and openings equal the unsharded tree's."""

    def __init__(self, tree_ops, local_values, n_global, dist):
        self.ops, self.dist = tree_ops, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        assert n_global & (n_global - 1) == 0 and self.world & (self.world - 1) == 0 and n_global >= self.world
        self.n, self.n_local = n_global, n_global // self.world
        assert len(local_values) == self.n_local
        self.h_local = self.n_local.bit_length() - 1
        self.h_top = self.world.bit_length() - 1
        sub_root, self.handle = self.ops.commit(local_values)
        gathered = _all_gather_bytes(dist, bytes(sub_root), 32)                      # exchange: G subtree roots
        level = [g.tobytes() for g in gathered]
        self.top = [level]                                                          # top[0] = subtree roots ... top[-1] = [root]
        while len(level) > 1:
            level = [self.ops.sha3(level[2 * i] + level[2 * i + 1]) for i in range(len(level) // 2)]
            self.top.append(level)
        self.root = level[0]

    @property
    def height(self):
        return self.h_local + self.h_top

    def open(self, index):
        """-> (siblings[height,32] uint8, dirs[height] uint8, leaf value); IndexError like error.IndexOutOfBounds."""
        if not 0 <= index < self.n:
            raise IndexError("IndexOutOfBounds")
        owner, local = divmod(index, self.n_local)
        rec = 8 + 33 * self.h_local
        payload = b""
        if self.rank == owner:
            sib, dirs, leaf = self.ops.open(self.handle, local)
            payload = (np.array([leaf], dtype="<u8").tobytes() + np.ascontiguousarray(sib, dtype=np.uint8).tobytes() +
                       np.ascontiguousarray(dirs, dtype=np.uint8).tobytes())
        chunk = _all_gather_bytes(self.dist, payload, rec)[owner]                   # exchange: the owner's local path
        leaf = int(np.frombuffer(chunk[:8].tobytes(), dtype="<u8")[0])
        siblings = np.zeros((self.height, 32), dtype=np.uint8)
        dirs = np.zeros(self.height, dtype=np.uint8)
        siblings[: self.h_local] = chunk[8:8 + 32 * self.h_local].reshape(self.h_local, 32)
        dirs[: self.h_local] = chunk[8 + 32 * self.h_local:rec]
        pos = owner
        for l in range(self.h_top):                                                 # top levels: known to every rank
            siblings[self.h_local + l] = np.frombuffer(self.top[l][pos ^ 1], dtype=np.uint8)
            dirs[self.h_local + l] = pos & 1
            pos >>= 1
        return siblings, dirs, leaf

    def close(self):
        self.ops.destroy(self.handle)
        self.handle = None


def _all_gather_u64(dist, value):
    import torch
    dev = _dist_device(dist)
    t = torch.tensor([int(value)], dtype=torch.int64, device=dev)
    outs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, t)
    return [int(o.item()) for o in outs]


class GpuEngine:
    """Column engine on libzigz_hip.so: host uint64 columns or (device pointer, stride)."""

    def __init__(self, ctx):
        self.ctx = ctx

    def begin(self, local_cols, ncols, nv):
        from .hip import CommitJob
        if isinstance(local_cols, tuple):
            d_ptr, stride = local_cols
            return CommitJob(self.ctx, d_cols=d_ptr, ncols=ncols, nv=nv, col_stride=stride)
        return CommitJob(self.ctx, cols=np.ascontiguousarray(local_cols, dtype=np.uint64))


class GpuTreeOps:
    """Local subtree work of RowShardedMerkle on libzigz_hip.so."""

    def __init__(self, ctx):
        self.ctx = ctx

    def commit(self, values):
        from .hip import SimpleMerkleTree
        t = SimpleMerkleTree(self.ctx, np.ascontiguousarray(values, dtype=np.uint64))
        return t.getRoot(), t

    def open(self, tree, index):
        o = tree.open(index)
        h = tree.height
        return (np.frombuffer(o["siblings"], dtype=np.uint8).reshape(h, 32), np.frombuffer(o["directions"], dtype=np.uint8),
                o["value"])

    def sha3(self, data):
        from .hip import sha3_256
        return sha3_256(data)

    def destroy(self, tree):
        tree.deinit()


class GpuOps:
    """Row-sharded sumcheck ops on libzigz_hip.so; tables are (device pointer, capacity) pairs."""

    def __init__(self, ctx):
        self.ctx = ctx
        self._bufs = []

    def upload(self, values):
        n = len(values)
        d = self.ctx.dev_alloc(max(n, 4) * 4)
        self._bufs.append(d)
        self.ctx.upload(values, d)
        return d

    def download(self, d, n):
        return self.ctx.download(d, n)

    def half_sums(self, d, n):
        return self.ctx.dev_mle_half_sums(d, n)

    def bind_sums(self, d, n, r):
        out = self.ctx.dev_alloc(max(n // 2, 4) * 4)
        self._bufs.append(out)
        return out, self.ctx.dev_mle_bind_sums(d, n, r, out)

    def bind(self, d, n, r):
        out = self.ctx.dev_alloc(max(n // 2, 4) * 4)
        self._bufs.append(out)
        self.ctx.dev_mle_bind(d, n, r, out)
        return out

    def sumcheck_tail(self, d, n, tr):
        """Finish the last log2(G) rounds on the gathered table, continuing transcript `tr`."""
        rounds, point = [], []
        cur = d
        while n > 1:
            s = self.ctx.dev_mle_half_sums(cur, n)
            c0, c1 = s[0] % P, (s[1] - s[0]) % P
            rounds += [c0, c1]
            tr.append_field(c0); tr.append_field(c1)
            ch = tr.challenge()
            point.append(ch)
            cur = self.bind(cur, n, ch)
            n //= 2
        return rounds, point, int(self.ctx.download(cur, 1)[0])

    def close(self):
        for d in self._bufs:
            self.ctx.dev_free(d)
        self._bufs = []
