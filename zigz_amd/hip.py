"""Python face of the C ABI (include/zigz_hip.h): a Context plus thin wrappers whose names and
argument meanings follow the reference types they stand in for -- Multilinear
(src/poly/multilinear.zig), SumcheckProver (src/proofs/sumcheck_prover.zig), SimpleMerkleTree
(src/commitments/merkle_tree.zig), CommitmentScheme (src/commitments/polynomial_commit.zig),
LassoProver (src/lookups/lasso_prover.zig).  All arithmetic happens in libzigz_hip.so on the GPU.
"""
import ctypes as C

import numpy as np

from . import errors
from ._ffi import BenchResult, KernelStats, lib, u8p, u64p, vp

P = 2013265921
NUM_COLUMNS = 43
# zigz_trace_step (include/zigz_hip.h): one compact record per executed step, 48 bytes
TRACE_STEP_DTYPE = np.dtype([("pc", "<u8"), ("rd_value", "<u8"), ("mem_addr", "<u8"), ("mem_value", "<u8"), ("imm", "<i8"),
                             ("opcode", "u1"), ("rd", "u1"), ("rs1", "u1"), ("rs2", "u1"), ("funct3", "u1"), ("funct7", "u1"),
                             ("wr_reg", "u1"), ("mem_is_read", "u1")])
assert TRACE_STEP_DTYPE.itemsize == 48
# the 32-byte record and its side list (include/zigz_hip.h: zigz_trace_step32 / zigz_mem_access)
TRACE_STEP32_DTYPE = np.dtype([("pc", "<u8"), ("rd_value", "<u8"), ("imm", "<i4"), ("mem_index", "<u4"), ("opcode", "u1"), ("rd", "u1"),
                               ("rs1", "u1"), ("rs2", "u1"), ("funct3", "u1"), ("funct7", "u1"), ("wr_reg", "u1"), ("mem_is_read", "u1")])
MEM_ACCESS_DTYPE = np.dtype([("addr", "<u8"), ("value", "<u8")])
assert TRACE_STEP32_DTYPE.itemsize == 32 and MEM_ACCESS_DTYPE.itemsize == 16
NO_MEM_ACCESS = 0xFFFFFFFF


def compact_steps32(steps, has_access):
    """48-byte records -> (32-byte records, side list): has_access[i] says whether step i has a memory access at all (a LOAD /
    STORE; its address and value may well be 0).  imm must fit 32 bits (it does in every RV64IM format)."""
    steps = np.ascontiguousarray(steps, dtype=TRACE_STEP_DTYPE)
    has_access = np.asarray(has_access, dtype=bool)
    out = np.zeros(len(steps), dtype=TRACE_STEP32_DTYPE)
    for f in ("pc", "rd_value", "opcode", "rd", "rs1", "rs2", "funct3", "funct7", "wr_reg", "mem_is_read"):
        out[f] = steps[f]
    assert np.all(steps["imm"] == steps["imm"].astype(np.int32)), "an immediate does not fit 32 bits"
    out["imm"] = steps["imm"].astype(np.int32)
    idx = np.flatnonzero(has_access)
    out["mem_index"] = NO_MEM_ACCESS
    out["mem_index"][idx] = np.arange(len(idx), dtype=np.uint32)
    mem = np.zeros(len(idx), dtype=MEM_ACCESS_DTYPE)
    mem["addr"], mem["value"] = steps["mem_addr"][idx], steps["mem_value"][idx]
    return out, mem


# the 16-byte record and the code table (include/zigz_hip.h: zigz_trace_step16 / zigz_code_entry)
TRACE_STEP16_DTYPE = np.dtype([("pc_word", "<u4"), ("mem_wr", "<u4"), ("rd_value", "<u8")])
CODE_ENTRY_DTYPE = np.dtype([("imm", "<i4"), ("opcode", "u1"), ("rd", "u1"), ("rs1", "u1"), ("rs2", "u1"), ("funct3", "u1"), ("funct7", "u1"),
                             ("reserved0", "u1"), ("reserved1", "u1")])
assert TRACE_STEP16_DTYPE.itemsize == 16 and CODE_ENTRY_DTYPE.itemsize == 12
NO_MEM_ACCESS16 = (1 << 27) - 1


def compact_steps16(steps, has_access):
    """48-byte records -> (16-byte records, side list, code_base, code table), or None when the trace does not fit the form (a pc off
    the 4-byte grid or 2^32 past the lowest one, one pc with two different decodings, 2^27 - 1 or more accesses)."""
    steps = np.ascontiguousarray(steps, dtype=TRACE_STEP_DTYPE)
    s32, mem = compact_steps32(steps, has_access)
    if len(steps) == 0 or len(mem) >= NO_MEM_ACCESS16:
        return None
    base = int(steps["pc"].min())
    off = steps["pc"] - np.uint64(base)
    if base & 3 or int(off.max()) >= 1 << 32 or np.any(off & np.uint64(3)) or np.any(steps["wr_reg"] >= 32) or np.any(steps["mem_is_read"] > 1):
        return None
    idx = (off >> np.uint64(2)).astype(np.int64)
    code = np.zeros(int(idx.max()) + 1, dtype=CODE_ENTRY_DTYPE)
    fields = ("imm", "opcode", "rd", "rs1", "rs2", "funct3", "funct7")
    first = {}
    for i in np.unique(idx, return_index=True)[1]:
        first[int(idx[i])] = int(i)
    for ci, i in first.items():
        for f in fields:
            code[f][ci] = s32[f][i]
    for f in fields:  # every step must agree with the entry of its pc
        if np.any(code[f][idx] != s32[f]):
            return None
    out = np.zeros(len(steps), dtype=TRACE_STEP16_DTYPE)
    out["pc_word"] = off.astype(np.uint32) | (steps["mem_is_read"].astype(np.uint32) & 1)
    mi = np.where(s32["mem_index"] == NO_MEM_ACCESS, np.uint32(NO_MEM_ACCESS16), s32["mem_index"]).astype(np.uint32)
    out["mem_wr"] = mi | (steps["wr_reg"].astype(np.uint32) << 27)
    out["rd_value"] = steps["rd_value"]
    return out, mem, base, code


def _name(code):
    return lib.zigz_status_name(code).decode()


def _u64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if a.size == 0:
        a = np.zeros(1, dtype=np.uint64)
    return a, a.ctypes.data_as(u64p)


def _out_u64(n):
    a = np.zeros(max(int(n), 1), dtype=np.uint64)
    return a, a.ctypes.data_as(u64p)


def _out_u8(n):
    a = np.zeros(max(int(n), 1), dtype=np.uint8)
    return a, a.ctypes.data_as(u8p)


def device_count():
    n = C.c_int(0)
    lib.zigz_device_count(C.byref(n))
    return n.value


class Context:
    """zigz_ctx: one HIP device + stream + workspace.  Raises ZigzError(NoDevice) without a gfx950 GPU."""

    def __init__(self, device=0, _borrowed=None):
        self.owned = _borrowed is None
        if _borrowed is not None:  # a context somebody else owns (a GPU slot of host.Slots): never destroyed from here
            self.h = _borrowed
            return
        h = vp()
        rc = lib.zigz_ctx_create(device, C.byref(h))
        if rc != 0:
            raise errors.ZigzError(rc, _name(rc), "zigz_ctx_create: a gfx950 (MI355X) device is required; no CPU fallback")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            if self.owned:
                lib.zigz_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def check(self, rc):
        if rc != 0:
            raise errors.ZigzError(rc, _name(rc), lib.zigz_last_error(self.h).decode(errors="replace"))

    # ---- stream / memory
    def set_stream(self, hip_stream):
        self.check(lib.zigz_ctx_set_stream(self.h, vp(hip_stream) if hip_stream else None))

    def synchronize(self):
        self.check(lib.zigz_ctx_synchronize(self.h))

    def dev_alloc(self, nbytes):
        p = vp()
        self.check(lib.zigz_dev_alloc(self.h, nbytes, C.byref(p)))
        return p.value

    def dev_free(self, ptr):
        self.check(lib.zigz_dev_free(self.h, vp(ptr)))

    def upload(self, values, d_ptr):
        a, ap = _u64(values)
        self.check(lib.zigz_dev_upload_u64(self.h, ap, len(values), vp(d_ptr)))

    def reduce_upload(self, raw_u64, d_ptr):
        a, ap = _u64(raw_u64)
        self.check(lib.zigz_dev_reduce_u64(self.h, ap, len(raw_u64), vp(d_ptr)))

    def witness_from_rows(self, rows, nv, d_cols, stride):
        """WitnessGenerator.generate on the device from packed trace rows [num_steps, 43] (raw u64)."""
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        self.check(lib.zigz_dev_witness_from_rows(self.h, rows.ctypes.data_as(u64p), rows.shape[0], nv, vp(d_cols), stride))

    def witness_from_steps(self, steps, nv, d_cols, stride, initial_regs=None):
        """WitnessGenerator.generate on the device from compact records (structured array of TRACE_STEP_DTYPE)."""
        steps = np.ascontiguousarray(steps, dtype=TRACE_STEP_DTYPE)
        ir = None
        if initial_regs is not None:
            ira = np.ascontiguousarray(initial_regs, dtype=np.uint64)
            assert ira.size == 32
            ir = ira.ctypes.data_as(u64p)
        self.check(lib.zigz_dev_witness_from_steps(self.h, vp(steps.ctypes.data), steps.shape[0], nv, ir, vp(d_cols), stride))

    def witness_from_steps32(self, steps32, mem, nv, d_cols, stride, initial_regs=None):
        """the same from the 32-byte records + side list of memory accesses (zigz_dev_witness_from_steps32)"""
        steps32 = np.ascontiguousarray(steps32, dtype=TRACE_STEP32_DTYPE)
        mem = np.ascontiguousarray(mem, dtype=MEM_ACCESS_DTYPE)
        ir = None
        if initial_regs is not None:
            ira = np.ascontiguousarray(initial_regs, dtype=np.uint64)
            assert ira.size == 32
            ir = ira.ctypes.data_as(u64p)
        self.check(lib.zigz_dev_witness_from_steps32(self.h, vp(steps32.ctypes.data), steps32.shape[0], vp(mem.ctypes.data) if len(mem) else None,
                                                     len(mem), nv, ir, vp(d_cols), stride))

    def witness_from_steps16(self, steps16, mem, code_base, code, nv, d_cols, stride, initial_regs=None):
        """the same from the 16-byte records + side list + code table (zigz_dev_witness_from_steps16)"""
        steps16 = np.ascontiguousarray(steps16, dtype=TRACE_STEP16_DTYPE)
        mem = np.ascontiguousarray(mem, dtype=MEM_ACCESS_DTYPE)
        code = np.ascontiguousarray(code, dtype=CODE_ENTRY_DTYPE)
        ir = None
        if initial_regs is not None:
            ira = np.ascontiguousarray(initial_regs, dtype=np.uint64)
            assert ira.size == 32
            ir = ira.ctypes.data_as(u64p)
        self.check(lib.zigz_dev_witness_from_steps16(self.h, vp(steps16.ctypes.data), steps16.shape[0], vp(mem.ctypes.data) if len(mem) else None,
                                                     len(mem), int(code_base), vp(code.ctypes.data) if len(code) else None, len(code), nv, ir,
                                                     vp(d_cols), stride))

    def download(self, d_ptr, n):
        o, op = _out_u64(n)
        self.check(lib.zigz_dev_download_u64(self.h, vp(d_ptr), n, op))
        return o[:n]

    def enable_timing(self, on=True):
        self.check(lib.zigz_ctx_enable_timing(self.h, 1 if on else 0))

    def set_option(self, name, value):
        self.check(lib.zigz_ctx_set_option(self.h, name.encode(), int(value)))

    def release_workspaces(self):
        """give the context's workspaces back to the device (they regrow on demand)"""
        self.check(lib.zigz_ctx_release_workspaces(self.h))

    def mem_info(self):
        """(free, total) bytes of HBM on this context's device"""
        f, t = C.c_size_t(), C.c_size_t()
        self.check(lib.zigz_dev_mem_info(self.h, C.byref(f), C.byref(t)))
        return f.value, t.value

    def get_option(self, name):
        v = C.c_int64()
        self.check(lib.zigz_ctx_get_option(self.h, name.encode(), C.byref(v)))
        return v.value & 0xFFFFFFFFFFFFFFFF

    def stats(self):
        s = KernelStats()
        self.check(lib.zigz_ctx_get_stats(self.h, C.byref(s)))
        return {f: getattr(s, f) for f, _ in KernelStats._fields_}

    def set_epoch(self, owner=None):
        """zigz_ctx_set_epoch: a new epoch of this context's (owner None) or the adoption of another context's: what the
        times of launch_log() count from."""
        self.check(lib.zigz_ctx_set_epoch(self.h, (owner or self).h))

    def launch_log(self):
        """zigz_ctx_launch_log: [(class, permutations, start_us, end_us)] of the timed launches of the last commit job."""
        from ._ffi import LaunchRec
        buf = (LaunchRec * 80)()
        n = C.c_size_t()
        self.check(lib.zigz_ctx_launch_log(self.h, buf, 80, C.byref(n)))
        return [(buf[i].cls, buf[i].perms, buf[i].start_us, buf[i].end_us) for i in range(min(n.value, 80))]

    def bench_kernel(self, kernel, nv, ncols, iters=10, cold=True):
        """zigz_bench_kernel: per-launch kernel durations of one hot kernel on a synthetic resident table."""
        r = BenchResult()
        self.check(lib.zigz_bench_kernel(self.h, kernel.encode(), nv, ncols, iters, 1 if cold else 0, C.byref(r)))
        return {f: getattr(r, f) for f, _ in BenchResult._fields_}

    # ---- Multilinear(F) seams, host buffers
    def mle_bind(self, evals, r):
        """partialEval(self, r), multilinear.zig:154"""
        a, ap = _u64(evals)
        o, op = _out_u64(len(evals) // 2)
        self.check(lib.zigz_mle_bind(self.h, ap, len(evals), int(r), op))
        return o[: len(evals) // 2]

    def mle_round_poly(self, evals):
        """roundPolynomial(self), multilinear.zig:205"""
        a, ap = _u64(evals)
        o, op = _out_u64(2)
        self.check(lib.zigz_mle_round_poly(self.h, ap, len(evals), op))
        return [int(o[0]), int(o[1])]

    def mle_sum(self, evals):
        """sumOverHypercube(self), multilinear.zig:188"""
        a, ap = _u64(evals)
        out = C.c_uint64()
        self.check(lib.zigz_mle_sum(self.h, ap, len(evals), C.byref(out)))
        return out.value

    def mle_eval(self, evals, point):
        """eval(self, point), multilinear.zig:110"""
        a, ap = _u64(evals)
        q, qp = _u64(point)
        out = C.c_uint64()
        self.check(lib.zigz_mle_eval(self.h, ap, len(evals), qp, len(point), C.byref(out)))
        return out.value

    # ---- SumcheckProver(F)
    def sumcheck_prove(self, evals, challenges=None):
        """prove(poly) / proveInteractive(poly, challenges), sumcheck_prover.zig:26,97.
        Returns (rounds[2v], final_point[v], final_eval)."""
        a, ap = _u64(evals)
        n = len(evals)
        nv = max(n.bit_length() - 1, 0)
        r, rp = _out_u64(2 * nv)
        pt, ptp = _out_u64(nv)
        fe = C.c_uint64()
        if challenges is None:
            self.check(lib.zigz_sumcheck_prove(self.h, ap, n, rp, ptp, C.byref(fe)))
        else:
            c, cp = _u64(challenges)
            self.check(lib.zigz_sumcheck_prove_interactive(self.h, ap, n, cp, len(challenges), rp, ptp, C.byref(fe)))
        return r[: 2 * nv].copy(), pt[:nv].copy(), fe.value

    # ---- device-resident variants (packed u32 canonical in HBM)
    def dev_mle_bind(self, d_in, n, r, d_out):
        self.check(lib.zigz_dev_mle_bind(self.h, vp(d_in), n, int(r), vp(d_out)))

    def dev_mle_bind_sums(self, d_in, n, r, d_out):
        o, op = _out_u64(2)
        self.check(lib.zigz_dev_mle_bind_sums(self.h, vp(d_in), n, int(r), vp(d_out), op))
        return [int(o[0]), int(o[1])]

    def dev_mle_half_sums(self, d_in, n):
        o, op = _out_u64(2)
        self.check(lib.zigz_dev_mle_half_sums(self.h, vp(d_in), n, op))
        return [int(o[0]), int(o[1])]

    def dev_mle_eval(self, d_in, n, point):
        q, qp = _u64(point)
        out = C.c_uint64()
        self.check(lib.zigz_dev_mle_eval(self.h, vp(d_in), n, qp, len(point), C.byref(out)))
        return out.value

    def dev_sumcheck_prove(self, d_in, n, challenges=None, d_scratch=None):
        nv = max(n.bit_length() - 1, 0)
        r, rp = _out_u64(2 * nv)
        pt, ptp = _out_u64(nv)
        fe = C.c_uint64()
        cp = None
        if challenges is not None:
            c, cp = _u64(challenges)
        self.check(lib.zigz_dev_sumcheck_prove(self.h, vp(d_in), n, vp(d_scratch) if d_scratch else None, cp, rp, ptp,
                                               C.byref(fe)))
        return r[: 2 * nv].copy(), pt[:nv].copy(), fe.value

    def dev_sumcheck_prove_sharded(self, d_local, n_local, rank, world, allgather, user=None):
        """zigz_dev_sumcheck_prove_sharded: SumcheckProver.prove of ONE table sharded by rows (element i on rank i mod world
        at local index i // world).  `allgather`: an _ffi.ALLGATHER_FN (shard.make_allgather(dist))."""
        nv = (n_local * world).bit_length() - 1
        r, rp = _out_u64(2 * nv)
        pt, ptp = _out_u64(nv)
        fe = C.c_uint64()
        self.check(lib.zigz_dev_sumcheck_prove_sharded(self.h, vp(d_local), n_local, rank, world, allgather, user, rp, ptp,
                                                       C.byref(fe)))
        return r[: 2 * nv].copy(), pt[:nv].copy(), fe.value

    def dev_sumcheck_prove_rccl(self, d_local, n_local, comm):
        """zigz_dev_sumcheck_prove_rccl: the same proof with a shard.RcclComm as the transport -- the partial block sums of
        every radix stage are all-reduced in HBM on this context's stream, the tail all-gathered through the comm."""
        nv = (n_local * comm.world).bit_length() - 1
        r, rp = _out_u64(2 * nv)
        pt, ptp = _out_u64(nv)
        fe = C.c_uint64()
        self.check(lib.zigz_dev_sumcheck_prove_rccl(self.h, vp(d_local), n_local, comm.h, rp, ptp, C.byref(fe)))
        return r[: 2 * nv].copy(), pt[:nv].copy(), fe.value

    # ---- Lasso
    def lasso_fingerprints(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        n, w = rows.shape
        o, op = _out_u64(n)
        self.check(lib.zigz_lasso_fingerprints(self.h, rows.ctypes.data_as(u64p), n, w, op))
        return o[:n]

    def lasso_prove(self, table, queries, n_in=2, n_out=1, mapping=None):
        """LassoProver.prove / proveWithMapping, lasso_prover.zig:103,179"""
        w = n_in + n_out
        t, tp = _u64(np.asarray(table, dtype=np.uint64).reshape(-1))
        q, qp = _u64(np.asarray(queries, dtype=np.uint64).reshape(-1))
        rows = 0 if len(table) == 0 else len(np.asarray(table).reshape(-1)) // w
        nq = len(queries)
        npad = 1
        while npad < max(nq, 1):
            npad <<= 1
        nvmax = npad.bit_length() - 1
        r, rp = _out_u64(2 * nvmax)
        pt, ptp = _out_u64(nvmax)
        fe = C.c_uint64()
        nv = C.c_size_t()
        qc, qcp = _out_u8(32)
        tc, tcp = _out_u8(32)
        if mapping is None:
            self.check(lib.zigz_lasso_prove(self.h, tp, rows, qp, nq, n_in, n_out, C.byref(nv), rp, ptp, C.byref(fe),
                                            qcp, tcp))
        else:
            m, mp = _u64(mapping)
            self.check(lib.zigz_lasso_prove_with_mapping(self.h, tp, rows, qp, nq, n_in, n_out, mp, len(mapping),
                                                         C.byref(nv), rp, ptp, C.byref(fe), qcp, tcp))
        v = nv.value
        return dict(nv=v, rounds=r[: 2 * v].copy(), point=pt[:v].copy(), final_eval=fe.value,
                    query_commit=qc.tobytes(), table_commit=tc.tobytes())


class SimpleMerkleTree:
    """SimpleMerkleTree(F, SHA3Hasher), merkle_tree.zig:273 -- all levels resident in HBM."""

    def __init__(self, ctx, values):
        self.ctx = ctx
        a, ap = _u64(values)
        root, rp = _out_u8(32)
        h = C.c_size_t()
        t = vp()
        ctx.check(lib.zigz_merkle_commit(ctx.h, ap, len(values), rp, C.byref(h), C.byref(t)))
        self.root_hash = root.tobytes()
        self.height = h.value
        self.n_values = len(values)
        self.t = t

    @classmethod
    def build(cls, ctx, values):
        return cls(ctx, values)

    def getRoot(self):
        return self.root_hash

    def open(self, index):
        sib, sp = _out_u8(32 * self.height)
        dirs, dp = _out_u8(self.height)
        leaf = C.c_uint64()
        self.ctx.check(lib.zigz_merkle_open(self.ctx.h, self.t, index, sp, dp, C.byref(leaf)))
        return dict(index=index, value=leaf.value, siblings=sib[: 32 * self.height].tobytes(),
                    directions=dirs[: self.height].tobytes())

    def deinit(self):
        if self.t:
            lib.zigz_merkle_destroy(self.ctx.h, self.t)
            self.t = None

    def __del__(self):
        try:
            self.deinit()
        except Exception:
            pass


class CommitmentScheme:
    """CommitmentSchemeSHA3(F), polynomial_commit.zig:58-185"""

    @staticmethod
    def commit(ctx, evals):
        tree = SimpleMerkleTree(ctx, evals)
        return tree.root_hash, tree

    @staticmethod
    def open(ctx, evals, tree, point):
        nv = len(point)
        a, ap = (None, None) if evals is None else _u64(evals)
        n = tree.n_values if evals is None else len(evals)
        q, qp = _u64(point)
        sib, sp = _out_u8(32 * nv)
        dirs, dp = _out_u8(nv)
        val, idx, leaf = C.c_uint64(), C.c_uint64(), C.c_uint64()
        ctx.check(lib.zigz_commit_open(ctx.h, ap, n, tree.t, qp, nv, C.byref(val), C.byref(idx), sp, dp, C.byref(leaf)))
        return dict(value=val.value, index=idx.value, leaf=leaf.value, siblings=sib[: 32 * nv].tobytes(),
                    directions=dirs[:nv].tobytes())


class CommitJob:
    """Prover.generateCommitments split at its transcript dependencies (prover.zig:366-467)."""

    def __init__(self, ctx, cols=None, d_cols=None, ncols=NUM_COLUMNS, nv=None, col_stride=None, d_cols_list=None):
        self.ctx = ctx
        j = vp()
        if d_cols_list is not None:  # zigz_commit_begin_batch: several proofs' resident columns in one job
            k = len(d_cols_list)
            arr = (vp * k)(*[vp(d) for d in d_cols_list])
            ctx.check(lib.zigz_commit_begin_batch(ctx.h, arr, k, ncols, col_stride or (1 << nv), nv, C.byref(j)))
            ncols = ncols * k
        elif cols is not None:
            cols = np.ascontiguousarray(cols, dtype=np.uint64)
            ncols, N = cols.shape
            nv = N.bit_length() - 1
            self._keep = cols
            ctx.check(lib.zigz_commit_begin(ctx.h, cols.ctypes.data_as(u64p), ncols, N, nv, C.byref(j)))
        else:
            ctx.check(lib.zigz_commit_begin_dev(ctx.h, vp(d_cols), ncols, col_stride or (1 << nv), nv, C.byref(j)))
        self.j, self.ncols, self.nv = j, ncols, nv

    def roots(self):
        r, rp = _out_u8(self.ncols * 32)
        self.ctx.check(lib.zigz_commit_roots(self.j, rp))
        return r[: self.ncols * 32].reshape(self.ncols, 32)

    def open_all(self, points):
        nc, nv = self.ncols, self.nv
        p, pp = _u64(np.asarray(points, dtype=np.uint64).reshape(-1))
        values, vpp = _out_u64(nc)
        idx, ip = _out_u64(nc)
        leaves, lp = _out_u64(nc)
        sib, sp = _out_u8(nc * nv * 32)
        dirs, dp = _out_u8(nc * nv)
        self.ctx.check(lib.zigz_commit_open_all(self.j, pp, vpp, ip, lp, sp, dp))
        return dict(values=values[:nc], indices=idx[:nc], leaves=leaves[:nc],
                    siblings=sib[: nc * nv * 32].reshape(nc, nv, 32), dirs=dirs[: nc * nv].reshape(nc, nv))

    def tree(self):
        """(device address, bytes per column) of the built trees, internal node form -- for node-by-node comparisons."""
        d, n = vp(), C.c_size_t()
        self.ctx.check(lib.zigz_commit_job_tree(self.j, C.byref(d), C.byref(n)))
        return d.value, n.value

    def end(self):
        if self.j:
            lib.zigz_commit_end(self.j)
            self.j = None

    def __del__(self):
        try:
            self.end()
        except Exception:
            pass


class Transcript:
    """FiatShamirTranscript (src/core/hash.zig:255-324), BabyBear challenges."""

    def __init__(self):
        self.h = lib.zigz_transcript_new()

    def __del__(self):
        if getattr(self, "h", None):
            lib.zigz_transcript_free(self.h)
            self.h = None

    def append_bytes(self, b):
        lib.zigz_transcript_append_bytes(self.h, bytes(b), len(b))

    def append_field(self, v):
        lib.zigz_transcript_append_field(self.h, int(v))

    def append_tagged_counter(self, tag, start, count):
        lib.zigz_transcript_append_tagged_counter(self.h, bytes(tag), len(tag), start, count)

    def challenge(self):
        return lib.zigz_transcript_challenge(self.h)


def sha3_256(b):
    o, op = _out_u8(32)
    lib.zigz_sha3_256(bytes(b), len(b), op)
    return o.tobytes()


def sha256(b):
    o, op = _out_u8(32)
    lib.zigz_sha256(bytes(b), len(b), op)
    return o.tobytes()
