"""ctypes face of libzigz_host.so (include/zigz_host.h): the C++ mirror of the Zig host --
Prover.prove / Verifier.verify / BinarySerializer / VMState / WitnessGenerator -- on top of the C ABI.
Host-only entry points (VM, witness, verifier, serializer) run without a GPU; everything that proves
needs a Context (gfx950)."""
import ctypes as C
import os

import numpy as np

from . import errors
from ._ffi import u8p, u64p, vp

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libzigz_host.so")
if not os.path.exists(LIB_PATH):
    raise ImportError(f"{LIB_PATH} is missing: build it with `python -m zigz_amd.build`")
lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)

szp = C.POINTER(C.c_size_t)
SIGNATURES = {
    "zigzh_last_error": (C.c_char_p, []),
    "zigzh_free": (None, [vp]),
    "zigzh_last_timings": (None, [C.POINTER(C.c_double)]),
    "zigzh_prove": (C.c_int, [vp, C.c_char_p, C.c_size_t, C.c_uint64, u64p, C.c_size_t, C.c_int, C.c_size_t, u64p,
                              C.c_size_t, C.POINTER(u8p), szp, szp]),
    "zigzh_verify": (C.c_int, [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]),
    "zigzh_reserialize": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(u8p), szp]),
    "zigzh_execute": (C.c_int, [C.c_char_p, C.c_size_t, C.c_uint64, u64p, C.c_size_t, C.c_int, C.c_size_t, u64p,
                                C.c_size_t, C.POINTER(vp)]),
    "zigzh_trace_free": (None, [vp]),
    "zigzh_trace_num_steps": (C.c_size_t, [vp]),
    "zigzh_trace_num_vars": (C.c_size_t, [vp]),
    "zigzh_trace_num_lookups": (C.c_size_t, [vp]),
    "zigzh_trace_rows": (u64p, [vp]),
    "zigzh_trace_steps": (vp, [vp]),
    "zigzh_trace_initial_regs": (u64p, [vp]),
    "zigzh_trace_pin": (C.c_int, [vp, vp]),
    "zigzh_trace_upload_form": (C.c_int, [vp, C.POINTER(C.c_size_t)]),
    "zigzh_commit_path_repeat": (C.c_int, [vp, vp, C.c_size_t, C.c_size_t, u64p, C.POINTER(C.c_int64), C.c_size_t]),
    "zigzh_trace_witness": (C.c_int, [vp, u64p]),
    "zigzh_trace_witness_dev": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "zigzh_trace_witness_dev_async": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "zigzh_prove_trace": (C.c_int, [vp, vp, vp, C.c_size_t, C.c_int, C.POINTER(u8p), szp]),
    "zigzh_slots_create": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(vp)]),
    "zigzh_slots_destroy": (None, [vp]),
    "zigzh_slots_set_batching": (C.c_int, [vp, C.c_uint, C.c_double, C.c_size_t]),
    "zigzh_slots_size": (C.c_size_t, [vp]),
    "zigzh_slots_ctx": (vp, [vp, C.c_size_t]),
    "zigzh_slots_acquire": (vp, [vp]),
    "zigzh_slots_release": (None, [vp, vp]),
    "zigzh_prove_trace_slots": (C.c_int, [vp, vp, vp, C.c_size_t, C.POINTER(u8p), szp, vp, vp, C.c_size_t, szp]),
    "zigzh_prove_trace_slots_repeat": (C.c_int, [vp, vp, vp, C.c_size_t, C.c_size_t, C.POINTER(u8p), szp, vp, C.POINTER(C.c_double)]),
    "zigzh_stats_add": (None, [vp, vp]),
    "zigzh_prove_trace_sharded": (C.c_int, [vp, vp, vp, C.c_size_t, C.c_int, C.c_int, vp, vp, C.POINTER(u8p), szp]),
    "zigzh_vm_run": (C.c_int, [C.c_char_p, C.c_size_t, C.c_uint64, C.c_size_t, u64p, u64p, szp]),
    "zigzh_sumcheck_prove_bytes": (C.c_int, [vp, u64p, C.c_size_t, u8p, szp]),
    "zigzh_lasso_prove_table": (C.c_int, [vp, C.c_int, C.c_size_t, u64p, C.c_size_t, u64p, C.c_size_t, u8p, szp, u8p,
                                          u8p, szp]),
    "zigzh_commit_open_verify": (C.c_int, [vp, u64p, C.c_size_t, u64p, C.c_size_t, u8p, u64p, u64p, C.POINTER(C.c_int)]),
}
for _n, (_r, _a) in SIGNATURES.items():
    _f = getattr(lib, _n)
    _f.restype = _r
    _f.argtypes = _a

HOST_ERROR_NAMES = {19: "UnimplementedInstruction", 20: "UnimplementedSYSTEM", 21: "InvalidOP32", 22: "InvalidLoadFunct3",
                    23: "InvalidStoreFunct3", 24: "InvalidBranchFunct3", 25: "ProgramHashMismatch", 26: "InvalidMagicNumber",
                    27: "UnsupportedVersion", 28: "FieldMismatch", 29: "InvalidData", 30: "MaxStepsExceeded", 31: "VMHalted"}
VERIFICATION_RESULTS = ["Accept", "RejectInvalidSumcheck", "RejectInvalidLookup", "RejectInvalidCommitment",
                        "RejectInvalidPublicIO"]


def _check(rc):
    if rc != 0:
        msg = lib.zigzh_last_error().decode(errors="replace")
        name = msg.split(":")[0].replace("error.", "") if msg.startswith("error.") else HOST_ERROR_NAMES.get(rc, "Error")
        raise errors.ZigzError(rc, name, msg)


def _u64(a):
    a = np.ascontiguousarray(a if a is not None else [], dtype=np.uint64)
    if a.size == 0:
        a = np.zeros(1, dtype=np.uint64)
    return a, a.ctypes.data_as(u64p)


def _take(ptr, n):
    try:
        return C.string_at(ptr, n.value)
    finally:
        lib.zigzh_free(ptr)


TIMING_NAMES = ["commit_begin", "sumcheck_transcript", "lasso_transcript", "wait_roots", "roots_challenges", "open_all",
                "packaging", "serialize", "wait_slot", "in_slot"]


def last_timings():
    """Per-phase wall-clock seconds of the last Trace.prove on this thread."""
    a = (C.c_double * 10)()
    lib.zigzh_last_timings(a)
    return dict(zip(TIMING_NAMES, list(a)))


def prove(ctx, program, entry_pc=0x1000, initial_regs=None, max_steps=1 << 20, inputs=None):
    """Prover(F).prove(...) then BinarySerializer.serialize -> (proof bytes, num_steps)."""
    ir, irp = _u64(initial_regs)
    inp, inpp = _u64(inputs)
    out, n, ns = u8p(), C.c_size_t(), C.c_size_t()
    _check(lib.zigzh_prove(ctx.h, bytes(program), len(program), entry_pc, irp,
                           0 if initial_regs is None else len(initial_regs), 0 if initial_regs is None else 1, max_steps,
                           inpp if inputs is not None else None, 0 if inputs is None else len(inputs),
                           C.byref(out), C.byref(n), C.byref(ns)))
    return _take(out, n), ns.value


def verify(proof, program):
    """BinarySerializer.deserialize + Verifier.verify -> VerificationResult name (host only)."""
    res = C.c_int(-1)
    _check(lib.zigzh_verify(bytes(proof), len(proof), bytes(program), len(program), C.byref(res)))
    return VERIFICATION_RESULTS[res.value]


def reserialize(proof):
    out, n = u8p(), C.c_size_t()
    _check(lib.zigzh_reserialize(bytes(proof), len(proof), C.byref(out), C.byref(n)))
    return _take(out, n)


class Trace:
    """An executed program ([1/6] of Prover.prove): packed trace rows + public IO."""

    def __init__(self, program, entry_pc=0x1000, initial_regs=None, max_steps=1 << 20, inputs=None):
        ir, irp = _u64(initial_regs)
        inp, inpp = _u64(inputs)
        h = vp()
        _check(lib.zigzh_execute(bytes(program), len(program), entry_pc, irp,
                                 0 if initial_regs is None else len(initial_regs), 0 if initial_regs is None else 1,
                                 max_steps, inpp if inputs is not None else None, 0 if inputs is None else len(inputs),
                                 C.byref(h)))
        self.h = h
        self.num_steps = lib.zigzh_trace_num_steps(h)
        self.num_vars = lib.zigzh_trace_num_vars(h)
        self.num_lookups = lib.zigzh_trace_num_lookups(h)

    def __del__(self):
        if getattr(self, "h", None):
            lib.zigzh_trace_free(self.h)
            self.h = None

    def rows(self):
        p = lib.zigzh_trace_rows(self.h)
        return np.ctypeslib.as_array(p, shape=(self.num_steps, 43)).copy()

    def steps(self):
        """The compact records (numpy structured array, dtype hip.TRACE_STEP_DTYPE) and the initial register file."""
        from .hip import TRACE_STEP_DTYPE
        n = self.num_steps
        buf = (C.c_uint8 * (n * 48)).from_address(lib.zigzh_trace_steps(self.h))
        regs = np.ctypeslib.as_array(lib.zigzh_trace_initial_regs(self.h), shape=(32,)).copy()
        return np.frombuffer(buf, dtype=TRACE_STEP_DTYPE, count=n).copy(), regs

    def pin(self, ctx):
        """Page-lock the records: repeated witness_to_device uploads then run at PCIe rate."""
        _check(lib.zigzh_trace_pin(self.h, ctx.h))

    def upload_form(self):
        """(bytes per step of the record a service uploads for this trace: 16 / 32 / 48, bytes that cross PCIe per proof)"""
        b = C.c_size_t()
        return int(lib.zigzh_trace_upload_form(self.h, C.byref(b))), int(b.value)

    def witness(self):
        """WitnessGenerator.generate: [43, 2^nv] canonical uint64 (host)."""
        cols = np.zeros((43, 1 << self.num_vars), dtype=np.uint64)
        _check(lib.zigzh_trace_witness(self.h, cols.ctypes.data_as(u64p)))
        return cols

    def witness_to_device(self, ctx, d_cols, stride, wait=True):
        """Upload the compact records and build the 43 columns in HBM.  wait=False: enqueue only (pinned traces)."""
        if wait:
            _check(lib.zigzh_trace_witness_dev(self.h, ctx.h, vp(d_cols), stride))
        else:
            _check(lib.zigzh_trace_witness_dev_async(self.h, ctx.h, vp(d_cols), stride))

    def prove(self, ctx, d_cols=None, stride=0, want_bytes=True):
        """want_bytes: True -> proof bytes; False -> None (proof struct only);
        "borrow" -> BorrowedProof (overlapped serialisation into a library-owned buffer, no copy)."""
        out, n = u8p(), C.c_size_t()
        mode = 2 if want_bytes == "borrow" else (1 if want_bytes else 0)
        _check(lib.zigzh_prove_trace(self.h, ctx.h, vp(d_cols) if d_cols else None, stride, mode, C.byref(out), C.byref(n)))
        if mode == 2:
            return BorrowedProof(out, n.value)
        return _take(out, n) if want_bytes else None

    def prove_slots(self, slots, d_cols=None, stride=0, want_log=False):
        """zigzh_prove_trace_slots: the proof of a thread of a proving service -- the transcript of steps 4-5 holding nothing
        on the GPU, then a GPU slot of `slots` for begin -> roots -> challenges -> open_all -> end.  d_cols None: the compact
        records are uploaded and expanded inside the slot (pin the trace first).  Returns (BorrowedProof, stats of the
        context the proof ran on, its launch log in timing mode if want_log else None)."""
        from ._ffi import KernelStats, LaunchRec
        out, n = u8p(), C.c_size_t()
        st = KernelStats()
        log = (LaunchRec * 80)() if want_log else None
        ln = C.c_size_t()
        _check(lib.zigzh_prove_trace_slots(self.h, slots.h, vp(d_cols) if d_cols else None, stride, C.byref(out), C.byref(n),
                                           C.byref(st), log, 80 if want_log else 0, C.byref(ln)))
        stats = {f: getattr(st, f) for f, _ in KernelStats._fields_}
        recs = [(log[i].cls, log[i].perms, log[i].start_us, log[i].end_us) for i in range(ln.value)] if want_log else None
        return BorrowedProof(out, n.value), stats, recs

    def prove_slots_repeat(self, slots, reps, d_cols=None, stride=0):
        """zigzh_prove_trace_slots_repeat: `reps` proofs back to back inside the library (a lane of a service) -> (BorrowedProof
        of the last one, field-wise sum of the proofs' kernel statistics, sum of their phase timings)."""
        from ._ffi import KernelStats
        out, n = u8p(), C.c_size_t()
        st = KernelStats()
        tm = (C.c_double * 10)()
        _check(lib.zigzh_prove_trace_slots_repeat(self.h, slots.h, vp(d_cols) if d_cols else None, stride, reps, C.byref(out),
                                                  C.byref(n), C.byref(st), tm))
        return BorrowedProof(out, n.value), {f: getattr(st, f) for f, _ in KernelStats._fields_}, dict(zip(TIMING_NAMES, list(tm)))

    def prove_sharded(self, ctx, d_cols, stride, dist, allgather=None):
        """ONE proof over dist.get_world_size() GPUs, sharded by column (zigzh_prove_trace_sharded): every rank holds
        the same trace and a resident copy of the 43 columns, commits / opens its own block, and returns the complete
        proof (BorrowedProof; identical on every rank and to the unsharded proof)."""
        from .shard import ShmComm, RcclComm
        out, n = u8p(), C.c_size_t()
        if isinstance(allgather, (ShmComm, RcclComm)):  # a built-in transport (shared memory / native RCCL): no torch involved
            _check(lib.zigzh_prove_trace_sharded(self.h, ctx.h, vp(d_cols), stride, allgather.rank, allgather.world,
                                                 C.cast(allgather.hook, vp), allgather.user, C.byref(out), C.byref(n)))
            return BorrowedProof(out, n.value)
        cb = allgather or make_allgather(dist)
        _check(lib.zigzh_prove_trace_sharded(self.h, ctx.h, vp(d_cols), stride, dist.get_rank(), dist.get_world_size(),
                                             C.cast(cb, vp), None, C.byref(out), C.byref(n)))
        return BorrowedProof(out, n.value)


from ._ffi import ALLGATHER_FN  # noqa: E402  zigzh_allgather_fn == zigz_allgather_fn
from .shard import make_allgather  # noqa: E402,F401  (the hook on torch.distributed)


class Slots:
    """GpuSlots (csrc/host/zigz_host.hpp): k contexts shared by the proving threads of a service; a proof holds one only for
    its GPU phases (Trace.prove_slots).  ctx(i) / borrow(): the contexts themselves, for set-up and for work of the caller's
    own (allocations, witness uploads)."""

    def __init__(self, device, k):
        h = vp()
        _check(lib.zigzh_slots_create(device, k, C.byref(h)))
        self.h, self.k, self.device = h, k, device

    def set_batching(self, max_batch, linger_us=150.0, max_nv=17):
        """small traces share commit jobs (GpuBatcher): up to max_batch proofs that arrive within linger_us of each other"""
        _check(lib.zigzh_slots_set_batching(self.h, max_batch, linger_us, max_nv))

    def ctx(self, i):
        from .hip import Context
        return Context(_borrowed=vp(lib.zigzh_slots_ctx(self.h, i)))

    def contexts(self):
        return [self.ctx(i) for i in range(self.k)]

    def borrow(self):
        slots = self

        class _Lease:
            def __enter__(self_l):
                from .hip import Context
                self_l.raw = vp(lib.zigzh_slots_acquire(slots.h))
                return Context(_borrowed=self_l.raw)

            def __exit__(self_l, *a):
                lib.zigzh_slots_release(slots.h, self_l.raw)
        return _Lease()

    def commit_path_repeat(self, d_cols, stride, nv, points, masks, reps):
        """zigzh_commit_path_repeat: the commit path alone (slot, begin on the 43 resident columns, roots, open_all, end), `reps`
        times back to back inside the library -- measurement of the GPU side without the interpreter in the loop."""
        pts = np.ascontiguousarray(points, dtype=np.uint64)
        assert pts.shape == (43, nv)
        m = (C.c_int64 * 3)(*[int(x) for x in masks])
        _check(lib.zigzh_commit_path_repeat(self.h, vp(d_cols), stride, nv, pts.ctypes.data_as(u64p), m, reps))

    def close(self):
        if getattr(self, "h", None):
            lib.zigzh_slots_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


class BorrowedProof:
    """Proof bytes living in a library-owned buffer (valid until the next prove on this thread)."""

    def __init__(self, ptr, n):
        self.ptr, self.n = ptr, n

    def __len__(self):
        return self.n

    def tobytes(self):
        return C.string_at(self.ptr, self.n)


def vm_run(program, entry_pc, max_steps):
    """VMState.init + run(max_steps) -> (status, regs[32], pc, steps) (host only)."""
    regs = np.zeros(32, dtype=np.uint64)
    pc, steps = C.c_uint64(), C.c_size_t()
    rc = lib.zigzh_vm_run(bytes(program), len(program), entry_pc, max_steps, regs.ctypes.data_as(u64p), C.byref(pc),
                          C.byref(steps))
    return rc, [int(x) for x in regs], pc.value, steps.value


def sumcheck_prove_bytes(ctx, evals):
    a, ap = _u64(evals)
    n = len(evals)
    nv = max(n.bit_length() - 1, 0)
    out = np.zeros((3 * nv + 2) * 8, dtype=np.uint8)
    ln = C.c_size_t()
    _check(lib.zigzh_sumcheck_prove_bytes(ctx.h, ap, n, out.ctypes.data_as(u8p), C.byref(ln)))
    return out[: ln.value].tobytes()


def lasso_prove_table(ctx, kind, bits, queries, mapping=None):
    q, qp = _u64(np.asarray(queries, dtype=np.uint64).reshape(-1))
    nq = len(queries)
    m, mp = _u64(mapping)
    out = np.zeros(4096, dtype=np.uint8)
    ln, nl = C.c_size_t(), C.c_size_t()
    qc = np.zeros(32, dtype=np.uint8)
    tc = np.zeros(32, dtype=np.uint8)
    _check(lib.zigzh_lasso_prove_table(ctx.h, kind, bits, qp, nq, mp if mapping is not None else None,
                                       0 if mapping is None else len(mapping), out.ctypes.data_as(u8p), C.byref(ln),
                                       qc.ctypes.data_as(u8p), tc.ctypes.data_as(u8p), C.byref(nl)))
    return out[: ln.value].tobytes(), qc.tobytes(), tc.tobytes(), nl.value


def commit_open_verify(ctx, evals, point):
    a, ap = _u64(evals)
    q, qp = _u64(point)
    root = np.zeros(32, dtype=np.uint8)
    val, idx, ok = C.c_uint64(), C.c_uint64(), C.c_int()
    _check(lib.zigzh_commit_open_verify(ctx.h, ap, len(evals), qp, len(point), root.ctypes.data_as(u8p), C.byref(val),
                                        C.byref(idx), C.byref(ok)))
    return root.tobytes(), val.value, idx.value, bool(ok.value)
