"""ctypes binding of the CPU parity oracle (oracle/libzigz_oracle.so).  Test infrastructure only:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by zigz_amd."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
P_BB = 2013265921
NCOL = 43

u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)
szp = C.POINTER(C.c_size_t)


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def _load():
    so = os.path.join(ORACLE_DIR, "libzigz_oracle.so")
    src = os.path.join(ORACLE_DIR, "zigz_oracle.c")
    if not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        build_oracle()
    return C.CDLL(so)


lib = _load()


def _sig(name, res, *args):
    f = getattr(lib, name)
    f.restype = res
    f.argtypes = list(args)
    return f


_sig("orc_f_init", C.c_uint64, C.c_uint64, C.c_uint64)
for _n in ("orc_f_add", "orc_f_sub", "orc_f_mul", "orc_f_pow"):
    _sig(_n, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64)
_sig("orc_f_neg", C.c_uint64, C.c_uint64, C.c_uint64)
_sig("orc_f_inv", C.c_int, C.c_uint64, C.c_uint64, u64p)
_sig("orc_sha3_256", None, C.c_char_p, C.c_size_t, u8p)
_sig("orc_sha256", None, C.c_char_p, C.c_size_t, u8p)
_sig("orc_xxh3_64", C.c_uint64, C.c_uint64, C.c_char_p, C.c_size_t)
_sig("orc_hash_leaf", None, C.c_uint64, u8p)
_sig("orc_hash_internal", None, C.c_char_p, C.c_char_p, u8p)
_sig("orc_tr_new", C.c_void_p)
_sig("orc_tr_free", None, C.c_void_p)
_sig("orc_tr_append_bytes", None, C.c_void_p, C.c_char_p, C.c_size_t)
_sig("orc_tr_append_field", None, C.c_void_p, C.c_uint64)
_sig("orc_tr_challenge", C.c_uint64, C.c_void_p, C.c_uint64)
_sig("orc_mle_eval", C.c_int, C.c_uint64, u64p, C.c_size_t, u64p, C.c_size_t, u64p)
_sig("orc_mle_partial_eval", C.c_int, C.c_uint64, u64p, C.c_size_t, C.c_uint64, u64p)
_sig("orc_mle_round_poly", C.c_int, C.c_uint64, u64p, C.c_size_t, u64p)
_sig("orc_mle_sum", C.c_uint64, C.c_uint64, u64p, C.c_size_t)
_sig("orc_eval_univariate", C.c_uint64, C.c_uint64, u64p, C.c_size_t, C.c_uint64)
_sig("orc_sumcheck_prove", C.c_int, C.c_uint64, u64p, C.c_size_t, u64p, u64p, u64p)
_sig("orc_sumcheck_prove_interactive", C.c_int, C.c_uint64, u64p, C.c_size_t, u64p, C.c_size_t, u64p, u64p, u64p)
_sig("orc_sumcheck_to_bytes", C.c_size_t, C.c_size_t, u64p, u64p, C.c_uint64, u8p)
_sig("orc_sumcheck_verify", C.c_int, C.c_uint64, u64p, C.c_size_t, C.c_uint64, u64p, u64p, C.c_uint64)
_sig("orc_merkle_build", C.c_int, u64p, C.c_size_t, u8p, szp)
_sig("orc_merkle_open", C.c_int, u64p, C.c_size_t, C.c_size_t, u8p, u8p, u64p)
_sig("orc_merkle_verify", C.c_int, C.c_char_p, C.c_uint64, C.c_char_p, C.c_char_p, C.c_size_t)
_sig("orc_merkle_levels", C.c_int, u64p, C.c_size_t, u8p, szp)
_sig("orc_point_to_index", C.c_size_t, u64p, C.c_size_t)
_sig("orc_commit_open", C.c_int, C.c_uint64, u64p, C.c_size_t, u64p, C.c_size_t, u64p, u64p, u8p, u8p, u64p)
_sig("orc_lasso_hash_row", C.c_uint64, C.c_uint64, u64p, C.c_size_t)
_sig("orc_lasso_commit", None, u64p, C.c_size_t, u8p)
_sig("orc_lasso_prove", C.c_int, C.c_uint64, u64p, C.c_size_t, u64p, C.c_size_t, C.c_size_t, C.c_size_t,
     szp, u64p, u64p, u64p, u8p, u8p)
_sig("orc_lasso_prove_with_mapping", C.c_int, C.c_uint64, u64p, C.c_size_t, u64p, C.c_size_t, C.c_size_t,
     C.c_size_t, u64p, C.c_size_t, szp, u64p, u64p, u64p, u8p, u8p)
_sig("orc_build_table", None, C.c_uint64, C.c_int, C.c_size_t, u64p)
_sig("orc_vm_run_kat", C.c_int, C.c_char_p, C.c_size_t, C.c_uint64, C.c_size_t, u64p, u64p, szp)
_sig("orc_trace_new", C.c_void_p)
_sig("orc_trace_free", None, C.c_void_p)
_sig("orc_vm_run", C.c_int, C.c_char_p, C.c_size_t, C.c_uint64, u64p, C.c_size_t, C.c_size_t, u64p, C.c_size_t, C.c_void_p)
_sig("orc_witness", C.c_int, C.c_uint64, C.c_void_p, u64p, szp)
_sig("orc_log2_ceil", C.c_size_t, C.c_size_t)
_sig("orc_prove", C.c_int, C.c_uint64, C.c_char_p, C.c_size_t, C.c_uint64, u64p, C.c_size_t, C.c_int, C.c_size_t,
     u64p, C.c_size_t, C.POINTER(u8p), szp, szp)
_sig("orc_generate_commitments", C.c_int, C.c_uint64, C.c_void_p, u64p, C.c_size_t, u8p, u64p, u64p, u64p, u64p, u8p, u8p)
_sig("orc_generate_commitments_fast", C.c_int, C.c_uint64, C.c_void_p, u64p, C.c_size_t, u8p, u64p, u64p, u64p, u64p, u8p, u8p)
_sig("orc_commit_column_literal", C.c_int, C.c_uint64, u64p, C.c_size_t, u64p, u8p, u64p, u64p, u8p, u8p, u64p)
_sig("orc_verify", C.c_int, C.c_uint64, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_int))
_sig("orc_proof_size", C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t)
_sig("orc_free", None, C.c_void_p)


class OracleError(Exception):
    def __init__(self, code):
        super().__init__(f"oracle status {code}")
        self.code = code


def _chk(rc):
    if rc != 0:
        raise OracleError(rc)


def _u64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a, a.ctypes.data_as(u64p)


def _out_u64(n):
    a = np.zeros(max(n, 1), dtype=np.uint64)
    return a, a.ctypes.data_as(u64p)


def _out_u8(n):
    a = np.zeros(max(n, 1), dtype=np.uint8)
    return a, a.ctypes.data_as(u8p)


def sha3_256(b):
    o, op = _out_u8(32)
    lib.orc_sha3_256(bytes(b), len(b), op)
    return o.tobytes()


def sha256(b):
    o, op = _out_u8(32)
    lib.orc_sha256(bytes(b), len(b), op)
    return o.tobytes()


def xxh3_64(b, seed=0):
    return lib.orc_xxh3_64(seed, bytes(b), len(b))


def hash_leaf(v):
    o, op = _out_u8(32)
    lib.orc_hash_leaf(int(v), op)
    return o.tobytes()


def hash_internal(l, r):
    o, op = _out_u8(32)
    lib.orc_hash_internal(bytes(l), bytes(r), op)
    return o.tobytes()


class Transcript:
    def __init__(self):
        self.h = lib.orc_tr_new()

    def __del__(self):
        if getattr(self, "h", None):
            lib.orc_tr_free(self.h)
            self.h = None

    def append_bytes(self, b):
        lib.orc_tr_append_bytes(self.h, bytes(b), len(b))

    def append_field(self, v):
        lib.orc_tr_append_field(self.h, int(v))

    def challenge(self, p=P_BB):
        return lib.orc_tr_challenge(self.h, p)


def mle_eval(p, ev, pt):
    a, ap = _u64(ev)
    q, qp = _u64(pt) if len(pt) else _out_u64(1)
    out = C.c_uint64()
    _chk(lib.orc_mle_eval(p, ap, len(a), qp, len(pt), C.byref(out)))
    return out.value


def mle_partial_eval(p, ev, r):
    a, ap = _u64(ev)
    o, op = _out_u64(len(a) // 2)
    _chk(lib.orc_mle_partial_eval(p, ap, len(a), int(r), op))
    return o[: len(a) // 2]


def mle_round_poly(p, ev):
    a, ap = _u64(ev)
    o, op = _out_u64(2)
    _chk(lib.orc_mle_round_poly(p, ap, len(a), op))
    return [int(o[0]), int(o[1])]


def mle_sum(p, ev):
    a, ap = _u64(ev)
    return lib.orc_mle_sum(p, ap, len(a))


def sumcheck_prove(p, ev, challenges=None):
    a, ap = _u64(ev)
    n = len(a)
    nv = max(n.bit_length() - 1, 0)
    r, rp = _out_u64(2 * nv)
    pt, ptp = _out_u64(nv)
    fe = C.c_uint64()
    if challenges is None:
        _chk(lib.orc_sumcheck_prove(p, ap, n, rp, ptp, C.byref(fe)))
    else:
        c, cp = _u64(challenges)
        _chk(lib.orc_sumcheck_prove_interactive(p, ap, n, cp, len(c), rp, ptp, C.byref(fe)))
    return r[: 2 * nv].copy(), pt[:nv].copy(), fe.value


def sumcheck_to_bytes(rounds, point, fe):
    nv = len(point)
    r, rp = _u64(rounds) if nv else _out_u64(1)
    q, qp = _u64(point) if nv else _out_u64(1)
    o, op = _out_u8((3 * nv + 2) * 8)
    n = lib.orc_sumcheck_to_bytes(nv, rp, qp, int(fe), op)
    return o[:n].tobytes()


def sumcheck_verify(p, ev, claimed, rounds, point, fe):
    a, ap = _u64(ev)
    r, rp = _u64(rounds)
    q, qp = _u64(point)
    return bool(lib.orc_sumcheck_verify(p, ap, len(a), int(claimed), rp, qp, int(fe)))


def merkle_build(values):
    a, ap = _u64(values)
    root, rp = _out_u8(32)
    h = C.c_size_t()
    _chk(lib.orc_merkle_build(ap, len(a), rp, C.byref(h)))
    return root.tobytes(), h.value


def merkle_open(values, index):
    a, ap = _u64(values)
    npad = 1
    while npad < len(a):
        npad <<= 1
    h = npad.bit_length() - 1
    sib, sp = _out_u8(32 * h)
    dirs, dp = _out_u8(h)
    leaf = C.c_uint64()
    _chk(lib.orc_merkle_open(ap, len(a), index, sp, dp, C.byref(leaf)))
    return sib[: 32 * h].tobytes(), dirs[:h].tobytes(), leaf.value


def merkle_verify(root, value, siblings, dirs):
    return bool(lib.orc_merkle_verify(bytes(root), int(value), bytes(siblings) or b"\0", bytes(dirs) or b"\0", len(dirs)))


def merkle_levels(values):
    a, ap = _u64(values)
    npad = 1
    while npad < len(a):
        npad <<= 1
    lv, lp = _out_u8(2 * npad * 32)
    h = C.c_size_t()
    _chk(lib.orc_merkle_levels(ap, len(a), lp, C.byref(h)))
    return lv, h.value


def commit_open(p, ev, pt):
    a, ap = _u64(ev)
    nv = len(pt)
    q, qp = _u64(pt) if nv else _out_u64(1)
    sib, sp = _out_u8(32 * nv)
    dirs, dp = _out_u8(nv)
    val, idx, leaf = C.c_uint64(), C.c_uint64(), C.c_uint64()
    _chk(lib.orc_commit_open(p, ap, len(a), qp, nv, C.byref(val), C.byref(idx), sp, dp, C.byref(leaf)))
    return val.value, idx.value, sib[: 32 * nv].tobytes(), dirs[:nv].tobytes(), leaf.value


def lasso_hash_row(p, fields):
    a, ap = _u64(fields)
    return lib.orc_lasso_hash_row(p, ap, len(a))


def build_table(p, kind, bits):
    o, op = _out_u64((1 << (2 * bits)) * 3)
    lib.orc_build_table(p, kind, bits, op)
    return o.reshape(-1, 3)


def lasso_prove(p, table, queries, n_in=2, n_out=1, mapping=None):
    t, tp = _u64(np.asarray(table, dtype=np.uint64).reshape(-1))
    q, qp = _u64(np.asarray(queries, dtype=np.uint64).reshape(-1)) if len(queries) else _out_u64(1)
    w = n_in + n_out
    nq = len(queries)
    rows = len(t) // w
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
        _chk(lib.orc_lasso_prove(p, tp, rows, qp, nq, n_in, n_out, C.byref(nv), rp, ptp, C.byref(fe), qcp, tcp))
    else:
        m, mp = _u64(mapping) if len(mapping) else _out_u64(1)
        _chk(lib.orc_lasso_prove_with_mapping(p, tp, rows, qp, nq, n_in, n_out, mp, len(mapping), C.byref(nv),
                                              rp, ptp, C.byref(fe), qcp, tcp))
    v = nv.value
    return dict(nv=v, rounds=r[: 2 * v].copy(), point=pt[:v].copy(), final_eval=fe.value,
                query_commit=qc.tobytes(), table_commit=tc.tobytes())


def vm_run_kat(program, entry_pc, max_steps):
    regs, rp = _out_u64(32)
    pc = C.c_uint64()
    steps = C.c_size_t()
    rc = lib.orc_vm_run_kat(bytes(program), len(program), entry_pc, max_steps, rp, C.byref(pc), C.byref(steps))
    return rc, [int(x) for x in regs], pc.value, steps.value


def witness_from_program(p, program, entry_pc, initial_regs=None, max_steps=1 << 20, inputs=None):
    """Runs the VM then builds the 43 columns. Returns (cols[43,N] uint64, nv, num_steps, n_lookups)."""
    tr = lib.orc_trace_new()
    try:
        ir, irp = (_u64(initial_regs) if initial_regs is not None and len(initial_regs) else _out_u64(1))
        inp, inpp = (_u64(inputs) if inputs is not None and len(inputs) else _out_u64(1))
        _chk(lib.orc_vm_run(bytes(program), len(program), entry_pc, irp, 0 if initial_regs is None else len(initial_regs),
                            max_steps, inpp, 0 if inputs is None else len(inputs), tr))
        ns = C.cast(tr, szp)[0]
        nv = lib.orc_log2_ceil(ns) if ns else 0
        cols, cp = _out_u64(NCOL << nv)
        nvo = C.c_size_t()
        _chk(lib.orc_witness(p, tr, cp, C.byref(nvo)))
        return cols.reshape(NCOL, 1 << nv), nv, ns
    finally:
        lib.orc_trace_free(tr)


class OrcTrace(C.Structure):  # struct orc_trace of oracle/zigz_oracle.h
    _fields_ = [("num_steps", C.c_size_t), ("capacity", C.c_size_t), ("pc", u64p), ("regs_after", u64p),
                ("opcode", u8p), ("rd", u8p), ("rs1", u8p), ("rs2", u8p), ("funct3", u8p), ("funct7", u8p),
                ("imm", C.POINTER(C.c_int64)), ("mem_kind", u8p), ("mem_addr", u64p), ("mem_value", u64p),
                ("is_lookup", u8p), ("final_pc", C.c_uint64), ("final_regs", C.c_uint64 * 32), ("outputs", u64p),
                ("n_outputs", C.c_size_t), ("cap_outputs", C.c_size_t), ("halted", C.c_int)]


def vm_trace(program, entry_pc, initial_regs=None, max_steps=1 << 20, inputs=None):
    """The oracle VM's execution trace (the loop of prover.zig:117-142) as numpy arrays, one entry per recorded step:
    pc, regs_after[ns,32], opcode, rd, rs1, rs2, funct3, funct7, imm (int64), mem_kind (0 none / 1 load / 2 store),
    mem_addr, mem_value, is_lookup; plus num_steps and num_lookups."""
    tr = lib.orc_trace_new()
    try:
        ir, irp = (_u64(initial_regs) if initial_regs is not None and len(initial_regs) else _out_u64(1))
        inp, inpp = (_u64(inputs) if inputs is not None and len(inputs) else _out_u64(1))
        _chk(lib.orc_vm_run(bytes(program), len(program), entry_pc, irp, 0 if initial_regs is None else len(initial_regs),
                            max_steps, inpp, 0 if inputs is None else len(inputs), tr))
        t = C.cast(tr, C.POINTER(OrcTrace)).contents
        ns = t.num_steps

        def arr(ptr, n, dt):
            return np.ctypeslib.as_array(ptr, shape=(max(n, 1),))[:n].astype(dt, copy=True) if n else np.zeros(0, dtype=dt)
        out = {"num_steps": ns}
        for k in ("pc", "mem_addr", "mem_value"):
            out[k] = arr(getattr(t, k), ns, np.uint64)
        out["regs_after"] = arr(t.regs_after, ns * 32, np.uint64).reshape(ns, 32)
        for k in ("opcode", "rd", "rs1", "rs2", "funct3", "funct7", "mem_kind", "is_lookup"):
            out[k] = arr(getattr(t, k), ns, np.uint8)
        out["imm"] = arr(t.imm, ns, np.int64)
        out["num_lookups"] = int(out["is_lookup"].sum())
        return out
    finally:
        lib.orc_trace_free(tr)


def prove(p, program, entry_pc=0x1000, initial_regs=None, max_steps=1 << 20, inputs=None):
    ir, irp = (_u64(initial_regs) if initial_regs is not None and len(initial_regs) else _out_u64(1))
    inp, inpp = (_u64(inputs) if inputs is not None and len(inputs) else _out_u64(1))
    out = u8p()
    n = C.c_size_t()
    ns = C.c_size_t()
    _chk(lib.orc_prove(p, bytes(program), len(program), entry_pc, irp,
                       0 if initial_regs is None else len(initial_regs), 0 if initial_regs is None else 1,
                       max_steps, inpp, 0 if inputs is None else len(inputs), C.byref(out), C.byref(n), C.byref(ns)))
    try:
        return C.string_at(out, n.value), ns.value
    finally:
        lib.orc_free(out)


def generate_commitments(p, tr, cols, fast=False):
    """cols: [43, N] canonical uint64.  Continues transcript `tr`.  Returns dict of outputs."""
    cols = np.ascontiguousarray(cols, dtype=np.uint64)
    N = cols.shape[1]
    nv = N.bit_length() - 1
    roots, rp = _out_u8(NCOL * 32)
    points, pp = _out_u64(NCOL * nv)
    values, vp = _out_u64(NCOL)
    indices, ip = _out_u64(NCOL)
    leaves, lp = _out_u64(NCOL)
    sib, sp = _out_u8(NCOL * nv * 32)
    dirs, dp = _out_u8(NCOL * nv)
    fn = lib.orc_generate_commitments_fast if fast else lib.orc_generate_commitments
    _chk(fn(p, tr.h, cols.ctypes.data_as(u64p), nv, rp, pp, vp, ip, lp, sp, dp))
    return dict(roots=roots[: NCOL * 32].reshape(NCOL, 32), points=points[: NCOL * nv].reshape(NCOL, nv),
                values=values[:NCOL], indices=indices[:NCOL], leaves=leaves[:NCOL],
                siblings=sib[: NCOL * nv * 32].reshape(NCOL, nv, 32), dirs=dirs[: NCOL * nv].reshape(NCOL, nv))


def commit_column_literal(p, col, point):
    """Literal per-column reference work (commit + eval + open with its second eval and level rebuild)."""
    a, ap = _u64(col)
    nv = len(point)
    q, qp = _u64(point) if nv else _out_u64(1)
    root, rp = _out_u8(32)
    sib, sp = _out_u8(32 * nv)
    dirs, dp = _out_u8(nv)
    val, idx, leaf = C.c_uint64(), C.c_uint64(), C.c_uint64()
    _chk(lib.orc_commit_column_literal(p, ap, nv, qp, rp, C.byref(val), C.byref(idx), sp, dp, C.byref(leaf)))
    return root.tobytes(), val.value, idx.value, sib[: 32 * nv].tobytes(), dirs[:nv].tobytes(), leaf.value


def verify(p, proof, program):
    res = C.c_int(-1)
    rc = lib.orc_verify(p, bytes(proof), len(proof), bytes(program), len(program), C.byref(res))
    return rc, res.value


def proof_size(nv, n_initial_regs, n_outputs, n_lookups):
    return lib.orc_proof_size(nv, n_initial_regs, n_outputs, n_lookups)


def splitmix64_field(seed, n, p=P_BB):
    """splitmix64(seed) stream reduced mod p (the synthetic-input generator named in SURVEY s8d)."""
    x = np.uint64(seed)
    idx = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z % np.uint64(p)
