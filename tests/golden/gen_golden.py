#!/usr/bin/env python3
"""Independent Python restatement of the zigz reference hot path -> tests/golden/golden.json.

Purpose: the Zig reference cannot be built or run here (no zig toolchain, remote hash-zig
dependency), and its tests pin no digest / challenge / root / proof byte.  The strongest pin
available offline is two independent readings of the same source: this file (pure Python on top of
hashlib.sha3_256 / hashlib.sha256 / xxhash.xxh3_64, written from the reference's .zig text) and the
C oracle (oracle/zigz_oracle.c, own Keccak/SHA-256/XXH3).  tests/test_oracle_golden.py checks the C
oracle against the vectors emitted here; the GPU path is then checked against the C oracle and
against these vectors directly.

Run in the build container only:  python tests/golden/gen_golden.py
Everything below cites reference file:line (relative to the reference repo root).
"""
import hashlib, json, os, struct
import xxhash

P_BB = 2013265921  # src/core/field_presets.zig:19
P_17 = 17
M64 = (1 << 64) - 1


# ---------------------------------------------------------------- field (src/core/field.zig:36-147)
def fadd(p, a, b): return (a + b) % p
def fsub(p, a, b): return (a - b) % p
def fmul(p, a, b): return (a * b) % p


def le64(v): return struct.pack("<Q", v)


# ---------------------------------------------------------------- hashes
def sha3(b): return hashlib.sha3_256(b).digest()
def hash_leaf(v): return sha3(le64(v))          # src/core/hash.zig:135-147
def hash_internal(l, r): return sha3(l + r)     # src/core/hash.zig:187-195


class Transcript:  # src/core/hash.zig:255-324
    def __init__(self): self.h = hashlib.sha3_256()
    def append_bytes(self, b): self.h.update(b)
    def append_field(self, v): self.h.update(le64(v))
    def challenge(self, p):
        d = self.h.copy().digest()                      # clone + final (:305-306)
        r = int.from_bytes(d[:8], "little") % p         # digestToFieldElement (:228-242)
        self.h.update(d)                                # absorb the digest (:313)
        return r


# ---------------------------------------------------------------- multilinear (src/poly/multilinear.zig)
def mle_eval(p, ev, pt):  # :110-144, point[0] <-> least significant index bit
    assert len(ev) == 1 << len(pt)
    res = 0
    for idx, e in enumerate(ev):
        term, index = e, idx
        for v in range(len(pt)):
            term = fmul(p, term, pt[v] if (index & 1) else fsub(p, 1, pt[v]))
            index >>= 1
        res = fadd(p, res, term)
    return res


def mle_partial(p, ev, r):  # :154-180, binds the most significant index bit
    h = len(ev) // 2
    return [fadd(p, fmul(p, fsub(p, 1, r), ev[i]), fmul(p, r, ev[i + h])) for i in range(h)]


def mle_round(p, ev):  # :205-232
    h = len(ev) // 2
    s0 = sum(ev[:h]) % p
    s1 = sum(ev[h:]) % p
    return [s0, fsub(p, s1, s0)]


# ---------------------------------------------------------------- sumcheck (src/proofs/sumcheck_prover.zig:26-91)
def sumcheck_prove(p, ev, challenges=None):
    nv = len(ev).bit_length() - 1
    tr = Transcript()  # fresh transcript, sumcheck_protocol.zig:161
    cur, rounds, point = list(ev), [], []
    for rnd in range(nv):
        c = mle_round(p, cur)
        rounds += c
        if challenges is None:
            tr.append_field(c[0]); tr.append_field(c[1])  # protocol:176-184
            ch = tr.challenge(p)
        else:
            ch = challenges[rnd]
        point.append(ch)
        cur = mle_partial(p, cur, ch)
    return rounds, point, cur[0]


def sumcheck_bytes(nv, rounds, point, fe):  # sumcheck_protocol.zig:76-107
    return le64(nv) + b"".join(le64(x) for x in rounds) + b"".join(le64(x) for x in point) + le64(fe)


# ---------------------------------------------------------------- Merkle (src/commitments/merkle_tree.zig:283-400)
def merkle_levels(values):
    n = len(values)
    npad = 1
    while npad < n: npad <<= 1
    lv = [hash_leaf(v) for v in values] + [hash_leaf(0)] * (npad - n)
    levels = [lv]
    while len(lv) > 1:
        lv = [hash_internal(lv[2 * i], lv[2 * i + 1]) for i in range(len(lv) // 2)]
        levels.append(lv)
    return levels


def merkle_open(values, index):  # :324-360
    levels = merkle_levels(values)
    sib, dirs, ci = [], [], index
    for lv in levels[:-1]:
        sib.append(lv[ci ^ 1]); dirs.append(ci & 1); ci >>= 1
    return levels[-1][0], sib, dirs


def point_to_index(pt):  # src/commitments/polynomial_commit.zig:178-183
    return 0 if not pt else pt[0] % (1 << len(pt))


# ---------------------------------------------------------------- Lasso (src/lookups/lasso_prover.zig:103-252)
def lasso_hash_row(p, fields):  # :208-239
    h = 0
    for f in fields:
        h ^= f
        h = xxhash.xxh3_64_intdigest(le64(h), seed=0)
    return h % p


def lasso_commit(ev):  # :242-252
    return sha3(b"".join(le64(e) for e in ev))


def build_table(kind, bits):  # src/lookups/table_builder.zig:126-213
    m = 1 << bits
    rows = []
    for a in range(m):
        for b in range(m):
            rows.append([a, b, (a + b) % m if kind == 0 else (a ^ b) if kind == 1 else (a & b)])
    return rows


def lasso_prove(p, table_rows, query_rows):
    tev = [lasso_hash_row(p, r) for r in table_rows]
    npad = 1
    while npad < len(query_rows): npad <<= 1
    qev = [lasso_hash_row(p, r) for r in query_rows] + [0] * (npad - len(query_rows))
    rounds, point, fe = sumcheck_prove(p, qev)
    return dict(nv=npad.bit_length() - 1, rounds=rounds, point=point, final_eval=fe,
                query_commit=lasso_commit(qev).hex(), table_commit=lasso_commit(tev).hex())


# ---------------------------------------------------------------- VM (src/vm/state.zig, src/isa/rv64i.zig)
def sx(v, bits):
    v &= (1 << bits) - 1
    return v - (1 << bits) if v >> (bits - 1) else v


def s64(v): return sx(v, 64)
def s32(v): return sx(v, 32)


I_OPS = {0x13, 0x1b, 0x67, 0x03, 0x07, 0x0f, 0x73}


def decode(w):  # rv64i.zig:124-233
    op = w & 0x7f
    if op == 0:
        return None
    d = dict(op=op, rd=(w >> 7) & 31, f3=(w >> 12) & 7, rs1=(w >> 15) & 31, rs2=(w >> 20) & 31, f7=(w >> 25) & 0x7f)
    if op in I_OPS:
        imm = sx(w >> 20, 12)
    elif op in (0x23, 0x27):
        imm = sx(((w >> 25) << 5) | ((w >> 7) & 31), 12)
    elif op == 0x63:
        imm = sx((((w >> 31) & 1) << 12) | (((w >> 7) & 1) << 11) | (((w >> 25) & 0x3f) << 5) | (((w >> 8) & 0xf) << 1), 13)
    elif op in (0x37, 0x17):
        imm = s32(w & 0xfffff000)
    elif op == 0x6f:
        imm = sx((((w >> 31) & 1) << 20) | (((w >> 12) & 0xff) << 12) | (((w >> 20) & 1) << 11) | (((w >> 21) & 0x3ff) << 1), 21)
    else:
        imm = 0
    d["imm"] = imm
    return d


class VMError(Exception):
    pass


def tdiv(a, b):  # truncating signed division
    q = abs(a) // abs(b)
    return q if (a < 0) == (b < 0) else -q


def run_vm(program, entry_pc, initial_regs=None, max_steps=1 << 20, inputs=None):
    """Loop of Prover.prove (src/prover/prover.zig:117-142). Returns trace dict."""
    mem = {}
    for i, b in enumerate(program):
        mem[(entry_pc + i) & M64] = b
    regs = [0] * 32
    if initial_regs:
        for i, v in enumerate(initial_regs[:32]):
            if i: regs[i] = v & M64
    inputs = list(inputs or [])
    ipos, outputs, steps = 0, [], []
    pc, halted, count = entry_pc, False, 0

    def lb(a): return mem.get(a & M64, 0)
    def load(a, n): return sum(lb(a + k) << (8 * k) for k in range(n))
    def store(a, v, n):
        for k in range(n): mem[(a + k) & M64] = (v >> (8 * k)) & 0xff

    while not halted and count < max_steps:
        d = decode(load(pc, 4))
        if d is None:
            halted = True
            break  # InvalidInstruction: no step recorded (state.zig:136-140)
        op, f3, f7, imm = d["op"], d["f3"], d["f7"], d["imm"]
        a, b = regs[d["rs1"]], regs[d["rs2"]]
        immu = imm & M64
        mem_acc, nxt, res = None, (pc + 4) & M64, None
        if op == 0x33:
            if f7 == 1:
                sa, sb = s64(a), s64(b)
                res = [lambda: a * b, lambda: (sa * sb) >> 64, lambda: (sa * b) >> 64, lambda: (a * b) >> 64,
                       lambda: -1 if sb == 0 else (a if (sa == -(1 << 63) and sb == -1) else tdiv(sa, sb)),
                       lambda: M64 if b == 0 else a // b,
                       lambda: a if sb == 0 else (0 if (sa == -(1 << 63) and sb == -1) else sa - sb * tdiv(sa, sb)),
                       lambda: a if b == 0 else a % b][f3]() & M64
            else:
                sh = b & 63
                res = [lambda: (a - b) if f7 == 0x20 else (a + b), lambda: a << sh,
                       lambda: int(s64(a) < s64(b)), lambda: int(a < b), lambda: a ^ b,
                       lambda: (s64(a) >> sh) if f7 == 0x20 else (a >> sh), lambda: a | b, lambda: a & b][f3]() & M64
        elif op == 0x3b:
            x, y = a & 0xffffffff, b & 0xffffffff
            if f7 == 1:
                sx_, sy = s32(x), s32(y)
                if f3 == 0: r32 = x * y
                elif f3 == 4: r32 = -1 if sy == 0 else (x if (sx_ == -(1 << 31) and sy == -1) else tdiv(sx_, sy))
                elif f3 == 5: r32 = 0xffffffff if y == 0 else x // y
                elif f3 == 6: r32 = x if sy == 0 else (0 if (sx_ == -(1 << 31) and sy == -1) else sx_ - sy * tdiv(sx_, sy))
                elif f3 == 7: r32 = x if y == 0 else x % y
                else: raise VMError("InvalidOP32M")
            else:
                sh = y & 31
                if f3 == 0: r32 = (x - y) if f7 == 0x20 else (x + y)
                elif f3 == 1: r32 = x << sh
                elif f3 == 5: r32 = (s32(x) >> sh) if f7 == 0x20 else (x >> sh)
                else: raise VMError("InvalidOP32")
            res = s32(r32) & M64
        elif op == 0x13:
            sh = immu & 63
            res = [lambda: a + immu, lambda: a << sh, lambda: int(s64(a) < imm), lambda: int(a < immu),
                   lambda: a ^ immu, lambda: (s64(a) >> sh) if f7 == 0x20 else (a >> sh),
                   lambda: a | immu, lambda: a & immu][f3]() & M64
        elif op == 0x1b:
            x, sh = a & 0xffffffff, immu & 31
            if f3 == 0: r32 = x + (immu & 0xffffffff)
            elif f3 == 1: r32 = x << sh
            elif f3 == 5: r32 = (s32(x) >> sh) if f7 == 0x20 else (x >> sh)
            else: raise VMError("InvalidOPIMM32")
            res = s32(r32) & M64
        elif op == 0x03:
            addr = (a + immu) & M64
            if f3 == 0: res = sx(load(addr, 1), 8) & M64
            elif f3 == 1: res = sx(load(addr, 2), 16) & M64
            elif f3 == 2: res = sx(load(addr, 4), 32) & M64
            elif f3 == 3: res = load(addr, 8)
            elif f3 == 4: res = load(addr, 1)
            elif f3 == 5: res = load(addr, 2)
            elif f3 == 6: res = load(addr, 4)
            else: raise VMError("InvalidLoadFunct3")
            mem_acc = (1, addr, res)
        elif op == 0x23:
            addr = (a + immu) & M64
            if f3 > 3: raise VMError("InvalidStoreFunct3")
            store(addr, b, 1 << f3)
            mem_acc = (2, addr, b)
        elif op == 0x63:
            if f3 in (2, 3): raise VMError("InvalidBranchFunct3")
            taken = {0: a == b, 1: a != b, 4: s64(a) < s64(b), 5: s64(a) >= s64(b), 6: a < b, 7: a >= b}[f3]
            if taken: nxt = (pc + immu) & M64
        elif op == 0x6f:
            res = (pc + 4) & M64; nxt = (pc + immu) & M64
        elif op == 0x67:
            res = (pc + 4) & M64; nxt = (a + immu) & M64 & ~1
        elif op == 0x37:
            res = immu
        elif op == 0x17:
            res = (pc + immu) & M64
        elif op == 0x73:
            if f3 == 0 and imm == 0:
                sc = regs[17]
                if sc == 1: outputs.append(regs[10])
                elif sc == 2:
                    if ipos < len(inputs): regs[10] = inputs[ipos] & M64; ipos += 1
                    else: regs[10] = 0
            elif f3 == 0 and imm == 1:
                halted = True; nxt = pc
            else:
                raise VMError("UnimplementedSYSTEM")
        elif op == 0x0f:
            pass
        else:
            raise VMError("UnimplementedInstruction")
        if res is not None and op not in (0x23, 0x63, 0x73, 0x0f) and d["rd"] != 0:
            regs[d["rd"]] = res & M64
        steps.append(dict(pc=pc, inst=d, regs=list(regs), mem=mem_acc,
                          lookup=op in (0x33, 0x13, 0x03, 0x23, 0x63)))  # instruction_table.zig:243-274
        pc = nxt
        count += 1
    return dict(steps=steps, final_pc=pc, final_regs=list(regs), outputs=outputs)


def witness(p, tr):  # src/constraints/witness.zig:29-270, order src/prover/prover.zig:376-390
    st = tr["steps"]
    ns = len(st)
    nv = 0 if ns <= 1 else (ns - 1).bit_length()
    N = 1 << nv
    cols = [[0] * N for _ in range(43)]
    for i in range(N):
        s = st[min(i, ns - 1)]
        cols[0][i] = s["pc"] % p
        for r in range(32): cols[1 + r][i] = s["regs"][r] % p
        if i < ns:
            d = s["inst"]
            for c, k in ((33, "op"), (34, "rd"), (35, "rs1"), (36, "rs2"), (37, "f3"), (38, "f7")):
                cols[c][i] = d[k] % p
            cols[39][i] = (d["imm"] & M64) % p
            if s["mem"]:
                kind, addr, val = s["mem"]
                cols[40][i], cols[41][i], cols[42][i] = addr % p, val % p, 1 if kind == 1 else 0
    return nv, cols


def prove(p, program, entry_pc, initial_regs, max_steps, inputs=None):
    """Prover.prove + BinarySerializer.serialize with an exact buffer (prover.zig:73-226, serialization.zig)."""
    t = Transcript()
    ph = hashlib.sha256(bytes(program)).digest()
    t.append_bytes(ph)
    t.append_field(entry_pc % p)
    if initial_regs is not None:
        for r in initial_regs: t.append_field(r % p)
    tr = run_vm(program, entry_pc, initial_regs, max_steps, inputs)
    ns = len(tr["steps"])
    assert ns > 0
    nv, cols = witness(p, tr)
    L = sum(1 for s in tr["steps"] if s["lookup"])
    t.append_bytes(b"SUMCHECK_BEGIN"); t.append_field(ns % p); t.append_field(nv % p)
    cpoint = []
    for _ in range(nv):
        for _ in range(4): t.append_field(0)
        cpoint.append(t.challenge(p))
    t.append_bytes(b"LASSO_BEGIN")
    for i in range(L):
        t.append_bytes(b"LASSO_TABLE"); t.append_field(i % p)
    levels = [merkle_levels(c) for c in cols]
    roots = [lv[-1][0] for lv in levels]
    t.append_bytes(b"POLY_COMMITMENTS")
    for r in roots: t.append_bytes(r)
    openings = []
    for c in range(43):
        pt = [t.challenge(p) for _ in range(nv)]
        val = mle_eval(p, cols[c], pt)
        idx = point_to_index(pt)
        _, sib, dirs = merkle_open(cols[c], idx)
        openings.append((pt, val, idx, cols[c][idx], sib, dirs))
    t.append_bytes(b"OPENING_CLAIMS")
    for o in openings: t.append_field(o[1])
    out = b"ZIGZ" + struct.pack("<IQQII", 1, p, ns, nv, 0)
    out += ph + struct.pack("<QQ", entry_pc, tr["final_pc"])
    ir = initial_regs or []
    out += struct.pack("<I", len(ir)) + b"".join(le64(r) for r in ir)
    out += struct.pack("<I", 32) + b"".join(le64(r) for r in tr["final_regs"])
    out += le64(ns) + struct.pack("<I", len(tr["outputs"])) + b"".join(le64(o) for o in tr["outputs"])
    out += le64(0) * (4 * nv) + b"".join(le64(x) for x in cpoint) + le64(0)
    out += struct.pack("<I", L)
    for i in range(L): out += struct.pack("<IQIQ", i, 1, 0, 0)
    for c in range(43):
        pt, val, idx, leaf, sib, dirs = openings[c]
        out += roots[c] + b"".join(le64(x) for x in pt) + le64(val) + le64(val) + le64(idx) + le64(leaf)
        out += struct.pack("<I", nv) + b"".join(sib) + bytes(dirs)
    return out, ns, nv, L, cols


# ---------------------------------------------------------------- deterministic inputs
def splitmix64(seed):
    x = seed & M64
    while True:
        x = (x + 0x9E3779B97F4A7C15) & M64
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
        yield z ^ (z >> 31)


def rand_field(seed, n, p):
    g = splitmix64(seed)
    return [next(g) % p for _ in range(n)]


def enc(x): return [str(v) for v in x]


def asm_fib(n_loops):
    """Hand-assembled RV64IM loop in the spirit of examples/fibonacci_guest/src/main.zig:19-35:
    ECALL read (a7=2) -> n; loop: t=a+b; a=b; b=t; i+=1; BNE; 2x ECALL commit (a7=1); EBREAK."""
    def I(op, rd, f3, rs1, imm): return ((imm & 0xfff) << 20) | (rs1 << 15) | (f3 << 12) | (rd << 7) | op
    def R(op, rd, f3, rs1, rs2, f7): return (f7 << 25) | (rs2 << 20) | (rs1 << 15) | (f3 << 12) | (rd << 7) | op
    def B(f3, rs1, rs2, off):
        o = off & 0x1fff
        return (((o >> 12) & 1) << 31) | (((o >> 5) & 0x3f) << 25) | (rs2 << 20) | (rs1 << 15) | (f3 << 12) | (((o >> 1) & 0xf) << 8) | (((o >> 11) & 1) << 7) | 0x63
    prog = [
        I(0x13, 17, 0, 0, 2),      # ADDI a7, x0, 2
        0x00000073,                # ECALL (read n -> a0)
        I(0x13, 5, 0, 10, 0),      # ADDI t0, a0, 0   (n)
        I(0x13, 6, 0, 0, 0),       # ADDI t1, x0, 0   (a)
        I(0x13, 7, 0, 0, 1),       # ADDI t2, x0, 1   (b)
        I(0x13, 28, 0, 0, 0),      # ADDI t3, x0, 0   (i)
        # loop:
        R(0x33, 29, 0, 6, 7, 0),   # ADD t4, t1, t2
        I(0x13, 6, 0, 7, 0),       # ADDI t1, t2, 0
        I(0x13, 7, 0, 29, 0),      # ADDI t2, t4, 0
        R(0x33, 30, 0, 29, 28, 1), # MUL t5, t4, t3   (RV64M in the mix)
        I(0x13, 28, 0, 28, 1),     # ADDI t3, t3, 1
        B(1, 28, 5, -20),          # BNE t3, t0, loop
        I(0x13, 17, 0, 0, 1),      # ADDI a7, x0, 1
        I(0x13, 10, 0, 6, 0),      # ADDI a0, t1, 0
        0x00000073,                # ECALL commit
        I(0x13, 10, 0, 30, 0),     # ADDI a0, t5, 0
        0x00000073,                # ECALL commit
        0x00100073,                # EBREAK
    ]
    return b"".join(struct.pack("<I", w) for w in prog), [n_loops]


def main():
    G = {}
    # (1) hash known answers (FIPS 202 / FIPS 180-4 / XXH3) -- pins the oracle's own primitives
    G["hash_kats"] = {
        "sha3_256": {m.hex(): sha3(m).hex() for m in [b"", le64(0), le64(1), le64(P_BB - 1), b"Hello, zigz!",
                                                     b"a" * 135, b"a" * 136, b"a" * 137, bytes(range(200))]},
        "sha256": {m.hex(): hashlib.sha256(m).hexdigest() for m in [b"", b"abc", b"a" * 55, b"a" * 56, b"a" * 64,
                                                                   bytes(range(119)), bytes(range(120))]},
        "xxh3_64_seed0": {m.hex(): str(xxhash.xxh3_64_intdigest(m, seed=0)) for m in
                          [le64(0), le64(1), le64(0x0123456789abcdef), le64(M64), b"\x01\x02\x03\x04", b"abcdef"]},
        "domain_strings": {s: sha3(s.encode()).hex() for s in
                           ["SUMCHECK_BEGIN", "LASSO_BEGIN", "LASSO_TABLE", "POLY_COMMITMENTS", "OPENING_CLAIMS"]},
        "merge_leaf1_leaf2": hash_internal(hash_leaf(1), hash_leaf(2)).hex(),
    }
    # (2) transcript KAT: absorb LE64(3), LE64(4) -> challenge twice (values differ: hash.zig:301-316)
    t = Transcript(); t.append_field(3); t.append_field(4)
    G["transcript_kat"] = {"absorb": ["3", "4"], "challenges_babybear": [str(t.challenge(P_BB)), str(t.challenge(P_BB))]}
    t = Transcript(); t.append_bytes(b"zigz"); c17 = [t.challenge(P_17) for _ in range(3)]
    G["transcript_kat"]["bytes_zigz_f17"] = enc(c17)

    # (3) the reference's own MLE known answers (src/poly/multilinear.zig:383-506), F17
    G["ref_mle_kats_f17"] = {
        "evals": enc([1, 2, 3, 4]),
        "eval": [{"point": enc(pt), "value": str(v)} for pt, v in (([0, 0], 1), ([1, 0], 2), ([0, 1], 3), ([1, 1], 4))],
        "eval_01": [{"evals": enc([0, 1]), "point": enc([2]), "value": "2"}, {"evals": enc([0, 1]), "point": enc([5]), "value": "5"}],
        "partial_eval_0": enc([1, 2]), "sum": "10", "round_poly": enc([3, 4]),
        "univariate_3_5x": {"coeffs": enc([3, 5]), "at": enc([0, 1, 2]), "values": enc([3, 8, 13])},  # sumcheck_protocol.zig:219-236
    }
    for e in G["ref_mle_kats_f17"]["eval"]:
        assert mle_eval(P_17, [1, 2, 3, 4], [int(x) for x in e["point"]]) == int(e["value"])
    assert mle_partial(P_17, [1, 2, 3, 4], 0) == [1, 2] and mle_round(P_17, [1, 2, 3, 4]) == [3, 4]

    # (4) sumcheck proofs
    sc = []
    cases = [(P_17, [1, 2, 3, 4]), (P_BB, [1, 2, 3, 4]), (P_BB, [5, 7]),
             (P_BB, [(i + 1) % P_BB for i in range(1 << 12)])]  # config 1: evals[i] = i+1, 2^12
    for nv, seed in ((1, 11), (2, 12), (5, 15), (10, 20)):
        cases.append((P_BB, rand_field(seed, 1 << nv, P_BB)))
    for p, ev in cases:
        rounds, point, fe = sumcheck_prove(p, ev)
        nv = len(point)
        assert mle_eval(p, ev, point[::-1]) == fe  # final_eval == eval(reverse(final_point))
        entry = dict(p=str(p), nv=nv, rounds=enc(rounds), point=enc(point), final_eval=str(fe),
                     eval_at_point=str(mle_eval(p, ev, point)),
                     bytes_sha3=sha3(sumcheck_bytes(nv, rounds, point, fe)).hex())
        if len(ev) <= 32: entry["evals"] = enc(ev)
        elif ev[0] == 1 and ev[1] == 2: entry["evals_gen"] = "iota1"
        else: entry["evals_gen"] = {"splitmix64_seed": {1: 11, 2: 12, 5: 15, 10: 20}[nv]}
        if len(ev) <= 4: entry["bytes"] = sumcheck_bytes(nv, rounds, point, fe).hex()
        sc.append(entry)
    G["sumcheck"] = sc
    # interactive variant
    ev = rand_field(77, 8, P_BB); chs = [5, 1000000007 % P_BB, 123456789]
    rounds, point, fe = sumcheck_prove(P_BB, ev, chs)
    G["sumcheck_interactive"] = dict(evals=enc(ev), challenges=enc(chs), rounds=enc(rounds), final_eval=str(fe))

    # (5) Merkle: [1,2,3,4], 5 -> 8 padded, single leaf
    mk = []
    for vals in ([1, 2, 3, 4], [1, 2, 3, 4, 5], [7], rand_field(5, 16, P_BB)):
        lv = merkle_levels(vals)
        ent = dict(values=enc(vals), root=lv[-1][0].hex(), height=len(lv) - 1, openings=[])
        for idx in range(len(vals)):
            _, sib, dirs = merkle_open(vals, idx)
            ent["openings"].append(dict(index=idx, siblings=[s.hex() for s in sib], dirs=dirs))
        mk.append(ent)
    G["merkle"] = mk

    # (6) commit/open: eval + index + path
    co = []
    for nv, seed in ((2, 31), (4, 32), (6, 33)):
        ev = rand_field(seed, 1 << nv, P_BB); pt = rand_field(seed + 100, nv, P_BB)
        idx = point_to_index(pt); root, sib, dirs = merkle_open(ev, idx)
        co.append(dict(nv=nv, evals_seed=seed, point=enc(pt), value=str(mle_eval(P_BB, ev, pt)), index=idx,
                       leaf=str(ev[idx]), root=root.hex(), siblings=[s.hex() for s in sib], dirs=dirs))
    G["commit_open"] = co

    # (7) Lasso: 2-bit ADD/XOR/AND tables (lasso_prover.zig:312-451) + a 4-bit XOR with 11 queries
    la = []
    for kind, bits, qs in ((0, 2, [[1, 2, 3], [2, 3, 1]]), (1, 2, [[3, 2, 1], [0, 0, 0]]),
                           (2, 2, [[3, 2, 2], [1, 1, 1], [2, 3, 2]]),
                           (1, 4, [[a, (a * 7 + 3) % 16, a ^ ((a * 7 + 3) % 16)] for a in range(11)])):
        for p in (P_17, P_BB):
            tab = [[x % p for x in r] for r in build_table(kind, bits)]
            q = [[x % p for x in r] for r in qs]
            d = lasso_prove(p, tab, q)
            d.update(p=str(p), kind=kind, bits=bits, queries=[enc(r) for r in q],
                     rounds=enc(d["rounds"]), point=enc(d["point"]), final_eval=str(d["final_eval"]))
            la.append(d)
    G["lasso"] = la
    G["lasso_row_hash"] = [{"p": str(p), "fields": enc(f), "hash": str(lasso_hash_row(p, f))}
                           for p in (P_17, P_BB) for f in ([5, 7, 12], [0, 0, 0], [3, 2, 1])]

    # (8) end-to-end proofs (tests/integration_tests.zig fixtures + prover.zig:589-592 + fib loop)
    nop = lambda n: bytes([0x13, 0, 0, 0]) * n
    add_prog = bytes([0x93, 0x00, 0x50, 0x00, 0x13, 0x01, 0xA0, 0x00, 0xB3, 0x01, 0x20, 0x00, 0x13, 0x00, 0x00, 0x00])
    addi42 = bytes([0x93, 0x00, 0xA0, 0x02, 0, 0, 0, 0])
    fib, fib_in = asm_fib(12)
    pr = []
    for name, prog, entry, iregs, ms, inp in (
            ("createAddProgram", add_prog, 0x1000, None, 100, None),
            ("addi42", addi42, 0x1000, None, 100, None),
            ("nop1", nop(1), 0x1000, None, 100, None),
            ("nop4", nop(4), 0x1000, None, 100, None), ("nop8", nop(8), 0x1000, None, 100, None),
            ("nop16", nop(16), 0x1000, None, 100, None), ("nop32", nop(32), 0x1000, None, 100, None),
            ("nop64", nop(64), 0x1000, None, 100, None),
            ("nop5_entry2000", nop(5), 0x2000, None, 100, None),
            ("add_with_regs", add_prog, 0x1000, [0, 11, 22, 33], 100, None),
            ("nop64_max10", nop(64), 0x1000, None, 10, None),
            ("fib12", fib, 0x1000, None, 1 << 20, fib_in)):
        out, ns, nv, L, cols = prove(P_BB, prog, entry, iregs, ms, inp)
        ent = dict(name=name, program=prog.hex(), entry_pc=entry, initial_regs=None if iregs is None else enc(iregs),
                   max_steps=ms, input=None if inp is None else enc(inp), num_steps=ns, nv=nv, L=L,
                   proof_len=len(out), proof_sha3=sha3(out).hex())
        if len(out) <= 20000: ent["proof"] = out.hex()
        if name in ("createAddProgram", "fib12"):
            ent["witness_sha3"] = sha3(b"".join(le64(v) for c in cols for v in c)).hex()
        if name == "createAddProgram":
            ent["witness"] = [enc(c) for c in cols]
        pr.append(ent)
    G["prove"] = pr

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden.json")
    json.dump(G, open(path, "w"), indent=0, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")
    print("sumcheck [1,2,3,4] BabyBear:", sc[1]["rounds"], sc[1]["point"], sc[1]["final_eval"])


if __name__ == "__main__":
    main()
