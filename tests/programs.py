"""Hand-assembled RV64IM test programs (synthetic traces for the BASELINE configs)."""
import struct


def _I(op, rd, f3, rs1, imm): return ((imm & 0xfff) << 20) | (rs1 << 15) | (f3 << 12) | (rd << 7) | op
def _R(op, rd, f3, rs1, rs2, f7): return (f7 << 25) | (rs2 << 20) | (rs1 << 15) | (f3 << 12) | (rd << 7) | op
def _S(f3, rs1, rs2, imm): return (((imm >> 5) & 0x7f) << 25) | (rs2 << 20) | (rs1 << 15) | (f3 << 12) | ((imm & 31) << 7) | 0x23
def _U(op, rd, imm20): return ((imm20 & 0xfffff) << 12) | (rd << 7) | op


def _B(f3, rs1, rs2, off):
    o = off & 0x1fff
    return ((((o >> 12) & 1) << 31) | (((o >> 5) & 0x3f) << 25) | (rs2 << 20) | (rs1 << 15) | (f3 << 12) |
            (((o >> 1) & 0xf) << 8) | (((o >> 11) & 1) << 7) | 0x63)


def _pack(words): return b"".join(struct.pack("<I", w) for w in words)


def _li(rd, value):
    """LUI+ADDI load of a 32-bit positive constant (value < 2^31)."""
    hi = (value + 0x800) >> 12
    lo = value - (hi << 12)
    return [_U(0x37, rd, hi), _I(0x13, rd, 0, rd, lo)]


def add_xor_loop(iterations):
    """ADDI x1,x0,1; loop: ADD x2,x2,x1; XOR x3,x3,x2; ADDI x4,x4,1; BNE x4,x5,loop.
    num_steps = 3 + 4*iterations (2 LI words + ADDI + 4/iter), every step is a lookup step except LUI."""
    w = _li(5, iterations) + [_I(0x13, 1, 0, 0, 1)]
    w += [_R(0x33, 2, 0, 2, 1, 0), _R(0x33, 3, 4, 3, 2, 0), _I(0x13, 4, 0, 4, 1), _B(1, 4, 5, -12)]
    return _pack(w)


def register_round_robin(iterations):
    """The worst case for a run-aware build of the register columns: a loop that writes 30 DIFFERENT registers in turn
    (x1..x29 += k, x31 = counter; x30 holds the bound), so every register column changes once every 31 steps and the
    change points are spread evenly over all of them.  num_steps = 2 + 31*iterations."""
    w = _li(30, iterations)
    body = [_I(0x13, r, 0, r, r) for r in range(1, 30)] + [_I(0x13, 31, 0, 31, 1)]
    w += body + [_B(1, 31, 30, -4 * len(body))]
    return _pack(w)


def mixed_loop(iterations):
    """RV64IM mix -- ADD/XOR/MUL/DIVU/REM, SD/LD, ADDIW, branch. 12 steps per iteration."""
    w = _li(5, iterations) + [_I(0x13, 1, 0, 0, 3), _I(0x13, 6, 0, 0, 7), _I(0x13, 8, 0, 0, 0x100)]
    body = [
        _R(0x33, 2, 0, 2, 1, 0),      # ADD  x2,x2,x1
        _R(0x33, 3, 4, 3, 2, 0),      # XOR  x3,x3,x2
        _R(0x33, 7, 0, 2, 6, 1),      # MUL  x7,x2,x6
        _R(0x33, 9, 5, 7, 1, 1),      # DIVU x9,x7,x1
        _R(0x33, 10, 6, 7, 6, 1),     # REM  x10,x7,x6
        _S(3, 8, 7, 0),               # SD   x7,0(x8)
        _I(0x03, 11, 3, 8, 0),        # LD   x11,0(x8)
        _I(0x1b, 12, 0, 11, 5),       # ADDIW x12,x11,5
        _R(0x3b, 13, 0, 12, 3, 0),    # ADDW x13,x12,x3
        _I(0x13, 1, 0, 1, 2),         # ADDI x1,x1,2
        _I(0x13, 4, 0, 4, 1),         # ADDI x4,x4,1
    ]
    w += body + [_B(1, 4, 5, -4 * len(body))]
    return _pack(w)


def fibonacci(n):
    """examples/fibonacci_guest semantics -- ECALL read n; loop; 2x ECALL commit; EBREAK.
    Returns (program, input_tape).  num_steps = 6 + 5*n + 6."""
    w = [
        _I(0x13, 17, 0, 0, 2), 0x00000073,       # a7=2; ECALL read -> a0
        _I(0x13, 5, 0, 10, 0),                   # t0 = n
        _I(0x13, 6, 0, 0, 0), _I(0x13, 7, 0, 0, 1), _I(0x13, 28, 0, 0, 0),  # a=0, b=1, i=0
        _R(0x33, 29, 0, 6, 7, 0),                # loop: t4 = a+b
        _I(0x13, 6, 0, 7, 0), _I(0x13, 7, 0, 29, 0),   # a=b; b=t4
        _I(0x13, 28, 0, 28, 1), _B(1, 28, 5, -16),     # i++; BNE i,n,loop
        _I(0x13, 17, 0, 0, 1), _I(0x13, 10, 0, 6, 0), 0x00000073,  # a7=1; a0=a; ECALL commit
        _I(0x13, 10, 0, 28, 0), 0x00000073,      # a0=i; ECALL commit
        0x00100073,                              # EBREAK
    ]
    return _pack(w), [n]


def random_program(rng, n_insts=60):
    """Random straight-line RV64IM program (ALU / ALU-imm / *W / M / loads+stores at small addresses / LUI / AUIPC):
    exercises sign extension, shifts and division corner cases.  rng: numpy Generator."""
    ops_r = [(0x33, f3, f7) for f3 in range(8) for f7 in (0, 0x20, 1)] + [(0x3b, f3, f7) for f3, f7 in
             ((0, 0), (0, 0x20), (1, 0), (5, 0), (5, 0x20), (0, 1), (4, 1), (5, 1), (6, 1), (7, 1))]
    words = []
    for r in range(1, 8):
        words.append((int(rng.integers(0, 1 << 20)) << 12) | (r << 7) | 0x37)
        words.append((int(rng.integers(0, 1 << 12)) << 20) | (r << 15) | (r << 7) | 0x13)
        if rng.integers(0, 2):
            words.append((int(rng.integers(0, 64)) << 20) | (r << 15) | (1 << 12) | (r << 7) | 0x13)
    for _ in range(n_insts):
        kind = int(rng.integers(0, 5))
        rd, rs1, rs2 = (int(x) for x in rng.integers(0, 8, 3))
        if kind == 0:
            op, f3, f7 = ops_r[int(rng.integers(0, len(ops_r)))]
            if op == 0x33 and f7 == 0x20 and f3 not in (0, 5):
                f7 = 0
            words.append((f7 << 25) | (rs2 << 20) | (rs1 << 15) | (f3 << 12) | (rd << 7) | op)
        elif kind == 1:
            f3 = int(rng.integers(0, 8))
            imm = int(rng.integers(0, 1 << 12))
            if f3 == 1:
                imm &= 63
            if f3 == 5:
                imm = (imm & 63) | (0x400 if rng.integers(0, 2) else 0)
            words.append((imm << 20) | (rs1 << 15) | (f3 << 12) | (rd << 7) | 0x13)
        elif kind == 2:
            f3 = [0, 1, 5][int(rng.integers(0, 3))]
            imm = int(rng.integers(0, 1 << 12)) if f3 == 0 else (int(rng.integers(0, 32)) | (0x400 if (f3 == 5 and rng.integers(0, 2)) else 0))
            words.append((imm << 20) | (rs1 << 15) | (f3 << 12) | (rd << 7) | 0x1b)
        elif kind == 3:
            f3s = int(rng.integers(0, 4)); off = int(rng.integers(0, 64))
            words.append(((off >> 5) << 25) | (rs2 << 20) | (0 << 15) | (f3s << 12) | ((off & 31) << 7) | 0x23)
            f3l = int(rng.integers(0, 7))
            words.append((off << 20) | (0 << 15) | (f3l << 12) | (rd << 7) | 0x03)
        else:
            words.append((int(rng.integers(0, 1 << 20)) << 12) | (rd << 7) | (0x17 if rng.integers(0, 2) else 0x37))
    return _pack(words)


def straight_line_program(seed, n_insts):
    """A program that never loops: n_insts random RV64IM instructions over all 31 registers (OP, OP-IMM, M extension, LUI,
    SD / LD at small addresses off x0), each executed exactly once.  Vectorised (a 2^20-step trace takes ~0.1 s to
    generate); num_steps = n_insts.  The worst case for the content-addressed group (nothing repeats) and a hard one for the
    run-aware register levels (one of 31 registers changes at almost every step)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    kind = rng.integers(0, 6, n_insts)
    rd, rs1, rs2 = (rng.integers(1, 32, n_insts) for _ in range(3))
    f3 = rng.integers(0, 8, n_insts)
    imm = rng.integers(0, 1 << 12, n_insts)
    sh = np.where(f3 == 1, imm & 63, np.where(f3 == 5, (imm & 63) | (rng.integers(0, 2, n_insts) << 10), imm))
    off = rng.integers(0, 32, n_insts) * 8
    w = np.select(
        [kind == 0, kind == 1, kind == 2, kind == 3, kind == 4],
        [(rs2 << 20) | (rs1 << 15) | (f3 << 12) | (rd << 7) | 0x33,                       # OP, funct7 = 0
         (sh << 20) | (rs1 << 15) | (f3 << 12) | (rd << 7) | 0x13,                        # OP-IMM
         (1 << 25) | (rs2 << 20) | (rs1 << 15) | (f3 << 12) | (rd << 7) | 0x33,           # MUL .. REMU
         (rng.integers(0, 1 << 20, n_insts) << 12) | (rd << 7) | 0x37,                    # LUI
         ((off >> 5) << 25) | (rs2 << 20) | (3 << 12) | ((off & 31) << 7) | 0x23],        # SD rs2, off(x0)
        default=(off << 20) | (3 << 12) | (rd << 7) | 0x03)                               # LD rd, off(x0)
    return w.astype("<u4").tobytes()
