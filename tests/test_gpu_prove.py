"""GPU end-to-end tests (-m gpu): Prover.prove through the C++ host mirror + HIP backend must emit
proofs byte-identical to the oracle's (and to the committed golden proofs); Verifier.verify accepts."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as O
import programs

pytestmark = pytest.mark.gpu
P = O.P_BB
HERE = os.path.dirname(os.path.abspath(__file__))
G = json.load(open(os.path.join(HERE, "golden", "golden.json")))


@pytest.fixture(scope="module")
def ctx():
    import zigz_amd
    c = zigz_amd.Context(0)
    yield c
    c.close()


def ints(xs):
    return None if xs is None else [int(x) for x in xs]


@pytest.mark.parametrize("i", range(len(G["prove"])))
def test_prove_golden_bytes(ctx, i):
    from zigz_amd import host
    e = G["prove"][i]
    prog = bytes.fromhex(e["program"])
    proof, ns = host.prove(ctx, prog, e["entry_pc"], ints(e["initial_regs"]), e["max_steps"], ints(e["input"]))
    assert ns == e["num_steps"] and len(proof) == e["proof_len"]
    assert hashlib.sha3_256(proof).hexdigest() == e["proof_sha3"]
    if "proof" in e:
        assert proof.hex() == e["proof"]
    oproof, _ = O.prove(P, prog, e["entry_pc"], ints(e["initial_regs"]), e["max_steps"], ints(e["input"]))
    assert proof == oproof
    assert host.verify(proof, prog) == "Accept" and O.verify(P, proof, prog) == (0, 0)
    # same proof from device-resident witness columns
    t = host.Trace(prog, e["entry_pc"], ints(e["initial_regs"]), e["max_steps"], ints(e["input"]))
    N = 1 << t.num_vars
    stride = max(N, 4)
    d = ctx.dev_alloc(43 * stride * 4)
    try:
        t.witness_to_device(ctx, d, stride)
        assert t.prove(ctx, d, stride) == proof
        assert t.prove(ctx) == proof
        assert t.prove(ctx, d, stride, want_bytes="borrow").tobytes() == proof  # overlapped serialisation path
        assert t.prove(ctx, want_bytes="borrow").tobytes() == proof
    finally:
        ctx.dev_free(d)


def test_prove_errors(ctx):
    import zigz_amd
    from zigz_amd import host
    with pytest.raises(zigz_amd.ZigzError) as e:
        host.prove(ctx, b"\0\0\0\0")
    assert e.value.name == "EmptyTrace"
    with pytest.raises(zigz_amd.ZigzError) as e:  # CSRRW: UnimplementedSYSTEM propagates out of prove
        host.prove(ctx, (0x00001073).to_bytes(4, "little"))
    assert e.value.name == "UnimplementedSYSTEM"


@pytest.mark.parametrize("maker,arg", [("fibonacci", 50), ("add_xor_loop", 300), ("mixed_loop", 200), ("fibonacci", 1500)])
def test_prove_programs_vs_oracle(ctx, maker, arg):
    from zigz_amd import host
    r = getattr(programs, maker)(arg)
    prog, inp = r if isinstance(r, tuple) else (r, None)
    proof, ns = host.prove(ctx, prog, 0x1000, None, 1 << 20, inp)
    oproof, ons = O.prove(P, prog, 0x1000, None, 1 << 20, inp)
    assert ns == ons and proof == oproof
    assert host.verify(proof, prog) == "Accept"


@pytest.mark.parametrize("maker,arg", [("mixed_loop", 300), ("fibonacci", 2500)])
def test_prove_from_pinned_compact_trace_async(ctx, maker, arg):
    """The service path of bench.py's PCIe-inclusive leg: pinned compact trace -> asynchronous upload + witness kernels
    -> proof on the resident columns, three times over the same buffers; bytes identical to the oracle's proof."""
    from zigz_amd import host
    r = getattr(programs, maker)(arg)
    prog, inp = r if isinstance(r, tuple) else (r, None)
    oproof, ons = O.prove(P, prog, 0x1000, None, 1 << 20, inp)
    tr = host.Trace(prog, 0x1000, None, 1 << 20, inp)
    N = 1 << tr.num_vars
    d = ctx.dev_alloc(43 * max(N, 4) * 4)
    try:
        tr.pin(ctx)
        for _ in range(3):
            tr.witness_to_device(ctx, d, max(N, 4), wait=False)
            proof = tr.prove(ctx, d, max(N, 4), want_bytes=True)
            assert tr.num_steps == ons and proof == oproof
        # the pipelined form of bench.py's PCIe-inclusive leg: a SECOND context (its own stream) uploads the next proof's
        # trace from the same pinned records while the first proves
        import zigz_amd
        up = zigz_amd.Context(0)
        d2 = up.dev_alloc(43 * max(N, 4) * 4)
        try:
            for _ in range(3):
                tr.witness_to_device(up, d2, max(N, 4), wait=False)
                assert tr.prove(ctx, d, max(N, 4), want_bytes=True) == oproof
                up.synchronize()
                d, d2 = d2, d
        finally:
            up.dev_free(d2)
            up.close()
    finally:
        ctx.dev_free(d)
        del tr


def test_prove_config2_fibonacci_2_16_bit_exact(ctx):
    """BASELINE config 2: fibonacci proved end-to-end at a 2^16 trace, byte-identical to the CPU oracle
    (literal reference algorithm: naive eval twice per column, recompute-on-open Merkle)."""
    from zigz_amd import host
    n = (65536 - 12) // 5  # num_steps = 12 + 5n <= 2^16 < 2*num_steps
    prog, inp = programs.fibonacci(n)
    proof, ns = host.prove(ctx, prog, 0x1000, None, 1 << 20, inp)
    assert 32768 < ns <= 65536
    oproof, ons = O.prove(P, prog, 0x1000, None, 1 << 20, inp)
    assert ons == ns and hashlib.sha3_256(proof).digest() == hashlib.sha3_256(oproof).digest() and proof == oproof
    assert host.verify(proof, prog) == "Accept"


def test_prove_2_20_properties(ctx):
    """BASELINE config 3 (synthetic RV64I ADD/XOR loop, 2^20 trace): the literal oracle would need minutes, so
    check size-independent properties: both verifiers accept; roots/values/paths agree with the oracle's
    keep-levels commit on the same witness; transcript-derived points match; proof size formula."""
    from zigz_amd import host
    iters = ((1 << 20) - 3) // 4
    prog = programs.add_xor_loop(iters)
    t = host.Trace(prog, 0x1000, None, 1 << 21)
    assert t.num_vars == 20 and (1 << 19) < t.num_steps <= (1 << 20)
    proof = t.prove(ctx)
    assert len(proof) == O.proof_size(20, 0, 0, t.num_lookups)
    assert host.verify(proof, prog) == "Accept" and O.verify(P, proof, prog) == (0, 0)
    # oracle: same transcript prefix, then generateCommitments (keep-levels variant, equal to the literal one)
    cols = t.witness()
    tr = O.Transcript()
    tr.append_bytes(hashlib.sha256(prog).digest()); tr.append_field(0x1000 % P)
    tr.append_bytes(b"SUMCHECK_BEGIN"); tr.append_field(t.num_steps % P); tr.append_field(20)
    cpoint = []
    for _ in range(20):
        for _ in range(4):
            tr.append_field(0)
        cpoint.append(tr.challenge(P))
    tr.append_bytes(b"LASSO_BEGIN")
    O.lib.orc_tr_append_bytes  # (loop below is the literal schedule)
    for i in range(t.num_lookups):
        tr.append_bytes(b"LASSO_TABLE"); tr.append_field(i % P)
    exp = O.generate_commitments(P, tr, cols, fast=True)
    off = 32 + 324 + (40 * 20 + 8) + (4 + 24 * t.num_lookups)
    for c in range(43):
        rec = proof[off + c * (68 + 41 * 20): off + (c + 1) * (68 + 41 * 20)]
        assert rec[:32] == exp["roots"][c].tobytes()
        pts = np.frombuffer(rec[32:32 + 160], dtype="<u8")
        assert np.array_equal(pts, exp["points"][c])
        val, val2, idx, leaf = (int(x) for x in np.frombuffer(rec[192:224], dtype="<u8"))
        assert val == val2 == int(exp["values"][c]) and idx == int(exp["indices"][c]) and leaf == int(exp["leaves"][c])
        assert rec[228:228 + 640] == exp["siblings"][c].tobytes() and rec[868:888] == exp["dirs"][c].tobytes()


def test_mirror_classes(ctx):
    """SumcheckProver / LassoProver / CommitmentScheme through the C++ mirror classes."""
    from zigz_amd import host
    for nv in (1, 4, 12):
        ev = O.splitmix64_field(31 + nv, 1 << nv)
        r, pt, fe = O.sumcheck_prove(P, ev)
        assert host.sumcheck_prove_bytes(ctx, ev) == O.sumcheck_to_bytes(r, pt, fe)
    ev12 = [(i + 1) % P for i in range(1 << 12)]  # BASELINE config 1 input
    r, pt, fe = O.sumcheck_prove(P, ev12)
    assert host.sumcheck_prove_bytes(ctx, ev12) == O.sumcheck_to_bytes(r, pt, fe)
    for e in G["lasso"]:
        if e["p"] != str(P):
            continue
        q = np.array([[int(x) for x in r] for r in e["queries"]], dtype=np.uint64)
        sb, qc, tc, nl = host.lasso_prove_table(ctx, e["kind"], e["bits"], q)
        exp = O.sumcheck_to_bytes([int(x) for x in e["rounds"]], [int(x) for x in e["point"]], int(e["final_eval"]))
        assert sb == exp and qc.hex() == e["query_commit"] and tc.hex() == e["table_commit"] and nl == len(q)
    # mapping index 14 for (3,2)->1 in the 2-bit XOR table plus a second query (lasso_prover.zig:352-382)
    sb, qc, tc, nl = host.lasso_prove_table(ctx, 1, 2, [[3, 2, 1], [0, 0, 0]], mapping=[14, 0])
    assert nl == 2
    ev = O.splitmix64_field(9, 64)
    pt = O.splitmix64_field(10, 6)
    root, val, idx, ok = host.commit_open_verify(ctx, ev, pt)
    assert ok and root == O.merkle_build(ev)[0] and val == O.mle_eval(P, ev, pt) and idx == int(pt[0]) % 64


def test_concurrent_proofs_on_one_gpu(ctx):
    """Several proofs share one GPU (one host thread + one context/stream each, as bench.py --batch does):
    every proof must still be byte-identical to the oracle's."""
    import threading
    import zigz_amd
    from zigz_amd import host
    jobs = [programs.fibonacci(40 + 7 * k) for k in range(4)] + [(programs.mixed_loop(30 + k), None) for k in range(2)]
    expect = [O.prove(P, p, 0x1000, None, 1 << 20, i)[0] for p, i in jobs]
    got = [None] * len(jobs)
    errs = []

    def work(k):
        try:
            c = zigz_amd.Context(0)
            for _ in range(3):
                t = host.Trace(jobs[k][0], 0x1000, None, 1 << 20, jobs[k][1])
                got[k] = t.prove(c, want_bytes="borrow").tobytes()
                assert got[k] == expect[k]
            c.close()
        except Exception as e:  # noqa: BLE001
            errs.append((k, repr(e)))

    th = [threading.Thread(target=work, args=(k,)) for k in range(len(jobs))]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    assert got == expect


def test_concurrent_proofs_through_the_sponge_service(ctx):
    """zigz_host_sponge_servers: the transcripts of proofs in flight advance in lock step on server threads (8-way AVX-512
    permutation).  Ten concurrent proofs with >= 3 500 lookup steps each (long enough to be handed to the service), two
    rounds: proof bytes identical to the oracle's."""
    import threading
    import zigz_amd
    from zigz_amd import host
    from zigz_amd._ffi import lib
    if " avx512f" not in open("/proc/cpuinfo").read():
        pytest.skip("the sponge service needs AVX-512F")
    jobs = [programs.fibonacci(750 + 130 * k) for k in range(6)] + [(programs.mixed_loop(400 + 50 * k), None) for k in range(4)]
    expect = [O.prove(P, p, 0x1000, None, 1 << 20, i)[0] for p, i in jobs]
    errs = []

    def work(k):
        try:
            c = zigz_amd.Context(0)
            for _ in range(2):
                t = host.Trace(jobs[k][0], 0x1000, None, 1 << 20, jobs[k][1])
                assert t.num_lookups * 19 >= 65536
                assert t.prove(c, want_bytes="borrow").tobytes() == expect[k]
            c.close()
        except Exception as e:  # noqa: BLE001
            errs.append((k, repr(e)))

    lib.zigz_host_sponge_servers(2)
    try:
        assert lib.zigz_host_sponge_batching() == 1
        th = [threading.Thread(target=work, args=(k,)) for k in range(len(jobs))]
        [t.start() for t in th]
        [t.join() for t in th]
    finally:
        lib.zigz_host_sponge_servers(0)
    assert not errs, errs


def test_prove_random_programs_vs_oracle(ctx):
    """25 random straight-line RV64IM programs (random lengths => ragged num_steps, loads/stores, initial registers):
    proof bytes from the HIP path == proof bytes from the oracle."""
    from zigz_amd import host
    rng = np.random.default_rng(2024)
    for trial in range(25):
        prog = programs.random_program(rng, n_insts=int(rng.integers(1, 200)))
        iregs = None if trial % 3 else [0] + [int(x) for x in rng.integers(0, 2**63, size=int(rng.integers(1, 12)), dtype=np.int64)]
        proof, ns = host.prove(ctx, prog, 0x1000 + 4 * trial, iregs, 1 << 20)
        oproof, ons = O.prove(P, prog, 0x1000 + 4 * trial, iregs, 1 << 20)
        assert ns == ons and proof == oproof, trial
        assert host.verify(proof, prog) == "Accept"


def test_prove_with_run_aware_merkle_is_byte_identical(ctx):
    """The run-aware Merkle levels are an optimisation only: same proof bytes with the option on every column, on the
    register columns (Prover's default), and off (2^17 ADD/XOR trace: most registers never change)."""
    from zigz_amd import host
    prog = programs.add_xor_loop(((1 << 17) - 3) // 4)
    t = host.Trace(prog, 0x1000, None, 1 << 18)
    assert t.num_vars == 17
    a = t.prove(ctx)
    st_default = ctx.stats()
    os.environ["ZIGZ_DENSE_MERKLE"] = "1"
    try:
        dense = t.prove(ctx)
        st_dense = ctx.stats()
        ctx.set_option("merkle_dedup", 1)
        b = t.prove(ctx)
        st = ctx.stats()
    finally:
        ctx.set_option("merkle_dedup", 0)
        os.environ.pop("ZIGZ_DENSE_MERKLE", None)
    assert a == b == dense
    assert st_dense["run_aware_columns"] == 0 and st_dense["keccak_permutations"] == 43 * ((2 << 17) - 1)
    # default: 31 registers + 2 memory columns run-aware, the 10 instruction-determined columns content-addressed
    assert st_default["run_aware_columns"] == 33 and st_default["cons_columns"] == 10 and st["run_aware_columns"] == 43
    assert st["run_aware_hashed"] < st["run_aware_dense_nodes"] // 2
    assert st["keccak_permutations"] == 43 * ((2 << 17) - 1) - (st["run_aware_dense_nodes"] - st["run_aware_hashed"])
    assert host.verify(b, prog) == "Accept"


def _run_config(config, timeout):
    """tests/run_config.py in its own process (frees the 10-45 GB of HBM it uses when it exits)."""
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "run_config.py"), "--config", str(config), "--check-cols", "1"],
                       capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_config4_mixed_rv64im_2_22_full_size():
    """BASELINE config 4 at full size (RV64IM mixed loop with MUL/DIV/REM/LD/SD, 2^22 trace) on one GPU: both
    verifiers accept, exact proof size, and for a sampled column the oracle re-derives root, opened leaf and value."""
    out = _run_config(4, 600)
    assert out["nv"] == 22 and out["checked_columns_vs_oracle"] == 1
    assert out["proof_bytes"] == O.proof_size(22, 0, 0, out["lookups"])


@pytest.mark.skipif(os.environ.get("ZIGZ_TEST_SKIP_FULL") == "1", reason="2^24 run (45 GB HBM, 12 GB host, ~25 s) skipped by request")
def test_config5_fibonacci_2_24_full_size():
    """BASELINE config 5 at full size (fibonacci guest semantics, 2^24 trace, 403 MB proof), same checks as config 4."""
    out = _run_config(5, 1200)
    assert out["nv"] == 24 and out["checked_columns_vs_oracle"] == 1


def test_prove_through_gpu_slots_is_byte_identical(ctx):
    """The service form (GpuSlots, csrc/host/zigz_host.hpp): many proving threads, few contexts -- a thread runs
    its transcript holding nothing on the GPU and takes one of K slots only for begin -> roots -> challenges -> open_all ->
    end (prover.zig:405-424: the trees do not depend on the transcript, the points do).  9 threads x 3 proofs each over 2
    slots, on resident columns and with the witness built inside the slot from the pinned compact trace, every proof against
    the oracle's bytes -- including looping traces >= 2^15 steps, where the slots' contexts learn list capacities from each
    other's traces (a build that runs out of room is repeated inside the slot)."""
    import threading
    from zigz_amd import host
    cases = [("fibonacci", 50), ("add_xor_loop", 300), ("mixed_loop", 200), ("fibonacci", 1500), ("add_xor_loop", 9000),
             ("mixed_loop", 3000), ("register_round_robin", 1100), ("add_xor_loop", 8190), ("add_xor_loop", 5)]
    traces, want = [], []
    for maker, arg in cases:
        r = getattr(programs, maker)(arg)
        prog, inp = r if isinstance(r, tuple) else (r, None)
        tr = host.Trace(prog, 0x1000, None, 1 << 20, inp)
        tr.pin(ctx)
        traces.append(tr)
        want.append(hashlib.sha3_256(O.prove(P, prog, 0x1000, None, 1 << 20, inp)[0]).hexdigest())
    slots = host.Slots(0, 2)
    bufs = []
    for tr in traces:
        N = max(1 << tr.num_vars, 4)
        d = ctx.dev_alloc(43 * N * 4)
        tr.witness_to_device(ctx, d, N)
        bufs.append((d, N))
    errs, got = [], {}

    def worker(i):
        try:
            tr, (d, N) = traces[i], bufs[i]
            for rep, dc in enumerate((d, None, d)):
                proof, st, _ = tr.prove_slots(slots, dc, N)
                got[(i, rep)] = hashlib.sha3_256(proof.tobytes()).hexdigest()
                assert st["keccak_permutations"] > 0
                t = host.last_timings()
                assert t["in_slot"] > 0 and t["commit_begin"] <= t["in_slot"]
        except Exception as e:  # noqa: BLE001
            errs.append((i, repr(e)))
    th = [threading.Thread(target=worker, args=(i,)) for i in range(len(cases))]
    [t.start() for t in th]
    [t.join() for t in th]
    try:
        assert not errs, errs
        for i in range(len(cases)):
            for rep in range(3):
                assert got[(i, rep)] == want[i], (cases[i], rep)
        # one slot context after the run: no job left active, options as its owner left them (0)
        for c in slots.contexts():
            assert c.get_option("run_aware_mask") == 0 and c.get_option("cons_group_mask") == 0
    finally:
        for d, _ in bufs:
            ctx.dev_free(d)
        slots.close()
        del traces


_ORACLE_SHA3 = {}  # (program, pc, input) -> sha3 of the oracle's proof (6 s per 2^16 trace: once per session)


def _oracle_sha3(prog, pc, regs, ms, inp):
    key = (bytes(prog), pc, tuple(regs or ()), ms, tuple(inp or ()))
    if key not in _ORACLE_SHA3:
        _ORACLE_SHA3[key] = hashlib.sha3_256(O.prove(P, prog, pc, regs, ms, inp)[0]).hexdigest()
    return _ORACLE_SHA3[key]


@pytest.mark.parametrize("max_batch,linger_us", [(4, 2000.0), (16, 300.0)])
def test_prove_small_traces_in_shared_commit_jobs(ctx, max_batch, linger_us):
    """proofs of small traces that reach their GPU phase together share ONE commit job (GpuBatcher +
    zigz_commit_begin_batch); every proof keeps its own transcript and must come out byte-identical to the oracle's.
    The reference's own sizes -- createAddProgram (4 steps) and NOP programs of 4 .. 64 steps
    (tests/integration_tests.zig:171-206) --, the golden programs, 2^10 .. 2^14 (flat batches) and 2^15 / 2^16 (arena batches:
    BASELINE config 2's size, looping and straight-line traces side by side), 24 threads x 2 rounds over 3 slots."""
    import threading
    from zigz_amd import host
    nop = (0x00000013).to_bytes(4, "little")
    progs = [(bytes.fromhex(e["program"]), e["entry_pc"], ints(e["initial_regs"]), e["max_steps"], ints(e["input"])) for e in G["prove"][:6]]
    progs += [(nop * n, 0x1000, None, 1 << 20, None) for n in (4, 8, 16, 32, 64)]
    for maker, arg in (("add_xor_loop", 200), ("add_xor_loop", 255), ("mixed_loop", 300), ("add_xor_loop", 3000), ("mixed_loop", 1200),
                       ("add_xor_loop", 8190), ("add_xor_loop", 8000), ("register_round_robin", 1000), ("add_xor_loop", 16000),
                       ("mixed_loop", 5000), ("straight_line_program", None), ("add_xor_loop", 15000), ("fibonacci", 1500)):
        r = programs.straight_line_program(5, 60000) if maker == "straight_line_program" else getattr(programs, maker)(arg)
        prog, inp = r if isinstance(r, tuple) else (r, None)
        progs.append((prog, 0x1000, None, 1 << 20, inp))
    traces, want, bufs = [], [], []
    for prog, pc, regs, ms, inp in progs:
        tr = host.Trace(prog, pc, regs, ms, inp)
        traces.append(tr)
        want.append(_oracle_sha3(prog, pc, regs, ms, inp))
        N = max(1 << tr.num_vars, 4)
        d = ctx.dev_alloc(43 * N * 4)
        tr.witness_to_device(ctx, d, N)
        bufs.append((d, N))
    slots = host.Slots(0, 3)
    slots.set_batching(max_batch, linger_us, 17)
    errs, got = [], {}

    def worker(i):
        try:
            for rep in range(2):
                proof, st, _ = traces[i].prove_slots(slots, bufs[i][0], bufs[i][1])
                got[(i, rep)] = hashlib.sha3_256(proof.tobytes()).hexdigest()
        except Exception as e:  # noqa: BLE001
            errs.append((i, repr(e)))
    th = [threading.Thread(target=worker, args=(i,)) for i in range(len(progs))]
    [t.start() for t in th]
    [t.join() for t in th]
    try:
        assert not errs, errs
        for i in range(len(progs)):
            for rep in range(2):
                assert got[(i, rep)] == want[i], (i, rep, traces[i].num_steps)
        for c in slots.contexts():
            assert c.get_option("run_aware_mask") == 0 and c.get_option("cons_group_mask") == 0
    finally:
        for d, _ in bufs:
            ctx.dev_free(d)
        slots.close()
        del traces
