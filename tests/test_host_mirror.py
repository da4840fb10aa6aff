"""CPU-only tests of the C++ host mirror's sequential parts (no GPU needed): RV64IM VM against the
reference's own known answers, witness columns, ZIGZ v1 (de)serializer, Verifier, host transcript --
checked against the golden fixtures and the oracle."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
from zigz_amd import ZigzError, Transcript, host, sha256, sha3_256

P = O.P_BB
HERE = os.path.dirname(os.path.abspath(__file__))
G = json.load(open(os.path.join(HERE, "golden", "golden.json")))
VM_KATS = json.load(open(os.path.join(HERE, "golden", "vm_kats.json")))


def ints(xs):
    return None if xs is None else [int(x) for x in xs]


@pytest.mark.parametrize("i", range(len(VM_KATS)))
def test_vm_reference_kats(i):
    k = VM_KATS[i]
    limit = k["run_limit"] if k["run_limit"] is not None else k["n_step_calls"]
    rc, regs, pc, steps = host.vm_run(bytes(k["program"]), k["entry_pc"], limit)
    assert rc in (0, 30)
    for r, v in k["expected_regs"].items():
        assert regs[int(r)] == int(v), (k["name"], r)
    if k["expected_pc"] is not None:
        assert pc == int(k["expected_pc"])
    orc = O.vm_run_kat(bytes(k["program"]), k["entry_pc"], limit)
    assert (rc, regs, pc, steps) == orc


@pytest.mark.parametrize("i", range(len(G["prove"])))
def test_trace_and_witness_golden(i):
    e = G["prove"][i]
    prog = bytes.fromhex(e["program"])
    t = host.Trace(prog, e["entry_pc"], ints(e["initial_regs"]), e["max_steps"], ints(e["input"]))
    assert (t.num_steps, t.num_vars, t.num_lookups) == (e["num_steps"], e["nv"], e["L"])
    cols, nv, ns = O.witness_from_program(P, prog, e["entry_pc"], ints(e["initial_regs"]), e["max_steps"], ints(e["input"]))
    w = t.witness()
    assert np.array_equal(w, cols)
    if "witness" in e:
        assert [[str(v) for v in c] for c in w] == e["witness"]
    rows = t.rows()
    assert rows.shape == (ns, 43) and np.array_equal(rows[:, :] % np.uint64(P), w[:, :ns].T)


def test_vm_random_programs_vs_oracle():
    """Random straight-line RV64IM programs (ALU/ALU-imm/*W/M/load/store/LUI/AUIPC): final registers and
    witness must equal the oracle's, including sign/shift/division corner cases."""
    import programs
    rng = np.random.default_rng(42)
    for trial in range(30):
        prog = programs.random_program(rng)
        a = host.vm_run(prog, 0x1000, 10000)
        b = O.vm_run_kat(prog, 0x1000, 10000)
        assert a == b, trial
        t = host.Trace(prog, 0x1000)
        cols, nv, ns = O.witness_from_program(P, prog, 0x1000)
        assert np.array_equal(t.witness(), cols)


@pytest.mark.parametrize("i", [i for i, e in enumerate(G["prove"]) if "proof" in e])
def test_verifier_and_serializer_golden(i):
    e = G["prove"][i]
    prog, proof = bytes.fromhex(e["program"]), bytes.fromhex(e["proof"])
    assert host.verify(proof, prog) == "Accept"          # integration_tests.zig:55-84
    assert host.reserialize(proof) == proof              # integration_tests.zig:90-127
    with pytest.raises(ZigzError) as err:                # integration_tests.zig:133-165
        host.verify(proof, prog + b"\x13\0\0\0")
    assert err.value.name == "ProgramHashMismatch"
    nv, L = e["nv"], e["L"]
    n_in = 0 if e["initial_regs"] is None else len(e["initial_regs"])
    n_out = int.from_bytes(proof[32 + 32 + 16 + 4 + 8 * n_in + 4 + 256 + 8:][:4], "little")
    off = 32 + (324 + 8 * n_in + 8 * n_out) + (40 * nv + 8) + (4 + 24 * L)
    t = bytearray(proof); t[off] ^= 1                    # tampered commitment (integration_tests.zig:251-290)
    assert host.verify(bytes(t), prog) == "RejectInvalidCommitment"
    t = bytearray(proof); t[off + 32 + 8 * nv] ^= 1      # tampered claim (integration_tests.zig:292-329)
    assert host.verify(bytes(t), prog) == "RejectInvalidCommitment"
    assert O.verify(P, bytes(t), prog) == (0, 3)
    for bad, name in ((b"ZIGY" + proof[4:], "InvalidMagicNumber"), (proof[:4] + b"\2\0\0\0" + proof[8:], "UnsupportedVersion"),
                      (proof[:8] + (17).to_bytes(8, "little") + proof[16:], "FieldMismatch"), (proof[:-5], None)):
        with pytest.raises(ZigzError) as err:
            host.verify(bad, prog)
        if name:
            assert err.value.name == name


def test_host_hashes_and_transcript():
    rng = np.random.default_rng(1)
    for n in (0, 1, 8, 55, 56, 64, 135, 136, 137, 500):
        m = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert sha3_256(m) == O.sha3_256(m) and sha256(m) == O.sha256(m)
    a, b = Transcript(), O.Transcript()
    for t in (a, b):
        t.append_bytes(b"POLY_COMMITMENTS"); t.append_field(P - 1); t.append_bytes(b"x" * 200)
    a.append_tagged_counter(b"LASSO_TABLE", 0, 300)
    for i in range(300):
        b.append_bytes(b"LASSO_TABLE"); b.append_field(i)
    assert [a.challenge() for _ in range(4)] == [b.challenge(P) for _ in range(4)]
    t = Transcript(); t.append_field(3); t.append_field(4)
    assert [str(t.challenge()), str(t.challenge())] == G["transcript_kat"]["challenges_babybear"]
    # the bulk form keeps (start + k) mod p incrementally: wrap-around at p, long tags, chunk boundaries
    for tag, start, count in ((b"LASSO_TABLE", P - 3, 7), (b"T" * 30, 5, 40), (b"", P - 1, 3), (b"LASSO_TABLE", 0, 4000)):
        a, b = Transcript(), O.Transcript()
        a.append_tagged_counter(tag, start, count)
        for i in range(count):
            b.append_bytes(tag); b.append_field((start + i) % P)
        assert a.challenge() == b.challenge(P), (tag, start, count)


def test_host_keccak_dispatch_matches_portable_code():
    """The AVX-512 single-state Keccak-f[1600] used by the host sponge (when the CPU has it) must equal the
    portable implementation and the oracle's on random states."""
    from zigz_amd._ffi import lib, u64p
    rng = np.random.default_rng(5)
    assert lib.zigz_host_keccak_impl() in (b"avx512vl", b"avx512f", b"bmi2", b"scalar")
    for _ in range(200):
        st = rng.integers(0, 2**64, 25, dtype=np.uint64)
        outs = []
        for which in (0, 1, 2, 3, 4):  # picked, scalar, bmi2, avx512f, avx512vl (unsupported ones fall back to the picked one)
            a = st.copy()
            lib.zigz_host_keccak_permute(a.ctypes.data_as(u64p), which)
            outs.append(a)
        assert all(np.array_equal(outs[0], o) for o in outs[1:])
    # one-block SHA3 through the dispatched permutation == hashlib
    import hashlib
    for n in (0, 8, 64, 135):
        m = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert sha3_256(m) == hashlib.sha3_256(m).digest()


def _has_avx512f():
    try:
        return "avx512f" in open("/proc/cpuinfo").read()
    except OSError:
        return False


@pytest.mark.skipif(not _has_avx512f(), reason="the sponge service needs AVX-512F")
def test_keccak_x8_equals_eight_single_permutations():
    from zigz_amd._ffi import lib, u64p
    rng = np.random.default_rng(8)
    for _ in range(50):
        raw = np.zeros(25 * 8 + 8, dtype=np.uint64)
        off = (-raw.ctypes.data // 8) % 8          # 64-byte aligned window
        st = raw[off:off + 200].reshape(25, 8)
        st[:] = rng.integers(0, 2**64, (25, 8), dtype=np.uint64)
        want = st.copy()
        for s in range(8):
            col = np.ascontiguousarray(want[:, s])
            lib.zigz_host_keccak_permute(col.ctypes.data_as(u64p), 1)
            want[:, s] = col
        lib.zigz_host_keccak_permute_x8(st.ctypes.data_as(u64p))
        assert np.array_equal(st, want)


@pytest.mark.skipif(not _has_avx512f(), reason="the sponge service needs AVX-512F")
def test_sponge_service_absorbs_the_same_bytes():
    """zigz_host_sponge_servers: long tagged-counter runs of several transcripts advance in lock step on a server thread
    (8-way permutation).  Every challenge must equal the sequential code's, whatever the sponge position at hand-over, the
    tag length, a counter that wraps at p, and how the jobs overlap in time."""
    import threading
    from zigz_amd._ffi import lib
    rng = np.random.default_rng(99)
    jobs = []
    for k in range(24):
        prefix = bytes(rng.integers(0, 256, int(rng.integers(0, 300)), dtype=np.uint8))
        tag = [b"LASSO_TABLE", b"", b"T" * 30, b"x" * 250, b"LASSO_TABLE"][k % 5]
        start = [0, P - 1000, 12345, P - 1][k % 4]
        rec = len(tag) + 8
        count = (65536 // rec) + int(rng.integers(1, 3000))     # >= the 64 KiB threshold of the service
        if k == 7:
            count = (65536 + 136 * 50 - len(prefix) % 136 + rec - 1) // rec  # ends close to a block boundary
        jobs.append((prefix, tag, start, count))
    # a run that ends EXACTLY on a block boundary: 136 records of 17 + 8 = 25 bytes per 25 blocks, from position 0
    jobs.append((b"", b"Q" * 17, 5, 136 * 30))

    def run(job):
        prefix, tag, start, count = job
        t = Transcript()
        t.append_bytes(prefix)
        t.append_tagged_counter(tag, start, count)
        t.append_field(7)
        return [t.challenge() for _ in range(3)]

    assert lib.zigz_host_sponge_batching() == 0
    want = [run(j) for j in jobs]
    # spot-check the sequential answers themselves against the oracle's transcript
    for j, w in list(zip(jobs, want))[:3]:
        o = O.Transcript()
        o.append_bytes(j[0])
        for i in range(j[3]):
            o.append_bytes(j[1]); o.append_field((j[2] + i) % P)
        o.append_field(7)
        assert [o.challenge(P) for _ in range(3)] == w
    for servers in (1, 3):
        lib.zigz_host_sponge_servers(servers)
        try:
            assert lib.zigz_host_sponge_batching() == 1
            got = [None] * len(jobs)

            def work(i):
                got[i] = run(jobs[i])
            th = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
            [t.start() for t in th]
            [t.join() for t in th]
            assert got == want, servers
            assert run(jobs[0]) == want[0]          # a lone job through the service
        finally:
            lib.zigz_host_sponge_servers(0)
        assert lib.zigz_host_sponge_batching() == 0
    assert run(jobs[1]) == want[1]


def test_stats_add_covers_every_field():
    """zigzh_stats_add (what zigzh_prove_trace_slots_repeat sums a lane's proofs with) adds EVERY field of zigz_kernel_stats."""
    import ctypes as C
    from zigz_amd import host
    from zigz_amd._ffi import KernelStats
    a, b = KernelStats(), KernelStats()
    for i, (f, t) in enumerate(KernelStats._fields_):
        setattr(a, f, t(i + 1).value)
        setattr(b, f, t(100 + i).value)
    host.lib.zigzh_stats_add(C.byref(a), C.byref(b))
    for i, (f, _) in enumerate(KernelStats._fields_):
        assert getattr(a, f) == 101 + 2 * i, f


@pytest.mark.parametrize("maker,arg", [("mixed_loop", 300), ("fibonacci", 40), ("add_xor_loop", 500), ("register_round_robin", 20)])
def test_compact_records_widen_back_to_the_trace(maker, arg):
    """The 32-byte and the 16-byte record (include/zigz_hip.h) are lossless restatements of the VM's 48-byte steps: compacted
    by the numpy helpers and widened again by the rule the device kernels follow (k_steps_widen / k_steps_widen16: side list by
    index, code table by (pc - base) / 4, wr_reg and the access index out of one word), they give every field of every step
    back.  What the 16-byte form refuses is checked too: a pc off the 4-byte grid, one pc with two decodings."""
    import programs
    from zigz_amd.hip import TRACE_STEP_DTYPE, compact_steps16, compact_steps32, NO_MEM_ACCESS, NO_MEM_ACCESS16
    made = getattr(programs, maker)(arg)
    prog, inp = made if isinstance(made, tuple) else (made, None)
    tr = host.Trace(prog, 0x1000, None, 1 << 20, inp)
    st, _ = tr.steps()
    has = np.isin(st["opcode"], (0x03, 0x23))
    fields = ("opcode", "rd", "rs1", "rs2", "funct3", "funct7")
    s32, mem = compact_steps32(st, has)
    back = np.zeros(len(st), dtype=TRACE_STEP_DTYPE)
    for f in ("pc", "rd_value", "wr_reg", "mem_is_read") + fields:
        back[f] = s32[f]
    back["imm"] = s32["imm"].astype(np.int64)
    ok = s32["mem_index"] != NO_MEM_ACCESS
    back["mem_addr"][ok] = mem["addr"][s32["mem_index"][ok]]
    back["mem_value"][ok] = mem["value"][s32["mem_index"][ok]]
    assert back.tobytes() == np.ascontiguousarray(st, dtype=TRACE_STEP_DTYPE).tobytes()
    s16, mem16, base, code = compact_steps16(st, has)
    assert s16.nbytes == 16 * len(st) and len(code) <= len(prog) // 4
    back = np.zeros(len(st), dtype=TRACE_STEP_DTYPE)
    ci = (s16["pc_word"] >> 2).astype(np.int64)
    back["pc"] = np.uint64(base) + (s16["pc_word"] & ~np.uint32(3)).astype(np.uint64)
    back["mem_is_read"] = s16["pc_word"] & 1
    back["wr_reg"] = s16["mem_wr"] >> 27
    back["rd_value"] = s16["rd_value"]
    for f in fields:
        back[f] = code[f][ci]
    back["imm"] = code["imm"][ci].astype(np.int64)
    mi = s16["mem_wr"] & np.uint32(NO_MEM_ACCESS16)
    ok = mi != NO_MEM_ACCESS16
    back["mem_addr"][ok] = mem16["addr"][mi[ok]]
    back["mem_value"][ok] = mem16["value"][mi[ok]]
    assert back.tobytes() == np.ascontiguousarray(st, dtype=TRACE_STEP_DTYPE).tobytes()
    bad = st.copy()
    bad["pc"][len(bad) // 2] += 2
    assert compact_steps16(bad, has) is None
    if len(st) > 8:
        bad = st.copy()
        k = int(np.flatnonzero(st["pc"] == st["pc"][len(st) // 2])[0])
        bad["rd"][k] ^= 1  # (the same pc executed later with another decoding -- or this is its only execution: then pick another)
        again = np.flatnonzero(st["pc"] == st["pc"][k])
        if len(again) > 1:
            assert compact_steps16(bad, has) is None
