"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the
same seeded inputs, against the committed golden fixtures, and -- at BASELINE sizes -- through
size-independent properties.  Bar: bit-exact (integer field / byte work)."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

P = O.P_BB
HERE = os.path.dirname(os.path.abspath(__file__))
G = json.load(open(os.path.join(HERE, "golden", "golden.json")))


@pytest.fixture(scope="module")
def ctx():
    import zigz_amd
    c = zigz_amd.Context(0)  # raises NoDevice (no CPU fallback) when the HIP path cannot run
    yield c
    c.close()


def ints(xs):
    return [int(x) for x in xs]


def rnd(seed, n):
    return O.splitmix64_field(seed, n)


# ---------------------------------------------------------------- K1/K2/K3/K4: Multilinear seams
@pytest.mark.parametrize("nv", [1, 2, 3, 5, 8, 11, 12, 13, 14, 16, 18])
def test_mle_bind_round_sum(ctx, nv):
    ev = rnd(100 + nv, 1 << nv)
    for r in (0, 1, P - 1, int(rnd(7, 1)[0])):
        got = ctx.mle_bind(ev, r)
        assert np.array_equal(got, O.mle_partial_eval(P, ev, r)), (nv, r)
    assert ctx.mle_round_poly(ev) == O.mle_round_poly(P, ev)
    assert ctx.mle_sum(ev) == O.mle_sum(P, ev)


def test_mle_edge_values(ctx):
    # all p-1 (maximum sums), all zero, and the reference's own [1,2,3,4] table
    for nv in (1, 4, 13, 16):
        ev = np.full(1 << nv, P - 1, dtype=np.uint64)
        assert ctx.mle_round_poly(ev) == O.mle_round_poly(P, ev)
        assert ctx.mle_sum(ev) == O.mle_sum(P, ev)
        assert np.array_equal(ctx.mle_bind(ev, P - 1), O.mle_partial_eval(P, ev, P - 1))
        z = np.zeros(1 << nv, dtype=np.uint64)
        assert ctx.mle_sum(z) == 0 and ctx.mle_round_poly(z) == [0, 0]
    assert list(ctx.mle_bind([1, 2, 3, 4], 0)) == [1, 2]  # multilinear.zig:436-463
    assert ctx.mle_sum([1, 2, 3, 4]) == 10 and ctx.mle_round_poly([1, 2, 3, 4]) == [3, 4]  # :465-506 (values < 17)


@pytest.mark.parametrize("nv", [0, 1, 2, 3, 6, 10, 12, 13, 15])
def test_mle_eval(ctx, nv):
    ev = rnd(200 + nv, 1 << nv)
    for s in range(3):
        pt = rnd(300 + 10 * nv + s, nv)
        assert ctx.mle_eval(ev, pt) == O.mle_eval(P, ev, pt), (nv, s)
    # boolean points select table entries LSB-first (multilinear.zig:383-413)
    if nv and nv <= 6:
        for idx in range(1 << nv):
            pt = [(idx >> v) & 1 for v in range(nv)]
            assert ctx.mle_eval(ev, pt) == int(ev[idx])


def test_mle_errors(ctx):
    import zigz_amd
    E = zigz_amd.errors
    cases = [(lambda: ctx.mle_bind([1, 2, 3], 1), E.LENGTH_NOT_POWER_OF_TWO),
             (lambda: ctx.mle_bind([], 1), E.EMPTY_EVALUATIONS),
             (lambda: ctx.mle_bind([5], 1), E.NO_VARIABLES_TO_FIX),
             (lambda: ctx.mle_round_poly([5]), E.NO_VARIABLES),
             (lambda: ctx.mle_eval([1, 2, 3, 4], [1]), E.WRONG_NUMBER_OF_VARIABLES),
             (lambda: ctx.sumcheck_prove([5]), E.NO_VARIABLES),
             (lambda: ctx.sumcheck_prove([1, 2, 3, 4], [1]), E.WRONG_NUMBER_OF_CHALLENGES),
             (lambda: ctx.mle_sum([P, 1]), E.NOT_CANONICAL),
             (lambda: ctx.mle_bind([1, 2], P), E.NOT_CANONICAL)]
    for fn, code in cases:
        with pytest.raises(zigz_amd.ZigzError) as e:
            fn()
        assert e.value.code == code, e.value
    assert ctx.mle_eval([9], []) == 9


# ---------------------------------------------------------------- A5: sumcheck
@pytest.mark.parametrize("nv", [1, 2, 3, 4, 7, 10, 12, 13, 14, 15, 17])
def test_sumcheck_vs_oracle(ctx, nv):
    ev = rnd(400 + nv, 1 << nv)
    r, pt, fe = ctx.sumcheck_prove(ev)
    r0, pt0, fe0 = O.sumcheck_prove(P, ev)
    assert np.array_equal(r, r0) and np.array_equal(pt, pt0) and fe == fe0
    assert O.sumcheck_to_bytes(r, pt, fe) == O.sumcheck_to_bytes(r0, pt0, fe0)
    chs = rnd(500 + nv, nv)
    r, pt, fe = ctx.sumcheck_prove(ev, chs)
    r0, pt0, fe0 = O.sumcheck_prove(P, ev, chs)
    assert np.array_equal(r, r0) and np.array_equal(pt, pt0) and fe == fe0


@pytest.mark.parametrize("i", [i for i, e in enumerate(G["sumcheck"]) if e["p"] == str(P)])
def test_sumcheck_golden(ctx, i):
    e = G["sumcheck"][i]
    if "evals" in e:
        ev = ints(e["evals"])
    elif e.get("evals_gen") == "iota1":
        ev = [(k + 1) % P for k in range(1 << e["nv"])]  # config 1: 2^12-entry BabyBear MLE
    else:
        ev = rnd(e["evals_gen"]["splitmix64_seed"], 1 << e["nv"])
    r, pt, fe = ctx.sumcheck_prove(ev)
    assert [str(x) for x in r] == e["rounds"] and [str(x) for x in pt] == e["point"] and str(fe) == e["final_eval"]
    assert hashlib.sha3_256(O.sumcheck_to_bytes(r, pt, fe)).hexdigest() == e["bytes_sha3"]


def test_sumcheck_large_properties(ctx):
    """2^20 and 2^22 elements: full oracle sumcheck would take long; check the size-independent
    properties instead: round-0 sums, g_i(0)+g_i(1) chain, final_eval == eval(reverse(point))."""
    for nv in (20, 22):
        ev = rnd(600 + nv, 1 << nv)
        r, pt, fe = ctx.sumcheck_prove(ev)
        total = int(ev.sum(dtype=np.uint64) % np.uint64(P)) if nv <= 30 else None
        s0 = int(r[0]); s1 = (int(r[0]) + int(r[1])) % P
        assert (s0 + s1) % P == total
        assert [s0, int(r[1])] == O.mle_round_poly(P, ev)
        claim = total
        for k in range(nv):
            c0, c1 = int(r[2 * k]), int(r[2 * k + 1])
            assert (2 * c0 + c1) % P == claim  # g(0) + g(1)
            claim = (c0 + c1 * int(pt[k])) % P
        assert claim == fe
        # transcript: challenges re-derived by the oracle's transcript
        t = O.Transcript()
        for k in range(nv):
            t.append_field(int(r[2 * k])); t.append_field(int(r[2 * k + 1]))
            assert t.challenge(P) == int(pt[k])
        assert ctx.mle_eval(ev, list(pt)[::-1]) == fe  # SURVEY s0 fact 7 (GPU eval, itself oracle-checked above)


# ---------------------------------------------------------------- A7/A8/A9: Merkle + commitment scheme
@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 8, 33, 256, 257, 512, 1000, 1024, 4096, 5000])
def test_merkle_vs_oracle(ctx, n):
    import zigz_amd
    vals = rnd(700 + n, n)
    t = zigz_amd.SimpleMerkleTree.build(ctx, vals)
    root, h = O.merkle_build(vals)
    assert t.getRoot() == root and t.height == h
    for idx in sorted({0, n - 1, n // 2, (n * 7) // 11}):
        o = t.open(idx)
        sib, dirs, leaf = O.merkle_open(vals, idx)
        assert o["siblings"] == sib and o["directions"] == dirs and o["value"] == leaf
        assert O.merkle_verify(root, o["value"], o["siblings"], o["directions"])
    with pytest.raises(zigz_amd.ZigzError) as e:
        t.open(n)
    assert e.value.code == zigz_amd.errors.INDEX_OUT_OF_BOUNDS
    t.deinit()


@pytest.mark.parametrize("i", range(len(G["merkle"])))
def test_merkle_golden(ctx, i):
    import zigz_amd
    e = G["merkle"][i]
    vals = ints(e["values"])
    t = zigz_amd.SimpleMerkleTree.build(ctx, vals)
    assert t.getRoot().hex() == e["root"] and t.height == e["height"]
    for o in e["openings"]:
        g = t.open(o["index"])
        assert [g["siblings"][32 * k:32 * k + 32].hex() for k in range(t.height)] == o["siblings"]
        assert list(g["directions"]) == o["dirs"]
    t.deinit()


def test_merkle_errors(ctx):
    import zigz_amd
    with pytest.raises(zigz_amd.ZigzError) as e:
        zigz_amd.SimpleMerkleTree.build(ctx, [])
    assert e.value.code == zigz_amd.errors.EMPTY_VALUES


@pytest.mark.parametrize("i", range(len(G["commit_open"])))
def test_commit_open_golden(ctx, i):
    import zigz_amd
    e = G["commit_open"][i]
    ev = rnd(e["evals_seed"], 1 << e["nv"])
    root, tree = zigz_amd.CommitmentScheme.commit(ctx, ev)
    assert root.hex() == e["root"]
    o = zigz_amd.CommitmentScheme.open(ctx, ev, tree, ints(e["point"]))
    assert str(o["value"]) == e["value"] and o["index"] == e["index"] and str(o["leaf"]) == e["leaf"]
    assert [o["siblings"][32 * k:32 * k + 32].hex() for k in range(e["nv"])] == e["siblings"]
    assert list(o["directions"]) == e["dirs"]
    with pytest.raises(zigz_amd.ZigzError) as err:
        zigz_amd.CommitmentScheme.open(ctx, ev, tree, ints(e["point"])[:-1])
    assert err.value.code == zigz_amd.errors.POINT_DIMENSION_MISMATCH
    tree.deinit()


# ---------------------------------------------------------------- A10: generateCommitments (43 columns)
@pytest.mark.parametrize("nv", [0, 1, 2, 3, 5, 9, 10, 13])
def test_commit_job_vs_oracle(ctx, nv):
    import zigz_amd
    N = 1 << nv
    cols = rnd(0x5A49475A + nv, 43 * N).reshape(43, N)
    # oracle: literal generateCommitments continuing a transcript
    to = O.Transcript(); to.append_bytes(b"prefix")
    exp = O.generate_commitments(P, to, cols, fast=(nv > 10))
    # HIP path: begin -> roots -> host transcript -> open_all
    job = zigz_amd.CommitJob(ctx, cols=cols)
    roots = job.roots()
    assert np.array_equal(roots, exp["roots"])
    tg = zigz_amd.Transcript(); tg.append_bytes(b"prefix")
    tg.append_bytes(b"POLY_COMMITMENTS")
    for c in range(43):
        tg.append_bytes(roots[c].tobytes())
    points = np.array([[tg.challenge() for _ in range(nv)] for _ in range(43)], dtype=np.uint64).reshape(43, nv)
    assert np.array_equal(points, exp["points"])
    got = job.open_all(points)
    job.end()
    for k in ("values", "indices", "leaves", "siblings", "dirs"):
        assert np.array_equal(got[k], exp[k]), k
    tg.append_bytes(b"OPENING_CLAIMS")
    for c in range(43):
        tg.append_field(int(got["values"][c]))
    assert tg.challenge() == to.challenge(P)


@pytest.mark.parametrize("nv", [3, 11, 15])
def test_commit_job_survives_interleaved_calls(ctx, nv):
    """A host that uses the context between zigz_commit_begin and zigz_commit_roots / open_all (prover.zig:405-424:
    the reference evaluates and hashes freely while it owns the trees) still gets the oracle's roots and openings:
    every staging path of the context (pinned words, workspaces) is exercised in between."""
    import zigz_amd
    N = 1 << nv
    cols = rnd(0xC0117 + nv, 43 * N).reshape(43, N)
    exp = O.generate_commitments(P, O.Transcript(), cols, fast=(nv > 10))
    other = rnd(77 + nv, 1 << 15)
    pt = rnd(78 + nv, 15)
    job = zigz_amd.CommitJob(ctx, cols=cols)
    # all of these stage data through the context while the job's roots are in flight
    assert ctx.mle_eval(other, pt) == O.mle_eval(P, other, pt)
    assert ctx.mle_round_poly(other) == O.mle_round_poly(P, other)
    r, p_, fe = ctx.sumcheck_prove(other[: 1 << 12])
    r0, p0, fe0 = O.sumcheck_prove(P, other[: 1 << 12])
    assert np.array_equal(r, r0) and fe == fe0
    t = zigz_amd.SimpleMerkleTree(ctx, other[:1000])
    lv, _ = O.merkle_levels(other[:1000])
    assert t.root_hash == lv[(2 * 1024 - 2) * 32:(2 * 1024 - 1) * 32].tobytes()
    t.deinit()
    with pytest.raises(zigz_amd.errors.ZigzError) as err:  # a second job on the same context is refused, not corrupted
        zigz_amd.CommitJob(ctx, cols=cols)
    assert err.value.code == zigz_amd.errors.BAD_STATE
    roots = job.roots()
    assert np.array_equal(roots, exp["roots"])
    assert ctx.mle_eval(other, pt) == O.mle_eval(P, other, pt)  # ... and between roots and open_all
    got = job.open_all(exp["points"])
    job.end()
    for k in ("values", "indices", "leaves", "siblings", "dirs"):
        assert np.array_equal(got[k], exp[k]), k


@pytest.mark.parametrize("nv", [10, 12, 15])
def test_small_domain_tables_identical_trees(ctx, nv):
    """Option "small_domain_mask": hinted columns take leaf + level-1 digests from the constant tables.  Roots and every
    opening equal the oracle's (literal hashing) for columns that satisfy the bound (values < 128), columns that violate
    it everywhere (hint wrong: the device notices and hashes) and columns that violate it in a few places only."""
    import zigz_amd
    N = 1 << nv
    rng = np.random.default_rng(nv)
    cols = rnd(0x5D00 + nv, 43 * N).reshape(43, N)
    small = [1, 33, 34, 35, 36, 37, 38, 42]
    for c in small:
        cols[c] = rng.integers(0, 128, size=N, dtype=np.uint64)
    cols[1] = 0
    cols[42] = rng.integers(0, 2, size=N, dtype=np.uint64)
    cols[36, 777] = 128               # one value just outside the domain
    cols[37, N - 1] = P - 1           # ... and one far outside, in the last pair
    hinted = small + [5, 20]          # two hinted columns hold arbitrary field elements: the hint is simply wrong there
    mask = sum(1 << c for c in hinted)
    exp = O.generate_commitments(P, O.Transcript(), cols, fast=(nv > 10))
    ctx.set_option("small_domain_mask", mask)
    try:
        job = zigz_amd.CommitJob(ctx, cols=cols)
        roots = job.roots()
        st = ctx.stats()
        got = job.open_all(exp["points"])
        job.end()
    finally:
        ctx.set_option("small_domain_mask", 0)
    assert st["small_domain_columns"] == len(hinted)
    # columns 5 and 20 fall back on every wave, 36 and 37 on one wave each
    assert st["small_domain_fallback_waves"] == 2 * (N // 2 // 64) + 2
    assert np.array_equal(roots, exp["roots"])
    for k in ("values", "indices", "leaves", "siblings", "dirs"):
        assert np.array_equal(got[k], exp[k]), k
    # openings that pass through looked-up nodes: leaf index pairs inside hinted columns
    for c in (1, 36, 42, 5):
        lv, _ = O.merkle_levels(cols[c])
        assert lv[(2 * N - 2) * 32:(2 * N - 1) * 32].tobytes() == roots[c].tobytes()


def test_commit_job_large_properties(ctx):
    """43 x 2^18 columns (BASELINE-scale shape): every opening verifies against its root through the
    oracle's Merkle verifier; values match the oracle's fold-eval on sampled columns."""
    import zigz_amd
    nv = 18
    N = 1 << nv
    cols = rnd(991, 43 * N).reshape(43, N)
    job = zigz_amd.CommitJob(ctx, cols=cols)
    roots = job.roots()
    points = rnd(992, 43 * nv).reshape(43, nv)
    got = job.open_all(points)
    job.end()
    for c in range(43):
        idx = int(points[c, 0]) % N
        assert int(got["indices"][c]) == idx and int(got["leaves"][c]) == int(cols[c, idx])
        assert O.merkle_verify(roots[c].tobytes(), int(got["leaves"][c]), got["siblings"][c].tobytes(),
                               got["dirs"][c].tobytes())
    for c in (0, 17, 42):
        lv, h = O.merkle_levels(cols[c])
        assert lv[(2 * N - 2) * 32:(2 * N - 1) * 32].tobytes() == roots[c].tobytes()
        # eval via oracle folds == sumcheck-interactive final eval with reversed point
        _, _, fe = O.sumcheck_prove(P, cols[c], list(points[c])[::-1])
        assert fe == int(got["values"][c])


# ---------------------------------------------------------------- A11: Lasso
@pytest.mark.parametrize("i", [i for i, e in enumerate(G["lasso"]) if e["p"] == str(P)])
def test_lasso_golden(ctx, i):
    e = G["lasso"][i]
    tab = O.build_table(P, e["kind"], e["bits"])
    q = np.array([ints(r) for r in e["queries"]], dtype=np.uint64)
    d = ctx.lasso_prove(tab, q)
    assert d["nv"] == e["nv"] and [str(x) for x in d["rounds"]] == e["rounds"]
    assert [str(x) for x in d["point"]] == e["point"] and str(d["final_eval"]) == e["final_eval"]
    assert d["query_commit"].hex() == e["query_commit"] and d["table_commit"].hex() == e["table_commit"]


def test_lasso_vs_oracle_and_errors(ctx):
    import zigz_amd
    E = zigz_amd.errors
    tab = O.build_table(P, 1, 8)  # 8-bit XOR subtable: 2^16 rows (table_decomposition scale)
    rng = np.random.default_rng(3)
    a = rng.integers(0, 256, 1000); b = rng.integers(0, 256, 1000)
    q = np.stack([a, b, a ^ b], axis=1).astype(np.uint64)
    mapping = (a * 256 + b).astype(np.uint64)
    d = ctx.lasso_prove(tab, q, mapping=mapping)
    o = O.lasso_prove(P, tab, q, mapping=mapping)
    for k in ("nv", "final_eval", "query_commit", "table_commit"):
        assert d[k] == o[k], k
    assert np.array_equal(d["rounds"], o["rounds"]) and np.array_equal(d["point"], o["point"])
    fp = ctx.lasso_fingerprints(q)
    assert [int(x) for x in fp[:50]] == [O.lasso_hash_row(P, r) for r in q[:50]]
    for fn, code in [(lambda: ctx.lasso_prove(tab, np.zeros((0, 3))), E.NO_QUERIES),
                     (lambda: ctx.lasso_prove(tab, q[:1]), E.NO_VARIABLES),
                     (lambda: ctx.lasso_prove(tab[:3], q[:2]), E.LENGTH_NOT_POWER_OF_TWO),
                     (lambda: ctx.lasso_prove(tab, q[:2], mapping=[0]), E.MAPPING_LENGTH_MISMATCH),
                     (lambda: ctx.lasso_prove(tab, q[:2], mapping=[1 << 20, 0]), E.INVALID_MAPPING),
                     (lambda: ctx.lasso_prove(tab, q[:2], mapping=[0, 0]), E.QUERY_TABLE_MISMATCH)]:
        with pytest.raises(zigz_amd.ZigzError) as e:
            fn()
        assert e.value.code == code, e.value


# ---------------------------------------------------------------- device-resident entry points
def test_device_resident_api(ctx):
    nv = 16
    n = 1 << nv
    ev = rnd(1234, n)
    d_in = ctx.dev_alloc(n * 4)
    d_out = ctx.dev_alloc(n * 2)
    try:
        ctx.upload(ev, d_in)
        assert np.array_equal(ctx.download(d_in, n), ev)
        s = ctx.dev_mle_half_sums(d_in, n)
        rp = O.mle_round_poly(P, ev)
        assert s[0] == rp[0] and (s[1] - s[0]) % P == rp[1]
        r = int(rnd(5, 1)[0])
        ctx.dev_mle_bind(d_in, n, r, d_out)
        exp = O.mle_partial_eval(P, ev, r)
        assert np.array_equal(ctx.download(d_out, n // 2), exp)
        s2 = ctx.dev_mle_bind_sums(d_in, n, r, d_out)
        rp2 = O.mle_round_poly(P, exp)
        assert s2[0] == rp2[0] and (s2[1] - s2[0]) % P == rp2[1]
        assert np.array_equal(ctx.download(d_out, n // 2), exp)
        pt = rnd(6, nv)
        assert ctx.dev_mle_eval(d_in, n, pt) == O.mle_eval(P, ev, pt)
        rr, pp, fe = ctx.dev_sumcheck_prove(d_in, n)
        r0, p0, f0 = O.sumcheck_prove(P, ev)
        assert np.array_equal(rr, r0) and np.array_equal(pp, p0) and fe == f0
        assert np.array_equal(ctx.download(d_in, n), ev)  # input table untouched
        # K8: F.init over raw 64-bit words
        raw = np.array([0, 1, P - 1, P, P + 1, 2**32, 2**63, 2**64 - 1, 0xFFFFFFFFFFFFF000], dtype=np.uint64)
        ctx.reduce_upload(raw, d_out)
        assert [int(x) for x in ctx.download(d_out, len(raw))] == [int(x) % P for x in raw]
    finally:
        ctx.dev_free(d_in)
        ctx.dev_free(d_out)


def test_host_transcript_vs_oracle(ctx):
    import zigz_amd
    a, b = zigz_amd.Transcript(), O.Transcript()
    for t in (a, b):
        t.append_bytes(b"SUMCHECK_BEGIN"); t.append_field(12345)
    a.append_tagged_counter(b"LASSO_TABLE", 0, 1000)
    for i in range(1000):
        b.append_bytes(b"LASSO_TABLE"); b.append_field(i)
    for _ in range(5):
        assert a.challenge() == b.challenge(P)


# ---------------------------------------------------------------- A0 / K8: witness columns built on the device
@pytest.mark.parametrize("ns", [1, 2, 3, 63, 64, 65, 100, 1000, 4097])
def test_witness_from_rows(ctx, ns):
    """Packed trace rows [ns][43] (raw u64, full 64-bit range) -> 43 padded columns, vs the reference rule:
    cell = x mod p; padding repeats the last row for pc/registers (columns 0..32) and is 0 elsewhere."""
    rng = np.random.default_rng(ns)
    rows = rng.integers(0, 2**64, size=(ns, 43), dtype=np.uint64)
    rows[ns // 2, :5] = [0, P - 1, P, P + 1, 2**64 - 1]
    nv = 0 if ns == 1 else int(ns - 1).bit_length()
    N = 1 << nv
    stride = max(N, 4)
    d = ctx.dev_alloc(43 * stride * 4)
    try:
        ctx.witness_from_rows(rows, nv, d, stride)
        got = ctx.download(d, 43 * stride).reshape(43, stride)[:, :N]
    finally:
        ctx.dev_free(d)
    exp = np.zeros((43, N), dtype=np.uint64)
    exp[:, :ns] = (rows % np.uint64(P)).T
    exp[:33, ns:] = exp[:33, ns - 1:ns]
    assert np.array_equal(got, exp)


def _expand_reference(steps, ns, N, init):
    """numpy restatement of witness.zig:65-270 on compact records: regs_after by carrying writes forward, x mod p,
    padding rule.  Returns [43, N] uint64."""
    p = np.uint64(P)
    exp = np.zeros((43, N), dtype=np.uint64)
    exp[0, :ns] = steps["pc"] % p
    for r in range(1, 32):
        idx = np.where(steps["wr_reg"] == r, np.arange(ns), -1)
        last = np.maximum.accumulate(idx)
        vals = np.where(last >= 0, steps["rd_value"][np.maximum(last, 0)], np.uint64(init[r]))
        exp[1 + r, :ns] = vals % p
    for c, f in ((33, "opcode"), (34, "rd"), (35, "rs1"), (36, "rs2"), (37, "funct3"), (38, "funct7")):
        exp[c, :ns] = steps[f]
    exp[39, :ns] = steps["imm"].astype(np.int64).view(np.uint64) % p
    exp[40, :ns] = steps["mem_addr"] % p
    exp[41, :ns] = steps["mem_value"] % p
    exp[42, :ns] = steps["mem_is_read"]
    exp[:33, ns:] = exp[:33, ns - 1:ns]
    return exp


@pytest.mark.parametrize("ns", [1, 2, 3, 63, 64, 65, 127, 129, 1000, 4097, 70000])
def test_witness_from_steps_synthetic(ctx, ns):
    """Compact records with full-range 64-bit values and sparse / dense register writes (registers written once, never,
    or on every step; writes at chunk boundaries) -> the 43 padded columns, vs the reference rule restated in numpy."""
    import zigz_amd
    rng = np.random.default_rng(1000 + ns)
    st = np.zeros(ns, dtype=zigz_amd.hip.TRACE_STEP_DTYPE)
    for f in ("pc", "rd_value", "mem_addr", "mem_value"):
        st[f] = rng.integers(0, 2**64, size=ns, dtype=np.uint64)
    st["imm"] = rng.integers(-2**63, 2**63, size=ns, dtype=np.int64)
    for f, hi in (("opcode", 128), ("rd", 32), ("rs1", 32), ("rs2", 32), ("funct3", 8), ("funct7", 128), ("mem_is_read", 2)):
        st[f] = rng.integers(0, hi, size=ns)
    wr = rng.integers(0, 8, size=ns)          # x1..x7 dense, 0 = no write
    sparse = rng.random(ns) < 0.01
    wr[sparse] = rng.integers(8, 30, size=int(sparse.sum()))  # x8..x29 rarely; x30, x31 never
    wr[0] = 9
    if ns > 64:
        wr[63], wr[64] = 10, 10             # last lane of a chunk / first lane of the next
    st["wr_reg"] = wr
    st["rd_value"][ns // 2] = P             # p, p-1 and 2^64-1 as written values
    st["rd_value"][ns // 3] = 2**64 - 1
    init = rng.integers(0, 2**64, size=32, dtype=np.uint64)
    nv = 0 if ns == 1 else int(ns - 1).bit_length()
    N = 1 << nv
    stride = max(N, 4) + (4 if ns % 2 else 0)  # also a stride larger than the column
    d = ctx.dev_alloc(43 * stride * 4)
    try:
        ctx.witness_from_steps(st, nv, d, stride, init)
        got = ctx.download(d, 43 * stride).reshape(43, stride)[:, :N]
    finally:
        ctx.dev_free(d)
    init[0] = 0
    exp = _expand_reference(st, ns, N, init)
    bad = np.argwhere(got != exp)
    assert bad.size == 0, (bad[:5], got[tuple(bad[0])], exp[tuple(bad[0])])


@pytest.mark.parametrize("maker,arg,regs", [("fibonacci", 3, None), ("fibonacci", 60, None), ("mixed_loop", 40, None),
                                            ("mixed_loop", 700, [0, 5, 1 << 40, 2**64 - 1] + [7] * 28),
                                            ("add_xor_loop", 2000, None)])
def test_witness_from_steps_vs_oracle(ctx, maker, arg, regs):
    """Real traces: host VM (compact records) -> device expansion == the oracle's VM + WitnessGenerator (witness.zig),
    and == the packed-rows path."""
    import programs
    from zigz_amd import host
    made = getattr(programs, maker)(arg)
    prog, inp = made if isinstance(made, tuple) else (made, None)
    tr = host.Trace(prog, 0x1000, regs, 1 << 20, inp)
    cols, nv, ns = O.witness_from_program(P, prog, 0x1000, regs, 1 << 20, inp)
    assert (tr.num_steps, tr.num_vars) == (ns, nv)
    N = 1 << nv
    stride = max(N, 4)
    d = ctx.dev_alloc(43 * stride * 4)
    try:
        tr.witness_to_device(ctx, d, stride)  # zigz_dev_witness_from_steps
        got = ctx.download(d, 43 * stride).reshape(43, stride)[:, :N]
        assert np.array_equal(got, cols)
        ctx.witness_from_rows(tr.rows(), nv, d, stride)  # the packed-rows entry on the host-expanded rows
        got = ctx.download(d, 43 * stride).reshape(43, stride)[:, :N]
        assert np.array_equal(got, cols)
    finally:
        ctx.dev_free(d)
    # the records themselves against the oracle's trace
    st, init = tr.steps()
    ot = O.vm_trace(prog, 0x1000, regs, 1 << 20, inp)
    assert np.array_equal(st["pc"], ot["pc"]) and np.array_equal(st["imm"], ot["imm"])
    for f in ("opcode", "rd", "rs1", "rs2", "funct3", "funct7"):
        assert np.array_equal(st[f], ot[f]), f
    assert np.array_equal(st["mem_is_read"], (ot["mem_kind"] == 1).astype(np.uint8))


# ---------------------------------------------------------------- A5: both sumcheck forms agree
@pytest.mark.parametrize("nv", [10, 11, 12, 13, 14, 15, 16, 18, 19, 21, 24])
def test_sumcheck_radix_equals_per_round(ctx, nv):
    """Tables >= 2^11 take the radix-2^k form (k rounds per pass from block sums); it must reproduce the
    one-launch-per-round form bit for bit, with Fiat-Shamir and with fixed challenges, and the oracle where
    the oracle is affordable."""
    n = 1 << nv
    ev = rnd(900 + nv, n)
    d = ctx.dev_alloc(n * 4)
    try:
        ctx.upload(ev, d)
        chs = rnd(950 + nv, nv)
        ctx.set_option("per_round_sumcheck", 0)
        a = ctx.dev_sumcheck_prove(d, n)
        a2 = ctx.dev_sumcheck_prove(d, n, chs)
        ctx.set_option("per_round_sumcheck", 1)
        b = ctx.dev_sumcheck_prove(d, n)
        b2 = ctx.dev_sumcheck_prove(d, n, chs)
        ctx.set_option("per_round_sumcheck", 0)
    finally:
        ctx.dev_free(d)
    for x, y in ((a, b), (a2, b2)):
        assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and x[2] == y[2]
    if nv <= 19:
        o = O.sumcheck_prove(P, ev)
        assert np.array_equal(a[0], o[0]) and np.array_equal(a[1], o[1]) and a[2] == o[2]
        o2 = O.sumcheck_prove(P, ev, chs)
        assert np.array_equal(a2[0], o2[0]) and a2[2] == o2[2]


@pytest.mark.parametrize("nv", [14, 15, 17, 20])
def test_eval_radix_equals_folds(ctx, nv):
    """Evals of tables >= 2^14 run as one radix pass + a 1024-term weighted dot; must equal the fold form and
    (where affordable) the oracle's naive eval, batched over ragged column counts."""
    N = 1 << nv
    d = ctx.dev_alloc(N * 4)
    try:
        ev = rnd(1200 + nv, N)
        ctx.upload(ev, d)
        for s in range(2):
            pt = rnd(1300 + nv + s, nv)
            ctx.set_option("fold_eval", 0)
            a = ctx.dev_mle_eval(d, N, pt)
            ctx.set_option("fold_eval", 1)
            b = ctx.dev_mle_eval(d, N, pt)
            ctx.set_option("fold_eval", 0)
            assert a == b
            if nv <= 17:
                assert a == O.mle_eval(P, ev, pt)
        for pt in ([0] * nv, [1] * nv, [P - 1] * nv, [1] + [0] * (nv - 1)):
            ctx.set_option("fold_eval", 1)
            b = ctx.dev_mle_eval(d, N, pt)
            ctx.set_option("fold_eval", 0)
            assert ctx.dev_mle_eval(d, N, pt) == b
        assert ctx.dev_mle_eval(d, N, [1] + [0] * (nv - 1)) == int(ev[1])  # point[0] is the LSB
    finally:
        ctx.dev_free(d)


def test_unaligned_device_pointers(ctx):
    """Device-resident entry points must accept tables whose base is only 4-byte aligned (e.g. a slice of a larger
    buffer): the 16-byte vector kernels then give way to the scalar ones -- same results, no fault."""
    nv = 15
    n = 1 << nv
    ev = rnd(4242, n)
    d = ctx.dev_alloc((n + 8) * 4)
    o = ctx.dev_alloc((n + 8) * 4)
    try:
        for off in (1, 2, 3):
            ctx.upload(ev, d + 4 * off)
            r = int(rnd(off, 1)[0])
            ctx.dev_mle_bind(d + 4 * off, n, r, o + 4 * off)
            assert np.array_equal(ctx.download(o + 4 * off, n // 2), O.mle_partial_eval(P, ev, r))
            s = ctx.dev_mle_half_sums(d + 4 * off, n)
            rp = O.mle_round_poly(P, ev)
            assert s[0] == rp[0] and (s[1] - s[0]) % P == rp[1]
            pt = rnd(10 + off, nv)
            assert ctx.dev_mle_eval(d + 4 * off, n, pt) == O.mle_eval(P, ev, pt)
            a = ctx.dev_sumcheck_prove(d + 4 * off, n)
            b = O.sumcheck_prove(P, ev)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    finally:
        ctx.dev_free(d)
        ctx.dev_free(o)


def _run_tile_nodes(N, l):
    """kernels.hpp run_tile_nodes: the nodes one segment of a stage covers at level l (its first node is always hashed)."""
    if l == 0:
        return 4096
    s = (l - 1) // 6
    n_in = N >> (6 * s)
    return min(4096, n_in) >> (l - 6 * s)


def _list_levels(nv):
    """the list-driven levels of a 2^nv tree: 0 .. nv - 8 (down to 256 nodes per column)"""
    return nv - 8 + 1


def _run_aware_hashed(cols, levels):
    """numpy model of k_runs_stage: nodes hashed (not copied from the left neighbour) on levels 0..levels-1; the first node
    of every tile (what one segment of a stage covers at that level) is always hashed."""
    total = 0
    for col in np.asarray(cols):
        uni = np.ones(col.size, dtype=bool)
        for l in range(levels):
            if l:
                half = col[(1 << (l - 1))::(1 << l)]
                uni = uni[0::2] & uni[1::2] & (col[::(1 << l)] == half)
            val = col[::(1 << l)]
            copy = np.zeros(val.size, dtype=bool)
            copy[1:] = uni[1:] & uni[:-1] & (val[1:] == val[:-1])
            copy[::_run_tile_nodes(col.size, l)] = False
            total += int((~copy).sum())
    return total


def _tree_words(ctx, job, ncols):
    d, per_col = job.tree()
    return ctx.download(d, ncols * per_col // 4).astype(np.uint32)


@pytest.mark.parametrize("nv", [15, 16, 18])
def test_run_aware_levels_identical_trees(ctx, nv):
    """Option "run_aware_mask": on the levels 0 .. v - 8 a node that is a copy of its left neighbour (both subtrees
    uniform, same value) is copied instead of hashed.  EVERY node of EVERY tree must equal the dense build's (the device
    trees are compared word for word), whatever the columns look like and whatever the hint says; roots also vs the oracle."""
    import zigz_amd
    N = 1 << nv
    nc = 16
    cols = rnd(5150 + nv, nc * N).reshape(nc, N).copy()
    cols[1, :] = 5                                               # constant
    cols[2, :] = 0                                               # all zero (an unused register)
    cols[3, :] = np.repeat(rnd(1, N // 1000 + 1), 1000)[:N]      # runs of 1000: boundaries anywhere
    cols[4, :] = 9; cols[4, 70000 % N] = 10                      # uniform except one leaf
    cols[5, : N // 2] = 123                                      # constant first half, random second half
    cols[6, :] = np.repeat(rnd(2, N // 256), 256)                # aligned runs of 256, all different
    cols[7, :] = np.arange(N) % 4                                # period 4: never uniform above the leaves
    cols[8, -300:] = 0                                           # random, then a zero tail
    cols[9, :] = np.repeat(rnd(3, N // 4096), 4096)              # runs that are exactly the kernel's tiles
    cols[10, :] = 1
    for pos in (63, 64, 65, 4095, 4096, 4097, N - 1, N // 2, N // 2 + 1):   # changes next to chunk / tile boundaries
        cols[10, pos:] += 1
    cols[11, :] = np.repeat(rnd(4, N // 2), 2)                   # runs of 2
    cols[12, :] = np.repeat(rnd(5, N // 3 + 1), 3)[:N]           # runs of 3
    cols[14, :] = np.repeat(np.arange(N // 512) % 128, 512)      # small-domain AND piecewise constant
    cols[15, :] = 7                                              # constant but not hinted: dense
    run_mask = sum(1 << c for c in range(13)) | (1 << 14)
    trees = []
    for masks in ((0, 0), (run_mask, 0), (run_mask, 1 << 14), (~0 & ((1 << nc) - 1), 0)):
        ctx.set_option("run_aware_mask", masks[0])
        ctx.set_option("small_domain_mask", masks[1])
        ctx.set_option("run_aware_materialize", 1)   # write the copies too: whole trees are compared below
        try:
            job = zigz_amd.CommitJob(ctx, cols=cols)
            roots = job.roots()
            st = ctx.stats()
            trees.append((roots.copy(), _tree_words(ctx, job, nc), st))
            job.end()
        finally:
            ctx.set_option("run_aware_mask", 0)
            ctx.set_option("small_domain_mask", 0)
            ctx.set_option("run_aware_materialize", 0)
    r0, t0, s0 = trees[0]
    assert s0["run_aware_columns"] == 0 and s0["keccak_permutations"] == nc * (2 * N - 1)
    for k, (r, t, st) in enumerate(trees[1:], 1):
        assert np.array_equal(r, r0), k
        bad = np.nonzero(t != t0)[0]
        assert bad.size == 0, (k, "first differing node", int(bad[0]) // 8, "of", t.size // 8)
    s1 = trees[1][2]
    levels = _list_levels(nv)
    assert s1["run_aware_columns"] == 14
    assert s1["run_aware_dense_nodes"] == 14 * sum(N >> l for l in range(levels))
    assert s1["run_aware_hashed"] == _run_aware_hashed(cols[[c for c in range(nc) if (run_mask >> c) & 1]], levels)
    assert s1["run_aware_hashed"] < s1["run_aware_dense_nodes"] * 0.75
    assert s1["keccak_permutations"] == nc * (2 * N - 1) - (s1["run_aware_dense_nodes"] - s1["run_aware_hashed"])
    assert trees[2][2]["run_aware_columns"] == 13 and trees[2][2]["small_domain_columns"] == 1
    for c in (1, 3, 4, 10, 12):
        lv, h = O.merkle_levels(cols[c])
        assert lv[(2 * N - 2) * 32:(2 * N - 1) * 32].tobytes() == r0[c].tobytes()


@pytest.mark.parametrize("nv", [15, 17, 21])  # 21: eight run-aware levels
def test_run_aware_virtual_copies_open_like_the_dense_tree(ctx, nv):
    """In a commit job the copies of the run-aware levels are not written at all: the next level's hashes and the openings
    read a copy through its leader (bitmap of hashed nodes + last hashed node before the chunk).  Roots and the openings at
    many indices -- next to change points, chunk and tile boundaries, the ends, random -- must equal the dense build's."""
    import zigz_amd
    N = 1 << nv
    nc = 12
    cols = rnd(7150 + nv, nc * N).reshape(nc, N).copy()
    cols[1, :] = 5
    cols[2, :] = 0
    cols[3, :] = np.repeat(rnd(1, N // 1000 + 1), 1000)[:N]
    cols[4, :] = 9; cols[4, 20000] = 10
    cols[5, : N // 2] = 123
    cols[6, :] = np.repeat(rnd(2, N // 256), 256)
    cols[7, :] = np.repeat(rnd(3, N // 4096), 4096)
    cols[8, :] = 1
    for pos in (63, 64, 65, 4095, 4096, 4097, N - 1, N // 2, N // 2 + 1):
        cols[8, pos:] += 1
    cols[9, :] = np.repeat(rnd(4, N // 3 + 1), 3)[:N]
    cols[10, :] = np.repeat(np.arange(N // 512) % 128, 512)
    rng = np.random.default_rng(nv)
    special = [0, 1, 62, 63, 64, 65, 127, 128, 999, 1000, 1001, 4094, 4095, 4096, 4097, 8191, 8192, 19999, 20000, 20001,
               N // 2 - 1, N // 2, N // 2 + 1, N - 2, N - 1]
    index_sets = [np.full(nc, i) for i in special] + [rng.integers(0, N, nc) for _ in range(25)]
    if nv > 17:
        index_sets = index_sets[::3]

    def run(run_mask, sd_mask):
        ctx.set_option("run_aware_mask", run_mask)
        ctx.set_option("small_domain_mask", sd_mask)
        try:
            outs = []
            for idx in index_sets:  # a job opens once: one build per index set
                job = zigz_amd.CommitJob(ctx, cols=cols)
                try:
                    roots = job.roots().copy()
                    st = ctx.stats()
                    pts = rnd(int(idx[0]) + 5, nc * nv).reshape(nc, nv)
                    pts[:, 0] = idx
                    o = job.open_all(pts)
                    outs.append({k: v.copy() for k, v in o.items()})
                finally:
                    job.end()
        finally:
            ctx.set_option("run_aware_mask", 0)
            ctx.set_option("small_domain_mask", 0)
        return roots, outs, st

    r0, o0, s0 = run(0, 0)
    for masks in ((0xBFF, 0), (0xFFF, 0), (0xBFF, 1 << 10)):
        r1, o1, s1 = run(*masks)
        assert np.array_equal(r0, r1), masks
        assert s1["run_aware_columns"] >= 10 and s1["run_aware_hashed"] < s1["run_aware_dense_nodes"]
        for a, b in zip(o0, o1):
            for k in a:
                assert np.array_equal(a[k], b[k]), (masks, k, a["indices"][:3])
    lv, h = O.merkle_levels(cols[8])
    assert lv[(2 * N - 2) * 32:(2 * N - 1) * 32].tobytes() == r0[8].tobytes()
    k = 4  # one of the constructed sets (every column opens the same index there)
    sib, dirs, leaf = O.merkle_open(cols[8], int(index_sets[k][8]))
    assert o0[k]["siblings"][8].tobytes() == sib and int(o0[k]["leaves"][8]) == leaf


@pytest.mark.parametrize("nv", [15, 17])
def test_content_addressed_group_identical_trees(ctx, nv):
    """Option "cons_group_mask": the levels 0 .. v - 8 of a column group are content-addressed (one representative
    per distinct node, found through a device hash table; only representatives hashed).  With the copies written out the
    whole device trees equal the dense build's, for groups that repeat (loops of power-of-two and odd periods), that do not
    repeat at all, whose columns repeat in DIFFERENT places (the tuple decides, not one column), and next to the other hints."""
    import zigz_amd
    N = 1 << nv
    nc = 12
    cols = rnd(9150 + nv, nc * N).reshape(nc, N).copy()
    step = np.arange(N)
    cols[0, :] = 0x1000 + 4 * (step % 4)                       # a 4-step loop: "pc"
    cols[1, :] = (step % 4) * 7 + 3                            # a function of it
    cols[2, :] = 0                                             # constant (x0)
    cols[3, :] = 0x2000 + 4 * (step % 31)                      # a 31-step loop (odd period: never aligned)
    cols[4, :] = (step % 31) % 128
    cols[5, :] = np.where(step < N // 2, step % 5, step % 7)   # two loops one after the other
    cols[6, :] = rnd(77, N)                                    # no repetition at all
    cols[7, :] = np.repeat(rnd(3, N // 64), 64)                # piecewise constant (run-aware hint)
    cols[8, :] = step % 128                                    # small-domain
    cols[9, :] = step % 3                                      # period 3: repeats where column 0 does not
    groups = [sum(1 << c for c in (0, 1, 2)), sum(1 << c for c in (3, 4)), sum(1 << c for c in (0, 1, 2, 5, 9)),
              sum(1 << c for c in (0, 6)), sum(1 << c for c in (0, 1, 2, 3, 4, 5, 8, 9))]
    trees = []
    for g in [0] + groups:
        ctx.set_option("cons_group_mask", g)
        ctx.set_option("run_aware_mask", 1 << 7)
        ctx.set_option("small_domain_mask", 1 << 8)
        ctx.set_option("run_aware_materialize", 1)
        try:
            job = zigz_amd.CommitJob(ctx, cols=cols)
            roots = job.roots().copy()
            st = ctx.stats()
            trees.append((roots, _tree_words(ctx, job, nc), st))
            job.end()
        finally:
            for o in ("cons_group_mask", "run_aware_mask", "small_domain_mask", "run_aware_materialize"):
                ctx.set_option(o, 0)
    r0, t0, s0 = trees[0]
    assert s0["cons_columns"] == 0
    levels = _list_levels(nv)
    for g, (r, t, st) in zip(groups, trees[1:]):
        assert np.array_equal(r, r0), g
        bad = np.nonzero(t != t0)[0]
        assert bad.size == 0, (g, "first differing node", int(bad[0]) // 8)
        ng = bin(g).count("1")
        if st["cons_probe_distinct"] > N // 4:   # the probe found (almost) no repetition: built like any other columns
            assert st["cons_columns"] == 0 and st["cons_dense_nodes"] == 0
        else:
            assert st["cons_columns"] == ng and st["cons_dense_nodes"] == ng * sum(N >> l for l in range(levels))
        assert st["keccak_permutations"] == nc * (2 * N - 1) - st["small_domain_columns"] * (N + N // 2) - \
            (st["run_aware_dense_nodes"] - st["run_aware_hashed"]) - (st["cons_dense_nodes"] - st["cons_hashed"])
    # a 4-step loop has 4 distinct leaves, 2 distinct pairs, then 1 node per level; a 31-step loop at most 31 per level
    assert trees[1][2]["cons_hashed"] == 3 * (4 + 2 + max(0, levels - 2))
    assert trees[2][2]["cons_hashed"] <= 2 * 31 * levels
    assert trees[1][2]["cons_probe_distinct"] == 4 and trees[2][2]["cons_probe_distinct"] == 31
    # a random column in the group: nothing repeats, the probe says so and the group is dropped
    assert trees[4][2]["cons_probe_distinct"] >= N - 16 and trees[4][2]["cons_columns"] == 0


@pytest.mark.parametrize("nv", [15, 18])
def test_content_addressed_group_virtual_openings(ctx, nv):
    """In a commit job only representatives hold a digest below the top content-addressed level; the next level's hashes and
    the openings go through the representative array.  Roots and openings at many indices equal the dense build's."""
    import zigz_amd
    N = 1 << nv
    nc = 10
    step = np.arange(N)
    cols = rnd(9350 + nv, nc * N).reshape(nc, N).copy()
    cols[0, :] = 0x1000 + 4 * (step % 12)
    cols[1, :] = (step % 12) % 5
    cols[2, :] = 0
    cols[3, :] = np.where(step % 12 == 3, 1, 0)
    cols[4, :] = np.repeat(rnd(5, N // 128), 128)
    cols[5, :] = (step % 12) * 1000003 % P
    cols[0, N - 1000:] = 0x1000 + 4 * 11                      # "padding": pc repeats its last value ...
    for c in (1, 3, 5):
        cols[c, N - 1000:] = 0                                 # ... while the instruction fields drop to 0
    group = sum(1 << c for c in (0, 1, 2, 3, 5))
    rng = np.random.default_rng(nv)
    special = [0, 1, 11, 12, 13, 63, 64, 4095, 4096, N - 1001, N - 1000, N - 999, N // 2, N - 1]
    index_sets = [np.full(nc, i) for i in special] + [rng.integers(0, N, nc) for _ in range(12)]

    def run(g):
        outs = []
        ctx.set_option("cons_group_mask", g)
        ctx.set_option("run_aware_mask", 1 << 4)
        try:
            for idx in index_sets:
                job = zigz_amd.CommitJob(ctx, cols=cols)
                try:
                    roots = job.roots().copy()
                    st = ctx.stats()
                    pts = rnd(int(idx[0]) + 9, nc * nv).reshape(nc, nv)
                    pts[:, 0] = idx
                    outs.append({k: v.copy() for k, v in job.open_all(pts).items()})
                finally:
                    job.end()
        finally:
            ctx.set_option("cons_group_mask", 0)
            ctx.set_option("run_aware_mask", 0)
        return roots, outs, st

    r0, o0, s0 = run(0)
    r1, o1, s1 = run(group)
    assert np.array_equal(r0, r1) and s1["cons_columns"] == 5 and s1["cons_hashed"] < s1["cons_dense_nodes"] // 100
    for a, b in zip(o0, o1):
        for k in a:
            assert np.array_equal(a[k], b[k]), (k, a["indices"][:3])
    lv, h = O.merkle_levels(cols[0])
    assert lv[(2 * N - 2) * 32:(2 * N - 1) * 32].tobytes() == r1[0].tobytes()


def test_run_aware_hash_counts(ctx):
    """What the run-aware levels cost: a constant column needs one hash per tile (the range one segment covers at a level);
    a change point at most two more per level."""
    import zigz_amd
    nv = 16
    N = 1 << nv
    levels = _list_levels(nv)                                    # 9 list-driven levels: 65536 .. 256 nodes
    tiles = sum((N >> l) // _run_tile_nodes(N, l) for l in range(levels))
    cols = np.zeros((4, N), dtype=np.uint64)
    cols[1, :] = 77
    cols[2, 12345:] = 3                                          # one change point
    cols[3, :] = np.repeat(rnd(9, 16), N // 16)                  # 15 change points, all on tile boundaries
    ctx.set_option("run_aware_mask", 0xF)
    try:
        job = zigz_amd.CommitJob(ctx, cols=cols)
        roots = job.roots()
        st = ctx.stats()
        job.end()
    finally:
        ctx.set_option("run_aware_mask", 0)
    assert st["run_aware_columns"] == 4
    assert st["run_aware_hashed"] == _run_aware_hashed(cols, levels)
    # one hash per tile for the constant columns; column 2's change point costs 1 leaf + at most 2 nodes on each higher level;
    # column 3's 15 change points sit on tile starts at the leaves and cost at most one node each above
    assert 4 * tiles < st["run_aware_hashed"] <= 4 * tiles + (1 + 2 * (levels - 1)) + 15 * levels
    assert _run_aware_hashed(cols[:2], levels) == 2 * tiles
    for c in range(4):
        lv, h = O.merkle_levels(cols[c])
        assert lv[(2 * N - 2) * 32:(2 * N - 1) * 32].tobytes() == roots[c].tobytes()


@pytest.mark.parametrize("n", [32769, 40000, 70001])
def test_run_aware_single_tree_with_padding(ctx, n):
    """SimpleMerkleTree.build with the hint set: fewer values than leaves (padding = hashLeaf(0)); root and openings vs the
    oracle, including the last value, the first padding leaf's sibling and the far end."""
    import zigz_amd
    ev = np.repeat(rnd(n, n // 37 + 1), 37)[:n].astype(np.uint64)
    ctx.set_option("run_aware_mask", 1)
    try:
        tree = zigz_amd.SimpleMerkleTree(ctx, ev)
    finally:
        ctx.set_option("run_aware_mask", 0)
    lv, h = O.merkle_levels(ev)
    npad = 1 << (n - 1).bit_length()
    assert tree.root_hash == lv[(2 * npad - 2) * 32:(2 * npad - 1) * 32].tobytes()
    for i in (0, 1, 36, 37, 4095, 4096, n - 2, n - 1):
        o = tree.open(i)
        assert o["value"] == int(ev[i])
        assert O.merkle_verify(tree.root_hash, o["value"], o["siblings"], o["directions"])
        assert (o["siblings"], o["directions"], o["value"]) == O.merkle_open(ev, i)
    tree.deinit()


# ---------------------------------------------------------------- reference-held KATs, second batch, on the HIP path
# tests/golden/ref_kats2.json (extract_ref_kats2.py).  See tests/test_oracle_golden.py for the three reference tests
# whose stated expectations the reference's own code does not meet; the HIP path follows the code, like the oracle.
K2 = json.load(open(os.path.join(HERE, "golden", "ref_kats2.json")))


@pytest.mark.parametrize("i", range(len(K2["witness"])))
def test_ref_witness_kats_gpu(ctx, i):
    """witness.zig:384-464 through the device witness build (compact trace -> 43 columns in HBM) + device eval."""
    from zigz_amd import host
    k = K2["witness"][i]
    prog = bytes(k["program"])
    tr = host.Trace(prog, k["entry_pc"], None, k["steps"])
    nv, ns = tr.num_vars, tr.num_steps
    N, stride = 1 << nv, max(1 << nv, 4)
    d = ctx.dev_alloc(43 * stride * 4)
    try:
        tr.witness_to_device(ctx, d, stride)
        cols = ctx.download(d, 43 * stride).reshape(43, stride)[:, :N]
        if "num_steps" in k:
            assert ns == k["num_steps"]
        if "witness_size" in k:
            assert cols.size == k["witness_size"]
        if "num_vars" in k and k["num_steps"] > 1:
            assert nv == k["num_vars"]
        colmap = {"pc": 0, "mem.is_read": 42, "x10": 11}
        for e in k["evals"]:
            c = colmap[e["column"]]
            if e["column"] == "mem.is_read":
                step = {0: 1, 1: 2}[e["value"]]  # store at step 1 -> 0, load at step 2 -> 1
                assert int(cols[c, step]) == e["value"]
                lsb = sum(int(b) << j for j, b in enumerate(e["point"]))
                assert ctx.dev_mle_eval(d + c * stride * 4, N, e["point"]) == int(cols[c, lsb])
            else:
                assert int(cols[c, 0]) == e["value"]
        ocols, onv, ons = O.witness_from_program(P, prog, k["entry_pc"], None, k["steps"])
        assert (onv, ons) == (nv, ns) and np.array_equal(cols, ocols)
    finally:
        ctx.dev_free(d)


@pytest.mark.parametrize("i", range(len(K2["tables"])))
def test_ref_table_builder_kats_gpu(ctx, i):
    """table_builder.zig:292-365: the tables are Lasso inputs; their row fingerprints (K9) on the device equal the
    oracle's for the reference's 2- and 3-bit ADD / XOR / AND tables, and the looked-up row is the expected one."""
    k = K2["tables"][i]
    kind = {"add": 0, "xor": 1, "and": 2}[k["kind"]]
    tab = O.build_table(P, kind, k["bits"])
    fp = ctx.lasso_fingerprints(tab)
    assert [int(x) for x in fp] == [O.lasso_hash_row(P, [int(v) for v in r]) for r in tab]
    if k["output"] is not None:
        row = k["inputs"] + [k["output"]]
        j = [list(map(int, r)) for r in tab].index(row)
        assert int(ctx.lasso_fingerprints(np.array([row], dtype=np.uint64))[0]) == int(fp[j])


@pytest.mark.parametrize("i", range(len(K2["commit"])))
def test_ref_polynomial_commit_kats_gpu(ctx, i):
    """polynomial_commit.zig:261-451 (F17 tables; the leaves hash the integers, so the trees are field-independent and
    the boolean-point evaluations equal table entries in any field): commit / open / verify through the C ABI."""
    import zigz_amd
    k = K2["commit"][i]
    polys = [[v % k["modulus"] for v in q["evals"]] for q in k["polys"]]
    trees = [zigz_amd.SimpleMerkleTree(ctx, ev) for ev in polys]
    try:
        for ev, t in zip(polys, trees):
            assert t.getRoot() == O.merkle_build(ev)[0] and len(t.getRoot()) == 32
        if k.get("deterministic"):
            t2 = zigz_amd.SimpleMerkleTree(ctx, polys[0])
            assert t2.getRoot() == trees[0].getRoot()
            t2.deinit()
        for o in k["opens"]:
            ev, t = polys[o["poly"]], trees[o["poly"]]
            g = zigz_amd.CommitmentScheme.open(ctx, ev, t, o["point"])
            val, idx, sib, dirs, leaf = O.commit_open(k["modulus"], ev, o["point"])
            assert (g["value"], g["index"], g["leaf"], g["siblings"], g["directions"]) == (val, idx, leaf, sib, dirs)
            assert O.merkle_verify(t.getRoot(), g["leaf"], g["siblings"], g["directions"])
    finally:
        for t in trees:
            t.deinit()


def _open_sets(nc, nv, N, seed):
    rng = np.random.default_rng(seed)
    special = [0, 1, 63, 64, 4095, 4096, N // 2, N - 2, N - 1]
    return [np.full(nc, i) for i in special] + [rng.integers(0, N, nc) for _ in range(6)]


def test_lists_that_outgrow_their_room_are_rebuilt(ctx):
    """The lists of the structure-aware levels (and the digests stored in list order) are sized from what the context's earlier
    builds needed, not for the worst case.  A job whose columns need more -- here: run-aware columns that change at every
    leaf, after the context has only seen constant ones -- notices on the device, and zigz_commit_roots repeats the build with
    more room: roots and openings still equal the dense build's, stats.rebuilds says what happened, and the NEXT job of the
    same kind fits at once."""
    import zigz_amd
    nv, nc = 16, 8
    N = 1 << nv
    quiet = np.zeros((nc, N), dtype=np.uint64) + 5
    busy = rnd(8800, nc * N).reshape(nc, N).copy()
    busy[0, :] = 3
    sets = _open_sets(nc, nv, N, 5)

    def run(cols, mask):
        ctx.set_option("run_aware_mask", mask)
        try:
            outs = []
            for idx in sets[:4]:
                job = zigz_amd.CommitJob(ctx, cols=cols)
                try:
                    roots = job.roots().copy()
                    st = ctx.stats()
                    pts = rnd(int(idx[0]) + 3, nc * nv).reshape(nc, nv)
                    pts[:, 0] = idx
                    outs.append({k: v.copy() for k, v in job.open_all(pts).items()})
                finally:
                    job.end()
        finally:
            ctx.set_option("run_aware_mask", 0)
        return roots, outs, st

    r_dense, o_dense, _ = run(busy, 0)
    r0 = ctx.stats()["rebuilds"]
    run(quiet, 0xFF)                                   # the context learns: constant columns need next to nothing
    r1 = ctx.stats()["rebuilds"]
    r_list, o_list, st = run(busy, 0xFF)               # ... and then meets columns where every node is hashed
    r2 = ctx.stats()["rebuilds"]
    assert r1 == r0 and r2 >= r1 + 1, (r0, r1, r2)
    assert r2 - r1 <= 2                                # one repeat per job at most, and only for the first job(s) of the kind
    assert np.array_equal(r_dense, r_list)
    assert st["run_aware_hashed"] > 7 * N              # seven random columns: (almost) every node of every list level
    for a, b in zip(o_dense, o_list):
        for k in a:
            assert np.array_equal(a[k], b[k]), k
    run(busy, 0xFF)
    assert ctx.stats()["rebuilds"] == r2               # the room learnt is kept


def test_dropped_group_gets_slabs_by_a_rebuild(ctx):
    """A content-addressed group that does not repeat is dropped ON THE DEVICE; in a commit job its columns then have nowhere to
    be built densely (no slabs were set aside for them), so the first such job on a context is built twice.  Openings through
    the dropped group's columns -- small-domain members with virtual leaves included -- equal the dense build's."""
    import zigz_amd
    nv, nc = 15, 6
    N = 1 << nv
    cols = rnd(9900, nc * N).reshape(nc, N).copy()
    cols[1, :] = np.arange(N) % 128                    # a small-domain member of the group
    cols[2, :] = 0
    sets = _open_sets(nc, nv, N, 9)

    def run(group, sd):
        ctx.set_option("cons_group_mask", group)
        ctx.set_option("small_domain_mask", sd)
        try:
            outs = []
            for idx in sets:
                job = zigz_amd.CommitJob(ctx, cols=cols)
                try:
                    roots = job.roots().copy()
                    st = ctx.stats()
                    pts = rnd(int(idx[0]) + 1, nc * nv).reshape(nc, nv)
                    pts[:, 0] = idx
                    outs.append({k: v.copy() for k, v in job.open_all(pts).items()})
                finally:
                    job.end()
        finally:
            ctx.set_option("cons_group_mask", 0)
            ctx.set_option("small_domain_mask", 0)
        return roots, outs, st

    r0, o0, _ = run(0, 0)
    before = ctx.stats()["rebuilds"]
    ctx.set_option("cons_always", 1)                   # (a context stops trying after two drops in a row: not here)
    try:
        r1, o1, st = run(0b111, 0b110)                 # columns 0 (random), 1, 2: nothing repeats -> dropped
    finally:
        ctx.set_option("cons_always", 0)
    assert st["cons_columns"] == 0 and st["cons_probe_distinct"] > N // 4 and st["small_domain_columns"] == 2
    assert ctx.stats()["rebuilds"] == before + 1       # once: the context now sets slabs aside for this group
    r2, o2, st2 = run(0b111, 0b110)                    # the default: after two drops in a row the group is not even tried
    assert st2["cons_probe_distinct"] == 0 and st2["cons_columns"] == 0 and st2["small_domain_columns"] == 2
    assert np.array_equal(r0, r2)
    for a, b in zip(o0, o2):
        for k in a:
            assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(r0, r1)
    for a, b in zip(o0, o1):
        for k in a:
            assert np.array_equal(a[k], b[k]), k


def test_commit_begin_does_not_wait_for_the_device(ctx):
    """zigz_commit_begin_dev enqueues and returns (prover.zig:405-416: the trees do not depend on the transcript, so the host
    goes on absorbing while they build): the keep / drop decision of the content-addressed group is taken on the device.  On a
    2^20 x 43 job begin must return in a fraction of the time the build takes."""
    import time
    import zigz_amd
    from zigz_amd import host
    import programs
    nv = 20
    N = 1 << nv
    tr = host.Trace(programs.add_xor_loop((N - 3) // 4), 0x1000, None, 2 * N)
    d = ctx.dev_alloc(43 * N * 4)
    try:
        tr.witness_to_device(ctx, d, N)
        ctx.set_option("small_domain_mask", (1 << 1) | (0x3f << 33) | (1 << 42))
        ctx.set_option("run_aware_mask", (0x7fffffff << 2) | (3 << 40))
        ctx.set_option("cons_group_mask", 1 | (1 << 1) | (0x7f << 33) | (1 << 42))
        begins, builds = [], []
        for it in range(6):
            ctx.synchronize()
            t0 = time.perf_counter()
            job = zigz_amd.CommitJob(ctx, d_cols=d, ncols=43, nv=nv, col_stride=N)
            t1 = time.perf_counter()
            job.roots()
            t2 = time.perf_counter()
            job.end()
            if it >= 2:  # (the first jobs allocate workspaces and learn the list sizes)
                begins.append(t1 - t0)
                builds.append(t2 - t0)
        st = ctx.stats()
        assert st["cons_columns"] == 10 and st["run_aware_columns"] == 33
        print("commit_begin %.3f ms, begin + build %.3f ms (2^20 x 43)" % (min(begins) * 1e3, min(builds) * 1e3))
        assert min(begins) < 0.6e-3, begins           # ~45 launches enqueued; a build takes ~0.9 ms of device time alone
        assert min(begins) < 0.5 * min(builds), (begins, builds)
    finally:
        for o in ("small_domain_mask", "run_aware_mask", "cons_group_mask"):
            ctx.set_option(o, 0)
        ctx.dev_free(d)


@pytest.mark.gpu
def test_commit_job_with_sleeping_waits(ctx):
    """zigz_device_set_blocking_sync(on): the two waits of a commit job poll the completion word the last kernel stores in pinned
    memory (DoneFlag, csrc/kernels.hpp) instead of asking the runtime.  Same roots, values, leaves, paths as with the runtime's
    waits, job after job on one context (the word carries a sequence number), with and without timing mode (which keeps the
    runtime's waits), for two shapes."""
    import numpy as np
    import zigz_amd
    rng = np.random.default_rng(77)
    lib = zigz_amd._ffi.lib

    def run(c, cols, pts):
        out = []
        for _ in range(3):
            job = zigz_amd.CommitJob(c, cols)
            roots = job.roots()
            opened = job.open_all(pts)
            job.end()
            out.append((bytes(roots), [np.asarray(opened[k]).tobytes() for k in sorted(opened)]))
        assert out[0] == out[1] == out[2]
        return out[0]

    for ncols, nv in ((43, 15), (5, 9)):
        cols = rng.integers(0, 2013265921, size=(ncols, 1 << nv), dtype=np.uint64)
        cols[3 % ncols] = 7  # (a constant column and a piecewise-constant one for the run-aware levels)
        cols[2 % ncols] = np.repeat(rng.integers(0, 2013265921, size=(1 << nv) // 64, dtype=np.uint64), 64)
        pts = rng.integers(0, 2013265921, size=(ncols, nv), dtype=np.uint64)
        ctx.set_option("run_aware_mask", 0b1100 if nv >= 15 else 0)
        try:
            want = run(ctx, cols, pts)
            lib.zigz_device_set_blocking_sync(0, 1)  # (the device flag itself may be refused on an active device: the waits switch regardless)
            try:
                c2 = zigz_amd.Context(0)
                try:
                    c2.set_option("run_aware_mask", 0b1100 if nv >= 15 else 0)
                    assert run(c2, cols, pts) == want
                    c2.enable_timing(True)
                    assert run(c2, cols, pts) == want
                    c2.enable_timing(False)
                    assert run(c2, cols, pts) == want
                finally:
                    c2.close()
            finally:
                lib.zigz_device_set_blocking_sync(0, 0)
        finally:
            ctx.set_option("run_aware_mask", 0)


@pytest.mark.gpu
@pytest.mark.parametrize("nv", [15, 17])
def test_eval_leaves_out_constant_columns_only(ctx, nv):
    """zigz_commit_open_all does not read a hinted column again that the run-aware structure pass of the same job found
    constant (EvalSkip, csrc/kernels.hpp: the extension of a constant is that constant).  Every value must still be the
    oracle's eval -- for constant columns, for columns that are constant except in ONE place (the last leaf, the first leaf,
    the first leaf of a later 4096-leaf segment, the leaf before it, one leaf in the middle of a chunk) and for columns the hint
    does not cover -- and the count of skipped columns must be exactly the constant ones under the hint."""
    import zigz_amd
    N = 1 << nv
    cols = rnd(0xE7A1 + nv, 43 * N).reshape(43, N).copy()
    const = {2: 0, 3: 5, 4: 2013265920, 20: 77}
    for c, v in const.items():
        cols[c, :] = v
    near = {5: N - 1, 6: 0, 7: 4096 * 3, 8: 4096 * 3 - 1, 9: 64 * 100 + 17, 10: 4096}
    for c, i in near.items():
        cols[c, :] = 9
        cols[c, i] = 10
    cols[11, :] = 4
    cols[11, N // 2:] = 6            # two runs
    cols[40, :] = 0                  # a memory column without accesses (hinted)
    cols[0, :] = 3                   # constant, but NOT hinted: read like any other column
    mask = (0x7fffffff << 2) | (3 << 40)
    pts = rnd(0xE7A2 + nv, 43 * nv).reshape(43, nv)
    ctx.set_option("run_aware_mask", mask)
    try:
        for it in range(2):
            job = zigz_amd.CommitJob(ctx, cols=cols)
            job.roots()
            got = job.open_all(pts)
            job.end()
            st = ctx.stats()
            assert st["run_aware_columns"] == 33
            assert st["eval_constant_columns"] == len(const) + 1, st["eval_constant_columns"]  # + column 40
            for c in range(43):
                want = O.mle_eval(P, cols[c], pts[c])
                assert int(got["values"][c]) == want, (c, int(got["values"][c]), want)
            cols[3, 12345 % N] += 1      # second round: a constant column stops being one
            const.pop(3, None)
    finally:
        ctx.set_option("run_aware_mask", 0)


@pytest.mark.gpu
def test_commit_job_survives_a_hinted_build_in_between(ctx):
    """a commit job's openings read the list counters of ITS build again (the "column is not constant" words of
    EvalSkip, the "group dropped" word).  include/zigz_hip.h allows other calls on the context between zigz_commit_begin and
    zigz_commit_open_all; a hinted zigz_merkle_commit of >= 2^15 values in that window runs the same structure passes and
    must count in counters of its own.  Trace-like columns (loops: the group is kept; busy and constant registers), every
    opening against the oracle; the tree built in between against the oracle too."""
    import zigz_amd
    nv = 15
    N = 1 << nv
    rng = np.random.default_rng(0xAD71CE)
    cols = np.zeros((43, N), dtype=np.uint64)
    body = rng.integers(0, 2013265921, size=(10, 12), dtype=np.uint64)  # a 12-step loop: the ten instruction columns repeat
    grp = [0, 1, 33, 34, 35, 36, 37, 38, 39, 42]
    for j, c in enumerate(grp):
        cols[c] = np.tile(body[j], N // 12 + 1)[:N]
    cols[1] = 0
    for c in range(2, 33):       # registers: a few busy ones, the rest constant -- but NOT constant in the tree built in between
        cols[c] = 7 * c
    cols[5] = np.repeat(rng.integers(0, 2013265921, size=N // 64, dtype=np.uint64), 64)
    cols[9] = np.repeat(rng.integers(0, 2013265921, size=N // 4096, dtype=np.uint64), 4096)
    cols[40] = 0
    cols[41] = 0
    exp = O.generate_commitments(P, O.Transcript(), cols, fast=True)
    other = np.repeat(rng.integers(0, 2013265921, size=N // 8, dtype=np.uint64), 8)  # changes everywhere, never constant
    saved = {k: ctx.get_option(k) for k in ("run_aware_mask", "cons_group_mask", "small_domain_mask")}
    try:
        ctx.set_option("small_domain_mask", (1 << 1) | (0x3f << 33) | (1 << 42))
        ctx.set_option("run_aware_mask", (0x7fffffff << 2) | (3 << 40))
        ctx.set_option("cons_group_mask", 1 | (1 << 1) | (0x7f << 33) | (1 << 42))
        ctx.set_option("cons_always", 1)
        for where in ("before_roots", "after_roots"):
            job = zigz_amd.CommitJob(ctx, cols=cols)
            if where == "after_roots":
                assert np.array_equal(job.roots(), exp["roots"])
            # a single tree whose one column is hinted as run-aware AND as a (one-column) group: clears and rewrites counters
            ctx.set_option("run_aware_mask", 1)
            ctx.set_option("cons_group_mask", 0)
            t1 = zigz_amd.SimpleMerkleTree(ctx, other)
            ctx.set_option("run_aware_mask", 0)
            ctx.set_option("cons_group_mask", 1)
            t2 = zigz_amd.SimpleMerkleTree(ctx, other)
            lv, _ = O.merkle_levels(other)
            want_root = lv[(2 * N - 2) * 32:(2 * N - 1) * 32].tobytes()
            assert t1.root_hash == want_root and t2.root_hash == want_root
            t1.deinit()
            t2.deinit()
            ctx.set_option("run_aware_mask", (0x7fffffff << 2) | (3 << 40))
            ctx.set_option("cons_group_mask", 1 | (1 << 1) | (0x7f << 33) | (1 << 42))
            if where == "before_roots":
                assert np.array_equal(job.roots(), exp["roots"])
                st = ctx.stats()  # (the stats are the last build's: in the other order the single trees have overwritten them)
                assert st["cons_columns"] == 10 and st["eval_constant_columns"] >= 29, st
            got = job.open_all(exp["points"])
            job.end()
            for k in ("values", "indices", "leaves", "siblings", "dirs"):
                assert np.array_equal(got[k], exp[k]), (where, k)
    finally:
        ctx.set_option("cons_always", 0)
        for k, v in saved.items():
            ctx.set_option(k, v)


def _trace_cols(ctx, prog, nv):
    """resident witness columns of a program's trace (must have 2^nv padded rows) -> (device pointer, stride)"""
    from zigz_amd import host
    tr = host.Trace(prog, 0x1000, None, 1 << 20)
    assert tr.num_vars == nv, (tr.num_vars, nv)
    N = max(1 << nv, 4)
    d = ctx.dev_alloc(43 * N * 4)
    tr.witness_to_device(ctx, d, N)
    return d, N


@pytest.mark.gpu
@pytest.mark.parametrize("nv", [0, 3, 5, 10, 13, 14, 15, 16, 18, 20])
def test_commit_batch_equals_single_jobs(ctx, nv):
    """zigz_commit_begin_batch: several proofs' 43 witness columns in ONE commit job -- one structure pass, one
    k_level_hash per level, one top launch, one eval, one path launch for all of them -- must give every proof exactly the
    roots and openings its own job gives (which the other tests pin to the oracle).  Flat form below 2^15 (dense trees),
    arena form from 2^15 (structure-aware levels, worst-case list room); looping programs of different lengths, the RV64IM
    mix with loads and stores, the worst-case register trace and a straight-line program whose group is DROPPED on the
    device while its neighbours' groups are kept; batches of 1, 2 and 5."""
    import programs
    import zigz_amd
    steps = (1 << nv)
    if nv >= 15:
        progs = [programs.add_xor_loop((steps - 3) // 4), programs.mixed_loop((steps - 8) // 12 - 3),
                 programs.straight_line_program(7, int(0.9 * steps)), programs.register_round_robin((steps - 2) // 31 - 1),
                 programs.add_xor_loop((steps - 3) // 4 - 777)]
    elif nv >= 3:
        progs = [programs.add_xor_loop(max((steps - 3) // 4 - i, 1 if nv > 3 else 1)) for i in range(4)] + \
                [programs.straight_line_program(9, steps - 1)]
    else:
        progs = [programs.straight_line_program(s_, 1) for s_ in range(5)]
    small = (1 << 1) | (0x3f << 33) | (1 << 42)
    masks = {"small_domain_mask": small, "run_aware_mask": (0x7fffffff << 2) | (3 << 40),
             "cons_group_mask": 1 | (1 << 1) | (0x7f << 33) | (1 << 42)}
    saved = {k: ctx.get_option(k) for k in masks}
    bufs = []
    try:
        for p_ in progs:
            bufs.append(_trace_cols(ctx, p_, nv))
        stride = bufs[0][1]
        for k, v in masks.items():
            ctx.set_option(k, v)
        ctx.set_option("cons_always", 1)
        rng = np.random.default_rng(nv)
        pts = rng.integers(0, 2013265921, size=(len(progs), 43, max(nv, 1)), dtype=np.uint64)[:, :, :nv]
        single = []
        for i, (d, _) in enumerate(bufs):
            job = zigz_amd.CommitJob(ctx, d_cols=d, ncols=43, nv=nv, col_stride=stride)
            r = job.roots()
            o = job.open_all(pts[i])
            job.end()
            single.append((r, o))
        if nv >= 15:  # the straight-line program's group does not repeat, the loops' do
            assert ctx.stats()["rebuilds"] >= 0
        for sel in ([0], [1, 2], [0, 1, 2, 3, 4], [2, 2, 0]):
            job = zigz_amd.CommitJob(ctx, d_cols_list=[bufs[i][0] for i in sel], ncols=43, nv=nv, col_stride=stride)
            r = job.roots()
            st = ctx.stats()
            o = job.open_all(np.concatenate([pts[i] for i in sel]).reshape(len(sel) * 43, nv) if nv else np.zeros((len(sel) * 43, 0), dtype=np.uint64))
            job.end()
            for z, i in enumerate(sel):
                assert np.array_equal(r[z * 43:(z + 1) * 43], single[i][0]), (sel, z, "roots")
                for key in ("values", "indices", "leaves", "siblings", "dirs"):
                    assert np.array_equal(o[key][z * 43:(z + 1) * 43], single[i][1][key]), (sel, z, key)
            if nv >= 15 and len(sel) > 1:
                assert st["run_aware_columns"] == 33 and st["keccak_permutations"] > 0
    finally:
        ctx.set_option("cons_always", 0)
        for k, v in saved.items():
            ctx.set_option(k, v)
        for d, _ in bufs:
            ctx.dev_free(d)


@pytest.mark.gpu
@pytest.mark.parametrize("ns", [1, 63, 64, 65, 4096, 4097, 70000])
def test_witness_from_32_byte_records_synthetic(ctx, ns):
    """the 32-byte step record (imm as i32, the memory triple in a side list a step refers to by index --
    zigz_trace_step32 / zigz_mem_access) must expand to the same 43 columns as the 48-byte record of the same steps and as
    the numpy restatement of witness.zig: full-range i32 immediates (min, max, -1), full-range 64-bit values, steps with and
    without a memory access in every mix (none at all, every step, the first / last step only), an index past the list
    (reads as "no access"), accesses whose address and value are 0."""
    import zigz_amd
    from zigz_amd.hip import TRACE_STEP_DTYPE, compact_steps32, NO_MEM_ACCESS
    rng = np.random.default_rng(3200 + ns)
    nv = 0 if ns == 1 else int(ns - 1).bit_length()
    N = 1 << nv
    stride = max(N, 4)
    init = rng.integers(0, 2**64, size=32, dtype=np.uint64)
    d = ctx.dev_alloc(43 * stride * 4)
    try:
        for mix in ("some", "none", "all", "ends"):
            st = np.zeros(ns, dtype=TRACE_STEP_DTYPE)
            for f in ("pc", "rd_value"):
                st[f] = rng.integers(0, 2**64, size=ns, dtype=np.uint64)
            st["imm"] = rng.integers(-2**31, 2**31, size=ns, dtype=np.int64)
            st["imm"][0] = -2**31
            st["imm"][ns // 2] = 2**31 - 1
            st["imm"][ns - 1] = -1
            for f, hi in (("opcode", 128), ("rd", 32), ("rs1", 32), ("rs2", 32), ("funct3", 8), ("funct7", 128), ("mem_is_read", 2)):
                st[f] = rng.integers(0, hi, size=ns)
            st["wr_reg"] = rng.integers(0, 32, size=ns)
            has = {"some": rng.random(ns) < 0.3, "none": np.zeros(ns, dtype=bool), "all": np.ones(ns, dtype=bool),
                   "ends": np.isin(np.arange(ns), (0, ns - 1))}[mix]
            st["mem_addr"] = np.where(has, rng.integers(0, 2**64, size=ns, dtype=np.uint64), 0)
            st["mem_value"] = np.where(has, rng.integers(0, 2**64, size=ns, dtype=np.uint64), 0)
            if has.any():  # an access at address 0 with value 0 is still an access
                k = int(np.flatnonzero(has)[0])
                st["mem_addr"][k] = 0
                st["mem_value"][k] = 0
            s32, mem = compact_steps32(st, has)
            assert s32.nbytes == 32 * ns and mem.nbytes == 16 * int(has.sum())
            if mix == "some" and ns > 2 and not has[1]:
                s32["mem_index"][1] = len(mem) + 5  # past the list: no access, like NO_MEM_ACCESS
            ctx.witness_from_steps32(s32, mem, nv, d, stride, init)
            got = ctx.download(d, 43 * stride).reshape(43, stride)[:, :N]
            i0 = init.copy()
            i0[0] = 0
            exp = _expand_reference(st, ns, N, i0)
            bad = np.argwhere(got != exp)
            assert bad.size == 0, (mix, bad[:5])
            ctx.witness_from_steps(st, nv, d, stride, init)  # ... and the 48-byte record of the same steps
            assert np.array_equal(ctx.download(d, 43 * stride).reshape(43, stride)[:, :N], exp), mix
    finally:
        ctx.dev_free(d)


@pytest.mark.gpu
@pytest.mark.parametrize("maker,arg", [("mixed_loop", 700), ("fibonacci", 60), ("add_xor_loop", 2000)])
def test_witness_from_32_byte_records_vs_oracle(ctx, maker, arg):
    """Real traces through the 32-byte record: the host VM's steps, compacted (a side-list entry for every LOAD / STORE step),
    expanded on the device == the oracle's VM + WitnessGenerator; and a proof whose witness is built from them inside its GPU
    slot (the service's upload path) is byte-identical to the oracle's."""
    import hashlib
    import programs
    from zigz_amd import host
    from zigz_amd.hip import compact_steps32
    made = getattr(programs, maker)(arg)
    prog, inp = made if isinstance(made, tuple) else (made, None)
    tr = host.Trace(prog, 0x1000, None, 1 << 20, inp)
    cols, nv, ns = O.witness_from_program(P, prog, 0x1000, None, 1 << 20, inp)
    st, init = tr.steps()
    has = np.isin(st["opcode"], (0x03, 0x23))
    if maker == "mixed_loop":
        assert has.any() and (st["mem_addr"][~has] == 0).all()
    s32, mem = compact_steps32(st, has)
    N = 1 << nv
    d = ctx.dev_alloc(43 * max(N, 4) * 4)
    try:
        ctx.witness_from_steps32(s32, mem, nv, d, max(N, 4), init)
        assert np.array_equal(ctx.download(d, 43 * max(N, 4)).reshape(43, max(N, 4))[:, :N], cols)
    finally:
        ctx.dev_free(d)
    tr.pin(ctx)  # builds + page-locks the 32-byte form inside the host mirror
    slots = host.Slots(0, 1)
    try:
        proof, _, _ = tr.prove_slots(slots, None, 0)
        assert hashlib.sha3_256(proof.tobytes()).hexdigest() == hashlib.sha3_256(O.prove(P, prog, 0x1000, None, 1 << 20, inp)[0]).hexdigest()
    finally:
        slots.close()
        del tr


@pytest.mark.gpu
def test_group_probe_left_out_after_kept_builds_then_a_trace_that_does_not_repeat(ctx):
    """A context whose last builds all kept their content-addressed group leaves the probe pass out (two launches that find out
    what is known).  The trace that does not repeat after all must still come out right: the full insert drops the group, the
    build is repeated with slabs for its columns -- roots and openings equal a fresh context's (which probes), and the looping
    traces before and after it equal theirs too."""
    import programs
    import zigz_amd
    nv = 16
    steps = 1 << nv
    loop_a, loop_b = programs.add_xor_loop((steps - 3) // 4), programs.mixed_loop((steps - 8) // 12 - 3)
    straight = programs.straight_line_program(11, int(0.9 * steps))
    small = (1 << 1) | (0x3f << 33) | (1 << 42)
    masks = {"small_domain_mask": small, "run_aware_mask": (0x7fffffff << 2) | (3 << 40),
             "cons_group_mask": 1 | (1 << 1) | (0x7f << 33) | (1 << 42)}
    fresh = zigz_amd.Context(0)
    saved = {k: ctx.get_option(k) for k in masks}
    bufs = []
    try:
        for c in (ctx, fresh):
            for k, v in masks.items():
                c.set_option(k, v)
        pts = np.random.default_rng(5).integers(0, 2013265921, size=(43, nv), dtype=np.uint64)

        def run(c, d, stride):
            job = zigz_amd.CommitJob(c, d_cols=d, ncols=43, nv=nv, col_stride=stride)
            r = job.roots()
            o = job.open_all(pts)
            job.end()
            return r, o
        seq = [loop_a, loop_a, loop_b, loop_a, straight, loop_a, straight, loop_b]
        for i, prog in enumerate(seq):
            d, stride = _trace_cols(ctx, prog, nv)
            bufs.append(d)
            got = run(ctx, d, stride)
            f2 = zigz_amd.Context(0)  # no history at all: probes
            try:
                for k, v in masks.items():
                    f2.set_option(k, v)
                want = run(f2, d, stride)
            finally:
                f2.close()
            assert np.array_equal(got[0], want[0]), (i, "roots")
            for key in ("values", "indices", "leaves", "siblings", "dirs"):
                assert np.array_equal(got[1][key], want[1][key]), (i, key)
    finally:
        for k, v in saved.items():
            ctx.set_option(k, v)
        for d in bufs:
            ctx.dev_free(d)
        fresh.close()


@pytest.mark.gpu
@pytest.mark.parametrize("ns", [1, 2, 3, 64, 65, 1000, 4097, 70000])
def test_witness_from_16_byte_records_synthetic(ctx, ns):
    """The 16-byte step record (zigz_trace_step16: pc as an offset into a CODE TABLE that carries the seven instruction fields
    once per pc, wr_reg and the side-list index packed into one word, mem_is_read in the pc word's bit 0) must expand to the same
    43 columns as the 48-byte record of the same steps and as the numpy restatement of witness.zig: programs of 1, 7 and ~ns / 3
    instructions at a 64-bit base, full-range i32 immediates, every access mix, an index past the side list and the "none" index,
    a pc_word past the code table (zero fields), wr_reg 0 and 31."""
    import zigz_amd
    from zigz_amd.hip import TRACE_STEP_DTYPE, compact_steps16, NO_MEM_ACCESS16
    rng = np.random.default_rng(1600 + ns)
    nv = 0 if ns == 1 else int(ns - 1).bit_length()
    N = 1 << nv
    stride = max(N, 4)
    init = rng.integers(0, 2**64, size=32, dtype=np.uint64)
    d = ctx.dev_alloc(43 * stride * 4)
    try:
        for mix, ninst in (("some", 7), ("none", 1), ("all", max(1, ns // 3)), ("ends", 7)):
            base = int(rng.integers(0, 2**62)) & ~3
            # the program: ninst decoded instructions; a step executes one of them
            code = {f: rng.integers(0, hi, size=ninst) for f, hi in (("opcode", 128), ("rd", 32), ("rs1", 32), ("rs2", 32), ("funct3", 8),
                                                                    ("funct7", 128))}
            code["imm"] = rng.integers(-2**31, 2**31, size=ninst, dtype=np.int64)
            code["imm"][0] = -2**31
            code["imm"][ninst // 2] = 2**31 - 1
            code["imm"][ninst - 1] = -1 if ninst > 2 else code["imm"][ninst - 1]
            at = rng.integers(0, ninst, size=ns)
            at[0] = ninst - 1  # (the table's last entry is used)
            st = np.zeros(ns, dtype=TRACE_STEP_DTYPE)
            st["pc"] = np.uint64(base) + (at * 4).astype(np.uint64)
            st["rd_value"] = rng.integers(0, 2**64, size=ns, dtype=np.uint64)
            for f in code:
                st[f] = code[f][at]
            st["wr_reg"] = rng.integers(0, 32, size=ns)
            st["wr_reg"][0] = 31
            st["wr_reg"][ns - 1] = 0
            st["mem_is_read"] = rng.integers(0, 2, size=ns)
            has = {"some": rng.random(ns) < 0.3, "none": np.zeros(ns, dtype=bool), "all": np.ones(ns, dtype=bool),
                   "ends": np.isin(np.arange(ns), (0, ns - 1))}[mix]
            st["mem_addr"] = np.where(has, rng.integers(0, 2**64, size=ns, dtype=np.uint64), 0)
            st["mem_value"] = np.where(has, rng.integers(0, 2**64, size=ns, dtype=np.uint64), 0)
            if has.any():
                k = int(np.flatnonzero(has)[0])
                st["mem_addr"][k] = 0
                st["mem_value"][k] = 0
            c = compact_steps16(st, has)
            assert c is not None
            s16, mem, cbase, ctab = c
            assert s16.nbytes == 16 * ns and mem.nbytes == 16 * int(has.sum()) and cbase == int(st["pc"].min())
            if mix == "some" and ns > 2 and not has[1]:
                s16["mem_wr"][1] = (s16["mem_wr"][1] & ~np.uint32(NO_MEM_ACCESS16)) | np.uint32(len(mem) + 5)  # past the list: none
            ctx.witness_from_steps16(s16, mem, cbase, ctab, nv, d, stride, init)
            got = ctx.download(d, 43 * stride).reshape(43, stride)[:, :N]
            i0 = init.copy()
            i0[0] = 0
            exp = _expand_reference(st, ns, N, i0)
            bad = np.argwhere(got != exp)
            assert bad.size == 0, (mix, bad[:5])
            ctx.witness_from_steps(st, nv, d, stride, init)  # ... and the 48-byte record of the same steps
            assert np.array_equal(ctx.download(d, 43 * stride).reshape(43, stride)[:, :N], exp), mix
            if mix == "some" and ns > 3:  # a pc past the table: that step's seven fields read as zero
                k = ns // 2
                s16["pc_word"][k] = np.uint32(4 * len(ctab) + 8) | (s16["pc_word"][k] & np.uint32(1))
                st2 = st.copy()
                st2["pc"][k] = np.uint64(cbase + 4 * len(ctab) + 8)
                for f in code:
                    st2[f][k] = 0
                ctx.witness_from_steps16(s16, mem, cbase, ctab, nv, d, stride, init)
                # (the access of step 1 was cut off above: its columns are those of a step without one)
                if not has[1]:
                    pass
                exp2 = _expand_reference(st2, ns, N, i0)
                assert np.array_equal(ctx.download(d, 43 * stride).reshape(43, stride)[:, :N], exp2)
        # what the form refuses: a pc off the grid, one pc with two decodings
        st = np.zeros(4, dtype=TRACE_STEP_DTYPE)
        st["pc"] = [0x1000, 0x1004, 0x1006, 0x1008]
        assert compact_steps16(st, np.zeros(4, dtype=bool)) is None
        st["pc"] = [0x1000, 0x1004, 0x1000, 0x1008]
        st["rd"] = [1, 2, 3, 4]
        assert compact_steps16(st, np.zeros(4, dtype=bool)) is None
    finally:
        ctx.dev_free(d)


def _self_modifying_program():
    """Executes the word at 0x1010 twice: as ADDI x2,x0,1, then -- after storing another instruction over it -- as ADDI x2,x0,2."""
    import programs as pg
    new = pg._I(0x13, 2, 0, 0, 2)                        # ADDI x2,x0,2
    w = pg._li(3, new)                                   # 0x1000, 0x1004: x3 = the new instruction word
    w += [pg._U(0x37, 4, 1)]                             # 0x1008: LUI x4,1 -> x4 = 0x1000
    w += [pg._I(0x13, 5, 0, 5, 1)]                       # 0x100c: ADDI x5,x5,1 (pass counter)
    w += [pg._I(0x13, 2, 0, 0, 1)]                       # 0x1010: ADDI x2,x0,1  <- rewritten
    w += [pg._S(2, 4, 3, 0x10)]                          # 0x1014: SW x3,0x10(x4)
    w += [pg._I(0x13, 6, 0, 0, 2)]                       # 0x1018: ADDI x6,x0,2
    w += [pg._B(1, 5, 6, -16)]                           # 0x101c: BNE x5,x6,-16 -> 0x100c (second pass)
    w += [pg._I(0x13, 7, 0, 2, 0)]                       # 0x1020: ADDI x7,x2,0
    return pg._pack(w)


@pytest.mark.gpu
@pytest.mark.parametrize("maker,arg", [("mixed_loop", 700), ("fibonacci", 60), ("add_xor_loop", 2000), ("random_program", 7),
                                       ("self_modifying", 0)])
def test_witness_from_16_byte_records_vs_oracle(ctx, maker, arg):
    """Real traces through the 16-byte record: the host VM's steps, compacted (code table from the decoded fields of the steps
    themselves), expanded on the device == the oracle's VM + WitnessGenerator; the host mirror picks the form for a pinned trace
    and a proof whose witness is built from it inside its GPU slot is byte-identical to the oracle's.  A program that REWRITES an
    instruction it executes twice does not fit the form (one pc, two decodings): the mirror falls back to the 32-byte record, the
    proof is the oracle's all the same."""
    import hashlib
    import programs
    from zigz_amd import host
    from zigz_amd.hip import compact_steps16
    if maker == "self_modifying":
        made = _self_modifying_program()
    elif maker == "random_program":
        made = programs.random_program(np.random.default_rng(arg), n_insts=80)
    else:
        made = getattr(programs, maker)(arg)
    prog, inp = made if isinstance(made, tuple) else (made, None)
    tr = host.Trace(prog, 0x1000, None, 1 << 20, inp)
    cols, nv, ns = O.witness_from_program(P, prog, 0x1000, None, 1 << 20, inp)
    st, init = tr.steps()
    has = np.isin(st["opcode"], (0x03, 0x23))
    c = compact_steps16(st, has)
    N = 1 << nv
    if maker == "self_modifying":
        assert c is None and int(cols[1 + 2].max()) == 2  # (column 1 + r is xr: x2 ended as 2 -- the rewritten instruction did run)
    else:
        s16, mem, cbase, ctab = c
        d = ctx.dev_alloc(43 * max(N, 4) * 4)
        try:
            ctx.witness_from_steps16(s16, mem, cbase, ctab, nv, d, max(N, 4), init)
            assert np.array_equal(ctx.download(d, 43 * max(N, 4)).reshape(43, max(N, 4))[:, :N], cols)
        finally:
            ctx.dev_free(d)
    tr.pin(ctx)  # builds + page-locks the forms inside the host mirror
    form, nbytes = tr.upload_form()
    assert form == (32 if maker == "self_modifying" else 16), (form, nbytes)
    if form == 16:
        assert nbytes == 16 * ns + 16 * int(has.sum()) + 12 * len(ctab)
    slots = host.Slots(0, 1)
    try:
        proof, _, _ = tr.prove_slots(slots, None, 0)
        assert hashlib.sha3_256(proof.tobytes()).hexdigest() == hashlib.sha3_256(O.prove(P, prog, 0x1000, None, 1 << 20, inp)[0]).hexdigest()
    finally:
        slots.close()
        del tr


@pytest.mark.gpu
@pytest.mark.parametrize("nv,ncols", [(1, 86), (2, 86), (2, 688), (3, 344), (2, 43), (3, 1000)])
def test_commit_job_many_tiny_columns(ctx, nv, ncols):
    """The eval of a commit job over MANY columns of 2, 4 or 8 rows (what a batched job of small traces is): values against the
    oracle's eval, five times over (the fold buffers were once sized without the 4-element column stride, and N = 4 let a
    launch overwrite inputs other workgroups were still reading whenever ncols * 2 was a multiple of 4)."""
    import zigz_amd
    N = 1 << nv
    cols = rnd(0x71A7 + 31 * nv + ncols, ncols * N).reshape(ncols, N)
    pts = rnd(0x71A8 + nv + ncols, ncols * nv).reshape(ncols, nv)
    want = [O.mle_eval(P, cols[c], pts[c]) for c in range(ncols)]
    for _ in range(5):
        job = zigz_amd.CommitJob(ctx, cols=cols)
        job.roots()
        got = job.open_all(pts)
        job.end()
        assert [int(v) for v in got["values"]] == want
        assert np.array_equal(got["indices"], pts[:, 0] % np.uint64(N))
        assert np.array_equal(got["leaves"], cols[np.arange(ncols), (pts[:, 0] % np.uint64(N)).astype(np.int64)])
