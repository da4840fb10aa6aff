"""Oracle-backed stand-ins for zigz_amd.shard's compute objects, so the N > 1 partition / exchange logic
can be exercised with gloo on CPU (the product engines need a GPU).  Test infrastructure only."""
import numpy as np

import oracle_lib as O

P = O.P_BB


class _FakeJob:
    def __init__(self, cols, nv):
        self.cols = np.ascontiguousarray(cols, dtype=np.uint64)
        self.nv = nv

    def roots(self):
        return np.stack([np.frombuffer(O.merkle_build(c)[0], dtype=np.uint8) for c in self.cols])

    def open_all(self, points):
        nc, nv = len(self.cols), self.nv
        out = dict(values=np.zeros(nc, dtype=np.uint64), indices=np.zeros(nc, dtype=np.uint64),
                   leaves=np.zeros(nc, dtype=np.uint64), siblings=np.zeros((nc, nv, 32), dtype=np.uint8),
                   dirs=np.zeros((nc, nv), dtype=np.uint8))
        for k in range(nc):
            val, idx, sib, dirs, leaf = O.commit_open(P, self.cols[k], [int(x) for x in points[k]])
            out["values"][k], out["indices"][k], out["leaves"][k] = val, idx, leaf
            out["siblings"][k] = np.frombuffer(sib, dtype=np.uint8).reshape(nv, 32)
            out["dirs"][k] = np.frombuffer(dirs, dtype=np.uint8)
        return out

    def end(self):
        pass


class FakeEngine:
    def begin(self, local_cols, ncols, nv):
        assert len(local_cols) == ncols
        return _FakeJob(local_cols, nv)


class FakeOps:
    def upload(self, values):
        return np.array(values, dtype=np.uint64)

    def download(self, t, n):
        return t[:n]

    def half_sums(self, t, n):
        h = max(n // 2, 1)
        return [int(t[:h].sum(dtype=np.uint64) % np.uint64(P)), int(t[h:n].sum(dtype=np.uint64) % np.uint64(P)) if n > 1 else 0]

    def bind(self, t, n, r):
        return np.array(O.mle_partial_eval(P, t[:n], r), dtype=np.uint64)

    def bind_sums(self, t, n, r):
        o = self.bind(t, n, r)
        return o, self.half_sums(o, n // 2)

    def sumcheck_tail(self, t, n, tr):
        rounds, point = [], []
        cur = t[:n]
        while n > 1:
            c = O.mle_round_poly(P, cur)
            rounds += c
            tr.append_field(c[0]); tr.append_field(c[1])
            ch = tr.challenge()
            point.append(ch)
            cur = self.bind(cur, n, ch)
            n //= 2
        return rounds, point, int(cur[0])


class FakeRadixOps:
    """The three data passes of the radix sumcheck (zigz_radix_ops) on a numpy table, exact Python-int arithmetic:
    what k_block_sums / k_radix_fold + k_radix_finalize do on the GPU."""

    def __init__(self, table):
        self.t = [int(x) for x in table]

    def block_sums(self, k):
        m = len(self.t) >> k
        return [sum(self.t[b * m:(b + 1) * m]) for b in range(1 << k)]  # exact, unreduced (like the u64 sums)

    def fold(self, k, weights, k_next):
        m = len(self.t) >> k
        self.t = [sum(weights[b] * self.t[b * m + i] for b in range(1 << k)) % P for i in range(m)]
        if not k_next:
            return None
        m2 = m >> k_next
        return [sum(self.t[b * m2:(b + 1) * m2]) for b in range(1 << k_next)]

    def read_tail(self, m):
        assert m == len(self.t)
        return self.t


class FakeTreeOps:
    """Local subtree work of shard.RowShardedMerkle on the oracle."""

    def commit(self, values):
        v = np.ascontiguousarray(values, dtype=np.uint64)
        return O.merkle_build(v)[0], v

    def open(self, v, index):
        h = (len(v) - 1).bit_length()
        sib, dirs, leaf = O.merkle_open(v, index)
        return (np.frombuffer(sib, dtype=np.uint8).reshape(h, 32), np.frombuffer(dirs, dtype=np.uint8), leaf)

    def sha3(self, data):
        import hashlib
        return hashlib.sha3_256(data).digest()

    def destroy(self, v):
        pass
