// Transcript, Multilinear, sumcheck, Merkle, commitment scheme, Lasso: thin host objects whose
// arithmetic is delegated to libzigz_hip.so.
#include <cstring>

#include "zigz_host.hpp"

namespace zigz {

void check(zigz_ctx *ctx, zigz_status s) {
    if (s == ZIGZ_OK) return;
    std::string msg = std::string("error.") + zigz_status_name(s);
    if (ctx && (s >= 100)) msg += std::string(": ") + zigz_last_error(ctx);
    throw Error(s, msg);
}

// ---------------------------------------------------------------- transcript
FiatShamirTranscript::FiatShamirTranscript() : t_(zigz_transcript_new()) {
    if (!t_) throw Error(ZIGZ_ERR_OUT_OF_MEMORY, "error.OutOfMemory");
}
FiatShamirTranscript::~FiatShamirTranscript() { zigz_transcript_free(t_); }
void FiatShamirTranscript::reset() {
    zigz_transcript_free(t_);
    t_ = zigz_transcript_new();
    if (!t_) throw Error(ZIGZ_ERR_OUT_OF_MEMORY, "error.OutOfMemory");
}
void FiatShamirTranscript::appendFieldElement(F e) { zigz_transcript_append_field(t_, e); }
void FiatShamirTranscript::appendFieldElements(const std::vector<F> &es) {
    for (F e : es) zigz_transcript_append_field(t_, e);
}
void FiatShamirTranscript::appendBytes(const void *data, size_t len) {
    zigz_transcript_append_bytes(t_, (const uint8_t *)data, len);
}
void FiatShamirTranscript::appendBytes(const char *s) { appendBytes(s, strlen(s)); }
void FiatShamirTranscript::appendTaggedCounter(const char *tag, uint64_t start, uint64_t count) {
    zigz_transcript_append_tagged_counter(t_, (const uint8_t *)tag, strlen(tag), start, count);
}
F FiatShamirTranscript::challenge() { return zigz_transcript_challenge(t_); }

// ---------------------------------------------------------------- Multilinear
static size_t log2_floor(size_t n) { size_t l = 0; while (n > 1) { n >>= 1; l++; } return l; }

Multilinear Multilinear::init(zigz_ctx *ctx, const std::vector<F> &evals) {
    if (evals.empty()) throw Error(ZIGZ_ERR_EMPTY_EVALUATIONS, "error.EmptyEvaluations");
    if (evals.size() & (evals.size() - 1)) throw Error(ZIGZ_ERR_LENGTH_NOT_POWER_OF_TWO, "error.LengthNotPowerOfTwo");
    Multilinear m;
    m.evaluations = evals;  // allocator.dupe, multilinear.zig:47
    m.num_vars = log2_floor(evals.size());
    m.ctx = ctx;
    return m;
}
F Multilinear::eval(const std::vector<F> &point) const {
    F out = 0;
    check(ctx, zigz_mle_eval(ctx, evaluations.data(), evaluations.size(), point.data(), point.size(), &out));
    return out;
}
Multilinear Multilinear::partialEval(F r) const {
    Multilinear q;
    q.ctx = ctx;
    q.num_vars = num_vars ? num_vars - 1 : 0;
    q.evaluations.resize(evaluations.size() / 2);
    check(ctx, zigz_mle_bind(ctx, evaluations.data(), evaluations.size(), r, q.evaluations.data()));
    return q;
}
F Multilinear::sumOverHypercube() const {
    F out = 0;
    check(ctx, zigz_mle_sum(ctx, evaluations.data(), evaluations.size(), &out));
    return out;
}
std::vector<F> Multilinear::roundPolynomial() const {
    std::vector<F> c(2);
    check(ctx, zigz_mle_round_poly(ctx, evaluations.data(), evaluations.size(), c.data()));
    return c;
}

// ---------------------------------------------------------------- sumcheck
static inline F fadd(F a, F b) { F s = a + b; return s >= BABYBEAR ? s - BABYBEAR : s; }
static inline F fmul(F a, F b) { return (F)(((unsigned __int128)a * b) % BABYBEAR); }

F evalUnivariateCoeffs(const F *c, size_t n, F x) {
    if (n == 0) return 0;
    F r = c[n - 1];
    for (size_t i = n - 1; i > 0; i--) r = fadd(fmul(r, x), c[i - 1]);
    return r;
}

std::vector<uint8_t> SumcheckProof::toBytes() const {
    std::vector<uint8_t> b((1 + 2 * num_vars + num_vars + 1) * 8);
    size_t off = 0;
    auto w = [&](uint64_t v) { memcpy(b.data() + off, &v, 8); off += 8; };
    w(num_vars);
    for (auto &rp : round_polynomials) { w(rp[0]); w(rp[1]); }
    for (F x : final_point) w(x);
    w(final_eval);
    return b;
}

static SumcheckProof pack(size_t nv, const std::vector<F> &rounds, const std::vector<F> &point, F fe) {
    SumcheckProof p;
    p.num_vars = nv;
    p.round_polynomials.resize(nv);
    for (size_t i = 0; i < nv; i++) p.round_polynomials[i] = {rounds[2 * i], rounds[2 * i + 1]};
    p.final_point = point;
    p.final_eval = fe;
    return p;
}

SumcheckProof SumcheckProver::prove(const Multilinear &poly) {
    size_t nv = poly.num_vars;
    std::vector<F> rounds(2 * nv + 2), point(nv + 1);
    F fe = 0;
    check(poly.ctx, zigz_sumcheck_prove(poly.ctx, poly.evaluations.data(), poly.evaluations.size(), rounds.data(),
                                       point.data(), &fe));
    point.resize(nv);
    return pack(nv, rounds, point, fe);
}
SumcheckProof SumcheckProver::proveInteractive(const Multilinear &poly, const std::vector<F> &challenges) {
    size_t nv = poly.num_vars;
    std::vector<F> rounds(2 * nv + 2), point(nv + 1);
    F fe = 0;
    check(poly.ctx, zigz_sumcheck_prove_interactive(poly.ctx, poly.evaluations.data(), poly.evaluations.size(),
                                                   challenges.data(), challenges.size(), rounds.data(), point.data(), &fe));
    point.resize(nv);
    return pack(nv, rounds, point, fe);
}

// ---------------------------------------------------------------- Merkle
SimpleMerkleTree SimpleMerkleTree::build(zigz_ctx *ctx, const std::vector<F> &values) {
    SimpleMerkleTree t;
    t.ctx_ = ctx;
    check(ctx, zigz_merkle_commit(ctx, values.data(), values.size(), t.root_hash.data(), &t.height, &t.t_));
    return t;
}
SimpleMerkleTree::SimpleMerkleTree(SimpleMerkleTree &&o) noexcept { *this = std::move(o); }
SimpleMerkleTree &SimpleMerkleTree::operator=(SimpleMerkleTree &&o) noexcept {
    if (this != &o) {
        if (t_) zigz_merkle_destroy(ctx_, t_);
        ctx_ = o.ctx_; t_ = o.t_; root_hash = o.root_hash; height = o.height;
        o.t_ = nullptr;
    }
    return *this;
}
SimpleMerkleTree::~SimpleMerkleTree() {
    if (t_) zigz_merkle_destroy(ctx_, t_);
}
MerkleOpening SimpleMerkleTree::open(size_t index) const {
    MerkleOpening o;
    o.index = index;
    o.path.siblings.resize(height);
    o.path.directions.resize(height);
    check(ctx_, zigz_merkle_open(ctx_, t_, index, height ? o.path.siblings[0].data() : nullptr,
                                 height ? o.path.directions.data() : nullptr, &o.value));
    return o;
}
bool SimpleMerkleTree::verify(const Hash &root, const MerkleOpening &proof) {
    uint8_t cur[32], buf[64];
    uint64_t v = proof.value;
    zigz_sha3_256((const uint8_t *)&v, 8, cur);  // hashLeaf, little-endian host
    for (size_t l = 0; l < proof.path.siblings.size(); l++) {
        if (proof.path.directions[l]) { memcpy(buf, proof.path.siblings[l].data(), 32); memcpy(buf + 32, cur, 32); }
        else { memcpy(buf, cur, 32); memcpy(buf + 32, proof.path.siblings[l].data(), 32); }
        zigz_sha3_256(buf, 64, cur);
    }
    return memcmp(cur, root.data(), 32) == 0;
}

CommitmentScheme::Commit CommitmentScheme::commit(const Multilinear &poly) {
    SimpleMerkleTree tree = SimpleMerkleTree::build(poly.ctx, poly.evaluations);
    Hash root = tree.getRoot();
    return Commit{root, poly.num_vars, std::move(tree)};
}
PolyOpeningProof CommitmentScheme::open(const Multilinear &poly, const SimpleMerkleTree &tree, const std::vector<F> &point) {
    PolyOpeningProof p;
    size_t nv = point.size();
    p.point = point;
    p.merkle_proof.path.siblings.resize(nv);
    p.merkle_proof.path.directions.resize(nv);
    uint64_t index = 0;
    check(poly.ctx, zigz_commit_open(poly.ctx, poly.evaluations.data(), poly.evaluations.size(), tree.handle(), point.data(),
                                     nv, &p.value, &index, nv ? p.merkle_proof.path.siblings[0].data() : nullptr,
                                     nv ? p.merkle_proof.path.directions.data() : nullptr, &p.merkle_proof.value));
    p.merkle_proof.index = index;
    return p;
}
bool CommitmentScheme::verify(const Hash &commitment, size_t num_vars, const PolyOpeningProof &proof) {
    if (proof.point.size() != num_vars) return false;  // polynomial_commit.zig:123-125
    return SimpleMerkleTree::verify(commitment, proof.merkle_proof);
}

// ---------------------------------------------------------------- Lasso
static DenseTable build_table(int kind, size_t bits) {
    DenseTable t;
    t.num_inputs = 2; t.num_outputs = 1;
    uint64_t m = (uint64_t)1 << bits;
    t.entries.reserve((size_t)1 << (2 * bits));
    for (uint64_t a = 0; a < m; a++)
        for (uint64_t b = 0; b < m; b++) {
            uint64_t r = kind == 0 ? (a + b) % m : kind == 1 ? (a ^ b) : (a & b);
            t.entries.push_back(TableEntry{{finit(a), finit(b)}, {finit(r)}});
        }
    return t;
}
DenseTable buildAddTable(size_t bits) { return build_table(0, bits); }
DenseTable buildXorTable(size_t bits) { return build_table(1, bits); }
DenseTable buildAndTable(size_t bits) { return build_table(2, bits); }

static void flatten(const DenseTable &table, const std::vector<LookupQuery> &queries, std::vector<F> &tf,
                    std::vector<F> &qf, size_t &n_in, size_t &n_out) {
    n_in = table.num_inputs; n_out = table.num_outputs;
    if (!table.entries.empty()) { n_in = table.entries[0].inputs.size(); n_out = table.entries[0].outputs.size(); }
    for (auto &e : table.entries) { tf.insert(tf.end(), e.inputs.begin(), e.inputs.end()); tf.insert(tf.end(), e.outputs.begin(), e.outputs.end()); }
    for (auto &q : queries) { qf.insert(qf.end(), q.inputs.begin(), q.inputs.end()); qf.insert(qf.end(), q.expected_outputs.begin(), q.expected_outputs.end()); }
}

static LassoProofFull lasso_run(zigz_ctx *ctx, const DenseTable &table, const std::vector<LookupQuery> &queries,
                                const std::vector<size_t> *mapping) {
    std::vector<F> tf, qf;
    size_t n_in, n_out;
    flatten(table, queries, tf, qf, n_in, n_out);
    size_t padded = 1;
    while (padded < queries.size()) padded <<= 1;
    size_t nvmax = log2_floor(padded);
    std::vector<F> rounds(2 * nvmax + 2), point(nvmax + 1);
    LassoProofFull out;
    size_t nv = 0;
    F fe = 0;
    if (mapping) {
        std::vector<uint64_t> m(mapping->begin(), mapping->end());
        check(ctx, zigz_lasso_prove_with_mapping(ctx, tf.data(), table.entries.size(), qf.data(), queries.size(), n_in, n_out,
                                                 m.data(), m.size(), &nv, rounds.data(), point.data(), &fe,
                                                 out.query_commitment.data(), out.table_commitment.data()));
    } else {
        check(ctx, zigz_lasso_prove(ctx, tf.data(), table.entries.size(), qf.data(), queries.size(), n_in, n_out, &nv,
                                    rounds.data(), point.data(), &fe, out.query_commitment.data(), out.table_commitment.data()));
    }
    point.resize(nv);
    out.sumcheck_proof = pack(nv, rounds, point, fe);
    out.num_lookups = queries.size();
    return out;
}
LassoProofFull LassoProver::prove(zigz_ctx *ctx, const DenseTable &table, const std::vector<LookupQuery> &queries) {
    return lasso_run(ctx, table, queries, nullptr);
}
LassoProofFull LassoProver::proveWithMapping(zigz_ctx *ctx, const DenseTable &table, const std::vector<LookupQuery> &queries,
                                             const std::vector<size_t> &mapping) {
    return lasso_run(ctx, table, queries, &mapping);
}

}  // namespace zigz
