// WitnessGenerator, Proof, Prover, BinarySerializer, Verifier: the host orchestration of
// src/prover/prover.zig with its exact Fiat-Shamir schedule.  The 43-column
// Merkle/eval/open work runs on the GPU through the zigz_commit_* job of the C ABI and overlaps the
// sequential O(L) transcript absorption of the Lasso placeholders.
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

#include "zigz_host.hpp"
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace zigz {

static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// One helper thread per proving thread, started on first use and kept: it runs the closure a proof hands it (the early
// sections of the serialised proof, underneath the transcript), so that a proof does not create and destroy a thread -- with
// 80 proving threads in one process that is 2 000 clone / 8 MB stack map / unmap cycles per second on the shared address space.
namespace {
class Helper {
  public:
    void run(std::function<void()> f) {
        std::unique_lock<std::mutex> lk(m_);
        if (!th_.joinable()) th_ = std::thread([this] { loop(); });
        job_ = std::move(f);
        state_ = 1;
        err_ = nullptr;
        cv_.notify_all();
    }
    void wait() {  // returns when the closure has run; rethrows what it threw
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [this] { return state_ == 0; });
        if (err_) {
            std::exception_ptr e = err_;
            err_ = nullptr;
            std::rethrow_exception(e);
        }
    }
    ~Helper() {
        {
            std::lock_guard<std::mutex> lk(m_);
            quit_ = true;
        }
        cv_.notify_all();
        if (th_.joinable()) th_.join();
    }

  private:
    void loop() {
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            cv_.wait(lk, [this] { return quit_ || state_ == 1; });
            if (state_ != 1) return;  // quit
            std::function<void()> f = std::move(job_);
            state_ = 2;
            lk.unlock();
            std::exception_ptr e;
            try { f(); } catch (...) { e = std::current_exception(); }
            lk.lock();
            err_ = e;
            state_ = 0;
            cv_.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::function<void()> job_;
    std::exception_ptr err_;
    int state_ = 0;  // 0 idle, 1 handed over, 2 running
    bool quit_ = false;
    std::thread th_;
};
thread_local Helper t_helper;
}  // namespace

static size_t log2_int_ceil(size_t n) {  // std.math.log2_int_ceil; n may come from an untrusted proof header
    size_t l = 0;
    while (l < 63 && ((size_t)1 << l) < n) l++;
    return (l == 63 && ((size_t)1 << 63) < n) ? 64 : l;
}

// ---------------------------------------------------------------- witness (witness.zig:29-270)
Witness WitnessGenerator::generate(const ExecutionTrace &trace) {
    Witness w;
    const size_t ns = trace.stepCount();
    w.num_steps = ns;
    w.num_vars = ns == 0 ? 0 : log2_int_ceil(ns);
    const size_t N = (size_t)1 << w.num_vars;
    w.columns.assign(ROW_WORDS * N, 0);
    if (ns == 0) return w;
    const std::vector<uint64_t> rows = trace.expandRows();
    for (size_t c = 0; c < ROW_WORDS; c++) {
        F *col = w.columns.data() + c * N;
        for (size_t i = 0; i < ns; i++) col[i] = finit(rows[i * ROW_WORDS + c]);
        if (c <= 32) {  // pc and registers repeat the last value (:80-87,116-123); the rest pads with 0
            const F last = col[ns - 1];
            for (size_t i = ns; i < N; i++) col[i] = last;
        }
    }
    return w;
}

// ---------------------------------------------------------------- Proof (proof.zig)
Proof Proof::init(size_t num_steps) {  // :224-261
    Proof p;
    const size_t nv = log2_int_ceil(num_steps);
    p.constraint_proof.num_vars = nv;
    p.constraint_proof.num_coeffs = 4;  // degree 3
    p.constraint_proof.round_polynomials.assign(nv * 4, 0);
    p.constraint_proof.final_point.assign(nv, 0);
    p.witness_commitments.resize(ZIGZ_NUM_COLUMNS);
    for (auto &o : p.witness_commitments) o.point.assign(nv, 0);
    p.metadata.num_steps = num_steps;
    p.metadata.num_vars = nv;
    return p;
}

LassoProof Proof::lookupAt(size_t i) const {
    if (i >= lookup_placeholders) return lookup_proofs.at(i - lookup_placeholders);
    LassoProof l;  // prover.zig:302-349
    l.table_id = (uint32_t)i;
    l.num_lookups = 1;
    l.multiset_proof.num_vars = 0;
    l.multiset_proof.num_coeffs = 3;
    l.multiset_proof.final_eval = 0;
    return l;
}

size_t Proof::estimateSize() const {  // :279-312
    size_t size = 32 + 8 + 8 + 8;
    if (public_io.initial_regs) size += public_io.initial_regs->size() * 8;
    if (public_io.final_regs) size += public_io.final_regs->size() * 8;
    size += metadata.num_vars * 4 * 8 + metadata.num_vars * 8 + 8;
    for (size_t i = 0; i < lookupCount(); i++) size += 4 + 8 + lookupAt(i).multiset_proof.num_vars * 3 * 8;
    size += witness_commitments.size() * 32 + witness_commitments.size() * 20 * 32;
    return size;
}

// ---------------------------------------------------------------- Prover
bool Prover::defaultSmallDomain() {
    const char *e = getenv("ZIGZ_DENSE_MERKLE");
    return !(e && e[0] == '1');
}

int Prover::defaultRunAware() {
    if (!defaultSmallDomain()) return 0;
    const char *e = getenv("ZIGZ_RUN_AWARE");
    if (!e || !e[0]) return 4;
    return strcmp(e, "off") == 0 ? 0 : strcmp(e, "regs") == 0 ? 1 : strcmp(e, "all") == 0 ? 2 : strcmp(e, "struct") == 0 ? 3 : 4;
}

void Prover::bindPublicInputs(const Hash &program_hash, uint64_t entry_pc, const std::vector<uint64_t> *initial_regs) {
    transcript_.reset();                                   // prover.zig:91
    transcript_.appendBytes(program_hash.data(), 32);      // :98-100
    transcript_.appendFieldElement(finit(entry_pc));       // :103
    if (initial_regs)
        for (uint64_t r : *initial_regs) transcript_.appendFieldElement(finit(r));  // :106-110
}

void Prover::generateSumcheckProof(Proof &proof, size_t num_steps, size_t num_vars) {  // :229-289
    transcript_.appendBytes("SUMCHECK_BEGIN");
    transcript_.appendFieldElement(finit(num_steps));
    transcript_.appendFieldElement(finit(num_vars));
    proof.constraint_proof.final_eval = 0;
    for (size_t round = 0; round < num_vars; round++) {
        for (int k = 0; k < 4; k++) {
            proof.constraint_proof.round_polynomials[round * 4 + k] = 0;  // zero polynomial, :267-272
            transcript_.appendFieldElement(0);
        }
        proof.constraint_proof.final_point[round] = transcript_.challenge();
    }
}

void Prover::generateLassoProofs(Proof &proof, size_t num_lookups) {  // :292-363
    transcript_.appendBytes("LASSO_BEGIN");
    // per lookup constraint i: appendBytes("LASSO_TABLE"); appendFieldElement(F.init(i))  (:311-312)
    transcript_.appendTaggedCounter("LASSO_TABLE", 0, num_lookups);
    // placeholder proofs (the rng fill loops run zero times): proof.lookup_placeholders was set by the
    // caller BEFORE the serialiser thread started reading it; it is not written again here (no concurrent store)
    (void)proof;
}

void Prover::generateCommitments(Proof &proof, const CommitSteps &gpu, size_t nv) {  // :366-467
    const size_t NC = ZIGZ_NUM_COLUMNS;
    const int world = shard_.world > 1 ? shard_.world : 1;
    size_t c0 = 0, c1 = NC;
    if (world > 1) columnBlock(NC, world, shard_.rank, c0, c1);  // the job holds this rank's columns [c0, c1)
    const size_t nloc = c1 - c0, nmax = (NC + (size_t)world - 1) / (size_t)world;
    auto exchange = [&](const std::vector<uint8_t> &send, size_t rec, std::vector<uint8_t> &all) {
        // send: nmax records of `rec` bytes (zero padded); all: the NC records in column order
        std::vector<uint8_t> recv((size_t)world * nmax * rec);
        if (!shard_.allgather || shard_.allgather(shard_.user, send.data(), nmax * rec, recv.data()) != 0)
            throw Error(ZIGZ_ERR_COMM, "sharded prove: the all-gather hook failed");
        all.resize(NC * rec);
        for (int r = 0; r < world; r++) {
            size_t a, b;
            columnBlock(NC, world, r, a, b);
            memcpy(all.data() + a * rec, recv.data() + (size_t)r * nmax * rec, (b - a) * rec);
        }
    };
    std::vector<uint8_t> roots(NC * 32);
    double t0 = now_s();
    if (world == 1) {
        gpu.roots(roots.data());                                       // PHASE 1 results
    } else {
        std::vector<uint8_t> mine(nmax * 32, 0);
        gpu.roots(mine.data());
        exchange(mine, 32, roots);                                     // exchange 1: 43 x 32 B
    }
    timings[3] = now_s() - t0;
    t0 = now_s();
    transcript_.appendBytes("POLY_COMMITMENTS");                       // PHASE 2, :413-416
    for (size_t c = 0; c < NC; c++) {
        memcpy(proof.witness_commitments[c].commitment.data(), roots.data() + 32 * c, 32);
        transcript_.appendBytes(roots.data() + 32 * c, 32);
    }
    std::vector<F> points(NC * nv + 1);                                // PHASE 3, :420-424
    for (size_t c = 0; c < NC; c++)
        for (size_t j = 0; j < nv; j++) points[c * nv + j] = transcript_.challenge();
    std::vector<F> values(NC), indices(NC), leaves(NC);
    std::vector<uint8_t> sib(NC * nv * 32 + 1), dirs(NC * nv + 1);
    timings[4] = now_s() - t0;
    t0 = now_s();
    gpu.open_all(points.data() + c0 * nv, values.data() + c0, indices.data() + c0, leaves.data() + c0, sib.data() + c0 * nv * 32,
                 dirs.data() + c0 * nv);                                                     // :427-431
    if (world > 1) {                                                   // exchange 2: value, index, leaf, path per column
        const size_t rec = 24 + 33 * nv;
        std::vector<uint8_t> mine(nmax * rec, 0), all;
        for (size_t k = 0; k < nloc; k++) {
            uint8_t *q = mine.data() + k * rec;
            const uint64_t head[3] = {values[c0 + k], indices[c0 + k], leaves[c0 + k]};
            memcpy(q, head, 24);
            memcpy(q + 24, sib.data() + (c0 + k) * nv * 32, nv * 32);
            memcpy(q + 24 + nv * 32, dirs.data() + (c0 + k) * nv, nv);
        }
        exchange(mine, rec, all);
        for (size_t c = 0; c < NC; c++) {
            const uint8_t *q = all.data() + c * rec;
            uint64_t head[3];
            memcpy(head, q, 24);
            values[c] = head[0]; indices[c] = head[1]; leaves[c] = head[2];
            memcpy(sib.data() + c * nv * 32, q + 24, nv * 32);
            memcpy(dirs.data() + c * nv, q + 24 + nv * 32, nv);
        }
    }
    timings[5] = now_s() - t0;
    t0 = now_s();
    for (size_t c = 0; c < NC; c++) {
        CommitmentOpening &o = proof.witness_commitments[c];
        o.point.assign(points.begin() + c * nv, points.begin() + (c + 1) * nv);
        o.value = values[c];
        o.proof.point = o.point;
        o.proof.value = values[c];
        o.proof.merkle_proof.index = indices[c];
        o.proof.merkle_proof.value = leaves[c];
        o.proof.merkle_proof.path.siblings.resize(nv);
        o.proof.merkle_proof.path.directions.resize(nv);
        for (size_t l = 0; l < nv; l++) {
            memcpy(o.proof.merkle_proof.path.siblings[l].data(), sib.data() + (c * nv + l) * 32, 32);
            o.proof.merkle_proof.path.directions[l] = dirs[c * nv + l];
        }
    }
    transcript_.appendBytes("OPENING_CLAIMS");                         // PHASE 4, :463-466
    for (size_t c = 0; c < NC; c++) transcript_.appendFieldElement(proof.witness_commitments[c].value);
    timings[6] = now_s() - t0;
}

Proof Prover::proveWitness(const PublicIO &io, size_t num_lookups, const Witness *witness, const uint32_t *d_cols,
                           size_t d_col_stride, size_t num_vars, const std::vector<uint64_t> *initial_regs) {
    return proveWitnessImpl(io, num_lookups, witness, d_cols, d_col_stride, num_vars, initial_regs, nullptr);
}

void Prover::proveWitnessToBytes(const PublicIO &io, size_t num_lookups, const Witness *witness, const uint32_t *d_cols,
                                 size_t d_col_stride, size_t num_vars, const std::vector<uint64_t> *initial_regs,
                                 std::vector<uint8_t> &out) {
    (void)proveWitnessImpl(io, num_lookups, witness, d_cols, d_col_stride, num_vars, initial_regs, &out);
}

void Prover::proveStepsToBytes(const PublicIO &io, size_t num_lookups, const TraceRecords &records, size_t num_vars,
                               const std::vector<uint64_t> *initial_regs, std::vector<uint8_t> &out) {
    if (!slots_) throw Error(ZIGZ_ERR_INVALID_ARGUMENT, "proveStepsToBytes: needs a prover of a service (GpuSlots)");
    if (!records) throw Error(ZIGZ_ERR_INVALID_ARGUMENT, "proveStepsToBytes: no trace records");
    (void)proveWitnessImpl(io, num_lookups, nullptr, nullptr, 0, num_vars, initial_regs, &out, records);
}

// ---------------------------------------------------------------- GpuSlots
GpuSlots::GpuSlots(int device, size_t k) {
    if (k == 0) throw Error(ZIGZ_ERR_INVALID_ARGUMENT, "GpuSlots: no slots");
    for (size_t i = 0; i < k; i++) {
        zigz_ctx *c = nullptr;
        const zigz_status st = zigz_ctx_create(device, &c);
        if (st != ZIGZ_OK) {
            for (zigz_ctx *x : all_) zigz_ctx_destroy(x);
            throw Error(st, std::string("GpuSlots: zigz_ctx_create failed: ") + zigz_status_name(st));
        }
        all_.push_back(c);
    }
    free_.assign(all_.rbegin(), all_.rend());  // (slot 0 is handed out first)
}
GpuSlots::~GpuSlots() {
    for (zigz_ctx *c : all_) zigz_ctx_destroy(c);
}
zigz_ctx *GpuSlots::acquire() {
    std::unique_lock<std::mutex> lk(m_);
    const uint64_t mine = next_ticket_++;
    cv_.wait(lk, [&] { return serving_ == mine && !free_.empty(); });
    zigz_ctx *c = free_.back();
    free_.pop_back();
    serving_++;
    if (!free_.empty() && serving_ != next_ticket_) cv_.notify_all();  // the next waiter may go at once
    return c;
}
void GpuSlots::release(zigz_ctx *ctx) {
    {
        std::lock_guard<std::mutex> lk(m_);
        free_.push_back(ctx);
    }
    cv_.notify_all();
}

// ---------------------------------------------------------------- GpuBatcher
struct GpuBatcher::Group {
    size_t nv = 0, stride = 0;
    uint64_t m_small = 0, m_run = 0, m_cons = 0;
    std::vector<const uint32_t *> cols;
    int phase = 0;  // 0 collecting, 1 building, 2 roots there, 3 opening, 4 done, -1 failed
    double deadline = 0;
    bool full = false;  // max_batch members: the driver need not linger
    std::array<uint64_t, 5> key{};
    std::condition_variable cv;
    zigz_ctx *ctx = nullptr;
    zigz_commit_job *job = nullptr;
    GpuSlots::Lease lease;
    int64_t sd0 = 0, ra0 = 0, cg0 = 0;  // the slot context's own hint masks, put back when the group is done
    std::vector<uint8_t> roots, sib, dirs;
    std::vector<F> points, values, indices, leaves;
    unsigned handed_in = 0;
    std::vector<uint8_t> present;  // members still taking part (abandon clears)
    std::exception_ptr err;
    zigz_kernel_stats stats{};
};

// begin + roots for a group (its driver's thread; the lock is NOT held).  The group stays open while the driver waits for a GPU
// slot: with a backlog in front of the GPU -- the only time sharing launches matters -- proofs that arrive meanwhile join, and
// the group is as large as the backlog allows without anybody having waited for it; it closes when the slot is there.
void GpuBatcher::build(const std::shared_ptr<Group> &g) {
    try {
        g->lease.slots = slots_;
        g->lease.ctx = g->ctx = slots_->acquire();
        {
            std::lock_guard<std::mutex> lk(m_);
            auto it = open_.find(g->key);
            if (it != open_.end() && it->second == g) open_.erase(it);
            g->phase = 1;
        }
        zigz_ctx *c = g->ctx;
        (void)zigz_ctx_get_option(c, "small_domain_mask", &g->sd0);
        (void)zigz_ctx_get_option(c, "run_aware_mask", &g->ra0);
        (void)zigz_ctx_get_option(c, "cons_group_mask", &g->cg0);
        check(c, zigz_ctx_set_option(c, "small_domain_mask", (int64_t)g->m_small));
        check(c, zigz_ctx_set_option(c, "run_aware_mask", (int64_t)g->m_run));
        check(c, zigz_ctx_set_option(c, "cons_group_mask", (int64_t)g->m_cons));
        (void)zigz_ctx_set_option(c, "batch_reserve", (int64_t)max_batch_);  // (workspaces for a full group from the start)
        const size_t k = g->cols.size();
        g->roots.resize(k * ZIGZ_NUM_COLUMNS * 32);
        check(c, zigz_commit_begin_batch(c, g->cols.data(), k, ZIGZ_NUM_COLUMNS, g->stride, g->nv, &g->job));
        check(c, zigz_commit_roots(g->job, g->roots.data()));
        g->points.assign(k * ZIGZ_NUM_COLUMNS * g->nv + 1, 0);
    } catch (...) {
        g->err = std::current_exception();
    }
    {
        std::lock_guard<std::mutex> lk(m_);
        auto it = open_.find(g->key);  // (no slot: the group never closed)
        if (it != open_.end() && it->second == g) open_.erase(it);
        g->phase = g->err ? -1 : 2;
        if (g->err) finish(g);
    }
    g->cv.notify_all();
}
// gives the job, the masks and the slot back (lock held or not: touches only the group)
void GpuBatcher::finish(const std::shared_ptr<Group> &g) {
    if (g->job) zigz_commit_end(g->job);
    g->job = nullptr;
    if (g->ctx) {
        (void)zigz_ctx_set_option(g->ctx, "small_domain_mask", g->sd0);
        (void)zigz_ctx_set_option(g->ctx, "run_aware_mask", g->ra0);
        (void)zigz_ctx_set_option(g->ctx, "cons_group_mask", g->cg0);
    }
    g->ctx = nullptr;
    g->lease.drop();
}

// open_all + end for a group whose members have all handed in their points (or given up); the lock is NOT held
void GpuBatcher::run_open_all(const std::shared_ptr<Group> &g) {
    const size_t nc = ZIGZ_NUM_COLUMNS, nv = g->nv, k = g->cols.size();
    try {
        g->values.resize(k * nc);
        g->indices.resize(k * nc);
        g->leaves.resize(k * nc);
        g->sib.resize(k * nc * nv * 32 + 1);
        g->dirs.resize(k * nc * nv + 1);
        check(g->ctx, zigz_commit_open_all(g->job, g->points.data(), g->values.data(), g->indices.data(), g->leaves.data(),
                                           g->sib.data(), g->dirs.data()));
        (void)zigz_ctx_get_stats(g->ctx, &g->stats);
    } catch (...) {
        g->err = std::current_exception();
    }
    finish(g);
    {
        std::lock_guard<std::mutex> lk(m_);
        g->phase = g->err ? -1 : 4;
    }
    g->cv.notify_all();
}

std::shared_ptr<GpuBatcher::Group> GpuBatcher::join(const uint32_t *d_cols, size_t stride, size_t nv, uint64_t m_small, uint64_t m_run,
                                                     uint64_t m_cons, unsigned *idx, uint8_t *roots43) {
    const std::array<uint64_t, 5> key{(uint64_t)nv, (uint64_t)stride, m_small, m_run, m_cons};
    std::shared_ptr<Group> g;
    bool run_build = false;  // the member that opened the group drives it: lingers, takes the slot, builds
    {
        std::unique_lock<std::mutex> lk(m_);
        auto it = open_.find(key);
        if (it == open_.end()) {
            g = std::make_shared<Group>();
            g->nv = nv;
            g->stride = stride;
            g->m_small = m_small;
            g->m_run = m_run;
            g->m_cons = m_cons;
            g->key = key;
            g->deadline = now_s() + linger_s_;
            open_[key] = g;
            run_build = true;
        } else {
            g = it->second;
        }
        *idx = (unsigned)g->cols.size();
        g->cols.push_back(d_cols);
        g->present.push_back(1);
        if (g->cols.size() >= max_batch_) {  // full: the next arrival opens a new group
            auto it2 = open_.find(key);
            if (it2 != open_.end() && it2->second == g) open_.erase(it2);
            g->full = true;
            g->cv.notify_all();
        }
        if (run_build) {
            while (!g->full) {
                const double left = g->deadline - now_s();
                if (left <= 0) break;  // nobody else came in time
                g->cv.wait_for(lk, std::chrono::duration<double>(left));
            }
        }
    }
    if (run_build) build(g);
    {
        std::unique_lock<std::mutex> lk(m_);
        g->cv.wait(lk, [&] { return g->phase >= 2 || g->phase < 0; });
        if (g->phase < 0) std::rethrow_exception(g->err);
    }
    memcpy(roots43, g->roots.data() + (size_t)*idx * ZIGZ_NUM_COLUMNS * 32, ZIGZ_NUM_COLUMNS * 32);
    return g;
}

void GpuBatcher::open(const std::shared_ptr<Group> &g, unsigned idx, const F *points, F *values, F *indices, F *leaves,
                      uint8_t *sib, uint8_t *dirs, zigz_kernel_stats *stats) {
    const size_t nc = ZIGZ_NUM_COLUMNS, nv = g->nv, k = g->cols.size();
    bool run_open = false;
    {
        std::unique_lock<std::mutex> lk(m_);
        if (g->phase < 0) std::rethrow_exception(g->err);
        memcpy(g->points.data() + (size_t)idx * nc * nv, points, nc * nv * sizeof(F));
        if (++g->handed_in == k) {
            g->phase = 3;
            run_open = true;
        }
    }
    if (run_open) run_open_all(g);
    {
        std::unique_lock<std::mutex> lk(m_);
        g->cv.wait(lk, [&] { return g->phase == 4 || g->phase < 0; });
        if (g->phase < 0) std::rethrow_exception(g->err);
    }
    memcpy(values, g->values.data() + (size_t)idx * nc, nc * sizeof(F));
    memcpy(indices, g->indices.data() + (size_t)idx * nc, nc * sizeof(F));
    memcpy(leaves, g->leaves.data() + (size_t)idx * nc, nc * sizeof(F));
    if (nv) {
        memcpy(sib, g->sib.data() + (size_t)idx * nc * nv * 32, nc * nv * 32);
        memcpy(dirs, g->dirs.data() + (size_t)idx * nc * nv, nc * nv);
    }
    if (stats) {  // the group's statistics, this proof's share of the counts
        *stats = g->stats;
        stats->keccak_permutations /= k;
        stats->list_hash_perms /= k;
        stats->run_aware_hashed /= k;
        stats->cons_hashed /= k;
    }
}

void GpuBatcher::abandon(const std::shared_ptr<Group> &g, unsigned idx) {
    // the member's points stay zero (valid field elements): the group completes, the abandoned results are nobody's
    bool run_open = false;
    {
        std::unique_lock<std::mutex> lk(m_);
        if (idx >= g->present.size() || !g->present[idx] || g->phase < 2 || g->phase >= 3) return;
        g->present[idx] = 0;
        if (++g->handed_in == g->cols.size()) {
            g->phase = 3;
            run_open = true;
        }
    }
    if (run_open) run_open_all(g);  // (the last one: the members that did hand in their points are waiting for their openings)
}

Proof Prover::proveWitnessImpl(const PublicIO &io, size_t num_lookups, const Witness *witness, const uint32_t *d_cols,
                               size_t d_col_stride, size_t num_vars, const std::vector<uint64_t> *initial_regs,
                               std::vector<uint8_t> *bytes_out, const TraceRecords &records) {
    const bool steps = (bool)records;
    // transcript binding of the public inputs (prover.zig:91-110)
    bindPublicInputs(io.program_hash, io.initial_pc, initial_regs);
    const size_t num_steps = io.num_steps;
    if (num_steps == 0) throw Error(ZIGZ_ERR_EMPTY_TRACE, "error.EmptyTrace");
    Proof proof = Proof::init(num_steps);
    if (proof.metadata.num_vars != num_vars) throw Error(ZIGZ_ERR_INVALID_ARGUMENT, "witness num_vars != log2_int_ceil(num_steps)");
    for (double &t : timings) t = 0;
    // RAII, in the order the destructors must run: the job is ended (and the context's active-job slot released) and the hint
    // masks are put back BEFORE a borrowed context goes back to its pool -- on every exit path, including a bad_alloc between
    // begin and the transcript work
    GpuSlots::Lease lease;
    // the hint masks are options of the CONTEXT: remember what its owner had set and put that back when this proof is done
    // (on every exit path), whatever this proof sets below
    struct MaskRestore {
        zigz_ctx *ctx = nullptr;
        int64_t sd = 0, ra = 0, cg = 0;
        void arm(zigz_ctx *c) {
            ctx = c;
            (void)zigz_ctx_get_option(ctx, "small_domain_mask", &sd);
            (void)zigz_ctx_get_option(ctx, "run_aware_mask", &ra);
            (void)zigz_ctx_get_option(ctx, "cons_group_mask", &cg);
        }
        void restore() {
            if (!ctx) return;
            (void)zigz_ctx_set_option(ctx, "small_domain_mask", sd);
            (void)zigz_ctx_set_option(ctx, "run_aware_mask", ra);
            (void)zigz_ctx_set_option(ctx, "cons_group_mask", cg);
            ctx = nullptr;
        }
        ~MaskRestore() { restore(); }
    } mask_restore;
    struct JobGuard {
        zigz_commit_job *job = nullptr;
        ~JobGuard() { if (job) zigz_commit_end(job); }
    } guard;
    size_t c0 = 0, c1 = ZIGZ_NUM_COLUMNS;  // sharded: this rank commits its block of columns only
    if (shard_.world > 1) {
        if (shard_.rank < 0 || shard_.rank >= shard_.world || shard_.world > (int)ZIGZ_NUM_COLUMNS)
            throw Error(ZIGZ_ERR_INVALID_ARGUMENT, "sharded prove: bad rank / world");
        columnBlock(ZIGZ_NUM_COLUMNS, shard_.world, shard_.rank, c0, c1);
    }
    // [6/6 phase 1] Merkle builds do not depend on the transcript (prover.zig:405-416).  A prover with a context of its own
    // starts them first, asynchronously on the GPU, so that they run underneath its sequential host absorption of steps 4 and
    // 5; a prover of a service takes a GPU slot only after step 5 and starts them then (GpuSlots, zigz_host.hpp).
    // the hint masks of this proof's columns (the comments at begin_job say what they are)
    uint64_t m_small = 0, m_run = 0, m_cons = 0;
    {
        const uint64_t structural = (1ull << 1) | (0x3full << 33) | (1ull << 42), sel = (1ull << (c1 - c0)) - 1;
        if (small_domain_tables) m_small = (structural >> c0) & sel;
        if (run_aware) {
            const uint64_t small = small_domain_tables ? structural : 0;
            const uint64_t regs = 0x7fffffffull << 2, mem = 3ull << 40;
            const uint64_t hinted = run_aware == 2 ? ((1ull << ZIGZ_NUM_COLUMNS) - 1) & ~small : run_aware >= 3 ? regs | mem : regs;
            if (run_aware == 4) m_cons = ((1ull | (1ull << 1) | (0x7full << 33) | (1ull << 42)) >> c0) & sel;
            m_run = (hinted >> c0) & sel;
        }
    }
    auto begin_job = [&]() {
        double t0 = now_s();
        mask_restore.arm(ctx_);
        // Columns that hold values < 128 by construction (x0; opcode, rd, rs1, rs2, funct3, funct7; mem.is_read --
        // witness.zig:164-169,239, registers.zig:38-48): their leaf and level-1 digests come from constant tables
        // (zigz_hip.h, option "small_domain_mask"; checked on the device, identical trees for any input).
        if (small_domain_tables) check(ctx_, zigz_ctx_set_option(ctx_, "small_domain_mask", (int64_t)m_small));
        // The register columns x1..x31 are piecewise constant by construction: a step writes at most one register
        // (VMState.writeReg, src/vm/state.zig), so together they change at most once per step.  Their large Merkle levels are
        // built run-aware (zigz_hip.h, option "run_aware_mask": a node that repeats its left neighbour is copied, decided from
        // the values on the device; identical trees for any input).  So are mem.address and mem.value between memory accesses:
        // a step that is not a LOAD / STORE records 0 in both (witness.zig:236-253), so each of them changes at most twice per
        // memory access.  run_aware: 1 = the registers, 3 = registers + the two memory columns, 2 = every column that is not
        // small-domain, 4 (default) = 3 + the ten columns that are functions of the instruction at pc -- pc, x0, opcode, rd, rs1,
        // rs2, funct3, funct7, imm, is_read -- as a content-addressed group (zigz_hip.h, option "cons_group_mask": wherever the
        // program loops the same nodes recur in all ten; probed first, dropped for a trace that does not repeat).
        if (run_aware) {
            if (run_aware == 4) check(ctx_, zigz_ctx_set_option(ctx_, "cons_group_mask", (int64_t)m_cons));
            check(ctx_, zigz_ctx_set_option(ctx_, "run_aware_mask", (int64_t)m_run));
        }
        if (steps) {  // the witness from the compact trace, inside the slot: upload + expansion + builds on one stream
            const uint32_t *wc = nullptr;
            size_t ws = 0;
            if (records.s16)
                check(ctx_, zigz_dev_witness_from_steps16_ws(ctx_, records.s16, num_steps, records.mem, records.nmem, records.code_base,
                                                             records.code, records.ncode, num_vars, records.regs_before, &wc, &ws));
            else if (records.s32)
                check(ctx_, zigz_dev_witness_from_steps32_ws(ctx_, records.s32, num_steps, records.mem, records.nmem, num_vars,
                                                             records.regs_before, &wc, &ws));
            else
                check(ctx_, zigz_dev_witness_from_steps_ws(ctx_, records.s48, num_steps, num_vars, records.regs_before, &wc, &ws));
            check(ctx_, zigz_commit_begin_dev(ctx_, wc + c0 * ws, c1 - c0, ws, num_vars, &guard.job));
        } else if (witness) {
            check(ctx_, zigz_commit_begin(ctx_, witness->columns.data() + c0 * ((size_t)1 << num_vars), c1 - c0,
                                          (size_t)1 << num_vars, num_vars, &guard.job));
        } else {
            check(ctx_, zigz_commit_begin_dev(ctx_, d_cols + c0 * d_col_stride, c1 - c0, d_col_stride, num_vars, &guard.job));
        }
        timings[0] = now_s() - t0;
    };
    if (!slots_) begin_job();
    // packagePublicIO (:514-559) only copies VM results; doing it here lets the serialiser start early
    proof.public_io = io;
    if (initial_regs) proof.public_io.initial_regs = *initial_regs;
    else proof.public_io.initial_regs.reset();
    proof.lookup_placeholders = num_lookups;  // final before the writer thread exists
    bool writing = false;
    double t_slot = 0;
    try {
        double t0 = now_s();
        generateSumcheckProof(proof, num_steps, num_vars);  // [4/6]
        timings[1] = now_s() - t0;
        if (bytes_out) {
            // everything up to the 43 openings is final now: write it while the transcript absorbs step 5
            size_t total = BinarySerializer::prefixSize(proof) + ZIGZ_NUM_COLUMNS * (68 + 41 * num_vars);
            if (bytes_out->size() != total) bytes_out->resize(total);
            uint8_t *buf = bytes_out->data();
            const Proof *pp = &proof;
            t_helper.run([pp, buf] { BinarySerializer::writePrefix(*pp, buf); });
            writing = true;
        }
        t0 = now_s();
        generateLassoProofs(proof, num_lookups);            // [5/6]
        timings[2] = now_s() - t0;
        // small traces of a service: this proof's GPU steps as a member of a batch (GpuBatcher) -- resident columns, the whole
        // witness, a size at which a proof's launches are mostly latency
        const bool batched = slots_ && batcher_ && d_cols && !witness && !steps && shard_.world <= 1 && num_vars <= batcher_->maxNv() &&
                             (num_vars < 15 || (((m_run | m_cons) & ((1ull << ZIGZ_NUM_COLUMNS) - 1)) == (1ull << ZIGZ_NUM_COLUMNS) - 1));
        CommitSteps gpu;
        std::shared_ptr<GpuBatcher::Group> grp;
        unsigned gidx = 0;
        if (batched) {
            gpu.roots = [&](uint8_t *roots43) {
                const double tb = now_s();
                grp = batcher_->join(d_cols, d_col_stride, num_vars, m_small, m_run, m_cons, &gidx, roots43);
                timings[8] = now_s() - tb;
            };
            gpu.open_all = [&](const F *pts, F *values, F *indices, F *leaves, uint8_t *sib, uint8_t *dirs) {
                batcher_->open(grp, gidx, pts, values, indices, leaves, sib, dirs, &last_stats);
            };
        } else {
            if (slots_) {
                t0 = now_s();
                lease.slots = slots_;
                lease.ctx = ctx_ = slots_->acquire();
                t_slot = now_s();
                timings[8] = t_slot - t0;
                begin_job();
            }
            gpu.roots = [&](uint8_t *roots43) { check(ctx_, zigz_commit_roots(guard.job, roots43)); };
            gpu.open_all = [&](const F *pts, F *values, F *indices, F *leaves, uint8_t *sib, uint8_t *dirs) {
                check(ctx_, zigz_commit_open_all(guard.job, pts, values, indices, leaves, sib, dirs));
            };
        }
        try {
            generateCommitments(proof, gpu, num_vars);      // [6/6]
        } catch (...) {
            if (grp) batcher_->abandon(grp, gidx);  // (the other members of the batch must not wait for this proof's points)
            throw;
        }
        if (batched) {
            last_log.clear();
            timings[9] = timings[3] + timings[4] + timings[5];
            ctx_ = nullptr;
        }
    } catch (...) {
        if (writing) try { t_helper.wait(); } catch (...) {}  // (the helper reads `proof`: it must be done before the unwind)
        if (slots_) ctx_ = nullptr;  // (the guards release the job, the masks and the slot, in that order)
        throw;
    }
    if (guard.job) zigz_commit_end(guard.job);
    guard.job = nullptr;
    if (slots_ && lease.ctx) {  // what the caller may want to know about the context this proof ran on, before somebody else has it
        (void)zigz_ctx_get_stats(ctx_, &last_stats);
        size_t n = 0;
        last_log.resize(80);
        if (zigz_ctx_launch_log(ctx_, last_log.data(), last_log.size(), &n) != ZIGZ_OK) n = 0;
        last_log.resize(n < last_log.size() ? n : last_log.size());
        mask_restore.restore();
        lease.drop();
        ctx_ = nullptr;
        timings[9] = now_s() - t_slot;
    }
    if (bytes_out) {
        double t0 = now_s();
        t_helper.wait();
        BinarySerializer::writeCommitments(proof, bytes_out->data() + BinarySerializer::prefixSize(proof));
        timings[7] = now_s() - t0;
    }
    return proof;
}

Proof Prover::prove(const std::vector<uint8_t> &program, uint64_t entry_pc, const std::vector<uint64_t> *initial_regs,
                    size_t max_steps, const std::vector<Segment> *segments, const std::vector<uint64_t> *input,
                    std::vector<uint8_t> *serialized) {
    if (verbose) fprintf(stderr, "\n=== zkVM Prover ===\nProgram size: %zu bytes\nEntry PC: 0x%llx\n", program.size(),
                         (unsigned long long)entry_pc);
    PublicIO io;
    zigz_sha256(program.data(), program.size(), io.program_hash.data());
    io.initial_pc = entry_pc;
    // [1/6] execute (prover.zig:117-149)
    std::unique_ptr<VMState> vm(segments ? new VMState(*segments, entry_pc, input) : new VMState(program, entry_pc, input));
    if (initial_regs)
        for (size_t i = 0; i < initial_regs->size() && i < 32; i++) vm->writeReg((unsigned)i, (*initial_regs)[i]);
    static thread_local std::vector<zigz_trace_step> step_pool;  // recycled trace storage (avoids refaulting 48 B per step)
    vm->trace.steps.swap(step_pool);
    struct Recycle {
        std::vector<zigz_trace_step> &pool, &steps;
        ~Recycle() { if (steps.capacity() > pool.capacity()) pool.swap(steps); }
    } recycle{step_pool, vm->trace.steps};
    vm->trace.reserveSteps(max_steps < ((size_t)1 << 22) ? max_steps : ((size_t)1 << 22));
    size_t step_count = 0;
    while (!vm->halted && step_count < max_steps) {
        vm->step();
        if (vm->invalid_instruction) break;  // "Program halted at step N": normal termination
        step_count++;
    }
    const size_t num_steps = vm->trace.stepCount();
    if (num_steps == 0) throw Error(ZIGZ_ERR_EMPTY_TRACE, "error.EmptyTrace");
    // [2/6] witness (prover.zig:156-162): built directly in HBM from the compact trace records (K8); the host
    // WitnessGenerator::generate stays available for callers that want the columns on the host
    const size_t nv = log2_int_ceil(num_steps), N = (size_t)1 << nv, stride = N < 4 ? 4 : N;
    struct DevCols {
        zigz_ctx *ctx;
        void *p = nullptr;
        ~DevCols() { if (p) zigz_dev_free(ctx, p); }
    } dcols{ctx_};
    if (!slots_) {  // (a prover of a service builds the witness inside its GPU slot, from the records: proveWitnessImpl)
        check(ctx_, zigz_dev_alloc(ctx_, ROW_WORDS * stride * sizeof(uint32_t), &dcols.p));
        check(ctx_, zigz_dev_witness_from_steps(ctx_, vm->trace.steps.data(), num_steps, nv, vm->trace.initial_regs, (uint32_t *)dcols.p,
                                                stride));
    }
    // [3/6] constraint system: only the number of lookup constraints is observable (builder.zig:253-267)
    size_t L = 0;
    for (uint8_t f : vm->trace.is_lookup) L += f;
    io.final_pc = vm->pc;
    std::vector<uint64_t> fr(32);
    for (unsigned r = 0; r < 32; r++) fr[r] = vm->readReg(r);
    io.final_regs = fr;
    io.num_steps = num_steps;
    if (!vm->output_tape.empty()) io.outputs = vm->output_tape;
    if (slots_) {
        TraceRecords rec;
        rec.s48 = vm->trace.steps.data();
        rec.regs_before = vm->trace.initial_regs;
        return proveWitnessImpl(io, L, nullptr, nullptr, 0, nv, initial_regs, serialized, rec);
    }
    return proveWitnessImpl(io, L, nullptr, (const uint32_t *)dcols.p, stride, nv, initial_regs, serialized);
}

// ---------------------------------------------------------------- BinarySerializer (serialization.zig)
namespace {
struct R {
    const uint8_t *b;
    size_t len, pos = 0;
    const uint8_t *bytes(size_t n) {
        if (len - pos < n) throw Error(ERR_INVALID_DATA, "error.EndOfStream");
        const uint8_t *q = b + pos;
        pos += n;
        return q;
    }
    uint32_t u32() { uint32_t v; memcpy(&v, bytes(4), 4); return v; }
    uint64_t u64() { uint64_t v; memcpy(&v, bytes(8), 8); return v; }
};
size_t sumcheck_size(const ProverSumcheckProof &s) { return (s.round_polynomials.size() + s.final_point.size() + 1) * 8; }
}  // namespace

namespace {
struct RawW {  // unchecked writer into a caller-sized buffer
    uint8_t *b;
    size_t pos = 0;
    void bytes(const void *d, size_t n) { memcpy(b + pos, d, n); pos += n; }
    void u8(uint8_t v) { b[pos++] = v; }
    void u32(uint32_t v) { bytes(&v, 4); }
    void u64(uint64_t v) { bytes(&v, 8); }
};
void write_sumcheck(RawW &w, const ProverSumcheckProof &s) {  // writeConstraintProof, :296-311
    for (F c : s.round_polynomials) w.u64(c);
    for (F c : s.final_point) w.u64(c);
    w.u64(s.final_eval);
}
}  // namespace

size_t BinarySerializer::prefixSize(const Proof &p) {
    size_t size = 32;  // header
    size += 32 + 8 + 8 + 4 + 4 + 8 + 4;
    if (p.public_io.initial_regs) size += p.public_io.initial_regs->size() * 8;
    if (p.public_io.final_regs) size += p.public_io.final_regs->size() * 8;
    if (p.public_io.outputs) size += p.public_io.outputs->size() * 8;
    size += sumcheck_size(p.constraint_proof);
    size += 4 + p.lookup_placeholders * 24;
    for (auto &l : p.lookup_proofs) size += 16 + sumcheck_size(l.multiset_proof);
    return size;
}

size_t BinarySerializer::exactSize(const Proof &p) {
    size_t size = prefixSize(p);
    for (auto &o : p.witness_commitments)
        size += 32 + o.point.size() * 8 + 8 + (8 + 8 + 8 + 4 + o.proof.merkle_proof.path.siblings.size() * 33);
    return size;
}

// The L placeholder records of writeLassoProofs (serialization.zig:333-344) -- 24 bytes each, 25 MB at a 2^20 trace -- are
// the bulk of a proof and are written once, never read back by the prover: they go out with non-temporal 8-byte stores
// (movnti: no read-for-ownership of the destination lines, no cache pollution next to the sponge servers), which cuts the
// host CPU time of a proof's serialisation from ~3.4 ms to under 1 ms.  A record is three 8-byte words; the destination is
// 4-byte aligned by the layout of the prefix, so depending on its phase a word holds (id, 1) | 0 | 0 or 1 | 0 | (0, next id).
static void write_placeholders(uint8_t *q, size_t n) {
#if defined(__x86_64__)
    if (n >= 16 && ((uintptr_t)q & 3) == 0) {
        size_t i = 0;
        if ((uintptr_t)q & 4) {  // phase 4: the first id on its own, then words that straddle two records
            const uint32_t id0 = 0;
            memcpy(q, &id0, 4);
            long long *w = reinterpret_cast<long long *>(q + 4);
            for (; i + 1 < n; i++, w += 3) {
                _mm_stream_si64(w + 0, 1ll);                                   // num_lookups = 1
                _mm_stream_si64(w + 1, 0ll);                                   // num_vars = 0 | final_eval low
                _mm_stream_si64(w + 2, (long long)((uint64_t)(uint32_t)(i + 1) << 32));  // final_eval high | next table_id
            }
            _mm_sfence();
            uint8_t tail[20] = {0};  // the last record's remaining 20 bytes
            tail[0] = 1;
            memcpy(q + 24 * i + 4, tail, 20);
            return;
        }
        long long *w = reinterpret_cast<long long *>(q);
        for (; i < n; i++, w += 3) {
            _mm_stream_si64(w + 0, (long long)((uint64_t)(uint32_t)i | (1ull << 32)));  // table_id = i | num_lookups low = 1
            _mm_stream_si64(w + 1, 0ll);
            _mm_stream_si64(w + 2, 0ll);
        }
        _mm_sfence();
        return;
    }
#endif
    uint8_t rec[24] = {0};
    rec[4] = 1;
    for (size_t i = 0; i < n; i++, q += 24) {
        memcpy(q, rec, 24);
        const uint32_t id = (uint32_t)i;
        memcpy(q, &id, 4);
    }
}

void BinarySerializer::writePrefix(const Proof &p, uint8_t *buf) {
    RawW w{buf};
    w.bytes("ZIGZ", 4);  // writeHeader, :175-182
    w.u32(1);
    w.u64(p.metadata.field_modulus);
    w.u64(p.metadata.num_steps);
    w.u32((uint32_t)p.metadata.num_vars);
    w.u32(0);
    w.bytes(p.public_io.program_hash.data(), 32);  // writePublicIO, :209-245
    w.u64(p.public_io.initial_pc);
    w.u64(p.public_io.final_pc);
    auto regs = [&](const std::optional<std::vector<uint64_t>> &r) {
        if (r) { w.u32((uint32_t)r->size()); for (uint64_t v : *r) w.u64(v); }
        else w.u32(0);
    };
    regs(p.public_io.initial_regs);
    regs(p.public_io.final_regs);
    w.u64(p.public_io.num_steps);
    regs(p.public_io.outputs);
    write_sumcheck(w, p.constraint_proof);
    w.u32((uint32_t)p.lookupCount());  // writeLassoProofs, :333-344
    {   // placeholders: u32 table_id = i, u64 num_lookups = 1, u32 num_vars = 0, u64 final_eval = 0
        write_placeholders(buf + w.pos, p.lookup_placeholders);
        w.pos += p.lookup_placeholders * 24;
    }
    for (auto &l : p.lookup_proofs) {
        w.u32(l.table_id); w.u64(l.num_lookups); w.u32((uint32_t)l.multiset_proof.num_vars);
        write_sumcheck(w, l.multiset_proof);
    }
}

void BinarySerializer::writeCommitments(const Proof &p, uint8_t *buf) {
    RawW w{buf};
    for (auto &o : p.witness_commitments) {  // writeWitnessCommitments + writeMerkleProof, :374-429
        w.bytes(o.commitment.data(), 32);
        for (F c : o.point) w.u64(c);
        w.u64(o.value);
        w.u64(o.proof.value);
        w.u64(o.proof.merkle_proof.index);
        w.u64(o.proof.merkle_proof.value);
        w.u32((uint32_t)o.proof.merkle_proof.path.siblings.size());
        for (auto &sib : o.proof.merkle_proof.path.siblings) w.bytes(sib.data(), 32);
        for (uint8_t d : o.proof.merkle_proof.path.directions) w.u8(d ? 1 : 0);
    }
}

std::vector<uint8_t> BinarySerializer::serialize(const Proof &p) {
    // The reference sizes a fixed buffer from an estimate that under-counts; this
    // writes the same layout into an exact-size buffer.
    std::vector<uint8_t> buf(exactSize(p));
    writePrefix(p, buf.data());
    writeCommitments(p, buf.data() + prefixSize(p));
    return buf;
}

Proof BinarySerializer::deserialize(const uint8_t *data, size_t len) {  // :100-131
    R r{data, len};
    if (memcmp(r.bytes(4), "ZIGZ", 4) != 0) throw Error(ERR_INVALID_MAGIC, "error.InvalidMagicNumber");
    if (r.u32() != 1) throw Error(ERR_UNSUPPORTED_VERSION, "error.UnsupportedVersion");
    ProofMetadata md;
    md.field_modulus = r.u64();
    md.num_steps = (size_t)r.u64();
    md.num_vars = r.u32();
    (void)r.u32();
    if (md.field_modulus != BABYBEAR) throw Error(ERR_FIELD_MISMATCH, "error.FieldMismatch");
    if (md.num_steps == 0) throw Error(ERR_INVALID_DATA, "error.InvalidData");
    Proof p = Proof::init(md.num_steps);
    p.metadata = md;
    memcpy(p.public_io.program_hash.data(), r.bytes(32), 32);  // readPublicIO, :247-294
    p.public_io.initial_pc = r.u64();
    p.public_io.final_pc = r.u64();
    auto regs = [&](std::optional<std::vector<uint64_t>> &out) {
        uint32_t n = r.u32();
        if (n > 0) {
            if ((r.len - r.pos) / 8 < n) throw Error(ERR_INVALID_DATA, "error.EndOfStream");
            std::vector<uint64_t> v(n);
            for (auto &x : v) x = r.u64();
            out = std::move(v);
        } else out.reset();
    };
    regs(p.public_io.initial_regs);
    regs(p.public_io.final_regs);
    p.public_io.num_steps = (size_t)r.u64();
    regs(p.public_io.outputs);
    auto read_sumcheck = [&](ProverSumcheckProof &s) {  // readConstraintProof, :313-331
        for (auto &c : s.round_polynomials) c = finit(r.u64());
        for (auto &c : s.final_point) c = finit(r.u64());
        s.final_eval = finit(r.u64());
    };
    read_sumcheck(p.constraint_proof);
    uint32_t n_lasso = r.u32();  // readLassoProofs, :346-372
    bool placeholders_only = true;
    for (uint32_t i = 0; i < n_lasso; i++) {
        LassoProof l;
        l.table_id = r.u32();
        l.num_lookups = (size_t)r.u64();
        uint32_t nv = r.u32();
        if ((r.len - r.pos) / 8 < (size_t)nv * 4) throw Error(ERR_INVALID_DATA, "error.EndOfStream");
        l.multiset_proof.num_vars = nv;
        l.multiset_proof.num_coeffs = 3;
        l.multiset_proof.round_polynomials.assign((size_t)nv * 3, 0);
        l.multiset_proof.final_point.assign(nv, 0);
        read_sumcheck(l.multiset_proof);
        const bool is_placeholder = l.table_id == i && l.num_lookups == 1 && nv == 0 && l.multiset_proof.final_eval == 0;
        if (placeholders_only && is_placeholder) p.lookup_placeholders++;
        else { placeholders_only = false; p.lookup_proofs.push_back(std::move(l)); }
    }
    for (auto &o : p.witness_commitments) {  // readWitnessCommitments + readMerkleProof, :389-477
        memcpy(o.commitment.data(), r.bytes(32), 32);
        for (auto &c : o.point) c = finit(r.u64());
        o.value = finit(r.u64());
        o.proof.value = finit(r.u64());
        o.proof.merkle_proof.index = (size_t)r.u64();
        o.proof.merkle_proof.value = finit(r.u64());
        uint32_t plen = r.u32();
        if ((r.len - r.pos) / 33 < plen) throw Error(ERR_INVALID_DATA, "error.EndOfStream");
        o.proof.merkle_proof.path.siblings.resize(plen);
        o.proof.merkle_proof.path.directions.resize(plen);
        for (auto &s : o.proof.merkle_proof.path.siblings) memcpy(s.data(), r.bytes(32), 32);
        for (auto &d : o.proof.merkle_proof.path.directions) d = *r.bytes(1) != 0;
        o.proof.point = o.point;
    }
    return p;
}

// ---------------------------------------------------------------- Verifier (verifier.zig)
static inline F vadd(F a, F b) { F s = a + b; return s >= BABYBEAR ? s - BABYBEAR : s; }
static inline F vmul(F a, F b) { return (F)(((unsigned __int128)a * b) % BABYBEAR); }

VerificationResult Verifier::verifySumcheckProof(const ProverSumcheckProof &p) {  // :182-238
    transcript_.appendBytes("SUMCHECK_BEGIN");
    transcript_.appendFieldElement(finit(p.num_vars));
    const F claimed_sum = p.final_eval;
    const size_t nc = p.num_coeffs;
    for (size_t round = 0; round < p.num_vars; round++) {
        const F *c = p.round_polynomials.data() + round * nc;
        F g1 = 0;
        for (size_t k = 0; k < nc; k++) g1 = vadd(g1, c[k]);
        if (round == 0 && vadd(c[0], g1) != claimed_sum) return VerificationResult::RejectInvalidSumcheck;
        const F ch = transcript_.challenge();
        F ev = 0, pw = 1;
        for (size_t k = 0; k < nc; k++) { ev = vadd(ev, vmul(c[k], pw)); pw = vmul(pw, ch); }
        transcript_.appendFieldElement(ev);
    }
    return VerificationResult::Accept;
}

VerificationResult Verifier::verify(const Proof &proof, const std::vector<uint8_t> &program) {  // :49-91
    transcript_.reset();
    Hash ph;
    zigz_sha256(program.data(), program.size(), ph.data());  // bindPublicInputs, :95-122
    if (ph != proof.public_io.program_hash) throw Error(ERR_PROGRAM_HASH_MISMATCH, "error.ProgramHashMismatch");
    transcript_.appendBytes(ph.data(), 32);
    transcript_.appendFieldElement(finit(proof.public_io.initial_pc));
    if (proof.public_io.initial_regs)
        for (uint64_t r : *proof.public_io.initial_regs) transcript_.appendFieldElement(finit(r));
    transcript_.appendBytes("POLY_COMMITMENTS");  // bindPolynomialCommitments, :126-137
    for (auto &o : proof.witness_commitments) transcript_.appendBytes(o.commitment.data(), 32);
    for (auto &o : proof.witness_commitments)      // deriveAndBindOpeningClaims, :146-179
        for (size_t j = 0; j < o.point.size(); j++) (void)transcript_.challenge();
    transcript_.appendBytes("OPENING_CLAIMS");
    for (auto &o : proof.witness_commitments) transcript_.appendFieldElement(o.value);
    if (verifySumcheckProof(proof.constraint_proof) != VerificationResult::Accept)  // PHASE 4
        return VerificationResult::RejectInvalidSumcheck;
    for (size_t i = 0; i < proof.lookupCount(); i++) {  // PHASE 5, verifyLassoProof :241-267
        LassoProof l = proof.lookupAt(i);
        transcript_.appendBytes("LASSO_BEGIN");
        transcript_.appendBytes("LASSO_TABLE");
        transcript_.appendFieldElement(finit(l.table_id));
        if (verifySumcheckProof(l.multiset_proof) != VerificationResult::Accept) return VerificationResult::RejectInvalidLookup;
    }
    for (auto &o : proof.witness_commitments) {  // PHASE 6, verifyOpening :270-294
        if (o.value != o.proof.value) return VerificationResult::RejectInvalidCommitment;
        if (!CommitmentScheme::verify(o.commitment, o.point.size(), o.proof)) return VerificationResult::RejectInvalidCommitment;
    }
    return VerificationResult::Accept;
}

}  // namespace zigz
