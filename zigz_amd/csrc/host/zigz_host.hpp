// C++ mirror of the Zig host side of zigz, written ABOVE the C ABI of include/zigz_hip.h.
//
// The reference host is Zig (absent from this toolchain), so the host side that a Zig maintainer
// would keep is restated here in C++ with the reference's names, argument meaning and error
// behaviour: Multilinear, SumcheckProver, SimpleMerkleTree, CommitmentScheme, LassoProver,
// VMState/ExecutionTrace, WitnessGenerator, Prover, BinarySerializer, Verifier.  Every field or hash
// operation on the data path goes through libzigz_hip.so (zigz_* calls); this layer holds only the
// sequential host logic (VM, transcript schedule, proof packaging).
#pragma once
#include <array>
#include <condition_variable>
#include <cstdint>
#include <exception>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "zigz_hip.h"

namespace zigz {

constexpr uint64_t BABYBEAR = ZIGZ_BABYBEAR_P;  // src/core/field_presets.zig:19
using F = uint64_t;                             // canonical value of Field(u64, BabyBear)
using Hash = std::array<uint8_t, 32>;           // src/commitments/merkle_tree.zig:25

// Zig error unions -> one exception type carrying the status code / Zig error name
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &what) : std::runtime_error(what), code(c) {}
};
// host-only error codes (continue the zigz_status numbering; shared with oracle numbering)
enum : int {
    ERR_UNIMPLEMENTED_INSTRUCTION = 19, ERR_UNIMPLEMENTED_SYSTEM = 20, ERR_INVALID_OP32 = 21,
    ERR_INVALID_LOAD_FUNCT3 = 22, ERR_INVALID_STORE_FUNCT3 = 23, ERR_INVALID_BRANCH_FUNCT3 = 24,
    ERR_PROGRAM_HASH_MISMATCH = 25, ERR_INVALID_MAGIC = 26, ERR_UNSUPPORTED_VERSION = 27,
    ERR_FIELD_MISMATCH = 28, ERR_INVALID_DATA = 29, ERR_MAX_STEPS_EXCEEDED = 30, ERR_VM_HALTED = 31
};
void check(zigz_ctx *ctx, zigz_status s);  // throws Error on s != ZIGZ_OK

inline F finit(uint64_t v) { return v % BABYBEAR; }  // F.init, src/core/field.zig:36-38

// ---------------------------------------------------------------- FiatShamirTranscript (src/core/hash.zig:255-324)
class FiatShamirTranscript {
  public:
    FiatShamirTranscript();
    ~FiatShamirTranscript();
    FiatShamirTranscript(const FiatShamirTranscript &) = delete;
    FiatShamirTranscript &operator=(const FiatShamirTranscript &) = delete;
    void reset();  // = FiatShamirTranscript.init()
    void appendFieldElement(F e);
    void appendFieldElements(const std::vector<F> &es);
    void appendBytes(const void *data, size_t len);
    void appendBytes(const char *s);
    void appendTaggedCounter(const char *tag, uint64_t start, uint64_t count);
    F challenge();

  private:
    zigz_transcript *t_;
};

// ---------------------------------------------------------------- Multilinear(F) (src/poly/multilinear.zig)
struct Multilinear {
    std::vector<F> evaluations;
    size_t num_vars = 0;
    zigz_ctx *ctx = nullptr;

    static Multilinear init(zigz_ctx *ctx, const std::vector<F> &evals);  // :36-54
    F eval(const std::vector<F> &point) const;                            // :110-144
    Multilinear partialEval(F r) const;                                   // :154-180
    F sumOverHypercube() const;                                           // :188-194
    std::vector<F> roundPolynomial() const;                               // :205-232
};

// ---------------------------------------------------------------- sumcheck (src/proofs/)
struct SumcheckProof {  // sumcheck_protocol.zig:24-109
    std::vector<std::array<F, 2>> round_polynomials;
    std::vector<F> final_point;
    F final_eval = 0;
    size_t num_vars = 0;
    std::vector<uint8_t> toBytes() const;  // :76-107
};
F evalUnivariateCoeffs(const F *coeffs, size_t n, F x);  // :113-123 (host scalar Horner)
struct SumcheckProver {                                  // sumcheck_prover.zig
    static SumcheckProof prove(const Multilinear &poly);                                  // :26-91
    static SumcheckProof proveInteractive(const Multilinear &poly, const std::vector<F> &challenges);  // :97-144
};

// ---------------------------------------------------------------- Merkle + commitments
struct MerklePath {  // merkle_tree.zig:39-60
    std::vector<Hash> siblings;
    std::vector<uint8_t> directions;  // 0 = left, 1 = right
};
struct MerkleOpening {  // merkle_tree.zig:63-75 OpeningProof(F)
    size_t index = 0;
    F value = 0;
    MerklePath path;
};
class SimpleMerkleTree {  // merkle_tree.zig:273-401, HashFn = SHA3Hasher
  public:
    static SimpleMerkleTree build(zigz_ctx *ctx, const std::vector<F> &values);
    SimpleMerkleTree(SimpleMerkleTree &&o) noexcept;
    SimpleMerkleTree &operator=(SimpleMerkleTree &&o) noexcept;
    ~SimpleMerkleTree();
    Hash getRoot() const { return root_hash; }
    MerkleOpening open(size_t index) const;
    static bool verify(const Hash &root, const MerkleOpening &proof);  // :362-373 (host SHA3, verifier side)
    Hash root_hash{};
    size_t height = 0;
    zigz_merkle *handle() const { return t_; }

  private:
    SimpleMerkleTree() = default;
    zigz_ctx *ctx_ = nullptr;
    zigz_merkle *t_ = nullptr;
};
struct PolyOpeningProof {  // polynomial_commit.zig:42-55 OpeningProof(F)
    std::vector<F> point;
    F value = 0;
    MerkleOpening merkle_proof;
};
struct CommitmentScheme {  // polynomial_commit.zig:58-185 (CommitmentSchemeSHA3)
    struct Commit { Hash commitment; size_t num_vars; SimpleMerkleTree tree; };
    static Commit commit(const Multilinear &poly);
    static PolyOpeningProof open(const Multilinear &poly, const SimpleMerkleTree &tree, const std::vector<F> &point);
    static bool verify(const Hash &commitment, size_t num_vars, const PolyOpeningProof &proof);
};

// ---------------------------------------------------------------- Lasso (src/lookups/)
struct TableEntry { std::vector<F> inputs, outputs; };  // table_builder.zig:14-35
struct DenseTable {                                      // table_builder.zig:38-84
    std::vector<TableEntry> entries;
    size_t num_inputs = 0, num_outputs = 0;
};
DenseTable buildAddTable(size_t bits);  // table_builder.zig:126-153
DenseTable buildXorTable(size_t bits);  // :156-183
DenseTable buildAndTable(size_t bits);  // :186-213
struct LookupQuery { std::vector<F> inputs, expected_outputs; };  // lasso_prover.zig:65-86
struct LassoProofFull {                                            // lasso_prover.zig:27-62
    SumcheckProof sumcheck_proof;
    Hash query_commitment{}, table_commitment{};
    size_t num_lookups = 0;
};
struct LassoProver {
    static LassoProofFull prove(zigz_ctx *ctx, const DenseTable &table, const std::vector<LookupQuery> &queries);  // :103-173
    static LassoProofFull proveWithMapping(zigz_ctx *ctx, const DenseTable &table, const std::vector<LookupQuery> &queries,
                                           const std::vector<size_t> &mapping);  // :179-205
};

// ---------------------------------------------------------------- VM (src/vm/, src/isa/)
struct Instruction {  // rv64i.zig:111-151
    uint8_t opcode = 0, rd = 0, funct3 = 0, rs1 = 0, rs2 = 0, funct7 = 0;
    int64_t imm = 0;
    static bool decode(uint32_t word, Instruction &out);  // false = error.InvalidInstruction
};
bool hasLookupTable(const Instruction &inst);  // getTableMetadata(inst) != null, instruction_table.zig:243-274
struct Segment { uint64_t vaddr; std::vector<uint8_t> data; };  // src/elf.zig:8-11

// One recorded step = one compact record (zigz_trace_step, include/zigz_hip.h): pc, the decoded instruction fields, the one
// register the step wrote and its memory access -- what trace.zig:73-97 `Step` holds minus the two full register files.
// The 43 raw witness words of a step (prover.zig:376-390 column order; witness.zig:76,112,164-170,237-239) are rebuilt
// from the records by carrying register values forward: on the GPU (zigz_dev_witness_from_steps) for proving, on the
// host (expandRows) for callers that want the rows or the host-side WitnessGenerator.
constexpr size_t ROW_WORDS = 43;
struct ExecutionTrace {  // trace.zig:16-70
    std::vector<zigz_trace_step> steps;  // capacity grows in large chunks; see appendStep
    std::vector<uint8_t> is_lookup;      // per step
    uint64_t initial_regs[32] = {0};     // register file before the first recorded step
    size_t stepCount() const { return is_lookup.size(); }
    void reserveSteps(size_t n);      // virtual reservation only: pages are touched as steps are recorded
    zigz_trace_step *appendStep();    // storage of the next step's record (trace.addStep)
    std::vector<uint64_t> expandRows() const;  // [stepCount()][43] raw u64 words
};
class VMState {  // state.zig:35-598
  public:
    VMState(const std::vector<uint8_t> &program, uint64_t start_pc, const std::vector<uint64_t> *input);
    VMState(const std::vector<Segment> &segments, uint64_t entry_pc, const std::vector<uint64_t> *input);
    ~VMState();
    void step();                  // throws Error (InvalidInstruction is reported via `invalid_instruction`)
    void run(size_t max_steps);   // state.zig:172-184
    uint64_t readReg(unsigned r) const { return r ? regs_[r] : 0; }
    void writeReg(unsigned r, uint64_t v) {
        if (r) { regs_[r] = v; wr_reg_ = (uint8_t)r; wr_val_ = v; }
    }
    uint64_t pc = 0;
    bool halted = false;
    bool invalid_instruction = false;  // set when step() hit error.InvalidInstruction
    ExecutionTrace trace;
    std::vector<uint64_t> output_tape;
    size_t step_count = 0;

  private:
    struct Mem;
    Mem *mem_;
    uint64_t regs_[32] = {0};
    uint8_t wr_reg_ = 0;   // register written by the instruction being executed (0 = none) and its value
    uint64_t wr_val_ = 0;
    std::vector<uint64_t> input_tape_;
    size_t input_pos_ = 0;
    uint64_t execute(const Instruction &in, uint64_t *mem_row /*[3]: addr,value,is_read*/);
};

// ---------------------------------------------------------------- witness (src/constraints/witness.zig)
struct Witness {  // :274-313; the 43 columns in prover order, column-major
    size_t num_vars = 0, num_steps = 0;
    std::vector<F> columns;  // [43][2^num_vars] canonical
    const F *column(size_t c) const { return columns.data() + (c << num_vars); }
    size_t size() const { return columns.size(); }
};
struct WitnessGenerator {
    static Witness generate(const ExecutionTrace &trace);  // :29-61
};

// ---------------------------------------------------------------- proof structures (src/prover/proof.zig)
struct PublicIO {  // :18-50
    Hash program_hash{};
    uint64_t initial_pc = 0;
    std::optional<std::vector<uint64_t>> initial_regs;
    uint64_t final_pc = 0;
    std::optional<std::vector<uint64_t>> final_regs;
    size_t num_steps = 0;
    std::optional<std::vector<uint64_t>> outputs;
};
struct ProverSumcheckProof {  // :53-99 (degree known from the coefficient count)
    size_t num_vars = 0;
    size_t num_coeffs = 0;             // degree + 1
    std::vector<F> round_polynomials;  // [num_vars][num_coeffs]
    std::vector<F> final_point;
    F final_eval = 0;
};
struct LassoProof {  // :102-144
    uint32_t table_id = 0;
    size_t num_lookups = 0;
    ProverSumcheckProof multiset_proof;
};
struct CommitmentOpening {  // :147-191
    Hash commitment{};
    std::vector<F> point;
    F value = 0;
    PolyOpeningProof proof;
};
struct ProofMetadata { size_t num_steps = 0, num_vars = 0; uint64_t field_modulus = BABYBEAR; uint32_t version = 1; };  // :317-329
struct Proof {  // :194-314
    PublicIO public_io;
    ProverSumcheckProof constraint_proof;
    // Prover.prove emits one identical placeholder per lookup step (prover.zig:302-322: table_id = i,
    // num_lookups = 1, 0 variables, final_eval = 0).  They are kept as a count instead of 2^20 heap
    // objects; lookup_proofs holds only entries that are NOT of that form (deserialised foreign data).
    // Logical list = placeholders [0, lookup_placeholders) followed by lookup_proofs.
    size_t lookup_placeholders = 0;
    std::vector<LassoProof> lookup_proofs;
    size_t lookupCount() const { return lookup_placeholders + lookup_proofs.size(); }
    LassoProof lookupAt(size_t i) const;
    std::vector<CommitmentOpening> witness_commitments;  // 43
    ProofMetadata metadata;
    static Proof init(size_t num_steps);  // :224-261
    size_t estimateSize() const;          // :279-312
};
enum class VerificationResult { Accept = 0, RejectInvalidSumcheck, RejectInvalidLookup, RejectInvalidCommitment, RejectInvalidPublicIO };  // :335-341

// ---------------------------------------------------------------- Prover / serializer / verifier
// One proof over several GPUs ("by column"): rank r commits and opens a contiguous block of the 43 columns;
// the two exchanges of generateCommitments -- the 43 roots before the transcript absorbs them, the 43 openings after --
// go through an all-gather hook supplied by the host (RCCL / MPI / torch.distributed): every rank contributes `bytes`
// from `send`, `recv` receives world * bytes in rank order; returns 0 on success.  Transcripts run in lockstep, so every
// rank ends with the same, complete proof.
typedef int (*AllGatherFn)(void *user, const void *send, size_t bytes, void *recv);
struct ShardSpec {
    int rank = 0, world = 1;
    AllGatherFn allgather = nullptr;
    void *user = nullptr;
};
// contiguous blocks, sizes differing by at most one: 43 over 8 -> 6,6,6,5,5,5,5,5
inline void columnBlock(size_t ncols, int world, int rank, size_t &c0, size_t &c1) {
    const size_t base = ncols / (size_t)world, extra = ncols % (size_t)world, r = (size_t)rank;
    c0 = r * base + (r < extra ? r : extra);
    c1 = c0 + base + (r < extra ? 1 : 0);
}

// GPU slots of a proving service.  One proof needs the GPU for a few milliseconds -- Merkle builds, roots, openings -- and a
// host core for tens: its sequential O(L) transcript (prover.zig:292-363).  A service that proves many traces at once
// therefore does NOT give every proof in flight a context of its own (stream + tree / list workspaces: 0.6 GiB at 2^20,
// 9-10 GiB at 2^24, held through the whole transcript): it keeps K contexts, and a proving thread takes one only between the
// end of its step [5/6] and the end of its openings.  The reference's order permits that: the trees do not depend on the
// transcript, the points do (prover.zig:405-424), so "begin" may run at any time before "roots".  Proofs in flight are then
// bounded by host cores and by the witnesses themselves, not by HBM for workspaces; the overlap of one proof's build with
// its own transcript is given up (it is other proofs' builds that keep the GPU busy meanwhile).
// FIFO among waiters; the most recently released context is handed out first, so K may be generous: contexts that are never
// needed never grow workspaces.
class GpuSlots {
  public:
    GpuSlots(int device, size_t k);
    ~GpuSlots();
    GpuSlots(const GpuSlots &) = delete;
    GpuSlots &operator=(const GpuSlots &) = delete;
    zigz_ctx *acquire();
    void release(zigz_ctx *ctx);
    size_t size() const { return all_.size(); }
    zigz_ctx *at(size_t i) const { return all_[i]; }  // set-up only (options, timing): while nobody proves
    struct Lease {
        GpuSlots *slots = nullptr;
        zigz_ctx *ctx = nullptr;
        ~Lease() { drop(); }
        void drop() {
            if (slots && ctx) slots->release(ctx);
            ctx = nullptr;
        }
    };

  private:
    std::mutex m_;
    std::condition_variable cv_;
    std::vector<zigz_ctx *> all_, free_;
    uint64_t next_ticket_ = 0, serving_ = 0;
};

// Small traces: proofs that arrive at their GPU phase within a short while of each other share ONE commit job
// (zigz_commit_begin_batch: one structure pass, one hash launch per level, one eval, one path launch for all of them) --
// at 2^16 and below a proof's ~35 launches are mostly latency, and a service's small proofs arrive in bursts anyway (the
// lanes of one sponge server leave their transcripts together).  A group forms per (size, hints); the member that opened it
// waits until it is full or `linger` has passed, then for a GPU slot -- the group stays open meanwhile, so with a backlog in
// front of the GPU it grows to what is waiting -- closes it and runs begin + roots for all; each
// member then absorbs ITS roots and draws ITS challenges on its own thread (the transcripts stay per proof: the bytes of a
// proof do not depend on whom it shared a launch with), and the last member to hand in its points runs open_all + end for
// all.  Results are per proof exactly what a job of its own gives (tests/test_gpu_parity.py: batch == single jobs).
class GpuBatcher {
  public:
    struct Group;
    GpuBatcher(GpuSlots *slots, unsigned max_batch, double linger_s, size_t max_nv)
        : slots_(slots), max_batch_(max_batch), linger_s_(linger_s), max_nv_(max_nv) {}
    size_t maxNv() const { return max_nv_; }
    // joins (or opens) a group; returns when the group's roots are there: *idx = this proof's position, roots43 = its 43 roots
    std::shared_ptr<Group> join(const uint32_t *d_cols, size_t stride, size_t nv, uint64_t m_small, uint64_t m_run, uint64_t m_cons,
                                unsigned *idx, uint8_t *roots43);
    // hands in this proof's 43 * nv points; returns with its openings when the group's open_all has run
    void open(const std::shared_ptr<Group> &g, unsigned idx, const F *points, F *values, F *indices, F *leaves, uint8_t *sib,
              uint8_t *dirs, zigz_kernel_stats *stats);
    // a member that cannot go on (its transcript work threw) must still let the others finish
    void abandon(const std::shared_ptr<Group> &g, unsigned idx);

  private:
    void build(const std::shared_ptr<Group> &g);      // (called without the lock held)
    void finish(const std::shared_ptr<Group> &g);
    void run_open_all(const std::shared_ptr<Group> &g);
    GpuSlots *slots_;
    unsigned max_batch_;
    double linger_s_;
    size_t max_nv_;
    std::mutex m_;
    std::map<std::array<uint64_t, 5>, std::shared_ptr<Group>> open_;
};

// the compact trace records of a proof whose witness is built inside its GPU slot: the 48-byte form, the 32-byte form + the
// side list of memory accesses, or the 16-byte form + side list + code table (zigz_hip.h)
struct TraceRecords {
    const zigz_trace_step *s48 = nullptr;
    const zigz_trace_step32 *s32 = nullptr;
    const zigz_trace_step16 *s16 = nullptr;  // the 16-byte form + the code table (instruction fields once per pc)
    const zigz_code_entry *code = nullptr;
    size_t ncode = 0;
    uint64_t code_base = 0;
    const zigz_mem_access *mem = nullptr;
    size_t nmem = 0;
    const uint64_t *regs_before = nullptr;
    explicit operator bool() const { return s48 || s32 || s16; }
};

class Prover {  // src/prover/prover.zig
  public:
    Prover(zigz_ctx *ctx, uint64_t seed) : ctx_(ctx), seed_(seed) {}
    // a prover of a service: the context is taken from `slots` for the GPU phases of each proof only (see GpuSlots)
    Prover(GpuSlots *slots, uint64_t seed, GpuBatcher *batcher = nullptr) : ctx_(nullptr), seed_(seed), slots_(slots), batcher_(batcher) {}
    void setShard(const ShardSpec &s) { shard_ = s; }
    // prove(program, entry_pc, initial_regs, max_steps, segments, input), :73-226
    // serialized (optional): also BinarySerializer.serialize, with the early sections written underneath the transcript
    Proof prove(const std::vector<uint8_t> &program, uint64_t entry_pc, const std::vector<uint64_t> *initial_regs,
                size_t max_steps, const std::vector<Segment> *segments, const std::vector<uint64_t> *input,
                std::vector<uint8_t> *serialized = nullptr);
    // Steps [4/6]-[6/6] + packagePublicIO on an existing witness (the data-parallel part): the columns
    // may be host canonical u64 (`witness`) or already resident in HBM as packed u32 (`d_cols`).
    Proof proveWitness(const PublicIO &io_template, size_t num_lookups, const Witness *witness, const uint32_t *d_cols,
                       size_t d_col_stride, size_t num_vars, const std::vector<uint64_t> *initial_regs);
    // proveWitness + BinarySerializer.serialize with the serialisation of everything but the 43 openings
    // overlapped (helper thread) with the sequential transcript; `out` is resized to the exact proof size.
    void proveWitnessToBytes(const PublicIO &io_template, size_t num_lookups, const Witness *witness, const uint32_t *d_cols,
                             size_t d_col_stride, size_t num_vars, const std::vector<uint64_t> *initial_regs,
                             std::vector<uint8_t> &out);
    // The same from the compact trace records (zigz_trace_step, page-locked): the witness is built inside the proof's GPU
    // slot, in a column buffer the slot's context owns (upload + expansion kernels + commit job on one stream) -- a proof in
    // flight then holds nothing in HBM outside its slot.  Needs a prover of a service (GpuSlots).
    void proveStepsToBytes(const PublicIO &io_template, size_t num_lookups, const TraceRecords &records, size_t num_vars,
                           const std::vector<uint64_t> *initial_regs, std::vector<uint8_t> &out);
    // wall-clock seconds of the phases of the last proveWitness: 0 commit_begin, 1 sumcheck transcript,
    // 2 lasso transcript, 3 wait for roots, 4 absorb roots + challenges, 5 open_all, 6 packaging, 7 serialisation tail,
    // 8 wait for a GPU slot, 9 the whole time in the slot
    double timings[10] = {0};
    // a prover of a service: the kernel statistics (and, in timing mode, the launch log) of the context the last proof used,
    // copied while it still held it
    zigz_kernel_stats last_stats{};
    std::vector<zigz_launch_rec> last_log;
    bool verbose = false;  // the reference prints progress banners unconditionally (prover.zig:82-85); off by default here
    // leaf / level-1 digests of the structurally small-domain witness columns by table lookup (identical trees); the
    // environment variable ZIGZ_DENSE_MERKLE=1 turns it off process-wide (A/B measurements)
    bool small_domain_tables = defaultSmallDomain();
    static bool defaultSmallDomain();
    // run-aware Merkle levels (identical trees): 0 off; 1 the register columns x1..x31 (at most one of them changes per
    // step, whatever the program); 3 the registers and mem.address / mem.value (0 on every step that is not a
    // LOAD / STORE); 2 every column that is not small-domain; 4 (default) = 3 + the ten instruction-determined columns as a
    // content-addressed group.  Environment: ZIGZ_RUN_AWARE=off|regs|struct|all|cons (default cons); ZIGZ_DENSE_MERKLE=1
    // turns this off too
    int run_aware = defaultRunAware();
    static int defaultRunAware();

  private:
    void bindPublicInputs(const Hash &program_hash, uint64_t entry_pc, const std::vector<uint64_t> *initial_regs);  // :91-110
    void generateSumcheckProof(Proof &proof, size_t num_steps, size_t num_vars);                                     // :229-289
    void generateLassoProofs(Proof &proof, size_t num_lookups);                                                      // :292-363
    // the two GPU steps of generateCommitments: the 43 roots of this proof; its openings at the 43 points.  Bound to a commit
    // job of this proof's own, or to its place in a batch (GpuBatcher)
    struct CommitSteps {
        std::function<void(uint8_t *roots43)> roots;
        std::function<void(const F *points, F *values, F *indices, F *leaves, uint8_t *sib, uint8_t *dirs)> open_all;
    };
    void generateCommitments(Proof &proof, const CommitSteps &gpu, size_t num_vars);                                 // :366-467
    Proof proveWitnessImpl(const PublicIO &io, size_t num_lookups, const Witness *witness, const uint32_t *d_cols,
                           size_t d_col_stride, size_t num_vars, const std::vector<uint64_t> *initial_regs,
                           std::vector<uint8_t> *bytes_out, const TraceRecords &records = TraceRecords());
    zigz_ctx *ctx_;
    uint64_t seed_;
    GpuSlots *slots_ = nullptr;
    GpuBatcher *batcher_ = nullptr;
    FiatShamirTranscript transcript_;
    ShardSpec shard_;
};

struct BinarySerializer {  // src/prover/serialization.zig
    static std::vector<uint8_t> serialize(const Proof &proof);             // :70-97 with an exact-size buffer
    static Proof deserialize(const uint8_t *data, size_t len);             // :100-131
    static size_t exactSize(const Proof &proof);
    // The same byte layout written in two independent parts, so the large part that is known early (header,
    // public IO, constraint proof, Lasso placeholders) can be written on a helper thread while the main
    // thread is still inside the transcript: [0, prefixSize) then [prefixSize, exactSize).
    static size_t prefixSize(const Proof &proof);
    static void writePrefix(const Proof &proof, uint8_t *buf);
    static void writeCommitments(const Proof &proof, uint8_t *buf_at_prefix_end);
};

class Verifier {  // src/verifier/verifier.zig:26-300 (host-only: 43*v SHA3 merges)
  public:
    VerificationResult verify(const Proof &proof, const std::vector<uint8_t> &program);  // :49-91

  private:
    FiatShamirTranscript transcript_;
    VerificationResult verifySumcheckProof(const ProverSumcheckProof &p);  // :182-238
};

}  // namespace zigz
