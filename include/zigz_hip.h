/*
 * zigz_hip.h -- C ABI of libzigz_hip.so: the MI355X (gfx950) backend for zigz's data-parallel
 * hot path (BabyBear MLE bind/eval, sumcheck rounds, SHA3 Merkle commit/open, Lasso fingerprints).
 *
 * This is the drop-in boundary.  The reference (Zig, single-threaded, no FFI of its own) keeps its
 * host code; each entry point below replaces the body of ONE reference function, cited per
 * declaration as reference file:line.  A Zig `extern "c"` shim that binds these symbols is shown
 * in INTEGRATION.md.
 *
 * Conventions (mirroring the reference):
 *  - Field elements cross the boundary as canonical u64 little-endian values in [0, p),
 *    p = 2013265921 (BabyBear, src/core/field_presets.zig:19) -- the in-memory layout of
 *    `[]F` for `F = Field(u64, p)` (src/core/field.zig:27), so Zig slices pass as-is.
 *    Non-canonical inputs are rejected with ZIGZ_ERR_NOT_CANONICAL.
 *  - Results are written into caller-allocated buffers (Zig callers own every returned slice);
 *    opaque handles have an explicit *_destroy.
 *  - Every function returns a zigz_status; the codes map 1:1 onto the Zig error names.
 *  - No global state: all state lives in a zigz_ctx (one per thread / per GPU).  A context is
 *    bound to one HIP device and one stream; calls on one context must not race.
 *  - Device-resident variants (zigz_dev_*) take device pointers to packed u32 canonical elements
 *    (4 B/element in HBM) and run on the context's stream; plain pointers, no torch types.
 *  - The library never falls back to the CPU for field or hash work: without a usable gfx950
 *    device zigz_ctx_create fails with ZIGZ_ERR_NO_DEVICE.
 */
#ifndef ZIGZ_HIP_H
#define ZIGZ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZIGZ_BABYBEAR_P 2013265921ull
#define ZIGZ_NUM_COLUMNS 43 /* src/prover/prover.zig:376-390 */
#define ZIGZ_HASH_SIZE 32   /* src/commitments/merkle_tree.zig:24 */
#define ZIGZ_ABI_VERSION 1

typedef int32_t zigz_status;
enum {
    ZIGZ_OK = 0,
    ZIGZ_ERR_EMPTY_EVALUATIONS = 1,         /* error.EmptyEvaluations      multilinear.zig:37-39 */
    ZIGZ_ERR_LENGTH_NOT_POWER_OF_TWO = 2,   /* error.LengthNotPowerOfTwo   multilinear.zig:42-44 */
    ZIGZ_ERR_WRONG_NUMBER_OF_VARIABLES = 3, /* error.WrongNumberOfVariables multilinear.zig:111-113 */
    ZIGZ_ERR_NO_VARIABLES_TO_FIX = 4,       /* error.NoVariablesToFix      multilinear.zig:155-157 */
    ZIGZ_ERR_NO_VARIABLES = 5,              /* error.NoVariables           multilinear.zig:206-208 */
    ZIGZ_ERR_PROTOCOL_ERROR = 6,            /* error.ProtocolError         sumcheck_prover.zig:80-82 */
    ZIGZ_ERR_EMPTY_VALUES = 7,              /* error.EmptyValues           merkle_tree.zig:284 */
    ZIGZ_ERR_TOO_MANY_VALUES = 8,           /* error.TooManyValues         merkle_tree.zig:287 */
    ZIGZ_ERR_INDEX_OUT_OF_BOUNDS = 9,       /* error.IndexOutOfBounds      merkle_tree.zig:325 */
    ZIGZ_ERR_POINT_DIMENSION_MISMATCH = 10, /* error.PointDimensionMismatch polynomial_commit.zig:92-94 */
    ZIGZ_ERR_NO_QUERIES = 11,               /* error.NoQueries             lasso_prover.zig:108-110 */
    ZIGZ_ERR_TOO_MANY_QUERIES = 12,         /* error.TooManyQueries        lasso_prover.zig:131 */
    ZIGZ_ERR_MAPPING_LENGTH_MISMATCH = 13,  /* error.MappingLengthMismatch lasso_prover.zig:185-187 */
    ZIGZ_ERR_INVALID_MAPPING = 14,          /* error.InvalidMapping        lasso_prover.zig:191-193 */
    ZIGZ_ERR_QUERY_TABLE_MISMATCH = 15,     /* error.QueryTableMismatch    lasso_prover.zig:198-200 */
    ZIGZ_ERR_EMPTY_TRACE = 16,              /* error.EmptyTrace            prover.zig:147-149 */
    ZIGZ_ERR_OUT_OF_MEMORY = 17,            /* error.OutOfMemory (host or device) */
    ZIGZ_ERR_WRONG_NUMBER_OF_CHALLENGES = 18, /* error.WrongNumberOfChallenges sumcheck_prover.zig:105-107 */
    /* backend-specific (no Zig counterpart) */
    ZIGZ_ERR_NO_DEVICE = 100,     /* no HIP device / not gfx950 / kernels not loadable */
    ZIGZ_ERR_HIP = 101,           /* a HIP runtime call failed; see zigz_last_error */
    ZIGZ_ERR_NOT_CANONICAL = 102, /* a field element >= p crossed the boundary */
    ZIGZ_ERR_INVALID_ARGUMENT = 103,
    ZIGZ_ERR_BAD_STATE = 104,     /* commit-job calls out of order */
    ZIGZ_ERR_COMM = 105           /* a sharded proof's exchange hook failed, or another rank reported an error */
};

typedef struct zigz_ctx zigz_ctx;
typedef struct zigz_merkle zigz_merkle;         /* one committed column: all tree levels in HBM */
typedef struct zigz_commit_job zigz_commit_job; /* a batch of columns going through commit -> open */
typedef struct zigz_transcript zigz_transcript; /* host-side SHA3 Fiat-Shamir sponge */

/* ---------------------------------------------------------------- context */
uint32_t zigz_abi_version(void);
const char *zigz_status_name(zigz_status s); /* the Zig error name, e.g. "LengthNotPowerOfTwo" */
zigz_status zigz_device_count(int *count);
/* Binds a context to HIP device `device` (must be gfx950) and creates its stream + workspace. */
zigz_status zigz_ctx_create(int device, zigz_ctx **out);
/* How host threads of this process wait for the device (stream / event synchronisation inside the calls below): on != 0
 * they sleep until the device signals instead of spinning.  For a host that keeps more proofs in flight than it has cores
 * (one thread per proof, most of them waiting for the GPU or for the sponge service); costs a wake-up per wait, so leave it
 * off for a lone proof.  Call before the contexts of `device` are created.  With it on, the two waits of a commit job
 * (zigz_commit_roots, zigz_commit_open_all) do not use the runtime's wait at all: the last kernel stores a completion word
 * into pinned memory next to the results it wrote there, and the thread looks at it between sleeps of 30-150 us (the
 * runtime's interrupt wait costs 0.2-0.5 ms of CPU per wait once tens of threads wait at once). */
zigz_status zigz_device_set_blocking_sync(int device, int on);
void zigz_ctx_destroy(zigz_ctx *ctx);
const char *zigz_last_error(const zigz_ctx *ctx);
/* Run on a caller-owned hipStream_t (e.g. torch's current stream); NULL restores the own stream. */
zigz_status zigz_ctx_set_stream(zigz_ctx *ctx, void *hip_stream);
void *zigz_ctx_get_stream(zigz_ctx *ctx);
zigz_status zigz_ctx_synchronize(zigz_ctx *ctx);
/* Device memory for callers without their own allocator (hipMalloc / hipFree). */
zigz_status zigz_dev_alloc(zigz_ctx *ctx, size_t bytes, void **d_out);
zigz_status zigz_dev_free(zigz_ctx *ctx, void *d_ptr);
/* Gives the context's workspaces back to the device (they are grown on demand and otherwise kept for the life of the
 * context: trees, lists, staging).  For a host that has just put an unusually large job through a context it keeps many of;
 * what the context learnt about its traces (the room its lists need) stays.  ZIGZ_ERR_BAD_STATE while a commit job is active. */
zigz_status zigz_ctx_release_workspaces(zigz_ctx *ctx);
/* free / total HBM of the context's device (hipMemGetInfo): a service sizes its number of proofs in flight from it */
zigz_status zigz_dev_mem_info(zigz_ctx *ctx, size_t *free_bytes, size_t *total_bytes);
/* canonical u64 host -> packed u32 device (validates < p), and back */
zigz_status zigz_dev_upload_u64(zigz_ctx *ctx, const uint64_t *h_in, size_t n, uint32_t *d_out);
zigz_status zigz_dev_download_u64(zigz_ctx *ctx, const uint32_t *d_in, size_t n, uint64_t *h_out);

/* ---------------------------------------------------------------- Multilinear(F), host buffers
 * Drop-in for src/poly/multilinear.zig; `in` is `self.evaluations` (len n = 2^v). */
/* partialEval(self, r, allocator) !Self   multilinear.zig:154-180: out[i]=(1-r)a[i]+r*a[i+n/2], i<n/2 */
zigz_status zigz_mle_bind(zigz_ctx *ctx, const uint64_t *in, size_t n, uint64_t r, uint64_t *out);
/* roundPolynomial(self, allocator) ![]F   multilinear.zig:205-232: out = [s0, s1 - s0] */
zigz_status zigz_mle_round_poly(zigz_ctx *ctx, const uint64_t *in, size_t n, uint64_t out[2]);
/* sumOverHypercube(self) F                multilinear.zig:188-194 */
zigz_status zigz_mle_sum(zigz_ctx *ctx, const uint64_t *in, size_t n, uint64_t *out);
/* eval(self, point) !F                    multilinear.zig:110-144 (point[0] <-> LSB of the index) */
zigz_status zigz_mle_eval(zigz_ctx *ctx, const uint64_t *in, size_t n, const uint64_t *point,
                          size_t point_len, uint64_t *out);

/* ---------------------------------------------------------------- SumcheckProver(F)
 * prove(poly, allocator) !Proof           src/proofs/sumcheck_prover.zig:26-91
 * rounds: 2*v values [c0,c1] per round; point: v challenges in binding order; SHA3 transcript
 * (fresh per call, src/proofs/sumcheck_protocol.zig:161,176-184) runs on the host inside. */
zigz_status zigz_sumcheck_prove(zigz_ctx *ctx, const uint64_t *in, size_t n, uint64_t *rounds,
                                uint64_t *point, uint64_t *final_eval);
/* proveInteractive(poly, challenges, allocator)  sumcheck_prover.zig:97-144 */
zigz_status zigz_sumcheck_prove_interactive(zigz_ctx *ctx, const uint64_t *in, size_t n,
                                            const uint64_t *challenges, size_t n_challenges,
                                            uint64_t *rounds, uint64_t *point, uint64_t *final_eval);

/* ---------------------------------------------------------------- SimpleMerkleTree(F, SHA3Hasher)
 * build(values, allocator) !Self          src/commitments/merkle_tree.zig:283-318 (pads with hashLeaf(0)) */
zigz_status zigz_merkle_commit(zigz_ctx *ctx, const uint64_t *values, size_t n, uint8_t root[32],
                               size_t *height, zigz_merkle **out);
/* open(self, index) !OpeningProof(F)      merkle_tree.zig:324-360; siblings: 32*height B, dirs: height B */
zigz_status zigz_merkle_open(zigz_ctx *ctx, const zigz_merkle *tree, size_t index, uint8_t *siblings,
                             uint8_t *dirs, uint64_t *leaf_value);
void zigz_merkle_destroy(zigz_ctx *ctx, zigz_merkle *tree);
/* CommitmentScheme.open(poly, tree, point, allocator) !Proof   src/commitments/polynomial_commit.zig:86-115
 * = eval(point) + tree.open(point[0].value mod 2^v)  (pointToIndex, :178-183) */
zigz_status zigz_commit_open(zigz_ctx *ctx, const uint64_t *evals, size_t n, const zigz_merkle *tree,
                             const uint64_t *point, size_t point_len, uint64_t *value, uint64_t *index,
                             uint8_t *siblings, uint8_t *dirs, uint64_t *leaf_value);

/* ---------------------------------------------------------------- Prover.generateCommitments (batched)
 * src/prover/prover.zig:366-467, split at the two transcript dependencies so the caller keeps its
 * own Fiat-Shamir transcript:  begin (Merkle builds, async) -> roots -> [caller absorbs roots, draws
 * ncols*v challenges] -> open_all (evals + paths) -> end.
 * cols: column-major, column c at cols + c*col_stride (elements), each 2^nv long. */
zigz_status zigz_commit_begin(zigz_ctx *ctx, const uint64_t *cols, size_t ncols, size_t col_stride,
                              size_t nv, zigz_commit_job **out);
/* device-resident columns (packed u32 canonical); the buffer must stay valid until zigz_commit_end */
zigz_status zigz_commit_begin_dev(zigz_ctx *ctx, const uint32_t *d_cols, size_t ncols, size_t col_stride,
                                  size_t nv, zigz_commit_job **out);
/* Several proofs in ONE job (a proving service with many SMALL traces: the reference's own tests and examples prove 4 .. 64
 * steps, tests/integration_tests.zig:171-206, BASELINE config 2 is 2^16 -- sizes at which a proof's ~35 launches are mostly
 * latency).  d_cols[z]: proof z's `ncols` device-resident columns (stride col_stride), all proofs 2^nv rows; nproofs <= 32,
 * nproofs * ncols <= 4096.  The job then behaves like one of nproofs * ncols columns numbered proof by proof: roots, points,
 * values, indices, leaves, siblings and dirs of proof z occupy the positions [z * ncols, (z + 1) * ncols) of the arrays of
 * zigz_commit_roots / zigz_commit_open_all; every proof's results are the ones its own job would give.  The columns are
 * copied at begin (the callers' buffers are not needed after it returns ... after the stream has passed the copy:
 * zigz_commit_roots).  2^nv < 2^15: the proofs' trees are built densely like any table's; 2^15 <= 2^nv <= 2^18: with the
 * structure-aware levels (option "batch_reserve" = n sizes the context's workspace for n proofs from its first batched job on, so
 * that a larger batch than any before does not reallocate gigabytes in the middle of a service's work), every column of a proof must then be hinted ("run_aware_mask" / "cons_group_mask" cover all of
 * them, as the witness's 43 are in host/prover.cpp), else ZIGZ_ERR_INVALID_ARGUMENT; larger tables: one job per proof. */
zigz_status zigz_commit_begin_batch(zigz_ctx *ctx, const uint32_t *const *d_cols, size_t nproofs, size_t ncols,
                                    size_t col_stride, size_t nv, zigz_commit_job **out);
/* zigz_commit_begin* ENQUEUES the Merkle builds and returns (nothing on that path is read back by the host: ~0.15 ms for the
 * ~45 launches of a 2^20 x 43 job); zigz_commit_roots waits for them; roots: ncols*32 bytes (prover.zig:405-410).
 * One job per context at a time (a second begin returns ZIGZ_ERR_BAD_STATE).  Other calls on the same context
 * between begin and end are allowed -- e.g. a Zig host evaluating one MLE while the trees build: they queue behind
 * the builds on the context's stream, and the job's roots travel through a pinned buffer of their own, so they
 * arrive intact (tests/test_gpu_parity.py::test_commit_job_survives_interleaved_calls). */
zigz_status zigz_commit_roots(zigz_commit_job *job, uint8_t *roots);
/* points: ncols*nv challenges (row c = point of column c).  Outputs per column: value = eval(point)
 * (prover.zig:427), index = point[0] mod 2^nv, leaf = evaluations[index], siblings ncols*nv*32 B,
 * dirs ncols*nv B (prover.zig:431 -> polynomial_commit.zig:86-115 -> merkle_tree.zig:324-360). */
zigz_status zigz_commit_open_all(zigz_commit_job *job, const uint64_t *points, uint64_t *values,
                                 uint64_t *indices, uint64_t *leaves, uint8_t *siblings, uint8_t *dirs);
/* diagnostic (tests): device address of the job's trees once built -- per column `bytes_per_column` = 2 * 2^nv nodes x 32 B
 * in the kernels' internal node form (level l at node offset 2N - 2(N >> l)); valid until zigz_commit_end.  Lets a test
 * compare two builds of the same columns node for node (dense vs table / run-aware / content-addressed levels).  Whole
 * node-addressed trees exist only when no column is list-built or the job was begun with "run_aware_materialize" = 1:
 * ZIGZ_ERR_BAD_STATE otherwise. */
zigz_status zigz_commit_job_tree(zigz_commit_job *job, const void **d_tree, size_t *bytes_per_column);
void zigz_commit_end(zigz_commit_job *job);

/* ---------------------------------------------------------------- LassoProver(F)
 * prove(table, queries, allocator) !Proof   src/lookups/lasso_prover.zig:103-173
 * table / queries: row-major, each row = n_in input fields then n_out output fields.
 * Outputs: nv = log2(ceilPow2(n_queries)); rounds 2*nv; point nv; the two flat SHA3 commitments. */
zigz_status zigz_lasso_prove(zigz_ctx *ctx, const uint64_t *table, size_t table_rows, const uint64_t *queries,
                             size_t n_queries, size_t n_in, size_t n_out, size_t *nv_out, uint64_t *rounds,
                             uint64_t *point, uint64_t *final_eval, uint8_t query_commitment[32],
                             uint8_t table_commitment[32]);
/* proveWithMapping(table, queries, mapping, allocator)   lasso_prover.zig:179-205 */
zigz_status zigz_lasso_prove_with_mapping(zigz_ctx *ctx, const uint64_t *table, size_t table_rows,
                                          const uint64_t *queries, size_t n_queries, size_t n_in, size_t n_out,
                                          const uint64_t *mapping, size_t n_mapping, size_t *nv_out,
                                          uint64_t *rounds, uint64_t *point, uint64_t *final_eval,
                                          uint8_t query_commitment[32], uint8_t table_commitment[32]);
/* hashEntry / hashQuery fingerprints (lasso_prover.zig:208-239) for `rows` rows of `width` fields */
zigz_status zigz_lasso_fingerprints(zigz_ctx *ctx, const uint64_t *rows_in, size_t rows, size_t width,
                                    uint64_t *out);

/* ---------------------------------------------------------------- witness columns (K8)
 * F.init(u64) = x mod p over raw 64-bit trace words (src/constraints/witness.zig:76,112,164-170,237-239):
 * d_out[i] = h_in[i] mod p, packed u32, for building device-resident columns from raw trace data. */
zigz_status zigz_dev_reduce_u64(zigz_ctx *ctx, const uint64_t *h_in, size_t n, uint32_t *d_out);
/* WitnessGenerator.generate on the device   src/constraints/witness.zig:29-270, column order prover.zig:376-390.
 * h_rows: the execution trace as packed rows, [num_steps][43] raw 64-bit words in column order (pc, x0..x31,
 * opcode, rd, rs1, rs2, funct3, funct7, imm as u64(bitcast i64), mem address, mem value, is_read).  Builds the 43
 * padded columns of 2^nv packed-u32 cells (nv = log2_int_ceil(num_steps)) at d_cols + c*col_stride. */
zigz_status zigz_dev_witness_from_rows(zigz_ctx *ctx, const uint64_t *h_rows, size_t num_steps, size_t nv,
                                       uint32_t *d_cols, size_t col_stride);

/* The same columns from the COMPACT trace: one 48-byte record per executed step instead of its 43 raw words (344 B).
 * A `Step` of src/vm/trace.zig:73-97 carries two full register files, but regs_after differs from the previous step's
 * in at most the one register the instruction wrote (src/vm/state.zig:188-597: every instruction writes <= 1 register;
 * x0 never changes, src/vm/registers.zig:38-48), so a step is: pc, the instruction's decoded fields, the register
 * write (wr_reg, rd_value) and the memory access.  The device rebuilds the 32 register columns by carrying each
 * register's last written value forward (witness.zig:65-123), reduces every cell mod p (F.init, witness.zig:76,112,
 * 164-170,237-239) and applies the padding rule (pc and registers repeat their last value, the rest is 0:
 * witness.zig:80-87,116-123).  A Zig host fills one record per trace.steps.items[i]:
 *   pc = step.pc; opcode..funct7, imm = step.instruction.*; (wr_reg, rd_value) = the r with regs_after[r] != regs_before[r]
 *   (0 when none); mem_* = step.memory_access (zeros / mem_is_read = 0 when null; mem_is_read = 1 for .Load). */
typedef struct zigz_trace_step {
    uint64_t pc;        /* step.pc */
    uint64_t rd_value;  /* regs_after[wr_reg]; ignored when wr_reg == 0 */
    uint64_t mem_addr;  /* memory_access.address (0 when none) */
    uint64_t mem_value; /* memory_access.value   (0 when none) */
    int64_t imm;        /* instruction.imm */
    uint8_t opcode, rd, rs1, rs2, funct3, funct7; /* @intFromEnum(opcode) and the raw fields, witness.zig:164-169 */
    uint8_t wr_reg;      /* 1..31: the register this step wrote; 0 (or any value >= 32): none */
    uint8_t mem_is_read; /* 1 = load, 0 = store or no access */
} zigz_trace_step;
/* initial_regs: the 32 register values before the first step (NULL = all zero); x0 is forced to 0.  h_steps may be
 * pageable or pinned (zigz_host_register) memory; the call returns when the copy has completed. */
zigz_status zigz_dev_witness_from_steps(zigz_ctx *ctx, const zigz_trace_step *h_steps, size_t num_steps, size_t nv,
                                        const uint64_t *initial_regs, uint32_t *d_cols, size_t col_stride);
/* The same, without waiting: copy, expansion and whatever the caller enqueues next (zigz_commit_begin_dev on d_cols)
 * run on the context's stream underneath the host's transcript work.  h_steps MUST be page-locked (zigz_host_register)
 * and stay unmodified until the stream has passed the copy (zigz_commit_roots / zigz_ctx_synchronize). */
zigz_status zigz_dev_witness_from_steps_async(zigz_ctx *ctx, const zigz_trace_step *h_steps, size_t num_steps, size_t nv,
                                              const uint64_t *initial_regs, uint32_t *d_cols, size_t col_stride);
/* The asynchronous form into a column buffer the CONTEXT owns (a workspace, reused call after call: nothing is allocated
 * on the steady-state path of a service that proves trace after trace on a pool of contexts).  *d_cols / *col_stride are
 * valid until the next call of this function, zigz_commit_begin (host columns), zigz_bench_kernel or
 * zigz_ctx_release_workspaces on the context -- in particular through a commit job begun on them. */
zigz_status zigz_dev_witness_from_steps_ws(zigz_ctx *ctx, const zigz_trace_step *h_steps, size_t num_steps, size_t nv,
                                           const uint64_t *initial_regs, const uint32_t **d_cols, size_t *col_stride);
/* The 32-BYTE record (what crosses PCIe per step decides the rate of a service whose traces arrive from the host: 48 B per step
 * are 50 MB per 2^20 proof).  Two observations about src/vm/trace.zig:73-97: instruction.imm is a sign-extended field of at
 * most 32 bits in every RV64IM format (src/isa/rv64i.zig:156-233), and memory_access is null on every step that is not a
 * LOAD / STORE.  So: imm as i32, and the memory triple in a SIDE LIST of zigz_mem_access (16 B per access) that a step refers
 * to by index -- ZIGZ_NO_MEM_ACCESS on the steps without one (an index >= num_mem reads as "none" as well).  mem_is_read
 * stays in the record.  A kernel widens the records on the device (zigz_trace_step, above) and the same expansion follows:
 * identical columns (tests/test_gpu_parity.py: both records against the numpy restatement and the oracle). */
typedef struct zigz_trace_step32 {
    uint64_t pc;        /* step.pc */
    uint64_t rd_value;  /* regs_after[wr_reg]; ignored when wr_reg == 0 */
    int32_t imm;        /* instruction.imm (sign-extended to i64 on the device) */
    uint32_t mem_index; /* index of this step's access in the side list, or ZIGZ_NO_MEM_ACCESS */
    uint8_t opcode, rd, rs1, rs2, funct3, funct7;
    uint8_t wr_reg;      /* 1..31: the register this step wrote; 0 (or any value >= 32): none */
    uint8_t mem_is_read; /* 1 = load, 0 = store or no access */
} zigz_trace_step32;
typedef struct zigz_mem_access {
    uint64_t addr, value; /* memory_access.address / .value */
} zigz_mem_access;
#define ZIGZ_NO_MEM_ACCESS 4294967295
/* as zigz_dev_witness_from_steps (waits for the copies) */
zigz_status zigz_dev_witness_from_steps32(zigz_ctx *ctx, const zigz_trace_step32 *h_steps, size_t num_steps,
                                          const zigz_mem_access *h_mem, size_t num_mem, size_t nv, const uint64_t *initial_regs,
                                          uint32_t *d_cols, size_t col_stride);
/* as zigz_dev_witness_from_steps_ws (asynchronous, into the context's own column buffer; page-locked records) */
zigz_status zigz_dev_witness_from_steps32_ws(zigz_ctx *ctx, const zigz_trace_step32 *h_steps, size_t num_steps,
                                             const zigz_mem_access *h_mem, size_t num_mem, size_t nv, const uint64_t *initial_regs,
                                             const uint32_t **d_cols, size_t *col_stride);
/* The 16-BYTE record.  One more observation about src/vm/trace.zig:73-97: the seven instruction fields of a step (opcode, rd, rs1,
 * rs2, funct3, funct7, imm) are the decoded instruction AT step.pc (src/vm/state.zig:128-140: fetch, decode, execute) -- a
 * program of P instructions has P different field tuples however long it runs.  So the fields travel ONCE per instruction, in a
 * code table indexed by (pc - code_base) / 4, and a step carries where it was, what it wrote and which access it made:
 *   pc_word : (pc - code_base) with mem_is_read in bit 0 (pc - code_base is a multiple of 4 below 2^32: RV64IM has no 16-bit forms)
 *   mem_wr  : bits 0-26 the index of the step's access in the side list (ZIGZ_NO_MEM_ACCESS16 = none; an index >= num_mem reads as
 *             none as well), bits 27-31 wr_reg
 * A caller whose trace does not fit -- a pc that is not 4-aligned or 2^32 past the base, the same pc executed with two different
 * decodings (a program that rewrote itself), 2^27 - 1 or more accesses -- uses the 32-byte record (the host mirror does: host/capi.cpp).
 * A step whose pc_word points past the table gets zero fields.  The kernel widens to zigz_trace_step and the same expansion
 * follows: identical columns (tests/test_gpu_parity.py).  16.8 MB instead of 33.6 cross PCIe for a 2^20 trace of a loop. */
typedef struct zigz_trace_step16 {
    uint32_t pc_word;
    uint32_t mem_wr;
    uint64_t rd_value;  /* regs_after[wr_reg]; ignored when wr_reg == 0 */
} zigz_trace_step16;
typedef struct zigz_code_entry {
    int32_t imm;        /* instruction.imm (sign-extended to i64 on the device) */
    uint8_t opcode, rd, rs1, rs2, funct3, funct7;
    uint8_t reserved0, reserved1; /* 0 */
} zigz_code_entry;
#define ZIGZ_NO_MEM_ACCESS16 134217727
/* as zigz_dev_witness_from_steps32 (waits for the copies) */
zigz_status zigz_dev_witness_from_steps16(zigz_ctx *ctx, const zigz_trace_step16 *h_steps, size_t num_steps,
                                          const zigz_mem_access *h_mem, size_t num_mem, uint64_t code_base,
                                          const zigz_code_entry *h_code, size_t num_code, size_t nv, const uint64_t *initial_regs,
                                          uint32_t *d_cols, size_t col_stride);
/* as zigz_dev_witness_from_steps32_ws (asynchronous, into the context's own column buffer; page-locked records, side list and table) */
zigz_status zigz_dev_witness_from_steps16_ws(zigz_ctx *ctx, const zigz_trace_step16 *h_steps, size_t num_steps,
                                             const zigz_mem_access *h_mem, size_t num_mem, uint64_t code_base,
                                             const zigz_code_entry *h_code, size_t num_code, size_t nv, const uint64_t *initial_regs,
                                             const uint32_t **d_cols, size_t *col_stride);
/* Page-lock a host buffer the caller reuses for uploads (trace records, witness columns): H2D copies from registered
 * memory run at PCIe rate without the staging copy.  zigz_host_unregister before freeing the buffer; page-locking belongs
 * to the process, so unregister accepts ctx == NULL (the registering context may already be destroyed). */
zigz_status zigz_host_register(zigz_ctx *ctx, void *h_ptr, size_t bytes);
zigz_status zigz_host_unregister(zigz_ctx *ctx, void *h_ptr);

/* ---------------------------------------------------------------- device-resident MLE / sumcheck */
zigz_status zigz_dev_mle_bind(zigz_ctx *ctx, const uint32_t *d_in, size_t n, uint64_t r, uint32_t *d_out);
/* fused: d_out = bind(d_in, r) and sums[0..1] = (sum of low half, sum of high half) of d_out */
zigz_status zigz_dev_mle_bind_sums(zigz_ctx *ctx, const uint32_t *d_in, size_t n, uint64_t r, uint32_t *d_out,
                                   uint64_t half_sums[2]);
zigz_status zigz_dev_mle_half_sums(zigz_ctx *ctx, const uint32_t *d_in, size_t n, uint64_t half_sums[2]);
zigz_status zigz_dev_mle_eval(zigz_ctx *ctx, const uint32_t *d_in, size_t n, const uint64_t *point,
                              size_t point_len, uint64_t *out);
/* d_scratch: n/2 + n/4 ... < n elements of scratch (may be NULL: taken from the context workspace) */
zigz_status zigz_dev_sumcheck_prove(zigz_ctx *ctx, const uint32_t *d_in, size_t n, uint32_t *d_scratch,
                                    const uint64_t *fixed_challenges, uint64_t *rounds, uint64_t *point,
                                    uint64_t *final_eval);

/* ---------------------------------------------------------------- one sumcheck over several GPUs (row sharding)
 * SumcheckProver.prove (src/proofs/sumcheck_prover.zig:26-91) of ONE table of n = n_local * world elements, element i on
 * rank i mod world at local index i / world (so every MSB-first bind pair of partialEval, multilinear.zig:166-173, is
 * rank-local).  Radix form: per stage of k <= 10 rounds the ranks exchange 2^k exact u64 partial block sums, then one
 * exchange re-assembles the last <= 1024 * world entries: 2-3 exchanges per proof, all through `allgather` (every rank
 * contributes `bytes` from `send`; `recv` gets world * bytes in rank order; return 0 on success -- bind it to RCCL,
 * MPI or torch.distributed).  Every rank returns the same rounds / point / final_eval as the unsharded prover.
 * Failure is collective: every payload carries one status word, so a rank whose local pass failed (HIP error, a callback's
 * status) takes part in the next exchange with that status and ALL ranks stop there -- the failing rank returns its own
 * error, the others ZIGZ_ERR_COMM -- instead of waiting for the transport's timeout; a hook that returns nonzero is
 * ZIGZ_ERR_COMM as well. */
typedef int (*zigz_allgather_fn)(void *user, const void *send, size_t bytes, void *recv);
zigz_status zigz_dev_sumcheck_prove_sharded(zigz_ctx *ctx, const uint32_t *d_local, size_t n_local, int rank, int world,
                                            zigz_allgather_fn allgather, void *user, uint64_t *rounds, uint64_t *point,
                                            uint64_t *final_eval);
/* Built-in zigz_allgather_fn for the ranks of ONE node: a mailbox in POSIX shared memory.  What the sharded provers
 * exchange is host-resident (it was read back for the SHA3 transcript) and <= 64 KiB, so between processes of a node
 * this is the shortest path (a few microseconds per exchange); across nodes bind the hook to RCCL / MPI.  Every rank
 * calls create with the same job-unique `name` (<= 80 chars, no '/'), world and max_bytes (largest `bytes` of any
 * exchange); rank 0 creates the segment, the others attach (waiting up to timeout_s; <= 0: 60 s).  Creation is a
 * collective with a handshake: an attacher accepts a segment only once the live rank 0 has echoed the random token it
 * wrote there (a segment left behind by a crashed job under the same name has nobody to answer and is let go), and rank 0
 * returns when every rank has been acknowledged.  `name` must still be unique among jobs that START at the same time: put a
 * job id or a nonce broadcast by rank 0 into it.  Pass the comm as the hook's `user`.  A rank that waits longer than
 * timeout_s inside an exchange returns nonzero instead of hanging. */
typedef struct zigz_shm_comm zigz_shm_comm;
zigz_status zigz_shm_comm_create(const char *name, int rank, int world, size_t max_bytes, double timeout_s,
                                 zigz_shm_comm **out);
int zigz_shm_allgather(void *comm, const void *send, size_t bytes, void *recv);
void zigz_shm_comm_destroy(zigz_shm_comm *comm);
/* RCCL as the transport, without torch: for ranks on different GPUs of an xGMI node (or on several nodes).  The 128-byte
 * unique id comes from ONE rank (zigz_rccl_unique_id) and reaches the others however the host distributes small blobs
 * (a file, MPI, its own socket); every rank then creates its communicator for HIP device `device`.  max_bytes: the largest
 * `bytes` of any exchange.  librccl.so is loaded on first use; ZIGZ_ERR_NO_DEVICE when it cannot be.
 *  - zigz_rccl_allgather is a zigz_allgather_fn (user = the comm): host buffers, staged through pinned + device memory;
 *  - zigz_rccl_allreduce_u64 sums n u64 words over the ranks (host buffers); ..._dev does it in place on words already in
 *    HBM on the given hipStream_t, no staging and no synchronisation;
 *  - zigz_dev_sumcheck_prove_rccl is zigz_dev_sumcheck_prove_sharded with the partial block sums of every radix stage
 *    (k <= 10 rounds of round-polynomial sums, sumcheck_prover.zig:50-77) all-reduced in HBM on the context's stream --
 *    one RCCL all-reduce per stage of rounds -- and the last <= 1024 * world entries all-gathered through the comm.
 * RCCL itself never times out, so every wait behind a collective here has a deadline (zigz_rccl_comm_set_timeout, default
 * 120 s): when it passes -- a peer never entered the collective -- the communicator is aborted (ncclCommAbort), the call
 * returns nonzero / ZIGZ_ERR_COMM and every later call on that communicator fails at once.  Failure of a sharded proof is
 * collective here too: the all-reduce of a stage carries one extra word, the number of ranks whose local pass failed, so
 * all ranks leave the proof at the same collective (the failing one with its own error, the others with ZIGZ_ERR_COMM).
 * zigz_rccl_stream_wait is the deadline wait for a stream that holds a ..._dev collective; zigz_rccl_comm_abort aborts by
 * hand.  (ncclCommInitRank inside zigz_rccl_comm_create has no deadline: a rank that never calls create blocks the others.) */
#define ZIGZ_RCCL_UNIQUE_ID_BYTES 128
typedef struct zigz_rccl_comm zigz_rccl_comm;
zigz_status zigz_rccl_unique_id(uint8_t id[ZIGZ_RCCL_UNIQUE_ID_BYTES]);
zigz_status zigz_rccl_comm_create(int device, const uint8_t id[ZIGZ_RCCL_UNIQUE_ID_BYTES], int rank, int world,
                                  size_t max_bytes, zigz_rccl_comm **out);
int zigz_rccl_allgather(void *comm, const void *send, size_t bytes, void *recv);
int zigz_rccl_allreduce_u64(zigz_rccl_comm *comm, const uint64_t *send, size_t n, uint64_t *recv);
int zigz_rccl_allreduce_u64_dev(zigz_rccl_comm *comm, uint64_t *d_words, size_t n, void *hip_stream);
int zigz_rccl_stream_wait(zigz_rccl_comm *comm, void *hip_stream);
void zigz_rccl_comm_set_timeout(zigz_rccl_comm *comm, double seconds);
void zigz_rccl_comm_abort(zigz_rccl_comm *comm);
int zigz_rccl_comm_rank(const zigz_rccl_comm *comm);
int zigz_rccl_comm_world(const zigz_rccl_comm *comm);
void zigz_rccl_comm_destroy(zigz_rccl_comm *comm);
zigz_status zigz_dev_sumcheck_prove_rccl(zigz_ctx *ctx, const uint32_t *d_local, size_t n_local, zigz_rccl_comm *comm,
                                         uint64_t *rounds, uint64_t *point, uint64_t *final_eval);
/* The same orchestration over caller-supplied data passes on the local table (what the GPU passes of the call above do):
 * block_sums: exact u64 sums of the 2^k contiguous blocks of the current table; fold: current := sum_b w[b] *
 * current[b*m + i] (2^k canonical weights, m = length / 2^k) and, when k_next != 0, the 2^k_next block sums of the result;
 * read_tail: the current table (m canonical values).  Each returns a zigz_status.  No GPU is touched. */
typedef struct zigz_radix_ops {
    void *user;
    zigz_status (*block_sums)(void *user, unsigned k, uint64_t *sums);
    zigz_status (*fold)(void *user, unsigned k, const uint64_t *weights, unsigned k_next, uint64_t *next_sums);
    zigz_status (*read_tail)(void *user, size_t m, uint64_t *out);
} zigz_radix_ops;
zigz_status zigz_sumcheck_radix_run(const zigz_radix_ops *ops, size_t n_local, int rank, int world,
                                    zigz_allgather_fn allgather, void *comm_user, const uint64_t *fixed_challenges,
                                    uint64_t *rounds, uint64_t *point, uint64_t *final_eval);
/* ... over data passes whose sums are ALREADY the sums over all ranks (reduced inside the pass by a collective of its own,
 * as the passes of zigz_dev_sumcheck_prove_rccl do with RCCL): no exchange per stage.  A pass that fails -- or learns inside
 * its collective that a peer failed -- returns nonzero on every rank at the same stage; the run then skips the remaining
 * passes everywhere and all ranks leave through the tail exchange, which carries the status words. */
zigz_status zigz_sumcheck_radix_run_reduced(const zigz_radix_ops *ops, size_t n_local, int rank, int world,
                                            zigz_allgather_fn allgather, void *comm_user, const uint64_t *fixed_challenges,
                                            uint64_t *rounds, uint64_t *point, uint64_t *final_eval);

/* ---------------------------------------------------------------- host SHA3 sponge / transcript
 * FiatShamirTranscript   src/core/hash.zig:255-324 (sequential by construction: stays on the host) */
zigz_transcript *zigz_transcript_new(void);
void zigz_transcript_free(zigz_transcript *t);
void zigz_transcript_append_bytes(zigz_transcript *t, const uint8_t *data, size_t len);
void zigz_transcript_append_field(zigz_transcript *t, uint64_t canonical_value);
/* appends `count` times:  tag bytes then LE64((start + k) mod p), k = 0..count-1
 * (the "LASSO_TABLE" loop of prover.zig:302-312 in one call) */
void zigz_transcript_append_tagged_counter(zigz_transcript *t, const uint8_t *tag, size_t tag_len,
                                           uint64_t start, uint64_t count);
uint64_t zigz_transcript_challenge(zigz_transcript *t); /* BabyBear; hash.zig:301-316 */
void zigz_sha3_256(const uint8_t *data, size_t len, uint8_t out[32]);
void zigz_sha256(const uint8_t *data, size_t len, uint8_t out[32]);
/* which single-state Keccak-f[1600] the host sponge uses ("scalar", "bmi2", "avx512f" or "avx512vl": the fastest
 * supported variant, timed at load; env ZIGZ_HOST_KECCAK overrides), and one permutation through a chosen variant
 * (0 = the picked one, 1 = scalar, 2 = bmi2, 3 = avx512f, 4 = avx512vl) -- diagnostics / self-test */
const char *zigz_host_keccak_impl(void);
void zigz_host_keccak_permute(uint64_t state[25], int which);
/* Sponge service for a host that proves several traces at once (one thread per proof).  A proof's transcript is a chain of
 * ~147 k dependent permutations at a 2^20 trace (prover.zig:302-312) and one permutation costs the same ~880 cycles whether
 * its lanes sit in xmm or zmm registers, so `n` server threads each advance up to 8 transcripts' tagged-counter absorptions
 * in lock step -- one 8-way AVX-512 permutation per block -- while the proofs' own threads sleep.  Same bytes absorbed, same
 * challenges.  n = 0 (default): every transcript absorbs on its own thread.  Ignored without AVX-512F.  Call while no
 * transcript is absorbing.  zigz_host_keccak_permute_x8: the 8-way permutation on states[lane][slot] (25 x 8 u64, 64-byte
 * aligned) -- diagnostics / self-test, requires AVX-512F. */
void zigz_host_sponge_servers(int n);
int zigz_host_sponge_batching(void);
void zigz_host_keccak_permute_x8(uint64_t *states);

/* ---------------------------------------------------------------- measurement hooks (bench.py)
 * Average device time (microseconds, HIP events on the context's stream) of the last call's
 * dominant kernels; reset by each API call that records them. */
typedef struct zigz_kernel_stats {
    double merkle_build_us;   /* all Keccak leaf + level launches of the last commit_begin */
    double eval_us;           /* all MLE fold / eval launches of the last commit_open_all */
    double path_us;           /* path gather */
    double bind_us;           /* last zigz_dev_mle_bind / bind_sums launch, or all binds of the last sumcheck */
    uint64_t bind_launches;
    uint64_t keccak_permutations; /* Keccak-f[1600] actually computed by the last commit (table look-ups and chained uniform blocks excluded) */
    /* the bulk MLE-bind launches of the last eval / sumcheck / bind call, each timed on its own: total device time,
     * launch count, algorithmic bytes.  Kernel k_radix_fold (one pass binding v-10 variables, 4 B read per element;
     * timed with the dispatch's own begin/end timestamps) for evals of tables >= 2^14, k_bind_vec (6 B per table
     * element; HIP event pair around the launch) otherwise. */
    double bind_vec_us;
    uint64_t bind_vec_launches;
    uint64_t bind_vec_bytes;
    /* run-aware Merkle levels (option "run_aware_mask") of the last batched commit: columns built that way, the nodes
     * of the levels they covered (levels 0 .. v - 8: what a dense build hashes there), how many of those were hashed rather
     * than copied from the left neighbour, and the kernel time of the structure + hashing launches (timing mode).  0 when the
     * option is off or the trees have fewer than 2^15 leaves. */
    uint64_t run_aware_columns;
    uint64_t run_aware_dense_nodes;
    uint64_t run_aware_hashed;
    double run_aware_us;
    /* content-addressed levels (option "cons_group_mask") of the last batched commit: columns of the group, the nodes of the
     * levels they covered (what a dense build hashes there) and the digests actually computed (representatives x columns) */
    uint64_t cons_columns;
    uint64_t cons_dense_nodes;
    uint64_t cons_hashed;
    uint64_t cons_probe_distinct; /* distinct leaves (tuples) of the group; > 1/4 of the leaves: dropped, cons_columns = 0 (0 when the group was not tried) */
    /* Keccak launches of the last batched commit by class, each launch timed with its own begin / end timestamps
     * (kernel time as rocprofv3 --kernel-trace reports it; the gaps between launches are in merkle_build_us only):
     * k_keccak_leaves; k_keccak_level<4> (the large levels); k_keccak_level<1> (the small levels); hashes = permutations */
    double keccak_leaves_us;
    uint64_t keccak_leaves_perms;
    double keccak_level_wide_us;
    uint64_t keccak_level_wide_perms;
    double keccak_level_small_us;
    uint64_t keccak_level_small_perms;
    /* option "small_domain_mask": columns whose levels 0-1 came from the constant tables in the last batched commit, the
     * kernel time of that launch, and the number of waves that found a value >= 128 and hashed instead */
    uint64_t small_domain_columns;
    double small_domain_us;
    uint64_t small_domain_fallback_waves;
    /* the structure-aware build of the last batched commit (timing mode), by class: structure_us = the passes that decide
     * WHICH nodes are hashed (run-aware stages k_runs_stage, content-addressing table passes k_cons_*: no hashing);
     * list_hash_us / list_hash_perms = the per-level launches that hash those lists (k_level_hash) and the permutations they
     * computed; top_us / top_perms = the last 8 levels (k_merkle_top).  run_aware_us = structure_us + list_hash_us. */
    double structure_us;
    double list_hash_us;
    uint64_t list_hash_perms;
    double top_us;
    uint64_t top_perms;
    /* builds that zigz_commit_roots had to repeat on this context because a list of the structure-aware levels ran out of the
     * room learnt from earlier builds (or the group was dropped and its columns had no slabs): a running count */
    uint64_t rebuilds;
    /* hinted (run-aware) columns of the last commit job whose N values are all equal -- established from the values by the
     * structure pass -- and which zigz_commit_open_all therefore did not read again: the multilinear extension of a constant is
     * that constant */
    uint64_t eval_constant_columns;
} zigz_kernel_stats;
/* One hot kernel, `iters` (<= 64) launches on a synthetic device-resident table of ncols columns x 2^nv elements, each
 * launch timed by its own begin / end timestamps.  kernel: "k_bind_vec" (partialEval, multilinear.zig:154-180, 6 B per
 * table element), "k_bind_vec_sums" (the same fused with the next roundPolynomial), "k_half_sums" / "k_block_sums"
 * (roundPolynomial / sumOverHypercube, multilinear.zig:188-232: kernel k_block_sums with 2 resp. 1024 blocks, 4 B per
 * element; "k_block_sums": ncols = 1),
 * "k_radix_fold" (eval, multilinear.zig:110-144, 4 B per element + partial sums), "k_keccak_leaves" / "k_keccak_level"
 * (hash.zig:135-147,187-195; units = permutations), "k_lasso_fingerprints" (lasso_prover.zig:208-239; units = rows of 3
 * fields).  cold != 0: a 1 GiB read sweep before every launch empties L2 / Infinity Cache of the table. */
typedef struct zigz_bench_result {
    double avg_us, min_us, max_us;
    uint64_t algorithmic_bytes; /* per launch */
    uint64_t units;             /* table elements / permutations / rows per launch */
    uint32_t launches;
} zigz_bench_result;
zigz_status zigz_bench_kernel(zigz_ctx *ctx, const char *kernel, size_t nv, size_t ncols, int iters, int cold,
                              zigz_bench_result *out);
zigz_status zigz_ctx_enable_timing(zigz_ctx *ctx, int enable);
/* Timing mode, launch by launch: the timed launches of the context's last commit job (its build after zigz_commit_roots, its
 * eval fold after zigz_commit_open_all) with their own begin / end on ONE time axis -- microseconds since the epoch of
 * `zigz_ctx_set_epoch` -- so that a caller that runs many contexts at once can take the UNION of a kernel class's intervals
 * (time during which at least one launch of the class was on the GPU) instead of the sum of durations of launches that
 * overlap.  The events carry the dispatch's own timestamps (first wave started .. last wave ended: within 1 us of a clock
 * read inside the kernel, tools/event_semantics.hip; a launch's wait behind its predecessor in the stream is not part of it,
 * its wait for wave slots next to other streams' kernels is), and are comparable across the streams of a device.
 * zigz_ctx_set_epoch(ctx, owner): owner == ctx records a new epoch on the context's stream; otherwise ctx adopts the
 * owner's (which must outlive its use).  Without an epoch, times count from the job's first timed launch.
 * class: 0 k_keccak_leaves, 1 / 2 k_keccak_level (wide / small), 3 k_keccak_small_l01, 4 structure passes (k_runs_stage,
 * k_cons_*: first .. last launch of each group), 5 k_level_hash, 6 k_merkle_top, 7 k_radix_fold (eval). */
typedef struct zigz_launch_rec {
    uint32_t cls;
    uint32_t reserved;
    uint64_t perms;      /* Keccak permutations where the host knows them at launch time (classes 0-2, 6), else 0 */
    double start_us, end_us;
} zigz_launch_rec;
zigz_status zigz_ctx_set_epoch(zigz_ctx *ctx, zigz_ctx *owner);
zigz_status zigz_ctx_launch_log(zigz_ctx *ctx, zigz_launch_rec *out, size_t cap, size_t *n);
/* tuning / test switches: "per_round_sumcheck" = 1 forces the one-launch-per-round sumcheck form;
 * "fold_eval" = 1 forces eval by v successive binds instead of the one-pass radix form;
 * "run_aware_mask" = bit c set: column c of the following batched commits (<= 64 columns, 2^15 .. 2^26 rows) is expected to be
 *   piecewise constant -- in the witness of prover.zig:376-390 the registers: every RV64IM step writes at most one of them
 *   (state.zig writeReg), so the 31 columns x1..x31 together change at most once per step.  On the levels 0 .. v - 8 (down to
 *   256 nodes per column) a node whose subtree and its left neighbour's are uniform with the same value takes the
 *   neighbour's digest instead of being hashed.  Decided from the VALUES on the device, never from the hint: identical trees
 *   for ANY input; with c change points in a column a level costs <= min(nodes, 2 c + one per tile) hashes (a tile is what one
 *   4096-leaf segment covers at that level).
 *   In a commit job only the hashed nodes have a digest at all, stored in LIST ORDER; the next level's hashes and
 *   zigz_commit_open_all find a copy's digest through a per-level bitmap of hashed nodes.  The room for those lists is what the
 *   context's earlier jobs needed (+ 25 %), not the worst case: a job that runs out notices on the device and
 *   zigz_commit_roots repeats its build with more (kernel_stats.rebuilds) -- typically once, the first time a context meets
 *   a new kind of trace.  "run_aware_materialize" = 1 writes every digest of every tree into node-addressed trees (tests that
 *   compare whole trees); single trees (zigz_merkle_commit) always do.
 * "cons_group_mask" = bit c set: the columns of this set repeat in the same places -- in the witness of prover.zig:376-390
 *   the ten columns that are functions of the instruction at pc (pc, x0, opcode, rd, rs1, rs2, funct3, funct7, imm, is_read):
 *   wherever the program loops, the same nodes recur in all of them, at most P distinct ones per level for a loop of P steps.
 *   On the levels 0 .. v - 8 a device hash table finds for every node the first node of its level with the same content in
 *   ALL columns of the group (leaves: the tuple of values, fingerprinted and verified; above: the pair of the children's
 *   representatives -- the identity of the hash input itself), and only representatives are hashed, once per column.  Takes
 *   precedence over the two other hints; identical trees for ANY input; the other nodes are virtual like run-aware copies.
 *   A group that does not repeat -- more than a quarter of its leaves distinct -- is DROPPED: its columns are built like any
 *   others (the other hints then apply to them).  That decision is taken on the device after the leaf level's table pass and
 *   read by the later launches from device memory: zigz_commit_begin* never waits for it.  (The first job of a context that
 *   drops its group is built twice -- the columns had no node-addressed trees to be built into; a context whose last two
 *   jobs dropped the group does not try it in its next 15 jobs; "cons_always" = 1 tries in every job.)
 * "merkle_dedup" = 1 / 0 is shorthand for run_aware_mask = all ones / 0;
 * "small_domain_mask" = bit c set: column c of the following batched commits (<= 64 columns, >= 1024 rows) holds values
 *   < 128 BY CONSTRUCTION -- in the witness of prover.zig:376-390 that is x0 (always 0, registers.zig:38-48), the
 *   instruction fields opcode / rd / rs1 / rs2 / funct3 / funct7 (7-, 5-, 5-, 5-, 3-, 7-bit fields, rv64i.zig:124-151) and
 *   mem.is_read (0 / 1) -- so its leaf digests SHA3(LE64(v)) and level-1 nodes are looked up in two constant tables
 *   (128 + 128^2 digests) instead of hashed: 1.5 N of the 2 N permutations of such a column.  The bound is checked per
 *   wave on the device; where it does not hold the digests are hashed, so the trees are identical for ANY input. */
zigz_status zigz_ctx_set_option(zigz_ctx *ctx, const char *name, int64_t value);
/* the current value of an option (every name above except the shorthand "merkle_dedup"): lets a caller that changes the
 * hint masks for one proof put back what the context's owner had set (host/prover.cpp does) */
zigz_status zigz_ctx_get_option(zigz_ctx *ctx, const char *name, int64_t *value);
zigz_status zigz_ctx_get_stats(zigz_ctx *ctx, zigz_kernel_stats *out);

#ifdef __cplusplus
}
#endif
#endif
