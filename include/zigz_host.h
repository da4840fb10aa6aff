/*
 * zigz_host.h -- C face of libzigz_host.so, the C++ mirror of the Zig host (zigz_amd/csrc/host/).
 *
 * NOT the drop-in boundary (that is zigz_hip.h): this exists so that tests, bench.py and
 * __graft_entry__ can drive the host mirror -- Prover.prove / Verifier.verify / BinarySerializer /
 * VMState / WitnessGenerator / SumcheckProver / LassoProver with the reference's semantics -- from
 * Python through ctypes.  Status codes: zigz_status values (zigz_hip.h) plus the host-only codes
 * 19..31 (VM and serializer errors, same numbering as the oracle).  zigzh_last_error() returns the
 * Zig-style error name of the last failure on the calling thread.
 */
#ifndef ZIGZ_HOST_H
#define ZIGZ_HOST_H
#include "zigz_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct zigzh_trace zigzh_trace; /* an executed program: compact trace records + public IO */

const char *zigzh_last_error(void);
void zigzh_free(void *p);
/* wall-clock seconds of the phases of the last zigzh_prove_trace[_slots] on this thread: commit_begin, sumcheck transcript,
 * lasso transcript, wait for roots, roots+challenges, open_all, packaging, serialize, wait for a GPU slot, time in the slot */
void zigzh_last_timings(double out[10]);

/* Prover(F).prove(program, entry_pc, initial_regs, max_steps, null, input) + BinarySerializer.serialize
 * src/prover/prover.zig:73-226, src/prover/serialization.zig:70-97 */
int zigzh_prove(zigz_ctx *ctx, const uint8_t *program, size_t program_len, uint64_t entry_pc,
                const uint64_t *initial_regs, size_t n_initial_regs, int has_initial_regs, size_t max_steps,
                const uint64_t *input, size_t n_input, uint8_t **proof_out, size_t *proof_len, size_t *num_steps);
/* BinarySerializer.deserialize + Verifier.verify   src/verifier/verifier.zig:49-91; result = VerificationResult */
int zigzh_verify(const uint8_t *proof, size_t proof_len, const uint8_t *program, size_t program_len, int *result);
/* deserialize then serialize again (round-trip, tests/integration_tests.zig:90-127) */
int zigzh_reserialize(const uint8_t *proof, size_t proof_len, uint8_t **out, size_t *out_len);

/* [1/6] of Prover.prove: run the VM exactly as prover.zig:117-142 does */
int zigzh_execute(const uint8_t *program, size_t program_len, uint64_t entry_pc, const uint64_t *initial_regs,
                  size_t n_initial_regs, int has_initial_regs, size_t max_steps, const uint64_t *input, size_t n_input,
                  zigzh_trace **out);
void zigzh_trace_free(zigzh_trace *t);
size_t zigzh_trace_num_steps(const zigzh_trace *t);
size_t zigzh_trace_num_vars(const zigzh_trace *t);
size_t zigzh_trace_num_lookups(const zigzh_trace *t);
const uint64_t *zigzh_trace_rows(const zigzh_trace *t); /* [num_steps][43] raw u64 (expanded on first use) */
/* the trace as recorded: num_steps compact records (zigz_trace_step, zigz_hip.h) + the register file before step 0 */
const void *zigzh_trace_steps(const zigzh_trace *t);
const uint64_t *zigzh_trace_initial_regs(const zigzh_trace *t);
/* page-lock the records for repeated uploads (zigz_host_register); released by zigzh_trace_free */
int zigzh_trace_pin(zigzh_trace *t, zigz_ctx *ctx);
/* which record a service uploads for this trace once it is pinned -- 16 (zigz_trace_step16 + code table: the trace fits the form),
 * 32 (zigz_trace_step32) or 48 bytes per step -- and (optional) the bytes that cross PCIe per proof: records + side list + table */
int zigzh_trace_upload_form(const zigzh_trace *t, size_t *bytes);
/* WitnessGenerator.generate -> 43 columns of 2^nv canonical u64, column-major   witness.zig:29-61 */
int zigzh_trace_witness(const zigzh_trace *t, uint64_t *cols_out);
/* builds the 43 witness columns directly in HBM (packed u32, column stride `stride` elements) */
int zigzh_trace_witness_dev(const zigzh_trace *t, zigz_ctx *ctx, uint32_t *d_cols, size_t stride);
/* the same without waiting for the copy (the trace must be pinned on this context: zigzh_trace_pin); the upload and
 * the witness kernels then run underneath the transcript work of the zigzh_prove_trace call that follows */
int zigzh_trace_witness_dev_async(const zigzh_trace *t, zigz_ctx *ctx, uint32_t *d_cols, size_t stride);
/* steps [4/6]..[6/6] + packagePublicIO for an executed trace; d_cols == NULL: host witness is generated and
 * uploaded; otherwise the resident columns are used.  want_bytes: 0 = proof struct only; 1 = also serialize ("ZIGZ"
 * v1) into a malloc'd buffer the caller frees with zigzh_free; 2 = serialize with the early sections written on a
 * helper thread underneath the transcript, into a thread-local buffer the caller BORROWS (valid until the next
 * call on this thread; do not free). */
int zigzh_prove_trace(const zigzh_trace *t, zigz_ctx *ctx, const uint32_t *d_cols, size_t stride, int want_bytes,
                      uint8_t **proof_out, size_t *proof_len);

/* GPU slots of a proving service (zigz_host.hpp: GpuSlots): k contexts (stream + workspaces each) shared by any number of
 * proving threads.  zigzh_prove_trace_slots is zigzh_prove_trace(want_bytes = 2) for such a thread: the transcript of
 * steps [4/6]-[5/6] runs holding nothing on the GPU, then the thread takes a slot for begin -> roots -> challenges ->
 * open_all -> end (the trees do not depend on the transcript, the points do: prover.zig:405-424) and gives it back.
 * d_cols: the 43 resident columns (any allocation of the process on that device), or NULL: the compact trace records
 * (pin them: zigzh_trace_pin) are uploaded and expanded inside the slot, into a column buffer the slot owns -- a proof in
 * flight then holds no HBM at all outside its few milliseconds in a slot.  stats_out / log_out (optional): the kernel
 * statistics and, in timing mode, the launch log (zigz_ctx_launch_log) of the context the proof ran on.
 * zigzh_slots_ctx: the i-th context, for set-up (options, timing, epochs) while nobody proves; zigzh_slots_acquire /
 * _release: a slot for work of the caller's own (uploads, allocations). */
typedef struct zigzh_slots zigzh_slots;
int zigzh_slots_create(int device, size_t k, zigzh_slots **out);
/* measurement: the commit path alone -- slot, begin on 43 resident columns, roots, open_all at `points` (43 x nv), end -- `reps`
 * times back to back inside the library; masks = {small_domain_mask, run_aware_mask, cons_group_mask} for the jobs */
int zigzh_commit_path_repeat(zigzh_slots *s, const uint32_t *d_cols, size_t stride, size_t nv, const uint64_t *points,
                             const int64_t masks[3], size_t reps);
/* small traces (2^nv rows, nv <= max_nv <= 18): proofs that reach their GPU phase within linger_us of each other share ONE
 * commit job of up to max_batch (<= 32) proofs (zigz_commit_begin_batch; zigz_host.hpp: GpuBatcher) -- each proof keeps its
 * own transcript and gets exactly the roots and openings a job of its own would give.  max_batch <= 1: off (the default).
 * Set while nobody proves. */
int zigzh_slots_set_batching(zigzh_slots *s, unsigned max_batch, double linger_us, size_t max_nv);
void zigzh_slots_destroy(zigzh_slots *s);
size_t zigzh_slots_size(const zigzh_slots *s);
zigz_ctx *zigzh_slots_ctx(zigzh_slots *s, size_t i);
zigz_ctx *zigzh_slots_acquire(zigzh_slots *s);
void zigzh_slots_release(zigzh_slots *s, zigz_ctx *ctx);
int zigzh_prove_trace_slots(const zigzh_trace *t, zigzh_slots *s, const uint32_t *d_cols, size_t stride,
                            uint8_t **proof_out, size_t *proof_len, zigz_kernel_stats *stats_out, zigz_launch_rec *log_out,
                            size_t log_cap, size_t *log_n);

/* `reps` proofs of the same trace back to back on the calling thread (a lane of a service), without returning to the caller
 * in between; stats_sum / timings_sum (optional): field-wise sums over the proofs; the proof is the last one's (borrowed). */
int zigzh_prove_trace_slots_repeat(const zigzh_trace *t, zigzh_slots *s, const uint32_t *d_cols, size_t stride, size_t reps,
                                   uint8_t **proof_out, size_t *proof_len, zigz_kernel_stats *stats_sum, double timings_sum[10]);
void zigzh_stats_add(zigz_kernel_stats *a, const zigz_kernel_stats *b);

/* One proof over `world` GPUs, sharded by column: every rank calls this with the SAME trace and its own
 * context / resident copy of the 43 columns; rank r builds, commits and opens only its contiguous block of columns
 * (43 over 8 -> 6,6,6,5,5,5,5,5).  The two exchanges of Prover.generateCommitments (prover.zig:366-467) -- 43 roots
 * before the transcript absorbs them, 43 openings after -- go through `allgather`: every rank contributes `bytes` from
 * `send`; `recv` gets world*bytes in rank order; return 0 on success (bind it to RCCL, MPI or torch.distributed).
 * Transcripts run in lockstep, so every rank returns the same complete proof, byte-identical to the unsharded one, in
 * a thread-local buffer the caller BORROWS (valid until the next prove on this thread; do not free). */
typedef int (*zigzh_allgather_fn)(void *user, const void *send, size_t bytes, void *recv);
int zigzh_prove_trace_sharded(const zigzh_trace *t, zigz_ctx *ctx, const uint32_t *d_cols, size_t stride, int rank,
                              int world, zigzh_allgather_fn allgather, void *user, uint8_t **proof_out,
                              size_t *proof_len);

/* VMState.init + run(max_steps)   src/vm/state.zig:72-93,172-184 (VM known-answer tests) */
int zigzh_vm_run(const uint8_t *program, size_t program_len, uint64_t entry_pc, size_t max_steps,
                 uint64_t final_regs[32], uint64_t *final_pc, size_t *steps);

/* Multilinear.init + SumcheckProver.prove + SumcheckProof.toBytes through the C++ mirror classes */
int zigzh_sumcheck_prove_bytes(zigz_ctx *ctx, const uint64_t *evals, size_t n, uint8_t *out /*(3v+2)*8*/, size_t *out_len);
/* build{Add,Xor,And}Table(bits) + LassoProver.prove / proveWithMapping (mapping may be NULL)
 * queries: n_queries rows of 3 fields.  out: sumcheck toBytes, then the two 32-byte commitments */
int zigzh_lasso_prove_table(zigz_ctx *ctx, int kind, size_t bits, const uint64_t *queries, size_t n_queries,
                            const uint64_t *mapping, size_t n_mapping, uint8_t *sumcheck_bytes, size_t *sumcheck_len,
                            uint8_t query_commitment[32], uint8_t table_commitment[32], size_t *num_lookups);
/* Multilinear + CommitmentScheme.commit/open/verify through the mirror classes; returns verify() result in *ok */
int zigzh_commit_open_verify(zigz_ctx *ctx, const uint64_t *evals, size_t n, const uint64_t *point, size_t npoint,
                             uint8_t root[32], uint64_t *value, uint64_t *index, int *ok);

#ifdef __cplusplus
}
#endif
#endif
