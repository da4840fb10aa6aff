// extern "C" face of the host mirror (include/zigz_host.h) for ctypes callers.
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "zigz_host.h"
#include "zigz_host.hpp"

using namespace zigz;

static thread_local std::string g_err;
static thread_local double g_timings[10] = {0};
// trace storage recycled across executions on this thread (first-touch page faults of a fresh buffer are not free)
static thread_local std::vector<zigz_trace_step> g_step_pool;
static thread_local std::vector<uint8_t> g_proof;  // borrowed-proof buffer of zigzh_prove_trace(want_bytes = 2)

template <class Fn>
static int guard(Fn &&fn) {
    try {
        fn();
        return 0;
    } catch (const Error &e) {
        g_err = e.what();
        return e.code;
    } catch (const std::bad_alloc &) {
        g_err = "error.OutOfMemory";
        return ZIGZ_ERR_OUT_OF_MEMORY;
    } catch (const std::exception &e) {
        g_err = e.what();
        return ZIGZ_ERR_INVALID_ARGUMENT;
    }
}

static uint8_t *dup_bytes(const std::vector<uint8_t> &v) {
    uint8_t *p = (uint8_t *)malloc(v.size() ? v.size() : 1);
    if (!p) throw std::bad_alloc();
    memcpy(p, v.data(), v.size());
    return p;
}

extern "C" const char *zigzh_last_error(void) { return g_err.c_str(); }
extern "C" void zigzh_free(void *p) { free(p); }
extern "C" void zigzh_last_timings(double out[10]) { memcpy(out, g_timings, sizeof(g_timings)); }

struct zigzh_trace {
    PublicIO io;
    ExecutionTrace trace;
    size_t num_lookups = 0, num_vars = 0;
    std::optional<std::vector<uint64_t>> initial_regs;
    mutable std::vector<uint64_t> rows_cache;  // expandRows(), built on first use by zigzh_trace_rows
    zigz_ctx *registered = nullptr;            // context that page-locked trace.steps (zigzh_trace_pin)
    // the 32-byte form of the records + the side list of memory accesses (zigz_hip.h: zigz_trace_step32): what a service
    // uploads; built (and page-locked) by zigzh_trace_pin
    std::vector<zigz_trace_step32> steps32;
    std::vector<zigz_mem_access> mem;
    bool registered32 = false, registered_mem = false;
    // the 16-byte form (zigz_trace_step16 + the code table: the instruction fields once per pc) -- what a service uploads when
    // the trace allows it; shares the side list `mem` with the 32-byte form
    std::vector<zigz_trace_step16> steps16;
    std::vector<zigz_code_entry> code;
    uint64_t code_base = 0;
    bool registered16 = false, registered_code = false, tried16 = false;
    // false (and nothing kept) when the trace does not fit the form: a pc off the 4-byte grid or 2^32 past the lowest one, a code
    // span out of proportion to the trace, a pc executed with two different decodings, too many accesses
    bool compact16() {
        if (tried16) return !steps16.empty();
        tried16 = true;
        compact32();
        const size_t n = trace.stepCount();
        if (n == 0 || mem.size() >= ZIGZ_NO_MEM_ACCESS16) return false;
        uint64_t lo = ~0ull, hi = 0;
        for (size_t i = 0; i < n; i++) {
            const uint64_t pc = trace.steps[i].pc;
            lo = pc < lo ? pc : lo;
            hi = pc > hi ? pc : hi;
        }
        const uint64_t span = hi - lo;
        if ((lo & 3) || span >= (1ull << 32) || span / 4 + 1 > 4 * n + 1024) return false;
        std::vector<zigz_code_entry> tab(span / 4 + 1);
        std::vector<uint8_t> seen(tab.size(), 0);
        std::vector<zigz_trace_step16> out(n);
        for (size_t i = 0; i < n; i++) {
            const zigz_trace_step &a = trace.steps[i];
            const uint64_t off = a.pc - lo;
            if ((off & 3) || a.imm != (int64_t)(int32_t)a.imm || a.wr_reg >= 32 || a.mem_is_read > 1) return false;
            zigz_code_entry e{};
            e.imm = (int32_t)a.imm;
            e.opcode = a.opcode; e.rd = a.rd; e.rs1 = a.rs1; e.rs2 = a.rs2; e.funct3 = a.funct3; e.funct7 = a.funct7;
            zigz_code_entry &t = tab[off / 4];
            if (!seen[off / 4]) {
                t = e;
                seen[off / 4] = 1;
            } else if (memcmp(&t, &e, sizeof(e)) != 0) {
                return false;  // the program rewrote this instruction
            }
            const uint32_t mi = steps32[i].mem_index == (uint32_t)ZIGZ_NO_MEM_ACCESS ? (uint32_t)ZIGZ_NO_MEM_ACCESS16 : steps32[i].mem_index;
            out[i].pc_word = (uint32_t)off | (a.mem_is_read & 1u);
            out[i].mem_wr = mi | ((uint32_t)a.wr_reg << 27);
            out[i].rd_value = a.rd_value;
        }
        steps16.swap(out);
        code.swap(tab);
        code_base = lo;
        return true;
    }
    void compact32() {
        if (!steps32.empty() || trace.stepCount() == 0) return;
        const size_t n = trace.stepCount();
        steps32.resize(n);
        for (size_t i = 0; i < n; i++) {
            const zigz_trace_step &a = trace.steps[i];
            zigz_trace_step32 &b = steps32[i];
            if (a.imm != (int64_t)(int32_t)a.imm) throw Error(ZIGZ_ERR_INVALID_ARGUMENT, "an immediate does not fit 32 bits");
            b.pc = a.pc;
            b.rd_value = a.rd_value;
            b.imm = (int32_t)a.imm;
            b.opcode = a.opcode; b.rd = a.rd; b.rs1 = a.rs1; b.rs2 = a.rs2; b.funct3 = a.funct3; b.funct7 = a.funct7;
            b.wr_reg = a.wr_reg;
            b.mem_is_read = a.mem_is_read;
            // (a step has a memory access iff it is a LOAD (opcode 3) or a STORE (35): src/vm/state.zig:467-504)
            if (a.opcode == 0x03 || a.opcode == 0x23) {
                b.mem_index = (uint32_t)mem.size();
                mem.push_back(zigz_mem_access{a.mem_addr, a.mem_value});
            } else {
                b.mem_index = (uint32_t)ZIGZ_NO_MEM_ACCESS;
            }
        }
    }
};

extern "C" int zigzh_execute(const uint8_t *program, size_t program_len, uint64_t entry_pc, const uint64_t *initial_regs,
                             size_t n_initial_regs, int has_initial_regs, size_t max_steps, const uint64_t *input,
                             size_t n_input, zigzh_trace **out) {
    return guard([&] {
        std::vector<uint8_t> prog(program, program + program_len);
        std::vector<uint64_t> in(input, input + (input ? n_input : 0));
        auto t = std::make_unique<zigzh_trace>();
        zigz_sha256(prog.data(), prog.size(), t->io.program_hash.data());
        t->io.initial_pc = entry_pc;
        if (has_initial_regs) t->initial_regs = std::vector<uint64_t>(initial_regs, initial_regs + n_initial_regs);
        VMState vm(prog, entry_pc, input ? &in : nullptr);
        if (t->initial_regs)
            for (size_t i = 0; i < t->initial_regs->size() && i < 32; i++) vm.writeReg((unsigned)i, (*t->initial_regs)[i]);
        vm.trace.steps.swap(g_step_pool);  // recycled capacity (contents are overwritten step by step)
        vm.trace.reserveSteps(max_steps < ((size_t)1 << 22) ? max_steps : ((size_t)1 << 22));
        size_t step_count = 0;
        while (!vm.halted && step_count < max_steps) {  // prover.zig:132-142
            vm.step();
            if (vm.invalid_instruction) break;
            step_count++;
        }
        t->trace = std::move(vm.trace);
        const size_t ns = t->trace.stepCount();
        for (uint8_t f : t->trace.is_lookup) t->num_lookups += f;
        while (((size_t)1 << t->num_vars) < ns) t->num_vars++;
        t->io.final_pc = vm.pc;
        std::vector<uint64_t> fr(32);
        for (unsigned r = 0; r < 32; r++) fr[r] = vm.readReg(r);
        t->io.final_regs = fr;
        t->io.num_steps = ns;
        if (!vm.output_tape.empty()) t->io.outputs = vm.output_tape;
        t->io.initial_regs = t->initial_regs;
        *out = t.release();
    });
}
extern "C" void zigzh_trace_free(zigzh_trace *t) {
    if (!t) return;
    // (the context that registered the buffer may be gone by now: page-locking is process-wide, so no context is named)
    if (t->registered16) (void)zigz_host_unregister(nullptr, t->steps16.data());
    if (t->registered_code) (void)zigz_host_unregister(nullptr, t->code.data());
    if (t->registered32) (void)zigz_host_unregister(nullptr, t->steps32.data());
    if (t->registered_mem) (void)zigz_host_unregister(nullptr, t->mem.data());
    if (t->registered) (void)zigz_host_unregister(nullptr, t->trace.steps.data());
    else if (t->trace.steps.capacity() > g_step_pool.capacity()) g_step_pool.swap(t->trace.steps);
    delete t;
}
// which record a service uploads for this trace (after zigzh_trace_pin): 16, 32 or 48 bytes per step; *bytes = what crosses
// PCIe per proof (records + side list + code table)
extern "C" int zigzh_trace_upload_form(const zigzh_trace *t, size_t *bytes) {
    if (!t) return 0;
    const size_t n = t->trace.stepCount();
    int form = 48;
    size_t b = n * sizeof(zigz_trace_step);
    if (!t->steps16.empty()) {
        form = 16;
        b = n * 16 + t->mem.size() * 16 + t->code.size() * sizeof(zigz_code_entry);
    } else if (!t->steps32.empty()) {
        form = 32;
        b = n * 32 + t->mem.size() * 16;
    }
    if (bytes) *bytes = b;
    return form;
}
extern "C" int zigzh_trace_pin(zigzh_trace *t, zigz_ctx *ctx) {
    return guard([&] {
        if (t->registered || t->trace.stepCount() == 0) return;
        check(ctx, zigz_host_register(ctx, t->trace.steps.data(), t->trace.stepCount() * sizeof(zigz_trace_step)));
        t->registered = ctx;
        // the form a service uploads (zigzh_prove_trace_slots without resident columns): 16-byte records + code table when the trace
        // fits them -- the 32-byte records are then given back, a 2^24 trace is 0.5 GiB of them --, else the 32-byte records
        // (ZIGZ_TRACE32=1 / ZIGZ_TRACE48=1 at this point: keep the 32-byte form for an A/B)
        const bool keep32 = getenv("ZIGZ_TRACE32") || getenv("ZIGZ_TRACE48");
        const bool fits16 = !keep32 && t->compact16();
        if (fits16) {
            std::vector<zigz_trace_step32>().swap(t->steps32);
            check(ctx, zigz_host_register(ctx, t->steps16.data(), t->steps16.size() * sizeof(zigz_trace_step16)));
            t->registered16 = true;
            check(ctx, zigz_host_register(ctx, t->code.data(), t->code.size() * sizeof(zigz_code_entry)));
            t->registered_code = true;
        } else {
            t->compact32();
            check(ctx, zigz_host_register(ctx, t->steps32.data(), t->steps32.size() * sizeof(zigz_trace_step32)));
            t->registered32 = true;
        }
        if (!t->mem.empty()) {
            check(ctx, zigz_host_register(ctx, t->mem.data(), t->mem.size() * sizeof(zigz_mem_access)));
            t->registered_mem = true;
        }
    });
}
extern "C" const void *zigzh_trace_steps(const zigzh_trace *t) { return t->trace.steps.data(); }
extern "C" const uint64_t *zigzh_trace_initial_regs(const zigzh_trace *t) { return t->trace.initial_regs; }
extern "C" size_t zigzh_trace_num_steps(const zigzh_trace *t) { return t->trace.stepCount(); }
extern "C" size_t zigzh_trace_num_vars(const zigzh_trace *t) { return t->num_vars; }
extern "C" size_t zigzh_trace_num_lookups(const zigzh_trace *t) { return t->num_lookups; }
extern "C" const uint64_t *zigzh_trace_rows(const zigzh_trace *t) {
    if (t->rows_cache.empty()) t->rows_cache = t->trace.expandRows();
    return t->rows_cache.data();
}

extern "C" int zigzh_trace_witness(const zigzh_trace *t, uint64_t *cols_out) {
    return guard([&] {
        Witness w = WitnessGenerator::generate(t->trace);
        memcpy(cols_out, w.columns.data(), w.columns.size() * sizeof(uint64_t));
    });
}

extern "C" int zigzh_trace_witness_dev(const zigzh_trace *t, zigz_ctx *ctx, uint32_t *d_cols, size_t stride) {
    return guard([&] {
        // the trace is recorded as compact 48-byte steps: one H2D + the expansion kernels (K8)
        check(ctx, zigz_dev_witness_from_steps(ctx, t->trace.steps.data(), t->trace.stepCount(), t->num_vars,
                                               t->trace.initial_regs, d_cols, stride));
    });
}

extern "C" int zigzh_trace_witness_dev_async(const zigzh_trace *t, zigz_ctx *ctx, uint32_t *d_cols, size_t stride) {
    return guard([&] {
        // page-locking is a property of the process, not of the context that asked for it: any context may upload from it
        if (!t->registered) throw Error(ZIGZ_ERR_BAD_STATE, "zigzh_trace_witness_dev_async: pin the trace first (zigzh_trace_pin)");
        check(ctx, zigz_dev_witness_from_steps_async(ctx, t->trace.steps.data(), t->trace.stepCount(), t->num_vars,
                                                     t->trace.initial_regs, d_cols, stride));
    });
}

extern "C" int zigzh_prove_trace(const zigzh_trace *t, zigz_ctx *ctx, const uint32_t *d_cols, size_t stride, int want_bytes,
                                 uint8_t **proof_out, size_t *proof_len) {
    return guard([&] {
        if (t->trace.stepCount() == 0) throw Error(ZIGZ_ERR_EMPTY_TRACE, "error.EmptyTrace");
        Prover prover(ctx, 0);
        Proof proof;
        const std::vector<uint64_t> *ir = t->initial_regs ? &*t->initial_regs : nullptr;
        if (want_bytes == 2) {  // overlapped serialisation into a reusable thread-local buffer (borrowed by the caller)
            if (d_cols) {
                prover.proveWitnessToBytes(t->io, t->num_lookups, nullptr, d_cols, stride, t->num_vars, ir, g_proof);
            } else {
                Witness w = WitnessGenerator::generate(t->trace);
                prover.proveWitnessToBytes(t->io, t->num_lookups, &w, nullptr, 0, w.num_vars, ir, g_proof);
            }
            memcpy(g_timings, prover.timings, sizeof(g_timings));
            *proof_out = g_proof.data();
            *proof_len = g_proof.size();
            return;
        }
        if (d_cols) {
            proof = prover.proveWitness(t->io, t->num_lookups, nullptr, d_cols, stride, t->num_vars, ir);
        } else {
            Witness w = WitnessGenerator::generate(t->trace);
            proof = prover.proveWitness(t->io, t->num_lookups, &w, nullptr, 0, w.num_vars, ir);
        }
        memcpy(g_timings, prover.timings, sizeof(g_timings));
        if (want_bytes == 1) {
            auto t0 = std::chrono::steady_clock::now();
            std::vector<uint8_t> b = BinarySerializer::serialize(proof);
            *proof_out = dup_bytes(b);
            *proof_len = b.size();
            g_timings[7] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
    });
}

// ---------------------------------------------------------------- GPU slots of a proving service (zigz_host.hpp: GpuSlots)
struct zigzh_slots {
    GpuSlots slots;
    std::unique_ptr<GpuBatcher> batcher;  // small traces share commit jobs (zigzh_slots_set_batching)
    zigzh_slots(int device, size_t k) : slots(device, k) {}
};
extern "C" int zigzh_slots_set_batching(zigzh_slots *s, unsigned max_batch, double linger_us, size_t max_nv) {
    return guard([&] {
        if (!s || max_batch > 32) throw Error(ZIGZ_ERR_INVALID_ARGUMENT, "zigzh_slots_set_batching: bad argument");
        if (max_batch <= 1) s->batcher.reset();
        else s->batcher.reset(new GpuBatcher(&s->slots, max_batch, linger_us * 1e-6, max_nv < 20 ? max_nv : 20));
    });
}
extern "C" int zigzh_slots_create(int device, size_t k, zigzh_slots **out) {
    return guard([&] { *out = new zigzh_slots(device, k); });
}
extern "C" void zigzh_slots_destroy(zigzh_slots *s) { delete s; }
extern "C" size_t zigzh_slots_size(const zigzh_slots *s) { return s ? s->slots.size() : 0; }
extern "C" zigz_ctx *zigzh_slots_ctx(zigzh_slots *s, size_t i) { return s && i < s->slots.size() ? s->slots.at(i) : nullptr; }
extern "C" zigz_ctx *zigzh_slots_acquire(zigzh_slots *s) { return s ? s->slots.acquire() : nullptr; }
extern "C" void zigzh_slots_release(zigzh_slots *s, zigz_ctx *ctx) {
    if (s && ctx) s->slots.release(ctx);
}

// the records a proof uploads inside its slot: the 16-byte form when the trace has it (pinned traces that fit it), else the
// 32-byte one, else the 48-byte one (ZIGZ_TRACE32=1 / ZIGZ_TRACE48=1 force those forms: A/B)
static TraceRecords records_of(const zigzh_trace *t) {
    TraceRecords r;
    r.regs_before = t->trace.initial_regs;
    static const bool force48 = getenv("ZIGZ_TRACE48") && getenv("ZIGZ_TRACE48")[0] == '1';
    static const bool force32 = getenv("ZIGZ_TRACE32") && getenv("ZIGZ_TRACE32")[0] == '1';
    if (!t->steps16.empty() && !force48 && !force32) {
        r.s16 = t->steps16.data();
        r.code = t->code.data();
        r.ncode = t->code.size();
        r.code_base = t->code_base;
        r.mem = t->mem.data();
        r.nmem = t->mem.size();
    } else if (!t->steps32.empty() && !force48) {
        r.s32 = t->steps32.data();
        r.mem = t->mem.data();
        r.nmem = t->mem.size();
    } else {
        r.s48 = t->trace.steps.data();
    }
    return r;
}

extern "C" int zigzh_prove_trace_slots(const zigzh_trace *t, zigzh_slots *s, const uint32_t *d_cols, size_t stride,
                                       uint8_t **proof_out, size_t *proof_len, zigz_kernel_stats *stats_out,
                                       zigz_launch_rec *log_out, size_t log_cap, size_t *log_n) {
    return guard([&] {
        if (!t || !s || !proof_out || !proof_len) throw Error(ZIGZ_ERR_INVALID_ARGUMENT, "null argument");
        if (t->trace.stepCount() == 0) throw Error(ZIGZ_ERR_EMPTY_TRACE, "error.EmptyTrace");
        Prover prover(&s->slots, 0, s->batcher.get());
        const std::vector<uint64_t> *ir = t->initial_regs ? &*t->initial_regs : nullptr;
        if (d_cols) prover.proveWitnessToBytes(t->io, t->num_lookups, nullptr, d_cols, stride, t->num_vars, ir, g_proof);
        else prover.proveStepsToBytes(t->io, t->num_lookups, records_of(t), t->num_vars, ir, g_proof);
        memcpy(g_timings, prover.timings, sizeof(g_timings));
        *proof_out = g_proof.data();
        *proof_len = g_proof.size();
        if (stats_out) *stats_out = prover.last_stats;
        if (log_n) {
            const size_t n = prover.last_log.size() < log_cap ? prover.last_log.size() : log_cap;
            if (n && log_out) memcpy(log_out, prover.last_log.data(), n * sizeof(zigz_launch_rec));
            *log_n = log_out ? n : 0;
        }
    });
}

// field by field (tests/test_host_mirror.py checks that no field of zigz_kernel_stats is left out)
static void stats_add(zigz_kernel_stats &a, const zigz_kernel_stats &b) {
    a.merkle_build_us += b.merkle_build_us;
    a.eval_us += b.eval_us;
    a.path_us += b.path_us;
    a.bind_us += b.bind_us;
    a.bind_launches += b.bind_launches;
    a.keccak_permutations += b.keccak_permutations;
    a.bind_vec_us += b.bind_vec_us;
    a.bind_vec_launches += b.bind_vec_launches;
    a.bind_vec_bytes += b.bind_vec_bytes;
    a.run_aware_columns += b.run_aware_columns;
    a.run_aware_dense_nodes += b.run_aware_dense_nodes;
    a.run_aware_hashed += b.run_aware_hashed;
    a.run_aware_us += b.run_aware_us;
    a.cons_columns += b.cons_columns;
    a.cons_dense_nodes += b.cons_dense_nodes;
    a.cons_hashed += b.cons_hashed;
    a.cons_probe_distinct += b.cons_probe_distinct;
    a.keccak_leaves_us += b.keccak_leaves_us;
    a.keccak_leaves_perms += b.keccak_leaves_perms;
    a.keccak_level_wide_us += b.keccak_level_wide_us;
    a.keccak_level_wide_perms += b.keccak_level_wide_perms;
    a.keccak_level_small_us += b.keccak_level_small_us;
    a.keccak_level_small_perms += b.keccak_level_small_perms;
    a.small_domain_columns += b.small_domain_columns;
    a.small_domain_us += b.small_domain_us;
    a.small_domain_fallback_waves += b.small_domain_fallback_waves;
    a.structure_us += b.structure_us;
    a.list_hash_us += b.list_hash_us;
    a.list_hash_perms += b.list_hash_perms;
    a.top_us += b.top_us;
    a.top_perms += b.top_perms;
    a.rebuilds += b.rebuilds;
    a.eval_constant_columns += b.eval_constant_columns;
}
static_assert(sizeof(zigz_kernel_stats) == 33 * 8, "zigz_kernel_stats grew: add the new fields to stats_add");
extern "C" void zigzh_stats_add(zigz_kernel_stats *a, const zigz_kernel_stats *b) { stats_add(*a, *b); }

// `reps` proofs of the same trace back to back on the calling thread, without returning to the caller in between: a lane of a
// proving service.  (A Python caller pays ~70 us of interpreter time per proof for the call and its result objects -- with
// 93 lane threads behind one interpreter lock that caps the process at ~14 k proofs/s, which small traces exceed.)
// stats_sum / timings_sum: field-wise sums over the proofs; the proof bytes are the last proof's (borrowed).
extern "C" int zigzh_prove_trace_slots_repeat(const zigzh_trace *t, zigzh_slots *s, const uint32_t *d_cols, size_t stride, size_t reps,
                                              uint8_t **proof_out, size_t *proof_len, zigz_kernel_stats *stats_sum,
                                              double timings_sum[10]) {
    return guard([&] {
        if (!t || !s || !proof_out || !proof_len || reps == 0) throw Error(ZIGZ_ERR_INVALID_ARGUMENT, "null argument");
        if (t->trace.stepCount() == 0) throw Error(ZIGZ_ERR_EMPTY_TRACE, "error.EmptyTrace");
        const std::vector<uint64_t> *ir = t->initial_regs ? &*t->initial_regs : nullptr;
        zigz_kernel_stats sum{};
        double tsum[10] = {0};
        for (size_t r = 0; r < reps; r++) {
            Prover prover(&s->slots, 0, s->batcher.get());
            if (d_cols) prover.proveWitnessToBytes(t->io, t->num_lookups, nullptr, d_cols, stride, t->num_vars, ir, g_proof);
            else prover.proveStepsToBytes(t->io, t->num_lookups, records_of(t), t->num_vars, ir, g_proof);
            stats_add(sum, prover.last_stats);
            for (int i = 0; i < 10; i++) tsum[i] += prover.timings[i];
            memcpy(g_timings, prover.timings, sizeof(g_timings));
        }
        *proof_out = g_proof.data();
        *proof_len = g_proof.size();
        if (stats_sum) *stats_sum = sum;
        if (timings_sum) memcpy(timings_sum, tsum, sizeof(tsum));
    });
}

// The commit path alone, `reps` times back to back inside the library (measurement: what the GPU needs per proof when no host
// transcript stands in front of it -- bench.py's gpu_bound legs; a worker in the interpreter would add its own ~70 us per job and
// serialise with the other workers behind the interpreter lock): take a slot, begin on the 43 resident columns, roots, open_all
// at `points` (43 x nv canonical), end, give the slot back.  masks: small_domain / run_aware / cons_group options for the jobs.
extern "C" int zigzh_commit_path_repeat(zigzh_slots *s, const uint32_t *d_cols, size_t stride, size_t nv, const uint64_t *points,
                                        const int64_t masks[3], size_t reps) {
    return guard([&] {
        if (!s || !d_cols || !points || !masks || reps == 0) throw Error(ZIGZ_ERR_INVALID_ARGUMENT, "null argument");
        const size_t nc = ZIGZ_NUM_COLUMNS;
        std::vector<uint8_t> roots(nc * 32), sib(nc * nv * 32 + 1), dirs(nc * nv + 1);
        std::vector<uint64_t> values(nc), indices(nc), leaves(nc);
        static const char *const names[3] = {"small_domain_mask", "run_aware_mask", "cons_group_mask"};
        for (size_t r = 0; r < reps; r++) {
            GpuSlots::Lease lease;
            lease.slots = &s->slots;
            zigz_ctx *c = lease.ctx = s->slots.acquire();
            int64_t saved[3];
            for (int i = 0; i < 3; i++) {
                (void)zigz_ctx_get_option(c, names[i], &saved[i]);
                check(c, zigz_ctx_set_option(c, names[i], masks[i]));
            }
            zigz_commit_job *job = nullptr;
            zigz_status st = zigz_commit_begin_dev(c, d_cols, nc, stride, nv, &job);
            if (st == ZIGZ_OK) st = zigz_commit_roots(job, roots.data());
            if (st == ZIGZ_OK) st = zigz_commit_open_all(job, points, values.data(), indices.data(), leaves.data(), sib.data(), dirs.data());
            if (job) zigz_commit_end(job);
            for (int i = 0; i < 3; i++) (void)zigz_ctx_set_option(c, names[i], saved[i]);
            check(c, st);
        }
    });
}

extern "C" int zigzh_prove(zigz_ctx *ctx, const uint8_t *program, size_t program_len, uint64_t entry_pc,
                           const uint64_t *initial_regs, size_t n_initial_regs, int has_initial_regs, size_t max_steps,
                           const uint64_t *input, size_t n_input, uint8_t **proof_out, size_t *proof_len, size_t *num_steps) {
    return guard([&] {
        std::vector<uint8_t> prog(program, program + program_len);
        std::vector<uint64_t> regs, in;
        if (has_initial_regs) regs.assign(initial_regs, initial_regs + n_initial_regs);
        if (input) in.assign(input, input + n_input);
        Prover prover(ctx, 0);  // main.cmdProve: Prover(F).init(allocator, 0), src/main.zig:148
        // serialisation overlapped with the transcript, into the thread's reusable buffer; one copy into the caller's
        Proof proof = prover.prove(prog, entry_pc, has_initial_regs ? &regs : nullptr, max_steps, nullptr, input ? &in : nullptr,
                                   &g_proof);
        *proof_out = dup_bytes(g_proof);
        *proof_len = g_proof.size();
        if (num_steps) *num_steps = proof.public_io.num_steps;
    });
}

extern "C" int zigzh_verify(const uint8_t *proof, size_t proof_len, const uint8_t *program, size_t program_len, int *result) {
    return guard([&] {
        Proof p = BinarySerializer::deserialize(proof, proof_len);
        Verifier v;
        *result = (int)v.verify(p, std::vector<uint8_t>(program, program + program_len));
    });
}

extern "C" int zigzh_reserialize(const uint8_t *proof, size_t proof_len, uint8_t **out, size_t *out_len) {
    return guard([&] {
        Proof p = BinarySerializer::deserialize(proof, proof_len);
        std::vector<uint8_t> b = BinarySerializer::serialize(p);
        *out = dup_bytes(b);
        *out_len = b.size();
    });
}

extern "C" int zigzh_prove_trace_sharded(const zigzh_trace *t, zigz_ctx *ctx, const uint32_t *d_cols, size_t stride, int rank,
                                         int world, zigzh_allgather_fn allgather, void *user, uint8_t **proof_out,
                                         size_t *proof_len) {
    return guard([&] {
        if (!t || !ctx || !d_cols || !proof_out || !proof_len) throw Error(ZIGZ_ERR_INVALID_ARGUMENT, "null argument");
        if (t->trace.stepCount() == 0) throw Error(ZIGZ_ERR_EMPTY_TRACE, "error.EmptyTrace");
        Prover prover(ctx, 0);
        ShardSpec sp;
        sp.rank = rank;
        sp.world = world;
        sp.allgather = allgather;
        sp.user = user;
        prover.setShard(sp);
        const std::vector<uint64_t> *ir = t->initial_regs ? &*t->initial_regs : nullptr;
        // serialisation overlapped with the transcript, into the thread-local buffer the caller borrows
        prover.proveWitnessToBytes(t->io, t->num_lookups, nullptr, d_cols, stride, t->num_vars, ir, g_proof);
        memcpy(g_timings, prover.timings, sizeof(g_timings));
        *proof_out = g_proof.data();
        *proof_len = g_proof.size();
    });
}

extern "C" int zigzh_vm_run(const uint8_t *program, size_t program_len, uint64_t entry_pc, size_t max_steps,
                            uint64_t final_regs[32], uint64_t *final_pc, size_t *steps) {
    std::unique_ptr<VMState> vm;
    int rc = guard([&] {
        vm.reset(new VMState(std::vector<uint8_t>(program, program + program_len), entry_pc, nullptr));
        vm->run(max_steps);
    });
    if (vm) {
        for (unsigned r = 0; r < 32; r++) final_regs[r] = vm->readReg(r);
        *final_pc = vm->pc;
        *steps = vm->trace.stepCount();
    }
    return rc;
}

extern "C" int zigzh_sumcheck_prove_bytes(zigz_ctx *ctx, const uint64_t *evals, size_t n, uint8_t *out, size_t *out_len) {
    return guard([&] {
        Multilinear poly = Multilinear::init(ctx, std::vector<F>(evals, evals + n));
        if (poly.num_vars == 0) throw Error(ZIGZ_ERR_NO_VARIABLES, "error.NoVariables");  // sumcheck_prover.zig:30-32
        SumcheckProof p = SumcheckProver::prove(poly);
        std::vector<uint8_t> b = p.toBytes();
        memcpy(out, b.data(), b.size());
        *out_len = b.size();
    });
}

extern "C" int zigzh_lasso_prove_table(zigz_ctx *ctx, int kind, size_t bits, const uint64_t *queries, size_t n_queries,
                                       const uint64_t *mapping, size_t n_mapping, uint8_t *sumcheck_bytes,
                                       size_t *sumcheck_len, uint8_t query_commitment[32], uint8_t table_commitment[32],
                                       size_t *num_lookups) {
    return guard([&] {
        DenseTable table = kind == 0 ? buildAddTable(bits) : kind == 1 ? buildXorTable(bits) : buildAndTable(bits);
        std::vector<LookupQuery> qs(n_queries);
        for (size_t j = 0; j < n_queries; j++) {
            qs[j].inputs = {queries[3 * j], queries[3 * j + 1]};
            qs[j].expected_outputs = {queries[3 * j + 2]};
        }
        LassoProofFull p;
        if (mapping) p = LassoProver::proveWithMapping(ctx, table, qs, std::vector<size_t>(mapping, mapping + n_mapping));
        else p = LassoProver::prove(ctx, table, qs);
        std::vector<uint8_t> b = p.sumcheck_proof.toBytes();
        memcpy(sumcheck_bytes, b.data(), b.size());
        *sumcheck_len = b.size();
        memcpy(query_commitment, p.query_commitment.data(), 32);
        memcpy(table_commitment, p.table_commitment.data(), 32);
        *num_lookups = p.num_lookups;
    });
}

extern "C" int zigzh_commit_open_verify(zigz_ctx *ctx, const uint64_t *evals, size_t n, const uint64_t *point, size_t npoint,
                                        uint8_t root[32], uint64_t *value, uint64_t *index, int *ok) {
    return guard([&] {
        Multilinear poly = Multilinear::init(ctx, std::vector<F>(evals, evals + n));
        auto c = CommitmentScheme::commit(poly);
        memcpy(root, c.commitment.data(), 32);
        PolyOpeningProof pr = CommitmentScheme::open(poly, c.tree, std::vector<F>(point, point + npoint));
        *value = pr.value;
        *index = pr.merkle_proof.index;
        *ok = CommitmentScheme::verify(c.commitment, c.num_vars, pr) ? 1 : 0;
    });
}
