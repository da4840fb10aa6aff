// C ABI of libzigz_hip.so (include/zigz_hip.h), part 1: contexts, workspaces, options, boundary conversion (uploads, the witness
// from trace records), the host transcript.  No CPU fallback for field or hash work on the data path.
#include "api_internal.hpp"

using namespace zk;

void set_err(zigz_ctx *ctx, const char *fmt, ...) {
    if (!ctx) return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
    va_end(ap);
}

zigz_status ws_get(zigz_ctx *ctx, int slot, size_t bytes, void **out) {
    if (bytes == 0) bytes = 16;
    if (ctx->ws_bytes[slot] < bytes) {
        if (ctx->ws[slot]) {
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            HIPCHK(ctx, hipFree(ctx->ws[slot]));
            ctx->ws[slot] = nullptr;
            ctx->ws_bytes[slot] = 0;
        }
        size_t want = bytes + bytes / 8;  // a little slack so slowly growing sizes do not realloc each call
        want = (want + 255) & ~(size_t)255;
        hipError_t e = hipMalloc(&ctx->ws[slot], want);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            want = (bytes + 255) & ~(size_t)255;
            e = hipMalloc(&ctx->ws[slot], want);
            if (e != hipSuccess) {
                (void)hipGetLastError();  // (not left behind for the next launch check on this thread to find)
                ctx->ws[slot] = nullptr;
                set_err(ctx, "workspace %d: hipMalloc(%zu bytes) failed: %s", slot, want, hipGetErrorString(e));
                return e == hipErrorOutOfMemory ? ZIGZ_ERR_OUT_OF_MEMORY : ZIGZ_ERR_HIP;
            }
        }
        ctx->ws_bytes[slot] = want;
    }
    *out = ctx->ws[slot];
    return ZIGZ_OK;
}

extern "C" uint32_t zigz_abi_version(void) { return ZIGZ_ABI_VERSION; }

extern "C" const char *zigz_status_name(zigz_status s) {
    switch (s) {
    case ZIGZ_OK: return "OK";
    case ZIGZ_ERR_EMPTY_EVALUATIONS: return "EmptyEvaluations";
    case ZIGZ_ERR_LENGTH_NOT_POWER_OF_TWO: return "LengthNotPowerOfTwo";
    case ZIGZ_ERR_WRONG_NUMBER_OF_VARIABLES: return "WrongNumberOfVariables";
    case ZIGZ_ERR_NO_VARIABLES_TO_FIX: return "NoVariablesToFix";
    case ZIGZ_ERR_NO_VARIABLES: return "NoVariables";
    case ZIGZ_ERR_PROTOCOL_ERROR: return "ProtocolError";
    case ZIGZ_ERR_EMPTY_VALUES: return "EmptyValues";
    case ZIGZ_ERR_TOO_MANY_VALUES: return "TooManyValues";
    case ZIGZ_ERR_INDEX_OUT_OF_BOUNDS: return "IndexOutOfBounds";
    case ZIGZ_ERR_POINT_DIMENSION_MISMATCH: return "PointDimensionMismatch";
    case ZIGZ_ERR_NO_QUERIES: return "NoQueries";
    case ZIGZ_ERR_TOO_MANY_QUERIES: return "TooManyQueries";
    case ZIGZ_ERR_MAPPING_LENGTH_MISMATCH: return "MappingLengthMismatch";
    case ZIGZ_ERR_INVALID_MAPPING: return "InvalidMapping";
    case ZIGZ_ERR_QUERY_TABLE_MISMATCH: return "QueryTableMismatch";
    case ZIGZ_ERR_EMPTY_TRACE: return "EmptyTrace";
    case ZIGZ_ERR_OUT_OF_MEMORY: return "OutOfMemory";
    case ZIGZ_ERR_WRONG_NUMBER_OF_CHALLENGES: return "WrongNumberOfChallenges";
    case ZIGZ_ERR_NO_DEVICE: return "NoDevice";
    case ZIGZ_ERR_HIP: return "HipError";
    case ZIGZ_ERR_NOT_CANONICAL: return "NotCanonical";
    case ZIGZ_ERR_INVALID_ARGUMENT: return "InvalidArgument";
    case ZIGZ_ERR_COMM: return "CommError";
    case ZIGZ_ERR_BAD_STATE: return "BadState";
    default: return "Unknown";
    }
}

extern "C" zigz_status zigz_device_count(int *count) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    if (count) *count = n;
    return n > 0 ? ZIGZ_OK : ZIGZ_ERR_NO_DEVICE;
}

// Waiting host threads sleep (zigz_device_set_blocking_sync): the runtime's own interrupt wait costs 0.2-0.5 ms of CPU per
// wait once tens of threads of a process wait at the same time (measured with CLOCK_THREAD_CPUTIME_ID around the calls:
// 0.46 ms in the hipStreamSynchronize of zigz_commit_open_all, 0.15-0.23 ms in the event wait of zigz_commit_roots, with 80
// proving threads), so the two waits of a proof's commit path poll a completion word that the last kernel stores into pinned
// memory (DoneFlag, kernels.hpp), sleeping 30 -> 150 us between looks.
std::atomic<int> g_sleep_wait{0};
bool sleep_wait(const unsigned long long *flag, unsigned long long seq) {
    long ns = 30000;
    for (int i = 0; i < 20000; i++) {  // ~3 s, then the caller asks the runtime (which also reports a fault)
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return true;
        timespec ts{0, ns};
        nanosleep(&ts, nullptr);
        if (ns < 150000) ns += ns / 2;
    }
    return false;
}
// the short waits of a sumcheck's data passes (tens of microseconds): look at the word without sleeping for ~0.3 ms, then give up
// (the caller asks the runtime)
bool spin_wait(const unsigned long long *flag, unsigned long long seq) {
    for (int i = 0; i < 400000; i++) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return true;
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    return false;
}
DoneFlag done_flag(zigz_ctx *ctx, int which) {  // which: 0 = the roots of a commit job, 1 = its openings, 2 = a sumcheck's pass
    DoneFlag d;
    d.count = ctx->d_flag + 4;
    d.flag = (unsigned long long *)(ctx->h_roots + ROOTS_MAX_COLS * 32 + JOB_SUMMARY_WORDS * 8) + which;
    d.seq = ++ctx->done_seq;
    return d;
}

extern "C" zigz_status zigz_device_set_blocking_sync(int device, int on) {
    int n = 0;
    if (zigz_device_count(&n) != ZIGZ_OK || device < 0 || device >= n) return ZIGZ_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return ZIGZ_ERR_HIP;
    g_sleep_wait.store(on ? 1 : 0);
    return hipSetDeviceFlags(on ? hipDeviceScheduleBlockingSync : hipDeviceScheduleAuto) == hipSuccess ? ZIGZ_OK : ZIGZ_ERR_HIP;
}

extern "C" zigz_status zigz_ctx_create(int device, zigz_ctx **out) {
    if (!out) return ZIGZ_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int n = 0;
    if (zigz_device_count(&n) != ZIGZ_OK || device < 0 || device >= n) return ZIGZ_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return ZIGZ_ERR_NO_DEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return ZIGZ_ERR_NO_DEVICE;  // kernels are built for gfx950 only
    zigz_ctx *ctx = new (std::nothrow) zigz_ctx();
    if (!ctx) return ZIGZ_ERR_OUT_OF_MEMORY;
    memset(ctx, 0, sizeof(*ctx));
    ctx->device = device;
    zigz_status st = ZIGZ_OK;
    auto fail = [&](hipError_t e) { return e != hipSuccess; };
    if (fail(hipSetDevice(device)) || fail(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) ||
        fail(hipMalloc((void **)&ctx->d_sums, SUMS_SLOTS * sizeof(unsigned long long))) ||
        fail(hipMalloc((void **)&ctx->d_flag, 64)) || fail(hipMalloc((void **)&ctx->d_run_count, RUN_CTR_WORDS * 8)) || fail(hipMalloc((void **)&ctx->d_cons_count, RUN_CTR_WORDS * 8)) ||
        fail(hipMalloc((void **)&ctx->d_run_aux, RUN_CTR_WORDS * 8)) || fail(hipMalloc((void **)&ctx->d_cons_aux, RUN_CTR_WORDS * 8)) ||
        fail(hipHostMalloc((void **)&ctx->h_pin, PIN_WORDS * sizeof(uint64_t), hipHostMallocDefault)) ||
        fail(hipHostMalloc((void **)&ctx->h_roots, ROOTS_MAX_COLS * 32 + JOB_SUMMARY_WORDS * 8 + 64, hipHostMallocDefault)))
        st = ZIGZ_ERR_HIP;
    // (word 4 of d_flag: the DoneFlag counter, zero between launches.  Zeroed on the context's own stream: a plain hipMemset
    // would bring the legacy null stream into the process, and with it implicit synchronisation against every other stream)
    if (st == ZIGZ_OK && (fail(hipMemsetAsync(ctx->d_flag, 0, 64, ctx->own_stream)) || fail(hipStreamSynchronize(ctx->own_stream))))
        st = ZIGZ_ERR_HIP;
    if (st == ZIGZ_OK) memset(ctx->h_roots + ROOTS_MAX_COLS * 32 + JOB_SUMMARY_WORDS * 8, 0, 64);
    for (int i = 0; st == ZIGZ_OK && i < 6; i++)
        if (fail(hipEventCreate(&ctx->ev[i]))) st = ZIGZ_ERR_HIP;
    for (int i = 0; st == ZIGZ_OK && i < 128; i++)
        if (fail(hipEventCreate(&ctx->pool[i]))) st = ZIGZ_ERR_HIP;
    for (int i = 0; st == ZIGZ_OK && i < 2 * KEV_MAX; i++)
        if (fail(hipEventCreate(&ctx->kev[i]))) st = ZIGZ_ERR_HIP;
    if (st == ZIGZ_OK && fail(hipEventCreate(&ctx->epoch_own))) st = ZIGZ_ERR_HIP;
    if (st != ZIGZ_OK) {
        zigz_ctx_destroy(ctx);
        return st;
    }
    ctx->stream = ctx->own_stream;
    *out = ctx;
    return ZIGZ_OK;
}

extern "C" void zigz_ctx_destroy(zigz_ctx *ctx) {
    ZIGZ_ENTER(ctx);
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < WS_SLOTS; i++)
        if (ctx->ws[i]) (void)hipFree(ctx->ws[i]);
    if (ctx->d_sums) (void)hipFree(ctx->d_sums);
    if (ctx->d_flag) (void)hipFree(ctx->d_flag);
    if (ctx->d_run_count) (void)hipFree(ctx->d_run_count);
    if (ctx->d_cons_count) (void)hipFree(ctx->d_cons_count);
    if (ctx->d_run_aux) (void)hipFree(ctx->d_run_aux);
    if (ctx->d_cons_aux) (void)hipFree(ctx->d_cons_aux);
    if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
    if (ctx->h_roots) (void)hipHostFree(ctx->h_roots);
    for (int i = 0; i < 6; i++)
        if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
    for (int i = 0; i < 128; i++)
        if (ctx->pool[i]) (void)hipEventDestroy(ctx->pool[i]);
    for (int i = 0; i < 2 * KEV_MAX; i++)
        if (ctx->kev[i]) (void)hipEventDestroy(ctx->kev[i]);
    if (ctx->epoch_own) (void)hipEventDestroy(ctx->epoch_own);
    if (ctx->d_flush) (void)hipFree(ctx->d_flush);
    if (ctx->d_sd_tables) (void)hipFree(ctx->d_sd_tables);
    if (ctx->d_sd_fallbacks) (void)hipFree(ctx->d_sd_fallbacks);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

extern "C" const char *zigz_last_error(const zigz_ctx *ctx) { return ctx ? ctx->err : "no context"; }

extern "C" zigz_status zigz_ctx_set_stream(zigz_ctx *ctx, void *hip_stream) {
    ZIGZ_ENTER(ctx);
    if (!ctx) return ZIGZ_ERR_INVALID_ARGUMENT;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return ZIGZ_OK;
}
extern "C" void *zigz_ctx_get_stream(zigz_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

extern "C" zigz_status zigz_ctx_synchronize(zigz_ctx *ctx) {
    ZIGZ_ENTER(ctx);
    if (!ctx) return ZIGZ_ERR_INVALID_ARGUMENT;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return ZIGZ_OK;
}

extern "C" zigz_status zigz_dev_alloc(zigz_ctx *ctx, size_t bytes, void **d_out) {
    ZIGZ_ENTER(ctx);
    if (!ctx || !d_out) return ZIGZ_ERR_INVALID_ARGUMENT;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMalloc(d_out, bytes ? bytes : 16));
    return ZIGZ_OK;
}
extern "C" zigz_status zigz_ctx_release_workspaces(zigz_ctx *ctx) {
    ZIGZ_ENTER(ctx);
    if (!ctx) return ZIGZ_ERR_INVALID_ARGUMENT;
    if (ctx->active_job) return ZIGZ_ERR_BAD_STATE;  // the job's trees live in them
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    // An event recorded around a launch (timing mode) keeps that launch's command alive, and the command the buffers it used:
    // hipFree of such a buffer returns, but the memory does not come back until the event is recorded again or destroyed
    // (measured: a context whose last timed proof used a 3 GiB tree kept those 3 GiB through this call).  So the events go too.
    if (ctx->events_recorded) {
        for (int i = 0; i < 6; i++) {
            (void)hipEventDestroy(ctx->ev[i]);
            ctx->ev[i] = nullptr;
            HIPCHK(ctx, hipEventCreate(&ctx->ev[i]));
        }
        for (int i = 0; i < 128; i++) {
            (void)hipEventDestroy(ctx->pool[i]);
            ctx->pool[i] = nullptr;
            HIPCHK(ctx, hipEventCreate(&ctx->pool[i]));
        }
        for (int i = 0; i < 2 * KEV_MAX; i++) {
            (void)hipEventDestroy(ctx->kev[i]);
            ctx->kev[i] = nullptr;
            HIPCHK(ctx, hipEventCreate(&ctx->kev[i]));
        }
        ctx->events_recorded = ctx->timing;
        ctx->pool_used = 0;
        ctx->kev_n = 0;
    }
    for (int i = 0; i < WS_SLOTS; i++)
        if (ctx->ws[i]) {
            (void)hipFree(ctx->ws[i]);
            ctx->ws[i] = nullptr;
            ctx->ws_bytes[i] = 0;
        }
    ctx->cons_table = nullptr;  // (the content-addressing table went with them: the next build clears its new one)
    ctx->cons_table_bytes = 0;
    return ZIGZ_OK;
}
extern "C" zigz_status zigz_dev_mem_info(zigz_ctx *ctx, size_t *free_bytes, size_t *total_bytes) {
    ZIGZ_ENTER(ctx);
    if (!ctx || !free_bytes || !total_bytes) return ZIGZ_ERR_INVALID_ARGUMENT;
    HIPCHK(ctx, hipMemGetInfo(free_bytes, total_bytes));
    return ZIGZ_OK;
}
extern "C" zigz_status zigz_dev_free(zigz_ctx *ctx, void *d_ptr) {
    ZIGZ_ENTER(ctx);
    if (!ctx) return ZIGZ_ERR_INVALID_ARGUMENT;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipFree(d_ptr));
    return ZIGZ_OK;
}

extern "C" zigz_status zigz_ctx_enable_timing(zigz_ctx *ctx, int enable) {
    ZIGZ_ENTER(ctx);
    if (!ctx) return ZIGZ_ERR_INVALID_ARGUMENT;
    ctx->timing = enable != 0;
    if (enable) ctx->events_recorded = true;
    return ZIGZ_OK;
}
extern "C" zigz_status zigz_ctx_set_option(zigz_ctx *ctx, const char *name, int64_t value) {
    if (!ctx || !name) return ZIGZ_ERR_INVALID_ARGUMENT;
    if (strcmp(name, "per_round_sumcheck") == 0) { ctx->per_round_sumcheck = value != 0; return ZIGZ_OK; }
    if (strcmp(name, "fold_eval") == 0) { ctx->fold_eval = value != 0; return ZIGZ_OK; }
    if (strcmp(name, "merkle_dedup") == 0) { ctx->run_aware_mask = value != 0 ? ~0ull : 0; return ZIGZ_OK; }
    if (strcmp(name, "run_aware_mask") == 0) { ctx->run_aware_mask = (uint64_t)value; return ZIGZ_OK; }
    if (strcmp(name, "run_aware_materialize") == 0) { ctx->run_aware_materialize = value != 0; return ZIGZ_OK; }
    if (strcmp(name, "cons_group_mask") == 0) { ctx->cons_group_mask = (uint64_t)value; return ZIGZ_OK; }
    if (strcmp(name, "cons_always") == 0) { ctx->cons_always = value != 0; return ZIGZ_OK; }
    if (strcmp(name, "debug_skip") == 0) { ctx->debug_skip = (int)value; return ZIGZ_OK; }
    if (strcmp(name, "small_domain_mask") == 0) { ctx->small_domain_mask = (uint64_t)value; return ZIGZ_OK; }
    if (strcmp(name, "batch_reserve") == 0) { ctx->batch_reserve = value < 0 ? 0 : value > BATCH_MAX ? BATCH_MAX : (unsigned)value; return ZIGZ_OK; }
    return ZIGZ_ERR_INVALID_ARGUMENT;
}
extern "C" zigz_status zigz_ctx_get_option(zigz_ctx *ctx, const char *name, int64_t *value) {
    if (!ctx || !name || !value) return ZIGZ_ERR_INVALID_ARGUMENT;
    if (strcmp(name, "per_round_sumcheck") == 0) { *value = ctx->per_round_sumcheck; return ZIGZ_OK; }
    if (strcmp(name, "fold_eval") == 0) { *value = ctx->fold_eval; return ZIGZ_OK; }
    if (strcmp(name, "run_aware_mask") == 0) { *value = (int64_t)ctx->run_aware_mask; return ZIGZ_OK; }
    if (strcmp(name, "run_aware_materialize") == 0) { *value = ctx->run_aware_materialize; return ZIGZ_OK; }
    if (strcmp(name, "cons_group_mask") == 0) { *value = (int64_t)ctx->cons_group_mask; return ZIGZ_OK; }
    if (strcmp(name, "cons_always") == 0) { *value = ctx->cons_always; return ZIGZ_OK; }
    if (strcmp(name, "small_domain_mask") == 0) { *value = (int64_t)ctx->small_domain_mask; return ZIGZ_OK; }
    return ZIGZ_ERR_INVALID_ARGUMENT;
}
extern "C" zigz_status zigz_ctx_set_epoch(zigz_ctx *ctx, zigz_ctx *owner) {
    ZIGZ_ENTER(ctx);
    if (!ctx || !owner || owner->device != ctx->device) return ZIGZ_ERR_INVALID_ARGUMENT;
    if (owner == ctx) {
        HIPCHK(ctx, hipEventRecord(ctx->epoch_own, ctx->stream));
        HIPCHK(ctx, hipEventSynchronize(ctx->epoch_own));
    }
    ctx->epoch = owner->epoch_own;
    return ZIGZ_OK;
}
extern "C" zigz_status zigz_ctx_launch_log(zigz_ctx *ctx, zigz_launch_rec *out, size_t cap, size_t *n) {
    if (!ctx || !n || (cap && !out)) return ZIGZ_ERR_INVALID_ARGUMENT;
    const size_t k = (size_t)ctx->log_n < cap ? (size_t)ctx->log_n : cap;
    if (k) memcpy(out, ctx->log, k * sizeof(zigz_launch_rec));
    *n = (size_t)ctx->log_n;
    return ZIGZ_OK;
}
// one timed launch (both events stamped by the dispatch itself) -> the log; `first`: the job's first timed launch, the origin
// of the time axis when the context has no epoch
zigz_status log_launch(zigz_ctx *ctx, int cls, uint64_t perms, hipEvent_t start, hipEvent_t stop, hipEvent_t first, double *dur_us) {
    float d = 0, b = 0;
    HIPCHK(ctx, hipEventElapsedTime(&d, start, stop));
    *dur_us = (double)d * 1000.0;
    if (ctx->log_n >= KEV_MAX + 2) return ZIGZ_OK;
    const hipEvent_t origin = ctx->epoch ? ctx->epoch : first;
    if (origin == stop || hipEventElapsedTime(&b, origin, stop) != hipSuccess) {  // (origin == start of this very launch)
        (void)hipGetLastError();
        b = origin == start ? d : 0.0f;
    }
    zigz_launch_rec &r = ctx->log[ctx->log_n++];
    r.cls = (uint32_t)cls;
    r.reserved = 0;
    r.perms = perms;
    r.end_us = (double)b * 1000.0;
    r.start_us = r.end_us - *dur_us;
    return ZIGZ_OK;
}
extern "C" zigz_status zigz_ctx_get_stats(zigz_ctx *ctx, zigz_kernel_stats *out) {
    ZIGZ_ENTER(ctx);
    if (!ctx || !out) return ZIGZ_ERR_INVALID_ARGUMENT;
    *out = ctx->stats;
    return ZIGZ_OK;
}

// ------------------------------------------------------------------ boundary conversion
// canonical u64 host image -> packed u32 in device memory (validates < p)
zigz_status upload_u64(zigz_ctx *ctx, const uint64_t *h_in, size_t n, uint32_t *d_out, bool reduce) {
    if (n == 0) return ZIGZ_OK;
    void *d64;
    CHK(ws_get(ctx, WS_IN64, n * sizeof(uint64_t), &d64));
    HIPCHK(ctx, hipMemcpyAsync(d64, h_in, n * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    if (reduce) {
        launch_reduce_u64((const uint64_t *)d64, d_out, n, ctx->stream);
        HIPCHK(ctx, hipGetLastError());
        return ZIGZ_OK;
    }
    HIPCHK(ctx, hipMemsetAsync(ctx->d_flag, 0, 4, ctx->stream));
    launch_narrow_u64((const uint64_t *)d64, d_out, n, ctx->d_flag, ctx->stream);
    HIPCHK(ctx, hipGetLastError());
    uint32_t *hflag = (uint32_t *)ctx->h_pin;
    HIPCHK(ctx, hipMemcpyAsync(hflag, ctx->d_flag, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (*hflag) {
        set_err(ctx, "input contains a value >= p (not a canonical BabyBear element)");
        return ZIGZ_ERR_NOT_CANONICAL;
    }
    return ZIGZ_OK;
}

zigz_status download_u64(zigz_ctx *ctx, const uint32_t *d_in, size_t n, uint64_t *h_out) {
    if (n == 0) return ZIGZ_OK;
    void *d64;
    CHK(ws_get(ctx, WS_OUT64, n * sizeof(uint64_t), &d64));
    launch_widen_u32(d_in, (uint64_t *)d64, n, ctx->stream);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(h_out, d64, n * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return ZIGZ_OK;
}

extern "C" zigz_status zigz_dev_upload_u64(zigz_ctx *ctx, const uint64_t *h_in, size_t n, uint32_t *d_out) {
    ZIGZ_ENTER(ctx);
    if (!ctx || (n && (!h_in || !d_out))) return ZIGZ_ERR_INVALID_ARGUMENT;
    return upload_u64(ctx, h_in, n, d_out, false);
}
extern "C" zigz_status zigz_dev_reduce_u64(zigz_ctx *ctx, const uint64_t *h_in, size_t n, uint32_t *d_out) {
    ZIGZ_ENTER(ctx);
    if (!ctx || (n && (!h_in || !d_out))) return ZIGZ_ERR_INVALID_ARGUMENT;
    CHK(upload_u64(ctx, h_in, n, d_out, true));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return ZIGZ_OK;
}
extern "C" zigz_status zigz_dev_witness_from_rows(zigz_ctx *ctx, const uint64_t *h_rows, size_t num_steps, size_t nv,
                                                  uint32_t *d_cols, size_t col_stride) {
    ZIGZ_ENTER(ctx);
    if (!ctx || !h_rows || !d_cols || nv > 40) return ZIGZ_ERR_INVALID_ARGUMENT;
    if (num_steps == 0) return ZIGZ_ERR_EMPTY_TRACE;
    const size_t npad = (size_t)1 << nv;
    if (num_steps > npad || (nv > 0 && num_steps <= npad / 2) || col_stride < npad) return ZIGZ_ERR_INVALID_ARGUMENT;
    void *d_rows;
    CHK(ws_get(ctx, WS_IN64, num_steps * 43 * sizeof(uint64_t), &d_rows));
    HIPCHK(ctx, hipMemcpyAsync(d_rows, h_rows, num_steps * 43 * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    launch_witness_rows((const uint64_t *)d_rows, num_steps, npad, d_cols, col_stride, ctx->stream);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // h_rows may be pageable: do not return before the copy is done
    return ZIGZ_OK;
}

static_assert(sizeof(zigz_trace_step) == sizeof(TraceStep), "zigz_trace_step and its device mirror differ");

static zigz_status witness_from_steps(zigz_ctx *ctx, const zigz_trace_step *h_steps, size_t num_steps, size_t nv,
                                      const uint64_t *initial_regs, uint32_t *d_cols, size_t col_stride, bool wait) {
    if (!ctx || !h_steps || !d_cols || nv > 40) return ZIGZ_ERR_INVALID_ARGUMENT;
    if (num_steps == 0) return ZIGZ_ERR_EMPTY_TRACE;
    const size_t npad = (size_t)1 << nv;
    if (num_steps > npad || (nv > 0 && num_steps <= npad / 2) || col_stride < npad) return ZIGZ_ERR_INVALID_ARGUMENT;
    void *d_steps, *d_ws;
    CHK(ws_get(ctx, WS_IN64, num_steps * sizeof(zigz_trace_step), &d_steps));
    CHK(ws_get(ctx, WS_WITNESS, witness_steps_ws_words(npad) * 4, &d_ws));
    HIPCHK(ctx, hipMemcpyAsync(d_steps, h_steps, num_steps * sizeof(zigz_trace_step), hipMemcpyHostToDevice, ctx->stream));
    Regs32 init;
    for (int r = 0; r < 32; r++) init.v[r] = (r && initial_regs) ? (uint32_t)(initial_regs[r] % (uint64_t)P) : 0u;
    launch_witness_steps((const TraceStep *)d_steps, num_steps, npad, init, (uint32_t *)d_ws, d_cols, col_stride, ctx->stream);
    HIPCHK(ctx, hipGetLastError());
    if (wait) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // h_steps may be reused by the caller as soon as this returns
    return ZIGZ_OK;
}

extern "C" zigz_status zigz_dev_witness_from_steps(zigz_ctx *ctx, const zigz_trace_step *h_steps, size_t num_steps, size_t nv,
                                                   const uint64_t *initial_regs, uint32_t *d_cols, size_t col_stride) {
    ZIGZ_ENTER(ctx);
    return witness_from_steps(ctx, h_steps, num_steps, nv, initial_regs, d_cols, col_stride, true);
}
extern "C" zigz_status zigz_dev_witness_from_steps_async(zigz_ctx *ctx, const zigz_trace_step *h_steps, size_t num_steps,
                                                         size_t nv, const uint64_t *initial_regs, uint32_t *d_cols,
                                                         size_t col_stride) {
    ZIGZ_ENTER(ctx);
    return witness_from_steps(ctx, h_steps, num_steps, nv, initial_regs, d_cols, col_stride, false);
}

static_assert(sizeof(zigz_trace_step32) == sizeof(TraceStep32) && sizeof(zigz_mem_access) == sizeof(MemAccess), "32-byte record mirrors differ");
static zigz_status witness_from_steps32(zigz_ctx *ctx, const zigz_trace_step32 *h_steps, size_t num_steps, const zigz_mem_access *h_mem,
                                        size_t num_mem, size_t nv, const uint64_t *initial_regs, uint32_t *d_cols, size_t col_stride,
                                        bool wait) {
    if (!ctx || !h_steps || !d_cols || nv > 40 || (num_mem && !h_mem) || num_mem > 0xfffffffeull) return ZIGZ_ERR_INVALID_ARGUMENT;
    if (num_steps == 0) return ZIGZ_ERR_EMPTY_TRACE;
    const size_t npad = (size_t)1 << nv;
    if (num_steps > npad || (nv > 0 && num_steps <= npad / 2) || col_stride < npad) return ZIGZ_ERR_INVALID_ARGUMENT;
    // staging: [48-byte records the expansion reads | the 32-byte records as uploaded | the side list]
    const size_t wide_b = (num_steps * sizeof(zigz_trace_step) + 255) & ~(size_t)255, raw_b = (num_steps * 32 + 255) & ~(size_t)255;
    void *d_st, *d_ws;
    CHK(ws_get(ctx, WS_IN64, wide_b + raw_b + num_mem * 16 + 256, &d_st));
    CHK(ws_get(ctx, WS_WITNESS, witness_steps_ws_words(npad) * 4, &d_ws));
    uint8_t *q = (uint8_t *)d_st;
    HIPCHK(ctx, hipMemcpyAsync(q + wide_b, h_steps, num_steps * 32, hipMemcpyHostToDevice, ctx->stream));
    if (num_mem) HIPCHK(ctx, hipMemcpyAsync(q + wide_b + raw_b, h_mem, num_mem * 16, hipMemcpyHostToDevice, ctx->stream));
    launch_steps_widen((const TraceStep32 *)(q + wide_b), num_steps, (const MemAccess *)(q + wide_b + raw_b), num_mem, (TraceStep *)q,
                       ctx->stream);
    Regs32 init;
    for (int r = 0; r < 32; r++) init.v[r] = (r && initial_regs) ? (uint32_t)(initial_regs[r] % (uint64_t)P) : 0u;
    launch_witness_steps((const TraceStep *)q, num_steps, npad, init, (uint32_t *)d_ws, d_cols, col_stride, ctx->stream);
    HIPCHK(ctx, hipGetLastError());
    if (wait) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return ZIGZ_OK;
}
extern "C" zigz_status zigz_dev_witness_from_steps32(zigz_ctx *ctx, const zigz_trace_step32 *h_steps, size_t num_steps,
                                                     const zigz_mem_access *h_mem, size_t num_mem, size_t nv, const uint64_t *initial_regs,
                                                     uint32_t *d_cols, size_t col_stride) {
    ZIGZ_ENTER(ctx);
    return witness_from_steps32(ctx, h_steps, num_steps, h_mem, num_mem, nv, initial_regs, d_cols, col_stride, true);
}
extern "C" zigz_status zigz_dev_witness_from_steps32_ws(zigz_ctx *ctx, const zigz_trace_step32 *h_steps, size_t num_steps,
                                                        const zigz_mem_access *h_mem, size_t num_mem, size_t nv,
                                                        const uint64_t *initial_regs, const uint32_t **d_cols, size_t *col_stride) {
    ZIGZ_ENTER(ctx);
    if (!ctx || !d_cols || !col_stride || nv > 40) return ZIGZ_ERR_INVALID_ARGUMENT;
    if (ctx->active_job) return ZIGZ_ERR_BAD_STATE;
    const size_t N = (size_t)1 << nv, stride = N < 4 ? 4 : N;
    void *d;
    CHK(ws_get(ctx, WS_COLS, ZIGZ_NUM_COLUMNS * stride * 4, &d));
    CHK(witness_from_steps32(ctx, h_steps, num_steps, h_mem, num_mem, nv, initial_regs, (uint32_t *)d, stride, false));
    *d_cols = (const uint32_t *)d;
    *col_stride = stride;
    return ZIGZ_OK;
}

static_assert(sizeof(zigz_trace_step16) == sizeof(TraceStep16) && sizeof(zigz_code_entry) == sizeof(CodeEntry), "16-byte record mirrors differ");
static zigz_status witness_from_steps16(zigz_ctx *ctx, const zigz_trace_step16 *h_steps, size_t num_steps, const zigz_mem_access *h_mem,
                                        size_t num_mem, uint64_t code_base, const zigz_code_entry *h_code, size_t num_code, size_t nv,
                                        const uint64_t *initial_regs, uint32_t *d_cols, size_t col_stride, bool wait) {
    if (!ctx || !h_steps || !d_cols || nv > 40 || (num_mem && !h_mem) || num_mem >= ZIGZ_NO_MEM_ACCESS16 || (num_code && !h_code) ||
        num_code > ((size_t)1 << 30))
        return ZIGZ_ERR_INVALID_ARGUMENT;
    if (num_steps == 0) return ZIGZ_ERR_EMPTY_TRACE;
    const size_t npad = (size_t)1 << nv;
    if (num_steps > npad || (nv > 0 && num_steps <= npad / 2) || col_stride < npad) return ZIGZ_ERR_INVALID_ARGUMENT;
    // staging: [48-byte records the expansion reads | the 16-byte records as uploaded | the side list | the code table]
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t wide_b = al(num_steps * sizeof(zigz_trace_step)), raw_b = al(num_steps * 16), mem_b = al(num_mem * 16);
    void *d_st, *d_ws;
    CHK(ws_get(ctx, WS_IN64, wide_b + raw_b + mem_b + num_code * sizeof(zigz_code_entry) + 256, &d_st));
    CHK(ws_get(ctx, WS_WITNESS, witness_steps_ws_words(npad) * 4, &d_ws));
    uint8_t *q = (uint8_t *)d_st;
    HIPCHK(ctx, hipMemcpyAsync(q + wide_b, h_steps, num_steps * 16, hipMemcpyHostToDevice, ctx->stream));
    if (num_mem) HIPCHK(ctx, hipMemcpyAsync(q + wide_b + raw_b, h_mem, num_mem * 16, hipMemcpyHostToDevice, ctx->stream));
    if (num_code)
        HIPCHK(ctx, hipMemcpyAsync(q + wide_b + raw_b + mem_b, h_code, num_code * sizeof(zigz_code_entry), hipMemcpyHostToDevice, ctx->stream));
    launch_steps_widen16((const TraceStep16 *)(q + wide_b), num_steps, (const MemAccess *)(q + wide_b + raw_b), num_mem, code_base,
                         (const CodeEntry *)(q + wide_b + raw_b + mem_b), num_code, (TraceStep *)q, ctx->stream);
    Regs32 init;
    for (int r = 0; r < 32; r++) init.v[r] = (r && initial_regs) ? (uint32_t)(initial_regs[r] % (uint64_t)P) : 0u;
    launch_witness_steps((const TraceStep *)q, num_steps, npad, init, (uint32_t *)d_ws, d_cols, col_stride, ctx->stream);
    HIPCHK(ctx, hipGetLastError());
    if (wait) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return ZIGZ_OK;
}
extern "C" zigz_status zigz_dev_witness_from_steps16(zigz_ctx *ctx, const zigz_trace_step16 *h_steps, size_t num_steps,
                                                     const zigz_mem_access *h_mem, size_t num_mem, uint64_t code_base,
                                                     const zigz_code_entry *h_code, size_t num_code, size_t nv,
                                                     const uint64_t *initial_regs, uint32_t *d_cols, size_t col_stride) {
    ZIGZ_ENTER(ctx);
    return witness_from_steps16(ctx, h_steps, num_steps, h_mem, num_mem, code_base, h_code, num_code, nv, initial_regs, d_cols, col_stride, true);
}
extern "C" zigz_status zigz_dev_witness_from_steps16_ws(zigz_ctx *ctx, const zigz_trace_step16 *h_steps, size_t num_steps,
                                                        const zigz_mem_access *h_mem, size_t num_mem, uint64_t code_base,
                                                        const zigz_code_entry *h_code, size_t num_code, size_t nv,
                                                        const uint64_t *initial_regs, const uint32_t **d_cols, size_t *col_stride) {
    ZIGZ_ENTER(ctx);
    if (!ctx || !d_cols || !col_stride || nv > 40) return ZIGZ_ERR_INVALID_ARGUMENT;
    if (ctx->active_job) return ZIGZ_ERR_BAD_STATE;
    const size_t N = (size_t)1 << nv, stride = N < 4 ? 4 : N;
    void *d;
    CHK(ws_get(ctx, WS_COLS, ZIGZ_NUM_COLUMNS * stride * 4, &d));
    CHK(witness_from_steps16(ctx, h_steps, num_steps, h_mem, num_mem, code_base, h_code, num_code, nv, initial_regs, (uint32_t *)d, stride, false));
    *d_cols = (const uint32_t *)d;
    *col_stride = stride;
    return ZIGZ_OK;
}

extern "C" zigz_status zigz_dev_witness_from_steps_ws(zigz_ctx *ctx, const zigz_trace_step *h_steps, size_t num_steps, size_t nv,
                                                      const uint64_t *initial_regs, const uint32_t **d_cols, size_t *col_stride) {
    ZIGZ_ENTER(ctx);
    if (!ctx || !d_cols || !col_stride || nv > 40) return ZIGZ_ERR_INVALID_ARGUMENT;
    if (ctx->active_job) return ZIGZ_ERR_BAD_STATE;  // (the job may be reading the workspace)
    const size_t N = (size_t)1 << nv, stride = N < 4 ? 4 : N;
    void *d;
    CHK(ws_get(ctx, WS_COLS, ZIGZ_NUM_COLUMNS * stride * 4, &d));
    CHK(witness_from_steps(ctx, h_steps, num_steps, nv, initial_regs, (uint32_t *)d, stride, false));
    *d_cols = (const uint32_t *)d;
    *col_stride = stride;
    return ZIGZ_OK;
}

extern "C" zigz_status zigz_host_register(zigz_ctx *ctx, void *h_ptr, size_t bytes) {
    ZIGZ_ENTER(ctx);
    if (!ctx || !h_ptr || !bytes) return ZIGZ_ERR_INVALID_ARGUMENT;
    HIPCHK(ctx, hipHostRegister(h_ptr, bytes, hipHostRegisterDefault));
    return ZIGZ_OK;
}
extern "C" zigz_status zigz_host_unregister(zigz_ctx *ctx, void *h_ptr) {
    ZIGZ_ENTER(ctx);
    if (!h_ptr) return ZIGZ_ERR_INVALID_ARGUMENT;
    // page-locking belongs to the process, not to the context that asked for it: ctx may be NULL (or already destroyed by
    // the time a buffer is released -- callers then pass NULL)
    const hipError_t e = hipHostUnregister(h_ptr);
    if (e != hipSuccess) {
        if (ctx) set_err(ctx, "hipHostUnregister failed: %s", hipGetErrorString(e));
        return ZIGZ_ERR_HIP;
    }
    return ZIGZ_OK;
}

extern "C" zigz_status zigz_dev_download_u64(zigz_ctx *ctx, const uint32_t *d_in, size_t n, uint64_t *h_out) {
    ZIGZ_ENTER(ctx);
    if (!ctx || (n && (!h_out || !d_in))) return ZIGZ_ERR_INVALID_ARGUMENT;
    return download_u64(ctx, d_in, n, h_out);
}

zigz_status mle_check(size_t n) {  // Multilinear.init, multilinear.zig:36-44
    if (n == 0) return ZIGZ_ERR_EMPTY_EVALUATIONS;
    if (!is_pow2(n)) return ZIGZ_ERR_LENGTH_NOT_POWER_OF_TWO;
    return ZIGZ_OK;
}

uint32_t host_to_mont(uint64_t canonical) { return (uint32_t)((canonical << 32) % (uint64_t)P); }

// reads `words` u64 from device through the pinned staging buffer (synchronises the stream)
zigz_status read_u64(zigz_ctx *ctx, const void *d_src, size_t words, uint64_t *dst) {
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_pin, d_src, words * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(dst, ctx->h_pin, words * 8);
    return ZIGZ_OK;
}

// ------------------------------------------------------------------ host transcript
struct zigz_transcript {
    Transcript t;
};
extern "C" zigz_transcript *zigz_transcript_new(void) { return new (std::nothrow) zigz_transcript(); }
extern "C" void zigz_transcript_free(zigz_transcript *t) { delete t; }
extern "C" void zigz_transcript_append_bytes(zigz_transcript *t, const uint8_t *data, size_t len) {
    if (t && (data || !len)) t->t.append_bytes(data, len);
}
extern "C" void zigz_transcript_append_field(zigz_transcript *t, uint64_t v) {
    if (t) t->t.append_field(v);
}
extern "C" void zigz_transcript_append_tagged_counter(zigz_transcript *t, const uint8_t *tag, size_t tag_len,
                                                      uint64_t start, uint64_t count) {
    if (t) t->t.append_tagged_counter(tag, tag_len, start, count);
}
extern "C" uint64_t zigz_transcript_challenge(zigz_transcript *t) { return t ? t->t.challenge() : 0; }
extern "C" void zigz_sha3_256(const uint8_t *data, size_t len, uint8_t out[32]) { sha3_256(data, len, out); }
extern "C" void zigz_sha256(const uint8_t *data, size_t len, uint8_t out[32]) { sha256(data, len, out); }
extern "C" void zigz_host_sponge_servers(int n) { host_sponge_servers(n); }
extern "C" int zigz_host_sponge_batching(void) { return host_sponge_batching() ? 1 : 0; }
extern "C" void zigz_host_keccak_permute_x8(uint64_t *states) { host_keccak_permute_x8(reinterpret_cast<uint64_t (*)[8]>(states)); }
extern "C" const char *zigz_host_keccak_impl(void) { return host_keccak_impl(); }
extern "C" void zigz_host_keccak_permute(uint64_t state[25], int which) { host_keccak_permute(state, which); }

