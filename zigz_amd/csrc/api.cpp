// C ABI of libzigz_hip.so (include/zigz_hip.h): contexts, workspaces, boundary conversion and the
// orchestration of the gfx950 kernels.  No CPU fallback for field or hash work on the data path: the
// only host arithmetic is the sequential SHA3 Fiat-Shamir sponge (K11) and O(v) scalar bookkeeping.
#include "zigz_hip.h"

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <atomic>
#include <new>
#include <thread>
#include <vector>

#include "field.hpp"
#include "host_hash.hpp"
#include "kernels.hpp"

using namespace zk;

// ------------------------------------------------------------------ context
enum { WS_IN64 = 0, WS_IN32, WS_OUT32, WS_OUT64, WS_SCRATCH, WS_TREE, WS_FOLD, WS_MISC, WS_COLS, WS_LASSO, WS_DEDUP, WS_WITNESS, WS_RUNS, WS_RUNMETA, WS_CONS, WS_CONSMETA, WS_BATCH, WS_SLOTS };

constexpr int KEV_MAX = 72;
struct ListCaps {
    size_t npad;
    unsigned rn, gn;
    unsigned r[RUN_MAX_LEVELS], g[RUN_MAX_LEVELS];  // entries per sub-list and level
    bool g_slabs;  // this context's traces made the group be dropped: give its columns slabs up front
    unsigned g_drops, g_skip;  // consecutive builds that dropped the group; builds left that do not even try it
};
struct zigz_ctx {
    int device;
    hipStream_t own_stream;
    hipStream_t stream;
    char err[512];
    void *ws[WS_SLOTS];
    size_t ws_bytes[WS_SLOTS];
    unsigned long long *d_sums;  // SUMS_SLOTS u64
    uint32_t *d_flag;
    uint64_t *h_pin;  // pinned staging, PIN_WORDS u64
    uint64_t h_sums[2048];  // host copy of padded / replicated half sums (dev_half_sums)
    uint8_t *h_roots;  // pinned, ROOTS_MAX_COLS * 32 B: the active commit job's roots travel through this buffer ONLY, so any
                       // other call on the context between zigz_commit_begin* and zigz_commit_roots leaves them intact
    bool events_recorded;  // timing mode has been on: the context's events may still refer to launches (and so to their buffers)
    uint64_t done_seq;  // last sequence number handed to a launch that signals its completion in pinned memory (DoneFlag)
    bool timing;
    bool per_round_sumcheck;  // force the one-launch-per-round form (tests, A/B timing)
    bool fold_eval;           // force eval by v successive binds instead of the one-pass radix form
    uint64_t cons_group_mask;    // option: columns (bit c) that repeat in the same places -> content-addressed levels
    unsigned long long *d_cons_count;
    bool run_aware_materialize;  // option (tests): write the copies of every run-aware level (no virtual copies)
    bool cons_always;            // option (tests): try the content-addressed group in every job, however often it was dropped
    int debug_skip;              // option (measurement only, wrong trees): 1 = no level hashing / top, 2 = no structure passes
    uint64_t run_aware_mask;  // option: columns (bit c) whose Merkle levels are built run-aware (copies of the left neighbour
                              // are copied, not hashed); "merkle_dedup" = 1 is all columns
    unsigned long long *d_run_count;  // counters of the run-aware lists of the ACTIVE COMMIT JOB's build (and its "column is
                                      // not constant" words): read again by the job's openings, so nothing else adds to them
    unsigned long long *d_run_aux, *d_cons_aux;  // the same counters for builds outside a job (zigz_merkle_commit): a hinted
                                                 // build between a job's begin and its open_all must not clear the job's words
    hipEvent_t ev[6];
    hipEvent_t pool[2 * 64];  // per-launch event pairs timing the bulk MLE-bind launches (k_radix_fold / k_bind_vec)
    int pool_used;
    bool pool_is_fold;  // pool[0..1] carry the k_radix_fold launch of a commit job's eval
    uint64_t pool_bytes;
    // kernel-exact timestamps of the Keccak launches of the last batched commit (timing mode): pair i = kev[2i], kev[2i+1]
    hipEvent_t kev[2 * KEV_MAX];
    uint8_t kev_class[KEV_MAX];  // 0 leaves, 1 level (HPT hashes per thread), 2 level (1 hash per thread), 3 table look-ups,
                                 // 4 run-aware levels
    uint64_t kev_perms[KEV_MAX];
    int kev_n;
    // launch log of the last commit job (timing mode): begin / end of every timed launch since the epoch (zigz_ctx_set_epoch)
    hipEvent_t epoch_own, epoch;  // epoch: the event times are counted from (this context's or another's), or null
    zigz_launch_rec log[KEV_MAX + 2];
    int log_n;
    void *d_flush;          // 1 GiB read-only scratch of zigz_bench_kernel (cold-HBM runs), allocated on first use
    uint64_t small_domain_mask;  // option: columns (bit c) whose values are < 128 by construction -> levels 0-1 by table
    uint8_t *d_sd_tables;        // T0 | T1 (kernels.hpp SD_TABLE_BYTES), built on first use
    unsigned long long *d_sd_fallbacks;
    zigz_kernel_stats stats;
    zigz_commit_job *active_job;
    // content-addressing table of the last build (generation-tagged slots: cleared only when new or out of generations)
    void *cons_table;
    size_t cons_table_bytes;
    unsigned cons_gen;
    // what the last build asked for; turned into stats when its counters have arrived (zigz_commit_roots)
    uint64_t build_cons_hinted, build_cons_levels_nodes, build_cons_sd, build_top_perms;
    ListCaps caps;  // room for the lists of the structure-aware levels, learnt from earlier builds (caps_for)
    size_t batch_tab_S, batch_tab_off;  // the content-addressing tables of the batched jobs' arenas (WS_BATCH) as last cleared
    unsigned batch_tab_nz, batch_gen;
    unsigned batch_reserve;  // option: proofs to size the batched jobs' workspaces for (a service's largest batch), so that they
                             // are allocated once and not again when a larger batch than any before comes along
};
static const size_t FLUSH_BYTES = (size_t)1 << 30;
static const size_t SUMS_SLOTS = 8192;  // [0, 4096): results of the API calls; [4096, 8192): scratch of the measurement hook
constexpr unsigned RADIX_MAX_K = 10;     // 1024 block sums per radix sumcheck stage
constexpr size_t RADIX_MIN_N = 1 << 11;  // smaller tables use the per-round form (one launch + read-back per round)
constexpr size_t HOST_TAIL_MAX = 1024;
static const size_t PIN_WORDS = 1 << 19;  // 4 MiB: the openings of a batched job (32 proofs x 43 x (24 + 33 v) bytes) fit the zero-copy path
static const size_t ROOTS_MAX_COLS = 4096;
constexpr unsigned BATCH_MAX = 32;  // proofs per batched commit job (kernels.hpp: ColSrcs)

static void set_err(zigz_ctx *ctx, const char *fmt, ...) {
    if (!ctx) return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
    va_end(ap);
}

#define HIPCHK(ctx, call)                                                                        \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            set_err(ctx, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return e_ == hipErrorOutOfMemory ? ZIGZ_ERR_OUT_OF_MEMORY : ZIGZ_ERR_HIP;            \
        }                                                                                        \
    } while (0)
// host-side std::vector / std::string allocations must not throw through the C ABI
#define ZIGZ_NOTHROW_BEGIN try {
#define ZIGZ_NOTHROW_END(ctx)                                         \
    }                                                                 \
    catch (const std::bad_alloc &) {                                  \
        set_err(ctx, "host allocation failed");                       \
        return ZIGZ_ERR_OUT_OF_MEMORY;                                \
    }
#define CHK(expr)                          \
    do {                                   \
        zigz_status s_ = (expr);           \
        if (s_ != ZIGZ_OK) return s_;      \
    } while (0)

// HIP's current device is per thread: make the context's device current for the calling thread (multi-GPU
// ranks that see every device, helper threads, hosts that also drive torch on another device).
#define ZIGZ_ENTER(ctx)                                                              \
    do {                                                                             \
        if (ctx) {                                                                   \
            int d_ = -1;                                                             \
            if (hipGetDevice(&d_) != hipSuccess || d_ != (ctx)->device) (void)hipSetDevice((ctx)->device); \
        }                                                                            \
    } while (0)

static bool is_pow2(size_t n) { return n && !(n & (n - 1)); }
static unsigned log2_floor(size_t n) { unsigned l = 0; while (n > 1) { n >>= 1; l++; } return l; }
static size_t ceil_pow2(size_t n) { size_t v = 1; while (v < n) v <<= 1; return v; }

static zigz_status ws_get(zigz_ctx *ctx, int slot, size_t bytes, void **out) {
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
static std::atomic<int> g_sleep_wait{0};
static bool sleep_wait(const unsigned long long *flag, unsigned long long seq) {
    long ns = 30000;
    for (int i = 0; i < 20000; i++) {  // ~3 s, then the caller asks the runtime (which also reports a fault)
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return true;
        timespec ts{0, ns};
        nanosleep(&ts, nullptr);
        if (ns < 150000) ns += ns / 2;
    }
    return false;
}
static DoneFlag done_flag(zigz_ctx *ctx, int which) {  // which: 0 = the roots of a commit job, 1 = its openings
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
static zigz_status log_launch(zigz_ctx *ctx, int cls, uint64_t perms, hipEvent_t start, hipEvent_t stop, hipEvent_t first, double *dur_us) {
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
static zigz_status upload_u64(zigz_ctx *ctx, const uint64_t *h_in, size_t n, uint32_t *d_out, bool reduce) {
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

static zigz_status download_u64(zigz_ctx *ctx, const uint32_t *d_in, size_t n, uint64_t *h_out) {
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

static zigz_status mle_check(size_t n) {  // Multilinear.init, multilinear.zig:36-44
    if (n == 0) return ZIGZ_ERR_EMPTY_EVALUATIONS;
    if (!is_pow2(n)) return ZIGZ_ERR_LENGTH_NOT_POWER_OF_TWO;
    return ZIGZ_OK;
}

static inline uint32_t host_to_mont(uint64_t canonical) { return (uint32_t)((canonical << 32) % (uint64_t)P); }

// reads `words` u64 from device through the pinned staging buffer (synchronises the stream)
static zigz_status read_u64(zigz_ctx *ctx, const void *d_src, size_t words, uint64_t *dst) {
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_pin, d_src, words * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(dst, ctx->h_pin, words * 8);
    return ZIGZ_OK;
}

// ------------------------------------------------------------------ device-resident MLE ops
static zigz_status dev_half_sums(zigz_ctx *ctx, const uint32_t *d_in, size_t n, uint64_t out[2]) {
    // one big table: thousands of waves add into two counters; they get cache lines of their own and up to 64 copies,
    // added here (<= 2^40 elements < 2^31 each: the u64 totals cannot overflow)
    const SumsLayout lay = (aligned16(d_in)) ? half_sums_layout(n, 1, 2048) : SumsLayout{2, 1, 0, 1};
    const size_t words = lay.nslots > 1 || lay.col_stride != 2 ? (size_t)lay.nslots * 32 : 2;
    HIPCHK(ctx, hipMemsetAsync(ctx->d_sums, 0, words * 8, ctx->stream));
    launch_half_sums(d_in, n, n, 1, ctx->d_sums, ctx->stream, nullptr, &lay);
    HIPCHK(ctx, hipGetLastError());
    CHK(read_u64(ctx, ctx->d_sums, words, ctx->h_sums));
    out[0] = out[1] = 0;
    for (unsigned k = 0; k < lay.nslots; k++) {
        out[0] += ctx->h_sums[k * lay.slot_stride];
        out[1] += ctx->h_sums[k * lay.slot_stride + lay.bin_stride];
    }
    return ZIGZ_OK;
}

extern "C" zigz_status zigz_dev_mle_half_sums(zigz_ctx *ctx, const uint32_t *d_in, size_t n, uint64_t half_sums[2]) {
    ZIGZ_ENTER(ctx);
    if (!ctx || !d_in || !half_sums) return ZIGZ_ERR_INVALID_ARGUMENT;
    CHK(mle_check(n));
    uint64_t s[2];
    CHK(dev_half_sums(ctx, d_in, n, s));
    half_sums[0] = s[0] % P;
    half_sums[1] = s[1] % P;
    return ZIGZ_OK;
}

static zigz_status timed_begin(zigz_ctx *ctx, int ev) {
    if (ctx->timing) HIPCHK(ctx, hipEventRecord(ctx->ev[ev], ctx->stream));
    return ZIGZ_OK;
}
static zigz_status timed_end(zigz_ctx *ctx, int ev, double *us_out) {
    if (ctx->timing) {
        HIPCHK(ctx, hipEventRecord(ctx->ev[ev + 1], ctx->stream));
        HIPCHK(ctx, hipEventSynchronize(ctx->ev[ev + 1]));
        float ms = 0;
        HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev[ev], ctx->ev[ev + 1]));
        *us_out = (double)ms * 1000.0;
    }
    return ZIGZ_OK;
}

// launch_bind with a private HIP event pair around every vector-path launch (timing mode only)
static zigz_status bind_launch(zigz_ctx *ctx, const uint32_t *d_in, size_t in_stride, uint32_t *d_out, size_t out_stride,
                               size_t half, size_t ncols, uint32_t r_m, const uint32_t *d_r_m, unsigned long long *d_sums,
                               const SumsLayout *lay = nullptr) {
    const bool rec = ctx->timing && ctx->pool_used < 64 && bind_uses_vec(half, d_sums != nullptr, in_stride, out_stride, d_in, d_out);
    if (rec) HIPCHK(ctx, hipEventRecord(ctx->pool[2 * ctx->pool_used], ctx->stream));
    launch_bind(d_in, in_stride, d_out, out_stride, half, ncols, r_m, d_r_m, d_sums, ctx->stream, nullptr, lay);
    if (rec) {
        HIPCHK(ctx, hipEventRecord(ctx->pool[2 * ctx->pool_used + 1], ctx->stream));
        ctx->pool_used++;
        ctx->pool_bytes += (uint64_t)ncols * half * 2 * 6;  // table of 2*half u32: read 8*half B, write 4*half B
    }
    return ZIGZ_OK;
}
static void bind_pool_reset(zigz_ctx *ctx) {
    ctx->pool_used = 0;
    ctx->pool_bytes = 0;
}
// call after the stream has been synchronised past the last recorded launch
static zigz_status bind_pool_collect(zigz_ctx *ctx) {
    if (!ctx->timing) return ZIGZ_OK;
    double us = 0;
    for (int i = 0; i < ctx->pool_used; i++) {
        float ms = 0;
        HIPCHK(ctx, hipEventSynchronize(ctx->pool[2 * i + 1]));
        if (ctx->pool_is_fold && i == 0) {  // the eval's k_radix_fold of a commit job: into the job's launch log as well
            double d = 0;
            CHK(log_launch(ctx, 7, 0, ctx->pool[0], ctx->pool[1], ctx->pool[0], &d));
            us += d;
            continue;
        }
        HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->pool[2 * i], ctx->pool[2 * i + 1]));
        us += (double)ms * 1000.0;
    }
    ctx->pool_is_fold = false;
    ctx->stats.bind_vec_us = us;
    ctx->stats.bind_vec_launches = (uint64_t)ctx->pool_used;
    ctx->stats.bind_vec_bytes = ctx->pool_bytes;
    return ZIGZ_OK;
}

// bind of ONE table fused with the half sums of the result: the counters of a large table are padded and replicated
// (k_bind_vec<true> adds one partial sum per workgroup), read back through the pinned buffer and added here
static zigz_status bind_with_sums(zigz_ctx *ctx, const uint32_t *d_in, size_t n, uint32_t *d_out, uint32_t r_m, uint64_t out[2]) {
    const size_t half = n / 2;
    const bool vec = bind_uses_vec(half, true, n, half, d_in, d_out);
    const SumsLayout lay = vec ? bind_sums_layout(half, 1, 2048) : SumsLayout{2, 1, 0, 1};
    const size_t words = lay.col_stride != 2 ? (size_t)lay.nslots * 32 : 2;
    unsigned long long *d_s = ctx->d_sums + 4096;  // the scratch half of d_sums: [0, 4096) holds per-round results
    HIPCHK(ctx, hipMemsetAsync(d_s, 0, words * 8, ctx->stream));
    CHK(bind_launch(ctx, d_in, n, d_out, half, half, 1, r_m, nullptr, d_s, &lay));
    HIPCHK(ctx, hipGetLastError());
    CHK(read_u64(ctx, d_s, words, ctx->h_sums));
    out[0] = out[1] = 0;
    for (unsigned k = 0; k < lay.nslots; k++) {
        out[0] += ctx->h_sums[k * lay.slot_stride];
        out[1] += ctx->h_sums[k * lay.slot_stride + lay.bin_stride];
    }
    return ZIGZ_OK;
}

extern "C" zigz_status zigz_dev_mle_bind(zigz_ctx *ctx, const uint32_t *d_in, size_t n, uint64_t r, uint32_t *d_out) {
    ZIGZ_ENTER(ctx);
    if (!ctx || !d_in || !d_out) return ZIGZ_ERR_INVALID_ARGUMENT;
    CHK(mle_check(n));
    if (n == 1) return ZIGZ_ERR_NO_VARIABLES_TO_FIX;
    if (r >= P) return ZIGZ_ERR_NOT_CANONICAL;
    bind_pool_reset(ctx);
    CHK(timed_begin(ctx, 0));
    CHK(bind_launch(ctx, d_in, n, d_out, n / 2, n / 2, 1, host_to_mont(r), nullptr, nullptr));
    HIPCHK(ctx, hipGetLastError());
    CHK(timed_end(ctx, 0, &ctx->stats.bind_us));
    CHK(bind_pool_collect(ctx));
    ctx->stats.bind_launches = 1;
    return ZIGZ_OK;
}

extern "C" zigz_status zigz_dev_mle_bind_sums(zigz_ctx *ctx, const uint32_t *d_in, size_t n, uint64_t r,
                                              uint32_t *d_out, uint64_t half_sums[2]) {
    ZIGZ_ENTER(ctx);
    if (!ctx || !d_in || !d_out || !half_sums) return ZIGZ_ERR_INVALID_ARGUMENT;
    CHK(mle_check(n));
    if (n == 1) return ZIGZ_ERR_NO_VARIABLES_TO_FIX;
    if (r >= P) return ZIGZ_ERR_NOT_CANONICAL;
    bind_pool_reset(ctx);
    uint64_t s[2];
    CHK(timed_begin(ctx, 0));
    CHK(bind_with_sums(ctx, d_in, n, d_out, host_to_mont(r), s));
    CHK(timed_end(ctx, 0, &ctx->stats.bind_us));
    CHK(bind_pool_collect(ctx));
    ctx->stats.bind_launches = 1;
    half_sums[0] = s[0] % P;
    half_sums[1] = s[1] % P;
    return ZIGZ_OK;
}

// eval(point) for tables >= 2^14, batched over columns, in ONE pass over the data: the first k1 = v - 10 variables
// (MSB side, i.e. point[v-1] ... point[10]) are bound by a radix-2^k1 fold with eq weights built on the device,
// leaving 1024 elements per column that a weighted dot product with the eq weights of point[9..0] finishes.
// HBM traffic 4*N B per column instead of 12*N for v successive binds.  Exact arithmetic => same value.
static zigz_status dev_eval_radix(zigz_ctx *ctx, const uint32_t *d_cols, size_t col_stride, size_t ncols, size_t nv,
                                  const uint64_t *points, uint32_t *d_vals, const EvalSkip *skip = nullptr) {
    const size_t N = (size_t)1 << nv;
    const unsigned k2 = 10, k1 = (unsigned)nv - k2;
    const size_t m = (size_t)1 << k2, nb = (size_t)1 << k1;
    // a thread folds rloops x 16 rows: fewer when most columns are skipped, so that the launch still fills the chip (13 of 43
    // columns x 16 groups are 208 workgroups on 256 CUs)
    const size_t active = skip ? ncols - (size_t)ctx->stats.eval_constant_columns : ncols;
    const int rloops = active * 2 <= ncols && nb % 16 == 0 ? (active * 4 <= ncols + 3 ? 1 : 2) : 4;
    const size_t groups = radix_fold_groups(nb, rloops);
    if (nv * ncols * 4 > PIN_WORDS * 8 / 2) return ZIGZ_ERR_INVALID_ARGUMENT;
    uint32_t *rt = (uint32_t *)(ctx->h_pin + PIN_WORDS / 2);  // [col][j], j-th bound variable = point[v-1-j]
    for (size_t c = 0; c < ncols; c++)
        for (size_t j = 0; j < nv; j++) {
            const uint64_t r = points[c * nv + (nv - 1 - j)];
            if (r >= P) return ZIGZ_ERR_NOT_CANONICAL;
            rt[c * nv + j] = host_to_mont(r);
        }
    // zero-copy: k_eq_weights reads the few KB of points from the pinned buffer (both callers wait for the stream before they
    // return, so the buffer is not rewritten under it)
    const uint32_t *d_rt = rt;
    // workspace: part[ncols][groups][m] u64 | W1[ncols][nb] u32 | W2[ncols][m] u32 | T1[ncols][m] u32
    void *ws;
    CHK(ws_get(ctx, WS_FOLD, ncols * (groups * m * 8 + nb * 4 + m * 4 + m * 4) + 256, &ws));
    unsigned long long *d_part = (unsigned long long *)ws;
    uint32_t *d_w1 = (uint32_t *)(d_part + ncols * groups * m), *d_w2 = d_w1 + ncols * nb, *d_t1 = d_w2 + ncols * m;
    launch_eq_weights2(d_rt, nv, k1, d_w1, nb, k2, d_w2, m, ncols, ctx->stream);  // (one launch for both tables)
    bind_pool_reset(ctx);
    const bool rec = ctx->timing;
    // the one pass over the data; in timing mode the events carry the dispatch's own begin/end timestamps
    launch_radix_fold(d_cols, col_stride, m, nb, d_w1, nb, d_part, groups * m, ncols, ctx->stream, rec ? ctx->pool[0] : nullptr,
                      rec ? ctx->pool[1] : nullptr, skip, rloops);
    if (rec) {
        ctx->pool_used = 1;
        ctx->pool_is_fold = ctx->active_job != nullptr;
        // one read of the tables (those of the columns that are not skipped) + the partial sums
        ctx->pool_bytes = (uint64_t)(ncols - (skip ? ctx->stats.eval_constant_columns : 0)) * (N * 4 + groups * m * 8);
    }
    // (finalize and dot stay two launches: fused into one workgroup per column they took 60-69 us in a batch against 13 + 9 --
    // a column's 64 groups summed by ONE workgroup instead of four)
    launch_radix_finalize(d_part, groups * m, groups, d_t1, m, m, 0, nullptr, ncols, ctx->stream, skip);
    launch_weighted_dot(d_t1, m, d_w2, m, m, d_vals, ncols, ctx->stream, skip, d_cols, col_stride);
    HIPCHK(ctx, hipGetLastError());
    return ZIGZ_OK;
}

// eval(point), multilinear.zig:110-144: point[0] <-> LSB.  Computed as v MSB-first binds with the
// point reversed (exact arithmetic => the same canonical value as the reference's O(v*2^v) loop).
// Batched over `ncols` columns, column c using point row c.  Result words land in d_vals[ncols].
static zigz_status dev_eval_folds(zigz_ctx *ctx, const uint32_t *d_cols, size_t col_stride, size_t ncols, size_t nv,
                                  const uint64_t *points /*host, ncols*nv*/, uint32_t *d_vals, const EvalSkip *skip = nullptr) {
    const size_t N = (size_t)1 << nv;
    if (nv == 0) {
        launch_gather_first(d_cols, col_stride, d_vals, ncols, ctx->stream);
        HIPCHK(ctx, hipGetLastError());
        return ZIGZ_OK;
    }
    if (nv >= 14 && nv <= 24 && col_stride % 4 == 0 && aligned16(d_cols) && !ctx->fold_eval)
        return dev_eval_radix(ctx, d_cols, col_stride, ncols, nv, points, d_vals, skip);
    // r table in Montgomery form, [round][col], staged in the upper half of the pinned buffer so the
    // asynchronous H2D copy never reads freed host memory
    if (nv * ncols * 4 > PIN_WORDS * 8 / 2) return ZIGZ_ERR_INVALID_ARGUMENT;
    uint32_t *rt = (uint32_t *)(ctx->h_pin + PIN_WORDS / 2);
    for (size_t k = 0; k < nv; k++)
        for (size_t c = 0; c < ncols; c++) {
            uint64_t r = points[c * nv + (nv - 1 - k)];
            if (r >= P) return ZIGZ_ERR_NOT_CANONICAL;
            rt[k * ncols + c] = host_to_mont(r);
        }
    void *d_rt;
    CHK(ws_get(ctx, WS_MISC, nv * ncols * 4 + 64, &d_rt));
    HIPCHK(ctx, hipMemcpyAsync(d_rt, rt, nv * ncols * 4, hipMemcpyHostToDevice, ctx->stream));
    void *fold;
    // (a column of fewer than 4 elements still takes 4 -- dst_stride below -- so that every column stays 16-byte aligned: the two
    // buffers are sized with that stride.  Sized by the element counts alone, N = 4 let round 1 write its results over the
    // columns round 1 was still reading whenever ncols * 2 was a multiple of 4: a batched job of 16 x 43 columns found it)
    const size_t a_elems = ncols * (N / 2 < 4 ? 4 : N / 2), b_elems = ncols * (N / 4 < 4 ? 4 : N / 4);
    CHK(ws_get(ctx, WS_FOLD, (a_elems + b_elems) * 4, &fold));
    uint32_t *bufA = (uint32_t *)fold, *bufB = bufA + a_elems;
    bind_pool_reset(ctx);
    const uint32_t *src = d_cols;
    size_t src_stride = col_stride, len = N;
    for (size_t k = 0; k < nv; k++) {
        uint32_t *dst = (k % 2 == 0) ? bufA : bufB;
        size_t half = len / 2;
        size_t dst_stride = half < 4 ? 4 : half;  // keep 16-byte alignment of every column
        CHK(bind_launch(ctx, src, src_stride, dst, dst_stride, half, ncols, 0, (const uint32_t *)d_rt + k * ncols, nullptr));
        src = dst;
        src_stride = dst_stride;
        len = half;
    }
    HIPCHK(ctx, hipGetLastError());
    launch_gather_first(src, src_stride, d_vals, ncols, ctx->stream);
    HIPCHK(ctx, hipGetLastError());
    return ZIGZ_OK;
}

extern "C" zigz_status zigz_dev_mle_eval(zigz_ctx *ctx, const uint32_t *d_in, size_t n, const uint64_t *point,
                                         size_t point_len, uint64_t *out) {
    ZIGZ_ENTER(ctx);
    if (!ctx || !d_in || !out || (point_len && !point)) return ZIGZ_ERR_INVALID_ARGUMENT;
    CHK(mle_check(n));
    if (point_len != log2_floor(n)) return ZIGZ_ERR_WRONG_NUMBER_OF_VARIABLES;
    void *misc;
    CHK(ws_get(ctx, WS_OUT32, 64, &misc));
    CHK(dev_eval_folds(ctx, d_in, n, 1, point_len, point, (uint32_t *)misc));
    uint32_t *h = (uint32_t *)ctx->h_pin;
    HIPCHK(ctx, hipMemcpyAsync(h, misc, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *out = h[0];
    return ZIGZ_OK;
}

static zigz_status sumcheck_radix(zigz_ctx *ctx, const uint32_t *d_in, size_t n, const uint64_t *fixed, uint64_t *rounds,
                                  uint64_t *point, uint64_t *final_eval);

// ------------------------------------------------------------------ sumcheck (device-resident core)
// SumcheckProver.prove, sumcheck_prover.zig:26-91.  Per round: [s0, s1-s0] -> host transcript ->
// challenge -> fused bind + next-round half sums (one launch, one 16-byte read-back per round).
static zigz_status sumcheck_core(zigz_ctx *ctx, const uint32_t *d_in, size_t n, uint32_t *d_scratch,
                                 const uint64_t *fixed, uint64_t *rounds, uint64_t *point, uint64_t *final_eval) {
    const size_t nv = log2_floor(n);
    if (2 * (nv + 1) > 4096) return ZIGZ_ERR_INVALID_ARGUMENT;
    if (n >= RADIX_MIN_N && aligned16(d_in) && !ctx->per_round_sumcheck && !ctx->timing)
        return sumcheck_radix(ctx, d_in, n, fixed, rounds, point, final_eval);
    if (!d_scratch) {
        void *s;
        CHK(ws_get(ctx, WS_SCRATCH, (n / 2 + n / 4 + 8) * 4, &s));
        d_scratch = (uint32_t *)s;
    }
    uint32_t *bufA = d_scratch, *bufB = d_scratch + n / 2;
    HIPCHK(ctx, hipMemsetAsync(ctx->d_sums, 0, 2 * (nv + 1) * sizeof(unsigned long long), ctx->stream));
    launch_half_sums(d_in, n, n, 1, ctx->d_sums, ctx->stream);  // K3/K2 for round 0 (sum = s0+s1, prover:39)
    HIPCHK(ctx, hipGetLastError());
    uint64_t s[2];
    CHK(read_u64(ctx, ctx->d_sums, 2, s));
    Transcript tr;  // fresh transcript per sumcheck, sumcheck_protocol.zig:161
    bind_pool_reset(ctx);
    const uint32_t *cur = d_in;
    size_t len = n;
    double bind_us = 0;
    for (size_t round = 0; round < nv; round++) {
        uint64_t c0 = s[0] % P, s1 = s[1] % P;
        uint64_t c1 = s1 >= c0 ? s1 - c0 : s1 + P - c0;  // roundPolynomial: [q(0), q(1)-q(0)], multilinear.zig:228-229
        rounds[2 * round] = c0;
        rounds[2 * round + 1] = c1;
        uint64_t ch;
        if (fixed) {
            ch = fixed[round];
            if (ch >= P) return ZIGZ_ERR_NOT_CANONICAL;
        } else {
            tr.append_field(c0);  // generateChallenge, sumcheck_protocol.zig:176-184
            tr.append_field(c1);
            ch = tr.challenge();
        }
        point[round] = ch;
        uint32_t *dst = (round % 2 == 0) ? bufA : bufB;
        const bool last = (len == 2);
        if (ctx->timing) CHK(timed_begin(ctx, 0));
        if (last) {
            CHK(bind_launch(ctx, cur, len, dst, len / 2, len / 2, 1, host_to_mont(ch), nullptr, nullptr));
            HIPCHK(ctx, hipGetLastError());
        } else {
            CHK(bind_with_sums(ctx, cur, len, dst, host_to_mont(ch), s));  // bind + the next round's half sums, read back
        }
        if (ctx->timing) {
            double us = 0;
            CHK(timed_end(ctx, 0, &us));
            bind_us += us;
        }
        cur = dst;
        len /= 2;
    }
    if (len != 1) return ZIGZ_ERR_PROTOCOL_ERROR;  // sumcheck_prover.zig:80-82
    uint32_t *h = (uint32_t *)ctx->h_pin;
    HIPCHK(ctx, hipMemcpyAsync(h, cur, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *final_eval = h[0];
    if (ctx->timing) {
        ctx->stats.bind_us = bind_us;
        ctx->stats.bind_launches = nv;
        CHK(bind_pool_collect(ctx));
    }
    return ZIGZ_OK;
}

// ------------------------------------------------------------------ sumcheck, radix-2^k form
// The per-round form above costs one launch + one host round trip per round (~17 us each on MI355X), which
// dwarfs the HBM time of even a 2^24 table.  Round polynomials only need HALF SUMS of the bound table, and
// binding is linear, so the sums of the next k rounds follow from the 2^k block sums of the current table:
//   pass 1  GPU: block sums B[2^k] of the table (one read of the table)
//   host    k rounds on the 2^k-entry sums table (SHA3 challenge per round, O(2^k) scalar field ops in total)
//   pass 2  GPU: T'[i] = sum_b eq(r_0..r_{k-1}; b) * T[b*m + i]  (one more read, writes n/2^k) + next block sums
// i.e. two passes over the table per k <= 10 rounds and two host round trips instead of k.  The O(n) data work
// stays on the GPU; the host touches only the <= 1024-entry sums tables (and the final <= 1024-entry table).
// Exact field arithmetic => identical round polynomials, challenges and final_eval (tests compare both forms).
namespace {
inline uint64_t h_add(uint64_t a, uint64_t b) { uint64_t s = a + b; return s >= P ? s - P : s; }
inline uint64_t h_sub(uint64_t a, uint64_t b) { return a >= b ? a - b : a + P - b; }
inline uint64_t h_mul(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) % P); }
}  // namespace

// The radix sumcheck as orchestration over three data passes (RadixOps) and, when the table is sharded by rows over
// several GPUs, one exchange hook.  Row sharding: global index i lives on rank i mod G at local index
// i / G, so the MSB-first bind pairs (i, i + n/2) of the first v - log2 G rounds are rank-local, the top k index bits
// of i are the top k bits of the local index -- a rank's block sums are its share of the global block sums -- and the
// fold T'[i'] = sum_b eq_b T[b*m + i'] is rank-local too.  Per stage of k <= 10 rounds the ranks exchange 2^k <= 1024
// exact u64 partial sums (ONE all-gather, added locally = an all-reduce), and once the local tables are <= 1024
// entries one all-gather re-assembles the remaining table (local index j of rank g -> global index j*G + g) that
// every rank finishes identically.  2-3 exchanges per proof instead of one per round; transcripts run in lockstep.
namespace {
struct RadixOps {
    void *user;
    // exact u64 sums of the 2^k contiguous blocks of the current local table
    zigz_status (*block_sums)(void *user, unsigned k, uint64_t *sums);
    // current := fold of the current table with the 2^k canonical weights (length / 2^k entries); when k_next != 0 also
    // the exact u64 sums of the 2^k_next blocks of the result
    zigz_status (*fold)(void *user, unsigned k, const uint64_t *weights, unsigned k_next, uint64_t *next_sums);
    // the current local table (m canonical values)
    zigz_status (*read_tail)(void *user, size_t m, uint64_t *out);
};
struct ShardComm {
    int rank, world;
    zigz_allgather_fn allgather;
    void *user;
    bool sums_global;  // the data passes already return the sums over ALL ranks (reduced on the device: RCCL all-reduce)
};

zigz_status radix_run(zigz_ctx *ctx, const RadixOps &ops, size_t n_local, const ShardComm *comm, const uint64_t *fixed,
                      uint64_t *rounds, uint64_t *point, uint64_t *final_eval) {
    ZIGZ_NOTHROW_BEGIN
    const size_t world = comm && comm->world > 1 ? (size_t)comm->world : 1;
    const unsigned nv = log2_floor(n_local) + log2_floor(world);
    Transcript tr;  // fresh transcript per sumcheck, sumcheck_protocol.zig:161
    size_t round = 0;
    auto next_challenge = [&](uint64_t c0, uint64_t c1, uint64_t *ch) -> zigz_status {
        rounds[2 * round] = c0;
        rounds[2 * round + 1] = c1;
        if (fixed) {
            if (fixed[round] >= P) return ZIGZ_ERR_NOT_CANONICAL;
            *ch = fixed[round];
        } else {
            tr.append_field(c0);  // generateChallenge, sumcheck_protocol.zig:176-184
            tr.append_field(c1);
            *ch = tr.challenge();
        }
        point[round++] = *ch;
        return ZIGZ_OK;
    };
    std::vector<uint64_t> gather, wire;
    // One exchange: every rank contributes `v` (all ranks the same length) behind ONE status word.  A rank whose local pass
    // failed still takes part -- with its status and a zero payload -- so that all ranks leave the proof at the same
    // exchange: the failing rank with its own error, the others with ZIGZ_ERR_COMM (instead of sitting in the transport's
    // timeout while the failed rank has long returned).
    zigz_status local = ZIGZ_OK;
    auto exchange = [&](const std::vector<uint64_t> &v) -> zigz_status {  // gather := world x v
        const size_t n = v.size();
        wire.assign(n + 1, 0);
        wire[0] = (uint64_t)(uint32_t)local;
        if (local == ZIGZ_OK) memcpy(wire.data() + 1, v.data(), n * 8);
        std::vector<uint64_t> all(world * (n + 1));
        if (!comm->allgather || comm->allgather(comm->user, wire.data(), (n + 1) * 8, all.data()) != 0) {
            set_err(ctx, "sharded sumcheck: the all-gather hook failed");
            return local != ZIGZ_OK ? local : ZIGZ_ERR_COMM;
        }
        gather.resize(world * n);
        bool peer_failed = false;
        for (size_t r = 0; r < world; r++) {
            if (all[r * (n + 1)] != ZIGZ_OK) peer_failed = true;
            memcpy(gather.data() + r * n, all.data() + r * (n + 1) + 1, n * 8);
        }
        if (local != ZIGZ_OK) return local;
        if (peer_failed) {
            set_err(ctx, "sharded sumcheck: another rank reported an error");
            return ZIGZ_ERR_COMM;
        }
        return ZIGZ_OK;
    };
    // a local data pass: alone, its status is returned at once; sharded, it is carried into the next exchange
#define ZK_LOCAL(expr)                                     \
    do {                                                   \
        if (local == ZIGZ_OK) local = (expr);              \
        if (local != ZIGZ_OK && world == 1) return local;  \
    } while (0)
    // partial sums of every rank -> totals (exact: < 2^31 * 2^40 per rank, a few ranks)
    auto sum_over_ranks = [&](std::vector<uint64_t> &v) -> zigz_status {
        if (world == 1 || comm->sums_global) return ZIGZ_OK;
        CHK(exchange(v));
        for (size_t i = 0; i < v.size(); i++) {
            uint64_t t = 0;
            for (size_t r = 0; r < world; r++) t += gather[r * v.size() + i];
            v[i] = t;
        }
        return ZIGZ_OK;
    };
    std::vector<uint64_t> B, W, tail;
    size_t len = n_local;
    if (len > HOST_TAIL_MAX) {
        unsigned k = log2_floor(len) - 8 < RADIX_MAX_K ? log2_floor(len) - 8 : RADIX_MAX_K;
        B.assign((size_t)1 << k, 0);
        ZK_LOCAL(ops.block_sums(ops.user, k, B.data()));
        CHK(sum_over_ranks(B));
        for (;;) {
            for (auto &b : B) b %= P;
            W.assign(1, 1);
            for (unsigned j = 0; j < k; j++) {  // k rounds on the block-sums table (MSB-first, like partialEval)
                const size_t half = B.size() / 2;
                uint64_t s0 = 0, s1 = 0;
                for (size_t x = 0; x < half; x++) { s0 = h_add(s0, B[x]); s1 = h_add(s1, B[x + half]); }
                uint64_t ch;
                CHK(next_challenge(s0, h_sub(s1, s0), &ch));
                for (size_t x = 0; x < half; x++) B[x] = h_add(B[x], h_mul(ch, h_sub(B[x + half], B[x])));
                B.resize(half);
                std::vector<uint64_t> W2(W.size() * 2);
                const uint64_t one_minus = h_sub(1, ch);
                for (size_t x = 0; x < W.size(); x++) { W2[2 * x] = h_mul(W[x], one_minus); W2[2 * x + 1] = h_mul(W[x], ch); }
                W.swap(W2);
            }
            const size_t m = len >> k;
            const unsigned lm = log2_floor(m);
            const unsigned k_next = m <= HOST_TAIL_MAX ? 0 : (lm - 8 < RADIX_MAX_K ? lm - 8 : RADIX_MAX_K);
            B.assign(k_next ? (size_t)1 << k_next : 0, 0);
            ZK_LOCAL(ops.fold(ops.user, k, W.data(), k_next, k_next ? B.data() : nullptr));
            len = m;
            if (!k_next) break;
            CHK(sum_over_ranks(B));
            k = k_next;
        }
    }
    // the remaining table: len local entries per rank, global index j*G + g
    std::vector<uint64_t> mine(len);
    ZK_LOCAL(ops.read_tail(ops.user, len, mine.data()));
#undef ZK_LOCAL
    if (world == 1) {
        tail.swap(mine);
    } else {
        CHK(exchange(mine));
        tail.resize(world * len);
        for (size_t r = 0; r < world; r++)
            for (size_t j = 0; j < len; j++) {
                if (gather[r * len + j] >= P) return ZIGZ_ERR_NOT_CANONICAL;  // (the same verdict on every rank)
                tail[j * world + r] = gather[r * len + j];
            }
    }
    while (tail.size() > 1) {  // last rounds on the <= 1024 * G entry table, identical on every rank
        const size_t half = tail.size() / 2;
        uint64_t s0 = 0, s1 = 0;
        for (size_t x = 0; x < half; x++) { s0 = h_add(s0, tail[x]); s1 = h_add(s1, tail[x + half]); }
        uint64_t ch;
        CHK(next_challenge(s0, h_sub(s1, s0), &ch));
        for (size_t x = 0; x < half; x++) tail[x] = h_add(tail[x], h_mul(ch, h_sub(tail[x + half], tail[x])));
        tail.resize(half);
    }
    if (round != nv) return ZIGZ_ERR_PROTOCOL_ERROR;  // sumcheck_prover.zig:80-82
    *final_eval = tail[0];
    return ZIGZ_OK;
    ZIGZ_NOTHROW_END(ctx)
}

// the three passes on the GPU, buffers from the context's workspaces (sized by the first, largest stage; two output
// regions used alternately; no allocation inside the loop)
struct GpuRadix {
    zigz_ctx *ctx;
    const uint32_t *cur;
    size_t len, m0;
    unsigned long long *d_part;
    uint32_t *d_outs;
    void *wbuf;
    unsigned stage;
    zigz_rccl_comm *rccl;  // != nullptr: block sums are all-reduced over the ranks in HBM, on the context's stream
};
// The sums a data pass has just produced in d_sums[0, n) -> `out`.  Sharded over RCCL they are first all-reduced in place, on
// the context's stream -- together with word n, the number of ranks whose local pass failed (st != OK: this rank adds 1) -- so
// the collective is issued on EVERY rank whatever happened locally, and all ranks learn of a failure in the same collective:
// the failing rank returns its own error, the others ZIGZ_ERR_COMM, and radix_run then skips the remaining passes on all of
// them alike (their collectives stay matched).  The wait behind the collective has the communicator's deadline: a peer that
// never enters it costs an abort and ZIGZ_ERR_COMM here, not a hang (RCCL has no timeout of its own).
zigz_status sums_out(GpuRadix *g, unsigned long long *d_sums, size_t n, zigz_status st, uint64_t *out) {
    zigz_ctx *ctx = g->ctx;
    if (!g->rccl) {
        CHK(st);
        return read_u64(ctx, d_sums, n, out);
    }
    if (st != ZIGZ_OK) (void)hipMemsetAsync(d_sums + n, 1, 1, ctx->stream);  // (the word was zeroed with the sums: now 1)
    const int rc = zigz_rccl_allreduce_u64_dev(g->rccl, (uint64_t *)d_sums, n + 1, ctx->stream);
    if (st != ZIGZ_OK) {
        (void)zigz_rccl_stream_wait(g->rccl, ctx->stream);
        return st;
    }
    if (rc != 0) {
        set_err(ctx, "sharded sumcheck: the RCCL all-reduce could not be enqueued (%d)", rc);
        return ZIGZ_ERR_COMM;
    }
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_pin, d_sums, (n + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (const int w = zigz_rccl_stream_wait(g->rccl, ctx->stream)) {
        set_err(ctx, "sharded sumcheck: the RCCL all-reduce did not complete (%d): communicator aborted", w);
        return ZIGZ_ERR_COMM;
    }
    memcpy(out, ctx->h_pin, n * 8);
    if (ctx->h_pin[n] != 0) {
        set_err(ctx, "sharded sumcheck: another rank reported an error");
        return ZIGZ_ERR_COMM;
    }
    return ZIGZ_OK;
}
zigz_status gpu_block_sums(void *user, unsigned k, uint64_t *sums) {
    GpuRadix *g = (GpuRadix *)user;
    zigz_ctx *ctx = g->ctx;
    const size_t nb = (size_t)1 << k;
    auto local = [&]() -> zigz_status {
        HIPCHK(ctx, hipMemsetAsync(ctx->d_sums, 0, (nb + 1) * 8, ctx->stream));
        launch_block_sums(g->cur, g->len, g->len, log2_floor(g->len >> k), 1, ctx->d_sums, SumsLayout{0, 1, 0, 1}, ctx->stream);
        HIPCHK(ctx, hipGetLastError());
        return ZIGZ_OK;
    };
    return sums_out(g, ctx->d_sums, nb, local(), sums);
}
zigz_status gpu_fold(void *user, unsigned k, const uint64_t *weights, unsigned k_next, uint64_t *next_sums) {
    GpuRadix *g = (GpuRadix *)user;
    zigz_ctx *ctx = g->ctx;
    const size_t nb = (size_t)1 << k, m = g->len >> k;
    uint32_t *d_out = g->d_outs + (g->stage & 1) * g->m0;
    // the next stage's block sums alternate between two regions of d_sums ((1 << RADIX_MAX_K) + 1 words each: sums + the
    // failure word of sums_out), so a stage's memset never touches words a read-back of the stage before may still copy
    unsigned long long *d_B2 = ctx->d_sums + ((g->stage + 1) & 1 ? 2048 : 0);
    auto local = [&]() -> zigz_status {
        uint32_t *wst = (uint32_t *)(ctx->h_pin + PIN_WORDS / 2);
        for (size_t b = 0; b < nb; b++) wst[b] = host_to_mont(weights[b]);
        HIPCHK(ctx, hipMemcpyAsync(g->wbuf, wst, nb * 4, hipMemcpyHostToDevice, ctx->stream));
        const size_t groups = radix_fold_groups(nb);
        launch_radix_fold(g->cur, 0, m, nb, (const uint32_t *)g->wbuf, 0, g->d_part, 0, 1, ctx->stream);
        HIPCHK(ctx, hipGetLastError());
        if (k_next) {
            HIPCHK(ctx, hipMemsetAsync(d_B2, 0, (((size_t)1 << k_next) + 1) * 8, ctx->stream));
            launch_radix_finalize(g->d_part, 0, groups, d_out, 0, m, log2_floor(m) - k_next, d_B2, 1, ctx->stream);
        } else {
            launch_radix_finalize(g->d_part, 0, groups, d_out, 0, m, 0, nullptr, 1, ctx->stream);
        }
        HIPCHK(ctx, hipGetLastError());
        return ZIGZ_OK;
    };
    const zigz_status st = local();
    if (k_next) CHK(sums_out(g, d_B2, (size_t)1 << k_next, st, next_sums));
    else CHK(st);
    g->cur = d_out;
    g->len = m;
    g->stage++;
    return ZIGZ_OK;
}
zigz_status gpu_read_tail(void *user, size_t m, uint64_t *out) {
    GpuRadix *g = (GpuRadix *)user;
    zigz_ctx *ctx = g->ctx;
    if (m > PIN_WORDS) return ZIGZ_ERR_INVALID_ARGUMENT;
    uint32_t *h32 = (uint32_t *)ctx->h_pin;
    HIPCHK(ctx, hipMemcpyAsync(h32, g->cur, m * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < m; i++) out[i] = h32[i];
    return ZIGZ_OK;
}
}  // namespace

static zigz_status sumcheck_radix_sharded(zigz_ctx *ctx, const uint32_t *d_in, size_t n, const ShardComm *comm,
                                          const uint64_t *fixed, uint64_t *rounds, uint64_t *point, uint64_t *final_eval,
                                          zigz_rccl_comm *rccl = nullptr) {
    GpuRadix g{ctx, d_in, n, 0, nullptr, nullptr, nullptr, 0, rccl};
    if (n > HOST_TAIL_MAX) {
        const unsigned lv = log2_floor(n);
        const unsigned k = lv - 8 < RADIX_MAX_K ? lv - 8 : RADIX_MAX_K;
        g.m0 = n >> k;
        const size_t g0 = radix_fold_groups((size_t)1 << k);
        void *ws;
        CHK(ws_get(ctx, WS_SCRATCH, g0 * g.m0 * 8 + 2 * g.m0 * 4 + 256, &ws));
        g.d_part = (unsigned long long *)ws;
        g.d_outs = (uint32_t *)(g.d_part + g0 * g.m0);
        CHK(ws_get(ctx, WS_MISC, ((size_t)1 << RADIX_MAX_K) * 4 + 64, &g.wbuf));
    }
    const RadixOps ops{&g, gpu_block_sums, gpu_fold, gpu_read_tail};
    return radix_run(ctx, ops, n, comm, fixed, rounds, point, final_eval);
}

static zigz_status sumcheck_radix(zigz_ctx *ctx, const uint32_t *d_in, size_t n, const uint64_t *fixed, uint64_t *rounds,
                                  uint64_t *point, uint64_t *final_eval) {
    return sumcheck_radix_sharded(ctx, d_in, n, nullptr, fixed, rounds, point, final_eval);
}

// SumcheckProver.prove (src/proofs/sumcheck_prover.zig:26-91) of ONE table sharded by rows over `world` GPUs
extern "C" zigz_status zigz_dev_sumcheck_prove_sharded(zigz_ctx *ctx, const uint32_t *d_local, size_t n_local, int rank,
                                                       int world, zigz_allgather_fn allgather, void *user,
                                                       uint64_t *rounds, uint64_t *point, uint64_t *final_eval) {
    ZIGZ_ENTER(ctx);
    if (!ctx || !d_local || !rounds || !point || !final_eval) return ZIGZ_ERR_INVALID_ARGUMENT;
    CHK(mle_check(n_local));
    if (world < 1 || rank < 0 || rank >= world || !is_pow2((size_t)world) || (world > 1 && !allgather)) return ZIGZ_ERR_INVALID_ARGUMENT;
    if (n_local * (size_t)world == 1) return ZIGZ_ERR_NO_VARIABLES;
    if (!aligned16(d_local)) return ZIGZ_ERR_INVALID_ARGUMENT;
    const ShardComm comm{rank, world, allgather, user, false};
    return sumcheck_radix_sharded(ctx, d_local, n_local, &comm, nullptr, rounds, point, final_eval);
}

// The same proof with RCCL as the transport, natively: the partial block sums of every radix stage (k <= 10 rounds' worth of
// round-polynomial sums) are all-reduced IN HBM on the context's stream before they are read back for the transcript -- the
// north-star's RCCL all-reduce of the round sums, once per stage instead of once per round -- and the last <= 1024 * world
// table entries are all-gathered through the communicator's staging buffers.
extern "C" zigz_status zigz_dev_sumcheck_prove_rccl(zigz_ctx *ctx, const uint32_t *d_local, size_t n_local, zigz_rccl_comm *rccl,
                                                    uint64_t *rounds, uint64_t *point, uint64_t *final_eval) {
    ZIGZ_ENTER(ctx);
    if (!ctx || !d_local || !rccl || !rounds || !point || !final_eval) return ZIGZ_ERR_INVALID_ARGUMENT;
    CHK(mle_check(n_local));
    const int world = zigz_rccl_comm_world(rccl), rank = zigz_rccl_comm_rank(rccl);
    if (world < 1 || !is_pow2((size_t)world)) return ZIGZ_ERR_INVALID_ARGUMENT;
    if (n_local * (size_t)world == 1) return ZIGZ_ERR_NO_VARIABLES;
    if (!aligned16(d_local)) return ZIGZ_ERR_INVALID_ARGUMENT;
    const ShardComm comm{rank, world, zigz_rccl_allgather, rccl, true};
    // (one rank: the all-reduce is RCCL's identity, and the path is the one several ranks take)
    return sumcheck_radix_sharded(ctx, d_local, n_local, &comm, nullptr, rounds, point, final_eval, rccl);
}

// The orchestration alone, over caller-supplied data passes (multi-process tests on CPU drive exactly the code path of
// zigz_dev_sumcheck_prove_sharded with stand-in passes; a host with its own kernels could do the same)
extern "C" zigz_status zigz_sumcheck_radix_run(const zigz_radix_ops *ops, size_t n_local, int rank, int world,
                                               zigz_allgather_fn allgather, void *comm_user, const uint64_t *fixed_challenges,
                                               uint64_t *rounds, uint64_t *point, uint64_t *final_eval) {
    if (!ops || !ops->block_sums || !ops->fold || !ops->read_tail || !rounds || !point || !final_eval) return ZIGZ_ERR_INVALID_ARGUMENT;
    CHK(mle_check(n_local));
    if (world < 1 || rank < 0 || rank >= world || !is_pow2((size_t)world) || (world > 1 && !allgather)) return ZIGZ_ERR_INVALID_ARGUMENT;
    if (n_local * (size_t)world == 1) return ZIGZ_ERR_NO_VARIABLES;
    const ShardComm comm{rank, world, allgather, comm_user, false};
    const RadixOps r{ops->user, ops->block_sums, ops->fold, ops->read_tail};
    return radix_run(nullptr, r, n_local, &comm, fixed_challenges, rounds, point, final_eval);
}

// ... for data passes that return the sums over ALL ranks already (reduced inside the pass, as the RCCL passes above do)
extern "C" zigz_status zigz_sumcheck_radix_run_reduced(const zigz_radix_ops *ops, size_t n_local, int rank, int world,
                                                       zigz_allgather_fn allgather, void *comm_user, const uint64_t *fixed_challenges,
                                                       uint64_t *rounds, uint64_t *point, uint64_t *final_eval) {
    if (!ops || !ops->block_sums || !ops->fold || !ops->read_tail || !rounds || !point || !final_eval) return ZIGZ_ERR_INVALID_ARGUMENT;
    CHK(mle_check(n_local));
    if (world < 1 || rank < 0 || rank >= world || !is_pow2((size_t)world) || (world > 1 && !allgather)) return ZIGZ_ERR_INVALID_ARGUMENT;
    if (n_local * (size_t)world == 1) return ZIGZ_ERR_NO_VARIABLES;
    const ShardComm comm{rank, world, allgather, comm_user, true};
    const RadixOps r{ops->user, ops->block_sums, ops->fold, ops->read_tail};
    return radix_run(nullptr, r, n_local, &comm, fixed_challenges, rounds, point, final_eval);
}

extern "C" zigz_status zigz_dev_sumcheck_prove(zigz_ctx *ctx, const uint32_t *d_in, size_t n, uint32_t *d_scratch,
                                               const uint64_t *fixed_challenges, uint64_t *rounds, uint64_t *point,
                                               uint64_t *final_eval) {
    ZIGZ_ENTER(ctx);
    if (!ctx || !d_in || !rounds || !point || !final_eval) return ZIGZ_ERR_INVALID_ARGUMENT;
    CHK(mle_check(n));
    if (n == 1) return ZIGZ_ERR_NO_VARIABLES;
    return sumcheck_core(ctx, d_in, n, d_scratch, fixed_challenges, rounds, point, final_eval);
}

// ------------------------------------------------------------------ host-buffer seams: Multilinear
static zigz_status stage_in(zigz_ctx *ctx, const uint64_t *in, size_t n, uint32_t **d_out) {
    void *d32;
    CHK(ws_get(ctx, WS_IN32, n * 4, &d32));
    CHK(upload_u64(ctx, in, n, (uint32_t *)d32, false));
    *d_out = (uint32_t *)d32;
    return ZIGZ_OK;
}

extern "C" zigz_status zigz_mle_bind(zigz_ctx *ctx, const uint64_t *in, size_t n, uint64_t r, uint64_t *out) {
    ZIGZ_ENTER(ctx);
    if (!ctx) return ZIGZ_ERR_INVALID_ARGUMENT;
    CHK(mle_check(n));
    if (n == 1) return ZIGZ_ERR_NO_VARIABLES_TO_FIX;
    if (!in || !out) return ZIGZ_ERR_INVALID_ARGUMENT;
    if (r >= P) return ZIGZ_ERR_NOT_CANONICAL;
    uint32_t *d_in;
    CHK(stage_in(ctx, in, n, &d_in));
    void *d_o;
    CHK(ws_get(ctx, WS_OUT32, (n / 2) * 4, &d_o));
    CHK(zigz_dev_mle_bind(ctx, d_in, n, r, (uint32_t *)d_o));
    return download_u64(ctx, (uint32_t *)d_o, n / 2, out);
}

extern "C" zigz_status zigz_mle_round_poly(zigz_ctx *ctx, const uint64_t *in, size_t n, uint64_t out[2]) {
    ZIGZ_ENTER(ctx);
    if (!ctx) return ZIGZ_ERR_INVALID_ARGUMENT;
    CHK(mle_check(n));
    if (n == 1) return ZIGZ_ERR_NO_VARIABLES;
    if (!in || !out) return ZIGZ_ERR_INVALID_ARGUMENT;
    uint32_t *d_in;
    CHK(stage_in(ctx, in, n, &d_in));
    uint64_t s[2];
    CHK(dev_half_sums(ctx, d_in, n, s));
    uint64_t s0 = s[0] % P, s1 = s[1] % P;
    out[0] = s0;
    out[1] = s1 >= s0 ? s1 - s0 : s1 + P - s0;
    return ZIGZ_OK;
}

extern "C" zigz_status zigz_mle_sum(zigz_ctx *ctx, const uint64_t *in, size_t n, uint64_t *out) {
    ZIGZ_ENTER(ctx);
    if (!ctx) return ZIGZ_ERR_INVALID_ARGUMENT;
    CHK(mle_check(n));
    if (!in || !out) return ZIGZ_ERR_INVALID_ARGUMENT;
    uint32_t *d_in;
    CHK(stage_in(ctx, in, n, &d_in));
    uint64_t s[2];
    CHK(dev_half_sums(ctx, d_in, n, s));
    *out = (s[0] % P + s[1] % P) % P;
    return ZIGZ_OK;
}

extern "C" zigz_status zigz_mle_eval(zigz_ctx *ctx, const uint64_t *in, size_t n, const uint64_t *point,
                                     size_t point_len, uint64_t *out) {
    ZIGZ_ENTER(ctx);
    if (!ctx) return ZIGZ_ERR_INVALID_ARGUMENT;
    CHK(mle_check(n));
    if (point_len != log2_floor(n)) return ZIGZ_ERR_WRONG_NUMBER_OF_VARIABLES;
    if (!in || !out) return ZIGZ_ERR_INVALID_ARGUMENT;
    uint32_t *d_in;
    CHK(stage_in(ctx, in, n, &d_in));
    return zigz_dev_mle_eval(ctx, d_in, n, point, point_len, out);
}

extern "C" zigz_status zigz_sumcheck_prove(zigz_ctx *ctx, const uint64_t *in, size_t n, uint64_t *rounds,
                                           uint64_t *point, uint64_t *final_eval) {
    ZIGZ_ENTER(ctx);
    if (!ctx) return ZIGZ_ERR_INVALID_ARGUMENT;
    CHK(mle_check(n));
    if (n == 1) return ZIGZ_ERR_NO_VARIABLES;
    if (!in || !rounds || !point || !final_eval) return ZIGZ_ERR_INVALID_ARGUMENT;
    uint32_t *d_in;
    CHK(stage_in(ctx, in, n, &d_in));
    return sumcheck_core(ctx, d_in, n, nullptr, nullptr, rounds, point, final_eval);
}

extern "C" zigz_status zigz_sumcheck_prove_interactive(zigz_ctx *ctx, const uint64_t *in, size_t n,
                                                       const uint64_t *challenges, size_t n_challenges,
                                                       uint64_t *rounds, uint64_t *point, uint64_t *final_eval) {
    ZIGZ_ENTER(ctx);
    if (!ctx) return ZIGZ_ERR_INVALID_ARGUMENT;
    CHK(mle_check(n));
    if (n == 1) return ZIGZ_ERR_NO_VARIABLES;
    if (n_challenges != log2_floor(n)) return ZIGZ_ERR_WRONG_NUMBER_OF_CHALLENGES;
    if (!in || !challenges || !rounds || !point || !final_eval) return ZIGZ_ERR_INVALID_ARGUMENT;
    uint32_t *d_in;
    CHK(stage_in(ctx, in, n, &d_in));
    return sumcheck_core(ctx, d_in, n, nullptr, challenges, rounds, point, final_eval);
}

// ------------------------------------------------------------------ Merkle
// How much room the lists (and the digests stored in list order) of the structure-aware levels get: learnt from what the
// context's previous builds needed, not sized for the worst case -- a build that runs out says so and is repeated with more
// (zigz_commit_roots), which costs one extra build the first time a context meets a new kind of trace.
static void caps_for(zigz_ctx *ctx, size_t npad, unsigned rn, unsigned gn) {
    ListCaps &c = ctx->caps;
    if (c.npad == npad && c.rn == rn && c.gn == gn) return;
    c.npad = npad;
    c.rn = rn;
    c.gn = gn;
    c.g_slabs = false;
    c.g_drops = c.g_skip = 0;
    const LevelLists rw = runs_lists(npad, rn ? rn : 1), gw = cons_lists(npad);
    for (unsigned l = 0; l < RUN_MAX_LEVELS; l++) {
        c.r[l] = l <= rw.top ? (rw.cap[l] / 8 > 256 ? rw.cap[l] / 8 : 256) : 0;  // an eighth of "every node hashed"
        c.g[l] = l <= gw.top ? 256 : 0;                                           // 8192 distinct nodes per level
    }
}

// Builds all levels of `ncols` trees, asynchronously on the stream.
// ref != nullptr (a commit job): the digests of the list-built levels stay in list order (TreeRef: stores), only densely
//   built columns get node-addressed slabs (taken from WS_TREE); *ref describes where everything is and stays valid while
//   the context's WS_RUNMETA / WS_CONSMETA / WS_TREE workspaces are untouched -- until the job ends.
// ref == nullptr, or option "run_aware_materialize": every digest of every tree is written into node-addressed slabs
//   (d_slab if given: single trees that outlive the call; else WS_TREE), one per column.
static zigz_status build_trees(zigz_ctx *ctx, const uint32_t *d_vals, size_t val_stride, size_t n_values, size_t npad,
                               uint8_t *d_slab, size_t ncols, bool record = false, TreeRef *ref = nullptr) {
    const unsigned height = log2_floor(npad);
    const size_t stride = tree_nodes(npad);
    ctx->stats.small_domain_columns = 0;
    ctx->stats.run_aware_columns = 0;
    ctx->stats.run_aware_dense_nodes = 0;
    ctx->stats.cons_columns = 0;
    ctx->stats.cons_dense_nodes = 0;
    ctx->stats.cons_probe_distinct = 0;
    ctx->build_cons_hinted = 0;
    ctx->build_cons_levels_nodes = 0;
    ctx->build_cons_sd = 0;
    ctx->build_top_perms = 0;
    if (record) ctx->kev_n = 0;
    // the list counters of a commit job's build are read again by its openings (EvalSkip, the "group dropped" word): they are
    // the job's; any other build on the context counts in the auxiliary pair
    unsigned long long *const r_ctr = ref ? ctx->d_run_count : ctx->d_run_aux, *const g_ctr = ref ? ctx->d_cons_count : ctx->d_cons_aux;
    // timing mode: every launch (or bracketed group of launches) carries its own begin / end timestamps, by class
    KTime kt_store;
    auto stamp = [&](int cls, uint64_t perms) -> const KTime * {
        if (!record || ctx->kev_n >= KEV_MAX) return nullptr;
        kt_store = KTime{ctx->kev[2 * ctx->kev_n], ctx->kev[2 * ctx->kev_n + 1]};
        ctx->kev_class[ctx->kev_n] = (uint8_t)cls;
        ctx->kev_perms[ctx->kev_n] = perms;
        ctx->kev_n++;
        return &kt_store;
    };
    // Four kinds of columns:
    //   G  hinted as a group that repeats in the same places (the columns that are functions of the instruction at pc): the
    //      levels 0 .. v - 8 are content-addressed -- takes precedence over H and R.  Whether the group repeats enough to be
    //      worth it is decided ON THE DEVICE after the leaf level's table pass (more than a quarter of the leaves distinct:
    //      dropped); a dropped group's columns are built like H (its small-domain members) and D (the rest) by launches that
    //      read the same device flag, so nothing here waits for the device;
    //   H  hinted small-domain (values < 128 by construction): levels 0 and 1 from two constant tables, checked per wave
    //      and hashed where the bound does not hold;
    //   R  hinted run-aware (piecewise constant): the levels 0 .. v - 8 from lists of the nodes that are not a copy of their
    //      left neighbour -- decided from the values, so the hint cannot make a tree wrong;
    //   D  the rest: hashed densely.
    // The top kernel (256 nodes per column -> root) takes all columns together.
    ColMap H{}, R{}, D{}, G{}, GS{};
    const bool big = npad >= RUN_MIN_LEAVES && npad <= RUN_MAX_LEAVES && ncols <= 64;
    const bool sd_ok = ctx->small_domain_mask && npad >= 1024 && ncols <= 64 && val_stride % 2 == 0 && ((uintptr_t)d_vals & 7) == 0;
    const bool run_ok = ctx->run_aware_mask && big;
    // (a content-addressing key packs two child list slots into RUN_NODE_BITS bits each, and a slot is sub-list * capacity +
    // position: at npad == 2^26 a nearly full last sub-list reaches 2^26 + 2047 -- the group path stops one size short of that)
    bool cons_ok = ctx->cons_group_mask && big && npad < RUN_MAX_LEAVES;
    unsigned gn_hinted = 0;  // (what the context learnt is filed under the hints, not under whether this build tries the group)
    for (size_t c = 0; cons_ok && c < ncols; c++) gn_hinted += (unsigned)((ctx->cons_group_mask >> c) & 1);
    if (run_ok || cons_ok) {
        unsigned rn_hinted = 0;
        for (size_t c = 0; run_ok && c < ncols; c++)
            rn_hinted += (unsigned)(((ctx->run_aware_mask >> c) & 1) && !(cons_ok && ((ctx->cons_group_mask >> c) & 1)) &&
                                    !(sd_ok && ((ctx->small_domain_mask >> c) & 1)));
        caps_for(ctx, npad, rn_hinted, gn_hinted);
    }
    // A context whose last two jobs dropped the group (its traces do not loop) stops trying for a while: the group's columns
    // are then H / D from the start -- no table passes that find nothing, and the tuned dense kernels instead of the list
    // kernel's dense branch -- and every 16th job looks again.
    if (cons_ok && ref && !ctx->cons_always && ctx->caps.npad == npad && ctx->caps.g_skip) {
        ctx->caps.g_skip--;
        cons_ok = false;
    }
    auto kind = [&](size_t c) -> int {  // 0 D, 1 H, 2 R, 3 G
        if (cons_ok && ((ctx->cons_group_mask >> c) & 1)) return 3;
        if (sd_ok && ((ctx->small_domain_mask >> c) & 1)) return 1;
        if (run_ok && ((ctx->run_aware_mask >> c) & 1)) return 2;
        return 0;
    };
    if (sd_ok || run_ok || cons_ok)
        for (size_t c = 0; c < ncols; c++) {
            const int kd = kind(c);
            ColMap &m = kd == 3 ? G : kd == 1 ? H : kd == 2 ? R : D;
            m.c[m.n++] = (uint8_t)c;
            if (kd == 3 && sd_ok && ((ctx->small_domain_mask >> c) & 1)) GS.c[GS.n++] = (uint8_t)c;
        }
    const bool lists = R.n || G.n;
    const bool whole = ref == nullptr || ctx->run_aware_materialize;  // every digest into node-addressed slabs
    const bool virt = !whole;  // copies / non-representatives / table leaves never written
    // ---- where the digests go
    TreeRef t{};
    t.npad = npad;
    for (int c = 0; c < 64; c++) {
        t.slab_of_col[c] = -1;
        t.y_of_col[c] = -1;
        t.g_j_of_col[c] = -1;
    }
    size_t nslab = 0;
    if (whole || !lists) {
        nslab = ncols;
        for (size_t c = 0; c < ncols && c < 64; c++) t.slab_of_col[c] = (signed char)c;
    } else {
        for (size_t c = 0; c < ncols; c++) {
            const int kd = kind(c);
            if (kd == 0 || kd == 1 || (kd == 3 && ctx->caps.g_slabs)) t.slab_of_col[c] = (signed char)nslab++;
        }
    }
    if (d_slab) t.slab = d_slab;  // (a single tree: ncols slabs of the caller's)
    else if (nslab) {
        void *w;
        CHK(ws_get(ctx, WS_TREE, nslab * stride * 32, &w));
        t.slab = (uint8_t *)w;
    }
    auto slab_map = [&](const ColMap &m) {  // for the dense kernels: entry k of m -> its slab
        ColMap o{};
        o.n = m.n;
        for (unsigned k = 0; k < m.n; k++) o.c[k] = (uint8_t)t.slab_of_col[m.c[k]];
        return o;
    };
    if (H.n == 0 && R.n == 0 && G.n == 0) {
        launch_keccak_leaves(d_vals, val_stride, n_values, npad, t.slab, stride, ncols, ctx->stream, stamp(0, (uint64_t)ncols * npad));
    } else if (D.n) {
        const ColMap ds = slab_map(D);
        launch_keccak_leaves(d_vals, val_stride, n_values, npad, t.slab, stride, ncols, ctx->stream, stamp(0, (uint64_t)D.n * npad), &D, &ds);
    }
    void *sd_todo = nullptr;
    if (H.n || GS.n) {
        if (!ctx->d_sd_tables) {
            HIPCHK(ctx, hipMalloc((void **)&ctx->d_sd_tables, SD_TABLE_BYTES));
            HIPCHK(ctx, hipMalloc((void **)&ctx->d_sd_fallbacks, 64));
            launch_sd_tables(ctx->d_sd_tables, ctx->stream);
        }
        CHK(ws_get(ctx, WS_DEDUP, (sd_todo_words(npad, H.n) + sd_todo_words(npad, GS.n)) * 4, &sd_todo));
    }
    // every counter this build's kernels add to, zeroed by ONE launch (three memsets are three commands in the stream)
    if (H.n || GS.n || R.n || G.n)
        launch_zero_counters((H.n || GS.n) ? ctx->d_sd_fallbacks : nullptr, R.n ? r_ctr : nullptr,
                             G.n ? g_ctr : nullptr, ctx->stream);
    if (H.n) {
        // in a commit job the leaf digests of these columns are left out (virtual): only an opening reads one, and it
        // hashes that value itself
        const ColMap hs = slab_map(H);
        launch_keccak_small_l01(d_vals, val_stride, n_values, npad, t.slab, stride, H, ctx->d_sd_tables, ctx->d_sd_fallbacks,
                                (uint32_t *)sd_todo, ctx->stream, stamp(3, 0), !virt, nullptr, &hs);
        if (virt)
            for (unsigned k = 0; k < H.n; k++) t.virtual_leaves |= 1ull << H.c[k];
        ctx->stats.small_domain_columns = H.n;
    }
    MerkleBuild b{};
    unsigned top = 0;
    if (lists) {
        top = run_top_level(npad);
        t.lists = 1;
        t.top = top;
        b.vals = d_vals;
        b.val_stride = val_stride;
        b.n_values = n_values;
        b.npad = npad;
        b.rcols = R;
        b.gcols = G;
        b.gcols_sd = GS;
        uint64_t level_nodes = 0;
        for (unsigned l = 0; l <= top; l++) level_nodes += npad >> l;
        // what outlives the build (read by the openings): a commit job keeps it in workspaces of its own, which nothing but
        // the next commit job touches; otherwise it is scratch like the rest
        const bool keep = ref != nullptr;
        const size_t upper_bytes = ncols * 512 * 32;
        if (R.n) {
            // (only a commit job can repeat a build that ran out of room: anything else gets the worst case)
            t.r_lists = runs_lists(npad, R.n, ref ? ctx->caps.r : nullptr);
            unsigned long long uoff[RUN_MAX_LEVELS] = {0};
            const size_t units = runs_units(npad, R.n, uoff);
            for (unsigned l = 0; l < RUN_MAX_LEVELS; l++) t.ubase_off[l] = uoff[l];
            const size_t list_bytes = (size_t)t.r_lists.entries * 4, stage_bytes = runs_stage_scratch_bytes(npad, R.n);
            const size_t meta_n = runs_meta_words(npad, R.n);
            // kept: bitmap | prev | woff | ubase | digests in list order
            const size_t kept = meta_n * 12 + units * 4 + 64 + (size_t)t.r_lists.entries * 32 + 64;
            void *w, *mw;
            CHK(ws_get(ctx, WS_RUNS, ((list_bytes + 63) & ~(size_t)63) + stage_bytes + 64 + (keep ? 0 : kept), &w));
            b.r_list = (uint32_t *)w;
            b.r_stage = (uint8_t *)w + ((list_bytes + 63) & ~(size_t)63);
            if (keep) CHK(ws_get(ctx, WS_RUNMETA, kept + upper_bytes, &mw));
            else mw = (uint8_t *)w + ((((list_bytes + 63) & ~(size_t)63) + stage_bytes + 64 + 63) & ~(size_t)63);
            uint8_t *q = (uint8_t *)mw;
            t.bitmap = (unsigned long long *)q; q += meta_n * 8;
            t.prev = (unsigned short *)q; q += meta_n * 2;
            t.woff = (unsigned short *)q; q += meta_n * 2;
            t.ubase = (uint32_t *)q; q += (units * 4 + 63) & ~(size_t)63;
            t.r_store = q; q += (size_t)t.r_lists.entries * 32;
            if (keep) t.upper = (uint8_t *)mw + kept;
            t.ncols = R.n;
            for (unsigned y = 0; y < R.n; y++) t.y_of_col[R.c[y]] = (signed char)y;
            b.r_ctr = r_ctr;
            ctx->stats.run_aware_columns = R.n;
            ctx->stats.run_aware_dense_nodes = (uint64_t)R.n * level_nodes;
        }
        if (G.n) {
            // table (generation-tagged: cleared only when the workspace is new or the generations run out) + list: scratch;
            // the representative slots and the digests in list order are kept while the trees are read through them
            t.g_lists = cons_lists(npad, ref ? ctx->caps.g : nullptr);
            const size_t key_bytes = 2 * npad * 8, idx_bytes = 2 * npad * 4, list_bytes = ((size_t)t.g_lists.entries * 4 + 63) & ~(size_t)63;
            const size_t kept = 2 * npad * 4 + (size_t)t.g_lists.entries * G.n * 32 + 64;
            const bool upper_here = keep && !R.n;
            void *w, *mw;
            CHK(ws_get(ctx, WS_CONS, key_bytes + idx_bytes + list_bytes + 64 + (keep ? 0 : kept), &w));
            b.g_keys = (unsigned long long *)w;
            b.g_idx = (uint32_t *)((uint8_t *)w + key_bytes);
            b.g_list = (uint32_t *)((uint8_t *)w + key_bytes + idx_bytes);
            if (keep) CHK(ws_get(ctx, WS_CONSMETA, kept + (upper_here ? upper_bytes : 0), &mw));
            else mw = (uint8_t *)w + ((key_bytes + idx_bytes + list_bytes + 64 + 63) & ~(size_t)63);
            b.g_rep = (uint32_t *)mw;
            t.g_rep = b.g_rep;
            t.g_store = (uint8_t *)mw + 2 * npad * 4;
            if (upper_here) t.upper = (uint8_t *)mw + kept;
            if (ctx->cons_table != w || ctx->cons_table_bytes != ctx->ws_bytes[WS_CONS] || ctx->cons_gen + RUN_MAX_LEVELS + 1 >= 4096) {
                HIPCHK(ctx, hipMemsetAsync(w, 0, key_bytes, ctx->stream));  // generation 0 = free
                ctx->cons_table = w;
                ctx->cons_table_bytes = ctx->ws_bytes[WS_CONS];
                ctx->cons_gen = 1;
            }
            b.g_gen = ctx->cons_gen;
            ctx->cons_gen += top + 1;
            b.g_ctr = g_ctr;
            b.g_has_slabs = whole || ctx->caps.g_slabs;
            t.g_ncols = G.n;
            t.g_dropped = g_ctr + 8;
            for (unsigned k = 0; k < G.n; k++) t.g_j_of_col[G.c[k]] = (signed char)k;
            if (virt)
                for (unsigned k = 0; k < GS.n; k++) t.g_sd_mask |= 1ull << GS.c[k];
            ctx->build_cons_hinted = G.n;
            ctx->build_cons_levels_nodes = level_nodes;
            ctx->build_cons_sd = GS.n;
        }
        if (!t.upper) {  // not a job: the top levels are scratch too
            void *u;
            CHK(ws_get(ctx, WS_OUT64, upper_bytes, &u));
            t.upper = (uint8_t *)u;
        }
        b.t = t;
        if (ctx->debug_skip != 2) {
            launch_runs_structure(b, ctx->stream, R.n ? stamp(4, 0) : nullptr);
            launch_cons_structure(b, ctx->stream, G.n ? stamp(4, 0) : nullptr);
        }
        if (GS.n && b.g_has_slabs) {  // only if the group was dropped: its small-domain members' levels 0 and 1 by table
            const ColMap gs = slab_map(GS);
            launch_keccak_small_l01(d_vals, val_stride, n_values, npad, t.slab, stride, GS, ctx->d_sd_tables, ctx->d_sd_fallbacks + 1,
                                    (uint32_t *)sd_todo + sd_todo_words(npad, H.n), ctx->stream, stamp(3, 0), !virt, g_ctr + 8,
                                    &gs);
        }
    }
    if (ref) *ref = t;
    unsigned first_top = 0;  // the level the top kernel starts from
    if (lists) {
        for (unsigned l = 0; l <= top; l++) {
            if (ctx->debug_skip != 1) launch_level_hash(b, l, ctx->stream, stamp(5, 0));
            if (l == top) break;
            ColMap m{};  // the densely built columns that already have level l: D, and H from level 1
            for (size_t c = 0; c < ncols; c++) {
                const int kd = kind(c);
                if (kd == 0 || (kd == 1 && l >= 1)) m.c[m.n++] = (uint8_t)c;
            }
            if (m.n) {
                const ColMap ms = slab_map(m);
                launch_keccak_level(t.slab, stride, tree_level_offset(npad, l), tree_level_offset(npad, l + 1), npad >> (l + 1), ncols,
                                    ctx->stream, stamp(keccak_level_is_wide(npad >> (l + 1), m.n) ? 1 : 2, (uint64_t)m.n * (npad >> (l + 1))), &ms);
            }
        }
        first_top = top;
    } else {
        for (unsigned l = 0; l < height; l++) {  // level l + 1 from level l, for the columns that do not have it yet
            const size_t n_out = npad >> (l + 1);
            first_top = l;
            if (n_out <= 256) break;  // all columns are complete here (H stops at level 1): the top kernel takes over
            ColMap m{};
            const ColMap *pm = nullptr;
            size_t nc = ncols;
            if (H.n && l < 1) {  // the table columns join at level 1
                for (size_t c = 0; c < ncols; c++)
                    if (kind(c) == 0) m.c[m.n++] = (uint8_t)c;
                if (m.n == 0) continue;
                pm = &m;
                nc = m.n;
            }
            launch_keccak_level(t.slab, stride, tree_level_offset(npad, l), tree_level_offset(npad, l + 1), n_out, ncols, ctx->stream,
                                stamp(keccak_level_is_wide(n_out, nc) ? 1 : 2, (uint64_t)nc * n_out), pm);
        }
    }
    if (height) {
        ctx->build_top_perms = (uint64_t)ncols * ((npad >> first_top) - 1);
        if (ctx->debug_skip != 1) launch_merkle_top(t, first_top, height, ncols, ctx->stream, stamp(6, ctx->build_top_perms));
    }
    if (lists && whole) launch_fill_virtual(b, ctx->stream);
    HIPCHK(ctx, hipGetLastError());
    return ZIGZ_OK;
}

// after the stream has passed the last recorded launch: per-class device time of the last recorded build
static zigz_status keccak_times_collect(zigz_ctx *ctx) {
    // 0 leaves, 1 wide levels, 2 small levels, 3 small-domain table lookups, 4 structure passes (run-aware stages + content-
    // addressing table passes: no hashing), 5 list-driven level hashing, 6 the top of the trees
    double us[7] = {0, 0, 0, 0, 0, 0, 0};
    uint64_t perms[7] = {0, 0, 0, 0, 0, 0, 0};
    ctx->log_n = 0;
    for (int i = 0; i < ctx->kev_n; i++) {
        double d = 0;
        HIPCHK(ctx, hipEventSynchronize(ctx->kev[2 * i + 1]));
        CHK(log_launch(ctx, ctx->kev_class[i], ctx->kev_perms[i], ctx->kev[2 * i], ctx->kev[2 * i + 1], ctx->kev[0], &d));
        us[ctx->kev_class[i]] += d;
        perms[ctx->kev_class[i]] += ctx->kev_perms[i];
    }
    ctx->stats.keccak_leaves_us = us[0];
    ctx->stats.keccak_leaves_perms = perms[0];
    ctx->stats.keccak_level_wide_us = us[1];
    ctx->stats.keccak_level_wide_perms = perms[1];
    ctx->stats.keccak_level_small_us = us[2];
    ctx->stats.keccak_level_small_perms = perms[2];
    ctx->stats.small_domain_us = us[3];
    ctx->stats.structure_us = us[4];
    ctx->stats.list_hash_us = us[5];
    ctx->stats.top_us = us[6];
    ctx->stats.top_perms = perms[6];
    ctx->stats.run_aware_us = us[4] + us[5];
    ctx->kev_n = 0;
    return ZIGZ_OK;
}

struct zigz_merkle {
    uint32_t *d_vals;  // stored values (SimpleMerkleTree.values, merkle_tree.zig:291)
    uint8_t *d_tree;
    size_t n_values, npad;
    unsigned height;
};

extern "C" void zigz_merkle_destroy(zigz_ctx *ctx, zigz_merkle *t) {
    ZIGZ_ENTER(ctx);
    if (!t) return;
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
    if (t->d_vals) (void)hipFree(t->d_vals);
    if (t->d_tree) (void)hipFree(t->d_tree);
    delete t;
}

extern "C" zigz_status zigz_merkle_commit(zigz_ctx *ctx, const uint64_t *values, size_t n, uint8_t root[32],
                                          size_t *height, zigz_merkle **out) {
    ZIGZ_ENTER(ctx);
    if (!ctx) return ZIGZ_ERR_INVALID_ARGUMENT;
    if (n == 0) return ZIGZ_ERR_EMPTY_VALUES;                      // merkle_tree.zig:284
    if (n > ((size_t)1 << 40)) return ZIGZ_ERR_TOO_MANY_VALUES;    // merkle_tree.zig:287 (device-size cap)
    if (!values || !root) return ZIGZ_ERR_INVALID_ARGUMENT;
    const size_t npad = ceil_pow2(n);
    zigz_merkle *t = new (std::nothrow) zigz_merkle();
    if (!t) return ZIGZ_ERR_OUT_OF_MEMORY;
    memset(t, 0, sizeof(*t));
    t->n_values = n;
    t->npad = npad;
    t->height = log2_floor(npad);
    zigz_status st = ZIGZ_OK;
    auto body = [&]() -> zigz_status {
        HIPCHK(ctx, hipMalloc((void **)&t->d_vals, n * 4));
        HIPCHK(ctx, hipMalloc((void **)&t->d_tree, tree_nodes(npad) * 32));
        CHK(upload_u64(ctx, values, n, t->d_vals, false));
        CHK(build_trees(ctx, t->d_vals, n, n, npad, t->d_tree, 1));
        void *d_root;  // the root leaves the device through the gather kernel: tree form -> canonical SHA3 bytes
        CHK(ws_get(ctx, WS_MISC, 64, &d_root));
        launch_gather_nodes(t->d_tree, tree_nodes(npad), tree_level_offset(npad, t->height), (uint8_t *)d_root, 1, ctx->stream);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_pin, d_root, 32, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        memcpy(root, ctx->h_pin, 32);
        return ZIGZ_OK;
    };
    st = body();
    if (st != ZIGZ_OK) {
        zigz_merkle_destroy(ctx, t);
        return st;
    }
    if (height) *height = t->height;
    if (out) *out = t;
    else zigz_merkle_destroy(ctx, t);
    return ZIGZ_OK;
}

static zigz_status open_paths(zigz_ctx *ctx, const TreeRef &tree, unsigned height, const uint32_t *d_vals,
                              size_t val_stride, const uint64_t *h_idx, size_t ncols, uint8_t *siblings, uint8_t *dirs,
                              uint64_t *leaves) {
    ZIGZ_NOTHROW_BEGIN
    // device scratch layout: idx[ncols] u64 | sib[ncols*h*32] | leaf[ncols] u32 | dirs[ncols*h]
    const size_t sib_b = ncols * height * 32, idx_b = ncols * 8, leaf_b = ncols * 4, dir_b = ncols * height;
    void *w;
    CHK(ws_get(ctx, WS_OUT32, idx_b + sib_b + leaf_b + dir_b + 64, &w));
    uint8_t *base = (uint8_t *)w;
    uint64_t *d_idx = (uint64_t *)base;
    uint8_t *d_sib = base + idx_b;
    uint32_t *d_leaf = (uint32_t *)(d_sib + sib_b);
    uint8_t *d_dirs = (uint8_t *)(d_leaf + ncols);
    HIPCHK(ctx, hipMemcpyAsync(d_idx, h_idx, idx_b, hipMemcpyHostToDevice, ctx->stream));
    // (n_values = npad: leaf digests are virtual only in commit jobs, whose columns have exactly npad values)
    launch_paths(tree, tree.npad, height, d_vals, val_stride, d_idx, d_sib, d_dirs, d_leaf, ncols, ctx->stream);
    HIPCHK(ctx, hipGetLastError());
    std::vector<uint32_t> hl(ncols);
    if (sib_b) HIPCHK(ctx, hipMemcpyAsync(siblings, d_sib, sib_b, hipMemcpyDeviceToHost, ctx->stream));
    if (dir_b) HIPCHK(ctx, hipMemcpyAsync(dirs, d_dirs, dir_b, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(hl.data(), d_leaf, leaf_b, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t c = 0; c < ncols; c++) leaves[c] = hl[c];
    return ZIGZ_OK;
    ZIGZ_NOTHROW_END(ctx)
}

extern "C" zigz_status zigz_merkle_open(zigz_ctx *ctx, const zigz_merkle *t, size_t index, uint8_t *siblings,
                                        uint8_t *dirs, uint64_t *leaf_value) {
    ZIGZ_ENTER(ctx);
    if (!ctx || !t || !leaf_value) return ZIGZ_ERR_INVALID_ARGUMENT;
    if (index >= t->n_values) return ZIGZ_ERR_INDEX_OUT_OF_BOUNDS;  // merkle_tree.zig:325 (values.len)
    if (t->height && (!siblings || !dirs)) return ZIGZ_ERR_INVALID_ARGUMENT;
    uint64_t idx = index;
    return open_paths(ctx, slab_tree_ref(t->d_tree, t->npad), t->height, t->d_vals, t->n_values, &idx, 1, siblings, dirs, leaf_value);
}

extern "C" zigz_status zigz_commit_open(zigz_ctx *ctx, const uint64_t *evals, size_t n, const zigz_merkle *tree,
                                        const uint64_t *point, size_t point_len, uint64_t *value, uint64_t *index,
                                        uint8_t *siblings, uint8_t *dirs, uint64_t *leaf_value) {
    ZIGZ_ENTER(ctx);
    if (!ctx || !tree || !value || !index || !leaf_value) return ZIGZ_ERR_INVALID_ARGUMENT;
    CHK(mle_check(n));
    if (point_len != log2_floor(n)) return ZIGZ_ERR_POINT_DIMENSION_MISMATCH;  // polynomial_commit.zig:92-94
    const uint32_t *d_ev = tree->d_vals;
    if (evals) {
        uint32_t *d;
        CHK(stage_in(ctx, evals, n, &d));
        d_ev = d;
    } else if (tree->n_values != n) {
        return ZIGZ_ERR_INVALID_ARGUMENT;
    }
    CHK(zigz_dev_mle_eval(ctx, d_ev, n, point, point_len, value));  // polynomial_commit.zig:97
    size_t idx = point_len == 0 ? 0 : (size_t)(point[0] % ((uint64_t)1 << point_len));  // pointToIndex, :178-183
    *index = idx;
    return zigz_merkle_open(ctx, tree, idx, siblings, dirs, leaf_value);  // :105
}

// ------------------------------------------------------------------ batched commit job (generateCommitments)
struct zigz_commit_job {
    zigz_ctx *ctx;
    size_t ncols, nv, N, col_stride;
    const uint32_t *d_cols;
    int state;  // 0 begun, 1 roots read, 2 opened
    hipEvent_t built;
    uint64_t const_cols; // hinted columns the structure pass found constant (summary word 7): not read by the eval
    uint64_t roots_seq;  // the DoneFlag sequence number of the (last) build's summary launch
    TreeRef tree;  // where the digests are (the context's WS_TREE / WS_RUNMETA / WS_CONSMETA workspaces)
    bool whole;    // built with every digest in node-addressed slabs (option run_aware_materialize)
    // the hints the job was begun with (a repeated build -- zigz_commit_roots, when a list ran out of room -- uses the same) and
    // what its build asked for (turned into stats when the counters have arrived; other calls may run in between)
    uint64_t m_small, m_run, m_cons;
    bool m_whole;
    uint64_t run_cols, run_dense, sd_cols, cons_hinted, cons_levels_nodes, cons_sd, perms0;
    // a batched job (zigz_commit_begin_batch): nz proofs of ncols1 columns each; ncols = nz * ncols1.  arena: every proof's
    // build lives in its own zstride bytes of the context's WS_BATCH workspace (TreeRef::zstride); flat (nz > 1, zstride == 0):
    // the proofs' columns were gathered into one table of ncols columns and built densely like any other.
    unsigned nz;
    size_t ncols1, zstride;
    size_t off_r_ctr, off_g_ctr;  // byte offsets of a proof's list counters in its arena
    bool no_eval_skip;  // built without its structure passes (option debug_skip 2, measurement only): the "column changed" words
                        // were never written, so the eval must not take them for "constant"
};

// enqueues the builds of a job, the gather of its roots + counters into ONE pinned buffer, and the "built" event
static zigz_status job_build(zigz_commit_job *job) {
    zigz_ctx *ctx = job->ctx;
    const size_t ncols = job->ncols, nv = job->nv;
    CHK(timed_begin(ctx, 2));
    job->whole = job->m_whole;
    {   // build with the hints of the job's begin, whatever the context's options say by now
        const uint64_t s0 = ctx->small_domain_mask, r0 = ctx->run_aware_mask, c0 = ctx->cons_group_mask;
        const bool w0 = ctx->run_aware_materialize;
        ctx->small_domain_mask = job->m_small;
        ctx->run_aware_mask = job->m_run;
        ctx->cons_group_mask = job->m_cons;
        ctx->run_aware_materialize = job->m_whole;
        const zigz_status bs = build_trees(ctx, job->d_cols, job->col_stride, job->N, job->N, nullptr, ncols, ctx->timing, &job->tree);
        ctx->small_domain_mask = s0;
        ctx->run_aware_mask = r0;
        ctx->cons_group_mask = c0;
        ctx->run_aware_materialize = w0;
        CHK(bs);
    }
    if (ctx->timing) HIPCHK(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
    // roots + the counters of the build -> contiguous device buffer -> pinned staging (async), then the "built" event
    // zero-copy: the summary kernel stores the roots and counters into the pinned host buffer itself (no copy command)
    const DoneFlag done = done_flag(ctx, 0);
    job->roots_seq = done.seq;
    launch_job_summary(job->tree, (unsigned)nv, ctx->h_roots, ncols,
                       ctx->stats.run_aware_columns ? ctx->d_run_count : nullptr,
                       (ctx->stats.small_domain_columns || ctx->build_cons_sd) ? ctx->d_sd_fallbacks : nullptr,
                       ctx->build_cons_hinted ? ctx->d_cons_count : nullptr, ctx->stream, done);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipEventRecord(job->built, ctx->stream));
    // hinted columns take levels 0 and 1 (N + N/2 digests) from the tables (waves that had to hash are counted in
    // small_domain_fallback_waves, read in zigz_commit_roots; they are not added back here)
    job->run_cols = ctx->stats.run_aware_columns;
    job->run_dense = ctx->stats.run_aware_dense_nodes;
    job->sd_cols = ctx->stats.small_domain_columns;
    job->cons_hinted = ctx->build_cons_hinted;
    job->cons_levels_nodes = ctx->build_cons_levels_nodes;
    job->cons_sd = ctx->build_cons_sd;
    job->perms0 = (uint64_t)ncols * (2 * job->N - 1) - job->sd_cols * (job->N + job->N / 2);
    job->no_eval_skip = ctx->debug_skip == 2;
    return ZIGZ_OK;
}

static zigz_status job_begin(zigz_ctx *ctx, const uint32_t *d_cols, size_t ncols, size_t col_stride, size_t nv,
                             zigz_commit_job **out) {
    if (ctx->active_job) {
        set_err(ctx, "a commit job is already active on this context");
        return ZIGZ_ERR_BAD_STATE;
    }
    zigz_commit_job *job = new (std::nothrow) zigz_commit_job();
    if (!job) return ZIGZ_ERR_OUT_OF_MEMORY;
    memset(job, 0, sizeof(*job));
    job->ctx = ctx;
    job->ncols = ncols;
    job->nv = nv;
    job->N = (size_t)1 << nv;
    job->col_stride = col_stride;
    job->d_cols = d_cols;
    job->m_small = ctx->small_domain_mask;
    job->m_run = ctx->run_aware_mask;
    job->m_cons = ctx->cons_group_mask;
    job->m_whole = ctx->run_aware_materialize;
    auto body = [&]() -> zigz_status {
        HIPCHK(ctx, hipEventCreateWithFlags(&job->built, hipEventDisableTiming));
        return job_build(job);
    };
    zigz_status st = body();
    if (st != ZIGZ_OK) {
        if (job->built) (void)hipEventDestroy(job->built);
        delete job;
        return st;
    }
    ctx->active_job = job;
    *out = job;
    return ZIGZ_OK;
}

// ---- a batched job: several proofs' columns in ONE commit job (zigz_commit_begin_batch)
// Small traces make a proof's ~35 launches mostly latency (2^16: 13 us of work per launch); nz proofs of the same shape share
// every launch instead.  Two forms, chosen by the size:
//   flat  (N < 2^15: trees that are built densely anyway) -- the proofs' columns are gathered into one table of nz * ncols1
//         columns and committed like any other table;
//   arena (2^15 <= N <= 2^18: the structure-aware levels) -- every proof gets an arena with the same layout for everything
//         its build reads or writes (a copy of its columns, list counters, lists, leader tables, content-addressing table,
//         digest stores, upper levels, slabs for a dropped group), the kernels take the proof from gridDim.z and move every
//         pointer by proof * arena size (kernels.hpp: TreeRef::zstride).  The lists get their WORST-CASE room (every node
//         hashed: affordable at these sizes, ~0.25 GiB per proof at 2^16), so a batched build is never repeated; every column
//         must be hinted run-aware or member of the content-addressed group (the witness's 43 are: host/prover.cpp).
static size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
static zigz_status job_build_batch_arena(zigz_commit_job *job, const uint32_t *const *srcs, size_t src_stride) {
    zigz_ctx *ctx = job->ctx;
    const size_t nc = job->ncols1, N = job->N, npad = N;
    const unsigned nz = job->nz, height = (unsigned)job->nv;
    ColMap R{}, G{};
    for (size_t c = 0; c < nc; c++) {
        if ((job->m_cons >> c) & 1) G.c[G.n++] = (uint8_t)c;
        else if ((job->m_run >> c) & 1) R.c[R.n++] = (uint8_t)c;
        else return ZIGZ_ERR_INVALID_ARGUMENT;  // (a densely built column: not in this form)
    }
    const size_t stride = N;  // column stride inside an arena
    TreeRef t{};
    t.npad = npad;
    for (int c = 0; c < 64; c++) {
        t.slab_of_col[c] = -1;
        t.y_of_col[c] = -1;
        t.g_j_of_col[c] = -1;
    }
    t.lists = 1;
    t.top = run_top_level(npad);
    // ---- the arena's layout (byte offsets, the same for every proof)
    size_t at = 0;
    auto take = [&](size_t bytes) { const size_t o = at; at += al256(bytes); return o; };
    const size_t o_cols = take(nc * stride * 4);
    const size_t o_rctr = take(RUN_CTR_WORDS * 8), o_gctr = take(RUN_CTR_WORDS * 8);
    size_t o_rlist = 0, o_rstage = 0, o_bitmap = 0, o_prev = 0, o_woff = 0, o_ubase = 0, o_rstore = 0;
    size_t meta_n = 0;
    if (R.n) {
        t.r_lists = runs_lists(npad, R.n, nullptr);
        unsigned long long uoff[RUN_MAX_LEVELS] = {0};
        const size_t units = runs_units(npad, R.n, uoff);
        for (unsigned l = 0; l < RUN_MAX_LEVELS; l++) t.ubase_off[l] = uoff[l];
        meta_n = runs_meta_words(npad, R.n);
        o_rlist = take((size_t)t.r_lists.entries * 4);
        o_rstage = take(runs_stage_scratch_bytes(npad, R.n) + 64);
        o_bitmap = take(meta_n * 8);
        o_prev = take(meta_n * 2);
        o_woff = take(meta_n * 2);
        o_ubase = take(units * 4 + 64);
        o_rstore = take((size_t)t.r_lists.entries * 32);
        t.ncols = R.n;
        for (unsigned y = 0; y < R.n; y++) t.y_of_col[R.c[y]] = (signed char)y;
    }
    size_t o_keys = 0, o_idx = 0, o_glist = 0, o_grep = 0, o_gstore = 0, o_slab = 0;
    const size_t key_bytes = 2 * npad * 8;
    if (G.n) {
        t.g_lists = cons_lists(npad, nullptr);
        o_keys = take(key_bytes);
        o_idx = take(2 * npad * 4);
        o_glist = take((size_t)t.g_lists.entries * 4);
        o_grep = take(2 * npad * 4);
        o_gstore = take((size_t)t.g_lists.entries * G.n * 32);
        o_slab = take((size_t)G.n * tree_nodes(npad) * 32);  // where a dropped group's columns are built densely
        t.g_ncols = G.n;
        for (unsigned k = 0; k < G.n; k++) {
            t.g_j_of_col[G.c[k]] = (signed char)k;
            t.slab_of_col[G.c[k]] = (signed char)k;
        }
    }
    const size_t o_upper = take(nc * 512 * 32);
    const size_t S = al256(at);
    if ((size_t)nz * S > ((size_t)48 << 30)) return ZIGZ_ERR_OUT_OF_MEMORY;
    void *w;
    const void *w_before = ctx->ws[WS_BATCH];
    const size_t n_res = nz > ctx->batch_reserve ? nz : ctx->batch_reserve;
    if ((size_t)n_res * S <= ((size_t)48 << 30)) CHK(ws_get(ctx, WS_BATCH, (size_t)n_res * S, &w));
    else CHK(ws_get(ctx, WS_BATCH, (size_t)nz * S, &w));
    uint8_t *a0 = (uint8_t *)w;
    // the content-addressing tables (generation-tagged slots): cleared when the workspace or the layout is new, or the
    // generations run out -- all nz of them with one strided fill
    // (generations of their own -- batch_gen --: the single jobs' table in WS_CONS starts its count over whenever IT is new)
    if (G.n && (w != w_before || ctx->batch_tab_S != S || ctx->batch_tab_nz < nz || ctx->batch_tab_off != o_keys ||
                ctx->batch_gen == 0 || ctx->batch_gen + RUN_MAX_LEVELS + 1 >= 4096)) {
        const size_t n_tabs = ctx->ws_bytes[WS_BATCH] / S;  // (every arena the workspace has room for: a later, larger batch finds them clear)
        HIPCHK(ctx, hipMemset2DAsync(a0 + o_keys, S, 0, key_bytes, n_tabs, ctx->stream));
        ctx->batch_tab_S = S;
        ctx->batch_tab_nz = (unsigned)n_tabs;
        ctx->batch_tab_off = o_keys;
        ctx->batch_gen = 1;
    }
    t.upper = a0 + o_upper;
    MerkleBuild b{};
    b.vals = (const uint32_t *)(a0 + o_cols);
    b.val_stride = stride;
    b.n_values = N;
    b.npad = npad;
    b.rcols = R;
    b.gcols = G;
    if (R.n) {
        t.bitmap = (unsigned long long *)(a0 + o_bitmap);
        t.prev = (unsigned short *)(a0 + o_prev);
        t.woff = (unsigned short *)(a0 + o_woff);
        t.ubase = (uint32_t *)(a0 + o_ubase);
        t.r_store = a0 + o_rstore;
        b.r_list = (uint32_t *)(a0 + o_rlist);
        b.r_stage = a0 + o_rstage;
        b.r_ctr = (unsigned long long *)(a0 + o_rctr);
    }
    if (G.n) {
        t.slab = a0 + o_slab;
        t.g_rep = (const uint32_t *)(a0 + o_grep);
        t.g_store = a0 + o_gstore;
        t.g_dropped = (const unsigned long long *)(a0 + o_gctr) + 8;
        b.g_keys = (unsigned long long *)(a0 + o_keys);
        b.g_idx = (uint32_t *)(a0 + o_idx);
        b.g_list = (uint32_t *)(a0 + o_glist);
        b.g_rep = (uint32_t *)(a0 + o_grep);
        b.g_ctr = (unsigned long long *)(a0 + o_gctr);
        b.g_has_slabs = 1;
        b.g_gen = ctx->batch_gen;
        ctx->batch_gen += t.top + 1;
    }
    b.t = t;
    set_zstride(b, S, nz);
    t = b.t;
    ColSrcs cs{};
    for (unsigned z = 0; z < nz; z++) cs.p[z] = srcs[z];
    launch_gather_cols(cs, nz, nc, N, src_stride, (uint32_t *)(a0 + o_cols), stride, S, ctx->stream);
    launch_zero_counters(nullptr, R.n ? b.r_ctr : nullptr, G.n ? b.g_ctr : nullptr, ctx->stream, nz, S);
    if (ctx->debug_skip != 2) {
        launch_runs_structure(b, ctx->stream, nullptr);
        launch_cons_structure(b, ctx->stream, nullptr);
    }
    for (unsigned l = 0; l <= t.top; l++)
        if (ctx->debug_skip != 1) launch_level_hash(b, l, ctx->stream, nullptr);
    if (height && ctx->debug_skip != 1) launch_merkle_top(t, t.top, height, nc, ctx->stream, nullptr);
    HIPCHK(ctx, hipGetLastError());
    job->tree = t;
    job->d_cols = (const uint32_t *)(a0 + o_cols);
    job->col_stride = stride;
    job->zstride = S;
    job->off_r_ctr = o_rctr;
    job->off_g_ctr = o_gctr;
    job->whole = false;
    const DoneFlag done = done_flag(ctx, 0);
    job->roots_seq = done.seq;
    launch_job_summary(t, height, ctx->h_roots, nc, R.n ? b.r_ctr : nullptr, nullptr, G.n ? b.g_ctr : nullptr, ctx->stream, done);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipEventRecord(job->built, ctx->stream));
    uint64_t level_nodes = 0;
    for (unsigned l = 0; l <= t.top; l++) level_nodes += npad >> l;
    job->run_cols = R.n;
    job->run_dense = (uint64_t)R.n * level_nodes;
    job->sd_cols = 0;
    job->cons_hinted = G.n;
    job->cons_levels_nodes = level_nodes;
    job->cons_sd = 0;
    job->perms0 = (uint64_t)nc * (2 * N - 1);
    job->no_eval_skip = ctx->debug_skip == 2;
    return ZIGZ_OK;
}

extern "C" zigz_status zigz_commit_begin_batch(zigz_ctx *ctx, const uint32_t *const *d_cols, size_t nproofs, size_t ncols,
                                               size_t col_stride, size_t nv, zigz_commit_job **out) {
    ZIGZ_ENTER(ctx);
    if (!ctx || !d_cols || !out || ncols == 0 || nproofs == 0 || nproofs > BATCH_MAX || nv > 40) return ZIGZ_ERR_INVALID_ARGUMENT;
    if (ncols * nproofs > ROOTS_MAX_COLS || col_stride < ((size_t)1 << nv)) return ZIGZ_ERR_INVALID_ARGUMENT;
    for (size_t z = 0; z < nproofs; z++)
        if (!d_cols[z]) return ZIGZ_ERR_INVALID_ARGUMENT;
    if (nproofs == 1) return job_begin(ctx, d_cols[0], ncols, col_stride, nv, out);
    if (ctx->active_job) {
        set_err(ctx, "a commit job is already active on this context");
        return ZIGZ_ERR_BAD_STATE;
    }
    const size_t N = (size_t)1 << nv;
    const bool arena = N >= RUN_MIN_LEAVES && N <= ((size_t)1 << 18) && ncols <= 64 && (ctx->run_aware_mask || ctx->cons_group_mask);
    if (N >= RUN_MIN_LEAVES && !arena) return ZIGZ_ERR_INVALID_ARGUMENT;  // (large tables: one job per proof)
    zigz_commit_job *job = new (std::nothrow) zigz_commit_job();
    if (!job) return ZIGZ_ERR_OUT_OF_MEMORY;
    memset(job, 0, sizeof(*job));
    job->ctx = ctx;
    job->nz = (unsigned)nproofs;
    job->ncols1 = ncols;
    job->ncols = ncols * nproofs;
    job->nv = nv;
    job->N = N;
    auto body = [&]() -> zigz_status {
        HIPCHK(ctx, hipEventCreateWithFlags(&job->built, hipEventDisableTiming));
        if (arena) {
            job->m_small = 0;
            job->m_run = ctx->run_aware_mask;
            job->m_cons = ctx->cons_group_mask;
            return job_build_batch_arena(job, d_cols, col_stride);
        }
        // flat: one table of nz * ncols columns, built densely (no hints: they are per 64 columns of ONE proof)
        const size_t dstride = N < 4 ? 4 : N;
        void *d;
        CHK(ws_get(ctx, WS_COLS, (job->nz > ctx->batch_reserve ? job->nz : ctx->batch_reserve) * ncols * dstride * 4, &d));
        ColSrcs cs{};
        for (size_t z = 0; z < nproofs; z++) cs.p[z] = d_cols[z];
        launch_gather_cols(cs, job->nz, ncols, N, col_stride, (uint32_t *)d, dstride, ncols * dstride * 4, ctx->stream);
        HIPCHK(ctx, hipGetLastError());
        job->d_cols = (const uint32_t *)d;
        job->col_stride = dstride;
        job->m_small = job->m_run = job->m_cons = 0;
        job->m_whole = false;
        return job_build(job);
    };
    const zigz_status st = body();
    if (st != ZIGZ_OK) {
        if (job->built) (void)hipEventDestroy(job->built);
        delete job;
        return st;
    }
    ctx->active_job = job;
    *out = job;
    return ZIGZ_OK;
}

extern "C" zigz_status zigz_commit_begin_dev(zigz_ctx *ctx, const uint32_t *d_cols, size_t ncols, size_t col_stride,
                                             size_t nv, zigz_commit_job **out) {
    ZIGZ_ENTER(ctx);
    if (!ctx || !d_cols || !out || ncols == 0 || nv > 40) return ZIGZ_ERR_INVALID_ARGUMENT;
    if (ncols > ROOTS_MAX_COLS) return ZIGZ_ERR_INVALID_ARGUMENT;
    if (col_stride < ((size_t)1 << nv)) return ZIGZ_ERR_INVALID_ARGUMENT;
    return job_begin(ctx, d_cols, ncols, col_stride, nv, out);
}

extern "C" zigz_status zigz_commit_begin(zigz_ctx *ctx, const uint64_t *cols, size_t ncols, size_t col_stride,
                                         size_t nv, zigz_commit_job **out) {
    ZIGZ_ENTER(ctx);
    if (!ctx || !cols || !out || ncols == 0 || nv > 40) return ZIGZ_ERR_INVALID_ARGUMENT;
    if (ncols > ROOTS_MAX_COLS) return ZIGZ_ERR_INVALID_ARGUMENT;
    const size_t N = (size_t)1 << nv;
    if (col_stride < N) return ZIGZ_ERR_INVALID_ARGUMENT;
    if (ctx->active_job) return ZIGZ_ERR_BAD_STATE;
    const size_t dstride = N < 4 ? 4 : N;
    void *d;
    CHK(ws_get(ctx, WS_COLS, ncols * dstride * 4, &d));
    uint32_t *d_cols = (uint32_t *)d;
    if (col_stride == N && dstride == N) {
        CHK(upload_u64(ctx, cols, ncols * N, d_cols, false));
    } else {
        for (size_t c = 0; c < ncols; c++) CHK(upload_u64(ctx, cols + c * col_stride, N, d_cols + c * dstride, false));
    }
    return job_begin(ctx, d_cols, ncols, dstride, nv, out);
}

extern "C" zigz_status zigz_commit_roots(zigz_commit_job *job, uint8_t *roots) {
    if (job) ZIGZ_ENTER(job->ctx);
    if (!job || !roots) return ZIGZ_ERR_INVALID_ARGUMENT;
    zigz_ctx *ctx = job->ctx;
    if (job->state != 0) return ZIGZ_ERR_BAD_STATE;
    const unsigned long long *h_cnt = (const unsigned long long *)(ctx->h_roots + job->ncols * 32);
    for (int attempt = 0;; attempt++) {
        const unsigned long long *h_done = (const unsigned long long *)(ctx->h_roots + ROOTS_MAX_COLS * 32 + JOB_SUMMARY_WORDS * 8);
        if (!(g_sleep_wait.load() && !ctx->timing && sleep_wait(h_done, job->roots_seq)))
            HIPCHK(ctx, hipEventSynchronize(job->built));
        // What the lists of the structure-aware levels needed: the context remembers it for its next builds, and a build that
        // ran out of room (or found its group dropped with nowhere to build the columns densely) is repeated here with more.
        // This is the one place where a proof may pay for a second build: the first time a context meets a new kind of trace.
        const unsigned long long flags = h_cnt[6];
        const bool r_over = (flags & 1) != 0, g_over = ((flags >> 8) & 1) != 0, g_noslab = ((flags >> 8) & 2) != 0;
        const bool dropped = h_cnt[4] != 0;
        ListCaps &c = ctx->caps;
        bool again = false;
        if (job->zstride) {  // a batched job's lists have their worst-case room: nothing to learn, nothing can have run out
            for (unsigned z = 0; z < job->nz; z++)
                if (h_cnt[(size_t)z * JOB_SUMMARY_WORDS + 6]) {
                    set_err(ctx, "batched commit job: a list ran out of its worst-case room (proof %u)", z);
                    return ZIGZ_ERR_BAD_STATE;
                }
        } else if (job->tree.lists && c.npad == job->N) {
            for (unsigned l = 0; l <= job->tree.top; l++) {
                const unsigned long long ru = h_cnt[8 + l], gu = h_cnt[8 + RUN_MAX_LEVELS + l];
                if (job->run_cols && (r_over ? ru > c.r[l] : ru * 10 > (unsigned long long)c.r[l] * 8))
                    c.r[l] = (unsigned)(ru + ru / 4 + 64);
                if (job->cons_hinted && !dropped && (g_over ? gu > c.g[l] : gu * 10 > (unsigned long long)c.g[l] * 8))
                    c.g[l] = (unsigned)(gu + gu / 4 + 64);
            }
            if (dropped) c.g_slabs = true;  // this context's traces do not repeat: give the group's columns slabs from now on
            if (job->cons_hinted) {
                c.g_drops = dropped ? c.g_drops + 1 : 0;
                // ... and after the second drop in a row, skip the attempt for 15 jobs -- twice as many after every further attempt
                // that is dropped again (a context shared by a service's lanes sees hundreds of jobs of one kind of trace)
                if (c.g_drops >= 2) c.g_skip = 15u << (c.g_drops - 2 < 6 ? c.g_drops - 2 : 6);
            }
            again = r_over || (g_over && !dropped) || g_noslab;
        }
        if (!again) break;
        if (attempt >= 3) {
            set_err(ctx, "commit job: the lists of the structure-aware levels still do not fit after %d builds", attempt + 1);
            return ZIGZ_ERR_BAD_STATE;
        }
        ctx->stats.rebuilds++;
        CHK(job_build(job));
    }
    memcpy(roots, ctx->h_roots, job->ncols * 32);
    if (job->zstride) {  // a batched job: the sums over its proofs
        const uint64_t nz = job->nz;
        uint64_t r_hashed = 0, g_hashed = 0, g_kept = 0, g_distinct = 0, constant = 0, dense_g = 0;
        for (unsigned z = 0; z < job->nz; z++) {
            const unsigned long long *h = h_cnt + (size_t)z * JOB_SUMMARY_WORDS;
            r_hashed += job->run_cols ? h[0] : 0;
            constant += job->run_cols ? h[7] : 0;
            if (job->cons_hinted) {
                g_distinct += h[5];
                if (!h[4]) { g_kept++; g_hashed += h[3]; }
                else dense_g += job->cons_hinted * job->cons_levels_nodes;  // dropped: its columns were hashed densely
            }
        }
        ctx->stats.run_aware_columns = job->run_cols;
        ctx->stats.run_aware_dense_nodes = job->run_dense * nz;
        ctx->stats.run_aware_hashed = r_hashed;
        ctx->stats.small_domain_columns = 0;
        ctx->stats.small_domain_fallback_waves = 0;
        ctx->stats.cons_columns = g_kept ? job->cons_hinted : 0;
        ctx->stats.cons_dense_nodes = job->cons_hinted * job->cons_levels_nodes * g_kept;
        ctx->stats.cons_hashed = g_hashed;
        ctx->stats.cons_probe_distinct = g_distinct;
        ctx->stats.list_hash_perms = r_hashed + g_hashed + dense_g;
        ctx->stats.keccak_permutations = job->perms0 * nz - (job->run_dense * nz - r_hashed) - (ctx->stats.cons_dense_nodes - g_hashed);
        job->const_cols = constant;
        ctx->stats.eval_constant_columns = constant;
        job->state = 1;
        return ZIGZ_OK;
    }
    // the run-aware levels hashed h_cnt[0] of their run_dense nodes
    const uint64_t N = job->N;
    ctx->stats.run_aware_columns = job->run_cols;
    ctx->stats.run_aware_dense_nodes = job->run_dense;
    ctx->stats.small_domain_columns = job->sd_cols;
    ctx->stats.keccak_permutations = job->perms0;
    ctx->stats.run_aware_hashed = job->run_cols ? h_cnt[0] : 0;
    job->const_cols = job->run_cols ? h_cnt[7] : 0;
    ctx->stats.eval_constant_columns = job->const_cols;
    ctx->stats.keccak_permutations -= ctx->stats.run_aware_dense_nodes - ctx->stats.run_aware_hashed;
    ctx->stats.small_domain_fallback_waves = job->sd_cols ? h_cnt[1] : 0;
    ctx->stats.list_hash_perms = ctx->stats.run_aware_hashed;
    // the group: kept (digests computed for its cons_dense_nodes nodes: h_cnt[3]) or dropped on the device (its small-domain
    // members then took levels 0 and 1 from the tables, everything else was hashed densely)
    ctx->stats.cons_columns = ctx->stats.cons_dense_nodes = ctx->stats.cons_hashed = 0;
    ctx->stats.cons_probe_distinct = 0;
    if (job->cons_hinted) {
        ctx->stats.cons_probe_distinct = h_cnt[5];
        if (!h_cnt[4]) {
            ctx->stats.cons_columns = job->cons_hinted;
            ctx->stats.cons_dense_nodes = job->cons_hinted * job->cons_levels_nodes;
            ctx->stats.cons_hashed = h_cnt[3];
            ctx->stats.keccak_permutations -= ctx->stats.cons_dense_nodes - ctx->stats.cons_hashed;
            ctx->stats.list_hash_perms += ctx->stats.cons_hashed;
        } else {
            ctx->stats.small_domain_columns += job->cons_sd;
            ctx->stats.keccak_permutations -= job->cons_sd * (N + N / 2);
            ctx->stats.small_domain_fallback_waves += h_cnt[2];
            ctx->stats.list_hash_perms += job->cons_hinted * job->cons_levels_nodes - job->cons_sd * (N + N / 2);
        }
    }
    if (ctx->timing) {
        float ms = 0;
        HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]));
        ctx->stats.merkle_build_us = (double)ms * 1000.0;
        CHK(keccak_times_collect(ctx));
    }
    job->state = 1;
    return ZIGZ_OK;
}

extern "C" zigz_status zigz_commit_open_all(zigz_commit_job *job, const uint64_t *points, uint64_t *values,
                                            uint64_t *indices, uint64_t *leaves, uint8_t *siblings, uint8_t *dirs) {
    ZIGZ_NOTHROW_BEGIN
    if (job) ZIGZ_ENTER(job->ctx);
    if (!job || !values || !indices || !leaves) return ZIGZ_ERR_INVALID_ARGUMENT;
    zigz_ctx *ctx = job->ctx;
    if (job->state != 1) return ZIGZ_ERR_BAD_STATE;
    const size_t nv = job->nv, ncols = job->ncols;
    if (nv && (!points || !siblings || !dirs)) return ZIGZ_ERR_INVALID_ARGUMENT;
    // What a proof needs back -- 43 evaluations, leaves, sibling digests, directions: ~30 KB -- is written by the kernels
    // straight into the pinned staging buffer, and the indices are read from it: no copy command at all, one wait (0.51 vs
    // 0.52 ms of GPU per proof with 14 lanes; ONE packed copy through the same buffer was slower than the five small copies
    // to pageable memory of the fallback below: 0.56 ms).
    const size_t sib_b = ncols * nv * 32, leaf_b = ncols * 4, val_b = ncols * 4, dir_b = ncols * nv;
    const size_t out_b = sib_b + leaf_b + val_b + dir_b;
    if (out_b <= PIN_WORDS * 8 / 2 && nv * ncols * 4 + ncols * 8 <= PIN_WORDS * 8 / 2) {
        uint8_t *h = (uint8_t *)ctx->h_pin;
        uint8_t *z_sib = h;
        uint32_t *z_leaf = (uint32_t *)(h + sib_b), *z_val = z_leaf + ncols;
        uint8_t *z_dirs = (uint8_t *)(z_val + ncols);
        uint64_t *h_idx = ctx->h_pin + PIN_WORDS - ncols;
        for (size_t c = 0; c < ncols; c++) {
            h_idx[c] = nv == 0 ? 0 : points[c * nv] % ((uint64_t)1 << nv);  // pointToIndex
            indices[c] = h_idx[c];
        }
        // columns the run-aware structure pass of THIS job found constant are not read again (EvalSkip, kernels.hpp): of the 43
        // witness columns of a program that uses a handful of registers, most
        EvalSkip skip;
        if (job->zstride) {  // a batched job in arenas: column c = column c % ncols1 of proof c / ncols1 (kernels.hpp: EvalSkip)
            skip.ncols1 = (unsigned)job->ncols1;
            skip.z_in = job->zstride / 4;
            skip.z_changed = job->zstride / 8;
            memcpy(skip.y_of_col, job->tree.y_of_col, sizeof(skip.y_of_col));
            if (job->run_cols && !job->no_eval_skip) {
                skip.changed = (const unsigned long long *)((const uint8_t *)ctx->ws[WS_BATCH] + job->off_r_ctr) + RUN_CHANGED;
                ctx->stats.eval_constant_columns = job->const_cols;
            } else {
                ctx->stats.eval_constant_columns = 0;
            }
            CHK(timed_begin(ctx, 4));
            CHK(dev_eval_radix(ctx, job->d_cols, job->col_stride, ncols, nv, points, z_val, &skip));
            CHK(timed_end(ctx, 4, &ctx->stats.eval_us));
        } else {
            if (job->tree.lists && job->run_cols && job->col_stride >= job->N && !job->no_eval_skip) {
                skip.changed = ctx->d_run_count + RUN_CHANGED;  // (the job's own counters: no other build on the context adds to them)
                ctx->stats.eval_constant_columns = job->const_cols;  // (what dev_eval_radix sizes its launch by: this job's count)
                memcpy(skip.y_of_col, job->tree.y_of_col, sizeof(skip.y_of_col));
            }
            CHK(timed_begin(ctx, 4));
            CHK(dev_eval_folds(ctx, job->d_cols, job->col_stride, ncols, nv, points, z_val, skip.changed ? &skip : nullptr));
            CHK(timed_end(ctx, 4, &ctx->stats.eval_us));
        }
        const DoneFlag done = done_flag(ctx, 1);
        launch_paths(job->tree, job->tree.npad, (unsigned)nv, job->d_cols, job->col_stride, h_idx, z_sib, z_dirs, z_leaf,
                     job->zstride ? job->ncols1 : ncols, ctx->stream, done);
        HIPCHK(ctx, hipGetLastError());
        if (!(g_sleep_wait.load() && !ctx->timing && sleep_wait(done.flag, done.seq)))
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if (sib_b) memcpy(siblings, z_sib, sib_b);
        for (size_t c = 0; c < ncols; c++) {
            leaves[c] = z_leaf[c];
            values[c] = z_val[c];
        }
        if (dir_b) memcpy(dirs, z_dirs, dir_b);
        CHK(bind_pool_collect(ctx));
        job->state = 2;
        return ZIGZ_OK;
    }
    if (job->zstride) {
        set_err(ctx, "batched commit job: the openings do not fit the staging buffer");
        return ZIGZ_ERR_INVALID_ARGUMENT;
    }
    void *dv;
    CHK(ws_get(ctx, WS_SCRATCH, ncols * 4 + 64, &dv));
    CHK(timed_begin(ctx, 4));
    CHK(dev_eval_folds(ctx, job->d_cols, job->col_stride, ncols, nv, points, (uint32_t *)dv));
    CHK(timed_end(ctx, 4, &ctx->stats.eval_us));
    std::vector<uint32_t> hv(ncols);
    HIPCHK(ctx, hipMemcpyAsync(hv.data(), dv, ncols * 4, hipMemcpyDeviceToHost, ctx->stream));
    std::vector<uint64_t> idx(ncols);
    for (size_t c = 0; c < ncols; c++) {
        idx[c] = nv == 0 ? 0 : points[c * nv] % ((uint64_t)1 << nv);  // pointToIndex
        indices[c] = idx[c];
    }
    CHK(open_paths(ctx, job->tree, (unsigned)nv, job->d_cols, job->col_stride, idx.data(), ncols, siblings, dirs, leaves));
    for (size_t c = 0; c < ncols; c++) values[c] = hv[c];
    CHK(bind_pool_collect(ctx));
    job->state = 2;
    return ZIGZ_OK;
    ZIGZ_NOTHROW_END(job->ctx)
}

extern "C" zigz_status zigz_commit_job_tree(zigz_commit_job *job, const void **d_tree, size_t *bytes_per_column) {
    if (job) ZIGZ_ENTER(job->ctx);
    if (!job || !d_tree || !bytes_per_column) return ZIGZ_ERR_INVALID_ARGUMENT;
    // node-addressed trees of every column exist only when nothing was list-built, or the job was begun with
    // "run_aware_materialize" (otherwise the list-built levels live in list order: there is no whole tree to look at)
    if (job->tree.lists && !job->whole) return ZIGZ_ERR_BAD_STATE;
    HIPCHK(job->ctx, hipEventSynchronize(job->built));
    *d_tree = job->tree.slab;
    *bytes_per_column = tree_nodes(job->N) * 32;
    return ZIGZ_OK;
}

extern "C" void zigz_commit_end(zigz_commit_job *job) {
    if (job) ZIGZ_ENTER(job->ctx);
    if (!job) return;
    zigz_ctx *ctx = job->ctx;
    (void)hipStreamSynchronize(ctx->stream);
    if (job->built) (void)hipEventDestroy(job->built);
    if (ctx->active_job == job) ctx->active_job = nullptr;
    delete job;
}

// ------------------------------------------------------------------ Lasso (simplified), lasso_prover.zig:103-252
static void flat_commit(const uint32_t *ev, size_t n, uint8_t out[32]) {  // commitToPolynomial, :242-252 (K10, host)
    Sha3_256 h;
    for (size_t i = 0; i < n; i++) h.update_le64(ev[i]);
    h.finalize(out);
}

extern "C" zigz_status zigz_lasso_fingerprints(zigz_ctx *ctx, const uint64_t *rows_in, size_t rows, size_t width,
                                               uint64_t *out) {
    ZIGZ_ENTER(ctx);
    if (!ctx || !rows_in || !out || width == 0) return ZIGZ_ERR_INVALID_ARGUMENT;
    if (rows == 0) return ZIGZ_OK;
    uint32_t *d_rows;
    CHK(stage_in(ctx, rows_in, rows * width, &d_rows));
    void *d_o;
    CHK(ws_get(ctx, WS_OUT32, rows * 4, &d_o));
    launch_lasso_fingerprints(d_rows, rows, width, (uint32_t *)d_o, ctx->stream);
    HIPCHK(ctx, hipGetLastError());
    return download_u64(ctx, (uint32_t *)d_o, rows, out);
}

extern "C" zigz_status zigz_lasso_prove(zigz_ctx *ctx, const uint64_t *table, size_t table_rows, const uint64_t *queries,
                                        size_t n_queries, size_t n_in, size_t n_out, size_t *nv_out, uint64_t *rounds,
                                        uint64_t *point, uint64_t *final_eval, uint8_t query_commitment[32],
                                        uint8_t table_commitment[32]) {
    ZIGZ_ENTER(ctx);
    if (!ctx) return ZIGZ_ERR_INVALID_ARGUMENT;
    if (n_queries == 0) return ZIGZ_ERR_NO_QUERIES;  // :108-110
    const size_t w = n_in + n_out;
    if (!table || !queries || w == 0 || !nv_out || !final_eval || !query_commitment || !table_commitment)
        return ZIGZ_ERR_INVALID_ARGUMENT;
    CHK(mle_check(table_rows));  // Multilinear.init(table_evals), :124
    if (n_queries > ((size_t)1 << 40)) return ZIGZ_ERR_TOO_MANY_QUERIES;
    const size_t padded = ceil_pow2(n_queries);  // :131
    // The two flat SHA3 commitments (K10, :242-252) are sequential sponges over 8 B per element -- the longest part of a
    // Lasso proof by far (2^20 queries: 8 MiB = 62 k dependent permutations) -- and depend only on the fingerprints, not
    // on the sumcheck: each runs on a helper thread as soon as its fingerprints are on the host, underneath the uploads,
    // the other fingerprint kernel and the whole GPU sumcheck.
    std::vector<uint32_t> hq, ht;  // declared before the threads that read them: destroyed after the joiner below
    std::thread th_table, th_query;
    struct Joiner {
        std::thread &a, &b;
        ~Joiner() {
            if (a.joinable()) a.join();
            if (b.joinable()) b.join();
        }
    } joiner{th_table, th_query};
    ZIGZ_NOTHROW_BEGIN
    hq.resize(padded);
    ht.resize(table_rows);
    // fingerprints of table rows and queries (K9)
    void *d_fp;
    CHK(ws_get(ctx, WS_LASSO, (table_rows + padded) * 4, &d_fp));
    uint32_t *d_tev = (uint32_t *)d_fp, *d_qev = d_tev + table_rows;
    uint32_t *d_rows;
    CHK(stage_in(ctx, table, table_rows * w, &d_rows));
    CHK(timed_begin(ctx, 0));
    launch_lasso_fingerprints(d_rows, table_rows, w, d_tev, ctx->stream);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(ht.data(), d_tev, table_rows * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    th_table = std::thread([&ht, table_rows, table_commitment] { flat_commit(ht.data(), table_rows, table_commitment); });
    CHK(stage_in(ctx, queries, n_queries * w, &d_rows));
    HIPCHK(ctx, hipMemsetAsync(d_qev, 0, padded * 4, ctx->stream));  // zero-pad, :139-142
    launch_lasso_fingerprints(d_rows, n_queries, w, d_qev, ctx->stream);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(hq.data(), d_qev, padded * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    th_query = std::thread([&hq, padded, query_commitment] { flat_commit(hq.data(), padded, query_commitment); });
    *nv_out = log2_floor(padded);
    if (padded == 1) return ZIGZ_ERR_NO_VARIABLES;  // SumcheckProver.prove on a 0-variable poly, :160
    if (!rounds || !point) return ZIGZ_ERR_INVALID_ARGUMENT;
    CHK(sumcheck_core(ctx, d_qev, padded, nullptr, nullptr, rounds, point, final_eval));
    th_table.join();
    th_query.join();
    return ZIGZ_OK;
    ZIGZ_NOTHROW_END(ctx)
}

extern "C" zigz_status zigz_lasso_prove_with_mapping(zigz_ctx *ctx, const uint64_t *table, size_t table_rows,
                                                     const uint64_t *queries, size_t n_queries, size_t n_in,
                                                     size_t n_out, const uint64_t *mapping, size_t n_mapping,
                                                     size_t *nv_out, uint64_t *rounds, uint64_t *point,
                                                     uint64_t *final_eval, uint8_t query_commitment[32],
                                                     uint8_t table_commitment[32]) {
    ZIGZ_NOTHROW_BEGIN
    ZIGZ_ENTER(ctx);
    if (!ctx) return ZIGZ_ERR_INVALID_ARGUMENT;
    if (n_queries != n_mapping) return ZIGZ_ERR_MAPPING_LENGTH_MISMATCH;  // :185-187
    const size_t w = n_in + n_out;
    if (n_queries && (!table || !queries || !mapping)) return ZIGZ_ERR_INVALID_ARGUMENT;
    for (size_t j = 0; j < n_queries; j++) {  // O(Q*w) host-side equality scan of caller data, :190-201
        if (mapping[j] >= table_rows) return ZIGZ_ERR_INVALID_MAPPING;
        if (memcmp(queries + j * w, table + mapping[j] * w, w * sizeof(uint64_t)) != 0)
            return ZIGZ_ERR_QUERY_TABLE_MISMATCH;
    }
    return zigz_lasso_prove(ctx, table, table_rows, queries, n_queries, n_in, n_out, nv_out, rounds, point, final_eval,
                            query_commitment, table_commitment);
    ZIGZ_NOTHROW_END(ctx)
}

// ------------------------------------------------------------------ measurement hook: one hot kernel on synthetic tables
// Launches the named kernel `iters` times on a device-resident synthetic table (ncols columns of 2^nv canonical
// elements) and reports each launch's own duration (dispatch begin / end timestamps, what rocprofv3 --kernel-trace
// shows).  cold != 0: a 1 GiB read-only sweep precedes every launch, so the inputs come from HBM and not from the
// 256 MB Infinity Cache or L2 (a read sweep leaves no dirty lines behind, unlike a memset).
extern "C" zigz_status zigz_bench_kernel(zigz_ctx *ctx, const char *kernel, size_t nv, size_t ncols, int iters, int cold,
                                         zigz_bench_result *out) {
    ZIGZ_ENTER(ctx);
    if (!ctx || !kernel || !out || nv < 13 || nv > 30 || ncols == 0 || ncols > 4096 || iters < 1) return ZIGZ_ERR_INVALID_ARGUMENT;
    if (ctx->active_job) return ZIGZ_ERR_BAD_STATE;
    if (iters > 64) iters = 64;
    const size_t N = (size_t)1 << nv;
    memset(out, 0, sizeof(*out));
    enum { K_BIND, K_BIND_SUMS, K_HALF, K_BLOCK, K_FOLD, K_LEAVES, K_LEVEL, K_LASSO } which;
    if (!strcmp(kernel, "k_bind_vec")) which = K_BIND;
    else if (!strcmp(kernel, "k_bind_vec_sums")) which = K_BIND_SUMS;
    else if (!strcmp(kernel, "k_half_sums")) which = K_HALF;
    else if (!strcmp(kernel, "k_block_sums")) which = K_BLOCK;
    else if (!strcmp(kernel, "k_radix_fold")) which = K_FOLD;
    else if (!strcmp(kernel, "k_keccak_leaves")) which = K_LEAVES;
    else if (!strcmp(kernel, "k_keccak_level")) which = K_LEVEL;
    else if (!strcmp(kernel, "k_lasso_fingerprints")) which = K_LASSO;
    else return ZIGZ_ERR_INVALID_ARGUMENT;
    if (which == K_BLOCK && ncols != 1) return ZIGZ_ERR_INVALID_ARGUMENT;
    if (which == K_FOLD && (nv < 14 || nv > 24)) return ZIGZ_ERR_INVALID_ARGUMENT;
    if (which != K_BLOCK && ncols * 32 > 4096) return ZIGZ_ERR_INVALID_ARGUMENT;
    void *d_in_v, *d_out_v = nullptr, *d_tree_v = nullptr;
    const size_t in_elems = which == K_LASSO ? ncols * N * 3 : ncols * N;
    CHK(ws_get(ctx, WS_COLS, in_elems * 4, &d_in_v));
    uint32_t *d_in = (uint32_t *)d_in_v;
    launch_fill_pattern(d_in, in_elems, 0x5A49475A, ctx->stream);
    const unsigned k2 = 10, k1 = nv >= k2 ? (unsigned)nv - k2 : 0;
    const size_t m = (size_t)1 << k2, nb = (size_t)1 << k1, groups = radix_fold_groups(nb);
    unsigned long long *d_part = nullptr;
    uint32_t *d_w1 = nullptr;
    switch (which) {
    case K_BIND: case K_BIND_SUMS: case K_LASSO:
        CHK(ws_get(ctx, WS_FOLD, ncols * N * 4, &d_out_v));
        break;
    case K_FOLD:
        CHK(ws_get(ctx, WS_FOLD, ncols * (groups * m * 8 + nb * 4) + 256, &d_out_v));
        d_part = (unsigned long long *)d_out_v;
        d_w1 = (uint32_t *)(d_part + ncols * groups * m);
        launch_fill_pattern(d_w1, ncols * nb, 7, ctx->stream);
        break;
    case K_LEAVES: case K_LEVEL:
        CHK(ws_get(ctx, WS_TREE, ncols * tree_nodes(N) * 32, &d_tree_v));
        launch_keccak_leaves(d_in, N, N, N, (uint8_t *)d_tree_v, tree_nodes(N), ncols, ctx->stream);
        break;
    default: break;
    }
    if (cold && !ctx->d_flush) {
        HIPCHK(ctx, hipMalloc(&ctx->d_flush, FLUSH_BYTES));
        launch_fill_pattern((uint32_t *)ctx->d_flush, FLUSH_BYTES / 4, 99, ctx->stream);
    }
    HIPCHK(ctx, hipGetLastError());
    const uint32_t r_m = host_to_mont(123456789);
    for (int it = 0; it < iters; it++) {
        if (which != K_FOLD && which != K_LEAVES && which != K_LEVEL && which != K_LASSO)
            HIPCHK(ctx, hipMemsetAsync(ctx->d_sums, 0, 4096 * sizeof(unsigned long long), ctx->stream));
        if (cold) {
            const SumsLayout fl = half_sums_layout(FLUSH_BYTES / 4, 1, 2048);
            launch_half_sums((const uint32_t *)ctx->d_flush, FLUSH_BYTES / 4, FLUSH_BYTES / 4, 1, ctx->d_sums + 4096, ctx->stream, nullptr, &fl);
        }
        const KTime kt{ctx->pool[2 * it], ctx->pool[2 * it + 1]};
        switch (which) {
        case K_BIND:
            launch_bind(d_in, N, (uint32_t *)d_out_v, N / 2, N / 2, ncols, r_m, nullptr, nullptr, ctx->stream, &kt);
            break;
        case K_BIND_SUMS:
        {   // as bind_with_sums launches it
            const SumsLayout bl = bind_sums_layout(N / 2, ncols, 4096);
            launch_bind(d_in, N, (uint32_t *)d_out_v, N / 2, N / 2, ncols, r_m, nullptr, ctx->d_sums, ctx->stream, &kt, &bl);
            break;
        }
        case K_HALF: {  // as dev_half_sums launches it: padded, replicated counters
            const SumsLayout lay = half_sums_layout(N, ncols, 4096);
            launch_half_sums(d_in, N, N, ncols, ctx->d_sums, ctx->stream, &kt, &lay);
            break;
        }
        case K_BLOCK: launch_block_sums(d_in, N, N, (unsigned)nv - 10, 1, ctx->d_sums, SumsLayout{0, 1, 0, 1}, ctx->stream, &kt); break;
        case K_FOLD:
            launch_radix_fold(d_in, N, m, nb, d_w1, nb, d_part, groups * m, ncols, ctx->stream, kt.start, kt.stop);
            break;
        case K_LEAVES:
            launch_keccak_leaves(d_in, N, N, N, (uint8_t *)d_tree_v, tree_nodes(N), ncols, ctx->stream, &kt);
            break;
        case K_LEVEL:
            launch_keccak_level((uint8_t *)d_tree_v, tree_nodes(N), tree_level_offset(N, 0), tree_level_offset(N, 1), N / 2, ncols,
                                ctx->stream, &kt);
            break;
        case K_LASSO: launch_lasso_fingerprints(d_in, ncols * N, 3, (uint32_t *)d_out_v, ctx->stream, &kt); break;
        }
        HIPCHK(ctx, hipGetLastError());
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    double sum = 0, mn = 1e30, mx = 0;
    for (int it = 0; it < iters; it++) {
        float ms = 0;
        HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->pool[2 * it], ctx->pool[2 * it + 1]));
        const double us = (double)ms * 1000.0;
        sum += us;
        if (us < mn) mn = us;
        if (us > mx) mx = us;
    }
    out->avg_us = sum / iters;
    out->min_us = mn;
    out->max_us = mx;
    out->launches = (uint32_t)iters;
    const uint64_t n_all = (uint64_t)ncols * N;
    switch (which) {  // algorithmic bytes per launch (SURVEY s8d, 4 B per element)
    case K_BIND: case K_BIND_SUMS: out->algorithmic_bytes = 6 * n_all; out->units = n_all; break;  // read n, write n/2
    case K_HALF: case K_BLOCK: out->algorithmic_bytes = 4 * n_all; out->units = n_all; break;
    case K_FOLD: out->algorithmic_bytes = ncols * (N * 4 + groups * m * 8); out->units = n_all; break;
    case K_LEAVES: out->algorithmic_bytes = n_all * (4 + 32); out->units = n_all; break;            // 1 permutation per leaf
    case K_LEVEL: out->algorithmic_bytes = (n_all / 2) * (64 + 32); out->units = n_all / 2; break;  // 1 permutation per node
    case K_LASSO: out->algorithmic_bytes = n_all * 16; out->units = n_all; break;                    // 3 x 4 B in, 4 B out per row
    }
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
