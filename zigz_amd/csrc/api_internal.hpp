#pragma once
// Shared by the translation units behind the C ABI of libzigz_hip.so (include/zigz_hip.h) -- api.cpp (contexts, workspaces,
// boundary conversion, transcript), api_mle.cpp (MLE ops, sumcheck), api_commit.cpp (Merkle trees, commit jobs), api_misc.cpp
// (Lasso, the measurement hook): the context, its workspaces, the error macros and the helpers one unit needs from another.  No CPU fallback for field or hash work on the data path: the
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
    unsigned g_kept;           // consecutive builds that kept it (from the second on the probe pass is left out)
    unsigned r_last[RUN_MAX_LEVELS], g_last[RUN_MAX_LEVELS];  // the longest sub-list of the LAST build per level (0: none yet): launch sizing only
    bool last_dropped;  // ... and whether it dropped its group (whose columns are then hashed densely by the level launches)
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
constexpr unsigned BATCH_ARENA_MAX_NV = 20;  // largest table of the arena form (worst-case lists: 3.8 GiB per proof of 43 columns at 2^20)


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

inline bool is_pow2(size_t n) { return n && !(n & (n - 1)); }
inline unsigned log2_floor(size_t n) { unsigned l = 0; while (n > 1) { n >>= 1; l++; } return l; }
inline size_t ceil_pow2(size_t n) { size_t v = 1; while (v < n) v <<= 1; return v; }


// ---- helpers one translation unit needs from another (hidden: not part of the ABI)
#pragma GCC visibility push(hidden)
void set_err(zigz_ctx *ctx, const char *fmt, ...);
zigz_status ws_get(zigz_ctx *ctx, int slot, size_t bytes, void **out);
bool sleep_wait(const unsigned long long *flag, unsigned long long seq);
bool spin_wait(const unsigned long long *flag, unsigned long long seq);
DoneFlag done_flag(zigz_ctx *ctx, int which);
zigz_status log_launch(zigz_ctx *ctx, int cls, uint64_t perms, hipEvent_t start, hipEvent_t stop, hipEvent_t first, double *dur_us);
zigz_status upload_u64(zigz_ctx *ctx, const uint64_t *h_in, size_t n, uint32_t *d_out, bool reduce);
zigz_status download_u64(zigz_ctx *ctx, const uint32_t *d_in, size_t n, uint64_t *h_out);
zigz_status mle_check(size_t n);
uint32_t host_to_mont(uint64_t canonical);
zigz_status read_u64(zigz_ctx *ctx, const void *d_src, size_t words, uint64_t *dst);
zigz_status dev_half_sums(zigz_ctx *ctx, const uint32_t *d_in, size_t n, uint64_t out[2]);
zigz_status timed_begin(zigz_ctx *ctx, int ev);
zigz_status timed_end(zigz_ctx *ctx, int ev, double *us_out);
zigz_status bind_launch(zigz_ctx *ctx, const uint32_t *d_in, size_t in_stride, uint32_t *d_out, size_t out_stride,
                               size_t half, size_t ncols, uint32_t r_m, const uint32_t *d_r_m, unsigned long long *d_sums,
                               const SumsLayout *lay = nullptr);
void bind_pool_reset(zigz_ctx *ctx);
zigz_status bind_pool_collect(zigz_ctx *ctx);
zigz_status dev_eval_radix(zigz_ctx *ctx, const uint32_t *d_cols, size_t col_stride, size_t ncols, size_t nv,
                                  const uint64_t *points, uint32_t *d_vals, const EvalSkip *skip = nullptr);
zigz_status dev_eval_folds(zigz_ctx *ctx, const uint32_t *d_cols, size_t col_stride, size_t ncols, size_t nv,
                                  const uint64_t *points /*host, ncols*nv*/, uint32_t *d_vals, const EvalSkip *skip = nullptr);
zigz_status sumcheck_core(zigz_ctx *ctx, const uint32_t *d_in, size_t n, uint32_t *d_scratch,
                                 const uint64_t *fixed, uint64_t *rounds, uint64_t *point, uint64_t *final_eval);
zigz_status stage_in(zigz_ctx *ctx, const uint64_t *in, size_t n, uint32_t **d_out);
zigz_status build_trees(zigz_ctx *ctx, const uint32_t *d_vals, size_t val_stride, size_t n_values, size_t npad,
                               uint8_t *d_slab, size_t ncols, bool record = false, TreeRef *ref = nullptr);
zigz_status keccak_times_collect(zigz_ctx *ctx);
extern std::atomic<int> g_sleep_wait;
#pragma GCC visibility pop
