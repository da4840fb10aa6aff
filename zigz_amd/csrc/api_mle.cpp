// C ABI of libzigz_hip.so, part 2: the MLE operations (bind, round sums, eval) and the sumcheck provers (per-round, radix, sharded by
// rows over several GPUs).  The only host arithmetic is the sequential SHA3 Fiat-Shamir sponge and O(v) scalar bookkeeping.
#include "api_internal.hpp"

using namespace zk;

// ------------------------------------------------------------------ device-resident MLE ops
zigz_status dev_half_sums(zigz_ctx *ctx, const uint32_t *d_in, size_t n, uint64_t out[2]) {
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

zigz_status timed_begin(zigz_ctx *ctx, int ev) {
    if (ctx->timing) HIPCHK(ctx, hipEventRecord(ctx->ev[ev], ctx->stream));
    return ZIGZ_OK;
}
zigz_status timed_end(zigz_ctx *ctx, int ev, double *us_out) {
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
zigz_status bind_launch(zigz_ctx *ctx, const uint32_t *d_in, size_t in_stride, uint32_t *d_out, size_t out_stride,
                               size_t half, size_t ncols, uint32_t r_m, const uint32_t *d_r_m, unsigned long long *d_sums,
                               const SumsLayout *lay) {
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
void bind_pool_reset(zigz_ctx *ctx) {
    ctx->pool_used = 0;
    ctx->pool_bytes = 0;
}
// call after the stream has been synchronised past the last recorded launch
zigz_status bind_pool_collect(zigz_ctx *ctx) {
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
zigz_status dev_eval_radix(zigz_ctx *ctx, const uint32_t *d_cols, size_t col_stride, size_t ncols, size_t nv,
                                  const uint64_t *points, uint32_t *d_vals, const EvalSkip *skip) {
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
zigz_status dev_eval_folds(zigz_ctx *ctx, const uint32_t *d_cols, size_t col_stride, size_t ncols, size_t nv,
                                  const uint64_t *points /*host, ncols*nv*/, uint32_t *d_vals, const EvalSkip *skip) {
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
zigz_status sumcheck_core(zigz_ctx *ctx, const uint32_t *d_in, size_t n, uint32_t *d_scratch,
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
    bool tail_pending = false;  // the last fold's finalize publishes the remaining table itself: read_tail only waits for it
    DoneFlag tail_done{};
};
// The sums a data pass has just produced in d_sums[0, n) -> `out`.  Sharded over RCCL they are first all-reduced in place, on
// the context's stream -- together with word n, the number of ranks whose local pass failed (st != OK: this rank adds 1) -- so
// the collective is issued on EVERY rank whatever happened locally, and all ranks learn of a failure in the same collective:
// the failing rank returns its own error, the others ZIGZ_ERR_COMM, and radix_run then skips the remaining passes on all of
// them alike (their collectives stay matched).  The wait behind the collective has the communicator's deadline: a peer that
// never enters it costs an abort and ZIGZ_ERR_COMM here, not a hang (RCCL has no timeout of its own).
// n words at d_src (u64, or u32 widened) -> out through pinned memory, written there by a kernel the host polls for: a copy
// command and a stream wait cost 15-25 us a time, a sumcheck of 2^24 entries is 30 us of data passes and three such hand-overs.
// rezero: the words are left zero (the sums of a later pass accumulate into them).  Many waiting threads (blocking sync): the
// sleeping wait, as in the commit path.
static zigz_status published(zigz_ctx *ctx, const DoneFlag &done, size_t n, bool u32, uint64_t *out);
static zigz_status publish_out(zigz_ctx *ctx, void *d_src, size_t n, bool u32, bool rezero, uint64_t *out) {
    if (n > PIN_WORDS / 4) return ZIGZ_ERR_INVALID_ARGUMENT;
    const DoneFlag done = done_flag(ctx, 2);
    if (u32) launch_publish_u32((const uint32_t *)d_src, n, (uint32_t *)ctx->h_pin, ctx->stream, done);
    else launch_publish_u64((unsigned long long *)d_src, n, (unsigned long long *)ctx->h_pin, rezero, ctx->stream, done);
    HIPCHK(ctx, hipGetLastError());
    return published(ctx, done, n, u32, out);
}
// waits for what a kernel has been asked to publish into ctx->h_pin under `done`, and hands it out
static zigz_status published(zigz_ctx *ctx, const DoneFlag &done, size_t n, bool u32, uint64_t *out) {
    const bool seen = !ctx->timing && (g_sleep_wait.load() ? sleep_wait(done.flag, done.seq) : spin_wait(done.flag, done.seq));
    if (!seen) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (u32) {
        const uint32_t *h = (const uint32_t *)ctx->h_pin;
        for (size_t i = 0; i < n; i++) out[i] = h[i];
    } else {
        memcpy(out, ctx->h_pin, n * 8);
    }
    return ZIGZ_OK;
}
zigz_status sums_out(GpuRadix *g, unsigned long long *d_sums, size_t n, zigz_status st, uint64_t *out) {
    zigz_ctx *ctx = g->ctx;
    if (!g->rccl) {
        CHK(st);
        return publish_out(ctx, d_sums, n, false, true, out);
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
        // (alone, both regions of d_sums were zeroed once when the sumcheck began and every read-back leaves what it read zero;
        // over RCCL the sums and the failure word behind them are filled before every pass, as the all-reduce wants them)
        if (g->rccl) HIPCHK(ctx, hipMemsetAsync(ctx->d_sums, 0, (nb + 1) * 8, ctx->stream));
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
    DoneFlag fin_done{};
    bool fin_published = false;
    auto local = [&]() -> zigz_status {
        uint32_t *wst = (uint32_t *)(ctx->h_pin + PIN_WORDS / 2);
        for (size_t b = 0; b < nb; b++) wst[b] = host_to_mont(weights[b]);
        HIPCHK(ctx, hipMemcpyAsync(g->wbuf, wst, nb * 4, hipMemcpyHostToDevice, ctx->stream));
        const size_t groups = radix_fold_groups(nb);
        launch_radix_fold(g->cur, 0, m, nb, (const uint32_t *)g->wbuf, 0, g->d_part, 0, 1, ctx->stream);
        HIPCHK(ctx, hipGetLastError());
        // alone, the finalize kernel's last workgroup hands the result to the host itself (the next stage's sums, or -- last
        // fold -- the remaining table): no read-back launch behind it
        FinalizePublish pub;
        if (!g->rccl && m <= PIN_WORDS / 4) {
            fin_done = done_flag(ctx, 2);
            pub.kind = k_next ? 1 : 2;
            pub.n = (unsigned)(k_next ? (size_t)1 << k_next : m);
            pub.h_dst = ctx->h_pin;
            pub.count = fin_done.count;
            pub.flag = fin_done.flag;
            pub.seq = fin_done.seq;
        }
        if (k_next) {
            if (g->rccl) HIPCHK(ctx, hipMemsetAsync(d_B2, 0, (((size_t)1 << k_next) + 1) * 8, ctx->stream));
            launch_radix_finalize(g->d_part, 0, groups, d_out, 0, m, log2_floor(m) - k_next, d_B2, 1, ctx->stream, nullptr, &pub);
        } else {
            launch_radix_finalize(g->d_part, 0, groups, d_out, 0, m, 0, nullptr, 1, ctx->stream, nullptr, &pub);
        }
        fin_published = pub.kind != 0;
        HIPCHK(ctx, hipGetLastError());
        return ZIGZ_OK;
    };
    const zigz_status st = local();
    if (k_next) {
        if (fin_published && st == ZIGZ_OK) CHK(published(ctx, fin_done, (size_t)1 << k_next, false, next_sums));
        else CHK(sums_out(g, d_B2, (size_t)1 << k_next, st, next_sums));
    } else {
        CHK(st);
        g->tail_pending = fin_published;
        g->tail_done = fin_done;
    }
    g->cur = d_out;
    g->len = m;
    g->stage++;
    return ZIGZ_OK;
}
zigz_status gpu_read_tail(void *user, size_t m, uint64_t *out) {
    GpuRadix *g = (GpuRadix *)user;
    zigz_ctx *ctx = g->ctx;
    if (m > PIN_WORDS) return ZIGZ_ERR_INVALID_ARGUMENT;
    if (g->tail_pending) {  // (the last fold's finalize kernel is publishing exactly these m words)
        g->tail_pending = false;
        return published(ctx, g->tail_done, m, true, out);
    }
    if (!g->rccl) return publish_out(ctx, (void *)g->cur, m, true, false, out);
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
    // both regions of the stage sums zero before the first pass (one fill per sumcheck; every read-back re-zeroes what it read)
    if (!rccl && n > HOST_TAIL_MAX) HIPCHK(ctx, hipMemsetAsync(ctx->d_sums, 0, (2048 + ((size_t)1 << RADIX_MAX_K) + 1) * 8, ctx->stream));
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
zigz_status stage_in(zigz_ctx *ctx, const uint64_t *in, size_t n, uint32_t **d_out) {
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

