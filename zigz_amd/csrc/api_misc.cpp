// C ABI of libzigz_hip.so, part 4: the simplified Lasso prover and the measurement hook (one hot kernel on synthetic tables).
#include "api_internal.hpp"

using namespace zk;

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

