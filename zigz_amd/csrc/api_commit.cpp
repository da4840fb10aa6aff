// C ABI of libzigz_hip.so, part 3: Merkle trees (dense and structure-aware builds), the commitment scheme and the commit jobs of
// Prover.generateCommitments (single and batched).
#include "api_internal.hpp"

using namespace zk;

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
    c.g_drops = c.g_skip = c.g_kept = 0;
    memset(c.r_last, 0, sizeof(c.r_last));
    memset(c.g_last, 0, sizeof(c.g_last));
    c.last_dropped = false;
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
zigz_status build_trees(zigz_ctx *ctx, const uint32_t *d_vals, size_t val_stride, size_t n_values, size_t npad,
                               uint8_t *d_slab, size_t ncols, bool record, TreeRef *ref) {
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
            static const bool always_probe = getenv("ZIGZ_CONS_PROBE_ALWAYS") != nullptr;  // (A/B)
            b.g_no_probe = ref && ctx->caps.npad == npad && ctx->caps.g_kept >= 2 && !ctx->cons_always && !always_probe;
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
            size_t expect = 0;  // what the last build of this shape held here, + 25 % + the sub-lists' slack (sizing only)
            if (ref && ctx->caps.npad == npad && (ctx->caps.r_last[l] || ctx->caps.g_last[l])) {
                auto more = [](unsigned v) { return (size_t)RUN_SUBS * (v + v / 4 + 8); };
                expect = (R.n ? more(ctx->caps.r_last[l]) : 0) + (G.n ? more(ctx->caps.g_last[l]) * G.n : 0) +
                         (G.n && ctx->caps.last_dropped ? (size_t)G.n * (npad >> l) : 0);
            }
            if (ctx->debug_skip != 1) launch_level_hash(b, l, ctx->stream, stamp(5, 0), expect);
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
zigz_status keccak_times_collect(zigz_ctx *ctx) {
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
    const bool arena = N >= RUN_MIN_LEAVES && N <= ((size_t)1 << BATCH_ARENA_MAX_NV) && ncols <= 64 && (ctx->run_aware_mask || ctx->cons_group_mask);
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
                c.r_last[l] = job->run_cols ? (unsigned)(ru ? ru : 1) : 0;
                c.g_last[l] = job->cons_hinted && !dropped ? (unsigned)(gu ? gu : 1) : 0;
                if (job->run_cols && (r_over ? ru > c.r[l] : ru * 10 > (unsigned long long)c.r[l] * 8))
                    c.r[l] = (unsigned)(ru + ru / 4 + 64);
                if (job->cons_hinted && !dropped && (g_over ? gu > c.g[l] : gu * 10 > (unsigned long long)c.g[l] * 8))
                    c.g[l] = (unsigned)(gu + gu / 4 + 64);
            }
            c.last_dropped = dropped;
            if (dropped) c.g_slabs = true;  // this context's traces do not repeat: give the group's columns slabs from now on
            if (job->cons_hinted) {
                c.g_drops = dropped ? c.g_drops + 1 : 0;
                c.g_kept = dropped ? 0 : c.g_kept + 1;
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

