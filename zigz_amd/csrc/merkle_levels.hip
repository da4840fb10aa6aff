// Structure-aware Merkle levels for gfx950 (kernels.hpp: MerkleBuild).  Which nodes of a level have to be hashed -- for the
// piecewise-constant columns (R) and the content-addressed group (G) -- depends on the VALUES only, so the lists of all
// levels are produced first (HBM / latency-bound passes, no hashing) and then hashed level by level, one launch each.
//
//   k_runs_stage      R: one workgroup = 4096 input nodes of one column -> change bitmap (ballots) -> need-bits of up to
//                     seven levels by bit arithmetic in one wave -> per-level bitmap / leader table / list entries
//   k_cons_leaf_insert, k_cons_pass, k_cons_decide
//                     G: per level a wave-deduplicated pass over a generation-tagged open-addressing table
//   k_level_hash      one launch per level: the R list, the G list x columns, or the group's columns densely when dropped
//   k_merkle_top      256 nodes per column -> root, one workgroup per column, no re-arm pauses (a dependent chain)
//   k_runs_fill_level, k_cons_fill_level   materialise the virtual nodes (tests, single trees)
//
// Reference loops: SimpleMerkleTree.build, src/commitments/merkle_tree.zig:283-318 (every digest equals the dense build's).
#include "kernels.hpp"
#include "tree_dev.hpp"

namespace zk {

#ifndef ZK_CONS_TPB
#define ZK_CONS_TPB 256
#endif
constexpr int CONS_TPB = ZK_CONS_TPB;  // threads per workgroup of the table passes (its waves share one LDS key set)

// ------------------------------------------------------------------ layout helpers (host)
static unsigned log2u(size_t n) { unsigned l = 0; while (((size_t)1 << l) < n) l++; return l; }

// stage s >= 1 reads, per hinted column, the first values (u32) and "not uniform" bits (u64 words) of the nodes of level 6 s
static size_t stage_fv_bytes(size_t npad, size_t ncols, unsigned s) { return ncols * (npad >> (s * RUN_STAGE_LEVELS)) * 4; }
static size_t stage_inner_bytes(size_t npad, size_t ncols, unsigned s) {
    const size_t n = npad >> (s * RUN_STAGE_LEVELS);
    return ncols * ((n + 63) / 64) * 8;
}
static size_t stage_off(size_t npad, size_t ncols, unsigned s) {  // byte offset of stage s's input arrays in r_stage
    size_t off = 0;
    for (unsigned t = 1; t < s; t++) off += stage_fv_bytes(npad, ncols, t) + stage_inner_bytes(npad, ncols, t);
    return off;
}
size_t runs_stage_scratch_bytes(size_t npad, size_t ncols) {
    const unsigned top = run_top_level(npad);
    size_t b = 64;
    for (unsigned s = 1; s * RUN_STAGE_LEVELS < top; s++) b += stage_fv_bytes(npad, ncols, s) + stage_inner_bytes(npad, ncols, s);
    return b;
}

static void stage_shape(size_t npad, unsigned l, size_t &seg, size_t &nseg, unsigned &rel) {  // the stage that emits level l
    const unsigned s = l == 0 ? 0 : (l - 1) / RUN_STAGE_LEVELS;
    const size_t n_in = npad >> (s * RUN_STAGE_LEVELS);
    seg = n_in < RUN_SEG ? n_in : RUN_SEG;
    nseg = n_in / seg;
    rel = l - s * RUN_STAGE_LEVELS;
}
LevelLists runs_lists(size_t npad, size_t ncols, const unsigned *cap_in) {
    LevelLists L{};
    L.top = run_top_level(npad);
    unsigned long long at = 0;
    for (unsigned l = 0; l <= L.top; l++) {
        size_t seg, nseg;
        unsigned rel;
        stage_shape(npad, l, seg, nseg, rel);
        const size_t units = ncols * nseg;  // (column, segment) pairs of the stage
        const size_t worst = (units + RUN_SUBS - 1) / RUN_SUBS * (seg >> rel);
        L.base[l] = at;
        L.cap[l] = (unsigned)(cap_in && cap_in[l] && cap_in[l] < worst ? cap_in[l] : worst);
        at += (unsigned long long)L.cap[l] * RUN_SUBS;
    }
    L.entries = at;
    return L;
}
size_t runs_units(size_t npad, size_t ncols, unsigned long long ubase_off[RUN_MAX_LEVELS]) {
    size_t at = 0;
    const unsigned top = run_top_level(npad);
    for (unsigned l = 0; l <= top; l++) {
        size_t seg, nseg;
        unsigned rel;
        stage_shape(npad, l, seg, nseg, rel);
        ubase_off[l] = at;
        at += ncols * nseg;
    }
    return at;
}
LevelLists cons_lists(size_t npad, const unsigned *cap_in) {
    LevelLists L{};
    L.top = run_top_level(npad);
    unsigned long long at = 0;
    for (unsigned l = 0; l <= L.top; l++) {
        const size_t worst = ((npad >> l) + RUN_SUBS - 1) / RUN_SUBS + 64;  // every node its own representative (+ slack: the
        L.base[l] = at;                                                     // sub-lists fill unevenly by at most a wave each)
        L.cap[l] = (unsigned)(cap_in && cap_in[l] && cap_in[l] < worst ? cap_in[l] : worst);
        at += (unsigned long long)L.cap[l] * RUN_SUBS;
    }
    L.entries = at;
    return L;
}

// ------------------------------------------------------------------ R: run-aware structure
constexpr unsigned long long BITS_ODD = 0xAAAAAAAAAAAAAAAAull;
// even bits of x -> its low 32 bits
__device__ __forceinline__ unsigned long long compress_even64(unsigned long long x) {
    x &= 0x5555555555555555ull;
    x = (x | (x >> 1)) & 0x3333333333333333ull;
    x = (x | (x >> 2)) & 0x0f0f0f0f0f0f0f0full;
    x = (x | (x >> 4)) & 0x00ff00ff00ff00ffull;
    x = (x | (x >> 8)) & 0x0000ffff0000ffffull;
    x = (x | (x >> 16)) & 0x00000000ffffffffull;
    return x;
}

// wave-wide inclusive scans on the DPP network (row shifts within the four rows of 16 lanes, then the two row broadcasts):
// six VALU instructions each instead of six LDS-crossbar round trips -- the scans are the serial part of a segment
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ int dpp_i32(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, BANK_MASK, false); }
__device__ __forceinline__ unsigned wave_scan_add(unsigned v) {
    int s = (int)v;
    s += dpp_i32<0x111, 0xf, 0xf>(0, s);  // row_shr:1
    s += dpp_i32<0x112, 0xf, 0xf>(0, s);  // row_shr:2
    s += dpp_i32<0x114, 0xf, 0xf>(0, s);  // row_shr:4
    s += dpp_i32<0x118, 0xf, 0xf>(0, s);  // row_shr:8
    s += dpp_i32<0x142, 0xa, 0xf>(0, s);  // row_bcast:15 into rows 1 and 3
    s += dpp_i32<0x143, 0xc, 0xf>(0, s);  // row_bcast:31 into rows 2 and 3
    return (unsigned)s;
}
__device__ __forceinline__ int wave_scan_max(int v) {  // values >= -1
    int s = v, t;
    t = dpp_i32<0x111, 0xf, 0xf>(-1, s); s = s > t ? s : t;
    t = dpp_i32<0x112, 0xf, 0xf>(-1, s); s = s > t ? s : t;
    t = dpp_i32<0x114, 0xf, 0xf>(-1, s); s = s > t ? s : t;
    t = dpp_i32<0x118, 0xf, 0xf>(-1, s); s = s > t ? s : t;
    t = dpp_i32<0x142, 0xa, 0xf>(-1, s); s = s > t ? s : t;
    t = dpp_i32<0x143, 0xc, 0xf>(-1, s); s = s > t ? s : t;
    return s;
}

// One WAVE = one segment of `seg` = 2^seg_log2 consecutive input nodes (leaves for stage 0, the nodes of level 6 s otherwise)
// of one hinted column; a workgroup is four neighbouring segments that share nothing (no workgroup barrier anywhere).  With
// x[i] the first value under input node i, c[i] = (x[i] != x[i-1]) and, per relative level r (node j covers the inputs
// j 2^r .. (j+1) 2^r - 1):
//     edge_r[j]  = c[j 2^r]                                   the node starts at a change
//     inner_r[j] = some change strictly inside the node       (or an input node that is itself not uniform)
//     need_r[j]  = inner_r[j] | inner_r[j-1] | edge_r[j]      not a copy of its left neighbour -> hashed
// and the first node of the segment's range is always hashed (need = 1), so a copy's leader lies in the same range.  The lane
// that owns input chunk q (64 nodes) gets its change word from ONE ballot over a coalesced row of the column; the words of
// level r + 1 follow from those of level r by OR-ing bit pairs and compressing the even bits (one lane per 64-bit word, two
// lanes' halves joined by a cross-lane read) -- a few hundred instructions for all levels, then the list entries.
constexpr int RUNS_WAVES = TPB / 64;
template <bool STAGE0>
__global__ __launch_bounds__(TPB) void k_runs_stage(MerkleBuild b, unsigned stage, unsigned seg_log2, unsigned rmax, unsigned nseg,
                                                    size_t in_off /*bytes, stage >= 1*/, size_t out_off /*bytes, next stage or ~0*/) {
    ZK_PRIO_SMALL();
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned y = blockIdx.y, segi = blockIdx.x * RUNS_WAVES + wave;
    if (segi >= nseg) return;  // (wave-uniform; the waves of a workgroup never meet)
    const unsigned seg = 1u << seg_log2, words = seg / 64;
    const unsigned l_in = stage * RUN_STAGE_LEVELS;
    const size_t n_in = b.npad >> l_in;
    const size_t first = (size_t)segi * seg;
    // 1. change words: lane q ends up with the word of input chunk q (E) and, above stage 0, its "not uniform" word (I)
    unsigned long long E = 0, I = 0;
    uint32_t fv_mine = 0;  // lane q: the first value of chunk q (what the next stage reads for relative level 6)
    {
        const uint32_t *src;
        size_t limit;  // inputs at or beyond it read as 0 (padding leaves, merkle_tree.zig:302-306)
        if (STAGE0) {
            src = b.vals + (size_t)b.rcols.c[y] * b.val_stride + first;
            limit = b.n_values > first ? b.n_values - first : 0;
        } else {
            src = reinterpret_cast<const uint32_t *>(b.r_stage + in_off) + (size_t)y * n_in + first;
            limit = seg;
            const unsigned long long *inner = reinterpret_cast<const unsigned long long *>(
                b.r_stage + in_off + (size_t)b.rcols.n * n_in * 4) + ((size_t)y * n_in + first) / 64;
            if (lane < words) I = inner[lane];
        }
        uint32_t prev_last = 0;  // x[64 q - 1] (wave-uniform)
        for (unsigned q0 = 0; q0 < words; q0 += 32) {
            uint32_t x[32];
#pragma unroll
            for (unsigned j = 0; j < 32; j++) {  // 32 coalesced rows (8 KiB per wave) in flight
                const unsigned i = (q0 + j) * 64 + lane;
                x[j] = (q0 + j < words && i < limit) ? src[i] : 0u;
            }
#pragma unroll
            for (unsigned j = 0; j < 32; j++) {
                const unsigned q = q0 + j;
                uint32_t left = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x[j], 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
                if (lane == 0) left = prev_last;
                const bool c = (q == 0 && lane == 0) || x[j] != left;
                const unsigned long long m = __ballot(c);
                const uint32_t x_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)x[j]);
                if (q < words) {
                    if (lane == q) { E = m; fv_mine = x_first; }
                    prev_last = (uint32_t)__builtin_amdgcn_readlane((int)x[j], 63);
                }
            }
        }
    }
    if (STAGE0) {  // is the column still constant?  Not if this segment holds a change (beyond the forced bit of its first leaf) or
        // starts with another value than the column does; no wave says so <=> all N leaves are equal (EvalSkip, kernels.hpp)
        const unsigned long long inside = lane < words ? (lane == 0 ? E & ~1ull : E) : 0;
        const uint32_t col_first = b.n_values ? b.vals[(size_t)b.rcols.c[y] * b.val_stride] : 0u;
        const uint32_t seg_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)fv_mine);  // (lane 0 holds chunk 0's first value)
        if (__ballot(inside != 0) || seg_first != col_first)
            if (lane == 0) b.r_ctr[RUN_CHANGED + y] = 1;
    }
    // 2. all levels of the segment: lane j < words >> r holds word j of relative level r
    unsigned w = words;
    unsigned long long need_r[RUN_STAGE_LEVELS + 1];
    unsigned off_r[RUN_STAGE_LEVELS + 1];
    unsigned my_tot = 0;
#pragma unroll
    for (unsigned r = 0; r <= RUN_STAGE_LEVELS; r++) {
        need_r[r] = 0;
        off_r[r] = 0;
        if (r > rmax) continue;
        if (r) {
            const unsigned long long t = I | (E & BITS_ODD);
            const unsigned long long i32 = compress_even64(t | (t >> 1)), e32 = compress_even64(E);
            const unsigned long long ilo = __shfl(i32, (2 * lane) & 63), ihi = __shfl(i32, (2 * lane + 1) & 63);
            const unsigned long long elo = __shfl(e32, (2 * lane) & 63), ehi = __shfl(e32, (2 * lane + 1) & 63);
            w >>= 1;
            I = lane < w ? (ilo | (ihi << 32)) : 0;
            E = lane < w ? (elo | (ehi << 32)) : 0;
        }
        if (!STAGE0 && r == 0) continue;  // the input level was emitted by the previous stage
        unsigned carry = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(I >> 63), 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        if (lane == 0) carry = 1;  // the first node of the segment's range is always hashed
        const unsigned long long need = lane < w ? (I | (I << 1) | carry | E) : 0;
        const unsigned cnt = (unsigned)__builtin_popcountll(need);
        const unsigned incl = wave_scan_add(cnt);
        // last hashed node up to the end of this word -> the word before gives "the last hashed node before the chunk"
        const int last = wave_scan_max(need ? (int)(lane * 64 + 63 - __builtin_clzll(need)) : -1);
        int prev = __builtin_amdgcn_update_dpp(0, last, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        if (lane == 0) prev = 0;
        const unsigned tot = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);  // (lanes >= w add nothing)
        if (lane == r) my_tot = tot;
        const unsigned l_abs = l_in + r;
        if (lane < w) {
            const size_t e = run_meta_base(b.npad, b.t.ncols, l_abs) + ((size_t)y * (b.npad >> l_abs) + (first >> r)) / 64 + lane;
            b.t.bitmap[e] = need;
            b.t.prev[e] = (unsigned short)prev;
            b.t.woff[e] = (unsigned short)(incl - cnt);
        }
        need_r[r] = need;
        off_r[r] = incl - cnt;
    }
    // what the next stage reads: first value and "not uniform" bit of the 64 nodes of relative level 6
    if (out_off != ~(size_t)0) {  // (then seg = 4096 and rmax = 6: I is one word, in lane 0)
        const size_t n_out = n_in >> RUN_STAGE_LEVELS;
        uint32_t *fv = reinterpret_cast<uint32_t *>(b.r_stage + out_off) + (size_t)y * n_out + (size_t)segi * 64;
        unsigned long long *inner = reinterpret_cast<unsigned long long *>(b.r_stage + out_off + (size_t)b.rcols.n * n_out * 4) +
                                    ((size_t)y * n_out) / 64 + segi;
        fv[lane] = fv_mine;
        if (lane == 0) *inner = I;
    }
    // one list reservation per level, each by a lane of its own (the sub-list counters sit in lines of their own).  The unit's
    // first slot goes to t.ubase: with the per-chunk offsets above it is how a reader finds a node's digest in list order.
    unsigned my_slot = 0, my_room = 0;  // lane r: first list slot of this unit at relative level r, and how many entries fit
    if (lane <= rmax && (STAGE0 || lane != 0)) {
        const unsigned l_abs = l_in + lane;
        const unsigned sub = (y * nseg + segi) % RUN_SUBS;
        const unsigned cap = b.t.r_lists.cap[l_abs];
        const unsigned long long pos = atomicAdd(&b.r_ctr[run_ctr_index(l_abs, sub)], (unsigned long long)my_tot);
        if (pos + my_tot > cap) atomicOr(&b.r_ctr[10], 1ull);  // out of room: zigz_commit_roots repeats the build with more
        my_room = pos >= cap ? 0u : (unsigned)(cap - pos < my_tot ? cap - pos : my_tot);
        my_slot = sub * cap + (unsigned)(pos < cap ? pos : cap);
        b.t.ubase[b.t.ubase_off[l_abs] + (size_t)y * nseg + segi] = my_slot;
    }
    // 3. the list entries of every level: word by word (wave-uniform), lane j <-> node j of the word, so a word's entries go out
    // as one coalesced store; words without a hashed node are skipped
#pragma unroll
    for (unsigned r = 0; r <= RUN_STAGE_LEVELS; r++) {
        if (r > rmax || (!STAGE0 && r == 0)) continue;
        const unsigned slot0 = __shfl(my_slot, r), room = __shfl(my_room, r);
        uint32_t *list = b.r_list + b.t.r_lists.base[l_in + r];
        const size_t node0 = first >> r;
        const unsigned wr = words >> r;
        for (unsigned q = 0; q < wr; q++) {
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)need_r[r], q);
            const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(need_r[r] >> 32), q);
            if ((lo | hi) == 0) continue;
            const unsigned o = (unsigned)__builtin_amdgcn_readlane((int)off_r[r], q);
            const unsigned long long m = ((unsigned long long)hi << 32) | lo;
            const unsigned at = o + __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0));
            if (((m >> lane) & 1) && at < room) list[slot0 + at] = ((uint32_t)y << RUN_NODE_BITS) | (uint32_t)(node0 + q * 64 + lane);
        }
    }
}

void launch_runs_structure(const MerkleBuild &b, hipStream_t s, const KTime *kt) {
    if (b.rcols.n == 0) return;
    const unsigned top = b.t.r_lists.top;
    unsigned nstages = 1;
    while (nstages * RUN_STAGE_LEVELS < top) nstages++;
    for (unsigned st = 0; st < nstages; st++) {
        const size_t n_in = b.npad >> (st * RUN_STAGE_LEVELS);
        const size_t seg = n_in < RUN_SEG ? n_in : RUN_SEG;
        const unsigned rmax = top - st * RUN_STAGE_LEVELS < RUN_STAGE_LEVELS ? top - st * RUN_STAGE_LEVELS : RUN_STAGE_LEVELS;
        const bool has_next = st + 1 < nstages;
        const size_t in_off = st ? stage_off(b.npad, b.rcols.n, st) : 0;
        const size_t out_off = has_next ? stage_off(b.npad, b.rcols.n, st + 1) : ~(size_t)0;
        const unsigned nseg = (unsigned)(n_in / seg);
        const dim3 grid((nseg + RUNS_WAVES - 1) / RUNS_WAVES, b.rcols.n, b.t.nz ? b.t.nz : 1);
        hipEvent_t e0 = kt && st == 0 ? kt->start : nullptr, e1 = kt && st + 1 == nstages ? kt->stop : nullptr;
        if (st == 0) {
            if (e0 || e1) hipExtLaunchKernelGGL(k_runs_stage<true>, grid, dim3(TPB), 0, s, e0, e1, 0, b, st, log2u(seg), rmax, nseg, in_off, out_off);
            else hipLaunchKernelGGL(k_runs_stage<true>, grid, dim3(TPB), 0, s, b, st, log2u(seg), rmax, nseg, in_off, out_off);
        } else {
            if (e0 || e1) hipExtLaunchKernelGGL(k_runs_stage<false>, grid, dim3(TPB), 0, s, e0, e1, 0, b, st, log2u(seg), rmax, nseg, in_off, out_off);
            else hipLaunchKernelGGL(k_runs_stage<false>, grid, dim3(TPB), 0, s, b, st, log2u(seg), rmax, nseg, in_off, out_off);
        }
    }
}

// ------------------------------------------------------------------ G: content-addressed structure
// One open-addressing table of 2 npad slots serves every level.  A slot's word is generation << 52 | payload -- the
// generation is the level's (b.g_gen + level), so entries of earlier levels and earlier builds read as free and the table
// is never cleared -- payload = 52 bits of the leaf tuple's fingerprint at level 0 (verified against the representative's
// tuple when it is read back) and the pair of the children's representatives above, which IS the identity of the hash input.
// Whoever inserts a key first makes its node the representative of the nodes with that key: it takes the next LIST SLOT of
// the level (idx[table slot] = list slot, list[list slot] = node), and a node's representative is known by that slot from
// then on -- the slot is where its digests are stored (TreeRef::g_store), and the pair of the children's slots is the key of
// the next level.
constexpr unsigned CONS_GEN_SHIFT = 52;
__device__ __forceinline__ unsigned long long cons_mix(unsigned long long x) {  // splitmix64 finaliser
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}
constexpr unsigned CONS_TUP = 16;  // a leaf's tuple is kept in registers up to this many columns (the witness group has ten)
__device__ __forceinline__ unsigned long long cons_leaf_payload(const MerkleBuild &b, size_t k, uint32_t *tup = nullptr) {
    unsigned long long h = 0x243f6a8885a308d3ull;
    if (tup && b.gcols.n <= CONS_TUP) {
#pragma unroll
        for (unsigned j = 0; j < CONS_TUP; j++) {
            tup[j] = (j < b.gcols.n && k < b.n_values) ? b.vals[(size_t)b.gcols.c[j] * b.val_stride + k] : 0u;  // padding leaves hold 0
            if (j < b.gcols.n) h = cons_mix(h ^ tup[j]);
        }
    } else {
        for (unsigned j = 0; j < b.gcols.n; j++)
            h = cons_mix(h ^ (k < b.n_values ? b.vals[(size_t)b.gcols.c[j] * b.val_stride + k] : 0u));
    }
    return h & ((1ull << CONS_GEN_SHIFT) - 1);
}
// Lanes of a wave that hold the same key elect the lowest of them: in a loop-dominated trace the 64 nodes of a wave share a
// handful of keys, and one table access per distinct key instead of one per lane is what keeps a million nodes from
// hammering four words.  Returns the elected lane (the lane itself when it is elected or does not take part).
__device__ __forceinline__ unsigned wave_leader(unsigned long long key, bool valid, unsigned lane) {
    unsigned long long pend = __ballot(valid);
    unsigned mine = lane;
    while (pend) {  // one round per distinct key of the wave (<= 64)
        const unsigned ld = (unsigned)__builtin_ctzll(pend);
        const unsigned long long kk = __shfl(key, ld);
        const bool same = valid && key == kk;
        if (same) mine = ld;
        pend &= ~__ballot(same);
    }
    return mine;
}
// the next list slots of level l for the lanes of this wave that ask for one (one atomic per wave; the wave's sub-list is
// fixed by its position in the level, so a sub-list receives at most the nodes of its own waves).  Out of room: the slot is
// clamped to the sub-list's last one (nothing out of bounds is ever written) and the build is flagged.
__device__ __forceinline__ unsigned cons_take_slot(const MerkleBuild &b, unsigned l, bool want, unsigned lane, size_t wave_global) {
    const unsigned long long m = __ballot(want);
    if (!m) return 0;
    const unsigned sub = (unsigned)(wave_global % RUN_SUBS);
    const unsigned cap = b.t.g_lists.cap[l];
    unsigned long long base = 0;
    if (lane == (unsigned)__builtin_ctzll(m)) base = atomicAdd(&b.g_ctr[run_ctr_index(l, sub)], (unsigned long long)__builtin_popcountll(m));
    base = __shfl(base, (unsigned)__builtin_ctzll(m));
    unsigned long long pos = base + (unsigned)__builtin_popcountll(m & ((1ull << lane) - 1));
    if (want && pos >= cap) {
        atomicOr(&b.g_ctr[10], 1ull);
        pos = cap - 1;
    }
    return sub * cap + (unsigned)pos;
}
// Inserts `key`; returns true when THIS thread put it there (its node becomes the representative and must publish its list
// slot with cons_publish).
__device__ __forceinline__ bool cons_insert(unsigned long long *keys, size_t mask, unsigned long long key, unsigned g_cur, unsigned g_prev,
                                            size_t *where) {
    size_t slot = cons_mix(key) & mask;
    for (size_t tries = 0; tries <= mask; tries++) {  // live keys fill at most 3/4 of the slots: a free or matching one is reached
        // look through the caches first: a key, once in its slot, stays for the rest of its generation, so a (possibly stale)
        // cached copy that shows it is proof enough -- and thousands of waves looking at the same few slots are served by their
        // CU's L1 instead of queueing at one memory channel (same-address atomic loads serialise at ~15 ns each: 125 us for
        // the 8192 waves of level 1 of a 4-step loop)
        if (keys[slot] == key) return false;
        unsigned long long cur = __atomic_load_n(&keys[slot], __ATOMIC_RELAXED);
        for (;;) {  // until this slot holds a live key
            if (cur == key) return false;
            const unsigned g = (unsigned)(cur >> CONS_GEN_SHIFT);
            if (g == g_cur || g == g_prev) break;  // somebody else's live key: next slot
            const unsigned long long old = atomicCAS(&keys[slot], cur, key);
            if (old == cur) {
                *where = slot;
                return true;
            }
            cur = old;
        }
        slot = (slot + 1) & mask;
    }
    return false;
}
__device__ __forceinline__ uint32_t cons_lookup(const unsigned long long *keys, const uint32_t *idx, size_t mask, unsigned long long key) {
    size_t slot = cons_mix(key) & mask;
    for (size_t tries = 0; tries <= mask; tries++) {
        if (keys[slot] == key) return idx[slot];
        slot = (slot + 1) & mask;
    }
    return 0xffffffffu;  // cannot happen: the key was inserted by the previous launch
}
// Second filter, per workgroup: of the wave leaders that hold the same key only the first to put it into the workgroup's key
// set (LDS) goes to the table in memory.  A loop-dominated level then costs a handful of table accesses per 1024 nodes; without
// it every wave of the level queues at the same few words (same-address atomics serialise at ~15 ns: 85 us for level 1).
constexpr unsigned CONS_SET = 2 * CONS_TPB;  // slots: at most CONS_TPB keys arrive
__device__ __forceinline__ bool wg_first(unsigned long long *set, unsigned long long key) {
    unsigned slot = (unsigned)(cons_mix(key) >> 40) & (CONS_SET - 1);
    for (;;) {
        const unsigned long long old = atomicCAS(&set[slot], 0ull, key);  // (keys are never 0: the generation is >= 1)
        if (old == 0) return true;
        if (old == key) return false;
        slot = (slot + 1) & (CONS_SET - 1);
    }
}
// the common tail of both passes: the elected lanes insert their key; the ones that put it there take a list slot for their
// node and publish it
__device__ __forceinline__ void cons_insert_level(const MerkleBuild &b, unsigned long long *s_set, unsigned l, unsigned long long key,
                                                  bool elected, uint32_t node, unsigned lane, size_t wave_global) {
    const size_t mask = 2 * b.npad - 1;
    size_t where = 0;
    const unsigned g = b.g_gen + l;
    const bool won = elected && wg_first(s_set, key) && cons_insert(b.g_keys, mask, key, g, l ? g - 1 : g, &where);
    const unsigned slot = cons_take_slot(b, l, won, lane, wave_global);
    if (won) {
        b.g_list[b.t.g_lists.base[l] + slot] = node;
        b.g_idx[where] = slot;
    }
}

// sample != 0: the PROBE -- only every CONS_SAMPLE-th chunk of 64 leaves takes part (spread over the whole trace: a loop of P
// steps shows all its P tuples in any such sample, a program that never repeats shows nothing but distinct ones), so that a
// group that does not repeat is found out and dropped (k_cons_decide) for a sixteenth of the price of inserting every leaf --
// 12 ms for 2^20 all-distinct leaves, whose waves go through the election loop 64 times -- and the full pass (sample == 0)
// returns at once when the probe has dropped the group.  Keys the probe inserted are found again by the full pass: they keep
// the list slots they took.
constexpr unsigned CONS_SAMPLE = 16;
__global__ __launch_bounds__(CONS_TPB) void k_cons_leaf_insert(MerkleBuild b, int sample) {
    ZK_PRIO_SMALL();
    __shared__ unsigned long long s_set[CONS_SET];
    if (!sample && b.g_ctr[8]) return;  // dropped by the probe
    for (unsigned i = threadIdx.x; i < CONS_SET; i += CONS_TPB) s_set[i] = 0;
    __syncthreads();
    const size_t k = (size_t)blockIdx.x * CONS_TPB + threadIdx.x;
    if (sample && ((k / 64) % CONS_SAMPLE) != 0) return;  // (wave-uniform; no barrier below)
    const unsigned lane = threadIdx.x & 63;
    const bool valid = k < b.npad;
    const unsigned long long key = ((unsigned long long)b.g_gen << CONS_GEN_SHIFT) | (valid ? cons_leaf_payload(b, k) : 0);
    const unsigned ld = wave_leader(key, valid, lane);
    cons_insert_level(b, s_set, 0, key, valid && ld == lane, (uint32_t)k, lane, k / 64);
}

// Resolves level lr (every node learns the list slot of its representative) and, if do_insert, inserts the keys of level
// lr + 1: node c / 2's key is the pair of the slots of c and c + 1, which sit in neighbouring lanes.
template <bool LEAF>
__global__ __launch_bounds__(CONS_TPB) void k_cons_pass(MerkleBuild b, unsigned lr, int do_insert) {
    ZK_PRIO_SMALL();
    __shared__ unsigned long long s_set[CONS_SET];
    if (b.g_ctr[8]) return;  // the group was dropped (k_cons_decide)
    for (unsigned i = threadIdx.x; i < CONS_SET; i += CONS_TPB) s_set[i] = 0;
    __syncthreads();
    const size_t n = b.npad >> lr, c = (size_t)blockIdx.x * CONS_TPB + threadIdx.x;
    const unsigned lane = threadIdx.x & 63;
    const bool valid = c < n;
    const size_t mask = 2 * b.npad - 1;
    const unsigned g_r = b.g_gen + lr;
    const size_t off = 2 * b.npad - 2 * n;                            // first node of level lr in g_rep
    const size_t off_in = LEAF ? 0 : 2 * b.npad - 2 * (b.npad >> (lr - 1));  // ... of level lr - 1
    unsigned long long key = (unsigned long long)g_r << CONS_GEN_SHIFT;
    uint32_t tup[CONS_TUP];  // (leaves) this leaf's tuple: compared with the representative's below without reading it again
    if (valid) {
        if (LEAF) key |= cons_leaf_payload(b, c, tup);
        else {
            const uint2 p = *reinterpret_cast<const uint2 *>(b.g_rep + off_in + 2 * c);
            key |= ((unsigned long long)p.x << RUN_NODE_BITS) | p.y;
        }
    }
    const unsigned ld = wave_leader(key, valid, lane);
    uint32_t r = 0xffffffffu;
    if (valid && ld == lane) r = cons_lookup(b.g_keys, b.g_idx, mask, key);
    r = __shfl(r, ld);
    bool alone = false;  // (leaves only) equal fingerprints are not yet equal tuples: verify against the representative's tuple,
    if (LEAF && valid && r != 0xffffffffu) {  // else this leaf stands for itself, in a list slot of its own
        const size_t rn = b.g_list[b.t.g_lists.base[0] + r];
        if (rn != c) {
            if (b.gcols.n <= CONS_TUP) {
#pragma unroll
                for (unsigned j = 0; j < CONS_TUP; j++)
                    if (j < b.gcols.n && tup[j] != (rn < b.n_values ? b.vals[(size_t)b.gcols.c[j] * b.val_stride + rn] : 0u)) alone = true;
            } else {
                for (unsigned j = 0; j < b.gcols.n; j++) {
                    const uint32_t *v = b.vals + (size_t)b.gcols.c[j] * b.val_stride;
                    if ((c < b.n_values ? v[c] : 0u) != (rn < b.n_values ? v[rn] : 0u)) { alone = true; break; }
                }
            }
        }
    }
    if (valid && r == 0xffffffffu) alone = true;  // (cannot happen)
    const unsigned own = cons_take_slot(b, lr, alone, lane, c / 64);
    if (alone) {
        r = own;
        b.g_list[b.t.g_lists.base[lr] + own] = (uint32_t)c;
    }
    if (valid) b.g_rep[off + c] = r;
    if (do_insert) {
        const uint32_t r_hi = __shfl_down(r, 1);
        const bool mine = valid && !(c & 1);
        const unsigned long long k2 = ((unsigned long long)(g_r + 1) << CONS_GEN_SHIFT) | ((unsigned long long)r << RUN_NODE_BITS) | r_hi;
        const unsigned ld2 = wave_leader(k2, mine, lane);
        cons_insert_level(b, s_set, lr + 1, k2, mine && ld2 == lane, (uint32_t)(c >> 1), lane, c / 128);
    }
}

// keep or drop: a group whose leaves are mostly distinct does not repeat, and its table passes would find nothing.  After the
// probe (sample != 0) the count is of the sampled leaves; after the full pass of all.
__global__ __launch_bounds__(64) void k_cons_decide(MerkleBuild b, int sample) {
    ZK_PRIO_SMALL();
    unsigned long long c = threadIdx.x < RUN_SUBS ? b.g_ctr[run_ctr_index(0, threadIdx.x)] : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if (threadIdx.x == 0) {
        if (sample) {  // the probe: a sixteenth of the leaves
            const bool drop = c > b.npad / CONS_SAMPLE / 4;
            b.g_ctr[8] = drop ? 1 : 0;
            b.g_ctr[9] = drop ? c * CONS_SAMPLE : 0;  // (distinct leaves: an estimate when the probe drops the group)
        } else if (!b.g_ctr[8]) {  // (a group the probe dropped stays dropped: the full pass did not run)
            b.g_ctr[9] = c;
            b.g_ctr[8] = c > b.npad / 4 ? 1 : 0;
        }
        if (!sample && b.g_ctr[8] && !b.g_has_slabs) atomicOr(&b.g_ctr[10], 2ull);  // nowhere to build the columns densely: repeated
    }
}

void launch_cons_structure(const MerkleBuild &b, hipStream_t s, const KTime *kt) {
    if (b.gcols.n == 0) return;
    const unsigned top = b.t.g_lists.top;
    const unsigned nz = b.t.nz ? b.t.nz : 1;
    const dim3 g0((unsigned)((b.npad + CONS_TPB - 1) / CONS_TPB), 1, nz);
    // the probe first (trees of >= 2^16 leaves: below that the whole insert costs less than a launch), then the full pass
    // -- unless the context's recent builds all kept their group (a service's slots see one kind of trace after the other): the
    // probe is then two launches that find out what is known; a trace that does not repeat after all is dropped by the full pass,
    // the expensive way, once
    const bool probe = b.npad >= ((size_t)1 << 16) && !b.g_no_probe;
    if (probe) {
        if (kt) hipExtLaunchKernelGGL(k_cons_leaf_insert, g0, dim3(CONS_TPB), 0, s, kt->start, nullptr, 0, b, 1);
        else hipLaunchKernelGGL(k_cons_leaf_insert, g0, dim3(CONS_TPB), 0, s, b, 1);
        hipLaunchKernelGGL(k_cons_decide, dim3(1, 1, nz), dim3(64), 0, s, b, 1);
    }
    if (kt && !probe) hipExtLaunchKernelGGL(k_cons_leaf_insert, g0, dim3(CONS_TPB), 0, s, kt->start, nullptr, 0, b, 0);
    else hipLaunchKernelGGL(k_cons_leaf_insert, g0, dim3(CONS_TPB), 0, s, b, 0);
    hipLaunchKernelGGL(k_cons_pass<true>, g0, dim3(CONS_TPB), 0, s, b, 0u, top >= 1 ? 1 : 0);
    hipLaunchKernelGGL(k_cons_decide, dim3(1, 1, nz), dim3(64), 0, s, b, 0);
    for (unsigned lr = 1; lr <= top; lr++) {
        const dim3 g((unsigned)(((b.npad >> lr) + CONS_TPB - 1) / CONS_TPB), 1, nz);
        const int ins = lr < top ? 1 : 0;
        if (kt && lr == top) hipExtLaunchKernelGGL(k_cons_pass<false>, g, dim3(CONS_TPB), 0, s, nullptr, kt->stop, 0, b, lr, ins);
        else hipLaunchKernelGGL(k_cons_pass<false>, g, dim3(CONS_TPB), 0, s, b, lr, ins);
    }
}

// ------------------------------------------------------------------ hashing a level from its lists
// Level L of the R and G columns in ONE launch: [the R list][the G list x the group's columns][when the group was dropped:
// every node of the columns in gdense].  A fixed grid strides over that index space, one hash per thread and step, every
// lane busy whatever mix of constant and busy columns produced the lists.  A list entry's digest is stored at ITS LIST SLOT;
// children are read where their digests are: an R child at its leader's slot, a G child at its representative's.
// PAUSE = false for the small levels, whose few waves run alone on their SIMDs (keccak.hpp).
#ifndef ZK_LEVEL_HASH_MIN_WAVES
#define ZK_LEVEL_HASH_MIN_WAVES 1  // (A/B: waves per SIMD the compiler must leave room for)
#endif
template <bool LEAF, bool PAUSE>
__global__ __launch_bounds__(TPB, ZK_LEVEL_HASH_MIN_WAVES) void k_level_hash(MerkleBuild b, unsigned L, ColMap gdense, int prio) {
    __shared__ unsigned long long s_r[RUN_SUBS + 1], s_g[RUN_SUBS + 1];  // exclusive prefixes of the sub-list lengths
#ifdef ZK_LEVEL_HASH_LDS_PAD  // (A/B: unused LDS per workgroup caps the workgroups per CU)
    __shared__ unsigned s_pad[ZK_LEVEL_HASH_LDS_PAD / 4];
    if (b.npad == 3) s_pad[threadIdx.x] = L;  // (never true: keeps the array)
#endif
    // A level that cannot fill the chip is a link in its proof's chain of dependent launches: its waves go first on the SIMDs they
    // share with other proofs' big levels (tools/prio_probe.hip: a lone hashing wave among six others takes 80-250 us for a
    // permutation at priority 0, 10-15 us -- what it takes alone -- at priority 3)
    if (prio) ZK_PRIO_SMALL();
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool dropped = b.gcols.n != 0 && b.g_ctr[8] != 0;
    if (wave < 2) {
        const bool use = wave == 0 ? b.rcols.n != 0 : (b.gcols.n != 0 && !dropped);
        const unsigned long long *ctr = wave == 0 ? b.r_ctr : b.g_ctr;
        const unsigned cap = wave == 0 ? b.t.r_lists.cap[L] : b.t.g_lists.cap[L];
        unsigned long long c = use && lane < RUN_SUBS ? ctr[run_ctr_index(L, lane)] : 0;
        if (c > cap) c = cap;  // (a list that ran out of room: the build is flagged and will be repeated)
        unsigned long long incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long a = __shfl_up(incl, off);
            if (lane >= (unsigned)off) incl += a;
        }
        unsigned long long *dst = wave == 0 ? s_r : s_g;
        if (lane < RUN_SUBS) dst[lane] = incl - c;
        if (lane == RUN_SUBS - 1) dst[RUN_SUBS] = incl;
    }
    __syncthreads();
    // Everything that depends on the level only -- list lengths (out of LDS, i.e. in vector registers), capacities and bases
    // (indexed by L in the kernel arguments), the R metadata of the children's level -- as wave-uniform scalars, fetched once:
    // what the loop keeps in vector registers through a permutation is its index and where the result goes, nothing else.
    // No 64-bit division inside the loop either (its magic numbers are loop invariants in vector registers too).
    const size_t n_L = uniform64(b.npad >> L);
    const unsigned n_L_log2 = (unsigned)__builtin_ctzll(n_L);  // (a power of two)
    const unsigned gn = b.gcols.n;
    const size_t cR = uniform64(s_r[RUN_SUBS]), cG = uniform64(s_g[RUN_SUBS]) * gn,
                 cD = (dropped && b.g_has_slabs) ? (size_t)gdense.n * n_L : 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // nodes hashed by the whole build
        if (cR) atomicAdd(&b.r_ctr[0], (unsigned long long)cR);
        if (cG) atomicAdd(&b.g_ctr[0], (unsigned long long)cG);
    }
    const size_t total = cR + cG + cD;
    const RLevel rl = r_level(b.t, L, true), rc = r_level(b.t, LEAF ? 0 : L - 1, true);
    const size_t r_cap = uniform64(b.t.r_lists.cap[L]), g_cap = uniform64(b.t.g_lists.cap[L]);
    const size_t g_base = uniform64(b.t.g_lists.base[L]), g_base_c = uniform64(b.t.g_lists.base[LEAF ? 0 : L - 1]);
    const size_t g_rep_c = slab_level_offset(b.npad, LEAF ? 0 : L - 1);
    // floor(2^32 / group size); a group of ONE column: 2^32 does not fit, and 2^32 - 1 gives "the quotient or one short of it" too
    const unsigned g_magic = gn > 1 ? (unsigned)(0x100000000ull / gn) : 0xffffffffu;
#pragma unroll 1
    for (size_t e = (size_t)blockIdx.x * TPB + threadIdx.x; e < total; e += (size_t)gridDim.x * TPB) {
        size_t col, k;
        uint8_t *out;
        const uint8_t *in0 = nullptr, *in1 = nullptr;
#ifdef ZK_LH_FAKE  // measurement only (wrong trees): no list, no leader search, no gathers -- what the loop costs without them
        if (ZK_LH_FAKE) {
            col = 0;
            k = e;
            out = b.t.upper + (e & 4095) * 32;
            in0 = b.t.upper + ((2 * e) & 4095) * 32;
            in1 = in0 + 32;
        } else
#endif
        if (e < cR) {
            unsigned sub = 0;  // the sub-list that holds entry e: the last one that starts at or before it
#pragma unroll
            for (unsigned step = RUN_SUBS / 2; step; step >>= 1)
                if (s_r[sub + step] <= e) sub += step;
            const size_t slot = (size_t)sub * r_cap + (e - s_r[sub]);
            const uint32_t ent = b.r_list[rl.list_base + slot];
            const unsigned y = ent >> RUN_NODE_BITS;
            col = b.rcols.c[y];
            k = ent & ((1u << RUN_NODE_BITS) - 1);
            out = r_slot_ptr(b.t, rl, slot);
            if (!LEAF) {
                size_t c0, c1;
                r_slot_pair(b.t, rc, y, 2 * k, &c0, &c1);
                in0 = r_slot_ptr(b.t, rc, c0);
                in1 = r_slot_ptr(b.t, rc, c1);
            }
        } else if (e < cR + cG) {
            // (entry, column) = divmod(t, group size): t < 2^32 (launch_level_hash), so the multiply-high with
            // floor(2^32 / n) is the quotient or one short of it
            const unsigned t32 = (unsigned)(e - cR);
            unsigned ei = __umulhi(t32, g_magic), j = t32 - ei * gn;
            if (j >= gn) {
                ei++;
                j -= gn;
            }
            col = b.gcols.c[j];
            unsigned sub = 0;
#pragma unroll
            for (unsigned step = RUN_SUBS / 2; step; step >>= 1)
                if (s_g[sub + step] <= ei) sub += step;
            const size_t slot = (size_t)sub * g_cap + (ei - s_g[sub]);
            k = b.g_list[g_base + slot];
            out = b.t.g_store + ((g_base + slot) * gn + j) * 32;
            if (!LEAF) {
                const uint2 p = *reinterpret_cast<const uint2 *>(b.g_rep + g_rep_c + 2 * k);
                in0 = b.t.g_store + ((g_base_c + p.x) * gn + j) * 32;
                in1 = b.t.g_store + ((g_base_c + p.y) * gn + j) * 32;
            }
        } else {
            const size_t t = e - cR - cG;
            col = gdense.c[t >> n_L_log2];
            k = t & (n_L - 1);
            out = slab_ptr(b.t, col, L, k);
            if (!LEAF) {
                in0 = slab_ptr(b.t, col, L - 1, 2 * k);
                in1 = in0 + 32;
            }
        }
        Digest d;
        if (LEAF) d = sha3_leaf<PAUSE>((uint64_t)(k < b.n_values ? b.vals[col * b.val_stride + k] : 0u));
        else d = sha3_node<PAUSE>(load_digest_at(in0), load_digest_at(in1));
        store_digest_at(out, d);
    }
}

void launch_level_hash(const MerkleBuild &b, unsigned L, hipStream_t s, const KTime *kt, size_t expect) {
    if (b.rcols.n == 0 && b.gcols.n == 0) return;
    // when the group was dropped its columns are hashed densely here -- except the small-domain members on the levels 0 and 1,
    // which come from the tables (launch_keccak_small_l01, guarded by the same flag)
    ColMap gd{};
    for (unsigned j = 0; j < b.gcols.n; j++) {
        bool sd = false;
        for (unsigned i = 0; i < b.gcols_sd.n; i++) sd |= b.gcols_sd.c[i] == b.gcols.c[j];
        if (L >= 2 || !sd) gd.c[gd.n++] = b.gcols.c[j];
    }
    const size_t n_L = b.npad >> L;
    size_t wgs = ((size_t)(b.rcols.n + b.gcols.n) * n_L + TPB - 1) / TPB;  // an upper bound of the work; the lists say how much
    if (wgs > 2048) wgs = 2048;                                            // there is (8 workgroups per CU stride over it)
#ifndef ZK_LH_GRID_BY_NODES
    {   // ... and the room the lists have (what the context's earlier builds needed + 25 %) says it better: a small level of a
        // looping trace holds a few thousand entries, not (columns x nodes) -- and every workgroup of a launch has to find a
        // free place on a chip full of other proofs' hash waves before it can even see that it has nothing to do.  (The loop
        // strides over whatever there is: a group dropped on the device, whose columns are hashed densely here, takes more
        // steps per thread, not more threads.)
        size_t room = (b.rcols.n ? (size_t)RUN_SUBS * b.t.r_lists.cap[L] : 0) + (b.gcols.n ? (size_t)RUN_SUBS * b.t.g_lists.cap[L] * b.gcols.n : 0);
        // `expect`: what the context's LAST build of this shape held on this level (+ 25 %) -- the room only ever grows, and a
        // context that has met one trace with long lists would size and schedule every later one as if it were that
        if (expect && expect < room) room = expect;
        size_t w2 = (room + TPB - 1) / TPB;
        if (w2 < 8) w2 = 8;
        if (w2 < wgs) wgs = w2;
    }
#endif
    // fewer than two waves per SIMD even if every node were hashed: the re-arm pauses would only add latency
    const bool small = (size_t)(b.rcols.n + b.gcols.n) * n_L <= (size_t)256 * 4 * 2 * 64;
    // a batched job: the proofs share the launch (gridDim.z), and with them the chip -- the grid is per proof
    const unsigned nz = b.t.nz ? b.t.nz : 1;
    if (nz > 1 && wgs > (2048 + nz - 1) / nz) wgs = (2048 + nz - 1) / nz;
    const bool pause = !small || (size_t)nz * (b.rcols.n + b.gcols.n) * n_L > (size_t)256 * 4 * 2 * 64;
    // priority for the launches that cannot fill the chip (the room of their lists is under half of the 1536 workgroups it holds)
#ifndef ZK_LH_PRIO_WGS
#define ZK_LH_PRIO_WGS 768
#endif
    const int prio = (size_t)wgs * nz <= ZK_LH_PRIO_WGS ? 1 : 0;
    // Unused dynamic LDS on the launches that fill the chip caps their workgroups at five per CU (160 KB / CU; six fit by
    // registers) and leaves registers and a wave slot per SIMD for the short launches of other proofs, which otherwise wait for
    // a workgroup of a big level to retire before they can start at all (A/B, GPU-bound ms per proof, two rounds: no cap 0.504 /
    // 0.503, five 0.498 / 0.494, four 0.519 / 0.524; the small levels take 37-45 us in company instead of 42-56)
#ifndef ZK_LH_BIG_LDS
#define ZK_LH_BIG_LDS 28672
#endif
    const unsigned lds = prio ? 0u : (unsigned)ZK_LH_BIG_LDS;
    if (L == 0) ZK_LAUNCH(kt, (k_level_hash<true, true>), dim3((unsigned)wgs, 1, nz), dim3(TPB), lds, s, b, L, gd, prio);
    else if (!pause) ZK_LAUNCH(kt, (k_level_hash<false, false>), dim3((unsigned)wgs, 1, nz), dim3(TPB), lds, s, b, L, gd, prio);
    else ZK_LAUNCH(kt, (k_level_hash<false, true>), dim3((unsigned)wgs, 1, nz), dim3(TPB), lds, s, b, L, gd, prio);
}

// ------------------------------------------------------------------ the top of the trees
// From a level of at most 2 TPB nodes up to the root in ONE launch, one workgroup per column; levels hand over through
// LDS.  Every hash here is a link in a dependent chain run by a few lone waves, so the permutation is the variant without
// re-arm pauses.  The levels it computes go to TreeRef::upper (a build with lists) or into the slabs.
__global__ __launch_bounds__(TPB) void k_merkle_top(TreeRef t, unsigned first_level, unsigned height) {
    ZK_PRIO_SMALL();
    __shared__ Digest s_d[TPB];  // the level just computed: the next one reads its children here, not from global memory
    const size_t col = blockIdx.y;
    for (unsigned l = first_level; l < height; l++) {
        const size_t n_out = t.npad >> (l + 1);
        Digest d;
        if (threadIdx.x < n_out) {
            Digest a, c;
            if (l == first_level) {
                a = load_digest_at(node_ptr(t, col, l, 2 * threadIdx.x));
                c = load_digest_at(node_ptr(t, col, l, 2 * threadIdx.x + 1));
            } else {
                a = s_d[2 * threadIdx.x];
                c = s_d[2 * threadIdx.x + 1];
            }
            d = sha3_node<false>(a, c);
            store_digest_at(t.lists ? upper_ptr(t, col, l + 1, threadIdx.x) : slab_ptr(t, col, l + 1, threadIdx.x), d);
        }
        __syncthreads();  // everybody has read its children
        if (threadIdx.x < n_out) s_d[threadIdx.x] = d;
        __syncthreads();
    }
}
void launch_merkle_top(const TreeRef &t, unsigned first_level, unsigned height, size_t ncols, hipStream_t s, const KTime *kt) {
    ZK_LAUNCH(kt, k_merkle_top, dim3(1, (unsigned)ncols, t.nz ? t.nz : 1), dim3(TPB), 0, s, t, first_level, height);
}

// ------------------------------------------------------------------ materialising whole trees
// every node of level L of the R / G columns (list levels) and of the levels above them (all columns) into the slabs
__global__ __launch_bounds__(TPB) void k_fill_level(TreeRef t, unsigned L, ColMap cols) {
    const size_t n = t.npad >> L, k = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (k >= n) return;
    const size_t col = cols.c[blockIdx.y];
    const uint8_t *src = node_ptr(t, col, L, k);
    uint8_t *dst = slab_ptr(t, col, L, k);
    if (src != dst) store_digest_at(dst, load_digest_at(src));
}
void launch_fill_virtual(const MerkleBuild &b, hipStream_t s) {
    if (!b.t.lists) return;
    unsigned height = 0;
    while (((size_t)1 << height) < b.npad) height++;
    ColMap rg{};  // the columns whose list levels live in the stores
    for (unsigned c = 0; c < 64; c++)
        if (b.t.y_of_col[c] >= 0 || b.t.g_j_of_col[c] >= 0) rg.c[rg.n++] = (uint8_t)c;
    ColMap all{};
    for (unsigned c = 0; c < 64; c++)
        if (b.t.slab_of_col[c] >= 0) all.c[all.n++] = (uint8_t)c;
    for (unsigned L = 0; L <= height; L++) {
        const ColMap &m = L <= b.t.top ? rg : all;
        if (m.n == 0) continue;
        hipLaunchKernelGGL(k_fill_level, dim3((unsigned)(((b.npad >> L) + TPB - 1) / TPB), m.n), dim3(TPB), 0, s, b.t, L, m);
    }
}

}  // namespace zk
