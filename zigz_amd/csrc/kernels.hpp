// Launch interface of the gfx950 kernels (kernels.hip).  All pointers are device pointers; tables
// are packed u32 canonical BabyBear elements; every launch is asynchronous on `stream`.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace zk {

// Optional kernel-exact timing of one launch: both events are stamped with the dispatch's own begin / end timestamps
// (hipExtLaunchKernelGGL), not with the position of an event command in the stream, so the elapsed time between them is
// the kernel's duration as rocprofv3 --kernel-trace reports it.
struct KTime {
    hipEvent_t start, stop;
};

// Merkle tree of one column kept in HBM: levels bottom-up, level l (N>>l nodes of 32 B) starts at
// node offset 2N - 2*(N>>l); root at node 2N-2.  Column stride = 2N nodes.
inline size_t tree_level_offset(size_t npad, unsigned level) { return 2 * npad - 2 * (npad >> level); }
inline size_t tree_nodes(size_t npad) { return 2 * npad; }

// u64 host image (already on device) -> packed u32; flag[0] |= 1 when some value >= p.
void launch_narrow_u64(const uint64_t *d_in, uint32_t *d_out, size_t n, uint32_t *d_flag, hipStream_t s);
// d_out[i] = d_in[i] mod p  (F.init on raw 64-bit words, src/core/field.zig:36-38)
void launch_reduce_u64(const uint64_t *d_in, uint32_t *d_out, size_t n, hipStream_t s);
// K8 (witness): packed trace rows [num_steps][43] raw u64 -> 43 columns of 2^nv packed u32, column stride
// `stride`: cols[c][i] = rows[i][c] mod p for i < num_steps; padding i >= num_steps repeats the last row for
// c <= 32 (pc, x0..x31) and is 0 otherwise (src/constraints/witness.zig:80-87,116-123,174-182,249-253).
void launch_witness_rows(const uint64_t *d_rows, size_t num_steps, size_t npad, uint32_t *d_cols, size_t stride,
                         hipStream_t s);
void launch_widen_u32(const uint32_t *d_in, uint64_t *d_out, size_t n, hipStream_t s);
// K8 from the compact trace (include/zigz_hip.h: zigz_trace_step, 48 B per step; TraceStep is its device mirror).
// Four launches: per-64-step summaries of the register writes, a two-level fill-forward scan over the chunks per register, the
// expansion into the 43 padded columns.  ws: witness_steps_ws_words(npad) u32 of scratch.
struct TraceStep {
    uint64_t pc, rd_value, mem_addr, mem_value;
    int64_t imm;
    uint8_t opcode, rd, rs1, rs2, funct3, funct7, wr_reg, mem_is_read;
};
static_assert(sizeof(TraceStep) == 48, "TraceStep must mirror zigz_trace_step (48 bytes)");
// the 32-byte record and its side list (zigz_hip.h: zigz_trace_step32 / zigz_mem_access), widened on the device
struct TraceStep32 {
    uint64_t pc, rd_value;
    int32_t imm;
    uint32_t mem_index;
    uint8_t opcode, rd, rs1, rs2, funct3, funct7, wr_reg, mem_is_read;
};
static_assert(sizeof(TraceStep32) == 32, "TraceStep32 must mirror zigz_trace_step32 (32 bytes)");
struct MemAccess {
    uint64_t addr, value;
};
void launch_steps_widen(const TraceStep32 *d_in, size_t num_steps, const MemAccess *d_mem, size_t num_mem, TraceStep *d_out,
                        hipStream_t s);
// the 16-byte record and the code table (zigz_hip.h: zigz_trace_step16 / zigz_code_entry), widened on the device
struct TraceStep16 {
    uint32_t pc_word, mem_wr;
    uint64_t rd_value;
};
struct CodeEntry {
    int32_t imm;
    uint8_t opcode, rd, rs1, rs2, funct3, funct7, reserved[2];
};
static_assert(sizeof(TraceStep16) == 16 && sizeof(CodeEntry) == 12, "16-byte record / code entry mirrors");
void launch_steps_widen16(const TraceStep16 *d_in, size_t num_steps, const MemAccess *d_mem, size_t num_mem, uint64_t code_base,
                          const CodeEntry *d_code, size_t num_code, TraceStep *d_out, hipStream_t s);
struct Regs32 {
    uint32_t v[32];  // initial register values mod p (x0 = 0)
};
inline size_t witness_steps_ws_words(size_t npad) {
    const size_t nchunks = (npad + 63) / 64;
    return 65 * nchunks + 64 + 32 * ((nchunks + 255) / 256) + 64;  // summaries | carries | flags | per-group carries
}
void launch_witness_steps(const TraceStep *d_steps, size_t num_steps, size_t npad, const Regs32 &init, uint32_t *d_ws,
                          uint32_t *d_cols, size_t stride, hipStream_t s, const KTime *kt_expand = nullptr);

// Where k_block_sums adds its partial sums: counter of (column c, block b, copy k) = sums[k*slot_stride + c*col_stride +
// b*bin_stride], copy k = wave index mod nslots.  u64 atomics serialise per 128-byte cache line (~15 ns each), so when
// few counters receive many partial sums they sit in lines of their own and are replicated; the consumer adds the copies.
struct SumsLayout {
    size_t col_stride, bin_stride, slot_stride;
    unsigned nslots;
};
// K1 (+K2 fused): batched MLE bind.  For column c in [0,ncols):
//   out[c*out_stride + i] = in[c*in_stride + i] + r_c * (in[c*in_stride + i + half] - in[...+ i]),  i < half
// r_c (Montgomery form) = d_r_m ? d_r_m[c] : r_m.  If d_sums: sums[2c] += sum of out[0..half/2),
// sums[2c+1] += sum of out[half/2..half)  (exact u64 sums; must be zeroed by the caller).
void launch_bind(const uint32_t *d_in, size_t in_stride, uint32_t *d_out, size_t out_stride, size_t half,
                 size_t ncols, uint32_t r_m, const uint32_t *d_r_m, unsigned long long *d_sums, hipStream_t s,
                 const KTime *kt = nullptr, const SumsLayout *lay = nullptr);
// layout for the fused sums of launch_bind (plain {2, 1, 0, 1} unless one table feeds > 64 workgroups into a counter)
SumsLayout bind_sums_layout(size_t half, size_t ncols, size_t max_words);
// true when launch_bind takes the vectorised k_bind_vec path for these arguments
bool bind_uses_vec(size_t half, bool with_sums, size_t in_stride, size_t out_stride, const void *in, const void *out);
inline bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }
// layout for half sums of ncols tables of n elements within max_words u64 of sums space: padded + replicated when the
// vector path runs and the words suffice, the plain {2, 1, 0, 1} (sums[2c], sums[2c+1]) otherwise
SumsLayout half_sums_layout(size_t n, size_t ncols, size_t max_words);
// K2/K3: (plain layout) sums[2c] += sum in[0..n/2), sums[2c+1] += sum in[n/2..n)   (n >= 2); n == 1: sums[2c] += in[0].
// lay: optional non-plain layout from half_sums_layout (aligned power-of-two tables of >= 2048 elements only).
void launch_half_sums(const uint32_t *d_in, size_t in_stride, size_t n, size_t ncols,
                      unsigned long long *d_sums, hipStream_t s, const KTime *kt = nullptr, const SumsLayout *lay = nullptr);

// Block sums (k_block_sums): counter(c, b) += sum of in[c][b*m .. (b+1)*m), b < n/m, m = 2^log2_m >= 256,
// n a power of two >= 2*m, exact u64; sums must be zeroed by the caller.  Serves roundPolynomial /
// sumOverHypercube (2 blocks) and the radix-2^k sumcheck stage (<= 1024 blocks; see DESIGN.md "Sumcheck").
void launch_block_sums(const uint32_t *d_in, size_t in_stride, size_t n, unsigned log2_m, size_t ncols,
                       unsigned long long *d_sums, SumsLayout lay, hipStream_t s, const KTime *kt = nullptr);
// Radix-2^k sumcheck stage (k rounds per pass over the table; pass 1 = launch_block_sums):
//  (2) part[c][g][i] = sum over the g-th group of 64 consecutive b of W[c][b] * in[c][b*m + i]   (exact u64; i < m,
//      m % 4 == 0, column c < ncols; W in Montgomery form; G = radix_fold_groups(nb) groups; strides in elements)
size_t radix_fold_groups(size_t nb, int rloops = 0);  // rloops: 16-row chunks per thread (0 = the default, 4)
//      t_start/t_stop (both or neither): events stamped with the dispatch's own begin/end timestamps.
// Columns the eval may leave out: column c with y = y_of_col[c] >= 0 and changed[y] == 0 is constant (established from the
// values by the run-aware structure pass of the same commit job), and the multilinear extension of a constant is that
// constant -- the eq weights of a point sum to 1 -- so its 4 N bytes are not read: k_radix_fold returns at once for it and
// k_weighted_dot writes the column's first value.  changed == nullptr: nothing is skipped.
// A batched commit job (several proofs' columns in one job, MerkleBuild below: zstride) numbers its columns proof by proof:
// column c is column c % ncols1 of proof c / ncols1, whose data starts z_in elements / whose `changed` words z_changed words
// behind the previous proof's.  ncols1 == 0: one proof, plain column stride.
struct EvalSkip {
    const unsigned long long *changed = nullptr;
    signed char y_of_col[64];
    unsigned ncols1 = 0;
    size_t z_in = 0, z_changed = 0;
};
void launch_radix_fold(const uint32_t *d_in, size_t in_stride, size_t m, size_t nb, const uint32_t *d_w_m,
                       size_t w_stride, unsigned long long *d_part, size_t part_col_stride, size_t ncols, hipStream_t s,
                       hipEvent_t t_start = nullptr, hipEvent_t t_stop = nullptr, const EvalSkip *skip = nullptr, int rloops = 0);
//  (3) out[c][i] = (sum_g part[c][g][i]) mod p; if d_sums (ncols == 1): sums[i >> log2_m2] += out[i]  (block sums of
//      the next stage, m2 >= 256; must be zeroed by the caller)
//      publish (single-column use): the LAST workgroup to finish hands the result to the host itself -- kind 1: the n next-stage
//      sums (u64; left zero afterwards), kind 2: the first n outputs (u32) -- into pinned memory and signals `done` there
struct DoneFlag;
struct FinalizePublish {
    int kind = 0;
    unsigned n = 0;
    void *h_dst = nullptr;
    unsigned *count = nullptr;           // (DoneFlag's three words: the struct is declared further down)
    unsigned long long *flag = nullptr;
    unsigned long long seq = 0;
};
void launch_radix_finalize(const unsigned long long *d_part, size_t part_col_stride, size_t groups, uint32_t *d_out,
                           size_t out_stride, size_t m, unsigned log2_m2, unsigned long long *d_sums, size_t ncols,
                           hipStream_t s, const EvalSkip *skip = nullptr, const FinalizePublish *pub = nullptr);
// eq weights of k variables, Montgomery form: W[c][b] = prod_j (bit_j(b) ? r_cj : 1 - r_cj) with bit 0 of the
// loop being the MOST significant bit of b; d_r_m[c*r_stride + j] = r_cj in Montgomery form.  k <= 14.
void launch_eq_weights(const uint32_t *d_r_m, size_t r_stride, unsigned k, uint32_t *d_w_m, size_t w_stride, size_t ncols,
                       hipStream_t s);
// out[c] = sum_i W[c][i] * in[c][i] mod p  (i < n <= 16384; W Montgomery, in canonical)
// the two weight tables of a radix eval (variables 0..kA-1 and kA..kA+kB-1 of every column's point) in one launch
void launch_eq_weights2(const uint32_t *d_r_m, size_t r_stride, unsigned kA, uint32_t *d_wA, size_t strideA, unsigned kB,
                        uint32_t *d_wB, size_t strideB, size_t ncols, hipStream_t s);
void launch_weighted_dot(const uint32_t *d_in, size_t in_stride, const uint32_t *d_w_m, size_t w_stride, size_t n,
                         uint32_t *d_out, size_t ncols, hipStream_t s, const EvalSkip *skip = nullptr,
                         const uint32_t *d_cols = nullptr, size_t col_stride = 0);

// The columns a Keccak launch works on: blockIdx.y = k -> column c[k] (n == 0: identity, column = blockIdx.y).
struct ColMap {
    uint8_t n;
    uint8_t c[64];
};
// ---------------------------------------------------------------- structure-aware levels (merkle_levels.hip)
// The levels 0 .. v - 8 of a tree with >= 2^15 leaves (down to 256 nodes per column) can be built from LISTS of the nodes
// that have to be hashed instead of densely, for two kinds of hinted columns:
//   R  run-aware (piecewise constant columns: the registers, mem.address / mem.value): a node whose subtree and whose left
//      neighbour's are uniform with the same value is a COPY of that neighbour;
//   G  content-addressed group (the columns that are functions of the instruction at pc): a node with the same content in
//      ALL columns of the group as an earlier node of its level shares that node's digests (its REPRESENTATIVE).
// Both are decided from the VALUES on the device, never from the hint, so every node has the dense tree's digest for any
// input.  Copies / non-representatives are VIRTUAL in a commit job: their 32 bytes are never written, readers (the next
// level's hashes, the openings) resolve a node to its leader / representative through RunMeta.
//
// Structure first, hashing second: which nodes are hashed depends on the values only, not on digests, so ALL levels' lists
// are produced up front -- R: one pass over the leaves per stage (k_runs_stage: a workgroup takes 4096 leaves of one column
// and derives the change bitmap and from it, by bit arithmetic in one wave, the lists of levels 0..6; a second, tiny stage
// takes the 4096-node segments of level 6 to levels 7..12, a third would reach 18); G: one table pass per level
// (k_cons_pass: resolve level l-1 + insert level l, wave-deduplicated, generation-tagged slots so the table is never
// cleared) -- and then ONE launch per level hashes the R list, the G list and, when the group was dropped, its columns'
// dense nodes (k_level_hash).  Nothing on this path is read back by the host: whether the group repeats enough to be kept is
// decided on the device (k_cons_decide sets RunMeta::cons_dropped) and every later launch reads that flag.
constexpr unsigned RUN_SEG = 4096;        // input nodes per segment of a stage
constexpr unsigned RUN_STAGE_LEVELS = 6;  // levels a stage adds above its input level (stage 0 also emits its input level 0)
constexpr unsigned RUN_MAX_LEVELS = 20;   // list-driven levels 0 .. v - 8 for v <= 26 (+ 1 spare)
constexpr unsigned RUN_SUBS = 32;         // sub-lists per level, each with a counter in a 128-byte line of its own
constexpr unsigned RUN_NODE_BITS = 26;    // R list entry = hinted-column index << 26 | node
constexpr size_t RUN_MIN_LEAVES = 32768;  // smaller trees are built densely
constexpr size_t RUN_MAX_LEAVES = (size_t)1 << RUN_NODE_BITS;
// counters of a build: word 0 = nodes hashed in all; word 8 (the group's array only) = "group dropped" flag, word 9 = distinct
// leaves found; per level l and sub-list s the length at word (1 + l * RUN_SUBS + s) * 16
constexpr unsigned RUN_CTRS = (1 + RUN_MAX_LEVELS * RUN_SUBS) * 16;
// ... followed, in the R counters, by one word per hinted column: != 0 -> the column is NOT constant (k_runs_stage sets it when
// a segment holds a change or starts with a value other than the column's first).  What the eval of a commit job skips (EvalSkip).
constexpr unsigned RUN_CHANGED = RUN_CTRS;
constexpr unsigned RUN_CTR_WORDS = RUN_CTRS + 64;
__host__ __device__ inline size_t run_ctr_index(unsigned level, unsigned sub) { return (size_t)(1 + level * RUN_SUBS + sub) * 16; }
// the last list-driven level: 256 nodes per column
inline unsigned run_top_level(size_t npad) { unsigned v = 0; while (((size_t)1 << v) < npad) v++; return v - 8; }
// Leader rule.  At level l the nodes of a column fall into TILES; the first node of a tile is always hashed, so a copy's
// leader (the nearest hashed node at or before it) is inside its tile.  A tile is what ONE segment of a stage covers at
// that level: 4096 >> l nodes at the levels 0..6, (4096 or the whole level-6 row) >> (l - 6) at 7..12, and so on.
__host__ __device__ inline size_t run_tile_nodes(size_t npad, unsigned l) {
    if (l == 0) return RUN_SEG;
    const unsigned s = (l - 1) / RUN_STAGE_LEVELS;           // stage that emits level l
    const size_t n_in = npad >> (s * RUN_STAGE_LEVELS);      // nodes per column of the stage's input level
    const size_t seg = n_in < RUN_SEG ? n_in : RUN_SEG;
    return seg >> (l - s * RUN_STAGE_LEVELS);
}
// where the lists of the levels live: level l has RUN_SUBS sub-lists of cap[l] entries starting at entry base[l]
struct LevelLists {
    unsigned top;                               // levels 0 .. top
    unsigned long long base[RUN_MAX_LEVELS];
    unsigned cap[RUN_MAX_LEVELS];
    unsigned long long entries;                 // total
};
// A device pointer of a build that may be BATCHED (TreeRef::nz proofs built by the same launches, gridDim.z = proof): every
// proof's data lives in an arena of the same layout, zs bytes after the previous proof's, and p is proof 0's.  On the device
// the pointer reads as proof blockIdx.z's -- a scalar multiply-add where it is used; the structs that hold these stay
// untouched kernel arguments (a kernel that ADJUSTED its by-value MerkleBuild made the compiler copy all 1.3 KB of it to
// scratch memory: every k_level_hash launch four times slower).  On the host it is the plain pointer p.  zs == 0: one proof.
template <class T>
struct ZPtr {
    T *p;
    size_t zs;
    __host__ __device__ ZPtr &operator=(T *q) {
        p = q;
        return *this;
    }
    __host__ __device__ operator T *() const {
#if defined(__HIP_DEVICE_COMPILE__)
        // (a null pointer of a batched build does not stay null: nothing on the device tests these for null, they are
        // dereferenced only where the build has the data they point to)
        return reinterpret_cast<T *>(reinterpret_cast<uintptr_t>(p) + (size_t)blockIdx.z * zs);
#else
        return p;
#endif
    }
};
// Where the digest of (column, level, node) of a build lives (tree_dev.hpp: node_ptr).  Three kinds of storage:
//   slab    node-addressed, 2 npad nodes per column (level l at node offset 2 npad - 2 (npad >> l)): the columns that are
//           built densely -- and every column of a build that materialises whole trees (single trees, whole-tree tests);
//   stores  the list levels 0..top of the R and G columns in LIST ORDER: a hashed node's digest sits at its list slot, a copy
//           / non-representative resolves to its leader's / representative's slot.  Sized from what the context's previous
//           builds needed, not for the worst case (every node hashed): a build that runs out of room says so (counter word
//           10) and zigz_commit_roots repeats it with more -- so a proof in flight holds ~0.5 GiB of HBM instead of 3.4;
//   upper   the levels above the list levels (255 nodes per column) of all columns of a build with lists.
struct TreeRef {
    size_t npad;
    ZPtr<uint8_t> slab;
    signed char slab_of_col[64];  // column -> index of its slab, -1 = none (columns >= 64: their own index)
    int lists;                    // != 0: levels 0..top of the R / G columns are list-built
    unsigned top;
    ZPtr<uint8_t> upper;          // [column][512 nodes]: level l > top at node offset 512 - 2 (256 >> (l - top))
    // R (run-aware)
    ZPtr<unsigned long long> bitmap;  // per level l at word run_meta_base(l): [hinted column y][node / 64], bit = hashed
    ZPtr<unsigned short> prev;    // same indexing: local index (in the tile) of the last hashed node before the chunk
    ZPtr<unsigned short> woff;    // same indexing: list offset (within its unit) of the chunk's first hashed node
    ZPtr<uint32_t> ubase;         // per level l at ubase_off[l]: [unit = y * (n_l / tile_l) + node / tile_l]: the list slot (within
    unsigned long long ubase_off[RUN_MAX_LEVELS];  // the level: sub-list * capacity + position) of the unit's first hashed node
    unsigned ncols;               // hinted (R) columns
    signed char y_of_col[64];     // column -> hinted index, -1 = not hinted
    ZPtr<uint8_t> r_store;        // digest of list slot i of level l: r_store + (r_lists.base[l] + i) * 32
    LevelLists r_lists;
    // G (content-addressed group)
    signed char g_j_of_col[64];   // column -> index in the group, -1 = not a member
    unsigned g_ncols;
    ZPtr<const uint32_t> g_rep;   // per level (tree_level_offset order): the list slot of the node's representative
    ZPtr<uint8_t> g_store;        // digest of (list slot i, group column j) of level l: g_store + ((g_lists.base[l] + i) * g_ncols + j) * 32
    LevelLists g_lists;
    ZPtr<const unsigned long long> g_dropped;  // device word: != 0 -> the group did not repeat; its columns live in slabs
    unsigned long long g_sd_mask;         // columns of the group whose leaf digests are virtual when the group was dropped
    unsigned long long virtual_leaves;    // bit c: the leaf digests of column c were not written (small-domain columns of a
                                          // commit job): an opening hashes the sibling value itself
    // A BATCHED job -- nz > 1 proofs of the same shape built by the same launches (gridDim.z / .y = proof), for traces so small
    // that one proof's launches are mostly latency: everything a proof's build reads or writes lives in ONE arena of zstride
    // bytes with the same layout for every proof, so that EVERY device pointer of this struct (and of MerkleBuild) moves by
    // proof * zstride; the pointers stored here are proof 0's and carry the stride themselves (ZPtr; set_zstride below).
    // nz == 0: one proof.
    size_t zstride;
    unsigned nz;
};
// first entry of level l in TreeRef::bitmap / ::prev / ::woff (one entry per 64 nodes, levels stored one after the other)
__host__ __device__ inline size_t run_meta_base(size_t npad, unsigned ncols, unsigned l) {
    return (size_t)ncols * ((2 * npad - 2 * (npad >> l)) / 64);
}
inline size_t runs_meta_words(size_t npad, size_t ncols) { return ncols * (2 * npad / 64); }
// cap_in[l] = entries per sub-list the caller wants for level l (0 or nullptr: the worst case -- every node hashed); the
// result holds min(cap_in, worst case)
LevelLists runs_lists(size_t npad, size_t ncols, const unsigned *cap_in = nullptr);  // R: entries are (y << 26 | node)
LevelLists cons_lists(size_t npad, const unsigned *cap_in = nullptr);                // G: entries are nodes (representatives)
// units (column, segment) of the R stages per level and where each level's ubase entries start
size_t runs_units(size_t npad, size_t ncols, unsigned long long ubase_off[RUN_MAX_LEVELS]);
// scratch of the R stages above stage 0: per hinted column the first values of the level-6 (12, 18) nodes and their
// "not uniform" bits (u32 words, then u64 words)
size_t runs_stage_scratch_bytes(size_t npad, size_t ncols);

struct MerkleBuild {   // everything the structure-aware launches share (device pointers)
    ZPtr<const uint32_t> vals;
    size_t val_stride, n_values, npad;
    TreeRef t;                    // where digests go (and how copies / non-representatives resolve)
    // R
    ColMap rcols;
    ZPtr<uint32_t> r_list;
    ZPtr<unsigned long long> r_ctr;  // RUN_CTR_WORDS words, zeroed before the build; word 10: != 0 -> a list ran out of room
    ZPtr<uint8_t> r_stage;        // runs_stage_scratch_bytes()
    // G
    ColMap gcols;                 // the group, ascending
    ColMap gcols_sd;              // its small-domain members (levels 0-1 from the tables when the group is dropped)
    int g_has_slabs;              // the group's columns have slabs to be built into when the group is dropped
    int g_no_probe;               // the context's last builds all kept the group: no probe pass before the full insert
    ZPtr<unsigned long long> g_keys;  // 2 npad slots: generation << 52 | payload
    ZPtr<uint32_t> g_idx;         // 2 npad: the list slot of the node that inserted the slot's key
    ZPtr<uint32_t> g_rep;         // 2 npad: list slot of the representative of every node of the levels 0..top
    ZPtr<uint32_t> g_list;        // list slot -> node
    ZPtr<unsigned long long> g_ctr;  // RUN_CTRS words, zeroed before the build; word 10: bit 0 a list ran out of room, bit 1 the
                                  // group was dropped but its columns have no slabs
    unsigned g_gen;               // generation of level 0 (level l uses g_gen + l); < 4096 - RUN_MAX_LEVELS
};
// a batched build: the arena stride into every pointer of the build (after they have been set)
inline void set_zstride(MerkleBuild &b, size_t zs, unsigned nz) {
    TreeRef &t = b.t;
    t.zstride = zs;
    t.nz = nz;
    t.slab.zs = t.upper.zs = t.bitmap.zs = t.prev.zs = t.woff.zs = t.ubase.zs = t.r_store.zs = t.g_rep.zs = t.g_store.zs = t.g_dropped.zs = zs;
    b.vals.zs = b.r_list.zs = b.r_ctr.zs = b.r_stage.zs = b.g_keys.zs = b.g_idx.zs = b.g_rep.zs = b.g_list.zs = b.g_ctr.zs = zs;
}
// R: all lists, bitmaps and leader tables of the levels 0..top (two or three launches, no hashing)
void launch_runs_structure(const MerkleBuild &b, hipStream_t s, const KTime *kt = nullptr);
// G: the table passes of the levels 0..top and the keep / drop decision (top + 3 launches, no hashing)
void launch_cons_structure(const MerkleBuild &b, hipStream_t s, const KTime *kt = nullptr);
// level L (<= top) of the R and G columns: hashes the two lists (and, when the group was dropped, its columns densely)
void launch_level_hash(const MerkleBuild &b, unsigned L, hipStream_t s, const KTime *kt = nullptr, size_t expect = 0);  // expect: entries the level is expected to hold (sizing only; 0: the lists' room)
// writes the copies / non-representatives of the levels 0..top (whole-tree comparisons, single trees that outlive the call)
void launch_fill_virtual(const MerkleBuild &b, hipStream_t s);
// from level `first_level` (at most 512 nodes per column, found through t) to the root, one workgroup per column; the levels
// it computes go to t.upper (a build with lists) or the slabs
void launch_merkle_top(const TreeRef &t, unsigned first_level, unsigned height, size_t ncols, hipStream_t s, const KTime *kt = nullptr);
// K5: leaf hashes.  tree[c][i] = SHA3(LE64(i < n_values ? vals[c][i] : 0)), i < npad
// (cols: the columns to work on, default all; slabs: for entry k of cols the index of its slab in d_tree, default the column)
void launch_keccak_leaves(const uint32_t *d_vals, size_t val_stride, size_t n_values, size_t npad,
                          uint8_t *d_tree, size_t tree_stride_nodes, size_t ncols, hipStream_t s, const KTime *kt = nullptr,
                          const ColMap *cols = nullptr, const ColMap *slabs = nullptr);
// K6: one level.  out node i = SHA3(in node 2i || in node 2i+1), i < n_out  (node offsets per column)
void launch_keccak_level(uint8_t *d_tree, size_t tree_stride_nodes, size_t in_off, size_t out_off, size_t n_out,
                         size_t ncols, hipStream_t s, const KTime *kt = nullptr, const ColMap *slabs = nullptr);
// Small-domain columns (values < 128 by construction: the instruction-field columns of the witness, x0, is_read): leaf
// digests and level-1 nodes are looked up in two constant tables instead of hashed -- T0[v] = SHA3(LE64(v)), v < 128, and
// T1[a*128+b] = SHA3(T0[a] || T0[b]) (128 + 16384 digests in tree form, 516 KiB) -- whenever all 128 values under a
// wave's 64 level-1 nodes are < 128; otherwise the wave hashes its 3 x 64 digests.  Identical trees either way.
constexpr unsigned SD_DOMAIN = 128;
constexpr size_t SD_TABLE_BYTES = (size_t)(SD_DOMAIN + SD_DOMAIN * SD_DOMAIN) * 32;
void launch_sd_tables(uint8_t *d_tables, hipStream_t s);
// levels 0 and 1 of the columns in `cols` (npad >= 2).  d_todo_count (zeroed by the caller) / d_todo: the waves that found
// a value outside the domain, hashed by a second launch; sd_todo_words(npad, ncols) u32 of list space
inline size_t sd_todo_words(size_t npad, size_t ncols) { return 2 * ncols * ((npad / 2 + 63) / 64) + 2; }
void launch_keccak_small_l01(const uint32_t *d_vals, size_t val_stride, size_t n_values, size_t npad, uint8_t *d_tree,
                             size_t tree_stride_nodes, const ColMap &cols, const uint8_t *d_tables,
                             unsigned long long *d_todo_count, uint32_t *d_todo, hipStream_t s, const KTime *kt = nullptr,
                             bool write_leaves = true, const unsigned long long *d_only_if = nullptr,
                             const ColMap *slabs = nullptr);
// true when launch_keccak_level runs k_keccak_level<HPT> (several hashes per thread) for this level
bool keccak_level_is_wide(size_t n_out, size_t ncols);
// K7: authentication paths.  For column c: index d_idx[c]; siblings -> d_sib[c][l][32], dirs -> d_dirs[c][l],
// leaf value -> d_leaf[c].
// Completion flag of a launch whose results go straight into pinned host memory: `count` is a device word that is zero
// between launches, `flag` a pinned host word that receives `seq` once every workgroup's stores are visible to the host
// (flag == nullptr: no signalling).  The host then waits by polling `flag` with short sleeps instead of a runtime wait.
struct DoneFlag {
    unsigned *count = nullptr;
    unsigned long long *flag = nullptr;
    unsigned long long seq = 0;
};
// n words of device memory into pinned host memory by a kernel that then signals `done` (the host polls it: no copy command, no
// stream wait); rezero: leaves the source zero for the sums of the next pass
void launch_publish_u64(unsigned long long *d_src, size_t n, unsigned long long *h_dst, bool rezero, hipStream_t s, DoneFlag done);
void launch_publish_u32(const uint32_t *d_src, size_t n, uint32_t *h_dst, hipStream_t s, DoneFlag done);
void launch_paths(const TreeRef &t, size_t n_values, unsigned height, const uint32_t *d_vals, size_t val_stride,
                  const uint64_t *d_idx, uint8_t *d_sib, uint8_t *d_dirs, uint32_t *d_leaf, size_t ncols, hipStream_t s,
                  DoneFlag done = DoneFlag());
// a TreeRef for plain node-addressed trees (column c -> slab c)
TreeRef slab_tree_ref(uint8_t *d_tree, size_t npad);
// copies node `node` of every column's tree into d_out[c][32]
void launch_gather_nodes(const uint8_t *d_tree, size_t tree_stride_nodes, size_t node, uint8_t *d_out, size_t ncols,
                         hipStream_t s);
// The roots (canonical bytes) and behind them the counters of the build, in ONE buffer: ncols x 32 B, then JOB_SUMMARY_WORDS u64:
//   [0] d_r_ctr[0] (nodes hashed on the run-aware levels), [1] / [2] d_sd_ctr[0] / [1] (waves that left the small-domain
//   tables), [3] d_g_ctr[0] (digests computed on the content-addressed levels), [4] d_g_ctr[8] (group dropped?), [5] d_g_ctr[9]
//   (its distinct leaves), [6] d_r_ctr[10] | d_g_ctr[10] << 8 (out of room / slabs missing), [7] constant R columns, then per level l < RUN_MAX_LEVELS
//   the longest sub-list of the R lists [8 + l] and of the G lists [8 + RUN_MAX_LEVELS + l]; null pointers read as 0
constexpr unsigned JOB_SUMMARY_WORDS = 8 + 2 * RUN_MAX_LEVELS;
// the counters a build's kernels add to (2 words of small-domain fall-backs, RUN_CTR_WORDS words each of the R and G lists; null =
// not used by this build), zeroed by one launch
void launch_zero_counters(unsigned long long *d_sd_ctr, unsigned long long *d_r_ctr, unsigned long long *d_g_ctr, hipStream_t s,
                          unsigned nz = 0, size_t zstride = 0);
// the columns of the nz proofs of a batched job, gathered into their arenas: proof z's ncols columns of n elements (column
// stride src_stride) -> d_dst + z * zstride bytes, column stride dst_stride (n % 4 == 0, everything 16-byte aligned)
struct ColSrcs {
    const uint32_t *p[32];
};
void launch_gather_cols(const ColSrcs &srcs, unsigned nz, size_t ncols, size_t n, size_t src_stride, uint32_t *d_dst,
                        size_t dst_stride, size_t zstride, hipStream_t s);
void launch_job_summary(const TreeRef &t, unsigned height, uint8_t *d_out, size_t ncols, const unsigned long long *d_r_ctr,
                        const unsigned long long *d_sd_ctr, const unsigned long long *d_g_ctr, hipStream_t s,
                        DoneFlag done = DoneFlag());
// gather element 0 of each column of a strided table
void launch_gather_first(const uint32_t *d_in, size_t stride, uint32_t *d_out, size_t ncols, hipStream_t s);
// K9: Lasso fingerprints (src/lookups/lasso_prover.zig:208-239): rows x width canonical u32 -> u32
void launch_lasso_fingerprints(const uint32_t *d_rows, size_t rows, size_t width, uint32_t *d_out, hipStream_t s,
                               const KTime *kt = nullptr);
// synthetic canonical table for the measurement hooks: out[i] = splitmix-style hash of (seed, i) mod p
void launch_fill_pattern(uint32_t *d_out, size_t n, uint32_t seed, hipStream_t s);

}  // namespace zk
