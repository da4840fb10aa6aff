// Device-side helpers shared by the translation units that build Merkle trees (kernels.hip, merkle_levels.hip): the
// workgroup size, digest loads / stores in the kernels' tree form (keccak.hpp), the timed-launch macro.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "kernels.hpp"
#include "keccak.hpp"

namespace zk {

constexpr int TPB = 256;      // 4 waves per workgroup

// Issue priority of the short dependent launches (structure passes, small levels, tops, evaluations, paths): their waves share
// SIMDs with the big hash levels of other proofs, whose waves always have a VALU instruction ready; the chip runs at most four
// launches at a time (tools/stream_concurrency.hip), so a short launch that crawls holds a quarter of the launch slots.
#ifndef ZK_SMALL_PRIO
#define ZK_SMALL_PRIO 3
#endif
#define ZK_PRIO_SMALL() __builtin_amdgcn_s_setprio(ZK_SMALL_PRIO)

// launch with (kt != nullptr) or without kernel-exact timestamps
#define ZK_LAUNCH(kt, kern, grid, block, lds, s, ...)                                                             \
    do {                                                                                                          \
        if (kt) hipExtLaunchKernelGGL(kern, grid, block, lds, s, (kt)->start, (kt)->stop, 0, __VA_ARGS__);         \
        else hipLaunchKernelGGL(kern, grid, block, lds, s, __VA_ARGS__);                                          \
    } while (0)

// Non-temporal stores: a level's 1-3 GiB of digests are read back once by the next level, from HBM either way; kept out
// of the caches they do not leave the L2 / 256 MB Infinity Cache full of dirty lines whose write-back would compete
// with the next reader of the witness columns (the eval pass right after the build runs at 33 us instead of 37 us,
// tools/merkle_rate.hip; the build itself is unchanged).
__device__ __forceinline__ void store_digest(uint8_t *tree, size_t node, const Digest &d) {
    unsigned long long *q = reinterpret_cast<unsigned long long *>(tree + node * 32);
    __builtin_nontemporal_store(d.w[0], q + 0);  // merged into two global_store_dwordx4 ... nt
    __builtin_nontemporal_store(d.w[1], q + 1);
    __builtin_nontemporal_store(d.w[2], q + 2);
    __builtin_nontemporal_store(d.w[3], q + 3);
}
typedef unsigned int zk_v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_store16(uint4 *dst, const uint4 &v) {  // global_store_dwordx4 ... nt
    zk_v4u x = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(x, reinterpret_cast<zk_v4u *>(dst));
}
__device__ __forceinline__ Digest load_digest(const uint8_t *tree, size_t node) {
    const ulonglong2 *q = reinterpret_cast<const ulonglong2 *>(tree + node * 32);
    ulonglong2 x = q[0], y = q[1];
    return Digest{{x.x, x.y, y.x, y.y}};
}

__device__ __forceinline__ void store_digest_plain(uint8_t *tree, size_t node, const Digest &d) {
    unsigned long long *q = reinterpret_cast<unsigned long long *>(tree + node * 32);
    q[0] = d.w[0]; q[1] = d.w[1]; q[2] = d.w[2]; q[3] = d.w[3];
}

// ---- where a digest lives (kernels.hpp: TreeRef)
__device__ __host__ __forceinline__ size_t slab_level_offset(size_t npad, unsigned l) { return 2 * npad - 2 * (npad >> l); }
// What r_slot needs to know about level l of the R columns -- all of it the same for every thread of a launch.  A kernel that
// resolves many nodes of ONE level (k_level_hash) builds it once with `uniform` = true: the values then sit in scalar
// registers instead of being recomputed, or kept in vector registers, by every lane through every permutation.
struct RLevel {
    size_t n_l, tile, meta_base, ubase_off, list_base;
};
__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {  // a value every lane holds -> scalar registers
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ RLevel r_level(const TreeRef &t, unsigned l, bool uniform = false) {
    RLevel v{t.npad >> l, run_tile_nodes(t.npad, l), run_meta_base(t.npad, t.ncols, l), t.ubase_off[l], t.r_lists.base[l]};
    if (uniform) v = RLevel{uniform64(v.n_l), uniform64(v.tile), uniform64(v.meta_base), uniform64(v.ubase_off), uniform64(v.list_base)};
    return v;
}
// the list slot (within level l) that holds the digest of node k of hinted (R) column y: its own if it was hashed, else its
// leader's -- the nearest hashed node before it in its tile (kernels.hpp: run_tile_nodes)
__device__ __forceinline__ size_t r_slot(const TreeRef &t, const RLevel &v, unsigned y, size_t k) {
    const size_t n_l = v.n_l, tile = v.tile;
    const size_t e0 = v.meta_base + ((size_t)y * n_l) / 64;
    size_t e = e0 + k / 64;
    const unsigned q = (unsigned)(k & 63);
    unsigned long long m = t.bitmap[e] & (q == 63 ? ~0ull : ((2ull << q) - 1));  // hashed nodes of the chunk at or before k
    size_t j;
    if (m) {
        j = (k & ~(size_t)63) + (63 - __builtin_clzll(m));
        m &= ~(1ull << (j & 63));  // hashed nodes of the chunk before the leader
    } else {                       // nothing in this chunk: the last hashed node before it
        j = (k & ~(tile - 1)) + t.prev[e];
        e = e0 + j / 64;
        m = t.bitmap[e] & ((1ull << (j & 63)) - 1);
    }
    const size_t u = (size_t)y * (n_l / tile) + j / tile;
    return (size_t)t.ubase[v.ubase_off + u] + t.woff[e] + (unsigned)__builtin_popcountll(m);
}
__device__ __forceinline__ size_t r_slot(const TreeRef &t, unsigned y, unsigned l, size_t k) { return r_slot(t, r_level(t, l), y, k); }
// the slots of the two children 2 k2, 2 k2 + 1 of a node: they share a 64-node chunk, and list order is node order, so the
// odd child's digest sits right behind the even child's if the odd child was hashed itself -- and in the SAME slot if it is a
// copy (its leader is the even child or the even child's leader).  One leader search instead of two.
__device__ __forceinline__ void r_slot_pair(const TreeRef &t, const RLevel &v, unsigned y, size_t k_even, size_t *s0, size_t *s1) {
    const unsigned long long w = t.bitmap[v.meta_base + ((size_t)y * v.n_l + k_even) / 64];
    *s0 = r_slot(t, v, y, k_even);
    *s1 = *s0 + (unsigned)((w >> ((k_even & 63) + 1)) & 1);
}
__device__ __forceinline__ uint8_t *r_slot_ptr(const TreeRef &t, const RLevel &v, size_t slot) { return t.r_store + (v.list_base + slot) * 32; }
__device__ __forceinline__ uint8_t *r_slot_ptr(const TreeRef &t, unsigned l, size_t slot) {
    return t.r_store + (t.r_lists.base[l] + slot) * 32;
}
__device__ __forceinline__ uint8_t *g_slot_ptr(const TreeRef &t, unsigned l, size_t slot, unsigned j) {
    return t.g_store + ((t.g_lists.base[l] + slot) * t.g_ncols + j) * 32;
}
__device__ __forceinline__ uint8_t *slab_ptr(const TreeRef &t, size_t col, unsigned l, size_t k) {
    if (col < 64 && t.slab_of_col[col] < 0)  // a dropped group whose columns have no slabs: the build is flagged and will be
        return t.upper;                      // repeated; until then nothing may point outside the job's memory
    const size_t s = col < 64 ? (size_t)t.slab_of_col[col] : col;
    return t.slab + (s * 2 * t.npad + slab_level_offset(t.npad, l) + k) * 32;
}
__device__ __forceinline__ uint8_t *upper_ptr(const TreeRef &t, size_t col, unsigned l, size_t k) {
    return t.upper + (col * 512 + 512 - 2 * ((size_t)256 >> (l - t.top)) + k) * 32;
}
// the address of the digest of node k of level l of column col
__device__ __forceinline__ uint8_t *node_ptr(const TreeRef &t, size_t col, unsigned l, size_t k) {
    if (t.lists) {
        if (l > t.top) return upper_ptr(t, col, l, k);
        if (col < 64) {
            const int y = t.y_of_col[col];
            if (y >= 0) return r_slot_ptr(t, l, r_slot(t, (unsigned)y, l, k));
            const int j = t.g_j_of_col[col];
            if (j >= 0 && !*t.g_dropped) return g_slot_ptr(t, l, t.g_rep[slab_level_offset(t.npad, l) + k], (unsigned)j);
        }
    }
    return slab_ptr(t, col, l, k);
}
__device__ __forceinline__ Digest load_digest_at(const uint8_t *p) {
    const ulonglong2 *q = reinterpret_cast<const ulonglong2 *>(p);
    const ulonglong2 x = q[0], y = q[1];
    return Digest{{x.x, x.y, y.x, y.y}};
}
__device__ __forceinline__ void store_digest_at(uint8_t *p, const Digest &d) {
    unsigned long long *q = reinterpret_cast<unsigned long long *>(p);
    q[0] = d.w[0]; q[1] = d.w[1]; q[2] = d.w[2]; q[3] = d.w[3];
}

}  // namespace zk
