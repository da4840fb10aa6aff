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

// the node whose digest node k of level l of hinted (R) column y has: itself if it was hashed, else the nearest hashed node
// before it in its tile (kernels.hpp: RunMeta, run_tile_nodes)
__device__ __forceinline__ size_t run_leader(const RunMeta &m, size_t npad, unsigned y, unsigned l, size_t k) {
    const size_t e = run_meta_base(npad, m.ncols, l) + ((size_t)y * (npad >> l) + k) / 64;
    const unsigned q = (unsigned)(k & 63);
    const unsigned long long mm = m.bitmap[e] & (q == 63 ? ~0ull : ((2ull << q) - 1));
    if (mm) return (k & ~(size_t)63) + (63 - __builtin_clzll(mm));
    return (k & ~(run_tile_nodes(npad, l) - 1)) + m.prev[e];
}
// where the digest of node k of level l of column col is stored: the node itself unless the level was built from lists
__device__ __forceinline__ size_t resolve_node(const RunMeta &m, size_t npad, size_t col, unsigned l, size_t k) {
    if (col >= 64) return k;
    const int y = m.y_of_col[col];
    if (y >= 0 && l < m.run_levels) return run_leader(m, npad, (unsigned)y, l, k);
    if (((m.cons_mask >> col) & 1) && l < m.cons_levels && !*m.cons_dropped) return m.cons_rep[2 * npad - 2 * (npad >> l) + k];
    return k;
}

}  // namespace zk
