// Hand-written gfx950 (CDNA4, wave64) kernels for the zigz hot path.  See kernels.hpp for the launch
// interface and DESIGN.md for the per-kernel roofline accounting.
//
//  K1/K2  k_bind_vec / k_bind_small   MLE bind (+ fused next-round half sums)   HBM-bound, 6 B/output-pair
//  K2/K3  k_block_sums / k_half_sums_small  round-polynomial half sums, block sums   HBM-bound, 4 B/element
//  K5     k_keccak_leaves             SHA3-256 leaf hashes                      int-VALU bound
//  K6     k_keccak_level / _top       SHA3-256 level merges                     int-VALU bound
//  K7     k_paths                     authentication-path gather               latency
//  K9     k_lasso_fingerprints        XXH3-64 row fingerprints                  int-VALU bound
#include "kernels.hpp"
#include <hip/hip_ext.h>

#include "field.hpp"
#include "keccak.hpp"
#include "tree_dev.hpp"

namespace zk {

constexpr int UNROLL = 4;     // 16-byte chunks per thread per tile: 8 x 16 B loads in flight
constexpr size_t VEC_MIN_HALF = 4096;  // vector path needs half % (4*TPB*UNROLL) == 0


// ------------------------------------------------------------------ reductions
__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Adds the block's (s0, s1) into sums[0], sums[1]: wave shuffle -> LDS -> one atomic pair per block.
__device__ __forceinline__ void block_add2(unsigned long long s0, unsigned long long s1,
                                           unsigned long long *sums) {
    __shared__ unsigned long long lds[2][TPB / 64];
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        lds[0][wave] = s0;
        lds[1][wave] = s1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t0 = 0, t1 = 0;
#pragma unroll
        for (int w = 0; w < TPB / 64; w++) {
            t0 += lds[0][w];
            t1 += lds[1][w];
        }
        if (t0) atomicAdd(&sums[0], t0);
        if (t1) atomicAdd(&sums[1], t1);
    }
}

// one sum per workgroup: wave shuffle -> LDS -> one atomic
__device__ __forceinline__ void block_add1(unsigned long long v, unsigned long long *dst) {
    __shared__ unsigned long long lds1[TPB / 64];
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) lds1[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
#pragma unroll
        for (int w = 0; w < TPB / 64; w++) t += lds1[w];
        if (t) atomicAdd(dst, t);
    }
}

// ------------------------------------------------------------------ K1 (+K2): MLE bind
// One tile = TPB*UNROLL uint4 chunks of outputs.  Loads are 16 B/lane, fully coalesced (1 KiB per
// wave instruction); 2*UNROLL independent loads are issued before the first use.
template <bool SUMS>
__global__ __launch_bounds__(TPB) void k_bind_vec(const uint32_t *__restrict__ in, size_t in_stride,
                                                  uint32_t *__restrict__ out, size_t out_stride, size_t half,
                                                  uint32_t r_m_scalar, const uint32_t *__restrict__ d_r_m,
                                                  unsigned long long *__restrict__ sums, SumsLayout lay) {
    const size_t col = blockIdx.y;
    const uint32_t r_m = d_r_m ? d_r_m[col] : r_m_scalar;
    const uint4 *lo = reinterpret_cast<const uint4 *>(in + col * in_stride);
    const uint4 *hi = lo + half / 4;
    uint4 *o = reinterpret_cast<uint4 *>(out + col * out_stride);
    const size_t base = (size_t)blockIdx.x * (TPB * UNROLL) + threadIdx.x;

    uint4 a[UNROLL], b[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
        a[u] = lo[base + (size_t)u * TPB];
        b[u] = hi[base + (size_t)u * TPB];
    }
    unsigned long long acc = 0;
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
        uint4 r;
        r.x = bind1(a[u].x, b[u].x, r_m);
        r.y = bind1(a[u].y, b[u].y, r_m);
        r.z = bind1(a[u].z, b[u].z, r_m);
        r.w = bind1(a[u].w, b[u].w, r_m);
        o[base + (size_t)u * TPB] = r;
        if (SUMS) acc += (unsigned long long)r.x + r.y + r.z + r.w;
    }
    if (SUMS) {
        // a tile never straddles the middle of the output (tiles are 4096 outputs, half/2 % 4096 == 0
        // on this path), so the whole block adds to one of the two sums.
        const bool upper = (size_t)blockIdx.x * (TPB * UNROLL * 4) >= half / 2;
        // counter of (column, half) in copy (workgroup mod nslots): see SumsLayout
        block_add1(acc, sums + (size_t)(blockIdx.x % lay.nslots) * lay.slot_stride + col * lay.col_stride + (upper ? lay.bin_stride : 0));
    }
}

// Small tables (half < 8192 with sums / < 4096 without): one workgroup per column, scalar accesses.
template <bool SUMS>
__global__ __launch_bounds__(TPB) void k_bind_small(const uint32_t *__restrict__ in, size_t in_stride,
                                                    uint32_t *__restrict__ out, size_t out_stride, size_t half,
                                                    uint32_t r_m_scalar, const uint32_t *__restrict__ d_r_m,
                                                    unsigned long long *__restrict__ sums) {
    const size_t col = blockIdx.y;
    const uint32_t r_m = d_r_m ? d_r_m[col] : r_m_scalar;
    const uint32_t *p = in + col * in_stride;
    uint32_t *o = out + col * out_stride;
    unsigned long long s0 = 0, s1 = 0;
    for (size_t i = threadIdx.x; i < half; i += TPB) {
        uint32_t v = bind1(p[i], p[i + half], r_m);
        o[i] = v;
        if (SUMS) {
            if (half == 1 || i < half / 2) s0 += v;
            else s1 += v;
        }
    }
    if (SUMS) block_add2(s0, s1, sums + 2 * col);
}

// the vector kernels use 16-byte accesses: every column base must be 16-byte aligned, otherwise the scalar
// workgroup-per-column kernels run (any alignment)
bool bind_uses_vec(size_t half, bool with_sums, size_t in_stride, size_t out_stride, const void *in, const void *out) {
    return half >= (with_sums ? 2 * VEC_MIN_HALF : VEC_MIN_HALF) && (in_stride % 4 == 0) && (out_stride % 4 == 0) &&
           aligned16(in) && aligned16(out);
}

void launch_bind(const uint32_t *d_in, size_t in_stride, uint32_t *d_out, size_t out_stride, size_t half,
                 size_t ncols, uint32_t r_m, const uint32_t *d_r_m, unsigned long long *d_sums, hipStream_t s, const KTime *kt,
                 const SumsLayout *lay_in) {
    if (half == 0 || ncols == 0) return;
    SumsLayout lay = lay_in ? *lay_in : SumsLayout{2, 1, 0, 1};
    if (lay.nslots == 0) lay.nslots = 1;
    const bool vec = bind_uses_vec(half, d_sums != nullptr, in_stride, out_stride, d_in, d_out);
    if (vec) {
        dim3 grid((unsigned)(half / (4 * TPB * UNROLL)), (unsigned)ncols);
        if (d_sums) ZK_LAUNCH(kt, k_bind_vec<true>, grid, dim3(TPB), 0, s, d_in, in_stride, d_out, out_stride, half, r_m, d_r_m, d_sums, lay);
        else ZK_LAUNCH(kt, k_bind_vec<false>, grid, dim3(TPB), 0, s, d_in, in_stride, d_out, out_stride, half, r_m, d_r_m, d_sums, lay);
    } else {
        dim3 grid(1, (unsigned)ncols);
        if (d_sums) ZK_LAUNCH(kt, k_bind_small<true>, grid, dim3(TPB), 0, s, d_in, in_stride, d_out, out_stride, half, r_m, d_r_m, d_sums);
        else ZK_LAUNCH(kt, k_bind_small<false>, grid, dim3(TPB), 0, s, d_in, in_stride, d_out, out_stride, half, r_m, d_r_m, d_sums);
    }
}

// ------------------------------------------------------------------ K2/K3: block sums (half sums = 2 blocks)
// sums[c][b] += sum of in[c][b*m .. (b+1)*m), b < n/m, m = 2^log2_m >= 256.  The unit of work is the WAVE: each wave
// streams W = 256*iters CONTIGUOUS elements that lie inside one block (W <= m), 4 x 16-byte non-temporal loads per lane in flight,
// accumulates exactly in u64, reduces with shuffles and issues ONE atomic.  (The previous form -- one 16 KiB tile per
// workgroup, then shuffle -> LDS -> barrier -> atomic -- spent most of a workgroup's life in that tail with no loads in
// flight: 3.4 TB/s for 43 x 2^20 and, with every workgroup adding to the same two words, 1.2 TB/s for one 2^24 table;
// same-address atomics serialise at ~15 ns each.)  nslots > 1 spreads the adds of one (c, b) over nslots copies of the
// sums array, slot_stride words apart; the consumer adds the copies.
#ifndef ZK_BS_INFLIGHT
#define ZK_BS_INFLIGHT 4  // (16 loses a quarter on the 13 us launches of one 2^24 table; 4 and 8 tie on 43 x 2^20, 4 wins there)
#endif
constexpr int BS_INFLIGHT = ZK_BS_INFLIGHT;
// A table that is read ONCE by a pass says so: non-temporal loads (global_load_dwordx4 ... nt) do not displace what the caches
// hold for somebody else and stream faster on this part -- cold 43 x 2^20 launches, A/B on one box (profiles/r04_ab_notes.txt):
// k_block_sums 35.3 -> 31.4 us (0.64 -> 0.72 of 8 TB/s), k_radix_fold 34.4 -> 30.5 us (0.68 -> 0.76), one 2^24 table 14.4 ->
// 13.3 us.  (-DZK_STREAM_PLAIN: the plain loads of rounds 1-3, for A/B.)
#ifndef ZK_STREAM_PLAIN
__device__ __forceinline__ uint4 zk_stream_load(const uint4 *q) {
    const zk_v4u v = __builtin_nontemporal_load(reinterpret_cast<const zk_v4u *>(q));
    return make_uint4(v.x, v.y, v.z, v.w);
}
#define ZK_STREAM_LOAD(q) zk_stream_load(q)
#else
#define ZK_STREAM_LOAD(q) (*(q))
#endif
#ifndef ZK_BS_ITERS_SHIFT
#define ZK_BS_ITERS_SHIFT 0
#endif
#ifndef ZK_FOLD_RLOOPS
#define ZK_FOLD_RLOOPS 4
#endif
__global__ __launch_bounds__(TPB) void k_block_sums(const uint32_t *__restrict__ in, size_t in_stride, size_t n, unsigned log2_m,
                                                    unsigned iters, unsigned long long *__restrict__ sums, SumsLayout lay) {
    const size_t col = blockIdx.y;
    const unsigned lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
    const size_t c0 = wave * (size_t)iters * 64;  // first 16-byte chunk of this wave
    if (c0 >= n / 4) return;                      // wave-uniform; no barrier below
    const uint4 *p = reinterpret_cast<const uint4 *>(in + col * in_stride) + c0 + lane;
    unsigned long long acc = 0;
#pragma unroll 1
    for (unsigned it = 0; it < iters; it += BS_INFLIGHT) {
        uint4 a[BS_INFLIGHT];
#pragma unroll
        for (int j = 0; j < BS_INFLIGHT; j++) a[j] = ZK_STREAM_LOAD(p + (size_t)(it + j < iters ? it + j : iters - 1) * 64);  // clamped, not branched
#pragma unroll
        for (int j = 0; j < BS_INFLIGHT; j++)
            acc += it + j < iters ? (unsigned long long)a[j].x + a[j].y + a[j].z + a[j].w : 0ull;
    }
    acc = wave_sum(acc);
    if (lane == 0 && acc)
        atomicAdd(&sums[(size_t)(wave % lay.nslots) * lay.slot_stride + col * lay.col_stride + ((c0 * 4) >> log2_m) * lay.bin_stride], acc);
}

__global__ __launch_bounds__(TPB) void k_half_sums_small(const uint32_t *__restrict__ in, size_t in_stride, size_t n,
                                                         unsigned long long *__restrict__ sums) {
    const size_t col = blockIdx.y;
    const uint32_t *p = in + col * in_stride;
    unsigned long long s0 = 0, s1 = 0;
    for (size_t i = threadIdx.x; i < n; i += TPB) {
        if (n == 1 || i < n / 2) s0 += p[i];
        else s1 += p[i];
    }
    block_add2(s0, s1, sums + 2 * col);
}

static unsigned floor_log2(size_t v) { unsigned l = 0; while (v > 1) { v >>= 1; l++; } return l; }

// 16-byte chunks x 64 lanes per wave iteration; aim at ~8192 waves per launch (8 per SIMD: enough bytes in flight for
// the HBM, and as few atomics as that allows), a wave's range inside one block
static size_t block_sums_iters(size_t n, unsigned log2_m, size_t ncols) {
    const size_t m = (size_t)1 << log2_m;
    size_t iters = (ncols * n / 256) / 8192;
    iters = iters < 1 ? 1 : (size_t)1 << floor_log2(iters);
    iters <<= ZK_BS_ITERS_SHIFT;
    if (iters > m / 256) iters = m / 256;
    if (iters > 64) iters = 64;
    return iters;
}

SumsLayout half_sums_layout(size_t n, size_t ncols, size_t max_words) {
    // counters in 128-byte lines of their own (atomics serialise per cache line), copies while more than 64 waves
    // would add into one counter and the words last
    if (n < 2048 || (n & (n - 1))) return SumsLayout{2, 1, 0, 1};
    const size_t per_addr = (n / 2) / (256 * block_sums_iters(n, floor_log2(n) - 1, ncols));
    size_t slots = (per_addr + 63) / 64;
    if (slots > 64) slots = 64;
    while (slots > 1 && slots * ncols * 32 > max_words) slots--;
    if (ncols * 32 > max_words) return SumsLayout{2, 1, 0, 1};
    return SumsLayout{32, 16, ncols * 32, (unsigned)(slots < 1 ? 1 : slots)};
}

SumsLayout bind_sums_layout(size_t half, size_t ncols, size_t max_words) {
    // k_bind_vec<true>: one atomic per workgroup of 4096 outputs; (half/2)/4096 of them per counter
    const size_t per_addr = half / 2 / 4096;
    if (per_addr <= 64 || ncols * 32 > max_words) return SumsLayout{2, 1, 0, 1};
    size_t slots = (per_addr + 63) / 64;
    if (slots > 64) slots = 64;
    while (slots > 1 && slots * ncols * 32 > max_words) slots--;
    return SumsLayout{32, 16, ncols * 32, (unsigned)slots};
}

void launch_block_sums(const uint32_t *d_in, size_t in_stride, size_t n, unsigned log2_m, size_t ncols,
                       unsigned long long *d_sums, SumsLayout lay, hipStream_t s, const KTime *kt) {
    const size_t iters = block_sums_iters(n, log2_m, ncols);
    const size_t waves = n / (256 * iters);
    if (lay.nslots == 0) lay.nslots = 1;
    dim3 grid((unsigned)((waves + TPB / 64 - 1) / (TPB / 64)), (unsigned)ncols);
    ZK_LAUNCH(kt, k_block_sums, grid, dim3(TPB), 0, s, d_in, in_stride, n, log2_m, (unsigned)iters, d_sums, lay);
}

void launch_half_sums(const uint32_t *d_in, size_t in_stride, size_t n, size_t ncols, unsigned long long *d_sums,
                      hipStream_t s, const KTime *kt, const SumsLayout *lay) {
    if (n == 0 || ncols == 0) return;
    if (n >= 2048 && (n & (n - 1)) == 0 && in_stride % 4 == 0 && aligned16(d_in)) {
        launch_block_sums(d_in, in_stride, n, floor_log2(n) - 1, ncols, d_sums, lay ? *lay : SumsLayout{2, 1, 0, 1}, s, kt);
    } else {
        // (only reached with the plain layout: half_sums_layout returns it for these shapes; unaligned callers pass none)
        ZK_LAUNCH(kt, k_half_sums_small, dim3(1, (unsigned)ncols), dim3(TPB), 0, s, d_in, in_stride, n, d_sums);
    }
}

// ------------------------------------------------------------------ radix-2^k sumcheck stage
// Radix-2^k fold  T'[i] = sum_b W[b] * T[b*m + i].  Work split: a workgroup owns 1024 consecutive outputs (one
// 16-byte load per lane per table row) and RB*RLOOPS = 64 CONSECUTIVE rows, i.e. it streams one contiguous range when
// m = 1024 and long row segments otherwise -- the HBM access pattern of k_bind_vec.  Each (row-group g, output i)
// partial sum is written once as an exact u64 (part[col][g][i]; no atomics, no memset); k_radix_finalize adds the
// G = nb/64 partials and reduces mod p.
// The Montgomery reduction is deferred: 32-bit multiplies issue at under half the VALU rate on gfx950
// (tools/fold_rate.hip), and four of them per term made this kernel multiplier-bound (41 us against a 33.5 us read
// ceiling at v = 20).  The 64-bit products W[b]*T[..] are accumulated as two exact u64 sums of their 32-bit halves
// (64 terms: each < 2^38); one reduction per output then gives  hi + lo * 2^-32  ==  sum_b w_b * T[b*m+i]  (mod p).
constexpr int RB = 16;     // independent 16-byte loads in flight per lane
constexpr int RLOOPS = ZK_FOLD_RLOOPS;  // RB-chunks per thread: accumulators stay in registers
// FULL: nb is a multiple of RB*RLOOPS (every launch but the late, small sumcheck stages): no row needs a bounds test
// and the RB loads of a chunk are issued back to back.  Otherwise rows past nb are clamped to a valid row and given
// weight 0 (per-row branches would make the compiler serialise the loads behind s_waitcnt vmcnt(0)).
__device__ __forceinline__ bool eval_skips(const EvalSkip &skip, size_t col) {
    if (!skip.changed) return false;
    size_t z = 0;
    if (skip.ncols1) {  // a batched job: column col % ncols1 of proof col / ncols1
        z = col / skip.ncols1;
        col -= z * skip.ncols1;
    }
    if (col >= 64) return false;
    const int y = skip.y_of_col[col];
    return y >= 0 && skip.changed[z * skip.z_changed + y] == 0;
}
// first element of column `col` of the committed table (batched jobs: kernels.hpp EvalSkip)
__device__ __forceinline__ size_t eval_col_off(const EvalSkip &skip, size_t col, size_t stride) {
    if (!skip.ncols1) return col * stride;
    const size_t z = col / skip.ncols1;
    return z * skip.z_in + (col - z * skip.ncols1) * stride;
}
template <bool FULL>
__global__ __launch_bounds__(TPB) void k_radix_fold(const uint32_t *__restrict__ in, size_t in_stride, size_t m, size_t nb,
                                                    const uint32_t *__restrict__ w_m, size_t w_stride,
                                                    unsigned long long *__restrict__ part, size_t part_col_stride, EvalSkip skip,
                                                    int rloops) {
    ZK_PRIO_SMALL();
    const size_t q = (size_t)blockIdx.x * TPB + threadIdx.x;  // uint4 index of the outputs
    if (q * 4 >= m) return;
    const size_t col = blockIdx.z;
    if (eval_skips(skip, col)) return;  // a constant column: k_weighted_dot writes its value, its partial sums are never used
    const uint4 *p = reinterpret_cast<const uint4 *>(in + eval_col_off(skip, col, in_stride)) + q;
    const uint32_t *w = w_m + col * w_stride;
    const size_t mq = m / 4;
    unsigned long long lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
#pragma unroll 1
    for (int l = 0; l < rloops; l++) {
        const size_t b0 = ((size_t)blockIdx.y * rloops + l) * RB;
        if (!FULL && b0 >= nb) break;
        uint4 v[RB];
#pragma unroll
        for (int j = 0; j < RB; j++) v[j] = ZK_STREAM_LOAD(p + (FULL || b0 + j < nb ? b0 + j : nb - 1) * mq);
#pragma unroll
        for (int j = 0; j < RB; j++) {
            // wave-uniform, Montgomery form, < p
            const uint32_t wj = FULL ? w[b0 + j] : (w[b0 + j < nb ? b0 + j : nb - 1] & (b0 + j < nb ? ~0u : 0u));
            const uint32_t e[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const unsigned long long pr = (unsigned long long)wj * e[c];
                lo[c] += (uint32_t)pr;
                hi[c] += pr >> 32;
            }
        }
    }
    unsigned long long s[4];
#pragma unroll
    for (int c = 0; c < 4; c++) s[c] = hi[c] + monty_reduce(lo[c]);  // lo < 2^38 < p * 2^32
    ulonglong2 *o = reinterpret_cast<ulonglong2 *>(part + col * part_col_stride + (size_t)blockIdx.y * m + q * 4);
    o[0] = make_ulonglong2(s[0], s[1]);
    o[1] = make_ulonglong2(s[2], s[3]);
}

size_t radix_fold_groups(size_t nb, int rloops) {
    if (rloops <= 0) rloops = RLOOPS;
    return (nb + (size_t)RB * rloops - 1) / ((size_t)RB * rloops);
}

void launch_radix_fold(const uint32_t *d_in, size_t in_stride, size_t m, size_t nb, const uint32_t *d_w_m,
                       size_t w_stride, unsigned long long *d_part, size_t part_col_stride, size_t ncols, hipStream_t s,
                       hipEvent_t t_start, hipEvent_t t_stop, const EvalSkip *skip, int rloops) {
    if (rloops <= 0) rloops = RLOOPS;
    dim3 grid((unsigned)((m / 4 + TPB - 1) / TPB), (unsigned)radix_fold_groups(nb, rloops), (unsigned)ncols);
    auto kern = nb % ((size_t)RB * rloops) == 0 ? k_radix_fold<true> : k_radix_fold<false>;
    const EvalSkip sk = skip ? *skip : EvalSkip();
    if (t_start && t_stop)  // kernel-exact timing for the roofline figure (an event pair around a launch adds the gaps)
        hipExtLaunchKernelGGL(kern, grid, dim3(TPB), 0, s, t_start, t_stop, 0, d_in, in_stride, m, nb, d_w_m, w_stride,
                              d_part, part_col_stride, sk, rloops);
    else
        hipLaunchKernelGGL(kern, grid, dim3(TPB), 0, s, d_in, in_stride, m, nb, d_w_m, w_stride, d_part, part_col_stride, sk,
                           rloops);
}

__global__ __launch_bounds__(TPB) void k_radix_finalize(const unsigned long long *__restrict__ part, size_t part_col_stride,
                                                        size_t groups, uint32_t *__restrict__ out, size_t out_stride,
                                                        size_t m, unsigned log2_m2, unsigned long long *__restrict__ sums,
                                                        EvalSkip skip, FinalizePublish pub) {
    ZK_PRIO_SMALL();
    __shared__ int s_last;
    const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    const size_t col = blockIdx.z;
    if (eval_skips(skip, col)) return;  // (its partial sums were never written)
    uint32_t v = 0;
    if (i < m) {
        const unsigned long long *pp = part + col * part_col_stride + i;
        unsigned long long t = 0;
        for (size_t g = 0; g < groups; g++) t += pp[g * m];
        v = (uint32_t)(t % (unsigned long long)P);
        out[col * out_stride + i] = v;
    }
    if (sums) {  // single-column use; m2 >= 256: the 64 outputs of a wave fall into one block of the next stage
        unsigned long long t = wave_sum((unsigned long long)v);
        if ((threadIdx.x & 63) == 0 && i < m && t) atomicAdd(&sums[i >> log2_m2], t);
    }
    if (pub.kind) {  // the last workgroup to get here publishes (a sumcheck's read-back without a launch of its own)
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) s_last = atomicAdd(pub.count, 1u) == gridDim.x - 1;
        __syncthreads();
        if (s_last) {
            __threadfence();
            if (pub.kind == 1) {
                for (unsigned j = threadIdx.x; j < pub.n; j += TPB) {
                    reinterpret_cast<unsigned long long *>(pub.h_dst)[j] = __hip_atomic_load(&sums[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    sums[j] = 0;
                }
            } else {
                for (unsigned j = threadIdx.x; j < pub.n; j += TPB) reinterpret_cast<uint32_t *>(pub.h_dst)[j] = out[j];
            }
            __threadfence_system();
            __syncthreads();
            if (threadIdx.x == 0) {
                *pub.count = 0;
                __hip_atomic_store(pub.flag, pub.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

void launch_radix_finalize(const unsigned long long *d_part, size_t part_col_stride, size_t groups, uint32_t *d_out,
                           size_t out_stride, size_t m, unsigned log2_m2, unsigned long long *d_sums, size_t ncols,
                           hipStream_t s, const EvalSkip *skip, const FinalizePublish *pub) {
    dim3 grid((unsigned)((m + TPB - 1) / TPB), 1, (unsigned)ncols);
    hipLaunchKernelGGL(k_radix_finalize, grid, dim3(TPB), 0, s, d_part, part_col_stride, groups, d_out, out_stride, m, log2_m2,
                       d_sums, skip ? *skip : EvalSkip(), pub && ncols == 1 ? *pub : FinalizePublish());
}

// eq weights by doubling in LDS, one workgroup per column (all values Montgomery form: mont_mul keeps the form)
__global__ __launch_bounds__(TPB) void k_eq_weights(const uint32_t *__restrict__ r_m, size_t r_stride, unsigned k,
                                                    uint32_t *__restrict__ w_m, size_t w_stride) {
    ZK_PRIO_SMALL();
    extern __shared__ uint32_t eqw[];
    const size_t col = blockIdx.x;
    if (threadIdx.x == 0) eqw[0] = R_MOD_P;
    __syncthreads();
    for (unsigned j = 0; j < k; j++) {
        const uint32_t r = r_m[col * r_stride + j];
        const uint32_t one_minus = sub_mod(R_MOD_P, r);
        const size_t cur = (size_t)1 << j;
        // in-place doubling from the top so sources are read before they are overwritten: W'[2x+1], W'[2x] <- W[x]
        for (size_t base = 0; base < cur; base += TPB) {
            const size_t x = cur - 1 - (base + threadIdx.x);  // descending order across iterations
            uint32_t v = 0;
            const bool live = base + threadIdx.x < cur;
            if (live) v = eqw[x];
            __syncthreads();
            if (live) {
                eqw[2 * x + 1] = mont_mul(v, r);
                eqw[2 * x] = mont_mul(v, one_minus);
            }
            __syncthreads();
        }
    }
    for (size_t i = threadIdx.x; i < ((size_t)1 << k); i += TPB) w_m[col * w_stride + i] = eqw[i];
}

// both weight tables of the radix eval in ONE launch: blockIdx.y = 0 -> the first kA variables into wA, 1 -> the next kB into wB
__global__ __launch_bounds__(TPB) void k_eq_weights2(const uint32_t *__restrict__ r_m, size_t r_stride, unsigned kA,
                                                     uint32_t *__restrict__ wA, size_t strideA, unsigned kB,
                                                     uint32_t *__restrict__ wB, size_t strideB) {
    ZK_PRIO_SMALL();
    extern __shared__ uint32_t eqw[];
    const size_t col = blockIdx.x;
    const bool second = blockIdx.y != 0;
    const unsigned k = second ? kB : kA;
    const uint32_t *rr = r_m + col * r_stride + (second ? kA : 0);
    uint32_t *w = second ? wB + col * strideB : wA + col * strideA;
    if (threadIdx.x == 0) eqw[0] = R_MOD_P;
    __syncthreads();
    for (unsigned j = 0; j < k; j++) {
        const uint32_t r = rr[j];
        const uint32_t one_minus = sub_mod(R_MOD_P, r);
        const size_t cur = (size_t)1 << j;
        for (size_t base = 0; base < cur; base += TPB) {  // (as k_eq_weights: in-place doubling from the top)
            const size_t x = cur - 1 - (base + threadIdx.x);
            uint32_t v = 0;
            const bool live = base + threadIdx.x < cur;
            if (live) v = eqw[x];
            __syncthreads();
            if (live) {
                eqw[2 * x + 1] = mont_mul(v, r);
                eqw[2 * x] = mont_mul(v, one_minus);
            }
            __syncthreads();
        }
    }
    for (size_t i = threadIdx.x; i < ((size_t)1 << k); i += TPB) w[i] = eqw[i];
}
void launch_eq_weights2(const uint32_t *d_r_m, size_t r_stride, unsigned kA, uint32_t *d_wA, size_t strideA, unsigned kB,
                        uint32_t *d_wB, size_t strideB, size_t ncols, hipStream_t s) {
    const unsigned kmax = kA > kB ? kA : kB;
    hipLaunchKernelGGL(k_eq_weights2, dim3((unsigned)ncols, 2), dim3(TPB), ((size_t)1 << kmax) * 4, s, d_r_m, r_stride, kA, d_wA,
                       strideA, kB, d_wB, strideB);
}

void launch_eq_weights(const uint32_t *d_r_m, size_t r_stride, unsigned k, uint32_t *d_w_m, size_t w_stride, size_t ncols,
                       hipStream_t s) {
    hipLaunchKernelGGL(k_eq_weights, dim3((unsigned)ncols), dim3(TPB), ((size_t)1 << k) * 4, s, d_r_m, r_stride, k, d_w_m,
                       w_stride);
}

__global__ __launch_bounds__(TPB) void k_weighted_dot(const uint32_t *__restrict__ in, size_t in_stride,
                                                      const uint32_t *__restrict__ w_m, size_t w_stride, size_t n,
                                                      uint32_t *__restrict__ out, EvalSkip skip, const uint32_t *__restrict__ cols,
                                                      size_t col_stride) {
    ZK_PRIO_SMALL();
    __shared__ unsigned long long red[TPB / 64];
    const size_t col = blockIdx.x;
    if (cols && eval_skips(skip, col)) {  // (workgroup-uniform) the extension of a constant column is the constant
        if (threadIdx.x == 0) out[col] = cols[eval_col_off(skip, col, col_stride)];
        return;
    }
    unsigned long long acc = 0;
    for (size_t i = threadIdx.x; i < n; i += TPB) acc += mont_mul(w_m[col * w_stride + i], in[col * in_stride + i]);
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < TPB / 64; w++) t += red[w];
        out[col] = (uint32_t)(t % (unsigned long long)P);
    }
}

void launch_weighted_dot(const uint32_t *d_in, size_t in_stride, const uint32_t *d_w_m, size_t w_stride, size_t n,
                         uint32_t *d_out, size_t ncols, hipStream_t s, const EvalSkip *skip, const uint32_t *d_cols,
                         size_t col_stride) {
    hipLaunchKernelGGL(k_weighted_dot, dim3((unsigned)ncols), dim3(TPB), 0, s, d_in, in_stride, d_w_m, w_stride, n, d_out,
                       skip ? *skip : EvalSkip(), skip ? d_cols : nullptr, col_stride);
}

// ------------------------------------------------------------------ layout conversion at the boundary
__global__ __launch_bounds__(TPB) void k_narrow_u64(const uint64_t *__restrict__ in, uint32_t *__restrict__ out,
                                                    size_t n, uint32_t *flag) {
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    const size_t step = (size_t)gridDim.x * TPB;
    uint32_t bad = 0;
    for (; i < n; i += step) {
        uint64_t v = in[i];
        bad |= (v >= (uint64_t)P) ? 1u : 0u;
        out[i] = (uint32_t)v;
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

__global__ __launch_bounds__(TPB) void k_reduce_u64(const uint64_t *__restrict__ in, uint32_t *__restrict__ out,
                                                    size_t n) {
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    const size_t step = (size_t)gridDim.x * TPB;
    for (; i < n; i += step) out[i] = (uint32_t)(in[i] % (uint64_t)P);
}

__global__ __launch_bounds__(TPB) void k_widen_u32(const uint32_t *__restrict__ in, uint64_t *__restrict__ out,
                                                   size_t n) {
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    const size_t step = (size_t)gridDim.x * TPB;
    for (; i < n; i += step) out[i] = in[i];
}

// ------------------------------------------------------------------ K8: witness columns from packed trace rows
// One workgroup transposes a tile of 64 steps x 43 words through LDS: coalesced 8-byte reads of the row-major
// trace, `x mod p`, then per column 64 consecutive u32 (256 B per wave store).  LDS row pitch 43 words (odd)
// keeps the transposed reads conflict-free.
constexpr int WROW = 43, WTILE = 64;
__global__ __launch_bounds__(TPB) void k_witness_rows(const uint64_t *__restrict__ rows, size_t num_steps, size_t npad,
                                                      uint32_t *__restrict__ cols, size_t stride) {
    __shared__ uint32_t tile[WTILE * WROW];
    const size_t base = (size_t)blockIdx.x * WTILE;
    for (int k = threadIdx.x; k < WTILE * WROW; k += TPB) {
        const size_t step = base + (size_t)(k / WROW);
        const int c = k % WROW;
        uint32_t v = 0;
        if (step < num_steps) v = (uint32_t)(rows[base * WROW + k] % (uint64_t)P);
        else if (c <= 32) v = (uint32_t)(rows[(num_steps - 1) * WROW + c] % (uint64_t)P);
        tile[k] = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (base + lane < npad)
        for (int c = wave; c < WROW; c += TPB / 64) cols[(size_t)c * stride + base + lane] = tile[lane * WROW + c];
}

void launch_witness_rows(const uint64_t *d_rows, size_t num_steps, size_t npad, uint32_t *d_cols, size_t stride,
                         hipStream_t s) {
    if (num_steps == 0) return;
    hipLaunchKernelGGL(k_witness_rows, dim3((unsigned)((npad + WTILE - 1) / WTILE)), dim3(TPB), 0, s, d_rows, num_steps, npad,
                       d_cols, stride);
}

// ------------------------------------------------------------------ K8 from the compact trace
// One wave = one chunk of 64 consecutive steps.  Register r's column is the fill-forward of the values written to r:
//   pass 1 (k_steps_summary)  per chunk and register: the last value written inside the chunk, if any
//   pass 2 (k_steps_scan)     per register: carry[r][chunk] = value of r before the chunk (fill-forward over chunks)
//   pass 3 (k_steps_expand)   per step: last write to r at or before the step inside the chunk (ballot + prefix mask +
//                             find-last-set + cross-lane read), else the carry; 43 coalesced 256-byte column stores.
// Reads 48 B and writes 172 B per step (the packed-rows kernel reads 344 B); the H2D of the trace shrinks by 7.2x.
__device__ __forceinline__ uint32_t mod_p64(uint64_t x) { return (uint32_t)(x % (uint64_t)P); }

__global__ __launch_bounds__(TPB) void k_steps_summary(const TraceStep *__restrict__ steps, size_t num_steps, size_t nchunks,
                                                       uint32_t *__restrict__ sum_val, uint32_t *__restrict__ sum_has) {
    const size_t chunk = (size_t)blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
    if (chunk >= nchunks) return;  // wave-uniform
    const unsigned lane = threadIdx.x & 63;
    const size_t i = chunk * 64 + lane;
    unsigned wr = 0;
    uint32_t val = 0;
    if (i < num_steps) {
        wr = steps[i].wr_reg;
        val = mod_p64(steps[i].rd_value);
    }
    uint32_t mine = 0;  // lane r keeps register r's summary
    unsigned long long has = 0;
#pragma unroll 1
    for (unsigned r = 1; r < 32; r++) {
        const unsigned long long mask = __ballot(wr == r);
        if (mask) {  // wave-uniform
            const int j = 63 - __builtin_clzll(mask);
            const uint32_t v = __shfl(val, j, 64);
            if (lane == r) mine = v;
            has |= 1ull << r;
        }
    }
    if (lane >= 1 && lane < 32) sum_val[(size_t)lane * nchunks + chunk] = mine;
    if (lane == 0) sum_has[chunk] = (uint32_t)has;
}

// Fill-forward over the chunks in two levels.  A GROUP is TPB consecutive chunks (16 Ki steps).
//   k_steps_scan_local   one workgroup per (group, register): an inclusive "last defined value" scan over the group's 256
//                        chunk summaries (shuffles inside a wave, LDS across the four waves) -> per chunk the value of the
//                        register BEFORE the chunk if it was written earlier in the group, else UNDEF; per group the last value
//                        written in it (or UNDEF)
//   k_steps_scan_groups  one wave per register: the same scan over the group summaries, seeded with the initial register
//                        file -> the value before each group
// k_steps_expand reads the chunk's entry and falls back to its group's.  (One workgroup per register walking all chunks in
// 64 dependent rounds took 60-70 us at 2^20 steps; this takes two launches of a few microseconds.)
constexpr uint32_t STEPS_UNDEF = 0xffffffffu;  // (values are < p < 2^31)
__global__ __launch_bounds__(TPB) void k_steps_scan_local(const uint32_t *__restrict__ sum_val, const uint32_t *__restrict__ sum_has,
                                                          size_t nchunks, uint32_t *__restrict__ carry, uint32_t *__restrict__ gsum,
                                                          size_t ngroups) {
    __shared__ uint32_t w_val[TPB / 64], w_has[TPB / 64];
    const unsigned r = blockIdx.y + 1;
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t c = (size_t)blockIdx.x * TPB + threadIdx.x;
    const bool live = c < nchunks;
    uint32_t h = live ? (sum_has[c] >> r) & 1u : 0u;
    uint32_t v = h ? sum_val[(size_t)r * nchunks + c] : 0u;
    // inclusive scan inside the wave: (h, v) <- the nearest defined entry at or before this lane
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t ph = __shfl_up(h, off, 64), pv = __shfl_up(v, off, 64);
        if (lane >= (unsigned)off && !h) { h = ph; v = pv; }
    }
    if (lane == 63) { w_has[wave] = h; w_val[wave] = v; }
    __syncthreads();
    // value before this lane's chunk: the previous lane's inclusive result, else the previous waves', else undefined here
    uint32_t ph = __shfl_up(h, 1, 64), pv = __shfl_up(v, 1, 64);
    if (lane == 0) { ph = 0; pv = 0; }
    uint32_t before = STEPS_UNDEF;
    bool found = false;
    if (ph) { before = pv; found = true; }
    for (int w = (int)wave - 1; w >= 0 && !found; w--)
        if (w_has[w]) { before = w_val[w]; found = true; }
    if (live) carry[(size_t)r * nchunks + c] = before;
    if (threadIdx.x == TPB - 1) {  // the last value written in this group
        uint32_t nv = STEPS_UNDEF;
        bool f = false;
        if (h) { nv = v; f = true; }
        for (int w = (int)wave - 1; w >= 0 && !f; w--)
            if (w_has[w]) { nv = w_val[w]; f = true; }
        gsum[(size_t)r * ngroups + blockIdx.x] = nv;
    }
}
__global__ __launch_bounds__(64) void k_steps_scan_groups(uint32_t *__restrict__ gsum /* in: summaries, out: value before the group */,
                                                          size_t ngroups, Regs32 init) {
    const unsigned r = blockIdx.x + 1, lane = threadIdx.x;
    uint32_t run = init.v[r];  // the value before the current block of 64 groups (wave-uniform)
    for (size_t base = 0; base < ngroups; base += 64) {
        const size_t g = base + lane;
        uint32_t v = g < ngroups ? gsum[(size_t)r * ngroups + g] : STEPS_UNDEF;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {  // inclusive: the nearest defined summary at or before this lane
            const uint32_t pv = __shfl_up(v, off, 64);
            if (lane >= (unsigned)off && v == STEPS_UNDEF) v = pv;
        }
        uint32_t before = __shfl_up(v, 1, 64);
        if (lane == 0 || before == STEPS_UNDEF) before = run;
        const uint32_t last = __shfl(v, 63, 64);
        if (g < ngroups) gsum[(size_t)r * ngroups + g] = before;
        if (last != STEPS_UNDEF) run = last;
    }
}

__global__ __launch_bounds__(TPB) void k_steps_expand(const TraceStep *__restrict__ steps, size_t num_steps, size_t npad,
                                                      const uint32_t *__restrict__ carry, const uint32_t *__restrict__ gcarry,
                                                      size_t nchunks, size_t ngroups, uint32_t *__restrict__ cols, size_t stride) {
    const size_t chunk = (size_t)blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
    if (chunk >= nchunks) return;  // wave-uniform
    const unsigned lane = threadIdx.x & 63;
    const size_t i = chunk * 64 + lane;
    const bool live = i < num_steps, store = i < npad;
    const TraceStep st = steps[live ? i : num_steps - 1];  // padding repeats the last step's pc (witness.zig:80-87)
    const unsigned wr = live ? st.wr_reg : 0;
    const uint32_t val = mod_p64(st.rd_value);
    if (store) {
        cols[i] = mod_p64(st.pc);                                   // column 0: pc
        cols[1 * stride + i] = 0;                                   // x0
        cols[33 * stride + i] = live ? st.opcode : 0;               // instruction fields pad with 0 (witness.zig:174-182)
        cols[34 * stride + i] = live ? st.rd : 0;
        cols[35 * stride + i] = live ? st.rs1 : 0;
        cols[36 * stride + i] = live ? st.rs2 : 0;
        cols[37 * stride + i] = live ? st.funct3 : 0;
        cols[38 * stride + i] = live ? st.funct7 : 0;
        cols[39 * stride + i] = live ? mod_p64((uint64_t)st.imm) : 0;  // u64(bitcast(i64 imm)) mod p (witness.zig:170)
        cols[40 * stride + i] = live ? mod_p64(st.mem_addr) : 0;    // memory columns pad with 0 (witness.zig:249-253)
        cols[41 * stride + i] = live ? mod_p64(st.mem_value) : 0;
        cols[42 * stride + i] = live ? st.mem_is_read : 0;
    }
    const unsigned long long le = lane == 63 ? ~0ull : ((2ull << lane) - 1);  // lanes <= mine
#pragma unroll 1
    for (unsigned r = 1; r < 32; r++) {
        const unsigned long long m = __ballot(wr == r) & le;
        const int src = m ? 63 - __builtin_clzll(m) : 0;
        const uint32_t w = __shfl(val, src, 64);
        uint32_t v = w;
        if (!m) {  // not written in this chunk so far: what it held before the chunk (else before the chunk's group)
            v = carry[(size_t)r * nchunks + chunk];
            if (v == STEPS_UNDEF) v = gcarry[(size_t)r * ngroups + chunk / TPB];
        }
        if (store) cols[(size_t)(1 + r) * stride + i] = v;  // registers repeat their last value in the padding (:116-123)
    }
}

// 32-byte records + the side list of memory accesses -> the 48-byte records the expansion reads (80 B of HBM traffic per step:
// noise next to the 32 B per step that crossed PCIe to get here)
__global__ __launch_bounds__(TPB) void k_steps_widen(const TraceStep32 *__restrict__ in, size_t n, const MemAccess *__restrict__ mem,
                                                     size_t num_mem, TraceStep *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const uint4 a = reinterpret_cast<const uint4 *>(in)[2 * i], b = reinterpret_cast<const uint4 *>(in)[2 * i + 1];
    uint64_t addr = 0, value = 0;
    if (b.y < num_mem) {  // (ZIGZ_NO_MEM_ACCESS and any other index past the list: no access)
        const ulonglong2 m = reinterpret_cast<const ulonglong2 *>(mem)[b.y];
        addr = m.x;
        value = m.y;
    }
    const long long imm = (long long)(int)b.x;
    uint4 *o = reinterpret_cast<uint4 *>(out) + 3 * i;
    o[0] = a;                                                                       // pc, rd_value
    o[1] = make_uint4((uint32_t)addr, (uint32_t)(addr >> 32), (uint32_t)value, (uint32_t)(value >> 32));
    o[2] = make_uint4((uint32_t)imm, (uint32_t)((unsigned long long)imm >> 32), b.z, b.w);  // imm, the eight field bytes
}
void launch_steps_widen(const TraceStep32 *d_in, size_t num_steps, const MemAccess *d_mem, size_t num_mem, TraceStep *d_out,
                        hipStream_t s) {
    if (num_steps == 0) return;
    hipLaunchKernelGGL(k_steps_widen, dim3((unsigned)((num_steps + TPB - 1) / TPB)), dim3(TPB), 0, s, d_in, num_steps, d_mem, num_mem,
                       d_out);
}

// 16-byte records + side list + code table -> the same 48-byte records (64 B of HBM traffic per step + the table, which a loop keeps
// in the caches)
__global__ __launch_bounds__(TPB) void k_steps_widen16(const TraceStep16 *__restrict__ in, size_t n, const MemAccess *__restrict__ mem,
                                                       size_t num_mem, uint64_t code_base, const uint32_t *__restrict__ code,
                                                       size_t num_code, TraceStep *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const uint4 a = reinterpret_cast<const uint4 *>(in)[i];  // pc_word, mem_wr, rd_value
    const uint64_t pc = code_base + (a.x & ~3u);
    const uint32_t mi = a.y & 0x7ffffffu, wr = a.y >> 27, is_read = a.x & 1u, ci = a.x >> 2;
    uint64_t addr = 0, value = 0;
    if (mi < num_mem) {  // (ZIGZ_NO_MEM_ACCESS16 and any other index past the list: no access)
        const ulonglong2 m = reinterpret_cast<const ulonglong2 *>(mem)[mi];
        addr = m.x;
        value = m.y;
    }
    uint32_t c0 = 0, c1 = 0, c2 = 0;  // imm | opcode rd rs1 rs2 | funct3 funct7 - -
    if (ci < num_code) {
        c0 = code[3 * (size_t)ci];
        c1 = code[3 * (size_t)ci + 1];
        c2 = code[3 * (size_t)ci + 2];
    }
    const long long imm = (long long)(int)c0;
    uint4 *o = reinterpret_cast<uint4 *>(out) + 3 * i;
    o[0] = make_uint4((uint32_t)pc, (uint32_t)(pc >> 32), a.z, a.w);  // pc, rd_value
    o[1] = make_uint4((uint32_t)addr, (uint32_t)(addr >> 32), (uint32_t)value, (uint32_t)(value >> 32));
    // imm, then the eight field bytes: opcode rd rs1 rs2 | funct3 funct7 wr_reg mem_is_read
    o[2] = make_uint4((uint32_t)imm, (uint32_t)((unsigned long long)imm >> 32), c1, (c2 & 0xffffu) | (wr << 16) | (is_read << 24));
}
void launch_steps_widen16(const TraceStep16 *d_in, size_t num_steps, const MemAccess *d_mem, size_t num_mem, uint64_t code_base,
                          const CodeEntry *d_code, size_t num_code, TraceStep *d_out, hipStream_t s) {
    if (num_steps == 0) return;
    hipLaunchKernelGGL(k_steps_widen16, dim3((unsigned)((num_steps + TPB - 1) / TPB)), dim3(TPB), 0, s, d_in, num_steps, d_mem, num_mem,
                       code_base, (const uint32_t *)d_code, num_code, d_out);
}

void launch_witness_steps(const TraceStep *d_steps, size_t num_steps, size_t npad, const Regs32 &init, uint32_t *d_ws,
                          uint32_t *d_cols, size_t stride, hipStream_t s, const KTime *kt_expand) {
    if (num_steps == 0) return;
    const size_t nchunks = (npad + 63) / 64;
    const size_t ngroups = (nchunks + TPB - 1) / TPB;
    uint32_t *sum_val = d_ws, *carry = d_ws + 32 * nchunks, *sum_has = d_ws + 64 * nchunks, *gsum = d_ws + 65 * nchunks + 64;
    const dim3 grid((unsigned)((nchunks + TPB / 64 - 1) / (TPB / 64)));
    hipLaunchKernelGGL(k_steps_summary, grid, dim3(TPB), 0, s, d_steps, num_steps, nchunks, sum_val, sum_has);
    hipLaunchKernelGGL(k_steps_scan_local, dim3((unsigned)ngroups, 31), dim3(TPB), 0, s, sum_val, sum_has, nchunks, carry, gsum, ngroups);
    hipLaunchKernelGGL(k_steps_scan_groups, dim3(31), dim3(64), 0, s, gsum, ngroups, init);
    ZK_LAUNCH(kt_expand, k_steps_expand, grid, dim3(TPB), 0, s, d_steps, num_steps, npad, carry, gsum, nchunks, ngroups, d_cols, stride);
}

static unsigned stream_grid(size_t n) {
    size_t b = (n + TPB - 1) / TPB;
    return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

void launch_narrow_u64(const uint64_t *d_in, uint32_t *d_out, size_t n, uint32_t *d_flag, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_narrow_u64, dim3(stream_grid(n)), dim3(TPB), 0, s, d_in, d_out, n, d_flag);
}
void launch_reduce_u64(const uint64_t *d_in, uint32_t *d_out, size_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_reduce_u64, dim3(stream_grid(n)), dim3(TPB), 0, s, d_in, d_out, n);
}
void launch_widen_u32(const uint32_t *d_in, uint64_t *d_out, size_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_widen_u32, dim3(stream_grid(n)), dim3(TPB), 0, s, d_in, d_out, n);
}

// ------------------------------------------------------------------ K5/K6: Keccak Merkle
// Tree nodes are stored in the bit-interleaved "tree form" of keccak.hpp (32 B per node, the same size as the SHA3
// byte string); k_paths / k_gather_nodes convert to canonical bytes on the way out.
// (digest loads / stores: tree_dev.hpp)
// HPT hashes per thread (strided by the workgroup size so loads/stores stay coalesced): amortises wave launch
// and set-up over several ~4.2 k-instruction permutations.
// occupancy experiments (tools/merkle_rate.hip): unused dynamic LDS per workgroup caps the workgroups per CU
#ifndef ZK_LEAVES_DYN_LDS
#define ZK_LEAVES_DYN_LDS 0
#endif
#ifndef ZK_LEVEL_DYN_LDS
#define ZK_LEVEL_DYN_LDS 0
#endif
#ifndef ZK_HPT
#define ZK_HPT 4
#endif
constexpr int HPT = ZK_HPT;
#ifndef ZK_LEAVES_MIN_WAVES
#define ZK_LEAVES_MIN_WAVES 1
#endif
__global__ __launch_bounds__(TPB, ZK_LEAVES_MIN_WAVES) void k_keccak_leaves(const uint32_t *__restrict__ vals, size_t val_stride,
                                                       size_t n_values, size_t npad, uint8_t *__restrict__ tree,
                                                       size_t tree_stride_nodes, ColMap cmap, ColMap tmap) {
    const size_t col = cmap.n ? cmap.c[blockIdx.y] : blockIdx.y;
    uint8_t *t = tree + (tmap.n ? (size_t)tmap.c[blockIdx.y] : col) * tree_stride_nodes * 32;  // (the column's slab)
    const uint32_t *v = vals + col * val_stride;
#pragma unroll 1
    for (int h = 0; h < HPT; h++) {
        const size_t i = ((size_t)blockIdx.x * HPT + h) * TPB + threadIdx.x;
        if (i >= npad) return;
        const uint64_t x = i < n_values ? v[i] : 0;  // pad with hashLeaf(0), merkle_tree.zig:302-306
        store_digest(t, i, sha3_leaf(x));
    }
}

// One level.  The 64 hashes of a wave consume 4 KiB of contiguous child digests: they are fetched with fully
// coalesced 16-byte loads (1 KiB per wave instruction) and handed to their lanes through LDS (rows padded to 80 B so
// the per-lane 4 x ds_read_b128 are bank-conflict free), instead of four 64-byte-strided loads per lane.
// Requires n_out % 64 == 0 (levels handled here have >= 512 nodes).
#ifndef ZK_LEVEL_MIN_WAVES
#define ZK_LEVEL_MIN_WAVES 1
#endif
template <int H>
__global__ __launch_bounds__(TPB, ZK_LEVEL_MIN_WAVES) void k_keccak_level(uint8_t *__restrict__ tree, size_t tree_stride_nodes,
                                                      size_t in_off, size_t out_off, size_t n_out, ColMap cmap) {
    __shared__ uint4 stage[TPB / 64][64 * 5];
    const size_t col = cmap.n ? cmap.c[blockIdx.y] : blockIdx.y;
    uint8_t *t = tree + col * tree_stride_nodes * 32;
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll 1
    for (int h = 0; h < H; h++) {
        const size_t first = ((size_t)blockIdx.x * H + h) * TPB + (size_t)wave * 64;  // first output node of this wave
        const bool active = first < n_out;                                          // wave-uniform
        if (active) {
            const uint4 *g = reinterpret_cast<const uint4 *>(t + (in_off + 2 * first) * 32);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const unsigned c = lane + 64 * k;  // 16-byte chunk of the wave's 4 KiB
                stage[wave][(c >> 2) * 5 + (c & 3)] = g[c];  // (non-temporal loads here keep the witness in the
                // Infinity Cache for the eval pass, 28 us instead of 33, but slow this kernel by 3 %: not taken)
            }
        }
        __syncthreads();
        if (active) {
            const uint4 *r = &stage[wave][lane * 5];
            const uint4 a0 = r[0], a1 = r[1], b0 = r[2], b1 = r[3];
            Digest l{{((uint64_t)a0.y << 32) | a0.x, ((uint64_t)a0.w << 32) | a0.z, ((uint64_t)a1.y << 32) | a1.x,
                      ((uint64_t)a1.w << 32) | a1.z}};
            Digest rr{{((uint64_t)b0.y << 32) | b0.x, ((uint64_t)b0.w << 32) | b0.z, ((uint64_t)b1.y << 32) | b1.x,
                       ((uint64_t)b1.w << 32) | b1.z}};
            // (measured and not kept: digest stores through wave-private LDS, 1 805 vs 1 797 us; wave barriers instead of
            // the workgroup barriers, same; requesting the next iteration's children before the hash, 1 975 vs 1 853 us)
            store_digest(t, out_off + first + lane, sha3_node(l, rr));
        }
        __syncthreads();
    }
}

void launch_keccak_leaves(const uint32_t *d_vals, size_t val_stride, size_t n_values, size_t npad, uint8_t *d_tree,
                          size_t tree_stride_nodes, size_t ncols, hipStream_t s, const KTime *kt, const ColMap *cols,
                          const ColMap *slabs) {
    ColMap cm{}, tm{};
    if (cols) cm = *cols;
    if (slabs) tm = *slabs;
    if (cols && cm.n == 0) return;  // an explicit, empty column list
    dim3 grid((unsigned)((npad + TPB * HPT - 1) / (TPB * HPT)), (unsigned)(cols ? cm.n : ncols));
    ZK_LAUNCH(kt, k_keccak_leaves, grid, dim3(TPB), ZK_LEAVES_DYN_LDS, s, d_vals, val_stride, n_values, npad, d_tree,
              tree_stride_nodes, cm, tm);
}

#ifndef ZK_WIDE_MIN_WGS
#define ZK_WIDE_MIN_WGS 4096  // (512, i.e. four hashes per thread down to 0.5 M nodes: 1.585 against 1.555 ms of GPU per proof at 14
#endif                        // lanes, tools/ab_runs.sh -- the small levels want the waves)
bool keccak_level_is_wide(size_t n_out, size_t ncols) { return n_out * ncols >= (size_t)TPB * HPT * ZK_WIDE_MIN_WGS; }

void launch_keccak_level(uint8_t *d_tree, size_t tree_stride_nodes, size_t in_off, size_t out_off, size_t n_out,
                         size_t ncols, hipStream_t s, const KTime *kt, const ColMap *cols) {
    ColMap cm{};
    if (cols) cm = *cols;
    if (cols && cm.n == 0) return;
    const size_t nc = cols ? cm.n : ncols;
    // several hashes per thread only while that still leaves >= 16 workgroups per CU (small levels need the waves)
    if (keccak_level_is_wide(n_out, nc)) {
        dim3 grid((unsigned)((n_out + TPB * HPT - 1) / (TPB * HPT)), (unsigned)nc);
        ZK_LAUNCH(kt, k_keccak_level<HPT>, grid, dim3(TPB), ZK_LEVEL_DYN_LDS, s, d_tree, tree_stride_nodes, in_off, out_off, n_out, cm);
    } else {
        dim3 grid((unsigned)((n_out + TPB - 1) / TPB), (unsigned)nc);
        ZK_LAUNCH(kt, k_keccak_level<1>, grid, dim3(TPB), ZK_LEVEL_DYN_LDS, s, d_tree, tree_stride_nodes, in_off, out_off, n_out, cm);
    }
}

// ------------------------------------------------------------------ small-domain columns: levels 0 and 1 by table
__global__ __launch_bounds__(TPB) void k_sd_tables(uint8_t *__restrict__ tables) {
    const unsigned t = blockIdx.x * TPB + threadIdx.x;  // (a, b) = (t / 128, t % 128)
    if (t >= SD_DOMAIN * SD_DOMAIN) return;
    const unsigned a = t / SD_DOMAIN, b = t % SD_DOMAIN;
    const Digest la = sha3_leaf(a), lb = sha3_leaf(b);
    store_digest(tables + (size_t)SD_DOMAIN * 32, t, sha3_node(la, lb));
    if (b == 0) store_digest(tables, a, la);
}
void launch_sd_tables(uint8_t *d_tables, hipStream_t s) {
    hipLaunchKernelGGL(k_sd_tables, dim3(SD_DOMAIN * SD_DOMAIN / TPB), dim3(TPB), 0, s, d_tables);
}

// Pass 1: the lookups.  One thread = one level-1 node = two leaves.  A wave whose 128 values are not all inside the
// domain does nothing but append itself to the to-do list (kept out of this kernel so that it stays small: a few VGPRs,
// full occupancy, HBM-write-bound).  Pass 2 (k_keccak_small_fallback) hashes the listed waves; it is always launched
// with a small fixed grid that walks the list, and costs a few microseconds when the list is empty.
__global__ __launch_bounds__(TPB) void k_keccak_small_l01(const uint32_t *__restrict__ vals, size_t val_stride, size_t n_values,
                                                          size_t npad, uint8_t *__restrict__ tree, size_t tree_stride_nodes,
                                                          ColMap cmap, const uint8_t *__restrict__ tables,
                                                          unsigned long long *__restrict__ todo_count,
                                                          uint32_t *__restrict__ todo, int write_leaves,
                                                          const unsigned long long *__restrict__ only_if, ColMap tmap) {
    if (only_if && !*only_if) return;  // these columns went another way (a content-addressed group that was kept)
    const size_t col = cmap.c[blockIdx.y];
    const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;  // level-1 node
    const bool live = i < npad / 2;
    uint8_t *t = tree + (tmap.n ? (size_t)tmap.c[blockIdx.y] : col) * tree_stride_nodes * 32;  // (the column's slab)
    const uint2 *v = reinterpret_cast<const uint2 *>(vals + col * val_stride);
    // padding leaves hash the value 0 (merkle_tree.zig:302-306)
    uint2 x = make_uint2(0, 0);
    if (live && 2 * i + 1 < n_values) x = v[i];
    else if (live && 2 * i < n_values) x.x = vals[col * val_stride + 2 * i];
    if (__all(x.x < SD_DOMAIN && x.y < SD_DOMAIN)) {  // wave-uniform: every value under this wave's nodes is in the domain
        // (npad >= 1024: a wave is entirely live or entirely past the end)
        if (live) {
            // the wave's 128 leaf digests are 4 KiB of contiguous tree, its 64 nodes 2 KiB: hand them through LDS so every
            // store instruction writes 1 KiB of consecutive bytes (32-byte digests stored lane by lane are quarter-line
            // partial writes: 1.1 TB/s instead of HBM rate)
            __shared__ uint4 sh[TPB / 64][64 * 6];
            uint4 *mine = sh[threadIdx.x >> 6];
            const unsigned lane = threadIdx.x & 63;
            const uint4 *t0 = reinterpret_cast<const uint4 *>(tables);
            const uint4 *t1 = reinterpret_cast<const uint4 *>(tables + (size_t)SD_DOMAIN * 32);
            const size_t e = (size_t)x.x * SD_DOMAIN + x.y;
            if (write_leaves) {  // (a commit job leaves them out: nothing but an opening reads a leaf digest of these
                // columns, and k_paths hashes that one value itself)
                mine[4 * lane + 0] = t0[2 * x.x];
                mine[4 * lane + 1] = t0[2 * x.x + 1];
                mine[4 * lane + 2] = t0[2 * x.y];
                mine[4 * lane + 3] = t0[2 * x.y + 1];
            }
            mine[256 + 2 * lane + 0] = t1[2 * e];
            mine[256 + 2 * lane + 1] = t1[2 * e + 1];
            __builtin_amdgcn_wave_barrier();  // LDS is in order within a wave; keep the compiler from moving the reads up
            const size_t w0 = i - lane;       // first node of this wave
            uint4 *leaves = reinterpret_cast<uint4 *>(t + 2 * w0 * 32), *nodes = reinterpret_cast<uint4 *>(t + (npad + w0) * 32);
            if (write_leaves) {
#pragma unroll
                for (int k = 0; k < 4; k++) nt_store16(leaves + 64 * k + lane, mine[64 * k + lane]);
            }
#pragma unroll
            for (int k = 0; k < 2; k++) nt_store16(nodes + 64 * k + lane, mine[256 + 64 * k + lane]);
        }
    } else if ((threadIdx.x & 63) == 0) {  // the bound does not hold here (a caller's hint was wrong): leave it to pass 2
        const unsigned long long k = atomicAdd(todo_count, 1ull);
        todo[2 * k] = (uint32_t)blockIdx.y;
        todo[2 * k + 1] = (uint32_t)(i >> 6);
    }
}

__global__ __launch_bounds__(TPB) void k_keccak_small_fallback(const uint32_t *__restrict__ vals, size_t val_stride,
                                                               size_t n_values, size_t npad, uint8_t *__restrict__ tree,
                                                               size_t tree_stride_nodes, ColMap cmap,
                                                               const unsigned long long *__restrict__ todo_count,
                                                               const uint32_t *__restrict__ todo, int write_leaves,
                                                               const unsigned long long *__restrict__ only_if, ColMap tmap) {
    if (only_if && !*only_if) return;
    const unsigned long long count = *todo_count;
    const unsigned lane = threadIdx.x & 63;
    for (unsigned long long w = (unsigned long long)blockIdx.x * (TPB / 64) + (threadIdx.x >> 6); w < count;
         w += (unsigned long long)gridDim.x * (TPB / 64)) {
        const size_t col = cmap.c[todo[2 * w]];
        const size_t i = (size_t)todo[2 * w + 1] * 64 + lane;
        if (i >= npad / 2) continue;
        uint8_t *t = tree + (tmap.n ? (size_t)tmap.c[todo[2 * w]] : col) * tree_stride_nodes * 32;
        const uint32_t *v = vals + col * val_stride;
        const uint64_t v0 = 2 * i < n_values ? v[2 * i] : 0, v1 = 2 * i + 1 < n_values ? v[2 * i + 1] : 0;
        const Digest l0 = sha3_leaf(v0), l1 = sha3_leaf(v1);
        if (write_leaves) {
            store_digest(t, 2 * i, l0);
            store_digest(t, 2 * i + 1, l1);
        }
        store_digest(t, npad + i, sha3_node(l0, l1));
    }
}

void launch_keccak_small_l01(const uint32_t *d_vals, size_t val_stride, size_t n_values, size_t npad, uint8_t *d_tree,
                             size_t tree_stride_nodes, const ColMap &cols, const uint8_t *d_tables,
                             unsigned long long *d_todo_count, uint32_t *d_todo, hipStream_t s, const KTime *kt,
                             bool write_leaves, const unsigned long long *d_only_if, const ColMap *slabs) {
    if (cols.n == 0 || npad < 2) return;
    ColMap tm{};
    if (slabs) tm = *slabs;
    dim3 grid((unsigned)((npad / 2 + TPB - 1) / TPB), (unsigned)cols.n);
    ZK_LAUNCH(kt, k_keccak_small_l01, grid, dim3(TPB), 0, s, d_vals, val_stride, n_values, npad, d_tree, tree_stride_nodes, cols,
              d_tables, d_todo_count, d_todo, write_leaves ? 1 : 0, d_only_if, tm);
    // list-driven; 512 workgroups walk the to-do list (empty unless a hint was wrong: then a few microseconds)
    hipLaunchKernelGGL(k_keccak_small_fallback, dim3(512), dim3(TPB), 0, s, d_vals, val_stride, n_values, npad, d_tree,
                       tree_stride_nodes, cols, d_todo_count, d_todo, write_leaves ? 1 : 0, d_only_if, tm);
}

// ------------------------------------------------------------------ K7: authentication paths
// The last workgroup of a launch that wrote its results into pinned host memory says so there: every workgroup makes its
// stores visible system-wide, then counts itself; the one that completes the count stores the sequence number the host polls
// for (and leaves the counter at zero for the next launch).  A workgroup here is one wave.
__device__ __forceinline__ void signal_done(const DoneFlag &done, unsigned n_groups) {
    if (!done.flag) return;
    __threadfence_system();
    if (threadIdx.x == 0) {
        const unsigned prev = atomicAdd(done.count, 1u);
        if (prev == n_groups - 1) {
            *done.count = 0;
            __hip_atomic_store(done.flag, done.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__global__ __launch_bounds__(64) void k_paths(TreeRef t, size_t n_values, unsigned height, const uint32_t *__restrict__ vals,
                                              size_t val_stride, const uint64_t *__restrict__ idx, uint8_t *__restrict__ sib,
                                              uint8_t *__restrict__ dirs, uint32_t *__restrict__ leaf, DoneFlag done) {
    ZK_PRIO_SMALL();
    const size_t col = blockIdx.x;
    const size_t fc = (size_t)blockIdx.z * gridDim.x + col;  // a batched job: proof blockIdx.z; results are numbered proof by proof
    if (t.zstride) vals += (size_t)blockIdx.z * (t.zstride / 4);
    const size_t index = idx[fc];
    const unsigned l = threadIdx.x;
    if (l == 0) leaf[fc] = vals[col * val_stride + index];
    if (l < height) {
        const size_t ci = index >> l;  // current_index at level l, merkle_tree.zig:335-352
        const size_t node = ci ^ 1;    // the sibling: a copy / non-representative resolves to where its digest is stored (node_ptr)
        bool virt_leaf = l == 0 && col < 64 && ((t.virtual_leaves >> col) & 1);  // leaf digests of this column were never written
        if (l == 0 && col < 64 && ((t.g_sd_mask >> col) & 1) && *t.g_dropped) virt_leaf = true;
        Digest d;
        if (virt_leaf) d = sha3_leaf<false>(node < n_values ? (uint64_t)vals[col * val_stride + node] : 0);
        else d = load_digest_at(node_ptr(t, col, l, node));
        d = canonical_digest(d);  // tree form -> SHA3 bytes at the boundary
        ulonglong2 *q = reinterpret_cast<ulonglong2 *>(sib + (fc * height + l) * 32);
        q[0] = make_ulonglong2(d.w[0], d.w[1]);
        q[1] = make_ulonglong2(d.w[2], d.w[3]);
        dirs[fc * height + l] = (uint8_t)(ci & 1);  // directions[l] = is_right
    }
    signal_done(done, gridDim.x * gridDim.z);
}

void launch_paths(const TreeRef &t, size_t n_values, unsigned height, const uint32_t *d_vals, size_t val_stride,
                  const uint64_t *d_idx, uint8_t *d_sib, uint8_t *d_dirs, uint32_t *d_leaf, size_t ncols, hipStream_t s,
                  DoneFlag done) {
    hipLaunchKernelGGL(k_paths, dim3((unsigned)ncols, 1, t.nz ? t.nz : 1), dim3(64), 0, s, t, n_values, height, d_vals, val_stride, d_idx,
                       d_sib, d_dirs, d_leaf, done);
}
TreeRef slab_tree_ref(uint8_t *d_tree, size_t npad) {
    TreeRef t{};
    t.npad = npad;
    t.slab = d_tree;
    for (int c = 0; c < 64; c++) {
        t.slab_of_col[c] = (signed char)c;
        t.y_of_col[c] = -1;
        t.g_j_of_col[c] = -1;
    }
    return t;
}

__global__ void k_gather_nodes(const uint8_t *__restrict__ tree, size_t tree_stride_nodes, size_t node,
                               uint8_t *__restrict__ out, size_t ncols) {
    size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncols) return;
    const Digest d = canonical_digest(load_digest(tree + c * tree_stride_nodes * 32, node));  // tree form -> SHA3 bytes
    ulonglong2 *q = reinterpret_cast<ulonglong2 *>(out + c * 32);
    q[0] = make_ulonglong2(d.w[0], d.w[1]);
    q[1] = make_ulonglong2(d.w[2], d.w[3]);
}
// roots of a commit job + the build's counters behind them (kernels.hpp: JOB_SUMMARY_WORDS u64; a null pointer reads as 0): ONE
// buffer, one copy
__global__ void k_job_summary(TreeRef t, unsigned height, uint8_t *__restrict__ out, size_t ncols, const unsigned long long *r_ctr,
                              const unsigned long long *sd_ctr, const unsigned long long *g_ctr, DoneFlag done) {
    ZK_PRIO_SMALL();
    const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    // a batched job (proof blockIdx.y of gridDim.y): the roots of all proofs one after the other, then every proof's counters
    const unsigned z = blockIdx.z, nz = gridDim.z;
    if (t.zstride && z) {  // (plain pointers: this proof's counters sit z arenas further on)
        const size_t o = (size_t)z * t.zstride;
        if (r_ctr) r_ctr = reinterpret_cast<const unsigned long long *>(reinterpret_cast<uintptr_t>(r_ctr) + o);
        if (sd_ctr) sd_ctr = reinterpret_cast<const unsigned long long *>(reinterpret_cast<uintptr_t>(sd_ctr) + o);
        if (g_ctr) g_ctr = reinterpret_cast<const unsigned long long *>(reinterpret_cast<uintptr_t>(g_ctr) + o);
    }
    if (c < ncols) {
        const Digest d = canonical_digest(load_digest_at(node_ptr(t, c, height, 0)));  // tree form -> SHA3 bytes
        ulonglong2 *q = reinterpret_cast<ulonglong2 *>(out + ((size_t)z * ncols + c) * 32);
        q[0] = make_ulonglong2(d.w[0], d.w[1]);
        q[1] = make_ulonglong2(d.w[2], d.w[3]);
    }
    unsigned long long *cnt = reinterpret_cast<unsigned long long *>(out + (size_t)nz * ncols * 32) + (size_t)z * JOB_SUMMARY_WORDS;
    if (c == 0) {
        cnt[0] = r_ctr ? r_ctr[0] : 0;
        cnt[1] = sd_ctr ? sd_ctr[0] : 0;
        cnt[2] = sd_ctr ? sd_ctr[1] : 0;
        cnt[3] = g_ctr ? g_ctr[0] : 0;
        cnt[4] = g_ctr ? g_ctr[8] : 0;
        cnt[5] = g_ctr ? g_ctr[9] : 0;
        cnt[6] = (r_ctr ? r_ctr[10] : 0) | ((g_ctr ? g_ctr[10] : 0) << 8);
        unsigned long long constant = 0;  // hinted (R) columns whose N leaves are all equal: the eval leaves them out (EvalSkip)
        if (r_ctr)
            for (unsigned y = 0; y < t.ncols && y < 64; y++) constant += r_ctr[RUN_CHANGED + y] == 0;
        cnt[7] = constant;
    }
    if (blockIdx.x == 0 && threadIdx.x < 2 * RUN_MAX_LEVELS) {  // the longest sub-list of every level (what the next build needs)
        const unsigned l = threadIdx.x % RUN_MAX_LEVELS;
        const unsigned long long *ctr = threadIdx.x < RUN_MAX_LEVELS ? r_ctr : g_ctr;
        unsigned long long mx = 0;
        if (ctr)
            for (unsigned sub = 0; sub < RUN_SUBS; sub++) {
                const unsigned long long v = ctr[run_ctr_index(l, sub)];
                mx = v > mx ? v : mx;
            }
        cnt[8 + threadIdx.x] = mx;
    }
    signal_done(done, gridDim.x * gridDim.z);
}
// n words of device memory -> pinned host memory, by the device (no copy command, no stream wait: the host polls `done`);
// rezero: the words are zero again afterwards -- the sums of the next pass accumulate into them without a fill command.
// W = 8: u64 words; W = 4: u32 words (widened by the host)
template <int W>
__global__ __launch_bounds__(TPB) void k_publish(void *d_src, size_t n, void *h_dst, int rezero, DoneFlag done) {
    ZK_PRIO_SMALL();
    const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i < n) {
        if (W == 8) {
            unsigned long long *s = reinterpret_cast<unsigned long long *>(d_src);
            reinterpret_cast<unsigned long long *>(h_dst)[i] = s[i];
            if (rezero) s[i] = 0;
        } else {
            uint32_t *s = reinterpret_cast<uint32_t *>(d_src);
            reinterpret_cast<uint32_t *>(h_dst)[i] = s[i];
            if (rezero) s[i] = 0;
        }
    }
    signal_done(done, gridDim.x);
}
void launch_publish_u64(unsigned long long *d_src, size_t n, unsigned long long *h_dst, bool rezero, hipStream_t s, DoneFlag done) {
    hipLaunchKernelGGL(k_publish<8>, dim3((unsigned)((n + TPB - 1) / TPB)), dim3(TPB), 0, s, (void *)d_src, n, (void *)h_dst, rezero ? 1 : 0, done);
}
void launch_publish_u32(const uint32_t *d_src, size_t n, uint32_t *h_dst, hipStream_t s, DoneFlag done) {
    hipLaunchKernelGGL(k_publish<4>, dim3((unsigned)((n + TPB - 1) / TPB)), dim3(TPB), 0, s, (void *)d_src, n, (void *)h_dst, 0, done);
}

__global__ __launch_bounds__(TPB) void k_zero_counters(unsigned long long *sd, unsigned long long *r, unsigned long long *g, size_t zstride) {
    ZK_PRIO_SMALL();
    const unsigned i = blockIdx.x * TPB + threadIdx.x;
    if (zstride && blockIdx.y) {
        const size_t o = (size_t)blockIdx.y * zstride / 8;
        if (sd) sd += o;
        if (r) r += o;
        if (g) g += o;
    }
    if (i < RUN_CTR_WORDS) {
        if (r) r[i] = 0;
        if (g) g[i] = 0;
    }
    if (sd && i < 2) sd[i] = 0;
}
void launch_zero_counters(unsigned long long *d_sd_ctr, unsigned long long *d_r_ctr, unsigned long long *d_g_ctr, hipStream_t s,
                          unsigned nz, size_t zstride) {
    hipLaunchKernelGGL(k_zero_counters, dim3((RUN_CTR_WORDS + TPB - 1) / TPB, nz ? nz : 1), dim3(TPB), 0, s, d_sd_ctr, d_r_ctr, d_g_ctr,
                       nz > 1 ? zstride : (size_t)0);
}
// the columns of the proofs of a batched job into their arenas / their common table (VEC: 16-byte copies)
template <bool VEC>
__global__ __launch_bounds__(TPB) void k_gather_cols(ColSrcs srcs, size_t n, size_t src_stride, uint32_t *__restrict__ dst, size_t dst_stride,
                                                     size_t zstride) {
    const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    const uint32_t *sp = srcs.p[blockIdx.z] + (size_t)blockIdx.y * src_stride;
    uint32_t *dp = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(dst) + (size_t)blockIdx.z * zstride) + (size_t)blockIdx.y * dst_stride;
    if (VEC) {
        if (i < n / 4) reinterpret_cast<uint4 *>(dp)[i] = reinterpret_cast<const uint4 *>(sp)[i];
    } else if (i < n) {
        dp[i] = sp[i];
    }
}
void launch_gather_cols(const ColSrcs &srcs, unsigned nz, size_t ncols, size_t n, size_t src_stride, uint32_t *d_dst,
                        size_t dst_stride, size_t zstride, hipStream_t s) {
    bool vec = n % 4 == 0 && src_stride % 4 == 0 && dst_stride % 4 == 0 && zstride % 16 == 0 && aligned16(d_dst);
    for (unsigned z = 0; z < nz; z++) vec = vec && aligned16(srcs.p[z]);
    if (vec)
        hipLaunchKernelGGL(k_gather_cols<true>, dim3((unsigned)((n / 4 + TPB - 1) / TPB), (unsigned)ncols, nz), dim3(TPB), 0, s, srcs, n,
                           src_stride, d_dst, dst_stride, zstride);
    else
        hipLaunchKernelGGL(k_gather_cols<false>, dim3((unsigned)((n + TPB - 1) / TPB), (unsigned)ncols, nz), dim3(TPB), 0, s, srcs, n,
                           src_stride, d_dst, dst_stride, zstride);
}
void launch_job_summary(const TreeRef &t, unsigned height, uint8_t *d_out, size_t ncols, const unsigned long long *d_r_ctr,
                        const unsigned long long *d_sd_ctr, const unsigned long long *d_g_ctr, hipStream_t s, DoneFlag done) {
    hipLaunchKernelGGL(k_job_summary, dim3((unsigned)((ncols + 63) / 64), 1, t.nz ? t.nz : 1), dim3(64), 0, s, t, height, d_out, ncols, d_r_ctr,
                       d_sd_ctr, d_g_ctr, done);
}
void launch_gather_nodes(const uint8_t *d_tree, size_t tree_stride_nodes, size_t node, uint8_t *d_out, size_t ncols,
                         hipStream_t s) {
    hipLaunchKernelGGL(k_gather_nodes, dim3((unsigned)((ncols + 63) / 64)), dim3(64), 0, s, d_tree, tree_stride_nodes, node,
                       d_out, ncols);
}

__global__ void k_gather_first(const uint32_t *__restrict__ in, size_t stride, uint32_t *__restrict__ out,
                               size_t ncols) {
    size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < ncols) out[c] = in[c * stride];
}
void launch_gather_first(const uint32_t *d_in, size_t stride, uint32_t *d_out, size_t ncols, hipStream_t s) {
    hipLaunchKernelGGL(k_gather_first, dim3((unsigned)((ncols + 63) / 64)), dim3(64), 0, s, d_in, stride, d_out, ncols);
}

// ------------------------------------------------------------------ K9: Lasso fingerprints
// XXH3-64 (seed 0) of exactly 8 input bytes = the published 4..8-byte branch (XXH3_len_4to8_64b):
// std.hash.XxHash3.hash(0, asBytes(&h)) in src/lookups/lasso_prover.zig:213.
__device__ __forceinline__ uint64_t xxh3_64_of_u64(uint64_t h) {
    const uint64_t bitflip = 0x1cad21f72c81017cull ^ 0xdb979083e96dd4deull;  // kSecret[8..16) ^ kSecret[16..24)
    uint64_t in64 = (h >> 32) + (h << 32);  // input2 + (input1 << 32)
    uint64_t k = in64 ^ bitflip;
    k ^= rotl64(k, 49) ^ rotl64(k, 24);
    k *= 0x9FB21C651E98DF25ull;
    k ^= (k >> 35) + 8;
    k *= 0x9FB21C651E98DF25ull;
    return k ^ (k >> 28);
}

__global__ __launch_bounds__(TPB) void k_lasso_fingerprints(const uint32_t *__restrict__ rows, size_t n_rows,
                                                            size_t width, uint32_t *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n_rows) return;
    uint64_t h = 0;
    for (size_t f = 0; f < width; f++) {
        h ^= rows[i * width + f];
        h = xxh3_64_of_u64(h);
    }
    out[i] = (uint32_t)(h % (uint64_t)P);
}

void launch_lasso_fingerprints(const uint32_t *d_rows, size_t rows, size_t width, uint32_t *d_out, hipStream_t s,
                               const KTime *kt) {
    if (rows)
        ZK_LAUNCH(kt, k_lasso_fingerprints, dim3((unsigned)((rows + TPB - 1) / TPB)), dim3(TPB), 0, s, d_rows, rows, width,
                  d_out);
}

// ------------------------------------------------------------------ synthetic data for the measurement hooks
__global__ __launch_bounds__(TPB) void k_fill_pattern(uint32_t *__restrict__ out, size_t n, uint32_t seed) {
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    const size_t step = (size_t)gridDim.x * TPB;
    for (; i < n; i += step) {
        uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull + seed;  // splitmix64 finaliser
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        out[i] = (uint32_t)(z % (uint64_t)P);
    }
}
void launch_fill_pattern(uint32_t *d_out, size_t n, uint32_t seed, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_fill_pattern, dim3(stream_grid(n)), dim3(TPB), 0, s, d_out, n, seed);
}

}  // namespace zk
