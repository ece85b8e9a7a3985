// Locality ordering of a symmetric contact matrix (gfx950).
//
// Why: the KR SpMV and the Louvain neighbour scan gather one 8-byte (4-byte) vector element per
// stored entry.  With contig ids in assembly order those gathers are random over the whole vector:
// every one costs an L2->L1 line fill for 8 useful bytes and the kernels run at the L2 fill rate
// (measured 0.99 TB/s algorithmic at 1M contigs / 201M entries), not at the HBM rate.  Contacts are
// overwhelmingly intra-genome, so relabelling contigs such that strongly linked ones sit next to each
// other turns the gathers into L1 hits.
//
// How: three Boruvka rounds.  Round 1: every contig points at its heaviest off-diagonal neighbour (one streaming
// pass, no gathers); mutual pairs become roots and pointer jumping labels each tree.  Rounds 2 and 3: every cluster
// picks the heaviest edge that leaves it (per-cluster 64-bit atomicMax on (float weight, ~target cluster): order
// independent, ties to the lowest target) and the clusters merge the same way.  Contigs are then ordered by
// (cluster_3, cluster_2, cluster_1, id), so every level of the merge tree is contiguous.  On the C2-like map this puts
// 71 % of the entries within 512 positions of the diagonal (ideal genome order: 78 %; one round: 21 %).
// The order only affects speed: all results are returned in the caller's numbering.
#include "common.hpp"

#include <algorithm>
#include <cstdlib>
#include <hipcub/hipcub.hpp>

int b3c_exclusive_scan_i64(b3c_ctx *ctx, const int64_t *in, int64_t *out, int64_t count);

namespace {

constexpr int BLK = 256;

// heaviest off-diagonal neighbour of every row (ties -> lowest column); rows without one point at themselves
template <typename T, int LPR>
__global__ void __launch_bounds__(BLK) k_heaviest(int64_t r0, int64_t n, const int64_t *__restrict__ indptr,
                                                   const int32_t *__restrict__ indices, const T *__restrict__ a,
                                                   int32_t *__restrict__ parent)
{
    // rows [r0, n): all rows, or the rows of one rank of the sharded path
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR, sl = lane % LPR;
    const int64_t wave = (int64_t)blockIdx.x * (BLK / 64) + (threadIdx.x >> 6);
    const int64_t nwave = (int64_t)gridDim.x * (BLK / 64);
    for (int64_t row = r0 + wave * RPW + sub; row < n; row += nwave * RPW) {
        const int64_t s = indptr[row], e = indptr[row + 1];
        double best = -1.0;
        int32_t bj = (int32_t)row;
        int64_t k = s + sl;
        for (; k + 3 * LPR < e; k += 4 * LPR) {
            int32_t j0 = indices[k], j1 = indices[k + LPR], j2 = indices[k + 2 * LPR], j3 = indices[k + 3 * LPR];
            double v0 = (double)a[k], v1 = (double)a[k + LPR], v2 = (double)a[k + 2 * LPR], v3 = (double)a[k + 3 * LPR];
            if (j0 != (int32_t)row && v0 > best) { best = v0; bj = j0; }
            if (j1 != (int32_t)row && v1 > best) { best = v1; bj = j1; }
            if (j2 != (int32_t)row && v2 > best) { best = v2; bj = j2; }
            if (j3 != (int32_t)row && v3 > best) { best = v3; bj = j3; }
        }
        for (; k < e; k += LPR) {
            int32_t j = indices[k];
            double v = (double)a[k];
            if (j != (int32_t)row && v > best) { best = v; bj = j; }
        }
        // columns ascend along k, so within a lane the first maximum has the lowest column; across lanes
        // prefer the larger value, then the lower column
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) {
            double ob = __shfl_down(best, o, LPR);
            int32_t oj = __shfl_down(bj, o, LPR);
            if (ob > best || (ob == best && oj < bj)) { best = ob; bj = oj; }
        }
        if (sl == 0) parent[row] = best >= 0.0 ? bj : (int32_t)row;
    }
}

// mutual pairs: the smaller id becomes the root
__global__ void k_break_pairs(int64_t n, const int32_t *__restrict__ parent, int32_t *__restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t j = parent[i];
    out[i] = (parent[j] == (int32_t)i && (int32_t)i < j) ? (int32_t)i : j;
}

__global__ void k_jump(int64_t n, const int32_t *__restrict__ in, int32_t *__restrict__ out, int *__restrict__ changed)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int c = 0;
    if (i < n) {
        const int32_t p = in[i];
        const int32_t g = in[p];
        out[i] = g;
        c = g != p;
    }
    // a plain store: every writer stores the same value (16k waves queueing on one atomic cost 180 us at n = 10^6)
    if (__ballot(c) && (threadIdx.x & 63) == 0) *changed = 1;
}

// Boruvka round: heaviest edge leaving the cluster of every node, folded per cluster with a 64-bit
// atomicMax on (float bits of the weight, ~target cluster) -- order independent, ties to the lowest target.
template <typename T, int LPR>
__global__ void __launch_bounds__(BLK) k_best_external(int64_t r0, int64_t n, const int64_t *__restrict__ indptr,
                                                        const int32_t *__restrict__ indices, const T *__restrict__ a,
                                                        const int32_t *__restrict__ label,
                                                        unsigned long long *__restrict__ cbest,
                                                        unsigned long long *__restrict__ rowbest, int sample)
{
    // rows [r0, n).  rowbest != nullptr (sharded path): the best key of every row is stored per ROW instead of being
    // folded into its cluster here -- the rows of a cluster live on several ranks; k_fold_rowbest folds the gathered keys
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR, sl = lane % LPR;
    const int64_t wave = (int64_t)blockIdx.x * (BLK / 64) + (threadIdx.x >> 6);
    const int64_t nwave = (int64_t)gridDim.x * (BLK / 64);
    for (int64_t row = r0 + wave * RPW + sub; row < n; row += nwave * RPW) {
        const int64_t s = indptr[row];
        int64_t e = indptr[row + 1];
        if (sample > 0 && e - s > sample) e = s + sample;     // the first entries of a row stand for all of them (B3C_ORDER_SAMPLE)
        const int32_t li = label[row];
        unsigned long long best = 0;
        if (e - s <= 4 * LPR && sample == 0) {                  // (B3C_ORDER_SAMPLE=-1 forces the full-gather loop below: the A/B of the test)
            // The whole row sits in the group's registers (round 3).  What bounds this kernel is one random 4-byte label gather
            // per stored entry (2.0e8 of them at C3, twice per step); but the heaviest entry LEAVING the cluster is almost
            // always among the few heaviest entries of the row -- the clusters are small at this point.  So: T = the 4th
            // largest DISTINCT weight of the row (weights only, no gathers), labels are gathered for the entries of weight
            // >= T alone, and if one of them leaves the cluster the best such entry is the row's answer: every entry outside
            // the set is lighter, and all entries of equal weight are inside it, so the (weight, lowest label) tie-break sees
            // the same candidates.  Only if none of them leaves the cluster are the remaining labels fetched.  Same result
            // bit for bit, a fraction of the gathers.
            int32_t j[4];
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int64_t k = s + q * LPR + sl;
                const bool ok = k < e;
                const int64_t kc = ok ? k : s;
                j[q] = ok ? indices[kc] : (int32_t)row;
                v[q] = ok ? (float)a[kc] : 0.f;
            }
            float thr = INFINITY;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float m = 0.f;
#pragma unroll
                for (int q = 0; q < 4; q++) m = (v[q] < thr && v[q] > m) ? v[q] : m;
#pragma unroll
                for (int o = LPR / 2; o > 0; o >>= 1) { const float t = __shfl_xor(m, o, LPR); m = t > m ? t : m; }
                thr = m;                                    // 0 once the row has fewer distinct positive weights: the set is the whole row
            }
            int32_t l[4];
#pragma unroll
            for (int q = 0; q < 4; q++) l[q] = (v[q] >= thr && v[q] > 0.f) ? label[j[q]] : li;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const unsigned long long c = ((unsigned long long)__float_as_uint(v[q]) << 32) | (0xFFFFFFFFu - (uint32_t)l[q]);
                if (l[q] != li && v[q] > 0.f && c > best) best = c;
            }
            // did any lane of the group find an entry that leaves the cluster?
            unsigned long long any = best;
#pragma unroll
            for (int o = LPR / 2; o > 0; o >>= 1) any |= (unsigned long long)__shfl_xor((long long)any, o, LPR);
            if (any == 0 && thr > 0.f) {
#pragma unroll
                for (int q = 0; q < 4; q++) l[q] = (v[q] < thr && v[q] > 0.f) ? label[j[q]] : li;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const unsigned long long c = ((unsigned long long)__float_as_uint(v[q]) << 32) | (0xFFFFFFFFu - (uint32_t)l[q]);
                    if (l[q] != li && v[q] > 0.f && c > best) best = c;
                }
            }
        } else
        // four masked slots per lane and pass: all index loads, then all label gathers are in flight together (a slot past
        // the end of the row points at the row itself, whose label is li, so it never wins)
        for (int64_t base = s; base < e; base += 4 * LPR) {
            int32_t j[4];
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int64_t k = base + q * LPR + sl;
                const bool ok = k < e;
                const int64_t kc = ok ? k : s;
                j[q] = ok ? indices[kc] : (int32_t)row;
                v[q] = ok ? (float)a[kc] : 0.f;
            }
            int32_t l[4];
#pragma unroll
            for (int q = 0; q < 4; q++) l[q] = label[j[q]];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const unsigned long long c = ((unsigned long long)__float_as_uint(v[q]) << 32) | (0xFFFFFFFFu - (uint32_t)l[q]);
                if (l[q] != li && v[q] > 0.f && c > best) best = c;
            }
        }
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) {
            unsigned long long ob = (unsigned long long)__shfl_down((long long)best, o, LPR);
            best = ob > best ? ob : best;
        }
        if (sl == 0) {
            if (rowbest) rowbest[row] = best;
            else if (best) atomicMax(&cbest[li], best);
        }
    }
}

__global__ void k_fold_rowbest(int64_t n, const unsigned long long *__restrict__ rowbest, const int32_t *__restrict__ label,
                               unsigned long long *__restrict__ cbest)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long b = rowbest[i];
    if (b) atomicMax(&cbest[label[i]], b);
}

__global__ void k_cluster_ptr(int64_t n, const unsigned long long *__restrict__ cbest, int32_t *__restrict__ cptr)
{
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n) {
        const unsigned long long key = cbest[c];
        cptr[c] = key ? (int32_t)(0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFu)) : (int32_t)c;
    }
}

__global__ void k_relabel(int64_t n, int32_t *__restrict__ label, const int32_t *__restrict__ root)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) label[i] = root[label[i]];
}

__global__ void k_iota(int64_t n, int32_t *__restrict__ v)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = (int32_t)i;
}

// key of row i = its cluster labels of all rounds, the last round in the highest bits; value = i
__global__ void k_pack_labels(int64_t n, int rounds, int bits, const int32_t *__restrict__ lab, unsigned long long *__restrict__ key,
                              int32_t *__restrict__ val)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long k = 0;
    for (int r = rounds - 1; r >= 0; r--) k = (k << bits) | (unsigned long long)(uint32_t)lab[(size_t)r * n + i];
    key[i] = k;
    val[i] = (int32_t)i;
}

__global__ void k_gather_i32(int64_t n, const int32_t *__restrict__ src, const int32_t *__restrict__ idx,
                             int32_t *__restrict__ dst)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

__global__ void k_invert_perm(int64_t n, const int32_t *__restrict__ new2old, int32_t *__restrict__ old2new)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) old2new[new2old[k]] = (int32_t)k;
}

__global__ void k_new_degrees(int64_t n, int64_t row0, int64_t row1, const int64_t *__restrict__ indptr,
                              const int32_t *__restrict__ new2old, int64_t *__restrict__ deg)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) {
        const int32_t o = new2old[k];
        deg[k] = k >= row0 && k < row1 ? indptr[o + 1] - indptr[o] : 0;   // rows outside the block stay empty
    }
    if (k == n) deg[k] = 0;
}

__global__ void __launch_bounds__(BLK) k_perm_keys(int64_t n, const int64_t *__restrict__ indptr,
                                                    const int32_t *__restrict__ indices,
                                                    const int32_t *__restrict__ old2new, int64_t *__restrict__ keys)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (BLK / 64) + (threadIdx.x >> 6);
    const int64_t nwave = (int64_t)gridDim.x * (BLK / 64);
    for (int64_t row = wave; row < n; row += nwave) {
        const int64_t s = indptr[row], e = indptr[row + 1];
        const int64_t nr = (int64_t)old2new[row] * n;
        for (int64_t k = s + lane; k < e; k += 64) keys[k] = nr + old2new[indices[k]];
    }
}

// rows moved, columns relabelled, order inside a row kept as it was (i.e. no longer ascending)
template <typename T>
__global__ void __launch_bounds__(BLK) k_permute_rows(int64_t n, const int64_t *__restrict__ indptr,
                                                       const int32_t *__restrict__ indices, const T *__restrict__ a,
                                                       const int32_t *__restrict__ new2old,
                                                       const int32_t *__restrict__ old2new,
                                                       const int64_t *__restrict__ out_indptr,
                                                       int32_t *__restrict__ out_indices, T *__restrict__ out_a)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (BLK / 64) + (threadIdx.x >> 6);
    const int64_t nwave = (int64_t)gridDim.x * (BLK / 64);
    for (int64_t k = wave; k < n; k += nwave) {
        const int32_t old = new2old[k];
        const int64_t s = indptr[old];
        const int len = (int)(indptr[old + 1] - s);
        const int64_t o = out_indptr[k];
        if (out_indptr[k + 1] == o) continue;             // empty in the output: an empty row, or a row of another rank's block
        for (int base = 0; base < len; base += 256) {
            int32_t c[4];
            T v[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int e = base + t * 64 + lane;
                const bool ok = e < len;
                c[t] = ok ? indices[s + e] : 0;
                v[t] = ok ? a[s + e] : T(0);
            }
#pragma unroll
            for (int t = 0; t < 4; t++) c[t] = old2new[c[t]];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int e = base + t * 64 + lane;
                if (e < len) { out_indices[o + e] = c[t]; out_a[o + e] = v[t]; }
            }
        }
    }
}

// Register form of the packed sort: lane l holds the R keys of positions l*R .. l*R+R-1, so the compare-exchange
// stages with distance < R stay inside a lane and the others are one cross-lane shuffle per key -- 84 LDS-pipe operations
// for a 256-entry row against 576 for the in-LDS network above, which is bound by exactly that pipe (C3: 4.6 -> see
// DESIGN.md).  Loads and stores are R consecutive entries per lane.
template <int R>
__device__ __forceinline__ void bitonic_regs(uint32_t (&k)[R], int lane)
{
    constexpr int N = 64 * R;
#pragma unroll
    for (int kk = 2; kk <= N; kk <<= 1) {
#pragma unroll
        for (int jj = kk >> 1; jj > 0; jj >>= 1) {
            if (jj < R) {
#pragma unroll
                for (int r = 0; r < R; r++) {
                    if ((r & jj) == 0) {
                        const bool asc = (((lane * R + r) & kk) == 0);
                        const uint32_t a = k[r], b = k[r | jj];
                        const bool sw = (a > b) == asc;
                        k[r] = sw ? b : a;
                        k[r | jj] = sw ? a : b;
                    }
                }
            } else {
                const int lx = jj / R;
                const bool lower = (lane & lx) == 0;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const uint32_t other = (uint32_t)__shfl_xor((int)k[r], lx, 64);
                    const bool asc = (((lane * R + r) & kk) == 0);
                    const uint32_t mn = k[r] < other ? k[r] : other, mx = k[r] < other ? other : k[r];
                    k[r] = (lower == asc) ? mn : mx;
                }
            }
        }
    }
}

template <typename T, int R>
__device__ __forceinline__ void permute_row_regs(int lane, int pb, int len, const int32_t *__restrict__ ci,
                                                 const T *__restrict__ av, const int32_t *__restrict__ old2new,
                                                 int32_t *__restrict__ oi, T *__restrict__ oa)
{
    const uint32_t pmask = (1u << pb) - 1u;
    uint32_t k[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int e = lane * R + r;
        k[r] = e < len ? ((uint32_t)old2new[ci[e]] << pb) | (uint32_t)e : 0xFFFFFFFFu;
    }
    bitonic_regs<R>(k, lane);
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int e = lane * R + r;
        if (e < len) {
            oi[e] = (int32_t)(k[r] >> pb);
            oa[e] = av[k[r] & pmask];
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(BLK) k_permute_rows_regs(int64_t n, int pb, int max_len, const int64_t *__restrict__ indptr,
                                                            const int32_t *__restrict__ indices, const T *__restrict__ a,
                                                            const int32_t *__restrict__ new2old,
                                                            const int32_t *__restrict__ old2new,
                                                            const int64_t *__restrict__ out_indptr,
                                                            int32_t *__restrict__ out_indices, T *__restrict__ out_a)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (BLK / 64) + (threadIdx.x >> 6);
    const int64_t nwave = (int64_t)gridDim.x * (BLK / 64);
    for (int64_t k = wave; k < n; k += nwave) {
        const int32_t old = new2old[k];
        const int64_t s = indptr[old];
        const int len = (int)(indptr[old + 1] - s);
        const int64_t o = out_indptr[k];
        if (out_indptr[k + 1] == o) continue;             // empty in the output: an empty row, or a row of another rank's block
        if (len > max_len) continue;                      // long rows: k_long_keys / k_long_scatter
        // wave-uniform choice of the network size
        if (len <= 64) permute_row_regs<T, 1>(lane, pb, len, indices + s, a + s, old2new, out_indices + o, out_a + o);
        else if (len <= 128) permute_row_regs<T, 2>(lane, pb, len, indices + s, a + s, old2new, out_indices + o, out_a + o);
        else if (len <= 256) permute_row_regs<T, 4>(lane, pb, len, indices + s, a + s, old2new, out_indices + o, out_a + o);
        else permute_row_regs<T, 8>(lane, pb, len, indices + s, a + s, old2new, out_indices + o, out_a + o);
    }
}

// ---- rows too long for the register network: their entries (and only theirs) go through a radix sort ----
// flags / entry counts of the long NEW rows
__global__ void k_long_flags(int64_t n, int max_len, const int64_t *__restrict__ out_indptr, int64_t *__restrict__ rflag,
                             int64_t *__restrict__ ecnt)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k > n) return;
    const int64_t len = k < n ? out_indptr[k + 1] - out_indptr[k] : 0;
    rflag[k] = len > max_len ? 1 : 0;
    ecnt[k] = len > max_len ? len : 0;
}

// key = (index among the long rows) << 32 | new column, payload = position of the entry in its old row
__global__ void __launch_bounds__(BLK) k_long_keys(int64_t n, const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                    const int32_t *__restrict__ new2old, const int32_t *__restrict__ old2new,
                                                    const int64_t *__restrict__ rflag, const int64_t *__restrict__ rpos,
                                                    const int64_t *__restrict__ epos, int32_t *__restrict__ long_rows,
                                                    unsigned long long *__restrict__ keys, uint32_t *__restrict__ vals)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (BLK / 64) + (threadIdx.x >> 6);
    const int64_t nwave = (int64_t)gridDim.x * (BLK / 64);
    for (int64_t k = wave; k < n; k += nwave) {
        if (!rflag[k]) continue;
        const int32_t old = new2old[k];
        const int64_t s = indptr[old], len = indptr[old + 1] - s;
        const unsigned long long hi = (unsigned long long)rpos[k] << 32;
        if (lane == 0) long_rows[rpos[k]] = (int32_t)k;
        const int64_t o = epos[k];
        for (int64_t e = lane; e < len; e += 64) {
            keys[o + e] = hi | (uint32_t)old2new[indices[s + e]];
            vals[o + e] = (uint32_t)e;
        }
    }
}

template <typename T>
__global__ void k_long_scatter(int64_t total, const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ vals,
                               const int32_t *__restrict__ long_rows, const int64_t *__restrict__ epos,
                               const int64_t *__restrict__ indptr, const T *__restrict__ a, const int32_t *__restrict__ new2old,
                               const int64_t *__restrict__ out_indptr, int32_t *__restrict__ out_indices, T *__restrict__ out_a)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const unsigned long long key = keys[t];
    const int32_t k = long_rows[key >> 32];
    const int64_t dst = out_indptr[k] + (t - epos[k]);       // the sorted stream is the long rows one after the other
    out_indices[dst] = (int32_t)(key & 0xFFFFFFFFull);
    out_a[dst] = a[indptr[new2old[k]] + vals[t]];
}

__global__ void k_max_len(int64_t n, const int64_t *__restrict__ indptr, unsigned long long *__restrict__ out)
{
    max_row_len_body(n, indptr, (unsigned long long *)out);
}

__global__ void k_cols_from_keys(int64_t nnz, int64_t n, const int64_t *__restrict__ keys, int32_t *__restrict__ indices)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += stride) indices[k] = (int32_t)(keys[k] % n);
}

}  // namespace

// pointer jumping to the roots of a forest given as parent pointers (mutual pairs broken first)
static int forest_roots(b3c_ctx *ctx, int64_t n, int32_t *p0, int32_t *p1, int *changed, int32_t **roots)
{
    hipStream_t s = ctx->stream;
    const unsigned gn = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_break_pairs, dim3(gn), dim3(256), 0, s, n, p0, p1);
    int32_t *cur = p1, *nxt = p0;
    for (int round = 0; round < 64; round++) {
        HIP_TRY(ctx, hipMemsetAsync(changed, 0, sizeof(int), s));
        hipLaunchKernelGGL(k_jump, dim3(gn), dim3(256), 0, s, n, cur, nxt, changed);
        {                     // a second jump (idempotent once converged) is cheaper than a host round trip
            hipLaunchKernelGGL(k_jump, dim3(gn), dim3(256), 0, s, n, nxt, cur, changed);
            int32_t *t2 = cur; cur = nxt; nxt = t2;
        }
        int h = 0;
        { const int rb_ = b3c_readback(ctx, &h, changed, sizeof(int)); if (rb_) return rb_; }
        int32_t *t = cur; cur = nxt; nxt = t;
        if (!h) break;
    }
    *roots = cur;
    return B3C_OK;
}

// new2old / old2new: DEVICE int32[n].  Works for B3C_F64 and B3C_U32 matrices.
// Three Boruvka rounds (heaviest neighbour, then twice heaviest edge leaving the cluster); contigs are
// ordered by (cluster_3, cluster_2, cluster_1, id) so that every level of the merge tree is contiguous.
int b3c_locality_order_dev(b3c_ctx *ctx, const b3c_mat *a, int32_t *new2old, int32_t *old2new)
{
    hipStream_t s = ctx->stream;
    const int64_t n = a->n;
    if (n == 0) return B3C_OK;
    constexpr int ROUNDS = 3;         // (2 / 3 / 4 rounds at C3: whole step 77.3 / 73.0 / 73.6 ms in round 2; two rounds: ordering -1.7 ms, SpMV +4.3 ms)
    const int order_sample = (int)CFG.num("B3C_ORDER_SAMPLE", 0);
    DevBuf<int32_t> p0(ctx, n), p1(ctx, n), lab(ctx, (size_t)n * ROUNDS), kin(ctx, n), kout(ctx, n), vout(ctx, n);
    DevBuf<unsigned long long> cbest(ctx, n);
    DevBuf<int> changed(ctx, 1);
    if (!p0.p || !p1.p || !lab.p || !kin.p || !kout.p || !vout.p || !cbest.p || !changed.p)
        return ctx->fail(B3C_E_NOMEM, "b3c_locality_order");
    const double avg = (double)a->nnz / (double)n;
    // Sharded path (b3c_ctx_set_dist): the two passes over the matrix -- heaviest neighbour of every row, heaviest edge
    // leaving its cluster -- run on this rank's rows only; their per-row results are all-gathered (4 and 8 bytes per row
    // through the label and vector exchange buffers) and everything after them (cluster folds, pointer jumping, the
    // sorts) is n-vector work every rank repeats.  Per-row results and max-folds do not depend on who computed them: the
    // order is bit for bit the single-GPU one.
    const b3c_dist &D = ctx->dist;
    const bool dist = ctx->has_dist && D.world > 1 && (int64_t)D.world * D.chunk >= n;
    int64_t r0 = 0, r1 = n;
    if (dist) ctx->dist_rows(n, D.rank, &r0, &r1);
    const int g16 = grid_for(r1 - r0, (BLK / 64) * 4, 16 * ctx->n_cu), g64 = grid_for(r1 - r0, BLK / 64, 16 * ctx->n_cu);
    const unsigned gn = (unsigned)((n + 255) / 256);
    const bool wide = avg >= 48;
    int32_t *roots = nullptr;
    int rc;
    for (int r = 0; r < ROUNDS; r++) {
        int32_t *cur = lab.p + (size_t)r * n;
        if (r == 0) {
            int32_t *par = dist ? D.labels : p0.p;         // sharded: straight into this rank's slice of the exchange buffer
            if (a->kind == B3C_F64) {
                if (wide) hipLaunchKernelGGL((k_heaviest<double, 64>), dim3(g64), dim3(BLK), 0, s, r0, r1, a->indptr, a->indices, (const double *)a->data, par);
                else hipLaunchKernelGGL((k_heaviest<double, 16>), dim3(g16), dim3(BLK), 0, s, r0, r1, a->indptr, a->indices, (const double *)a->data, par);
            } else {
                if (wide) hipLaunchKernelGGL((k_heaviest<uint32_t, 64>), dim3(g64), dim3(BLK), 0, s, r0, r1, a->indptr, a->indices, (const uint32_t *)a->data, par);
                else hipLaunchKernelGGL((k_heaviest<uint32_t, 16>), dim3(g16), dim3(BLK), 0, s, r0, r1, a->indptr, a->indices, (const uint32_t *)a->data, par);
            }
            if (dist) {
                if ((rc = b3c_dist_exchange(ctx, B3C_BUF_LABELS))) return rc;
                HIP_TRY(ctx, hipMemcpyAsync(p0.p, D.labels, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, s));
            }
            if ((rc = forest_roots(ctx, n, p0.p, p1.p, changed.p, &roots))) return rc;
            HIP_TRY(ctx, hipMemcpyAsync(cur, roots, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, s));
        } else {
            const int32_t *prev = lab.p + (size_t)(r - 1) * n;
            HIP_TRY(ctx, hipMemsetAsync(cbest.p, 0, sizeof(unsigned long long) * (size_t)n, s));
            unsigned long long *rowbest = dist ? (unsigned long long *)D.vec : nullptr;   // 8-byte slots, one per row
            if (a->kind == B3C_F64) {
                if (wide) hipLaunchKernelGGL((k_best_external<double, 64>), dim3(g64), dim3(BLK), 0, s, r0, r1, a->indptr, a->indices, (const double *)a->data, prev, cbest.p, rowbest, order_sample);
                else hipLaunchKernelGGL((k_best_external<double, 16>), dim3(g16), dim3(BLK), 0, s, r0, r1, a->indptr, a->indices, (const double *)a->data, prev, cbest.p, rowbest, order_sample);
            } else {
                if (wide) hipLaunchKernelGGL((k_best_external<uint32_t, 64>), dim3(g64), dim3(BLK), 0, s, r0, r1, a->indptr, a->indices, (const uint32_t *)a->data, prev, cbest.p, rowbest, order_sample);
                else hipLaunchKernelGGL((k_best_external<uint32_t, 16>), dim3(g16), dim3(BLK), 0, s, r0, r1, a->indptr, a->indices, (const uint32_t *)a->data, prev, cbest.p, rowbest, order_sample);
            }
            if (dist) {
                if ((rc = b3c_dist_exchange(ctx, B3C_BUF_VEC))) return rc;
                hipLaunchKernelGGL(k_fold_rowbest, dim3(gn), dim3(256), 0, s, n, rowbest, prev, cbest.p);
            }
            hipLaunchKernelGGL(k_cluster_ptr, dim3(gn), dim3(256), 0, s, n, cbest.p, p0.p);
            if ((rc = forest_roots(ctx, n, p0.p, p1.p, changed.p, &roots))) return rc;
            HIP_TRY(ctx, hipMemcpyAsync(cur, prev, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, s));
            hipLaunchKernelGGL(k_relabel, dim3(gn), dim3(256), 0, s, n, cur, roots);
        }
    }
    int bits = 1;
    while ((1ll << bits) < n) bits++;
    if (ROUNDS * bits <= 64) {
        // ONE stable sort on the composite key (cluster_R, ..., cluster_2, cluster_1): the order the LSD passes below give, from a
        // third of the launches (1M keys go through rocPRIM's merge sort, ~20 small kernels per call; n up to 2^21 with three rounds)
        DevBuf<unsigned long long> k64(ctx, n), k64o(ctx, n);
        if (!k64.p || !k64o.p) return ctx->fail(B3C_E_NOMEM, "b3c_locality_order: sort keys");
        hipLaunchKernelGGL(k_pack_labels, dim3(gn), dim3(256), 0, s, n, ROUNDS, bits, lab.p, k64.p, vout.p);
        size_t tb = 0;
        HIP_TRY(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tb, k64.p, k64o.p, vout.p, new2old, (int64_t)n, 0, ROUNDS * bits, s));
        DevBuf<char> tmp(ctx, tb);
        if (!tmp.p) return ctx->fail(B3C_E_NOMEM, "b3c_locality_order: sort workspace");
        HIP_TRY(ctx, hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, k64.p, k64o.p, vout.p, new2old, (int64_t)n, 0, ROUNDS * bits, s));
        HIP_TRY(ctx, hipStreamSynchronize(s));        // the scratch above goes back to the pool here
    } else {
    // LSD: stable sorts by cluster_1, cluster_2, cluster_3 starting from the identity
    hipLaunchKernelGGL(k_iota, dim3(gn), dim3(256), 0, s, n, new2old);
    size_t tb = 0;
    HIP_TRY(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tb, kin.p, kout.p, new2old, vout.p, (int64_t)n, 0, bits, s));
    DevBuf<char> tmp(ctx, tb);
    if (!tmp.p) return ctx->fail(B3C_E_NOMEM, "b3c_locality_order: sort workspace");
    for (int r = 0; r < ROUNDS; r++) {
        hipLaunchKernelGGL(k_gather_i32, dim3(gn), dim3(256), 0, s, n, lab.p + (size_t)r * n, new2old, kin.p);
        HIP_TRY(ctx, hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, kin.p, kout.p, new2old, vout.p, (int64_t)n, 0, bits, s));
        HIP_TRY(ctx, hipMemcpyAsync(new2old, vout.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, s));
    }
    }
    hipLaunchKernelGGL(k_invert_perm, dim3(gn), dim3(256), 0, s, n, new2old, old2new);
    HIP_TRY(ctx, hipGetLastError());
    return B3C_OK;
}

// Symmetric permutation P A P^T into a new canonical CSR (columns sorted in the new numbering).
// Rows outside [row0, row1) of the result stay empty (the row block of one rank of the sharded path).
int b3c_permute_csr_rows(b3c_ctx *ctx, const b3c_mat *a, const int32_t *new2old, const int32_t *old2new, b3c_mat **out,
                         bool sort_cols, int64_t row0, int64_t row1)
{
    hipStream_t s = ctx->stream;
    const int64_t n = a->n;
    const bool block = row0 > 0 || row1 < n;
    DevBuf<int64_t> deg(ctx, n + 1), optr(ctx, n + 1);
    if (!deg.p || !optr.p) return ctx->fail(B3C_E_NOMEM, "b3c_permute_csr");
    hipLaunchKernelGGL(k_new_degrees, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, s, n, row0, row1, a->indptr, new2old, deg.p);
    int rc;
    if ((rc = b3c_exclusive_scan_i64(ctx, deg.p, optr.p, n + 1))) return rc;
    int64_t nnz = a->nnz;
    if (block) { const int rb_ = b3c_readback(ctx, &nnz, optr.p + n, sizeof(int64_t)); if (rb_) return rb_; }
    MatOwner own(ctx);            // freed on every early return below
    if ((rc = b3c_mat_alloc(ctx, n, nnz, a->kind, &own.m))) return rc;
    b3c_mat *m = own.m;
    HIP_TRY(ctx, hipMemcpyAsync(m->indptr, optr.p, sizeof(int64_t) * (size_t)(n + 1), hipMemcpyDeviceToDevice, s));
    m->sorted_cols = a->sorted_cols && sort_cols;
    // canonical output: per-row LDS sort when every row fits, else the global radix sort below
    unsigned long long maxlen = 0;
    if (nnz && sort_cols) {
        DevBuf<unsigned long long> ml(ctx, 1);
        if (!ml.p) { return ctx->fail(B3C_E_NOMEM, "b3c_permute_csr"); }
        HIP_TRY(ctx, hipMemsetAsync(ml.p, 0, sizeof(unsigned long long), s));
        hipLaunchKernelGGL(k_max_len, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 2048)), dim3(256), 0, s, n, a->indptr, ml.p);
        { const int rb_ = b3c_readback(ctx, &maxlen, ml.p, sizeof(maxlen)); if (rb_) return rb_; }
    }
    constexpr int CAPR = 512;
    int nb = 1;
    while ((1ll << nb) < n) nb++;
    const bool regs_ok = nnz && sort_cols && nb + 9 <= 32;
    if (regs_ok) {
        // rows of up to CAPR entries: bitonic network in registers on packed (new column, position) keys ...
        const int g = grid_for(n, BLK / 64, 16 * ctx->n_cu);
        int pb = 6;
        while ((1ull << pb) < std::min<unsigned long long>(maxlen, CAPR)) pb++;
        if (a->kind == B3C_F64)
            hipLaunchKernelGGL(k_permute_rows_regs<double>, dim3(g), dim3(BLK), 0, s, n, pb, CAPR, a->indptr, a->indices,
                               (const double *)a->data, new2old, old2new, m->indptr, m->indices, (double *)m->data);
        else
            hipLaunchKernelGGL(k_permute_rows_regs<uint32_t>, dim3(g), dim3(BLK), 0, s, n, pb, CAPR, a->indptr, a->indices,
                               (const uint32_t *)a->data, new2old, old2new, m->indptr, m->indices, (uint32_t *)m->data);
        if (maxlen > CAPR) {
            // ... and the longer ones (hubs) through ONE radix sort of just their entries: a single long row no longer sends
            // the whole matrix down the global-sort path
            DevBuf<int64_t> rflag(ctx, n + 1), ecnt(ctx, n + 1), rpos(ctx, n + 2), epos(ctx, n + 2);
            if (!rflag.p || !ecnt.p || !rpos.p || !epos.p) { return ctx->fail(B3C_E_NOMEM, "b3c_permute_csr: long rows"); }
            hipLaunchKernelGGL(k_long_flags, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, s, n, CAPR, m->indptr, rflag.p, ecnt.p);
            if ((rc = b3c_exclusive_scan_i64(ctx, rflag.p, rpos.p, n + 1)) || (rc = b3c_exclusive_scan_i64(ctx, ecnt.p, epos.p, n + 1))) {
                return rc;
            }
            int64_t tot[2] = {0, 0};
            { const int rb_ = b3c_readback(ctx, &tot[0], rpos.p + n, sizeof(int64_t)); if (rb_) return rb_; }
            { const int rb_ = b3c_readback(ctx, &tot[1], epos.p + n, sizeof(int64_t)); if (rb_) return rb_; }
            const int64_t n_long = tot[0], n_ent = tot[1];
            // (a row block: the long rows may all sit in other ranks' blocks -- nothing to sort here)
            if (n_ent > 0) {
            if (n_ent >= (1ll << 31)) { return ctx->fail(B3C_E_ARG, "b3c_permute_csr: more than 2^31 entries in long rows"); }
            DevBuf<int32_t> long_rows(ctx, n_long + 1);
            DevBuf<unsigned long long> k_in(ctx, n_ent + 1), k_out(ctx, n_ent + 1);
            DevBuf<uint32_t> v_in(ctx, n_ent + 1), v_out(ctx, n_ent + 1);
            if (!long_rows.p || !k_in.p || !k_out.p || !v_in.p || !v_out.p) { return ctx->fail(B3C_E_NOMEM, "b3c_permute_csr: long rows"); }
            hipLaunchKernelGGL(k_long_keys, dim3(grid_for(n, BLK / 64, 16 * ctx->n_cu)), dim3(BLK), 0, s, n, a->indptr, a->indices, new2old,
                               old2new, rflag.p, rpos.p, epos.p, long_rows.p, k_in.p, v_in.p);
            int end_bit = 33;
            while (end_bit < 64 && (1ll << (end_bit - 32)) < n_long) end_bit++;
            size_t tb = 0;
            hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, tb, k_in.p, k_out.p, v_in.p, v_out.p, (int64_t)n_ent, 0, end_bit, s);
            if (e != hipSuccess) { return ctx->fail(B3C_E_HIP, "radix sort sizing failed"); }
            DevBuf<char> tmp(ctx, tb);
            if (!tmp.p) { return ctx->fail(B3C_E_NOMEM, "b3c_permute_csr: sort workspace"); }
            e = hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, k_in.p, k_out.p, v_in.p, v_out.p, (int64_t)n_ent, 0, end_bit, s);
            if (e != hipSuccess) { return ctx->fail(B3C_E_HIP, "radix sort failed"); }
            const unsigned gs = (unsigned)((n_ent + 255) / 256);
            if (a->kind == B3C_F64)
                hipLaunchKernelGGL(k_long_scatter<double>, dim3(gs), dim3(256), 0, s, n_ent, k_out.p, v_out.p, long_rows.p, epos.p, a->indptr,
                                   (const double *)a->data, new2old, m->indptr, m->indices, (double *)m->data);
            else
                hipLaunchKernelGGL(k_long_scatter<uint32_t>, dim3(gs), dim3(256), 0, s, n_ent, k_out.p, v_out.p, long_rows.p, epos.p, a->indptr,
                                   (const uint32_t *)a->data, new2old, m->indptr, m->indices, (uint32_t *)m->data);
            HIP_TRY(ctx, hipStreamSynchronize(s));       // the scratch above goes back to the pool here
            }
        }
    } else if (nnz && !sort_cols) {
        const int g = grid_for(n, BLK / 64, 16 * ctx->n_cu);
        if (a->kind == B3C_F64)
            hipLaunchKernelGGL(k_permute_rows<double>, dim3(g), dim3(BLK), 0, s, n, a->indptr, a->indices, (const double *)a->data,
                               new2old, old2new, m->indptr, m->indices, (double *)m->data);
        else
            hipLaunchKernelGGL(k_permute_rows<uint32_t>, dim3(g), dim3(BLK), 0, s, n, a->indptr, a->indices,
                               (const uint32_t *)a->data, new2old, old2new, m->indptr, m->indices, (uint32_t *)m->data);
    } else if (nnz && block) {
        return ctx->fail(B3C_E_ARG, "b3c_permute_csr: the global-sort path cannot produce a row block");
    } else if (nnz) {
        DevBuf<int64_t> k_in(ctx, nnz), k_out(ctx, nnz);
        if (!k_in.p || !k_out.p) { return ctx->fail(B3C_E_NOMEM, "b3c_permute_csr: sort keys"); }
        hipLaunchKernelGGL(k_perm_keys, dim3(grid_for(n, BLK / 64, 16 * ctx->n_cu)), dim3(BLK), 0, s, n, a->indptr,
                           a->indices, old2new, k_in.p);
        int end_bit = 1;
        while (end_bit < 63 && ((unsigned __int128)1 << end_bit) < (unsigned __int128)n * (unsigned __int128)n) end_bit++;
        size_t tb = 0;
        hipError_t e;
        if (a->kind == B3C_F64)
            e = hipcub::DeviceRadixSort::SortPairs(nullptr, tb, k_in.p, k_out.p, (const double *)a->data, (double *)m->data, (int64_t)nnz, 0, end_bit, s);
        else
            e = hipcub::DeviceRadixSort::SortPairs(nullptr, tb, k_in.p, k_out.p, (const uint32_t *)a->data, (uint32_t *)m->data, (int64_t)nnz, 0, end_bit, s);
        if (e != hipSuccess) { return ctx->fail(B3C_E_HIP, "radix sort sizing failed"); }
        DevBuf<char> tmp(ctx, tb);
        if (!tmp.p) { return ctx->fail(B3C_E_NOMEM, "b3c_permute_csr: sort workspace"); }
        if (a->kind == B3C_F64)
            e = hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, k_in.p, k_out.p, (const double *)a->data, (double *)m->data, (int64_t)nnz, 0, end_bit, s);
        else
            e = hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, k_in.p, k_out.p, (const uint32_t *)a->data, (uint32_t *)m->data, (int64_t)nnz, 0, end_bit, s);
        if (e != hipSuccess) { return ctx->fail(B3C_E_HIP, "radix sort failed"); }
        hipLaunchKernelGGL(k_cols_from_keys, dim3(grid_for(nnz, 1024, 16 * ctx->n_cu)), dim3(256), 0, s, nnz, n, k_out.p, m->indices);
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(s));
    *out = own.release();
    return B3C_OK;
}

int b3c_permute_csr(b3c_ctx *ctx, const b3c_mat *a, const int32_t *new2old, const int32_t *old2new, b3c_mat **out,
                    bool sort_cols)
{
    return b3c_permute_csr_rows(ctx, a, new2old, old2new, out, sort_cols, 0, a->n);
}

// ---------------------------------------------------------------------------------------------
// public entry points (include/b3c.h)
// ---------------------------------------------------------------------------------------------
extern "C" int b3c_locality_order(b3c_ctx *ctx, const b3c_mat *a, int32_t *new2old, int32_t *old2new)
{
    if (!ctx) return B3C_E_ARG;
    if (!a || !new2old || !old2new) return ctx->fail(B3C_E_ARG, "b3c_locality_order: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf<int32_t> n2o(ctx, a->n), o2n(ctx, a->n);
    if (!n2o.p || !o2n.p) return ctx->fail(B3C_E_NOMEM, "b3c_locality_order");
    int rc = b3c_locality_order_dev(ctx, a, n2o.p, o2n.p);
    if (rc) return rc;
    if ((rc = b3c_copy_out(ctx, new2old, n2o.p, sizeof(int32_t) * (size_t)a->n))) return rc;
    return b3c_copy_out(ctx, old2new, o2n.p, sizeof(int32_t) * (size_t)a->n);
}

extern "C" int b3c_mat_permute_rows(b3c_ctx *ctx, const b3c_mat *a, const int32_t *new2old, int flags, int64_t row0, int64_t row1,
                                    b3c_mat **out)
{
    if (!ctx) return B3C_E_ARG;
    if (!a || !new2old || !out || row0 < 0 || row1 < row0 || row1 > a->n) return ctx->fail(B3C_E_ARG, "b3c_mat_permute: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t n = a->n;
    DevBuf<int32_t> n2o(ctx, n), o2n(ctx, n);
    if (!n2o.p || !o2n.p) return ctx->fail(B3C_E_NOMEM, "b3c_mat_permute");
    int rc = b3c_copy_in(ctx, n2o.p, new2old, sizeof(int32_t) * (size_t)n);
    if (rc) return rc;
    const unsigned gn = (unsigned)((n + 255) / 256);
    if (n) hipLaunchKernelGGL(k_invert_perm, dim3(gn), dim3(256), 0, ctx->stream, n, n2o.p, o2n.p);
    MatOwner own(ctx);
    if ((rc = b3c_permute_csr_rows(ctx, a, n2o.p, o2n.p, &own.m, !(flags & B3C_PERMUTE_UNSORTED), row0, row1))) return rc;
    b3c_mat *m = own.m;
    // caller-side id of every row: compose with the input's ids if it was permuted before
    m->ids = (int32_t *)ctx->alloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    if (!m->ids) return ctx->fail(B3C_E_NOMEM, "b3c_mat_permute");
    if (n) {
        if (a->ids) hipLaunchKernelGGL(k_gather_i32, dim3(gn), dim3(256), 0, ctx->stream, n, a->ids, n2o.p, m->ids);
        else HIP_TRY(ctx, hipMemcpyAsync(m->ids, n2o.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
    }
    m->ordered = true;
    m->value_symmetric = a->value_symmetric;
    m->row0 = row0;
    m->row1 = row1;
    m->row_block = row0 > 0 || row1 < n;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *out = own.release();
    return B3C_OK;
}

extern "C" int b3c_mat_permute(b3c_ctx *ctx, const b3c_mat *a, const int32_t *new2old, int flags, b3c_mat **out)
{
    if (!ctx) return B3C_E_ARG;
    if (!a) return ctx->fail(B3C_E_ARG, "b3c_mat_permute: bad argument");
    return b3c_mat_permute_rows(ctx, a, new2old, flags, 0, a->n, out);
}
