// Louvain community detection on gfx950 -- replaces to_graph's weight scaling (mzd/cluster.py:316-321)
// and louvain.cluster -> community.best_partition (mzd/louvain.py:70).
//
// Schedule and arithmetic are specified by oracle/louvain_bsp.c (the scalar restatement the tests
// compare against, bit for bit): fixed-point edge weights, exact integer gains, a sweep of n_sub
// bulk-synchronous sub-rounds, ties to the lowest community id, singleton guard, python-louvain's
// 1e-7 stopping rules.  Because every sum is an integer sum, no result depends on the order in which
// lanes, waves or atomics happen to run.
//
// Kernels (all HBM/L2-bound integer work; no MFMA):
//   lv_decide   neighbour-community weights accumulated in an LDS hash table (open addressing, ds_cmpst +
//               ds_add_u64), own-community weight in registers, exact 128-bit gain per candidate, order-free
//               arg-max.  By row length: k_lv_decide_g (several nodes per wave, entries in registers, one 16-byte
//               {community, total} record gathered per neighbour, list positions dealt XCD by XCD),
//               k_lv_decide (one wave per node), k_lv_decide_wg (one workgroup per node, table of 1024 /
//               4096 / 8192 slots), k_lv_mega_* (rows beyond 6553 entries: table in global memory, chunks
//               pre-summed in LDS, evaluation split over the table), k_lv_decide_direct (levels of <= 2048 nodes)
//   lv_apply    moves of one sub-round: integer atomics on the community totals
//   lv_modularity / lv_tot2   exact modularity numerator
//   aggregation: kc x kc integer bins (few communities); else per-community hashing (k_aggh_*); on row blocks
//               (community pair, weight) keys -> radix sort + reduce-by-key (hipcub) -> next level CSR
#include "common.hpp"

#include <algorithm>
#include <array>
#include <cstdlib>
#include <hipcub/hipcub.hpp>

int b3c_exclusive_scan_i64(b3c_ctx *ctx, const int64_t *in, int64_t *out, int64_t count);
int b3c_locality_order_dev(b3c_ctx *ctx, const b3c_mat *a, int32_t *new2old, int32_t *old2new);
int b3c_permute_csr(b3c_ctx *ctx, const b3c_mat *a, const int32_t *new2old, const int32_t *old2new, b3c_mat **out,
                    bool sort_cols);

namespace {

constexpr int BLK = 256;
constexpr int WPB = BLK / 64;
typedef __int128 i128;
typedef unsigned long long u64;
// 16-byte loads of entry streams that are only as aligned as their elements (a row starts anywhere).  A CU's texture addresser spends
// ~26 cycles on every vector-memory instruction of a wave whatever it moves (tools/probes/ta_probe.hip): four consecutive entries
// per lane and load instead of one.  The streams carry 32 bytes of slack behind their last entry.
typedef int32_t i32x4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef long long i64x2_u __attribute__((ext_vector_type(2), aligned(8)));
typedef double f64x2_u __attribute__((ext_vector_type(2), aligned(8)));
// The entry streams of a large level (ids, weights: 50-200x the labels and totals the gathers want to find in the L2) are loaded
// with the non-temporal policy, so that they pass through without pushing those out -- under the same rule as the KR tail pass's
// words (kr.hip, KR_STREAM_NT_MIN): only when the level does not fit the caches anyway (C3: the scan 31.8 -> 31.3 ms per step;
// C2's level 0, 120 MB, stays resident between sub-rounds and keeps the default policy).  Measured on `k_lv_decide_g` only: the
// same bit on the recount, tracking, workgroup and set-up kernels together cost 0.7 ms at C3.
constexpr int64_t LV_STREAM_NT_MIN = 192ll << 20;        // bytes of a level's entries (12 per stored entry)
template <class T>
__device__ __forceinline__ T lv_stream(const T *p, int nt)
{
    return nt ? __builtin_nontemporal_load(p) : *p;
}

__host__ __device__ __forceinline__ uint32_t bucket_of(int64_t i, int n_sub)
{
    uint32_t h = (uint32_t)i * 2654435761u;
    return (h >> 16) % (uint32_t)n_sub;
}

// ONE rule, used by the host (b3c_louvain: `incr`) and by the recount kernels (HALF mode): the sweep that follows one which moved
// under 35 % of the level's nodes tracks the internal weight through the movers' rows (and prunes), the others recount it in full
__host__ __device__ __forceinline__ bool lv_track_next(unsigned long long moved, unsigned long long n_level) { return moved * 20ull < n_level * 7ull; }

struct LvGraph {
    int64_t n = 0, nnz = 0;
    int64_t *indptr = nullptr;
    int32_t *indices = nullptr;
    int64_t *q = nullptr;
    bool owned = false;
    // row-sharded runs: this rank holds the rows [starts[rank], starts[rank + 1]) only (the others are empty); starts: world + 1
    // offsets, the same on every rank.  Level 0 on a row block of the map, and every induced graph built by OWNER (round 6)
    bool block = false;
    std::vector<int64_t> starts;
};

// ---------------------------------------------------------------------------------------------
// graph weights + quantisation
// ---------------------------------------------------------------------------------------------
// w = a * scl with the value of the LOWER triangle entry on both sides (cluster.py:319-321: a later
// add_edge(v,u) overwrites the earlier (u,v)).  Rows must have sorted columns.
__global__ void __launch_bounds__(BLK) k_graph_w(int64_t n, const int64_t *__restrict__ indptr,
                                                  const int32_t *__restrict__ indices,
                                                  const double *__restrict__ a, const int32_t *__restrict__ ids,
                                                  int sorted, double scl, double *__restrict__ w,
                                                  double *__restrict__ part_max)
{
    __shared__ double lds[WPB];
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6);
    const int64_t nwave = (int64_t)gridDim.x * WPB;
    double mx = 0.0;
    for (int64_t row = wave; row < n; row += nwave) {
        const int64_t s = indptr[row], e = indptr[row + 1];
        const int32_t idr = ids ? ids[row] : (int32_t)row;
        // four independent binary searches per lane keep the dependent probes of one from stalling the wave
        for (int64_t base = s; base < e; base += 256) {
            int32_t j[4];
            double val[4];
            int64_t lo[4], hi[4], je[4];
            bool need[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int64_t k = base + t * 64 + lane;
                const bool ok = k < e;
                j[t] = ok ? indices[k] : (int32_t)row;
                val[t] = ok ? a[k] : 0.0;
            }
#pragma unroll
            for (int t = 0; t < 4; t++) {
                // "lower triangle" is meant in the caller's numbering, whatever the storage order
                need[t] = (ids ? ids[j[t]] : j[t]) > idr;
                lo[t] = need[t] ? indptr[j[t]] : 0;
                hi[t] = need[t] ? indptr[j[t] + 1] : 0;
                je[t] = hi[t];
            }
            bool any = true;
            while (any) {
                any = false;
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    if (lo[t] < hi[t]) {
                        if (sorted) {
                            const int64_t mid = (lo[t] + hi[t]) >> 1;
                            if (indices[mid] < (int32_t)row) lo[t] = mid + 1; else hi[t] = mid;
                        } else if (indices[lo[t]] == (int32_t)row) {
                            hi[t] = lo[t];                      // found: stop scanning this one
                        } else {
                            lo[t]++;
                        }
                        any = true;
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int64_t k = base + t * 64 + lane;
                if (k >= e) continue;
                if (need[t] && lo[t] < je[t] && indices[lo[t]] == (int32_t)row) val[t] = a[lo[t]];
                const double ww = val[t] * scl;
                w[k] = ww;
                mx = fmax(mx, ww);
            }
        }
    }
    mx = wave_max(mx);
    if (lane == 0) lds[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = lds[0];
        for (int i = 1; i < WPB; i++) r = fmax(r, lds[i]);
        part_max[blockIdx.x] = r;
    }
}

// sum of ceil(w * 2^(20-e)), diagonal twice: exact integer bound on 2m (oracle: orc_quant_shift)
__global__ void __launch_bounds__(BLK) k_quant_total(int64_t n, const int64_t *__restrict__ indptr,
                                                      const int32_t *__restrict__ indices,
                                                      const double *__restrict__ w, double scl, int ex, u64 *__restrict__ total)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6);
    const int64_t nwave = (int64_t)gridDim.x * WPB;
    u64 acc = 0;
    for (int64_t row = wave; row < n; row += nwave) {
        const int64_t s = indptr[row], e = indptr[row + 1];
        for (int64_t k = s + lane; k < e; k += 64) {
            u64 c = (u64)ceil(ldexp(w[k] * scl, ex));
            acc += indices[k] == (int32_t)row ? 2 * c : c;
        }
    }
    __shared__ long long red[WPB];
    block_atomic_add_u64((long long)acc, total, red);
}

__global__ void k_quantise(int64_t nnz, const double *__restrict__ w, double scl, int shift, int64_t *__restrict__ q)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += stride)
        q[k] = __double2ll_rn(ldexp(w[k] * scl, shift));
}

// value-symmetric matrices: only the largest weight is needed before the quantisation passes form a * scl themselves
__global__ void __launch_bounds__(BLK) k_graph_wmax_sym(int64_t nnz, const double *__restrict__ a, double scl,
                                                         double *__restrict__ part_max)
{
    __shared__ double lds[WPB];
    double mx = 0.0;
    const int64_t stride = (int64_t)gridDim.x * BLK;
    for (int64_t k = (int64_t)blockIdx.x * BLK + threadIdx.x; k < nnz; k += stride) mx = fmax(mx, a[k] * scl);
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = lds[0];
        for (int i = 1; i < WPB; i++) r = fmax(r, lds[i]);
        part_max[blockIdx.x] = r;
    }
}

// ---------------------------------------------------------------------------------------------
// level set-up
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BLK) k_lv_init(int64_t n, const int64_t *__restrict__ indptr,
                                                  const int32_t *__restrict__ indices, const int64_t *__restrict__ q,
                                                  const int32_t *__restrict__ ids, int64_t *__restrict__ kdeg,
                                                  int64_t *__restrict__ tot, int32_t *__restrict__ com,
                                                  int32_t *__restrict__ nxt, int32_t *__restrict__ csize,
                                                  u64 *__restrict__ m2, int64_t *__restrict__ wnl, int64_t r0, int64_t r1,
                                                  u64 *__restrict__ i2, const double *__restrict__ wq, double wscl, int shift,
                                                  int64_t *__restrict__ q_out)
{
    // wq != nullptr (level 0 of a value-symmetric map): the fixed-point weights are FORMED here, q = rint(a * scl * 2^shift)
    // exactly as k_quantise forms them, and written to q_out for the kernels that follow -- the quantisation costs no pass of
    // its own (round 3)
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6);
    const int64_t nwave = (int64_t)gridDim.x * WPB;
    long long total = 0, loops = 0;
    // two rows per wave and pass: the loads of both are in flight together
    for (int64_t row0 = wave; row0 < n; row0 += 2 * nwave) {
        int64_t rw[2], rs_[2], re_[2];
        rw[0] = row0;
        rw[1] = row0 + nwave < n ? row0 + nwave : -1;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            rs_[h] = rw[h] >= 0 ? indptr[rw[h]] : 0;
            re_[h] = rw[h] >= 0 ? indptr[rw[h] + 1] : 0;
        }
        long long d[2] = {0, 0}, lw[2] = {0, 0};
        const int64_t span = (re_[0] - rs_[0]) > (re_[1] - rs_[1]) ? (re_[0] - rs_[0]) : (re_[1] - rs_[1]);
        for (int64_t off = 0; off < span; off += 256) {
            int32_t j[2][4];
            long long w[2][4];
            // (four consecutive entries per lane as 16-byte vectors; a lane past the row's end re-reads its beginning and drops it)
            if (wq) {
                double a[2][4];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int64_t k0 = rs_[h] + off + 4 * lane, kb = k0 < re_[h] ? k0 : rs_[h];
                    const i32x4_u jv = *(const i32x4_u *)(indices + kb);
                    const f64x2_u a0 = *(const f64x2_u *)(wq + kb), a1 = *(const f64x2_u *)(wq + kb + 2);
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        const bool ok = k0 + t < re_[h];
                        j[h][t] = ok ? jv[t] : -1;
                        a[h][t] = ok ? (t < 2 ? a0[t] : a1[t - 2]) : 0.0;
                    }
                }
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int64_t k0 = rs_[h] + off + 4 * lane;
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        w[h][t] = j[h][t] >= 0 ? __double2ll_rn(ldexp(a[h][t] * wscl, shift)) : 0;
                        if (j[h][t] >= 0) q_out[k0 + t] = w[h][t];
                    }
                }
            } else {
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int64_t k0 = rs_[h] + off + 4 * lane, kb = k0 < re_[h] ? k0 : rs_[h];
                    const i32x4_u jv = *(const i32x4_u *)(indices + kb);
                    const i64x2_u w0 = *(const i64x2_u *)(q + kb), w1 = *(const i64x2_u *)(q + kb + 2);
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        const bool ok = k0 + t < re_[h];
                        j[h][t] = ok ? jv[t] : -1;
                        w[h][t] = ok ? (t < 2 ? w0[t] : w1[t - 2]) : 0;
                    }
                }
            }
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    d[h] += j[h][t] == (int32_t)rw[h] ? 2 * w[h][t] : w[h][t];
                    lw[h] += j[h][t] == (int32_t)rw[h] ? w[h][t] : 0;
                }
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const long long dd = wave_sum_ll(d[h]), ll = wave_sum_ll(lw[h]);
            const int64_t row = rw[h];
            if (lane == 0 && row >= 0) {
                const int32_t id = ids ? ids[row] : (int32_t)row;   // community ids live in the caller's numbering
                kdeg[row] = dd;
                wnl[row] = dd - 2 * ll;                             // weight towards the OTHER nodes
                tot[id] = dd;
                com[row] = id;
                nxt[row] = id;
                csize[id] = re_[h] > rs_[h] ? 1 : 0;   // empty rows are not graph nodes (cluster.py:321 never adds them)
                total += dd;
                if (row >= r0 && row < r1) loops += 2 * ll;         // every node alone: I2 is the self-loops (this rank's rows)
            }
        }
    }
    __shared__ long long red[WPB];
    block_atomic_add_u64(lane == 0 ? total : 0, m2, red);
    block_atomic_add_u64(lane == 0 ? loops : 0, i2, red);
}

// ---------------------------------------------------------------------------------------------
// delta-Q neighbour scan
// ---------------------------------------------------------------------------------------------
constexpr int32_t EMPTY = -1;
constexpr int LV_DIRECT_CAP = 2048;     // levels of at most this many nodes: k_lv_decide_direct

__device__ __forceinline__ uint32_t hash_slot(int32_t c, int lg) { return ((uint32_t)c * 0x9E3779B1u) >> (32 - lg); }

// gain of a move as an exact 128-bit integer: d * 2m - tot * k_i.  All four factors are non-negative and below 2^64, so both
// products are UNSIGNED 64 x 64 -> 128 multiplications (four v_mad_u64_u32 each); written with signed 128-bit operands the
// second one carried a dozen instructions of sign handling per evaluation (round 4, from the ISA of k_lv_decide_g).
__device__ __forceinline__ i128 gain128(u64 d, u64 m2, u64 t, u64 k)
{
    return (i128)((unsigned __int128)d * (unsigned __int128)m2) - (i128)((unsigned __int128)t * (unsigned __int128)k);
}

struct Best {
    i128 gain;
    int32_t com;
    int32_t have;
};

__device__ __forceinline__ void best_take(Best &b, i128 g, int32_t c)
{
    if (g > 0 && (!b.have || g > b.gain || (g == b.gain && c < b.com))) {
        b.gain = g;
        b.com = c;
        b.have = 1;
    }
}

// Arg-max of the lanes' candidates over the wave: largest gain, ties to the lowest community id.  A candidate's gain is positive
// (best_take), a lane without one holds gain 0, so the 128-bit gains compare as unsigned integers word by word: four 32-bit
// maxima, each among the lanes that matched the words above, then the minimum id among the lanes that hold the maximum -- five
// DPP reductions (wave_max_u32, common.hpp: six v_max_u32_dpp and a v_readlane each), the result in scalar registers.  The same
// arg-max as six rounds of __shfl_down over (gain, id, have) -- 24 ds_bpermute and six 128-bit compare-and-select blocks, 375 of
// the ~1400 instructions of k_lv_decide_g's loop body (round 4, from its ISA) -- and uniform over the wave instead of lane 0 only.
__device__ __forceinline__ Best best_wave_reduce(Best b)
{
    const u64 hi = (u64)(b.gain >> 64), lo = (u64)b.gain;
    const uint32_t w[4] = {(uint32_t)(hi >> 32), (uint32_t)hi, (uint32_t)(lo >> 32), (uint32_t)lo};
    uint32_t m[4];
    bool cand = b.have != 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        m[k] = wave_max_u32(cand ? w[k] : 0u);
        cand = cand && w[k] == m[k];
    }
    const uint32_t mc = wave_max_u32(cand ? ~(uint32_t)b.com : 0u);      // ids are >= 0: ~id has the top bit set, 0 = no candidate
    Best r;
    r.have = (m[0] | m[1] | m[2] | m[3]) != 0u ? 1 : 0;
    r.gain = (i128)(((unsigned __int128)(((u64)m[0] << 32) | m[1]) << 64) | (unsigned __int128)(((u64)m[2] << 32) | m[3]));
    r.com = r.have ? (int32_t)~mc : b.com;
    return r;
}

// Largest 64-bit value over the LPR lanes of a group, in every lane of the group (all lanes of the wave active).
template <int LPR>
__device__ __forceinline__ u64 group_max_u64(u64 v)
{
    if constexpr (LPR == 64) {
        const uint32_t hi = wave_max_u32((uint32_t)(v >> 32));
        const uint32_t lo = wave_max_u32((uint32_t)(v >> 32) == hi ? (uint32_t)v : 0u);
        return ((u64)hi << 32) | lo;
    } else {
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) {
            const u64 t = (u64)__shfl_xor((long long)v, o, LPR);
            v = t > v ? t : v;
        }
        return v;
    }
}

struct NodeRec {
    int32_t node;
    int32_t deg;
    int64_t start;
};

// table entries needed for a node of degree deg (worst case: every neighbour in its own community)
__host__ __device__ __forceinline__ int table_lg(int64_t deg)
{
    int lg = 6;
    while ((1ll << lg) < deg + deg / 4 + 1) lg++;
    return lg;
}

// CAP = LDS table entries per wave.  Nodes whose table does not fit use the global table at gkeys/gvals + goff[i].
// Loads first: every lane pulls four (index, weight) pairs, then the four community gathers, then inserts, so
// that a wave keeps ~3 KB of independent loads in flight instead of one dependent chain per 64 entries.
template <int CAP>
__global__ void __launch_bounds__(BLK) k_lv_decide(int64_t r0, int64_t n, const NodeRec *__restrict__ hubs, int skip_upto,
                                                    int n_sub, int round, u64 m2,
                                                    const int64_t *__restrict__ indptr,
                                                    const int32_t *__restrict__ indices,
                                                    const int64_t *__restrict__ q, const int32_t *__restrict__ ids,
                                                    const int64_t *__restrict__ kdeg,
                                                    const int64_t *__restrict__ tot, const int32_t *__restrict__ com,
                                                    const int32_t *__restrict__ csize, int32_t *__restrict__ nxt,
                                                    const int64_t *__restrict__ goff, int32_t *__restrict__ gkeys,
                                                    u64 *__restrict__ gvals, const int64_t *__restrict__ cnt_dev)
{
    // cnt_dev: the list was thinned on the device (k_lv_prune) and its length sits in device memory
    if (cnt_dev) { const int64_t c = *cnt_dev; n = r0 + (c < n - r0 ? c : n - r0); }
    __shared__ int32_t skeys[WPB][CAP];
    __shared__ u64 svals[WPB][CAP];
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int64_t wave = (int64_t)blockIdx.x * WPB + wib;
    const int64_t nwave = (int64_t)gridDim.x * WPB;
    int32_t *lk = skeys[wib];
    u64 *lv = svals[wib];
    for (int t = lane; t < CAP; t += 64) { lk[t] = EMPTY; lv[t] = 0; }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");

    // hubs != nullptr: iterate the prepared list hubs[r0 .. n) of this sub-round; else rows [r0, n) of this rank
    for (int64_t it = r0 + wave; it < n; it += nwave) {
        const int64_t i = hubs ? hubs[it].node : it;
        const int32_t id = ids ? ids[i] : (int32_t)i;
        if (bucket_of(id, n_sub) != (uint32_t)round) continue;
        const long long ki = kdeg[i];
        if (ki == 0) continue;
        const int64_t s = indptr[i], e = indptr[i + 1];
        const int64_t deg = e - s;
        if (deg <= skip_upto) continue;      // short nodes are taken by k_lv_decide_g
        const int32_t a = com[i];
        const int lg = table_lg(deg);
        const bool use_lds = (1ll << lg) <= CAP;
        const uint32_t mask = (1u << lg) - 1u;
        int32_t *keys = use_lds ? lk : gkeys + goff[i];
        u64 *vals = use_lds ? lv : gvals + goff[i];

        long long own = 0;
        // A lane that turns an EMPTY slot into its community "claims" it; each distinct community is claimed exactly
        // once per node, so evaluating the claimed slots visits every candidate without scanning (and re-clearing)
        // the whole table.  Nodes of at most 256 entries keep their claims in four registers per lane.
        const bool small = deg <= 256;
        uint32_t claim[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        bool any_claim = false;
        for (int64_t base = s; base < e; base += 256) {
            int32_t j[4];
            u64 qv[4];
            int32_t c[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int64_t k = base + t * 64 + lane;
                const bool ok = k < e;
                j[t] = ok ? indices[k] : (int32_t)i;
                qv[t] = ok ? (u64)q[k] : 0;
            }
#pragma unroll
            for (int t = 0; t < 4; t++) c[t] = j[t] != (int32_t)i ? com[j[t]] : EMPTY;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                if (c[t] == EMPTY) continue;
                if (c[t] == a) { own += (long long)qv[t]; continue; }
                uint32_t slot = hash_slot(c[t], lg);
                for (;;) {
                    int32_t prev = atomicCAS(&keys[slot], EMPTY, c[t]);
                    if (prev == EMPTY || prev == c[t]) {
                        atomicAdd(&vals[slot], qv[t]);
                        if (prev == EMPTY) { claim[t] = slot; any_claim = true; }
                        break;
                    }
                    slot = (slot + 1) & mask;
                }
            }
        }
        own = wave_sum_ll(own);
        if (use_lds) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        else __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");

        Best b;
        b.gain = 0; b.com = a; b.have = 0;
        if (small) {
            if (__ballot(any_claim)) {
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    if (claim[t] == 0xFFFFFFFFu) continue;
                    const int32_t c = keys[claim[t]];
                    const u64 d = vals[claim[t]];
                    keys[claim[t]] = EMPTY;
                    vals[claim[t]] = 0;
                    const i128 g = gain128((u64)d, m2, (u64)tot[c], (u64)ki);
                    best_take(b, g, c);
                }
            }
        } else {
            // four slots per lane in flight: the totals of the communities found are gathered together, not one dependent
            // round trip per slot (a 2048-slot table is 32 slots per lane; round 3)
            for (uint32_t t0 = lane; t0 <= mask; t0 += 4 * 64) {
                int32_t c[4];
                u64 d[4];
                long long tc[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t t = t0 + u * 64;
                    c[u] = t <= mask ? keys[t] : EMPTY;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t t = t0 + u * 64;
                    d[u] = c[u] != EMPTY ? vals[t] : 0;
                    tc[u] = c[u] != EMPTY ? tot[c[u]] : 0;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (c[u] == EMPTY) continue;
                    const uint32_t t = t0 + u * 64;
                    keys[t] = EMPTY;
                    vals[t] = 0;
                    const i128 g = gain128((u64)d[u], m2, (u64)tc[u], (u64)ki);
                    best_take(b, g, c[u]);
                }
            }
        }
        if (lane == 0) {
            // staying: i removed from its own community first (python-louvain __remove)
            const i128 g = gain128((u64)own, m2, (u64)(tot[a] - ki), (u64)ki);
            best_take(b, g, a);
        }
        b = best_wave_reduce(b);
        if (lane == 0) {
            int32_t best = b.have ? b.com : a;
            if (best != a && csize[a] == 1 && csize[best] == 1 && best > a) best = a;
            nxt[i] = best;
        }
        if (use_lds) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        else __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    }
}

// MEGA rows: one WORKGROUP per node.  A row of 10^5 .. 10^6 entries -- a planted hub contig at level 0, or at level 1 the
// community of a genome that holds a third of all contigs, which the noise links to every other community -- kept ONE wave of
// k_lv_decide busy for milliseconds per sub-round while the rest of the chip idled (round 3, the surveyed genome-size law:
// 5.3 ms per sub-round of level 1, 40 % of the whole step).  Same protocol in two passes: workgroups of 4096 entries insert into
// the node's table in global memory (goff) by compare-and-swap, then one workgroup per node evaluates and clears every slot;
// own-community weight and arg-max are reduced with integer sums and an order-free arg-max (ties to the lowest id): the same
// result as one wave's.
constexpr int64_t SIDE_MIN_HUBS = 512;     // hub nodes of a sub-round from which the side stream pays (see b3c_louvain)
constexpr int64_t PRUNE_MIN_HUBS = 1024;   // ... and from which pruning them is worth its launch
constexpr int MEGA_NT = 1024;
constexpr int MEGA_CHUNK = 4 * MEGA_NT;    // entries one workgroup of the insert pass takes

// pass 1: workgroup (node m, chunk c) inserts entries [c * MEGA_CHUNK, ...) of the node's row into the node's table and adds
// its share of the own-community weight to own_acc[m] (integer atomics: order-free)
__global__ void __launch_bounds__(MEGA_NT) k_lv_mega_insert(const NodeRec *__restrict__ list, int chunks, const int32_t *__restrict__ indices,
                                                             const int64_t *__restrict__ q, const int32_t *__restrict__ com,
                                                             const int64_t *__restrict__ goff, int32_t *__restrict__ gkeys,
                                                             u64 *__restrict__ gvals, u64 *__restrict__ own_acc, const int64_t *__restrict__ cnt_dev = nullptr)
{
    __shared__ long long red[MEGA_NT / 64];
    const int m = blockIdx.x / chunks, ch = blockIdx.x % chunks;
    if (cnt_dev && m >= *cnt_dev) return;                            // (the list was thinned on the device: k_lv_prune)
    const NodeRec rec = list[m];
    if ((int64_t)ch * MEGA_CHUNK >= rec.deg) return;                 // (uniform per workgroup)
    const int32_t i = rec.node;
    const int32_t a = com[i];
    const int lg = table_lg(rec.deg);
    const uint32_t mask = (1u << lg) - 1u;
    int32_t *keys = gkeys + goff[i];
    u64 *vals = gvals + goff[i];
    const int64_t end = rec.start + rec.deg;
    const int64_t base = rec.start + (int64_t)ch * MEGA_CHUNK + threadIdx.x;
    int32_t j[4];
    u64 qv[4];
    int32_t c[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int64_t k = base + (int64_t)t * MEGA_NT;
        const bool ok = k < end;
        j[t] = ok ? indices[k] : i;
        qv[t] = ok ? (u64)q[k] : 0;
    }
#pragma unroll
    for (int t = 0; t < 4; t++) c[t] = j[t] != i ? com[j[t]] : EMPTY;
    long long own = 0;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        if (c[t] == EMPTY) continue;
        if (c[t] == a) { own += (long long)qv[t]; continue; }
        uint32_t slot = hash_slot(c[t], lg);
        for (;;) {
            const int32_t prev = atomicCAS(&keys[slot], EMPTY, c[t]);
            if (prev == EMPTY || prev == c[t]) { atomicAdd(&vals[slot], qv[t]); break; }
            slot = (slot + 1) & mask;
        }
    }
    block_atomic_add_u64(own, own_acc + m, red);
}

// pass 2 of the mega rows, split over the table (round 4): one workgroup per node used to walk the node's whole table -- 2^18 slots for
// the 147k-entry row of the surveyed law's level 1, 134 us per sub-round with the rest of the chip idle.  Workgroup (node m, chunk
// ch) evaluates and clears MEGA_ESL slots and leaves its best candidate in part[]; the LAST workgroup of a node to finish (a ticket
// per node) folds the partial results in chunk order with the same order-free arg-max (ties to the lowest community id) and
// settles the node.  Same candidates, same comparisons: the same decision as one workgroup's walk.
constexpr int MEGA_ESL = 16384;            // table slots per workgroup of the split evaluation
struct MegaPart { u64 hi, lo, com_have; };
__global__ void __launch_bounds__(MEGA_NT) k_lv_mega_eval_split(const NodeRec *__restrict__ list, int echunks, u64 m2, const int64_t *__restrict__ kdeg,
                                                                 const int64_t *__restrict__ tot, const int32_t *__restrict__ com,
                                                                 const int32_t *__restrict__ csize, int32_t *__restrict__ nxt,
                                                                 const int64_t *__restrict__ goff, int32_t *__restrict__ gkeys,
                                                                 u64 *__restrict__ gvals, u64 *__restrict__ own_acc, MegaPart *__restrict__ part,
                                                                 unsigned int *__restrict__ ticket, const int64_t *__restrict__ cnt_dev = nullptr)
{
    __shared__ long long red_hi[MEGA_NT / 64];
    __shared__ u64 red_lo[MEGA_NT / 64];
    __shared__ int32_t red_com[MEGA_NT / 64], red_have[MEGA_NT / 64];
    __shared__ int last;
    const int m = blockIdx.x / echunks, ch = blockIdx.x % echunks;
    if (cnt_dev && m >= *cnt_dev) return;
    const NodeRec rec = list[m];
    const int lg = table_lg(rec.deg);
    const uint32_t size = 1u << lg;
    const int nch = (int)((size + MEGA_ESL - 1) / MEGA_ESL);
    if (ch >= nch) return;                                            // (uniform per workgroup)
    const int32_t i = rec.node;
    const long long ki = kdeg[i];
    const int32_t a = com[i];
    int32_t *keys = gkeys + goff[i];
    u64 *vals = gvals + goff[i];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    Best b;
    b.gain = 0; b.com = a; b.have = 0;
    const uint32_t s0 = (uint32_t)ch * MEGA_ESL, s1 = s0 + MEGA_ESL < size ? s0 + MEGA_ESL : size;
    for (uint32_t t0 = s0 + threadIdx.x; t0 < s1; t0 += 4 * MEGA_NT) {
        int32_t c[4];
        u64 d[4];
        long long tc[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t t = t0 + u * MEGA_NT;
            c[u] = t < s1 ? keys[t] : EMPTY;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t t = t0 + u * MEGA_NT;
            d[u] = c[u] != EMPTY ? vals[t] : 0;
            tc[u] = c[u] != EMPTY ? tot[c[u]] : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (c[u] == EMPTY) continue;
            const uint32_t t = t0 + u * MEGA_NT;
            keys[t] = EMPTY;
            vals[t] = 0;
            const i128 g = gain128((u64)d[u], m2, (u64)tc[u], (u64)ki);
            best_take(b, g, c[u]);
        }
    }
    b = best_wave_reduce(b);
    if (lane == 0) {
        red_hi[wv] = (long long)(b.gain >> 64);
        red_lo[wv] = (u64)b.gain;
        red_com[wv] = b.com;
        red_have[wv] = b.have;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        Best r;
        r.gain = 0; r.com = a; r.have = 0;
        for (int w = 0; w < MEGA_NT / 64; w++) {
            const i128 g = ((i128)red_hi[w] << 64) | (i128)red_lo[w];
            if (red_have[w] && (!r.have || g > r.gain || (g == r.gain && red_com[w] < r.com))) { r.gain = g; r.com = red_com[w]; r.have = 1; }
        }
        // the partial result travels through agent-scope atomics: the workgroups of a node run on different XCDs, each with its own L2
        u64 *slot = (u64 *)&part[(int64_t)m * echunks + ch];
        __hip_atomic_store(slot + 0, (u64)(long long)(r.gain >> 64), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(slot + 1, (u64)r.gain, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(slot + 2, (u64)(uint32_t)r.com | ((u64)(uint32_t)r.have << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = __hip_atomic_fetch_add(&ticket[m], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(nch - 1);
        if (last) {
            __hip_atomic_store(&ticket[m], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next sub-round
            Best f;
            f.gain = 0; f.com = a; f.have = 0;
            for (int k = 0; k < nch; k++) {
                u64 *q = (u64 *)&part[(int64_t)m * echunks + k];
                const u64 qh = __hip_atomic_load(q + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const u64 ql = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const u64 qc = __hip_atomic_load(q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const i128 g = ((i128)(long long)qh << 64) | (i128)ql;
                const int32_t qcom = (int32_t)(uint32_t)qc, qhave = (int32_t)(qc >> 32);
                if (qhave && (!f.have || g > f.gain || (g == f.gain && qcom < f.com))) { f.gain = g; f.com = qcom; f.have = 1; }
            }
            const long long o = (long long)own_acc[m];
            own_acc[m] = 0;
            // staying: i removed from its own community first (python-louvain __remove)
            const i128 g = gain128((u64)o, m2, (u64)(tot[a] - ki), (u64)ki);
            best_take(f, g, a);
            int32_t best = f.have ? f.com : a;
            if (best != a && csize[a] == 1 && csize[best] == 1 && best > a) best = a;
            nxt[i] = best;
        }
    }
}

// Rows between the grouped kernel's reach and the mega rows: one WORKGROUP per node (round 4).  The induced graph of a level
// that ended with 10^5 communities (the surveyed genome-size law at C3: 185,559 nodes, ~1000 entries per row) is made of such rows,
// and one WAVE per node with a 2048-slot table of its own (96 KB per four waves: one workgroup per CU, four waves to hide the
// community gathers of a 1000-entry row) took 1.9 ms per sub-round there.  Here the NT threads of a workgroup share ONE table of
// 2^LGS slots in LDS (12 bytes per slot): 12 / 48 / 96 KB for rows of up to 819 / 3276 / 6553 entries, i.e. 8 / 3 / 1 workgroups
// per CU with every lane holding four gathers in flight.  Same protocol as k_lv_decide -- insert by compare-and-swap, evaluate and
// clear every slot of the node's table size, exact 128-bit gains, order-free arg-max (ties to the lowest community id) -- hence
// the same result.
constexpr int LV_WG_CLASSES = 3;
// the scan's workgroup classes (round 5: a 2048-slot class between the first two -- rows of 820 .. 1638 entries, the bulk of level 1 on
// the contract workload, get six workgroups per CU instead of three)
__host__ __device__ __forceinline__ int lv_dwg_max_deg(int cls) { return cls == 0 ? 819 : cls == 1 ? 1638 : cls == 2 ? 3276 : 6553; }   // table_lg() <= 10 / 11 / 12 / 13
__host__ __device__ __forceinline__ int lv_wg_max_deg(int cls) { return cls == 0 ? 819 : cls == 1 ? 3276 : 6553; }   // table_lg() <= 10 / 12 / 13

template <int LGS, int NT>
__global__ void __launch_bounds__(NT) k_lv_decide_wg(int64_t cnt_max, const int64_t *__restrict__ cnt_dev, const NodeRec *__restrict__ list, u64 m2,
                                                      const int32_t *__restrict__ indices, const int64_t *__restrict__ q,
                                                      const int64_t *__restrict__ kdeg, const int64_t *__restrict__ tot,
                                                      const int32_t *__restrict__ com, const int32_t *__restrict__ csize,
                                                      int32_t *__restrict__ nxt)
{
    constexpr int SLOTS = 1 << LGS;
    constexpr int NW = NT / 64;
    extern __shared__ __align__(16) unsigned char lv_wg_smem[];
    u64 *vals = (u64 *)lv_wg_smem;
    int32_t *keys = (int32_t *)(vals + SLOTS);
    long long *red_hi = (long long *)(keys + SLOTS);
    u64 *red_lo = (u64 *)(red_hi + NW);
    long long *red_own = (long long *)(red_lo + NW);
    int32_t *red_com = (int32_t *)(red_own + NW);
    int32_t *red_have = red_com + NW;
    const int64_t cnt = cnt_dev ? (*cnt_dev < cnt_max ? *cnt_dev : cnt_max) : cnt_max;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int t = tid; t < SLOTS; t += NT) { keys[t] = EMPTY; vals[t] = 0; }
    __syncthreads();
    for (int64_t pos = blockIdx.x; pos < cnt; pos += gridDim.x) {
        const NodeRec rec = list[pos];
        const int32_t i = rec.node;
        const int32_t a = com[i];
        const long long ki = kdeg[i];
        const int lgt = table_lg(rec.deg);
        const int lg = lgt < LGS ? lgt : LGS;        // (the host's classes guarantee lgt <= LGS; never index past the LDS table)
        const uint32_t mask = (1u << lg) - 1u;
        const int32_t *__restrict__ ci = indices + rec.start;
        const int64_t *__restrict__ qi = q + rec.start;
        long long own = 0;
        // (four CONSECUTIVE entries per thread as 16-byte vectors: one load of ids and two of weights instead of four each -- the
        // texture addresser charges per instruction; the contender rule of k_lv_decide_g was tried here too: a second pass over the
        // table and two more barriers per node cost what the saved gathers of tot[] gained -- 172 -> 180 us per launch, not kept)
        for (int base = 4 * tid; base < rec.deg; base += 4 * NT) {
            const i32x4_u jv = *(const i32x4_u *)(ci + base);
            const i64x2_u w0 = *(const i64x2_u *)(qi + base), w1 = *(const i64x2_u *)(qi + base + 2);
            int32_t j[4];
            u64 qv[4];
            int32_t c[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const bool ok = base + u < rec.deg;
                j[u] = ok ? jv[u] : i;
                qv[u] = ok ? (u64)(u < 2 ? w0[u] : w1[u - 2]) : 0;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) c[u] = j[u] != i ? com[j[u]] : EMPTY;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (c[u] == EMPTY) continue;
                if (c[u] == a) { own += (long long)qv[u]; continue; }
                uint32_t slot = hash_slot(c[u], lg);
                for (;;) {
                    const int32_t prev = atomicCAS(&keys[slot], EMPTY, c[u]);
                    if (prev == EMPTY || prev == c[u]) { atomicAdd(&vals[slot], qv[u]); break; }
                    slot = (slot + 1) & mask;
                }
            }
        }
        __syncthreads();
        Best b;
        b.gain = 0; b.com = a; b.have = 0;
        for (uint32_t t0 = tid; t0 <= mask; t0 += 4 * NT) {
            int32_t c[4];
            u64 d[4];
            long long tc[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t t = t0 + u * NT;
                c[u] = t <= mask ? keys[t] : EMPTY;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t t = t0 + u * NT;
                d[u] = c[u] != EMPTY ? vals[t] : 0;
                tc[u] = c[u] != EMPTY ? tot[c[u]] : 0;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (c[u] == EMPTY) continue;
                const uint32_t t = t0 + u * NT;
                keys[t] = EMPTY;
                vals[t] = 0;
                const i128 g = gain128((u64)d[u], m2, (u64)tc[u], (u64)ki);
                best_take(b, g, c[u]);
            }
        }
        b = best_wave_reduce(b);
        own = wave_sum_ll(own);
        if (lane == 0) {
            red_hi[wv] = (long long)(b.gain >> 64);
            red_lo[wv] = (u64)b.gain;
            red_com[wv] = b.com;
            red_have[wv] = b.have;
            red_own[wv] = own;
        }
        __syncthreads();
        if (tid == 0) {
            long long o = 0;
            Best r;
            r.gain = 0; r.com = a; r.have = 0;
            for (int w = 0; w < NW; w++) {
                o += red_own[w];
                const i128 g = ((i128)red_hi[w] << 64) | (i128)red_lo[w];
                if (red_have[w] && (!r.have || g > r.gain || (g == r.gain && red_com[w] < r.com))) { r.gain = g; r.com = red_com[w]; r.have = 1; }
            }
            // staying: i removed from its own community first (python-louvain __remove)
            const i128 g = gain128((u64)o, m2, (u64)(tot[a] - ki), (u64)ki);
            best_take(r, g, a);
            int32_t best = r.have ? r.com : a;
            if (best != a && csize[a] == 1 && csize[best] == 1 && best > a) best = a;
            nxt[i] = best;
        }
        __syncthreads();          // the table is clear again and the reduction slots are free
    }
}
template <int LGS, int NT>
constexpr size_t lv_wg_lds() { return (size_t)(1 << LGS) * 12 + (size_t)(NT / 64) * 32; }

// pass 1 of the mega rows with the chunk's entries summed per community in LDS first (round 4): from the second sweep on most
// entries of a mega row point into a few large communities, and 4096 lanes adding to the same few slots of the node's table in
// global memory serialise in the L2 (the surveyed law, level 1: 0.3 ms per sub-round in the first sweep, 1.2 ms in the last ones).
// One compare-and-swap insertion per DISTINCT community of the chunk reaches the global table.  Integer sums: same table content.
constexpr int MEGA_LDS_LG = 13;            // table_lg(MEGA_CHUNK)
__global__ void __launch_bounds__(MEGA_NT) k_lv_mega_insert_lds(const NodeRec *__restrict__ list, int chunks, const int32_t *__restrict__ indices,
                                                                 const int64_t *__restrict__ q, const int32_t *__restrict__ com,
                                                                 const int64_t *__restrict__ goff, int32_t *__restrict__ gkeys,
                                                                 u64 *__restrict__ gvals, u64 *__restrict__ own_acc, const int64_t *__restrict__ cnt_dev = nullptr)
{
    constexpr int SLOTS = 1 << MEGA_LDS_LG;
    extern __shared__ __align__(16) unsigned char lv_wg_smem[];
    u64 *lvals = (u64 *)lv_wg_smem;
    int32_t *lkeys = (int32_t *)(lvals + SLOTS);
    long long *red = (long long *)(lkeys + SLOTS);
    const int m = blockIdx.x / chunks, ch = blockIdx.x % chunks;
    if (cnt_dev && m >= *cnt_dev) return;                            // (the list was thinned on the device: k_lv_prune)
    const NodeRec rec = list[m];
    if ((int64_t)ch * MEGA_CHUNK >= rec.deg) return;                 // (uniform per workgroup)
    for (int t = threadIdx.x; t < SLOTS; t += MEGA_NT) { lkeys[t] = EMPTY; lvals[t] = 0; }
    __syncthreads();
    const int32_t i = rec.node;
    const int32_t a = com[i];
    const int64_t end = rec.start + rec.deg;
    const int64_t base = rec.start + (int64_t)ch * MEGA_CHUNK + threadIdx.x;
    int32_t j[4];
    u64 qv[4];
    int32_t c[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int64_t k = base + (int64_t)t * MEGA_NT;
        const bool ok = k < end;
        j[t] = ok ? indices[k] : i;
        qv[t] = ok ? (u64)q[k] : 0;
    }
#pragma unroll
    for (int t = 0; t < 4; t++) c[t] = j[t] != i ? com[j[t]] : EMPTY;
    long long own = 0;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        if (c[t] == EMPTY) continue;
        if (c[t] == a) { own += (long long)qv[t]; continue; }
        uint32_t slot = hash_slot(c[t], MEGA_LDS_LG);
        for (;;) {
            const int32_t prev = atomicCAS(&lkeys[slot], EMPTY, c[t]);
            if (prev == EMPTY || prev == c[t]) { atomicAdd(&lvals[slot], qv[t]); break; }
            slot = (slot + 1) & (SLOTS - 1);
        }
    }
    __syncthreads();
    const int lg = table_lg(rec.deg);
    const uint32_t mask = (1u << lg) - 1u;
    int32_t *keys = gkeys + goff[i];
    u64 *vals = gvals + goff[i];
    for (int t = threadIdx.x; t < SLOTS; t += MEGA_NT) {
        const int32_t cc = lkeys[t];
        if (cc == EMPTY) continue;
        const u64 d = lvals[t];
        uint32_t slot = hash_slot(cc, lg);
        for (;;) {
            const int32_t prev = atomicCAS(&keys[slot], EMPTY, cc);
            if (prev == EMPTY || prev == cc) { atomicAdd(&vals[slot], d); break; }
            slot = (slot + 1) & mask;
        }
    }
    block_atomic_add_u64(own, own_acc + m, red);
}

// Small levels (the induced graphs: n <= CAPD nodes, often dense rows): the weight towards every community goes into a
// DIRECT table indexed by the community id in the wave's LDS -- no hashing, no compare-and-swap, and no global-memory
// tables for rows of more than CAP/2 entries.  Same gains, same arg-max (ties to the lowest id) as k_lv_decide.
template <int CAPD>
__global__ void __launch_bounds__(BLK) k_lv_decide_direct(int64_t n, int n_sub, int round, u64 m2,
                                                           const int64_t *__restrict__ indptr,
                                                           const int32_t *__restrict__ indices,
                                                           const int64_t *__restrict__ q, const int32_t *__restrict__ ids,
                                                           const int64_t *__restrict__ kdeg,
                                                           const int64_t *__restrict__ tot, const int32_t *__restrict__ com,
                                                           const int32_t *__restrict__ csize, int32_t *__restrict__ nxt)
{
    __shared__ u64 svals[WPB][CAPD];
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int64_t wave = (int64_t)blockIdx.x * WPB + wib;
    const int64_t nwave = (int64_t)gridDim.x * WPB;
    u64 *lv = svals[wib];
    for (int t = lane; t < CAPD; t += 64) lv[t] = 0;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    for (int64_t i = wave; i < n; i += nwave) {
        const int32_t id = ids ? ids[i] : (int32_t)i;
        if (bucket_of(id, n_sub) != (uint32_t)round) continue;
        const long long ki = kdeg[i];
        if (ki == 0) continue;
        const int64_t s = indptr[i], e = indptr[i + 1];
        const int32_t a = com[i];
        long long own = 0;
        for (int64_t base = s; base < e; base += 256) {
            int32_t j[4];
            u64 qv[4];
            int32_t c[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int64_t k = base + t * 64 + lane;
                const bool ok = k < e;
                j[t] = ok ? indices[k] : (int32_t)i;
                qv[t] = ok ? (u64)q[k] : 0;
            }
#pragma unroll
            for (int t = 0; t < 4; t++) c[t] = j[t] != (int32_t)i ? com[j[t]] : EMPTY;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                if (c[t] == EMPTY) continue;
                if (c[t] == a) own += (long long)qv[t];
                else atomicAdd(&lv[c[t]], qv[t]);
            }
        }
        own = wave_sum_ll(own);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        Best b;
        b.gain = 0; b.com = a; b.have = 0;
        for (int32_t c = lane; c < (int32_t)n; c += 64) {
            const u64 d = lv[c];
            if (d) {
                lv[c] = 0;
                const i128 g = gain128((u64)d, m2, (u64)tot[c], (u64)ki);
                best_take(b, g, c);
            }
        }
        if (lane == 0) {
            const i128 g = gain128((u64)own, m2, (u64)(tot[a] - ki), (u64)ki);
            best_take(b, g, a);
        }
        b = best_wave_reduce(b);
        if (lane == 0) {
            int32_t best = b.have ? b.com : a;
            if (best != a && csize[a] == 1 && csize[best] == 1 && best > a) best = a;
            nxt[i] = best;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
}

// ---------------------------------------------------------------------------------------------
// Grouped, software-pipelined form of the neighbour scan for nodes of at most LPR*UN entries (all but the
// hubs).  LPR lanes own a node and hold its UN entries per lane in registers; while the community gathers of
// node t are in flight, the record, the entries and the own-community of the group's next node are already
// being fetched.  After the first sweeps nearly every neighbour sits in the node's own community, which is
// summed in registers; only groups that saw a foreign community take turns on the wave's LDS table.
// The node list of a sub-round is fixed for a level (the bucket depends on the id only), so it is built once.
// ---------------------------------------------------------------------------------------------
// Node lists of one level, all at once.  Without hubs: segment r = the nodes of sub-round r.  With hubs every sub-round has
// LV_SEGS adjacent segments by row length: [0] at most thr[0] entries (grouped kernel), [1] at most thr[1] (wave-per-node kernel),
// [2 .. 5] at most thr[2 .. 5] (workgroup-per-node kernel with a table of 1024 / 2048 / 4096 / 8192 slots), [6] the MEGA rows beyond
// (workgroups per node, table in global memory) -- adjacent, so that the passes that treat them alike (apply, internal-weight
// delta) cover a sub-round in ONE launch.  A node sets the flag of its segment; ONE exclusive scan over the n_seg * nloc flags
// gives every listed node its position in the common list and, at the segment starts, the list boundaries.  Inside a segment
// the nodes ascend, as they did when every segment had its own scan.
constexpr int LV_SEGS = 7;
struct LvThr { int t[LV_SEGS - 1]; };
__device__ __forceinline__ int lv_list_segment(int64_t i, int n_sub, LvThr thr, int with_hubs, const int64_t *__restrict__ indptr,
                                               const int32_t *__restrict__ ids, const int64_t *__restrict__ kdeg)
{
    if (kdeg[i] == 0) return -1;
    const int64_t deg = indptr[i + 1] - indptr[i];
    const int32_t id = ids ? ids[i] : (int32_t)i;
    const int b = (int)bucket_of(id, n_sub);
    if (deg <= thr.t[0]) return with_hubs ? LV_SEGS * b : b;
    if (!with_hubs) return -1;
    int c = 1;
#pragma unroll
    for (int k = 1; k < LV_SEGS - 1; k++) c += deg > thr.t[k] ? 1 : 0;
    return LV_SEGS * b + c;
}

__global__ void k_lv_list_flags(int64_t r0, int64_t r1, int n_sub, LvThr thr, int with_hubs,
                                const int64_t *__restrict__ indptr, const int32_t *__restrict__ ids,
                                const int64_t *__restrict__ kdeg, int64_t *__restrict__ flag)     // zeroed by the caller
{
    const int64_t i = r0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= r1) return;
    const int seg = lv_list_segment(i, n_sub, thr, with_hubs, indptr, ids, kdeg);
    if (seg >= 0) flag[(int64_t)seg * (r1 - r0) + (i - r0)] = 1;
}

__global__ void k_lv_list_fill(int64_t r0, int64_t r1, int n_sub, LvThr thr, int with_hubs,
                               const int64_t *__restrict__ indptr, const int32_t *__restrict__ ids,
                               const int64_t *__restrict__ kdeg, const int64_t *__restrict__ pos, NodeRec *__restrict__ list)
{
    const int64_t i = r0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= r1) return;
    const int seg = lv_list_segment(i, n_sub, thr, with_hubs, indptr, ids, kdeg);
    if (seg < 0) return;
    NodeRec rec;
    rec.node = (int32_t)i;
    rec.start = indptr[i];
    rec.deg = (int32_t)(indptr[i + 1] - rec.start);
    list[pos[(int64_t)seg * (r1 - r0) + (i - r0)]] = rec;
}

__global__ void k_lv_list_bounds(int n_seg, int64_t nloc, const int64_t *__restrict__ pos, int64_t *__restrict__ bnd)
{
    for (int t = threadIdx.x; t <= n_seg; t += blockDim.x) bnd[t] = pos[(int64_t)t * nloc];
}

// The grouped neighbour scan gathers ONE 4-byte label per stored neighbour and nothing else (round 5).
//
// What the scan needs of a neighbour j is its community c = com[j] and, for the gain d_c * 2m - tot_c * k_i, the total of c.  Round 4
// gathered both as one 16-byte record per neighbour; on the contract workload (the surveyed genome-size law: one genome of 383k
// contigs, a uniform random graph of 180 neighbours per node) those records are 6 MB against the 4 MB of an XCD's L2 and the counters
// see 4.4 GB cross the fabric per sub-round for 0.81 GB of algorithmic bytes -- 34 M operand lines of 128 bytes fetched for 16 of them
// each (profiles/r05_c3_FETCH_SIZE.csv): the kernel runs at the fabric's rate on wasted lines.  The labels alone are 4 bytes per node
// (the whole level: 3.9 MB, the giant genome: 1.5 MB): they stay in L2.
//
// And the totals are needed for a FEW candidates only, exactly:  a candidate's gain is at most d_c * 2m (totals are not negative), so with
// D = the largest d_c of the node and any band b >= 0 the candidates with d_c < D - b ("non-contenders") have gains below
// X = (D - b) * 2m.  The contenders (d_c >= D - b) get their exact totals (one gather each, by the lane that claimed the slot) and
// exact 128-bit gains; if the best of {contenders, staying, 0} reaches X, no non-contender can win or tie and the decision is the
// exact arg-max -- else (the leader sits in a community so large that its tot * k term eats the band: rare, and never wrong) the
// non-contenders are evaluated as well.  b only steers how often that happens: (k_i >> band_shift) + 1, i.e. the tot * k / 2m of a
// community of about 32 average nodes.  On a metagenome d_c decides almost alone (tot_c * k_i / 2m is 10^-3 of d_c at level 0), so a
// node has one or two contenders among its ~100 candidate communities: the 128-bit gain arithmetic of the others is gone as well.
template <int LPR, int UN>
__global__ void __launch_bounds__(BLK) __attribute__((amdgpu_waves_per_eu(LPR == 64 && UN <= 4 ? 5 : 2))) k_lv_decide_g(
    int64_t cnt_max, const int64_t *__restrict__ cnt_dev, const NodeRec *__restrict__ list, u64 m2, const int32_t *__restrict__ indices,
    const int64_t *__restrict__ q, const int64_t *__restrict__ kdeg, const int64_t *__restrict__ tot, const int32_t *__restrict__ com,
    const int32_t *__restrict__ csize, int32_t *__restrict__ nxt, int64_t *__restrict__ own_out, int32_t *__restrict__ stamp_out,
    int32_t epoch, int xcd, int band_shift, int stream_nt)
{
    constexpr int CAP = LPR * UN > 384 ? 1024 : 512;       // >= 1.25 * LPR * UN + 1
    constexpr int LG = LPR * UN > 384 ? 10 : 9;
    static_assert(LPR * UN * 5 / 4 + 1 <= CAP, "table too small");
    // the list may have been thinned on the device (k_lv_prune): its length then sits in device memory
    const int64_t cnt = cnt_dev ? (*cnt_dev < cnt_max ? *cnt_dev : cnt_max) : cnt_max;
    __shared__ int32_t skeys[WPB][CAP];
    __shared__ u64 svals[WPB][CAP];
    constexpr int RPW = 64 / LPR;
    constexpr u64 GMASK = LPR == 64 ? ~0ull : ((1ull << LPR) - 1ull);
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // (uniform: list positions of a one-node-per-wave shape are scalars)
    const int sub = LPR == 64 ? 0 : lane / LPR, sl = LPR == 64 ? lane : lane % LPR;
    int32_t *keys = skeys[wib];
    u64 *vals = svals[wib];
    for (int t = lane; t < CAP; t += 64) { keys[t] = EMPTY; vals[t] = 0; }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");

    // XCD-aware numbering (round 4): workgroup b runs on XCD b mod 8, and every XCD has its own 4 MB L2.  With xcd set (a grid of a
    // multiple of eight workgroups) XCD k walks the k-th EIGHTH of the list -- the list is in row order, i.e. in locality order, so
    // the labels its gathers touch are one slice of that vector, which its L2 can hold -- instead of every XCD seeing all of it.
    int64_t step = (int64_t)gridDim.x * WPB * RPW;
    int64_t pos = ((int64_t)blockIdx.x * WPB + wib) * RPW + sub;
    int64_t cnt_end = cnt;
    if (xcd) {
        const int64_t k = blockIdx.x & 7, per = (cnt + 7) >> 3;
        step = (int64_t)(gridDim.x >> 3) * WPB * RPW;
        pos = k * per + ((int64_t)(blockIdx.x >> 3) * WPB + wib) * RPW + sub;
        cnt_end = (k + 1) * per < cnt ? (k + 1) * per : cnt;
    }
    // What a CU's texture addresser charges is the INSTRUCTION: ~26 cycles per vector-memory instruction of a wave whatever it
    // moves -- 4 bytes per lane or 16, one active lane or 64 (tools/probes/ta_probe.hip, round 5) -- and with its gathers in L2 this
    // kernel sat exactly there: ~23 vector loads per node x 977 nodes per CU x 26 cycles = its 250 us.  Hence: a lane owns UN
    // CONSECUTIVE entries of the row and fetches them as 16-byte vectors (ids: UN / 4 loads, weights: UN / 2, instead of UN each); with
    // one node per wave everything that belongs to the NODE -- its list record, label, degree, its community's total and size -- has
    // a wave-uniform address and comes through the scalar cache, which the addresser never sees.  Left per node: the UN label gathers,
    // the contenders' totals, one store.
    //
    // Software pipeline.  Once its gathers stay in L2 the kernel waits on its dependent round trips -- node record -> entries ->
    // labels of the neighbours -> totals of the contenders -- with a handful of waves per SIMD to hide them.  One node per wave
    // (LPR == 64: DEEP): the first three belong to LATER nodes while the current one is worked on --
    //   cur  node (i, a, ki, ta) with the labels cm[] and weights qv[] of its neighbours in registers,
    //   nx1  record, own label / degree and neighbour ids j1[] of the node after it   (its labels are requested in this pass),
    //   nx2  record of the one after that                                               (its entries are requested in this pass).
    // Several nodes per wave (shorter rows, eight entries per lane): the groups of a wave are that many gather streams already and the
    // registers are spoken for, so the node whose labels were just requested is the one worked on (nx1 IS the current node).
    constexpr bool DEEP = LPR == 64 && UN <= 4;
    auto load_rec = [&](int64_t p) -> NodeRec {
        NodeRec r;
        r.node = -1; r.deg = 0; r.start = 0;
        if (p < cnt_end) r = list[p];
        return r;
    };
    // entries UN * sl .. UN * sl + UN - 1 of a row as 16-byte vectors (a lane past the row's end re-reads its beginning and drops it;
    // the streams carry 32 bytes of slack behind their last entry)
    auto load_ids = [&](const NodeRec &r, int32_t (&jj)[UN]) {
        const int eb = UN * sl < r.deg ? UN * sl : 0;
#pragma unroll
        for (int v = 0; v < UN / 4; v++) {
            const i32x4_u x = lv_stream((const i32x4_u *)(indices + r.start + eb + 4 * v), stream_nt);
#pragma unroll
            for (int k = 0; k < 4; k++) jj[4 * v + k] = UN * sl + 4 * v + k < r.deg ? x[k] : r.node;
        }
    };
    auto load_q = [&](const NodeRec &r, u64 (&qq)[UN]) {
        const int eb = UN * sl < r.deg ? UN * sl : 0;
#pragma unroll
        for (int v = 0; v < UN / 2; v++) {
            const i64x2_u y = lv_stream((const i64x2_u *)(q + r.start + eb + 2 * v), stream_nt);
#pragma unroll
            for (int k = 0; k < 2; k++) qq[2 * v + k] = UN * sl + 2 * v + k < r.deg ? (u64)y[k] : 0;
        }
    };
    NodeRec rec1 = load_rec(pos + (DEEP ? step : 0)), rec2 = load_rec(pos + (DEEP ? 2 : 1) * step);
    int32_t i = -1, a = 0;                    // DEEP: the node at `pos`, its labels and weights loaded
    long long ki = 0, ta = 0;
    int32_t cm[DEEP ? UN : 1];
    u64 qv[DEEP ? UN : 1];
    int32_t a1 = 0;
    long long ki1 = 0;
    int32_t j1[UN];
    {
        NodeRec rec0;
        rec0.node = -1; rec0.deg = 0; rec0.start = 0;
        if constexpr (DEEP) rec0 = load_rec(pos);
        i = rec0.node;
        if (i >= 0) { a = com[i]; ki = kdeg[i]; }
        if (rec1.node >= 0) { a1 = com[rec1.node]; ki1 = kdeg[rec1.node]; }
        load_ids(rec1, j1);
        if constexpr (DEEP) {
            int32_t j0[UN];
            load_ids(rec0, j0);
            load_q(rec0, qv);
#pragma unroll
            for (int t = 0; t < UN; t++) cm[t] = j0[t] != i && j0[t] >= 0 ? com[j0[t]] : EMPTY;
            if (i >= 0) ta = tot[a];
        }
    }
    auto group_best = [&](Best v) -> Best {
        if constexpr (LPR == 64) return best_wave_reduce(v);
        else {
#pragma unroll
            for (int o = LPR / 2; o > 0; o >>= 1) {
                long long hi = __shfl_xor((long long)(v.gain >> 64), o, LPR);
                u64 lo = (u64)__shfl_xor((long long)(u64)v.gain, o, LPR);
                int32_t c = __shfl_xor(v.com, o, LPR);
                int32_t hv = __shfl_xor(v.have, o, LPR);
                i128 g = ((i128)hi << 64) | (i128)lo;
                if (hv && (!v.have || g > v.gain || (g == v.gain && c < v.com))) {
                    v.gain = g;
                    v.com = c;
                    v.have = 1;
                }
            }
            return v;
        }
    };
    // the loop runs while ANY group of the wave still has a node (groups take turns on the shared table)
    while (__ballot((DEEP ? i : rec1.node) >= 0)) {
        // 1. record of the node furthest ahead
        // (all loads of this stretch are UNCONDITIONAL -- a lane without an entry reads a harmless place and drops the value --
        // so that they issue back to back instead of one exec-masked branch each)
        const NodeRec rec3 = load_rec(pos + (DEEP ? 3 : 2) * step);
        // 2. labels of nx1's neighbours, its weights and its community's total
        int32_t cm1[UN];
        u64 qv1[UN];
#pragma unroll
        for (int t = 0; t < UN; t++) {
            const int32_t jj = j1[t] >= 0 ? j1[t] : 0;
            const int32_t cv = com[jj];
            cm1[t] = j1[t] != rec1.node ? cv : EMPTY;
        }
        load_q(rec1, qv1);
        const long long ta1 = tot[a1];                  // (a1 is a valid community id, 0 for a lane group without a node)
        // 3. neighbour ids, own label and degree of nx2
        int32_t j2[UN];
        const int32_t n2 = rec2.node >= 0 ? rec2.node : 0;
        const int32_t a2 = com[n2];
        const long long ki2 = rec2.node >= 0 ? kdeg[n2] : 0;
        load_ids(rec2, j2);
        // the node worked on in this pass (register renames, no copies)
        const int32_t wi = DEEP ? i : rec1.node, wa = DEEP ? a : a1;
        const long long wki = DEEP ? ki : ki1, wta = DEEP ? ta : ta1;
        int32_t wcm[UN];
        u64 wqv[UN];
#pragma unroll
        for (int t = 0; t < UN; t++) {
            if constexpr (DEEP) { wcm[t] = cm[t]; wqv[t] = qv[t]; }
            else { wcm[t] = cm1[t]; wqv[t] = qv1[t]; }
        }
        // 4. own-community weight in registers; does the group see any foreign community?
        long long own = 0;
        bool foreign = false;
#pragma unroll
        for (int t = 0; t < UN; t++) {
            if (wcm[t] == EMPTY) continue;
            if (wcm[t] == wa) own += (long long)wqv[t];
            else foreign = true;
        }
        if constexpr (LPR == 64) own = wave_sum_ll_all(own);
        else {
#pragma unroll
            for (int o = LPR / 2; o > 0; o >>= 1) own += __shfl_xor(own, o, LPR);
        }
        const u64 fmask = __ballot(foreign);
        // 5. groups with foreign neighbours take the wave's table one after the other: insert, then every lane reads back (and clears)
        //    the slots it claimed -- a lane that turned an EMPTY slot into a community holds that community's summed weight d_c
        int32_t cs[UN];                        // slots this lane claimed (-1: none)
        u64 dv[UN];
#pragma unroll
        for (int t = 0; t < UN; t++) { cs[t] = -1; dv[t] = 0; }
        u64 dmax = 0;
        for (int gsel = 0; gsel < RPW; gsel++) {
            if (!(fmask & (GMASK << (gsel * LPR)))) continue;                     // wave-uniform
            if (sub == gsel) {
#pragma unroll
                for (int t = 0; t < UN; t++) {
                    if (wcm[t] != EMPTY && wcm[t] != wa) {
                        uint32_t slot = hash_slot(wcm[t], LG);
                        for (;;) {
                            const int32_t prev = atomicCAS(&keys[slot], EMPTY, wcm[t]);
                            if (prev == EMPTY || prev == wcm[t]) {
                                atomicAdd(&vals[slot], wqv[t]);
                                if (prev == EMPTY) cs[t] = (int32_t)slot;
                                break;
                            }
                            slot = (slot + 1) & (CAP - 1);
                        }
                    }
                }
            }
            // (the table is this WAVE's: its LDS operations complete in order, so a wavefront-scope fence -- compiler ordering only -- is all the
            // read-back needs.  A workgroup-scope fence also waits for every global load in flight (s_waitcnt vmcnt(0)): the prefetched
            // records, entries and labels of the NEXT nodes -- the whole software pipeline stalled on it once per node (round 5, from the ISA))
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            if (sub == gsel) {
#pragma unroll
                for (int t = 0; t < UN; t++) {
                    if (cs[t] == -1) continue;
                    dv[t] = vals[cs[t]];
                    keys[cs[t]] = EMPTY;
                    vals[cs[t]] = 0;
                    dmax = dv[t] > dmax ? dv[t] : dmax;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        }
        // 6. contenders: the candidates within the band of the node's largest d_c get their exact totals and gains
        const u64 D = group_max_u64<LPR>(dmax);
        const u64 band = ((u64)wki >> band_shift) + 1ull;
        const u64 thr = D > band ? D - band : 0ull;
        Best b;
        b.gain = 0; b.com = wa; b.have = 0;
        if constexpr (LPR == 64) {
            // One node per wave: the handful of contenders (one or two of a node's ~100 candidate communities) are walked by a UNIFORM
            // loop over the ballots of the contender flags -- label, weight and total of each come out of its lane by v_readlane, the
            // 128-bit gain and the running arg-max are scalar arithmetic.  As per-lane vector code (four slots of 128-bit products per
            // lane, then five DPP reductions for the arg-max over the wave) this stretch was 600 of the loop's 1300 instructions.
            // (a contender's total: its community id is uniform here, so the load is a SCALAR one -- the texture addresser, which
            // bounds this kernel, never sees it)
            u64 cmask[UN];
#pragma unroll
            for (int t = 0; t < UN; t++) cmask[t] = __ballot(cs[t] != -1 && dv[t] >= thr);
            const u64 ki_u = ((u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((u64)wki >> 32)) << 32) |
                             (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u64)wki);
            const int32_t a_u = __builtin_amdgcn_readfirstlane(wa);
            const u64 ta_u = ((u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((u64)wta >> 32)) << 32) |
                             (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u64)wta);      // (lane 0 = sl 0 holds tot[a])
            i128 bg = 0;
            int32_t bc = a_u;
            int bh = 0;
#pragma unroll
            for (int t = 0; t < UN; t++) {
                u64 mk = cmask[t];
                while (mk) {
                    const int l = __ffsll((unsigned long long)mk) - 1;
                    mk &= mk - 1;
                    const int32_t c = __builtin_amdgcn_readlane(wcm[t], l);
                    const u64 d = ((u64)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(dv[t] >> 32), l) << 32) |
                                  (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)dv[t], l);
                    const u64 tt = (u64)tot[c];
                    const i128 g = gain128(d, m2, tt, ki_u);
                    if (g > 0 && (!bh || g > bg || (g == bg && c < bc))) { bg = g; bc = c; bh = 1; }
                }
            }
            if (wi >= 0) {       // (wave-uniform: one node per wave)
                // staying: i removed from its own community first (python-louvain __remove)
                const i128 g = gain128((u64)own, m2, ta_u - ki_u, ki_u);
                if (g > 0 && (!bh || g > bg || (g == bg && a_u < bc))) { bg = g; bc = a_u; bh = 1; }
                if (own_out && sl == 0) { own_out[wi] = own; stamp_out[wi] = epoch; }      // row-sharded pruning: own_i as of this sub-round's start
            }
            b.gain = bg; b.com = bc; b.have = bh;       // (the same in every lane)
        } else {
            {
                long long tc[UN];
#pragma unroll
                for (int t = 0; t < UN; t++) tc[t] = (cs[t] != -1 && dv[t] >= thr) ? tot[wcm[t]] : 0;
#pragma unroll
                for (int t = 0; t < UN; t++) {
                    if (!(cs[t] != -1 && dv[t] >= thr)) continue;
                    const i128 g = gain128(dv[t], m2, (u64)tc[t], (u64)wki);
                    best_take(b, g, wcm[t]);
                }
            }
            if (wi >= 0 && sl == 0) {
                // staying: i removed from its own community first (python-louvain __remove)
                const i128 g = gain128((u64)own, m2, (u64)(wta - wki), (u64)wki);
                best_take(b, g, wa);
                if (own_out) { own_out[wi] = own; stamp_out[wi] = epoch; }      // row-sharded pruning: own_i as of this sub-round's start
            }
            b = group_best(b);
        }
        // every non-contender's gain is below thr * 2m: settled unless the best so far (0 when nothing gains) stays under that
        if (thr != 0ull) {
            const i128 x128 = (i128)((unsigned __int128)thr * (unsigned __int128)m2);
            const i128 top = b.have ? b.gain : (i128)0;
            if (top < x128) {                                   // group-uniform (b is the group's after the reduction)
                long long tc[UN];
#pragma unroll
                for (int t = 0; t < UN; t++) tc[t] = (cs[t] != -1 && dv[t] < thr) ? tot[wcm[t]] : 0;
#pragma unroll
                for (int t = 0; t < UN; t++) {
                    if (!(cs[t] != -1 && dv[t] < thr)) continue;
                    const i128 g = gain128(dv[t], m2, (u64)tc[t], (u64)wki);
                    best_take(b, g, wcm[t]);
                }
                b = group_best(b);
            }
        }
        if (wi >= 0 && sl == 0) {
            int32_t best = b.have ? b.com : wa;
            if (best > wa && csize[wa] == 1 && csize[best] == 1) best = wa;      // (best > a implies best != a: the sizes are looked at for upward moves only)
            nxt[wi] = best;
        }
        // 7. advance the pipeline
        pos += step;
        if constexpr (DEEP) {
            i = rec1.node;
            a = a1;
            ki = ki1;
            ta = ta1;
#pragma unroll
            for (int t = 0; t < UN; t++) { cm[t] = cm1[t]; qv[t] = qv1[t]; }
        }
#pragma unroll
        for (int t = 0; t < UN; t++) j1[t] = j2[t];
        rec1 = rec2;
        a1 = a2;
        ki1 = ki2;
        rec2 = rec3;
    }
}

// The FIRST sub-round of a level, for the nodes of the grouped list: every node is alone, so every stored neighbour j != i IS a
// candidate community of its own -- weight towards it q_ij, total kdeg[j], size 1 -- the own community holds nothing else (stay
// gain 0, not taken) and the singleton rule lets i move only to a LOWER community id.  The decision is the arg-max over the
// entries of q_ij * 2m - kdeg[j] * k_i (ties to the lowest community id): no table.  Round 5: and no gather per entry either -- the
// contender rule of k_lv_decide_g with d_c = q_ij: only the entries within the band of the row's largest weight fetch their
// neighbour's label and degree (round 4 gathered both for every entry: 5.9 GB across the fabric for this one launch on the contract
// workload, the 8 MB degree vector against 4 MB of L2), the others are settled by q_ij * 2m < (Q - band) * 2m <= the best exact gain.
template <int LPR, int UN>
__global__ void __launch_bounds__(BLK) k_lv_first_g(int64_t cnt, const NodeRec *__restrict__ list, u64 m2,
                                                     const int32_t *__restrict__ indices, const int64_t *__restrict__ q,
                                                     const int64_t *__restrict__ kdeg, const int32_t *__restrict__ com,
                                                     const int32_t *__restrict__ csize, int32_t *__restrict__ nxt, int xcd, int band_shift)
{
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR, sl = lane % LPR;
    int64_t step = (int64_t)gridDim.x * WPB * RPW;
    int64_t pos = ((int64_t)blockIdx.x * WPB + (threadIdx.x >> 6)) * RPW + sub;
    if (xcd) {                                   // XCD k walks the k-th eighth of the list (see k_lv_decide_g)
        const int64_t k = blockIdx.x & 7, per = (cnt + 7) >> 3;
        step = (int64_t)(gridDim.x >> 3) * WPB * RPW;
        pos = k * per + ((int64_t)(blockIdx.x >> 3) * WPB + (threadIdx.x >> 6)) * RPW + sub;
        cnt = (k + 1) * per < cnt ? (k + 1) * per : cnt;
    }
    // software pipeline: record, entries and own label of the group's NEXT node are requested while the current one is evaluated
    int32_t i = -1, a = 0;
    int deg = 0;
    long long ki = 0;
    int32_t j[UN];
    u64 qv[UN];
    if (pos < cnt) {
        const NodeRec rec = list[pos];
        i = rec.node;
        deg = rec.deg;
        a = com[i];
        ki = kdeg[i];
#pragma unroll
        for (int t = 0; t < UN; t++) {
            const int e = t * LPR + sl;
            j[t] = e < deg ? indices[rec.start + e] : i;
            qv[t] = e < deg ? (u64)q[rec.start + e] : 0;
        }
    }
    auto group_best = [&](Best v) -> Best {
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) {
            long long hi = __shfl_xor((long long)(v.gain >> 64), o, LPR);
            u64 lo = (u64)__shfl_xor((long long)(u64)v.gain, o, LPR);
            int32_t c = __shfl_xor(v.com, o, LPR);
            int32_t hv = __shfl_xor(v.have, o, LPR);
            i128 g = ((i128)hi << 64) | (i128)lo;
            if (hv && (!v.have || g > v.gain || (g == v.gain && c < v.com))) {
                v.gain = g;
                v.com = c;
                v.have = 1;
            }
        }
        return v;
    };
    while (__ballot(i >= 0)) {
        const int64_t posn = pos + step;
        NodeRec recn;
        recn.node = -1; recn.deg = 0; recn.start = 0;
        if (i >= 0 && posn < cnt) recn = list[posn];
        // contenders of the current node: the entries within the band of its largest weight
        u64 qmax = 0;
#pragma unroll
        for (int t = 0; t < UN; t++) qmax = (i >= 0 && j[t] != i && qv[t] > qmax) ? qv[t] : qmax;
        const u64 Q = group_max_u64<LPR>(qmax);
        const u64 band = ((u64)ki >> band_shift) + 1ull;
        const u64 thr = Q > band ? Q - band : 0ull;
        int32_t cm[UN];
        long long tv[UN];
#pragma unroll
        for (int t = 0; t < UN; t++) {
            const bool nb = i >= 0 && j[t] != i && qv[t] >= thr;
            cm[t] = nb ? com[j[t]] : EMPTY;
            tv[t] = nb ? kdeg[j[t]] : 0;
        }
        int32_t an = 0;
        long long kin = 0;
        int32_t j2[UN];
        u64 q2[UN];
        {
            const int32_t inn = recn.node;
            if (inn >= 0) { an = com[inn]; kin = kdeg[inn]; }
#pragma unroll
            for (int t = 0; t < UN; t++) {
                const int e = t * LPR + sl;
                j2[t] = e < recn.deg ? indices[recn.start + e] : inn;
                q2[t] = e < recn.deg ? (u64)q[recn.start + e] : 0;
            }
        }
        Best b;
        b.gain = 0; b.com = a; b.have = 0;
#pragma unroll
        for (int t = 0; t < UN; t++) {
            if (cm[t] == EMPTY) continue;
            const i128 g = gain128((u64)qv[t], m2, (u64)tv[t], (u64)ki);
            best_take(b, g, cm[t]);
        }
        b = group_best(b);
        if (thr != 0ull) {
            const i128 x128 = (i128)((unsigned __int128)thr * (unsigned __int128)m2);
            const i128 top = b.have ? b.gain : (i128)0;
            if (top < x128) {                       // group-uniform: the band was too narrow for this node -- every entry, exactly
#pragma unroll
                for (int t = 0; t < UN; t++) {
                    const bool nb = i >= 0 && j[t] != i && qv[t] < thr;
                    cm[t] = nb ? com[j[t]] : EMPTY;
                    tv[t] = nb ? kdeg[j[t]] : 0;
                }
#pragma unroll
                for (int t = 0; t < UN; t++) {
                    if (cm[t] == EMPTY) continue;
                    const i128 g = gain128((u64)qv[t], m2, (u64)tv[t], (u64)ki);
                    best_take(b, g, cm[t]);
                }
                b = group_best(b);
            }
        }
        if (i >= 0 && sl == 0) {
            int32_t best = b.have ? b.com : a;
            // two singletons: only towards the lower id (a column whose own row is empty is no graph node: size 0, as in k_lv_decide_g)
            if (best > a && csize[a] == 1 && csize[best] == 1) best = a;
            nxt[i] = best;
        }
        pos = posn;
        i = recn.node;
        deg = recn.deg;
        a = an;
        ki = kin;
#pragma unroll
        for (int t = 0; t < UN; t++) { j[t] = j2[t]; qv[t] = q2[t]; }
    }
}

// Sharded level 0 on a ROW BLOCK (this rank's rows hold entries, the others are empty here): the weighted degrees are
// gathered over the ranks with a "row is not empty" flag riding in bit 62 (degrees stay below 2^61), then totals, community
// sizes, the initial labels and 2m are formed from the gathered vector by every rank.
// offsets of `nb` ascending bounds in a sorted key array (first key >= bound): the segments of a run list by owner
__global__ void k_key_bounds(int64_t nu, const int64_t *__restrict__ keys, int nb, const int64_t *__restrict__ bounds, int64_t *__restrict__ off)
{
    const int t = threadIdx.x;
    if (t >= nb) return;
    const int64_t b = bounds[t];
    int64_t lo = 0, hi = nu;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < b) lo = mid + 1; else hi = mid;
    }
    off[t] = lo;
}

__global__ void k_lv_pack_flag(int64_t r0, int64_t r1, const int64_t *__restrict__ indptr, int64_t *__restrict__ kdeg)
{
    const int64_t i = r0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < r1 && indptr[i + 1] > indptr[i]) kdeg[i] |= (int64_t)1 << 62;
}

__global__ void __launch_bounds__(BLK) k_lv_unpack_init(int64_t n, const int32_t *__restrict__ ids, int64_t *__restrict__ kdeg,
                                                         int64_t *__restrict__ tot, int32_t *__restrict__ csize,
                                                         int32_t *__restrict__ labels, u64 *__restrict__ m2)
{
    const int64_t stride = (int64_t)gridDim.x * BLK;
    long long sum = 0;
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += stride) {
        const int64_t kd = kdeg[i];
        const int flag = (int)((kd >> 62) & 1);
        const int64_t d = kd & ~((int64_t)1 << 62);
        const int32_t id = ids ? ids[i] : (int32_t)i;
        kdeg[i] = d;
        tot[id] = d;
        csize[id] = flag;
        if (labels) labels[i] = flag ? (int32_t)i : -1;       // (level 0 only: later levels relabel through cn)
        sum += d;
    }
    sum = wave_sum_ll(sum);
    __shared__ long long red[WPB];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long t = 0;
        for (int w = 0; w < WPB; w++) t += red[w];
        if (t) atomicAdd(m2, (u64)t);
    }
}

// Exact pruning of the neighbour scan.  With own_i = weight of node i towards the other members of its community a (kept
// current by k_lv_internal / k_lv_delta_internal), every foreign community c has dnc(c) <= F_i = wnl_i - own_i, hence
// gain(c) = dnc(c)*2m - tot(c)*k_i <= F_i*2m.  If that bound is below the (positive) gain of staying, or F_i == 0, no
// neighbour community can win whatever the labels of the neighbours are: the node stays and its row is never read.
// The others are appended to `out` (any order: decisions do not depend on one another inside a sub-round).
// Row-sharded runs (stamp != nullptr): a rank keeps own_i exact with respect to the movers whose rows it holds; a mover of
// ANOTHER rank changes own_i only if it leaves or joins i's community, and every rank knows which communities the foreign
// movers of a sub-round touched (tstamp[c] = that sub-round, k_lv_apply) without seeing their rows.  own_i is trusted iff
// it was formed (stamp[i]: by the neighbour scan, the full recount, or as a mover) after the last such touch of its
// community; the others go through the scan, which refreshes them.  No exchange beyond the label gather is needed.
__global__ void __launch_bounds__(BLK) k_lv_prune(int64_t cnt, const NodeRec *__restrict__ list, u64 m2,
                                                   const int64_t *__restrict__ kdeg, const int64_t *__restrict__ wnl,
                                                   const int64_t *__restrict__ own, const int64_t *__restrict__ tot,
                                                   const int32_t *__restrict__ com, int32_t *__restrict__ nxt,
                                                   NodeRec *__restrict__ out, unsigned long long *__restrict__ out_cnt,
                                                   const int32_t *__restrict__ stamp, const int32_t *__restrict__ tstamp)
{
    // A workgroup settles PRUNE_PER * BLK consecutive list positions and reserves the places of its survivors with ONE atomic: the
    // per-wave reservation of rounds 2-4 queued 3,900 atomics of a 250k-node list on one address at ~11 ns each -- 43 of the
    // kernel's 51 us (round 5, from the kernel trace).  Inside a workgroup the survivors keep their list order.
    constexpr int PER = 4;
    __shared__ unsigned int wcnt[PER][WPB];
    __shared__ unsigned long long wbase;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int64_t chunk = blockIdx.x; chunk * (PER * BLK) < cnt; chunk += gridDim.x) {
        NodeRec rec[PER];
        bool keep[PER];
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int64_t t = chunk * (PER * BLK) + u * BLK + threadIdx.x;
            keep[u] = false;
            rec[u].node = 0; rec[u].deg = 0; rec[u].start = 0;
            if (t < cnt) rec[u] = list[t];
        }
        int32_t a[PER];
        long long ki[PER], o[PER], f[PER];
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int64_t t = chunk * (PER * BLK) + u * BLK + threadIdx.x;
            const int32_t i = rec[u].node;
            a[u] = t < cnt ? com[i] : 0;
            ki[u] = t < cnt ? kdeg[i] : 0;
            o[u] = t < cnt ? own[i] : 0;
            f[u] = t < cnt ? wnl[i] - o[u] : 0;
        }
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int64_t t = chunk * (PER * BLK) + u * BLK + threadIdx.x;
            if (t < cnt) {
                const int32_t i = rec[u].node;
                const i128 stay = gain128((u64)o[u], m2, (u64)(tot[a[u]] - ki[u]), (u64)ki[u]);
                bool stays = f[u] <= 0 || (stay > 0 && (i128)f[u] * (i128)m2 < stay);
                if (stamp && !(tstamp[a[u]] < stamp[i])) stays = false;      // own_i may miss a foreign mover: rescan
                if (stays) nxt[i] = a[u];
                keep[u] = !stays;
            }
        }
        unsigned long long m[PER];
#pragma unroll
        for (int u = 0; u < PER; u++) {
            m[u] = __ballot(keep[u]);
            if (lane == 0) wcnt[u][wv] = (unsigned int)__popcll(m[u]);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned int total = 0;
#pragma unroll
            for (int u = 0; u < PER; u++)
#pragma unroll
                for (int w = 0; w < WPB; w++) total += wcnt[u][w];
            wbase = total ? atomicAdd(out_cnt, (unsigned long long)total) : 0ull;
        }
        __syncthreads();
        unsigned long long at = wbase;
#pragma unroll
        for (int u = 0; u < PER; u++) {
#pragma unroll
            for (int w = 0; w < WPB; w++) {
                if (w == wv && keep[u]) out[at + __popcll(m[u] & ((1ull << lane) - 1ull))] = rec[u];
                at += wcnt[u][w];
            }
        }
        __syncthreads();            // wcnt / wbase are reused by the next chunk
    }
}

__global__ void __launch_bounds__(BLK) k_lv_apply(int64_t n, int n_sub, int round, const int32_t *__restrict__ ids,
                                                   const int64_t *__restrict__ kdeg,
                                                   int64_t *__restrict__ tot, int32_t *__restrict__ com,
                                                   const int32_t *__restrict__ nxt, int32_t *__restrict__ csize,
                                                   u64 *__restrict__ moved, int64_t r0, int64_t r1, int32_t *__restrict__ tstamp,
                                                   int32_t epoch)
{
    const int64_t stride = (int64_t)gridDim.x * BLK;
    int cnt = 0;
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += stride) {
        if (bucket_of(ids ? ids[i] : (int32_t)i, n_sub) != (uint32_t)round) continue;
        const int32_t a = com[i], b = nxt[i];
        if (a == b) continue;
        // a mover whose row another rank holds: its old and new community are "touched" in this sub-round (k_lv_prune)
        if (tstamp && (i < r0 || i >= r1)) { tstamp[a] = epoch; tstamp[b] = epoch; }
        const long long ki = kdeg[i];
        atomicAdd((u64 *)&tot[a], (u64)(-ki));
        atomicAdd((u64 *)&tot[b], (u64)ki);
        atomicSub(&csize[a], 1);
        atomicAdd(&csize[b], 1);
        com[i] = b;
        cnt++;
    }
    __shared__ int bcnt;
    if (threadIdx.x == 0) bcnt = 0;
    __syncthreads();
    const long long c = wave_sum_ll(cnt);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&bcnt, (int)c);
    __syncthreads();
    if (threadIdx.x == 0 && bcnt) atomicAdd(moved, (u64)bcnt);
}

// tot[key] += val, csize[key] += dsz for the lanes with `active`, where MANY lanes may name the same community (level 1 and up: the
// community of an abundant genome absorbs tens of thousands of nodes in one sub-round, and their atomics queue up on ONE address
// at ~11 ns each -- 333 us for an apply launch of the surveyed law's level 1).  Up to two rounds: the lanes that share the first
// active lane's community are summed over the wave and added by one atomic; whoever is left adds for itself.  All 64 lanes call it.
__device__ __forceinline__ void wave_agg_add(bool active, int32_t key, long long val, int dsz, int64_t *__restrict__ tot, int32_t *__restrict__ csize)
{
    const int lane = threadIdx.x & 63;
#pragma unroll 1
    for (int r = 0; r < 2; r++) {
        const u64 act = __ballot(active);
        if (__popcll(act) < 8) break;                      // wave-uniform
        const int l = __ffsll((unsigned long long)act) - 1;
        const int32_t k0 = __builtin_amdgcn_readlane(key, l);
        const bool same = active && key == k0;
        const u64 sm = __ballot(same);
        if (__popcll(sm) < 4) break;
        const long long sum = wave_sum_ll_all(same ? val : 0);
        if (lane == l) {
            atomicAdd((u64 *)&tot[k0], (u64)sum);
            atomicAdd(&csize[k0], dsz * (int)__popcll(sm));
        }
        active = active && !same;
    }
    if (active) {
        atomicAdd((u64 *)&tot[key], (u64)val);
        atomicAdd(&csize[key], dsz);
    }
}

// the same over a prepared node list (single-GPU: only the nodes of the sub-round are visited)
__global__ void __launch_bounds__(BLK) k_lv_apply_list(int64_t cnt, const NodeRec *__restrict__ list,
                                                        const int64_t *__restrict__ kdeg, int64_t *__restrict__ tot,
                                                        int32_t *__restrict__ com, const int32_t *__restrict__ nxt,
                                                        int32_t *__restrict__ csize, u64 *__restrict__ moved)
{
    const int64_t stride = (int64_t)gridDim.x * BLK;
    int c = 0;
    for (int64_t base = (int64_t)blockIdx.x * BLK; base < cnt; base += stride) {        // (whole waves stay in the loop: see wave_agg_add)
        const int64_t t = base + threadIdx.x;
        int32_t i = 0, a = 0, b = 0;
        if (t < cnt) {
            i = list[t].node;
            a = com[i];
            b = nxt[i];
        }
        const bool mv = t < cnt && a != b;
        if (!__ballot(mv)) continue;
        const long long ki = mv ? kdeg[i] : 0;
        wave_agg_add(mv, b, ki, 1, tot, csize);
        wave_agg_add(mv, a, -ki, -1, tot, csize);
        if (mv) { com[i] = b; c++; }
    }
    __shared__ int bcnt;
    if (threadIdx.x == 0) bcnt = 0;
    __syncthreads();
    const long long w = wave_sum_ll(c);
    if ((threadIdx.x & 63) == 0 && w) atomicAdd(&bcnt, (int)w);
    __syncthreads();
    if (threadIdx.x == 0 && bcnt) atomicAdd(moved, (u64)bcnt);
}

// I2 = sum_i ( sum_{j != i, com j == com i} q_ij + 2 q_ii )
__global__ void __launch_bounds__(BLK) k_lv_internal(int64_t r0, int64_t n, const int64_t *__restrict__ indptr,
                                                      const int32_t *__restrict__ indices,
                                                      const int64_t *__restrict__ q, const int32_t *__restrict__ com,
                                                      u64 *__restrict__ i2, int64_t *__restrict__ own,
                                                      int32_t *__restrict__ stamp, int32_t epoch, int64_t long_deg,
                                                      const u64 *__restrict__ moved_dev, int64_t n_level)
{
    // rows of more than long_deg entries are left to k_lv_internal_long (workgroups per row): this pass only clears their own[]
    // HALF mode (round 5).  The graph is symmetric (q_ij == q_ji, both stored), so I2 is twice the sum over the entries above the
    // diagonal plus the loops: the entries below it are skipped without gathering their labels.  own[] -- the per-node by-product the
    // exact pruning starts from -- needs whole rows, but is only read if the NEXT sweep tracks the internal weight through the movers'
    // rows, i.e. if the sweep just finished moved under 35 % of the nodes (lv_track_next): the host applies that rule when the count comes back
    // (b3c_louvain: `incr`), this kernel reads the same count from device memory.  Both rules are one expression: keep them one.
    const bool half = moved_dev != nullptr && !lv_track_next(*moved_dev, (u64)n_level);
    if (half) own = nullptr;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6);
    const int64_t nwave = (int64_t)gridDim.x * WPB;
    long long acc = 0;
    // two rows per wave and pass, so that the loads of both are in flight together (bounds -> entries -> label gathers are
    // dependent round trips; walked row by row the pass ran at 2.8 TB/s)
    for (int64_t row0 = r0 + wave; row0 < n; row0 += 2 * nwave) {
        int64_t rw[2], rs_[2], re_[2];
        int32_t a[2];
        rw[0] = row0;
        rw[1] = row0 + nwave < n ? row0 + nwave : -1;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            rs_[h] = rw[h] >= 0 ? indptr[rw[h]] : 0;
            re_[h] = rw[h] >= 0 ? indptr[rw[h] + 1] : 0;
            a[h] = rw[h] >= 0 ? com[rw[h]] : -1;
            if (re_[h] - rs_[h] > long_deg) re_[h] = rs_[h];
        }
        long long mine[2] = {0, 0};                           // weight towards the other members of the own community
        const int64_t span = (re_[0] - rs_[0]) > (re_[1] - rs_[1]) ? (re_[0] - rs_[0]) : (re_[1] - rs_[1]);
        // 256 entries per row in flight, four CONSECUTIVE ones per lane as 16-byte vectors (the texture addresser charges per
        // instruction, not per byte: one load of ids and two of weights instead of four each; a lane past the row's end re-reads its
        // beginning and drops it): index and weight loads first, then the label gathers (integer sums: any order)
        for (int64_t off = 0; off < span; off += 256) {
            int32_t j[2][4];
            long long w[2][4];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int64_t k0 = rs_[h] + off + 4 * lane;
                const int64_t kb = k0 < re_[h] ? k0 : rs_[h];
                const i32x4_u jv = *(const i32x4_u *)(indices + kb);
                const i64x2_u w0 = *(const i64x2_u *)(q + kb), w1 = *(const i64x2_u *)(q + kb + 2);
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const bool ok = k0 + t < re_[h];
                    j[h][t] = ok ? jv[t] : -1;
                    w[h][t] = ok ? (t < 2 ? w0[t] : w1[t - 2]) : 0;
                }
            }
            int32_t c[2][4];
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    if (half && j[h][t] < (int32_t)rw[h]) j[h][t] = -1;        // below the diagonal: its mirror entry counts for both
                    c[h][t] = j[h][t] >= 0 ? com[j[h][t]] : -2;
                }
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    if (j[h][t] < 0) continue;
                    if (j[h][t] == (int32_t)rw[h]) acc += 2 * w[h][t];
                    else if (c[h][t] == a[h]) { acc += half ? 2 * w[h][t] : w[h][t]; mine[h] += w[h][t]; }
                }
        }
        if (own) {                                            // by-product for k_lv_prune (wave-uniform branch)
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const long long m = wave_sum_ll(mine[h]);
                if (lane == 0 && rw[h] >= 0) {
                    own[rw[h]] = m;
                    if (stamp) stamp[rw[h]] = epoch;
                }
            }
        }
    }
    __shared__ long long red[WPB];
    block_atomic_add_u64(acc, i2, red);
}

// The rows k_lv_internal leaves out (the MEGA rows of the level's node lists: one wave took milliseconds for a row of 10^5
// entries -- 1.35 ms per recount at level 1 of the surveyed genome-size law): workgroup (m, c) takes entries [c * MEGA_CHUNK, ...)
// of node list[m]; integer sums, own[] was cleared by k_lv_internal.
__global__ void __launch_bounds__(1024) k_lv_internal_long(const NodeRec *__restrict__ list, int chunks, const int32_t *__restrict__ indices,
                                                            const int64_t *__restrict__ q, const int32_t *__restrict__ com,
                                                            u64 *__restrict__ i2, int64_t *__restrict__ own,
                                                            const u64 *__restrict__ moved_dev, int64_t n_level)
{
    const bool half = moved_dev != nullptr && !lv_track_next(*moved_dev, (u64)n_level);        // see k_lv_internal
    if (half) own = nullptr;
    const NodeRec rec = list[blockIdx.x / chunks];
    const int64_t c0 = (int64_t)(blockIdx.x % chunks) * 4096;
    if (c0 >= rec.deg) return;
    const int64_t s = rec.start + c0, e = rec.start + (c0 + 4096 < rec.deg ? c0 + 4096 : rec.deg);
    const int32_t row = rec.node, a = com[row];
    long long acc = 0, mine = 0;
    int32_t j[4];
    long long w[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int64_t k = s + threadIdx.x + (int64_t)t * 1024;
        const bool ok = k < e;
        j[t] = ok ? indices[k] : -1;
        w[t] = ok ? q[k] : 0;
    }
    int32_t c[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        if (half && j[t] < row) j[t] = -1;
        c[t] = j[t] >= 0 ? com[j[t]] : -2;
    }
#pragma unroll
    for (int t = 0; t < 4; t++) {
        if (j[t] < 0) continue;
        if (j[t] == row) acc += 2 * w[t];
        else if (c[t] == a) { acc += half ? 2 * w[t] : w[t]; mine += w[t]; }
    }
    acc = wave_sum_ll(acc);
    mine = wave_sum_ll(mine);
    if ((threadIdx.x & 63) == 0) {
        if (acc) atomicAdd(i2, (u64)acc);
        if (own && mine) atomicAdd((u64 *)&own[row], (u64)mine);
    }
}

// Change of I2 caused by the moves a sub-round is about to apply (com = labels before, nxt = labels after; nxt == com
// for every node that does not move).  An off-diagonal entry (v, u) changes its indicator only if v or u moves; scanning
// the rows of the MOVERS covers the entries (v, *) directly and, by symmetry of the graph, the mirror entries (u, v) of
// the neighbours that stay -- hence the factor two for those.  Exact integer arithmetic, any order.
constexpr int LV_DELTA_NPW = 8;          // list positions a wave of k_lv_delta_internal looks at per pass
__global__ void __launch_bounds__(BLK) k_lv_delta_internal(int64_t cnt, const NodeRec *__restrict__ list,
                                                            const int32_t *__restrict__ indices, const int64_t *__restrict__ q,
                                                            const int32_t *__restrict__ com, const int32_t *__restrict__ nxt,
                                                            u64 *__restrict__ i2, int64_t *__restrict__ own, int64_t r0, int64_t r1,
                                                            int32_t *__restrict__ stamp, int32_t epoch)
{
    // Round 5: the list holds every node of the sub-round, the movers are a few per cent of it.  A wave looks at LV_DELTA_NPW nodes per
    // pass (one per lane: their records and labels in flight together), then walks the rows of the movers among them one after the
    // other.  One wave per NODE -- a dependent record -> labels round trip for each of 250,000 nodes to find 10,000 movers -- was
    // 60-115 us per sub-round on the contract workload (from the kernel trace); 64 nodes per pass left too few waves for the rows
    // (193 us: a wave then walks several movers' rows one after the other).
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6);
    const int64_t nwave = (int64_t)gridDim.x * WPB;
    long long acc = 0;
    for (int64_t base = wave * LV_DELTA_NPW; base < cnt; base += nwave * LV_DELTA_NPW) {
        const int64_t t = lane < LV_DELTA_NPW ? base + lane : cnt;
        NodeRec rec;
        rec.node = 0; rec.deg = 0; rec.start = 0;
        int32_t la = 0, lb = 0;
        if (t < cnt) {
            rec = list[t];
            la = com[rec.node];
            lb = nxt[rec.node];
        }
        u64 mk = __ballot(t < cnt && la != lb);
        while (mk) {
            const int l = __ffsll((unsigned long long)mk) - 1;
            mk &= mk - 1;
            const int32_t node = __builtin_amdgcn_readlane(rec.node, l), deg = __builtin_amdgcn_readlane(rec.deg, l);
            const int64_t start = (int64_t)(((u64)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((u64)rec.start >> 32), l) << 32) |
                                            (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u64)rec.start, l));
            const int32_t a = __builtin_amdgcn_readlane(la, l), b = __builtin_amdgcn_readlane(lb, l);
            long long mine = 0;                                   // the mover's weight towards its NEW community, after all moves
            for (int64_t k = start + lane; k < start + deg; k += 64) {
                const int32_t j = indices[k];
                if (j == node) continue;
                const int32_t cj = com[j], nj = nxt[j];
                const int d = (int)(nj == b) - (int)(cj == a);
                if (d) acc += (long long)d * q[k] * (cj == nj ? 2 : 1);
                if (own) {
                    if (nj == b) mine += q[k];
                    // a neighbour that stays: this mover leaves or joins its community (movers recompute theirs from scratch)
                    // (row-sharded: own[] of another rank's nodes is that rank's business -- see k_lv_prune)
                    if (cj == nj && (cj == a || cj == b) && j >= r0 && j < r1) atomicAdd((u64 *)&own[j], (u64)(cj == a ? -q[k] : q[k]));
                }
            }
            if (own) {
                mine = wave_sum_ll(mine);
                if (lane == 0) {
                    own[node] = mine;
                    if (stamp) stamp[node] = epoch;                     // formed from the labels AFTER this sub-round's moves
                }
            }
        }
    }
    acc = wave_sum_ll(acc);
    __shared__ long long red[WPB];
    if (lane == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long tsum = 0;
        for (int w = 0; w < WPB; w++) tsum += red[w];
        if (tsum) atomicAdd(i2, (u64)tsum);
    }
}

// T2 = sum_c tot_c^2 as four 32-bit limbs, each accumulated in a u64
__global__ void __launch_bounds__(BLK) k_lv_tot2(int64_t n, const int64_t *__restrict__ tot, u64 *__restrict__ limbs)
{
    const int64_t stride = (int64_t)gridDim.x * BLK;
    u64 l0 = 0, l1 = 0, l2 = 0, l3 = 0;
    for (int64_t c = (int64_t)blockIdx.x * BLK + threadIdx.x; c < n; c += stride) {
        const unsigned __int128 t = (unsigned __int128)(u64)tot[c] * (u64)tot[c];
        l0 += (u64)(t & 0xFFFFFFFFu);
        l1 += (u64)((t >> 32) & 0xFFFFFFFFu);
        l2 += (u64)((t >> 64) & 0xFFFFFFFFu);
        l3 += (u64)(t >> 96);
    }
    l0 = (u64)wave_sum_ll((long long)l0);
    l1 = (u64)wave_sum_ll((long long)l1);
    l2 = (u64)wave_sum_ll((long long)l2);
    l3 = (u64)wave_sum_ll((long long)l3);
    // per-block fold first: thousands of waves hammering four addresses serialise in the atomic unit
    __shared__ u64 red[WPB][4];
    const int wib = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wib][0] = l0; red[wib][1] = l1; red[wib][2] = l2; red[wib][3] = l3; }
    __syncthreads();
    if (threadIdx.x < 4) {
        u64 t = 0;
        for (int w = 0; w < WPB; w++) t += red[w][threadIdx.x];
        if (t) atomicAdd(&limbs[threadIdx.x], t);
    }
}

// ---------------------------------------------------------------------------------------------
// aggregation
// ---------------------------------------------------------------------------------------------
__global__ void k_flag_used(int64_t n, const int32_t *__restrict__ csize, int64_t *__restrict__ used)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) used[i] = csize[i] > 0;
    if (i == n) used[i] = 0;
}

__global__ void k_node_newcom(int64_t n, const int32_t *__restrict__ com, const int64_t *__restrict__ newid,
                              const int32_t *__restrict__ csize, int32_t *__restrict__ cn)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int32_t c = com[i];
        cn[i] = csize[c] > 0 ? (int32_t)newid[c] : -1;   // -1 only for empty-row nodes of level 0
    }
}

__global__ void k_update_labels(int64_t n0, int32_t *__restrict__ labels, const int32_t *__restrict__ cn)
{
    int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v < n0) {
        const int32_t l = labels[v];
        labels[v] = l >= 0 ? cn[l] : -1;
    }
}

// Pre-reduced pairs: per node ONE pair for everything that stays inside its new community (loops doubled) and one
// pair per entry that leaves it.  After convergence ~90 % of the entries are internal, so the sort below handles
// n + (foreign entries) keys instead of nnz.
__global__ void __launch_bounds__(BLK) k_pairs_count(int64_t n, const int64_t *__restrict__ indptr,
                                                      const int32_t *__restrict__ indices, const int32_t *__restrict__ cn,
                                                      int64_t *__restrict__ cnt)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6);
    const int64_t nwave = (int64_t)gridDim.x * WPB;
    for (int64_t row = wave; row < n; row += nwave) {
        const int64_t s = indptr[row], e = indptr[row + 1];
        const int32_t c1 = cn[row];
        int f = 0;
        for (int64_t base = s; base < e; base += 256) {
            int32_t j[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int64_t k = base + t * 64 + lane;
                j[t] = k < e ? indices[k] : (int32_t)row;
            }
#pragma unroll
            for (int t = 0; t < 4; t++) f += (j[t] != (int32_t)row && cn[j[t]] != c1) ? 1 : 0;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) f += __shfl_down(f, o, 64);
        if (lane == 0) cnt[row] = e > s ? f + 1 : 0;      // +1: the node's own-community pair
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) cnt[n] = 0;
}

__global__ void __launch_bounds__(BLK) k_pairs_fill(int64_t n, int64_t kc, const int64_t *__restrict__ indptr,
                                                     const int32_t *__restrict__ indices, const int64_t *__restrict__ q,
                                                     const int32_t *__restrict__ cn, const int64_t *__restrict__ pos,
                                                     int64_t *__restrict__ keys, int64_t *__restrict__ vals)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6);
    const int64_t nwave = (int64_t)gridDim.x * WPB;
    for (int64_t row = wave; row < n; row += nwave) {
        const int64_t s = indptr[row], e = indptr[row + 1];
        if (e == s) continue;
        const int64_t c1 = cn[row];
        int64_t o = pos[row] + 1;                           // slot pos[row] holds the own-community pair
        long long own = 0;
        for (int64_t base = s; base < e; base += 64) {
            const int64_t k = base + lane;
            int32_t j = (int32_t)row;
            long long qv = 0;
            if (k < e) { j = indices[k]; qv = q[k]; }
            const int32_t c2 = k < e ? cn[j] : (int32_t)c1;
            const bool foreign = k < e && j != (int32_t)row && c2 != (int32_t)c1;
            if (k < e && !foreign) own += j == (int32_t)row ? 2 * qv : qv;   // loops doubled so the internal total halves exactly
            const unsigned long long b = __ballot(foreign);
            if (foreign) {
                const int at = __popcll(b & ((1ull << lane) - 1ull));
                keys[o + at] = c1 * kc + c2;
                vals[o + at] = qv;
            }
            o += __popcll(b);
        }
        own = wave_sum_ll(own);
        if (lane == 0) {
            keys[pos[row]] = c1 * kc + c1;
            vals[pos[row]] = own;
        }
    }
}

// Aggregation by DIRECT BINNING, for levels that end with few communities (kc <= AGG_BINS_KC, i.e. always at level 0 of
// a metagenome): bins[c1*kc + c2] += q with 64-bit integer atomics (exact: any order gives the same sums).  Per node the
// entries that stay inside its new community are summed in registers first (loops doubled, as in k_pairs_fill), so
// ~n + 10 % of nnz atomics reach L2.  The induced graph holds the diagonal of every community and the off-diagonal pairs
// of non-zero weight (a pair whose quantised weights sum to zero, which the sort path would keep as an entry of weight 0,
// can neither attract a node nor change a degree, a total or the modularity: labels, sweeps and Q are unaffected).
constexpr int64_t AGG_BINS_KC = 4096;

constexpr int AGG_SLOTS = 4;

// A block takes `rpb` CONSECUTIVE rows: in a locality-ordered map they belong to one or two communities, whose bin rows
// are kept in LDS (the first `nslot` communities the block meets claim a slot; rows of any further community go straight
// to the global bins) and flushed once at the end -- a tenth of the global atomics on the bench maps.
__global__ void __launch_bounds__(BLK) k_agg_bins(int64_t n, int64_t kc, int64_t rpb, int nslot, int64_t long_deg, const int64_t *__restrict__ indptr,
                                                   const int32_t *__restrict__ indices, const int64_t *__restrict__ q,
                                                   const int32_t *__restrict__ cn, u64 *__restrict__ bins, const int32_t *__restrict__ order)
{
    // order != nullptr: the rows are taken in the order of their NEW community (levels beyond the first, whose nodes are numbered
    // by the communities of the level before: the members of a new community are scattered, a workgroup's rows would belong to
    // as many communities as it has rows and nearly every entry would be an atomic on the global bins)
    extern __shared__ u64 agg_lds[];                    // nslot * kc
    __shared__ int32_t slot_com[AGG_SLOTS];
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int64_t ra = (int64_t)blockIdx.x * rpb, rb = ra + rpb < n ? ra + rpb : n;
    for (int64_t t = threadIdx.x; t < (int64_t)nslot * kc; t += BLK) agg_lds[t] = 0;
    if (threadIdx.x < AGG_SLOTS) slot_com[threadIdx.x] = -1;
    __syncthreads();
    // two rows per wave and pass: the loads of both are in flight together (a row is one or two dependent round trips --
    // bounds, entries, community gathers -- and a wave that walks its rows one by one waits for each of them in turn)
    for (int64_t row0 = ra + wib; row0 < rb; row0 += 2 * WPB) {
        int64_t rw[2], rs_[2], re_[2], c1[2];
        rw[0] = order ? order[row0] : row0;
        rw[1] = row0 + WPB < rb ? (order ? order[row0 + WPB] : row0 + WPB) : -1;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            rs_[h] = rw[h] >= 0 ? indptr[rw[h]] : 0;
            re_[h] = rw[h] >= 0 ? indptr[rw[h] + 1] : 0;
            c1[h] = rw[h] >= 0 ? cn[rw[h]] : 0;
            if (re_[h] - rs_[h] > long_deg) re_[h] = rs_[h];          // rows of k_agg_bins_long
        }
        u64 *lrow[2], *grow[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            int sl = -1;
            if (lane == 0 && re_[h] > rs_[h]) {
                for (int k = 0; k < nslot; k++) {
                    const int32_t prev = atomicCAS(&slot_com[k], -1, (int32_t)c1[h]);
                    if (prev == -1 || prev == (int32_t)c1[h]) { sl = k; break; }
                }
            }
            sl = __shfl(sl, 0, 64);
            lrow[h] = sl >= 0 ? agg_lds + (int64_t)sl * kc : nullptr;
            grow[h] = bins + c1[h] * kc;
        }
        long long own[2] = {0, 0};
        const int64_t span = (re_[0] - rs_[0]) > (re_[1] - rs_[1]) ? (re_[0] - rs_[0]) : (re_[1] - rs_[1]);
        for (int64_t off = 0; off < span; off += 256) {
            int32_t j[2][4];
            long long qv[2][4];
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const int64_t k = rs_[h] + off + t * 64 + lane;
                    const bool ok = k < re_[h];
                    j[h][t] = ok ? indices[k] : -1;
                    qv[h][t] = ok ? q[k] : 0;
                }
            int32_t c2[2][4];
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int t = 0; t < 4; t++) c2[h][t] = j[h][t] >= 0 ? cn[j[h][t]] : (int32_t)c1[h];
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    if (j[h][t] < 0) continue;
                    if (j[h][t] == (int32_t)rw[h]) own[h] += 2 * qv[h][t];
                    else if (c2[h][t] == (int32_t)c1[h]) own[h] += qv[h][t];
                    else if (qv[h][t]) {
                        if (lrow[h]) atomicAdd(&lrow[h][c2[h][t]], (u64)qv[h][t]);
                        else atomicAdd(&grow[h][c2[h][t]], (u64)qv[h][t]);
                    }
                }
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const long long o = wave_sum_ll(own[h]);
            if (lane == 0 && o) {
                if (lrow[h]) atomicAdd(&lrow[h][c1[h]], (u64)o);
                else atomicAdd(&grow[h][c1[h]], (u64)o);
            }
        }
    }
    __syncthreads();
    for (int k = 0; k < nslot; k++) {
        const int32_t c = slot_com[k];
        if (c < 0) continue;
        for (int64_t t = threadIdx.x; t < kc; t += BLK) {
            const u64 v = agg_lds[(int64_t)k * kc + t];
            if (v) atomicAdd(&bins[(int64_t)c * kc + t], v);
        }
    }
}

// Rows of more than AGG_LONG entries -- at level 1 of a heavy-tailed map the community of a genome that the noise links to every
// other community: 10^5 entries and more -- kept ONE wave of k_agg_bins busy for milliseconds while the chip idled (the surveyed
// genome-size law at C3: 8 ms for this pass).  They are listed (any order: the sums are integers) and taken by workgroups of
// AGG_LONG entries each; all entries of a workgroup belong to one row, so its foreign weights go into ONE LDS row of kc sums.
constexpr int64_t AGG_LONG = 8192;
__global__ void k_iota32(int64_t n, int32_t *__restrict__ v)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = (int32_t)i;
}

__global__ void k_agg_long_rows(int64_t n, const int64_t *__restrict__ indptr, int32_t *__restrict__ rows, unsigned long long *__restrict__ cnt)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && indptr[i + 1] - indptr[i] > AGG_LONG) rows[atomicAdd(cnt, 1ull)] = (int32_t)i;
}

__global__ void __launch_bounds__(BLK) k_agg_bins_long(int64_t kc, int chunks, const int32_t *__restrict__ rows, const int64_t *__restrict__ indptr,
                                                        const int32_t *__restrict__ indices, const int64_t *__restrict__ q,
                                                        const int32_t *__restrict__ cn, u64 *__restrict__ bins)
{
    extern __shared__ u64 agg_lds[];                    // kc sums
    const int64_t row = rows[blockIdx.x / chunks];
    const int64_t s = indptr[row] + (int64_t)(blockIdx.x % chunks) * AGG_LONG;
    const int64_t e = indptr[row + 1] < s + AGG_LONG ? indptr[row + 1] : s + AGG_LONG;
    if (s >= e) return;                                 // (uniform per workgroup)
    for (int64_t t = threadIdx.x; t < kc; t += BLK) agg_lds[t] = 0;
    __syncthreads();
    const int32_t c1 = cn[row];
    long long own = 0;
    for (int64_t base = s + threadIdx.x; base < e; base += 4 * BLK) {
        int32_t j[4];
        long long qv[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int64_t k = base + (int64_t)t * BLK;
            const bool ok = k < e;
            j[t] = ok ? indices[k] : -1;
            qv[t] = ok ? q[k] : 0;
        }
        int32_t c2[4];
#pragma unroll
        for (int t = 0; t < 4; t++) c2[t] = j[t] >= 0 ? cn[j[t]] : c1;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            if (j[t] < 0) continue;
            if (j[t] == (int32_t)row) own += 2 * qv[t];
            else if (c2[t] == c1) own += qv[t];
            else if (qv[t]) atomicAdd(&agg_lds[c2[t]], (u64)qv[t]);
        }
    }
    own = wave_sum_ll(own);
    if ((threadIdx.x & 63) == 0 && own) atomicAdd(&agg_lds[c1], (u64)own);
    __syncthreads();
    for (int64_t t = threadIdx.x; t < kc; t += BLK) {
        const u64 v = agg_lds[t];
        if (v) atomicAdd(&bins[(int64_t)c1 * kc + t], v);
    }
}

// entries per row of the induced graph: the diagonal and the non-zero bins (one wave per row)
__global__ void __launch_bounds__(BLK) k_bins_count(int64_t kc, const u64 *__restrict__ bins, int64_t *__restrict__ cnt)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6);
    if (row > kc) return;
    if (row == kc) { if (lane == 0) cnt[row] = 0; return; }
    long long c = 0;
    for (int64_t col = lane; col < kc; col += 64) c += (col == row || bins[row * kc + col] != 0) ? 1 : 0;
    c = wave_sum_ll(c);
    if (lane == 0) cnt[row] = c;
}

__global__ void __launch_bounds__(BLK) k_bins_fill(int64_t kc, const u64 *__restrict__ bins, const int64_t *__restrict__ indptr,
                                                    int32_t *__restrict__ indices, int64_t *__restrict__ q)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6);
    if (row >= kc) return;
    int64_t o = indptr[row];
    for (int64_t c0 = 0; c0 < kc; c0 += 64) {
        const int64_t c = c0 + lane;
        u64 sum = 0;
        bool on = false;
        if (c < kc) {
            sum = bins[row * kc + c];
            on = c == row || sum != 0;
        }
        const unsigned long long m = __ballot(on);
        if (on) {
            const int64_t at = o + __popcll(m & ((1ull << lane) - 1ull));
            indices[at] = (int32_t)c;
            q[at] = c == row ? (int64_t)(sum / 2) : (int64_t)sum;
        }
        o += __popcll(m);
    }
}

// ---------------------------------------------------------------------------------------------
// Aggregation by hashing, for levels that end with MORE communities than the bin table holds (round 4).  The pair sort moves
// every foreign entry through five radix passes (the surveyed genome-size law at C3: level 0 ends with 185,559 communities of ~5
// contigs, 2.0e8 pairs, 12.4 ms).  Here the rows of one new community are read once and their entries summed per target
// community in a hash table -- in LDS for the communities whose rows hold at most 6553 entries together (one workgroup each,
// tables of 1024 / 4096 / 8192 slots as in k_lv_decide_wg), in global memory for the others (one workgroup per member row
// inserts, one per community emits).  The distinct (target, sum) pairs of community c land at ubpos[c] of a scratch area laid
// out by the upper bounds ub[c] = entries of c's rows; a scan of the counts and one copy make the CSR of the next level.
// Integer sums: the induced graph is the one the sort path builds, entry for entry -- only the ORDER of the entries inside a
// row differs (slot order instead of ascending columns), which no kernel of the next level depends on.
// ---------------------------------------------------------------------------------------------
// sort key of a node: its new community; the nodes that are in no community (empty rows of level 0: cn = -1) go behind all of them
__global__ void k_aggh_keys(int64_t n, int64_t kc, const int32_t *__restrict__ cn, int32_t *__restrict__ key)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) key[i] = cn[i] >= 0 ? cn[i] : (int32_t)kc;
}

__global__ void k_aggh_starts(int64_t n, int64_t kc, const int32_t *__restrict__ skey, int64_t *__restrict__ cstart)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    if (i == n) { if (n == 0 || skey[n - 1] < kc) cstart[kc] = n; return; }
    if (i == 0 || skey[i] != skey[i - 1]) cstart[skey[i] < kc ? skey[i] : kc] = i;        // (every community 0 .. kc-1 has a member)
}

__global__ void k_aggh_ub(int64_t n, const int64_t *__restrict__ indptr, const int32_t *__restrict__ cn, unsigned long long *__restrict__ ub)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t d = indptr[i + 1] - indptr[i];
    if (d) atomicAdd(&ub[cn[i]], (unsigned long long)d);
}

// class of every community by ub: 0 .. 2 = LDS tables of 1024 / 4096 / 8192 slots, 3 = global table; lists by atomic append (any
// order), table sizes of class 3
__global__ void k_aggh_classify(int64_t kc, const unsigned long long *__restrict__ ub, int32_t *__restrict__ lists, unsigned long long *__restrict__ lcnt,
                                int64_t *__restrict__ tsz, int64_t *__restrict__ need)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int cls = -1;
    if (c == kc) { tsz[kc] = 0; need[kc] = 0; }
    if (c < kc) {
        const long long u = (long long)ub[c];
        need[c] = u + 1;                                  // scratch slots of the community: every entry a target of its own, plus the diagonal
        cls = u <= lv_wg_max_deg(0) ? 0 : u <= lv_wg_max_deg(1) ? 1 : u <= lv_wg_max_deg(2) ? 2 : 3;
        tsz[c] = cls == 3 ? (1ll << table_lg(u)) : 0;
    }
    // one atomic per wave and class (185k communities appending one by one queue up on four addresses: 1.6 ms at C3's surveyed law)
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const unsigned long long m = __ballot(cls == k);
        if (!m) continue;
        unsigned long long at = 0;
        if (lane == 0) at = atomicAdd(&lcnt[k], (unsigned long long)__popcll(m));
        at = (unsigned long long)__shfl((long long)at, 0, 64);
        if (cls == k) lists[(int64_t)k * kc + at + __popcll(m & ((1ull << lane) - 1ull))] = (int32_t)c;
    }
}

// block-wide exclusive prefix of one count per thread (NT threads); returns the thread's offset, *total = the block's sum
template <int NT>
__device__ __forceinline__ int block_excl_scan(int v, int *lds, int *total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) lds[wv] = incl;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < NT / 64; w++) {
        const int t = lds[w];
        if (w < wv) base += t;
        tot += t;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

template <int LGS, int NT>
__global__ void __launch_bounds__(NT) k_aggh_wg(int64_t cnt, const int32_t *__restrict__ list, const int64_t *__restrict__ cstart,
                                                 const int32_t *__restrict__ ord, const int64_t *__restrict__ indptr,
                                                 const int32_t *__restrict__ indices, const int64_t *__restrict__ q,
                                                 const int32_t *__restrict__ cn, const int64_t *__restrict__ ubpos,
                                                 int32_t *__restrict__ tix, int64_t *__restrict__ tq, int64_t *__restrict__ dcnt)
{
    constexpr int SLOTS = 1 << LGS;
    constexpr int NW = NT / 64;
    extern __shared__ __align__(16) unsigned char lv_wg_smem[];
    u64 *vals = (u64 *)lv_wg_smem;
    int32_t *keys = (int32_t *)(vals + SLOTS);
    int *sc = (int *)(keys + SLOTS);                   // NW ints of the scan, then NW 64-bit sums
    long long *red = (long long *)(sc + NW + (NW & 1));
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int t = tid; t < SLOTS; t += NT) { keys[t] = EMPTY; vals[t] = 0; }
    __syncthreads();
    for (int64_t pos = blockIdx.x; pos < cnt; pos += gridDim.x) {
        const int32_t c1 = list[pos];
        const int64_t m0 = cstart[c1], m1 = cstart[c1 + 1];
        long long own = 0;
        // wave w takes the member rows w, w + NW, ...; its lanes stride the row
        for (int64_t m = m0 + wv; m < m1; m += NW) {
            const int32_t i = ord[m];
            const int64_t s = indptr[i], e = indptr[i + 1];
            for (int64_t base = s; base < e; base += 4 * 64) {
                int32_t j[4], c2[4];
                u64 qv[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int64_t k = base + u * 64 + lane;
                    j[u] = k < e ? indices[k] : -1;
                    qv[u] = k < e ? (u64)q[k] : 0;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) c2[u] = j[u] >= 0 ? cn[j[u]] : EMPTY;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (j[u] < 0) continue;
                    if (c2[u] == c1) { own += j[u] == i ? 2 * (long long)qv[u] : (long long)qv[u]; continue; }   // loops doubled: the internal total halves exactly
                    uint32_t slot = hash_slot(c2[u], LGS);
                    for (;;) {
                        const int32_t prev = atomicCAS(&keys[slot], EMPTY, c2[u]);
                        if (prev == EMPTY || prev == c2[u]) { atomicAdd(&vals[slot], qv[u]); break; }
                        slot = (slot + 1) & (SLOTS - 1);
                    }
                }
            }
        }
        own = wave_sum_ll(own);
        if (lane == 0) red[wv] = own;
        __syncthreads();
        // emit: the diagonal first, then every non-empty slot
        int mine = 0;
        for (int t = tid; t < SLOTS; t += NT) mine += keys[t] != EMPTY ? 1 : 0;
        int total = 0;
        int at = block_excl_scan<NT>(mine, sc, &total);
        const int64_t o = ubpos[c1];
        if (tid == 0) {
            long long t = 0;
            for (int w = 0; w < NW; w++) t += red[w];
            tix[o] = c1;
            tq[o] = t / 2;
            dcnt[c1] = total + 1;
        }
        for (int t = tid; t < SLOTS; t += NT) {
            const int32_t k = keys[t];
            if (k == EMPTY) continue;
            tix[o + 1 + at] = k;
            tq[o + 1 + at] = (int64_t)vals[t];
            at++;
            keys[t] = EMPTY;
            vals[t] = 0;
        }
        __syncthreads();
    }
}

// communities beyond the LDS tables: one workgroup per MEMBER ROW inserts into the community's table in global memory.  The lane
// that turns an EMPTY slot into a target community CLAIMS it: it reserves the next output position of the community (dcnt) and
// writes the target there at once -- so the emit pass below walks the community's DISTINCT targets (a few thousand) instead of
// its table (sized by all entries of its rows, internal ones included: 2^19 slots for a genome of 2000 contigs) ...
__global__ void __launch_bounds__(BLK) k_aggh_big_insert(int64_t nrows, const int32_t *__restrict__ rows, const int64_t *__restrict__ indptr,
                                                          const int32_t *__restrict__ indices, const int64_t *__restrict__ q,
                                                          const int32_t *__restrict__ cn, const unsigned long long *__restrict__ ub,
                                                          const int64_t *__restrict__ toff, int32_t *__restrict__ gkeys, u64 *__restrict__ gvals,
                                                          unsigned long long *__restrict__ own_acc, const int64_t *__restrict__ ubpos,
                                                          int32_t *__restrict__ tix, unsigned long long *__restrict__ dcnt)
{
    __shared__ long long red[BLK / 64];
    for (int64_t pos = blockIdx.x; pos < nrows; pos += gridDim.x) {
        const int32_t i = rows[pos];
        const int32_t c1 = cn[i];
        const int lg = table_lg((int64_t)ub[c1]);
        const uint32_t mask = (1u << lg) - 1u;
        int32_t *keys = gkeys + toff[c1];
        u64 *vals = gvals + toff[c1];
        const int64_t s = indptr[i], e = indptr[i + 1], o = ubpos[c1];
        long long own = 0;
        for (int64_t k = s + threadIdx.x; k < e; k += BLK) {
            const int32_t j = indices[k];
            const u64 qv = (u64)q[k];
            const int32_t c2 = cn[j];
            if (c2 == c1) { own += j == i ? 2 * (long long)qv : (long long)qv; continue; }
            uint32_t slot = hash_slot(c2, lg);
            for (;;) {
                const int32_t prev = atomicCAS(&keys[slot], EMPTY, c2);
                if (prev == EMPTY || prev == c2) {
                    atomicAdd(&vals[slot], qv);
                    if (prev == EMPTY) tix[o + 1 + (int64_t)atomicAdd(&dcnt[c1], 1ull)] = c2;      // (slot 0 of the community: its diagonal)
                    break;
                }
                slot = (slot + 1) & mask;
            }
        }
        block_atomic_add_u64(own, own_acc + c1, red);
    }
}

// ... one workgroup per community: the diagonal, and the sum of every claimed target by probing the table once more
__global__ void __launch_bounds__(BLK) k_aggh_big_emit(int64_t cnt, const int32_t *__restrict__ list, const unsigned long long *__restrict__ ub,
                                                        const int64_t *__restrict__ toff, const int32_t *__restrict__ gkeys,
                                                        const u64 *__restrict__ gvals, const unsigned long long *__restrict__ own_acc,
                                                        const int64_t *__restrict__ ubpos, int32_t *__restrict__ tix, int64_t *__restrict__ tq,
                                                        unsigned long long *__restrict__ dcnt)
{
    for (int64_t pos = blockIdx.x; pos < cnt; pos += gridDim.x) {
        const int32_t c1 = list[pos];
        const int lg = table_lg((int64_t)ub[c1]);
        const uint32_t mask = (1u << lg) - 1u;
        const int32_t *keys = gkeys + toff[c1];
        const u64 *vals = gvals + toff[c1];
        const int64_t o = ubpos[c1];
        const int64_t nd = (int64_t)dcnt[c1];
        for (int64_t t = threadIdx.x; t < nd; t += BLK) {
            const int32_t c2 = tix[o + 1 + t];
            uint32_t slot = hash_slot(c2, lg);
            while (keys[slot] != c2) slot = (slot + 1) & mask;
            tq[o + 1 + t] = (int64_t)vals[slot];
        }
        __syncthreads();
        if (threadIdx.x == 0) { tix[o] = c1; tq[o] = (int64_t)(own_acc[c1] / 2); dcnt[c1] = (unsigned long long)nd + 1; }
    }
}

__global__ void k_aggh_big_rows(int64_t n, const int32_t *__restrict__ cn, const int64_t *__restrict__ tsz, const int64_t *__restrict__ indptr,
                                int32_t *__restrict__ rows, unsigned long long *__restrict__ nrows)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool big = i < n && indptr[i + 1] > indptr[i] && tsz[cn[i]] > 0;
    const unsigned long long m = __ballot(big);
    if (!m) return;
    unsigned long long at = 0;
    if ((threadIdx.x & 63) == 0) at = atomicAdd(nrows, (unsigned long long)__popcll(m));
    at = (unsigned long long)__shfl((long long)at, 0, 64);
    if (big) rows[at + __popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull))] = (int32_t)i;
}

// copy the dcnt[c] pairs of community c from the scratch layout to row c of the next level's CSR (one wave per community)
__global__ void __launch_bounds__(BLK) k_aggh_compact(int64_t kc, const int64_t *__restrict__ ubpos, const int64_t *__restrict__ hptr,
                                                       const int32_t *__restrict__ tix, const int64_t *__restrict__ tq,
                                                       int32_t *__restrict__ hix, int64_t *__restrict__ hq)
{
    const int lane = threadIdx.x & 63;
    for (int64_t c = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6); c < kc; c += (int64_t)gridDim.x * WPB) {
        const int64_t src = ubpos[c], dst = hptr[c], len = hptr[c + 1] - dst;
        // four consecutive pairs per lane as 16-byte vectors (the addresser charges per instruction: 3 loads + 3 stores per four pairs
        // instead of 8 + 8), the last len % 4 pairs one by one
        const int64_t len4 = len & ~3ll;
        for (int64_t t = 4 * lane; t < len4; t += 256) {
            const i32x4_u iv = *(const i32x4_u *)(tix + src + t);
            const i64x2_u q0 = *(const i64x2_u *)(tq + src + t), q1 = *(const i64x2_u *)(tq + src + t + 2);
            *(i32x4_u *)(hix + dst + t) = iv;
            *(i64x2_u *)(hq + dst + t) = q0;
            *(i64x2_u *)(hq + dst + t + 2) = q1;
        }
        if (lane < len - len4) { hix[dst + len4 + lane] = tix[src + len4 + lane]; hq[dst + len4 + lane] = tq[src + len4 + lane]; }
    }
}

__global__ void k_rows_from_keys(int64_t kc, int64_t nu, const int64_t *__restrict__ ukeys,
                                 int64_t *__restrict__ indptr)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > kc) return;
    const int64_t target = r * kc;   // first key of row r
    int64_t lo = 0, hi = nu;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (ukeys[mid] < target) lo = mid + 1; else hi = mid;
    }
    indptr[r] = lo;
}

__global__ void k_fill_from_keys(int64_t kc, int64_t nu, const int64_t *__restrict__ ukeys,
                                 const int64_t *__restrict__ usums, int32_t *__restrict__ indices,
                                 int64_t *__restrict__ q)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nu; e += stride) {
        const int64_t key = ukeys[e];
        const int64_t r = key / kc, c = key - r * kc;
        indices[e] = (int32_t)c;
        q[e] = r == c ? usums[e] / 2 : usums[e];
    }
}

__global__ void k_gather_labels(int64_t n, const int32_t *__restrict__ src, const int32_t *__restrict__ idx,
                                int32_t *__restrict__ dst)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

__global__ void k_init_labels(int64_t n, const int64_t *__restrict__ indptr, int32_t *__restrict__ labels)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) labels[i] = indptr[i + 1] > indptr[i] ? (int32_t)i : -1;
}

// table in global memory for the rows whose table does not fit `cap` slots of LDS, and for every row beyond deg_thr entries (the
// mega rows, whatever the threshold: k_lv_mega_* work on global tables only)
__global__ void k_big_table_sizes(int64_t n, int cap, int64_t deg_thr, const int64_t *__restrict__ indptr, int64_t *__restrict__ sz)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int64_t deg = indptr[i + 1] - indptr[i];
        const int lg = table_lg(deg);
        sz[i] = ((1ll << lg) > cap || deg > deg_thr) ? (1ll << lg) : 0;
    }
    if (i == n) sz[i] = 0;
}

__global__ void k_max_degree(int64_t n, const int64_t *__restrict__ indptr, u64 *__restrict__ out)
{
    max_row_len_body(n, indptr, (unsigned long long *)out);
}

struct Events {
    std::vector<hipEvent_t> ev;
    bool on = false;
    hipStream_t s;
    b3c_ctx *ctx = nullptr;
    void rec()
    {
        if (!on) return;
        hipEvent_t e = ctx->ev_get();
        if (e) { (void)hipEventRecord(e, s); ev.push_back(e); }
    }
    double total_ms(size_t first_pairs = 0, double *first_ms = nullptr)
    {
        double ms = 0;
        for (size_t i = 0; i + 1 < ev.size(); i += 2) {
            float f = 0;
            if (hipEventElapsedTime(&f, ev[i], ev[i + 1]) == hipSuccess) {
                ms += f;
                if (first_ms && i / 2 < first_pairs) *first_ms += f;
            }
        }
        for (auto e : ev) ctx->ev_put(e);
        ev.clear();
        return ms;
    }
};

struct DeferredQuant { const double *src = nullptr; double scl = 1.0; };      // set: k_lv_init forms q itself from src * scl

int quantise_graph(b3c_ctx *ctx, const b3c_mat *a, const int32_t *ids, double scl, int64_t *q, int32_t *shift_out, bool blockm = false,
                   DeferredQuant *defer = nullptr)
{
    hipStream_t s = ctx->stream;
    const int64_t n = a->n, nnz = a->nnz;
    *shift_out = 0;
    if (nnz == 0 && !blockm) return B3C_OK;          // (an EMPTY row block still takes part in the two exchanges below)
    // value-symmetric matrices (the normal case: graph form) need no materialised weights: a * scl is formed on the fly
    const bool direct = a->value_symmetric;
    DevBuf<double> w(ctx, direct ? 1 : nnz);
    const int g = grid_for(n, WPB, 16 * ctx->n_cu);
    DevBuf<double> part(ctx, g);
    DevBuf<u64> total(ctx, 1);
    if (!w.p || !part.p || !total.p) return ctx->fail(B3C_E_NOMEM, "quantise_graph");
    const double *wsrc = direct ? (const double *)a->data : w.p;
    const double wscl = direct ? scl : 1.0;
    double wmax = 0.0;
    if (direct && a->vmax_valid && scl > 0.0 && scl < INFINITY) {
        // rounding is monotone: the largest a*scl is (largest a)*scl, and the largest a is on record
        const double t = a->vmax * scl;
        wmax = t > 0.0 ? t : 0.0;
    } else {
        if (direct)
            hipLaunchKernelGGL(k_graph_wmax_sym, dim3(g), dim3(BLK), 0, s, nnz, (const double *)a->data, scl, part.p);
        else
            hipLaunchKernelGGL(k_graph_w, dim3(g), dim3(BLK), 0, s, n, a->indptr, a->indices, (const double *)a->data, ids,
                               a->sorted_cols ? 1 : 0, scl, w.p, part.p);
        std::vector<double> hp(g);
        { const int rb_ = b3c_readback(ctx, hp.data(), part.p, sizeof(double) * g); if (rb_) return rb_; }
        for (int i = 0; i < g; i++) wmax = hp[i] > wmax ? hp[i] : wmax;
    }
    if (blockm) {
        // a row block sees its rows only: the largest weight and (below) the integer total are taken over the ranks
        const b3c_dist &D = ctx->dist;
        HIP_TRY(ctx, hipMemcpyAsync(D.scal + 4 * D.rank, &wmax, sizeof(double), hipMemcpyHostToDevice, s));
        { const int rx_ = b3c_dist_exchange(ctx, B3C_BUF_SCAL); if (rx_) return rx_; }
        std::vector<double> all(4 * (size_t)D.world);
        { const int rb_ = b3c_readback(ctx, all.data(), D.scal, sizeof(double) * all.size()); if (rb_) return rb_; }
        for (int r = 0; r < D.world; r++) wmax = all[4 * (size_t)r] > wmax ? all[4 * (size_t)r] : wmax;
    }
    int shift = 0;
    if (wmax > 0.0) {
        int e;
        frexp(wmax, &e);   // wmax < 2^e
        u64 tot = 0;
        bool known = false;
        if (direct && !blockm && a->vsum_valid && a->vmax_valid && scl > 0.0 && scl < INFINITY && a->vsum2 > 0.0 && a->vsum2 < INFINITY) {
            // Only the BIT LENGTH of the total matters (the shift below), and the total is pinned between two bounds by what the
            // kernel that wrote the values left behind: T = sum ceil(a*scl*2^(20-e)) (diagonal twice) lies in
            // [X(1 - d), X(1 + d) + N] with X = vsum2*scl*2^(20-e), N = nnz + ndiag (one unit of ceil per term) and d = 1e-6, far
            // above the rounding of the products and of the (order-free, atomically accumulated) sum.  If both bounds have the
            // same bit length it is the total's; otherwise (about one map in 10^4) the exact pass runs.
            const long double X = (long double)a->vsum2 * (long double)scl * ldexpl(1.0L, 20 - e);
            const long double lo = X * (1.0L - 1e-6L), hi = X * (1.0L + 1e-6L) + (long double)(a->nnz + a->ndiag);
            if (lo >= 1.0L && hi < 18446744073709551615.0L) {
                const u64 l = (u64)lo, h = (u64)hi + 1;
                int bl = 0, bh = 0;
                while (bl < 64 && (l >> bl)) bl++;
                while (bh < 64 && (h >> bh)) bh++;
                if (bl == bh) { tot = l; known = true; }
            }
        }
        if (!known) {
            HIP_TRY(ctx, hipMemsetAsync(total.p, 0, sizeof(u64), s));
            hipLaunchKernelGGL(k_quant_total, dim3(g), dim3(BLK), 0, s, n, a->indptr, a->indices, wsrc, wscl, 20 - e, total.p);
            { const int rb_ = b3c_readback(ctx, &tot, total.p, sizeof(u64)); if (rb_) return rb_; }
        }
        if (blockm) {
            const b3c_dist &D = ctx->dist;
            HIP_TRY(ctx, hipMemcpyAsync(D.isum + 2 * D.rank, total.p, sizeof(u64), hipMemcpyDeviceToDevice, s));
            { const int rx_ = b3c_dist_exchange(ctx, B3C_BUF_ISUM); if (rx_) return rx_; }
            std::vector<int64_t> parts(2 * (size_t)D.world);
            { const int rb_ = b3c_readback(ctx, parts.data(), D.isum, sizeof(int64_t) * parts.size()); if (rb_) return rb_; }
            tot = 0;
            for (int r = 0; r < D.world; r++) tot += (u64)parts[2 * (size_t)r];
        }
        int bits = 0;
        while (bits < 64 && (tot >> bits)) bits++;
        shift = 61 - (bits - 20 + e);
    }
    if (defer && direct && nnz > 0) {
        defer->src = wsrc;              // (the matrix's own values: they outlive this call)
        defer->scl = wscl;
    } else {
        hipLaunchKernelGGL(k_quantise, dim3(grid_for(nnz, BLK * 4, 8 * ctx->n_cu)), dim3(BLK), 0, s, nnz, wsrc, wscl, shift, q);
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(s));
    *shift_out = shift;
    return B3C_OK;
}

void free_graph(b3c_ctx *ctx, LvGraph &g)
{
    if (g.owned) {
        ctx->release(g.indptr);
        ctx->release(g.indices);
        ctx->release(g.q);
    }
    g = LvGraph();
}

}  // namespace

extern "C" int b3c_graph_quantise(b3c_ctx *ctx, const b3c_mat *a, double scl, int64_t *q, int32_t *shift)
{
    if (!ctx) return B3C_E_ARG;
    if (!a || a->kind != B3C_F64 || !q || !shift) return ctx->fail(B3C_E_ARG, "b3c_graph_quantise: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf<int64_t> dq(ctx, a->nnz);
    if (!dq.p) return ctx->fail(B3C_E_NOMEM, "b3c_graph_quantise");
    int rc = quantise_graph(ctx, a, a->ids, scl, dq.p, shift);
    if (rc) return rc;
    return b3c_copy_out(ctx, q, dq.p, sizeof(int64_t) * (size_t)a->nnz);
}

extern "C" int b3c_louvain(b3c_ctx *ctx, const b3c_mat *a, double scl, int n_sub, int max_sweeps, int32_t *labels_out,
                           b3c_louvain_info *info)
{
    if (!ctx) return B3C_E_ARG;
    if (!a || a->kind != B3C_F64 || !labels_out || !info) return ctx->fail(B3C_E_ARG, "b3c_louvain: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    memset(info, 0, sizeof(*info));
    if (n_sub <= 0) n_sub = 4;
    hipStream_t s = ctx->stream;
    const int64_t n0 = a->n;
    if (n0 == 0) return B3C_OK;
    const double PL_MIN = 0.0000001;   // community_louvain.__MIN

    const bool lv_debug = CFG.on("B3C_LV_DEBUG");
    Events evs;
    evs.on = CFG.on("B3C_PROFILE");
    evs.s = s;
    evs.ctx = ctx;

    // Locality: a matrix that the caller has not ordered (b3c_mat_permute) gets an internal ordered
    // working copy; community ids, sub-round hashing and tie-breaks stay in the caller's numbering
    // through `ids`, so the labels are identical either way (tests/test_gpu_louvain.py).
    struct Guard { b3c_ctx *c; b3c_mat *m = nullptr; ~Guard() { if (m) b3c_mat_free(c, m); } } perm{ctx};
    const bool reorder = !a->ordered && n0 >= 32768 && a->nnz >= (1 << 21) && !CFG.on("B3C_NO_REORDER");
    DevBuf<int32_t> new2old(ctx, reorder ? n0 : 1), old2new(ctx, reorder ? n0 : 1);
    int rc;
    if (reorder) {
        if (!new2old.p || !old2new.p) return ctx->fail(B3C_E_NOMEM, "b3c_louvain");
        if ((rc = b3c_locality_order_dev(ctx, a, new2old.p, old2new.p))) return rc;
        if ((rc = b3c_permute_csr(ctx, a, new2old.p, old2new.p, &perm.m, false))) return rc;
        a = perm.m;
    }
    const int32_t *ids0 = reorder ? new2old.p : a->ids;

    DevBuf<int32_t> labels(ctx, n0);
    DevBuf<int64_t> q0(ctx, a->nnz + 8);           // (+ 8: the scans read the entry streams as 16-byte vectors, up to 7 entries past a row)
    if (!labels.p || !q0.p) return ctx->fail(B3C_E_NOMEM, "b3c_louvain");
    int32_t shift = 0;
    // sharded level 0 on a row block (b3c_mat_permute_rows + compress): this rank holds the entries of its rows only
    const bool blockm = ctx->has_dist && a->row_block;
    if (blockm && reorder) return ctx->fail(B3C_E_ARG, "b3c_louvain: a row block must be ordered by the caller");
    DeferredQuant dq;
    rc = quantise_graph(ctx, a, ids0, scl, q0.p, &shift, blockm, &dq);
    if (rc) return rc;
    info->quant_shift = shift;
    hipLaunchKernelGGL(k_init_labels, dim3((unsigned)((n0 + 255) / 256)), dim3(256), 0, s, n0, a->indptr, labels.p);

    LvGraph g;
    g.n = n0; g.nnz = a->nnz; g.indptr = a->indptr; g.indices = a->indices; g.q = q0.p; g.owned = false;
    const int32_t *ids = ids0;   // level 0 only; induced graphs are numbered by ascending community id
    struct GraphGuard { b3c_ctx *c; LvGraph *g; ~GraphGuard() { free_graph(c, *g); } } graph_guard{ctx, &g};   // every exit path

    double mod = 0.0;
    bool first = true;
    DevBuf<u64> ctr(ctx, 8);   // [0] m2, [1] moved, [2] i2, [3..6] tot2 limbs, [7] max degree
    if (!ctr.p) return ctx->fail(B3C_E_NOMEM, "b3c_louvain");

    for (;;) {
        const int64_t n = g.n;
        DevBuf<int64_t> kdeg(ctx, n), tot(ctx, n), goff(ctx, n + 1), gsz(ctx, n + 1);
        DevBuf<int32_t> com(ctx, n), nxt(ctx, n), csize(ctx, n);
        DevBuf<int64_t> wnl(ctx, n), own(ctx, n);            // weight towards the other nodes / towards the own community (k_lv_prune)
        if (!kdeg.p || !tot.p || !goff.p || !gsz.p || !com.p || !nxt.p || !csize.p || !wnl.p || !own.p) {
            free_graph(ctx, g);
            return ctx->fail(B3C_E_NOMEM, "b3c_louvain: level buffers");
        }
        // multi-GPU: level 0 is row-sharded (decide + internal-weight scan).  An induced graph is REPLICATED (every rank builds the
        // same one from the reduced aggregation), but when it is still large -- the contract workload at C3: level 1 has 185,559
        // nodes and 1.4e8 entries, as many as level 0 -- its sweeps are sharded too (round 6): every rank scans and recounts a row
        // range of its copy, the decisions of a sub-round are gathered like level 0's, moves are applied by every rank.  Small
        // levels stay whole on every rank: four label gathers and one partial sum per sweep would cost more than their scan
        // (B3C_LV_DIST_MIN_NNZ: the tests shard every level)
        const int64_t lv_dist_min_nnz = CFG.num("B3C_LV_DIST_MIN_NNZ", 1 << 24);
        const bool dist = ctx->has_dist && (first || g.block || (g.nnz >= lv_dist_min_nnz && n >= 2 * (int64_t)ctx->dist.world)) &&
                          (int64_t)ctx->dist.world * ctx->dist.chunk >= n;
        if (g.block && !dist) { free_graph(ctx, g); return ctx->fail(B3C_E_ARG, "b3c_louvain: a row-block level without its sharding"); }
        if (dist && !first) info->levels_sharded++;
        const b3c_dist &D = ctx->dist;
        if (first && blockm && !dist) { free_graph(ctx, g); return ctx->fail(B3C_E_ARG, "b3c_louvain: row block larger than the exchange buffers"); }
        // this level's graph is a ROW BLOCK: level 0 on a row block of the map, or an induced graph the owners built (round 6);
        // otherwise every rank holds it whole (sharded or not)
        const bool blk0 = dist && (first ? blockm : g.block);
        // rows of every rank: equal chunks of a whole matrix, or the row blocks the ranks hold (their bounds are exchanged)
        std::vector<int64_t> starts(dist ? (size_t)D.world + 1 : 2, 0);
        DevBuf<int64_t> starts_dev(ctx, dist ? D.world + 1 : 1);
        if (dist) {
            if (!starts_dev.p) { free_graph(ctx, g); return ctx->fail(B3C_E_NOMEM, "b3c_louvain"); }
            if (blk0 && !first) {
                starts = g.starts;                  // the owners' ranges: the same on every rank, nothing to exchange
            } else if (blk0) {
                const int64_t mine[2] = {a->row0, a->row1};
                HIP_TRY(ctx, hipMemcpyAsync(D.isum + 2 * D.rank, mine, sizeof(mine), hipMemcpyHostToDevice, s));
                { const int rx_ = b3c_dist_exchange(ctx, B3C_BUF_ISUM); if (rx_) return rx_; }
                std::vector<int64_t> all(2 * (size_t)D.world);
                { const int rb_ = b3c_readback(ctx, all.data(), D.isum, sizeof(int64_t) * all.size()); if (rb_) return rb_; }
                for (int r = 0; r < D.world; r++) {
                    starts[r] = all[2 * (size_t)r];
                    if (r > 0 && all[2 * (size_t)r] != all[2 * (size_t)r - 1]) { free_graph(ctx, g); return ctx->fail(B3C_E_ARG, "b3c_louvain: the ranks' row blocks do not tile the matrix"); }
                }
                starts[D.world] = all[2 * (size_t)D.world - 1];
                if (starts[0] != 0 || starts[D.world] != n) { free_graph(ctx, g); return ctx->fail(B3C_E_ARG, "b3c_louvain: the ranks' row blocks do not tile the matrix"); }
            } else if (first) {
                for (int r = 0; r < D.world; r++) { int64_t e_; ctx->dist_rows(n, r, &starts[r], &e_); }
                starts[D.world] = n;
            } else {
                // a later level: row ranges of the replicated induced graph balanced by STORED ENTRIES (its rows run from a handful
                // of entries to a mega row of 10^5); every rank holds the same graph, so the ranges need no exchange
                // (a transport without ragged slices stages a rank's labels through its equal slice of the exchange buffer: equal
                // row counts then, which always fit)
                const bool ragged_ok = D.starts && (D.comm || D.allgatherv);
                if (ragged_ok) {
                    std::vector<int64_t> hip_(n + 1);
                    { const int rb_ = b3c_readback(ctx, hip_.data(), g.indptr, sizeof(int64_t) * (size_t)(n + 1)); if (rb_) return rb_; }
                    const int64_t tot_e = hip_[n];
                    for (int r = 1; r < D.world; r++) {
                        const int64_t target = (int64_t)((unsigned __int128)tot_e * (unsigned)r / (unsigned)D.world);
                        int64_t cut = std::lower_bound(hip_.begin(), hip_.end(), target) - hip_.begin();
                        cut = std::min<int64_t>(n, (cut + 15) / 16 * 16);
                        starts[r] = std::max(cut, starts[r - 1]);
                    }
                } else {
                    const int64_t per = ((n + D.world - 1) / D.world + 15) / 16 * 16;
                    for (int r = 1; r < D.world; r++) starts[r] = std::min<int64_t>(n, (int64_t)r * per);
                }
                starts[D.world] = n;
            }
            HIP_TRY(ctx, hipMemcpyAsync(starts_dev.p, starts.data(), sizeof(int64_t) * starts.size(), hipMemcpyHostToDevice, s));
            HIP_TRY(ctx, hipStreamSynchronize(s));
        }
        const int64_t r0 = dist ? starts[D.rank] : 0;
        const int64_t r1 = dist ? starts[D.rank + 1] : n;
        int32_t *nxt_p = nxt.p;
        const int gwl = grid_for(r1 - r0, WPB, 16 * ctx->n_cu);   // wave-per-row kernels over this rank's rows
        const int gw = grid_for(n, WPB, 16 * ctx->n_cu);     // wave-per-row kernels
        const int gv = grid_for(n, BLK, 8 * ctx->n_cu);      // thread-per-node kernels
        HIP_TRY(ctx, hipMemsetAsync(ctr.p, 0, sizeof(u64) * 8, s));
        const bool form_q = first && dq.src != nullptr;       // level 0: the weights are quantised by this pass (see k_lv_init)
        hipLaunchKernelGGL(k_lv_init, dim3(gw), dim3(BLK), 0, s, n, g.indptr, g.indices, form_q ? nullptr : g.q, ids, kdeg.p, tot.p, com.p,
                           nxt_p, csize.p, ctr.p, wnl.p, r0, r1, ctr.p + 2, form_q ? dq.src : nullptr, dq.scl, (int)shift, form_q ? q0.p : nullptr);
        hipLaunchKernelGGL(k_max_degree, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 2048)), dim3(256), 0, s, n, g.indptr, ctr.p + 7);
        if (blk0) {
            // degrees (and which rows exist) of the other ranks' rows, then totals / sizes / labels / 2m from the whole vector
            if (r1 > r0) hipLaunchKernelGGL(k_lv_pack_flag, dim3((unsigned)((r1 - r0 + 255) / 256)), dim3(256), 0, s, r0, r1, g.indptr, kdeg.p);
            if ((rc = b3c_dist_allgather_rows(ctx, kdeg.p, 8, starts.data(), starts_dev.p))) { free_graph(ctx, g); return rc; }
            HIP_TRY(ctx, hipMemsetAsync(ctr.p, 0, sizeof(u64), s));
            hipLaunchKernelGGL(k_lv_unpack_init, dim3(gv), dim3(BLK), 0, s, n, ids, kdeg.p, tot.p, csize.p, first ? labels.p : (int32_t *)nullptr, ctr.p);
        }
        u64 hc[8];
        { const int rb_ = b3c_readback(ctx, hc, ctr.p, sizeof(hc)); if (rb_) return rb_; }
        const u64 m2 = hc[0];
        const int64_t maxdeg = (int64_t)hc[7];
        if (m2 == 0) break;   // no links: every node alone (generate_dendrogram special case)

        // ---- row-length classes of this level (see lv_list_segment) ----
        const double avgdeg = (double)g.nnz / (double)n;
        const int short_max = avgdeg * 1.3 <= 32 ? 32 : avgdeg * 1.3 <= 64 ? 64 : avgdeg * 1.3 <= 128 ? 128 : 256;
        const bool grouped = n >= 4096;
        const int64_t nloc = r1 - r0;
        const bool need_wave_kernel = !grouped || maxdeg > short_max;
        // Levels with node lists (`grouped`): rows beyond the grouped kernel's reach get one WORKGROUP each with a table in LDS
        // (k_lv_decide_wg: up to 6553 entries); the wave-per-node kernel keeps the rows of up to 409 entries (512-slot tables: six
        // workgroups per CU; rows of 257 .. 512 entries take the grouped kernel with eight entries per lane instead).  Levels without
        // lists (under 4096 nodes, beyond the direct-table kernel's 2048): the wave-per-node kernel on all rows.
        const bool use_wg = grouped;
        const bool wide_g = short_max == 256;                 // rows of 257 .. 512 entries by k_lv_decide_g<64, 8>
        const int wave_thr = use_wg ? std::min(lv_dwg_max_deg(0), std::max(short_max, wide_g ? 512 : 409)) : 0;
        // LDS table of the wave-per-node kernel: 12 bytes per slot and wave, four waves per workgroup (2048 slots are 96 KB, one
        // workgroup per CU: small levels only); rows beyond it use their table in global memory
        const int64_t wave_max = use_wg ? std::min<int64_t>(maxdeg, std::max<int64_t>(64, wave_thr)) : maxdeg;
        const int64_t need = 1ll << table_lg(wave_max);
        const int cap = need <= 128 ? 128 : need <= 256 ? 256 : need <= 512 ? 512 : need <= 1024 ? 1024 : 2048;
        // rows beyond mega_deg entries get workgroups and the node's table in global memory (k_lv_mega_*): whatever the largest LDS
        // table of the workgroup kernel cannot hold (or fewer: B3C_LV_MEGA_DEG, which the tests use to reach those kernels on small graphs)
        const int mega_deg = (int)std::min<int64_t>(lv_dwg_max_deg(3), std::max<int64_t>(short_max, CFG.num("B3C_LV_MEGA_DEG", lv_dwg_max_deg(3))));
        // rows whose table does not fit the LDS of the kernel that takes them use a table in global memory (goff / gkeys / gvals)
        const int cap_tbl = use_wg ? (1 << 13) : cap;
        hipLaunchKernelGGL(k_big_table_sizes, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, s, n, cap_tbl, use_wg ? (int64_t)mega_deg : INT64_MAX,
                           g.indptr, gsz.p);
        if ((rc = b3c_exclusive_scan_i64(ctx, gsz.p, goff.p, n + 1))) { free_graph(ctx, g); return rc; }
        int64_t gtab = 0;
        { const int rb_ = b3c_readback(ctx, &gtab, goff.p + n, sizeof(int64_t)); if (rb_) return rb_; }
        DevBuf<int32_t> gkeys(ctx, gtab + 1);
        DevBuf<u64> gvals(ctx, gtab + 1);
        if (!gkeys.p || !gvals.p) { free_graph(ctx, g); return ctx->fail(B3C_E_NOMEM, "b3c_louvain: hub tables"); }
        if (gtab) {
            HIP_TRY(ctx, hipMemsetAsync(gkeys.p, 0xFF, sizeof(int32_t) * (size_t)gtab, s));
            HIP_TRY(ctx, hipMemsetAsync(gvals.p, 0, sizeof(u64) * (size_t)gtab, s));
        }

        // node lists: per sub-round, the nodes of this rank by row-length class: one flag pass, one scan, one fill, one read-back
        // of the boundaries for all sub-rounds
        DevBuf<NodeRec> lists(ctx, grouped ? nloc + 1 : 1);
        const NodeRec *nlist_p = lists.p;
        // off[r][k] .. off[r][k + 1]: segment k of sub-round r in the common list (k = 0 grouped, 1 wave kernel, 2 .. 5 workgroup
        // kernel, 6 mega rows)
        std::vector<std::array<int64_t, LV_SEGS + 1>> off(n_sub);
        for (auto &o : off) o.fill(0);
        LvThr thr;
        thr.t[0] = short_max;
        if (use_wg) {
            // ascending by construction (lv_list_segment counts the bounds a row exceeds): the wave kernel's bound never passes the
            // first workgroup class -- a row of the 1024-slot class with a longer table would write past its LDS
            thr.t[2] = std::min(mega_deg, std::max(short_max, lv_dwg_max_deg(0)));
            thr.t[3] = std::min(mega_deg, std::max(short_max, lv_dwg_max_deg(1)));
            thr.t[4] = std::min(mega_deg, std::max(short_max, lv_dwg_max_deg(2)));
            thr.t[5] = mega_deg;
            thr.t[1] = std::min(thr.t[2], std::max(short_max, wave_thr));
        } else {
            thr.t[1] = thr.t[2] = thr.t[3] = thr.t[4] = thr.t[5] = mega_deg;
        }
        if (lv_debug) fprintf(stderr, "[b3c] level %d: %lld nodes, %lld entries, longest row %lld: classes <= %d | %d | %d | %d | %d | %d | mega\n",
                              (int)info->n_levels, (long long)n, (long long)g.nnz, (long long)maxdeg, thr.t[0], thr.t[1], thr.t[2], thr.t[3], thr.t[4], thr.t[5]);
        if (grouped) {
            const int n_seg = need_wave_kernel ? LV_SEGS * n_sub : n_sub;
            const int64_t tot = (int64_t)n_seg * nloc + 1;
            DevBuf<int64_t> lflag(ctx, tot), lpos(ctx, tot), bnd(ctx, n_seg + 1);
            if (!lists.p || !lflag.p || !lpos.p || !bnd.p) { free_graph(ctx, g); return ctx->fail(B3C_E_NOMEM, "b3c_louvain: node lists"); }
            const unsigned gl = (unsigned)((nloc + 255) / 256);
            HIP_TRY(ctx, hipMemsetAsync(lflag.p, 0, sizeof(int64_t) * (size_t)tot, s));
            if (nloc) hipLaunchKernelGGL(k_lv_list_flags, dim3(gl), dim3(256), 0, s, r0, r1, n_sub, thr, need_wave_kernel ? 1 : 0, g.indptr, ids, kdeg.p, lflag.p);
            if ((rc = b3c_exclusive_scan_i64(ctx, lflag.p, lpos.p, tot))) { free_graph(ctx, g); return rc; }
            if (nloc) hipLaunchKernelGGL(k_lv_list_fill, dim3(gl), dim3(256), 0, s, r0, r1, n_sub, thr, need_wave_kernel ? 1 : 0, g.indptr, ids, kdeg.p, lpos.p, lists.p);
            hipLaunchKernelGGL(k_lv_list_bounds, dim3(1), dim3(64), 0, s, n_seg, nloc, lpos.p, bnd.p);
            std::vector<int64_t> hb(n_seg + 1, 0);
            { const int rb_ = b3c_readback(ctx, hb.data(), bnd.p, sizeof(int64_t) * hb.size()); if (rb_) return rb_; }
            for (int r = 0; r < n_sub; r++)
                for (int k = 0; k <= LV_SEGS; k++)
                    off[r][k] = need_wave_kernel ? hb[LV_SEGS * r + k] : hb[r + (k > 0 ? 1 : 0)];
        }
        // [loff[r], hoff[r]) grouped nodes of sub-round r, [hoff[r], moff[r]) its hubs (by class: off[r][1 .. 5]), [moff[r], lend[r]) its mega rows
        std::vector<int64_t> loff(n_sub + 1, 0), hoff(n_sub + 1, 0), moff(n_sub + 1, 0), lend(n_sub + 1, 0);
        for (int r = 0; r < n_sub; r++) { loff[r] = off[r][0]; hoff[r] = off[r][1]; moff[r] = off[r][LV_SEGS - 1]; lend[r] = off[r][LV_SEGS]; }

        // full: recount the internal weight; otherwise ctr[2] was kept up to date by k_lv_delta_internal.  The sums are
        // launched before the move count of the sweep is known and come back with it in ONE read-back (a host round
        // trip costs as much as k_lv_tot2); moved == 0 leaves *out alone.
        // exact pruning of the neighbour scan (k_lv_prune): single GPU, grouped lists, during the sweeps that track the
        // internal weight through the movers' rows (the same pass keeps own[] current)
        // Row-sharded runs prune too (round 3): validity stamps replace the exchange the neighbours' own[] would need
        const bool can_prune = grouped && !CFG.on("B3C_LV_NO_PRUNE");
        const bool mega_lds = use_wg;                          // mega-row chunks summed per community in LDS first (from the second sweep on)
        // band of the grouped scan's contender rule (k_lv_decide_g): (k_i >> band_shift) + 1, the tot * k / 2m of a community of about
        // 32 average nodes of this level.  Any value gives the same decisions; it only steers how often the exact fall-back runs
        const int stream_nt = g.nnz * 12 > CFG.num("B3C_LV_NT_MIN", LV_STREAM_NT_MIN) ? 1 : 0;      // (lv_stream; the tests force it with 0)
        int band_shift = 0;
        while ((2ll << band_shift) <= n) band_shift++;          // floor(log2 n)
        band_shift = std::max(0, band_shift - 5);
        const size_t mega_lds_bytes = (size_t)(1 << MEGA_LDS_LG) * 12 + (MEGA_NT / 64) * 8;
        if (use_wg && ctx->attr_first(3)) {
            // tables beyond 64 KB of LDS are dynamic shared memory that a kernel must be allowed first (once per context = device)
            HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_lv_decide_wg<13, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lv_wg_lds<13, 1024>()));
            HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_lv_mega_insert_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mega_lds_bytes));
        }
        int64_t *own_p = can_prune ? own.p : nullptr;
        const bool stamps = can_prune && dist;
        DevBuf<int32_t> stamp(ctx, stamps ? n : 1), tstamp(ctx, stamps ? n : 1);
        if (!stamp.p || !tstamp.p) return ctx->fail(B3C_E_NOMEM, "b3c_louvain: stamps");
        if (stamps) {
            HIP_TRY(ctx, hipMemsetAsync(stamp.p, 0, sizeof(int32_t) * (size_t)n, s));
            HIP_TRY(ctx, hipMemsetAsync(tstamp.p, 0, sizeof(int32_t) * (size_t)n, s));
        }
        int32_t *stamp_p = stamps ? stamp.p : nullptr, *tstamp_p = stamps ? tstamp.p : nullptr;
        int32_t epoch = 0;                               // sub-rounds of this level so far
        int64_t max_list = 0;
        for (int r = 0; r < n_sub; r++) max_list = std::max(max_list, hoff[r] - loff[r]);
        int64_t max_hlist = 0, max_mega = 0;
        for (int r = 0; r < n_sub; r++) { max_hlist = std::max(max_hlist, moff[r] - hoff[r]); max_mega = std::max(max_mega, lend[r] - moff[r]); }
        DevBuf<u64> mega_own(ctx, max_mega + 1);            // own-community weight of the mega rows of a sub-round (cleared by the eval pass)
        // split evaluation of the mega rows' tables (k_lv_mega_eval_split): partial results and tickets
        const bool mega_split = max_mega > 0;
        const int mega_echunks = mega_split ? (int)(((1ll << table_lg(maxdeg)) + MEGA_ESL - 1) / MEGA_ESL) : 1;
        DevBuf<MegaPart> mega_part(ctx, mega_split ? (size_t)max_mega * mega_echunks + 1 : 1);
        DevBuf<unsigned int> mega_ticket(ctx, max_mega + 1);
        if (!mega_part.p || !mega_ticket.p) return ctx->fail(B3C_E_NOMEM, "b3c_louvain: mega rows");
        HIP_TRY(ctx, hipMemsetAsync(mega_ticket.p, 0, sizeof(unsigned int) * (size_t)(max_mega + 1), s));
        if (!mega_own.p) return ctx->fail(B3C_E_NOMEM, "b3c_louvain: mega rows");
        HIP_TRY(ctx, hipMemsetAsync(mega_own.p, 0, sizeof(u64) * (size_t)(max_mega + 1), s));
        DevBuf<NodeRec> plist(ctx, can_prune ? max_list + 1 : 1), plist_h(ctx, can_prune ? max_hlist + 1 : 1), plist_m(ctx, can_prune ? max_mega + 1 : 1);
        DevBuf<unsigned long long> pcnt(ctx, LV_SEGS);    // [0] grouped nodes left after pruning, [1 .. 5] hubs left, by class
        if (!plist.p || !plist_h.p || !plist_m.p || !pcnt.p) { free_graph(ctx, g); return ctx->fail(B3C_E_NOMEM, "b3c_louvain: pruned list"); }
        auto modularity = [&](double *out, bool full, u64 *moved) -> int {
            HIP_TRY(ctx, hipMemsetAsync(ctr.p + 3, 0, sizeof(u64) * 4, s));
            if (full) {
                HIP_TRY(ctx, hipMemsetAsync(ctr.p + 2, 0, sizeof(u64), s));
                const bool long_int = grouped && max_mega > 0;
                // (moved != nullptr: the recount after a sweep -- ctr[1] holds that sweep's move count, see the kernel's HALF mode)
                const u64 *mv_dev = moved ? ctr.p + 1 : nullptr;
                hipLaunchKernelGGL(k_lv_internal, dim3(gwl), dim3(BLK), 0, s, r0, r1, g.indptr, g.indices, g.q, com.p, ctr.p + 2, own_p,
                                   stamp_p, epoch + 1, long_int ? (int64_t)mega_deg : INT64_MAX, mv_dev, n);
                if (long_int) {
                    const int chunks = (int)((maxdeg + 4095) / 4096);
                    for (int r = 0; r < n_sub; r++)
                        if (lend[r] > moff[r])
                            hipLaunchKernelGGL(k_lv_internal_long, dim3((unsigned)((lend[r] - moff[r]) * chunks)), dim3(1024), 0, s, nlist_p + moff[r], chunks,
                                               g.indices, g.q, com.p, ctr.p + 2, own_p, mv_dev, n);
                }
            }
            hipLaunchKernelGGL(k_lv_tot2, dim3(gv), dim3(BLK), 0, s, n, tot.p, ctr.p + 3);
            u64 h[8];
            { const int rb_ = b3c_readback(ctx, h, ctr.p, sizeof(h)); if (rb_) return rb_; }
            if (moved) {
                *moved = h[1];
                if (h[1] == 0) return B3C_OK;
            }
            if (dist) {
                // integer partial sums of the ranks: exact, so the order of the final addition is irrelevant
                HIP_TRY(ctx, hipMemcpyAsync(D.isum + 2 * D.rank, ctr.p + 2, sizeof(u64), hipMemcpyDeviceToDevice, s));
                { const int rx_ = b3c_dist_exchange(ctx, B3C_BUF_ISUM); if (rx_) return rx_; }
                std::vector<int64_t> parts(2 * (size_t)D.world);
                { const int rb_ = b3c_readback(ctx, parts.data(), D.isum, sizeof(int64_t) * parts.size()); if (rb_) return rb_; }
                u64 tot_i2 = 0;
                for (int r = 0; r < D.world; r++) tot_i2 += (u64)parts[2 * (size_t)r];
                h[2] = tot_i2;
            }
            unsigned __int128 t2 = (unsigned __int128)h[3] + ((unsigned __int128)h[4] << 32) +
                                   ((unsigned __int128)h[5] << 64) + ((unsigned __int128)h[6] << 96);
            i128 num = (i128)m2 * (i128)h[2] - (i128)t2;
            *out = (double)num / ((double)m2 * (double)m2);
            return B3C_OK;
        };

        // every exit from here on -- an early return between a fork onto the side streams and its join included -- waits for the side
        // streams BEFORE the blocks their kernels read and write (declared above: tables, lists, nxt, mega_own) go back to the pool
        struct SideJoin {
            b3c_ctx *c;
            ~SideJoin() { if (c->side) { (void)hipStreamSynchronize(c->side); (void)hipStreamSynchronize(c->side2); } }
        } side_join{ctx};
        double cur_mod, new_mod;
        // all nodes alone: k_lv_init has left the self-loop weight in ctr[2], no scan of the graph needed
        if ((rc = modularity(&new_mod, false, nullptr))) { free_graph(ctx, g); return rc; }
        int sweep = 0;
        u64 prev_moved = ~0ull;
        // (tracking from the first sweep on: 15.3 ms against 14.0 at C3 on the variant law -- its sweeps 2 and 3 still move 77 % and 25 %
        // of the nodes.  Round 6, contract law: the bar is 35 % instead of 25 % -- level 0's third sweep follows one that moved 32 %,
        // moves 13 % itself and was recounted in full for 1.3 ms; level 1's fifth follows 33.6 % and moves 1 %)
        for (;;) {
            cur_mod = new_mod;
            sweep++;
            info->n_sweeps++;
            HIP_TRY(ctx, hipMemsetAsync(ctr.p + 1, 0, sizeof(u64), s));
            // few movers expected (the sweep before moved under a quarter of the nodes): track the internal weight through
            // the movers' rows instead of rescanning the whole graph after the sweep
            const bool incr = grouped && prev_moved != ~0ull && lv_track_next(prev_moved, (u64)n);
            for (int r = 0; r < n_sub; r++) {
                epoch++;
                evs.rec();
                // (both pruned-list counters are cleared BEFORE the fork below: the side stream only waits for what precedes it)
                if (incr && can_prune) HIP_TRY(ctx, hipMemsetAsync(pcnt.p, 0, sizeof(unsigned long long) * LV_SEGS, s));
                // many hub nodes or any mega row: they run on side streams BESIDE the grouped scan (both read the state of the
                // sub-round's start and write nxt[] of their own nodes).  A handful of hubs (C3: 17 us of work) stays on the main
                // stream: forking and joining costs as much as it hides (A/B at C3: 14.80 ms with, 14.70 without)
                const bool hub_side = grouped && need_wave_kernel && hoff[r] > loff[r] && (lend[r] > moff[r] || moff[r] - hoff[r] >= SIDE_MIN_HUBS) &&
                                      ctx->side_ok();
                hipStream_t hs = s, ms = s;               // streams of the hub and of the mega-row kernels
                if (hub_side) {
                    HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, s));
                    HIP_TRY(ctx, hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
                    hs = ctx->side;
                    if (lend[r] > moff[r]) {
                        HIP_TRY(ctx, hipStreamWaitEvent(ctx->side2, ctx->ev_fork, 0));
                        ms = ctx->side2;
                    } else {
                        ms = hs;
                    }
                }
                if (grouped && hoff[r] > loff[r]) {
                    const int64_t cntr = hoff[r] - loff[r];
                    const NodeRec *lst = nlist_p + loff[r];
                    const int64_t *cnt_dev = nullptr;
                    if (incr && can_prune) {
                        // nodes that provably stay are settled from four numbers each; the rest go through the scan
                        hipLaunchKernelGGL(k_lv_prune, dim3(grid_for(cntr, 4 * BLK, 8 * ctx->n_cu)), dim3(BLK), 0, s, cntr, lst, m2, kdeg.p,
                                           wnl.p, own.p, tot.p, com.p, nxt_p, plist.p, pcnt.p, stamp_p, tstamp_p);
                        lst = plist.p;
                        cnt_dev = (const int64_t *)pcnt.p;
                    }
                    // grids of a multiple of eight workgroups number their list positions XCD by XCD (see the kernel)
#define LV_GRID_G(L) ([&]() { int g_ = grid_for(cntr, WPB * (64 / L), 16 * ctx->n_cu); if (g_ >= 64) g_ = (g_ + 7) & ~7; return g_; }())
#define LV_DECIDE_G(L, U)                                                                                                \
    { const int g_ = LV_GRID_G(L);                                                                                       \
      hipLaunchKernelGGL((k_lv_decide_g<L, U>), dim3(g_), dim3(BLK), 0, s, cntr, cnt_dev, lst,                            \
                       m2, g.indices, g.q, kdeg.p, tot.p, com.p, csize.p, nxt_p, incr && stamps ? own.p : nullptr, stamp_p, epoch, \
                       g_ >= 64 ? 1 : 0, band_shift, stream_nt); }
                    if (sweep == 1 && r == 0 && !incr && !CFG.on("B3C_LV_NO_FIRST")) {
                        // every node is still alone: the table-free form (k_lv_first_g)
#define LV_FIRST_G(L, U)                                                                                                 \
    { const int g_ = LV_GRID_G(L);                                                                                       \
      hipLaunchKernelGGL((k_lv_first_g<L, U>), dim3(g_), dim3(BLK), 0, s, cntr, lst, m2,                                  \
                       g.indices, g.q, kdeg.p, com.p, csize.p, nxt_p, g_ >= 64 ? 1 : 0, band_shift); }
                        switch (short_max) {
                        case 256: LV_FIRST_G(64, 4); break;
                        case 128: LV_FIRST_G(16, 8); break;
                        case 64: LV_FIRST_G(8, 8); break;
                        default: LV_FIRST_G(4, 8); break;
                        }
#undef LV_FIRST_G
                    } else
                    switch (short_max) {
                    case 256: LV_DECIDE_G(64, 4); break;      // (C3 sweep of shapes, ms of delta-Q scan per step: 64x4 11.3, 32x8 12.2, 16x16 22.4)
                    case 128: LV_DECIDE_G(16, 8); break;
                    case 64: LV_DECIDE_G(8, 8); break;
                    default: LV_DECIDE_G(4, 8); break;
                    }
#undef LV_DECIDE_G
#undef LV_GRID_G
                }
                const bool direct_level = !grouped && !dist && n <= LV_DIRECT_CAP;
                if (direct_level) {
                    // small level: direct community table in LDS
                    hipLaunchKernelGGL(k_lv_decide_direct<LV_DIRECT_CAP>, dim3(grid_for(n, WPB, 16 * ctx->n_cu)), dim3(BLK), 0, s, n, n_sub, r, m2,
                                       g.indptr, g.indices, g.q, ids, kdeg.p, tot.p, com.p, csize.p, nxt_p);
                }
                if (grouped && lend[r] > moff[r]) {       // mega rows: workgroups per node (side stream, beside the others)
                    const int64_t nm = lend[r] - moff[r];
                    const int chunks = (int)((maxdeg + MEGA_CHUNK - 1) / MEGA_CHUNK);
                    // pruned like the other classes in the sweeps that track own[] (round 6: BASELINE's C5 has thousands of rows of
                    // 10^4 .. 10^6 entries, most of them settled after the first sweeps, and inserting and evaluating their tables
                    // in every sub-round was two thirds of its Louvain stage).  Row-sharded runs keep scanning them (see k_lv_prune)
                    const NodeRec *mlist = nlist_p + moff[r];
                    const int64_t *mcnt_dev = nullptr;
                    if (incr && can_prune && !stamps) {
                        hipLaunchKernelGGL(k_lv_prune, dim3(grid_for(nm, 4 * BLK, 8 * ctx->n_cu)), dim3(BLK), 0, ms, nm, mlist, m2, kdeg.p,
                                           wnl.p, own.p, tot.p, com.p, nxt_p, plist_m.p, pcnt.p + (LV_SEGS - 1), stamp_p, tstamp_p);
                        mlist = plist_m.p;
                        mcnt_dev = (const int64_t *)(pcnt.p + (LV_SEGS - 1));
                    }
                    // from the second sweep on the chunk's entries are summed per community in LDS before they reach the node's table
                    if (mega_lds && sweep > 1)
                        hipLaunchKernelGGL(k_lv_mega_insert_lds, dim3((unsigned)(nm * chunks)), dim3(MEGA_NT), mega_lds_bytes, ms, mlist, chunks, g.indices, g.q,
                                           com.p, goff.p, gkeys.p, gvals.p, mega_own.p, mcnt_dev);
                    else
                        hipLaunchKernelGGL(k_lv_mega_insert, dim3((unsigned)(nm * chunks)), dim3(MEGA_NT), 0, ms, mlist, chunks, g.indices, g.q,
                                           com.p, goff.p, gkeys.p, gvals.p, mega_own.p, mcnt_dev);
                    hipLaunchKernelGGL(k_lv_mega_eval_split, dim3((unsigned)(nm * mega_echunks)), dim3(MEGA_NT), 0, ms, mlist, mega_echunks, m2, kdeg.p,
                                       tot.p, com.p, csize.p, nxt_p, goff.p, gkeys.p, gvals.p, mega_own.p, mega_part.p, mega_ticket.p, mcnt_dev);
                    if (ms != s && ms != hs) {
                        HIP_TRY(ctx, hipEventRecord(ctx->ev_join2, ms));
                        HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->ev_join2, 0));
                    }
                }
                if (!direct_level && need_wave_kernel && (!grouped || moff[r] > hoff[r])) {
#define LV_DECIDE(C, W0, W1, HUBS, HCNT)                                                                                 \
    hipLaunchKernelGGL(k_lv_decide<C>, dim3(grid_for((W1) - (W0), WPB, 16 * ctx->n_cu)), dim3(BLK), 0, hs, W0, W1, HUBS, -1, n_sub, r, m2, g.indptr, \
                       g.indices, g.q, ids, kdeg.p, tot.p, com.p, csize.p, nxt_p, goff.p, gkeys.p, gvals.p, HCNT)
#define LV_DECIDE_CAP(W0, W1, HUBS, HCNT)                                                                                \
    switch (cap) {                                                                                                       \
    case 128: LV_DECIDE(128, W0, W1, HUBS, HCNT); break;                                                                 \
    case 256: LV_DECIDE(256, W0, W1, HUBS, HCNT); break;                                                                 \
    case 512: LV_DECIDE(512, W0, W1, HUBS, HCNT); break;                                                                 \
    case 1024: LV_DECIDE(1024, W0, W1, HUBS, HCNT); break;                                                               \
    default: LV_DECIDE(2048, W0, W1, HUBS, HCNT); break;                                                                 \
    }
                    if (!grouped) {
                        // all rows of this rank (small levels)
                        LV_DECIDE_CAP(r0, r1, (const NodeRec *)nullptr, (const int64_t *)nullptr);
                    } else {
                        // the hubs the grouped kernel left out, class by class: [1] one wave per node, [2 .. 5] one workgroup per node
                        for (int k = 1; k < LV_SEGS - 1; k++) {
                            const int64_t w1 = off[r][k + 1] - off[r][k];
                            if (w1 <= 0) continue;
                            const NodeRec *hubs = nlist_p + off[r][k];
                            const int64_t *hcnt_dev = nullptr;
                            if (incr && can_prune && !stamps && w1 >= PRUNE_MIN_HUBS) {
                                // the hubs are pruned like the grouped nodes (round 3: on a heavy-tailed map with hundreds of entries per
                                // row -- C5 -- most of the WEIGHT sits in rows beyond the grouped kernel's 256 entries, and scanning all of
                                // them in every late sweep was 85 % of the stage).  Row-sharded runs keep scanning them: these kernels
                                // do not refresh own[] / stamp[] (see k_lv_prune)
                                NodeRec *pl = plist_h.p + (off[r][k] - off[r][1]);
                                hipLaunchKernelGGL(k_lv_prune, dim3(grid_for(w1, 4 * BLK, 8 * ctx->n_cu)), dim3(BLK), 0, hs, w1, hubs, m2, kdeg.p,
                                                   wnl.p, own.p, tot.p, com.p, nxt_p, pl, pcnt.p + k, stamp_p, tstamp_p);
                                hubs = pl;
                                hcnt_dev = (const int64_t *)(pcnt.p + k);
                            }
                            if (k == 1 && use_wg && wide_g && thr.t[1] <= 512) {
                                // rows of 257 .. 512 entries (15 % of the rows at C3, a quarter of the entries): the grouped kernel
                                // with eight entries per lane -- entries in registers, claims instead of a scan of the table -- on
                                // the hub stream, instead of the wave-per-node kernel
                                const int g_ = grid_for(w1, WPB, 16 * ctx->n_cu);
                                hipLaunchKernelGGL((k_lv_decide_g<64, 8>), dim3(g_), dim3(BLK), 0, hs, w1, hcnt_dev, hubs, m2, g.indices, g.q, kdeg.p, tot.p,
                                                   com.p, csize.p, nxt_p, (int64_t *)nullptr, (int32_t *)nullptr, epoch, 0, band_shift, stream_nt);
                            } else if (k == 1 || !use_wg) {
                                LV_DECIDE_CAP((int64_t)0, w1, hubs, hcnt_dev);
                            } else {
#define LV_DECIDE_WG(LGS, NT, MAXG)                                                                                      \
    hipLaunchKernelGGL((k_lv_decide_wg<LGS, NT>), dim3((unsigned)std::min<int64_t>(w1, MAXG)), dim3(NT), (lv_wg_lds<LGS, NT>()), hs, w1, hcnt_dev,  \
                       hubs, m2, g.indices, g.q, kdeg.p, tot.p, com.p, csize.p, nxt_p)
                                if (k == 2) LV_DECIDE_WG(10, 256, 16 * (int64_t)ctx->n_cu);
                                else if (k == 3) LV_DECIDE_WG(11, 256, 12 * (int64_t)ctx->n_cu);
                                else if (k == 4) LV_DECIDE_WG(12, 256, 6 * (int64_t)ctx->n_cu);
                                else LV_DECIDE_WG(13, 1024, 2 * (int64_t)ctx->n_cu);
#undef LV_DECIDE_WG
                            }
                        }
                    }
#undef LV_DECIDE_CAP
#undef LV_DECIDE
                }
                if (hub_side) {
                    HIP_TRY(ctx, hipEventRecord(ctx->ev_join, ctx->side));
                    HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->ev_join, 0));
                }
                evs.rec();
                // every rank decided for its rows: gather the decisions
                if (dist && (rc = b3c_dist_allgather_rows(ctx, nxt.p, 4, starts.data(), starts_dev.p))) { free_graph(ctx, g); return rc; }
                info->n_move_launch++;
                if (incr) {
                    // grouped nodes and hubs of the sub-round are adjacent in the list: one launch
                    const int64_t c1 = lend[r] - loff[r];
                    if (c1) hipLaunchKernelGGL(k_lv_delta_internal, dim3(grid_for(c1, WPB * LV_DELTA_NPW, 32 * ctx->n_cu)), dim3(BLK), 0, s, c1,
                                               nlist_p + loff[r], g.indices, g.q, com.p, nxt_p, ctr.p + 2, own_p, r0, r1, stamp_p, epoch + 1);
                }
                if (grouped && !dist) {
                    // only the nodes that decided in this sub-round can have moved
                    const int64_t c1 = lend[r] - loff[r];
                    if (c1) hipLaunchKernelGGL(k_lv_apply_list, dim3(grid_for(c1, BLK, 4 * ctx->n_cu)), dim3(BLK), 0, s, c1,
                                               nlist_p + loff[r], kdeg.p, tot.p, com.p, nxt_p, csize.p, ctr.p + 1);
                } else {
                    hipLaunchKernelGGL(k_lv_apply, dim3(gv), dim3(BLK), 0, s, n, n_sub, r, ids, kdeg.p, tot.p, com.p, nxt_p,
                                       csize.p, ctr.p + 1, r0, r1, tstamp_p, epoch);
                }
            }
            info->move_nnz += g.nnz;
            info->move_rows += g.n;
            u64 moved = 0;
            if ((rc = modularity(&new_mod, !incr, &moved))) { free_graph(ctx, g); return rc; }
            info->n_moves += (int64_t)moved;
            if (lv_debug) fprintf(stderr, "[b3c] level %d sweep %d: %llu of %lld nodes moved\n", (int)info->n_levels, sweep, (unsigned long long)moved, (long long)n);
            if (moved == 0) break;
            prev_moved = moved;
            if (new_mod - cur_mod < PL_MIN) break;
            if (max_sweeps > 0 && sweep >= max_sweeps) break;
        }
        if (!first && new_mod - mod < PL_MIN) break;
        first = false;
        mod = new_mod;
        info->n_levels++;

        // ---- renumber by ascending community id, relabel the original nodes ----
        DevBuf<int64_t> used(ctx, n + 1), newid(ctx, n + 1);
        DevBuf<int32_t> cn(ctx, n);
        if (!used.p || !newid.p || !cn.p) { free_graph(ctx, g); return ctx->fail(B3C_E_NOMEM, "b3c_louvain: renumber"); }
        hipLaunchKernelGGL(k_flag_used, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, s, n, csize.p, used.p);
        if ((rc = b3c_exclusive_scan_i64(ctx, used.p, newid.p, n + 1))) { free_graph(ctx, g); return rc; }
        int64_t kc = 0;
        HIP_TRY(ctx, hipMemcpyAsync(&kc, newid.p + n, sizeof(int64_t), hipMemcpyDeviceToHost, s));
        hipLaunchKernelGGL(k_node_newcom, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, com.p, newid.p, csize.p, cn.p);
        hipLaunchKernelGGL(k_update_labels, dim3((unsigned)((n0 + 255) / 256)), dim3(256), 0, s, n0, labels.p, cn.p);
        HIP_TRY(ctx, hipStreamSynchronize(s));
        info->n_comm = (int32_t)kc;
        if (info->n_levels == 1) info->n_comm_level0 = kc;

        // ---- induced graph ----
        info->agg_nnz += g.nnz;
        LvGraph h;
        GraphGuard next_guard{ctx, &h};               // an early return while h is under construction releases its blocks
        if (kc <= AGG_BINS_KC && !CFG.on("B3C_LV_AGG_SORT")) {
            // few communities: exact integer sums straight into kc x kc bins (see k_agg_bins)
            const size_t nb = (size_t)kc * (size_t)kc;
            DevBuf<u64> bins(ctx, nb);
            DevBuf<int64_t> rcnt(ctx, kc + 1);
            h.n = kc; h.owned = true;
            h.indptr = (int64_t *)ctx->alloc(sizeof(int64_t) * (size_t)(kc + 1));
            if (!bins.p || !rcnt.p || !h.indptr) {
                free_graph(ctx, h);
                free_graph(ctx, g);
                return ctx->fail(B3C_E_NOMEM, "b3c_louvain: aggregation bins");
            }
            HIP_TRY(ctx, hipMemsetAsync(bins.p, 0, sizeof(u64) * nb, s));
            {
                const int nslot = kc <= 512 ? AGG_SLOTS : kc <= 2048 ? 2 : 1;
                const int64_t rpb = std::max<int64_t>(64, std::min<int64_t>(512, n / (4 * (int64_t)ctx->n_cu)));   // at least four blocks per CU
                const size_t lds = sizeof(u64) * (size_t)nslot * (size_t)kc;       // at most 32 KB
                const bool long_rows = maxdeg > AGG_LONG;
                // beyond the first level: rows in the order of their new community (one stable sort of n labels)
                const bool by_com = !first && n >= 4096;
                DevBuf<int32_t> okey(ctx, by_com ? n : 1), oval(ctx, by_com ? n : 1), oiota(ctx, by_com ? n : 1);
                if (by_com) {
                    if (!okey.p || !oval.p || !oiota.p) { free_graph(ctx, h); free_graph(ctx, g); return ctx->fail(B3C_E_NOMEM, "b3c_louvain: aggregation order"); }
                    hipLaunchKernelGGL(k_iota32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, oiota.p);
                    int kb = 1;
                    while ((1ll << kb) < kc) kb++;
                    size_t tb = 0;
                    HIP_TRY(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tb, cn.p, okey.p, oiota.p, oval.p, n, 0, kb, s));
                    DevBuf<char> tmp(ctx, tb);
                    if (!tmp.p) { free_graph(ctx, h); free_graph(ctx, g); return ctx->fail(B3C_E_NOMEM, "b3c_louvain: aggregation order"); }
                    HIP_TRY(ctx, hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, cn.p, okey.p, oiota.p, oval.p, n, 0, kb, s));
                    HIP_TRY(ctx, hipStreamSynchronize(s));        // the scratch goes back to the pool here
                }
                hipLaunchKernelGGL(k_agg_bins, dim3((unsigned)((n + rpb - 1) / rpb)), dim3(BLK), lds, s, n, kc, rpb, nslot,
                                   long_rows ? AGG_LONG : INT64_MAX, g.indptr, g.indices, g.q, cn.p, bins.p, by_com ? oval.p : nullptr);
                if (long_rows) {
                    DevBuf<int32_t> lrows(ctx, n);
                    DevBuf<unsigned long long> lcnt(ctx, 1);
                    if (!lrows.p || !lcnt.p) { free_graph(ctx, h); free_graph(ctx, g); return ctx->fail(B3C_E_NOMEM, "b3c_louvain: long rows of the aggregation"); }
                    HIP_TRY(ctx, hipMemsetAsync(lcnt.p, 0, sizeof(unsigned long long), s));
                    hipLaunchKernelGGL(k_agg_long_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, g.indptr, lrows.p, lcnt.p);
                    unsigned long long nl = 0;
                    { const int rb_ = b3c_readback(ctx, &nl, lcnt.p, sizeof(nl)); if (rb_) { free_graph(ctx, h); free_graph(ctx, g); return rb_; } }
                    if (nl) {
                        const int chunks = (int)((maxdeg + AGG_LONG - 1) / AGG_LONG);
                        hipLaunchKernelGGL(k_agg_bins_long, dim3((unsigned)(nl * (unsigned long long)chunks)), dim3(BLK), sizeof(u64) * (size_t)kc, s, kc, chunks,
                                           lrows.p, g.indptr, g.indices, g.q, cn.p, bins.p);
                        HIP_TRY(ctx, hipStreamSynchronize(s));        // lrows goes back to the pool at the end of this scope
                    }
                }
                if (by_com) HIP_TRY(ctx, hipStreamSynchronize(s));        // so does the row order
            }
            if (blk0 && (rc = b3c_dist_allreduce_u64(ctx, bins.p, (int64_t)nb))) { free_graph(ctx, h); free_graph(ctx, g); return rc; }
            const unsigned gk = (unsigned)((kc + 1 + WPB - 1) / WPB);
            hipLaunchKernelGGL(k_bins_count, dim3(gk), dim3(BLK), 0, s, kc, bins.p, rcnt.p);
            if ((rc = b3c_exclusive_scan_i64(ctx, rcnt.p, h.indptr, kc + 1))) { free_graph(ctx, h); free_graph(ctx, g); return rc; }
            int64_t nu = 0;
            { const int rb_ = b3c_readback(ctx, &nu, h.indptr + kc, sizeof(int64_t)); if (rb_) { free_graph(ctx, h); free_graph(ctx, g); return rb_; } }
            h.nnz = nu;
            h.indices = (int32_t *)ctx->alloc(sizeof(int32_t) * (size_t)(nu + 8));      // (+ 8: vector loads past the last row)
            h.q = (int64_t *)ctx->alloc(sizeof(int64_t) * (size_t)(nu + 8));
            if (!h.indices || !h.q) {
                free_graph(ctx, h);
                free_graph(ctx, g);
                return ctx->fail(B3C_E_NOMEM, "b3c_louvain: next level");
            }
            hipLaunchKernelGGL(k_bins_fill, dim3(gk), dim3(BLK), 0, s, kc, bins.p, h.indptr, h.indices, h.q);
            HIP_TRY(ctx, hipGetLastError());
            HIP_TRY(ctx, hipStreamSynchronize(s));
        } else {
        // many communities: sort (community pair) keys, sum weights.  On a row block every rank reduces the pairs of ITS rows,
        // the reduced runs travel to the OWNER of their row community (below) and are reduced there once more: integer sums, so the
        // induced graph is bit for bit the single-GPU one whatever the split ("tuples keyed by community pair,
        // then local sort-reduce"; the graph of the next level is small against level 0 and stays replicated).
        const bool agg_hash = !blk0 && !CFG.on("B3C_LV_NO_AGG_HASH");
        if (agg_hash) {
            // ---- by hashing: the rows of a community are read once, no pair ever sorted (see k_aggh_wg) ----
            DevBuf<int32_t> skey(ctx, n), ord(ctx, n), iota(ctx, n), lists(ctx, 4 * (size_t)kc + 1), brows(ctx, n);
            DevBuf<int64_t> cstart(ctx, kc + 1), tsz(ctx, kc + 1), toff(ctx, kc + 1), need(ctx, kc + 1), ubpos(ctx, kc + 1), dcnt(ctx, kc + 1);
            DevBuf<unsigned long long> ub(ctx, kc + 1), lcnt(ctx, 5), own_acc(ctx, kc + 1);
            DevBuf<int32_t> tix(ctx, (size_t)g.nnz + (size_t)kc + 1);
            DevBuf<int64_t> tq(ctx, (size_t)g.nnz + (size_t)kc + 1);
            if (!skey.p || !ord.p || !iota.p || !lists.p || !brows.p || !cstart.p || !tsz.p || !toff.p || !need.p || !ubpos.p || !dcnt.p || !ub.p ||
                !lcnt.p || !own_acc.p || !tix.p || !tq.p)
                return ctx->fail(B3C_E_NOMEM, "b3c_louvain: hash aggregation");
            hipLaunchKernelGGL(k_iota32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, iota.p);
            {
                DevBuf<int32_t> key_in(ctx, n);
                if (!key_in.p) return ctx->fail(B3C_E_NOMEM, "b3c_louvain: hash aggregation");
                hipLaunchKernelGGL(k_aggh_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, kc, cn.p, key_in.p);
                int kb = 1;
                while ((1ll << kb) <= kc) kb++;                // (keys 0 .. kc: kc = "in no community")
                size_t tb = 0;
                HIP_TRY(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tb, key_in.p, skey.p, iota.p, ord.p, n, 0, kb, s));
                DevBuf<char> tmp(ctx, tb);
                if (!tmp.p) return ctx->fail(B3C_E_NOMEM, "b3c_louvain: hash aggregation");
                HIP_TRY(ctx, hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, key_in.p, skey.p, iota.p, ord.p, n, 0, kb, s));
                HIP_TRY(ctx, hipStreamSynchronize(s));        // the scratch goes back to the pool here
            }
            HIP_TRY(ctx, hipMemsetAsync(ub.p, 0, sizeof(unsigned long long) * (size_t)(kc + 1), s));
            HIP_TRY(ctx, hipMemsetAsync(own_acc.p, 0, sizeof(unsigned long long) * (size_t)(kc + 1), s));
            HIP_TRY(ctx, hipMemsetAsync(lcnt.p, 0, sizeof(unsigned long long) * 5, s));
            HIP_TRY(ctx, hipMemsetAsync(dcnt.p, 0, sizeof(int64_t) * (size_t)(kc + 1), s));      // (class 3 counts its claims in it)
            hipLaunchKernelGGL(k_aggh_starts, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, s, n, kc, skey.p, cstart.p);
            hipLaunchKernelGGL(k_aggh_ub, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, g.indptr, cn.p, ub.p);
            hipLaunchKernelGGL(k_aggh_classify, dim3((unsigned)((kc + 256) / 256)), dim3(256), 0, s, kc, ub.p, lists.p, lcnt.p, tsz.p, need.p);
            if ((rc = b3c_exclusive_scan_i64(ctx, tsz.p, toff.p, kc + 1))) return rc;
            if ((rc = b3c_exclusive_scan_i64(ctx, need.p, ubpos.p, kc + 1))) return rc;
            hipLaunchKernelGGL(k_aggh_big_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, cn.p, tsz.p, g.indptr, brows.p, lcnt.p + 4);
            unsigned long long hl[5];
            int64_t gtot = 0;
            { const int rb_ = b3c_readback(ctx, hl, lcnt.p, sizeof(hl)); if (rb_) return rb_; }
            { const int rb_ = b3c_readback(ctx, &gtot, toff.p + kc, sizeof(int64_t)); if (rb_) return rb_; }
            DevBuf<int32_t> hk(ctx, (size_t)gtot + 1);
            DevBuf<u64> hv(ctx, (size_t)gtot + 1);
            if (!hk.p || !hv.p) return ctx->fail(B3C_E_NOMEM, "b3c_louvain: hash aggregation tables");
            if (gtot) {
                HIP_TRY(ctx, hipMemsetAsync(hk.p, 0xFF, sizeof(int32_t) * (size_t)gtot, s));
                HIP_TRY(ctx, hipMemsetAsync(hv.p, 0, sizeof(u64) * (size_t)gtot, s));
            }
            if (ctx->attr_first(4)) HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_aggh_wg<13, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lv_wg_lds<13, 1024>()));
#define LV_AGGH(LGS, NT, CLS, MAXG)                                                                                      \
    if (hl[CLS])                                                                                                         \
        hipLaunchKernelGGL((k_aggh_wg<LGS, NT>), dim3((unsigned)std::min<int64_t>((int64_t)hl[CLS], MAXG)), dim3(NT), (lv_wg_lds<LGS, NT>()), s, (int64_t)hl[CLS], \
                           lists.p + (int64_t)CLS * kc, cstart.p, ord.p, g.indptr, g.indices, g.q, cn.p, ubpos.p, tix.p, tq.p, dcnt.p)
            LV_AGGH(10, 256, 0, 16 * (int64_t)ctx->n_cu);
            LV_AGGH(12, 256, 1, 6 * (int64_t)ctx->n_cu);
            LV_AGGH(13, 1024, 2, 2 * (int64_t)ctx->n_cu);
#undef LV_AGGH
            if (hl[3]) {
                if (hl[4])
                    hipLaunchKernelGGL(k_aggh_big_insert, dim3((unsigned)std::min<int64_t>((int64_t)hl[4], 32 * (int64_t)ctx->n_cu)), dim3(BLK), 0, s, (int64_t)hl[4], brows.p,
                                       g.indptr, g.indices, g.q, cn.p, ub.p, toff.p, hk.p, hv.p, own_acc.p, ubpos.p, tix.p, (unsigned long long *)dcnt.p);
                hipLaunchKernelGGL(k_aggh_big_emit, dim3((unsigned)std::min<int64_t>((int64_t)hl[3], 16 * (int64_t)ctx->n_cu)), dim3(BLK), 0, s, (int64_t)hl[3], lists.p + 3 * kc,
                                   ub.p, toff.p, hk.p, hv.p, own_acc.p, ubpos.p, tix.p, tq.p, (unsigned long long *)dcnt.p);
            }
            h.n = kc; h.owned = true;
            h.indptr = (int64_t *)ctx->alloc(sizeof(int64_t) * (size_t)(kc + 1));
            if (!h.indptr) return ctx->fail(B3C_E_NOMEM, "b3c_louvain: next level");
            if ((rc = b3c_exclusive_scan_i64(ctx, dcnt.p, h.indptr, kc + 1))) { free_graph(ctx, h); return rc; }
            int64_t nu = 0;
            { const int rb_ = b3c_readback(ctx, &nu, h.indptr + kc, sizeof(int64_t)); if (rb_) { free_graph(ctx, h); return rb_; } }
            h.nnz = nu;
            h.indices = (int32_t *)ctx->alloc(sizeof(int32_t) * (size_t)(nu + 8));      // (+ 8: vector loads past the last row)
            h.q = (int64_t *)ctx->alloc(sizeof(int64_t) * (size_t)(nu + 8));
            if (!h.indices || !h.q) { free_graph(ctx, h); return ctx->fail(B3C_E_NOMEM, "b3c_louvain: next level"); }
            hipLaunchKernelGGL(k_aggh_compact, dim3(grid_for(kc, WPB, 16 * ctx->n_cu)), dim3(BLK), 0, s, kc, ubpos.p, h.indptr, tix.p, tq.p, h.indices, h.q);
            {
                const hipError_t e1 = hipGetLastError(), e2 = hipStreamSynchronize(s);      // the scratch is read until here
                if (e1 != hipSuccess || e2 != hipSuccess) {
                    free_graph(ctx, h);
                    return ctx->fail(B3C_E_HIP, "b3c_louvain: hash aggregation failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
                }
            }
        } else {
        DevBuf<int64_t> pcnt(ctx, n + 1), ppos(ctx, n + 1);
        if (!pcnt.p || !ppos.p) return ctx->fail(B3C_E_NOMEM, "b3c_louvain: aggregation counts");
        hipLaunchKernelGGL(k_pairs_count, dim3(gw), dim3(BLK), 0, s, n, g.indptr, g.indices, cn.p, pcnt.p);
        if ((rc = b3c_exclusive_scan_i64(ctx, pcnt.p, ppos.p, n + 1))) return rc;
        int64_t m = 0;
        { const int rb_ = b3c_readback(ctx, &m, ppos.p + n, sizeof(int64_t)); if (rb_) return rb_; }
        int end_bit = 1;
        while (end_bit < 63 && ((unsigned __int128)1 << end_bit) < (unsigned __int128)kc * (unsigned __int128)kc) end_bit++;
        // (keys, vals)[0, cnt) in ka/va -> sorted by key, equal keys summed, back in ka/va; kb/vb are scratch of the same size
        auto sort_reduce = [&](int64_t *ka, int64_t *va, int64_t *kb, int64_t *vb, int64_t cnt, int64_t *nruns_dev, int64_t *nu_out) -> int {
            *nu_out = 0;
            if (cnt == 0) return B3C_OK;
            size_t tb = 0;
            HIP_TRY(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tb, ka, kb, va, vb, cnt, 0, end_bit, s));
            {
                DevBuf<char> tmp(ctx, tb);
                if (!tmp.p) return ctx->fail(B3C_E_NOMEM, "sort workspace");
                HIP_TRY(ctx, hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, ka, kb, va, vb, cnt, 0, end_bit, s));
            }
            tb = 0;
            HIP_TRY(ctx, hipcub::DeviceReduce::ReduceByKey(nullptr, tb, kb, ka, vb, va, nruns_dev, hipcub::Sum(), cnt, s));
            {
                DevBuf<char> tmp(ctx, tb);
                if (!tmp.p) return ctx->fail(B3C_E_NOMEM, "reduce workspace");
                HIP_TRY(ctx, hipcub::DeviceReduce::ReduceByKey(tmp.p, tb, kb, ka, vb, va, nruns_dev, hipcub::Sum(), cnt, s));
            }
            return b3c_readback(ctx, nu_out, nruns_dev, sizeof(int64_t));
        };
        DevBuf<int64_t> nruns(ctx, 1);
        if (!nruns.p) return ctx->fail(B3C_E_NOMEM, "b3c_louvain: aggregation buffers");
        int64_t nu = 0;
        struct Runs { b3c_ctx *c; int64_t *p = nullptr; ~Runs() { c->release(p); } } merged{ctx};   // gathered runs (row blocks)
        const int64_t *ukeys = nullptr, *uvals = nullptr;
        DevBuf<int64_t> k_in(ctx, m), k_out(ctx, m), v_in(ctx, m), v_out(ctx, m);
        if (!k_in.p || !k_out.p || !v_in.p || !v_out.p) return ctx->fail(B3C_E_NOMEM, "b3c_louvain: aggregation buffers");
        if (m) hipLaunchKernelGGL(k_pairs_fill, dim3(gw), dim3(BLK), 0, s, n, kc, g.indptr, g.indices, g.q, cn.p, ppos.p, k_in.p, v_in.p);
        if ((rc = sort_reduce(k_in.p, v_in.p, k_out.p, v_out.p, m, nruns.p, &nu))) return rc;
        ukeys = k_in.p;
        uvals = v_in.p;
        if (blk0) {
            // ---- aggregation by OWNER (round 6) ----
            // The new communities are dealt to the ranks in equal id ranges; every rank sends each owner the reduced runs whose
            // row community the owner holds (the runs are sorted by key = row community * kc + column community: one contiguous
            // segment per owner), the owner reduces what it received -- a community whose members sit in several ranks' blocks has
            // runs from each -- and keeps the rows it owns: the next level is a ROW BLOCK again, nobody ever holds or reduces the
            // whole induced graph (rounds 3-5 gathered all runs to every rank and reduced them there a second time: at C3 on the
            // contract law 1.4e8 pairs on every rank).  Integer sums: the graph is bit for bit the single-GPU one.
            const int W = D.world;
            std::vector<int64_t> cb(W + 1, 0);
            const int64_t per = ((kc + W - 1) / W + 15) / 16 * 16;
            for (int o = 1; o < W; o++) cb[o] = std::min<int64_t>(kc, (int64_t)o * per);
            cb[W] = kc;
            std::vector<int64_t> bnd(W + 1), soff(W + 1, 0);
            for (int o = 0; o <= W; o++) bnd[o] = cb[o] * kc;
            DevBuf<int64_t> bnd_dev(ctx, W + 1), soff_dev(ctx, W + 1);
            DevBuf<u64> cmat_dev(ctx, (size_t)W * W);
            if (!bnd_dev.p || !soff_dev.p || !cmat_dev.p) return ctx->fail(B3C_E_NOMEM, "b3c_louvain: aggregation by owner");
            HIP_TRY(ctx, hipMemcpyAsync(bnd_dev.p, bnd.data(), sizeof(int64_t) * (size_t)(W + 1), hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_key_bounds, dim3(1), dim3(((W + 1 + 63) / 64) * 64), 0, s, nu, k_in.p, W + 1, bnd_dev.p, soff_dev.p);
            { const int rb_ = b3c_readback(ctx, soff.data(), soff_dev.p, sizeof(int64_t) * (size_t)(W + 1)); if (rb_) return rb_; }
            // who sends how much to whom: every rank fills its row of the matrix, an integer all-reduce completes it
            std::vector<int64_t> cmat((size_t)W * W, 0);
            for (int o = 0; o < W; o++) cmat[(size_t)D.rank * W + o] = soff[o + 1] - soff[o];
            HIP_TRY(ctx, hipMemcpyAsync(cmat_dev.p, cmat.data(), sizeof(int64_t) * cmat.size(), hipMemcpyHostToDevice, s));
            if ((rc = b3c_dist_allreduce_u64(ctx, cmat_dev.p, (int64_t)W * W))) return rc;
            { const int rb_ = b3c_readback(ctx, cmat.data(), cmat_dev.p, sizeof(int64_t) * cmat.size()); if (rb_) return rb_; }
            int64_t total = 0;
            for (int sr = 0; sr < W; sr++) total += cmat[(size_t)sr * W + D.rank];
            merged.p = (int64_t *)ctx->alloc(sizeof(int64_t) * 2 * (size_t)(total > 0 ? total : 1));
            DevBuf<int64_t> k2(ctx, total), v2(ctx, total);
            if (!merged.p || !k2.p || !v2.p) return ctx->fail(B3C_E_NOMEM, "b3c_louvain: aggregation buffers");
            if ((rc = b3c_dist_alltoallv_u64(ctx, k_in.p, merged.p, cmat.data()))) return rc;
            if ((rc = b3c_dist_alltoallv_u64(ctx, v_in.p, merged.p + total, cmat.data()))) return rc;
            if ((rc = sort_reduce(merged.p, merged.p + total, k2.p, v2.p, total, nruns.p, &nu))) return rc;
            ukeys = merged.p;
            uvals = merged.p + total;
            info->agg_gathered_runs += total;
            h.block = true;
            h.starts = cb;
        }

        h.n = kc; h.nnz = nu; h.owned = true;
        h.indptr = (int64_t *)ctx->alloc(sizeof(int64_t) * (size_t)(kc + 1));
        h.indices = (int32_t *)ctx->alloc(sizeof(int32_t) * (size_t)(nu + 8));      // (+ 8: vector loads past the last row)
        h.q = (int64_t *)ctx->alloc(sizeof(int64_t) * (size_t)(nu + 8));
        if (!h.indptr || !h.indices || !h.q) {
            free_graph(ctx, h);
            return ctx->fail(B3C_E_NOMEM, "b3c_louvain: next level");
        }
        hipLaunchKernelGGL(k_rows_from_keys, dim3((unsigned)((kc + 256) / 256)), dim3(256), 0, s, kc, nu, ukeys, h.indptr);
        if (nu)
            hipLaunchKernelGGL(k_fill_from_keys, dim3(grid_for(nu, 256, 8 * ctx->n_cu)), dim3(256), 0, s, kc, nu, ukeys,
                               uvals, h.indices, h.q);
        {
            const hipError_t e1 = hipGetLastError(), e2 = hipStreamSynchronize(s);      // the runs are read until here
            if (e1 != hipSuccess || e2 != hipSuccess) {
                free_graph(ctx, h);
                return ctx->fail(B3C_E_HIP, "b3c_louvain: aggregation failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
            }
        }
        }
        }
        free_graph(ctx, g);
        g = h;
        h = LvGraph();                                // handed over
        ids = nullptr;
    }
    free_graph(ctx, g);
    info->modularity = mod;
    info->first_sweep_launches = evs.on ? std::min<int64_t>(n_sub, info->n_move_launch) : 0;
    info->move_ms = evs.total_ms((size_t)info->first_sweep_launches, &info->first_sweep_ms);
    HIP_TRY(ctx, hipGetLastError());
    if (reorder) {
        DevBuf<int32_t> lo(ctx, n0);
        if (!lo.p) return ctx->fail(B3C_E_NOMEM, "b3c_louvain");
        hipLaunchKernelGGL(k_gather_labels, dim3((unsigned)((n0 + 255) / 256)), dim3(256), 0, s, n0, labels.p, old2new.p, lo.p);
        return b3c_copy_out(ctx, labels_out, lo.p, sizeof(int32_t) * (size_t)n0);
    }
    return b3c_copy_out(ctx, labels_out, labels.p, sizeof(int32_t) * (size_t)n0);
}
