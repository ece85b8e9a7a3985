// libb3c internals shared by the .hip translation units (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "b3c.h"

// ---------------------------------------------------------------------------------------------
// The B3C_* switches of the process environment, read ONCE -- by b3c_ctx_create, and again only when the
// caller asks (b3c_ctx_reload_env).  No entry point of the library looks at the environment itself.
// ---------------------------------------------------------------------------------------------
struct b3c_cfg {
    std::map<std::string, std::string> kv;
    void parse();                                   // api.hip: the one place that reads the environment
    const char *str(const char *k) const
    {
        auto it = kv.find(k);
        return it == kv.end() ? nullptr : it->second.c_str();
    }
    bool on(const char *k) const { return kv.find(k) != kv.end(); }
    long long num(const char *k, long long dflt) const
    {
        const char *v = str(k);
        if (!v) return dflt;
        return (v[0] == '0' && (v[1] == 'x' || v[1] == 'X')) ? strtoll(v, nullptr, 16) : strtoll(v, nullptr, 10);
    }
};

// ---------------------------------------------------------------------------------------------
// context: device, stream, caching HBM pool, error text
// ---------------------------------------------------------------------------------------------
struct b3c_ctx {
    b3c_cfg cfg;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    std::multimap<size_t, void *> pool;   // free blocks by size
    std::map<void *, size_t> live;        // every block handed out or pooled -> size
    int64_t bytes = 0;
    int n_cu = 256;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> evpool;   // timing events are recycled (creating one costs a driver call per launch timed)
    hipEvent_t ev_get()
    {
        if (!evpool.empty()) { hipEvent_t e = evpool.back(); evpool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        return hipEventCreate(&e) == hipSuccess ? e : nullptr;
    }
    void ev_put(hipEvent_t e) { if (e) evpool.push_back(e); }
    // a second stream for short kernels that are independent of the one just launched (Louvain: the few hub nodes of a
    // sub-round beside the grouped scan); forked and joined with the two events below
    hipStream_t side = nullptr, side2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join2 = nullptr;
    bool side_ok()
    {
        if (side) return true;
        if (hipStreamCreateWithFlags(&side, hipStreamNonBlocking) != hipSuccess) { side = nullptr; return false; }
        if (hipStreamCreateWithFlags(&side2, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ev_join, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ev_join2, hipEventDisableTiming) != hipSuccess) {
            (void)hipStreamDestroy(side);
            if (side2) (void)hipStreamDestroy(side2);
            side = side2 = nullptr;
            return false;
        }
        return true;
    }
    // hipFuncSetAttribute (dynamic LDS beyond 64 KB) is per DEVICE and costs ~150 us a call: every context -- one device each --
    // sets each attribute once (a process-wide flag would leave the kernels of a second GPU without it: ADVICE round 3)
    uint32_t attr_mask = 0;
    bool attr_first(int bit) { const bool f = !(attr_mask & (1u << bit)); attr_mask |= 1u << bit; return f; }
    void *pin = nullptr;      // pinned host staging for small read-backs (a pageable destination costs a staged copy)
    b3c_dist dist;            // row-sharded multi-GPU execution (b3c_ctx_set_dist); world <= 1: off
    std::vector<int64_t> dist_starts;   // the library's copy of b3c_dist.starts (dist.starts points here), empty: equal ranges
    // rows [*r0, *r1) of rank `rank` of an n-row matrix under the installed sharding
    void dist_rows(int64_t n, int rank, int64_t *r0, int64_t *r1) const
    {
        if (dist.starts) {
            *r0 = dist.starts[rank] < n ? dist.starts[rank] : n;
            *r1 = dist.starts[rank + 1] < n ? dist.starts[rank + 1] : n;
            if (rank == dist.world - 1) *r1 = n;
        } else {
            *r0 = (int64_t)rank * dist.chunk < n ? (int64_t)rank * dist.chunk : n;
            *r1 = *r0 + dist.chunk < n ? *r0 + dist.chunk : n;
        }
    }
    bool has_dist = false;
    int64_t n_exchanges = 0;  // collectives issued (b3c_dist_exchange)
    // direct transport (dist.hip, b3c_dist_window_*): a window in this rank's HBM that the peers write into (IPC-mapped), two
    // slots of [staging bytes | one sequence flag per source rank]; peer[r] = rank r's window as mapped here (peer[rank] = local)
    struct Window {
        char *local = nullptr;
        std::vector<char *> peer;
        std::vector<char> opened;       // peer[r] came from hipIpcOpenMemHandle
        int64_t slot_bytes = 0;         // staging bytes of one slot (flags follow)
        unsigned long long seq = 0;     // exchanges done through the window
        char **peer_dev = nullptr;      // the peer pointers in device memory
        unsigned int *ticket = nullptr; // last-block ticket of the push kernel
        bool open = false;
    } win;
    int64_t n_exchanges_direct = 0;

    int fail(int code, const char *fmt, ...)
    {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        err = buf;
        return code;
    }

    // Pooled allocation: HBM blocks are recycled between calls so that the timed path issues no
    // hipMalloc once warm.
    void *alloc(size_t nbytes)
    {
        if (nbytes == 0) nbytes = 16;
        nbytes = (nbytes + 255) & ~size_t(255);
        auto it = pool.lower_bound(nbytes);
        if (it != pool.end() && it->first <= nbytes + nbytes / 4 + 4096) {
            void *p = it->second;
            pool.erase(it);
            pooled[p] = false;
            return p;
        }
        void *p = nullptr;
        if (hipMalloc(&p, nbytes) != hipSuccess) {
            // release the pool and retry once
            trim();
            if (hipMalloc(&p, nbytes) != hipSuccess) return nullptr;
        }
        live[p] = nbytes;
        bytes += (int64_t)nbytes;
        return p;
    }
    std::map<void *, bool> pooled;          // blocks sitting in the pool (a second release of one must not hand it out twice)
    void release(void *p)
    {
        if (!p) return;
        auto it = live.find(p);
        if (it == live.end()) return;
        if (pooled[p]) return;             // released already
        pooled[p] = true;
        pool.insert({it->second, p});
    }
    void trim()
    {
        for (auto &kv : pool) {
            (void)hipFree(kv.second);
            bytes -= (int64_t)kv.first;
            live.erase(kv.second);
            pooled.erase(kv.second);
        }
        pool.clear();
    }
};

struct b3c_mat {
    int64_t n = 0, nnz = 0;
    int kind = B3C_F64;
    int64_t *indptr = nullptr;
    int32_t *indices = nullptr;
    void *data = nullptr;
    // locality-ordered matrices (b3c_mat_permute) remember the caller-side id of every row: Louvain
    // hashes / tie-breaks on these so that its result does not depend on the storage order
    int32_t *ids = nullptr;
    bool ordered = false;
    // values are bitwise symmetric (a_ij == a_ji): set by b3c_scale_sym_graph, kept by compress / permute;
    // lets Louvain take the edge weight straight from the entry instead of searching its mirror
    bool value_symmetric = false;
    // column indices ascend inside every row (canonical CSR).  b3c_mat_permute with B3C_PERMUTE_UNSORTED clears it;
    // kernels that look an entry up then scan the row instead of bisecting it
    bool sorted_cols = true;
    // rows [row0, row1) are the ones this matrix holds; the others are empty (row block of one rank of the sharded path).
    // row1 < 0: all rows
    int64_t row0 = 0, row1 = -1;
    // this matrix IS such a row block (b3c_mat_permute_rows with a proper sub-range; kept by site_norm / compress): a flag
    // and not a test of the bounds, because after compress a block may legitimately cover [0, n) -- all kept rows in one
    // rank's block -- while another rank's block is empty, and the ranks must agree on which code path they are on
    bool row_block = false;
    // largest stored value with fmax semantics (-inf: no entries), once a kernel or b3c_max_value has seen all of them.
    // Matrices are immutable after creation; b3c_mat_device_ptrs hands out writable pointers and therefore drops it.
    double vmax = 0.0;
    bool vmax_valid = false;
    // sum of the stored values with the diagonal counted twice and (an upper bound of) the number of diagonal entries, left behind by the kernel
    // that wrote the values (k_compress_fill_ns): an ESTIMATE (floating-point atomics, any order) from which the Louvain
    // quantisation derives its shift without a pass of its own whenever the estimate is conclusive (quantise_graph)
    double vsum2 = 0.0;
    int64_t ndiag = 0;
    bool vsum_valid = false;
    // B3C_U32 matrices: the largest off-diagonal count of every ROW (= of every column when the map is symmetric,
    // sparse_utils.py:269-281), a by-product of the pass b3c_kr_balance_counts makes over the counts anyway (kr_prep_counts);
    // b3c_max_offdiag_u32(assume_symmetric) then answers from it without another pass.  Pooled block of n uint32, or null.
    uint32_t *rowmax = nullptr;
    bool rowmax_valid = false;      // cleared when b3c_mat_device_ptrs hands out a writable data pointer
};

#define CFG (ctx->cfg)      // the switches as the context read them (every user has `ctx` in scope)

#define HIP_TRY(ctx, expr)                                                                          \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess)                                                                       \
            return (ctx)->fail(B3C_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

// a matrix under construction: freed on every early return, handed over with release()
int b3c_mat_free_internal(b3c_ctx *ctx, b3c_mat *m);
struct MatOwner {
    b3c_ctx *ctx;
    b3c_mat *m = nullptr;
    explicit MatOwner(b3c_ctx *c) : ctx(c) {}
    ~MatOwner() { if (m) b3c_mat_free_internal(ctx, m); }
    MatOwner(const MatOwner &) = delete;
    MatOwner &operator=(const MatOwner &) = delete;
    b3c_mat *release() { b3c_mat *t = m; m = nullptr; return t; }
};

// scoped pooled buffer
template <typename T>
struct DevBuf {
    b3c_ctx *ctx;
    T *p;
    DevBuf(b3c_ctx *c, size_t count) : ctx(c), p((T *)c->alloc(count * sizeof(T))) {}
    ~DevBuf() { ctx->release(p); }
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    operator T *() const { return p; }
};

static inline int grid_for(int64_t work_items, int block, int max_blocks)
{
    int64_t g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (int)g;
}

// ---------------------------------------------------------------------------------------------
// device helpers: wave64 reductions with a fixed shuffle tree (deterministic)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_min(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_down(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    return v;
}
// longest row of a CSR matrix into *out (zeroed by the caller).  The running maximum is read first: a wave whose rows
// cannot raise it skips the atomic (a stale read only costs an atomic that changes nothing), so the 16k waves of a
// million-row matrix do not queue up on one address.
__device__ __forceinline__ void max_row_len_body(int64_t n, const int64_t *__restrict__ indptr, unsigned long long *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    long long d = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const long long l = indptr[i + 1] - indptr[i];
        d = l > d ? l : d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const long long t = __shfl_down(d, o, 64);
        d = t > d ? t : d;
    }
    if ((threadIdx.x & 63) == 0 && (unsigned long long)d > __atomic_load_n(out, __ATOMIC_RELAXED)) atomicMax(out, (unsigned long long)d);
}

// Wave reductions by DPP (round 4): row_shr 1 / 2 / 4 / 8 leave every row's result in its lane 15, row_bcast:15 (rows 1, 3) and
// row_bcast:31 (rows 2, 3) carry it on to lane 63, v_readlane makes it uniform.  Six VALU instructions per 32-bit word instead of
// six ds_bpermute round trips and their address arithmetic; ALL 64 lanes must be active (wave-uniform control flow).  A lane the
// shift has no source for, or whose row the mask leaves out, reads the identity (`old` = 0).
template <int CTRL, int ROWS>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWS, 0xf, false);
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    v = max(v, dpp_u32<0x111, 0xf>(v));
    v = max(v, dpp_u32<0x112, 0xf>(v));
    v = max(v, dpp_u32<0x114, 0xf>(v));
    v = max(v, dpp_u32<0x118, 0xf>(v));
    v = max(v, dpp_u32<0x142, 0xa>(v));
    v = max(v, dpp_u32<0x143, 0xc>(v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

template <int CTRL, int ROWS>
__device__ __forceinline__ long long dpp_add_ll(long long v)
{
    const uint32_t lo = dpp_u32<CTRL, ROWS>((uint32_t)v), hi = dpp_u32<CTRL, ROWS>((uint32_t)((unsigned long long)v >> 32));
    return v + (long long)(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ long long wave_sum_ll(long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// sum over the wave, in every lane (integers: the order of the additions is immaterial); all 64 lanes active
__device__ __forceinline__ long long wave_sum_ll_all(long long v)
{
    v = dpp_add_ll<0x111, 0xf>(v);
    v = dpp_add_ll<0x112, 0xf>(v);
    v = dpp_add_ll<0x114, 0xf>(v);
    v = dpp_add_ll<0x118, 0xf>(v);
    v = dpp_add_ll<0x142, 0xa>(v);
    v = dpp_add_ll<0x143, 0xc>(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((unsigned long long)v >> 32), 63);
    return (long long)(((unsigned long long)hi << 32) | lo);
}

// A sum every wave of the grid contributes to: ONE atomic per block.  Thousands of waves that finish together queue up
// on a single address at ~11 ns per atomic (16k waves: 180 us at the end of the kernel).  Every thread of the block
// must call it; `red` is LDS for blockDim.x / 64 values.
__device__ __forceinline__ void block_atomic_add_u64(long long v, unsigned long long *__restrict__ dst, long long *red)
{
    v = wave_sum_ll(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long t = 0;
        for (unsigned w = 0; w < blockDim.x / 64; w++) t += red[w];
        if (t) atomicAdd(dst, (unsigned long long)t);
    }
    __syncthreads();
}

// internal cross-file entry points
int b3c_dist_exchange(b3c_ctx *ctx, int buf);   // in-place all-gather of an exchange buffer: RCCL or the caller's callback
int b3c_dist_allreduce_u64(b3c_ctx *ctx, void *dev, int64_t count);
int b3c_dist_allgather_rows(b3c_ctx *ctx, void *full, int esz, const int64_t *starts_host, const int64_t *starts_dev);
int b3c_dist_allgatherv(b3c_ctx *ctx, void *base, int esz, const int64_t *starts_host);   // ragged, in place
// ragged gather of (key, value) pairs of 64-bit words; *out: pooled block of 2 * *total words (keys of all ranks in rank
// order, then the values), released by the caller with ctx->release
int b3c_dist_alltoallv_u64(b3c_ctx *ctx, const int64_t *send, int64_t *recv, const int64_t *cmat);   // cmat[s * world + d]: words s -> d
int b3c_mat_alloc(b3c_ctx *ctx, int64_t n, int64_t nnz, int kind, b3c_mat **out);
int b3c_copy_in(b3c_ctx *ctx, void *dst, const void *src, size_t bytes);
int b3c_copy_out(b3c_ctx *ctx, void *dst, const void *src, size_t bytes);
int b3c_copy_out_lazy(b3c_ctx *ctx, void *dst, const void *src, size_t bytes);   // no wait when dst is device memory
// device -> host read-back of a few values the host logic waits for: through the context's pinned buffer, then sync
int b3c_readback(b3c_ctx *ctx, void *dst, const void *src, size_t bytes);
constexpr size_t B3C_PIN_BYTES = 64 * 1024;
