// Multi-GPU exchanges of libb3c (include/b3c.h "multi-GPU"): one process per GPU, contig-row ranges of one map.
//
// Every exchange is an in-place all-gather of a buffer laid out as `world` equal slices.  With a communicator installed
// (b3c_dist.comm, created by b3c_dist_comm_create from a unique id the caller broadcasts) the library issues the RCCL
// collective itself on the context's stream -- no host round trip, no callback; the vector + scalar pair that follows
// every KCD launch is one grouped launch.  Without one, the caller-supplied callback moves the data (the gloo path of the
// CPU/one-GPU tests).  RCCL is bound at run time (dlopen): the copy already in the process (PyTorch ships one under the
// same soname) is preferred, so that a process never holds two.
#include "common.hpp"

#include <dlfcn.h>

#include <algorithm>

namespace {

// the few RCCL/NCCL 2.x entry points used; enum values from rccl.h (stable across 2.x)
typedef int (*fn_get_unique_id)(void *);
typedef int (*fn_comm_init_rank)(void **, int, unsigned char[128], int);   // ncclUniqueId is passed BY VALUE (128 bytes)
typedef int (*fn_comm_destroy)(void *);
typedef int (*fn_all_gather)(const void *, void *, size_t, int, void *, hipStream_t);
typedef int (*fn_all_reduce)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef int (*fn_broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef int (*fn_send)(const void *, size_t, int, int, void *, hipStream_t);
typedef int (*fn_recv)(void *, size_t, int, int, void *, hipStream_t);
typedef int (*fn_group)(void);
typedef const char *(*fn_err)(int);

struct IdByValue { char bytes[128]; };
typedef int (*fn_comm_init_rank_v)(void **, int, IdByValue, int);

struct Rccl {
    void *h = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank_v comm_init_rank = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_all_gather all_gather = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_broadcast broadcast = nullptr;
    fn_send send = nullptr;
    fn_recv recv = nullptr;
    fn_group group_start = nullptr, group_end = nullptr;
    fn_err err = nullptr;
    bool ok = false;
};

constexpr int NCCL_INT8 = 0, NCCL_INT32 = 2, NCCL_INT64 = 4, NCCL_UINT64 = 5, NCCL_FLOAT64 = 8;
constexpr int NCCL_SUM = 0;

Rccl *rccl()
{
    static Rccl r;
    static bool tried = false;
    if (tried) return &r;
    tried = true;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    r.h = dlopen(names[0], RTLD_NOW | RTLD_NOLOAD);                // already in the process (PyTorch)?
    for (int i = 0; !r.h && i < 3; i++) r.h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!r.h) return &r;
    r.get_unique_id = (fn_get_unique_id)dlsym(r.h, "ncclGetUniqueId");
    r.comm_init_rank = (fn_comm_init_rank_v)dlsym(r.h, "ncclCommInitRank");
    r.comm_destroy = (fn_comm_destroy)dlsym(r.h, "ncclCommDestroy");
    r.all_gather = (fn_all_gather)dlsym(r.h, "ncclAllGather");
    r.all_reduce = (fn_all_reduce)dlsym(r.h, "ncclAllReduce");
    r.broadcast = (fn_broadcast)dlsym(r.h, "ncclBroadcast");
    r.send = (fn_send)dlsym(r.h, "ncclSend");
    r.recv = (fn_recv)dlsym(r.h, "ncclRecv");
    r.group_start = (fn_group)dlsym(r.h, "ncclGroupStart");
    r.group_end = (fn_group)dlsym(r.h, "ncclGroupEnd");
    r.err = (fn_err)dlsym(r.h, "ncclGetErrorString");
    r.ok = r.get_unique_id && r.comm_init_rank && r.comm_destroy && r.all_gather && r.all_reduce && r.broadcast && r.group_start && r.group_end;
    return &r;
}

int rccl_fail(b3c_ctx *ctx, const char *what, int rc)
{
    Rccl *r = rccl();
    return ctx->fail(B3C_E_RCCL, "%s failed: %s", what, r->err ? r->err(rc) : "RCCL error");
}

struct BufDesc { void *p; size_t count; int type; };

bool describe(const b3c_dist &D, int buf, BufDesc *out)
{
    switch (buf) {
    case B3C_BUF_VEC: *out = {D.vec, (size_t)D.chunk, NCCL_FLOAT64}; return true;
    case B3C_BUF_SCAL: *out = {D.scal, 4, NCCL_FLOAT64}; return true;
    case B3C_BUF_LABELS: *out = {D.labels, (size_t)D.chunk, NCCL_INT32}; return true;
    case B3C_BUF_ISUM: *out = {D.isum, 2, NCCL_INT64}; return true;
    default: return false;
    }
}

}  // namespace

extern "C" int b3c_dist_unique_id(void *id128)
{
    Rccl *r = rccl();
    if (!id128 || !r->ok) return B3C_E_RCCL;
    return r->get_unique_id(id128) == 0 ? B3C_OK : B3C_E_RCCL;
}

extern "C" int b3c_dist_comm_create(b3c_ctx *ctx, int world, int rank, const void *id128, void **comm_out)
{
    if (!ctx) return B3C_E_ARG;
    if (!id128 || !comm_out || world < 1 || rank < 0 || rank >= world) return ctx->fail(B3C_E_ARG, "b3c_dist_comm_create: bad argument");
    Rccl *r = rccl();
    if (!r->ok) return ctx->fail(B3C_E_RCCL, "b3c_dist_comm_create: librccl.so.1 could not be loaded");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    IdByValue id;
    memcpy(id.bytes, id128, sizeof(id.bytes));
    void *comm = nullptr;
    const int rc = r->comm_init_rank(&comm, world, id, rank);
    if (rc != 0) return rccl_fail(ctx, "ncclCommInitRank", rc);
    *comm_out = comm;
    return B3C_OK;
}

extern "C" int b3c_dist_comm_destroy(b3c_ctx *ctx, void *comm)
{
    if (!ctx) return B3C_E_ARG;
    if (!comm) return B3C_OK;
    Rccl *r = rccl();
    if (!r->ok) return ctx->fail(B3C_E_RCCL, "b3c_dist_comm_destroy: librccl.so.1 could not be loaded");
    (void)hipStreamSynchronize(ctx->stream);
    const int rc = r->comm_destroy(comm);
    return rc == 0 ? B3C_OK : rccl_fail(ctx, "ncclCommDestroy", rc);
}

namespace {
// RCCL calls of a ragged in-place all-gather: rank r broadcasts its bytes [starts[r], starts[r + 1]) * esz of `base` to everybody.
// Inside a group these are ONE launch (RCCL has no all-gather with per-rank counts; a group of broadcasts is its documented
// equivalent).  The caller opens and closes the group.
int ragged_calls(b3c_ctx *ctx, Rccl *r, void *base, int esz, const int64_t *starts)
{
    const b3c_dist &D = ctx->dist;
    for (int k = 0; k < D.world; k++) {
        const int64_t len = starts[k + 1] - starts[k];
        if (len <= 0) continue;
        char *p = (char *)base + starts[k] * esz;
        const int rc = r->broadcast(p, p, (size_t)len * esz, NCCL_INT8, k, D.comm, ctx->stream);
        if (rc != 0) return rccl_fail(ctx, "ncclBroadcast", rc);
    }
    return B3C_OK;
}
}  // namespace

namespace { int win_allgather(b3c_ctx *ctx, int nitems, void *a_base, int a_esz, const int64_t *a_starts, void *b_base, int b_esz, const int64_t *b_starts); }

// Ragged in-place all-gather of a row-indexed vector: rank r owns elements [starts[r], starts[r + 1]) of `base` (esz 4 or 8).
int b3c_dist_allgatherv(b3c_ctx *ctx, void *base, int esz, const int64_t *starts)
{
    const b3c_dist &D = ctx->dist;
    if (ctx->win.open) return win_allgather(ctx, 1, base, esz, starts, nullptr, 0, nullptr);
    ctx->n_exchanges++;
    if (!D.comm) {
        if (!D.allgatherv) return ctx->fail(B3C_E_ARG, "ragged multi-GPU exchange without communicator or callback");
        if (D.allgatherv(D.user, base, esz, starts, D.world) != 0) return ctx->fail(B3C_E_RCCL, "ragged all-gather callback failed");
        return B3C_OK;
    }
    Rccl *r = rccl();
    if (!r->ok) return ctx->fail(B3C_E_RCCL, "librccl.so.1 could not be loaded");
    int rc = 0;
    if ((rc = r->group_start()) != 0) return rccl_fail(ctx, "ncclGroupStart", rc);
    if ((rc = ragged_calls(ctx, r, base, esz, starts))) { (void)r->group_end(); return rc; }   // (never leave a group open)
    if ((rc = r->group_end()) != 0) return rccl_fail(ctx, "ncclGroupEnd", rc);
    return B3C_OK;
}

// ---------------------------------------------------------------------------------------------
// Direct transport (round 6): the in-place all-gathers as PEER WRITES instead of collectives.
//
// Every rank owns a window in its HBM -- two slots of staging bytes, each followed by one 64-bit sequence flag per source rank --
// and maps the windows of all peers (hipIpcGetMemHandle / hipIpcOpenMemHandle: between GPUs of a node the stores travel over xGMI,
// between two processes on ONE GPU, the tests, they stay in its HBM).  An exchange k (slot k & 1) is two small launches on the
// context's stream:
//   push  this rank's range of every item (up to two: the scalar and the vector buffer of a KR tick) is stored into EVERY peer's
//         slot at the offset it has in the full vector; every workgroup waits for its stores (vmcnt), one thread of it releases at
//         system scope, the LAST workgroup to finish (a ticket) stores k into its flag of every peer's slot;
//   pull  every workgroup polls this rank's slot until all `world` flags show k (one lane per source rank, bounded: a trap after
//         five seconds instead of a hang), acquires, and copies the OTHER ranks' ranges from the slot into the full vector.
// No collective, no rendezvous: ~2 launches and one fabric round trip per exchange against 20-30 us of an RCCL all-gather.  Two
// slots suffice: a rank can run at most one exchange ahead of a peer -- it completes exchange k + 1 only with that peer's push of
// k + 1, which the peer issues after its own pull of k.  The pull grid is a few dozen workgroups, so ranks sharing a GPU cannot
// starve each other of compute units.  All-reduces and the all-to-all of the aggregation stay on the RCCL / callback transport.
// ---------------------------------------------------------------------------------------------
namespace {
struct WinItem { char *base; int64_t off, bytes; };     // this rank's range of a full vector: bytes [off, off + bytes) of base
constexpr int WIN_MAXW = 16;                              // ranks whose ranges travel as kernel arguments (a node has 8 GPUs)
struct WinRanges { int64_t a[WIN_MAXW + 1], b[WIN_MAXW + 1]; };

constexpr unsigned long long WIN_TIMEOUT = 500000000ull;      // wall_clock64 ticks (100 MHz): five seconds

__global__ void __launch_bounds__(256) k_win_push(int world, int me, char *const *__restrict__ peer, int64_t slot_off, int64_t flag_off,
                                                  WinItem a, WinItem b, int64_t b_stage, unsigned long long seq, unsigned int *__restrict__ ticket)
{
    // 16-byte granules where the range allows (offsets of the row-indexed buffers are multiples of 64 rows), 4-byte ones else
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int it = 0; it < 2; it++) {
        const WinItem w = it == 0 ? a : b;
        if (!w.base || w.bytes <= 0) continue;
        const int64_t stage = it == 0 ? 0 : b_stage;
        const bool wide = ((w.off | w.bytes | stage) & 15) == 0 && (((uintptr_t)w.base) & 15) == 0;
        for (int pr = 0; pr < world; pr++) {
            if (pr == me) continue;
            char *dst = peer[pr] + slot_off + stage + w.off;
            const char *src = w.base + w.off;
            if (wide) for (int64_t i = t0 * 16; i < w.bytes; i += stride * 16) *(uint4 *)(dst + i) = *(const uint4 *)(src + i);
            else for (int64_t i = t0 * 4; i < w.bytes; i += stride * 4) *(uint32_t *)(dst + i) = *(const uint32_t *)(src + i);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");          // this wave's stores have been acknowledged
    __syncthreads();
    __shared__ int last_s;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");               // ... and are visible system-wide before the ticket
        const unsigned int t = atomicAdd(ticket, 1u);
        last_s = t == gridDim.x - 1;
        if (last_s) *ticket = 0u;
    }
    __syncthreads();
    if (last_s && (int)threadIdx.x < world && (int)threadIdx.x != me)
        __hip_atomic_store((unsigned long long *)(peer[threadIdx.x] + flag_off) + me, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void __launch_bounds__(256) k_win_pull(int world, int me, char *__restrict__ local, int64_t slot_off, int64_t flag_off, int nitems,
                                                  char *a_base, int a_esz, char *b_base, int b_esz, WinRanges rg, int64_t b_stage,
                                                  unsigned long long seq)
{
    const int64_t *a_starts = rg.a, *b_starts = rg.b;
    if ((int)threadIdx.x < world && (int)threadIdx.x != me) {
        const unsigned long long *f = (const unsigned long long *)(local + flag_off) + threadIdx.x;
        const unsigned long long t0 = wall_clock64();
        unsigned spins = 0;
        while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 1023u) == 0 && wall_clock64() - t0 > WIN_TIMEOUT) __builtin_trap();      // a peer never arrived: fail, do not hang
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");                   // nothing of the slot cached from an earlier exchange
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int it = 0; it < nitems; it++) {
        char *base = it == 0 ? a_base : b_base;
        const int esz = it == 0 ? a_esz : b_esz;
        const int64_t *st = it == 0 ? a_starts : b_starts;
        const int64_t stage = it == 0 ? 0 : b_stage;
        for (int r = 0; r < world; r++) {
            if (r == me) continue;
            const int64_t off = st[r] * esz, bytes = (st[r + 1] - st[r]) * esz;
            const char *src = local + slot_off + stage + off;
            char *dst = base + off;
            const bool wide = ((off | bytes | stage) & 15) == 0 && (((uintptr_t)base) & 15) == 0;
            if (wide) for (int64_t i = t0 * 16; i < bytes; i += stride * 16) *(uint4 *)(dst + i) = *(const uint4 *)(src + i);
            else for (int64_t i = t0 * 4; i < bytes; i += stride * 4) *(uint32_t *)(dst + i) = *(const uint32_t *)(src + i);
        }
    }
}

// one exchange through the window: up to two full vectors (esz 4 | 8), rank r owning elements [starts[r], starts[r + 1]) of each
int win_allgather(b3c_ctx *ctx, int nitems, void *a_base, int a_esz, const int64_t *a_starts, void *b_base, int b_esz, const int64_t *b_starts)
{
    b3c_ctx::Window &Wn = ctx->win;
    const b3c_dist &D = ctx->dist;
    const int W = D.world, me = D.rank;
    const int64_t a_total = a_starts[W] * a_esz;
    const int64_t b_stage = (a_total + 255) & ~(int64_t)255;
    const int64_t need = b_stage + (nitems > 1 ? b_starts[W] * b_esz : 0);
    if (need > Wn.slot_bytes) return ctx->fail(B3C_E_ARG, "direct exchange of %lld bytes beyond the window's %lld", (long long)need, (long long)Wn.slot_bytes);
    Wn.seq++;
    ctx->n_exchanges++;
    ctx->n_exchanges_direct++;
    const int64_t slot = (int64_t)(Wn.seq & 1ull);
    const int64_t slot_stride = Wn.slot_bytes + 8 * (int64_t)((W + 31) & ~31);
    const int64_t slot_off = slot * slot_stride, flag_off = slot_off + Wn.slot_bytes;
    // (the ranges ride along as kernel arguments: nothing to upload, nothing to wait for)
    WinRanges rg;
    memset(&rg, 0, sizeof(rg));
    for (int r = 0; r <= W; r++) { rg.a[r] = a_starts[r]; if (nitems > 1) rg.b[r] = b_starts[r]; }
    WinItem ia{(char *)a_base, a_starts[me] * a_esz, (a_starts[me + 1] - a_starts[me]) * a_esz};
    WinItem ib{nitems > 1 ? (char *)b_base : nullptr, nitems > 1 ? b_starts[me] * b_esz : 0, nitems > 1 ? (b_starts[me + 1] - b_starts[me]) * b_esz : 0};
    const int64_t most = std::max(ia.bytes, ib.bytes);
    const int gp = (int)std::max<int64_t>(1, std::min<int64_t>(64, (most * (W - 1) / 16 + 255) / 256));
    hipLaunchKernelGGL(k_win_push, dim3(gp), dim3(256), 0, ctx->stream, W, me, (char *const *)Wn.peer_dev, slot_off, flag_off, ia, ib, b_stage, Wn.seq, Wn.ticket);
    const int64_t all = a_total + (nitems > 1 ? b_starts[W] * b_esz : 0);
    const int gq = (int)std::max<int64_t>(1, std::min<int64_t>(64, (all / 16 + 255) / 256));
    hipLaunchKernelGGL(k_win_pull, dim3(gq), dim3(256), 0, ctx->stream, W, me, Wn.local, slot_off, flag_off, nitems, (char *)a_base, a_esz,
                       (char *)b_base, b_esz, rg, b_stage, Wn.seq);
    HIP_TRY(ctx, hipGetLastError());
    return B3C_OK;
}
}  // namespace

extern "C" int b3c_dist_window_create(b3c_ctx *ctx, int64_t slot_bytes, void *handle64)
{
    if (!ctx) return B3C_E_ARG;
    if (!handle64 || slot_bytes < 256 || !ctx->has_dist && ctx->dist.world < 1) return ctx->fail(B3C_E_ARG, "b3c_dist_window_create: bad argument (install the sharding first)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    b3c_ctx::Window &Wn = ctx->win;
    if (Wn.local) return ctx->fail(B3C_E_ARG, "b3c_dist_window_create: a window exists already");
    const int W = ctx->dist.world;
    Wn.slot_bytes = (slot_bytes + 255) & ~(int64_t)255;
    const int64_t slot_stride = Wn.slot_bytes + 8 * (int64_t)((W + 31) & ~31);
    void *p = nullptr;
    HIP_TRY(ctx, hipMalloc(&p, (size_t)(2 * slot_stride)));            // (its own allocation: an IPC handle exports a whole one)
    HIP_TRY(ctx, hipMemset(p, 0, (size_t)(2 * slot_stride)));
    Wn.local = (char *)p;
    hipIpcMemHandle_t h;
    static_assert(sizeof(h) == 64, "hipIpcMemHandle_t is exchanged as 64 bytes");
    HIP_TRY(ctx, hipIpcGetMemHandle(&h, p));
    memcpy(handle64, &h, 64);
    return B3C_OK;
}

extern "C" int b3c_dist_window_open(b3c_ctx *ctx, const void *handles)
{
    if (!ctx) return B3C_E_ARG;
    b3c_ctx::Window &Wn = ctx->win;
    if (!handles || !Wn.local || Wn.open) return ctx->fail(B3C_E_ARG, "b3c_dist_window_open: bad argument");
    if (ctx->dist.world > WIN_MAXW) return ctx->fail(B3C_E_ARG, "b3c_dist_window_open: at most %d ranks", WIN_MAXW);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int W = ctx->dist.world, me = ctx->dist.rank;
    Wn.peer.assign(W, nullptr);
    Wn.opened.assign(W, 0);
    for (int r = 0; r < W; r++) {
        if (r == me) { Wn.peer[r] = Wn.local; continue; }
        hipIpcMemHandle_t h;
        memcpy(&h, (const char *)handles + 64 * (size_t)r, 64);
        void *p = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) return ctx->fail(B3C_E_HIP, "b3c_dist_window_open: hipIpcOpenMemHandle of rank %d: %s", r, hipGetErrorString(e));
        Wn.peer[r] = (char *)p;
        Wn.opened[r] = 1;
    }
    void *pd = nullptr, *tk = nullptr;
    HIP_TRY(ctx, hipMalloc(&pd, sizeof(char *) * (size_t)W));
    HIP_TRY(ctx, hipMemcpy(pd, Wn.peer.data(), sizeof(char *) * (size_t)W, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMalloc(&tk, 256));
    HIP_TRY(ctx, hipMemset(tk, 0, 256));
    Wn.peer_dev = (char **)pd;
    Wn.ticket = (unsigned int *)tk;
    Wn.seq = 0;
    Wn.open = true;
    return B3C_OK;
}

extern "C" int b3c_dist_window_close(b3c_ctx *ctx)
{
    if (!ctx) return B3C_E_ARG;
    b3c_ctx::Window &Wn = ctx->win;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (size_t r = 0; r < Wn.peer.size(); r++)
        if (Wn.opened[r] && Wn.peer[r]) (void)hipIpcCloseMemHandle(Wn.peer[r]);
    if (Wn.peer_dev) (void)hipFree(Wn.peer_dev);
    if (Wn.ticket) (void)hipFree(Wn.ticket);
    if (Wn.local) (void)hipFree(Wn.local);
    Wn = b3c_ctx::Window();
    return B3C_OK;
}

extern "C" int64_t b3c_ctx_exchanges_direct(const b3c_ctx *ctx) { return ctx ? ctx->n_exchanges_direct : 0; }

// All-gather exchange buffer `buf` in place (B3C_BUF_SCAL_VEC: the scalar and the vector buffer, one grouped launch).  The scalar
// and integer buffers are `world` equal slices; the vector and label buffers are indexed by contig row, every rank's slice being
// its row range: equal slices of `chunk`, or -- with b3c_dist.starts -- the ragged ranges the caller balanced by stored entries.
int b3c_dist_exchange(b3c_ctx *ctx, int buf)
{
    const b3c_dist &D = ctx->dist;
    const bool ragged = D.starts != nullptr;
    if (ctx->win.open) {
        // through the window: the scalar / integer buffers as `world` equal slices, the row-indexed ones by the installed row ranges
        const int W = D.world;
        std::vector<int64_t> eq(W + 1), rows(W + 1);
        auto ranges = [&](int id, void **base, int *esz, const int64_t **st) -> bool {
            BufDesc d;
            if (!describe(D, id, &d)) return false;
            *base = d.p;
            *esz = d.type == NCCL_INT32 ? 4 : 8;
            if (id == B3C_BUF_SCAL || id == B3C_BUF_ISUM) {
                for (int r = 0; r <= W; r++) eq[r] = (int64_t)r * (int64_t)d.count;
                *st = eq.data();
            } else if (ragged) {
                *st = D.starts;
            } else {
                for (int r = 0; r <= W; r++) rows[r] = (int64_t)r * D.chunk;
                *st = rows.data();
            }
            return true;
        };
        void *ba = nullptr, *bb = nullptr;
        int ea = 0, eb = 0;
        const int64_t *sa = nullptr, *sb = nullptr;
        if (buf == B3C_BUF_SCAL_VEC) {
            if (!ranges(B3C_BUF_SCAL, &ba, &ea, &sa) || !ranges(B3C_BUF_VEC, &bb, &eb, &sb)) return ctx->fail(B3C_E_ARG, "unknown exchange buffer %d", buf);
            return win_allgather(ctx, 2, ba, ea, sa, bb, eb, sb);
        }
        if (!ranges(buf, &ba, &ea, &sa)) return ctx->fail(B3C_E_ARG, "unknown exchange buffer %d", buf);
        return win_allgather(ctx, 1, ba, ea, sa, nullptr, 0, nullptr);
    }
    if (!D.comm) {
        if (ragged && buf != B3C_BUF_SCAL && buf != B3C_BUF_ISUM) {
            // callback transport with ragged ranges: the scalars by the slice callback, the row-indexed buffer by the ragged one
            int rc = B3C_OK;
            if (buf == B3C_BUF_SCAL_VEC) {
                ctx->n_exchanges++;
                if (!D.allgather || D.allgather(D.user, B3C_BUF_SCAL) != 0) return ctx->fail(B3C_E_RCCL, "all-gather callback failed (buffer %d)", B3C_BUF_SCAL);
                ctx->n_exchanges--;               // (the pair counts as ONE exchange, as on the RCCL transport)
            }
            const bool lab = buf == B3C_BUF_LABELS;
            rc = b3c_dist_allgatherv(ctx, lab ? (void *)D.labels : (void *)D.vec, lab ? 4 : 8, D.starts);
            return rc;
        }
        ctx->n_exchanges++;
        if (!D.allgather) return ctx->fail(B3C_E_ARG, "multi-GPU exchange without communicator or callback");
        if (D.allgather(D.user, buf) != 0) return ctx->fail(B3C_E_RCCL, "all-gather callback failed (buffer %d)", buf);
        return B3C_OK;
    }
    ctx->n_exchanges++;
    Rccl *r = rccl();
    if (!r->ok) return ctx->fail(B3C_E_RCCL, "librccl.so.1 could not be loaded");
    int ids[2] = {buf, -1};
    if (buf == B3C_BUF_SCAL_VEC) { ids[0] = B3C_BUF_SCAL; ids[1] = B3C_BUF_VEC; }
    int rc = 0;
    const bool grouped = ids[1] >= 0 || (ragged && buf != B3C_BUF_SCAL && buf != B3C_BUF_ISUM);
    if (grouped && (rc = r->group_start()) != 0) return rccl_fail(ctx, "ncclGroupStart", rc);
    for (int k = 0; k < 2 && ids[k] >= 0; k++) {
        BufDesc d;
        if (!describe(D, ids[k], &d)) { if (grouped) (void)r->group_end(); return ctx->fail(B3C_E_ARG, "unknown exchange buffer %d", ids[k]); }
        const size_t esz = d.type == NCCL_INT32 ? 4 : 8;
        if (ragged && (ids[k] == B3C_BUF_VEC || ids[k] == B3C_BUF_LABELS)) {
            if ((rc = ragged_calls(ctx, r, d.p, (int)esz, D.starts))) { if (grouped) (void)r->group_end(); return rc; }
            continue;
        }
        // in place: the send buffer is this rank's slice of the receive buffer (NCCL/RCCL define this case)
        rc = r->all_gather((const char *)d.p + (size_t)D.rank * d.count * esz, d.p, d.count, d.type, D.comm, ctx->stream);
        if (rc != 0) { if (grouped) (void)r->group_end(); return rccl_fail(ctx, "ncclAllGather", rc); }
    }
    if (grouped && (rc = r->group_end()) != 0) return rccl_fail(ctx, "ncclGroupEnd", rc);
    return B3C_OK;
}

// Sum `count` unsigned 64-bit integers at `dev` over all ranks, in place (exact: integer sums do not depend on the order).
int b3c_dist_allreduce_u64(b3c_ctx *ctx, void *dev, int64_t count)
{
    const b3c_dist &D = ctx->dist;
    ctx->n_exchanges++;
    if (!D.comm) {
        if (!D.allreduce) return ctx->fail(B3C_E_ARG, "multi-GPU all-reduce without communicator or callback");
        if (D.allreduce(D.user, dev, count) != 0) return ctx->fail(B3C_E_RCCL, "all-reduce callback failed");
        return B3C_OK;
    }
    Rccl *r = rccl();
    if (!r->ok) return ctx->fail(B3C_E_RCCL, "librccl.so.1 could not be loaded");
    const int rc = r->all_reduce(dev, dev, (size_t)count, NCCL_UINT64, NCCL_SUM, D.comm, ctx->stream);
    return rc == 0 ? B3C_OK : rccl_fail(ctx, "ncclAllReduce", rc);
}

namespace {
// copy the gathered slots (slot r: len[r] elements at byte offset r * slot_bytes) to their places in the full vector
__global__ void k_unpack_slots(int world, int skip, int64_t slot_bytes, int esz, const int64_t *__restrict__ starts,
                               const char *__restrict__ slots, char *__restrict__ full)
{
    for (int r = 0; r < world; r++) {
        if (r == skip) continue;
        const int64_t bytes = (starts[r + 1] - starts[r]) * esz;
        const char *src = slots + (int64_t)r * slot_bytes;
        char *dst = full + starts[r] * esz;
        // 4-byte granules (esz is 4 or 8, offsets are multiples of it)
        for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < bytes; i += (int64_t)gridDim.x * blockDim.x * 4)
            *(uint32_t *)(dst + i) = *(const uint32_t *)(src + i);
    }
}
}  // namespace

// Ragged all-gather of a row-indexed vector: rank r owns elements [starts[r], starts[r + 1]) of `full` (element size 4 or
// 8 bytes); afterwards every rank holds all of it.  Staged through the vector exchange buffer (one slice of chunk doubles per
// rank), so a rank's range must fit `chunk * 8` bytes.  `starts_dev`: the same world + 1 offsets in device memory.
int b3c_dist_allgather_rows(b3c_ctx *ctx, void *full, int esz, const int64_t *starts_host, const int64_t *starts_dev)
{
    const b3c_dist &D = ctx->dist;
    // a transport that moves ragged slices gathers the vector where it stands.  Only a session that INSTALLED ragged ranges has
    // shown that its stack can (DistSession's self-test gates set_starts: ADVICE round 4); one that fell back to equal ranges keeps
    // the staged equal-slice path below for Louvain's row-block gathers too
    if (ctx->win.open || (D.starts && (D.comm || D.allgatherv))) return b3c_dist_allgatherv(ctx, full, esz, starts_host);
    const int64_t slot_bytes = D.chunk * 8;
    const int64_t mine = (starts_host[D.rank + 1] - starts_host[D.rank]) * esz;
    for (int r = 0; r < D.world; r++)
        if ((starts_host[r + 1] - starts_host[r]) * esz > slot_bytes) return ctx->fail(B3C_E_ARG, "b3c_dist_allgather_rows: a rank's range exceeds its exchange slice");
    if (mine)
        HIP_TRY(ctx, hipMemcpyAsync((char *)D.vec + (int64_t)D.rank * slot_bytes, (const char *)full + starts_host[D.rank] * esz, (size_t)mine,
                                    hipMemcpyDeviceToDevice, ctx->stream));
    int rc = b3c_dist_exchange(ctx, B3C_BUF_VEC);
    if (rc) return rc;
    hipLaunchKernelGGL(k_unpack_slots, dim3(4 * ctx->n_cu), dim3(256), 0, ctx->stream, D.world, D.rank, slot_bytes, esz, starts_dev,
                       (const char *)D.vec, (char *)full);
    return B3C_OK;
}

// All-to-all of 64-bit words with per-pair counts (round 6: the aggregation by OWNER).  cmat[s * world + d] = words rank s sends to
// rank d -- the whole matrix, known to every rank (the caller all-reduces it first).  `send` holds this rank's words ordered by
// destination, `recv` receives the words for this rank ordered by source.  RCCL: one grouped launch of ncclSend / ncclRecv pairs
// (point to point over xGMI: every word crosses one link once).  Callback transport (the gloo tests): a sum into a zeroed buffer of
// the global layout -- every word has one non-zero contributor, so the integer all-reduce IS the exchange, bit for bit.
int b3c_dist_alltoallv_u64(b3c_ctx *ctx, const int64_t *send, int64_t *recv, const int64_t *cmat)
{
    const b3c_dist &D = ctx->dist;
    const int W = D.world, me = D.rank;
    ctx->n_exchanges++;
    std::vector<int64_t> soff(W + 1, 0), roff(W + 1, 0);
    for (int d = 0; d < W; d++) soff[d + 1] = soff[d] + cmat[(size_t)me * W + d];
    for (int sr = 0; sr < W; sr++) roff[sr + 1] = roff[sr] + cmat[(size_t)sr * W + me];
    if (D.comm) {
        Rccl *r = rccl();
        if (!r->ok || !r->send || !r->recv) return ctx->fail(B3C_E_RCCL, "librccl.so.1 lacks ncclSend / ncclRecv");
        int rc = 0;
        if ((rc = r->group_start()) != 0) return rccl_fail(ctx, "ncclGroupStart", rc);
        for (int pr = 0; pr < W; pr++) {
            const int64_t ns = soff[pr + 1] - soff[pr], nr = roff[pr + 1] - roff[pr];
            if (ns > 0 && (rc = r->send(send + soff[pr], (size_t)ns, NCCL_INT64, pr, D.comm, ctx->stream)) != 0) { (void)r->group_end(); return rccl_fail(ctx, "ncclSend", rc); }
            if (nr > 0 && (rc = r->recv(recv + roff[pr], (size_t)nr, NCCL_INT64, pr, D.comm, ctx->stream)) != 0) { (void)r->group_end(); return rccl_fail(ctx, "ncclRecv", rc); }
        }
        if ((rc = r->group_end()) != 0) return rccl_fail(ctx, "ncclGroupEnd", rc);
        return B3C_OK;
    }
    // global layout: destination-major, then source
    std::vector<int64_t> goff((size_t)W * W + 1, 0);
    for (int d = 0; d < W; d++)
        for (int sr = 0; sr < W; sr++) goff[(size_t)d * W + sr + 1] = goff[(size_t)d * W + sr] + cmat[(size_t)sr * W + d];
    const int64_t total = goff[(size_t)W * W];
    if (total == 0) return B3C_OK;
    int64_t *g = (int64_t *)ctx->alloc(sizeof(int64_t) * (size_t)total);
    if (!g) return ctx->fail(B3C_E_NOMEM, "b3c_dist_alltoallv_u64: %lld words", (long long)total);
    struct Hold { b3c_ctx *c; int64_t *p; ~Hold() { c->release(p); } } hold{ctx, g};
    HIP_TRY(ctx, hipMemsetAsync(g, 0, sizeof(int64_t) * (size_t)total, ctx->stream));
    for (int d = 0; d < W; d++) {
        const int64_t ns = soff[d + 1] - soff[d];
        if (ns > 0) HIP_TRY(ctx, hipMemcpyAsync(g + goff[(size_t)d * W + me], send + soff[d], sizeof(int64_t) * (size_t)ns, hipMemcpyDeviceToDevice, ctx->stream));
    }
    ctx->n_exchanges--;                      // (the all-reduce below counts itself: ONE exchange)
    int rc = b3c_dist_allreduce_u64(ctx, g, total);
    if (rc) return rc;
    const int64_t nr = roff[W];
    if (nr > 0) HIP_TRY(ctx, hipMemcpyAsync(recv, g + goff[(size_t)me * W], sizeof(int64_t) * (size_t)nr, hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));       // (g goes back to the pool)
    return B3C_OK;
}

// Maximum of one double over the ranks (host value in, host value out); identity when no sharding is installed.
extern "C" int b3c_dist_max_f64(b3c_ctx *ctx, double *value)
{
    if (!ctx || !value) return B3C_E_ARG;
    if (!ctx->has_dist) return B3C_OK;
    const b3c_dist &D = ctx->dist;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(D.scal + 4 * D.rank, value, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    int rc = b3c_dist_exchange(ctx, B3C_BUF_SCAL);
    if (rc) return rc;
    std::vector<double> all(4 * (size_t)D.world);
    if ((rc = b3c_readback(ctx, all.data(), D.scal, sizeof(double) * all.size()))) return rc;
    double m = *value;
    for (int r = 0; r < D.world; r++) m = all[4 * (size_t)r] > m ? all[4 * (size_t)r] : m;
    *value = m;
    return B3C_OK;
}

// All-gather of a row-indexed int32 vector of which rank r owns its row range (b3c_ctx::dist_rows): `vec` (device or
// host, n elements) holds this rank's part on entry and everything on return.  Used for the per-contig signal of the mask.
extern "C" int b3c_dist_allgather_i32(b3c_ctx *ctx, int32_t *vec, int64_t n)
{
    if (!ctx || !vec) return B3C_E_ARG;
    if (!ctx->has_dist) return B3C_OK;
    const b3c_dist &D = ctx->dist;
    if ((int64_t)D.world * D.chunk < n) return ctx->fail(B3C_E_ARG, "b3c_dist_allgather_i32: vector longer than the exchange buffers");
    if (D.starts && D.starts[D.world] != n)
        return ctx->fail(B3C_E_ARG, "b3c_dist_allgather_i32: the installed row ranges end at %lld, the vector has %lld elements", (long long)D.starts[D.world], (long long)n);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int64_t r0 = 0, r1 = 0;
    ctx->dist_rows(n, D.rank, &r0, &r1);
    if (r1 > r0)
        HIP_TRY(ctx, hipMemcpyAsync(D.labels + r0, vec + r0, sizeof(int32_t) * (size_t)(r1 - r0), hipMemcpyDefault, ctx->stream));
    int rc = b3c_dist_exchange(ctx, B3C_BUF_LABELS);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(vec, D.labels, sizeof(int32_t) * (size_t)n, hipMemcpyDefault, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return B3C_OK;
}

extern "C" int64_t b3c_ctx_exchanges(const b3c_ctx *ctx) { return ctx ? ctx->n_exchanges : 0; }
