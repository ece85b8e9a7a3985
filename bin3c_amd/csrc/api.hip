// Context and matrix management for libb3c (see include/b3c.h).
#include "common.hpp"

#include <cstdlib>

#include <cstddef>

extern "C" const char *b3c_version(void) { return "b3c 0.1 (gfx950)"; }

extern "C" int b3c_abi_layout(int which, int32_t *out, int cap)
{
#define F(T, f) (int32_t)offsetof(T, f)
    static const int32_t kr[] = {(int32_t)sizeof(b3c_kr_info), 0,
                                 F(b3c_kr_info, n_iter), F(b3c_kr_info, n_outer), F(b3c_kr_info, n_zero_diag),
                                 F(b3c_kr_info, is_symmetric), F(b3c_kr_info, hit_max_iter), F(b3c_kr_info, hit_delta),
                                 F(b3c_kr_info, hit_Delta), F(b3c_kr_info, rout), F(b3c_kr_info, spmv_ms), F(b3c_kr_info, n_spmv),
                                 F(b3c_kr_info, bitwise_symmetric), F(b3c_kr_info, value_bits), F(b3c_kr_info, tcd_tiles),
                                 F(b3c_kr_info, tcd_dict_max), F(b3c_kr_info, tcd_dict_total), F(b3c_kr_info, tcd_tail_entries),
                                 F(b3c_kr_info, tcd_xt_tails), F(b3c_kr_info, big_counts), F(b3c_kr_info, tcd_word_bytes),
                                 F(b3c_kr_info, xt_word_bytes), F(b3c_kr_info, xt_residual_words), F(b3c_kr_info, spmv_stream_bytes)};
    static const int32_t lv[] = {(int32_t)sizeof(b3c_louvain_info), 0,
                                 F(b3c_louvain_info, n_comm), F(b3c_louvain_info, n_levels), F(b3c_louvain_info, n_sweeps),
                                 F(b3c_louvain_info, n_moves), F(b3c_louvain_info, quant_shift), F(b3c_louvain_info, levels_sharded),
                                 F(b3c_louvain_info, modularity), F(b3c_louvain_info, move_ms), F(b3c_louvain_info, n_move_launch),
                                 F(b3c_louvain_info, move_nnz), F(b3c_louvain_info, move_rows), F(b3c_louvain_info, n_comm_level0),
                                 F(b3c_louvain_info, agg_gathered_runs), F(b3c_louvain_info, first_sweep_ms),
                                 F(b3c_louvain_info, first_sweep_launches), F(b3c_louvain_info, agg_nnz)};
    static const int32_t ds[] = {(int32_t)sizeof(b3c_dist), 0,
                                 F(b3c_dist, rank), F(b3c_dist, world), F(b3c_dist, chunk), F(b3c_dist, vec), F(b3c_dist, scal),
                                 F(b3c_dist, labels), F(b3c_dist, isum), F(b3c_dist, allgather), F(b3c_dist, user),
                                 F(b3c_dist, comm), F(b3c_dist, allreduce), F(b3c_dist, starts), F(b3c_dist, allgatherv)};
#undef F
    const int32_t *src = which == B3C_ABI_KR_INFO ? kr : which == B3C_ABI_LOUVAIN_INFO ? lv : which == B3C_ABI_DIST ? ds : nullptr;
    const int len = which == B3C_ABI_KR_INFO ? (int)(sizeof(kr) / 4) : which == B3C_ABI_LOUVAIN_INFO ? (int)(sizeof(lv) / 4)
                                                                                                   : (int)(sizeof(ds) / 4);
    if (!src || !out || cap < len) return B3C_E_ARG;
    for (int i = 0; i < len; i++) out[i] = src[i];
    out[1] = len - 2;
    return len;
}

extern char **environ;

// THE parser of the B3C_* switches: every entry of the process environment whose name starts with "B3C_" is copied into the
// context's table.  Nothing else in libb3c reads the environment.
void b3c_cfg::parse()
{
    kv.clear();
    for (char **e = environ; e && *e; e++) {
        if (strncmp(*e, "B3C_", 4) != 0) continue;
        const char *eq = strchr(*e, '=');
        if (!eq) continue;
        kv[std::string(*e, (size_t)(eq - *e))] = std::string(eq + 1);
    }
}

extern "C" int b3c_ctx_reload_env(b3c_ctx *ctx)
{
    if (!ctx) return B3C_E_ARG;
    ctx->cfg.parse();
    return B3C_OK;
}

extern "C" int b3c_ctx_create(int device, void *stream, b3c_ctx **out)
{
    if (!out) return B3C_E_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return B3C_E_HIP;
    if (device < 0 || device >= count) return B3C_E_ARG;
    if (hipSetDevice(device) != hipSuccess) return B3C_E_HIP;
    b3c_ctx *ctx = new b3c_ctx();
    ctx->cfg.parse();
    ctx->device = device;
    if (stream) {
        ctx->stream = (hipStream_t)stream;
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
            delete ctx;
            return B3C_E_HIP;
        }
        ctx->own_stream = true;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->n_cu = prop.multiProcessorCount;
    (void)hipEventCreate(&ctx->ev0);
    (void)hipEventCreate(&ctx->ev1);
    if (hipHostMalloc(&ctx->pin, B3C_PIN_BYTES, hipHostMallocDefault) != hipSuccess) ctx->pin = nullptr;
    *out = ctx;
    return B3C_OK;
}

extern "C" int b3c_ctx_destroy(b3c_ctx *ctx)
{
    if (!ctx) return B3C_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->trim();
    for (auto &kv : ctx->live) (void)hipFree(kv.first);
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    for (hipEvent_t e : ctx->evpool) (void)hipEventDestroy(e);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->side) {
        (void)hipStreamSynchronize(ctx->side);
        (void)hipStreamSynchronize(ctx->side2);
        (void)hipEventDestroy(ctx->ev_fork);
        (void)hipEventDestroy(ctx->ev_join);
        (void)hipEventDestroy(ctx->ev_join2);
        (void)hipStreamDestroy(ctx->side);
        (void)hipStreamDestroy(ctx->side2);
    }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return B3C_OK;
}

extern "C" const char *b3c_last_error(const b3c_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

extern "C" int b3c_ctx_sync(b3c_ctx *ctx)
{
    if (!ctx) return B3C_E_ARG;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return B3C_OK;
}

extern "C" int b3c_ctx_set_dist(b3c_ctx *ctx, const b3c_dist *dist)
{
    if (!ctx) return B3C_E_ARG;
    if (!dist) { ctx->has_dist = false; return B3C_OK; }
    if (dist->world < 1 || dist->rank < 0 || dist->rank >= dist->world || dist->chunk < 1 || (!dist->allgather && !dist->comm) ||
        !dist->vec || !dist->scal || !dist->labels || !dist->isum)
        return ctx->fail(B3C_E_ARG, "b3c_ctx_set_dist: bad argument");
    ctx->dist = *dist;
    ctx->dist_starts.clear();
    if (dist->starts) {
        // the caller's array is copied (the library never retains caller memory) and checked: ascending, from 0, inside the buffers
        ctx->dist_starts.assign(dist->starts, dist->starts + dist->world + 1);
        bool ok = ctx->dist_starts[0] == 0 && ctx->dist_starts[dist->world] <= (int64_t)dist->world * dist->chunk;
        for (int r = 0; r < dist->world; r++) ok = ok && ctx->dist_starts[r] <= ctx->dist_starts[r + 1];
        if (!ok || (!dist->comm && !dist->allgatherv)) {
            ctx->dist_starts.clear();
            ctx->dist.starts = nullptr;
            ctx->has_dist = false;
            return ctx->fail(B3C_E_ARG, "b3c_ctx_set_dist: bad row ranges");
        }
        ctx->dist.starts = ctx->dist_starts.data();
    }
    // B3C_DIST_FORCE: run the sharded code path (callbacks, exchange buffers) with a single rank too -- a test hook that
    // lets one GPU exercise the RCCL plumbing of the caller
    ctx->has_dist = dist->world > 1 || (dist->world == 1 && CFG.on("B3C_DIST_FORCE"));
    return B3C_OK;
}

extern "C" int64_t b3c_ctx_bytes(const b3c_ctx *ctx) { return ctx ? ctx->bytes : 0; }

extern "C" int b3c_ctx_trim(b3c_ctx *ctx)
{
    if (!ctx) return B3C_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->trim();
    return B3C_OK;
}

int b3c_copy_in(b3c_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    if (bytes == 0) return B3C_OK;
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, ctx->stream));
    return B3C_OK;
}

int b3c_copy_out(b3c_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    if (bytes == 0) return B3C_OK;
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return B3C_OK;
}

// the same without the wait when `dst` is device memory: the copy is ordered on the context's stream, which is all a caller
// that keeps working on that stream needs (small helpers called several times per step)
int b3c_copy_out_lazy(b3c_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    if (bytes == 0) return B3C_OK;
    hipPointerAttribute_t at;
    const bool dev = hipPointerGetAttributes(&at, dst) == hipSuccess && at.type == hipMemoryTypeDevice;
    if (!dev) (void)hipGetLastError();           // an unregistered host pointer reports an error: not ours
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, ctx->stream));
    if (!dev) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return B3C_OK;
}

int b3c_readback(b3c_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    if (bytes == 0) return B3C_OK;
    if (bytes > B3C_PIN_BYTES || !ctx->pin) {
        HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return B3C_OK;
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->pin, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(dst, ctx->pin, bytes);
    return B3C_OK;
}

int b3c_mat_alloc(b3c_ctx *ctx, int64_t n, int64_t nnz, int kind, b3c_mat **out)
{
    b3c_mat *m = new b3c_mat();
    m->n = n;
    m->nnz = nnz;
    m->kind = kind;
    size_t esz = kind == B3C_U32 ? 4 : 8;
    m->indptr = (int64_t *)ctx->alloc(sizeof(int64_t) * (size_t)(n + 1));
    // 32 bytes of slack: kernels that read the entry streams as 4-element vectors may touch up to 3 entries past nnz
    m->indices = (int32_t *)ctx->alloc(sizeof(int32_t) * (size_t)nnz + 32);
    m->data = ctx->alloc(esz * (size_t)nnz + 32);
    if (!m->indptr || !m->indices || !m->data) {
        ctx->release(m->indptr);
        ctx->release(m->indices);
        ctx->release(m->data);
        delete m;
        return ctx->fail(B3C_E_NOMEM, "out of device memory allocating %lld x %lld matrix", (long long)n, (long long)nnz);
    }
    *out = m;
    return B3C_OK;
}

extern "C" int b3c_mat_upload(b3c_ctx *ctx, int64_t n, int64_t nnz, const int64_t *indptr, const int32_t *indices,
                              const void *data, int kind, b3c_mat **out)
{
    if (!ctx) return B3C_E_ARG;
    if (!out || n < 0 || nnz < 0 || !indptr || (nnz > 0 && (!indices || !data)) || (kind != B3C_F64 && kind != B3C_U32))
        return ctx->fail(B3C_E_ARG, "b3c_mat_upload: bad argument");
    if (n >= (int64_t)INT32_MAX) return ctx->fail(B3C_E_ARG, "b3c_mat_upload: n must fit int32 column ids");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    MatOwner own(ctx);
    int rc = b3c_mat_alloc(ctx, n, nnz, kind, &own.m);
    if (rc) return rc;
    b3c_mat *m = own.m;
    size_t esz = kind == B3C_U32 ? 4 : 8;
    if ((rc = b3c_copy_in(ctx, m->indptr, indptr, sizeof(int64_t) * (size_t)(n + 1))) ||
        (rc = b3c_copy_in(ctx, m->indices, indices, sizeof(int32_t) * (size_t)nnz)) ||
        (rc = b3c_copy_in(ctx, m->data, data, esz * (size_t)nnz)))
        return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *out = own.release();
    return B3C_OK;
}

int b3c_mat_free_internal(b3c_ctx *ctx, b3c_mat *m) { return b3c_mat_free(ctx, m); }

extern "C" int b3c_mat_free(b3c_ctx *ctx, b3c_mat *m)
{
    if (!ctx || !m) return B3C_OK;
    ctx->release(m->indptr);
    ctx->release(m->indices);
    ctx->release(m->data);
    ctx->release(m->ids);
    ctx->release(m->rowmax);
    delete m;
    return B3C_OK;
}

extern "C" int b3c_mat_ids(b3c_ctx *ctx, const b3c_mat *m, int32_t *out)
{
    if (!ctx || !m || !out) return B3C_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!m->ids) return 1;   // identity: nothing written
    return b3c_copy_out(ctx, out, m->ids, sizeof(int32_t) * (size_t)m->n);
}

extern "C" int b3c_mat_info(const b3c_mat *m, int64_t *n, int64_t *nnz, int *kind)
{
    if (!m) return B3C_E_ARG;
    if (n) *n = m->n;
    if (nnz) *nnz = m->nnz;
    if (kind) *kind = m->kind;
    return B3C_OK;
}

extern "C" int b3c_mat_download(b3c_ctx *ctx, const b3c_mat *m, int64_t *indptr, int32_t *indices, void *data)
{
    if (!ctx || !m) return B3C_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    size_t esz = m->kind == B3C_U32 ? 4 : 8;
    if (indptr) HIP_TRY(ctx, hipMemcpyAsync(indptr, m->indptr, sizeof(int64_t) * (size_t)(m->n + 1), hipMemcpyDefault, ctx->stream));
    if (indices && m->nnz) HIP_TRY(ctx, hipMemcpyAsync(indices, m->indices, sizeof(int32_t) * (size_t)m->nnz, hipMemcpyDefault, ctx->stream));
    if (data && m->nnz) HIP_TRY(ctx, hipMemcpyAsync(data, m->data, esz * (size_t)m->nnz, hipMemcpyDefault, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return B3C_OK;
}

extern "C" int b3c_mat_device_ptrs(const b3c_mat *m, void **indptr, void **indices, void **data)
{
    if (!m) return B3C_E_ARG;
    if (indptr) *indptr = m->indptr;
    if (indices) *indices = m->indices;
    if (data) {
        *data = m->data;
        const_cast<b3c_mat *>(m)->vmax_valid = false;      // the caller may write through this pointer
        const_cast<b3c_mat *>(m)->rowmax_valid = false;
        const_cast<b3c_mat *>(m)->vsum_valid = false;
    }
    return B3C_OK;
}
