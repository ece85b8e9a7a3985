// COO pair stream -> symmetric CSR count matrix, on the device (gfx950).
//
// Replaces the tail of the ingest, Sparse2DAccumulator.get_coo (mzd/sparse_utils.py:246-266): duplicate (i,j)
// observations are summed, the strict upper triangle is mirrored (`_m += tril(_m.T, -1)`) and the diagonal kept,
// giving the uint32 symmetric matrix without duplicate coordinates that the cluster path starts from.  The
// reference does this with a Python dict and scipy; here it is two radix sorts and a reduce-by-key, so a contact
// stream that already sits in HBM never has to visit the host.  Integer work: bit-exact against the host twin
// (bin3c_amd/synth.py::symmetrise_to_csr) in tests/test_gpu_accumulate.py.
#include "common.hpp"

#include <hipcub/hipcub.hpp>

namespace {

__global__ void k_pair_keys(int64_t m, int64_t n, const int64_t *__restrict__ lo, const int64_t *__restrict__ hi,
                            const uint32_t *__restrict__ c, int64_t *__restrict__ keys, unsigned long long *__restrict__ vals)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < m; k += stride) {
        const int64_t a = lo[k], b = hi[k];
        const int64_t i = a < b ? a : b, j = a < b ? b : a;
        keys[k] = i * n + j;
        vals[k] = c[k];
    }
}

// every unique upper pair (i<j) -> entries (i,j) and (j,i); pairs with i == j are dropped (self-pairs come in `diag`)
__global__ void k_mirror(int64_t u, int64_t n, const int64_t *__restrict__ ukeys, const unsigned long long *__restrict__ usums,
                         int64_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < u; k += stride) {
        const int64_t key = ukeys[k];
        const int64_t i = key / n, j = key - i * n;
        const unsigned long long sv = usums[k];
        const uint32_t v = sv > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)sv;
        const bool off = i != j;
        keys[2 * k] = off ? key : INT64_MAX;            // INT64_MAX sorts to the end and is cut off
        vals[2 * k] = v;
        keys[2 * k + 1] = off ? j * n + i : INT64_MAX;
        vals[2 * k + 1] = v;
    }
}

__global__ void k_diag_entries(int64_t n, const uint32_t *__restrict__ diag, int64_t base, int64_t *__restrict__ keys,
                               uint32_t *__restrict__ vals)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const uint32_t d = diag[i];
        keys[base + i] = d ? i * n + i : INT64_MAX;
        vals[base + i] = d;
    }
}

__global__ void k_count_valid(int64_t total, const int64_t *__restrict__ keys, unsigned long long *__restrict__ cnt)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned long long c = 0;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += stride) c += keys[k] != INT64_MAX;
    c = (unsigned long long)wave_sum_ll((long long)c);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(cnt, c);
}

__global__ void k_csr_from_sorted(int64_t nnz, int64_t n, const int64_t *__restrict__ keys, int32_t *__restrict__ indices)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += stride) indices[k] = (int32_t)(keys[k] % n);
}

__global__ void k_indptr_from_sorted(int64_t n, int64_t nnz, const int64_t *__restrict__ keys, int64_t *__restrict__ indptr)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n) return;
    const int64_t target = r * n;
    int64_t lo = 0, hi = nnz;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < target) lo = mid + 1; else hi = mid;
    }
    indptr[r] = lo;
}

int sort_bits(int64_t n)
{
    int b = 1;
    while (b < 63 && ((unsigned __int128)1 << b) < (unsigned __int128)n * (unsigned __int128)n) b++;
    return b;
}

}  // namespace

extern "C" int b3c_coo_symmetrise(b3c_ctx *ctx, int64_t n, int64_t m, const int64_t *lo, const int64_t *hi,
                                  const uint32_t *counts, const uint32_t *diag, b3c_mat **out, int64_t *n_edges)
{
    if (!ctx) return B3C_E_ARG;
    if (n < 0 || m < 0 || !out || (m > 0 && (!lo || !hi || !counts)) || n >= (int64_t)INT32_MAX)
        return ctx->fail(B3C_E_ARG, "b3c_coo_symmetrise: bad argument");
    if (m >= ((int64_t)1 << 40)) return ctx->fail(B3C_E_ARG, "b3c_coo_symmetrise: more than 2^40 pairs per call");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int gm = grid_for(m, 1024, 16 * ctx->n_cu);
    const int bits = sort_bits(n > 0 ? n : 1);

    // 1. sum duplicate pairs
    int64_t u = 0;
    DevBuf<int64_t> ukeys(ctx, m);
    DevBuf<unsigned long long> usums(ctx, m);
    if (m) {
        DevBuf<int64_t> dlo(ctx, m), dhi(ctx, m), k_in(ctx, m), k_out(ctx, m), nruns(ctx, 1);
        DevBuf<uint32_t> dc(ctx, m);
        DevBuf<unsigned long long> v_in(ctx, m), v_out(ctx, m);
        if (!ukeys.p || !usums.p || !dlo.p || !dhi.p || !k_in.p || !k_out.p || !nruns.p || !dc.p || !v_in.p || !v_out.p)
            return ctx->fail(B3C_E_NOMEM, "b3c_coo_symmetrise");
        int rc;
        if ((rc = b3c_copy_in(ctx, dlo.p, lo, sizeof(int64_t) * (size_t)m)) || (rc = b3c_copy_in(ctx, dhi.p, hi, sizeof(int64_t) * (size_t)m)) ||
            (rc = b3c_copy_in(ctx, dc.p, counts, sizeof(uint32_t) * (size_t)m)))
            return rc;
        hipLaunchKernelGGL(k_pair_keys, dim3(gm), dim3(256), 0, s, m, n, dlo.p, dhi.p, dc.p, k_in.p, v_in.p);
        size_t tb = 0;
        HIP_TRY(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tb, k_in.p, k_out.p, v_in.p, v_out.p, (int64_t)m, 0, bits, s));
        {
            DevBuf<char> tmp(ctx, tb);
            if (!tmp.p) return ctx->fail(B3C_E_NOMEM, "b3c_coo_symmetrise: sort workspace");
            HIP_TRY(ctx, hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, k_in.p, k_out.p, v_in.p, v_out.p, (int64_t)m, 0, bits, s));
        }
        tb = 0;
        HIP_TRY(ctx, hipcub::DeviceReduce::ReduceByKey(nullptr, tb, k_out.p, ukeys.p, v_out.p, usums.p, nruns.p, hipcub::Sum(), (int64_t)m, s));
        {
            DevBuf<char> tmp(ctx, tb);
            if (!tmp.p) return ctx->fail(B3C_E_NOMEM, "b3c_coo_symmetrise: reduce workspace");
            HIP_TRY(ctx, hipcub::DeviceReduce::ReduceByKey(tmp.p, tb, k_out.p, ukeys.p, v_out.p, usums.p, nruns.p, hipcub::Sum(), (int64_t)m, s));
        }
        { const int rb_ = b3c_readback(ctx, &u, nruns.p, sizeof(int64_t)); if (rb_) return rb_; }
    }

    // 2. mirror + diagonal, sort by (row, col)
    const int64_t total = 2 * u + (diag ? n : 0);
    DevBuf<int64_t> e_in(ctx, total), e_out(ctx, total);
    DevBuf<uint32_t> w_in(ctx, total), w_out(ctx, total), dd(ctx, diag ? n : 1);
    DevBuf<unsigned long long> cnt(ctx, 1);
    if (!e_in.p || !e_out.p || !w_in.p || !w_out.p || !dd.p || !cnt.p) return ctx->fail(B3C_E_NOMEM, "b3c_coo_symmetrise");
    if (u) hipLaunchKernelGGL(k_mirror, dim3(grid_for(u, 1024, 16 * ctx->n_cu)), dim3(256), 0, s, u, n, ukeys.p, usums.p, e_in.p, w_in.p);
    if (diag && n) {
        int rc = b3c_copy_in(ctx, dd.p, diag, sizeof(uint32_t) * (size_t)n);
        if (rc) return rc;
        hipLaunchKernelGGL(k_diag_entries, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, dd.p, 2 * u, e_in.p, w_in.p);
    }
    int64_t nnz = 0;
    if (total) {
        size_t tb = 0;
        HIP_TRY(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tb, e_in.p, e_out.p, w_in.p, w_out.p, (int64_t)total, 0, 64, s));
        DevBuf<char> tmp(ctx, tb);
        if (!tmp.p) return ctx->fail(B3C_E_NOMEM, "b3c_coo_symmetrise: sort workspace");
        HIP_TRY(ctx, hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, e_in.p, e_out.p, w_in.p, w_out.p, (int64_t)total, 0, 64, s));
        HIP_TRY(ctx, hipMemsetAsync(cnt.p, 0, sizeof(unsigned long long), s));
        hipLaunchKernelGGL(k_count_valid, dim3(grid_for(total, 1024, 16 * ctx->n_cu)), dim3(256), 0, s, total, e_out.p, cnt.p);
        unsigned long long hc = 0;
        { const int rb_ = b3c_readback(ctx, &hc, cnt.p, sizeof(hc)); if (rb_) return rb_; }
        nnz = (int64_t)hc;
    }
    MatOwner own(ctx);            // freed on every early return below
    int rc = b3c_mat_alloc(ctx, n, nnz, B3C_U32, &own.m);
    if (rc) return rc;
    b3c_mat *mat = own.m;
    hipLaunchKernelGGL(k_indptr_from_sorted, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, s, n, nnz, e_out.p, mat->indptr);
    if (nnz) {
        hipLaunchKernelGGL(k_csr_from_sorted, dim3(grid_for(nnz, 1024, 16 * ctx->n_cu)), dim3(256), 0, s, nnz, n, e_out.p, mat->indices);
        HIP_TRY(ctx, hipMemcpyAsync(mat->data, w_out.p, sizeof(uint32_t) * (size_t)nnz, hipMemcpyDeviceToDevice, s));
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(s));
    // undirected off-diagonal edges: the mirrored entries come in pairs
    if (n_edges) {
        // entries on the diagonal = valid diag entries; off-diagonal = nnz - that
        unsigned long long hd = 0;
        if (diag && n) {
            HIP_TRY(ctx, hipMemsetAsync(cnt.p, 0, sizeof(unsigned long long), s));
            hipLaunchKernelGGL(k_count_valid, dim3(grid_for(n, 1024, 4 * ctx->n_cu)), dim3(256), 0, s, n, e_in.p + 2 * u, cnt.p);
            { const int rb_ = b3c_readback(ctx, &hd, cnt.p, sizeof(hd)); if (rb_) return rb_; }
        }
        *n_edges = (nnz - (int64_t)hd) / 2;
    }
    *out = own.release();
    return B3C_OK;
}
