// CSR helper kernels around the KR iteration (gfx950): site normalisation, symmetric scaling,
// off-diagonal maximum, mask compaction, global maximum.  All are single streaming passes over the
// stored entries (HBM-bound integer/fp64 work; no MFMA).
#include "common.hpp"

#include <hipcub/hipcub.hpp>

namespace {

constexpr int BLK = 256;

// Iterate the stored entries of a CSR matrix with LPR lanes per row; f(row, k, lane_in_row).
template <int LPR, typename F>
__device__ __forceinline__ void for_rows(int64_t n, const int64_t *__restrict__ indptr, F f)
{
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR, sl = lane % LPR;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    const int64_t nwave = (int64_t)gridDim.x * (blockDim.x / 64);
    for (int64_t row = wave * RPW + sub; row < n; row += nwave * RPW) {
        const int64_t s = indptr[row], e = indptr[row + 1];
        for (int64_t k = s + sl; k < e; k += LPR) f(row, k);
    }
}

// fast_norm_fullseq_bysite (contact_map.py:100-113) on astype(float) counts (:929); sites 0 -> 1 (:1107)
template <int LPR>
__global__ void __launch_bounds__(BLK) k_site_norm(int64_t n, const int64_t *__restrict__ indptr,
                                                    const int32_t *__restrict__ indices,
                                                    const uint32_t *__restrict__ counts,
                                                    const double *__restrict__ sites, double *__restrict__ out)
{
    for_rows<LPR>(n, indptr, [&](int64_t row, int64_t k) {
        double si = sites[row];
        double sj = sites[indices[k]];
        si = si == 0.0 ? 1.0 : si;
        sj = sj == 0.0 ? 1.0 : sj;
        out[k] = (double)counts[k] * (1.0 / (si * sj));
    });
}

// X.T.dot(_orig.dot(X)) (sparse_utils.py:223-224): x_i * (a_ij * x_j)
template <int LPR>
__global__ void __launch_bounds__(BLK) k_scale_sym(int64_t n, const int64_t *__restrict__ indptr,
                                                    const int32_t *__restrict__ indices, double *__restrict__ a,
                                                    const double *__restrict__ x)
{
    for_rows<LPR>(n, indptr, [&](int64_t row, int64_t k) {
        double t = a[k] * x[indices[k]];
        a[k] = x[row] * t;
    });
}

// Graph form of the same scaling for a bitwise-symmetric A: both triangles get the value the LOWER
// triangle has in scipy's result, x_hi * (a * x_lo) with hi/lo by caller-side id -- the value that
// nx.Graph ends up holding for the edge (cluster.py:319-321: the later add_edge(v,u) overwrites (u,v)).
template <int LPR>
__global__ void __launch_bounds__(BLK) k_scale_sym_graph(int64_t n, const int64_t *__restrict__ indptr,
                                                          const int32_t *__restrict__ indices, double *__restrict__ a,
                                                          const double *__restrict__ x, const int32_t *__restrict__ ids)
{
    for_rows<LPR>(n, indptr, [&](int64_t row, int64_t k) {
        const int32_t j = indices[k];
        const int32_t idr = ids ? ids[row] : (int32_t)row, idj = ids ? ids[j] : j;
        const double xr = x[row], xj = x[j];
        const double xhi = idr >= idj ? xr : xj, xlo = idr >= idj ? xj : xr;
        a[k] = xhi * (a[k] * xlo);
    });
}

// site normalisation (contact_map.py:100-113, 1103-1108) fused with the final scaling of KR (sparse_utils.py:223-224):
// a = counts * (1 / (s_i * s_j)) exactly as the reference forms it, then x_i*(a*x_j) or, in graph form, x_hi*(a*x_lo)
template <int LPR>
__global__ void __launch_bounds__(BLK) k_norm_scale(int64_t n, const int64_t *__restrict__ indptr,
                                                     const int32_t *__restrict__ indices,
                                                     const uint32_t *__restrict__ counts,
                                                     const double *__restrict__ sites, const double *__restrict__ x,
                                                     const int32_t *__restrict__ ids, int graph_form,
                                                     double *__restrict__ out)
{
    for_rows<LPR>(n, indptr, [&](int64_t row, int64_t k) {
        const int32_t j = indices[k];
        double si = sites[row], sj = sites[j];
        si = si == 0.0 ? 1.0 : si;
        sj = sj == 0.0 ? 1.0 : sj;
        const double a = (double)counts[k] * (1.0 / (si * sj));
        const double xr = x[row], xj = x[j];
        if (graph_form) {
            const int32_t idr = ids ? ids[row] : (int32_t)row, idj = ids ? ids[j] : j;
            const double xhi = idr >= idj ? xr : xj, xlo = idr >= idj ? xj : xr;
            out[k] = xhi * (a * xlo);
        } else {
            out[k] = xr * (a * xj);
        }
    });
}

// max_offdiag (sparse_utils.py:269-281), symmetric input: row maximum with the diagonal skipped
template <int LPR>
__global__ void __launch_bounds__(BLK) k_max_offdiag_rows(int64_t n, const int64_t *__restrict__ indptr,
                                                           const int32_t *__restrict__ indices,
                                                           const uint32_t *__restrict__ a, uint32_t *__restrict__ out)
{
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR, sl = lane % LPR;
    const int64_t wave = (int64_t)blockIdx.x * (BLK / 64) + (threadIdx.x >> 6);
    const int64_t nwave = (int64_t)gridDim.x * (BLK / 64);
    for (int64_t row = wave * RPW + sub; row < n; row += nwave * RPW) {
        const int64_t s = indptr[row], e = indptr[row + 1];
        uint32_t m = 0;
        for (int64_t k = s + sl; k < e; k += LPR) {
            uint32_t val = a[k];
            if (indices[k] != (int32_t)row && val > m) m = val;
        }
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) {
            uint32_t other = (uint32_t)__shfl_down((int)m, o, LPR);
            m = other > m ? other : m;
        }
        if (sl == 0) out[row] = m;
    }
}

// general input: column maximum by integer atomics (exact, order independent)
template <int LPR>
__global__ void __launch_bounds__(BLK) k_max_offdiag_cols(int64_t n, const int64_t *__restrict__ indptr,
                                                           const int32_t *__restrict__ indices,
                                                           const uint32_t *__restrict__ a, uint32_t *__restrict__ out)
{
    for_rows<LPR>(n, indptr, [&](int64_t row, int64_t k) {
        const int32_t j = indices[k];
        const uint32_t val = a[k];
        if (j != (int32_t)row && val != 0) atomicMax(&out[j], val);
    });
}

// compress (sparse_utils.py:284-314), pass 1: kept entries per kept row
template <int LPR>
__global__ void __launch_bounds__(BLK) k_compress_count(int64_t n, const int64_t *__restrict__ indptr,
                                                         const int32_t *__restrict__ indices,
                                                         const uint8_t *__restrict__ mask,
                                                         const int64_t *__restrict__ newid, int64_t *__restrict__ cnt)
{
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR, sl = lane % LPR;
    const int64_t wave = (int64_t)blockIdx.x * (BLK / 64) + (threadIdx.x >> 6);
    const int64_t nwave = (int64_t)gridDim.x * (BLK / 64);
    for (int64_t row = wave * RPW + sub; row < n; row += nwave * RPW) {
        if (!mask[row]) continue;
        const int64_t s = indptr[row], e = indptr[row + 1];
        int c = 0;
        for (int64_t k = s + sl; k < e; k += LPR) c += mask[indices[k]] != 0;
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) c += __shfl_down(c, o, LPR);
        if (sl == 0) cnt[newid[row]] = c;
    }
}

// pass 1 on a STRUCTURALLY SYMMETRIC whole matrix with few rows dropped (round 6): kept entries of kept row i = its length minus its
// dropped neighbours -- and those are found from the DROPPED rows' side: row j dropped, (j, i) stored <=> (i, j) stored.  A metagenome
// map drops a few per cent of its contigs (C3: 19 k of 1 M), so this walks 4 M entries instead of gathering a mask byte for each of
// the 201 M (k_compress_count: 1.0 ms of the 4.7 ms sub-map stage).  Integer atomics: the counts are exact in any order.
__global__ void k_compress_cnt_init(int64_t n, const int64_t *__restrict__ indptr, const uint8_t *__restrict__ mask,
                                    const int64_t *__restrict__ newid, int64_t *__restrict__ cnt)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && mask[i]) cnt[newid[i]] = indptr[i + 1] - indptr[i];
}

__global__ void __launch_bounds__(BLK) k_compress_cnt_sub(int64_t n, const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                           const uint8_t *__restrict__ mask, const int64_t *__restrict__ newid,
                                                           int64_t *__restrict__ cnt)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (BLK / 64) + (threadIdx.x >> 6);
    const int64_t nwave = (int64_t)gridDim.x * (BLK / 64);
    // 64 rows per wave and pass: a lane looks at one row's flag, the dropped ones are then walked by the whole wave
    for (int64_t base = wave * 64; base < n; base += nwave * 64) {
        const int64_t r = base + lane;
        unsigned long long dropped = __ballot(r < n && !mask[r]);
        while (dropped) {
            const int b = __ffsll((unsigned long long)dropped) - 1;
            dropped &= dropped - 1;
            const int64_t row = base + b;
            const int64_t s0 = indptr[row], e0 = indptr[row + 1];
            for (int64_t k = s0 + lane; k < e0; k += 64) {
                const int32_t j = indices[k];
                if (mask[j]) atomicAdd((unsigned long long *)&cnt[newid[j]], ~0ull);        // - 1
            }
        }
    }
}

// pass 2: order-preserving fill, one wave per kept row (ballot compaction keeps column order)
template <typename T>
__global__ void __launch_bounds__(BLK) k_compress_fill(int64_t n, const int64_t *__restrict__ indptr,
                                                        const int32_t *__restrict__ indices, const T *__restrict__ a,
                                                        const uint8_t *__restrict__ mask,
                                                        const int64_t *__restrict__ newid,
                                                        const int64_t *__restrict__ out_indptr,
                                                        int32_t *__restrict__ out_indices, T *__restrict__ out_a)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (BLK / 64) + (threadIdx.x >> 6);
    const int64_t nwave = (int64_t)gridDim.x * (BLK / 64);
    for (int64_t row = wave; row < n; row += nwave) {
        if (!mask[row]) continue;
        const int64_t s = indptr[row], e = indptr[row + 1];
        int64_t o = out_indptr[newid[row]];
        for (int64_t base = s; base < e; base += 64) {
            const int64_t k = base + lane;
            int32_t j = 0;
            bool keep = false;
            if (k < e) {
                j = indices[k];
                keep = mask[j] != 0;
            }
            const unsigned long long b = __ballot(keep);
            if (keep) {
                const int pos = __popcll(b & ((1ull << lane) - 1ull));
                out_indices[o + pos] = (int32_t)newid[j];
                out_a[o + pos] = a[k];
            }
            o += __popcll(b);
        }
    }
}

// ids of a compressed, locality-ordered matrix: rank of the caller-side id among the kept caller-side ids
__global__ void k_scatter_keep(int64_t n, const uint8_t *__restrict__ mask, const int32_t *__restrict__ ids,
                               int64_t *__restrict__ keep_orig)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keep_orig[ids[i]] = mask[i] != 0;
}

__global__ void k_sub_ids(int64_t n, const uint8_t *__restrict__ mask, const int32_t *__restrict__ ids,
                          const int64_t *__restrict__ newid, const int64_t *__restrict__ rank_orig,
                          int32_t *__restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && mask[i]) out[newid[i]] = (int32_t)rank_orig[ids[i]];
}

__global__ void k_mask_to_i64(int64_t n, const uint8_t *__restrict__ mask, int64_t *__restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = mask[i] != 0;
}

// _map.max() (cluster.py:316): per-block partial maxima
__global__ void __launch_bounds__(BLK) k_max_value(int64_t nnz, const double *__restrict__ a, double *__restrict__ part)
{
    __shared__ double lds[BLK / 64];
    double m = -INFINITY;
    const int64_t stride = (int64_t)gridDim.x * BLK;
    for (int64_t k = (int64_t)blockIdx.x * BLK + threadIdx.x; k < nnz; k += stride) m = fmax(m, a[k]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = lds[0];
        for (int w = 1; w < BLK / 64; w++) r = fmax(r, lds[w]);
        part[blockIdx.x] = r;
    }
}

int pick_lpr(const b3c_mat *m)
{
    const double avg = m->n ? (double)m->nnz / (double)m->n : 0.0;
    return avg >= 96 ? 64 : avg >= 40 ? 32 : avg >= 18 ? 16 : avg >= 8 ? 8 : 4;
}

#define DISPATCH_LPR(lpr, KERNEL, grid, stream, ...)                                                  \
    switch (lpr) {                                                                                    \
    case 64: hipLaunchKernelGGL(KERNEL<64>, dim3(grid), dim3(BLK), 0, stream, __VA_ARGS__); break;    \
    case 32: hipLaunchKernelGGL(KERNEL<32>, dim3(grid), dim3(BLK), 0, stream, __VA_ARGS__); break;    \
    case 16: hipLaunchKernelGGL(KERNEL<16>, dim3(grid), dim3(BLK), 0, stream, __VA_ARGS__); break;    \
    case 8: hipLaunchKernelGGL(KERNEL<8>, dim3(grid), dim3(BLK), 0, stream, __VA_ARGS__); break;      \
    default: hipLaunchKernelGGL(KERNEL<4>, dim3(grid), dim3(BLK), 0, stream, __VA_ARGS__); break;     \
    }

int row_grid(const b3c_ctx *ctx, int64_t n, int lpr)
{
    return grid_for(n, (BLK / 64) * (64 / lpr), 16 * ctx->n_cu);
}

}  // namespace

// exclusive prefix sum of int64 (device), used by compress and by the Louvain aggregation
int b3c_exclusive_scan_i64(b3c_ctx *ctx, const int64_t *in, int64_t *out, int64_t count)
{
    size_t tmp_bytes = 0;
    HIP_TRY(ctx, hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, in, out, (int64_t)count, ctx->stream));
    DevBuf<char> tmp(ctx, tmp_bytes);
    if (!tmp.p) return ctx->fail(B3C_E_NOMEM, "scan workspace");
    HIP_TRY(ctx, hipcub::DeviceScan::ExclusiveSum(tmp.p, tmp_bytes, in, out, (int64_t)count, ctx->stream));
    return B3C_OK;
}

extern "C" int b3c_max_offdiag_u32(b3c_ctx *ctx, const b3c_mat *counts, int assume_symmetric, uint32_t *out)
{
    if (!ctx) return B3C_E_ARG;
    if (!counts || counts->kind != B3C_U32 || !out) return ctx->fail(B3C_E_ARG, "b3c_max_offdiag_u32: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t n = counts->n;
    if (n == 0) return B3C_OK;
    // left behind by the balancing's pass over the counts (kr_prep_counts): no second pass
    if (assume_symmetric && counts->rowmax && counts->rowmax_valid) return b3c_copy_out(ctx, out, counts->rowmax, sizeof(uint32_t) * (size_t)n);
    DevBuf<uint32_t> d(ctx, n);
    if (!d.p) return ctx->fail(B3C_E_NOMEM, "b3c_max_offdiag_u32");
    const int lpr = pick_lpr(counts);
    const int g = row_grid(ctx, n, lpr);
    if (assume_symmetric) {
        DISPATCH_LPR(lpr, k_max_offdiag_rows, g, ctx->stream, n, counts->indptr, counts->indices,
                     (const uint32_t *)counts->data, d.p);
    } else {
        HIP_TRY(ctx, hipMemsetAsync(d.p, 0, sizeof(uint32_t) * (size_t)n, ctx->stream));
        DISPATCH_LPR(lpr, k_max_offdiag_cols, g, ctx->stream, n, counts->indptr, counts->indices,
                     (const uint32_t *)counts->data, d.p);
    }
    HIP_TRY(ctx, hipGetLastError());
    return b3c_copy_out(ctx, out, d.p, sizeof(uint32_t) * (size_t)n);
}

namespace {
__global__ void k_accept_mask(int64_t n, const int64_t *__restrict__ len, const uint32_t *__restrict__ sig, int64_t min_len, int64_t min_sig,
                              uint8_t *__restrict__ mask)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) mask[i] = (len[i] >= min_len && (int64_t)sig[i] >= min_sig) ? 1 : 0;
}

template <typename T>
__global__ void k_vec_permute(int64_t n, const T *__restrict__ src, const int32_t *__restrict__ idx, T *__restrict__ dst, int scatter)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (scatter) dst[idx[i]] = src[i];
    else dst[i] = src[idx[i]];
}
}  // namespace

extern "C" int b3c_acceptance_mask(b3c_ctx *ctx, int64_t n, const int64_t *lengths, const uint32_t *signal, int64_t min_len,
                                   int64_t min_sig, uint8_t *mask)
{
    if (!ctx) return B3C_E_ARG;
    if (n < 0 || (n > 0 && (!lengths || !signal || !mask))) return ctx->fail(B3C_E_ARG, "b3c_acceptance_mask: bad argument");
    if (n == 0) return B3C_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf<int64_t> dl(ctx, n);
    DevBuf<uint32_t> ds(ctx, n);
    DevBuf<uint8_t> dm(ctx, n);
    if (!dl.p || !ds.p || !dm.p) return ctx->fail(B3C_E_NOMEM, "b3c_acceptance_mask");
    int rc;
    if ((rc = b3c_copy_in(ctx, dl.p, lengths, sizeof(int64_t) * (size_t)n)) || (rc = b3c_copy_in(ctx, ds.p, signal, sizeof(uint32_t) * (size_t)n)))
        return rc;
    hipLaunchKernelGGL(k_accept_mask, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n, dl.p, ds.p, min_len, min_sig, dm.p);
    HIP_TRY(ctx, hipGetLastError());
    return b3c_copy_out_lazy(ctx, mask, dm.p, (size_t)n);
}

extern "C" int b3c_vec_permute(b3c_ctx *ctx, int64_t n, int elem_bytes, const void *src, const int32_t *idx, void *dst, int scatter)
{
    if (!ctx) return B3C_E_ARG;
    if (n < 0 || (elem_bytes != 1 && elem_bytes != 4 && elem_bytes != 8) || (n > 0 && (!src || !idx || !dst)))
        return ctx->fail(B3C_E_ARG, "b3c_vec_permute: bad argument");
    if (n == 0) return B3C_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf<char> ds(ctx, (size_t)n * elem_bytes), dd(ctx, (size_t)n * elem_bytes);
    DevBuf<int32_t> di(ctx, n);
    if (!ds.p || !dd.p || !di.p) return ctx->fail(B3C_E_NOMEM, "b3c_vec_permute");
    int rc;
    if ((rc = b3c_copy_in(ctx, ds.p, src, (size_t)n * elem_bytes)) || (rc = b3c_copy_in(ctx, di.p, idx, sizeof(int32_t) * (size_t)n))) return rc;
    const unsigned g = (unsigned)((n + 255) / 256);
    if (elem_bytes == 8) hipLaunchKernelGGL(k_vec_permute<uint64_t>, dim3(g), dim3(256), 0, ctx->stream, n, (const uint64_t *)ds.p, di.p, (uint64_t *)dd.p, scatter);
    else if (elem_bytes == 4) hipLaunchKernelGGL(k_vec_permute<uint32_t>, dim3(g), dim3(256), 0, ctx->stream, n, (const uint32_t *)ds.p, di.p, (uint32_t *)dd.p, scatter);
    else hipLaunchKernelGGL(k_vec_permute<uint8_t>, dim3(g), dim3(256), 0, ctx->stream, n, (const uint8_t *)ds.p, di.p, (uint8_t *)dd.p, scatter);
    HIP_TRY(ctx, hipGetLastError());
    return b3c_copy_out_lazy(ctx, dst, dd.p, (size_t)n * elem_bytes);
}

namespace {
__global__ void k_row_nnz(int64_t n, const int64_t *__restrict__ indptr, const int32_t *__restrict__ idx, int64_t *__restrict__ out,
                          int *__restrict__ bad)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int64_t r = idx ? idx[k] : k;
    if ((uint64_t)r >= (uint64_t)n) {        // a caller-supplied row id outside the matrix: reported, never dereferenced
        out[k] = 0;
        *bad = 1;
        return;
    }
    out[k] = indptr[r + 1] - indptr[r];
}
}  // namespace

extern "C" int b3c_mat_row_nnz(b3c_ctx *ctx, const b3c_mat *m, const int32_t *idx, int64_t *out)
{
    if (!ctx) return B3C_E_ARG;
    if (!m || !out) return ctx->fail(B3C_E_ARG, "b3c_mat_row_nnz: bad argument");
    const int64_t n = m->n;
    if (n == 0) return B3C_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf<int64_t> dd(ctx, n);
    DevBuf<int32_t> di(ctx, idx ? n : 1), bad(ctx, 1);
    if (!dd.p || !di.p || !bad.p) return ctx->fail(B3C_E_NOMEM, "b3c_mat_row_nnz");
    int rc;
    if (idx && (rc = b3c_copy_in(ctx, di.p, idx, sizeof(int32_t) * (size_t)n))) return rc;
    HIP_TRY(ctx, hipMemsetAsync(bad.p, 0, sizeof(int32_t), ctx->stream));
    hipLaunchKernelGGL(k_row_nnz, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n, m->indptr, idx ? di.p : nullptr, dd.p, bad.p);
    HIP_TRY(ctx, hipGetLastError());
    int32_t hbad = 0;
    if ((rc = b3c_readback(ctx, &hbad, bad.p, sizeof(hbad)))) return rc;
    if (hbad) return ctx->fail(B3C_E_ARG, "b3c_mat_row_nnz: idx holds a row id outside [0, %lld)", (long long)n);
    return b3c_copy_out(ctx, out, dd.p, sizeof(int64_t) * (size_t)n);
}

extern "C" int b3c_site_norm(b3c_ctx *ctx, const b3c_mat *counts, const double *sites, b3c_mat **out)
{
    if (!ctx) return B3C_E_ARG;
    if (!counts || counts->kind != B3C_U32 || !sites || !out) return ctx->fail(B3C_E_ARG, "b3c_site_norm: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t n = counts->n;
    MatOwner own(ctx);            // released on every early return below (HIP_TRY included)
    int rc = b3c_mat_alloc(ctx, n, counts->nnz, B3C_F64, &own.m);
    if (rc) return rc;
    b3c_mat *m = own.m;
    DevBuf<double> ds(ctx, n);
    if (!ds.p) return ctx->fail(B3C_E_NOMEM, "b3c_site_norm");
    if ((rc = b3c_copy_in(ctx, ds.p, sites, sizeof(double) * (size_t)n))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(m->indptr, counts->indptr, sizeof(int64_t) * (size_t)(n + 1), hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(m->indices, counts->indices, sizeof(int32_t) * (size_t)counts->nnz, hipMemcpyDeviceToDevice, ctx->stream));
    if (n) {
        const int lpr = pick_lpr(counts);
        DISPATCH_LPR(lpr, k_site_norm, row_grid(ctx, n, lpr), ctx->stream, n, counts->indptr, counts->indices,
                     (const uint32_t *)counts->data, ds.p, (double *)m->data);
    }
    m->ordered = counts->ordered;
    m->row0 = counts->row0;
    m->row1 = counts->row1;
    m->row_block = counts->row_block;
    m->sorted_cols = counts->sorted_cols;
    if (counts->ids) {
        m->ids = (int32_t *)ctx->alloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
        if (!m->ids) return ctx->fail(B3C_E_NOMEM, "b3c_site_norm");
        HIP_TRY(ctx, hipMemcpyAsync(m->ids, counts->ids, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *out = own.release();
    return B3C_OK;
}

extern "C" int b3c_scale_sym(b3c_ctx *ctx, b3c_mat *a, const double *x)
{
    if (!ctx) return B3C_E_ARG;
    if (!a || a->kind != B3C_F64 || !x) return ctx->fail(B3C_E_ARG, "b3c_scale_sym: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t n = a->n;
    if (n == 0) return B3C_OK;
    DevBuf<double> dx(ctx, n);
    if (!dx.p) return ctx->fail(B3C_E_NOMEM, "b3c_scale_sym");
    int rc = b3c_copy_in(ctx, dx.p, x, sizeof(double) * (size_t)n);
    if (rc) return rc;
    const int lpr = pick_lpr(a);
    DISPATCH_LPR(lpr, k_scale_sym, row_grid(ctx, n, lpr), ctx->stream, n, a->indptr, a->indices, (double *)a->data, dx.p);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return B3C_OK;
}

extern "C" int b3c_scale_sym_graph(b3c_ctx *ctx, b3c_mat *a, const double *x)
{
    if (!ctx) return B3C_E_ARG;
    if (!a || a->kind != B3C_F64 || !x) return ctx->fail(B3C_E_ARG, "b3c_scale_sym_graph: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t n = a->n;
    if (n == 0) return B3C_OK;
    DevBuf<double> dx(ctx, n);
    if (!dx.p) return ctx->fail(B3C_E_NOMEM, "b3c_scale_sym_graph");
    int rc = b3c_copy_in(ctx, dx.p, x, sizeof(double) * (size_t)n);
    if (rc) return rc;
    const int lpr = pick_lpr(a);
    DISPATCH_LPR(lpr, k_scale_sym_graph, row_grid(ctx, n, lpr), ctx->stream, n, a->indptr, a->indices, (double *)a->data,
                 dx.p, a->ids);
    a->value_symmetric = true;
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return B3C_OK;
}

extern "C" int b3c_normalise_scale(b3c_ctx *ctx, const b3c_mat *counts, const double *sites, const double *x, int graph_form,
                                   b3c_mat **out)
{
    if (!ctx) return B3C_E_ARG;
    if (!counts || counts->kind != B3C_U32 || !sites || !x || !out) return ctx->fail(B3C_E_ARG, "b3c_normalise_scale: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t n = counts->n;
    MatOwner own(ctx);            // released on every early return below (HIP_TRY included)
    int rc = b3c_mat_alloc(ctx, n, counts->nnz, B3C_F64, &own.m);
    if (rc) return rc;
    b3c_mat *m = own.m;
    DevBuf<double> ds(ctx, n), dx(ctx, n);
    if (!ds.p || !dx.p) return ctx->fail(B3C_E_NOMEM, "b3c_normalise_scale");
    if ((rc = b3c_copy_in(ctx, ds.p, sites, sizeof(double) * (size_t)n)) || (rc = b3c_copy_in(ctx, dx.p, x, sizeof(double) * (size_t)n))) {
        return rc;
    }
    HIP_TRY(ctx, hipMemcpyAsync(m->indptr, counts->indptr, sizeof(int64_t) * (size_t)(n + 1), hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(m->indices, counts->indices, sizeof(int32_t) * (size_t)counts->nnz, hipMemcpyDeviceToDevice, ctx->stream));
    if (n) {
        const int lpr = pick_lpr(counts);
        DISPATCH_LPR(lpr, k_norm_scale, row_grid(ctx, n, lpr), ctx->stream, n, counts->indptr, counts->indices,
                     (const uint32_t *)counts->data, ds.p, dx.p, counts->ids, graph_form ? 1 : 0, (double *)m->data);
    }
    m->ordered = counts->ordered;
    m->row0 = counts->row0;
    m->row1 = counts->row1;
    m->row_block = counts->row_block;
    m->sorted_cols = counts->sorted_cols;
    m->value_symmetric = graph_form != 0;
    if (counts->ids) {
        m->ids = (int32_t *)ctx->alloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
        if (!m->ids) return ctx->fail(B3C_E_NOMEM, "b3c_normalise_scale");
        HIP_TRY(ctx, hipMemcpyAsync(m->ids, counts->ids, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *out = own.release();
    return B3C_OK;
}

// what k_compress_fill_ns needs of a contig, as a row and as a stored column: one 32-byte record
struct __attribute__((aligned(32))) ColRec {
    double2 sx;      // (site count, x)
    int2 ni;         // (new id | -1 when masked out, caller-side id)
    int2 pad;
};

// compress of the balanced map WITHOUT materialising it: the kept entries are computed on the fly from the raw counts
// exactly as k_norm_scale computes them (same expression, same operand order: bit-identical values).  Per column one
// gathered record -- (site count, x, new id or -1 if masked, caller-side id) -- and up to 256 entries of a row in flight at once.
__global__ void __launch_bounds__(BLK) k_compress_fill_ns(int64_t n, const int64_t *__restrict__ indptr,
                                                           const int32_t *__restrict__ indices,
                                                           const uint32_t *__restrict__ counts,
                                                           const ColRec *__restrict__ rec,      // per contig: site count, x, new id | -1, caller-side id
                                                           int graph_form,
                                                           const int64_t *__restrict__ out_indptr,
                                                           int32_t *__restrict__ out_indices, double *__restrict__ out_a,
                                                           unsigned long long *__restrict__ vstat,   // [0] bits of the largest value, [1] a negative one was seen
                                                           double *__restrict__ vpart)               // per-workgroup sums of the values (diagonal twice)
{
    double vsum = 0.0;            // sum of the values written, diagonal entries twice (an estimate: see b3c_mat.vsum2)
    constexpr int U = 4;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (BLK / 64) + (threadIdx.x >> 6);
    const int64_t nwave = (int64_t)gridDim.x * (BLK / 64);
    double vmax = -INFINITY;
    // software pipeline over the rows of this wave: while the gathers of row r are in flight, the header and the first
    // 256 entries of the wave's next row are already being fetched (a dropped row passes through as an empty one)
    int64_t row = wave;
    int2 nr = make_int2(-1, 0);
    int64_t s = 0, e = 0;
    if (row < n) {
        nr = rec[row].ni;
        if (nr.x >= 0) { s = indptr[row]; e = indptr[row + 1]; }
    }
    int32_t j[U];
    uint32_t c[U];
#pragma unroll
    for (int t = 0; t < U; t++) {
        const int64_t k = s + t * 64 + lane;
        const bool ok = k < e;
        j[t] = ok ? indices[k] : -1;
        c[t] = ok ? counts[k] : 0u;
    }
    while (row < n) {
        const int64_t rown = row + nwave;
        int2 nrn = make_int2(-1, 0);
        int64_t sn = 0, en = 0;
        if (rown < n) {
            nrn = rec[rown].ni;
            if (nrn.x >= 0) { sn = indptr[rown]; en = indptr[rown + 1]; }
        }
        int64_t o = 0;
        double2 pr = make_double2(1.0, 0.0);
        if (nr.x >= 0) { o = out_indptr[nr.x]; pr = rec[row].sx; }
        const double si = pr.x == 0.0 ? 1.0 : pr.x;
        const double xr = pr.y;
        const int32_t idr = nr.y;
        for (int64_t base = s; base < e; base += 64 * U) {
            // everything the kernel needs of a column sits in ONE 32-byte record (two loads from one line; as two tables -- 8 + 16
            // bytes -- every stored column cost two line look-ups)
            int2 nj[U];
            double2 pj[U];
#pragma unroll
            for (int t = 0; t < U; t++) {
                nj[t] = j[t] >= 0 ? rec[j[t]].ni : make_int2(-1, 0);
                pj[t] = j[t] >= 0 ? rec[j[t]].sx : make_double2(1.0, 0.0);
            }
            // the next 256 entries: of this row, or (after its last chunk) the first ones of the wave's next row
            const bool last = base + 64 * U >= e;
            const int64_t nb = last ? sn : base + 64 * U, ne = last ? en : e;
            int32_t j2[U];
            uint32_t c2[U];
#pragma unroll
            for (int t = 0; t < U; t++) {
                const int64_t k = nb + t * 64 + lane;
                const bool ok = k < ne;
                j2[t] = ok ? indices[k] : -1;
                c2[t] = ok ? counts[k] : 0u;
            }
#pragma unroll
            for (int t = 0; t < U; t++) {
                const bool keep = nj[t].x >= 0;
                const unsigned long long b = __ballot(keep);
                if (keep) {
                    const int pos = __popcll(b & ((1ull << lane) - 1ull));
                    const double sj = pj[t].x == 0.0 ? 1.0 : pj[t].x;
                    const double a = (double)c[t] * (1.0 / (si * sj));
                    const double xj = pj[t].y;
                    double val;
                    if (graph_form) {
                        const double xhi = idr >= nj[t].y ? xr : xj, xlo = idr >= nj[t].y ? xj : xr;
                        val = xhi * (a * xlo);
                    } else {
                        val = xr * (a * xj);
                    }
                    out_indices[o + pos] = nj[t].x;
                    out_a[o + pos] = val;
                    vmax = fmax(vmax, val);
                    const bool dg = nj[t].x == nr.x;
                    vsum += dg ? 2.0 * val : val;
                }
                o += __popcll(b);
            }
#pragma unroll
            for (int t = 0; t < U; t++) { j[t] = j2[t]; c[t] = c2[t]; }
        }
        if (e == s) {
            // nothing of this row was processed: the prefetch of the next row has not been issued yet
#pragma unroll
            for (int t = 0; t < U; t++) {
                const int64_t k = sn + t * 64 + lane;
                const bool ok = k < en;
                j[t] = ok ? indices[k] : -1;
                c[t] = ok ? counts[k] : 0u;
            }
        }
        row = rown;
        nr = nrn;
        s = sn;
        e = en;
    }
    // the largest value written, for b3c_max_value (fmax semantics: NaNs are skipped); bit patterns of non-negative
    // doubles order like the values, and the running maximum is read first so that few waves reach the atomic
    vmax = wave_max(vmax);
    if (lane == 0) {
        if (vmax > 0.0) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(vmax);
            if (bits > __atomic_load_n(&vstat[0], __ATOMIC_RELAXED)) atomicMax(&vstat[0], bits);
        } else if (vmax < 0.0 && vmax > -INFINITY) {
            vstat[1] = 1ull;
        }
    }
    // one partial per workgroup, summed by the host: no atomics (a double atomicAdd is a compare-and-swap loop here; and even
    // an integer atomic per workgroup on the line every wave polls for the maximum cost the C2-sized kernel 390 of its 476 us)
    __shared__ double vred[BLK / 64];
    vsum = wave_sum(vsum);
    if (lane == 0) vred[threadIdx.x >> 6] = vsum;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < BLK / 64; w++) t += vred[w];
        vpart[blockIdx.x] = t;
    }
}

struct NormScale {          // when set, compress_impl computes the values from raw counts (k_compress_fill_ns)
    const double *sites, *x;   // per row, in HBM
    int graph_form;
};

__global__ void k_pack_rec(int64_t n, const double *__restrict__ sites, const double *__restrict__ x, const uint8_t *__restrict__ mask,
                           const int64_t *__restrict__ newid, const int32_t *__restrict__ ids, ColRec *__restrict__ rec)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        ColRec r;
        r.sx = make_double2(sites[i], x[i]);
        r.ni = make_int2(mask[i] ? (int)newid[i] : -1, ids ? ids[i] : (int)i);
        r.pad = make_int2(0, 0);
        rec[i] = r;
    }
}

static int compress_impl(b3c_ctx *ctx, const b3c_mat *a, const uint8_t *mask, const NormScale *ns, b3c_mat **out);

extern "C" int b3c_compress(b3c_ctx *ctx, const b3c_mat *a, const uint8_t *mask, b3c_mat **out)
{
    if (!ctx) return B3C_E_ARG;
    if (!a || !mask || !out) return ctx->fail(B3C_E_ARG, "b3c_compress: bad argument");
    return compress_impl(ctx, a, mask, nullptr, out);
}

extern "C" int b3c_normalise_scale_compress(b3c_ctx *ctx, const b3c_mat *counts, const double *sites, const double *x,
                                            int graph_form, const uint8_t *mask, b3c_mat **out)
{
    if (!ctx) return B3C_E_ARG;
    if (!counts || counts->kind != B3C_U32 || !sites || !x || !mask || !out)
        return ctx->fail(B3C_E_ARG, "b3c_normalise_scale_compress: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t n = counts->n;
    DevBuf<double> ds(ctx, n), dx(ctx, n);
    if (!ds.p || !dx.p) return ctx->fail(B3C_E_NOMEM, "b3c_normalise_scale_compress");
    int rc;
    if ((rc = b3c_copy_in(ctx, ds.p, sites, sizeof(double) * (size_t)n)) || (rc = b3c_copy_in(ctx, dx.p, x, sizeof(double) * (size_t)n)))
        return rc;
    const NormScale ns{ds.p, dx.p, graph_form ? 1 : 0};
    return compress_impl(ctx, counts, mask, &ns, out);
}

static int compress_impl(b3c_ctx *ctx, const b3c_mat *a, const uint8_t *mask, const NormScale *ns, b3c_mat **out)
{
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t n = a->n;
    hipStream_t s = ctx->stream;
    DevBuf<uint8_t> dm(ctx, n + 1);
    DevBuf<int64_t> keep(ctx, n + 1), newid(ctx, n + 1), cnt(ctx, n + 1), optr(ctx, n + 2);
    if (!dm.p || !keep.p || !newid.p || !cnt.p || !optr.p) return ctx->fail(B3C_E_NOMEM, "b3c_compress");
    int rc = b3c_copy_in(ctx, dm.p, mask, (size_t)n);
    if (rc) return rc;
    // newid = i - cumsum(~mask)[i] for kept i  ==  exclusive scan of the keep flags (sparse_utils.py:307-312)
    HIP_TRY(ctx, hipMemsetAsync(keep.p, 0, sizeof(int64_t) * (size_t)(n + 1), s));
    if (n) hipLaunchKernelGGL(k_mask_to_i64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, dm.p, keep.p);
    if ((rc = b3c_exclusive_scan_i64(ctx, keep.p, newid.p, n + 1))) return rc;
    int64_t m = 0;
    { const int rb_ = b3c_readback(ctx, &m, newid.p + n, sizeof(int64_t)); if (rb_) return rb_; }
    HIP_TRY(ctx, hipMemsetAsync(cnt.p, 0, sizeof(int64_t) * (size_t)(m + 1), s));
    const int lpr = pick_lpr(a);
    // (the graph form is only asked for on a bitwise symmetric map: b3c_kr_info.bitwise_symmetric)
    const bool from_dropped = ns && ns->graph_form && !a->row_block && (n - m) * 4 < n;
    if (n && from_dropped) {
        hipLaunchKernelGGL(k_compress_cnt_init, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, a->indptr, dm.p, newid.p, cnt.p);
        if (m < n)
            hipLaunchKernelGGL(k_compress_cnt_sub, dim3(grid_for(n, 64 * (BLK / 64), 8 * ctx->n_cu)), dim3(BLK), 0, s, n, a->indptr, a->indices, dm.p, newid.p, cnt.p);
    } else if (n) {
        DISPATCH_LPR(lpr, k_compress_count, row_grid(ctx, n, lpr), s, n, a->indptr, a->indices, dm.p, newid.p, cnt.p);
    }
    if ((rc = b3c_exclusive_scan_i64(ctx, cnt.p, optr.p, m + 1))) return rc;
    int64_t nnz = 0;
    { const int rb_ = b3c_readback(ctx, &nnz, optr.p + m, sizeof(int64_t)); if (rb_) return rb_; }
    MatOwner own(ctx);            // freed on every early return below
    if ((rc = b3c_mat_alloc(ctx, m, nnz, ns ? B3C_F64 : a->kind, &own.m))) return rc;
    b3c_mat *o = own.m;
    HIP_TRY(ctx, hipMemcpyAsync(o->indptr, optr.p, sizeof(int64_t) * (size_t)(m + 1), hipMemcpyDeviceToDevice, s));
    if (n) {
        const int g = grid_for(n, BLK / 64, 16 * ctx->n_cu);
        if (ns) {
            DevBuf<ColRec> rec(ctx, n + 1);
            if (!rec.p) { return ctx->fail(B3C_E_NOMEM, "b3c_normalise_scale_compress"); }
            hipLaunchKernelGGL(k_pack_rec, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, ns->sites, ns->x, dm.p, newid.p, a->ids, rec.p);
            // [4 counters | one value-sum partial per workgroup]: read back together through the pinned staging buffer
            DevBuf<unsigned long long> vstat(ctx, 4 + (size_t)g);
            if (!vstat.p) { return ctx->fail(B3C_E_NOMEM, "b3c_normalise_scale_compress"); }
            HIP_TRY(ctx, hipMemsetAsync(vstat.p, 0, sizeof(unsigned long long) * 4, s));
            hipLaunchKernelGGL(k_compress_fill_ns, dim3(g), dim3(BLK), 0, s, n, a->indptr, a->indices, (const uint32_t *)a->data,
                               rec.p, ns->graph_form, optr.p, o->indices, (double *)o->data, vstat.p, (double *)(vstat.p + 4));
            // (the read-back also ends the use of `rec`, which goes back to the pool at the end of this scope)
            std::vector<unsigned long long> hvv(4 + (size_t)g, 0ull);
            { const int rb_ = b3c_readback(ctx, hvv.data(), vstat.p, sizeof(unsigned long long) * hvv.size()); if (rb_) return rb_; }
            const unsigned long long *hv = hvv.data();
            if (hv[1] == 0 && nnz > 0) {
                // all values non-negative: the largest one is known without another pass (b3c_max_value)
                double vm;
                memcpy(&vm, &hv[0], sizeof(double));
                o->vmax = vm;
                o->vmax_valid = true;
                double vs = 0.0;
                for (int b = 0; b < g; b++) { double t; memcpy(&t, &hv[4 + b], sizeof(double)); vs += t; }
                o->vsum2 = vs;
                o->ndiag = o->n;                     // (an upper bound is all the estimate needs)
                o->vsum_valid = true;
            }
        }
        else if (a->kind == B3C_F64)
            hipLaunchKernelGGL(k_compress_fill<double>, dim3(g), dim3(BLK), 0, s, n, a->indptr, a->indices,
                               (const double *)a->data, dm.p, newid.p, optr.p, o->indices, (double *)o->data);
        else
            hipLaunchKernelGGL(k_compress_fill<uint32_t>, dim3(g), dim3(BLK), 0, s, n, a->indptr, a->indices,
                               (const uint32_t *)a->data, dm.p, newid.p, optr.p, o->indices, (uint32_t *)o->data);
    }
    o->ordered = a->ordered;
    o->value_symmetric = ns ? ns->graph_form != 0 : a->value_symmetric;
    o->sorted_cols = a->sorted_cols;      // compaction keeps the relative order inside a row
    if (a->row1 >= 0) {
        // a row block stays a row block: its bounds in the compacted numbering.  Bounds at 0 and n are known (0 and m) -- a
        // whole matrix that went through b3c_mat_permute carries [0, n) and costs no read-back; an inner bound is newid[bound]
        int64_t b[2] = {0, m};
        const bool need0 = a->row0 > 0 && a->row0 < n, need1 = a->row1 > 0 && a->row1 < n;
        if (a->row0 >= n) b[0] = m;
        if (a->row1 <= 0) b[1] = 0;
        if ((need0 || need1) && ctx->pin) {
            int64_t *pin = (int64_t *)ctx->pin;
            if (need0) HIP_TRY(ctx, hipMemcpyAsync(pin, newid.p + a->row0, sizeof(int64_t), hipMemcpyDeviceToHost, s));
            if (need1) HIP_TRY(ctx, hipMemcpyAsync(pin + 1, newid.p + a->row1, sizeof(int64_t), hipMemcpyDeviceToHost, s));
            HIP_TRY(ctx, hipStreamSynchronize(s));      // both bounds, one wait
            if (need0) b[0] = pin[0];
            if (need1) b[1] = pin[1];
        } else {
            if (need0) { const int rb_ = b3c_readback(ctx, &b[0], newid.p + a->row0, sizeof(int64_t)); if (rb_) { return rb_; } }
            if (need1) { const int rb_ = b3c_readback(ctx, &b[1], newid.p + a->row1, sizeof(int64_t)); if (rb_) { return rb_; } }
        }
        o->row0 = b[0];
        o->row1 = b[1];
        o->row_block = a->row_block;
    }
    if (a->ids && n) {
        DevBuf<int64_t> keep_o(ctx, n + 1), rank_o(ctx, n + 1);
        o->ids = (int32_t *)ctx->alloc(sizeof(int32_t) * (size_t)(m > 0 ? m : 1));
        if (!keep_o.p || !rank_o.p || !o->ids) { return ctx->fail(B3C_E_NOMEM, "b3c_compress"); }
        HIP_TRY(ctx, hipMemsetAsync(keep_o.p, 0, sizeof(int64_t) * (size_t)(n + 1), s));
        hipLaunchKernelGGL(k_scatter_keep, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, dm.p, a->ids, keep_o.p);
        if ((rc = b3c_exclusive_scan_i64(ctx, keep_o.p, rank_o.p, n + 1))) { return rc; }
        hipLaunchKernelGGL(k_sub_ids, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, dm.p, a->ids, newid.p, rank_o.p, o->ids);
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(s));
    *out = own.release();
    return B3C_OK;
}

extern "C" int b3c_max_value(b3c_ctx *ctx, const b3c_mat *a, double *out)
{
    if (!ctx) return B3C_E_ARG;
    if (!a || a->kind != B3C_F64 || !out) return ctx->fail(B3C_E_ARG, "b3c_max_value: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    double m = -INFINITY;
    if (a->vmax_valid) {
        m = a->vmax;              // left behind by the kernel that wrote the values (k_compress_fill_ns)
    } else {
        const int g = grid_for(a->nnz, BLK * 4, 4 * ctx->n_cu);
        DevBuf<double> part(ctx, g);
        if (!part.p) return ctx->fail(B3C_E_NOMEM, "b3c_max_value");
        hipLaunchKernelGGL(k_max_value, dim3(g), dim3(BLK), 0, ctx->stream, a->nnz, (const double *)a->data, part.p);
        std::vector<double> h(g);
        { const int rb_ = b3c_readback(ctx, h.data(), part.p, sizeof(double) * g); if (rb_) return rb_; }
        for (int i = 0; i < g; i++) m = h[i] > m ? h[i] : m;
        b3c_mat *am = const_cast<b3c_mat *>(a);
        am->vmax = m;
        am->vmax_valid = true;
    }
    // implicit zeros take part in scipy's max unless the matrix is completely dense
    if ((double)a->nnz < (double)a->n * (double)a->n && m < 0.0) m = 0.0;
    *out = m;
    return B3C_OK;
}
