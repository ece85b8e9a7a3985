// Block reduction of a device-resident CSR map for the heat-map step that follows clustering
// (reference: mzd/sparse_utils.py:48-87 `downsample`, called from ContactMap.plot, contact_map.py:1340-1349).
//
// out[I, J] = reduce over the block_size x block_size cells (i, j), i in [I*b, (I+1)*b), j in [J*b, (J+1)*b)
//   mean: (sum of the stored cells) * (1 / b^2)   -- zero padding at the edges is not compensated, as in the reference
//   max : maximum of the cells, implicit zeros (missing cells, padding) taking part
//
// One workgroup per block-row.  The b input rows of a block-row are contiguous in CSR, so the block-row is ONE
// flat entry range.  Which output columns exist is found with a bitmap in LDS (one bit per block-column, in
// chunks of DS_CHUNK columns), whose per-word prefix popcounts also give the output slot of every entry.  Values
// are accumulated row by row in a fixed order (runs of entries that fall into the same cell block are reduced
// first, rows follow each other): no floating-point atomics, the result is the same on every run.
// Plot-sized outputs (up to 32768 block-columns) take the wave-per-block-row form below.
#include "common.hpp"

#include <cstdlib>

int b3c_exclusive_scan_i64(b3c_ctx *ctx, const int64_t *in, int64_t *out, int64_t count);

namespace {

constexpr int DS_BLK = 256;
constexpr int DS_WORDS = 2048;                 // 64-bit bitmap words per chunk (16 KB of LDS)
constexpr int64_t DS_CHUNK = 64ll * DS_WORDS;   // block-columns per chunk
constexpr int DS_WPT = DS_WORDS / DS_BLK;       // words per thread in the prefix pass

struct DsLds {
    unsigned long long bits[DS_WORDS];
    unsigned int prefix[DS_WORDS];
    unsigned int part[DS_BLK];
    unsigned int total;
};

// Marks the block-columns of chunk [c0, c0 + DS_CHUNK) present in entries [e0, e1); leaves bits/prefix/total in LDS.
__device__ __forceinline__ void ds_mark(DsLds &L, const int32_t *__restrict__ indices, int64_t e0, int64_t e1, uint32_t b,
                                        int64_t c0)
{
    const int t = threadIdx.x;
    for (int w = t; w < DS_WORDS; w += DS_BLK) L.bits[w] = 0ull;
    __syncthreads();
    for (int64_t e = e0 + t; e < e1; e += DS_BLK) {
        const int64_t J = (int64_t)((uint32_t)indices[e] / b) - c0;
        if (J >= 0 && J < DS_CHUNK) atomicOr(&L.bits[J >> 6], 1ull << (J & 63));
    }
    __syncthreads();
    unsigned int loc[DS_WPT];
    unsigned int sum = 0;
#pragma unroll
    for (int k = 0; k < DS_WPT; k++) {
        loc[k] = sum;
        sum += (unsigned int)__popcll(L.bits[t * DS_WPT + k]);
    }
    L.part[t] = sum;
    __syncthreads();
    // exclusive scan of the per-thread sums (Hillis-Steele over 256 values)
    unsigned int v = sum;
    for (int o = 1; o < DS_BLK; o <<= 1) {
        const unsigned int add = t >= o ? L.part[t - o] : 0u;
        __syncthreads();
        v += add;
        L.part[t] = v;
        __syncthreads();
    }
    const unsigned int excl = v - sum;
#pragma unroll
    for (int k = 0; k < DS_WPT; k++) L.prefix[t * DS_WPT + k] = excl + loc[k];
    if (t == DS_BLK - 1) L.total = v;
    __syncthreads();
}

__global__ void __launch_bounds__(DS_BLK) k_ds_count(int64_t n, int64_t nb, uint32_t b, const int64_t *__restrict__ indptr,
                                                     const int32_t *__restrict__ indices, int64_t *__restrict__ cnt)
{
    __shared__ DsLds L;
    for (int64_t I = blockIdx.x; I < nb; I += gridDim.x) {
        const int64_t r0 = I * (int64_t)b, r1 = min(n, r0 + (int64_t)b);
        const int64_t e0 = indptr[r0], e1 = indptr[r1];
        int64_t total = 0;
        if (e1 > e0)
            for (int64_t c0 = 0; c0 < nb; c0 += DS_CHUNK) {
                ds_mark(L, indices, e0, e1, b, c0);
                total += L.total;
                __syncthreads();
            }
        if (threadIdx.x == 0) cnt[I] = total;
    }
}

template <bool MAX>
__global__ void __launch_bounds__(DS_BLK) k_ds_fill(int64_t n, int64_t nb, uint32_t b, const int64_t *__restrict__ indptr,
                                                    const int32_t *__restrict__ indices, const double *__restrict__ data,
                                                    const int64_t *__restrict__ optr, int32_t *__restrict__ oidx,
                                                    double *odata, unsigned int *ocnt,
                                                    double inv_area, unsigned long long area, int skip_diag)
{
    __shared__ DsLds L;
    const int t = threadIdx.x;
    for (int64_t I = blockIdx.x; I < nb; I += gridDim.x) {
        const int64_t r0 = I * (int64_t)b, r1 = min(n, r0 + (int64_t)b);
        const int64_t e0 = indptr[r0], e1 = indptr[r1];
        if (e1 <= e0) continue;
        int64_t base = optr[I];
        for (int64_t c0 = 0; c0 < nb; c0 += DS_CHUNK) {
            ds_mark(L, indices, e0, e1, b, c0);
            const unsigned int total = L.total;
            if (total == 0) { __syncthreads(); continue; }
            // output columns of this chunk, ascending; accumulators start from the identity
#pragma unroll
            for (int k = 0; k < DS_WPT; k++) {
                const int w = t * DS_WPT + k;
                unsigned long long m = L.bits[w];
                int64_t slot = base + L.prefix[w];
                while (m) {
                    const int bit = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    oidx[slot] = (int32_t)(c0 + (int64_t)w * 64 + bit);
                    odata[slot] = MAX ? -INFINITY : 0.0;
                    if (MAX) ocnt[slot] = 0u;
                    slot++;
                }
            }
            __threadfence_block();
            __syncthreads();
            // rows in order; inside a row every run of entries of one block-column is summed left to right by the
            // thread that owns its first entry, and no two runs of a row share a slot
            for (int64_t r = r0; r < r1; r++) {
                const int64_t a0 = indptr[r], a1 = indptr[r + 1];
                for (int64_t e = a0 + t; e < a1; e += DS_BLK) {
                    const uint32_t Jg = (uint32_t)indices[e] / b;
                    const int64_t J = (int64_t)Jg - c0;
                    if (J < 0 || J >= DS_CHUNK) continue;
                    if (e > a0 && (uint32_t)indices[e - 1] / b == Jg) continue;     // not the head of its run
                    const bool dg = skip_diag && (int64_t)indices[e] == r;         // a zeroed diagonal is an absent cell
                    double acc = dg ? (MAX ? -INFINITY : 0.0) : data[e];
                    unsigned int c = dg ? 0u : 1u;
                    for (int64_t f = e + 1; f < a1 && (uint32_t)indices[f] / b == Jg; f++) {
                        if (skip_diag && (int64_t)indices[f] == r) continue;
                        acc = MAX ? fmax(acc, data[f]) : acc + data[f];
                        c++;
                    }
                    const int w = (int)(J >> 6);
                    const int64_t slot = base + L.prefix[w] + __popcll(L.bits[w] & ((1ull << (J & 63)) - 1ull));
                    odata[slot] = MAX ? fmax(odata[slot], acc) : odata[slot] + acc;
                    if (MAX) ocnt[slot] += c;
                }
                __threadfence_block();
                __syncthreads();
            }
            for (int64_t k = t; k < (int64_t)total; k += DS_BLK) {
                const int64_t slot = base + k;
                if (MAX) {
                    if ((unsigned long long)ocnt[slot] < area) odata[slot] = fmax(odata[slot], 0.0);
                } else
                    odata[slot] = odata[slot] * inv_area;
            }
            base += total;
            __syncthreads();
        }
    }
}

// ---- wave-per-block-row form for plot-sized outputs (nb <= DW_CHUNK): no workgroup barriers, 4 block-rows per workgroup.
// A wave executes its LDS and memory operations in program order, so the row-by-row accumulation needs only compiler
// fences between the steps.
constexpr int DW_WORDS = 512;                     // 32768 block-columns, 4 KB + 2 KB of LDS per wave
constexpr int64_t DW_CHUNK = 64ll * DW_WORDS;
constexpr int DW_WPL = DW_WORDS / 64;             // words per lane

struct DwLds {
    unsigned long long bits[DW_WORDS];
    unsigned int prefix[DW_WORDS];
};

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
}

// -> number of distinct block-columns in entries [e0, e1); bits/prefix left in the wave's LDS
__device__ __forceinline__ unsigned int dw_mark(DwLds &L, const int32_t *__restrict__ indices, int64_t e0, int64_t e1, uint32_t b)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < DW_WPL; k++) L.bits[lane * DW_WPL + k] = 0ull;
    wave_sync();
    for (int64_t e = e0 + lane; e < e1; e += 64) {
        const uint32_t J = (uint32_t)indices[e] / b;
        atomicOr(&L.bits[J >> 6], 1ull << (J & 63));
    }
    wave_sync();
    unsigned int loc[DW_WPL];
    unsigned int sum = 0;
#pragma unroll
    for (int k = 0; k < DW_WPL; k++) {
        loc[k] = sum;
        sum += (unsigned int)__popcll(L.bits[lane * DW_WPL + k]);
    }
    unsigned int v = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned int up = __shfl_up(v, o, 64);
        if (lane >= o) v += up;
    }
    const unsigned int excl = v - sum;
#pragma unroll
    for (int k = 0; k < DW_WPL; k++) L.prefix[lane * DW_WPL + k] = excl + loc[k];
    wave_sync();
    return __shfl(v, 63, 64);
}

__global__ void __launch_bounds__(DS_BLK) k_dw_count(int64_t n, int64_t nb, uint32_t b, const int64_t *__restrict__ indptr,
                                                     const int32_t *__restrict__ indices, int64_t *__restrict__ cnt)
{
    __shared__ DwLds LL[DS_BLK / 64];
    DwLds &L = LL[threadIdx.x >> 6];
    for (int64_t I = (int64_t)blockIdx.x * (DS_BLK / 64) + (threadIdx.x >> 6); I < nb; I += (int64_t)gridDim.x * (DS_BLK / 64)) {
        const int64_t r0 = I * (int64_t)b, r1 = min(n, r0 + (int64_t)b);
        const int64_t e0 = indptr[r0], e1 = indptr[r1];
        const unsigned int total = e1 > e0 ? dw_mark(L, indices, e0, e1, b) : 0u;
        if ((threadIdx.x & 63) == 0) cnt[I] = total;
    }
}

template <bool MAX>
__global__ void __launch_bounds__(DS_BLK) k_dw_fill(int64_t n, int64_t nb, uint32_t b, const int64_t *__restrict__ indptr,
                                                    const int32_t *__restrict__ indices, const double *__restrict__ data,
                                                    const int64_t *__restrict__ optr, int32_t *__restrict__ oidx,
                                                    double *odata, unsigned int *ocnt, double inv_area,
                                                    unsigned long long area, int skip_diag)
{
    __shared__ DwLds LL[DS_BLK / 64];
    DwLds &L = LL[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    for (int64_t I = (int64_t)blockIdx.x * (DS_BLK / 64) + (threadIdx.x >> 6); I < nb; I += (int64_t)gridDim.x * (DS_BLK / 64)) {
        const int64_t r0 = I * (int64_t)b, r1 = min(n, r0 + (int64_t)b);
        const int64_t e0 = indptr[r0], e1 = indptr[r1];
        if (e1 <= e0) continue;
        const int64_t base = optr[I];
        const unsigned int total = dw_mark(L, indices, e0, e1, b);
#pragma unroll
        for (int k = 0; k < DW_WPL; k++) {
            const int w = lane * DW_WPL + k;
            unsigned long long m = L.bits[w];
            int64_t slot = base + L.prefix[w];
            while (m) {
                const int bit = __ffsll((long long)m) - 1;
                m &= m - 1;
                oidx[slot] = (int32_t)(w * 64 + bit);
                odata[slot] = MAX ? -INFINITY : 0.0;
                if (MAX) ocnt[slot] = 0u;
                slot++;
            }
        }
        wave_sync();
        for (int64_t r = r0; r < r1; r++) {
            const int64_t a0 = indptr[r], a1 = indptr[r + 1];
            for (int64_t ee = a0; ee < a1; ee += 64) {
                // segmented reduction of the slice by block-column (columns ascend, so a segment is contiguous):
                // fixed shuffle tree, then the last lane of every segment updates the slot
                const int64_t e = ee + lane;
                const bool live = e < a1;
                const uint32_t col = live ? (uint32_t)indices[e] : 0u;
                const uint32_t Jg = live ? col / b : 0xffffffffu;
                const bool dg = skip_diag && live && (int64_t)col == r;            // a zeroed diagonal is an absent cell
                double acc = live && !dg ? data[e] : (MAX ? -INFINITY : 0.0);
                unsigned int c = dg ? 0u : 1u;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const double pa = __shfl_up(acc, o, 64);
                    const unsigned int pc = __shfl_up(c, o, 64);
                    const uint32_t pj = __shfl_up(Jg, o, 64);
                    if (lane >= o && pj == Jg) {
                        acc = MAX ? fmax(pa, acc) : pa + acc;
                        c += pc;
                    }
                }
                const uint32_t nj = __shfl_down(Jg, 1, 64);
                if (live && (lane == 63 || nj != Jg)) {
                    const int w = (int)(Jg >> 6);
                    const int64_t slot = base + L.prefix[w] + __popcll(L.bits[w] & ((1ull << (Jg & 63)) - 1ull));
                    odata[slot] = MAX ? fmax(odata[slot], acc) : odata[slot] + acc;
                    if (MAX) ocnt[slot] += c;
                }
                wave_sync();      // a run that continues in the next slice meets the same slot again
            }
        }
        for (int64_t k = lane; k < (int64_t)total; k += 64) {
            const int64_t slot = base + k;
            if (MAX) {
                if ((unsigned long long)ocnt[slot] < area) odata[slot] = fmax(odata[slot], 0.0);
            } else
                odata[slot] = odata[slot] * inv_area;
        }
        wave_sync();
    }
}

}  // namespace

extern "C" int b3c_downsample(b3c_ctx *ctx, const b3c_mat *a, int64_t block_size, int method, b3c_mat **out)
{
    if (!ctx) return B3C_E_ARG;
    if (!a || !out || a->kind != B3C_F64) return ctx->fail(B3C_E_ARG, "b3c_downsample: needs a float64 matrix");
    if (block_size < 2 || block_size > 0x7fffffffll) return ctx->fail(B3C_E_ARG, "b3c_downsample: block_size must be an integer larger than 1");
    const int skip_diag = (method & B3C_DS_SKIP_DIAG) ? 1 : 0;
    method &= ~B3C_DS_SKIP_DIAG;
    if (method != B3C_DS_MEAN && method != B3C_DS_MAX) return ctx->fail(B3C_E_ARG, "b3c_downsample: unknown method");
    if (!a->sorted_cols) return ctx->fail(B3C_E_ARG, "b3c_downsample: needs canonical CSR (ascending columns)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int64_t n = a->n, b = block_size, nb = (n + b - 1) / b;
    DevBuf<int64_t> cnt(ctx, nb + 1), optr(ctx, nb + 2);
    if (!cnt.p || !optr.p) return ctx->fail(B3C_E_NOMEM, "b3c_downsample");
    HIP_TRY(ctx, hipMemsetAsync(cnt.p, 0, sizeof(int64_t) * (size_t)(nb + 1), s));
    // plot-sized outputs: one wave per block-row; larger ones: one workgroup per block-row, column chunks
    const bool wave_form = nb <= DW_CHUNK && !CFG.on("B3C_DS_NO_WAVE");
    const int g = wave_form ? grid_for(nb, DS_BLK / 64, 16 * ctx->n_cu) : grid_for(nb, 1, 8 * ctx->n_cu);
    if (nb && wave_form)
        hipLaunchKernelGGL(k_dw_count, dim3(g), dim3(DS_BLK), 0, s, n, nb, (uint32_t)b, a->indptr, a->indices, cnt.p);
    else if (nb)
        hipLaunchKernelGGL(k_ds_count, dim3(g), dim3(DS_BLK), 0, s, n, nb, (uint32_t)b, a->indptr, a->indices, cnt.p);
    int rc;
    if ((rc = b3c_exclusive_scan_i64(ctx, cnt.p, optr.p, nb + 1))) return rc;
    int64_t nnz = 0;
    { const int rb_ = b3c_readback(ctx, &nnz, optr.p + nb, sizeof(int64_t)); if (rb_) return rb_; }
    MatOwner own(ctx);            // released on every early return below
    if ((rc = b3c_mat_alloc(ctx, nb, nnz, B3C_F64, &own.m))) return rc;
    b3c_mat *o = own.m;
    HIP_TRY(ctx, hipMemcpyAsync(o->indptr, optr.p, sizeof(int64_t) * (size_t)(nb + 1), hipMemcpyDeviceToDevice, s));
    if (nnz) {
        const double area = (double)b * (double)b;
        if (method == B3C_DS_MAX) {
            DevBuf<unsigned int> oc(ctx, nnz);
            if (!oc.p) return ctx->fail(B3C_E_NOMEM, "b3c_downsample");
            if (wave_form)
                hipLaunchKernelGGL(k_dw_fill<true>, dim3(g), dim3(DS_BLK), 0, s, n, nb, (uint32_t)b, a->indptr, a->indices,
                                   (const double *)a->data, optr.p, o->indices, (double *)o->data, oc.p, 1.0 / area,
                                   (unsigned long long)b * (unsigned long long)b, skip_diag);
            else
                hipLaunchKernelGGL(k_ds_fill<true>, dim3(g), dim3(DS_BLK), 0, s, n, nb, (uint32_t)b, a->indptr, a->indices,
                                   (const double *)a->data, optr.p, o->indices, (double *)o->data, oc.p, 1.0 / area,
                                   (unsigned long long)b * (unsigned long long)b, skip_diag);
            HIP_TRY(ctx, hipStreamSynchronize(s));
        } else if (wave_form)
            hipLaunchKernelGGL(k_dw_fill<false>, dim3(g), dim3(DS_BLK), 0, s, n, nb, (uint32_t)b, a->indptr, a->indices,
                               (const double *)a->data, optr.p, o->indices, (double *)o->data, (unsigned int *)nullptr,
                               1.0 / area, (unsigned long long)b * (unsigned long long)b, skip_diag);
        else
            hipLaunchKernelGGL(k_ds_fill<false>, dim3(g), dim3(DS_BLK), 0, s, n, nb, (uint32_t)b, a->indptr, a->indices,
                               (const double *)a->data, optr.p, o->indices, (double *)o->data, (unsigned int *)nullptr,
                               1.0 / area, (unsigned long long)b * (unsigned long long)b, skip_diag);
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(s));
    *out = own.release();
    return B3C_OK;
}
