/*
 * ORACLE (test infrastructure, not product code): scalar CPU restatement of the reference's
 * Knight-Ruiz balancing and the CSR helpers around it.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may call this.  Pinned against tests/golden/kr_*.npz, which were
 * produced by executing the reference's own function text (tools/make_golden.py).
 *
 * Every function cites the reference lines it follows (paths relative to the reference root).
 * Input precondition: canonical CSR (sorted column indices, no duplicates), int64 indptr,
 * int32 indices.  Build: gcc -O2 -ffp-contract=off (no FMA contraction: scipy/numpy do none).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

/* Threads for the row-parallel loops below (bench.py's multi-thread CPU baseline; SURVEY section 8(d) iii).  With
 * one thread (the default, and what every test uses) the pragmas are inert and the code is the plain sequential loop;
 * with more, every row is still summed sequentially by one thread, so the SpMV result does not change. */
static int g_threads = 1;
void orc_set_threads(int n) { g_threads = n > 1 ? n : 1; }
int orc_get_threads(void) { return g_threads; }

/* y = A x, row-sequential accumulation like scipy's csr_matvec (the m.dot() calls at
 * mzd/sparse_utils.py:136,165,196). */
static void csr_matvec(int64_t n, const int64_t *ip, const int32_t *ix, const double *a, const double *x, double *y)
{
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
    for (int64_t i = 0; i < n; i++) {
        double s = 0.0;
        for (int64_t k = ip[i]; k < ip[i + 1]; k++) s += a[k] * x[ix[k]];
        y[i] = s;
    }
}

static double dot(int64_t n, const double *a, const double *b)
{
    double s = 0.0;
    for (int64_t i = 0; i < n; i++) s += a[i] * b[i];
    return s;
}

/*
 * Working copy with zero diagonals replaced by one: mzd/sparse_utils.py:110-118.
 * `m.diagonal() == 0` is true both for a stored zero and for a structurally absent entry; the LIL
 * assignment `m[ix, ix] = 1` overwrites the former and inserts the latter in sorted position.
 */
static int64_t build_working(int64_t n, const int64_t *ip, const int32_t *ix, const double *a, int64_t **wip,
                             int32_t **wix, double **wa)
{
    int64_t n_zero = 0;
    *wip = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    /* pass 1 (row-parallel): is the diagonal stored, is it zero */
    uint8_t *absent = (uint8_t *)malloc((size_t)(n > 0 ? n : 1));
#pragma omp parallel for schedule(static) num_threads(g_threads) reduction(+ : n_zero) if (g_threads > 1)
    for (int64_t i = 0; i < n; i++) {
        int found = 0;
        double d = 0.0;
        for (int64_t k = ip[i]; k < ip[i + 1]; k++)
            if (ix[k] == i) { found = 1; d = a[k]; break; }
        absent[i] = !found;
        if (!found || d == 0.0) n_zero++;
    }
    int64_t o = 0;
    for (int64_t i = 0; i < n; i++) { (*wip)[i] = o; o += ip[i + 1] - ip[i] + absent[i]; }
    (*wip)[n] = o;
    const int64_t nnz = o;
    *wix = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1));
    *wa = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
    /* pass 2 (row-parallel): copy the row, the unit diagonal overwriting a stored zero or inserted in sorted position */
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
    for (int64_t i = 0; i < n; i++) {
        int64_t o = (*wip)[i];
        int placed = 0;
        for (int64_t k = ip[i]; k < ip[i + 1]; k++) {
            if (!placed && ix[k] >= i) {
                if (ix[k] == i) {
                    (*wix)[o] = (int32_t)i;
                    (*wa)[o++] = (a[k] == 0.0) ? 1.0 : a[k];
                    placed = 1;
                    continue;
                }
                (*wix)[o] = (int32_t)i;
                (*wa)[o++] = 1.0;
                placed = 1;
            }
            (*wix)[o] = ix[k];
            (*wa)[o++] = a[k];
        }
        if (!placed) {
            (*wix)[o] = (int32_t)i;
            (*wa)[o++] = 1.0;
        }
    }
    free(absent);
    return n_zero;
}

/* lookup a_ij by binary search in a sorted row; 0 when absent */
static double csr_get(const int64_t *ip, const int32_t *ix, const double *a, int64_t i, int32_t j)
{
    int64_t lo = ip[i], hi = ip[i + 1];
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (ix[mid] < j) lo = mid + 1; else hi = mid;
    }
    return (lo < ip[i + 1] && ix[lo] == j) ? a[lo] : 0.0;
}

/* Sparse restatement of is_hermitian (mzd/sparse_utils.py:10-18): any |m_ij - m_ji| >= tol ? */
static int is_symmetric(int64_t n, const int64_t *ip, const int32_t *ix, const double *a, double tol)
{
    int bad = 0;
#pragma omp parallel for schedule(static) num_threads(g_threads) reduction(| : bad) if (g_threads > 1)
    for (int64_t i = 0; i < n; i++) {
        if (bad) continue;
        for (int64_t k = ip[i]; k < ip[i + 1]; k++) {
            double t = csr_get(ip, ix, a, ix[k], (int32_t)i);
            if (fabs(a[k] - t) >= tol) { bad = 1; break; }
        }
    }
    return !bad;
}

/*
 * Knight-Ruiz: mzd/sparse_utils.py:90-224, statement for statement.  Returns ORC_OK, ORC_E_NAN
 * (:192-193) or ORC_E_NOCONV (:213-214); st receives what the caller needs to re-emit the
 * reference's warnings (:113, :121, :220-221) and branch statistics for the tests.
 */
int orc_kr_balance(int64_t n, const int64_t *ip, const int32_t *ix, const double *a_orig, double tol, double delta,
                   double Delta, int64_t max_iter, double *x, orc_kr_stats *st)
{
    int64_t *wip; int32_t *wix; double *wa;
    memset(st, 0, sizeof(*st));
    st->n_zero_diag = build_working(n, ip, ix, a_orig, &wip, &wix, &wa);        /* :110-115 */
    st->is_symmetric = is_symmetric(n, wip, wix, wa, tol);                       /* :120-121 */

    double *v = malloc(sizeof(double) * n), *rk = malloc(sizeof(double) * n), *y = malloc(sizeof(double) * n);
    double *ynew = malloc(sizeof(double) * n), *p = malloc(sizeof(double) * n), *Z = malloc(sizeof(double) * n);
    double *w = malloc(sizeof(double) * n), *ap = malloc(sizeof(double) * n), *t = malloc(sizeof(double) * n);
    int rc = ORC_OK;

    const double g = 0.9, etamax = 0.1;                                          /* :129-130 */
    double eta = etamax;
    const double stop_tol = tol * 0.5;                                           /* :132 */
    for (int64_t i = 0; i < n; i++) x[i] = 1.0;                                  /* :126-127,134 */
    const double rt = tol * tol;                                                 /* :135 */
    csr_matvec(n, wip, wix, wa, x, t);
    for (int64_t i = 0; i < n; i++) v[i] = x[i] * t[i];                          /* :136 */
    for (int64_t i = 0; i < n; i++) rk[i] = 1.0 - v[i];                          /* :138 */
    double rho_km1 = dot(n, rk, rk), rho_km2 = 0.0;                              /* :139 */
    double rout = rho_km1, rold = rout;                                          /* :140-141 */
    int64_t n_iter = 0;

    while (rout > rt && n_iter < max_iter) {                                     /* :146 */
        int64_t k = 0;
        for (int64_t i = 0; i < n; i++) y[i] = 1.0;                              /* :150 */
        double inner_tol = fmax(rout * (eta * eta), rt);                         /* :152 */
        st->n_outer++;

        while (rho_km1 > inner_tol) {                                            /* :154 */
            k++;
            if (k == 1) {
                for (int64_t i = 0; i < n; i++) { Z[i] = rk[i] / v[i]; p[i] = Z[i]; }   /* :158-159 */
                rho_km1 = dot(n, rk, Z);                                         /* :160 */
            } else {
                double beta = rho_km1 / rho_km2;                                 /* :162 */
                for (int64_t i = 0; i < n; i++) p[i] = Z[i] + beta * p[i];       /* :163 */
            }
            for (int64_t i = 0; i < n; i++) t[i] = x[i] * p[i];
            csr_matvec(n, wip, wix, wa, t, w);
            for (int64_t i = 0; i < n; i++) w[i] = x[i] * w[i] + v[i] * p[i];    /* :165 */
            double alpha = rho_km1 / dot(n, p, w);                               /* :166 */
            double mn = INFINITY, mx = -INFINITY;
            for (int64_t i = 0; i < n; i++) {
                ap[i] = alpha * p[i];                                            /* :167 */
                ynew[i] = y[i] + ap[i];                                          /* :169 */
                if (ynew[i] < mn) mn = ynew[i];
                if (ynew[i] > mx) mx = ynew[i];
            }
            if (mn <= delta) {                                                   /* :171 */
                st->hit_delta++;
                if (delta == 0) break;                                           /* :172-173 */
                double gamma = INFINITY;
                for (int64_t i = 0; i < n; i++)
                    if (ap[i] < 0) { double q = (delta - y[i]) / ap[i]; if (q < gamma) gamma = q; }   /* :174-175 */
                for (int64_t i = 0; i < n; i++) y[i] += gamma * ap[i];           /* :176 */
                break;
            }
            if (mx >= Delta) {                                                   /* :179 */
                st->hit_Delta++;
                double gamma = INFINITY;
                for (int64_t i = 0; i < n; i++)
                    if (ynew[i] > Delta) { double q = (Delta - y[i]) / ap[i]; if (q < gamma) gamma = q; }   /* :180-181 */
                for (int64_t i = 0; i < n; i++) y[i] += gamma * ap[i];           /* :182 */
                break;
            }
            for (int64_t i = 0; i < n; i++) {
                y[i] = ynew[i];                                                  /* :185 */
                rk[i] = rk[i] - alpha * w[i];                                    /* :186 */
            }
            rho_km2 = rho_km1;                                                   /* :187 */
            for (int64_t i = 0; i < n; i++) Z[i] = rk[i] * v[i];                 /* :189 (multiply, sic) */
            rho_km1 = dot(n, rk, Z);                                             /* :190 */
            for (int64_t i = 0; i < n; i++)
                if (isnan(x[i])) { rc = ORC_E_NAN; goto done; }                  /* :192-193 */
        }

        for (int64_t i = 0; i < n; i++) x[i] *= y[i];                            /* :195 */
        csr_matvec(n, wip, wix, wa, x, t);
        for (int64_t i = 0; i < n; i++) { v[i] = x[i] * t[i]; rk[i] = 1.0 - v[i]; }   /* :196-198 */
        rho_km1 = dot(n, rk, rk);                                                /* :199 */
        rout = rho_km1;
        n_iter += k + 1;                                                         /* :201 */

        double rat = rout / rold;                                                /* :203 */
        rold = rout;
        double res_norm = sqrt(rout);
        double eta_o = eta;
        eta = g * rat;                                                           /* :207 */
        if (g * (eta_o * eta_o) > 0.1) eta = fmax(eta, g * (eta_o * eta_o));      /* :209-210 */
        eta = fmax(fmin(eta, etamax), stop_tol / res_norm);                      /* :211 */
    }
    if (n_iter > max_iter) rc = ORC_E_NOCONV;                                    /* :213-214 */
    st->hit_max_iter = (n_iter >= max_iter);                                     /* :220-221 */

done:
    st->n_iter = n_iter;
    st->rout = rout;
    free(v); free(rk); free(y); free(ynew); free(p); free(Z); free(w); free(ap); free(t);
    free(wip); free(wix); free(wa);
    return rc;
}

/* X.T * (_orig * X): mzd/sparse_utils.py:223-224 -> out_ij = x_i * (a_ij * x_j) on the ORIGINAL matrix */
void orc_scale_sym(int64_t n, const int64_t *ip, const int32_t *ix, const double *a, const double *x, double *out)
{
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
    for (int64_t i = 0; i < n; i++)
        for (int64_t k = ip[i]; k < ip[i + 1]; k++) out[k] = x[i] * (a[k] * x[ix[k]]);
}

/* fast_norm_fullseq_bysite, mzd/contact_map.py:100-113, with the sites==0 -> 1 rule of :1103-1108 */
void orc_site_norm(int64_t n, const int64_t *ip, const int32_t *ix, const double *sites, double *a)
{
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
    for (int64_t i = 0; i < n; i++) {
        double si = sites[i] == 0.0 ? 1.0 : sites[i];
        for (int64_t k = ip[i]; k < ip[i + 1]; k++) {
            double sj = sites[ix[k]] == 0.0 ? 1.0 : sites[ix[k]];
            a[k] *= 1.0 / (si * sj);
        }
    }
}

/* max_offdiag, mzd/sparse_utils.py:269-281: setdiag(0) then column max (axis=0); implicit zeros
 * make the floor 0. */
void orc_max_offdiag_u32(int64_t n, const int64_t *ip, const int32_t *ix, const uint32_t *a, uint32_t *out)
{
    for (int64_t j = 0; j < n; j++) out[j] = 0;
    if (g_threads <= 1) {
        for (int64_t i = 0; i < n; i++)
            for (int64_t k = ip[i]; k < ip[i + 1]; k++)
                if (ix[k] != i && a[k] > out[ix[k]]) out[ix[k]] = a[k];
        return;
    }
    /* threads: rows in parallel, the column maximum raised by compare-and-swap (a maximum does not depend on the order) */
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int64_t i = 0; i < n; i++)
        for (int64_t k = ip[i]; k < ip[i + 1]; k++) {
            if (ix[k] == i) continue;
            uint32_t cur = __atomic_load_n(&out[ix[k]], __ATOMIC_RELAXED);
            while (a[k] > cur && !__atomic_compare_exchange_n(&out[ix[k]], &cur, a[k], 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
        }
}

/* compress, mzd/sparse_utils.py:284-314: keep (i,j) with mask[i] && mask[j], re-index by
 * i - cumsum(~mask)[i].  Pass out_ix == NULL to count only.  Returns the new dimension. */
int64_t orc_compress(int64_t n, const int64_t *ip, const int32_t *ix, const double *a, const uint8_t *mask,
                     int64_t *out_ip, int32_t *out_ix, double *out_a)
{
    int64_t *newid = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    int64_t *kept = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    int64_t shift = 0, m = 0;
    for (int64_t i = 0; i < n; i++) {
        if (!mask[i]) shift++;
        newid[i] = i - shift;                                                    /* :307,310-312 */
        if (mask[i]) m++;
    }
    /* kept entries per kept row (row-parallel), then the row pointers */
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
    for (int64_t i = 0; i < n; i++) {
        int64_t c = 0;
        if (mask[i])
            for (int64_t k = ip[i]; k < ip[i + 1]; k++) c += mask[ix[k]] != 0;   /* :302 */
        kept[i] = c;
    }
    int64_t o = 0;
    for (int64_t i = 0; i < n; i++)
        if (mask[i]) { out_ip[newid[i]] = o; o += kept[i]; }
    out_ip[m] = o;
    if (out_ix) {
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
        for (int64_t i = 0; i < n; i++) {
            if (!mask[i]) continue;
            int64_t w = out_ip[newid[i]];
            for (int64_t k = ip[i]; k < ip[i + 1]; k++)
                if (mask[ix[k]]) { out_ix[w] = (int32_t)newid[ix[k]]; out_a[w] = a[k]; w++; }
        }
    }
    free(newid);
    free(kept);
    return m;
}

/* _map.max() of mzd/cluster.py:316 (includes the diagonal; implicit zeros floor it at 0 when the
 * matrix is not completely dense) */
double orc_max_value(int64_t n, const int64_t *ip, const double *a)
{
    double mx = -INFINITY;
    for (int64_t k = 0; k < ip[n]; k++) if (a[k] > mx) mx = a[k];
    if ((double)ip[n] < (double)n * (double)n && mx < 0.0) mx = 0.0;
    return mx;
}
