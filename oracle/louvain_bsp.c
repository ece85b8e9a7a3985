/*
 * ORACLE (test infrastructure, not product code): the deterministic bulk-synchronous Louvain
 * schedule that the HIP kernels run, restated as a scalar loop.  PARITY UNPINNED with respect to
 * python-louvain (see louvain_seq.c header): sequential Louvain is order dependent and cannot be
 * reproduced by a data-parallel sweep, so the spec is this file; the GPU must match it bit for bit
 * and its modularity is reported next to orc_louvain_seq's.
 *
 * What is kept from python-louvain 0.11 (community_louvain.py; call site mzd/louvain.py:70):
 *   - gain of moving i into c:  dnc(c) - tot(c) * k_i / 2m  with i removed from its own community,
 *     candidates = communities of i's neighbours, baseline 0 for "stay" (best_increase = 0), strict >
 *   - sweep until nothing moves or a sweep gains < 1e-7 modularity; levels until a level gains < 1e-7
 *   - induced graph: community self-loop = internal edges once + member loops
 * What differs (needed for a parallel, reproducible schedule):
 *   - weights are fixed point (orc_quantise) and every comparison is exact integer arithmetic, so
 *     no result depends on summation order
 *   - a sweep = n_sub sub-rounds; nodes of sub-round r = {i : bucket(i) == r} decide simultaneously
 *     from the state at the start of the sub-round, then all moves are applied
 *   - ties -> lowest community id; a singleton never joins a higher-numbered singleton
 *   - communities are renumbered by ascending id between levels
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <omp.h>

#include "oracle.h"

typedef __int128 i128;
#define PL_MIN 0.0000001

/* smallest S such that quantised totals stay below 2^61: see DESIGN.md "fixed-point weights" */
int32_t orc_quant_shift(int64_t n, const int64_t *ip, const int32_t *ix, const double *w)
{
    double wmax = 0.0;
    for (int64_t k = 0; k < ip[n]; k++) if (w[k] > wmax) wmax = w[k];
    if (!(wmax > 0.0)) return 0;
    int e;
    frexp(wmax, &e); /* wmax < 2^e */
    uint64_t total = 0;
    for (int64_t i = 0; i < n; i++)
        for (int64_t k = ip[i]; k < ip[i + 1]; k++) {
            uint64_t c = (uint64_t)ceil(ldexp(w[k], 20 - e));
            total += (ix[k] == i) ? 2 * c : c;
        }
    int bits = 0;
    while (bits < 64 && (total >> bits)) bits++; /* total < 2^bits */
    return 61 - (bits - 20 + e);
}

void orc_quantise(int64_t nnz, const double *w, int32_t shift, int64_t *q)
{
    for (int64_t k = 0; k < nnz; k++) q[k] = llrint(ldexp(w[k], shift));
}

static inline uint32_t bucket(int64_t i, int32_t n_sub)
{
    uint32_t h = (uint32_t)i * 2654435761u;
    return (h >> 16) % (uint32_t)n_sub;
}

typedef struct { int64_t n; int64_t *ip; int32_t *ix; int64_t *q; } qgraph_t;

static double exact_modularity(const qgraph_t *g, const int32_t *com, const int64_t *tot, int64_t m2)
{
    /* Q = sum_c I2_c / M2 - sum_c tot_c^2 / M2^2, all integers exact */
    i128 i2 = 0, t2 = 0;
    for (int64_t i = 0; i < g->n; i++)
        for (int64_t k = g->ip[i]; k < g->ip[i + 1]; k++) {
            int64_t j = g->ix[k];
            if (j == i) i2 += 2 * (i128)g->q[k];
            else if (com[j] == com[i]) i2 += g->q[k];
        }
    for (int64_t c = 0; c < g->n; c++) t2 += (i128)tot[c] * tot[c];
    i128 num = (i128)m2 * i2 - t2;
    return (double)num / ((double)m2 * (double)m2);
}

typedef struct { int64_t key; int64_t q; } qagg_t;
static int qagg_cmp(const void *a, const void *b)
{
    const qagg_t *x = a, *y = b;
    return x->key < y->key ? -1 : (x->key > y->key);
}

int orc_louvain_bsp(int64_t n, const int64_t *ip, const int32_t *ix, const int64_t *q, int32_t n_sub,
                    int32_t max_sweeps, int32_t *labels, int32_t *n_comm, double *q_out, int32_t *n_levels,
                    int64_t *n_sweeps)
{
    qgraph_t g;
    g.n = n;
    g.ip = malloc(sizeof(int64_t) * (n + 1)); memcpy(g.ip, ip, sizeof(int64_t) * (n + 1));
    g.ix = malloc(sizeof(int32_t) * (ip[n] > 0 ? ip[n] : 1)); memcpy(g.ix, ix, sizeof(int32_t) * ip[n]);
    g.q = malloc(sizeof(int64_t) * (ip[n] > 0 ? ip[n] : 1)); memcpy(g.q, q, sizeof(int64_t) * ip[n]);
    *n_levels = 0; *n_sweeps = 0; *q_out = 0.0; *n_comm = 0;
    for (int64_t i = 0; i < n; i++) labels[i] = ip[i + 1] > ip[i] ? (int32_t)i : -1;

    double mod = 0.0;
    int first = 1;
    for (;;) {
        int64_t nn = g.n;
        int64_t *kdeg = calloc(nn > 0 ? nn : 1, sizeof(int64_t)), *tot = calloc(nn > 0 ? nn : 1, sizeof(int64_t));
        int32_t *com = malloc(sizeof(int32_t) * (nn > 0 ? nn : 1)), *nxt = malloc(sizeof(int32_t) * (nn > 0 ? nn : 1));
        int32_t *csize = malloc(sizeof(int32_t) * (nn > 0 ? nn : 1));
        int64_t *dnc = malloc(sizeof(int64_t) * (nn > 0 ? nn : 1));
        int32_t *touched = malloc(sizeof(int32_t) * (nn > 0 ? nn : 1));
        int64_t m2 = 0;
        for (int64_t i = 0; i < nn; i++) {
            int64_t d = 0;
            for (int64_t k = g.ip[i]; k < g.ip[i + 1]; k++) d += (g.ix[k] == i) ? 2 * g.q[k] : g.q[k];
            kdeg[i] = d; tot[i] = d; com[i] = (int32_t)i; dnc[i] = -1;
            csize[i] = g.ip[i + 1] > g.ip[i] ? 1 : 0; /* empty rows are not in the graph (add_edge never saw them) */
            m2 += d;
        }
        if (m2 == 0) { /* no links: every node alone */
            free(kdeg); free(tot); free(com); free(nxt); free(csize); free(dnc); free(touched);
            break;
        }
        /* per-thread scratch of the threaded decide loop (orc_set_threads; one thread: the arrays above) */
        const int nthr = nn >= 4096 ? orc_get_threads() : 1;
        int64_t *tdnc = NULL;
        int32_t *ttouched = NULL;
        if (nthr > 1) {
            tdnc = malloc(sizeof(int64_t) * (size_t)nthr * (size_t)nn);
            ttouched = malloc(sizeof(int32_t) * (size_t)nthr * (size_t)nn);
            for (size_t c = 0; c < (size_t)nthr * (size_t)nn; c++) tdnc[c] = -1;
        }
        double cur_mod, new_mod = exact_modularity(&g, com, tot, m2);
        int32_t sweep = 0;
        for (;;) {
            cur_mod = new_mod;
            int64_t moved = 0;
            sweep++; (*n_sweeps)++;
            for (int32_t r = 0; r < n_sub; r++) {
                /* decide from the snapshot: the decisions of a sub-round do not depend on one another, so the nodes may be
                 * taken in any order -- with threads (orc_set_threads, the multi-thread CPU baseline) in parallel, each
                 * thread with its own neighbour-community scratch; the result is the same bit for bit */
#pragma omp parallel num_threads(nthr) if (nthr > 1)
                {
                int64_t *my_dnc = dnc;
                int32_t *my_touched = touched;
                if (nthr > 1) {
                    my_dnc = tdnc + (size_t)omp_get_thread_num() * (size_t)nn;
                    my_touched = ttouched + (size_t)omp_get_thread_num() * (size_t)nn;
                }
#pragma omp for schedule(dynamic, 1024)
                for (int64_t i = 0; i < nn; i++) {
                    nxt[i] = com[i];
                    if (bucket(i, n_sub) != (uint32_t)r || kdeg[i] == 0) continue;
                    int32_t a = com[i];
                    int64_t nt = 0;
                    for (int64_t k = g.ip[i]; k < g.ip[i + 1]; k++) {
                        int64_t j = g.ix[k];
                        if (j == i) continue;
                        int32_t c = com[j];
                        if (my_dnc[c] < 0) { my_dnc[c] = 0; my_touched[nt++] = c; }
                        my_dnc[c] += g.q[k];
                    }
                    int32_t best = a;
                    i128 best_gain = 0;
                    int have = 0;
                    for (int64_t t = 0; t < nt; t++) {
                        int32_t c = my_touched[t];
                        int64_t tc = tot[c] - (c == a ? kdeg[i] : 0);
                        i128 gain = (i128)my_dnc[c] * m2 - (i128)tc * kdeg[i];
                        if (gain > best_gain || (have && gain == best_gain && c < best)) {
                            best_gain = gain; best = c; have = 1;
                        }
                        my_dnc[c] = -1;
                    }
                    if (best != a && csize[a] == 1 && csize[best] == 1 && best > a) best = a;
                    nxt[i] = best;
                }
                }
                /* apply */
                for (int64_t i = 0; i < nn; i++) {
                    if (nxt[i] == com[i]) continue;
                    tot[com[i]] -= kdeg[i]; csize[com[i]]--;
                    tot[nxt[i]] += kdeg[i]; csize[nxt[i]]++;
                    com[i] = nxt[i];
                    moved++;
                }
            }
            if (moved == 0) break;
            new_mod = exact_modularity(&g, com, tot, m2);
            if (new_mod - cur_mod < PL_MIN) break;
            if (max_sweeps > 0 && sweep >= max_sweeps) break;
        }
        free(tdnc); free(ttouched);
        if (!first && new_mod - mod < PL_MIN) {
            free(kdeg); free(tot); free(com); free(nxt); free(csize); free(dnc); free(touched);
            break;
        }
        first = 0;
        mod = new_mod;
        (*n_levels)++;
        /* renumber by ascending community id */
        int32_t *newid = malloc(sizeof(int32_t) * (nn > 0 ? nn : 1));
        int64_t kc = 0;
        for (int64_t c = 0; c < nn; c++) newid[c] = csize[c] > 0 ? (int32_t)kc++ : -1;
        for (int64_t i = 0; i < n; i++) if (labels[i] >= 0) labels[i] = newid[com[labels[i]]];
        /* induced graph, exact */
        int64_t m = g.ip[nn];
        qagg_t *ed = malloc(sizeof(qagg_t) * (m > 0 ? m : 1));
        for (int64_t i = 0; i < nn; i++)
            for (int64_t k = g.ip[i]; k < g.ip[i + 1]; k++) {
                int64_t c1 = newid[com[i]], c2 = newid[com[g.ix[k]]];
                ed[k].key = c1 * kc + c2;
                /* loops count twice so that the internal total is 2*L and halves exactly */
                ed[k].q = (g.ix[k] == i) ? 2 * g.q[k] : g.q[k];
            }
        qsort(ed, m, sizeof(qagg_t), qagg_cmp);
        int64_t nu = 0;
        for (int64_t e = 0; e < m; e++) if (e == 0 || ed[e].key != ed[e - 1].key) nu++;
        qgraph_t h;
        h.n = kc;
        h.ip = calloc(kc + 1, sizeof(int64_t));
        h.ix = malloc(sizeof(int32_t) * (nu > 0 ? nu : 1));
        h.q = malloc(sizeof(int64_t) * (nu > 0 ? nu : 1));
        int64_t o = 0;
        for (int64_t e = 0; e < m;) {
            int64_t f = e, s = 0;
            while (f < m && ed[f].key == ed[e].key) s += ed[f++].q;
            int64_t c1 = ed[e].key / kc, c2 = ed[e].key % kc;
            h.ix[o] = (int32_t)c2;
            h.q[o] = (c1 == c2) ? s / 2 : s;
            o++;
            h.ip[c1 + 1]++;
            e = f;
        }
        for (int64_t c = 0; c < kc; c++) h.ip[c + 1] += h.ip[c];
        free(ed); free(newid);
        free(kdeg); free(tot); free(com); free(nxt); free(csize); free(dnc); free(touched);
        free(g.ip); free(g.ix); free(g.q);
        g = h;
    }
    *q_out = mod;
    int32_t mx = -1;
    for (int64_t i = 0; i < n; i++) if (labels[i] > mx) mx = labels[i];
    *n_comm = mx + 1;
    free(g.ip); free(g.ix); free(g.q);
    return ORC_OK;
}
