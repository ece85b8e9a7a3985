/*
 * ORACLE (test infrastructure, not product code): sequential Louvain as python-louvain 0.11 does it.
 *
 * The reference's Louvain arithmetic is a third-party dependency that is NOT in /root/reference:
 * `import community as com` (mzd/louvain.py:2), `com.best_partition(g)` (mzd/louvain.py:21,70), pinned
 * python-louvain==0.11 on networkx<2 (Pipfile.lock).  This file restates that package's PUBLISHED
 * algorithm (community/community_louvain.py: best_partition -> generate_dendrogram -> __one_level /
 * __modularity / __renumber / induced_graph, class Status) from its documented behaviour.
 *
 * PARITY UNPINNED: neither python-louvain nor networkx 1.x is installed here, the reference holds
 * no test or golden vector for this call, and the reference's own caller is broken downstream
 * (mzd/cluster.py:208-209).  The choices that Python-2 dict ordering would make are fixed here as:
 * nodes ascending, neighbours in CSR (ascending id) order, candidate communities ascending id.
 *
 * Input graph = what to_graph builds (mzd/cluster.py:316-321): see orc_graph_weights.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

#define PL_MIN 0.0000001 /* community_louvain.__MIN */

/*
 * to_graph's edge weights (mzd/cluster.py:316-321): g.add_edge(u, v, weight=w*scl) for every stored
 * (u,v) in row-major COO order, a later (v,u) overwriting an earlier (u,v).  For u < v the LOWER
 * triangle entry (row v) is visited last, so both directions carry fl(a_vu * scl).
 */
void orc_graph_weights(int64_t n, const int64_t *ip, const int32_t *ix, const double *a, double scl, double *w)
{
    for (int64_t i = 0; i < n; i++)
        for (int64_t k = ip[i]; k < ip[i + 1]; k++) {
            int64_t j = ix[k];
            if (j <= i) { w[k] = a[k] * scl; continue; }
            /* find (j, i) */
            int64_t lo = ip[j], hi = ip[j + 1];
            while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (ix[mid] < i) lo = mid + 1; else hi = mid; }
            w[k] = (lo < ip[j + 1] && ix[lo] == i) ? a[lo] * scl : a[k] * scl;
        }
}

typedef struct {
    int64_t n;
    int64_t *ip;
    int32_t *ix;
    double *w;
    uint8_t *present; /* node is in the graph (has at least one stored entry / is an induced node) */
} graph_t;

typedef struct {
    int32_t *node2com;
    double total_weight;
    double *degrees, *gdegrees, *internals, *loops;
} status_t;

static void status_init(status_t *s, const graph_t *g)
{
    int64_t n = g->n;
    s->node2com = malloc(sizeof(int32_t) * n);
    s->degrees = calloc(n, sizeof(double));
    s->gdegrees = calloc(n, sizeof(double));
    s->internals = calloc(n, sizeof(double));
    s->loops = calloc(n, sizeof(double));
    double degsum = 0.0;
    for (int64_t i = 0; i < n; i++) {
        s->node2com[i] = -1;
        if (!g->present[i]) continue;
        /* Graph.degree(node, weight): sum over neighbours, self-loop added once more */
        double d = 0.0, loop = 0.0;
        for (int64_t k = g->ip[i]; k < g->ip[i + 1]; k++) {
            d += g->w[k];
            if (g->ix[k] == i) loop = g->w[k];
        }
        d += loop;
        s->node2com[i] = (int32_t)i; /* Status.init: count over nodes; with all ids present count == i */
        s->degrees[i] = d;
        s->gdegrees[i] = d;
        s->loops[i] = loop;
        s->internals[i] = loop;
        degsum += d;
    }
    s->total_weight = degsum / 2.0; /* Graph.size(weight) = sum(degree)/2 */
}

static void status_free(status_t *s)
{
    free(s->node2com); free(s->degrees); free(s->gdegrees); free(s->internals); free(s->loops);
}

/* __modularity(status) */
static double status_modularity(const status_t *s, int64_t n, uint8_t *seen)
{
    double links = s->total_weight, result = 0.0;
    memset(seen, 0, n);
    for (int64_t i = 0; i < n; i++) if (s->node2com[i] >= 0) seen[s->node2com[i]] = 1;
    for (int64_t c = 0; c < n; c++) {
        if (!seen[c]) continue;
        if (links > 0) {
            double t = s->degrees[c] / (2.0 * links);
            result += s->internals[c] / links - t * t;
        }
    }
    return result;
}

/* __one_level: sweeps until no node moves or the modularity gain of a sweep drops under __MIN */
static int64_t one_level(const graph_t *g, status_t *s, double *nw, int32_t *touched, uint8_t *seen)
{
    int64_t n = g->n, sweeps = 0;
    int modified = 1;
    double cur_mod, new_mod = status_modularity(s, n, seen);
    while (modified) {
        cur_mod = new_mod;
        modified = 0;
        sweeps++;
        for (int64_t node = 0; node < n; node++) {
            if (!g->present[node]) continue;
            int32_t com_node = s->node2com[node];
            double degc_totw = s->gdegrees[node] / (s->total_weight * 2.0);
            /* __neighcom */
            int64_t nt = 0;
            for (int64_t k = g->ip[node]; k < g->ip[node + 1]; k++) {
                int32_t nb = g->ix[k];
                if (nb == node) continue;
                int32_t c = s->node2com[nb];
                if (nw[c] < 0) { nw[c] = 0.0; touched[nt++] = c; }
                nw[c] += g->w[k];
            }
            /* __remove */
            double wown = nw[com_node] < 0 ? 0.0 : nw[com_node];
            s->degrees[com_node] -= s->gdegrees[node];
            s->internals[com_node] = s->internals[com_node] - wown - s->loops[node];
            s->node2com[node] = -1;
            int32_t best_com = com_node;
            double best_inc = 0.0;
            int have = 0;
            for (int64_t t = 0; t < nt; t++) {
                int32_t c = touched[t];
                double incr = nw[c] - s->degrees[c] * degc_totw; /* resolution = 1 */
                if (incr > best_inc || (have && incr == best_inc && c < best_com)) {
                    best_inc = incr; best_com = c; have = 1;
                }
            }
            /* __insert */
            double wbest = nw[best_com] < 0 ? 0.0 : nw[best_com];
            s->node2com[node] = best_com;
            s->degrees[best_com] += s->gdegrees[node];
            s->internals[best_com] = s->internals[best_com] + wbest + s->loops[node];
            if (best_com != com_node) modified = 1;
            for (int64_t t = 0; t < nt; t++) nw[touched[t]] = -1.0;
        }
        new_mod = status_modularity(s, n, seen);
        if (new_mod - cur_mod < PL_MIN) break;
    }
    return sweeps;
}

typedef struct { int64_t key; int64_t seq; double w; } agg_t;
static int agg_cmp(const void *a, const void *b)
{
    const agg_t *x = a, *y = b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->seq < y->seq ? -1 : (x->seq > y->seq);
}

/* induced_graph(partition, graph): one node per community, edge weights summed edge by edge in
 * edges_iter order (each undirected edge once, reported from its lower endpoint) */
static void induced_graph(const graph_t *g, const int32_t *part, int64_t k, graph_t *out)
{
    int64_t n = g->n, m = 0;
    for (int64_t u = 0; u < n; u++)
        for (int64_t e = g->ip[u]; e < g->ip[u + 1]; e++) if (g->ix[e] >= u) m++;
    agg_t *ed = malloc(sizeof(agg_t) * (m > 0 ? m : 1));
    int64_t o = 0;
    for (int64_t u = 0; u < n; u++)
        for (int64_t e = g->ip[u]; e < g->ip[u + 1]; e++) {
            int64_t v = g->ix[e];
            if (v < u) continue;
            int64_t c1 = part[u], c2 = part[v];
            int64_t lo = c1 < c2 ? c1 : c2, hi = c1 < c2 ? c2 : c1;
            ed[o].key = lo * k + hi; ed[o].seq = o; ed[o].w = g->w[e]; o++;
        }
    qsort(ed, m, sizeof(agg_t), agg_cmp);
    /* unique pairs */
    int64_t nu = 0;
    for (int64_t i = 0; i < m; i++) if (i == 0 || ed[i].key != ed[i - 1].key) nu++;
    int64_t *cnt = calloc(k + 1, sizeof(int64_t));
    for (int64_t i = 0; i < m; i++) {
        if (i && ed[i].key == ed[i - 1].key) continue;
        int64_t lo = ed[i].key / k, hi = ed[i].key % k;
        cnt[lo + 1]++;
        if (hi != lo) cnt[hi + 1]++;
    }
    out->n = k;
    out->ip = malloc(sizeof(int64_t) * (k + 1));
    out->ip[0] = 0;
    for (int64_t c = 0; c < k; c++) out->ip[c + 1] = out->ip[c] + cnt[c + 1];
    int64_t nnz = out->ip[k];
    out->ix = malloc(sizeof(int32_t) * (nnz > 0 ? nnz : 1));
    out->w = malloc(sizeof(double) * (nnz > 0 ? nnz : 1));
    out->present = malloc(k > 0 ? k : 1);
    memset(out->present, 1, k);
    int64_t *fill = malloc(sizeof(int64_t) * (k + 1));
    memcpy(fill, out->ip, sizeof(int64_t) * (k + 1));
    /* keys ascend by (lo, hi): filling rows lo with hi ascending keeps columns sorted only for the
     * upper part; do two passes so every row ends up sorted: first lower-triangle mirrors. */
    for (int pass = 0; pass < 2; pass++) {
        int64_t i = 0;
        while (i < m) {
            int64_t j = i;
            double wsum = 0.0; /* w_prec + edge_weight, starting from the {weight: 0} default */
            while (j < m && ed[j].key == ed[i].key) { wsum = wsum + ed[j].w; j++; }
            int64_t lo = ed[i].key / k, hi = ed[i].key % k;
            if (pass == 0 && hi != lo) { out->ix[fill[hi]] = (int32_t)lo; out->w[fill[hi]++] = wsum; }
            if (pass == 1) { out->ix[fill[lo]] = (int32_t)hi; out->w[fill[lo]++] = wsum; }
            i = j;
        }
    }
    (void)nu;
    free(fill); free(cnt); free(ed);
}

static void graph_free(graph_t *g) { free(g->ip); free(g->ix); free(g->w); free(g->present); }

/* __renumber: communities numbered by first appearance in node order */
static int64_t renumber(const int32_t *node2com, int64_t n, int32_t *part, int32_t *newval)
{
    int64_t count = 0;
    for (int64_t i = 0; i < n; i++) newval[i] = -1;
    for (int64_t i = 0; i < n; i++) {
        if (node2com[i] < 0) { part[i] = -1; continue; }
        if (newval[node2com[i]] == -1) newval[node2com[i]] = (int32_t)count++;
        part[i] = newval[node2com[i]];
    }
    return count;
}

int orc_louvain_seq(int64_t n, const int64_t *ip, const int32_t *ix, const double *w, int32_t *labels,
                    int32_t *n_comm, double *q_out, int32_t *n_levels, int64_t *n_sweeps)
{
    graph_t cur;
    cur.n = n;
    cur.ip = malloc(sizeof(int64_t) * (n + 1)); memcpy(cur.ip, ip, sizeof(int64_t) * (n + 1));
    cur.ix = malloc(sizeof(int32_t) * (ip[n] > 0 ? ip[n] : 1)); memcpy(cur.ix, ix, sizeof(int32_t) * ip[n]);
    cur.w = malloc(sizeof(double) * (ip[n] > 0 ? ip[n] : 1)); memcpy(cur.w, w, sizeof(double) * ip[n]);
    cur.present = malloc(n > 0 ? n : 1);
    int64_t n_present = 0;
    for (int64_t i = 0; i < n; i++) { cur.present[i] = ip[i + 1] > ip[i]; n_present += cur.present[i]; }

    double *nw = malloc(sizeof(double) * (n > 0 ? n : 1));
    int32_t *touched = malloc(sizeof(int32_t) * (n > 0 ? n : 1));
    int32_t *part = malloc(sizeof(int32_t) * (n > 0 ? n : 1));
    int32_t *newval = malloc(sizeof(int32_t) * (n > 0 ? n : 1));
    uint8_t *seen = malloc(n > 0 ? n : 1);
    for (int64_t i = 0; i < n; i++) nw[i] = -1.0;
    *n_levels = 0; *n_sweeps = 0; *q_out = 0.0;

    /* generate_dendrogram special case: no links -> everyone alone (part[node] = node) */
    if (ip[n] == 0) {
        for (int64_t i = 0; i < n; i++) labels[i] = -1;
        *n_comm = 0;
        goto out;
    }
    for (int64_t i = 0; i < n; i++) labels[i] = cur.present[i] ? (int32_t)i : -1;

    status_t st;
    status_init(&st, &cur);
    *n_sweeps += one_level(&cur, &st, nw, touched, seen);
    double mod = status_modularity(&st, cur.n, seen);
    int64_t k = renumber(st.node2com, cur.n, part, newval);
    for (int64_t i = 0; i < n; i++) if (labels[i] >= 0) labels[i] = part[labels[i]];
    *n_levels = 1;
    for (;;) {
        graph_t nxt;
        induced_graph(&cur, part, k, &nxt);
        status_free(&st);
        graph_free(&cur);
        cur = nxt;
        status_init(&st, &cur);
        *n_sweeps += one_level(&cur, &st, nw, touched, seen);
        double new_mod = status_modularity(&st, cur.n, seen);
        if (new_mod - mod < PL_MIN) break;
        k = renumber(st.node2com, cur.n, part, newval);
        for (int64_t i = 0; i < n; i++) if (labels[i] >= 0) labels[i] = part[labels[i]];
        mod = new_mod;
        (*n_levels)++;
    }
    status_free(&st);
    *q_out = mod;
    {
        int32_t mx = -1;
        for (int64_t i = 0; i < n; i++) if (labels[i] > mx) mx = labels[i];
        *n_comm = mx + 1;
    }
out:
    graph_free(&cur);
    free(nw); free(touched); free(part); free(newval); free(seen);
    return ORC_OK;
}

/* community.modularity(partition, graph): Q of an arbitrary labelling (labels < 0 = absent) */
double orc_modularity(int64_t n, const int64_t *ip, const int32_t *ix, const double *w, const int32_t *labels)
{
    int32_t mx = -1;
    for (int64_t i = 0; i < n; i++) if (labels[i] > mx) mx = labels[i];
    if (mx < 0) return 0.0;
    double *inc = calloc(mx + 1, sizeof(double)), *deg = calloc(mx + 1, sizeof(double));
    double degsum = 0.0;
    for (int64_t i = 0; i < n; i++) {
        if (labels[i] < 0) continue;
        int32_t c = labels[i];
        for (int64_t k = ip[i]; k < ip[i + 1]; k++) {
            int64_t j = ix[k];
            double ww = w[k];
            deg[c] += (j == i) ? 2.0 * ww : ww;
            degsum += (j == i) ? 2.0 * ww : ww;
            if (labels[j] == c) inc[c] += (j == i) ? ww : ww / 2.0;
        }
    }
    double links = degsum / 2.0, res = 0.0;
    if (links > 0)
        for (int32_t c = 0; c <= mx; c++) {
            double t = deg[c] / (2.0 * links);
            res += inc[c] / links - t * t;
        }
    free(inc); free(deg);
    return res;
}
