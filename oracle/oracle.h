/*
 * ORACLE (test infrastructure, not product code) -- CPU restatement of the reference hot path.
 * See kr.c / louvain_seq.c / louvain_bsp.c for the reference lines each function follows.
 */
#ifndef B3C_ORACLE_H
#define B3C_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_OK 0
#define ORC_E_NAN 1
#define ORC_E_NOCONV 2
#define ORC_E_ARG 3

typedef struct {
    int64_t n_iter;
    int64_t n_outer;
    int64_t n_zero_diag;
    int32_t is_symmetric;
    int32_t hit_max_iter;
    int64_t hit_delta;
    int64_t hit_Delta;
    double rout;
} orc_kr_stats;

/* threads of the row-parallel loops (SpMV, site normalisation, final scaling); 1 = sequential (default) */
void orc_set_threads(int n);
int orc_get_threads(void);

int orc_kr_balance(int64_t n, const int64_t *ip, const int32_t *ix, const double *a, double tol, double delta,
                   double Delta, int64_t max_iter, double *x, orc_kr_stats *st);
void orc_scale_sym(int64_t n, const int64_t *ip, const int32_t *ix, const double *a, const double *x, double *out);
void orc_site_norm(int64_t n, const int64_t *ip, const int32_t *ix, const double *sites, double *a);
void orc_max_offdiag_u32(int64_t n, const int64_t *ip, const int32_t *ix, const uint32_t *a, uint32_t *out);
int64_t orc_compress(int64_t n, const int64_t *ip, const int32_t *ix, const double *a, const uint8_t *mask,
                     int64_t *out_ip, int32_t *out_ix, double *out_a);
double orc_max_value(int64_t n, const int64_t *ip, const double *a);

/* graph weights of to_graph (mzd/cluster.py:316-321) on a structurally symmetric CSR */
void orc_graph_weights(int64_t n, const int64_t *ip, const int32_t *ix, const double *a, double scl, double *w);

/* sequential python-louvain 0.11 restatement; labels[i] = -1 for nodes with an empty row */
int orc_louvain_seq(int64_t n, const int64_t *ip, const int32_t *ix, const double *w, int32_t *labels,
                    int32_t *n_comm, double *q_out, int32_t *n_levels, int64_t *n_sweeps);

/* fixed-point quantisation shared by the BSP spec */
int32_t orc_quant_shift(int64_t n, const int64_t *ip, const int32_t *ix, const double *w);
void orc_quantise(int64_t nnz, const double *w, int32_t shift, int64_t *q);

/* deterministic bulk-synchronous Louvain (the schedule the GPU runs) on fixed-point weights */
int orc_louvain_bsp(int64_t n, const int64_t *ip, const int32_t *ix, const int64_t *q, int32_t n_sub,
                    int32_t max_sweeps, int32_t *labels, int32_t *n_comm, double *q_out, int32_t *n_levels,
                    int64_t *n_sweeps);

/* modularity of an arbitrary labelling on fp64 weights (python-louvain `modularity`) */
double orc_modularity(int64_t n, const int64_t *ip, const int32_t *ix, const double *w, const int32_t *labels);

#ifdef __cplusplus
}
#endif
#endif
