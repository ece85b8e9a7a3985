/*
 * libb3c -- MI355X (gfx950) native hot path of bin3C: contact-map normalisation, Knight-Ruiz
 * balancing, mask compaction and Louvain clustering.  C ABI, plain pointers and sizes only.
 *
 * Every entry point names the reference interface it replaces (paths relative to cerebis/bin3C).
 * The reference is pure Python, so its "FFI" for this path is a ctypes binding: see INTEGRATION.md
 * for the stub a maintainer adds to mzd/sparse_utils.py, mzd/contact_map.py and mzd/cluster.py.
 *
 * Conventions
 *  - Matrices are canonical CSR: int64 indptr[n+1], int32 indices[nnz] sorted within a row and free
 *    of duplicates, values double (B3C_F64) or uint32 (B3C_U32).
 *  - Array arguments marked "host|dev" may be host or device pointers (resolved with
 *    hipMemcpyDefault); the library never retains or frees caller memory.
 *  - Every call returns B3C_OK or a negative error code; b3c_last_error() gives the text.  Calls are
 *    synchronous with respect to their outputs unless the name ends in _async.
 *  - One b3c_ctx per host thread and device; a ctx is not re-entrant.  A b3c_mat belongs to the context that created it and must
 *    not be passed to two calls at once: calls leave by-products with the (otherwise immutable) matrix -- its largest value, the
 *    largest off-diagonal count of every row -- which later calls answer from.
 *  - Completion: a call has finished all its work when it returns, with ONE exception: b3c_acceptance_mask and b3c_vec_permute
 *    with a DEVICE destination return once the result is enqueued on the context's stream (they are the per-contig plumbing
 *    between stages; a host destination is complete on return).  A caller that reads such a device result on another stream, or
 *    from the host, orders itself behind the context's stream first (an event on it, or b3c_ctx_sync).
 */
#ifndef B3C_H
#define B3C_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define B3C_OK 0
#define B3C_E_ARG (-1)    /* bad argument (AssertionError in the reference: sparse_utils.py:103-104) */
#define B3C_E_HIP (-2)    /* HIP runtime failure */
#define B3C_E_NAN (-3)    /* RuntimeError('scale vector has developed invalid values (NANs)!') sparse_utils.py:193 */
#define B3C_E_NOCONV (-4) /* RuntimeError('matrix balancing failed to converge in {} iterations') sparse_utils.py:214 */
#define B3C_E_NOMEM (-5)
#define B3C_E_RCCL (-6)   /* a collective of the row-sharded multi-GPU path failed (no reference counterpart) */

#define B3C_F64 0
#define B3C_U32 1

typedef struct b3c_ctx b3c_ctx;
typedef struct b3c_mat b3c_mat;

/* Outcome of b3c_kr_balance; the warn-only conditions let the caller re-emit the reference's log lines. */
typedef struct {
    int64_t n_iter;       /* sparse_utils.py:201,218 -- number of SpMVs minus one */
    int64_t n_outer;      /* outer (Newton) iterations */
    int64_t n_zero_diag;  /* 'treating {} zeros on diagonal as ones' sparse_utils.py:113 */
    int32_t is_symmetric; /* 0 -> 'input matrix is expected to be fully symmetric' sparse_utils.py:121 */
    int32_t hit_max_iter; /* 1 -> 'maximum number of iterations ({}) reached' sparse_utils.py:220-221 */
    int64_t hit_delta;    /* times the delta bound cut the inner loop (sparse_utils.py:171) */
    int64_t hit_Delta;    /* times the Delta bound cut the inner loop (sparse_utils.py:179) */
    double rout;          /* final squared residual */
    double spmv_ms;       /* device time spent in the SpMV kernel: mean over the launches bracketed by HIP events (every
                           * 4th one, B3C_PROFILE=1) times n_spmv; 0 if not measured */
    int64_t n_spmv;       /* SpMV launches that did work */
    int32_t bitwise_symmetric; /* 1: a_ij == a_ji exactly (checksum path); precondition of b3c_scale_sym_graph */
    int32_t value_bits;   /* width of the matrix values the SpMV streamed: 64 (fp64), 32 (counts) or 16 (counts <= 65535) */
    /* tile-local column dictionaries (16-bit counts): tiles of the SpMV, the largest dictionary, the sum of all
     * dictionary sizes and the entries outside the dictionaries (tails); all 0 for the other forms */
    int64_t tcd_tiles, tcd_dict_max, tcd_dict_total, tcd_tail_entries;
    /* tails summed by their own pass (kr_kb_tails, maps beyond 2^19 contigs on one GPU); 0: by the row kernel.  With
     * tcd_dict_total == 0 every stored entry below the hub length is such a tail word (the library's choice when the
     * dictionaries would hold under a fifth of the entries) */
    int64_t tcd_xt_tails;
    /* counts beyond 65535 handled by the correction list beside the 16-bit forms (0: none, or too many: 32-bit values) */
    int64_t big_counts;
    /* what ONE SpMV streams BY CONSTRUCTION in the layout this call ran on (this rank's rows), for the roofline: bytes of a
     * dictionary entry (4, or 3 as 24-bit words; 0: no dictionary form), bytes of a tail entry (8: count + column or a 64-bit
     * word; 4: packed words of the tail pass), tail words that stayed 64-bit beside the packed stream, and the total over all
     * streams of the product -- entry words, group bases, dictionaries (ids in, operands gathered), row tables, the n-vectors of
     * the fused epilogue, the fixed-point row sums of the tail pass (read, cleared, flushed) */
    int64_t tcd_word_bytes, xt_word_bytes, xt_residual_words, spmv_stream_bytes;
} b3c_kr_info;

typedef struct {
    int32_t n_comm;
    int32_t n_levels;
    int64_t n_sweeps;
    int64_t n_moves;
    int32_t quant_shift;  /* fixed-point scale 2^shift applied to the edge weights */
    int32_t levels_sharded; /* row-sharded runs: levels BEYOND the first whose sweeps were sharded over the ranks too (their induced graph
                             * is replicated; every rank scans a row range of it balanced by stored entries); 0 on one GPU */
    double modularity;    /* exact-integer modularity of the returned partition, as double */
    double move_ms;       /* device time in the delta-Q neighbour-scan kernel */
    int64_t n_move_launch;
    int64_t move_nnz;     /* stored entries scanned by those launches (for the roofline) */
    int64_t move_rows;    /* rows visited by those launches */
    int64_t n_comm_level0; /* communities level 0 ended with (later levels merge them: n_comm is the final count) */
    int64_t agg_gathered_runs; /* row blocks, levels with more communities than the bin table holds: (community pair,
                                * weight) runs gathered over the ranks for the second reduction; 0 otherwise */
    double first_sweep_ms;     /* the part of move_ms spent in the FIRST sweep of level 0: its n_sub launches scan every row
                                * (no node can be settled without its row yet), the honest figure for a roofline */
    int64_t first_sweep_launches;
    int64_t agg_nnz;      /* stored entries of the graphs that were aggregated into a next level, summed over the levels (for the
                           * whole-step algorithmic byte count of the bench line: SURVEY 8(d) charges 2 * 16 bytes per entry) */
} b3c_louvain_info;

/* ---- context ------------------------------------------------------------------------------- */

/* Create a context on `device`.  `stream` is a hipStream_t to launch on (NULL: the library creates
 * its own non-blocking stream). */
int b3c_ctx_create(int device, void *stream, b3c_ctx **out);
int b3c_ctx_destroy(b3c_ctx *ctx);
/* The B3C_* development switches of the process environment are read ONCE, by b3c_ctx_create; no other
 * entry point looks at the environment.  A caller that changes them afterwards (the test suite, the profiling tools) calls this to
 * have the context read them again. */
int b3c_ctx_reload_env(b3c_ctx *ctx);
const char *b3c_last_error(const b3c_ctx *ctx);
int b3c_ctx_sync(b3c_ctx *ctx);
/* HBM bytes currently held by the context's matrices and workspaces. */
int64_t b3c_ctx_bytes(const b3c_ctx *ctx);
/* Return the context's cached (pooled, currently unused) HBM blocks to the driver.  Workspaces are recycled between calls so that
 * a warm step issues no hipMalloc; a caller about to need the memory for something else (a 16 GB map next to a pool sized by the
 * last one) trims first.  Matrices and blocks in use are not touched. */
int b3c_ctx_trim(b3c_ctx *ctx);

/* ---- multi-GPU: contig-row ranges of ONE map across ranks ---------------------------------- */

/* One process per GPU.  Every rank WORKS on one contiguous range of rows of the ordered matrix (b3c_dist.starts; equal ranges
 * [rank*chunk, (rank+1)*chunk) when none are given) -- either of a
 * whole matrix, or of a row block that holds only those rows (b3c_mat_permute_rows; the other rows are empty);
 * what the other ranks need is exchanged through four caller-owned device buffers, each laid out as `world` equal
 * slices, by an in-place all-gather ("my slice is already written").  All ranks reduce the gathered partials in rank
 * order, so every rank takes identical decisions and results do not depend on timing.
 *
 * Transport: with `comm` set (an RCCL communicator from b3c_dist_comm_create) libb3c issues ncclAllGather itself on the
 * context's stream -- xGMI between the GPUs of a node, no host round trip.  With comm == NULL it calls
 * `allgather(user, buffer_id)` instead, which must enqueue the same collective on the context's stream and return 0
 * (torch.distributed with gloo in the CPU / one-GPU tests, bin3c_amd/dist.py).
 *
 * KR: ONE exchange per SpMV (B3C_BUF_SCAL_VEC, one grouped launch) -- the rank's partial of p.w | rk.rk together with its
 * slice of w (inner iteration) or of v = x*(A x) (outer iteration).  The vector steps on both sides of the SpMV (alpha,
 * y, rk, Z, the bound checks; then p and the operand x*p) run over ALL rows on every rank from the gathered slices, so
 * their reductions are complete everywhere and nothing else travels: 8n bytes + 32 bytes per rank and SpMV.
 * Louvain level 0: the decided labels once per sub-round (B3C_BUF_LABELS), one integer per modularity evaluation
 * (B3C_BUF_ISUM); moves are applied by every rank.
 * b3c_locality_order: the two passes over the (whole, un-ordered) matrix run on this rank's rows; their per-row results
 * travel through B3C_BUF_LABELS (once) and B3C_BUF_VEC (twice).  All of these calls are collective over the ranks. */
#define B3C_BUF_VEC 0     /* double[world*chunk]  KR: slices of w | v               */
#define B3C_BUF_SCAL 1    /* double[world*4]      KR partial sums / min / max      */
#define B3C_BUF_LABELS 2  /* int32[world*chunk]   Louvain decided communities      */
#define B3C_BUF_ISUM 3    /* int64[world*2]       Louvain integer partial sums     */
#define B3C_BUF_SCAL_VEC 4 /* B3C_BUF_SCAL and B3C_BUF_VEC in one (grouped) exchange */
typedef int (*b3c_allgather_fn)(void *user, int buffer_id);
/* comm == NULL only: sum `count` uint64 values at device pointer `dev` over all ranks, in place, on the context's stream */
typedef int (*b3c_allreduce_fn)(void *user, void *dev, int64_t count);
/* comm == NULL only: ragged in-place all-gather of a row-indexed vector at device pointer `dev` (elements of esz = 4 or 8 bytes):
 * rank r owns elements [starts[r], starts[r + 1]) (starts: world + 1 host offsets); afterwards every rank holds [0, starts[world]) */
typedef int (*b3c_allgatherv_fn)(void *user, void *dev, int esz, const int64_t *starts, int world);
typedef struct {
    int32_t rank, world;
    int64_t chunk;        /* exchange buffers hold world*chunk >= n elements (vec, labels: indexed by contig row) */
    double *vec;
    double *scal;
    int32_t *labels;
    int64_t *isum;
    b3c_allgather_fn allgather;   /* used when comm == NULL */
    void *user;
    void *comm;           /* ncclComm_t of b3c_dist_comm_create, or NULL */
    b3c_allreduce_fn allreduce;   /* used when comm == NULL (Louvain: community-pair bins of the sharded aggregation) */
    /* Row ranges of the ranks: rank r works on rows [starts[r], starts[r + 1]) of the ordered map (world + 1 ascending host
     * offsets, starts[0] = 0, starts[world] = n; copied by b3c_ctx_set_dist).  NULL: equal ranges [r*chunk, (r+1)*chunk).  The
     * caller balances them by STORED ENTRIES -- a prefix sum over the ordered row lengths -- because the rows of a
     * contact map are anything but equally long (one genome may hold a third of the contigs, hub contigs 10^5..10^6 contacts).
     * The vector and label exchanges then move ragged slices (one grouped launch of per-rank broadcasts, or `allgatherv`). */
    const int64_t *starts;
    b3c_allgatherv_fn allgatherv; /* used when comm == NULL and starts != NULL */
} b3c_dist;
/* Install (or with NULL remove) the sharding description; it applies to b3c_kr_balance and b3c_louvain
 * calls on matrices with n <= world*chunk rows.  No reference counterpart (bin3C is single process). */
int b3c_ctx_set_dist(b3c_ctx *ctx, const b3c_dist *dist);
/* RCCL communicator over the ranks' GPUs: rank 0 obtains a unique id (128 bytes, host memory), the caller hands it to
 * every rank (any side channel: torch.distributed broadcast, MPI, a file), every rank creates its communicator on its
 * context's device.  librccl.so.1 is bound at run time. */
int b3c_dist_unique_id(void *id128);
int b3c_dist_comm_create(b3c_ctx *ctx, int world, int rank, const void *id128, void **comm_out);
int b3c_dist_comm_destroy(b3c_ctx *ctx, void *comm);
/* Row-sharded pipeline helpers (identity when no sharding is installed): the maximum of a host double over the ranks
 * (1/max of the sub-map, cluster.py:316), and the all-gather of a per-contig int32 vector of which every rank computed
 * the rows [rank*chunk, (rank+1)*chunk) (the off-diagonal signal of the acceptance mask, contact_map.py:899). */
int b3c_dist_max_f64(b3c_ctx *ctx, double *value);
int b3c_dist_allgather_i32(b3c_ctx *ctx, int32_t *vec /*host|dev*/, int64_t n);
/* Exchanges issued through this context since it was created (every transport). */
int64_t b3c_ctx_exchanges(const b3c_ctx *ctx);
/* Direct transport (optional, after b3c_ctx_set_dist): the in-place all-gathers of the exchange buffers and of the ragged row
 * vectors as PEER WRITES into a window instead of collectives.  Every rank creates a window of two slots of `slot_bytes` staging
 * bytes (at least 8 * world * chunk + 64 * world + 256: the vector and the scalar buffer of a KR tick travel together) in its
 * HBM and obtains its 64-byte IPC handle (hipIpcGetMemHandle); the caller hands all ranks' handles to every rank (any side
 * channel), b3c_dist_window_open maps them (hipIpcOpenMemHandle: xGMI between the GPUs of a node; the same GPU between two
 * processes sharing one).  From then on an exchange is two small launches on the context's stream -- push this rank's range into
 * every peer's slot and raise a sequence flag there, wait for all flags of the own slot and copy the others' ranges out -- with
 * no collective and no rendezvous; all-reduces and the aggregation's all-to-all keep using `comm` / the callbacks.
 * b3c_ctx_exchanges_direct counts the exchanges that went through the window.  Collective over the ranks, like the exchanges. */
int b3c_dist_window_create(b3c_ctx *ctx, int64_t slot_bytes, void *handle64 /*host, 64 bytes out*/);
int b3c_dist_window_open(b3c_ctx *ctx, const void *handles /*host, world * 64 bytes*/);
int b3c_dist_window_close(b3c_ctx *ctx);
int64_t b3c_ctx_exchanges_direct(const b3c_ctx *ctx);

/* ---- matrices ------------------------------------------------------------------------------ */

/* Copy a CSR matrix into HBM.  Replaces the scipy objects handed between contact_map.py:929-945. */
int b3c_mat_upload(b3c_ctx *ctx, int64_t n, int64_t nnz, const int64_t *indptr /*host|dev*/,
                   const int32_t *indices /*host|dev*/, const void *data /*host|dev*/, int kind, b3c_mat **out);
int b3c_mat_free(b3c_ctx *ctx, b3c_mat *m);
int b3c_mat_info(const b3c_mat *m, int64_t *n, int64_t *nnz, int *kind);
/* Stored entries of the rows: out[k] = indptr[idx[k] + 1] - indptr[idx[k]] (idx == NULL: row k).  idx int32[n], out int64[n];
 * host|dev.  What the sharded pipeline balances its row ranges by (np.diff(indptr) of the scipy object, in the working order). */
int b3c_mat_row_nnz(b3c_ctx *ctx, const b3c_mat *m, const int32_t *idx /*host|dev, may be NULL*/, int64_t *out /*host|dev*/);
/* Copy back; any of the three output pointers may be NULL. */
int b3c_mat_download(b3c_ctx *ctx, const b3c_mat *m, int64_t *indptr, int32_t *indices, void *data);
/* Raw device pointers of a matrix (for zero-copy consumers such as torch tensors).  Matrices are immutable once created;
 * a caller that nevertheless writes through `data` must ask for the pointer again afterwards (the call drops the maximum
 * that b3c_max_value / b3c_normalise_scale_compress keep with the matrix). */
int b3c_mat_device_ptrs(const b3c_mat *m, void **indptr, void **indices, void **data);

/* ---- ingest tail: pair stream -> symmetric count matrix ------------------------------------- */

/* Sparse2DAccumulator.get_coo (sparse_utils.py:246-266) on the device: `m` observations (lo[k], hi[k], counts[k])
 * in any order and orientation are summed per unordered pair (saturating at 2^32-1), self pairs are dropped, the
 * strict upper triangle is mirrored, and diag[n] (NULL: none) supplies the self-contact counts.  Result: canonical
 * symmetric B3C_U32 CSR, the input of the cluster path.  n_edges (may be NULL) receives the number of undirected
 * off-diagonal edges.  lo/hi: int64[m], counts: uint32[m], diag: uint32[n]; all host|dev. */
int b3c_coo_symmetrise(b3c_ctx *ctx, int64_t n, int64_t m, const int64_t *lo, const int64_t *hi, const uint32_t *counts,
                       const uint32_t *diag, b3c_mat **out, int64_t *n_edges);

/* ---- locality ordering (performance only; results never depend on it) -------------------- */

/* Order the contigs of a symmetric matrix so that strongly linked ones are adjacent (three Boruvka
 * rounds on the heaviest contacts, reorder.hip).  new2old / old2new: int32[n], host|dev.  No reference
 * counterpart: the reference keeps assembly order and pays for it in cache misses inside scipy. */
int b3c_locality_order(b3c_ctx *ctx, const b3c_mat *a, int32_t *new2old /*host|dev*/, int32_t *old2new /*host|dev*/);

/* P A P^T as a new canonical CSR whose row k is row new2old[k] of `a`.  The result remembers the
 * caller-side id of every row (b3c_mat_ids) and is marked ordered: KR and Louvain then skip their
 * own internal ordering, site_norm / compress carry the ids along, and b3c_louvain hashes and
 * tie-breaks on the ids so that its labels equal those computed on `a` (returned per ROW of the
 * permuted matrix).  Same mechanism as ContactMap._reorder_seq (contact_map.py:1066-1085). */
#define B3C_PERMUTE_UNSORTED 1 /* move rows and relabel columns only: the order inside a row is kept, so columns no
                                * longer ascend.  ~7x cheaper than the canonical form (no sort of the entries); every
                                * libb3c entry point accepts such a matrix, but it is not canonical CSR on download. */
int b3c_mat_permute(b3c_ctx *ctx, const b3c_mat *a, const int32_t *new2old /*host|dev*/, int flags, b3c_mat **out);
/* The same, but only the rows [row0, row1) of the result are formed; the others stay empty.  This is the ROW BLOCK a rank of
 * the sharded path owns: site_norm / max_offdiag / b3c_kr_balance_counts / b3c_normalise_scale_compress / b3c_louvain then
 * touch 1/world of the entries each, and combine what must be global (degrees, totals, checksums, community-pair sums)
 * through the exchange buffers of b3c_ctx_set_dist.  row0 / row1 must be this rank's range ([starts[rank], starts[rank+1]), or
 * [rank*chunk, (rank+1)*chunk) without starts). */
int b3c_mat_permute_rows(b3c_ctx *ctx, const b3c_mat *a, const int32_t *new2old /*host|dev*/, int flags, int64_t row0,
                         int64_t row1, b3c_mat **out);

/* Caller-side row ids of a permuted (and possibly compressed) matrix, int32[n] host|dev.
 * Returns 1 and writes nothing when the matrix carries no ids (identity). */
int b3c_mat_ids(b3c_ctx *ctx, const b3c_mat *m, int32_t *out /*host|dev*/);

/* ---- CSR helpers --------------------------------------------------------------------------- */

/* max_offdiag (sparse_utils.py:269-281): per-column maximum with the diagonal zeroed; out is
 * uint32[n].  `assume_symmetric` != 0 lets the kernel reduce rows instead (seq_map is symmetric by
 * construction, sparse_utils.py:261-266). */
int b3c_max_offdiag_u32(b3c_ctx *ctx, const b3c_mat *counts, int assume_symmetric, uint32_t *out /*host|dev*/);

/* set_primary_acceptance_mask (contact_map.py:888-905) from its two per-contig inputs: mask[i] = lengths[i] >= min_len and
 * signal[i] >= min_sig, with signal = b3c_max_offdiag_u32.  lengths: int64[n], signal: uint32[n], mask: uint8[n]; host|dev. */
int b3c_acceptance_mask(b3c_ctx *ctx, int64_t n, const int64_t *lengths /*host|dev*/, const uint32_t *signal /*host|dev*/,
                        int64_t min_len, int64_t min_sig, uint8_t *mask /*host|dev*/);

/* Per-contig vectors between the caller's numbering and a permuted one (b3c_locality_order): gather dst[i] = src[idx[i]], or with
 * scatter != 0 dst[idx[i]] = src[i]; elem_bytes 1, 4 or 8; idx int32[n]; host|dev.  (SeqOrder does the same on the host,
 * contact_map.py:203-251.) */
int b3c_vec_permute(b3c_ctx *ctx, int64_t n, int elem_bytes, const void *src /*host|dev*/, const int32_t *idx /*host|dev*/,
                    void *dst /*host|dev*/, int scatter);

/* _norm_seq(use_sites=True) (contact_map.py:1110-1129) + fast_norm_fullseq_bysite (:100-113) after
 * the astype(float) of :929: out_ij = counts_ij * (1.0 / (s_i * s_j)), zero sites -> 1 (:1103-1108).
 * Creates a new B3C_F64 matrix that shares no storage with `counts`. */
int b3c_site_norm(b3c_ctx *ctx, const b3c_mat *counts, const double *sites /*host|dev*/, b3c_mat **out);

/* X.T * (orig * X) (sparse_utils.py:223-224): a_ij <- x_i * (a_ij * x_j), in place. */
int b3c_scale_sym(b3c_ctx *ctx, b3c_mat *a, const double *x /*host|dev*/);

/* The same scaling in "graph form", for a BITWISE symmetric `a` (b3c_kr_info.is_symmetric came from
 * the checksum path): both triangles receive x_hi * (a_ij * x_lo), hi/lo by caller-side id, i.e. the value
 * scipy's result holds in the LOWER triangle and nx.Graph ends up holding for the edge (cluster.py:319-321,
 * the later add_edge(v,u) overwrites (u,v)).  The upper triangle then differs from sparse_utils.py:223-224 by
 * at most two ulps; in exchange b3c_louvain needs no mirror lookups.  Marks the matrix value-symmetric. */
int b3c_scale_sym_graph(b3c_ctx *ctx, b3c_mat *a, const double *x /*host|dev*/);

/* compress (sparse_utils.py:284-314): keep rows/columns with mask != 0, ids re-indexed by
 * i - cumsum(~mask)[i]. */
int b3c_compress(b3c_ctx *ctx, const b3c_mat *a, const uint8_t *mask /*host|dev*/, b3c_mat **out);

/* _map.max() (cluster.py:316). */
int b3c_max_value(b3c_ctx *ctx, const b3c_mat *a, double *out);

/* downsample (sparse_utils.py:48-87; caller ContactMap.plot, contact_map.py:1340-1349): reduce a float64 canonical
 * CSR map by block_size in each dimension.  The result is ceil(n / block_size) square, zero padded at the edge.
 * B3C_DS_MEAN: block sum times 1/block_size^2 (padding not compensated, as in the reference);
 * B3C_DS_MAX : block maximum, implicit zeros taking part.  Deterministic (no floating-point atomics). */
#define B3C_DS_MEAN 0
#define B3C_DS_MAX 1
#define B3C_DS_SKIP_DIAG 2 /* or-ed into method: treat the diagonal as zeroed first (`_map.setdiag(0)`, contact_map.py:1324) */
int b3c_downsample(b3c_ctx *ctx, const b3c_mat *a, int64_t block_size, int method, b3c_mat **out);

/* ---- Knight-Ruiz --------------------------------------------------------------------------- */

/* kr_biostochastic (sparse_utils.py:90-224) minus the final scaling (use b3c_scale_sym):
 * x (double[n], host|dev) such that diag(x) A diag(x) is doubly stochastic.  `a` is not modified;
 * zero diagonals are treated as one on an internal working term (sparse_utils.py:110-115).
 * Returns B3C_E_NOCONV / B3C_E_NAN like the reference raises; `info` is filled in either case. */
int b3c_kr_balance(b3c_ctx *ctx, const b3c_mat *a, double tol, double delta, double Delta, int64_t max_iter,
                   double *x /*host|dev*/, b3c_kr_info *info);

/* The same balancing on the RAW COUNT matrix: A = diag(1/s) C diag(1/s) is never materialised, the SpMV streams
 * 8 bytes per stored entry (uint32 count + int32 column) instead of 12 and folds the two 1/s factors into its operand
 * and its row epilogue.  Replaces the pair contact_map.py:933 (_norm_seq) + :939 (_bisto_seq) of prepare_seq_map.
 * Mathematically identical to b3c_kr_balance(b3c_site_norm(counts, sites)); x agrees to rounding (the reference
 * rounds 1/(s_i*s_j) once per entry, this form rounds 1/s_i and 1/s_j) -- tests hold it to 1e-9 of the reference. */
int b3c_kr_balance_counts(b3c_ctx *ctx, const b3c_mat *counts, const double *sites /*host|dev*/, double tol, double delta,
                          double Delta, int64_t max_iter, double *x /*host|dev*/, b3c_kr_info *info);

/* Companion of b3c_kr_balance_counts: out_ij = x_i * (a_ij * x_j) with a_ij = counts_ij * (1 / (s_i * s_j)) formed
 * exactly as contact_map.py:110-113 does, i.e. bit-identical to b3c_site_norm + b3c_scale_sym (graph_form = 0) or
 * + b3c_scale_sym_graph (graph_form != 0; requires b3c_kr_info.bitwise_symmetric).  One pass instead of two. */
int b3c_normalise_scale(b3c_ctx *ctx, const b3c_mat *counts, const double *sites /*host|dev*/, const double *x /*host|dev*/,
                        int graph_form, b3c_mat **out);

/* The same followed by b3c_compress (get_subspace, contact_map.py:977-981) in ONE pass: the kept entries of the balanced
 * map are computed straight from the counts, bit-identical to b3c_normalise_scale + b3c_compress, and the full balanced
 * map (2.4 GB at 1M contigs / 100M edges) is never written.  For callers that only need the masked sub-map -- the cluster
 * step; processed_map itself (contact_map.py:945) stays available through b3c_normalise_scale. */
int b3c_normalise_scale_compress(b3c_ctx *ctx, const b3c_mat *counts, const double *sites /*host|dev*/,
                                 const double *x /*host|dev*/, int graph_form, const uint8_t *mask /*host|dev*/, b3c_mat **out);

/* ---- Louvain ------------------------------------------------------------------------------- */

/* to_graph's weight scaling (cluster.py:316-321) + louvain.cluster / community.best_partition
 * (louvain.py:70) on a structurally symmetric B3C_F64 matrix: labels[i] in [0, n_comm) or -1 for a
 * node whose row is empty (never added to the graph by cluster.py:321).  scl multiplies every
 * weight (pass 1/max for scale=True).  n_sub = sub-rounds per sweep (0 -> default 4).
 * The graph is UNDIRECTED: the weight of (i, j) as the library forms it (the lower-triangle value on both sides, cluster.py:319-321)
 * is symmetric by construction, and the internal-weight recount relies on it (it counts the entries above the diagonal twice after
 * a sweep that moved many nodes); the structure must hold both (i, j) and (j, i).
 * Row-sharded runs: a row block (b3c_mat_permute_rows) must come ordered by the caller; maps of more than 2^23 contigs whose rows
 * all stay within 512 entries cannot be cut into row blocks by b3c_mat_permute_rows (B3C_E_ARG) -- sharding is supported up to
 * 2^23 contigs, the single-GPU path up to 2^24 (the tail pass of the balancing) and 2^31 - 1 (everything else). */
int b3c_louvain(b3c_ctx *ctx, const b3c_mat *a, double scl, int n_sub, int max_sweeps,
                int32_t *labels /*host|dev*/, b3c_louvain_info *info);

/* Fixed-point edge weights the Louvain kernels work on (int64[nnz], host|dev) and their shift. */
int b3c_graph_quantise(b3c_ctx *ctx, const b3c_mat *a, double scl, int64_t *q /*host|dev*/, int32_t *shift);

/* ---- build info ---------------------------------------------------------------------------- */
const char *b3c_version(void);

/* Layout of the three structs that cross this ABI, as the library was COMPILED: out[0] = sizeof, out[1] = number of
 * fields, out[2 ..] = offsetof of every field in declaration order.  which: B3C_ABI_KR_INFO, B3C_ABI_LOUVAIN_INFO,
 * B3C_ABI_DIST.  Returns the number of ints written, or B3C_E_ARG (unknown struct, cap too small).  A binding checks
 * this against its own struct definitions before the first call that writes through them (bin3c_amd/_native.py does at
 * load time; tests/test_abi.py holds header, library, ctypes table and the INTEGRATION.md stub to one layout). */
#define B3C_ABI_KR_INFO 0
#define B3C_ABI_LOUVAIN_INFO 1
#define B3C_ABI_DIST 2
int b3c_abi_layout(int which, int32_t *out, int cap);

#ifdef __cplusplus
}
#endif
#endif
