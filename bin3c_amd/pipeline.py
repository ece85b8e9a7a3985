"""
The cluster-step hot path, device resident from the raw count matrix to the community labels.

Stage by stage this is what ``bin3C.py cluster`` does between ``load_object`` and the writers
(reference call stack: cluster.py:156 to_graph -> contact_map.py:856 set_primary_acceptance_mask ->
:911 prepare_seq_map -> :1110 _norm_seq -> :1087 _bisto_seq -> sparse_utils.py:90 kr_biostochastic ->
contact_map.py:947 get_subspace -> sparse_utils.py:284 compress -> cluster.py:316 scl ->
louvain.py:70 best_partition), with the matrix never leaving HBM.
"""
import logging
import time

import numpy as np

from . import _native

logger = logging.getLogger(__name__)


class HotPathResult(object):
    """Everything the host-side callers (ContactMap / cluster_map mirrors, bench) need back."""

    def __init__(self):
        self.mask = None            # bool[n]     primary acceptance mask (contact_map.py:888-905)
        self.signal = None          # uint32[n]   max_offdiag
        self.x = None               # float64[n]  bisto_scale (contact_map.py:941)
        self.processed = None       # _native.Mat balanced full map, device resident (processed_map :945)
        self.sub = None             # _native.Mat masked sub-map (get_subspace :977-981); None if nothing masked
        self.scl = None             # 1 / max (cluster.py:316)
        self.labels = None          # int32[n_accepted] community per accepted (gapless) contig, -1 = not in graph
        self.kr_info = None
        self.lv_info = None
        self.timings = {}
        self.perm = (None, None)    # (new2old, old2new) of the locality ordering, or (None, None)

    def free(self):
        for m in (self.processed, self.sub):
            if m is not None:
                m.free()
        self.processed = self.sub = None


def acceptance_mask(ctx, counts, lengths, min_len, min_sig):
    """set_primary_acceptance_mask (contact_map.py:888-905): length and off-diagonal signal filters."""
    signal = ctx.max_offdiag_u32(counts, assume_symmetric=True)   # seq_map is symmetric: sparse_utils.py:261-266
    mask = (np.asarray(lengths) >= min_len) & (signal >= min_sig)
    return mask, signal


def run_hot_path(ctx, counts, sites, lengths, min_len, min_sig, tol=1e-6, n_sub=0, cluster=True, keep=False,
                 reorder=True):
    """
    :param ctx: _native.Context
    :param counts: _native.Mat (B3C_U32) raw symmetric contact counts, already in HBM
    :param sites: float64[n] restriction-site counts per contig (zeros are treated as one)
    :param lengths: int[n] contig lengths
    :param reorder: run the whole path on a locality-ordered copy of the matrix (reorder.hip); every
                    returned vector is in the caller's numbering either way
    :return: HotPathResult (device matrices freed unless keep=True)
    """
    res = HotPathResult()
    t = res.timings
    sites = np.asarray(sites, dtype=np.float64)
    lengths = np.asarray(lengths)
    n2o = o2n = None
    work = counts
    t0 = time.perf_counter()
    if reorder and counts.n >= 32768 and counts.nnz >= (1 << 21):
        n2o, o2n = ctx.locality_order(counts)
        work = ctx.permute(counts, n2o)
        sites, lengths = sites[n2o], lengths[n2o]
    res.perm = (n2o, o2n)
    t['order'] = time.perf_counter() - t0

    try:
        t0 = time.perf_counter()
        mask_w, signal_w = acceptance_mask(ctx, work, lengths, min_len, min_sig)
        t['mask'] = time.perf_counter() - t0

        t0 = time.perf_counter()
        a = ctx.site_norm(work, sites)                         # contact_map.py:929-933
        t['norm'] = time.perf_counter() - t0
    finally:
        if work is not counts:
            work.free()

    t0 = time.perf_counter()
    rc, x_w, res.kr_info = ctx.kr_balance(a, tol=tol, check=False)       # contact_map.py:939 -> sparse_utils.py:90
    if rc != _native.OK:
        a.free()
        ctx._check(rc)
    ctx.scale_sym(a, x_w)                                  # sparse_utils.py:223-224
    t['kr'] = time.perf_counter() - t0
    res.processed = a

    t0 = time.perf_counter()
    if mask_w.all():                                       # contact_map.py:977: only compress when something is masked
        sub = a
    else:
        sub = ctx.compress(a, mask_w)
        res.sub = sub
    res.scl = 1.0 / ctx.max_value(sub)                     # cluster.py:316
    t['compress'] = time.perf_counter() - t0

    if o2n is None:
        res.mask, res.signal, res.x = mask_w, signal_w, x_w
    else:
        res.mask, res.signal, res.x = mask_w[o2n], signal_w[o2n], x_w[o2n]

    if cluster:
        t0 = time.perf_counter()
        labels_w, res.lv_info = ctx.louvain(sub, res.scl, n_sub=n_sub)    # cluster.py:319-321 + louvain.py:70
        ids = ctx.mat_ids(sub)                             # gapless caller-side id of every row of `sub`
        if ids is None:
            res.labels = labels_w
        else:
            res.labels = np.empty_like(labels_w)
            res.labels[ids] = labels_w
        t['louvain'] = time.perf_counter() - t0
    if not keep:
        res.free()
    return res


def processed_to_host(ctx, res):
    """processed_map (contact_map.py:945) as host CSR arrays in the caller's numbering."""
    m = res.processed
    n2o, o2n = res.perm
    if n2o is None:
        return m.download()
    back = ctx.permute(m, o2n)          # row k of the result = row o2n[k] of the ordered matrix = caller row k
    try:
        return back.download()
    finally:
        back.free()
