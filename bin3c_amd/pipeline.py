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


def run_hot_path(ctx, counts, sites, lengths, min_len, min_sig, tol=1e-6, n_sub=0, cluster=True, keep=False):
    """
    :param ctx: _native.Context
    :param counts: _native.Mat (B3C_U32) raw symmetric contact counts, already in HBM
    :param sites: float64[n] restriction-site counts per contig (zeros are treated as one)
    :param lengths: int[n] contig lengths
    :return: HotPathResult (device matrices freed unless keep=True)
    """
    res = HotPathResult()
    t = res.timings
    t0 = time.perf_counter()
    res.mask, res.signal = acceptance_mask(ctx, counts, lengths, min_len, min_sig)
    t['mask'] = time.perf_counter() - t0

    t0 = time.perf_counter()
    a = ctx.site_norm(counts, sites)                       # contact_map.py:929-933
    t['norm'] = time.perf_counter() - t0

    t0 = time.perf_counter()
    rc, res.x, res.kr_info = ctx.kr_balance(a, tol=tol, check=False)     # contact_map.py:939 -> sparse_utils.py:90
    if rc != _native.OK:
        a.free()
        ctx._check(rc)
    ctx.scale_sym(a, res.x)                                # sparse_utils.py:223-224
    t['kr'] = time.perf_counter() - t0
    res.processed = a

    t0 = time.perf_counter()
    if res.mask.all():                                     # contact_map.py:977: only compress when something is masked
        sub = a
    else:
        sub = ctx.compress(a, res.mask)
        res.sub = sub
    res.scl = 1.0 / ctx.max_value(sub)                     # cluster.py:316
    t['compress'] = time.perf_counter() - t0

    if cluster:
        t0 = time.perf_counter()
        res.labels, res.lv_info = ctx.louvain(sub, res.scl, n_sub=n_sub)   # cluster.py:319-321 + louvain.py:70
        t['louvain'] = time.perf_counter() - t0
    if not keep:
        res.free()
    return res
