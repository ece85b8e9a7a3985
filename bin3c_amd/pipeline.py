"""
The cluster-step hot path, device resident from the raw count matrix to the community labels.

Stage by stage this is what ``bin3C.py cluster`` does between ``load_object`` and the writers
(reference call stack: cluster.py:156 to_graph -> contact_map.py:856 set_primary_acceptance_mask ->
:911 prepare_seq_map -> :1110 _norm_seq -> :1087 _bisto_seq -> sparse_utils.py:90 kr_biostochastic ->
contact_map.py:947 get_subspace -> sparse_utils.py:284 compress -> cluster.py:316 scl ->
louvain.py:70 best_partition), with the matrix never leaving HBM.

torch is used here for what it is good at -- device buffers for the n-vectors and a stream shared with
libb3c -- so that no vector bounces through the host between stages; all arithmetic of the path is
in libb3c's HIP kernels.
"""
import logging
import time

import numpy as np

from . import _native

logger = logging.getLogger(__name__)

REORDER_MIN_N = 32768          # below this the gathered vectors sit in L2 anyway
REORDER_MIN_NNZ = 1 << 21


def pin_to_device_cpus(device=0):
    """Restrict this process to the CPUs of the NUMA node the GPU hangs off (sysfs local_cpulist).  The KR iteration is
    62 x 4 dependent kernel launches; issued from the far socket of a two-socket box every launch costs ~10 us more
    and the step 4 % (C2: 25 %).  One process per GPU, so this is the usual placement.  B3C_NO_PIN=1 leaves the
    affinity alone; anything unexpected in sysfs does too.  -> the CPU set, or None."""
    import os
    import torch
    if os.environ.get('B3C_NO_PIN') or not hasattr(os, 'sched_setaffinity'):
        return None
    try:
        p = torch.cuda.get_device_properties(device)
        bdf = '%04x:%02x:%02x.0' % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        with open('/sys/bus/pci/devices/%s/local_cpulist' % bdf) as fh:
            text = fh.read().strip()
        cpus = set()
        for part in text.split(','):
            lo, _, hi = part.partition('-')
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus or cpus == os.sched_getaffinity(0):
            return None
        os.sched_setaffinity(0, cpus)
        return cpus
    except (OSError, ValueError, AttributeError):
        return None


def make_context(device=0, pin=None):
    """A libb3c context that launches on a torch stream of `device` (so torch ops and kernels interleave in order).
    pin=True restricts the PROCESS to the CPUs of the GPU's NUMA node (pin_to_device_cpus: worth 4-25 % of a step on
    two-socket hosts) -- a process-wide side effect, so library callers get it only on request or with B3C_PIN=1;
    bench.py and the bin3C.py command line ask for it."""
    import torch
    if not torch.cuda.is_available():
        raise _native.B3CError(_native.E_HIP, 'no MI355X visible: torch.cuda.is_available() is False and libb3c has '
                                              'no CPU fallback')
    import os
    if pin or (pin is None and os.environ.get('B3C_PIN')):
        pin_to_device_cpus(device)
    stream = torch.cuda.Stream(device=device)
    ctx = _native.Context(device, stream=stream.cuda_stream)
    ctx.torch_stream = stream
    return ctx


class HotPathResult(object):
    """Everything the host-side callers (ContactMap / cluster_map mirrors, bench) need back."""

    def __init__(self):
        self.mask = None            # bool[n]     primary acceptance mask (contact_map.py:888-905)
        self.signal = None          # uint32[n]   max_offdiag
        self.x = None               # float64[n]  bisto_scale (contact_map.py:941)
        self._processed = None      # _native.Mat balanced full map, device resident (processed_map :945); see `processed`
        self._lazy = None           # (ctx, counts in the working order, sites, x, graph_form, owned) to form it on demand
        self.sub = None             # _native.Mat masked sub-map (get_subspace :977-981); None if nothing masked
        self.scl = None             # 1 / max (cluster.py:316)
        self.labels = None          # int32[n_accepted] community per accepted (gapless) contig, -1 = not in graph
        self.kr_info = None
        self.kr_strict = False      # x comes from the second, explicit-fp64 balance (run_hot_path(strict_kr=True) on a hard map)
        self.lv_info = None
        self.timings = {}
        self._perm = None           # see `perm`
        self._n2o = self._o2n = None  # locality ordering on the device (int32 new2old, old2new), or None
        self.row_block = None       # (r0, r1): rows of the ordered map this rank formed and worked on (sharded run), else None
        self.work_nnz = None        # stored entries of the map this rank worked on
        self.sub_nnz = None         # stored entries of the masked sub-map Louvain's level 0 ran on (this rank's rows when sharded)

    @property
    def perm(self):
        """(new2old, old2new) int32 host arrays of the locality ordering, or (None, None); copied from HBM on first access
        (8 MB at C3 that the cluster step itself never needs on the host)"""
        if self._perm is None:
            self._perm = (None, None) if self._n2o is None else (self._n2o.cpu().numpy(), self._o2n.cpu().numpy())
        return self._perm

    @property
    def processed(self):
        """The balanced full map in HBM.  The cluster step itself never needs it (the masked sub-map is formed straight
        from the counts, b3c_normalise_scale_compress), so it is written on first access: 2.4 GB at C3."""
        if self._processed is None and self._lazy is not None:
            ctx, work, sites_t, x_w, graph_form, _owned = self._lazy
            import torch
            with torch.cuda.stream(ctx.torch_stream):
                self._processed = ctx.normalise_scale(work, sites_t, x_w, graph_form=graph_form)
        return self._processed

    @processed.setter
    def processed(self, mat):
        self._processed = mat

    def has_processed(self):
        """the balanced map is in HBM or can be formed without the caller's help"""
        return self._processed is not None or self._lazy is not None

    def free(self):
        for m in (self._processed, self.sub):
            if m is not None:
                m.free()
        if self._lazy is not None and self._lazy[5]:
            self._lazy[1].free()
        self._processed = self.sub = self._lazy = None


def acceptance_mask(ctx, counts, lengths, min_len, min_sig):
    """set_primary_acceptance_mask (contact_map.py:888-905) on host arrays: length and off-diagonal signal filters."""
    signal = ctx.max_offdiag_u32(counts, assume_symmetric=True)   # seq_map is symmetric: sparse_utils.py:261-266
    mask = (np.asarray(lengths) >= min_len) & (signal >= min_sig)
    return mask, signal


STRICT_KR_ITER = 100      # strict_kr: balance again in the reference's own formulation from this many iterations on


def run_hot_path(ctx, counts, sites, lengths, min_len, min_sig, tol=1e-6, n_sub=0, cluster=True, keep=False,
                 reorder=True, mask=None, adopt_counts=False, max_iter=1000, strict_kr=None, balance_rows=True):
    """
    :param ctx: context from :func:`make_context`
    :param counts: _native.Mat (B3C_U32) raw symmetric contact counts, already in HBM
    :param sites: float64[n] restriction-site counts per contig (zeros are treated as one); numpy or device tensor
    :param lengths: int[n] contig lengths; numpy or device tensor
    :param reorder: run the whole path on a locality-ordered copy of the matrix (reorder.hip); every
                    returned vector is in the caller's numbering either way
    :param mask: bool[n] acceptance mask decided by the caller (ContactMap.primary_acceptance_mask); when
                 None it is computed here from lengths / min_len / min_sig
    :param max_iter: iteration limit of the balancing (sparse_utils.py:90, default 1000 like the reference)
    :param strict_kr: which formulation of the balancing returns x.  The count form (b3c_kr_balance_counts: raw counts, the two
                      1/s factors folded into operand and epilogue, 6 bytes per entry) perturbs every matrix entry by an ulp
                      (1/s_i * 1/s_j against 1/(s_i s_j)); on a tiny ill-conditioned map that has moved x by 3e-6 once in 1700
                      (tests/golden/illcond_kr_counts.npz: the reference's two environments agree to 8e-9 there), beyond the
                      1e-6 contract.  The reference's own formulation (contact_map.py:933 then sparse_utils.py:90: explicit
                      fp64 matrix, 12 bytes per entry, b3c_site_norm + b3c_kr_balance) stays within 4.3e-7 on all 1715 stable
                      maps of the campaign (tools/fuzz_ill.py).
                      None (default): maps below REORDER_MIN_NNZ entries -- where the count form buys nothing: the whole
                      balancing is launch latency -- are balanced in the reference's formulation; larger maps in the count form
                      (x within 1e-9 of the reference on every fixture and at C2 / C3).
                      True: in addition, a larger map that needed STRICT_KR_ITER iterations or more is balanced AGAIN in the
                      reference's formulation and that x is returned (doubles the balancing time where it is longest).
                      False: the count form whatever the size.
    :param balance_rows: row-sharded runs: row blocks balanced by stored entries (default) instead of equal row counts
    :param adopt_counts: the result takes `counts` over (frees it with res.free()); with keep=True the balanced map can then
                         still be formed on demand when no ordered copy was made
    :return: HotPathResult (device matrices freed unless keep=True)
    """
    import torch
    res = HotPathResult()
    t = res.timings
    dev = torch.device('cuda', ctx.device)
    n = counts.n
    with torch.cuda.stream(ctx.torch_stream):
        # structured-array fields (SeqOrder.lengths()) are strided views: make them contiguous first
        sites_t = torch.as_tensor(sites if torch.is_tensor(sites) else np.ascontiguousarray(sites, dtype=np.float64),
                                  device=dev).to(torch.float64)
        lengths_t = torch.as_tensor(lengths if torch.is_tensor(lengths) else np.ascontiguousarray(lengths, dtype=np.int64),
                                    device=dev).to(torch.int64)

        t0 = time.perf_counter()
        work = counts
        n2o = o2n = None
        # row-sharded run (bin3c_amd.dist.DistSession installed on this context): every rank holds the caller's map, orders
        # it, and then forms and works on ITS ROW BLOCK of the ordered map only -- mask, balancing, sub-map and Louvain touch
        # 1/world of the entries per rank
        sess = getattr(ctx, 'dist_session', None)
        block = None
        if sess is not None:
            sess.set_starts(None)                  # the ordering passes run on the un-ordered map: equal row ranges
        if reorder and n >= REORDER_MIN_N and counts.nnz >= REORDER_MIN_NNZ:
            try:
                n2o = torch.empty(n, dtype=torch.int32, device=dev)
                o2n = torch.empty(n, dtype=torch.int32, device=dev)
                ctx._check(ctx.lib.b3c_locality_order(ctx.h, counts.h, _native._ptr(n2o), _native._ptr(o2n)))
                if sess is not None and sess.world > 1 and sess.n == n:
                    # row blocks of the ORDERED map, balanced by stored entries: every rank holds the whole raw
                    # map, hence all row lengths, and the same order -- the ranges need no exchange
                    if balance_rows:
                        from .dist import balanced_starts
                        lens = ctx.row_nnz(counts, n2o, torch.empty(n, dtype=torch.int64, device=dev))
                        sess.set_starts(balanced_starts(lens, sess.world))
                    block = (sess.r0, sess.r1)
                # canonical (columns re-sorted inside each row): unsorted rows cost the SpMV 12 % more than the sort saves
                work = ctx.permute(counts, n2o, rows=block)
            except Exception:
                if adopt_counts:
                    counts.free()              # the result never took it over
                raise
            if adopt_counts:
                counts.free()                  # handed over: from here on the ordered copy (a rank's row block when sharded) is the map
            # the per-contig tables in the working order (libb3c gathers: torch is plumbing here, not arithmetic)
            sites_t = ctx.vec_permute(sites_t, n2o, torch.empty_like(sites_t))
            lengths_t = ctx.vec_permute(lengths_t, n2o, torch.empty_like(lengths_t))
        res._n2o, res._o2n = n2o, o2n
        res.row_block, res.work_nnz = block, work.nnz
        t['order'] = time.perf_counter() - t0

        try:
            # contact_map.py:933 + :939 (-> sparse_utils.py:90): site normalisation folded into the balancing of the
            # raw counts (8 bytes per entry instead of 12), then one pass writes the balanced map: normalisation
            # exactly as the reference forms it and X A X (sparse_utils.py:223-224).  A bitwise-symmetric A (the
            # normal case: counts are symmetric by construction) is written in graph form so that Louvain needs no
            # mirror lookups; otherwise exactly as scipy does.
            t0 = time.perf_counter()
            x_w = torch.empty(n, dtype=torch.float64, device=dev)
            res.kr_strict = False
            if strict_kr is not False and block is None and work.nnz < REORDER_MIN_NNZ:
                # small map: the reference's own formulation (explicit fp64 matrix) -- see strict_kr
                nm = ctx.site_norm(work, sites_t)
                try:
                    rc, _, res.kr_info = ctx.kr_balance(nm, tol=tol, max_iter=max_iter, x_out=x_w, check=False)
                finally:
                    nm.free()
                res.kr_strict = True
            else:
                rc, _, res.kr_info = ctx.kr_balance_counts(work, sites_t, tol=tol, max_iter=max_iter, x_out=x_w, check=False)
            if strict_kr is True and not res.kr_strict and rc in (0, _native.E_NOCONV) and res.kr_info.n_iter >= STRICT_KR_ITER:
                first = res.kr_info
                nm = ctx.site_norm(work, sites_t)
                try:
                    rc, _, res.kr_info = ctx.kr_balance(nm, tol=tol, max_iter=max_iter, x_out=x_w, check=False)
                finally:
                    nm.free()
                # what the count pass established about the INPUT stays (the fp64 pass sees the normalised matrix)
                res.kr_info.bitwise_symmetric = first.bitwise_symmetric
                res.kr_strict = True
            ctx._check(rc)
            t['kr'] = time.perf_counter() - t0
            # set_primary_acceptance_mask (contact_map.py:888-905).  After the balancing: its pass over the counts leaves the
            # off-diagonal maximum of every row with the matrix, so the signal costs no pass of its own (the mask is first
            # needed by the sub-map below; the balancing works on the whole map, contact_map.py:929-945)
            t0 = time.perf_counter()
            signal_w = None
            if mask is None:
                signal_w = torch.empty(n, dtype=torch.int32, device=dev)          # uint32 bit pattern
                ctx.max_offdiag_u32(work, assume_symmetric=True, out=signal_w)
                if block is not None:
                    ctx.dist_allgather_i32(signal_w)           # every rank scanned its rows
                mask_w = ctx.acceptance_mask(lengths_t, signal_w, min_len, min_sig, out=torch.empty(n, dtype=torch.uint8, device=dev))
            else:
                mask_w = torch.as_tensor(np.asarray(mask, dtype=np.uint8), device=dev)
                if n2o is not None:
                    mask_w = ctx.vec_permute(mask_w, n2o, torch.empty_like(mask_w))
            t['mask'] = time.perf_counter() - t0

        except Exception:
            if work is not counts:
                work.free()
            if adopt_counts:
                counts.free()
            raise
        # the balanced map (normalisation exactly as the reference forms it, then X A X, sparse_utils.py:223-224; graph
        # form when A is bitwise symmetric so that Louvain needs no mirror lookups) is formed on demand from these
        graph_form = bool(res.kr_info.bitwise_symmetric)
        res._lazy = (ctx, work, sites_t, x_w, graph_form, work is not counts or adopt_counts)
        if adopt_counts and work is not counts:
            counts.free()                      # the ordered copy is what the result works on
        if work is counts and keep and not adopt_counts:
            res.processed                      # the caller owns `counts` and may free it: materialise now
            res._lazy = None

        # mask, signal and x are final here: their way back to the host (gather into the caller's numbering, copy into
        # pinned memory) runs on a side stream while the cluster stage keeps libb3c's stream busy; only the labels are
        # left for the end
        early = None
        if cluster:
            ready = torch.cuda.Event()
            ready.record(ctx.torch_stream)          # libb3c's stream: mask and signal are enqueued on it, not waited for
            early = _download_worker(ctx).submit(_start_download, ctx, o2n, mask_w, signal_w, x_w, ready)

        labels_t = None
        if cluster:
            labels_t = _cluster_stage(ctx, res, mask_w, n_sub)

        # results to the host, in the caller's numbering
        t0 = time.perf_counter()
        early = _start_download(ctx, o2n, mask_w, signal_w, x_w) if early is None else early.result()
        h_labels = None
        if labels_t is not None:
            h_labels = torch.empty(labels_t.shape, dtype=labels_t.dtype, pin_memory=True)
            h_labels.copy_(labels_t, non_blocking=True)
        h_mask, h_signal, h_x, dl = early
        dl.synchronize()
        torch.cuda.current_stream().synchronize()
        res.mask = h_mask.numpy().astype(bool)
        if h_signal is not None:
            res.signal = h_signal.numpy().view(np.uint32)
        res.x = h_x.numpy()
        if h_labels is not None:
            res.labels = h_labels.numpy()
        t['download'] = time.perf_counter() - t0
    if not keep:
        res.free()
    return res


def _download_worker(ctx):
    """one helper thread per context: the dozen torch calls of _start_download cost ~0.5 ms of HOST time, during which libb3c's
    stream would sit idle between the balancing and the sub-map (C2: 7 % of a step); libb3c and torch both release the GIL"""
    ex = getattr(ctx, '_dl_worker', None)
    if ex is None:
        from concurrent.futures import ThreadPoolExecutor
        ex = ctx._dl_worker = ThreadPoolExecutor(max_workers=1, thread_name_prefix='b3c-download')
    return ex


def _start_download(ctx, o2n, mask_w, signal_w, x_w, ready=None):
    """Enqueue, on the context's side stream, the per-contig results' way to the host: back to the caller's numbering, then
    asynchronous copies into pinned host tensors (torch caches pinned blocks, so a warm step allocates nothing).
    -> (mask, signal | None, x, stream): host tensors valid once the stream is synchronised."""
    import torch
    dl = getattr(ctx, '_dl_stream', None)
    if dl is None:
        dl = ctx._dl_stream = torch.cuda.Stream(device=ctx.device)
    if ready is None:
        ready = torch.cuda.Event()
        ready.record(ctx.torch_stream)
    # (this may run on the helper thread, whose current device and stream are not the caller's)
    with torch.cuda.device(ctx.device), torch.cuda.stream(dl):
        dl.wait_event(ready)
        back = o2n.long() if o2n is not None else None
        outs = []
        for src in (mask_w, signal_w, x_w):
            if src is None:
                outs.append(None)
                continue
            src.record_stream(dl)
            d = src[back] if back is not None else src
            h = torch.empty(d.shape, dtype=d.dtype, pin_memory=True)
            h.copy_(d, non_blocking=True)
            outs.append(h)
    return outs[0], outs[1], outs[2], dl


def _cluster_stage(ctx, res, mask_w, n_sub=0):
    """get_subspace + to_graph scaling + Louvain on the device-resident balanced map; mask_w in the map's row order."""
    import torch
    t = res.timings
    t0 = time.perf_counter()
    if res.sub is not None:
        res.sub.free()
        res.sub = None
    if res._processed is not None:
        if bool(mask_w.all().item()):                                      # contact_map.py:977: compress only if needed
            sub = res._processed
        else:
            sub = ctx.compress(res._processed, mask_w)
            res.sub = sub
    else:
        # get_subspace of the balanced map without writing the balanced map: values straight from the counts (an all-true
        # mask gives the whole balanced map, bit for bit what `res.processed` would form: no test of the mask beforehand)
        _c, work, sites_t, x_w, graph_form, _owned = res._lazy
        sub = ctx.normalise_scale_compress(work, sites_t, x_w, mask_w, graph_form=graph_form)
        res.sub = sub
    # a rank of a row-sharded run decides from the GLOBAL entry count: its own block may be empty (all its rows masked) while
    # the others still cluster, and the calls below are collective over the ranks
    res.sub_nnz = sub.nnz
    nnz_all = sub.nnz if res.row_block is None else int(ctx.dist_max(float(sub.nnz)))
    if sub.n == 0 or nnz_all == 0:
        # nothing passed the acceptance mask: the reference raises NoneAcceptedException one level up
        # (contact_map.py:927); at this level the answer is simply "no accepted contigs, no labels"
        res.scl = None
        res.lv_info = _native.LouvainInfo()
        t['compress'] = time.perf_counter() - t0
        t['louvain'] = 0.0
        return torch.full((sub.n,), -1, dtype=torch.int32, device=mask_w.device)
    res.scl = 1.0 / ctx.dist_max(ctx.max_value(sub))                       # cluster.py:316 (row blocks: over the ranks)
    t['compress'] = time.perf_counter() - t0

    t0 = time.perf_counter()
    dev = mask_w.device
    labels_w = torch.empty(sub.n, dtype=torch.int32, device=dev)
    _, res.lv_info = ctx.louvain(sub, res.scl, n_sub=n_sub, labels_out=labels_w)   # cluster.py:319-321 + louvain.py:70
    ids = torch.empty(sub.n, dtype=torch.int32, device=dev)
    has_ids = ctx.lib.b3c_mat_ids(ctx.h, sub.h, _native._ptr(ids))
    if has_ids == 1:
        labels_t = labels_w
    else:
        ctx._check(has_ids)
        labels_t = ctx.vec_permute(labels_w, ids, torch.empty_like(labels_w), scatter=True)   # back to gapless caller-side ids
    t['louvain'] = time.perf_counter() - t0
    return labels_t


def cluster_prepared(ctx, res, mask, n_sub=0):
    """Cluster a map that :func:`run_hot_path` (cluster=False, keep=True) left in HBM, under a caller-side mask."""
    import torch
    dev = torch.device('cuda', ctx.device)
    with torch.cuda.stream(ctx.torch_stream):
        mask_w = torch.as_tensor(np.asarray(mask, dtype=np.uint8), device=dev)
        if res._n2o is not None:
            mask_w = ctx.vec_permute(mask_w, res._n2o, torch.empty_like(mask_w))
        labels_t = _cluster_stage(ctx, res, mask_w, n_sub)
        res.mask = np.asarray(mask, dtype=bool)
        res.labels = labels_t.cpu().numpy()
    return res


def device_subspace(ctx, res, mask, order=None):
    """get_subspace (contact_map.py:947-999) without leaving HBM: the balanced map under a caller-side mask, rows in
    gapless caller order or, with `order` (SeqOrder.gapless_positions()), in that order.  -> _native.Mat (caller frees)."""
    import torch
    dev = torch.device('cuda', ctx.device)
    with torch.cuda.stream(ctx.torch_stream):
        mask_w = torch.as_tensor(np.asarray(mask, dtype=np.uint8), device=dev)
        if res._n2o is not None:
            mask_w = ctx.vec_permute(mask_w, res._n2o, torch.empty_like(mask_w))
        if res._processed is not None or res._lazy is None:
            sub = ctx.compress(res.processed, mask_w)
        else:
            _c, work, sites_t, x_w, graph_form, _owned = res._lazy
            sub = ctx.normalise_scale_compress(work, sites_t, x_w, mask_w, graph_form=graph_form)
        ids = ctx.mat_ids(sub)                       # gapless caller-side id of every row, None = already in caller order
        m = sub.n
        if ids is None and order is None:
            return sub
        inv = np.arange(m, dtype=np.int64)
        if ids is not None:
            inv = np.empty(m, dtype=np.int64)
            inv[ids] = np.arange(m)
        perm = inv if order is None else inv[np.asarray(order, dtype=np.int64)]
        try:
            return ctx.permute(sub, perm.astype(np.int32))
        finally:
            sub.free()


def processed_to_host(ctx, res):
    """processed_map (contact_map.py:945) as host CSR arrays in the caller's numbering, with the values scipy holds:
    X.T * (A * X) entry by entry (sparse_utils.py:223-224).  The device copy used for clustering may be in graph form (both
    triangles carry the lower-triangle value, an ulp or two off in the upper one), so the exported map is formed apart."""
    import torch
    n2o, o2n = res.perm
    own = None
    if res._lazy is not None:
        _c, work, sites_t, x_w, graph_form, _owned = res._lazy
        if graph_form or res._processed is None:
            with torch.cuda.stream(ctx.torch_stream):
                own = ctx.normalise_scale(work, sites_t, x_w, graph_form=False)
    m = own if own is not None else res.processed
    try:
        if n2o is None:
            return m.download()
        back = ctx.permute(m, o2n)          # row k of the result = row o2n[k] of the ordered matrix = caller row k
        try:
            return back.download()
        finally:
            back.free()
    finally:
        if own is not None:
            own.free()
