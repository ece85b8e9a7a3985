"""
Host-side mirror of the reference's ``mzd/contact_map.py`` for the cluster hot path: ``SeqInfo``,
``SeqOrder`` and the ``ContactMap`` methods that sit between ``load_object`` and the solver
(set_primary_acceptance_mask :856, prepare_seq_map :911, get_subspace :947, _bisto_seq :1087,
_get_sites :1103, _norm_seq :1110).  Attribute names are the reference's (contact_map.py:492-518,
:567-572), because the pickled object IS the interface between ``mkmap`` and ``cluster``.

What differs: the arithmetic runs in libb3c on the GPU (no CPU fallback), the per-contig loops of
SeqOrder are vectorised numpy (integer work, identical results -- tests/test_host_mirror.py checks
them against values produced by the reference's own class text), and BAM/FASTA ingest
(contact_map.py:488-809) is out of scope: build instances with :meth:`ContactMap.from_arrays`.
"""
import logging
from collections import namedtuple

import numpy as np
import scipy.sparse as sp

from . import sparse_utils
from .exceptions import NoneAcceptedException

logger = logging.getLogger(__name__)

SeqInfo = namedtuple('SeqInfo', ['offset', 'refid', 'name', 'length', 'sites'])


class SeqOrder(object):
    """Order, orientation and mask state of the contigs (contact_map.py:159-483)."""

    FORWARD = 1
    REVERSE = -1
    ACCEPTED = True
    EXCLUDED = False

    STRUCT_TYPE = np.dtype([('pos', np.int32), ('ori', np.int8), ('mask', np.bool_), ('length', np.int32)])
    INDEX_TYPE = np.dtype([('index', np.int32), ('ori', np.int8)])

    def __init__(self, seq_info):
        n = len(seq_info)
        self.order = np.zeros(n, dtype=SeqOrder.STRUCT_TYPE)
        self.order['pos'] = np.arange(n, dtype=np.int32)
        self.order['ori'] = SeqOrder.FORWARD
        self.order['mask'] = SeqOrder.ACCEPTED
        self.order['length'] = [si.length for si in seq_info]
        self._positions = None
        self._update_positions()

    @staticmethod
    def asindex(_ord):
        """Plain ids -> INDEX_TYPE records, all forward (contact_map.py:193-201)."""
        assert isinstance(_ord, (list, np.ndarray)), 'input must be a list or ndarray'
        out = np.empty(len(_ord), dtype=SeqOrder.INDEX_TYPE)
        out['index'] = _ord
        out['ori'] = 1
        return out

    def _update_positions(self):
        """Masked contigs last, then by current position (contact_map.py:203-213), without the Python loop."""
        srt = np.lexsort([self.order['pos'], ~self.order['mask']])
        self.order['pos'][srt] = np.arange(len(srt), dtype=np.int32)
        self._positions = np.argsort(self.order['pos'])

    def remap_gapless(self, gapless_indices):
        """Dense ids (solver output) -> original, gapped surrogate ids (contact_map.py:215-251)."""
        mask = self.order['mask']
        shift = np.cumsum(~mask)[mask]
        if isinstance(gapless_indices, np.ndarray) and gapless_indices.dtype == SeqOrder.INDEX_TYPE:
            out = gapless_indices.copy()
            out['index'] = gapless_indices['index'] + shift[gapless_indices['index']]
            return out
        idx = np.asarray(gapless_indices)
        if idx.size == 0:
            return np.array([])
        return idx + shift[idx]

    def accepted_positions(self, copy=True):
        return self.all_positions(copy=copy)[:self.count_accepted()]

    def all_positions(self, copy=True):
        return self._positions.copy() if copy else self._positions

    @staticmethod
    def double_order(_ord):
        _ord = np.asarray(_ord)
        return np.stack((2 * _ord, 2 * _ord + 1), axis=1).ravel()

    def gapless_positions(self):
        """Positional order of the accepted contigs as dense ids (contact_map.py:285-300)."""
        gap_shift = np.cumsum(~self.order['mask'])
        _p = np.argsort(self.order['pos'])[:self.count_accepted()]
        return _p - gap_shift[_p]

    def set_mask_only(self, _mask):
        """Replace the mask, keep relative order (contact_map.py:302-316)."""
        _mask = np.asarray(_mask, dtype=np.bool_)
        assert len(_mask) == len(self.order), 'supplied mask must be the same length as existing order'
        self.order['mask'] = _mask
        self._update_positions()

    def set_order_only(self, _ord, implicit_excl=False):
        assert isinstance(_ord, (list, np.ndarray)), 'Wrong type supplied, order must be a list or ndarray'
        _ord = np.ravel(np.asarray(_ord))
        self.set_order_and_orientation(SeqOrder.asindex(_ord), implicit_excl=implicit_excl)

    def set_order_and_orientation(self, _ord, implicit_excl=False):
        """contact_map.py:334-378."""
        assert _ord.dtype == SeqOrder.INDEX_TYPE, 'Wrong type supplied, _ord should be of INDEX_TYPE'
        if len(_ord) < len(self.order):
            assert implicit_excl, 'Use implicit_excl=True for automatic handling ' \
                                  'of orders only mentioning accepted sequences'
            assert len(_ord) == self.count_accepted(), 'new order must mention all currently accepted sequences'
            mentioned = set(_ord['index'].tolist())
            assert not (mentioned & set(self.excluded().tolist())), \
                'new order and excluded must not overlap when using implicit assignment'
            assert not (mentioned ^ set(self.accepted().tolist())), 'incomplete new order supplied, missing accepted ids'
            self.order['pos'][_ord['index']] = np.arange(len(_ord), dtype=np.int32)
            self.order['ori'][_ord['index']] = _ord['ori']
            _mask = np.zeros(len(self.order), dtype=np.bool_)
            _mask[_ord['index']] = True
            self.set_mask_only(_mask)
        else:
            assert len(_ord) == len(self.order), 'new order was a different length'
            self.order['pos'][_ord['index']] = np.arange(len(_ord), dtype=np.int32)
            self.order['ori'][_ord['index']] = _ord['ori']
        self._update_positions()

    def accepted_order(self):
        idx = np.flatnonzero(self.order['mask'])
        return SeqOrder.asindex(idx)

    def mask_vector(self):
        return self.order['mask']

    def mask(self, _id):
        self.order['mask'][_id] = False
        self._update_positions()

    def count_accepted(self):
        return int(self.order['mask'].sum())

    def count_excluded(self):
        return len(self.order) - self.count_accepted()

    def accepted(self):
        return np.flatnonzero(self.order['mask'])

    def excluded(self):
        return np.flatnonzero(~self.order['mask'])

    def flip(self, _id):
        self.order['ori'][_id] *= -1

    def lengths(self, exclude_masked=False):
        if exclude_masked:
            return self.order['length'][self.order['mask']]
        return self.order['length']

    def shuffle(self):
        np.random.shuffle(self.order['pos'])
        self._update_positions()

    def before(self, a, b):
        assert a != b, 'Surrogate ids must be different'
        return self.order['pos'][a] < self.order['pos'][b]

    def intervening(self, a, b):
        assert a != b, 'Surrogate ids must be different'
        pa, pb = sorted((self.order['pos'][a], self.order['pos'][b]))
        return int(np.sum(self.order['length'][self._positions[pa + 1:pb]]))


class ContactMap(object):
    """
    The pickled object that ``mkmap`` hands to ``cluster`` (contact_map.py:486).  Only the cluster-side
    methods are implemented; instances come from :meth:`from_arrays` (synthetic maps, converted
    reference pickles) because BAM/FASTA ingest is out of scope.
    """

    def __init__(self, bam_file=None, enzymes=None, seq_file=None, min_insert=None, min_mapq=0, min_len=0, min_sig=1,
                 min_extent=0, min_size=0, max_fold=None, random_seed=None, strong=None, bin_size=None, tip_size=None,
                 precount=False):
        # attribute surface of contact_map.py:492-518
        self.strong = strong
        self.bam_file = bam_file
        self.bin_size = bin_size
        self.min_mapq = min_mapq
        self.min_insert = min_insert
        self.min_len = min_len
        self.min_sig = min_sig
        self.min_extent = min_extent
        self.min_size = min_size
        self.max_fold = max_fold
        self.random_state = np.random.RandomState(random_seed)
        self.seq_info = []
        self.seq_map = None
        self.seq_file = seq_file
        self.grouping = None
        self.extent_map = None
        self.order = None
        self.tip_size = tip_size
        self.precount = precount
        self.total_reads = None
        self.cov_info = None
        self.primary_acceptance_mask = None
        self.bisto_scale = None
        self.seq_analyzer = None
        self.enzymes = enzymes
        self.total_len = 0
        self.total_seq = 0
        self.current_mask = None
        self._processed_host = None     # scipy CSR in the caller's numbering (what the reference caches, :945)
        self._hot = None                # HotPathResult with device-resident matrices (not pickled)
        self._ctx = None
        if bam_file is not None:
            raise NotImplementedError('BAM/FASTA ingest (contact_map.py:520-600) is outside the accelerated hot path; '
                                      'use ContactMap.from_arrays')

    # ---- construction -------------------------------------------------------------------------
    @classmethod
    def from_arrays(cls, seq_map, lengths, sites, names=None, min_len=1000, min_sig=5, min_extent=50000, **kwargs):
        """
        :param seq_map: symmetric scipy sparse uint32 matrix, the output contract of Sparse2DAccumulator.get_coo
                        (sparse_utils.py:246-266)
        """
        cm = cls(min_len=min_len, min_sig=min_sig, min_extent=min_extent, **kwargs)
        n = seq_map.shape[0]
        names = names if names is not None else ['seq{}'.format(i) for i in range(n)]
        offs = np.concatenate(([0], np.cumsum(lengths)[:-1])) if n else np.zeros(0)
        cm.seq_info = [SeqInfo(int(offs[i]), i, names[i], int(lengths[i]), int(sites[i])) for i in range(n)]
        cm.seq_map = seq_map.tocoo().astype(np.uint32)
        cm.total_len = int(np.sum(lengths))
        cm.total_seq = n
        cm.current_mask = np.ones(n, dtype=np.bool_)
        cm.order = SeqOrder(cm.seq_info)
        return cm

    def __setstate__(self, state):
        """Also accepts the attribute dictionary of a map pickled by the reference (old-style class, no private
        fields): its cached ``processed_map`` becomes the host copy, device handles start empty."""
        from .io_utils import revive
        state = dict(state)
        for k in ('seq_map', 'processed_map', 'extent_map'):          # scipy matrices of a reference-era (scipy < 1.8) pickle
            if k in state:
                state[k] = revive(state[k])
        if 'processed_map' in state:
            state['_processed_host'] = state.pop('processed_map')
        for k in ('_processed_host', '_hot', '_ctx'):
            state.setdefault(k, None)
        self.__dict__.update(state)
        if isinstance(self.order, SeqOrder) and self.order.order.dtype != SeqOrder.STRUCT_TYPE:
            self.order.order = self.order.order.astype(SeqOrder.STRUCT_TYPE)

    def __getstate__(self):
        """The attribute dictionary the reference's ContactMap has (contact_map.py:488-529): the balanced map sits under its
        reference name ``processed_map`` (materialised on the host before the device handles go away), nothing private."""
        d = self.__dict__.copy()
        d['processed_map'] = self.processed_map
        for k in ('_processed_host', '_hot', '_ctx'):
            d.pop(k, None)
        return d

    # ---- device plumbing ----------------------------------------------------------------------
    def _context(self):
        if self._ctx is None:
            from ..pipeline import make_context
            self._ctx = make_context(0)
        return self._ctx

    def _counts_csr(self):
        c = self.seq_map.tocsr()
        c.sum_duplicates()
        c.sort_indices()
        return c

    def _drop_device(self):
        if self._hot is not None:
            self._hot.free()
            self._hot = None

    @property
    def processed_map(self):
        """Balanced full map as scipy CSR (contact_map.py:945); pulled from HBM on first access."""
        if self._processed_host is None and self._hot is not None and self._hot.processed is not None:
            from ..pipeline import processed_to_host
            ip, ix, data = processed_to_host(self._context(), self._hot)
            n = ip.shape[0] - 1
            self._processed_host = sp.csr_matrix((data, ix, ip), shape=(n, n))
        return self._processed_host

    def has_processed_map(self):
        """`processed_map is not None` without pulling a device-resident map to the host"""
        return self._processed_host is not None or (self._hot is not None and self._hot.has_processed())

    @processed_map.setter
    def processed_map(self, value):
        self._processed_host = value
        if value is None:
            self._drop_device()

    # ---- reference surface --------------------------------------------------------------------
    def map_weight(self):
        return self.seq_map.sum()

    def is_empty(self):
        return self.map_weight() == 0

    def is_tipbased(self):
        return self.tip_size is not None

    def get_primary_acceptance_mask(self):
        assert self.primary_acceptance_mask is not None, 'Primary acceptance mask has not be initialized'
        return self.primary_acceptance_mask.copy()

    def set_primary_acceptance_mask(self, min_len=None, min_sig=None, max_fold=None, update=False):
        """contact_map.py:856-909: mask = (length >= min_len) & (max off-diagonal count >= min_sig)."""
        assert max_fold is None, 'Filtering on max_fold is currently disabled'
        if not min_len:
            min_len = self.min_len
        if not min_sig:
            min_sig = self.min_sig
        assert min_len, 'Filtering criteria min_len is None'
        assert min_sig, 'Filtering criteria min_sig is None'
        logger.debug('Setting primary acceptance mask with '
                     'filtering criterion min_len: {} min_sig: {}'.format(min_len, min_sig))
        if not update and self.primary_acceptance_mask is not None:
            logger.debug('Using existing mask')
            return self.get_primary_acceptance_mask()
        assert not self.is_tipbased(), 'tip-based maps are outside the accelerated path'
        acceptance_mask = np.ones(self.total_seq, dtype=np.bool_)
        _mask = self.order.lengths() >= min_len
        logger.debug('Minimum length threshold removing: {}'.format(self.total_seq - _mask.sum()))
        acceptance_mask &= _mask
        signal = sparse_utils.max_offdiag(self.seq_map, ctx=self._context())
        _mask = signal >= min_sig
        logger.debug('Minimum signal threshold removing: {}'.format(self.total_seq - _mask.sum()))
        acceptance_mask &= _mask
        self.primary_acceptance_mask = acceptance_mask
        logger.debug('Accepted sequences: {}'.format(self.primary_acceptance_mask.sum()))
        return self.get_primary_acceptance_mask()

    def _get_sites(self):
        """contact_map.py:1103-1108: unobserved sites count as one."""
        _sites = np.array([si.sites for si in self.seq_info], dtype=np.float64)
        _sites[_sites == 0] = 1
        return _sites

    def prepare_seq_map(self, norm=True, bisto=False, mean_type='geometric'):
        """
        contact_map.py:911-945.  With norm and bisto (the cluster path, cluster.py:308-309) the whole chain
        runs device resident and leaves the balanced map in HBM for get_subspace / cluster_map.
        """
        logger.info('Preparing sequence map with full dimensions: {}'.format(self.seq_map.shape))
        _mask = self.get_primary_acceptance_mask()
        self.order.set_mask_only(_mask)
        if self.order.count_accepted() < 1:
            raise NoneAcceptedException()
        assert mean_type == 'geometric' and norm, 'only site-based normalisation is on the accelerated path'
        self.processed_map = None
        ctx = self._context()
        c = self._counts_csr()
        if bisto:
            from ..pipeline import run_hot_path
            counts = ctx.upload(c.indptr, c.indices, c.data)
            res = run_hot_path(ctx, counts, self._get_sites(), self.order.lengths(), self.min_len, self.min_sig,
                               cluster=False, keep=True, mask=_mask, adopt_counts=True)
            self._report_kr(res.kr_info)
            self.bisto_scale = res.x
            logger.debug('Map balanced')
            self._hot = res
        else:
            counts = ctx.upload(c.indptr, c.indices, c.data)
            a = ctx.site_norm(counts, self._get_sites())
            counts.free()
            _, _, data = a.download(structure=False)
            a.free()
            self._processed_host = sp.csr_matrix((data, c.indices, c.indptr), shape=c.shape)
            logger.debug('Map normalized')

    @staticmethod
    def _report_kr(info):
        """The log lines kr_biostochastic writes (sparse_utils.py:113,121,218,221)."""
        ulog = logging.getLogger(sparse_utils.__name__)
        if info.n_zero_diag > 0:
            ulog.warning('treating {} zeros on diagonal as ones'.format(info.n_zero_diag))
        if not info.is_symmetric:
            ulog.warning('input matrix is expected to be fully symmetric')
        ulog.debug('It took {} iterations to achieve bistochasticity'.format(info.n_iter))
        if info.hit_max_iter:
            ulog.warning('Warning: maximum number of iterations ({}) reached without convergence'.format(1000))

    def _norm_seq(self, _map, tip_based=False, use_sites=True, mean_type='geometric'):
        """contact_map.py:1110-1145 (site-based branch): data_ij *= 1 / (sites_i * sites_j)."""
        assert use_sites and not tip_based, 'only site-based normalisation of 2-D maps is implemented'
        ctx = self._context()
        c = _map.tocsr().astype(np.uint32)
        c.sum_duplicates()
        c.sort_indices()
        counts = ctx.upload(c.indptr, c.indices, c.data)
        a = ctx.site_norm(counts, self._get_sites())
        _, _, data = a.download(structure=False)
        counts.free()
        a.free()
        return sp.csr_matrix((data, c.indices, c.indptr), shape=c.shape).tocoo()

    def _bisto_seq(self, _map):
        """contact_map.py:1087-1101."""
        logger.debug('Balancing contact map')
        assert not self.is_tipbased(), 'tip-based maps are outside the accelerated path'
        return sparse_utils.kr_biostochastic(_map, ctx=self._context())

    def get_subspace(self, permute=False, external_mask=None, marginalise=False, flatten=True, dtype=np.float64):
        """contact_map.py:947-999 for 2-D maps: masked (and optionally reordered) copy of the processed map."""
        assert (not marginalise and not flatten) or np.logical_xor(marginalise, flatten), \
            'marginalise and flatten are mutually exclusive'
        _map = self.processed_map.astype(dtype)
        if external_mask is not None:
            _mask = self.get_primary_acceptance_mask()
            logger.info('Beginning with sequences after primary filtering: {}'.format(_mask.sum()))
            _mask &= external_mask
            logger.info('Active sequences after applying external mask: {}'.format(_mask.sum()))
            self.order.set_mask_only(_mask)
        if self.order.count_accepted() < self.total_seq:
            _map = sparse_utils.compress(_map.tocoo(), self.order.mask_vector(), ctx=self._context())
            logger.info('After removing filtered sequences map dimensions: {}'.format(_map.shape))
        if permute:
            _map = self._reorder_seq(_map)
            logger.debug('Map reordered')
        return _map

    def _reorder_seq(self, _map, flatten=False):
        """contact_map.py:1066-1085: P A P^T with P[i, order[i]] = 1 (b3c_mat_permute on the device)."""
        assert sp.isspmatrix(_map), 'reordering expects a sparse matrix type'
        _order = self.order.gapless_positions()
        assert not (self.is_tipbased() and flatten), 'tip-based maps are outside the accelerated path'
        assert _map.shape[0] == _order.shape[0], 'supplied map and unmasked order are different sizes'
        ctx = self._context()
        c = sparse_utils._canonical_csr(_map, None if _map.dtype == np.uint32 else np.float64)
        mat = ctx.upload(c.indptr, c.indices, c.data)
        try:
            out = ctx.permute(mat, np.ascontiguousarray(_order, dtype=np.int32))
            try:
                ip, ix, data = out.download()
            finally:
                out.free()
        finally:
            mat.free()
        return sp.csr_matrix((data.astype(_map.dtype, copy=False), ix, ip), shape=_map.shape)

    MAX_RASTER_CELLS = 400 * 1000 * 1000      # dense float64 raster limit (3.2 GB)

    def plot_matrix(self, simple=True, norm=True, permute=False, zero_diag=None, alpha=0.01, max_image_size=None,
                    tick_locs=None, flatten=False):
        """
        The raster behind ContactMap.plot (contact_map.py:1316-1357) without the drawing: masked / permuted
        subspace of the processed map, x100, optionally block-reduced on the device (sparse_utils.downsample),
        densified and log(x + alpha) transformed.

        :return: (2-D ndarray of log intensities, tick_locs rescaled like contact_map.py:1348)
        """
        assert simple or self.bin_size is None, 'extent (binned) maps are outside the accelerated path'
        if not self.has_processed_map():
            self.prepare_seq_map(norm=norm, bisto=True)
        if self._hot is not None and self._hot.has_processed() and not self.is_tipbased():
            # the balanced map is in HBM: mask, reorder and block-reduce there, only the raster crosses PCIe
            from ..pipeline import device_subspace
            ctx = self._context()
            dm = device_subspace(ctx, self._hot, self.order.mask_vector(),
                                 self.order.gapless_positions() if permute else None)
            try:
                full_size = (dm.n, dm.n)
                if max_image_size is not None and dm.n > max_image_size:
                    reduce_factor = int(np.ceil(dm.n / float(max_image_size)))
                    logger.info('Full {} image reduction factor: {}'.format(full_size, reduce_factor))
                    small = ctx.downsample(dm, reduce_factor, 'mean', skip_diag=zero_diag is None)
                    try:
                        ip, ix, data = small.download()
                        nb = small.n
                    finally:
                        small.free()
                    if tick_locs is not None:
                        tick_locs = np.floor(np.asarray(tick_locs).astype(float) / reduce_factor)
                    logger.info('Map has been reduced from {} to {}'.format(full_size, (nb, nb)))
                    _map = sp.csr_matrix((data * 100, ix, ip), shape=(nb, nb))
                else:
                    if dm.n * dm.n > self.MAX_RASTER_CELLS:
                        raise MemoryError('a {} x {} raster would not fit in host memory: set max_image_size'.format(*full_size))
                    ip, ix, data = dm.download()
                    _map = sp.csr_matrix((data, ix, ip), shape=full_size).tolil()
                    if zero_diag is None:
                        _map.setdiag(0)
                    _map = _map.tocsr() * 100
            finally:
                dm.free()
        else:
            _map = self.get_subspace(permute=permute, marginalise=False if flatten else True, flatten=flatten).tolil()
            if zero_diag is None:
                _map.setdiag(0)
            _map = _map.tocsr()
            _map *= 100
            if max_image_size is not None:
                full_size = _map.shape
                if np.max(full_size) > max_image_size:
                    reduce_factor = int(np.ceil(np.max(full_size) / float(max_image_size)))
                    logger.info('Full {} image reduction factor: {}'.format(full_size, reduce_factor))
                    _map = sparse_utils.downsample(_map, reduce_factor, ctx=self._context())
                    if tick_locs is not None:
                        tick_locs = np.floor(np.asarray(tick_locs).astype(float) / reduce_factor)
                    logger.info('Map has been reduced from {} to {}'.format(full_size, _map.shape))
        if _map.shape[0] * _map.shape[1] > self.MAX_RASTER_CELLS:
            raise MemoryError('a {} x {} raster would not fit in host memory: set max_image_size'.format(*_map.shape))
        _map = _map.toarray()
        if zero_diag:
            logger.debug('Removing diagonal')
            np.fill_diagonal(_map, 0)
        return np.log(_map + alpha), tick_locs

    def plot(self, fname, simple=False, tick_locs=None, tick_labs=None, norm=True, permute=False, pattern_only=False,
             dpi=180, width=25, height=22, zero_diag=None, alpha=0.01, robust=False, max_image_size=None,
             flatten=False):
        """contact_map.py:1285-1382. The raster comes from :meth:`plot_matrix`; drawing needs matplotlib, which is the
        reference's presentation layer and not part of the accelerated path -- without it the raster is written as
        ``fname + '.npy'`` next to the tick positions so that it can be rendered elsewhere."""
        assert not pattern_only, 'pattern_only (spy) plots are outside the accelerated path'
        img, tick_locs = self.plot_matrix(simple=simple, norm=norm, permute=permute, zero_diag=zero_diag, alpha=alpha,
                                          max_image_size=max_image_size, tick_locs=tick_locs, flatten=flatten)
        try:
            import matplotlib
            matplotlib.use('Agg')
            import matplotlib.pyplot as plt
        except ImportError:
            np.save(fname + '.npy', img)
            if tick_locs is not None:
                np.save(fname + '.ticks.npy', np.asarray(tick_locs))
            logger.warning('matplotlib is not installed: raster saved as {}.npy'.format(fname))
            return
        fig = plt.figure()
        fig.set_figwidth(width)
        fig.set_figheight(height)
        ax = fig.add_subplot(111)
        ax.imshow(img, interpolation='nearest', aspect='equal')
        if tick_locs is not None:
            ax.set_xticks(np.asarray(tick_locs))
            ax.set_yticks(np.asarray(tick_locs))
            if tick_labs is not None:
                mid = (np.asarray(tick_locs)[:-1] + np.asarray(tick_locs)[1:]) / 2.0
                ax.set_xticks(mid, minor=True)
                ax.set_xticklabels(tick_labs, minor=True, rotation=45, ha='right', fontsize=10)
                ax.set_xticklabels([])
        fig.savefig(fname, dpi=dpi)
        plt.close(fig)
