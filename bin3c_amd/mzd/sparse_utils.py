"""
Host-side mirror of the reference's ``mzd/sparse_utils.py`` for the hot path: same function names,
arguments, return types, log lines and exceptions, with the arithmetic done by libb3c on the MI355X.

    downsample         mzd/sparse_utils.py:48-87
    kr_biostochastic   mzd/sparse_utils.py:90-224
    max_offdiag        mzd/sparse_utils.py:269-281
    compress           mzd/sparse_utils.py:284-314

There is no CPU fallback: without libb3c.so and a GPU these functions raise.
"""
import logging

import numpy as np
import scipy.sparse as scisp

from .. import _native

logger = logging.getLogger(__name__)


def _canonical_csr(m, dtype=None):
    """scipy matrix -> CSR with sorted indices and summed duplicates (what tolil()/tocsr() at
    sparse_utils.py:110-118 produce), never modifying the caller's object."""
    c = m.tocsr()
    if c is m:
        c = c.copy()
    if dtype is not None and c.dtype != dtype:
        c = c.astype(dtype)
    c.sum_duplicates()
    c.sort_indices()
    return c


def downsample(m, block_size, method='mean', ctx=None):
    """
    Perform a down-sampling of a 2D matrix (scipy.sparse or ndarray) by a factor of block_size in each dimension
    (sparse_utils.py:48-87). Block size must be an integer larger than 1. When employing mean method, zero padding
    on edges is not compensated for.

    :param m: a matrix (scipy.sparse or ndarray)
    :param block_size: an integer reduction factor
    :param method: per block mean or maximum.
    :return: scipy.sparse.csr_matrix
    """
    assert isinstance(m, (np.ndarray, scisp.spmatrix)), 'supplied array must be of type np.ndarray or scipy.spmatrix'
    assert block_size > 1 and isinstance(block_size, int), 'block_size must be an integer larger than 1'
    if method not in ('mean', 'max'):
        return m                       # the reference falls through both branches and returns its COO untouched
    ctx = ctx or _native.default_context()
    in_dtype = m.dtype
    c = _canonical_csr(scisp.csr_matrix(m), np.float64)
    rows, cols = c.shape
    side = max(rows, cols)
    if rows != cols:                   # libb3c maps are square: embed (the extra cells are implicit zeros, like padding)
        c.resize((side, side))
    mat = ctx.upload(c.indptr, c.indices, c.data)
    try:
        out = ctx.downsample(mat, block_size, method)
        try:
            ip, ix, data = out.download()
        finally:
            out.free()
    finally:
        mat.free()
    nb = ip.shape[0] - 1
    r = scisp.csr_matrix((data, ix, ip), shape=(nb, nb))
    out_shape = (-(-rows // block_size), -(-cols // block_size))
    if out_shape != r.shape:
        r = r[:out_shape[0], :out_shape[1]]
    return r.astype(in_dtype) if method == 'max' and in_dtype != np.float64 else r


def kr_biostochastic(m, tol=1e-6, x0=None, delta=0.1, Delta=3, max_iter=1000, ctx=None):
    """
    Normalise a matrix to be bistochastic using the Knight-Ruiz algorithm (sparse_utils.py:90-224).

    :param m: the input matrix (fully symmetric), any scipy.sparse type
    :param tol: precision tolerance
    :param x0: an initial guess -- only None is supported, as in the reference (``if not x0`` on an
               ndarray raises at sparse_utils.py:126)
    :param delta: how close balancing vector can get
    :param Delta: how far balancing vector can get
    :param max_iter: maximum number of iterations before abandoning.
    :return: tuple containing the bistochastic matrix (CSR) and the scale factors
    """
    assert scisp.isspmatrix(m), 'input matrix must be sparse matrix from scipy.spmatrix'
    assert m.shape[0] == m.shape[1], 'input matrix must be square'
    if x0 is not None:
        raise ValueError('x0 is not supported (the reference cannot evaluate `not x0` on an array either)')

    ctx = ctx or _native.default_context()
    c = _canonical_csr(m, np.float64)
    mat = ctx.upload(c.indptr, c.indices, c.data)
    try:
        rc, x, info = ctx.kr_balance(mat, tol, delta, Delta, max_iter, check=False)
        # re-emit the reference's log lines in its order (:113, :121)
        if info.n_zero_diag > 0:
            logger.warning('treating {} zeros on diagonal as ones'.format(info.n_zero_diag))
        if not info.is_symmetric:
            logger.warning('input matrix is expected to be fully symmetric')
        if rc == _native.E_NAN:
            raise RuntimeError('scale vector has developed invalid values (NANs)!')
        if rc == _native.E_NOCONV:
            raise RuntimeError('matrix balancing failed to converge in {} iterations'.format(info.n_iter))
        ctx._check(rc)
        logger.debug('It took {} iterations to achieve bistochasticity'.format(info.n_iter))
        if info.hit_max_iter:
            logger.warning('Warning: maximum number of iterations ({}) reached without convergence'.format(max_iter))
        ctx.scale_sym(mat, x)
        _, _, data = mat.download(structure=False)
    finally:
        mat.free()
    return scisp.csr_matrix((data, c.indices, c.indptr), shape=c.shape), x


def max_offdiag(_m, ctx=None):
    """
    Determine the maximum off-diagonal values of a given symmetric matrix (sparse_utils.py:269-281):
    the diagonal is zeroed, then the per-column maximum is returned (implicit zeros included).

    :param _m: a scipy.sparse matrix
    :return: the off-diagonal maximum values
    """
    assert scisp.isspmatrix(_m), 'Input matrix is not a scipy.sparse object'
    ctx = ctx or _native.default_context()
    # the reference's only caller passes the uint32 sequence map (sparse_utils.py:237, contact_map.py:899); other
    # non-negative integer maps are taken when their values fit, anything else is refused -- there is no host path
    if _m.dtype != np.uint32:
        c = _m.tocsr()
        if not np.issubdtype(c.dtype, np.integer) or (c.nnz and (c.data.min() < 0 or c.data.max() > 0xFFFFFFFF)):
            raise TypeError('max_offdiag runs on the device for uint32 count maps only (got {})'.format(_m.dtype))
        _m = c.astype(np.uint32)
    c = _canonical_csr(_m)
    mat = ctx.upload(c.indptr, c.indices, c.data)
    try:
        return ctx.max_offdiag_u32(mat, assume_symmetric=False)
    finally:
        mat.free()


def compress(_m, _mask, ctx=None):
    """
    Remove rows and columns using a 1d boolean mask (sparse_utils.py:284-314).

    :param _mask: True (keep), False (drop)
    :return: a coo_matrix of only the accepted rows/columns
    """
    assert scisp.isspmatrix(_m), 'Input matrix is not a scipy sparse matrix type'
    ctx = ctx or _native.default_context()
    _mask = np.asarray(_mask, dtype=bool)
    c = _canonical_csr(_m, None if _m.dtype == np.uint32 else np.float64)
    mat = ctx.upload(c.indptr, c.indices, c.data)
    try:
        out = ctx.compress(mat, _mask)
        ip, ix, data = out.download()
        out.free()
    finally:
        mat.free()
    n = ip.shape[0] - 1
    return scisp.csr_matrix((data.astype(_m.dtype, copy=False), ix, ip), shape=(n, n)).tocoo()


class Coo4D(object):
    """Minimal stand-in for the pydata ``sparse.COO`` objects of the tip-based path (absent from this image): the
    three attributes the reference touches -- ``coords`` (4 x nnz), ``data`` (nnz), ``shape`` (N, N, 2, 2)."""

    def __init__(self, coords, data, shape):
        self.coords = np.asarray(coords)
        self.data = np.asarray(data)
        self.shape = tuple(shape)
        assert self.coords.shape == (4, self.data.shape[0]) and len(self.shape) == 4

    @property
    def nnz(self):
        return self.data.shape[0]

    def astype(self, dtype):
        return Coo4D(self.coords, self.data.astype(dtype), self.shape)


def dotdot(_m, _a):
    """
    dot(A.T, dot(M, A)) for a diagonal A given by its trace ``_a`` on a 4-D sparse tensor (sparse_utils.py:480-492):
    every stored element is scaled by ``_a[i] * _a[j]``, in place.
    """
    _a = np.asarray(_a)
    _m.data *= _a[_m.coords[0]] * _a[_m.coords[1]]
    return _m


def kr_biostochastic_4d(m4d, **kwargs):
    """
    Knight-Ruiz for an N x N x 2 x 2 tip tensor (sparse_utils.py:495-508): scale factors from the N x N matrix of
    2 x 2 block sums (balanced on the GPU), then every tensor element scaled by a_i * a_j.

    :param m4d: object with ``coords`` (4 x nnz), ``data``, ``shape`` -- pydata sparse.COO or :class:`Coo4D`
    :return: (scaled tensor as float, scale factors)
    """
    n = m4d.shape[0]
    data = np.asarray(m4d.data, dtype=np.float64)
    m2d = scisp.coo_matrix((data, (np.asarray(m4d.coords[0]), np.asarray(m4d.coords[1]))), shape=(n, n)).tocsr()
    _, scl = kr_biostochastic(m2d, **kwargs)
    return dotdot(m4d.astype(np.float64), scl), scl
