"""
ORACLE -- test infrastructure, NOT product code.

ctypes front end of the scalar C restatement of the reference hot path (kr.c, louvain_seq.c,
louvain_bsp.c; each function cites the reference lines it follows).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this package, and only
as the checker / the timed CPU baseline.  ``bin3c_amd`` never imports it.

KR and the CSR helpers are pinned against ``tests/golden/*.npz`` (outputs of the reference's own
function text, see tools/make_golden.py).  Louvain: PARITY UNPINNED (python-louvain 0.11 is a
third-party dependency absent from the reference tree and from this image; see louvain_seq.c).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

OK, E_NAN, E_NOCONV, E_ARG = 0, 1, 2, 3


class KRStats(C.Structure):
    _fields_ = [('n_iter', C.c_int64), ('n_outer', C.c_int64), ('n_zero_diag', C.c_int64),
                ('is_symmetric', C.c_int32), ('hit_max_iter', C.c_int32), ('hit_delta', C.c_int64),
                ('hit_Delta', C.c_int64), ('rout', C.c_double)]


def build(force=False):
    """Compile oracle/_build/liboracle.so with gcc (a few seconds)."""
    out = os.path.join(_HERE, '_build', 'liboracle.so')
    srcs = [os.path.join(_HERE, f) for f in ('kr.c', 'louvain_seq.c', 'louvain_bsp.c', 'oracle.h', 'Makefile')]
    if force or not os.path.exists(out) or any(os.path.getmtime(s) > os.path.getmtime(out) for s in srcs):
        subprocess.check_call(['make', '-s', '-C', _HERE])
    return out


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, '_build', 'liboracle.so')
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_max_value.restype = C.c_double
        _LIB.orc_modularity.restype = C.c_double
        _LIB.orc_compress.restype = C.c_int64
        _LIB.orc_quant_shift.restype = C.c_int32
    return _LIB


def set_threads(n):
    """Threads of the row-parallel loops (SpMV, site normalisation, final scaling); 1 = the sequential oracle."""
    lib().orc_set_threads(int(n))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _csr(indptr, indices):
    return (np.ascontiguousarray(indptr, dtype=np.int64), np.ascontiguousarray(indices, dtype=np.int32))


def kr_balance(indptr, indices, data, tol=1e-6, delta=0.1, Delta=3.0, max_iter=1000):
    """-> (rc, x, KRStats)"""
    ip, ix = _csr(indptr, indices)
    a = np.ascontiguousarray(data, dtype=np.float64)
    n = ip.shape[0] - 1
    x = np.empty(n, dtype=np.float64)
    st = KRStats()
    rc = lib().orc_kr_balance(C.c_int64(n), _p(ip), _p(ix), _p(a), C.c_double(tol), C.c_double(delta),
                              C.c_double(Delta), C.c_int64(max_iter), _p(x), C.byref(st))
    return rc, x, st


def scale_sym(indptr, indices, data, x):
    ip, ix = _csr(indptr, indices)
    a = np.ascontiguousarray(data, dtype=np.float64)
    out = np.empty_like(a)
    lib().orc_scale_sym(C.c_int64(ip.shape[0] - 1), _p(ip), _p(ix), _p(a), _p(np.ascontiguousarray(x)), _p(out))
    return out


def site_norm(indptr, indices, data, sites):
    ip, ix = _csr(indptr, indices)
    a = np.array(data, dtype=np.float64)
    s = np.ascontiguousarray(sites, dtype=np.float64)
    lib().orc_site_norm(C.c_int64(ip.shape[0] - 1), _p(ip), _p(ix), _p(s), _p(a))
    return a


def max_offdiag_u32(indptr, indices, data):
    ip, ix = _csr(indptr, indices)
    a = np.ascontiguousarray(data, dtype=np.uint32)
    out = np.empty(ip.shape[0] - 1, dtype=np.uint32)
    lib().orc_max_offdiag_u32(C.c_int64(ip.shape[0] - 1), _p(ip), _p(ix), _p(a), _p(out))
    return out


def compress(indptr, indices, data, mask):
    ip, ix = _csr(indptr, indices)
    a = np.ascontiguousarray(data, dtype=np.float64)
    msk = np.ascontiguousarray(mask, dtype=np.uint8)
    n = ip.shape[0] - 1
    oip = np.zeros(n + 1, dtype=np.int64)
    m = lib().orc_compress(C.c_int64(n), _p(ip), _p(ix), _p(a), _p(msk), _p(oip), None, None)
    nnz = int(oip[m])
    oix = np.empty(nnz, dtype=np.int32)
    oa = np.empty(nnz, dtype=np.float64)
    lib().orc_compress(C.c_int64(n), _p(ip), _p(ix), _p(a), _p(msk), _p(oip), _p(oix), _p(oa))
    return int(m), oip[:m + 1].copy(), oix, oa


def max_value(indptr, data):
    ip = np.ascontiguousarray(indptr, dtype=np.int64)
    a = np.ascontiguousarray(data, dtype=np.float64)
    return lib().orc_max_value(C.c_int64(ip.shape[0] - 1), _p(ip), _p(a))


def graph_weights(indptr, indices, data, scl):
    ip, ix = _csr(indptr, indices)
    a = np.ascontiguousarray(data, dtype=np.float64)
    w = np.empty_like(a)
    lib().orc_graph_weights(C.c_int64(ip.shape[0] - 1), _p(ip), _p(ix), _p(a), C.c_double(scl), _p(w))
    return w


def louvain_seq(indptr, indices, w):
    """-> dict(labels, n_comm, Q, n_levels, n_sweeps)"""
    ip, ix = _csr(indptr, indices)
    ww = np.ascontiguousarray(w, dtype=np.float64)
    n = ip.shape[0] - 1
    labels = np.empty(n, dtype=np.int32)
    nc, nl, ns, q = C.c_int32(), C.c_int32(), C.c_int64(), C.c_double()
    lib().orc_louvain_seq(C.c_int64(n), _p(ip), _p(ix), _p(ww), _p(labels), C.byref(nc), C.byref(q), C.byref(nl),
                          C.byref(ns))
    return dict(labels=labels, n_comm=nc.value, Q=q.value, n_levels=nl.value, n_sweeps=ns.value)


def quant_shift(indptr, indices, w):
    ip, ix = _csr(indptr, indices)
    ww = np.ascontiguousarray(w, dtype=np.float64)
    return int(lib().orc_quant_shift(C.c_int64(ip.shape[0] - 1), _p(ip), _p(ix), _p(ww)))


def quantise(w, shift):
    ww = np.ascontiguousarray(w, dtype=np.float64)
    q = np.empty(ww.shape[0], dtype=np.int64)
    lib().orc_quantise(C.c_int64(ww.shape[0]), _p(ww), C.c_int32(shift), _p(q))
    return q


def louvain_bsp(indptr, indices, q, n_sub=4, max_sweeps=0):
    ip, ix = _csr(indptr, indices)
    qq = np.ascontiguousarray(q, dtype=np.int64)
    n = ip.shape[0] - 1
    labels = np.empty(n, dtype=np.int32)
    nc, nl, ns, qo = C.c_int32(), C.c_int32(), C.c_int64(), C.c_double()
    lib().orc_louvain_bsp(C.c_int64(n), _p(ip), _p(ix), _p(qq), C.c_int32(n_sub), C.c_int32(max_sweeps), _p(labels),
                          C.byref(nc), C.byref(qo), C.byref(nl), C.byref(ns))
    return dict(labels=labels, n_comm=nc.value, Q=qo.value, n_levels=nl.value, n_sweeps=ns.value)


def modularity(indptr, indices, w, labels):
    ip, ix = _csr(indptr, indices)
    ww = np.ascontiguousarray(w, dtype=np.float64)
    lab = np.ascontiguousarray(labels, dtype=np.int32)
    return lib().orc_modularity(C.c_int64(ip.shape[0] - 1), _p(ip), _p(ix), _p(ww), _p(lab))


def downsample_dense(m, block_size, method='mean'):
    """Dense numpy restatement of mzd/sparse_utils.py:48-87 (`downsample`): zero-pad to a multiple of block_size,
    view as (R/b, b, C/b, b) and reduce axes (1, 3); mean = block sum times 1/b^2 (:81-82), max = block maximum
    with the zero cells taking part (:83-84).  PARITY UNPINNED: the reference body needs pydata `sparse`
    (sparse_utils.py:4, :77), which is not installed here, so its text cannot be executed for golden vectors.
    Small inputs only (densifies)."""
    import numpy as np
    m = np.asarray(m.todense()) if hasattr(m, 'todense') else np.asarray(m)
    b = int(block_size)
    pr, pc = (-m.shape[0]) % b, (-m.shape[1]) % b
    p = np.pad(m, ((0, pr), (0, pc)))
    v = p.reshape(p.shape[0] // b, b, p.shape[1] // b, b)
    if method == 'mean':
        return v.sum(axis=(1, 3)) * (1.0 / b ** 2)
    return v.max(axis=(1, 3))
