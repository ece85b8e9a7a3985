"""
ctypes binding of libb3c.so (include/b3c.h).  This is the ONLY route from the Python host code to the
device: there is no CPU fallback.  Import fails loudly when the library has not been built, and every
compute call fails loudly when no MI355X is visible.
"""
import ctypes as C
import weakref
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libb3c.so')

OK, E_ARG, E_HIP, E_NAN, E_NOCONV, E_NOMEM = 0, -1, -2, -3, -4, -5
F64, U32 = 0, 1


class B3CError(RuntimeError):
    def __init__(self, code, message):
        RuntimeError.__init__(self, message)
        self.code = code


class KRInfo(C.Structure):
    _fields_ = [('n_iter', C.c_int64), ('n_outer', C.c_int64), ('n_zero_diag', C.c_int64),
                ('is_symmetric', C.c_int32), ('hit_max_iter', C.c_int32), ('hit_delta', C.c_int64),
                ('hit_Delta', C.c_int64), ('rout', C.c_double), ('spmv_ms', C.c_double), ('n_spmv', C.c_int64),
                ('bitwise_symmetric', C.c_int32), ('value_bits', C.c_int32),
                ('tcd_tiles', C.c_int64), ('tcd_dict_max', C.c_int64), ('tcd_dict_total', C.c_int64),
                ('tcd_tail_entries', C.c_int64), ('tcd_xt_tails', C.c_int64), ('big_counts', C.c_int64),
                ('tcd_word_bytes', C.c_int64), ('xt_word_bytes', C.c_int64), ('xt_residual_words', C.c_int64),
                ('spmv_stream_bytes', C.c_int64)]


class LouvainInfo(C.Structure):
    _fields_ = [('n_comm', C.c_int32), ('n_levels', C.c_int32), ('n_sweeps', C.c_int64), ('n_moves', C.c_int64),
                ('quant_shift', C.c_int32), ('levels_sharded', C.c_int32), ('modularity', C.c_double),
                ('move_ms', C.c_double), ('n_move_launch', C.c_int64), ('move_nnz', C.c_int64),
                ('move_rows', C.c_int64), ('n_comm_level0', C.c_int64), ('agg_gathered_runs', C.c_int64),
                ('first_sweep_ms', C.c_double), ('first_sweep_launches', C.c_int64), ('agg_nnz', C.c_int64)]


class B3CDist(C.Structure):
    _fields_ = [('rank', C.c_int32), ('world', C.c_int32), ('chunk', C.c_int64), ('vec', C.c_void_p),
                ('scal', C.c_void_p), ('labels', C.c_void_p), ('isum', C.c_void_p), ('allgather', C.c_void_p),
                ('user', C.c_void_p), ('comm', C.c_void_p), ('allreduce', C.c_void_p), ('starts', C.c_void_p),
                ('allgatherv', C.c_void_p)]


# structs that cross the ABI, by the `which` of b3c_abi_layout
ABI_STRUCTS = {0: KRInfo, 1: LouvainInfo, 2: B3CDist}


def struct_layout(cls):
    """[sizeof, n_fields, offset of every field in declaration order] -- the form b3c_abi_layout reports"""
    return [C.sizeof(cls), len(cls._fields_)] + [getattr(cls, f[0]).offset for f in cls._fields_]


def check_abi(lib):
    """The library's compiled struct layouts against this module's ctypes definitions; a mismatch would let the library
    write past a ctypes object, so it is an ImportError, not a warning."""
    for which, cls in ABI_STRUCTS.items():
        buf = (C.c_int32 * 64)()
        k = lib.b3c_abi_layout(which, buf, 64)
        got = list(buf[:max(k, 0)])
        if got != struct_layout(cls):
            raise ImportError('libb3c.so and bin3c_amd/_native.py disagree on the layout of {}: library {} vs binding {} '
                              '(rebuild the library, or update the binding from include/b3c.h)'.format(cls.__name__, got, struct_layout(cls)))


# every symbol include/b3c.h declares: (name, restype, argtypes)
_VP = C.c_void_p
SYMBOLS = [
    ('b3c_ctx_create', C.c_int, [C.c_int, _VP, C.POINTER(_VP)]),
    ('b3c_ctx_destroy', C.c_int, [_VP]),
    ('b3c_ctx_reload_env', C.c_int, [_VP]),
    ('b3c_last_error', C.c_char_p, [_VP]),
    ('b3c_ctx_sync', C.c_int, [_VP]),
    ('b3c_ctx_bytes', C.c_int64, [_VP]),
    ('b3c_ctx_trim', C.c_int, [_VP]),
    ('b3c_ctx_set_dist', C.c_int, [_VP, _VP]),
    ('b3c_dist_unique_id', C.c_int, [_VP]),
    ('b3c_dist_comm_create', C.c_int, [_VP, C.c_int, C.c_int, _VP, C.POINTER(_VP)]),
    ('b3c_dist_comm_destroy', C.c_int, [_VP, _VP]),
    ('b3c_ctx_exchanges', C.c_int64, [_VP]),
    ('b3c_ctx_exchanges_direct', C.c_int64, [_VP]),
    ('b3c_dist_window_create', C.c_int, [_VP, C.c_int64, _VP]),
    ('b3c_dist_window_open', C.c_int, [_VP, _VP]),
    ('b3c_dist_window_close', C.c_int, [_VP]),
    ('b3c_dist_max_f64', C.c_int, [_VP, C.POINTER(C.c_double)]),
    ('b3c_dist_allgather_i32', C.c_int, [_VP, _VP, C.c_int64]),
    ('b3c_mat_permute_rows', C.c_int, [_VP, _VP, _VP, C.c_int, C.c_int64, C.c_int64, C.POINTER(_VP)]),
    ('b3c_mat_upload', C.c_int, [_VP, C.c_int64, C.c_int64, _VP, _VP, _VP, C.c_int, C.POINTER(_VP)]),
    ('b3c_mat_free', C.c_int, [_VP, _VP]),
    ('b3c_mat_info', C.c_int, [_VP, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    ('b3c_mat_row_nnz', C.c_int, [_VP, _VP, _VP, _VP]),
    ('b3c_mat_download', C.c_int, [_VP, _VP, _VP, _VP, _VP]),
    ('b3c_mat_device_ptrs', C.c_int, [_VP, C.POINTER(_VP), C.POINTER(_VP), C.POINTER(_VP)]),
    ('b3c_max_offdiag_u32', C.c_int, [_VP, _VP, C.c_int, _VP]),
    ('b3c_acceptance_mask', C.c_int, [_VP, C.c_int64, _VP, _VP, C.c_int64, C.c_int64, _VP]),
    ('b3c_vec_permute', C.c_int, [_VP, C.c_int64, C.c_int, _VP, _VP, _VP, C.c_int]),
    ('b3c_site_norm', C.c_int, [_VP, _VP, _VP, C.POINTER(_VP)]),
    ('b3c_scale_sym', C.c_int, [_VP, _VP, _VP]),
    ('b3c_scale_sym_graph', C.c_int, [_VP, _VP, _VP]),
    ('b3c_compress', C.c_int, [_VP, _VP, _VP, C.POINTER(_VP)]),
    ('b3c_max_value', C.c_int, [_VP, _VP, C.POINTER(C.c_double)]),
    ('b3c_downsample', C.c_int, [_VP, _VP, C.c_int64, C.c_int, C.POINTER(_VP)]),
    ('b3c_kr_balance', C.c_int, [_VP, _VP, C.c_double, C.c_double, C.c_double, C.c_int64, _VP, C.POINTER(KRInfo)]),
    ('b3c_kr_balance_counts', C.c_int, [_VP, _VP, _VP, C.c_double, C.c_double, C.c_double, C.c_int64, _VP, C.POINTER(KRInfo)]),
    ('b3c_normalise_scale', C.c_int, [_VP, _VP, _VP, _VP, C.c_int, C.POINTER(_VP)]),
    ('b3c_normalise_scale_compress', C.c_int, [_VP, _VP, _VP, _VP, C.c_int, _VP, C.POINTER(_VP)]),
    ('b3c_louvain', C.c_int, [_VP, _VP, C.c_double, C.c_int, C.c_int, _VP, C.POINTER(LouvainInfo)]),
    ('b3c_graph_quantise', C.c_int, [_VP, _VP, C.c_double, _VP, C.POINTER(C.c_int32)]),
    ('b3c_coo_symmetrise', C.c_int, [_VP, C.c_int64, C.c_int64, _VP, _VP, _VP, _VP, C.POINTER(_VP), C.POINTER(C.c_int64)]),
    ('b3c_locality_order', C.c_int, [_VP, _VP, _VP, _VP]),
    ('b3c_mat_permute', C.c_int, [_VP, _VP, _VP, C.c_int, C.POINTER(_VP)]),
    ('b3c_mat_ids', C.c_int, [_VP, _VP, _VP]),
    ('b3c_version', C.c_char_p, []),
    ('b3c_abi_layout', C.c_int, [C.c_int, C.POINTER(C.c_int32), C.c_int]),
]

_lib = None


def load():
    """dlopen libb3c.so and type every entry point.  Raises ImportError if it was never built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError('{} is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                              '(hipcc --offload-arch=gfx950); there is no CPU fallback'.format(LIB_PATH))
        # torch bundles its own HIP/HSA runtime; a process must not end up with two of them.  Loading
        # torch first makes its copy the one libb3c binds to as well (same sonames), whatever the
        # import order of the caller.  Without torch, libb3c uses /opt/rocm's runtime.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        check_abi(lib)
        if os.environ.get('B3C_TRACE'):
            lib = _Traced(lib)
        _lib = lib
    return _lib


class _Traced(object):
    """B3C_TRACE=1: print the wall time of every C-ABI call (development aid)."""

    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        import sys
        import time
        fn = getattr(self._lib, name)

        def call(*a):
            t0 = time.perf_counter()
            r = fn(*a)
            dt = (time.perf_counter() - t0) * 1e3
            if dt > 0.05:
                sys.stderr.write('[b3c] %-22s %9.3f ms\n' % (name, dt))
            return r
        return call


def _ptr(a):
    """host numpy array, raw integer device pointer, or None -> c_void_p"""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    if isinstance(a, int):
        return C.c_void_p(a)
    if hasattr(a, 'data_ptr'):  # torch tensor (device or host)
        return C.c_void_p(a.data_ptr())
    raise TypeError('cannot pass {} to libb3c'.format(type(a)))


class Mat(object):
    """Device-resident CSR matrix handle (b3c_mat*)."""

    def __init__(self, ctx, handle):
        self.ctx = ctx
        self.h = handle
        n, nnz, kind = C.c_int64(), C.c_int64(), C.c_int()
        ctx.lib.b3c_mat_info(handle, C.byref(n), C.byref(nnz), C.byref(kind))
        self.n, self.nnz, self.kind = n.value, nnz.value, kind.value

    def download(self, structure=True):
        """-> (indptr int64, indices int32, data) as numpy arrays"""
        ip = np.empty(self.n + 1, dtype=np.int64) if structure else None
        ix = np.empty(self.nnz, dtype=np.int32) if structure else None
        data = np.empty(self.nnz, dtype=np.uint32 if self.kind == U32 else np.float64)
        self.ctx._check(self.ctx.lib.b3c_mat_download(self.ctx.h, self.h, _ptr(ip), _ptr(ix), _ptr(data)))
        return ip, ix, data

    def device_ptrs(self):
        a, b, c = _VP(), _VP(), _VP()
        self.ctx.lib.b3c_mat_device_ptrs(self.h, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def free(self):
        if self.h is not None:
            self.ctx.lib.b3c_mat_free(self.ctx.h, self.h)
            self.h = None

    def __del__(self):
        try:
            if self.ctx.h is not None:
                self.free()
        except Exception:
            pass


_LIVE = weakref.WeakSet()      # open contexts (reload_env_all: the test suite flips switches between calls)


def reload_env_all():
    """Have every open context read the B3C_* switches again (b3c_ctx_reload_env)."""
    for c in list(_LIVE):
        if c.h is not None:
            c.reload_env()


class Context(object):
    """b3c_ctx*: one device, one stream, one caching HBM pool."""

    def __init__(self, device=0, stream=None):
        self.lib = load()
        h = _VP()
        rc = self.lib.b3c_ctx_create(int(device), _VP(stream) if stream else None, C.byref(h))
        if rc != OK:
            raise B3CError(rc, 'b3c_ctx_create(device={}) failed with code {}: no usable MI355X/HIP device; '
                               'libb3c has no CPU fallback'.format(device, rc))
        self.h = h
        self.device = device
        _LIVE.add(self)

    def close(self):
        if self.h is not None:
            self.lib.b3c_ctx_destroy(self.h)
            self.h = None
        _LIVE.discard(self)

    def reload_env(self):
        """libb3c reads its B3C_* switches once, in b3c_ctx_create; call this after changing them in os.environ."""
        self._check(self.lib.b3c_ctx_reload_env(self.h))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != OK:
            msg = self.lib.b3c_last_error(self.h)
            raise B3CError(rc, msg.decode() if msg else 'libb3c error {}'.format(rc))

    def sync(self):
        self._check(self.lib.b3c_ctx_sync(self.h))

    def bytes_in_use(self):
        return int(self.lib.b3c_ctx_bytes(self.h))

    def trim(self):
        """give the pooled (unused) HBM blocks of this context back to the driver"""
        self._check(self.lib.b3c_ctx_trim(self.h))

    # ---- matrices ----
    def upload(self, indptr, indices, data, n=None, nnz=None):
        """CSR -> HBM.  numpy arrays are converted to the ABI dtypes; device pointers need n/nnz."""
        if isinstance(indptr, np.ndarray):
            indptr = np.ascontiguousarray(indptr, dtype=np.int64)
            indices = np.ascontiguousarray(indices, dtype=np.int32)
            if data.dtype == np.uint32:
                kind = U32
                data = np.ascontiguousarray(data)
            else:
                kind = F64
                data = np.ascontiguousarray(data, dtype=np.float64)
            n = indptr.shape[0] - 1
            nnz = int(indptr[-1])
            assert indices.shape[0] >= nnz and data.shape[0] >= nnz
        else:
            kind = U32 if getattr(data, 'element_size', lambda: 8)() == 4 else F64
        h = _VP()
        self._check(self.lib.b3c_mat_upload(self.h, n, nnz, _ptr(indptr), _ptr(indices), _ptr(data), kind, C.byref(h)))
        return Mat(self, h)

    def clone(self, mat):
        """a second device copy of a matrix (device-to-device; a plain CSR copy: ordering ids and cached maxima are not carried)"""
        a, b, c = mat.device_ptrs()
        h = _VP()
        self._check(self.lib.b3c_mat_upload(self.h, mat.n, mat.nnz, C.c_void_p(a), C.c_void_p(b), C.c_void_p(c), mat.kind, C.byref(h)))
        return Mat(self, h)

    # ---- ingest tail ----
    def coo_symmetrise(self, n, lo, hi, counts, diag=None):
        """pair stream -> (symmetric uint32 CSR Mat, undirected off-diagonal edge count)"""
        if isinstance(lo, np.ndarray):
            lo = np.ascontiguousarray(lo, dtype=np.int64)
            hi = np.ascontiguousarray(hi, dtype=np.int64)
            counts = np.ascontiguousarray(counts, dtype=np.uint32)
            m = lo.shape[0]
        else:
            m = lo.numel()
        if isinstance(diag, np.ndarray):
            diag = np.ascontiguousarray(diag, dtype=np.uint32)
        h = _VP()
        ne = C.c_int64()
        self._check(self.lib.b3c_coo_symmetrise(self.h, int(n), int(m), _ptr(lo), _ptr(hi), _ptr(counts), _ptr(diag), C.byref(h),
                                                C.byref(ne)))
        return Mat(self, h), ne.value

    # ---- locality ordering ----
    def locality_order(self, mat):
        """-> (new2old, old2new) int32[n]"""
        n2o = np.empty(mat.n, dtype=np.int32)
        o2n = np.empty(mat.n, dtype=np.int32)
        self._check(self.lib.b3c_locality_order(self.h, mat.h, _ptr(n2o), _ptr(o2n)))
        return n2o, o2n

    def permute(self, mat, new2old, unsorted=False, rows=None):
        """P A P^T; unsorted=True skips re-sorting the columns inside each row (fast, not canonical CSR); rows=(r0, r1)
        forms only that ROW BLOCK of the result (the other rows stay empty): the part a rank of the sharded path owns."""
        if isinstance(new2old, np.ndarray):
            new2old = np.ascontiguousarray(new2old, dtype=np.int32)
        h = _VP()
        flags = 1 if unsorted else 0
        if rows is None:
            self._check(self.lib.b3c_mat_permute(self.h, mat.h, _ptr(new2old), flags, C.byref(h)))
        else:
            self._check(self.lib.b3c_mat_permute_rows(self.h, mat.h, _ptr(new2old), flags, int(rows[0]), int(rows[1]), C.byref(h)))
        return Mat(self, h)

    def dist_max(self, value):
        """maximum of a host float over the ranks of the installed sharding (identity without one)"""
        out = C.c_double(float(value))
        self._check(self.lib.b3c_dist_max_f64(self.h, C.byref(out)))
        return out.value

    def dist_allgather_i32(self, vec):
        """all-gather of a per-contig int32 vector (every rank computed its row range) in place"""
        n = vec.numel() if hasattr(vec, 'numel') else vec.shape[0]
        self._check(self.lib.b3c_dist_allgather_i32(self.h, _ptr(vec), int(n)))
        return vec

    def mat_ids(self, mat):
        """caller-side row ids of a permuted matrix, or None for identity"""
        out = np.empty(mat.n, dtype=np.int32)
        rc = self.lib.b3c_mat_ids(self.h, mat.h, _ptr(out))
        if rc == 1:
            return None
        self._check(rc)
        return out

    # ---- helpers ----
    def max_offdiag_u32(self, mat, assume_symmetric=False, out=None):
        res = np.empty(mat.n, dtype=np.uint32) if out is None else out
        self._check(self.lib.b3c_max_offdiag_u32(self.h, mat.h, int(bool(assume_symmetric)), _ptr(res)))
        return res

    def acceptance_mask(self, lengths, signal, min_len, min_sig, out=None):
        """set_primary_acceptance_mask (contact_map.py:888-905): (lengths >= min_len) & (signal >= min_sig) -> uint8[n]"""
        n = lengths.numel() if hasattr(lengths, 'numel') else len(lengths)
        if isinstance(lengths, np.ndarray):
            lengths = np.ascontiguousarray(lengths, dtype=np.int64)
        if isinstance(signal, np.ndarray):
            signal = np.ascontiguousarray(signal, dtype=np.uint32)
        mask = np.empty(n, dtype=np.uint8) if out is None else out
        self._check(self.lib.b3c_acceptance_mask(self.h, int(n), _ptr(lengths), _ptr(signal), int(min_len), int(min_sig), _ptr(mask)))
        return mask

    def vec_permute(self, src, idx, out, scatter=False):
        """out[i] = src[idx[i]] (or out[idx[i]] = src[i]); numpy arrays or device tensors of 1-, 4- or 8-byte elements"""
        n = idx.numel() if hasattr(idx, 'numel') else len(idx)
        esz = src.element_size() if hasattr(src, 'element_size') else src.dtype.itemsize
        self._check(self.lib.b3c_vec_permute(self.h, int(n), int(esz), _ptr(src), _ptr(idx), _ptr(out), 1 if scatter else 0))
        return out

    def row_nnz(self, mat, idx=None, out=None):
        """stored entries of the rows of `mat`, in the order idx (int32 tensor / array) or row order -> out (int64)"""
        if out is None:
            out = np.empty(mat.n, dtype=np.int64)
        self._check(self.lib.b3c_mat_row_nnz(self.h, mat.h, _ptr(idx) if idx is not None else None, _ptr(out)))
        return out

    def site_norm(self, counts, sites):
        if isinstance(sites, np.ndarray):
            sites = np.ascontiguousarray(sites, dtype=np.float64)
        h = _VP()
        self._check(self.lib.b3c_site_norm(self.h, counts.h, _ptr(sites), C.byref(h)))
        return Mat(self, h)

    def scale_sym(self, mat, x):
        if isinstance(x, np.ndarray):
            x = np.ascontiguousarray(x, dtype=np.float64)
        self._check(self.lib.b3c_scale_sym(self.h, mat.h, _ptr(x)))

    def scale_sym_graph(self, mat, x):
        if isinstance(x, np.ndarray):
            x = np.ascontiguousarray(x, dtype=np.float64)
        self._check(self.lib.b3c_scale_sym_graph(self.h, mat.h, _ptr(x)))

    def compress(self, mat, mask):
        if isinstance(mask, np.ndarray):
            mask = np.ascontiguousarray(mask, dtype=np.uint8)
        h = _VP()
        self._check(self.lib.b3c_compress(self.h, mat.h, _ptr(mask), C.byref(h)))
        return Mat(self, h)

    def downsample(self, mat, block_size, method='mean', skip_diag=False):
        h = _VP()
        self._check(self.lib.b3c_downsample(self.h, mat.h, int(block_size), {'mean': 0, 'max': 1}[method] | (2 if skip_diag else 0),
                                            C.byref(h)))
        return Mat(self, h)

    def max_value(self, mat):
        out = C.c_double()
        self._check(self.lib.b3c_max_value(self.h, mat.h, C.byref(out)))
        return out.value

    # ---- KR ----
    def kr_balance(self, mat, tol=1e-6, delta=0.1, Delta=3.0, max_iter=1000, x_out=None, check=True):
        """-> (rc, x, KRInfo); raises B3CError for rc != OK when check"""
        x = np.empty(mat.n, dtype=np.float64) if x_out is None else x_out
        info = KRInfo()
        rc = self.lib.b3c_kr_balance(self.h, mat.h, tol, delta, Delta, int(max_iter), _ptr(x), C.byref(info))
        if check:
            self._check(rc)
        return rc, x, info

    def kr_balance_counts(self, counts, sites, tol=1e-6, delta=0.1, Delta=3.0, max_iter=1000, x_out=None, check=True):
        """KR on the raw count matrix with 1/sites folded in (8 B per entry streamed) -> (rc, x, KRInfo)"""
        if isinstance(sites, np.ndarray):
            sites = np.ascontiguousarray(sites, dtype=np.float64)
        x = np.empty(counts.n, dtype=np.float64) if x_out is None else x_out
        info = KRInfo()
        rc = self.lib.b3c_kr_balance_counts(self.h, counts.h, _ptr(sites), tol, delta, Delta, int(max_iter), _ptr(x),
                                            C.byref(info))
        if check:
            self._check(rc)
        return rc, x, info

    def normalise_scale(self, counts, sites, x, graph_form=False):
        if isinstance(sites, np.ndarray):
            sites = np.ascontiguousarray(sites, dtype=np.float64)
        if isinstance(x, np.ndarray):
            x = np.ascontiguousarray(x, dtype=np.float64)
        h = _VP()
        self._check(self.lib.b3c_normalise_scale(self.h, counts.h, _ptr(sites), _ptr(x), 1 if graph_form else 0, C.byref(h)))
        return Mat(self, h)

    def normalise_scale_compress(self, counts, sites, x, mask, graph_form=False):
        """masked sub-map of the balanced map straight from the counts (the full balanced map is not written)"""
        if isinstance(sites, np.ndarray):
            sites = np.ascontiguousarray(sites, dtype=np.float64)
        if isinstance(x, np.ndarray):
            x = np.ascontiguousarray(x, dtype=np.float64)
        if isinstance(mask, np.ndarray):
            mask = np.ascontiguousarray(mask, dtype=np.uint8)
        h = _VP()
        self._check(self.lib.b3c_normalise_scale_compress(self.h, counts.h, _ptr(sites), _ptr(x), 1 if graph_form else 0,
                                                          _ptr(mask), C.byref(h)))
        return Mat(self, h)

    # ---- Louvain ----
    def louvain(self, mat, scl=1.0, n_sub=0, max_sweeps=0, labels_out=None):
        labels = np.empty(mat.n, dtype=np.int32) if labels_out is None else labels_out
        info = LouvainInfo()
        self._check(self.lib.b3c_louvain(self.h, mat.h, float(scl), int(n_sub), int(max_sweeps), _ptr(labels),
                                         C.byref(info)))
        return labels, info

    def graph_quantise(self, mat, scl=1.0):
        q = np.empty(mat.nnz, dtype=np.int64)
        shift = C.c_int32()
        self._check(self.lib.b3c_graph_quantise(self.h, mat.h, float(scl), _ptr(q), C.byref(shift)))
        return q, shift.value


_default_ctx = {}


def default_context(device=0):
    """Process-wide context per device (created on first use)."""
    ctx = _default_ctx.get(device)
    if ctx is None or ctx.h is None:
        ctx = Context(device)
        _default_ctx[device] = ctx
    return ctx
