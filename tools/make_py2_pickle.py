#!/usr/bin/env python3
"""
Write tests/golden/contact_map_py2.p.gz: a ContactMap in the byte format the Python-2 reference writes it --
``cPickle.dump(obj, fh)`` (mzd/io_utils.py:12-21), i.e. pickle PROTOCOL 0 of CPython 2.7 -- for the pickle-compatibility row
of SURVEY 8(f)2.  Python 2 is not installed here, so the stream is ASSEMBLED by this script, opcode for opcode as
CPython 2.7's cPickle emits them:

  * old-style class instances (``class ContactMap:``, ``class SeqOrder:``) as ``(i<module>\\n<class>\\n`` INST + dict + BUILD
  * ``str`` as ``S'...'`` with Python-2 repr escapes (numpy buffers included: raw bytes inside an S string), ints as ``I``,
    booleans as ``I01`` / ``I00``, floats as ``F``, None as ``N``, tuples / lists / dicts with ``(...t``, ``(lp``, ``(dp``
  * new-style objects through ``copy_reg._reconstructor`` (the SeqInfo namedtuples, the scipy matrices of scipy < 1.8 with
    their ``row`` / ``col`` / ``data`` / ``_shape`` attribute dictionaries under ``scipy.sparse.coo`` / ``.csr``)
  * numpy arrays and dtypes through ``numpy.core.multiarray._reconstruct`` / ``numpy.dtype`` with their version-1 / -3
    states, ``numpy.random.RandomState`` through ``numpy.random.__RandomState_ctor`` with the ('MT19937', key, pos, has_gauss,
    cached) tuple

It is a fixture generator: the OUTPUT (data) is committed, nothing of the reference's source is.  The object is a small
synthetic map (bin3c_amd.synth) dressed with the attribute list of contact_map.py:488-529.
"""
import gzip
import io
import os
import pickle
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden', 'contact_map_py2.p.gz')


def py2_repr(b):
    """repr() of a Python-2 str holding bytes b (single quotes unless the text holds one and no double quote)"""
    quote = b'"' if (b"'" in b and b'"' not in b) else b"'"
    out = bytearray(quote)
    for ch in b:
        c = bytes([ch])
        if c == quote or c == b'\\':
            out += b'\\' + c
        elif c == b'\t':
            out += b'\\t'
        elif c == b'\n':
            out += b'\\n'
        elif c == b'\r':
            out += b'\\r'
        elif ch < 0x20 or ch >= 0x7f:
            out += b'\\x%02x' % ch
        else:
            out += c
    out += quote
    return bytes(out)


class OldStyle(object):
    """marks an object to be written as an old-style class instance: INST module class + __dict__ + BUILD"""

    def __init__(self, module, name, state):
        self.module, self.name, self.state = module, name, state


class Reduce(object):
    """an object written as callable(*args) [+ BUILD state]"""

    def __init__(self, func, args, state=None):
        self.func, self.args, self.state = func, args, state


class Global(object):
    def __init__(self, module, name):
        self.module, self.name = module, name


class Py2Pickler(object):
    """protocol 0, CPython 2.7 cPickle flavour, over the small type universe above"""

    def __init__(self):
        self.out = io.BytesIO()
        self.memo = {}
        self.keep = []

    def _put(self, obj):
        idx = len(self.memo)
        self.memo[id(obj)] = idx
        self.keep.append(obj)
        self.out.write(b'p%d\n' % idx)

    def _get(self, obj):
        if id(obj) in self.memo:
            self.out.write(b'g%d\n' % self.memo[id(obj)])
            return True
        return False

    def dump(self, obj):
        self.save(obj)
        self.out.write(b'.')
        return self.out.getvalue()

    def save(self, o):
        w = self.out.write
        if o is None:
            w(b'N')
        elif o is True:
            w(b'I01\n')
        elif o is False:
            w(b'I00\n')
        elif isinstance(o, (int, np.integer)):
            w(b'I%d\n' % int(o) if -2 ** 31 <= int(o) < 2 ** 31 else b'L%dL\n' % int(o))
        elif isinstance(o, (float, np.floating)):
            w(b'F' + repr(float(o)).encode() + b'\n')
        elif isinstance(o, (str, bytes)):
            if self._get(o):
                return
            w(b'S' + py2_repr(o.encode('latin1') if isinstance(o, str) else o) + b'\n')
            self._put(o)
        elif isinstance(o, tuple):
            if not o:
                w(b'(t')
                return
            if self._get(o):
                return
            w(b'(')
            for x in o:
                self.save(x)
            w(b't')
            self._put(o)
        elif isinstance(o, list):
            if self._get(o):
                return
            w(b'(l')
            self._put(o)
            for x in o:
                self.save(x)
                w(b'a')
        elif isinstance(o, dict):
            if self._get(o):
                return
            w(b'(d')
            self._put(o)
            for k, v in o.items():
                self.save(k)
                self.save(v)
                w(b's')
        elif isinstance(o, Global):
            if self._get(o):
                return
            w(b'c' + o.module.encode() + b'\n' + o.name.encode() + b'\n')
            self._put(o)
        elif isinstance(o, OldStyle):
            if self._get(o):
                return
            w(b'(i' + o.module.encode() + b'\n' + o.name.encode() + b'\n')
            self._put(o)
            self.save(o.state)
            w(b'b')
        elif isinstance(o, Reduce):
            if self._get(o):
                return
            self.save(o.func)
            self.save(o.args)
            w(b'R')
            self._put(o)
            if o.state is not None:
                self.save(o.state)
                w(b'b')
        elif isinstance(o, np.dtype):
            self.save(self.dtype_reduce(o))
        elif isinstance(o, np.ndarray):
            self.save(self.array_reduce(o))
        else:
            raise TypeError(type(o))

    # numpy 1.x reductions -------------------------------------------------------------------
    _dtypes = {}

    def dtype_reduce(self, dt):
        key = dt.str if dt.names is None else repr(dt.descr)
        if key in self._dtypes:
            return self._dtypes[key]
        if dt.names is None:
            code = dt.str[1:] if dt.str[0] in '<>|=' else dt.str
            order = '|' if dt.itemsize == 1 or dt.kind in 'b' else '<'
            r = Reduce(Global('numpy', 'dtype'), (code, 0, 1), (3, order, None, None, None, -1, -1, 0))
        else:
            fields = {}
            for name in dt.names:
                fdt, off = dt.fields[name][:2]
                fields[name] = (self.dtype_reduce(fdt), off)
            r = Reduce(Global('numpy', 'dtype'), ('V%d' % dt.itemsize, 0, 1),
                       (3, '|', None, tuple(dt.names), fields, dt.itemsize, dt.alignment, 16 if dt.isalignedstruct else 0))
        self._dtypes[key] = r
        return r

    _recon = Global('numpy.core.multiarray', '_reconstruct')
    _ndarray = Global('numpy', 'ndarray')

    def array_reduce(self, a):
        a = np.ascontiguousarray(a)
        return Reduce(self._recon, (self._ndarray, (0,), 'b'), (1, tuple(int(x) for x in a.shape), self.dtype_reduce(a.dtype), False, a.tobytes()))


def old_scipy(m):
    """a scipy.sparse matrix as scipy < 1.8 pickled it under Python 2: object.__reduce_ex__(0) over its attribute dictionary"""
    recon = Global('copy_reg', '_reconstructor')
    obj = Global('__builtin__', 'object')
    if sp.isspmatrix_coo(m):
        state = {'_shape': tuple(int(x) for x in m.shape), 'maxprint': 50, 'row': m.row.astype(np.int32), 'col': m.col.astype(np.int32),
                 'data': m.data, 'has_canonical_format': False}
        return Reduce(recon, (Global('scipy.sparse.coo', 'coo_matrix'), obj, None), state)
    m = m.tocsr()
    state = {'_shape': tuple(int(x) for x in m.shape), 'maxprint': 50, 'indices': m.indices.astype(np.int32),
             'indptr': m.indptr.astype(np.int32), 'data': m.data}
    return Reduce(recon, (Global('scipy.sparse.csr', 'csr_matrix'), obj, None), state)


def main():
    from bin3c_amd.synth import make_synth
    sm = make_synth(60, 500, 77, genome_law='sqrt')
    n = sm.n
    seq_map = sm.to_scipy(np.uint32).tocoo()
    seqinfo_cls = Global('mzd.contact_map', 'SeqInfo')
    recon = Global('copy_reg', '_reconstructor')
    tup = Global('__builtin__', 'tuple')
    seq_info = [Reduce(recon, (seqinfo_cls, tup, (int(sm.lengths[:i].sum()), i, 'contig%03d' % i, int(sm.lengths[i]), int(sm.sites[i]))))
                for i in range(n)]
    struct = np.dtype([('pos', np.int32), ('ori', np.int8), ('mask', np.bool_), ('length', np.int32)])
    order = np.zeros(n, dtype=struct)
    order['pos'] = np.arange(n)
    order['ori'] = 1
    order['mask'] = True
    order['length'] = sm.lengths
    seq_order = OldStyle('mzd.contact_map', 'SeqOrder', {'_positions': np.arange(n, dtype=np.int32), 'order': order})
    rs = np.random.RandomState(1234)
    st = rs.get_state()
    random_state = Reduce(Global('numpy.random', '__RandomState_ctor'), (), ('MT19937', st[1], int(st[2]), int(st[3]), float(st[4])))
    mask = (sm.lengths >= 1000)
    attrs = {
        'strong': None, 'bam_file': 'hic2ctg.bam', 'bin_size': None, 'min_mapq': 60, 'min_insert': 0, 'min_len': 1000,
        'min_sig': 5, 'min_extent': 50000, 'min_size': 5, 'max_fold': None, 'random_state': random_state,
        'seq_info': seq_info, 'seq_map': old_scipy(seq_map), 'seq_file': 'contigs.fasta', 'grouping': None, 'extent_map': None,
        'order': seq_order, 'tip_size': None, 'precount': False, 'total_reads': 123456, 'cov_info': None,
        'processed_map': None, 'primary_acceptance_mask': mask, 'bisto_scale': None, 'seq_analyzer': None,
        'enzymes': ['MluCI', 'Sau3AI'], 'total_len': int(sm.lengths.sum()), 'total_seq': n,
    }
    raw = Py2Pickler().dump(OldStyle('mzd.contact_map', 'ContactMap', attrs))
    # sanity: Python 3's own unpickler must accept the stream once the reference's names resolve (tests do the real check)
    with gzip.GzipFile(OUT, 'wb', mtime=0) as fh:
        fh.write(raw)
    np.savez_compressed(OUT.replace('.p.gz', '_expect.npz'), indptr=sm.indptr, indices=sm.indices, counts=sm.counts,
                        lengths=sm.lengths, sites=sm.sites, mask=mask, rs_next=np.random.RandomState(1234).randint(0, 1 << 30, 5))
    print('wrote {} ({} bytes raw, protocol 0, starts {!r})'.format(OUT, len(raw), raw[:40]))


if __name__ == '__main__':
    main()
