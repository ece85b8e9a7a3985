"""
STAGED pins: three rows of SURVEY 8 are "partial" only because software is absent from this image -- python-louvain (a11:
tests/test_oracle_louvain.py::test_against_python_louvain), pydata `sparse` ((f)4: the body of sparse_utils.downsample) and a
Python 2 interpreter ((f)2: what cPickle really writes).  Each test below runs the moment its package / interpreter exists and is
skipped otherwise, so a box that has them turns the row green without a code change.  CPU only; nothing here touches libb3c.
"""
import importlib
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse as scisp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'


def _reference_text(relpath, first, last):
    """lines first..last (1-based, inclusive) of a reference source file -- read where it lies, never copied"""
    path = os.path.join(REF, relpath)
    if not os.path.exists(path):
        pytest.skip('the reference tree is not on this box')
    with open(path) as fh:
        return ''.join(fh.readlines()[first - 1:last])


@pytest.mark.parametrize('shape,block,method', [((64, 64), 4, 'mean'), ((64, 64), 4, 'max'), ((61, 61), 7, 'mean'), ((61, 61), 7, 'max'),
                                                ((200, 200), 25, 'mean'), ((200, 200), 25, 'max'), ((33, 33), 50, 'mean')])
def test_downsample_oracle_against_the_reference_text(oracle_lib, shape, block, method):
    """(f)4: `oracle.downsample_dense` (what the device's b3c_downsample is held to) against the reference's own `downsample`
    (sparse_utils.py:48-87), executed from its text.  Needs pydata `sparse` (the function pads and reshapes a sparse.COO)."""
    sparse = pytest.importorskip('sparse')
    import ast
    import math
    text = _reference_text('mzd/sparse_utils.py', 48, 87)

    class Py2Division(ast.NodeTransformer):
        """the reference is Python 2: `/` between two ints floors (sparse_utils.py:81, :84 reshape by shape / block_size)"""

        def visit_BinOp(self, node):
            self.generic_visit(node)
            if isinstance(node.op, ast.Div):
                return ast.copy_location(ast.Call(func=ast.Name(id='_py2_div', ctx=ast.Load()), args=[node.left, node.right], keywords=[]), node)
            return node

    def _py2_div(a, b):
        ints = (int, np.integer)
        return a // b if isinstance(a, ints) and isinstance(b, ints) else a / b

    tree = ast.fix_missing_locations(Py2Division().visit(ast.parse(text)))
    ns = {'np': np, 'sparse': sparse, 'scisp': scisp, 'ceil': math.ceil, 'xrange': range, '_py2_div': _py2_div}
    exec(compile(tree, 'sparse_utils.py:48-87', 'exec'), ns)
    rs = np.random.RandomState(shape[0] * 131 + block)
    m = scisp.random(shape[0], shape[1], density=0.2, random_state=rs, format='csr', dtype=np.float64)
    m = (m + m.T).tocsr()
    m.sort_indices()
    ref = ns['downsample'](m, block, method)
    got = oracle_lib.downsample_dense(m, block, method)
    np.testing.assert_allclose(got, np.asarray(ref.todense()), rtol=1e-13, atol=0)


PY2_PROBE = r'''
import cPickle, sys
class ContactMap:
    pass
class SeqOrder:
    pass
o = SeqOrder()
o.__dict__.update({'_positions': [0, 1, 2], 'flag': True})
c = ContactMap()
c.__dict__.update({'strong': None, 'bam_file': 'hic2ctg.bam', 'min_mapq': 60, 'min_extent': 50000, 'precount': False,
                   'big': 2 ** 40, 'ratio': 0.1, 'tiny': 1e-300, 'tab': 'a\tb\n\\x', 'quote': "it's", 'tuple': (1, 'x', None), 'empty': (),
                   'enzymes': ['MluCI', 'Sau3AI'], 'order': o, 'again': o})
sys.stdout.write(cPickle.dumps(c))
'''


def _py2():
    for name in ('python2.7', 'python2'):
        exe = shutil.which(name)
        if exe:
            return exe
    pytest.skip('no Python 2 interpreter on this box')


def test_py2_pickle_model_against_a_real_python2():
    """(f)2: tools/make_py2_pickle.py ASSEMBLES the protocol-0 stream of tests/golden/contact_map_py2.p.gz from a model of what
    CPython 2.7's cPickle emits.  With a Python 2 at hand the model is checked against the real thing: the same object graph --
    old-style instances with shared references, str escapes, ints / longs / bools / floats / None, tuples, lists, dicts -- pickled
    by cPickle must load to the same values as the model's stream, and the model's opcode sequence must be cPickle's."""
    import pickle
    import pickletools
    exe = _py2()
    real = subprocess.run([exe, '-c', PY2_PROBE], stdout=subprocess.PIPE, check=True).stdout
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    try:
        mk = importlib.import_module('make_py2_pickle')
    finally:
        sys.path.pop(0)
    order = mk.OldStyle('__main__', 'SeqOrder', {'_positions': [0, 1, 2], 'flag': True})
    attrs = {'strong': None, 'bam_file': 'hic2ctg.bam', 'min_mapq': 60, 'min_extent': 50000, 'precount': False,
             'big': 2 ** 40, 'ratio': 0.1, 'tiny': 1e-300, 'tab': 'a\tb\n\\x', 'quote': "it's", 'tuple': (1, 'x', None), 'empty': (),
             'enzymes': ['MluCI', 'Sau3AI'], 'order': order, 'again': order}
    model = mk.Py2Pickler().dump(mk.OldStyle('__main__', 'ContactMap', attrs))

    class _U(pickle.Unpickler):
        def find_class(self, module, name):
            return type(name, (object,), {})

    import io

    def load(raw):
        obj = _U(io.BytesIO(raw), encoding='latin1').load()
        d = dict(obj.__dict__)
        assert d['order'] is d['again']                      # the shared reference survives
        d['order'] = dict(d['order'].__dict__)
        d.pop('again')
        return d

    assert load(real) == load(model)
    # same opcodes in the same order, dict-iteration order aside: compare the multiset of (opcode, argument) pairs and the framing
    ops = lambda raw: sorted((op.name, repr(arg)) for op, arg, _pos in pickletools.genops(raw) if op.name not in ('PUT', 'GET'))
    assert ops(real) == ops(model)


def test_py2_written_contact_map_loads():
    """(f)2, the full pin: a ContactMap written by the REFERENCE under Python 2 (`mzd.io_utils.save_object`, io_utils.py:12-32).
    Needs a Python 2 with numpy / scipy and the reference tree; `B3C_PY2_CONTACT_MAP` may instead name such a file made elsewhere."""
    from bin3c_amd.mzd import io_utils
    path = os.environ.get('B3C_PY2_CONTACT_MAP')
    if not path:
        pytest.skip('no reference-written Python-2 contact map at hand (set B3C_PY2_CONTACT_MAP to one)')
    cm = io_utils.load_object(path)
    assert cm.seq_map.shape[0] == cm.seq_map.shape[1] == len(cm.seq_info)
    assert cm.order.count_accepted() <= cm.seq_map.shape[0]
