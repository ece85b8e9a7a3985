"""Container-only helper of tools/kr_spread.py: run the REFERENCE's kr_biostochastic text (sliced from
/root/reference/mzd/sparse_utils.py at run time, never stored) on the cases of an .npz and save x / n_iter per case.
Written to run unchanged under /usr/bin/python3 (scipy 1.15, numpy 2.2) and /opt/conda/bin/python3.9 (scipy 1.7.1,
numpy 1.26): the two environments whose results are compared.   usage: kr_spread_ref.py cases.npz out.npz"""
import logging
import re
import sys

import numpy as np
import scipy
import scipy.sparse as scisp

REF = '/root/reference/mzd/sparse_utils.py'


class _Cap(logging.Handler):
    """keeps the messages of the reference's logger: 'It took {} iterations ...' (sparse_utils.py:218) carries n_iter"""

    def __init__(self):
        logging.Handler.__init__(self)
        self.msgs = []

    def emit(self, record):
        self.msgs.append(record.getMessage())


CAP = _Cap()


def ref_kr():
    lines = open(REF).read().split('\n')
    start = [i for i, l in enumerate(lines) if re.match(r'def kr_biostochastic\(', l)][0]
    end = [i for i in range(start + 1, len(lines)) if lines[i] and not lines[i][0].isspace() and not lines[i].startswith('#')][0]

    def is_hermitian(m, tol=1e-6):          # the predicate of sparse_utils.py:18 evaluated sparsely (warn-only; `.H` is gone from scipy >= 1.8)
        return (abs(m - m.T.conj()) >= tol).nnz == 0
    log = logging.getLogger('kr_spread')
    log.setLevel(logging.DEBUG)
    log.addHandler(CAP)
    log.propagate = False
    ns = {'np': np, 'scisp': scisp, 'logger': log, 'is_hermitian': is_hermitian}
    exec('\n'.join(lines[start:end]), ns)
    return ns['kr_biostochastic']


def main():
    kr = ref_kr()
    z = np.load(sys.argv[1])
    out = {'env': np.array('python %d.%d scipy %s numpy %s' % (sys.version_info[0], sys.version_info[1], scipy.__version__, np.__version__))}
    k = 0
    while 'ip_%d' % k in z:
        ip, ix, a = z['ip_%d' % k], z['ix_%d' % k], z['a_%d' % k]
        n = ip.shape[0] - 1
        m = scisp.csr_matrix((a, ix, ip), shape=(n, n))
        try:
            CAP.msgs = []
            _, x = kr(m, max_iter=int(z['max_iter']))
            out['x_%d' % k] = np.asarray(x).ravel()
            out['ok_%d' % k] = np.array(1)
            its = [int(re.search(r'(\d+)', t).group(1)) for t in CAP.msgs if t.startswith('It took')]
            out['it_%d' % k] = np.array(its[-1] if its else -1)
        except RuntimeError as ex:
            out['ok_%d' % k] = np.array(0)
        k += 1
    np.savez(sys.argv[2], **out)


if __name__ == '__main__':
    main()
