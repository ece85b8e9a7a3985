import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def oracle_lib():
    """The CPU oracle (test infrastructure); compiled on first use with gcc."""
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope='session')
def ctx():
    """libb3c context on cuda:0 -- fails loudly if the HIP library is missing or there is no GPU."""
    from bin3c_amd.pipeline import make_context
    c = make_context(0)
    yield c
    c.close()
