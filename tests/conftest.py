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


class _EnvPatch(object):
    """pytest's monkeypatch, with one addition: libb3c reads its B3C_* switches once per context (b3c_ctx_create), so a test
    that flips one has every open context read them again (b3c_ctx_reload_env)."""

    def __init__(self, mp):
        self._mp = mp

    def __getattr__(self, name):
        return getattr(self._mp, name)

    @staticmethod
    def _reload():
        from bin3c_amd import _native
        _native.reload_env_all()

    def setenv(self, name, value, prepend=None):
        self._mp.setenv(name, value, prepend)
        if name.startswith('B3C_'):
            self._reload()

    def delenv(self, name, raising=True):
        self._mp.delenv(name, raising)
        if name.startswith('B3C_'):
            self._reload()


@pytest.fixture
def monkeypatch(monkeypatch):
    mp = _EnvPatch(monkeypatch)
    yield mp
    monkeypatch.undo()
    mp._reload()
