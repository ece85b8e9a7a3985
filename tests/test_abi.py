"""CPU suite: libb3c.so loads and exports every symbol include/b3c.h declares (no compute calls)."""
import os
import re

from conftest import ROOT


def _declared():
    with open(os.path.join(ROOT, 'include', 'b3c.h')) as fh:
        txt = fh.read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(b3c_[a-z0-9_]+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()
    from bin3c_amd import _native
    lib = _native.load()
    names = _declared()
    assert len(names) >= 15
    for name in names:
        assert hasattr(lib, name), 'libb3c.so does not export {}'.format(name)
    bound = {s[0] for s in _native.SYMBOLS}
    assert bound == set(names), 'ctypes table and header disagree: {}'.format(bound ^ set(names))


def test_version_string():
    from bin3c_amd import _native
    assert b'gfx950' in _native.load().b3c_version()


def test_no_gpu_fails_loudly():
    """Without a device the context must refuse (no silent CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        return
    import pytest
    from bin3c_amd import _native
    with pytest.raises(_native.B3CError):
        _native.Context(0)
