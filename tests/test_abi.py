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


def _gen():
    import importlib.util
    spec = importlib.util.spec_from_file_location('gen_abi_stub', os.path.join(ROOT, 'tools', 'gen_abi_stub.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_struct_layouts_agree_everywhere():
    """include/b3c.h (parsed), libb3c.so (b3c_abi_layout = offsetof as compiled), bin3c_amd/_native.py (ctypes) and the
    stub printed in INTEGRATION.md: one layout, field names included.  A field added in one place only fails here."""
    import ctypes as C
    import __graft_entry__
    __graft_entry__.build()
    from bin3c_amd import _native
    lib = _native.load()
    g = _gen()
    hdr = g.parse_header()
    # the stub of INTEGRATION.md, executed as written (class definitions only: nothing in the block touches a library)
    block = g.doc_block()
    assert block == g.stub_text(), 'INTEGRATION.md section 1 is stale: run `python tools/gen_abi_stub.py --write`'
    ns = {'C': C}
    exec(block, ns)
    for cname, pyname, which in g.STRUCTS:
        want = g.layout(hdr[cname])
        buf = (C.c_int32 * 64)()
        k = lib.b3c_abi_layout(which, buf, 64)
        assert list(buf[:k]) == want, (cname, 'library vs header', list(buf[:k]), want)
        native = _native.ABI_STRUCTS[which]
        assert _native.struct_layout(native) == want, (cname, '_native.py vs header')
        assert [f[0] for f in native._fields_] == [f[0] for f in hdr[cname]], (cname, 'field names in _native.py')
        doc = ns[pyname]
        assert _native.struct_layout(doc) == want, (cname, 'INTEGRATION.md vs header')
        assert [f[0] for f in doc._fields_] == [f[0] for f in hdr[cname]], (cname, 'field names in INTEGRATION.md')
    ns['check_abi'](lib)                              # the stub's own guard accepts the library
    assert lib.b3c_abi_layout(99, buf, 64) < 0 and lib.b3c_abi_layout(0, buf, 3) < 0


def test_layout_guard_rejects_a_drifted_binding():
    """the load-time guard: a binding with one field fewer than the library is refused"""
    import ctypes as C
    import pytest
    from bin3c_amd import _native
    lib = _native.load()

    class Short(C.Structure):
        _fields_ = _native.KRInfo._fields_[:12]       # round 2's INTEGRATION.md stub: 12 of the 18 fields
    saved = _native.ABI_STRUCTS[0]
    _native.ABI_STRUCTS[0] = Short
    try:
        with pytest.raises(ImportError):
            _native.check_abi(lib)
    finally:
        _native.ABI_STRUCTS[0] = saved
