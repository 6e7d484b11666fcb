"""CPU: the C-ABI library is built for gfx950, loads, and exports every symbol the header declares.
No compute is called (no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def built():
    import __graft_entry__
    __graft_entry__.build()
    from yue_amd import _shim
    return _shim


def test_library_exports_header_symbols(built):
    hdr = open(os.path.join(ROOT, 'include', 'yue_hip.h')).read()
    declared = sorted(set(re.findall(r'\b(yue_[a-z0-9_]+)\s*\(', hdr)))
    assert declared, 'no declarations parsed'
    lib = ctypes.CDLL(built.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), 'libyue_hip.so lacks ' + name
    assert sorted(built.SYMBOLS) == declared
    assert lib.yue_version() >= 1


def test_code_object_targets_gfx950(built):
    blob = open(built.LIB_PATH, 'rb').read()
    assert b'gfx950' in blob


def test_product_has_no_cpu_fallback(built, monkeypatch):
    # the product path must fail loudly when the HIP extension is missing
    monkeypatch.setattr(built, 'LIB_PATH', '/nonexistent/libyue_hip.so')
    monkeypatch.setattr(built, '_lib', None)
    with pytest.raises(built.YueHipError):
        built.load_library()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'yue_amd')
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.hpp', '.h')):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(import|from)\s+oracle\b', src, re.M), f
                assert 'liboracle' not in src, f
