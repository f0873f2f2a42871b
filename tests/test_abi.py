"""CPU: the C-ABI shared library builds (hipcc cross-compiles gfx950 without a GPU), loads, and exports exactly the
entry points include/cat_hip.h declares (no compute calls here)."""
import os
import re

from cat_amd import _build, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, 'include', 'cat_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(cat_[a-z0-9_]+)\s*\(', text)))


def test_library_builds_and_exports_header_symbols():
    path = _build.build(verbose=False)
    assert os.path.exists(path)
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/cat_hip.h but not exported by libcat_hip.so'
    assert lib.cat_hip_version() >= 2
    assert lib.cat_hip_last_error() is not None


def test_ctypes_signatures_cover_the_header():
    assert sorted(_lib.SIGNATURES) == _declared()


def test_workspace_queries_are_pure_host_functions():
    import ctypes as C
    lib = _lib.load()
    g = _lib.ConvGeom(16, 64, 64, 77, 80, 64, 64, 18, 20, 5, 5, 1, 2, 1, 0, 0.0, 20)
    assert lib.cat_conv2d_wgrad_ws_bytes(C.byref(g)) > 0
    n = _lib.NormGeom(16, 4096, 77, 80, _lib.NORM_BATCH, 1e-5, 0.1, 1, 0.0)
    assert lib.cat_norm_ws_bytes(C.byref(n)) > 0
    assert lib.cat_ka_ws_bytes(16) > 0


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under cat_amd/ may import it (or torch's CPU ops as a fallback path)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'cat_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in re.sub(r'""".*?"""', '', src, flags=re.S), f
